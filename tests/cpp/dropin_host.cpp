// Stand-in for the reference host's use of the dsp/ classes (interface/sdrinterface.cpp:878-922:
// CFft::PutInDisplayFFT + CDemodulator::ProcessData on 256-sample packets; interface/soundout.cpp:204:
// CFractResampler; the blanker CNoiseProc runs in place in front of both, sdrinterface.cpp:884),
// compiled against the drop-in headers with plain g++.
//   dropin_host <in.bin> <out_prefix> <mode> <fs> <freq> [<switch_at> <new_fs>]
// With the last two: at sample <switch_at> the radio's bandwidth is switched the way CSdrInterface does it
// (interface/sdrinterface.cpp:751-755): SetFftSize (-> CFft::SetFFTParams with the new rate), then
// CDemodulator::SetInputSampleRate, and NO SetDemod.
// in.bin: interleaved doubles.  Writes <prefix>.audio (doubles), <prefix>.spec (int32 x 700),
// <prefix>.meta (text).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "dsp/demodulator.h"
#include "dsp/fft.h"
#include "dsp/fastfir.h"
#include "dsp/downconvert.h"
#include "dsp/agc.h"
#include "dsp/smeter.h"
#include "dsp/fir.h"
#include "dsp/iir.h"
#include "dsp/amdemod.h"
#include "dsp/samdemod.h"
#include "dsp/fmdemod.h"
#include "dsp/ssbdemod.h"
#include "dsp/fractresampler.h"
#include "dsp/noiseproc.h"

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: dropin_host in.bin out_prefix mode fs freq\n"); return 2; }
    const int mode = std::atoi(argv[3]);
    const double fs = std::atof(argv[4]), freq = std::atof(argv[5]);
    FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    std::vector<TYPECPX> x;
    TYPECPX s;
    while (std::fread(&s, sizeof(s), 1, fi) == 1) x.push_back(s);
    std::fclose(fi);

    CDemodulator demod;                      // by-value members, as CSdrInterface holds them
    CFft fft;
    CFractResampler rs;
    CNoiseProc nb;
    tDemodInfo info;
    info.HiCut = 5000; info.HiCutmin = 5000; info.HiCutmax = 15000; info.LowCut = -5000; info.LowCutmin = -15000;
    info.LowCutmax = -5000; info.FilterClickResolution = 100; info.Offset = 0; info.SquelchValue = 0;
    info.AgcSlope = 0; info.AgcThresh = -100; info.AgcManualGain = 30; info.AgcDecay = 200;
    info.AgcOn = true; info.AgcHangOn = false; info.Symetric = true; info.txt = "FM";
    demod.SetInputSampleRate(fs);
    demod.SetDemod(mode, info);
    demod.SetDemodFreq(freq);
    fft.SetFFTParams(4096, false, 0.0, fs);
    fft.SetFFTAve(1);
    rs.Init(8192);
    nb.SetupBlanker(true, 40.0, 10.0, fs);

    std::vector<double> audio, snd(16384);
    std::vector<TYPEREAL> out(8192);         // the host's stack buffer, sdrinterface.cpp:910
    size_t fftpos = 0;
    int total = 0, rtotal = 0;
    double rate = demod.GetOutputRate() / 48000.0;
    const size_t switch_at = argc >= 8 ? (size_t)std::atoll(argv[6]) : (size_t)-1;
    for (size_t i = 0; i + 256 <= x.size(); i += 256) {
        if (i == switch_at) {
            const double nfs = std::atof(argv[7]);
            fft.SetFFTParams(4096, false, 0.0, nfs);
            demod.SetInputSampleRate(nfs);
            rate = demod.GetOutputRate() / 48000.0;     // m_pSoundCardOut->ChangeUserDataRate(GetOutputRate())
            fftpos = i;
        }
        nb.ProcessBlanker(256, &x[i], &x[i]);       // in place, as the host does
        if (i + 256 - fftpos >= 4096) { fft.PutInDisplayFFT(4096, &x[fftpos]); fftpos += 4096; }
        const int n = demod.ProcessData(256, &x[i], out.data());
        if (n > 0) {
            total += n;
            const int k = n > 8192 ? 8192 : n;
            audio.insert(audio.end(), out.begin(), out.begin() + k);
            rtotal += rs.Resample(k, rate, out.data(), snd.data());
        }
    }
    std::vector<qint32> pix(700);
    const bool ov = fft.GetScreenIntegerFFTData(255, 700, 0.0, -160.0, -900000, 900000, pix.data());
    std::string p(argv[2]);
    FILE *fo = std::fopen((p + ".audio").c_str(), "wb");
    std::fwrite(audio.data(), sizeof(double), audio.size(), fo); std::fclose(fo);
    fo = std::fopen((p + ".spec").c_str(), "wb");
    std::fwrite(pix.data(), sizeof(qint32), pix.size(), fo); std::fclose(fo);
    fo = std::fopen((p + ".meta").c_str(), "w");
    std::fprintf(fo, "%d %d %.6f %.6f %d\n", total, rtotal, demod.GetOutputRate(), demod.GetSMeterAve(), ov ? 1 : 0);
    std::fclose(fo);
    return 0;
}
