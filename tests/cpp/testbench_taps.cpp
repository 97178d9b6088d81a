// The drop-in CDemodulator compiled the way a host WITH the test bench compiles it (-DCSDR_DROPIN_TESTBENCH, the host's
// own gui/testbench.h on the include path -- here the stub under tests/cpp/stub): every pass hands PROFILE_1..4 to
// g_pTestBench->DisplayData in the reference's order (dsp/demodulator.cpp:175,180,187,208).  Prints one line per call.
#define CSDR_DROPIN_TESTBENCH
#include <cstdio>
#include <cmath>
#include <vector>
#include "dsp/demodulator.h"

CTestBench *g_pTestBench = nullptr;

int main()
{
    CTestBench tb;
    g_pTestBench = &tb;
    CDemodulator demod;
    tDemodInfo info;
    info.HiCut = 2800; info.HiCutmin = 500; info.HiCutmax = 20000; info.LowCut = 100; info.LowCutmin = 0; info.LowCutmax = 200;
    info.FilterClickResolution = 100; info.Offset = 0; info.SquelchValue = 0;
    info.AgcSlope = 0; info.AgcThresh = -100; info.AgcManualGain = 30; info.AgcDecay = 200;
    info.AgcOn = true; info.AgcHangOn = false; info.Symetric = false; info.txt = "USB";
    demod.SetInputSampleRate(2e6);
    demod.SetDemod(DEMOD_USB, info);
    demod.SetDemodFreq(-100e3);
    std::vector<TYPECPX> x(256);
    std::vector<TYPEREAL> out(8192);
    int total = 0;
    for (int call = 0; call < 78 * 6; call++) {              // six windows of 19968 samples in 256-sample host calls
        for (int i = 0; i < 256; i++) {
            const double t = (call * 256 + i) / 2e6, ph = 2.0 * 3.14159265358979323846 * 101200.0 * t;
            x[i].re = 3000.0 * std::cos(ph); x[i].im = 3000.0 * std::sin(ph);
        }
        total += demod.ProcessData(256, x.data(), out.data());
    }
    for (const auto &c : tb.calls) std::printf("%d %d %d %.1f %.6g\n", c.profile, c.n, c.cpx ? 1 : 0, c.rate, c.first);
    std::printf("total %d\n", total);
    return 0;
}
