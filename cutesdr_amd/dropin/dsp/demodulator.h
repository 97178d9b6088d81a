// dsp/demodulator.h drop-in: tDemodInfo and class CDemodulator with the reference's public surface
// (reference dsp/demodulator.h:20-100).  The whole chain runs device-resident behind
// csdr_demod_* of libcutesdr_mi.
#ifndef DEMODULATOR_H
#define DEMODULATOR_H
// the same headers the reference header re-exports (dsp/demodulator.h:11-18): host code includes only this one
// and still names CIir, CFir, CAgc ... (interface/sdrinterface.h:15,178)
#include "dsp/downconvert.h"
#include "dsp/fastfir.h"
#include "smeter.h"
#include "dsp/agc.h"
#include "dsp/amdemod.h"
#include "dsp/samdemod.h"
#include "dsp/fmdemod.h"
#include "dsp/ssbdemod.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QString>
typedef QString csdr_label_t;
#else
#include <string>
typedef std::string csdr_label_t;
#endif

// A host that keeps the reference's test bench defines CSDR_DROPIN_TESTBENCH (and has its own gui/testbench.h on the include
// path): every pass of the chain then hands its four test points to g_pTestBench->DisplayData(n, buf, m_OutputRate,
// PROFILE_1..4) exactly where the reference does (dsp/demodulator.cpp:175,180,187,208), through csdr_demod_set_taps.  With the
// taps on every pass waits for its results -- a diagnostic build, as a host with the test bench open is.
// CSDR_DROPIN_TESTBENCH_MASK (default 15) selects the profiles: bit k-1 = PROFILE_k.
#ifdef CSDR_DROPIN_TESTBENCH
#include "gui/testbench.h"
#ifndef CSDR_DROPIN_TESTBENCH_MASK
#define CSDR_DROPIN_TESTBENCH_MASK 15
#endif
#endif

#define DEMOD_AM 0
#define DEMOD_SAM 1
#define DEMOD_FM 2
#define DEMOD_USB 3
#define DEMOD_LSB 4
#define DEMOD_CWU 5
#define DEMOD_CWL 6
#define NUM_DEMODS 7
#define MAX_INBUFSIZE 250000
#define MAX_MAGBUFSIZE 32000

typedef struct _sdmd
{
    int HiCut;
    int HiCutmin;
    int HiCutmax;
    int LowCut;
    int LowCutmin;
    int LowCutmax;
    int FilterClickResolution;
    int Offset;
    int SquelchValue;
    int AgcSlope;
    int AgcThresh;
    int AgcManualGain;
    int AgcDecay;
    bool AgcOn;
    bool AgcHangOn;
    bool Symetric;
    csdr_label_t txt;
} tDemodInfo;

class CDemodulator
{
public:
    CDemodulator() : m_h(csdr_dropin_handle(csdr_demod_create(CSDR_DEVICE, CSDR_FASTFIR_SIZE), "CDemodulator"))
    {
#ifdef CSDR_DROPIN_TESTBENCH
        csdr_dropin_count(csdr_demod_set_taps(m_h, CSDR_DROPIN_TESTBENCH_MASK, &CDemodulator::TestBenchTap, this), "CDemodulator taps");
#elif defined(CSDR_DROPIN_DEFERRED)
        // a live receiver, whose audio goes through the sound card's queue anyway: every pass hands over the previous
        // pass's audio and ProcessData never waits for the device (csdr_demod_set_deferred; one window = 10 ms later)
        csdr_dropin_count(csdr_demod_set_deferred(m_h, 1), "CDemodulator deferred output");
#endif
    }
    virtual ~CDemodulator() { csdr_demod_destroy(m_h); }
    CDemodulator(const CDemodulator &) = delete;
    CDemodulator &operator=(const CDemodulator &) = delete;

    void SetInputSampleRate(TYPEREAL InputRate)
    { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_demod_set_input_rate(m_h, InputRate), "CDemodulator::SetInputSampleRate"); }
    double GetOutputRate() { CSDR_LOCK(); return csdr_demod_get_output_rate(m_h); }
    double GetSMeterPeak() { CSDR_LOCK(); return csdr_demod_get_smeter_peak(m_h); }
    double GetSMeterAve() { CSDR_LOCK(); return csdr_demod_get_smeter_ave(m_h); }
    void SetDemod(int Mode, tDemodInfo CurrentDemodInfo)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        csdr_demod_info i;
        i.HiCut = CurrentDemodInfo.HiCut; i.HiCutmin = CurrentDemodInfo.HiCutmin; i.HiCutmax = CurrentDemodInfo.HiCutmax;
        i.LowCut = CurrentDemodInfo.LowCut; i.LowCutmin = CurrentDemodInfo.LowCutmin; i.LowCutmax = CurrentDemodInfo.LowCutmax;
        i.FilterClickResolution = CurrentDemodInfo.FilterClickResolution; i.Offset = CurrentDemodInfo.Offset;
        i.SquelchValue = CurrentDemodInfo.SquelchValue; i.AgcSlope = CurrentDemodInfo.AgcSlope;
        i.AgcThresh = CurrentDemodInfo.AgcThresh; i.AgcManualGain = CurrentDemodInfo.AgcManualGain;
        i.AgcDecay = CurrentDemodInfo.AgcDecay; i.AgcOn = CurrentDemodInfo.AgcOn; i.AgcHangOn = CurrentDemodInfo.AgcHangOn;
        i.Symetric = CurrentDemodInfo.Symetric;
        csdr_dropin_count(csdr_demod_set_demod(m_h, Mode, &i), "CDemodulator::SetDemod");
    }
    void SetDemodFreq(TYPEREAL Freq) { CSDR_LOCK(); csdr_dropin_count(csdr_demod_set_freq(m_h, Freq), "CDemodulator::SetDemodFreq"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPEREAL *pOutData)
    { std::lock_guard<std::mutex> g(m_Mutex); return csdr_dropin_count(csdr_demod_process_mono(m_h, InLength, &pInData->re, pOutData), "CDemodulator::ProcessData"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData)
    { std::lock_guard<std::mutex> g(m_Mutex); return csdr_dropin_count(csdr_demod_process_stereo(m_h, InLength, &pInData->re, &pOutData->re), "CDemodulator::ProcessData"); }

private:
#ifdef CSDR_DROPIN_TESTBENCH
    static void TestBenchTap(void *, int profile, int n, const double *data, int is_complex, double rate)
    {
        if (!g_pTestBench) return;
        // TYPECPX is two doubles (dsp/datatypes.h); DisplayData takes non-const pointers and does not write through them
        if (is_complex) g_pTestBench->DisplayData(n, reinterpret_cast<TYPECPX *>(const_cast<double *>(data)), rate, profile);
        else g_pTestBench->DisplayData(n, const_cast<TYPEREAL *>(data), rate, profile);
    }
#endif
    csdr_demod *m_h;
    std::mutex m_Mutex;
};
#endif  // DEMODULATOR_H
