// ref_constants.hpp -- the reference's named constants, each ONCE: every #define of /root/reference/dsp/*.cpp and
// dsp/*.h that enters this library's arithmetic (host set-up math and kernels alike), under a name of its own with the
// place it comes from, and the window coefficients.  The code uses these names only; csdr__constants() (capi_core.hip)
// exports the table, and tests/test_reference_constants.py -- build container only, the reference never travels --
// compares every value with the reference's TEXT (and with the CPU checker's own table): a constant mistyped in both
// restatements cannot survive that.
#pragma once
#include "resampler_kernels.h"      // RS_PTS, RS_PERIODS, RS_LEN (fractresampler.cpp:50-57)
#include "frontend_kernels.h"       // NB_MAX_WIDTH, NB_HIST (noiseproc.cpp:49-51)
#include "postchain.h"              // PC_FIR_MAX, PC_AGC_RING (fir.h:16, agc.h:16)

namespace csdr {
namespace refc {

constexpr double AGC_DELAY_TIMECONST = .015;          // agc.cpp:50
constexpr double AGC_WINDOW_TIMECONST = .018;         // agc.cpp:53
constexpr double AGC_ATTACK_RISE_TIMECONST = .002;    // agc.cpp:57
constexpr double AGC_ATTACK_FALL_TIMECONST = .005;    // agc.cpp:58
constexpr double AGC_DECAY_RISEFALL_RATIO = .3;       // agc.cpp:60
constexpr double AGC_RELEASE_TIMECONST = .05;         // agc.cpp:64
constexpr double AGC_OUTSCALE = 0.7;                  // agc.cpp:67
constexpr double AGC_MAX_AMPLITUDE = 32767.0;         // agc.cpp:69
constexpr double AGC_MAX_MANUAL_AMPLITUDE = 32767.0;  // agc.cpp:70
constexpr double AGC_MIN_CONSTANT = 3.2767e-4;        // agc.cpp:72
constexpr double AM_DC_ALPHA = 0.99;                  // amdemod.cpp:44 DC_ALPHA
constexpr double DCV_MIN_OUTPUT_RATE = 7900.0 * 2.0;  // downconvert.cpp:52
constexpr double FF_WIN_A0 = 0.3635819, FF_WIN_A1 = 0.4891775, FF_WIN_A2 = 0.1365995, FF_WIN_A3 = 0.0106411;   // fastfir.cpp:95-98
constexpr double FFT_K_AMPMAX = 32767.0;              // fft.cpp:19
constexpr double FFT_K_MAXDB = 0.0;                   // fft.cpp:20
constexpr double FFT_K_MINDB = -220.0;                // fft.cpp:21
constexpr double FFT_OVER_LIMIT = 32000.0;            // fft.cpp:23
constexpr int FFT_MAX_SIZE = 65536, FFT_MIN_SIZE = 512;   // fft.h:21-22
constexpr double FM_PLL_RANGE = 6000.0;               // fmdemod.cpp:45 FMPLL_RANGE
constexpr double FM_VOICE_BANDWIDTH = 3000.0;         // fmdemod.cpp:46
constexpr double FM_PLL_ZETA = .707;                  // fmdemod.cpp:49
constexpr double FM_DC_ALPHA = 0.01;                  // fmdemod.cpp:51 FMDC_ALPHA
constexpr double FM_MAX_OUT = 25000.0;                // fmdemod.cpp:53 MAX_FMOUT
constexpr double FM_SQUELCH_MAX = 5000.0;             // fmdemod.cpp:55
constexpr double FM_SQUELCHAVE_TIMECONST = .02;       // fmdemod.cpp:56
constexpr double FM_SQUELCH_HYSTERESIS = 100.0;       // fmdemod.cpp:57
constexpr double RS_MAX_SOUNDCARDVAL = 32767.0;       // fractresampler.cpp:59
constexpr double RS_WIN_A0 = 0.35875, RS_WIN_A1 = 0.48829, RS_WIN_A2 = 0.14128, RS_WIN_A3 = 0.01168;           // fractresampler.cpp:102-105
constexpr double NB_MAGAVE_TIME = 0.005;              // noiseproc.cpp:53
constexpr double SAM_DC_ALPHA = 0.99;                 // samdemod.cpp:45 DC_ALPHA
constexpr double SAM_PLL_BW = 100.0;                  // samdemod.cpp:47
constexpr double SAM_PLL_ZETA = .707;                 // samdemod.cpp:48
constexpr double SAM_PLL_LIMIT = 1000.0;              // samdemod.cpp:49
constexpr double SM_ATTACK_TIMECONST = .01;           // smeter.cpp:42
constexpr double SM_DECAY_TIMECONST = .5;             // smeter.cpp:43
constexpr double SM_CALIBRATION = 5.0;                // smeter.cpp:45 SMETER_CALIBRATION
constexpr double SM_MAX_PWR = 32767.0 * 32767.0;      // smeter.cpp:47
constexpr int DEMOD_MAX_INBUFSIZE = 250000;           // demodulator.h:30
// FMPLL_BW is written VOICE_BANDWIDTH*2.0 without parentheses (fmdemod.cpp:48) and used as 2.0*FMPLL_ZETA*FMPLL_BW*norm:
// the product is evaluated left to right, so the code spells FM_PLL_ZETA * FM_VOICE_BANDWIDTH * 2.0 in that order
constexpr double FM_PLL_BW = FM_VOICE_BANDWIDTH * 2.0;
// fp32 forms the kernels use (each the correctly rounded value of the constant above)
constexpr float AGC_MIN_CONSTANT_F = (float)AGC_MIN_CONSTANT;
constexpr float AGC_LOG10_MAX_AMPLITUDE_F = (float)4.5154366811416989;     // log10(32767.0): agc.cpp:200 `- log10(MAX_AMPLITUDE)`
constexpr float SM_INV_MAX_PWR_F = 1.0f / (32767.0f * 32767.0f);           // smeter.cpp:79 `/ MAX_PWR`
constexpr float FFT_OVER_LIMIT_F = (float)FFT_OVER_LIMIT;
constexpr float RS_MAX_SOUNDCARDVAL_F = (float)RS_MAX_SOUNDCARDVAL;

struct Entry { const char *name; double value; };
// name = "<reference file>:<its #define>" (window coefficients: "<file>:WIN_A<k>"); "derived:" entries are values this
// library computes from them once and keeps as literals
constexpr Entry TABLE[] = {
    {"agc.cpp:DELAY_TIMECONST", AGC_DELAY_TIMECONST}, {"agc.cpp:WINDOW_TIMECONST", AGC_WINDOW_TIMECONST},
    {"agc.cpp:ATTACK_RISE_TIMECONST", AGC_ATTACK_RISE_TIMECONST}, {"agc.cpp:ATTACK_FALL_TIMECONST", AGC_ATTACK_FALL_TIMECONST},
    {"agc.cpp:DECAY_RISEFALL_RATIO", AGC_DECAY_RISEFALL_RATIO}, {"agc.cpp:RELEASE_TIMECONST", AGC_RELEASE_TIMECONST},
    {"agc.cpp:AGC_OUTSCALE", AGC_OUTSCALE}, {"agc.cpp:MAX_AMPLITUDE", AGC_MAX_AMPLITUDE},
    {"agc.cpp:MAX_MANUAL_AMPLITUDE", AGC_MAX_MANUAL_AMPLITUDE}, {"agc.cpp:MIN_CONSTANT", AGC_MIN_CONSTANT},
    {"agc.h:MAX_DELAY_BUF", PC_AGC_RING}, {"amdemod.cpp:DC_ALPHA", AM_DC_ALPHA},
    {"downconvert.cpp:MIN_OUTPUT_RATE", DCV_MIN_OUTPUT_RATE},
    {"fastfir.cpp:WIN_A0", FF_WIN_A0}, {"fastfir.cpp:WIN_A1", FF_WIN_A1}, {"fastfir.cpp:WIN_A2", FF_WIN_A2}, {"fastfir.cpp:WIN_A3", FF_WIN_A3},
    {"fft.cpp:K_AMPMAX", FFT_K_AMPMAX}, {"fft.cpp:K_MAXDB", FFT_K_MAXDB}, {"fft.cpp:K_MINDB", FFT_K_MINDB}, {"fft.cpp:OVER_LIMIT", FFT_OVER_LIMIT},
    {"fft.h:MAX_FFT_SIZE", FFT_MAX_SIZE}, {"fft.h:MIN_FFT_SIZE", FFT_MIN_SIZE}, {"fir.h:MAX_NUMCOEF", PC_FIR_MAX},
    {"fmdemod.cpp:FMPLL_RANGE", FM_PLL_RANGE}, {"fmdemod.cpp:VOICE_BANDWIDTH", FM_VOICE_BANDWIDTH}, {"fmdemod.cpp:FMPLL_BW", FM_PLL_BW},
    {"fmdemod.cpp:FMPLL_ZETA", FM_PLL_ZETA}, {"fmdemod.cpp:FMDC_ALPHA", FM_DC_ALPHA}, {"fmdemod.cpp:MAX_FMOUT", FM_MAX_OUT},
    {"fmdemod.cpp:SQUELCH_MAX", FM_SQUELCH_MAX}, {"fmdemod.cpp:SQUELCHAVE_TIMECONST", FM_SQUELCHAVE_TIMECONST},
    {"fmdemod.cpp:SQUELCH_HYSTERESIS", FM_SQUELCH_HYSTERESIS},
    {"fractresampler.cpp:SINC_PERIOD_PTS", RS_PTS}, {"fractresampler.cpp:SINC_PERIODS", RS_PERIODS}, {"fractresampler.cpp:SINC_LENGTH", RS_LEN},
    {"fractresampler.cpp:MAX_SOUNDCARDVAL", RS_MAX_SOUNDCARDVAL},
    {"fractresampler.cpp:WIN_A0", RS_WIN_A0}, {"fractresampler.cpp:WIN_A1", RS_WIN_A1}, {"fractresampler.cpp:WIN_A2", RS_WIN_A2}, {"fractresampler.cpp:WIN_A3", RS_WIN_A3},
    {"noiseproc.cpp:MAX_WIDTH", NB_MAX_WIDTH}, {"noiseproc.cpp:MAX_AVE", NB_HIST}, {"noiseproc.cpp:MAGAVE_TIME", NB_MAGAVE_TIME},
    {"samdemod.cpp:DC_ALPHA", SAM_DC_ALPHA}, {"samdemod.cpp:PLL_BW", SAM_PLL_BW}, {"samdemod.cpp:PLL_ZETA", SAM_PLL_ZETA}, {"samdemod.cpp:PLL_LIMIT", SAM_PLL_LIMIT},
    {"smeter.cpp:ATTACK_TIMECONST", SM_ATTACK_TIMECONST}, {"smeter.cpp:DECAY_TIMECONST", SM_DECAY_TIMECONST},
    {"smeter.cpp:SMETER_CALIBRATION", SM_CALIBRATION}, {"smeter.cpp:MAX_PWR", SM_MAX_PWR},
    {"demodulator.h:MAX_INBUFSIZE", DEMOD_MAX_INBUFSIZE},
    {"derived:f32(agc.cpp:MIN_CONSTANT)", AGC_MIN_CONSTANT_F}, {"derived:f32(log10(agc.cpp:MAX_AMPLITUDE))", AGC_LOG10_MAX_AMPLITUDE_F},
    {"derived:f32(1/smeter.cpp:MAX_PWR)", SM_INV_MAX_PWR_F},
};
constexpr int TABLE_N = (int)(sizeof(TABLE) / sizeof(TABLE[0]));

}  // namespace refc
}  // namespace csdr
