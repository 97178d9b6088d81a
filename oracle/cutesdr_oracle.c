/*
 * cutesdr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cutesdr_oracle.h).
 *
 * fp64, single-thread restatement of the reference dsp/ chain.  Written from the
 * algorithm, not transliterated; each block cites the reference
 * file:line whose behaviour it reproduces, including the deliberate quirks.
 */
#include "cutesdr_oracle.h"
#include "../include/csdr_hb_taps.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TWO_PI (2.0 * 3.14159265358979323846)   /* K_2PI, dsp/datatypes.h:44 */
#define ONE_PI (3.14159265358979323846)

static void *zalloc(size_t n) { void *p = calloc(1, n ? n : 1); return p; }

/* ==================================================================================== */
/* The reference's named constants: every #define of the dsp sources and headers that enters the arithmetic below,
 * under a name of its own with the file it comes from, and the window coefficients.  The code below uses
 * these names only; orc_constants() exports the table, and tests/test_reference_constants.py (build
 * container only -- /root/reference never travels) compares every value with the reference's TEXT.       */
/* ==================================================================================== */
#define AGC_DELAY_TIMECONST .015            /* agc.cpp:50 DELAY_TIMECONST */
#define AGC_WINDOW_TIMECONST .018           /* agc.cpp:53 */
#define AGC_ATTACK_RISE_TIMECONST .002      /* agc.cpp:57 */
#define AGC_ATTACK_FALL_TIMECONST .005      /* agc.cpp:58 */
#define AGC_DECAY_RISEFALL_RATIO .3         /* agc.cpp:60 */
#define AGC_RELEASE_TIMECONST .05           /* agc.cpp:64 */
#define AGC_OUTSCALE 0.7                    /* agc.cpp:67 AGC_OUTSCALE */
#define AGC_MAX_AMPLITUDE 32767.0           /* agc.cpp:69 */
#define AGC_MAX_MANUAL_AMPLITUDE 32767.0    /* agc.cpp:70 */
#define AGC_MIN_CONSTANT 3.2767e-4          /* agc.cpp:72 */
#define AGC_RING 2048                       /* agc.h:16 MAX_DELAY_BUF */
#define AM_DC_ALPHA 0.99                    /* amdemod.cpp:44 DC_ALPHA */
#define DCV_MIN_OUTPUT_RATE (7900.0 * 2.0)  /* downconvert.cpp:52 */
#define DC_HB_SCRATCH 32768                 /* downconvert.cpp:54 MAX_HALF_BAND_BUFSIZE */
#define DC_MAX_STAGES 9                     /* downconvert.h:18 MAX_DECSTAGES - 1 (one null at the end of the list) */
#define FF_WIN_A0 0.3635819                 /* fastfir.cpp:95-98: Blackman-Nuttall */
#define FF_WIN_A1 0.4891775
#define FF_WIN_A2 0.1365995
#define FF_WIN_A3 0.0106411
#define FFT_K_AMPMAX 32767.0                /* fft.cpp:19 */
#define FFT_K_MAXDB 0.0                     /* fft.cpp:20 */
#define FFT_K_MINDB -220.0                  /* fft.cpp:21 */
#define FFT_OVER_LIMIT 32000.0              /* fft.cpp:23 */
#define FFT_MAX_SIZE 65536                  /* fft.h:21 */
#define FFT_MIN_SIZE 512                    /* fft.h:22 */
#define FIR_MAX 75                          /* fir.h:16 MAX_NUMCOEF */
#define FM_PLL_RANGE 6000.0                 /* fmdemod.cpp:45 FMPLL_RANGE */
#define FM_VOICE_BANDWIDTH 3000.0           /* fmdemod.cpp:46 */
#define FM_PLL_BW FM_VOICE_BANDWIDTH * 2.0  /* fmdemod.cpp:48 FMPLL_BW (no parentheses there either) */
#define FM_PLL_ZETA .707                    /* fmdemod.cpp:49 */
#define FM_DC_ALPHA 0.01                    /* fmdemod.cpp:51 FMDC_ALPHA */
#define FM_MAX_OUT 25000.0                  /* fmdemod.cpp:53 MAX_FMOUT */
#define FM_SQUELCH_MAX 5000.0               /* fmdemod.cpp:55 */
#define FM_SQUELCHAVE_TIMECONST .02         /* fmdemod.cpp:56 */
#define FM_SQUELCH_HYSTERESIS 100.0         /* fmdemod.cpp:57 */
#define FM_SQBUF 16384                      /* fmdemod.h:14 MAX_SQBUF_SIZE */
#define RS_PTS 10000                        /* fractresampler.cpp:50 SINC_PERIOD_PTS */
#define RS_PERIODS 28                       /* fractresampler.cpp:53 SINC_PERIODS */
#define RS_LEN (RS_PERIODS * RS_PTS + 1)    /* fractresampler.cpp:57 SINC_LENGTH */
#define RS_MAX_SOUNDCARDVAL 32767.0         /* fractresampler.cpp:59 */
#define RS_WIN_A0 0.35875                   /* fractresampler.cpp:102-105: Blackman-Harris */
#define RS_WIN_A1 0.48829
#define RS_WIN_A2 0.14128
#define RS_WIN_A3 0.01168
#define NB_MAX_WIDTH 4096                   /* noiseproc.cpp:49 */
#define NB_MAX_DELAY 4096                   /* noiseproc.cpp:50 */
#define NB_MAX_AVE 32768                    /* noiseproc.cpp:51 */
#define NB_MAGAVE_TIME 0.005                /* noiseproc.cpp:53 */
#define SAM_DC_ALPHA 0.99                   /* samdemod.cpp:45 DC_ALPHA */
#define SAM_PLL_BW 100.0                    /* samdemod.cpp:47 */
#define SAM_PLL_ZETA .707                   /* samdemod.cpp:48 */
#define SAM_PLL_LIMIT 1000.0                /* samdemod.cpp:49 */
#define SM_ATTACK_TIMECONST .01             /* smeter.cpp:42 */
#define SM_DECAY_TIMECONST .5               /* smeter.cpp:43 */
#define SM_CALIBRATION 5.0                  /* smeter.cpp:45 SMETER_CALIBRATION */
#define SM_MAX_PWR (32767.0 * 32767.0)      /* smeter.cpp:47 */
#define DEMOD_BUF 250000                    /* demodulator.h:30 MAX_INBUFSIZE */

/* name = "<reference file>:<its #define>" (window coefficients: "<file>:WIN_A<k>"), value as used below */
int orc_constants(const char **names, double *values, int cap)
{
    static const struct { const char *name; double value; } tab[] = {
        {"agc.cpp:DELAY_TIMECONST", AGC_DELAY_TIMECONST}, {"agc.cpp:WINDOW_TIMECONST", AGC_WINDOW_TIMECONST},
        {"agc.cpp:ATTACK_RISE_TIMECONST", AGC_ATTACK_RISE_TIMECONST}, {"agc.cpp:ATTACK_FALL_TIMECONST", AGC_ATTACK_FALL_TIMECONST},
        {"agc.cpp:DECAY_RISEFALL_RATIO", AGC_DECAY_RISEFALL_RATIO}, {"agc.cpp:RELEASE_TIMECONST", AGC_RELEASE_TIMECONST},
        {"agc.cpp:AGC_OUTSCALE", AGC_OUTSCALE}, {"agc.cpp:MAX_AMPLITUDE", AGC_MAX_AMPLITUDE},
        {"agc.cpp:MAX_MANUAL_AMPLITUDE", AGC_MAX_MANUAL_AMPLITUDE}, {"agc.cpp:MIN_CONSTANT", AGC_MIN_CONSTANT},
        {"agc.h:MAX_DELAY_BUF", AGC_RING}, {"amdemod.cpp:DC_ALPHA", AM_DC_ALPHA},
        {"downconvert.cpp:MIN_OUTPUT_RATE", DCV_MIN_OUTPUT_RATE}, {"downconvert.cpp:MAX_HALF_BAND_BUFSIZE", DC_HB_SCRATCH},
        {"downconvert.h:MAX_DECSTAGES", DC_MAX_STAGES + 1},
        {"fastfir.cpp:WIN_A0", FF_WIN_A0}, {"fastfir.cpp:WIN_A1", FF_WIN_A1}, {"fastfir.cpp:WIN_A2", FF_WIN_A2}, {"fastfir.cpp:WIN_A3", FF_WIN_A3},
        {"fft.cpp:K_AMPMAX", FFT_K_AMPMAX}, {"fft.cpp:K_MAXDB", FFT_K_MAXDB}, {"fft.cpp:K_MINDB", FFT_K_MINDB}, {"fft.cpp:OVER_LIMIT", FFT_OVER_LIMIT},
        {"fft.h:MAX_FFT_SIZE", FFT_MAX_SIZE}, {"fft.h:MIN_FFT_SIZE", FFT_MIN_SIZE}, {"fir.h:MAX_NUMCOEF", FIR_MAX},
        {"fmdemod.cpp:FMPLL_RANGE", FM_PLL_RANGE}, {"fmdemod.cpp:VOICE_BANDWIDTH", FM_VOICE_BANDWIDTH}, {"fmdemod.cpp:FMPLL_BW", FM_PLL_BW},
        {"fmdemod.cpp:FMPLL_ZETA", FM_PLL_ZETA}, {"fmdemod.cpp:FMDC_ALPHA", FM_DC_ALPHA}, {"fmdemod.cpp:MAX_FMOUT", FM_MAX_OUT},
        {"fmdemod.cpp:SQUELCH_MAX", FM_SQUELCH_MAX}, {"fmdemod.cpp:SQUELCHAVE_TIMECONST", FM_SQUELCHAVE_TIMECONST},
        {"fmdemod.cpp:SQUELCH_HYSTERESIS", FM_SQUELCH_HYSTERESIS}, {"fmdemod.h:MAX_SQBUF_SIZE", FM_SQBUF},
        {"fractresampler.cpp:SINC_PERIOD_PTS", RS_PTS}, {"fractresampler.cpp:SINC_PERIODS", RS_PERIODS}, {"fractresampler.cpp:SINC_LENGTH", RS_LEN},
        {"fractresampler.cpp:MAX_SOUNDCARDVAL", RS_MAX_SOUNDCARDVAL},
        {"fractresampler.cpp:WIN_A0", RS_WIN_A0}, {"fractresampler.cpp:WIN_A1", RS_WIN_A1}, {"fractresampler.cpp:WIN_A2", RS_WIN_A2}, {"fractresampler.cpp:WIN_A3", RS_WIN_A3},
        {"noiseproc.cpp:MAX_WIDTH", NB_MAX_WIDTH}, {"noiseproc.cpp:MAX_DELAY", NB_MAX_DELAY}, {"noiseproc.cpp:MAX_AVE", NB_MAX_AVE},
        {"noiseproc.cpp:MAGAVE_TIME", NB_MAGAVE_TIME},
        {"samdemod.cpp:DC_ALPHA", SAM_DC_ALPHA}, {"samdemod.cpp:PLL_BW", SAM_PLL_BW}, {"samdemod.cpp:PLL_ZETA", SAM_PLL_ZETA}, {"samdemod.cpp:PLL_LIMIT", SAM_PLL_LIMIT},
        {"smeter.cpp:ATTACK_TIMECONST", SM_ATTACK_TIMECONST}, {"smeter.cpp:DECAY_TIMECONST", SM_DECAY_TIMECONST},
        {"smeter.cpp:SMETER_CALIBRATION", SM_CALIBRATION}, {"smeter.cpp:MAX_PWR", SM_MAX_PWR},
        {"demodulator.h:MAX_INBUFSIZE", DEMOD_BUF}, {"datatypes.h:K_2PI", TWO_PI}, {"datatypes.h:K_PI", ONE_PI},
    };
    const int n = (int)(sizeof(tab) / sizeof(tab[0]));
    int i;
    for (i = 0; i < n && i < cap; i++) { names[i] = tab[i].name; values[i] = tab[i].value; }
    return n;
}

/* ==================================================================================== */
/* FFT: plain iterative radix-2, same transform as the reference's Ooura cdft           */
/* (dsp/fft.cpp:416-426: isgn=+1 forward has the POSITIVE exponent, no scaling).        */
/* ==================================================================================== */
/* cos/sin(2 pi k / n), k < n/2, built once per size (threads may race to build: the loser frees its
 * copy).  A stage of length len uses every (n/len)-th entry: k/len and (k n/len)/n are the same
 * double, so the values are exactly those of a per-stage cos/sin call. */
static double *fft_twiddles(int n)
{
    static double *cache[32];
    int lg = 0, k;
    double *t;
    while ((1 << lg) < n) lg++;
    t = __atomic_load_n(&cache[lg], __ATOMIC_ACQUIRE);
    if (t) return t;
    t = (double *)malloc(sizeof(double) * (size_t)n);
    for (k = 0; k < n / 2; k++) {
        const double ang = TWO_PI * (double)k / (double)n;
        t[2 * k] = cos(ang); t[2 * k + 1] = sin(ang);
    }
    {
        double *expect = NULL;
        if (!__atomic_compare_exchange_n(&cache[lg], &expect, t, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) { free(t); t = expect; }
    }
    return t;
}

void orc_fft(int n, int sign, orc_cpx *a)
{
    int i, j, len;
    const double *tw = n > 1 ? fft_twiddles(n) : NULL;
    const double sg = (double)sign;
    for (i = 1, j = 0; i < n; i++) {            /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { orc_cpx t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        int k, b;
        for (b = 0; b < n; b += len) {
            for (k = 0; k < half; k++) {
                const double wr = tw[2 * k * step], wi = sg * tw[2 * k * step + 1];
                orc_cpx u = a[b + k], v = a[b + k + half], t;
                t.re = v.re * wr - v.im * wi;
                t.im = v.re * wi + v.im * wr;
                a[b + k].re = u.re + t.re;          a[b + k].im = u.im + t.im;
                a[b + k + half].re = u.re - t.re;   a[b + k + half].im = u.im - t.im;
            }
        }
    }
}

/* ==================================================================================== */
/* CFft: display spectrum + the convolution FFT with its display side effect             */
/* ==================================================================================== */
struct orc_cfft {
    int overload, invert, ave_count, total_count, size, last_size, ave_size;
    int start_hz, stop_hz, bin_min, bin_max, plot_w;
    double kc, kb, db_comp, fs;
    int *xlat;
    double *win, *pwr_ave, *ave, *sum;
    orc_cpx *work;
};

static void cfft_release(orc_cfft *f)
{
    free(f->xlat); free(f->win); free(f->pwr_ave); free(f->ave); free(f->sum); free(f->work);
    f->xlat = NULL; f->win = f->pwr_ave = f->ave = f->sum = NULL; f->work = NULL;
}

/* dsp/fft.cpp:248-259 */
void orc_cfft_reset(orc_cfft *f)
{
    int i;
    for (i = 0; i < f->size; i++) { f->ave[i] = 0.0; f->sum[i] = 0.0; }
    f->ave_count = 0;
    f->total_count = 0;
}

/* dsp/fft.cpp:103-113 */
void orc_cfft_set_ave(orc_cfft *f, int ave)
{
    if (f->ave_size != ave) f->ave_size = ave > 0 ? ave : 1;
    orc_cfft_reset(f);
}

/* dsp/fft.cpp:118-243: clamp size to [512,65536]; on a size or dBcomp change rebuild the
 * Hann*2 window (:196-198) and K_B/K_C (:186-188); always reset the averages. */
void orc_cfft_set_params(orc_cfft *f, int size, int invert, double db_comp, double fs)
{
    int i;
    if (size == 0) return;
    f->bin_min = f->bin_max = 0;
    f->start_hz = f->stop_hz = 0;
    f->plot_w = 0;
    f->invert = invert;
    f->fs = fs;
    if (f->db_comp != db_comp) { f->last_size = 0; f->db_comp = db_comp; }
    if (size < FFT_MIN_SIZE) f->size = FFT_MIN_SIZE;
    else if (size > FFT_MAX_SIZE) f->size = FFT_MAX_SIZE;
    else f->size = size;
    if (f->last_size != f->size) {
        int n = f->size;
        f->last_size = n;
        cfft_release(f);
        f->win = (double *)zalloc(sizeof(double) * n);
        f->pwr_ave = (double *)zalloc(sizeof(double) * n);
        f->ave = (double *)zalloc(sizeof(double) * n);
        f->sum = (double *)zalloc(sizeof(double) * n);
        f->work = (orc_cpx *)zalloc(sizeof(orc_cpx) * n);
        f->xlat = (int *)zalloc(sizeof(int) * n);
        f->kb = f->db_comp - 20 * log10((double)n * FFT_K_AMPMAX / 2.0);
        f->kc = pow(10.0, (FFT_K_MINDB - f->kb) / 10.0);
        f->kb = f->kb / 10.0;
        for (i = 0; i < n; i++)
            f->win[i] = 2.0 * (.5 - .5 * cos((TWO_PI * i) / (n - 1)));
    }
    orc_cfft_reset(f);
}

orc_cfft *orc_cfft_new(void)
{   /* ctor, dsp/fft.cpp:41-62: 2048 points, Fs 1000, ave 1 */
    orc_cfft *f = (orc_cfft *)zalloc(sizeof(*f));
    f->ave_size = 1;
    f->size = 1024;
    f->db_comp = 0.0;
    orc_cfft_set_params(f, 2048, 0, 0.0, 1000);
    orc_cfft_set_ave(f, 1);
    return f;
}
void orc_cfft_free(orc_cfft *f) { if (f) { cfft_release(f); free(f); } }
int orc_cfft_size(const orc_cfft *f) { return f->size; }
const double *orc_cfft_avebuf(const orc_cfft *f) { return f->ave; }

/* The tail of CFft::CpxFFT (dsp/fft.cpp:515-517, 562-589): every forward transform bumps
 * the counters and folds |X|^2 into the running power average, fft-shifted
 * (natural bin k -> display index (k+N/2) mod N), then stores log10(mean+K_C)+K_B. */
static void cfft_fold_power(orc_cfft *f, const orc_cpx *x)
{
    int n = f->size, k;
    for (k = 0; k < n; k++) {
        int j = (k + n / 2) % n;
        double p = x[k].re * x[k].re + x[k].im * x[k].im;
        if (f->total_count <= f->ave_size) f->sum[j] = f->sum[j] + p;
        else                               f->sum[j] = f->sum[j] - f->pwr_ave[j] + p;
        f->pwr_ave[j] = f->sum[j] / (double)f->ave_count;
        f->ave[j] = log10(f->pwr_ave[j] + f->kc) + f->kb;
    }
}
static void cfft_count(orc_cfft *f)
{
    f->total_count++;
    if (f->ave_count < f->ave_size) f->ave_count++;
}

void orc_cfft_fwd(orc_cfft *f, orc_cpx *a)
{   /* dsp/fft.cpp:416-420 -> CpxFFT incl. side effect */
    cfft_count(f);
    orc_fft(f->size, +1, a);
    cfft_fold_power(f, a);
}
void orc_cfft_rev(orc_cfft *f, orc_cpx *a)
{   /* dsp/fft.cpp:422-426: cftbsub has no side effect */
    orc_fft(f->size, -1, a);
}

/* dsp/fft.cpp:267-288: window, swap I/Q, transform, fold.  Returns m_TotalCount. */
int orc_cfft_put_display(orc_cfft *f, int n, const orc_cpx *in)
{
    int i;
    f->overload = 0;
    for (i = 0; i < n; i++) {
        if (in[i].re > FFT_OVER_LIMIT) f->overload = 1;
        f->work[i].im = f->win[i] * in[i].re;
        f->work[i].re = f->win[i] * in[i].im;
    }
    cfft_count(f);
    orc_fft(f->size, +1, f->work);
    cfft_fold_power(f, f->work);
    return f->total_count;
}

/* dsp/fft.cpp:308-410 */
int orc_cfft_get_screen(orc_cfft *f, int max_h, int max_w, double max_db, double min_db,
                        int start_hz, int stop_hz, int *out)
{
    int i, x, y, ymax = 10000, xprev = -1, n = f->size;
    double off = max_db / 10.0, gain = -10.0 / (max_db - min_db);
    if (f->start_hz != start_hz || f->stop_hz != stop_hz || f->plot_w != max_w) {
        int maxbin = n - 1;
        f->start_hz = start_hz; f->stop_hz = stop_hz; f->plot_w = max_w;
        f->bin_min = (int)((double)start_hz * (double)n / f->fs) + n / 2;
        f->bin_max = (int)((double)stop_hz * (double)n / f->fs) + n / 2;
        if (f->bin_min < 0) f->bin_min = 0;
        if (f->bin_min >= maxbin) f->bin_min = maxbin;
        if (f->bin_max < 0) f->bin_max = 0;
        if (f->bin_max >= maxbin) f->bin_max = maxbin;
        if ((f->bin_max - f->bin_min) > f->plot_w) {
            for (i = f->bin_min; i <= f->bin_max; i++)
                f->xlat[i] = ((i - f->bin_min) * f->plot_w) / (f->bin_max - f->bin_min);
        } else {
            for (i = 0; i < f->plot_w && i < n; i++)   /* table has n entries (:165) */
                f->xlat[i] = f->bin_min + (i * (f->bin_max - f->bin_min)) / f->plot_w;
        }
    }
#define SCREEN_Y(bin) do { \
        int b_ = f->invert ? (n - (bin)) : (bin); \
        if (b_ >= n) b_ = n - 1;   /* reference reads one past the end for invert&&bin==0 */ \
        y = (int)((double)max_h * gain * (f->ave[b_] - off)); \
        if (y < 0) y = 0; \
        if (y > max_h) y = max_h; } while (0)
    if ((f->bin_max - f->bin_min) > f->plot_w) {
        for (i = f->bin_min; i <= f->bin_max; i++) {
            SCREEN_Y(i);
            x = f->xlat[i];
            if (x == xprev) {
                if (y < ymax) { out[x] = y; ymax = y; }
            } else {
                out[x] = y; xprev = x; ymax = y;
            }
        }
    } else {
        for (x = 0; x < f->plot_w; x++) {
            i = f->xlat[x];
            SCREEN_Y(i);
            out[x] = y;
        }
    }
#undef SCREEN_Y
    return f->overload;
}

/* CPlotter's palette (gui/plotter.cpp:67-83), QColor::setRgb(r, g, b) as 0xFFRRGGBB */
void orc_plotter_color_table(unsigned int *out256)
{
    int i;
    for (i = 0; i < 256; i++) {
        int r = 0, g = 0, b = 0;
        if (i < 43) { r = 0; g = 0; b = 255 * (i) / 43; }
        if ((i >= 43) && (i < 87)) { r = 0; g = 255 * (i - 43) / 43; b = 255; }
        if ((i >= 87) && (i < 120)) { r = 0; g = 255; b = 255 - (255 * (i - 87) / 32); }
        if ((i >= 120) && (i < 154)) { r = (255 * (i - 120) / 33); g = 255; b = 0; }
        if ((i >= 154) && (i < 217)) { r = 255; g = 255 - (255 * (i - 154) / 62); b = 0; }
        if (i >= 217) { r = 255; g = 0; b = 128 * (i - 217) / 38; }
        out256[i] = 0xff000000u | ((unsigned)r << 16) | ((unsigned)g << 8) | (unsigned)b;
    }
}
/* the new top line of the waterfall (CPlotter::draw, gui/plotter.cpp:425-441): levels at MaxHeight 255, pixel i
 * painted m_ColorTbl[255 - fftbuf[i]].  levels: max_w + 1 ints of work space (the reference's stack buffer), pre-set
 * by the caller; rgb[i] is written for i < max_w.  Returns the overload flag. */
int orc_plotter_waterfall_line(orc_cfft *f, int max_w, double max_db, double min_db, int start_hz, int stop_hz,
                               int *levels, unsigned int *rgb)
{
    unsigned int tbl[256];
    int i, ov = orc_cfft_get_screen(f, 255, max_w, max_db, min_db, start_hz, stop_hz, levels);
    orc_plotter_color_table(tbl);
    for (i = 0; i < max_w; i++) {
        int y = levels[i];
        if (y >= 0 && y <= 255) rgb[i] = tbl[255 - y];
    }
    return ov;
}

/* ==================================================================================== */
/* CFastFIR                                                                              */
/* ==================================================================================== */
struct orc_fastfir {
    int n, p;                 /* FFT size, taps = n/2+1 */
    int pos, faithful;
    double flo, fhi, off, fs;
    double *win;
    orc_cpx *coef, *buf, *ovl;
    orc_cfft *side;           /* only used when faithful: the shared CFft's display state */
};

orc_fastfir *orc_fastfir_new(int fft_size)
{   /* ctor dsp/fastfir.cpp:67-130, with CONV_FFT_SIZE -> n, CONV_FIR_SIZE -> n/2+1 */
    orc_fastfir *f = (orc_fastfir *)zalloc(sizeof(*f));
    int i;
    f->n = fft_size; f->p = fft_size / 2 + 1;
    f->win = (double *)zalloc(sizeof(double) * f->p);
    f->coef = (orc_cpx *)zalloc(sizeof(orc_cpx) * f->n);
    f->buf = (orc_cpx *)zalloc(sizeof(orc_cpx) * f->n);
    f->ovl = (orc_cpx *)zalloc(sizeof(orc_cpx) * f->p);
    f->pos = f->p - 1;
    for (i = 0; i < f->p; i++)      /* Blackman-Nuttall, :93-101 */
        f->win[i] = FF_WIN_A0
                  - FF_WIN_A1 * cos((TWO_PI * i) / (f->p - 1))
                  + FF_WIN_A2 * cos((2.0 * TWO_PI * i) / (f->p - 1))
                  - FF_WIN_A3 * cos((3.0 * TWO_PI * i) / (f->p - 1));
    f->flo = -1.0; f->fhi = 1.0; f->off = 1.0; f->fs = 1.0;
    f->side = orc_cfft_new();
    orc_cfft_set_params(f->side, f->n, 0, 0.0, 1.0);
    return f;
}
void orc_fastfir_free(orc_fastfir *f)
{
    if (!f) return;
    free(f->win); free(f->coef); free(f->buf); free(f->ovl); orc_cfft_free(f->side); free(f);
}
void orc_fastfir_set_faithful(orc_fastfir *f, int on) { f->faithful = on; }
const orc_cpx *orc_fastfir_coef(const orc_fastfir *f) { return f->coef; }

static void fastfir_fwd(orc_fastfir *f, orc_cpx *a)
{
    if (f->faithful) orc_cfft_fwd(f->side, a);
    else orc_fft(f->n, +1, a);
}

/* dsp/fastfir.cpp:178-259.  Returns 1 if new coefficients were designed, 0 if unchanged,
 * -1 if the parameters were rejected (reference: debug print, old taps kept). */
int orc_fastfir_setup(orc_fastfir *f, double flo, double fhi, double offset, double fs)
{
    int i;
    double nfl, nfh, nfc, nfs, centre;
    if (flo == f->flo && fhi == f->fhi && offset == f->off && fs == f->fs) return 0;
    f->flo = flo; f->fhi = fhi; f->off = offset; f->fs = fs;
    flo += offset; fhi += offset;
    if (flo >= fhi || flo >= fs / 2.0 || flo <= -fs / 2.0 || fhi >= fs / 2.0 || fhi <= -fs / 2.0)
        return -1;
    nfl = flo / fs; nfh = fhi / fs;
    nfc = (nfh - nfl) / 2.0;
    nfs = TWO_PI * (nfh + nfl) / 2.0;
    centre = 0.5 * (double)(f->p - 1);
    memset(f->coef, 0, sizeof(orc_cpx) * f->n);
    for (i = 0; i < f->p; i++) {
        double x = (double)i - centre, z;
        if ((double)i == centre) z = 2.0 * nfc;
        else z = sin(TWO_PI * x * nfc) / (ONE_PI * x) * f->win[i];
        f->coef[i].re = z * cos(nfs * x) / (double)f->n;
        f->coef[i].im = z * sin(nfs * x) / (double)f->n;
    }
    fastfir_fwd(f, f->coef);
    return 1;
}

/* dsp/fastfir.cpp:268-321: overlap-save; block = [P-1 saved | L new], emit P-1..N-1 */
int orc_fastfir_process(orc_fastfir *f, int n, const orc_cpx *in, orc_cpx *out)
{
    int i, j, produced = 0, hop_start = f->n - f->p + 1;
    for (i = 0; i < n; i++) {
        j = f->pos - hop_start;
        if (j >= 0) f->ovl[j] = in[i];
        f->buf[f->pos++] = in[i];
        if (f->pos >= f->n) {
            fastfir_fwd(f, f->buf);
            for (j = 0; j < f->n; j++) {
                double sr = f->buf[j].re, si = f->buf[j].im;
                f->buf[j].re = f->coef[j].re * sr - f->coef[j].im * si;
                f->buf[j].im = f->coef[j].re * si + f->coef[j].im * sr;
            }
            orc_fft(f->n, -1, f->buf);
            for (j = f->p - 1; j < f->n; j++) out[produced++] = f->buf[j];
            for (j = 0; j < f->p - 1; j++) f->buf[j] = f->ovl[j];
            f->pos = f->p - 1;
        }
    }
    return produced;
}

/* ==================================================================================== */
/* CDownConvert                                                                          */
/* ==================================================================================== */

typedef struct {
    int kind;                 /* 3 = CIC3, 11 = unrolled HB11, else generic HB length */
    double h[CSDR_HB_MAX_LEN];
    orc_cpx hist[CSDR_HB_MAX_LEN];    /* HB11: d0..d9; generic: unused (scratch keeps it) */
    orc_cpx xodd, xeven;      /* CIC3 state */
    orc_cpx *scratch;         /* generic HB work buffer */
} dc_stage;

struct orc_downconv {
    double out_rate, nco_freq, cw_off, nco_inc, in_rate, max_bw;
    double osc_cos, osc_sin;
    orc_cpx osc1;
    int nstages;
    dc_stage st[DC_MAX_STAGES + 1];
};

orc_downconv *orc_downconv_new(void)
{   /* ctor dsp/downconvert.cpp:60-73 */
    orc_downconv *d = (orc_downconv *)zalloc(sizeof(*d));
    d->in_rate = 100000.0; d->max_bw = 10000.0;
    d->osc1.re = 1.0; d->osc1.im = 0.0;
    d->osc_cos = 1.0; d->osc_sin = 0.0;     /* reference leaves these unset until SetFrequency */
    return d;
}
static void dc_drop_stages(orc_downconv *d)
{
    int i;
    for (i = 0; i < d->nstages; i++) free(d->st[i].scratch);
    memset(d->st, 0, sizeof(d->st));
    d->nstages = 0;
}
void orc_downconv_free(orc_downconv *d) { if (d) { dc_drop_stages(d); free(d); } }
void orc_downconv_set_cw_offset(orc_downconv *d, double off) { d->cw_off = off; }
double orc_downconv_nco_freq(const orc_downconv *d) { return d->nco_freq; }

/* dsp/downconvert.cpp:98-107.  NB: the CW offset is folded into the stored frequency,
 * so a later SetDataRate() (which re-calls this with the stored value) adds it again. */
void orc_downconv_set_frequency(orc_downconv *d, double f)
{
    d->nco_freq = f + d->cw_off;
    d->nco_inc = TWO_PI * d->nco_freq / d->in_rate;
    d->osc_cos = cos(d->nco_inc);
    d->osc_sin = sin(d->nco_inc);
}

/* dsp/downconvert.cpp:114-173 */
double orc_downconv_set_data_rate(orc_downconv *d, double in_rate, double max_bw)
{
    double f = in_rate;
    if (d->in_rate != in_rate || d->max_bw != max_bw) {
        d->in_rate = in_rate; d->max_bw = max_bw;
        dc_drop_stages(d);
        while (f > (max_bw / csdr_hb_maxbw[CSDR_HB_NUM_FILTERS - 1]) && f > DCV_MIN_OUTPUT_RATE) {
            dc_stage *s;
            int k;
            if (d->nstages >= DC_MAX_STAGES) break;   /* reference: unchecked (downconvert.h:18) */
            s = &d->st[d->nstages];
            if (f >= (max_bw / CSDR_CIC3_MAXBW)) {
                s->kind = 3;
                d->nstages++;
            } else {
                for (k = 0; k < CSDR_HB_NUM_FILTERS; k++) {
                    if (f >= (max_bw / csdr_hb_maxbw[k])) {
                        s->kind = csdr_hb_expand(k, s->h);
                        if (s->kind != 11)
                            s->scratch = (orc_cpx *)zalloc(sizeof(orc_cpx) * DC_HB_SCRATCH);
                        d->nstages++;
                        break;
                    }
                }
            }
            f /= 2.0;
        }
        d->out_rate = f;
        orc_downconv_set_frequency(d, d->nco_freq);
    }
    return d->out_rate;
}
int orc_downconv_stages(const orc_downconv *d, int *codes)
{
    int i;
    for (i = 0; i < d->nstages; i++) codes[i] = d->st[i].kind;
    return d->nstages;
}

/* CIC3: dsp/downconvert.cpp:444-460 */
static int dec_cic3(dc_stage *s, int n, orc_cpx *x)
{
    int i, j;
    for (i = 0, j = 0; i < n; i += 2, j++) {
        orc_cpx ev = x[i], od = x[i + 1];
        x[j].re = .125 * (od.re + s->xeven.re + 3.0 * (s->xodd.re + ev.re));
        x[j].im = .125 * (od.im + s->xeven.im + 3.0 * (s->xodd.im + ev.im));
        s->xodd = od; s->xeven = ev;
    }
    return j;
}

/* generic half band: dsp/downconvert.cpp:286-320.  In place like the reference call
 * (in==out): outputs overwrite the front of x before the history tail is saved. */
static int dec_hb(dc_stage *s, int n, orc_cpx *x)
{
    int L = s->kind, c = (L - 1) / 2, i, j, nout = 0;
    orc_cpx *w = s->scratch;
    if (n < L) return n / 2;                 /* :291-292, nothing written */
    if (n + L - 1 > DC_HB_SCRATCH) return -1;  /* reference would overrun its scratch buffer */
    memcpy(w + (L - 1), x, sizeof(orc_cpx) * n);
    for (i = 0; i < n; i += 2) {
        double ar = w[i].re * s->h[0], ai = w[i].im * s->h[0];
        for (j = 2; j < L; j += 2) { ar += w[i + j].re * s->h[j]; ai += w[i + j].im * s->h[j]; }
        ar += w[i + c].re * s->h[c];
        ai += w[i + c].im * s->h[c];
        x[nout].re = ar; x[nout].im = ai; nout++;
    }
    for (i = 0, j = n - L + 1; i < L - 1; i++) w[i] = x[j++];
    return nout;
}

/* unrolled 11-tap half band: dsp/downconvert.cpp:348-423.  Same arithmetic as the generic
 * form but the products are summed in tap order 0,2,4,5,6,8,10 and the history is the
 * ten samples d0..d9 = x[n-10..n-1] read back from the (in-place) buffer. */
static int dec_hb11(dc_stage *s, int n, orc_cpx *x)
{
    static const int tap[7] = { 0, 2, 4, 5, 6, 8, 10 };
    orc_cpx first[9];
    int j, k, nout = n / 2;
    for (j = 0; j < 9 && j < nout; j++) {      /* outputs that still reach into history */
        double ar = 0, ai = 0;
        for (k = 0; k < 7; k++) {
            int idx = 2 * j + tap[k] - 10;     /* <0: history d[idx+10] */
            orc_cpx v = idx < 0 ? s->hist[idx + 10] : x[idx];
            if (k == 0) { ar = s->h[tap[k]] * v.re; ai = s->h[tap[k]] * v.im; }
            else { ar += s->h[tap[k]] * v.re; ai += s->h[tap[k]] * v.im; }
        }
        first[j].re = ar; first[j].im = ai;
    }
    for (j = 9; j < 9 + (n - 11 - 6) / 2; j++) {
        const orc_cpx *p = &x[2 * j - 10];
        double ar = s->h[0] * p[0].re, ai = s->h[0] * p[0].im;
        for (k = 1; k < 7; k++) { ar += s->h[tap[k]] * p[tap[k]].re; ai += s->h[tap[k]] * p[tap[k]].im; }
        x[j].re = ar; x[j].im = ai;
    }
    for (j = 0; j < 9 && j < nout; j++) x[j] = first[j];
    for (k = 0; k < 10; k++) s->hist[k] = x[n - 10 + k];
    return nout;
}

/* dsp/downconvert.cpp:186-263 */
int orc_downconv_process(orc_downconv *d, int n, orc_cpx *in, orc_cpx *out)
{
    int i, j, m = n;
    for (i = 0; i < n; i++) {
        orc_cpx s = in[i], osc;
        double g;
        osc.re = d->osc1.re * d->osc_cos - d->osc1.im * d->osc_sin;
        osc.im = d->osc1.im * d->osc_cos + d->osc1.re * d->osc_sin;
        g = 1.95 - (d->osc1.re * d->osc1.re + d->osc1.im * d->osc1.im);
        d->osc1.re = g * osc.re;
        d->osc1.im = g * osc.im;
        in[i].re = (s.re * osc.re) - (s.im * osc.im);
        in[i].im = (s.re * osc.im) + (s.im * osc.re);
    }
    for (j = 0; j < d->nstages; j++) {
        dc_stage *s = &d->st[j];
        if (s->kind == 3) m = dec_cic3(s, m, in);
        else if (s->kind == 11) m = dec_hb11(s, m, in);
        else m = dec_hb(s, m, in);
    }
    for (i = 0; i < m; i++) out[i] = in[i];
    return m;
}

/* ==================================================================================== */
/* CFir                                                                                   */
/* ==================================================================================== */
struct orc_fir {
    double fs;
    int ntaps, state;
    double coef[2 * FIR_MAX], icoef[2 * FIR_MAX], qcoef[2 * FIR_MAX];
    double rz[FIR_MAX];
    orc_cpx cz[FIR_MAX];
};
orc_fir *orc_fir_new(void)
{
    orc_fir *f = (orc_fir *)zalloc(sizeof(*f));
    f->ntaps = 1; f->state = 0;            /* dsp/fir.cpp:56-60 */
    return f;
}
void orc_fir_free(orc_fir *f) { free(f); }

static void fir_clear(orc_fir *f)
{
    int i;
    for (i = 0; i < f->ntaps; i++) { f->rz[i] = 0.0; f->cz[i].re = 0.0; f->cz[i].im = 0.0; }
    f->state = 0;
}
static void fir_dup(orc_fir *f)
{   /* doubled coefficient arrays + I/Q copies, dsp/fir.cpp:218-227 */
    int i;
    for (i = 0; i < f->ntaps; i++) f->coef[i + f->ntaps] = f->coef[i];
    for (i = 0; i < 2 * f->ntaps; i++) { f->icoef[i] = f->coef[i]; f->qcoef[i] = f->coef[i]; }
}

/* dsp/fir.cpp:133-153 */
void orc_fir_init_const(orc_fir *f, int ntaps, const double *coef)
{
    int i;
    f->ntaps = ntaps > FIR_MAX ? FIR_MAX : ntaps;
    for (i = 0; i < f->ntaps; i++) { f->coef[i] = coef[i]; f->coef[f->ntaps + i] = coef[i]; }
    fir_clear(f);
}

/* I0 series, dsp/fir.cpp:414-432 */
static double bessel_i0(double x)
{
    double x2 = x / 2.0, sum = 1.0, ds = 1.0, di = 1.0, t;
    do { t = x2 / di; t *= t; ds *= t; sum += ds; di += 1.0; } while (ds >= 1e-9 * sum);
    return sum;
}
static double kaiser_beta(double astop)
{   /* dsp/fir.cpp:184-190 */
    if (astop < 20.96) return 0;
    if (astop >= 50.0) return .1102 * (astop - 8.71);
    return .5842 * pow((astop - 20.96), 0.4) + .07886 * (astop - 20.96);
}

/* dsp/fir.cpp:173-261 */
int orc_fir_init_lp(orc_fir *f, double scale, double astop, double fpass, double fstop, double fs)
{
    int n;
    double npass = fpass / fs, nstop = fstop / fs, ncut = (nstop + npass) / 2.0;
    double beta = kaiser_beta(astop), centre, izb;
    f->fs = fs;
    f->ntaps = (int)((astop - 8.0) / (2.285 * TWO_PI * (nstop - npass)) + 1);
    if (f->ntaps > FIR_MAX) f->ntaps = FIR_MAX;
    if (f->ntaps < 3) f->ntaps = 3;
    centre = .5 * (double)(f->ntaps - 1);
    izb = bessel_i0(beta);
    for (n = 0; n < f->ntaps; n++) {
        double x = (double)n - centre, c;
        if ((double)n == centre) c = 2.0 * ncut;
        else c = sin(TWO_PI * x * ncut) / (ONE_PI * x);
        x = ((double)n - ((double)f->ntaps - 1.0) / 2.0) / (((double)f->ntaps - 1.0) / 2.0);
        f->coef[n] = scale * c * bessel_i0(beta * sqrt(1 - (x * x))) / izb;
    }
    fir_dup(f);
    fir_clear(f);
    return f->ntaps;
}

/* dsp/fir.cpp:278-367 */
int orc_fir_init_hp(orc_fir *f, double scale, double astop, double fpass, double fstop, double fs)
{
    int n;
    double npass = fpass / fs, nstop = fstop / fs, ncut = (nstop + npass) / 2.0;
    double beta = kaiser_beta(astop), centre, izb;
    f->fs = fs;
    f->ntaps = (int)((astop - 8.0) / (2.285 * TWO_PI * (npass - nstop)) + 1);
    if (f->ntaps > (FIR_MAX - 1)) f->ntaps = FIR_MAX - 1;
    if (f->ntaps < 3) f->ntaps = 3;
    f->ntaps |= 1;
    izb = bessel_i0(beta);
    centre = .5 * (double)(f->ntaps - 1);
    for (n = 0; n < f->ntaps; n++) {
        double x = (double)n - (double)(f->ntaps - 1) / 2.0, c;
        if ((double)n == centre) c = 1.0 - 2.0 * ncut;
        else c = sin(ONE_PI * x) / (ONE_PI * x) - sin(TWO_PI * x * ncut) / (ONE_PI * x);
        x = ((double)n - ((double)f->ntaps - 1.0) / 2.0) / (((double)f->ntaps - 1.0) / 2.0);
        f->coef[n] = scale * c * bessel_i0(beta * sqrt(1 - (x * x))) / izb;
    }
    fir_dup(f);
    fir_clear(f);
    return f->ntaps;
}

/* dsp/fir.cpp:374-407: Hilbert band-pass pair, state untouched */
void orc_fir_gen_hilbert(orc_fir *f, double freq_offset)
{
    int n;
    for (n = 0; n < f->ntaps; n++) {
        double a = (TWO_PI * freq_offset / f->fs) * ((double)n - ((double)(f->ntaps - 1) / 2.0));
        f->icoef[n] = 2.0 * f->coef[n] * cos(a);
        f->qcoef[n] = 2.0 * f->coef[n] * sin(a);
    }
    for (n = 0; n < f->ntaps; n++) {
        f->icoef[n + f->ntaps] = f->icoef[n];
        f->qcoef[n + f->ntaps] = f->qcoef[n];
    }
}
int orc_fir_taps(const orc_fir *f, double *coef, double *icoef, double *qcoef)
{
    int i;
    for (i = 0; i < f->ntaps; i++) {
        if (coef) coef[i] = f->coef[i];
        if (icoef) icoef[i] = f->icoef[i];
        if (qcoef) qcoef[i] = f->qcoef[i];
    }
    return f->ntaps;
}

/* dsp/fir.cpp:72-92: ring with the newest sample at z[state]; taps walk from
 * coef[ntaps-state] so that y[n] = sum_k h[k] x[n-k], summed in ring order. */
void orc_fir_process_real(orc_fir *f, int n, const double *in, double *out)
{
    int i, j;
    for (i = 0; i < n; i++) {
        const double *h = &f->coef[f->ntaps - f->state];
        double acc;
        f->rz[f->state] = in[i];
        acc = h[0] * f->rz[0];
        for (j = 1; j < f->ntaps; j++) acc += h[j] * f->rz[j];
        if (--f->state < 0) f->state += f->ntaps;
        out[i] = acc;
    }
}
/* dsp/fir.cpp:101-127: I taps on .re, Q taps on .im, no cross terms */
void orc_fir_process_cpx(orc_fir *f, int n, const orc_cpx *in, orc_cpx *out)
{
    int i, j;
    for (i = 0; i < n; i++) {
        const double *hi = &f->icoef[f->ntaps - f->state];
        const double *hq = &f->qcoef[f->ntaps - f->state];
        double ar, ai;
        f->cz[f->state] = in[i];
        ar = hi[0] * f->cz[0].re; ai = hq[0] * f->cz[0].im;
        for (j = 1; j < f->ntaps; j++) { ar += hi[j] * f->cz[j].re; ai += hq[j] * f->cz[j].im; }
        if (--f->state < 0) f->state += f->ntaps;
        out[i].re = ar; out[i].im = ai;
    }
}

/* ==================================================================================== */
/* CIir: RBJ biquads, direct form II (dsp/iir.cpp:86-201)                                */
/* ==================================================================================== */
struct orc_iir { double a1, a2, b0, b1, b2, w1a, w2a, w1b, w2b; };
void orc_iir_init(orc_iir *f, int kind, double f0, double q, double fs)
{
    double w0 = TWO_PI * f0 / fs, alpha = sin(w0) / (2.0 * q), A = 1.0 / (1.0 + alpha);
    switch (kind) {
    case 0: f->b0 = A * ((1.0 - cos(w0)) / 2.0); f->b1 = A * (1.0 - cos(w0)); f->b2 = A * ((1.0 - cos(w0)) / 2.0); break;
    case 1: f->b0 = A * ((1.0 + cos(w0)) / 2.0); f->b1 = -A * (1.0 + cos(w0)); f->b2 = A * ((1.0 + cos(w0)) / 2.0); break;
    case 2: f->b0 = A * alpha; f->b1 = 0.0; f->b2 = A * -alpha; break;
    default: f->b0 = A * 1.0; f->b1 = A * (-2.0 * cos(w0)); f->b2 = A * 1.0; break;
    }
    f->a1 = A * (-2.0 * cos(w0));
    f->a2 = A * (1.0 - alpha);
    f->w1a = f->w2a = f->w1b = f->w2b = 0.0;
}
orc_iir *orc_iir_new(void)
{
    orc_iir *f = (orc_iir *)zalloc(sizeof(*f));
    orc_iir_init(f, 3, 25000, 1000.0, 100000);      /* ctor: dsp/iir.cpp:77-80 */
    return f;
}
void orc_iir_free(orc_iir *f) { free(f); }
void orc_iir_coefs(const orc_iir *f, double *c)
{ c[0] = f->b0; c[1] = f->b1; c[2] = f->b2; c[3] = f->a1; c[4] = f->a2; }
void orc_iir_process_real(orc_iir *f, int n, const double *in, double *out)
{
    int i;
    for (i = 0; i < n; i++) {
        double w0 = in[i] - f->a1 * f->w1a - f->a2 * f->w2a;
        out[i] = f->b0 * w0 + f->b1 * f->w1a + f->b2 * f->w2a;
        f->w2a = f->w1a; f->w1a = w0;
    }
}
void orc_iir_process_cpx(orc_iir *f, int n, const orc_cpx *in, orc_cpx *out)
{
    int i;
    for (i = 0; i < n; i++) {
        double wa = in[i].re - f->a1 * f->w1a - f->a2 * f->w2a, wb;
        out[i].re = f->b0 * wa + f->b1 * f->w1a + f->b2 * f->w2a;
        f->w2a = f->w1a; f->w1a = wa;
        wb = in[i].im - f->a1 * f->w1b - f->a2 * f->w2b;
        out[i].im = f->b0 * wb + f->b1 * f->w1b + f->b2 * f->w2b;
        f->w2b = f->w1b; f->w1b = wb;
    }
}

/* ==================================================================================== */
/* CAgc (dsp/agc.cpp)                                                                    */
/* ==================================================================================== */
struct orc_agc {
    int on, hang, thresh, manual, decay;
    double slope_factor;               /* TYPEREAL compared against int, agc.cpp:109 */
    double fs, manual_gain, decay_ave, attack_ave;
    double att_rise, att_fall, dec_rise, dec_fall;
    double fixed_gain, knee, gain_slope, peak;
    int dly_pos, mag_pos, dly_n, win_n, hang_time, hang_timer;
    orc_cpx dly[AGC_RING];
    double mag[AGC_RING];
};
orc_agc *orc_agc_new(void)
{   /* ctor dsp/agc.cpp:80-89; ring state is first written by SetParameters */
    orc_agc *a = (orc_agc *)zalloc(sizeof(*a));
    a->on = 1; a->fs = 100.0;
    return a;
}
void orc_agc_free(orc_agc *a) { free(a); }

/* dsp/agc.cpp:104-167 */
void orc_agc_set(orc_agc *a, int on, int hang, int thresh, int manual_gain, int slope,
                 int decay, double fs)
{
    int i;
    if (on == a->on && hang == a->hang && thresh == a->thresh && manual_gain == a->manual &&
        (double)slope == a->slope_factor && decay == a->decay && fs == a->fs)
        return;
    a->on = on; a->hang = hang; a->thresh = thresh; a->manual = manual_gain;
    a->slope_factor = slope; a->decay = decay;
    if (a->fs != fs) {
        a->fs = fs;
        for (i = 0; i < AGC_RING; i++) { a->dly[i].re = 0.0; a->dly[i].im = 0.0; a->mag[i] = -16.0; }
        a->dly_pos = 0; a->hang_timer = 0;
        a->peak = -16.0; a->decay_ave = -5.0; a->attack_ave = -5.0;
        a->mag_pos = 0;
    }
    a->manual_gain = AGC_MAX_MANUAL_AMPLITUDE * pow(10.0, -(100 - (double)a->manual) / 20.0);
    a->knee = (double)a->thresh / 20.0;
    a->gain_slope = a->slope_factor / (100.0);
    a->fixed_gain = AGC_OUTSCALE * pow(10.0, a->knee * (a->gain_slope - 1.0));
    a->att_rise = (1.0 - exp(-1.0 / (a->fs * AGC_ATTACK_RISE_TIMECONST)));
    a->att_fall = (1.0 - exp(-1.0 / (a->fs * AGC_ATTACK_FALL_TIMECONST)));
    a->dec_rise = (1.0 - exp(-1.0 / (a->fs * (double)a->decay * .001 * AGC_DECAY_RISEFALL_RATIO)));
    a->hang_time = (int)(a->fs * (double)a->decay * .001);
    if (a->hang) a->dec_fall = (1.0 - exp(-1.0 / (a->fs * AGC_RELEASE_TIMECONST)));
    else         a->dec_fall = (1.0 - exp(-1.0 / (a->fs * (double)a->decay * .001)));
    a->dly_n = (int)(a->fs * AGC_DELAY_TIMECONST);
    a->win_n = (int)(a->fs * AGC_WINDOW_TIMECONST);
    if (a->dly_n >= AGC_RING - 1) a->dly_n = AGC_RING - 1;
    /* window is NOT clamped in the reference (App. A.6); stay below 113.7 kS/s */
}

/* one step of the log-magnitude tracker shared by both variants (agc.cpp:206-283);
 * returns the gain to apply to the delayed sample */
static double agc_track(orc_agc *a, double mag)
{
    double oldest = a->mag[a->mag_pos];
    int i;
    a->mag[a->mag_pos++] = mag;
    if (a->mag_pos >= a->win_n) a->mag_pos = 0;
    if (mag > a->peak) {
        a->peak = mag;
    } else if (oldest == a->peak) {
        a->peak = -8.0;
        for (i = 0; i < a->win_n; i++) if (a->mag[i] > a->peak) a->peak = a->mag[i];
    }
    if (a->peak > a->attack_ave) a->attack_ave = (1.0 - a->att_rise) * a->attack_ave + a->att_rise * a->peak;
    else                         a->attack_ave = (1.0 - a->att_fall) * a->attack_ave + a->att_fall * a->peak;
    if (a->hang) {
        if (a->peak > a->decay_ave) {
            a->decay_ave = (1.0 - a->dec_rise) * a->decay_ave + a->dec_rise * a->peak;
            a->hang_timer = 0;
        } else if (a->hang_timer < a->hang_time) {
            a->hang_timer++;
        } else {
            a->decay_ave = (1.0 - a->dec_fall) * a->decay_ave + a->dec_fall * a->peak;
        }
    } else {
        if (a->peak > a->decay_ave) a->decay_ave = (1.0 - a->dec_rise) * a->decay_ave + a->dec_rise * a->peak;
        else                        a->decay_ave = (1.0 - a->dec_fall) * a->decay_ave + a->dec_fall * a->peak;
    }
    mag = a->attack_ave > a->decay_ave ? a->attack_ave : a->decay_ave;
    if (mag <= a->knee) return a->fixed_gain;
    return AGC_OUTSCALE * pow(10.0, mag * (a->gain_slope - 1.0));
}

/* dsp/agc.cpp:174-296 */
void orc_agc_process_cpx(orc_agc *a, int n, const orc_cpx *in, orc_cpx *out)
{
    int i;
    if (!a->on) {
        for (i = 0; i < n; i++) { out[i].re = a->manual_gain * in[i].re; out[i].im = a->manual_gain * in[i].im; }
        return;
    }
    for (i = 0; i < n; i++) {
        orc_cpx x = in[i], delayed = a->dly[a->dly_pos];
        double mag = fabs(x.re), mim = fabs(x.im), g;
        a->dly[a->dly_pos++] = x;
        if (a->dly_pos >= a->dly_n) a->dly_pos = 0;
        if (mim > mag) mag = mim;
        mag = log10(mag + AGC_MIN_CONSTANT) - log10(AGC_MAX_AMPLITUDE);
        g = agc_track(a, mag);
        out[i].re = delayed.re * g;
        out[i].im = delayed.im * g;
    }
}
/* dsp/agc.cpp:301-401 (real variant shares the .re lane of the delay ring) */
void orc_agc_process_real(orc_agc *a, int n, const double *in, double *out)
{
    int i;
    if (!a->on) { for (i = 0; i < n; i++) out[i] = a->manual_gain * in[i]; return; }
    for (i = 0; i < n; i++) {
        double x = in[i], delayed = a->dly[a->dly_pos].re, g;
        a->dly[a->dly_pos++].re = x;
        if (a->dly_pos >= a->dly_n) a->dly_pos = 0;
        g = agc_track(a, log10(fabs(x) + AGC_MIN_CONSTANT) - log10(AGC_MAX_AMPLITUDE));
        out[i] = delayed * g;
    }
}

/* ==================================================================================== */
/* CSMeter (dsp/smeter.cpp:49-112)                                                        */
/* ==================================================================================== */
struct orc_smeter { double ave_mag, peak_mag, fs, att_ave, dec_ave, att_a, dec_a; };
orc_smeter *orc_smeter_new(void)
{
    orc_smeter *s = (orc_smeter *)zalloc(sizeof(*s));
    s->peak_mag = 0; s->fs = 1.0; s->att_a = 1.0; s->dec_a = 1.0;
    s->att_ave = -120.0; s->dec_ave = -120.0;
    return s;
}
void orc_smeter_free(orc_smeter *s) { free(s); }
void orc_smeter_process(orc_smeter *s, int n, const orc_cpx *in, double fs)
{
    int i;
    if (fs != s->fs) {
        s->fs = fs;
        s->att_a = (1.0 - exp(-1.0 / (fs * SM_ATTACK_TIMECONST)));
        s->dec_a = (1.0 - exp(-1.0 / (fs * SM_DECAY_TIMECONST)));
    }
    for (i = 0; i < n; i++) {
        double mag = 10.0 * log10((in[i].re * in[i].re + in[i].im * in[i].im) / SM_MAX_PWR + 1e-50);
        s->att_ave = (1.0 - s->att_a) * s->att_ave + s->att_a * mag;
        s->dec_ave = (1.0 - s->dec_a) * s->dec_ave + s->dec_a * mag;
        if (s->att_ave > s->dec_ave) { s->ave_mag = s->att_ave; s->dec_ave = s->att_ave; }
        else s->ave_mag = s->dec_ave;
        if (mag > s->peak_mag) s->peak_mag = mag;
    }
}
double orc_smeter_peak(orc_smeter *s) { double x = s->peak_mag; s->peak_mag = 0; return x + SM_CALIBRATION; }
double orc_smeter_ave(orc_smeter *s) { return s->ave_mag + SM_CALIBRATION; }

/* ==================================================================================== */
/* AM (dsp/amdemod.cpp)                                                                   */
/* ==================================================================================== */
struct orc_amdemod { double fs, z1; orc_fir *fir; };
orc_amdemod *orc_amdemod_new(double fs)
{
    orc_amdemod *d = (orc_amdemod *)zalloc(sizeof(*d));
    d->fs = fs; d->fir = orc_fir_new();
    orc_fir_init_lp(d->fir, 1.0, 50.0, 10000, 10000 * 1.8, fs);
    return d;
}
void orc_amdemod_free(orc_amdemod *d) { if (d) { orc_fir_free(d->fir); free(d); } }
void orc_amdemod_set_bandwidth(orc_amdemod *d, double bw)
{ orc_fir_init_lp(d->fir, 1.0, 50.0, bw, bw * 1.8, d->fs); }
static double am_env(orc_amdemod *d, orc_cpx x)
{   /* envelope then H(z)=(1-z^-1)/(1-.99 z^-1), amdemod.cpp:70-80 */
    double mag = sqrt(x.re * x.re + x.im * x.im);
    double z0 = mag + (d->z1 * AM_DC_ALPHA), y = z0 - d->z1;
    d->z1 = z0;
    return y;
}
int orc_amdemod_process_mono(orc_amdemod *d, int n, const orc_cpx *in, double *out)
{
    int i;
    for (i = 0; i < n; i++) out[i] = am_env(d, in[i]);
    orc_fir_process_real(d->fir, n, out, out);
    return n;
}
int orc_amdemod_process_stereo(orc_amdemod *d, int n, const orc_cpx *in, orc_cpx *out)
{
    int i;
    for (i = 0; i < n; i++) { double y = am_env(d, in[i]); out[i].re = y; out[i].im = y; }
    orc_fir_process_cpx(d->fir, n, out, out);
    return n;
}

/* ==================================================================================== */
/* SAM (dsp/samdemod.cpp)                                                                 */
/* ==================================================================================== */
struct orc_samdemod {
    double fs, z1, y1, phase, freq, lo, hi, alpha, beta;
    orc_fir *fir;
};
orc_samdemod *orc_samdemod_new(double fs)
{   /* ctor :54-73 */
    orc_samdemod *d = (orc_samdemod *)zalloc(sizeof(*d));
    double norm = TWO_PI / fs;
    d->fs = fs;
    d->lo = -SAM_PLL_LIMIT * norm; d->hi = SAM_PLL_LIMIT * norm;
    d->alpha = 2.0 * SAM_PLL_ZETA * SAM_PLL_BW * norm;
    d->beta = (d->alpha * d->alpha) / (4.0 * SAM_PLL_ZETA * SAM_PLL_ZETA);
    d->fir = orc_fir_new();
    orc_fir_init_lp(d->fir, 1.0, 40.0, 4500, 5500, fs);
    orc_fir_gen_hilbert(d->fir, 5000.0);
    return d;
}
void orc_samdemod_free(orc_samdemod *d) { if (d) { orc_fir_free(d->fir); free(d); } }
static orc_cpx sam_pll(orc_samdemod *d, orc_cpx x, double sgn)
{   /* sgn=-1: mono (:83-97), sgn=+1: stereo (:120-134) */
    double s = sgn * sin(d->phase), c = cos(d->phase), err;
    orc_cpx t;
    t.re = c * x.re - s * x.im;
    t.im = c * x.im + s * x.re;
    err = -sgn * atan2(t.im, t.re);
    d->freq += (d->beta * err);
    if (d->freq > d->hi) d->freq = d->hi;
    else if (d->freq < d->lo) d->freq = d->lo;
    d->phase += (d->freq + d->alpha * err);
    return t;
}
int orc_samdemod_process_mono(orc_samdemod *d, int n, const orc_cpx *in, double *out)
{
    int i;
    for (i = 0; i < n; i++) {
        orc_cpx t = sam_pll(d, in[i], -1.0);
        double z0 = t.re + (d->z1 * SAM_DC_ALPHA);
        out[i] = (z0 - d->z1);
        d->z1 = z0;
    }
    d->phase = fmod(d->phase, TWO_PI);
    return n;
}
int orc_samdemod_process_stereo(orc_samdemod *d, int n, const orc_cpx *in, orc_cpx *out)
{
    int i;
    for (i = 0; i < n; i++) {
        orc_cpx t = sam_pll(d, in[i], +1.0);
        double z0 = t.re + (d->z1 * SAM_DC_ALPHA), y0 = t.im + (d->y1 * SAM_DC_ALPHA);
        out[i].re = (z0 - d->z1);
        out[i].im = (y0 - d->y1);
        d->y1 = y0; d->z1 = z0;
    }
    d->phase = fmod(d->phase, TWO_PI);
    orc_fir_process_cpx(d->fir, n, out, out);
    for (i = 0; i < n; i++) {
        orc_cpx t = out[i];
        out[i].im = t.re - t.im;
        out[i].re = t.re + t.im;
    }
    return n;
}

/* ==================================================================================== */
/* NBFM (dsp/fmdemod.cpp)                                                                 */
/* ==================================================================================== */
struct orc_fmdemod {
    int squelched;
    double fs, hp_freq, out_gain, err_dc, dc_alpha, phase, freq, lo, hi, alpha, beta;
    double sq_thresh, sq_ave, sq_alpha;
    double tmp[FM_SQBUF], sq[FM_SQBUF];
    orc_fir *hp; orc_iir *lp;
};
static void fm_init_squelch(orc_fmdemod *d)
{ orc_fir_init_hp(d->hp, 1.0, 50.0, d->hp_freq, d->hp_freq * .6, d->fs); }
orc_fmdemod *orc_fmdemod_new(double fs)
{   /* ctor :62-89.  m_SquelchThreshold is not set there; 0 here (== forced mute). */
    orc_fmdemod *d = (orc_fmdemod *)zalloc(sizeof(*d));
    double norm = TWO_PI / fs;
    d->fs = fs;
    d->lo = -FM_PLL_RANGE * norm; d->hi = FM_PLL_RANGE * norm;
    d->alpha = 2.0 * FM_PLL_ZETA * FM_PLL_BW * norm;
    d->beta = (d->alpha * d->alpha) / (4.0 * FM_PLL_ZETA * FM_PLL_ZETA);
    d->out_gain = FM_MAX_OUT / d->hi;
    d->dc_alpha = (1.0 - exp(-1.0 / (fs * FM_DC_ALPHA)));
    d->hp_freq = FM_VOICE_BANDWIDTH;
    d->sq_ave = 0.0; d->squelched = 1;
    d->sq_alpha = (1.0 - exp(-1.0 / (fs * FM_SQUELCHAVE_TIMECONST)));
    d->hp = orc_fir_new(); d->lp = orc_iir_new();
    orc_iir_init(d->lp, 0, FM_VOICE_BANDWIDTH, 1.0, fs);
    fm_init_squelch(d);
    return d;
}
void orc_fmdemod_free(orc_fmdemod *d) { if (d) { orc_fir_free(d->hp); orc_iir_free(d->lp); free(d); } }
void orc_fmdemod_set_squelch(orc_fmdemod *d, int value)
{ d->sq_thresh = (double)(FM_SQUELCH_MAX - ((FM_SQUELCH_MAX * value) / 99)); }    /* :95-98 */
int orc_fmdemod_squelched(const orc_fmdemod *d) { return d->squelched; }

/* :113-152: one hysteresis decision per call, after the whole block's EMA */
static void fm_squelch(orc_fmdemod *d, int n, double *audio)
{
    int i;
    if (n > FM_SQBUF) return;
    orc_fir_process_real(d->hp, n, audio, d->sq);
    for (i = 0; i < n; i++)
        d->sq_ave = (1.0 - d->sq_alpha) * d->sq_ave + d->sq_alpha * fabs(d->sq[i]);
    if (0 == d->sq_thresh) d->squelched = 1;
    else if (d->squelched) { if (d->sq_ave < (d->sq_thresh - FM_SQUELCH_HYSTERESIS)) d->squelched = 0; }
    else { if (d->sq_ave >= (d->sq_thresh + FM_SQUELCH_HYSTERESIS)) d->squelched = 1; }
    if (d->squelched) for (i = 0; i < n; i++) audio[i] = 0.0;
    else orc_iir_process_real(d->lp, n, audio, audio);
}
static void fm_pll_block(orc_fmdemod *d, int n, double fm_bw, const orc_cpx *in, double *audio)
{   /* :157-192 */
    int i;
    if (d->hp_freq != fm_bw) { d->hp_freq = fm_bw; fm_init_squelch(d); }
    for (i = 0; i < n; i++) {
        double s = sin(d->phase), c = cos(d->phase), err;
        orc_cpx t;
        t.re = c * in[i].re - s * in[i].im;
        t.im = c * in[i].im + s * in[i].re;
        err = -atan2(t.im, t.re);
        d->freq += (d->beta * err);
        if (d->freq > d->hi) d->freq = d->hi;
        else if (d->freq < d->lo) d->freq = d->lo;
        d->phase += (d->freq + d->alpha * err);
        d->err_dc = (1.0 - d->dc_alpha) * d->err_dc + d->dc_alpha * d->freq;
        audio[i] = (d->freq - d->err_dc) * d->out_gain;
    }
    d->phase = fmod(d->phase, TWO_PI);
    fm_squelch(d, n, audio);
}
int orc_fmdemod_process_mono(orc_fmdemod *d, int n, double fm_bw, const orc_cpx *in, double *out)
{
    fm_pll_block(d, n, fm_bw, in, out);
    return n;
}
int orc_fmdemod_process_stereo(orc_fmdemod *d, int n, double fm_bw, const orc_cpx *in, orc_cpx *out)
{   /* :197-236 uses the member buffer (<=16384 samples) */
    int i;
    if (n > FM_SQBUF) n = FM_SQBUF;
    fm_pll_block(d, n, fm_bw, in, d->tmp);
    for (i = 0; i < n; i++) { out[i].re = d->tmp[i]; out[i].im = d->tmp[i]; }
    return n;
}

/* SSB/CW (dsp/ssbdemod.cpp:48-60) */
int orc_ssbdemod_process_mono(int n, const orc_cpx *in, double *out)
{ int i; for (i = 0; i < n; i++) out[i] = in[i].re; return n; }
int orc_ssbdemod_process_stereo(int n, const orc_cpx *in, orc_cpx *out)
{ int i; for (i = 0; i < n; i++) out[i] = in[i]; return n; }

/* ==================================================================================== */
/* CFractResampler (dsp/fractresampler.cpp)                                               */
/* ==================================================================================== */
struct orc_resampler { double t; double *sinc; orc_cpx *buf; int cap; };
orc_resampler *orc_resampler_new(void) { return (orc_resampler *)zalloc(sizeof(orc_resampler)); }
void orc_resampler_free(orc_resampler *r) { if (r) { free(r->sinc); free(r->buf); free(r); } }
void orc_resampler_init(orc_resampler *r, int max_input)
{   /* :85-135: Blackman-Harris windowed sinc, 10000 points per zero crossing */
    int i;
    max_input += RS_PERIODS;
    if (!r->sinc) r->sinc = (double *)zalloc(sizeof(double) * RS_LEN);
    free(r->buf);
    r->buf = (orc_cpx *)zalloc(sizeof(orc_cpx) * max_input);
    r->cap = max_input;
    for (i = 0; i < RS_LEN; i++) {
        double w = RS_WIN_A0
                 - RS_WIN_A1 * cos((TWO_PI * i) / (RS_LEN - 1))
                 + RS_WIN_A2 * cos((2.0 * TWO_PI * i) / (RS_LEN - 1))
                 - RS_WIN_A3 * cos((3.0 * TWO_PI * i) / (RS_LEN - 1));
        double fi = ONE_PI * (double)(i - RS_LEN / 2) / (double)RS_PTS;
        r->sinc[i] = (i != RS_LEN / 2) ? w * sin(fi) / fi : 1.0;
    }
    r->t = 0.0;
}
/* shared kernel of the four overloads (:144-352): outputs while floor(t) < n, each a
 * 28-tap dot with table index trunc((j-t)*10000); carries t-n and the last 28 inputs */
static int rs_run(orc_resampler *r, int n, double rate, int is_cpx, const void *in,
                  double *out_real, orc_cpx *out_cpx, short *out_i16, double gain)
{
    int i, j, it = (int)r->t, nout = 0;
    if (is_cpx) for (i = 0; i < n; i++) r->buf[RS_PERIODS + i] = ((const orc_cpx *)in)[i];
    else        for (i = 0; i < n; i++) r->buf[RS_PERIODS + i].re = ((const double *)in)[i];
    while (it < n) {
        double ar = 0.0, ai = 0.0;
        for (i = 1; i <= RS_PERIODS; i++) {
            int k;
            j = it + i;
            k = (int)(((double)j - r->t) * (double)RS_PTS);
            ar += (r->buf[j].re * r->sinc[k]);
            if (is_cpx) ai += (r->buf[j].im * r->sinc[k]);
        }
        if (out_i16) {
            double a = ar * gain, b = ai * gain;
            if (a > RS_MAX_SOUNDCARDVAL) a = RS_MAX_SOUNDCARDVAL;
            if (a < -RS_MAX_SOUNDCARDVAL) a = -RS_MAX_SOUNDCARDVAL;
            if (b > RS_MAX_SOUNDCARDVAL) b = RS_MAX_SOUNDCARDVAL;
            if (b < -RS_MAX_SOUNDCARDVAL) b = -RS_MAX_SOUNDCARDVAL;
            if (is_cpx) { out_i16[2 * nout] = (short)a; out_i16[2 * nout + 1] = (short)b; }
            else out_i16[nout] = (short)a;
        } else if (is_cpx) { out_cpx[nout].re = ar; out_cpx[nout].im = ai; }
        else out_real[nout] = ar;
        nout++;
        r->t += rate;
        it = (int)r->t;
    }
    r->t -= (double)n;
    if (is_cpx) for (i = 0; i < RS_PERIODS; i++) r->buf[i] = r->buf[n + i];
    else        for (i = 0; i < RS_PERIODS; i++) r->buf[i].re = r->buf[n + i].re;
    return nout;
}
int orc_resampler_real(orc_resampler *r, int n, double rate, const double *in, double *out)
{ return rs_run(r, n, rate, 0, in, out, NULL, NULL, 0); }
int orc_resampler_cpx(orc_resampler *r, int n, double rate, const orc_cpx *in, orc_cpx *out)
{ return rs_run(r, n, rate, 1, in, NULL, out, NULL, 0); }
int orc_resampler_real_i16(orc_resampler *r, int n, double rate, const double *in, short *out, double gain)
{ return rs_run(r, n, rate, 0, in, NULL, NULL, out, gain); }
int orc_resampler_cpx_i16(orc_resampler *r, int n, double rate, const orc_cpx *in, short *out, double gain)
{ return rs_run(r, n, rate, 1, in, NULL, NULL, out, gain); }

/* ==================================================================================== */
/* CDemodulator (dsp/demodulator.cpp)                                                     */
/* ==================================================================================== */
typedef struct { double *v; int n, cap; } tapvec;
struct orc_demod {
    orc_downconv *dc; orc_fastfir *ff; orc_agc *agc; orc_smeter *sm;
    orc_demod_info info;
    double in_rate, out_rate, want_bw, cw_off;
    orc_cpx *inbuf, *tmpbuf;
    int mode, pos, limit;
    orc_amdemod *am; orc_samdemod *sam; orc_fmdemod *fm; int ssb;
    int taps_on; tapvec tap[4];
    int perturb; double perturb_eps, perturb_max; unsigned long long perturb_state;    /* orc_demod_perturb_filter_output */
};
/* TEST-OF-THE-TESTS hook (tests/test_oracle_independent.py): what an fp32 filter in front of the fp64 stages does to
 * THIS chain's own output.  The filter output z of every pass is replaced by
 *   mode 1: z rounded to fp32;   mode 2: that, moved by -1 / 0 / +1 ulp(fp32) at random;
 *   mode 3: z + eps * M * u, u uniform in [-1, 1) per component, M = the largest filter INPUT component so far -- the
 *           ABSOLUTE error floor of an fp32 FFT filter (its rounding is relative to the transform's largest values:
 *           the product's K1 measures 3e-7 of the block's largest input sample), which is what reaches the AGC at
 *           full gain while the filter's OUTPUT is still starting up on samples of rounding size.
 *   mode 4: z + q * k, q = eps * M, k an integer uniform in [-3, 3] per component -- the same floor as a GRID: where the
 *           true output is far below it (the filter's start-up, 1e-12) an fp32 filter's samples are small multiples of one
 *           quantum (the product's 2048-point filter: multiples of 2^-15 for a carrier of 3277, eps = 1e-8), so their PHASES
 *           take a handful of values and can drive a PLL coherently -- noise of mode 3 cannot.
 * mode 0 (default): nothing -- the oracle proper.  The spread between mode 0 and the others is the room a start-up
 * tolerance of the GPU chain can claim, and no more. */
static double perturb_uniform(orc_demod *d)
{   /* SplitMix64 -> [-1, 1) */
    unsigned long long z = (d->perturb_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
static double perturb_f32(orc_demod *d, double v)
{
    float f = (float)v;
    if (d->perturb == 2) {
        const double u = perturb_uniform(d);
        if (u < -1.0 / 3.0) f = nextafterf(f, -INFINITY);
        else if (u > 1.0 / 3.0) f = nextafterf(f, INFINITY);
    }
    return (double)f;
}
static void perturb_filter_output(orc_demod *d, int n, orc_cpx *z)
{
    int i;
    if (d->perturb == 1 || d->perturb == 2) {
        for (i = 0; i < n; i++) { z[i].re = perturb_f32(d, z[i].re); z[i].im = perturb_f32(d, z[i].im); }
    } else if (d->perturb == 4) {
        const double q = d->perturb_eps * d->perturb_max;
        for (i = 0; i < n; i++) {
            z[i].re += q * floor(3.5 * (perturb_uniform(d) + 1.0) - 3.0);      /* -3 .. 3 */
            z[i].im += q * floor(3.5 * (perturb_uniform(d) + 1.0) - 3.0);
        }
    } else if (d->perturb == 3) {
        const double mx = d->perturb_max;
        for (i = 0; i < n; i++) { z[i].re += d->perturb_eps * mx * perturb_uniform(d); z[i].im += d->perturb_eps * mx * perturb_uniform(d); }
    }
}
void orc_demod_perturb_filter_output(orc_demod *d, int mode, double eps, unsigned long long seed)
{ d->perturb = mode; d->perturb_eps = eps; d->perturb_state = seed; d->perturb_max = 0.0; }
static void tap_push(tapvec *t, const double *src, int ndoubles)
{
    if (t->n + ndoubles > t->cap) {
        t->cap = (t->n + ndoubles) * 2 + 1024;
        t->v = (double *)realloc(t->v, sizeof(double) * t->cap);
    }
    memcpy(t->v + t->n, src, sizeof(double) * ndoubles);
    t->n += ndoubles;
}
static void demod_drop(orc_demod *d)
{
    orc_amdemod_free(d->am); orc_samdemod_free(d->sam); orc_fmdemod_free(d->fm);
    d->am = NULL; d->sam = NULL; d->fm = NULL; d->ssb = 0;
}
void orc_demod_set_freq(orc_demod *d, double f)
{   /* dsp/demodulator.h:68-69 */
    orc_downconv_set_cw_offset(d->dc, d->cw_off);
    orc_downconv_set_frequency(d->dc, f);
}
orc_demod *orc_demod_new(int fastfir_n)
{   /* ctor dsp/demodulator.cpp:47-60 in zeroed storage */
    orc_demod *d = (orc_demod *)zalloc(sizeof(*d));
    d->dc = orc_downconv_new(); d->ff = orc_fastfir_new(fastfir_n);
    d->agc = orc_agc_new(); d->sm = orc_smeter_new();
    d->want_bw = 48000.0; d->out_rate = 48000.0;
    d->inbuf = (orc_cpx *)zalloc(sizeof(orc_cpx) * DEMOD_BUF);
    d->tmpbuf = (orc_cpx *)zalloc(sizeof(orc_cpx) * DEMOD_BUF);
    d->limit = 1000; d->mode = -1;
    orc_demod_set_freq(d, 0.0);
    return d;
}
void orc_demod_free(orc_demod *d)
{
    int i;
    if (!d) return;
    demod_drop(d);
    orc_downconv_free(d->dc); orc_fastfir_free(d->ff); orc_agc_free(d->agc); orc_smeter_free(d->sm);
    free(d->inbuf); free(d->tmpbuf);
    for (i = 0; i < 4; i++) free(d->tap[i].v);
    free(d);
}
void orc_demod_set_input_rate(orc_demod *d, double rate)
{   /* :92-99 */
    if (d->in_rate != rate) {
        d->in_rate = rate;
        d->out_rate = orc_downconv_set_data_rate(d->dc, d->in_rate, d->want_bw);
    }
}
/* :107-157 */
void orc_demod_set_demod(orc_demod *d, int mode, const orc_demod_info *info)
{
    d->info = *info;
    if (d->mode != mode) {
        demod_drop(d);
        d->mode = mode;
        if (mode == ORC_DEMOD_LSB || mode == ORC_DEMOD_CWL) d->want_bw = -d->info.LowCutmin;
        else d->want_bw = d->info.HiCutmax;
        d->out_rate = orc_downconv_set_data_rate(d->dc, d->in_rate, d->want_bw);
        switch (mode) {
        case ORC_DEMOD_AM:  d->am = orc_amdemod_new(d->out_rate); break;
        case ORC_DEMOD_SAM: d->sam = orc_samdemod_new(d->out_rate); break;
        case ORC_DEMOD_FM:  d->fm = orc_fmdemod_new(d->out_rate); break;
        case ORC_DEMOD_USB: case ORC_DEMOD_LSB: case ORC_DEMOD_CWU: case ORC_DEMOD_CWL:
            d->ssb = 1; break;
        }
    }
    d->cw_off = d->info.Offset;
    orc_downconv_set_cw_offset(d->dc, d->cw_off);
    orc_fastfir_setup(d->ff, d->info.LowCut, d->info.HiCut, d->cw_off, d->out_rate);
    d->limit = (int)((d->out_rate / 100.0) * d->in_rate / d->out_rate);
    d->limit &= 0xFFFFFF00;
    orc_agc_set(d->agc, d->info.AgcOn, d->info.AgcHangOn, d->info.AgcThresh,
                d->info.AgcManualGain, d->info.AgcSlope, d->info.AgcDecay, d->out_rate);
    if (d->fm) orc_fmdemod_set_squelch(d->fm, d->info.SquelchValue);
    if (d->am) orc_amdemod_set_bandwidth(d->am, (d->info.HiCut - d->info.LowCut) / 2.0);
}
double orc_demod_output_rate(const orc_demod *d) { return d->out_rate; }
double orc_demod_smeter_peak(orc_demod *d) { return orc_smeter_peak(d->sm); }
double orc_demod_smeter_ave(orc_demod *d) { return orc_smeter_ave(d->sm); }
int orc_demod_buf_limit(const orc_demod *d) { return d->limit; }
void orc_demod_enable_taps(orc_demod *d, int on) { d->taps_on = on; }
int orc_demod_tap_len(const orc_demod *d, int tap) { return d->tap[tap - 1].n; }
const double *orc_demod_tap_data(const orc_demod *d, int tap) { return d->tap[tap - 1].v; }
void orc_demod_clear_taps(orc_demod *d) { int i; for (i = 0; i < 4; i++) d->tap[i].n = 0; }

/* one pass of the chain over the filled input buffer (:172-207) */
static int demod_chain(orc_demod *d, int stereo, double *out_real, orc_cpx *out_cpx)
{
    int n = orc_downconv_process(d->dc, d->pos, d->inbuf, d->inbuf);
    if (d->taps_on) tap_push(&d->tap[0], (double *)d->inbuf, 2 * n);
    if (d->perturb >= 3) {
        int i;
        for (i = 0; i < n; i++) {
            const double a = fabs(d->inbuf[i].re), b = fabs(d->inbuf[i].im);
            if (a > d->perturb_max) d->perturb_max = a;
            if (b > d->perturb_max) d->perturb_max = b;
        }
    }
    n = orc_fastfir_process(d->ff, n, d->inbuf, d->tmpbuf);
    if (d->perturb) perturb_filter_output(d, n, d->tmpbuf);
    if (d->taps_on) tap_push(&d->tap[1], (double *)d->tmpbuf, 2 * n);
    orc_smeter_process(d->sm, n, d->tmpbuf, d->out_rate);
    orc_agc_process_cpx(d->agc, n, d->tmpbuf, d->tmpbuf);
    if (d->taps_on) tap_push(&d->tap[2], (double *)d->tmpbuf, 2 * n);
    if (stereo) {
        if (d->am) n = orc_amdemod_process_stereo(d->am, n, d->tmpbuf, out_cpx);
        else if (d->sam) n = orc_samdemod_process_stereo(d->sam, n, d->tmpbuf, out_cpx);
        else if (d->fm) n = orc_fmdemod_process_stereo(d->fm, n, d->info.HiCut, d->tmpbuf, out_cpx);
        else if (d->ssb) n = orc_ssbdemod_process_stereo(n, d->tmpbuf, out_cpx);
        if (d->taps_on) tap_push(&d->tap[3], (double *)out_cpx, 2 * n);
    } else {
        if (d->am) n = orc_amdemod_process_mono(d->am, n, d->tmpbuf, out_real);
        else if (d->sam) n = orc_samdemod_process_mono(d->sam, n, d->tmpbuf, out_real);
        else if (d->fm) n = orc_fmdemod_process_mono(d->fm, n, d->info.HiCut, d->tmpbuf, out_real);
        else if (d->ssb) n = orc_ssbdemod_process_mono(n, d->tmpbuf, out_real);
        if (d->taps_on) tap_push(&d->tap[3], out_real, n);
    }
    d->pos = 0;
    return n;
}
/* :163-215 -- every pass writes at out[0]; the return value is the SUM */
int orc_demod_process_mono(orc_demod *d, int n, const orc_cpx *in, double *out)
{
    int i, ret = 0;
    for (i = 0; i < n; i++) {
        d->inbuf[d->pos++] = in[i];
        if (d->pos >= d->limit) ret += demod_chain(d, 0, out, NULL);
    }
    return ret;
}
int orc_demod_process_stereo(orc_demod *d, int n, const orc_cpx *in, orc_cpx *out)
{
    int i, ret = 0;
    for (i = 0; i < n; i++) {
        d->inbuf[d->pos++] = in[i];
        if (d->pos >= d->limit) ret += demod_chain(d, 1, NULL, out);
    }
    return ret;
}
int orc_demod_process_mono_append(orc_demod *d, int n, const orc_cpx *in, double *out)
{
    int i, ret = 0;
    for (i = 0; i < n; i++) {
        d->inbuf[d->pos++] = in[i];
        if (d->pos >= d->limit) ret += demod_chain(d, 0, out + ret, NULL);
    }
    return ret;
}

/* ==================================================================================== */
/* CNoiseProc::ProcessBlanker  (dsp/noiseproc.cpp:78-176)  -- SURVEY 8(f) row f1           */
/* ==================================================================================== */
struct orc_noiseproc {
    int on, dptr, mptr, blank, delay_n, mag_n, width_n, configured;
    double thresh, width, fs, ratio, sum;
    double dly[2 * NB_MAX_DELAY], mag[NB_MAX_AVE];
};
orc_noiseproc *orc_noiseproc_new(void)
{
    orc_noiseproc *p = (orc_noiseproc *)zalloc(sizeof(*p));
    p->thresh = -1;                                      /* ctor: SetupBlanker(false, 50, 2, 1000), :66 */
    orc_noiseproc_setup(p, 0, 50.0, 2.0, 1000.0);
    return p;
}
void orc_noiseproc_free(orc_noiseproc *p) { free(p); }
int orc_noiseproc_setup(orc_noiseproc *p, int on, double thresh, double width, double fs)
{
    /* :80-86: returns early when threshold, width and on are unchanged -- the sample-rate term of
     * the test is `SampleRate==SampleRate`, always true, so a rate-only change is ignored */
    if (p->configured && thresh == p->thresh && width == p->width && p->on == on) return 0;
    p->configured = 1;
    p->on = on; p->thresh = thresh; p->width = width; p->fs = fs;
    p->width_n = (int)(width * 1e-6 * fs);               /* :92-96 */
    if (p->width_n < 1) p->width_n = 1;
    else if (p->width_n > NB_MAX_WIDTH) p->width_n = NB_MAX_WIDTH;
    p->mag_n = (int)(NB_MAGAVE_TIME * fs);               /* :98 */
    if (p->mag_n > NB_MAX_AVE - 1) return -1;            /* the reference would overrun m_MagBuf here */
    p->ratio = .005 * thresh * (double)p->mag_n;         /* :100 */
    p->delay_n = p->width_n / 2;                         /* :102 */
    p->dptr = p->mptr = p->blank = 0; p->sum = 0.0;
    memset(p->dly, 0, sizeof(p->dly)); memset(p->mag, 0, sizeof(p->mag));
    return 1;
}
/* in place allowed (the host calls it in place, sdrinterface.cpp:884); off: output untouched (:125-129) */
void orc_noiseproc_process(orc_noiseproc *p, int n, const double *in, double *out)
{
    int i;
    if (!p->on) return;
    for (i = 0; i < n; i++) {
        const double re = in[2 * i], im = in[2 * i + 1];
        const double mre = fabs(re), mim = fabs(im), mag = (mre > mim) ? mre : mim;   /* :137-139 */
        double ore, oim;
        p->sum -= p->mag[p->mptr];                       /* :143-147: moving sum over mag_n+1 entries */
        p->sum += mag;
        p->mag[p->mptr++] = mag;
        if (p->mptr > p->mag_n) p->mptr = 0;
        ore = p->dly[2 * p->dptr]; oim = p->dly[2 * p->dptr + 1];      /* :150-153: delay of delay_n+1 */
        p->dly[2 * p->dptr] = re; p->dly[2 * p->dptr + 1] = im;
        if (++p->dptr > p->delay_n) p->dptr = 0;
        if (mag * p->ratio > p->sum) p->blank = p->width_n;            /* :155-158 */
        if (p->blank) { p->blank--; out[2 * i] = 0.0; out[2 * i + 1] = 0.0; }   /* :160-166 */
        else { out[2 * i] = ore; out[2 * i + 1] = oim; }
    }
}

/* ==================================================================================== */
/* IQ wire format -> samples  (interface/netiobase.cpp:479-527)  -- SURVEY 8(f) row f2      */
/* ==================================================================================== */
/* one UDP packet: 4 header bytes, then little-endian I,Q,I,Q...; 1028 bytes = 256 samples of
 * 16 bit, 1444 bytes = 240 samples of 24 bit scaled by 1/256 onto the 16-bit range.
 * Returns the number of complex samples written, -1 for any other packet length. */
int orc_unpack_packet(const unsigned char *pkt, int len, double *out)
{
    int i, j;
    if (len == 1444) {
        for (i = 4, j = 0; i < len; i += 3, j++) {
            const uint32_t u = ((uint32_t)pkt[i] << 8) | ((uint32_t)pkt[i + 1] << 16) | ((uint32_t)pkt[i + 2] << 24);
            out[j] = (double)(int32_t)u / 65536.0;       /* :497-503 */
        }
        return j / 2;
    }
    if (len == 1028) {
        for (i = 4, j = 0; i < len; i += 2, j++)
            out[j] = (double)(int16_t)((uint16_t)pkt[i] | ((uint16_t)pkt[i + 1] << 8));   /* :521-526 */
        return j / 2;
    }
    return -1;
}
/* CSdrInterface::NcoSpurCalibrate (interface/sdrinterface.cpp:829-848): running I/Q means,
 * alpha = 1e-5, over interleaved doubles; dc[0] = I offset, dc[1] = Q offset */
void orc_spurcal(double *dc, int n_doubles, const double *data)
{
    int i;
    for (i = 0; i < n_doubles; i++) {
        if (i & 1) dc[1] = (1.0 - 1.0 / 100000.0) * dc[1] + (1.0 / 100000.0) * data[i];
        else       dc[0] = (1.0 - 1.0 / 100000.0) * dc[0] + (1.0 / 100000.0) * data[i];
    }
}


/* ==================================================================================== */
/* CSoundOut queue + rate-error loop (interface/soundout.cpp)  -- SURVEY 8(f) row f3        */
/* ==================================================================================== */
/* Blocking mode (Start(..., BlockingMode), :86-90): PutOutQueue waits (msleep(10)) while the queue is full and
 * drops nothing (:209-220, :267-278); GetOutQueue returns right after popping (:354-358, :428-432).  This
 * single-threaded restatement cannot wait: its blocking put queues what fits and returns -1 - (samples left over)
 * when the reference would have started to sleep -- the tests drive it so that this never happens.
 * The non-blocking mode of the sound sink: PutOutQueue resamples to the sound-card rate with
 * Rate = m_OutRatio * (1 + m_RateCorrection) and the volume gain, pushes into a 16384-entry ring
 * (overflow: drop a quarter of the queue), GetOutQueue pops for the audio thread (start-up silence
 * until the queue is half full, underflow: back up a quarter), both track the average fill level,
 * and once per second of consumed samples CalcError sets the correction from it.               */
#define SS_OUTQSIZE 16384               /* soundout.h:18 */
#define SS_RATE 48000                   /* soundout.cpp:48 */
#define SS_ALPHA 0.001                  /* soundout.cpp:51 */
#define SS_PGAIN 2.38e-7                /* soundout.cpp:52 */
struct orc_soundsink {
    orc_resampler *rs;
    int stereo, startup, blocking;
    double user_rate, out_ratio, rate_corr, gain, ave_level;
    int head, tail, level, rate_count, ppm;
    short q[2 * SS_OUTQSIZE];
};
orc_soundsink *orc_soundsink_new(int stereo)
{
    orc_soundsink *s = (orc_soundsink *)zalloc(sizeof(*s));     /* ctor soundout.cpp:60-76 */
    s->rs = orc_resampler_new();
    orc_resampler_init(s->rs, 8192);
    s->stereo = stereo; s->user_rate = SS_RATE; s->out_ratio = 1.0; s->rate_corr = 0.0; s->gain = 1.0; s->startup = 1;
    return s;
}
void orc_soundsink_free(orc_soundsink *s) { if (s) { orc_resampler_free(s->rs); free(s); } }
void orc_soundsink_change_rate(orc_soundsink *s, double rate)   /* ChangeUserDataRate :155-175 */
{
    if (s->user_rate != rate) {
        s->user_rate = rate;
        memset(s->q, 0, sizeof(s->q));
        s->out_ratio = rate / (double)SS_RATE;
        s->head = s->tail = s->level = 0;
        s->ave_level = SS_OUTQSIZE / 2;
        s->startup = 1;
    }
}
void orc_soundsink_set_blocking(orc_soundsink *s, int on) { s->blocking = on != 0; }      /* Start :86-90 */
void orc_soundsink_set_volume(orc_soundsink *s, int vol)        /* SetVolume :180-189 */
{
    if (vol == 0) s->gain = 0.0;
    else if (vol <= 99) s->gain = pow(10.0, ((double)vol - 99.0) / 39.2);
}
static void ss_calc_error(orc_soundsink *s)                     /* CalcError :456-468 */
{
    double error = (double)(s->ave_level - SS_OUTQSIZE / 2);
    error = error * SS_PGAIN;
    s->rate_corr = error;
    s->ppm = (int)(s->rate_corr * 1e6);
}
/* PutOutQueue, non-blocking branch (:196-247 stereo, :254-305 mono); in: n reals or n complex pairs.
 * Returns the resampled samples that were produced (queued unless the overflow rule dropped them). */
int orc_soundsink_put(orc_soundsink *s, int n, const double *in)
{
    static short r[2 * SS_OUTQSIZE];
    int i, overflow = 0, k;
    const double rate = 1.0 * s->out_ratio * (1.0 + s->rate_corr);
    if (n == 0) return 0;
    k = s->stereo ? orc_resampler_cpx_i16(s->rs, n, rate, (const orc_cpx *)in, r, s->gain)
                  : orc_resampler_real_i16(s->rs, n, rate, in, r, s->gain);
    if (s->blocking) {                                           /* :209-220 / :267-278 */
        for (i = 0; i < k; i++) {
            if (((s->head + 1) & (SS_OUTQSIZE - 1)) == s->tail) return -1 - (k - i);     /* the reference sleeps here */
            if (s->stereo) { s->q[2 * s->head] = r[2 * i]; s->q[2 * s->head + 1] = r[2 * i + 1]; }
            else s->q[s->head] = r[i];
            s->head = (s->head + 1) & (SS_OUTQSIZE - 1);
            s->level++;
        }
        return k;
    }
    for (i = 0; i < k; i++) {
        if (s->stereo) { s->q[2 * s->head] = r[2 * i]; s->q[2 * s->head + 1] = r[2 * i + 1]; }
        else s->q[s->head] = r[i];
        s->head = (s->head + 1) & (SS_OUTQSIZE - 1);
        s->level++;
        if (s->head == s->tail) {                                /* full: drop a quarter of the queue */
            s->tail = (s->tail + SS_OUTQSIZE / 4) & (SS_OUTQSIZE - 1);
            s->level -= SS_OUTQSIZE / 4;
            overflow = 1;
            break;
        }
    }
    if (overflow) s->ave_level = s->level;
    s->ave_level = (1.0 - SS_ALPHA) * s->ave_level + SS_ALPHA * (double)s->level;
    return k;
}
/* GetOutQueue (:311-375 mono, :381-445 stereo): n samples (mono) or n L/R pairs (stereo) */
void orc_soundsink_get(orc_soundsink *s, int n, short *out)
{
    int i, underflow = 0;
    const int w = s->stereo ? 2 : 1;
    if (s->startup) {
        for (i = 0; i < w * n; i++) out[i] = 0;
        if (s->level > SS_OUTQSIZE / 2) {
            s->startup = 0;
            s->rate_count = -5 * SS_RATE;
            s->ppm = 0;
            s->ave_level = s->level;
        } else return;
    }
    for (i = 0; i < n; i++) {
        if (s->head != s->tail) {
            if (s->stereo) { out[2 * i] = s->q[2 * s->tail]; out[2 * i + 1] = s->q[2 * s->tail + 1]; }
            else out[i] = s->q[s->tail];
            s->tail = (s->tail + 1) & (SS_OUTQSIZE - 1);
            s->level--;
        } else {                                                 /* empty: back up and repeat older data */
            s->tail = (s->tail - SS_OUTQSIZE / 4) & (SS_OUTQSIZE - 1);
            if (s->stereo) { out[2 * i] = s->q[2 * s->tail]; out[2 * i + 1] = s->q[2 * s->tail + 1]; }
            else out[i] = s->q[s->tail];
            s->level += SS_OUTQSIZE / 4;
            underflow = 1;
        }
    }
    if (s->blocking) return;                                     /* :354-358 / :428-432 */
    s->ave_level = (1.0 - SS_ALPHA) * s->ave_level + SS_ALPHA * s->level;
    if (underflow) s->ave_level = s->level;
    s->rate_count += n;
    if (s->rate_count >= SS_RATE) { ss_calc_error(s); s->rate_count = 0; }
}
double orc_soundsink_rate_correction(const orc_soundsink *s) { return s->rate_corr; }
double orc_soundsink_ave_level(const orc_soundsink *s) { return s->ave_level; }
int orc_soundsink_level(const orc_soundsink *s) { return s->level; }
int orc_soundsink_ppm(const orc_soundsink *s) { return s->ppm; }
