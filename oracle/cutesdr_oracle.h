/*
 * cutesdr_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of the CuteSDR dsp/ receive chain (reference: muonzoo/cutesdr,
 * /root/reference/dsp/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / timed baseline.
 * Nothing under cutesdr_amd/ links, imports or calls it.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"): the reference cannot be built in this image
 * (every dsp/ translation unit includes Qt headers -- <QMutex>, <QtGui/QApplication>,
 * <QDebug> -- and Qt is not installed; writing stand-ins for them is not allowed), and
 * the reference ships no tests, fixtures or golden vectors.  The
 * oracle is therefore pinned only by (i) the known-answer anchors that SURVEY.md
 * section 8c / App. A.9 recorded from a run of the reference, checked in
 * tests/test_oracle_anchors.py, and (ii) independent numpy/scipy restatements of the
 * published algorithms (DFT, windowed-sinc, Kaiser, RBJ biquad).  Where neither
 * reaches, parity is "unpinned" and says so in the test names.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef CUTESDR_ORACLE_H
#define CUTESDR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } orc_cpx;

/* ---- the reference's named constants as this file USES them ("<reference file>:<its #define>" -> value; the
 * window coefficients as "<file>:WIN_A<k>").  Fills up to cap entries, returns how many there are.
 * tests/test_reference_constants.py compares every entry with the reference's text (build container only). */
int orc_constants(const char **names, double *values, int cap);

/* ---- complex FFT, reference sign conventions (dsp/fft.cpp:416-426) ---------------
 * sign=+1: X[k]=sum x[n] e^{+j2pi nk/N} (CFft::FwdFFT); sign=-1: CFft::RevFFT.
 * Unnormalised, natural order in and out, n a power of two. */
void orc_fft(int n, int sign, orc_cpx *a);

/* ---- CFft display path (dsp/fft.cpp:118-410, 510-591) ----------------------------- */
typedef struct orc_cfft orc_cfft;
orc_cfft *orc_cfft_new(void);
void orc_cfft_free(orc_cfft *f);
void orc_cfft_set_params(orc_cfft *f, int size, int invert, double db_comp, double fs);
void orc_cfft_set_ave(orc_cfft *f, int ave);
void orc_cfft_reset(orc_cfft *f);
int  orc_cfft_put_display(orc_cfft *f, int n, const orc_cpx *in);
void orc_plotter_color_table(unsigned int *out256);
int  orc_plotter_waterfall_line(orc_cfft *f, int max_w, double max_db, double min_db, int start_hz, int stop_hz,
                                int *levels, unsigned int *rgb);
int  orc_cfft_get_screen(orc_cfft *f, int max_h, int max_w, double max_db, double min_db,
                         int start_hz, int stop_hz, int *out);
void orc_cfft_fwd(orc_cfft *f, orc_cpx *a);      /* includes the power/log10 side effect */
void orc_cfft_rev(orc_cfft *f, orc_cpx *a);
int  orc_cfft_size(const orc_cfft *f);
const double *orc_cfft_avebuf(const orc_cfft *f); /* m_pFFTAveBuf (bels, fft-shifted) */

/* ---- CFastFIR (dsp/fastfir.cpp), FFT size parameterised ---------------- */
typedef struct orc_fastfir orc_fastfir;
orc_fastfir *orc_fastfir_new(int fft_size);       /* taps = fft_size/2+1 */
void orc_fastfir_free(orc_fastfir *f);
void orc_fastfir_set_faithful(orc_fastfir *f, int on); /* keep F4 log10 side effect */
int  orc_fastfir_setup(orc_fastfir *f, double flo, double fhi, double offset, double fs);
int  orc_fastfir_process(orc_fastfir *f, int n, const orc_cpx *in, orc_cpx *out);
const orc_cpx *orc_fastfir_coef(const orc_fastfir *f);  /* H[k], natural order */

/* ---- CDownConvert (dsp/downconvert.cpp) ---------------------------------------------- */
typedef struct orc_downconv orc_downconv;
orc_downconv *orc_downconv_new(void);
void orc_downconv_free(orc_downconv *d);
void orc_downconv_set_cw_offset(orc_downconv *d, double off);
void orc_downconv_set_frequency(orc_downconv *d, double f);
double orc_downconv_set_data_rate(orc_downconv *d, double in_rate, double max_bw);
int  orc_downconv_process(orc_downconv *d, int n, orc_cpx *in, orc_cpx *out);
int  orc_downconv_stages(const orc_downconv *d, int *codes); /* 3=CIC3 else HB length */
double orc_downconv_nco_freq(const orc_downconv *d);

/* ---- CFir (dsp/fir.cpp) ------------------------------------------------------------ */
typedef struct orc_fir orc_fir;
orc_fir *orc_fir_new(void);
void orc_fir_free(orc_fir *f);
void orc_fir_init_const(orc_fir *f, int ntaps, const double *coef);
int  orc_fir_init_lp(orc_fir *f, double scale, double astop, double fpass, double fstop, double fs);
int  orc_fir_init_hp(orc_fir *f, double scale, double astop, double fpass, double fstop, double fs);
void orc_fir_gen_hilbert(orc_fir *f, double freq_offset);
void orc_fir_process_real(orc_fir *f, int n, const double *in, double *out);
void orc_fir_process_cpx(orc_fir *f, int n, const orc_cpx *in, orc_cpx *out);
int  orc_fir_taps(const orc_fir *f, double *coef, double *icoef, double *qcoef);

/* ---- CIir (dsp/iir.cpp) ------------------------------------------------------------ */
typedef struct orc_iir orc_iir;
orc_iir *orc_iir_new(void);
void orc_iir_free(orc_iir *f);
void orc_iir_init(orc_iir *f, int kind /*0 LP,1 HP,2 BP,3 BR*/, double f0, double q, double fs);
void orc_iir_process_real(orc_iir *f, int n, const double *in, double *out);
void orc_iir_process_cpx(orc_iir *f, int n, const orc_cpx *in, orc_cpx *out);
void orc_iir_coefs(const orc_iir *f, double *b0b1b2a1a2);

/* ---- CAgc (dsp/agc.cpp) ------------------------------------------------------------ */
typedef struct orc_agc orc_agc;
orc_agc *orc_agc_new(void);
void orc_agc_free(orc_agc *a);
void orc_agc_set(orc_agc *a, int on, int hang, int thresh, int manual_gain, int slope,
                 int decay, double fs);
void orc_agc_process_cpx(orc_agc *a, int n, const orc_cpx *in, orc_cpx *out);
void orc_agc_process_real(orc_agc *a, int n, const double *in, double *out);

/* ---- CSMeter (dsp/smeter.cpp) -------------------------------------------------------- */
typedef struct orc_smeter orc_smeter;
orc_smeter *orc_smeter_new(void);
void orc_smeter_free(orc_smeter *s);
void orc_smeter_process(orc_smeter *s, int n, const orc_cpx *in, double fs);
double orc_smeter_peak(orc_smeter *s);
double orc_smeter_ave(orc_smeter *s);

/* ---- demodulators (dsp/amdemod.cpp, samdemod.cpp, fmdemod.cpp, ssbdemod.cpp) ------- */
typedef struct orc_amdemod orc_amdemod;
orc_amdemod *orc_amdemod_new(double fs);
void orc_amdemod_free(orc_amdemod *d);
void orc_amdemod_set_bandwidth(orc_amdemod *d, double bw);
int  orc_amdemod_process_mono(orc_amdemod *d, int n, const orc_cpx *in, double *out);
int  orc_amdemod_process_stereo(orc_amdemod *d, int n, const orc_cpx *in, orc_cpx *out);

typedef struct orc_samdemod orc_samdemod;
orc_samdemod *orc_samdemod_new(double fs);
void orc_samdemod_free(orc_samdemod *d);
int  orc_samdemod_process_mono(orc_samdemod *d, int n, const orc_cpx *in, double *out);
int  orc_samdemod_process_stereo(orc_samdemod *d, int n, const orc_cpx *in, orc_cpx *out);

typedef struct orc_fmdemod orc_fmdemod;
orc_fmdemod *orc_fmdemod_new(double fs);
void orc_fmdemod_free(orc_fmdemod *d);
void orc_fmdemod_set_squelch(orc_fmdemod *d, int value);
int  orc_fmdemod_process_mono(orc_fmdemod *d, int n, double fm_bw, const orc_cpx *in, double *out);
int  orc_fmdemod_process_stereo(orc_fmdemod *d, int n, double fm_bw, const orc_cpx *in, orc_cpx *out);
int  orc_fmdemod_squelched(const orc_fmdemod *d);

int  orc_ssbdemod_process_mono(int n, const orc_cpx *in, double *out);
int  orc_ssbdemod_process_stereo(int n, const orc_cpx *in, orc_cpx *out);

/* ---- CFractResampler (dsp/fractresampler.cpp) ---------------------------------------- */
typedef struct orc_resampler orc_resampler;
orc_resampler *orc_resampler_new(void);
void orc_resampler_free(orc_resampler *r);
void orc_resampler_init(orc_resampler *r, int max_input);
int  orc_resampler_real(orc_resampler *r, int n, double rate, const double *in, double *out);
int  orc_resampler_cpx(orc_resampler *r, int n, double rate, const orc_cpx *in, orc_cpx *out);
int  orc_resampler_real_i16(orc_resampler *r, int n, double rate, const double *in, short *out, double gain);
int  orc_resampler_cpx_i16(orc_resampler *r, int n, double rate, const orc_cpx *in, short *out, double gain);

/* ---- CDemodulator (dsp/demodulator.cpp) ------------------------------------------------ */
typedef struct {            /* POD mirror of tDemodInfo (dsp/demodulator.h:35-54), no txt */
    int HiCut, HiCutmin, HiCutmax, LowCut, LowCutmin, LowCutmax;
    int FilterClickResolution, Offset, SquelchValue;
    int AgcSlope, AgcThresh, AgcManualGain, AgcDecay;
    int AgcOn, AgcHangOn, Symetric;
} orc_demod_info;

enum { ORC_DEMOD_AM = 0, ORC_DEMOD_SAM, ORC_DEMOD_FM, ORC_DEMOD_USB, ORC_DEMOD_LSB,
       ORC_DEMOD_CWU, ORC_DEMOD_CWL };

typedef struct orc_demod orc_demod;
orc_demod *orc_demod_new(int fastfir_n);
void orc_demod_free(orc_demod *d);
void orc_demod_set_input_rate(orc_demod *d, double rate);
void orc_demod_set_demod(orc_demod *d, int mode, const orc_demod_info *info);
void orc_demod_set_freq(orc_demod *d, double f);
double orc_demod_output_rate(const orc_demod *d);
double orc_demod_smeter_peak(orc_demod *d);
double orc_demod_smeter_ave(orc_demod *d);
int  orc_demod_buf_limit(const orc_demod *d);
/* reference semantics incl. the F8 overwrite quirk (dsp/demodulator.cpp:163-215) */
int  orc_demod_process_mono(orc_demod *d, int n, const orc_cpx *in, double *out);
int  orc_demod_process_stereo(orc_demod *d, int n, const orc_cpx *in, orc_cpx *out);
/* same chain, but appends every inner pass's output (batch harness form) */
int  orc_demod_process_mono_append(orc_demod *d, int n, const orc_cpx *in, double *out);
/* stage taps (the reference's DisplayData PROFILE_1..4 points, demodulator.cpp:175-208):
 * when enabled, each inner pass appends its stage buffers here. tap: 1..4 */
void orc_demod_enable_taps(orc_demod *d, int on);
/* test-of-the-tests hook: perturb the filter output of every pass (mode 0 off, 1 round to fp32, 2 fp32 +-1 ulp at random,
 * 3 additive eps * max|z| uniform noise); see cutesdr_oracle.c */
void orc_demod_perturb_filter_output(orc_demod *d, int mode, double eps, unsigned long long seed);
int  orc_demod_tap_len(const orc_demod *d, int tap);
const double *orc_demod_tap_data(const orc_demod *d, int tap); /* cpx interleaved for 1-3; real for 4 */
void orc_demod_clear_taps(orc_demod *d);

/* CNoiseProc (dsp/noiseproc.h:23-58) */
typedef struct orc_noiseproc orc_noiseproc;
orc_noiseproc *orc_noiseproc_new(void);
void orc_noiseproc_free(orc_noiseproc *p);
int orc_noiseproc_setup(orc_noiseproc *p, int on, double thresh, double width, double fs);
void orc_noiseproc_process(orc_noiseproc *p, int n, const double *in, double *out);
/* wire format (interface/netiobase.cpp:479-527) and NCO-spur DC estimate (sdrinterface.cpp:829-848) */
int orc_unpack_packet(const unsigned char *pkt, int len, double *out);
void orc_spurcal(double *dc, int n_doubles, const double *data);

/* CSoundOut queue + rate-error loop, non-blocking mode (interface/soundout.cpp:155-468) */
typedef struct orc_soundsink orc_soundsink;
orc_soundsink *orc_soundsink_new(int stereo);
void orc_soundsink_free(orc_soundsink *s);
void orc_soundsink_change_rate(orc_soundsink *s, double rate);
void orc_soundsink_set_volume(orc_soundsink *s, int vol);
void orc_soundsink_set_blocking(orc_soundsink *s, int on);
int orc_soundsink_put(orc_soundsink *s, int n, const double *in);
void orc_soundsink_get(orc_soundsink *s, int n, short *out);
double orc_soundsink_rate_correction(const orc_soundsink *s);
double orc_soundsink_ave_level(const orc_soundsink *s);
int orc_soundsink_level(const orc_soundsink *s);
int orc_soundsink_ppm(const orc_soundsink *s);

#ifdef __cplusplus
}
#endif
#endif
