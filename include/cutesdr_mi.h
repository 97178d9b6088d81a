/*
 * cutesdr_mi.h -- C ABI of libcutesdr_mi.so, the MI355X (gfx950) implementation of the
 * CuteSDR dsp/ receive chain.
 *
 * The reference has no plugin/FFI layer: its host code instantiates the dsp/
 * classes by value and calls their public methods.  The drop-in boundary is therefore this C
 * ABI plus the same-named C++ classes in cutesdr_amd/dropin/dsp/ that forward to it.  Each
 * entry point cites the reference method it stands in for (file:line in the reference tree).
 *
 * Conventions
 *  - opaque handles, plain pointers and sizes, no C++ or torch types;
 *  - "host" entry points take interleaved double I/Q (TYPECPX = {double re, im},
 *    dsp/datatypes.h:24-41) exactly like the reference methods, may run in place, return the
 *    number of samples produced (>= 0) or a negative CSDR_E* code (a new failure class the
 *    reference does not have: HIP errors, bad handles);
 *  - "batch" entry points are the multi-channel extension: device-resident interleaved fp32
 *    I/Q, channel-major [channels][stride], no host copies, launched on the caller's stream;
 *  - the library never falls back to a CPU path: without a usable GPU every create() fails;
 *  - batch process calls only enqueue work on `stream` (NULL = the default stream) and return; the
 *    caller synchronises.  Setters, getters and the host entry points synchronise the device;
 *  - threading: a handle is not locked internally -- ONE thread at a time per handle, setters and getters
 *    included.  A setter uploads parameters or state with device copies and a getter may read and reset
 *    device state (csdr_*_get_smeter_peak); issued while a process call of the same handle is running on
 *    another thread they would race with it.  The drop-in C++ classes (cutesdr_amd/dropin/dsp/) provide the
 *    guarantee the reference's QMutex members give its host: every method that reaches this ABI holds the
 *    object's lock, so GUI-thread setters and the IQ thread's ProcessData serialise per object.  Different
 *    handles are independent and may be used from different threads at the same time.
 */
#ifndef CUTESDR_MI_H
#define CUTESDR_MI_H

#ifdef __cplusplus
extern "C" {
#endif

#define CSDR_OK 0
#define CSDR_EINVAL (-1)      /* bad argument / handle */
#define CSDR_EHIP (-2)        /* HIP runtime error, see csdr_last_error() */
#define CSDR_ENOMEM (-3)
#define CSDR_ESTATE (-4)      /* call order / configuration error */

int csdr_version(void);
const char *csdr_last_error(void);          /* thread-local, never NULL */
int csdr_device_count(void);                /* 0 when no GPU is visible */

/* device memory helpers for callers without their own allocator (tests, C hosts) */
void *csdr_dev_alloc(int device, unsigned long long bytes);
int csdr_dev_free(int device, void *p);
int csdr_dev_upload(int device, void *dst, const void *src, unsigned long long bytes);
int csdr_dev_download(int device, void *dst, const void *src, unsigned long long bytes);
int csdr_dev_sync(int device);

/* ----------------------------------------------------------------------------------------
 * CFastFIR -- overlap-save FFT band-pass (dsp/fastfir.h:17-44, dsp/fastfir.cpp).
 * fft_size in {2048, 4096, 8192, 16384}; taps = fft_size/2+1, hop = fft_size/2
 * (2048 is the reference's compiled-in size, fastfir.cpp:55-56).
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_fastfir csdr_fastfir;

/* CFastFIR::CFastFIR (fastfir.cpp:67-130) */
csdr_fastfir *csdr_fastfir_create(int device, int fft_size);
void csdr_fastfir_destroy(csdr_fastfir *f);
/* CFastFIR::SetupParameters (fastfir.cpp:178-259).  Returns 1 = new filter designed,
 * 0 = unchanged, CSDR_EINVAL = rejected by the reference's sanity check (old taps kept). */
int csdr_fastfir_setup(csdr_fastfir *f, double flo, double fhi, double offset, double fs);
/* CFastFIR::ProcessData (fastfir.cpp:268-306).  in/out: n interleaved double pairs; out needs
 * room for n + fft_size/2 samples; in == out allowed.  Returns samples written. */
int csdr_fastfir_process(csdr_fastfir *f, int n, const double *in_iq, double *out_iq);

/* batched, device-resident form of the same filter */
typedef struct csdr_fastfir_batch csdr_fastfir_batch;
csdr_fastfir_batch *csdr_fastfir_batch_create(int device, int channels, int fft_size);
void csdr_fastfir_batch_destroy(csdr_fastfir_batch *b);
/* channel = -1: one filter shared by every channel; otherwise that channel's own filter */
int csdr_fastfir_batch_setup(csdr_fastfir_batch *b, int channel, double flo, double fhi,
                             double offset, double fs);
/* clears the overlap history (as a freshly constructed CFastFIR) */
int csdr_fastfir_batch_reset(csdr_fastfir_batch *b);
/* d_in/d_out: device pointers, interleaved fp32 I/Q, [channels][stride] (strides in complex
 * samples).  n_per_channel must be a positive multiple of fft_size/2; produces exactly
 * n_per_channel samples per channel (first call: the stream delayed behind fft_size/2 zeros,
 * as the reference's zero-initialised overlap does).  stream: hipStream_t or NULL.
 * blocks_per_wg <= 0 picks a default.  Asynchronous; returns CSDR_OK or an error. */
int csdr_fastfir_batch_process(csdr_fastfir_batch *b, const float *d_in, long long in_stride,
                               int n_per_channel, float *d_out, long long out_stride,
                               void *stream, int blocks_per_wg);
/* copy of the designed frequency response H[k] (natural order, fp64 pairs) for tests */
int csdr_fastfir_batch_get_response(csdr_fastfir_batch *b, int channel, double *h_out);

/* ----------------------------------------------------------------------------------------
 * CDownConvert -- NCO mixer + decimate-by-2^n cascade (dsp/downconvert.h:24-120).
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_downconvert csdr_downconvert;

/* CDownConvert::CDownConvert (downconvert.cpp:60-73): 100 kHz in, no stages, phasor 1+0j */
csdr_downconvert *csdr_downconvert_create(int device);
void csdr_downconvert_destroy(csdr_downconvert *d);
/* CDownConvert::SetCwOffset (downconvert.h:30) */
int csdr_downconvert_set_cw_offset(csdr_downconvert *d, double offset);
/* CDownConvert::SetFrequency (downconvert.cpp:98-107); stores freq + CW offset */
int csdr_downconvert_set_frequency(csdr_downconvert *d, double freq);
/* CDownConvert::SetDataRate (downconvert.cpp:114-173): rebuilds the stage list when
 * (in_rate, max_bw) change, re-applies the stored frequency, returns the output rate */
double csdr_downconvert_set_data_rate(csdr_downconvert *d, double in_rate, double max_bw);
/* CDownConvert::ProcessData (downconvert.cpp:186-263): n interleaved double pairs in,
 * n / 2^stages out (n must be a multiple of 2^stages and even, as in the reference);
 * in == out allowed.  The reference also scribbles intermediate data over pInData; this
 * implementation leaves the input untouched. */
int csdr_downconvert_process(csdr_downconvert *d, int n, const double *in_iq, double *out_iq);
/* introspection for tests: stage codes (3 = CIC3, else half-band length), stored NCO freq */
int csdr_downconvert_get_stages(csdr_downconvert *d, int *codes, int cap);
double csdr_downconvert_get_nco_freq(csdr_downconvert *d);

/* batched device-resident form; channel = -1 addresses every channel */
typedef struct csdr_downconvert_batch csdr_downconvert_batch;
csdr_downconvert_batch *csdr_downconvert_batch_create(int device, int channels);
void csdr_downconvert_batch_destroy(csdr_downconvert_batch *b);
int csdr_downconvert_batch_set_cw_offset(csdr_downconvert_batch *b, int channel, double offset);
int csdr_downconvert_batch_set_frequency(csdr_downconvert_batch *b, int channel, double freq);
double csdr_downconvert_batch_set_data_rate(csdr_downconvert_batch *b, int channel, double in_rate,
                                            double max_bw);
int csdr_downconvert_batch_get_stages(csdr_downconvert_batch *b, int channel, int *codes, int cap);
double csdr_downconvert_batch_get_nco_freq(csdr_downconvert_batch *b, int channel);
/* samples channel `channel` produces for n_in input samples (n_in >> stages) */
int csdr_downconvert_batch_out_count(csdr_downconvert_batch *b, int channel, int n_in);
/* d_in [channels][in_stride] -> d_out [channels][out_stride], interleaved fp32 I/Q on the
 * device; n_per_channel even and a multiple of every channel's 2^stages.  Asynchronous. */
int csdr_downconvert_batch_process(csdr_downconvert_batch *b, const float *d_in, long long in_stride,
                                   int n_per_channel, float *d_out, long long out_stride,
                                   void *stream);

/* ----------------------------------------------------------------------------------------
 * CAgc (dsp/agc.h:19-62), CSMeter (dsp/smeter.h:13-28), CFir (dsp/fir.h:20-43),
 * CIir (dsp/iir.h:17-39) -- the leaf objects the reference also exposes on their own.
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_agc csdr_agc;
csdr_agc *csdr_agc_create(int device);
void csdr_agc_destroy(csdr_agc *a);
/* CAgc::SetParameters (agc.cpp:104-167) */
int csdr_agc_set_parameters(csdr_agc *a, int on, int use_hang, int threshold, int manual_gain,
                            int slope, int decay, double sample_rate);
/* CAgc::ProcessData complex (agc.cpp:174-296) / real (agc.cpp:301-401); in == out allowed */
int csdr_agc_process_cpx(csdr_agc *a, int n, const double *in_iq, double *out_iq);
int csdr_agc_process_real(csdr_agc *a, int n, const double *in, double *out);

typedef struct csdr_smeter csdr_smeter;
csdr_smeter *csdr_smeter_create(int device);
void csdr_smeter_destroy(csdr_smeter *s);
/* CSMeter::ProcessData (smeter.cpp:62-93), GetPeak (:98-103, resets the peak), GetAve (:108-112) */
int csdr_smeter_process(csdr_smeter *s, int n, const double *in_iq, double sample_rate);
double csdr_smeter_get_peak(csdr_smeter *s);
double csdr_smeter_get_ave(csdr_smeter *s);

typedef struct csdr_fir csdr_fir;
csdr_fir *csdr_fir_create(int device);
void csdr_fir_destroy(csdr_fir *f);
/* CFir::InitConstFir (fir.cpp:133-153), InitLPFilter (:173-261), InitHPFilter (:278-367) return
 * the tap count; GenerateHBFilter (:374-407) keeps the delay line */
int csdr_fir_init_const(csdr_fir *f, int ntaps, const double *coef);
int csdr_fir_init_lp(csdr_fir *f, double scale, double astop, double fpass, double fstop, double fs);
int csdr_fir_init_hp(csdr_fir *f, double scale, double astop, double fpass, double fstop, double fs);
int csdr_fir_generate_hb(csdr_fir *f, double freq_offset);
int csdr_fir_get_taps(csdr_fir *f, double *coef, double *icoef, double *qcoef);   /* up to 75 each */
/* CFir::ProcessFilter real (fir.cpp:72-92) / complex (:101-127) */
int csdr_fir_process_real(csdr_fir *f, int n, const double *in, double *out);
int csdr_fir_process_cpx(csdr_fir *f, int n, const double *in_iq, double *out_iq);

typedef struct csdr_iir csdr_iir;
csdr_iir *csdr_iir_create(int device);
void csdr_iir_destroy(csdr_iir *f);
/* CIir::InitLP/HP/BP/BR (iir.cpp:86-165): kind 0 LP, 1 HP, 2 BP, 3 BR */
int csdr_iir_init(csdr_iir *f, int kind, double f0, double q, double sample_rate);
int csdr_iir_get_coefs(csdr_iir *f, double *b0b1b2a1a2);
/* CIir::ProcessFilter (iir.cpp:171-201) */
int csdr_iir_process_real(csdr_iir *f, int n, const double *in, double *out);
int csdr_iir_process_cpx(csdr_iir *f, int n, const double *in_iq, double *out_iq);

/* ----------------------------------------------------------------------------------------
 * Demodulators: CAmDemod (dsp/amdemod.h), CSamDemod (dsp/samdemod.h), CFmDemod
 * (dsp/fmdemod.h), CSsbDemod (dsp/ssbdemod.h).  mono: n complex in -> n real out;
 * stereo: n complex in -> n complex out.
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_amdemod csdr_amdemod;
csdr_amdemod *csdr_amdemod_create(int device, double sample_rate);            /* amdemod.cpp:50-54 */
void csdr_amdemod_destroy(csdr_amdemod *d);
int csdr_amdemod_set_bandwidth(csdr_amdemod *d, double bandwidth);            /* :56-60 */
int csdr_amdemod_process_mono(csdr_amdemod *d, int n, const double *in_iq, double *out);     /* :66-82 */
int csdr_amdemod_process_stereo(csdr_amdemod *d, int n, const double *in_iq, double *out_iq); /* :87-104 */

typedef struct csdr_samdemod csdr_samdemod;
csdr_samdemod *csdr_samdemod_create(int device, double sample_rate);          /* samdemod.cpp:54-73 */
void csdr_samdemod_destroy(csdr_samdemod *d);
int csdr_samdemod_process_mono(csdr_samdemod *d, int n, const double *in_iq, double *out);     /* :78-110 */
int csdr_samdemod_process_stereo(csdr_samdemod *d, int n, const double *in_iq, double *out_iq); /* :115-158 */

typedef struct csdr_fmdemod csdr_fmdemod;
csdr_fmdemod *csdr_fmdemod_create(int device, double sample_rate);            /* fmdemod.cpp:62-89 */
void csdr_fmdemod_destroy(csdr_fmdemod *d);
int csdr_fmdemod_set_squelch(csdr_fmdemod *d, int value);                     /* :95-98 */
int csdr_fmdemod_process_mono(csdr_fmdemod *d, int n, double fm_bw, const double *in_iq, double *out);     /* :157-192 */
int csdr_fmdemod_process_stereo(csdr_fmdemod *d, int n, double fm_bw, const double *in_iq, double *out_iq); /* :197-236 */
int csdr_fmdemod_get_squelched(csdr_fmdemod *d);                              /* m_SquelchState, for tests */

int csdr_ssbdemod_process_mono(int n, const double *in_iq, double *out);      /* ssbdemod.cpp:48-53 */
int csdr_ssbdemod_process_stereo(int n, const double *in_iq, double *out_iq); /* :55-60 */

/* ----------------------------------------------------------------------------------------
 * CDemodulator -- the whole receive chain (dsp/demodulator.h:56-100).
 * -------------------------------------------------------------------------------------- */
/* POD mirror of tDemodInfo (dsp/demodulator.h:35-54) without the QString label */
typedef struct csdr_demod_info {
    int HiCut, HiCutmin, HiCutmax, LowCut, LowCutmin, LowCutmax;
    int FilterClickResolution, Offset, SquelchValue;
    int AgcSlope, AgcThresh, AgcManualGain, AgcDecay;
    int AgcOn, AgcHangOn, Symetric;
} csdr_demod_info;

#define CSDR_DEMOD_AM 0      /* DEMOD_* (dsp/demodulator.h:20-26) */
#define CSDR_DEMOD_SAM 1
#define CSDR_DEMOD_FM 2
#define CSDR_DEMOD_USB 3
#define CSDR_DEMOD_LSB 4
#define CSDR_DEMOD_CWU 5
#define CSDR_DEMOD_CWL 6

typedef struct csdr_demod csdr_demod;
/* CDemodulator::CDemodulator (demodulator.cpp:47-60); fastfir_n 2048 is the reference's filter */
csdr_demod *csdr_demod_create(int device, int fastfir_n);
void csdr_demod_destroy(csdr_demod *d);
int csdr_demod_set_input_rate(csdr_demod *d, double rate);                    /* :92-99 */
int csdr_demod_set_demod(csdr_demod *d, int mode, const csdr_demod_info *info); /* :107-157 */
int csdr_demod_set_freq(csdr_demod *d, double freq);                          /* demodulator.h:68-69 */
double csdr_demod_get_output_rate(csdr_demod *d);
double csdr_demod_get_smeter_peak(csdr_demod *d);
double csdr_demod_get_smeter_ave(csdr_demod *d);
int csdr_demod_get_buf_limit(csdr_demod *d);                                  /* m_InBufLimit */
/* CDemodulator::ProcessData mono (:163-215) / stereo (:221-273): buffers n samples, runs the
 * chain every m_InBufLimit samples, every pass writes at out[0], returns the summed count */
int csdr_demod_process_mono(csdr_demod *d, int n, const double *in_iq, double *out);
int csdr_demod_process_stereo(csdr_demod *d, int n, const double *in_iq, double *out_iq);
/* same chain, passes append instead of overwriting (batch harness form) */
int csdr_demod_process_mono_append(csdr_demod *d, int n, const double *in_iq, double *out);
/* Deferred output (opt-in; no counterpart in the reference, whose ProcessData computes on the caller's thread).  With it
 * on, a pass that is due to return samples (dsp/demodulator.cpp:169-214, every m_InBufLimit samples) returns the samples
 * of the PREVIOUS such pass -- finished while the caller was handing over this window -- and leaves its own on the way:
 * the call never waits for the device, the chain's pass and the caller's next m_InBufLimit samples overlap.  The same
 * samples in the same order, one window later (10 ms at 2 MSPS); the first pass returns 0.  csdr_demod_flush hands over
 * the last pass (returns its count: mono reals, stereo complex pairs; cap in doubles) -- before a switch back, a mode
 * change between mono and stereo calls, or the end of the stream.  Not together with the stage taps. */
int csdr_demod_set_deferred(csdr_demod *d, int on);
int csdr_demod_flush(csdr_demod *d, double *out, int cap);
/* Stage taps of the chain: what the reference hands to g_pTestBench->DisplayData(n, buf, m_OutputRate, PROFILE_k) in
 * every pass (dsp/demodulator.cpp:175,180,187,208; gui/testbench.h:29-38) -- PROFILE_1 the down-converter's output,
 * PROFILE_2 the band-pass filter's, PROFILE_3 the AGC's (n complex samples each, n = 0 while the filter is filling),
 * PROFILE_4 the audio (mono: n reals, stereo: n complex).  mask bit k-1 switches PROFILE_k on; 0 = off (the default:
 * nothing of this costs anything then).  With taps on every pass waits for its results, the AGC's output goes through
 * device memory (S-meter + AGC and the demodulator as two launches; same words out), and per pass and profile either
 * `fn` is called -- from inside the ProcessData call, with `user` -- or, fn == NULL, the samples are appended to a
 * per-profile host buffer that csdr_demod_get_tap returns and empties (returns the number of doubles written; complex
 * samples are two). */
typedef void (*csdr_tap_fn)(void *user, int profile, int n, const double *data, int is_complex, double rate);
int csdr_demod_set_taps(csdr_demod *d, int mask, csdr_tap_fn fn, void *user);
int csdr_demod_get_tap(csdr_demod *d, int profile, double *out, int cap);

/* batched device-resident chain: configure every channel, commit once, then process */
typedef struct csdr_demod_batch csdr_demod_batch;
csdr_demod_batch *csdr_demod_batch_create(int device, int channels, int fastfir_n);
void csdr_demod_batch_destroy(csdr_demod_batch *b);
/* CDemodulator::SetInputSampleRate (dsp/demodulator.cpp:92-99) of every receiver.  Before commit(): recorded.  After
 * commit(): applied at once, at any time, as the host does on every bandwidth switch of the radio
 * (interface/sdrinterface.cpp:753-754): every down-converter is rebuilt for the new rate (new stage list from zeroed
 * histories; oscillator phase and amplitude kept; the CW offset added once more, downconvert.cpp:169), the output rate
 * and the S-meter's time constants follow, and -- like the reference -- filter taps and overlap, AGC constants and rings
 * and the demodulator objects stay as they are until the receiver's next SetDemod, which for an unchanged mode keeps the
 * demodulator built for the old output rate.  Receivers keep their rows while a plan group still shares one decimation;
 * one whose new chain decimates differently from its group's moves as after a mode change.  Synchronises the device. */
int csdr_demod_batch_set_input_rate(csdr_demod_batch *b, double rate);
/* CDemodulator::SetDemod (dsp/demodulator.cpp:107-157) of one receiver.  Before commit(): recorded.  After commit():
 * applied at once, like the reference between two ProcessData calls -- also a mode whose maximum bandwidth (hence
 * decimator chain and output rate) differs from the present one: the receiver then continues in a plan group of its
 * own with everything the reference keeps across SetDemod (oscillator, filter overlap and partly filled filter input,
 * AGC, S-meter; new demodulator, decimator from zero histories); nothing else of the batch is disturbed.  The call
 * synchronises the device. */
int csdr_demod_batch_set_demod(csdr_demod_batch *b, int channel, int mode, const csdr_demod_info *info);
int csdr_demod_batch_commit(csdr_demod_batch *b);
/* Receivers cut from shared streams (one radio, many receivers: SURVEY 8e "if channels are cut from one stream"):
 * receiver c reads row input_row[c] of d_in (or datagram stream input_row[c]) instead of row c; several receivers
 * may name the same row, each keeps its own CDownConvert / CFastFIR / AGC / demodulator state as in the reference,
 * where every CDemodulator is handed the same buffer (interface/sdrinterface.cpp:903).  input_row: `channels` ints on
 * the host, each in [0, channels); NULL restores row c for receiver c.  Before or after commit(); the call
 * synchronises the device. */
int csdr_demod_batch_set_input_rows(csdr_demod_batch *b, const int *input_row);
int csdr_demod_batch_set_freq(csdr_demod_batch *b, int channel, double freq);
double csdr_demod_batch_get_output_rate(csdr_demod_batch *b, int channel);
double csdr_demod_batch_get_smeter_ave(csdr_demod_batch *b, int channel);
/* CSMeter::GetAve / GetPeak (smeter.cpp:98-112) of every channel at once into device arrays [channels]
 * (either may be NULL); reading the peaks resets them, as GetPeak does.  Asynchronous on `stream`: ordered
 * behind the process calls issued on it.  What a multi-GPU host gathers from its ranks. */
int csdr_demod_batch_get_smeter_all(csdr_demod_batch *b, float *d_ave, float *d_peak, void *stream);
/* d_in [channels][in_stride] complex fp32 -> d_out [channels][out_stride] fp32 mono audio.
 * Numerical contract of the recurrent stages: the scans that replace the reference's per-sample loops (averagers,
 * AGC, locked PLL tiles) differ from the sequential fp64 loop by rounding only, and an FM tile whose loop is NOT
 * locked is walked by all threads at once, each from a warmed-up state that is CHECKED to meet its predecessor's to
 * 1e-9 turns -- a bound, not bit equality: the audio of such tiles (noise, acquisition) can differ from a one-thread
 * walk, and between builds with another tiling, by up to ~1e-6 of full scale (tests: 1e-6; CSDR_PLL_OVERLAP=0 selects
 * the one-thread walk).  Strict and pipelined mode run the same kernels with the same tiling: identical words.
 * The audio WORDS do not depend on how a stream is cut into calls either, as long as every call is a whole number of
 * 512-sample tiles (the down-converter re-anchors its oscillator on a grid counted from the receiver's first sample,
 * the filter walks whole hops, the post-chain whole bursts): one call of 24 windows = 24 calls of one, bit for bit. */
int csdr_demod_batch_process(csdr_demod_batch *b, const float *d_in, long long in_stride,
                             int n_per_channel, float *d_out, long long out_stride, void *stream);
/* Pipelined mode for streaming hosts (off by default).  on != 0: a process call only enqueues, and a call's post-chains
 * (S-meter, AGC, demodulator) run on internal streams beside the NEXT call's down-converters.  In the caller's stream
 * order, after process call k+1 the INPUT buffer of call k has been consumed and the OUTPUT rows of call k-1 are complete
 * (the default form completes those of call k as well); after csdr_demod_batch_flush everything issued so far is
 * complete.  Same results as the strict mode, word for word.
 * on = 1 / 2: the chained form (round 6) -- the strict mode's schedule (one down-converter at a time, every group's
 * filter in queue order behind its down-converter) carried across calls, two streams per plan group.  on = 3 (or
 * CSDR_PIPE_KIND=3 in the environment): the three-stage form of rounds 3-5 -- every group's down-converter at once,
 * filter and post-chain on two more streams per group over double buffers; it wants more than HIP's default four
 * hardware queues (GPU_MAX_HW_QUEUES=8 or more) and is the slower one (1.75-1.80 against 1.65-1.68 ms per call on 256
 * mixed receivers; the strict mode: 1.60-1.65). */
int csdr_demod_batch_set_pipelined(csdr_demod_batch *b, int on);
int csdr_demod_batch_flush(csdr_demod_batch *b, void *stream);
/* the stereo overload (dsp/demodulator.cpp:221-273: AM/FM duplicate the audio into both halves, SAM splits the
 * sidebands, SSB/CW copy the filtered I/Q) for every channel: d_out_iq [channels][out_stride] complex fp32,
 * out_stride in complex samples */
int csdr_demod_batch_process_stereo(csdr_demod_batch *b, const float *d_in, long long in_stride,
                                    int n_per_channel, float *d_out_iq, long long out_stride, void *stream);
int csdr_demod_batch_out_count(csdr_demod_batch *b, int channel);
/* The same stage taps for the receivers of a batch (strict mode only).  With a non-zero mask every group keeps the
 * LAST call's down-converter output, filter output and AGC output in device memory (the AGC's through a split launch,
 * as above); csdr_demod_batch_get_tap copies receiver `channel`'s samples of that call to host memory -- PROFILE_1 ..
 * 3: interleaved fp32 I/Q, PROFILE_1 counts the samples the down-converter appended in that call, 2 and 3 the samples
 * the filter released -- and returns the number of floats (synchronous; PROFILE_4 is the caller's own output row). */
int csdr_demod_batch_set_taps(csdr_demod_batch *b, int mask);
int csdr_demod_batch_get_tap(csdr_demod_batch *b, int channel, int profile, float *out, int cap);
/* Diagnostics: the number of plan groups the batch runs per call (rows that share one decimation; every group is one
 * set of launches), and with rows != NULL the number of rows -- live and muted -- of all groups.  A mode change to a
 * chain of the same decimation stays in its row; one to another decimation moves the receiver into a muted row of a
 * matching group when there is one, else into a group of its own; a group whose rows are all muted is dropped. */
int csdr_demod_batch_group_count(csdr_demod_batch *b, int *rows);
/* ---- the same chain over SEVERAL devices from one host object -------------------------------------------
 * Receivers are independent: shard s owns the contiguous channel range [s C / N, (s+1) C / N) on device devices[s], with
 * all its state resident there; a process call is N asynchronous batch calls, no collective on the data path.  Global
 * channel ids everywhere; set_demod / set_freq are routed to the owning shard.  What crosses devices is what the
 * survey names: the S-meters gathered to the host, and -- for receivers cut from ONE radio's stream, as
 * CSdrInterface hands every CDemodulator the same buffer (interface/sdrinterface.cpp:903) -- the broadcast of that
 * block to every other shard's device (hipMemcpyPeerAsync over xGMI, one copy per device on the shard's stream).
 * Several shards may name the same device.  A host with one PROCESS per GPU (bench.py --gpus N, torch.distributed /
 * RCCL) uses csdr_demod_batch per rank instead. */
typedef struct csdr_demod_shard csdr_demod_shard;
csdr_demod_shard *csdr_demod_shard_create(const int *devices, int nshards, int channels, int fastfir_n);
void csdr_demod_shard_destroy(csdr_demod_shard *s);
int csdr_demod_shard_count(csdr_demod_shard *s);
int csdr_demod_shard_range(csdr_demod_shard *s, int shard, int *first, int *count, int *device);
int csdr_demod_shard_set_input_rate(csdr_demod_shard *s, double rate);   /* every shard's batch; before or after commit */
int csdr_demod_shard_set_demod(csdr_demod_shard *s, int channel, int mode, const csdr_demod_info *info);
int csdr_demod_shard_commit(csdr_demod_shard *s);
int csdr_demod_shard_set_freq(csdr_demod_shard *s, int channel, double freq);
double csdr_demod_shard_get_output_rate(csdr_demod_shard *s, int channel);
int csdr_demod_shard_set_pipelined(csdr_demod_shard *s, int on);
/* d_in[k] / d_out[k]: shard k's rows [count_k][stride] on ITS device; streams[k] or NULL (the object's own streams) */
int csdr_demod_shard_process(csdr_demod_shard *s, const float *const *d_in, long long in_stride, int n_per_channel,
                             float *const *d_out, long long out_stride, void *const *streams);
int csdr_demod_shard_out_count(csdr_demod_shard *s, int channel);
/* shared-stream mode: receiver c reads row input_row[c] (< nrows <= receivers per shard) of a wide-band block that
 * process_shared takes ONCE, resident on src_device behind src_stream's work, and copies to the other devices */
int csdr_demod_shard_set_input_rows(csdr_demod_shard *s, const int *input_row, int nrows);
int csdr_demod_shard_process_shared(csdr_demod_shard *s, const float *d_block, int src_device, void *src_stream,
                                    long long in_stride, int n_per_channel, float *const *d_out, long long out_stride);
/* The datagram and blanker forms of the one-device batch on every shard.  set_blanker = CNoiseProc::SetupBlanker for all
 * receivers (noiseproc.cpp:78-119; the shard object owns one csdr_noiseproc_batch per device); process_packets takes
 * d_packets[s] = shard s's receivers' datagrams on ITS device ([count_s][npackets][pkt_len] bytes, as
 * csdr_demod_batch_process_packets) and runs the blanker fused in front when one is set; process_blanked is
 * csdr_demod_shard_process with that blanker in front of fp32 rows. */
int csdr_demod_shard_set_blanker(csdr_demod_shard *s, int on, double threshold, double width_us, double sample_rate);
int csdr_demod_shard_process_packets(csdr_demod_shard *s, const void *const *d_packets, int npackets, int pkt_len,
                                     float *const *d_out, long long out_stride, void *const *streams);
int csdr_demod_shard_process_blanked(csdr_demod_shard *s, const float *const *d_in, long long in_stride, int n_per_channel,
                                     float *const *d_out, long long out_stride, void *const *streams);
/* waits for everything issued on the shards' own streams (pipelined mode: flushes first) */
int csdr_demod_shard_sync(csdr_demod_shard *s);
/* CSMeter::GetAve / GetPeak of every receiver into HOST arrays indexed by global channel (either may be NULL; reading
 * the peaks resets them): the gather of SURVEY 8e.  Synchronous. */
int csdr_demod_shard_get_smeter_all(csdr_demod_shard *s, float *h_ave, float *h_peak);

/* The same pass fed with the datagrams as they arrived (interface/netiobase.cpp:479-527):
 * d_packets [channels][npackets][pkt_len] bytes on the device, 4-byte aligned, pkt_len 1028 (16 bit) or 1444
 * (24 bit).  No unpack pass: the first kernel at input rate decodes the datagrams in its own loads -- the
 * down-converter, or the blanker when nb is given (what CSdrInterface::ProcessIQData runs in place before the
 * chain, sdrinterface.cpp:884; nb may be NULL, it must have as many channels).  npackets * samples-per-packet
 * must be a multiple of the largest decimation in use.  Forward declaration: csdr_noiseproc_batch is defined
 * further down. */
struct csdr_noiseproc_batch;
int csdr_demod_batch_process_packets(csdr_demod_batch *b, const void *d_packets, int npackets, int pkt_len,
                                     struct csdr_noiseproc_batch *nb, float *d_out, long long out_stride,
                                     void *stream);

/* csdr_demod_batch_process with CNoiseProc's blanker in front (interface/sdrinterface.cpp:884 runs it in place before
 * the chain), fused: the blanker pass leaves one bit per sample and the down-converter zeroes the delayed sample under
 * it in its own load -- no blanked copy of d_in exists.  nb: as many channels as b; every receiver reads its own row
 * (not with csdr_demod_batch_set_input_rows).  The two-pass equivalent: csdr_noiseproc_batch_process into a buffer of
 * the caller's, then csdr_demod_batch_process on that. */
int csdr_demod_batch_process_blanked(csdr_demod_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                                     struct csdr_noiseproc_batch *nb, float *d_out, long long out_stride, void *stream);

/* ----------------------------------------------------------------------------------------
 * CFft -- display spectrum and plain transforms (dsp/fft.h:24-85).  FFT sizes 512..65536 as in
 * the reference (clamped to that range, fft.cpp:140-145); a size that is not a power of two
 * returns CSDR_EINVAL.
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_fft csdr_fft;
csdr_fft *csdr_fft_create(int device);                                       /* fft.cpp:41-62 */
void csdr_fft_destroy(csdr_fft *f);
int csdr_fft_set_params(csdr_fft *f, int size, int invert, double db_comp, double fs);  /* :118-243 */
int csdr_fft_set_ave(csdr_fft *f, int ave);                                  /* :103-113 */
int csdr_fft_reset(csdr_fft *f);                                             /* :248-259 */
/* CFft::PutInDisplayFFT (:267-288): n complex doubles (n = FFT size); returns m_TotalCount */
int csdr_fft_put_display(csdr_fft *f, int n, const double *in_iq);
/* CFft::GetScreenIntegerFFTData (:308-410): fills out[0..max_w), returns the overload flag */
int csdr_fft_get_screen(csdr_fft *f, int max_h, int max_w, double max_db, double min_db,
                        int start_hz, int stop_hz, int *out);
/* m_pFFTAveBuf (bels, display order, `size` floats) for tests; returns the size */
int csdr_fft_get_ave(csdr_fft *f, float *out);
/* CFft::FwdFFT / RevFFT (:416-426): in-place, unnormalised, FwdFFT has the POSITIVE exponent */
int csdr_fft_fwd(csdr_fft *f, double *inout_iq);
int csdr_fft_rev(csdr_fft *f, double *inout_iq);

/* batched spectra: [channels][in_stride] complex fp32 on the device, nframes frames per row */
typedef struct csdr_fft_batch csdr_fft_batch;
csdr_fft_batch *csdr_fft_batch_create(int device, int channels);
void csdr_fft_batch_destroy(csdr_fft_batch *f);
int csdr_fft_batch_set_params(csdr_fft_batch *f, int size, int invert, double db_comp, double fs);
int csdr_fft_batch_set_ave(csdr_fft_batch *f, int ave);
int csdr_fft_batch_reset(csdr_fft_batch *f);
int csdr_fft_batch_size(csdr_fft_batch *f);
int csdr_fft_batch_put_display(csdr_fft_batch *f, const float *d_in, long long in_stride, int nframes,
                               void *stream);
int csdr_fft_batch_get_ave(csdr_fft_batch *f, int channel, float *out);
int csdr_fft_batch_get_total_count(csdr_fft_batch *f, int channel);
int csdr_fft_batch_get_screen(csdr_fft_batch *f, int channel, int max_h, int max_w, double max_db,
                              double min_db, int start_hz, int stop_hz, int *out);
/* the same mapping for every channel at once, on the device: d_out is
 * [channels][out_stride] int32 on the device, out_stride >= max_w; pixels that no bin maps to keep
 * what d_out held (the reference leaves them untouched too).  d_overload: optional [channels] int32,
 * the overload flag of each channel's last PutInDisplayFFT.  Asynchronous on `stream`. */
int csdr_fft_batch_get_screen_all(csdr_fft_batch *f, int max_h, int max_w, double max_db, double min_db,
                                  int start_hz, int stop_hz, int *d_out, long long out_stride, int *d_overload,
                                  void *stream);
/* CPlotter's waterfall palette (gui/plotter.cpp:67-83): 256 colours, 0xFFRRGGBB each (QRgb order) */
void csdr_plotter_color_table(unsigned int *out256);
/* The new top line of the waterfall of every channel (CPlotter::draw, gui/plotter.cpp:425-441):
 * GetScreenIntegerFFTData(255, max_w, ...) and, per pixel, the palette entry 255 - y.  d_rgb is
 * [channels][out_stride] 0xFFRRGGBB on the device, out_stride >= max_w; pixels that no bin maps to keep what
 * d_rgb held (the reference paints them from an uninitialised buffer).  d_overload as above.  Asynchronous. */
int csdr_fft_batch_get_waterfall_all(csdr_fft_batch *f, int max_w, double max_db, double min_db, int start_hz,
                                     int stop_hz, unsigned int *d_rgb, long long out_stride, int *d_overload,
                                     void *stream);

/* ----------------------------------------------------------------------------------------
 * CFractResampler (dsp/fractresampler.h:17-33)
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_resampler csdr_resampler;
csdr_resampler *csdr_resampler_create(int device);
void csdr_resampler_destroy(csdr_resampler *r);
int csdr_resampler_init(csdr_resampler *r, int max_input_size);              /* fractresampler.cpp:85-135 */
/* Resample overloads: rate = input rate / output rate; return the output sample count */
int csdr_resampler_resample_real(csdr_resampler *r, int n, double rate, const double *in, double *out);       /* :258-297 */
int csdr_resampler_resample_cpx(csdr_resampler *r, int n, double rate, const double *in_iq, double *out_iq);  /* :144-184 */
int csdr_resampler_resample_real_i16(csdr_resampler *r, int n, double rate, const double *in, short *out,
                                     double gain);                           /* :306-352 */
int csdr_resampler_resample_cpx_i16(csdr_resampler *r, int n, double rate, const double *in_iq, short *out_lr,
                                    double gain);                            /* :194-249 */

/* device-resident batch form for the chain's mono audio rows: [channels][stride] fp32 in; fp32 or
 * int16 (scaled by gain, clipped, truncated: fractresampler.cpp:306-352) out.  Every channel runs
 * on the same clock: one rate and one time accumulator (m_FloatTime) for the batch.  Returns the
 * output samples per channel; asynchronous on `stream`. */
typedef struct csdr_resampler_batch csdr_resampler_batch;
csdr_resampler_batch *csdr_resampler_batch_create(int device, int channels);
void csdr_resampler_batch_destroy(csdr_resampler_batch *b);
int csdr_resampler_batch_resample(csdr_resampler_batch *b, const float *d_in, long long in_stride, int n, double rate,
                                  float *d_out_f32, short *d_out_i16, long long out_stride, double gain, void *stream);

/* ----------------------------------------------------------------------------------------
 * Sound-sink adaptation: the queue and rate-error loop of CSoundOut
 * (interface/soundout.cpp, both modes) around the device resampler -- the step right after the path in
 * a live receiver.  put = PutOutQueue (:196-305): resample to 48 kHz int16 with Rate = m_OutRatio *
 * (1 + m_RateCorrection) and the volume gain, queue (16384 entries; overflow drops a quarter); get = GetOutQueue
 * (:311-445) for the audio thread (silence until half full, underflow backs up a quarter); every second of
 * consumed samples the P-controller CalcError (:456-468) sets the correction from the average fill level.
 * One producer thread and one consumer thread as in the reference: the caller serialises put and get per handle.
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_soundsink csdr_soundsink;
csdr_soundsink *csdr_soundsink_create(int device, int stereo);               /* soundout.cpp:60-76 */
void csdr_soundsink_destroy(csdr_soundsink *s);
int csdr_soundsink_change_user_data_rate(csdr_soundsink *s, double rate);    /* :155-175 */
int csdr_soundsink_set_blocking(csdr_soundsink *s, int on);                  /* Start(..., BlockingMode) :86-90: put waits while the queue is full (:209-220), get skips the rate loop (:354-358) */
int csdr_soundsink_set_volume(csdr_soundsink *s, int vol);                   /* :180-189 */
/* in: n doubles (mono sink) or n interleaved double pairs (stereo sink), at most 8192 samples; returns the
 * resampled samples produced */
int csdr_soundsink_put(csdr_soundsink *s, int n, const double *in);
/* out: n shorts, or n L/R pairs of a stereo sink; returns n */
int csdr_soundsink_get(csdr_soundsink *s, int n, short *out);
double csdr_soundsink_get_rate_correction(csdr_soundsink *s);                /* m_RateCorrection */
double csdr_soundsink_get_ave_level(csdr_soundsink *s);                      /* m_AveOutQLevel */
int csdr_soundsink_get_level(csdr_soundsink *s);                             /* m_OutQLevel */
int csdr_soundsink_get_ppm_error(csdr_soundsink *s);                         /* m_PpmError */

/* ----------------------------------------------------------------------------------------
 * Input-rate stages in front of the down-converter
 *
 * CNoiseProc (dsp/noiseproc.h:23-58): impulse blanker.  setup = SetupBlanker (noiseproc.cpp:78-119,
 * including its quirk that a change of the sample rate alone is ignored); process =
 * ProcessBlanker (:121-176).  Returns CSDR_EINVAL for sample rates whose 5 ms magnitude window
 * does not fit the reference's 32768-entry buffer (the reference overruns it).
 * -------------------------------------------------------------------------------------- */
typedef struct csdr_noiseproc csdr_noiseproc;
csdr_noiseproc *csdr_noiseproc_create(int device);                           /* noiseproc.cpp:59-65 */
void csdr_noiseproc_destroy(csdr_noiseproc *p);
int csdr_noiseproc_setup(csdr_noiseproc *p, int on, double threshold, double width_us, double sample_rate);
/* n complex doubles in, n out (in == out allowed, as the host calls it: sdrinterface.cpp:884) */
int csdr_noiseproc_process(csdr_noiseproc *p, int n, const double *in_iq, double *out_iq);

/* device-resident batch form: [channels][stride] complex fp32; d_out must not alias d_in */
typedef struct csdr_noiseproc_batch csdr_noiseproc_batch;
csdr_noiseproc_batch *csdr_noiseproc_batch_create(int device, int channels);
void csdr_noiseproc_batch_destroy(csdr_noiseproc_batch *b);
/* channel < 0: every channel */
int csdr_noiseproc_batch_setup(csdr_noiseproc_batch *b, int channel, int on, double threshold, double width_us,
                               double sample_rate);
int csdr_noiseproc_batch_process(csdr_noiseproc_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                                 float *d_out, long long out_stride, void *stream);

/* IQ wire format (interface/netiobase.cpp:479-527): whole UDP datagrams, 4 header bytes then
 * little-endian I,Q pairs; pkt_len 1028 = 256 samples of 16 bit, 1444 = 240 samples of 24 bit
 * (scaled onto the 16-bit range).  d_packets: [channels][npackets][pkt_len] bytes on the device;
 * d_out: [channels][out_stride] complex fp32; d_dc: optional [channels][2] doubles (I, Q offsets,
 * subtracted as CSdrInterface::ProcessIQData does for the display path, sdrinterface.cpp:889-894).
 * d_packets 4-byte aligned, d_out 16-byte aligned, out_stride even.
 * Returns the complex samples written per channel. */
int csdr_ingest_unpack(int device, const void *d_packets, int channels, int npackets, int pkt_len, float *d_out,
                       long long out_stride, const double *d_dc, void *stream);
/* host form of the same conversion: packets in host memory, complex doubles out */
int csdr_ingest_unpack_host(int device, const void *packets, int npackets, int pkt_len, double *out_iq);
/* CSdrInterface::NcoSpurCalibrate (sdrinterface.cpp:829-848): d_dc[ch][0..1] <- running I/Q means
 * (alpha 1e-5) advanced over n more samples of each row of d_iq */
int csdr_ingest_spurcal(int device, const float *d_iq, long long in_stride, int channels, int n, double *d_dc,
                        void *stream);
int csdr_ingest_spurcal_host(int device, int n, const double *in_iq, double *dc_iq);

#ifdef __cplusplus
}
#endif
#endif
