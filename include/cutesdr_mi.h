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
 *  - the library never falls back to a CPU path: without a usable GPU every create() fails.
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

#ifdef __cplusplus
}
#endif
#endif
