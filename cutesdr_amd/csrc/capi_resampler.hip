// capi_resampler.hip -- C ABI for CFractResampler (dsp/fractresampler.h:17-33).
#include "capi_common.hpp"
#include "resampler_kernels.h"
#include "host_math.hpp"
#include <vector>

using namespace csdr;

struct csdr_resampler {
    int device;
    double t;                            // m_FloatTime
    float *d_sinc, *d_buf, *d_out; double *d_times;
    size_t cap_in, cap_out;
    std::vector<float> st;
    std::vector<double> times;
    std::vector<short> st16;
};

static int rs_run(csdr_resampler *r, int n, double rate, bool cpx, const double *in, double *out_f, short *out_i,
                  double gain)
{
    if (!r || n < 0 || (n && !in) || rate <= 0) return fail(CSDR_EINVAL, "bad argument");
    if (!r->d_sinc) return fail(CSDR_ESTATE, "Init() first");
    if ((size_t)n + RS_PERIODS > r->cap_in) return fail(CSDR_EINVAL, "more input than Init(MaxInputSize) allows");
    if (!device_ok(r->device)) return CSDR_EHIP;
    // output times: the reference's sequential fp64 accumulation (fractresampler.cpp:157-178)
    r->times.clear();
    int it = (int)r->t;
    while (it < n) { r->times.push_back(r->t); r->t += rate; it = (int)r->t; }
    r->t -= (double)n;
    const int nout = (int)r->times.size();
    if ((size_t)nout > r->cap_out) {
        if (r->d_out) (void)hipFree(r->d_out);
        if (r->d_times) (void)hipFree(r->d_times);
        r->d_out = nullptr; r->d_times = nullptr; r->cap_out = 0;
        CSDR_HIP(hipMalloc((void **)&r->d_out, (size_t)nout * 8));
        CSDR_HIP(hipMalloc((void **)&r->d_times, (size_t)nout * 8));
        r->cap_out = nout;
    }
    r->st.resize(2 * (size_t)n);
    if (cpx) for (size_t i = 0; i < 2 * (size_t)n; i++) r->st[i] = (float)in[i];
    else for (size_t i = 0; i < (size_t)n; i++) { r->st[2 * i] = (float)in[i]; r->st[2 * i + 1] = 0.f; }
    if (n) CSDR_HIP(hipMemcpy(r->d_buf + 2 * RS_PERIODS, r->st.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    if (nout) CSDR_HIP(hipMemcpy(r->d_times, r->times.data(), (size_t)nout * 8, hipMemcpyHostToDevice));
    ResampleArgs a;
    a.buf = r->d_buf; a.buf_rw = r->d_buf; a.sinc = r->d_sinc; a.times = r->d_times;
    a.out_f32 = out_i ? nullptr : r->d_out; a.out_i16 = out_i ? (short *)r->d_out : nullptr;
    a.gain = (float)gain; a.nout = nout; a.cpx = cpx ? 1 : 0;
    CSDR_HIP(resample_launch(a, n, nullptr));
    if (nout) {
        const size_t ne = cpx ? 2 * (size_t)nout : (size_t)nout;
        if (out_i) {
            CSDR_HIP(hipMemcpy(out_i, r->d_out, ne * 2, hipMemcpyDeviceToHost));
        } else {
            r->st.resize(ne);
            CSDR_HIP(hipMemcpy(r->st.data(), r->d_out, ne * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ne; i++) out_f[i] = (double)r->st[i];
        }
    } else {
        CSDR_HIP(hipDeviceSynchronize());
    }
    return nout;
}

extern "C" {

csdr_resampler *csdr_resampler_create(int device)
{
    if (!device_ok(device)) return nullptr;
    csdr_resampler *r = new csdr_resampler();
    r->device = device; r->t = 0.0;
    r->d_sinc = r->d_buf = r->d_out = nullptr; r->d_times = nullptr; r->cap_in = r->cap_out = 0;
    return r;
}
void csdr_resampler_destroy(csdr_resampler *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->d_sinc) (void)hipFree(r->d_sinc);
    if (r->d_buf) (void)hipFree(r->d_buf);
    if (r->d_out) (void)hipFree(r->d_out);
    if (r->d_times) (void)hipFree(r->d_times);
    delete r;
}
/* CFractResampler::Init (fractresampler.cpp:85-135) */
int csdr_resampler_init(csdr_resampler *r, int max_input_size)
{
    if (!r || max_input_size < 0) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(r->device)) return CSDR_EHIP;
    const size_t cap = (size_t)max_input_size + RS_PERIODS;
    if (!r->d_sinc) {
        std::vector<float> tab(RS_LEN);
        for (int i = 0; i < RS_LEN; i++) {
            const double w = refc::RS_WIN_A0 - refc::RS_WIN_A1 * std::cos((kTwoPi * i) / (RS_LEN - 1)) +
                             refc::RS_WIN_A2 * std::cos((2.0 * kTwoPi * i) / (RS_LEN - 1)) -
                             refc::RS_WIN_A3 * std::cos((3.0 * kTwoPi * i) / (RS_LEN - 1));
            const double fi = kPi * (double)(i - RS_LEN / 2) / (double)RS_PTS;
            tab[i] = (i != RS_LEN / 2) ? (float)(w * std::sin(fi) / fi) : 1.0f;
        }
        CSDR_HIP(hipMalloc((void **)&r->d_sinc, sizeof(float) * RS_LEN));
        CSDR_HIP(hipMemcpy(r->d_sinc, tab.data(), sizeof(float) * RS_LEN, hipMemcpyHostToDevice));
    }
    if (r->d_buf) (void)hipFree(r->d_buf);
    r->d_buf = nullptr;
    CSDR_HIP(hipMalloc((void **)&r->d_buf, cap * 8));
    CSDR_HIP(hipMemset(r->d_buf, 0, cap * 8));
    r->cap_in = cap;
    r->t = 0.0;
    return CSDR_OK;
}
/* CFractResampler::Resample, the four overloads (fractresampler.cpp:144-184, :194-249, :258-297,
 * :306-352).  Return the number of output samples. */
int csdr_resampler_resample_real(csdr_resampler *r, int n, double rate, const double *in, double *out)
{ return rs_run(r, n, rate, false, in, out, nullptr, 0); }
int csdr_resampler_resample_cpx(csdr_resampler *r, int n, double rate, const double *in_iq, double *out_iq)
{ return rs_run(r, n, rate, true, in_iq, out_iq, nullptr, 0); }
int csdr_resampler_resample_real_i16(csdr_resampler *r, int n, double rate, const double *in, short *out, double gain)
{ return rs_run(r, n, rate, false, in, nullptr, out, gain); }
int csdr_resampler_resample_cpx_i16(csdr_resampler *r, int n, double rate, const double *in_iq, short *out_lr, double gain)
{ return rs_run(r, n, rate, true, in_iq, nullptr, out_lr, gain); }

}  // extern "C"

/* ---------------- batch form: every channel on one clock ---------------- */
struct csdr_resampler_batch {
    int device, channels;
    double t = 0.0;                      // m_FloatTime, shared
    float *d_sinc = nullptr, *d_hist = nullptr;      // hist: [2][channels][RS_PERIODS]
    double *d_times = nullptr; size_t cap_times = 0;
    int cur = 0;
    std::vector<double> times;
};

static int rs_build_sinc(float **d_sinc)
{
    std::vector<float> tab(RS_LEN);
    for (int i = 0; i < RS_LEN; i++) {
        const double w = refc::RS_WIN_A0 - refc::RS_WIN_A1 * std::cos((kTwoPi * i) / (RS_LEN - 1)) +
                         refc::RS_WIN_A2 * std::cos((2.0 * kTwoPi * i) / (RS_LEN - 1)) -
                         refc::RS_WIN_A3 * std::cos((3.0 * kTwoPi * i) / (RS_LEN - 1));
        const double fi = kPi * (double)(i - RS_LEN / 2) / (double)RS_PTS;
        tab[i] = (i != RS_LEN / 2) ? (float)(w * std::sin(fi) / fi) : 1.0f;
    }
    CSDR_HIP(hipMalloc((void **)d_sinc, sizeof(float) * RS_LEN));
    CSDR_HIP(hipMemcpy(*d_sinc, tab.data(), sizeof(float) * RS_LEN, hipMemcpyHostToDevice));
    return CSDR_OK;
}

extern "C" {

csdr_resampler_batch *csdr_resampler_batch_create(int device, int channels)
{
    if (channels < 1) { fail(CSDR_EINVAL, "channels >= 1"); return nullptr; }
    if (!device_ok(device)) return nullptr;
    csdr_resampler_batch *b = new csdr_resampler_batch();
    b->device = device; b->channels = channels;
    const size_t hb = sizeof(float) * 2 * channels * RS_PERIODS;
    if (rs_build_sinc(&b->d_sinc) != CSDR_OK || hipMalloc((void **)&b->d_hist, hb) != hipSuccess ||
        hipMemset(b->d_hist, 0, hb) != hipSuccess) {
        csdr_resampler_batch_destroy(b);
        return nullptr;
    }
    return b;
}
void csdr_resampler_batch_destroy(csdr_resampler_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->d_sinc) (void)hipFree(b->d_sinc);
    if (b->d_hist) (void)hipFree(b->d_hist);
    if (b->d_times) (void)hipFree(b->d_times);
    delete b;
}
int csdr_resampler_batch_resample(csdr_resampler_batch *b, const float *d_in, long long in_stride, int n, double rate,
                                  float *d_out_f32, short *d_out_i16, long long out_stride, double gain, void *stream)
{
    if (!b || !d_in || n < 0 || rate <= 0 || (!d_out_f32 == !d_out_i16)) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(b->device)) return CSDR_EHIP;
    // output times: the reference's sequential fp64 accumulation (fractresampler.cpp:157-178)
    b->times.clear();
    int it = (int)b->t;
    while (it < n) { b->times.push_back(b->t); b->t += rate; it = (int)b->t; }
    b->t -= (double)n;
    const int nout = (int)b->times.size();
    if (nout > out_stride) return fail(CSDR_EINVAL, "out_stride %lld < %d output samples", out_stride, nout);
    if ((size_t)nout > b->cap_times) {
        CSDR_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (b->d_times) (void)hipFree(b->d_times);
        b->d_times = nullptr; b->cap_times = 0;
        CSDR_HIP(hipMalloc((void **)&b->d_times, (size_t)nout * 2 * 8));
        b->cap_times = (size_t)nout * 2;
    }
    if (nout) CSDR_HIP(hipMemcpyAsync(b->d_times, b->times.data(), (size_t)nout * 8, hipMemcpyHostToDevice, (hipStream_t)stream));
    CSDR_HIP(hipStreamSynchronize((hipStream_t)stream));      // b->times is reused by the next call
    const size_t half = (size_t)b->channels * RS_PERIODS;
    ResampleBatchArgs a;
    a.in = d_in; a.in_stride = in_stride; a.hist = b->d_hist + b->cur * half; a.hist_next = b->d_hist + (b->cur ^ 1) * half;
    a.sinc = b->d_sinc; a.times = b->d_times; a.out_f32 = d_out_f32; a.out_i16 = d_out_i16; a.out_stride = out_stride;
    a.gain = (float)gain; a.channels = b->channels; a.n = n; a.nout = nout;
    CSDR_HIP(resample_batch_launch(a, (hipStream_t)stream));
    b->cur ^= 1;
    return nout;
}

}  // extern "C"
