// capi_fft.hip -- C ABI for CFft (dsp/fft.h:24-85): display spectrum + plain transforms.
#include "capi_common.hpp"
#include "spectrum_kernels.h"
#include "host_math.hpp"
#include <cstring>
#include <vector>

using namespace csdr;

static int log2_of(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return ((1 << l) == n) ? l : -1;
}

struct csdr_fft_batch {
    int device, channels;
    int size, last_size, ave_size, invert;       // m_FFTSize, m_LastFFTSize, m_AveSize, m_Invert
    double kc, kb, db_comp, fs;
    // screen mapping state (GetScreenIntegerFFTData)
    int start_hz, stop_hz, bin_min, bin_max, plot_w;
    std::vector<int> xlat;
    float *d_win, *d_tw1, *d_tw2, *d_sum, *d_pwr, *d_ave;
    float *d_work;                               // transform work space of the multi-launch sizes
    float *d_part = nullptr; size_t part_cap = 0; // frame-group partial sums
    int *d_cnt, *d_over;
    int *d_scr = nullptr; size_t scr_cap = 0;           // levels of a waterfall line (csdr_fft_batch_get_waterfall_all)
    std::vector<float> h_ave;
    std::vector<int> h_over;
    // the readers (GetScreenIntegerFFTData from the GUI thread, fft.cpp:308-410) wait for THIS object's last
    // PutInDisplayFFT only -- the copy rides on that call's stream -- not for whatever else the device is running
    hipStream_t last_stream = nullptr; bool have_stream = false;
};

// device -> host behind the object's own work: on the stream of its last put_display (until round 5: hipDeviceSynchronize)
static int fft_read(csdr_fft_batch *f, void *dst, const void *src, size_t bytes)
{
    if (!f->have_stream) { CSDR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return CSDR_OK; }
    CSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, f->last_stream));
    CSDR_HIP(hipStreamSynchronize(f->last_stream));
    return CSDR_OK;
}

static void fft_free_dev(csdr_fft_batch *f)
{
    float **ps[] = {&f->d_win, &f->d_tw1, &f->d_tw2, &f->d_sum, &f->d_pwr, &f->d_ave, &f->d_work};
    for (auto p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
}

static int fft_reset(csdr_fft_batch *f)
{   // CFft::ResetFFT (fft.cpp:248-259): clears the averaged and summed buffers and both counters
    const size_t nb = (size_t)f->channels * f->size * 4;
    CSDR_HIP(hipMemset(f->d_ave, 0, nb));
    CSDR_HIP(hipMemset(f->d_sum, 0, nb));
    CSDR_HIP(hipMemset(f->d_cnt, 0, sizeof(int) * 2 * f->channels));
    return CSDR_OK;
}

static int fft_set_params(csdr_fft_batch *f, int size, int invert, double db_comp, double fs)
{   // CFft::SetFFTParams (fft.cpp:118-243)
    if (size == 0) return CSDR_OK;
    f->bin_min = f->bin_max = 0; f->start_hz = f->stop_hz = 0; f->plot_w = 0;
    f->invert = invert; f->fs = fs;
    if (f->db_comp != db_comp) { f->last_size = 0; f->db_comp = db_comp; }
    int n = size < 512 ? 512 : (size > 65536 ? 65536 : size);
    const int l2 = log2_of(n);
    if (l2 < 0) return fail(CSDR_EINVAL, "FFT size %d is not a power of two", n);
    f->size = n;
    if (f->last_size != n) {
        f->last_size = n;
        fft_free_dev(f);
        const size_t nb = (size_t)f->channels * n * 4;
        CSDR_HIP(hipMalloc((void **)&f->d_win, (size_t)n * 4));
        CSDR_HIP(hipMalloc((void **)&f->d_tw1, 8192));
        CSDR_HIP(hipMalloc((void **)&f->d_tw2, 8192));
        CSDR_HIP(hipMalloc((void **)&f->d_sum, nb));
        CSDR_HIP(hipMalloc((void **)&f->d_pwr, nb));
        CSDR_HIP(hipMalloc((void **)&f->d_ave, nb));
        if (l2 < 11 || l2 > 14) CSDR_HIP(hipMalloc((void **)&f->d_work, nb * 4));   // [2][channels][N] complex
        CSDR_HIP(hipMemset(f->d_pwr, 0, nb));
        f->kb = f->db_comp - 20 * std::log10((double)n * refc::FFT_K_AMPMAX / 2.0);
        f->kc = std::pow(10.0, (refc::FFT_K_MINDB - f->kb) / 10.0);
        f->kb = f->kb / 10.0;
        std::vector<float> win(n), tw1(2048), tw2(2048);
        for (int i = 0; i < n; i++) win[i] = (float)(2.0 * (.5 - .5 * std::cos((kTwoPi * i) / (n - 1))));
        for (int i = 0; i < 1024; i++) {
            const double a = kTwoPi * (double)i / (double)n;
            tw1[2 * i] = (float)std::cos(a); tw1[2 * i + 1] = (float)std::sin(a);
        }
        for (int k = 0; k < 32; k++)
            for (int i = 0; i < 32; i++) {
                const double a = kTwoPi * (double)(i * k) / 1024.0;
                tw2[2 * (k * 32 + i)] = (float)std::cos(a); tw2[2 * (k * 32 + i) + 1] = (float)std::sin(a);
            }
        CSDR_HIP(hipMemcpy(f->d_win, win.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(f->d_tw1, tw1.data(), 8192, hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(f->d_tw2, tw2.data(), 8192, hipMemcpyHostToDevice));
        f->xlat.assign(n, 0);
    }
    return fft_reset(f);
}

extern "C" {

csdr_fft_batch *csdr_fft_batch_create(int device, int channels)
{
    if (channels < 1) { fail(CSDR_EINVAL, "channels >= 1"); return nullptr; }
    if (!device_ok(device)) return nullptr;
    csdr_fft_batch *f = new csdr_fft_batch();
    f->device = device; f->channels = channels;
    f->size = 1024; f->last_size = 0; f->ave_size = 1; f->invert = 0; f->db_comp = 0.0; f->fs = 1000;
    f->d_win = f->d_tw1 = f->d_tw2 = f->d_sum = f->d_pwr = f->d_ave = f->d_work = nullptr;
    f->d_cnt = nullptr; f->d_over = nullptr;
    if (hipMalloc((void **)&f->d_cnt, sizeof(int) * 2 * channels) != hipSuccess ||
        hipMalloc((void **)&f->d_over, sizeof(int) * channels) != hipSuccess ||
        fft_set_params(f, 2048, 0, 0.0, 1000) != CSDR_OK) {          // ctor, fft.cpp:41-62
        csdr_fft_batch_destroy(f);
        return nullptr;
    }
    return f;
}
void csdr_fft_batch_destroy(csdr_fft_batch *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    fft_free_dev(f);
    if (f->d_cnt) (void)hipFree(f->d_cnt);
    if (f->d_over) (void)hipFree(f->d_over);
    if (f->d_scr) (void)hipFree(f->d_scr);
    if (f->d_part) (void)hipFree(f->d_part);
    delete f;
}
int csdr_fft_batch_set_params(csdr_fft_batch *f, int size, int invert, double db_comp, double fs)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    return fft_set_params(f, size, invert, db_comp, fs);
}
int csdr_fft_batch_set_ave(csdr_fft_batch *f, int ave)
{   // CFft::SetFFTAve (fft.cpp:103-113)
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    if (f->ave_size != ave) f->ave_size = ave > 0 ? ave : 1;
    return fft_reset(f);
}
int csdr_fft_batch_reset(csdr_fft_batch *f)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    return fft_reset(f);
}
int csdr_fft_batch_size(csdr_fft_batch *f) { return f ? f->size : fail(CSDR_EINVAL, "bad handle"); }

/* nframes frames of `size` samples per channel, back to back in each row; asynchronous */
int csdr_fft_batch_put_display(csdr_fft_batch *f, const float *d_in, long long in_stride, int nframes, void *stream)
{
    if (!f || !d_in || nframes < 0) return fail(CSDR_EINVAL, "bad argument");
    if (nframes == 0) return CSDR_OK;
    if (!device_ok(f->device)) return CSDR_EHIP;
    f->last_stream = (hipStream_t)stream; f->have_stream = true;
    CSDR_HIP(hipMemsetAsync(f->d_over, 0, sizeof(int) * f->channels, (hipStream_t)stream));   // m_Overload = FALSE
    SpectrumArgs a;
    a.in = d_in; a.in_stride = in_stride; a.win = f->d_win; a.tw1 = f->d_tw1; a.tw2 = f->d_tw2;
    a.sum = f->d_sum; a.pwr = f->d_pwr; a.ave = f->d_ave; a.counters = f->d_cnt; a.overload = f->d_over;
    a.channels = f->channels; a.nframes = nframes; a.ave_size = f->ave_size;
    // long calls on few channels: cut each channel's frames into groups so that ONE round of workgroups fills the chip --
    // 4 / 2 / 1 workgroups per CU at 4096 / 8192 / 16384 points (their LDS images; 2048 points fit eight, and measure 2 %
    // better with four).  (Until round 4 the target was 1024 workgroups for every size: at 16384 points that was four
    // rounds, and a workgroup's start -- tables, window, sums, the first frame's latency -- costs five frames' time:
    // 0.39 ms for 256 channels x 32 frames against 0.33; 8192 points -2 %.)
    a.nparts = 1; a.part = nullptr; a.alpha = nullptr;
    {
        const int l2n = log2_of(f->size);
        const long per_cu = l2n <= 12 ? 4 : (l2n == 13 ? 2 : 1);
        const long slots = (l2n >= 11 && l2n <= 14) ? 256 * per_cu : 1024;
        long np = (slots + f->channels - 1) / f->channels;
        if (np > nframes / 8) np = nframes / 8;
        if (np > 1) {
            const size_t need = (size_t)f->channels * np * f->size + (size_t)f->channels * (np + 1);   // + the group weights
            if (need > f->part_cap) {
                CSDR_HIP(hipStreamSynchronize((hipStream_t)stream));
                if (f->d_part) (void)hipFree(f->d_part);
                f->d_part = nullptr; f->part_cap = 0;
                CSDR_HIP(hipMalloc((void **)&f->d_part, need * 4));
                f->part_cap = need;
            }
            a.nparts = (int)np; a.part = f->d_part;
            a.alpha = f->d_part + (size_t)f->channels * np * f->size;
        }
    }
    a.kc = (float)f->kc; a.kb = f->kb;
    const int l2 = log2_of(f->size);
    if (l2 >= 11 && l2 <= 14) CSDR_HIP(spectrum_launch(l2, a, (hipStream_t)stream));
    else CSDR_HIP(spectrum_generic_launch(l2, a, f->d_work, (hipStream_t)stream));
    return CSDR_OK;
}
/* copy of m_pFFTAveBuf of one channel (bels, display order), synchronises */
int csdr_fft_batch_get_ave(csdr_fft_batch *f, int channel, float *out)
{
    if (!f || !out || channel < 0 || channel >= f->channels) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    { const int rc = fft_read(f, out, f->d_ave + (size_t)channel * f->size, (size_t)f->size * 4); if (rc) return rc; }
    return f->size;
}
int csdr_fft_batch_get_total_count(csdr_fft_batch *f, int channel)
{
    if (!f || channel < 0 || channel >= f->channels) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    int c[2];
    { const int rc = fft_read(f, c, f->d_cnt + 2 * channel, sizeof(c)); if (rc) return rc; }
    return c[1];
}

/* CFft::GetScreenIntegerFFTData (fft.cpp:308-410) on the N-float read-back of one channel.
 * Returns 1 if the last PutInDisplayFFT saw an overload, 0 otherwise. */
int csdr_fft_batch_get_screen(csdr_fft_batch *f, int channel, int max_h, int max_w, double max_db, double min_db,
                              int start_hz, int stop_hz, int *out)
{
    if (!f || !out || channel < 0 || channel >= f->channels) return fail(CSDR_EINVAL, "bad argument");
    const int n = f->size;
    f->h_ave.resize(n);
    int rc = csdr_fft_batch_get_ave(f, channel, f->h_ave.data());
    if (rc < 0) return rc;
    int over = 0;
    { const int rc2 = fft_read(f, &over, f->d_over + channel, sizeof(int)); if (rc2) return rc2; }
    const float *ave = f->h_ave.data();
    int i, x, y, ymax = 10000, xprev = -1;
    const double off = max_db / 10.0, gain = -10.0 / (max_db - min_db);
    if (f->start_hz != start_hz || f->stop_hz != stop_hz || f->plot_w != max_w) {
        const int maxbin = n - 1;
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
            for (i = 0; i < f->plot_w && i < n; i++)
                f->xlat[i] = f->bin_min + (i * (f->bin_max - f->bin_min)) / f->plot_w;
        }
    }
    auto level = [&](int bin) {
        int b = f->invert ? (n - bin) : bin;
        if (b >= n) b = n - 1;                         // the reference reads one past the end here
        int v = (int)((double)max_h * gain * ((double)ave[b] - off));
        if (v < 0) v = 0;
        if (v > max_h) v = max_h;
        return v;
    };
    if ((f->bin_max - f->bin_min) > f->plot_w) {
        for (i = f->bin_min; i <= f->bin_max; i++) {
            y = level(i);
            x = f->xlat[i];
            if (x >= f->plot_w) continue;              // the reference writes OutBuf[MaxWidth] here (one past the end)
            if (x == xprev) {
                if (y < ymax) { out[x] = y; ymax = y; }
            } else {
                out[x] = y; xprev = x; ymax = y;
            }
        }
    } else {
        for (x = 0; x < f->plot_w; x++) out[x] = level(f->xlat[x < n ? x : n - 1]);
    }
    return over ? 1 : 0;
}

int csdr_fft_batch_get_screen_all(csdr_fft_batch *f, int max_h, int max_w, double max_db, double min_db,
                                  int start_hz, int stop_hz, int *d_out, long long out_stride, int *d_overload,
                                  void *stream)
{
    if (!f || !d_out || max_w < 0 || out_stride < max_w) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    const int n = f->size, maxbin = n - 1;
    ScreenArgs a;
    a.ave = f->d_ave; a.out = d_out; a.out_stride = out_stride; a.n = n; a.channels = f->channels;
    a.bin_min = (int)((double)start_hz * (double)n / f->fs) + n / 2;      // fft.cpp:336-349
    a.bin_max = (int)((double)stop_hz * (double)n / f->fs) + n / 2;
    if (a.bin_min < 0) a.bin_min = 0;
    if (a.bin_min >= maxbin) a.bin_min = maxbin;
    if (a.bin_max < 0) a.bin_max = 0;
    if (a.bin_max >= maxbin) a.bin_max = maxbin;
    a.plot_w = max_w; a.max_h = max_h; a.invert = f->invert;
    a.off = max_db / 10.0; a.gain = -10.0 / (max_db - min_db);
    CSDR_HIP(screen_launch(a, (hipStream_t)stream));
    if (d_overload)
        CSDR_HIP(hipMemcpyAsync(d_overload, f->d_over, sizeof(int) * f->channels, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CSDR_OK;
}

void csdr_plotter_color_table(unsigned int *out256)
{
    if (out256) for (int i = 0; i < 256; i++) out256[i] = plotter_color(i);
}

int csdr_fft_batch_get_waterfall_all(csdr_fft_batch *f, int max_w, double max_db, double min_db, int start_hz,
                                     int stop_hz, unsigned int *d_rgb, long long out_stride, int *d_overload,
                                     void *stream)
{
    if (!f || !d_rgb || max_w < 0 || out_stride < max_w) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    if (max_w == 0) return CSDR_OK;
    const size_t need = (size_t)f->channels * max_w;
    if (need > f->scr_cap) {                             // the levels of the line, [channels][max_w]
        CSDR_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (f->d_scr) (void)hipFree(f->d_scr);
        f->d_scr = nullptr; f->scr_cap = 0;
        CSDR_HIP(hipMalloc((void **)&f->d_scr, need * sizeof(int)));
        f->scr_cap = need;
    }
    CSDR_HIP(hipMemsetAsync(f->d_scr, 0xff, need * sizeof(int), (hipStream_t)stream));     // -1: not touched
    int rc = csdr_fft_batch_get_screen_all(f, 255, max_w, max_db, min_db, start_hz, stop_hz, f->d_scr, max_w, d_overload, stream);
    if (rc < 0) return rc;
    CSDR_HIP(waterfall_color_launch(f->d_scr, max_w, d_rgb, out_stride, max_w, f->channels, (hipStream_t)stream));
    return CSDR_OK;
}

}  // extern "C"

/* ---------------- single-channel host form: CFft drop-in ---------------- */
struct csdr_fft {
    csdr_fft_batch *b;
    float *d_buf; size_t cap;
    PinnedBuf pin;                   // page-locked fp32 staging of the frame (both directions of the plain transforms)
    hipStream_t s = nullptr;         // the object's stream: the frame's copy and its kernels in order, no host wait in PutInDisplayFFT
    hipEvent_t ev_h2d = nullptr;     // the last frame has left the pinned buffer
    bool h2d_busy = false;
    int total = 0;                   // m_TotalCount mirrored on the host: frames since the last reset (fft.cpp:515-517)
};

static int fft_host_buf(csdr_fft *f, size_t n)
{
    if (n <= f->cap) return CSDR_OK;
    if (f->d_buf) (void)hipFree(f->d_buf);
    f->d_buf = nullptr; f->cap = 0;
    CSDR_HIP(hipMalloc((void **)&f->d_buf, n * 8));
    f->cap = n;
    return CSDR_OK;
}

extern "C" {

csdr_fft *csdr_fft_create(int device)
{
    csdr_fft_batch *b = csdr_fft_batch_create(device, 1);
    if (!b) return nullptr;
    csdr_fft *f = new csdr_fft();
    f->b = b; f->d_buf = nullptr; f->cap = 0;
    if (hipStreamCreateWithFlags(&f->s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&f->ev_h2d, hipEventDisableTiming) != hipSuccess) {
        if (f->s) (void)hipStreamDestroy(f->s);
        csdr_fft_batch_destroy(b); delete f; fail(CSDR_EHIP, "stream creation failed"); return nullptr;
    }
    return f;
}
void csdr_fft_destroy(csdr_fft *f)
{
    if (!f) return;
    (void)hipSetDevice(f->b->device);
    if (f->s) { (void)hipStreamSynchronize(f->s); (void)hipStreamDestroy(f->s); }
    if (f->ev_h2d) (void)hipEventDestroy(f->ev_h2d);
    if (f->d_buf) (void)hipFree(f->d_buf);
    csdr_fft_batch_destroy(f->b);
    delete f;
}
// (the setters below synchronise the device inside the batch object: nothing of this object's stream is in flight after them)
int csdr_fft_set_params(csdr_fft *f, int size, int invert, double db_comp, double fs)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (f->s) CSDR_HIP(hipStreamSynchronize(f->s));
    f->total = 0;
    return csdr_fft_batch_set_params(f->b, size, invert, db_comp, fs);
}
int csdr_fft_set_ave(csdr_fft *f, int ave)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (f->s) CSDR_HIP(hipStreamSynchronize(f->s));
    f->total = 0;                                        // SetFFTAve ends in ResetFFT (fft.cpp:103-113)
    return csdr_fft_batch_set_ave(f->b, ave);
}
int csdr_fft_reset(csdr_fft *f)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (f->s) CSDR_HIP(hipStreamSynchronize(f->s));
    f->total = 0;
    return csdr_fft_batch_reset(f->b);
}

/* CFft::PutInDisplayFFT (fft.cpp:267-288): n should equal the FFT size; returns m_TotalCount */
int csdr_fft_put_display(csdr_fft *f, int n, const double *in_iq)
{
    if (!f || n < 0 || (n && !in_iq)) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->b->device)) return CSDR_EHIP;
    const int N = f->b->size;
    int rc = fft_host_buf(f, (size_t)N);
    if (rc) return rc;
    if ((rc = f->pin.reserve(2 * (size_t)N))) return rc;
    // The frame goes fp64 -> fp32 into pinned memory and from there to the device on the object's stream, the spectrum
    // kernels behind it: nothing here waits for the GPU (the reference's caller does not either -- the readers,
    // GetScreenIntegerFFTData / get_ave, synchronise).  The pinned buffer is reused once its last copy has left.
    if (f->h2d_busy) { CSDR_HIP(hipEventSynchronize(f->ev_h2d)); f->h2d_busy = false; }
    const int m = n < N ? n : N;
    cvt_to_f32(f->pin.p, in_iq, 2 * (size_t)m);
    if (m < N) memset(f->pin.p + 2 * (size_t)m, 0, 2 * (size_t)(N - m) * sizeof(float));   // the reference keeps stale data past n; zeros here
    CSDR_HIP(hipMemcpyAsync(f->d_buf, f->pin.p, (size_t)N * 8, hipMemcpyHostToDevice, f->s));
    CSDR_HIP(hipEventRecord(f->ev_h2d, f->s));
    f->h2d_busy = true;
    rc = csdr_fft_batch_put_display(f->b, f->d_buf, N, 1, (void *)f->s);
    if (rc) return rc;
    return ++f->total;                                   // m_TotalCount (fft.cpp:287): frames since the last reset
}
/* CFft::GetScreenIntegerFFTData (fft.cpp:308-410); returns the overload flag */
int csdr_fft_get_screen(csdr_fft *f, int max_h, int max_w, double max_db, double min_db, int start_hz,
                        int stop_hz, int *out)
{ return f ? csdr_fft_batch_get_screen(f->b, 0, max_h, max_w, max_db, min_db, start_hz, stop_hz, out)
           : fail(CSDR_EINVAL, "bad handle"); }
int csdr_fft_get_ave(csdr_fft *f, float *out)
{ return f ? csdr_fft_batch_get_ave(f->b, 0, out) : fail(CSDR_EINVAL, "bad handle"); }

/* CFft::FwdFFT / RevFFT (fft.cpp:416-426): in-place N-point transform of interleaved doubles.
 * The reference's FwdFFT also feeds the display average; this one does not. */
static int fft_plain(csdr_fft *f, double *inout, int sign)
{
    if (!f || !inout) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->b->device)) return CSDR_EHIP;
    const int N = f->b->size;
    int rc = fft_host_buf(f, (size_t)N);
    if (rc) return rc;
    if ((rc = f->pin.reserve(2 * (size_t)N))) return rc;
    if (f->h2d_busy) { CSDR_HIP(hipEventSynchronize(f->ev_h2d)); f->h2d_busy = false; }
    cvt_to_f32(f->pin.p, inout, 2 * (size_t)N);
    hipStream_t st = f->s;
    CSDR_HIP(hipMemcpyAsync(f->d_buf, f->pin.p, (size_t)N * 8, hipMemcpyHostToDevice, st));
    const int l2 = log2_of(N);
    if (l2 >= 11 && l2 <= 14) CSDR_HIP(fft_plain_launch(l2, sign, f->d_buf, f->d_buf, f->b->d_tw1, f->b->d_tw2, st));
    else CSDR_HIP(fft_generic_plain_launch(l2, sign, f->d_buf, f->d_buf, f->b->d_work, st));
    CSDR_HIP(hipMemcpyAsync(f->pin.p, f->d_buf, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    CSDR_HIP(hipStreamSynchronize(st));
    cvt_to_f64(inout, f->pin.p, 2 * (size_t)N);
    return CSDR_OK;
}
int csdr_fft_fwd(csdr_fft *f, double *inout_iq) { return fft_plain(f, inout_iq, +1); }
int csdr_fft_rev(csdr_fft *f, double *inout_iq) { return fft_plain(f, inout_iq, -1); }

}  // extern "C"
