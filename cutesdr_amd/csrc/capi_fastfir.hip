// capi_fastfir.hip -- C ABI for CFastFIR (single-channel host form and batched device form).
#include "capi_common.hpp"
#include <algorithm>
#include "fastfir_kernels.h"
#include "host_math.hpp"
#include "patch_queue.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace csdr;

static int log2_of(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return ((1 << l) == n) ? l : -1;
}

struct csdr_fastfir_batch {
    int device, channels, n, log2n;
    int cus;                          // compute units of the device (run-length heuristic)
    hipStream_t last_stream;          // stream of the most recent process call (setup waits for it)
    bool per_channel;                 // false: one shared filter
    float *d_h;                       // [filters][n] complex fp32 in pass-F3 register order of the generic kernel
    float *d_h2;                      // N = 16384: the same responses in the pipelined kernel's order (it consumes H as
                                      // its tail groups finish bins); both are kept, a launch may go to either kernel
    float *d_hist;                    // 2 x [channels][n/2] complex fp32 (ping-pong)
    int hist_cur;                     // which half holds the previous call's tail
    int dbg_stage; float *dbg_out;    // diagnostics only (csdr__dbg_fastfir_stage)
    int variant;                      // 0: generic kernel (fastfir_kernels.hip); 2: pipelined build (fastfir2_kernels.hip), every size
    float *d_tw1, *d_tw2;
    double flo, fhi, off, fs;         // last shared-filter parameters (early-out like the reference)
    std::vector<std::vector<cd>> resp;   // natural-order fp64 response per filter
    std::vector<int> perm, perm2;     // device slot -> natural bin, generic / pipelined kernel
    // A new frequency response does not stop anything: SetupParameters designs it on the host and queues the fp32 words
    // of both kernel orders as patches; the NEXT process call applies them on its own stream in front of its launch
    // (patch_queue.hpp) -- in stream order behind every earlier call's reads of H, so one buffer per filter suffices.
    PatchQueue patches;
};

static void build_perm(csdr_fastfir_batch *b)
{
    const int T = b->n / 32;
    b->perm.resize(b->n);
    // slot layout: float4 index j*T + t holds registers r = 2j, 2j+1 of thread t
    for (int j = 0; j < 16; j++)
        for (int t = 0; t < T; t++)
            for (int e = 0; e < 2; e++)
                b->perm[(j * T + t) * 2 + e] = fastfir_bin_of(b->log2n, t, 2 * j + e);
    b->perm2.clear();
    {
        b->perm2.resize(b->n);
        for (int j = 0; j < 16; j++)
            for (int t = 0; t < T; t++)
                for (int e = 0; e < 2; e++) b->perm2[(j * T + t) * 2 + e] = fastfir2_bin_of(b->log2n, t, j, e);
    }
}

static int upload_response(csdr_fastfir_batch *b, int slot, const std::vector<cd> &H)
{
    std::vector<float> dev(2 * (size_t)b->n);
    for (int pass = 0; pass < 2; pass++) {
        const std::vector<int> &perm = pass ? b->perm2 : b->perm;
        float *dst = pass ? b->d_h2 : b->d_h;
        if (perm.empty() || !dst) continue;
        for (int i = 0; i < b->n; i++) {
            const cd v = H[perm[i]];
            dev[2 * i] = (float)v.real();
            dev[2 * i + 1] = (float)v.imag();
        }
        const int rc = b->patches.add(dst + (size_t)slot * 2 * b->n, dev.data(), dev.size() * sizeof(float));
        if (rc) return rc;
    }
    return CSDR_OK;
}

extern "C" {

csdr_fastfir_batch *csdr_fastfir_batch_create(int device, int channels, int fft_size)
{
    const int l2 = log2_of(fft_size);
    if (l2 < 11 || l2 > 14 || channels < 1) {
        fail(CSDR_EINVAL, "fft_size must be 2048..16384 (power of two), channels >= 1");
        return nullptr;
    }
    if (!device_ok(device)) return nullptr;
    csdr_fastfir_batch *b = new csdr_fastfir_batch();
    b->device = device; b->channels = channels; b->n = fft_size; b->log2n = l2;
    b->per_channel = false; b->hist_cur = 0; b->dbg_stage = 0; b->dbg_out = nullptr;
    b->last_stream = nullptr;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        b->cus = cus;
    }
    {
        // N = 16384 runs the software-pipelined build; CSDR_FASTFIR_VARIANT=0
        // forces the generic kernel (diagnostics; launches it cannot take fall back to the generic one anyway)
        const char *v = getenv("CSDR_FASTFIR_VARIANT");
        b->variant = (v && atoi(v) == 0) ? 0 : 2;
    }
    b->d_h = b->d_h2 = b->d_hist = b->d_tw1 = b->d_tw2 = nullptr;
    b->flo = -1.0; b->fhi = 1.0; b->off = 1.0; b->fs = 1.0;      // fastfir.cpp:126-129
    build_perm(b);
    const size_t hbytes = (size_t)fft_size * 8, histbytes = 2 * (size_t)channels * (fft_size / 2) * 8;
    std::vector<float> tw1(2 * 1024), tw2(2 * 1024);
    for (int i = 0; i < 1024; i++) {
        const double a = kTwoPi * (double)i / (double)fft_size;
        tw1[2 * i] = (float)std::cos(a); tw1[2 * i + 1] = (float)std::sin(a);
    }
    for (int k = 0; k < 32; k++)
        for (int i = 0; i < 32; i++) {
            const double a = kTwoPi * (double)(i * k) / 1024.0;
            tw2[2 * (k * 32 + i)] = (float)std::cos(a); tw2[2 * (k * 32 + i) + 1] = (float)std::sin(a);
        }
    bool ok = hipMalloc((void **)&b->d_h, hbytes) == hipSuccess &&
              hipMalloc((void **)&b->d_h2, hbytes) == hipSuccess && hipMemset(b->d_h2, 0, hbytes) == hipSuccess &&
              hipMalloc((void **)&b->d_hist, histbytes) == hipSuccess &&
              hipMalloc((void **)&b->d_tw1, 8192) == hipSuccess &&
              hipMalloc((void **)&b->d_tw2, 8192) == hipSuccess &&
              hipMemset(b->d_h, 0, hbytes) == hipSuccess &&
              hipMemset(b->d_hist, 0, histbytes) == hipSuccess &&
              hipMemcpy(b->d_tw1, tw1.data(), 8192, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_tw2, tw2.data(), 8192, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        fail(CSDR_ENOMEM, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        csdr_fastfir_batch_destroy(b);
        return nullptr;
    }
    b->resp.assign(1, std::vector<cd>(fft_size, cd(0, 0)));
    return b;
}

void csdr_fastfir_batch_destroy(csdr_fastfir_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->d_h) (void)hipFree(b->d_h);
    if (b->d_h2) (void)hipFree(b->d_h2);
    if (b->d_hist) (void)hipFree(b->d_hist);
    if (b->d_tw1) (void)hipFree(b->d_tw1);
    if (b->d_tw2) (void)hipFree(b->d_tw2);
    delete b;
}

int csdr_fastfir_batch_setup(csdr_fastfir_batch *b, int channel, double flo, double fhi,
                             double offset, double fs)
{
    if (!b || channel < -1 || channel >= b->channels) return fail(CSDR_EINVAL, "bad handle/channel");
    if (!device_ok(b->device)) return CSDR_EHIP;
    if (channel < 0) {
        // shared filter: same early-out as the reference (fastfir.cpp:182-186)
        if (!b->per_channel && flo == b->flo && fhi == b->fhi && offset == b->off && fs == b->fs) return 0;
        b->flo = flo; b->fhi = fhi; b->off = offset; b->fs = fs;
    }
    std::vector<cd> H;
    if (!fastfir_design(b->n, flo, fhi, offset, fs, H))
        return fail(CSDR_EINVAL, "filter parameter error (reference keeps the previous taps)");
    if (channel >= 0 && !b->per_channel) {
        // switch to one filter per channel, seeded with the shared one: a reallocation, once in an object's life -- the
        // only part of a set-up that waits (for this handle's queued patches and whatever still reads the old buffer)
        { const int rcp = b->patches.flush(b->last_stream); if (rcp) return rcp; }
        CSDR_HIP(hipDeviceSynchronize());
        float *nh = nullptr;
        const size_t one = (size_t)b->n * 8;
        CSDR_HIP(hipMalloc((void **)&nh, one * b->channels));
        for (int c = 0; c < b->channels; c++)
            CSDR_HIP(hipMemcpy((char *)nh + one * c, b->d_h, one, hipMemcpyDeviceToDevice));
        CSDR_HIP(hipFree(b->d_h));
        b->d_h = nh;
        if (b->d_h2) {
            float *nh2 = nullptr;
            CSDR_HIP(hipMalloc((void **)&nh2, one * b->channels));
            for (int c = 0; c < b->channels; c++)
                CSDR_HIP(hipMemcpy((char *)nh2 + one * c, b->d_h2, one, hipMemcpyDeviceToDevice));
            CSDR_HIP(hipFree(b->d_h2));
            b->d_h2 = nh2;
        }
        b->resp.resize(b->channels, b->resp[0]);
        b->per_channel = true;
    }
    if (channel < 0 && b->per_channel) {
        for (int c = 0; c < b->channels; c++) {
            int rc = upload_response(b, c, H);
            if (rc) return rc;
            b->resp[c] = H;
        }
    } else {
        const int slot = channel < 0 ? 0 : channel;
        int rc = upload_response(b, slot, H);
        if (rc) return rc;
        b->resp[slot] = H;
    }
    return 1;
}

int csdr_fastfir_batch_reset(csdr_fastfir_batch *b)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipMemset(b->d_hist, 0, 2 * (size_t)b->channels * (b->n / 2) * 8));
    return CSDR_OK;
}

int csdr_fastfir_batch_get_response(csdr_fastfir_batch *b, int channel, double *h_out)
{
    if (!b || !h_out || channel < 0 || channel >= b->channels) return fail(CSDR_EINVAL, "bad argument");
    const std::vector<cd> &H = b->resp[b->per_channel ? channel : 0];
    memcpy(h_out, H.data(), sizeof(cd) * H.size());
    return CSDR_OK;
}

int csdr_fastfir_batch_process(csdr_fastfir_batch *b, const float *d_in, long long in_stride,
                               int n_per_channel, float *d_out, long long out_stride,
                               void *stream, int blocks_per_wg)
{
    if (!b || !d_in || !d_out) return fail(CSDR_EINVAL, "bad handle or null buffer");
    const int L = b->n / 2;
    if (n_per_channel <= 0 || n_per_channel % L != 0)
        return fail(CSDR_EINVAL, "n_per_channel (%d) must be a positive multiple of the hop %d", n_per_channel, L);
    if (in_stride < n_per_channel || out_stride < n_per_channel)
        return fail(CSDR_EINVAL, "channel stride shorter than n_per_channel");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15 || (in_stride & 1) || (out_stride & 1))
        return fail(CSDR_EINVAL, "buffers must be 16-byte aligned and strides even");
    // the kernels address a channel row through a 32-bit buffer descriptor range and byte offsets
    if ((long long)n_per_channel * 8 >= (1ll << 31))
        return fail(CSDR_EINVAL, "n_per_channel (%d) too large for one call: at most %d samples", n_per_channel, (1 << 28) - L);
    if (!device_ok(b->device)) return CSDR_EHIP;
    hipStream_t s = (hipStream_t)stream;
    b->last_stream = s;
    { const int rcp = b->patches.flush(s); if (rcp) return rcp; }      // responses set up since the last call
    FastFirArgs a;
    const size_t hist_half = (size_t)b->channels * L * 2;      // floats
    a.in = (const v2f_h *)d_in; a.out = (v2f_h *)d_out;
    a.hist = (const v2f_h *)(b->d_hist + b->hist_cur * hist_half);
    a.hist_next = (v2f_h *)(b->d_hist + (b->hist_cur ^ 1) * hist_half);
    a.h = (const v4f_h *)b->d_h; a.tw1 = (const v2f_h *)b->d_tw1; a.tw2 = (const v2f_h *)b->d_tw2;
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.h_stride = b->per_channel ? b->n / 2 : 0;
    a.channels = b->channels;
    a.nblocks = n_per_channel / L;
    if (blocks_per_wg <= 0) {
        // One workgroup walks a run of consecutive blocks of one channel.  Runs per channel: the count
        // whose workgroups fill whole rounds of the resident slots best (CUs of the device x workgroups per CU
        // at this size: the LDS block is N*8.5 bytes), fewest runs on a tie -- longer runs re-read
        // less overlap.  C3 (256 channels, N=16384): one run of 64 blocks per channel.
        // (N = 4096, pipelined build: two workgroups per CU of two blocks each)
        const long per_cu = b->n >= 16384 ? 1 : (b->n >= 8192 ? 2 : (b->n >= 4096 ? 4 : 8));   // (4096 / 2048: 2 workgroups x 2 / 4 blocks)
        const long slots = (long)b->cus * per_cu;
        long best_runs = 1; double best_eff = -1.0;
        const long max_runs = std::min<long>(a.nblocks, std::max<long>(1, 4 * ((slots + b->channels - 1) / b->channels)));
        for (long runs = 1; runs <= max_runs; runs++) {
            const long wgs = runs * b->channels;
            const double eff = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
            if (eff > best_eff + 1e-9) { best_eff = eff; best_runs = runs; }
        }
        blocks_per_wg = (int)((a.nblocks + best_runs - 1) / best_runs);
    }
    if (blocks_per_wg > a.nblocks) blocks_per_wg = a.nblocks;
    a.blocks_per_run = blocks_per_wg;
    a.runs = (a.nblocks + blocks_per_wg - 1) / blocks_per_wg;
    a.dbg_stage = b->dbg_stage; a.dbg = (v2f_h *)b->dbg_out;
    if (b->variant >= 2) {
        a.h = (const v4f_h *)b->d_h2;         // its own H order
        CSDR_HIP(fastfir2_launch(b->log2n, a, s));     // any block count (pairs, then a single trailing block)
    }
    else CSDR_HIP(fastfir_launch(b->log2n, a, s));
    b->hist_cur ^= 1;        // the kernel left this call's tail in the other half
    return CSDR_OK;
}

/* internal (csdr_demod_batch_set_demod moving a receiver to another plan group): row `sr` of `src` continues as row
 * `dr` of `dst` -- the overlap of the stream so far and the frequency response in use (a later SetupParameters with a
 * parameter error keeps it, fastfir.cpp:195-203).  Same FFT size, both handles idle. */
int csdr__fastfir_batch_copy_row(csdr_fastfir_batch *dst, int dr, csdr_fastfir_batch *src, int sr)
{
    if (!dst || !src || dst->n != src->n || dr < 0 || dr >= dst->channels || sr < 0 || sr >= src->channels)
        return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(dst->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    const int L = dst->n / 2;
    const size_t shalf = (size_t)src->channels * L * 2, dhalf = (size_t)dst->channels * L * 2;
    CSDR_HIP(hipMemcpy(dst->d_hist + dst->hist_cur * dhalf + (size_t)dr * L * 2,
                       src->d_hist + src->hist_cur * shalf + (size_t)sr * L * 2, (size_t)L * 8, hipMemcpyDeviceToDevice));
    const std::vector<cd> &H = src->resp[src->per_channel ? sr : 0];
    if (dst->channels > 1 && !dst->per_channel) return fail(CSDR_ESTATE, "destination rows share one filter");
    const int slot = dst->per_channel ? dr : 0;
    int rc = upload_response(dst, slot, H);
    if (rc) return rc;
    dst->resp[slot] = H;
    dst->flo = dst->fhi = dst->off = dst->fs = std::nan("");      // parameters unknown: the next setup always designs
    return CSDR_OK;
}

/* test-only hook, not part of the public ABI: choose the kernel build of this object (A/B timing in one process) */
int csdr__fastfir_set_variant(csdr_fastfir_batch *b, int variant)
{
    if (!b || (variant != 0 && variant != 2)) return CSDR_EINVAL;
    b->variant = variant;
    return CSDR_OK;
}

/* test-only hook, not part of the public ABI: dump the LDS image after a pass */
int csdr__dbg_fastfir_stage(csdr_fastfir_batch *b, int stage, float *d_dbg)
{
    if (!b) return CSDR_EINVAL;
    b->dbg_stage = stage; b->dbg_out = d_dbg;
    return CSDR_OK;
}

/* ---------------- single-channel host form: CFastFIR drop-in ---------------- */
}  // extern "C"

struct csdr_fastfir {
    csdr_fastfir_batch *b;
    int n, pending;                  // pending < n/2 input samples not yet processed (they sit at the front of pin_in)
    PinnedBuf pin_in, pin_out;       // page-locked fp32 staging, both directions
    hipStream_t s = nullptr;         // the object's stream: H2D copy, filter, D2H copy in order, ONE wait per call
    float *d_in, *d_out;
    size_t cap;                      // device staging capacity in samples
};

static int ensure_cap(csdr_fastfir *f, size_t samples)
{
    if (samples <= f->cap) return CSDR_OK;
    if (f->d_in) (void)hipFree(f->d_in);
    if (f->d_out) (void)hipFree(f->d_out);
    f->d_in = f->d_out = nullptr; f->cap = 0;
    CSDR_HIP(hipMalloc((void **)&f->d_in, samples * 8));
    CSDR_HIP(hipMalloc((void **)&f->d_out, samples * 8));
    f->cap = samples;
    return CSDR_OK;
}

extern "C" {

csdr_fastfir *csdr_fastfir_create(int device, int fft_size)
{
    csdr_fastfir_batch *b = csdr_fastfir_batch_create(device, 1, fft_size);
    if (!b) return nullptr;
    csdr_fastfir *f = new csdr_fastfir();
    f->b = b; f->n = fft_size; f->pending = 0; f->d_in = f->d_out = nullptr; f->cap = 0;
    if (hipStreamCreateWithFlags(&f->s, hipStreamNonBlocking) != hipSuccess) { csdr_fastfir_destroy(f); fail(CSDR_EHIP, "stream creation failed"); return nullptr; }
    return f;
}

void csdr_fastfir_destroy(csdr_fastfir *f)
{
    if (!f) return;
    (void)hipSetDevice(f->b->device);
    if (f->d_in) (void)hipFree(f->d_in);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->s) { (void)hipStreamSynchronize(f->s); (void)hipStreamDestroy(f->s); }
    csdr_fastfir_batch_destroy(f->b);
    delete f;
}

int csdr_fastfir_setup(csdr_fastfir *f, double flo, double fhi, double offset, double fs)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    return csdr_fastfir_batch_setup(f->b, -1, flo, fhi, offset, fs);
}

int csdr_fastfir_process(csdr_fastfir *f, int n, const double *in_iq, double *out_iq)
{
    if (!f || n < 0 || (n > 0 && (!in_iq || !out_iq))) return fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return 0;
    if (!device_ok(f->b->device)) return CSDR_EHIP;
    const int L = f->n / 2;
    // append to the pending (not yet hop-complete) samples, fp64 -> fp32 at the boundary, straight into pinned memory
    int rc = f->pin_in.reserve(2 * ((size_t)f->pending + n));
    if (rc) return rc;
    cvt_to_f32(f->pin_in.p + 2 * (size_t)f->pending, in_iq, 2 * (size_t)n);
    const int avail = f->pending + n;
    const int nproc = (avail / L) * L;
    if (nproc == 0) { f->pending = avail; return 0; }
    // (an error below keeps the call's samples: they are appended to the pending ones, and a later call retries the hops)
    if ((rc = ensure_cap(f, (size_t)nproc))) { f->pending = avail; return rc; }
    if ((rc = f->pin_out.reserve(2 * (size_t)nproc))) { f->pending = avail; return rc; }
    CSDR_HIP(hipMemcpyAsync(f->d_in, f->pin_in.p, (size_t)nproc * 8, hipMemcpyHostToDevice, f->s));
    rc = csdr_fastfir_batch_process(f->b, f->d_in, nproc, nproc, f->d_out, nproc, (void *)f->s, 0);
    if (rc) {                                             // the copy above may still be reading the pinned buffer
        (void)hipStreamSynchronize(f->s);
        f->pending = avail;
        return rc;
    }
    CSDR_HIP(hipMemcpyAsync(f->pin_out.p, f->d_out, (size_t)nproc * 8, hipMemcpyDeviceToHost, f->s));
    CSDR_HIP(hipStreamSynchronize(f->s));
    cvt_to_f64(out_iq, f->pin_out.p, 2 * (size_t)nproc);
    f->pending = avail - nproc;
    if (f->pending > 0) memmove(f->pin_in.p, f->pin_in.p + 2 * (size_t)nproc, 2 * (size_t)f->pending * sizeof(float));
    return nproc;
}

}  // extern "C"
