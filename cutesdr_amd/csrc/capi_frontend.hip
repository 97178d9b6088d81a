// capi_frontend.hip -- C ABI of the input-rate stages in front of the down-converter:
// CNoiseProc (dsp/noiseproc.h:23-58), wire-format unpack (interface/netiobase.cpp:479-527) and the
// NCO-spur DC estimate (interface/sdrinterface.cpp:829-848).
#include "capi_common.hpp"
#include "frontend_kernels.h"
#include "ref_constants.hpp"
#include <cmath>
#include <cstdint>
#include <vector>

using namespace csdr;

struct NbHost {                          // the SetupBlanker comparands (noiseproc.cpp:80-86)
    bool configured = false, on = false;
    double thresh = 0, width = 0, fs = 0;
    int mag_n = 0;                       // as on the device (a rate-only change does not reconfigure: not derivable from fs)
    bool integral = true;                // every sample since the last set-up (which clears sum and buffers) came from datagrams:
                                         // the moving sum and the history are whole multiples of 2^-8 (frontend_kernels.hip)
};

struct csdr_noiseproc_batch {
    int device, channels;
    NbChan *d_chan = nullptr;            // [2][channels], ping-pong with the history
    float *d_hist = nullptr;             // [2][channels][NB_HIST] complex
    int cur = 0;
    std::vector<NbHost> h;
    ~csdr_noiseproc_batch()
    {
        if (d_chan) (void)hipFree(d_chan);
        if (d_hist) (void)hipFree(d_hist);
    }
};

// tests (A/B of the two mask kernels in one process): 0 = always the general kernel
static int &nb_int_switch() { static int on = !(getenv("CSDR_NB_INT") && atoi(getenv("CSDR_NB_INT")) == 0); return on; }
static bool nb_int_enabled() { return nb_int_switch() != 0; }
extern "C" int csdr__noiseproc_set_int(int on) { if (on >= 0) nb_int_switch() = on; return nb_int_switch(); }

static int nb_setup_one(csdr_noiseproc_batch *b, int c, int on, double thresh, double width, double fs)
{
    NbHost &h = b->h[c];
    // the reference's test ends in `SampleRate==SampleRate`: a rate-only change does not reconfigure
    if (h.configured && thresh == h.thresh && width == h.width && h.on == (on != 0)) return CSDR_OK;
    NbChan n;
    n.on = on != 0;
    n.width_n = (int)(width * 1e-6 * fs);
    if (n.width_n < 1) n.width_n = 1;
    else if (n.width_n > NB_MAX_WIDTH) n.width_n = NB_MAX_WIDTH;
    n.mag_n = (int)(refc::NB_MAGAVE_TIME * fs);
    if (n.mag_n > NB_HIST - 1)
        return fail(CSDR_EINVAL, "sample rate %.0f: the 5 ms magnitude window (%d samples) exceeds the "
                    "reference's 32768-entry buffer", fs, n.mag_n);
    n.ratio = .005 * thresh * (double)n.mag_n;
    n.delay_n = n.width_n / 2;
    n.sum = 0.0;
    n.since_trig = 1LL << 40;
    h.configured = true; h.on = on != 0; h.thresh = thresh; h.width = width; h.fs = fs; h.mag_n = n.mag_n;
    h.integral = true;                   // (the buffers are cleared below)
    CSDR_HIP(hipDeviceSynchronize());
    for (int k = 0; k < 2; k++) CSDR_HIP(hipMemcpy(b->d_chan + (size_t)k * b->channels + c, &n, sizeof(n), hipMemcpyHostToDevice));
    const size_t row = (size_t)NB_HIST * 8, half = (size_t)b->channels * row;
    for (int k = 0; k < 2; k++)           // SetupBlanker clears the delay and magnitude buffers (:108-115)
        CSDR_HIP(hipMemset((char *)b->d_hist + k * half + (size_t)c * row, 0, row));
    return CSDR_OK;
}

extern "C" {

csdr_noiseproc_batch *csdr_noiseproc_batch_create(int device, int channels)
{
    if (channels < 1) { fail(CSDR_EINVAL, "channels >= 1"); return nullptr; }
    if (!device_ok(device)) return nullptr;
    csdr_noiseproc_batch *b = new csdr_noiseproc_batch();
    b->device = device; b->channels = channels; b->h.resize(channels);
    if (hipMalloc((void **)&b->d_chan, sizeof(NbChan) * 2 * channels) != hipSuccess ||
        hipMalloc((void **)&b->d_hist, (size_t)2 * channels * NB_HIST * 8) != hipSuccess) {
        fail(CSDR_EHIP, "hipMalloc failed");
        delete b;
        return nullptr;
    }
    for (int c = 0; c < channels; c++)    // ctor: SetupBlanker(false, 50.0, 2.0, 1000.0)
        if (nb_setup_one(b, c, 0, 50.0, 2.0, 1000.0) != CSDR_OK) { delete b; return nullptr; }
    return b;
}
void csdr_noiseproc_batch_destroy(csdr_noiseproc_batch *b) { delete b; }
int csdr_noiseproc_batch_setup(csdr_noiseproc_batch *b, int channel, int on, double threshold, double width_us,
                               double sample_rate)
{
    if (!b || channel >= b->channels) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(b->device)) return CSDR_EHIP;
    for (int c = (channel < 0 ? 0 : channel); c < (channel < 0 ? b->channels : channel + 1); c++) {
        const int rc = nb_setup_one(b, c, on, threshold, width_us, sample_rate);
        if (rc) return rc;
    }
    return CSDR_OK;
}
static int nb_run(csdr_noiseproc_batch *b, const float *d_in, long long in_stride, const WireIn &wire, int n_per_channel,
                  float *d_out, long long out_stride, void *stream, unsigned *d_mask = nullptr, long long mask_stride = 0);
int csdr_noiseproc_batch_process(csdr_noiseproc_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                                 float *d_out, long long out_stride, void *stream)
{
    if (!b || !d_in || !d_out || n_per_channel < 0) return fail(CSDR_EINVAL, "bad argument");
    if (d_in == d_out) return fail(CSDR_EINVAL, "the device form reads samples behind the write position: "
                                   "d_out must not alias d_in");
    return nb_run(b, d_in, in_stride, WireIn{nullptr, 0, 0, 0}, n_per_channel, d_out, out_stride, stream);
}
/* internal (not in the public header): the same pass reading the samples straight from datagrams
 * ([channels][npackets][pkt_len] bytes, interface/netiobase.cpp:479-527) -- no unpacked copy in between */
int csdr__noiseproc_batch_process_packets(csdr_noiseproc_batch *b, const void *d_packets, int npackets, int pkt_len,
                                          float *d_out, long long out_stride, void *stream)
{
    if (!b || !d_packets || !d_out || npackets < 0) return fail(CSDR_EINVAL, "bad argument");
    if (pkt_len != 1028 && pkt_len != 1444) return fail(CSDR_EINVAL, "packet length %d", pkt_len);
    const int per = pkt_len == 1444 ? 240 : 256;
    if ((long)npackets * pkt_len >= (1l << 31)) return fail(CSDR_EINVAL, "a channel's datagrams of one call must stay below 2 GiB");
    // the kernel fetches the 16-bit samples with 32-bit loads (wire_format.hpp): same rule as the down-converter
    if ((uintptr_t)d_packets & 3) return fail(CSDR_EINVAL, "datagram buffer must be 4-byte aligned");
    return nb_run(b, nullptr, 0, WireIn{(const unsigned char *)d_packets, (long)npackets * pkt_len, pkt_len, per},
                  npackets * per, d_out, out_stride, stream);
}
/* internal: the shape of a blanker object (csdr_demod_batch_process_packets / _process_blanked check that it is theirs:
 * the mask rows, and the state / history the down-converter indexes by input row, are sized by the CHAIN's width) */
int csdr__noiseproc_batch_shape(csdr_noiseproc_batch *b, int *channels, int *device)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (channels) *channels = b->channels;
    if (device) *device = b->device;
    return CSDR_OK;
}
/* internal (csdr_demod_batch_process_packets / _process with a blanker): MASK MODE -- the blanker decides, the
 * down-converter applies.  One pass over the call's samples (float rows, or datagrams when d_packets is given) leaves one
 * bit per sample in d_mask ([channels][mask_stride] words; set = blanked) and advances the blanker's state and
 * history; nothing else is written.  *d_state / *d_hist receive the state array ([channels] NbChan: on, delay_n) and
 * the raw-sample history ([channels][NB_HIST] complex: the inputs before this call) the consumer needs to take
 * x[i - delay_n - 1] itself -- the halves as they were BEFORE this call (the kernel does not touch them). */
int csdr__noiseproc_batch_mask(csdr_noiseproc_batch *b, const float *d_in, long long in_stride, const void *d_packets,
                               int npackets, int pkt_len, int n_per_channel, unsigned *d_mask, long long mask_stride,
                               const void **d_state, const float **d_hist, void *stream)
{
    if (!b || (!d_in && !d_packets) || !d_mask || n_per_channel < 0 || !d_state || !d_hist) return fail(CSDR_EINVAL, "bad argument");
    if (mask_stride * 32 < n_per_channel) return fail(CSDR_EINVAL, "mask rows too short");
    WireIn wire{nullptr, 0, 0, 0};
    if (d_packets) {
        if (pkt_len != 1028 && pkt_len != 1444) return fail(CSDR_EINVAL, "packet length %d", pkt_len);
        const int per = pkt_len == 1444 ? 240 : 256;
        if ((long)npackets * pkt_len >= (1l << 31) || ((uintptr_t)d_packets & 3) || npackets * per != n_per_channel)
            return fail(CSDR_EINVAL, "datagram buffer: 4-byte aligned, below 2 GiB per channel, npackets x samples = n");
        wire = WireIn{(const unsigned char *)d_packets, (long)npackets * pkt_len, pkt_len, per};
    }
    const size_t half = (size_t)b->channels * NB_HIST * 2;
    *d_state = b->d_chan + (size_t)b->cur * b->channels;
    *d_hist = b->d_hist + b->cur * half;
    return nb_run(b, d_in, in_stride, wire, n_per_channel, nullptr, 0, stream, d_mask, mask_stride);
}
static int nb_run(csdr_noiseproc_batch *b, const float *d_in, long long in_stride, const WireIn &wire, int n_per_channel,
                  float *d_out, long long out_stride, void *stream, unsigned *d_mask, long long mask_stride)
{
    if (n_per_channel == 0) return CSDR_OK;
    if (!device_ok(b->device)) return CSDR_EHIP;
    const size_t half = (size_t)b->channels * NB_HIST * 2;
    NbArgs a;
    a.wire = wire;
    a.chan = b->d_chan + (size_t)b->cur * b->channels; a.chan_next = b->d_chan + (size_t)(b->cur ^ 1) * b->channels; a.in = d_in; a.in_stride = in_stride; a.out = d_out; a.out_stride = out_stride;
    a.hist = b->d_hist + b->cur * half; a.hist_next = b->d_hist + (b->cur ^ 1) * half;
    a.mask = d_mask; a.mask_stride = mask_stride;
    a.channels = b->channels; a.n = n_per_channel;
    static const bool ring_env = !(getenv("CSDR_NB_RING") && atoi(getenv("CSDR_NB_RING")) == 0);
    a.ring = ring_env;
    // float rows: from here on the sums may carry roundings -- the integer form is off until the next set-up clears them
    if (!wire.pk) for (auto &h : b->h) h.integral = false;
    a.int_ok = nb_int_enabled() && wire.pk != nullptr;
    for (int c = 0; c < b->channels && a.int_ok; c++) if (b->h[c].on && !b->h[c].integral) a.int_ok = 0;
    for (int c = 0; c < b->channels && a.ring; c++)
        if (b->h[c].on && (b->h[c].mag_n + 1 < noiseblank_ring_min(d_mask != nullptr) ||
                           b->h[c].mag_n + 1 > noiseblank_ring_max(d_mask != nullptr))) a.ring = 0;
    // segments: enough workgroups to fill the chip, each at least 32 tiles long (the longest blank
    // width is 4 tiles, the moving-sum reduction at a segment start another ~10-32 tiles' worth of reads)
    // (mask mode: 74 registers, three 512-thread workgroups per CU: 3072 workgroups = four rounds measured best,
    // 3.28 ms for the C4 share's datagram-fed chain against 3.36 with 2048 and 3.49 with 4096)
    // (with the ring: two workgroups per CU -- 64 KB of ring each -- and ONE round of them: mask form 1.07 ms against
    // 1.11 with two rounds, 1.18 with six; sample form 2.20 against 2.24 and 2.29 with four)
    static const long want_env = getenv("CSDR_NB_WGS") ? atol(getenv("CSDR_NB_WGS")) : 0;
    const long want_wgs = want_env > 0 ? want_env : (a.ring ? 512 : (d_mask ? 3072 : 2048));
    long nseg = (want_wgs + b->channels - 1) / b->channels;
    const long tile = noiseblank_tile(d_mask != nullptr);
    const long min_seg = 32 * tile;
    if (nseg > n_per_channel / min_seg) nseg = n_per_channel / min_seg;
    if (nseg < 1) nseg = 1;
    long seg_len = (n_per_channel + nseg - 1) / nseg;
    seg_len = (seg_len + tile - 1) / tile * tile;
    a.seg_len = (int)seg_len;
    a.nseg = (int)((n_per_channel + seg_len - 1) / seg_len);
    CSDR_HIP(noiseblank_launch(a, (hipStream_t)stream));
    b->cur ^= 1;
    return CSDR_OK;
}

}  // extern "C"

/* ---------------- single-channel host form: CNoiseProc drop-in ---------------- */
struct csdr_noiseproc {
    csdr_noiseproc_batch *b;
    float *d_in = nullptr, *d_out = nullptr; size_t cap = 0;
    std::vector<float> st;
};

extern "C" {

csdr_noiseproc *csdr_noiseproc_create(int device)
{
    csdr_noiseproc_batch *b = csdr_noiseproc_batch_create(device, 1);
    if (!b) return nullptr;
    csdr_noiseproc *p = new csdr_noiseproc();
    p->b = b;
    return p;
}
void csdr_noiseproc_destroy(csdr_noiseproc *p)
{
    if (!p) return;
    (void)hipSetDevice(p->b->device);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    csdr_noiseproc_batch_destroy(p->b);
    delete p;
}
int csdr_noiseproc_setup(csdr_noiseproc *p, int on, double threshold, double width_us, double sample_rate)
{ return p ? csdr_noiseproc_batch_setup(p->b, 0, on, threshold, width_us, sample_rate) : fail(CSDR_EINVAL, "bad handle"); }
int csdr_noiseproc_process(csdr_noiseproc *p, int n, const double *in_iq, double *out_iq)
{
    if (!p || n < 0 || (n && (!in_iq || !out_iq))) return fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return 0;
    if (!device_ok(p->b->device)) return CSDR_EHIP;
    if ((size_t)n > p->cap) {
        if (p->d_in) (void)hipFree(p->d_in);
        if (p->d_out) (void)hipFree(p->d_out);
        p->d_in = p->d_out = nullptr; p->cap = 0;
        CSDR_HIP(hipMalloc((void **)&p->d_in, (size_t)n * 8));
        CSDR_HIP(hipMalloc((void **)&p->d_out, (size_t)n * 8));
        p->cap = n;
    }
    p->st.resize(2 * (size_t)n);
    for (size_t i = 0; i < 2 * (size_t)n; i++) p->st[i] = (float)in_iq[i];
    CSDR_HIP(hipMemcpy(p->d_in, p->st.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    int rc = csdr_noiseproc_batch_process(p->b, p->d_in, n, n, p->d_out, n, nullptr);
    if (rc) return rc;
    CSDR_HIP(hipMemcpy(p->st.data(), p->d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < 2 * (size_t)n; i++) out_iq[i] = (double)p->st[i];
    return n;
}

/* ---------------- wire format and DC estimate ---------------- */
int csdr_ingest_unpack(int device, const void *d_packets, int channels, int npackets, int pkt_len, float *d_out,
                       long long out_stride, const double *d_dc, void *stream)
{
    if (!d_packets || !d_out || channels < 1 || npackets < 0) return fail(CSDR_EINVAL, "bad argument");
    if (pkt_len != 1028 && pkt_len != 1444)
        return fail(CSDR_EINVAL, "packet length %d: the wire format has 1028-byte (16 bit) and 1444-byte (24 bit) "
                    "datagrams", pkt_len);
    const int per = pkt_len == 1444 ? 240 : 256;
    if ((long long)npackets * per > out_stride) return fail(CSDR_EINVAL, "out_stride too small");
    if ((out_stride & 1) || ((uintptr_t)d_out & 15) || ((uintptr_t)d_packets & 3))
        return fail(CSDR_EINVAL, "unpack: d_packets must be 4-byte aligned, d_out 16-byte aligned, out_stride even");
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(unpack_launch((const unsigned char *)d_packets, (long)npackets * pkt_len, channels, npackets, pkt_len,
                           d_out, out_stride, d_dc, (hipStream_t)stream));
    return npackets * per;
}
int csdr_ingest_unpack_host(int device, const void *packets, int npackets, int pkt_len, double *out_iq)
{
    if (!packets || !out_iq || npackets < 0) return fail(CSDR_EINVAL, "bad argument");
    if (pkt_len != 1028 && pkt_len != 1444) return fail(CSDR_EINVAL, "packet length %d", pkt_len);
    if (npackets == 0) return 0;
    if (!device_ok(device)) return CSDR_EHIP;
    const int per = pkt_len == 1444 ? 240 : 256;
    const size_t nb = (size_t)npackets * pkt_len, ns = (size_t)npackets * per;
    unsigned char *d_p = nullptr; float *d_o = nullptr;
    CSDR_HIP(hipMalloc((void **)&d_p, nb));
    if (hipMalloc((void **)&d_o, ns * 8) != hipSuccess) { (void)hipFree(d_p); return fail(CSDR_EHIP, "hipMalloc failed"); }
    int rc = CSDR_OK;
    std::vector<float> st(2 * ns);
    if (hipMemcpy(d_p, packets, nb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(CSDR_EHIP, "copy in failed");
    if (!rc) { rc = csdr_ingest_unpack(device, d_p, 1, npackets, pkt_len, d_o, (long long)ns, nullptr, nullptr); if (rc > 0) rc = CSDR_OK; }
    if (!rc && hipMemcpy(st.data(), d_o, ns * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CSDR_EHIP, "copy out failed");
    (void)hipFree(d_p); (void)hipFree(d_o);
    if (rc) return rc;
    for (size_t i = 0; i < 2 * ns; i++) out_iq[i] = (double)st[i];
    return (int)ns;
}
int csdr_ingest_spurcal(int device, const float *d_iq, long long in_stride, int channels, int n, double *d_dc,
                        void *stream)
{
    if (!d_iq || !d_dc || channels < 1 || n < 0) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(spurcal_launch(d_iq, in_stride, channels, n, d_dc, (hipStream_t)stream));
    return CSDR_OK;
}
int csdr_ingest_spurcal_host(int device, int n, const double *in_iq, double *dc_iq)
{
    if (!in_iq || !dc_iq || n < 0) return fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return CSDR_OK;
    if (!device_ok(device)) return CSDR_EHIP;
    float *d_x = nullptr; double *d_dc = nullptr;
    std::vector<float> st(2 * (size_t)n);
    for (size_t i = 0; i < st.size(); i++) st[i] = (float)in_iq[i];
    CSDR_HIP(hipMalloc((void **)&d_x, (size_t)n * 8));
    if (hipMalloc((void **)&d_dc, 16) != hipSuccess) { (void)hipFree(d_x); return fail(CSDR_EHIP, "hipMalloc failed"); }
    int rc = CSDR_OK;
    if (hipMemcpy(d_x, st.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_dc, dc_iq, 16, hipMemcpyHostToDevice) != hipSuccess) rc = fail(CSDR_EHIP, "copy in failed");
    if (!rc) rc = csdr_ingest_spurcal(device, d_x, n, 1, n, d_dc, nullptr);
    if (!rc && hipMemcpy(dc_iq, d_dc, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CSDR_EHIP, "copy out failed");
    (void)hipFree(d_x); (void)hipFree(d_dc);
    return rc;
}

}  // extern "C"
