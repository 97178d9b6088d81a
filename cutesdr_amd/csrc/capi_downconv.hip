// capi_downconv.hip -- C ABI for CDownConvert (single-channel host form + batched device form).
#include "capi_common.hpp"
#include "downconv_kernels.h"
#include "dc_host.hpp"
#include "patch_queue.hpp"
#include <cstring>
#include <cstdlib>
#include <vector>

// the shortest segment a launch is cut into: tiles, and multiples of the plan's warm-up (see the segment rule below)
#ifndef CSDR_DC_MINSEG_TILES
#define CSDR_DC_MINSEG_TILES 1
#endif
#ifndef CSDR_DC_MINSEG_W
#define CSDR_DC_MINSEG_W 1
#endif

using namespace csdr;

struct csdr_downconvert_batch {
    int device, channels;
    std::vector<DcHostChan> ch;            // host mirror of every channel's CDownConvert state
    std::vector<DcPlan> plans;             // distinct (in_rate, max_bw) decimator chains in use
    std::vector<int> plan_of;              // channel -> plan index
    int hist_stride;                       // samples per channel in each history half
    float *d_hist; int hist_cur;           // 2 x [channels][hist_stride] mixed samples
    DcChan *d_chan; int *d_list; float *d_amp;
    // The NCO state lives on the device (2 x [channels], ping-pong: the kernel advances it) and the channel
    // lists change only with the plans: a process call copies nothing.  The host mirror runs the same
    // arithmetic; after a retune or a rate change (state_dirty / lists_dirty) the next call uploads it.
    int chan_cur; bool state_dirty, lists_dirty;
    // A RETUNE (SetFrequency: a new increment, nothing else) does not stop anything: the channel is marked and the next
    // process call patches its DcChan -- phase and age from the host mirror, which runs the kernel's own arithmetic -- in
    // that call's stream order (patch_queue.hpp).  state_dirty / lists_dirty (rate changes, rows that move) keep the
    // synchronising upload.
    std::vector<char> retuned;
    PatchQueue patches;
    std::vector<int> list_off, list_len;   // per plan: offset and length of its channel list in d_list
    long wgs_hint = 0;                     // > 0: workgroups of the next launches (csdr__downconvert_batch_set_wgs), else one round
};

static int ensure_hist(csdr_downconvert_batch *b)
{
    int need = 2;
    for (auto &p : b->plans) if (p.W > need) need = p.W;
    if (need <= b->hist_stride && b->d_hist) return CSDR_OK;
    // growing the history only happens together with a chain rebuild of the channels that need
    // it; keep the other channels' histories by copying them over
    float *nh = nullptr;
    const size_t half_new = (size_t)b->channels * need * 2;
    CSDR_HIP(hipMalloc((void **)&nh, half_new * 2 * sizeof(float)));
    CSDR_HIP(hipMemset(nh, 0, half_new * 2 * sizeof(float)));
    if (b->d_hist) {
        // old history of channel c (length old stride, right aligned semantics: index 0 is the oldest
        // of the last W samples of ITS plan, W <= old stride) stays valid at the front of the new row
        const size_t half_old = (size_t)b->channels * b->hist_stride * 2;
        CSDR_HIP(hipMemcpy2D(nh, (size_t)need * 8, b->d_hist + b->hist_cur * half_old,
                             (size_t)b->hist_stride * 8, (size_t)b->hist_stride * 8, b->channels,
                             hipMemcpyDeviceToDevice));
        CSDR_HIP(hipFree(b->d_hist));
    }
    b->d_hist = nh; b->hist_stride = need; b->hist_cur = 0;
    return CSDR_OK;
}

static int find_plan(csdr_downconvert_batch *b, double in_rate, double max_bw)
{
    for (size_t i = 0; i < b->plans.size(); i++)
        if (b->plans[i].in_rate == in_rate && b->plans[i].max_bw == max_bw) return (int)i;
    b->plans.push_back(dc_make_plan(in_rate, max_bw));
    return (int)b->plans.size() - 1;
}

/* internal (csdr_demod_batch_set_demod moving a receiver to another plan group): channel `sc` of `src` continues as
 * channel `dc` of `dst` -- the whole CDownConvert object: rates, NCO frequency and CW offset, oscillator phase and
 * age, decimator chain and its stage histories.  Both handles idle (the caller has synchronised). */
extern "C" int csdr__downconvert_batch_copy_channel(csdr_downconvert_batch *dst, int dc, csdr_downconvert_batch *src, int sc)
{
    if (!dst || !src || dc < 0 || dc >= dst->channels || sc < 0 || sc >= src->channels) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(dst->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    dst->ch[dc] = src->ch[sc];
    const DcPlan &sp = src->plans[src->plan_of[sc]];
    dst->plan_of[dc] = find_plan(dst, sp.in_rate, sp.max_bw);
    dst->state_dirty = true; dst->lists_dirty = true;
    int rc = ensure_hist(dst);
    if (rc) return rc;
    const int w = src->hist_stride < dst->hist_stride ? src->hist_stride : dst->hist_stride;
    const size_t shalf = (size_t)src->channels * src->hist_stride * 2, dhalf = (size_t)dst->channels * dst->hist_stride * 2;
    CSDR_HIP(hipMemcpy(dst->d_hist + dst->hist_cur * dhalf + (size_t)dc * dst->hist_stride * 2,
                       src->d_hist + src->hist_cur * shalf + (size_t)sc * src->hist_stride * 2, (size_t)w * 8,
                       hipMemcpyDeviceToDevice));
    return CSDR_OK;
}

/* internal (tests): 1 = every down-converter launch takes the run-time-plan kernel, 0 = precompiled plans where
 * they exist, -1 = query only; returns the number of precompiled plans in the library */
extern "C" int csdr__downconv_force_dynamic(int on) { return downconv_force_dynamic(on); }
/* internal (the batch chain): how many one-wave workgroups the next launches of this handle are cut into; 0 = the
 * default, one round of the chip's 4096 slots.  A down-converter that starts while another group's walks hold part
 * of the chip is sized for what is left, so that it still runs as ONE round (capi_demod.hip). */
extern "C" int csdr__downconvert_batch_set_wgs(csdr_downconvert_batch *b, long wgs)
{
    if (!b || wgs < 0) return fail(CSDR_EINVAL, "bad argument");
    b->wgs_hint = wgs;
    return CSDR_OK;
}
extern "C" int csdr__downconvert_batch_process_rows(csdr_downconvert_batch *b, const float *d_in, long long in_stride,
                                                    const int *d_in_rows, int n_per_channel, float *d_out,
                                                    long long out_stride, void *stream, const void *d_packets, int pkt_len,
                                                    const csdr::DcBlank *blank);

extern "C" {

csdr_downconvert_batch *csdr_downconvert_batch_create(int device, int channels)
{
    if (channels < 1) { fail(CSDR_EINVAL, "channels >= 1"); return nullptr; }
    if (!device_ok(device)) return nullptr;
    csdr_downconvert_batch *b = new csdr_downconvert_batch();
    b->device = device; b->channels = channels;
    b->ch.assign(channels, DcHostChan());
    b->plans.push_back(dc_make_plan(0, 0));          // plan 0: no stages (ctor state)
    b->plan_of.assign(channels, 0);
    b->hist_stride = 0; b->d_hist = nullptr; b->hist_cur = 0;
    b->d_chan = nullptr; b->d_list = nullptr; b->d_amp = nullptr;
    b->chan_cur = 0; b->state_dirty = true; b->lists_dirty = true;
    std::vector<float> amp(DC_AMP_N);
    dc_amp_table(amp.data(), DC_AMP_N);
    bool ok = hipMalloc((void **)&b->d_chan, sizeof(DcChan) * channels * 2) == hipSuccess &&
              hipMalloc((void **)&b->d_list, sizeof(int) * channels) == hipSuccess &&
              hipMalloc((void **)&b->d_amp, sizeof(float) * DC_AMP_N) == hipSuccess &&
              hipMemcpy(b->d_amp, amp.data(), sizeof(float) * DC_AMP_N, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok || ensure_hist(b) != CSDR_OK) {
        fail(CSDR_ENOMEM, "device allocation failed");
        csdr_downconvert_batch_destroy(b);
        return nullptr;
    }
    return b;
}

void csdr_downconvert_batch_destroy(csdr_downconvert_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->d_hist) (void)hipFree(b->d_hist);
    if (b->d_chan) (void)hipFree(b->d_chan);
    if (b->d_list) (void)hipFree(b->d_list);
    if (b->d_amp) (void)hipFree(b->d_amp);
    delete b;
}

#define DCB_CHECK(b, c) \
    if (!(b) || (c) < -1 || (c) >= (b)->channels) return fail(CSDR_EINVAL, "bad handle/channel")
#define DCB_FOR(b, c, i) for (int i = ((c) < 0 ? 0 : (c)); i < ((c) < 0 ? (b)->channels : (c) + 1); i++)

int csdr_downconvert_batch_set_cw_offset(csdr_downconvert_batch *b, int channel, double offset)
{
    DCB_CHECK(b, channel);
    DCB_FOR(b, channel, i) b->ch[i].cw_offset = offset;
    return CSDR_OK;
}

int csdr_downconvert_batch_set_frequency(csdr_downconvert_batch *b, int channel, double freq)
{
    DCB_CHECK(b, channel);
    if (b->retuned.size() != (size_t)b->channels) b->retuned.assign(b->channels, 0);
    DCB_FOR(b, channel, i) { b->ch[i].set_frequency(freq); b->retuned[i] = 1; }   // reaches the device with the next process call
    return CSDR_OK;
}

double csdr_downconvert_batch_set_data_rate(csdr_downconvert_batch *b, int channel, double in_rate, double max_bw)
{
    if (!b || channel < -1 || channel >= b->channels) { fail(CSDR_EINVAL, "bad handle/channel"); return -1.0; }
    if (!device_ok(b->device)) return -1.0;
    double out = 0;
    DCB_FOR(b, channel, i) {
        DcHostChan &c = b->ch[i];
        if (c.in_rate != in_rate || c.max_bw != max_bw) {       // downconvert.cpp:118-119
            c.in_rate = in_rate; c.max_bw = max_bw;
            const int pi = find_plan(b, in_rate, max_bw);
            b->plan_of[i] = pi;
            c.out_rate = b->plans[pi].out_rate;
            // dirty from here on: plan_of / plans have changed, and the calls below can fail -- the next process call
            // must rebuild its per-plan channel lists either way
            b->state_dirty = true; b->lists_dirty = true;
            if (ensure_hist(b) != CSDR_OK) return -1.0;
            // a rebuilt chain starts from zeroed stage histories (ctor of every stage)
            const size_t half = (size_t)b->channels * b->hist_stride * 2;
            if (hipMemset(b->d_hist + b->hist_cur * half + (size_t)i * b->hist_stride * 2, 0,
                          (size_t)b->hist_stride * 8) != hipSuccess) {
                fail(CSDR_EHIP, "hipMemset failed");
                return -1.0;
            }
            c.set_frequency(c.nco_freq);                          // :169, re-adds the CW offset
            b->state_dirty = true; b->lists_dirty = true;
        }
        out = c.out_rate;
    }
    return out;
}

int csdr_downconvert_batch_get_stages(csdr_downconvert_batch *b, int channel, int *codes, int cap)
{
    if (!b || channel < 0 || channel >= b->channels) return fail(CSDR_EINVAL, "bad handle/channel");
    const DcPlan &p = b->plans[b->plan_of[channel]];
    for (int s = 0; s < p.nstages && s < cap; s++) codes[s] = p.kind[s];
    return p.nstages;
}

double csdr_downconvert_batch_get_nco_freq(csdr_downconvert_batch *b, int channel)
{
    if (!b || channel < 0 || channel >= b->channels) return 0.0;
    return b->ch[channel].nco_freq;
}

int csdr_downconvert_batch_out_count(csdr_downconvert_batch *b, int channel, int n_in)
{
    if (!b || channel < 0 || channel >= b->channels) return fail(CSDR_EINVAL, "bad handle/channel");
    return n_in >> b->plans[b->plan_of[channel]].nstages;
}

int csdr_downconvert_batch_process(csdr_downconvert_batch *b, const float *d_in, long long in_stride,
                                   int n_per_channel, float *d_out, long long out_stride, void *stream)
{
    return csdr__downconvert_batch_process_rows(b, d_in, in_stride, nullptr, n_per_channel, d_out, out_stride, stream, nullptr, 0, nullptr);
}

/* internal (not in the public header): input row of channel c is in_rows[c] (device array); with d_packets the
 * samples are read from datagrams ([rows][n_per_channel / per][pkt_len] bytes, wire_format.hpp) instead of d_in; with
 * `blank` the noise blanker's mask (csdr__noiseproc_batch_mask) is applied in the kernel's own loads */
int csdr__downconvert_batch_process_rows(csdr_downconvert_batch *b, const float *d_in, long long in_stride,
                                         const int *d_in_rows, int n_per_channel, float *d_out,
                                         long long out_stride, void *stream, const void *d_packets, int pkt_len,
                                         const DcBlank *blank)
{
    if (!b || (!d_in && !d_packets) || !d_out) return fail(CSDR_EINVAL, "bad handle or null buffer");
    if (n_per_channel <= 0 || (n_per_channel & 1)) return fail(CSDR_EINVAL, "n_per_channel must be even and > 0");
    WireIn wire{nullptr, 0, 0, 0};
    if (d_packets) {
        if (pkt_len != 1028 && pkt_len != 1444) return fail(CSDR_EINVAL, "packet length %d", pkt_len);
        wire.per = pkt_len == 1444 ? 240 : 256;
        if (n_per_channel % wire.per || ((uintptr_t)d_packets & 3)) return fail(CSDR_EINVAL, "whole, 4-byte aligned datagrams");
        wire.pk = (const unsigned char *)d_packets; wire.pkt_len = pkt_len;
        wire.chan_stride = (long)(n_per_channel / wire.per) * pkt_len;
        if (wire.chan_stride >= (1l << 31)) return fail(CSDR_EINVAL, "a channel's datagrams of one call must stay below 2 GiB");
    } else if (((uintptr_t)d_in & 15) || (in_stride & 1) || in_stride < n_per_channel)
        return fail(CSDR_EINVAL, "input must be 16-byte aligned, input stride even and >= n");
    if ((uintptr_t)d_out & 7) return fail(CSDR_EINVAL, "output must be 8-byte aligned");
    if (!device_ok(b->device)) return CSDR_EHIP;
    hipStream_t s = (hipStream_t)stream;
    for (size_t pi = 0; pi < b->plans.size(); pi++)
        for (int i = 0; i < b->channels; i++)
            if (b->plan_of[i] == (int)pi && (n_per_channel & ((1 << b->plans[pi].nstages) - 1)))
                return fail(CSDR_EINVAL, "n_per_channel (%d) must be a multiple of 2^%d for channel %d "
                            "(reference: InLength must be a multiple of 2^stages, downconvert.cpp:181-183)",
                            n_per_channel, b->plans[pi].nstages, i);
    // after a retune / rate change: the host mirror's state and the channel lists go to the device once
    // (setters synchronise: nothing of this handle may still be running on the old values)
    if (b->state_dirty || b->lists_dirty) {
        CSDR_HIP(hipDeviceSynchronize());
        for (char r : b->retuned) if (r) b->state_dirty = true;      // (a retune waiting beside a rate change rides along)
        if (b->state_dirty) {
            std::vector<DcChan> hc(b->channels);
            for (int i = 0; i < b->channels; i++) { hc[i].phase = b->ch[i].phase; hc[i].inc = b->ch[i].inc; hc[i].age = b->ch[i].age; }
            CSDR_HIP(hipMemcpy(b->d_chan + (size_t)b->chan_cur * b->channels, hc.data(), sizeof(DcChan) * b->channels,
                               hipMemcpyHostToDevice));
            b->state_dirty = false;
            b->retuned.assign(b->channels, 0);          // (the full upload carried every increment)
        }
        if (b->lists_dirty) {
            std::vector<int> all(b->channels);
            b->list_off.assign(b->plans.size(), 0); b->list_len.assign(b->plans.size(), 0);
            size_t off = 0;
            for (size_t pi = 0; pi < b->plans.size(); pi++) {
                b->list_off[pi] = (int)off;
                for (int i = 0; i < b->channels; i++) if (b->plan_of[i] == (int)pi) all[off++] = i;
                b->list_len[pi] = (int)off - b->list_off[pi];
            }
            CSDR_HIP(hipMemcpy(b->d_list, all.data(), sizeof(int) * b->channels, hipMemcpyHostToDevice));
            b->lists_dirty = false;
        }
    }
    else if (!b->retuned.empty()) {
        // retunes since the last call: one patch per retuned channel, applied on this call's stream in front of its kernels
        bool any = false;
        for (int i = 0; i < b->channels; i++)
            if (b->retuned[i]) {
                DcChan hc;
                hc.phase = b->ch[i].phase; hc.inc = b->ch[i].inc; hc.age = b->ch[i].age;
                const int rc = b->patches.add(b->d_chan + (size_t)b->chan_cur * b->channels + i, &hc, sizeof(hc));
                if (rc) return rc;
                b->retuned[i] = 0; any = true;
            }
        if (any) { const int rc = b->patches.flush(s); if (rc) return rc; }
    }
    const size_t half = (size_t)b->channels * b->hist_stride * 2;
    std::vector<DcArgs> launches;
    for (size_t pi = 0; pi < b->plans.size(); pi++) {
        if (b->list_len[pi] == 0) continue;
        const DcPlan &p = b->plans[pi];
        DcArgs a;
        memset(&a, 0, sizeof(a));
        a.in = (const dc_v2f *)d_in; a.in_stride = in_stride; a.wire = wire;
        a.out = (dc_v2f *)d_out; a.out_stride = out_stride;
        a.hist = (const dc_v2f *)(b->d_hist + b->hist_cur * half);
        a.hist_next = (dc_v2f *)(b->d_hist + (b->hist_cur ^ 1) * half);
        a.hist_stride = b->hist_stride;
        a.chan = b->d_chan + (size_t)b->chan_cur * b->channels;
        a.chan_next = b->d_chan + (size_t)(b->chan_cur ^ 1) * b->channels;
        a.chan_list = b->d_list + b->list_off[pi]; a.amp = b->d_amp; a.in_rows = d_in_rows;
        if (blank) {
            a.nb_mask = blank->mask; a.nb_mask_stride = blank->mask_stride;
            a.nb_state = (const NbChan *)blank->state; a.nb_hist = (const dc_v2f *)blank->hist;
        }
        a.nchan = b->list_len[pi]; a.n_in = n_per_channel; a.nstages = p.nstages; a.W = p.W;
        for (int q = 0; q < p.nstages; q++) { a.st[q] = p.st[q]; a.kind[q] = p.kind[q]; }
        // segments: enough workgroups to fill the chip, each at least one tile and one warm-up long.  (8 and 8 until round 4:
        // the floor only binds for one or a few receivers and for short calls, where it left most of the chip idle -- a
        // 10 MSPS receiver's call of 2^24 samples ran as 868 workgroups, a 19968-sample pass of the host form as two.
        // Inside one round a launch's time goes as segment + warm-up, so more, shorter segments win until the round is
        // full: C5's down-converter 117 -> 45 us, the host form's 40 -> 13 us per pass, 188 -> 260 MS/s.)
        long min_seg = (long)DC_TILE_SAMPLES * CSDR_DC_MINSEG_TILES;
        if (min_seg < (long)p.W * CSDR_DC_MINSEG_W) min_seg = (long)p.W * CSDR_DC_MINSEG_W;
        // (a segment must not be shorter than the warm-up: segment s > 0 reads the W samples in front of it from THIS
        // call's input -- min_seg < W would read in front of the caller's buffer)
        // ONE full round of the chip's 4096 one-wave slots (256 CUs x 16) -- round 4: every segment pays its warm-up
        // (W samples run through the cascade for nothing: 1024 for the FM plan, 2560 for AM), and with two rounds (8192,
        // rounds 1-3) an 86-receiver group's segments were 43 tiles long: 5-12 % of warm-up.  One round: per-plan launches
        // -3 / -4 / -5 % (FM / SSB / AM, 86 receivers x 2^21), the strict chain -1.6 %, pipelined +-0.  (A few workgroups more
        // than a whole number of rounds start a nearly empty one: 86 x 96 = 8256 took 1.8x the time of 85 x 96.)
        static const long dc_wgs = getenv("CSDR_DC_WGS") ? atol(getenv("CSDR_DC_WGS")) : 4096;
        long nseg = (b->wgs_hint > 0 ? b->wgs_hint : dc_wgs) / a.nchan;
        if (nseg > n_per_channel / min_seg) nseg = n_per_channel / min_seg;
        if (nseg < 1) nseg = 1;
        long seg_len = (n_per_channel + nseg - 1) / nseg;
        seg_len = (seg_len + DC_TILE_SAMPLES - 1) / DC_TILE_SAMPLES * DC_TILE_SAMPLES;
        a.seg_len = (int)seg_len;
        a.nseg = (int)((n_per_channel + seg_len - 1) / seg_len);
        launches.push_back(a);
    }
    // a call shorter than a channel's warm-up keeps part of the old history: every row of the
    // next history half is fully rewritten by the kernel (tail copy + new samples)
    for (auto &la : launches) CSDR_HIP(downconv_launch(la, s));
    b->hist_cur ^= 1;
    b->chan_cur ^= 1;                                   // the kernels left the advanced NCO state in the other half
    for (int i = 0; i < b->channels; i++) {
        b->ch[i].phase += b->ch[i].inc * (unsigned long long)n_per_channel;   // the mirror runs the same arithmetic
        b->ch[i].age += (unsigned long long)n_per_channel;
    }
    return CSDR_OK;
}

}  // extern "C"

/* ---------------- single-channel host form: CDownConvert drop-in ---------------- */
struct csdr_downconvert {
    csdr_downconvert_batch *b;
    float *d_in, *d_out;
    size_t cap;
    std::vector<float> stage;
};

extern "C" {

csdr_downconvert *csdr_downconvert_create(int device)
{
    csdr_downconvert_batch *b = csdr_downconvert_batch_create(device, 1);
    if (!b) return nullptr;
    csdr_downconvert *d = new csdr_downconvert();
    d->b = b; d->d_in = d->d_out = nullptr; d->cap = 0;
    return d;
}
void csdr_downconvert_destroy(csdr_downconvert *d)
{
    if (!d) return;
    (void)hipSetDevice(d->b->device);
    if (d->d_in) (void)hipFree(d->d_in);
    if (d->d_out) (void)hipFree(d->d_out);
    csdr_downconvert_batch_destroy(d->b);
    delete d;
}
int csdr_downconvert_set_cw_offset(csdr_downconvert *d, double offset)
{ return d ? csdr_downconvert_batch_set_cw_offset(d->b, 0, offset) : fail(CSDR_EINVAL, "bad handle"); }
int csdr_downconvert_set_frequency(csdr_downconvert *d, double freq)
{ return d ? csdr_downconvert_batch_set_frequency(d->b, 0, freq) : fail(CSDR_EINVAL, "bad handle"); }
double csdr_downconvert_set_data_rate(csdr_downconvert *d, double in_rate, double max_bw)
{ if (!d) { fail(CSDR_EINVAL, "bad handle"); return -1.0; } return csdr_downconvert_batch_set_data_rate(d->b, 0, in_rate, max_bw); }
int csdr_downconvert_get_stages(csdr_downconvert *d, int *codes, int cap)
{ return d ? csdr_downconvert_batch_get_stages(d->b, 0, codes, cap) : fail(CSDR_EINVAL, "bad handle"); }
double csdr_downconvert_get_nco_freq(csdr_downconvert *d)
{ return d ? csdr_downconvert_batch_get_nco_freq(d->b, 0) : 0.0; }

int csdr_downconvert_process(csdr_downconvert *d, int n, const double *in_iq, double *out_iq)
{
    if (!d || n < 0 || (n > 0 && (!in_iq || !out_iq))) return fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return 0;
    if (!device_ok(d->b->device)) return CSDR_EHIP;
    if ((size_t)n > d->cap) {
        if (d->d_in) (void)hipFree(d->d_in);
        if (d->d_out) (void)hipFree(d->d_out);
        d->d_in = d->d_out = nullptr; d->cap = 0;
        CSDR_HIP(hipMalloc((void **)&d->d_in, (size_t)n * 8));
        CSDR_HIP(hipMalloc((void **)&d->d_out, (size_t)n * 8));
        d->cap = n;
    }
    d->stage.resize(2 * (size_t)n);
    for (size_t i = 0; i < 2 * (size_t)n; i++) d->stage[i] = (float)in_iq[i];
    CSDR_HIP(hipMemcpy(d->d_in, d->stage.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    int rc = csdr_downconvert_batch_process(d->b, d->d_in, n, n, d->d_out, n, nullptr);
    if (rc) return rc;
    const int nout = csdr_downconvert_batch_out_count(d->b, 0, n);
    CSDR_HIP(hipMemcpy(d->stage.data(), d->d_out, (size_t)nout * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < 2 * (size_t)nout; i++) out_iq[i] = (double)d->stage[i];
    return nout;
}

}  // extern "C"
