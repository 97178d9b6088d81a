// capi_shard.hip -- C ABI of the multi-GPU form of the batched receive chain: ONE host object that owns
// N shards of a csdr_demod_batch, each on its own device.  Receivers are independent (no cross-channel term anywhere in
// dsp/), so shard s owns the contiguous channel range [s C / N, (s+1) C / N) with all its state resident on its device
// and the data path has no collective: a process call is N asynchronous batch calls on N streams.  What does cross
// devices is what SURVEY 8e names: the S-meters gathered to the host (csdr_demod_shard_get_smeter_all) and, for
// receivers cut from one radio's stream (interface/sdrinterface.cpp:903 hands every CDemodulator the same buffer),
// the broadcast of that wide-band block from the device it arrived on to every other shard's device
// (csdr_demod_shard_process_shared: hipMemcpyPeerAsync over xGMI, one copy per device, each on its shard's stream).
// Several shards may name the same device ordinal (tests: two shards on device 0 equal one batch of twice the width).
// bench.py --gpus N keeps its one-process-per-GPU form (torch.distributed over RCCL): this object is for a host that
// drives a whole node from one process, like the reference's single CSdrInterface.
#include "capi_common.hpp"
#include <map>
#include <vector>

using namespace csdr;

struct csdr_demod_shard {
    int channels = 0, fft_n = 2048;
    std::vector<int> device, first, count;               // per shard: device ordinal, channel range
    std::vector<csdr_demod_batch *> b;
    std::vector<hipStream_t> stream;                     // the shard's own stream (calls given no stream use it)
    std::vector<float *> d_block; std::vector<size_t> block_cap;    // shared-stream mode: the wide-band block on each device
    std::vector<float *> d_sm;                           // 2 x count floats per shard: S-meter averages, peaks
    std::vector<int> input_row; int nrows = 0;           // shared-stream mode: receiver -> row of the block
    std::vector<csdr_noiseproc_batch *> nb;              // per shard: CNoiseProc's blanker of its receivers (csdr_demod_shard_set_blanker):
                                                         // empty (none) or one per shard, never partly built
    std::vector<hipEvent_t> ev_block;                    // per shard: process_shared's reads of the caller's block are done
    std::map<int, hipEvent_t> ev_ready;                  // per source device: the caller's block is complete (process_shared)
    bool committed = false, pipelined = false;
    // tests on a one-GPU box (csdr__demod_shard_force_peer): every shard but the first takes the PEER copy of the broadcast
    // even when it names the source's device (hipMemcpyPeerAsync between two ordinals that are equal is legal); peer_copies
    // counts the peer copies issued so far
    bool force_peer = false;
    long peer_copies = 0;
    ~csdr_demod_shard()
    {
        for (size_t s = 0; s < b.size(); s++) {
            (void)hipSetDevice(device[s]);
            if (stream[s]) { (void)hipStreamSynchronize(stream[s]); }
            if (b[s]) csdr_demod_batch_destroy(b[s]);
            if (s < nb.size() && nb[s]) csdr_noiseproc_batch_destroy(nb[s]);
            if (d_block[s]) (void)hipFree(d_block[s]);
            if (d_sm[s]) (void)hipFree(d_sm[s]);
            if (s < ev_block.size() && ev_block[s]) (void)hipEventDestroy(ev_block[s]);
            if (stream[s]) (void)hipStreamDestroy(stream[s]);
        }
        for (auto &e : ev_ready) { (void)hipSetDevice(e.first); (void)hipEventDestroy(e.second); }
    }
    int shard_of(int channel) const
    {
        for (size_t s = 0; s < first.size(); s++) if (channel >= first[s] && channel < first[s] + count[s]) return (int)s;
        return -1;
    }
};

extern "C" {

csdr_demod_shard *csdr_demod_shard_create(const int *devices, int nshards, int channels, int fastfir_n)
{
    if (!devices || nshards < 1 || channels < nshards) { fail(CSDR_EINVAL, "need devices, 1 <= shards <= channels"); return nullptr; }
    csdr_demod_shard *S = new csdr_demod_shard();
    S->channels = channels; S->fft_n = fastfir_n;
    for (int s = 0; s < nshards; s++) {
        const int lo = (int)((long)channels * s / nshards), hi = (int)((long)channels * (s + 1) / nshards);
        S->device.push_back(devices[s]); S->first.push_back(lo); S->count.push_back(hi - lo);
        S->b.push_back(nullptr); S->stream.push_back(nullptr); S->d_block.push_back(nullptr); S->block_cap.push_back(0);
        S->d_sm.push_back(nullptr);
    }
    for (int s = 0; s < nshards; s++) {
        if (!device_ok(devices[s])) { delete S; return nullptr; }
        S->b[s] = csdr_demod_batch_create(devices[s], S->count[s], fastfir_n);
        if (!S->b[s] || hipStreamCreateWithFlags(&S->stream[s], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&S->d_sm[s], sizeof(float) * 2 * S->count[s]) != hipSuccess) {
            if (S->b[s]) fail(CSDR_EHIP, "stream / buffer creation failed on device %d", devices[s]);
            delete S;
            return nullptr;
        }
    }
    return S;
}
void csdr_demod_shard_destroy(csdr_demod_shard *S) { delete S; }

int csdr_demod_shard_count(csdr_demod_shard *S) { return S ? (int)S->b.size() : fail(CSDR_EINVAL, "bad handle"); }
int csdr_demod_shard_range(csdr_demod_shard *S, int shard, int *first, int *count, int *device)
{
    if (!S || shard < 0 || shard >= (int)S->b.size()) return fail(CSDR_EINVAL, "bad argument");
    if (first) *first = S->first[shard];
    if (count) *count = S->count[shard];
    if (device) *device = S->device[shard];
    return CSDR_OK;
}
int csdr_demod_shard_set_input_rate(csdr_demod_shard *S, double rate)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    for (auto *b : S->b) { const int rc = csdr_demod_batch_set_input_rate(b, rate); if (rc) return rc; }
    return CSDR_OK;
}
#define SHARD_OF(S, channel, s)                                                                   \
    if (!(S) || (channel) < 0 || (channel) >= (S)->channels) return fail(CSDR_EINVAL, "bad handle / channel"); \
    const int s = (S)->shard_of(channel)
/* CDemodulator::SetDemod of receiver `channel` (global id), routed to the shard that owns it */
int csdr_demod_shard_set_demod(csdr_demod_shard *S, int channel, int mode, const csdr_demod_info *info)
{
    SHARD_OF(S, channel, s);
    return csdr_demod_batch_set_demod(S->b[s], channel - S->first[s], mode, info);
}
int csdr_demod_shard_set_freq(csdr_demod_shard *S, int channel, double freq)
{
    SHARD_OF(S, channel, s);
    return csdr_demod_batch_set_freq(S->b[s], channel - S->first[s], freq);
}
double csdr_demod_shard_get_output_rate(csdr_demod_shard *S, int channel)
{
    if (!S || channel < 0 || channel >= S->channels) return 0.0;
    const int s = S->shard_of(channel);
    return csdr_demod_batch_get_output_rate(S->b[s], channel - S->first[s]);
}
int csdr_demod_shard_out_count(csdr_demod_shard *S, int channel)
{
    SHARD_OF(S, channel, s);
    return csdr_demod_batch_out_count(S->b[s], channel - S->first[s]);
}
int csdr_demod_shard_commit(csdr_demod_shard *S)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    for (auto *b : S->b) { const int rc = csdr_demod_batch_commit(b); if (rc) return rc; }
    S->committed = true;
    return CSDR_OK;
}
int csdr_demod_shard_set_pipelined(csdr_demod_shard *S, int on)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    for (auto *b : S->b) { const int rc = csdr_demod_batch_set_pipelined(b, on); if (rc) return rc; }
    S->pipelined = on != 0;
    return CSDR_OK;
}
/* One pass of every shard: d_in[s] = shard s's rows [count_s][in_stride] complex fp32 resident on ITS device,
 * d_out[s] = [count_s][out_stride] fp32 audio there.  streams: one per shard, or NULL for the object's own.
 * Asynchronous: N batch calls on N devices, no collective. */
int csdr_demod_shard_process(csdr_demod_shard *S, const float *const *d_in, long long in_stride, int n_per_channel,
                             float *const *d_out, long long out_stride, void *const *streams)
{
    if (!S || !d_in || !d_out) return fail(CSDR_EINVAL, "bad argument");
    int err = 0;
    for (size_t s = 0; s < S->b.size(); s++) {
        const int rc = csdr_demod_batch_process(S->b[s], d_in[s], in_stride, n_per_channel, d_out[s], out_stride,
                                                streams ? streams[s] : (void *)S->stream[s]);
        if (rc < 0 && !err) err = rc;
    }
    return err;
}
/* CNoiseProc::SetupBlanker (noiseproc.cpp:78-119) for every receiver of every shard: the blanker that
 * csdr_demod_shard_process_packets / _process_blanked run fused in front of the chain (on = 0: none). */
int csdr_demod_shard_set_blanker(csdr_demod_shard *S, int on, double threshold, double width_us, double sample_rate)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    // every shard's blanker or none: built aside and published only when all of them exist and are set up
    std::vector<csdr_noiseproc_batch *> nb = S->nb;
    const bool fresh = nb.empty();
    int rc = CSDR_OK;
    if (fresh)
        for (size_t s = 0; s < S->b.size() && rc == CSDR_OK; s++) {
            nb.push_back(csdr_noiseproc_batch_create(S->device[s], S->count[s]));
            if (!nb.back()) { nb.pop_back(); rc = CSDR_EHIP; }       // (the error text is csdr_noiseproc_batch_create's)
        }
    for (size_t s = 0; s < nb.size() && rc == CSDR_OK; s++)
        rc = csdr_noiseproc_batch_setup(nb[s], -1, on, threshold, width_us, sample_rate);
    if (rc != CSDR_OK) {
        if (fresh) for (auto *p : nb) csdr_noiseproc_batch_destroy(p);
        return rc;             // (existing blankers: SetupBlanker of the shards before the failing one has been applied)
    }
    S->nb = nb;
    return CSDR_OK;
}
/* csdr_demod_batch_process_packets on every shard: d_packets[s] = shard s's receivers' datagrams on ITS device
 * ([count_s][npackets][pkt_len] bytes), d_out[s] its audio rows.  With a blanker set (csdr_demod_shard_set_blanker) it
 * runs fused in front, as in the one-device call. */
int csdr_demod_shard_process_packets(csdr_demod_shard *S, const void *const *d_packets, int npackets, int pkt_len,
                                     float *const *d_out, long long out_stride, void *const *streams)
{
    if (!S || !d_packets || !d_out) return fail(CSDR_EINVAL, "bad argument");
    if (!S->nb.empty() && S->nb.size() != S->b.size()) return fail(CSDR_ESTATE, "blanker set up for %zu of %zu shards", S->nb.size(), S->b.size());
    int err = 0;
    for (size_t s = 0; s < S->b.size(); s++) {
        const int rc = csdr_demod_batch_process_packets(S->b[s], d_packets[s], npackets, pkt_len,
                                                        S->nb.empty() ? nullptr : S->nb[s], d_out[s], out_stride,
                                                        streams ? streams[s] : (void *)S->stream[s]);
        if (rc < 0 && !err) err = rc;
    }
    return err;
}
/* csdr_demod_shard_process with the blanker of csdr_demod_shard_set_blanker fused in front (fp32 rows) */
int csdr_demod_shard_process_blanked(csdr_demod_shard *S, const float *const *d_in, long long in_stride, int n_per_channel,
                                     float *const *d_out, long long out_stride, void *const *streams)
{
    if (!S || !d_in || !d_out) return fail(CSDR_EINVAL, "bad argument");
    if (S->nb.size() != S->b.size()) return fail(CSDR_ESTATE, "csdr_demod_shard_set_blanker first");
    int err = 0;
    for (size_t s = 0; s < S->b.size(); s++) {
        const int rc = csdr_demod_batch_process_blanked(S->b[s], d_in[s], in_stride, n_per_channel, S->nb[s], d_out[s], out_stride,
                                                        streams ? streams[s] : (void *)S->stream[s]);
        if (rc < 0 && !err) err = rc;
    }
    return err;
}
/* Shared-stream mode (one radio, many receivers): receiver c reads row input_row[c] of a block of `nrows` wide-band
 * streams; every shard gets the map of its own receivers.  nrows must not exceed the smallest shard's receiver count
 * (the batch object bounds its row indices by its width).  NULL: receiver c reads row c - first_s of its shard's input. */
int csdr_demod_shard_set_input_rows(csdr_demod_shard *S, const int *input_row, int nrows)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    if (!input_row) {
        for (auto *b : S->b) { const int rc = csdr_demod_batch_set_input_rows(b, nullptr); if (rc) return rc; }
        S->input_row.clear(); S->nrows = 0;
        return CSDR_OK;
    }
    for (size_t s = 0; s < S->b.size(); s++) if (nrows < 1 || nrows > S->count[s]) return fail(CSDR_EINVAL, "1 <= nrows <= receivers per shard");
    for (int c = 0; c < S->channels; c++) if (input_row[c] < 0 || input_row[c] >= nrows) return fail(CSDR_EINVAL, "input row %d of receiver %d", input_row[c], c);
    for (size_t s = 0; s < S->b.size(); s++) {
        const int rc = csdr_demod_batch_set_input_rows(S->b[s], input_row + S->first[s]);
        if (rc) return rc;
    }
    S->input_row.assign(input_row, input_row + S->channels); S->nrows = nrows;
    return CSDR_OK;
}
/* One pass in shared-stream mode: d_block [nrows][in_stride] complex fp32 resident on device `src_device`, handed over
 * ONCE; the object copies it to every other shard's device (hipMemcpyPeerAsync on that shard's stream, behind
 * `src_stream`'s work so far) and runs every shard on its copy.  The one broadcast SURVEY 8e names.
 * The caller's block is FREE AGAIN IN src_stream's ORDER when the call returns: src_stream is made to wait for every read
 * of d_block this call enqueues -- the peer copies, and the shards that sit on src_device: strict mode reads the block
 * in place (the whole pass is then in front of src_stream's next work), pipelined mode takes a device-to-device copy
 * first, because there the down-converters of call k still run while call k+1 is being issued.  The per-shard copies
 * d_block[s] are single buffers: before refilling one, the shard's stream waits until the previous call's
 * down-converters have consumed it (csdr__demod_batch_wait_input_free) -- round-4 ADVICE: without that, pipelined call
 * k+1's copy could overwrite what call k was still reading. */
extern "C" int csdr__demod_batch_wait_input_free(csdr_demod_batch *b, void *stream);
int csdr_demod_shard_process_shared(csdr_demod_shard *S, const float *d_block, int src_device, void *src_stream,
                                    long long in_stride, int n_per_channel, float *const *d_out, long long out_stride)
{
    if (!S || !d_block || !d_out || S->nrows < 1) return fail(CSDR_EINVAL, "bad argument (set the input rows first)");
    if (in_stride < n_per_channel) return fail(CSDR_EINVAL, "input stride < n");
    const size_t bytes = (size_t)S->nrows * (size_t)in_stride * 8;
    if (!device_ok(src_device)) return CSDR_EHIP;
    hipEvent_t &ready = S->ev_ready[src_device];
    if (!ready) CSDR_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    CSDR_HIP(hipEventRecord(ready, (hipStream_t)src_stream));
    S->ev_block.resize(S->b.size(), nullptr);
    std::vector<char> recorded(S->b.size(), 0);
    int err = 0;
    // (errors inside the loop are captured, never returned from: the tail below -- src_stream waits for every read of the
    // block already enqueued, the current device goes back to the source's -- runs on every path)
    for (size_t s = 0; s < S->b.size(); s++) {
        if (!device_ok(S->device[s])) { err = CSDR_EHIP; break; }
        if (!S->ev_block[s] && hipEventCreateWithFlags(&S->ev_block[s], hipEventDisableTiming) != hipSuccess) {
            S->ev_block[s] = nullptr;
            err = fail(CSDR_EHIP, "hipEventCreateWithFlags failed for shard %zu", s);
            break;
        }
        const float *in = d_block;
        const bool peer = S->device[s] != src_device || (S->force_peer && s > 0);
        const bool copy = peer || S->pipelined;
        hipError_t e = hipStreamWaitEvent(S->stream[s], ready, 0);
        if (e == hipSuccess && copy) {
            // the previous call's readers of d_block[s] first (pipelined mode; a no-op in strict mode)
            const int rcw = csdr__demod_batch_wait_input_free(S->b[s], (void *)S->stream[s]);
            if (rcw) { err = rcw; break; }
            if (bytes > S->block_cap[s]) {
                (void)hipStreamSynchronize(S->stream[s]);
                if (S->d_block[s]) (void)hipFree(S->d_block[s]);
                S->d_block[s] = nullptr; S->block_cap[s] = 0;
                e = hipMalloc((void **)&S->d_block[s], bytes);
                if (e == hipSuccess) S->block_cap[s] = bytes;
            }
            if (e == hipSuccess) {
                e = peer ? hipMemcpyPeerAsync(S->d_block[s], S->device[s], d_block, src_device, bytes, S->stream[s])
                         : hipMemcpyAsync(S->d_block[s], d_block, bytes, hipMemcpyDeviceToDevice, S->stream[s]);
                if (e == hipSuccess && peer) S->peer_copies++;
            }
            if (e == hipSuccess) e = hipEventRecord(S->ev_block[s], S->stream[s]);     // the caller's block has been read
            in = S->d_block[s];
        }
        if (e != hipSuccess) { err = fail(CSDR_EHIP, "broadcast to device %d: %s", S->device[s], hipGetErrorString(e)); break; }
        const int rc = csdr_demod_batch_process(S->b[s], in, in_stride, n_per_channel, d_out[s], out_stride, (void *)S->stream[s]);
        if (rc < 0 && !err) err = rc;
        // in place (strict mode, same device): the pass itself is the reader; it has joined the shard's stream
        if (!copy && hipEventRecord(S->ev_block[s], S->stream[s]) != hipSuccess && !err) err = fail(CSDR_EHIP, "hipEventRecord");
        recorded[s] = 1;
    }
    // (src_stream may be the NULL stream, which is the CURRENT device's: back to the source device first)
    (void)hipSetDevice(src_device);
    for (size_t s = 0; s < S->b.size(); s++)
        if (recorded[s] && hipStreamWaitEvent((hipStream_t)src_stream, S->ev_block[s], 0) != hipSuccess && !err)
            err = fail(CSDR_EHIP, "hipStreamWaitEvent");
    return err;
}
/* internal (tests on a one-GPU box): on != 0 makes every shard but the first take the peer-copy branch of
 * csdr_demod_shard_process_shared even when it sits on the source's device; returns the number of peer copies issued
 * so far (on < 0: query only) */
long csdr__demod_shard_force_peer(csdr_demod_shard *S, int on)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    if (on >= 0) S->force_peer = on != 0;
    return S->peer_copies;
}
/* waits for everything issued on the shards' own streams */
int csdr_demod_shard_sync(csdr_demod_shard *S)
{
    if (!S) return fail(CSDR_EINVAL, "bad handle");
    for (size_t s = 0; s < S->b.size(); s++) {
        if (!device_ok(S->device[s])) return CSDR_EHIP;
        const int rc = csdr_demod_batch_flush(S->b[s], (void *)S->stream[s]);
        if (rc) return rc;
        CSDR_HIP(hipStreamSynchronize(S->stream[s]));
    }
    return CSDR_OK;
}
/* CSMeter::GetAve / GetPeak of EVERY receiver gathered to host arrays indexed by global channel (either may be NULL;
 * reading the peaks resets them, smeter.cpp:98-103): one collect launch and one small copy per shard.  Synchronous. */
int csdr_demod_shard_get_smeter_all(csdr_demod_shard *S, float *h_ave, float *h_peak)
{
    if (!S || (!h_ave && !h_peak)) return fail(CSDR_EINVAL, "bad argument");
    for (size_t s = 0; s < S->b.size(); s++) {
        if (!device_ok(S->device[s])) return CSDR_EHIP;
        float *da = S->d_sm[s], *dp = S->d_sm[s] + S->count[s];
        const int rc = csdr_demod_batch_get_smeter_all(S->b[s], h_ave ? da : nullptr, h_peak ? dp : nullptr, (void *)S->stream[s]);
        if (rc) return rc;
        if (h_ave) CSDR_HIP(hipMemcpyAsync(h_ave + S->first[s], da, sizeof(float) * S->count[s], hipMemcpyDeviceToHost, S->stream[s]));
        if (h_peak) CSDR_HIP(hipMemcpyAsync(h_peak + S->first[s], dp, sizeof(float) * S->count[s], hipMemcpyDeviceToHost, S->stream[s]));
    }
    for (size_t s = 0; s < S->b.size(); s++) {
        if (!device_ok(S->device[s])) return CSDR_EHIP;
        CSDR_HIP(hipStreamSynchronize(S->stream[s]));
    }
    return CSDR_OK;
}

}  // extern "C"
