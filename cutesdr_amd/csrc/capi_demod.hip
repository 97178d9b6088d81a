// capi_demod.hip -- C ABI of the full receive chain, CDemodulator (dsp/demodulator.h:56-100):
// CDownConvert -> CFastFIR -> CSMeter -> CAgc -> AM/SAM/FM/SSB demodulator, device resident
// between the stages.  Two forms: the single-channel host object that mirrors
// CDemodulator::ProcessData call for call, and the batched multi-channel form.
#include "capi_common.hpp"
#include "pc_unit.hpp"
#include "dc_host.hpp"
#include "stream_pool.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

using namespace csdr;

extern "C" int csdr__downconvert_batch_process_rows(csdr_downconvert_batch *b, const float *d_in, long long in_stride,
                                                    const int *d_in_rows, int n_per_channel, float *d_out,
                                                    long long out_stride, void *stream, const void *d_packets, int pkt_len,
                                                    const csdr::DcBlank *blank);
extern "C" int csdr__downconvert_batch_set_wgs(csdr_downconvert_batch *b, long wgs);
extern "C" int csdr__noiseproc_batch_mask(struct csdr_noiseproc_batch *b, const float *d_in, long long in_stride, const void *d_packets,
                                          int npackets, int pkt_len, int n_per_channel, unsigned *d_mask, long long mask_stride,
                                          const void **d_state, const float **d_hist, void *stream);
extern "C" int csdr__noiseproc_batch_shape(struct csdr_noiseproc_batch *b, int *channels, int *device);
extern "C" int csdr__downconvert_batch_copy_channel(csdr_downconvert_batch *dst, int dc, csdr_downconvert_batch *src, int sc);
extern "C" int csdr__fastfir_batch_copy_row(csdr_fastfir_batch *dst, int dr, csdr_fastfir_batch *src, int sr);
extern "C" int csdr__noiseproc_batch_process_packets(struct csdr_noiseproc_batch *b, const void *d_packets, int npackets,
                                                     int pkt_len, float *d_out, long long out_stride, void *stream);

namespace {

// move the not-yet-filtered tail of every row to the front of the staging buffer (dst == src) or of the other
// staging buffer (pipelined mode)
__global__ void shift_rows_kernel(float *dst, const float *src, long stride, int src_off, int count)
{
    float2 *drow = reinterpret_cast<float2 *>(dst) + (long)blockIdx.x * stride;
    const float2 *srow = reinterpret_cast<const float2 *>(src) + (long)blockIdx.x * stride;
    // src_off >= count whenever at least one hop was consumed, so the ranges do not overlap
    for (int i = threadIdx.x; i < count; i += blockDim.x) drow[i] = srow[src_off + i];
}

// `rows` channels that share one decimator plan: staging, pending counts, the three stage objects
struct ChainCore {
    int device = 0, rows = 0, fft_n = 2048, L = 1024;
    csdr_downconvert_batch *dc = nullptr;
    csdr_fastfir_batch *ff = nullptr;
    PcUnit pc;
    float *d_stage = nullptr, *d_filt = nullptr, *d_agc = nullptr;
    // Pipelined mode (csdr_demod_batch_set_pipelined): three stages on three streams -- down-converter on the
    // group's stream, filter (+ staging shift) on s_fir, post-chain on s_post -- with the staging and the filter
    // output ping-ponging between two buffers each, so that stage i of call k+1 never waits for stage i+1 of
    // call k: the down-converters of successive calls run back to back.
    float *d_stage2 = nullptr, *d_filt2 = nullptr;
    long cap = 0;                                        // capacity of every staging row (complex samples)
    hipStream_t s_fir = nullptr, s_post = nullptr;
    hipEvent_t ev_dc = nullptr;                          // down-converter of the current call done
    hipEvent_t ev_stage_free[2] = {nullptr, nullptr};    // filter + shift have finished with staging buffer i
    hipEvent_t ev_fir[2] = {nullptr, nullptr};           // filter output buffer i written
    hipEvent_t ev_post[2] = {nullptr, nullptr};          // post-chain has finished with filter output buffer i
    bool stage_busy[2] = {false, false}, post_pending[2] = {false, false};
    int stage_cur = 0, filt_cur = 0, last_post = -1;
    // long calls run S-meter | AGC | demodulator as a pipeline of launches over burst groups
    hipStream_t s_dem = nullptr, s_sm = nullptr;
    hipEvent_t ev_fork = nullptr, ev_dem = nullptr, ev_sm = nullptr, ev_agc[8] = {};                       // staging capacity per row (complex samples)
    const void *pk = nullptr; int pk_len = 0;   // this call's input as datagrams (csdr_demod_batch_process_packets)
    const DcBlank *blank = nullptr;     // this call's blanker mask, applied by the down-converter (or nullptr)
    int pending = 0;                    // decimated samples waiting for a full hop (same in every row)
    int last_out = 0;
    // stage taps (csdr_demod_set_taps / csdr_demod_batch_set_taps): bit k-1 = PROFILE_k.  Tap 1 -- this call's down-converter
    // output -- is copied to d_tap1 before the staging shift; tap 2 is d_filt; with tap 3 on the post-chain runs as
    // S-meter + AGC into d_agc, then the demodulator from there (the words are those of the fused walk)
    int taps = 0;
    float *d_tap1 = nullptr; long tap1_cap = 0; int tap1_n = 0;

    ~ChainCore()
    {
        if (dc) csdr_downconvert_batch_destroy(dc);
        if (ff) csdr_fastfir_batch_destroy(ff);
        if (d_stage) (void)hipFree(d_stage);
        if (d_filt) (void)hipFree(d_filt);
        if (d_agc) (void)hipFree(d_agc);
        if (d_filt2) (void)hipFree(d_filt2);
        if (d_stage2) (void)hipFree(d_stage2);
        if (d_tap1) (void)hipFree(d_tap1);
        if (ev_ff2) (void)hipEventDestroy(ev_ff2);
        for (hipEvent_t e : ev_pdone2) if (e) (void)hipEventDestroy(e);
        if (s_post) stream_pool().put(device, s_post);
        if (s_fir) stream_pool().put(device, s_fir);
        if (ev_dc) (void)hipEventDestroy(ev_dc);
        for (hipEvent_t e : ev_stage_free) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_fir) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_post) if (e) (void)hipEventDestroy(e);
        if (s_dem) stream_pool().put(device, s_dem);
        if (s_sm) stream_pool().put(device, s_sm);
        for (hipEvent_t e : {ev_fork, ev_dem, ev_sm}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_agc) if (e) (void)hipEventDestroy(e);
    }
    int pipeline_init()
    {
        if (s_dem) return CSDR_OK;
        CSDR_HIP(stream_pool().get(device, 0, &s_dem, STREAM_SIDE));
        CSDR_HIP(stream_pool().get(device, 0, &s_sm, STREAM_SIDE));
        CSDR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        CSDR_HIP(hipEventCreateWithFlags(&ev_dem, hipEventDisableTiming));
        CSDR_HIP(hipEventCreateWithFlags(&ev_sm, hipEventDisableTiming));
        for (auto &e : ev_agc) CSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return CSDR_OK;
    }
    // S-meter, AGC and demodulator of nb bursts: one fused launch.  Optionally (long calls, see below) the
    // S-meter on its own stream and AGC -> demodulator pipelined over burst groups through d_agc.
    int post(const float *filt, float *d_out, long out_stride, const int *d_out_rows, bool stereo, int nb, hipStream_t s)
    {
        const int st = stereo ? PC_STEREO : 0;
        // parameters set since the last call: applied HERE, on the stream every launch below is ordered behind (the optional
        // stage pipeline forks to side streams; a patch kernel on one of them would not be ordered before the others)
        { const int rcp = pc.patches.flush(s); if (rcp) return rcp; }
        // off by default: with four waves per channel one fused launch already fills the chip and the
        // extra launches cost more than the overlap returns (CSDR_CHAIN_PIPELINE=1 turns it on)
        static const bool pipelined = getenv("CSDR_CHAIN_PIPELINE") && atoi(getenv("CSDR_CHAIN_PIPELINE")) != 0;
        if (taps & 4) {                 // PROFILE_3: the AGC's output through device memory
            int rc = pc.run(PC_DO_SMETER | PC_DO_AGC, filt, cap, d_agc, cap, nb, L, s, nullptr);
            if (rc) return rc;
            return pc.run(PC_DO_DEMOD | st, d_agc, cap, d_out, out_stride, nb, L, s, d_out_rows);
        }
        if (nb < 16 || !pipelined)
            return pc.run(PC_DO_SMETER | PC_DO_AGC | PC_DO_DEMOD | st, filt, cap, d_out, out_stride, nb, L, s, d_out_rows);
        int rc = pipeline_init();
        if (rc) return rc;
        const int G = 8;
        CSDR_HIP(hipEventRecord(ev_fork, s));
        CSDR_HIP(hipStreamWaitEvent(s_sm, ev_fork, 0));
        CSDR_HIP(hipStreamWaitEvent(s_dem, ev_fork, 0));
        if ((rc = pc.run(PC_DO_SMETER, filt, cap, nullptr, 0, nb, L, s_sm, nullptr))) return rc;
        for (int g = 0; g < G; g++) {
            const int b0 = (int)((long)nb * g / G), b1 = (int)((long)nb * (g + 1) / G);
            if (b1 == b0) continue;
            const size_t off = (size_t)b0 * L;
            if ((rc = pc.run(PC_DO_AGC, filt + 2 * off, cap, d_agc + 2 * off, cap, b1 - b0, L, s, nullptr))) return rc;
            CSDR_HIP(hipEventRecord(ev_agc[g], s));
            CSDR_HIP(hipStreamWaitEvent(s_dem, ev_agc[g], 0));
            if ((rc = pc.run(PC_DO_DEMOD | st, d_agc + 2 * off, cap, d_out + (stereo ? 2 : 1) * off, out_stride,
                             b1 - b0, L, s_dem, d_out_rows))) return rc;
        }
        CSDR_HIP(hipEventRecord(ev_dem, s_dem));
        CSDR_HIP(hipEventRecord(ev_sm, s_sm));
        CSDR_HIP(hipStreamWaitEvent(s, ev_dem, 0));
        CSDR_HIP(hipStreamWaitEvent(s, ev_sm, 0));
        return CSDR_OK;
    }
    int pipelined_init()
    {
        if (s_post) return CSDR_OK;
        CSDR_HIP(hipDeviceSynchronize());
        int pr_lo = 0, pr_hi = 0;                        // the post-chain is the long pole of a call: highest priority
        CSDR_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        CSDR_HIP(stream_pool().get(device, pr_hi, &s_post, STREAM_STAGE_POST));
        CSDR_HIP(stream_pool().get(device, pr_hi, &s_fir, STREAM_STAGE_FIR));
        CSDR_HIP(hipEventCreateWithFlags(&ev_dc, hipEventDisableTiming));
        for (auto &e : ev_stage_free) CSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ev_fir) CSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ev_post) CSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (cap > 0) {
            if (!d_filt2) CSDR_HIP(hipMalloc((void **)&d_filt2, (size_t)rows * cap * 8));
            if (!d_stage2) CSDR_HIP(hipMalloc((void **)&d_stage2, (size_t)rows * cap * 8));
        }
        stage_cur = 0;                                   // the pending samples sit in d_stage
        return CSDR_OK;
    }
    int init(int dev, int nrows, int n)
    {
        device = dev; rows = nrows; fft_n = n; L = n / 2;
        dc = csdr_downconvert_batch_create(dev, nrows);
        ff = csdr_fastfir_batch_create(dev, nrows, n);
        if (!dc || !ff) return CSDR_EHIP;
        return pc.init(dev, nrows);
    }
    int ensure(long need)
    {
        if (need <= cap) return CSDR_OK;
        need = (need + L + 1023) / 1024 * 1024;
        // growing the staging (rare): the pending samples may still be in flight on a non-blocking stream
        CSDR_HIP(hipDeviceSynchronize());
        float *ns = nullptr, *nf = nullptr, *na = nullptr, *nf2 = nullptr, *ns2 = nullptr;
        CSDR_HIP(hipMalloc((void **)&ns, (size_t)rows * need * 8));
        CSDR_HIP(hipMalloc((void **)&nf, (size_t)rows * need * 8));
        CSDR_HIP(hipMalloc((void **)&na, (size_t)rows * need * 8));
        if (s_post) {
            CSDR_HIP(hipMalloc((void **)&nf2, (size_t)rows * need * 8));
            CSDR_HIP(hipMalloc((void **)&ns2, (size_t)rows * need * 8));
        }
        const float *cur = stage_cur ? d_stage2 : d_stage;     // where the pending samples sit
        if (cur && pending > 0)
            CSDR_HIP(hipMemcpy2D(ns, (size_t)need * 8, cur, (size_t)cap * 8, (size_t)pending * 8, rows,
                                 hipMemcpyDeviceToDevice));
        for (float *p : {d_stage, d_filt, d_agc, d_filt2, d_stage2}) if (p) (void)hipFree(p);
        d_stage = ns; d_filt = nf; d_agc = na; d_filt2 = nf2; d_stage2 = ns2; cap = need;
        stage_cur = 0;
        stage_busy[0] = stage_busy[1] = post_pending[0] = post_pending[1] = false;
        return CSDR_OK;
    }
    // one pass of the chain over n input samples per row (demodulator.cpp:172-207); returns the
    // audio samples produced per row (0 or a multiple of the FastFIR hop)
    int step(const float *d_in, long in_stride, const int *d_in_rows, int n, float *d_out, long out_stride,
             const int *d_out_rows, bool stereo, hipStream_t s, hipEvent_t dc_after = nullptr, hipEvent_t dc_done = nullptr)
    {
        int rc = step_dc(d_in, in_stride, d_in_rows, n, s, dc_after, dc_done);
        if (rc < 0) return rc;
        return step_post(d_out, out_stride, d_out_rows, stereo, s);
    }
    // the two halves of step(): the down-converter of this call into the staging rows ...
    int m_call = 0;                     // decimated samples the down-converter of this call appended
    // chained pipeline (csdr_demod_batch_set_pipelined, round 6): filter + shift stay in the down-converter's stream, the
    // post-chain goes to a second one
    hipEvent_t ev_ff2 = nullptr, ev_pdone2[2] = {nullptr, nullptr};
    bool post_busy2[2] = {false, false};
    int filt_cur2 = 0;
    int step_dc(const float *d_in, long in_stride, const int *d_in_rows, int n, hipStream_t s, hipEvent_t dc_after,
                hipEvent_t dc_done)
    {
        if (s_post) return fail(CSDR_ESTATE, "pipelined objects take step_pipelined()");
        const int m = csdr_downconvert_batch_out_count(dc, 0, n);
        if (m < 0) return m;
        int rc = ensure((long)pending + m);
        if (rc) return rc;
        // the down-converters of the groups run one after the other (each fills the chip on its own);
        // what follows a group's down-converter overlaps with the next group's
        if (dc_after) CSDR_HIP(hipStreamWaitEvent(s, dc_after, 0));
        rc = csdr__downconvert_batch_process_rows(dc, d_in, in_stride, d_in_rows, n, d_stage + 2 * (size_t)pending,
                                                  cap, s, pk, pk_len, blank);
        if (rc) return rc;
        if (dc_done) CSDR_HIP(hipEventRecord(dc_done, s));
        m_call = m;
        return CSDR_OK;
    }
    // The chained pipeline's pass: the down-converter, the filter and the staging shift in stream s -- so that the filter
    // reaches the chip in queue order behind its down-converter, BEFORE the next group's down-converter, which waits for an
    // event between two streams (HISTORY, round 6 (d): the other order starves the filter for a whole launch) -- and the
    // post-chain in stream sp, where it may run on into the next call: s is free for the next call's down-converter as soon
    // as the filter has left.  The filter's output alternates between d_filt and d_agc (idle without the stage taps): with one
    // buffer the next call's filter waited for this call's walk, and a first group's post-chain -- 1.5 ms when its peaks
    // kernel is starved beside the down-converters -- set the period.  *joined: the stream the call's last work is in.
    int step_split(const float *d_in, long in_stride, const int *d_in_rows, int n, float *d_out, long out_stride,
                   const int *d_out_rows, bool stereo, hipStream_t s, hipStream_t sp, hipEvent_t dc_after, hipEvent_t dc_done,
                   hipStream_t *joined)
    {
        if (taps) return fail(CSDR_ESTATE, "stage taps need the strict mode");
        if (!ev_ff2) {
            CSDR_HIP(hipEventCreateWithFlags(&ev_ff2, hipEventDisableTiming));
            for (auto &e : ev_pdone2) CSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        static const bool one_buffer = getenv("CSDR_CHAIN_PIPELINE") && atoi(getenv("CSDR_CHAIN_PIPELINE")) != 0;   // (post() then uses d_agc itself)
        *joined = s;
        int rc = step_dc(d_in, in_stride, d_in_rows, n, s, dc_after, dc_done);
        if (rc < 0) return rc;
        const int total = pending + m_call, nb = total / L;
        m_call = 0; last_out = 0; last_post = -1;
        if (nb == 0) { pending = total; return 0; }
        const int fc = one_buffer ? 0 : filt_cur2;
        filt_cur2 ^= 1;
        float *fb = fc ? d_agc : d_filt;
        if (post_busy2[fc]) { CSDR_HIP(hipStreamWaitEvent(s, ev_pdone2[fc], 0)); post_busy2[fc] = false; }
        rc = csdr_fastfir_batch_process(ff, d_stage, cap, nb * L, fb, cap, s, 0);
        if (rc) return rc;
        CSDR_HIP(hipEventRecord(ev_ff2, s));
        const int rest = total - nb * L;
        if (rest > 0) {
            hipLaunchKernelGGL(shift_rows_kernel, dim3(rows), dim3(256), 0, s, d_stage, d_stage, cap, nb * L, rest);
            CSDR_HIP(hipGetLastError());
        }
        pending = rest;
        CSDR_HIP(hipStreamWaitEvent(sp, ev_ff2, 0));
        rc = post(fb, d_out, out_stride, d_out_rows, stereo, nb, sp);
        if (rc) return rc;
        CSDR_HIP(hipEventRecord(ev_pdone2[fc], sp));
        post_busy2[fc] = true;
        *joined = sp;
        last_out = nb * L;
        return last_out;
    }
    // ... and everything behind it: filter, S-meter, AGC, demodulator, the staging shift
    int step_post(float *d_out, long out_stride, const int *d_out_rows, bool stereo, hipStream_t s)
    {
        const int total = pending + m_call, nb = total / L;
        if (taps & 1) {                 // PROFILE_1: what the down-converter appended in this call, before the staging moves
            if (m_call > tap1_cap) {
                CSDR_HIP(hipStreamSynchronize(s));
                if (d_tap1) (void)hipFree(d_tap1);
                d_tap1 = nullptr; tap1_cap = 0;
                const long want = ((long)m_call + 1023) / 1024 * 1024;
                CSDR_HIP(hipMalloc((void **)&d_tap1, (size_t)rows * want * 8));
                tap1_cap = want;
            }
            if (m_call > 0)
                CSDR_HIP(hipMemcpy2DAsync(d_tap1, (size_t)tap1_cap * 8, d_stage + 2 * (size_t)pending, (size_t)cap * 8,
                                          (size_t)m_call * 8, rows, hipMemcpyDeviceToDevice, s));
            tap1_n = m_call;
        }
        m_call = 0;
        last_out = 0;
        last_post = -1;
        if (nb > 0) {
            int rc = csdr_fastfir_batch_process(ff, d_stage, cap, nb * L, d_filt, cap, s, 0);
            if (rc) return rc;
            rc = post(d_filt, d_out, out_stride, d_out_rows, stereo, nb, s);
            if (rc) return rc;
            const int rest = total - nb * L;
            if (rest > 0) {
                hipLaunchKernelGGL(shift_rows_kernel, dim3(rows), dim3(256), 0, s, d_stage, d_stage, cap, nb * L, rest);
                CSDR_HIP(hipGetLastError());
            }
            pending = rest;
            last_out = nb * L;
        } else {
            pending = total;
        }
        return last_out;
    }
    // The same pass in pipelined mode: s = the group's stream (down-converter only).
    int step_pipelined(const float *d_in, long in_stride, const int *d_in_rows, int n, float *d_out, long out_stride,
                       const int *d_out_rows, bool stereo, hipStream_t s, hipEvent_t dc_after, hipEvent_t dc_done)
    {
        const int m = csdr_downconvert_batch_out_count(dc, 0, n);
        if (m < 0) return m;
        int rc = ensure((long)pending + m);
        if (rc) return rc;
        const int sc = stage_cur;
        float *stage = sc ? d_stage2 : d_stage, *other = sc ? d_stage : d_stage2;
        if (dc_after) CSDR_HIP(hipStreamWaitEvent(s, dc_after, 0));
        // the filter + shift of the call that last used this staging buffer must have finished with it
        if (stage_busy[sc]) { CSDR_HIP(hipStreamWaitEvent(s, ev_stage_free[sc], 0)); stage_busy[sc] = false; }
        rc = csdr__downconvert_batch_process_rows(dc, d_in, in_stride, d_in_rows, n, stage + 2 * (size_t)pending, cap, s,
                                                  pk, pk_len, blank);
        if (rc) return rc;
        CSDR_HIP(hipEventRecord(ev_dc, s));
        if (dc_done) CSDR_HIP(hipEventRecord(dc_done, s));
        const int total = pending + m, nb = total / L;
        last_out = 0;
        last_post = -1;
        if (nb == 0) { pending = total; return 0; }      // not a hop yet: the next call appends to the same buffer
        const int fc = filt_cur;
        filt_cur ^= 1;
        float *fb = fc ? d_filt2 : d_filt;
        CSDR_HIP(hipStreamWaitEvent(s_fir, ev_dc, 0));
        if (post_pending[fc]) CSDR_HIP(hipStreamWaitEvent(s_fir, ev_post[fc], 0));
        rc = csdr_fastfir_batch_process(ff, stage, cap, nb * L, fb, cap, s_fir, 0);
        if (rc) return rc;
        CSDR_HIP(hipEventRecord(ev_fir[fc], s_fir));
        const int rest = total - nb * L;
        if (rest > 0) {                                   // the tail moves to the front of the OTHER staging buffer
            hipLaunchKernelGGL(shift_rows_kernel, dim3(rows), dim3(256), 0, s_fir, other, stage, cap, nb * L, rest);
            CSDR_HIP(hipGetLastError());
        }
        CSDR_HIP(hipEventRecord(ev_stage_free[sc], s_fir));
        stage_busy[sc] = true;
        stage_cur ^= 1;
        pending = rest;
        CSDR_HIP(hipStreamWaitEvent(s_post, ev_fir[fc], 0));
        rc = post(fb, d_out, out_stride, d_out_rows, stereo, nb, s_post);
        if (rc) return rc;
        CSDR_HIP(hipEventRecord(ev_post[fc], s_post));
        post_pending[fc] = true;
        last_post = fc;
        last_out = nb * L;
        return last_out;
    }
};

struct DemodInfo {                      // csdr_demod_info
    int HiCut, HiCutmin, HiCutmax, LowCut, LowCutmin, LowCutmax, FilterClickResolution, Offset, SquelchValue;
    int AgcSlope, AgcThresh, AgcManualGain, AgcDecay, AgcOn, AgcHangOn, Symetric;
};

// per-channel CDemodulator bookkeeping (host side)
struct ChanCfg {
    int mode = -1;
    int pending = -1;                   // batch form: mode requested before commit
    DemodInfo info{};
    double out_rate = 48000.0, want_bw = 48000.0, cw_off = 0.0;
    double demod_rate = 48000.0;        // m_SampleRate of the demodulator OBJECT: the output rate at the time the mode was
                                        // set (amdemod.cpp:50, fmdemod.cpp:62); an input-rate change does not touch it
};

// CDemodulator::SetDemod (dsp/demodulator.cpp:107-157) for row r of core k
int apply_set_demod(ChainCore &k, int r, ChanCfg &c, double in_rate, int mode, const DemodInfo &info)
{
    c.info = info;
    int rc;
    if (c.mode != mode) {
        c.mode = mode;
        if (mode == PC_MODE_LSB || mode == PC_MODE_CWL) c.want_bw = -info.LowCutmin;
        else c.want_bw = info.HiCutmax;
        c.out_rate = csdr_downconvert_batch_set_data_rate(k.dc, r, in_rate, c.want_bw);
        if (c.out_rate < 0) return CSDR_EHIP;
        if ((rc = k.pc.pull(r))) return rc;
        PcChannel &h = k.pc.h[r];
        h.mode = mode;
        c.demod_rate = c.out_rate;
        switch (mode) {                 // new demodulator object = fresh state
        case PC_MODE_AM:  am_init(h.am, k.pc.fir_am[r], c.demod_rate); break;
        case PC_MODE_SAM: sam_init(h.sam, k.pc.fir_sam[r], c.demod_rate); break;
        case PC_MODE_FM:  fm_init(h.fm, k.pc.fir_fm[r], c.demod_rate); break;
        default: break;
        }
        if ((rc = k.pc.push(r))) return rc;
    }
    c.cw_off = info.Offset;
    csdr_downconvert_batch_set_cw_offset(k.dc, r, c.cw_off);
    rc = csdr_fastfir_batch_setup(k.ff, k.rows == 1 ? -1 : r, info.LowCut, info.HiCut, c.cw_off, c.out_rate);
    if (rc < 0 && rc != CSDR_EINVAL) return rc;      // EINVAL = reference's "parameter error": keep old taps
    rc = k.pc.agc_set(r, info.AgcOn, info.AgcHangOn, info.AgcThresh, info.AgcManualGain, info.AgcSlope,
                      info.AgcDecay, c.out_rate);
    if (rc) return rc;
    if ((rc = k.pc.smeter_rate_set(r, c.out_rate))) return rc;
    // (parameter patches: nothing is read back from the device, nothing waits -- pc_unit.hpp)
    if (mode == PC_MODE_FM) rc = k.pc.fm_params_set(r, info.SquelchValue, c.demod_rate, (double)info.HiCut);   // fmdemod.cpp:95-98, :160-164 (the object's own rate)
    else if (mode == PC_MODE_AM) rc = k.pc.am_bandwidth_set(r, c.demod_rate, (info.HiCut - info.LowCut) / 2.0);   // amdemod.cpp:56-60
    return rc;
}

}  // namespace

/* =================== single-channel host form: CDemodulator drop-in =================== */
struct csdr_demod {
    ChainCore k;
    ChanCfg c;
    double in_rate = 0.0;
    int limit = 1000;                   // m_InBufLimit (demodulator.cpp:54)
    // m_pDemodInBuf as fp32 pairs in PINNED memory, two windows used in turn: while window w is on its way to the
    // device and through the chain (asynchronous, on the object's stream), the caller's next samples are converted
    // into window w^1.  The host waits for the GPU only when a pass is due to RETURN samples (a whole hop of audio),
    // as the reference's call pattern demands; passes that only fill the filter return at once.
    // ZERO COPY (round 5): the pinned windows are mapped into the device's address space and the down-converter reads
    // them over PCIe in its own loads, the post-chain writes the audio into the pinned output buffer -- no copy engine, no
    // cross-engine dependency between a copy and the kernel behind it (9 us of copy + 10 us until the kernel started, per
    // pass, and a 4 us copy kernel on the way back).  CSDR_HOST_ZEROCOPY=0 restores the copies (d_in / d_out).
    bool zero_copy = true;
    PinnedBuf win[2], pin_out, pin_out2;
    // DEFERRED output (csdr_demod_set_deferred): a pass that is due to return samples returns the PREVIOUS such pass's
    // instead -- already complete, so the call does not wait -- and leaves its own in the other pinned buffer: the chain's
    // pass runs while the caller converts the next window (host-form throughput x1.4-1.6), every sample comes out one
    // window (10 ms at 2 MSPS) later, the last window's by csdr_demod_flush_*.
    bool deferred = false;
    int out_cur = 0;
    int pend_k = 0, pend_buf = 0; bool pend_stereo = false;
    hipEvent_t ev_out[2] = {nullptr, nullptr};
    int cur = 0;
    bool win_busy[2] = {false, false};
    hipEvent_t ev_win[2] = {nullptr, nullptr};   // window w's copy to the device has left the pinned buffer
    hipStream_t s = nullptr;
    int pos = 0;
    float *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    // stage taps (csdr_demod_set_taps): per pass either the callback or the per-profile accumulators
    csdr_tap_fn tap_fn = nullptr; void *tap_user = nullptr;
    std::vector<double> tap_acc[4];
    std::vector<float> tap_tmp;
    ~csdr_demod()
    {
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        for (hipEvent_t e : ev_win) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_out) if (e) (void)hipEventDestroy(e);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
    }
};

/* =================== batched device-resident form =================== */
struct csdr_demod_batch {
    int device, channels, fft_n;
    double in_rate = 0.0;
    std::vector<ChanCfg> cfg;
    std::vector<int> core_of, row_of;                 // channel -> (core, row)
    std::vector<int> in_row;                          // channel -> row of the caller's input it reads (csdr_demod_batch_set_input_rows)
    std::vector<ChainCore *> cores;                   // one per distinct decimator plan
    std::vector<std::vector<int>> members;            // core -> channel ids (row order)
    std::vector<int *> d_rows;                        // core -> device array of channel ids (input rows)
    std::vector<int *> d_out_rows;                    // core -> the same for the outputs, -1 = muted row (its receiver has
                                                      // moved to another plan group: csdr_demod_batch_set_demod)
    std::vector<std::vector<int>> row_in_last;        // core -> input row of each of its rows as last uploaded
    // the groups are independent: each runs on its own stream, forked from and joined to the caller's
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> joins;
    std::vector<int> order;                           // cores, heaviest post-chain first
    std::vector<hipEvent_t> dc_done;                  // core -> its down-converter has been issued and finished
    hipEvent_t fork = nullptr;
    bool pipelined = false;                           // csdr_demod_batch_set_pipelined
    bool chained = false;                             // ... its chained form (ChainCore::step_split): the cores stay plain
    std::vector<hipStream_t> post_streams;            // chained pipeline: core -> its post-chain's stream
    bool have_last_dc = false;                        // chained pipeline: dc_done[order.back()] holds the previous call's record
    int taps = 0;                                     // csdr_demod_batch_set_taps (new groups inherit it)
    bool rate_change_failed = false;                  // csdr_demod_batch_set_input_rate stopped half way: no processing until one succeeds
    std::vector<int> prev_post;                       // pipelined: per core, the post-chain event of the previous call
    std::vector<char> prev_join;                      // pipelined: per core, joins[] of the previous call not yet waited for
    float *d_blank = nullptr;                         // blanked input of process_packets (two-pass form)
    long raw_cap = 0;
    unsigned *d_mask = nullptr; long mask_cap = 0;    // the blanker's mask of process_packets (fused form): [channels][mask_cap] words
    DcBlank blank{};
    ~csdr_demod_batch()
    {
        for (auto *k : cores) delete k;
        for (auto *p : d_rows) if (p) (void)hipFree(p);
        for (auto *p : d_out_rows) if (p) (void)hipFree(p);
        if (d_blank) (void)hipFree(d_blank);
        if (d_mask) (void)hipFree(d_mask);
        for (auto st : streams) stream_pool().put(device, st);
        for (auto st : post_streams) stream_pool().put(device, st);
        for (auto ev : joins) (void)hipEventDestroy(ev);
        for (auto ev : dc_done) (void)hipEventDestroy(ev);
        if (fork) (void)hipEventDestroy(fork);
    }
};

// The group whose post-chain is the longest pole (FM: PLL + squelch filters, at the highest decimated rate) goes
// first: its down-converter should not share the chip with the other groups' while its demodulators wait.
static void batch_order(csdr_demod_batch *b)
{
    b->have_last_dc = false;                            // (the chained pipeline's link to the previous call's last group)
    std::vector<double> weight(b->cores.size(), 0.0);
    for (int c = 0; c < b->channels; c++) {
        if (b->core_of[c] < 0) continue;
        const int m = b->cfg[c].mode;
        const double w = (m == PC_MODE_FM ? 3.0 : m == PC_MODE_SAM ? 2.5 : m == PC_MODE_AM ? 1.5 : 1.0) * b->cfg[c].out_rate;
        weight[b->core_of[c]] = std::max(weight[b->core_of[c]], w);
    }
    b->order.resize(b->cores.size());
    for (size_t i = 0; i < b->order.size(); i++) b->order[i] = (int)i;
    std::stable_sort(b->order.begin(), b->order.end(), [&](int x, int y) { return weight[x] > weight[y]; });
}
// one stream and two events per plan group, the fork event: whatever is still missing
static int batch_plumbing(csdr_demod_batch *b)
{
    if (!b->fork) CSDR_HIP(hipEventCreateWithFlags(&b->fork, hipEventDisableTiming));
    int pr_lo = 0, pr_hi = 0;                          // numerically lower = higher priority
    CSDR_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
    while (b->streams.size() < b->cores.size()) {
        const size_t ki = b->streams.size();
        size_t rank = 0;
        while (rank < b->order.size() && b->order[rank] != (int)ki) rank++;
        int pr = pr_hi + (int)rank;
        if (pr > pr_lo) pr = pr_lo;
        hipStream_t st;
        CSDR_HIP(stream_pool().get(b->device, pr, &st));
        b->streams.push_back(st);
    }
    while (b->joins.size() < b->cores.size()) {
        hipEvent_t ev;
        CSDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); b->joins.push_back(ev);
    }
    while (b->dc_done.size() < b->cores.size()) {
        hipEvent_t ev;
        CSDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); b->dc_done.push_back(ev);
    }
    b->prev_post.resize(b->cores.size(), -1);
    b->prev_join.resize(b->cores.size(), 0);
    return CSDR_OK;
}

// a plan group's stream and events for group number `have` .. `want`-1 (the new groups get the lowest priority until
// batch_order / the next plumbing pass ranks them); nothing is published on failure
static int batch_plumbing_reserve(csdr_demod_batch *b, size_t want)
{
    if (!b->fork) CSDR_HIP(hipEventCreateWithFlags(&b->fork, hipEventDisableTiming));
    int pr_lo = 0, pr_hi = 0;
    CSDR_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
    while (b->streams.size() < want) {
        hipStream_t st;
        CSDR_HIP(stream_pool().get(b->device, pr_lo, &st));
        b->streams.push_back(st);
    }
    while (b->joins.size() < want) { hipEvent_t ev; CSDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); b->joins.push_back(ev); }
    while (b->dc_done.size() < want) { hipEvent_t ev; CSDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); b->dc_done.push_back(ev); }
    if (b->prev_post.size() < want) b->prev_post.resize(want, -1);
    if (b->prev_join.size() < want) b->prev_join.resize(want, 0);
    return CSDR_OK;
}

// a plan group none of whose rows has a receiver any more (all moved away) leaves the batch: no more launches for it
static void batch_drop_core(csdr_demod_batch *b, int ki)
{
    delete b->cores[ki];
    b->cores.erase(b->cores.begin() + ki);
    b->members.erase(b->members.begin() + ki);
    if ((size_t)ki < b->row_in_last.size()) b->row_in_last.erase(b->row_in_last.begin() + ki);
    if (b->d_rows[ki]) (void)hipFree(b->d_rows[ki]);
    if (b->d_out_rows[ki]) (void)hipFree(b->d_out_rows[ki]);
    b->d_rows.erase(b->d_rows.begin() + ki);
    b->d_out_rows.erase(b->d_out_rows.begin() + ki);
    if ((size_t)ki < b->streams.size()) { stream_pool().put(b->device, b->streams[ki]); b->streams.erase(b->streams.begin() + ki); }
    if ((size_t)ki < b->post_streams.size()) { stream_pool().put(b->device, b->post_streams[ki]); b->post_streams.erase(b->post_streams.begin() + ki); }
    b->have_last_dc = false;
    if ((size_t)ki < b->joins.size()) { (void)hipEventDestroy(b->joins[ki]); b->joins.erase(b->joins.begin() + ki); }
    if ((size_t)ki < b->dc_done.size()) { (void)hipEventDestroy(b->dc_done[ki]); b->dc_done.erase(b->dc_done.begin() + ki); }
    if ((size_t)ki < b->prev_post.size()) b->prev_post.erase(b->prev_post.begin() + ki);
    if ((size_t)ki < b->prev_join.size()) b->prev_join.erase(b->prev_join.begin() + ki);
    for (int &c : b->core_of) if (c > ki) c--;
}

/* CDemodulator::SetDemod with a NEW MODE whose decimator chain has another number of stages -- hence another output
 * rate, hop count and staging fill -- on a committed batch (dsp/demodulator.cpp:107-157).  The receiver leaves its plan
 * group (whose rows share one decimation) with everything the reference keeps across SetDemod: the down-converter's
 * oscillator, the filter's overlap AND its partly filled input (samples at the OLD rate: fastfir.cpp:278-285 never
 * resets m_InBufInPos), AGC and S-meter objects; the new demodulator starts fresh and the rebuilt decimator from zero
 * histories, as there.  It continues in a muted row of a group that already has the new decimation and the same
 * staging fill (a receiver that left earlier: retuning back and forth does not grow the batch), else in a group of
 * its own.  Its old row stays behind muted (no output, no S-meter; the group still filters it) and a group left with
 * muted rows only is dropped.  Transactional: the new row is complete before anything of the batch changes, and any
 * failure leaves the batch as it was.  A receiver already alone in its group changes in place, exactly like the
 * single-channel object. */
template <class Apply>                                // apply(core, row, cfg): what makes the receiver's chain the new one
static int batch_move_row(csdr_demod_batch *b, int channel, int new_stages, Apply apply)
{
    CSDR_HIP(hipDeviceSynchronize());                  // control plane: nothing of this batch in flight from here on
    const int ka = b->core_of[channel], r = b->row_of[channel];
    ChainCore &A = *b->cores[ka];
    if (A.rows == 1) return apply(A, 0, b->cfg[channel]);
    // ---- where to: a muted row of a group with the new decimation and the same staging fill, else a new group
    int kb = -1, rb = -1;
    for (size_t ki = 0; ki < b->cores.size() && kb < 0; ki++) {
        ChainCore &B = *b->cores[ki];
        if ((int)ki == ka || B.pending != A.pending || B.rows < 2) continue;
        // (the muted row's own chain is its group's: a muted row follows its group through every rate change, while row 0
        // may be a receiver that is itself about to leave)
        for (size_t q = 0; q < b->members[ki].size(); q++)
            if (b->members[ki][q] < 0 && csdr_downconvert_batch_out_count(B.dc, (int)q, 1 << 12) == (1 << 12) >> new_stages) {
                kb = (int)ki; rb = (int)q; break;
            }
    }
    ChainCore *S = nullptr;
    int *dr = nullptr, *dor = nullptr;
    int rc = CSDR_OK;
    auto hip = [&](hipError_t e) { if (e != hipSuccess && rc == CSDR_OK) rc = fail(CSDR_EHIP, "%s", hipGetErrorString(e)); return e == hipSuccess; };
    if (kb < 0) {
        S = new ChainCore();
        rc = S->init(b->device, 1, b->fft_n);
        if (rc == CSDR_OK && b->pipelined) rc = S->pipelined_init();
        if (rc == CSDR_OK) rc = S->ensure((long)A.pending + 1);
        if (rc == CSDR_OK) { hip(hipMalloc((void **)&dr, sizeof(int))) && hip(hipMalloc((void **)&dor, sizeof(int))); }
        if (rc == CSDR_OK) rc = batch_plumbing_reserve(b, b->cores.size() + 1);
    }
    ChainCore &T = kb < 0 ? *S : *b->cores[kb];
    const int tr = kb < 0 ? 0 : rb;
    ChanCfg cfg = b->cfg[channel];                     // committed only when everything has worked
    if (rc == CSDR_OK) rc = csdr__downconvert_batch_copy_channel(T.dc, tr, A.dc, r);
    if (rc == CSDR_OK) rc = csdr__fastfir_batch_copy_row(T.ff, tr, A.ff, r);
    if (rc == CSDR_OK) rc = T.pc.import_channel(tr, A.pc, r);
    if (rc == CSDR_OK && A.pending > 0) {
        const float *cur = (A.stage_cur ? A.d_stage2 : A.d_stage) + (size_t)r * A.cap * 2;
        float *dst = (T.stage_cur ? T.d_stage2 : T.d_stage) + (size_t)tr * T.cap * 2;
        hip(hipMemcpy(dst, cur, (size_t)A.pending * 8, hipMemcpyDeviceToDevice));
    }
    if (rc == CSDR_OK) rc = apply(T, tr, cfg);
    const int muted = -1;
    int *t_in = kb < 0 ? dr : b->d_rows[kb] + rb, *t_out = kb < 0 ? dor : b->d_out_rows[kb] + rb;
    if (rc == CSDR_OK) hip(hipMemcpy(t_in, &b->in_row[channel], sizeof(int), hipMemcpyHostToDevice));
    if (rc == CSDR_OK) hip(hipMemcpy(t_out, &channel, sizeof(int), hipMemcpyHostToDevice));
    if (rc == CSDR_OK) hip(hipMemcpy(b->d_out_rows[ka] + r, &muted, sizeof(int), hipMemcpyHostToDevice));
    if (rc != CSDR_OK) {                               // nothing published: the batch is as it was (a reused muted row
        delete S;                                      // holds copied state nobody reads)
        if (dr) (void)hipFree(dr);
        if (dor) (void)hipFree(dor);
        if (kb >= 0) (void)hipMemcpy(b->d_out_rows[kb] + rb, &muted, sizeof(int), hipMemcpyHostToDevice);
        return rc;
    }
    // ---- publish (host bookkeeping only from here on: cannot fail)
    b->cfg[channel] = cfg;
    b->members[ka][r] = -1;
    if (kb < 0) {
        S->pending = A.pending;
        b->cores.push_back(S);
        b->members.push_back(std::vector<int>(1, channel));
        b->d_rows.push_back(dr); b->d_out_rows.push_back(dor);
        b->row_in_last.resize(b->cores.size());
        b->row_in_last.back().assign(1, b->in_row[channel]);
        b->core_of[channel] = (int)b->cores.size() - 1; b->row_of[channel] = 0;
    } else {
        b->members[kb][rb] = channel;
        if ((size_t)kb < b->row_in_last.size() && (size_t)rb < b->row_in_last[kb].size()) b->row_in_last[kb][rb] = b->in_row[channel];
        b->core_of[channel] = kb; b->row_of[channel] = rb;
    }
    bool empty = true;
    for (int m : b->members[ka]) empty = empty && m < 0;
    if (empty) batch_drop_core(b, ka);
    batch_order(b);
    return CSDR_OK;
}
static int batch_move_channel(csdr_demod_batch *b, int channel, int mode, const DemodInfo &di, int new_stages)
{
    const double in_rate = b->in_rate;
    return batch_move_row(b, channel, new_stages, [&](ChainCore &k, int row, ChanCfg &cfg) {
        return apply_set_demod(k, row, cfg, in_rate, mode, di);
    });
}

// CDemodulator::SetInputSampleRate (dsp/demodulator.cpp:92-99) for row r of core k: the down-converter is rebuilt for
// the new input rate (CDownConvert::SetDataRate, downconvert.cpp:114-173: new stage list from zeroed histories, the
// oscillator keeps phase and amplitude, the CW offset is added once more, :169), m_OutputRate follows -- and nothing else:
// filter taps and overlap, AGC constants and rings and the demodulator object stay as they are until the next SetDemod
// (which, for the same mode, keeps the demodulator built for the OLD output rate: ChanCfg::demod_rate).  The S-meter is
// handed m_OutputRate with every pass (demodulator.cpp:183), so its time constants follow at once.
static int apply_input_rate(ChainCore &k, int r, ChanCfg &c, double rate)
{
    const double out = csdr_downconvert_batch_set_data_rate(k.dc, r, rate, c.want_bw);
    if (out < 0) return CSDR_EHIP;
    c.out_rate = out;
    return k.pc.smeter_rate_set(r, out);
}

extern "C" {

csdr_demod *csdr_demod_create(int device, int fastfir_n)
{
    if (!device_ok(device)) return nullptr;
    csdr_demod *d = new csdr_demod();
    if (d->k.init(device, 1, fastfir_n) != CSDR_OK) { delete d; return nullptr; }
    bool ok = hipStreamCreateWithFlags(&d->s, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : d->ev_win) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto &e : d->ev_out) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { fail(CSDR_EHIP, "stream / event creation failed"); delete d; return nullptr; }
    csdr_downconvert_batch_set_cw_offset(d->k.dc, 0, 0.0);      // ctor: SetDemodFreq(0.0)
    csdr_downconvert_batch_set_frequency(d->k.dc, 0, 0.0);
    d->zero_copy = !(getenv("CSDR_HOST_ZEROCOPY") && atoi(getenv("CSDR_HOST_ZEROCOPY")) == 0);
    return d;
}
void csdr_demod_destroy(csdr_demod *d) { delete d; }

/* CDemodulator::SetInputSampleRate (demodulator.cpp:92-99) */
int csdr_demod_set_input_rate(csdr_demod *d, double rate)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    if (d->s) CSDR_HIP(hipStreamSynchronize(d->s));   // control plane: a pass that returned no samples may still be in flight
    if (!(rate > 0.0) || !std::isfinite(rate)) return fail(CSDR_EINVAL, "input rate %g", rate);
    if (d->in_rate != rate) {
        const double r = csdr_downconvert_batch_set_data_rate(d->k.dc, 0, rate, d->c.want_bw);
        if (r < 0) return CSDR_EHIP;                      // (in_rate keeps its old value: the same call again is not a no-op)
        d->in_rate = rate;
        d->c.out_rate = r;
        // Everything else stays as it is until the next SetDemod: filter taps, AGC constants and rings, m_InBufLimit and
        // the demodulator object (built for the OLD output rate; a same-mode SetDemod does not rebuild it,
        // demodulator.cpp:111-137).  Only the S-meter follows at once: it is handed m_OutputRate with every pass (:183)
        int rc = d->k.pc.smeter_rate_set(0, r);
        if (rc) return rc;
    }
    return CSDR_OK;
}
/* CDemodulator::SetDemod (demodulator.cpp:107-157) */
int csdr_demod_set_demod(csdr_demod *d, int mode, const csdr_demod_info *info)
{
    if (!d || !info || mode < 0 || mode > 6) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(d->k.device)) return CSDR_EHIP;
    if (d->s) CSDR_HIP(hipStreamSynchronize(d->s));   // control plane: a pass that returned no samples may still be in flight
    DemodInfo di;
    memcpy(&di, info, sizeof(di));
    int rc = apply_set_demod(d->k, 0, d->c, d->in_rate, mode, di);
    if (rc) return rc;
    d->limit = (int)((d->c.out_rate / 100.0) * d->in_rate / d->c.out_rate);
    d->limit &= 0xFFFFFF00;
    return CSDR_OK;
}
/* CDemodulator::SetDemodFreq (demodulator.h:68-69) */
int csdr_demod_set_freq(csdr_demod *d, double freq)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    csdr_downconvert_batch_set_cw_offset(d->k.dc, 0, d->c.cw_off);
    return csdr_downconvert_batch_set_frequency(d->k.dc, 0, freq);
}
double csdr_demod_get_output_rate(csdr_demod *d) { return d ? d->c.out_rate : 0.0; }
double csdr_demod_get_smeter_peak(csdr_demod *d) { return d ? d->k.pc.smeter_peak(0) : 0.0; }
double csdr_demod_get_smeter_ave(csdr_demod *d) { return d ? d->k.pc.smeter_ave(0) : 0.0; }
int csdr_demod_get_buf_limit(csdr_demod *d) { return d ? d->limit : fail(CSDR_EINVAL, "bad handle"); }
/* the chain's test points (dsp/demodulator.cpp:175,180,187,208): see include/cutesdr_mi.h */
int csdr_demod_set_taps(csdr_demod *d, int mask, csdr_tap_fn fn, void *user)
{
    if (d && mask && d->deferred) return fail(CSDR_ESTATE, "stage taps and deferred output exclude each other");
    if (!d || mask < 0 || mask > 15) return fail(CSDR_EINVAL, "bad argument");
    if (d->s) CSDR_HIP(hipStreamSynchronize(d->s));
    d->k.taps = mask; d->tap_fn = fn; d->tap_user = user;
    for (auto &a : d->tap_acc) a.clear();
    return CSDR_OK;
}
int csdr_demod_get_tap(csdr_demod *d, int profile, double *out, int cap)
{
    if (!d || profile < 1 || profile > 4 || cap < 0 || (cap && !out)) return fail(CSDR_EINVAL, "bad argument");
    std::vector<double> &a = d->tap_acc[profile - 1];
    if ((size_t)cap < a.size()) return fail(CSDR_EINVAL, "tap %d holds %zu doubles, room for %d", profile, a.size(), cap);
    const int n = (int)a.size();
    if (n) memcpy(out, a.data(), sizeof(double) * (size_t)n);
    a.clear();
    return n;
}

/* CDemodulator::ProcessData (demodulator.cpp:163-215 mono, :221-273 stereo).  Every inner pass
 * writes its output at out[0] and the return value is the SUM over the passes, exactly as the
 * reference does; append != 0 selects the batch-harness form that appends. */
// the stage taps of the pass that has just been issued (csdr_demod_set_taps): waits for it, then per profile the callback or
// the accumulator, in the reference's order (demodulator.cpp:175,180,187,208)
static int demod_emit_taps(csdr_demod *d, int k, bool stereo)
{
    ChainCore &c = d->k;
    CSDR_HIP(hipStreamSynchronize(d->s));
    auto emit = [&](int profile, const float *dev_or_host, bool on_device, int n, bool cpx) -> int {
        if (!(c.taps & (1 << (profile - 1)))) return CSDR_OK;
        const size_t nf = (size_t)n * (cpx ? 2 : 1);
        d->tap_tmp.resize(nf ? nf : 1);
        if (nf) {
            if (on_device) CSDR_HIP(hipMemcpy(d->tap_tmp.data(), dev_or_host, nf * 4, hipMemcpyDeviceToHost));
            else memcpy(d->tap_tmp.data(), dev_or_host, nf * 4);
        }
        if (d->tap_fn) {
            std::vector<double> v(nf ? nf : 1);
            for (size_t i = 0; i < nf; i++) v[i] = (double)d->tap_tmp[i];
            d->tap_fn(d->tap_user, profile, n, v.data(), cpx ? 1 : 0, d->c.out_rate);
        } else {
            std::vector<double> &a = d->tap_acc[profile - 1];
            for (size_t i = 0; i < nf; i++) a.push_back((double)d->tap_tmp[i]);
        }
        return CSDR_OK;
    };
    int rc;
    if ((rc = emit(1, c.d_tap1, true, c.tap1_n, true))) return rc;
    if ((rc = emit(2, c.d_filt, true, k, true))) return rc;
    if ((rc = emit(3, c.d_agc, true, k, true))) return rc;
    const float *audio = d->zero_copy ? d->pin_out.p : d->d_out;
    return emit(4, audio, !d->zero_copy, k, stereo);
}
// deferred mode: the pending pass's samples (if any) to `out`; returns their count and clears the slot
static int demod_take_pending(csdr_demod *d, double *out)
{
    const int k = d->pend_k;
    if (k <= 0) return 0;
    CSDR_HIP(hipEventSynchronize(d->ev_out[d->pend_buf]));
    const PinnedBuf &pb = d->pend_buf ? d->pin_out2 : d->pin_out;
    cvt_to_f64(out, pb.p, d->pend_stereo ? 2 * (size_t)k : (size_t)k);
    d->pend_k = 0;
    return k;
}
static int demod_process(csdr_demod *d, int n, const double *in_iq, double *out, bool stereo, bool append)
{
    if (!d || n < 0 || (n && (!in_iq || !out))) return fail(CSDR_EINVAL, "bad argument");
    if (d->limit <= 0 || d->limit > refc::DEMOD_MAX_INBUFSIZE) return fail(CSDR_ESTATE, "input buffer limit %d out of range", d->limit);
    if (!device_ok(d->k.device)) return CSDR_EHIP;
    int ret = 0;
    for (int i = 0; i < n; ) {
        PinnedBuf &w = d->win[d->cur];
        // the samples that fit before the window is full (at least one: a limit lowered by SetDemod below the fill
        // runs the pass at the next sample, as `if (m_InBufPos >= m_InBufLimit)` does, demodulator.cpp:169-174)
        int take = d->limit - d->pos;
        if (take < 1) take = 1;
        if (take > n - i) take = n - i;
        if (d->win_busy[d->cur]) { CSDR_HIP(hipEventSynchronize(d->ev_win[d->cur])); d->win_busy[d->cur] = false; }
        // (a whole window at once: grown step by step, a window filled in 256-sample calls was reallocated -- pinned
        // malloc, copy, free -- some ten times during its first fill, on the per-datagram path)
        const size_t fill = (size_t)d->pos + take;
        const size_t want = 2 * (fill > (size_t)d->limit ? fill : (size_t)d->limit);
        if (want > w.cap && d->zero_copy) CSDR_HIP(hipStreamSynchronize(d->s));   // nothing may still be reading the old buffer
        int rc = w.reserve(want);
        if (rc) return rc;
        cvt_to_f32(w.p + 2 * (size_t)d->pos, in_iq + 2 * (size_t)i, 2 * (size_t)take);
        d->pos += take; i += take;
        if (d->pos < d->limit) break;                     // the call's samples are in; the window is not full yet
        const int len = d->pos;
        d->pos = 0;
        const size_t need_out = (size_t)len + d->k.L;
        const float *chain_in = nullptr;
        float *chain_out = nullptr;
        size_t out_stride = 0;
        // (deferred: the two output buffers in turn -- the one written now last held the pass before the pending one)
        const int ob = d->deferred ? d->out_cur : 0;
        PinnedBuf &pout = ob ? d->pin_out2 : d->pin_out;
        if (d->zero_copy) {
            if (2 * need_out > pout.cap) {
                CSDR_HIP(hipStreamSynchronize(d->s));
                if ((rc = pout.reserve(2 * need_out))) return rc;
            }
            void *pi = nullptr, *po = nullptr;
            CSDR_HIP(hipHostGetDevicePointer(&pi, w.p, 0));
            CSDR_HIP(hipHostGetDevicePointer(&po, pout.p, 0));
            chain_in = (const float *)pi; chain_out = (float *)po; out_stride = pout.cap / 2;
        } else {
            if ((size_t)len > d->cap_in) {
                CSDR_HIP(hipStreamSynchronize(d->s));
                if (d->d_in) (void)hipFree(d->d_in);
                d->d_in = nullptr; d->cap_in = 0;
                CSDR_HIP(hipMalloc((void **)&d->d_in, (size_t)len * 8));
                d->cap_in = len;
            }
            if (need_out > d->cap_out) {
                CSDR_HIP(hipStreamSynchronize(d->s));
                if (d->d_out) (void)hipFree(d->d_out);
                d->d_out = nullptr; d->cap_out = 0;
                CSDR_HIP(hipMalloc((void **)&d->d_out, need_out * 8));
                d->cap_out = need_out;
            }
            // window -> device -> chain, all on the object's stream (d_in is reused in stream order)
            CSDR_HIP(hipMemcpyAsync(d->d_in, w.p, (size_t)len * 8, hipMemcpyHostToDevice, d->s));
            chain_in = d->d_in; chain_out = d->d_out; out_stride = d->cap_out;
        }
        const int wcur = d->cur;
        d->cur ^= 1;
        const int k = d->k.step(chain_in, len, nullptr, len, chain_out, (long)out_stride, nullptr, stereo, d->s);
        // the window is free again when the down-converter (zero copy) / the copy has read it: the whole pass, here
        CSDR_HIP(hipEventRecord(d->ev_win[wcur], d->s));
        d->win_busy[wcur] = true;
        if (k < 0) return k;
        if (d->k.taps) { const int rct = demod_emit_taps(d, k, stereo); if (rct) return rct; }
        if (k > 0 && d->deferred) {
            // this pass's samples stay where they are; the pending pass's -- complete, or nearly -- are handed over
            const size_t nf = stereo ? 2 * (size_t)k : (size_t)k;
            if (!d->zero_copy) {
                if (nf > pout.cap) { CSDR_HIP(hipStreamSynchronize(d->s)); if ((rc = pout.reserve(nf))) return rc; }
                CSDR_HIP(hipMemcpyAsync(pout.p, d->d_out, nf * 4, hipMemcpyDeviceToHost, d->s));
            }
            CSDR_HIP(hipEventRecord(d->ev_out[ob], d->s));
            const int got = demod_take_pending(d, append ? out + (d->pend_stereo ? 2 : 1) * (size_t)ret : out);
            if (got < 0) return got;
            d->pend_k = k; d->pend_buf = ob; d->pend_stereo = stereo;
            d->out_cur ^= 1;
            ret += got;
            continue;
        }
        if (k > 0) {                                      // a pass that returns samples: the one wait of this call
            const size_t nf = stereo ? 2 * (size_t)k : (size_t)k;
            if (!d->zero_copy) {
                if ((rc = d->pin_out.reserve(nf))) return rc;
                CSDR_HIP(hipMemcpyAsync(d->pin_out.p, d->d_out, nf * 4, hipMemcpyDeviceToHost, d->s));
            }
            CSDR_HIP(hipStreamSynchronize(d->s));             // (polling hipStreamQuery first: measured +-0)
            cvt_to_f64(append ? out + (stereo ? 2 : 1) * (size_t)ret : out, d->pin_out.p, nf);
        }
        ret += k;
    }
    return ret;
}
int csdr_demod_process_mono(csdr_demod *d, int n, const double *in_iq, double *out)
{ return demod_process(d, n, in_iq, out, false, false); }
int csdr_demod_process_stereo(csdr_demod *d, int n, const double *in_iq, double *out_iq)
{ return demod_process(d, n, in_iq, out_iq, true, false); }
int csdr_demod_process_mono_append(csdr_demod *d, int n, const double *in_iq, double *out)
{ return demod_process(d, n, in_iq, out, false, true); }
int csdr_demod_set_deferred(csdr_demod *d, int on)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    if (on && d->k.taps) return fail(CSDR_ESTATE, "stage taps and deferred output exclude each other");
    if (!on && d->pend_k > 0) return fail(CSDR_ESTATE, "a pass is pending: csdr_demod_flush first");
    d->deferred = on != 0;
    return CSDR_OK;
}
int csdr_demod_flush(csdr_demod *d, double *out, int cap)
{
    if (!d || cap < 0 || (cap && !out)) return fail(CSDR_EINVAL, "bad argument");
    const int need = d->pend_stereo ? 2 * d->pend_k : d->pend_k;
    if (need > cap) return fail(CSDR_EINVAL, "the pending pass holds %d values, room for %d", need, cap);
    if (!device_ok(d->k.device)) return CSDR_EHIP;
    return demod_take_pending(d, out);
}

/* internal (bench.py `host_form`, tests; not in the public header): the reference's call pattern in one C loop --
 * n_total samples handed over in calls of call_len (one datagram: 240 / 256 samples, interface/sdrinterface.cpp:903),
 * every call's audio appended at out.  Returns the audio samples produced.  Saves the measurement the per-call cost
 * of the Python binding, nothing else. */
int csdr__demod_process_calls(csdr_demod *d, int n_total, int call_len, const double *in_iq, double *out)
{
    if (!d || call_len < 1 || n_total < 0 || (n_total && (!in_iq || !out))) return fail(CSDR_EINVAL, "bad argument");
    int total = 0;
    for (int i = 0; i < n_total; i += call_len) {
        const int n = n_total - i < call_len ? n_total - i : call_len;
        const int k = demod_process(d, n, in_iq + 2 * (size_t)i, out + total, false, true);
        if (k < 0) return k;
        total += k;
    }
    return total;
}

/* ------------------------------- batch ------------------------------- */
csdr_demod_batch *csdr_demod_batch_create(int device, int channels, int fastfir_n)
{
    if (channels < 1) { fail(CSDR_EINVAL, "channels >= 1"); return nullptr; }
    if (!device_ok(device)) return nullptr;
    csdr_demod_batch *b = new csdr_demod_batch();
    b->device = device; b->channels = channels; b->fft_n = fastfir_n;
    b->cfg.assign(channels, ChanCfg());
    b->core_of.assign(channels, -1); b->row_of.assign(channels, -1);
    b->in_row.resize(channels);
    for (int c = 0; c < channels; c++) b->in_row[c] = c;
    return b;
}
void csdr_demod_batch_destroy(csdr_demod_batch *b) { delete b; }
/* CDemodulator::SetInputSampleRate (dsp/demodulator.cpp:92-99) for every receiver of the batch, at any time -- the host
 * calls it on every bandwidth switch of the radio (interface/sdrinterface.cpp:753-754).  Before the commit it only
 * records the rate.  On a committed batch every receiver's down-converter is rebuilt for the new rate with what the
 * reference keeps (apply_input_rate above); receivers stay in their rows as long as the rows of a plan group still share
 * one decimation (they do whenever they share one bandwidth limit, which is how the commit groups them); a receiver whose
 * new chain has another number of stages than its group's leaves for a matching muted row or a group of its own, exactly
 * as after a mode change (batch_move_row).  Control plane: synchronises the device; a failure in the planning phase
 * leaves the batch as it was. */
int csdr_demod_batch_set_input_rate(csdr_demod_batch *b, double rate)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (!(rate > 0.0) || !std::isfinite(rate)) return fail(CSDR_EINVAL, "input rate %g", rate);     // before anything records it
    if (b->cores.empty() || (rate == b->in_rate && !b->rate_change_failed)) { b->in_rate = rate; return CSDR_OK; }
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());                  // control plane: nothing of this batch in flight from here on
    // ---- plan (nothing changes yet): every receiver's new stage count, every group's (its first live row's)
    std::vector<int> stages(b->channels, -1);
    for (int c = 0; c < b->channels; c++) {
        if (b->core_of[c] < 0) continue;
        const DcPlan p = dc_make_plan(rate, b->cfg[c].want_bw);
        if (p.nstages < 0 || p.nstages > DC_MAX_STAGES) return fail(CSDR_EINVAL, "no decimator chain for rate %g", rate);
        stages[c] = p.nstages;
    }
    // (a group keeps the decimation MOST of its live rows get -- the first of them on a tie -- so that as few receivers as
    // possible have to move; rows grouped at the commit share one bandwidth limit and all agree)
    std::vector<int> group_stages(b->cores.size(), -1);
    std::vector<double> group_bw(b->cores.size(), 0.0);
    for (size_t ki = 0; ki < b->cores.size(); ki++) {
        int votes[DC_MAX_STAGES + 1] = {0}, best = -1;
        for (int c : b->members[ki]) if (c >= 0) votes[stages[c]]++;
        for (int c : b->members[ki]) if (c >= 0 && (best < 0 || votes[stages[c]] > votes[best])) best = stages[c];
        group_stages[ki] = best;
        for (int c : b->members[ki]) if (c >= 0 && stages[c] == best) { group_bw[ki] = b->cfg[c].want_bw; break; }
    }
    // ---- the rows that keep their group: in place (a muted row follows its group, it only has to decimate alike)
    std::vector<int> movers;
    for (size_t ki = 0; ki < b->cores.size(); ki++) {
        ChainCore &k = *b->cores[ki];
        for (size_t q = 0; q < b->members[ki].size(); q++) {
            const int c = b->members[ki][q];
            if (c >= 0 && stages[c] != group_stages[ki]) { movers.push_back(c); continue; }
            // (from the first row that has taken the new rate a failure leaves the groups' rows on DIFFERENT decimations while
            // the staging is sized from row 0: the batch refuses to process until a set_input_rate has gone through -- the
            // same call again finishes the job, every step above is idempotent)
            if (c >= 0) { const int rc = apply_input_rate(k, (int)q, b->cfg[c], rate); if (rc) { b->rate_change_failed = true; return rc; } }
            else if (group_stages[ki] >= 0 && csdr_downconvert_batch_set_data_rate(k.dc, (int)q, rate, group_bw[ki]) < 0) { b->rate_change_failed = true; return CSDR_EHIP; }
        }
    }
    b->in_rate = rate;
    b->rate_change_failed = false;
    // ---- the others move, with all their state, like a receiver whose new mode decimates differently; the row each
    // leaves behind is muted and takes its old group's new chain
    int err = CSDR_OK;
    std::map<ChainCore *, double> bw_of;               // (group indices shift when a move empties a group)
    for (size_t ki = 0; ki < b->cores.size(); ki++) bw_of[b->cores[ki]] = group_bw[ki];
    for (int c : movers) {
        const int ka = b->core_of[c], r = b->row_of[c];
        ChainCore *A = b->cores[ka];
        const bool alone = A->rows == 1;
        const double bw = bw_of[A];
        const int rc = batch_move_row(b, c, stages[c], [&](ChainCore &k, int row, ChanCfg &cfg) { return apply_input_rate(k, row, cfg, rate); });
        if (rc) { if (!err) err = rc; continue; }
        // (batch_move_row may have dropped group ka -- then A is gone; it drops a group only when every row is muted)
        bool still = false;
        for (auto *k : b->cores) still = still || k == A;
        if (!alone && still && csdr_downconvert_batch_set_data_rate(A->dc, r, rate, bw) < 0 && !err) err = CSDR_EHIP;
    }
    batch_order(b);
    if (err) b->rate_change_failed = true;               // a receiver that should have moved did not: see above
    return err;
}
/* Configure every channel, then call csdr_demod_batch_commit() once: channels that decimate by
 * the same chain are grouped and run together. */
int csdr_demod_batch_set_demod(csdr_demod_batch *b, int channel, int mode, const csdr_demod_info *info)
{
    if (!b || !info || channel < 0 || channel >= b->channels || mode < 0 || mode > 6)
        return fail(CSDR_EINVAL, "bad argument");
    ChanCfg &c = b->cfg[channel];
    if (b->core_of[channel] >= 0) {
        // already committed.  The reference rebuilds the down-converter only when the MODE changes
        // (demodulator.cpp:111-121); a new mode whose chain has as many stages as the old one stays in its row (the
        // down-converter object holds a plan per row), one with another decimation moves (batch_move_channel)
        DemodInfo di; memcpy(&di, info, sizeof(di));
        if (!device_ok(b->device)) return CSDR_EHIP;
        ChainCore &k = *b->cores[b->core_of[channel]];
        if (c.mode != mode) {
            const double bw = (mode == PC_MODE_LSB || mode == PC_MODE_CWL) ? -di.LowCutmin : di.HiCutmax;
            const int new_stages = dc_make_plan(b->in_rate, bw).nstages;
            int codes[DC_MAX_STAGES];
            const int old_stages = csdr_downconvert_batch_get_stages(k.dc, b->row_of[channel], codes, DC_MAX_STAGES);
            if (new_stages != old_stages) return batch_move_channel(b, channel, mode, di, new_stages);
        }
        return apply_set_demod(k, b->row_of[channel], c, b->in_rate, mode, di);
    }
    memcpy(&c.info, info, sizeof(DemodInfo));
    c.pending = mode;                    // applied at commit
    c.want_bw = (mode == PC_MODE_LSB || mode == PC_MODE_CWL) ? -c.info.LowCutmin : c.info.HiCutmax;
    return CSDR_OK;
}
// device copies of every group's input-row list, from members[] and in_row[]
static int batch_upload_input_rows(csdr_demod_batch *b)
{
    std::vector<int> rows;
    b->row_in_last.resize(b->cores.size());
    for (size_t ki = 0; ki < b->cores.size(); ki++) {
        b->row_in_last[ki].resize(b->members[ki].size(), 0);
        rows.clear();
        for (size_t q = 0; q < b->members[ki].size(); q++) {   // (a muted row keeps reading the row it last had)
            const int c = b->members[ki][q];
            rows.push_back(c >= 0 ? b->in_row[c] : b->row_in_last[ki][q]);
        }
        b->row_in_last[ki] = rows;
        CSDR_HIP(hipMemcpy(b->d_rows[ki], rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice));
    }
    return CSDR_OK;
}
int csdr_demod_batch_set_input_rows(csdr_demod_batch *b, const int *input_row)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (input_row)
        for (int c = 0; c < b->channels; c++)
            if (input_row[c] < 0 || input_row[c] >= b->channels) return fail(CSDR_EINVAL, "input row %d of receiver %d", input_row[c], c);
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());                  // control plane: a call in flight still reads the old lists
    for (int c = 0; c < b->channels; c++) b->in_row[c] = input_row ? input_row[c] : c;
    return b->cores.empty() ? CSDR_OK : batch_upload_input_rows(b);
}
int csdr_demod_batch_commit(csdr_demod_batch *b)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(b->device)) return CSDR_EHIP;
    if (!b->cores.empty()) return fail(CSDR_ESTATE, "already committed");
    std::map<long long, std::vector<int>> groups;
    for (int c = 0; c < b->channels; c++) {
        if (b->cfg[c].pending < 0)
            return fail(CSDR_ESTATE, "channel %d has no demodulator configured", c);
        groups[(long long)llround(b->cfg[c].want_bw * 1000.0)].push_back(c);
    }
    for (auto &g : groups) {
        ChainCore *k = new ChainCore();
        if (k->init(b->device, (int)g.second.size(), b->fft_n) != CSDR_OK) { delete k; return CSDR_EHIP; }
        const int ki = (int)b->cores.size();
        b->cores.push_back(k);
        b->members.push_back(g.second);
        int *dr = nullptr;
        CSDR_HIP(hipMalloc((void **)&dr, sizeof(int) * g.second.size()));
        b->d_rows.push_back(dr);                       // filled by batch_upload_input_rows below
        int *dor = nullptr;
        CSDR_HIP(hipMalloc((void **)&dor, sizeof(int) * g.second.size()));
        CSDR_HIP(hipMemcpy(dor, g.second.data(), sizeof(int) * g.second.size(), hipMemcpyHostToDevice));
        b->d_out_rows.push_back(dor);
        for (size_t r = 0; r < g.second.size(); r++) {
            const int c = g.second[r];
            b->core_of[c] = ki; b->row_of[c] = (int)r;
            const int mode = b->cfg[c].pending;
            DemodInfo di = b->cfg[c].info;
            csdr_downconvert_batch_set_frequency(k->dc, (int)r, 0.0);
            int rc = apply_set_demod(*k, (int)r, b->cfg[c], b->in_rate, mode, di);
            if (rc) return rc;
        }
    }
    {
        int rc = batch_upload_input_rows(b);
        if (rc) return rc;
    }
    batch_order(b);                                    // heaviest post-chain first, and on the highest-priority stream
    if (b->cores.size() > 1) {
        int rc = batch_plumbing(b);
        if (rc) return rc;
    }
    return CSDR_OK;
}
/* Pipelined mode.  on != 0: a process call only enqueues on internal streams; in the caller's stream order the
 * INPUT buffer of call k has been consumed and the OUTPUT rows of call k-1 are complete after process call k+1
 * (everything after csdr_demod_batch_flush).  Results are identical to the strict mode. */
int csdr_demod_batch_set_pipelined(csdr_demod_batch *b, int on)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    if (on) {                                          // a single plan group normally runs on the caller's stream
        int rc = batch_plumbing(b);
        if (rc) return rc;
    }
    // Two forms.  CHAINED (round 6, the default; on == 2 asks for it by name): the strict mode's schedule -- one
    // down-converter at a time, each group's filter in queue order behind it -- carried across calls: the first group's next
    // down-converter follows the last group's, the post-chains run in streams of their own and the caller joins a call behind
    // the next call's launches.  Two streams per group, no second staging buffer.  1.65-1.68 ms per call of the C4 share, the
    // strict mode's period, against 1.75-1.80 for THREE-STAGE (rounds 3-5; on == 3 or CSDR_PIPE_KIND=3): every group's
    // down-converter at once, filter and post-chain on two more streams per group over double buffers.  A batch that has ever
    // run the three-stage form keeps its cores' extra streams and stays with it.
    static const int kind_env = getenv("CSDR_PIPE_KIND") ? atoi(getenv("CSDR_PIPE_KIND")) : 0;
    bool plain = true;
    for (auto *k : b->cores) plain = plain && !k->s_post;
    const bool chained = on && plain && on != 3 && (on == 2 || kind_env != 3);
    if (on && !chained) for (auto *k : b->cores) { int rc = k->pipelined_init(); if (rc) return rc; }
    b->prev_post.assign(b->cores.size(), -1);
    b->prev_join.assign(b->cores.size(), 0);
    for (auto *k : b->cores) { k->post_busy2[0] = k->post_busy2[1] = false; }
    b->have_last_dc = false;
    b->pipelined = on != 0;
    b->chained = chained;
    return CSDR_OK;
}
/* stream-orders the caller's stream behind everything the batch has in flight (pipelined mode: the post-chain
 * of the last call) */
int csdr_demod_batch_flush(csdr_demod_batch *b, void *stream)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(b->device)) return CSDR_EHIP;
    for (size_t ki = 0; ki < b->prev_post.size(); ki++) {
        if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, b->joins[ki], 0)); b->prev_join[ki] = 0; }
        if (b->prev_post[ki] >= 0) {
            CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, b->cores[ki]->ev_post[b->prev_post[ki]], 0));
            b->prev_post[ki] = -1;
        }
    }
    return CSDR_OK;
}
/* internal (csdr_demod_shard_process_shared): orders `stream` behind the batch's reads of the INPUT of its previous
 * call.  Strict mode: nothing to do (a process call joins the caller's stream itself).  Pipelined mode: the previous
 * call's down-converters run on the batch's own streams and the caller's stream joins them only inside the NEXT process
 * call -- too late for a caller that refills the input buffer on that stream first. */
int csdr__demod_batch_wait_input_free(csdr_demod_batch *b, void *stream)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (!b->pipelined) return CSDR_OK;
    if (!device_ok(b->device)) return CSDR_EHIP;
    for (size_t ki = 0; ki < b->cores.size() && ki < b->prev_join.size(); ki++)
        if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, b->joins[ki], 0)); b->prev_join[ki] = 0; }
    return CSDR_OK;
}
int csdr_demod_batch_set_freq(csdr_demod_batch *b, int channel, double freq)
{
    if (!b || channel < 0 || channel >= b->channels) return fail(CSDR_EINVAL, "bad argument");
    if (b->core_of[channel] < 0) return fail(CSDR_ESTATE, "commit first");
    ChainCore &k = *b->cores[b->core_of[channel]];
    csdr_downconvert_batch_set_cw_offset(k.dc, b->row_of[channel], b->cfg[channel].cw_off);
    return csdr_downconvert_batch_set_frequency(k.dc, b->row_of[channel], freq);
}
double csdr_demod_batch_get_output_rate(csdr_demod_batch *b, int channel)
{
    if (!b || channel < 0 || channel >= b->channels) return 0.0;
    return b->cfg[channel].out_rate;
}
double csdr_demod_batch_get_smeter_ave(csdr_demod_batch *b, int channel)
{
    if (!b || channel < 0 || channel >= b->channels || b->core_of[channel] < 0) return 0.0;
    return b->cores[b->core_of[channel]]->pc.smeter_ave(b->row_of[channel]);
}
extern "C" int csdr_demod_batch_flush(csdr_demod_batch *b, void *stream);
/* CSMeter::GetAve / GetPeak of every channel into device arrays indexed by channel (either may be NULL);
 * reading the peak resets it, as GetPeak does (smeter.cpp:98-103).  Asynchronous on `stream`. */
int csdr_demod_batch_get_smeter_all(csdr_demod_batch *b, float *d_ave, float *d_peak, void *stream)
{
    if (!b || (!d_ave && !d_peak)) return fail(CSDR_EINVAL, "bad argument");
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (!device_ok(b->device)) return CSDR_EHIP;
    int rcf = csdr_demod_batch_flush(b, stream);         // pipelined mode: behind the last call's post-chain
    if (rcf) return rcf;
    for (size_t ki = 0; ki < b->cores.size(); ki++)
        CSDR_HIP(smeter_collect_launch(b->cores[ki]->pc.d_chan, b->cores[ki]->rows, b->d_out_rows[ki], d_ave, d_peak,
                                       (hipStream_t)stream));
    return CSDR_OK;
}
/* d_in: [channels][in_stride] complex fp32; d_out: [channels][out_stride] fp32 mono audio.
 * Chunking: one call = one pass of the chain over n_per_channel samples (the host form uses
 * m_InBufLimit-sized passes; decimator, filter and post-chain do not depend on the chunking, word for word, for calls
 * of whole 512-sample tiles -- the decimator re-anchors its oscillator on an absolute grid --, the squelch
 * decision is taken once per FastFIR hop either way).  Asynchronous. */
static int demod_batch_run(csdr_demod_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                           float *d_out, long long out_stride, void *stream, bool stereo,
                           const void *d_packets = nullptr, int pkt_len = 0, const DcBlank *blank = nullptr)
{
    if (!b || (!d_in && !d_packets) || !d_out) return fail(CSDR_EINVAL, "bad argument");
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (b->rate_change_failed)
        return fail(CSDR_ESTATE, "a csdr_demod_batch_set_input_rate failed half way (rows of one group decimate differently): call it again");
    if (!device_ok(b->device)) return CSDR_EHIP;
    hipStream_t caller = (hipStream_t)stream;
    const bool forked = b->cores.size() > 1 || b->pipelined;
    if (forked) CSDR_HIP(hipEventRecord(b->fork, caller));
    int err = 0;
    // CSDR_CHAIN_PHASED=1 (measured in round 4, not the default): strict mode in two phases -- every group's
    // down-converter first, concurrently, so that together they fill the chip like ONE launch with nothing else
    // resident (1.05 ms for the C4 share), then every group's filter and post-chain (0.9 ms: the walks are the long
    // pole whatever runs beside them).  2.05 ms against 1.82 interleaved, where group g's post-chain runs beside group
    // g+1's down-converter: the walks' latency has to be overlapped with something, not queued behind everything.
    // 2 = phased with the down-converters chained one after the other (2.11 ms).
    static const int phased = getenv("CSDR_CHAIN_PHASED") ? atoi(getenv("CSDR_CHAIN_PHASED")) : 0;
    bool plain = true;                                   // (an object that was ever pipelined keeps its three-stream cores)
    for (auto *k : b->cores) plain = plain && !k->s_post;
    if (forked && !b->pipelined && phased && plain && b->cores.size() > 1) {
        std::vector<char> ok(b->cores.size(), 0);
        for (size_t oi = 0; oi < b->cores.size(); oi++) {
            const size_t ki = (size_t)b->order[oi];
            ChainCore &k = *b->cores[ki];
            k.pk = d_packets; k.pk_len = pkt_len; k.blank = blank;
            CSDR_HIP(hipStreamWaitEvent(b->streams[ki], b->fork, 0));
            const int rc = k.step_dc(d_in, in_stride, b->d_rows[ki], n_per_channel, b->streams[ki],
                                     phased == 2 && oi > 0 ? b->dc_done[b->order[oi - 1]] : nullptr, b->dc_done[ki]);
            if (rc < 0) { if (!err) err = rc; CSDR_HIP(hipEventRecord(b->dc_done[ki], b->streams[ki])); }
            else ok[ki] = 1;
        }
        for (size_t oi = 0; oi < b->cores.size(); oi++) {
            const size_t ki = (size_t)b->order[oi];
            ChainCore &k = *b->cores[ki];
            hipStream_t st = b->streams[ki];
            for (size_t kj = 0; kj < b->cores.size(); kj++)
                if (kj != ki) CSDR_HIP(hipStreamWaitEvent(st, b->dc_done[kj], 0));
            if (ok[ki]) {
                const int rc = k.step_post(d_out, out_stride, b->d_out_rows[ki], stereo, st);
                if (rc < 0 && !err) err = rc;
            }
            CSDR_HIP(hipEventRecord(b->joins[ki], st));
            CSDR_HIP(hipStreamWaitEvent(caller, b->joins[ki], 0));
        }
        return err ? err : CSDR_OK;
    }
    // Strict mode, several groups: every down-converter behind the first starts while the previous group's filter, S-meter,
    // peaks and walk hold part of the chip, and its workgroups are long (one wave walks its whole segment: 350 us) -- the
    // ones that do not fit at once start only when the first ones END, a second round that costs a whole workgroup time
    // for a few hundred stragglers (tools/wg_trace.py: 4031 of 4080 at once for the second group, 3277 of 4042 for the
    // third).  Alone the kernel loses 4-7 % at 13 / 12 waves per CU instead of 16 (tools/experiments/r6_k2_grid.sh), so the
    // later groups are cut into 13 x CUs and 12 x CUs workgroups and run as ONE round: strict C4 step 1.75-1.78 -> 1.67 ms.
    // CSDR_DC_WGS_CORUN="a[,b]" overrides (second group's, later groups' workgroups; 0 = one full round for all).
    static long corun_wgs[2] = {-1, -1};
    if (corun_wgs[0] < 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
        const char *e = getenv("CSDR_DC_WGS_CORUN");
        corun_wgs[1] = e && strchr(e, ',') ? atol(strchr(e, ',') + 1) : (e ? atol(e) : 12L * cus);
        corun_wgs[0] = e ? atol(e) : 13L * cus;
    }
    const bool strict_multi = forked && !b->pipelined && plain && b->cores.size() > 1;
    if (b->pipelined && b->chained && plain) {
        int pr_lo = 0, pr_hi = 0;
        CSDR_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        while (b->post_streams.size() < b->cores.size()) {
            hipStream_t st;
            CSDR_HIP(stream_pool().get(b->device, pr_hi, &st, STREAM_POST));
            b->post_streams.push_back(st);
        }
        // the first group's down-converter runs beside the previous call's last walks: the co-run grid for it too
        static const long first_wgs = getenv("CSDR_PIPE_FIRST_WGS") ? atol(getenv("CSDR_PIPE_FIRST_WGS")) : -1;
        const size_t ng = b->cores.size();
        for (size_t oi = 0; oi < ng; oi++) {
            const size_t ki = (size_t)b->order[oi];
            ChainCore &k = *b->cores[ki];
            k.pk = d_packets; k.pk_len = pkt_len; k.blank = blank;
            hipStream_t st = b->streams[ki], sp = b->post_streams[ki];
            long wgs = ng > 1 ? (oi > 0 ? corun_wgs[oi > 1 ? 1 : 0] : (b->have_last_dc ? (first_wgs >= 0 ? first_wgs : corun_wgs[1]) : 0)) : 0;
            csdr__downconvert_batch_set_wgs(k.dc, wgs);
            CSDR_HIP(hipStreamWaitEvent(st, b->fork, 0));
            // the caller's stream catches up with the PREVIOUS call behind this call's fork event (the pipelined contract)
            if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent(caller, b->joins[ki], 0)); b->prev_join[ki] = 0; }
            // one down-converter at a time, across calls: behind the previous group's, the first behind the previous call's last
            hipEvent_t after = oi > 0 ? b->dc_done[b->order[oi - 1]] : (b->have_last_dc ? b->dc_done[b->order[ng - 1]] : nullptr);
            k.pc.sm_borrow = nullptr; k.pc.sm_own_side = false;
            hipStream_t joined = st;
            const int rc = k.step_split(d_in, in_stride, b->d_rows[ki], n_per_channel, d_out, out_stride, b->d_out_rows[ki], stereo,
                                        st, sp, after, b->dc_done[ki], &joined);
            if (rc < 0 && !err) err = rc;
            CSDR_HIP(hipEventRecord(b->joins[ki], joined));
            b->prev_join[ki] = 1;
            b->prev_post[ki] = -1;
        }
        b->have_last_dc = !err;
        return err ? err : CSDR_OK;
    }
    // strict mode: CSDR_CHAIN_DC_CHAINED=0 starts every group's down-converter at once (A/B)
    static const bool dc_chained = !(getenv("CSDR_CHAIN_DC_CHAINED") && atoi(getenv("CSDR_CHAIN_DC_CHAINED")) == 0);
    for (size_t oi = 0; oi < b->cores.size(); oi++) {
        const size_t ki = (size_t)b->order[oi];
        ChainCore &k = *b->cores[ki];
        k.pk = d_packets; k.pk_len = pkt_len;            // this call's input as datagrams, or nullptr
        k.blank = blank;                                  // this call's blanker mask, or nullptr
        hipStream_t st = forked ? b->streams[ki] : caller;
        if (strict_multi) csdr__downconvert_batch_set_wgs(k.dc, oi > 0 ? corun_wgs[oi > 1 ? 1 : 0] : 0);
        else csdr__downconvert_batch_set_wgs(k.dc, 0);
        if (forked) CSDR_HIP(hipStreamWaitEvent(st, b->fork, 0));
        // pipelined: the caller's stream catches up with the PREVIOUS call only now, behind this call's fork
        // event, so that this call's down-converter is not held back by it: previous input consumed, output
        // rows of the call before that complete
        if (b->pipelined) {
            if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent(caller, b->joins[ki], 0)); b->prev_join[ki] = 0; }
            if (b->prev_post[ki] >= 0) { CSDR_HIP(hipStreamWaitEvent(caller, k.ev_post[b->prev_post[ki]], 0)); b->prev_post[ki] = -1; }
        }
        // strict mode: the groups' down-converters run one after the other (each fills the chip on its own) and
        // what follows a group's down-converter overlaps the next group's; pipelined mode: all at once, the
        // overlap comes from the next call
        int rc;
        if (k.s_post)
            rc = k.step_pipelined(d_in, in_stride, b->d_rows[ki], n_per_channel, d_out, out_stride, b->d_out_rows[ki], stereo,
                                  st, !b->pipelined && oi > 0 ? b->dc_done[b->order[oi - 1]] : nullptr, b->dc_done[ki]);
        else
        {
            // strict mode, several groups: the LAST group's filter, S-meter, peaks and walk are the end of the call, and
            // its S-meter -- which nothing in the call waits for -- goes to the first group's stream, long idle by then:
            // 30 us less on the critical path.  (CSDR_CHAIN_SM_BORROW=0: in the group's own stream, in front of the peaks.)
            static const bool borrow = !(getenv("CSDR_CHAIN_SM_BORROW") && atoi(getenv("CSDR_CHAIN_SM_BORROW")) == 0);
            k.pc.sm_borrow = (borrow && forked && !b->pipelined && oi > 0 && oi + 1 == b->cores.size())
                                 ? b->streams[b->order[0]] : nullptr;
            // ONE group (a single receiver, or receivers of one plan): the call is that group's walk from end to end, and
            // the S-meter scan beside it on a side stream of its own is 6 % of a C2 / C5 call
            k.pc.sm_own_side = borrow && !forked;
            rc = k.step(d_in, in_stride, b->d_rows[ki], n_per_channel, d_out, out_stride, b->d_out_rows[ki], stereo, st,
                        forked && oi > 0 && dc_chained ? b->dc_done[b->order[oi - 1]] : nullptr, forked ? b->dc_done[ki] : nullptr);
            k.pc.sm_borrow = nullptr;
        }
        if (rc < 0 && !err) err = rc;
        if (forked) {                                   // join even after an error: the caller's stream stays ordered
            CSDR_HIP(hipEventRecord(b->joins[ki], st));  // the input has been consumed (+ filter and shift, strict mode)
            if (b->pipelined) {                          // joined by the next call / flush
                b->prev_join[ki] = 1;
                b->prev_post[ki] = k.last_post;
            } else {
                CSDR_HIP(hipStreamWaitEvent(caller, b->joins[ki], 0));
                if (k.last_post >= 0) CSDR_HIP(hipStreamWaitEvent(caller, k.ev_post[k.last_post], 0));
            }
        }
    }
    return err ? err : CSDR_OK;
}
int csdr_demod_batch_process(csdr_demod_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                             float *d_out, long long out_stride, void *stream)
{ return demod_batch_run(b, d_in, in_stride, n_per_channel, d_out, out_stride, stream, false); }
/* the stereo overload of CDemodulator::ProcessData (demodulator.cpp:221-273) for every channel:
 * d_out_iq [channels][out_stride] complex fp32 (out_stride in complex samples) */
int csdr_demod_batch_process_stereo(csdr_demod_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                                    float *d_out_iq, long long out_stride, void *stream)
{ return demod_batch_run(b, d_in, in_stride, n_per_channel, d_out_iq, out_stride, stream, true); }
// the caller's blanker must be as wide as the chain and on its device: the mask has one row per receiver, and the
// down-converter indexes the blanker's state and history by input row
static int batch_blanker_fits(csdr_demod_batch *b, struct csdr_noiseproc_batch *nb)
{
    int ch = 0, dev = -1;
    const int rc = csdr__noiseproc_batch_shape(nb, &ch, &dev);
    if (rc) return rc;
    if (ch != b->channels || dev != b->device)
        return fail(CSDR_EINVAL, "blanker of %d channels on device %d given to a chain of %d on device %d", ch, dev,
                    b->channels, b->device);
    return CSDR_OK;
}
// the blanker's mask rows of a call of n samples per channel (fused form): [channels][mask_cap] words, grown when needed
static int batch_mask_rows(csdr_demod_batch *b, long n)
{
    const long words = (n + 31) / 32 + 64;
    if (words > b->mask_cap) {
        CSDR_HIP(hipDeviceSynchronize());
        if (b->d_mask) (void)hipFree(b->d_mask);
        b->d_mask = nullptr; b->mask_cap = 0;
        CSDR_HIP(hipMalloc((void **)&b->d_mask, (size_t)b->channels * words * sizeof(unsigned)));
        b->mask_cap = words;
    }
    b->blank.mask = b->d_mask; b->blank.mask_stride = b->mask_cap;
    return CSDR_OK;
}
int csdr_demod_batch_process_packets(csdr_demod_batch *b, const void *d_packets, int npackets, int pkt_len,
                                     struct csdr_noiseproc_batch *nb, float *d_out, long long out_stride,
                                     void *stream)
{
    if (!b || !d_packets || !d_out || npackets < 0) return fail(CSDR_EINVAL, "bad argument");
    if (pkt_len != 1028 && pkt_len != 1444) return fail(CSDR_EINVAL, "packet length %d", pkt_len);
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (npackets == 0) return CSDR_OK;
    if (!device_ok(b->device)) return CSDR_EHIP;
    const long n = (long)npackets * (pkt_len == 1444 ? 240 : 256);
    if (n > 0x7fffffffL) return fail(CSDR_EINVAL, "%d datagrams are more samples than one call can take", npackets);
    if (!nb)        // the down-converter decodes the datagrams in its own loads: no unpacked copy, no extra pass
        return demod_batch_run(b, nullptr, 0, (int)n, d_out, out_stride, stream, false, d_packets, pkt_len);
    { const int rcs = batch_blanker_fits(b, nb); if (rcs) return rcs; }
    // With the blanker.  The internal buffers below (mask / blanked samples) are single-buffered, and the blanker's
    // history halves alternate per call: in pipelined mode the down-converters of the PREVIOUS call (on the batch's own
    // streams) may still be reading them, and the caller's stream -- on which the blanker of this call runs -- has not
    // joined them yet (demod_batch_run does that, later)
    if (b->pipelined)
        for (size_t ki = 0; ki < b->cores.size(); ki++)
            if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, b->joins[ki], 0)); b->prev_join[ki] = 0; }
    // FUSED (default): the blanker decides, the down-converter applies -- noiseblank_kernel leaves one bit per sample,
    // downconv_kernel<.., BLK> reads the datagram sample delay_n + 1 behind and zeroes it under the mask in its own
    // load.  No blanked copy of the input: 8 B written + 8 B read back per sample less, and one input stream less in
    // the blanker ("fuses naturally into the NCO kernel's load").  CSDR_BLANK_FUSED=0: the two-pass form.
    static const bool fused = !(getenv("CSDR_BLANK_FUSED") && atoi(getenv("CSDR_BLANK_FUSED")) == 0);
    if (fused) {
        { const int rcm = batch_mask_rows(b, n); if (rcm) return rcm; }
        int rc = csdr__noiseproc_batch_mask(nb, nullptr, 0, d_packets, npackets, pkt_len, (int)n, b->d_mask, b->mask_cap,
                                            &b->blank.state, &b->blank.hist, stream);
        if (rc < 0) return rc;
        return demod_batch_run(b, nullptr, 0, (int)n, d_out, out_stride, stream, false, d_packets, pkt_len, &b->blank);
    }
    // two passes: the blanker decodes the datagrams in ITS loads and leaves blanked fp32 samples for the chain
    if (n > b->raw_cap) {
        CSDR_HIP(hipDeviceSynchronize());
        if (b->d_blank) (void)hipFree(b->d_blank);
        b->d_blank = nullptr; b->raw_cap = 0;
        CSDR_HIP(hipMalloc((void **)&b->d_blank, (size_t)b->channels * n * 8));
        b->raw_cap = n;
    }
    int rc = csdr__noiseproc_batch_process_packets(nb, d_packets, npackets, pkt_len, b->d_blank, b->raw_cap, stream);
    if (rc < 0) return rc;
    return demod_batch_run(b, b->d_blank, b->raw_cap, (int)n, d_out, out_stride, stream, false);
}
/* The strict / pipelined pass on fp32 rows with CNoiseProc's blanker in front (what CSdrInterface::ProcessIQData runs in
 * place before the chain, sdrinterface.cpp:884), FUSED like the datagram form: the blanker kernel leaves one bit per sample,
 * the down-converter takes the delayed sample from d_in itself and zeroes it under the mask -- no blanked copy of the
 * input is written or read. */
int csdr_demod_batch_process_blanked(csdr_demod_batch *b, const float *d_in, long long in_stride, int n_per_channel,
                                     struct csdr_noiseproc_batch *nb, float *d_out, long long out_stride, void *stream)
{
    if (!b || !d_in || !d_out || !nb || n_per_channel < 0) return fail(CSDR_EINVAL, "bad argument");
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (n_per_channel == 0) return CSDR_OK;
    if (!device_ok(b->device)) return CSDR_EHIP;
    for (size_t ki = 0; ki < b->cores.size(); ki++)          // rows shared between receivers would be blanked once per reader
        for (size_t q = 0; q < b->members[ki].size(); q++)
            if (b->members[ki][q] >= 0 && b->in_row[b->members[ki][q]] != b->members[ki][q])
                return fail(CSDR_ESTATE, "process_blanked: every receiver reads its own row (csdr_demod_batch_set_input_rows is off)");
    { const int rcs = batch_blanker_fits(b, nb); if (rcs) return rcs; }
    const long n = n_per_channel;
    if (b->pipelined)                                         // (the single-buffered mask, as in process_packets)
        for (size_t ki = 0; ki < b->cores.size(); ki++)
            if (b->prev_join[ki]) { CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, b->joins[ki], 0)); b->prev_join[ki] = 0; }
    { const int rcm = batch_mask_rows(b, n); if (rcm) return rcm; }
    int rc = csdr__noiseproc_batch_mask(nb, d_in, in_stride, nullptr, 0, 0, (int)n, b->d_mask, b->mask_cap,
                                        &b->blank.state, &b->blank.hist, stream);
    if (rc < 0) return rc;
    return demod_batch_run(b, d_in, in_stride, (int)n, d_out, out_stride, stream, false, nullptr, 0, &b->blank);
}
/* internal (diagnostics: tools/experiments/r6_repro_mode3.py): how fast each plan group's own buffers stream -- a
 * device-to-device copy of the filter-output rows into the spare rows, timed with events, per group in the batch's launch
 * order; us_out[k] = microseconds of the k-th group's copy, bytes_out[k] its size.  Synchronises. */
extern "C" int csdr__demod_batch_probe(csdr_demod_batch *b, double *us_out, double *bytes_out, int cap)
{
    if (!b || !us_out || !bytes_out) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CSDR_HIP(hipEventCreate(&e0)); CSDR_HIP(hipEventCreate(&e1));
    int n = 0;
    for (size_t oi = 0; oi < b->cores.size() && n < cap; oi++, n++) {
        ChainCore &k = *b->cores[(size_t)b->order[oi]];
        const size_t bytes = (size_t)k.rows * (size_t)k.cap * 8;
        us_out[n] = 0.0; bytes_out[n] = (double)bytes;
        if (!k.d_filt || !k.d_agc || !bytes) continue;
        for (int rep = 0; rep < 3; rep++) CSDR_HIP(hipMemcpyAsync(k.d_agc, k.d_filt, bytes, hipMemcpyDeviceToDevice, nullptr));
        CSDR_HIP(hipEventRecord(e0, nullptr));
        for (int rep = 0; rep < 10; rep++) CSDR_HIP(hipMemcpyAsync(k.d_agc, k.d_filt, bytes, hipMemcpyDeviceToDevice, nullptr));
        CSDR_HIP(hipEventRecord(e1, nullptr));
        CSDR_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        CSDR_HIP(hipEventElapsedTime(&ms, e0, e1));
        us_out[n] = ms * 100.0;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return n;
}
/* stage taps of a batch's receivers: see include/cutesdr_mi.h */
int csdr_demod_batch_set_taps(csdr_demod_batch *b, int mask)
{
    if (!b || mask < 0 || mask > 15) return fail(CSDR_EINVAL, "bad argument");
    if (b->cores.empty()) return fail(CSDR_ESTATE, "commit first");
    if (mask && b->pipelined) return fail(CSDR_ESTATE, "stage taps need the strict mode");
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    for (auto *k : b->cores) { k->taps = mask; k->tap1_n = 0; }
    b->taps = mask;
    return CSDR_OK;
}
int csdr_demod_batch_get_tap(csdr_demod_batch *b, int channel, int profile, float *out, int cap)
{
    if (!b || channel < 0 || channel >= b->channels || profile < 1 || profile > 3 || cap < 0 || (cap && !out))
        return fail(CSDR_EINVAL, "bad argument (PROFILE_4 is the caller's own output row)");
    if (b->core_of[channel] < 0) return fail(CSDR_ESTATE, "commit first");
    ChainCore &k = *b->cores[b->core_of[channel]];
    if (!(k.taps & (1 << (profile - 1)))) return fail(CSDR_ESTATE, "tap %d is not switched on", profile);
    if (!device_ok(b->device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    const int r = b->row_of[channel];
    const int n = profile == 1 ? k.tap1_n : k.last_out;
    if (2 * n > cap) return fail(CSDR_EINVAL, "tap %d holds %d floats, room for %d", profile, 2 * n, cap);
    const float *src = profile == 1 ? k.d_tap1 + 2 * (size_t)r * k.tap1_cap
                                    : (profile == 2 ? k.d_filt : k.d_agc) + 2 * (size_t)r * k.cap;
    if (n) CSDR_HIP(hipMemcpy(out, src, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 2 * n;
}
int csdr_demod_batch_group_count(csdr_demod_batch *b, int *rows)
{
    if (!b) return fail(CSDR_EINVAL, "bad handle");
    if (rows) { *rows = 0; for (auto *k : b->cores) *rows += k->rows; }
    return (int)b->cores.size();
}
/* audio samples channel `channel` received in the last process call */
int csdr_demod_batch_out_count(csdr_demod_batch *b, int channel)
{
    if (!b || channel < 0 || channel >= b->channels || b->core_of[channel] < 0) return fail(CSDR_EINVAL, "bad argument");
    return b->cores[b->core_of[channel]]->last_out;
}

}  // extern "C"
