// pc_unit.hpp -- device-side state block of `channels` post-filter chains (S-meter, AGC,
// demodulator) with its host mirror; shared by the leaf objects and the CDemodulator chain.
#pragma once
#include <vector>
#include <cstring>
#include <cstdlib>
#include "capi_common.hpp"
#include "postchain.h"
#include "pc_host.hpp"
#include "patch_queue.hpp"
#include <cstddef>

namespace csdr {

// CSDR_FM_DEFER=0 keeps the squelch inside the walk (A/B runs, diagnostics)
inline bool defer_fm_squelch()
{
    static const bool on = !(getenv("CSDR_FM_DEFER") && atoi(getenv("CSDR_FM_DEFER")) == 0);
    return on;
}

// CSDR_AGC_PRE=0 keeps the AGC's magnitudes and sliding maximum inside the walk (A/B runs, diagnostics)
inline bool agc_prepass()
{
    static const bool on = !(getenv("CSDR_AGC_PRE") && atoi(getenv("CSDR_AGC_PRE")) == 0);
    return on;
}

// CSDR_SM_CALL=0 keeps the S-meter inside the walk (A/B runs, diagnostics)
inline bool smeter_whole_call()
{
    static const bool on = !(getenv("CSDR_SM_CALL") && atoi(getenv("CSDR_SM_CALL")) == 0);
    return on;
}

// CSDR_PLL_OVERLAP=0: an FM tile whose PLL is not locked is walked by one thread, as before pll_overlap (A/B, tests)
inline bool pll_overlap_on()
{
    static const bool on = !(getenv("CSDR_PLL_OVERLAP") && atoi(getenv("CSDR_PLL_OVERLAP")) == 0);
    return on;
}

struct PcUnit {
    int device = 0, channels = 0;
    PcChannel *d_chan = nullptr;
    float *d_dly = nullptr, *d_mag = nullptr, *d_scratch = nullptr;
    long scratch_cap = 0;
    double *d_sqbuf = nullptr; long sqbuf_cap = 0;       // per-burst records of the deferred FM squelch (fm_squelch_launch)
    float *d_pkbuf = nullptr, *d_magtail = nullptr; long pkbuf_cap = 0;   // AGC peaks of a call (agc_peaks_launch)
    bool no_output = false;
    hipStream_t s_side = nullptr;                        // the whole-call S-meter runs here, beside the walk
    hipEvent_t ev_side_fork = nullptr, ev_side_join = nullptr;
    bool sm_own_side = false;                            // set by the batch chain when it is ONE plan group: the S-meter may
                                                         // take a side stream of this unit's own (one more stream in all)
    hipStream_t sm_borrow = nullptr;                     // set per call by the batch chain, not owned: an EXISTING stream
                                                         // that has nothing left to do in this call -- the whole-call
                                                         // S-meter of the group whose walk ends the call runs there
    // Parameter changes between two calls (AGC constants, S-meter rate, squelch, the demodulators' FIRs) do not stop anything:
    // the host mirror below is authoritative for PARAMETERS, a setter computes the new words there and queues them as patches
    // of exactly those fields; run() applies the queue on its own stream in front of its first launch (patch_queue.hpp).
    // The STATE words of a channel (averagers, PLL, filter memories, ring positions) live on the device only and are
    // never sent from the mirror -- except where the reference resets them (AGC at a new sample rate, a redesigned FIR).
    PatchQueue patches;
    std::vector<PcChannel> h;            // host mirror (authoritative for parameters)
    std::vector<HostAgc> hagc;
    std::vector<HostFir> fir_am, fir_sam, fir_fm;

    ~PcUnit()
    {
        if (d_chan) (void)hipFree(d_chan);
        if (d_dly) (void)hipFree(d_dly);
        if (d_mag) (void)hipFree(d_mag);
        if (d_scratch) (void)hipFree(d_scratch);
        if (d_sqbuf) (void)hipFree(d_sqbuf);
        if (d_pkbuf) (void)hipFree(d_pkbuf);
        if (d_magtail) (void)hipFree(d_magtail);
        if (d_sm) (void)hipFree(d_sm);
        if (s_side) { (void)hipStreamSynchronize(s_side); (void)hipStreamDestroy(s_side); }
        if (ev_side_fork) (void)hipEventDestroy(ev_side_fork);
        if (ev_side_join) (void)hipEventDestroy(ev_side_join);
    }
    int init(int dev, int nch)
    {
        device = dev; channels = nch;
        h.resize(nch); hagc.resize(nch); fir_am.resize(nch); fir_sam.resize(nch); fir_fm.resize(nch);
        for (int c = 0; c < nch; c++) {
            memset(&h[c], 0, sizeof(PcChannel));
            h[c].mode = PC_MODE_NONE;
            smeter_init(h[c].sm);
            h[c].agc.on = 1; h[c].agc.dly_n = 1; h[c].agc.win_n = 1;      // agc.cpp:80-89
            h[c].agc.peak = -16.0; h[c].agc.attack_ave = -5.0; h[c].agc.decay_ave = -5.0;
        }
        CSDR_HIP(hipMalloc((void **)&d_chan, sizeof(PcChannel) * nch));
        CSDR_HIP(hipMalloc((void **)&d_dly, sizeof(float) * 2 * PC_AGC_RING * nch));
        CSDR_HIP(hipMalloc((void **)&d_mag, sizeof(float) * PC_AGC_RING * nch));
        CSDR_HIP(hipMemset(d_dly, 0, sizeof(float) * 2 * PC_AGC_RING * nch));
        std::vector<float> m16((size_t)PC_AGC_RING * nch, -16.0f);
        CSDR_HIP(hipMemcpy(d_mag, m16.data(), m16.size() * 4, hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(d_chan, h.data(), sizeof(PcChannel) * nch, hipMemcpyHostToDevice));
        return CSDR_OK;
    }
    int pull(int c)      // device -> host mirror (state advances on the device); rows that MOVE and new demodulator objects only
    {
        CSDR_HIP(hipSetDevice(device));
        { const int rcp = patches.flush(nullptr); if (rcp) return rcp; }     // nothing queued may be lost to the copy below
        CSDR_HIP(hipDeviceSynchronize());
        CSDR_HIP(hipMemcpy(&h[c], d_chan + c, sizeof(PcChannel), hipMemcpyDeviceToHost));
        return CSDR_OK;
    }
    int push(int c)
    {
        CSDR_HIP(hipSetDevice(device));
        { const int rcp = patches.flush(nullptr); if (rcp) return rcp; }     // older queued words must not land on top of these
        CSDR_HIP(hipMemcpy(d_chan + c, &h[c], sizeof(PcChannel), hipMemcpyHostToDevice));
        return CSDR_OK;
    }
    int agc_rings_clear(int c)
    {
        CSDR_HIP(hipMemset(d_dly + (size_t)c * 2 * PC_AGC_RING, 0, sizeof(float) * 2 * PC_AGC_RING));
        std::vector<float> m16(PC_AGC_RING, -16.0f);
        CSDR_HIP(hipMemcpy(d_mag + (size_t)c * PC_AGC_RING, m16.data(), m16.size() * 4, hipMemcpyHostToDevice));
        return CSDR_OK;
    }
    // channel `sc` of `src` continues as channel `c` here: S-meter, AGC (rings included) and demodulator objects
    // with all their state (csdr_demod_batch_set_demod moving a receiver to another plan group)
    int import_channel(int c, PcUnit &src, int sc)
    {
        int rc = src.pull(sc);
        if (rc) return rc;
        h[c] = src.h[sc]; hagc[c] = src.hagc[sc];
        fir_am[c] = src.fir_am[sc]; fir_sam[c] = src.fir_sam[sc]; fir_fm[c] = src.fir_fm[sc];
        CSDR_HIP(hipMemcpy(d_dly + (size_t)c * 2 * PC_AGC_RING, src.d_dly + (size_t)sc * 2 * PC_AGC_RING,
                           sizeof(float) * 2 * PC_AGC_RING, hipMemcpyDeviceToDevice));
        CSDR_HIP(hipMemcpy(d_mag + (size_t)c * PC_AGC_RING, src.d_mag + (size_t)sc * PC_AGC_RING,
                           sizeof(float) * PC_AGC_RING, hipMemcpyDeviceToDevice));
        return push(c);
    }
    // a run of fields [first, last] of row c's PcChannel, from the host mirror, as one patch
    template <class A, class B> int patch_fields(int c, const A &first, const B &last)
    {
        const unsigned char *b0 = reinterpret_cast<const unsigned char *>(&h[c]);
        const unsigned char *f0 = reinterpret_cast<const unsigned char *>(&first), *f1 = reinterpret_cast<const unsigned char *>(&last) + sizeof(B);
        return patches.add(reinterpret_cast<unsigned char *>(d_chan + c) + (f0 - b0), f0, (size_t)(f1 - f0));
    }
    // CAgc::SetParameters (agc.cpp:104-167): constants always; delay line, window and averagers only when the sample rate
    // changed (:121-136) -- nothing is read back, nobody waits
    int agc_set(int c, int on, int hang, int thresh, int manual, int slope, int decay, double fs)
    {
        const int ch = hagc[c].set(h[c].agc, on != 0, hang != 0, thresh, manual, slope, decay, fs);
        if (ch == 0) return CSDR_OK;
        PcAgc &a = h[c].agc;
        int rc = patch_fields(c, a.on, a.hang_time);
        if (!rc) rc = patch_fields(c, a.manual_gain, a.dec_fall);
        if (!rc && ch == 2) {
            rc = patch_fields(c, a.dly_pos, a.hang_timer);
            if (!rc) rc = patch_fields(c, a.peak, a.decay_ave);
            const float m16 = -16.0f;
            unsigned w16; memcpy(&w16, &m16, 4);
            if (!rc) rc = patches.add_fill(d_dly + (size_t)c * 2 * PC_AGC_RING, 0u, sizeof(float) * 2 * PC_AGC_RING);
            if (!rc) rc = patches.add_fill(d_mag + (size_t)c * PC_AGC_RING, w16, sizeof(float) * PC_AGC_RING);
        }
        return rc;
    }
    int smeter_rate_set(int c, double fs)
    {
        if (h[c].sm.fs == fs) return CSDR_OK;
        smeter_rate(h[c].sm, fs);
        return patch_fields(c, h[c].sm.att_a, h[c].sm.fs);
    }
    // CFmDemod::SetSquelch (fmdemod.cpp:95-98) and the HiCut-dependent high-pass of ProcessData (:160-164: redesigned, its
    // delay line cleared, when the bandwidth changes)
    int fm_params_set(int c, int squelch_value, double demod_rate, double fm_bw)
    {
        PcFm &f = h[c].fm;
        fm_set_squelch(f, squelch_value);
        int rc = patch_fields(c, f.sq_thresh, f.sq_thresh);
        const double before = f.hp_freq;
        fm_set_bw(f, fir_fm[c], demod_rate, fm_bw);
        if (!rc && f.hp_freq != before) {
            rc = patch_fields(c, f.hp_freq, f.hp_freq);
            if (!rc) rc = patch_fields(c, f.hp, f.hp);
        }
        return rc;
    }
    // CAmDemod::SetBandwidth (amdemod.cpp:56-60): a new low-pass, its delay line cleared (fir.cpp:229-235), on every SetDemod
    int am_bandwidth_set(int c, double demod_rate, double bw)
    {
        am_bandwidth(h[c].am, fir_am[c], demod_rate, bw);
        return patch_fields(c, h[c].am.fir, h[c].am.fir);
    }
    // CSMeter::GetPeak resets the peak (smeter.cpp:98-103).  Read and reset on the device, one field: the rest of
    // the channel state (AGC, PLL, filter memories) is never written back from a stale host copy.
    double *d_sm = nullptr;              // scratch of the single-channel getters
    int smeter_read(int c, bool want_peak, double *out)
    {
        CSDR_HIP(hipSetDevice(device));
        CSDR_HIP(hipDeviceSynchronize());
        if (!d_sm) CSDR_HIP(hipMalloc((void **)&d_sm, 2 * sizeof(double)));
        CSDR_HIP(smeter_collect_launch(d_chan + c, 1, nullptr, want_peak ? nullptr : d_sm, want_peak ? d_sm + 1 : nullptr, nullptr));
        double v[2] = {0, 0};
        CSDR_HIP(hipMemcpy(v, d_sm, sizeof(v), hipMemcpyDeviceToHost));
        *out = want_peak ? v[1] : v[0];
        return CSDR_OK;
    }
    double smeter_peak(int c)
    {
        double x = 0.0;
        return smeter_read(c, true, &x) ? 0.0 : x;
    }
    double smeter_ave(int c)
    {
        double x = 0.0;
        return smeter_read(c, false, &x) ? 0.0 : x;
    }
    int run(int flags, const float *d_in, long in_stride, float *d_out, long out_stride, int nbursts,
            int burst, hipStream_t stream, const int *d_out_rows = nullptr)
    {
        CSDR_HIP(hipSetDevice(device));
        { const int rcp = patches.flush(stream); if (rcp) return rcp; }       // parameters set since the last call
        if (burst > scratch_cap) {
            if (d_scratch) (void)hipFree(d_scratch);
            d_scratch = nullptr; scratch_cap = 0;
            CSDR_HIP(hipMalloc((void **)&d_scratch, sizeof(float) * (size_t)burst * channels));
            scratch_cap = burst;
        }
        PcArgs a;
        a.chan = d_chan; a.agc_dly = d_dly; a.agc_mag = d_mag;
        a.in = d_in; a.in_stride = in_stride;
        a.out = no_output ? nullptr : d_out; a.out_stride = out_stride; a.out_rows = d_out_rows;
        a.scratch = d_scratch; a.scratch_stride = scratch_cap;
        a.channels = channels; a.nbursts = nbursts; a.burst = burst; a.flags = flags;
        a.sqbuf = nullptr;
        // FM receivers in a call of several bursts (the chain): the squelch half of CFmDemod -- high-pass, average,
        // decision, low-pass -- leaves the sequential walk and follows as burst-parallel launches; same words out
        bool defer = (flags & PC_DO_DEMOD) && nbursts >= 4 && burst >= PC_FIR_MAX && burst <= 16384 && a.out && defer_fm_squelch();
        if (defer) {
            defer = false;
            for (int c = 0; c < channels; c++) if (h[c].mode == PC_MODE_FM) { defer = true; break; }
        }
        if (defer) {
            const long need = (long)channels * nbursts * PC_SQ_REC;
            if (need > sqbuf_cap) {
                CSDR_HIP(hipStreamSynchronize(stream));              // an earlier launch may still be using the old buffer
                if (d_sqbuf) (void)hipFree(d_sqbuf);
                d_sqbuf = nullptr; sqbuf_cap = 0;
                CSDR_HIP(hipMalloc((void **)&d_sqbuf, sizeof(double) * need));
                sqbuf_cap = need;
            }
            a.flags |= PC_FM_DEFER; a.sqbuf = d_sqbuf;
        }
        if (!pll_overlap_on()) a.flags |= PC_PLL_SEQ;
        // the S-meter of a call of several bursts: one scan per receiver instead of two per walked tile, in front of
        // the peaks kernel and the walk
        bool sm_forked = false;
        if ((a.flags & PC_DO_SMETER) && (long)nbursts * burst >= 4096 && smeter_whole_call()) {
            // (a stream of its own: measured 2.31 against 1.80 ms strict and 4.7-6.1 against 1.74 pipelined -- with one more
            // stream per plan group the process goes past the hardware queues it is given; opt-in for diagnostics only)
            static const bool side = getenv("CSDR_SM_SIDE") && atoi(getenv("CSDR_SM_SIDE")) != 0;
            const bool alone = !(a.flags & (PC_DO_AGC | PC_DO_DEMOD));
            if ((side || sm_borrow || sm_own_side) && !alone) {
                if (!sm_borrow && !s_side) CSDR_HIP(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
                if (!ev_side_fork) {
                    CSDR_HIP(hipEventCreateWithFlags(&ev_side_fork, hipEventDisableTiming));
                    CSDR_HIP(hipEventCreateWithFlags(&ev_side_join, hipEventDisableTiming));
                }
                hipStream_t ss = sm_borrow ? sm_borrow : s_side;
                CSDR_HIP(hipEventRecord(ev_side_fork, stream));
                CSDR_HIP(hipStreamWaitEvent(ss, ev_side_fork, 0));
                CSDR_HIP(smeter_call_launch(a, ss));
                CSDR_HIP(hipEventRecord(ev_side_join, ss));
                sm_forked = true;
            } else {
                CSDR_HIP(smeter_call_launch(a, stream));
            }
            a.flags &= ~PC_DO_SMETER;
            if (alone) return CSDR_OK;
        }
        // the AGC's log magnitudes and sliding maximum of the whole call: burst-parallel, in front of the walk
        a.pkbuf = nullptr; a.magtail = nullptr;
        bool pre = (flags & PC_DO_AGC) && !(flags & PC_AGC_REAL) && nbursts >= 4 && agc_prepass();
        if (pre) {
            pre = false;
            for (int c = 0; c < channels; c++) if (h[c].agc.on) { pre = true; break; }
        }
        if (pre) {
            const long need = (long)channels * nbursts * burst;
            if (need > pkbuf_cap) {
                CSDR_HIP(hipStreamSynchronize(stream));
                if (d_pkbuf) (void)hipFree(d_pkbuf);
                d_pkbuf = nullptr; pkbuf_cap = 0;
                CSDR_HIP(hipMalloc((void **)&d_pkbuf, sizeof(float) * need));
                pkbuf_cap = need;
            }
            if (!d_magtail) CSDR_HIP(hipMalloc((void **)&d_magtail, sizeof(float) * PC_AGC_RING * channels));
            a.flags |= PC_AGC_PRE; a.pkbuf = d_pkbuf; a.magtail = d_magtail;
            CSDR_HIP(agc_peaks_launch(a, stream));
        }
        {   // may the walk take its lean instantiation?  (the conditions its compiled-out code would have served)
            bool any_fm = false, any_agc = false;
            for (int c = 0; c < channels; c++) { any_fm = any_fm || h[c].mode == PC_MODE_FM; any_agc = any_agc || h[c].agc.on; }
            const bool agc_ok = !(a.flags & PC_DO_AGC) || (!(a.flags & PC_AGC_REAL) && (pre || !any_agc));
            const bool fm_ok = !(a.flags & PC_DO_DEMOD) || defer || !any_fm;
            if (!(a.flags & PC_DO_SMETER) && agc_ok && fm_ok) a.flags |= PC_LEAN;
        }
        CSDR_HIP(postchain_launch(a, stream));
        if (defer) CSDR_HIP(fm_squelch_launch(a, stream));
        if (sm_forked) CSDR_HIP(hipStreamWaitEvent(stream, ev_side_join, 0));
        return CSDR_OK;
    }
};

}  // namespace csdr
