// pc_host.hpp -- host-side fp64 setup math of the sample-rate stages (filter design, time
// constants, PLL gains).  Restates the reference's setup functions; runs on parameter changes.
#pragma once
#include <cmath>
#include <cstring>
#include "host_math.hpp"
#include "postchain.h"
#include "ref_constants.hpp"

namespace csdr {

// ---- CFir design (dsp/fir.cpp:173-432) -------------------------------------------------------------
struct HostFir {
    int ntaps = 1;
    double fs = 0;
    double coef[PC_FIR_MAX] = {0}, icoef[PC_FIR_MAX] = {0}, qcoef[PC_FIR_MAX] = {0};

    static double izero(double x)                  // :414-432
    {
        double x2 = x / 2.0, sum = 1.0, ds = 1.0, di = 1.0, t;
        do { t = x2 / di; t *= t; ds *= t; sum += ds; di += 1.0; } while (ds >= 1e-9 * sum);
        return sum;
    }
    static double beta_of(double astop)            // :184-190
    {
        if (astop < 20.96) return 0;
        if (astop >= 50.0) return .1102 * (astop - 8.71);
        return .5842 * std::pow((astop - 20.96), 0.4) + .07886 * (astop - 20.96);
    }
    void finish()
    {
        for (int n = 0; n < ntaps; n++) { icoef[n] = coef[n]; qcoef[n] = coef[n]; }
    }
    void init_const(int n, const double *c)        // :133-153
    {
        ntaps = n > PC_FIR_MAX ? PC_FIR_MAX : n;
        for (int i = 0; i < ntaps; i++) coef[i] = c[i];
        finish();
    }
    int init_lp(double scale, double astop, double fpass, double fstop, double fsamp)   // :173-261
    {
        fs = fsamp;
        const double npass = fpass / fsamp, nstop = fstop / fsamp, ncut = (nstop + npass) / 2.0;
        const double beta = beta_of(astop);
        ntaps = (int)((astop - 8.0) / (2.285 * kTwoPi * (nstop - npass)) + 1);
        if (ntaps > PC_FIR_MAX) ntaps = PC_FIR_MAX;
        if (ntaps < 3) ntaps = 3;
        const double centre = .5 * (double)(ntaps - 1), izb = izero(beta);
        for (int n = 0; n < ntaps; n++) {
            double x = (double)n - centre, c;
            if ((double)n == centre) c = 2.0 * ncut;
            else c = std::sin(kTwoPi * x * ncut) / (kPi * x);
            x = ((double)n - ((double)ntaps - 1.0) / 2.0) / (((double)ntaps - 1.0) / 2.0);
            coef[n] = scale * c * izero(beta * std::sqrt(1 - (x * x))) / izb;
        }
        finish();
        return ntaps;
    }
    int init_hp(double scale, double astop, double fpass, double fstop, double fsamp)   // :278-367
    {
        fs = fsamp;
        const double npass = fpass / fsamp, nstop = fstop / fsamp, ncut = (nstop + npass) / 2.0;
        const double beta = beta_of(astop);
        ntaps = (int)((astop - 8.0) / (2.285 * kTwoPi * (npass - nstop)) + 1);
        if (ntaps > (PC_FIR_MAX - 1)) ntaps = PC_FIR_MAX - 1;
        if (ntaps < 3) ntaps = 3;
        ntaps |= 1;
        const double izb = izero(beta), centre = .5 * (double)(ntaps - 1);
        for (int n = 0; n < ntaps; n++) {
            double x = (double)n - (double)(ntaps - 1) / 2.0, c;
            if ((double)n == centre) c = 1.0 - 2.0 * ncut;
            else c = std::sin(kPi * x) / (kPi * x) - std::sin(kTwoPi * x * ncut) / (kPi * x);
            x = ((double)n - ((double)ntaps - 1.0) / 2.0) / (((double)ntaps - 1.0) / 2.0);
            coef[n] = scale * c * izero(beta * std::sqrt(1 - (x * x))) / izb;
        }
        finish();
        return ntaps;
    }
    void gen_hilbert(double off)                   // :374-407
    {
        for (int n = 0; n < ntaps; n++) {
            const double a = (kTwoPi * off / fs) * ((double)n - ((double)(ntaps - 1) / 2.0));
            icoef[n] = 2.0 * coef[n] * std::cos(a);
            qcoef[n] = 2.0 * coef[n] * std::sin(a);
        }
    }
    // taps to the device struct; reset=true also clears the delay line (every Init* does)
    void upload(PcFir &d, bool reset) const
    {
        d.ntaps = ntaps;
        for (int i = 0; i < PC_FIR_MAX; i++) {
            d.coef[i] = i < ntaps ? (float)coef[i] : 0.f;
            d.icoef[i] = i < ntaps ? (float)icoef[i] : 0.f;
            d.qcoef[i] = i < ntaps ? (float)qcoef[i] : 0.f;
        }
        if (reset) { d.pos = 0; memset(d.zreal, 0, sizeof(d.zreal)); memset(d.zr, 0, sizeof(d.zr)); memset(d.zi, 0, sizeof(d.zi)); }
    }
};

// ---- CIir design (dsp/iir.cpp:86-165), kind 0 LP, 1 HP, 2 BP, 3 BR; clears the state -----------------
inline void iir_design(PcIir &f, int kind, double f0, double q, double fs)
{
    const double w0 = kTwoPi * f0 / fs, alpha = std::sin(w0) / (2.0 * q), A = 1.0 / (1.0 + alpha);
    switch (kind) {
    case 0: f.b0 = A * ((1.0 - std::cos(w0)) / 2.0); f.b1 = A * (1.0 - std::cos(w0)); f.b2 = A * ((1.0 - std::cos(w0)) / 2.0); break;
    case 1: f.b0 = A * ((1.0 + std::cos(w0)) / 2.0); f.b1 = -A * (1.0 + std::cos(w0)); f.b2 = A * ((1.0 + std::cos(w0)) / 2.0); break;
    case 2: f.b0 = A * alpha; f.b1 = 0.0; f.b2 = A * -alpha; break;
    default: f.b0 = A * 1.0; f.b1 = A * (-2.0 * std::cos(w0)); f.b2 = A * 1.0; break;
    }
    f.a1 = A * (-2.0 * std::cos(w0));
    f.a2 = A * (1.0 - alpha);
    f.w1a = f.w2a = f.w1b = f.w2b = 0.0;
}

// ---- CSMeter (dsp/smeter.cpp:49-72) -------------------------------------------------------------------
inline void smeter_init(PcSMeter &s)
{
    s.peak_mag = 0; s.fs = 1.0; s.att_a = 1.0; s.dec_a = 1.0; s.att_ave = -120.0; s.dec_ave = -120.0; s.ave_mag = 0;
}
inline void smeter_rate(PcSMeter &s, double fs)
{
    if (fs != s.fs) {
        s.fs = fs;
        s.att_a = (1.0 - std::exp(-1.0 / (fs * refc::SM_ATTACK_TIMECONST)));
        s.dec_a = (1.0 - std::exp(-1.0 / (fs * refc::SM_DECAY_TIMECONST)));
    }
}

// ---- CAgc::SetParameters (dsp/agc.cpp:104-167) ---------------------------------------------------------
struct HostAgc {
    bool on = true, hang = false;
    int thresh = 0, manual = 0, decay = 0;
    double slope = 0, fs = 100.0;
    // returns 0 unchanged, 1 parameters changed, 2 parameters changed and rings must be cleared
    int set(PcAgc &d, bool on_, bool hang_, int thresh_, int manual_, int slope_, int decay_, double fs_)
    {
        if (on_ == on && hang_ == hang && thresh_ == thresh && manual_ == manual &&
            (double)slope_ == slope && decay_ == decay && fs_ == fs)
            return 0;
        on = on_; hang = hang_; thresh = thresh_; manual = manual_; slope = slope_; decay = decay_;
        int rc = 1;
        if (fs != fs_) {
            fs = fs_;
            d.dly_pos = 0; d.hang_timer = 0; d.peak = -16.0; d.decay_ave = -5.0; d.attack_ave = -5.0; d.mag_pos = 0;
            rc = 2;
        }
        d.on = on; d.hang = hang;
        d.manual_gain = refc::AGC_MAX_MANUAL_AMPLITUDE * std::pow(10.0, -(100 - (double)manual) / 20.0);
        d.knee = (double)thresh / 20.0;
        d.gain_slope = slope / (100.0);
        d.fixed_gain = refc::AGC_OUTSCALE * std::pow(10.0, d.knee * (d.gain_slope - 1.0));
        d.att_rise = (1.0 - std::exp(-1.0 / (fs * refc::AGC_ATTACK_RISE_TIMECONST)));
        d.att_fall = (1.0 - std::exp(-1.0 / (fs * refc::AGC_ATTACK_FALL_TIMECONST)));
        d.dec_rise = (1.0 - std::exp(-1.0 / (fs * (double)decay * .001 * refc::AGC_DECAY_RISEFALL_RATIO)));
        d.hang_time = (int)(fs * (double)decay * .001);
        if (hang) d.dec_fall = (1.0 - std::exp(-1.0 / (fs * refc::AGC_RELEASE_TIMECONST)));
        else      d.dec_fall = (1.0 - std::exp(-1.0 / (fs * (double)decay * .001)));
        d.dly_n = (int)(fs * refc::AGC_DELAY_TIMECONST);
        d.win_n = (int)(fs * refc::AGC_WINDOW_TIMECONST);
        if (d.dly_n >= PC_AGC_RING - 1) d.dly_n = PC_AGC_RING - 1;
        if (d.win_n > PC_AGC_RING) d.win_n = PC_AGC_RING;      // reference overruns its buffer here (App. A.6)
        if (d.dly_n < 1) d.dly_n = 1;
        if (d.win_n < 1) d.win_n = 1;
        return rc;
    }
};

// ---- demodulator constructors -----------------------------------------------------------------------------
inline void am_init(PcAm &d, HostFir &fir, double fs)           // amdemod.cpp:50-54
{
    d.z1 = 0.0;
    fir.init_lp(1.0, 50.0, 10000, 10000 * 1.8, fs);
    fir.upload(d.fir, true);
}
inline void am_bandwidth(PcAm &d, HostFir &fir, double fs, double bw)   // :56-60 (resets the FIR state)
{
    fir.init_lp(1.0, 50.0, bw, bw * 1.8, fs);
    fir.upload(d.fir, true);
}
inline void sam_init(PcSam &d, HostFir &fir, double fs)         // samdemod.cpp:54-73
{
    const double norm = kTwoPi / fs;
    d.y1 = d.z1 = 0.0; d.phase = 0.0; d.freq = 0.0;
    d.lo = -refc::SAM_PLL_LIMIT * norm; d.hi = refc::SAM_PLL_LIMIT * norm;
    d.alpha = 2.0 * refc::SAM_PLL_ZETA * refc::SAM_PLL_BW * norm;
    d.beta = (d.alpha * d.alpha) / (4.0 * refc::SAM_PLL_ZETA * refc::SAM_PLL_ZETA);
    fir.init_lp(1.0, 40.0, 4500, 5500, fs);
    fir.gen_hilbert(5000.0);
    fir.upload(d.fir, true);
}
inline void fm_init(PcFm &d, HostFir &hp, double fs)            // fmdemod.cpp:62-89
{
    const double norm = kTwoPi / fs;
    d.err_dc = 0.0; d.phase = 0.0; d.freq = 0.0;
    d.lo = -refc::FM_PLL_RANGE * norm; d.hi = refc::FM_PLL_RANGE * norm;
    d.alpha = 2.0 * refc::FM_PLL_ZETA * refc::FM_VOICE_BANDWIDTH * 2.0 * norm;      // 2.0*FMPLL_ZETA*FMPLL_BW*norm, FMPLL_BW = VOICE_BANDWIDTH*2.0
    d.beta = (d.alpha * d.alpha) / (4.0 * refc::FM_PLL_ZETA * refc::FM_PLL_ZETA);
    d.out_gain = refc::FM_MAX_OUT / d.hi;
    d.dc_alpha = (1.0 - std::exp(-1.0 / (fs * refc::FM_DC_ALPHA)));
    d.hp_freq = refc::FM_VOICE_BANDWIDTH;
    d.sq_ave = 0.0; d.squelched = 1; d.sq_thresh = 0.0;
    d.sq_alpha = (1.0 - std::exp(-1.0 / (fs * refc::FM_SQUELCHAVE_TIMECONST)));
    iir_design(d.lp, 0, refc::FM_VOICE_BANDWIDTH, 1.0, fs);
    hp.init_hp(1.0, 50.0, d.hp_freq, d.hp_freq * .6, fs);
    hp.upload(d.hp, true);
}
inline void fm_set_bw(PcFm &d, HostFir &hp, double fs, double fm_bw)   // fmdemod.cpp:160-164
{
    if (d.hp_freq != fm_bw) {
        d.hp_freq = fm_bw;
        hp.init_hp(1.0, 50.0, d.hp_freq, d.hp_freq * .6, fs);
        hp.upload(d.hp, true);
    }
}
inline void fm_set_squelch(PcFm &d, int value)                  // fmdemod.cpp:95-98
{
    d.sq_thresh = (double)(refc::FM_SQUELCH_MAX - ((refc::FM_SQUELCH_MAX * value) / 99));
}

}  // namespace csdr
