// postchain_kernels.hip -- S-meter, AGC and demodulators for gfx950 (K4 in DESIGN.md).
//
// Replaces, per channel and per burst of band-pass output, the tail of
// CDemodulator::ProcessData (reference dsp/demodulator.cpp:182-207):
//   CSMeter::ProcessData (dsp/smeter.cpp:62-93) -> CAgc::ProcessData (dsp/agc.cpp:174-296)
//   -> C{Am,Sam,Fm,Ssb}Demod::ProcessData (dsp/amdemod.cpp:66-104, samdemod.cpp:78-158,
//   fmdemod.cpp:113-236, ssbdemod.cpp:48-60) with their CFir (dsp/fir.cpp:72-127) and CIir
//   (dsp/iir.cpp:171-201) helpers.
//
// These stages are strictly sequential in time inside one channel (sliding-window peak with an
// equality test, attack/decay averagers, second-order PLLs, biquads, a hysteresis squelch that
// is decided once per burst), so the parallel axis is the channel: one lane per channel, 64
// channels per wave, every lane walking its own burst.  The rate here is the decimated one
// (<= 78 kS/s per channel): this kernel is latency-bound by construction, not bandwidth-bound.
#include <hip/hip_runtime.h>
#include "postchain.h"

namespace csdr {

constexpr double kTwoPiD = 2.0 * 3.14159265358979323846;

// ---- CFir: y[n] = sum_k h[k] x[n-k] over a ring (dsp/fir.cpp:72-127) -----------------------------
__device__ __forceinline__ float fir_real(PcFir &f, float x)
{
    int p = f.pos + 1;
    if (p >= f.ntaps) p = 0;
    f.pos = p;
    f.zreal[p] = x;
    float acc = 0.f;
    int q = p;
    for (int k = 0; k < f.ntaps; k++) {
        acc += f.coef[k] * f.zreal[q];
        if (--q < 0) q = f.ntaps - 1;
    }
    return acc;
}
__device__ __forceinline__ void fir_cpx(PcFir &f, float &re, float &im)
{
    int p = f.pos + 1;
    if (p >= f.ntaps) p = 0;
    f.pos = p;
    f.zr[p] = re; f.zi[p] = im;
    float ar = 0.f, ai = 0.f;
    int q = p;
    for (int k = 0; k < f.ntaps; k++) {
        ar += f.icoef[k] * f.zr[q];
        ai += f.qcoef[k] * f.zi[q];
        if (--q < 0) q = f.ntaps - 1;
    }
    re = ar; im = ai;
}
// ---- CIir (dsp/iir.cpp:171-201) ------------------------------------------------------------------
__device__ __forceinline__ float iir_a(PcIir &f, float x)
{
    const double w0 = (double)x - f.a1 * f.w1a - f.a2 * f.w2a;
    const double y = f.b0 * w0 + f.b1 * f.w1a + f.b2 * f.w2a;
    f.w2a = f.w1a; f.w1a = w0;
    return (float)y;
}
__device__ __forceinline__ float iir_b(PcIir &f, float x)
{
    const double w0 = (double)x - f.a1 * f.w1b - f.a2 * f.w2b;
    const double y = f.b0 * w0 + f.b1 * f.w1b + f.b2 * f.w2b;
    f.w2b = f.w1b; f.w1b = w0;
    return (float)y;
}

// ---- CSMeter (dsp/smeter.cpp:62-93) --------------------------------------------------------------
__device__ __forceinline__ void smeter_step(PcSMeter &s, float re, float im)
{
    // 10 log10(|x|^2/32767^2 + 1e-50): the 1e-50 floor only matters for an all-zero sample
    const float pw = (re * re + im * im) * (1.0f / (32767.0f * 32767.0f));
    const double mag = pw > 0.f ? 10.0 * (double)log10f(pw) : -500.0;
    s.att_ave = (1.0 - s.att_a) * s.att_ave + s.att_a * mag;
    s.dec_ave = (1.0 - s.dec_a) * s.dec_ave + s.dec_a * mag;
    if (s.att_ave > s.dec_ave) { s.ave_mag = s.att_ave; s.dec_ave = s.att_ave; }
    else s.ave_mag = s.dec_ave;
    if (mag > s.peak_mag) s.peak_mag = mag;
}

// ---- CAgc (dsp/agc.cpp:174-296 / 301-401) ----------------------------------------------------------
// one tracker step; mag ring in HBM; returns the gain for the delayed sample
__device__ __forceinline__ float agc_track(PcAgc &a, float *ring, float mag)
{
    const float oldest = ring[a.mag_pos];
    ring[a.mag_pos] = mag;
    if (++a.mag_pos >= a.win_n) a.mag_pos = 0;
    float peak = (float)a.peak;
    if (mag > peak) {
        peak = mag;
    } else if (oldest == peak) {                 // the evicted sample was the peak: rescan (:220-230)
        peak = -8.0f;
        for (int i = 0; i < a.win_n; i++) { const float v = ring[i]; if (v > peak) peak = v; }
    }
    a.peak = peak;
    const double pk = peak;
    if (pk > a.attack_ave) a.attack_ave = (1.0 - a.att_rise) * a.attack_ave + a.att_rise * pk;
    else                   a.attack_ave = (1.0 - a.att_fall) * a.attack_ave + a.att_fall * pk;
    if (a.hang) {
        if (pk > a.decay_ave) {
            a.decay_ave = (1.0 - a.dec_rise) * a.decay_ave + a.dec_rise * pk;
            a.hang_timer = 0;
        } else if (a.hang_timer < a.hang_time) {
            a.hang_timer++;
        } else {
            a.decay_ave = (1.0 - a.dec_fall) * a.decay_ave + a.dec_fall * pk;
        }
    } else {
        if (pk > a.decay_ave) a.decay_ave = (1.0 - a.dec_rise) * a.decay_ave + a.dec_rise * pk;
        else                  a.decay_ave = (1.0 - a.dec_fall) * a.decay_ave + a.dec_fall * pk;
    }
    const double m = a.attack_ave > a.decay_ave ? a.attack_ave : a.decay_ave;
    if (m <= a.knee) return (float)a.fixed_gain;
    return 0.7f * exp10f((float)(m * (a.gain_slope - 1.0)));
}
__device__ __forceinline__ void agc_cpx(PcAgc &a, float *dly, float *ring, float &re, float &im)
{
    if (!a.on) { re *= (float)a.manual_gain; im *= (float)a.manual_gain; return; }
    const float dr = dly[2 * a.dly_pos], di = dly[2 * a.dly_pos + 1];
    dly[2 * a.dly_pos] = re; dly[2 * a.dly_pos + 1] = im;
    if (++a.dly_pos >= a.dly_n) a.dly_pos = 0;
    float mag = fabsf(re);
    const float mim = fabsf(im);
    if (mim > mag) mag = mim;
    mag = log10f(mag + 3.2767e-4f) - 4.51543987f;          // log10(32767)
    const float g = agc_track(a, ring, mag);
    re = dr * g; im = di * g;
}
__device__ __forceinline__ float agc_real(PcAgc &a, float *dly, float *ring, float x)
{
    if (!a.on) return x * (float)a.manual_gain;
    const float d = dly[2 * a.dly_pos];
    dly[2 * a.dly_pos] = x;
    if (++a.dly_pos >= a.dly_n) a.dly_pos = 0;
    const float g = agc_track(a, ring, log10f(fabsf(x) + 3.2767e-4f) - 4.51543987f);
    return d * g;
}

// ---- second-order PLL step shared by SAM and FM (samdemod.cpp:83-97, fmdemod.cpp:166-184) ------
// rotates x by sgn*phase, returns the rotated sample, advances phase/freq
__device__ __forceinline__ void pll_step(double &phase, double &freq, double lo, double hi, double alpha,
                                         double beta, float sgn, float xr, float xi, float &tr, float &ti,
                                         float &err_out)
{
    // the reference wraps its fp64 phase once per call; wrap here every sample so that the fp32
    // sincos sees a small argument (same value of sin/cos)
    float s, c;
    sincosf((float)phase, &s, &c);
    s *= sgn;
    tr = c * xr - s * xi;
    ti = c * xi + s * xr;
    const float err = -sgn * atan2f(ti, tr);
    freq += beta * (double)err;
    if (freq > hi) freq = hi;
    else if (freq < lo) freq = lo;
    phase += freq + alpha * (double)err;
    if (phase > 3.14159265358979323846) phase -= kTwoPiD;
    else if (phase < -3.14159265358979323846) phase += kTwoPiD;
    err_out = err;
}

__global__ __launch_bounds__(64)
void postchain_kernel(PcArgs a)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= a.channels) return;
    PcChannel &C = a.chan[ch];
    float *dly = a.agc_dly + (long)ch * PC_AGC_RING * 2;
    float *ring = a.agc_mag + (long)ch * PC_AGC_RING;
    const float *in = a.in + 2 * (long)ch * a.in_stride;
    const bool stereo = a.flags & PC_STEREO;
    float *out = a.out + (stereo ? 2 : 1) * (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    float *scr = a.scratch + (long)ch * a.scratch_stride;
    const int mode = (a.flags & PC_DO_DEMOD) ? C.mode : PC_MODE_NONE;

    // scalar state into registers for the duration of the call
    PcSMeter sm = C.sm;
    PcAgc agc = C.agc;

    for (int b = 0; b < a.nbursts; b++) {
        const float *x = in + 2 * (long)b * a.burst;
        float *y = out + (stereo ? 2 : 1) * (long)b * a.burst;
        const int n = a.burst;
        if (mode == PC_MODE_FM) {
            PcFm &F = C.fm;
            double phase = F.phase, freq = F.freq, dc = F.err_dc;
            for (int i = 0; i < n; i++) {
                float re = x[2 * i], im = x[2 * i + 1];
                if (a.flags & PC_DO_SMETER) smeter_step(sm, re, im);
                if (a.flags & PC_DO_AGC) agc_cpx(agc, dly, ring, re, im);
                float tr, ti, err;
                pll_step(phase, freq, F.lo, F.hi, F.alpha, F.beta, 1.0f, re, im, tr, ti, err);
                dc = (1.0 - F.dc_alpha) * dc + F.dc_alpha * freq;
                scr[i] = (float)((freq - dc) * F.out_gain);
            }
            F.phase = phase; F.freq = freq; F.err_dc = dc;
            // noise squelch: HP FIR -> |.| EMA over the burst, ONE hysteresis decision (:113-152)
            if (n <= 16384) {
                double ave = F.sq_ave;
                for (int i = 0; i < n; i++) {
                    const float hp = fir_real(F.hp, scr[i]);
                    ave = (1.0 - F.sq_alpha) * ave + F.sq_alpha * (double)fabsf(hp);
                }
                F.sq_ave = ave;
                if (0 == F.sq_thresh) F.squelched = 1;
                else if (F.squelched) { if (ave < (F.sq_thresh - 100.0)) F.squelched = 0; }
                else { if (ave >= (F.sq_thresh + 100.0)) F.squelched = 1; }
                if (F.squelched) for (int i = 0; i < n; i++) scr[i] = 0.f;
                else for (int i = 0; i < n; i++) scr[i] = iir_a(F.lp, scr[i]);
            }
            if (stereo) for (int i = 0; i < n; i++) { y[2 * i] = scr[i]; y[2 * i + 1] = scr[i]; }
            else for (int i = 0; i < n; i++) y[i] = scr[i];
        } else if (mode == PC_MODE_SAM) {
            PcSam &S = C.sam;
            double phase = S.phase, freq = S.freq, z1 = S.z1, y1 = S.y1;
            const float sgn = stereo ? 1.0f : -1.0f;       // mono: e^{-j phi}, stereo: e^{+j phi}
            for (int i = 0; i < n; i++) {
                float re = x[2 * i], im = x[2 * i + 1];
                if (a.flags & PC_DO_SMETER) smeter_step(sm, re, im);
                if (a.flags & PC_DO_AGC) agc_cpx(agc, dly, ring, re, im);
                float tr, ti, err;
                pll_step(phase, freq, S.lo, S.hi, S.alpha, S.beta, sgn, re, im, tr, ti, err);
                const double z0 = (double)tr + z1 * 0.99;
                if (stereo) {
                    const double y0 = (double)ti + y1 * 0.99;
                    float orr = (float)(z0 - z1), oi = (float)(y0 - y1);
                    y1 = y0;
                    fir_cpx(S.fir, orr, oi);
                    y[2 * i] = orr + oi;                    // lower sideband -> left
                    y[2 * i + 1] = orr - oi;                // upper sideband -> right
                } else {
                    y[i] = (float)(z0 - z1);
                }
                z1 = z0;
            }
            S.phase = phase; S.freq = freq; S.z1 = z1; S.y1 = y1;
        } else if (mode == PC_MODE_AM) {
            PcAm &A = C.am;
            double z1 = A.z1;
            for (int i = 0; i < n; i++) {
                float re = x[2 * i], im = x[2 * i + 1];
                if (a.flags & PC_DO_SMETER) smeter_step(sm, re, im);
                if (a.flags & PC_DO_AGC) agc_cpx(agc, dly, ring, re, im);
                const double mag = (double)sqrtf(re * re + im * im);
                const double z0 = mag + z1 * 0.99;
                float v = (float)(z0 - z1);
                z1 = z0;
                if (stereo) {
                    float vr = v, vi = v;
                    fir_cpx(A.fir, vr, vi);
                    y[2 * i] = vr; y[2 * i + 1] = vi;
                } else {
                    y[i] = fir_real(A.fir, v);
                }
            }
            A.z1 = z1;
        } else {
            // SSB / CW (real part or copy), or no demodulator (AGC / S-meter only)
            const bool demod = mode >= PC_MODE_USB;
            for (int i = 0; i < n; i++) {
                float re = x[2 * i], im = x[2 * i + 1];
                if (a.flags & PC_DO_SMETER) smeter_step(sm, re, im);
                if (a.flags & PC_DO_AGC) {
                    if (a.flags & PC_AGC_REAL) re = agc_real(agc, dly, ring, re);
                    else agc_cpx(agc, dly, ring, re, im);
                }
                if (stereo || !demod) {
                    if (a.out) { y[2 * i] = re; y[2 * i + 1] = im; }
                } else {
                    y[i] = re;
                }
            }
        }
    }
    C.sm = sm;
    C.agc = agc;
}

// stand-alone CFir / CIir objects (one lane each): op 0 FIR real, 1 FIR complex, 2 IIR real, 3 IIR complex
__global__ void filter_leaf_kernel(PcFir *fir, PcIir *iir, const float *in, float *out, int n, int op)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == 0) {
        for (int i = 0; i < n; i++) out[i] = fir_real(*fir, in[i]);
    } else if (op == 1) {
        for (int i = 0; i < n; i++) {
            float re = in[2 * i], im = in[2 * i + 1];
            fir_cpx(*fir, re, im);
            out[2 * i] = re; out[2 * i + 1] = im;
        }
    } else if (op == 2) {
        for (int i = 0; i < n; i++) out[i] = iir_a(*iir, in[i]);
    } else {
        for (int i = 0; i < n; i++) { out[2 * i] = iir_a(*iir, in[2 * i]); out[2 * i + 1] = iir_b(*iir, in[2 * i + 1]); }
    }
}

hipError_t filter_leaf_launch(PcFir *fir, PcIir *iir, const float *in, float *out, int n, int op, hipStream_t stream)
{
    hipLaunchKernelGGL(filter_leaf_kernel, dim3(1), dim3(64), 0, stream, fir, iir, in, out, n, op);
    return hipGetLastError();
}

hipError_t postchain_launch(const PcArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(postchain_kernel, dim3((a.channels + 63) / 64), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace csdr
