// capi_hosttest.hip -- test-only exports of the HOST-side setup math (no GPU needed), so that the
// CPU test suite can check filter design, stage selection and the spectrum-order bookkeeping
// against the test-side checker.  Not part of the public ABI (csdr__ prefix, absent from cutesdr_mi.h).
#include "capi_common.hpp"
#include "host_math.hpp"
#include "dc_host.hpp"
#include "pc_host.hpp"
#include "fastfir_kernels.h"
#include <cstring>

using namespace csdr;

extern "C" {

int csdr__host_fastfir_design(int n, double flo, double fhi, double off, double fs, double *h_out)
{
    std::vector<cd> H;
    if (!fastfir_design(n, flo, fhi, off, fs, H)) return -1;
    memcpy(h_out, H.data(), sizeof(cd) * n);
    return 0;
}
int csdr__host_fastfir_bin_of(int log2n, int t, int r) { return fastfir_bin_of(log2n, t, r); }

int csdr__host_dc_plan(double in_rate, double bw, int *codes, double *out_rate, int *warmup)
{
    DcPlan p = dc_make_plan(in_rate, bw);
    for (int i = 0; i < p.nstages; i++) codes[i] = p.kind[i];
    *out_rate = p.out_rate; *warmup = p.W;
    return p.nstages;
}
// expanded taps of stage s as the dense vector h[0..L-1] (float precision as uploaded)
int csdr__host_dc_stage_taps(double in_rate, double bw, int s, double *h)
{
    DcPlan p = dc_make_plan(in_rate, bw);
    if (s < 0 || s >= p.nstages) return -1;
    const DcStage &st = p.st[s];
    const int L = st.hist + (p.kind[s] == 3 ? 2 : 1);
    for (int i = 0; i < L; i++) h[i] = 0;
    if (st.center >= 0) h[st.center] = st.ccoef;
    for (int q = 0; q < st.npairs; q++) { h[st.a[q]] = st.c[q]; h[st.b[q]] = st.c[q]; }
    return L;
}
void csdr__host_dc_nco(double freq, double cw, double in_rate, unsigned long long *inc, double *stored)
{
    DcHostChan c;
    c.in_rate = in_rate; c.cw_offset = cw;
    c.set_frequency(freq);
    *inc = c.inc; *stored = c.nco_freq;
}
int csdr__host_fir_design(int kind, double scale, double astop, double fpass, double fstop, double fs,
                          double hilbert_off, double *coef, double *icoef, double *qcoef)
{
    HostFir f;
    if (kind == 0) f.init_lp(scale, astop, fpass, fstop, fs);
    else f.init_hp(scale, astop, fpass, fstop, fs);
    if (hilbert_off != 0.0) f.gen_hilbert(hilbert_off);
    for (int i = 0; i < f.ntaps; i++) { coef[i] = f.coef[i]; icoef[i] = f.icoef[i]; qcoef[i] = f.qcoef[i]; }
    return f.ntaps;
}
void csdr__host_iir_design(int kind, double f0, double q, double fs, double *c5)
{
    PcIir f;
    iir_design(f, kind, f0, q, fs);
    c5[0] = f.b0; c5[1] = f.b1; c5[2] = f.b2; c5[3] = f.a1; c5[4] = f.a2;
}
// CAgc::SetParameters derived values: knee, slope, fixed gain, manual gain, 4 alphas, delay, window, hang time
void csdr__host_agc_params(int on, int hang, int thresh, int manual, int slope, int decay, double fs, double *out12)
{
    HostAgc h;
    PcAgc d;
    memset(&d, 0, sizeof(d));
    h.set(d, on != 0, hang != 0, thresh, manual, slope, decay, fs);
    out12[0] = d.knee; out12[1] = d.gain_slope; out12[2] = d.fixed_gain; out12[3] = d.manual_gain;
    out12[4] = d.att_rise; out12[5] = d.att_fall; out12[6] = d.dec_rise; out12[7] = d.dec_fall;
    out12[8] = d.dly_n; out12[9] = d.win_n; out12[10] = d.hang_time; out12[11] = 0;
}

}  // extern "C"
