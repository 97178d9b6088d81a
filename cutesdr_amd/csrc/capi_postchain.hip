// capi_postchain.hip -- C ABI of the sample-rate leaf objects: CAgc, CSMeter, CFir, CIir and the
// AM / SAM / FM / SSB demodulators (single-channel host forms), plus the shared PcUnit plumbing
// used by the full CDemodulator chain (capi_demod.hip).
#include "capi_common.hpp"
#include "pc_unit.hpp"

using namespace csdr;

// ------------------------------------------------------------------------------------------------
// Host-form helper: one channel, caller hands interleaved doubles
// ------------------------------------------------------------------------------------------------
struct PcHostObj {
    PcUnit u;
    float *d_in = nullptr, *d_out = nullptr;
    size_t cap = 0;                          // samples
    std::vector<float> st;
    int ensure(size_t n)
    {
        if (n <= cap) return CSDR_OK;
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        d_in = d_out = nullptr; cap = 0;
        CSDR_HIP(hipMalloc((void **)&d_in, n * 8));
        CSDR_HIP(hipMalloc((void **)&d_out, n * 8));
        cap = n;
        return CSDR_OK;
    }
    ~PcHostObj()
    {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
    }
    // in: n samples (complex pairs if in_cpx else reals, widened to (x,0)); out: complex pairs if
    // out_cpx else reals.  One burst of n samples through the stages selected by flags.
    int run(int flags, int n, const double *in, bool in_cpx, double *out, bool out_cpx)
    {
        if (n <= 0) return 0;
        if (!device_ok(u.device)) return CSDR_EHIP;
        int rc = ensure((size_t)n);
        if (rc) return rc;
        st.resize(2 * (size_t)n);
        if (in_cpx) for (size_t i = 0; i < 2 * (size_t)n; i++) st[i] = (float)in[i];
        else for (size_t i = 0; i < (size_t)n; i++) { st[2 * i] = (float)in[i]; st[2 * i + 1] = 0.f; }
        CSDR_HIP(hipMemcpy(d_in, st.data(), (size_t)n * 8, hipMemcpyHostToDevice));
        rc = u.run(flags, d_in, n, d_out, n, 1, n, nullptr);
        if (rc) return rc;
        if (!out) return n;
        const bool dev_cpx = (flags & PC_STEREO) || !(flags & PC_DO_DEMOD) || u.h[0].mode == PC_MODE_NONE;
        const size_t nf = dev_cpx ? 2 * (size_t)n : (size_t)n;
        CSDR_HIP(hipMemcpy(st.data(), d_out, nf * 4, hipMemcpyDeviceToHost));
        if (out_cpx) {
            if (dev_cpx) for (size_t i = 0; i < 2 * (size_t)n; i++) out[i] = (double)st[i];
            else for (size_t i = 0; i < (size_t)n; i++) { out[2 * i] = st[i]; out[2 * i + 1] = st[i]; }
        } else {
            if (dev_cpx) for (size_t i = 0; i < (size_t)n; i++) out[i] = (double)st[2 * i];
            else for (size_t i = 0; i < (size_t)n; i++) out[i] = (double)st[i];
        }
        return n;
    }
};

struct csdr_agc { PcHostObj o; };
struct csdr_smeter { PcHostObj o; };
struct csdr_amdemod { PcHostObj o; double fs; };
struct csdr_samdemod { PcHostObj o; };
struct csdr_fmdemod { PcHostObj o; double fs; };
struct csdr_fir { int device; HostFir h; PcFir *d; float *d_in, *d_out; size_t cap; std::vector<float> st; };
struct csdr_iir { int device; PcIir *d; float *d_in, *d_out; size_t cap; std::vector<float> st; };

template <class T>
static int leaf_filter(T *f, PcFir *fir, PcIir *iir, int n, const double *in, double *out, int op)
{
    if (!f || n < 0 || (n && (!in || !out))) return fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return 0;
    if (!device_ok(f->device)) return CSDR_EHIP;
    const size_t nf = (op & 1) ? 2 * (size_t)n : (size_t)n;
    if (nf > f->cap) {
        if (f->d_in) (void)hipFree(f->d_in);
        if (f->d_out) (void)hipFree(f->d_out);
        f->d_in = f->d_out = nullptr; f->cap = 0;
        CSDR_HIP(hipMalloc((void **)&f->d_in, nf * 4));
        CSDR_HIP(hipMalloc((void **)&f->d_out, nf * 4));
        f->cap = nf;
    }
    f->st.resize(nf);
    for (size_t i = 0; i < nf; i++) f->st[i] = (float)in[i];
    CSDR_HIP(hipMemcpy(f->d_in, f->st.data(), nf * 4, hipMemcpyHostToDevice));
    CSDR_HIP(filter_leaf_launch(fir, iir, f->d_in, f->d_out, n, op, nullptr));
    CSDR_HIP(hipMemcpy(f->st.data(), f->d_out, nf * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nf; i++) out[i] = (double)f->st[i];
    return n;
}

template <class T> static T *make_obj(int device, int mode)
{
    if (!device_ok(device)) return nullptr;
    T *x = new T();
    if (x->o.u.init(device, 1) != CSDR_OK) { delete x; return nullptr; }
    x->o.u.h[0].mode = mode;
    return x;
}

extern "C" {

/* ---------------- CAgc (dsp/agc.h:19-62) ---------------- */
csdr_agc *csdr_agc_create(int device)
{
    csdr_agc *a = make_obj<csdr_agc>(device, PC_MODE_NONE);
    if (a && a->o.u.push(0) != CSDR_OK) { delete a; return nullptr; }
    return a;
}
void csdr_agc_destroy(csdr_agc *a) { delete a; }
int csdr_agc_set_parameters(csdr_agc *a, int on, int use_hang, int threshold, int manual_gain,
                            int slope, int decay, double sample_rate)
{
    if (!a) return fail(CSDR_EINVAL, "bad handle");
    return a->o.u.agc_set(0, on, use_hang, threshold, manual_gain, slope, decay, sample_rate);
}
int csdr_agc_process_cpx(csdr_agc *a, int n, const double *in_iq, double *out_iq)
{
    if (!a || n < 0) return fail(CSDR_EINVAL, "bad argument");
    return a->o.run(PC_DO_AGC, n, in_iq, true, out_iq, true);
}
int csdr_agc_process_real(csdr_agc *a, int n, const double *in, double *out)
{
    if (!a || n < 0) return fail(CSDR_EINVAL, "bad argument");
    return a->o.run(PC_DO_AGC | PC_AGC_REAL, n, in, false, out, false);
}

/* ---------------- CSMeter (dsp/smeter.h:13-28) ---------------- */
csdr_smeter *csdr_smeter_create(int device)
{
    csdr_smeter *s = make_obj<csdr_smeter>(device, PC_MODE_NONE);
    if (s && s->o.u.push(0) != CSDR_OK) { delete s; return nullptr; }
    return s;
}
void csdr_smeter_destroy(csdr_smeter *s) { delete s; }
int csdr_smeter_process(csdr_smeter *s, int n, const double *in_iq, double sample_rate)
{
    if (!s || n < 0) return fail(CSDR_EINVAL, "bad argument");
    int rc = s->o.u.smeter_rate_set(0, sample_rate);
    if (rc) return rc;
    s->o.u.no_output = true;
    rc = s->o.run(PC_DO_SMETER, n, in_iq, true, nullptr, true);
    return rc < 0 ? rc : CSDR_OK;
}
double csdr_smeter_get_peak(csdr_smeter *s) { return s ? s->o.u.smeter_peak(0) : 0.0; }
double csdr_smeter_get_ave(csdr_smeter *s) { return s ? s->o.u.smeter_ave(0) : 0.0; }

/* ---------------- CAmDemod (dsp/amdemod.h:14-25) ---------------- */
csdr_amdemod *csdr_amdemod_create(int device, double sample_rate)
{
    csdr_amdemod *d = make_obj<csdr_amdemod>(device, PC_MODE_AM);
    if (!d) return nullptr;
    d->fs = sample_rate;
    am_init(d->o.u.h[0].am, d->o.u.fir_am[0], sample_rate);
    if (d->o.u.push(0) != CSDR_OK) { delete d; return nullptr; }
    return d;
}
void csdr_amdemod_destroy(csdr_amdemod *d) { delete d; }
int csdr_amdemod_set_bandwidth(csdr_amdemod *d, double bandwidth)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    int rc = d->o.u.pull(0);
    if (rc) return rc;
    am_bandwidth(d->o.u.h[0].am, d->o.u.fir_am[0], d->fs, bandwidth);
    return d->o.u.push(0);
}
int csdr_amdemod_process_mono(csdr_amdemod *d, int n, const double *in_iq, double *out)
{ return d ? d->o.run(PC_DO_DEMOD, n, in_iq, true, out, false) : fail(CSDR_EINVAL, "bad handle"); }
int csdr_amdemod_process_stereo(csdr_amdemod *d, int n, const double *in_iq, double *out_iq)
{ return d ? d->o.run(PC_DO_DEMOD | PC_STEREO, n, in_iq, true, out_iq, true) : fail(CSDR_EINVAL, "bad handle"); }

/* ---------------- CSamDemod (dsp/samdemod.h:14-32) ---------------- */
csdr_samdemod *csdr_samdemod_create(int device, double sample_rate)
{
    csdr_samdemod *d = make_obj<csdr_samdemod>(device, PC_MODE_SAM);
    if (!d) return nullptr;
    sam_init(d->o.u.h[0].sam, d->o.u.fir_sam[0], sample_rate);
    if (d->o.u.push(0) != CSDR_OK) { delete d; return nullptr; }
    return d;
}
void csdr_samdemod_destroy(csdr_samdemod *d) { delete d; }
int csdr_samdemod_process_mono(csdr_samdemod *d, int n, const double *in_iq, double *out)
{ return d ? d->o.run(PC_DO_DEMOD, n, in_iq, true, out, false) : fail(CSDR_EINVAL, "bad handle"); }
int csdr_samdemod_process_stereo(csdr_samdemod *d, int n, const double *in_iq, double *out_iq)
{ return d ? d->o.run(PC_DO_DEMOD | PC_STEREO, n, in_iq, true, out_iq, true) : fail(CSDR_EINVAL, "bad handle"); }

/* ---------------- CFmDemod (dsp/fmdemod.h:17-54) ---------------- */
csdr_fmdemod *csdr_fmdemod_create(int device, double sample_rate)
{
    csdr_fmdemod *d = make_obj<csdr_fmdemod>(device, PC_MODE_FM);
    if (!d) return nullptr;
    d->fs = sample_rate;
    fm_init(d->o.u.h[0].fm, d->o.u.fir_fm[0], sample_rate);
    if (d->o.u.push(0) != CSDR_OK) { delete d; return nullptr; }
    return d;
}
void csdr_fmdemod_destroy(csdr_fmdemod *d) { delete d; }
int csdr_fmdemod_set_squelch(csdr_fmdemod *d, int value)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    int rc = d->o.u.pull(0);
    if (rc) return rc;
    fm_set_squelch(d->o.u.h[0].fm, value);
    return d->o.u.push(0);
}
static int fm_run(csdr_fmdemod *d, int flags, int n, double fm_bw, const double *in, double *out, bool out_cpx)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    if (n > 16384 && (flags & PC_STEREO)) n = 16384;          // member buffer size, fmdemod.h:15
    if (d->o.u.h[0].fm.hp_freq != fm_bw) {                     // fmdemod.cpp:160-164
        int rc = d->o.u.pull(0);
        if (rc) return rc;
        fm_set_bw(d->o.u.h[0].fm, d->o.u.fir_fm[0], d->fs, fm_bw);
        rc = d->o.u.push(0);
        if (rc) return rc;
    }
    return d->o.run(flags, n, in, true, out, out_cpx);
}
int csdr_fmdemod_process_mono(csdr_fmdemod *d, int n, double fm_bw, const double *in_iq, double *out)
{ return fm_run(d, PC_DO_DEMOD, n, fm_bw, in_iq, out, false); }
int csdr_fmdemod_process_stereo(csdr_fmdemod *d, int n, double fm_bw, const double *in_iq, double *out_iq)
{ return fm_run(d, PC_DO_DEMOD | PC_STEREO, n, fm_bw, in_iq, out_iq, true); }
int csdr_fmdemod_get_squelched(csdr_fmdemod *d)
{
    if (!d) return fail(CSDR_EINVAL, "bad handle");
    if (d->o.u.pull(0)) return CSDR_EHIP;
    return d->o.u.h[0].fm.squelched;
}

/* ---------------- CSsbDemod (dsp/ssbdemod.h:13-19): .re / copy (ssbdemod.cpp:48-60) ---------------- */
int csdr_ssbdemod_process_mono(int n, const double *in_iq, double *out)
{
    if (n < 0 || (n && (!in_iq || !out))) return fail(CSDR_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) out[i] = in_iq[2 * i];
    return n;
}
int csdr_ssbdemod_process_stereo(int n, const double *in_iq, double *out_iq)
{
    if (n < 0 || (n && (!in_iq || !out_iq))) return fail(CSDR_EINVAL, "bad argument");
    if (in_iq != out_iq) memmove(out_iq, in_iq, sizeof(double) * 2 * (size_t)n);
    return n;
}

/* ---------------- CFir (dsp/fir.h:20-43) ---------------- */
static int fir_push(csdr_fir *f, bool reset)
{
    PcFir tmp;
    if (!reset) CSDR_HIP(hipMemcpy(&tmp, f->d, sizeof(PcFir), hipMemcpyDeviceToHost));
    f->h.upload(tmp, reset);
    CSDR_HIP(hipMemcpy(f->d, &tmp, sizeof(PcFir), hipMemcpyHostToDevice));
    return CSDR_OK;
}
csdr_fir *csdr_fir_create(int device)
{
    if (!device_ok(device)) return nullptr;
    csdr_fir *f = new csdr_fir();
    f->device = device; f->d = nullptr; f->d_in = f->d_out = nullptr; f->cap = 0;
    f->h.ntaps = 1;                                          // fir.cpp:56-60
    if (hipMalloc((void **)&f->d, sizeof(PcFir)) != hipSuccess || fir_push(f, true) != CSDR_OK) {
        fail(CSDR_ENOMEM, "device allocation failed");
        delete f;
        return nullptr;
    }
    return f;
}
void csdr_fir_destroy(csdr_fir *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->d) (void)hipFree(f->d);
    if (f->d_in) (void)hipFree(f->d_in);
    if (f->d_out) (void)hipFree(f->d_out);
    delete f;
}
int csdr_fir_init_const(csdr_fir *f, int ntaps, const double *coef)
{
    if (!f || ntaps < 1 || !coef) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    f->h.init_const(ntaps, coef);
    return fir_push(f, true);
}
int csdr_fir_init_lp(csdr_fir *f, double scale, double astop, double fpass, double fstop, double fs)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    const int n = f->h.init_lp(scale, astop, fpass, fstop, fs);
    int rc = fir_push(f, true);
    return rc ? rc : n;
}
int csdr_fir_init_hp(csdr_fir *f, double scale, double astop, double fpass, double fstop, double fs)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    const int n = f->h.init_hp(scale, astop, fpass, fstop, fs);
    int rc = fir_push(f, true);
    return rc ? rc : n;
}
int csdr_fir_generate_hb(csdr_fir *f, double freq_offset)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    if (!device_ok(f->device)) return CSDR_EHIP;
    f->h.gen_hilbert(freq_offset);
    return fir_push(f, false);                              // GenerateHBFilter keeps the delay line
}
int csdr_fir_get_taps(csdr_fir *f, double *coef, double *icoef, double *qcoef)
{
    if (!f) return fail(CSDR_EINVAL, "bad handle");
    for (int i = 0; i < f->h.ntaps; i++) {
        if (coef) coef[i] = f->h.coef[i];
        if (icoef) icoef[i] = f->h.icoef[i];
        if (qcoef) qcoef[i] = f->h.qcoef[i];
    }
    return f->h.ntaps;
}
int csdr_fir_process_real(csdr_fir *f, int n, const double *in, double *out)
{ return leaf_filter(f, f ? f->d : nullptr, nullptr, n, in, out, 0); }
int csdr_fir_process_cpx(csdr_fir *f, int n, const double *in_iq, double *out_iq)
{ return leaf_filter(f, f ? f->d : nullptr, nullptr, n, in_iq, out_iq, 1); }

/* ---------------- CIir (dsp/iir.h:17-39) ---------------- */
csdr_iir *csdr_iir_create(int device)
{
    if (!device_ok(device)) return nullptr;
    csdr_iir *f = new csdr_iir();
    f->device = device; f->d = nullptr; f->d_in = f->d_out = nullptr; f->cap = 0;
    PcIir h;
    iir_design(h, 3, 25000, 1000.0, 100000);                // ctor: InitBR (iir.cpp:77-80)
    if (hipMalloc((void **)&f->d, sizeof(PcIir)) != hipSuccess ||
        hipMemcpy(f->d, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
        fail(CSDR_ENOMEM, "device allocation failed");
        delete f;
        return nullptr;
    }
    return f;
}
void csdr_iir_destroy(csdr_iir *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->d) (void)hipFree(f->d);
    if (f->d_in) (void)hipFree(f->d_in);
    if (f->d_out) (void)hipFree(f->d_out);
    delete f;
}
int csdr_iir_init(csdr_iir *f, int kind, double f0, double q, double sample_rate)
{
    if (!f || kind < 0 || kind > 3) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(f->device)) return CSDR_EHIP;
    PcIir h;
    iir_design(h, kind, f0, q, sample_rate);
    CSDR_HIP(hipMemcpy(f->d, &h, sizeof(h), hipMemcpyHostToDevice));
    return CSDR_OK;
}
int csdr_iir_get_coefs(csdr_iir *f, double *b0b1b2a1a2)
{
    if (!f || !b0b1b2a1a2) return fail(CSDR_EINVAL, "bad argument");
    PcIir h;
    CSDR_HIP(hipMemcpy(&h, f->d, sizeof(h), hipMemcpyDeviceToHost));
    b0b1b2a1a2[0] = h.b0; b0b1b2a1a2[1] = h.b1; b0b1b2a1a2[2] = h.b2; b0b1b2a1a2[3] = h.a1; b0b1b2a1a2[4] = h.a2;
    return CSDR_OK;
}
int csdr_iir_process_real(csdr_iir *f, int n, const double *in, double *out)
{ return leaf_filter(f, nullptr, f ? f->d : nullptr, n, in, out, 2); }
int csdr_iir_process_cpx(csdr_iir *f, int n, const double *in_iq, double *out_iq)
{ return leaf_filter(f, nullptr, f ? f->d : nullptr, n, in_iq, out_iq, 3); }

}  // extern "C"
