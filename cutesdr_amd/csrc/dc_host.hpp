// dc_host.hpp -- host-side (fp64) bookkeeping of CDownConvert: stage selection, tap tables,
// NCO frequency/phase state.  Runs on parameter changes only.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include "downconv_kernels.h"
#include "host_math.hpp"
#include "../../include/csdr_hb_taps.h"

namespace csdr {

struct DcPlan {
    double in_rate, max_bw, out_rate;
    int nstages;
    int kind[DC_MAX_STAGES];          // 3 = CIC3, otherwise half-band length
    DcStage st[DC_MAX_STAGES];
    int W;                            // warm-up length: >= sum_s hist_s 2^s, multiple of 2^nstages
};

// CDownConvert::SetDataRate stage selection (dsp/downconvert.cpp:127-166, filtercoef.h:17-28)
inline DcPlan dc_make_plan(double in_rate, double max_bw)
{
    DcPlan p;
    memset(&p, 0, sizeof(p));
    p.in_rate = in_rate; p.max_bw = max_bw;
    double f = in_rate;
    int n = 0;
    long need = 0;
    while (max_bw > 0 && f > (max_bw / csdr_hb_maxbw[CSDR_HB_NUM_FILTERS - 1]) && f > refc::DCV_MIN_OUTPUT_RATE &&
           n < DC_MAX_STAGES) {
        DcStage &s = p.st[n];
        if (f >= (max_bw / CSDR_CIC3_MAXBW)) {
            // y[j] = .125 (x[2j+1] + x[2j-2] + 3 (x[2j-1] + x[2j])), history 2 (:453-454)
            p.kind[n] = 3;
            s.hist = 2; s.npairs = 2; s.center = -1; s.ccoef = 0.f;
            s.a[0] = 0; s.b[0] = 3; s.c[0] = 0.125f;
            s.a[1] = 1; s.b[1] = 2; s.c[1] = 0.375f;
        } else {
            for (int k = 0; k < CSDR_HB_NUM_FILTERS; k++) {
                if (f >= (max_bw / csdr_hb_maxbw[k])) {
                    double h[CSDR_HB_MAX_LEN];
                    const int L = csdr_hb_expand(k, h);
                    p.kind[n] = L;
                    s.hist = L - 1; s.center = (L - 1) / 2; s.ccoef = (float)h[(L - 1) / 2];
                    s.npairs = 0;
                    for (int i = 0; i < (L - 1) / 2; i += 2) {     // even taps, symmetric
                        s.a[s.npairs] = (short)i; s.b[s.npairs] = (short)(L - 1 - i);
                        s.c[s.npairs] = (float)h[i];
                        s.npairs++;
                    }
                    break;
                }
            }
        }
        need += (long)s.hist << n;
        n++;
        f /= 2.0;
    }
    p.nstages = n;
    p.out_rate = f;
    // whole tiles: a segment's warm-up then runs through the kernel's complete-tile path only (a short tile takes
    // the general stage code, several times a complete tile's cost), and 512 is a multiple of every 2^n here
    const long unit = DC_TILE_SAMPLES > (1L << n) ? DC_TILE_SAMPLES : (1L << n);
    p.W = n ? (int)((need + unit - 1) / unit * unit) : 0;
    return p;
}

// amplitude of the reference NCO phasor: a_0 = 1, a_{n+1} = a_n (1.95 - a_n^2) (downconvert.cpp:210-216)
inline void dc_amp_table(float *amp, int n)
{
    double a = 1.0;
    for (int i = 0; i < n; i++) { amp[i] = (float)a; a = a * (1.95 - a * a); }
}

struct DcHostChan {
    double nco_freq = 0.0, cw_offset = 0.0, in_rate = 100000.0, max_bw = 10000.0, out_rate = 0.0;
    unsigned long long phase = 0, inc = 0, age = 0;
    // CDownConvert::SetFrequency (:98-107): the CW offset is folded into the stored frequency
    void set_frequency(double f)
    {
        nco_freq = f + cw_offset;
        double turns = nco_freq / in_rate;
        turns -= std::floor(turns + 0.5);
        inc = (unsigned long long)(long long)std::llround(turns * 18446744073709551616.0 * 0.5) * 2ull;
    }
};

}  // namespace csdr
