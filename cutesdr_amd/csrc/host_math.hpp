// host_math.hpp -- fp64 host-side setup math (runs once per parameter change, never per sample).
#pragma once
#include <cmath>
#include <complex>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "ref_constants.hpp"

namespace csdr {

typedef std::complex<double> cd;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;   // K_2PI, dsp/datatypes.h:44
constexpr double kPi = 3.14159265358979323846;

// Tables a design reuses (round 6: a same-mode SetDemod is the control plane's whole cost -- 84 us per call, two thirds of it
// the 2047 twiddles and the 3075 window cosines recomputed every time): cos / sin(2 pi j / n), j < n / 2, per transform size,
// and CFastFIR's window per tap count.  A stage of length len uses every (n / len)-th entry: k / len and (k n / len) / n are
// the same double (a power-of-two scaling), so the values are exactly those of a per-stage cos / sin call.  Built once,
// under a lock; readers take a shared pointer.
inline std::shared_ptr<const std::vector<double>> host_twiddles(size_t n)
{
    static std::mutex m;
    static std::map<size_t, std::shared_ptr<const std::vector<double>>> cache;
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find(n);
    if (it != cache.end()) return it->second;
    auto t = std::make_shared<std::vector<double>>(n);                 // [2 j] = cos, [2 j + 1] = sin, j < n / 2
    for (size_t j = 0; j < n / 2; j++) {
        const double ang = kTwoPi * (double)j / (double)n;
        (*t)[2 * j] = std::cos(ang); (*t)[2 * j + 1] = std::sin(ang);
    }
    cache[n] = t;
    return t;
}
inline std::shared_ptr<const std::vector<double>> fastfir_window(int p)    // Blackman-Nuttall, dsp/fastfir.cpp:93-101
{
    static std::mutex m;
    static std::map<int, std::shared_ptr<const std::vector<double>>> cache;
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find(p);
    if (it != cache.end()) return it->second;
    auto w = std::make_shared<std::vector<double>>(p);
    for (int i = 0; i < p; i++)
        (*w)[i] = refc::FF_WIN_A0 - refc::FF_WIN_A1 * std::cos((kTwoPi * i) / (p - 1)) +
                  refc::FF_WIN_A2 * std::cos((2.0 * kTwoPi * i) / (p - 1)) -
                  refc::FF_WIN_A3 * std::cos((3.0 * kTwoPi * i) / (p - 1));
    cache[p] = w;
    return w;
}

// Unnormalised complex DFT, sign=+1 is the reference's FwdFFT convention (dsp/fft.cpp:416-420).
inline void host_fft(std::vector<cd> &a, int sign)
{
    const size_t n = a.size();
    const auto tw = host_twiddles(n);
    const double *T = tw->data();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        for (size_t k = 0; k < half; k++) {
            const size_t j = k * (n / len);
            const cd w(T[2 * j], sign >= 0 ? T[2 * j + 1] : -T[2 * j + 1]);
            for (size_t b = k; b < n; b += len) {
                const cd u = a[b], v = a[b + half] * w;
                a[b] = u + v;
                a[b + half] = u - v;
            }
        }
    }
}

// CFastFIR::SetupParameters (dsp/fastfir.cpp:178-259) with FFT size n, taps n/2+1:
// Blackman-Nuttall windowed sinc (window :93-101), shifted to the pass-band centre, scaled by
// 1/n, zero padded and forward transformed.  Returns false when the reference's sanity check
// rejects the edges (it then keeps the old taps).
inline bool fastfir_design(int n, double flo, double fhi, double offset, double fs, std::vector<cd> &H)
{
    const int p = n / 2 + 1;
    flo += offset;
    fhi += offset;
    if (flo >= fhi || flo >= fs / 2.0 || flo <= -fs / 2.0 || fhi >= fs / 2.0 || fhi <= -fs / 2.0)
        return false;
    const double nfl = flo / fs, nfh = fhi / fs;
    const double nfc = (nfh - nfl) / 2.0, nfs = kTwoPi * (nfh + nfl) / 2.0;
    const double centre = 0.5 * (double)(p - 1);
    H.assign(n, cd(0.0, 0.0));
    const auto win = fastfir_window(p);
    for (int i = 0; i < p; i++) {
        const double x = (double)i - centre;
        double z;
        if ((double)i == centre) {
            z = 2.0 * nfc;
        } else {
            const double w = (*win)[i];
            z = std::sin(kTwoPi * x * nfc) / (kPi * x) * w;
        }
        H[i] = cd(z * std::cos(nfs * x) / (double)n, z * std::sin(nfs * x) / (double)n);
    }
    host_fft(H, +1);
    return true;
}

}  // namespace csdr
