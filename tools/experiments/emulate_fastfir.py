#!/usr/bin/env python3
"""numpy emulation of fastfir_os_kernel's index algebra (passes F1..I3, twiddles, H order),
to validate the decomposition on the CPU.  Mirrors cutesdr_amd/csrc/fastfir_kernels.hip."""
import sys
import numpy as np


def bitrev(v, R):
    r, m = 0, 1
    while m < R:
        r = (r << 1) | (v & 1); v >>= 1; m <<= 1
    return r


def dif(x, sign):            # natural in -> X[k] at bitrev(k)
    R = len(x); x = x.copy(); ln = R
    while ln >= 2:
        h = ln // 2
        for b in range(0, R, ln):
            for i in range(h):
                u, v = x[b + i], x[b + i + h]
                x[b + i] = u + v
                x[b + i + h] = (u - v) * np.exp(sign * 2j * np.pi * i / ln)
        ln //= 2
    return x


def dit(x, sign):            # y[k] at bitrev(k) in -> natural out
    R = len(x); x = x.copy(); ln = 2
    while ln <= R:
        h = ln // 2
        for b in range(0, R, ln):
            for i in range(h):
                u, v = x[b + i], x[b + i + h] * np.exp(sign * 2j * np.pi * i / ln)
                x[b + i] = u + v
                x[b + i + h] = u - v
        ln *= 2
    return x


def block(xin, Hnat, N, stages=None):
    T, R0 = N // 32, N // 1024
    G = 32 // R0
    lds = np.zeros(N, dtype=complex)
    tw1 = np.exp(2j * np.pi * np.arange(1024) / N)
    tw2 = np.array([[np.exp(2j * np.pi * n * k / 1024) for n in range(32)] for k in range(32)])
    # F1
    for t in range(T):
        for e in range(G):
            n2 = G * t + e
            y = dif(np.array([xin[1024 * n1 + n2] for n1 in range(R0)]), +1)
            for r in range(R0):
                k0 = bitrev(r, R0)
                lds[1024 * k0 + n2] = y[r] * tw1[n2] ** k0
    if stages is not None: stages.append(lds.copy())
    # F2
    new = lds.copy()
    for t in range(T):
        sb, sn = t >> 5, t & 31
        y = dif(np.array([lds[1024 * sb + sn + 32 * n1] for n1 in range(32)]), +1)
        for r in range(32):
            k1 = bitrev(r, 32)
            new[1024 * sb + sn + 32 * k1] = y[r] * tw2[k1, sn]
    lds = new
    if stages is not None: stages.append(lds.copy())
    # F3 + H + I1
    new = lds.copy()
    for t in range(T):
        y = dif(lds[32 * t: 32 * t + 32], +1)
        for r in range(32):
            k2 = bitrev(r, 32)
            k = (t >> 5) + R0 * ((t & 31) + 32 * k2)
            y[r] *= Hnat[k]
        new[32 * t: 32 * t + 32] = dit(y, -1)
    lds = new
    if stages is not None: stages.append(lds.copy())
    # I2
    new = lds.copy()
    for t in range(T):
        sb, sn = t >> 5, t & 31
        y = np.zeros(32, dtype=complex)
        for r in range(32):
            k1 = bitrev(r, 32)
            y[r] = lds[1024 * sb + sn + 32 * k1] * np.conj(tw2[k1, sn])
        y = dit(y, -1)
        for n1 in range(32):
            new[1024 * sb + sn + 32 * n1] = y[n1]
    lds = new
    if stages is not None: stages.append(lds.copy())
    # I3
    out = np.zeros(N, dtype=complex)
    for t in range(T):
        for e in range(G):
            n2 = G * t + e
            y = np.zeros(R0, dtype=complex)
            for r in range(R0):
                k0 = bitrev(r, R0)
                y[r] = lds[1024 * k0 + n2] * np.conj(tw1[n2] ** k0)
            y = dit(y, -1)
            for n1 in range(R0):
                out[1024 * n1 + n2] = y[n1]
    return out


if __name__ == "__main__":
    for N in (2048, 4096, 16384) if len(sys.argv) < 2 else [int(sys.argv[1])]:
        rng = np.random.default_rng(0)
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        H = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        got = block(x, H, N)
        X = N * np.fft.ifft(x)                 # forward = +exponent
        want = np.fft.fft(X * H)               # reverse = -exponent, unnormalised
        print(N, np.abs(got - want).max() / np.abs(want).max())
