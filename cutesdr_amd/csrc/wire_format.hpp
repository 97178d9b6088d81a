// wire_format.hpp -- the datagram sample formats of CUdpThread::OnreadyRead (reference
// interface/netiobase.cpp:479-527) as device loads, so that the kernels at input rate (down-converter, noise
// blanker) read the datagrams as they arrived instead of a float copy made by a separate unpack pass.
// A channel's datagrams are contiguous: [npackets][pkt_len] bytes, 4 header bytes each, then little-endian I,Q
// pairs; pkt_len 1028 = 256 samples of 16 bit, 1444 = 240 samples of 24 bit (scaled by 1/256 onto the 16-bit
// range: value << 8 in an int32, / 65536).  Every 16- and 24-bit value is exact in fp32.
// Offsets are 32-bit and the two sample counts per datagram are compile-time constants inside each branch (a
// division by a variable, or a 64-bit one, costs more than the decode itself at input rate): a channel's
// datagrams of one call stay below 2 GiB.
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

struct WireIn {                          // optional packed input of a kernel; pk == nullptr: complex fp32 input
    const unsigned char *pk;             // [channels][chan_stride] bytes
    long chan_stride;                    // bytes between channels (= npackets * pkt_len)
    int pkt_len, per;                    // 1028 / 256 or 1444 / 240
};

typedef float wf2 __attribute__((ext_vector_type(2)));
typedef float wf4 __attribute__((ext_vector_type(4)));

// sample i of a channel's datagram sequence
__device__ __forceinline__ wf2 wire_sample(const unsigned char *chan, int pkt_len, long i64)
{
    const unsigned i = (unsigned)i64;
    if (pkt_len == 1444) {
        const unsigned q = i / 240u, j = i - q * 240u;
        const unsigned short *h = reinterpret_cast<const unsigned short *>(chan + (q * 1444u + 4u + 6u * j));   // 2-byte aligned
        const unsigned h0 = h[0], h1 = h[1], h2 = h[2];
        const int vi = (int)((h0 << 8) | ((h1 & 0xffu) << 24));
        const int vq = (int)(((h1 >> 8) << 8) | (h2 << 16));
        return wf2{(float)vi * (1.0f / 65536.0f), (float)vq * (1.0f / 65536.0f)};
    }
    const unsigned q = i >> 8, j = i & 255u;
    const unsigned d = *reinterpret_cast<const unsigned *>(chan + (q * 1028u + 4u + 4u * j));
    return wf2{(float)(short)(d & 0xffffu), (float)(short)(d >> 16)};
}

// Samples i, i+1 (i even: a pair never straddles a datagram, both sample counts are even) with aligned 32-bit
// loads, in two halves for kernels that prefetch: wire_pair_fetch issues the loads and returns the raw words
// (2 or 3 of them), wire_pair_decode turns them into two complex samples where they are consumed -- decoding
// at the fetch would make the prefetch wait for its own loads.
__device__ __forceinline__ wf4 wire_pair_fetch(const unsigned char *chan, int pkt_len, long i64)
{
    const unsigned i = (unsigned)i64;
    if (pkt_len == 1444) {
        const unsigned q = i / 240u, j = (i - q * 240u) >> 1;
        const unsigned *w = reinterpret_cast<const unsigned *>(chan + (q * 1444u + 4u + 12u * j));
        return wf4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), 0.f};
    }
    const unsigned q = i >> 8, j = (i & 255u) >> 1;
    const unsigned *w = reinterpret_cast<const unsigned *>(chan + (q * 1028u + 4u + 8u * j));
    return wf4{__uint_as_float(w[0]), __uint_as_float(w[1]), 0.f, 0.f};
}
__device__ __forceinline__ wf4 wire_pair_decode(wf4 r, int pkt_len)
{
    const unsigned d0 = __float_as_uint(r.x), d1 = __float_as_uint(r.y), d2 = __float_as_uint(r.z);
    if (pkt_len == 1444) {
        const int i0 = (int)(d0 << 8);
        const int q0 = (int)(((d0 >> 24) << 8) | (d1 << 16));
        const int i1 = (int)(((d1 >> 16) << 8) | (d2 << 24));
        const int q1 = (int)(d2 & 0xffffff00u);
        return wf4{(float)i0 * (1.0f / 65536.0f), (float)q0 * (1.0f / 65536.0f), (float)i1 * (1.0f / 65536.0f),
                   (float)q1 * (1.0f / 65536.0f)};
    }
    return wf4{(float)(short)(d0 & 0xffffu), (float)(short)(d0 >> 16), (float)(short)(d1 & 0xffffu), (float)(short)(d1 >> 16)};
}

}  // namespace csdr
