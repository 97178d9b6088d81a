// patch_queue.hpp -- the control plane without device-wide synchronisation (VERDICT r5 task 6).
//
// The reference's setters are a mutex and a few stores (dsp/demodulator.h:68-69: the GUI calls SetDemodFreq on every mouse
// move); their effect is seen by the NEXT ProcessData.  Here a setter computes the new parameter words on the host and
// queues them as PATCHES -- (device address, bytes) pairs whose data sit in a pinned, device-mapped arena -- and the next
// process call applies the whole queue with ONE small kernel on ITS OWN stream, in front of its first launch: ordered
// behind everything that call's predecessors still have in flight on that stream, ordered before everything that reads
// the new words, and nobody else's work is waited for.  (Until round 5 every retune ended in hipDeviceSynchronize(): with
// 256 receivers per GPU one user dragging one frequency stalled everyone.)
//
// Two arenas (lists of pinned chunks, never freed while the object lives) are used in turn; an arena is rewritten only when
// the kernel that read it has finished (its own event -- normally long past).  Patches are 4-byte granular.  A FILL patch
// writes one 32-bit word over a range (ring clears).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "capi_common.hpp"

namespace csdr {

struct PatchDesc {
    unsigned long long dst;      // device address
    unsigned long long src;      // device address of the data (in a pinned chunk); fill patches: the 32-bit word
    unsigned bytes;              // multiple of 4
    unsigned fill;               // != 0: write `src` (low 32 bits) over the range
};
hipError_t patch_apply_launch(const PatchDesc *d_list, int n, hipStream_t s);

struct PatchQueue {
    // pinned, device-mapped memory in CHUNKS that are never freed or moved while the object lives: growing an arena by
    // reallocation meant hipHostFree, and hipHostFree waits for the whole device -- the very wait this file removes
    struct Chunk { unsigned char *p = nullptr, *dev = nullptr; size_t cap = 0, used = 0; };
    struct Arena {
        std::vector<Chunk> chunks;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    static constexpr size_t CHUNK = 1u << 20;
    Arena arena[2];
    int cur = 0;
    std::vector<PatchDesc> list;
    std::vector<unsigned char *> host_of;      // per list entry: where its data sits on the host side (same-words replacement)

    ~PatchQueue()
    {
        for (Arena &a : arena) {
            if (a.done) { if (a.in_flight) (void)hipEventSynchronize(a.done); (void)hipEventDestroy(a.done); }
            for (Chunk &c : a.chunks) if (c.p) (void)hipHostFree(c.p);
        }
    }
    bool empty() const { return list.empty(); }
    // `bytes` of pinned memory in the current arena: host pointer and its device address
    int take(size_t bytes, unsigned char **host, unsigned char **dev)
    {
        Arena &a = arena[cur];
        if (a.in_flight) {                       // the kernel that read this arena two flushes ago: long done, normally
            CSDR_HIP(hipEventSynchronize(a.done));
            a.in_flight = false;
            for (Chunk &c : a.chunks) c.used = 0;
        }
        bytes = (bytes + 15) & ~(size_t)15;
        for (Chunk &c : a.chunks)
            if (c.cap - c.used >= bytes) { *host = c.p + c.used; *dev = c.dev + c.used; c.used += bytes; return CSDR_OK; }
        Chunk c;
        c.cap = bytes > CHUNK ? bytes : CHUNK;
        if (hipHostMalloc((void **)&c.p, c.cap, hipHostMallocDefault) != hipSuccess)
            return fail(CSDR_ENOMEM, "hipHostMalloc(%zu) failed", c.cap);
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c.p, 0) != hipSuccess) { (void)hipHostFree(c.p); return fail(CSDR_EHIP, "hipHostGetDevicePointer failed"); }
        c.dev = (unsigned char *)dp;
        c.used = bytes;
        *host = c.p; *dev = c.dev;
        a.chunks.push_back(c);
        return CSDR_OK;
    }
    int add(void *dst, const void *src, size_t bytes)
    {
        if (bytes == 0) return CSDR_OK;
        if ((bytes & 3) || ((uintptr_t)dst & 3)) return fail(CSDR_EINVAL, "patches are 4-byte granular");
        // the patches of one flush are applied by concurrent workgroups: a second patch of the same words (a receiver
        // retuned twice between two calls) replaces the first one's data instead of racing with it.  (Callers patch a
        // field group always as the same range, so ranges are equal or disjoint.)
        for (size_t i = 0; i < list.size(); i++)
            if (!list[i].fill && list[i].dst == (unsigned long long)(uintptr_t)dst && list[i].bytes == (unsigned)bytes) {
                memcpy(host_of[i], src, bytes);
                return CSDR_OK;
            }
        unsigned char *h = nullptr, *d = nullptr;
        const int rc = take(bytes, &h, &d);
        if (rc) return rc;
        memcpy(h, src, bytes);
        list.push_back(PatchDesc{(unsigned long long)(uintptr_t)dst, (unsigned long long)(uintptr_t)d, (unsigned)bytes, 0u});
        host_of.push_back(h);
        return CSDR_OK;
    }
    int add_fill(void *dst, unsigned word, size_t bytes)
    {
        if (bytes == 0) return CSDR_OK;
        if ((bytes & 3) || ((uintptr_t)dst & 3)) return fail(CSDR_EINVAL, "patches are 4-byte granular");
        if (arena[cur].in_flight) { unsigned char *h, *d; const int rc = take(16, &h, &d); if (rc) return rc; }   // (reclaims the arena)
        list.push_back(PatchDesc{(unsigned long long)(uintptr_t)dst, (unsigned long long)word, (unsigned)bytes, 1u});
        host_of.push_back(nullptr);
        return CSDR_OK;
    }
    // everything queued so far, applied in `s`'s order; the queue is empty afterwards
    int flush(hipStream_t s)
    {
        if (list.empty()) return CSDR_OK;
        unsigned char *h = nullptr, *d = nullptr;
        int rc = take(list.size() * sizeof(PatchDesc), &h, &d);
        if (rc) return rc;
        memcpy(h, list.data(), list.size() * sizeof(PatchDesc));
        CSDR_HIP(patch_apply_launch((const PatchDesc *)d, (int)list.size(), s));
        Arena &a = arena[cur];
        if (!a.done) CSDR_HIP(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
        CSDR_HIP(hipEventRecord(a.done, s));
        a.in_flight = true;
        list.clear(); host_of.clear();
        cur ^= 1;
        return CSDR_OK;
    }
};

}  // namespace csdr
