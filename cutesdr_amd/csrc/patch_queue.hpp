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
// Two arenas are used in turn; an arena is rewritten only when the kernel that read it has finished (its own event --
// normally long past).  Patches are 4-byte granular.  A FILL patch writes one 32-bit word over a range (ring clears).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "capi_common.hpp"

namespace csdr {

struct PatchDesc {
    unsigned long long dst;      // device address
    unsigned long long src;      // byte offset of the data in the arena; fill patches: the 32-bit word
    unsigned bytes;              // multiple of 4
    unsigned fill;               // != 0: write `src` (low 32 bits) over the range
};
hipError_t patch_apply_launch(const PatchDesc *d_list, const unsigned char *d_arena, int n, hipStream_t s);

struct PatchQueue {
    struct Arena {
        unsigned char *p = nullptr;      // pinned host memory, device-mapped
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    Arena arena[2];
    int cur = 0;
    size_t used = 0;
    std::vector<PatchDesc> list;

    ~PatchQueue()
    {
        for (Arena &a : arena) {
            if (a.done) { if (a.in_flight) (void)hipEventSynchronize(a.done); (void)hipEventDestroy(a.done); }
            if (a.p) (void)hipHostFree(a.p);
        }
    }
    bool empty() const { return list.empty(); }
    // room for `bytes` more in the current arena (data first, the descriptor list is appended at flush)
    int reserve(size_t bytes)
    {
        Arena &a = arena[cur];
        if (a.in_flight) {                       // the kernel that read this arena two flushes ago: long done, normally
            CSDR_HIP(hipEventSynchronize(a.done));
            a.in_flight = false;
        }
        const size_t need = used + bytes;
        if (need <= a.cap) return CSDR_OK;
        size_t ncap = a.cap ? a.cap : 4096;
        while (ncap < need) ncap *= 2;
        unsigned char *q = nullptr;
        if (hipHostMalloc((void **)&q, ncap, hipHostMallocDefault) != hipSuccess)
            return fail(CSDR_ENOMEM, "hipHostMalloc(%zu) failed", ncap);
        if (a.p) { memcpy(q, a.p, used); (void)hipHostFree(a.p); }
        a.p = q; a.cap = ncap;
        return CSDR_OK;
    }
    int add(void *dst, const void *src, size_t bytes)
    {
        if (bytes == 0) return CSDR_OK;
        if ((bytes & 3) || ((uintptr_t)dst & 3)) return fail(CSDR_EINVAL, "patches are 4-byte granular");
        // the patches of one flush are applied by concurrent workgroups: a second patch of the same words (a receiver
        // retuned twice between two calls) replaces the first one's data instead of racing with it.  (Callers patch a
        // field group always as the same range, so ranges are equal or disjoint.)
        for (PatchDesc &d : list)
            if (!d.fill && d.dst == (unsigned long long)(uintptr_t)dst && d.bytes == (unsigned)bytes) {
                memcpy(arena[cur].p + d.src, src, bytes);
                return CSDR_OK;
            }
        const int rc = reserve((bytes + 15) & ~(size_t)15);
        if (rc) return rc;
        memcpy(arena[cur].p + used, src, bytes);
        list.push_back(PatchDesc{(unsigned long long)(uintptr_t)dst, (unsigned long long)used, (unsigned)bytes, 0u});
        used += (bytes + 15) & ~(size_t)15;
        return CSDR_OK;
    }
    int add_fill(void *dst, unsigned word, size_t bytes)
    {
        if (bytes == 0) return CSDR_OK;
        if ((bytes & 3) || ((uintptr_t)dst & 3)) return fail(CSDR_EINVAL, "patches are 4-byte granular");
        list.push_back(PatchDesc{(unsigned long long)(uintptr_t)dst, (unsigned long long)word, (unsigned)bytes, 1u});
        return CSDR_OK;
    }
    // everything queued so far, applied in `s`'s order; the queue is empty afterwards
    int flush(hipStream_t s)
    {
        if (list.empty()) return CSDR_OK;
        const size_t lbytes = list.size() * sizeof(PatchDesc);
        const size_t loff = (used + 15) & ~(size_t)15;
        used = loff;
        int rc = reserve(lbytes);
        if (rc) return rc;
        Arena &a = arena[cur];
        memcpy(a.p + loff, list.data(), lbytes);
        void *dp = nullptr;
        CSDR_HIP(hipHostGetDevicePointer(&dp, a.p, 0));
        const unsigned char *base = (const unsigned char *)dp;
        CSDR_HIP(patch_apply_launch((const PatchDesc *)(base + loff), base, (int)list.size(), s));
        if (!a.done) CSDR_HIP(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
        CSDR_HIP(hipEventRecord(a.done, s));
        a.in_flight = true;
        list.clear();
        used = 0;
        cur ^= 1;
        return CSDR_OK;
    }
};

}  // namespace csdr
