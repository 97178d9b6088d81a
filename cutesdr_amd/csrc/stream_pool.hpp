// stream_pool.hpp -- the library's process-wide pool of internal HIP streams.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <cstdlib>
#include <tuple>
#include <vector>

// The batch objects' internal streams come from a process-wide pool and go back to it: what a stream costs or gains
// depends on the hardware queue the runtime gave it when it was created, and a strict-mode object made after a
// pipelined one had been destroyed ran 2.1-2.3 ms per C4 call against 1.8 for the same object in a fresh process
// (new streams landing beside the queues the old ones had held).  Reused, a plan group's stream is the same stream
// for every object the process makes.  (An idle pooled stream may still have work of its former owner in flight: a
// stream is in-order, the new owner's work queues behind it.)
namespace csdr {
// A stream also keeps the ROLE it was created for (a plan group's stream, a chained pipeline's post-chain stream, the
// three-stage pipeline's filter / post streams ...): the streams of one priority are not interchangeable -- a strict object
// whose first group ran on a stream that had been created as a post-chain stream took 2.14-2.32 ms per C4 call instead of
// 1.56 (tools/experiments/r6_repro_mode2.py: every second strict object after a pipelined one, by the order in which the
// pool handed the streams out).  With the role in the key an object of either mode gets, role by role, the streams the first
// object of that mode created.
enum StreamRole { STREAM_GROUP = 0, STREAM_POST = 1, STREAM_STAGE_POST = 2, STREAM_STAGE_FIR = 3, STREAM_SIDE = 4 };
struct StreamPool {
    std::mutex m;
    std::map<std::tuple<int, int, int>, std::vector<hipStream_t>> idle;      // (device, priority, role) -> streams
    std::map<hipStream_t, int> role_of;
    std::map<hipStream_t, long> born;                     // creation order: the pool hands out the OLDEST idle stream of a class
    long next_born = 0;
    // The chain runs at its speed only when the DEFAULT stream's hardware queue was made before the streams created here:
    // 1.57-1.66 ms per C4 call then, whatever stream the caller uses, against 1.83-2.0 when the first streams of the process
    // were the host's own or the library's (tools/experiments/r6_own_stream.py: OWN / NULLFIRST; HISTORY round 6).  The runtime
    // makes a stream's queue at its first use, so: four bytes through the default stream, once per device, before the first
    // stream is created.  (The one place the library touches the default stream; at that moment it has no work in flight.)
    std::map<int, bool> touched;
    void touch_default_stream(int device)
    {
        {
            std::lock_guard<std::mutex> g(m);
            if (touched[device]) return;
            touched[device] = true;
        }
        static const bool off = getenv("CSDR_NO_DEFAULT_STREAM_TOUCH") && atoi(getenv("CSDR_NO_DEFAULT_STREAM_TOUCH")) != 0;
        if (off) return;
        void *p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return;
        (void)hipMemsetAsync(p, 0, 4, nullptr);
        (void)hipStreamSynchronize(nullptr);
        (void)hipFree(p);
    }
    hipError_t get(int device, int prio, hipStream_t *out, int role = STREAM_GROUP)
    {
        {
            std::lock_guard<std::mutex> g(m);
            auto &v = idle[std::make_tuple(device, prio, role)];
            // the oldest stream of the class first, whatever order they came back in: the runtime gave the process's first
            // streams its first hardware queues, and a lone object is to run on those -- two objects alive at once and then
            // both dropped left the LATER pair's streams at the front of a first-in-first-out list, and the next object on
            // them ran 1.93-1.98 ms per C4 call instead of 1.57 (tools/experiments/r6_own_stream.py)
            if (!v.empty()) {
                size_t best = 0;
                for (size_t i = 1; i < v.size(); i++) if (born[v[i]] < born[v[best]]) best = i;
                *out = v[best]; v.erase(v.begin() + best); return hipSuccess;
            }
        }
        touch_default_stream(device);
        const hipError_t e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
        if (e == hipSuccess) { std::lock_guard<std::mutex> g(m); role_of[*out] = role; born[*out] = next_born++; }
        return e;
    }
    void put(int device, hipStream_t s)
    {
        int prio = 0;
        if (hipStreamGetPriority(s, &prio) != hipSuccess) { (void)hipStreamDestroy(s); return; }
        std::lock_guard<std::mutex> g(m);
        const auto it = role_of.find(s);
        idle[std::make_tuple(device, prio, it == role_of.end() ? (int)STREAM_GROUP : it->second)].push_back(s);
    }
};
inline StreamPool &stream_pool() { static StreamPool *p = new StreamPool(); return *p; }   // never destroyed: outlives every object
}  // namespace csdr
