// stream_pool.hpp -- the library's process-wide pool of internal HIP streams.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <vector>

// The batch objects' internal streams come from a process-wide pool and go back to it: what a stream costs or gains
// depends on the hardware queue the runtime gave it when it was created, and a strict-mode object made after a
// pipelined one had been destroyed ran 2.1-2.3 ms per C4 call against 1.8 for the same object in a fresh process
// (new streams landing beside the queues the old ones had held).  Reused, a plan group's stream is the same stream
// for every object the process makes.  (An idle pooled stream may still have work of its former owner in flight: a
// stream is in-order, the new owner's work queues behind it.)
namespace csdr {
struct StreamPool {
    std::mutex m;
    std::map<std::pair<int, int>, std::vector<hipStream_t>> idle;      // (device, priority) -> streams
    hipError_t get(int device, int prio, hipStream_t *out)
    {
        {
            std::lock_guard<std::mutex> g(m);
            auto &v = idle[{device, prio}];
            if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        }
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
    }
    void put(int device, hipStream_t s)
    {
        int prio = 0;
        if (hipStreamGetPriority(s, &prio) != hipSuccess) { (void)hipStreamDestroy(s); return; }
        std::lock_guard<std::mutex> g(m);
        idle[{device, prio}].push_back(s);
    }
};
inline StreamPool &stream_pool() { static StreamPool *p = new StreamPool(); return *p; }   // never destroyed: outlives every object
}  // namespace csdr

