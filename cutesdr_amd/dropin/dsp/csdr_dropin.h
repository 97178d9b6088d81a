// csdr_dropin.h -- shared plumbing of the drop-in classes: device selection, error reporting and
// the per-object lock that stands in for the reference's QMutex members.
#ifndef CSDR_DROPIN_H
#define CSDR_DROPIN_H
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "cutesdr_mi.h"

#ifndef CSDR_DEVICE
#define CSDR_DEVICE csdr_dropin_device()
#endif
#ifndef CSDR_FASTFIR_SIZE
#define CSDR_FASTFIR_SIZE 2048      // the reference's CONV_FFT_SIZE (dsp/fastfir.cpp:55); 16384 for the long filter
#endif

inline int csdr_dropin_device()
{
    static int dev = [] { const char *e = std::getenv("CSDR_DEVICE"); return e ? std::atoi(e) : 0; }();
    return dev;
}
// the reference surfaces no errors: counts are >= 0.  New failure classes (HIP errors, no GPU) are
// logged once per call site and reported as "0 samples".
inline int csdr_dropin_count(int rc, const char *what)
{
    if (rc >= 0) return rc;
    std::fprintf(stderr, "cutesdr_mi: %s failed (%d): %s\n", what, rc, csdr_last_error());
    return 0;
}
template <class H> inline H *csdr_dropin_handle(H *h, const char *what)
{
    if (!h) std::fprintf(stderr, "cutesdr_mi: %s failed: %s\n", what, csdr_last_error());
    return h;
}
#endif  // CSDR_DROPIN_H
