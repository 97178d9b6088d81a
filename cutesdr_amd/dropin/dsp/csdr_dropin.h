// csdr_dropin.h -- shared plumbing of the drop-in classes: device selection, error reporting and
// the per-object lock that stands in for the reference's QMutex members.
#ifndef CSDR_DROPIN_H
#define CSDR_DROPIN_H
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "cutesdr_mi.h"

// The reference headers include <QMutex> where the class owns one (agc.h:14, downconvert.h:15, fastfir.h:17,
// fft.h:19, fir.h:17, noiseproc.h:14) and host code may rely on that; CSDR_DROPIN_QT is set by datatypes.h
// when Qt is on the include path.  The lock itself is a std::mutex either way (non-recursive, like QMutex):
// every drop-in method that reaches the C ABI holds its object's lock, so a setter on the GUI thread and
// ProcessData on the IQ thread never run inside one handle at the same time.
#define CSDR_LOCK() std::lock_guard<std::mutex> csdr_guard_(m_Mutex)

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
