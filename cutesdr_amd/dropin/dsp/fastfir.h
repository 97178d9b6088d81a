// dsp/fastfir.h drop-in: class CFastFIR with the reference's public surface
// (reference dsp/fastfir.h:17-44), forwarding to the C ABI of libcutesdr_mi.
#ifndef FASTFIR_H
#define FASTFIR_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#include "dsp/fft.h"        // as the reference header does (dsp/fastfir.h:16)
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

class CFastFIR
{
public:
    CFastFIR() : m_h(csdr_dropin_handle(csdr_fastfir_create(CSDR_DEVICE, CSDR_FASTFIR_SIZE), "CFastFIR")) {}
    virtual ~CFastFIR() { csdr_fastfir_destroy(m_h); }
    CFastFIR(const CFastFIR &) = delete;
    CFastFIR &operator=(const CFastFIR &) = delete;

    void SetupParameters(TYPEREAL FLoCut, TYPEREAL FHiCut, TYPEREAL Offset, TYPEREAL SampleRate)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        const int rc = csdr_fastfir_setup(m_h, FLoCut, FHiCut, Offset, SampleRate);
        if (rc == CSDR_EINVAL) std::fprintf(stderr, "Filter Parameter error\n");      // fastfir.cpp:201
        else csdr_dropin_count(rc, "CFastFIR::SetupParameters");
    }
    int ProcessData(int InLength, TYPECPX *InBuf, TYPECPX *OutBuf)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        return csdr_dropin_count(csdr_fastfir_process(m_h, InLength, &InBuf->re, &OutBuf->re), "CFastFIR::ProcessData");
    }

private:
    csdr_fastfir *m_h;
    std::mutex m_Mutex;
};
#endif  // FASTFIR_H
