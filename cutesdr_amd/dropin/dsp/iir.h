// dsp/iir.h drop-in: class CIir (reference dsp/iir.h:17-39).
#ifndef IIR_H
#define IIR_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"

class CIir
{
public:
    CIir() : m_h(csdr_dropin_handle(csdr_iir_create(CSDR_DEVICE), "CIir")) {}
    ~CIir() { csdr_iir_destroy(m_h); }
    CIir(const CIir &) = delete;
    CIir &operator=(const CIir &) = delete;
    void InitLP(TYPEREAL F0Freq, TYPEREAL FilterQ, TYPEREAL SampleRate) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_init(m_h, 0, F0Freq, FilterQ, SampleRate), "CIir::InitLP"); }
    void InitHP(TYPEREAL F0Freq, TYPEREAL FilterQ, TYPEREAL SampleRate) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_init(m_h, 1, F0Freq, FilterQ, SampleRate), "CIir::InitHP"); }
    void InitBP(TYPEREAL F0Freq, TYPEREAL FilterQ, TYPEREAL SampleRate) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_init(m_h, 2, F0Freq, FilterQ, SampleRate), "CIir::InitBP"); }
    void InitBR(TYPEREAL F0Freq, TYPEREAL FilterQ, TYPEREAL SampleRate) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_init(m_h, 3, F0Freq, FilterQ, SampleRate), "CIir::InitBR"); }
    void ProcessFilter(int InLength, TYPEREAL *InBuf, TYPEREAL *OutBuf) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_process_real(m_h, InLength, InBuf, OutBuf), "CIir::ProcessFilter"); }
    void ProcessFilter(int InLength, TYPECPX *InBuf, TYPECPX *OutBuf) { CSDR_LOCK(); csdr_dropin_count(csdr_iir_process_cpx(m_h, InLength, &InBuf->re, &OutBuf->re), "CIir::ProcessFilter"); }
private:
    csdr_iir *m_h;
    std::mutex m_Mutex;
};
#endif  // IIR_H
