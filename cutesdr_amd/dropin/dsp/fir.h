// dsp/fir.h drop-in: class CFir (reference dsp/fir.h:20-43).
#ifndef FIR_H
#define FIR_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

#define MAX_NUMCOEF 75

class CFir
{
public:
    CFir() : m_h(csdr_dropin_handle(csdr_fir_create(CSDR_DEVICE), "CFir")) {}
    ~CFir() { csdr_fir_destroy(m_h); }
    CFir(const CFir &) = delete;
    CFir &operator=(const CFir &) = delete;
    void InitConstFir(int NumTaps, const double *pCoef)
    { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_fir_init_const(m_h, NumTaps, pCoef), "CFir::InitConstFir"); }
    int InitLPFilter(TYPEREAL Scale, TYPEREAL Astop, TYPEREAL Fpass, TYPEREAL Fstop, TYPEREAL Fsamprate)
    { std::lock_guard<std::mutex> g(m_Mutex); return csdr_dropin_count(csdr_fir_init_lp(m_h, Scale, Astop, Fpass, Fstop, Fsamprate), "CFir::InitLPFilter"); }
    int InitHPFilter(TYPEREAL Scale, TYPEREAL Astop, TYPEREAL Fpass, TYPEREAL Fstop, TYPEREAL Fsamprate)
    { std::lock_guard<std::mutex> g(m_Mutex); return csdr_dropin_count(csdr_fir_init_hp(m_h, Scale, Astop, Fpass, Fstop, Fsamprate), "CFir::InitHPFilter"); }
    void GenerateHBFilter(TYPEREAL FreqOffset) { CSDR_LOCK(); csdr_dropin_count(csdr_fir_generate_hb(m_h, FreqOffset), "CFir::GenerateHBFilter"); }
    void ProcessFilter(int InLength, TYPEREAL *InBuf, TYPEREAL *OutBuf)
    { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_fir_process_real(m_h, InLength, InBuf, OutBuf), "CFir::ProcessFilter"); }
    void ProcessFilter(int InLength, TYPECPX *InBuf, TYPECPX *OutBuf)
    { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_fir_process_cpx(m_h, InLength, &InBuf->re, &OutBuf->re), "CFir::ProcessFilter"); }
private:
    csdr_fir *m_h;
    std::mutex m_Mutex;
};
#endif  // FIR_H
