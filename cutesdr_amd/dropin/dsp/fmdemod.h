// dsp/fmdemod.h drop-in: class CFmDemod (reference dsp/fmdemod.h:17-54).
#ifndef FMDEMOD_H
#define FMDEMOD_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#include "dsp/fir.h"        // as the reference header does (dsp/fmdemod.h:11-12)
#include "dsp/iir.h"

#define MAX_SQBUF_SIZE 16384

class CFmDemod
{
public:
    CFmDemod(TYPEREAL samplerate) : m_h(csdr_dropin_handle(csdr_fmdemod_create(CSDR_DEVICE, samplerate), "CFmDemod")) {}
    ~CFmDemod() { csdr_fmdemod_destroy(m_h); }
    CFmDemod(const CFmDemod &) = delete;
    CFmDemod &operator=(const CFmDemod &) = delete;
    int ProcessData(int InLength, TYPEREAL FmBW, TYPECPX *pInData, TYPECPX *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_fmdemod_process_stereo(m_h, InLength, FmBW, &pInData->re, &pOutData->re), "CFmDemod::ProcessData"); }
    int ProcessData(int InLength, TYPEREAL FmBW, TYPECPX *pInData, TYPEREAL *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_fmdemod_process_mono(m_h, InLength, FmBW, &pInData->re, pOutData), "CFmDemod::ProcessData"); }
    void SetSquelch(int Value) { CSDR_LOCK(); csdr_dropin_count(csdr_fmdemod_set_squelch(m_h, Value), "CFmDemod::SetSquelch"); }
private:
    csdr_fmdemod *m_h;
    std::mutex m_Mutex;
};
#endif  // FMDEMOD_H
