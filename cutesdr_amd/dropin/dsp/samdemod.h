// dsp/samdemod.h drop-in: class CSamDemod (reference dsp/samdemod.h:14-32).
#ifndef SAMDEMOD_H
#define SAMDEMOD_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#include "dsp/fir.h"        // as the reference header does (dsp/samdemod.h:11)

class CSamDemod
{
public:
    CSamDemod(TYPEREAL samplerate) : m_h(csdr_dropin_handle(csdr_samdemod_create(CSDR_DEVICE, samplerate), "CSamDemod")) {}
    ~CSamDemod() { csdr_samdemod_destroy(m_h); }
    CSamDemod(const CSamDemod &) = delete;
    CSamDemod &operator=(const CSamDemod &) = delete;
    int ProcessData(int InLength, TYPECPX *pInData, TYPEREAL *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_samdemod_process_mono(m_h, InLength, &pInData->re, pOutData), "CSamDemod::ProcessData"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_samdemod_process_stereo(m_h, InLength, &pInData->re, &pOutData->re), "CSamDemod::ProcessData"); }
private:
    csdr_samdemod *m_h;
    std::mutex m_Mutex;
};
#endif  // SAMDEMOD_H
