// dsp/amdemod.h drop-in: class CAmDemod (reference dsp/amdemod.h:14-25).
#ifndef AMDEMOD_H
#define AMDEMOD_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#include "dsp/fir.h"        // as the reference header does (dsp/amdemod.h:11)

class CAmDemod
{
public:
    CAmDemod(TYPEREAL samplerate) : m_h(csdr_dropin_handle(csdr_amdemod_create(CSDR_DEVICE, samplerate), "CAmDemod")) {}
    ~CAmDemod() { csdr_amdemod_destroy(m_h); }
    CAmDemod(const CAmDemod &) = delete;
    CAmDemod &operator=(const CAmDemod &) = delete;
    void SetBandwidth(TYPEREAL Bandwidth) { CSDR_LOCK(); csdr_dropin_count(csdr_amdemod_set_bandwidth(m_h, Bandwidth), "CAmDemod::SetBandwidth"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPEREAL *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_amdemod_process_mono(m_h, InLength, &pInData->re, pOutData), "CAmDemod::ProcessData"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData) { CSDR_LOCK(); return csdr_dropin_count(csdr_amdemod_process_stereo(m_h, InLength, &pInData->re, &pOutData->re), "CAmDemod::ProcessData"); }
private:
    csdr_amdemod *m_h;
    std::mutex m_Mutex;
};
#endif  // AMDEMOD_H
