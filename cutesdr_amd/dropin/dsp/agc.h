// dsp/agc.h drop-in: class CAgc (reference dsp/agc.h:19-62).
#ifndef AGCX_H
#define AGCX_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

#define MAX_DELAY_BUF 2048

class CAgc
{
public:
    CAgc() : m_h(csdr_dropin_handle(csdr_agc_create(CSDR_DEVICE), "CAgc")) {}
    virtual ~CAgc() { csdr_agc_destroy(m_h); }
    CAgc(const CAgc &) = delete;
    CAgc &operator=(const CAgc &) = delete;
    void SetParameters(bool AgcOn, bool UseHang, int Threshold, int ManualGain, int Slope, int Decay, TYPEREAL SampleRate)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        csdr_dropin_count(csdr_agc_set_parameters(m_h, AgcOn, UseHang, Threshold, ManualGain, Slope, Decay, SampleRate), "CAgc::SetParameters");
    }
    void ProcessData(int Length, TYPECPX *pInData, TYPECPX *pOutData)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        csdr_dropin_count(csdr_agc_process_cpx(m_h, Length, &pInData->re, &pOutData->re), "CAgc::ProcessData");
    }
    void ProcessData(int Length, TYPEREAL *pInData, TYPEREAL *pOutData)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        csdr_dropin_count(csdr_agc_process_real(m_h, Length, pInData, pOutData), "CAgc::ProcessData");
    }
private:
    csdr_agc *m_h;
    std::mutex m_Mutex;
};
#endif  // AGCX_H
