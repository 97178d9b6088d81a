// dsp/downconvert.h drop-in: class CDownConvert (reference dsp/downconvert.h:24-120).
#ifndef DOWNCONVERT_H
#define DOWNCONVERT_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

#define MAX_DECSTAGES 10

class CDownConvert
{
public:
    CDownConvert() : m_h(csdr_dropin_handle(csdr_downconvert_create(CSDR_DEVICE), "CDownConvert")) {}
    virtual ~CDownConvert() { csdr_downconvert_destroy(m_h); }
    CDownConvert(const CDownConvert &) = delete;
    CDownConvert &operator=(const CDownConvert &) = delete;

    void SetFrequency(TYPEREAL NcoFreq) { CSDR_LOCK(); csdr_dropin_count(csdr_downconvert_set_frequency(m_h, NcoFreq), "CDownConvert::SetFrequency"); }
    void SetCwOffset(TYPEREAL offset) { CSDR_LOCK(); csdr_dropin_count(csdr_downconvert_set_cw_offset(m_h, offset), "CDownConvert::SetCwOffset"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        return csdr_dropin_count(csdr_downconvert_process(m_h, InLength, &pInData->re, &pOutData->re), "CDownConvert::ProcessData");
    }
    TYPEREAL SetDataRate(TYPEREAL InRate, TYPEREAL MaxBW)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        const double r = csdr_downconvert_set_data_rate(m_h, InRate, MaxBW);
        if (r < 0) { csdr_dropin_count(CSDR_EHIP, "CDownConvert::SetDataRate"); return InRate; }
        return r;
    }

private:
    csdr_downconvert *m_h;
    std::mutex m_Mutex;
};
#endif  // DOWNCONVERT_H
