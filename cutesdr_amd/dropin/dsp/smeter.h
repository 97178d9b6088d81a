// dsp/smeter.h drop-in: class CSMeter (reference dsp/smeter.h:13-28).
#ifndef SMETER_H
#define SMETER_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"

class CSMeter
{
public:
    CSMeter() : m_h(csdr_dropin_handle(csdr_smeter_create(CSDR_DEVICE), "CSMeter")) {}
    ~CSMeter() { csdr_smeter_destroy(m_h); }
    CSMeter(const CSMeter &) = delete;
    CSMeter &operator=(const CSMeter &) = delete;
    void ProcessData(int length, TYPECPX *pInData, TYPEREAL SampleRate)
    { CSDR_LOCK(); csdr_dropin_count(csdr_smeter_process(m_h, length, &pInData->re, SampleRate), "CSMeter::ProcessData"); }
    TYPEREAL GetPeak() { CSDR_LOCK(); return csdr_smeter_get_peak(m_h); }
    TYPEREAL GetAve() { CSDR_LOCK(); return csdr_smeter_get_ave(m_h); }
private:
    csdr_smeter *m_h;
    std::mutex m_Mutex;
};
#endif  // SMETER_H
