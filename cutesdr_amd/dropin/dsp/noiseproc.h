// dsp/noiseproc.h drop-in: class CNoiseProc (reference dsp/noiseproc.h:23-58).
#ifndef NOISEPROC_H
#define NOISEPROC_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

typedef struct _snproc
{
    bool NBOn;
    int NBThreshold;
    int NBWidth;
} tNoiseProcdInfo;

class CNoiseProc
{
public:
    CNoiseProc() : m_h(csdr_dropin_handle(csdr_noiseproc_create(CSDR_DEVICE), "CNoiseProc")) {}
    virtual ~CNoiseProc() { csdr_noiseproc_destroy(m_h); }
    CNoiseProc(const CNoiseProc &) = delete;
    CNoiseProc &operator=(const CNoiseProc &) = delete;
    void SetupBlanker(bool On, TYPEREAL Threshold, TYPEREAL Width, TYPEREAL SampleRate)
    { CSDR_LOCK(); csdr_dropin_count(csdr_noiseproc_setup(m_h, On, Threshold, Width, SampleRate), "CNoiseProc::SetupBlanker"); }
    void ProcessBlanker(int InLength, TYPECPX *pInData, TYPECPX *pOutData)
    { CSDR_LOCK(); csdr_dropin_count(csdr_noiseproc_process(m_h, InLength, &pInData->re, &pOutData->re), "CNoiseProc::ProcessBlanker"); }
private:
    csdr_noiseproc *m_h;
    std::mutex m_Mutex;       // the reference locks m_Mutex in SetupBlanker and ProcessBlanker (noiseproc.cpp:80,123)
};
#endif  // NOISEPROC_H
