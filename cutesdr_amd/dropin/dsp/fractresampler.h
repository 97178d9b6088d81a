// dsp/fractresampler.h drop-in: class CFractResampler (reference dsp/fractresampler.h:17-33).
#ifndef FRACTRESAMPLER_H
#define FRACTRESAMPLER_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"

class CFractResampler
{
public:
    CFractResampler() : m_h(csdr_dropin_handle(csdr_resampler_create(CSDR_DEVICE), "CFractResampler")) {}
    virtual ~CFractResampler() { csdr_resampler_destroy(m_h); }
    CFractResampler(const CFractResampler &) = delete;
    CFractResampler &operator=(const CFractResampler &) = delete;
    void Init(int MaxInputSize) { CSDR_LOCK(); csdr_dropin_count(csdr_resampler_init(m_h, MaxInputSize), "CFractResampler::Init"); }
    int Resample(int InLength, TYPEREAL Rate, TYPEREAL *pInBuf, TYPEREAL *pOutBuf) { CSDR_LOCK(); return csdr_dropin_count(csdr_resampler_resample_real(m_h, InLength, Rate, pInBuf, pOutBuf), "CFractResampler::Resample"); }
    int Resample(int InLength, TYPEREAL Rate, TYPECPX *pInBuf, TYPECPX *pOutBuf) { CSDR_LOCK(); return csdr_dropin_count(csdr_resampler_resample_cpx(m_h, InLength, Rate, &pInBuf->re, &pOutBuf->re), "CFractResampler::Resample"); }
    int Resample(int InLength, TYPEREAL Rate, TYPEREAL *pInBuf, TYPEMONO16 *pOutBuf, TYPEREAL gain) { CSDR_LOCK(); return csdr_dropin_count(csdr_resampler_resample_real_i16(m_h, InLength, Rate, pInBuf, pOutBuf, gain), "CFractResampler::Resample"); }
    int Resample(int InLength, TYPEREAL Rate, TYPECPX *pInBuf, TYPESTEREO16 *pOutBuf, TYPEREAL gain) { CSDR_LOCK(); return csdr_dropin_count(csdr_resampler_resample_cpx_i16(m_h, InLength, Rate, &pInBuf->re, &pOutBuf->re, gain), "CFractResampler::Resample"); }
private:
    csdr_resampler *m_h;
    std::mutex m_Mutex;
};
#endif  // FRACTRESAMPLER_H
