// dsp/fft.h drop-in: class CFft (reference dsp/fft.h:24-85).
#ifndef FFT_H
#define FFT_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"
#ifdef CSDR_DROPIN_QT
#include <QMutex>
#endif

#define MAX_FFT_SIZE 65536
#define MIN_FFT_SIZE 512

class CFft
{
public:
    CFft() : m_h(csdr_dropin_handle(csdr_fft_create(CSDR_DEVICE), "CFft")) {}
    virtual ~CFft() { csdr_fft_destroy(m_h); }
    CFft(const CFft &) = delete;
    CFft &operator=(const CFft &) = delete;

    void SetFFTParams(qint32 size, bool invert, double dBCompensation, double SampleFreq)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        csdr_dropin_count(csdr_fft_set_params(m_h, size, invert, dBCompensation, SampleFreq), "CFft::SetFFTParams");
    }
    void SetFFTAve(qint32 ave) { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_fft_set_ave(m_h, ave), "CFft::SetFFTAve"); }
    void ResetFFT() { std::lock_guard<std::mutex> g(m_Mutex); csdr_dropin_count(csdr_fft_reset(m_h), "CFft::ResetFFT"); }
    bool GetScreenIntegerFFTData(qint32 MaxHeight, qint32 MaxWidth, double MaxdB, double MindB,
                                 qint32 StartFreq, qint32 StopFreq, qint32 *OutBuf)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        return csdr_dropin_count(csdr_fft_get_screen(m_h, MaxHeight, MaxWidth, MaxdB, MindB, StartFreq, StopFreq,
                                                     reinterpret_cast<int *>(OutBuf)), "CFft::GetScreenIntegerFFTData") != 0;
    }
    qint32 PutInDisplayFFT(qint32 n, TYPECPX *InBuf)
    {
        std::lock_guard<std::mutex> g(m_Mutex);
        return csdr_dropin_count(csdr_fft_put_display(m_h, n, &InBuf->re), "CFft::PutInDisplayFFT");
    }
    void FwdFFT(TYPECPX *pInOutBuf) { CSDR_LOCK(); csdr_dropin_count(csdr_fft_fwd(m_h, &pInOutBuf->re), "CFft::FwdFFT"); }
    void RevFFT(TYPECPX *pInOutBuf) { CSDR_LOCK(); csdr_dropin_count(csdr_fft_rev(m_h, &pInOutBuf->re), "CFft::RevFFT"); }

private:
    csdr_fft *m_h;
    std::mutex m_Mutex;
};
#endif  // FFT_H
