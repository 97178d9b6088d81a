// Compile-only check of the drop-in boundary as the reference HOST sees it: exactly the three dsp/ headers
// that interface/sdrinterface.h includes (:13, :15, :16), and the by-value members it declares from them
// (:173-178) -- among them `CIir m_Iir`, whose header the host never names: it arrives through
// dsp/demodulator.h -> dsp/fmdemod.h -> dsp/iir.h, as in the reference (dsp/demodulator.h:11-18,
// dsp/fmdemod.h:10-12).  interface/soundout.h:16 adds dsp/fractresampler.h, gui/testbench.h:19-20
// dsp/datatypes.h + dsp/fft.h.
#include "dsp/fft.h"
#include "dsp/demodulator.h"
#include "dsp/noiseproc.h"

class CSdrInterfaceMembers        // interface/sdrinterface.h:173-178
{
public:
    CFft m_Fft;
    CDemodulator m_Demodulator;
    CNoiseProc m_NoiseProc;
    CIir m_Iir;
};

// names the reference's demodulator.h makes visible to its includers
static CDownConvert *p1; static CFastFIR *p2; static CSMeter *p3; static CAgc *p4; static CAmDemod *p5;
static CSamDemod *p6; static CFmDemod *p7; static CSsbDemod *p8; static CFir *p9;

int host_includes_ok()
{
    (void)p1; (void)p2; (void)p3; (void)p4; (void)p5; (void)p6; (void)p7; (void)p8; (void)p9;
    tDemodInfo info;
    info.HiCut = 0; info.txt = "x";
    qint16 a = 1; qint32 b = 2; TYPECPX c = {0.0, 0.0}; TYPEREAL r = K_2PI;
    return (int)(a + b + c.re + r) + (int)sizeof(CSdrInterfaceMembers) + MAX_FFT_SIZE + MAX_INBUFSIZE + info.HiCut;
}
