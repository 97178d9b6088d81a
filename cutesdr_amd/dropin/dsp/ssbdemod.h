// dsp/ssbdemod.h drop-in: class CSsbDemod (reference dsp/ssbdemod.h:13-19).
#ifndef SSBDEMOD_H
#define SSBDEMOD_H
#include "dsp/datatypes.h"
#include "dsp/csdr_dropin.h"

class CSsbDemod
{
public:
    CSsbDemod() {}
    int ProcessData(int InLength, TYPECPX *pInData, TYPEREAL *pOutData) { return csdr_dropin_count(csdr_ssbdemod_process_mono(InLength, &pInData->re, pOutData), "CSsbDemod::ProcessData"); }
    int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData) { return csdr_dropin_count(csdr_ssbdemod_process_stereo(InLength, &pInData->re, &pOutData->re), "CSsbDemod::ProcessData"); }
};
#endif  // SSBDEMOD_H
