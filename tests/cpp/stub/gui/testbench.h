// tests/cpp/stub/gui/testbench.h -- TEST STUB (own code) with the shape of the host's gui/testbench.h as far as the
// drop-in CDemodulator touches it (reference gui/testbench.h:29-38 PROFILE_*, the two DisplayData overloads the chain
// calls at dsp/demodulator.cpp:175,180,187,208, the global g_pTestBench): records what it is handed.
#ifndef TESTBENCH_H
#define TESTBENCH_H
#include "dsp/datatypes.h"
#include <vector>
#define PROFILE_OFF 0
#define PROFILE_1 1
#define PROFILE_2 2
#define PROFILE_3 3
#define PROFILE_4 4
class CTestBench
{
public:
    struct Call { int profile, n; bool cpx; double rate, first; };
    std::vector<Call> calls;
    void DisplayData(int n, TYPECPX *pBuf, double samplerate, int profile)
    { calls.push_back(Call{profile, n, true, samplerate, n ? pBuf[0].re : 0.0}); }
    void DisplayData(int n, TYPEREAL *pBuf, double samplerate, int profile)
    { calls.push_back(Call{profile, n, false, samplerate, n ? pBuf[0] : 0.0}); }
};
extern CTestBench *g_pTestBench;
#endif
