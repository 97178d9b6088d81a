// capi_soundsink.hip -- C ABI of the sound-sink adaptation: the queue and the rate-error
// loop of CSoundOut (reference interface/soundout.cpp:155-468, both modes) around the device resampler.
// The step after the path in a live receiver: PutOutQueue resamples the demodulator's audio to the sound-card
// rate with Rate = m_OutRatio (1 + m_RateCorrection) (CFractResampler on the GPU, csdr_resampler_*), the audio
// thread pops with GetOutQueue, and once per second of consumed samples the P-controller CalcError (:456-468)
// turns the average queue fill into the next correction.  The queue, the fill average and the controller are
// scalar host logic; only the resampling is device work.  Blocking mode (Start(..., BlockingMode = true), :86-90):
// PutOutQueue never drops -- it waits, 10 ms at a time, while the queue is full (:209-220, :267-278) -- and GetOutQueue
// returns right after popping, without the fill average and the rate controller (:354-358, :428-432): the producer is
// paced by the sound card instead of being resampled to it.  put and get may come from two threads, as in the
// reference (the IQ thread and the audio thread): both run under the sink's mutex (m_Mutex).
#include "capi_common.hpp"
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#ifdef CSDR_SOUNDSINK_HOST_STUB
// Sanitizer harness only (tools/sanitize_host.sh, tests/cpp/soundsink_threads.cpp): the queue, the rate loop and the
// two-thread protocol of this file run under ThreadSanitizer / AddressSanitizer on a box WITHOUT a GPU, with the
// device resampler replaced by a nearest-sample stand-in.  Never defined in the library the product loads.
namespace {
struct stub_resampler { double t = 0.0; };
template <int W> int stub_resample(stub_resampler *r, int n, double rate, const double *in, short *out, double gain)
{
    int k = 0;
    while ((int)r->t < n) {
        for (int w = 0; w < W; w++) {
            double v = in[W * (int)r->t + w] * gain;
            out[W * k + w] = (short)(v > 32767.0 ? 32767.0 : (v < -32767.0 ? -32767.0 : v));
        }
        k++; r->t += rate;
    }
    r->t -= (double)n;
    return k;
}
}
#define csdr_resampler stub_resampler
#define csdr_resampler_create(dev) (new stub_resampler())
#define csdr_resampler_init(r, n) 0
#define csdr_resampler_destroy(r) delete (r)
#define csdr_resampler_resample_cpx_i16(r, n, rate, in, out, gain) stub_resample<2>(r, n, rate, in, out, gain)
#define csdr_resampler_resample_real_i16(r, n, rate, in, out, gain) stub_resample<1>(r, n, rate, in, out, gain)
#endif

namespace {
constexpr int kQ = 16384;                 // OUTQSIZE (soundout.h:18)
constexpr int kRate = 48000;              // SOUNDCARD_RATE (soundout.cpp:48)
constexpr double kAlpha = 0.001;          // FILTERQLEVEL_ALPHA (:51)
constexpr double kPGain = 2.38e-7;        // P_GAIN (:52)
}

struct csdr_soundsink {
    csdr_resampler *rs = nullptr;
    int stereo = 0;
    bool startup = true;                  // m_Startup
    double user_rate = kRate, out_ratio = 1.0, rate_corr = 0.0, gain = 1.0, ave_level = 0.0;
    int head = 0, tail = 0, level = 0, rate_count = 0, ppm = 0;
    bool blocking = false;                // m_BlockingMode
    std::mutex mu;                        // m_Mutex: queue, fill average, rate controller, the parameters put() reads
    std::mutex mu_put;                    // one producer at a time owns the resampler and its output buffer
    std::condition_variable cv;           // wakes a put that waits for room (the reference sleeps 10 ms and looks again)
    std::vector<short> q, r;              // the ring (2 shorts per entry when stereo), resampler output
};

extern "C" {

/* CSoundOut::CSoundOut (soundout.cpp:60-76) */
csdr_soundsink *csdr_soundsink_create(int device, int stereo)
{
    csdr_resampler *rs = csdr_resampler_create(device);
    if (!rs) return nullptr;
    if (csdr_resampler_init(rs, 8192) < 0) { csdr_resampler_destroy(rs); return nullptr; }
    csdr_soundsink *s = new csdr_soundsink();
    s->rs = rs; s->stereo = stereo != 0;
    s->q.assign((size_t)(s->stereo ? 2 : 1) * kQ, 0);
    s->r.assign((size_t)2 * kQ, 0);
    return s;
}
void csdr_soundsink_destroy(csdr_soundsink *s)
{
    if (!s) return;
    csdr_resampler_destroy(s->rs);
    delete s;
}
/* CSoundOut::Start's BlockingMode argument (:86-90) */
int csdr_soundsink_set_blocking(csdr_soundsink *s, int on)
{
    if (!s) return csdr::fail(CSDR_EINVAL, "bad handle");
    std::lock_guard<std::mutex> lock(s->mu);
    s->blocking = on != 0;
    s->cv.notify_all();
    return CSDR_OK;
}
/* CSoundOut::ChangeUserDataRate (:155-175) */
int csdr_soundsink_change_user_data_rate(csdr_soundsink *s, double rate)
{
    if (!s || !(rate > 0.0)) return csdr::fail(CSDR_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->user_rate != rate) {
        s->user_rate = rate;
        std::fill(s->q.begin(), s->q.end(), (short)0);
        s->out_ratio = rate / (double)kRate;
        s->head = s->tail = s->level = 0;
        s->ave_level = kQ / 2;
        s->startup = true;
    }
    return CSDR_OK;
}
/* CSoundOut::SetVolume (:180-189): 0 mutes, 1..99 = -50 dB .. 0 dB */
int csdr_soundsink_set_volume(csdr_soundsink *s, int vol)
{
    if (!s) return csdr::fail(CSDR_EINVAL, "bad handle");
    std::lock_guard<std::mutex> lock(s->mu);
    if (vol == 0) s->gain = 0.0;
    else if (vol <= 99) s->gain = std::pow(10.0, ((double)vol - 99.0) / 39.2);
    return CSDR_OK;
}
/* CSoundOut::PutOutQueue, non-blocking branch (:196-247 complex -> stereo, :254-305 real -> mono).  in: n doubles
 * (mono sink) or n interleaved double pairs (stereo sink); at most 8192 samples per call, as the resampler was
 * initialised (:71).  Returns the resampled samples produced. */
int csdr_soundsink_put(csdr_soundsink *s, int n, const double *in)
{
    if (!s || n < 0 || (n > 0 && !in)) return csdr::fail(CSDR_EINVAL, "bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> producer(s->mu_put);
    double rate, gain;
    {   // the rate and the gain as they are now: get() and the setters change them under the same mutex
        std::lock_guard<std::mutex> lock(s->mu);
        rate = 1.0 * s->out_ratio * (1.0 + s->rate_corr);          // TEST_ERROR * m_OutRatio * (1 + m_RateCorrection)
        gain = s->gain;
    }
    if ((double)n / rate + 8.0 > (double)kQ) return csdr::fail(CSDR_EINVAL, "call too long for the %d-entry queue", kQ);
    const int k = s->stereo ? csdr_resampler_resample_cpx_i16(s->rs, n, rate, in, s->r.data(), gain)
                            : csdr_resampler_resample_real_i16(s->rs, n, rate, in, s->r.data(), gain);
    if (k < 0) return k;
    std::unique_lock<std::mutex> lock(s->mu);
    int i = 0;
    if (s->blocking) {                                  // :209-220 / :267-278: wait while the queue is full, drop nothing
        for (; i < k; i++) {
            while (s->blocking && ((s->head + 1) & (kQ - 1)) == s->tail)
                s->cv.wait_for(lock, std::chrono::milliseconds(10));
            if (!s->blocking) break;                    // the mode was switched off while waiting: the rest of the call
                                                        // takes the non-blocking branch below, queue-full rule included
            if (s->stereo) { s->q[2 * s->head] = s->r[2 * i]; s->q[2 * s->head + 1] = s->r[2 * i + 1]; }
            else s->q[s->head] = s->r[i];
            s->head = (s->head + 1) & (kQ - 1);
            s->level++;
        }
        if (i == k) return k;
    }
    bool overflow = false;
    for (; i < k; i++) {
        if (s->stereo) { s->q[2 * s->head] = s->r[2 * i]; s->q[2 * s->head + 1] = s->r[2 * i + 1]; }
        else s->q[s->head] = s->r[i];
        s->head = (s->head + 1) & (kQ - 1);
        s->level++;
        if (s->head == s->tail) {                       // full: drop a quarter of the queue (:228-236)
            s->tail = (s->tail + kQ / 4) & (kQ - 1);
            s->level -= kQ / 4;
            overflow = true;
            break;
        }
    }
    if (overflow) s->ave_level = s->level;
    s->ave_level = (1.0 - kAlpha) * s->ave_level + kAlpha * (double)s->level;
    return k;
}
/* CSoundOut::GetOutQueue (:311-375 mono, :381-445 stereo): n samples, or n L/R pairs of a stereo sink */
int csdr_soundsink_get(csdr_soundsink *s, int n, short *out)
{
    if (!s || n < 0 || (n > 0 && !out)) return csdr::fail(CSDR_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(s->mu);
    const int w = s->stereo ? 2 : 1;
    if (s->startup) {                                   // silence until the queue is half full (:316-333)
        std::memset(out, 0, sizeof(short) * (size_t)w * n);
        if (s->level > kQ / 2) {
            s->startup = false;
            s->rate_count = -5 * kRate;                 // first update delayed to let the level settle
            s->ppm = 0;
            s->ave_level = s->level;
        } else return n;
    }
    bool underflow = false;
    for (int i = 0; i < n; i++) {
        if (s->head != s->tail) {
            if (s->stereo) { out[2 * i] = s->q[2 * s->tail]; out[2 * i + 1] = s->q[2 * s->tail + 1]; }
            else out[i] = s->q[s->tail];
            s->tail = (s->tail + 1) & (kQ - 1);
            s->level--;
        } else {                                        // empty: back up a quarter and repeat older data (:344-351)
            s->tail = (s->tail - kQ / 4) & (kQ - 1);
            if (s->stereo) { out[2 * i] = s->q[2 * s->tail]; out[2 * i + 1] = s->q[2 * s->tail + 1]; }
            else out[i] = s->q[s->tail];
            s->level += kQ / 4;
            underflow = true;
        }
    }
    s->cv.notify_all();                                 // room for a waiting put
    if (s->blocking) return n;                          // :354-358 / :428-432: no fill average, no rate controller
    s->ave_level = (1.0 - kAlpha) * s->ave_level + kAlpha * s->level;
    if (underflow) s->ave_level = s->level;
    s->rate_count += n;
    if (s->rate_count >= kRate) {                       // CalcError (:456-468), every second of consumed samples
        s->rate_corr = (double)(s->ave_level - kQ / 2) * kPGain;
        s->ppm = (int)(s->rate_corr * 1e6);
        s->rate_count = 0;
    }
    return n;
}
double csdr_soundsink_get_rate_correction(csdr_soundsink *s)
{
    if (!s) return 0.0;
    std::lock_guard<std::mutex> lock(s->mu);
    return s->rate_corr;
}
double csdr_soundsink_get_ave_level(csdr_soundsink *s)
{
    if (!s) return 0.0;
    std::lock_guard<std::mutex> lock(s->mu);
    return s->ave_level;
}
int csdr_soundsink_get_level(csdr_soundsink *s)
{
    if (!s) return csdr::fail(CSDR_EINVAL, "bad handle");
    std::lock_guard<std::mutex> lock(s->mu);
    return s->level;
}
int csdr_soundsink_get_ppm_error(csdr_soundsink *s)
{
    if (!s) return 0;
    std::lock_guard<std::mutex> lock(s->mu);
    return s->ppm;
}

}  // extern "C"
