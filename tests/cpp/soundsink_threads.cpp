// soundsink_threads.cpp -- two-thread drive of csdr_soundsink's queue for the CPU sanitizer harness
// (tools/sanitize_host.sh builds it with capi_soundsink.hip -DCSDR_SOUNDSINK_HOST_STUB under ThreadSanitizer and under
// AddressSanitizer + UBSan; no GPU involved).  The roles of the reference: the IQ thread calls PutOutQueue, the audio
// thread GetOutQueue (interface/soundout.cpp:196-445); a third thread flips the blocking mode while a put waits on a
// full queue (the case ADVICE r3 found: the put must neither write into a full queue nor lose the level count).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../include/cutesdr_mi.h"

int main()
{
    for (int stereo = 0; stereo < 2; stereo++) {
        csdr_soundsink *s = csdr_soundsink_create(0, stereo);
        if (!s) { std::printf("create failed: %s\n", csdr_last_error()); return 2; }
        csdr_soundsink_change_user_data_rate(s, 48000.0);
        csdr_soundsink_set_volume(s, 99);
        csdr_soundsink_set_blocking(s, 1);
        std::atomic<bool> done{false};
        std::atomic<long> put{0}, got{0};
        std::thread producer([&] {
            std::vector<double> x(2 * 1024);
            for (int k = 0; k < 400; k++) {
                for (size_t i = 0; i < x.size(); i++) x[i] = 1000.0 * ((k + (int)i) % 17 - 8);
                const int r = csdr_soundsink_put(s, 1024, x.data());
                if (r < 0) { std::printf("put failed: %s\n", csdr_last_error()); break; }
                put += r;
            }
            done = true;
        });
        std::thread consumer([&] {
            std::vector<short> y(2 * 512);
            while (!done || csdr_soundsink_get_level(s) > 600) {
                if (csdr_soundsink_get(s, 512, y.data()) != 512) break;
                got += 512;
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        });
        std::thread flipper([&] {
            for (int k = 0; k < 40 && !done; k++) {
                std::this_thread::sleep_for(std::chrono::milliseconds(3));
                csdr_soundsink_set_blocking(s, k & 1);
                (void)csdr_soundsink_get_rate_correction(s); (void)csdr_soundsink_get_ave_level(s); (void)csdr_soundsink_get_ppm_error(s);
                if (k == 20) csdr_soundsink_set_volume(s, 50);
            }
            csdr_soundsink_set_blocking(s, 1);
        });
        producer.join(); flipper.join(); consumer.join();
        const int level = csdr_soundsink_get_level(s);
        std::printf("stereo %d: put %ld, got %ld, level %d\n", stereo, put.load(), got.load(), level);
        if (level < 0 || level > 16384) { std::printf("level out of range\n"); return 1; }
        csdr_soundsink_destroy(s);
    }
    std::printf("soundsink threads ok\n");
    return 0;
}
