#!/bin/bash
# CPU-side sanitizer sweep of the PRODUCT's host code (build container, no GPU).
#  1. libcutesdr_mi host halves under AddressSanitizer + UBSan (cutesdr_amd/_san/asan/, device code unchanged,
#     -fno-gpu-sanitize): the CPU test suite's product-facing tests run against it -- filter / decimator / AGC design
#     math (tests/test_host_logic.py), the ABI surface and the no-GPU failure paths of every create() (tests/test_capi_abi.py).
#  2. the sound sink's queue, rate loop and two-thread protocol (capi_soundsink.hip with its device resampler replaced by
#     a host stand-in, -DCSDR_SOUNDSINK_HOST_STUB) driven by tests/cpp/soundsink_threads.cpp under ThreadSanitizer and
#     under AddressSanitizer + UBSan.
# GPU AddressSanitizer / xnack runs are not available on this pool; nothing here touches a GPU.
set -e
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
python3 -m cutesdr_amd._build --sanitize=asan > /dev/null
echo "== pytest against the ASan+UBSan host build"
CSDR_LIB_PATH=$PWD/cutesdr_amd/_san/asan/libcutesdr_mi_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 -m pytest tests/test_host_logic.py tests/test_capi_abi.py -q -x -p no:cacheprovider
mkdir -p cutesdr_amd/_san/drv
for kind in thread address,undefined; do
  out=cutesdr_amd/_san/drv/soundsink_${kind%%,*}
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=$kind -fno-gpu-sanitize -fno-omit-frame-pointer \
    -DCSDR_SOUNDSINK_HOST_STUB cutesdr_amd/csrc/capi_soundsink.hip cutesdr_amd/csrc/capi_core.hip tests/cpp/soundsink_threads.cpp -o $out
  echo "== soundsink two-thread drive under -fsanitize=$kind"
  TSAN_OPTIONS=halt_on_error=1 ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1 $out
done
echo "sanitize_host: clean"
