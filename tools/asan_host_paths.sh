#!/bin/bash
# AddressSanitizer (or, with SAN=undefined, UndefinedBehaviorSanitizer) over the HOST code of libnttx (GPU code is
# not instrumented): builds the library and tools/asan_host_paths.cpp with -fsanitize=$SAN on the host side only,
# into tools/bin/.
# Build here:       [SAN=undefined] tools/asan_host_paths.sh build
# Run on a GPU box: [SAN=undefined] tools/asan_host_paths.sh run
# (ThreadSanitizer builds the same way but its runtime does not start on the GPU boxes: it exits at once, silently.)
set -e
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN=${SAN:-address}
case "$1" in
build)
  make -C ntt_amd/csrc variant NAME=$SAN OUT=/dev/null DEFS="-Xarch_host -fsanitize=$SAN -g" -j8 > /dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=$SAN build/obj_$SAN/*.o -o tools/bin/libnttx_$SAN.so
  $CLANG -O1 -g -std=c++17 -fsanitize=$SAN tools/asan_host_paths.cpp -o tools/bin/host_paths_$SAN -Ltools/bin -lnttx_$SAN -lpthread -Wl,-rpath,'$ORIGIN'
  $CLANG -O1 -g -std=c++17 -fsanitize=$SAN tests/cpp/test_dft.cpp -o tools/bin/test_dft_$SAN -Ltools/bin -lnttx_$SAN -Wl,-rpath,'$ORIGIN'
  ;;
run)
  # the HIP runtime maps memory where the sanitizer expects its shadow gap; its own start-up allocations are not ours
  export ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:abort_on_error=0
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
  # Verdict from the output, not the exit status: at process exit the HSA runtime's own teardown can trip an internal
  # check of the sanitizer's device allocator (frames in libhsa-runtime64 under __cxa_finalize), after our code is done.
  out=$(mktemp)
  timeout -k 10 300 tools/bin/host_paths_$SAN > $out 2>&1 || true; cat $out
  grep -q "host paths clean" $out || exit 1
  ! grep -q "ERROR: AddressSanitizer\|runtime error:" $out || exit 1
  timeout -k 10 300 tools/bin/test_dft_$SAN > $out 2>&1 || true; cat $out
  grep -q "all host-mirror tests passed" $out || exit 1
  ! grep -q "ERROR: AddressSanitizer\|runtime error:" $out || exit 1
  echo "sanitizer run ($SAN): clean"
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
