#!/bin/bash
# Compiles the kernel sources + C ABI for the SIMT emulator with AddressSanitizer, then with UBSan, and runs the parity
# checks of every entry point against each build: out-of-bounds accesses of "device" memory (malloc'd in the emulator),
# of LDS arrays (thread-local statics) and undefined behaviour in kernels or host code abort the run.  CPU only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tests/_build
mkdir -p "$OUT"
cd "$ROOT/orb_slam3_rgbl_amd/csrc"
for san in address undefined; do
  lib=$OUT/librgbl_frontend_emu_$san.so
  g++ -O1 -g -std=c++17 -fPIC -pthread -ffp-contract=off -DRGBL_EMU -fsanitize=$san -fno-sanitize-recover=all -fno-omit-frame-pointer \
      -I../../tests/emu -Wno-unknown-pragmas -shared -o "$lib" -x c++ extractor.hip -x c++ depth.hip -x c++ matcher.hip -x c++ records.hip -x c++ gather.hip -x c++ ../../tests/emu/hip_emu.cpp -x c++ ../../tests/emu/nccl_emu.cpp
  rt=$(gcc -print-file-name=$([ $san = address ] && echo libasan.so || echo libubsan.so))
  for part in extract depth match misc gather; do
    LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 RGBL_SANITIZED_LIB=$lib python "$ROOT/tools/sanitizer_checks.py" $part 2>&1 | tail -1
  done
done
# ThreadSanitizer: the reference's concurrent callers (two extractor threads, three matcher threads) on the emulator built with
# -fsanitize=thread (tests/test_shim_threads.py builds it; the emulator's fibers are announced to TSan)
cd "$ROOT" && RGBL_TSAN=1 python -m pytest tests/test_shim_threads.py -q -k thread_sanitizer 2>&1 | tail -1
