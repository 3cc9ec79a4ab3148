#!/bin/bash
# Host side of libcalitas_hip under AddressSanitizer / ThreadSanitizer (CPU build only: never on the GPU box, whose pool has no GPU
# sanitizer).  Builds calitas_amd/libcalitas_hip_{address,thread}.so with `make SAN=...` (only the .cpp objects are instrumented)
# and runs the CPU tests that go through the library on a host-only context: reference packing, window tables, the per-window filter,
# removeOverlaps / rows on the worker pool, the variant-window producer, the gloo multi-process partitions.
#   tools/sanitize.sh address|thread [pytest args]        (summary: profiles/r03_sanitizers.txt is a committed run of both)
set -e
SAN=${1:-address}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/calitas_amd/csrc" SAN=$SAN -j8 >/dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.$([ $SAN = address ] && echo asan || echo tsan)-x86_64.so | head -1)
export CALITAS_LIB_PATH="$ROOT/calitas_amd/libcalitas_hip_$SAN.so"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1      # the interpreter's own allocations are not ours to chase
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 suppressions=$ROOT/tools/tsan.supp"
TESTS=${@:-tests/test_host_logic.py tests/test_variants_host.py tests/test_distributed_gloo.py}
cd "$ROOT"
LD_PRELOAD="$RT" python -m pytest -x -q -m "not gpu" -p no:cacheprovider $TESTS
