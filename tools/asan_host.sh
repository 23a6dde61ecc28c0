#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of the library (api.cpp, ordering.cpp, symbolic.cpp; the device objects are
# linked as they are) and over the oracle, driven by the CPU test suite.  GPU sanitizers are not available on this pool;
# this covers the C++ that runs on the host: ordering, symbolic analysis, argument checking, the C ABI.
#   bash tools/asan_host.sh        (from the repo root, in the build container)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/csparse3_amd/csrc
OUT=$SRC/build/asan
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
mkdir -p $OUT
make -C $SRC > /dev/null
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in api ordering symbolic; do
  /opt/rocm/bin/hipcc -std=c++17 $SAN -fno-gpu-sanitize -c $SRC/$f.cpp -o $OUT/$f.o
done
/opt/rocm/bin/hipcc -shared $SAN -fno-gpu-sanitize --offload-arch=gfx950 -o $OUT/libcsparse3_hip_asan.so \
    $OUT/api.o $OUT/ordering.o $OUT/symbolic.o $SRC/build/kernels.o $SRC/build/substrate.o
$CLANG -std=c99 $SAN -shared -o $OUT/liboracle_asan.so $ROOT/oracle/cs_oracle.c -lm
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  CS3_LIB_PATH=$OUT/libcsparse3_hip_asan.so ORC_LIB_PATH=$OUT/liboracle_asan.so CS3_SYSTEM_HIP=1 \
  python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tee $OUT/asan_pytest.log | tail -15
if grep -q "ERROR: AddressSanitizer\|runtime error:" $OUT/asan_pytest.log; then echo "SANITIZER FINDINGS"; exit 1; else echo "sanitizers: clean"; fi
