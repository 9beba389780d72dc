#!/bin/bash
# HOST-side AddressSanitizer + UndefinedBehaviorSanitizer pass (this container, no GPU; GPU ASan is not available on the pool):
# a scratch copy of libhskinfu.so with its host code instrumented (device code as shipped: -fno-gpu-sanitize) and the C oracle
# instrumented by the same compiler, then the whole CPU suite (-m "not gpu") against both through tools/san_plugin.py.
# (One test is left out: it links a gcc-built C program against the library, which an instrumented library cannot satisfy.)
# Nothing under housescan_amd/ or oracle/ is touched: objects and libraries go to gpurun_out/san (scratch).
#   usage: tools/sanitize_cpu.sh [pytest args]      result: gpurun_out/san/report.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/san
CL=/opt/rocm/lib/llvm/bin
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p $S
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -fno-omit-frame-pointer -g"
FL="-O1 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-bitwise-instead-of-logical -mllvm -amdgpu-kernarg-preload-count=16 $SAN -fno-gpu-sanitize"
cd $ROOT/housescan_amd/csrc
python3 build_id.py --header $S/build/build_id.h 2>/dev/null || { mkdir -p $S/build; python3 build_id.py --header $S/build/build_id.h; }
OBJ=
for f in integrate.hip raycast.hip exchange.hip extract.hip kernels_image.hip kernels_selftest.hip hskinfu_api.hip hskinfu_group.hip synth.cpp products.cpp house.cpp; do
  /opt/rocm/bin/hipcc $FL -I$S -I. -x hip -c $f -o $S/$f.o || exit 1
  OBJ="$OBJ $S/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o $S/libhskinfu.so $OBJ -ldl || exit 1
OF="-O1 -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math $SAN"
$CL/clang $OF -shared -o $S/libkinfu_oracle.so $ROOT/oracle/kinfu_oracle.c -lm || exit 1
$CL/clang $OF -fopenmp -shared -o $S/libkinfu_oracle_omp.so $ROOT/oracle/kinfu_oracle.c -lm || exit 1
cd $ROOT
# leaks: python itself never frees everything; the allocator's own reports (overflow, use after free, UB) are what is looked for
HSK_SAN_DIR=$S LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest tests -q -m "not gpu" -p tools.san_plugin --deselect tests/test_abi.py::test_native_rooms_harness_builds_against_the_header "$@" > $S/report.txt 2>&1
echo "exit $?" >> $S/report.txt
grep -c "ERROR: AddressSanitizer\|runtime error:" $S/report.txt
tail -5 $S/report.txt
