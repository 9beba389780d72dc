#!/bin/bash
# usage: tools/rooms_native.sh [N=512] [FRAMES=240] [room|open] [ahead=1] [dev|host] [use_graph=0] -- builds tools/rooms_native.c and runs 1, 2, 4 and 8 rooms;
# with TRACE=M also a rocprofv3 --kernel-trace of the M-room run and its overlap summary (tools/rooms_overlap.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rooms_native
mkdir -p $OUT
gcc -O2 -std=c11 -pthread -I$ROOT/include $ROOT/tools/rooms_native.c -L$ROOT/housescan_amd -lhskinfu -ldl -Wl,-rpath,$ROOT/housescan_amd -Wl,-rpath-link,/opt/rocm/lib -o $OUT/rooms_native || exit 1
for M in ${ROOMS:-1 2 4 8}; do
  $OUT/rooms_native $M "${1:-512}" "${2:-240}" "${3:-room}" "${4:-1}" "${5:-host}" "${6:-0}" 2>&1 | grep -v amdgpu.ids
done
if [ -n "$TRACE" ]; then
  rm -rf $OUT/trace; cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $OUT/rooms_native $TRACE "${1:-512}" "${2:-240}" "${3:-room}" "${4:-1}" "${5:-host}" "${6:-0}" 2>&1 | grep -v amdgpu.ids | tail -1
  python3 $ROOT/tools/rooms_overlap.py $OUT/trace
fi
