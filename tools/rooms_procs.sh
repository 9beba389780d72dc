#!/bin/bash
# usage: tools/rooms_procs.sh M [N=512] [FRAMES=240] [room|open] [mask|nomask] -- M PROCESSES, a room each, on the one GPU; with `mask` every
# process keeps 1/M of the CUs of every XCD to itself (HSA_CU_MASK, applied by the ROCm runtime to every queue of the process).
# The processes render their frames first (seconds), meet at a file barrier, then run; the sum of their frame rates is printed.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rooms_native
mkdir -p $OUT
gcc -O2 -std=c11 -pthread -I$ROOT/include $ROOT/tools/rooms_native.c -L$ROOT/housescan_amd -lhskinfu -ldl -Wl,-rpath,$ROOT/housescan_amd -Wl,-rpath-link,/opt/rocm/lib -o $OUT/rooms_native || exit 1
M=$1; N=${2:-512}; F=${3:-240}; S=${4:-room}; MODE=${5:-mask}
rm -f $OUT/proc_*.json
for r in $(seq 0 $((M-1))); do
  c0=$((r*256/M)); c1=$(((r+1)*256/M-1))
  if [ "$MODE" = mask ]; then export HSA_CU_MASK="0:$c0-$c1"; else unset HSA_CU_MASK; fi
  ROOMS_FIRST_VARIANT=$r ROOMS_START_FILE=$OUT/go $OUT/rooms_native 1 $N $F $S 1 host 0 0 > $OUT/proc_$r.json 2> $OUT/proc_$r.err &
done
sleep ${WARM_S:-25}; touch $OUT/go
wait
rm -f $OUT/go
python3 - <<PY
import json,glob
tot=0
for f in sorted(glob.glob("$OUT/proc_*.json")):
    try:
        j=json.loads(open(f).read().splitlines()[0]); tot+=j["frames_per_s_in_all"]; print("  ", f.split("/")[-1], j["frames_per_s_in_all"], "lost", j["lost_frames"])
    except Exception as e:
        print("  ", f, "failed", e, open(f.replace(".json",".err")).read()[-300:])
print("$M processes ($MODE), a room each: %.1f frames/s in all" % tot)
PY
