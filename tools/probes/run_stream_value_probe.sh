#!/bin/bash
# GPU box: build and run the probe (local, then the two-process form on device 0)
cd ${GRAFT_REPO_ROOT:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/svp tools/probes/stream_value_probe.hip -lrt 2>&1 | grep -E "error" 
timeout 60 /tmp/svp local
name=/hsk_probe_$$
timeout 60 /tmp/svp ipc A $name > /tmp/svp_a.log 2>&1 &
pa=$!
sleep 1
timeout 60 /tmp/svp ipc B $name
wait $pa; echo "A exit $?"; cat /tmp/svp_a.log
