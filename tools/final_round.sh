#!/bin/bash
# The round's bench lines and suites on the GPU box, into gpurun_out/final (tools/collect_round.py copies them into profiles/rNN):
#   bench_default_200steps.json, bench_driver_window_20steps.json, bench_sync_api_quick.json, bench_gpus2_share_gpu.json,
#   bench_gpus8_share_gpu.json, gputest_final.txt, fuzz_campaign.txt, chunk_stress.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
F=$ROOT/gpurun_out/final
mkdir -p $F; cd $ROOT
python -m pytest tests -m gpu -q > $F/gputest_final.txt 2>&1; tail -1 $F/gputest_final.txt
python bench.py --steps 20 --warmup 5 > $F/bench_driver_window_20steps.json 2> $F/bench_driver_window.err
python bench.py > $F/bench_default_200steps.json 2> $F/bench_default.err
python bench.py --quick --sync-api --steps 200 --warmup 20 > $F/bench_sync_api_quick.json 2>/dev/null
python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --no-1024 > $F/bench_gpus2_share_gpu.json 2> $F/bench_gpus2.err
python bench.py --gpus 8 --share-gpu --volume 256 --steps 6 --warmup 2 --no-1024 > $F/bench_gpus8_share_gpu.json 2> $F/bench_gpus8.err
[ "$SKIP_FUZZ" ] || python tools/fuzz_campaign.py 1000 200 > $F/fuzz_campaign.txt 2>&1
[ "$SKIP_FUZZ" ] || python tools/chunk_stress.py > $F/chunk_stress.txt 2>&1
tail -1 $F/fuzz_campaign.txt $F/chunk_stress.txt 2>/dev/null
