#!/bin/bash
# One GPU-box pass that produces everything profiles/ cites for a round: the bench lines, kernel-trace stats of the bench modes, the
# PMC traffic passes (separate from the stats passes), training stats.  Usage (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --precision fp32 --also off > $out/bench_fp32.json 2> $out/bench_fp32.err
python3 bench.py --precision bf16 --storage bf16 > $out/bench_bf16_storage.json 2> $out/bench_bf16_storage.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -o s -- python3 bench.py --cpu-baseline off --also off > $out/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_fp32 -o s -- python3 bench.py --precision fp32 --cpu-baseline off --also off > $out/stats_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_bf16_storage -o s -- python3 bench.py --precision bf16 --storage bf16 --cpu-baseline off > $out/stats_bf16_storage.log 2>&1
for p in fp32 f16x2; do
  WMH_ITERS=3 WMH_PRECISION=$p rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train_$p -o s -- python3 tools/train_bench.py > $out/stats_train_$p.log 2>&1
done
for mode in f16x2 fp32 bf16_storage; do
  for c in FETCH_SIZE WRITE_SIZE; do
    # batch = what bench.py's predictor puts into one forward (16 patches: two tiles' mirrored copies)
    b=16
    WMH_BATCH=$b WMH_PMC_MODE=$mode rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${mode}_$c -o p -- python3 tools/pmc_traffic.py > $out/pmc_${mode}_$c.log 2>&1
  done
done
# the pair weight-gradient kernel: traffic with its z-first tile order and with the plain order, per-layer times and stamps
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_wgrad_$c -o p -- python3 tools/pmc_wgrad.py > $out/pmc_wgrad_$c.log 2>&1
done
WMH_WGRAD_PLAIN_ORDER=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_wgrad_plain_order_FETCH_SIZE -o p -- python3 tools/pmc_wgrad.py > $out/pmc_wgrad_plain_order.log 2>&1
WMH_STAMPS=1 python3 tools/wgrad_check.py > $out/wgrad_check_per_layer.txt 2>&1
find $out -name '*.db' -delete; find $out -name '*_agent_info.csv' -delete
# keep the CSVs small enough to travel: drop the per-dispatch kernel traces of the stats runs (the stats tables stay)
find $out -name '*kernel_trace.csv' -path '*stats_*' -delete
du -sh $out
