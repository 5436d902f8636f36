#!/bin/bash
# One GPU-box pass that produces everything profiles/ cites for a round: kernel-trace stats of the two bench modes, the PMC traffic
# passes, and the bench lines.  Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --precision bf16 --storage bf16 > $out/bench_bf16_storage.json 2> $out/bench_bf16_storage.err
python3 bench.py --precision bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -o s -- python3 bench.py --cpu-baseline off > $out/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_bf16_storage -o s -- python3 bench.py --precision bf16 --storage bf16 --cpu-baseline off > $out/stats_bf16_storage.log 2>&1
for mode in fp32 bf16 bf16_storage f16x2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    WMH_PMC_MODE=$mode rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${mode}_$c -o p -- python3 tools/pmc_traffic.py > $out/pmc_${mode}_$c.log 2>&1
  done
done
find $out -name '*.db' -delete; find $out -name '*_agent_info.csv' -delete
# keep the CSVs small enough to travel: drop the per-dispatch kernel traces of the stats runs (the stats tables stay)
find $out -name '*kernel_trace.csv' -path '*stats_*' -delete
du -sh $out
