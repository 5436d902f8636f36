#!/bin/bash
# Same-box A/B of two builds of the library (product + diagnostic pairs): interleaved rounds of tools/f16x2_check.py (per-layer times and,
# with the diagnostic libraries, the kernels' own cycle stamps), then bench.py's headline mode on each.  Usage (on the GPU box):
#   bash tools/ab_libs.sh <tag> <A product .so> <A diag .so> <B product .so> <B diag .so>
tag=$1; A=$2; AD=$3; B=$4; BD=$5
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  WMH_LIB_PATH=$A python3 tools/f16x2_check.py > $out/check_A_$r.txt 2>&1
  WMH_LIB_PATH=$B python3 tools/f16x2_check.py > $out/check_B_$r.txt 2>&1
done
WMH_STAMPS=1 WMH_DIAG_LIB_PATH=$AD python3 tools/f16x2_check.py > $out/stamps_A.txt 2>&1
WMH_STAMPS=1 WMH_DIAG_LIB_PATH=$BD python3 tools/f16x2_check.py > $out/stamps_B.txt 2>&1
for r in 1 2; do
  WMH_LIB_PATH=$A WMH_DIAG_LIB_PATH=$AD python3 bench.py --also off --cpu-baseline off --steps 4 --detail-file none > $out/bench_A_$r.json 2> $out/bench_A_$r.err
  WMH_LIB_PATH=$B WMH_DIAG_LIB_PATH=$BD python3 bench.py --also off --cpu-baseline off --steps 4 --detail-file none > $out/bench_B_$r.json 2> $out/bench_B_$r.err
done
grep -h "8x128\|8x64\|8x32\|stride 2" $out/check_A_2.txt | sed 's/^/A /'; grep -h "8x128\|8x64\|8x32\|stride 2" $out/check_B_2.txt | sed 's/^/B /'
for f in $out/bench_*.json; do echo $f; python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('in_kernel'))"; done
