set -u
out=gpurun_out/r03e
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_kernels.py tests/test_gpu_network.py -x -q > $out/tests.log 2>&1; echo rc=$? >> $out/tests.log; tail -4 $out/tests.log
for st in 0 1; do
  WMH_SUPERTILE=$st python3 bench.py --steps 8 --warmup 2 --cpu-baseline off --also off > $out/bench_st$st.json 2> $out/bench_st$st.err
  WMH_SUPERTILE=$st python3 bench.py --precision fp32 --steps 4 --warmup 1 --cpu-baseline off --also off > $out/bench_fp32_st$st.json 2>> $out/bench_st$st.err
  for c in FETCH_SIZE WRITE_SIZE; do
    WMH_SUPERTILE=$st WMH_PMC_MODE=f16x2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/st$st/pmc_f16x2_$c -o p -- python3 tools/pmc_traffic.py > $out/pmc_st${st}_$c.log 2>&1
  done
done
find $out -name '*.db' -delete; find $out -name '*_agent_info.csv' -delete
python3 - <<'PY'
import json
for st in (0,1):
    for m in ('', 'fp32_'):
        try:
            d=json.loads(open('gpurun_out/r03e/bench_%sst%d.json'%(m,st)).read().strip().splitlines()[-1])
            print('supertile',st,m or 'f16x2',round(d['value'],4),'v/s', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
        except Exception as e: print('ERR',st,m,e)
PY
du -sh $out
