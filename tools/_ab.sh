set -u
out=gpurun_out/$1
mkdir -p $out
python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_kernels.py tests/test_gpu_cl.py tests/test_gpu_bf16_storage.py -x -q > $out/tests.log 2>&1; echo rc=$? >> $out/tests.log; tail -3 $out/tests.log
python3 bench.py --steps 8 --warmup 2 --cpu-baseline off --also off > $out/bench_f16x2.json 2> $out/bench.err
python3 bench.py --precision fp32 --steps 4 --warmup 1 --cpu-baseline off --also off > $out/bench_fp32.json 2>> $out/bench.err
python3 bench.py --precision bf16 --storage bf16 --steps 2 --warmup 1 --cpu-baseline off > $out/bench_cl.json 2>> $out/bench.err
python3 bench.py --precision bf16 --steps 3 --warmup 1 --cpu-baseline off > $out/bench_bf16.json 2>> $out/bench.err
python3 - $out <<'PY'
import json,sys
for m in ('f16x2','fp32','cl','bf16'):
    try:
        d=json.loads(open('%s/bench_%s.json'%(sys.argv[1],m)).read().strip().splitlines()[-1])
        print(m,round(d['value'],4),'v/s', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
    except Exception as e: print('ERR',m,e)
PY
WMH_STAMPS=1 python tools/f16x2_check.py 2>&1 | grep -A2 "8x128^3 32+0\|8x64^3 64+0" 
