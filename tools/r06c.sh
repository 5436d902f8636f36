mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
D=deepwmh_amd/lib/libdeepwmh_amd_diag.so
python tools/seghead_bench.py > $O/seghead_product.txt 2>&1
for b in 1 2 4 8; do echo "BPC $b" >> $O/seghead_bpc.txt; WMH_DIAG_LIB=1 WMH_HEAD_BPC=$b python tools/seghead_bench.py >> $O/seghead_bpc.txt 2>&1; done
python tools/in_bwd_bench.py > $O/in_bwd_new.txt 2>&1
WMH_LIB_PATH=$D WMH_IN_BWD_OLD=1 python tools/in_bwd_bench.py > $O/in_bwd_old.txt 2>&1
python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_trainer.py tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_gpu_f16x2.py -m gpu -q > $O/tests.txt 2>&1
WMH_ITERS=8 WMH_PRECISION=f16x2 python tools/train_bench.py > $O/train_new.txt 2>&1
WMH_LIB_PATH=$D WMH_IN_BWD_OLD=1 WMH_ITERS=8 WMH_PRECISION=f16x2 python tools/train_bench.py > $O/train_old.txt 2>&1
python bench.py --also off --cpu-baseline off --steps 10 --warmup 2 > $O/bench_quick.json 2> $O/bench_quick.err
grep -h TB $O/seghead_product.txt $O/seghead_bpc.txt; tail -1 $O/in_bwd_new.txt $O/in_bwd_old.txt; tail -3 $O/tests.txt; tail -2 $O/train_new.txt $O/train_old.txt; python -c "import json;d=json.load(open('$O/bench_quick.json'));print(d['value'],d['ms_per_step'])"
