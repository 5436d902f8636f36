# Same-box A/B of the pair weight gradient: an earlier build of the library (ab_old/libA.so + ab_old/libA_diag.so, copied from
# deepwmh_amd/lib/ before rebuilding; ab_old/ is git-ignored) against the current one.  Usage (on the GPU box): bash tools/ab_wgrad.sh
mkdir -p gpurun_out/r04m
python -m pytest tests/test_gpu_train_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -4
for r in 1 2; do
 WMH_LIB_PATH=ab_old/libA.so python3 tools/wgrad_check.py > gpurun_out/r04m/wg_A_$r.txt 2>&1
 python3 tools/wgrad_check.py > gpurun_out/r04m/wg_B_$r.txt 2>&1
done
WMH_STAMPS=1 WMH_DIAG_LIB_PATH=ab_old/libA_diag.so python3 tools/wgrad_check.py > gpurun_out/r04m/wg_A_st.txt 2>&1
WMH_STAMPS=1 python3 tools/wgrad_check.py > gpurun_out/r04m/wg_B_st.txt 2>&1
for r in 1 2; do
 WMH_LIB_PATH=ab_old/libA.so WMH_PRECISION=f16x2 WMH_ITERS=8 python3 tools/train_bench.py 2>&1 | tail -3 > gpurun_out/r04m/train_A_$r.txt
 WMH_PRECISION=f16x2 WMH_ITERS=8 python3 tools/train_bench.py 2>&1 | tail -3 > gpurun_out/r04m/train_B_$r.txt
done
tail -n 3 gpurun_out/r04m/wg_A_2.txt gpurun_out/r04m/wg_B_2.txt; grep -h "32+0->32\|ticks" gpurun_out/r04m/wg_A_st.txt | sed -n 3,4p; grep -h "32+0->32\|ticks" gpurun_out/r04m/wg_B_st.txt | sed -n 3,4p; cat gpurun_out/r04m/train_*.txt
