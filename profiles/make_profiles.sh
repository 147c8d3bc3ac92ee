#!/bin/bash
# Regenerates the round-6 evidence files under gpurun_out/ (copy to profiles/ afterwards).  Run on the GPU box from the repo root.
# The headline leg of bench.py is STRICT fp32 (matmul_precision f32): every file below describes that leg.  (Round-5 version: git history.)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r06_f32_bench_default.json 2> $R/gpurun_out/r06_f32_bench_default.err
rm -rf $R/gpurun_out/prof_k $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare > $R/gpurun_out/r06_f32_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_w.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/r06_f32_bench_under_rocprof.log > gpurun_out/r06_f32_bench_under_rocprof.json
python3 profiles/summarize_rocpd.py $(find gpurun_out/prof_k -name "*.db" | head -1) > gpurun_out/r06_f32_bench_kernel_stats.md
python3 profiles/pmc_traffic.py $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) gpurun_out/r06_f32_pmc_traffic.json \
    up2proj_fused gemm_f32_kernel gemm_nt16_f32_kernel gemm_tn_f32_kernel up2proj_fwd_combine_rows_kernel up2proj_bwd_combine_kernel up2proj_bwd_combine_walk_kernel up2proj_proj_reduce_kernel conv3x3_halo_f32_kernel conv3x3_halo_f32s_kernel wgrad3x3_halo_kernel igemm_fwd_kernel igemm_wgrad_kernel gemm_small > /dev/null
python3 profiles/tools/r3_percall.py --precision f32 --top 90 > gpurun_out/r06_f32_percall.md 2>/dev/null
python3 profiles/tools/r6_fused_probe.py > gpurun_out/r06_fused_probe.txt 2>/dev/null
python3 profiles/tools/r5_upproj_probe.py > gpurun_out/r06_upproj_probe.txt 2>/dev/null
python3 profiles/tools/r6_bwd_walk_probe.py > gpurun_out/r06_bwd_walk_probe.txt 2>/dev/null
python3 profiles/tools/r5_gemm_library_probe.py > gpurun_out/r06_gemm_library_probe.txt 2>/dev/null
# hardware counters: the fused forward kernel; the dense kernels of the projected pairs + the library yardstick (one counter set per pass, kernel trace + pmc only)
bash profiles/tools/r6_fused_pmc.sh
bash profiles/tools/r5_gemm_pmc.sh; mv gpurun_out/r05_gemm_pmc.txt gpurun_out/r06_gemm_pmc.txt
# A/B on this box: the forward pairs as two launches (round 5), fused wherever eligible
for lvl in 0 2; do
python3 bench.py --steps 20 --warmup 5 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --no-compare --set fused_projected_forward=$lvl 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_f32_bench_fused_level$lvl.json
done
# ... and the last pair's backward combination: conv2d_8's backward + the combination (two launches, dy through HBM) against the one launch
python3 bench.py --steps 20 --warmup 5 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --no-compare --set fused_projection_backward=0 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_f32_bench_bwdproj0.json
python3 bench.py --config codes/celeba_r8k50_config.json --steps 20 --warmup 5 --repeats 2 --sustained-seconds 0 --no-cpu-baseline --no-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_bench_r8k50.json
# BASELINE configs[1] / configs[0] (MNIST-fashion, MNIST-digit: hipGraph replay) on the final build -- SURVEY 8(d): us / iteration, launch count, fraction of peak
python3 bench.py --config codes/mnist_fashion_config.json --steps 200 --warmup 20 --repeats 3 --sustained-seconds 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_bench_mnist_fashion.json
python3 bench.py --config codes/mnist_digit_config.json --steps 200 --warmup 20 --repeats 3 --sustained-seconds 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_bench_mnist_digit.json
python3 profiles/tools/r3_celeba_epochs.py 25600 4 > gpurun_out/r06_celeba_epochs_f32.txt 2>/dev/null
LADDER_BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_bench_2ranks_1gpu.json
tail -c 1500 gpurun_out/r06_f32_bench_default.json
