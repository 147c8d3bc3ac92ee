#!/bin/bash
# Regenerates the round-5 evidence files under gpurun_out/ (copy to profiles/ afterwards).  Run on the GPU box from the repo root.
# The headline leg of bench.py is STRICT fp32 (matmul_precision f32): every file below describes that leg.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r05_f32_bench_default.json 2> $R/gpurun_out/r05_f32_bench_default.err
rm -rf $R/gpurun_out/prof_k $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare > $R/gpurun_out/r05_f32_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_w.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/r05_f32_bench_under_rocprof.log > gpurun_out/r05_f32_bench_under_rocprof.json
python3 profiles/summarize_rocpd.py $(find gpurun_out/prof_k -name "*.db" | head -1) > gpurun_out/r05_f32_bench_kernel_stats.md
python3 profiles/pmc_traffic.py $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) gpurun_out/r05_f32_pmc_traffic.json \
    gemm_f32_kernel gemm_nt16_f32_kernel gemm_tn_f32_kernel up2proj_fwd_combine_rows_kernel up2proj_bwd_combine_kernel conv3x3_halo_f32_kernel conv3x3_halo_f32s_kernel wgrad3x3_halo_kernel igemm_fwd_kernel igemm_wgrad_kernel gemm_small > /dev/null
python3 profiles/tools/r3_percall.py --precision f32 --top 90 > gpurun_out/r05_f32_percall.md 2>/dev/null
python3 profiles/tools/r5_upproj_probe.py > gpurun_out/r05_upproj_probe.txt 2>/dev/null
python3 profiles/tools/r5_gemm_library_probe.py > gpurun_out/r05_gemm_library_probe.txt 2>/dev/null
LADDER_UP2PROJ_SEG=0 python3 profiles/tools/r5_upproj_probe.py 2>/dev/null | grep "fwd combine" > gpurun_out/r05_upproj_one_pixel_combine.txt
if [ "$R5_OLD_FORMS" = "1" ]; then   # the tap-folded forms of the first half of the round (upsample_fused_convs: 3): profiles/r05_small_maps.txt, r05_helpers.txt
python3 profiles/tools/r5_small_maps.py > gpurun_out/r05_small_maps.txt 2>/dev/null
python3 profiles/tools/r5_edges_probe.py > gpurun_out/r05_helpers.txt 2>/dev/null
python3 profiles/tools/r5_wgrad_probe.py >> gpurun_out/r05_helpers.txt 2>/dev/null
fi
python3 profiles/tools/r3_celeba_epochs.py 25600 4 > gpurun_out/r05_celeba_epochs_f32.txt 2>/dev/null
LADDER_BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_bench_2ranks_1gpu.json
# hardware counters of the dense kernels of the projected pairs + the library yardstick (one counter set per pass, kernel trace + pmc only)
bash profiles/tools/r5_gemm_pmc.sh
# the same leg with the tap-folded forms of the first half of the round, for the A/B line of DESIGN 5
python3 bench.py --steps 20 --warmup 5 --repeats 2 --sustained-seconds 0 --no-cpu-baseline --no-compare --set upsample_fused_convs=3 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_f32_bench_level3.json
python3 bench.py --config codes/celeba_r8k50_config.json --steps 20 --warmup 5 --repeats 2 --sustained-seconds 0 --no-cpu-baseline --no-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_bench_r8k50.json
tail -c 1500 gpurun_out/r05_f32_bench_default.json
