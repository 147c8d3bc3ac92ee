#!/bin/bash
# Regenerates the round-4 evidence files under gpurun_out/ (copy to profiles/ afterwards).  Run on the GPU box from the repo root.
# The headline leg of bench.py is STRICT fp32 (matmul_precision f32): every file below describes that leg.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r04_f32_bench_default.json 2> $R/gpurun_out/r04_f32_bench_default.err
rm -rf $R/gpurun_out/prof_k $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare > $R/gpurun_out/r04_f32_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_w.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/r04_f32_bench_under_rocprof.log > gpurun_out/r04_f32_bench_under_rocprof.json
python3 profiles/summarize_rocpd.py $(find gpurun_out/prof_k -name "*.db" | head -1) > gpurun_out/r04_f32_bench_kernel_stats.md
python3 profiles/pmc_traffic.py $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) gpurun_out/r04_f32_pmc_traffic.json \
    conv3x3_halo_f32_kernel wgrad3x3_halo_kernel igemm_fwd_kernel igemm_wgrad_kernel gemm_small > /dev/null
python3 profiles/tools/r3_percall.py --top 90 > gpurun_out/r04_f32_percall.md 2>/dev/null
tail -c 1500 gpurun_out/r04_f32_bench_default.json
