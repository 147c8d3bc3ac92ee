#!/bin/bash
# Regenerates the round-3 evidence files under gpurun_out/ (copy to profiles/ afterwards).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r03_bench_default.json 2> $R/gpurun_out/r03_bench_default.err
rm -rf $R/gpurun_out/prof_k $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare > $R/gpurun_out/r03_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-profile --no-compare > $R/gpurun_out/pmc_w.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/r03_bench_under_rocprof.log > gpurun_out/r03_bench_under_rocprof.json
python3 profiles/summarize_rocpd.py $(find gpurun_out/prof_k -name "*.db" | head -1) > gpurun_out/r03_bench_kernel_stats.md
python3 profiles/pmc_traffic.py $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) gpurun_out/r03_pmc_traffic.json > /dev/null
tail -c 1500 gpurun_out/r03_bench_default.json
