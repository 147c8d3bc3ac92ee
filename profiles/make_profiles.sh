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
    conv3x3_halo_f32_kernel conv3x3_halo_f32s_kernel wgrad3x3_up2_f32_kernel wgrad3x3_halo_kernel igemm_fwd_kernel igemm_wgrad_kernel gemm_small > /dev/null
python3 profiles/tools/r3_percall.py --precision f32 --top 90 > gpurun_out/r05_f32_percall.md 2>/dev/null
python3 profiles/tools/r5_small_maps.py > gpurun_out/r05_small_maps.txt 2>/dev/null
python3 profiles/tools/r5_edges_probe.py > gpurun_out/r05_helpers.txt 2>/dev/null
python3 profiles/tools/r5_wgrad_probe.py >> gpurun_out/r05_helpers.txt 2>/dev/null
python3 profiles/tools/r3_celeba_epochs.py 25600 4 > gpurun_out/r05_celeba_epochs_f32.txt 2>/dev/null
LADDER_BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 1 --sustained-seconds 0 --no-cpu-baseline --no-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_bench_2ranks_1gpu.json
# hardware counters of the hot fp32 kernels, one counter set per pass (kernel trace + pmc only)
rm -rf gpurun_out/pmc_hot; mkdir -p gpurun_out/pmc_hot
cd /tmp
i=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_hot/p$i -- python3 $R/profiles/tools/r5_pmc_probe.py > $R/gpurun_out/pmc_hot/p$i.log 2>&1
done
cd $R
python3 profiles/tools/r5_pmc_probe.py --show $(find gpurun_out/pmc_hot -name "*.db") > gpurun_out/r05_f32_hot_kernels_pmc.txt 2>&1
tail -c 1500 gpurun_out/r05_f32_bench_default.json
