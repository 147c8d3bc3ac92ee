# hardware counters of the fused forward kernel (one counter set per pass, kernel trace + pmc only) -> gpurun_out/r06_fused_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fused; mkdir -p $R/gpurun_out/pmc_fused
i=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_fused/p$i -- python3 $R/profiles/tools/r6_fused_pmc_probe.py > $R/gpurun_out/pmc_fused/p$i.log 2>&1
done
cd $R
python3 profiles/tools/r6_fused_pmc_probe.py --show $(find gpurun_out/pmc_fused -name "*.db") > gpurun_out/r06_fused_pmc.txt 2>&1
