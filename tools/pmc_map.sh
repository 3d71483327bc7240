cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GDF_MAP_DBG=9
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_map_c/$n -o r --output-format csv -- python3 $R/tools/pmc_attn_map.py 8 > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_map_c | grep -A40 "attn_map"
