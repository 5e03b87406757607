# PMC passes over the Llama-shape attention kernels (tools/attn_one.py); summaries: tools/pmc_attn_parse.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwd}
mkdir -p $R/gpurun_out/pmc_attn
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_attn/p$i -o pmc --output-format csv -- python3 $R/tools/attn_one.py $MODE > $R/gpurun_out/pmc_attn/log$i.txt 2>&1 || echo "pass $i failed"
done
echo done
