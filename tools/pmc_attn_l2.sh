# L2 / fabric counters for the attention kernels (tools/attn_one.py); summary: tools/pmc_attn_parse.py gpurun_out/pmc_attn_l2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-bwd}
mkdir -p $R/gpurun_out/pmc_attn_l2
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_attn_l2/p$i -o pmc --output-format csv -- python3 $R/tools/attn_one.py $MODE > $R/gpurun_out/pmc_attn_l2/log$i.txt 2>&1 || echo "pass $i failed"
done
echo done
