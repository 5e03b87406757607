# HBM-side traffic of the batch-1 gate|up projection (608 x 22016 x 4096), warm (one weight, repeated) and cold (8 weights in rotation): is the
# cold launch slower because it re-fetches, or because of latency?  Counter passes on their own (--pmc + --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_cold
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_cold/warm_$C -o pmc --output-format csv -- python3 $R/tools/gemm_one.py 608 22016 4096 > $R/gpurun_out/pmc_cold/warm_$C.log 2>&1 || echo "warm pass $C failed"
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_cold/cold_$C -o pmc --output-format csv -- python3 $R/tools/gemm_one_cold.py 608 22016 4096 > $R/gpurun_out/pmc_cold/cold_$C.log 2>&1 || echo "cold pass $C failed"
done
echo done
