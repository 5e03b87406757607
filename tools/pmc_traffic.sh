# HBM-side traffic of the dominant GEMM shape (gate|up forward, 4864 x 22016 x 4096): two separate counter passes, as
# MI355X_MICROARCH.md "HBM" prescribes (FETCH_SIZE x2 on gfx950 for wide coalesced reads; WRITE_SIZE as is; both in KiB).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc3/$C -o pmc --output-format csv -- python3 $R/tools/gemm_one.py 4864 22016 4096 > $R/gpurun_out/pmc3/$C.log 2>&1 || echo "pass $C failed"
done
echo done
