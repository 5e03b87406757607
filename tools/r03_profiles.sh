# Round-3 profile collection (run through gpurun from the repo root): rocprofv3 kernel stats of the bench step and of the batch-1 chunk, then the
# two PMC passes of the dominant GEMM shape.  Counters are collected in their own runs (--pmc with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03prof/bench -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-eager-baseline --no-cpu-baseline --no-inference > $R/gpurun_out/r03prof/bench_under_rocprof.json 2> $R/gpurun_out/r03prof/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03prof/infer -o ip -- python3 $R/tools/infer_profile.py > $R/gpurun_out/r03prof/infer.log 2>&1
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/r03prof/pmc.log 2>&1
echo profiles-done
