#!/bin/bash
# Builds libovla_hip.so (gfx950 only) in-tree next to the sources.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libovla_hip.so
SRCS="core.hip gemm_nt.hip gemm_tn.hip attention.hip elementwise.hip head_optim.hip selftest.hip"
newest=$(ls -t $SRCS common.h ../../include/ovla.h build.sh | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then echo "libovla_hip.so up to date"; exit 0; fi
mkdir -p ../_build
pids=()
for s in $SRCS; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -c "$s" -o "../_build/${s%.hip}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""; for s in $SRCS; do objs="$objs ../_build/${s%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $(realpath $OUT)"
