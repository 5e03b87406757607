#!/bin/bash
# Builds libovla_hip.so (gfx950 only) in-tree next to the sources.  hipcc cross-compiles without a GPU.
#   build.sh          the product library
#   build.sh ablate   libovla_hip_ablate.so: the same sources with -DOVLA_GEMM_ABLATE (timing ablations of gemm_nt, tools/gemm_ablate.py)
# Staleness is decided by CONTENT, not mtime: the sha256 of every source / header / this script is compiled into the library
# (ovla_build_hash()) and written next to it; _lib.py refuses to load a library whose hash differs from the sources it sits beside.
set -euo pipefail
cd "$(dirname "$0")"
MODE="${1:-product}"
OUT=../libovla_hip.so; EXTRA=""; BUILD=../_build
if [ "$MODE" = "ablate" ]; then OUT=../libovla_hip_ablate.so; EXTRA="-DOVLA_GEMM_ABLATE"; BUILD=../_build_ablate; fi
SRCS="core.hip gemm_nt.hip gemm_tn.hip attention.hip elementwise.hip head_optim.hip selftest.hip"
HASH=$(cat $(ls *.hip *.h | LC_ALL=C sort) build.sh ../../include/ovla.h | sha256sum | cut -c1-32)
if [ -f "$OUT" ] && [ -f "$OUT.hash" ] && [ "$(cat "$OUT.hash")" = "$HASH" ]; then echo "$(basename $OUT) up to date ($HASH)"; exit 0; fi
mkdir -p $BUILD
pids=()
for s in $SRCS; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $EXTRA -DOVLA_SRC_HASH="\"$HASH\"" -c "$s" -o "$BUILD/${s%.hip}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""; for s in $SRCS; do objs="$objs $BUILD/${s%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "$HASH" > "$OUT.hash"
echo "built $(realpath $OUT) ($HASH)"
