#!/bin/bash
# Builds libneosr_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${NEOSR_AMD_OUT:-$HERE/../lib}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm"
OBJS=""
PIDS=""
for f in api prof conv_mfma conv_glds conv_wino conv_wino4 conv_wino4_chain conv_thin wgrad elementwise degrade layers gemm_mfma attn attn_wave attn_flash cab augment ssim color losses optim nets blocks; do
  "$HIPCC" $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" "$@" &
  PIDS="$PIDS $!"
  OBJS="$OBJS $OUT/$f.o"
done
for p in $PIDS; do wait "$p" || { echo "compile failed" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libneosr_amd.so"
echo "built $OUT/libneosr_amd.so"
