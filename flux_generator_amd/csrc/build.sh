#!/usr/bin/env bash
# Build libfluxhip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU present.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
SRCS="api gemm gemm_conv gemm_x3f8 gemm_f16 gemm_conv_f16 small_linear norm groupnorm attention elementwise unet_ops"
pids=()
for s in $SRCS; do
  if [ ! -f build/$s.o ] || [ $s.hip -nt build/$s.o ] || [ common.h -nt build/$s.o ] || \
     [ gemm_core.h -nt build/$s.o ] || [ gemm_tiles.h -nt build/$s.o ] || [ ../../include/fluxhip.h -nt build/$s.o ]; then
    $HIPCC $FLAGS -c $s.hip -o build/$s.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
objs=""
for s in $SRCS; do objs="$objs build/$s.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libfluxhip.so
echo "built $OUT/libfluxhip.so"
