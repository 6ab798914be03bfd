#!/usr/bin/env bash
# Build libfluxhip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU present.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${FLUXHIP_OUT_DIR:-../lib}
BUILD=${FLUXHIP_BUILD_DIR:-build}      # (A/B builds: another object directory and output directory)
mkdir -p "$OUT" "$BUILD"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -packed-fp32-ops: no compiler-formed v_pk_{mul,add,fma}_f32.  Round 4 traced run-to-run differences of the whole denoise step
# under GPU sharing (two processes on one MI355X) to ONE such sequence in qk_norm_rope_vt_kernel whose low results were wrong in
# lanes 48-63 next to a co-tenant and bit-stable alone (norm.hip; tools/flux_contention_bisect.py).  That statement is now four
# plain VALU instructions in inline asm, and the formation is switched off for the whole library: a same-box A/B of the two
# builds measures no difference (22.05 / 21.98 vs 22.09 / 22.01 images/s, denoise step 18.00 / 18.04 vs 18.04 / 18.07 ms).
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops ${FLUXHIP_EXTRA_FLAGS:-}"
SRCS="api gemm gemm_conv gemm_x3f8 gemm_mx gemm_f16 gemm_conv_f16 small_linear norm groupnorm attention elementwise unet_ops unet_x3"
pids=()
for s in $SRCS; do
  if [ ! -f $BUILD/$s.o ] || [ $s.hip -nt $BUILD/$s.o ] || [ common.h -nt $BUILD/$s.o ] || \
     [ gemm_core.h -nt $BUILD/$s.o ] || [ gemm_tiles.h -nt $BUILD/$s.o ] || [ ../../include/fluxhip.h -nt $BUILD/$s.o ]; then
    $HIPCC $FLAGS -c $s.hip -o $BUILD/$s.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
objs=""
for s in $SRCS; do objs="$objs $BUILD/$s.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libfluxhip.so
echo "built $OUT/libfluxhip.so"
