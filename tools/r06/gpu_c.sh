#!/usr/bin/env bash
# round 6, GPU call C: same-box build-to-build A/B (round-5 library vs this tree) on the SDXL UNet step and the headline
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out/r06c; mkdir -p $O
OLD=$(pwd)/flux_generator_amd/lib_ab/libfluxhip.so
for i in 1 2; do
  FLUXHIP_LIB=$OLD FLUXHIP_LIB_AB=1 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_r5lib_$i.json
  python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_r6lib_$i.json
done
for i in 1 2; do
  FLUXHIP_LIB=$OLD FLUXHIP_LIB_AB=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_r5lib_$i.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_r6lib_$i.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06c/sdxl_*.json')):
    d=json.load(open(f)); print(f, round(d['unet_step_ms'],3), round(d['vae_decode_ms'],2), round(d['images_per_sec'],2))
for f in sorted(glob.glob('gpurun_out/r06c/bench_*.json')):
    d=json.load(open(f)); c=d['config']; print(f, round(d['value'],3), round(d['ms_per_step'],3), round(c['denoise_step_ms_in_loop'],3), round(c['vae_decode_ms'],3))
P
