#!/usr/bin/env bash
# round 6: balanced staging (FLUXHIP_PP_GA) - correctness, then build-to-build A/B (GA=0 library in lib_ab)
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python -m pytest -x -q -m gpu tests/test_ops_gpu.py -k "gemm" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
timeout 600 python -m pytest -x -q -m gpu tests/test_flux_gpu.py tests/test_golden_gpu.py tests/test_sd_f16_gpu.py -k "not pipeline" > $O/t_flux.log 2>&1; tail -2 $O/t_flux.log
python tools/lib_ab.py ga0=flux_generator_amd/lib_ab/libfluxhip.so ga1=flux_generator_amd/lib/libfluxhip.so > $O/lib_ab.txt 2>&1; grep -v amdgpu $O/lib_ab.txt
python tools/gemm_phase_trace2.py 2>/dev/null | grep -v amdgpu > $O/phase_trace.txt; cat $O/phase_trace.txt
OLD=$(pwd)/flux_generator_amd/lib_ab/libfluxhip.so
for i in 1 2; do
  FLUXHIP_LIB=$OLD FLUXHIP_LIB_AB=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_ga0_$i.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_ga1_$i.json
done
FLUXHIP_LIB=$OLD FLUXHIP_LIB_AB=1 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_ga0.json
python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_ga1.json
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06g/bench_*.json')):
    d=json.load(open(f)); c=d['config']; print(f, round(d['value'],3), round(d['ms_per_step'],3), round(c['denoise_step_ms_in_loop'],3), round(c['vae_decode_ms'],3), {k:v['ms'] for k,v in list(c['kernel_breakdown_one_forward'].items())[:5]})
for f in sorted(glob.glob('gpurun_out/r06g/sdxl_*.json')):
    d=json.load(open(f)); print(f, round(d['unet_step_ms'],3), round(d['vae_decode_ms'],2))
P
