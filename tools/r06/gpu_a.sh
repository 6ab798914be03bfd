#!/usr/bin/env bash
# round 6, GPU call A: the new 128x160 tile (tests + sweep), the live fixtures, mods-ahead A/B
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest -x -q -m gpu tests/test_ops_gpu.py -k "gemm_bias or epilogue_paths" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
timeout 900 python -m pytest -x -q -m gpu -s tests/test_sd_f16_gpu.py -k "gemm_f16 or gelu_erf or float16_is" > $O/t_f16.log 2>&1; tail -3 $O/t_f16.log; grep "gelu_erf float16" $O/t_f16.log
timeout 600 python -m pytest -x -q -m gpu -s tests/test_vae_gpu.py -k "elementwise or groupnorm_x3" > $O/t_vae.log 2>&1; tail -3 $O/t_vae.log; grep "groupnorm+silu" $O/t_vae.log
timeout 600 python -m pytest -x -q -m gpu tests/test_flux_gpu.py > $O/t_flux.log 2>&1; tail -3 $O/t_flux.log
timeout 1500 python -m pytest -x -q -m gpu -s tests/test_full_size_parity_gpu.py -k "stored" > $O/t_full.log 2>&1; tail -3 $O/t_full.log; grep "^\[c" $O/t_full.log
TUNE_F16=1 TUNE_GRAPH=1 TUNE_SHAPES="o:4096:1280:1280:2,q:4096:1280:1280:0,ff2:4096:1280:5120:2,qk:4096:2560:1280:0,o32:16384:640:640:2,q32:16384:640:640:0,ff2_32:16384:640:2560:2" python tools/gemm_tune.py 0 55 57 54 49 51 47 > $O/tune_f16.txt 2>&1; grep BEST $O/tune_f16.txt
TUNE_GRAPH=1 TUNE_SHAPES="proj:1280:3072:3072:2,projdev:4608:3072:3072:2,qkv:1280:9216:3072:0" python tools/gemm_tune.py 0 47 57 51 55 > $O/tune_bf16.txt 2>&1; grep BEST $O/tune_bf16.txt
for i in 1 2; do
  FLUXHIP_MOD_AHEAD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_ahead0_$i.json
  FLUXHIP_MOD_AHEAD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_ahead1_$i.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06a/bench_ahead*.json')):
    try:
        d=json.load(open(f)); c=d['config']
        print(f, round(d['value'],3), round(d['ms_per_step'],3), round(c['denoise_step_ms_in_loop'],3), round(c['vae_decode_ms'],3), {k:v['ms'] for k,v in list(c['kernel_breakdown_one_forward'].items())[:8]})
    except Exception as e: print(f, 'ERR', e)
P
