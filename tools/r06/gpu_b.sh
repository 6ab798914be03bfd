#!/usr/bin/env bash
# round 6, GPU call B: deeper-ring 128x160 tiles, V^T fork A/B on the SDXL UNet step
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out/r06b; mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu -s tests/test_vae_gpu.py -k "elementwise" > $O/t_vae.log 2>&1; tail -2 $O/t_vae.log; grep "groupnorm+silu" $O/t_vae.log
timeout 900 python -m pytest -x -q -m gpu tests/test_ops_gpu.py -k "gemm_bias" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
TUNE_F16=1 TUNE_GRAPH=1 TUNE_SHAPES="o:4096:1280:1280:2,q:4096:1280:1280:0,ff2:4096:1280:5120:2,qk:4096:2560:1280:0,o32:16384:640:640:2,ff2_32:16384:640:2560:2" python tools/gemm_tune.py 0 55 57 58 59 54 > $O/tune_f16.txt 2>&1; grep BEST $O/tune_f16.txt
TUNE_GRAPH=1 TUNE_SHAPES="proj:1280:3072:3072:2,t5wo:256:4096:10240:0" python tools/gemm_tune.py 0 47 57 58 59 > $O/tune_bf16.txt 2>&1; grep BEST $O/tune_bf16.txt
for i in 1 2; do
  FLUXHIP_UNET_VT_FORK=0 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_fork0_$i.json
  FLUXHIP_UNET_VT_FORK=1 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_fork1_$i.json
done
for f in $O/sdxl_fork*.json; do echo $f; cut -c1-260 $f; done
