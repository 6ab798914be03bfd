#!/usr/bin/env bash
# A/B of two libfluxhip builds on the attention shapes (tools/attn_bench.py) and in situ (bench.py), interleaved, one box.
# usage (GPU box, repo root): bash tools/ab_attn.sh        (ab/libfluxhip_base.so = the build to compare against)
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_flux_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "attn or attention or flux or golden" > $O/ab_attn_tests_full.txt 2>&1
grep -E "passed|failed|error" $O/ab_attn_tests_full.txt | tail -2
{
for i in 1 2; do
echo "## base"; FLUXHIP_LIB=ab/libfluxhip_base.so FLUXHIP_LIB_AB=1 python tools/attn_bench.py 0 2>&1 | grep "^B"
echo "## new";  python tools/attn_bench.py 0 2>&1 | grep "^B"
done
} > $O/ab_attn_micro.txt 2>&1
cat $O/ab_attn_micro.txt
ex='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; k=c["kernel_breakdown_one_forward"]; print(round(d["value"],3), round(c["denoise_step_ms_in_loop"],3), round(c["denoise_step_ms"],3), k["fluxhip_attention_d128_bf16"]["ms"])'
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
{
for i in 1 2; do
  echo -n "base  "; FLUXHIP_LIB=ab/libfluxhip_base.so FLUXHIP_LIB_AB=1 $B 2>/dev/null | tail -1 | python -c "$ex"
  echo -n "new   "; $B 2>/dev/null | tail -1 | python -c "$ex"
done
} > $O/ab_attn_insitu.txt 2>&1
cat $O/ab_attn_insitu.txt
