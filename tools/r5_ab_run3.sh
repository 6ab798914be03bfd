set -u
B=$PWD/flux_generator_amd/lib_ab/base/libfluxhip.so
O=gpurun_out/r5c
mkdir -p $O
export FLUX_ALLOW_RANDOM_INIT=1
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_flux_gpu.py tests/test_text_gpu.py tests/test_sd_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "attention or lora or http or c1_" > $O/tests2.log 2>&1
tail -5 $O/tests2.log
echo "--- attention: base lib vs new lib"
FLUXHIP_LIB=$B FLUXHIP_LIB_AB=1 timeout 300 python tools/attn_bench.py 0 > $O/attn_base.txt 2>&1
timeout 300 python tools/attn_bench.py 0 > $O/attn_new.txt 2>&1
paste -d'\n' $O/attn_base.txt $O/attn_new.txt
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1; }
for i in 1 2; do
  FLUXHIP_LIB=$B run > $O/bench_base_$i.json
  run > $O/bench_new_$i.json
  FLUXHIP_PLAN_TILES="3072x12288=825,3072x15360=825" run > $O/bench_new_cfg57_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/bench_*.json')):
    try:
        d=json.load(open(f)); c=d['config']; kb=c['kernel_breakdown_one_forward']
        print(f.split('/')[-1], round(d['value'],3), round(c['denoise_step_ms_in_loop'],3), round(c['denoise_mfma_frac_in_loop'],4),
              'attn', kb['fluxhip_attention_d128_bf16']['ms'], {k.split('/')[-1]:v['ms'] for k,v in kb.items() if 'gemm' in k and v['ms']>0.5})
    except Exception as e: print(f, 'ERR', e)
PY
