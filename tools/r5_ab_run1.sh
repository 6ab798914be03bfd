set -u
B=$PWD/flux_generator_amd/lib_ab/base/libfluxhip.so
N=$PWD/flux_generator_amd/lib/libfluxhip.so
mkdir -p gpurun_out/r5a
export FLUX_ALLOW_RANDOM_INIT=1
python tools/lib_ab.py base=$B new=$N > gpurun_out/r5a/ab_gemm.txt 2>&1
for i in 1 2; do
  FLUXHIP_LIB=$B python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r5a/bench_base_$i.json
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r5a/bench_new_$i.json
done
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_flux_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r5a/tests.log 2>&1
tail -3 gpurun_out/r5a/tests.log
cat gpurun_out/r5a/ab_gemm.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/bench_*.json')):
    try:
        d=json.load(open(f)); c=d['config']
        print(f, d['value'], c['denoise_step_ms_in_loop'], c['denoise_mfma_frac_in_loop'], c['vae_decode_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
