#!/usr/bin/env bash
set -u
export FLUX_ALLOW_RANDOM_INIT=1
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_full_size_parity_gpu.py -k "not test_gemm_bias" --last-failed-no-failures all > $O/gputest_rest.log 2>&1; tail -3 $O/gputest_rest.log
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_ln_$i.json
  FLUXHIP_EXPERIMENT_SKIP_LN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_skipln_$i.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06f/bench_*.json')):
    d=json.load(open(f)); c=d['config']; print(f, round(d['value'],3), round(d['ms_per_step'],3), round(c['denoise_step_ms_in_loop'],3), round(c['denoise_step_ms'],3))
P
