#!/usr/bin/env bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root):
#   rocprofv3 kernel stats of the headline bench (in situ: hipGraph replays), HBM traffic (FETCH_SIZE / WRITE_SIZE in their own
#   passes, as the MI355X guide prescribes), SQ counters (MFMA busy, LDS bank conflicts, wave wait states) for the GEMM and
#   attention kernels, the split-K phase trace, the norm-pass and attention-variant micro-benchmarks, and the bench JSON lines
#   of the other configurations.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
set -u
# (every rocprofv3 call runs under `timeout` and writes csv: a call that writes the default rocpd database did not exit on this pool)
export FLUX_ALLOW_RANDOM_INIT=1      # the loaders random-initialise only on request (no checkpoints in this image)
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o kt -- $B --steps 5 --warmup 2 --profile-only --no-other-configs >/dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- $B --steps 1 --warmup 1 --no-graph --profile-only --dump-plan-bytes $O/r06_plan_bytes.json >/dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- $B --steps 1 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p4 -o a -- $B --steps 1 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p5 -o b -- $B --steps 1 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -o kt -- python $R/tools/prof_vae.py >/dev/null 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$(echo "$set" | md5sum | cut -c1-6)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pa_$i -o a -- python $R/tools/attn_one.py >/dev/null 2>&1
done
cd $R
python tools/prof_summary.py $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_bench_n1.csv > /dev/null
# (round 6: the algorithmic bytes of every GEMM of the plan next to the counter bytes -> columns algorithmic_MB, ratio; bench.py reads the ratio's inputs)
python tools/pmc_summary.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/r06_hbm_traffic_pmc.csv "flux-schnell B1 T1280" $O/r06_plan_bytes.json > /dev/null
python tools/prof_summary.py $(find /tmp/p6 -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_vae_fp32.csv > /dev/null
{ echo "# rocprofv3 --pmc (two passes of 4 SQ counters) -- python bench.py --steps 1 --warmup 1 --no-graph --profile-only"
  echo "# per-kernel averages per launch, summed over the launch's waves; SQ_WAVE_CYCLES / WAIT / ACTIVE in quad-cycles, MFMA_BUSY in cycles"
  cat $(find /tmp/p4 -name "*counter_collection.csv" | head -1) > /tmp/pmc_all.csv
  tail -n +2 $(find /tmp/p5 -name "*counter_collection.csv" | head -1) >> /tmp/pmc_all.csv
  python tools/pmc_gemm.py /tmp/pmc_all.csv | grep -v "attn_kernel" ; } > $O/r06_gemm_pmc.txt
{ echo "# rocprofv3 --pmc (3 passes) -- python tools/attn_one.py ; attn_kernel<128, 4, 0, 2, 1> (two wave sets), B=1 H=24 T=1280"
  f=$(ls /tmp/pa_*/*/*counter_collection.csv /tmp/pa_*/*counter_collection.csv 2>/dev/null | head -1)
  head -1 $f > /tmp/pmc_attn.csv
  for g in $(find /tmp/pa_* -name "*counter_collection.csv"); do tail -n +2 $g >> /tmp/pmc_attn.csv; done
  python tools/pmc_gemm.py /tmp/pmc_attn.csv ; } > $O/r06_attention_pmc.txt
$B --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_r06_n1.json
python tools/bench_text.py 2>/dev/null | tail -1 > $O/bench_r06_text.json
$B --fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_r06_fp8_b4_1024.json
$B --model flux-dev --image-size 1024 --denoise-steps 28 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_r06_dev1024_n1.json
BENCH_FORCE_DIST=1 $B --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_r06_n1_rccl_ws1.json
python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/bench_r06_sdxl_b16.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -o kt -- python $R/tools/bench_sdxl.py >/dev/null 2>&1 )
python tools/prof_summary.py $(find /tmp/p7 -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_sdxl_b16.csv > /dev/null
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -o kt -- python $R/bench.py --fp8 --steps 2 --warmup 1 --profile-only >/dev/null 2>&1 )
python tools/prof_summary.py $(find /tmp/p8 -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_fp8_b4_1024.csv > /dev/null
python tools/rs_phase_trace.py 3 > $O/r06_splitk_phase_trace.txt 2>&1
python tools/gemm_phase_trace2.py > $O/r06_gemm_phase_trace_nonsplit.txt 2>&1
python tools/attn_bench.py 0 > $O/r06_attn_bench.txt 2>&1
timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_text_gpu.py -q -s -m gpu -k "fp8" 2>&1 | grep -i "fp8\|passed\|failed" > $O/r06_fp8_measured.txt
timeout 1500 python -m pytest tests/test_full_size_parity_gpu.py -q -s -m gpu 2>&1 | grep "^\[\|passed\|failed" > $O/r06_parity_run.txt
cp $O/parity_full_size.json $O/r06_parity_full_size.json 2>/dev/null
