set -u
export FLUX_ALLOW_RANDOM_INIT=1
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -o kt -- python $R/bench.py --fp8 --steps 3 --warmup 1 --profile-only --no-other-configs >/dev/null 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/p7 -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_fp8_mx.csv > /dev/null
python bench.py --fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_r04_fp8_mx.json
head -16 $O/r04_kernel_stats_fp8_mx.csv | cut -c1-150
