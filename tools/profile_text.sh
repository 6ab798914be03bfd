export FLUX_ALLOW_RANDOM_INIT=1
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o kt -- python $R/tools/bench_text.py >/dev/null 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/pt -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kernel_stats_text.csv > /dev/null
head -24 gpurun_out/r04_kernel_stats_text.csv | cut -c1-150
