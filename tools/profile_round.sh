cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $R/bench.py --steps 2 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $R/bench.py --steps 2 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) gpurun_out/r01_kernel_stats_bench_n1.csv > /dev/null
python tools/pmc_summary.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) gpurun_out/r01_hbm_traffic_pmc.csv > /dev/null
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_r01_n1.json
head -12 gpurun_out/r01_kernel_stats_bench_n1.csv; head -12 gpurun_out/r01_hbm_traffic_pmc.csv
