#!/usr/bin/env bash
# Round-5 closing profile set (the library at the round's last kernel commit): the subset of tools/profile_round5.sh that the
# bench line and DESIGN's tables quote - kernel stats of the headline bench (hipGraph replays), HBM traffic (FETCH_SIZE / WRITE_SIZE
# in their own passes, as the MI355X guide prescribes), kernel stats of C4 (SDXL batch 16) and C5 (fp8, B = 4, 1024^2), the
# attention micro-benchmark.  Run through gpurun from the repo root; everything lands in gpurun_out/, the summaries are copied to
# profiles/ by hand.
set -u
export FLUX_ALLOW_RANDOM_INIT=1
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o kt -- $B --steps 5 --warmup 2 --profile-only --no-other-configs >/dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- $B --steps 1 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- $B --steps 1 --warmup 1 --no-graph --profile-only >/dev/null 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats_bench_n1.csv > /dev/null
python tools/pmc_summary.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/r05_hbm_traffic_pmc.csv > /dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -o kt -- python $R/tools/bench_sdxl.py >/dev/null 2>&1 )
python tools/prof_summary.py $(find /tmp/p7 -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats_sdxl_b16.csv > /dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -o kt -- python $R/bench.py --fp8 --steps 2 --warmup 1 --profile-only >/dev/null 2>&1 )
python tools/prof_summary.py $(find /tmp/p8 -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats_fp8_b4_1024.csv > /dev/null
python tools/attn_bench.py 0 > $O/r05_attn_bench.txt 2>&1
$B --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_r05_n1.json
ls -la $O | tail -20
