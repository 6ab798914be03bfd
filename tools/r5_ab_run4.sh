set -u
B=$PWD/flux_generator_amd/lib_ab/base/libfluxhip.so
O=gpurun_out/r5d
mkdir -p $O
export FLUX_ALLOW_RANDOM_INIT=1
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/tests_full.log 2>&1
tail -8 $O/tests_full.log
timeout 300 python tools/gemm_phase_trace2.py > $O/phase_trace2.txt 2>&1; cat $O/phase_trace2.txt
timeout 300 python tools/rs_phase_trace.py 3 > $O/rs_phase_trace.txt 2>&1; cat $O/rs_phase_trace.txt
for i in 1 2; do
  FLUXHIP_LIB=$B timeout 600 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_base_$i.json
  timeout 600 python tools/bench_sdxl.py 2>/dev/null | tail -1 > $O/sdxl_new_$i.json
done
head -c 1200 $O/sdxl_base_1.json; echo; head -c 1200 $O/sdxl_new_1.json; echo
