#!/usr/bin/env bash
# SQ / GRBM counters (three --pmc passes) and launch durations of the fp8 GEMM kernels of `bench.py --fp8` (C5 shape), run from the repo root on the GPU box
R=$(pwd); cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --fp8 --steps 1 --warmup 1 --no-graph --profile-only"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/q1 -o a -- $B >/dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/q2 -o b -- $B >/dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/q3 -o c -- $B >/dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/q4 -o kt -- $B >/dev/null 2>&1
cd $R
cat $(find /tmp/q1 -name "*counter_collection.csv" | head -1) > /tmp/pmc_all.csv
tail -n +2 $(find /tmp/q2 -name "*counter_collection.csv" | head -1) >> /tmp/pmc_all.csv
tail -n +2 $(find /tmp/q3 -name "*counter_collection.csv" | head -1) >> /tmp/pmc_all.csv
python tools/pmc_gemm.py /tmp/pmc_all.csv > gpurun_out/r03_fp8_gemm_pmc.txt
python - <<'P' >> gpurun_out/r03_fp8_gemm_pmc.txt
import csv,glob,collections
f=glob.glob('/tmp/q4/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm_nt_kernel' in n or 'attn_kernel' in n:
        d[n.split('(')[0]+' grid='+str(r.get('Grid_Size') or r.get('Grid_Size_X'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print('DUR',k,len(v),'avg_us',sum(v)/len(v))
P
grep -A14 "256, 224, 4, 2, 0, 2, 6, 4" gpurun_out/r03_fp8_gemm_pmc.txt | head -60; grep DUR gpurun_out/r03_fp8_gemm_pmc.txt
