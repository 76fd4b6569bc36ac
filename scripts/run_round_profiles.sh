#!/bin/bash
# run_round_profiles.sh <rNN>: the round's evidence on the FINAL build, one gpurun call (copy gpurun_out/<rNN>_* to profiles/):
#   full default bench line, kernel stats + PMC passes of the C2-joint workload (one stream), kernel stats of the shipped
#   configurations, K1p launch times.
R=$1
mkdir -p gpurun_out
python bench.py > gpurun_out/${R}_bench_c2_joint.json 2> gpurun_out/${R}_bench_c2_joint.err
bash scripts/collect_profiles.sh ${R}_f16x3_c2_joint --groups 1 --no-shipped > gpurun_out/${R}_collect.log 2>&1
bash scripts/prof_shipped.sh > gpurun_out/${R}_prof_shipped.log 2>&1
python scripts/gpu_time_k1p.py final > gpurun_out/${R}_k1p_launch_times.txt 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${R}_bench_c2_joint.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('readings_inst_per_s'))"
cat gpurun_out/${R}_f16x3_c2_joint_kernel_stats.txt | head -20
cat gpurun_out/${R}_traffic.json
