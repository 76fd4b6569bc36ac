#!/bin/bash
# c2_joint2048 (the literal "2048 pts/instance" reading of the joint loop): group-count A/B and the one-stream kernel split
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
for g in 1 2 3; do
  python $R/bench.py --workload c2_joint2048 --groups $g --steps 3 --warmup 1 --no-cpu-baseline --no-exact 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groups $g:', o['value'], 'inst/s', o['ms_per_step'], 'ms/step')"
done > $out/r05_joint2048_groups.txt 2>&1
cat $out/r05_joint2048_groups.txt
d=/tmp/prof_j2048; rm -rf $d; mkdir -p $d
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python $R/bench.py --workload c2_joint2048 --groups 1 --steps 2 --warmup 1 --no-cpu-baseline --no-exact > $out/r05_joint2048_bench_under_rocprof.log 2>&1)
for k in kernel_stats kernel_trace; do f=$(find $d -name "*_$k.csv" | head -1); [ -n "$f" ] && mv "$f" $d/ks_$k.csv; done
python $R/scripts/summarize_prof.py $d ks $out/r05_joint2048_kernel_stats.txt
head -20 $out/r05_joint2048_kernel_stats.txt
