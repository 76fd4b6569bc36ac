#!/bin/bash
# rocprofv3 kernel stats of the default bench schedule (two instance groups per call): kernel durations overlap here
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pg && mkdir -p /tmp/pg
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o p -- python $R/bench.py --steps 2 --warmup 1 --no-exact --no-cpu-baseline > $R/gpurun_out/r05_groups2_bench_under_rocprof.json 2> /tmp/pg/err)
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/r05_f16x3_c2_joint_groups2_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats: python bench.py --steps 2 (default schedule: two instance groups of 32 per call on two internal streams;")
print("kernels of the two groups overlap, so a duration here includes the other group's kernels sharing the CUs: compare r05_f16x3_c2_joint_kernel_stats.txt = one stream)")
print(f"{'kernel':44s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n[:44]:44s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f}")
PY
