#!/bin/bash
# rocprofv3 kernel stats of the shipped configurations at their real render-block sizes (bench.py --shipped-only), round 4; file names carry the round they were last taken in
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for c in configs0_wild_pepper configs2_challenge_pepper configs4_lab_pepper_berry; do
  rm -rf /tmp/prof_$c && mkdir -p /tmp/prof_$c
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o p -- python $R/bench.py --shipped-only $c --groups 1 --steps 3 --warmup 1 > $R/gpurun_out/r06_${c}_bench_under_rocprof.json 2> /tmp/prof_$c.err)
  f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
  python - "$f" > $R/gpurun_out/r06_${c}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats: bench.py --shipped-only (--groups 1: one stream, un-overlapped kernel durations; 1 warm-up + 3 timed + 1 counted optimisation of 64 fruits, L = 32, f16x3)")
print(f"{'kernel':44s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'share':>7s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n[:44]:44s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}%")
PY
done
