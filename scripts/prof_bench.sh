#!/bin/bash
# prof_bench.sh [bench args]: per-kernel average durations of one short bench run (rocprofv3 kernel trace + stats)
export TMPDIR=/tmp
rm -rf /tmp/pb && mkdir -p /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact "$@" > /tmp/pb/log 2>&1
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("k_"):
        print(f"{n.split('(')[0]:28s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
tail -1 /tmp/pb/log | cut -c80-135
