#!/bin/bash
# per-kernel average durations of the drop-in single-fruit path (scripts/single_fruit_latency.py) under rocprofv3
export TMPDIR=/tmp
rm -rf /tmp/psf && mkdir -p /tmp/psf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psf -o p -- python scripts/single_fruit_latency.py > /tmp/psf/log 2>&1
f=$(find /tmp/psf -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{n.split('(')[0][:44]:44s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} share={100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
tail -12 /tmp/psf/log
