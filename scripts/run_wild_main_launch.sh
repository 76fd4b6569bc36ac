export TMPDIR=/tmp
python scripts/gpu_wild_main_launch.py 2>&1 | grep -v amdgpu
rm -rf /tmp/pw && mkdir -p /tmp/pw
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_wild_main_launch.py > /tmp/pw/log 2>&1)
f=$(find /tmp/pw -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n[:44]:44s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} us  {float(r['TotalDurationNs'])/1e6:10.2f} ms")
PY
