export TMPDIR=/tmp
rm -rf /tmp/pw && mkdir -p /tmp/pw
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_wild_tail.py > /tmp/pw/log 2>/tmp/pw/err)
grep active /tmp/pw/log
f=$(find /tmp/pw -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_decoder_h<0" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = len(d) // 2
print("main-launch durations [ms] of the second optimisation:", " ".join(f"{x:.2f}" for x in d[n:]))
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_decoder_h<1" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("render-Jacobian launch durations [ms]:", " ".join(f"{x:.2f}" for x in d[len(d) // 2:]))
PY
