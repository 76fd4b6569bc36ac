#!/bin/bash
# Same-box A/B of the working tree against a previous commit unpacked and built under ab_old/ (git archive <rev> | tar -x -C ab_old;
# python -m hortimapping_amd.build there): C2-joint primary workload only, interleaved, two rounds.
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["value"], "inst/s  main launch", d["roofline"]["avg_launch_ms"], "ms")'
A="--steps 5 --warmup 1 --no-cpu-baseline --no-exact --no-shipped"
for i in 1 2; do
  (cd ab_old && python bench.py $A 2>/dev/null | python -c "$P" "old tree (K4h default)      ")
  python bench.py $A --k4 1 2>/dev/null | python -c "$P" "new tree --k4 1              "
  python bench.py $A 2>/dev/null | python -c "$P" "new tree (fp32 K4 default)   "
  (cd ab_old && python bench.py $A --k4 0 2>/dev/null | python -c "$P" "old tree --k4 0              ")
done
