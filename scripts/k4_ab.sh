#!/bin/bash
# round 5: normal equations A/B on one box, interleaved: 0 = fp32-input kernel, 1 = K4h (tiles), 2 = K4w (one workgroup per instance)
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in c2_joint c2_joint2048; do for rep in 1 2; do for k in 0 1 2; do
  python $R/bench.py --workload $w --k4 $k --steps 5 --warmup 1 --no-cpu-baseline --no-exact 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w k4=$k:', o['value'])"
done; done; done
