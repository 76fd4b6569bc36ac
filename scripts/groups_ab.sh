#!/bin/bash
# round 5: 2 vs 3 instance groups, interleaved, on the headline workload, the literal joint-2048 reading and the shipped configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in c2_joint c2_joint2048; do for rep in 1 2; do for g in 2 3; do
  python $R/bench.py --workload $w --groups $g --steps 5 --warmup 1 --no-cpu-baseline --no-exact 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w groups $g:', o['value'])"
done; done; done
for c in configs0_wild_pepper configs2_challenge_pepper configs4_lab_pepper_berry; do for g in 0 3 0 3; do
  python $R/bench.py --shipped-only $c --groups $g --steps 3 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=list(o)[0]; print(k, 'groups $g:', o[k]['value'])"
done; done
