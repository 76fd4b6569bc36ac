#!/bin/bash
# round 5: 2 vs 3 instance groups, interleaved, on the headline workload and the shipped configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for g in 2 3; do
  python $R/bench.py --groups $g --steps 5 --warmup 1 --no-cpu-baseline --no-exact --no-shipped 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2_joint groups $g:', o['value'], 'c2_joint2048', o['c2_joint2048']['value'], 'c2_sdf', o['c2_sdf']['value'], 'batch_256', o['batch_256']['value'])"
done; done
for c in configs0_wild_pepper configs2_challenge_pepper configs4_lab_pepper_berry; do for g in 0 3 0 3; do
  python $R/bench.py --shipped-only $c --groups $g --steps 3 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=list(o)[0]; print(k, 'groups $g:', o[k]['value'])"
done; done
