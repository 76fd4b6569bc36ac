# same-box A/B of the number of instance groups per call (bench.py --groups), two repetitions each
for rep in 1 2; do for w in c2_joint c2_sdf; do for g in 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no-exact --no-cpu-baseline --workload $w --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w groups $g:', d['value'], d['ms_per_step'])"; done; done; done
