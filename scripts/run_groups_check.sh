timeout 800 python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -4
for g in 0 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no-exact --no-cpu-baseline --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('groups $g:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['step']['frac'])"; done
