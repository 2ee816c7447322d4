#!/bin/bash
for cfg in "256 3" "512 2" "512 1"; do
  set -- $cfg
  THALLO_THREADS=$1 THALLO_PER_CU=$2 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('THREADS=$1 PER_CU=$2', 'it/s=%.0f' % d['value'], 'step1_us=%.1f' % (d['roofline']['avg_launch_ms']*1e3), 'step2_us=%.1f' % (d['roofline']['pcg_step2']['avg_launch_ms']*1e3), 'cost', d['final_cost'])
"
done
