#!/bin/bash
# experiment: cache-policy (non-temporal) masks for PCGStep1 (NT1) and PCGStep2 (NT2); prints PCG it/s and kernel means
for cfg in "0 0" "1 0" "3 0" "1 1" "1 3" "3 3" "0 3" "3 1" "67 3" "35 3" "7 3" "11 3" "3 7" "3 11"; do
  set -- $cfg
  THALLO_NT1=$1 THALLO_NT2=$2 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('NT1=$1 NT2=$2', 'it/s=%.0f' % d['value'], 'step1_us=%.1f' % (d['roofline']['avg_launch_ms']*1e3), 'step2_us=%.1f' % (d['roofline']['pcg_step2']['avg_launch_ms']*1e3), 'cost=%.6g' % d['final_cost'])
"
done
