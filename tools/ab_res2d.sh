#!/bin/bash
# usage: bash tools/ab_res2d.sh "ENV.." ...  -- 2-D residual-only bench line (1000^2) per environment set
for e in "$@"; do
  env $e python bench.py --dim 2 --residual-only --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done
