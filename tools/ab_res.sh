#!/bin/bash
# usage: bash tools/ab_res.sh "ENV.." ...  -- residual-only bench line per environment set
for e in "$@"; do
  env $e python bench.py --steps 20 --warmup 3 --residual-only --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
