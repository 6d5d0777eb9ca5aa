#!/bin/bash
# usage: bash tools/ab_2d.sh "ENV.." ...  -- 2-D Jacobian bench line (1000^2) per environment set
for e in "$@"; do
  env $e python bench.py --dim 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
