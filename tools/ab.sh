#!/bin/bash
# usage: bash tools/ab.sh "ENV1=.. ENV2=.." "ENVA=.." ...   -- one bench line per environment set (10 steps, no extras)
for e in "$@"; do
  env $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
