#!/bin/bash
# usage: tools/prof_kernel.sh <tag> <kernel-substring> -- <bench args...>   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; pat=$2; shift 3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/prof_$tag/p_kernel_stats.csv')):
    if '$pat' in r['Name']:
        print('$tag', r['Name'].split('(')[0][-40:], r['Calls'], round(float(r['AverageNs'])/1e6, 4), 'ms')
PY
