#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-substring> "<counters>" -- <bench args...>   (run on the GPU box)
# counters in their own run, kernel-trace only (no other trace domains)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; pat=$2; ctr=$3; shift 4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$R/gpurun_out/pmc_$tag/p_counter_collection.csv')):
    if '$pat' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('$tag', k, 'mean/launch', sum(v)/len(v), 'n', len(v))
PY
