#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <kernel-substring> "<counters>" -- <command...>   (run on the GPU box)
# like pmc_kernel.sh for an arbitrary command (paths relative to the repo root); also prints the kernel's mean time
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; pat=$2; ctr=$3; shift 4
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- "$@" ) > $R/gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$R/gpurun_out/pmc_$tag/p_counter_collection.csv')):
    if '$pat' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        acc['VGPR'].append(float(r.get('VGPR_Count', 0) or 0)); acc['AGPR'].append(float(r.get('Accum_VGPR_Count', 0) or 0))
        acc['SCRATCH'].append(float(r.get('Scratch_Size', 0) or 0)); acc['LDS'].append(float(r.get('LDS_Block_Size', 0) or 0))
for k, v in sorted(acc.items()):
    print('$tag', k, 'mean/launch', sum(v)/len(v), 'n', len(v))
PY
