#!/bin/bash
# usage (GPU box): bash tools/pmc_stall.sh <outdir> -- <command...>   where the waves of every kernel of the command spend their cycles
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; shift; case "$O" in /*) ;; *) O=$R/$O;; esac
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/raw
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/raw -o a -- "$@" > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/raw -o b -- "$@" > $O/pmc_b.log 2>&1
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/raw/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:105]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    print("   " + " ".join("%s=%.3e" % (c.replace("SQ_", ""), v) for c, v in sorted(m.items())))
PY
rm -rf $O/raw
