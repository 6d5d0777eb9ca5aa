#!/bin/bash
# usage (GPU box): bash tools/pmc_wr.sh <outdir> [env...]   L2 -> memory write requests of the kernels of one bench.py run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; case "$O" in /*) ;; *) O=$R/$O;; esac
mkdir -p $O; cd /tmp && export TMPDIR=/tmp; rm -rf $O/raw
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_WRITE_sum --output-format csv -d $O/raw -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc.log 2>&1
python - <<PY
import csv, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/raw/p_counter_collection.csv")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m and "k_cart" in r["Kernel_Name"]:
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
rm -rf $O/raw
