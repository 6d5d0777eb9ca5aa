#!/bin/bash
# usage (GPU box): bash tools/pmc_cache.sh <outdir> [env...] -- instruction / scalar-data cache counters of one bench.py run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; case "$O" in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/raw
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH \
   --output-format csv -d $O/raw -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_cache.log 2>&1
cp $O/raw/p_counter_collection.csv $O/pmc_cache.csv 2>/dev/null; rm -rf $O/raw
python - <<PY
import csv, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/pmc_cache.csv")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m and "pfm" in r["Kernel_Name"]:
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if k.startswith("k_cart"):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
