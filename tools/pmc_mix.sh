#!/bin/bash
# usage (GPU box): bash tools/pmc_mix.sh <outdir> [env assignments...] -- dynamic instruction mix of one bench.py run
# One rocprofv3 PMC pass (kernel trace + SQ counters only); writes <outdir>/pmc_mix.csv and prints per-kernel means.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; case "$O" in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_mix_raw
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS \
   --output-format csv -d $O/pmc_mix_raw -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mix.log 2>&1
cp $O/pmc_mix_raw/p_counter_collection.csv $O/pmc_mix.csv 2>/dev/null
rm -rf $O/pmc_mix_raw
python - <<PY
import csv, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/pmc_mix.csv")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m and "pfm" in r["Kernel_Name"]:
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    f64 = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0)
    m["fp64_share_of_valu"] = f64 / max(m.get("SQ_INSTS_VALU", 1), 1)
    out[k] = m
    print(k, {a: ("%.4g" % b) for a, b in m.items()})
json.dump(out, open("$O/instruction_mix_dynamic.json", "w"), indent=1)
PY
