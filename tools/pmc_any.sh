#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <outdir> -- <command...>   dynamic instruction mix + time of the kernels of any command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; shift; case "$O" in /*) ;; *) O=$R/$O;; esac
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/raw $O/raw2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/raw -o p -- "$@" > $O/pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw2 -o p -- "$@" > $O/stats.log 2>&1
python - <<PY
import csv, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/raw/p_counter_collection.csv")):
    acc[r["Kernel_Name"][:105]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open("$O/raw2/p_kernel_stats.csv")):
    dur[r["Name"][:105]] = (int(r["Calls"]), float(r["AverageNs"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    f = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0)
    print("%-106s calls %4d avg %9.1f us  VALU %.3e FP64 %.3e (%.0f%%) LDS %.2e waves %.2e" % (k, dur.get(k, (0, 0))[0], dur.get(k, (0, 0))[1] / 1e3, m.get("SQ_INSTS_VALU", 0), f, 100 * f / max(m.get("SQ_INSTS_VALU", 1), 1), m.get("SQ_INSTS_LDS", 0), m.get("SQ_WAVES", 0)))
PY
rm -rf $O/raw $O/raw2
