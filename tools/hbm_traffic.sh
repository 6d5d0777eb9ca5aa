#!/bin/bash
# usage: tools/hbm_traffic.sh <out.json> -- <bench args...>      (run on the GPU box)
# HBM bytes per launch of every pfm kernel: two separate PMC passes (WRITE_SIZE, FETCH_SIZE), kernel-trace only.
# rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B
# (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so fetch bytes are doubled.  WRITE_SIZE was calibrated on
# k_cart_uu3, whose only stores are the 19.86 GB of (u,u) values (measured 19.81 GB).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift 2
cd /tmp && export TMPDIR=/tmp
for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf $R/gpurun_out/pmc_$ctr
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$ctr -o p -- python $R/bench.py "$@" > $R/gpurun_out/pmc_$ctr.log 2>&1
done
python - <<PY
import csv, collections, json, re
res = collections.defaultdict(dict)
for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("$R/gpurun_out/pmc_%s/p_counter_collection.csv" % ctr)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m and "pfm" in r["Kernel_Name"]:
            acc[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        scale = 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
        res[k][ctr.lower().replace("_size", "_bytes")] = sum(v) / len(v) * scale
        res[k]["launches"] = len(v)
json.dump({"args": "$*", "per_launch": res}, open("$R/$out", "w"), indent=1)
for k, d in res.items():
    print(k, {a: ("%.3e" % b if isinstance(b, float) else b) for a, b in d.items()})
PY
