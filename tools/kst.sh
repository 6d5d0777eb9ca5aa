#!/bin/bash
# usage: tools/kst.sh <out.txt> [ENV=..] -- <python script + args>   kernel durations (rocprofv3 --kernel-trace --stats) of any command
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
mkdir -p $(dirname $R/$out); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kst
env "${envs[@]}" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kst -o p -- "$@" > /tmp/prof_kst.log 2>&1
python3 - > $R/$out <<'PY'
import csv,re
rows=list(csv.DictReader(open("/tmp/prof_kst/p_kernel_stats.csv")))
for r in rows[:30]:
    n=r["Name"]; m=re.search(r"(k_[a-z0-9_]+)",n)
    t=n[n.find("<"):n.find(">")+1] if "<" in n else ""
    print(f'{(m.group(1) if m else n[:30]):28s} {t[:44]:44s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]}')
PY
