#!/bin/bash
# usage: tools/profile_round.sh <tag>      (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r02')
# One pass over everything the round's DESIGN/VERDICT numbers come from; results under gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
HASH=$(python -c "import bench; print(bench.kernel_source_hash())")
python bench.py > $O/bench_216cube.json 2> $O/bench_216cube.err
python bench.py --residual-only --no-cpu-baseline --no-extras > $O/bench_216cube_residual_only.json 2>/dev/null
python bench.py --dim 2 --residual-only --no-cpu-baseline --no-extras > $O/bench_2d_1000sq_residual_only.json 2>/dev/null
python bench.py --dim 2 --no-cpu-baseline --no-extras > $O/bench_2d_1000sq_jacobian.json 2>/dev/null
PFM_RES_KERNEL=1 python bench.py --no-cpu-baseline --no-extras > $O/bench_216cube_residual_kernel.json 2>/dev/null
python bench.py --n 100 --path general --no-cpu-baseline --no-extras --steps 5 > $O/bench_100cube_general.json 2>/dev/null
python bench.py --dim 2 --path general --no-cpu-baseline --no-extras > $O/bench_2d_1000sq_general.json 2>/dev/null
python tools/bench_extra.py config5 --levels 8 --meshes 5 --world 1 --out $O/config5_miehe_amr.json > /dev/null 2>&1
(cd tools/microbench && [ -x ./lat ] && timeout 120 ./lat > $O/microbench_load_latency.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
run_prof () { # name, extra rocprof args..., -- bench args
  name=$1; shift
  rm -rf $O/$name
  timeout 600 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1
}
run_prof stats --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras
cp $O/stats/p_kernel_stats.csv $O/rocprofv3_kernel_stats_216cube.csv 2>/dev/null
run_prof stats_res --stats --output-format csv -d $O/stats_res -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --residual-only
cp $O/stats_res/p_kernel_stats.csv $O/rocprofv3_kernel_stats_216cube_residual_only.csv 2>/dev/null
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
# the 2-D kernels (BASELINE config 2 and the 2-D Jacobian): kernel durations and where their waves spend the cycles
run_prof stats_2d --stats --output-format csv -d $O/stats_2d -o p -- python $R/bench.py --dim 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras
cp $O/stats_2d/p_kernel_stats.csv $O/rocprofv3_kernel_stats_2d_1000sq_jacobian.csv 2>/dev/null
run_prof stats_2d_res --stats --output-format csv -d $O/stats_2d_res -o p -- python $R/bench.py --dim 2 --residual-only --steps 50 --warmup 5 --no-cpu-baseline --no-extras
cp $O/stats_2d_res/p_kernel_stats.csv $O/rocprofv3_kernel_stats_2d_1000sq_residual_only.csv 2>/dev/null
run_prof pmc_sq_2d --pmc $SQ --output-format csv -d $O/pmc_sq_2d -o p -- python $R/bench.py --dim 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras
cp $O/pmc_sq_2d/p_counter_collection.csv $O/rocprofv3_pmc_SQ_2d_1000sq_jacobian.csv 2>/dev/null
run_prof pmc_sq_2d_res --pmc $SQ --output-format csv -d $O/pmc_sq_2d_res -o p -- python $R/bench.py --dim 2 --residual-only --steps 3 --warmup 1 --no-cpu-baseline --no-extras
cp $O/pmc_sq_2d_res/p_counter_collection.csv $O/rocprofv3_pmc_SQ_2d_1000sq_residual_only.csv 2>/dev/null
rm -rf $O/stats_2d $O/stats_2d_res $O/pmc_sq_2d $O/pmc_sq_2d_res
# the 3-D overlay (refined block, 1.1e6 cells): kernel durations
cat > /tmp/ov3run.py <<PYX
import sys, torch
sys.path.insert(0, "$R")
import bench
bench.overlay_3d(torch.device("cuda:0"), 0, 6, 84)
PYX
run_prof stats_ov3 --stats --output-format csv -d $O/stats_ov3 -o p -- python /tmp/ov3run.py
cp $O/stats_ov3/p_kernel_stats.csv $O/rocprofv3_kernel_stats_overlay3d.csv 2>/dev/null
rm -rf $O/stats_ov3
run_prof pmc_sq --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
cp $O/pmc_sq/p_counter_collection.csv $O/rocprofv3_pmc_SQ_216cube.csv 2>/dev/null
run_prof pmc_lds --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/pmc_lds -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
cp $O/pmc_lds/p_counter_collection.csv $O/rocprofv3_pmc_LDS_216cube.csv 2>/dev/null
for ctr in WRITE_SIZE FETCH_SIZE; do
  run_prof pmc_$ctr --pmc $ctr --output-format csv -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
  cp $O/pmc_$ctr/p_counter_collection.csv $O/rocprofv3_pmc_${ctr}_216cube.csv 2>/dev/null
done
MIX="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS"
mix () { # tag, bench args...
  tag=$1; shift
  run_prof pmc_mix_$tag --pmc $MIX --output-format csv -d $O/pmc_mix_$tag -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@"
  cp $O/pmc_mix_$tag/p_counter_collection.csv $O/rocprofv3_pmc_MIX_$tag.csv 2>/dev/null
  rm -rf $O/pmc_mix_$tag
}
mix 3d_216
mix 3d_216_residual --residual-only
mix 2d_1000 --dim 2
mix 2d_1000_residual --dim 2 --residual-only
# config-5 stand-in (general family + cartesian overlay): per-kernel times and instruction mix of one mesh of the sequence
run_prof stats_c5 --stats --output-format csv -d $O/stats_c5 -o p -- python $R/tools/bench_extra.py config5 --levels 8 --meshes 1 --world 1
cp $O/stats_c5/p_kernel_stats.csv $O/rocprofv3_kernel_stats_config5.csv 2>/dev/null
run_prof pmc_mix_c5 --pmc $MIX --output-format csv -d $O/pmc_mix_c5 -o p -- python $R/tools/bench_extra.py config5 --levels 8 --meshes 1 --world 1
cp $O/pmc_mix_c5/p_counter_collection.csv $O/rocprofv3_pmc_MIX_config5.csv 2>/dev/null
run_prof pmc_sq_c5 --pmc $SQ --output-format csv -d $O/pmc_sq_c5 -o p -- python $R/tools/bench_extra.py config5 --levels 8 --meshes 1 --world 1
cp $O/pmc_sq_c5/p_counter_collection.csv $O/rocprofv3_pmc_SQ_config5.csv 2>/dev/null
rm -rf $O/stats_c5 $O/pmc_mix_c5 $O/pmc_sq_c5
# general family in 3-D (10^6 hexes forced onto it): time, instruction mix and line traffic per colour-class launch
(cd $R && bash tools/pmc_any.sh $O/general_3d -- python $R/bench.py --n 100 --path general --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 | cut -c1-240 | grep k_assemble_general > $O/general_3d_100cube_kernels.txt
 bash tools/hbm_traffic.sh gpurun_out/$TAG/general_3d_100cube_traffic.json -- --n 100 --path general --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 | grep k_assemble_general >> $O/general_3d_100cube_kernels.txt
 python bench.py --n 100 --path general --residual-only --no-cpu-baseline --no-extras --steps 5 > $O/bench_100cube_general_residual_only.json 2>/dev/null
 rm -rf $O/general_3d gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_FETCH_SIZE)
cd /tmp
python - <<PY
import csv, collections, json, re
O = "$O"
HASH = "$HASH"
def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z0-9_]+)(<[0-9]+>)?", r["Kernel_Name"])  # (k_cart2d_cells<0> and <1>: two launches per assembly)
        if m and "pfm" in r["Kernel_Name"]:
            acc[m.group(1) + (m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    # the 3-D residual kernels: the run's set-up (all three vectors through the state) launches another variant once --
    # not a part of the assemblies the record describes
    r3 = {k: max(len(v) for v in acc[k].values()) for k in acc if k.startswith("k_cart_residual3")}
    for k, n in r3.items():
        if n < max(r3.values()):
            del out[k]
    return out
# HBM traffic (FETCH_SIZE doubled on gfx950, KiB units: MI355X_MICROARCH.md, HBM section)
res = collections.defaultdict(dict)
for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
    for k, d in per_kernel(f"{O}/rocprofv3_pmc_{ctr}_216cube.csv").items():
        res[k][ctr.lower().replace("_size", "_bytes")] = d[ctr] * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
keep = ("k_cart_residual3", "k_cart_residual3d", "k_cart_residual3x", "k_cart_uu3", "k_cart_phi4", "k_state_set")
json.dump({"args": "--steps 3 --warmup 1 --no-cpu-baseline --no-extras", "kernel_source_hash": HASH, "per_launch": {k: v for k, v in res.items() if k in keep}},
          open(f"{O}/hbm_traffic_3d_216.json", "w"), indent=1)
for tag in ("3d_216", "3d_216_residual", "2d_1000", "2d_1000_residual"):
    try:
        mixd = per_kernel(f"{O}/rocprofv3_pmc_MIX_{tag}.csv")
    except OSError:
        continue
    for k, d in mixd.items():
        f64 = d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0) + d.get("SQ_INSTS_VALU_FMA_F64", 0)
        d["fp64_share_of_valu"] = f64 / max(d.get("SQ_INSTS_VALU", 1), 1)
        print("mix", tag, k, "VALU %.3e FP64 %.3e share %.3f" % (d.get("SQ_INSTS_VALU", 0), f64, d["fp64_share_of_valu"]))
    json.dump({"counters": "$MIX (wave-instructions per launch, mean over the launches of the run)", "kernel_source_hash": HASH,
               "per_launch": {k: v for k, v in mixd.items() if k != "k_aos_to_soa"}}, open(f"{O}/instruction_mix_{tag}.json", "w"), indent=1)
sq = per_kernel(f"{O}/rocprofv3_pmc_SQ_216cube.csv")
json.dump({"counter": "SQ_INSTS_VALU (wave-instructions per launch)", "per_launch": {k: v["SQ_INSTS_VALU"] for k, v in sq.items() if k in keep and k != "k_state_set"}},
          open(f"{O}/valu_instructions_3d_216.json", "w"), indent=1)
for k, d in sq.items():
    if k in keep:
        wc = d["SQ_WAVE_CYCLES"]
        print(k, "VALU/wave-cycle %.3f  wait %.3f  issue-stall %.3f  active-any %.3f  INSTS_VALU %.3e" % (
            d["SQ_ACTIVE_INST_VALU"] / wc, d["SQ_WAIT_ANY"] / wc, d["SQ_WAIT_INST_ANY"] / wc, d["SQ_ACTIVE_INST_ANY"] / wc, d["SQ_INSTS_VALU"]))
for k, d in res.items():
    if k in keep:
        print(k, {a: "%.3e" % b for a, b in d.items()})
PY
rm -rf $O/stats $O/stats_res $O/pmc_sq $O/pmc_lds $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE
# the headline line once more, now that the counter summaries of THESE kernel sources exist (bench.py quotes the newest
# profiles/r*/ summary whose source hash matches; the first run above could only see the previous ones)
mkdir -p $R/profiles/$TAG
cp $O/hbm_traffic_3d_216.json $O/instruction_mix_*.json $R/profiles/$TAG/
cd $R && python bench.py > $O/bench_216cube.json 2> $O/bench_216cube.err
python bench.py --residual-only --no-cpu-baseline --no-extras > $O/bench_216cube_residual_only.json 2>/dev/null
python bench.py --dim 2 --residual-only --no-cpu-baseline --no-extras > $O/bench_2d_1000sq_residual_only.json 2>/dev/null
python bench.py --dim 2 --no-cpu-baseline --no-extras > $O/bench_2d_1000sq_jacobian.json 2>/dev/null
cd /tmp
grep -h "k_cart\|k_state" $O/rocprofv3_kernel_stats_216cube.csv | cut -c1-60,150-260
python -c "
import json
for f in ('bench_216cube','bench_216cube_residual_only','bench_2d_1000sq_residual_only','bench_2d_1000sq_jacobian','bench_216cube_residual_kernel','bench_100cube_general','bench_2d_1000sq_general'):
    try:
        d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, 'ms/step %.3f kernel_ms %.3f median %.3f frac %.4f value %.3e ctx_create %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['frac'], d['value'], d['config'].get('ctx_create_s')))
    except Exception as e: print(f, 'FAILED', e)
"
# round 5: robustness runs, context phase clocks, the bitwise mode of the 3-D hanging-node cells
cd $R
python tools/stress.py 216 > $O/stress_216.txt 2>&1
python tools/stress.py 64 general > $O/stress_64_general.txt 2>&1
python tools/stress.py 24 hanging > $O/stress_24_hanging.txt 2>&1
PFM_HANGING_COLOURED=1 python tools/stress.py 24 hanging > $O/stress_24_hanging_coloured.txt 2>&1
PFM_HANGING_ATOMIC=1 python tools/stress.py 24 hanging > $O/stress_24_hanging_atomic.txt 2>&1
PFM_CTX_TIMING=1 python tools/ctx_timing.py c5 > $O/ctx_timing_config5.txt 2>&1
PFM_CTX_TIMING=1 python tools/ctx_timing.py 216 > $O/ctx_timing_216cube.txt 2>&1
# round 6: the three modes of the 3-D cells at hanging vertices on the overlay bench mesh, phase clocks of the Jacobian pair
for e in "A=default" "PFM_HANGING_ATOMIC=1" "PFM_HANGING_COLOURED=1"; do echo "$e $(env $e python tools/ov3_time.py 10 2>/dev/null | tail -1)"; done > $O/overlay3d_hanging_modes.txt
bash tools/kst.sh gpurun_out/$TAG/overlay3d_kernel_stats_sequential.txt PFM_GENERAL_SEQUENTIAL=1 -- python $R/tools/overlay3d_profile.py
cd $R
for m in 1 2; do PFM_PHI_CLK=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "phase clock" | tail -1; done > $O/phase_clock_k_cart_phi4.txt
PFM_UU_CLK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "phase clock" | tail -3 > $O/phase_clock_k_cart_uu3.txt
bash tools/ab.sh "A=pair" "PFM_JAC_SEQUENTIAL=1" "A=pair" "PFM_JAC_SEQUENTIAL=1" > $O/ab_pair_vs_sequential.txt 2>&1
(cd tools/microbench && [ -x ./fill ] && timeout 300 ./fill 32 > $O/microbench_fill_policies.txt 2>&1)
