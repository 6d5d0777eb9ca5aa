cd /root/repo
timeout 600 python -m pytest tests/test_gpu_split_corners.py tests/test_gpu_cart.py tests/test_glue_mock.py tests/test_gpu_parity.py -x -q > gpurun_out/t_tan.log 2>&1; tail -4 gpurun_out/t_tan.log
for e in "PFM_NO_PATCH=1" "PFM_X=1"; do echo "== $e"; env $e timeout 200 python tools/bench_extra.py config5 --levels 8 --meshes 2 --world 1 2>&1 | tail -1 | cut -c1-420; done
cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c5stat -o p -- python /root/repo/tools/bench_extra.py config5 --levels 8 --meshes 1 --world 1 > /dev/null 2>&1; python - <<PY
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/c5stat/p_kernel_stats.csv')):
    if 'k_assemble' in r['Name'] or 'patch' in r['Name']:
        print(r['Name'].split('(pfm')[0][-60:], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
