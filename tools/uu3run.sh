python -m pytest tests/test_gpu_cart.py -x -q -k "uu4_z_marching or full_matrix" 2>&1 | tail -4
b() { env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
b PFM_X=1
b PFM_RES_KERNEL=1
PFM_RES_KERNEL=1 PFM_UU_CLK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "phase clock" | tail -1
