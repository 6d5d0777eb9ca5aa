"""Jacobian / residual-only time of bench.py's 3-D overlay mesh alone (A/B runs of the hanging-node modes):
python tools/ov3_time.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.overlay_3d(dev, 0, steps)
print(json.dumps({k: (round(v["ms_per_call"], 3) if isinstance(v, dict) else v) for k, v in r.items() if k.startswith(("overlay_", "general_", "ctx_create", "regular_row_f"))}))
