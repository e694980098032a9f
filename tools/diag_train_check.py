"""Does the in-run check of bench.train_measure slow the timed steps down?  python tools/diag_train_check.py <0|1>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
chk = bool(int(sys.argv[1]))
for prec in ("f32", "bf16", "f32"):
    r = bench.train_measure(0, 1, dev, False, precision=prec, steps=5, warmup=2, roofline=False, check=chk)
    print(prec, "check", chk, "ms_per_step", r.get("ms_per_step"), r.get("check", {}).get("rel_diff"), flush=True)
