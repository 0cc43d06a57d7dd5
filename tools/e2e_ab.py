#!/usr/bin/env python3
"""Same-process A/B of bench.py's e2e leg: resident batches vs host->device copy inside the timed region, alternated
(the e2e leg of a default bench run comes last, on a chip two timed modes have warmed).  Prints ms/step per leg."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--precision", default="f32")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
a.extended, a.attention = False, False
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for r in range(a.rounds):
    for h2d in (False, True):
        res = bench.run_mode(a.precision, a, dev, 0, 1, None, 128, 200, 200, h2d=h2d, light=True)
        print("round %d  %-8s %.4f ms/step" % (r, "h2d" if h2d else "resident", res["elapsed"] / a.steps * 1e3), flush=True)
