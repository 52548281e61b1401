"""Development probe: does BASELINE configs[4]'s head leg of bench.py depend on what ran before it in the same process?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
timer = B.Timer(None, dev)
prof = B.Prof(0)
which = sys.argv[1]
if "head" in which:
    net, sd, eot, hc = B.build_net("f32", dev, 512)
    B.head_legs(net, dev, None, 0, 1, 0, 4, 2, timer)
if "peaks" in which:
    B.peaks_measured(dev, 0)
if "hbm" in which:
    B.hbm_kernel_legs(dev, 5.0)
out = B.config4_leg(dev, timer, prof, 1, 6)
print(which, json.dumps(out["head"]))
