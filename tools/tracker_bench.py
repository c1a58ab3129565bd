"""Run only bench_extra.tracker_leg (the Tracker iteration, fused and reference-shaped) and print its JSON.  `python tools/tracker_bench.py`"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import attentive_dfprior_amd as A  # noqa: E402
from attentive_dfprior_amd import synthetic  # noqa: E402
import bench  # noqa: E402
import bench_extra as BX  # noqa: E402

dev = torch.device('cuda:0')
scene, sd, dec = bench.build_scene(A, synthetic, 'room0', dev)
print(json.dumps(BX.tracker_leg(A, synthetic, scene, sd, dec, dev), indent=1))
