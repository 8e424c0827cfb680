"""Dense, 1,024 poses per step through the device-output call (bench.py's other_models step), with mi_gnina options
from the command line: python tools/experiments/dense_throughput.py [NAME=V ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
B, NL, NR = 1024, 24, 2400
m = capi.Model("dense")
sc = capi.Scorer([m])
rng = np.random.RandomState(0)
rx, rs = synth.make_receptor(rng, NR, synth.mapped_types(m.chan_of_smt(False)))
lx, ls = synth.make_ligand(rng, NL, synth.mapped_types(m.chan_of_smt(True)))
poses = synth.make_poses(np.random.RandomState(1000), lx, B)
sc.set_receptor(rx, rs)
d_lig = torch.from_numpy(poses).to("cuda:0")
d_o = torch.empty(4, B, dtype=torch.float32, device="cuda:0")


def step():
    sc.score_batch_device(d_lig.data_ptr(), ls, B, NL, d_o[0].data_ptr(), d_o[1].data_ptr(), d_o[2].data_ptr(), d_o[3].data_ptr())


t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    step()
    sc.synchronize()
blocks = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    sc.synchronize()
    blocks.append(B * 8 / (time.perf_counter() - t0))
print(f"{sys.argv[1:]} dense: {sorted(blocks)[1]:.0f} poses/s (blocks {[round(b) for b in blocks]}) checksum {float(d_o[0].double().sum()):.9f}")
