"""How long does the HOST take to enqueue a B = 1 call of gnina's default ensemble, against the whole call?  Device-output
calls return when they are enqueued: time to return, then time to the stream's synchronisation.
python tools/experiments/b1_enqueue_time.py [NAME=V ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
lig_smt = np.ascontiguousarray(lig_smt, np.int32)
s = capi.Scorer(names)
s.set_receptor(rec_xyz, rec_smt)
for _ in range(60):  # (host-output calls first: they tune the lane streams)
    s.score_batch(poses[:1], lig_smt)
d_lig = torch.from_numpy(np.ascontiguousarray(poses[:1])).to("cuda:0")
d_o = torch.empty(4, 1, dtype=torch.float32, device="cuda:0")
L = poses.shape[1]
enq, tot = [], []
for rep in range(200):
    t0 = time.perf_counter()
    s.score_batch_device(d_lig.data_ptr(), lig_smt, 1, L, d_o[0].data_ptr(), d_o[1].data_ptr(), d_o[2].data_ptr(), d_o[3].data_ptr())
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0)
    tot.append(t2 - t0)
print(f"{sys.argv[1:]} device-output B = 1 call of the default ensemble: enqueued after {np.median(enq) * 1e6:.0f} us (min {np.min(enq) * 1e6:.0f}), "
      f"done after {np.median(tot) * 1e6:.0f} us (min {np.min(tot) * 1e6:.0f})")
