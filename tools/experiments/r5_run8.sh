#!/bin/bash
# round 5, GPU call 8: (1) the DLScorer adapter test that missed the goldens: with lanes, without, and the driver's own lines;
# (2) what the voxelizer's time is made of (MI_VOX_DBG timing switches: 1 hits not evaluated, 2 flushes empty, 4 no window
# stores, 8 no hits)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
for nl in "" 1; do
  echo "== test_host_adapter [MI_GNINA_NO_LANES=$nl]"
  MI_GNINA_NO_LANES=$nl timeout 600 python -m pytest tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -12
done
python - <<'PY'
import os, struct, subprocess, numpy as np
G = np.load("tests/golden/cnn_goldens.npz")
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
base = names[0]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
with open("/tmp/atoms.bin", "wb") as f:
    f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), len(poses), 0))
    f.write(rec_xyz.astype("<f4").tobytes()); f.write(rec_smt.astype("<i4").tobytes()); f.write(lig_smt.astype("<i4").tobytes()); f.write(poses.astype("<f4").tobytes())
want = np.mean([G[n + "/affinity"] for n in names], axis=0)
print("want affinity", want)
for env in ({}, {"MI_GNINA_NO_LANES": "1"}, {"MI_GNINA_D16_PERSIST": "0", "MI_GNINA_K1S_PERSIST": "0"}, {"MI_GNINA_NO_DENSE_SPLIT": "1"}):
    e = dict(os.environ); e.update(env)
    for rep in range(2):
        r = subprocess.run(["gnina_amd/lib/test_host_scorer", "/tmp/atoms.bin", "gnina_amd/weights"], capture_output=True, text=True, env=e)
        lines = [l for l in r.stdout.split("\n") if l.startswith(("single", "batch", "grad"))]
        print(env, rep, "rc", r.returncode)
        for l in lines: print("   ", l)
PY
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
for k in d.get('kernels', []):
    if 'vox' in k['kernel'] or 'gather' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for dbg in 0 1 2 4 8 3 7 6; do
  echo "== voxelizer MI_VOX_DBG=$dbg"
  MI_VOX_DBG=$dbg timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
done
