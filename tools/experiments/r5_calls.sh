#!/bin/bash
# Round 5's GPU calls, one function per call (they were 57 separate three-to-thirty-line scripts; DESIGN / LAB.md quote them as
# "r5_run<N>.sh").  usage on the GPU box:  bash tools/experiments/r5_calls.sh <N> [<N> ...]   |   r5_calls.sh list
# Each call keeps its question in the comment above its function.
R=${GRAFT_REPO_ROOT:-$(pwd)}

# round 5, GPU call 1: Dense on split tensors (first run) + the voxelizer's conflict-free transpose
call_1() {
cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense split tests"
timeout 900 python -m pytest tests/test_gpu_dense_split.py -x -q -s 2>&1 | tail -40
echo "== voxelizer parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_h2.py -x -q 2>&1 | tail -8
echo "== bench dense (split)"
timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench dense (round-4 kernels: MI_GNINA_NO_DENSE_SPLIT)"
MI_GNINA_NO_DENSE_SPLIT=1 timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench dense NP=1"
MI_GNINA_D16_NP=1 timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench default2017"
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
}

# round 5, GPU call 2: what bounds conv3d_h2_d16_kernel -- timing-only switches (MI_GNINA_H2_DBG bits: 2 no K loop, 4 no
# tile DMA, 8 no weight DMA, 32 contiguous tile sources) and cache counters
call_2() {
cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'sp_h2' in k['kernel'] or 'vox' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for dbg in 0 2 4 8 32 34 12 14; do
  echo "== dense, MI_GNINA_H2_DBG=$dbg"
  MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
export TMPDIR=/tmp
OUT=$R/gpurun_out/r5/pmc_dense; mkdir -p $OUT; cd /tmp
BENCH="python $R/bench.py --model dense --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -f csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/c -o p -- $BENCH > $OUT/c.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUSY_avr --kernel-trace -f csv -d $OUT/d -o p -- $BENCH > $OUT/d.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0]
        if 'd16' in k or 'k1s' in k or 'conv3d_h2_kernel' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-34s mean %.4g  n %d  min %.4g max %.4g' % (c, sum(x)/len(x), len(x), min(x), max(x)))
PY
tail -2 $OUT/a.log $OUT/d.log | cut -c1-300
}

# round 5, GPU call 3: cheap probes -- poses per internal chunk (MALL residency of the activations), the d16 epilogue,
# tiles of the 1x1x1 transitions (z-run length of the DMA sources)
call_3() {
cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
blk=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's24' in k['kernel'] and 'to16_sp' in k['kernel'])
print('   block0 (24^3 d16 layers) %.3f ms' % blk)
for k in d.get('kernels', []):
    if 'conv1' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for ch in 32 64 128 256 512; do
  echo "== dense --chunk $ch"
  timeout 300 python bench.py --model dense --chunk $ch --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
for dbg in 64 78; do
  echo "== dense, MI_GNINA_H2_DBG=$dbg"
  MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
for t in 1,1,12 1,2,6 2,1,6 1,4,4 2,2,3 4,2,2; do
  echo "== dense, MI_GNINA_K1_TILE=$t"
  MI_GNINA_K1_TILE=$t timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
}

# round 5, GPU call 4: the whole -m gpu suite on the current tree (lean push: the 57 build-time weight blobs stay behind,
# their per-model tests skip), k1s probes, C5 entry
call_4() {
cd $R
mkdir -p gpurun_out/r5
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'conv1' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for dbg in 0 2 4 6 64 70; do
  echo "== dense, MI_GNINA_K1S_DBG=$dbg"
  MI_GNINA_K1S_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
echo "== C5"
timeout 900 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from gnina_amd import capi, synth
capi.init(0)
print(json.dumps(bench.config_c5(capi, synth), indent=1, default=float))
PY
}

# round 5, GPU call 5: persistent d16 / k1s launches, lanes for small ensemble calls, tests touched by the Dense work
call_5() {
cd $R
mkdir -p gpurun_out/r5
echo "== pytest (dense split, refine, gradient, h2, parity, host adapter)"
timeout 1500 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_cnn_refine.py tests/test_gpu_gradient.py tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_host_adapter.py tests/test_gpu_h2_range.py -m gpu -q 2>&1 | tail -15
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
blk=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's24' in k['kernel'] and 'to16_sp' in k['kernel'])
blk1=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's12' in k['kernel'] and 'to16_sp' in k['kernel'])
print('   block0 (24^3 d16 layers) %.3f ms   block1 (12^3) %.3f ms' % (blk, blk1))
for k in d.get('kernels', []):
    if 'conv1' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense (persistent, defaults)"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_PERSIST=0 MI_GNINA_K1S_PERSIST=0"
MI_GNINA_D16_PERSIST=0 MI_GNINA_K1S_PERSIST=0 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_PERSIST=3"
MI_GNINA_D16_PERSIST=3 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_NP=1"
MI_GNINA_D16_NP=1 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== seam B=1 (lanes) / MI_GNINA_NO_LANES=1"
for nl in "" 1; do
MI_GNINA_NO_LANES=$nl timeout 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from gnina_amd import capi, synth
capi.init(0)
print(os.environ.get("MI_GNINA_NO_LANES"), json.dumps(bench.config_seam_b1(capi, synth), default=float))
PY
done
}

# round 5, GPU call 6 (second session): the state of the tree -- whole -m gpu suite, the full bench line, kernel stats + PMC
# passes of the headline and of the Dense model
call_6() {
cd $R
mkdir -p gpurun_out/r5
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_full.json 2> gpurun_out/r5/bench_full.err; tail -c 400 gpurun_out/r5/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_full.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
print(json.dumps(d.get('also'), default=float)[:6000])
print(json.dumps(d.get('cpu_baseline'), default=float))
PY
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null > gpurun_out/r5/bench_dense.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_dense.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
PY
echo "== profiles"
bash tools/profile_gpu.sh r5a > gpurun_out/r5/prof_r5a.log 2>&1
bash tools/profile_gpu.sh r5a_dense --model dense > gpurun_out/r5/prof_r5a_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5a > gpurun_out/r5/pmc_r5a.txt
python tools/pmc_summary.py gpurun_out/prof_r5a_dense > gpurun_out/r5/pmc_r5a_dense.txt
ls gpurun_out/prof_r5a/trace
}

# round 5, GPU call 7: (1) which part of a small ensemble call misses the goldens (lanes?), (2) persistent conv3d_h2_kernel
# (first convolutions): parity tests, then A/B timings -- MI_GNINA_H2_PERSIST=0 (one workgroup per item, the round-4 kernel),
# default (persistent + early DMA), MI_GNINA_H2_EARLY=0, a cap of 2 workgroups per CU
call_7() {
cd $R
mkdir -p gpurun_out/r5
echo "== lanes diagnostic"
timeout 600 python tools/experiments/lanes_diag.py 2>&1 | tail -20
echo "== parity tests on the persistent kernel"
timeout 1200 python -m pytest tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_dense_split.py tests/test_gpu_h2_range.py -m gpu -x -q 2>&1 | tail -8
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'conv3_s24' in k['kernel'] or 'vox' in k['kernel'] or 's12_32' in k['kernel'] or 's6_64' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for env in "" "MI_GNINA_H2_PERSIST=0" "MI_GNINA_H2_EARLY=0" "MI_GNINA_H2_PERSIST=2" "MI_GNINA_H2_PERSIST=0 MI_GNINA_H2_DBG=2" "MI_GNINA_H2_DBG=2" "MI_GNINA_H2_DBG=6" "MI_GNINA_H2_PERSIST=0 MI_GNINA_H2_DBG=6"; do
  echo "== default2017 [$env]"
  env $env timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
for m in crossdock_default2018 dense; do
  for env in "" "MI_GNINA_H2_PERSIST=0"; do
    echo "== $m [$env]"
    env $env timeout 300 python bench.py --model $m --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
  done
done
}

# round 5, GPU call 8: (1) the DLScorer adapter test that missed the goldens: with lanes, without, and the driver's own lines;
# (2) what the voxelizer's time is made of (MI_VOX_DBG timing switches: 1 hits not evaluated, 2 flushes empty, 4 no window
# stores, 8 no hits)
call_8() {
cd $R
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
}

# round 5, GPU call 9: how often does a small ensemble call deviate, and under what: 16 runs of the DLScorer driver per setting
call_9() {
cd $R
python - <<'PY'
import os, struct, subprocess, numpy as np
G = np.load("tests/golden/cnn_goldens.npz")
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
base = names[0]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
with open("/tmp/atoms.bin", "wb") as f:
    f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), len(poses), 0))
    f.write(rec_xyz.astype("<f4").tobytes()); f.write(rec_smt.astype("<i4").tobytes()); f.write(lig_smt.astype("<i4").tobytes()); f.write(poses.astype("<f4").tobytes())
ref = None
for env in ({"MI_GNINA_NO_LANES": "1"}, {}, {"AMD_SERIALIZE_KERNEL": "3"}, {"MI_GNINA_NO_DENSE_SPLIT": "1"}, {"MI_GNINA_LANES_MAX_B": "1"}, {"MI_GNINA_CONV_PATH": "0"}, {"MI_GNINA_NO_LANES": "1"}):
    e = dict(os.environ); e.update(env)
    bad = 0; detail = []
    for rep in range(16):
        r = subprocess.run(["gnina_amd/lib/test_host_scorer", "/tmp/atoms.bin", "gnina_amd/weights"], capture_output=True, text=True, env=e)
        vals = []
        for l in r.stdout.split("\n"):
            t = l.split()
            if l.startswith("single"): vals += [float(t[2]), float(t[3]), float(t[7])]
            if l.startswith("batch"): vals += [float(t[2]), float(t[3])]
        vals = np.array(vals)
        if ref is None: ref = vals
        key = "ref" if env.get("MI_GNINA_NO_DENSE_SPLIT") or env.get("MI_GNINA_CONV_PATH") else None
        base_ = ref
        if key:  # different arithmetic: compare with this setting's own first run
            if rep == 0: own = vals
            base_ = own
        d = np.abs(vals - base_)
        if d.max() > 0: bad += 1; detail.append((rep, int(d.argmax()), float(d.max())))
    print(env, "runs that differ from the reference run:", bad, "of 16", detail[:6])
PY
}

call_10() {
cd $R
timeout 800 python tools/experiments/lanes_diag2.py 2>&1 | tail -30
}

call_11() {
cd $R
timeout 800 python tools/experiments/concurrency_diag.py 2>&1 | tail -30
}

# two scorers on two threads: which switch makes the deviations go away
call_12() {
cd $R
export DIAG_CALLS=150
for env in "" "MI_GNINA_NO_DENSE_SPLIT=1" "MI_GNINA_CONV_PATH=0" "MI_GNINA_NO_LAT=1" "MI_GNINA_D16_NP=1" "MI_GNINA_H2_NO_SPLIT_TENSORS=1" "MI_GNINA_H2_WLDS=0" "AMD_SERIALIZE_KERNEL=3" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

# two scorers on two threads: is it the runtime's handling of scratch (private segment) memory across hardware queues?
call_13() {
cd $R
export DIAG_CALLS=150
for env in "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_NO_SCRATCH_RECLAIM=1" "HSA_NO_SCRATCH_THREAD_LIMITER=1" "HSA_SCRATCH_SINGLE_LIMIT=0" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

# two scorers on two threads: activations in uncached device memory (is it a stale line in an XCD's L2?)
call_14() {
cd $R
export DIAG_CALLS=200
for env in "MI_DEVBUF_UNCACHED=1" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_15() {
cd $R
timeout 600 python tools/experiments/concurrency_diag2.py crossdock_default2018_KD_4 dense_1_3 2>&1 | tail -22
timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -22
MI_GNINA_CONV_PATH=0 timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -22
}

call_16() {
cd $R
export DIAG_CALLS=300
for env in "MI_VOX_DBG=16" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_17() {
cd $R
export DIAG_CALLS=250
MI_GNINA_H2_NO_SPLIT_TENSORS=1 timeout 600 python tools/experiments/concurrency_diag2.py crossdock_default2018_KD_4 dense_1_3 2>&1 | tail -16
MI_GNINA_H2_NO_SPLIT_TENSORS=1 timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -16
}

call_18() {
cd $R
timeout 600 python tools/experiments/alternate_diag.py 2>&1 | tail -5
}

call_19() {
cd $R
export DIAG_CALLS=200 DIAG_DETAIL=1
MI_GNINA_H2_NO_SPLIT_TENSORS=1 timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | grep -v "^ops\|amdgpu.ids" | cut -c1-330 | head -70
}

call_20() {
cd $R
export DIAG_CALLS=300
for env in "MI_VOX_DBG=64" "MI_VOX_DBG=96" "MI_VOX_DBG=112" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_21() {
cd $R
export DIAG_CALLS=300
for env in "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=0" "HIP_LAUNCH_BLOCKING=1" "MI_VOX_DBG=8"; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_22() {
cd $R
timeout 600 python tools/experiments/concurrency_diag3.py crossdock_default2018_KD_4 dense_1_3 2>&1 | grep -v amdgpu.ids | tail -30
}

call_23() {
cd $R
export DIAG_CALLS=400
for env in "MI_VOX_DBG=128" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_24() {
cd $R
export DIAG_CALLS=400
echo "== round-4 voxelize.hip"
timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
}

call_25() {
cd $R
echo "== concurrency test"
timeout 600 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -6
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
}

# persistent voxelizer wavefronts: parity, then A/B
call_26() {
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid96.py tests/test_gpu_h2.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'vox' in k['kernel'] or 'gather' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for env in "MI_VOX_PERSIST=0" "" "MI_VOX_PERSIST=16" "MI_VOX_PERSIST=24" "MI_VOX_PERSIST=64" "MI_VOX_PERSIST=128" "MI_VOX_DBG=7" "MI_VOX_DBG=8"; do
  echo "== default2017 [$env]"
  env $env timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
echo "== dense_1_3 @ 96^3 forward (C5 grid): voxelizer share"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
}

# poses per internal chunk: does the pooled grid (2.2 MB per pose) staying in the 256 MB MALL between the voxelizer and the first conv pay?
call_27() {
cd $R
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for ch in 0 512 256 128 96 64 32; do
  echo "== default2017 --chunk $ch"
  timeout 300 python bench.py --chunk $ch --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
for ch in 0 256 128 64; do
  echo "== crossdock_default2018 --chunk $ch"
  timeout 300 python bench.py --model crossdock_default2018 --chunk $ch --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern | head -1
done
}

# k1s without its spills (opaque thread index for the per-item set-up and epilogue)
call_28() {
cd $R
timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -4
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
}

# two scorers on two threads with NO LDS-DMA kernel anywhere (the fp32-MFMA program: MI_GNINA_CONV_PATH=f32)
call_29() {
cd $R
export DIAG_CALLS=400
cat > /tmp/nolock.py <<'PY'
PY
for env in "MI_GNINA_CONV_PATH=f32" ""; do
  echo "== [$env] (per-device lock bypassed: MI_GNINA_NO_CALL_LOCK=1)"
  env $env MI_GNINA_NO_CALL_LOCK=1 timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_30() {
cd $R
export DIAG_CALLS=400 MI_GNINA_NO_CALL_LOCK=1
for env in "MI_VOX_DBG=16" "MI_VOX_LDS_FRONT=2048" "MI_VOX_LDS_FRONT=8192" "MI_VOX_LDS_PAD=8"; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
}

call_31() {
cd $R
export DIAG_CALLS=400 MI_GNINA_NO_CALL_LOCK=1
echo "== hit records through vector loads"
timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
}

# the whole -m gpu suite with all 64 weight blobs on the box (the per-model parity tests against the reference's TorchScript outputs)
call_32() {
cd $R
ls gnina_amd/weights | wc -l
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
}

# round 5, final profiles: the full bench line, kernel stats + PMC passes of the headline and of the Dense model
call_33() {
cd $R
mkdir -p gpurun_out/r5
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; tail -c 300 gpurun_out/r5/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_final.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense','fp32_mfma_only'): print(k, a[k].get('poses_per_s'))
print('c3', a['c3'].get('mc_s'), a['c3'].get('strict_mode'))
print('c3_real', a['c3_real'].get('mc_s'), a['c3_real'].get('chains_bit_identical_to_reference'), a['c3_real']['cpu_baseline'].get('value'))
print('c4', a['c4'].get('ligands_per_s'))
print('c5', json.dumps(a['c5'], default=float)[:1500])
print('seam', json.dumps(a['seam_b1'], default=float)[-700:])
print('grad', json.dumps(a['gradient_calls'], default=float)[-900:])
print(json.dumps(d.get('cpu_baseline'), default=float))
PY
echo "== profiles"
bash tools/profile_gpu.sh r5f > gpurun_out/r5/prof_r5f.log 2>&1
bash tools/profile_gpu.sh r5f_dense --model dense > gpurun_out/r5/prof_r5f_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5f > gpurun_out/r5/pmc_r5f.txt
python tools/pmc_summary.py gpurun_out/prof_r5f_dense > gpurun_out/r5/pmc_r5f_dense.txt
ls gpurun_out/prof_r5f/trace
}

call_34() {
cd $R
timeout 900 python -m pytest tests/test_gpu_custom_model.py -m gpu -q 2>&1 | tail -40
}

call_35() {
cd $R
timeout 600 python tools/experiments/seam_b1_breakdown.py default2017 crossdock_default2018_KD_4 dense_1_3 2>&1 | grep -v amdgpu.ids | tail -12
}

call_36() {
cd $R
for kb in 52 100 150; do
  echo "== MI_GNINA_H2_LDS_KB=$kb"
  MI_GNINA_H2_LDS_KB=$kb timeout 600 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-1500
done
}

# planar split format ([octet][h | l][voxel][8 fp16]): the whole GPU suite, then the bench lines
call_37() {
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for rep in 1 2; do
echo "== default2017 ($rep)"
timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
echo "== crossdock"
timeout 300 python bench.py --model crossdock_default2018 --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern | head -5
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
for dbg in 2 6; do
echo "== default2017 MI_GNINA_H2_DBG=$dbg"
MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern | sed -n 4p
done
}

# final sanity on the final tree: smoke, the suite (lean push), the one-rank torchrun launch of the bench
call_38() {
cd $R
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torchrun N=1:', d['value'], d['ms_per_step'], d.get('rccl'))"
}

# d16: up to four octets per K chunk for small calls
call_39() {
cd $R
timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_concurrency.py tests/test_host_adapter.py tests/test_gpu_custom_model.py -m gpu -x -q 2>&1 | tail -4
for cc in 1 2 4; do
  echo "== B = 1, MI_GNINA_D16_CC=$cc"
  MI_GNINA_D16_CC=$cc timeout 300 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-1300
done
echo "== default (auto)"
timeout 300 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'sp_h2' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense throughput"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
}

call_40() {
cd $R
timeout 120 tools/microbench/sload_vs_ldsdma 8 2>&1 | tail -8
}

# round 5, GPU calls 43, 44: the voxelizer with a straight-line density_add (44: + timing switches compiled out, SGPR-offset s_loads) (the kernel issues more SALU than VALU instructions:
# 543 M against 468 M per launch -- the nested zone branches' exec-mask bookkeeping): bits (voxel parity tests), then its time
call_43() {
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "voxelize or typer or goldens or nearly_empty" 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('   headline %.0f %s, %.3f ms/step' % (d['value'], d['unit'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'vox' in k['kernel'] or 'gather' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for i in 1 2; do timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | kern; done
}

# round 5, GPU call 45: the voxelizer's instruction mix after the SALU trims (PMC passes of the short bench)
call_45() {
cd $R
OUT=$R/gpurun_out/prof_r5vox; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES --kernel-trace -f csv -d $OUT/pmc_a -o p -- $BENCH > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $OUT/pmc_b -o p -- $BENCH > $OUT/pmc_b.log 2>&1
tail -2 $OUT/pmc_a.log | cut -c1-300; tail -2 $OUT/pmc_b.log | cut -c1-300
cd $R; python tools/pmc_summary.py gpurun_out/prof_r5vox 2>&1 | awk '/^voxelize_tiles/{f=1} /^[a-z_]/{if(!/^voxelize_tiles/)f=0} f' | head -40
}

# round 5, GPU call 48: the device CNN Monte-Carlo chains under --accurate_line_search / --simple_ascent (VERDICT r4 missing #7)
call_48() {
cd $R
timeout 900 python -m pytest tests/test_gpu_cnn_refine.py -m gpu -x -q -k "metropolis" 2>&1 | tail -15
}

# round 5, GPU call 49: strict summation order for eval_intramolecular of a model with flexible residues (VERDICT r4 missing #7)
call_49() {
cd $R
timeout 900 python -m pytest tests/test_gpu_vina_ref.py -m gpu -x -q -k "flexible or strict or final" 2>&1 | tail -15
}

# round 5, GPU call 50: where the headline step's time goes between its kernels (kernel trace with timestamps of the timed loop)
call_50() {
cd $R
OUT=$R/gpurun_out/prof_r5gap; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $OUT/bench.log 2>&1
tail -c 400 $OUT/bench.log | head -c 300; echo
cd $R; python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r5gap/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the headline's timed loop: consecutive runs of gather -> voxelize -> conv x3 -> fc
names = [r['Kernel_Name'] for r in rows]
def short(n): return n.replace('void mig::','').replace('mig::','').split('(')[0][:40]
seq = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# find steps: index of gather_pose_atoms followed by voxelize_tiles<1, true>
steps = [i for i in range(len(seq)-6) if seq[i][0].startswith('gather_pose_atoms') and seq[i+1][0].startswith('voxelize_tiles<1, true>')]
print('steps found', len(steps))
import statistics
# take the steps 3..12 (after warmup): print per-kernel durations and the gap before each kernel
for k in range(4, min(len(steps)-1, 9)):
    i = steps[k]; j = steps[k+1]
    line = []
    for q in range(i, j):
        n, s, e = seq[q]
        gap = s - seq[q-1][2]
        line.append('%s gap %.1f dur %.1f' % (n[:22], gap/1e3, (e-s)/1e3))
    print('step', k, 'total %.1f us' % ((seq[j][1]-seq[i][1])/1e3))
    for l in line: print('    ', l)
PY
}

# round 5, GPU call 51: the headline with the clocks spun up before the warm-up steps (bench.py spin_up), against --spinup-seconds 0
call_51() {
cd $R
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('   headline %.0f %s, %.3f ms/step, spinup %s, roofline %.3f (%.3f ms)' % (d['value'], d['unit'], d['ms_per_step'], d['spinup']['steps'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))
a=d.get('also',{})
for k in ('crossdock_default2018','dense'): print('   ', k, a.get(k,{}).get('poses_per_s'), a.get(k,{}).get('blocks_poses_per_s'))
"; }
for sp in 0.3 0.3 0.3; do echo "== spinup $sp"; timeout 300 python bench.py --no-configs --no-cpu-baseline --spinup-seconds $sp 2>/dev/null | kern; done
}

# round 5, GPU call 53: lanes with every voxel group voxelized before the first lane starts -- bits and B = 1 latency,
# at the default number of hardware queues and at 8
call_53() {
cd $R
for q in "" 8; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/experiments/lanes_diag3.py 300 2>&1 | tail -4
done
}

# round 5, GPU call 55: lanes on by default -- the concurrency tests, the DLScorer adapter test, the ensemble tests of the
# parity suite, and the B = 1 seam numbers
call_55() {
cd $R
timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ensemble or ragged or chunking" 2>&1 | tail -3
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from gnina_amd import capi, synth
capi.init(0)
print(json.dumps(bench.config_seam_b1(capi, synth), indent=1)[-900:])
PY
}

# round 5, GPU call 57: the timeline of one B = 1 call of gnina's default ensemble with lanes (kernel trace, per queue)
call_57() {
cd $R
OUT=$R/gpurun_out/prof_r5b1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py > $OUT/log.txt 2>&1
grep "median call" $OUT/log.txt
cd $R; python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r5b1/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('void mig::','').replace('mig::','').split('(')[0][:34]
# the last call: from the last-but-one gather pair to the end
gi = [i for i, r in enumerate(rows) if 'gather_pose_atoms' in r['Kernel_Name']]
start = gi[-2]   # two groups per call: the last call starts at the second-to-last gather
t0 = int(rows[start]['Start_Timestamp'])
prev_end = {}
for r in rows[start:]:
    q = r['Queue_Id']; s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('q%-3s %-34s start %7.1f dur %6.1f' % (q, short(r['Kernel_Name']), (s_ - t0) / 1e3, (e_ - s_) / 1e3))
print('span of the call on the GPU: %.1f us' % ((max(int(r['End_Timestamp']) for r in rows[start:]) - t0) / 1e3))
PY
}

# round 5, GPU call 58: per-group ligand description caches + lanes enqueued round robin: tests, seam numbers, timeline
call_58() {
cd $R
timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_host_adapter.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ensemble or ragged or chunking or error" 2>&1 | tail -3
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from gnina_amd import capi, synth
capi.init(0)
d = bench.config_seam_b1(capi, synth)
print(json.dumps({k: d[k] for k in ('default2017', 'default_ensemble')}))
PY
bash tools/experiments/r5_run57.sh 2>&1 | grep -v "d16_kernel\|k1s\|h2_16_kernel" | tail -22
}

# round 5, GPU call 59: how the lanes are enqueued (A/B in one process)
call_59() {
cd $R
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
}

# round 5, GPU call 60: the global max pool with eight threads per channel: Dense parity, B = 1 latency
call_60() {
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_custom_model.py -m gpu -x -q -k "goldens or ensemble or custom or given_grids or batch_independence" 2>&1 | tail -3
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
}

# round 5, GPU call 61: conv3d_h2_16_kernel with its weights requested four steps ahead: Dense parity, B = 1 latency, Dense rate
call_61() {
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_custom_model.py tests/test_gpu_h2_range.py -m gpu -x -q -k "goldens or ensemble or custom or given_grids or batch_independence or range" 2>&1 | tail -3
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
timeout 300 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.0f, %.3f ms/step' % (d['value'], d['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense'): print('   ', k, a[k].get('poses_per_s'), a[k].get('blocks_poses_per_s'))
"
}

# round 5, GPU call 62: the whole -m gpu suite on the final library (weights of the seven committed models; the other 57
# blobs are skipped by their test when absent -- call 32 ran all 64)
call_62() {
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
}

# round 5, GPU call 63: final bench line + kernel stats + PMC passes of the headline and of the Dense model
call_63() {
cd $R
mkdir -p gpurun_out/r5
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_final2.json 2> gpurun_out/r5/bench_final2.err; tail -c 300 gpurun_out/r5/bench_final2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_final2.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'), d.get('spinup'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense','fp32_mfma_only'): print(k, a[k].get('poses_per_s'))
print('c3', a['c3'].get('mc_s'), a['c3'].get('strict_mode'))
print('c3_real', a['c3_real'].get('mc_s'), a['c3_real'].get('chains_bit_identical_to_reference'))
print('c4', a['c4'].get('ligands_per_s'))
print('c5', {k: (v.get('poses_per_s_forward'), v.get('poses_per_s_forward_backward')) for k, v in a['c5'].items() if isinstance(v, dict) and 'poses_per_s_forward' in v}, a['c5'].get('refine'))
print('seam', json.dumps({k: a['seam_b1'][k] for k in ('default2017', 'default_ensemble')}))
print('grad', json.dumps({k: v for k, v in a['gradient_calls'].items() if k != 'note'}, default=float))
print(json.dumps(d.get('cpu_baseline'), default=float)[:400])
PY
echo "== profiles"
bash tools/profile_gpu.sh r5g > gpurun_out/r5/prof_r5g.log 2>&1
bash tools/profile_gpu.sh r5g_dense --model dense > gpurun_out/r5/prof_r5g_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5g > gpurun_out/r5/pmc_r5g.txt
python tools/pmc_summary.py gpurun_out/prof_r5g_dense > gpurun_out/r5/pmc_r5g_dense.txt
ls gpurun_out/prof_r5g/trace
}

# round 5, GPU call 65: the bench line after the last edits of bench.py (short form)
call_65() {
cd $R
timeout 120 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.0f, %.3f ms/step, steps %d warmup %d' % (d['value'], d['ms_per_step'], d['steps'], d['warmup']))
print(json.dumps(d['voxelizer'])[:700])
print(json.dumps(d['roofline'])[:300])
"
}

# round 5, GPU call 66: after the device guard around the lane streams' creation: the ensemble test
call_66() {
cd $R
timeout 100 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q -k "ensemble" 2>&1 | tail -2
}

if [ "$1" = list ] || [ -z "$1" ]; then grep "^# round 5" "$0"; exit 0; fi
for n in "$@"; do ( call_$n ); done
