#!/usr/bin/env python3
"""bench.py -- CNN-scored poses/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (atom gather -> voxelize -> CNN forward -> pose/affinity
scores) over one batch of synthetic poses (2,500-atom receptor, 32-atom
ligand, 1,024 rigid poses, 48^3 grid at 0.5 A) per GPU, inputs resident in HBM when the timed
region starts, real reference weights (gnina_amd/weights/*.mgw extracted from the reference's .pt).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model default2017] [--batch 1024]

N>1 is launched by the driver with torch.distributed.run (one rank per GPU); the path shards by
pose with no data-path collective (weak scaling: every rank scores its own 1,024-pose batch), so
the only collectives are the timing barrier and the max-over-ranks reduction.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The library asks the HIP runtime for 16 hardware queues per priority instead of 4 (mi_gnina_init; include/mi_gnina.h) -- which
# only counts before the process's FIRST HIP call, and in this process that call is torch's (device tensors, torch.distributed).
# So the variable is set here, as any host that initialises HIP before the library must (INTEGRATION.md "Threads"): with the
# default 4, gnina's default ensemble at B = 1 takes 736 instead of 592 us per call and four threads reach 1,805 instead of
# 3,361 poses/s (tools/experiments/seam_b1_in_bench.py).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (same guide)
PEAK_F16_MFMA_TFLOPS = 2500.0
DTYPE_NOTE = ("f32 (fp32 tensors and fp32 accumulation; forward convolutions on f16 MFMA with every operand split into two fp16 "
              "halves, products exact: scores <= 1e-5 from the fp32-MFMA kernels, see also.fp32_mfma_only)")
H2_SPLIT = 3.0                  # split-fp16 kernels (conv3d_h2.hip): f16 MFMA FLOPs executed per algorithmic (fp32-product) FLOP
PEAK_HBM_GBS = 8000.0
FLOP_PER_POSE = {"default2017": 1122895872, "crossdock_default2018": 998148096, "dense": 4541572416,
                 "dense_1_3@96": 36.33e9}   # SURVEY 8d / BASELINE.md section 2 (2 * MACs, unpadded)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="default2017")
    ap.add_argument("--batch", type=int, default=1024, help="poses per step per GPU")
    ap.add_argument("--chunk", type=int, default=0, help="poses per internal chunk (0 = engine default)")
    ap.add_argument("--n-rec", type=int, default=2500)
    ap.add_argument("--n-lig", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 entries of `also`")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--spinup-seconds", type=float, default=0.3,
                    help="untimed steps in front of the warm-up steps until the GPU's clocks are at their steady state")
    ap.add_argument("--only", default="", help="comma-separated keys of `also` to run (default: all): real_complex,c3,c3_real,c4,c5,seam_b1,gradient_calls")
    ap.add_argument("--extras-timeout", type=float, default=900.0,
                    help="seconds the sub-benchmarks behind the timed region may take before the headline line is printed without them")
    return ap.parse_args()


def spin_up(step, sync, seconds):
    """Untimed steps for `seconds`.  After an idle period the GPU needs ~30 ms of continuous work before its clocks are at
    their steady state: in a kernel trace of this bench the headline step takes 3.98, 3.66, 3.62, 3.54, 3.48, 3.46, 3.44,
    3.40 ms ... 3.40 ms after every pause (first conv 2.01 -> 1.71 ms), so three warm-up steps put the timed region on
    the ramp (tools/experiments/r5_calls.sh 50).  The W warm-up steps of the contract still follow."""
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        sync()
        n += 1
    return n


def cpu_baseline(args, blob_path, rec_xyz, rec_smt, lig_smt, poses, budget_s):
    """The oracle ("port" of the reference CPU path: libmolgrid-style voxelization in C, one
    thread like libmolgrid's CPU path, then the network with PyTorch CPU ops at B=1 on all host
    cores, which is what gnina does: torch::set_num_threads, main.cpp:1374) on a bounded sample."""
    import torch
    from oracle import cnn_ref, voxel
    blob = cnn_ref.Blob(blob_path)
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    ncpu = os.cpu_count() or 1

    def one(b):
        t0 = time.perf_counter()
        grid, _ = voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None, blob.resolution,
                                      blob.dimension, blob.radius_scaling)
        t1 = time.perf_counter()
        with torch.no_grad():
            p, a, l = cnn_ref.scores(blob, grid[None])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, float(p[0]), float(a[0])

    # gnina uses every hardware thread (torch::set_num_threads(settings.cpu), main.cpp:1374); at B=1
    # that oversubscribes big hosts, so be generous to the CPU: try a few thread counts, keep the best.
    best, cores = None, ncpu
    for t in sorted({ncpu, min(ncpu, 64), min(ncpu, 32), min(ncpu, 16), min(ncpu, 8)}, reverse=True):
        torch.set_num_threads(t)
        one(0)
        dt = min(sum(one(0)[:2]) for _ in range(2))
        if best is None or dt < best:
            best, cores = dt, t
    torch.set_num_threads(cores)
    one(0)  # warm-up
    tv = tc = 0.0
    n = 0
    scores = []
    t_start = time.perf_counter()
    while n < len(poses) and (time.perf_counter() - t_start) < budget_s:
        a, b, p, af = one(n)
        tv += a
        tc += b
        scores.append((p, af))
        n += 1
    # BASELINE.md section 4.2 also asks for the all-core variant of the voxelization stage: the same C oracle, one pose per
    # host thread (ctypes releases the GIL), as many poses in flight as there are cores -- a throughput libmolgrid's CPU
    # path does not have (it voxelizes one pose on one thread), reported beside the faithful figure, not inside `value`
    vox_all = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        m = min(len(poses), 2 * ncpu)

        def vox_only(b):
            voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None, blob.resolution, blob.dimension,
                                blob.radius_scaling)
        with ThreadPoolExecutor(max_workers=ncpu) as ex:
            list(ex.map(vox_only, range(min(m, ncpu))))      # warm-up
            t0 = time.perf_counter()
            list(ex.map(vox_only, range(m)))
            dt_all = time.perf_counter() - t0
        vox_all = {"threads": ncpu, "poses": m, "poses_per_s": round(m / dt_all, 1),
                   "note": "C oracle voxelizer, one pose per thread, all host cores (BASELINE.md 4.2's OpenMP-style variant)"}
    except Exception as e:
        vox_all = {"error": f"{type(e).__name__}: {e}"}
    return {"value": n / (tv + tc), "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": f"{n} poses of the same workload, B=1 per call like the reference "
                      f"(torch_model.cpp:179); voxelize {1e3 * tv / n:.1f} ms/pose (C oracle, 1 thread) + "
                      f"CNN {1e3 * tc / n:.1f} ms/pose (PyTorch CPU fp32, {cores} threads)",
            "voxelize_all_cores": vox_all}, scores


def pmc_entry(kernel_label):
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        pm = json.load(open(path))
        want = pm.get("roofline_kernel_map", {}).get(kernel_label)
        return pm["kernels"].get(want) if want else None
    except Exception:
        return None


def voxelizer_issue_committed(kernel_prefix="voxelize_tiles<1, true"):
    """The tile kernel's bound is the vector instruction issue rate (one VALU instruction per SIMD every four cycles), not
    HBM: SQ_INSTS_VALU x 4 cycles / 1,024 SIMDs against the kernel's own GPU cycles (GRBM_GUI_ACTIVE counts all eight XCDs),
    both from the COMMITTED rocprofv3 passes (profiles/latest_pmc.json) -- counters only, no clock assumed."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        ks = json.load(open(path))["kernels"]
        k = next(v for n, v in ks.items() if n.startswith(kernel_prefix))
        valu, salu, cyc = k["SQ_INSTS_VALU"], k["SQ_INSTS_SALU"], k["GRBM_GUI_ACTIVE"] / 8.0
        return {"valu_insts_per_launch": round(valu), "salu_insts_per_launch": round(salu),
                "gpu_cycles_per_launch": round(cyc), "frac": round(valu * 4.0 / 1024.0 / cyc, 4),
                "scalar_alu_frac": round(salu / 256.0 / cyc, 4),
                "note": "VALU instructions x 4 cycles / 1,024 SIMDs over the kernel's cycles; scalar: SALU instructions / 256 "
                        "CUs (one scalar ALU per CU) over the same cycles; profiles/latest_pmc.json (committed), not this run; instruction and "
                        "cycle counters come from SEPARATE rocprofv3 passes, so the ratio carries their run-to-run spread (a few per cent: "
                        "a value just above 1 means 'at the issue rate')"}
    except Exception:
        return None


def pmc_traffic_committed(kernel_label):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes -- not measured in this run
    (profiles/latest_pmc.json, written by tools/pmc_summary.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command; units/corrections per MI355X_MICROARCH.md).  bench.py
    cannot collect PMCs itself; null when no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    if not os.path.exists(path):
        return None
    try:
        pm = json.load(open(path))
        want = pm.get("roofline_kernel_map", {}).get(kernel_label)
        k = pm["kernels"].get(want) if want else None
        return k.get("hbm_bytes_per_launch") if k else None
    except Exception:
        return None


def roofline_block(dom_conv, avg_ms, exec_flops, exec_src):
    """`achieved` / `frac` = the MFMA FLOPs the kernel EXECUTES per launch / its launch time, against the fp32-MFMA peak:
    a roofline fraction, never above 1.  The zero-skipping conv executes about a third of the layer's algorithmic
    (dense) FLOPs -- the rest are products with exactly-zero operands it never issues -- so the layer's
    algorithmic-equivalent rate is reported beside it under its own name (it may exceed the pipe's peak; it is not a
    fraction of anything the hardware did), and `dense_no_skip` times the same kernel with skipping off."""
    algo = dom_conv["flops"] / dom_conv["launches"]
    algo_tf = algo / (avg_ms * 1e-3) / 1e12
    h2 = is_h2(dom_conv["kernel"])
    peak = PEAK_F16_MFMA_TFLOPS if h2 else PEAK_FP32_MFMA_TFLOPS
    ex = exec_flops if exec_flops else algo * (H2_SPLIT if h2 else 1.0)   # a kernel without counters executes everything
    ex_tf = ex / (avg_ms * 1e-3) / 1e12
    return {
        "bound": "mfma",
        "kernel": dom_conv["kernel"] + (" (conv3d_h2_kernel)" if h2 else " (conv3d_mfma_kernel)"),
        "achieved": round(ex_tf, 2),
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(ex_tf / peak, 4),
        "achieved_is": ("executed f16-MFMA FLOPs (split-fp16 path: three v_mfma_f32_32x32x16_f16 of 32,768 FLOPs per K = 16 "
                        "of fp32 products) / HIP-event launch time; peak = dense f16 MFMA") if h2 else
                       "executed MFMA FLOPs (32x32x2 fp32: 4,096 per instruction) / HIP-event launch time",
        "executed_flops_per_launch": ex,
        "executed_source": exec_src if exec_flops else "no device counters: executed == algorithmic" + (" x 3" if h2 else ""),
        "traffic": None,
        "traffic_from_committed_pmc": pmc_traffic_committed(dom_conv["kernel"]),
        "traffic_note": "HBM bytes per launch from profiles/latest_pmc.json (separate rocprofv3 --pmc FETCH_SIZE / "
                        "WRITE_SIZE passes of this command, committed); bench.py cannot collect PMCs itself, so `traffic` "
                        "measured in this run is null",
        "avg_launch_ms": round(avg_ms, 4),
        "poses_per_launch": dom_conv["poses"] // dom_conv["launches"],
        "algorithmic_flops_per_launch": algo,
        "algorithmic_equivalent_tflops": round(algo_tf, 2),
        "algorithmic_equivalent_over_fp32_mfma_peak": round(algo_tf / PEAK_FP32_MFMA_TFLOPS, 4),
        "mfma_executed_fraction": round(ex / (algo * (H2_SPLIT if h2 else 1.0)), 4),
    }


def is_h2(kernel_name):
    return kernel_name.endswith("_h2")


def profiled_conv_flops(prof):
    """(executed, algorithmic, pipe_seconds, executed-if-nothing-were-skipped) of the conv launches of a profile pass: counted kernels report what they
    issued, the others execute every algorithmic FLOP (x 3 on the split-fp16 kernels); pipe_seconds = the time the MFMA
    pipes need for the executed work at peak (fp32 MFMA and f16 MFMA launches priced at their own peaks)."""
    ex = al = pipe = full = 0.0
    for r in prof:
        if not r["kernel"].startswith("conv"):
            continue
        al += r["flops"]
        h2 = is_h2(r["kernel"])
        full += r["flops"] * (H2_SPLIT if h2 else 1.0)   # what the launch would execute without zero-skipping
        if r.get("mfma_counted_launches"):
            e = 4096.0 * r["mfma_executed"] * r["launches"] / r["mfma_counted_launches"]
        else:
            e = r["flops"] * (H2_SPLIT if h2 else 1.0)
        ex += e
        pipe += e / ((PEAK_F16_MFMA_TFLOPS if h2 else PEAK_FP32_MFMA_TFLOPS) * 1e12)
    return ex, al, pipe, full


def other_models(args, capi, synth, torch, dev):
    """The same C2 workload through the other shipped single models, outside the headline's timed region
    (BASELINE.json quotes "48^3 x 28ch": crossdock_default2018 is the shipped 28-channel network; default2017
    as shipped has 35 channels).  Same step/fence structure, same batch, a handful of steps each."""
    out = {}
    for name in (args.model, "crossdock_default2018", "dense"):
        if name in out:
            continue
        try:
            m = capi.Model(name)
            sc = capi.Scorer([m])
            rng = np.random.RandomState(0)
            rx, rs = synth.make_receptor(rng, args.n_rec, synth.mapped_types(m.chan_of_smt(False)))
            lx, ls = synth.make_ligand(rng, args.n_lig, synth.mapped_types(m.chan_of_smt(True)))
            poses = synth.make_poses(np.random.RandomState(1000), lx, args.batch)
            sc.set_receptor(rx, rs)
            d_lig = torch.from_numpy(poses).to(dev)
            d_o = torch.empty(4, args.batch, dtype=torch.float32, device=dev)
            def step():
                sc.score_batch_device(d_lig.data_ptr(), ls, args.batch, args.n_lig, d_o[0].data_ptr(),
                                      d_o[1].data_ptr(), d_o[2].data_ptr(), d_o[3].data_ptr())
            spin_up(step, sc.synchronize, args.spinup_seconds)
            for _ in range(2):
                step()
            sc.synchronize()
            k = max(3, min(args.steps, 10))
            blocks = []  # three timed blocks of k steps; the median one counts (a 30 ms stall of the runtime inside a 35 ms block halves it)
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(k):
                    step()
                sc.synchronize()
                blocks.append(time.perf_counter() - t0)
            dt = sorted(blocks)[1]  # the median block (the best of three is biased upward); all three are listed
            out[name] = {"poses_per_s": round(args.batch * k / dt, 1), "channels": m.n_channels,
                         "grid": m.grid_points, "steps": k, "blocks_poses_per_s": [round(args.batch * k / b, 1) for b in blocks],
                         "dtype": "f32 (split-fp16 forward convolutions)"}
            # the same model with fp32 MFMA in every layer (MI_PRECISION_FP32_MFMA), and how far the scores are apart
            s_split = d_o[:2].cpu().numpy().copy()
            sc.set_precision("fp32_mfma")
            for _ in range(2):
                step()
            sc.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            sc.synchronize()
            dt2 = time.perf_counter() - t0
            s_f32 = d_o[:2].cpu().numpy()
            out[name]["fp32_mfma_only"] = {"poses_per_s": round(args.batch * k / dt2, 1),
                                           "max_abs_dpose": float(np.abs(s_split[0] - s_f32[0]).max()),
                                           "max_abs_daffinity": float(np.abs(s_split[1] - s_f32[1]).max())}
            del sc, m
        except Exception as e:  # the headline line must still print
            out[name] = {"error": str(e)}
    # the headline model's fp32-MFMA-only figures under their own key
    if args.model in out and "fp32_mfma_only" in out[args.model]:
        out["fp32_mfma_only"] = dict(out.pop(args.model)["fp32_mfma_only"], model=args.model,
                                     note="the headline workload with v_mfma_f32_32x32x2_f32 / 16x16x4_f32 in every layer "
                                          "(the round-2 path); score differences are split-fp16 minus fp32 MFMA on this batch")
    return out


def conv1_dense(args, scorer, step, steps):
    """The dominant kernel with zero-quad skipping switched off (MI_GNINA_NO_SPARSE=1, read per launch): every
    algorithmic MAC is executed, so this rate IS the MFMA pipe's -- it does not depend on how empty the receptor is."""
    from gnina_amd import capi
    capi.set_option("MI_GNINA_NO_SPARSE", "1")
    try:
        spin_up(step, scorer.synchronize, args.spinup_seconds)
        scorer.enable_profile(True)
        for _ in range(steps):
            step()
        prof = scorer.profile()
        scorer.enable_profile(False)
    finally:
        capi.set_option("MI_GNINA_NO_SPARSE", None)
    conv = max((r for r in prof if r["kernel"].startswith("conv")), key=lambda r: r["ms_total"])
    ms = conv["ms_total"] / conv["launches"]
    tf = conv["flops"] / conv["launches"] / (ms * 1e-3) / 1e12
    h2 = is_h2(conv["kernel"])
    out = {"kernel": conv["kernel"], "avg_launch_ms": round(ms, 4), "tflops": round(tf, 2),
           "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
           "note": "no zero-skipping: executed == algorithmic FLOPs"}
    if h2:
        out["executed_f16_mfma_tflops"] = round(H2_SPLIT * tf, 2)
        out["frac_of_f16_mfma_peak"] = round(H2_SPLIT * tf / PEAK_F16_MFMA_TFLOPS, 4)
        out["note"] = ("no zero-skipping: executed f16-MFMA FLOPs == 3 x algorithmic; `tflops` is the algorithmic "
                       "(fp32-product) rate, which may exceed the fp32-MFMA peak -- the roofline of this kernel is the f16 one")
    return out


def config_c3(capi, cpu_seconds=0.0):
    """BASELINE config C3 as specified: ONE complex, exhaustiveness 64, gnina's step count, Monte-Carlo + BFGS on the
    cache grids, then merge -> refine -> CNN rescore (default ensemble) -> final energies -> rank.  A latency
    workload: 64 chains on a chip with 4,096 wave slots (evaluations/s is the figure SURVEY 8d asks for)."""
    from gnina_amd import vina_scene
    sc = vina_scene.build(0)
    lig = sc["lig"]
    T = lig["n_tors"]
    n = np.ceil(sc["size"] / np.float32(0.375)).astype(np.int32)
    span = np.float32(0.375) * n.astype(np.float32)
    begin = sc["center"].astype(np.float32) - span / 2
    end = begin + span
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    t0 = time.perf_counter()
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(begin, end, n, types, 1e3)
    vina.set_ligand(lig)
    t_setup = time.perf_counter() - t0
    n_mov = len(lig["smt"])
    steps = int(70 * 3 * (50 + n_mov + 10 * (6 + T)) / 2)      # main.cpp:441-443
    iters = (25 + n_mov) // 3
    P = capi.McParams.default(steps, iters, 50)
    seeds = np.arange(1, 65, dtype=np.uint64) * np.uint64(7919)
    vina.mc_batch(seeds[:4], begin, end, capi.McParams.default(20, iters, 50))   # warm-up (module load, allocations)
    t0 = time.perf_counter()
    cnt, e, cf, xyz, ev = vina.mc_batch(seeds, begin, end, P)
    t_mc = time.perf_counter() - t0
    t0 = time.perf_counter()
    me, mcf, mxyz = capi.merge_mc_outputs(cnt, e, cf, xyz, 2.0, 50)
    er, rcf, tries = vina.refine_batch(mcf)
    _, _, co = vina.eval_batch(rcf, want_coords=True)
    scorer = capi.Scorer(["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"])   # gnina's default ensemble
    scorer.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    out = scorer.score_batch(co, lig["smt"])
    ef, intra = vina.final_energies(rcf, float(T))
    heavy = np.nonzero(lig["smt"] > 1)[0]
    keep = capi.rank_poses(out["pose"], out["affinity"], ef, co[:, heavy], 0, 1.0)
    t_tail = time.perf_counter() - t0
    res = {"workload": f"C3: 1 complex, 64 chains x {steps} steps, {n_mov}-atom ligand / {T} torsions, receptor "
                       f"{len(sc['rec_smt'])} atoms, then merge/refine/CNN-rescore(default ensemble)/rank",
           "total_s": round(t_setup + t_mc + t_tail, 3), "setup_s": round(t_setup, 3), "mc_s": round(t_mc, 3),
           "tail_s": round(t_tail, 3), "mc_evals": int(ev.sum()), "mc_evals_per_s": round(float(ev.sum()) / t_mc),
           "poses_reported": int(len(keep)), "bound": "latency (dependent evaluations); no roofline fraction, SURVEY 8d"}
    # the bit-exact configuration: mi_vina_set_strict_order (energy sums in the reference's order -- whole chains are then
    # bit-identical to gnina's, tests/test_gpu_vina_ref.py); timed on steps // 8 of the same 64 chains, both modes
    try:
        Ps = capi.McParams.default(max(steps // 8, 50), iters, 50)
        t0 = time.perf_counter()
        _, _, _, _, ev_d = vina.mc_batch(seeds, begin, end, Ps)
        t_def = time.perf_counter() - t0
        vina.set_strict_order(True)
        t0 = time.perf_counter()
        _, _, _, _, ev_s = vina.mc_batch(seeds, begin, end, Ps)
        t_str = time.perf_counter() - t0
        vina.set_strict_order(False)
        res["strict_mode"] = {"steps": int(Ps.n_steps), "default_mode_s": round(t_def, 3), "strict_mode_s": round(t_str, 3),
                              "strict_over_default": round(t_str / t_def, 3),
                              "strict_mode_s_full_run_estimate": round(t_mc * t_str / t_def, 2),
                              "evals_per_s_strict": round(float(ev_s.sum()) / t_str)}
    except Exception as e:
        res["strict_mode"] = {"error": f"{type(e).__name__}: {e}"}
    # the chains over every visible GPU (mi_vina_pool: split by chain id, parallel_mc.cpp:183-214's fan-out over devices)
    try:
        ndev = capi.lib().mi_gnina_device_count()
        if ndev > 1:
            pool = capi.VinaPool(list(range(ndev)))

            def configure(v, rank):
                v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
                v.build_cache(begin, end, n, types, 1e3)
                v.set_ligand(lig)

            t0 = time.perf_counter()
            pool.configure(configure)
            t_cfg = time.perf_counter() - t0
            pool.mc_batch(seeds[:ndev], begin, end, capi.McParams.default(20, iters, 50))
            t0 = time.perf_counter()
            pc, pe, _, _, pev = pool.mc_batch(seeds, begin, end, P)
            t_pool = time.perf_counter() - t0
            same = bool(np.array_equal(pc, cnt) and all(np.array_equal(pe[b, :cnt[b]], e[b, :cnt[b]]) for b in range(len(seeds))))
            res["vina_pool"] = {"devices": ndev, "configure_s": round(t_cfg, 3), "mc_s": round(t_pool, 3),
                                "speedup_over_one_device": round(t_mc / t_pool, 2), "equal_to_one_handle": same}
    except Exception as e:
        res["vina_pool"] = {"error": f"{type(e).__name__}: {e}"}
    if cpu_seconds > 0:
        try:
            res["cpu_baseline"] = c3_cpu_port(vina, lig, types, begin, end, n, seeds, steps, iters, cpu_seconds)
        except Exception as e:
            res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def c3_cpu_port(vina, lig, types, begin, end, n, seeds, steps, iters, budget_s):
    """The same 64 chains on the host: oracle/vina_ref.c (the C restatement that tests/test_ref_vina.py holds
    bit-identical to the reference's monte_carlo.cpp) on the same ligand, box, seeds and cache grids, one chain per
    host thread like parallel_mc.cpp:206-210 -- on a bounded sample of the steps, scaled to the full search."""
    import threading
    from oracle import vina as ov
    tables = ov.Tables()
    gd = ov.GridDims()
    for k in range(3):
        gd.begin[k], gd.end[k], gd.n[k] = float(begin[k]), float(end[k]), int(n[k])
    grids = {t: np.ascontiguousarray(vina.cache_grid(t)) for t in types}      # the lattice the device chains ran on
    threads = min(len(seeds), os.cpu_count() or 1)
    scene0 = ov.Scene(tables, gd, grids, ov.LigandHandle(lig))
    t0 = time.perf_counter()
    _, _, _, ev0 = ov.mc_chain(scene0, begin, end, int(seeds[0]), 40, iters)
    per_step = (time.perf_counter() - t0) / 40
    rounds = -(-len(seeds) // threads)                     # chains every thread runs one after the other
    sample = int(max(20, min(steps, budget_s / (per_step * rounds))))
    evs = [0] * len(seeds)

    def work(k):
        sc = ov.Scene(tables, gd, grids, ov.LigandHandle(lig))
        for b in range(k, len(seeds), threads):
            evs[b] = ov.mc_chain(sc, begin, end, int(seeds[b]), sample, iters)[3]

    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(dt * steps / sample, 2), "unit": "s for the Monte-Carlo stage (64 chains)", "cores": threads,
            "kind": "port", "evals_per_s": round(sum(evs) / dt),
            "sample": f"{len(seeds)} chains x {sample} of {steps} steps (measured {dt:.1f} s, scaled by {steps / sample:.1f}); "
                      f"oracle/vina_ref.c = restatement of monte_carlo.cpp, bit-identical to oracle/_ref; one chain per thread"}


def config_c3_real(capi, cpu_seconds):
    """C3 on a REAL complex, GPU and the REFERENCE ITSELF side by side: the GSK3B receptor (3,460 atoms) and the 32-atom /
    10-torsion adduct ligand of the reference's test data (tests/golden/real_complex.npz holds the PDBQT texts),
    exhaustiveness 64 at gnina's step count.  CPU leg: oracle/_ref = gnina's own monte_carlo.cpp / quasi_newton.cpp /
    cache.cpp compiled unmodified at -O3, parallel_mc's fan-out (private model per task, shared cache, one chain per host
    thread); skipped where the prebuilt library is absent."""
    import tempfile
    F = np.load(os.path.join(ROOT, "tests", "golden", "real_complex.npz"))
    G = np.load(os.path.join(ROOT, "tests", "golden", "vina_goldens.npz"))
    rec_text, lig_text = bytes(F["rec_pdbqt"]).decode(), bytes(F["lig_adduct_pdbqt"]).decode()
    with tempfile.NamedTemporaryFile("w", suffix=".pdbqt", delete=False) as f:
        f.write(rec_text)
        rec_path = f.name
    try:
        rec_xyz, rec_smt = capi.read_pdbqt_receptor(rec_path)
    finally:
        os.unlink(rec_path)
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    begin, end, n = G["adduct/begin"], G["adduct/end"], [int(k) for k in G["adduct/n"]]
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    t0 = time.perf_counter()
    vina = capi.Vina()
    vina.set_receptor(rec_xyz, rec_smt)
    vina.build_cache(list(begin), list(end), n, types, 1e3)
    vina.set_ligand(lig)
    t_setup = time.perf_counter() - t0
    n_mov, T = len(lig["smt"]), lig["n_tors"]
    steps = int(70 * 3 * (50 + n_mov + 10 * (6 + T)) / 2)      # main.cpp:441-443
    iters = (25 + n_mov) // 3
    seeds = np.arange(1, 65, dtype=np.uint64) * np.uint64(7919)
    vina.mc_batch(seeds[:4], begin, end, capi.McParams.default(20, iters, 50))
    t0 = time.perf_counter()
    cnt, e, cf, xyz, ev = vina.mc_batch(seeds, begin, end, capi.McParams.default(steps, iters, 50))
    t_mc = time.perf_counter() - t0
    res = {"workload": f"C3 on a real complex: GSK3B ({len(rec_smt)} atoms, PDBQT through mi_pdbqt_read_receptor) + "
                       f"{n_mov}-atom / {T}-torsion ligand, 64 chains x {steps} steps",
           "setup_s": round(t_setup, 3), "mc_s": round(t_mc, 3), "mc_evals": int(ev.sum()),
           "mc_evals_per_s": round(float(ev.sum()) / t_mc), "best_energy": float(e[:, 0].min())}
    # the same full-length run in the mode whose chains are bit-identical to the reference (strict summation order): THIS is
    # the number to set against the reference's time when the bit-identity claim is quoted
    try:
        vina.set_strict_order(True)
        t0 = time.perf_counter()
        _, e_s, _, _, ev_s = vina.mc_batch(seeds, begin, end, capi.McParams.default(steps, iters, 50))
        t_strict = time.perf_counter() - t0
        res["strict_order_mode"] = {"mc_s": round(t_strict, 3), "mc_evals_per_s": round(float(ev_s.sum()) / t_strict),
                                    "best_energy": float(e_s[:, 0].min()),
                                    "note": "full-length run, every chain bit-identical to the reference's (see chains_bit_identical_to_reference)"}
    except Exception as ex:
        res["strict_order_mode"] = {"error": f"{type(ex).__name__}: {ex}"}
    finally:
        vina.set_strict_order(False)
    if cpu_seconds > 0:
        try:
            from oracle import ref
            if not os.path.exists(ref.LIB):
                res["cpu_baseline"] = None
                res["cpu_baseline_note"] = "oracle/_ref/libgnina_ref.so not present (built only where /root/reference is)"
            else:
                sc = ref.Scene(rec_text, lig_text)
                sc.build_grids(G["adduct/center"], G["adduct/size"])
                threads = min(64, os.cpu_count() or 1)
                sec, _ = sc.mc_parallel(seeds[:1].astype(np.uint32), 1, 40, begin, end, max_iters=iters)
                rounds = -(-64 // threads)
                sample = int(max(20, min(steps, cpu_seconds / (sec / 40 * rounds))))
                sec, best = sc.mc_parallel((seeds % np.uint64(2 ** 32)).astype(np.uint32), threads, sample, begin, end,
                                           max_iters=iters)
                # parity at the full BASELINE size: the same chains in strict-order mode must END where the reference's end
                vina.set_strict_order(True)
                _, es, _, _, _ = vina.mc_batch(seeds, begin, end, capi.McParams.default(sample, iters, 50))
                vina.set_strict_order(False)
                same = int((es[:, 0].astype(np.float32).view(np.uint32) == best.view(np.uint32)).sum())
                res["chains_bit_identical_to_reference"] = f"{same} of {len(seeds)} (strict-order mode, {sample} steps each: " \
                                                           f"best energy of every chain compared bit for bit)"
                res["cpu_baseline"] = {
                    "value": round(sec * steps / sample, 2), "unit": "s for the Monte-Carlo stage (64 chains)",
                    "cores": threads, "kind": "reference", "best_energy": float(best.min()),
                    "sample": f"64 chains x {sample} of {steps} steps (measured {sec:.1f} s, scaled by {steps / sample:.1f}); "
                              f"gnina's own monte_carlo / quasi_newton / cache compiled unmodified (oracle/_ref, g++ -O3), "
                              f"parallel_mc.cpp:183-214's fan-out"}
                cpu_s = res["cpu_baseline"]["value"]
                res["gpu_over_reference_cpu"] = {"default_mode": round(cpu_s / t_mc, 2),
                                                 "strict_order_bit_identical_mode": round(cpu_s / res["strict_order_mode"]["mc_s"], 2)
                                                 if "mc_s" in res.get("strict_order_mode", {}) else None,
                                                 "note": "the reference's Monte-Carlo stage on this box's host cores / the device's; the "
                                                         "bit-identity claim belongs to the strict-order figure"}
        except Exception as ex:
            res["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
    return res


def config_real_complex(capi, synth, torch, dev, args):
    """The headline workload on a REAL receptor instead of the synthetic one: GSK3B (3,460 atoms incl. polar hydrogens,
    the reference's test/gnina/data PDBQT, read by the native reader mi_pdbqt_read_receptor) and 1,024 rigid poses of the
    reference's 32-atom adduct ligand about its crystal position, default2017.  The zero-skip fraction of the first conv
    depends on how full the grid is, so it is measured here too; scores are checked against the CPU oracle."""
    import tempfile
    from oracle import cnn_ref, voxel
    F = np.load(os.path.join(ROOT, "tests", "golden", "real_complex.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".pdbqt", delete=False) as f:
        f.write(bytes(F["rec_pdbqt"]).decode())
        rec_path = f.name
    try:
        rec_xyz, rec_smt = capi.read_pdbqt_receptor(rec_path)
    finally:
        os.unlink(rec_path)
    lig = capi.read_pdbqt_ligand(bytes(F["lig_adduct_pdbqt"]).decode(), is_text=True)
    heavy = lig["smt"] > 1                           # DLScorer::setLigand hands over every ligand atom; H map to no channel
    lig_xyz, lig_smt = lig["coords0"].astype(np.float32), lig["smt"].astype(np.int32)
    B = args.batch
    poses = synth.make_poses(np.random.RandomState(77), lig_xyz, B)
    name = "default2017"
    m = capi.Model(name)
    sc = capi.Scorer([m])
    sc.set_receptor(rec_xyz, rec_smt)
    d_lig = torch.from_numpy(poses).to(dev)
    d_o = torch.empty(4, B, dtype=torch.float32, device=dev)
    L = poses.shape[1]

    def step():
        sc.score_batch_device(d_lig.data_ptr(), lig_smt, B, L, d_o[0].data_ptr(), d_o[1].data_ptr(), d_o[2].data_ptr(),
                              d_o[3].data_ptr())
    spin_up(step, sc.synchronize, args.spinup_seconds)
    k = max(3, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    sc.synchronize()
    dt = time.perf_counter() - t0
    sc.enable_profile(True)
    for _ in range(k):
        step()
    prof = sc.profile()
    sc.enable_profile(False)
    got = d_o.cpu().numpy()
    convs = [r for r in prof if r["kernel"].startswith("conv")]
    dom = max(convs, key=lambda r: r["ms_total"])
    avg_ms = dom["ms_total"] / dom["launches"]
    exf = 4096.0 * dom["mfma_executed"] / dom["mfma_counted_launches"] if dom.get("mfma_counted_launches") else None
    rb = roofline_block(dom, avg_ms, exf, "device counters of this run")
    blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    dp = da = 0.0
    nchk = 3
    nz = 0.0
    for b in range(nchk):
        grid, _ = voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None, blob.resolution,
                                      blob.dimension, blob.radius_scaling)
        nz += float((grid != 0).mean()) / nchk
        with torch.no_grad():
            p_, a_, _l = cnn_ref.scores(blob, grid[None])
        dp, da = max(dp, abs(float(p_[0]) - got[0, b])), max(da, abs(float(a_[0]) - got[1, b]))
    vox = [r for r in prof if r["kernel"].startswith("voxelize")][0]
    return {"workload": f"GSK3B receptor ({len(rec_smt)} atoms, native PDBQT reader) + {int(heavy.sum())}-heavy-atom "
                        f"ligand ({L} atoms passed), {B} rigid poses, {name}, 48^3 x {m.n_channels}ch",
            "poses_per_s": round(B * k / dt, 1), "ms_per_step": round(1e3 * dt / k, 3),
            "voxelize_ms": round(vox["ms_total"] / vox["launches"], 4),
            "nonzero_voxel_fraction": round(nz, 4),
            "roofline": {kk: rb[kk] for kk in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                               "mfma_executed_fraction", "algorithmic_equivalent_tflops")},
            "score_delta_vs_cpu_oracle": {"poses": nchk, "max_abs_dpose": dp, "max_abs_daffinity": da}}


def config_seam_b1(capi, synth):
    """What an UNMODIFIED gnina sees through the DLScorer seam: one pose per call (DLScorer::score, cnn_torch_scorer.cpp:105-198;
    do_search's rescoring tail makes up to 50 of them per ligand, main.cpp:324-361), host pointers, synchronous.  Per-call
    latency on the default (split-fp16) path, forward and forward + backward, from one thread and from four threads with
    one scorer each (gnina's threading model: one fresh_copy() per worker, main.cpp:1438)."""
    import threading
    rng = np.random.RandomState(0)
    m0 = capi.Model("crossdock_default2018")
    rt, lt = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rt)
    lx, ls = synth.make_ligand(rng, 32, lt)
    pose1 = synth.make_poses(rng, lx, 1)
    out = {"note": "B = 1 per call, host pointers, synchronous; median of 60 calls after 0.3 s of warm-up calls; microseconds.  "
                   "two / four_threads: scorers on host threads of their own (gnina: one fresh_copy() per worker thread), scoring "
                   "side by side -- round 6 removed the per-device lock of round 5 (its cause, packed-fp32 instructions next to "
                   "another queue's MFMAs, is compiled out: DESIGN.md §6); a scorer that has the device to itself runs an "
                   "ensemble's models on streams of their own (lanes; gradient calls too), scorers that share it stay on their own streams; "
                   "per-pose launches have their own tiles, operand rings and prologue fetches (DESIGN.md §3.2)"}
    for label, models in (("default2017", ["default2017"]),
                          ("default_ensemble", ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"])):
        s = capi.Scorer(models)
        s.set_receptor(rec_xyz, rec_smt)
        row = {}
        for grad in (False, True):
            f = (lambda: s.score_grad(pose1, ls)) if grad else (lambda: s.score_batch(pose1, ls))
            t_end = time.perf_counter() + 0.3   # (warm-up calls until the clocks have settled: spin_up's docstring)
            while time.perf_counter() < t_end:
                f()
            ts = []
            for _ in range(60):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            row["fwd_bwd_us" if grad else "fwd_us"] = round(float(np.median(ts)) * 1e6, 1)
        # a ligand's nine output poses in one call (lanes serve calls of up to 64 poses: DESIGN.md §3.2)
        pose9 = synth.make_poses(np.random.RandomState(9), lx, 9)
        for _ in range(20):
            s.score_batch(pose9, ls)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            s.score_batch(pose9, ls)
            ts.append(time.perf_counter() - t0)
        row["fwd_us_nine_poses_per_call"] = round(float(np.median(ts)) * 1e6, 1)
        # four worker threads, one scorer each (ctypes releases the GIL during the call)
        scorers = []
        for _ in range(4):
            si = capi.Scorer(models)
            si.set_receptor(rec_xyz, rec_smt)
            si.score_batch(pose1, ls)
            scorers.append(si)
        n_calls = 150

        def work(si):
            for _ in range(n_calls):
                si.score_batch(pose1, ls)

        for nt, key in ((2, "two_threads_poses_per_s"), (4, "four_threads_poses_per_s")):
            th = [threading.Thread(target=work, args=(si,)) for si in scorers[:nt]]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - t0
            row[key] = round(nt * n_calls / dt, 1)
            time.sleep(0.05)  # (the scorers' "somebody else is active" window runs out before the next measurement)
        row["one_thread_poses_per_s"] = round(1e6 / row["fwd_us"], 1)
        out[label] = row
    return out


def config_c4(capi, synth):
    """BASELINE config C4, one GPU's share of it: a ragged batch of 1,024 ligands x 9 poses through the 15-model
    crossdock_default2018 ensemble (one voxelization, 15 forwards per pose), host pointers in and out."""
    names = ["crossdock_default2018" + s for s in ("", "_1", "_2", "_3", "_4", "_1_3", "_1_3_1", "_1_3_2", "_1_3_3",
                                                    "_1_3_4", "_KD_1", "_KD_2", "_KD_3", "_KD_4", "_KD_5")]
    have = [n for n in names if os.path.exists(os.path.join(ROOT, "gnina_amd", "weights", n + ".mgw"))]
    models = [capi.Model(n) for n in have] + [capi.Model("crossdock_default2018") for _ in range(15 - len(have))]
    s = capi.Scorer(models)
    rng = np.random.RandomState(0)
    m0 = models[0]
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m0.chan_of_smt(False)))
    lig_types = synth.mapped_types(m0.chan_of_smt(True))
    s.set_receptor(rec_xyz, rec_smt)
    n_lig, P, Lmax = 1024, 9, 48
    xyz = np.zeros((n_lig * P, Lmax, 3), dtype=np.float32)
    smt = np.full((n_lig * P, Lmax), -1, dtype=np.int32)
    for i in range(n_lig):
        L = rng.randint(16, 49)
        lx, ls = synth.make_ligand(rng, L, lig_types)
        xyz[i * P:(i + 1) * P, :L] = synth.make_poses(rng, lx, P)
        smt[i * P:(i + 1) * P, :L] = ls
    s.score_ragged(xyz[:P * 64], smt[:P * 64])
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        s.score_ragged(xyz, smt)
    dt = (time.perf_counter() - t0) / reps
    fwd = 15 * n_lig * P / dt
    tf = fwd * FLOP_PER_POSE["crossdock_default2018"] / 1e12
    s.enable_profile(True)                       # one more pass, untimed: what the MFMA pipe executed
    s.score_ragged(xyz, smt)
    ex, al, pipe_s, full = profiled_conv_flops(s.profile())
    s.enable_profile(False)
    ex_tf = ex / dt / 1e12
    return {"workload": f"C4 (one GPU's shard): 1,024 ligands x 9 poses, L ~ U{{16..48}}, ragged, 15 x Default2018 "
                        f"({len(have)} distinct weight blobs), host pointers",
            "ligands_per_s": round(n_lig / dt, 1), "poses_per_s": round(n_lig * P / dt, 1),
            "model_forwards_per_s": round(fwd, 1), "s_per_100k_ligands_1gpu": round(1e5 / (n_lig / dt), 1),
            "roofline": {"bound": "mfma", "achieved": round(ex_tf, 2), "peak": round(ex / pipe_s / 1e12, 1) if pipe_s else None,
                         "unit": "TFLOP/s", "frac": round(pipe_s / dt, 4),
                         "achieved_is": "executed MFMA FLOPs of all conv launches / wall time of the call; peak = the same "
                                        "FLOPs / the time the pipes need for them at peak (f16-MFMA launches of the "
                                        "split-fp16 path and fp32-MFMA launches priced at their own peaks)",
                         "mfma_executed_fraction": round(ex / full, 4) if full else None,
                         "algorithmic_equivalent_tflops": round(tf, 2),
                         "algorithmic_equivalent_over_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         "note": "end to end incl. voxelization, PCIe and host set-up"}}


def config_c5(capi, synth):
    """BASELINE config C5's network on one GPU: dense_1_3 at 0.25 A (96^3, 36.3 GFLOP per pose), B = 256, forward and
    forward + backward (one BFGS evaluation of CNN refinement): the parity path, the bf16-MFMA path and (round 6) the fp16 mode."""
    m = capi.Model("dense_1_3", resolution=0.25, dimension=23.75)
    s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    B = 256
    poses = synth.make_poses(rng, lx, B)
    out = {"workload": "C5 network: dense_1_3 @ 0.25 A (96^3 x 28ch), B = 256, host pointers"}
    gf = FLOP_PER_POSE["dense_1_3@96"]
    for tag, bf, peak in (("f32", False, PEAK_FP32_MFMA_TFLOPS), ("bf16", True, PEAK_BF16_MFMA_TFLOPS), ("fp16", "fp16", PEAK_BF16_MFMA_TFLOPS)):
        # (fp16, round 6: MI_PRECISION_FP16 -- the parity path's kernels and tensors with the h * h MFMA only; its gradient calls
        # are the parity path's)
        s.set_precision(bf)
        dt = dg = 1e30
        for _ in range(3):                    # warm-up at the full batch (activation buffers are allocated once; the clocks need
            s.score_batch(poses, ls)          # ~30 ms of work to settle: spin_up's docstring), then the better of two calls
        for _ in range(2):
            t0 = time.perf_counter()
            s.score_batch(poses, ls)
            dt = min(dt, time.perf_counter() - t0)
        s.score_grad(poses, ls)
        for _ in range(2):
            t0 = time.perf_counter()
            s.score_grad(poses, ls)
            dg = min(dg, time.perf_counter() - t0)
        tf = B / dt * gf / 1e12
        s.enable_profile(True)
        s.score_batch(poses, ls)
        ex, al, pipe_s, full = profiled_conv_flops(s.profile())
        s.enable_profile(False)
        if bf:      # bf16 / fp16 kernels: executed == algorithmic (one MFMA per product), priced at the bf16 = f16 peak
            ach, pk, frac = tf, peak, tf / peak
        else:       # parity path: split-fp16 (f16 peak) and fp32-MFMA launches, each priced at its own peak
            ach = ex / dt / 1e12
            pk = ex / pipe_s / 1e12 if pipe_s else peak
            frac = pipe_s / dt
        out[tag] = {"poses_per_s_forward": round(B / dt, 1), "poses_per_s_forward_backward": round(B / dg, 1),
                    "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": round(pk, 1), "unit": "TFLOP/s",
                                 "frac": round(frac, 4),
                                 "achieved_is": "executed MFMA FLOPs / wall time of the forward call (peak: the launches' own "
                                                "MFMA peaks, FLOP-weighted)",
                                 "algorithmic_equivalent_tflops": round(tf, 2),
                                 "mfma_executed_fraction": round(ex / full, 4) if full else None,
                                 "note": "forward, end to end (voxelization, PCIe, host set-up included)"}}
    s.set_precision(False)
    out["refine"] = config_c5_refine(capi, s)
    return out


def config_c5_refine(capi, scorer):
    """C5 as BASELINE.json configures it: `--cnn_scoring refinement` with the Dense model at 96^3 -- refine_structure on
    non_cache_cnn (main.cpp:131-171; one TorchModel::forward + backward + GridMaker::backward per BFGS evaluation,
    torch_model.cpp:197-221) -- end to end through mi_cnn_refine_batch: B = 64 perturbed conformations of the C3 ligand
    (32 atoms, 6 torsions) in the C3 receptor, default iteration cap (25 + n_atoms) / 3, parity path (split-fp16 / fp32)."""
    from gnina_amd import vina_scene
    sc = vina_scene.build(seed=3)
    lig = sc["lig"]
    v = capi.Vina()
    v.set_ligand(lig)
    scorer.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(5)
    B = 64
    confs = np.repeat(lig["conf0"].astype(np.float32)[None], B, 0)
    confs[:, :3] += rng.uniform(-0.5, 0.5, (B, 3)).astype(np.float32)
    confs[:, 7:] += rng.uniform(-0.4, 0.4, (B, confs.shape[1] - 7)).astype(np.float32)
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    box = capi.CnnBox.make(23.75, lo, hi)
    start, _ = v.cnn_eval_batch(scorer, confs, box, None, deriv=False)
    v.cnn_refine_batch(scorer, confs[:8], box, max_iters=2)     # warm-up (allocations at the gradient program's sizes)
    t0 = time.perf_counter()
    e, out, tries, evals = v.cnn_refine_batch(scorer, confs, box)
    dt = time.perf_counter() - t0
    return {"workload": "dense_1_3 @ 0.25 A (96^3), mi_cnn_refine_batch, B = 64 conformations, max_iters = (25 + 32) / 3 = 19",
            "seconds": round(dt, 3), "poses_refined_per_s": round(B / dt, 2),
            "cnn_evaluations": int(evals.sum()), "cnn_evaluations_per_s": round(float(evals.sum()) / dt, 1),
            "mean_loss_before": round(float(start.mean()), 5), "mean_loss_after": round(float(e.mean()), 5),
            "all_losses_lowered_or_equal": bool((e <= start + 2e-5 * np.maximum(1.0, np.abs(start))).all())}


def config_gradient_calls(capi, synth):
    """One BFGS evaluation of CNN refinement (score + d loss / d atoms, torch_model.cpp:197-221; non_cache_cnn.cpp:79-169)
    at the headline grid, B = 256 poses per call, host pointers: the 3x3x3 transposed convs on the split-fp16 kernel (the
    default) against the same call with every transposed conv on fp32 MFMA (MI_GNINA_NO_H2_BWD=1, the round-3 path), and
    how far the atom gradients of the two are apart."""
    out = {"note": "48^3 grid, B = 256, receptor 2500 atoms, ligand 32 atoms; wall time of mi_scorer_score_grad, best of 5 after 0.3 s of warm-up calls; "
                   "max_rel_grad_diff = max over poses of max |g_split - g_fp32| / max |g_fp32|"}
    B = 256
    for name in ("default2017", "crossdock_default2018", "dense"):
        m = capi.Model(name)
        s = capi.Scorer([m])
        rng = np.random.RandomState(0)
        rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
        lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
        s.set_receptor(rec_xyz, rec_smt)
        poses = synth.make_poses(rng, lx, B)
        row, grads = {}, {}
        for tag, env in (("split_fp16", None), ("fp32_mfma_transposed", "1")):
            capi.set_option("MI_GNINA_NO_H2_BWD", env)
            grads[tag] = s.score_grad(poses, ls)["lig_grad"]
            t_end = time.perf_counter() + 0.3   # (warm-up calls until the clocks have settled: spin_up's docstring)
            while time.perf_counter() < t_end:
                s.score_grad(poses, ls)
            best = 1e30
            for _ in range(5):
                t0 = time.perf_counter()
                s.score_grad(poses, ls)
                best = min(best, time.perf_counter() - t0)
            row[tag + "_poses_per_s"] = round(B / best, 1)
        capi.set_option("MI_GNINA_NO_H2_BWD", None)
        ga, gb = grads["split_fp16"].reshape(B, -1), grads["fp32_mfma_transposed"].reshape(B, -1)
        row["max_rel_grad_diff"] = float((np.abs(ga - gb).max(1) / np.maximum(np.abs(gb).max(1), 1e-30)).max())
        row["speedup"] = round(row["split_fp16_poses_per_s"] / row["fp32_mfma_transposed_poses_per_s"], 3)
        out[name] = row
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: one rank per GPU over RCCL
        # (ProcessGroupNCCL's heartbeat monitor thread: the same step measures 3.71 ms without it against 3.77 with it,
        # and 3.63 ms in a process without a process group -- DESIGN.md §5)
        os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gnina_amd import capi, synth
    capi.init(local_rank)
    model = capi.Model(args.model)
    scorer = capi.Scorer([model])
    if args.chunk:
        scorer.set_chunk(args.chunk)
    rec_types = synth.mapped_types(model.chan_of_smt(False))
    lig_types = synth.mapped_types(model.chan_of_smt(True))
    # every rank has the same receptor (replicated) and its own shard of poses: rank 0 makes the receptor
    # and the ligand and hands them out with one broadcast over RCCL, like a screening driver would
    from gnina_amd import shard
    rng0 = np.random.RandomState(0)
    if rank == 0:
        rec_xyz, rec_smt = synth.make_receptor(rng0, args.n_rec, rec_types)
        lig_xyz, lig_smt = synth.make_ligand(rng0, args.n_lig, lig_types)
        blob = [rec_xyz, rec_smt, lig_xyz, lig_smt]
    else:
        blob = None
    rec_xyz, rec_smt, lig_xyz, lig_smt = shard.broadcast_arrays(blob, dist, dev)
    bcast_bytes = int(rec_xyz.nbytes + rec_smt.nbytes + lig_xyz.nbytes + lig_smt.nbytes)
    poses = synth.make_poses(np.random.RandomState(1000 + rank), lig_xyz, args.batch)
    scorer.set_receptor(rec_xyz, rec_smt)

    B, L = args.batch, args.n_lig
    d_lig = torch.from_numpy(poses).to(dev)
    d_out = torch.empty(4, B, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        scorer.score_batch_device(d_lig.data_ptr(), lig_smt, B, L, d_out[0].data_ptr(), d_out[1].data_ptr(),
                                  d_out[2].data_ptr(), d_out[3].data_ptr())

    def fence():
        scorer.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # (for one round, so that the round-1-4 lines stay comparable: the same W + K steps timed WITHOUT the spin-up in front of
    # them -- what bench.py measured until round 4 -- reported as value_without_spinup; the device has been idle since the
    # set-up, as it was then)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    scorer.synchronize()
    torch.cuda.synchronize()
    elapsed_cold = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed_cold], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_cold = float(t.item())

    spinup_steps = spin_up(step, scorer.synchronize, args.spinup_seconds)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    scorer.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel timing of the same steps with HIP events on the scorer's stream (separate pass so the
    # event records do not perturb the timed region)
    scorer.enable_profile(True)
    for _ in range(args.steps):
        step()
    prof = scorer.profile()
    scorer.enable_profile(False)
    gpu_scores = d_out.cpu().numpy()

    # The headline is complete here.  Everything below is outside the timed region ("also": other models, C3 / C4 / C5,
    # the RCCL screening pattern, the in-process pool): a hang in one of those extras -- a collective that a rank never
    # reaches, the pool driver's first run on a multi-GPU box -- must not cost the headline line.  A watchdog on every rank
    # bounds the extras: when it fires, rank 0 prints the line with what is there and names the phase that did not
    # return, and every rank leaves (os._exit: the blocked thread cannot be joined).
    res = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        dom = max(prof, key=lambda r: r["ms_total"])
        convs = [r for r in prof if r["kernel"].startswith("conv")]
        dom_conv = max(convs, key=lambda r: r["ms_total"])
        avg_ms = dom_conv["ms_total"] / dom_conv["launches"]
        achieved = dom_conv["flops"] / dom_conv["launches"] / (avg_ms * 1e-3) / 1e12
        total_kernel_ms = sum(r["ms_total"] for r in prof) / args.steps
        if dom_conv.get("mfma_counted_launches"):
            exec_flops = 4096.0 * dom_conv["mfma_executed"] / dom_conv["mfma_counted_launches"]
            exec_src = "device counters of this run (mi_scorer_enable_profile)"
        else:  # a library without the counters: fall back to the committed rocprofv3 PMC summary
            exec_flops = (pmc_entry(dom_conv["kernel"]) or {}).get("mfma_executed_flops_per_launch")
            exec_src = "profiles/latest_pmc.json (SQ_INSTS_VALU_MFMA_MOPS_F32)"
        vox = [r for r in prof if r["kernel"].startswith("voxelize")][0]
        vox_ms = vox["ms_total"] / vox["launches"]
        res = {
            "metric": "CNN-scored poses/sec (voxelize + CNN forward, 48^3 grid)",
            "value": round(value, 1),
            "unit": "poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup": {"seconds": args.spinup_seconds, "steps": spinup_steps,
                       "note": "untimed steps in front of the W warm-up steps: the clocks need ~30 ms of continuous work "
                               "after an idle period (bench.py spin_up)"},
            "value_without_spinup": round(world * B * args.steps / elapsed_cold, 1),
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_NOTE if os.environ.get("MI_GNINA_CONV_PATH") != "f32" else "f32",
            "data": "synthetic atoms (SURVEY 8d C2, seeded); real reference weights extracted from the shipped .pt",
            "config": {
                "workload": f"C2: {B} poses/GPU/step, receptor {args.n_rec} atoms, ligand {args.n_lig} atoms, "
                            f"{model.grid_points}^3 x {model.n_channels}ch grid at {model.resolution} A, "
                            f"model {model.name} (as shipped)",
                "model_file": model.name, "channels": model.n_channels, "batch_per_gpu": B,
                "sharding": f"pose-sharded x{world}, no data-path collective",
            },
            "roofline": roofline_block(dom_conv, avg_ms, exec_flops, exec_src),
            "kernels": [{"kernel": r["kernel"], "launches_per_step": r["launches"] // args.steps,
                         "ms_per_step": round(r["ms_total"] / args.steps, 4),
                         "tflops": round(r["flops"] / (r["ms_total"] * 1e-3) / 1e12, 2) if r["flops"] else None,
                         "gbs_algorithmic": (round(r["bytes"] / (r["ms_total"] * 1e-3) / 1e9, 1)
                                             if not r["kernel"].startswith("voxelize") else None)}
                        for r in prof],
            "voxelizer": {"bound": "hbm", "ms_per_launch": round(vox_ms, 4),
                          "written_GBs": round(vox["bytes"] / 8 / vox["launches"] / (vox_ms * 1e-3) / 1e9, 1),
                          "peak_GBs": PEAK_HBM_GBS,
                          "frac": round(vox["bytes"] / 8 / vox["launches"] / (vox_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                          "poses_per_launch": vox["poses"] // vox["launches"],
                          "valu_issue_from_committed_pmc": voxelizer_issue_committed(),
                          "note": "bytes actually written: the 2x2x2-pooled grid, C*(N/2)^3*4 per pose (the un-fused "
                                  "C*N^3*4 figure of SURVEY 8d is 8x that and never exists in HBM); bound by its vector "
                                  "instruction issue rate (valu_issue_from_committed_pmc), not by HBM"},
            "sum_kernel_ms_per_step": round(total_kernel_ms, 3),
            "dominant_kernel_overall": dom["kernel"],
        }

    import threading
    extras_phase = {"name": "start"}

    def extras_timeout():
        if rank == 0 and res is not None:
            res["extras_incomplete"] = {"phase": extras_phase["name"], "seconds": args.extras_timeout,
                                        "note": "the watchdog ended the extras; the headline (timed region) above is complete"}
            print(json.dumps(res, default=float), flush=True)
        os._exit(0)

    watchdog = threading.Timer(args.extras_timeout, extras_timeout)
    watchdog.daemon = True
    watchdog.start()

    # outside the timed region: the screening pattern end to end on a fixed pose set -- contiguous pose shards scored
    # by their ranks, one all_gather of 4 floats per pose over RCCL (shard.score_sharded) -- checked on rank 0 against
    # scoring the whole set alone: the same bits, in pose order
    rccl = None
    extras_phase["name"] = "rccl: sharded screening pattern (all_gather)"
    if dist is not None:
        vposes = synth.make_poses(np.random.RandomState(4242), lig_xyz, 256 * world + 3)

        def score_np(p):
            o = scorer.score_batch(p, lig_smt)
            return np.stack([o["pose"], o["affinity"], o["loss"], o["variance"]], 1)

        gathered = shard.score_sharded(score_np, vposes, dist, dev)
        if rank == 0:
            alone = score_np(vposes)
            rccl = {"ranks": world, "backend": dist.get_backend(), "receptor_broadcast_bytes": bcast_bytes,
                    "sharded_poses": int(len(vposes)), "allgather_equals_single_rank": bool(np.array_equal(gathered, alone))}
            assert rccl["allgather_equals_single_rank"], "sharded scores differ from single-rank scores"
        dist.barrier()

    # strong scaling beside the weak-scaling headline: ONE batch of 8,192 poses, contiguous shards over the ranks
    # (SURVEY 8e), inputs resident in HBM, barrier + max over ranks like the timed region above
    strong = None
    extras_phase["name"] = "strong scaling (one 8,192-pose batch over the ranks)"
    try:
        S_TOTAL = 8192
        sp = synth.make_poses(np.random.RandomState(777), lig_xyz, S_TOTAL)
        lo, hi = shard.shard_range(S_TOTAL, rank, world)
        d_sp = torch.from_numpy(np.ascontiguousarray(sp[lo:hi])).to(dev)
        d_so = torch.empty(4, hi - lo, dtype=torch.float32, device=dev)

        def sstep():
            scorer.score_batch_device(d_sp.data_ptr(), lig_smt, hi - lo, L, d_so[0].data_ptr(), d_so[1].data_ptr(),
                                      d_so[2].data_ptr(), d_so[3].data_ptr())
        sstep()
        fence()
        t0 = time.perf_counter()
        for _ in range(3):
            sstep()
        scorer.synchronize()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 3
        if dist is not None:
            dist.barrier()
            t = torch.tensor([dt_s], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
        strong = {"total_poses": S_TOTAL, "ranks": world, "ms": round(1e3 * dt_s, 3), "poses_per_s": round(S_TOTAL / dt_s, 1),
                  "note": "one fixed batch, contiguous pose shards, max over ranks"}
    except Exception as e:      # the headline line must still print
        strong = {"error": f"{type(e).__name__}: {e}"}

    # the C++ in-process pool (mi_pool: one host thread + scorer per GPU, RCCL scatter / gather for device-resident
    # poses) over the same GPUs, driven by its C++ test driver in a child process while the ranks wait: pool sizes
    # 1, 2, 4 .. N on one 8,192-pose batch, bit-equality with the single scorer checked inside the driver
    pool_res = None
    extras_phase["name"] = "pool_in_process (tests/cpp/test_pool.cpp child process)"
    if rank == 0:
        import subprocess
        exe = os.path.join(ROOT, "gnina_amd", "lib", "test_pool")
        try:
            env = dict(os.environ)
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            out_txt, rc, err_txt = "", None, ""
            try:
                r = subprocess.run([exe, os.path.join(ROOT, "gnina_amd", "weights"), str(world), "8192"],
                                   capture_output=True, text=True, timeout=150, env=env)
                out_txt, rc, err_txt = r.stdout, r.returncode, r.stderr
            except subprocess.TimeoutExpired as te:       # keep what the driver printed before it stalled
                out_txt = te.stdout.decode() if isinstance(te.stdout, bytes) else (te.stdout or "")
                err_txt = "timeout after 150 s"
            rows = [l.split() for l in out_txt.split("\n") if l.startswith("pool ")]
            hrows = [l.split() for l in out_txt.split("\n") if l.startswith("pool_host ")]
            pool_res = {"driver": "tests/cpp/test_pool.cpp (C++ only: mi_pool over the visible GPUs, one 8,192-pose batch)",
                        "returncode": rc,
                        "host_path": [{"devices": int(l[2]), "equal_to_single_scorer": l[6] == "1", "poses_per_s": float(l[8])}
                                      for l in hrows],
                        "device_path_rccl": [{"devices": int(l[2]), "equal_to_single_scorer": l[10] == "1",
                                              "poses_per_s": float(l[12])} for l in rows]}
            if rc != 0:
                pool_res["stderr"] = err_txt[-400:]
        except Exception as e:
            pool_res = {"error": f"{type(e).__name__}: {e}"}
    if dist is not None:
        dist.barrier()

    if rank == 0:
        extras_phase["name"] = "cpu_baseline"
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_scores = cpu_baseline(args, os.path.join(ROOT, "gnina_amd", "weights", args.model + ".mgw"),
                                          rec_xyz, rec_smt, lig_smt, poses, args.cpu_seconds)
            res["cpu_baseline"] = cb
            cs = np.array(cpu_scores, dtype=np.float64)
            n = len(cs)
            res["score_delta_vs_cpu_oracle"] = {
                "poses": n,
                "max_abs_dpose": float(np.abs(gpu_scores[0, :n] - cs[:, 0]).max()),
                "max_abs_daffinity": float(np.abs(gpu_scores[1, :n] - cs[:, 1]).max())}
            res["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
        if rccl is not None:
            res["rccl"] = rccl
        res["strong_scaling"] = strong
        res["pool_in_process"] = pool_res
        if world == 1:
            extras_phase["name"] = "also: other models"
            res["also"] = other_models(args, capi, synth, torch, dev)
            res["roofline"]["dense_no_skip"] = conv1_dense(args, scorer, step, max(3, min(args.steps, 10)))
            if not args.no_configs:
                cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
                for key, fn in (("real_complex", lambda: config_real_complex(capi, synth, torch, dev, args)),
                                ("c3", lambda: config_c3(capi, cpu_s)), ("c3_real", lambda: config_c3_real(capi, cpu_s)),
                                ("c4", lambda: config_c4(capi, synth)), ("c5", lambda: config_c5(capi, synth)),
                                ("seam_b1", lambda: config_seam_b1(capi, synth)),
                                ("gradient_calls", lambda: config_gradient_calls(capi, synth))):
                    if args.only and key not in args.only.split(","):
                        continue
                    extras_phase["name"] = "also." + key
                    try:
                        res["also"][key] = fn()
                    except Exception as e:  # the headline line must still print
                        res["also"][key] = {"error": f"{type(e).__name__}: {e}"}
        watchdog.cancel()
        print(json.dumps(res, default=float), flush=True)   # (numpy scalars from the sub-benchmarks)
    watchdog.cancel()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
