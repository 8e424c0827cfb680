#!/usr/bin/env python3
"""What does a split-operand half-precision MFMA path cost in score accuracy?  (VERDICT r2, next #6.)

CPU experiment, no GPU: runs the CNN forward with every convolution's operands (BatchNorm'ed input, weights) replaced
by the sum of the products a split path keeps, accumulated in float64 and rounded to fp32 per layer -- so the number
it prints is the split error alone, without accumulation-order noise:

  f32      : operands as they are (the fp32-MFMA path, modulo summation order)
  f16x2    : a = h + l, both fp16 (RN), products h*h + h*l + l*h      (3 MFMAs, 16x the fp32-MFMA rate each)
  f16x2s   : the same with power-of-two pre-scaling of weights (2^8) and activations (2^4) against fp16 underflow
  bf16x3   : a = h + m + l in bf16, products hh + hm + mh + mm + hl + lh (6 MFMAs)
  bf16x2   : h + l in bf16, 3 products  (for scale: the cheap split is NOT parity grade)

Prints max |delta| of pose score and affinity against the float64 forward of the unsplit fp32 operands.
    python tools/experiments/split_precision_probe.py [model ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import synth  # noqa: E402
from oracle import cnn_ref, voxel  # noqa: E402


FTZ = bool(int(os.environ.get("SPLIT_FTZ", "0")))  # what if the MFMA flushed fp16 subnormal inputs?


def split(x, dt, n, scale=1.0):
    """x (float64 holding fp32 values) -> n parts of dtype dt (as float64), RN at each step."""
    parts, r = [], x * scale
    for _ in range(n):
        p = r.to(torch.float32).to(dt).to(torch.float64)
        if FTZ and dt == torch.float16:
            p = torch.where(p.abs() < 2.0 ** -14, torch.zeros_like(p), p)
        parts.append(p)
        r = r - p
    return parts


def conv_split(xin, w, k, mode):
    if mode == "f32":
        return F.conv3d(xin, w, None, padding=k // 2)
    if mode.startswith("f16x2"):
        # f16x2:<sa>:<sw> = explicit scales; sw "a" = per layer, the largest power of two with max|w| * sw <= 2^14
        sa, sw = (16.0, 256.0) if mode == "f16x2s" else (1.0, 1.0)
        if ":" in mode:
            _, sa_s, sw_s = mode.split(":")
            sa = float(sa_s)
            sw = 2.0 ** np.floor(14 - np.log2(float(w.abs().max()))) if sw_s == "a" else float(sw_s)
        a, b = split(xin, torch.float16, 2, sa), split(w, torch.float16, 2, sw)
        keep = [(0, 0), (0, 1), (1, 0)]
        inv = 1.0 / (sa * sw)
    elif mode == "bf16x3":
        a, b = split(xin, torch.bfloat16, 3), split(w, torch.bfloat16, 3)
        keep = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
        inv = 1.0
    elif mode == "bf16x2":
        a, b = split(xin, torch.bfloat16, 2), split(w, torch.bfloat16, 2)
        keep = [(0, 0), (0, 1), (1, 0)]
        inv = 1.0
    else:
        raise ValueError(mode)
    y = 0
    for i, j in keep:
        y = y + F.conv3d(a[i], b[j], None, padding=k // 2)
    return y * inv


def forward(blob, grid, mode):
    """oracle/cnn_ref.forward_logits with the convolutions' operands split; activations are rounded to fp32 per layer"""
    f32 = lambda t: t.to(torch.float32).to(torch.float64)
    bufs = {0: torch.as_tensor(grid).to(torch.float64)}
    out = None
    for t in blob.ops:
        if t[0] == "pool":
            src, dst = int(t[2]), int(t[3])
            bufs[dst] = F.max_pool3d(bufs[src], 2, 2) if t[1] == "max" else f32(F.avg_pool3d(bufs[src], 2, 2))
        elif t[0] == "conv":
            k, src, dst, cin, cout, c0, relu = (int(v) for v in t[1:8])
            w_off, b_off, s_off, t_off = (int(v) for v in t[8:12])
            w = blob.tensor(w_off, (k * k * k, cin, cout)).double()
            w = w.reshape(k, k, k, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
            b = blob.tensor(b_off, (cout,)).double()
            xin = bufs[src][:, :cin]
            if s_off >= 0:
                s = blob.tensor(s_off, (cin,)).view(1, -1, 1, 1, 1)
                sh = blob.tensor(t_off, (cin,)).view(1, -1, 1, 1, 1)
                xin = (xin.float() * s + sh).double()  # fp32 BatchNorm, as the staging does it
            y = f32(conv_split(xin, w, k, mode) + b.view(1, -1, 1, 1, 1))
            if relu:
                y = torch.relu(y)
            bufs[dst] = torch.cat([bufs[src], y], 1) if src == dst else y
        elif t[0] == "gmax":
            bufs[int(t[2])] = torch.amax(bufs[int(t[1])], dim=(2, 3, 4), keepdim=True)
        elif t[0] == "fc":
            src, n_in, w_off, b_off = (int(v) for v in t[1:5])
            xin = bufs[src]
            B, C = xin.shape[0], xin.shape[1]
            S3 = n_in // C
            w = blob.tensor(w_off, (3, S3, C)).double()
            b = blob.tensor(b_off, (3,)).double()
            flat = xin.reshape(B, C, S3).permute(0, 2, 1).reshape(B, S3 * C)
            out = flat @ w.reshape(3, S3 * C).t() + b
    logp = torch.log_softmax(out[:, :2], 1)
    return torch.softmax(logp, 1)[:, 1], out[:, 2]


MODES = os.environ.get("SPLIT_MODES", "f16x2,f16x2s,bf16x3,bf16x2").split(",")


def main():
    names = sys.argv[1:] or ["default2017", "crossdock_default2018", "dense"]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for name in names:
        blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
        rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
        rec_xyz, rec_smt, lig_smt, poses = synth.make_complex(
            1, synth.mapped_types(rmap[0]), synth.mapped_types(lmap[0]), n_rec=2500, n_lig=32, n_poses=4)
        grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None, blob.resolution,
                                              blob.dimension, blob.radius_scaling)[0] for b in range(4)])
        with torch.no_grad():
            p0, a0 = forward(blob, grids, "f32")
            print(f"{name}: pose {p0.numpy().round(5)} affinity {a0.numpy().round(4)}")
            for mode in MODES:
                p, a = forward(blob, grids, mode)
                print(f"  {mode:12s} max|d pose| {float((p - p0).abs().max()):.3e}   max|d affinity| {float((a - a0).abs().max()):.3e}")


if __name__ == "__main__":
    main()
