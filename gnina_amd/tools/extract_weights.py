#!/usr/bin/env python3
"""TorchScript gnina model (.pt) -> MIGNINA1 weight blob (.mgw).

gnina embeds TorchScript zips in its binary (gninasrc/CMakeLists.txt:95-188,
gninasrc/lib/make_model_cpp.py:25-39) and loads them with torch::jit::load together with a JSON
"metadata" extra file (gninasrc/lib/torch_model.cpp:49-106).  The MI355X engine does not link
libtorch; it executes a small layer program over raw fp32 tensors.  This tool walks the
state_dict of the three shipped families and writes:

    "MIGNINA1" | u32 header_len | header text | pad to 64 B | fp32 data (little endian)

Header = one `key value...` record per line:
    name / family / resolution / dimension / radius_scaling / skip_softmax / apply_logistic_loss
    recmap <names of one channel>      (one line per channel, FileMappedGninaTyper order)
    ligmap <names of one channel>
    buf  <id> <S> <C>                  activation buffer [B][S][S][S][C], channels last
    pool <max|avg> <src> <dst>         2x2x2 stride 2
    conv <k> <src> <dst> <cin> <cout> <dst_c0> <relu> <w_off> <b_off> <bn_scale_off> <bn_shift_off>
         reads channels [0,cin) of src, writes [dst_c0,dst_c0+cout) of dst; weights [k^3][cin][cout]
         (tap index = (kx*3+ky)*3+kz, x = slowest grid axis); offsets in floats, -1 = absent.
         eval-mode BatchNorm is folded to y = scale*x + shift applied to the conv INPUT before
         zero padding (SURVEY "Hard parts").
    gmax <src> <dst>                   global max pool over space (Dense family)
    fc   <src> <n_in> <w_off> <b_off>  3 outputs (pose logit 0, pose logit 1, affinity);
         weights [3][S^3][C] in the channels-last flatten order of src.
    ndata <floats>

Usage: extract_weights.py model.pt [out.mgw]      (or --all <models_dir> <out_dir>)
"""
import json
import os
import struct
import sys

import numpy as np
import torch

MAGIC = b"MIGNINA1"

DEFAULT_RECMAP = """AliphaticCarbonXSHydrophobe
AliphaticCarbonXSNonHydrophobe
AromaticCarbonXSHydrophobe
AromaticCarbonXSNonHydrophobe
Bromine Iodine Chlorine Fluorine
Nitrogen NitrogenXSAcceptor
NitrogenXSDonor NitrogenXSDonorAcceptor
Oxygen OxygenXSAcceptor
OxygenXSDonorAcceptor OxygenXSDonor
Sulfur SulfurAcceptor
Phosphorus
Calcium
Zinc
GenericMetal Boron Manganese Magnesium Iron
"""  # torch_model.cpp:16-30 (used when the .pt carries no metadata)

DEFAULT_LIGMAP = """AliphaticCarbonXSHydrophobe
AliphaticCarbonXSNonHydrophobe
AromaticCarbonXSHydrophobe
AromaticCarbonXSNonHydrophobe
Bromine Iodine
Chlorine
Fluorine
Nitrogen NitrogenXSAcceptor
NitrogenXSDonor NitrogenXSDonorAcceptor
Oxygen OxygenXSAcceptor
OxygenXSDonorAcceptor OxygenXSDonor
Sulfur SulfurAcceptor
Phosphorus
GenericMetal Boron Manganese Magnesium Zinc Calcium Iron
"""  # torch_model.cpp:32-46


class Blob:
    def __init__(self):
        self.lines = []
        self.data = []
        self.n = 0
        self.nbuf = 0

    def add(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        off = self.n
        self.data.append(arr)
        self.n += arr.size
        pad = (-self.n) % 16  # keep every tensor 64-byte aligned
        if pad:
            self.data.append(np.zeros(pad, dtype=np.float32))
            self.n += pad
        return off

    def buf(self, S, C):
        i = self.nbuf
        self.nbuf += 1
        self.lines.append(f"buf {i} {S} {C}")
        return i

    def conv(self, sd, prefix, src, dst, dst_c0, relu, bn_prefix=None):
        w = sd[prefix + ".weight"].double()
        b = sd[prefix + ".bias"]
        co, ci, k = w.shape[0], w.shape[1], w.shape[2]
        # [co][ci][kx][ky][kz] -> [tap][ci][co]
        wt = w.permute(2, 3, 4, 1, 0).reshape(k * k * k, ci, co).float().numpy()
        w_off = self.add(wt)
        b_off = self.add(b.numpy())
        s_off = t_off = -1
        if bn_prefix is not None:
            g, beta = sd[bn_prefix + ".weight"].float(), sd[bn_prefix + ".bias"].float()
            mean, var = sd[bn_prefix + ".running_mean"].float(), sd[bn_prefix + ".running_var"].float()
            # ATen eval batch norm (fp32): alpha = w / sqrt(var + eps); beta' = b - mean * alpha
            alpha = g / torch.sqrt(var + 1e-5)
            shift = beta - mean * alpha
            s_off = self.add(alpha.numpy())
            t_off = self.add(shift.numpy())
        self.lines.append(f"conv {k} {src} {dst} {ci} {co} {dst_c0} {int(relu)} {w_off} {b_off} {s_off} {t_off}")
        return ci, co

    def fc(self, sd, pose_key, aff_key, src, S, C):
        wp, bp = sd[pose_key + ".weight"], sd[pose_key + ".bias"]
        wa, ba = sd[aff_key + ".weight"], sd[aff_key + ".bias"]
        w = torch.cat([wp, wa], 0)  # [3][C*S^3], flat = c*S^3 + (x*S+y)*S+z  (view(-1, C*S^3) of NCDHW)
        w = w.reshape(3, C, S * S * S).permute(0, 2, 1).contiguous()  # -> [3][S^3][C]
        w_off = self.add(w.numpy())
        b_off = self.add(torch.cat([bp, ba]).numpy())
        self.lines.append(f"fc {src} {S * S * S * C} {w_off} {b_off}")


def _maps_to_lines(text, key):
    out = []
    for line in text.split("\n"):
        names = line.split()
        if names:
            out.append(key + " " + " ".join(names))
    return out


def convert(pt_path, name=None):
    extra = {"metadata": ""}
    m = torch.jit.load(pt_path, map_location="cpu", _extra_files=extra)
    md = extra["metadata"]
    md = md.decode() if isinstance(md, bytes) else md
    meta = json.loads(md) if md else {}
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    name = name or os.path.splitext(os.path.basename(pt_path))[0].replace(".", "_")  # make_model_cpp.py:27-29
    family = m.original_name
    res = float(meta.get("resolution", 0.5))
    dim = float(meta.get("dimension", 23.5))
    N = int(round(dim / res)) + 1
    blob = Blob()
    L = blob.lines
    L.append(f"name {name}")
    L.append(f"family {family}")
    L.append(f"resolution {res!r}")
    L.append(f"dimension {dim!r}")
    L.append(f"radius_scaling {float(meta.get('radius_scaling', 1.0))!r}")
    L.append(f"skip_softmax {int(bool(meta.get('skip_softmax', False)))}")
    L.append(f"apply_logistic_loss {int(bool(meta.get('apply_logistic_loss', False)))}")
    recmap = meta.get("recmap", DEFAULT_RECMAP)
    ligmap = meta.get("ligmap", DEFAULT_LIGMAP)
    rl, ll = _maps_to_lines(recmap, "recmap"), _maps_to_lines(ligmap, "ligmap")
    L.extend(rl)
    L.extend(ll)
    C0 = len(rl) + len(ll)
    S = N // 2

    keys = list(sd.keys())
    if any(k.endswith("unit1_conv1.weight") for k in keys):  # Default2017Affinity
        p = "features."
        assert sd[p + "unit1_conv1.weight"].shape[1] == C0
        b_in = blob.buf(N, C0)
        b0 = blob.buf(S, C0); L.append(f"pool max {b_in} {b0}")
        b1 = blob.buf(S, 32); blob.conv(sd, p + "unit1_conv1", b0, b1, 0, True)
        b2 = blob.buf(S // 2, 32); L.append(f"pool max {b1} {b2}")
        b3 = blob.buf(S // 2, 64); blob.conv(sd, p + "unit2_conv1", b2, b3, 0, True)
        b4 = blob.buf(S // 4, 64); L.append(f"pool max {b3} {b4}")
        b5 = blob.buf(S // 4, 128); blob.conv(sd, p + "unit3_conv1", b4, b5, 0, True)
        blob.fc(sd, "pose.pose_output", "affinity.affinity_output", b5, S // 4, 128)
    elif any(k.endswith("unit1_conv.weight") for k in keys):  # Default2018Affinity / Net
        p = "features." if any(k.startswith("features.") for k in keys) else ""
        assert sd[p + "unit1_conv.weight"].shape[1] == C0
        b_in = blob.buf(N, C0)
        b0 = blob.buf(S, C0); L.append(f"pool avg {b_in} {b0}")
        b1 = blob.buf(S, 32); blob.conv(sd, p + "unit1_conv", b0, b1, 0, True)
        b2 = blob.buf(S, 32); blob.conv(sd, p + "unit2_conv", b1, b2, 0, True)
        b3 = blob.buf(S // 2, 32); L.append(f"pool avg {b2} {b3}")
        b4 = blob.buf(S // 2, 64); blob.conv(sd, p + "unit3_conv", b3, b4, 0, True)
        b5 = blob.buf(S // 2, 64); blob.conv(sd, p + "unit4_conv", b4, b5, 0, True)
        b6 = blob.buf(S // 4, 64); L.append(f"pool avg {b5} {b6}")
        b7 = blob.buf(S // 4, 128); blob.conv(sd, p + "unit5_conv", b6, b7, 0, True)
        pose = [k for k in keys if k.endswith("pose_output.weight")][0][:-7]
        aff = [k for k in keys if k.endswith("affinity_output.weight")][0][:-7]
        blob.fc(sd, pose, aff, b7, S // 4, 128)
    elif any("dense_block_0" in k for k in keys):  # DenseAffinity / Dense
        p = "features." if any(k.startswith("features.") for k in keys) else ""
        bp = "blocks." if any(".blocks." in k for k in keys) else ""
        assert sd[p + "data_enc_init_conv.weight"].shape[1] == C0
        b_in = blob.buf(N, C0)
        b0 = blob.buf(S, C0); L.append(f"pool max {b_in} {b0}")
        cur_S = S
        cat = blob.buf(cur_S, 32 + 64)  # dense block 0 concat buffer: 32 + 4*16
        blob.conv(sd, p + "data_enc_init_conv", b0, cat, 0, True)
        cin = 32
        for lvl in range(3):
            for j in range(4):
                pre = f"{p}dense_block_{lvl}.{bp}data_enc_level{lvl}_"
                ci, co = blob.conv(sd, pre + f"conv{j}", cat, cat, cin, True, bn_prefix=pre + f"batchnorm_conv{j}")
                assert ci == cin and co == 16
                cin += 16
            if lvl < 2:
                bott = blob.buf(cur_S, cin)
                blob.conv(sd, f"{p}data_enc_level{lvl}_bottleneck", cat, bott, 0, True)
                cur_S //= 2
                cat = blob.buf(cur_S, cin + 64)
                L.append(f"pool max {bott} {cat}")
        g = blob.buf(1, cin); L.append(f"gmax {cat} {g}")
        pose = [k for k in keys if k.endswith("pose_output.weight")][0][:-7]
        aff = [k for k in keys if k.endswith("affinity_output.weight")][0][:-7]
        blob.fc(sd, pose, aff, g, 1, cin)
    elif family == "Overlap" and not keys:  # test toy: mean(rec * lig) over the grid, no parameters
        assert C0 == 2
        b_in = blob.buf(N, C0)
        L.append(f"overlap {b_in}")
    else:
        raise ValueError(f"unsupported model family {family} in {pt_path}")
    L.append(f"ndata {blob.n}")
    header = ("\n".join(L) + "\n").encode()
    pre = MAGIC + struct.pack("<I", len(header)) + header
    pre += b"\0" * ((-len(pre)) % 64)
    payload = np.concatenate(blob.data).astype("<f4").tobytes() if blob.data else b""
    return pre + payload, name


def main(argv):
    if len(argv) >= 3 and argv[0] == "--all":
        os.makedirs(argv[2], exist_ok=True)
        for f in sorted(os.listdir(argv[1])):
            if f.endswith(".pt"):
                data, name = convert(os.path.join(argv[1], f))
                open(os.path.join(argv[2], name + ".mgw"), "wb").write(data)
                print(name, len(data))
        return
    data, name = convert(argv[0])
    out = argv[1] if len(argv) > 1 else name + ".mgw"
    open(out, "wb").write(data)
    print("wrote", out, len(data), "bytes")


if __name__ == "__main__":
    main(sys.argv[1:])
