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


# ----------------------------------------------------------------------------------------------
# Generic path: any TorchScript module whose frozen graph is a feed-forward stack of the operators gnina's
# architectures are made of (what `--cnn_model file.pt` may hand to TorchModel, torch_model.cpp:49-118):
#   max_pool3d / avg_pool3d (kernel = stride = 2, or kernel = the whole grid: global max pool), _convolution / conv3d
#   (kernel 1 or 3, stride 1, "same" padding, no groups / dilation), relu behind a convolution, batch_norm in eval mode
#   (in front of a convolution: kept as the conv's input scale / shift; behind one: folded into its weights), cat along
#   the channels of tensors that extend one another (DenseNet blocks: cat([a, b]), cat([a, b, c]), ...), view / flatten,
#   two linear heads on the flattened features (pose: 2 outputs, affinity: 1), log_softmax / squeeze on the way out.
# The walk assigns every tensor a (buffer, first channel, channels, grid points per side) and emits the same program
# lines the family-specific paths above write by hand.  Anything else raises, naming the operator.
# ----------------------------------------------------------------------------------------------
def _const(v):
    n = v.node()
    if n.kind() != "prim::Constant":
        raise ValueError(f"expected a constant, got {n.kind()}")
    return v.toIValue()


def _ints(v):
    n = v.node()
    if n.kind() == "prim::ListConstruct":
        return [int(_const(i)) for i in n.inputs()]
    x = _const(v)
    return [int(i) for i in x] if isinstance(x, (list, tuple)) else [int(x)]


def convert_graph(module, blob, C0, N):
    """Walk the frozen graph of `module` (input: the [B, C0, N, N, N] grid) and append its program to `blob`."""
    frozen = torch.jit.freeze(module.eval())
    g = frozen.graph
    # a SCRIPTED module carries its Python control flow (BatchNorm's `if input.dim() != 5: raise ...`): with the input's
    # shape known the checks fold away and what is left is the operator stack a traced module has from the start
    torch._C._jit_pass_inline(g)
    torch._C._jit_pass_complete_shape_analysis(g, (torch.zeros(1, C0, N, N, N),), False)
    torch._C._jit_pass_peephole(g, False)
    torch._C._jit_pass_constant_propagation(g)
    torch._C._jit_pass_dce(g)
    L = blob.lines
    inputs = list(g.inputs())
    x_in = inputs[-1]
    b_in = blob.buf(N, C0)
    # tensor value -> dict(buf, c0, C, S, kind, node) ; kind: "grid" | "act" | "flat" | "pose" | "aff"
    T = {x_in.debugName(): dict(buf=b_in, c0=0, C=C0, S=N, kind="grid")}
    nodes = list(g.nodes())
    uses = {}
    for nd in nodes:
        for i in nd.inputs():
            uses.setdefault(i.debugName(), []).append(nd)

    # concat groups: a tensor that is the FIRST member of cats gets a buffer wide enough for the longest of them
    cat_total = {}   # first member name -> total channels (filled lazily once channel counts are known)
    cat_members = {}  # member name -> (first member name, index in the longest list)
    longest = {}
    for nd in nodes:
        if nd.kind() == "aten::cat":
            lst = nd.inputsAt(0).node()
            if lst.kind() != "prim::ListConstruct" or int(_const(nd.inputsAt(1))) != 1:
                raise ValueError("cat: only lists of tensors along the channel dimension")
            names = [i.debugName() for i in lst.inputs()]
            first = names[0]
            if first not in longest or len(names) > len(longest[first]):
                if first in longest and longest[first] != names[:len(longest[first])]:
                    raise ValueError("cat: lists starting at one tensor must extend one another")
                longest[first] = names
            elif names != longest[first][:len(names)]:
                raise ValueError("cat: lists starting at one tensor must extend one another")
    for first, names in longest.items():
        for k, n in enumerate(names):
            cat_members[n] = (first, k)

    pending_bn = {}  # tensor name (BN output) -> (source descriptor, scale, shift)
    conv_lines = {}  # conv output name -> index into L (to set its relu flag) and folded-BN bookkeeping
    out_pose = out_aff = None
    head = {}
    log_softmax_seen = False

    def conv_out_channels(nd):
        return nd.inputsAt(1).toIValue().shape[0]

    def channels_of(name, seen=None):
        """channels of a tensor that may not have been produced yet (needed to size a concat buffer up front)"""
        if name in T:
            return T[name]["C"]
        for nd in nodes:
            for o in nd.outputs():
                if o.debugName() == name:
                    k = nd.kind()
                    if k in ("aten::_convolution", "aten::conv3d"):
                        return conv_out_channels(nd)
                    if k in ("aten::relu", "aten::relu_", "aten::batch_norm", "aten::max_pool3d", "aten::avg_pool3d"):
                        return channels_of(nd.inputsAt(0).debugName())
                    if k == "aten::cat":
                        return sum(channels_of(i.debugName()) for i in nd.inputsAt(0).node().inputs())
        raise ValueError(f"cannot size tensor {name}")

    def place_output(name, C, S):
        """(buf, c0) where a freshly produced [C] x S^3 tensor must live: inside its concat buffer, or a buffer of its own"""
        if name in cat_members:
            first, k = cat_members[name]
            names = longest[first]
            if k == 0:
                total = sum(channels_of(n) for n in names)
                return blob.buf(S, total), 0
            base = T[first]
            c0 = sum(T[n]["C"] for n in names[:k])
            if base["S"] != S:
                raise ValueError("cat: members of different grid sizes")
            return base["buf"], c0
        return blob.buf(S, C), 0

    for nd in nodes:
        k = nd.kind()
        outs = [o.debugName() for o in nd.outputs()]
        if k in ("prim::Constant", "prim::ListConstruct"):
            continue
        if k in ("aten::max_pool3d", "aten::avg_pool3d"):
            src = T[nd.inputsAt(0).debugName()]
            ks, st = _ints(nd.inputsAt(1)), _ints(nd.inputsAt(2))
            st = st or ks
            pad = _ints(nd.inputsAt(3))
            if any(pad) or len(set(ks)) != 1 or ks != st:
                raise ValueError(f"{k}: kernel = stride, no padding, cubic")
            mode = "max" if k == "aten::max_pool3d" else "avg"
            if ks[0] == 2:
                S = src["S"] // 2
                buf, c0 = place_output(outs[0], src["C"], S)
                if c0 != 0 or src["c0"] != 0:
                    raise ValueError("pool: operands must start at channel 0 of their buffers")
                L.append(f"pool {mode} {src['buf']} {buf}")
                T[outs[0]] = dict(buf=buf, c0=0, C=src["C"], S=S, kind="act")
            elif ks[0] == src["S"] and mode == "max":
                buf = blob.buf(1, src["C"])
                L.append(f"gmax {src['buf']} {buf}")
                T[outs[0]] = dict(buf=buf, c0=0, C=src["C"], S=1, kind="act")
            else:
                raise ValueError(f"{k} with kernel {ks} on a grid of {src['S']}")
        elif k == "aten::adaptive_max_pool3d":
            src = T[nd.inputsAt(0).debugName()]
            if _ints(nd.inputsAt(1)) != [1, 1, 1]:
                raise ValueError("adaptive_max_pool3d: output size 1 only")
            buf = blob.buf(1, src["C"])
            L.append(f"gmax {src['buf']} {buf}")
            T[outs[0]] = dict(buf=buf, c0=0, C=src["C"], S=1, kind="act")
        elif k == "aten::batch_norm":
            src_name = nd.inputsAt(0).debugName()
            w, b, mean, var = (nd.inputsAt(i).toIValue() for i in (1, 2, 3, 4))
            training, eps = bool(_const(nd.inputsAt(5))), float(_const(nd.inputsAt(7)))
            if training:
                raise ValueError("batch_norm in training mode")
            alpha = (w.float() if w is not None else torch.ones_like(var.float())) / torch.sqrt(var.float() + eps)
            shift = (b.float() if b is not None else torch.zeros_like(var.float())) - mean.float() * alpha
            foldable = (src_name in conv_lines and len(uses.get(src_name, [])) == 1 and not conv_lines[src_name]["relu"] and
                        "fold" not in conv_lines[src_name])
            if foldable:  # conv -> BN (no ReLU in between, one BatchNorm only): fold into the conv
                conv_lines[src_name]["fold"] = (alpha, shift)
                T[outs[0]] = T[src_name]
                conv_lines[outs[0]] = conv_lines[src_name]
            else:  # BN -> conv (conv -> ReLU -> BN -> conv included): the NEXT conv's input scale / shift
                if any(u.kind() not in ("aten::_convolution", "aten::conv3d") for u in uses.get(outs[0], [])) or not uses.get(outs[0]):
                    raise ValueError("batch_norm: behind a ReLU (or a second BatchNorm) its output must feed convolutions only")
                pending_bn[outs[0]] = (T[src_name], alpha, shift)
        elif k in ("aten::_convolution", "aten::conv3d"):
            in_name = nd.inputsAt(0).debugName()
            bn = pending_bn.get(in_name)
            src = bn[0] if bn else T[in_name]
            w = nd.inputsAt(1).toIValue()
            bias = nd.inputsAt(2).toIValue()
            stride, padding, dilation = _ints(nd.inputsAt(3)), _ints(nd.inputsAt(4)), _ints(nd.inputsAt(5))
            if k == "aten::_convolution":
                transposed, groups = bool(_const(nd.inputsAt(6))), int(_const(nd.inputsAt(8)))
            else:
                transposed, groups = False, int(_const(nd.inputsAt(6)))
            co, ci, kk = w.shape[0], w.shape[1], w.shape[2]
            if (transposed or groups != 1 or set(stride) != {1} or set(dilation) != {1} or kk not in (1, 3) or
                    tuple(w.shape[2:]) != (kk, kk, kk) or set(padding) != {kk // 2}):
                raise ValueError("convolution: kernel 1 or 3, stride 1, 'same' padding, no groups / dilation / transposition")
            if src["c0"] != 0 or ci != src["C"]:
                raise ValueError("convolution: the input must be channels [0, C) of its buffer")
            # where the result lives is decided by the tensor the rest of the graph sees: the convolution's output, or what a
            # BatchNorm / ReLU that are its only consumers make of it
            final = outs[0]
            while len(uses.get(final, [])) == 1 and uses[final][0].kind() in ("aten::batch_norm", "aten::relu", "aten::relu_"):
                final = uses[final][0].output().debugName()
            buf, c0 = place_output(final, co, src["S"])
            conv_lines[outs[0]] = dict(w=w.double(), b=(bias.double() if bias is not None else torch.zeros(co, dtype=torch.float64)),
                                       src=src["buf"], dst=buf, c0=c0, relu=False, bn=(bn[1], bn[2]) if bn else None, k=kk,
                                       line=len(L))
            L.append(None)  # written once relu / a folded BatchNorm behind it are known
            T[outs[0]] = dict(buf=buf, c0=c0, C=co, S=src["S"], kind="act")
        elif k in ("aten::relu", "aten::relu_"):
            in_name = nd.inputsAt(0).debugName()
            if in_name not in conv_lines or len(uses.get(in_name, [])) != 1:
                raise ValueError("relu: only directly behind a convolution (or its folded BatchNorm)")
            conv_lines[in_name]["relu"] = True
            if in_name in cat_members:
                raise ValueError("cat of a pre-activation tensor")
            T[outs[0]] = T[in_name]
            conv_lines[outs[0]] = conv_lines[in_name]
        elif k == "aten::cat":
            names = [i.debugName() for i in nd.inputsAt(0).node().inputs()]
            first = T[names[0]]
            if first["c0"] != 0:
                raise ValueError("cat: the first member must start its buffer")
            c = 0
            for n in names:
                if T[n]["buf"] != first["buf"] or T[n]["c0"] != c:
                    raise ValueError("cat: members are not consecutive channel ranges of one buffer")
                c += T[n]["C"]
            T[outs[0]] = dict(buf=first["buf"], c0=0, C=c, S=first["S"], kind="act")
        elif k in ("aten::view", "aten::reshape", "aten::flatten"):
            src = T[nd.inputsAt(0).debugName()]
            if src["c0"] != 0:
                raise ValueError("flatten: of a whole buffer only")
            T[outs[0]] = dict(src, kind="flat")
        elif k == "aten::linear":
            src = T[nd.inputsAt(0).debugName()]
            if src["kind"] != "flat":
                raise ValueError("linear: on the flattened features only")
            w, b = nd.inputsAt(1).toIValue(), nd.inputsAt(2).toIValue()
            role = "pose" if w.shape[0] == 2 else "aff" if w.shape[0] == 1 else None
            if role is None or role in head or w.shape[1] != src["C"] * src["S"] ** 3:
                raise ValueError("linear: one head with 2 outputs (pose) and one with 1 (affinity) on the same features")
            head[role] = (w, b if b is not None else torch.zeros(w.shape[0]), src)
            T[outs[0]] = dict(buf=-1, c0=0, C=w.shape[0], S=0, kind=role)
        elif k in ("aten::log_softmax", "aten::squeeze", "aten::unsqueeze", "aten::dropout", "aten::contiguous"):
            src = T[nd.inputsAt(0).debugName()]
            if k == "aten::log_softmax":
                if src["kind"] != "pose":
                    raise ValueError("log_softmax: on the pose head only")
                log_softmax_seen = True
            T[outs[0]] = src
        elif k == "prim::TupleConstruct":
            a, b = (T[i.debugName()] for i in nd.inputs())
            if a["kind"] != "pose" or b["kind"] != "aff":
                raise ValueError("the module must return (pose logits [B, 2], affinity [B])")
            out_pose, out_aff = a, b
        else:
            raise ValueError(f"unsupported operator {k} in the model's graph")
    if out_pose is None or set(head) != {"pose", "aff"} or head["pose"][2] is not head["aff"][2] and head["pose"][2] != head["aff"][2]:
        raise ValueError("the module must return (pose logits [B, 2], affinity [B]) from two linear heads on the same features")
    # the convolutions, now that relu flags and folded BatchNorms are known
    done = set()
    for name, cl in conv_lines.items():
        if cl["line"] in done:
            continue
        done.add(cl["line"])
        w, b = cl["w"], cl["b"]
        if "fold" in cl:  # conv -> BN: W' = alpha[co] W, b' = alpha b + shift
            alpha, shift = cl["fold"]
            w = w * alpha.double().view(-1, 1, 1, 1, 1)
            b = b * alpha.double() + shift.double()
        co, ci, kk = w.shape[0], w.shape[1], cl["k"]
        wt = w.permute(2, 3, 4, 1, 0).reshape(kk ** 3, ci, co).float().numpy()
        w_off = blob.add(wt)
        b_off = blob.add(b.float().numpy())
        s_off = t_off = -1
        if cl["bn"] is not None:
            s_off = blob.add(cl["bn"][0].numpy())
            t_off = blob.add(cl["bn"][1].numpy())
        L[cl["line"]] = f"conv {kk} {cl['src']} {cl['dst']} {ci} {co} {cl['c0']} {int(cl['relu'])} {w_off} {b_off} {s_off} {t_off}"
    wp, bp, src = head["pose"]
    wa, ba, _ = head["aff"]
    S, C = src["S"], src["C"]
    w = torch.cat([wp, wa], 0).reshape(3, C, S * S * S).permute(0, 2, 1).contiguous()
    w_off = blob.add(w.float().numpy())
    b_off = blob.add(torch.cat([bp.float().reshape(-1), ba.float().reshape(-1)]).numpy())
    L.append(f"fc {src['buf']} {S * S * S * C} {w_off} {b_off}")
    return log_softmax_seen


def convert(pt_path, name=None, generic=False):
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
    if generic:
        keys = ["<generic>"]
    if generic or not (any(k.endswith(("unit1_conv1.weight", "unit1_conv.weight")) for k in keys) or
                       any("dense_block_0" in k for k in keys) or (family == "Overlap" and not keys)):
        # an architecture this file has no hand-written path for (`--cnn_model file.pt`): walk its graph
        blob.lines.append("generic 1")
        log_softmax = convert_graph(m, blob, C0, N)
        if bool(meta.get("skip_softmax", False)) and not log_softmax:
            raise ValueError("skip_softmax on a module that returns raw logits is not supported")
    elif any(k.endswith("unit1_conv1.weight") for k in keys):  # Default2017Affinity
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
    if len(argv) >= 2 and argv[0] == "--generic":  # force the graph walk (also for the shipped families: a self-check)
        data, name = convert(argv[1], generic=True)
        out = argv[2] if len(argv) > 2 else name + ".mgw"
        open(out, "wb").write(data)
        print("wrote", out, len(data), "bytes")
        return
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
