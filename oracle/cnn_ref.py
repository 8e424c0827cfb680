"""CPU oracle for the CNN forward of gnina's TorchScript models -- TEST INFRASTRUCTURE ONLY.

Executes the layer program of a MIGNINA1 blob (gnina_amd/tools/extract_weights.py) with plain
PyTorch fp32 (or fp64) CPU ops in the reference's own NCDHW layout.  It restates what
`module.forward` does at gninasrc/lib/torch_model.cpp:185 for the three shipped families
 and the score post-processing of torch_model.cpp:188-195:
    pose  = softmax(log_softmax(logits))[1]  (== softmax(logits)[1])
    aff   = affinity head
    loss  = cross_entropy(log_softmax(logits), label 1) = -log_softmax(log_softmax(z))[1]
Pinned by tests/test_oracle_cnn_golden.py against the reference's own TorchScript files
(torch.jit.load of /root/reference/gninasrc/lib/models/*.pt) and the committed fixtures
tests/golden/cnn_goldens.npz generated from them.
"""
import struct

import numpy as np
import torch
import torch.nn.functional as F

MAGIC = b"MIGNINA1"


class Blob:
    def __init__(self, path_or_bytes):
        raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        assert raw[:8] == MAGIC, "not a MIGNINA1 blob"
        (hl,) = struct.unpack("<I", raw[8:12])
        self.header = raw[12:12 + hl].decode()
        off = 12 + hl
        off += (-off) % 64
        self.data = np.frombuffer(raw, dtype="<f4", offset=off)
        self.meta = {}
        self.recmap, self.ligmap, self.ops, self.bufs = [], [], [], {}
        for line in self.header.split("\n"):
            t = line.split()
            if not t:
                continue
            k = t[0]
            if k == "recmap":
                self.recmap.append(t[1:])
            elif k == "ligmap":
                self.ligmap.append(t[1:])
            elif k == "buf":
                self.bufs[int(t[1])] = (int(t[2]), int(t[3]))
            elif k in ("pool", "conv", "gmax", "fc", "overlap"):
                self.ops.append(t)
            else:
                self.meta[k] = t[1] if len(t) > 1 else ""
        assert int(self.meta["ndata"]) == self.data.size
        self.resolution = float(self.meta["resolution"])
        self.dimension = float(self.meta["dimension"])
        self.radius_scaling = float(self.meta["radius_scaling"])
        self.skip_softmax = bool(int(self.meta["skip_softmax"]))
        self.apply_logistic_loss = bool(int(self.meta["apply_logistic_loss"]))
        self.n_rec_ch, self.n_lig_ch = len(self.recmap), len(self.ligmap)
        self.family = self.meta.get("family", "")

    def recmap_text(self):
        return "\n".join(" ".join(l) for l in self.recmap) + "\n"

    def ligmap_text(self):
        return "\n".join(" ".join(l) for l in self.ligmap) + "\n"

    def tensor(self, off, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(self.data[off:off + n].reshape(shape).copy())


def forward_logits(blob, grid, dtype=torch.float32):
    """grid: [B, C, N, N, N] (reference layout). Returns (logits [B,2], affinity [B])."""
    x = torch.as_tensor(grid).to(dtype)
    bufs = {0: x}
    out = None
    for t in blob.ops:
        if t[0] == "pool":
            src, dst = int(t[2]), int(t[3])
            y = F.max_pool3d(bufs[src], 2, 2) if t[1] == "max" else F.avg_pool3d(bufs[src], 2, 2)
            bufs[dst] = y  # concat buffers grow by torch.cat below
        elif t[0] == "conv":
            k, src, dst, cin, cout, c0, relu = (int(v) for v in t[1:8])
            w_off, b_off, s_off, t_off = (int(v) for v in t[8:12])
            w = blob.tensor(w_off, (k * k * k, cin, cout)).to(dtype)
            w = w.reshape(k, k, k, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
            b = blob.tensor(b_off, (cout,)).to(dtype)
            xin = bufs[src][:, :cin]
            if s_off >= 0:
                s = blob.tensor(s_off, (cin,)).to(dtype).view(1, -1, 1, 1, 1)
                sh = blob.tensor(t_off, (cin,)).to(dtype).view(1, -1, 1, 1, 1)
                xin = xin * s + sh
            y = F.conv3d(xin, w, b, padding=k // 2)
            if relu:
                y = torch.relu(y)
            if src == dst:  # dense block: concat in place
                assert c0 == bufs[src].shape[1]
                bufs[dst] = torch.cat([bufs[src], y], 1)
            else:
                assert c0 == 0
                bufs[dst] = y
        elif t[0] == "overlap":  # test/gnina/data/overlap*.pt (toy model of test_min.py): returns the module's
            # output itself, not logits -- hstack([0, where(mean(rec * lig) > 0, mean, 1e-20)]) and a zero affinity
            src = int(t[1])
            g = bufs[src]
            ave = F.avg_pool3d(g[:, 0:1] * g[:, 1:2], g.shape[-1]).flatten(1)
            ave0 = torch.where(ave > 0, ave, torch.full_like(ave, 9.9999999999999995e-21))
            return torch.cat([torch.zeros_like(ave0), ave0], 1), torch.zeros(g.shape[0], dtype=dtype)
        elif t[0] == "gmax":
            src, dst = int(t[1]), int(t[2])
            bufs[dst] = torch.amax(bufs[src], dim=(2, 3, 4), keepdim=True)
        elif t[0] == "fc":
            src, n_in, w_off, b_off = (int(v) for v in t[1:5])
            xin = bufs[src]
            B, C = xin.shape[0], xin.shape[1]
            S3 = n_in // C
            w = blob.tensor(w_off, (3, S3, C)).to(dtype)
            b = blob.tensor(b_off, (3,)).to(dtype)
            flat = xin.reshape(B, C, S3).permute(0, 2, 1).reshape(B, S3 * C)
            out = flat @ w.reshape(3, S3 * C).t() + b
    return out[:, :2], out[:, 2]


def module_output(blob, grid, dtype=torch.float32):
    """What the TorchScript module returns: (log_softmax(pose logits), affinity) for the CNN families; the
    Overlap toy returns its [0, ave] tensor as is."""
    out, aff = forward_logits(blob, grid, dtype)
    if blob.family == "Overlap":
        return out, aff
    return torch.log_softmax(out, 1), aff


def scores(blob, grid, dtype=torch.float32):
    """(pose, affinity, loss) per pose exactly as TorchModel::forward reports them."""
    logp, aff = module_output(blob, grid, dtype)
    if blob.skip_softmax:
        pose = logp[:, 1]
    else:
        pose = torch.softmax(logp, 1)[:, 1]
    if blob.apply_logistic_loss:
        loss = -torch.log(logp[:, 1])
    else:
        loss = F.cross_entropy(logp, torch.ones(logp.shape[0], dtype=torch.long), reduction="none")
    return pose, aff, loss


def loss_and_grid_gradient(blob, grid, dtype=torch.float32):
    """What TorchModel::forward does with compute_gradient (torch_model.cpp:192-199): loss =
    cross_entropy(log_softmax(z), 1); loss.backward(); returns (loss [B], d loss / d grid [B,C,N,N,N])."""
    x = torch.as_tensor(grid).to(dtype).clone().requires_grad_(True)
    logp, _ = module_output(blob, x, dtype)
    if blob.apply_logistic_loss:
        loss = -torch.log(logp[:, 1])
    else:
        loss = F.cross_entropy(logp, torch.ones(logp.shape[0], dtype=torch.long), reduction="none")
    loss.sum().backward()
    return loss.detach(), x.grad.detach()
