"""Three small architectures that are NOT among gnina's shipped families -- what `--cnn_model file.pt` may hand to TorchModel
(gninasrc/lib/torch_model.cpp:49-118) -- for the generic TorchScript path of gnina_amd/tools/extract_weights.py.
`Stack`: avg pool, 3x3x3 / 1x1x1 convolutions with channel counts the shipped models do not have, a BatchNorm BEHIND a
convolution (folded), two pools, fixed-size heads.  `MiniDense`: a two-layer DenseNet block (BatchNorm in FRONT of its
convolutions, concatenation), a 1x1x1 transition, a global max pool, size-independent heads.  `PostAct`: BatchNorm BEHIND a ReLU (conv -> ReLU -> BN -> conv) and behind
a pool -- not foldable into the producing conv (the converter once folded it anyway and mis-scored silently)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Stack(nn.Module):
    def __init__(self, cin=28):
        super().__init__()
        self.c1 = nn.Conv3d(cin, 24, 3, padding=1)
        self.c1b = nn.Conv3d(24, 24, 1)
        self.c2 = nn.Conv3d(24, 40, 3, padding=1)
        self.bn2 = nn.BatchNorm3d(40)
        self.c3 = nn.Conv3d(40, 48, 3, padding=1)
        self.pose = nn.Linear(10368, 2)
        self.aff = nn.Linear(10368, 1)

    def forward(self, x):
        x = F.avg_pool3d(x, 2)
        x = F.relu(self.c1(x))
        x = F.relu(self.c1b(x))
        x = F.max_pool3d(x, 2)
        x = F.relu(self.bn2(self.c2(x)))
        x = F.max_pool3d(x, 2)
        x = F.relu(self.c3(x))
        x = x.view(-1, 10368)
        return F.log_softmax(self.pose(x), dim=1), self.aff(x).squeeze(-1)


class MiniDense(nn.Module):
    def __init__(self, cin=28):
        super().__init__()
        self.init = nn.Conv3d(cin, 16, 3, padding=1)
        self.bn0 = nn.BatchNorm3d(16)
        self.d0 = nn.Conv3d(16, 8, 3, padding=1)
        self.bn1 = nn.BatchNorm3d(24)
        self.d1 = nn.Conv3d(24, 8, 3, padding=1)
        self.trans = nn.Conv3d(32, 32, 1)
        self.pose = nn.Linear(32, 2)
        self.aff = nn.Linear(32, 1)

    def forward(self, x):
        x = F.max_pool3d(x, 2)
        a = F.relu(self.init(x))
        b = F.relu(self.d0(self.bn0(a)))
        ab = torch.cat([a, b], 1)
        c = F.relu(self.d1(self.bn1(ab)))
        abc = torch.cat([a, b, c], 1)
        t = F.relu(self.trans(abc))
        t = F.max_pool3d(t, 2)
        g = F.max_pool3d(t, 12)
        g = g.view(-1, 32)
        return F.log_softmax(self.pose(g), dim=1), self.aff(g).squeeze(-1)


class PostAct(nn.Module):
    """conv -> ReLU -> BatchNorm -> conv: the BatchNorm sits BEHIND the activation, so it cannot be folded into the conv in
    front of it (relu(alpha y + shift) != alpha relu(y) + shift); it is the next conv's input transform."""

    def __init__(self, cin=28):
        super().__init__()
        self.c1 = nn.Conv3d(cin, 16, 3, padding=1)
        self.bn1 = nn.BatchNorm3d(16)
        self.c2 = nn.Conv3d(16, 24, 3, padding=1)
        self.bn2 = nn.BatchNorm3d(24)
        self.c3 = nn.Conv3d(24, 8, 1)
        self.pose = nn.Linear(1728, 2)
        self.aff = nn.Linear(1728, 1)

    def forward(self, x):
        x = F.max_pool3d(x, 2)
        x = self.bn1(F.relu(self.c1(x)))
        x = F.max_pool3d(F.relu(self.c2(x)), 2)
        x = F.relu(self.c3(self.bn2(x)))
        x = F.max_pool3d(x, 2)
        x = x.view(-1, 1728)
        return F.log_softmax(self.pose(x), dim=1), self.aff(x).squeeze(-1)


def make(kind, seed=0):
    """the module in eval mode with seeded weights and non-trivial BatchNorm statistics"""
    torch.manual_seed(seed)
    m = {"stack": Stack, "minidense": MiniDense, "postact": PostAct}[kind]()
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm3d):
            with torch.no_grad():
                mod.running_mean.uniform_(-0.3, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    with torch.no_grad():  # heads with some gain: default-initialised ones leave every pose within 1e-3 of the others
        m.pose.weight.mul_(40.0)
        m.aff.weight.mul_(40.0)
    return m.eval()


def save_scripted(kind, path, seed=0):
    import json
    m = torch.jit.script(make(kind, seed))
    meta = {"resolution": 0.5, "dimension": 23.5}
    torch.jit.save(m, path, _extra_files={"metadata": json.dumps(meta)})
    return m
