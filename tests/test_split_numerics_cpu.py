"""The arithmetic claim behind the split-fp16 convolution kernels (gnina_amd/csrc/conv3d_h2.hip), checked on CPU.

Every fp32 operand is written as a = h + l with h = RN_fp16(a), l = RN_fp16(a - h); a product keeps h_a*h_w + h_a*l_w +
l_a*h_w.  Products of fp16 numbers are exact in fp32 and the sum is accumulated in fp32 on the device; here it is
accumulated in float64, so what is measured is the split's own error:
  * the three kept terms reproduce a 3x3x3 convolution to ~2^-22 of sum |a||w| (the dropped l*l term and the two
    roundings of the halves), i.e. fp32-accumulation grade -- while two bf16 halves, also three MFMAs, are 60x worse;
  * the weights need their per-layer power-of-two scale: without it the low halves of small weights are fp16 subnormals;
  * the loader's split (mi_debug_split_f16, host code of libmi_gnina.so) gives the same halves as this emulation;
  * gradients (no fixed range) keep that grade at any magnitude once scaled by the power of two their per-pose maximum
    selects -- the gradient-pass kernels' ConvArgs::in_amax.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _split(x, dt, scale=1.0):
    h = (x * scale).to(torch.float32).to(dt)
    l = ((x * scale).to(torch.float32) - h.to(torch.float32)).to(dt)
    return h.to(torch.float64), l.to(torch.float64)


def _conv_split(a, w, dt, sw=1.0):
    ah, al = _split(a, dt)
    wh, wl = _split(w, dt, sw)
    y = F.conv3d(ah, wh, padding=1) + F.conv3d(ah, wl, padding=1) + F.conv3d(al, wh, padding=1)
    return y / sw


@pytest.fixture(scope="module")
def layer():
    g = torch.Generator().manual_seed(7)
    a = torch.relu(torch.randn(2, 24, 10, 10, 10, generator=g)) * torch.rand(2, 24, 10, 10, 10, generator=g) * 3.0
    w = torch.randn(16, 24, 3, 3, 3, generator=g) * 0.03
    a, w = a.to(torch.float32).to(torch.float64), w.to(torch.float32).to(torch.float64)
    exact = F.conv3d(a, w, padding=1)
    mag = F.conv3d(a.abs(), w.abs(), padding=1)   # sum |a||w| per output: the scale rounding errors live on
    return a, w, exact, mag


def test_two_fp16_halves_and_three_products_are_fp32_grade(layer):
    a, w, exact, mag = layer
    sw = 2.0 ** (13 - np.floor(np.log2(float(w.abs().max()))))   # the loader's scale: max |w| * sw in [2^13, 2^14)
    err = ((_conv_split(a, w, torch.float16, sw) - exact).abs() / mag).max().item()
    assert err <= 2.0 ** -21, err
    # the same three MFMAs on bf16 halves keep 16 bits, not 22
    err_bf = ((_conv_split(a, w, torch.bfloat16) - exact).abs() / mag).max().item()
    assert err_bf > 30 * err
    # and fp32's own rounding of each product-sum is of the order the split costs: the split is not a precision class
    y32 = F.conv3d(a.float(), w.float(), padding=1).double()
    err32 = ((y32 - exact).abs() / mag).max().item()
    assert err <= 8 * err32 + 2.0 ** -22


def test_weights_need_their_power_of_two_scale(layer):
    a, w, exact, mag = layer
    tiny = w * 2.0 ** -6          # weights around 5e-4: their low halves are fp16 subnormals without the scale
    ex = F.conv3d(a, tiny, padding=1)
    mg = F.conv3d(a.abs(), tiny.abs(), padding=1)
    sw = 2.0 ** (13 - np.floor(np.log2(float(tiny.abs().max()))))
    scaled = ((_conv_split(a, tiny, torch.float16, sw) - ex).abs() / mg).max().item()
    unscaled = ((_conv_split(a, tiny, torch.float16, 1.0) - ex).abs() / mg).max().item()
    assert scaled <= 2.0 ** -21 and unscaled > 8 * scaled


def _amax_scale(amax):
    """the gradient-pass kernels' scale (conv3d_h2_kernel<..., BWD>: ConvArgs::in_amax): 2^(14 - exponent of the pose's
    largest |g|), from the float's bits exactly as the kernel takes it"""
    e = (np.float32(amax).view(np.uint32) >> 23) & 0xff
    if e == 0 or e == 255:
        return 1.0
    eb = max(4, min(250, 268 - int(e)))
    return float(np.uint32(eb << 23).view(np.float32))


@pytest.mark.parametrize("magnitude", [1e-9, 1e-4, 1.0, 3e4, 1e9])
def test_a_gradient_scaled_by_its_maximum_is_fp32_grade_at_any_magnitude(layer, magnitude):
    """A gradient tensor has no fixed range (1e-9 is below fp16's, 1e9 above it): the transposed convs stage
    g * 2^(14 - exponent(max |g|)) and un-scale the accumulators.  Whatever the magnitude, the largest staged value lands
    in [2^14, 2^15), nothing overflows, and the three-product split keeps its 2^-21 of sum |g||w| -- elements far below the
    maximum lose RELATIVE precision (absolute error <= 2^-25 staged = 2^-39 of the maximum), which a sum dominated by the
    large ones cannot see."""
    a, w, _, _ = layer
    g = torch.Generator().manual_seed(11)
    # a ReLU-masked gradient with eight decades of dynamic range inside one pose
    grad = torch.randn(a.shape, generator=g, dtype=torch.float64) * torch.exp(torch.rand(a.shape, generator=g, dtype=torch.float64) * -18.0)
    grad = (grad * (a > 0) * magnitude).to(torch.float32).to(torch.float64)
    s = _amax_scale(float(grad.abs().max()))
    top = float(grad.abs().max()) * s
    assert 2.0 ** 14 <= top < 2.0 ** 15
    sw = 2.0 ** (13 - np.floor(np.log2(float(w.abs().max()))))
    exact = F.conv3d(grad, w, padding=1)
    mag = F.conv3d(grad.abs(), w.abs(), padding=1)
    got = _conv_split(grad * s, w, torch.float16, sw) / s
    ok = mag > 0
    err = ((got - exact).abs()[ok] / mag[ok]).max().item()
    assert err <= 2.0 ** -21, (magnitude, err)
    # without the scale the same tensor is either flushed (tiny) or infinite (huge) in fp16
    if magnitude <= 1e-9:
        assert float(_split(grad, torch.float16)[0].abs().max()) == 0.0
    if magnitude >= 1e9:
        assert not torch.isfinite(_split(grad, torch.float16)[0]).all()


def test_the_loaders_split_is_this_split(layer):
    from gnina_amd import capi
    _, w, _, _ = layer
    w32 = w.float().numpy().ravel()
    hi, lo, sw = capi.split_f16(w32)
    h, l = _split(torch.from_numpy(w32).double(), torch.float16, sw)
    assert sw == 2.0 ** (13 - np.floor(np.log2(np.abs(w32).max())))
    assert np.array_equal(hi.astype(np.float64), h.numpy()) and np.array_equal(lo.astype(np.float64), l.numpy())


def test_lds_layout_of_the_planar_halo_tile_is_conflict_free_for_the_shipped_tiles():
    """conv3d_h2_kernel's A operands are ds_read_b128 of a planar [h plane | l plane] halo tile; the engine picks the M-tile
    geometry and the pad slots with a bank model of MI355X's LDS (a group of 16 lanes is served in as many cycles as the
    largest number of its lanes that hit one 16-byte slot modulo 16 at different addresses).  Restated here, lane groups
    from the microarchitecture guide; the shipped networks' tiles must come out conflict free (the 6^3 tile: 2.0, raster
    order over an odd tile is all its kernel shape offers), and the engine's pick must be what the restatement finds."""
    import ctypes as C
    from gnina_amd import capi
    L = capi.lib()
    L.mi_debug_h2_layout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]

    def cycles(tc, mt, py, px):
        tcx, tcy, tcz = tc
        nc = tcx * tcy * tcz
        hy, hz = 2 * tcy + 2, 2 * tcz + 2
        sy = hz + py
        sx = hy * sy + px
        n_mt = tcx // 4 * tcy * tcz if mt == 1 else tcz if mt == 2 else (nc + 3) // 4
        tot = 0
        for m in range(n_mt):
            base = []
            for row in range(32):
                oz, oy, ox, cim = row & 1, (row >> 1) & 1, (row >> 3) & 1, ((row >> 2) & 1) + 2 * ((row >> 4) & 1)
                if mt == 1:
                    cz, cy, cx = m % tcz, (m // tcz) % tcy, 4 * (m // (tcz * tcy)) + cim
                elif mt == 2:
                    cz, cy, cx = m, cim & 1, cim >> 1
                else:
                    cell = min(4 * m + cim, nc - 1)
                    cz, cy, cx = cell % tcz, (cell // tcz) % tcy, cell // (tcz * tcy)
                base.append((2 * cx + ox) * sx + (2 * cy + oy) * sy + 2 * cz + oz)
            for g in groups:
                per = {}
                for lane in g:
                    per.setdefault(base[lane] % 16, set()).add(base[lane])
                tot += max(len(v) for v in per.values())
        return tot / (2.0 * n_mt)

    for tc, n_mtiles, mask, want in (((4, 4, 2), 8, 1 | 2 | 4, 1.0), ((2, 2, 6), 6, 1 | 4, 1.0), ((3, 3, 3), 8, 1 | 2 | 4, 2.0),
                                     ((2, 2, 4), 4, 1 | 4, 1.0), ((2, 2, 3), 4, 1 | 4, 1.0)):
        tcv = np.array(tc, dtype=np.int32)
        out = np.zeros(3, dtype=np.int32)
        cyc = C.c_float()
        assert L.mi_debug_h2_layout(tcv.ctypes.data, n_mtiles, mask, out.ctypes.data, C.byref(cyc)) == 0
        mt, py, px = (int(v) for v in out)
        assert (mask >> mt) & 1
        assert abs(cycles(tc, mt, py, px) - cyc.value) < 1e-6, (tc, mt, py, px)
        assert abs(cyc.value - want) < 1e-6, (tc, mt, py, px, cyc.value)
