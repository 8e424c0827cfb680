"""CPU-side checks of the C ABI (no GPU needed): the library builds, loads, exports every symbol
include/mi_gnina.h declares, and fails LOUDLY (no CPU fallback) when there is no device."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import build, capi as c
    build.build()
    return c


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "mi_gnina.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree(capi):
    decl = declared_functions()
    assert decl, "no declarations parsed"
    assert sorted(capi.SYMBOLS) == decl


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    for sym in declared_functions():
        assert hasattr(L, sym), f"libmi_gnina.so does not export {sym}"
    assert L.mi_gnina_abi_version() == capi.ABI_VERSION == 2


def test_no_torch_or_oracle_dependency(capi):
    """The product library must not link the oracle or torch (plain C ABI over HIP)."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out and "libtorch" not in out and "libc10" not in out
    assert "libamdhip64" in out


def test_product_sources_never_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/."""
    pkg = os.path.join(ROOT, "gnina_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
                assert "voxel_ref" not in txt, f


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: the failure path is for GPU-less hosts")
def test_device_code_has_no_packed_fp32_instructions(capi, tmp_path):
    """Round 6: chains of packed-fp32 VALU instructions give wrong results in lanes 32-63 while
    another queue's Dense conv kernels share the SIMD -- voxelize_tiles deviated next to a second scorer because of them.  The
    library is compiled with the instruction class switched off (gnina_amd/build.py FLAGS): no kernel may contain one."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not present")
    from gnina_amd import build as b
    so = str(tmp_path / "lib.so")
    shutil.copy(b.LIB, so)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=str(tmp_path))
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(objs) >= 6, objs  # one code object per .hip translation unit
    kernels, packed = 0, []
    for f in objs:
        out = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        kernels += out.count("s_endpgm")
        packed += [ln.strip() for ln in out.splitlines() if "v_pk_" in ln and "_f32" in ln]
    assert kernels > 50 and not packed, packed[:5]


def test_fails_loudly_without_gpu(capi):
    assert capi.lib().mi_gnina_device_count() == 0
    with pytest.raises(capi.MiGninaError):
        capi.init(0)
    with pytest.raises(capi.MiGninaError) as ei:
        capi.Model("default2017")
    assert "device" in str(ei.value).lower() or "hip" in str(ei.value).lower()


def test_null_and_error_paths_do_not_crash(capi):
    L = capi.lib()
    assert L.mi_scorer_create(None, 0) is None
    assert L.mi_last_error().decode() != ""
    assert L.mi_model_load(b"garbage", 7, b"x") is None
    assert "MIGNINA1" in L.mi_last_error().decode()
    assert L.mi_scorer_num_models(None) == 0
    L.mi_model_release(None)
    L.mi_scorer_destroy(None)
    assert L.mi_model_type_channel(None, 0, 2, None) == -1
    # every handle-taking entry point added later refuses NULL handles with a status, not a crash
    import ctypes as C
    one = (C.c_float * 8)()
    assert L.mi_scorer_set_flex(None, None, 0) != capi.MI_OK
    assert L.mi_scorer_set_precision(None, 1) != capi.MI_OK
    assert L.mi_scorer_score_ragged(None, one, None, 1, 1, None, one, one, one, None) != capi.MI_OK
    assert L.mi_scorer_score_flex(None, one, None, 1, 1, None, None, one, one, one, None, None, None) != capi.MI_OK
    assert L.mi_vina_coords_batch(None, one, 1, one) != capi.MI_OK
    assert L.mi_cnn_eval_batch(None, None, one, 1, None, None, 0, one, None) != capi.MI_OK
    assert L.mi_cnn_refine_batch(None, None, one, 1, None, 3, one, None, None) != capi.MI_OK
    assert L.mi_vina_set_screen(None, 0, None) != capi.MI_OK and L.mi_vina_screen_size(None) == 0
    assert L.mi_vina_screen_dims(None, None, None, None) != capi.MI_OK
    assert L.mi_vina_mc_screen(None, 1, one, one, one, one, None, one, one, None, None, None) != capi.MI_OK
    assert L.mi_vina_eval_screen(None, one, one, 1, one, 1, one, None, None) != capi.MI_OK
    assert L.mi_vina_refine_screen(None, one, one, 1, one, one, one, None) != capi.MI_OK
    assert L.mi_vina_final_energies_screen(None, one, one, 1, one, one, one, None) != capi.MI_OK
    assert L.mi_model_load_file_ex(b"/nonexistent.mgw", C.c_float(0.25), C.c_float(23.75)) is None
    assert L.mi_read_gninatypes(None, None, None, 0, None) != capi.MI_OK


def test_synth_generator_is_deterministic_and_in_spec():
    from gnina_amd import synth
    types = np.arange(2, 20, dtype=np.int32)
    a = synth.make_complex(3, types, types, 500, 16, 5)
    b = synth.make_complex(3, types, types, 500, 16, 5)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    rec_xyz, rec_smt, lig_smt, poses = a
    assert rec_xyz.shape == (500, 3) and poses.shape == (5, 16, 3)
    assert np.linalg.norm(rec_xyz, axis=1).min() >= 4.0 and np.abs(rec_xyz).max() <= 20.0
    # poses are rigid: pairwise distances preserved
    d0 = np.linalg.norm(poses[0][:, None] - poses[0][None], axis=-1)
    d3 = np.linalg.norm(poses[3][:, None] - poses[3][None], axis=-1)
    assert np.abs(d0 - d3).max() < 1e-4


def test_rank_poses_sort_and_remove_redundant(capi):
    """do_search's tail (main.cpp:182-192,348-361) is host logic: runs without a GPU."""
    rng = np.random.RandomState(0)
    base = rng.normal(size=(6, 10, 3)).astype(np.float32) * 5
    coords = np.concatenate([base, base[:2] + 0.1], axis=0)             # poses 6,7 are near-duplicates of 0,1
    score = np.array([0.9, 0.2, 0.5, 0.7, 0.1, 0.3, 0.95, 0.15], dtype=np.float32)
    aff = np.arange(8, dtype=np.float32)
    energy = -score
    keep = capi.rank_poses(score, aff, energy, coords, 0, 1.0)
    assert list(keep) == [6, 3, 2, 5, 1, 4]          # 0 dropped (dup of 6, lower score); 7 dropped (dup of 1)
    assert list(capi.rank_poses(score, aff, energy, coords, 2, 1.0)) == [6, 3, 2, 5, 1, 4]   # by energy ascending
    assert list(capi.rank_poses(score, aff, energy, coords, 1, 1.0)) == [7, 6, 5, 4, 3, 2]   # by affinity descending
    assert len(capi.rank_poses(score, aff, energy, coords, 0, 0.0)) == 8                      # nothing is redundant
    assert len(capi.rank_poses(score[:0], aff[:0], energy[:0], coords[:0], 0, 1.0)) == 0


def test_gninatypes_round_trip_and_errors(capi, tmp_path):
    """`.gninatypes` reader/writer (gninatyper.cpp:30-36: struct {float x, y, z; int type;} records)."""
    import struct
    rng = np.random.RandomState(0)
    xyz = rng.normal(0, 10, (57, 3)).astype(np.float32)
    smt = rng.randint(0, 28, 57).astype(np.int32)
    # a file written the way gninatyper writes it
    p = tmp_path / "lig.gninatypes"
    with open(p, "wb") as f:
        for i in range(57):
            f.write(struct.pack("<fffi", *xyz[i], smt[i]))
    x2, s2 = capi.read_gninatypes(str(p))
    assert np.array_equal(x2, xyz) and np.array_equal(s2, smt)
    # our writer produces the same bytes
    q = tmp_path / "out.gninatypes"
    capi.write_gninatypes(str(q), xyz, smt)
    assert open(p, "rb").read() == open(q, "rb").read()
    # empty file = zero atoms
    e = tmp_path / "empty.gninatypes"
    e.write_bytes(b"")
    x0, s0 = capi.read_gninatypes(str(e))
    assert x0.shape == (0, 3) and s0.shape == (0,)
    # truncated record, bad type index, missing file
    t = tmp_path / "trunc.gninatypes"
    t.write_bytes(open(p, "rb").read()[:-5])
    with pytest.raises(capi.MiGninaError, match="multiple of the 16-byte"):
        capi.read_gninatypes(str(t))
    b = tmp_path / "bad.gninatypes"
    b.write_bytes(struct.pack("<fffi", 0, 0, 0, 28))
    with pytest.raises(capi.MiGninaError, match="outside 0..27"):
        capi.read_gninatypes(str(b))
    with pytest.raises(capi.MiGninaError, match="could not open"):
        capi.read_gninatypes(str(tmp_path / "nope.gninatypes"))


def test_fp16_operand_split_of_the_conv_weights(capi):
    """The split-fp16 kernels (conv3d_h2.hip) write every fp32 operand as hi + lo, two fp16 numbers.  The loader's host
    code does it for the weights: round to nearest even like numpy's float16 (subnormals, ties, overflow included), a
    per-layer power-of-two scale, and hi + lo within 2^-21 of the scaled weight (22-23 significant bits)."""
    rng = np.random.RandomState(3)
    w = np.concatenate([rng.normal(0, 0.05, 5000), rng.uniform(-1, 1, 2000) * 10.0 ** rng.uniform(-9, 0, 2000),
                        [0.0, -0.0, 1.0, -1.0, 0.1, 6.1e-5, 5.96e-8, 2.98e-8, 3.0e-8, 65504.0, 65519.9, 1e-30]]).astype(np.float32)
    # explicit scale 1: plain conversions, compared bit for bit with numpy (values beyond fp16's range saturate to inf there too)
    hi, lo, used = capi.split_f16(w, scale=1.0)
    assert used == 1.0
    with np.errstate(over="ignore"):
        ref_hi = w.astype(np.float16)
        ref_lo = (w - ref_hi.astype(np.float32)).astype(np.float16)
    fin = np.isfinite(ref_hi)
    assert np.array_equal(hi.view(np.uint16), ref_hi.view(np.uint16))
    assert np.array_equal(lo.view(np.uint16)[fin], ref_lo.view(np.uint16)[fin])
    # the loader's own scale: a power of two, max |w| * scale in [2^13, 2^14), reconstruction to 2^-21 relative
    layer = rng.normal(0, 0.03, 4096).astype(np.float32)
    hi, lo, sw = capi.split_f16(layer)
    assert sw == 2.0 ** round(np.log2(sw)) and 2.0 ** 13 <= np.abs(layer).max() * sw < 2.0 ** 14
    back = (hi.astype(np.float64) + lo.astype(np.float64)) / sw
    big = np.abs(layer) > 1e-4
    assert (np.abs(back - layer)[big] <= 2.0 ** -21 * np.abs(layer)[big]).all()
    assert np.abs(back - layer).max() <= 2.0 ** -21 * np.abs(layer).max()
    assert capi.lib().mi_debug_split_f16(None, 4, 1.0, None, None, None) != capi.MI_OK
