"""Build libmi_gnina.so (HIP, gfx950) in-tree with hipcc.  No torch, no cmake: plain C ABI library.

    python -m gnina_amd.build            # incremental
    python -m gnina_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmi_gnina.so")
SOURCES = ["engine.cpp", "options.cpp", "model.cpp", "typer.cpp", "voxelize.hip", "conv3d.hip", "conv3d_bf16.hip", "conv3d_h2.hip", "conv3d_h2_dense.hip", "conv3d_h2_ws.hip", "vina.hip", "vina_host.cpp", "pool.cpp",
           "../host/typed_atoms.cpp", "../host/pdbqt.cpp"]
# -ffp-contract=off: fp32 ops round exactly as written (the voxelizer's in/out decisions must be
# bit-identical to the reference arithmetic); fused ops are spelled out (fmaf / MFMA builtins).
# -target-feature -packed-fp32-ops: NO packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in any
# kernel of the library.  Round 6: a chain of dependent packed-fp32 instructions gives wrong
# results in the upper half of a wavefront (lanes 32-63) in a few launches per thousand while another queue's Dense conv
# kernels share the SIMD -- what made voxelize_tiles deviate next to a second scorer; reproduced with a synthetic victim
# (tools/microbench/pk_f32_next_to_mfma.hip) next to the real aggressor.  The same arithmetic in scalar fp32 instructions is
# clean (0 of 50,000 launches) and costs nothing measurable (voxelizer 0.872 -> 0.871 ms, headline / Dense / fp32-MFMA
# unchanged: tools/experiments/r6_conc5.sh), so the instruction class is off for every translation unit: the conv epilogues
# and the Vina kernels used it too and may run next to another scorer's kernels just the same.  (The flag reaches the host
# compilation as well, which reports it as unknown: filtered below.)  tests/test_cabi_cpu.py disassembles the library.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-x", "hip"]
_HOST_NOISE = "is not a recognized feature for this target (ignoring feature)"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.exists(c) or c == "hipcc"):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mi_gnina.h"))
    headers.append(os.path.join(HERE, "host", "typed_atoms.h"))
    headers.append(os.path.join(HERE, "host", "pdbqt.h"))
    hdr_mtime = max(os.path.getmtime(h) for h in headers)
    # a library newer than every source and header is current even when the object files are gone (the GPU box receives the
    # built .so without lib/obj/: no reason to spend its minutes on recompiling everything)
    src_mtime = max([hdr_mtime] + [os.path.getmtime(os.path.join(CSRC, s_)) for s_ in SOURCES])
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= src_mtime:
        return LIB
    objs, todo = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_mtime):
            todo.append([hipcc()] + FLAGS + ["-c", sp, "-o", op])
    rebuilt = bool(todo)
    if todo:  # the translation units are independent: compile them side by side (conv3d_h2.hip alone takes ~1.5 min)
        from concurrent.futures import ThreadPoolExecutor

        def one(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            err = "\n".join(l for l in r.stderr.splitlines() if _HOST_NOISE not in l)
            if err.strip():
                print(err, file=sys.stderr, flush=True)
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)

        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(one, todo))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_host(force=False, verbose=False):
    """C++ host adapters (HipCNNScorer : DLScorer) + their test driver, plain g++ over the C ABI."""
    build(force=force, verbose=verbose)
    root = os.path.dirname(HERE)
    exe = os.path.join(LIBDIR, "test_host_scorer")
    srcs = [os.path.join(root, "tests", "cpp", "test_host_scorer.cpp"), os.path.join(HERE, "host", "hip_cnn_scorer.cpp"),
            os.path.join(HERE, "host", "hip_cache.cpp")]
    deps = srcs + [os.path.join(HERE, "host", "hip_cnn_scorer.h"), os.path.join(HERE, "host", "hip_cache.h"),
                   os.path.join(HERE, "host", "gnina_types.h"), LIB]
    if force or not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", exe] + srcs + ["-L" + LIBDIR, "-lmi_gnina",
                                                                      "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    # the multi-device pool's C++ driver (tests/cpp/test_pool.cpp): C ABI + the HIP runtime API for its device buffers
    pexe = os.path.join(LIBDIR, "test_pool")
    psrc = os.path.join(root, "tests", "cpp", "test_pool.cpp")
    if force or not os.path.exists(pexe) or os.path.getmtime(pexe) < max(os.path.getmtime(psrc), os.path.getmtime(LIB)):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", pexe, psrc,
               "-L" + LIBDIR, "-lmi_gnina", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    # the device-path protocol on a mock transport (no GPU, no RCCL): tests/test_pool_protocol_cpu.py runs it
    qexe = os.path.join(LIBDIR, "test_pool_protocol")
    qsrc = os.path.join(root, "tests", "cpp", "test_pool_protocol.cpp")
    qhdr = os.path.join(CSRC, "pool_protocol.h")
    if force or not os.path.exists(qexe) or os.path.getmtime(qexe) < max(os.path.getmtime(qsrc), os.path.getmtime(qhdr)):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-o", qexe, qsrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return exe


def extract_weights(models_dir="/root/reference/gninasrc/lib/models", verbose=False):
    """Weight blobs of ALL the reference's built-in models (cnn_torch_scorer.cpp:24-64: default2017, the default2018 /
    dense families, *_ensemble members ...) from its TorchScript files, into gnina_amd/weights/.  Seven blobs are
    committed; the other ~57 (116 MB) are derived data produced here whenever the reference is present (this
    container) and travel to the GPU box with the working tree -- they are git-ignored, not gpurun-ignored."""
    wdir = os.path.join(HERE, "weights")
    if not os.path.isdir(models_dir):
        return 0
    todo = [f for f in sorted(os.listdir(models_dir))
            if f.endswith(".pt") and not os.path.exists(os.path.join(wdir, f[:-3].replace(".", "_") + ".mgw"))]
    if not todo:
        return 0
    from gnina_amd.tools import extract_weights as ew
    n = 0
    for f in todo:
        data, name = ew.convert(os.path.join(models_dir, f))
        with open(os.path.join(wdir, name + ".mgw"), "wb") as out:
            out.write(data)
        n += 1
        if verbose:
            print("extracted", name, len(data))
    return n


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    print("weight blobs extracted:", extract_weights(verbose=True))
