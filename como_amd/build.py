"""Build libcomo_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m como_amd.build            # incremental
    python -m como_amd.build --force
    python -m como_amd.build --ab       # libcomo_hip_ab.so: the same sources with -DCOMO_AB_VARIANTS (the losing / ablation kernel
                                        # variants of csrc/ba.hip and csrc/chol.hip, for scripts/ab/ and scripts/micro/ only;
                                        # COMO_HIP_LIB=<path> makes como_amd._lib load it)

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcomo_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
FLAGS += os.environ.get("COMO_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "como_hip.h"))
    return hdr


def build(force=False, verbose=False, ab=False):
    libdir = os.path.join(HERE, "lib_ab") if ab else LIBDIR
    lib = os.path.join(libdir, "libcomo_hip_ab.so") if ab else LIB
    flags = FLAGS + (["-DCOMO_AB_VARIANTS"] if ab else [])
    os.makedirs(libdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    newest_hdr = max(os.path.getmtime(h) for h in _deps())
    procs = []
    for src in sources():
        obj = os.path.join(libdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_hdr)):
            continue
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out.decode()}")
    if procs or not os.path.exists(lib) or force:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ab="--ab" in sys.argv))
