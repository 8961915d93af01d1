"""Build oracle/_ref/como_backends_ref*.so from the reference's own CPU source.

Test infrastructure only.  Runs only where /root/reference exists (the build
container); the GPU box uses the prebuilt .so that travels with the snapshot.
Outputs go to oracle/_ref/ only (git-ignored).  No reference source is copied.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/como/backend/src/cov_cpu.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
MOD = "como_backends_ref"


def out_path():
    return os.path.join(OUT_DIR, MOD + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    if not os.path.exists(REF_SRC):
        return None
    out = out_path()
    bind = os.path.join(HERE, "ref_bind.cpp")
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) > max(os.path.getmtime(REF_SRC), os.path.getmtime(bind))):
        return out
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DTORCH_EXTENSION_NAME={MOD}",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *inc, REF_SRC, bind, "-o", out, f"-L{tl}", f"-Wl,-rpath,{tl}",
           "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    subprocess.check_call(cmd)
    return out


def load():
    """Import the prebuilt module (torch must be imported first)."""
    import importlib.util
    import torch  # noqa: F401
    p = out_path()
    if not os.path.exists(p):
        return None
    spec = importlib.util.spec_from_file_location(MOD, p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
