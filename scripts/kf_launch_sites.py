"""Which Python source lines launch the SMALL torch kernels of a keyframe frame (the pinned 640x480 sequence): torch's profiler
with stacks around the keyframe frames of the steady state; per (como_amd source line, op) the number of GPU kernels launched per
keyframe frame.  Native como:: launches have no torch op and are not listed.
    python scripts/kf_launch_sites.py [--frames 100] [--out gpurun_out/kf_launch_sites.txt]"""
import argparse
import collections
import os
import sys

import torch
import torch.utils._python_dispatch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from como_amd import synth  # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule  # noqa: E402
from como_amd.odom.sequential import ComoSeq  # noqa: E402
from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames  # noqa: E402


NO_KERNEL = {"aten::empty.memory_format", "aten::empty_strided", "aten::view", "aten::_unsafe_view", "aten::reshape", "aten::as_strided",
             "aten::select.int", "aten::slice.Tensor", "aten::unsqueeze", "aten::squeeze.dim", "aten::squeeze", "aten::expand", "aten::t",
             "aten::transpose.int", "aten::permute", "aten::detach", "aten::alias", "aten::_local_scalar_dense", "aten::is_nonzero",
             "aten::empty_like", "aten::unbind.int", "aten::split.Tensor", "aten::view_as_real", "aten::lift_fresh", "aten::narrow",
             "aten::resize_", "aten::set_.source_Storage_storage_offset", "aten::_reshape_alias", "aten::unfold", "aten::diagonal",
             "aten::movedim.int", "aten::flatten.using_ints", "aten::record_stream", "aten::is_pinned", "aten::_pin_memory"}


class _Census(torch.utils._python_dispatch.TorchDispatchMode):
    """aten ops that reach the dispatcher while a keyframe frame runs, by the innermost como_amd source line on the Python stack
    (ops on CPU tensors and pure views are skipped)."""

    def __init__(self, sites, root):
        super().__init__()
        self.sites, self.root = sites, root

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name + ("." + func._overloadname if func._overloadname and func._overloadname != "default" else "")
        if name not in NO_KERNEL:
            ts = [a for a in list(args) + ([out] if torch.is_tensor(out) else []) if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts):
                import traceback
                line = "?"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if "/como_amd/" in fr.filename:
                        line = fr.filename.split("/como_amd/")[-1] + ":" + str(fr.lineno)
                        break
                self.sites[(line, name)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--out", default="gpurun_out/kf_launch_sites.txt")
    args = ap.parse_args()
    dev = "cuda:0"
    G = dict(SEQ640, seed=1, nframes=args.frames)
    G["_rendered"] = render_frames(G)
    K, T, rgbs = G["_rendered"]
    rgbs = [r.to(dev) for r in rgbs]
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    odo = ComoSeq(loop_cfgs(G, "float", dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)
    mp = odo.mapping
    state = {"kf": False}
    inner = mp.add_keyframe

    def add_kf(*a, **k):
        state["kf"] = True
        return inner(*a, **k)
    mp.add_keyframe = add_kf
    # first pass: which frames insert a keyframe (the loop is deterministic)
    kinds = []
    for k in range(args.frames):
        state["kf"] = False
        odo.iter(1.0 + k, rgbs[k])
        kinds.append(state["kf"])
    torch.cuda.synchronize()
    del odo
    odo = ComoSeq(loop_cfgs(G, "float", dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)
    sites = collections.Counter()
    kernels = collections.Counter()
    nkf = 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in range(args.frames):
        if not (kinds[k] and k > 30):
            odo.iter(1.0 + k, rgbs[k])
            continue
        torch.cuda.synchronize()
        with _Census(sites, root):
            odo.iter(1.0 + k, rgbs[k])
        torch.cuda.synchronize()
        nkf += 1
    with open(args.out, "w") as f:
        f.write(f"{nkf} keyframe frames profiled; aten ops on GPU tensors per keyframe frame (views / allocations excluded: ~ one launch each), by como_amd source line\n")
        tot = sum(sites.values()) / max(nkf, 1)
        f.write(f"total {tot:.1f} per keyframe frame\n")
        for (line, name), v in sorted(sites.items(), key=lambda kv: -kv[1]):
            f.write(f"{v / nkf:7.2f}  {line:60s} {name}\n")
    print(open(args.out).read()[:6000])


if __name__ == "__main__":
    main()
