"""Where a keyframe insertion (Mapping.add_keyframe, reference Mapping.py:138-229) and the window rebuild that follows it spend
their launches and their time, on the pinned 640x480 sequence (scripts/ate_sequence.py SEQ640).

    python scripts/kf_insert_profile.py [--frames 100] [--out gpurun_out/kf_insert.txt]

Two runs of the loop: (1) synchronising timers around the sub-steps of add_keyframe and around the WindowBA construction + first
iteration; (2) a census: torch API calls (TorchFunctionMode, views / metadata excluded) and native calls (`_lib.check` names) per
como_amd source line, inside add_keyframe and inside the rebuild separately, per occurrence.  Run on the GPU box through gpurun.
"""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from como_amd import _lib, synth  # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule  # noqa: E402
from como_amd.odom.sequential import ComoSeq  # noqa: E402
import como_amd.odom.Mapping as map_mod  # noqa: E402
from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames  # noqa: E402

SKIP = {"__get__", "size", "dim", "shape", "is_contiguous", "data_ptr", "stride", "numel", "view", "reshape", "__getitem__",
        "expand", "unsqueeze", "squeeze", "permute", "transpose", "is_cuda", "device", "dtype", "element_size", "__len__",
        "storage_offset", "untyped_storage", "is_floating_point", "T", "view_as", "flatten", "movedim", "unbind", "__set__"}


def build(G, pix, dev, rgbs):
    K, T, _ = G["_rendered"]
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    return ComoSeq(loop_cfgs(G, pix, dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--pix", default="float")
    ap.add_argument("--out", default="gpurun_out/kf_insert.txt")
    args = ap.parse_args()
    dev = "cuda:0"
    G = dict(SEQ640, seed=1, nframes=args.frames)
    G["_rendered"] = render_frames(G)
    rgbs = [r.to(dev) for r in G["_rendered"][2]]
    lines = []

    # ---- (1) timers ------------------------------------------------------------------------------------------------------------
    odo = build(G, args.pix, dev, rgbs)
    parts = collections.OrderedDict()
    state = {"in_kf": False, "after_kf": False}

    def timed(obj, name, label, only_in_kf=True):
        fn = getattr(obj, name)

        def wrap(*a, **k):
            if only_in_kf and not state["in_kf"]:
                return fn(*a, **k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            parts.setdefault(label, []).append(time.perf_counter() - t0)
            return r
        setattr(obj, name, wrap)

    mp = odo.mapping
    for nm in ("get_img_and_grads", "run_model", "prep_predictor", "initialize_pose_vars", "initialize_kf_img_vars_vars",
               "initialize_sparse_pixel_vars", "initialize_sparse_landmark_vars", "store_vars", "prune_one_way", "depth_imgs_of"):
        timed(mp, nm, "add_keyframe." + nm)
    timed(map_mod, "track_and_init", "add_keyframe.track_and_init")
    import como_amd.odom.frontend.corr as corr_mod
    import como_amd.depth_cov.core.samplers as smp_mod
    for nm in ("reproject_and_filter", "distill_depth_from_scratch", "distill_conditional_depth_from_scratch", "sample_sparse_coords",
               "composeSE3", "backprojection", "_sample_at"):
        timed(corr_mod, nm, "  track_and_init." + nm)
    for nm in ("precalc_entropy_vars", "greedy_loop", "get_coords_domain", "get_cov_domain"):
        timed(smp_mod, nm, "    samplers." + nm)
    inner_add = mp.add_keyframe

    def add_kf(*a, **k):
        state["in_kf"] = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = inner_add(*a, **k)
        torch.cuda.synchronize()
        parts.setdefault("add_keyframe (total)", []).append(time.perf_counter() - t0)
        state["in_kf"] = False
        state["after_kf"] = True
        return r
    mp.add_keyframe = add_kf
    inner_it = mp.iterate

    def it(*a, **k):
        rebuild = mp._ba is None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = inner_it(*a, **k)
        torch.cuda.synchronize()
        lab = ("iterate: rebuild after a keyframe" if state["after_kf"] else "iterate: rebuild after a one-way frame") if rebuild else "iterate: same window"
        parts.setdefault(lab, []).append(time.perf_counter() - t0)
        state["after_kf"] = False
        return r
    mp.iterate = it
    for k in range(args.frames):
        odo.iter(1.0 + k, rgbs[k])
    torch.cuda.synchronize()
    lines.append("synchronising timers (ms): mean over the occurrences after the first two (warm-up) | n")
    for lab, v in parts.items():
        w = v[2:] if len(v) > 4 else v
        lines.append("  %-48s %8.3f  | %d" % (lab, 1e3 * sum(w) / len(w), len(v)))
    del odo

    # ---- (2) census ------------------------------------------------------------------------------------------------------------
    from torch.overrides import TorchFunctionMode
    odo = build(G, args.pix, dev, rgbs)
    mp = odo.mapping
    counts = {"add_keyframe": collections.Counter(), "rebuild after a keyframe": collections.Counter(),
              "rebuild after a one-way frame": collections.Counter(), "iterate, same window": collections.Counter()}
    occ = collections.Counter()
    cur = {"sect": None}

    def where():
        f = sys._getframe(2)
        while f is not None and ("como_amd" not in f.f_code.co_filename or f.f_code.co_filename.endswith("_lib.py")):
            f = f.f_back
        return (f.f_code.co_filename.split("como_amd/")[-1], f.f_lineno) if f is not None else ("?", 0)

    class Census(TorchFunctionMode):
        def __torch_function__(self, func, types, a=(), kw=None):
            if cur["sect"] is not None:
                name = getattr(func, "__name__", str(func))
                if name not in SKIP:
                    counts[cur["sect"]][where() + ("torch." + name,)] += 1
            return func(*a, **(kw or {}))

    inner_check = _lib.check

    def check(rc, name="", *a, **k):
        if cur["sect"] is not None:
            counts[cur["sect"]][where() + ("NATIVE " + str(name),)] += 1
        return inner_check(rc, name, *a, **k)
    _lib.check = check
    for mod in list(sys.modules.values()):                  # modules that did `from como_amd._lib import check`
        if mod is not None and getattr(mod, "__name__", "").startswith("como_amd") and getattr(mod, "check", None) is inner_check:
            mod.check = check
    inner_add2 = mp.add_keyframe
    st2 = {"after_kf": False}

    def add_kf2(*a, **k):
        prev, cur["sect"] = cur["sect"], "add_keyframe"
        occ["add_keyframe"] += 1
        r = inner_add2(*a, **k)
        cur["sect"] = prev
        st2["after_kf"] = True
        return r
    mp.add_keyframe = add_kf2
    inner_it2 = mp.iterate

    def it2(*a, **k):
        rebuild = mp._ba is None
        sect = ("rebuild after a keyframe" if st2["after_kf"] else "rebuild after a one-way frame") if rebuild else "iterate, same window"
        st2["after_kf"] = False
        prev, cur["sect"] = cur["sect"], sect
        occ[sect] += 1
        r = inner_it2(*a, **k)
        cur["sect"] = prev
        return r
    mp.iterate = it2
    with Census():
        for k in range(args.frames):
            if k == 12:                                       # (skip the initialisation and the first insertions' warm-up paths)
                for c in counts.values():
                    c.clear()
                occ.clear()
            odo.iter(1.0 + k, rgbs[k])
    torch.cuda.synchronize()
    for sect, c in counts.items():
        n = max(occ[sect], 1)
        tot_t = sum(v for (f, l, nm), v in c.items() if nm.startswith("torch."))
        tot_n = sum(v for (f, l, nm), v in c.items() if nm.startswith("NATIVE"))
        lines.append("")
        lines.append("== %s: %d occurrences; per occurrence %.1f torch API calls + %.1f native calls" % (sect, occ[sect], tot_t / n, tot_n / n))
        byfile = collections.Counter()
        for (f, l, nm), v in c.items():
            byfile[f] += v
        lines.append("   by file: " + ", ".join("%s %.1f" % (f, v / n) for f, v in byfile.most_common(14)))
        byline = collections.defaultdict(list)
        for (f, l, nm), v in c.items():
            byline[(f, l)].append((nm, v))
        for (f, l), v in sorted(byline.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:70]:
            lines.append("  %6.2f  %s:%d  %s" % (sum(x[1] for x in v) / n, f, l, ", ".join("%s x%.2g" % (nm.replace("torch.", ""), cnt / n) for nm, cnt in sorted(v, key=lambda t: -t[1]))))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
