"""Host timeline of the sequential loop's frames by request kind (None / one-way / keyframe): when, relative to the start of the
frame, the Python side enters and leaves each stage -- the tracker's read-back wait, the hand-over of the frame, the window
re-target, the enqueue of the iteration, the tracker refresh.  Second pass over the pinned 640x480 sequence (the first pass pays
the one-time captures).  No synchronisation is added; the stamps are host times.   python scripts/frame_host_timeline.py [out.txt]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from como_amd import synth                                                     # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule              # noqa: E402
from como_amd.odom import Mapping as mapping_mod                               # noqa: E402
from como_amd.odom import window_ba as wba                                     # noqa: E402
from como_amd.odom.backend import dense_ref, photo                             # noqa: E402
from como_amd.odom.sequential import ComoSeq                                   # noqa: E402
from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames              # noqa: E402

dev = "cuda:0"
G = dict(SEQ640, seed=1, nframes=100)
K, T, rgbs_cpu = render_frames(G)
rgbs = [r.to(dev) for r in rgbs_cpu]
model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
stamps = []            # (label, t_enter, t_exit) of the current frame
frame_t0 = [0.0]


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        stamps.append((label, t0 - frame_t0[0], time.perf_counter() - frame_t0[0]))
        return r
    if hasattr(fn, "__dict__") and not isinstance(obj, type) and not isinstance(fn, type):
        w.__dict__ = fn.__dict__          # (function attributes -- track_and_init.corr_host -- are read through the wrapper too)
    setattr(obj, name, w)


for cls, names in ((wba.WindowBA, ("retarget", "_load_images", "_load_frames", "_finish_topology", "_build_pair_table", "_patch_args", "step", "speculate", "__init__",
                                   "_prepare_topology", "_host_tables", "_prepare_fused",
                                   "linearize_fused", "_finalize", "snapshot_state")),
                   (mapping_mod.Mapping, ("handle_tracking_data", "add_one_way_frame", "add_keyframe", "get_curr_world_state", "_window_state",
                                          "iterate", "get_kf_ref_data", "_check_solver", "start_model", "take_model"))):
    for n in names:
        if hasattr(cls, n):
            wrap(cls, n, cls.__name__ + "." + n)
wrap(photo, "photo_system_factored", "photo.photo_system_factored")

for n in ("dense_reference_factored", "full_image_median"):
    wrap(dense_ref, n, "dense_ref." + n)
    if hasattr(wba, n):
        wrap(wba, n, "dense_ref." + n)
from como_amd.odom.frontend import corr as corr_mod                            # noqa: E402
from como_amd.odom import Tracking as tracking_mod                             # noqa: E402
for n in ("handle_frame", "update_kf_reference", "decide_frame", "_track_frame_graph"):
    wrap(tracking_mod.Tracking, n, "Tracking." + n)
if hasattr(mapping_mod, "track_and_init"):
    wrap(mapping_mod, "track_and_init", "corr.track_and_init")
    wrap(mapping_mod, "prepare_track_and_init", "corr.prepare_track_and_init")

for n in ("distill_depth_from_scratch", "distill_conditional_depth_from_scratch", "sample_sparse_coords", "reproject_and_filter",
          "reproject_points", "_sample_at", "get_correspondence_errors"):
    if hasattr(corr_mod, n):
        wrap(corr_mod, n, "corr:" + n)
for n in ("prep_predictor", "initialize_pose_vars", "initialize_kf_img_vars_vars", "initialize_sparse_pixel_vars", "initialize_sparse_landmark_vars",
          "store_vars", "get_img_and_grads", "depth_imgs_of", "get_kf_viz_data", "prune_one_way", "run_model"):
    wrap(mapping_mod.Mapping, n, "Mapping." + n)
from como_amd.depth_cov.core import samplers as samplers_mod                  # noqa: E402
for n in ("precalc_entropy_vars", "greedy_loop", "_thin", "greedy_conditional_entropy"):
    if hasattr(samplers_mod, n):
        wrap(samplers_mod, n, "samplers:" + n)
from como_amd.depth_cov.core import distill_depth as dd_mod                    # noqa: E402
for n in ("get_predictor",):
    if hasattr(dd_mod, n):
        wrap(dd_mod, n, "distill:" + n)

wrap(wba.smap, "subselect_pixels", "smap.subselect_pixels")
for n in ("prep_tracking_img", "gradient_module", "depth_pyr_module"):
    if hasattr(tracking_mod.Tracking, n):
        wrap(tracking_mod.Tracking, n, "Tracking." + n)
wrap(wba.photo, "PairTable", "photo.PairTable")

# device-side clock of a few host markers: an event where the tracker's result has just been read back (the device is idle there:
# device time = host time) and events at later host points; elapsed_time between them = when the DEVICE got there
gpu_marks, gpu_prev = [], []


def mark(obj, name, label, at_exit=False):
    fn = getattr(obj, name)

    def w(*a, **k):
        if not at_exit:
            e = torch.cuda.Event(enable_timing=True); e.record(); gpu_marks.append((label, time.perf_counter() - frame_t0[0], e))
        r = fn(*a, **k)
        if at_exit:
            e = torch.cuda.Event(enable_timing=True); e.record(); gpu_marks.append((label, time.perf_counter() - frame_t0[0], e))
        return r
    setattr(obj, name, w)


mark(tracking_mod.Tracking, "decide_frame", "A decision read back")
mark(mapping_mod.Mapping, "add_keyframe", "B add_keyframe done", at_exit=True)
mark(mapping_mod.Mapping, "add_one_way_frame", "B add_one_way_frame done", at_exit=True)
mark(wba.WindowBA, "step", "C iteration enqueue begins")
mark(wba.WindowBA, "step", "D iteration enqueued", at_exit=True)
mark(tracking_mod.Tracking, "update_kf_reference", "E tracker refresh enqueued", at_exit=True)
gagg = {}

agg = {}
for p in range(2):
    odo = ComoSeq(loop_cfgs(G, "float", dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)
    for k in range(100):
        del stamps[:]
        prev_marks, prev_kind = list(gpu_marks), (kind if k else None)
        del gpu_marks[:]
        frame_t0[0] = time.perf_counter()
        kind = odo.iter(1.0 + k, rgbs[k])
        if p == 1 and k > 6 and prev_marks and prev_marks[0][0].startswith("A") and gpu_marks and gpu_marks[0][0].startswith("A"):
            # (this frame's read-back has synchronised: the previous frame's events are complete)
            a0 = prev_marks[0]
            g = gagg.setdefault(str(prev_kind), {})
            for lab, th, e in prev_marks[1:]:
                v = g.setdefault(lab, [0, 0.0, 0.0])
                v[0] += 1
                v[1] += th - a0[1]
                v[2] += 1e-3 * a0[2].elapsed_time(e)
        t_end = time.perf_counter() - frame_t0[0]
        if p == 1 and odo.mapping.is_init and k > 5:
            a = agg.setdefault(str(kind), {"n": 0, "end": 0.0, "lab": {}})
            a["n"] += 1
            a["end"] += t_end
            seen = {}
            for lab, t0, t1 in stamps:
                seen[lab] = seen.get(lab, 0) + 1
                key = lab if seen[lab] == 1 else f"{lab} #{seen[lab]}"
                e = a["lab"].setdefault(key, [0, 0.0, 0.0])
                e[0] += 1
                e[1] += t0
                e[2] += t1
    torch.cuda.synchronize()
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
print("pair tables built ahead / adopted (both passes):", wba.SPEC_STATS, file=out)
for kind, a in agg.items():
    print(f"== {kind}: {a['n']} frames, host time of odo.iter {1e3 * a['end'] / a['n']:.3f} ms (enter .. leave, ms from the start of the frame; calls)",
          file=out)
    for lab, (c, t0, t1) in sorted(a["lab"].items(), key=lambda kv: kv[1][1] / kv[1][0]):
        print(f"   {1e3 * t0 / c:7.3f} .. {1e3 * t1 / c:7.3f}  ({1e3 * (t1 - t0) / c:6.3f})  x{c / a['n']:.2f}  {lab}", file=out)
for kind, g in gagg.items():
    print(f"== {kind}: host / device time after the tracker's result was read back, ms (device later than host = the device is behind: GPU-bound there)", file=out)
    for lab, (c, th, tg) in sorted(g.items()):
        print(f"   host {1e3 * th / c:7.3f}   device {1e3 * tg / c:7.3f}   {lab}", file=out)
