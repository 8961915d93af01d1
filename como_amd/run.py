"""Headless runner: the reference's entry script without the GUI.

    python -m como_amd.run --dataset_type tum|replica|scannet --dataset_dir D/ [--config config/como.yml] [--out results/x.txt]
                           [--device cuda:N] [--img_size 192 256] [--model_path models/scannet.ckpt | --random_weights SEED]

What it replaces, line by line:
  como/como_dataset.py:11-37        argparse (`--dataset_type`, `--dataset_dir`), `img_size = [192, 256]`, `get_dataset`,
                                    `yaml.safe_load("./config/como.yml")`, `torch.manual_seed(0)`
  como/gui/GuiWindow.py:528-599     `update_main`: the first frame of the loader only initialises the renderer (it never reaches
                                    `iter`); every later frame -> `ComoSeq.iter(timestamp, rgb)`
  como/gui/GuiWindow.py:329-357     `update_kf_vars`: the keyframe HISTORY (timestamps, poses of every keyframe that was ever in the
                                    window, overwritten in place while it is)
  como/gui/GuiWindow.py:359-367     `save_traj`: `./results/<dataset.save_traj_name>.txt`, the keyframe history in TUM format
                                    (`como/utils/io.py:4-23`)

The YAML schema is the reference's `config/como.yml` (sections `tracking` / `mapping`, constants of SURVEY.md section 5); keys this
build adds are optional: `mapping.pix_dtype` (element type of the per-pixel kernels: `float` | `double`, default `double` = the
mapping dtype), `mapping.network_size` ([192, 256], Mapping.py:399), `mapping.viewer_snapshots` (the runner sets it to false: no
viewer).  Both sections run on ONE device (`--device`; the reference's two-GPU tracking / mapping split is its multiprocessing
mode, out of scope); `device: cpu` is refused -- the HIP path has no CPU fallback.

Differences from the reference, on purpose: (i) the reference refreshes the keyframe history whenever a viewer snapshot is taken --
on every keyframe insertion AND whenever one second of wall-clock time has passed (MappingSeq.py:40-41), which makes the saved poses
of the last window depend on the machine's speed; here the history is refreshed on every keyframe insertion and once more after the
last frame (deterministic; equal to the reference's file when its timer fires on the last frame); (ii) the per-frame tracked poses
(`ComoSeq.est_poses`, which the reference records but never writes) go to `<out stem>_frames.txt`."""
import argparse
import os
import sys
import time

import torch

from como_amd.data.odom_datasets import get_dataset
from como_amd.utils.io import save_traj

DEFAULT_IMG_SIZE = [192, 256]                # como/como_dataset.py:34


def load_slam_cfg(path, device=None, pix_dtype=None):
    """`config/como.yml` -> {"tracking": ..., "mapping": ...} for `ComoSeq` (one device for both sections)."""
    import yaml
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    for sec in ("tracking", "mapping"):
        if sec not in cfg:
            raise ValueError(f"{path}: section `{sec}` is missing (schema of the reference's config/como.yml)")
    dev = device or cfg["mapping"].get("device", "cuda:0")
    if str(dev).startswith("cpu"):
        raise RuntimeError("como_amd.run: device cpu -- the HIP path has no CPU fallback (the reference's CPU mode is the oracle's job)")
    t, m = dict(cfg["tracking"]), dict(cfg["mapping"])
    t["device"] = m["device"] = dev
    m.setdefault("network_size", list(DEFAULT_IMG_SIZE))
    m["pix_dtype"] = pix_dtype or m.get("pix_dtype", m["dtype"])
    m.setdefault("viewer_snapshots", False)
    return {"tracking": t, "mapping": m}


def load_model(mcfg, device, random_weights=None):
    """The DepthCov weights: `mapping.model_path` (a Lightning checkpoint = a torch-saved dict with `state_dict`, Mapping.py:402-404)
    or, with --random_weights SEED, the seeded random network the parity fixtures use (there is no checkpoint in this tree)."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    if random_weights is not None:
        from como_amd import synth
        sd = synth.depthcov_state_dict(int(random_weights))
    else:
        path = mcfg.get("model_path")
        if not path or not os.path.exists(path):
            raise FileNotFoundError(f"como_amd.run: DepthCov checkpoint `{path}` not found (mapping.model_path); "
                                    "--random_weights SEED runs a seeded random network instead")
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck.get("state_dict", ck)
    return DepthCovModule({k: v.to(device) for k, v in sd.items() if torch.is_tensor(v)})


class KeyframeHistory:
    """GuiWindow.update_kf_vars (:329-357) for timestamps and poses: a growing list in which the current window occupies
    [start, start + B); the window start advances by one whenever the oldest keyframe's timestamp changes."""

    def __init__(self):
        self.timestamps, self.poses, self.start = [], [], 0

    def update(self, kf_timestamps, kf_poses):
        ts = [float(t) for t in kf_timestamps]
        if not ts:
            return
        if self.timestamps and ts[0] != self.timestamps[self.start]:
            self.start += 1
        poses = kf_poses.detach().to("cpu", torch.float64)
        del self.timestamps[self.start:], self.poses[self.start:]
        self.timestamps += ts
        self.poses += [poses[i] for i in range(len(ts))]


def run_sequence(dataset, slam_cfg, model, first_frame=1, max_frames=None, progress=None):
    """Drive `ComoSeq.iter` over `dataset` (frames first_frame .. : the reference's loop hands frame 0 to the renderer only).
    Returns (loop object, KeyframeHistory, list of the request of every frame)."""
    from como_amd.odom.sequential import ComoSeq
    dev = slam_cfg["mapping"]["device"]
    odo = ComoSeq(slam_cfg, dataset.intrinsics.clone(), tuple(dataset.img_size), model)
    hist, kinds = KeyframeHistory(), []
    n = len(dataset) if max_frames is None else min(len(dataset), first_frame + max_frames)
    for idx in range(first_frame, n):
        timestamp, rgb = dataset[idx]
        kind = odo.iter(timestamp, rgb.unsqueeze(0).to(dev))
        kinds.append(kind)
        if kind == "keyframe" or (kind == "init" and odo.mapping.is_init):
            hist.update(odo.mapping.kf_timestamps, odo.mapping.kf_poses)
        if progress and (idx - first_frame + 1) % progress == 0:
            print(f"frame {idx + 1}/{n}: keyframes {len(hist.timestamps)}", file=sys.stderr)
    if odo.mapping.is_init:
        hist.update(odo.mapping.kf_timestamps, odo.mapping.kf_poses)
    return odo, hist, kinds


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--dataset_type", type=str, required=True)
    ap.add_argument("--dataset_dir", type=str, required=True)
    ap.add_argument("--config", type=str, default="./config/como.yml")
    ap.add_argument("--out", type=str, default=None, help="trajectory file (default ./results/<save_traj_name>.txt, as the reference)")
    ap.add_argument("--device", type=str, default=None, help="one device for tracking and mapping (default: mapping.device of the YAML)")
    ap.add_argument("--img_size", type=int, nargs=2, default=DEFAULT_IMG_SIZE)
    ap.add_argument("--pix_dtype", type=str, default=None, choices=["float", "double"])
    ap.add_argument("--model_path", type=str, default=None)
    ap.add_argument("--random_weights", type=int, default=None, metavar="SEED")
    ap.add_argument("--first_frame", type=int, default=1, help="1 = the reference's loop (frame 0 only initialises its renderer)")
    ap.add_argument("--max_frames", type=int, default=None)
    ap.add_argument("--progress", type=int, default=0)
    args = ap.parse_args(argv)
    torch.manual_seed(0)
    dataset = get_dataset(args.dataset_type, list(args.img_size), args.dataset_dir)
    cfg = load_slam_cfg(args.config, args.device, args.pix_dtype)
    if args.model_path:
        cfg["mapping"]["model_path"] = args.model_path
    dev = cfg["mapping"]["device"]
    model = load_model(cfg["mapping"], dev, args.random_weights)
    t0 = time.time()
    odo, hist, kinds = run_sequence(dataset, cfg, model, args.first_frame, args.max_frames, args.progress)
    torch.cuda.synchronize(dev)
    dt = time.time() - t0
    out = args.out or os.path.join("./results", dataset.save_traj_name + ".txt")
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    if hist.timestamps:
        save_traj(out, hist.timestamps, torch.stack(hist.poses))
    else:
        open(out, "w").close()
    stem, ext = os.path.splitext(out)
    if odo.est_poses:
        save_traj(stem + "_frames" + ext, odo.timestamps, torch.cat([p.detach().reshape(1, 4, 4).cpu().double() for p in odo.est_poses]))
    print(f"Saved trajectory. {len(kinds)} frames in {dt:.2f} s ({len(kinds) / max(dt, 1e-9):.1f} frames/s), "
          f"{len(hist.timestamps)} keyframes -> {out}")
    return out


if __name__ == "__main__":
    main()
