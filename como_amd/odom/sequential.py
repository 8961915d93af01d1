"""Single-process odometry loop: the reference's sequential mode (como/odom/sequential/{ComoSeq,MappingSeq,TrackingSeq}.py)
without the GUI.  Per frame: track against the newest keyframe (once mapping is initialised), hand the tracker's request
(keyframe / one-way frame / initialisation frame) to the mapper, run ONE mapping GN iteration, and give the tracker the
refreshed keyframe reference (pose, affine parameters, dense depth) -- the order of ComoSeq.iter (ComoSeq.py:41-71)."""
import os
import time

import torch

from como_amd.odom.Mapping import Mapping
from como_amd.odom.Tracking import Tracking


def transfer_data(data, device, dtype):
    """Tensors of a message tuple to the receiver's device / dtype (como/utils/multiprocessing.py:16-21)."""
    return tuple(d.to(device=device, dtype=dtype, copy=False) if torch.is_tensor(d) else d for d in data)


class TrackingSeq(Tracking):
    def track(self, data):
        return self.handle_frame(data)


class MappingSeq(Mapping):
    def map(self, data):
        """MappingSeq.py:11-47 -> (kf_viz_data, kf_ref_data)"""
        kf_viz_data = kf_ref_data = None
        kf_updated = False
        if data is not None:
            if data[0] in ("one-way", "keyframe") and all(torch.is_tensor(t) and t.device == torch.device(self.device) for t in data[1:4]):
                pass        # (same device: handle_tracking_data widens the tracker's float32 values inside its fused kernels)
            else:
                data = transfer_data(data, self.device, self.dtype)
            if not self.is_init:
                if data[0] == "init":
                    kf_updated = self.attempt_two_frame_init(data[1], data[2])
            else:
                kf_viz_data, kf_updated = self.handle_tracking_data(data)
        if self.is_init and not self.converged:
            self.converged = self.iterate()
            kf_updated = True
        if self.is_init and (time.time() - self.last_kf_send_time > 1.0):
            kf_viz_data = self.get_kf_viz_data()
        if data is not None and data[0] == "keyframe":
            kf_viz_data = self.get_kf_viz_data()
        if kf_updated:
            kf_ref_data = self.get_kf_ref_data()
        return kf_viz_data, kf_ref_data


class ComoSeq:
    """Headless ComoSeq: `iter(timestamp, rgb)` per frame; the tracked world poses accumulate in `timestamps` / `est_poses`."""

    def __init__(self, slam_cfg, intrinsics, img_size, model=None):
        self.tracking = TrackingSeq(slam_cfg["tracking"], intrinsics.clone(), img_size)
        mcfg = slam_cfg["mapping"]
        # the next iteration's full-image median streamed into the host-bound hand-over of a one-way frame (Mapping.median_ahead_now):
        # on unless the configuration says otherwise (`median_ahead: false | gap | end`; COMO_MEDIAN_AHEAD_MODE=off|gap|end for A/B)
        mode = os.environ.get("COMO_MEDIAN_AHEAD_MODE", "gap")
        if "median_ahead" not in mcfg:
            mcfg = dict(mcfg, median_ahead=(False if mode in ("off", "0", "") else mode))
        self.mapping = MappingSeq(mcfg, intrinsics.clone())
        self.tracking.setup()
        self.mapping.setup(model)
        self.timestamps, self.est_poses = [], []
        self.last_kf_viz = None
        if torch.device(self.tracking.device) == torch.device(self.mapping.device):
            # one process, one device: while the tracker waits for a frame's result the mapper prepares the window that frame
            # would need as a one-way frame (Mapping.speculate_one_way)
            self.tracking.while_waiting = self.mapping.speculate_one_way
            if os.environ.get("COMO_KF_IMAGES_AHEAD", "1") != "0" and slam_cfg["mapping"].get("track_ref", {}).get("num_keyframes", 1) == 1:
                # ... and the tracker prepares its pyramids of a new keyframe's image while the insertion's network runs
                self.mapping.while_network_runs = self.tracking.prepare_kf_images
            if self.mapping.cfg.get("median_ahead", False) == "gap":
                self.tracking.after_decision = self.mapping.median_ahead_now

    def iter(self, timestamp, rgb):
        trk, mp = self.tracking, self.mapping
        if mp.is_init:
            # (the tracker gets its own copy of the frame: the conversion to its element type already is one -- a clone first only
            # when there is no conversion)
            # -- and not at all when the tracker runs its frame graph: that copies the frame into the graph's input buffer before
            # anything else and clones it again if it hands the frame to the mapper)
            own = rgb if (rgb.dtype != trk.dtype or trk.copies_its_input(rgb)) else rgb.clone()
            viz, to_map = trk.track(transfer_data((timestamp, own), trk.device, trk.dtype))
            self.timestamps.append(viz[0])
            self.est_poses.append(viz[1])
        else:
            to_map = ("init", timestamp, rgb.clone())
        kf_viz, kf_ref = mp.map(to_map)
        if kf_ref is not None:
            if trk.mapping_init and kf_ref[0][-1] == trk.kf_received_ts:
                # same keyframe image(s) as last time: the tracker only reads the poses and the depth (update_kf_reference),
                # so the window's colour images are not converted to its element type again on every frame
                kf_ref = (kf_ref[0], None) + tuple(kf_ref[2:])
            pose, aff = kf_ref[2], kf_ref[3]
            if (torch.is_tensor(pose) and torch.is_tensor(aff) and pose.is_cuda and aff.is_cuda and pose.dtype == aff.dtype != trk.dtype and
                    pose.device == aff.device == torch.device(trk.device)):
                # the keyframe pose and affine parameters in the tracker's element type: both conversions in one launch
                p2, a2 = torch.empty_like(pose, dtype=trk.dtype), torch.empty_like(aff, dtype=trk.dtype)
                torch._foreach_copy_([p2, a2], [pose, aff])
                kf_ref = tuple(kf_ref[:2]) + (p2, a2) + tuple(kf_ref[4:])
            trk.update_kf_reference(transfer_data(kf_ref, trk.device, trk.dtype))
        if kf_viz is not None:
            self.last_kf_viz = kf_viz
        return to_map[0] if to_map is not None else None
