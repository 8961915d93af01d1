"""Headless keyframe-window bundle adjustment: the call sequence of the reference's `Mapping.iterate`
(como/odom/Mapping.py:760-968) without the GUI / process plumbing around it.

    scaffold (project landmarks)          Mapping.prep_geometry_scaffold  :603-659   HIP win_scaffold   (csrc/window.hip)
    dense reference points                Mapping.prep_dense_ref          :661-699   HIP dense_ref      (csrc/densify.hip)
    photometric normal equations          create_photo_system   backend/photo.py:236-353   HIP ba_*     (csrc/ba.hip)
    priors                                Mapping.iterate                 :809-917   HIP win_priors     (csrc/window.hip)
    solve                                 lin_sys.solve_system  linear_system.py:101-112   HIP chol_*   (csrc/chol.hip)
    update                                lin_sys.update_vars   linear_system.py:115-152   HIP win_update

`fused=True` (default) runs the 31-launch HIP chain above (two stream branches, see linearize_fused); `fused=False` runs the same iteration through the
reference-signature mirrors (torch ops for the O(B m) parts) -- both are checked against the golden vectors.
State tensors live on one GPU and are updated in place (fixed addresses: the iteration is hipGraph-capturable).
`pix_dtype` is the element type of the per-pixel path (float32 = mixed precision, float64 = config/como.yml:28);
the system (H, g, poses, landmarks, priors, solve) is always float64.
"""
import ctypes
import os

import torch

import como_amd.odom.backend.linear_system as lin_sys
import como_amd.odom.backend.photo as photo
import como_amd.odom.backend.sparse_map as smap
from como_amd import _lib
from como_amd.geometry.camera import backprojection
from como_amd.odom.backend.dense_ref import BAND_MEDIAN, dense_reference_factored, dense_reference_planes, full_image_median, median_passes
from como_amd.odom.backend.graph_pair_construction import setup_photometric_pairs
from como_amd.odom.factors.depth_prior import log_depth_prior
from como_amd.odom.factors.gp_priors import gp_ml_cost, mean_log_depth_cost
from como_amd.odom.factors.pixel_prior import pixel_prior_cost
from como_amd.odom.factors.pose_prior_factors import linearize_pose_prior
from como_amd.odom.factors.scalar_prior_factors import linearize_multi_scalar_prior, linearize_scalar_prior

DEFAULT_CFG = {
    "photo_construction": {"nonmax_suppression_window": 4, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                           "degrees_thresh": 0.0},
    "sigmas": {"photo": 1e-1, "mean_depth_prior": 1e-2, "scale_prior": 1e-4, "pose_prior": 1e-6},
}


_FUSE_PASS1 = __import__("os").environ.get("COMO_BA_FUSE_PASS1", "1") != "0"      # 0: the residual pass as its own launch (A/B)
_REUSE_TOPOLOGY = __import__("os").environ.get("COMO_BA_REUSE", "1") != "0"      # (measurement switch)
_SPECULATE = __import__("os").environ.get("COMO_BA_SPECULATE", "1") != "0"      # 0: no pair table is built ahead of the tracker's decision (A/B)
SPEC_STATS = {"built": 0, "adopted": 0}     # pair tables built ahead / adopted by a re-target (process-wide; bench.py reports them)
_REUSE_WORKSPACES = __import__("os").environ.get("COMO_BA_REUSE_WS", "1") != "0"  # (measurement switch)
# The sequential loop's full-image median of iteration t+1 streamed right after iteration t's update, beside the tracker, instead of
# beside iteration t+1's block kernel (`WindowBA.median_ahead`, set by Mapping).  0: off (A/B); 2: every adopted median is ALSO computed
# the usual way and compared bit for bit (one host synchronisation per iteration: tests only) -- AHEAD_STATS["mismatch"].
_MEDIAN_AHEAD = int(__import__("os").environ.get("COMO_MEDIAN_AHEAD", "1"))
AHEAD_STATS = {"issued": 0, "adopted": 0, "checked": 0, "mismatch": 0}


_ARANGE = {}


def _ar(n, dev, dtype=torch.long):
    """torch.arange(n) as a READ-ONLY view of one cached ramp per (device, dtype): the window is rebuilt on every keyframe and
    one-way frame of the sequential loop, each rebuild with half a dozen tiny arange launches."""
    key = (str(dev), dtype)
    t = _ARANGE.get(key)
    if t is None or t.shape[0] < n:
        t = _ARANGE[key] = torch.arange(max(4096, 2 * n), device=dev, dtype=dtype)
    return t[:n]


class WindowBA:
    def __init__(self, state, cfg=None, pix_dtype=torch.float32, window_full=True, shard=None, fused=True, prev=None, rec_capacity=None,
                 band_median=None):
        """state: dict as produced by como_amd.synth.make_window (plus K_mm_inv / L_mm / Knm_Kmminv).
        shard: como_amd.dist.Shard for the one-process-per-GPU data-parallel mode (None = single GPU).
        prev: the WindowBA this one replaces (the sequential loop rebuilds the window on every keyframe): when
        the KEYFRAME SET is the same -- the very same image-stack and correspondence-mask tensors -- everything that depends on
        it only (reference-pixel selection, landmark remap / index tables) is taken over instead of being recomputed.
        rec_capacity: the largest number of one-way frames this keyframe set will see (Mapping: graph.num_one_way_frames): the state
        buffer, the system buffers and the per-frame scratch are sized for it once, so that `retarget` -- same keyframes, another
        set of one-way frames, 55 % of the sequential loop's frames -- only rewrites the frame count, the index tables that move
        with it and the pair table instead of building a new object (None: exactly the frames of `state`)."""
        # band_median: the full-image median of sub-selected windows from the pixels whose depth can still cross the median
        # (csrc/densify.hip depth_band_kernel) -- wins when the log-depths barely move between iterations (a converging window: 1-2 %
        # of the pixels are re-evaluated); in the sequential loop every iteration follows a new frame and ~64 % of the pixels stay
        # candidates (scripts/gpu_odometry_bench.py `band_median`): there the plain streaming pass on the matrix cores is faster.
        # None: the module default (dense_ref.BAND_MEDIAN).
        self.band_median = band_median
        self.median_ahead = False                            # (Mapping turns it on for the sequential loop)
        self._ahead = None
        self.shard = shard
        self._prev = prev
        self._src_kf_img, self._src_mask = state["kf_img_and_grads"], state["correspondence_mask"]
        if shard is not None and not fused:
            raise RuntimeError("como_amd: the multi-GPU window BA runs the fused chain only")
        self.events = None
        self.graph = None
        self.cfg = cfg or DEFAULT_CFG
        self.dev = state["kf_poses"].device
        self.dt = torch.float64
        self.pix_dtype = pix_dtype
        # the fused chain also serves the FILLING window (fewer keyframes than the graph holds): its only difference is the
        # scale prior -- mean predicted log-depth of keyframe 0 instead of landmark anchors (Mapping.py:884-917)
        self.fused = fused and (window_full or state.get("init_scale_anchor") is not None)
        f64 = lambda t: t.to(self.dt).contiguous()
        dev = self.dev
        B, _, self.Himg, self.Wimg = state["kf_img_and_grads"].shape
        self.B = B
        self.m = state["coords_m"].shape[1]
        rec_poses = state.get("recent_poses")
        nrec = int(rec_poses.shape[0]) if rec_poses is not None else 0      # one-way frames (Mapping.add_one_way_frame)
        self.rec_cap = max(nrec, int(rec_capacity or 0))
        self.intrinsics = f64(state["intrinsics"])
        # the state the iteration updates -- all frame poses | affine params (keyframes first) | landmarks | median depths -- in
        # ONE buffer (the views below alias it): the sequential loop publishes it after every iteration with one copy
        # (`snapshot_state`) instead of one per tensor.  Frame-indexed parts are laid out for the CAPACITY B + rec_cap.
        Fc, nP = B + self.rec_cap, int(state["P_m"].shape[0])
        self.Fcap = Fc
        ev = lambda n: (n + 1) // 2 * 2                       # 16-byte steps
        o_aff, o_P = 16 * Fc, 16 * Fc + ev(2 * Fc)
        o_med = o_P + ev(3 * nP)
        self._state_off = (o_aff, o_P, o_med, nP)
        self.state_flat = torch.zeros((o_med + ev(B),), device=dev, dtype=self.dt)
        self.P_m = self.state_flat[o_P:o_P + 3 * nP].view(nP, 3)
        self.median_depths = self.state_flat[o_med:o_med + B]
        self._load_frames(state)
        self.correspondence_mask = state["correspondence_mask"]
        self._corr_host = state.get("corr_host")             # numpy copy of it, when the caller keeps one (Mapping does)
        self.obs_ref_mask = state["obs_ref_mask"].contiguous()
        self.pm_first_obs = f64(state["pm_first_obs"])
        self.L_mm = f64(state["L_mm"])
        self.K_mm_inv = f64(state["K_mm_inv"])
        self.kf_timestamps = state["kf_timestamps"]
        if not self._load_images(state):
            raise RuntimeError("como_amd WindowBA: image stacks could not be addressed")       # (cannot happen: the concatenating form always applies)
        pix = lambda name: self._pix(state, name)
        self.Kt = pix("Knm_Kmminv").to(pix_dtype).reshape(B, self.Himg * self.Wimg, self.m).contiguous()
        self.K_pix = self.intrinsics[0].to(pix_dtype).contiguous()
        self.window_full = window_full
        self.pose_anchor = f64(state["pose_anchor"]) if "pose_anchor" in state else self.kf_poses[0:1].clone()
        self.aff_anchor = torch.zeros((1, 2, 1), device=dev, dtype=self.dt)
        self.P_m_anchors = f64(state["P_anchor"]) if "P_anchor" in state else self.P_m[self.correspondence_mask[0]].clone()
        self.init_scale_anchor = state.get("init_scale_anchor")
        self._prepare_topology()

    def _pix(self, state, name):
        # (the sequential loop hands over mirrors that are already in pix_dtype -- Mapping._cat -- so that a rebuild does not
        # convert the whole K~ window and every image stack again)
        t = state.get(name + "_pix")
        return t if t is not None and t.dtype == self.pix_dtype and t.shape == state[name].shape else state[name]

    def _load_frames(self, state, only_recent=False):
        """The frame set of `state` into the capacity-sized state buffer: frame count, the views of the first F frames, the values.
        only_recent (`retarget`): the keyframe poses / affine parameters / landmarks / medians of `state` ARE this buffer's current
        values (Mapping.iterate published them from it and nothing but an iteration changes them): only the one-way frames' rows
        are written."""
        B, dev = self.B, self.dev
        f64 = lambda t: t.to(self.dt).contiguous()
        rec_poses = state.get("recent_poses")
        nrec = int(rec_poses.shape[0]) if rec_poses is not None else 0
        if nrec > self.rec_cap:
            raise RuntimeError("como_amd WindowBA: more one-way frames than the window was sized for")
        F_ = self.F = B + nrec
        o_aff, o_P, o_med, nP = self._state_off
        self._state_layout = (("poses", 0, (F_, 4, 4)), ("aff", o_aff, (F_, 2)), ("P_m", o_P, (nP, 3)), ("median", o_med, (B,)))
        self.poses_all = self.state_flat[:16 * F_].view(F_, 4, 4)
        self.aff_all = self.state_flat[o_aff:o_aff + 2 * F_].view(F_, 2)
        if not only_recent:
            self._ahead = None                               # (a median streamed ahead belongs to the keyframe state it was computed from)
            self.poses_all[:B] = f64(state["kf_poses"])
            self.aff_all[:B] = f64(state["kf_aff_params"]).reshape(B, 2)
        self.kf_poses = self.poses_all[:B]
        self.kf_aff_params = self.aff_all[:B].view(B, 2, 1)
        self.recent_poses = self.poses_all[B:]
        self.recent_aff_params = self.aff_all[B:].view(nrec, 2, 1)
        if nrec:
            self.poses_all[B:] = f64(rec_poses)
            self.aff_all[B:] = f64(state["recent_aff_params"]).reshape(nrec, 2)
            self.recent_timestamps = state["recent_timestamps"]
        else:
            self.recent_timestamps = torch.empty((0,), device=dev, dtype=self.dt)
        if only_recent:
            return
        self.P_m.copy_(state["P_m"])
        if "median_depth_init" in state:
            self.median_depths.copy_(state["median_depth_init"])
        else:
            self.median_depths.fill_(1.0)

    def _load_images(self, state, allow_concat=True):
        """Per-pixel image data in pix_dtype: the keyframe stacks and the one-way frames' stacks, addressed as ONE buffer (targets
        are element offsets from the keyframe stack).  Returns False when that needs a concatenation and allow_concat is off."""
        nrec = self.F - self.B
        kf_img = self._pix(state, "kf_img_and_grads")
        rec_img = self._pix(state, "recent_img_and_grads") if nrec else None
        self._rec_img, self._rec_img_off = None, None
        pd = self.pix_dtype
        if (nrec and rec_img.dtype == pd and kf_img.dtype == pd and rec_img.is_contiguous() and kf_img.is_contiguous() and
                rec_img.device == kf_img.device and (rec_img.data_ptr() - kf_img.data_ptr()) % kf_img.element_size() == 0):
            # both stacks stay where they are (the sequential loop's sliding buffers): the one-way targets are addressed relative to
            # the keyframe stack by their element offset, whatever its sign -- no 122 MB concatenation per rebuild
            self.img, self._rec_img = kf_img, rec_img
            self._rec_img_off = (rec_img.data_ptr() - kf_img.data_ptr()) // kf_img.element_size()
            return True
        if nrec and not allow_concat:
            return False
        imgs = kf_img if not nrec else torch.cat((kf_img, rec_img.to(kf_img.dtype)))
        self.img = imgs.to(pd).contiguous()
        return True

    def retarget(self, state, window_full=None):
        """The SAME keyframe set with another set of one-way frames (`Mapping.add_one_way_frame`: 55 % of the sequential loop's frames
        rebuilt the whole window object for it: ~0.65 ms of host time and ~55 small launches while the GPU waited).  Everything
        sized by the frame count lives in capacity-sized buffers (`rec_capacity`), so this only loads the frames' values, moves the
        landmark columns of the system (they sit behind the frame blocks: D = 8 F + 3 L), rebuilds the pair table and patches the
        kernels' argument block; scratch, band-median state, reference pixels and landmark tables stay as they are.
        Returns False -- nothing touched that matters -- when this window cannot take the state (another keyframe set, not the
        fused single-GPU chain, more frames than its capacity, image stacks that would have to be concatenated): build a new one."""
        rec = state.get("recent_poses")
        nrec = int(rec.shape[0]) if rec is not None else 0
        if not (self.fused and self.shard is None and self.dev.type == "cuda" and _REUSE_TOPOLOGY and nrec <= self.rec_cap and
                state["kf_img_and_grads"] is self._src_kf_img and state["correspondence_mask"] is self._src_mask and
                int(state["P_m"].shape[0]) == self.L and (window_full is None or window_full == self.window_full) and
                getattr(self, "win_args", None) is not None):
            return False
        F_old = self.F
        self.F = self.B + nrec
        if not self._load_images(state, allow_concat=False):
            self.F = F_old
            return False
        self.graph = None                                    # (a captured iteration holds the old frame count)
        self._load_frames(state, only_recent=state.get("_published_by") is self)
        self._finish_topology(first=False)
        return True

    def speculate(self, kf_ts, rec_ts, rec_ptr):
        """Build -- while the host would only wait for the tracker's read-back -- the pair table of THIS keyframe set with the
        one-way frames `rec_ts` (host floats: what `Mapping.add_one_way_frame` would leave behind if the frame being tracked
        becomes a one-way frame), their image stack starting at device address `rec_ptr`.  The table (numpy + ONE pinned upload,
        on a stream of its own so that the tracker's read-back does not queue behind it) is only kept aside; `_build_pair_table`
        takes it when the topology a `retarget` arrives at is exactly this one, and builds its own otherwise.  A one-way frame
        spent 0.17 ms of host time there between the tracker's decision and the first kernel of its iteration."""
        self._spec = None
        pc = self.cfg["photo_construction"]
        hn = getattr(self, "_host_np", None)
        nrec = len(rec_ts)
        why = ("off" if not (_SPECULATE and _REUSE_TOPOLOGY) else "not the fused single-GPU chain" if not (
            self.fused and self.shard is None and self.dev.type == "cuda" and getattr(self, "win_args", None) is not None) else
            "no host mirror of the landmark tables" if hn is None else "frame count" if not 0 < nrec <= self.rec_cap else
            "image stack address unknown" if rec_ptr is None or getattr(self, "_rec_img_off", None) is None else
            "pose-dependent pair graph" if pc.get("radius_thresh", 0.0) > 0.0 and pc.get("degrees_thresh", 0.0) > 0.0 else None)
        if why is not None:
            SPEC_STATS["skipped: " + why] = SPEC_STATS.get("skipped: " + why, 0) + 1
            return
        import numpy as np
        from como_amd.odom.backend.graph_pair_construction import get_backward_edges, get_forward_edges, get_one_way_temporal_neighbors
        B, dev = self.B, self.dev
        rf, tf = get_forward_edges(B)
        rb, tb = get_backward_edges(B)
        ow_kf, ow_t = get_one_way_temporal_neighbors([float(t) for t in kf_ts], [float(t) for t in rec_ts])
        ref, tgt = rf + rb, tf + tb
        off = (int(rec_ptr) - self.img.data_ptr()) // self.img.element_size()
        lm_start = 8 * B + 8 * nrec
        ex = {"landmark_inds": hn["point_inds"] + lm_start,
              "fix_inds": (lm_start + 3 * hn["fix_idx"][:, None] + np.arange(3)[None]).reshape(-1)}
        recent_inds = _ar(8 * (B + nrec), dev)[8 * B:].reshape(nrec, 8)
        recent_inds._como_ramp = True
        side = getattr(self, "_spec_stream", None)
        if side is None:
            side = self._spec_stream = torch.cuda.Stream(device=dev)
        stack = 3 * self.channels * self.Himg * self.Wimg
        with torch.cuda.stream(side):
            table = photo.PairTable(ref + ow_kf, tgt + ow_t, [False] * len(ref) + [True] * len(ow_kf), B, self.kf_inds, recent_inds, None,
                                    stack, off, dev, channels=self.channels, landmark_inds_host=ex["landmark_inds"], extra_i64=ex)
            ev = torch.cuda.Event()
            ev.record(side)
        SPEC_STATS["built"] += 1
        self._spec = {"pairs": ([ref, tgt], [ow_kf, ow_t]), "nrec": nrec, "off": off, "table": table, "event": ev, "host_np": hn}

    @staticmethod
    def _fixed_remap(mask, m):
        """sparse_map.get_batch_remap_function (reference sparse_map.py:73-112) for a window in which every keyframe observes
        exactly m landmarks -- what the reference's own `landmark_to_batched_3d_point_inds` (.view(num_kf, -1)) requires --
        WITHOUT the host synchronisations of torch.nonzero / .item(): the window is rebuilt on every one-way frame and every
        keyframe of the sequential loop, and each synchronisation stalls the host behind the whole queued GPU work.
        Returns (remap, landmark_ids (B*m,2) as the reference's list, lm_of (B,m) landmark index per slot, ascending)."""
        B = mask.shape[0]
        lm_of = torch.argsort((~mask).to(torch.uint8), dim=1, stable=True)[:, :m].contiguous()     # the m True columns, in order
        rows = _ar(B, mask.device).repeat_interleave(m)
        landmark_ids = torch.stack((rows, lm_of.reshape(-1)), dim=1)

        def remap(variable, default_val=-1):           # variable (B,L,...) -> (B,m,...); every slot is filled
            idx = lm_of.reshape((B, m) + (1,) * (variable.dim() - 2)).expand((B, m) + tuple(variable.shape[2:]))
            return torch.gather(variable, 1, idx)

        return remap, landmark_ids, lm_of

    def _host_tables(self, mask):
        """The landmark index tables of `_prepare_topology` from a HOST copy of the correspondence mask (Mapping keeps one: numpy
        bool (B,L)), built with numpy and uploaded in ONE pinned copy -- ~25 small device launches (argsort, gathers, scatters,
        arange slices) of every keyframe insertion's window rebuild otherwise.  False when there is no usable host copy."""
        import numpy as np
        B, m, L, dev = self.B, self.m, self.L, self.dev
        if mask is None or dev.type != "cuda" or tuple(mask.shape) != (B, L) or not (mask.sum(axis=1) == m).all():
            return False
        from como_amd.utils.hostlist import to_device
        lm_of = np.stack([np.nonzero(mask[b])[0] for b in range(B)]).astype(np.int64)            # (B,m) landmark of every slot
        point_inds = (3 * np.repeat(lm_of, 3, axis=1) + np.tile(np.arange(3), m)[None]).astype(np.int64)
        first_obs = np.argmax(mask, axis=0).astype(np.int64)                                      # first observer keyframe (L,)
        fom = np.zeros_like(mask)
        fom[first_obs, np.arange(L)] = True
        first_obs_mask = np.take_along_axis(fom, lm_of, axis=1)
        slot_of = np.full((B, L), -1, dtype=np.int32)
        np.put_along_axis(slot_of, lm_of, np.arange(m, dtype=np.int32)[None].repeat(B, 0), axis=1)
        first_slot = slot_of[first_obs, np.arange(L)]
        # one byte block: [int64: lm_of | point_inds] [int32: lm_ids | first_frame | first_slot | pad] [bool: first_obs_mask]
        i64 = np.concatenate((lm_of.reshape(-1), point_inds.reshape(-1)))
        i32 = np.concatenate((lm_of.reshape(-1).astype(np.int32), first_obs.astype(np.int32), first_slot.astype(np.int32)))
        if i32.size % 2:
            i32 = np.concatenate((i32, np.zeros(1, np.int32)))
        raw = np.concatenate((i64.view(np.uint8), i32.view(np.uint8), first_obs_mask.reshape(-1).view(np.uint8)))
        d = to_device(raw, torch.uint8, dev)
        o1, o2 = i64.nbytes, i64.nbytes + i32.nbytes
        d64, d32 = d[:o1].view(torch.int64), d[o1:o2].view(torch.int32)
        lm_dev = d64[:B * m].view(B, m)
        self.point_inds = d64[B * m:].view(B, 3 * m)
        self.lm_ids = d32[:B * m].view(B, m)
        self.first_frame = d32[B * m:B * m + L]
        self.first_slot = d32[B * m + L:B * m + 2 * L]
        self.first_obs_mask = d[o2:o2 + B * m].view(torch.bool).view(B, m)
        self.fix_idx = lm_dev[0]
        self._host_np = {"point_inds": point_inds, "fix_idx": lm_of[0].copy()}      # (for the tables that move with the frame count)

        def remap(variable, default_val=-1):           # variable (B,L,...) -> (B,m,...); every slot is filled
            idx = lm_dev.reshape((B, m) + (1,) * (variable.dim() - 2)).expand((B, m) + tuple(variable.shape[2:]))
            return torch.gather(variable, 1, idx)

        self.remap = remap
        return True

    # ---- things that change only when the keyframe set changes ---------------------------------------------------
    _KF_SET_ATTRS = ("coords_n", "n_total", "pixidx", "channels", "vals_n", "n", "remap", "point_inds", "L", "kf_inds", "fix_idx",
                     "lm_ids", "first_obs_mask", "first_frame", "first_slot", "_host_np")

    def _same_keyframe_set(self, prev):
        return (prev is not None and _REUSE_TOPOLOGY and prev.shard is None and self.shard is None and prev._src_kf_img is self._src_kf_img and
                prev._src_mask is self._src_mask and prev.B == self.B and prev.m == self.m and prev.pix_dtype == self.pix_dtype and
                prev.P_m.shape == self.P_m.shape and prev.dev == self.dev and
                prev.cfg["photo_construction"]["nonmax_suppression_window"] == self.cfg["photo_construction"]["nonmax_suppression_window"])

    def _prepare_topology(self):
        B, dev, m = self.B, self.dev, self.m
        prev, self._prev = self._prev, None                  # (no chain of old windows kept alive)
        if self._same_keyframe_set(prev):
            for a in self._KF_SET_ATTRS:
                setattr(self, a, getattr(prev, a, None))
            self.idle = False
            self._inherit = prev                             # (its scratch workspaces too: _prepare_fused)
            self._finish_topology()
            self._inherit = None
            return
        if (prev is not None and _REUSE_TOPOLOGY and prev.shard is None and self.shard is None and prev.dev == self.dev and
                prev.pix_dtype == self.pix_dtype):
            self._inherit = prev                             # another keyframe set: only its scratch workspaces (_prepare_fused)
        w = self.cfg["photo_construction"]["nonmax_suppression_window"]
        coords_n, _ = smap.subselect_pixels(self.img[:B], w)                   # Mapping.py:665-668
        self.coords_n = coords_n
        self.n_total = coords_n.shape[1]
        self.pixidx = (coords_n[..., 0] * self.Wimg + coords_n[..., 1]).to(torch.int32).contiguous()
        # vals_n (B,n) gray / (B,n,c) colour = kf_img_and_grads[b, :c, row, col] (Mapping.py:677-681)
        c = self.channels = self.img.shape[1] // 3
        vals = torch.gather(self.img[:B, :c].reshape(B, c, -1), 2, self.pixidx.long()[:, None].expand(-1, c, -1))
        self.vals_n = (vals[:, 0] if c == 1 else vals.transpose(1, 2)).contiguous()
        self.idle = False
        if self.shard is not None:
            # One process per GPU: this rank owns a contiguous range of the reference pixels of EVERY keyframe -- to all
            # per-pixel kernels (dense reference, residual, blocks) it simply is a window with fewer reference pixels; only
            # the collectives (median histograms, per-pair sums) know about the other ranks.
            pb, pe = self.shard.pixel_range(self.n_total)
            self.idle = pe <= pb                           # more ranks than 64-pixel tiles: takes part in the collectives only
            if self.idle:
                pb, pe = 0, 1                              # placeholder arrays of one pixel; no pixel kernel is launched
            self.pixidx = self.pixidx[:, pb:pe].contiguous()
            self.vals_n = self.vals_n[:, pb:pe].contiguous()
            self.row_range = self.shard.row_range(self.Himg * self.Wimg)
        self.n = self.pixidx.shape[1]
        L = self.P_m.shape[0]
        self.L = L
        self.kf_inds = _ar(8 * B, dev).reshape(B, 8)
        if self._host_tables(getattr(self, "_corr_host", None)):
            self._finish_topology()
            self._inherit = None
            return
        self.remap, landmark_ids, lm_of = self._fixed_remap(self.correspondence_mask, m)
        self.point_inds = lin_sys.landmark_to_batched_3d_point_inds(landmark_ids, B)
        # index lists of the oldest keyframe's landmarks (their anchors, Mapping.py:884-898): precomputed -- boolean-mask
        # indexing inside the iteration would synchronise with the host (and cannot be captured in a hipGraph)
        self.fix_idx = lm_of[0]
        self.lm_ids = (self.point_inds[:, ::3] // 3).to(torch.int32).contiguous()          # (B,m)
        first_obs = torch.argmax(self.correspondence_mask.int(), dim=0)                    # first observer keyframe
        fom = torch.zeros_like(self.correspondence_mask)
        fom[first_obs, _ar(L, dev)] = True
        self.first_obs_mask = self.remap(fom, False)
        slot_of = torch.full((B, L), -1, device=dev, dtype=torch.int32)
        slot_of.scatter_(1, self.lm_ids.long(), _ar(m, dev, torch.int32)[None].expand(B, m).contiguous())
        self.first_frame = first_obs.to(torch.int32).contiguous()
        self.first_slot = slot_of[first_obs, _ar(L, dev)].contiguous()
        self._finish_topology()
        self._inherit = None

    # ---- ... and what also depends on the number of one-way frames --------------------------------------------------------
    def _finish_topology(self, first=True):
        """first: the buffers are created (sized for the capacity B + rec_cap frames); a `retarget` only re-derives what moves
        with the frame count."""
        B, dev, L = self.B, self.dev, self.L
        nrec = self.F - B
        self.dim = 8 * B + 8 * nrec + 3 * L
        self.recent_inds = _ar(8 * self.F, dev)[8 * B:].reshape(nrec, 8) if nrec else \
            torch.empty((0), device=dev, dtype=torch.long)
        self.recent_inds._como_ramp = True                   # (frame f owns rows 8 f .. 8 f + 7: photo.PairTable builds them on the host)
        self.kf_inds._como_ramp = True
        self.frame_inds = _ar(8 * self.F, dev).reshape(self.F, 8)
        self.lm_start = 8 * B + 8 * nrec
        self.landmark_inds_flat = _ar(self.lm_start + 3 * L, dev)[self.lm_start:].reshape(L, 3)
        hn = getattr(self, "_host_np", None)
        self._host_extra = None
        if hn is not None and dev.type == "cuda":
            # the two landmark index tables that move with the frame count, built on the host (Mapping's host copy of the
            # correspondence mask, _host_tables) and uploaded WITH the pair table: no device launch for them
            import numpy as np
            lmi = hn["point_inds"] + self.lm_start
            fix = (self.lm_start + 3 * hn["fix_idx"][:, None] + np.arange(3)[None]).reshape(-1)
            self._host_extra = {"landmark_inds": lmi, "fix_inds": fix}
        elif first:
            self.landmark_inds = (self.point_inds + self.lm_start).contiguous()
            self.fix_inds_flat = torch.empty((3 * int(self.fix_idx.numel()),), device=dev, dtype=torch.long)
            torch.index_select(self.landmark_inds_flat, 0, self.fix_idx, out=self.fix_inds_flat.view(-1, 3))
        else:
            torch.add(self.point_inds, self.lm_start, out=self.landmark_inds)      # (same address: the argument blocks keep it)
            torch.index_select(self.landmark_inds_flat, 0, self.fix_idx, out=self.fix_inds_flat.view(-1, 3))
        pc = self.cfg["photo_construction"]
        if self.fused and pc.get("radius_thresh", 0.0) > 0.0 and pc.get("degrees_thresh", 0.0) > 0.0:
            # The reference rebuilds the pair graph on every iterate from the CURRENT poses and median depths
            # (create_photo_system, photo.py:259-262).  With purely temporal edges (the shipped thresholds, 0) the graph only
            # depends on the topology; pose-dependent radius edges would go stale inside the fused / captured iteration.
            raise RuntimeError("como_amd: radius / degree pair edges need the pair graph rebuilt every iteration: use "
                               "WindowBA(fused=False) (it re-evaluates the pairs in linearize()) for radius_thresh, degrees_thresh > 0")
        self.table = None                                    # (its rows hold system indices: they moved with lm_start)
        self._ba_prepared = {}                               # (the marshalled argument block of the BA chain: photo.linearize)
        self._build_pair_table()
        # H | g | err(8) in ONE float64 buffer.  The fused chain does not accumulate into it: every contribution (pair blocks,
        # priors) goes through exact integer atomics into the fixed-point buffer `sysfix` (order-independent: the normal
        # equations are bit-identical from run to run, eager or graph replay, whatever order workgroups and streams finish
        # in) and como_sys_finalize converts once per iteration.  err block: [0] photometric, [1..6] the prior factors.
        # Both buffers are sized for the capacity's system once; a smaller system uses their head (plane stride = the capacity's).
        D = self.dim
        if first:
            Dc = 8 * self.Fcap + 3 * L
            self._sys_cap = torch.zeros((Dc * Dc + Dc + 8,), device=dev, dtype=self.dt)
            self.sysfix, self.fix_plane = None, 0
            if self.fused and dev.type == "cuda":
                self.fix_plane = int(_lib.lib().como_sys_fix_plane_elems(Dc))
                self.sysfix = torch.zeros((2 * self.fix_plane,), device=dev, dtype=torch.int64)
            self.sigma = torch.zeros(2, device=dev, dtype=self.pix_dtype)
        self.sys = self._sys_cap[:D * D + D + 8]
        self.H = self.sys[:D * D].view(D, D)
        self.g = self.sys[D * D:D * D + D]
        self.err8 = self.sys[D * D + D:]
        self.err = self.err8[0]
        self.prior_err = self.err8[1:7] if (self.fused and dev.type == "cuda") else torch.zeros(8, device=dev, dtype=self.dt)
        self.pix_range = (0, 0) if self.idle else None     # (the shard is already cut out of pixidx / vals_n)
        if first:
            self._prepare_fused()
        else:
            self._patch_args()

    def _patch_args(self):
        """The fields of the kernels' argument block that move with the frame count (everything else points into buffers that stay)."""
        a, ptr = self.win_args, _lib.ptr
        a.F, a.D = self.F, self.dim
        a.H, a.g = ptr(self.H), ptr(self.g)
        a.poses, a.aff = ptr(self.poses_all), ptr(self.aff_all)
        a.landmark_inds, a.fix_inds = ptr(self.landmark_inds), ptr(self.fix_inds_flat)

    def _build_pair_table(self):
        """Pair graph from the current poses / median depths (graph_pair_construction.py:155-182) -> device-resident PairTable."""
        B, dev = self.B, self.dev
        ref, tgt, ow_kf, ow_t = setup_photometric_pairs(self.kf_poses, self.recent_poses, self.kf_timestamps,
                                                        self.recent_timestamps, self.median_depths,
                                                        self.cfg["photo_construction"])
        pairs = ([ref, tgt], [ow_kf, ow_t])
        if getattr(self, "table", None) is not None and pairs == (self.kf_pairs, self.one_way_pairs):
            return
        self.kf_pairs, self.one_way_pairs = pairs
        stack = 3 * self.channels * self.Himg * self.Wimg          # one frame's [I | dI/dx | dI/dy] stack
        ex = getattr(self, "_host_extra", None)
        sp, self._spec = getattr(self, "_spec", None), None
        if (sp is not None and ex is not None and sp["pairs"] == pairs and sp["nrec"] == self.F - B and sp["off"] == self._rec_img_off and
                sp["host_np"] is getattr(self, "_host_np", None)):
            # the table `speculate` built for exactly this topology while the tracker ran: wait for its upload, adopt it
            main = torch.cuda.current_stream(dev)
            main.wait_event(sp["event"])
            self.table = sp["table"]
            for t in (self.table.ref_slot, self.table.tgt_img):
                t.record_stream(main)                        # (both are views of the table's one device block)
            self.landmark_inds = self.table.extra["landmark_inds"]
            self.fix_inds_flat = self.table.extra["fix_inds"]
            self.spec_hits = getattr(self, "spec_hits", 0) + 1
            SPEC_STATS["adopted"] += 1
            return
        self.table = photo.PairTable(ref + ow_kf, tgt + ow_t, [False] * len(ref) + [True] * len(ow_kf), B, self.kf_inds,
                                     self.recent_inds, None if ex is not None else self.landmark_inds, stack,
                                     B * stack if self._rec_img_off is None else self._rec_img_off, dev, channels=self.channels,
                                     landmark_inds_host=ex["landmark_inds"] if ex is not None else None, extra_i64=ex)
        if ex is not None:
            self.landmark_inds = self.table.extra["landmark_inds"]
            self.fix_inds_flat = self.table.extra["fix_inds"]

    def _prepare_fused(self):
        B, m, L, F, dev, p = self.B, self.m, self.L, self.F, self.dev, self.pix_dtype
        # the scaffold's outputs: ONE zeroed allocation per element type, carved into views (the window is rebuilt on every
        # keyframe and one-way frame of the sequential loop: twenty small zero-fills cost the host more than the kernels they feed)
        def carve(specs, dtype):
            sizes = [(k, shp, ((int(torch.Size(shp).numel()) + 3) // 4) * 4) for k, shp in specs]      # 16-byte (f32) / 32-byte steps
            buf = torch.zeros(sum(n for _, _, n in sizes), device=dev, dtype=dtype)
            out, o = {}, 0
            for k, shp, n in sizes:
                out[k] = buf[o:o + int(torch.Size(shp).numel())].view(shp)
                out["_off_" + k] = o
                o += n
            out["_buf"] = buf
            return out
        self.w = carve((("pm", (B, m, 2)), ("logzm", (B, m)), ("invz", (B, m)), ("dzdP", (B, 3)), ("dlogz_dT", (B, m, 6)),
                        ("dlogz_dP", (B, m, 3)), ("dp_dP", (B, m, 6)), ("dp_dT", (B, m, 12)), ("init_Pm", (L, 3))), self.dt)
        self._w_head = (self.w["_buf"], self.w["_off_logzm"])           # [pm | logzm | ...]: what snapshot_state copies
        self.w.update(carve((("px_logzm", (B, m)), ("px_invz", (B, m)), ("px_dzdP", (B, 3)), ("px_dlogz_dT", (B, m, 6)),
                             ("px_poses", (self.Fcap, 4, 4)), ("px_aff", (self.Fcap, 2)), ("med3", (B, 3)), ("med3_full", (B, 3))), p))
        self.w = {k: v for k, v in self.w.items() if not k.startswith("_")}
        self.w["reinit_flag"] = torch.zeros(L, device=dev, dtype=torch.int32)
        # The reference keeps TWO medians per keyframe: the one of the sub-selected reference pixels (setup_test_points,
        # sparse_map.py:220 -- only the pair graph reads it) and `self.median_depths` = the median of the FULL depth image
        # exp(K~ logz_m) (store_vars, Mapping.py:749-758), which the priors and the landmark re-initialisation use.  With
        # every pixel a reference pixel (window 1) the two coincide and the dense-reference median serves both.
        self.full_median = self.n_total != self.Himg * self.Wimg
        self.first_mask_u8 = self.obs_ref_mask.to(torch.uint8).contiguous()
        self.fix_lm = self.fix_idx.to(torch.int32).contiguous()
        self.aff_anchor2 = self.aff_anchor.reshape(2).contiguous()
        a = _lib.WinArgs()
        a.B, a.F, a.m, a.L, a.nfix = B, F, m, L, (int(self.fix_lm.numel()) if self.window_full else 0)
        a.pix_is_f64 = 1 if p == torch.float64 else 0
        a.median_new_is_f32 = 1 if p == torch.float32 else 0
        a.median_new_stride = 3
        a.D = self.dim
        ptr = _lib.ptr
        a.poses, a.aff, a.K, a.median = ptr(self.poses_all), ptr(self.aff_all), ptr(self.intrinsics), ptr(self.median_depths)
        a.pm_first, a.Kmm_inv = ptr(self.pm_first_obs), ptr(self.K_mm_inv)
        a.pose_anchor, a.aff_anchor, a.P_anchor, a.P_m = ptr(self.pose_anchor), ptr(self.aff_anchor2), ptr(self.P_m_anchors), ptr(self.P_m)
        a.lm_ids, a.first_frame, a.first_slot, a.fix_lm = ptr(self.lm_ids), ptr(self.first_frame), ptr(self.first_slot), ptr(self.fix_lm)
        a.first_mask, a.pose_inds, a.landmark_inds, a.fix_inds = (ptr(self.first_mask_u8), ptr(self.kf_inds), ptr(self.landmark_inds),
                                                                  ptr(self.fix_inds_flat))
        a.median_new = ptr(self.w["med3_full"] if self.full_median else self.w["med3"])
        for k in ("pm", "logzm", "invz", "dzdP", "dlogz_dT", "dlogz_dP", "dp_dP", "dp_dT", "init_Pm", "reinit_flag", "px_logzm",
                  "px_invz", "px_dzdP", "px_dlogz_dT", "px_poses", "px_aff"):
            setattr(a, k, ptr(self.w[k]))
        sg = self.cfg["sigmas"]
        a.s_gp, a.s_ld, a.s_px, a.s_pose, a.s_aff, a.s_lm = 1.0, 1.0, 1e-2, sg["pose_prior"], sg["scale_prior"], sg["scale_prior"]
        self._scratch_err = torch.zeros(8, device=dev, dtype=self.dt)
        a.H, a.g, a.err = ptr(self.H), ptr(self.g), ptr(self._scratch_err)
        a.sysfix, a.fix_plane = ptr(self.sysfix), self.fix_plane
        # the two radix-select workspaces of an iteration are cleared by the scaffold kernel (no fill launches)
        hb = _lib.lib().como_select_workspace_bytes()
        hall = torch.zeros((2 * B + 1) * hb // 4, dtype=torch.int32, device=self.dev)      # one allocation: dense-ref | full-image | BA
        self.w["hist_dr2"] = hall[:2 * B * hb // 4]
        self.w["hist_dr"] = self.w["hist_dr2"][:B * hb // 4]
        self.w["hist_full"] = self.w["hist_dr2"][B * hb // 4:]
        self.w["hist_ba"] = hall[2 * B * hb // 4:]
        a.zero_a, a.zero_a_bytes = ptr(self.w["hist_dr2"]), 2 * B * hb
        a.zero_b, a.zero_b_bytes = ptr(self.w["hist_ba"]), hb
        if self.sysfix is not None and (self.sysfix.numel() * 8) % 16 == 0:
            a.zero_c, a.zero_c_bytes = ptr(self.sysfix), self.sysfix.numel() * 8      # the scaffold launch clears it (no fill launch)
        a.median_out = ptr(self.median_depths)
        if not self.window_full:
            # mean_log_depth_cost (gp_priors.py:84-150): J = column means of keyframe 0's K~ (float64 sum of the pix-dtype rows, as the
            # mirror path forms it), anchor = the initial scale, sigma = cfg mean_depth_prior; constant for this topology
            inh0 = getattr(self, "_inherit", None)
            if (inh0 is not None and getattr(inh0, "mld_J", None) is not None and inh0.Kt.data_ptr() == self.Kt.data_ptr() and
                    inh0.Kt.shape == self.Kt.shape):
                self.mld_J = inh0.mld_J                     # same predictors (a rebuild on a one-way frame): same column means
            else:
                self.mld_J = (self.Kt[0].to(self.dt).sum(0) / self.Kt.shape[1]).contiguous()
            self.mld_anchor = torch.as_tensor(self.init_scale_anchor, dtype=self.dt, device=dev).reshape(-1)[:1].contiguous()
            a.mld_J, a.mld_anchor, a.s_mld = ptr(self.mld_J), ptr(self.mld_anchor), float(sg["mean_depth_prior"])
        self.win_args = a
        self.w["dr_ws"] = {}                               # dense-reference planes: owned here (captured graphs record them)
        self.w["ba_ws"] = {}                               # residual / validity / pair / partial-Gram scratch of the BA chain
        self.w["chol_ws"] = {}                             # Cholesky workspace + delta
        self.with_priors = True                            # tests: False leaves H = the photometric system alone
        self.overlap_priors = True
        inh = getattr(self, "_inherit", None)
        if (inh is not None and _REUSE_WORKSPACES and getattr(inh, "graph", None) is None and isinstance(getattr(inh, "w", None), dict)
                and "dr_ws" in inh.w and inh.m == self.m):
            # The retired window's scratch -- planes keyed by (B, n), grow-only buffers of the block chain, Cholesky workspaces keyed
            # by the system size -- is taken over instead of being allocated and zero-filled again in the first iteration (the
            # sequential loop rebuilds the window on every keyframe and one-way frame).  With the SAME predictors (a one-way frame)
            # the state of the band median carries over too -- it only depends on K~ and the log-depths: the previous medians move
            # into this window's buffer; with other predictors its planes are kept and the state is rebuilt in them.
            same_kt = inh.Kt.data_ptr() == self.Kt.data_ptr() and inh.Kt.shape == self.Kt.shape
            self.w["ba_ws"], self.w["chol_ws"] = inh.w["ba_ws"], inh.w["chol_ws"]
            self.w["dr_ws"] = {k: e for k, e in inh.w["dr_ws"].items() if isinstance(k, tuple) and B in k[2:4]}   # (planes of this B only)
            for e in self.w["dr_ws"].values():
                st = e.get("band") if isinstance(e, dict) else None
                if st is not None:
                    if same_kt and st["med"] == inh.w["med3_full"].data_ptr() and inh.w["med3_full"].shape == self.w["med3_full"].shape:
                        self.w["med3_full"].copy_(inh.w["med3_full"])
                    else:
                        st["ident"] = None                   # (full_image_median rebuilds the state in the same planes)
                    st["med"] = self.w["med3_full"].data_ptr()
        # (a side stream of the lowest priority was tried for the branch below: nothing in the eager loop, and +0.7 ms per replay
        # of a captured iteration -- 777 -> 506 it/s on the bench window -- so it is a plain pool stream, kept across rebuilds)
        if inh is not None and getattr(inh, "_side_stream", None) is not None:
            self._side_stream = inh._side_stream
        else:
            self._side_stream = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None

    def snapshot_state(self, dtype=None):
        """Copies of the iteration's state for the caller (Mapping.iterate): {"poses" (F,4,4), "aff" (F,2), "P_m" (L,3), "median" (B,),
        and on the fused chain "pm" (B,m,2), "logzm" (B,m)} -- views of TWO copied buffers (the state buffer and the head of the
        scaffold's output buffer) instead of seven separate clones."""
        dtype = dtype or self.dt
        both = self.fused and getattr(self, "_w_head", None) is not None
        if both and dtype == self.state_flat.dtype == self._w_head[0].dtype and self.state_flat.is_cuda:
            # (both copies in ONE launch: every dependent launch of the sequential loop's frame chain costs >= 4.5 us)
            buf, o_lz = self._w_head
            src_head = buf[:o_lz + self.B * self.m]
            snap, head = torch.empty_like(self.state_flat), torch.empty_like(src_head)
            torch._foreach_copy_([snap, head], [self.state_flat, src_head])
        else:
            snap, head = self.state_flat.to(dtype, copy=True), None
        out = {k: snap[o:o + int(torch.Size(shp).numel())].view(shp) for k, o, shp in self._state_layout}
        if both:
            buf, o_lz = self._w_head
            B, m = self.B, self.m
            if head is None:
                head = buf[:o_lz + B * m].to(dtype, copy=True)
            out["pm"] = head[:2 * B * m].view(B, m, 2)
            out["logzm"] = head[o_lz:o_lz + B * m].view(B, m)
        return out

    # ---- fused HIP chain -----------------------------------------------------------------------------------------
    def linearize_fused(self):
        L, a, w, dev = _lib.lib(), self.win_args, self.w, self.dev
        s = _lib.stream_ptr(dev)
        if self._ahead is not None:
            # a median streamed ahead computed its log-depths from the landmarks as they stand: the scaffold launch below moves the
            # re-initialised ones (win_apply_reinit) -- not before that read is done (back-to-back iterations; long done in the loop)
            torch.cuda.current_stream(dev).wait_event(self._ahead)
        _lib.check(L.como_win_scaffold(ctypes.byref(a), s), "como_win_scaffold")
        if not a.zero_c:
            self.sysfix.zero_()
        # Pass 1 of the photometric system (warp / sample / residual / validity + the first digit of the robust scale) rides in the
        # dense-reference launch (csrc/densify.hip DRFuse): the pair constants first (phase 1024), then the fused launch, then the
        # rest of the chain (phase 0xFE).  Gray images, the whole pixel range, one GPU, no per-kernel events.
        fuse = None
        if (_FUSE_PASS1 and self.shard is None and self.channels == 1 and self.events is None and self.pix_range is None and
                self.table.b > 0 and self.n > 0):
            planes = dense_reference_planes(self.Kt, self.n, True, w["dr_ws"])
            prep = self._ba_prepared
            photo.photo_system_factored(self.table, poses_all=w["px_poses"], aff_all=w["px_aff"], Pwn=planes["Pwn"], vals=self.vals_n,
                                        dPwn_dTwc=planes["dT"], uvec=None, Kt=self.Kt, pixidx=self.pixidx, invz=w["px_invz"],
                                        dzdP=w["px_dzdP"], img_base=self.img, K=self.K_pix, H_img=self.Himg, W_img=self.Wimg, H=None,
                                        g=None, err_out=None, sigma_out=self.sigma, zeroed_hists=w["hist_ba"], ws=w["ba_ws"],
                                        sysfix=self.sysfix, fix_plane=self.fix_plane, D=self.dim, prepared=prep, phase=1024)
            fuse = prep.get("fuse")
            if fuse is None and "aux" in prep:
                aux, b = prep["aux"], self.table.b
                fuse = _lib.DRFuse()
                fuse.ref_pairs, fuse.np_max = _lib.ptr(self.table.ref_pairs), int(self.table.np_max)
                fuse.pair_T = aux["pair"].data_ptr()
                fuse.pair_aff = aux["pair"].data_ptr() + 12 * b * aux["pair"].element_size()
                fuse.vals, fuse.img_base, fuse.tgt_img = _lib.ptr(self.vals_n), _lib.ptr(self.img), _lib.ptr(self.table.tgt_img)
                fuse.r_out, fuse.valid_out, fuse.rhists = aux["r"].data_ptr(), aux["valid"].data_ptr(), aux["hists"].data_ptr()
                fuse.H, fuse.W, fuse.anorm_f32 = self.Himg, self.Wimg, 0
                prep["fuse"] = fuse
        # (single GPU without per-kernel events: the marshalled calls of the dense reference / the full-image median are cached with
        # the BA chain's argument block, per topology; launches on the side stream take its handle instead of a stream context)
        cc = self._ba_prepared if (self.shard is None and self.events is None) else None
        dr = lambda part, stream=None: dense_reference_factored(w["px_logzm"], w["px_poses"][:self.B], self.Kt, self.pixidx, self.K_pix,
                                                                w["px_dlogz_dT"], self.Wimg, want_logz=False, med_out=w["med3"],
                                                                hists=w["hist_dr"], ws=w["dr_ws"], part=part, compact=True, fuse=fuse,
                                                                call_cache=cc, stream=stream)
        fork = self.shard is None and self.overlap_priors
        late_side = False
        fm = lambda part, stream=None: full_image_median(w["px_logzm"], self.Kt, w["med3_full"], w["dr_ws"], hists=w["hist_full"],
                                                         part=part, band=self.band_median, call_cache=cc, stream=stream)
        if self.shard is not None:
            return self._linearize_sharded(dr)
        if fork and self.full_median:
            # the full-image median needs only the scaffold's log-depths: its whole branch (depth image, select passes,
            # priors) runs beside the dense reference points and the photometric system.  The dense reference is SUBMITTED first:
            # the band kernel streams up to the whole K~ window (0.7 GB) for a result only the priors read, and launched first it
            # held the critical chain back
            main = torch.cuda.current_stream(dev)
            side = self._side_stream
            ev = torch.cuda.Event()
            ev.record(main)
            Pwn, dT, uvec, med, _ = dr("points")
            late_side = True                                 # (the branch is submitted AFTER the photometric chain, below)
        else:
            Pwn, dT, uvec, med, _ = dr("points" if fork else "all")
        if fork and not self.full_median:
            # Nothing between the reference points and the solve needs the median depths except the priors, and the priors
            # only ADD into H / g (atomics): the median's select passes and the prior kernel run on a second stream (a
            # parallel branch of the captured graph) beside the residual / block / assembly kernels -- ~50 us of small
            # launches off the critical path.
            main = torch.cuda.current_stream(dev)
            side = self._side_stream
            side.wait_stream(main)
            ss = side.cuda_stream
            dr("median", ss)
            if self.with_priors:
                _lib.check(L.como_win_priors(ctypes.byref(a), ss), "como_win_priors")
        # (eager launches execute in submission order whenever a kernel fills the chip: submitted between the dense reference and
        # the photometric chain, the band kernel + its select passes + the priors -- 165 us at 9 x 640x480 -- sat ON the critical
        # path of every eager iteration of the sequential loop although they are a side branch; submitted after the chain they run
        # beside / behind it and only the final pack waits for them.  The captured graph has the same two branches either way.)
        photo.photo_system_factored(self.table, poses_all=w["px_poses"], aff_all=w["px_aff"], Pwn=Pwn, vals=self.vals_n,
                                    dPwn_dTwc=dT, uvec=uvec, Kt=self.Kt, pixidx=self.pixidx, invz=w["px_invz"], dzdP=w["px_dzdP"],
                                    img_base=self.img, K=self.K_pix, H_img=self.Himg, W_img=self.Wimg, H=None, g=None,
                                    err_out=None, sigma_out=self.sigma, events=self.events, zeroed_hists=w["hist_ba"],
                                    ws=w["ba_ws"], sysfix=self.sysfix, fix_plane=self.fix_plane, D=self.dim,
                                    prepared=self._ba_prepared if self.events is None else None,
                                    phase=0xFE if fuse is not None else 0xFF)
        ahead, self._ahead = self._ahead, None
        if late_side and ahead is not None:
            # iteration t's epilogue already streamed this iteration's full-image median into med3_full on the side stream
            # (`_issue_median_ahead`): only the priors are left for the branch -- they follow it in stream order
            side.wait_event(ev)
            ss = side.cuda_stream
            AHEAD_STATS["adopted"] += 1
            if _MEDIAN_AHEAD == 2:
                side.synchronize()
                got = w["med3_full"].clone()
                fm("all", ss)
                side.synchronize()
                AHEAD_STATS["checked"] += 1
                if not torch.equal(got.contiguous().view(torch.uint8), w["med3_full"].contiguous().view(torch.uint8)):
                    AHEAD_STATS["mismatch"] += 1
            if self.with_priors:
                _lib.check(L.como_win_priors(ctypes.byref(a), ss), "como_win_priors")
        elif late_side:
            side.wait_event(ev)
            if self.band_median is False or (self.band_median is None and not BAND_MEDIAN):
                ss = side.cuda_stream                        # (the plain streaming pass: kernels only, no torch op in between)
                fm("all", ss)
                if self.with_priors:
                    _lib.check(L.como_win_priors(ctypes.byref(a), ss), "como_win_priors")
            else:
                with torch.cuda.stream(side):                # (the band form zeroes / copies state with torch ops: a stream context)
                    fm("all")
                    if self.with_priors:
                        _lib.check(L.como_win_priors(ctypes.byref(a), _lib.stream_ptr(dev)), "como_win_priors")
        if fork:
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            if self.full_median:
                fm("all")
            if self.with_priors:
                _lib.check(L.como_win_priors(ctypes.byref(a), s), "como_win_priors")   # also stores the new median depths
        self._finalize(s)
        return self.H, self.g

    def _finalize(self, s):
        """Fixed point -> float64 H / g / err AND the Cholesky solver's packed working copy, one launch."""
        cw = lin_sys.chol_workspace(self.dim, self.dev, self.w["chol_ws"])
        _lib.check(_lib.lib().como_sys_finalize_pack(self.sysfix.data_ptr(), self.fix_plane, self.dim, self.H.data_ptr(), self.g.data_ptr(),
                                                     self.err8.data_ptr(), cw[0].data_ptr(), cw[1].data_ptr(), s), "como_sys_finalize_pack")
        self._packed = True
        if getattr(self, "_marks", None) is not None:
            self._marks["packed"].record()

    def _linearize_sharded(self, dr):
        """The multi-GPU iteration: the same kernels on this rank's pixel range, plus two kinds of collectives (all
        `Shard.all_reduce_sum`, RCCL on the GPU): per radix-select digit pass ONE all-reduce carrying the histogram of the global
        robust scale together with the per-keyframe histograms of the median depth (3 for float keys, 6 for double), and one
        all-reduce of the fixed-point per-pair sums.  Everything after them is replicated and bit-identical."""
        L, a, w, dev = _lib.lib(), self.win_args, self.w, self.dev
        s = _lib.stream_ptr(dev)
        red = self.shard.all_reduce_sum
        B = self.B
        sfx = _lib.suffix(self.pix_dtype)
        npass = 3 if self.pix_dtype == torch.float32 else 6
        # --- the depth whose per-keyframe median the priors need: the full depth image (sub-selected windows) or the dense
        #     reference itself (every pixel a reference pixel); its select passes ride on the residual median's all-reduces
        if self.full_median:
            rb, re = self.row_range
            zmed = full_image_median(w["px_logzm"], self.Kt[:, rb:re], w["med3_full"], w["dr_ws"], hists=w["hist_full"], reduce="defer",
                                     reduce_max=self.shard.all_reduce_max)
            hmed, med_out, z_idle = w["hist_full"], w["med3_full"], False
        if not self.idle:
            Pwn, dT, uvec, _, _ = dr("points")
        else:
            z1 = torch.zeros((B, 1), device=dev, dtype=self.pix_dtype)
            Pwn, dT, uvec = z1.new_zeros((B, 3, 1)), z1.new_zeros((B, 6, 1)), None
        if not self.full_median:
            key = (str(dev), self.Kt.dtype, B, self.n, "compact")
            zmed = w["dr_ws"][key]["z"] if not self.idle else z1
            hmed, med_out, z_idle = w["hist_dr"], w["med3"], self.idle
        hv = hmed.view(B, 6, 2048)
        state = {"p": 0}
        # staging buffer of the ONE collective per digit pass (global residual histogram | B median-depth histograms): owned by
        # the window (a captured graph records its address; nothing is allocated inside the iteration)
        stage = w.get("hist_stage")
        if stage is None:
            stage = w["hist_stage"] = torch.empty((1 + B) * 2048, dtype=torch.int32, device=dev)
        fork = self.overlap_priors and self._side_stream is not None
        side = self._side_stream

        def medians_resolved():
            if fork:
                # every digit of the median depths is resolved: their finish + the prior factors (which only ADD into the
                # fixed-point system) run on the side stream beside the block kernel, the exchange of the pair sums and
                # their expansion -- the same two-branch shape as the single-GPU chain
                main = torch.cuda.current_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ss = _lib.stream_ptr(dev)
                    _lib.check(getattr(L, "como_select_finish_" + sfx)(hmed.data_ptr(), B, med_out.data_ptr(), ss), "como_select_finish")
                    if self.with_priors:
                        _lib.check(L.como_win_priors(ctypes.byref(a), ss), "como_win_priors")

        def reduce_both(h_ba):
            p = state["p"]
            stage[:2048].copy_(h_ba)
            stage[2048:].view(B, 2048).copy_(hv[:, p])
            red(stage)                                     # ONE collective per digit pass for both exact medians
            h_ba.copy_(stage[:2048])
            hv[:, p].copy_(stage[2048:].view(B, 2048))
            state["p"] = p + 1
            if exchange and p == 2:
                return                                     # digits 3..5: reduce_both.candidates
            if p + 1 < npass and not z_idle:
                _lib.check(getattr(L, "como_select_hist_" + sfx)(zmed.data_ptr(), None, zmed.shape[1], B, hmed.data_ptr(), p + 1, s),
                           "como_select_hist")
            if p + 1 == npass:
                medians_resolved()

        # float64 keys: three all-reduced digits (33 bits), then ONE all-gather of the keys that still match -- a handful per rank
        # and median -- from which every rank finishes digits 3..5 itself (csrc/select.hip, como_select_cand_*): 3 + 1 collectives
        # for both exact medians instead of 6 (+ the per-pair sums = 5 per iteration; `l1max` only when the band state is built)
        exchange = npass == 6 and os.environ.get("COMO_SEL_EXCHANGE", "1") != "0"
        if exchange:
            cw = L.como_select_cand_words()
            cstage = w.get("cand_stage")
            if cstage is None:
                cstage = w["cand_stage"] = (torch.zeros((1 + B, cw), dtype=torch.int32, device=dev),
                                            torch.zeros((self.shard.world, 1 + B, cw), dtype=torch.int32, device=dev))

            def candidates(h_ba):
                # (pass 3 of the residual select has just collected its candidates into h_ba's scratch)
                if not z_idle:
                    _lib.check(L.como_select_hist_f64(zmed.data_ptr(), None, zmed.shape[1], B, hmed.data_ptr(), 3 | 0x300, s), "como_select_hist")
                loc, allr = cstage
                _lib.check(L.como_select_cand_pack(h_ba.data_ptr(), 1, loc.data_ptr(), s), "como_select_cand_pack")
                _lib.check(L.como_select_cand_pack(hmed.data_ptr(), B, loc[1:].data_ptr(), s), "como_select_cand_pack")
                self.shard.all_gather(allr, loc)
                wd = self.shard.world
                _lib.check(L.como_select_cand_merge(h_ba.data_ptr(), 1, allr.data_ptr(), wd, 1 + B, 0, s), "como_select_cand_merge")
                _lib.check(L.como_select_cand_merge(hmed.data_ptr(), B, allr.data_ptr(), wd, 1 + B, 1, s), "como_select_cand_merge")
                state["p"] = npass
                medians_resolved()

            reduce_both.candidates = candidates

        photo.photo_system_factored(self.table, poses_all=w["px_poses"], aff_all=w["px_aff"], Pwn=Pwn, vals=self.vals_n,
                                    dPwn_dTwc=dT, uvec=uvec, Kt=self.Kt, pixidx=self.pixidx, invz=w["px_invz"], dzdP=w["px_dzdP"],
                                    img_base=self.img, K=self.K_pix, H_img=self.Himg, W_img=self.Wimg, H=None, g=None,
                                    err_out=None, sigma_out=self.sigma, pix_range=self.pix_range, reduce_hists=reduce_both,
                                    events=self.events, zeroed_hists=w["hist_ba"], ws=w["ba_ws"], sysfix=self.sysfix,
                                    fix_plane=self.fix_plane, D=self.dim, reduce_blocks=red)
        assert state["p"] == npass
        if fork:
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            _lib.check(getattr(L, "como_select_finish_" + sfx)(hmed.data_ptr(), B, med_out.data_ptr(), s), "como_select_finish")
            if self.with_priors:
                _lib.check(L.como_win_priors(ctypes.byref(a), s), "como_win_priors")
        self._finalize(s)
        return self.H, self.g

    def timed_iterate(self):
        """One eager iteration with three events on the current stream: "start", "packed" (the normal equations are complete and
        packed: everything after it -- solve, update -- is replicated on every rank) and "end" (como_amd/dist.py dist_record)."""
        ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "packed", "end")}
        self._marks = ev
        ev["start"].record()
        try:
            self.iterate()
        finally:
            self._marks = None
        ev["end"].record()
        return ev

    def iterate_fused(self):
        self._packed = False
        H, g = self.linearize_fused()
        delta = (lin_sys.solve_packed(self.dim, self.dev, self.w["chol_ws"]) if self._packed
                 else lin_sys.solve_system(H, g, ws=self.w["chol_ws"]))
        # the update is guarded by the solver's status word ON THE DEVICE: a solve that did not complete (non-positive pivot;
        # -1 = the persistent solver timed out and left delta unwritten) leaves the state as it is -- `check_solver` / the
        # sequential loop's read-back (Mapping._check_solver) decide what happens next
        self.info = lin_sys.solve_system.last_info
        rc = _lib.lib().como_win_update_checked(delta.data_ptr(), self.poses_all.data_ptr(), self.aff_all.data_ptr(),
                                                self.frame_inds.data_ptr(), self.F, self.P_m.data_ptr(), self.L, self.lm_start,
                                                self.info.data_ptr(), _lib.stream_ptr(self.dev))
        _lib.check(rc, "como_win_update")
        self.delta = delta
        if self.median_ahead and self.median_ahead != "gap" and _MEDIAN_AHEAD:
            self._issue_median_ahead()                       # ("gap": the caller issues it -- Mapping.median_ahead_now)
        return delta

    def _issue_median_ahead(self):
        """The sequential loop runs ONE iteration per frame; between two of them the tracker works (latency-bound level kernels, a
        host read-back) while nothing touches the keyframes' state -- so the next iteration's full-image median (0.7 GB of K~ at 9 x
        640x480: byte-bound, ~100 us alone, and the block kernel ran at half speed beside it) is streamed NOW on the side stream: the
        scaffold kernel itself computes the log-depths the next scaffold will compute (`como_win_logz_ahead`: same code, outputs into
        scratch), the depth-only pass + select passes write med3_full, and the next `linearize_fused` only submits the priors.
        Valid while this object keeps its keyframe state (`_load_frames` of another state drops it); plain streaming form, one GPU,
        eager iterations only."""
        self._ahead = None
        if not (self.fused and self.full_median and self.overlap_priors and self.shard is None and self.events is None and
                self.graph is None and self._side_stream is not None and self.with_priors and
                (self.band_median is False or (self.band_median is None and not BAND_MEDIAN)) and
                not torch.cuda.is_current_stream_capturing()):
            return
        ent = self._ba_prepared.get(("fm", "all")) if isinstance(self._ba_prepared, dict) else None
        st = getattr(self, "_ahead_state", None)
        if st is None:
            if ent is None:
                return                                       # (the first iteration of this object has not marshalled the call yet)
            L, w, a = _lib.lib(), self.w, self.win_args
            nb = int(L.como_win_logz_ahead_scratch_bytes(self.B, self.m, self.L, self.Fcap))
            hb = self.B * int(L.como_select_workspace_bytes())
            fn, args, _ = ent
            keep = self._ba_prepared[("fm_keep", "all")]
            if args[3] != w["px_logzm"].data_ptr() or args[16] != w["hist_full"].data_ptr() or args[17] != w["med3_full"].data_ptr():
                return                                       # (not the call this was written for)
            st = {"scratch": torch.empty((nb + 15) // 8, device=self.dev, dtype=torch.float64),
                  "lz": torch.empty_like(w["px_logzm"]), "hist": torch.zeros(hb // 4, device=self.dev, dtype=torch.int32),
                  "fn": fn, "nb": nb, "hb": hb, "keep": keep}
            st["args"] = args[:3] + (st["lz"].data_ptr(),) + args[4:16] + (st["hist"].data_ptr(),) + args[17:]
            self._ahead_state = st
        main, side = torch.cuda.current_stream(self.dev), self._side_stream
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        ss = side.cuda_stream
        _lib.check(_lib.lib().como_win_logz_ahead(ctypes.byref(self.win_args), st["scratch"].data_ptr(), st["nb"], st["lz"].data_ptr(),
                                                  st["hist"].data_ptr(), st["hb"], ss), "como_win_logz_ahead")
        read = torch.cuda.Event()
        read.record(side)                                    # (the state has been read: the next scaffold's re-initialisation may write it)
        _lib.check(st["fn"](*st["args"], ss), "como_dense_ref (depth only, ahead)")
        AHEAD_STATS["issued"] += 1
        self._ahead = read

    def check_solver(self):
        """Read the status of the last solve (one host synchronisation) and act on it -- the reference swallows it
        (linear_system.py:109 cholesky_ex(check_errors=False)); SURVEY.md section 5 asks for it to be acted on:
          0   -> True;
          -1  -> the persistent one-launch solver's waits timed out (its workgroups were not co-resident: another process or
                 kernel on the device): the guarded update did not run; the persistent form is switched off for the process, a
                 captured iteration is captured again on the multi-launch solver, the iteration is redone; returns False;
          > 0 -> H is not positive definite at that pivot (or holds a non-finite entry): RuntimeError, state untouched."""
        info = getattr(self, "info", None)
        if info is None:
            return True
        v = int(info.item())
        if v == 0:
            return True
        if v == -1:
            _lib.lib().como_chol_set_persistent(0)
            WindowBA.solver_fallbacks = getattr(WindowBA, "solver_fallbacks", 0) + 1
            if self.graph is not None:
                self.capture()
            self.step()
            if int(self.info.item()) != 0:
                raise RuntimeError(f"como_amd: window solve failed again on the multi-launch solver (info {int(self.info.item())})")
            return False
        raise RuntimeError(f"como_amd: the window's normal equations are not positive definite (Cholesky info {v}); "
                           "the state was left unchanged")

    # ---- the same iteration through the reference-signature mirrors (torch ops for the O(B m) parts) ---------------
    def scaffold(self):
        """Mapping.prep_geometry_scaffold (:603-659) without host synchronisation."""
        K = self.intrinsics
        depth_init = self.median_depths[:, None, None].expand(-1, self.m, 1)
        init_Pc, _ = backprojection(K[0], self.pm_first_obs, depth_init)
        init_Pw = smap.rigid_apply_exact(self.kf_poses, init_Pc)                  # (B,m,3)
        # re-initialisation point of a landmark = back-projection from its FIRST observer at the median depth (:625-634)
        lm_ids = self.lm_ids.long()
        fom = self.first_obs_mask
        init_Pm = torch.zeros_like(self.P_m)
        init_Pm.index_add_(0, lm_ids.reshape(-1), (init_Pw * fom[..., None]).reshape(-1, 3))
        out = smap.project_landmarks(self.kf_poses, self.P_m[lm_ids], K[0], init_Pm[lm_ids], self.median_depths)
        z_mask = out[2]
        # landmarks re-initialised in their first-observation frame are moved for good (Mapping.py:645-648)
        flag = torch.zeros((self.P_m.shape[0], 1), device=self.dev, dtype=self.dt)
        flag.index_add_(0, lm_ids.reshape(-1), (fom & z_mask).reshape(-1, 1).to(self.dt))
        self.P_m.copy_(torch.where(flag > 0, init_Pm, self.P_m))
        return out

    def linearize(self):
        if self.fused:
            return self.linearize_fused()
        pc = self.cfg["photo_construction"]
        if pc.get("radius_thresh", 0.0) > 0.0 and pc.get("degrees_thresh", 0.0) > 0.0:
            self._build_pair_table()          # pose-dependent edges: rebuilt every iteration, as create_photo_system does
        pm, logzm, z_mask, dlogzm_dzm, dzm_dPwm, dzm_dTwc, dpm_dPwm, dpm_dTwc = self.scaffold()
        dlogzm_dTwc = dlogzm_dzm @ dzm_dTwc
        dlogzm_dPwm = dlogzm_dzm @ dzm_dPwm
        p = self.pix_dtype
        Pwn, dPwn_dTwc, uvec, med, _ = dense_reference_factored(logzm.to(p), self.kf_poses.to(p), self.Kt, self.pixidx, self.K_pix,
                                                                dlogzm_dTwc.to(p), self.Wimg, want_logz=False, compact=True)
        self.median_subset = med
        if self.full_median:                                 # Mapping.store_vars: median of the full depth image
            if not hasattr(self, "_fm_ws"):
                self._fm_ws = ({}, torch.zeros((self.B, 3), device=self.dev, dtype=p))
            med = full_image_median(logzm.to(p), self.Kt, self._fm_ws[1], self._fm_ws[0])
        self.median_depths.copy_(med)
        self.pm, self.logzm = pm, logzm
        H, g = self.H, self.g
        self.sys.zero_()
        poses_all = self.poses_all.to(p).contiguous()
        aff_all = self.aff_all.to(p).contiguous()
        photo.photo_system_factored(self.table, poses_all=poses_all, aff_all=aff_all, Pwn=Pwn, vals=self.vals_n,
                                    dPwn_dTwc=dPwn_dTwc, uvec=uvec, Kt=self.Kt, pixidx=self.pixidx,
                                    invz=dlogzm_dzm[:, :, 0, 0].to(p).contiguous(), dzdP=dzm_dPwm[:, 0, 0, :].to(p).contiguous(),
                                    img_base=self.img, K=self.K_pix, H_img=self.Himg, W_img=self.Wimg, H=H, g=g,
                                    err_out=self.err, sigma_out=self.sigma, pix_range=self.pix_range,
                                    reduce_hists=(self.shard.all_reduce_sum if self.shard is not None else None),
                                    events=self.events, ws=self.w["ba_ws"])
        if self.shard is not None:
            self.shard.all_reduce_sum(self.sys)
        kf_pose_inds = self.kf_inds[:, :6]
        log_med = torch.log(self.median_depths[:, None, None])
        sg = self.cfg["sigmas"]
        e = [self.err.clone()]
        e.append(gp_ml_cost(logzm, log_med, self.L_mm, dlogzm_dPwm, dlogzm_dTwc, self.landmark_inds, kf_pose_inds, H, g, sigma=1e0))
        e.append(log_depth_prior(logzm, log_med, dlogzm_dPwm, dlogzm_dTwc, self.obs_ref_mask, self.landmark_inds, kf_pose_inds,
                                 H, g, mode="first_mean", sigma_first=1e0, sigma_all=1e-0))
        e.append(pixel_prior_cost(pm, self.pm_first_obs, dpm_dPwm, dpm_dTwc, self.obs_ref_mask, self.landmark_inds, kf_pose_inds,
                                  H, g, mode="first", pixel_sigma_first=1e-2, pixel_sigma_all=3.33e-1))
        e.append(linearize_pose_prior(self.kf_poses[0:1], self.pose_anchor[0:1], H, g, [0, 6], sigma=sg["pose_prior"]))
        e.append(linearize_scalar_prior(self.kf_aff_params[0, 0:1, :], self.aff_anchor[0, 0:1, :], H, g, [6, 7], sigma=sg["scale_prior"]))
        e.append(linearize_scalar_prior(self.kf_aff_params[0, 1:2, :], self.aff_anchor[0, 1:2, :], H, g, [7, 8], sigma=sg["scale_prior"]))
        if self.window_full:
            e.append(linearize_multi_scalar_prior(self.P_m[self.fix_idx].flatten(), self.P_m_anchors.flatten(), H, g,
                                                  self.fix_inds_flat, sigma=sg["scale_prior"]))
        else:
            e.append(mean_log_depth_cost(logzm[0:1], self.Kt[0:1].to(self.dt), self.init_scale_anchor, dlogzm_dPwm[0:1],
                                         dlogzm_dTwc[0:1], self.landmark_inds[0:1], kf_pose_inds[0:1], H, g,
                                         sg["mean_depth_prior"]))
        self.total_err = sum(x.reshape(()).to(torch.float64) for x in e)
        return H, g

    def iterate(self):
        """One GN iteration, eager.  State tensors are updated IN PLACE (fixed addresses -> capturable)."""
        if self.fused:
            return self.iterate_fused()
        H, g = self.linearize()
        delta = lin_sys.solve_system(H, g)
        kp, ka, rp, ra, Pn = lin_sys.update_vars(delta, self.kf_poses, self.kf_aff_params, self.kf_inds, self.recent_poses,
                                                 self.recent_aff_params, self.recent_inds, self.P_m, self.lm_start)
        self.kf_poses.copy_(kp)
        self.kf_aff_params.copy_(ka)
        if self.F > self.B:
            self.recent_poses.copy_(rp)
            self.recent_aff_params.copy_(ra)
        self.P_m.copy_(Pn)
        self.delta = delta
        return delta

    # ---- hipGraph: replaying a captured iteration removes the host from the loop -----------------------------------
    def capture(self, warmup=3):
        """Capture one GN iteration into a hipGraph (torch.cuda.CUDAGraph).  Returns True on success; on failure the
        object stays usable in eager mode (the reason is kept in self.capture_error)."""
        # multi-GPU: the collectives are captured with the kernels (RCCL supports stream capture); if the runtime refuses,
        # the except branch below leaves the object in eager mode
        self.graph = None
        if self.shard is not None and self.shard.collectives and not self.shard.capturable():
            # (a gloo group -- the single-GPU test rigs -- stages through the host: nothing to capture, and an aborted
            # capture leaves gloo's own streams in a state that crashed later collectives)
            self.capture_error = "process group backend cannot be captured into a hipGraph"
            return False
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.iterate()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        ev = self.events
        self.events = None
        # (thread_local: the process group's watchdog thread may query events while this thread captures)
        g, out = _lib.capture_graph(self.iterate, self.dev, thread_local=self.shard is not None)
        self.events = ev
        if g is None:
            self.capture_error = out
            return False
        self.graph = g
        return True

    def step(self):
        """One GN iteration: graph replay when captured, eager otherwise."""
        if self.graph is not None:
            self.graph.replay()
            return self.delta
        return self.iterate()
