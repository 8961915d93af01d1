"""Drop-in for the reference's como/odom/backend/photo.py on MI355X.

`batch_photo_cost` / `create_photo_system` keep the reference's argument lists and in-place
accumulation into `H`, `g` (photo.py:83-99, 236-258).  The work happens in the HIP kernel chain of
csrc/ba.hip through the C ABI entry `como_ba_linearize_*`; there is no CPU fallback.

`photo_system_factored` is the internal fast path SURVEY.md section 8(b) allows: the same maths from the
rank-1 factors of dPwn_dzm (K~ rows, R_wc ray z_n, 1/z_m) so the (b,n,3,m) tensor is never built.
"""
import ctypes

import torch

from como_amd import _lib
from como_amd.odom.backend.graph_pair_construction import setup_photometric_pairs

_ws = {}
BLOCK_VARIANT = int(__import__("os").environ.get("COMO_BA_VARIANT", "0"))   # 0 = software-pipelined block kernel, 1 = plain
CHUNKS_OVERRIDE = int(__import__("os").environ.get("COMO_BA_CHUNKS", "0"))   # tuning runs: pixel chunks per pair group
BLOCK_STAGGER = int(__import__("os").environ.get("COMO_BA_STAGGER", "0"))
ASM_GROUPED = __import__("os").environ.get("COMO_BA_ASM_GROUPED", "1") != "0"   # 0: the assembly expands / scatters every pair on its own (A/B)
ASM_GROUPED_MIN_PAIRS = 3      # the grouped assembly runs from this many pairs per reference keyframe on average (see linearize)
MIN_TILES_PER_CHUNK = int(__import__("os").environ.get("COMO_BA_MIN_TILES", "8"))   # lower bound on a chunk's 64-pixel tiles (0: off; see linearize)
last_aux = {}   # diagnostics of the most recent call: valid mask, sigma, nvalid (tests / callers that want them)


def _buf(name, shape, dtype, device, ws=None):
    """Scratch tensor `name` from the workspace dict `ws` (None: the process-wide one, for one-shot eager calls only --
    whoever captures a graph or keeps several windows alive passes its OWN dict: a graph records raw addresses, and a
    shared buffer that grows on a later, larger request would be freed under it)."""
    ws = _ws if ws is None else ws
    key = (name, str(device))
    t = ws.get(key)
    numel = 1
    for s in shape:
        numel *= int(s)
    if t is None or t.dtype != dtype or t.numel() < numel:
        t = torch.empty(max(numel, 1), device=device, dtype=dtype)
        ws[key] = t
    return t[:numel].view(*shape)


def default_chunks(b, n, dtype, per_cu=None):
    """Pixel chunks per pair (or per pair group): about two resident rounds of 256-thread workgroups on 256 CUs."""
    if per_cu is None:
        per_cu = 3 if dtype == torch.float32 else 1
    want = max(1, (2 * 256 * per_cu + b - 1) // b)
    return int(max(1, min(want, (n + 255) // 256)))


def linearize(*, dtype, b, n, m, H_img, W_img, zmode, Pwn, vals, dPwn_dTwc, zjac, poses_all, aff_all, img_base, K,
              ref_slot, ref_aff, tgt_aff, tgt_pose, tgt_img, pose_ref_inds, pose_tgt_inds, landmark_inds, dzdP, H, g,
              err_out, uvec=None, pixidx=None, invz=None, kt_slot_stride=0, chunks=None, phase=0xFF, want_pj=False,
              want_blocks=False, sigma_out=None, pix_range=None, reduce_hists=None, events=None, anorm_f32=False,
              grp_pairs=None, single_pairs=None, zeroed_hists=None, ws=None, sysfix=None, fix_plane=0, D=None,
              reduce_blocks=None, channels=1, pair_chan=None, ref_pose=None, prepared=None, asm_groups=None):
    """Thin marshalling layer over como_ba_linearize_* (see include/como_hip.h for every field).

    prepared: a caller-owned dict that lives as long as NOTHING about the call changes (same tensors, same sizes: one window
        topology): the first call leaves the marshalled argument block and the phase runner in it, every later call only replays
        the phases -- the ~70 attribute stores / pointer look-ups of this function were ~0.15 ms of host time per Gauss-Newton
        iteration, in front of kernels that take 10-30 us (single-GPU chains without events / reductions only).

    pix_range=(begin, end): this rank's share of the reference pixels of every pair (multi-GPU shard).
    reduce_hists(view): called on the (2048,) int32 histogram of each radix-select digit pass right after it is produced
        (multi-GPU: an all-reduce(sum), so every rank resolves the same exact median); splits the chain into phases.
    events: optional dict filled with (start, end) torch.cuda.Event pairs around the block kernel ("blocks").
    ws: caller-owned workspace dict (residuals, validity, pair tables, partials, ...); see _buf.
    sysfix / fix_plane / D: ORDER-INDEPENDENT assembly into the fixed-point system buffer (include/como_hip.h,
        como_sys_finalize) instead of floating-point atomics into H / g / err_out (which may then be None).
    reduce_blocks(t): multi-GPU with sysfix: called on the (b, 3936, 2) int64 fixed-point per-pair sums between the
        reduce and the expand stage (an integer all-reduce(sum): exact, so every rank continues with identical bits).
    ref_pose: zmode 2 (compact dense reference: dPwn_dTwc = the six planes dlogz_n/dT_wc, no uvec): (b,) int32 index into
        poses_all of every pair's REFERENCE keyframe pose.
    channels / pair_chan: colour images (`color: rgb`): vals is (slots,n,c), the image stacks are (3c,H,W) and every keyframe
        pair appears c times in the pair arrays, once per channel (`expand_channels`); b counts those entries."""
    if prepared is not None and "run" in prepared:
        prepared["run"](phase)
        last_aux.clear()
        last_aux.update(prepared["aux"])
        return prepared["aux"]["sigma"]
    dev = Pwn.device
    L = _lib.lib()
    pb, pe = pix_range if pix_range is not None else (0, n)
    nl = pe - pb
    idle = nl <= 0          # an idle shard (more ranks than 64-pixel tiles) owns nothing but takes part in every collective
    if idle:
        pb, pe, nl = 0, 1, 1                       # placeholder geometry for the argument struct; no pixel kernel is launched
    if chunks is None and CHUNKS_OVERRIDE > 0:
        chunks = min(CHUNKS_OVERRIDE, (nl + 255) // 256)
    if chunks is None:
        if grp_pairs is not None and grp_pairs.numel() > 0 and zmode == 2 and BLOCK_VARIANT not in (1, 2, 3, 8) and dtype == torch.float64:
            # float64 two-pair kernels (wave-specialised: 192 threads, three waves per SIMD; role-split: 128 threads, two waves
            # per SIMD): 4 workgroups per CU = one resident round of 1024
            chunks = default_chunks(grp_pairs.shape[0], nl, dtype, per_cu=2)
        elif grp_pairs is not None and grp_pairs.numel() > 0 and zmode == 2 and BLOCK_VARIANT in (0, 3, 8):
            # two-pair kernels: 2 workgroups per CU resident (f32: 256 threads, 2 waves / SIMD; f64: 128 threads, 1 wave / SIMD)
            # -> ONE resident round of 512 workgroups (measured, dense 8-keyframe window: 64 chunks x 8 groups 313 us,
            # 128 x 8 = two rounds 338 us in float32; 841 vs 882 us in float64)
            chunks = default_chunks(grp_pairs.shape[0], nl, dtype, per_cu=1)
        else:
            chunks = default_chunks(b, nl, dtype)
        if grp_pairs is not None and grp_pairs.numel() > 0 and zmode == 2:
            # A lower bound (8) on a chunk's 64-pixel tiles for sub-selected windows (n = 19,200 = 300 tiles at window 4: 75 -> 38 chunks,
            # the block kernel's fixed prologue / epilogue is shared by twice the tiles and the assembly reduces half the partial records),
            # itself bounded so that a launch never has fewer than one workgroup per compute unit -- the sequential loop's filling windows
            # have one to three pair groups, and fatter workgroups alone would only stretch their kernel (round 4: -4 % on the loop
            # without that bound).  Round 5, same box, two repetitions (scripts/ab/min_tiles_ab.sh, profiles/r5_min_tiles_ab.txt): window-4
            # float64 2342 / 2346 -> 2446 / 2444 it/s, float32 2822 / 2776 -> 2940 / 2918, the pinned odometry loop 294 / 313 -> 303 / 294
            # frames/s (unchanged within its spread).  COMO_BA_MIN_TILES=0 switches it off.
            if MIN_TILES_PER_CHUNK > 0:
                tiles = (nl + 63) // 64
                ngrp = max(1, int(grp_pairs.shape[0]))
                lowest = max((tiles + MIN_TILES_PER_CHUNK - 1) // MIN_TILES_PER_CHUNK, (256 + ngrp - 1) // ngrp)
                chunks = int(max(1, min(chunks, lowest)))
    a = _lib.BAArgs()
    a.b, a.n, a.m, a.H, a.W, a.zmode, a.chunks, a.phase = b, n, m, H_img, W_img, zmode, chunks, phase
    a.pix_begin, a.pix_end = pb, pe
    a.anorm_f32 = 1 if anorm_f32 else 0
    if grp_pairs is not None and grp_pairs.numel() > 0:
        a.grp_pairs, a.ngrp = _lib.ptr(grp_pairs), grp_pairs.shape[0]
        a.single_pairs = _lib.ptr(single_pairs) if single_pairs is not None and single_pairs.numel() > 0 else None
        a.nsingle = single_pairs.numel() if single_pairs is not None else 0
    a.variant = BLOCK_VARIANT
    a.stagger = BLOCK_STAGGER
    a.channels = int(channels)
    if zmode == 2:
        if ref_pose is None or ref_pose.numel() != b or ref_pose.dtype != torch.int32 or not ref_pose.is_contiguous():
            raise RuntimeError("como_amd: zmode 2 needs ref_pose, a contiguous int32 tensor with one pose index per pair entry")
        _lib.require_cuda(ref_pose)
        a.ref_pose = _lib.ptr(ref_pose)
    if channels > 1:
        if pair_chan is None or pair_chan.numel() != b or pair_chan.dtype != torch.int32 or not pair_chan.is_contiguous():
            raise RuntimeError("como_amd: pair_chan must be a contiguous int32 tensor with one channel per pair entry")
        _lib.require_cuda(pair_chan)
        a.pair_chan = _lib.ptr(pair_chan)
    # (only where a group holds three or more pairs on average -- the sequential loop's windows with their one-way frames: with
    # the two pairs per keyframe of a plain window the grouped launch has half the workgroups and is slower, 39 vs 29 us at 14 pairs)
    if (asm_groups is not None and ASM_GROUPED and sysfix is not None and reduce_blocks is None and not want_blocks and
            ASM_GROUPED_MIN_PAIRS * int(asm_groups[2]) <= b):
        # (start offsets, pair list, number of groups): the assembly sums what the pairs of one reference keyframe share first
        a.asm_grp_start, a.asm_grp_list, a.n_asm_grp = _lib.ptr(asm_groups[0]), _lib.ptr(asm_groups[1]), int(asm_groups[2])
    if sysfix is not None:
        a.h_is_f64, a.fix_plane = 2, int(fix_plane)
    else:
        a.h_is_f64 = 1 if H.dtype == torch.float64 else 0
    blocks_fix = _buf("blocks_fix", (b, 3936, 2), torch.int64, dev, ws) if reduce_blocks is not None else None
    a.blocks_fix = _lib.ptr(blocks_fix)
    ws_r = _buf("r", (b, nl), dtype, dev, ws)
    ws_valid = _buf("valid", (b, nl), torch.uint8, dev, ws)
    # zeroed_hists: caller-owned select workspace that is ALREADY zero (the fused window path clears it elsewhere)
    ws_hists = zeroed_hists if zeroed_hists is not None else _buf("hists", (L.como_select_workspace_bytes() // 4,), torch.int32, dev, ws)
    ws_pair = _buf("pair", (b * 26,), dtype, dev, ws)
    ws_part = _buf("partials", (L.como_ba_partials_elems(b, chunks, m),), dtype, dev, ws)
    if sigma_out is None:
        sigma_out = _buf("sigma", (2,), dtype, dev, ws)
    pj = _buf("pj", (b, nl, 2), dtype, dev, ws) if want_pj else None
    blocks = _buf("blocks", (b, 3936), torch.float64, dev, ws) if want_blocks else None
    keep = [Pwn, vals, dPwn_dTwc, zjac, uvec, pixidx, invz, poses_all, aff_all, img_base, K, ref_slot, ref_aff, tgt_aff,
            tgt_pose, tgt_img, pose_ref_inds, pose_tgt_inds, landmark_inds, dzdP, H, g, err_out]
    _lib.require_cuda(*[t for t in keep if t is not None])
    for name, t in (("Pwn", Pwn), ("vals", vals), ("dPwn_dTwc", dPwn_dTwc), ("zjac", zjac), ("uvec", uvec),
                    ("pixidx", pixidx), ("invz", invz), ("poses_all", poses_all), ("aff_all", aff_all), ("K", K),
                    ("ref_slot", ref_slot), ("ref_aff", ref_aff), ("tgt_aff", tgt_aff), ("tgt_pose", tgt_pose),
                    ("tgt_img", tgt_img), ("pose_ref_inds", pose_ref_inds), ("pose_tgt_inds", pose_tgt_inds),
                    ("landmark_inds", landmark_inds), ("dzdP", dzdP), ("Hmat", H), ("gvec", g), ("err_out", err_out)):
        if t is not None and not t.is_contiguous():
            raise RuntimeError(f"como_amd: {name} must be contiguous")
        setattr(a, name, _lib.ptr(t))
    a.img_base = img_base if isinstance(img_base, int) else _lib.ptr(img_base)
    a.kt_slot_stride = int(kt_slot_stride)
    a.D = int(D) if D is not None else H.shape[1]
    if sysfix is not None:
        _lib.require_cuda(sysfix)
        a.Hmat = _lib.ptr(sysfix)
    a.sigma_out, a.pj_out, a.pair_blocks_out = _lib.ptr(sigma_out), _lib.ptr(pj), _lib.ptr(blocks)
    a.ws_r, a.ws_valid, a.ws_hists, a.ws_pair, a.ws_partials = (_lib.ptr(ws_r), _lib.ptr(ws_valid), _lib.ptr(ws_hists),
                                                                 _lib.ptr(ws_pair), _lib.ptr(ws_part))
    fn = getattr(L, "como_ba_linearize_" + _lib.suffix(dtype))
    zflag = 256 if zeroed_hists is not None else 0

    def run(ph, mode=0):
        a.phase = ph | zflag
        a.reduce_mode = mode
        _lib.check(fn(ctypes.byref(a), _lib.stream_ptr(dev)), "como_ba_linearize")     # (the CURRENT stream: eager, warm-up or capture)

    def assemble():
        if reduce_blocks is None:
            if not idle:
                run(128)
            return
        if idle:
            blocks_fix.zero_()
        else:
            run(128, 1)                            # this shard's per-pair sums, fixed point
        reduce_blocks(blocks_fix)                  # exact integer all-reduce
        run(128, 2)                                # every rank expands / scatters the same bits

    if reduce_hists is None and events is None and reduce_blocks is None:
        run(phase)
    else:
        npass = 3 if dtype == torch.float32 else 6
        if idle and zeroed_hists is None:
            ws_hists.zero_()
        if not idle:
            run(1)
        # float64 keys, multi-GPU: after three all-reduced digits the few keys that still match are exchanged ONCE
        # (reduce_hists.candidates: pack + all-gather + merge, csrc/select.hip) instead of three more histogram all-reduces
        cand = getattr(reduce_hists, "candidates", None) if dtype == torch.float64 else None
        for ps in range(npass):
            if reduce_hists is not None:
                reduce_hists(ws_hists[ps * 2048:(ps + 1) * 2048])
            if cand is not None and ps == 2:
                if not idle:
                    run((2 << 2) | 512)                    # pass 3: collect the candidates, no local tail
                cand(ws_hists)
                break
            if ps + 1 < npass and not idle:
                run(2 << ps)
        if idle:
            pass
        elif events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(64)
            e1.record()
            events.setdefault("blocks", []).append((e0, e1))
        else:
            run(64)
        assemble()
    last_aux.clear()
    last_aux.update({"valid": ws_valid, "r": ws_r, "sigma": sigma_out, "pj": pj, "blocks": blocks, "hists": ws_hists,
                     "chunks": chunks, "pair": ws_pair, "b": b})
    if prepared is not None and reduce_hists is None and events is None and reduce_blocks is None:
        # (the argument block `a` and every tensor it points to stay alive in the closure / the aux dict)
        prepared["run"] = run
        prepared["aux"] = dict(last_aux)
        prepared["keep"] = keep + [ws_r, ws_valid, ws_hists, ws_pair, ws_part, sigma_out, blocks_fix, pj, blocks, ref_pose, pair_chan,
                                   grp_pairs, single_pairs, sysfix]
    return sigma_out


def _i32(x, device):
    return torch.as_tensor(x, dtype=torch.int32, device=device).contiguous()


def expand_channels(c, *per_pair):
    """A c-channel keyframe pair is c entries of the pair arrays (include/como_hip.h, como_ba_args.channels): entry p * c + ch
    = pair p, channel ch.  Returns the repeated per-pair tensors and the (b * c) int32 channel of every entry."""
    out = [t.repeat_interleave(c, dim=0).contiguous() for t in per_pair]
    b = per_pair[0].shape[0]
    chan = torch.arange(c, dtype=torch.int32, device=per_pair[0].device).repeat(b).contiguous()
    return out, chan


def batch_photo_cost(vals_i, aff_params_i, Pwn, Twcj, aff_params_j, img_and_grads_j, dPwn_dTwci, dPwn_dzm, dzm_dPwm,
                     pose_ref_inds, pose_target_inds, landmark_inds, intrinsics, H, g):
    """reference photo.py:83-233: accumulates the photometric normal equations of b pairs into H, g; returns the
    robustified total error (0-dim tensor).  c = vals_i.shape[2] image channels (gray 1, rgb 3)."""
    b, n, _, m, _ = dPwn_dzm.shape
    c = vals_i.shape[2]
    if img_and_grads_j.shape[1] != 3 * c:
        raise RuntimeError("como_amd: img_and_grads_j must hold 3 c channels [I | dI/dx | dI/dy] for vals_i of c channels")
    if m % 4 != 0 or m > 64:
        raise RuntimeError("como_amd: m must be a multiple of 4 and <= 64")
    dev, dt = vals_i.device, vals_i.dtype
    Hh, Ww = img_and_grads_j.shape[-2:]
    ar = torch.arange(b, dtype=torch.int32, device=dev)
    aff_all = torch.cat((aff_params_i.reshape(b, 2), aff_params_j.reshape(b, 2)), dim=0).contiguous()
    img = img_and_grads_j.contiguous()
    err = torch.zeros((), dtype=torch.float64, device=dev)
    tgt_img = torch.arange(b, dtype=torch.int64, device=dev) * (3 * c * Hh * Ww)
    pri, pti, lmi, chan = pose_ref_inds, pose_target_inds, landmark_inds, None
    if c > 1:
        (ar, tgt_img, pri, pti, lmi), chan = expand_channels(c, ar, tgt_img, pri, pti, lmi)
    linearize(dtype=dt, b=b * c, n=n, m=m, H_img=Hh, W_img=Ww, zmode=0, Pwn=Pwn.contiguous(), vals=vals_i.contiguous(),
              dPwn_dTwc=dPwn_dTwci.contiguous(), zjac=dPwn_dzm.contiguous(), poses_all=Twcj.contiguous(), aff_all=aff_all,
              img_base=img, K=intrinsics.contiguous(), ref_slot=ar, ref_aff=ar, tgt_aff=(ar + b).contiguous(), tgt_pose=ar,
              tgt_img=tgt_img.contiguous(), pose_ref_inds=pri.contiguous(), pose_tgt_inds=pti.contiguous(),
              landmark_inds=lmi.contiguous(), dzdP=dzm_dPwm[:, 0, 0, :].contiguous(), H=H, g=g, err_out=err,
              want_pj=True, channels=c, pair_chan=chan)
    # per keyframe pair as the reference returns them: the mask / projection do not depend on the channel; r is (b,n,c)
    last_aux["valid"] = last_aux["valid"][::c].bool()
    if c > 1:
        last_aux["pj"] = last_aux["pj"][::c]
        last_aux["r"] = last_aux["r"].view(b, c, -1).transpose(1, 2)
    return err.to(dt)


def create_photo_system(kf_poses, kf_aff_params, recent_poses, recent_aff_params, Pwn, dPwn_dTwc, dPwn_dzm, dzm_dPwm,
                        median_depths, vals_n, kf_img_and_grads, recent_img_and_grads, kf_timestamps, recent_timestamps,
                        intrinsics, H, g, photo_construction_cfg, kf_inds, recent_inds, landmark_inds):
    """reference photo.py:236-353: pair graph + batches of <= pairwise_batch_size pairs (the robust scale is per batch).
    Returns (err, [kf_ref_ids, kf_target_ids], [one_way_kf_ids, one_way_target_ids])."""
    kf_ref_ids, kf_target_ids, ow_kf_ids, ow_target_ids = setup_photometric_pairs(
        kf_poses, recent_poses, kf_timestamps, recent_timestamps, median_depths, photo_construction_cfg)
    all_ref = kf_ref_ids + ow_kf_ids
    nkf = len(kf_target_ids)
    bs = photo_construction_cfg["pairwise_batch_size"]
    dev = kf_poses.device
    err = 0.0
    for b1 in range(0, len(all_ref), bs):
        b2 = min(b1 + bs, len(all_ref))
        rid = torch.as_tensor(all_ref[b1:b2], device=dev)
        kt = torch.as_tensor(kf_target_ids[b1:min(b2, nkf)], dtype=torch.long, device=dev)
        rt = torch.as_tensor(ow_target_ids[max(b1, nkf) - nkf:max(b2, nkf) - nkf], dtype=torch.long, device=dev)
        if rt.numel():
            t_pose = torch.cat((kf_poses[kt], recent_poses[rt]))
            t_aff = torch.cat((kf_aff_params[kt], recent_aff_params[rt]))
            t_img = torch.cat((kf_img_and_grads[kt], recent_img_and_grads[rt]))
            t_ind = torch.cat((kf_inds[kt], recent_inds[rt]))
        else:
            t_pose, t_aff, t_img, t_ind = kf_poses[kt], kf_aff_params[kt], kf_img_and_grads[kt], kf_inds[kt]
        err = err + batch_photo_cost(vals_n[rid], kf_aff_params[rid], Pwn[rid], t_pose, t_aff, t_img, dPwn_dTwc[rid],
                                     dPwn_dzm[rid], dzm_dPwm[rid], kf_inds[rid], t_ind, landmark_inds[rid],
                                     intrinsics[0, ...], H, g)
    return err, [kf_ref_ids, kf_target_ids], [ow_kf_ids, ow_target_ids]


class PairTable:
    """Device-resident description of the pair graph of one window topology (built once per keyframe change,
    reused by every GN iteration: no per-iteration host->device traffic)."""

    def __init__(self, ref_ids, tgt_ids, tgt_is_recent, num_kf, kf_inds, recent_inds, landmark_inds, img_stride,
                 recent_img_offset, device, channels=1, landmark_inds_host=None, extra_i64=None):
        """landmark_inds_host: numpy (num_kf, 3m) copy of `landmark_inds` when the caller keeps one: the per-pair rows are then built
        here and travel in the table's one upload (no device gather).  extra_i64: optional dict name -> int64 numpy array uploaded in
        the same block and left in `self.extra[name]` (the window's own small index tables that move with the frame count)."""
        import numpy as np
        # colour images: every keyframe pair becomes `channels` consecutive entries, one per channel (como_ba_args.channels);
        # img_stride is the size of a frame's whole (3c,H,W) stack
        self.channels = c = int(channels)
        self.npairs = len(ref_ids)
        ref = np.repeat(np.asarray(ref_ids, dtype=np.int64), c)
        tgt = np.repeat(np.asarray(tgt_ids, dtype=np.int64), c)
        rec = np.repeat(np.asarray(tgt_is_recent, dtype=bool), c)
        b = int(ref.shape[0])
        self.b = b
        tgt_frame = tgt + np.where(rec, num_kf, 0)
        off = np.where(rec, recent_img_offset + tgt * img_stride, tgt * img_stride).astype(np.int64)
        chan = (np.arange(b) % c).astype(np.int64)
        # pairs sharing their reference keyframe (AND channel), two at a time (csrc/ba.hip ba_blocks_pair2_kernel); a lone pair rides
        # the same kernel with its second half masked.  Order: by first appearance of (reference, channel), pairs in index order.
        grp = []
        by_ref = {}
        for p_ in range(b):
            by_ref.setdefault((int(ref[p_]), int(chan[p_])), []).append(p_)
        for lst in by_ref.values():
            for q in range(0, len(lst) - 1, 2):
                grp.append((lst[q], lst[q + 1]))
            if len(lst) % 2:
                grp.append((lst[-1], -1))
        flat_grp = np.asarray(grp, dtype=np.int64).reshape(-1)
        ng = int(flat_grp.shape[0])
        # the pairs grouped by reference keyframe for the assembly (como_ba_args.asm_grp_*): group starts, then the pair list
        by_slot = {}
        for p_ in range(b):
            by_slot.setdefault(int(ref[p_]), []).append(p_)
        asm_list = np.asarray([p_ for lst in by_slot.values() for p_ in lst], dtype=np.int64)
        asm_start = np.cumsum([0] + [len(lst) for lst in by_slot.values()]).astype(np.int64)
        self.n_asm_grp = len(by_slot)
        # ... and per reference keyframe (slot) its pairs, -1 padded: the residual pass fused into the dense reference walks them
        self.np_max = max(len(v) for v in by_slot.values()) if by_slot else 1
        refp = -np.ones((num_kf, self.np_max), dtype=np.int64)
        for slot_, lst in by_slot.items():
            if 0 <= slot_ < num_kf:
                refp[slot_, :len(lst)] = lst
        # every host-built index array -- including the system rows of the reference / target FRAMES, which are plain ramps
        # (frame f owns rows 8 f .. 8 f + 7: `kf_inds` / `recent_inds` are views of one arange) -- in ONE host->device copy of a
        # pinned staging block: the window's pair table is rebuilt on every keyframe / one-way frame of the sequential loop
        ramp = _frame_rows_are_ramps(kf_inds, recent_inds, num_kf)
        ar8 = np.arange(8, dtype=np.int64)[None]
        rows_ref = (8 * ref[:, None] + ar8).reshape(-1) if ramp else np.zeros(0, np.int64)
        rows_tgt = (8 * tgt_frame[:, None] + ar8).reshape(-1) if ramp else np.zeros(0, np.int64)
        lm_rows = landmark_inds_host[ref].reshape(-1).astype(np.int64) if landmark_inds_host is not None else np.zeros(0, np.int64)
        extra = extra_i64 or {}
        h32 = np.concatenate((ref, tgt_frame, chan, flat_grp, asm_start, asm_list, refp.reshape(-1))).astype(np.int32)
        if h32.shape[0] % 2:
            h32 = np.concatenate((h32, np.zeros(1, np.int32)))
        h64 = np.concatenate([off, ref, rows_ref, rows_tgt, lm_rows] + [np.asarray(v, dtype=np.int64).reshape(-1) for v in extra.values()])
        nb32, nb = h32.nbytes, h32.nbytes + h64.nbytes                   # (int32 block first, padded to 8 bytes; then the int64 block)
        dev_t = torch.device(device)
        if dev_t.type == "cuda":
            stage = _pinned_stage(nb)
            sn = stage.numpy()
            sn[:nb32] = h32.view(np.uint8)
            sn[nb32:nb] = h64.view(np.uint8)
            raw = torch.empty(nb, dtype=torch.uint8, device=dev_t)
            raw.copy_(stage[:nb], non_blocking=True)
            _stage_fence(dev_t)
        else:
            raw = torch.from_numpy(np.concatenate((h32.view(np.uint8), h64.view(np.uint8)))).to(dev_t)
        i32 = raw[:nb32].view(torch.int32)
        i64 = raw[nb32:nb].view(torch.int64)
        self.ref_slot = i32[0:b]
        self.ref_pose = self.ref_slot                       # keyframes come first in the pose buffer: slot b = pose b
        self.ref_aff = self.ref_slot
        self.tgt_aff = i32[b:2 * b]
        self.tgt_pose = self.tgt_aff
        self.pair_chan = i32[2 * b:3 * b] if c > 1 else None
        self.grp_pairs = i32[3 * b:3 * b + ng].reshape(-1, 2)
        o32 = 3 * b + ng
        self.asm_grp_start = i32[o32:o32 + asm_start.shape[0]]
        self.asm_grp_list = i32[o32 + asm_start.shape[0]:o32 + asm_start.shape[0] + b]
        o32 += asm_start.shape[0] + b
        self.ref_pairs = i32[o32:o32 + num_kf * self.np_max].view(num_kf, self.np_max)
        self.tgt_img = i64[0:b]
        rid = i64[b:2 * b]
        o = 2 * b
        if ramp:
            self.pose_ref_inds = i64[o:o + 8 * b].view(b, 8)
            self.pose_tgt_inds = i64[o + 8 * b:o + 16 * b].view(b, 8)
            o += 16 * b
        else:
            self.pose_ref_inds = kf_inds[rid].contiguous()
            # system rows of every target frame with ONE gather (keyframes first, then the one-way frames)
            frame_rows = torch.cat((kf_inds, recent_inds), dim=0) if bool(rec.any()) else kf_inds
            self.pose_tgt_inds = frame_rows[torch.as_tensor(tgt_frame, dtype=torch.int64, device=dev_t)].contiguous()
        if landmark_inds_host is not None:
            w3m = int(landmark_inds_host.shape[1])
            self.landmark_inds = i64[o:o + b * w3m].view(b, w3m)
            o += b * w3m
        else:
            self.landmark_inds = landmark_inds.index_select(0, rid)
        self.extra = {}
        for k_, v in extra.items():
            nv = int(np.asarray(v).size)
            self.extra[k_] = i64[o:o + nv].view(tuple(np.asarray(v).shape))
            o += nv
        self.single_pairs = _empty_i32(dev_t)
        self.ngroups = len(grp)


_stage = {}


def _pinned_stage(n):
    """A pinned byte staging block (grow-only) for the pair table's one host->device copy."""
    t = _stage.get("buf")
    if t is None or t.numel() < n:
        # (1 MiB from the start: a pinned allocation is a 40 ms driver call -- growing from 64 KiB when the window first held 40
        # pairs stalled one frame of the sequential loop by that much)
        t = _stage["buf"] = torch.empty(max(1 << 20, 2 * n), dtype=torch.uint8).pin_memory()
        _stage["event"] = None
    else:
        ev = _stage.get("event")
        if ev is not None:
            ev.synchronize()                                  # the previous table's copy has left the block (long ago)
    return t


def _stage_fence(device):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _stage["event"] = ev


_empties = {}


def _empty_i32(device):
    t = _empties.get(str(device))
    if t is None:
        t = _empties[str(device)] = torch.zeros(0, dtype=torch.int32, device=device)
    return t


def _frame_rows_are_ramps(kf_inds, recent_inds, num_kf):
    """True when the caller's row tables are the window's own ramps (frame f -> rows 8 f .. 8 f + 7): shapes only -- WindowBA builds
    them as views of one arange; any other caller (tests with hand-made tables) takes the gather path."""
    return getattr(kf_inds, "_como_ramp", False) and (recent_inds.numel() == 0 or getattr(recent_inds, "_como_ramp", False))


def photo_system_factored(table, *, poses_all, aff_all, Pwn, vals, dPwn_dTwc, uvec, Kt, pixidx, invz, dzdP, img_base, K,
                          H_img, W_img, H, g, err_out, chunks=None, phase=0xFF, sigma_out=None, pix_range=None,
                          reduce_hists=None, events=None, zeroed_hists=None, ws=None, sysfix=None, fix_plane=0, D=None,
                          reduce_blocks=None, prepared=None):
    """Fast path: same normal equations as batch_photo_cost from the rank-1 factors of dPwn_dzm.
    Per-keyframe arrays (slots = keyframes), structure-of-arrays planes: Pwn (B,3,n) vals (B,n[,c]) invz (B,m)
    Kt (B,rows,m) dense predictor, pixidx (B,n) int32 rows of Kt (None = identity), dzdP (B,3), and either
      uvec (B,3,n) + dPwn_dTwc (B,18,n): materialised pose Jacobian (como_ba_args.zmode 1, plain kernel), or
      uvec None + dPwn_dTwc (B,6,n) = dlogz_n/dT_wc: the compact dense reference (zmode 2, the tuned kernels; the
      reference poses are poses_all[table.ref_pose])."""
    if prepared is not None and "run" in prepared:
        return linearize(dtype=None, b=0, n=0, m=0, H_img=0, W_img=0, zmode=0, Pwn=None, vals=None, dPwn_dTwc=None, zjac=None,
                         poses_all=None, aff_all=None, img_base=None, K=None, ref_slot=None, ref_aff=None, tgt_aff=None, tgt_pose=None,
                         tgt_img=None, pose_ref_inds=None, pose_tgt_inds=None, landmark_inds=None, dzdP=None, H=None, g=None,
                         err_out=None, phase=phase, prepared=prepared)
    B, n = vals.shape[:2]                                  # (B,n) gray or (B,n,c)
    if (vals.shape[2] if vals.dim() == 3 else 1) != table.channels:
        raise RuntimeError("como_amd: vals must be (B,n,c) with the pair table's channel count")
    m = Kt.shape[-1]
    return linearize(dtype=vals.dtype, b=table.b, channels=table.channels, pair_chan=table.pair_chan, n=n, m=m, H_img=H_img, W_img=W_img,
                     zmode=(2 if uvec is None else 1), ref_pose=(table.ref_pose if uvec is None else None), Pwn=Pwn, vals=vals,
                     dPwn_dTwc=dPwn_dTwc, zjac=Kt, uvec=uvec, pixidx=pixidx, invz=invz,
                     kt_slot_stride=Kt.stride(0), poses_all=poses_all, aff_all=aff_all, img_base=img_base, K=K,
                     ref_slot=table.ref_slot, ref_aff=table.ref_aff, tgt_aff=table.tgt_aff, tgt_pose=table.tgt_pose,
                     tgt_img=table.tgt_img, pose_ref_inds=table.pose_ref_inds, pose_tgt_inds=table.pose_tgt_inds,
                     landmark_inds=table.landmark_inds, dzdP=dzdP, H=H, g=g, err_out=err_out, chunks=chunks,
                     phase=phase, sigma_out=sigma_out, pix_range=pix_range, reduce_hists=reduce_hists, events=events,
                     grp_pairs=table.grp_pairs, single_pairs=table.single_pairs,
                     zeroed_hists=zeroed_hists, ws=ws, sysfix=sysfix, fix_plane=fix_plane, D=D, reduce_blocks=reduce_blocks,
                     prepared=prepared, asm_groups=(table.asm_grp_start, table.asm_grp_list, table.n_asm_grp))
