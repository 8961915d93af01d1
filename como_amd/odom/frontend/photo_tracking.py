"""Drop-in for the reference's como/odom/frontend/photo_tracking.py on MI355X.

Same names, argument meaning and return values as the reference functions; the per-pixel work of
`tracking_iter` (warp, sampling, residual, validity mask, exact median, Huber-weighted 8x8 normal
equations, solve, pose update) runs in the HIP kernel chain of csrc/track.hip through the C ABI
(`como_track_iter_*`; `como_track_*_channels_*` for colour images, c = 3).  No CPU fallback.
"""
import torch

from como_amd import _lib
from como_amd.geometry.lie_algebra import skew_symmetric

_ws_cache = {}


def _new_workspace(device, dtype, N):
    """N = residual entries = pixels x image channels."""
    L = _lib.lib()
    return {"r": torch.empty(N, device=device, dtype=dtype),
            "hists": torch.empty(L.como_select_workspace_bytes() // 4, device=device, dtype=torch.int32),
            "partials": torch.empty(L.como_track_partials_bytes() // 8, device=device, dtype=torch.float64)}


def _workspace(device, dtype, N):
    """Scratch of the eager entry points (one per size, a few kept).  Captured graphs own theirs (_LevelGraph.ws): a graph
    records raw addresses, so its scratch must live exactly as long as the graph."""
    key = (str(device), dtype, N)
    ws = _ws_cache.get(key)
    if ws is None:
        if len(_ws_cache) > 8:
            _ws_cache.clear()
        ws = _new_workspace(device, dtype, N)
        _ws_cache[key] = ws
    return ws


def precalc_jacobians(dI_dw, P, vals, intrinsics):
    """Inverse-compositional Jacobians at theta = 0 (reference photo_tracking.py:46-74).
    dI_dw (B,N,c,2), P (B,N,3), vals (B,N,c) -> (B,N,c,8).  Gray CUDA input runs csrc/image.hip."""
    if P.is_cuda and dI_dw.shape[2] == 1:
        B, N = P.shape[:2]
        dt = P.dtype
        J = torch.empty((B, N, 1, 8), dtype=dt, device=P.device)
        fn = getattr(_lib.lib(), "como_track_precalc_jac_" + _lib.suffix(dt))
        _lib.check(fn(dI_dw.to(dt).contiguous().data_ptr(), P.contiguous().data_ptr(), vals.to(dt).contiguous().data_ptr(),
                      intrinsics.to(dt).contiguous().data_ptr(), J.data_ptr(), B * N, _lib.stream_ptr(P.device)),
                   "como_track_precalc_jac")
        return J
    fx, fy = intrinsics[0, 0], intrinsics[1, 1]
    X, Y, Z = P[..., 0], P[..., 1], P[..., 2]
    zero = torch.zeros_like(Z)
    dp_dP = torch.stack((fx / Z, zero, -(fx * X / Z) / Z, zero, fy / Z, -(fy * Y / Z) / Z), dim=-1)
    dp_dP = dp_dP.reshape(P.shape[:-1] + (2, 3))
    eye = torch.eye(3, dtype=P.dtype, device=P.device).expand(P.shape[:-1] + (3, 3))
    dP_dT = torch.cat((-skew_symmetric(P), eye), dim=-1)
    dI_dT = dI_dw @ (dp_dP @ dP_dT)
    return torch.cat((dI_dT, vals.unsqueeze(-1), torch.ones_like(vals).unsqueeze(-1)), dim=-1)


def tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, want_proj=True, in_mask=None, ws=None):
    """Enqueue one GN iteration; returns (out[105], valid u8 (N,), pj (N,2) or None, depth (N,) or None).
    in_mask (N,) uint8: reference-side selection applied in the kernel (0 = ignore the point).
    c = img_j.shape[1] image channels: vals_i (1,N,c), dI_dT (1,N,c,8) (column 6 is overwritten in place as the reference does)."""
    _lib.require_cuda(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    if img_j.shape[0] != 1 or Pi.shape[0] != 1:
        raise RuntimeError("como_amd tracking: batch 1 only")
    dt, dev = Pi.dtype, Pi.device
    N = Pi.shape[1]
    c = img_j.shape[1]
    if vals_i.numel() != N * c or dI_dT.numel() != N * c * 8:
        raise RuntimeError("como_amd tracking: vals_i must be (1,N,c) and dI_dT (1,N,c,8) for an image of c channels")
    H, W = img_j.shape[-2:]
    for t in (Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT):
        if not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("como_amd tracking: inputs must be contiguous and share one dtype")
    ws = ws if ws is not None else _workspace(dev, dt, N * c)
    out = torch.empty(105, device=dev, dtype=dt)
    valid = torch.empty(N * c, device=dev, dtype=torch.uint8)
    pj = torch.empty((N, 2), device=dev, dtype=dt) if want_proj else None
    depth = torch.empty(N, device=dev, dtype=dt) if want_proj else None
    args = [_lib.ptr(Tji), _lib.ptr(intrinsics), _lib.ptr(aff), _lib.ptr(Pi), _lib.ptr(vals_i), _lib.ptr(img_j), H, W, N,
            _lib.ptr(dI_dT), _lib.ptr(ws["r"]), _lib.ptr(valid), _lib.ptr(pj), _lib.ptr(depth), _lib.ptr(ws["hists"]),
            _lib.ptr(ws["partials"]), _lib.ptr(out)]
    if in_mask is not None and (in_mask.dtype != torch.uint8 or in_mask.numel() != N or not in_mask.is_contiguous()):
        raise RuntimeError("como_amd tracking: in_mask must be a contiguous uint8 tensor of N elements")
    if c > 1:
        fn = getattr(_lib.lib(), "como_track_iter_channels_" + _lib.suffix(dt))
        args.insert(9, c)
        args.append(_lib.ptr(in_mask))
    elif in_mask is None:
        fn = getattr(_lib.lib(), "como_track_iter_" + _lib.suffix(dt))
    else:
        fn = getattr(_lib.lib(), "como_track_iter_masked_" + _lib.suffix(dt))
        args.append(_lib.ptr(in_mask))
    _lib.check(fn(*args, _lib.stream_ptr(dev)), "como_track_iter")
    return out, (valid if c == 1 else valid.view(N, c)[:, 0]), pj, depth


def tracking_iter(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, photo_sigma, A_norm):
    """reference photo_tracking.py:117-143.  Returns the same 8-tuple:
    (Tji_new (1,4,4), aff_new (1,2,1), delta (1,8,1), mean_sq_err, grad_norm, pj (1,N,2), valid_mask (1,N), depth_j (1,N,1)).
    `photo_sigma` and `A_norm` are accepted for signature parity (the reference ignores photo_sigma too; A_norm = 1/size is
    recomputed in-kernel in the same dtype)."""
    out, valid, pj, depth = tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    return (out[80:96].reshape(1, 4, 4), out[96:98].reshape(1, 2, 1), out[72:80].reshape(1, 8, 1), out[98], out[99],
            pj[None], valid[None].bool(), depth[None, :, None])


LOOKAHEAD = 4      # GN iterations enqueued per host read-back of the stop-test scalars


class _LevelGraph:
    """One tracking GN iteration (memset + residual + 2 select passes + reduce + finish + state write-back) captured in a
    hipGraph over FIXED buffers: the pose / affine state lives in `self.T`, `self.aff` and is advanced in place, so the
    host only replays the graph (~35 us per iteration instead of ~300 us of Python + 7 launches)."""

    def __init__(self, vals_i, Pi, dI_dT, img_j, intrinsics, in_mask=None):
        dev, dt = Pi.device, Pi.dtype
        self.args = (Pi, intrinsics, img_j, vals_i, dI_dT)
        self.in_mask = in_mask
        self.T = torch.zeros((1, 4, 4), device=dev, dtype=dt)
        self.aff = torch.zeros((1, 2, 1), device=dev, dtype=dt)
        self.ring = torch.zeros((LOOKAHEAD, 105), device=dev, dtype=dt)      # per-iteration results between read-backs
        self.ws = _new_workspace(dev, dt, Pi.shape[1] * img_j.shape[1])      # owned: the graph records its addresses
        self.out = None
        self.graph = None
        self.dev = dev

    def _iter(self):
        Pi, K, img_j, vals_i, dI_dT = self.args
        out, _, _, _ = tracking_iter_raw(self.T, Pi, K, img_j, self.aff, vals_i, dI_dT, want_proj=False, in_mask=self.in_mask,
                                         ws=self.ws)
        self.T.copy_(out[80:96].reshape(1, 4, 4))
        self.aff.copy_(out[96:98].reshape(1, 2, 1))
        return out

    def capture(self):
        keepT, keepA = self.T.clone(), self.aff.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._iter()                                       # warm-up outside capture (workspace allocation)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.T.copy_(keepT)
        self.aff.copy_(keepA)
        g, out = _lib.capture_graph(self._iter, self.dev)      # (on failure the eager path keeps working)
        self.T.copy_(keepT)                                    # the capture itself does not execute; keep the state exact
        self.aff.copy_(keepA)
        self.graph, self.out = (g, out) if g is not None else (None, None)
        return self.graph is not None

    def step(self):
        if self.graph is not None:
            self.graph.replay()
            return self.out
        return self._iter()


_level_graphs = {}


def _level_graph(vals_i, Pi, dI_dT, img_j, intrinsics, in_mask=None):
    """Graphs are keyed by the buffer addresses: they are meant for PERSISTENT buffers (photo_tracking_pyr copies every new
    reference / frame into the same ones); a caller passing its own tensors must keep them alive and unmodified."""
    key = (vals_i.data_ptr(), Pi.data_ptr(), dI_dT.data_ptr(), img_j.data_ptr(), intrinsics.data_ptr(), Pi.shape[1],
           tuple(img_j.shape), Pi.dtype, in_mask.data_ptr() if in_mask is not None else 0)
    lg = _level_graphs.get(key)
    if lg is None:
        if len(_level_graphs) > 16:
            _level_graphs.clear()
        lg = _LevelGraph(vals_i, Pi, dI_dT, img_j, intrinsics, in_mask)
        lg.capture()
        _level_graphs[key] = lg
    return lg


FUSED_LEVEL = __import__("os").environ.get("COMO_TRACK_FUSED", "1") != "0"
_level_ws = {}
UNCACHED_WS = __import__("os").environ.get("COMO_TRACK_UNCACHED_WS", "1") == "1"   # barrier workspace in uncached device memory


def photo_level_tracking_fused(Tji_init, aff_init, vals_i, Pi, dI_dT, img_j, intrinsics, term_criteria, in_mask=None, out=None,
                               ws_pair=None, prezeroed=False):
    """The whole level in ONE launch (csrc/track.hip track_level_kernel): Gauss-Newton loop + stop test on the device, no
    host read-back.  Returns (Tji (1,4,4), aff (1,2,1), out (106,)) -- all device tensors; out[105] = iterations run.
    None if the level does not fit the persistent kernel (then the per-iteration chain runs)."""
    dev, dt = Pi.device, Pi.dtype
    if dt != torch.float32 or not Pi.is_cuda or img_j.shape[0] != 1 or Pi.shape[0] != 1:
        return None
    L = _lib.lib()
    N = Pi.shape[1]
    c = img_j.shape[1]
    if vals_i.numel() != N * c or dI_dT.numel() != N * c * 8:
        raise RuntimeError("como_amd tracking: vals_i must be (1,N,c) and dI_dT (1,N,c,8) for an image of c channels")
    H, W = img_j.shape[-2:]
    for t in (Tji_init, Pi, intrinsics, img_j, aff_init, vals_i, dI_dT):
        if not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("como_amd tracking: inputs must be contiguous and share one dtype")
    # one barrier workspace per (device, stream): two level launches running concurrently on different streams must not share
    # the barrier counters / recycled histograms
    if ws_pair is not None:                                # a caller-owned workspace (the tracker's captured frame graph)
        ws, wsp = ws_pair
        if out is None:
            out = torch.empty(106, device=dev, dtype=dt)
        # (the cached workspace doubles as the XCD-local one of the coarse levels, csrc/track.hip)
        # prezeroed: the caller cleared BOTH workspaces of the pair on this stream (the frame graph does, in its first launch)
        rc = (L.como_track_level_prezeroed_f32 if prezeroed else L.como_track_level_local_f32)(_lib.ptr(Tji_init), _lib.ptr(intrinsics), _lib.ptr(aff_init), _lib.ptr(Pi),
                                          _lib.ptr(vals_i), _lib.ptr(img_j), H, W, N, c, _lib.ptr(dI_dT), _lib.ptr(in_mask),
                                          int(term_criteria["max_iter"]), float(term_criteria["delta_norm"]),
                                          float(term_criteria["rel_tol"]), float(term_criteria["grad_norm"]),
                                          wsp if wsp else _lib.ptr(ws), 1 if wsp else 0, _lib.ptr(ws), _lib.ptr(out),
                                          _lib.stream_ptr(dev))
        if rc == 1:
            return None
        _lib.check(rc, "como_track_level")
        return out[80:96].reshape(1, 4, 4), out[96:98].reshape(1, 2, 1), out
    key = f"{dev}:{torch.cuda.current_stream(dev).cuda_stream}"
    ws = _level_ws.get(key)
    if ws is None:
        ws = torch.zeros(L.como_track_level_workspace_bytes() // 4, device=dev, dtype=torch.int32)
        _level_ws[key] = ws
    uncached = 0
    ws_ptr = _lib.ptr(ws)
    if UNCACHED_WS and not torch.cuda.is_current_stream_capturing():
        wsp = _level_ws.get(key + ":uc")
        if wsp is None:
            with torch.cuda.device(dev):
                wsp = L.como_track_level_workspace_create()          # uncached device memory, once per device
            _level_ws[key + ":uc"] = wsp or 0
        if wsp:
            ws_ptr, uncached = wsp, 1
    if out is None:
        out = torch.empty(106, device=dev, dtype=dt)
    rc = L.como_track_level_local_f32(_lib.ptr(Tji_init), _lib.ptr(intrinsics), _lib.ptr(aff_init), _lib.ptr(Pi),
                                      _lib.ptr(vals_i), _lib.ptr(img_j), H, W, N, c, _lib.ptr(dI_dT), _lib.ptr(in_mask),
                                      int(term_criteria["max_iter"]), float(term_criteria["delta_norm"]),
                                      float(term_criteria["rel_tol"]), float(term_criteria["grad_norm"]), ws_ptr, uncached,
                                      _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr(dev))
    if rc == 1:                                            # COMO_ERR_ARG: more pixels than the persistent kernel holds
        return None
    _lib.check(rc, "como_track_level")
    return out[80:96].reshape(1, 4, 4), out[96:98].reshape(1, 2, 1), out


def photo_level_tracking(Tji_init, aff_init, vals_i, Pi, dI_dT, img_j, intrinsics, photo_sigma, term_criteria, use_graph=True,
                         in_mask=None, fused=None):
    """reference photo_tracking.py:147-185.  Default: the persistent one-launch kernel (photo_level_tracking_fused);
    fused=False (or COMO_TRACK_FUSED=0) runs the per-iteration chain described next.  The stop test needs 3 scalars of every iteration on the host: LOOKAHEAD
    iterations are enqueued per read-back, their results kept in a small ring; when the test fires at iteration j the state
    of iteration j is restored, so the result is the reference's, at the price of at most LOOKAHEAD-1 surplus iterations.
    use_graph: replay each iteration from a hipGraph over fixed buffers (inputs must stay alive and unmodified)."""
    if (FUSED_LEVEL if fused is None else fused):
        if in_mask is not None and (in_mask.dtype != torch.uint8 or in_mask.numel() != Pi.shape[1] or not in_mask.is_contiguous()):
            raise RuntimeError("como_amd tracking: in_mask must be a contiguous uint8 tensor of N elements")
        res = photo_level_tracking_fused(Tji_init.reshape(1, 4, 4).contiguous(), aff_init.reshape(1, 2, 1).contiguous(), vals_i, Pi,
                                         dI_dT, img_j, intrinsics, term_criteria, in_mask)
        if res is not None:
            photo_level_tracking.last_out = res[2]             # device tensor: [105] = iterations, [104] = status
            return res[0].clone(), res[1].clone()
    if use_graph:
        lg = _level_graph(vals_i, Pi, dI_dT, img_j, intrinsics, in_mask)
    else:
        lg = _LevelGraph(vals_i, Pi, dI_dT, img_j, intrinsics, in_mask)
    lg.T.copy_(Tji_init.reshape(1, 4, 4))
    lg.aff.copy_(aff_init.reshape(1, 2, 1))
    it = 0
    prev = float("inf")
    done = False
    while not done:
        for j in range(LOOKAHEAD):
            lg.ring[j].copy_(lg.step())
        rows = lg.ring.tolist()                                # ONE device->host copy per LOOKAHEAD iterations
        for j in range(LOOKAHEAD):
            mse, gnorm, dnorm = rows[j][98], rows[j][99], rows[j][103]
            it += 1
            if prev == float("inf"):
                rel = float("nan")
            elif prev == 0.0:                                  # torch: 0/0 -> nan, x/0 -> inf (a Python float division raises)
                rel = float("nan") if mse == 0.0 else float("inf")
            else:
                rel = abs((prev - mse) / prev)
            if (it >= term_criteria["max_iter"] or dnorm < term_criteria["delta_norm"] or rel < term_criteria["rel_tol"]
                    or gnorm < term_criteria["grad_norm"]):
                if j + 1 < LOOKAHEAD:                          # roll the state back to iteration j
                    lg.T.copy_(lg.ring[j, 80:96].reshape(1, 4, 4))
                    lg.aff.copy_(lg.ring[j, 96:98].reshape(1, 2, 1))
                done = True
                break
            prev = mse
    photo_level_tracking.last_iters = it
    return lg.T.clone(), lg.aff.clone()


class _PyrBuffers:
    """Persistent device buffers of one pyramid shape: reference arrays (all B*N points of every level, never gathered), the
    selection masks, the current frame's levels and intrinsics.  Their addresses never change, so each level's captured
    iteration graph is built once and serves every keyframe and every frame of the run."""

    def __init__(self, vals_i, Pi, dI_dT, img_j, intrinsics):
        self.levels = []
        for l in range(len(vals_i)):
            n = vals_i[l].shape[0] * vals_i[l].shape[1]
            dev, dt = Pi[l].device, Pi[l].dtype
            self.levels.append({"vals": torch.empty((1, n, vals_i[l].shape[2]), device=dev, dtype=dt),
                                "P": torch.empty((1, n, 3), device=dev, dtype=dt),
                                "dI": torch.empty((1, n) + tuple(dI_dT[l].shape[2:]), device=dev, dtype=dt),
                                "mask": torch.empty((n,), device=dev, dtype=torch.uint8),
                                "img": torch.empty_like(img_j[l]), "K": torch.empty_like(intrinsics[l])})
        self.src = None

    @classmethod
    def from_shapes(cls, b, c, sizes, device, dtype):
        """Private buffers for b reference keyframes of c channels at the pyramid levels `sizes` = [(h, w), ...] coarse -> fine."""
        self = cls.__new__(cls)
        self.levels = []
        for h, w in sizes:
            n = b * h * w
            self.levels.append({"vals": torch.empty((1, n, c), device=device, dtype=dtype),
                                "P": torch.empty((1, n, 3), device=device, dtype=dtype),
                                "dI": torch.empty((1, n, c, 8), device=device, dtype=dtype),
                                "mask": torch.empty((n,), device=device, dtype=torch.uint8),
                                "img": torch.empty((1, c, h, w), device=device, dtype=dtype),
                                "K": torch.empty((3, 3), device=device, dtype=dtype)})
        self.src = None
        return self

    def load_reference(self, vals_i, Pi, dI_dT, masks):
        src = tuple(vals_i) + tuple(Pi) + tuple(dI_dT) + tuple(masks)
        if self.src is not None and len(src) == len(self.src) and all(a is b for a, b in zip(src, self.src)):
            return
        for l, c in enumerate(self.levels):
            # (a caller that produced its reference arrays IN these buffers -- the tracker's reference kernels write here
            # directly -- hands views of them back: nothing to copy)
            for name, t in (("vals", vals_i[l]), ("P", Pi[l]), ("dI", dI_dT[l]), ("mask", masks[l])):
                if t.data_ptr() != c[name].data_ptr():
                    c[name].copy_(t.reshape(c[name].shape))
        self.src = src                                         # held: an id cannot be recycled while it is remembered


_pyr_buffers = {}


def pyr_buffers(vals_i, Pi, dI_dT, img_j, intrinsics):
    """The persistent buffers of this pyramid shape (created on first use)."""
    key = (str(Pi[0].device), Pi[0].dtype) + tuple((v.shape[0] * v.shape[1], v.shape[2]) + tuple(i.shape[-2:])
                                                   for v, i in zip(vals_i, img_j))          # (points, channels, H, W) per level
    pb = _pyr_buffers.get(key)
    if pb is None:
        if len(_pyr_buffers) > 2:
            _pyr_buffers.clear()
            _level_graphs.clear()
        pb = _PyrBuffers(vals_i, Pi, dI_dT, img_j, intrinsics)
        _pyr_buffers[key] = pb
    return pb


def level_workspace_pair(device):
    """A private (cached, uncached) barrier workspace pair for persistent level kernels launched from a captured graph."""
    L = _lib.lib()
    ws = torch.zeros(L.como_track_level_workspace_bytes() // 4, device=device, dtype=torch.int32)
    wsp = 0
    if UNCACHED_WS:
        with torch.cuda.device(device):
            wsp = L.como_track_level_workspace_create() or 0
    return ws, wsp


def photo_tracking_levels_static(Tji_init, aff_init, pb, img_j, intrinsics, term_criteria, ws_pair, prezeroed=False):
    """Coarse -> fine through the persistent level kernels on the persistent reference buffers `pb`, every level starting from
    the previous level's result record in place; no host synchronisation, nothing but launches (capturable).  Returns
    (Tji (1,4,4), aff (1,2,1), records (levels,106): [.,104] < 0 = a device-wide barrier timed out) or None when a level does
    not fit the persistent kernel."""
    nl = len(pb.levels)
    outs = torch.empty((nl, 106), device=Tji_init.device, dtype=Tji_init.dtype)
    Tji, aff = Tji_init, aff_init
    for l, c in enumerate(pb.levels):
        # (ws_pair: one pair for all levels, or one PER level -- then `prezeroed` says the caller has cleared them all)
        wp = ws_pair[l] if isinstance(ws_pair, list) else ws_pair
        res = photo_level_tracking_fused(Tji, aff, c["vals"], c["P"], c["dI"], img_j[l], intrinsics[l], term_criteria, c["mask"],
                                         out=outs[l], ws_pair=wp, prezeroed=prezeroed and isinstance(ws_pair, list))
        if res is None:
            return None
        Tji, aff = res[0], res[1]
    return Tji, aff, outs


def level_kernel_failed():
    """A persistent level kernel reported a negative status ([104] = -1: a barrier timed out, -2: an XCD-local level did not sit on
    one XCD -- csrc/track.hip's in-launch census): the caller discards the frame's result and tracks it again on the per-iteration
    chain; the XCD-local form is switched off for the rest of the process (every later level runs device-wide).  Returns True when
    that changed the form the next launches take -- a captured frame graph holds the old form and has to be dropped."""
    photo_tracking_pyr.fallbacks = getattr(photo_tracking_pyr, "fallbacks", 0) + 1
    return _lib.lib().como_track_level_set_local(0) == 1


def photo_tracking_pyr(Tji_init, aff_init, vals_i, Pi, dI_dT, masks, intrinsics, img_j, photo_sigma, term_criteria):
    """reference photo_tracking.py:10-42 (lists ordered coarse -> fine).
    The reference gathers the masked subset of vals / P / dI_dT on every frame; here the full arrays are copied into
    persistent buffers when (and only when) the reference tensors change, the masks go to the kernel
    (`como_track_iter_masked_*`), and every level replays one iteration graph captured once per pyramid shape."""
    pb = pyr_buffers(vals_i, Pi, dI_dT, img_j, intrinsics)
    pb.load_reference(vals_i, Pi, dI_dT, masks)
    for l, c in enumerate(pb.levels):
        c["img"].copy_(img_j[l])
        c["K"].copy_(intrinsics[l])

    def run(fused):
        Tji, aff, bad = Tji_init.clone(), aff_init.clone(), None
        for c in pb.levels:
            photo_level_tracking.last_out = None
            Tji, aff = photo_level_tracking(Tji, aff, c["vals"], c["P"], c["dI"], c["img"], c["K"], photo_sigma, term_criteria,
                                            in_mask=c["mask"], fused=fused)
            rec = photo_level_tracking.last_out
            if rec is not None:                                # the persistent level kernel ran: [104] < 0 = barrier time-out
                bad = (rec[104] < 0) if bad is None else (bad | (rec[104] < 0))
        return Tji, aff, bad

    Tji, aff, bad = run(None)
    # The persistent kernel's device-wide barrier gives up (instead of hanging) when its workgroups are not co-resident --
    # another process or concurrent kernels on the GPU; the level's result is then the last completed iteration's state, not
    # the converged one.  ONE scalar per frame decides (the caller synchronises right after tracking anyway for its keyframe
    # tests): on a time-out the frame is tracked again by the per-iteration chain.
    if bad is not None and bool(bad):
        level_kernel_failed()
        Tji, aff, _ = run(False)
    return Tji, aff
