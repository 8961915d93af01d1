"""Drop-in for the reference's como/odom/frontend/photo_tracking.py on MI355X.

Same names, argument meaning and return values as the reference functions; the per-pixel work of
`tracking_iter` (warp, sampling, residual, validity mask, exact median, Huber-weighted 8x8 normal
equations, solve, pose update) runs in the HIP kernel chain of csrc/track.hip through the C ABI
(`como_track_iter_*`).  Gray images (c = 1) only; no CPU fallback.
"""
import torch

from como_amd import _lib
from como_amd.geometry.lie_algebra import skew_symmetric

_ws_cache = {}


def _workspace(device, dtype, N):
    key = (str(device), dtype, N)
    ws = _ws_cache.get(key)
    if ws is None:
        L = _lib.lib()
        ws = {
            "r": torch.empty(N, device=device, dtype=dtype),
            "hists": torch.empty(L.como_select_workspace_bytes() // 4, device=device, dtype=torch.int32),
            "partials": torch.empty(L.como_track_partials_bytes() // 8, device=device, dtype=torch.float64),
        }
        _ws_cache.clear()
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


def tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, want_proj=True):
    """Enqueue one GN iteration; returns (out[105], valid u8 (N,), pj (N,2) or None, depth (N,) or None)."""
    _lib.require_cuda(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    if img_j.shape[0] != 1 or img_j.shape[1] != 1 or Pi.shape[0] != 1:
        raise RuntimeError("como_amd tracking: batch 1, gray (c = 1) only")
    dt, dev = Pi.dtype, Pi.device
    N = Pi.shape[1]
    H, W = img_j.shape[-2:]
    for t in (Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT):
        if not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("como_amd tracking: inputs must be contiguous and share one dtype")
    ws = _workspace(dev, dt, N)
    out = torch.empty(105, device=dev, dtype=dt)
    valid = torch.empty(N, device=dev, dtype=torch.uint8)
    pj = torch.empty((N, 2), device=dev, dtype=dt) if want_proj else None
    depth = torch.empty(N, device=dev, dtype=dt) if want_proj else None
    fn = getattr(_lib.lib(), "como_track_iter_" + _lib.suffix(dt))
    rc = fn(_lib.ptr(Tji), _lib.ptr(intrinsics), _lib.ptr(aff), _lib.ptr(Pi), _lib.ptr(vals_i), _lib.ptr(img_j), H, W, N,
            _lib.ptr(dI_dT), _lib.ptr(ws["r"]), _lib.ptr(valid), _lib.ptr(pj), _lib.ptr(depth), _lib.ptr(ws["hists"]),
            _lib.ptr(ws["partials"]), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "como_track_iter")
    return out, valid, pj, depth


def tracking_iter(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, photo_sigma, A_norm):
    """reference photo_tracking.py:117-143.  Returns the same 8-tuple:
    (Tji_new (1,4,4), aff_new (1,2,1), delta (1,8,1), mean_sq_err, grad_norm, pj (1,N,2), valid_mask (1,N), depth_j (1,N,1)).
    `photo_sigma` and `A_norm` are accepted for signature parity (the reference ignores photo_sigma too; A_norm = 1/size is
    recomputed in-kernel in the same dtype)."""
    out, valid, pj, depth = tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    return (out[80:96].reshape(1, 4, 4), out[96:98].reshape(1, 2, 1), out[72:80].reshape(1, 8, 1), out[98], out[99],
            pj[None], valid[None].bool(), depth[None, :, None])


class _LevelGraph:
    """One tracking GN iteration (memset + residual + 2 select passes + reduce + finish + state write-back) captured in a
    hipGraph over FIXED buffers: the pose / affine state lives in `self.T`, `self.aff` and is advanced in place, so the
    host only replays the graph and reads 3 scalars back for the reference's stop test (~35 us per iteration instead of
    ~300 us of Python + 7 launches)."""

    def __init__(self, vals_i, Pi, dI_dT, img_j, intrinsics):
        dev, dt = Pi.device, Pi.dtype
        self.args = (Pi, intrinsics, img_j, vals_i, dI_dT)
        self.T = torch.zeros((1, 4, 4), device=dev, dtype=dt)
        self.aff = torch.zeros((1, 2, 1), device=dev, dtype=dt)
        self.out = None
        self.graph = None
        self.dev = dev

    def _iter(self):
        Pi, K, img_j, vals_i, dI_dT = self.args
        out, _, _, _ = tracking_iter_raw(self.T, Pi, K, img_j, self.aff, vals_i, dI_dT, want_proj=False)
        self.T.copy_(out[80:96].reshape(1, 4, 4))
        self.aff.copy_(out[96:98].reshape(1, 2, 1))
        return out

    def capture(self):
        try:
            keepT, keepA = self.T.clone(), self.aff.clone()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._iter()                                   # warm-up outside capture (workspace allocation)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            self.T.copy_(keepT)
            self.aff.copy_(keepA)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.out = self._iter()
            self.T.copy_(keepT)                                # the capture itself does not execute; keep the state exact
            self.aff.copy_(keepA)
            self.graph = g
        except Exception:                                      # noqa: BLE001  (eager fallback keeps working)
            self.graph = None
            try:
                torch.cuda.synchronize(self.dev)
            except Exception:                                  # noqa: BLE001
                pass
        return self.graph is not None

    def step(self):
        if self.graph is not None:
            self.graph.replay()
            return self.out
        return self._iter()


_level_graphs = {}


def _level_graph(vals_i, Pi, dI_dT, img_j, intrinsics):
    """Graphs are keyed by the buffer addresses: a new keyframe (new reference arrays) builds a new one, successive frames
    tracked against the same keyframe WITH THE SAME target-image buffer reuse it."""
    key = (vals_i.data_ptr(), Pi.data_ptr(), dI_dT.data_ptr(), img_j.data_ptr(), intrinsics.data_ptr(), Pi.shape[1],
           tuple(img_j.shape), Pi.dtype)
    lg = _level_graphs.get(key)
    if lg is None:
        if len(_level_graphs) > 16:
            _level_graphs.clear()
        lg = _LevelGraph(vals_i, Pi, dI_dT, img_j, intrinsics)
        lg.capture()
        _level_graphs[key] = lg
    return lg


def photo_level_tracking(Tji_init, aff_init, vals_i, Pi, dI_dT, img_j, intrinsics, photo_sigma, term_criteria, use_graph=True):
    """reference photo_tracking.py:147-185 (one host read-back of 3 scalars per iteration for the stop test).
    use_graph: replay each iteration from a hipGraph over fixed buffers (inputs must stay alive and unmodified)."""
    if use_graph:
        lg = _level_graph(vals_i, Pi, dI_dT, img_j, intrinsics)
    else:
        lg = _LevelGraph(vals_i, Pi, dI_dT, img_j, intrinsics)
    lg.T.copy_(Tji_init.reshape(1, 4, 4))
    lg.aff.copy_(aff_init.reshape(1, 2, 1))
    it = 0
    prev = float("inf")
    while True:
        out = lg.step()
        sc = out[98:104].tolist()                              # one small D2H copy: mse, grad_norm, ., ., ., delta_norm
        mse, gnorm, dnorm = sc[0], sc[1], sc[5]
        it += 1
        rel = abs((prev - mse) / prev) if prev != float("inf") else float("nan")
        if (it >= term_criteria["max_iter"] or dnorm < term_criteria["delta_norm"] or rel < term_criteria["rel_tol"]
                or gnorm < term_criteria["grad_norm"]):
            break
        prev = mse
    photo_level_tracking.last_iters = it
    return lg.T.clone(), lg.aff.clone()


_pyr_cache = {}


def photo_tracking_pyr(Tji_init, aff_init, vals_i, Pi, dI_dT, masks, intrinsics, img_j, photo_sigma, term_criteria):
    """reference photo_tracking.py:10-42 (lists ordered coarse -> fine).
    The masked reference arrays of a keyframe are gathered once and kept (the reference re-gathers them every frame); the
    current frame's pyramid levels and intrinsics are copied into persistent buffers so that every level replays its
    captured iteration graph."""
    # keyed by the IDENTITY of every reference-side tensor; the entry keeps them alive, so neither an id nor a device
    # address can be recycled for different data while the entry exists (a refined depth map = new P / dI_dT / mask
    # tensors = a new entry)
    src = tuple(vals_i) + tuple(Pi) + tuple(dI_dT) + tuple(masks)
    key = tuple(id(t) for t in src)
    ent = _pyr_cache.get(key)
    if ent is None:
        if len(_pyr_cache) > 4:
            _pyr_cache.clear()
            _level_graphs.clear()
        lv = []
        for l in range(len(vals_i)):
            mk = masks[l]
            lv.append({"vals": vals_i[l][None, mk, :].contiguous(), "P": Pi[l][None, mk, :].contiguous(),
                       "dI": dI_dT[l][None, mk, :, :].contiguous(), "img": torch.empty_like(img_j[l]),
                       "K": torch.empty_like(intrinsics[l])})
        ent = (lv, src)
        _pyr_cache[key] = ent
    lv = ent[0]
    Tji = Tji_init.clone()
    aff = aff_init.clone()
    for l in range(len(vals_i)):
        c = lv[l]
        c["img"].copy_(img_j[l])
        c["K"].copy_(intrinsics[l])
        Tji, aff = photo_level_tracking(Tji, aff, c["vals"], c["P"], c["dI"], c["img"], c["K"], photo_sigma, term_criteria)
    return Tji, aff
