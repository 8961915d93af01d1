"""Normal-equation helpers, mirroring the reference's como/odom/backend/linear_system.py.

solve_system / update_vars keep the reference signatures (:101-152).  The dense float64 Cholesky (D = 8 B + 8 R + 3 L ~ 760..2700)
runs on the hand-written HIP solvers: csrc/cholp.hip -- the whole solve in ONE persistent launch, up to 34 column pairs
(D <= 2175) -- and csrc/chol.hip -- the multi-launch panel solver beyond that, or everywhere after
`como_chol_set_persistent(0)` / COMO_CHOLP=0.  `info` from the factorisation is kept in `solve_system.last_info` instead of
being swallowed (-1 = the persistent solver timed out); WindowBA guards its update with it and `WindowBA.check_solver` /
`Mapping._check_solver` act on it.
"""
import torch

import como_amd.geometry.lie_algebra as lie


def row_col_to_lin_index(row, col, width):
    return row * width + col


def accumulate_gradient_scatter(g_batch, grad, grad_inds):
    grad.index_add_(0, grad_inds, g_batch)


def accumulate_hessian_scatter(H_batch, H, H_inds):
    H.view(-1).index_add_(0, H_inds, H_batch)


def get_gradient(J, r):
    return -torch.sum(J * r[..., None], dim=(1, 2))


def get_hessian_diag_block(J):
    return torch.einsum("bnck,bncl->bkl", J, J)


def get_hessian_off_diag_block(J1, J2):
    return torch.einsum("bnck,bncl->bkl", J1, J2)


def landmark_to_batched_3d_point_inds(landmark_inds, num_kf):
    """(nnz,2) [kf, landmark] pairs (row-major nonzero order) -> (num_kf, 3m) indices 3*l + {0,1,2} (:78-89)."""
    lm = landmark_inds[:, 1].view(num_kf, -1)
    p = 3 * lm.repeat_interleave(3, dim=1)
    return p + torch.arange(3, device=lm.device).repeat(lm.shape[1])[None, :]


_chol_ws = {}
_CHOL_WS_MAX = 8          # process-wide store (reference-signature calls): at most this many system sizes keep a workspace


def _chol_entry(D, device, store, shared):
    """(workspace, info, delta) for a D x D float64 system in `store`.
    shared = True (a caller-owned store: one window, one stream): ONE grow-only workspace buffer serves every system size the
    store has seen -- the window is rebuilt with a new D on almost every keyframe, and a buffer per size (37 MB at D = 1240) kept by
    the inherited store would pile up over a long sequence; only the current size keeps its (info, delta) pair.
    shared = False (the process-wide store, which several streams may use with different sizes at once): a buffer per size, the
    oldest sizes evicted beyond _CHOL_WS_MAX."""
    from como_amd import _lib
    key = (str(device), D)
    w = store.get(key)
    if w is not None:
        return w
    need = _lib.lib().como_chol_workspace_bytes(D) // 8
    if shared:
        buf = store.get("buffer")
        if buf is None or buf.numel() < need or buf.device != torch.device(device):
            buf = store["buffer"] = torch.empty(need, dtype=torch.float64, device=device)
        for k in [k for k in store if isinstance(k, tuple)]:
            del store[k]
    else:
        buf = torch.empty(need, dtype=torch.float64, device=device)
        for k in [k for k in store if isinstance(k, tuple)][:max(0, len(store) + 1 - _CHOL_WS_MAX)]:
            del store[k]
    w = store[key] = (buf, torch.zeros(1, dtype=torch.int32, device=device), torch.empty((D, 1), dtype=torch.float64, device=device))
    return w


def solve_system(H, g, ws=None):
    """ws: caller-owned dict for the factorisation workspace (a captured graph records its address; None: process-wide).
    delta (D,1) = H^-1 g by dense Cholesky (reference :101-112).  float64 systems on the GPU run the HIP factorisation of
    csrc/cholp.hip (one persistent launch, D <= 2175) / csrc/chol.hip (multi-launch, larger systems; both graph-capturable);
    float32 systems (the reference-signature path with a float32 H) go through torch.linalg.  `solve_system.last_info` holds
    the factorisation status (device int: 0, the 1-based pivot that was not positive, or -1 = persistent solver timed out)."""
    if H.is_cuda and H.dtype == torch.float64:
        from como_amd import _lib
        L = _lib.lib()
        D = H.shape[0]
        own = ws is not None
        w = _chol_entry(D, H.device, ws if own else _chol_ws, shared=own)
        delta = w[2] if own else torch.empty((D, 1), dtype=torch.float64, device=H.device)
        Hc = H if H.is_contiguous() else H.contiguous()
        rc = L.como_chol_solve_f64(Hc.data_ptr(), g.contiguous().data_ptr(), delta.data_ptr(), w[0].data_ptr(), D,
                                   w[1].data_ptr(), _lib.stream_ptr(H.device))
        _lib.check(rc, "como_chol_solve")
        solve_system.last_info = w[1]
        return delta
    Lf, info = torch.linalg.cholesky_ex(H, upper=False, check_errors=False)
    solve_system.last_info = info
    return torch.cholesky_solve(g[:, None], Lf, upper=False)


def chol_workspace(D, device, ws):
    """(workspace, info, delta) of the HIP factorisation for a D x D float64 system, kept in the caller-owned dict `ws`
    (a captured graph records the addresses)."""
    return _chol_entry(D, device, ws, shared=True)


def solve_packed(D, device, ws):
    """delta = H^-1 g for a system `como_sys_finalize_pack` already packed into the solver's working copy (the fused window
    chain: the packing launch of solve_system is folded into the fixed-point -> float64 conversion)."""
    from como_amd import _lib
    w = chol_workspace(D, device, ws)
    rc = _lib.lib().como_chol_solve_packed_f64(w[2].data_ptr(), w[0].data_ptr(), D, w[1].data_ptr(), _lib.stream_ptr(device))
    _lib.check(rc, "como_chol_solve_packed")
    solve_system.last_info = w[1]
    return w[2]


def update_vars(delta, kf_poses, kf_aff_params, kf_inds, recent_poses, recent_aff_params, recent_inds, P,
                landmark_ind_start):
    d = delta.squeeze(-1)
    kd = d[kf_inds]
    kf_poses_new = lie.batch_se3(kf_poses, kd[:, :6])
    kf_aff_new = kf_aff_params + kd[:, 6:, None]
    if recent_inds.shape[0] > 0:
        rd = d[recent_inds]
        rp = lie.batch_se3(recent_poses, rd[:, :6])
        ra = recent_aff_params + rd[:, 6:, None]
    else:
        rp = torch.empty((0), device=d.device, dtype=d.dtype)
        ra = torch.empty((0), device=d.device, dtype=d.dtype)
    return kf_poses_new, kf_aff_new, rp, ra, P + d[landmark_ind_start:].view(-1, 3)
