"""Normal-equation helpers, mirroring the reference's como/odom/backend/linear_system.py.

solve_system / update_vars keep the reference signatures (:101-152).  The dense Cholesky runs on the
device through torch.linalg (rocSOLVER) -- D is 8 B + 8 R + 3 L ~ 760..2400; `info` from the
factorisation is kept in `solve_system.last_info` instead of being swallowed.
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


def solve_system(H, g, ws=None):
    """ws: caller-owned dict for the factorisation workspace (a captured graph records its address; None: process-wide).
    delta (D,1) = H^-1 g by dense Cholesky (reference :101-112).  float64 systems on the GPU run the blocked HIP
    factorisation of csrc/chol.hip (graph-capturable, ~10x hipSOLVER at D = 760); float32 systems (the reference-signature
    path with a float32 H) go through torch.linalg.  `solve_system.last_info` holds the factorisation status (device int)."""
    if H.is_cuda and H.dtype == torch.float64:
        from como_amd import _lib
        L = _lib.lib()
        D = H.shape[0]
        key = (str(H.device), D)
        store = _chol_ws if ws is None else ws
        ws = store.get(key)
        if ws is None:
            ws = (torch.empty(L.como_chol_workspace_bytes(D) // 8, dtype=torch.float64, device=H.device),
                  torch.zeros(1, dtype=torch.int32, device=H.device),
                  torch.empty((D, 1), dtype=torch.float64, device=H.device))
            store[key] = ws
        delta = ws[2] if store is not _chol_ws else torch.empty((D, 1), dtype=torch.float64, device=H.device)
        Hc = H if H.is_contiguous() else H.contiguous()
        rc = L.como_chol_solve_f64(Hc.data_ptr(), g.contiguous().data_ptr(), delta.data_ptr(), ws[0].data_ptr(), D,
                                   ws[1].data_ptr(), _lib.stream_ptr(H.device))
        _lib.check(rc, "como_chol_solve")
        solve_system.last_info = ws[1]
        return delta
    Lf, info = torch.linalg.cholesky_ex(H, upper=False, check_errors=False)
    solve_system.last_info = info
    return torch.cholesky_solve(g[:, None], Lf, upper=False)


def chol_workspace(D, device, ws):
    """(workspace, info, delta) of the blocked HIP factorisation for a D x D float64 system, kept in the caller-owned dict `ws`
    (a captured graph records the addresses)."""
    from como_amd import _lib
    key = (str(device), D)
    w = ws.get(key)
    if w is None:
        w = (torch.empty(_lib.lib().como_chol_workspace_bytes(D) // 8, dtype=torch.float64, device=device),
             torch.zeros(1, dtype=torch.int32, device=device), torch.empty((D, 1), dtype=torch.float64, device=device))
        ws[key] = w
    return w


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
