"""Small dense SPD systems on the GPU through csrc/smallsolve.hip (one launch per batch, the matrix resident in LDS):
the conditioning steps the reference does with torch.linalg on m x m (m <= 64) systems -- como/utils/lin_alg.py:82-87
(`lstsq_chol`), Mapping.py:450-458, depth_cov/core/distill_depth.py:30-48, depth_cov/core/samplers.py:110-165,
odom/frontend/two_frame_sfm.py:115-125, 288-293.  No CPU path and no library fallback: CUDA float32 / float64 tensors only."""
import torch

from como_amd import _lib

MAX_N = 80


def chol_small(A, want_L=True, want_inv=False, rhs=None, want_info=False):
    """A (B,n,n) or (n,n) SPD (lower triangle read).  Returns a dict with the requested pieces:
    "L" (B,n,n) lower Cholesky factor, "inv" (B,n,n) = A^-1, "X" (B,n,k) = A^-1 rhs, "info" (B,) int32 (cholesky_ex semantics)."""
    _lib.require_cuda(A)
    squeeze = A.dim() == 2
    A3 = (A[None] if squeeze else A).contiguous()
    B, n, n2 = A3.shape
    if n != n2 or n > MAX_N or A3.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"como_amd chol_small: needs (B,n,n) float32/float64 with n <= {MAX_N}, got {tuple(A.shape)} {A.dtype}")
    dev, dt = A3.device, A3.dtype
    L = torch.empty_like(A3) if want_L else None
    inv = torch.empty_like(A3) if want_inv else None
    X = r3 = None
    k = 0
    if rhs is not None:
        _lib.require_cuda(rhs)
        r3 = (rhs[None] if squeeze else rhs).to(dt).contiguous()
        if r3.dim() != 3 or r3.shape[0] != B or r3.shape[1] != n:
            raise RuntimeError("como_amd chol_small: rhs must be (B,n,k)")
        k = r3.shape[2]
        X = torch.empty_like(r3)
    # (both kernels write info[b] of every system: no zero-fill launch; an empty system reports 0)
    info = torch.empty(B, dtype=torch.int32, device=dev) if (n > 0 and B > 0) else torch.zeros(B, dtype=torch.int32, device=dev)
    if n > 0 and B > 0:                                   # (an empty system -- a keyframe that tracked no point of the previous one --
        fn = getattr(_lib.lib(), "como_chol_small_" + _lib.suffix(dt))      # has empty factors, as torch.linalg.cholesky returns)
        rc = fn(A3.data_ptr(), B, n, _lib.ptr(L), _lib.ptr(inv), _lib.ptr(r3), k, _lib.ptr(X), info.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(rc, "como_chol_small")
    out = {}
    un = (lambda t: t[0]) if squeeze else (lambda t: t)
    if want_L:
        out["L"] = un(L)
    if want_inv:
        out["inv"] = un(inv)
    if X is not None:
        out["X"] = un(X)
    if want_info:
        out["info"] = info
    return out


def trsm_lower(L, Bm, trans=False, out=None):
    """X = L^-1 Bm (trans: L^-T Bm), L (B,n,n) lower triangular (n <= 64), Bm (B,n,d) with many columns
    (torch.linalg.solve_triangular(L, Bm, upper=False); trans: the second half of torch.cholesky_solve)."""
    _lib.require_cuda(L, Bm)
    L3, B3 = L.contiguous(), Bm.to(L.dtype).contiguous()
    if L3.dim() != 3 or B3.dim() != 3 or L3.shape[0] != B3.shape[0] or L3.shape[1] != L3.shape[2] or L3.shape[1] != B3.shape[1]:
        raise RuntimeError("como_amd trsm_lower: L (B,n,n), Bm (B,n,d)")
    if L3.shape[1] > 64 or L3.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("como_amd trsm_lower: n <= 64, float32 / float64")
    # out: a contiguous (B,n,d) tensor of L's type that receives X (e.g. the leading rows of a larger buffer, B = 1)
    if out is not None and (out.shape != B3.shape or out.dtype != B3.dtype or not out.is_contiguous() or out.device != B3.device):
        raise RuntimeError("como_amd trsm_lower: `out` must be a contiguous tensor of the shape and type of the right-hand side")
    X = out if out is not None else torch.empty_like(B3)
    fn = getattr(_lib.lib(), "como_trsm_lower_" + _lib.suffix(L3.dtype))
    rc = fn(L3.data_ptr(), B3.data_ptr(), X.data_ptr(), L3.shape[0], L3.shape[1], B3.shape[2], 1 if trans else 0,
            _lib.stream_ptr(L3.device))
    _lib.check(rc, "como_trsm_lower")
    return X


def cholesky_solve_many(Bm, L):
    """torch.cholesky_solve(Bm, L, upper=False) for Bm (B,n,d) with many columns: L^-T (L^-1 Bm), two launches."""
    return trsm_lower(L, trsm_lower(L, Bm), trans=True)
