"""Exact medians on the device through the radix select of csrc/select.hip (`como_select_*`): torch.median semantics (the LOWER
median) for non-negative data such as depths, without the sort, the boolean-mask gather or the host synchronisation of
`torch.median(x[mask])` (the reference: como/odom/Tracking.py:345, Mapping.py:749-758, frontend/TwoFrameSfm.py:92)."""
import torch

from como_amd import _lib

_ws = {}


def median_workspace(device, nseg=1):
    """The select workspace `masked_median` uses for `nseg` segments on `device` (a caller that clears it itself -- the tracker's
    frame graph, inside its first launch -- passes prezeroed=True)."""
    key = (str(device), nseg)
    h = _ws.get(key)
    if h is None:
        h = _ws[key] = torch.empty(nseg * _lib.lib().como_select_workspace_bytes() // 4, dtype=torch.int32, device=device)
    return h


def masked_median(x, valid=None, prezeroed=False):
    """x (nseg, n) or (n,) float32 / float64 CUDA tensor of NON-NEGATIVE values; valid: same shape, bool / uint8, None = all.
    Returns (nseg,) [or a 0-dim tensor for 1-D input]: per segment the lower median of the valid entries (the select orders by
    |x|), NaN-free as long as the valid entries are.  A segment WITHOUT valid entries returns NaN (csrc/select.hip
    select_finish_kernel) -- callers that can meet an empty mask must test the count (Tracking.decide_frame does)."""
    _lib.require_cuda(x)
    one = x.dim() == 1
    x2 = (x[None] if one else x.reshape(x.shape[0], -1)).contiguous()
    if x2.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("como_amd masked_median: float32 / float64 only")
    nseg, n = x2.shape
    v = None
    if valid is not None:
        v = (valid[None] if one else valid.reshape(nseg, -1)).to(torch.uint8).contiguous()
    L = _lib.lib()
    dev, dt = x2.device, x2.dtype
    key = (str(dev), nseg)
    h = _ws.get(key)
    if h is None:
        h = _ws[key] = torch.empty(nseg * L.como_select_workspace_bytes() // 4, dtype=torch.int32, device=dev)
    s = _lib.stream_ptr(dev)
    sfx = _lib.suffix(dt)
    if not prezeroed:
        _lib.check(L.como_select_begin(h.data_ptr(), nseg, s), "como_select_begin")
    for p in range(3 if dt == torch.float32 else 6):
        _lib.check(getattr(L, "como_select_hist_" + sfx)(x2.data_ptr(), _lib.ptr(v), n, nseg, h.data_ptr(), p, s), "como_select_hist")
    out3 = torch.empty((nseg, 3), dtype=dt, device=dev)
    _lib.check(getattr(L, "como_select_finish_" + sfx)(h.data_ptr(), nseg, out3.data_ptr(), s), "como_select_finish")
    return out3[0, 0] if one else out3[:, 0]
