"""Small host-built index lists -> device tensors without a pageable, synchronising copy each.

The keyframe path builds a dozen short index lists on the host (kept correspondences, landmark bookkeeping, pair tables); as
`torch.tensor(list, device=...)` each is a pageable host->device copy that blocks the host for ~20-30 us.  `to_device` stages
through a ring of pinned blocks and issues an asynchronous copy: the block is reused only after the event recorded behind its
last copy has completed (round robin over RING blocks, so in practice never waited for)."""
import numpy as np
import torch

RING = 8
_ring = {"blocks": [None] * RING, "events": [None] * RING, "next": 0}


def _block(nbytes):
    k = _ring["next"]
    _ring["next"] = (k + 1) % RING
    blk, ev = _ring["blocks"][k], _ring["events"][k]
    if ev is not None:
        ev.synchronize()
    if blk is None or blk.numel() < nbytes:
        blk = _ring["blocks"][k] = torch.empty(max(1 << 16, 2 * nbytes), dtype=torch.uint8).pin_memory()
    return k, blk


def to_device(values, dtype, device):
    """values: a list / numpy array; dtype: a torch dtype (bool, int32, int64, float32, float64); returns a device tensor of the
    values' shape.  CPU devices get a plain tensor."""
    np_dt = {torch.bool: np.bool_, torch.uint8: np.uint8, torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32,
             torch.float64: np.float64}[dtype]
    arr = np.ascontiguousarray(np.asarray(values, dtype=np_dt))
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.from_numpy(arr.copy()).to(dev)
    nb = arr.nbytes
    out = torch.empty(arr.shape, dtype=dtype, device=dev)
    if nb == 0:
        return out
    k, blk = _block(nb)
    blk.numpy()[:nb] = arr.reshape(-1).view(np.uint8)
    out.view(torch.uint8).reshape(-1).copy_(blk[:nb], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _ring["events"][k] = ev
    return out


class ReadLater:
    """A small device tensor on its way to the host without a synchronisation: `read_later(t)` issues an asynchronous copy into a
    pinned block and records an event behind it; `ready()` polls the event, `value()` waits for it (no wait in practice when the
    caller has synchronised with the stream since) and returns the values as a CPU tensor."""

    def __init__(self, host, event):
        self.host, self.event = host, event

    def ready(self):
        return self.event.query()

    def value(self):
        self.event.synchronize()
        return self.host


_rl = {"blocks": [None] * 16, "next": 0}


def read_later(t):
    if not t.is_cuda:
        return ReadLater(t.detach().clone(), _Done())
    n = t.numel()
    k = _rl["next"]
    _rl["next"] = (k + 1) % len(_rl["blocks"])
    blk = _rl["blocks"][k]
    if blk is None or blk[0].dtype != t.dtype or blk[0].numel() < n:
        blk = _rl["blocks"][k] = (torch.empty(max(64, n), dtype=t.dtype).pin_memory(), None)
    host = blk[0][:n].view(t.shape)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    return ReadLater(host, ev)


class _Done:
    def query(self):
        return True

    def synchronize(self):
        pass
