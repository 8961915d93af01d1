"""One-process-per-GPU data parallelism for the window BA (torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference has no distributed code at all; this is the one strategy the path needs:
every rank holds the whole window (images 29 MB, K~ 630 MB at 8 x 640x480 -- nothing next to 288 GB of HBM) and
linearises its own contiguous share of the reference pixels of EVERY keyframe pair (perfect balance, unlike sharding
14 pairs over 8 ranks).  Two kinds of exchange per GN iteration:
  * the robust scale is a GLOBAL exact median (photo.py:124-128): the 2048-bin histogram of each radix-select digit
    pass is summed across ranks (8 KiB all-reduce, 3 for float keys) -- every rank then resolves the same k-th key, bit
    for bit; double keys: three histogram all-reduces (33 bits) + ONE all-gather of the handful of keys that still match
    (como_select_cand_*: every rank finishes digits 3..5 from the union), not six all-reduces;
  * the normal equations: the shards' per-pair Gram sums (b x 3936 values as fixed-point integer pairs: 0.9 MB at 14
    pairs, 3.9 MB at 62) in ONE integer all-reduce(sum) -- exact, so every rank continues with identical bits -- after
    which every rank expands them into H, adds the priors and solves redundantly (no broadcast of delta);
  * the per-keyframe median depth of the dense reference / the full depth image (Mapping.store_vars): each rank evaluates its
    pixel / row range, the digit histograms of all keyframes are all-reduced per pass (3 x 64 KiB for float keys).
Payloads are latency-bound on xGMI (tens of microseconds); see DESIGN.md for the accounting.
"""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver stack

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class Shard:
    def __init__(self, rank, world, group=None, force_collectives=False):
        """force_collectives: issue the collectives even with world == 1 (tests: a single-rank `nccl` group exercises the
        RCCL path -- dtypes, stream semantics, graph capture -- on a one-GPU box)."""
        self.rank, self.world, self.group = rank, world, group
        self.force = force_collectives
        self.n_collectives = 0            # data-path collectives issued so far (all_reduce_sum / _max / all_gather)
        self.timing = None                # a list: every data-path collective is bracketed by two events appended to it

    def pixel_range(self, n):
        per = (n + self.world - 1) // self.world
        per = ((per + 63) // 64) * 64                      # whole 64-pixel wave tiles
        b = min(n, self.rank * per)
        e = min(n, b + per)
        return b, max(b, e)                                # b == e: an idle rank (more ranks than 64-pixel tiles) owns nothing

    def row_range(self, rows):
        """This rank's share of `rows` items (any granularity), contiguous, covering [0, rows) exactly once over the ranks."""
        per = (rows + self.world - 1) // self.world
        b = min(rows, self.rank * per)
        return b, min(rows, b + per)

    @property
    def collectives(self):
        return self.world > 1 or self.force

    def capturable(self):
        """Can the collectives go into a hipGraph?  RCCL ("nccl") supports stream capture; gloo stages through the host."""
        try:
            return dist.is_initialized() and "nccl" in str(dist.get_backend(self.group)).lower()
        except Exception:   # noqa: BLE001
            return False

    def _bracket(self, fn):
        """Run one data-path collective; count it, and in timing mode bracket it with events on the current stream."""
        self.n_collectives += 1
        if self.timing is None or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.timing.append((e0, e1))
        return out

    def all_reduce_sum(self, t):
        if self.world > 1 or self.force:
            self._bracket(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
        return t

    def all_reduce_max(self, t):
        if self.world > 1 or self.force:
            self._bracket(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group))
        return t

    def all_gather(self, out, inp):
        """out (world, *inp.shape) <- every rank's inp, in rank order (the candidate exchange of the float64 select)."""
        if self.world > 1 or self.force:
            if "nccl" in str(dist.get_backend(self.group)).lower():
                self._bracket(lambda: dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group))
            else:                                           # gloo (test rigs): per-rank views of the same buffer
                self._bracket(lambda: dist.all_gather([out[r] for r in range(self.world)], inp, group=self.group))
        else:
            out[0].copy_(inp)
        return out

    def max_scalar(self, v, device):
        t = torch.tensor([v], dtype=torch.float64, device=device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def gather_scalars(self, v, device):
        """The value of every rank, in rank order (reporting only: e.g. per-replica throughput)."""
        t = torch.zeros(self.world, dtype=torch.float64, device=device)
        t[self.rank] = float(v)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [float(x) for x in t.tolist()]

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


def init_from_env(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and return (Shard, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if os.environ.get("COMO_SINGLE_DEVICE") == "1":       # test rigs with one GPU: every rank on device 0 (needs gloo)
        local = 0
    backend = backend or os.environ.get("COMO_DIST_BACKEND") or None
    device = torch.device(f"cuda:{local}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return Shard(rank, world), device


def device_identity(device):
    """What tells two GPUs apart: ordinal, PCI bus id and uuid as far as this torch build exposes them."""
    out = {"ordinal": int(device.index or 0) if device.type == "cuda" else -1}
    if device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        out["name"] = p.name
        for k in ("pci_bus_id", "pci_device_id", "pci_domain_id", "uuid"):
            if hasattr(p, k):
                out[k] = str(getattr(p, k))
    return out


def dist_record(shard, device, wb=None, graph_captured=None, eager_iters=5):
    """Self-verifying record of a (multi-)GPU run, gathered over the process group so that rank 0 can print it with the bench
    line: backend, world, every rank's device identity (N distinct GPUs?), the ranks that answered the all-gather, the data-path
    collectives one GN iteration issues (counted on the `Shard` calls, as tests/test_gpu_dist.py does), whether the iteration
    was captured into a hipGraph on every rank, and -- from HIP events around `eager_iters` eager iterations of `wb` -- where a
    rank's iteration goes: sharded per-pixel work + set-up (before the packed system), the collectives, the replicated tail
    (fixed-point -> float64 + packing, Cholesky solve, update).  Collective: every rank must call it."""
    me = {"rank": shard.rank, "device": device_identity(device), "graph_captured": None if graph_captured is None else bool(graph_captured)}
    if wb is not None and device.type == "cuda":
        n0 = shard.n_collectives
        wb.iterate()                                                 # (eager; also settles lazily created buffers)
        me["collectives_per_iteration"] = shard.n_collectives - n0
        tot = sh = co = rep = 0.0
        for _ in range(eager_iters):
            shard.timing = []
            marks = wb.timed_iterate()
            torch.cuda.synchronize(device)
            c = sum(a.elapsed_time(b) for a, b in shard.timing)
            t_all = marks["start"].elapsed_time(marks["end"])
            # ("packed" is recorded by the fused chain's finalize-and-pack launch; a window on the reference-signature path has no
            # such point: its tail is reported as 0 rather than raising inside a routine every rank must finish)
            t_tail = marks["packed"].elapsed_time(marks["end"]) if getattr(wb, "_packed", False) else 0.0
            tot, co, rep, sh = tot + t_all, co + c, rep + t_tail, sh + (t_all - t_tail - c)
        shard.timing = None
        k = 1e3 / eager_iters
        me["eager_us_per_iteration"] = {"total": tot * k, "sharded_and_setup": sh * k, "collectives": co * k, "replicated_tail": rep * k}
    backend = "none"
    if dist.is_initialized():
        backend = str(dist.get_backend(shard.group))
    ranks = [me]
    if shard.world > 1 and dist.is_initialized():
        ranks = [None] * shard.world
        dist.all_gather_object(ranks, me, group=shard.group)
    ids = {(r["device"].get("uuid") or r["device"].get("pci_bus_id") or r["device"]["ordinal"]) for r in ranks if r}
    return {"backend": backend, "world": shard.world, "ranks_answered": sorted(r["rank"] for r in ranks if r),
            "distinct_devices": len(ids), "graph_captured_all": all(bool(r["graph_captured"]) for r in ranks if r),
            "collectives_per_iteration": sorted({r.get("collectives_per_iteration") for r in ranks if r and "collectives_per_iteration" in r}),
            "per_rank": ranks}
