"""Time of the second sampling pass of a keyframe insertion at a 640x480 domain (34 current points -> 64): persistent loop vs the
launch-per-step loop.  python scripts/sampler_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from como_amd.depth_cov.core import samplers

g = torch.Generator().manual_seed(11)
h, w = 480, 640
x = torch.rand((1, 1, h, w), generator=g) * 2e-3 + 2e-4
z = torch.rand((1, 1, h, w), generator=g) * 2e-3 + 2e-4
o = (torch.rand((1, 1, h, w), generator=g) - 0.5) * 2e-4
cov = torch.cat((x, o, o, z), dim=1).cuda()
_, inds0 = samplers.sample_sparse_coords(cov, 34, "greedy_conditional_entropy", border=3, dist_thresh=0.05, signal_var=1.0, fixed_var=0.0)
dom = samplers.get_coords_domain(cov, border=3)
curr = dom[:, inds0[0]].double()
_orig = samplers.greedy_loop
_ev = []


def _timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _orig(*a, **k)
    e1.record()
    _ev.append((e0, e1))
    return r


samplers.greedy_loop = _timed
for flag in (True, False, True, False):
    samplers.PERSISTENT_LOOP = flag
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        c, inds = samplers.sample_sparse_coords(cov, 64, "greedy_conditional_entropy", border=3, dist_thresh=0.05, signal_var=1.0,
                                                fixed_var=0.0, curr_coords=curr)
        torch.cuda.synchronize()
        dt = time.time() - t
    print("persistent" if flag else "per-step  ", "%.3f ms whole pass, loop alone %.3f ms" % (dt * 1e3, _ev[-1][0].elapsed_time(_ev[-1][1])),
          inds[0, :4].tolist())
samplers.check_pending_info(wait=True)
