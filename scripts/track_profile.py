"""GPU box: phase timing of the persistent tracking-level kernel (build with COMO_EXTRA_HIPCC_FLAGS=-DCOMO_TL_PROFILE).
Prints the 100 MHz wall-clock deltas between the phase stamps of iteration 3 (workgroup 0)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import tracking_leg  # noqa
import como_amd.odom.frontend.photo_tracking as pt
from como_amd import _lib
dev = torch.device("cuda:0")
r = tracking_leg(dev, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print({k: r[k] for k in ("us_per_iter", "persistent_level_kernel")})
ws = pt._level_ws[str(dev)]  # (profile stamps are read from the torch workspace: run with COMO_TRACK_UNCACHED_WS=0)
L = _lib.lib()
off = (32 * 9 * 4 + 2 * 6 * 2048 * 4 + 256 * 8) // 8
st = ws.view(torch.int64)[off:off + 17].cpu().tolist()
names = ["A compute", "flush0", "barrier1", "B resolve+hist", "flush1", "barrier2", "C resolve+hist", "flush2", "barrier3",
         "D resolve+accumulate", "block reduce", "barrier4", "partial sums", "8x8 solve", "exp+update", "state bcast"]
for i, nm in enumerate(names):
    print(f"{nm:24s} {(st[i + 1] - st[i]) * 0.01:8.2f} us")
print("iteration total", (st[16] - st[0]) * 0.01, "us")
