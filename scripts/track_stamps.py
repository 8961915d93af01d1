"""GPU box: phase timing of the persistent tracking-level kernel at one pyramid level (library built with
COMO_EXTRA_HIPCC_FLAGS=-DCOMO_TL_PROFILE): the 100 MHz wall-clock deltas between the phase stamps of iteration 3 of workgroup 0.
    python scripts/track_stamps.py H W"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import como_amd.odom.frontend.photo_tracking as pt
from como_amd import _lib
from tests.test_gpu_r2 import _tracking_level_inputs

H, W = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
tp, K, P, vals, J = _tracking_level_inputs(H, W, 3)
aff = torch.zeros((1, 2, 1), device=dev)
term = {"max_iter": 8, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}
L = _lib.lib()
ws = torch.zeros(L.como_track_level_workspace_bytes() // 4, device=dev, dtype=torch.int32)
for _ in range(3):
    res = pt.photo_level_tracking_fused(tp["Tji_init"].reshape(1, 4, 4).contiguous(), aff, vals, P, J, tp["img_cur"], K, term, None,
                                        ws_pair=(ws, 0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    res = pt.photo_level_tracking_fused(tp["Tji_init"].reshape(1, 4, 4).contiguous(), aff, vals, P, J, tp["img_cur"], K, term, None,
                                        ws_pair=(ws, 0))
e1.record()
torch.cuda.synchronize()
print(f"{H}x{W}: {e0.elapsed_time(e1) * 1e3 / 20 / 8:.2f} us per iteration (8 iterations per launch), local form = {L.como_track_level_local_state()}, "
      f"status {int(res[2][104])}")
words = 32 * 10 + 2 * 6 * 2048                      # TL_BAR_WORDS + the two parities' digit histograms (uint32 words)
off = (words * 4 + 2 * 2 * 3 * 64 * 8) // 8         # + TL_SUM_WORDS 64-bit sums (two parities x {integer, fraction} x three planes x 64)
st = ws.view(torch.int64)[off:off + 17].cpu().tolist()
# stamps of iteration 3 (round 6 layout; 4-6 are only written when the first-digit speculation misses)
segs = [("A warp / residual / digit-0 histogram", 0, 1), ("flush 0 (+ speculated digit 1)", 1, 2), ("barrier 1", 2, 3),
        ("clear other parity, resolve digits 0-1, digit-2 histogram + split sums", 3, 7), ("wave-tree reduce + shares", 7, 8),
        ("flush 2", 8, 9), ("barrier 2 + resolve digit 2", 9, 10), ("exact: sums+shares | split: list staged", 10, 11),
        ("exact: barrier 3 | split: totals + listed pixels", 11, 12), ("sync", 12, 13), ("8x8 solve + exp", 13, 14), ("state", 14, 15),
        ("record + broadcast", 15, 16)]
for nm, a, b in segs:
    print(f"{nm:72s} {(st[b] - st[a]) * 0.01:8.2f} us")
print("iteration total", (st[16] - st[0]) * 0.01, "us;  speculation-miss stamps (4..6):", [(st[k] - st[3]) * 0.01 if st[k] else None for k in (4, 5, 6)])
