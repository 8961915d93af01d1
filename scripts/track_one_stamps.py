"""GPU box: phase timing of the ONE-workgroup tracking level kernel (csrc/track.hip track_level_one_kernel; library built with
-DCOMO_TL_PROFILE, e.g. como_amd/lib_prof): 100 MHz wall-clock deltas between the phase stamps of iteration 3, taken by thread 0 (a
pixel wave) and by lane 0 of the solver wave.
    COMO_HIP_LIB=$PWD/como_amd/lib_prof/libcomo_hip.so python scripts/track_one_stamps.py [H W] [threads]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import como_amd.odom.frontend.photo_tracking as pt
from como_amd import _lib
from tests.test_gpu_r2 import _tracking_level_inputs

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 80)
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
tp, K, P, vals, J = _tracking_level_inputs(H, W, 3)
aff = torch.zeros((1, 2, 1), device=dev)
term = {"max_iter": 8, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}
L = _lib.lib()
L.como_track_level_set_one(nt)
ws = torch.zeros(L.como_track_level_workspace_bytes() // 4, device=dev, dtype=torch.int32)
args = (tp["Tji_init"].reshape(1, 4, 4).contiguous(), aff, vals, P, J, tp["img_cur"], K, term, None)
for _ in range(3):
    res = pt.photo_level_tracking_fused(*args, ws_pair=(ws, 0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    res = pt.photo_level_tracking_fused(*args, ws_pair=(ws, 0))
e1.record()
torch.cuda.synchronize()
print(f"{H}x{W}, {nt} threads: {e0.elapsed_time(e1) * 1e3 / 20 / 8:.2f} us per iteration incl. launch (8 iterations per launch), status {int(res[2][104])}")
words = 32 * 10 + 2 * 6 * 2048
off = (words * 4 + 2 * 2 * 3 * 64 * 8) // 8
st = ws.view(torch.int64)[off:off + 32].cpu().tolist()
segs = [("clear histograms, P = K T, exp(-a)", 0, 1), ("warp / residual / first digit's histogram", 1, 2), ("wait for the other waves", 2, 3),
        ("resolve digit 0", 3, 4), ("(second digit's histogram on a miss) resolve digit 1", 4, 5), ("last digit's histogram + barrier", 5, 6),
        ("resolve digit 2", 6, 7), ("sums 0..23 + wave tree", 7, 8), ("sums 24..44 + wave tree", 8, 9), ("barrier (sums in LDS)", 9, 10),
        ("barrier (solver: totals)", 10, 11), ("barrier (solver: 8x8 solve, exp, record)", 11, 12), ("record out, stop test", 12, 13)]
for nm, a, b in segs:
    print(f"{nm:64s} {(st[b] - st[a]) * 0.01:8.2f} us")
print("iteration total", (st[13] - st[0]) * 0.01, "us")
s = st[16:]
print("solver wave: totals", (s[11] - s[10]) * 0.01, "| solve", (s[14] - s[11]) * 0.01, "| state + record", (s[15] - s[14]) * 0.01)
