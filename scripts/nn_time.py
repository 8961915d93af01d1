"""DepthCov network forward (192x256, seeded weights): wall time per forward (eager launches and one captured graph) and, with
--layers, the per-kernel breakdown from torch's profiler.  python scripts/nn_time.py [--layers]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from como_amd import synth
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule

DEV = "cuda:0"
model = DepthCovModule(synth.depthcov_state_dict(0, device=DEV))
x = torch.rand(1, 3, 192, 256, device=DEV)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model(x)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        y = model(x)
    g.replay()
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    st.synchronize()
print(f"graph: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")
# the form Mapping.run_model uses: only the finest covariance head (the level the odometry path reads)
with torch.cuda.stream(st):
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        y2 = model(x, finest_only=True)
    g2.replay()
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g2.replay()
    st.synchronize()
print(f"graph, finest head only (run_model): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward; "
      f"equal to the full forward's finest level: {bool(torch.equal(y[-1], y2[-1]))}")
if "--layers" in sys.argv:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
    rows = [(e.key, e.count, e.device_time_total / max(e.count, 1)) for e in prof.key_averages() if e.device_time_total > 0]
    tot = sum(c * t for _, c, t in rows) / 5
    for k, c, t in sorted(rows, key=lambda r: -r[1] * r[2])[:16]:
        print(f"{c // 5:4d} x {t:8.1f} us  {k[:100]}")
    print(f"sum of kernel time per forward: {tot / 1e3:.3f} ms")
