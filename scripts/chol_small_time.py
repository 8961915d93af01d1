"""Time como_chol_small_f64 (K_mm + 1e-6 I -> L_mm, K_mm^-1; the distillation solve) at m = 64: graph replay of `reps` calls.
COMO_CHOL_SMALL_FAST=0 selects the round-3 pivot-by-pivot LDS kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from como_amd.utils.lin_alg import chol_small


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    for n, mode in ((64, "L+inv"), (64, "solve"), (32, "L+inv"), (48, "L+inv+solve")):
        M = torch.randn((1, n, n + 8), generator=g, dtype=torch.float64)
        A = (M @ M.mT / (n + 8) + 0.05 * torch.eye(n, dtype=torch.float64)).to(dev)
        rhs = torch.randn((1, n, 1), generator=g, dtype=torch.float64).to(dev)
        kw = {"L+inv": dict(want_L=True, want_inv=True), "solve": dict(want_L=False, rhs=rhs),
              "L+inv+solve": dict(want_L=True, want_inv=True, rhs=rhs)}[mode]
        reps = 50
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                chol_small(A, **kw)
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    chol_small(A, **kw)
            gr.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            gr.replay()
            e1.record(st)
            st.synchronize()
        print(f"n={n} {mode}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (incl. the output allocations' fill kernels)", flush=True)


if __name__ == "__main__":
    main()
