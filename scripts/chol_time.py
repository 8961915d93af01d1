"""Time the dense SPD solve alone (como_chol_solve_f64) at the window system sizes: graph replay of `reps` solves, HIP events.
Usage: python scripts/chol_time.py [D ...]   (COMO_CHOL_RIDE_MAX=0 forces the separate back-substitution kernels)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import como_amd.odom.backend.linear_system as ls


def main():
    dims = [int(a) for a in sys.argv[1:]] or [760]
    dev = torch.device("cuda:0")
    for D in dims:
        g0 = torch.Generator().manual_seed(D)
        A = torch.randn((D, D + 8), generator=g0, dtype=torch.float64)
        H = (A @ A.T + 1e-3 * torch.eye(D, dtype=torch.float64)).to(dev)
        H[0, 0] += 1e12
        g = torch.randn(D, generator=g0, dtype=torch.float64).to(dev)
        ws = {}
        d = ls.solve_system(H, g, ws).clone()
        ref = torch.cholesky_solve(g[:, None], torch.linalg.cholesky(H))
        rel = ((d - ref).norm() / ref.norm()).item()
        resid = ((H @ d - g[:, None]).abs().max() / g.abs().max()).item()
        reps = 50
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                ls.solve_system(H, g, ws)
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    ls.solve_system(H, g, ws)
            gr.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            gr.replay()
            e1.record(st)
            st.synchronize()
        print(f"D={D} solve {e0.elapsed_time(e1) / reps * 1e3:.1f} us  rel={rel:.2e} resid={resid:.2e} info={int(ls.solve_system.last_info)}",
              flush=True)


if __name__ == "__main__":
    main()
