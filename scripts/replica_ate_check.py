"""One replica of config 5's correctness check: the 72-frame "ATE vs ref" sequence (tests/golden/ate_sequence.npz) through the
whole HIP odometry loop in THIS process, while other replicas run the same thing in their own processes on the same GPU
(tests/test_gpu_dist.py::test_replica_sequences_share_one_gpu starts them with different start offsets).

    python scripts/replica_ate_check.py --rank R --delay SECONDS --out FILE.json [--pix float|double]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--delay", type=float, default=0.0)
    ap.add_argument("--pix", default="double")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import numpy as np
    import torch
    from como_amd.utils.ate import ate_rmse
    from scripts.ate_sequence import run_ate_sequence
    from como_amd.odom.frontend.photo_tracking import photo_tracking_pyr
    d = np.load(os.path.join(ROOT, "tests", "golden", "ate_sequence.npz"))
    G = {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k]) for k in d.files}
    torch.zeros(1, device="cuda:0")                      # context up before the start offset
    time.sleep(args.delay)
    t0 = time.perf_counter()
    kinds, poses, odo = run_ate_sequence(G, args.pix, "cuda:0")
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ref_kinds = [int(x) for x in G["kinds"]]
    tr = [k for k in range(len(ref_kinds)) if bool(G["tracked"][k]) and k in poses]
    est, ref = [poses[k] for k in tr], [G["T_w_curr"][k] for k in tr]
    res = {"rank": args.rank, "delay": args.delay, "seconds": el, "frames": len(ref_kinds), "tracked": len(tr),
           "same_decisions": sum(int(a == b) for a, b in zip(kinds, ref_kinds)), "ate_rmse": ate_rmse(est, ref),
           "worst_pose_abs": max((e - r).abs().max().item() for e, r in zip(est, ref)),
           "kf_timestamps_equal": [float(t) for t in odo.mapping.kf_timestamps] == G["m_kf_timestamps"].tolist(),
           "tracking_chain_fallbacks": int(getattr(photo_tracking_pyr, "fallbacks", 0))}
    json.dump(res, open(args.out, "w"))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
