import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in d.files:
        a = d[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a        # strings (tags) stay numpy
    return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


# ---- diagnostics: every gpu test records the errors it measured; dumped to gpurun_out/gpu_report.json ----
_REPORT = []


def report(name, **vals):
    rec = {"name": name}
    for k, v in vals.items():
        if torch.is_tensor(v):
            v = v.item() if v.numel() == 1 else v.tolist()
        rec[k] = v
    _REPORT.append(rec)
    print("REPORT", rec)


def pytest_sessionfinish(session, exitstatus):
    if _REPORT:
        import json
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "gpu_report.json"), "w") as f:
            json.dump(_REPORT, f, indent=1, default=str)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def scaled_err(H, Href):
    """Scale-aware matrix error: max |D^-1/2 (H - Href) D^-1/2| with D = diag(Href) (rows with a zero diagonal are compared
    unscaled).  The Jacobi-scaled reference has a unit diagonal whatever the 1e12 pose anchor does to max|H|, so this
    measures every block -- photometric, prior, anchor -- relative to its own magnitude (a max-norm relative error of the
    full system only ever sees the anchor)."""
    H, Href = H.detach().double().cpu(), Href.detach().double().cpu()
    d = torch.sqrt(torch.diagonal(Href).abs())
    d = torch.where(d > 0, d, torch.ones_like(d))
    return ((H - Href).abs() / d[:, None] / d[None, :]).max().item()
