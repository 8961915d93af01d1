"""Greedy conditional-entropy inducing-point sampler on the GPU (reference como/depth_cov/core/samplers.py).

Same functions and arguments as the reference (get_coords_domain :10-25, get_cov_domain :28-34, sample_sparse_coords
:38-107, precalc_entropy_vars :125-193, greedy_loop :196-282, greedy_conditional_entropy :285-324).  The per-step work is
the native ops (csrc/cov.hip: cross_covariance, Cholesky append + variance downdate) plus `como_greedy_next_f32` for
get_next_inds; the chosen index stays on the device, so the 63 steps of a 64-point sample enqueue without a host
synchronisation (the reference synchronises at least once per step when terminate_early is set; so does this loop then).
"""
import torch

from como_amd import _lib
from como_amd import como_backends
from como_amd.depth_cov.core import gaussian_kernel as gk
from como_amd.utils.coords import normalize_coordinates
from como_amd.utils.lin_alg import chol_small, trsm_lower


_domain_cache = {}
_pending_info = []
_INFO_SYNC = __import__("os").environ.get("COMO_SAMPLER_INFO_SYNC", "0") == "1"


def check_info(info):
    v = int(info.max()) if int(info.min()) >= 0 else int(info.min())
    if v < 0:
        raise RuntimeError("como_amd greedy_loop: the persistent sampler kernel timed out waiting for its workgroups (another kernel "
                           "or process holds compute units); set COMO_GREEDY_PERSIST=0")
    if v != 0:
        raise RuntimeError(f"como_amd precalc_entropy_vars: K_nn is not positive definite (leading minor {v})")


def check_pending_info(wait=False):
    """Raise if a sampler set-up since the last call found its K_nn not positive definite (the reference's torch.linalg.cholesky
    raises inside precalc_entropy_vars).  The status words travel to the host asynchronously (utils/hostlist.read_later); the ones
    that have arrived are checked -- all of them with wait=True."""
    keep = []
    while _pending_info:
        h = _pending_info.pop(0)
        if wait or h.ready():
            check_info(h.value())
        else:
            keep.append(h)
    _pending_info.extend(keep)


def get_coords_domain(cov_params_img, border=0):
    b, c, h, w = cov_params_img.shape
    dev = cov_params_img.device
    y, x = torch.meshgrid(torch.arange(h, dtype=torch.long, device=dev), torch.arange(w, dtype=torch.long, device=dev),
                          indexing="ij")
    img = torch.dstack((y, x))[border:h - border, border:w - border, :]
    return img.reshape(-1, 2).unsqueeze(0).repeat(b, 1, 1)


def get_cov_domain(coord_vec, cov_params_img):
    b = cov_params_img.shape[0]
    vec = cov_params_img[:, :, coord_vec[0, :, 0], coord_vec[0, :, 1]]
    return torch.permute(vec, (0, 2, 1)).reshape(b, -1, 2, 2).contiguous()


def random_uniform(n, coords_domain_norm):
    """samplers.py:110-114"""
    weights = torch.ones(coords_domain_norm.shape[:-1], device=coords_domain_norm.device)
    return torch.multinomial(weights, n, replacement=False)


def get_obs_info(L, K_mn):
    """L^-1 K_mn (samplers.py:117-118): forward substitution, one thread per pixel of the domain (csrc/smallsolve.hip)."""
    return trsm_lower(L, K_mn)


def calc_var(obs_info, K_diag):
    return K_diag - torch.sum(obs_info * obs_info, dim=1)


def precalc_entropy_vars(E_domain, gaussian_covs, n, coords_domain_norm, curr_coords_norm, curr_var, fixed_var, scale, curr_E=None):
    b, m, _ = curr_coords_norm.shape
    dev, dt = E_domain.device, E_domain.dtype
    d = coords_domain_norm.shape[-2]
    coord_vec_inds = torch.empty((b, n), device=dev, dtype=torch.long)
    coords_n_norm = torch.empty((b, n, 2), device=dev, dtype=dt)
    E_n = torch.empty((b, n, 2, 2), device=dev, dtype=dt)
    L = torch.eye(n, device=dev, dtype=dt).unsqueeze(0).repeat(b, 1, 1)
    # (rows >= m are written by the step that adds them before any later step reads them, rows < m below: no 77 MB zero-fill at
    # a 640x480 domain)
    obs_info = torch.empty((b, n, d), device=dev, dtype=dt)
    if m > 0:
        coord_vec_inds[:, :m] = -1
        coords_n_norm[:, 0:m, :] = curr_coords_norm
        # (curr_E: the parameters at the current points when the caller already interpolated them with the coordinates)
        E_n[:, :m, :, :] = curr_E if curr_E is not None else gk.interpolate_kernel_params(gaussian_covs, curr_coords_norm)
    else:
        areas = E_domain[..., 0, 0] * E_domain[..., 1, 1] - E_domain[..., 0, 1] * E_domain[..., 1, 0]
        best = torch.argmax(areas.view(b, -1), dim=1)
        bi = torch.arange(b, device=dev)
        coord_vec_inds[:, 0] = best
        coords_n_norm[:, 0, :] = coords_domain_norm[bi, best, :]
        E_n[:, 0, :, :] = E_domain[bi, best, :, :]
        m = 1
    K_nn = como_backends.cross_covariance(coords_n_norm[:, :m, :], E_n[:, :m, :, :], coords_n_norm[:, :m, :].clone(),
                                          E_n[:, :m, :, :].clone(), scale)
    if curr_var.shape[1] > 0:
        assert curr_var.shape[1] == curr_coords_norm.shape[1]
        K_nn += torch.diag_embed(curr_var)
    if fixed_var is not None and float(fixed_var) != 0.0:           # (+ 0.0 on the diagonal changes no bit of K_nn)
        K_nn += torch.diag_embed(fixed_var * torch.ones(b, m, device=dev))
    f = chol_small(K_nn, want_L=True, want_info=True)          # the initial factor (torch.linalg.cholesky in the reference)
    # ... which raises on a non-positive-definite K_nn: so does this -- at the caller's next synchronisation point
    # (`check_pending_info`, called by Mapping.add_keyframe / by sample_sparse_coords' callers that read anything back) instead of
    # a read-back of its own in the middle of the sampler's set-up (COMO_SAMPLER_INFO_SYNC=1: at once, as before)
    if _INFO_SYNC:
        check_info(f["info"])
    else:
        from como_amd.utils.hostlist import read_later
        check_pending_info(wait=len(_pending_info) >= 8)
        _pending_info.append(read_later(f["info"]))
    L[:, :m, :m] = f["L"]
    K_md = como_backends.cross_covariance(coords_n_norm[:, :m, :], E_n[:, :m, :, :], coords_domain_norm.view(b, -1, 2), E_domain,
                                          scale)
    if b == 1:
        trsm_lower(L[:, :m, :m], K_md, out=obs_info[:, :m, :])      # straight into the leading rows (contiguous for one image)
    else:
        obs_info[:, :m, :] = get_obs_info(L[:, :m, :m], K_md)
    return coord_vec_inds, coords_n_norm, E_n, L, obs_info, m


class _Next:
    """get_next_inds with a device-resident running distance mask."""

    def __init__(self, coords_domain_norm, dist_thresh):
        self.dom = coords_domain_norm.contiguous()
        b, d, _ = self.dom.shape
        dev = self.dom.device
        self.mask = torch.ones((b, d), dtype=torch.uint8, device=dev)
        self.best = torch.empty((b,), dtype=torch.long, device=dev)
        self.sd = torch.empty((b,), dtype=torch.float32, device=dev)
        self.t2 = float(dist_thresh) * float(dist_thresh)
        self.b, self.d = b, d

    def __call__(self, var, new_chosen):
        ch = new_chosen.contiguous()
        rc = _lib.lib().como_greedy_next_f32(var.data_ptr(), self.dom.data_ptr(), ch.data_ptr(), ch.shape[1], self.mask.data_ptr(),
                                             self.t2, self.best.data_ptr(), self.sd.data_ptr(), self.b, self.d,
                                             _lib.stream_ptr(var.device))
        _lib.check(rc, "como_greedy_next_f32")
        return self.sd, self.best


def greedy_loop(coord_vec_inds, coords_n_norm, E_n, coords_domain_norm, E_domain, L, obs_info, m, n, signal_var, fixed_var,
                max_stdev_thresh, terminate_early, dist_thresh):
    _lib.require_cuda(coords_n_norm, E_n, coords_domain_norm, E_domain, L, obs_info)
    if coords_n_norm.dtype != torch.float32:
        raise RuntimeError("como_amd greedy_loop: float32 only (as the reference's native Cholesky kernels)")
    dev = coords_n_norm.device
    b = coords_n_norm.shape[0]
    bi = torch.arange(b, device=dev)
    sv = float(signal_var)                                             # ONE read-back, before the loop
    k_ii = sv + (float(fixed_var) if fixed_var is not None else 0.0)
    nxt = _Next(coords_domain_norm, dist_thresh)
    pred_var = calc_var(obs_info[:, :m, :], sv).contiguous()
    # (the loop's in / out arrays are this module's own temporaries -- only the returned indices leave it -- so strided views
    # are simply packed)
    coords_n_norm, E_n, L, obs_info, coord_vec_inds, E_domain = (t.contiguous() for t in (coords_n_norm, E_n, L, obs_info,
                                                                                          coord_vec_inds, E_domain))
    # the whole loop on the device: two launches per added point.  Early termination (samplers.py:255-259) is decided
    # afterwards from the per-step trace of the largest remaining standard deviation -- the greedy sequence does not
    # depend on where it is cut -- with ONE read-back instead of one per step.  (Only the returned indices leave this function:
    # the packed copies above are NOT written back into strided arguments.)
    trace = torch.zeros((n + 1, b), device=dev, dtype=torch.float32) if terminate_early else None
    if PERSISTENT_LOOP and b == 1 and nxt.d > 4096 and m < n:
        # ONE persistent launch (csrc/cov.hip greedy_persist_kernel): the obs_info columns stay in registers / LDS, a step is one
        # grid-wide exchange instead of a pass over the 34-63 previous obs_info rows (1.2 MB each at 640x480).  Its status word
        # (-1: a grid-wide wait timed out) is checked like the set-up's, at the caller's next synchronisation.
        Lb = _lib.lib()
        ws = torch.empty(Lb.como_greedy_persist_workspace_bytes(n, nxt.d) // 4, device=dev, dtype=torch.float32)
        status = torch.zeros(1, device=dev, dtype=torch.int32)
        rc = Lb.como_greedy_persist_f32(coords_n_norm.data_ptr(), E_n.data_ptr(), coord_vec_inds.data_ptr(), nxt.dom.data_ptr(),
                                        E_domain.data_ptr(), L.data_ptr(), obs_info.data_ptr(), pred_var.data_ptr(), nxt.mask.data_ptr(),
                                        nxt.best.data_ptr(), nxt.sd.data_ptr(), sv, k_ii, nxt.t2, n, nxt.d, m, _lib.ptr(trace),
                                        ws.data_ptr(), status.data_ptr(), _lib.stream_ptr(dev))
        if rc == 0:
            from como_amd.utils.hostlist import read_later
            _pending_info.append(read_later(status))
            if terminate_early:
                below = (trace[m:n] < max_stdev_thresh).all(dim=1).tolist()
                check_pending_info(wait=True)
                for k, stop in enumerate(below):
                    if stop:
                        return coord_vec_inds[:, :m + k]
            return coord_vec_inds
        if rc != 1:
            _lib.check(rc, "como_greedy_persist_f32")           # (1 = the shape does not fit the persistent form: the loop below)
    # per-workgroup argmax partials: one float4 per scan slice / per append workgroup (the append also scans: csrc/cov.hip)
    scratch = torch.empty((b * max(4096, 4 * ((nxt.d + 255) // 256)),), device=dev, dtype=torch.float32)
    rc = _lib.lib().como_greedy_loop_ws_f32(coords_n_norm.data_ptr(), E_n.data_ptr(), coord_vec_inds.data_ptr(),
                                            nxt.dom.data_ptr(), E_domain.data_ptr(), L.data_ptr(), obs_info.data_ptr(),
                                            pred_var.data_ptr(), nxt.mask.data_ptr(), nxt.best.data_ptr(), nxt.sd.data_ptr(),
                                            sv, k_ii, nxt.t2, b, n, nxt.d, m, _lib.ptr(trace), scratch.data_ptr(), scratch.numel(),
                                            _lib.stream_ptr(dev))
    _lib.check(rc, "como_greedy_loop_ws_f32")
    if terminate_early:
        below = (trace[m:n] < max_stdev_thresh).all(dim=1).tolist()
        check_pending_info(wait=True)                       # (that read-back synchronised: every pending status word has arrived)
        for k, stop in enumerate(below):
            if stop:
                return coord_vec_inds[:, :m + k]
    return coord_vec_inds


PERSISTENT_LOOP = __import__("os").environ.get("COMO_GREEDY_PERSIST", "1") != "0"   # 0: two launches per added point (A/B, tests flip the attribute)
THIN_KERNEL = __import__("os").environ.get("COMO_GREEDY_THIN", "1") != "0"      # 0: the generic path (A/B, tests flip the attribute)


def _thin(cdn, E_domain, n, signal_var, fixed_var, max_stdev_thresh, dist_thresh):
    """The thinning pass of a keyframe insertion -- given candidate points, no current points, early termination -- as ONE launch
    (csrc/cov.hip `como_greedy_thin_f32`: the m = 0 seed of precalc_entropy_vars, the greedy loop and the cut) and one read-back
    (the number of picks) instead of ~45 launches and the read-back of the trace.  cdn (1,d,2), E_domain (1,d,2,2) float32."""
    dev, d = cdn.device, cdn.shape[1]
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    coords_n, E_n, L, obs, var, trace = f(n, 2), f(n, 4), f(n, n), f(n, d), f(d), f(n + 1)
    mask = torch.empty(d, dtype=torch.uint8, device=dev)
    both = torch.empty(2 + n, dtype=torch.long, device=dev)           # [best index, count | picked indices]: ONE read-back
    aux, inds = both[:2], both[2:].view(1, n)
    rc = _lib.lib().como_greedy_thin_f32(cdn.contiguous().data_ptr(), E_domain.contiguous().data_ptr(), coords_n.data_ptr(), E_n.data_ptr(),
                                         inds.data_ptr(), L.data_ptr(), obs.data_ptr(), var.data_ptr(), mask.data_ptr(), aux.data_ptr(),
                                         trace.data_ptr(), float(signal_var), float(signal_var),
                                         float(fixed_var) if fixed_var is not None else 0.0, float(dist_thresh) * float(dist_thresh),
                                         float(max_stdev_thresh), n, d, aux[1:].data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "como_greedy_thin_f32")
    vals = both[1:].tolist()                                          # the one host synchronisation: count AND the picks
    count = int(vals[0])
    if count == 0:
        raise RuntimeError("como_amd sample_sparse_coords: K_nn is not positive definite")
    _thin.last_list = [int(v) for v in vals[1:1 + count]]
    return inds[:, :count]


def greedy_conditional_entropy(gaussian_covs, E_domain, n, coords_domain_norm, curr_coords_norm, curr_var, fixed_var,
                               signal_var, max_stdev_thresh, terminate_early, dist_thresh, curr_E=None):
    cvi, cn, E_n, L, obs, m = precalc_entropy_vars(E_domain, gaussian_covs, n, coords_domain_norm, curr_coords_norm, curr_var,
                                                   fixed_var, signal_var, curr_E=curr_E)
    return greedy_loop(cvi, cn, E_n, coords_domain_norm, E_domain, L, obs, m, n, signal_var, fixed_var, max_stdev_thresh,
                       terminate_early, dist_thresh)


def sample_sparse_coords(cov_params_img, num_samples, mode, max_stdev_thresh=-1e8, border=0, terminate_early=False,
                         dist_thresh=0.0, signal_var=None, fixed_var=None, curr_coords=None, curr_var=None, coords_domain=None,
                         dtype=torch.float):
    """samplers.py:38-107 (mode "greedy_conditional_entropy"; "random_uniform" is a torch.multinomial one-liner there)."""
    b = cov_params_img.shape[0]
    img_size = cov_params_img.shape[-2:]
    dev = cov_params_img.device
    cov = cov_params_img.to(device=dev, dtype=dtype)
    sample_sparse_coords.last_picked_list = None            # host copy of the returned indices, when a read-back produced one
    if curr_coords is None:
        curr_coords = torch.empty((b, 0, 2), device=dev, dtype=dtype)
    if curr_var is None:
        curr_var = torch.zeros((b, 0), device=dev, dtype=dtype)
    if coords_domain is None:
        # the pixel grid inside the border and its normalised form only depend on the image size: built once (READ-ONLY, shared)
        key = (b, int(img_size[0]), int(img_size[1]), int(border), str(dev), dtype)
        ent = _domain_cache.get(key)
        if ent is None:
            coords_domain = get_coords_domain(cov, border=border)
            ent = _domain_cache[key] = (coords_domain, normalize_coordinates(coords_domain, img_size).to(dtype))
        coords_domain, cdn = ent
        E_domain = get_cov_domain(coords_domain, cov)
    else:
        cdn, E_domain = gk.kernel_params_at(cov, coords_domain, dtype)
        if (THIN_KERNEL and mode == "greedy_conditional_entropy" and terminate_early and b == 1 and curr_coords.shape[1] == 0 and
                curr_var.shape[1] == 0 and dtype == torch.float32 and cdn.is_cuda and 0 < coords_domain.shape[1] <= 1024):
            inds = _thin(cdn, E_domain, min(num_samples, coords_domain.shape[1]), signal_var, fixed_var, max_stdev_thresh, dist_thresh)
            sample_sparse_coords.last_picked_list = _thin.last_list
            return coords_domain.index_select(1, inds[0]), inds
    if mode == "random_uniform":
        inds = random_uniform(num_samples - curr_coords.shape[-2], cdn)
    elif mode == "greedy_conditional_entropy":
        n = min(num_samples, coords_domain.shape[1])
        if curr_coords.shape[1] > 0:
            ccn, curr_E = gk.kernel_params_at(cov, curr_coords, dtype)
        else:
            ccn, curr_E = normalize_coordinates(curr_coords, img_size).to(dtype), None
        inds = greedy_conditional_entropy(cov, E_domain, n, cdn, ccn, curr_var, fixed_var, signal_var, max_stdev_thresh,
                                          terminate_early, dist_thresh, curr_E=curr_E)
    else:
        raise ValueError("sample_sparse_coords mode: " + mode + " is not implemented.")
    if mode == "greedy_conditional_entropy":
        # the -1 entries are exactly the leading columns of the current points (precalc_entropy_vars; none when it seeded the set
        # itself): a slice instead of a boolean column mask (which synchronises with the host to learn its size)
        domain_inds = inds[:, curr_coords.shape[1]:]
    else:
        domain_inds = inds[:, inds[0, :] >= 0]
    if b == 1:
        return coords_domain.index_select(1, domain_inds[0]), domain_inds
    bi = torch.arange(b, device=dev).unsqueeze(1).repeat(1, domain_inds.shape[1])
    return coords_domain[bi, domain_inds, :], domain_inds
