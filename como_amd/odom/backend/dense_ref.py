"""Factored dense reference points on the GPU (HIP kernel csrc/densify.hip `dense_ref`).

Replaces the per-pixel part of the reference's Mapping.prep_dense_ref / sparse_map.setup_test_points
(Mapping.py:661-699, sparse_map.py:184-230) without ever building dPwn_dzm (B,n,3,m): the outputs are the rank-1
factors the BA kernel consumes (structure-of-arrays planes), plus the exact per-keyframe median depth.
"""
import torch

from como_amd import _lib

_ws = {}


def dense_reference_planes(Kt, n, compact, ws=None):
    """The output planes `dense_reference_factored` writes for (Kt, n, compact) -- created on first use in the caller-owned dict `ws`
    (or the module's) -- without launching anything: {"Pwn", "dT", "uvec", "z", "logz", "med", "hists"}.  (A caller that marshals
    the consumers' argument blocks before the first launch needs their addresses.)"""
    dt, dev = Kt.dtype, Kt.device
    B = Kt.shape[0]
    key = (str(dev), dt, B, n) if not compact else (str(dev), dt, B, n, "compact")
    own = ws
    w = own.get(key) if own is not None else _ws.get(key)
    if w is None:
        w = {"Pwn": torch.empty((B, 3, n), device=dev, dtype=dt),
             "dT": torch.empty((B, 6 if compact else 18, n), device=dev, dtype=dt),
             "uvec": None if compact else torch.empty((B, 3, n), device=dev, dtype=dt),
             "z": torch.empty((B, n), device=dev, dtype=dt),
             "logz": torch.empty((B, n), device=dev, dtype=dt), "med": torch.empty((B, 3), device=dev, dtype=dt),
             "hists": torch.empty((B * _lib.lib().como_select_workspace_bytes() // 4,), device=dev, dtype=torch.int32)}
        if own is not None:
            own[key] = w
        else:
            _ws.clear()
            _ws[key] = w
    return w


def dense_reference_factored(logzm, Twc, Kt, pixidx, K, dlogzm_dTwc, Wimg, want_logz=True, med_out=None, pixcoord=None,
                             hists=None, ws=None, part="all", compact=False, fuse=None, call_cache=None, stream=None):
    """logzm (B,m[,1]) Twc (B,4,4) Kt (B,rows,m) pixidx (B,n) int32 or None K (3,3) dlogzm_dTwc (B,m[,1],6).
    pixcoord: optional (B,n) int32 linear pixel index (row*W+col) when it differs from the K~ row index.
    hists: optional caller-owned, ALREADY ZEROED select workspace (B * como_select_workspace_bytes()): skips the clear.
    med_out: optional caller-owned (B,3) buffer for {median depth, 1.4826*median, n} (fixed address for fused chains).
    ws: optional caller-owned dict that receives / holds the output planes (a captured graph records their addresses, so
    whoever owns the graph must own them).  part: "all", or "points" then "median" as two calls (the median half -- select
    passes + finish -- can run on another stream; only the priors read it).
    compact: the form the tuned block kernels consume (como_ba_args.zmode 2): instead of the 18 + 3 planes of dPwn_dTwc and
    uvec only the six planes dlogz_n/dT_wc = K~[n,:] dlogz_m/dT_wc are written -- the kernels rebuild
    dP_w/dT_wc = [-[u]x R, R] + u (x) dlogz_n/dT_wc with u = P_w - t_wc from the reference pose.
    Returns Pwn (B,3,n), dPwn_dTwc (B,18,n) [compact: dlogzn_dTwc (B,6,n)], uvec (B,3,n) [compact: None], median depth (B,),
    logzn (B,n)."""
    # call_cache: caller-owned dict valid while NOTHING about the call changes (one window topology: the same buffers every
    # iteration): the marshalled argument tuple is kept per `part` and replayed -- the conversions / views / pointer look-ups below
    # were ~40 us of host time per call.  stream: raw stream handle to launch on (None: the current stream).
    if call_cache is not None:
        ent = call_cache.get(("dr", part))
        if ent is not None:
            fn, args, fz, out = ent
            st = stream if stream is not None else _lib.stream_ptr(Kt.device)
            rc = fn(*args, fz, st) if fz is not None else fn(*args, st)
            _lib.check(rc, "como_dense_ref")
            return out
    _lib.require_cuda(logzm, Twc, Kt, K, dlogzm_dTwc)
    dt, dev = Kt.dtype, Kt.device
    B, rows, m = Kt.shape
    n = pixidx.shape[1] if pixidx is not None else rows
    L = _lib.lib()
    ws = dense_reference_planes(Kt, n, compact, ws)
    med = med_out if med_out is not None else ws["med"]
    lz = logzm.reshape(B, m).to(dt).contiguous()
    dl = dlogzm_dTwc.reshape(B, m, 6).to(dt).contiguous()
    Tw = Twc.to(dt).contiguous()
    Kc = K.to(dt).contiguous()
    flags = (1 if hists is not None else 0) | {"all": 0, "points": 2, "median": 4}[part] | (16 if compact else 0)
    args = (Kt.data_ptr(), Kt.stride(0), _lib.ptr(pixidx), lz.data_ptr(), Tw.data_ptr(), Kc.data_ptr(), dl.data_ptr(), B, n, m,
            int(Wimg), ws["Pwn"].data_ptr(), ws["dT"].data_ptr(), _lib.ptr(ws["uvec"]), ws["z"].data_ptr(),
            ws["logz"].data_ptr() if want_logz else None, (hists if hists is not None else ws["hists"]).data_ptr(), med.data_ptr(),
            _lib.ptr(pixcoord), flags)
    st = stream if stream is not None else _lib.stream_ptr(dev)
    out = (ws["Pwn"], ws["dT"], ws["uvec"], med[:, 0], ws["logz"])
    if fuse is not None and part != "median":
        # fuse: a filled _lib.DRFuse (pass 1 of the photometric system rides in this launch: include/como_hip.h como_dr_fuse)
        import ctypes
        fn, fz = getattr(L, "como_dense_ref_fused_" + _lib.suffix(dt)), ctypes.byref(fuse)
        rc = fn(*args, fz, st)
    else:
        fn, fz = getattr(L, "como_dense_ref_" + _lib.suffix(dt)), None
        rc = fn(*args, st)
    _lib.check(rc, "como_dense_ref")
    if call_cache is not None:
        # (the tensors behind the pointers stay alive in the entry)
        call_cache[("dr", part)] = (fn, args, fz, out)
        call_cache[("dr_keep", part)] = (lz, dl, Tw, Kc, ws, med, fuse, hists, pixidx, pixcoord, Kt)
    return out


_di_ws = {}


def depth_image(Kt, logzm, out=None, logz_out=None):
    """exp(K~ logz_m) for every row of K~ (Mapping.store_vars' depth images, Mapping.py:749-758; the tracker asks for the newest
    keyframe's on every frame): the depth-only pass of `como_dense_ref_*` (flag 8) -- one streaming pass over K~ at HBM rate --
    instead of a (rows x m) . (m x 1) library GEMM + exp (rocBLAS ran that GEMV shape at 0.7 TB/s: 216 us per 640x480 keyframe).
    Kt (B,rows,m) (a view with a row stride is fine), logzm (B,m[,1]); returns (B,rows) depths (into `out` when given);
    logz_out (B,rows): also receives the log-depths K~ logz_m themselves."""
    _lib.require_cuda(Kt, logzm)
    dt, dev = Kt.dtype, Kt.device
    B, rows, m = Kt.shape
    if Kt.stride(2) != 1 or Kt.stride(1) != m:
        Kt = Kt.contiguous()
    L = _lib.lib()
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream, dt, B, m)     # (per stream: see photo._buf)
    w = _di_ws.get(key)
    if w is None:
        w = _di_ws[key] = {"hists": torch.zeros((B * L.como_select_workspace_bytes() // 4,), device=dev, dtype=torch.int32),
                           "eye": torch.eye(4, device=dev, dtype=dt).repeat(B, 1, 1).contiguous(),
                           "K": torch.eye(3, device=dev, dtype=dt), "dl": torch.zeros((B, m, 6), device=dev, dtype=dt),
                           "med": torch.zeros((B, 3), device=dev, dtype=dt)}
    z = out if out is not None else torch.empty((B, rows), device=dev, dtype=dt)
    lz = logzm.reshape(B, m)
    if lz.dtype != dt or not lz.is_contiguous():
        lz = lz.to(dt).contiguous()
    rc = getattr(L, "como_dense_ref_" + _lib.suffix(dt))(
        Kt.data_ptr(), Kt.stride(0), None, lz.data_ptr(), w["eye"].data_ptr(), w["K"].data_ptr(), w["dl"].data_ptr(), B, rows, m, 1,
        None, None, None, z.data_ptr(), _lib.ptr(logz_out), w["hists"].data_ptr(), w["med"].data_ptr(), None, 8 | 2 | 64, _lib.stream_ptr(dev))
    _lib.check(rc, "como_dense_ref (depth image)")
    return z


BAND_MEDIAN = __import__("os").environ.get("COMO_BAND_MEDIAN", "1") != "0"


def full_image_median(logzm, Kt, med_out, ws, hists=None, part="all", reduce=None, band=None, reduce_max=None, call_cache=None,
                      stream=None):
    """Mapping.store_vars (Mapping.py:749-758): per keyframe the exact median of exp(K~ logz_m) over ALL rows of K~
    (the full depth image).  logzm (B,m), Kt (B,rows,m); med_out (B,3) caller-owned {median, 1.4826 median, n};
    ws: caller-owned dict (depth plane + select histograms; a captured graph records the addresses).
    band (default: on, COMO_BAND_MEDIAN=0 disables): `como_depth_band_*` instead of the depth-only pass -- only the pixels whose
    cached log-depth interval can straddle the new median are re-evaluated (state kept in `ws`; med_out must then be the SAME
    buffer on every call: it holds the previous median); the first call with a given `ws` builds the state.  Exact either way.
    reduce_max(t): multi-GPU (Kt = this rank's rows): all-reduce(MAX) applied ONCE, to the per-keyframe largest row norm when the
    state is built -- the bound on how far the GLOBAL median can move must cover every rank's rows.
    part: "all", or "points" (kernel + pass-0 histogram) then "median" (remaining select passes + finish).
    reduce: multi-GPU -- Kt is then this rank's ROW RANGE of every keyframe's predictor (a view) and the digit histograms
    are all-reduced between the passes (median_passes), so every rank gets the median of the whole image."""
    if call_cache is not None:
        ent = call_cache.get(("fm", part))
        if ent is not None:                                  # (the plain streaming pass only: the band form keeps host-side state)
            fn, args, out = ent
            _lib.check(fn(*args, stream if stream is not None else _lib.stream_ptr(Kt.device)), "como_dense_ref (depth only)")
            return out
    _lib.require_cuda(logzm, Kt)
    dt, dev = Kt.dtype, Kt.device
    B, rows, m = Kt.shape
    L = _lib.lib()
    key = ("full", str(dev), dt, B, rows)
    w = ws.get(key)
    if w is None:
        w = {"z": torch.empty((B, rows), device=dev, dtype=dt),
             "hists": torch.zeros((B * L.como_select_workspace_bytes() // 4,), device=dev, dtype=torch.int32),
             "eye": torch.eye(4, device=dev, dtype=dt).repeat(B, 1, 1).contiguous(),
             "K": torch.eye(3, device=dev, dtype=dt), "dl": torch.zeros((B, m, 6), device=dev, dtype=dt)}
        ws[key] = w
    lz = logzm.reshape(B, m)
    if lz.dtype != dt or not lz.is_contiguous():
        lz = lz.to(dt).contiguous()
    fn = getattr(L, "como_dense_ref_" + _lib.suffix(dt))
    h = hists if hists is not None else w["hists"]
    if reduce is not None:
        part_flag = 2                                    # kernel + pass-0 histogram here, the passes below
    else:
        part_flag = {"all": 0, "points": 2, "median": 4}[part]
    use_band = (BAND_MEDIAN if band is None else band) and hists is not None and part_flag != 4
    if use_band:
        st = w.get("band")
        # the band bound is only valid for the K~ it was built from: a reused workspace with another predictor (a window that
        # slid inside its buffer, another row range) rebuilds the state instead of returning medians of stale row norms
        ident = (Kt.data_ptr(), Kt.stride(0), B, rows, m)
        init = st is None
        if st is not None and st.get("ident") != ident:
            if st["lref"].shape == (B, rows) and st["lref"].dtype == dt and st["prev"].shape == (B, m) and st["med"] == med_out.data_ptr():
                # same shapes: the state is rebuilt IN the planes it has (the init pass overwrites every row of them; only the
                # running maximum of the row norms starts from zero)
                st["l1max"].zero_()
                st["ncand"].zero_()
                st["calls"], st["ident"] = 0, ident
                init = True
            else:
                st, init = None, True
        if st is None:
            st = w["band"] = {"lref": torch.zeros((B, rows), device=dev, dtype=dt), "err": torch.zeros((B, rows), device=dev, dtype=dt),
                              "l1": torch.zeros((B, rows), device=dev, dtype=dt), "l1max": torch.zeros(B, device=dev, dtype=torch.float32),
                              "prev": torch.zeros((B, m), device=dev, dtype=dt), "ncand": torch.zeros(B, device=dev, dtype=torch.int32),
                              "calls": 0, "med": med_out.data_ptr(), "ident": ident}
        if st["med"] != med_out.data_ptr():
            raise RuntimeError("como_amd full_image_median: the band state belongs to another med_out buffer (it carries the previous median)")
        st["calls"] += 1
        if not lz.is_contiguous():
            lz = lz.contiguous()
        rc = getattr(L, "como_depth_band_" + _lib.suffix(dt))(
            Kt.data_ptr(), Kt.stride(0), lz.data_ptr(), st["prev"].data_ptr(), B, rows, m, st["lref"].data_ptr(), st["err"].data_ptr(),
            st["l1"].data_ptr(), st["l1max"].data_ptr(), med_out.data_ptr(), w["z"].data_ptr(), h.data_ptr(), st["ncand"].data_ptr(),
            1 if init else 0, _lib.stream_ptr(dev))
        _lib.check(rc, "como_depth_band")
        if init and reduce_max is not None:
            reduce_max(st["l1max"])
        if part_flag == 0:                               # "all": the remaining select passes + finish right away
            rc = fn(Kt.data_ptr(), Kt.stride(0), None, lz.data_ptr(), w["eye"].data_ptr(), w["K"].data_ptr(), w["dl"].data_ptr(), B,
                    rows, m, 1, None, None, None, w["z"].data_ptr(), None, h.data_ptr(), med_out.data_ptr(), None, 8 | 1 | 4,
                    _lib.stream_ptr(dev))
            _lib.check(rc, "como_dense_ref (median of the band plane)")
    else:
        args = (Kt.data_ptr(), Kt.stride(0), None, lz.data_ptr(), w["eye"].data_ptr(), w["K"].data_ptr(), w["dl"].data_ptr(), B, rows,
                m, 1, None, None, None, w["z"].data_ptr(), None, h.data_ptr(), med_out.data_ptr(), None,
                8 | (1 if hists is not None else 0) | part_flag)
        rc = fn(*args, stream if stream is not None else _lib.stream_ptr(dev))
        _lib.check(rc, "como_dense_ref (depth only)")
        if call_cache is not None and reduce is None:
            call_cache[("fm", part)] = (fn, args, med_out[:, 0])
            call_cache[("fm_keep", part)] = (Kt, lz, w, h, med_out)
    if reduce == "defer":
        return w["z"]                                    # the caller drives the select passes (shared all-reduces)
    if reduce is not None:
        if hists is None:
            raise RuntimeError("como_amd: the sharded full-image median needs a caller-zeroed histogram workspace")
        median_passes(w["z"], h, med_out, reduce)
    return med_out[:, 0]


def median_passes(z, hists, med_out, reduce=None, idle=False):
    """The select passes after a `part="points"` call, one at a time, with `reduce(t)` (an all-reduce(sum) across the ranks
    that share the keyframes' pixels) applied to every digit histogram right after it is complete: every rank then resolves
    the same exact per-keyframe median of the UNION of the shards.  z (B,n_local) depths written by the points kernel,
    hists its select workspace (pass 0 already accumulated), med_out (B,3)."""
    L = _lib.lib()
    B, n = z.shape
    dt = z.dtype
    sfx = _lib.suffix(dt)
    npass = 3 if dt == torch.float32 else 6
    hv = hists.view(B, 6, 2048)
    s = _lib.stream_ptr(z.device)
    for p in range(npass):
        if reduce is not None:
            stage = hv[:, p].contiguous()
            reduce(stage)
            hv[:, p].copy_(stage)
        if p + 1 < npass and not idle:                       # an idle rank (no pixels) only takes part in the collectives
            _lib.check(getattr(L, "como_select_hist_" + sfx)(z.data_ptr(), None, n, B, hists.data_ptr(), p + 1, s), "como_select_hist")
    _lib.check(getattr(L, "como_select_finish_" + sfx)(hists.data_ptr(), B, med_out.data_ptr(), s), "como_select_finish")
    return med_out[:, 0]
