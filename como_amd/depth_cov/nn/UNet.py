"""DepthCov UNet on the HIP kernels of csrc/nn.hip (float32 inference).

Mirror of como/depth_cov/nn/UNet.py:8-78 and como/depth_cov/nn/layers.py:5-75.  Parameters come from a state dict with the
reference's own key names (``base.conv1.weight`` ... ``up_convs.4.upsample.1.bias``, ``feature_convs.3.weight``), so a
DepthCov checkpoint's ``gaussian_cov_net.*`` entries load unchanged.  Every convolution is an MFMA implicit GEMM; torch is
used for buffers only.
"""
import ctypes

import torch

from como_amd import _lib

LEAKY_SLOPE = 0.01      # nn.LeakyReLU default (layers.py:8)
GN_GROUPS = 16          # nn.GroupNorm(16, C) (layers.py:17)
GN_EPS = 1e-5


def _relayout(w):
    """torch conv weight (Cout, Cin, k, k) -> [k*k][CinP][Cout] with CinP = Cin rounded up to 4."""
    cout, cin, k, _ = w.shape
    cinp = (cin + 3) // 4 * 4
    out = torch.zeros((k * k, cinp, cout), dtype=torch.float32, device=w.device)
    out[:, :cin, :] = w.permute(2, 3, 1, 0).reshape(k * k, cin, cout)
    return out.contiguous(), cin, cinp, cout, k


class _Conv:
    def __init__(self, weight, bias):
        self.wt, self.cin, self.cinp, self.cout, self.k = _relayout(weight.float())
        self.bias = bias.float().contiguous()

    def __call__(self, x, out=None, coff=0, gn_sums=None):
        """gn_sums: optional zeroed (32, N, 16, 2) float64 tensor receiving the GroupNorm statistics of the output."""
        N, C, H, W = x.shape
        assert C == self.cin and x.dtype == torch.float32 and x.is_contiguous()
        if out is None:
            out = torch.empty((N, self.cout, H, W), dtype=torch.float32, device=x.device)
        rc = _lib.lib().como_nn_conv2d_f32(x.data_ptr(), self.wt.data_ptr(), self.bias.data_ptr(), out.data_ptr(), N,
                                           self.cin, self.cinp, self.cout, H, W, self.k, out.shape[1], coff,
                                           gn_sums.data_ptr() if gn_sums is not None else None, GN_GROUPS,
                                           _lib.stream_ptr(x.device))
        _lib.check(rc, "como_nn_conv2d_f32")
        return out


def _groupnorm(x, gamma, beta, act, residual=None, sums=None):
    """sums: the (32,N,16,2) float64 statistics a convolution accumulated for x; None = compute them here."""
    N, C, H, W = x.shape
    out = torch.empty_like(x)
    stats = torch.empty((N * GN_GROUPS * 2,), dtype=torch.float32, device=x.device) if sums is None else None
    rc = _lib.lib().como_nn_groupnorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          residual.data_ptr() if residual is not None else None, out.data_ptr(),
                                          stats.data_ptr() if stats is not None else None,
                                          sums.data_ptr() if sums is not None else None, N, C, GN_GROUPS, H * W, GN_EPS,
                                          LEAKY_SLOPE, act, _lib.stream_ptr(x.device))
    _lib.check(rc, "como_nn_groupnorm_f32")
    return out


# 1: the scale / shift of a wide level's GroupNorm formed by the producing convolution's last wave (csrc/nn.hip gn_arrive_finalize)
# instead of the separate gn_finalize launch.  Built, tested, measured SLOWER on MI355X (forward 0.703 -> 0.753 ms: the returning
# memory-side adds and the last wave's uncached read of the 32 slots cost a tiled layer 5-8 us, the launch they replace 4.7): OFF.
GN_IN_CONV = __import__("os").environ.get("COMO_NN_GN_IN_CONV", "0") == "1"


def gn_sums_doubles(N):
    """float64 words of one normalisation's statistics: 32 contention slots x N x groups x (sum, sum of squares) + the arrival
    counter of the in-kernel finalisation (one 8-byte word; + one of padding)."""
    return 32 * N * GN_GROUPS * 2 + 2


DEEP_MAX_PIXELS = 768   # levels of 24x32 and below run their 3x3 layers on the reduction-split kernels (csrc/nn.hip conv3_deep)
_deep_part = {}         # per (device, stream): the partial-sum scratch of the deep layers (grow-only)
_deep_retired = []      # superseded scratch blocks: NEVER released -- a captured forward (DepthCovModule.forward_graphed) has the raw
                        # address of the block that was current at capture time baked into its kernel arguments; handing that block
                        # back to the caching allocator would let a replay write partial sums into somebody else's tensor.
                        # (1.5 MB per network size: a handful of blocks per process at most.)


def _deep_scratch(dev, need):
    """Scratch of at least `need` floats for the reduction-split layers, keyed by device AND stream (two networks running on two
    streams of one device -- two mappers -- must not share partial sums, like distill_depth._gram_ws / dense_ref._di_ws)."""
    key = (str(dev), _lib.stream_ptr(dev))
    buf = _deep_part.get(key)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _deep_retired.append(buf)
        buf = _deep_part[key] = torch.empty(need, dtype=torch.float32, device=dev)
    return buf


def _conv3_any(conv, x, out=None, coff=0, pro_scsh=None, norm=None, sums=None):
    """One 3x3 layer of the fused network -> (output, scsh of the GroupNorm `norm` = (gamma, beta) that follows, or None).
    Deep levels: the reduction-split kernel + its deterministic reduce (statistics included); wide levels: the generic kernel with
    the statistics accumulated into `sums` (pre-zeroed (32,N,16,2) float64) and finalised by one tiny launch."""
    N, C, H, W = x.shape
    L, s = _lib.lib(), _lib.stream_ptr(x.device)
    if out is None:
        out = torch.empty((N, conv.cout, H, W), dtype=torch.float32, device=x.device)
    scsh = torch.empty((N, conv.cout, 2), dtype=torch.float32, device=x.device) if norm is not None else None
    if H * W <= DEEP_MAX_PIXELS and conv.cin >= 64:
        need = L.como_nn_deep_part_floats(N, conv.cin, conv.cout, H, W)
        part = _deep_scratch(x.device, need)
        rc = L.como_nn_conv3x3_deep_f32(x.data_ptr(), conv.wt.data_ptr(), conv.bias.data_ptr(), out.data_ptr(), N, conv.cin, conv.cinp,
                                        conv.cout, H, W, out.shape[1], coff, _lib.ptr(pro_scsh), LEAKY_SLOPE, part.data_ptr(),
                                        part.numel(), GN_GROUPS, norm[0].data_ptr() if norm else None,
                                        norm[1].data_ptr() if norm else None, GN_EPS, _lib.ptr(scsh), s)
        _lib.check(rc, "como_nn_conv3x3_deep_f32")
        return out, scsh
    if norm is not None and sums is None:
        sums = torch.zeros(gn_sums_doubles(N), dtype=torch.float64, device=x.device)
    if norm is not None and GN_IN_CONV:
        # the scale / shift of the normalisation that follows are formed by the convolution's last wave (csrc/nn.hip gn_arrive_finalize)
        rc = L.como_nn_conv2d_gn_f32(x.data_ptr(), conv.wt.data_ptr(), conv.bias.data_ptr(), out.data_ptr(), N, conv.cin, conv.cinp,
                                     conv.cout, H, W, 3, out.shape[1], coff, sums.data_ptr(), GN_GROUPS, _lib.ptr(pro_scsh),
                                     LEAKY_SLOPE, norm[0].data_ptr(), norm[1].data_ptr(), GN_EPS, scsh.data_ptr(), s)
        _lib.check(rc, "como_nn_conv2d_gn_f32")
        return out, scsh
    rc = L.como_nn_conv2d_fused_f32(x.data_ptr(), conv.wt.data_ptr(), conv.bias.data_ptr(), out.data_ptr(), N, conv.cin, conv.cinp,
                                    conv.cout, H, W, 3, out.shape[1], coff, sums.data_ptr() if norm is not None else None, GN_GROUPS,
                                    _lib.ptr(pro_scsh), None, None, LEAKY_SLOPE, s)
    _lib.check(rc, "como_nn_conv2d_fused_f32")
    if norm is not None:
        _lib.check(L.como_nn_gn_finalize_f32(sums.data_ptr(), norm[0].data_ptr(), norm[1].data_ptr(), N, conv.cout, GN_GROUPS, H * W,
                                             GN_EPS, scsh.data_ptr(), s), "como_nn_gn_finalize_f32")
    return out, scsh


class ResidualConv:
    """layers.py:5-27: act(conv3(x) + norm(conv2(act(norm(conv1(x)))))) -- ONE GroupNorm module used twice."""

    def __init__(self, sd, prefix):
        self.conv1 = _Conv(sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"])
        self.conv2 = _Conv(sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
        self.conv3 = _Conv(sd[prefix + "conv3.weight"], sd[prefix + "conv3.bias"])
        self.gamma = sd[prefix + "norm.weight"].float().contiguous()
        self.beta = sd[prefix + "norm.bias"].float().contiguous()

    def __call__(self, x, sums=None, out=None, coff=0):
        """The fused form: conv1 -> conv2 reading its input through norm + act -> conv3 with `+ norm(conv2)`, act in its epilogue.
        sums: (2, 32, N, 16, 2) zeroed float64 scratch for the statistics of the wide levels (None: allocated here);
        out / coff: write the block's output into channels [coff, coff + Cout) of `out` (the concatenation of UpConv)."""
        N, C, H, W = x.shape
        assert C == self.conv1.cin and x.dtype == torch.float32 and x.is_contiguous()
        norm = (self.gamma, self.beta)
        s1, s2 = (sums[0], sums[1]) if sums is not None else (None, None)
        y1, scsh1 = _conv3_any(self.conv1, x, norm=norm, sums=s1)
        y2, scsh2 = _conv3_any(self.conv2, y1, pro_scsh=scsh1, norm=norm, sums=s2)
        if out is None:
            out = torch.empty((N, self.conv3.cout, H, W), dtype=torch.float32, device=x.device)
        c3 = self.conv3
        rc = _lib.lib().como_nn_conv2d_fused_f32(x.data_ptr(), c3.wt.data_ptr(), c3.bias.data_ptr(), out.data_ptr(), N, c3.cin, c3.cinp,
                                                 c3.cout, H, W, 1, out.shape[1], coff, None, GN_GROUPS, None, y2.data_ptr(),
                                                 scsh2.data_ptr(), LEAKY_SLOPE, _lib.stream_ptr(x.device))
        _lib.check(rc, "como_nn_conv2d_fused_f32")
        return out

    def unfused(self, x, sums=None):
        """Round 2-4's form (five launches: two separate GroupNorm passes): kept as the layer-by-layer check of the fused one."""
        s1, s2 = (sums[0], sums[1]) if sums is not None else (None, None)
        y = _groupnorm(self.conv1(x, gn_sums=s1), self.gamma, self.beta, 1, sums=s1)
        y2 = self.conv2(y, gn_sums=s2)
        skip = self.conv3(x)
        return _groupnorm(y2, self.gamma, self.beta, 2, residual=skip, sums=s2)


def maxpool2(x):
    N, C, H, W = x.shape
    out = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().como_nn_maxpool2_f32(x.data_ptr(), out.data_ptr(), N * C, H, W, _lib.stream_ptr(x.device)),
               "como_nn_maxpool2_f32")
    return out


def upsample2x(x):
    N, C, H, W = x.shape
    out = torch.empty((N, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().como_nn_upsample2x_f32(x.data_ptr(), out.data_ptr(), N * C, H, W, _lib.stream_ptr(x.device)),
               "como_nn_upsample2x_f32")
    return out


def normalize_imagenet(x):
    """transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]) (UNet.py:24-27)."""
    N, C, H, W = x.shape
    assert C == 3
    out = torch.empty_like(x)
    mean = (ctypes.c_float * 3)(0.485, 0.456, 0.406)
    std = (ctypes.c_float * 3)(0.229, 0.224, 0.225)
    _lib.check(_lib.lib().como_nn_normalize_f32(x.data_ptr(), out.data_ptr(), N, H * W, mean, std, _lib.stream_ptr(x.device)),
               "como_nn_normalize_f32")
    return out


def cov_activation(f):
    """gk.kernel_params_to_covariance(gk.normalize_params_cov(f)) : (N,3,H,W) -> (N,4,H,W)."""
    N, C, H, W = f.shape
    assert C == 3
    out = torch.empty((N, 4, H, W), dtype=torch.float32, device=f.device)
    _lib.check(_lib.lib().como_nn_cov_act_f32(f.data_ptr(), out.data_ptr(), N, H * W, _lib.stream_ptr(f.device)),
               "como_nn_cov_act_f32")
    return out


def resize_aa(x, size):
    """TF.resize(x, size, BILINEAR, antialias=True) (Mapping.py:411-426) for float32 / float64 NCHW tensors."""
    _lib.require_cuda(x)
    x = x.contiguous()
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
    fn = getattr(_lib.lib(), "como_nn_resize_aa_" + _lib.suffix(x.dtype))
    _lib.check(fn(x.data_ptr(), out.data_ptr(), N * C, H, W, Ho, Wo, _lib.stream_ptr(x.device)), "como_nn_resize_aa")
    return out


class UNet:
    """UNet.forward (UNet.py:57-78): returns the per-level feature maps AFTER the output activation is applied by the caller
    (`feature_act`), finest level last."""

    def __init__(self, state_dict, num_levels=5, prefix="", feature_act=None):
        sd = state_dict
        self.num_levels = num_levels
        self.base = ResidualConv(sd, prefix + "base.")
        self.down = [ResidualConv(sd, f"{prefix}down_convs.{i}.conv_block.") for i in range(num_levels)]
        self.up_conv = [_Conv(sd[f"{prefix}up_convs.{i}.upsample.1.weight"], sd[f"{prefix}up_convs.{i}.upsample.1.bias"])
                        for i in range(num_levels)]
        self.up_block = [ResidualConv(sd, f"{prefix}up_convs.{i}.conv_block.") for i in range(num_levels)]
        self.feature = [_Conv(sd[f"{prefix}feature_convs.{i}.weight"], sd[f"{prefix}feature_convs.{i}.bias"])
                        for i in range(num_levels - 1)]
        self.feature_act = feature_act

    def forward(self, x, finest_only=False):
        """finest_only: evaluate only the LAST (finest) feature head -- the one level the odometry path reads
        (Mapping.run_model: `model(rgb)[-1]`); the other entries of the returned list are None."""
        _lib.require_cuda(x)
        x = normalize_imagenet(x.float().contiguous())
        # GroupNorm statistics of all 2 * (1 + 2 * levels) normalisations: one zero-fill, accumulated by the convolutions
        nres = 1 + 2 * self.num_levels
        N = x.shape[0]
        sums = torch.zeros((nres, 2, gn_sums_doubles(N)), dtype=torch.float64, device=x.device)   # 32 slots + arrival counter each
        # One image: an encoder block writes its output straight into the second half of the decoder's concatenation buffer of
        # its level (channels [c, 2c) of one image are one contiguous block, so the pooling that follows reads it in place)
        # instead of being copied there later.
        direct = N == 1
        cats = [None] * self.num_levels
        enc = []
        blocks = [self.base] + self.down
        cur = x
        for i in range(self.num_levels + 1):
            blk = blocks[i]
            c = blk.conv3.cout
            H, W = cur.shape[-2:]
            if direct and i < self.num_levels:
                cats[i] = torch.empty((N, 2 * c, H, W), dtype=torch.float32, device=x.device)
                blk(cur, sums[i], out=cats[i], coff=c)
                e = cats[i][:, c:]
            else:
                e = blk(cur, sums[i])
            enc.append(e)
            if i < self.num_levels:
                cur = maxpool2(e)                                            # DownConv (layers.py:30-43)
        out = []
        dec = enc[-1]
        for i in range(self.num_levels - 1, -1, -1):                         # UpConv (layers.py:46-75)
            skip = enc[i]
            _, c, H, W = skip.shape
            cat = cats[i]
            if cat is None:
                cat = torch.empty((N, 2 * c, H, W), dtype=torch.float32, device=x.device)
                cat[:, c:].copy_(skip)
            _conv3_any(self.up_conv[i], upsample2x(dec), out=cat, coff=0)
            dec = self.up_block[i](cat, sums[1 + self.num_levels + i])
            if i < self.num_levels - 1:
                if finest_only and i > 0:
                    out.append(None)
                    continue
                f = self.feature[i](dec)
                out.append(self.feature_act(f) if self.feature_act else f)
        return out

    __call__ = forward
