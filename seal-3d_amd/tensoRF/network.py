"""TensoRF (vector-matrix) backbone of BASELINE config 5 — counterpart of tensoRF/network.py:13-199.

Density rank 16x3, colour rank 48x3 plane/line factors sampled with stock `F.grid_sample` (the reference has no custom
kernel here either, tensoRF/network.py:125-126), `basis_mat` 144 -> 27, colour MLP on
FreqEncoder(27, deg 2) + FreqEncoder(3, deg 2) = 150 -> 128 -> 128 -> 3.  The hot-path pieces it exercises are the
HIP `freqencoder` and `raymarching` packages.  Parameter names follow the reference (sigma_mat/sigma_vec/color_mat/
color_vec/basis_mat/color_net) so checkpoints keep their keys.

On the GPU the twelve grid_sample calls + stack / cat / mul / sum of `get_sigma_feat` / `get_color_feat` run as one kernel
per call (`s3d_vm_features_forward`, csrc/tensorf.hip), and so do their parameter gradients
(`s3d_vm_features_backward`: points binned by plane tile / line chunk, LDS accumulation, instead of the ~2.8e8 global
fp32 atomics per step grid_sample's backward issues at the Lego sample count).  A gradient w.r.t. the coordinates (not
needed by any trainer of the reference) falls back to the grid_sample sequence under autograd."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import s3d_hip
from activation import trunc_exp
from encoding import get_encoder
from nerf.renderer import NeRFRenderer


def _source_check(net, factors, *more):
    """`net._s3d_found_inf` (set by a trainer whose GradScaler owns that flag, tensoRF/utils.py): the factor backward raises it
    itself when a bound is not finite, and the parameters it writes are marked so that the scaler's pass over the remaining
    gradients skips them for this step (nerf/optim.py: NativeGradScaler._check_plain)"""
    flag = getattr(net, "_s3d_found_inf", None)
    if flag is None:
        return None
    owners = {id(p): p for p in list(net.sigma_mat) + list(net.sigma_vec) + list(net.color_mat) + list(net.color_vec) + [net.basis_mat.weight]}
    for f in list(factors) + list(more):
        p = owners.get(id(f))
        if p is None:  # (not the network's own parameter objects: leave the gradient to the scaler's pass)
            return None
    for f in list(factors) + list(more):
        owners[id(f)]._s3d_grad_checked = flag
    return flag


class _VmFeatures(torch.autograd.Function):
    """forward: one HIP kernel; backward: binned HIP kernels for the factor gradients (coordinates' gradient, if ever
    asked for: the reference's grid_sample sequence re-run under autograd)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # grid_sample is an fp32 op under autocast too
    def forward(ctx, x, net, reduce, *factors):
        x = x.contiguous()
        mats, vecs = factors[:3], factors[3:]
        N = x.shape[0]
        out = torch.empty((N,) if reduce else (sum(m.shape[1] for m in mats), N), dtype=torch.float32, device=x.device)
        # (a padded sample batch announced by the renderer, s3d_hip.row_limit: rows behind the device-side count are absent —
        #  not written here, sorted behind every bin in the backward, nerf/renderer.py never reads them)
        nv = s3d_hip.active_row_limit(N)
        mats, vecs = [m.contiguous() for m in mats], [v.contiguous() for v in vecs]
        sh = _shadows(net, mats, vecs, net.resolution)
        s3d_hip.VmBackend.features_forward(x, mats, vecs, net.resolution, reduce, out, n_valid=nv, shadows=sh)
        ctx.save_for_backward(x, *factors)
        ctx.net, ctx.reduce, ctx.nv, ctx.sh = net, reduce, nv, sh
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, *factors = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            g = g if ctx.reduce else g.t()  # [rows, N] gradient of the `.T` consumer: point-major underneath
            gp, gl = s3d_hip.VmBackend.features_backward(x, [f.contiguous() for f in factors[:3]],
                                                         [f.contiguous() for f in factors[3:]], ctx.net.resolution, ctx.reduce,
                                                         g.float().contiguous(), _bins(ctx.net, x, factors[:3], ctx.nv),
                                                         _source_check(ctx.net, factors), n_valid=ctx.nv, shadows=ctx.sh)
            return (None, None, None) + tuple(gp) + tuple(gl)
        if ctx.nv is not None:  # (the torch route sums over every row: the absent ones hold anything)
            live = torch.arange(x.shape[0], device=x.device) < (ctx.nv.reshape(-1)[:1] + 127) // 128 * 128
            g = torch.where(live if ctx.reduce else live.unsqueeze(0), g, torch.zeros((), dtype=g.dtype, device=g.device))
        with torch.enable_grad():
            leaves = [f.detach().requires_grad_(True) for f in factors]
            xin = x.detach().requires_grad_(ctx.needs_input_grad[0])
            fn = ctx.net._sigma_feat_torch if ctx.reduce else ctx.net._color_prod_torch
            out = fn(xin, leaves[:3], leaves[3:])
            wanted = ([xin] if ctx.needs_input_grad[0] else []) + leaves
            grads = torch.autograd.grad(out, wanted, g.contiguous())
        gx = grads[0] if ctx.needs_input_grad[0] else None
        return (gx, None, None) + tuple(grads[1:] if ctx.needs_input_grad[0] else grads)


def _shadows(net, mats, vecs, resolution):
    """rank-fastest shadows of a factor set (VmBackend.transpose_factors), taken afresh at EVERY forward: the native optimizer
    rewrites the parameters without touching `_version`, so nothing cheaper than the ~10 us launch says whether they are stale;
    the forward and the backward of one step share them (the parameters do not change in between).  None: shapes the
    16-byte loads do not cover (a rank that is not a multiple of four) or the switch `fused_shadows` off."""
    if not getattr(net, "fused_shadows", True) or any(m.shape[1] % 4 for m in mats) or not all(t.is_cuda for t in mats):
        return None
    return s3d_hip.VmBackend.transpose_factors([m.detach() for m in mats], [v.detach() for v in vecs], resolution)


def _bins(net, x, mats, n_valid=None):
    """the points of x sorted by plane tile / line chunk (VmBackend.backward_bins): x and the resolution decide it, so the
    density and the colour features of one forward share the sort.  Kept on the network, keyed by the storage of x (alive
    until both backward nodes have run), dropped at the next forward."""
    cache = net.__dict__.setdefault("_vm_bins", {})
    key = (x.data_ptr(), x._version, x.shape[0], tuple(net.resolution), None if n_valid is None else n_valid.data_ptr())
    if key not in cache:
        if len(cache) >= 4:
            cache.clear()
        # (the entry keeps x itself: while it is cached no other tensor can live at its address; two consumers — the density
        #  and the colour features' backward — then the points and their six sorted index rows are let go)
        cache[key] = [x, s3d_hip.VmBackend.backward_bins(x, [m.contiguous() for m in mats], net.resolution, n_valid), 2]
    ent = cache[key]
    ent[2] -= 1
    if ent[2] <= 0:
        del cache[key]
    return ent[1]


class _VmColorBasis(torch.autograd.Function):
    """basis_mat((mat * vec).T) of tensoRF/network.py:149-153 in ONE kernel per direction (s3d_vm_color_forward / _backward): the
    [144, N] products stay in registers, the Linear's weight in LDS; the backward derives each product's gradient from the
    Linear's output gradient on the fly and accumulates basis_mat's own gradient next to the plane gradients.  Arithmetic of the
    fp16 autocast Linear: operands rounded to binary16, fp32 accumulation, binary16 output."""

    @staticmethod
    def forward(ctx, x, net, weight, *factors):
        x = x.float().contiguous()
        mats, vecs = [f.float().contiguous() for f in factors[:3]], [f.float().contiguous() for f in factors[3:]]
        w16 = weight.detach().to(torch.float16).contiguous()
        out = torch.empty(x.shape[0], w16.shape[0], dtype=torch.float16, device=x.device)
        nv = s3d_hip.active_row_limit(x.shape[0])
        sh = _shadows(net, mats, vecs, net.resolution)
        s3d_hip.VmBackend.color_forward(x, mats, vecs, net.resolution, w16, out, n_valid=nv, shadows=sh)
        ctx.save_for_backward(x, w16, *factors)
        ctx.net, ctx.nv, ctx.sh = net, nv, sh
        return out

    @staticmethod
    def backward(ctx, g):
        x, w16, *factors = ctx.saved_tensors
        mats, vecs = [f.float().contiguous() for f in factors[:3]], [f.float().contiguous() for f in factors[3:]]
        # (g as it arrives: a [:, :27] view of _MlpInput's zero-padded [N, 32] gradient is used in place, anything else is padded)
        gp, gl, gw = s3d_hip.VmBackend.color_backward(x, mats, vecs, ctx.net.resolution, w16, g.to(torch.float16),
                                                      _bins(ctx.net, x, mats, ctx.nv), _source_check(ctx.net, factors, ctx.net.basis_mat.weight),
                                                      n_valid=ctx.nv, shadows=ctx.sh)
        return (None, None, gw) + tuple(gp) + tuple(gl)


class _MlpInput(torch.autograd.Function):
    """cat([encoder(feat), encoder_dir(d)]) of tensoRF/network.py:160-166 as the fp16 autocast Linear sees it, in one launch per
    direction (s3d_freq_encode_pack_forward / _backward): fp16 features in, one fp16 [N, ld] row out (ld = the MLP kernels' padded
    input width), instead of a cast, two encoder launches, two casts, a fill and a cat forward and five launches backward.  The
    feature gradient leaves as a [:, :D] view of a zero-padded [N, 32] fp16 buffer — the row layout the factor backward reads."""

    @staticmethod
    def forward(ctx, feat, dirs, deg1, deg2, ld):
        feat = feat.to(torch.float16).contiguous()
        dirs = dirs.float().contiguous()
        out = torch.empty(feat.shape[0], ld, dtype=torch.float16, device=feat.device)
        nv = s3d_hip.active_row_limit(feat.shape[0])
        s3d_hip.FreqBackend.freq_encode_pack_forward(feat, dirs, deg1, deg2, out, n_valid=nv)
        ctx.save_for_backward(feat)
        ctx.deg1, ctx.nv = deg1, nv
        return out

    @staticmethod
    def backward(ctx, g):
        feat, = ctx.saved_tensors
        ga = torch.empty(feat.shape[0], max(32, feat.shape[1]), dtype=torch.float16, device=feat.device)
        s3d_hip.FreqBackend.freq_encode_pack_backward(g.to(torch.float16).contiguous(), feat, ctx.deg1, ga, n_valid=ctx.nv)
        ga._s3d_zero_padded = True  # (VmBackend.color_backward takes the buffer behind the view)
        return ga[:, :feat.shape[1]], None, None, None, None


class _PackChain(torch.autograd.Function):
    """the weights of the colour MLP's bias-free Linears as the flat fp16 vector of the ffmlp kernels ([W, in_pad] | ... | [16, W]),
    one launch per direction (s3d_pack_linear_chain / s3d_unpack_linear_chain) instead of pad / cat / cast and their backward"""

    @staticmethod
    def forward(ctx, in_pad, *weights):
        ws = [w.detach().float().contiguous() for w in weights]
        padded_rows = [w.shape[0] for w in ws[:-1]] + [16]
        ld = [in_pad] + [w.shape[1] for w in ws[1:]]
        flat = torch.empty(sum(r * v for r, v in zip(padded_rows, ld)), dtype=torch.float16, device=ws[0].device)
        s3d_hip.VmBackend.pack_linear_chain(ws, padded_rows, ld, flat)
        ctx.geom = (padded_rows, ld, [tuple(w.shape) for w in ws])
        return flat

    @staticmethod
    def backward(ctx, g):
        padded_rows, ld, shapes = ctx.geom
        gs = [torch.empty(s, dtype=torch.float32, device=g.device) for s in shapes]
        s3d_hip.VmBackend.unpack_linear_chain(g.to(torch.float16).contiguous(), gs, padded_rows, ld)
        return (None,) + tuple(gs)


class _TallLinear(torch.autograd.Function):
    """y = x W^T of a bias-free nn.Linear on [N, in] rows with N >> out * in (basis_mat 144 -> 27 and the colour MLP of
    tensoRF/network.py:71-83 at ~1e5 samples per step).  Forward and data gradient are the library GEMMs nn.Linear runs under
    autocast; the WEIGHT gradient dW = G^T X has the batch as its reduction dimension and an output of a few tiles — the
    library's pick for it (one 16x16 tile per workgroup, 18 workgroups walking 1e5 rows) took 0.72 ms for basis_mat and 0.32 ms
    per 128-wide layer, a third of the training step (profiles/r09_tensorf.md).  Here the rows are cut into chunks of 1,024:
    one batched GEMM gives a partial sum per chunk (hundreds of workgroups), summed in fp32."""
    CHUNK = 1024

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float16)
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            R = _TallLinear.CHUNK
            N = x.shape[0]
            nfull = N // R
            gw = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
            if nfull:
                xc = x[:nfull * R].reshape(nfull, R, x.shape[1])
                gc = g[:nfull * R].reshape(nfull, R, g.shape[1])
                gw += torch.bmm(gc.transpose(1, 2), xc).sum(0, dtype=torch.float32)
            if nfull * R < N:
                gw += (g[nfull * R:].t() @ x[nfull * R:]).float()
            gw = gw.to(weight.dtype)
        return gx, gw


def _linear(layer, x):
    """a bias-free nn.Linear; tall batches that record a weight gradient go through _TallLinear"""
    if (layer.bias is None and x.is_cuda and x.dim() == 2 and x.shape[0] >= 8 * _TallLinear.CHUNK and torch.is_grad_enabled()
            and layer.weight.requires_grad and torch.is_autocast_enabled("cuda")):
        return _TallLinear.apply(x, layer.weight)
    return layer(x)


class _MeanAbsSum(torch.autograd.Function):
    """sum_i mean|t_i| over a list of tensors (the L1 penalty of tensoRF/network.py:259-263) with multi-tensor launches: one
    `_foreach_norm` forward, `_foreach_sign` + two `_foreach_mul_` backward — the op-by-op expression costs ~10 launches per
    tensor and direction (abs, mean, their backward, the scalar products), 0.6 ms of a 3 ms step at resolution 300."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.save_for_backward(*tensors)
        norms = torch._foreach_norm([t.detach() for t in tensors], 1)
        return torch.stack([n / t.numel() for n, t in zip(norms, tensors)]).sum()

    @staticmethod
    def backward(ctx, g):
        signs = torch._foreach_sign([t.detach() for t in ctx.saved_tensors])
        torch._foreach_mul_(signs, g.to(signs[0].dtype))
        torch._foreach_mul_(signs, [1.0 / t.numel() for t in ctx.saved_tensors])
        return tuple(signs)


class NeRFNetwork(NeRFRenderer):
    def __init__(self, resolution=(128, 128, 128), sigma_rank=(16, 16, 16), color_rank=(48, 48, 48), color_feat_dim=27,
                 num_layers=3, hidden_dim=128, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.resolution = list(resolution)
        self.sigma_rank, self.color_rank, self.color_feat_dim = list(sigma_rank), list(color_rank), color_feat_dim
        self.mat_ids = [[0, 1], [0, 2], [1, 2]]
        self.vec_ids = [2, 1, 0]
        self.sigma_mat, self.sigma_vec = self.init_one_svd(self.sigma_rank, self.resolution)
        self.color_mat, self.color_vec = self.init_one_svd(self.color_rank, self.resolution)
        self.basis_mat = nn.Linear(sum(self.color_rank), color_feat_dim, bias=False)
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        self.encoder, enc_dim = get_encoder("frequency", input_dim=color_feat_dim, multires=2)
        self.encoder_dir, enc_dim_dir = get_encoder("frequency", input_dim=3, multires=2)
        self.in_dim = enc_dim + enc_dim_dir
        dims = [self.in_dim] + [hidden_dim] * (num_layers - 1) + [3]
        self.color_net = nn.ModuleList([nn.Linear(i, o, bias=False) for i, o in zip(dims[:-1], dims[1:])])
        if self.bg_radius > 0:
            raise NotImplementedError("background model is outside the BASELINE configs")

    def __getstate__(self):
        """copy.deepcopy (teacher creation, EMA) and pickling leave the backward's sorted-point cache behind"""
        state = self.__dict__.copy()
        state.pop("_vm_bins", None)
        state.pop("_l1_inv", None)
        state.pop("_s3d_found_inf", None)  # (a trainer's scaler flag: re-attached by the trainer that owns the copy)
        return state

    def init_one_svd(self, n_component, resolution, scale=0.1):
        mat, vec = [], []
        for i, vec_id in enumerate(self.vec_ids):
            m0, m1 = self.mat_ids[i]
            mat.append(nn.Parameter(scale * torch.randn((1, n_component[i], resolution[m1], resolution[m0]))))
            vec.append(nn.Parameter(scale * torch.randn((1, n_component[i], resolution[vec_id], 1))))
        return nn.ParameterList(mat), nn.ParameterList(vec)

    def _coords(self, x):
        mat = torch.stack([x[..., ids] for ids in self.mat_ids]).view(3, -1, 1, 2)
        vec = torch.stack([x[..., i] for i in self.vec_ids])
        vec = torch.stack((torch.zeros_like(vec), vec), dim=-1).view(3, -1, 1, 2)
        return mat, vec

    def _factors(self, mats, vecs, x):
        N = x.shape[0]
        mat_coord, vec_coord = self._coords(x)
        mf = [F.grid_sample(mats[i], mat_coord[[i]], align_corners=True).view(-1, N) for i in range(3)]
        vf = [F.grid_sample(vecs[i], vec_coord[[i]], align_corners=True).view(-1, N) for i in range(3)]
        return mf, vf

    fused_vm = True  # tests / A-B runs: False = the reference's grid_sample sequence on the GPU too

    def _sigma_feat_torch(self, x, mats, vecs):
        mf, vf = self._factors(mats, vecs, x)
        out = torch.zeros([x.shape[0]], device=x.device)
        for m, v in zip(mf, vf):
            out = out + torch.sum(m * v, dim=0)
        return out

    def _color_prod_torch(self, x, mats, vecs):
        mf, vf = self._factors(mats, vecs, x)
        return torch.cat(mf, dim=0) * torch.cat(vf, dim=0)  # [3R, N]

    def _use_native(self, x):
        return self.fused_vm and x.is_cuda and x.dim() == 2 and x.shape[0] > 0

    def get_sigma_feat(self, x):
        if self._use_native(x):
            return _VmFeatures.apply(x, self, True, *self.sigma_mat, *self.sigma_vec)
        return self._sigma_feat_torch(x, self.sigma_mat, self.sigma_vec)

    fused_basis = True  # A-B runs: False = the products through HBM and basis_mat as an nn.Linear

    def get_color_feat(self, x):
        if self._use_native(x):
            if (self.fused_basis and not x.requires_grad and self.basis_mat.bias is None and self.basis_mat.out_features <= 32
                    and sum(self.color_rank) <= 512 and max(self.color_rank) <= 64 and torch.is_autocast_enabled("cuda")
                    and torch.get_autocast_dtype("cuda") == torch.float16):
                return _VmColorBasis.apply(x, self, self.basis_mat.weight, *self.color_mat, *self.color_vec)
            return _linear(self.basis_mat, _VmFeatures.apply(x, self, False, *self.color_mat, *self.color_vec).T)
        return _linear(self.basis_mat, self._color_prod_torch(x, self.color_mat, self.color_vec).T)

    def _normalize(self, x):
        if (self.fused_vm and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and not x.requires_grad
                and self.aabb_train.is_cuda and self.aabb_train.dtype == torch.float32):
            out = torch.empty_like(x)  # (one launch instead of five, the same operations in the same order)
            s3d_hip.VmBackend.aabb_normalize(x, self.aabb_train.contiguous(), out)
            return out
        return 2 * (x - self.aabb_train[:3]) / (self.aabb_train[3:] - self.aabb_train[:3]) - 1

    def forward(self, x, d):
        x = self._normalize(x)
        self.__dict__["_vm_bins"] = {}  # (the previous forward's sorted points)
        sigma = trunc_exp(self.get_sigma_feat(x))
        cf = self.get_color_feat(x)
        if self._fused_mlp_ok(cf, d.requires_grad):
            return sigma, self._color_mlp_fused(cf, d)
        feat, dirs = self.encoder(cf), self.encoder_dir(d)
        h = torch.cat([feat, dirs], dim=-1)
        for k, layer in enumerate(self.color_net):
            h = _linear(layer, h)
            if k != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        return sigma, torch.sigmoid(h)

    fused_mlp = True  # A-B runs / tests: False = the nn.Linear chain (library GEMMs + elementwise launches)

    def _fused_mlp_ok(self, feat, d_requires_grad=False):
        """the colour MLP (tensoRF/network.py:71-83: bias-free Linear -> ReLU chain, 150 -> 128 -> 128 -> 3) on the MFMA kernels
        of the ffmlp package: hidden width 128, input padded to a multiple of 16 (<= 160), whole 128-row tiles, fp16 autocast"""
        net = self.color_net
        return (self.fused_mlp and feat.is_cuda and feat.dim() == 2 and feat.shape[0] > 0 and feat.shape[0] % 128 == 0
                and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
                and len(net) >= 2 and all(l.bias is None for l in net) and self.hidden_dim == 128
                and net[-1].out_features <= 16 and (self.in_dim + 15) // 16 * 16 <= 160
                and hasattr(self.encoder, "degree") and hasattr(self.encoder_dir, "degree") and not d_requires_grad)

    def _color_mlp_fused(self, cf, d):
        """the colours: same arithmetic as the encoders + Linear chain + sigmoid under autocast (fp32 encodings rounded to fp16 once,
        fp16 operands, fp32 accumulation, fp16 activations); the weights travel as the ffmlp layout [W, in_pad] | (n - 1) x [W, W] | [16, W] built from
        the nn.Linear parameters each step (_PackChain: ~39 K elements, one launch per direction)"""
        from ffmlp.ffmlp import _FFMLPForward
        net = self.color_net
        in_pad, out = (self.in_dim + 15) // 16 * 16, net[-1].out_features
        h = _MlpInput.apply(cf, d, self.encoder.degree, self.encoder_dir.degree, in_pad)
        flat = _PackChain.apply(in_pad, *[l.weight for l in net])
        # (the kernels write the output padded to 16 columns, ffmlp.py:117-118, 162-163)
        # (no graph being recorded — rendering: the inference entry point, no activation buffers)
        nv = s3d_hip.active_row_limit(cf.shape[0])  # (padded sample batch: the MLP kernels skip the absent rows too)
        out16 = _FFMLPForward.apply(h, flat, in_pad, 16, self.hidden_dim, len(net) - 1, 0, 6, not torch.is_grad_enabled(), True,
                                    None, None, 0, nv)
        if out == 3:
            # sigmoid of the three real columns as fp32 with torch.sigmoid's fp16 rounding, one launch per direction
            # (nerf/network_ff.py: _NgpRgb — instead of slice, sigmoid, cast and their four backward launches)
            from nerf.network_ff import _NgpRgb
            return _NgpRgb.apply(out16, nv)
        return torch.sigmoid(out16[:, :out])

    def density(self, x):
        return {"sigma": trunc_exp(self.get_sigma_feat(self._normalize(x)))}

    def density_loss(self):
        """L1 penalty on the density factors (tensoRF/network.py:259-263): sum over the three plane / line pairs of
        mean|sigma_mat| + mean|sigma_vec|; the trainer adds it times `l1_reg_weight` (tensoRF/utils.py:42-49)"""
        if self.fused_l1:
            return _MeanAbsSum.apply(*self.sigma_mat, *self.sigma_vec)
        loss = 0
        for m, v in zip(self.sigma_mat, self.sigma_vec):
            loss = loss + torch.mean(torch.abs(m)) + torch.mean(torch.abs(v))
        return loss

    fused_l1 = True  # tests / A-B runs: False = the reference's op-by-op expression

    @torch.no_grad()
    def density_loss_value(self):
        """the value of density_loss() without a graph (tensoRF/utils.py: the trainer whose optimizer forms the gradient itself)"""
        ts = [t.detach() for t in list(self.sigma_mat) + list(self.sigma_vec)]
        if self.fused_l1 and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts) and len(ts) <= 8:
            out = torch.empty((), dtype=torch.float32, device=ts[0].device)  # (two launches instead of eight)
            s3d_hip.VmBackend.weighted_abs_sum(ts, [1.0 / t.numel() for t in ts], out)
            return out
        key = tuple(t.numel() for t in ts)
        inv = self.__dict__.get("_l1_inv")
        if inv is None or inv[0] != key or inv[1].device != ts[0].device:
            inv = self.__dict__["_l1_inv"] = (key, torch.tensor([1.0 / n for n in key], dtype=torch.float32, device=ts[0].device))
        return (torch.stack(torch._foreach_norm(ts, 1)) * inv[1]).sum()

    def get_params(self, lr1, lr2=None):
        lr2 = lr1 if lr2 is None else lr2
        return [{"params": self.sigma_mat, "lr": lr1}, {"params": self.sigma_vec, "lr": lr1},
                {"params": self.color_mat, "lr": lr1}, {"params": self.color_vec, "lr": lr1},
                {"params": self.basis_mat.parameters(), "lr": lr2}, {"params": self.color_net.parameters(), "lr": lr2}]

    @torch.no_grad()
    def upsample_model(self, resolution):
        """bilinear re-sampling of all factors to a new resolution (tensoRF/network.py:266-318)"""
        def up(mats, vecs):
            for i, vec_id in enumerate(self.vec_ids):
                m0, m1 = self.mat_ids[i]
                mats[i] = nn.Parameter(F.interpolate(mats[i].data, size=(resolution[m1], resolution[m0]), mode="bilinear", align_corners=True))
                vecs[i] = nn.Parameter(F.interpolate(vecs[i].data, size=(resolution[vec_id], 1), mode="bilinear", align_corners=True))
        up(self.sigma_mat, self.sigma_vec)
        up(self.color_mat, self.color_vec)
        self.resolution = list(resolution)

    @torch.no_grad()
    def shrink_model(self):
        """crop aabb_train and every factor to the bounding box of the occupied cells of the coarsest density grid
        (tensoRF/network.py:273-318).  Returns (tl, br): the kept index range per axis.  The parameter set changes:
        re-create the optimizer afterwards (Trainer.rebuild_optimizer), as after upsample_model."""
        import raymarching
        hgs = self.bound / self.grid_size
        thresh = min(self.density_thresh, self.mean_density)
        occupied = self.density_grid[self.cascade - 1] > thresh
        pos = raymarching.morton3D_invert(torch.nonzero(occupied).squeeze(-1))
        if pos.shape[0] == 0:
            raise RuntimeError("shrink_model: no cell of the coarsest cascade is above the density threshold")
        pos = (2 * pos / (self.grid_size - 1) - 1) * (self.bound - hgs)
        min_pos, max_pos = pos.amin(0) - hgs, pos.amax(0) + hgs
        reso = torch.tensor(self.resolution, dtype=torch.long, device=self.aabb_train.device)
        units = (self.aabb_train[3:] - self.aabb_train[:3]) / reso
        tl = torch.round((min_pos - self.aabb_train[:3]) / units).long().clamp(min=0)
        br = torch.minimum(torch.round((max_pos - self.aabb_train[:3]) / units).long(), reso)
        tl_h, br_h = tl.tolist(), br.tolist()
        for i, vec_id in enumerate(self.vec_ids):
            m0, m1 = self.mat_ids[i]
            for vecs, mats in ((self.sigma_vec, self.sigma_mat), (self.color_vec, self.color_mat)):
                vecs[i] = nn.Parameter(vecs[i].data[..., tl_h[vec_id]:br_h[vec_id], :].contiguous())
                mats[i] = nn.Parameter(mats[i].data[..., tl_h[m1]:br_h[m1], tl_h[m0]:br_h[m0]].contiguous())
        self.aabb_train = torch.cat([min_pos, max_pos], dim=0).to(self.aabb_train.dtype)
        # (the reference leaves `self.resolution` stale after the crop — it only reads it again in the next upsample; the
        #  fused feature kernels take the factor extents from it, so it follows the parameters here)
        self.resolution = [b - a for a, b in zip(tl_h, br_h)]
        return tl_h, br_h
