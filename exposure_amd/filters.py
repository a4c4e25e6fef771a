"""The reference's ``Filter`` plugin protocol on PyTorch-ROCm, backed by the HIP library.

Mirrors ``/root/reference/filters.py`` class-for-class: same class names, method names,
arguments, parameter tensor shapes and return triple, so a torch restatement of
``agent_generator`` can run ``filter.apply(net, filter_features, high_res=high_res)``
(``agent.py:67-68``) unchanged.  What differs is where the work happens:

* ``process(img, param)`` does not compose elementwise tensor ops; it makes ONE call into
  ``libexposure_hip.so`` (``expo_filter_fwd``), and its autograd backward makes one call
  (``expo_filter_bwd``) that returns the image gradient and the per-image parameter
  gradients in a single pass.  There is no CPU / eager fallback: a CPU tensor raises.
* parameter regression (two tiny FCs + range squashing, ``filters.py:28-44`` and the
  per-class ``filter_param_regressor``) stays in torch -- it is (N x 4096) GEMM work that
  hipBLASLt handles, not pixel work.

Images are NHWC (as in the reference), float16 or float32, on a ROCm device.
"""
import math

import torch
from torch import nn

from . import _cabi
from .util import lerp, lrelu, tanh_range

FILTER_SHORT_NAMES = ('E', 'G', 'W', 'S+', 'T', 'Ct', 'BW', 'C')


class _PixelFilterFunction(torch.autograd.Function):
  """y = process_<fid>(img, packed) through the C-ABI; backward = expo_filter_bwd."""

  @staticmethod
  def forward(ctx, img, packed, fid, hsv_grad_mode):
    img = img.contiguous()
    packed = packed.contiguous().float()
    y = torch.empty_like(img)
    _cabi.filter_fwd(fid, img, y, packed)
    ctx.save_for_backward(img, packed)
    ctx.fid = fid
    ctx.hsv_grad_mode = hsv_grad_mode
    return y

  @staticmethod
  def backward(ctx, dy):
    img, packed = ctx.saved_tensors
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dparams = torch.empty_like(packed)
    _cabi.filter_bwd(ctx.fid, img, dy, dx, packed, dparams, ctx.hsv_grad_mode)
    return dx, dparams, None, None


class _PixelFilterPairFunction(torch.autograd.Function):
  """One filter, one parameter tensor, TWO images: the 64x64 proxy and the full-resolution image of
  ``Filter.apply(img, ..., high_res=...)`` (filters.py:88-96).  As one autograd node the parameter gradient is
  produced ONCE: the proxy's pass overwrites it and the full-resolution pass adds to it in the finish launch
  (``expo_filter_bwd_accumulate``) -- instead of two nodes whose (N, P) results autograd adds afterwards."""

  @staticmethod
  def forward(ctx, img, high, packed, fid, hsv_grad_mode):
    img, high = img.contiguous(), high.contiguous()
    packed = packed.contiguous().float()
    y, yh = torch.empty_like(img), torch.empty_like(high)
    _cabi.filter_fwd(fid, img, y, packed)
    _cabi.filter_fwd(fid, high, yh, packed)
    ctx.save_for_backward(img, high, packed)
    ctx.fid = fid
    ctx.hsv_grad_mode = hsv_grad_mode
    ctx.set_materialize_grads(False)  # an output nobody differentiates costs no launch
    return y, yh

  @staticmethod
  def backward(ctx, dy, dyh):
    img, high, packed = ctx.saved_tensors
    dparams = None
    grads = []
    for x, g, need in ((img, dy, ctx.needs_input_grad[0]), (high, dyh, ctx.needs_input_grad[1])):
      if g is None:
        grads.append(None)
        continue
      g = g.contiguous().to(x.dtype)
      dx = torch.empty_like(x) if need else None
      first = dparams is None
      if first:
        dparams = torch.empty_like(packed)
      _cabi.filter_bwd(ctx.fid, x, g, dx, packed, dparams, ctx.hsv_grad_mode, accumulate=not first)
      grads.append(dx)
    return grads[0], grads[1], dparams, None, None


class _OverexposurePenaltyFunction(torch.autograd.Function):
  """mean_{h,w,c} max(y - 1, 0)^2 per image (agent.py:249-251) -- expo_overexposure_penalty / _bwd.  The agent's
  default path gets this from the fused dispatch pass; this node serves the paths that cannot (cfg.masking,
  cfg.clamp: the penalty is then taken on an image no dispatch kernel produced)."""

  @staticmethod
  def forward(ctx, y):
    y = y.contiguous()
    pen = torch.empty((y.shape[0],), dtype=torch.float32, device=y.device)
    _cabi.overexposure_penalty(y, pen)
    ctx.save_for_backward(y)
    return pen

  @staticmethod
  def backward(ctx, dpen):
    y, = ctx.saved_tensors
    dy = torch.empty_like(y)
    _cabi.overexposure_penalty_bwd(y, dpen.contiguous().float(), dy)
    return dy


def overexposure_penalty(y):
  """(N,) float32: ``reduce_mean(maximum(net - 1, 0)**2, axis=(1, 2, 3))`` of agent.py:249-251, differentiable."""
  return _OverexposurePenaltyFunction.apply(y)


class _MaskedApplyFunction(torch.autograd.Function):
  """out = lerp(img, process(img, packed), mask(img, mask_params)) -- expo_filter_apply_fwd/bwd."""

  @staticmethod
  def forward(ctx, img, packed, mask_params, fid, sharp, min_strength, hsv_grad_mode):
    img = img.contiguous()
    packed = packed.contiguous().float()
    mask_params = mask_params.contiguous().float()
    y = torch.empty_like(img)
    _cabi.apply_fwd(fid, img, y, packed, mask_params, sharp, min_strength)
    ctx.save_for_backward(img, packed, mask_params)
    ctx.args = (fid, sharp, min_strength, hsv_grad_mode)
    return y

  @staticmethod
  def backward(ctx, dy):
    img, packed, mask_params = ctx.saved_tensors
    fid, sharp, min_strength, hsv_grad_mode = ctx.args
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dparams = torch.empty_like(packed)
    dmask = torch.empty_like(mask_params)
    _cabi.apply_bwd(fid, img, dy, dx, packed, dparams, mask_params, dmask, sharp, min_strength, hsv_grad_mode)
    return dx, dparams, dmask, None, None, None, None


class _CurveFunction(torch.autograd.Function):
  """Tone / Color with a step count the tuned kernels are not built for (cfg.curve_steps != 8):
  expo_curve_fwd / expo_curve_bwd, one element-wise pass per direction."""

  @staticmethod
  def forward(ctx, img, packed, curves, steps):
    img = img.contiguous()
    packed = packed.contiguous().float()
    y = torch.empty_like(img)
    _cabi.curve_fwd(img, y, packed, curves, steps)
    ctx.save_for_backward(img, packed)
    ctx.args = (curves, steps)
    return y

  @staticmethod
  def backward(ctx, dy):
    img, packed = ctx.saved_tensors
    curves, steps = ctx.args
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dparams = torch.empty_like(packed)
    _cabi.curve_bwd(img, dy, dx, packed, dparams, curves, steps)
    return dx, dparams, None, None


class _HeadsRegressSelect(torch.autograd.Function):
  """params24[n] = filter_param_regressor of the head image n selected, applied to that head's raw features, zero padded
  to EXPO_MAX_PARAMS -- ``expo_heads_regress_fwd / _bwd``: the eight regressors and the one-hot gather
  (agent.py:58-77, 119-125) as ONE launch each way instead of ~70 forward and ~75 backward torch launches."""

  @staticmethod
  def forward(ctx, selected, abi_ids, ranges, *raws):
    raws = tuple(r.contiguous().float() for r in raws)
    n = raws[0].shape[0]
    params = torch.empty((n, _cabi.EXPO_MAX_PARAMS), dtype=torch.float32, device=raws[0].device)
    selected = selected.contiguous().to(torch.int32)
    _cabi.heads_regress_fwd(raws, abi_ids, ranges, selected, params)
    ctx.save_for_backward(selected, *raws)
    ctx.meta = (tuple(abi_ids), tuple(ranges))
    return params

  @staticmethod
  def backward(ctx, dparams):
    selected, *raws = ctx.saved_tensors
    abi_ids, ranges = ctx.meta
    n = raws[0].shape[0]
    # one flat buffer, head-major: every head's gradient is a CONTIGUOUS view (its FC's backward GEMM takes it as is)
    widths = [r.shape[1] for r in raws]
    flat = torch.empty((n * sum(widths),), dtype=torch.float32, device=raws[0].device)
    draws, at = [], 0
    for w in widths:
      draws.append(flat[at:at + n * w].view(n, w))
      at += n * w
    _cabi.heads_regress_bwd(raws, draws, abi_ids, ranges, selected, dparams.contiguous().float())
    return (None, None, None) + tuple(draws)


def heads_regress_select(filter_modules, raws, selected):
  """The fused tail of the FC heads (see _HeadsRegressSelect).  ``filter_modules``: the agent's Filter list (their
  ``filter_id`` and ``cfg`` ranges), ``raws``: each head's second-FC output (N, P_j + 6), ``selected`` (N,) int32."""
  cfg = filter_modules[0].cfg
  tl, tr = cfg.tone_curve_range
  cl, cr = cfg.color_curve_range
  bias = lambda l, r, initial: math.atanh(2 * (initial - l) / (r - l) - 1) if initial is not None else 0.0
  ranges = (float(cfg.exposure_range), math.log(cfg.gamma_range), float(tl), float(tr), 0.0, float(cl), float(cr),
            bias(cl, cr, 1), bias(-cfg.exposure_range, cfg.exposure_range, 0))
  return _HeadsRegressSelect.apply(selected, tuple(int(f.filter_id) for f in filter_modules), ranges, *raws)


class _PackedHeadsFn(torch.autograd.Function):
  """All heads' ``fc2(lrelu(fc1(features)))`` (filters.py:28-42, eight times per step: agent.py:58-69) as ONE GEMM for
  the first layers -- features (N, F) x packed W1^T (F, K*H) --, one fused bias + lrelu launch, ONE batched GEMM for the
  second layers (K x (N, H) x (H, 32): outputs zero-padded to 32 columns), and the transposed pair in the backward:
  ~11 launches instead of ~90.  The packed operands ALIAS the heads' own parameters (``PackedHeads``), so nothing is
  gathered per step; the parameters enter as inputs only so that autograd routes the gradient slices to them."""

  @staticmethod
  def forward(ctx, features, pack, *leaves):
    n = features.shape[0]
    k, hid, pad = pack.k, pack.hidden, pack.PAD
    features = features.contiguous()
    z1 = features @ pack.w1.t()  # (N, K*H)
    h = torch.empty_like(z1)
    _cabi.bias_lrelu_fwd(z1, pack.b1, h, 0.2)
    h3 = h.view(n, k, hid).transpose(0, 1)  # (K, N, H): batch stride H, row stride K*H -- no copy
    out = torch.baddbmm(pack.b2.unsqueeze(1), h3, pack.w2.transpose(1, 2))  # (K, N, 32)
    ctx.save_for_backward(features, h)
    ctx.pack = pack
    return tuple(out.unbind(0))

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, *douts):
    features, h = ctx.saved_tensors
    pack = ctx.pack
    n = features.shape[0]
    k, hid, pad = pack.k, pack.hidden, pack.PAD
    first = douts[0]
    adjacent = all(d is not None and d.is_contiguous() and d.data_ptr() == first.data_ptr() + j * n * pad * 4
                   for j, d in enumerate(douts))
    if adjacent:  # the regress-and-select node hands over K slices of ONE head-major buffer (filters._HeadsRegressSelect)
      dout = torch.as_strided(first, (k, n, pad), (n * pad, pad, 1))
    else:  # any other consumer: heads nobody used arrive as None
      like = next(d for d in douts if d is not None)
      dout = torch.stack([d if d is not None else torch.zeros_like(like) for d in douts], dim=0)
    h3 = h.view(n, k, hid).transpose(0, 1)
    db2 = dout.sum(dim=1)  # (K, 32)
    dw2 = torch.bmm(dout.transpose(1, 2), h3)  # (K, 32, H)
    dh = torch.bmm(dout, pack.w2).transpose(0, 1).reshape(n, k * hid)  # (N, K*H): one small transposing copy
    dz1 = torch.empty_like(dh)
    _cabi.lrelu_bwd(h, dh, dz1, 0.2)
    db1 = dz1.sum(dim=0)
    dw1 = dz1.t() @ features  # (K*H, F)
    dfeat = dz1 @ pack.w1 if ctx.needs_input_grad[0] else None
    grads = []
    for j, width in enumerate(pack.widths):
      grads += [dw1[j * hid:(j + 1) * hid], db1[j * hid:(j + 1) * hid], dw2[j, :width], db2[j, :width]]
    return (dfeat, None) + tuple(grads)


class PackedHeads:
  """Storage of the agent's K filter heads packed so that their first layers are ONE weight matrix (K*H, F) and their
  second layers ONE zero-padded batch (K, 32, H): every ``filter.fc1.weight`` / ``.bias`` / ``fc2.weight`` / ``.bias``
  keeps its identity as a Parameter (names, state dict, optimiser, gradient buckets, checkpoint import all unchanged)
  but its ``.data`` is a view into the packed buffers -- the parameter-flattening trick of data-parallel wrappers.
  ``ensure()`` re-packs if anything replaced the parameters' storage (``module.to(...)``, a dtype change)."""
  PAD = 32

  def __init__(self, filter_modules):
    self.filters = list(filter_modules)
    self.k = len(self.filters)
    self.hidden = self.filters[0].fc1.out_features
    self.in_dim = self.filters[0].fc1.in_features
    self.widths = [f.fc2.out_features for f in self.filters]
    self.w1 = self.b1 = self.w2 = self.b2 = None
    self.generation = 0  # bumped by every (re)pack: holders of pointers into the packed buffers compare it

  def supported(self):
    return (all(f.fc1.out_features == self.hidden and f.fc1.in_features == self.in_dim and
                f.fc2.in_features == self.hidden and f.fc2.out_features <= self.PAD for f in self.filters) and
            all(p.dtype == torch.float32 for p in self.leaves()))

  def leaves(self):
    out = []
    for f in self.filters:
      out += [f.fc1.weight, f.fc1.bias, f.fc2.weight, f.fc2.bias]
    return out

  def _aliased(self):
    if self.w1 is None or self.w1.device != self.filters[0].fc1.weight.device:
      return False
    hid, f_in = self.hidden, self.in_dim
    for j, f in enumerate(self.filters):
      if (f.fc1.weight.data_ptr() != self.w1.data_ptr() + 4 * j * hid * f_in or
          f.fc1.bias.data_ptr() != self.b1.data_ptr() + 4 * j * hid or
          f.fc2.weight.data_ptr() != self.w2.data_ptr() + 4 * j * self.PAD * hid or
          f.fc2.bias.data_ptr() != self.b2.data_ptr() + 4 * j * self.PAD):
        return False
    return True

  def quick_aliased(self):
    """The first and the last leaf still live in the packed buffers (two pointer comparisons: what a graph-replaying
    caller can afford per step; ``module.to()`` / a dtype change / a restore that replaces storages moves them all)."""
    if self.w1 is None:
      return False
    first, last = self.filters[0].fc1.weight, self.filters[-1].fc2.bias
    return (first.data_ptr() == self.w1.data_ptr() and
            last.data_ptr() == self.b2.data_ptr() + 4 * (self.k - 1) * self.PAD)

  @torch.no_grad()
  def ensure(self):
    """Packs on first use; packs AGAIN when the parameters were re-allocated since.  A re-pack moves every head
    parameter to new storage: hipGraphs captured before keep reading and updating the OLD buffers -- whoever captured
    them must compare ``generation`` (``gan.GAN._check_heads`` drops its graphs); inside a capture it is refused."""
    if self._aliased():
      return
    if self.filters[0].fc1.weight.is_cuda and torch.cuda.is_current_stream_capturing():
      raise RuntimeError('exposure_amd: the filter heads\' parameters were re-allocated (module.to(), a dtype change, a '
                         'restore that replaces storages) and would have to be re-packed INSIDE a hipGraph capture; run '
                         'one eager step (or Agent.pack_heads()) first')
    self.generation += 1
    dev = self.filters[0].fc1.weight.device
    k, hid, f_in, pad = self.k, self.hidden, self.in_dim, self.PAD
    w1 = torch.empty((k * hid, f_in), dtype=torch.float32, device=dev)
    b1 = torch.empty((k * hid,), dtype=torch.float32, device=dev)
    w2 = torch.zeros((k, pad, hid), dtype=torch.float32, device=dev)
    b2 = torch.zeros((k, pad), dtype=torch.float32, device=dev)
    for j, f in enumerate(self.filters):
      width = self.widths[j]
      for view, prm in ((w1[j * hid:(j + 1) * hid], f.fc1.weight), (b1[j * hid:(j + 1) * hid], f.fc1.bias),
                        (w2[j, :width], f.fc2.weight), (b2[j, :width], f.fc2.bias)):
        view.copy_(prm.data)
        prm.data = view
    self.w1, self.b1, self.w2, self.b2 = w1, b1, w2, b2

  def __call__(self, features):
    """-> K tensors (N, 32): head j's second-FC output in its first ``widths[j]`` columns, zeros behind."""
    self.ensure()
    return list(_PackedHeadsFn.apply(features, self, *self.leaves()))


def pixel_filter(fid, img, packed, hsv_grad_mode=0):
  """Functional entry: filter ``fid`` (0..7, ``cfg.filters`` order) with packed (N,P) params."""
  return _PixelFilterFunction.apply(img, packed, fid, hsv_grad_mode)


def _shape_hwc(net):
  shape = tuple(net.shape) if hasattr(net, 'shape') else tuple(net)
  return tuple(int(s) for s in shape[1:])


class Filter(nn.Module):
  """filters.py:9-167.  ``net`` is an NHWC tensor (or its shape); ``cfg`` the config Dict."""

  filter_id = None

  def __init__(self, net, cfg):
    super().__init__()
    self.cfg = cfg
    self.height, self.width, self.channels = _shape_hwc(net)
    # Specified in child classes
    self.num_filter_parameters = None
    self.short_name = None
    self.filter_parameters = None
    self.fc1 = None
    self.fc2 = None

  def _build_regressor(self):
    """The two FCs of extract_parameters (filters.py:28-42), Xavier-initialised."""
    in_dim = self.cfg.feature_extractor_dims
    out_dim = self.get_num_filter_parameters() + self.get_num_mask_parameters()
    self.fc1 = nn.Linear(in_dim, self.cfg.fc1_size)
    self.fc2 = nn.Linear(self.cfg.fc1_size, out_dim)
    for fc in (self.fc1, self.fc2):
      nn.init.xavier_uniform_(fc.weight)
      nn.init.zeros_(fc.bias)

  def get_short_name(self):
    assert self.short_name
    return self.short_name

  def get_num_filter_parameters(self):
    assert self.num_filter_parameters
    return self.num_filter_parameters

  def extract_parameters(self, features):
    """filters.py:28-44."""
    features = lrelu(self.fc1(features))
    features = self.fc2(features)
    p = self.get_num_filter_parameters()
    return features[:, :p], features[:, p:]

  def filter_param_regressor(self, features):
    """filters.py:46-48: every filter class squashes its raw features into its parameter range here."""
    assert False

  def pack(self, param):
    """reference-shaped parameter tensor -> (N, P) float32 C-ABI layout."""
    return param.reshape(param.shape[0], self.get_num_filter_parameters())

  def process(self, img, param):
    """filters.py:50-53 (the unmasked pixel map): ONE HIP launch for the filter's id (``pixel_filter``)."""
    hsv_mode = int(self.cfg.get('hsv_grad_mode', 0)) if hasattr(self.cfg, 'get') else 0
    return pixel_filter(self.filter_id, img, self.pack(param), hsv_mode)

  def debug_info_batched(self):
    return False

  def no_high_res(self):
    return False

  def apply(self, img, img_features=None, specified_parameter=None, high_res=None):
    """filters.py:62-99 -> (low_res_output, high_res_output or None, debug_info).

    The reference's method name shadows ``nn.Module.apply(fn)``; a callable first argument (what
    ``module.apply(init_fn)`` passes when it recurses into the children of an Agent / GAN) is routed
    to the module protocol instead of tripping the assertion below."""
    if callable(img) and not isinstance(img, torch.Tensor) and img_features is None and \
        specified_parameter is None and high_res is None:
      return nn.Module.apply(self, img)
    assert (img_features is None) ^ (specified_parameter is None)
    if img_features is not None:
      filter_features, mask_parameters = self.extract_parameters(img_features)
      filter_parameters = self.filter_param_regressor(filter_features)
    else:
      assert not self.use_masking()
      filter_parameters = specified_parameter
      # the reference broadcasts a (1, P...) parameter over the batch (``param[:, None, None, :]``)
      if filter_parameters.shape[0] == 1 and img.shape[0] > 1:
        filter_parameters = filter_parameters.expand(img.shape[0], *filter_parameters.shape[1:])
      mask_parameters = torch.zeros((1, self.get_num_mask_parameters()), dtype=torch.float32,
                                    device=img.device)
    debug_info = {}
    if self.debug_info_batched():  # (otherwise the first image's parameters only, as in the reference's dict)
      debug_info['filter_parameters'] = filter_parameters
    else:
      debug_info['filter_parameters'] = filter_parameters[0]
    self.mask_parameters = mask_parameters
    if not self.use_masking():
      self.mask = self.get_mask(img, mask_parameters)
      debug_info['mask'] = self.mask[0]
      # lerp(img, process(img, p), ones(1,1,1,1)) == process(img, p): the constant-one mask of the
      # shipped configs (cfg.masking = False) is folded away instead of spending two more passes.
      apply_one = lambda im: self.process(im, filter_parameters)
      if high_res is not None and not self.no_high_res() and not self.uses_generic_kernels():
        # proxy + full-resolution image share the parameters: ONE autograd node, ONE parameter gradient
        hsv_mode = int(self.cfg.get('hsv_grad_mode', 0)) if hasattr(self.cfg, 'get') else 0
        low_res_output, high_res_output = _PixelFilterPairFunction.apply(img, high_res, self.pack(filter_parameters),
                                                                         self.filter_id, hsv_mode)
        return low_res_output, high_res_output, debug_info
    else:
      # masking on: mask evaluation, process() and the lerp are ONE kernel (expo_filter_apply_fwd)
      mp = tanh_range(-5, 5, initial=0)(mask_parameters)  # filters.py:121-123
      self.mask = self.get_mask(img[:1], mask_parameters[:1])  # debug output (first image) only
      debug_info['mask'] = self.mask[0]
      packed = self.pack(filter_parameters)
      hsv_mode = int(self.cfg.get('hsv_grad_mode', 0))
      apply_one = lambda im: _MaskedApplyFunction.apply(im, packed, mp, self.filter_id, float(
          self.cfg.maximum_sharpness), float(self.cfg.minimum_strength), hsv_mode)
      if self.uses_generic_kernels():
        # a curve filter with cfg.curve_steps != 8: process() on the generic kernel, mask and lerp as tensor ops
        # (filters.py:86-88 literally)
        apply_one = lambda im: lerp(im.float(), self.process(im, filter_parameters).float(),
                                    self.get_mask(im.float(), mask_parameters)).to(im.dtype)
    low_res_output = apply_one(img)
    if high_res is not None:
      if self.no_high_res():
        high_res_output = high_res
      else:
        high_res_output = apply_one(high_res)
    else:
      high_res_output = None
    return low_res_output, high_res_output, debug_info

  # nn.Module.apply(fn) is shadowed by the reference's method name on purpose; keep access to it.
  module_apply = nn.Module.apply

  def use_masking(self):
    return self.cfg.masking

  def uses_generic_kernels(self):
    """True for a filter whose configuration the tuned per-id kernels are not built for (Tone / Color with
    cfg.curve_steps != 8): its process() runs on expo_curve_*, and the agent selects by stack-and-reduce."""
    return False

  def get_num_mask_parameters(self):
    return 6

  def get_mask(self, img, mask_parameters):
    """filters.py:110-148 as a tensor.  With masking on, apply() does NOT use this (mask, process
    and lerp are fused in expo_filter_apply_fwd); it only feeds debug_info['mask'] / visualisers."""
    if not self.use_masking():
      return torch.ones((1, 1, 1, 1), dtype=torch.float32, device=img.device)
    filter_input_range = 5
    assert mask_parameters.shape[1] == self.get_num_mask_parameters()
    mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
    h, w = int(img.shape[1]), int(img.shape[2])
    se = min(h, w)
    gi = ((torch.arange(h, dtype=torch.float64) + (se - h) / 2.0) / se - 0.5).float().to(img.device)
    gj = ((torch.arange(w, dtype=torch.float64) + (se - w) / 2.0) / se - 0.5).float().to(img.device)
    from .util import rgb2lum
    inp = gi[None, :, None, None] * mp[:, None, None, 0, None] + gj[None, None, :, None] * mp[:, None, None, 1, None] + \
        mp[:, None, None, 2, None] * (rgb2lum(img.float()) - 0.5) + mp[:, None, None, 3, None] * 2
    inp = inp * (self.cfg.maximum_sharpness * mp[:, None, None, 4, None] / filter_input_range)
    mask = torch.sigmoid(inp)
    return mask * (mp[:, None, None, 5, None] / filter_input_range * 0.5 + 0.5) * \
        (1 - self.cfg.minimum_strength) + self.cfg.minimum_strength

  def visualize_filter(self, debug_info, canvas):
    raise NotImplementedError('cv2 drawing is out of scope')

  def visualize_mask(self, debug_info, res):
    raise NotImplementedError('cv2 drawing is out of scope')


class ExposureFilter(Filter):
  """filters.py:170-182."""
  filter_id = 0

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'E'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    return tanh_range(-self.cfg.exposure_range, self.cfg.exposure_range, initial=0)(features)


class GammaFilter(Filter):
  """filters.py:194-206."""
  filter_id = 1

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'G'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    log_gamma_range = math.log(self.cfg.gamma_range)
    return torch.exp(tanh_range(-log_gamma_range, log_gamma_range)(features))


class ImprovedWhiteBalanceFilter(Filter):
  """filters.py:215-238."""
  filter_id = 2

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'W'
    self.channels = 3
    self.num_filter_parameters = self.channels
    self._build_regressor()
    # filters.py:226-229: the first feature is masked out (a device buffer: no H2D copy per call,
    # which would also be illegal inside a hipGraph capture)
    self.register_buffer('feature_mask', torch.tensor([[0.0, 1.0, 1.0]]), persistent=False)

  def filter_param_regressor(self, features):
    log_wb_range = 0.5
    features = features * self.feature_mask.to(features.dtype)
    color_scaling = torch.exp(tanh_range(-log_wb_range, log_wb_range)(features))
    # normalize by luminance
    color_scaling = color_scaling * (1.0 / (1e-5 + 0.27 * color_scaling[:, 0] + 0.67 * color_scaling[:, 1] +
                                            0.06 * color_scaling[:, 2]))[:, None]
    return color_scaling


# the curve step count the streaming / dispatch / fused kernels are instantiated for (config_example.py:27, config_sintel.py:28);
# any other cfg.curve_steps runs the generic element-wise kernels (expo_curve_*) and the agent's stack-and-select path
CURVE_STEPS_TUNED = 8


class ColorFilter(Filter):
  """filters.py:247-273 (north_star calls it ColorCurveFilter)."""
  filter_id = 7

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.curve_steps = int(cfg.curve_steps)
    assert 1 <= self.curve_steps <= _cabi.EXPO_CURVE_MAX_STEPS, 'cfg.curve_steps must be in [1, 16]'
    self.short_name = 'C'
    self.num_filter_parameters = self.channels * cfg.curve_steps
    self._build_regressor()

  def uses_generic_kernels(self):
    return self.curve_steps != CURVE_STEPS_TUNED

  def process(self, img, param):
    if self.curve_steps == CURVE_STEPS_TUNED:
      return Filter.process(self, img, param)
    return _CurveFunction.apply(img, self.pack(param), 3, self.curve_steps)

  def filter_param_regressor(self, features):
    color_curve = features.reshape(-1, self.channels, self.cfg.curve_steps)[:, None, None, :]
    return tanh_range(*self.cfg.color_curve_range, initial=1)(color_curve)


ColorCurveFilter = ColorFilter


class ToneFilter(Filter):
  """filters.py:298-322."""
  filter_id = 4

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.curve_steps = int(cfg.curve_steps)
    assert 1 <= self.curve_steps <= _cabi.EXPO_CURVE_MAX_STEPS, 'cfg.curve_steps must be in [1, 16]'
    self.short_name = 'T'
    self.num_filter_parameters = cfg.curve_steps
    self._build_regressor()

  def uses_generic_kernels(self):
    return self.curve_steps != CURVE_STEPS_TUNED

  def process(self, img, param):
    if self.curve_steps == CURVE_STEPS_TUNED:
      return Filter.process(self, img, param)
    return _CurveFunction.apply(img, self.pack(param), 1, self.curve_steps)

  def filter_param_regressor(self, features):
    tone_curve = features.reshape(-1, 1, self.cfg.curve_steps)[:, None, None, :]
    return tanh_range(*self.cfg.tone_curve_range)(tone_curve)


class ContrastFilter(Filter):
  """filters.py:404-419."""
  filter_id = 5

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'Ct'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    return torch.tanh(features)


class WNBFilter(Filter):
  """filters.py:428-440."""
  filter_id = 6

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'BW'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    return torch.sigmoid(features)


class LevelFilter(Filter):
  """filters.py:449-466 (defined by the reference, not part of cfg.filters)."""
  filter_id = 8

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'Le'
    self.num_filter_parameters = 2
    self._build_regressor()

  def filter_param_regressor(self, features):
    return torch.sigmoid(features)


class _VignetApplyFunction(torch.autograd.Function):
  """out = img * (1 - mask(mask_params)) -- expo_vignet_apply_fwd/bwd (VignetFilter.apply, filters.py:341-396)."""

  @staticmethod
  def forward(ctx, img, mask_params, sharp, masking):
    img = img.contiguous()
    mask_params = mask_params.contiguous().float()
    y = torch.empty_like(img)
    _cabi.vignet_apply_fwd(img, y, mask_params, sharp, masking)
    ctx.save_for_backward(img, mask_params)
    ctx.args = (sharp, masking)
    return y

  @staticmethod
  def backward(ctx, dy):
    img, mask_params = ctx.saved_tensors
    sharp, masking = ctx.args
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dmask = torch.empty_like(mask_params)
    _cabi.vignet_apply_bwd(img, dy, dx, mask_params, dmask, sharp, masking)
    return dx, dmask, None, None


class VignetFilter(Filter):
  """filters.py:341-401.  In the reference ``process`` is ``img * 0`` (the additive term is commented out,
  filters.py:351-352) and the class is in no config, so ``apply`` = ``lerp(img, 0, mask)`` = ``img * (1 - mask)``
  with the elliptical mask of filters.py:360-396 (5 mask parameters; with masking off the mask is forced to 1 and
  the output is 0).  ``apply`` is ONE HIP kernel (``expo_vignet_apply_fwd``: mask evaluation + lerp), its backward
  one more (``expo_vignet_apply_bwd``: image gradient + the 5 mask-parameter gradients); the filter's own parameter
  (``sigmoid(features)``) reaches nothing, exactly as in the reference."""
  filter_id = None

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'V'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    return torch.sigmoid(features)

  def process(self, img, param):
    """filters.py:351-352: ``img * 0`` -- a zero mask makes expo_vignet_apply_fwd exactly that (mask forced to 1)."""
    zeros = torch.zeros((img.shape[0], self.get_num_mask_parameters()), dtype=torch.float32, device=img.device)
    return _VignetApplyFunction.apply(img, zeros, float(self.cfg.maximum_sharpness), False)

  def get_num_mask_parameters(self):
    return 5

  def get_mask(self, img, mask_parameters):
    """filters.py:360-396: sigmoid(((gx A)^2 + (gy B)^2 + C - 5) * sharp * D / 5) * (E / 5 * .5 + .5), as a tensor
    for debug_info / visualisers only -- ``apply`` evaluates the mask inside the kernel."""
    filter_input_range = 5
    assert mask_parameters.shape[1] == self.get_num_mask_parameters()
    mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
    h, w = int(img.shape[1]), int(img.shape[2])
    se = min(h, w)
    gi = ((torch.arange(h, dtype=torch.float64) + (se - h) / 2.0) / se - 0.5).float().to(img.device)
    gj = ((torch.arange(w, dtype=torch.float64) + (se - w) / 2.0) / se - 0.5).float().to(img.device)
    inp = (gi[None, :, None, None] * mp[:, None, None, 0, None])**2 + \
        (gj[None, None, :, None] * mp[:, None, None, 1, None])**2 + \
        mp[:, None, None, 2, None] - filter_input_range
    inp = inp * (self.cfg.maximum_sharpness * mp[:, None, None, 3, None] / filter_input_range)
    mask = torch.sigmoid(inp)
    mask = mask * (mp[:, None, None, 4, None] / filter_input_range * 0.5 + 0.5)
    if not self.use_masking():
      mask = mask * 0 + 1
    return mask

  def apply(self, img, img_features=None, specified_parameter=None, high_res=None):
    if callable(img) and not isinstance(img, torch.Tensor):
      return nn.Module.apply(self, img)
    assert (img_features is None) ^ (specified_parameter is None)
    if img_features is not None:
      filter_features, mask_parameters = self.extract_parameters(img_features)
      filter_parameters = self.filter_param_regressor(filter_features)
    else:
      assert not self.use_masking()
      filter_parameters = specified_parameter
      mask_parameters = torch.zeros((img.shape[0], self.get_num_mask_parameters()), dtype=torch.float32,
                                    device=img.device)
    debug_info = {'filter_parameters': filter_parameters[0]}
    self.mask_parameters = mask_parameters
    with torch.no_grad():  # debug output (first image) only
      self.mask = self.get_mask(img[:1], mask_parameters[:1])
    debug_info['mask'] = self.mask[0]
    mp = tanh_range(-5, 5, initial=0)(mask_parameters)  # filters.py:366-368
    if mp.shape[0] == 1 and img.shape[0] > 1:
      mp = mp.expand(img.shape[0], mp.shape[1])
    apply_one = lambda im: _VignetApplyFunction.apply(im, mp, float(self.cfg.maximum_sharpness),
                                                      bool(self.use_masking()))
    low = apply_one(img)
    high = apply_one(high_res) if high_res is not None else None
    return low, high, debug_info


class SaturationPlusFilter(Filter):
  """filters.py:474-498."""
  filter_id = 3

  def __init__(self, net, cfg):
    Filter.__init__(self, net, cfg)
    self.short_name = 'S+'
    self.num_filter_parameters = 1
    self._build_regressor()

  def filter_param_regressor(self, features):
    return torch.sigmoid(features)


ALL_FILTERS = (ExposureFilter, GammaFilter, ImprovedWhiteBalanceFilter, SaturationPlusFilter, ToneFilter,
               ContrastFilter, WNBFilter, ColorFilter)
# the classes whose filter_param_regressor expo_heads_regress_* restates (a subclass with its own regressor does not qualify)
FUSED_HEAD_TYPES = ALL_FILTERS + (LevelFilter,)


class _DispatchFunction(torch.autograd.Function):
  """Per-image filter choice + fused over-exposure penalty (expo_filter_dispatch_fwd/bwd)."""

  @staticmethod
  def forward(ctx, img, params24, filter_ids, hsv_grad_mode):
    img = img.contiguous()
    params24 = params24.contiguous().float()
    filter_ids = filter_ids.contiguous().to(torch.int32)
    y = torch.empty_like(img)
    penalty = torch.empty((img.shape[0],), dtype=torch.float32, device=img.device)
    _cabi.dispatch_fwd(filter_ids, img, y, params24, penalty)
    ctx.save_for_backward(img, params24, filter_ids)
    ctx.hsv_grad_mode = hsv_grad_mode
    return y, penalty

  @staticmethod
  def backward(ctx, dy, dpenalty):
    img, params24, filter_ids = ctx.saved_tensors
    dy = dy.contiguous().to(img.dtype)
    dpenalty = dpenalty.contiguous().float()
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dparams = torch.empty_like(params24)
    _cabi.dispatch_bwd(filter_ids, img, dy, dx, params24, dparams, dpenalty, ctx.hsv_grad_mode)
    return dx, dparams, None, None


class _MaskedDispatchFunction(torch.autograd.Function):
  """Per-image masked apply (expo_filter_apply_dispatch_fwd/bwd): the agent's step with cfg.masking = True."""

  @staticmethod
  def forward(ctx, img, params24, mask6, filter_ids, sharp, min_strength, hsv_grad_mode):
    img = img.contiguous()
    params24 = params24.contiguous().float()
    mask6 = mask6.contiguous().float()
    filter_ids = filter_ids.contiguous().to(torch.int32)
    y = torch.empty_like(img)
    _cabi.apply_dispatch_fwd(filter_ids, img, y, params24, mask6, sharp, min_strength)
    ctx.save_for_backward(img, params24, mask6, filter_ids)
    ctx.args = (sharp, min_strength, hsv_grad_mode)
    return y

  @staticmethod
  def backward(ctx, dy):
    img, params24, mask6, filter_ids = ctx.saved_tensors
    sharp, min_strength, hsv_grad_mode = ctx.args
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img) if ctx.needs_input_grad[0] else None
    dparams = torch.empty_like(params24)
    dmask = torch.empty_like(mask6)
    _cabi.apply_dispatch_bwd(filter_ids, img, dy, dx, params24, dparams, mask6, dmask, sharp, min_strength,
                             hsv_grad_mode)
    return dx, dparams, dmask, None, None, None, None


def dispatch_masked_filters(img, params24, mask6, filter_ids, maximum_sharpness, minimum_strength, hsv_grad_mode=0):
  """cfg.masking = True counterpart of :func:`dispatch_filters`: image n goes through the masked apply
  (filters.py:62-99, 110-148) of filter ``filter_ids[n]`` only.  mask6: (N, 6) = tanh_range(-5, 5)(raw mask parameters
  of the selected filter).  The reference computes all 8 masked applies and reduces with the one-hot (agent.py:58-77,
  119-125); values and gradients are identical because the one-hot zeroes the other seven."""
  return _MaskedDispatchFunction.apply(img, params24, mask6, filter_ids, float(maximum_sharpness),
                                       float(minimum_strength), hsv_grad_mode)


def dispatch_filters(img, params24, filter_ids, hsv_grad_mode=0):
  """One launch pair instead of "run all 8 filters, stack, one-hot, reduce_sum" (agent.py:58-77,
  119-125).  params24: (N, 24) float32, row n = packed params of filter ``filter_ids[n]`` in its
  first P slots.  Returns (y, overexposure_penalty[N]) -- the latter is agent.py:249-251's
  ``reduce_mean(maximum(net - 1, 0)**2, axis=(1,2,3))``."""
  return _DispatchFunction.apply(img, params24, filter_ids, hsv_grad_mode)


class _FusedSequenceFunction(torch.autograd.Function):
  """A FIXED per-image filter sequence in one pass each way (expo_chain_fused_fwd / expo_chain_fused_bwd)."""

  @staticmethod
  def forward(ctx, img, params24, filter_ids, hsv_grad_mode):
    img = img.contiguous()
    params24 = params24.contiguous().float()
    filter_ids = filter_ids.contiguous().to(torch.int32)
    y = torch.empty_like(img)
    _cabi.chain_fused_fwd(filter_ids, params24, img, y)
    ctx.save_for_backward(img, params24, filter_ids)
    ctx.hsv_grad_mode = hsv_grad_mode
    return y

  @staticmethod
  def backward(ctx, dy):
    img, params24, filter_ids = ctx.saved_tensors
    dy = dy.contiguous().to(img.dtype)
    dx = torch.empty_like(img)
    dparams = torch.empty_like(params24)
    _cabi.chain_fused_bwd(filter_ids, params24, img, dy, dx, dparams, ctx.hsv_grad_mode)
    return dx, dparams, None, None


def fused_sequence(img, params24, filter_ids, hsv_grad_mode=0):
  """Apply ``filter_ids[n, 0], filter_ids[n, 1], ...`` (C-ABI ids, -1 = nothing selected) with parameters
  ``params24[n, k, :P]`` to image n, differentiably, WITHOUT materialising the intermediate images: one read and
  one write of the image forward, one read of img and dy and one write of dx backward (activations recomputed in
  registers; at most ``_cabi.FUSED_BWD_MAX_STEPS`` steps when a gradient is needed).  Only for sequences that do
  not depend on the intermediate images -- in the reference's agent each step's parameters are regressed from the
  previous step's output (agent.py:30-125), which needs :func:`dispatch_filters` step by step; this serves the
  replayed sequence of the high-resolution path (net.py:796-821) and the ``chain_fused`` benchmark."""
  return _FusedSequenceFunction.apply(img, params24, filter_ids, hsv_grad_mode)

