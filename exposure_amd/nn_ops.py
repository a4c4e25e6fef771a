"""Autograd building blocks of the convnets that drive the filter path (``feature_extractor``
``/root/reference/agent.py:11-37``, ``cnn`` ``critics.py:6-38``, the FC heads): the stride-2 convolution and the
``bias + lrelu`` behind every layer.  On a ROCm device every convolution primitive runs on the in-house kernels of
``csrc/conv_ops.hip`` (f32 matrix cores; since round 6 the weight gradient and the first layers' data gradient too: no
MIOpen kernel, no zero fill); the FC layers' GEMMs stay with hipBLASLt.  CPU tensors (the tests' float64 runs) and shapes
the kernels do not take (an odd output width for the weight gradient, an output channel count that is not a multiple
of 4 for the data gradient, other dtypes) go through ``aten::convolution``.

* :func:`conv2d_nhwc` -- ``ly.conv2d(kernel_size=4, stride=2)`` (SAME) as a family of three autograd Functions that is
  closed under differentiation.  A convolution is bilinear in (input, weight), so every derivative of every order is
  one of three primitives: forward ``F(x, W)``, data gradient ``D(g, W)`` (a transposed convolution) and weight
  gradient ``G(x, g)``.  torch's generic double backward instead re-expresses the weight gradient of ``D`` as a
  FORWARD convolution with batch and channels swapped -- for the critic's first layer a 6-image batch of 64-channel
  64x64 inputs under a 32x32 dilated kernel (538 us + three layout transposes in MIOpen, gpurun r03p4).
* :func:`conv_trunk` -- a whole stack of layers as ONE once-differentiable node (the generator step); the critic
  update with its double backward is hand-scheduled (``exposure_amd/critic_direct.py``).
* :func:`bias_lrelu` -- ``lrelu(y + b)`` (util.py:225-229) in one HIP launch forward and one per backward
  (``expo_bias_lrelu_fwd`` / ``expo_lrelu_bwd``) instead of 2 + 4 element-wise torch launches.

Tensors are NHWC float32 (the reference's layout); convolutions see them as channels_last NCHW views, no copies.
"""
import torch
from torch.autograd.function import once_differentiable

from . import _cabi

_STRIDE, _PAD, _DIL = [2, 2], [1, 1], [1, 1]

# Python autograd Functions only know statically whether an input requires a gradient (ctx.needs_input_grad); they
# cannot see -- as torch's built-in nodes can -- that the CURRENT backward pass does not need it.  Two places of the
# training step differentiate through a convnet without wanting its parameter gradients, and would otherwise pay one
# weight-gradient convolution and one bias reduction per layer for results autograd throws away:
#   * the gradient penalty's inner `autograd.grad(D(x^), x^, create_graph=True)` (net.py:174-183)   -> _SKIP_PARAM_GRADS,
#     read at BACKWARD time (the same forward nodes serve the outer backward, which does need them);
#   * the generator step's passes through the critic and the value net, whose weights that step does not update
#     (net.py:222-241)                                                                              -> _FROZEN, read at
#     FORWARD time (the parameters enter the graph detached).
_SKIP_PARAM_GRADS = False
_FROZEN = False


class skip_parameter_gradients:
  """Context manager for a backward pass that only wants INPUT gradients (``torch.autograd.grad(y, x, ...)``)."""

  def __enter__(self):
    global _SKIP_PARAM_GRADS
    self.prev, _SKIP_PARAM_GRADS = _SKIP_PARAM_GRADS, True

  def __exit__(self, *exc):
    global _SKIP_PARAM_GRADS
    _SKIP_PARAM_GRADS = self.prev


class frozen_parameters:
  """Context manager for a forward pass whose convolution weights / biases take no part in the differentiation."""

  def __enter__(self):
    global _FROZEN
    self.prev, _FROZEN = _FROZEN, True

  def __exit__(self, *exc):
    global _FROZEN
    _FROZEN = self.prev


def _nchw(x_nhwc):
  return x_nhwc.permute(0, 3, 1, 2)


def _nhwc(x_nchw):
  y = x_nchw.permute(0, 2, 3, 1)
  return y if y.is_contiguous() else y.contiguous()


def _hip_conv(x, w):
  """The in-house implicit-GEMM kernels (csrc/conv_ops.hip) take float32 NHWC tensors on a device and a weight in
  channels_last memory order ([co][kh][kw][ci])."""
  return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 4 and
          tuple(w.shape[2:]) == (4, 4) and w.permute(0, 2, 3, 1).is_contiguous())


def _conv_fwd(x, w, bias=None, act=0, leak=0.2):
  """conv (+ bias + lrelu) of an NHWC tensor: ``expo_conv4x4s2_fwd`` or the library pair it replaces."""
  if _hip_conv(x, w):
    n, h, wd, _ = x.shape
    y = torch.empty((n, h // 2, wd // 2, w.shape[0]), dtype=torch.float32, device=x.device)
    _cabi.conv4x4s2_fwd(x, w, bias, y, act, leak)
    return y
  y = _nhwc(torch.ops.aten.convolution(_nchw(x), w, None, _STRIDE, _PAD, _DIL, False, [0, 0], 1))
  if act or bias is not None:
    assert act, 'a bias without the activation is not a layer of these nets'
    z = torch.empty_like(y)
    _cabi.bias_lrelu_fwd(y, bias, z, leak)
    return z
  return y


class _ConvF(torch.autograd.Function):
  """y = conv(x, W), NHWC in / NHWC out, kernel 4, stride 2, padding 1, no bias."""

  @staticmethod
  def forward(ctx, x, w):
    ctx.save_for_backward(x, w)
    return _conv_fwd(x, w)

  @staticmethod
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    gy = gy.contiguous()
    gx = _ConvD.apply(gy, w) if ctx.needs_input_grad[0] else None
    gw = _ConvG.apply(x, gy, w) if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS else None
    return gx, gw


class _ConvD(torch.autograd.Function):
  """dx = D(g, W): the data gradient of _ConvF = the transposed convolution of g (NHWC, spatial size doubles) --
  ``expo_conv4x4s2_bwd_data`` (four parity-class GEMMs on the f32 matrix cores; 6 / 17 input planes on the vector ALUs).
  The fallback is ``aten.convolution_backward`` with only the input gradient requested (the input operand only carries
  the shape) rather than a transposed forward convolution: that route picked a MIOpen kernel that faults on gfx950 /
  ROCm 7.2 (gpurun r03p9)."""

  @staticmethod
  def forward(ctx, g, w):
    ctx.save_for_backward(g, w)
    n, h, wd, _ = g.shape
    if _hip_conv(g, w) and w.shape[0] % 4 == 0:
      dx = torch.empty((n, 2 * h, 2 * wd, w.shape[1]), dtype=torch.float32, device=g.device)
      _cabi.conv4x4s2_bwd_data(g, w, dx)  # no zero fill: every element is written
      return dx
    x_like = torch.empty((n, w.shape[1], 2 * h, 2 * wd), dtype=g.dtype, device=g.device,
                         memory_format=torch.channels_last)
    return _nhwc(torch.ops.aten.convolution_backward(_nchw(g), x_like, w, None, _STRIDE, _PAD, _DIL, False, [0, 0], 1,
                                                     [True, False, False])[0])

  @staticmethod
  def backward(ctx, v):
    g, w = ctx.saved_tensors
    v = v.contiguous()
    gg = _ConvF.apply(v, w) if ctx.needs_input_grad[0] else None
    gw = _ConvG.apply(v, g, w) if ctx.needs_input_grad[1] else None
    return gg, gw


class _ConvG(torch.autograd.Function):
  """dW = G(x, g): the weight gradient of _ConvF (``expo_conv4x4s2_wrw``: a fixed summation order, no atomics, no zero
  fill); ``w_like`` only carries shape / layout."""

  @staticmethod
  def forward(ctx, x, g, w_like):
    ctx.save_for_backward(x, g)
    if _hip_conv(x, w_like) and g.dtype == torch.float32 and g.shape[2] % 2 == 0:  # (pixels are consumed in pairs)
      dw = torch.empty_like(w_like, memory_format=torch.preserve_format)  # the weight's own (channels_last) layout
      _cabi.conv4x4s2_wrw(x, g.contiguous(), dw)
      return dw
    return torch.ops.aten.convolution_backward(_nchw(g), _nchw(x), w_like, None, _STRIDE, _PAD, _DIL, False, [0, 0], 1,
                                               [False, True, False])[1]

  @staticmethod
  def backward(ctx, v):
    x, g = ctx.saved_tensors
    gx = _ConvD.apply(g, v) if ctx.needs_input_grad[0] else None
    gg = _ConvF.apply(x, v) if ctx.needs_input_grad[1] else None
    return gx, gg, None


def conv2d_nhwc(x, weight):
  """``ly.conv2d(x, C_out, kernel_size=4, stride=2)`` without bias and activation: NHWC float32 ``x`` (even H, W),
  ``weight`` (C_out, C_in, 4, 4) as ``nn.Conv2d`` holds it.  Differentiable to any order through library kernels."""
  assert x.dim() == 4 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and tuple(weight.shape[2:]) == (4, 4)
  return _ConvF.apply(x.contiguous(), weight.detach() if _FROZEN else weight)


class _LreluGrad(torch.autograd.Function):
  """dy = dz * slope(z) (``expo_lrelu_bwd``); linear in dz, so its backward is the same kernel on the incoming
  gradient (the slope is piecewise constant: no gradient reaches z)."""

  @staticmethod
  def forward(ctx, z, dz, leak):
    dz = dz.contiguous()
    dy = torch.empty_like(dz)
    _cabi.lrelu_bwd(z, dz, dy, leak)
    ctx.save_for_backward(z)
    ctx.leak = leak
    return dy

  @staticmethod
  def backward(ctx, v):
    z, = ctx.saved_tensors
    return None, _LreluGrad.apply(z, v, ctx.leak), None


class _LreluGradBias(torch.autograd.Function):
  """(dy, dbias) = (dz * slope(z), column sums of dy) in ONE pass (``expo_lrelu_bwd_bias``, round 4): the layer's bias
  gradient no longer costs a second reduction launch that re-reads dy.  Linear in dz; its own backward is the plain
  ``_LreluGrad`` of the two incoming gradients combined."""

  @staticmethod
  def forward(ctx, z, dz, leak):
    dz = dz.contiguous()
    dy = torch.empty_like(dz)
    db = torch.empty((z.shape[-1],), dtype=torch.float32, device=z.device)
    _cabi.lrelu_bwd_bias(z, dz, dy, db, leak)
    ctx.save_for_backward(z)
    ctx.leak = leak
    return dy, db

  @staticmethod
  def backward(ctx, v_dy, v_db):
    z, = ctx.saved_tensors
    if v_dy is None and v_db is None:
      return None, None, None
    v = v_dy if v_db is None else (v_db.expand_as(z) if v_dy is None else v_dy + v_db)
    return None, _LreluGrad.apply(z, v, ctx.leak), None


class _BiasLrelu(torch.autograd.Function):

  @staticmethod
  def forward(ctx, y, bias, leak):
    y = y.contiguous()
    z = torch.empty_like(y)
    _cabi.bias_lrelu_fwd(y, bias, z, leak)
    ctx.save_for_backward(z)
    ctx.leak = leak
    ctx.has_bias = bias is not None
    return z

  @staticmethod
  def backward(ctx, gz):
    z, = ctx.saved_tensors
    want_gb = ctx.has_bias and ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS
    if want_gb:
      gz = gz.contiguous()
      if _cabi.lrelu_bwd_bias_supported(z, gz):
        gy, gb = _LreluGradBias.apply(z, gz, ctx.leak)
        return gy, gb, None
    gy = _LreluGrad.apply(z, gz, ctx.leak)
    gb = gy.reshape(-1, gy.shape[-1]).sum(dim=0) if want_gb else None
    return gy, gb, None


class _ConvBiasLrelu(torch.autograd.Function):
  """z = lrelu(conv(x, W) + b): one layer of ``feature_extractor`` / ``cnn`` (agent.py:21-32, critics.py:13-35) as ONE
  launch (``expo_conv4x4s2_fwd`` with the bias and the activation in its epilogue).  The backward is the composition
  the two separate nodes had -- activation gradient (+ bias gradient in the same pass), data gradient, weight gradient
  -- built from Functions that are themselves differentiable, so the gradient penalty's double backward goes through."""

  @staticmethod
  def forward(ctx, x, w, b, leak):
    z = _conv_fwd(x, w, b, 1, leak)
    ctx.save_for_backward(x, w, z)
    ctx.leak = leak
    return z

  @staticmethod
  def backward(ctx, gz):
    x, w, z = ctx.saved_tensors
    want_gb = ctx.needs_input_grad[2] and not _SKIP_PARAM_GRADS
    gz = gz.contiguous()
    if want_gb and _cabi.lrelu_bwd_bias_supported(z, gz):
      gy, gb = _LreluGradBias.apply(z, gz, ctx.leak)
    else:
      gy = _LreluGrad.apply(z, gz, ctx.leak)
      gb = gy.reshape(-1, gy.shape[-1]).sum(dim=0) if want_gb else None
    gx = _ConvD.apply(gy, w) if ctx.needs_input_grad[0] else None
    gw = _ConvG.apply(x, gy, w) if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS else None
    return gx, gw, gb, None


def conv_bias_lrelu(x, weight, bias, leak=0.2):
  """``lrelu(ly.conv2d(x, C_out, kernel_size=4, stride=2) + bias)`` for NHWC float32 ``x``: fused where the in-house
  kernel runs, the ``conv2d_nhwc`` + ``bias_lrelu`` pair elsewhere (CPU, other dtypes)."""
  assert x.dim() == 4 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and tuple(weight.shape[2:]) == (4, 4)
  x = x.contiguous()
  if _hip_conv(x, weight) and bias is not None:
    if _FROZEN:
      weight, bias = weight.detach(), bias.detach()
    return _ConvBiasLrelu.apply(x, weight, bias, leak)
  return bias_lrelu(conv2d_nhwc(x, weight), bias, leak)


# ---- a whole stack of layers as one node (round 6) ------------------------------------------------------------------
# The generator step differentiates its convnets ONCE (net.py:222-241; only the critic update's gradient penalty needs the
# double backward, and that update is hand-scheduled: exposure_amd/critic_direct.py).  A stack `z_l = lrelu(conv(z_{l-1},
# W_l) + b_l)` as ONE autograd node runs its backward as the critic update does: the activation gradient of layer l - 1 in
# the EPILOGUE of layer l's data-gradient kernel, the bias gradient out of the weight-gradient launch -- per layer 1 + 2
# launches (data gradient; weight gradient + its reduce) instead of 5-6 (lrelu backward + bias finish, data gradient,
# zero fill + MIOpen's split-K weight gradient), and no library kernel.
_ONCE_DIFFERENTIABLE = False


class once_differentiable_convnets:
  """Context manager: inside, ``conv_trunk`` may run a stack of layers as one once-differentiable node."""

  def __enter__(self):
    global _ONCE_DIFFERENTIABLE
    self.prev, _ONCE_DIFFERENTIABLE = _ONCE_DIFFERENTIABLE, True

  def __exit__(self, *exc):
    global _ONCE_DIFFERENTIABLE
    _ONCE_DIFFERENTIABLE = self.prev


class _ConvTrunk(torch.autograd.Function):
  """z_L of the stack; backward: input gradient (if wanted) and every (dW_l, db_l) (unless the parameters are frozen /
  the pass only wants input gradients)."""

  @staticmethod
  def forward(ctx, x, leak, *wb):
    ws, bs = wb[0::2], wb[1::2]
    acts = [x]
    for w, b in zip(ws, bs):
      a = acts[-1]
      z = torch.empty((a.shape[0], a.shape[1] // 2, a.shape[2] // 2, w.shape[0]), dtype=torch.float32, device=a.device)
      _cabi.conv4x4s2_fwd(a, w, b, z, 1, leak)
      acts.append(z)
    ctx.save_for_backward(*acts, *ws)
    ctx.leak, ctx.layers = leak, len(ws)
    return acts[-1]

  @staticmethod
  @once_differentiable
  def backward(ctx, gz):
    n_l = ctx.layers
    acts, ws = ctx.saved_tensors[:n_l + 1], ctx.saved_tensors[n_l + 1:]
    want_w = any(ctx.needs_input_grad[2:]) and not _SKIP_PARAM_GRADS
    gy = torch.empty_like(acts[n_l])
    _cabi.lrelu_bwd(acts[n_l], gz.contiguous(), gy, ctx.leak)
    grads = [None] * (2 * n_l)
    gx = None
    wrw = []  # the layers' weight-gradient launches, finished by ONE reduce launch at the end (expo_conv4x4s2_wrw_group)
    for l in range(n_l, 0, -1):
      w = ws[l - 1]
      if want_w:
        dw = torch.empty_like(w, memory_format=torch.preserve_format)
        db = torch.empty((w.shape[0],), dtype=torch.float32, device=w.device)
        wrw.append((acts[l - 1], gy, dw, db, None))
        grads[2 * (l - 1)], grads[2 * (l - 1) + 1] = dw, db
      if l > 1:
        g = torch.empty_like(acts[l - 1])
        _cabi.conv4x4s2_bwd_data_mask(gy, w, acts[l - 1], g, ctx.leak)
        gy = g
      elif ctx.needs_input_grad[0]:
        gx = torch.empty_like(acts[0])
        _cabi.conv4x4s2_bwd_data(gy, w, gx)
    for k in range(0, len(wrw), 8):
      _cabi.conv4x4s2_wrw_group(wrw[k:k + 8])
    return (gx, None) + tuple(grads)


class _ConvTrunks(torch.autograd.Function):
  """Several stacks of equally many layers on ONE input (the agent's filter and selector extractors, agent.py:47-56): the
  outputs z_L of every stack.  Backward: the stacks' data gradients stack by stack, and the weight gradients of ALL their
  layers as one grouped launch (expo_conv4x4s2_wrw_group takes eight layers: two stacks of four share one grid and one
  reduce launch instead of two of each)."""

  @staticmethod
  def forward(ctx, x, leak, stacks, const_planes, *wb):
    n_l = len(wb) // (2 * stacks)
    ws = [wb[2 * n_l * s:2 * n_l * (s + 1):2] for s in range(stacks)]
    bs = [wb[2 * n_l * s + 1:2 * n_l * (s + 1):2] for s in range(stacks)]
    acts = [[x] for _ in range(stacks)]
    # two stacks of one architecture: layer l of both as ONE grid (expo_conv4x4s2_fwd_pair) -- at batch 64 a layer alone
    # leaves CUs idle
    paired = stacks == 2 and all(ws[0][l].shape == ws[1][l].shape for l in range(n_l))
    for l in range(n_l):
      zs = []
      for s in range(stacks):
        a, w = acts[s][-1], ws[s][l]
        zs.append(torch.empty((a.shape[0], a.shape[1] // 2, a.shape[2] // 2, w.shape[0]), dtype=torch.float32, device=a.device))
      # the first layer of an input whose channels 3 .. are per-image constants (planes_concat): K = 48 instead of 16 cin
      fold = l == 0 and const_planes and planes_fold(x.shape, ws[0][0].shape[0])
      if paired:
        pair = _cabi.conv4x4s2_fwd_planes_pair if fold else _cabi.conv4x4s2_fwd_pair
        pair((acts[0][-1], ws[0][l], bs[0][l], zs[0]), (acts[1][-1], ws[1][l], bs[1][l], zs[1]), 1, leak)
      else:
        for s in range(stacks):
          (_cabi.conv4x4s2_fwd_planes if fold else _cabi.conv4x4s2_fwd)(acts[s][-1], ws[s][l], bs[s][l], zs[s], 1, leak)
      for s in range(stacks):
        acts[s].append(zs[s])
    saved = [x]
    for s in range(stacks):
      saved += acts[s][1:]
    ctx.save_for_backward(*saved, *wb[0::2])
    ctx.leak, ctx.layers, ctx.stacks, ctx.paired = leak, n_l, stacks, paired
    return tuple(acts[s][-1] for s in range(stacks))

  @staticmethod
  @once_differentiable
  def backward(ctx, *gzs):
    n_l, stacks = ctx.layers, ctx.stacks
    x = ctx.saved_tensors[0]
    zs = ctx.saved_tensors[1:1 + n_l * stacks]
    ws_all = ctx.saved_tensors[1 + n_l * stacks:]
    grads = [None] * (2 * n_l * stacks)
    gx = None
    wrw = []
    live = [s for s in range(stacks) if gzs[s] is not None]
    acts = {s: (x,) + tuple(zs[n_l * s:n_l * (s + 1)]) for s in live}
    ws = {s: ws_all[n_l * s:n_l * (s + 1)] for s in live}
    want_w = {s: any(ctx.needs_input_grad[4 + 2 * n_l * s:4 + 2 * n_l * (s + 1)]) and not _SKIP_PARAM_GRADS for s in live}
    gy = {}
    for s in live:
      gy[s] = torch.empty_like(acts[s][n_l])
      _cabi.lrelu_bwd(acts[s][n_l], gzs[s].contiguous(), gy[s], ctx.leak)
    paired = ctx.paired and len(live) == 2
    for l in range(n_l, 0, -1):
      for s in live:
        if want_w[s]:
          w = ws[s][l - 1]
          dw = torch.empty_like(w, memory_format=torch.preserve_format)
          db = torch.empty((w.shape[0],), dtype=torch.float32, device=w.device)
          wrw.append((acts[s][l - 1], gy[s], dw, db, None))
          grads[2 * (n_l * s + l - 1)], grads[2 * (n_l * s + l - 1) + 1] = dw, db
      if l > 1:
        g = {s: torch.empty_like(acts[s][l - 1]) for s in live}
        if paired:  # the two stacks' data gradients as one grid
          a, b = live
          _cabi.conv4x4s2_bwd_data_mask_pair((gy[a], ws[a][l - 1], acts[a][l - 1], g[a]),
                                             (gy[b], ws[b][l - 1], acts[b][l - 1], g[b]), ctx.leak)
        else:
          for s in live:
            _cabi.conv4x4s2_bwd_data_mask(gy[s], ws[s][l - 1], acts[s][l - 1], g[s], ctx.leak)
        gy = g
      elif ctx.needs_input_grad[0]:
        for s in live:
          g = torch.empty_like(x)
          _cabi.conv4x4s2_bwd_data(gy[s], ws[s][0], g)
          gx = g if gx is None else gx + g
    for k in range(0, len(wrw), 8):
      _cabi.conv4x4s2_wrw_group(wrw[k:k + 8])
    return (gx, None, None, None) + tuple(grads)


def planes_fold(x_shape, cout):
  """A first layer whose constant planes are worth folding (expo_conv4x4s2_fwd_planes): the generator's 14 and the value
  net's 17 planes (10.8 against 14.3 us, 21.9 against 38.5); the critic's 6 planes run as fast on the row-staged kernel."""
  return x_shape[-1] >= 10 and _cabi.conv_planes_ok(x_shape, cout)


def _trunk_ok(x, convs):
  ok = _ONCE_DIFFERENTIABLE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
  h, w = x.shape[1], x.shape[2]
  for conv in convs:
    ok = ok and _hip_conv(x, conv.weight) and conv.bias is not None and conv.weight.shape[0] % 4 == 0 and \
        h % 2 == 0 and w % 2 == 0 and (w // 2) % 2 == 0
    h, w = h // 2, w // 2
  return ok


def conv_trunks(x, stacks, leak=0.2, const_planes=False):
  """``[conv_trunk(x, convs) for convs in stacks]`` -- as ONE once-differentiable node when every stack qualifies and they
  have equally many layers (their weight gradients then share one launch), stack by stack otherwise.  ``const_planes``: the
  caller vouches that channels 3 .. of ``x`` are per-image constants (``planes_concat``'s output): the first layers then
  run with those planes folded into a per-image term."""
  x = x.contiguous()
  stacks = [list(c) for c in stacks]
  if len(stacks) < 2 or len({len(c) for c in stacks}) != 1 or not all(_trunk_ok(x, c) for c in stacks):
    return [conv_trunk(x, c, leak) for c in stacks]
  wb = []
  for convs in stacks:
    for conv in convs:
      wb += [conv.weight.detach(), conv.bias.detach()] if _FROZEN else [conv.weight, conv.bias]
  return list(_ConvTrunks.apply(x, leak, len(stacks), bool(const_planes), *wb))


def conv_trunk(x, convs, leak=0.2):
  """The stack of ``nn.Conv2d(k=4, s=2, p=1)`` + lrelu layers ``convs`` on NHWC float32 ``x`` (agent.py:21-32,
  critics.py:13-35): one once-differentiable node inside ``once_differentiable_convnets()`` on a ROCm device (every layer's
  output width even: the weight-gradient kernel consumes pixels in pairs), layer by layer through ``conv_bias_lrelu``
  (differentiable to any order) everywhere else."""
  x = x.contiguous()
  ok = _ONCE_DIFFERENTIABLE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
  h, w = x.shape[1], x.shape[2]
  for conv in convs:
    ok = ok and _hip_conv(x, conv.weight) and conv.bias is not None and conv.weight.shape[0] % 4 == 0 and \
        h % 2 == 0 and w % 2 == 0 and (w // 2) % 2 == 0
    h, w = h // 2, w // 2
  if not ok:
    for conv in convs:
      x = conv_bias_lrelu(x, conv.weight, conv.bias, leak)
    return x
  wb = []
  for conv in convs:
    wb += [conv.weight.detach(), conv.bias.detach()] if _FROZEN else [conv.weight, conv.bias]
  return _ConvTrunk.apply(x, leak, *wb)


def bias_lrelu(y, bias=None, leak=0.2):
  """``lrelu(y + bias)`` with the channel as the last dimension of ``y`` (float32).  util.py:225-229:
  ``f1*x + f2*|x|``, f1 = (1+leak)/2, f2 = (1-leak)/2, i.e. ``x if x > 0 else leak*x`` (equal to the literal formula
  within 1 ulp: 0.6x + 0.4x vs x), with TF's sub-gradient f1 exactly at 0."""
  if _FROZEN and bias is not None:
    bias = bias.detach()
  return _BiasLrelu.apply(y, bias, leak)


class _GradPenaltyTerm(torch.autograd.Function):
  """Per image: ``term = max(sqrt(1e-6 + sum g^2) - 1, 0)^2`` and the norm itself (net.py:185-187) from the critic's
  input gradient g -- ``expo_grad_penalty_fwd`` (one block per image) and, for the gradient that flows on into the
  critic's double backward, ``expo_grad_penalty_bwd``: two launches instead of pow / sum / add / sqrt / sub / clamp /
  pow and their seven backward launches.  The norm output is not differentiated (a reported value)."""

  @staticmethod
  def forward(ctx, g):
    g = g.contiguous()
    n = g.shape[0]
    norm = torch.empty((n,), dtype=torch.float32, device=g.device)
    term = torch.empty((n,), dtype=torch.float32, device=g.device)
    _cabi.grad_penalty_fwd(g, norm, term)
    ctx.save_for_backward(g, norm)
    ctx.mark_non_differentiable(norm)
    return term, norm

  @staticmethod
  def backward(ctx, dterm, _dnorm):
    g, norm = ctx.saved_tensors
    dg = torch.empty_like(g)
    _cabi.grad_penalty_bwd(g, norm, dterm.contiguous().float(), dg)
    return dg


def grad_penalty_term(gradients):
  """(term (N,), norm (N,)) of the one-sided gradient penalty; fused on the device, torch ops elsewhere."""
  if gradients.is_cuda and gradients.dtype == torch.float32:
    return _GradPenaltyTerm.apply(gradients)
  norm = torch.sqrt(1e-6 + (gradients**2).sum(dim=tuple(range(1, gradients.dim()))))
  return torch.clamp_min(norm - 1.0, 0.0)**2, norm.detach()


class _PlanesConcat(torch.autograd.Function):
  """``cat([images, vec broadcast as planes], channel) - offset`` as float32 NHWC in one launch (``expo_planes_concat``).
  Linear: its backward is a channel slice and a per-image sum, left to torch so that every higher derivative (the
  gradient penalty differentiates the critic's input gradient again) comes from autograd."""

  @staticmethod
  def forward(ctx, images, vec, offset):
    images = images.contiguous()
    v = 0 if vec is None else vec.shape[1]
    out = torch.empty(tuple(images.shape[:-1]) + (3 + v,), dtype=torch.float32, device=images.device)
    _cabi.planes_concat(images, None if vec is None else vec.contiguous().float(), out, offset)
    ctx.img_dtype = images.dtype
    return out

  @staticmethod
  def backward(ctx, dout):
    d_images = d_vec = None
    if ctx.needs_input_grad[0]:
      d_images = dout[..., :3].to(ctx.img_dtype)
    if ctx.needs_input_grad[1]:
      d_vec = dout[..., 3:].sum(dim=tuple(range(1, dout.dim() - 1)))
    return d_images, d_vec, None


def planes_concat(images, vec, offset=0.5):
  """The input of ``cnn`` / ``feature_extractor``: NHWC ``images`` (3 channels) with the rows of ``vec`` (N, V) appended
  as V constant planes, minus ``offset`` -- critics.py:64-76, agent.py:17-19 + util.py:31-36."""
  if images.is_cuda and images.dtype in (torch.float16, torch.float32) and images.shape[-1] == 3:
    return _PlanesConcat.apply(images, vec, float(offset))
  net = images.float()
  if vec is not None:
    shape = tuple(images.shape[:-1]) + (vec.shape[1],)
    net = torch.cat([net, vec.float().reshape((vec.shape[0],) + (1,) * (images.dim() - 2) + (vec.shape[1],)).expand(shape)],
                    dim=images.dim() - 1)
  return net - offset


class _GeneratorLosses(torch.autograd.Function):
  """(g_loss, v_loss, reward, q) of the generator step from the per-image scalars (``expo_generator_losses``: ~25 tiny
  launches forward and as many backward become one + one).  The gradient of g_loss reaches fake_logit, new_value,
  surrogate and penalty; old_value's path is v_loss's (``_ValueLoss``), whose backward pass runs on its own."""

  @staticmethod
  def forward(ctx, fake_logit, fake_input_logit, new_value, old_value, new_states, penalty, surrogate, consts, use_td):
    n = fake_logit.numel()
    flat = lambda t: t.detach().reshape(n).contiguous().float()
    dev = fake_logit.device
    losses = torch.empty((2,), dtype=torch.float32, device=dev)
    reward = torch.empty((n,), dtype=torch.float32, device=dev)
    q = torch.empty((n,), dtype=torch.float32, device=dev)
    coef = torch.empty((5, n), dtype=torch.float32, device=dev)
    _cabi.generator_losses(flat(fake_logit), flat(fake_input_logit), flat(new_value), flat(old_value),
                           new_states.detach().contiguous().float(), None if penalty is None else flat(penalty),
                           flat(surrogate), consts, use_td, losses, reward, q, coef)
    ctx.save_for_backward(coef)
    ctx.shapes = (fake_logit.shape, new_value.shape, None if penalty is None else penalty.shape, surrogate.shape)
    # only g_loss carries this node: v_loss's own backward pass (which runs first, _ValueLoss) must not traverse it
    g_loss, v_val = losses[0], losses[1]
    ctx.mark_non_differentiable(v_val, reward, q, coef)
    return g_loss, v_val, reward, q, coef

  @staticmethod
  @once_differentiable
  def backward(ctx, dg, _dv, _dr, _dq, _dc):
    coef, = ctx.saved_tensors
    s_fl, s_nv, s_pen, s_sur = ctx.shapes
    g = coef[:4] * dg  # one launch: the four gradient rows
    return (g[0].reshape(s_fl), None, g[1].reshape(s_nv), None, None, None if s_pen is None else g[3].reshape(s_pen),
            g[2].reshape(s_sur), None, None)


class _ValueLoss(torch.autograd.Function):
  """v_loss = mean((q - old_value)^2) with q a constant (net.py:131-134): the value computed by ``_GeneratorLosses`` tied to
  ``old_value`` -- d v_loss / d old_value = -2 adv / N (row 4 of its coefficients)."""

  @staticmethod
  def forward(ctx, old_value, v_loss, coef):
    ctx.save_for_backward(coef)
    ctx.shape = old_value.shape
    return v_loss.clone()

  @staticmethod
  @once_differentiable
  def backward(ctx, dv):
    coef, = ctx.saved_tensors
    return (coef[4] * dv).reshape(ctx.shape), None, None


def generator_losses_fused(fake_logit, fake_input_logit, new_value, old_value, new_states, penalty, surrogate, consts,
                           use_td):
  """-> (g_loss, v_loss, reward (N, 1), q_value (N, 1)); see ``_GeneratorLosses``.  ``consts`` = (all_reward,
  critic_logit_multiplier, discount_factor, parameter_lr_mul, maximum_trajectory_length)."""
  g_loss, v_val, reward, q, coef = _GeneratorLosses.apply(fake_logit, fake_input_logit, new_value, old_value.detach(),
                                                          new_states, penalty, surrogate, tuple(consts), bool(use_td))
  v_loss = _ValueLoss.apply(old_value, v_val, coef)
  return g_loss, v_loss, reward.reshape(fake_logit.shape), q.reshape(fake_logit.shape)


def critic_step_inputs(real_data, fake_output, alpha):
  """(cat([real, fake]) as float32, real + alpha (fake - real)) -- ``expo_gp_inputs``: one launch for the two dtype
  conversions, the concatenation and the three element-wise passes of the interpolation (net.py:170-172)."""
  if real_data.is_cuda and real_data.dtype in (torch.float16, torch.float32) and real_data.dtype == fake_output.dtype:
    real_data, fake_output = real_data.contiguous(), fake_output.contiguous()
    n = real_data.shape[0]
    cat = torch.empty((2 * n,) + tuple(real_data.shape[1:]), dtype=torch.float32, device=real_data.device)
    interp = torch.empty(real_data.shape, dtype=torch.float32, device=real_data.device)
    _cabi.gp_inputs(real_data, fake_output, alpha.contiguous().float(), cat, interp)
    return cat, interp
  real_data, fake_output = real_data.float(), fake_output.float()
  return torch.cat([real_data, fake_output], dim=0), real_data + alpha * (fake_output - real_data)
