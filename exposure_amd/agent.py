"""Policy network of the retouching agent (``/root/reference/agent.py``) on PyTorch-ROCm.

``feature_extractor`` (agent.py:11-37) and ``agent_generator`` (agent.py:41-260) keep their
reference names, arguments and return structure.  What changes is the image path:

* the reference applies ALL filters to the same input, stacks the 8 results and multiplies by
  ``one_hot(selected_filter_id)`` (agent.py:58-77, 119-125) -- 8 full-image passes of which 7
  are discarded.  The selector's pdf does not depend on the filter outputs, so here the action
  is sampled first and ONE fused HIP launch pair (``expo_filter_dispatch_fwd``) applies, per
  image, only the selected filter -- and accumulates the over-exposure penalty
  (agent.py:249-251) in the same pass.  Gradients are identical: the one-hot product zeroes
  every non-selected filter's image and parameter gradients in the reference too.
* the convolutions / FCs are MIOpen / hipBLASLt GEMMs (MFMA) through torch; tensors stay NHWC
  (``channels_last``) end to end so no layout change sits between the filter kernels and conv1.

Stochastic inputs are explicit so runs are reproducible and parity-testable: ``z[:, 0]`` is
the selection noise (agent.py:47) and ``dropout_masks`` (two (N, 4096) 0/1 tensors) replace the
always-on ``tf.nn.dropout`` of agent.py:36 (pass ``None`` to draw them).
"""
import math
import os

import torch
from torch import nn

from . import filters as F
from .nn_ops import conv_trunk, conv_trunks, planes_concat
from .util import (STATE_DROPOUT_BEGIN, STATE_REWARD_DIM, STATE_STEP_DIM, STATE_STOPPED_DIM,
                   enrich_image_input, lrelu)


def _xavier_conv(conv):
  nn.init.xavier_uniform_(conv.weight)
  nn.init.zeros_(conv.bias)


class FeatureExtractor(nn.Module):
  """agent.py:11-37: ``x - 0.5``; conv4x4/s2 (SAME) + lrelu until 4x4; flatten (H,W,C order);
  dropout(keep_prob) ALWAYS on.  Input NHWC (N, S, S, C_in)."""

  def __init__(self, in_channels, output_dim, cfg, size=None):
    super().__init__()
    size = int(size or cfg.source_img_size)
    min_feature_map_size = 4
    assert output_dim % (min_feature_map_size**2) == 0, 'output dim=%d' % output_dim
    self.output_dim = output_dim
    self.keep_prob = float(cfg.dropout_keep_prob)
    channels = cfg.base_channels
    layers = []
    size //= 2
    layers.append(nn.Conv2d(in_channels, channels, kernel_size=4, stride=2, padding=1))
    prev = channels
    while size > min_feature_map_size:
      if size == min_feature_map_size * 2:
        channels = output_dim // (min_feature_map_size**2)
      else:
        channels *= 2
      assert size % 2 == 0
      size //= 2
      layers.append(nn.Conv2d(prev, channels, kernel_size=4, stride=2, padding=1))
      prev = channels
    self.convs = nn.ModuleList(layers)
    for c in self.convs:
      _xavier_conv(c)
    self.to(memory_format=torch.channels_last)

  def forward(self, net_nhwc, dropout_mask=None, centered=False, trunk_out=None):
    # NHWC end to end (the convolutions see channels_last views: no copy, MIOpen picks its NHWC kernels);
    # `centered`: the caller has already subtracted 0.5 (Agent.forward builds the enriched input of both extractors in
    # one launch, nn_ops.planes_concat); `trunk_out`: the caller has run the convolutions already (Agent.forward runs
    # both extractors' stacks as one node, nn_ops.conv_trunks)
    if trunk_out is None:
      net = net_nhwc if centered else net_nhwc.float() - 0.5
      net = conv_trunk(net, self.convs)
    else:
      net = trunk_out
    net = net.reshape(net.shape[0], self.output_dim)  # TF reshape order (H,W,C)
    if dropout_mask is None:
      dropout_mask = (torch.rand_like(net) < self.keep_prob).to(net.dtype)
    # tf.nn.dropout: x / keep_prob * mask
    return net * dropout_mask * (1.0 / self.keep_prob)


def feature_extractor(net, output_dim, cfg, module, dropout_mask=None):
  """Functional spelling of agent.py:11 for callers that hold the module."""
  assert module.output_dim == output_dim
  return module(net, dropout_mask)


def exclusive_cumsum(pdf):
  """``tf.cumsum(pdf, axis=1, exclusive=True)`` (pdf_sample_layer.py:7): a SHIFTED prefix scan --
  out[:, 0] = 0, out[:, j] = out[:, j-1] + pdf[:, j-1], accumulated left to right in the input dtype
  exactly like TF's CPU scan functor.  (``cumsum(pdf) - pdf`` is NOT the same in floating point: it
  differs by one ulp in ~22 % of the entries, enough to flip a sampled id when the noise lands on that
  ulp.)  K is the number of filters (8): the explicit column loop is K-1 tiny adds, runs identically on
  any device and does not depend on the association order of a parallel device scan."""
  k = pdf.shape[1]
  cols = [torch.zeros_like(pdf[:, 0])]
  for j in range(1, k):
    cols.append(cols[-1] + pdf[:, j - 1])
  return torch.stack(cols, dim=1)


def row_sum(pdf):
  """``tf.reduce_sum(pdf, axis=1, keep_dims=True)`` (pdf_sample_layer.py:6) with an EXPLICIT association
  order, so the integer ids that follow do not depend on the device's reduction tree: for K == 8 the
  order of Eigen's packet reducer that TF-1 CPU kernels use (accumulate 4-wide packets, then the
  horizontal add (x0+x2)+(x1+x3): ((p0+p4)+(p2+p6)) + ((p1+p5)+(p3+p7)), the same for SSE and AVX
  builds); any other K sums left to right.  TF's own order is an implementation detail of an absent
  dependency (parity unpinned); what is guaranteed is product == oracle, bit for bit, on any device."""
  k = pdf.shape[1]
  c = [pdf[:, j] for j in range(k)]
  if k == 8:
    total = ((c[0] + c[4]) + (c[2] + c[6])) + ((c[1] + c[5]) + (c[3] + c[7]))
  else:
    total = c[0]
    for j in range(1, k):
      total = total + c[j]
  return total[:, None]


def pdf_sample(pdf, uniform_noise):
  """pdf_sample_layer.py:5-10 (bit-identical integer result for identical pdf / noise)."""
  pdf = pdf / (row_sum(pdf) + 1e-36)
  cdf = exclusive_cumsum(pdf)
  return ((cdf < uniform_noise).sum(dim=1) - 1).to(torch.int32)


class _AgentSelect(torch.autograd.Function):
  """Action pdf, sampling, one-hot, surrogate, state update and the image-independent penalty terms of one agent step
  (agent.py:87-125, 207-252) -- ``expo_agent_select_fwd / _bwd``: one launch each way instead of ~65 + ~35 tiny torch
  launches.  Differentiable outputs: surrogate and penalty_base (-> logits); the rest is integer-valued, data, or (pdf,
  entropy) reported only."""

  @staticmethod
  def forward(ctx, logits, noise, states, progress, consts, is_train):
    logits = logits.contiguous().float()
    n, k = logits.shape
    dev = logits.device
    states = states.contiguous().float()
    noise = noise.contiguous().float()
    pdf, onehot = torch.empty((n, k), device=dev), torch.empty((n, k), device=dev)
    entropy, surrogate, pen = (torch.empty((n, 1), device=dev) for _ in range(3))
    selected = torch.empty((n,), dtype=torch.int32, device=dev)
    new_states = torch.empty_like(states)
    F._cabi.agent_select_fwd(logits, noise, states, progress, consts, is_train, pdf, entropy, selected, onehot, surrogate,
                             new_states, pen)
    ctx.save_for_backward(logits, selected, progress)
    ctx.meta = (tuple(consts), states.shape[1])
    ctx.mark_non_differentiable(pdf, entropy, selected, onehot, new_states)
    return pdf, entropy, selected, onehot, surrogate, new_states, pen

  @staticmethod
  def backward(ctx, _dpdf, _dent, _dsel, _doh, d_surrogate, _dns, d_pen):
    logits, selected, progress = ctx.saved_tensors
    consts, state_dim = ctx.meta
    d_logits = torch.empty_like(logits)
    F._cabi.agent_select_bwd(logits, selected, progress, consts, state_dim, d_surrogate.contiguous().float(),
                             d_pen.contiguous().float(), d_logits)
    return d_logits, None, None, None, None, None


class Agent(nn.Module):
  """``agent_generator`` (agent.py:41-260) as a module.  ``forward`` mirrors its signature:
  ``inp = (net, z, states)``, ``is_train`` (0/1), ``progress`` (float), optional ``high_res``."""

  def __init__(self, cfg, img_shape=None):
    super().__init__()
    self.cfg = cfg
    s = cfg.source_img_size
    img_shape = img_shape or (1, s, s, cfg.real_img_channels)
    self.filters = nn.ModuleList([x(img_shape, cfg) for x in cfg.filters])  # agent.py:45
    in_ch = cfg.real_img_channels + (cfg.num_state_dim if cfg.img_include_states else 0)
    # agent.py:62-65: the non-shared branch calls enrich_image_input(cfg, net) with a missing argument,
    # i.e. it raises a TypeError in the reference itself -- there is nothing to mirror
    # agent.py:62-65: the non-shared branch calls enrich_image_input(cfg, net) with a missing argument, i.e.
    # it raises a TypeError in the reference itself -- there is nothing to mirror
    assert cfg.shared_feature_extractor, 'only the shared feature extractor of the shipped configs is built'
    self.filter_features = FeatureExtractor(in_ch, cfg.feature_extractor_dims, cfg)
    self.selector_features = FeatureExtractor(in_ch, cfg.feature_extractor_dims, cfg)
    self.selector_fc1 = nn.Linear(cfg.feature_extractor_dims, cfg.fc1_size)
    self.selector_fc2 = nn.Linear(cfg.fc1_size, len(cfg.filters))
    for fc in (self.selector_fc1, self.selector_fc2):
      nn.init.xavier_uniform_(fc.weight)
      nn.init.zeros_(fc.bias)
    self._packed_heads = None  # filters.PackedHeads, built at the first fused forward on a device
    # position in cfg.filters -> C-ABI filter id (they differ when cfg.filters is a subset/reorder)
    self.register_buffer('abi_filter_ids', torch.tensor([f.filter_id for f in self.filters], dtype=torch.int32),
                         persistent=False)
    # cfg.filters in the C-ABI's own order (both shipped configs): the selected position IS the filter id -- no look-up
    # (compare, clamp, cast, gather, fill, where: six launches per step)
    self._abi_identity = [f.filter_id for f in self.filters] == list(range(len(self.filters)))

  def pack_heads(self, now=True):
    """Builds the packed storage of the K filter heads (filters.PackedHeads) and, with ``now``, moves the parameters
    into it right away -- what ``gan.GAN`` does once at construction, after ``.to(device)`` and BEFORE optimisers,
    gradient buckets and hipGraphs take pointers to the parameters.  -> the pack, or None when the heads do not fit it
    (EXPO_PACKED_HEADS=0, unequal layer sizes, not fp32, not on a device)."""
    if self._packed_heads is None:
      pack = F.PackedHeads(self.filters) if os.environ.get('EXPO_PACKED_HEADS', '1') == '1' else None
      self._packed_heads = pack if pack is not None and pack.supported() else False
    pack = self._packed_heads or None
    if pack is not None and now and self.filters[0].fc1.weight.is_cuda:
      pack.ensure()
    return pack

  def regress_all(self, filter_features):
    """Per-filter FC heads + range squashing -> (reference-shaped parameter tensors, raw mask parameters)."""
    out, mask_params = [], []
    for filt in self.filters:
      f, mp = filt.extract_parameters(filter_features)
      out.append(filt.filter_param_regressor(f))
      mask_params.append(mp)
    return out, mask_params

  def _apply_masked(self, net, params24, mask6, abi_ids):
    """cfg.masking = True (off in both shipped configs).  The reference runs every filter's masked ``apply`` on the
    whole batch, stacks the 8 results and reduces with the one-hot (agent.py:58-77, 119-125); the one-hot zeroes seven
    of them, so ONE HIP launch applies, per image, only the selected filter's mask + process + lerp
    (``expo_filter_apply_dispatch_fwd``) -- the same restructuring as the unmasked path, with the selected filter's
    6 mask parameters gathered like its filter parameters."""
    cfg = self.cfg
    return F.dispatch_masked_filters(net, params24, mask6, abi_ids, cfg.maximum_sharpness, cfg.minimum_strength,
                                     int(cfg.get('hsv_grad_mode', 0)))

  def action_pdf(self, selector_features):
    """agent.py:87-107 -> (pdf, entropy)."""
    cfg = self.cfg
    h = lrelu(self.selector_fc1(selector_features))
    pdf = torch.softmax(self.selector_fc2(h), dim=1) + 1e-37
    pdf = pdf * (1 - cfg.exploration) + cfg.exploration * 1.0 / len(self.filters)
    pdf = pdf / (pdf.sum(dim=1, keepdim=True) + 1e-30)
    entropy = (-pdf * torch.log(pdf)).sum(dim=1)[:, None]
    return pdf, entropy

  def forward(self, inp, is_train, progress, cfg=None, high_res=None, dropout_masks=None):
    cfg = cfg or self.cfg
    net, z, states = inp
    n = net.shape[0]
    k = len(self.filters)
    selection_noise = z[:, 0:1]
    masks = dropout_masks or (None, None)

    if net.is_cuda and cfg.img_include_states and net.shape[-1] == 3:
      enriched, centered = planes_concat(net, states, 0.5), True  # float, concat and `- 0.5` of both extractors at once
    else:
      enriched, centered = enrich_image_input(cfg, net.float(), states), False
    # both extractors read the same input: their stacks as ONE autograd node (the weight gradients of all eight layers then
    # share one launch, nn_ops.conv_trunks)
    trunk_f = trunk_s = None
    if centered:
      trunk_f, trunk_s = conv_trunks(enriched, [self.filter_features.convs, self.selector_features.convs],
                                     const_planes=True)  # (channels 3 ..: the states planes_concat broadcast)
    filter_features = self.filter_features(enriched, masks[0], centered=centered, trunk_out=trunk_f)
    # Training-time fast path (round 4): the regressors and the one-hot gather of the selected filter's parameters as
    # ONE kernel behind the heads' second FCs (filters.heads_regress_select).  Needs what the dispatch kernels need
    # (8-step curves, masking off) and a device; everything else takes the op-by-op path below.
    fused_heads = (not cfg.masking and net.is_cuda and high_res is None and len(self.filters) <= 16 and
                   not any(f.uses_generic_kernels() for f in self.filters) and
                   all(type(f) in F.FUSED_HEAD_TYPES for f in self.filters))
    if fused_heads:
      # the K heads' two FCs as one GEMM + one batched GEMM over parameters packed in place (filters.PackedHeads);
      # EXPO_PACKED_HEADS=0: one addmm / lrelu / addmm per head
      if self._packed_heads is None:
        self.pack_heads(now=False)
      if self._packed_heads and filter_features.dtype == torch.float32:
        raws = self._packed_heads(filter_features)
      else:
        raws = [filt.fc2(lrelu(filt.fc1(filter_features))) for filt in self.filters]
      params = mask_params = None
    else:
      params, mask_params = self.regress_all(filter_features)  # 8 x reference-shaped, 8 x (N, 6)

    selector_features = self.selector_features(enriched, masks[1], centered=centered, trunk_out=trunk_s)
    if fused_heads and k <= 16 and z.dtype == torch.float32 and states.shape[1] >= 3 + k:
      return self._forward_fused(net, z, states, raws, selector_features, is_train, progress)
    pdf, entropy = self.action_pdf(selector_features)
    random_filter_id = pdf_sample(pdf, selection_noise)
    max_filter_id = torch.argmax(pdf, dim=1).to(torch.int32)
    is_train = int(is_train)
    selected_filter_id = is_train * random_filter_id + (1 - is_train) * max_filter_id
    filter_one_hot = (selected_filter_id[:, None] == torch.arange(k, device=net.device)[None, :]).to(pdf.dtype)
    surrogate = (filter_one_hot * torch.log(pdf + 1e-10)).sum(dim=1, keepdim=True)

    if any(f.uses_generic_kernels() for f in self.filters):
      return self._forward_generic(net, states, params, mask_params, pdf, entropy, selected_filter_id, filter_one_hot,
                                   surrogate, progress, high_res)
    if fused_heads:
      params24 = F.heads_regress_select(list(self.filters), raws, selected_filter_id)
    else:
      # one-hot gather of the selected filter's packed parameters -> (N, 24); same gradient
      # routing as the reference's one-hot product over the stacked images
      params24 = net.new_zeros((n, F._cabi.EXPO_MAX_PARAMS), dtype=torch.float32)
      for j, (filt, p) in enumerate(zip(self.filters, params)):
        pj = filt.pack(p).float()
        params24 = params24 + torch.nn.functional.pad(pj, (0, F._cabi.EXPO_MAX_PARAMS - pj.shape[1])) * \
            filter_one_hot[:, j:j + 1]
    hsv_mode = int(cfg.get('hsv_grad_mode', 0))
    abi_ids = torch.where(selected_filter_id >= 0, self.abi_filter_ids[selected_filter_id.clamp_min(0).long()],
                          torch.full_like(selected_filter_id, -1))
    high_res_output = None
    if cfg.masking:
      from .util import tanh_range
      # one-hot gather of the selected filter's squashed mask parameters -> (N, 6) (filters.py:121-123)
      mask6 = net.new_zeros((n, 6), dtype=torch.float32)
      for j, mp in enumerate(mask_params):
        mask6 = mask6 + tanh_range(-5, 5, initial=0)(mp.float()) * filter_one_hot[:, j:j + 1]
      out = self._apply_masked(net, params24, mask6, abi_ids)
      overexposure = F.overexposure_penalty(out)
      if high_res is not None:
        high_res_output = self._apply_masked(high_res, params24, mask6, abi_ids)
    else:
      out, overexposure = F.dispatch_filters(net, params24, abi_ids, hsv_mode)
      if high_res is not None:
        high_res_output, _ = F.dispatch_filters(high_res, params24, abi_ids, hsv_mode)

    debug_info = {
        'state': states,
        'selected_filter_id': selected_filter_id[0],
        # (the fused training path regresses the SELECTED filter's parameters only: the per-filter debug list is empty there)
        'filter_debug_info': [{'filter_parameters': p[0]} for p in params] if params is not None else [],
        'pdf': pdf[0],
        # batched extras (not in the reference dict; used by tests / the eval loop)
        'selected_filter_ids': selected_filter_id,
        'abi_filter_ids': abi_ids,
        'pdf_batch': pdf,
        'params24': params24,
    }

    # Calculate new states (agent.py:207-238)
    is_last_step = (torch.abs(states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1] + 1 - cfg.test_steps) < 1e-4).to(
        states.dtype)
    submitted = is_last_step
    new_states = [None for _ in range(STATE_DROPOUT_BEGIN + 1)]
    new_states[STATE_REWARD_DIM] = submitted
    new_states[STATE_STOPPED_DIM] = submitted
    new_states[STATE_STEP_DIM] = (states[:, STATE_STEP_DIM] + 1)[:, None]
    filter_usage = states[:, STATE_STEP_DIM + 1:]
    early_stop_penalty = (1 - is_last_step) * submitted * cfg.early_stop_penalty
    usage_penalty = (filter_usage * filter_one_hot).sum(dim=1, keepdim=True)
    new_states[STATE_STEP_DIM + 1] = torch.maximum(filter_usage, filter_one_hot)
    new_states = torch.cat(new_states, dim=1)

    if cfg.clamp:
      # agent.py:240-241 (off in the shipped configs): only the proxy image is clipped, exactly as in
      # the reference -- high_res_output is not.  64x64 images: one tiny torch op, not a kernel.
      out = torch.clamp(out, 0.0, 5.0)
      # the penalty of agent.py:249-251 is taken on the CLIPPED image in the reference
      overexposure = F.overexposure_penalty(out)

    entropy_penalty = (1.0 - progress) * cfg.exploration_penalty * (-entropy + math.log(k))
    # Will be subtracted from the reward (agent.py:247-252)
    penalty = overexposure[:, None] + entropy_penalty + usage_penalty * cfg.filter_usage_penalty + \
        early_stop_penalty

    if high_res is None:
      return (out, new_states, surrogate, penalty), debug_info, None
    return (out, new_states, high_res_output), debug_info, None


def _new_states_and_penalty(cfg, states, filter_one_hot, entropy, overexposure, progress, k):
  """agent.py:207-252: the state update and the penalty that is subtracted from the reward."""
  is_last_step = (torch.abs(states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1] + 1 - cfg.test_steps) < 1e-4).to(states.dtype)
  submitted = is_last_step
  new_states = [None for _ in range(STATE_DROPOUT_BEGIN + 1)]
  new_states[STATE_REWARD_DIM] = submitted
  new_states[STATE_STOPPED_DIM] = submitted
  new_states[STATE_STEP_DIM] = (states[:, STATE_STEP_DIM] + 1)[:, None]
  filter_usage = states[:, STATE_STEP_DIM + 1:]
  early_stop_penalty = (1 - is_last_step) * submitted * cfg.early_stop_penalty
  usage_penalty = (filter_usage * filter_one_hot).sum(dim=1, keepdim=True)
  new_states[STATE_STEP_DIM + 1] = torch.maximum(filter_usage, filter_one_hot)
  new_states = torch.cat(new_states, dim=1)
  entropy_penalty = (1.0 - progress) * cfg.exploration_penalty * (-entropy + math.log(k))
  penalty = overexposure[:, None] + entropy_penalty + usage_penalty * cfg.filter_usage_penalty + early_stop_penalty
  return new_states, penalty


def _forward_generic(self, net, states, params, mask_params, pdf, entropy, selected_filter_id, filter_one_hot, surrogate,
                     progress, high_res):
  """The step for a configuration the per-image dispatch kernels are not instantiated for (a curve filter with
  cfg.curve_steps != 8; its parameter row would not fit EXPO_MAX_PARAMS either): the reference's own structure --
  every filter's ``apply`` on the whole batch, stack, reduce with the one-hot (agent.py:58-77, 119-125) -- with every
  filter still a HIP kernel (the curve filters on expo_curve_*).  Eight launches instead of one; 64x64 proxies."""
  cfg = self.cfg
  k = len(self.filters)

  def select(img):
    out = None
    for j, (filt, p) in enumerate(zip(self.filters, params)):
      if cfg.masking:
        from .util import lerp
        y = lerp(img.float(), filt.process(img, p).float(), filt.get_mask(img.float(), mask_params[j]))
      else:
        y = filt.process(img, p).float()
      term = y * filter_one_hot[:, j, None, None, None]
      out = term if out is None else out + term
    return out.to(img.dtype)

  out = select(net)
  high_res_output = select(high_res) if high_res is not None else None
  if cfg.clamp:
    out = torch.clamp(out, 0.0, 5.0)
  overexposure = F.overexposure_penalty(out)
  new_states, penalty = _new_states_and_penalty(cfg, states, filter_one_hot, entropy, overexposure, progress, k)
  debug_info = {
      'state': states,
      'selected_filter_id': selected_filter_id[0],
      'filter_debug_info': [{'filter_parameters': p[0]} for p in params],
      'pdf': pdf[0],
      'selected_filter_ids': selected_filter_id,
      'pdf_batch': pdf,
      'packed_params': [f.pack(p) for f, p in zip(self.filters, params)],
  }
  if high_res is None:
    return (out, new_states, surrogate, penalty), debug_info, None
  return (out, new_states, high_res_output), debug_info, None


def _forward_fused(self, net, z, states, raws, selector_features, is_train, progress):
  """The training-time step on the device with the glue fused (round 4): selector FCs -> ONE selection kernel
  (_AgentSelect) -> ONE regress-and-gather kernel (filters.heads_regress_select) -> ONE dispatch kernel with the
  over-exposure penalty.  Same values as the op-by-op path (tests/test_hip_agent.py); cfg.masking off, 8-step curves."""
  cfg = self.cfg
  k = len(self.filters)
  logits = self.selector_fc2(lrelu(self.selector_fc1(selector_features)))
  prog = progress if torch.is_tensor(progress) else torch.full((1,), float(progress), dtype=torch.float32, device=net.device)
  consts = (cfg.exploration, cfg.exploration_penalty, cfg.filter_usage_penalty, cfg.early_stop_penalty, cfg.test_steps)
  pdf, entropy, selected, one_hot, surrogate, new_states, pen_base = _AgentSelect.apply(
      logits, z, states, prog.reshape(1).float(), consts, int(is_train))
  params24 = F.heads_regress_select(list(self.filters), raws, selected)
  if self._abi_identity:
    abi_ids = selected
  else:
    abi_ids = torch.where(selected >= 0, self.abi_filter_ids[selected.clamp_min(0).long()], torch.full_like(selected, -1))
  out, overexposure = F.dispatch_filters(net, params24, abi_ids, int(cfg.get('hsv_grad_mode', 0)))
  if cfg.clamp:
    out = torch.clamp(out, 0.0, 5.0)
    overexposure = F.overexposure_penalty(out)
  penalty = overexposure[:, None] + pen_base
  debug_info = {
      'state': states,
      'selected_filter_id': selected[0],
      'filter_debug_info': [],  # (only the selected filter's parameters are regressed on this path)
      'pdf': pdf[0],
      'selected_filter_ids': selected,
      'abi_filter_ids': abi_ids,
      'pdf_batch': pdf,
      'params24': params24,
  }
  return (out, new_states, surrogate, penalty), debug_info, None


Agent._forward_generic = _forward_generic
Agent._forward_fused = _forward_fused


def agent_generator(inp, is_train, progress, cfg, high_res=None, alex_in=None, module=None, dropout_masks=None):
  """agent.py:41 signature; ``module`` is the :class:`Agent` holding the weights (the reference
  keeps them in the TF variable scope 'generator')."""
  assert module is not None, 'pass the Agent module that owns the generator weights'
  return module(inp, is_train, progress, cfg=cfg, high_res=high_res, dropout_masks=dropout_masks)
