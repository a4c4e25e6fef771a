"""WGAN-GP critic / value network (``/root/reference/critics.py``) on PyTorch-ROCm.

``critic(images, cfg, states=None)`` = three per-image statistics (critics.py:48-62) broadcast
as constant feature planes (+ optional state planes) -> ``cnn`` (critics.py:6-38: ``x - 0.5``,
four conv4x4/s2 + lrelu, no norm) -> FC128 lrelu -> FC1.

The statistics are HIP reductions INSIDE the training graph (:func:`stat_features`): forward
``expo_critic_stats``, first derivative ``expo_critic_stats_bwd`` (the generator's reward flows through
``critic(fake_output)``, net.py:68-90), and the derivative of that derivative -- the gradient-penalty term
differentiates ``D(x^)`` with respect to ``x^`` and then with respect to the critic's weights
(net.py:174-194) -- ``expo_critic_stats_jvp`` / ``expo_critic_stats_hvp``.  ~15 torch element-wise /
reduction launches forward, ~30 backward and ~60 in the double backward become 2 + 1 + 3.
``lrelu`` (``util.py:225-229``: ``0.6 x + 0.4 |x|``) is one fused forward kernel with a differentiable
backward (``exposure_amd.util._Lrelu``), so the double backward exists there too.
"""
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import _cabi
from .nn_ops import conv_trunk, planes_concat
from .util import lrelu


class _CriticStatsGrad(torch.autograd.Function):
  """dx = J^T g for the statistics' Jacobian J (``expo_critic_stats_bwd``); its own backward -- the double
  backward of the WGAN-GP term -- is J v for g (``expo_critic_stats_jvp``: the path to the critic's weights)
  and the second-order term d<J^T g, v>/dx for the image (``expo_critic_stats_hvp``)."""

  @staticmethod
  def forward(ctx, x, g, stats):
    g = g.contiguous().float()
    dx = torch.empty_like(x)
    _cabi.critic_stats_bwd(x, stats, g, dx)
    ctx.save_for_backward(x, g, stats)
    return dx

  @staticmethod
  @once_differentiable
  def backward(ctx, v):
    x, g, stats = ctx.saved_tensors
    v = v.contiguous().to(x.dtype)
    jv = torch.empty_like(g)
    _cabi.critic_stats_jvp(x, stats, v, jv)
    gx = None
    if ctx.needs_input_grad[0]:
      gx = torch.empty_like(x)
      _cabi.critic_stats_hvp(x, g, jv, v, gx)
    return gx, jv, None


class _CriticStats(torch.autograd.Function):
  """stats = [mean lum, var lum, mean saturation] per image (``expo_critic_stats``), twice differentiable."""

  @staticmethod
  def forward(ctx, x):
    stats = torch.empty((x.shape[0], 3), dtype=torch.float32, device=x.device)
    _cabi.critic_stats(x, stats)
    ctx.save_for_backward(x, stats)
    return stats

  @staticmethod
  def backward(ctx, g):
    x, stats = ctx.saved_tensors
    # the mean the first derivative reads is a function of x too: expo_critic_stats_hvp carries that dependence,
    # so `stats` enters as a constant here
    return _CriticStatsGrad.apply(x, g, stats.detach())


def stat_features(images):
  """critics.py:48-73 on NHWC ``images`` -> (N, 3) [mean lum, var lum, mean saturation], through the HIP
  reductions, differentiable twice.  The gradients are O(1 / (H W)) per pixel, far below fp16's normal range,
  so the image enters as float32 (the reference's dtype)."""
  return _CriticStats.apply(images.float().contiguous())


def critic_stats(images):
  """The same three statistics without autograd, on the image's own dtype (inference-side callers)."""
  stats = torch.empty((images.shape[0], 3), dtype=torch.float32, device=images.device)
  _cabi.critic_stats(images.contiguous(), stats)
  return stats


class Critic(nn.Module):
  """critics.py:42-98.  ``num_state_dim`` = 0 for the critic, ``cfg.num_state_dim`` for the value
  network (net.py:76-90 calls the same function with ``states=``)."""

  def __init__(self, cfg, num_state_dim=0):
    super().__init__()
    self.cfg = cfg
    self.num_state_dim = num_state_dim
    in_ch = cfg.real_img_channels + num_state_dim + 3
    channels = cfg.base_channels
    size = cfg.source_img_size // 2
    convs = [nn.Conv2d(in_ch, channels, kernel_size=4, stride=2, padding=1)]
    while size > 4:
      convs.append(nn.Conv2d(channels, channels * 2, kernel_size=4, stride=2, padding=1))
      channels *= 2
      size //= 2
    self.convs = nn.ModuleList(convs)
    self.flat = 4 * 4 * channels
    self.fc1 = nn.Linear(self.flat, cfg.fc1_size)
    self.fc2 = nn.Linear(cfg.fc1_size, 1)
    for m in list(self.convs) + [self.fc1, self.fc2]:
      nn.init.xavier_uniform_(m.weight)
      nn.init.zeros_(m.bias)
    self.to(memory_format=torch.channels_last)

  def forward(self, images, states=None):
    images = images.float()
    stat_feature = stat_features(images)
    if states is None:
      states = stat_feature
    else:
      assert states.dim() == stat_feature.dim()
      states = torch.cat([states.float(), stat_feature], dim=1)
    n = images.shape[0]
    # image channels + state / statistics planes, minus 0.5, as ONE launch on the device (nn_ops.planes_concat);
    # NHWC: the convolutions see channels_last views
    net = planes_concat(images, states, 0.5)
    net = conv_trunk(net, self.convs)
    net = net.reshape(n, self.flat)
    net = lrelu(self.fc1(net))
    return self.fc2(net)


def critic(images, cfg, states=None, is_train=None, reuse=False, module=None):
  """critics.py:42 signature -> (outputs, None, None); ``module`` owns the weights."""
  assert module is not None
  return module(images, states), None, None
