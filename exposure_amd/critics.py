"""WGAN-GP critic / value network (``/root/reference/critics.py``) on PyTorch-ROCm.

``critic(images, cfg, states=None)`` = three per-image statistics (critics.py:48-62) broadcast
as constant feature planes (+ optional state planes) -> ``cnn`` (critics.py:6-38: ``x - 0.5``,
four conv4x4/s2 + lrelu, no norm) -> FC128 lrelu -> FC1.

The statistics are computed with differentiable torch ops inside the training graph (the
gradient-penalty term needs a double backward through everything the critic does to its input,
net.py:174-194); :func:`critic_stats` is the fused HIP reduction (``expo_critic_stats``) for
inference-side callers.  ``lrelu`` (``util.py:225-229``: ``0.6 x + 0.4 |x|``) is one fused forward kernel with a
differentiable backward (``exposure_amd.util._Lrelu``), so the double backward exists.
"""
import torch
from torch import nn

from . import _cabi
from .util import lrelu


def stat_features(images):
  """critics.py:48-73 on NHWC ``images`` -> (N, 3) [mean lum, var lum, mean saturation]."""
  images = images.float()
  lum = images[:, :, :, 0] * 0.27 + images[:, :, :, 1] * 0.67 + images[:, :, :, 2] * 0.06 + 1e-5
  luminance = lum.mean(dim=(1, 2))
  contrast = lum.var(dim=(1, 2), unbiased=False)  # tf.nn.moments: population variance
  clipped = images.clamp(0.0, 1.0)
  # amax/amin split the gradient evenly between tied maxima, like tf.reduce_max/min
  i_max = clipped.amax(dim=3)
  i_min = clipped.amin(dim=3)
  a, b = i_max + i_min, 2.0 - i_max - i_min
  sat = (i_max - i_min) / (torch.where(a <= b, a, b) + 1e-2)  # tf.minimum(x, y): ties go to x
  saturation = sat.mean(dim=(1, 2))
  return torch.stack([luminance, contrast, saturation], dim=1)


def critic_stats(images):
  """Same three statistics through the HIP reduction kernel (no autograd)."""
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
    n, h, w, _ = images.shape
    planes = states[:, None, None, :].expand(n, h, w, states.shape[1])
    net = torch.cat([images, planes], dim=3)
    net = (net - 0.5).permute(0, 3, 1, 2)  # NHWC storage, channels_last view
    for conv in self.convs:
      net = lrelu(conv(net))
    net = net.permute(0, 2, 3, 1).reshape(n, self.flat)
    net = lrelu(self.fc1(net))
    return self.fc2(net)


def critic(images, cfg, states=None, is_train=None, reuse=False, module=None):
  """critics.py:42 signature -> (outputs, None, None); ``module`` owns the weights."""
  assert module is not None
  return module(images, states), None, None
