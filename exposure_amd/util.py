"""Host-side helpers mirrored from the reference's ``util.py`` (torch, any device).

These are the tiny scalar / activation helpers the regressors and the convnets use
(``/root/reference/util.py:13-16, 31-36, 40-72, 225-229, 271-294, 307-308``); the per-pixel
image maths itself lives in the HIP library, not here.
"""
import math

import torch

# util.py:13-16
STATE_REWARD_DIM = 0
STATE_STOPPED_DIM = 1
STATE_STEP_DIM = 2
STATE_DROPOUT_BEGIN = 3


class Dict(dict):
  """Attribute-style dict used for ``cfg`` (util.py:40-72)."""

  def __getattr__(self, key):
    try:
      return self[key]
    except KeyError as e:
      raise AttributeError(key) from e

  def __setattr__(self, key, value):
    self[key] = value

  def __delattr__(self, key):
    del self[key]


def lrelu(x, leak=0.2):
  """util.py:225-229: ``f1*x + f2*|x|`` with f1 = (1+leak)/2, f2 = (1-leak)/2, i.e. the leaky ReLU
  ``x if x > 0 else leak*x`` (agrees with the literal formula to 1 ulp: 0.6x + 0.4x vs x), with the reference's
  sub-gradient f1 at x == 0 (``tf.abs`` has gradient sign(0) = 0 there; ``leaky_relu``'s own backward would give
  ``leak``).  One HIP launch forward and one per backward (``exposure_amd.nn_ops``: ``expo_bias_lrelu_fwd`` /
  ``expo_lrelu_bwd``); the backward is linear in the incoming gradient and differentiable again, so the double
  backward the WGAN-GP term needs (net.py:174-194) exists."""
  from .nn_ops import bias_lrelu
  return bias_lrelu(x, None, leak)


def rgb2lum(image):
  """util.py:271-274 (NHWC)."""
  lum = 0.27 * image[:, :, :, 0] + 0.67 * image[:, :, :, 1] + 0.06 * image[:, :, :, 2]
  return lum[:, :, :, None]


def tanh01(x):
  """util.py:277-278."""
  return torch.tanh(x) * 0.5 + 0.5


def tanh_range(l, r, initial=None):
  """util.py:281-294."""

  def activation(x):
    if initial is not None:
      bias = math.atanh(2 * (initial - l) / (r - l) - 1)
    else:
      bias = 0
    return tanh01(x + bias) * (r - l) + l

  return activation


def lerp(a, b, l):
  """util.py:307-308."""
  return (1 - l) * a + l * b


def enrich_image_input(cfg, net, states):
  """util.py:31-36: broadcast the state vector to H x W planes and append as channels (NHWC)."""
  if cfg.img_include_states:
    n, h, w, _ = net.shape
    planes = states[:, None, None, :].to(net.dtype).expand(n, h, w, states.shape[1])
    net = torch.cat([net, planes], dim=3)
  return net


class capture_without_gc:
  """Around a hipGraph capture: collect Python garbage FIRST, keep the cyclic collector off while the stream captures.

  A dead reference cycle that owns device resources -- an earlier ``GAN`` with captured step graphs is one: module ->
  gradient buckets -> their ready callback -> module -- is freed whenever the cyclic collector happens to run.  If that
  is in the middle of a capture, destroying its ``CUDAGraph`` objects (hipGraphExecDestroy, the graph's memory pool) is
  an operation the runtime refuses while a stream captures; the error is thrown from a destructor and the process
  ABORTS (found by a gpu-suite run that died inside ``test_train_cli_runs``, "Garbage-collecting" on top of the fault
  handler's stack: round 5).  ``torch.cuda.graph`` used to collect before every capture; since it became optional
  (``torch.compiler.config.force_cudagraph_gc``, off by default) nothing does."""

  def __enter__(self):
    import gc
    gc.collect()
    self._was_enabled = gc.isenabled()
    gc.disable()
    return self

  def __exit__(self, *exc):
    import gc
    if self._was_enabled:
      gc.enable()
    return False
