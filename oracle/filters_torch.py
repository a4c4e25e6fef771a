"""torch-CPU op-by-op restatement of the reference's filters (autograd backward).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); PARITY UNPINNED by the
reference.  This file is the *second, independent* restatement: one torch op
per TF op of ``/root/reference/filters.py`` (same op granularity at which the
TF-1 graph executes), gradients by autograd.  It serves two purposes:

1. cross-check of ``oracle/filters_np.py``'s hand-derived backward
   (``tests/test_oracle_filters.py``);
2. the timed "CPU restatement (torch fp32, T threads)" leg of ``bench.py``
   (``cpu_baseline.kind = "port"``), as planned in BASELINE.md section 3.

TF gradient conventions are kept by using ``clamp`` / ``clamp_min`` /
``clamp_max`` (inclusive pass-through, like ``tf.clip_by_value`` /
``tf.maximum(x, const)`` / ``tf.minimum(x, const)``) instead of
``torch.maximum(x, tensor)`` which splits ties.
"""
import math

import torch

FILTER_NAMES = ('E', 'G', 'W', 'S+', 'T', 'Ct', 'BW', 'C', 'Le')
NUM_PARAMS = (1, 1, 3, 1, 8, 1, 1, 24, 2)
CURVE_STEPS = 8


def rgb2lum(image):
  """util.py:271-274."""
  image = 0.27 * image[:, :, :, 0] + 0.67 * image[:, :, :, 1] + 0.06 * image[:, :, :, 2]
  return image[:, :, :, None]


def lerp(a, b, l):
  """util.py:307-308."""
  return (1 - l) * a + l * b


def lrelu(x, leak=0.2):
  """util.py:225-229."""
  f1 = 0.5 * (1 + leak)
  f2 = 0.5 * (1 - leak)
  return f1 * x + f2 * torch.abs(x)


def tanh_range(l, r, initial=None):
  """util.py:281-294."""

  def activation(x):
    bias = math.atanh(2 * (initial - l) / (r - l) - 1) if initial is not None else 0
    return (torch.tanh(x + bias) * 0.5 + 0.5) * (r - l) + l

  return activation


def rgb_to_hsv(rgb):
  """tf.image.rgb_to_hsv (filters.py:486)."""
  r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
  v = rgb.max(dim=-1).values
  rng = v - rgb.min(dim=-1).values
  one = torch.ones_like(v)
  s = torch.where(v > 0, rng / torch.where(v > 0, v, one), torch.zeros_like(v))
  norm = 1.0 / (6.0 * torch.where(rng > 0, rng, one))
  h = torch.where(r == v, norm * (g - b),
                  torch.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
  h = torch.where(rng > 0, h, torch.zeros_like(h))
  h = torch.where(h < 0, h + 1, h)
  return torch.stack([h, s, v], dim=-1)


def hsv_to_rgb(hsv):
  """tf.image.hsv_to_rgb (filters.py:492)."""
  h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
  dh = h * 6
  dr = torch.clamp(torch.abs(dh - 3) - 1, 0, 1)
  dg = torch.clamp(2 - torch.abs(dh - 2), 0, 1)
  db = torch.clamp(2 - torch.abs(dh - 4), 0, 1)
  oms = 1 - s
  return torch.stack([(oms + s * dr) * v, (oms + s * dg) * v, (oms + s * db) * v], dim=-1)


# -- process(): reference-shaped params --------------------------------------
def exposure_process(img, param):
  """filters.py:181-182."""
  return img * torch.exp(param[:, None, None, :] * math.log(2))


def gamma_process(img, param):
  """filters.py:205-206."""
  return torch.pow(torch.clamp_min(img, 0.001), param[:, None, None, :])


def wb_process(img, param):
  """filters.py:237-238."""
  return img * param[:, None, None, :]


def _curve_process(img, param, L=None):
  L = param.shape[4] if L is None else L  # cfg.curve_steps
  curve_sum = torch.sum(param, dim=4) + 1e-30
  total = img * 0
  for i in range(L):
    total = total + torch.clamp(img - 1.0 * i / L, 0, 1.0 / L) * param[:, :, :, :, i]
  total = total * (L / curve_sum)
  return total


def color_process(img, param, L=None):
  """filters.py:264-273."""
  return _curve_process(img, param, L)


def tone_process(img, param, L=None):
  """filters.py:312-322."""
  return _curve_process(img, param, L)


def contrast_process(img, param):
  """filters.py:415-419."""
  luminance = torch.clamp_max(torch.clamp_min(rgb2lum(img), 0.0), 1.0)
  contrast_lum = -torch.cos(math.pi * luminance) * 0.5 + 0.5
  contrast_image = img / (luminance + 1e-6) * contrast_lum
  return lerp(img, contrast_image, param[:, :, None, None])


def wnb_process(img, param):
  """filters.py:438-440."""
  return lerp(img, rgb2lum(img), param[:, :, None, None])


def satplus_process(img, param, hsv_grad_mode=0):
  """filters.py:484-498.  hsv_grad_mode=0: TF-1.x (HSV ops not differentiable)."""
  img = torch.clamp_max(img, 1.0)
  src = img.detach() if hsv_grad_mode == 0 else img
  hsv = rgb_to_hsv(src)
  s = hsv[:, :, :, 1:2]
  v = hsv[:, :, :, 2:3]
  enhanced_s = s + (1 - s) * (0.5 - torch.abs(0.5 - v)) * 0.8
  hsv1 = torch.cat([hsv[:, :, :, 0:1], enhanced_s, hsv[:, :, :, 2:]], dim=3)
  full_color = hsv_to_rgb(hsv1)
  param = param[:, :, None, None]
  return img * (1.0 - param) + full_color * param


def level_process(img, param):
  """filters.py:460-466."""
  lower = param[:, 0]
  upper = param[:, 1] + 1
  lower = lower[:, None, None, None]
  upper = upper[:, None, None, None]
  return torch.clamp((img - lower) / (upper - lower + 1e-6), 0.0, 1.0)


def get_mask(img, mask_parameters, maximum_sharpness=1, minimum_strength=0.3):
  """filters.py:110-148 with cfg.masking = True; mask_parameters RAW (N,6)."""
  filter_input_range = 5
  mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
  h, w = img.shape[1], img.shape[2]
  se = min(h, w)
  gi = ((torch.arange(h, dtype=torch.float64) + (se - h) / 2.0) / se - 0.5).float().to(img.dtype)
  gj = ((torch.arange(w, dtype=torch.float64) + (se - w) / 2.0) / se - 0.5).float().to(img.dtype)
  g0 = gi[None, :, None, None]
  g1 = gj[None, None, :, None]
  inp = g0 * mp[:, None, None, 0, None] + g1 * mp[:, None, None, 1, None] + \
      mp[:, None, None, 2, None] * (rgb2lum(img) - 0.5) + mp[:, None, None, 3, None] * 2
  inp = inp * (maximum_sharpness * mp[:, None, None, 4, None] / filter_input_range)
  mask = torch.sigmoid(inp)
  mask = mask * (mp[:, None, None, 5, None] / filter_input_range * 0.5 + 0.5) * (1 - minimum_strength) + \
      minimum_strength
  return mask


def vignet_mask(img, mask_parameters, maximum_sharpness=1, masking=True):
  """VignetFilter.get_mask (filters.py:360-396); mask_parameters RAW (N,5)."""
  filter_input_range = 5
  mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
  h, w = img.shape[1], img.shape[2]
  se = min(h, w)
  gi = ((torch.arange(h, dtype=torch.float64) + (se - h) / 2.0) / se - 0.5).float().to(img.dtype)
  gj = ((torch.arange(w, dtype=torch.float64) + (se - w) / 2.0) / se - 0.5).float().to(img.dtype)
  inp = (gi[None, :, None, None] * mp[:, None, None, 0, None])**2 + \
      (gj[None, None, :, None] * mp[:, None, None, 1, None])**2 + mp[:, None, None, 2, None] - filter_input_range
  inp = inp * (maximum_sharpness * mp[:, None, None, 3, None] / filter_input_range)
  mask = torch.sigmoid(inp) * (mp[:, None, None, 4, None] / filter_input_range * 0.5 + 0.5)
  if not masking:
    mask = mask * 0 + 1  # filters.py:390-392
  return mask


def vignet_apply(img, mask_parameters, maximum_sharpness=1, masking=True):
  """VignetFilter.apply: process = img * 0 (filters.py:351-352), out = lerp(img, 0, mask) (filters.py:86-88)."""
  return lerp(img, img * 0, vignet_mask(img, mask_parameters, maximum_sharpness, masking))


def apply_masked(fid, img, packed, mask_parameters, maximum_sharpness=1, minimum_strength=0.3, hsv_grad_mode=0):
  """Filter.apply with cfg.masking = True (filters.py:86-88)."""
  mask = get_mask(img, mask_parameters, maximum_sharpness, minimum_strength)
  return lerp(img, process_packed(fid, img, packed, hsv_grad_mode), mask)


def apply_masked_backward(fid, img, packed, mask_parameters, dy, maximum_sharpness=1, minimum_strength=0.3,
                          hsv_grad_mode=0):
  img = img.detach().clone().requires_grad_(True)
  packed = packed.detach().clone().requires_grad_(True)
  mask_parameters = mask_parameters.detach().clone().requires_grad_(True)
  y = apply_masked(fid, img, packed, mask_parameters, maximum_sharpness, minimum_strength, hsv_grad_mode)
  return (y.detach(),) + torch.autograd.grad(y, [img, packed, mask_parameters], dy)


def unpack_params(fid, packed):
  n = packed.shape[0]
  if FILTER_NAMES[fid] == 'T':
    return packed.reshape(n, 1, 1, 1, packed.shape[1])
  if FILTER_NAMES[fid] == 'C':
    return packed.reshape(n, 1, 1, 3, packed.shape[1] // 3)
  return packed


def process_packed(fid, img, packed, hsv_grad_mode=0):
  p = unpack_params(fid, packed)
  name = FILTER_NAMES[fid]
  if name == 'E':
    return exposure_process(img, p)
  if name == 'G':
    return gamma_process(img, p)
  if name == 'W':
    return wb_process(img, p)
  if name == 'S+':
    return satplus_process(img, p, hsv_grad_mode)
  if name == 'T':
    return tone_process(img, p)
  if name == 'Ct':
    return contrast_process(img, p)
  if name == 'BW':
    return wnb_process(img, p)
  if name == 'C':
    return color_process(img, p)
  if name == 'Le':
    return level_process(img, p)
  raise ValueError(fid)


def backward_packed(fid, img, packed, dy, hsv_grad_mode=0):
  """(dx, dpacked) via autograd."""
  img = img.detach().clone().requires_grad_(True)
  packed = packed.detach().clone().requires_grad_(True)
  y = process_packed(fid, img, packed, hsv_grad_mode)
  dx, dp = torch.autograd.grad(y, [img, packed], dy)
  return dx, dp


def chain_fwd_bwd(x0, packed_list, dy, filter_ids=tuple(range(8)), hsv_grad_mode=0,
                  round_dtype=None):
  """The benchmark construct: filters applied sequentially
  in cfg.filters order, forward + backward.  ``round_dtype`` (e.g. torch.float16)
  rounds every step's output/gradient the way the fp16-storage GPU path does,
  with a straight-through gradient."""
  x = x0.detach().clone().requires_grad_(True)
  ps = [p.detach().clone().requires_grad_(True) for p in packed_list]
  cur = x
  for fid, p in zip(filter_ids, ps):
    cur = process_packed(fid, cur, p, hsv_grad_mode)
    if round_dtype is not None:
      cur = cur + (cur.detach().to(round_dtype).to(cur.dtype) - cur.detach())
  grads = torch.autograd.grad(cur, [x] + ps, dy)
  return cur.detach(), grads[0], list(grads[1:])
