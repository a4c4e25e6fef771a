"""NumPy restatement of the reference's per-pixel filters (forward AND backward).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); PARITY UNPINNED by the
reference (it has no tests) -- pinned by identities / known answers / finite
differences in ``tests/test_oracle_filters.py``.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  Images are NHWC ``(N, H, W, 3)``.  Arithmetic runs in the
dtype of ``img`` promoted to at least float32 (pass float64 arrays for the
"exact" oracle, float32 arrays to mimic the reference's TF float32 graph
op-for-op).

Two parameter conventions are provided:

* reference-shaped parameters, exactly what ``filter_param_regressor`` returns
  in the reference (``E,G: (N,1)``; ``W: (N,3)``; ``S+,Ct,BW: (N,1)``;
  ``T: (N,1,1,1,8)``; ``C: (N,1,1,3,8)``);
* *packed* parameters ``(N, P)`` float32, the C-ABI layout
  (``include/exposure_hip.h``): ``P = 1,1,3,1,8,1,1,24`` for filter ids
  ``0..7 = E,G,W,S+,T,Ct,BW,C`` (order of ``cfg.filters``,
  ``config_example.py:22-25``); Color packs ``channel*8 + knot``.

The backward functions are hand-derived (they are what the HIP kernels
implement); ``oracle/filters_torch.py`` obtains the same gradients from
autograd on an op-by-op transcription, and the tests require both to agree.
TF gradient conventions encoded here: ``tf.maximum(x, c)`` / ``tf.minimum(x,
c)`` pass the gradient to ``x`` on equality, ``tf.clip_by_value`` passes on
``lo <= x <= hi`` (both inclusive), ``abs'(0) = 0``, and TF-1.x registers
``RGBToHSV``/``HSVToRGB`` as not differentiable (``hsv_grad_mode=0``).
"""
import math

import numpy as np

# cfg.filters order, config_example.py:22-25
FILTER_NAMES = ('E', 'G', 'W', 'S+', 'T', 'Ct', 'BW', 'C', 'Le')  # 'Le' (LevelFilter) is not in cfg.filters
FILTER_ID = {n: i for i, n in enumerate(FILTER_NAMES)}
NUM_PARAMS = (1, 1, 3, 1, 8, 1, 1, 24, 2)
CURVE_STEPS = 8  # cfg.curve_steps, config_example.py:27
LUM_W = (0.27, 0.67, 0.06)  # util.py:271-274

DEFAULT_CFG = dict(
    curve_steps=8,
    gamma_range=3,
    exposure_range=3.5,
    color_curve_range=(0.90, 1.10),
    tone_curve_range=(0.5, 2),
)


def _ft(a):
  a = np.asarray(a)
  return a.dtype if a.dtype in (np.float32, np.float64) else np.dtype(np.float32)


# ---------------------------------------------------------------------------
# helpers: util.py:271-308
# ---------------------------------------------------------------------------
def rgb2lum(image):
  """util.py:271-274 -- keeps a trailing singleton channel."""
  dt = _ft(image)
  lum = dt.type(0.27) * image[..., 0] + dt.type(0.67) * image[..., 1] + dt.type(
      0.06) * image[..., 2]
  return lum[..., None]


def tanh01(x):
  """util.py:277-278."""
  dt = _ft(x)
  return np.tanh(x) * dt.type(0.5) + dt.type(0.5)


def tanh_range(l, r, initial=None):
  """util.py:281-294."""

  def activation(x):
    dt = _ft(x)
    if initial is not None:
      bias = math.atanh(2 * (initial - l) / (r - l) - 1)
    else:
      bias = 0
    return tanh01(x + dt.type(bias)) * dt.type(r - l) + dt.type(l)

  return activation


def lerp(a, b, l):
  """util.py:307-308."""
  return (1 - l) * a + l * b


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


# ---------------------------------------------------------------------------
# regressors (reference-shaped outputs)
# ---------------------------------------------------------------------------
def exposure_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:177-179."""
  return tanh_range(-cfg['exposure_range'], cfg['exposure_range'], initial=0)(f)


def gamma_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:201-203."""
  lg = np.log(cfg['gamma_range'])
  return np.exp(tanh_range(-lg, lg)(f))


def wb_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:224-235."""
  dt = _ft(f)
  mask = np.array((0, 1, 1), dtype=dt).reshape(1, 3)
  f = f * mask
  s = np.exp(tanh_range(-0.5, 0.5)(f))
  s = s * (dt.type(1.0) / (dt.type(1e-5) + dt.type(0.27) * s[:, 0] + dt.type(0.67) *
                           s[:, 1] + dt.type(0.06) * s[:, 2]))[:, None]
  return s


def color_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:256-262 -> (N,1,1,3,8)."""
  L = cfg['curve_steps']
  c = np.reshape(f, (-1, 3, L))[:, None, None, :]
  return tanh_range(*cfg['color_curve_range'], initial=1)(c)


def tone_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:306-310 -> (N,1,1,1,8)."""
  L = cfg['curve_steps']
  c = np.reshape(f, (-1, 1, L))[:, None, None, :]
  return tanh_range(*cfg['tone_curve_range'])(c)


def contrast_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:411-413."""
  return np.tanh(f)


def wnb_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:435-436."""
  return sigmoid(f)


def satplus_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:481-482."""
  return sigmoid(f)


def level_regressor(f, cfg=DEFAULT_CFG):
  """filters.py:457-458."""
  return sigmoid(f)


REGRESSORS = (exposure_regressor, gamma_regressor, wb_regressor, satplus_regressor,
              tone_regressor, contrast_regressor, wnb_regressor, color_regressor, level_regressor)


# ---------------------------------------------------------------------------
# TF image ops restated (tensorflow/core/kernels/adjust_hsv / colorspace_op.h;
# TensorFlow is an un-vendored, unpinned dependency: README.md:25)
# ---------------------------------------------------------------------------
def rgb_to_hsv(rgb):
  """tf.image.rgb_to_hsv as called at filters.py:486."""
  dt = _ft(rgb)
  r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
  v = np.maximum(np.maximum(r, g), b)
  mn = np.minimum(np.minimum(r, g), b)
  rng = v - mn
  with np.errstate(divide='ignore', invalid='ignore'):
    s = np.where(v > 0, rng / np.where(v > 0, v, 1), dt.type(0))
    norm = dt.type(1.0) / (dt.type(6.0) * np.where(rng > 0, rng, 1))
    h = np.where(r == v, norm * (g - b),
                 np.where(g == v, norm * (b - r) + dt.type(2.0 / 6.0),
                          norm * (r - g) + dt.type(4.0 / 6.0)))
  h = np.where(rng > 0, h, dt.type(0))
  h = np.where(h < 0, h + dt.type(1), h)
  return np.stack([h, s, v], axis=-1)


def hsv_to_rgb(hsv):
  """tf.image.hsv_to_rgb as called at filters.py:492."""
  dt = _ft(hsv)
  h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
  dh = h * dt.type(6)
  dr = np.clip(np.abs(dh - dt.type(3)) - dt.type(1), 0, 1)
  dg = np.clip(dt.type(2) - np.abs(dh - dt.type(2)), 0, 1)
  db = np.clip(dt.type(2) - np.abs(dh - dt.type(4)), 0, 1)
  oms = dt.type(1) - s
  return np.stack([(oms + s * dr) * v, (oms + s * dg) * v, (oms + s * db) * v], axis=-1)


# ---------------------------------------------------------------------------
# process(): forward, reference-shaped params
# ---------------------------------------------------------------------------
def exposure_process(img, param):
  """filters.py:181-182."""
  dt = _ft(img)
  return img * np.exp(param[:, None, None, :] * dt.type(np.log(2)))


def gamma_process(img, param):
  """filters.py:205-206."""
  dt = _ft(img)
  return np.power(np.maximum(img, dt.type(0.001)), param[:, None, None, :])


def wb_process(img, param):
  """filters.py:237-238."""
  return img * param[:, None, None, :]


def _curve_process(img, param, L):
  """Shared body of filters.py:264-273 (Color) and 312-322 (Tone)."""
  dt = _ft(img)
  curve_sum = np.sum(param, axis=4) + dt.type(1e-30)
  total = img * 0
  for i in range(L):
    total = total + np.clip(img - dt.type(1.0 * i / L), 0, dt.type(1.0 / L)) * param[:, :, :, :, i]
  total = total * (dt.type(L) / curve_sum)
  return total


def color_process(img, param, L=None):
  """filters.py:264-273; param (N,1,1,3,L), L = cfg.curve_steps (taken from the parameter's last dimension)."""
  return _curve_process(img, param, param.shape[4] if L is None else L)


def tone_process(img, param, L=None):
  """filters.py:312-322; param (N,1,1,1,L)."""
  return _curve_process(img, param, param.shape[4] if L is None else L)


def contrast_process(img, param):
  """filters.py:415-419; param (N,1)."""
  dt = _ft(img)
  luminance = np.minimum(np.maximum(rgb2lum(img), dt.type(0.0)), dt.type(1.0))
  contrast_lum = -np.cos(dt.type(math.pi) * luminance) * dt.type(0.5) + dt.type(0.5)
  contrast_image = img / (luminance + dt.type(1e-6)) * contrast_lum
  return lerp(img, contrast_image, param[:, :, None, None])


def wnb_process(img, param):
  """filters.py:438-440; param (N,1)."""
  luminance = rgb2lum(img)
  return lerp(img, luminance, param[:, :, None, None])


def satplus_full_color(img):
  """filters.py:485-492: returns (clamped img, full_color)."""
  dt = _ft(img)
  img = np.minimum(img, dt.type(1.0))
  hsv = rgb_to_hsv(img)
  s = hsv[..., 1:2]
  v = hsv[..., 2:3]
  enhanced_s = s + (1 - s) * (dt.type(0.5) - np.abs(dt.type(0.5) - v)) * dt.type(0.8)
  hsv1 = np.concatenate([hsv[..., 0:1], enhanced_s, hsv[..., 2:]], axis=3)
  return img, hsv_to_rgb(hsv1)


def satplus_process(img, param):
  """filters.py:484-498; param (N,1).  NB the blend uses the CLAMPED image
  (``img`` is reassigned at filters.py:485)."""
  img, full_color = satplus_full_color(img)
  param = param[:, :, None, None]
  return img * (1.0 - param) + full_color * param


def level_process(img, param):
  """filters.py:460-466; param (N,2)."""
  dt = _ft(img)
  lower = param[:, 0]
  upper = param[:, 1] + 1
  lower = lower[:, None, None, None]
  upper = upper[:, None, None, None]
  return np.clip((img - lower) / (upper - lower + dt.type(1e-6)), 0.0, 1.0)


PROCESS = (exposure_process, gamma_process, wb_process, satplus_process, tone_process,
           contrast_process, wnb_process, color_process, level_process)


def mask_grid(h, w, dtype=np.float64):
  """filters.py:124-133: the constant (1,H,W,2) coordinate grid."""
  se = min(h, w)
  gi = (np.arange(h, dtype=np.float64) + (se - h) / 2.0) / se - 0.5
  gj = (np.arange(w, dtype=np.float64) + (se - w) / 2.0) / se - 0.5
  grid = np.zeros((1, h, w, 2), dtype=np.float32)  # the reference builds it in float32
  grid[0, :, :, 0] = gi[:, None]
  grid[0, :, :, 1] = gj[None, :]
  return grid.astype(dtype)


def get_mask(img, mask_parameters, maximum_sharpness=1, minimum_strength=0.3):
  """filters.py:110-148 with cfg.masking = True; mask_parameters (N,6) RAW (pre tanh_range)."""
  dt = _ft(img)
  filter_input_range = 5
  mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
  grid = mask_grid(img.shape[1], img.shape[2], dt)
  inp = grid[:, :, :, 0, None] * mp[:, None, None, 0, None] + \
      grid[:, :, :, 1, None] * mp[:, None, None, 1, None] + \
      mp[:, None, None, 2, None] * (rgb2lum(img) - dt.type(0.5)) + \
      mp[:, None, None, 3, None] * 2
  inp = inp * (maximum_sharpness * mp[:, None, None, 4, None] / filter_input_range)
  mask = sigmoid(inp)
  mask = mask * (mp[:, None, None, 5, None] / filter_input_range * dt.type(0.5) + dt.type(0.5)) * \
      (1 - minimum_strength) + minimum_strength
  return mask


def apply_masked(fid, img, packed, mask_parameters, maximum_sharpness=1, minimum_strength=0.3):
  """Filter.apply with cfg.masking = True (filters.py:86-88)."""
  mask = get_mask(img, mask_parameters, maximum_sharpness, minimum_strength)
  return lerp(img, process_packed(fid, img, packed), mask)


def vignet_mask(img, mask_parameters, maximum_sharpness=1, masking=True):
  """VignetFilter.get_mask, filters.py:360-396; mask_parameters (N,5) RAW:
  sigmoid(((gx A)^2 + (gy B)^2 + C - 5) * sharp * D / 5) * (E / 5 * .5 + .5); forced to 1 with masking off."""
  dt = _ft(img)
  filter_input_range = 5
  mp = tanh_range(-filter_input_range, filter_input_range, initial=0)(mask_parameters)
  grid = mask_grid(img.shape[1], img.shape[2], dt)
  inp = (grid[:, :, :, 0, None] * mp[:, None, None, 0, None])**2 + \
      (grid[:, :, :, 1, None] * mp[:, None, None, 1, None])**2 + \
      mp[:, None, None, 2, None] - filter_input_range
  inp = inp * (maximum_sharpness * mp[:, None, None, 3, None] / filter_input_range)
  mask = sigmoid(inp)
  mask = mask * (mp[:, None, None, 4, None] / filter_input_range * dt.type(0.5) + dt.type(0.5))
  if not masking:
    mask = mask * 0 + 1
  return mask


def vignet_apply(img, mask_parameters, maximum_sharpness=1, masking=True):
  """VignetFilter: process = img * 0 (filters.py:351-352), out = lerp(img, 0, mask) (filters.py:86-88)."""
  return lerp(img, img * 0, vignet_mask(img, mask_parameters, maximum_sharpness, masking))


# ---------------------------------------------------------------------------
# packed <-> reference-shaped parameters
# ---------------------------------------------------------------------------
def unpack_params(fid, packed):
  packed = np.asarray(packed)
  n = packed.shape[0]
  # (the curve filters' step count is cfg.curve_steps: L = P for Tone, P / 3 for Color; 8 in the shipped configs)
  if fid == FILTER_ID['T']:
    return packed.reshape(n, 1, 1, 1, packed.shape[1])
  if fid == FILTER_ID['C']:
    assert packed.shape[1] % 3 == 0
    return packed.reshape(n, 1, 1, 3, packed.shape[1] // 3)
  assert packed.shape == (n, NUM_PARAMS[fid]), (packed.shape, fid)
  return packed


def pack_params(fid, param):
  param = np.asarray(param)
  return param.reshape(param.shape[0], -1)


def regress_packed(fid, features, cfg=DEFAULT_CFG):
  """features (N,P) raw FC outputs -> packed params (N,P)."""
  return pack_params(fid, REGRESSORS[fid](np.asarray(features), cfg))


def process_packed(fid, img, packed):
  return PROCESS[fid](img, unpack_params(fid, np.asarray(packed, dtype=_ft(img))))


# ---------------------------------------------------------------------------
# backward (hand-derived); all take packed params and return (dx, dpacked)
# ---------------------------------------------------------------------------
def _sum_hwc(a):
  return a.reshape(a.shape[0], -1).sum(axis=1)


def exposure_backward(img, p, dy):
  s = np.exp(p[:, None, None, :] * np.log(2))
  y = img * s
  return dy * s, (np.log(2) * _sum_hwc(dy * y))[:, None]


def gamma_backward(img, p, dy):
  g = p[:, None, None, :]
  xm = np.maximum(img, 0.001)
  y = np.power(xm, g)
  dx = dy * g * np.power(xm, g - 1) * (img >= 0.001)
  return dx, _sum_hwc(dy * y * np.log(xm))[:, None]


def wb_backward(img, p, dy):
  return dy * p[:, None, None, :], (dy * img).sum(axis=(1, 2))


def _curve_backward(img, k, dy, L):
  """k: (N,1,1,Cc,L) with Cc in {1,3}."""
  S = k.sum(axis=4) + 1e-30  # (N,1,1,Cc)
  T = img * 0
  slope = img * 0
  clips = []
  for i in range(L):
    t = img - 1.0 * i / L
    c = np.clip(t, 0, 1.0 / L)
    clips.append(c)
    T = T + c * k[..., i]
    slope = slope + k[..., i] * ((t >= 0) & (t <= 1.0 / L))
  y = T * (L / S)
  dx = dy * (L / S) * slope
  dk = np.stack([dy * ((L / S) * clips[i] - y / S) for i in range(L)], axis=-1)  # (N,H,W,3,L)
  return dx, dk


def tone_backward(img, p, dy, L=None):
  L = p.shape[1] if L is None else L
  k = p.reshape(-1, 1, 1, 1, L)
  dx, dk = _curve_backward(img, k, dy, L)
  return dx, dk.sum(axis=(1, 2, 3))


def color_backward(img, p, dy, L=None):
  L = p.shape[1] // 3 if L is None else L
  k = p.reshape(-1, 1, 1, 3, L)
  dx, dk = _curve_backward(img, k, dy, L)
  return dx, dk.sum(axis=(1, 2)).reshape(-1, 3 * L)


def contrast_backward(img, p, dy):
  pp = p[:, :, None, None]
  w = np.array(LUM_W, dtype=img.dtype)
  lraw = rgb2lum(img)
  l = np.minimum(np.maximum(lraw, 0.0), 1.0)
  m = ((lraw >= 0.0) & (lraw <= 1.0)).astype(img.dtype)
  eps = 1e-6
  cl = -np.cos(math.pi * l) * 0.5 + 0.5
  dcl = 0.5 * math.pi * np.sin(math.pi * l)
  ratio = cl / (l + eps)
  ci = img * ratio
  G = dcl / (l + eps) - cl / (l + eps)**2
  dot = (dy * img).sum(axis=3, keepdims=True)
  dx = (1 - pp) * dy + pp * (dy * ratio + w * (m * G * dot))
  return dx, _sum_hwc(dy * (ci - img))[:, None]


def wnb_backward(img, p, dy):
  pp = p[:, :, None, None]
  w = np.array(LUM_W, dtype=img.dtype)
  lum = rgb2lum(img)
  dx = (1 - pp) * dy + pp * w * dy.sum(axis=3, keepdims=True)
  return dx, _sum_hwc(dy * (lum - img))[:, None]


def _satplus_full_grad(xc, dyfull):
  """d(full_color)/d(xc) applied to dyfull, analytic (hsv_grad_mode=1).

  Uses the hue-free identity full_c = v(1-s') + (s' v / rng)(xc_c - mn) that
  holds for rng > 0 (hue only encodes (xc_c - mn)/rng).  Returns 0 gradient
  through full_color where rng == 0 (hue is a constant 0 there).
  """
  v = xc.max(axis=3, keepdims=True)
  mn = xc.min(axis=3, keepdims=True)
  rng = v - mn
  n, h, w_, _ = xc.shape
  # one-hot of arg max / arg min following TF tie order is irrelevant a.e.;
  # use first-occurrence (ties have measure zero and are excluded in tests).
  amax = np.eye(3, dtype=xc.dtype)[xc.argmax(axis=3)]
  amin = np.eye(3, dtype=xc.dtype)[xc.argmin(axis=3)]
  ok = (rng > 0) & (v > 0)
  rs = np.where(ok, rng, 1.0)
  vs = np.where(ok, v, 1.0)
  s = rs / vs
  tri = 0.5 - np.abs(0.5 - vs)
  dtri_dv = np.sign(0.5 - vs)  # d/dv (0.5 - |0.5 - v|) = sign(0.5 - v)
  sp = s + (1 - s) * tri * 0.8
  # ds/dxc = (d rng)/v - rng/v^2 dv = (amax-amin)/v - rng/v^2 amax
  ds = (amax - amin) / vs - (rs / vs**2) * amax
  dsp = ds * (1 - tri * 0.8) + (1 - s) * 0.8 * dtri_dv * amax
  d = (xc - mn) / rs  # (N,H,W,3)
  # full_c = v*(1 - sp) + sp*v*d_c
  # dfull_c/dx_j = amax_j*(1-sp) - v*dsp_j + dsp_j*v*d_c + sp*amax_j*d_c + sp*v*dd_c/dx_j
  # dd_c/dx_j = (delta_cj - amin_j)/rng - (xc_c - mn)/rng^2 * (amax_j - amin_j)
  gsum = dyfull.sum(axis=3, keepdims=True)
  gd = (dyfull * d).sum(axis=3, keepdims=True)
  out = amax * (1 - sp) * gsum - vs * dsp * gsum + dsp * vs * gd + sp * amax * gd
  out = out + sp * vs * (dyfull / rs - amin * gsum / rs - (amax - amin) * gd / rs)
  return np.where(ok, out, 0.0)


def satplus_backward(img, p, dy, hsv_grad_mode=0):
  pp = p[:, :, None, None]
  xc, full = satplus_full_color(img)
  mask = (img <= 1.0)
  dxc = dy * (1 - pp)
  if hsv_grad_mode == 1:
    dxc = dxc + _satplus_full_grad(xc, dy * pp)
  dx = dxc * mask
  return dx, _sum_hwc(dy * (full - xc))[:, None]


def level_backward(img, p, dy):
  lower = p[:, 0][:, None, None, None]
  upper = (p[:, 1] + 1)[:, None, None, None]
  r = 1.0 / (upper - lower + 1e-6)
  t = (img - lower) * r
  inside = ((t >= 0.0) & (t <= 1.0)).astype(img.dtype)
  dx = dy * r * inside
  dlower = _sum_hwc(dy * inside * r * (t - 1.0))
  dupper = _sum_hwc(-dy * inside * t * r)
  return dx, np.stack([dlower, dupper], axis=1)


def backward_packed(fid, img, packed, dy, hsv_grad_mode=0):
  """Gradient of sum(y*dy) w.r.t. (img, packed params). float64 recommended."""
  img = np.asarray(img)
  p = np.asarray(packed, dtype=img.dtype)
  dy = np.asarray(dy, dtype=img.dtype)
  name = FILTER_NAMES[fid]
  if name == 'E':
    return exposure_backward(img, p, dy)
  if name == 'G':
    return gamma_backward(img, p, dy)
  if name == 'W':
    return wb_backward(img, p, dy)
  if name == 'S+':
    return satplus_backward(img, p, dy, hsv_grad_mode)
  if name == 'T':
    return tone_backward(img, p, dy)
  if name == 'Ct':
    return contrast_backward(img, p, dy)
  if name == 'BW':
    return wnb_backward(img, p, dy)
  if name == 'C':
    return color_backward(img, p, dy)
  if name == 'Le':
    return level_backward(img, p, dy)
  raise ValueError(fid)


# ---------------------------------------------------------------------------
# per-element TERMS of the parameter gradients: dparams = sum(terms), A = sum(|terms|)
# ---------------------------------------------------------------------------
def param_grad_terms(fid, img, packed, dy):
  """The per-output-element terms ``dy_e * d y_e / d p_k`` whose sum over (H, W, 3) is the parameter gradient
  ``backward_packed`` returns: shape (N, H, W, 3, P).  Written separately from the ``*_backward`` functions
  above (the tests require ``terms.sum == dparams``); small sizes only (Color: 24 values per element)."""
  img = np.asarray(img)
  p = np.asarray(packed, dtype=img.dtype)
  dy = np.asarray(dy, dtype=img.dtype)
  name = FILTER_NAMES[fid]
  L = p.shape[1] // (3 if name == 'C' else 1) if name in ('T', 'C') else CURVE_STEPS  # cfg.curve_steps
  if name == 'E':  # y = x 2^p: dy/dp = ln2 y
    return (np.log(2) * dy * exposure_process(img, p))[..., None]
  if name == 'G':  # y = xm^g: dy/dg = y ln xm
    return (dy * gamma_process(img, p) * np.log(np.maximum(img, 0.001)))[..., None]
  if name == 'W':  # y_c = x_c s_c: only channel c's elements see s_c
    return (dy * img)[..., None] * np.eye(3, dtype=img.dtype)
  if name == 'S+':  # y = xc (1-p) + full p
    xc, full = satplus_full_color(img)
    return (dy * (full - xc))[..., None]
  if name in ('T', 'C'):
    cc = 1 if name == 'T' else 3
    k = p.reshape(-1, 1, 1, cc, L)
    S = k.sum(axis=4) + 1e-30
    clips = np.stack([np.clip(img - 1.0 * i / L, 0, 1.0 / L) for i in range(L)], axis=-1)  # (N,H,W,3,L)
    y = (clips * k).sum(axis=-1) * (L / S)
    per_knot = dy[..., None] * ((L / S)[..., None] * clips - (y / S)[..., None])  # (N,H,W,3,L)
    if name == 'T':
      return per_knot
    out = np.zeros(img.shape + (3 * L,), dtype=img.dtype)
    for c in range(3):
      out[..., c, c * L:(c + 1) * L] = per_knot[..., c, :]
    return out
  if name == 'Ct':  # y = lerp(x, ci, p)
    l = np.minimum(np.maximum(rgb2lum(img), 0.0), 1.0)
    ci = img / (l + 1e-6) * (-np.cos(math.pi * l) * 0.5 + 0.5)
    return (dy * (ci - img))[..., None]
  if name == 'BW':  # y = lerp(x, lum, p)
    return (dy * (rgb2lum(img) - img))[..., None]
  if name == 'Le':
    lower = p[:, 0][:, None, None, None]
    upper = (p[:, 1] + 1)[:, None, None, None]
    r = 1.0 / (upper - lower + 1e-6)
    t = (img - lower) * r
    inside = ((t >= 0.0) & (t <= 1.0)).astype(img.dtype)
    return np.stack([dy * inside * r * (t - 1.0), -dy * inside * t * r], axis=-1)
  raise ValueError(fid)


def param_grad_abs(fid, img, packed, dy):
  """``A[n, k] = sum_e |dy_e * d y_e / d p_k|``: the sum of the ABSOLUTE terms of each parameter gradient -- the
  scale against which the rounding of an fp32 accumulation is judged (tests/_tol.py).  (N, P)."""
  t = param_grad_terms(fid, img, packed, dy)
  return np.abs(t).sum(axis=(1, 2, 3))


def curve_grad_abs_pieces(fid, img, packed, dy):
  """Tone / Color only: the scale of a parameter gradient evaluated as TWO sums,
  ``dk_i = (L/S) sum dy clip_i  -  (1/S) sum dy y``  (what any implementation that accumulates the clipped values and
  ``sum dy y`` separately computes -- the HIP kernels keep Q_i = sum dy min(x^, i/L) and combine them per image):
  ``A3[n, k] = sum_e |dy_e| ((L/S) clip_i + |y_e| / S) >= A``.  On a SATURATED image (every x >= 1) the two sums
  cancel exactly term by term, A is 0 and only A3 describes the rounding of such an evaluation; the tests use A3 there
  (constant images at the clamp edge, steps of a sequence that follow a strong exposure) and the strict A elsewhere."""
  img = np.asarray(img)
  p = np.asarray(packed, dtype=img.dtype)
  dy = np.abs(np.asarray(dy, dtype=img.dtype))
  name = FILTER_NAMES[fid]
  assert name in ('T', 'C'), name
  cc = 1 if name == 'T' else 3
  L = p.shape[1] // cc
  k = p.reshape(-1, 1, 1, cc, L)
  S = k.sum(axis=4) + 1e-30
  clips = np.stack([np.clip(img - 1.0 * i / L, 0, 1.0 / L) for i in range(L)], axis=-1)
  y = (clips * k).sum(axis=-1) * (L / S)
  pieces = dy[..., None] * (np.abs(L / S)[..., None] * clips + np.abs(y / S)[..., None])  # (N,H,W,3,L)
  if name == 'T':
    return pieces.sum(axis=(1, 2, 3))
  return pieces.sum(axis=(1, 2)).reshape(img.shape[0], 3 * L)


def masked_raw_grad_abs(fid, img, packed, mask_parameters, dy, maximum_sharpness=1, minimum_strength=0.3, h=1e-6):
  """Scale of the RAW mask-parameter gradients of the masked apply ``out = lerp(img, process(img), mask)``
  (filters.py:86-88): the terms are ``dy_e (process_e - img_e) d mask_e / d raw_k``, and both the reference (fp32 TF)
  and the kernels form ``process - img`` from an fp32 ``process(img)``, i.e. to an ulp of the COLOUR, not of the
  difference.  The scale is therefore the sum over the operands of that subtraction,
  ``A2[n, k] = sum_e |dy_e| (|process_e| + |img_e|) |d mask_e / d raw_k|  >=  A``  (d mask / d raw by central
  differences of ``get_mask``)."""
  img = np.asarray(img, dtype=np.float64)
  raw = np.asarray(mask_parameters, dtype=np.float64)
  w = np.abs(np.asarray(dy, dtype=np.float64)) * (np.abs(process_packed(fid, img, packed)) + np.abs(img))
  out = np.zeros_like(raw)
  for k in range(raw.shape[1]):
    e = np.zeros_like(raw)
    e[:, k] = h
    dm = (get_mask(img, raw + e, maximum_sharpness, minimum_strength) -
          get_mask(img, raw - e, maximum_sharpness, minimum_strength)) / (2 * h)
    out[:, k] = (w * np.abs(dm)).reshape(img.shape[0], -1).sum(axis=1)
  return out


def abs_terms_fd(fn, params, dy, h=1e-6):
  """The same ``A`` for ANY per-image parameter vector of any restated map ``fn(params) -> y`` (mask parameters,
  vignet parameters, raw pre-activation parameters ...), from central differences of the float64 restatement:
  ``A[n, k] = sum_e |dy_e (y_e(p + h e_k) - y_e(p - h e_k)) / 2h|``.  Images are independent, so column k of every
  image is perturbed at once.  ``A`` is a SCALE (needs percent accuracy, not digits)."""
  params = np.asarray(params, dtype=np.float64)
  dy = np.asarray(dy, dtype=np.float64)
  out = np.zeros(params.shape, dtype=np.float64)
  flat = params.reshape(params.shape[0], -1)
  for k in range(flat.shape[1]):
    e = np.zeros_like(flat)
    e[:, k] = h
    d = (np.asarray(fn((flat + e).reshape(params.shape)), dtype=np.float64) -
         np.asarray(fn((flat - e).reshape(params.shape)), dtype=np.float64)) / (2 * h)
    out.reshape(flat.shape)[:, k] = np.abs(dy * d).reshape(flat.shape[0], -1).sum(axis=1)
  return out


# ---------------------------------------------------------------------------
# Filter.apply with specified_parameter (filters.py:62-99, masking off)
# ---------------------------------------------------------------------------
def apply_specified(fid, img, packed, high_res=None):
  """lerp(img, process(img, p), mask) with mask == ones(1,1,1,1)
  (filters.py:111-113, 88) == 0*img + 1*process."""
  dt = _ft(img)
  mask = np.ones((1, 1, 1, 1), dtype=dt)
  low = lerp(img, process_packed(fid, img, packed), mask)
  high = None
  if high_res is not None:
    high = lerp(high_res, process_packed(fid, high_res, packed), mask)
  return low, high
