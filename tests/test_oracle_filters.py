"""Pins the CPU oracle (no reference golden vectors exist -- SURVEY.md section 8c).

(1) identities of the reference code, (2) hand-computed points, (3) float64 finite
differences, (4) agreement of the two independent restatements, (5) matplotlib as an
independent HSV implementation.
"""
import math

import numpy as np
import pytest
import torch

from exposure_amd import synthetic
from oracle import filters_np as fnp
from oracle import filters_torch as ft

E, G, W, SP, T, CT, BW, C = range(8)


def rand_img(seed=0, shape=(2, 8, 8, 3)):
  return synthetic.make_images(np.random.default_rng(seed), shape).astype(np.float64)


# ---------------------------------------------------------------- identities
def test_identity_exposure_zero():
  x = rand_img()
  assert np.array_equal(fnp.exposure_process(x, np.zeros((2, 1))), x)


def test_identity_gamma_one():
  x = rand_img()
  np.testing.assert_allclose(fnp.gamma_process(x, np.ones((2, 1))), np.maximum(x, 1e-3), rtol=1e-15)


def test_identity_wb_zero_features():
  s = fnp.wb_regressor(np.zeros((2, 3)))
  np.testing.assert_allclose(s, np.ones((2, 3)) / (1 + 1e-5), rtol=1e-14)


def test_wb_first_feature_masked():
  a = fnp.wb_regressor(np.array([[5.0, 0.3, -0.2]]))
  b = fnp.wb_regressor(np.array([[-7.0, 0.3, -0.2]]))
  assert np.array_equal(a, b)


@pytest.mark.parametrize('fid', [T, C])
def test_identity_curves_equal_knots(fid):
  x = rand_img()
  p = np.full((2, fnp.NUM_PARAMS[fid]), 1.3)
  y = fnp.process_packed(fid, x, p)
  np.testing.assert_allclose(y, np.clip(x, 0, 1), atol=1e-15)
  y32 = fnp.process_packed(fid, x.astype(np.float32), p.astype(np.float32))
  assert np.abs(y32 - np.clip(x, 0, 1)).max() < 2e-7


def test_identity_contrast_wnb_zero():
  x = rand_img()
  z = np.zeros((2, 1))
  np.testing.assert_allclose(fnp.contrast_process(x, z), x, atol=0)
  np.testing.assert_allclose(fnp.wnb_process(x, z), x, atol=0)
  np.testing.assert_allclose(fnp.wnb_process(x, z + 1), np.broadcast_to(fnp.rgb2lum(x), x.shape),
                             atol=1e-16)


def test_identity_satplus_zero_and_grey():
  x = rand_img()
  np.testing.assert_allclose(fnp.satplus_process(x, np.zeros((2, 1))), np.minimum(x, 1.0), atol=0)
  grey = np.full((1, 1, 1, 3), 0.5)
  _, full = fnp.satplus_full_color(grey)
  # s=0, v=.5 -> s' = 0.4, hue 0 -> (v, v(1-s'), v(1-s'))
  np.testing.assert_allclose(full[0, 0, 0], [0.5, 0.3, 0.3], atol=1e-15)


# ------------------------------------------------------ hand-computed points
def test_known_points():
  one = np.ones((1, 1))
  x = np.full((1, 1, 1, 3), 0.25)
  np.testing.assert_allclose(fnp.exposure_process(x, one), 0.5, rtol=1e-15)
  np.testing.assert_allclose(fnp.gamma_process(x, one * 0.5), 0.5, rtol=1e-15)
  np.testing.assert_allclose(fnp.gamma_process(x * 0, one * 2.0), 1e-6, rtol=1e-12)
  l = 0.25
  expect = (0.5 - 0.5 * math.cos(math.pi / 4)) * 0.25 / (0.25 + 1e-6)
  np.testing.assert_allclose(fnp.contrast_process(x, one), expect, rtol=1e-14)
  k = np.array([[2.0, 1, 1, 1, 1, 1, 1, 1]])
  np.testing.assert_allclose(fnp.process_packed(T, np.full((1, 1, 1, 3), 0.125), k), 8 / 9 * 0.25,
                             rtol=1e-14)
  # tone saturates: x >= 1 -> 1, x <= 0 -> 0
  np.testing.assert_allclose(fnp.process_packed(T, np.full((1, 1, 1, 3), 1.7), k), 1.0, rtol=1e-14)
  assert fnp.process_packed(T, np.full((1, 1, 1, 3), -0.3), k).max() == 0.0
  # exposure regressor range, gamma regressor range
  assert abs(fnp.exposure_regressor(np.array([[50.0]]))[0, 0] - 3.5) < 1e-12
  assert abs(fnp.gamma_regressor(np.array([[-50.0]]))[0, 0] - 1 / 3) < 1e-12
  assert abs(fnp.tone_regressor(np.zeros((1, 8)))[0, 0, 0, 0, 0] - 1.25) < 1e-15
  assert abs(fnp.color_regressor(np.zeros((1, 24)))[0, 0, 0, 2, 7] - 1.0) < 1e-15
  assert fnp.color_regressor(np.zeros((3, 24))).shape == (3, 1, 1, 3, 8)
  assert fnp.tone_regressor(np.zeros((3, 8))).shape == (3, 1, 1, 1, 8)


def test_color_packing_is_channel_major():
  x = np.full((1, 1, 1, 3), 0.05)
  p = np.ones((1, 24))
  p[0, 8 + 0] = 1.1  # channel 1 (G), knot 0
  y = fnp.process_packed(C, x, p)
  assert y[0, 0, 0, 0] == pytest.approx(0.05) and y[0, 0, 0, 2] == pytest.approx(0.05)
  assert y[0, 0, 0, 1] == pytest.approx(0.05 * 1.1 * 8 / 8.1)


# ----------------------------------------------------------- HSV vs colorsys / matplotlib
def test_hsv_matches_colorsys():
  """TensorFlow's own unit test of tf.image.rgb_to_hsv / hsv_to_rgb (image_ops_test.py,
  RGBToHSVTest.testBatch, TF 1.x) checks the ops against Python's ``colorsys`` tuple by tuple; the
  oracle's restatement of the two ops (which are not in the reference tree) is held to the same
  check, grey and black pixels (rng = 0, v = 0) included."""
  import colorsys
  rng = np.random.default_rng(12)
  rgb = rng.random((2, 8, 8, 3))
  rgb[0, 0, :3] = [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5], [1.0, 1.0, 1.0]]  # black / grey / white
  rgb[0, 1, :3] = [[0.7, 0.7, 0.2], [0.2, 0.7, 0.7], [0.7, 0.2, 0.7]]  # tied maxima
  hsv = fnp.rgb_to_hsv(rgb)
  back = fnp.hsv_to_rgb(hsv)
  for px, got, rt in zip(rgb.reshape(-1, 3), hsv.reshape(-1, 3), back.reshape(-1, 3)):
    np.testing.assert_allclose(got, colorsys.rgb_to_hsv(*px), atol=1e-12)
    np.testing.assert_allclose(rt, colorsys.hsv_to_rgb(*got), atol=1e-12)


# ----------------------------------------------------------- HSV vs matplotlib (a second independent implementation)
def test_hsv_matches_matplotlib():
  mc = pytest.importorskip('matplotlib.colors')
  rgb = np.random.default_rng(3).random((4, 16, 16, 3))
  np.testing.assert_allclose(fnp.rgb_to_hsv(rgb), mc.rgb_to_hsv(rgb), atol=1e-14)
  hsv = np.random.default_rng(4).random((4, 16, 16, 3))
  np.testing.assert_allclose(fnp.hsv_to_rgb(hsv), mc.hsv_to_rgb(hsv), atol=1e-14)
  np.testing.assert_allclose(fnp.hsv_to_rgb(fnp.rgb_to_hsv(rgb)), rgb, atol=1e-14)


# ------------------------------------------- np (hand bwd) vs torch (autograd)
@pytest.mark.parametrize('fid', range(8))
@pytest.mark.parametrize('mode', [0, 1])
def test_np_vs_torch_forward_backward(fid, mode):
  if mode == 1 and fid != SP:
    pytest.skip('hsv_grad_mode only affects S+')
  x, dy, params = synthetic.make_case(11 + fid, (3, 16, 16, 3))
  x = x.astype(np.float64)
  dy = dy.astype(np.float64)
  p = params[fid].astype(np.float64)
  y_np = fnp.process_packed(fid, x, p)
  dx_np, dp_np = fnp.backward_packed(fid, x, p, dy, hsv_grad_mode=mode)
  tx, tp, tdy = map(torch.from_numpy, (x, p, dy))
  y_t = ft.process_packed(fid, tx, tp, mode).numpy()
  dx_t, dp_t = ft.backward_packed(fid, tx, tp, tdy, mode)
  np.testing.assert_allclose(y_np, y_t, rtol=1e-12, atol=1e-13)
  np.testing.assert_allclose(dx_np, dx_t.numpy(), rtol=1e-9, atol=1e-9)
  np.testing.assert_allclose(dp_np, dp_t.numpy(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('fid', range(8))
def test_np_float32_close_to_float64(fid):
  x, dy, params = synthetic.make_case(21 + fid, (2, 16, 16, 3))
  y64 = fnp.process_packed(fid, x.astype(np.float64), params[fid].astype(np.float64))
  y32 = fnp.process_packed(fid, x.astype(np.float32), params[fid])
  assert y32.dtype == np.float32
  assert np.abs(y32 - y64).max() <= 2e-6 * max(1.0, np.abs(y64).max())


# ----------------------------------------------------------- tie conventions
def test_gradient_ties_follow_tf():
  one = np.ones((1, 1))
  dy = np.ones((1, 1, 1, 3))
  # gamma: maximum(x, 0.001) passes gradient at equality, blocks below
  x = np.array([0.001, 0.0005, 0.5]).reshape(1, 1, 1, 3)
  dx, _ = fnp.backward_packed(G, x, one * 2.0, dy)
  assert dx[0, 0, 0, 0] == pytest.approx(2 * 0.001) and dx[0, 0, 0, 1] == 0
  dxt, _ = ft.backward_packed(G, torch.from_numpy(x), torch.from_numpy(one * 2.0), torch.from_numpy(dy))
  np.testing.assert_allclose(dx, dxt.numpy(), rtol=1e-12)
  # tone: x exactly on a knot boundary -> both neighbouring segments contribute
  k = np.arange(1, 9, dtype=np.float64).reshape(1, 8)
  x = np.array([0.25, 0.0, 1.0]).reshape(1, 1, 1, 3)
  dx, _ = fnp.backward_packed(T, x, k, dy)
  S = k.sum()
  np.testing.assert_allclose(dx[0, 0, 0], [8 / S * (k[0, 1] + k[0, 2]), 8 / S * k[0, 0], 8 / S * k[0, 7]],
                             rtol=1e-13)
  dxt, _ = ft.backward_packed(T, torch.from_numpy(x), torch.from_numpy(k), torch.from_numpy(dy))
  np.testing.assert_allclose(dx, dxt.numpy(), rtol=1e-12)
  # satplus: minimum(x, 1.0) passes at equality, blocks above
  x = np.array([1.0, 1.5, 0.2]).reshape(1, 1, 1, 3)
  dx, _ = fnp.backward_packed(SP, x, one * 0.25, dy)
  np.testing.assert_allclose(dx[0, 0, 0], [0.75, 0.0, 0.75], rtol=1e-13)
  # contrast: lum exactly 0 and exactly 1 still pass (inclusive both sides)
  for val in (0.0, 1.0):
    x = np.full((1, 1, 1, 3), val)
    dx, _ = fnp.backward_packed(CT, x, one * 0.5, dy)
    dxt, _ = ft.backward_packed(CT, torch.from_numpy(x), torch.from_numpy(one * 0.5), torch.from_numpy(dy))
    np.testing.assert_allclose(dx, dxt.numpy(), rtol=1e-9, atol=1e-12)


# -------------------------------------------------------- finite differences
@pytest.mark.parametrize('fid', range(8))
def test_backward_matches_finite_differences(fid):
  rng = np.random.default_rng(100 + fid)
  # away from kinks: avoid knots (k/8), the 1e-3 / 1.0 clamps and channel ties
  x = rng.uniform(0.02, 0.98, (2, 4, 4, 3))
  x = np.where(np.abs(x * 8 - np.round(x * 8)) < 0.02, x + 0.03 / 8, x)
  dy = rng.standard_normal(x.shape)
  p = synthetic.make_params(rng, fid, 2).astype(np.float64)
  mode = 1 if fid == SP else 0
  dx, dp = fnp.backward_packed(fid, x, p, dy, hsv_grad_mode=mode)

  def loss(xx, pp):
    return float((fnp.process_packed(fid, xx, pp) * dy).sum())

  h = 1e-6
  for idx in [(0, 0, 0, 0), (1, 3, 2, 1), (0, 2, 3, 2), (1, 1, 1, 0)]:
    xp, xm = x.copy(), x.copy()
    xp[idx] += h
    xm[idx] -= h
    fd = (loss(xp, p) - loss(xm, p)) / (2 * h)
    assert fd == pytest.approx(dx[idx], rel=1e-5, abs=1e-6), (fid, idx)
  for j in range(p.shape[1]):
    pp, pm = p.copy(), p.copy()
    pp[0, j] += h
    pm[0, j] -= h
    fd = (loss(x, pp) - loss(x, pm)) / (2 * h)
    assert fd == pytest.approx(dp[0, j], rel=1e-5, abs=1e-6), (fid, j)


def test_satplus_tf_mode_blocks_hsv_gradient():
  x, dy, params = synthetic.make_case(5, (1, 8, 8, 3))
  x, dy = x.astype(np.float64), dy.astype(np.float64)
  p = params[SP].astype(np.float64)
  dx0, dp0 = fnp.backward_packed(SP, x, p, dy, hsv_grad_mode=0)
  np.testing.assert_allclose(dx0, dy * (1 - p[:, :, None, None]) * (x <= 1.0), rtol=1e-14)
  dx1, dp1 = fnp.backward_packed(SP, x, p, dy, hsv_grad_mode=1)
  np.testing.assert_allclose(dp0, dp1)
  assert np.abs(dx0 - dx1).max() > 1e-3


def test_apply_specified_is_process():
  x, _, params = synthetic.make_case(6, (2, 8, 8, 3))
  x = x.astype(np.float64)
  hi = rand_img(9, (2, 16, 16, 3))
  low, high = fnp.apply_specified(E, x, params[E], high_res=hi)
  np.testing.assert_allclose(low, fnp.process_packed(E, x, params[E]), rtol=1e-15)
  np.testing.assert_allclose(high, fnp.process_packed(E, hi, params[E]), rtol=1e-15)


def test_chain_torch_matches_stepwise_numpy():
  x, dy, params = synthetic.make_case(7, (2, 8, 8, 3))
  x64 = x.astype(np.float64)
  cur = x64
  acts = [cur]
  for fid in range(8):
    cur = fnp.process_packed(fid, cur, params[fid].astype(np.float64))
    acts.append(cur)
  g = dy.astype(np.float64)
  dps = [None] * 8
  for fid in reversed(range(8)):
    g, dps[fid] = fnp.backward_packed(fid, acts[fid], params[fid].astype(np.float64), g)
  y_t, dx_t, dps_t = ft.chain_fwd_bwd(torch.from_numpy(x64),
                                      [torch.from_numpy(p.astype(np.float64)) for p in params],
                                      torch.from_numpy(dy.astype(np.float64)))
  np.testing.assert_allclose(acts[-1], y_t.numpy(), rtol=1e-10, atol=1e-12)
  np.testing.assert_allclose(g, dx_t.numpy(), rtol=1e-8, atol=1e-9)
  for a, b in zip(dps, dps_t):
    np.testing.assert_allclose(a, b.numpy(), rtol=1e-8, atol=1e-8)
