"""Vectors produced by RUNNING THE REFERENCE'S OWN NumPy CODE in the build container (tests/golden/
make_reference_vectors.py: definitions cut out of user_study_ui/filters.py, util.py and histogram_intersection.py and
executed with numpy + math only; the fixture holds data and the source files' sha256).  The one piece of evidence in
this repository that does not come from the build's own restatements: CPU -- the three filter oracles, the host helpers
and the metric reproduce it; GPU -- the HIP kernels do.  Covered: Exposure, Gamma (x >= 0.001), WhiteBalance (regressor
normalisation + process), rgb2lum, lerp, the ProPhoto linearisation, the metric's luminance statistics and histogram
arithmetic, the coordinate grid of the spatial masks.  NOT covered (no executable statement in /root/reference): everything that lives in TensorFlow."""
import os

import numpy as np
import pytest
import torch

from oracle import filters_np as fnp

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ref():
  return np.load(os.path.join(HERE, 'reference_numpy.npz'))


def wb_features(temp_tint):
  """Features f with filters.py:227-232's colour scaling == the UI's (1, exp(-tint), exp(-temperature)):
  exp(tanh_range(-0.5, 0.5)(f)) = exp(0.5 tanh f)  ->  f = atanh(-2 v).  Channel 0 is masked to 0 by the regressor."""
  f = np.zeros((temp_tint.shape[0], 3))
  f[:, 0] = 7.0  # must not matter (features * (0, 1, 1))
  f[:, 1] = np.arctanh(-2.0 * temp_tint[:, 1])
  f[:, 2] = np.arctanh(-2.0 * temp_tint[:, 0])
  return f


def cases(ref):
  """-> [(filter id, x, packed params (float64), reference output)]"""
  img = ref['ui_images'].astype(np.float64)
  wb = fnp.regress_packed(2, wb_features(ref['ui_wb_temp_tint']))
  return [(0, img, ref['ui_exposure_ev'][:, None], ref['ui_exposure_y']),
          (1, ref['ui_gamma_x'].astype(np.float64), ref['ui_gamma_g'][:, None], ref['ui_gamma_y']),
          (2, img, wb, ref['ui_wb_y'])]


def test_fixture_names_its_sources(ref):
  prov = [str(p) for p in ref['provenance']]
  assert len(prov) == 4 and all('sha256=' in p and len(p.split('sha256=')[1]) == 64 for p in prov)
  assert prov[0].startswith('user_study_ui/filters.py') and prov[1].startswith('util.py')
  # the generated parameters lie inside the ranges the TF path's regressors can produce
  assert np.abs(ref['ui_exposure_ev']).max() <= 3.5 and (ref['ui_gamma_g'] >= 1 / 3).all() and (ref['ui_gamma_g'] <= 3).all()
  assert np.abs(ref['ui_wb_temp_tint']).max() < 0.5 and (ref['ui_gamma_x'] >= np.float32(0.001)).all()
  assert (ref['ui_images'] < 0).any() and (ref['ui_images'] > 1).any()


def test_numpy_oracle_reproduces_the_reference_run(ref):
  # the UI computes in float32 (float32 images, float32 colour scaling): 2e-6 relative is its rounding, not slack
  for fid, x, p, want in cases(ref):
    got = fnp.process_packed(fid, x, p)
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-7, err_msg=fnp.FILTER_NAMES[fid])
  # the white-balance regressor alone: normalised scaling == the UI's (float32) colour_scaling
  tt = ref['ui_wb_temp_tint']
  s = np.stack([np.ones(len(tt)), np.exp(-tt[:, 1]), np.exp(-tt[:, 0])], axis=1)
  s = s / (1e-5 + 0.27 * s[:, :1] + 0.67 * s[:, 1:2] + 0.06 * s[:, 2:])
  np.testing.assert_allclose(fnp.regress_packed(2, wb_features(tt)), s, rtol=1e-12)
  np.testing.assert_allclose(fnp.rgb2lum(ref['ui_images'].astype(np.float64)), ref['ui_rgb2lum'], rtol=2e-6, atol=1e-7)
  a, b, l = (ref[k].astype(np.float64) for k in ('ui_images', 'ui_lerp_b', 'ui_lerp_alpha'))
  np.testing.assert_allclose(fnp.lerp(a, b, l), ref['ui_lerp_y'], rtol=1e-5, atol=2e-7)
  np.testing.assert_allclose(fnp.lerp(a, b, l), ref['util_lerp_y'], rtol=1e-14, atol=1e-16)


def test_c_and_torch_oracles_reproduce_the_reference_run(ref):
  from oracle import filters_c, filters_torch
  for fid, x, p, want in cases(ref):
    y = filters_c.process_packed(fid, x, p, dtype=np.float64)
    np.testing.assert_allclose(y, want, rtol=3e-6, atol=1e-7, err_msg='C %d' % fid)
    yt = filters_torch.process_packed(fid, torch.from_numpy(x), torch.from_numpy(p)).numpy()
    np.testing.assert_allclose(yt, want, rtol=3e-6, atol=1e-7, err_msg='torch %d' % fid)


def test_host_helpers_reproduce_the_reference_run(ref):
  from exposure_amd import evaluate, util
  img = torch.from_numpy(ref['ui_images'])
  np.testing.assert_allclose(util.rgb2lum(img).numpy(), ref['ui_rgb2lum'], rtol=2e-6, atol=1e-7)
  got = util.lerp(img, torch.from_numpy(ref['ui_lerp_b']), torch.from_numpy(ref['ui_lerp_alpha'])).numpy()
  np.testing.assert_allclose(got, ref['ui_lerp_y'], rtol=1e-5, atol=2e-7)
  pp = torch.from_numpy(ref['util_pp_rgb'])
  np.testing.assert_allclose(evaluate.linearize_ProPhotoRGB(pp).numpy(), ref['util_linearized'], rtol=1e-12)
  # (the reverse direction is the same statement with 1 / 1.8: the two must invert each other)
  np.testing.assert_allclose(ref['util_delinearized']**1.8, ref['util_pp_rgb'], rtol=1e-12)


def test_metric_reproduces_the_reference_run(ref):
  from exposure_amd import metrics
  for tag in 'ab':
    st = metrics.get_statistics(torch.from_numpy(ref['hi_images_%s' % tag])).double().numpy()
    np.testing.assert_allclose(st[:, :2], ref['hi_stats_%s' % tag], rtol=2e-5, atol=2e-6)
    for k in range(2):
      h = metrics.calc_hist(torch.from_numpy(ref['hi_stats_%s' % tag][:, k])).double().numpy()
      np.testing.assert_allclose(h, ref['hi_hists_%s' % tag][k], atol=1e-7)
  for k in range(2):
    got = metrics.hist_intersection(torch.from_numpy(ref['hi_hists_a'][k]), torch.from_numpy(ref['hi_hists_b'][k]))
    assert abs(float(got) - float(ref['hi_intersections'][k])) <= 1e-12
  # np.histogram's rules through the reference's own calc_hist: right edge inclusive, outside values dropped but counted
  # in the denominator
  h = metrics.calc_hist(torch.from_numpy(ref['hi_edge_values'])).double().numpy()
  np.testing.assert_allclose(h, ref['hi_edge_hist'], atol=1e-7)
  assert abs(ref['hi_edge_hist'].sum() - 0.8) < 1e-12


def test_mask_grid_equals_the_reference_statements(ref):
  """The constant coordinate grid of the spatial masks: the NumPy statements inside ``Filter.get_mask`` and
  ``VignetFilter.get_mask`` (filters.py:124-133, 371-380), executed in the build container for square, portrait,
  landscape and minimal sizes, against the oracle's closed form -- bit for bit in float32 (what the reference feeds to
  ``tf.constant``); the two reference methods build the same grid."""
  from oracle import filters_torch
  sizes = [tuple(int(v) for v in s) for s in ref['mask_grid_sizes']]
  assert (5, 9) in sizes and (9, 7) in sizes  # non-square both ways: the shorter-edge centring
  for h, w in sizes:
    want = ref['mask_grid_%dx%d' % (h, w)]
    assert want.dtype == np.float32 and want.shape == (1, h, w, 2)
    assert np.array_equal(ref['vignet_grid_%dx%d' % (h, w)], want)
    assert np.array_equal(fnp.mask_grid(h, w, np.float32), want), (h, w)
    # the float64 oracle's grid IS the float32 one widened
    assert np.array_equal(fnp.mask_grid(h, w, np.float64), want.astype(np.float64))
  # through the masks themselves: a mask that depends on the row only / the column only reproduces the grid's values
  img = np.zeros((1, 5, 9, 3))
  mp = np.zeros((1, 6))
  mp[0, 0], mp[0, 4], mp[0, 5] = 0.3, 0.7, 0.2  # A (row coefficient), sharpness, strength; B = C = D = 0
  m = fnp.get_mask(img, mp)
  t = lambda v: np.tanh(v) * 5.0  # tanh_range(-5, 5, initial=0)
  g = ref['mask_grid_5x9'].astype(np.float64)
  inp = (g[..., 0:1] * t(0.3) + t(0.0) * (0.0 - 0.5)) * (1 * t(0.7) / 5)
  want = 1 / (1 + np.exp(-inp)) * (t(0.2) / 5 * 0.5 + 0.5) * (1 - 0.3) + 0.3
  np.testing.assert_allclose(m, want, rtol=1e-12)
  tm = filters_torch.get_mask(torch.from_numpy(img), torch.from_numpy(mp)).numpy()
  np.testing.assert_allclose(tm, want, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_hip_kernels_reproduce_the_reference_run(ref, dtype, gpu_device):
  """expo_filter_fwd for Exposure, Gamma and WhiteBalance against the outputs of the reference's own NumPy filters.
  fp32 storage: 1e-5 relative (exp2 / log2 based pow on the device vs libm); fp16 storage: the input is rounded to
  fp16 FIRST and the reference output is compared at the oracle's value for that rounded input plus half an fp16 ulp
  -- the reference run itself enters through the float64 oracle, which test_numpy_oracle_... ties to it at 3e-6."""
  from exposure_amd import _cabi
  from tests._tol import assert_image_close
  dev = gpu_device
  for fid, x, p, want in cases(ref):
    tx = torch.from_numpy(x).to(dev).to(dtype)
    tp = torch.from_numpy(p.astype(np.float32)).to(dev)
    y = torch.empty_like(tx)
    _cabi.filter_fwd(fid, tx, y, tp)
    got = y.float().cpu().numpy()
    if dtype == torch.float32:
      np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=fnp.FILTER_NAMES[fid])
    else:
      xr = tx.double().cpu().numpy()
      assert_image_close(got, fnp.process_packed(fid, xr, p), np.float16, 'reference run, fp16 storage, filter %d' % fid)
      # and directly: within fp16 resolution of the reference's own output (input + output rounding, |dy/dx| <= 2^3.5 * 3)
      assert (np.abs(got - want) <= 1e-3 + 0.02 * np.abs(want)).all()


@pytest.mark.gpu
def test_metric_on_the_device_reproduces_the_reference_run(ref, gpu_device):
  from exposure_amd import metrics
  for tag in 'ab':
    st = metrics.get_statistics(torch.from_numpy(ref['hi_images_%s' % tag]).to(gpu_device)).double().cpu().numpy()
    np.testing.assert_allclose(st[:, :2], ref['hi_stats_%s' % tag], rtol=2e-5, atol=2e-6)
  a = torch.from_numpy(ref['hi_images_a']).to(gpu_device)
  b = torch.from_numpy(ref['hi_images_b']).to(gpu_device)
  ints, _ = metrics.histogram_intersection(a, b)
  for k in range(2):
    assert abs(ints[k] - float(ref['hi_intersections'][k])) <= 1e-6
