"""CPU: the parameter-gradient assertion of tests/_tol.py must be ABLE TO FAIL.

Round 3's bound (2e-4 * sum |dy|) let an all-zero `dparams` pass the 64x512x512 every-value test for five of the
eight filters.  The bound is now |err| <= 1e-4 |ref| + 2e-6 A with A the oracle's sum of the
absolute per-element terms.  Here FAKE gradients -- zeros, 0.99 x ref, -1 x ref, one wrong entry -- go through the very
assertion the gpu tests use, at the proxy size (64x64) and at the metric's image size (512x512), and every one of them
must be rejected, while the float64 value and an honest float32 accumulation of the same terms must be accepted."""
import numpy as np
import pytest

from exposure_amd import synthetic
from oracle import filters_c as fc
from oracle import filters_np as fnp
from oracle import nets_np as nn_np
from tests._tol import assert_param_grad_close


def rejected(got, ref, a):
  try:
    assert_param_grad_close(got, ref, a, 'mutant')
  except AssertionError:
    return True
  return False


@pytest.mark.parametrize('fid', range(8))
@pytest.mark.parametrize('shape', [(2, 64, 64, 3), (1, 512, 512, 3)])
@pytest.mark.parametrize('np_dt', [np.float16, np.float32])
def test_fake_parameter_gradients_are_rejected(fid, shape, np_dt):
  x, dy, params = synthetic.make_case(4242 + fid, shape, np_dt)
  x64, dy64, p64 = x.astype(np.float64), dy.astype(np.float64), params[fid].astype(np.float64)
  _, ref, a = fc.backward_packed(fid, x64, p64, dy64, with_abs=True)
  assert (np.abs(ref) > 0).all() and (a >= np.abs(ref)).all()
  assert not rejected(ref, ref, a)
  # an honest fp32 accumulation of the same terms (fp32 terms, fp32 pairwise sums) is inside the bound ...
  terms = fnp.param_grad_terms(fid, x64, p64, dy64).astype(np.float32)
  f32 = terms.reshape(shape[0], -1, terms.shape[-1]).sum(axis=1, dtype=np.float32)
  assert not rejected(f32, ref, a), np.abs(f32 - ref) / a
  # ... and a plain sequential fp32 accumulation in 256 per-thread partials + a tree, like the kernels'
  part = terms.reshape(shape[0], -1, 256, terms.shape[-1])
  acc = np.zeros(part.shape[0:1] + part.shape[2:], dtype=np.float32)
  for i in range(part.shape[1]) if part.shape[1] <= 64 else ():
    acc += part[:, i]
  if part.shape[1] <= 64:
    assert not rejected(acc.sum(axis=1, dtype=np.float32), ref, a)
  # the mutants
  assert rejected(np.zeros_like(ref), ref, a), 'an all-zero gradient passed'
  # a 1 % error is distinguishable from fp32 rounding only where the gradient is not itself a near-total cancellation
  # of its terms: 0.01 |ref| > 1e-4 |ref| + 2e-6 A  <=>  |ref| / A > 2.02e-4 (for dy ~ N(0, 1), |ref| / A ~ 1 / sqrt(H W 3):
  # 8e-3 at 64x64, 1e-3 at 512x512 -- the seeded cases sit above the limit except by chance)
  one = ref.copy()
  k = np.unravel_index((np.abs(ref) / a).argmax(), ref.shape)
  one[k] *= 1.01
  if (np.abs(ref) / a).max() > 2.1e-4:
    assert rejected(0.99 * ref, ref, a), '0.99 x ref passed'
    assert rejected(one, ref, a), 'one entry off by 1 % passed'
  else:
    assert (np.abs(ref) / a).max() > 2e-5 and shape[1] == 512, 'unexpectedly complete cancellation'
  assert rejected(-ref, ref, a), 'a sign-flipped gradient passed'
  assert rejected(0.5 * ref, ref, a)


def test_the_three_sums_of_absolute_terms_agree():
  """A from the per-element terms (NumPy), from the C restatement's accumulators and from central differences of the
  forward -- three routes to the same scale; and the terms sum to the hand-derived parameter gradient."""
  x, dy, params = synthetic.make_case(7, (2, 24, 40, 3), np.float32)
  x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
  for fid in range(9):
    p64 = (params[fid] if fid < 8 else synthetic.make_params(np.random.default_rng(3), 8, 2)).astype(np.float64)
    terms = fnp.param_grad_terms(fid, x64, p64, dy64)
    _, dp = fnp.backward_packed(fid, x64, p64, dy64)
    a = fnp.param_grad_abs(fid, x64, p64, dy64)
    assert np.abs(terms.sum(axis=(1, 2, 3)) - dp).max() <= 1e-12 * a.max()
    a_fd = fnp.abs_terms_fd(lambda q: fnp.process_packed(fid, x64, q), p64, dy64)
    np.testing.assert_allclose(a_fd, a, rtol=1e-5)
    if fid < 8:
      _, dp_c, a_c = fc.backward_packed(fid, x64, p64, dy64, with_abs=True)
      np.testing.assert_allclose(a_c, a, rtol=1e-12)
      assert np.abs(dp_c - dp).max() <= 1e-12 * a.max()


def test_fake_jvp_is_rejected():
  rng = np.random.default_rng(5)
  img = rng.random((3, 64, 64, 3)) * 1.5 - 0.2
  v = rng.normal(size=img.shape)
  _, cache = nn_np.stat_features(img)
  jv = nn_np.stat_features_jvp(cache, v)
  a = nn_np.stat_features_jvp_abs(cache, v)
  # the rows of J from the backward with unit upstream gradients reproduce J v
  e = np.eye(3)
  alt = np.stack([(nn_np.stat_features_backward(cache, np.tile(e[k], (3, 1))) * v).reshape(3, -1).sum(axis=1)
                  for k in range(3)], axis=1)
  assert np.abs(alt - jv).max() <= 1e-13 * a.max()
  assert not rejected(jv, jv, a)
  assert rejected(np.zeros_like(jv), jv, a) and rejected(0.99 * jv, jv, a) and rejected(-jv, jv, a)


def test_masked_apply_mutants_are_rejected():
  """The mask-parameter gradients (six per image): scale = sum over the operands of `process - img` (A2 >= A)."""
  import torch
  from oracle import filters_torch as ft
  rng = np.random.default_rng(2)
  shape = (2, 64, 64, 3)
  x, dy, params = synthetic.make_case(66, shape, np.float32)
  raw = rng.standard_normal((2, 6))
  fid = 5
  x64, dy64, p64 = x.astype(np.float64), dy.astype(np.float64), params[fid].astype(np.float64)
  _, _, rdp, rdraw = ft.apply_masked_backward(fid, torch.from_numpy(x64), torch.from_numpy(p64), torch.from_numpy(raw),
                                              torch.from_numpy(dy64), 1.0, 0.3)
  a_p = fnp.abs_terms_fd(lambda q: fnp.apply_masked(fid, x64, q, raw, 1.0, 0.3), p64, dy64)
  a_raw = fnp.masked_raw_grad_abs(fid, x64, p64, raw, dy64, 1.0, 0.3)
  assert (a_raw >= fnp.abs_terms_fd(lambda r: fnp.apply_masked(fid, x64, p64, r, 1.0, 0.3), raw, dy64)).all()
  for ref, a in ((rdp.numpy(), a_p), (rdraw.numpy(), a_raw)):
    assert not rejected(ref, ref, a)
    assert rejected(np.zeros_like(ref), ref, a) and rejected(0.99 * ref, ref, a) and rejected(-ref, ref, a)
