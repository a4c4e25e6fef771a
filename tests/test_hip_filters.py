"""GPU parity: every HIP filter kernel (through the C-ABI) vs the float64 oracle on the same
fp16/fp32-quantised inputs; edge cases; size-independent properties at full size."""
import os

import numpy as np
import pytest
import torch

from exposure_amd import _cabi, filters, synthetic
from exposure_amd.config import make_cfg
from oracle import filters_np as fnp
from tests._tol import assert_image_close, assert_param_grad_close

pytestmark = pytest.mark.gpu

NP_DT = {torch.float16: np.float16, torch.float32: np.float32}


def run_fwd_bwd(fid, x, dy, p, dtype, dev, mode=0, need_dx=True):
  tx = torch.from_numpy(x).to(dev).to(dtype)
  tdy = torch.from_numpy(dy).to(dev).to(dtype)
  tp = torch.from_numpy(p).to(dev)
  y = torch.empty_like(tx)
  _cabi.filter_fwd(fid, tx, y, tp)
  dx = torch.empty_like(tx) if need_dx else None
  dp = torch.full_like(tp, 7.0)  # must be overwritten
  _cabi.filter_bwd(fid, tx, tdy, dx, tp, dp, mode)
  torch.cuda.synchronize()
  return (y.float().cpu().numpy(), dx.float().cpu().numpy() if need_dx else None, dp.cpu().numpy())


def oracle(fid, x, dy, p, mode=0):
  x64, dy64, p64 = x.astype(np.float64), dy.astype(np.float64), p.astype(np.float64)
  y = fnp.process_packed(fid, x64, p64)
  dx, dp = fnp.backward_packed(fid, x64, p64, dy64, hsv_grad_mode=mode)
  return y, dx, dp


def grad_abs(fid, x, dy, p):
  """A = sum_e |dy_e dy_e/dp_k| per parameter (the oracle's sum of absolute terms): tests/_tol.py judges the fp32
  accumulation of the parameter gradients against 1e-4 |ref| + 2e-6 A."""
  return fnp.param_grad_abs(fid, x.astype(np.float64), p.astype(np.float64), dy.astype(np.float64))


@pytest.mark.parametrize('fid', range(8))
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(4, 64, 64, 3), (3, 16, 24, 3)])
def test_filter_matches_oracle(fid, dtype, shape, gpu_device):
  x, dy, params = synthetic.make_case(1000 + fid, shape, NP_DT[dtype])
  p = params[fid]
  y, dx, dp = run_fwd_bwd(fid, x, dy, p, dtype, gpu_device)
  ry, rdx, rdp = oracle(fid, x, dy, p)
  assert_image_close(y, ry, NP_DT[dtype], 'y fid %d' % fid)
  assert_image_close(dx, rdx, NP_DT[dtype], 'dx fid %d' % fid)
  assert_param_grad_close(dp, rdp, grad_abs(fid, x, dy, p), 'dparams fid %d' % fid)


@pytest.mark.parametrize('fid', range(8))
@pytest.mark.parametrize('shape', [(2, 5, 7, 3), (1, 1, 1, 3), (3, 1, 9, 3), (2, 33, 3, 3)])
def test_ragged_shapes_take_elementwise_path(fid, shape, gpu_device):
  x, dy, params = synthetic.make_case(2000 + fid, shape, np.float16)
  y, dx, dp = run_fwd_bwd(fid, x, dy, params[fid], torch.float16, gpu_device)
  ry, rdx, rdp = oracle(fid, x, dy, params[fid])
  assert_image_close(y, ry, np.float16, 'y')
  assert_image_close(dx, rdx, np.float16, 'dx')
  assert_param_grad_close(dp, rdp, grad_abs(fid, x, dy, params[fid]), 'dp')


def test_unaligned_base_pointer(gpu_device):
  # a view whose data_ptr is 2-byte aligned only -> must fall back to the element path
  x, dy, params = synthetic.make_case(31, (2, 8, 8, 3), np.float16)
  big = torch.zeros(2 * 8 * 8 * 3 + 1, dtype=torch.float16, device=gpu_device)
  tx = big[1:].view(2, 8, 8, 3)
  tx.copy_(torch.from_numpy(x))
  y = torch.empty_like(torch.from_numpy(x)).to(gpu_device)
  tp = torch.from_numpy(params[4]).to(gpu_device)
  _cabi.filter_fwd(4, tx, y, tp)
  ry, _, _ = oracle(4, x, dy, params[4])
  assert_image_close(y.float().cpu().numpy(), ry, np.float16)


@pytest.mark.parametrize('fid', range(8))
def test_dx_optional_and_in_place(fid, gpu_device):
  x, dy, params = synthetic.make_case(3000 + fid, (2, 16, 16, 3), np.float16)
  _, _, dp_ref = run_fwd_bwd(fid, x, dy, params[fid], torch.float16, gpu_device)
  _, none_dx, dp = run_fwd_bwd(fid, x, dy, params[fid], torch.float16, gpu_device, need_dx=False)
  assert none_dx is None
  np.testing.assert_allclose(dp, dp_ref, rtol=1e-5, atol=1e-5)
  # y may alias x, dx may alias dy
  tx = torch.from_numpy(x).to(gpu_device)
  tdy = torch.from_numpy(dy).to(gpu_device)
  tp = torch.from_numpy(params[fid]).to(gpu_device)
  y = torch.empty_like(tx)
  _cabi.filter_fwd(fid, tx, y, tp)
  dx = torch.empty_like(tx)
  dpp = torch.empty_like(tp)
  _cabi.filter_bwd(fid, tx, tdy, dx, tp, dpp)
  _cabi.filter_bwd(fid, tx, tdy, tdy, tp, dpp)
  assert torch.equal(tdy, dx)
  _cabi.filter_fwd(fid, tx, tx, tp)
  assert torch.equal(tx, y)


def test_known_answers_on_device(gpu_device):
  dev = gpu_device
  x = torch.full((1, 8, 8, 3), 0.25, dtype=torch.float32, device=dev)
  y = torch.empty_like(x)
  one = torch.ones((1, 1), device=dev)
  _cabi.filter_fwd(0, x, y, one)
  assert torch.allclose(y, torch.full_like(y, 0.5), atol=1e-6)
  _cabi.filter_fwd(1, x, y, one * 0.5)
  assert torch.allclose(y, torch.full_like(y, 0.5), atol=1e-6)
  k = torch.tensor([[2.0, 1, 1, 1, 1, 1, 1, 1]], device=dev)
  _cabi.filter_fwd(4, x * 0.5, y, k)
  assert torch.allclose(y, torch.full_like(y, 8 / 9 * 0.25), atol=1e-6)
  _cabi.filter_fwd(6, x, y, one * 0)  # WNB p = 0 -> identity
  assert torch.equal(y, x)
  _cabi.filter_fwd(5, x, y, one * 0)  # contrast p = 0 -> identity
  assert torch.allclose(y, x, atol=1e-7)
  grey = torch.full((1, 8, 8, 3), 0.5, dtype=torch.float32, device=dev)
  _cabi.filter_fwd(3, grey, y, one)  # S+ p = 1 on grey: TF hue 0 -> (v, v(1-s'), v(1-s')), s' = .4
  exp = torch.tensor([0.5, 0.3, 0.3], device=dev).expand_as(y)
  assert torch.allclose(y, exp, atol=1e-6)


def test_gradient_ties_follow_tf(gpu_device):
  dev = gpu_device
  # tone: x exactly on knots -> both neighbouring segments contribute (inclusive clip gradient)
  k = torch.arange(1, 9, dtype=torch.float32, device=dev).reshape(1, 8)
  vals = [0.25, 0.0, 1.0, 0.125, 0.875, 1.5, -0.5, 0.3]
  x = torch.tensor(vals, dtype=torch.float32, device=dev).repeat_interleave(3).reshape(1, 1, 8, 3).contiguous()
  dy = torch.ones_like(x)
  dx = torch.empty_like(x)
  dk = torch.empty_like(k)
  _cabi.filter_bwd(4, x, dy, dx, k, dk)
  rdx, rdk = fnp.backward_packed(4, x.cpu().numpy().astype(np.float64), k.cpu().numpy().astype(np.float64),
                                 dy.cpu().numpy().astype(np.float64))
  np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(dk.cpu().numpy(), rdk, rtol=1e-4, atol=1e-5)
  # gamma: x == 0.001 passes, below blocks; S+: x == 1 passes, above blocks
  x = torch.tensor([0.001, 0.0005, 0.5] * 8, dtype=torch.float32, device=dev).reshape(1, 1, 8, 3).contiguous()
  g = torch.full((1, 1), 2.0, device=dev)
  _cabi.filter_bwd(1, x, torch.ones_like(x), dx, g, torch.empty_like(g))
  rdx, _ = fnp.backward_packed(1, x.cpu().numpy().astype(np.float64), np.array([[2.0]]), np.ones((1, 1, 8, 3)))
  np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=1e-5, atol=1e-9)
  x = torch.tensor([1.0, 1.5, 0.2] * 8, dtype=torch.float32, device=dev).reshape(1, 1, 8, 3).contiguous()
  p = torch.full((1, 1), 0.25, device=dev)
  _cabi.filter_bwd(3, x, torch.ones_like(x), dx, p, torch.empty_like(p))
  np.testing.assert_allclose(dx.cpu().numpy()[0, 0, 0], [0.75, 0.0, 0.75], rtol=1e-6)


def test_satplus_analytic_mode(gpu_device):
  x, dy, params = synthetic.make_case(77, (2, 32, 32, 3), np.float32)
  y, dx, dp = run_fwd_bwd(3, x, dy, params[3], torch.float32, gpu_device, mode=1)
  ry, rdx, rdp = oracle(3, x, dy, params[3], mode=1)
  # the analytic HSV gradient has 1/rng and 1/v factors: compare with a relative bound
  err = np.abs(dx - rdx)
  assert (err <= 1e-3 + 1e-3 * np.abs(rdx)).all(), err.max()
  assert_param_grad_close(dp, rdp, grad_abs(3, x, dy, params[3]))


def test_autograd_function_matches_oracle(gpu_device):
  cfg = make_cfg()
  x, dy, params = synthetic.make_case(55, (2, 16, 16, 3), np.float32)
  for fid, cls in enumerate(cfg.filters):
    f = cls((2, 16, 16, 3), cfg).to(gpu_device)
    assert f.filter_id == fid
    tx = torch.from_numpy(x).to(gpu_device).requires_grad_(True)
    ref_shaped = torch.from_numpy(fnp.unpack_params(fid, params[fid])).to(gpu_device).requires_grad_(True)
    low, high, info = f.apply(tx, specified_parameter=ref_shaped)
    assert high is None and 'filter_parameters' in info and 'mask' in info
    low.backward(torch.from_numpy(dy).to(gpu_device))
    ry, rdx, rdp = oracle(fid, x, dy, params[fid])
    assert_image_close(low.detach().cpu().numpy(), ry, np.float32)
    assert_image_close(tx.grad.cpu().numpy(), rdx, np.float32)
    assert_param_grad_close(ref_shaped.grad.reshape(2, -1).cpu().numpy(), rdp, grad_abs(fid, x, dy, params[fid]))


def test_high_res_uses_same_parameters(gpu_device):
  cfg = make_cfg()
  f = filters.ToneFilter((1, 64, 64, 3), cfg).to(gpu_device)
  lo, _, p = synthetic.make_case(8, (2, 64, 64, 3), np.float16)
  hi, _, _ = synthetic.make_case(9, (2, 96, 128, 3), np.float16)
  feats = torch.randn(2, cfg.feature_extractor_dims, device=gpu_device)
  low, high, info = f.apply(torch.from_numpy(lo).to(gpu_device), img_features=feats,
                            high_res=torch.from_numpy(hi).to(gpu_device))
  prm = f.filter_param_regressor(f.extract_parameters(feats)[0]).detach()
  assert prm.shape == (2, 1, 1, 1, 8)
  packed = prm.reshape(2, 8).cpu().numpy()
  assert_image_close(low.detach().float().cpu().numpy(), fnp.process_packed(4, lo.astype(np.float64), packed), np.float16)
  assert_image_close(high.detach().float().cpu().numpy(), fnp.process_packed(4, hi.astype(np.float64), packed), np.float16)


@pytest.mark.parametrize('fid', [0, 4, 7])
def test_low_plus_high_res_node_accumulates_the_parameter_gradient_once(fid, gpu_device):
  """filters.py:88-96: the proxy and the full-resolution image go through the SAME parameters.  The pair node
  (expo_filter_bwd then expo_filter_bwd_accumulate) must give the image gradients of the two single applications
  and a parameter gradient equal to the float64 sum of theirs; an output nobody differentiates costs nothing."""
  dev = gpu_device
  lo, dlo, p = synthetic.make_case(8, (3, 64, 64, 3), np.float16)
  hi, dhi, _ = synthetic.make_case(9, (3, 96, 128, 3), np.float16)
  t = lambda a: torch.from_numpy(a).to(dev)
  packed = t(p[fid]).requires_grad_(True)
  xl, xh = t(lo).requires_grad_(True), t(hi).requires_grad_(True)
  yl, yh = filters._PixelFilterPairFunction.apply(xl, xh, packed, fid, 0)
  gl, gh, gp = torch.autograd.grad([yl, yh], [xl, xh, packed], [t(dlo), t(dhi)])
  rl = fnp.backward_packed(fid, lo.astype(np.float64), p[fid].astype(np.float64), dlo.astype(np.float64))
  rh = fnp.backward_packed(fid, hi.astype(np.float64), p[fid].astype(np.float64), dhi.astype(np.float64))
  assert_image_close(gl.float().cpu().numpy(), rl[0], np.float16, 'pair dx low')
  assert_image_close(gh.float().cpu().numpy(), rh[0], np.float16, 'pair dx high')
  a_lo, a_hi = grad_abs(fid, lo, dlo, p[fid]), grad_abs(fid, hi, dhi, p[fid])
  assert_param_grad_close(gp.cpu().numpy(), rl[1] + rh[1], a_lo + a_hi, 'pair dparams')
  # only the full-resolution output is differentiated: the proxy's pass is skipped, the gradient is the high one
  yl, yh = filters._PixelFilterPairFunction.apply(xl, xh, packed, fid, 0)
  gp_h, = torch.autograd.grad(yh, packed, t(dhi))
  assert_param_grad_close(gp_h.cpu().numpy(), rh[1], a_hi, 'pair dparams (high only)')


def test_chain_matches_stepwise_oracle(gpu_device):
  """Config 2 of BASELINE.json: full 8-filter chain fwd+bwd at 64x64x64x3 fp16, per-step check
  (each step's oracle input is the previous fp16 GPU output, so the bound stays per-pixel)."""
  shape = synthetic.SHAPES['A']
  x, dy, params = synthetic.make_case(1234, shape, np.float16)
  dev = gpu_device
  acts = [torch.from_numpy(x).to(dev)] + [torch.empty(shape, dtype=torch.float16, device=dev) for _ in range(8)]
  prm = [torch.from_numpy(p).to(dev) for p in params]
  _cabi.chain_fwd(list(range(8)), acts, prm)
  grads = [torch.empty(shape, dtype=torch.float16, device=dev) for _ in range(8)] + [torch.from_numpy(dy).to(dev)]
  dprm = [torch.empty_like(p) for p in prm]
  _cabi.chain_bwd(list(range(8)), acts, grads, prm, dprm)
  torch.cuda.synchronize()
  for i in range(8):
    xin = acts[i].cpu().numpy()
    gin = grads[i + 1].cpu().numpy()
    ry, rdx, rdp = oracle(i, xin, gin, params[i])
    assert_image_close(acts[i + 1].float().cpu().numpy(), ry, np.float16, 'chain y step %d' % i)
    assert_image_close(grads[i].float().cpu().numpy(), rdx, np.float16, 'chain dx step %d' % i)
    assert_param_grad_close(dprm[i].cpu().numpy(), rdp, grad_abs(i, xin, gin, params[i]), 'chain dp step %d' % i)


def test_full_size_identity_chain(gpu_device):
  """The headline workload (64x512x512x3 fp16, 8-step chain fwd+bwd through expo_chain_*) with
  every filter at its identity parameters: E 0 EV, gamma 1, WB (1,1,1), S+ 0, flat tone curve,
  contrast 0, BW 0, flat colour curves.  For x in (0.001, 1) every activation equals the input and
  the gradient passes through unchanged -- a size-independent check of the whole chain path."""
  dev = gpu_device
  shape = synthetic.SHAPES['C']
  n = shape[0]
  g = torch.Generator(device=dev).manual_seed(21)
  x = (torch.rand(shape, device=dev, generator=g) * 0.98 + 0.01).half()
  # keep x off the curve knots i/8: exactly on a knot TF's inclusive clip gradient adds both
  # neighbouring slopes (tested in test_gradient_ties_follow_tf), which is not the identity
  x = torch.where((x.float() * 8).frac() == 0, x + 0.001, x)
  dy = torch.randn(shape, device=dev, generator=g).half()
  ones = lambda k, v=1.0: torch.full((n, k), v, device=dev)
  prm = [ones(1, 0.0), ones(1), ones(3), ones(1, 0.0), ones(8, 0.7), ones(1, 0.0), ones(1, 0.0), ones(24, 1.3)]
  acts = [x] + [torch.empty_like(x) for _ in range(8)]
  grads = [torch.empty_like(x) for _ in range(8)] + [dy]
  dprm = [torch.empty_like(p) for p in prm]
  _cabi.chain_fwd(list(range(8)), acts, prm)
  _cabi.chain_bwd(list(range(8)), acts, grads, prm, dprm)
  for i in range(1, 9):
    # E, W, S+, Ct, BW are exact; gamma and the two curves may move a value by one fp16 ulp each
    assert (acts[i].float() - x.float()).abs().max().item() <= 3 * 2.0**-11, i
  for i in range(8):
    assert (grads[i].float() - dy.float()).abs().max().item() <= 3 * 2.0**-8, i  # |dy| < 8: 3 ulps
  assert all(torch.isfinite(d).all() for d in dprm)
  # exposure: d/dEV = ln2 * sum(dy * y); against a float64 sum of the same fp16 tensors
  ref = (grads[1].double() * acts[1].double()).sum(dim=(1, 2, 3)) * np.log(2.0)
  scale = (grads[1].double().abs() * acts[1].double()).sum(dim=(1, 2, 3)) * np.log(2.0)
  assert_param_grad_close(dprm[0][:, 0].cpu().numpy(), ref.cpu().numpy(), scale.cpu().numpy(), 'identity chain dEV')


@pytest.mark.parametrize('shape_name', ['B', 'C'])
def test_full_size_properties(shape_name, gpu_device):
  """BASELINE config 5 size (16x512x512x3 fp16) and the headline size (64x512x512x3 fp16):
  size-independent properties instead of a full oracle run -- identities, linearity of the backward
  in dy, and a sampled oracle check."""
  from oracle import filters_c as fc
  shape = synthetic.SHAPES[shape_name]
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(5)
  x = (torch.rand(shape, device=dev, generator=g)**2.2 * 1.02).half()
  dy = torch.randn(shape, device=dev, generator=g).half()
  n = shape[0]
  y = torch.empty_like(x)
  # identities
  _cabi.filter_fwd(0, x, y, torch.zeros(n, 1, device=dev))
  assert torch.equal(y, x)
  _cabi.filter_fwd(6, x, y, torch.zeros(n, 1, device=dev))
  assert torch.equal(y, x)
  _cabi.filter_fwd(4, x, y, torch.full((n, 8), 1.3, device=dev))
  assert (y.float() - x.float().clamp(0, 1)).abs().max() <= 2.0**-11
  _cabi.filter_fwd(7, x, y, torch.full((n, 24), 0.97, device=dev))
  assert (y.float() - x.float().clamp(0, 1)).abs().max() <= 2.0**-11
  _cabi.filter_fwd(3, x, y, torch.zeros(n, 1, device=dev))
  assert torch.equal(y, x.clamp(max=1.0))
  # backward is linear in dy: bwd(2 dy) == 2 bwd(dy) (power-of-two scaling is exact in fp32; only
  # results that land in the fp16 subnormal range may round differently, by <= 1 subnormal ulp)
  for fid in range(8):
    p = torch.from_numpy(synthetic.make_params(np.random.default_rng(fid), fid, n)).to(dev)
    dx1, dx2 = torch.empty_like(x), torch.empty_like(x)
    dp1, dp2 = torch.empty_like(p), torch.empty_like(p)
    _cabi.filter_bwd(fid, x, dy, dx1, p, dp1)
    _cabi.filter_bwd(fid, x, dy * 2, dx2, p, dp2)
    assert (dx1.float() * 2 - dx2.float()).abs().max().item() <= 2.0**-23, fid
    # parameter gradients: the block records are added in a fixed order, so doubling dy doubles every
    # partial sum exactly (power-of-two scaling) -- bit-identical, not merely close
    assert torch.equal(dp2, 2 * dp1), fid
    # sampled oracle check of the FULL-SIZE launches: forward output and dx on random rows of several
    # images, parameter gradients of two whole images
    rng = np.random.default_rng(100 + fid)
    yf = torch.empty_like(x)
    _cabi.filter_fwd(fid, x, yf, p)
    imgs = sorted(set(int(i) for i in rng.integers(0, n, 6)) | {0, n - 1})
    for i in imgs:
      rows = sorted(set(int(r) for r in rng.integers(0, shape[1], 3)) | ({0, shape[1] - 1} if i == imgs[0] else set()))
      xi = x[i:i + 1, rows].cpu().numpy().astype(np.float64)
      gi = dy[i:i + 1, rows].cpu().numpy().astype(np.float64)
      pi = p[i:i + 1].cpu().numpy().astype(np.float64)
      assert_image_close(yf[i:i + 1, rows].float().cpu().numpy(), fnp.process_packed(fid, xi, pi), np.float16,
                         'full-size fwd filter %d image %d' % (fid, i))
      rdx, _ = fnp.backward_packed(fid, xi, pi, gi)
      assert_image_close(dx1[i:i + 1, rows].float().cpu().numpy(), rdx, np.float16,
                         'full-size dx filter %d image %d' % (fid, i))
    for i in (imgs[0], imgs[-1]):
      xi = x[i:i + 1].cpu().numpy().astype(np.float64)
      gi = dy[i:i + 1].cpu().numpy().astype(np.float64)
      _, rdp, adp = fc.backward_packed(fid, xi, p[i:i + 1].cpu().numpy().astype(np.float64), gi, with_abs=True)
      assert_param_grad_close(dp1[i:i + 1].cpu().numpy(), rdp, adp,
                              'full-size dparams filter %d image %d' % (fid, i))


@pytest.mark.parametrize('shape_name,dtype', [('C', torch.float16), ('B', torch.float32), ('C', torch.float32)])
def test_full_size_chain_every_pixel_against_the_c_oracle(shape_name, dtype, gpu_device):
  """BASELINE's metric shape, 64x512x512x3, in fp16 AND in the reference's own fp32 storage (plus 16x512x512x3 fp32), all 8
  steps forward and backward: EVERY output value of every launch is compared with the float64 C restatement
  (oracle/filters_c.c, OpenMP) evaluated on the very inputs that launch read -- not a sample, not a property."""
  from oracle import filters_c as fc
  dev = gpu_device
  shape = synthetic.SHAPES[shape_name]
  np_dt = NP_DT[dtype]
  try:
    ncpu = len(os.sched_getaffinity(0))
  except (AttributeError, OSError):
    ncpu = os.cpu_count() or 1
  fc.set_threads(max(1, min(64, ncpu // 2)), np.float64)
  x, dy, params = synthetic.make_case(4242, shape, np_dt)
  ids = list(range(8))
  acts = [torch.from_numpy(x).to(dev)] + [torch.empty(shape, dtype=dtype, device=dev) for _ in ids]
  prm = [torch.from_numpy(p).to(dev) for p in params]
  grads = [torch.empty(shape, dtype=dtype, device=dev) for _ in ids] + [torch.from_numpy(dy).to(dev)]
  dprm = [torch.empty_like(p) for p in prm]
  _cabi.chain_fwd(ids, acts, prm)
  _cabi.chain_bwd(ids, acts, grads, prm, dprm)
  torch.cuda.synchronize()

  def check(got, ref, what):
    if np_dt == np.float16:
      np.clip(ref, -65504.0, 65504.0, out=ref)  # fp16 stores saturate
    got = got.cpu().numpy()
    assert_image_close(got, ref, np_dt, what)
    # north_star's PLAIN bound, 1e-3 per pixel: with fp32 storage it holds for every value; with fp16 storage for every
    # value below 2.0 (above, half an fp16 ulp alone exceeds 1e-3 -- that part is covered by the relative term above)
    err = np.abs(got.astype(np.float64) - ref)
    if np_dt == np.float16:
      err = err[np.abs(ref) < 2.0]
    assert float(err.max()) <= 1e-3, '%s: plain 1e-3 bound violated (%.3e)' % (what, err.max())

  for i in ids:
    xin = acts[i].cpu().numpy().astype(np.float64)
    p64 = params[i].astype(np.float64)
    check(acts[i + 1], fc.process_packed(i, xin, p64), 'forward of step %d' % i)
    gin = grads[i + 1].cpu().numpy().astype(np.float64)
    rdx, rdp, adp = fc.backward_packed(i, xin, p64, gin, with_abs=True)
    check(grads[i], rdx, 'dx of step %d' % i)
    # every parameter gradient: |err| <= 1e-4 |ref| + 2e-6 A, A = the oracle's sum of absolute terms (tests/_tol.py)
    assert_param_grad_close(dprm[i].cpu().numpy(), rdp, adp, 'dparams of step %d (%s %s)' % (i, shape_name, np_dt.__name__))
    del xin, gin, rdx


def test_bwd_accumulate(gpu_device):
  x, dy, params = synthetic.make_case(91, (3, 32, 32, 3), np.float16)
  dev = gpu_device
  for fid in (0, 4, 7):
    tx, tdy, tp = (torch.from_numpy(a).to(dev) for a in (x, dy, params[fid]))
    dp = torch.empty_like(tp)
    _cabi.filter_bwd(fid, tx, tdy, None, tp, dp)
    acc = torch.full_like(tp, 1.5)
    _cabi.filter_bwd(fid, tx, tdy, None, tp, acc, accumulate=True)
    _cabi.filter_bwd(fid, tx, tdy, None, tp, acc, accumulate=True)
    scale = dp.abs().max().item() + 1.0
    assert (acc - (1.5 + 2 * dp)).abs().max().item() <= 1e-3 * scale


# ------------------------------------------------------------------ LevelFilter + spatial mask
def masked_grad_abs(fid, x, dy, p, raw, sharp, ms):
  """Scales of the filter-parameter and the RAW mask-parameter gradients of the masked apply: the sum of absolute
  terms for the filter parameters (central differences of the float64 NumPy restatement); for the mask parameters,
  whose terms dy (process - img) dmask/draw contain a subtraction of two fp32 colours, the sum over the operands of that
  subtraction (oracle/filters_np.py::masked_raw_grad_abs)."""
  x64, dy64, p64, raw64 = (a.astype(np.float64) for a in (x, dy, p, raw))
  a_p = fnp.abs_terms_fd(lambda q: fnp.apply_masked(fid, x64, q, raw64, sharp, ms), p64, dy64)
  a_raw = fnp.masked_raw_grad_abs(fid, x64, p64, raw64, dy64, sharp, ms)
  return a_p, a_raw


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_level_filter_matches_oracle(dtype, gpu_device):
  x, dy, _ = synthetic.make_case(401, (3, 32, 48, 3), NP_DT[dtype])
  p = synthetic.make_params(np.random.default_rng(4), 8, 3)
  y, dx, dp = run_fwd_bwd(8, x, dy, p, dtype, gpu_device)
  ry, rdx, rdp = oracle(8, x, dy, p)
  assert_image_close(y, ry, NP_DT[dtype], 'level y')
  assert_image_close(dx, rdx, NP_DT[dtype], 'level dx')
  assert_param_grad_close(dp, rdp, grad_abs(8, x, dy, p), 'level dp')


@pytest.mark.parametrize('fid', range(9))
@pytest.mark.parametrize('shape', [(3, 64, 64, 3), (2, 17, 23, 3), (2, 40, 24, 3)])
def test_masked_apply_matches_oracle(fid, shape, gpu_device):
  """cfg.masking = True path: mask + process + lerp fused in one kernel (filters.py:86-88, 110-148)."""
  from oracle import filters_torch as ft
  from exposure_amd.util import tanh_range
  dev = gpu_device
  rng = np.random.default_rng(500 + fid)
  x, dy, _ = synthetic.make_case(600 + fid, shape, np.float32)
  p = synthetic.make_params(rng, fid, shape[0])
  raw = rng.standard_normal((shape[0], 6)).astype(np.float32)
  sharp, ms = 1.0, 0.3
  ry, rdx, rdp, rdraw = ft.apply_masked_backward(fid, torch.from_numpy(x).double(), torch.from_numpy(p).double(),
                                                 torch.from_numpy(raw).double(), torch.from_numpy(dy).double(), sharp, ms)
  tx = torch.from_numpy(x).to(dev).requires_grad_(True)
  tp = torch.from_numpy(p).to(dev).requires_grad_(True)
  traw = torch.from_numpy(raw).to(dev).requires_grad_(True)
  mp = tanh_range(-5, 5, initial=0)(traw)
  y = filters._MaskedApplyFunction.apply(tx, tp, mp, fid, sharp, ms, 0)
  y.backward(torch.from_numpy(dy).to(dev))
  assert_image_close(y.detach().cpu().numpy(), ry.numpy(), np.float32, 'masked y')
  assert_image_close(tx.grad.cpu().numpy(), rdx.numpy(), np.float32, 'masked dx')
  a_p, a_raw = masked_grad_abs(fid, x, dy, p, raw, sharp, ms)
  assert_param_grad_close(tp.grad.cpu().numpy(), rdp.numpy(), a_p, 'masked dparams fid %d' % fid)
  assert_param_grad_close(traw.grad.cpu().numpy(), rdraw.numpy(), a_raw, 'masked dmask fid %d' % fid)


@pytest.mark.parametrize('fid', [0, 3, 4, 5, 7])
def test_masked_apply_fp16_storage_every_value(fid, gpu_device):
  """The fp16 instantiations of the masked kernels (vector path; Tone / Color re-evaluate their forward through the
  per-wave segment table) on 2x256x256x3: every value of y and dx, the parameter and the mask-parameter gradients
  against the float64 torch restatement."""
  from oracle import filters_torch as ft
  from exposure_amd.util import tanh_range
  dev = gpu_device
  shape = (2, 256, 256, 3)
  rng = np.random.default_rng(900 + fid)
  x, dy, _ = synthetic.make_case(910 + fid, shape, np.float16)
  p = synthetic.make_params(rng, fid, shape[0])
  raw = rng.standard_normal((shape[0], 6)).astype(np.float32)
  sharp, ms = 1.0, 0.3
  ry, rdx, rdp, rdraw = ft.apply_masked_backward(fid, torch.from_numpy(x).double(), torch.from_numpy(p).double(),
                                                 torch.from_numpy(raw).double(), torch.from_numpy(dy).double(), sharp, ms)
  tx = torch.from_numpy(x).to(dev).requires_grad_(True)
  tp = torch.from_numpy(p).to(dev).requires_grad_(True)
  traw = torch.from_numpy(raw).to(dev).requires_grad_(True)
  mp = tanh_range(-5, 5, initial=0)(traw)
  y = filters._MaskedApplyFunction.apply(tx, tp, mp, fid, sharp, ms, 0)
  assert y.dtype == torch.float16
  y.backward(torch.from_numpy(dy).to(dev))
  assert_image_close(y.detach().float().cpu().numpy(), ry.numpy(), np.float16, 'masked y')
  assert_image_close(tx.grad.float().cpu().numpy(), rdx.numpy(), np.float16, 'masked dx')
  a_p, a_raw = masked_grad_abs(fid, x, dy, p, raw, sharp, ms)
  assert_param_grad_close(tp.grad.cpu().numpy(), rdp.numpy(), a_p, 'masked dparams fid %d' % fid)
  assert_param_grad_close(traw.grad.cpu().numpy(), rdraw.numpy(), a_raw, 'masked dmask fid %d' % fid)


def test_filter_apply_with_masking_enabled(gpu_device):
  from oracle import filters_torch as ft
  dev = gpu_device
  torch.manual_seed(0)
  cfg = make_cfg()
  cfg.masking = True
  f = filters.ContrastFilter((1, 32, 32, 3), cfg).to(dev)
  x, _, _ = synthetic.make_case(77, (2, 32, 32, 3), np.float16)
  hi, _, _ = synthetic.make_case(78, (2, 48, 64, 3), np.float16)
  feats = torch.randn(2, cfg.feature_extractor_dims, device=dev)
  low, high, info = f.apply(torch.from_numpy(x).to(dev), img_features=feats, high_res=torch.from_numpy(hi).to(dev))
  ff, mraw = f.extract_parameters(feats)
  p = f.filter_param_regressor(ff).detach().cpu().double()
  for got, src in ((low, x), (high, hi)):
    ref = ft.apply_masked(5, torch.from_numpy(src).double(), p, mraw.detach().cpu().double(), 1.0, 0.3).numpy()
    assert_image_close(got.detach().float().cpu().numpy(), ref, np.float16, 'apply(masking)')
  assert info['mask'].shape == (32, 32, 1)


@pytest.mark.parametrize('fid', range(9))
def test_out_of_range_inputs(fid, gpu_device):
  """Negative, zero, > 1 and exactly-on-threshold pixels (all the clamps / masks of every filter)."""
  rng = np.random.default_rng(700 + fid)
  shape = (2, 24, 16, 3)
  x = rng.uniform(-0.5, 2.0, shape).astype(np.float32)
  flat = x.reshape(-1)
  specials = np.array([0.0, 1.0, 0.001, 0.125, 0.5, 0.875, -0.0, 2.0], dtype=np.float32)
  flat[::11] = specials[np.arange(flat[::11].size) % specials.size]
  x[0, 0, :4] = 0.0  # whole black pixels (lum == 0)
  x[0, 1, :4] = 1.0  # whole white pixels (lum == 1, rng == 0)
  x[0, 2, :4] = -0.25  # all-negative pixels (S+: v <= 0 -> s = 0)
  dy = rng.standard_normal(shape).astype(np.float32)
  p = synthetic.make_params(rng, fid, shape[0])
  y, dx, dp = run_fwd_bwd(fid, x, dy, p, torch.float32, gpu_device)
  ry, rdx, rdp = oracle(fid, x, dy, p)
  assert np.isfinite(y).all() and np.isfinite(dx).all() and np.isfinite(dp).all()
  assert_image_close(y, ry, np.float32, 'y fid %d' % fid)
  # contrast: d/dx has a 1/(lum + 1e-6)^2 factor -> compare relative to the gradient's own scale
  err = np.abs(dx - rdx)
  assert (err <= 2e-4 + 2e-4 * np.abs(rdx)).all(), (fid, err.max())
  assert_param_grad_close(dp, rdp, grad_abs(fid, x, dy, p), 'dp fid %d' % fid)


def test_empty_batch_and_error_paths(gpu_device):
  dev = gpu_device
  x = torch.empty((0, 8, 8, 3), dtype=torch.float16, device=dev)
  _cabi.filter_fwd(0, x, torch.empty_like(x), torch.empty((0, 1), device=dev))  # no-op
  x = torch.zeros((2, 8, 8, 3), dtype=torch.float16, device=dev)
  with pytest.raises(_cabi.ExposureHipError):
    _cabi.filter_fwd(0, x, torch.empty_like(x), torch.zeros((2, 3), device=dev))  # wrong P
  with pytest.raises(_cabi.ExposureHipError):
    _cabi.filter_fwd(0, x.permute(0, 2, 1, 3), torch.empty_like(x), torch.zeros((2, 1), device=dev))  # not contiguous
  with pytest.raises(_cabi.ExposureHipError):
    _cabi.filter_fwd(0, x.double(), torch.empty_like(x).double(), torch.zeros((2, 1), device=dev))  # dtype
  lib = _cabi.load()
  assert lib.expo_filter_fwd(0, x.data_ptr(), x.data_ptr(), x.data_ptr(), 70000, 1, 1, 0, None) == -1  # grid.y limit


def test_single_huge_image_offsets(gpu_device):
  """One 16384 x 16384 fp16 image (1.5 GiB): byte offsets close to the 32-bit limit of the buffer
  resource.  Checked through identities and a sampled oracle comparison at the far end."""
  dev = gpu_device
  h = w = 16384
  g = torch.Generator(device=dev).manual_seed(1)
  x = torch.rand((1, h, w, 3), device=dev, generator=g, dtype=torch.float16)
  y = torch.empty_like(x)
  _cabi.filter_fwd(0, x, y, torch.zeros((1, 1), device=dev))
  assert torch.equal(y, x)
  p = torch.tensor([[1.0]], device=dev)
  _cabi.filter_fwd(0, x, y, p)
  tail = slice(h - 2, h)
  assert torch.equal(y[:, tail], (x[:, tail].float() * 2).half())
  k = torch.from_numpy(synthetic.make_params(np.random.default_rng(3), 4, 1)).to(dev)
  _cabi.filter_fwd(4, x, y, k)
  ref = fnp.process_packed(4, x[:, tail].cpu().numpy().astype(np.float64), k.cpu().numpy().astype(np.float64))
  assert_image_close(y[:, tail].float().cpu().numpy(), ref, np.float16)
  dp = torch.empty_like(k)
  _cabi.filter_bwd(4, x, x, y, k, dp)  # dy := x, dx -> y
  assert torch.isfinite(dp).all() and torch.isfinite(y[:, tail].float()).all()


def test_random_shape_sweep(gpu_device):
  """Seeded sweep over ragged shapes x filters x dtypes: exercises chunk tails of the vector path
  (H*W even / multiple of 8 / of 512 or not), the element-wise path (odd fp16 pixel counts) and
  batches with a single image; forward, backward and the dx-less variant against the oracle."""
  rng = np.random.default_rng(2024)
  dims = [1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 130]
  for case in range(72):
    n = int(rng.integers(1, 5))
    h, w = int(rng.choice(dims)), int(rng.choice(dims))
    fid = case % 9
    dtype = torch.float16 if (case // 9) % 2 == 0 else torch.float32
    x, dy, _ = synthetic.make_case(5000 + case, (n, h, w, 3), NP_DT[dtype])
    p = synthetic.make_params(rng, fid, n)
    y, dx, dp = run_fwd_bwd(fid, x, dy, p, dtype, gpu_device)
    ry, rdx, rdp = oracle(fid, x, dy, p)
    tag = 'case %d fid %d %s %dx%dx%d' % (case, fid, dtype, n, h, w)
    assert_image_close(y, ry, NP_DT[dtype], 'y ' + tag)
    assert_image_close(dx, rdx, NP_DT[dtype], 'dx ' + tag)
    assert_param_grad_close(dp, rdp, grad_abs(fid, x, dy, p), 'dp ' + tag)


def test_fp16_stores_saturate(gpu_device):
  dev = gpu_device
  x = torch.full((1, 8, 8, 3), 60000.0, dtype=torch.float16, device=dev)
  y = torch.empty_like(x)
  _cabi.filter_fwd(0, x, y, torch.full((1, 1), 3.5, device=dev))  # 60000 * 2^3.5 overflows fp16
  assert torch.isfinite(y).all() and float(y.max()) == 65504.0
  x5 = torch.full((1, 3, 5, 3), 60000.0, dtype=torch.float16, device=dev)  # odd pixel count: element path
  y5 = torch.empty_like(x5)
  _cabi.filter_fwd(0, x5, y5, torch.full((1, 1), 3.5, device=dev))
  assert float(y5.max()) == 65504.0
  # negative side, and the backward's dx store (dx = dy * 2^p)
  _cabi.filter_fwd(0, -x, y, torch.full((1, 1), 3.5, device=dev))
  assert torch.isfinite(y).all() and float(y.min()) == -65504.0
  _cabi.filter_fwd(0, -x5, y5, torch.full((1, 1), 3.5, device=dev))
  assert float(y5.min()) == -65504.0
  dx, dp = torch.empty_like(x), torch.empty(1, 1, device=dev)
  _cabi.filter_bwd(0, x * 0 + 1, -x, dx, torch.full((1, 1), 3.5, device=dev), dp)
  assert torch.isfinite(dx).all() and float(dx.min()) == -65504.0


def test_streaming_policy_equals_cached_policy(gpu_device):
  """Tensors >= 8 MiB take the nt-load / write-through-store kernels (IoStream), smaller ones the
  default cache policy.  The policy must not change a single bit of y or dx: run one 12.6 MB batch
  (streaming) and the same images as two 6.3 MB halves (cached)."""
  dev = gpu_device
  shape = (8, 512, 512, 3)
  g = torch.Generator(device=dev).manual_seed(11)
  x = (torch.rand(shape, device=dev, generator=g)**2.2 * 1.02).half()
  dy = torch.randn(shape, device=dev, generator=g).half()
  assert x.numel() * 2 >= (8 << 20) > x[:4].numel() * 2
  for fid in range(9):
    p = torch.from_numpy(synthetic.make_params(np.random.default_rng(40 + fid), fid, shape[0])).to(dev)
    y, dx, dp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(p)
    _cabi.filter_fwd(fid, x, y, p)
    _cabi.filter_bwd(fid, x, dy, dx, p, dp)
    # dx may alias dy (include/exposure_hip.h) -- also under the streaming policy
    inplace, dp_in = dy.clone(), torch.empty_like(p)
    _cabi.filter_bwd(fid, x, inplace, inplace, p, dp_in)
    assert torch.equal(inplace, dx), fid
    for lo in (0, 4):
      xs, dys, ps = x[lo:lo + 4].contiguous(), dy[lo:lo + 4].contiguous(), p[lo:lo + 4].contiguous()
      ys, dxs, dps = torch.empty_like(xs), torch.empty_like(xs), torch.empty_like(ps)
      _cabi.filter_fwd(fid, xs, ys, ps)
      _cabi.filter_bwd(fid, xs, dys, dxs, ps, dps)
      assert torch.equal(ys, y[lo:lo + 4]), fid
      assert torch.equal(dxs, dx[lo:lo + 4]), fid
      scale = dps.abs().max().item() + 1.0
      assert (dps - dp[lo:lo + 4]).abs().max().item() <= 1e-3 * scale, fid  # fp32 sums in another order


@pytest.mark.parametrize('fid', [4, 7])
def test_curve_backward_on_every_fp16_value(fid, gpu_device):
  """The fp16 curve backward looks the slope up by the HIGH BYTE of x's bit pattern (two entries per
  byte: low byte zero / non-zero).  Feed every finite fp16 bit pattern (and +-inf) as x: dx must be
  the oracle's slope -- TF's inclusive knot rule included -- for each of the 63 490 values, and the
  forward must agree on them as well."""
  dev = gpu_device
  bits = np.arange(65536, dtype=np.uint16)
  xs = bits.view(np.float16)
  xs = xs[~np.isnan(xs)]
  rng = np.random.default_rng(7)
  # three channels = three independent permutations of the value set (Color has a curve per channel)
  chans = [xs, rng.permutation(xs), rng.permutation(xs)]
  hw = xs.size
  pad = (-hw) % 8  # keep H*W a multiple of 8 pixels (vector path); pad with zeros
  x = np.stack([np.concatenate([c, np.zeros(pad, np.float16)]) for c in chans], axis=-1).reshape(1, 1, hw + pad, 3)
  p = synthetic.make_params(np.random.default_rng(70 + fid), fid, 1)
  dy = np.ones_like(x)
  y, dx, dp = run_fwd_bwd(fid, x, dy, p, torch.float16, dev)
  finite = np.isfinite(x.astype(np.float64))
  xo = np.where(finite, x.astype(np.float64), np.sign(x.astype(np.float64)) * 1e30)  # +-inf: far outside [0, 1]
  ry = fnp.process_packed(fid, xo, p.astype(np.float64))
  rdx, _ = fnp.backward_packed(fid, xo, p.astype(np.float64), dy.astype(np.float64))
  assert_image_close(y, ry, np.float16, 'y fid %d' % fid)
  # the slope itself is exact in fp32; dx is its fp16 rounding
  err = np.abs(dx.astype(np.float64) - rdx)
  tol = np.abs(rdx) * 2.0**-11 + 1e-6
  bad = err > tol
  assert not bad.any(), 'dx wrong for x = %r (got %r, want %r)' % (x[bad][:8], dx[bad][:8], rdx[bad][:8])


@pytest.mark.parametrize('fid', [4, 7])
@pytest.mark.parametrize('value', [-0.0, 0.0, -1.0, 1.0, 2.0, float('inf'), -float('inf')])
def test_curve_backward_accumulators_at_the_clamp_edges(fid, value, gpu_device):
  """Images made of ONE value at / beyond the ends of [0, 1] (fp16 storage: the packed accumulation clamps with
  one v_pk_max ... clamp and takes its minima on the bit patterns, which must see -0.0 as +0): the parameter
  gradients must be the oracle's -- exactly zero for x <= 0."""
  dev = gpu_device
  shape = (2, 16, 24, 3)
  x = np.full(shape, value, dtype=np.float16)
  rng = np.random.default_rng(11)
  dy = synthetic.make_grad(rng, shape, np.float16)
  p = synthetic.make_params(rng, fid, shape[0])
  _, dx, dp = run_fwd_bwd(fid, x, dy, p, torch.float16, dev)
  xo = np.clip(x.astype(np.float64), -1e30, 1e30)
  rdx, rdp = fnp.backward_packed(fid, xo, p.astype(np.float64), dy.astype(np.float64))
  assert_image_close(dx, rdx, np.float16, 'dx')
  if value <= 0:
    assert np.abs(rdp).max() == 0.0 and np.abs(dp).max() == 0.0, dp
  else:
    # a constant image at / beyond the upper clamp edge: every per-element term is exactly 0 (A = 0); the kernels'
    # two-sum evaluation is judged against the scale of its pieces (oracle/filters_np.py::curve_grad_abs_pieces)
    a3 = fnp.curve_grad_abs_pieces(fid, xo, p.astype(np.float64), dy.astype(np.float64))
    assert_param_grad_close(dp, rdp, a3, 'curve dparams on a constant image %r' % value)


def test_dispatch_streaming_policy_equals_cached_policy(gpu_device):
  """Same check for the per-image dispatch (the agent's one-hot select): 10 images x 512x512 (15.7 MB,
  streaming kernels) against the same images in two 5-image calls (7.9 MB, cached kernels); every
  filter id incl. the all-zero one-hot (-1), with the fused penalty and its gradient."""
  dev = gpu_device
  n, shape = 10, (10, 512, 512, 3)
  g = torch.Generator(device=dev).manual_seed(12)
  x = (torch.rand(shape, device=dev, generator=g)**2.2 * 1.3).half()
  dy = torch.randn(shape, device=dev, generator=g).half()
  assert x.numel() * 2 >= (8 << 20) > x[:5].numel() * 2
  ids = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, -1], dtype=torch.int32, device=dev)
  rng = np.random.default_rng(13)
  p24 = torch.zeros((n, 24), device=dev)
  for i, fid in enumerate(ids.tolist()):
    if fid >= 0:
      q = synthetic.make_params(rng, fid, 1)[0]
      p24[i, :q.size] = torch.from_numpy(q).to(dev)
  dpen = torch.rand(n, device=dev, generator=g)
  y, pen = torch.empty_like(x), torch.empty(n, device=dev)
  dx, dp = torch.empty_like(x), torch.empty_like(p24)
  _cabi.dispatch_fwd(ids, x, y, p24, pen)
  _cabi.dispatch_bwd(ids, x, dy, dx, p24, dp, dpen)
  for lo in (0, 5):
    sl = slice(lo, lo + 5)
    xs, dys = x[sl].contiguous(), dy[sl].contiguous()
    ys, pens = torch.empty_like(xs), torch.empty(5, device=dev)
    dxs, dps = torch.empty_like(xs), torch.empty((5, 24), device=dev)
    _cabi.dispatch_fwd(ids[sl].contiguous(), xs, ys, p24[sl].contiguous(), pens)
    _cabi.dispatch_bwd(ids[sl].contiguous(), xs, dys, dxs, p24[sl].contiguous(), dps, dpen[sl].contiguous())
    assert torch.equal(ys, y[sl]) and torch.equal(dxs, dx[sl])
    assert torch.allclose(pens, pen[sl], rtol=1e-4, atol=1e-7)
    scale = dps.abs().max().item() + 1.0
    assert (dps - dp[sl]).abs().max().item() <= 1e-3 * scale


def test_chain_calls_on_two_user_streams_share_the_helper_stream(gpu_device):
  """expo_chain_fwd / _bwd fork to a library-owned helper stream (since round 4 one per device AND caller
  stream -- the name of this test is round 3's).  Two chains enqueued from two user streams (own workspaces) must each
  be ordered only through their own stream: results bit-identical to the same chains run one after the other, and the
  split must actually be on for this shape."""
  dev = gpu_device
  shape = (28, 512, 512, 3)  # 44 MB per tensor: inside the [40 MiB, 256 MiB) gate
  assert _cabi.chain_streams(shape[0], shape[1], shape[2], _cabi.EXPO_F16) == 2
  assert _cabi.chain_streams(16, 512, 512, _cabi.EXPO_F16) == 1 and _cabi.chain_streams(64, 64, 64, _cabi.EXPO_F16) == 1
  ids = list(range(8))

  def make(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = (torch.rand(shape, device=dev, generator=g)**2.2).half()
    dy = torch.randn(shape, device=dev, generator=g).half()
    rng = np.random.default_rng(seed)
    prm = [torch.from_numpy(synthetic.make_params(rng, f, shape[0])).to(dev) for f in ids]
    acts = [x] + [torch.empty_like(x) for _ in ids]
    grads = [torch.empty_like(x) for _ in ids] + [dy]
    dprm = [torch.empty_like(p) for p in prm]
    ws = _cabi.new_workspace(dev, _cabi.workspace_bytes(shape[0], shape[1], shape[2], _cabi.EXPO_F16, 8))
    return acts, grads, prm, dprm, ws

  def run(c):
    acts, grads, prm, dprm, ws = c
    _cabi.chain_fwd(ids, acts, prm)
    _cabi.chain_bwd(ids, acts, grads, prm, dprm, workspace=ws)

  a, b = make(1), make(2)
  run(a), run(b)
  torch.cuda.synchronize()
  want = [[t.clone() for t in (c[0][8], c[1][0])] + [d.clone() for d in c[3]] for c in (a, b)]
  for c in (a, b):  # scribble over the outputs
    for t in c[0][1:] + c[1][:8] + c[3]:
      t.fill_(7.0)
  s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
  torch.cuda.synchronize()
  for _ in range(3):  # interleave the two callers
    with torch.cuda.stream(s1):
      run(a)
    with torch.cuda.stream(s2):
      run(b)
  torch.cuda.synchronize()
  for c, w in zip((a, b), want):
    got = [c[0][8], c[1][0]] + list(c[3])
    for g, r in zip(got, w):
      assert torch.equal(g, r)


def test_split_chain_replayed_from_a_hipgraph_equals_eager(gpu_device):
  """The fork / join of the two half-batch streams (one pair of library-owned events, re-recorded by every call) is
  captured as parallel branches: a graph holding TWO chain steps must reproduce the eager results bit for bit, replay
  after replay, and eager calls must keep working after the capture."""
  dev = gpu_device
  shape = (28, 512, 512, 3)
  assert _cabi.chain_streams(shape[0], shape[1], shape[2], _cabi.EXPO_F16) == 2
  ids = list(range(8))
  g = torch.Generator(device=dev).manual_seed(9)
  x = (torch.rand(shape, device=dev, generator=g)**2.2).half()
  dy = torch.randn(shape, device=dev, generator=g).half()
  rng = np.random.default_rng(9)
  prm = [torch.from_numpy(synthetic.make_params(rng, f, shape[0])).to(dev) for f in ids]
  acts = [x] + [torch.empty_like(x) for _ in ids]
  grads = [torch.empty_like(x) for _ in ids] + [dy]
  dprm = [torch.empty_like(p) for p in prm]
  ws = _cabi.new_workspace(dev, _cabi.workspace_bytes(shape[0], shape[1], shape[2], _cabi.EXPO_F16, 8))

  def step():
    _cabi.chain_fwd(ids, acts, prm)
    _cabi.chain_bwd(ids, acts, grads, prm, dprm, workspace=ws)

  step()
  torch.cuda.synchronize()
  want = [acts[8].clone(), grads[0].clone()] + [d.clone() for d in dprm]
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    step()
  torch.cuda.current_stream().wait_stream(side)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  from exposure_amd.util import capture_without_gc
  with capture_without_gc(), torch.cuda.graph(graph):
    step()
    step()
  for _ in range(3):
    for t in [acts[8], grads[0]] + dprm:
      t.fill_(3.0)
    graph.replay()
    torch.cuda.synchronize()
    for got, ref in zip([acts[8], grads[0]] + dprm, want):
      assert torch.equal(got, ref)
  for t in [acts[8], grads[0]] + dprm:
    t.fill_(3.0)
  step()  # eager again, after the events were last recorded inside a capture
  torch.cuda.synchronize()
  for got, ref in zip([acts[8], grads[0]] + dprm, want):
    assert torch.equal(got, ref)


def test_every_caller_stream_gets_a_helper_on_another_hardware_queue(gpu_device):
  """The runtime multiplexes a process's streams over a few hardware queues; a helper stream that lands on its caller's
  queue serialises the two lanes of a chain call (r04p14: 2.63 ms instead of 2.31 at 256x512x512).  The library probes
  each (caller, helper) pairing on the caller's first eager two-lane call (exposure_hip.hip::fork_join_for_device).
  Six caller streams -- more than there are hardware queues: every one is probed exactly once, results are identical
  on all of them, and no caller's chain step is more than 10 % slower than the fastest caller's (an aliased pair is
  +14 %)."""
  dev = gpu_device
  shape = (64, 512, 512, 3)
  assert _cabi.chain_streams(shape[0], shape[1], shape[2], _cabi.EXPO_F16) == 2
  ids = list(range(8))
  g = torch.Generator(device=dev).manual_seed(21)
  x = (torch.rand(shape, device=dev, generator=g)**2.2).half()
  dy = torch.randn(shape, device=dev, generator=g).half()
  rng = np.random.default_rng(21)
  prm = [torch.from_numpy(synthetic.make_params(rng, f, shape[0])).to(dev) for f in ids]
  acts = [x] + [torch.empty_like(x) for _ in ids]
  grads = [torch.empty_like(x) for _ in ids] + [dy]
  dprm = [torch.empty_like(p) for p in prm]
  ws = _cabi.new_workspace(dev, _cabi.workspace_bytes(shape[0], shape[1], shape[2], _cabi.EXPO_F16, 8))

  def step():
    _cabi.chain_fwd(ids, acts, prm)
    _cabi.chain_bwd(ids, acts, grads, prm, dprm, workspace=ws)

  step()
  torch.cuda.synchronize()
  want = [acts[8].clone(), grads[0].clone()] + [d.clone() for d in dprm]
  streams = [torch.cuda.Stream() for _ in range(6)]
  assert len({s.cuda_stream for s in streams}) == 6
  probed0, _ = _cabi.chain_helper_stats()
  ms = []
  for s in streams:
    for t in [acts[8], grads[0]] + dprm:
      t.fill_(3.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
      step()  # probes this caller's helper
      step()
      runs = []
      for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
          step()
        e1.record()
        e1.synchronize()
        runs.append(e0.elapsed_time(e1) / 4)
    torch.cuda.synchronize()
    ms.append(sorted(runs)[2])
    for got, ref in zip([acts[8], grads[0]] + dprm, want):
      assert torch.equal(got, ref)
  probed1, rejected = _cabi.chain_helper_stats()
  assert probed1 - probed0 >= 6  # one probe per new caller, more where a helper was rejected
  with torch.cuda.stream(streams[0]):
    step()
  torch.cuda.synchronize()
  assert _cabi.chain_helper_stats()[0] == probed1  # a known caller is not probed again
  print('chain step per caller stream (ms): %s; helpers rejected so far: %d' % (' '.join('%.4f' % v for v in ms), rejected))
  assert max(ms) <= 1.10 * min(ms), ms
  # expo_chain_prepare: the probe as an explicit set-up call -- the first chain call of that stream finds it done;
  # idempotent; expo_chain_release forgets a pairing (the next call of that stream starts over)
  fresh = torch.cuda.Stream()
  base = _cabi.chain_helper_stats()[0]
  _cabi.chain_prepare(fresh)
  after = _cabi.chain_helper_stats()[0]
  assert after > base
  _cabi.chain_prepare(fresh)
  with torch.cuda.stream(fresh):
    step()
  torch.cuda.synchronize()
  assert _cabi.chain_helper_stats()[0] == after
  for got, ref in zip([acts[8], grads[0]] + dprm, want):
    assert torch.equal(got, ref)
  _cabi.chain_release(fresh)
  _cabi.chain_release(fresh)  # (unknown now: still OK)
  with torch.cuda.stream(fresh):
    step()
  torch.cuda.synchronize()
  assert _cabi.chain_helper_stats()[0] > after
  for got, ref in zip([acts[8], grads[0]] + dprm, want):
    assert torch.equal(got, ref)
