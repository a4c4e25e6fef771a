"""cfg.curve_steps other than 8 (filters.py:264-273, 312-322 loop over cfg.curve_steps; config_example.py:27 sets 8).

CPU: the oracle restatements are step-count aware (three agree, finite differences), the Filter classes size their
heads by it, and the agent takes the stack-and-select path through the (mocked) generic entry points.
GPU (-m gpu): expo_curve_fwd / expo_curve_bwd against the float64 oracle for several step counts incl. non powers of
two, the generic kernels at L = 8 against the tuned ones, the Filter protocol and one agent / GAN step at L = 4 and 16."""
import numpy as np
import pytest
import torch

from exposure_amd import _cabi, filters, synthetic
from exposure_amd.config import make_cfg
from oracle import filters_np as fnp
from oracle import filters_torch as ft
from tests._tol import assert_image_close, assert_param_grad_close

NP_DT = {torch.float16: np.float16, torch.float32: np.float32}


def curve_params(rng, n, curves, steps):
  lo, hi = (0.5, 2.0) if curves == 1 else (0.9, 1.1)
  return rng.uniform(lo, hi, (n, curves * steps)).astype(np.float32)


@pytest.mark.parametrize('steps', [1, 4, 5, 12, 16])
@pytest.mark.parametrize('curves', [1, 3])
def test_oracles_agree_for_any_step_count(curves, steps):
  rng = np.random.default_rng(steps * 10 + curves)
  fid = 4 if curves == 1 else 7
  x = rng.uniform(-0.3, 1.4, (2, 9, 7, 3))
  dy = rng.standard_normal(x.shape)
  p = curve_params(rng, 2, curves, steps).astype(np.float64)
  y = fnp.process_packed(fid, x, p)
  # literal restatement of the reference loop
  k = p.reshape(2, 1, 1, curves, steps)
  ref = sum(np.clip(x - i / steps, 0, 1.0 / steps) * k[..., i] for i in range(steps)) * (steps / (k.sum(axis=4) + 1e-30))
  np.testing.assert_allclose(y, ref, rtol=1e-13, atol=1e-14)
  ty = ft.process_packed(fid, torch.from_numpy(x), torch.from_numpy(p)).numpy()
  np.testing.assert_allclose(ty, y, rtol=1e-12, atol=1e-13)
  dx, dp = fnp.backward_packed(fid, x, p, dy)
  tdx, tdp = ft.backward_packed(fid, torch.from_numpy(x), torch.from_numpy(p), torch.from_numpy(dy))
  np.testing.assert_allclose(tdx.numpy(), dx, rtol=1e-10, atol=1e-12)
  np.testing.assert_allclose(tdp.numpy(), dp, rtol=1e-9, atol=1e-10)
  terms = fnp.param_grad_terms(fid, x, p, dy)
  assert terms.shape[-1] == curves * steps
  np.testing.assert_allclose(terms.sum(axis=(1, 2, 3)), dp, rtol=1e-10, atol=1e-11)
  assert (fnp.curve_grad_abs_pieces(fid, x, p, dy) >= fnp.param_grad_abs(fid, x, p, dy) * (1 - 1e-12)).all()


def test_filter_classes_follow_cfg_curve_steps():
  cfg = make_cfg()
  cfg.curve_steps = 4
  tone = filters.ToneFilter((1, 64, 64, 3), cfg)
  color = filters.ColorFilter((1, 64, 64, 3), cfg)
  assert tone.get_num_filter_parameters() == 4 and color.get_num_filter_parameters() == 12
  assert tone.fc2.out_features == 4 + 6 and color.fc2.out_features == 12 + 6
  assert tone.uses_generic_kernels() and color.uses_generic_kernels()
  f = torch.randn(3, 12)
  assert color.filter_param_regressor(f).shape == (3, 1, 1, 3, 4)
  assert tone.filter_param_regressor(f[:, :4]).shape == (3, 1, 1, 1, 4)
  cfg8 = make_cfg()
  assert not filters.ToneFilter((1, 64, 64, 3), cfg8).uses_generic_kernels()
  cfg.curve_steps = 17
  with pytest.raises(AssertionError):
    filters.ToneFilter((1, 64, 64, 3), cfg)


@pytest.mark.parametrize('steps', [4, 16])
def test_agent_and_gan_step_with_other_curve_steps_cpu(steps):
  """The agent's stack-and-select path and one G / C step, C-ABI mocked by the oracle; output vs the NumPy agent."""
  from exposure_amd import checkpoint
  from exposure_amd.gan import GAN
  from oracle import nets_np as nn_np
  from tests._fake_hip import fake_hip
  from tests.test_oracle_nets import make_batch, spread_selection_noise
  torch.manual_seed(1)
  cfg = make_cfg()
  cfg.curve_steps = steps
  gan = GAN(cfg)
  n = 8
  fake_input, real, states, z, masks, alpha = make_batch(n, 3)
  t = torch.from_numpy
  with fake_hip():
    z = spread_selection_noise(gan, torch.device('cpu'), fake_input, z, states, masks)
    out = gan.generator_losses(t(fake_input), t(z), t(states), 0.3, 1, [t(m) for m in masks])
    ocfg = dict(nn_np.DEFAULT_CFG, curve_steps=steps)
    weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
    d = lambda a: a.astype(np.float64)
    ref = nn_np.generator_losses(d(fake_input), d(z), d(states), 0.3, ocfg, weights, [d(m) for m in masks], 1)
    assert np.array_equal(out['debug']['selected_filter_ids'].numpy(), ref['debug']['selected_filter_id'])
    assert set(ref['debug']['selected_filter_id'].tolist()) >= {4, 7}, 'the batch must select both curve filters'
    err = np.abs(out['fake_output'].detach().numpy() - ref['fake_output'])
    assert (err <= 1e-4 + 1e-4 * np.abs(ref['fake_output'])).all(), err.max()
    for key in ('g_loss', 'v_loss'):
      assert abs(float(out[key].detach()) - ref[key]) <= 1e-4 * max(1.0, abs(ref[key])), key
    g = gan.generator_step(t(fake_input), t(z), t(states), 0.3, it=2)
    assert torch.isfinite(g['g_loss'])
    c = gan.critic_step(t(real), g['fake_output'], it=2)
    assert torch.isfinite(c['c_loss'])


# ------------------------------------------------------------------------------------------------ GPU
def run_curve(x, dy, p, curves, steps, dtype, dev, need_dx=True):
  tx, tdy, tp = (torch.from_numpy(a).to(dev) for a in (x, dy, p))
  tx, tdy = tx.to(dtype), tdy.to(dtype)
  y = torch.empty_like(tx)
  _cabi.curve_fwd(tx, y, tp, curves, steps)
  dx = torch.full_like(tx, 3.0) if need_dx else None
  dp = torch.full_like(tp, float('nan'))
  _cabi.curve_bwd(tx, tdy, dx, tp, dp, curves, steps)
  torch.cuda.synchronize()
  return y.float().cpu().numpy(), None if dx is None else dx.float().cpu().numpy(), dp.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('steps', [1, 4, 5, 8, 12, 16])
@pytest.mark.parametrize('curves', [1, 3])
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(3, 64, 64, 3), (2, 7, 5, 3), (1, 256, 192, 3)])
def test_generic_curve_kernels_match_oracle(steps, curves, dtype, shape, gpu_device):
  rng = np.random.default_rng(steps * 100 + curves * 10 + shape[1])
  fid = 4 if curves == 1 else 7
  x = (synthetic.make_images(rng, shape, NP_DT[dtype]).astype(np.float32) * 1.3 - 0.1).astype(NP_DT[dtype])
  if steps in (4, 8, 16):  # exact knots (representable for powers of two): TF's inclusive clip gradient
    flat = x.reshape(-1)
    flat[::41] = (np.arange(flat[::41].size) % (steps + 1)).astype(np.float32) / steps
  dy = synthetic.make_grad(rng, shape, NP_DT[dtype])
  p = curve_params(rng, shape[0], curves, steps)
  y, dx, dp = run_curve(x, dy, p, curves, steps, dtype, gpu_device)
  x64, dy64, p64 = x.astype(np.float64), dy.astype(np.float64), p.astype(np.float64)
  ry = fnp.process_packed(fid, x64, p64)
  rdx, rdp = fnp.backward_packed(fid, x64, p64, dy64)
  assert_image_close(y, ry, NP_DT[dtype], 'generic curve y (L = %d)' % steps)
  # For step counts that are not powers of two the knots i/L are not representable: whether an input within rounding of
  # a knot (0.25 = 3/12 ...) counts as ON it -- both neighbouring slopes, TF's inclusive clip gradient -- is decided by
  # the rounding of `x - i/L` in whatever dtype evaluates it (float64 here, fp32 in TF, exact u = L x in the kernel).
  # Those elements have two equally valid sub-gradients and are left out of the dx comparison.
  on_knot = np.abs(x64 * steps - np.round(x64 * steps)) < 1e-6 if steps & (steps - 1) else np.zeros(x64.shape, bool)
  assert on_knot.mean() < 0.01
  assert_image_close(np.where(on_knot, 0, dx), np.where(on_knot, 0, rdx), NP_DT[dtype], 'generic curve dx (L = %d)' % steps)
  # (one step: y = clip(x, 0, 1) whatever k is -- every term of dk is exactly 0, A = 0; the kernels' two-sum evaluation is
  # judged against the scale of its pieces there, oracle/filters_np.py::curve_grad_abs_pieces)
  a_of = fnp.curve_grad_abs_pieces if steps == 1 else fnp.param_grad_abs
  assert_param_grad_close(dp, rdp, a_of(fid, x64, p64, dy64),
                          'generic curve dparams L=%d curves=%d %s' % (steps, curves, NP_DT[dtype].__name__))
  # dx optional; bit-reproducible
  _, none_dx, dp2 = run_curve(x, dy, p, curves, steps, dtype, gpu_device, need_dx=False)
  assert none_dx is None and np.array_equal(dp2.view(np.uint32), dp.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize('curves', [1, 3])
def test_generic_kernels_at_eight_steps_equal_the_tuned_ones(curves, gpu_device):
  rng = np.random.default_rng(4)
  fid = 4 if curves == 1 else 7
  shape = (3, 96, 80, 3)
  x, dy, params = synthetic.make_case(12, shape, np.float16)
  p = params[fid]
  y, dx, dp = run_curve(x, dy, p, curves, 8, torch.float16, gpu_device)
  tx, tdy, tp = (torch.from_numpy(a).to(gpu_device) for a in (x, dy, p))
  ty, tdx, tdp = torch.empty_like(tx), torch.empty_like(tx), torch.empty_like(tp)
  _cabi.filter_fwd(fid, tx, ty, tp)
  _cabi.filter_bwd(fid, tx, tdy, tdx, tp, tdp)
  assert np.abs(y - ty.float().cpu().numpy()).max() <= 2.0**-10  # each within half an fp16 ulp of the exact value
  assert np.abs(dx - tdx.float().cpu().numpy()).max() <= 2.0**-8
  a = fnp.param_grad_abs(fid, x.astype(np.float64), p.astype(np.float64), dy.astype(np.float64))
  assert_param_grad_close(dp, tdp.cpu().numpy(), 2 * a, 'generic vs tuned dparams')


@pytest.mark.gpu
def test_curve_entry_points_refuse_bad_arguments(gpu_device):
  x = torch.zeros((2, 8, 8, 3), dtype=torch.float16, device=gpu_device)
  p = torch.ones((2, 17), device=gpu_device)
  with pytest.raises(_cabi.ExposureHipError, match='steps'):
    _cabi.curve_fwd(x, torch.empty_like(x), p, 1, 17)
  lib = _cabi.load()
  assert lib.expo_curve_workspace_bytes(2, 8, 8, 2, 8) == 0 and lib.expo_curve_workspace_bytes(2, 8, 8, 3, 8) > 0
  rc = lib.expo_curve_bwd(x.data_ptr(), x.data_ptr(), None, p.data_ptr(), p.data_ptr(), 2, 8, 8, 0, 1, 8, None, 0, None)
  assert rc == -1 and b'workspace' in lib.expo_last_error()
  _cabi.curve_fwd(x[:0], torch.empty_like(x[:0]), p[:0, :8].contiguous(), 1, 8)  # empty batch: no-op


@pytest.mark.gpu
@pytest.mark.parametrize('steps', [4, 16])
def test_filter_protocol_and_agent_step_with_other_curve_steps_gpu(steps, gpu_device):
  from exposure_amd import checkpoint
  from exposure_amd.gan import GAN
  from oracle import nets_np as nn_np
  from tests.test_oracle_nets import make_batch, spread_selection_noise
  dev = gpu_device
  cfg = make_cfg()
  cfg.curve_steps = steps
  # Filter.apply with a specified parameter, low + high resolution, autograd
  rng = np.random.default_rng(steps)
  tone = filters.ToneFilter((1, 64, 64, 3), cfg).to(dev)
  lo, dlo, _ = synthetic.make_case(3, (2, 64, 64, 3), np.float16)
  hi, _, _ = synthetic.make_case(4, (2, 96, 128, 3), np.float16)
  p = curve_params(rng, 2, 1, steps)
  tp = torch.from_numpy(p.reshape(2, 1, 1, 1, steps)).to(dev).requires_grad_(True)
  low, high, info = tone.apply(torch.from_numpy(lo).to(dev), specified_parameter=tp, high_res=torch.from_numpy(hi).to(dev))
  assert_image_close(low.detach().float().cpu().numpy(), fnp.process_packed(4, lo.astype(np.float64), p.astype(np.float64)),
                     np.float16, 'tone low')
  assert_image_close(high.detach().float().cpu().numpy(), fnp.process_packed(4, hi.astype(np.float64), p.astype(np.float64)),
                     np.float16, 'tone high')
  low.backward(torch.from_numpy(dlo).to(dev))
  _, rdp = fnp.backward_packed(4, lo.astype(np.float64), p.astype(np.float64), dlo.astype(np.float64))
  assert_param_grad_close(tp.grad.reshape(2, steps).cpu().numpy(), rdp,
                          fnp.param_grad_abs(4, lo.astype(np.float64), p.astype(np.float64), dlo.astype(np.float64)), 'tone dparams')
  # one generator-loss evaluation against the NumPy agent, then a G and a C step
  torch.manual_seed(1)
  gan = GAN(cfg, device=dev)
  n = 8
  fake_input, real, states, z, masks, alpha = make_batch(n, 3)
  t = lambda a: torch.from_numpy(a).to(dev)
  z = spread_selection_noise(gan, dev, fake_input, z, states, masks)
  out = gan.generator_losses(t(fake_input), t(z), t(states), 0.3, 1, [t(m) for m in masks])
  ocfg = dict(nn_np.DEFAULT_CFG, curve_steps=steps)
  weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
  d = lambda a: a.astype(np.float64)
  ref = nn_np.generator_losses(d(fake_input), d(z), d(states), 0.3, ocfg, weights, [d(m) for m in masks], 1)
  assert np.array_equal(out['debug']['selected_filter_ids'].cpu().numpy(), ref['debug']['selected_filter_id'])
  err = np.abs(out['fake_output'].detach().float().cpu().numpy() - ref['fake_output'])
  assert (err <= 1e-3 + 1e-3 * np.abs(ref['fake_output'])).all(), err.max()
  for key in ('g_loss', 'v_loss'):
    assert abs(float(out[key].detach()) - ref[key]) <= 1e-4 * max(1.0, abs(ref[key])), key
  g = gan.generator_step(t(fake_input), t(z), t(states), 0.3, it=2)
  c = gan.critic_step(t(real), g['fake_output'], it=2)
  assert torch.isfinite(g['g_loss']) and torch.isfinite(c['c_loss'])
