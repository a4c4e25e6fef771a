"""COMPOSITION PIN (TF primitive semantics assumed): the fixture tests/golden/reference_facade.npz holds what the BODIES
of the reference's TensorFlow methods compute when executed in the build container with a NumPy facade in place of `tf`
(tests/golden/make_reference_facade.py: `process` / `filter_param_regressor` of the eight filters and LevelFilter,
filters.py:177-498; lrelu / rgb2lum / tanh_range / lerp, util.py:225-308; pdf_sample, pdf_sample_layer.py:5-10; the
selection / state / penalty statements of agent_generator, agent.py:100-123, 208-252).  Held against it:

  CPU  the three filter oracles (NumPy float64, torch, C), the host-side regressors of exposure_amd.filters, pdf_sample and
       the agent arithmetic of oracle/agent_np.py and exposure_amd/agent.py;
  GPU  expo_filter_fwd in fp32 storage (1e-5), expo_agent_select_fwd (ids and states bit-equal), the penalty kernel.

It pins how the restatements COMPOSE the primitives (operand order, broadcast axes, which tensor is clamped before which
blend, knot indexing, the epsilon guards, exclusive-cumsum sampling incl. id -1 at noise 0) -- not the primitives (the
facade's own clip / maximum / HSV are stand-ins), and no gradient."""
import os

import numpy as np
import pytest
import torch

from oracle import agent_np
from oracle import filters_np as fnp

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FILTERS = ['ExposureFilter', 'GammaFilter', 'ImprovedWhiteBalanceFilter', 'SaturationPlusFilter', 'ToneFilter',
           'ContrastFilter', 'WNBFilter', 'ColorFilter', 'LevelFilter']  # C-ABI ids 0 .. 8


@pytest.fixture(scope='module')
def ref():
  return np.load(os.path.join(HERE, 'reference_facade.npz'))


def test_fixture_names_its_sources_and_its_stand_ins(ref):
  prov = [str(p) for p in ref['provenance']]
  assert [p.split(' ')[0] for p in prov] == ['util.py', 'filters.py', 'pdf_sample_layer.py', 'agent.py']
  assert all(len(p.split('sha256=')[1]) == 64 for p in prov)
  ops = set(str(o) for o in ref['facade_ops'])
  # exactly the primitives the cut-out bodies call: a new one would mean the reference text changed under the fixture
  assert ops == {'abs', 'argmax', 'cast', 'clip_by_value', 'concat', 'cos', 'cumsum', 'exp', 'less', 'log', 'maximum', 'minimum',
                 'nn.softmax', 'one_hot', 'pow', 'reduce_mean', 'reduce_sum', 'reshape', 'sigmoid', 'tanh',
                 'image.rgb_to_hsv [matplotlib.colors]', 'image.hsv_to_rgb [matplotlib.colors]',
                 'image.rgb_to_hsv [TF formula: pixels outside [0, 1]]', 'image.hsv_to_rgb [TF formula: pixels outside [0, 1]]'}
  x = ref['x']
  assert (x < 0).any() and (x > 1).any() and (x[0, 0, :8, 0] == np.arange(8) / 8.0).all()  # knots, out-of-range channels


@pytest.mark.parametrize('fid', range(9))
def test_numpy_oracle_composes_like_the_reference(ref, fid):
  name = FILTERS[fid]
  f, want_p, want_y = ref[name + '_features'], ref[name + '_params'], ref[name + '_y']
  packed = fnp.regress_packed(fid, f)
  np.testing.assert_allclose(packed, want_p.reshape(want_p.shape[0], -1), rtol=1e-13, atol=1e-15, err_msg=name + ' regressor')
  # the reference-shaped parameter tensor has the shape filters.py gives it (N x 1 x 1 x C x L for the curves)
  assert fnp.unpack_params(fid, packed).shape == want_p.shape
  got = fnp.process_packed(fid, ref['x'], packed)
  np.testing.assert_allclose(got, want_y, rtol=1e-12, atol=1e-14, err_msg=name + ' process')


@pytest.mark.parametrize('fid', range(8))
def test_torch_and_c_oracles_compose_like_the_reference(ref, fid):
  from oracle import filters_c as fc
  from oracle import filters_torch as ft
  name = FILTERS[fid]
  want_p, want_y = ref[name + '_params'], ref[name + '_y']
  packed = want_p.reshape(want_p.shape[0], -1)
  got_t = ft.process_packed(fid, torch.from_numpy(ref['x']), torch.from_numpy(packed)).numpy()
  np.testing.assert_allclose(got_t, want_y, rtol=1e-12, atol=1e-14, err_msg=name + ' (torch oracle)')
  got_c = fc.process_packed(fid, ref['x'], packed)
  np.testing.assert_allclose(got_c, want_y, rtol=1e-11, atol=1e-13, err_msg=name + ' (C oracle)')


def test_product_regressors_compose_like_the_reference(ref):
  """exposure_amd.filters.<Filter>.filter_param_regressor (torch, host side) on the fixture's raw features."""
  from exposure_amd import filters as F
  from exposure_amd.config import make_cfg
  cfg = make_cfg()
  classes = [F.ExposureFilter, F.GammaFilter, F.ImprovedWhiteBalanceFilter, F.SaturationPlusFilter, F.ToneFilter,
             F.ContrastFilter, F.WNBFilter, F.ColorFilter, F.LevelFilter]
  for name, cls in zip(FILTERS, classes):
    filt = cls((1, 64, 64, 3), cfg)
    got = filt.filter_param_regressor(torch.from_numpy(ref[name + '_features']))
    want = ref[name + '_params']
    assert tuple(got.shape) == want.shape, (name, tuple(got.shape), want.shape)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=1e-14, err_msg=name)


def test_helpers_and_sampling_compose_like_the_reference(ref):
  from exposure_amd import agent as xagent
  from exposure_amd import util as xutil
  x = ref['util_lrelu_x']
  np.testing.assert_allclose(0.6 * x + 0.4 * np.abs(x), ref['util_lrelu_y'], rtol=1e-15)
  np.testing.assert_allclose(np.where(x > 0, x, 0.2 * x), ref['util_lrelu_y'], rtol=1e-15, atol=0)  # the kernels' form: 1 ulp (0.6 x + 0.4 x vs x)
  assert xutil.STATE_REWARD_DIM == 0 and xutil.STATE_STOPPED_DIM == 1 and xutil.STATE_STEP_DIM == 2
  pdf, noise, want = ref['pdf_sample_pdf'], ref['pdf_sample_noise'], ref['pdf_sample_ids']
  assert want[0] == -1 and want[1] == 7 and want.dtype == np.int32
  assert np.array_equal(agent_np.pdf_sample(pdf, noise), want)
  got = xagent.pdf_sample(torch.from_numpy(pdf), torch.from_numpy(noise))
  assert got.dtype == torch.int32 and np.array_equal(got.numpy(), want)


@pytest.mark.parametrize('tag,is_train', [('train', 1), ('eval', 0)])
def test_agent_arithmetic_composes_like_the_reference(ref, tag, is_train):
  """agent.py:100-123, 208-252 -- oracle/agent_np.py (float64): pdf, entropy, ids, one-hot, surrogate, states, penalty."""
  g = lambda k: ref['agent_%s_%s' % (tag, k)]
  logits, states, noise, net = ref['agent_logits'], ref['agent_states'], ref['agent_noise'], ref['agent_net']
  pdf, entropy, selected, one_hot, surrogate = agent_np.action_selection(logits, noise, is_train)
  np.testing.assert_allclose(pdf, g('pdf'), rtol=1e-13)
  np.testing.assert_allclose(entropy, g('entropy'), rtol=1e-12)
  assert np.array_equal(selected, g('selected_filter_id')) and np.array_equal(one_hot, g('filter_one_hot'))
  if is_train:
    assert selected[0] == -1 and one_hot[0].sum() == 0  # noise 0: nothing selected, an all-zero one-hot
  np.testing.assert_allclose(surrogate, g('surrogate'), rtol=1e-12, atol=1e-15)
  new_states, usage_penalty, is_last, submitted = agent_np.new_states(states, one_hot)
  assert np.array_equal(new_states, g('new_states')) and is_last.any() and not is_last.all()
  pen = agent_np.penalty(net, entropy, usage_penalty, is_last, submitted, float(ref['agent_progress']))
  np.testing.assert_allclose(pen, g('penalty'), rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('fid', range(9))
def test_hip_forward_kernels_compose_like_the_reference(ref, fid, gpu_device):
  """expo_filter_fwd on fp32 storage against the executed reference bodies: 1e-5 absolute + 1e-5 relative."""
  from exposure_amd import _cabi
  name = FILTERS[fid]
  want_p, want_y = ref[name + '_params'], ref[name + '_y']
  x = torch.from_numpy(ref['x'].astype(np.float32)).to(gpu_device)
  packed = torch.from_numpy(want_p.reshape(want_p.shape[0], -1).astype(np.float32)).to(gpu_device)
  y = torch.empty_like(x)
  _cabi.filter_fwd(fid, x, y, packed)
  # the oracle on the SAME fp32-rounded inputs separates the kernel's arithmetic from the input rounding
  same_in = fnp.process_packed(fid, x.cpu().numpy().astype(np.float64), packed.cpu().numpy().astype(np.float64))
  got = y.cpu().numpy().astype(np.float64)
  assert np.abs(got - same_in).max() <= 1e-5 + 1e-5 * np.abs(same_in).max(), name
  # and against the fixture itself: fp32 input rounding moves Gamma / Contrast near their kinks by a few 1e-6
  tol = 2e-5 + 2e-5 * np.abs(want_y)
  assert (np.abs(got - want_y) <= tol).all(), (name, np.abs(got - want_y).max())


@pytest.mark.gpu
@pytest.mark.parametrize('tag,is_train', [('train', 1), ('eval', 0)])
def test_hip_selection_kernel_composes_like_the_reference(ref, tag, is_train, gpu_device):
  """expo_agent_select_fwd + the over-exposure penalty kernel: ids, one-hot and new states BIT-EQUAL to the executed
  reference statements; pdf / entropy / surrogate / penalty within fp32 rounding."""
  from exposure_amd import _cabi
  dev = gpu_device
  g = lambda k: ref['agent_%s_%s' % (tag, k)]
  t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
  logits, states, noise = t32(ref['agent_logits']), t32(ref['agent_states']), t32(ref['agent_noise'])
  n, k = logits.shape
  # the ids depend on fp32 rounding of the cdf against the noise: the float64 fixture and the fp32 kernel agree wherever
  # the noise is not within 1e-6 of a cdf step -- true for this seeded draw (checked here, not assumed)
  pdf64 = g('pdf')
  cdf = np.concatenate([np.zeros((n, 1)), np.cumsum(pdf64 / pdf64.sum(1, keepdims=True), axis=1)[:, :-1]], axis=1)
  gap = np.abs(cdf - ref['agent_noise'])
  gap[(cdf == 0) & (ref['agent_noise'] == 0)] = 1.0  # (noise 0 against the leading 0 of the cdf: `0 < 0` is false in any precision)
  assert gap.min() > 1e-5
  progress = torch.tensor([float(ref['agent_progress'])], device=dev)
  pdf, onehot = torch.empty((n, k), device=dev), torch.empty((n, k), device=dev)
  entropy, surrogate, pen = (torch.empty((n, 1), device=dev) for _ in range(3))
  selected = torch.empty((n,), dtype=torch.int32, device=dev)
  new_states = torch.empty_like(states)
  consts = (0.05, 0.05, 1.0, 1.0, 5)  # exploration, exploration_penalty, filter_usage_penalty, early_stop_penalty, test_steps
  _cabi.agent_select_fwd(logits, noise, states, progress, consts, is_train, pdf, entropy, selected, onehot, surrogate,
                         new_states, pen)
  assert np.array_equal(selected.cpu().numpy(), g('selected_filter_id'))
  assert np.array_equal(onehot.cpu().numpy().astype(np.float64), g('filter_one_hot'))
  assert np.array_equal(new_states.cpu().numpy().astype(np.float64), g('new_states'))
  np.testing.assert_allclose(pdf.cpu().numpy(), pdf64, rtol=2e-6)
  np.testing.assert_allclose(entropy.cpu().numpy(), g('entropy'), rtol=1e-5)
  np.testing.assert_allclose(surrogate.cpu().numpy(), g('surrogate'), rtol=1e-5, atol=1e-6)
  net = t32(ref['agent_net'])
  over = torch.empty((n,), device=dev)
  _cabi.overexposure_penalty(net, over)
  total = over[:, None] + pen
  np.testing.assert_allclose(total.cpu().numpy(), g('penalty'), rtol=2e-5, atol=1e-6)
