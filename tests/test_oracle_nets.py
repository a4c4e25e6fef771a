"""Pins oracle/nets_np.py (the float64 restatement of feature_extractor / critic / loss graph,
agent.py:11-37, critics.py:6-98, net.py:92-194): hand-written backward passes against float64 central
finite differences, TF SAME-padding arithmetic against an independent implementation (torch's
cross-correlation with explicit padding), and then the torch modules of exposure_amd -- holding the
SAME weights -- against the oracle (CPU: the C-ABI binding is mocked by the oracle's filter maths;
the -m gpu twin of the last part is tests/test_hip_nets.py)."""
from unittest import mock

import numpy as np
import pytest
import torch

from exposure_amd import checkpoint
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from oracle import nets_np as nn_np
from tests._fake_hip import fake_hip


def small_cfg():
  return dict(nn_np.DEFAULT_CFG, base_channels=4, source_img_size=16)


def random_critic_weights(rng, cfg, prefix, in_ch):
  w = {}
  ch, size, cin, i = cfg['base_channels'], cfg['source_img_size'] // 2, in_ch, 0
  while True:
    name = nn_np.conv_names(prefix, i + 1)[i]
    w[name + '/weights'] = rng.normal(size=(4, 4, cin, ch)) * 0.3
    w[name + '/biases'] = rng.normal(size=(ch,)) * 0.1
    cin, i = ch, i + 1
    if size <= 4:
      break
    ch, size = ch * 2, size // 2
  flat = 4 * 4 * cin
  w[prefix + 'fully_connected/weights'] = rng.normal(size=(flat, cfg['fc1_size'])) * 0.1
  w[prefix + 'fully_connected/biases'] = rng.normal(size=(cfg['fc1_size'],)) * 0.1
  w[prefix + 'fully_connected_1/weights'] = rng.normal(size=(cfg['fc1_size'], 1)) * 0.3
  w[prefix + 'fully_connected_1/biases'] = rng.normal(size=(1,)) * 0.1
  return w


@pytest.mark.parametrize('size,stride', [(8, 2), (7, 2), (5, 1), (64, 2)])
def test_conv2d_same_matches_independent_cross_correlation(size, stride):
  rng = np.random.default_rng(size)
  n = 1 if size == 64 else 2
  x = rng.normal(size=(n, size, size + (0 if size == 64 else 1), 3))
  w = rng.normal(size=(4, 4, 3, 5))
  b = rng.normal(size=(5,))
  got = nn_np.conv2d_same(x, w, b, stride)
  # TF SAME: out = ceil(in/stride); pad_total = max((out-1)*stride + k - in, 0); extra pixel AFTER
  def pads(s):
    out = -(-s // stride)
    tot = max((out - 1) * stride + 4 - s, 0)
    return tot // 2, tot - tot // 2
  (pt, pb), (pl, pr) = pads(x.shape[1]), pads(x.shape[2])
  xt = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
  ref = torch.nn.functional.conv2d(xt, torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), stride=stride)
  ref = ref.permute(0, 2, 3, 1).numpy()
  assert got.shape == ref.shape == (n, -(-x.shape[1] // stride), -(-x.shape[2] // stride), 5)
  assert np.abs(got - ref).max() < 1e-12
  if size == 64:  # the networks' case: SAME for k=4, s=2 on even sizes is symmetric padding 1
    assert (pt, pb, pl, pr) == (1, 1, 1, 1)


def test_conv2d_input_grad_matches_finite_differences():
  rng = np.random.default_rng(1)
  x = rng.normal(size=(2, 7, 8, 3))
  w = rng.normal(size=(4, 4, 3, 4))
  b = rng.normal(size=(4,))
  dy = rng.normal(size=nn_np.conv2d_same(x, w, b).shape)
  got = nn_np.conv2d_same_input_grad(dy, w, x.shape)
  f = lambda xx: float((nn_np.conv2d_same(xx, w, b) * dy).sum())
  for idx in [(0, 0, 0, 0), (1, 6, 7, 2), (0, 3, 4, 1), (1, 0, 7, 0)]:
    e = np.zeros_like(x)
    e[idx] = 1e-5
    fd = (f(x + e) - f(x - e)) / 2e-5
    assert abs(fd - got[idx]) < 1e-7 * max(1.0, abs(fd))


def test_lrelu_is_the_reference_formula_and_gradient():
  x = np.array([-2.0, -0.0, 0.0, 3.0])
  assert np.allclose(nn_np.lrelu(x), [-0.4, 0.0, 0.0, 3.0])
  assert np.allclose(nn_np.lrelu_grad(x), [0.2, 0.6, 0.6, 1.0])  # abs'(0) = 0 -> f1 at 0


def test_stat_features_backward_matches_finite_differences():
  rng = np.random.default_rng(2)
  img = rng.random((2, 6, 5, 3)) * 1.3 - 0.1  # some values outside [0,1]
  dstats = rng.normal(size=(2, 3))
  _, cache = nn_np.stat_features(img)
  got = nn_np.stat_features_backward(cache, dstats)
  f = lambda a: float((nn_np.stat_features(a)[0] * dstats).sum())
  for idx in [(0, 0, 0, 0), (1, 5, 4, 2), (0, 2, 3, 1), (1, 1, 1, 0), (0, 4, 0, 2)]:
    e = np.zeros_like(img)
    e[idx] = 1e-6
    fd = (f(img + e) - f(img - e)) / 2e-6
    assert abs(fd - got[idx]) < 1e-6 * max(1.0, abs(fd)), idx


def _stats_case(seed=5):
  rng = np.random.default_rng(seed)
  img = rng.random((2, 6, 5, 3)) * 1.5 - 0.2  # values below 0 and above 1 (clip masks)
  return img, rng.normal(size=(2, 3)), rng.normal(size=img.shape)


def test_stat_features_jvp_matches_finite_differences():
  """J v (the path of the gradient penalty's double backward to the critic's weights) two ways: central
  differences of the statistics along v, and the adjoint identity <J^T g, v> == <g, J v>."""
  img, dstats, v = _stats_case()
  _, cache = nn_np.stat_features(img)
  jv = nn_np.stat_features_jvp(cache, v)
  fd = (nn_np.stat_features(img + 1e-6 * v)[0] - nn_np.stat_features(img - 1e-6 * v)[0]) / 2e-6
  assert np.abs(fd - jv).max() < 1e-8
  back = nn_np.stat_features_backward(cache, dstats)
  assert np.abs((back * v).sum(axis=(1, 2, 3)) - (jv * dstats).sum(axis=1)).max() < 1e-14


def test_stat_features_hvp_matches_finite_differences():
  img, dstats, v = _stats_case(6)
  _, cache = nn_np.stat_features(img)
  got = nn_np.stat_features_hvp(cache, dstats, v)
  f = lambda a: float((nn_np.stat_features_backward(nn_np.stat_features(a)[1], dstats) * v).sum())
  for idx in np.ndindex(*img.shape):
    e = np.zeros_like(img)
    e[idx] = 1e-6
    fd = (f(img + e) - f(img - e)) / 2e-6
    assert abs(fd - got[idx]) < 1e-7 * max(1.0, abs(fd)), idx
  assert np.abs(got).max() > 1e-2


def test_torch_stats_restatement_agrees_with_numpy_incl_second_order():
  """oracle/stats_torch.py (autograd) vs oracle/nets_np.py (hand-written): values, J^T g, and the double backward."""
  from oracle import stats_torch
  img, dstats, v = _stats_case(7)
  x = torch.tensor(img, requires_grad=True)
  g = torch.tensor(dstats, requires_grad=True)
  st = stats_torch.stat_features(x)
  ref, cache = nn_np.stat_features(img)
  assert np.abs(st.detach().numpy() - ref).max() < 1e-14
  dx, = torch.autograd.grad(st, x, g, create_graph=True)
  assert np.abs(dx.detach().numpy() - nn_np.stat_features_backward(cache, dstats)).max() < 1e-14
  gx, gg = torch.autograd.grad(dx, [x, g], torch.tensor(v))
  assert np.abs(gg.numpy() - nn_np.stat_features_jvp(cache, v)).max() < 1e-13
  assert np.abs(gx.numpy() - nn_np.stat_features_hvp(cache, dstats, v)).max() < 1e-13


@pytest.mark.parametrize('with_states', [False, True])
def test_critic_input_grad_matches_finite_differences(with_states):
  cfg = small_cfg()
  rng = np.random.default_rng(3)
  nstate = 11 if with_states else 0
  prefix = 'rl_value/critic/' if with_states else 'critic/'
  w = random_critic_weights(rng, cfg, prefix, 3 + nstate + 3)
  img = rng.random((2, 16, 16, 3))
  states = rng.random((2, nstate)) if with_states else None
  out, cache = nn_np.critic_forward(img, cfg, w, prefix, states)
  assert out.shape == (2, 1)
  got = nn_np.critic_input_grad(cache, w)
  f = lambda a: float(nn_np.critic(a, cfg, w, prefix, states).sum())
  worst = 0.0
  for idx in [(0, 0, 0, 0), (1, 15, 15, 2), (0, 7, 8, 1), (1, 3, 12, 0), (0, 15, 0, 2), (1, 8, 8, 1)]:
    e = np.zeros_like(img)
    e[idx] = 1e-6
    fd = (f(img + e) - f(img - e)) / 2e-6
    worst = max(worst, abs(fd - got[idx]) / max(1e-3, abs(fd)))
  assert worst < 1e-5, worst


def test_gradient_penalty_is_one_sided_with_the_reference_epsilon():
  """net.py:186-189: sqrt(1e-6 + sum g^2), max(norm - 1, 0)^2, lambda = 10."""
  cfg = small_cfg()
  rng = np.random.default_rng(4)
  w = random_critic_weights(rng, cfg, 'critic/', 6)
  real, fake = rng.random((3, 16, 16, 3)), rng.random((3, 16, 16, 3))
  alpha = rng.random((3, 1, 1, 1))
  out = nn_np.critic_losses(real, fake, alpha, cfg, w)
  norm = np.sqrt(1e-6 + (out['gradients']**2).sum(axis=(1, 2, 3)))
  gp = 10 * np.mean(np.maximum(norm - 1, 0)**2)
  assert np.isclose(out['gradient_penalty'], gp) and np.isclose(out['gradient_norm'], norm.mean())
  assert np.isclose(out['c_loss'], -out['emd'] + gp)
  # zero weights in the last layer -> zero gradient -> norm = 1e-3 exactly, no penalty (one-sided)
  w0 = dict(w)
  w0['critic/fully_connected_1/weights'] = np.zeros_like(w['critic/fully_connected_1/weights'])
  out0 = nn_np.critic_losses(real, fake, alpha, cfg, w0)
  assert np.isclose(out0['gradient_norm'], 1e-3) and out0['gradient_penalty'] == 0.0


# ------------------------------------------------------------------ torch modules vs the oracle
def make_batch(n, seed, dtype=np.float32):
  rng = np.random.default_rng(seed)
  fake_input = (rng.random((n, 64, 64, 3))**2.2).astype(dtype)
  real = (rng.random((n, 64, 64, 3))).astype(dtype)
  states = np.zeros((n, 11), dtype=np.float32)
  states[:, 2] = rng.integers(0, 7, n)  # step (some == 4 -> last step; some > trajectory length after +1)
  states[-1, 2] = 7
  states[:, 3:] = (rng.random((n, 8)) < 0.3)
  z = rng.random((n, 131)).astype(np.float32)
  masks = [(rng.random((n, 4096)) < 0.5).astype(np.float32) for _ in range(2)]
  alpha = rng.random((n, 1, 1, 1)).astype(np.float32)
  return fake_input, real, states, z, masks, alpha


def spread_selection_noise(gan, dev, fake_input, z, states, masks, progress=0.3):
  """Selection noise placed in the middle of filter (i % 8)'s interval of image i's action pdf (the pdf does not depend
  on z): a batch of 8 then selects every filter once -- every head receives a gradient, both curve filters run."""
  t = lambda a: torch.from_numpy(a).to(dev)
  n = fake_input.shape[0]
  with torch.no_grad():
    pdf = gan.generator_losses(t(fake_input), t(z), t(states), progress, 1,
                               [t(m) for m in masks])['debug']['pdf_batch'].double().cpu().numpy()
  cum = np.concatenate([np.zeros((n, 1)), np.cumsum(pdf / pdf.sum(axis=1, keepdims=True), axis=1)], axis=1)
  want = np.arange(n) % 8
  z[:, 0] = (0.5 * (cum[np.arange(n), want] + cum[np.arange(n), want + 1])).astype(np.float32)
  return z


GRAD_TENSORS = {
    # (loss, TF variable names): generator -- both trunks' first and last convolution, EVERY filter's output head (the
    # path the HIP kernels' parameter gradients take into theta_g), one hidden FC, the selector heads; value net and critic
    'g_loss': ['generator/Conv/weights', 'generator/Conv_3/weights', 'generator/filter_0/fc1/weights'] +
              ['generator/filter_%d/fc2/weights' % j for j in range(8)] + ['generator/filter_7/fc2/biases',
              'generator/action_selection/Conv_3/weights', 'generator/action_selection/selector_fc1/weights',
              'generator/action_selection/selector_fc2/weights', 'generator/action_selection/selector_fc2/biases'],
    'v_loss': ['rl_value/critic/Conv/weights', 'rl_value/critic/fully_connected/weights',
               'rl_value/critic/fully_connected_1/biases'],
    'c_loss': None,  # every critic variable (the double backward of the gradient penalty reaches all of them)
}


def weight_gradient_check(gan, dev, cfg, weights, batch, progress, rel, tensors=None):
  """WEIGHT gradients of g_loss (theta_g), v_loss (theta_v) and c_loss (theta_c) -- what an optimizer step consumes --
  against the float64 oracle.  The oracle has no backward through the networks; its forward is differentiated
  numerically instead: for a variable W and a direction D, (L(W + h D) - L(W - h D)) / 2h in float64 (stop_gradient
  operands frozen at the base point) must equal <dL/dW, D> with dL/dW from the product's autograd.  Two directions per
  variable: the product's own gradient (first-order sensitive to a wrong scale, sign or missing term; bound `rel`) and a
  seeded random one (first-order sensitive, in expectation, to an error in any direction; bound 20 x `rel`).

  Why the random direction is looser: a batch of 8 drives ~3 million lrelu units, and about one of them has a
  pre-activation within fp32 rounding (1e-7) of ZERO -- the float64 oracle and the fp32 product then sit on different
  sides of that kink and take different, equally valid sub-gradients (measured, r04: unit (21, 5, 9) of generator/Conv for
  one image, -9.8e-8 in float64 vs +9.3e-9 in fp32, moved <g, D> by 0.3 % of its typical size while the product's GPU and
  CPU gradients agreed to 5e-6 elementwise and every other image matched the oracle to 1e-5).  Along the gradient itself
  such a unit enters at second order."""
  fake_input, real, states, z, masks, alpha = batch
  t = lambda a: torch.from_numpy(a).to(dev)
  d = lambda a: a.astype(np.float64)
  name_map = {name: (p, kind) for name, p, kind in checkpoint.tf_name_map(gan)}
  out = gan.generator_losses(t(fake_input), t(z), t(states), progress, 1, [t(m) for m in masks])
  c = gan.critic_losses(t(real), t(fake_input), t(alpha))
  base = nn_np.generator_losses(d(fake_input), d(z), d(states), progress, cfg, weights, [d(m) for m in masks], 1)
  frozen = dict(q_value=base['q_value'], weight=base['weight'])

  def oracle_loss(key, w):
    if key == 'c_loss':
      return nn_np.critic_losses(d(real), d(fake_input), d(alpha), cfg, w)['c_loss']
    return nn_np.generator_losses(d(fake_input), d(z), d(states), progress, cfg, w, [d(m) for m in masks], 1,
                                  frozen=frozen)[key]

  rng = np.random.default_rng(99)
  report = {}
  for key, loss in (('g_loss', out['g_loss']), ('v_loss', out['v_loss']), ('c_loss', c['c_loss'])):
    if tensors is not None and key not in tensors:
      continue
    names = (tensors or GRAD_TENSORS)[key] or [nm for nm in name_map if nm.startswith('critic/')]
    grads = torch.autograd.grad(loss, [name_map[nm][0] for nm in names], retain_graph=True, allow_unused=True)
    for nm, g in zip(names, grads):
      assert g is not None, '%s does not reach %s' % (key, nm)
      g = checkpoint.to_tf_layout(g, name_map[nm][1]).astype(np.float64)
      w0 = weights[nm]
      s = max(float(np.abs(w0).std()), 0.02)  # step in units of the variable's own scale
      gnorm = float(np.sqrt((g**2).sum()))
      if gnorm == 0:  # a filter head no image of this batch selected: the oracle's loss must not move either
        direction = rng.standard_normal(g.shape)
        h = 1e-4 * s
        fd = (oracle_loss(key, dict(weights, **{nm: w0 + h * direction})) -
              oracle_loss(key, dict(weights, **{nm: w0 - h * direction}))) / (2 * h)
        assert abs(fd) <= 1e-9, '%s: all-zero gradient of %s but the oracle moves (%.3e)' % (nm, key, fd)
        report[(key, nm, 'zero')] = 0.0
        continue
      for tag, direction in (('own', g / gnorm * np.sqrt(g.size)), ('random', rng.standard_normal(g.shape))):
        got = float((g * direction).sum())
        typical = gnorm * float(np.sqrt((direction**2).sum())) / np.sqrt(g.size)  # E|<g, random D>|
        # the losses are piecewise smooth (lrelu, clips, the one-sided penalty): a step that carries some unit across
        # a kink pollutes the difference quotient at the 1e-3 level -- relatively more along a random direction, whose
        # derivative is a sum of ~sqrt(#units) random-sign unit contributions -- so the step is shrunk (at most three
        # times) before a mismatch counts; float64 leaves ~1e-7 of relative noise at the smallest step
        for h in (1e-6 * s, 2.5e-7 * s, 6e-8 * s, 1.5e-8 * s):
          lp = oracle_loss(key, dict(weights, **{nm: w0 + h * direction}))
          lm = oracle_loss(key, dict(weights, **{nm: w0 - h * direction}))
          fd = (lp - lm) / (2 * h)
          scale = max(abs(fd), 0.1 * typical) if tag == 'own' else max(abs(fd), typical)
          bound = rel if tag == 'own' else 20 * rel
          if abs(got - fd) <= bound * scale:
            break
        report[(key, nm, tag)] = abs(got - fd) / scale
        assert abs(got - fd) <= bound * scale, ('%s wrt %s along %s: autograd %.8g vs oracle finite difference %.8g' %
                                                (key, nm, tag, got, fd))
  return report


# a short list for the other loss branches (the full one runs on the shipped configuration)
GRAD_TENSORS_SHORT = {
    'g_loss': ['generator/Conv/weights', 'generator/filter_1/fc2/weights', 'generator/filter_3/fc2/weights',
               'generator/action_selection/selector_fc2/weights'],
    'v_loss': ['rl_value/critic/fully_connected/weights'],
    'c_loss': ['critic/Conv/weights', 'critic/fully_connected/weights'],
}


def compare_gan_with_oracle(gan, dev, n=4, seed=11, rel=1e-4, grad_rel=1e-3, grad_tensors=None):
  """Shared by the CPU (mocked C-ABI) and GPU (HIP library) tests.  The loss branches follow gan.cfg (cfg.gan,
  cfg.use_TD, cfg.gradient_penalty_lambda)."""
  cfg = dict(nn_np.DEFAULT_CFG, gan=gan.cfg.gan, use_TD=gan.cfg.use_TD,
             gradient_penalty_lambda=gan.cfg.gradient_penalty_lambda)
  weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
  fake_input, real, states, z, masks, alpha = make_batch(n, seed)
  t = lambda a: torch.from_numpy(a).to(dev)
  d = lambda a: a.astype(np.float64)
  progress = 0.3
  z = spread_selection_noise(gan, dev, fake_input, z, states, masks, progress)
  # --- features / logits (rows a-12, a-13)
  ag = gan.generator
  from exposure_amd.util import enrich_image_input
  enriched = enrich_image_input(gan.cfg, t(fake_input), t(states))
  feats = ag.filter_features(enriched, t(masks[0])).detach().cpu().numpy()
  o_feats = nn_np.feature_extractor(nn_np.enrich_image_input(cfg, d(fake_input), d(states)), 4096, cfg, weights,
                                    'generator/', d(masks[0]))
  scale = np.abs(o_feats).max()
  assert np.abs(feats - o_feats).max() <= rel * scale, ('feature_extractor', np.abs(feats - o_feats).max(), scale)
  # --- critic / value logits (row a-14)
  logit = gan.critic(t(fake_input)).detach().cpu().numpy()
  o_logit = nn_np.critic(d(fake_input), cfg, weights, 'critic/')
  assert np.abs(logit - o_logit).max() <= rel * max(1.0, np.abs(o_logit).max())
  value = gan.value(t(fake_input), t(states)).detach().cpu().numpy()
  o_value = nn_np.critic(d(fake_input), cfg, weights, 'rl_value/critic/', states=d(states))
  assert np.abs(value - o_value).max() <= rel * max(1.0, np.abs(o_value).max())
  # --- generator / value losses (row a-15)
  out = gan.generator_losses(t(fake_input), t(z), t(states), progress, 1, [t(m) for m in masks])
  ref = nn_np.generator_losses(d(fake_input), d(z), d(states), progress, cfg, weights, [d(m) for m in masks], 1)
  ids = out['debug']['selected_filter_ids'].cpu().numpy()
  assert ids.dtype == np.int32 and np.array_equal(ids, ref['debug']['selected_filter_id'])
  assert np.array_equal(out['new_states'].detach().cpu().numpy(), ref['new_states'])
  img_tol = 1e-3 + 1e-3 * np.abs(ref['fake_output'])
  assert (np.abs(out['fake_output'].detach().float().cpu().numpy() - ref['fake_output']) <= img_tol).all()
  for key in ('reward', 'q_value', 'fake_logit'):
    got = out[key].detach().cpu().numpy()
    assert np.abs(got - ref[key]).max() <= rel * max(1.0, np.abs(ref[key]).max()), key
  for key in ('g_loss', 'v_loss'):
    got = float(out[key].detach())
    assert abs(got - ref[key]) <= rel * max(1.0, abs(ref[key])), (key, got, ref[key])
  # --- critic loss with the gradient penalty (double backward in torch, hand backward in the oracle)
  c = gan.critic_losses(t(real), t(fake_input), t(alpha))
  rc = nn_np.critic_losses(d(real), d(fake_input), d(alpha), cfg, weights)
  for key in ('c_loss', 'emd', 'gradient_norm', 'gradient_penalty', 'c_average'):
    got = float(c[key].detach())
    assert abs(got - rc[key]) <= rel * max(1.0, abs(rc[key])), (key, got, rc[key])
  # --- weight gradients of the three losses (what the optimizers consume)
  grad_report = weight_gradient_check(gan, dev, cfg, weights, (fake_input, real, states, z, masks, alpha), progress,
                                      grad_rel, grad_tensors)
  return dict(g_loss=float(out['g_loss'].detach()), c_loss=float(c['c_loss'].detach()),
              gradient_norm=float(c['gradient_norm'].detach()), grad_report=grad_report)


def test_weight_gradient_check_rejects_wrong_gradients():
  """The check itself must be able to fail: a generator whose filter step returns parameter gradients scaled by 0.97
  (a wrong `dparams` from the kernels -- exactly what the loss values cannot see) and a critic loss whose penalty
  term carries a 3 % wrong weight are both rejected."""
  from tests import _fake_hip
  torch.manual_seed(0)
  gan = GAN(make_cfg())
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)
  cfg = dict(nn_np.DEFAULT_CFG)
  weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
  batch = list(make_batch(8, 11))
  batch[3][:, 0] = ((np.arange(8) % 8) + 0.5) / 8.0
  honest = _fake_hip._dispatch_bwd

  def scaled_dparams(ids, x, dy, dx, params, dparams, dpenalty=None, hsv_grad_mode=0):
    honest(ids, x, dy, dx, params, dparams, dpenalty, hsv_grad_mode)
    dparams.mul_(0.97)

  with mock.patch.object(_fake_hip, '_dispatch_bwd', scaled_dparams), fake_hip():
    with pytest.raises(AssertionError, match='g_loss wrt generator/'):
      weight_gradient_check(gan, torch.device('cpu'), cfg, weights, batch, 0.3, 1e-3)
  # the critic: a penalty weight of 9.7 instead of the oracle's 10 (a 3 % error in the double-backward term)
  with mock.patch.object(gan.cfg, 'gradient_penalty_lambda', 9.7), fake_hip():
    with pytest.raises(AssertionError, match='c_loss wrt critic/'):
      weight_gradient_check(gan, torch.device('cpu'), cfg, weights, batch, 0.3, 1e-3, {'c_loss': None})


def test_torch_nets_and_losses_match_oracle_cpu():
  torch.manual_seed(0)
  cfg = make_cfg()
  gan = GAN(cfg)
  with torch.no_grad():  # non-trivial biases and a gradient norm above 1 somewhere (exercise the penalty)
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)
  with fake_hip():
    res = compare_gan_with_oracle(gan, torch.device('cpu'), n=8)
  assert not any(k[2] == 'zero' and 'filter_' in k[1] for k in res['grad_report']), 'a filter head saw no gradient'
  assert max(v for k, v in res['grad_report'].items() if k[2] != 'random') <= 1e-3
  assert res['gradient_norm'] > 1e-3


@pytest.mark.parametrize('gan_kind,use_td,gp_lambda', [('ls', True, 10), ('ls', False, 10), ('w', False, 10), ('w', True, 0)])
def test_loss_branches_match_oracle_cpu(gan_kind, use_td, gp_lambda):
  """The configuration branches of net.py:100-199 beside the shipped one: LSGAN reward and discriminator loss
  (no gradient penalty; the reported norm is d fake_logit / d fake_output), the plain-reward policy gradient
  (use_TD = False), WGAN without the penalty term."""
  torch.manual_seed(2)
  cfg = make_cfg()
  cfg.gan, cfg.use_TD, cfg.gradient_penalty_lambda = gan_kind, use_td, gp_lambda
  gan = GAN(cfg)
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)
  with fake_hip():
    res = compare_gan_with_oracle(gan, torch.device('cpu'), grad_tensors=GRAD_TENSORS_SHORT)
    # the branch really is a different number than the shipped configuration's
    base = GAN(make_cfg())
    base.load_state_dict(gan.state_dict())
    ref = compare_gan_with_oracle(base, torch.device('cpu'), grad_tensors={})
  if gan_kind == 'ls' or not use_td:
    assert abs(res['g_loss'] - ref['g_loss']) > 1e-6
  if gan_kind == 'ls' or gp_lambda == 0:
    assert abs(res['c_loss'] - ref['c_loss']) > 1e-6


def test_weight_clipping_replaces_the_penalty_when_lambda_is_zero():
  """net.py:252-262: WGAN with gradient_penalty_lambda <= 0 clamps every critic variable to +-cfg.clamp_critic after
  each critic update; with the penalty (the shipped configuration) nothing is clamped."""
  torch.manual_seed(0)
  fake_input, real, _states, _z, _masks, alpha = make_batch(4, 3)
  for lam, clipped in ((0, True), (10, False)):
    cfg = make_cfg()
    cfg.gradient_penalty_lambda = lam
    gan = GAN(cfg)
    with torch.no_grad():
      gan.critic.fc1.bias.fill_(0.5)
    with fake_hip():
      out = gan.critic_step(torch.from_numpy(real), torch.from_numpy(fake_input), it=1, alpha=torch.from_numpy(alpha))
    worst = max(float(p.detach().abs().max()) for p in gan.critic.parameters())
    assert (worst <= cfg.clamp_critic + 1e-12) == clipped, (lam, worst)
    assert float(out['gradient_penalty']) == 0.0 if lam == 0 else True
    assert float(gan.value.fc1.weight.detach().abs().max()) > cfg.clamp_critic  # only theta_c is clipped


def test_supervised_configuration_is_refused_and_noise_types():
  from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
  cfg = make_cfg()
  cfg.supervised = True
  gan = GAN(cfg)
  fake_input, _real, states, z, masks, _alpha = make_batch(2, 1)
  with fake_hip(), pytest.raises(NotImplementedError, match='supervised'):
    gan.generator_losses(torch.from_numpy(fake_input), torch.from_numpy(z), torch.from_numpy(states), 0.5, 1,
                         [torch.from_numpy(m) for m in masks])
  for z_type, check in (('uniform', lambda t: 0.0 <= float(t.min()) and float(t.max()) <= 1.0),
                        ('normal', lambda t: float(t.min()) < -1.0 and abs(float(t.mean())) < 0.1)):
    cfg = make_cfg()
    cfg.z_type = z_type
    dev = torch.device('cpu')
    mem = ReplayMemory(cfg, SyntheticProvider(dev, seed=1), SyntheticProvider(dev, seed=2), seed=0)
    noise = mem.get_noise(64)
    assert noise.shape == (64, cfg.z_dim) and check(noise), z_type
  cfg.z_type = 'cauchy'
  with pytest.raises(AssertionError, match='Unknown noise type'):
    mem.get_noise(4)


def test_gradient_penalty_double_backward_wiring_cpu(monkeypatch):
  """The autograd wiring of the HIP statistics (exposure_amd/critics.py: _CriticStats -> _CriticStatsGrad, with
  the C-ABI calls mocked by the NumPy oracle): the critic-weight gradients of c_loss -- a double backward through
  the statistics planes -- must equal those of the same graph with torch-autograd statistics."""
  from exposure_amd import critics
  from oracle import stats_torch
  torch.manual_seed(1)
  gan = GAN(make_cfg())
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)
  fake_input, real, states, z, masks, alpha = make_batch(3, 5)
  t = torch.from_numpy

  def grads():
    out = gan.critic_losses(t(real), t(fake_input), t(alpha))
    return float(out['gradient_penalty']), torch.autograd.grad(out['c_loss'], list(gan.critic.parameters()))

  with fake_hip():
    pen_a, g_a = grads()
  monkeypatch.setattr(critics, 'stat_features', lambda im: stats_torch.stat_features(im.float()))
  with fake_hip():  # (the activation kernels stay mocked; only the statistics switch to torch autograd)
    pen_b, g_b = grads()
  assert pen_a > 1e-3 and abs(pen_a - pen_b) <= 1e-4 * pen_b
  for a, b in zip(g_a, g_b):
    assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-8


def test_skipping_unneeded_parameter_gradients_changes_nothing(monkeypatch):
  """nn_ops.skip_parameter_gradients (the gradient penalty's inner autograd.grad) and nn_ops.frozen_parameters (the
  generator step's passes through the critic / value net) only drop work whose results autograd discards: every
  gradient a step keeps must be bit-identical with and without them."""
  import contextlib
  from exposure_amd import gan as xgan
  fake_input, real, states, z, masks, alpha = make_batch(3, 9)
  t = torch.from_numpy

  def run():
    torch.manual_seed(5)
    gan = GAN(make_cfg())
    with torch.no_grad():
      gan.critic.fc2.weight.mul_(40.0)
    with fake_hip():
      c = gan.critic_losses(t(real), t(fake_input), t(alpha))
      gc = torch.autograd.grad(c['c_loss'], list(gan.critic.parameters()))
      g = gan.generator_losses(t(fake_input), t(z), t(states), 0.3, 1, [t(m) for m in masks])
      gg = torch.autograd.grad(g['g_loss'], list(gan.generator.parameters()), retain_graph=True, allow_unused=True)
      gv = torch.autograd.grad(g['v_loss'], list(gan.value.parameters()), allow_unused=True)
    return [x for x in list(gc) + list(gg) + list(gv) if x is not None]

  with_skips = run()
  monkeypatch.setattr(xgan, 'skip_parameter_gradients', contextlib.nullcontext)
  monkeypatch.setattr(xgan, 'frozen_parameters', contextlib.nullcontext)
  without = run()
  assert len(with_skips) == len(without) > 30
  for a, b in zip(with_skips, without):
    assert torch.equal(a, b)
