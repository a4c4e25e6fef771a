"""-m gpu: the critic statistics (critics.py:48-73) and their first and second derivatives through
libexposure_hip.so against the float64 oracle (oracle/nets_np.py), and the double backward of the WGAN-GP
term (net.py:174-194) end to end -- the critic-weight gradients of c_loss with the HIP statistics in the
graph against the same graph with the torch-autograd restatement of the statistics (oracle/stats_torch.py)."""
import numpy as np
import pytest
import torch

from exposure_amd import _cabi, critics, synthetic
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from oracle import agent_np
from oracle import nets_np as nn_np
from tests._tol import assert_param_grad_close

pytestmark = pytest.mark.gpu
NP_DT = {torch.float16: np.float16, torch.float32: np.float32}


def stats_case(seed, shape, np_dt):
  """Images with every branch of the statistics' derivatives: values below 0 and above 1 (clip masks), exactly 0
  and 1 (inclusive bounds), grey pixels (three-way ties of max and min), two equal channels (two-way ties),
  pixels with max + min on both sides of 1 (the tf.minimum switch) and exactly on it."""
  rng = np.random.default_rng(seed)
  x = synthetic.make_images(rng, shape, np.float32) * 1.5 - 0.1
  flat = x.reshape(-1, 3)
  m = flat.shape[0]
  k = max(1, m // 16)
  flat[0:k] = flat[0:k, :1]                       # grey
  flat[k:2 * k, 1] = flat[k:2 * k, 0]             # two-way tie
  flat[2 * k:3 * k, 2] = 1.0                      # exactly on the upper clip bound
  flat[3 * k:4 * k, 0] = 0.0                      # exactly on the lower clip bound
  flat[4 * k:5 * k] = np.array([0.75, 0.5, 0.25])  # max + min == 2 - max - min exactly
  return x.astype(np_dt), rng.standard_normal((shape[0], 3)).astype(np.float32), \
      synthetic.make_grad(rng, shape, np_dt)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('shape', [(5, 64, 64, 3), (3, 7, 5, 3), (2, 33, 3, 3), (2, 256, 512, 3)])
def test_stats_derivative_kernels_match_oracle(dtype, shape, gpu_device):
  dev = gpu_device
  x, g, v = stats_case(11, shape, NP_DT[dtype])
  tx, tg, tv = (torch.from_numpy(a).to(dev) for a in (x, g, v))
  n = shape[0]
  stats = torch.empty((n, 3), device=dev)
  _cabi.critic_stats(tx, stats)
  x64, g64, v64 = x.astype(np.float64), g.astype(np.float64), v.astype(np.float64)
  ref_stats, cache = nn_np.stat_features(x64)
  np.testing.assert_allclose(stats.cpu().numpy(), ref_stats, rtol=2e-4, atol=2e-6)
  np.testing.assert_allclose(ref_stats, agent_np.critic_stats(x64), rtol=1e-12)  # the two oracles agree
  # fp16 STORAGE rounds every written value to 11 bits (and flushes below 6e-8): compare those in relative terms
  # against the magnitude of the image's gradient; fp32 storage is held to 2e-5
  hw = shape[1] * shape[2]

  def close(got, ref, what):
    got = got.float().cpu().numpy().astype(np.float64)
    scale = np.abs(ref).reshape(n, -1).max(axis=1).reshape(n, 1, 1, 1)
    tol = (2e-3 if dtype == torch.float16 else 3e-5) * scale + (1e-7 if dtype == torch.float16 else 1e-12)
    err = np.abs(got - ref)
    assert (err <= tol).all(), '%s: worst %.3e (tol %.3e)' % (what, err.max(), tol.flat[(err - tol).argmax() % tol.size])

  dx = torch.empty_like(tx)
  _cabi.critic_stats_bwd(tx, stats, tg, dx)
  close(dx, nn_np.stat_features_backward(cache, g64), 'stats_bwd')
  jv = torch.empty((n, 3), device=dev)
  _cabi.critic_stats_jvp(tx, stats, tv, jv)
  ref_jv = nn_np.stat_features_jvp(cache, v64)
  # sums of H*W*3 signed terms J_ke v_e: |err| <= 1e-4 |ref| + 2e-6 A, A = sum of their magnitudes (tests/_tol.py)
  assert_param_grad_close(jv.cpu().numpy(), ref_jv, nn_np.stat_features_jvp_abs(cache, v64),
                          'stats J v %s %s' % (NP_DT[dtype].__name__, 'x'.join(map(str, shape))))
  out = torch.empty_like(tx)
  _cabi.critic_stats_hvp(tx, tg, jv, tv, out)
  close(out, nn_np.stat_features_hvp(cache, g64, v64), 'stats_hvp')
  # bit-reproducible reduction
  jv2 = torch.empty_like(jv)
  _cabi.critic_stats_jvp(tx, stats, tv, jv2)
  assert torch.equal(jv, jv2)


def test_stat_features_autograd_first_and_second_order(gpu_device):
  """critics.stat_features under torch.autograd: grad, and grad of grad with respect to BOTH inputs of the first
  backward (the upstream gradient g -> the path to the critic's weights; the image x -> the second-order term)."""
  dev = gpu_device
  x, g, v = stats_case(5, (4, 64, 64, 3), np.float32)
  tx = torch.from_numpy(x).to(dev).requires_grad_(True)
  tg = torch.from_numpy(g).to(dev).requires_grad_(True)
  st = critics.stat_features(tx)
  dx, = torch.autograd.grad(st, tx, tg, create_graph=True)
  gx, gg = torch.autograd.grad(dx, [tx, tg], torch.from_numpy(v).to(dev))
  _, cache = nn_np.stat_features(x.astype(np.float64))
  ref_dx = nn_np.stat_features_backward(cache, g.astype(np.float64))
  ref_gg = nn_np.stat_features_jvp(cache, v.astype(np.float64))
  ref_gx = nn_np.stat_features_hvp(cache, g.astype(np.float64), v.astype(np.float64))
  assert np.abs(dx.detach().cpu().numpy() - ref_dx).max() <= 3e-5 * np.abs(ref_dx).max()
  assert_param_grad_close(gg.cpu().numpy(), ref_gg, nn_np.stat_features_jvp_abs(cache, v.astype(np.float64)), 'autograd J v')
  assert np.abs(gx.cpu().numpy() - ref_gx).max() <= 3e-5 * np.abs(ref_gx).max()
  # fp16 images enter as float32 (gradients of O(1/HW) would be subnormal in fp16)
  st16 = critics.stat_features(torch.from_numpy(x.astype(np.float16)).to(dev).requires_grad_(True))
  assert st16.dtype == torch.float32 and st16.grad_fn is not None


@pytest.mark.parametrize('value_net', [False, True])
def test_gradient_penalty_double_backward_matches_torch_autograd(gpu_device, value_net, monkeypatch):
  """c_loss = mean(fake - real) + 10 mean(max(||d D(x^)/d x^|| - 1, 0)^2) differentiated with respect to the critic's
  weights: HIP statistics (expo_critic_stats / _bwd / _jvp) in the graph vs the torch-autograd restatement of the
  statistics in the same graph.  Also the value network's input gradient (statistics + 11 state planes)."""
  from oracle import stats_torch
  from tests.test_oracle_nets import make_batch
  dev = gpu_device
  torch.manual_seed(4)
  gan = GAN(make_cfg(), device=dev)
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)  # gradient norm > 1: the one-sided penalty is active
    gan.value.fc2.weight.mul_(40.0)
  fake_input, real, states, z, masks, alpha = make_batch(8, 31)
  t = lambda a: torch.from_numpy(a).to(dev)

  def grads():
    if not value_net:
      out = gan.critic_losses(t(real), t(fake_input), t(alpha))
      params = list(gan.critic.parameters())
      g = torch.autograd.grad(out['c_loss'], params)
      return [float(out['c_loss'].detach()), float(out['gradient_penalty'])], g
    x = t(fake_input).requires_grad_(True)
    val = gan.value(x, t(states))
    gx, = torch.autograd.grad(val.sum(), x, create_graph=True)
    pen = (gx**2).sum()
    # (the last layer's bias does not reach the input gradient: skip parameters outside the graph)
    params = [p for p in gan.value.parameters() if p is not gan.value.fc2.bias]
    return [float(val.sum()), float(pen)], torch.autograd.grad(pen, params)

  vals_hip, g_hip = grads()
  monkeypatch.setattr(critics, 'stat_features', lambda im: stats_torch.stat_features(im.float()))
  vals_ref, g_ref = grads()
  assert vals_hip[1] > 1e-3
  for a, b in zip(vals_hip, vals_ref):
    assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (vals_hip, vals_ref)
  for p, (a, b) in enumerate(zip(g_hip, g_ref)):
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (p, float((a - b).abs().max()), scale)
