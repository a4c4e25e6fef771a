"""GPU parity of the fused per-image dispatch (one-hot select + over-exposure penalty), the
per-image reductions, and the agent / GAN step running on the real HIP library."""
import numpy as np
import pytest
import torch

from exposure_amd import _cabi, agent as xagent, critics, filters, synthetic
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from oracle import agent_np
from oracle import filters_np as fnp
from tests._tol import assert_image_close, assert_param_grad_close

pytestmark = pytest.mark.gpu
NP_DT = {torch.float16: np.float16, torch.float32: np.float32}


def dispatch_case(seed, shape, np_dt):
  rng = np.random.default_rng(seed)
  n = shape[0]
  x = synthetic.make_images(rng, shape, np_dt)
  x *= np_dt(1.6)  # more over-exposed pixels so the penalty is exercised
  dy = synthetic.make_grad(rng, shape, np_dt)
  ids = rng.integers(-1, 8, n).astype(np.int32)
  ids[:9] = np.arange(-1, 8)[:min(9, n)]  # every branch at least once
  p24 = np.zeros((n, 24), dtype=np.float32)
  for i, fid in enumerate(ids):
    if fid >= 0:
      p24[i, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, int(fid), 1)[0]
  dpen = rng.standard_normal(n).astype(np.float32) * 50.0
  return x, dy, ids, p24, dpen


def dispatch_oracle(x, dy, ids, p24, dpen):
  n = x.shape[0]
  x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
  y = np.zeros_like(x64)
  dx = np.zeros_like(x64)
  dp = np.zeros((n, 24))
  adp = np.zeros((n, 24))  # sums of absolute terms of dp (tests/_tol.py); all-zero rows (id -1) must come out exactly 0
  cnt = x.shape[1] * x.shape[2] * 3
  for i, fid in enumerate(ids):
    if fid < 0:
      continue
    p = p24[i:i + 1, :fnp.NUM_PARAMS[fid]].astype(np.float64)
    y[i:i + 1] = fnp.process_packed(int(fid), x64[i:i + 1], p)
    g = dy64[i:i + 1] + (2.0 * np.maximum(y[i:i + 1] - 1, 0) * dpen[i] / cnt if dpen is not None else 0.0)
    gx, gp = fnp.backward_packed(int(fid), x64[i:i + 1], p, g)
    dx[i:i + 1] = gx
    dp[i, :fnp.NUM_PARAMS[fid]] = gp[0]
    adp[i, :fnp.NUM_PARAMS[fid]] = fnp.param_grad_abs(int(fid), x64[i:i + 1], p, g)[0]
  return y, agent_np.overexposure_penalty(y), dx, dp, adp


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(12, 64, 64, 3), (10, 9, 7, 3)])
@pytest.mark.parametrize('with_pen', [True, False])
def test_dispatch_matches_oracle(dtype, shape, with_pen, gpu_device):
  dev = gpu_device
  x, dy, ids, p24, dpen = dispatch_case(17, shape, NP_DT[dtype])
  tx, tdy = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
  tid, tp = torch.from_numpy(ids).to(dev), torch.from_numpy(p24).to(dev)
  y = torch.empty_like(tx)
  pen = torch.full((shape[0],), -3.0, device=dev) if with_pen else None
  _cabi.dispatch_fwd(tid, tx, y, tp, pen)
  dx = torch.empty_like(tx)
  dp = torch.full_like(tp, 9.0)
  tdpen = torch.from_numpy(dpen).to(dev) if with_pen else None
  _cabi.dispatch_bwd(tid, tx, tdy, dx, tp, dp, tdpen)
  ry, rpen, rdx, rdp, adp = dispatch_oracle(x, dy, ids, p24, dpen if with_pen else None)
  assert_image_close(y.float().cpu().numpy(), ry, NP_DT[dtype], 'dispatch y')
  assert_image_close(dx.float().cpu().numpy(), rdx, NP_DT[dtype], 'dispatch dx')
  assert_param_grad_close(dp.cpu().numpy(), rdp, adp, 'dispatch dparams')
  if with_pen:
    np.testing.assert_allclose(pen.cpu().numpy(), rpen, rtol=2e-4, atol=1e-7)
  # id -1: y == 0, dx == 0, dparams == 0
  sel = ids < 0
  assert sel.any()
  assert float(y[torch.from_numpy(sel).to(dev)].abs().max()) == 0.0
  assert float(dp[torch.from_numpy(sel).to(dev)].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(6, 64, 64, 3), (5, 9, 7, 3)])
def test_dispatch_penalty_of_curves_that_exceed_one(dtype, shape, gpu_device):
  """Tone / Color curves outside the reference's parameter ranges (falling tail -> T(x) > 1 in the middle):
  the fused penalty's gradient is live for those images and the backward must re-evaluate the forward
  (vector path: segment table; element-wise path: telescoped chain); a curve that stays below 1 beside
  them takes the skip."""
  dev = gpu_device
  x, dy, ids, p24, dpen = dispatch_case(31, shape, NP_DT[dtype])
  rng = np.random.default_rng(2)
  hump = np.array([2.0, 2.0, 2.0, 2.0, -0.5, -1.0, -1.0, -0.5], dtype=np.float32)  # T(1/2) = 8/3
  ids[:] = [4, 7, 7, 4, 7, 4][:shape[0]]
  p24[:] = 0
  p24[0, :8] = hump
  p24[1, :24] = np.concatenate([hump, synthetic.make_params(rng, 4, 1)[0], hump[::-1].copy() + 1.2])
  p24[2, :24] = synthetic.make_params(rng, 7, 1)[0]  # reference range: cannot exceed 1
  p24[3, :8] = synthetic.make_params(rng, 4, 1)[0]
  p24[4, :24] = np.tile(hump, 3) * np.float32(0.7)
  if shape[0] > 5:
    p24[5, :8] = hump * np.float32(3.0)
  tx, tdy = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
  tid, tp = torch.from_numpy(ids).to(dev), torch.from_numpy(p24).to(dev)
  y = torch.empty_like(tx)
  pen = torch.empty((shape[0],), device=dev)
  _cabi.dispatch_fwd(tid, tx, y, tp, pen)
  dx = torch.empty_like(tx)
  dp = torch.empty_like(tp)
  _cabi.dispatch_bwd(tid, tx, tdy, dx, tp, dp, torch.from_numpy(dpen).to(dev))
  ry, rpen, rdx, rdp, adp = dispatch_oracle(x, dy, ids, p24, dpen)
  assert rpen[0] > 1e-3 and rpen[1] > 1e-3 and rpen[4] > 1e-3, 'the case must exercise the live penalty'
  assert_image_close(y.float().cpu().numpy(), ry, NP_DT[dtype], 'dispatch y')
  assert_image_close(dx.float().cpu().numpy(), rdx, NP_DT[dtype], 'dispatch dx')
  np.testing.assert_allclose(pen.cpu().numpy(), rpen, rtol=2e-4, atol=1e-7)
  assert_param_grad_close(dp.cpu().numpy(), rdp, adp, 'dispatch dparams (live penalty)')


def test_dispatch_at_config5_size_every_pixel_against_the_c_oracle(gpu_device):
  """The agent's per-step kernels (one-hot select + fused over-exposure penalty, forward and backward) at
  16x512x512x3 fp16 -- streaming instantiations, light / curve launch pair -- with EVERY value of y and dx, the
  penalty and every parameter gradient checked against the float64 C restatement."""
  import os
  from oracle import filters_c as fc
  dev = gpu_device
  shape = synthetic.SHAPES['B']
  n, cnt = shape[0], shape[1] * shape[2] * 3
  try:
    ncpu = len(os.sched_getaffinity(0))
  except (AttributeError, OSError):
    ncpu = os.cpu_count() or 1
  fc.set_threads(max(1, min(64, ncpu // 2)), np.float64)
  rng = np.random.default_rng(91)
  x = synthetic.make_images(rng, shape, np.float16)
  x *= np.float16(1.6)
  dy = synthetic.make_grad(rng, shape, np.float16)
  ids = (np.arange(n) % 9 - 1).astype(np.int32)  # -1, 0..7, -1, 0..6
  p24 = np.zeros((n, 24), dtype=np.float32)
  for i, fid in enumerate(ids):
    if fid >= 0:
      p24[i, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, int(fid), 1)[0]
  dpen = (rng.standard_normal(n) * 50.0).astype(np.float32)
  tx, tdy = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
  tid, tp = torch.from_numpy(ids).to(dev), torch.from_numpy(p24).to(dev)
  y, dx = torch.empty_like(tx), torch.empty_like(tx)
  pen = torch.empty(n, device=dev)
  dp = torch.full_like(tp, float('nan'))
  _cabi.dispatch_fwd(tid, tx, y, tp, pen)
  _cabi.dispatch_bwd(tid, tx, tdy, dx, tp, dp, torch.from_numpy(dpen).to(dev))
  x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
  ry, rdx, rdp, adp = np.zeros_like(x64), np.zeros_like(x64), np.zeros((n, 24)), np.zeros((n, 24))
  for fid in range(8):
    sel = np.nonzero(ids == fid)[0]
    npar = fnp.NUM_PARAMS[fid]
    pp = p24[sel, :npar].astype(np.float64)
    ry[sel] = fc.process_packed(fid, x64[sel], pp)
    g = dy64[sel] + 2.0 * np.maximum(ry[sel] - 1, 0) * dpen[sel].astype(np.float64)[:, None, None, None] / cnt
    rdx[sel], rdp[sel, :npar], adp[sel, :npar] = fc.backward_packed(fid, x64[sel], pp, g, with_abs=True)
  rpen = np.mean(np.maximum(ry - 1, 0)**2, axis=(1, 2, 3))
  assert (rpen[ids >= 0] > 1e-4).any(), 'the case must exercise the penalty'
  np.testing.assert_allclose(pen.cpu().numpy(), rpen, rtol=2e-4, atol=1e-7)
  assert_image_close(y.cpu().numpy(), np.clip(ry, -65504.0, 65504.0), np.float16, 'dispatch y')
  assert_image_close(dx.cpu().numpy(), np.clip(rdx, -65504.0, 65504.0), np.float16, 'dispatch dx')
  assert_param_grad_close(dp.cpu().numpy(), rdp, adp, 'dispatch dparams (16x512x512)')


def test_dispatch_autograd_matches_per_filter(gpu_device):
  dev = gpu_device
  x, dy, ids, p24, _ = dispatch_case(23, (9, 32, 32, 3), np.float32)
  tx = torch.from_numpy(x).to(dev).requires_grad_(True)
  tp = torch.from_numpy(p24).to(dev).requires_grad_(True)
  y, pen = filters.dispatch_filters(tx, tp, torch.from_numpy(ids).to(dev))
  w = torch.linspace(-1, 1, 9, device=dev)
  ((y * torch.from_numpy(dy).to(dev)).sum() + (pen * w).sum()).backward()
  ry, rpen, rdx, rdp, adp = dispatch_oracle(x, dy, ids, p24, w.cpu().numpy())
  assert_image_close(tx.grad.cpu().numpy(), rdx, np.float32)
  assert_param_grad_close(tp.grad.cpu().numpy(), rdp, adp, 'dispatch autograd dparams')
  np.testing.assert_allclose(pen.detach().cpu().numpy(), rpen, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(5, 64, 64, 3), (3, 7, 5, 3), (2, 512, 512, 3)])
def test_critic_stats_and_penalty(dtype, shape, gpu_device):
  dev = gpu_device
  rng = np.random.default_rng(3)
  x = (synthetic.make_images(rng, shape, NP_DT[dtype]) * NP_DT[dtype](1.5))
  tx = torch.from_numpy(x).to(dev)
  stats = critics.critic_stats(tx)
  ref = agent_np.critic_stats(x.astype(np.float64))
  np.testing.assert_allclose(stats.cpu().numpy(), ref, rtol=2e-4, atol=2e-6)
  # the autograd entry the critic uses in training (same kernel, float32 image)
  np.testing.assert_allclose(critics.stat_features(tx).cpu().numpy(), ref, rtol=2e-4, atol=2e-6)
  pen = torch.empty(shape[0], device=dev)
  _cabi.overexposure_penalty(tx, pen)
  np.testing.assert_allclose(pen.cpu().numpy(), agent_np.overexposure_penalty(x.astype(np.float64)), rtol=2e-4,
                             atol=1e-8)


def test_agent_step_on_gpu_matches_oracle(gpu_device):
  dev = gpu_device
  torch.manual_seed(0)
  cfg = make_cfg()
  ag = xagent.Agent(cfg).to(dev)
  rng = np.random.default_rng(1)
  n = 16
  img = synthetic.make_images(rng, (n, 64, 64, 3), np.float16)
  states = np.zeros((n, 11), dtype=np.float32)
  states[:, 2] = rng.integers(0, 5, n)
  states[:, 3:] = rng.random((n, 8)) < 0.3
  z = rng.random((n, 131), dtype=np.float32)
  z[0, 0] = 0.0
  masks = [torch.from_numpy((rng.random((n, 4096)) < 0.5).astype(np.float32)).to(dev) for _ in range(2)]
  t = lambda a: torch.from_numpy(a).to(dev)
  (out, new_states, surrogate, penalty), dbg, _ = ag((t(img), t(z), t(states)), is_train=1, progress=0.3,
                                                    dropout_masks=masks)
  ids = dbg['selected_filter_ids'].cpu().numpy()
  pdf = dbg['pdf_batch'].detach().cpu().numpy().astype(np.float64)
  # integer outputs: bit-identical to the numpy restatement fed the same pdf and noise
  o_ids = agent_np.pdf_sample(pdf.astype(np.float32), z[:, 0:1])
  assert np.array_equal(ids, o_ids) and ids[0] == -1
  onehot = (ids[:, None] == np.arange(8)[None, :]).astype(np.float64)
  p24 = dbg['params24'].detach().cpu().numpy()
  params = [p24[:, :fnp.NUM_PARAMS[f]].astype(np.float64) for f in range(8)]
  ref_img = np.zeros((n, 64, 64, 3))
  for i in range(n):
    if ids[i] >= 0:
      ref_img[i:i + 1] = fnp.process_packed(int(ids[i]), img[i:i + 1].astype(np.float64), params[ids[i]][i:i + 1])
  assert out.dtype == torch.float16
  assert_image_close(out.detach().float().cpu().numpy(), ref_img, np.float16, 'agent out')
  o_states, o_usage, o_last, o_sub = agent_np.new_states(states.astype(np.float64), onehot)
  assert np.array_equal(new_states.cpu().numpy(), o_states.astype(np.float32))
  ent = -(pdf * np.log(pdf)).sum(axis=1, keepdims=True)
  # the fused penalty is computed on the fp32 filter output before the fp16 rounding
  o_pen = agent_np.penalty(ref_img, ent, o_usage, o_last, o_sub, 0.3)
  np.testing.assert_allclose(penalty.detach().cpu().numpy(), o_pen, rtol=1e-3, atol=1e-5)


def test_gan_steps_on_gpu(gpu_device):
  dev = gpu_device
  torch.manual_seed(0)
  cfg = make_cfg()
  gan = GAN(cfg, device=dev)
  rng = np.random.default_rng(2)
  n = cfg.batch_size
  t = lambda a: torch.from_numpy(a).to(dev)
  img = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  real = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  states = torch.zeros(n, 11, device=dev)
  z = t(rng.random((n, 131), dtype=np.float32))
  c0 = [p.detach().clone() for p in gan.critic.parameters()]
  g0 = [p.detach().clone() for p in gan.generator.parameters()]
  out = gan.generator_step(img, z, states, progress=0.1, it=3)
  assert torch.isfinite(out['g_loss']) and torch.isfinite(out['v_loss'])
  assert out['fake_output'].dtype == torch.float16 and out['fake_output'].shape == img.shape
  assert any(not torch.equal(a, b) for a, b in zip(g0, gan.generator.parameters()))
  assert all(torch.equal(a, b) for a, b in zip(c0, gan.critic.parameters()))
  out = gan.critic_step(real, out['fake_output'], it=3)
  assert torch.isfinite(out['c_loss']) and float(out['gradient_norm']) > 0
  assert any(not torch.equal(a, b) for a, b in zip(c0, gan.critic.parameters()))


def test_training_loop_on_gpu(gpu_device):
  """A few iterations of the reference's G/C alternation (net.py:307-365) with the device-resident
  replay memory: everything (filters, nets, pool) stays on the GPU."""
  from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
  dev = gpu_device
  torch.manual_seed(0)
  cfg = make_cfg()
  cfg.batch_size = 16
  cfg.replay_memory_size = 32
  cfg.max_iter_step = 3
  cfg.critic_initialization = 0
  cfg.citers = 2
  gan = GAN(cfg, device=dev)
  mem = ReplayMemory(cfg, SyntheticProvider(dev, dtype=torch.float16, seed=1),
                     SyntheticProvider(dev, gamma=1.0, dtype=torch.float16, seed=2), seed=0)

  class Few(type(gan)):
    pass

  # iteration 0 of GAN.train runs 100 generator steps; keep the test short by pre-rolling 6
  for _ in range(6):
    feed, feats = mem.get_feed_dict_and_states(cfg.batch_size)
    out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
    mem.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)
    # the host mirror of (stopped, step) -- derived from the agent's update rule, never read from the device -- equals
    # what the agent's kernels actually wrote
    assert mem.check_host_mirror()
  assert mem.images.is_cuda and mem.images.dtype == torch.float16
  assert int((mem.states[:, 1] > 0).sum()) > 0
  for it in range(1, 4):
    feed, feats = mem.get_feed_dict_and_states(cfg.batch_size)
    g = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], it / 3.0, it=it)
    mem.replace_memory(g['fake_output'], g['new_states'], feats, advanced=(it % 2 == 0))  # both paths
    assert mem.check_host_mirror()
    for _ in range(cfg.citers):
      rep = mem.get_replay_feed_dict(cfg.batch_size)
      c = gan.critic_step(rep['real_data'], rep['fake_output'], it=it)
    assert torch.isfinite(g['g_loss']) and torch.isfinite(c['c_loss'])
  assert len(mem) == 32


def test_iteration_graph_trains_bit_identically_to_the_step_calls(gpu_device):
  """``GAN.train_iteration``: the whole iteration planned ahead on the host and replayed as ONE hipGraph (pool gathers /
  scatters, G / V step, five critic steps, masks and alpha from the graph-registered generator, learning rates and
  progress from the plan record) against the same iterations through generator_step / replace_memory / critic_step:
  every reported value of every iteration, the weights of all three nets and the pools bit-equal."""
  from exposure_amd.replay_memory import ReplayMemory, ResidentProvider
  dev = gpu_device
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size, cfg.citers = 16, 48, 3
  runs = []
  for planned in (True, False):
    torch.manual_seed(0)
    gan = GAN(cfg, device=dev, use_graphs=True, seed=4)
    mem = ReplayMemory(cfg, ResidentProvider(dev, dtype=torch.float16, seed=1, count=256),
                       ResidentProvider(dev, gamma=1.0, dtype=torch.float16, seed=2, count=192), seed=0)
    for _ in range(7):  # iteration 0's roll-out (net.py:320-328), shortened: terminated records for the critic
      feed, feats = mem.get_feed_dict_and_states(cfg.batch_size, lazy=True)
      g = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
      mem.replace_memory(g['fake_output'], g['new_states'], feats, advanced=True)
    vals = []
    for it in range(1, 8):
      if planned:
        out = gan.train_iteration(mem, it)
      else:
        out = gan._iteration_stepwise(mem, it, float(it) / cfg.max_iter_step, cfg.batch_size)
      vals += [out['g'][k].clone().reshape(-1)[:1] for k in ('g_loss', 'v_loss')]
      vals += [out['c'][k].clone().reshape(-1)[:1] for k in ('c_loss', 'emd', 'gradient_norm', 'c_average')]
      assert mem.check_host_mirror()
    if planned:
      assert any(k[0] == 'it' and isinstance(v, tuple) for k, v in gan._graphs.items()), 'the iteration graph was never captured'
      assert gan._replay_steps and not any(k[0] == 'c' for k in gan._graphs), 'critic steps ran outside the iteration graph'
    assert gan.c_average_steps == 7 * cfg.citers
    runs.append((torch.cat(vals), mem.images, mem.states, mem.features, float(gan.c_average_biased)) +
                tuple(p.detach().clone() for p in gan.parameters()))
  assert bool(torch.isfinite(runs[0][0]).all())
  for i, (a, b) in enumerate(zip(*runs)):
    assert (a == b) if isinstance(a, float) else torch.equal(a, b), i


def test_lazy_feeds_train_bit_identically_on_gpu(gpu_device):
  """PoolRows gathered straight into the step graphs' inputs + views of the HBM-resident data sets (the training loops'
  feeds since round 6) against the gathered feeds of the same memories: every loss of every step bit-equal, eager and
  replayed from hipGraphs, and the pools end up identical."""
  from exposure_amd.replay_memory import ReplayMemory, ResidentProvider
  dev = gpu_device
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size, cfg.citers = 16, 32, 2
  for graphs in (False, True):
    runs = []
    for lazy in (True, False):
      torch.manual_seed(0)
      gan = GAN(cfg, device=dev, use_graphs=graphs, seed=4)
      mem = ReplayMemory(cfg, ResidentProvider(dev, dtype=torch.float16, seed=1, count=256),
                         ResidentProvider(dev, gamma=1.0, dtype=torch.float16, seed=2, count=256), seed=0)
      vals = []
      for it in range(9):
        feed, feats = mem.get_feed_dict_and_states(cfg.batch_size, lazy=lazy)
        g = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], it / 9.0, it=min(it, 1) * it)
        vals += [g['g_loss'].clone(), g['v_loss'].clone()]
        mem.replace_memory(g['fake_output'], g['new_states'], feats, advanced=True)
        if it >= 5:
          for _ in range(cfg.citers):
            rep = mem.get_replay_feed_dict(cfg.batch_size, lazy=lazy)
            c = gan.critic_step(rep['real_data'], rep['fake_output'], it=it)
            vals.append(c['c_loss'].clone())
      assert mem.check_host_mirror()
      runs.append((torch.stack([v.reshape(()) for v in vals]), mem.images, mem.states, mem.features))
    for a, b in zip(*runs):
      assert torch.equal(a, b), graphs
    assert bool(torch.isfinite(runs[0][0]).all())


def test_graphed_steps_match_eager(gpu_device):
  """hipGraph replay of the generator / critic step == the eager step (same inputs, masks, alpha)."""
  dev = gpu_device
  cfg = make_cfg()
  rng = np.random.default_rng(5)
  n = 16
  t = lambda a: torch.from_numpy(a).to(dev)
  img = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  real = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  states = torch.zeros(n, 11, device=dev)
  states[:, 2] = t(rng.integers(0, 4, n).astype(np.float32))
  z = t(rng.random((n, 131), dtype=np.float32))
  masks = [t((rng.random((n, 4096)) < 0.5).astype(np.float32)) for _ in range(2)]
  alpha = t(rng.random((n, 1, 1, 1), dtype=np.float32))
  results = []
  for use_graphs in (False, True):
    torch.manual_seed(11)
    gan = GAN(cfg, device=dev, use_graphs=use_graphs)
    for it in (3, 4, 5, 6):  # 1st call eager (warm-up), 2nd captures + replays, then pure replays
      g = gan.generator_step(img, z, states, progress=0.2, it=it, dropout_masks=masks)
      fake = g['fake_output'].clone()
      c = gan.critic_step(real, fake, it=it, alpha=alpha)
    results.append(([p.detach().clone() for p in gan.parameters()], float(g['g_loss']), float(c['c_loss'])))
  (pe, ge, ce), (pg, gg, cg) = results
  assert abs(ge - gg) <= 1e-3 * (1 + abs(ge)) and abs(ce - cg) <= 1e-3 * (1 + abs(ce))
  worst = max(float((a - b).abs().max()) for a, b in zip(pe, pg))
  assert worst < 3e-4, worst  # Adam steps are ~lr-sized; see tests/test_dist_gloo.py


def test_heads_are_packed_at_construction_and_a_repack_drops_the_captured_steps(gpu_device):
  """the filter heads' parameters live in the packed buffers from GAN construction on (optimisers, buckets
  and graphs are built over the final storage).  Parameters re-allocated behind captured step graphs -- here one head's
  weight replaced, as ``module.to(dtype)`` or a storage-replacing restore would -- are noticed at the next step: the
  heads are packed again, the captured steps dropped (warning) and captured afresh, and training goes on updating the
  parameters the module actually holds -- the same values as a run that never re-allocated."""
  import warnings
  dev = gpu_device
  cfg = make_cfg()
  rng = np.random.default_rng(9)
  n = 8
  t = lambda a: torch.from_numpy(a).to(dev)
  img = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  real = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  states = torch.zeros(n, 11, device=dev)
  z = t(rng.random((n, 131), dtype=np.float32))
  masks = [t((rng.random((n, 4096)) < 0.5).astype(np.float32)) for _ in range(2)]
  alpha = t(rng.random((n, 1, 1, 1), dtype=np.float32))
  finals = []
  for disturb in (False, True):
    torch.manual_seed(21)
    gan = GAN(cfg, device=dev, use_graphs=True)
    pack = gan.generator._packed_heads
    assert pack and pack is gan._heads_pack and pack.generation == 1 and pack.quick_aliased()
    head0 = gan.generator.filters[0].fc1.weight
    assert head0.data_ptr() == pack.w1.data_ptr()  # packed before the first forward
    for it in (3, 4, 5):  # eager warm-up, capture + replay, replay
      g = gan.generator_step(img, z, states, progress=0.2, it=it, dropout_masks=masks)
      gan.critic_step(real, g['fake_output'].clone(), it=it, alpha=alpha)
    assert any(e != 'warm' for e in gan._graphs.values())
    if disturb:
      head0.data = head0.data.clone()  # new storage, same values
      last = gan.generator.filters[-1].fc2.bias
      last.data = last.data.clone()
      assert not pack.quick_aliased()
      with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        g = gan.generator_step(img, z, states, progress=0.2, it=6, dropout_masks=masks)
      assert any('re-allocated' in str(w.message) for w in caught)
      assert pack.generation == 2 and pack.quick_aliased() and head0.data_ptr() == pack.w1.data_ptr()
      assert all(e == 'warm' for e in gan._graphs.values())  # every captured step dropped; this call ran eagerly
    else:
      g = gan.generator_step(img, z, states, progress=0.2, it=6, dropout_masks=masks)
    gan.critic_step(real, g['fake_output'].clone(), it=6, alpha=alpha)
    before = pack.w1.detach().clone()
    for it in (7, 8):  # (capture +) replay over the current storage
      g = gan.generator_step(img, z, states, progress=0.2, it=it, dropout_masks=masks)
      gan.critic_step(real, g['fake_output'].clone(), it=it, alpha=alpha)
    torch.cuda.synchronize()
    # Adam keeps updating the storage the module's parameters live in (the heads some image of the batch selected)
    assert head0.data_ptr() == pack.w1.data_ptr() and not torch.equal(before, pack.w1.detach())
    finals.append([p.detach().clone() for p in gan.parameters()])
  worst = max(float((a - b).abs().max()) for a, b in zip(*finals))
  assert worst < 3e-4, worst  # (eager vs replayed steps differ at Adam's step size at most: test_graphed_steps_match_eager)


def test_capture_survives_dead_graph_owners(gpu_device):
  """Round-5 incident: three DEAD GANs whose captured step graphs are only reachable through a reference cycle, then a
  live GAN captures its steps while the cyclic collector is as eager as it can be.  The product collects BEFORE the
  capture and keeps the collector off inside it (util.capture_without_gc), so the run finishes; without the guard the
  collector destroys the dead graphs inside the capture -- "operation not permitted when stream is capturing", thrown
  from a destructor -- and the process aborts (tools/r05/gc_capture_repro.py; the unguarded leg is run for the record:
  its outcome depends on the runtime and is reported, not asserted)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = os.path.join(root, 'tools', 'r05', 'gc_capture_repro.py')
  guarded = subprocess.run([sys.executable, script], cwd=root, capture_output=True, text=True, timeout=600)
  assert guarded.returncode == 0 and 'OK: captured' in guarded.stdout, guarded.stderr[-2000:]
  unguarded = subprocess.run([sys.executable, script, 'unguarded'], cwd=root, capture_output=True, text=True, timeout=600)
  print('unguarded run: rc=%d%s' % (unguarded.returncode, ' (' + unguarded.stderr.strip().splitlines()[1].strip() + ')'
                                     if unguarded.returncode and len(unguarded.stderr.strip().splitlines()) > 1 else ''))


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(4, 96, 128, 3), (3, 7, 9, 3)])
@pytest.mark.parametrize('steps', [8, 5, 1, 0])
def test_fused_chain_matches_oracle(dtype, shape, steps, gpu_device):
  """expo_chain_fused_fwd: per-image filter sequences applied in registers == the float64 chain.  The kernel runs
  two steps per loop trip (two pixel arrays swapping roles), so odd lengths -- the reference's own 5 steps -- end on a
  padding step; -1 (nothing selected) in the middle, at the start and at the end of a sequence; inf pixels through it."""
  dev = gpu_device
  rng = np.random.default_rng(31)
  n = shape[0]
  x = synthetic.make_images(rng, shape, NP_DT[dtype])
  ids = rng.integers(0, 9, (n, steps)).astype(np.int32)
  if steps == 8:
    ids[0] = np.arange(8)  # the cfg.filters order on image 0
    ids[1, 3] = -1  # an all-zero one-hot in the middle of image 1's sequence
  if steps >= 1:
    ids[2, 0] = -1
    ids[1, steps - 1] = -1
    x[2, 0, 0, :] = np.inf  # -1 must give exactly 0 whatever the pixel held
  p = np.zeros((n, steps, 24), dtype=np.float32)
  ref = x.astype(np.float64)
  for st in range(steps):
    nxt = np.zeros_like(ref)
    for i in range(n):
      fid = int(ids[i, st])
      if fid >= 0:
        p[i, st, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, fid, 1)[0]
        nxt[i:i + 1] = fnp.process_packed(fid, ref[i:i + 1], p[i:i + 1, st, :fnp.NUM_PARAMS[fid]].astype(np.float64))
    ref = nxt
  from exposure_amd import evaluate
  y = evaluate.fused_chain(torch.from_numpy(x).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(p).to(dev))
  assert_image_close(y.float().cpu().numpy(), ref, NP_DT[dtype], 'fused chain')
  if steps >= 1:
    assert float(y[1].abs().max()) == 0.0 and not bool(torch.signbit(y[1]).any())  # ends on -1: exactly +0
  if steps == 0:
    assert torch.equal(y, torch.from_numpy(x).to(dev))


def test_fused_chain_at_config5_size_every_pixel_against_the_c_oracle(gpu_device):
  """BASELINE config 5's kernel at 16x512x512x3 fp16: all 8 filters in registers, one read and one write; EVERY
  output value against the float64 C restatement applied step by step (no fp16 rounding between steps on either
  side).  Image 3 runs the sequence backwards."""
  import os
  from oracle import filters_c as fc
  dev = gpu_device
  shape = synthetic.SHAPES['B']
  n = shape[0]
  try:
    ncpu = len(os.sched_getaffinity(0))
  except (AttributeError, OSError):
    ncpu = os.cpu_count() or 1
  fc.set_threads(max(1, min(64, ncpu // 2)), np.float64)
  rng = np.random.default_rng(77)
  x = synthetic.make_images(rng, shape, np.float16)
  ids = np.tile(np.arange(8, dtype=np.int32), (n, 1))
  ids[3] = ids[3, ::-1]
  p = np.zeros((n, 8, 24), dtype=np.float32)
  for i in range(n):
    for st in range(8):
      fid = int(ids[i, st])
      p[i, st, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, fid, 1)[0]
  ref = x.astype(np.float64)
  for st in range(8):
    nxt = np.empty_like(ref)
    for fid in np.unique(ids[:, st]):
      sel = np.nonzero(ids[:, st] == fid)[0]
      nxt[sel] = fc.process_packed(int(fid), ref[sel], p[sel, st, :fnp.NUM_PARAMS[fid]].astype(np.float64))
    ref = nxt
  y = torch.empty(shape, dtype=torch.float16, device=dev)
  _cabi.chain_fused_fwd(torch.from_numpy(ids).to(dev), torch.from_numpy(p).to(dev), torch.from_numpy(x).to(dev), y)
  np.clip(ref, -65504.0, 65504.0, out=ref)
  assert_image_close(y.cpu().numpy(), ref, np.float16, 'fused chain at 16x512x512')


def test_retouch_fused_equals_stepwise_on_gpu(gpu_device):
  from exposure_amd import evaluate
  dev = gpu_device
  torch.manual_seed(3)
  cfg = make_cfg()
  ag = xagent.Agent(cfg).to(dev)
  hi = torch.from_numpy(synthetic.make_images(np.random.default_rng(8), (3, 200, 304, 3), np.float32)).to(dev)
  z = torch.rand(3, cfg.z_dim, device=dev)
  masks = [[(torch.rand(3, 4096, device=dev) < 0.5).float() for _ in range(2)] for _ in range(5)]
  a, la, sa, ta = evaluate.retouch(ag, hi, z=z, dropout_masks=masks, return_trace=True, fused=True)
  b, lb, sb, tb = evaluate.retouch(ag, hi, z=z, dropout_masks=masks, return_trace=True, fused=False)
  assert torch.equal(ta, tb) and torch.equal(sa, sb)
  assert_image_close(a.cpu().numpy(), b.cpu().numpy(), np.float32, 'fused vs stepwise')
  assert ta.shape == (3, 5) and sa[:, 2].tolist() == [5.0, 5.0, 5.0]


def test_train_cli_runs(gpu_device, capsys):
  from exposure_amd import train
  train.main(['--iters', '2', '--log-every', '1'])
  out = capsys.readouterr().out
  assert 'it     2' in out and 'Replay memory: size 128' in out


def test_training_step_graph_captures_rccl_collectives(gpu_device):
  """One rank under torchrun with EXPO_FORCE_COLLECTIVES=1: the gradient all-reduces go through RCCL even
  though the group has one member (launched from the buckets' backward hooks), and each optimisation step
  -- collectives included -- is captured into and replayed from one hipGraph.  Round 1 accepted the first
  clean run out of three because ~1 run in 15 aborted; the cause (ProcessGroupNCCL's watchdog polling an
  eager work's event while RCCL's stream was being captured -> hipErrorCapturedEvent) is fixed in
  GAN._replay, so a single run must pass."""
  import json
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, EXPO_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  env.pop('EXPO_GRAPH_COLLECTIVES', None)  # default 'auto': capture once the watchdog drain is verified
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
         '127.0.0.1', '--master-port', '29533', os.path.join(root, 'bench.py'), '--gpus', '1', '--workload', 'train',
         '--steps', '3', '--warmup', '2']
  last = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
  assert last.returncode == 0, last.stderr[-3000:]
  line = [l for l in last.stdout.splitlines() if l.startswith('{')][-1]
  d = json.loads(line)
  assert 'hipGraph' in d['config']['launch'], d['config']
  assert d['value'] > 0
  assert 'hipGraph capture of' not in last.stderr  # the eager fallback of GAN._replay was not taken
  # the drain in front of the capture was VERIFIED through the NCCL flight recorder (no timed grace period)
  assert d['config']['capture_drain_verified'] is True, d['config']


@pytest.mark.parametrize('workload', ['chain', 'chain_fused', 'infer'])
def test_bench_workloads_under_the_driver_launch_line(workload, gpu_device):
  """The driver's multi-GPU launch line (`python -m torch.distributed.run ... bench.py --gpus N ...`) with one rank:
  the process group is RCCL, the timed region is bracketed by the device-side barrier, the MAX over ranks is an
  all-reduce -- the code a > 1-GPU run executes, on the one GPU there is -- for the workloads that have no gradient
  exchange (the train workload: test_training_step_graph_captures_rccl_collectives)."""
  import json
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
         '127.0.0.1', '--master-port', str(29541 + ['chain', 'chain_fused', 'infer'].index(workload)),
         os.path.join(root, 'bench.py'), '--gpus', '1', '--workload', workload,
         '--steps', '5', '--warmup', '2', '--shape', 'B']
  if workload == 'chain':
    cmd += ['--no-cpu-baseline', '--cold-shape', 'none']
  out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 1 and d['steps'] == 5 and d['warmup'] == 2 and d['value'] > 0 and d['unit'] == 'Mpixels/s'
  assert d['scaling'] == 'weak' and d['data'] == 'synthetic' and 'roofline' in d
  if workload == 'chain':  # the default line's extra legs under a (1-rank) RCCL process group
    assert 'error' not in d['legs'] and 'error' not in d['legs']['train'] and 'allreduce' not in d['legs'], d['legs']
    assert d['legs']['train']['roofline']['flops_per_iteration'] > 1e11


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(12, 64, 64, 3), (10, 9, 7, 3)])
def test_masked_dispatch_equals_per_filter_masked_apply(dtype, shape, gpu_device):
  """expo_filter_apply_dispatch_fwd/bwd (the agent's step with cfg.masking = True: per image only the SELECTED filter's
  mask + process + lerp) against the per-filter entry points image by image -- same kernel bodies, so bit-identical
  images and parameter gradients -- and against the float64 restatement; id -1 gives zeros everywhere."""
  from oracle import filters_torch as ft
  dev = gpu_device
  x, dy, ids, p24, _ = dispatch_case(23, shape, NP_DT[dtype])
  ids[-1] = 8  # LevelFilter rides along (dispatch_case draws ids from -1..7)
  rng = np.random.default_rng(5)
  p24[-1] = 0
  p24[-1, :2] = synthetic.make_params(rng, 8, 1)[0]
  n = shape[0]
  mp = (np.tanh(rng.standard_normal((n, 6))) * 5).astype(np.float32)
  t = lambda a: torch.from_numpy(a).to(dev)
  tx, tdy, tid, tp, tmp = t(x), t(dy), t(ids), t(p24), t(mp)
  y = torch.empty_like(tx)
  _cabi.apply_dispatch_fwd(tid, tx, y, tp, tmp, 1.0, 0.3)
  dx = torch.empty_like(tx)
  dp = torch.full_like(tp, float('nan'))
  dm = torch.full_like(tmp, float('nan'))
  _cabi.apply_dispatch_bwd(tid, tx, tdy, dx, tp, dp, tmp, dm, 1.0, 0.3)
  for i in range(n):
    fid = int(ids[i])
    if fid < 0:
      assert float(y[i].float().abs().max()) == 0.0 and float(dx[i].float().abs().max()) == 0.0
      assert float(dp[i].abs().max()) == 0.0 and float(dm[i].abs().max()) == 0.0
      continue
    P = fnp.NUM_PARAMS[fid]
    xi, gi = tx[i:i + 1].contiguous(), tdy[i:i + 1].contiguous()
    pi, mi = tp[i:i + 1, :P].contiguous(), tmp[i:i + 1].contiguous()
    yi, dxi = torch.empty_like(xi), torch.empty_like(xi)
    dpi, dmi = torch.empty_like(pi), torch.empty_like(mi)
    _cabi.apply_fwd(fid, xi, yi, pi, mi, 1.0, 0.3)
    _cabi.apply_bwd(fid, xi, gi, dxi, pi, dpi, mi, dmi, 1.0, 0.3)
    assert torch.equal(y[i:i + 1], yi) and torch.equal(dx[i:i + 1], dxi), (i, fid)
    assert torch.equal(dp[i:i + 1, :P], dpi) and torch.equal(dm[i:i + 1], dmi), (i, fid)
    assert float(dp[i, P:].abs().max()) == 0.0 if P < 24 else True
    # and the float64 restatement (raw mask parameters that reproduce the float32 squashed ones)
    raw = torch.atanh(torch.from_numpy(mp[i:i + 1].astype(np.float64)) / 5.0)
    ref = ft.apply_masked(fid, torch.from_numpy(x[i:i + 1].astype(np.float64)),
                          torch.from_numpy(p24[i:i + 1, :P].astype(np.float64)), raw, 1.0, 0.3).numpy()
    assert_image_close(yi.float().cpu().numpy(), ref, NP_DT[dtype], 'masked dispatch image %d filter %d' % (i, fid))


def test_agent_with_masking_on_gpu(gpu_device):
  """Agent.forward with cfg.masking = True on the GPU: one masked-dispatch launch per step instead of eight masked
  applies; output against the float64 restatement from the agent's own parameters, gradients reach the mask rows of the
  selected filters' heads only."""
  from oracle import filters_torch as ft
  dev = gpu_device
  cfg = make_cfg()
  cfg.masking = True
  torch.manual_seed(2)
  ag = xagent.Agent(cfg).to(dev)
  n = 6
  rng = np.random.default_rng(3)
  img = torch.from_numpy((rng.random((n, 64, 64, 3))**2.2).astype(np.float32)).to(dev)
  states = torch.zeros(n, 11, device=dev)
  z = torch.from_numpy(rng.random((n, 131)).astype(np.float32)).to(dev)
  masks = [torch.from_numpy((rng.random((n, 4096)) < 0.5).astype(np.float32)).to(dev) for _ in range(2)]
  (out, new_states, surrogate, penalty), dbg, _ = ag((img, z, states), is_train=1, progress=0.5, dropout_masks=masks)
  out.sum().backward()
  ids = dbg['selected_filter_ids'].cpu().numpy()
  with torch.no_grad():
    feats = ag.filter_features(xagent.enrich_image_input(cfg, img, states), masks[0])
  for i in range(n):
    j = int(ids[i])
    filt = ag.filters[j]
    with torch.no_grad():
      f, mraw = filt.extract_parameters(feats[i:i + 1])
      packed = filt.pack(filt.filter_param_regressor(f)).double().cpu()
    ref = ft.apply_masked(filt.filter_id, img[i:i + 1].double().cpu(), packed, mraw.double().cpu(),
                          cfg.maximum_sharpness, cfg.minimum_strength).numpy()
    assert np.abs(out[i:i + 1].detach().cpu().numpy() - ref).max() < 5e-5
  chosen = set(int(v) for v in ids)
  for j, filt in enumerate(ag.filters):
    p = filt.get_num_filter_parameters()
    has = float(filt.fc2.weight.grad[p:].abs().max()) > 0.0
    assert has == (j in chosen), (j, has)
  ref_pen = (np.maximum(out.detach().cpu().numpy().astype(np.float64) - 1, 0)**2).mean(axis=(1, 2, 3))
  assert np.abs(penalty.detach().cpu().numpy()[:, 0] - ref_pen).max() < 2.0  # contains the entropy / usage terms too


def test_replay_memory_on_gpu_equals_the_round3_pool(gpu_device):
  """The slot pool with its pinned staging ring on the device against the round-3 implementation on the same device:
  identical batches draw for draw (tests/test_replay_and_loop.py runs the same comparison on the CPU)."""
  from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
  from tests import _replay_r03 as old
  dev = gpu_device
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size = 16, 48
  a = ReplayMemory(cfg, SyntheticProvider(dev, dtype=torch.float16, seed=5), SyntheticProvider(dev, gamma=1.0, seed=6), seed=7)
  b = old.ReplayMemory(cfg, old.SyntheticProvider(dev, dtype=torch.float16, seed=5), old.SyntheticProvider(dev, gamma=1.0, seed=6),
                       seed=7)
  for it in range(25):
    fa, feat_a = a.get_feed_dict_and_states(16)
    fb, feat_b = b.get_feed_dict_and_states(16)
    for k in fa:
      assert torch.equal(fa[k], fb[k]), (it, k)
    st = fa['states'].clone()
    stopped = ((st[:, 2] + 1 - cfg.test_steps).abs() < 1e-4).float()
    st[:, 0], st[:, 1], st[:, 2] = stopped, stopped, st[:, 2] + 1
    img = (fa['fake_input'].float() * 1.1).half()
    a.replace_memory(img, st, feat_a, advanced=True)
    b.replace_memory(img, st, feat_b)
    assert torch.equal(a.states, b.states) and torch.equal(a.images, b.images) and a.check_host_mirror()
    if int((b.states[:, 1] > 0).sum()) > 0:
      ra, rb = a.get_replay_feed_dict(16), b.get_replay_feed_dict(16)
      for k in ra:
        assert torch.equal(ra[k], rb[k]), (it, k)


def test_fused_heads_tail_matches_the_regressors(gpu_device):
  """expo_heads_regress_fwd / _bwd (the eight filter_param_regressors + the one-hot gather in one launch each way)
  against the op-by-op torch path of the same modules: parameters, and the gradients that reach every head's raw
  features; -1 (nothing selected) gives a zero row and no gradient; against the NumPy regressors too."""
  from exposure_amd import filters as F
  dev = gpu_device
  torch.manual_seed(3)
  cfg = make_cfg()
  ag = xagent.Agent(cfg).to(dev)
  n = 19
  rng = np.random.default_rng(4)
  raws = [torch.from_numpy(rng.standard_normal((n, f.get_num_filter_parameters() + 6)).astype(np.float32) * 1.5).to(dev).requires_grad_(True)
          for f in ag.filters]
  sel = torch.from_numpy(((np.arange(n) % 9) - 1).astype(np.int32)).to(dev)  # -1, 0..7, -1, ...
  p24 = F.heads_regress_select(list(ag.filters), raws, sel)
  # op-by-op: regress every head, gather with the one-hot
  onehot = (sel[:, None] == torch.arange(8, device=dev)[None, :]).float()
  ref_raws = [r.detach().clone().requires_grad_(True) for r in raws]
  ref = torch.zeros((n, 24), device=dev)
  for j, (f, r) in enumerate(zip(ag.filters, ref_raws)):
    pj = f.pack(f.filter_param_regressor(r[:, :f.get_num_filter_parameters()])).float()
    ref = ref + torch.nn.functional.pad(pj, (0, 24 - pj.shape[1])) * onehot[:, j:j + 1]
  assert float((p24 - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
  assert float(p24[sel < 0].abs().max()) == 0.0
  for i in range(n):  # and the NumPy restatement of the regressors
    j = int(sel[i])
    if j >= 0:
      fid = ag.filters[j].filter_id
      want = fnp.regress_packed(fid, raws[j][i:i + 1, :fnp.NUM_PARAMS[fid]].detach().cpu().numpy().astype(np.float64))
      assert np.abs(p24[i, :fnp.NUM_PARAMS[fid]].detach().cpu().numpy() - want[0]).max() <= 3e-6 * max(1.0, np.abs(want).max())
  w = torch.from_numpy(rng.standard_normal((n, 24)).astype(np.float32)).to(dev)
  (p24 * w).sum().backward()
  (ref * w).sum().backward()
  for j, (a, b) in enumerate(zip(raws, ref_raws)):
    scale = float(b.grad.abs().max()) + 1e-12
    assert float((a.grad - b.grad).abs().max()) <= 1e-5 * scale + 1e-7, (j, float((a.grad - b.grad).abs().max()), scale)
    assert float(a.grad[:, ag.filters[j].get_num_filter_parameters():].abs().max()) == 0.0  # mask features: no gradient (masking off)


@pytest.mark.parametrize('is_train', [1, 0])
def test_fused_agent_step_equals_the_op_by_op_step(is_train, gpu_device, monkeypatch):
  """The training-time fast path (one selection kernel + one regress-and-gather kernel + dispatch) against the op-by-op
  torch path of the SAME module: selected ids and new states bit-equal, pdf / surrogate / penalty / output image equal
  to rounding, and the gradients of a scalar of all differentiable outputs with respect to every generator parameter.
  The ids are also the NumPy sampler's for the kernel's own pdf (pdf_sample_layer.py:5-10), incl. noise 0 -> -1."""
  dev = gpu_device
  torch.manual_seed(5)
  cfg = make_cfg()
  ag = xagent.Agent(cfg).to(dev)
  with torch.no_grad():
    for p in ag.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.3)  # non-uniform pdf
  rng = np.random.default_rng(2)
  n = 24
  img = torch.from_numpy(synthetic.make_images(rng, (n, 64, 64, 3), np.float32)).to(dev)
  states = np.zeros((n, 11), dtype=np.float32)
  states[:, 2] = rng.integers(0, 6, n)
  states[:, 3:] = rng.random((n, 8)) < 0.3
  z = rng.random((n, 131), dtype=np.float32)
  z[0, 0] = 0.0
  masks = [torch.from_numpy((rng.random((n, 4096)) < 0.5).astype(np.float32)).to(dev) for _ in range(2)]
  t = lambda a: torch.from_numpy(a).to(dev)
  w_img = torch.randn(n, 64, 64, 3, device=dev) * 1e-3

  def run(fused):
    ag.zero_grad(set_to_none=True)
    if not fused:
      monkeypatch.setattr(xagent.Agent, '_forward_fused', None, raising=True)
      monkeypatch.setattr(filters, 'FUSED_HEAD_TYPES', ())
    (out, new_states, surrogate, penalty), dbg, _ = ag((img, t(z), t(states)), is_train=is_train, progress=0.3, dropout_masks=masks)
    loss = (out.float() * w_img).sum() + (surrogate * 0.7).sum() - (penalty * 1.3).sum()
    loss.backward()
    grads = [p.grad.detach().clone() if p.grad is not None else None for p in ag.parameters()]
    monkeypatch.undo()
    return out.detach(), new_states.detach(), surrogate.detach(), penalty.detach(), dbg, grads

  fo, fs, fsur, fpen, fdbg, fg = run(True)
  ro, rs, rsur, rpen, rdbg, rg = run(False)
  assert 'params24' in fdbg and fdbg['filter_debug_info'] == [] and len(rdbg['filter_debug_info']) == 8
  ids = fdbg['selected_filter_ids'].cpu().numpy()
  assert np.array_equal(ids, rdbg['selected_filter_ids'].cpu().numpy())
  if is_train:
    assert ids[0] == -1
    assert np.array_equal(ids, agent_np.pdf_sample(fdbg['pdf_batch'].cpu().numpy(), z[:, 0:1]))
  assert torch.equal(fs, rs)
  assert float((fdbg['pdf_batch'] - rdbg['pdf_batch']).abs().max()) <= 2e-7
  assert float((fsur - rsur).abs().max()) <= 2e-6 and float(((fpen - rpen).abs() / rpen.abs().clamp_min(1.0)).max()) <= 2e-6
  assert float((fo - ro).abs().max()) <= 2e-5 * max(1.0, float(ro.abs().max()))
  for (name, _), a, b in zip(ag.named_parameters(), fg, rg):
    assert (a is None) == (b is None), name
    if a is not None:
      scale = float(b.abs().max()) + 1e-12
      assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-8, (name, float((a - b).abs().max()), scale)


def test_packed_heads_equal_the_per_head_layers(gpu_device):
  """filters.PackedHeads: the K heads' fc2(lrelu(fc1(.))) as one GEMM + one batched GEMM over parameters packed IN
  PLACE.  Outputs (first P_j + 6 columns; zeros behind) and the gradients of every head parameter and of the features
  against the per-head nn.Linear path on the same weights; the parameters stay the optimiser's / the state dict's:
  an in-place update and a load_state_dict are seen by the packed operands, and .to() / a replaced storage re-packs."""
  from exposure_amd import filters as F
  from exposure_amd.util import lrelu
  dev = gpu_device
  torch.manual_seed(5)
  cfg = make_cfg()
  ag = xagent.Agent(cfg).to(dev)
  with torch.no_grad():
    for f in ag.filters:  # biases are zero-initialised: make them matter
      f.fc1.bias.normal_(0, 0.1)
      f.fc2.bias.normal_(0, 0.1)
  ref_state = {k: v.detach().clone() for k, v in ag.state_dict().items()}
  n = 13
  feats = torch.randn(n, cfg.feature_extractor_dims, device=dev)
  pack = F.PackedHeads(ag.filters)
  assert pack.supported()

  def compare():
    fa = feats.clone().requires_grad_(True)
    fb = feats.clone().requires_grad_(True)
    for p in ag.parameters():
      p.grad = None
    outs = pack(fa)
    ws = [torch.randn(n, pack.PAD, device=dev, generator=torch.Generator(device=dev).manual_seed(j)) for j in range(len(outs))]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    got = [p.grad.clone() for p in pack.leaves()] + [fa.grad.clone()]
    for p in ag.parameters():
      p.grad = None
    refs = [f.fc2(lrelu(f.fc1(fb))) for f in ag.filters]
    sum((r * w[:, :r.shape[1]]).sum() for r, w in zip(refs, ws)).backward()
    want = [p.grad.clone() for p in pack.leaves()] + [fb.grad.clone()]
    for j, (o, r) in enumerate(zip(outs, refs)):
      width = r.shape[1]
      assert o.shape == (n, pack.PAD) and float(o[:, width:].abs().max()) == 0.0
      assert float((o[:, :width] - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), j
    for i, (a, b) in enumerate(zip(got, want)):
      assert a.shape == b.shape
      assert float((a - b).abs().max()) <= 2e-5 * max(1e-3, float(b.abs().max())), i

  compare()
  assert pack._aliased()
  w1_ptr = pack.w1.data_ptr()
  with torch.no_grad():  # an optimiser-style in-place update of the Parameters is seen by the packed operands
    for p in pack.leaves():
      p.add_(0.01)
  compare()
  assert pack.w1.data_ptr() == w1_ptr and pack._aliased()
  ag.load_state_dict(ref_state)  # copies into the aliased storage
  assert pack._aliased() and torch.equal(ag.filters[3].fc1.weight, ref_state['filters.3.fc1.weight'])
  compare()
  # names and values of the state dict are the per-filter ones
  sd = ag.state_dict()
  assert all(torch.equal(sd[k], v) for k, v in ref_state.items())
  ag.filters[2].fc1.weight.data = ag.filters[2].fc1.weight.data.clone()  # someone replaced a storage: re-pack
  assert not pack._aliased()
  compare()
  assert pack._aliased() and pack.w1.data_ptr() != w1_ptr
