"""CPU tests of the host logic around the filter path (agent step, losses) with the C-ABI binding
mocked by the oracle (tests/_fake_hip.py).  Integer outputs must be bit-identical to the oracle."""
import numpy as np
import pytest
import torch

from exposure_amd import agent as xagent
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from oracle import agent_np
from oracle import filters_np as fnp
from tests._fake_hip import fake_hip


def test_pdf_sample_known_answers():
  # pdf_sample_layer.py:55-78: pdf ~ (2,4,8) -> u in (0,1/7] -> 0, (1/7,3/7] -> 1, else 2
  pdf = np.tile(np.array([[2.0, 4.0, 8.0]], dtype=np.float32), (7, 1))
  u = np.array([[0.0], [0.05], [1 / 7 - 1e-4], [1 / 7 + 1e-4], [0.4], [0.43], [0.99]], dtype=np.float32)
  want = np.array([-1, 0, 0, 1, 1, 2, 2], dtype=np.int32)
  assert np.array_equal(agent_np.pdf_sample(pdf, u), want)
  got = xagent.pdf_sample(torch.from_numpy(pdf), torch.from_numpy(u))
  assert got.dtype == torch.int32 and np.array_equal(got.numpy(), want)
  # empirical frequencies 1/7, 2/7, 4/7
  rng = np.random.default_rng(0)
  uu = rng.random((70000, 1), dtype=np.float32)
  ids = agent_np.pdf_sample(np.tile(pdf[:1], (70000, 1)), uu)
  freq = np.bincount(ids, minlength=3) / 70000.0
  assert np.abs(freq - np.array([1, 2, 4]) / 7).max() < 0.01


def _scan_reference(pdf_row):
  """Independent scalar restatement of tf.cumsum(exclusive=True) in fp32: out[j] = out[j-1] + p[j-1]."""
  out, acc = [], np.float32(0.0)
  for v in pdf_row:
    out.append(acc)
    acc = np.float32(acc + np.float32(v))
  return np.array(out, dtype=np.float32)


def test_exclusive_cumsum_is_a_shifted_scan_not_cumsum_minus_pdf():
  """pdf_sample_layer.py:7.  The round-1 form ``cumsum(pdf) - pdf`` differs from the true exclusive
  scan by one ulp in ~22 % of fp32 entries; both the product and the oracle must be the scan."""
  rng = np.random.default_rng(5)
  pdf = rng.random((4096, 8), dtype=np.float32) + np.float32(1e-3)
  pdf = (pdf / pdf.sum(axis=1, keepdims=True)).astype(np.float32)
  want = np.stack([_scan_reference(r) for r in pdf])
  assert np.array_equal(agent_np.exclusive_cumsum(pdf), want)
  assert np.array_equal(xagent.exclusive_cumsum(torch.from_numpy(pdf)).numpy(), want)
  naive = np.cumsum(pdf, axis=1) - pdf
  assert (naive != want).mean() > 0.05  # the two forms really differ -> this test can catch a regression


def test_pdf_sample_noise_on_every_cdf_knot_plus_minus_one_ulp():
  """north_star: bit-identical step indices.  Put the selection noise exactly on each cdf knot and one
  fp32 ulp either side: ``cdf < u`` is a strict comparison, so u == knot_j selects j-1, u just above
  selects j.  Product (torch) and oracle (numpy) must agree with a scalar restatement on every case."""
  rng = np.random.default_rng(6)
  pdf = rng.random((512, 8), dtype=np.float32) + np.float32(1e-3)
  norm = (pdf / (agent_np.row_sum(pdf) + np.float32(1e-36))).astype(np.float32)
  knots = np.stack([_scan_reference(r) for r in norm])  # (512, 8)
  rows, noise, want = [], [], []
  for i in range(pdf.shape[0]):
    for j in range(8):
      for u in (np.nextafter(knots[i, j], np.float32(-1)), knots[i, j], np.nextafter(knots[i, j], np.float32(2))):
        rows.append(pdf[i])
        noise.append(u)
        want.append(int((knots[i] < u).sum()) - 1)
  rows = np.stack(rows).astype(np.float32)
  noise = np.array(noise, dtype=np.float32)[:, None]
  want = np.array(want, dtype=np.int32)
  got_np = agent_np.pdf_sample(rows, noise)
  got_t = xagent.pdf_sample(torch.from_numpy(rows), torch.from_numpy(noise)).numpy()
  assert np.array_equal(got_np, want)
  assert np.array_equal(got_t, want)
  # on a knot the strict '<' excludes segment j itself; one ulp above includes it
  assert (want.reshape(-1, 3)[:, 2] >= want.reshape(-1, 3)[:, 1]).all()
  assert (want.reshape(-1, 3)[:, 2] > want.reshape(-1, 3)[:, 1]).mean() > 0.9


def make_inputs(n=6, seed=0, s=64):
  rng = np.random.default_rng(seed)
  img = (rng.random((n, s, s, 3), dtype=np.float32)**2.2).astype(np.float32)
  states = np.zeros((n, 11), dtype=np.float32)
  states[:, 2] = rng.integers(0, 5, n)  # step
  states[:, 3:] = (rng.random((n, 8)) < 0.3)
  z = rng.random((n, 131), dtype=np.float32)
  masks = [(rng.random((n, 4096)) < 0.5).astype(np.float32) for _ in range(2)]
  return img, states, z, masks


@pytest.mark.parametrize('is_train', [0, 1])
def test_agent_step_matches_oracle(is_train):
  torch.manual_seed(0)
  cfg = make_cfg()
  ag = xagent.Agent(cfg)
  img, states, z, masks = make_inputs()
  z[0, 0] = 0.0  # noise 0 -> id -1 (all-zero one-hot) when sampling
  t = lambda a: torch.from_numpy(a)
  with fake_hip():
    (out, new_states, surrogate, penalty), dbg, _ = ag((t(img), t(z), t(states)), is_train=is_train, progress=0.25,
                                                      dropout_masks=[t(m) for m in masks])
  # oracle from the SAME torch-computed logits / params (the nets are torch plumbing)
  pdf = dbg['pdf_batch'].detach().numpy()
  ids = dbg['selected_filter_ids'].numpy()
  assert ids.dtype == np.int32
  # recompute the selection from the selector logits with the numpy restatement
  with fake_hip(), torch.no_grad():
    enriched = xagent.enrich_image_input(cfg, t(img), t(states))
    sel = ag.selector_features(enriched, t(masks[1]))
    logits = ag.selector_fc2(xagent.lrelu(ag.selector_fc1(sel))).numpy()
    feats = ag.filter_features(enriched, t(masks[0]))
    params = [f.pack(p).numpy() for f, p in zip(ag.filters, ag.regress_all(feats)[0])]
  o_pdf, o_ent, o_ids, o_onehot, o_sur = agent_np.action_selection(logits.astype(np.float64), z[:, 0:1].astype(np.float64),
                                                                    is_train)
  assert np.array_equal(ids, o_ids), (ids, o_ids)
  if is_train:
    assert ids[0] == -1
  np.testing.assert_allclose(pdf, o_pdf, rtol=1e-5)
  np.testing.assert_allclose(surrogate.detach().numpy(), o_sur, rtol=1e-4, atol=1e-6)
  o_img = agent_np.apply_all_and_select(img.astype(np.float64), [p.astype(np.float64) for p in params], o_onehot)
  np.testing.assert_allclose(out.detach().numpy(), o_img, rtol=1e-5, atol=1e-6)
  o_states, o_usage, o_last, o_sub = agent_np.new_states(states.astype(np.float64), o_onehot)
  assert np.array_equal(new_states.numpy(), o_states.astype(np.float32))
  o_pen = agent_np.penalty(o_img, o_ent, o_usage, o_last, o_sub, 0.25)
  np.testing.assert_allclose(penalty.detach().numpy(), o_pen, rtol=1e-4, atol=1e-6)


def test_states_after_five_steps():
  torch.manual_seed(1)
  cfg = make_cfg()
  ag = xagent.Agent(cfg)
  img, states, z, _ = make_inputs(n=3, seed=2)
  states[:] = 0
  t = torch.from_numpy
  cur, st = t(img), t(states)
  used = np.zeros((3, 8))
  with fake_hip(), torch.no_grad():
    for k in range(5):
      (cur, st, _, _), dbg, _ = ag((cur, t(z), st), is_train=0, progress=0.0)
      used[np.arange(3), dbg['selected_filter_ids'].numpy()] = 1
      done = float(k + 1 == cfg.test_steps)
      assert np.array_equal(st[:, :3].numpy(), np.tile([[done, done, k + 1]], (3, 1)).astype(np.float32))
      assert np.array_equal(st[:, 3:].numpy(), used.astype(np.float32))


def test_high_res_path_uses_low_res_parameters():
  torch.manual_seed(2)
  cfg = make_cfg()
  ag = xagent.Agent(cfg)
  img, states, z, masks = make_inputs(n=2, seed=3)
  hi = (np.random.default_rng(5).random((2, 96, 80, 3), dtype=np.float32)**2.2)
  t = torch.from_numpy
  with fake_hip(), torch.no_grad():
    (low, new_states, high), dbg, _ = ag((t(img), t(z), t(states)), is_train=0, progress=0.0, high_res=t(hi),
                                         dropout_masks=[t(m) for m in masks])
  ids = dbg['selected_filter_ids'].numpy()
  p24 = dbg['params24'].numpy()
  for n in range(2):
    fid = int(ids[n])
    p = p24[n:n + 1, :fnp.NUM_PARAMS[fid]].astype(np.float64)
    np.testing.assert_allclose(high[n:n + 1].numpy(), fnp.process_packed(fid, hi[n:n + 1].astype(np.float64), p),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(low[n:n + 1].numpy(), fnp.process_packed(fid, img[n:n + 1].astype(np.float64), p),
                               rtol=1e-5, atol=1e-6)


def test_gan_steps_run_and_update_only_their_parameters():
  torch.manual_seed(3)
  cfg = make_cfg()
  gan = GAN(cfg)
  img, states, z, masks = make_inputs(n=4, seed=4)
  real = (np.random.default_rng(6).random((4, 64, 64, 3), dtype=np.float32))
  t = torch.from_numpy
  snap = lambda m: [p.detach().clone() for p in m.parameters()]
  g0, v0, c0 = snap(gan.generator), snap(gan.value), snap(gan.critic)
  with fake_hip():
    out = gan.generator_step(t(img), t(z), t(states), progress=0.1, it=5, dropout_masks=[t(m) for m in masks])
  assert np.isfinite(float(out['g_loss'])) and np.isfinite(float(out['v_loss']))
  changed = lambda a, m: any(not torch.equal(x, y) for x, y in zip(a, m.parameters()))
  assert changed(g0, gan.generator) and changed(v0, gan.value) and not changed(c0, gan.critic)
  g1, v1 = snap(gan.generator), snap(gan.value)
  with fake_hip():
    out = gan.critic_step(t(real), t(img), it=5)
  assert np.isfinite(float(out['c_loss'])) and float(out['gradient_norm']) > 0
  assert changed(c0, gan.critic) and not changed(g1, gan.generator) and not changed(v1, gan.value)
  # iteration 0 runs the generator with lr_g = 0 (net.py:327-328)
  g2 = snap(gan.generator)
  with fake_hip():
    gan.generator_step(t(img), t(z), t(states), progress=0.0, it=0)
  assert not changed(g2, gan.generator)


def test_lr_schedules():
  cfg = make_cfg()
  assert cfg.lr_g(0) == pytest.approx(0.3 * 5e-5)
  assert cfg.lr_c(20000) == pytest.approx(5e-5 * 0.1**3)
  assert cfg.lr_g(10000) == pytest.approx(0.3 * 5e-5 * 0.1**1.5)


def test_evaluate_loop_two_filter_config():
  """BASELINE config 1 plumbing: cfg.filters = [Exposure, Gamma], one image, 5 steps, CPU-only."""
  from exposure_amd import evaluate, filters
  torch.manual_seed(4)
  cfg = make_cfg(filters=[filters.GammaFilter, filters.ExposureFilter])  # reordered on purpose
  assert cfg.num_state_dim == 5
  ag = xagent.Agent(cfg)
  assert ag.abi_filter_ids.tolist() == [1, 0]
  hi = torch.from_numpy((np.random.default_rng(7).random((1, 96, 128, 3), dtype=np.float32)**2.2))
  masks = [[(torch.rand(1, 4096) < 0.5).float() for _ in range(2)] for _ in range(5)]
  z = torch.rand(1, cfg.z_dim)
  with fake_hip():
    out_hi, out_lo, states, trace = evaluate.retouch(ag, hi, z=z, dropout_masks=masks, return_trace=True)
    step_hi, _, _, trace2 = evaluate.retouch(ag, hi, z=z, dropout_masks=masks, return_trace=True, fused=False)
  assert torch.equal(trace, trace2)
  np.testing.assert_allclose(out_hi.numpy(), step_hi.numpy(), rtol=2e-5, atol=2e-6)  # fused == stepwise
  assert out_hi.shape == hi.shape and out_lo.shape == (1, 64, 64, 3)
  assert trace.shape == (1, 5)
  assert states[0, :3].tolist() == [1.0, 1.0, 5.0]
  # replay the trace with the oracle: parameters come from the low-res proxy, applied to high-res
  lo = evaluate.make_low_res(hi, 64)
  st = torch.zeros(1, 5)
  ref_hi = hi.numpy().astype(np.float64)
  with fake_hip(), torch.no_grad():
    for i in range(5):
      (lo2, st2, _), dbg, _ = ag((lo, z, st), is_train=0, progress=0.0, high_res=torch.from_numpy(ref_hi).float(),
                                 dropout_masks=masks[i])
      j = int(dbg['selected_filter_ids'][0])
      assert j == int(trace[0, i])
      fid = int(ag.abi_filter_ids[j])
      p = dbg['params24'][:, :fnp.NUM_PARAMS[fid]].numpy().astype(np.float64)
      ref_hi = fnp.process_packed(fid, ref_hi, p)
      lo, st = lo2, st2
  np.testing.assert_allclose(out_hi.numpy(), ref_hi, rtol=2e-5, atol=2e-6)


def test_packed_heads_host_logic_on_cpu():
  """filters.PackedHeads with the C-ABI mocked: the heads' Parameters become views of the packed matrices without
  changing a value or a state-dict entry, the fused node reproduces the per-head layers and their gradients, deepcopy
  keeps the aliasing inside the copy, and a replaced storage is detected and re-packed."""
  import copy
  from exposure_amd import filters as F
  from exposure_amd.util import lrelu
  torch.manual_seed(2)
  cfg = make_cfg()
  ag = xagent.Agent(cfg)
  with torch.no_grad():
    for f in ag.filters:
      f.fc1.bias.normal_(0, 0.1)
      f.fc2.bias.normal_(0, 0.1)
  before = {k: v.clone() for k, v in ag.state_dict().items()}
  pack = F.PackedHeads(ag.filters)
  assert pack.supported() and not pack._aliased()
  pack.ensure()
  assert pack._aliased()
  after = ag.state_dict()
  assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
  assert all(isinstance(p, torch.nn.Parameter) and p.is_leaf for p in pack.leaves())
  feats = torch.randn(6, cfg.feature_extractor_dims)
  with fake_hip():
    fa = feats.clone().requires_grad_(True)
    outs = pack(fa)
    sum((o * (j + 1)).sum() for j, o in enumerate(outs)).backward()
    got = [p.grad.clone() for p in pack.leaves()] + [fa.grad.clone()]
    for p in ag.parameters():
      p.grad = None
    fb = feats.clone().requires_grad_(True)
    refs = [f.fc2(lrelu(f.fc1(fb))) for f in ag.filters]
    sum((r * (j + 1)).sum() for j, r in enumerate(refs)).backward()
    want = [p.grad.clone() for p in pack.leaves()] + [fb.grad.clone()]
  for o, r in zip(outs, refs):
    assert torch.allclose(o[:, :r.shape[1]], r, rtol=1e-5, atol=1e-6) and float(o[:, r.shape[1]:].abs().max()) == 0.0
  for a, b in zip(got, want):
    assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-4, atol=1e-6)
  with fake_hip():  # a consumer that uses ONE head: the other heads' upstream gradients arrive as None
    for p in ag.parameters():
      p.grad = None
    outs = pack(feats.clone().requires_grad_(True))
    (outs[3].sum() * 2.0).backward()
  assert float(ag.filters[3].fc2.weight.grad.abs().sum()) > 0 and float(ag.filters[0].fc2.weight.grad.abs().sum()) == 0.0
  # a deep copy's Parameters own fresh storage (nn.Parameter.__deepcopy__ clones): its pack notices and re-packs from
  # the PARAMETERS (the source of truth), never from its stale packed copy
  ag._packed_heads = pack
  twin = copy.deepcopy(ag)
  with torch.no_grad():
    twin.filters[0].fc1.weight.add_(1.0)
  assert not twin._packed_heads._aliased()
  twin._packed_heads.ensure()
  assert twin._packed_heads._aliased() and twin._packed_heads.w1.data_ptr() != pack.w1.data_ptr()
  assert torch.equal(twin._packed_heads.w1[:128], before['filters.0.fc1.weight'] + 1.0)
  assert torch.equal(pack.w1[:128], before['filters.0.fc1.weight'])
  # anything that replaces a parameter's storage is noticed at the next call
  ag.filters[5].fc2.bias.data = ag.filters[5].fc2.bias.data.clone()
  assert not pack._aliased()
  pack.ensure()
  assert pack._aliased() and all(torch.equal(ag.state_dict()[k], before[k]) for k in before)
