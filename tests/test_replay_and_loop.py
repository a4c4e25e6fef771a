"""Replay memory semantics (replay_memory.py) and a short run of the G/C alternation loop
(net.py:307-365) on CPU with the C-ABI binding mocked by the oracle."""
import numpy as np
import torch

from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
from tests._fake_hip import fake_hip


def make_memory(cfg, seed=0):
  dev = torch.device('cpu')
  return ReplayMemory(cfg, SyntheticProvider(dev, dtype=torch.float32, seed=seed),
                      SyntheticProvider(dev, gamma=1.0, dtype=torch.float32, seed=seed + 1), seed=seed)


def test_pool_semantics():
  cfg = make_cfg()
  cfg.batch_size = 16
  cfg.replay_memory_size = 32
  mem = make_memory(cfg)
  assert len(mem) == 32 and float(mem.states.abs().max()) == 0.0
  feed, feats = mem.get_feed_dict_and_states(16)
  assert feed['fake_input'].shape == (16, 64, 64, 3) and feed['z'].shape == (16, cfg.z_dim)
  assert len(mem) == 16  # popped
  # return them: half terminated, a quarter over-length
  states = feed['states'].clone()
  states[:8, 1] = 1.0
  states[:, 2] = 3
  states[8:12, 2] = 9  # over maximum_trajectory_length -> kept with prob .5
  mem.replace_memory(feed['fake_input'], states, feats)
  assert len(mem) == 32
  n_done = int((mem.states[:, 1] > 0).sum())
  assert n_done == 8
  # generator batches never contain terminated records
  f2, _ = mem.get_feed_dict_and_states(16)
  assert float(f2['states'][:, 1].max()) == 0.0
  # critic replay contains ONLY terminated records (with repetition)
  mem.fill_pool()
  rep = mem.get_replay_feed_dict(16)
  assert rep['fake_output'].shape == (16, 64, 64, 3)
  _, st, _ = mem.replay_fake_batch(16)
  assert float(st[:, 1].min()) == 1.0


def test_slot_pool_equals_the_round3_pool_draw_for_draw():
  """Same seeds, same calls -> identical batches, states, features, noise and pool contents as the round-3
  implementation (tests/_replay_r03.py), through 40 rounds of pop / agent-like update / replace / 3 replays, with the
  host mirror (advanced=True) and with the read-back path; the mirror equals the device states throughout."""
  from tests import _replay_r03 as old
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size = 8, 24
  for advanced in (True, False):
    dev = torch.device('cpu')
    a = ReplayMemory(cfg, SyntheticProvider(dev, seed=5), SyntheticProvider(dev, gamma=1.0, seed=6), seed=7)
    b = old.ReplayMemory(cfg, old.SyntheticProvider(dev, seed=5), old.SyntheticProvider(dev, gamma=1.0, seed=6), seed=7)
    rng = np.random.default_rng(0)
    for it in range(40):
      fa, feat_a = a.get_feed_dict_and_states(8)
      fb, feat_b = b.get_feed_dict_and_states(8)
      for k in fa:
        assert torch.equal(fa[k], fb[k]), (it, k)
      assert torch.equal(feat_a, feat_b)
      # the agent's state update (agent.py:207-238) and some image change
      st = fa['states'].clone()
      stopped = ((st[:, 2] + 1 - cfg.test_steps).abs() < 1e-4).float()
      st[:, 0], st[:, 1], st[:, 2] = stopped, stopped, st[:, 2] + 1
      img = fa['fake_input'] * float(rng.uniform(0.5, 1.5))
      a.replace_memory(img, st, feat_a, advanced=advanced)
      b.replace_memory(img, st, feat_b)
      assert len(a) == len(b) and torch.equal(a.states, b.states) and torch.equal(a.images, b.images)
      assert torch.equal(a.features, b.features) and a.check_host_mirror()
      if int((b.states[:, 1] > 0).sum()) > 0:
        for _ in range(3):
          ra, rb = a.get_replay_feed_dict(8), b.get_replay_feed_dict(8)
          for k in ra:
            assert torch.equal(ra[k], rb[k]), (it, k)
    assert a.debug() == b.debug()


def test_training_loop_runs_and_terminates_trajectories():
  torch.manual_seed(0)
  cfg = make_cfg()
  cfg.batch_size = 4
  cfg.replay_memory_size = 8
  cfg.max_iter_step = 1000
  gan = GAN(cfg)
  mem = make_memory(cfg, seed=3)
  # shorten the warm-up constants of net.py:314-323 for the test
  cfg.critic_initialization = 0
  orig_giters = cfg.giters

  class Short(GAN):
    pass

  with fake_hip():
    # iteration 0 runs 100 generator steps with lr_g = 0 so terminated states exist (net.py:320-328);
    # run a reduced version: call the pieces of train() by hand
    for _ in range(6):
      feed, feats = mem.get_feed_dict_and_states(cfg.batch_size)
      out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
      mem.replace_memory(out['fake_output'], out['new_states'], feats)
    assert int((mem.states[:, 1] > 0).sum()) > 0  # step counter reached test_steps for some records
    hist = []
    for it in range(1, 3):
      feed, feats = mem.get_feed_dict_and_states(cfg.batch_size)
      g = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], it / 1000.0, it=it)
      mem.replace_memory(g['fake_output'], g['new_states'], feats)
      rep = mem.get_replay_feed_dict(cfg.batch_size)
      c = gan.critic_step(rep['real_data'], rep['fake_output'], it=it)
      hist.append((float(g['g_loss']), float(c['c_loss'])))
  assert all(np.isfinite(v) for pair in hist for v in pair)
  assert cfg.giters == orig_giters


def test_lazy_feeds_equal_the_gathered_ones():
  """``lazy=True`` hands out PoolRows (gathered by the consumer: GAN._replay writes them straight into the step graph's
  inputs): same seeds, same calls -> the same batches, features and pool contents as the gathered feeds, through pops,
  replaces with dropped rows (the trash-row scatter) and replays; a PoolRows used after the pool was written is an error."""
  import pytest
  from exposure_amd.replay_memory import PoolRows, materialize
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size = 8, 24
  dev = torch.device('cpu')
  a = ReplayMemory(cfg, SyntheticProvider(dev, seed=5), SyntheticProvider(dev, gamma=1.0, seed=6), seed=7)
  b = ReplayMemory(cfg, SyntheticProvider(dev, seed=5), SyntheticProvider(dev, gamma=1.0, seed=6), seed=7)
  lazies = 0
  for it in range(30):
    fa, feat_a = a.get_feed_dict_and_states(8, lazy=True)
    fb, feat_b = b.get_feed_dict_and_states(8)
    lazies += isinstance(fa['fake_input'], PoolRows)
    for k in fb:
      got = materialize(fa[k])
      assert got.shape == fb[k].shape and torch.equal(got, fb[k]), (it, k)
    if isinstance(fa['fake_input'], PoolRows):
      dst = torch.full(fa['fake_input'].shape, float('nan'))
      assert fa['fake_input'].into(dst) is dst and torch.equal(dst, fb['fake_input'])
    st = fb['states'].clone()
    stopped = ((st[:, 2] + 1 - cfg.test_steps).abs() < 1e-4).float()
    st[:, 0], st[:, 1], st[:, 2] = stopped, stopped, st[:, 2] + 1
    st[it % 8, 2] = 9  # over-length: dropped with probability 1 - over_length_keep_prob
    img = fb['fake_input'] * 0.9
    a.replace_memory(img, st, feat_a)  # (the features as PoolRows)
    b.replace_memory(img, st, feat_b)
    if isinstance(feat_a, PoolRows):
      with pytest.raises(AssertionError):
        feat_a.materialize()  # the pool has been written since
    assert len(a) == len(b) and torch.equal(a.states, b.states) and torch.equal(a.images, b.images)
    assert torch.equal(a.features, b.features) and a.check_host_mirror()
    if int((b.states[:, 1] > 0).sum()) > 0:
      ra, rb = a.get_replay_feed_dict(8, lazy=True), b.get_replay_feed_dict(8)
      for k in rb:
        assert torch.equal(materialize(ra[k]), rb[k]), (it, k)
  assert lazies >= 25


def test_resident_provider_serves_views_of_one_data_set():
  from exposure_amd.replay_memory import ResidentProvider
  dev = torch.device('cpu')
  p = ResidentProvider(dev, gamma=2.2, scale=0.5, dtype=torch.float16, seed=3, count=40)
  ref = SyntheticProvider(dev, gamma=2.2, scale=0.5, dtype=torch.float16, seed=3).get_next_batch(40)[0]
  assert torch.equal(p.images, ref) and p.images.dtype == torch.float16
  seen = []
  for _ in range(5):
    x, f = p.get_next_batch(16)
    assert x.shape == (16, 64, 64, 3) and x.data_ptr() >= p.images.data_ptr()  # a view, not a copy
    assert torch.equal(x, p.images[int(f[0]):int(f[0]) + 16])
    seen.append(int(f[0]))
  assert seen == [0, 16, 0, 16, 0]  # 40 images: two whole batches per epoch


def test_planned_iteration_equals_the_step_by_step_calls():
  """``plan_iteration`` (every pool decision of an iteration made ahead, on the host) + its device half against the
  step-by-step calls on a twin memory: same generator batch, noise, replays, real batches and pool contents, iteration after
  iteration, including iterations that drop over-length records and refill more than was popped."""
  from exposure_amd.replay_memory import ResidentProvider
  cfg = make_cfg()
  cfg.batch_size, cfg.replay_memory_size = 8, 24
  dev = torch.device('cpu')
  mk = lambda: ReplayMemory(cfg, ResidentProvider(dev, seed=5, count=64), ResidentProvider(dev, gamma=1.0, seed=6, count=48), seed=7)
  a, b = mk(), mk()

  def agent(img, st, it):
    st = st.clone()
    stopped = ((st[:, 2] + 1 - cfg.test_steps).abs() < 1e-4).float()
    st[:, 0], st[:, 1], st[:, 2] = stopped, stopped, st[:, 2] + 1
    return img * (0.8 + 0.01 * it), st

  planned = replays = 0
  for it in range(60):
    # the twin's generator part first: whether terminated records exist afterwards decides if critic steps can follow
    fb, feat_b = b.get_feed_dict_and_states(8)
    img_b, st_b = agent(fb['fake_input'], fb['states'], it)
    b.replace_memory(img_b, st_b, feat_b, advanced=True)
    citers = 3 if bool((b._h_stopped[b._order] > 0).any()) else 0
    replays += citers
    plan = a.plan_iteration(8, citers)
    if plan is None:  # the same step-by-step calls
      fa, feat_a = a.get_feed_dict_and_states(8)
      img, st = agent(fa['fake_input'], fa['states'], it)
      a.replace_memory(img, st, feat_a, advanced=True)
      reps_a = [a.get_replay_feed_dict(8) for _ in range(citers)]
      za, gi, gs, gf = fa['z'], fa['fake_input'], fa['states'], feat_a
    else:
      planned += 1
      t = lambda v: torch.from_numpy(np.ascontiguousarray(v))
      gi, gs, gf = a.planned_generator_batch(t(plan.g_slots))
      img, st = agent(gi, gs, it)
      a.planned_commit(t(plan.g_scatter), img, st, gf, t(plan.fresh_dst), t(plan.fresh_src))
      reps_a = []
      for j in range(citers):
        real, fake = a.planned_critic_batch(t(plan.c_slots[j]), t(plan.real_rows[j]))
        reps_a.append(dict(real_data=real, fake_output=fake))
      za = plan.z
    assert torch.equal(za, fb['z']) and torch.equal(gi, fb['fake_input']) and torch.equal(gs, fb['states'])
    assert torch.equal(gf, feat_b)
    for j in range(citers):
      rb = b.get_replay_feed_dict(8)
      assert torch.equal(reps_a[j]['real_data'], rb['real_data']) and torch.equal(reps_a[j]['fake_output'], rb['fake_output']), (it, j)
    assert len(a) == len(b) and torch.equal(a.states, b.states) and torch.equal(a.images, b.images)
    assert torch.equal(a.features, b.features) and a.check_host_mirror()
  assert planned >= 40 and replays >= 60
