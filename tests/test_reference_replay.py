"""The replay memory against a trace of the REFERENCE's own pool (tests/golden/reference_replay.json, produced by
tests/golden/make_reference_replay.py: /root/reference/replay_memory.py's ``ReplayMemory`` -- cut out of the file, its
TensorFlow ``__init__`` bypassed -- run in the build container with every ``random.shuffle`` recorded as the permutation
it applied and every ``random.random()`` as its value).

The implementations here draw their randomness from a torch generator, so they are driven through the reference's
DECISIONS instead of its seed: ``torch.randperm`` / ``torch.rand`` on the memory's generator return the recorded
permutation / coin flips.  Under the same decisions the round-3 specification pool (tests/_replay_r03.py) and the
product's slot pool (exposure_amd/replay_memory.py) must return the same records -- ids, (reward, stopped, step) -- for
every pop / replace / replay of the trace, and hold the same pool in the same order afterwards: 60 generator batches
incl. terminated records dropped in front of a pop, refills, over-length trajectories with their keep coin flips, 54
critic replays with repetition."""
import json
import os
from contextlib import contextmanager

import numpy as np
import pytest
import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def trace():
  return json.load(open(os.path.join(HERE, 'reference_replay.json')))


class IdProvider:
  """get_next_batch(n) -> (images, features): feature = a running record id, image = 2 x 2 x 3 filled with the id (the
  provider of the reference run)."""

  def __init__(self):
    self.device = torch.device('cpu')
    self.count = 0

  def get_next_batch(self, n):
    ids = torch.arange(self.count, self.count + n, dtype=torch.float32)
    self.count += n
    return ids[:, None, None, None].expand(n, 2, 2, 3).contiguous(), ids


class Decisions:
  """The decisions of one event, handed out in call order."""

  def __init__(self):
    self.queue = []
    self.keep = None  # the vector the next 1-D torch.rand on the memory's generator returns

  def load(self, decisions):
    assert not self.queue and self.keep is None, 'the previous event consumed all of its decisions'
    self.queue = [tuple(d) for d in decisions]

  def perm(self, n):
    kind, value = self.queue.pop(0)
    assert kind == 'perm' and len(value) == n, (kind, len(value), n)
    return torch.tensor(value, dtype=torch.int64)

  def uniforms(self, over_length):
    """A coin per record: the reference flips one only for an over-length record (``or`` short-circuit,
    replay_memory.py:203-204), in record order; the others keep regardless (0.0)."""
    u = np.zeros(len(over_length))
    for i, over in enumerate(over_length):
      if over:
        kind, value = self.queue.pop(0)
        assert kind == 'uniform'
        u[i] = value
    return torch.from_numpy(u).float()


@contextmanager
def driven_by(mem, decisions):
  real_randperm, real_rand = torch.randperm, torch.rand

  def randperm(n, *args, generator=None, **kw):
    if generator is mem.rng:
      return decisions.perm(n)
    return real_randperm(n, *args, generator=generator, **kw)

  def rand(*size, generator=None, **kw):
    if generator is mem.rng and len(size) == 1 and isinstance(size[0], int):
      assert decisions.keep is not None and decisions.keep.numel() == size[0]
      out, decisions.keep = decisions.keep, None
      return out
    return real_rand(*size, generator=generator, **kw)

  torch.randperm, torch.rand = randperm, rand
  try:
    yield
  finally:
    torch.randperm, torch.rand = real_randperm, real_rand


def make_memory(kind, cfg):
  from exposure_amd.config import make_cfg
  full = make_cfg()
  for k, v in cfg.items():
    full[k] = v
  if kind == 'spec':
    from tests import _replay_r03 as mod
  else:
    from exposure_amd import replay_memory as mod
  return mod.ReplayMemory(full, IdProvider(), IdProvider(), seed=0)


def pool_of(mem):
  return [float(v) for v in mem.features.tolist()], [[float(x) for x in row[:3]] for row in mem.states.tolist()]


@pytest.mark.parametrize('kind', ['spec', 'product'])
def test_pool_follows_the_reference_trace(trace, kind):
  cfg = trace['cfg']
  assert len(trace['provenance']) == 2 and all('sha256=' in p for p in trace['provenance'])
  assert trace['summary']['keep_coin_flips'] >= 10  # the over-length branch is exercised
  decisions = Decisions()
  events = trace['events']
  assert events[0]['op'] == 'load' and events[0]['decisions'] == []
  mem = make_memory(kind, cfg)  # fills the pool: replay_memory.py:51-52
  with driven_by(mem, decisions):
    ids, states = pool_of(mem)
    assert ids == events[0]['pool']['ids'] and states == events[0]['pool']['states']
    last = None  # the batch popped last: (images, states, features)
    counts = {'pop': 0, 'replace': 0, 'replay': 0}
    for k, ev in enumerate(events[1:], start=1):
      decisions.load(ev['decisions'])
      op = ev['op']
      counts[op] += 1
      if op == 'pop':
        images, st, feats = mem.get_next_fake_batch(cfg['batch_size'])
        assert [float(v) for v in feats.tolist()] == ev['ids'], (k, op)
        assert [[float(x) for x in row[:3]] for row in st.tolist()] == ev['states'], (k, op)
        # a record's image travels with it: the id is readable in its pixels (floor), as in the reference run
        assert torch.equal(images[:, 0, 0, 0].floor(), feats)
        last = (images, st, feats)
      elif op == 'replace':
        images, st, feats = last
        assert [float(v) for v in feats.tolist()] == ev['ids']
        new_states = st.clone()
        new_states[:, :3] = torch.tensor(ev['new_states'], dtype=st.dtype)
        over = [row[2] >= cfg['maximum_trajectory_length'] for row in ev['new_states']]
        # decisions of replace_memory: shuffle, one coin per over-length record, (refill,) shuffle
        first = decisions.queue.pop(0)
        coins = Decisions()
        coins.queue = [d for d in decisions.queue if d[0] == 'uniform']
        decisions.queue = [first] + [d for d in decisions.queue if d[0] == 'perm']
        decisions.keep = coins.uniforms(over)
        assert not coins.queue
        mem.replace_memory(images + 0.001, new_states, feats)
      else:
        images, st, feats = mem.replay_fake_batch(cfg['batch_size'])
        assert [float(v) for v in feats.tolist()] == ev['ids'], (k, op)
        assert [[float(x) for x in row[:3]] for row in st.tolist()] == ev['states'], (k, op)
        assert float(st[:, 1].min()) > 0 and torch.equal(images[:, 0, 0, 0].floor(), feats)
      assert not decisions.queue and decisions.keep is None, (k, op, 'decisions left over')
      ids, states = pool_of(mem)
      assert ids == ev['pool']['ids'], (k, op)
      assert states == ev['pool']['states'], (k, op)
  assert counts == {'pop': 60, 'replace': 60, 'replay': trace['summary']['events'] - 121}
