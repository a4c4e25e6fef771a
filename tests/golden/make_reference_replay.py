"""A trace of the REFERENCE's replay memory, produced by running its own code in the build container (never on the GPU
box: /root/reference does not exist there).

``/root/reference/replay_memory.py``'s pool logic is plain Python + NumPy -- ``fill_pool``, ``get_next_fake_batch`` (pop
non-terminated records, drop the terminated ones in front of them), ``replay_fake_batch`` (terminated records only, with
repetition), ``replace_memory`` (shuffle, keep a record unless it is over-length and loses a coin flip, refill, shuffle),
the record <-> array helpers -- only ``__init__`` builds TensorFlow placeholders.  The module cannot be imported
(``import tensorflow``; ``util.py`` does not parse on Python >= 3.7), so the class is cut out of the file by ``ast`` and
``Dict`` out of ``util.py`` by its line span, ``__init__`` is bypassed, and the class runs against a tiny provider whose
images carry their record id.  The module-level ``random`` the class calls is a recording wrapper around a seeded
``random.Random``: every shuffle is stored as the permutation it applied and every ``random()`` as its value, so that an
implementation with a DIFFERENT random source can be driven through the SAME decisions
(tests/test_reference_replay.py does that with the round-3 specification pool and the product's slot pool).

The committed fixture (tests/golden/reference_replay.json) holds numbers only: the configuration, per event the
decisions, the records returned (ids, states) and the pool afterwards, and the sha256 of the two source files.

  python tests/golden/make_reference_replay.py
"""
import ast
import hashlib
import json
import os
import random as pyrandom
import re
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
STATE_REWARD_DIM, STATE_STOPPED_DIM, STATE_STEP_DIM, STATE_DROPOUT_BEGIN = 0, 1, 2, 3  # util.py:13-16 (checked below)


def sha256(path):
  return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def cut_class(path, name):
  tree = ast.parse(open(path).read(), filename=path)
  node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
  return ast.Module(body=[node], type_ignores=[])


def cut_span(path, pattern):
  """A top-level statement of a file that does not parse as a whole: from the line matching ``pattern`` to the next
  top-level statement."""
  lines = open(path).read().split('\n')
  start = next(i for i, l in enumerate(lines) if re.match(pattern, l))
  end = start + 1
  while end < len(lines) and (lines[end].strip() == '' or lines[end][0] in ' \t'):
    end += 1
  return ast.parse('\n'.join(lines[start:end]), filename='%s:%d' % (path, start + 1))


class RecordingRandom:
  """Stands in for the ``random`` module inside the reference class: same results as ``random.Random(seed)``, every
  call logged."""

  def __init__(self, seed):
    self.rng = pyrandom.Random(seed)
    self.log = []

  def shuffle(self, pool):
    before = [float(r.feature) for r in pool]
    assert len(set(before)) == len(before), 'record ids are unique inside the pool'
    self.rng.shuffle(pool)
    after = [float(r.feature) for r in pool]
    index = {v: i for i, v in enumerate(before)}
    self.log.append(('perm', [index[v] for v in after]))  # new[i] = old[perm[i]]

  def random(self):
    u = self.rng.random()
    self.log.append(('uniform', u))
    return u

  def take(self):
    out, self.log = self.log, []
    return out


class Provider:
  """get_next_batch(n) -> (images, features): feature = a running record id, image = 2 x 2 x 3 filled with the id."""

  def __init__(self):
    self.count = 0

  def get_next_batch(self, n):
    ids = np.arange(self.count, self.count + n, dtype=np.float32)
    self.count += n
    return [np.full((2, 2, 3), i, dtype=np.float32) for i in ids], list(ids)


def main():
  util_path, rm_path = os.path.join(REF, 'util.py'), os.path.join(REF, 'replay_memory.py')
  ns = {'np': np}
  exec(compile(cut_span(util_path, r'class Dict\(dict\):'), '<reference util.Dict>', 'exec'), ns)
  for const, want in (('STATE_REWARD_DIM', 0), ('STATE_STOPPED_DIM', 1), ('STATE_STEP_DIM', 2), ('STATE_DROPOUT_BEGIN', 3)):
    exec(compile(cut_span(util_path, r'%s = ' % const), '<reference util>', 'exec'), ns)
    assert ns[const] == want
  rec = RecordingRandom(20260927)
  ns['random'] = rec
  exec(compile(cut_class(rm_path, 'ReplayMemory'), '<reference ReplayMemory>', 'exec'), ns)
  Ref = ns['ReplayMemory']

  cfg = types.SimpleNamespace(num_state_dim=11, filters=list(range(8)), batch_size=6, replay_memory_size=20,
                              maximum_trajectory_length=7, over_length_keep_prob=0.5, supervised=False, test_steps=5)
  mem = Ref.__new__(Ref)  # __init__ builds TensorFlow placeholders: bypassed
  mem.cfg = cfg
  mem.fake_dataset = Provider()
  mem.image_pool = []
  mem.target_pool_size = cfg.replay_memory_size

  def pool_snapshot():
    return {'ids': [float(r.feature) for r in mem.image_pool],
            'states': [[float(v) for v in r.state[:3]] for r in mem.image_pool]}

  events = []
  mem.load()
  events.append({'op': 'load', 'decisions': rec.take(), 'pool': pool_snapshot()})
  b = cfg.batch_size
  for it in range(60):
    images, states, features = mem.get_next_fake_batch(b)
    assert float(states[:, STATE_STOPPED_DIM].max()) == 0.0
    events.append({'op': 'pop', 'decisions': rec.take(), 'ids': [float(f) for f in features],
                   'states': states[:, :3].tolist(), 'images_ok': bool(all(float(im[0, 0, 0]) // 1 == float(f) for im, f in zip(images, features))),
                   'pool': pool_snapshot()})
    # the agent's state update (agent.py:207-238): step + 1, submitted at test_steps -- except that every seventh record
    # never submits, so that trajectories run past maximum_trajectory_length and the keep coin is flipped
    new_states = states.copy()
    step = states[:, STATE_STEP_DIM] + 1
    never = (np.asarray(features) % 7 == 3)
    stopped = ((np.abs(step - cfg.test_steps) < 1e-4) & ~never).astype(np.float32)
    new_states[:, STATE_REWARD_DIM] = stopped
    new_states[:, STATE_STOPPED_DIM] = stopped
    new_states[:, STATE_STEP_DIM] = step
    new_images = images + 0.001  # the retouched image: its id stays readable (floor)
    records = Ref.images_and_states_to_records(new_images, new_states, features)
    mem.replace_memory(records)
    events.append({'op': 'replace', 'decisions': rec.take(), 'ids': [float(f) for f in features],
                   'new_states': new_states[:, :3].tolist(), 'pool': pool_snapshot()})
    if any(r.state[STATE_STOPPED_DIM] > 0 for r in mem.image_pool) and it % 2 == 1:
      for _ in range(2):
        images, states, features = mem.replay_fake_batch(b)
        assert float(states[:, STATE_STOPPED_DIM].min()) > 0
        events.append({'op': 'replay', 'decisions': rec.take(), 'ids': [float(f) for f in features],
                       'states': states[:, :3].tolist(), 'pool': pool_snapshot()})
  n_uniform = sum(1 for e in events for d in e['decisions'] if d[0] == 'uniform')
  n_discard = sum(1 for e in events if e['op'] == 'pop')  # informative only
  out = {'cfg': vars(cfg), 'events': events,
         'provenance': ['replay_memory.py sha256=%s' % sha256(rm_path), 'util.py sha256=%s' % sha256(util_path)],
         'summary': {'events': len(events), 'keep_coin_flips': n_uniform, 'pops': n_discard,
                     'records_created': mem.fake_dataset.count}}
  path = os.path.join(HERE, 'reference_replay.json')
  json.dump(out, open(path, 'w'), separators=(',', ':'))
  print('wrote %s (%d bytes): %s' % (path, os.path.getsize(path), out['summary']))


if __name__ == '__main__':
  if not os.path.isdir(REF):
    sys.exit('needs /root/reference (the build container)')
  main()
