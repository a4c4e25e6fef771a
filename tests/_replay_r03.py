"""TEST INFRASTRUCTURE: the round-3 implementation of exposure_amd/replay_memory.py, kept verbatim as the executable
specification of the pool's record semantics and random-decision sequence -- the slot-based round-4 implementation must
return identical batches draw for draw (tests/test_replay_and_loop.py).  Never imported by the product.

Replay memory of the trainer (``/root/reference/replay_memory.py``) with a device-resident pool.

Same record semantics as the reference -- a pool of ``cfg.replay_memory_size`` (128) records
``(image, state, feature)``; generator batches pop NON-terminated records, the critic replays
TERMINATED records only, finished / over-length trajectories are replaced by fresh RAW images --
but the images never leave the GPU: the reference downloads ``fake_output`` and re-uploads it
through feed dicts every step (``net.py:325-342``), here the pool is three device tensors and a
step moves only indices.  Randomness comes from an explicit ``torch.Generator`` (host) so runs are
reproducible.
"""
import torch

from exposure_amd.util import STATE_STEP_DIM, STATE_STOPPED_DIM


class SyntheticProvider:
  """Stand-in for FiveKDataProvider / ArtistDataProvider (``data_provider.py:59-69`` crop+flip
  augmentation is replaced by drawing FiveK-shaped synthetic tensors): linear-RAW-like images
  ``U(0,1)**gamma`` of ``size x size x 3``.  float32 by default, the reference's dtype for the
  training pool (``replay_memory.py:16-40``); float16 storage is supported (saturating stores) but an
  untrained policy can push pixels past the fp16 range within one 5-step trajectory."""

  def __init__(self, device, size=64, gamma=2.2, scale=1.0, dtype=torch.float32, seed=0):
    self.device, self.size, self.gamma, self.scale, self.dtype = device, size, gamma, scale, dtype
    self.gen = torch.Generator(device=device).manual_seed(seed)
    self.count = 0

  def get_next_batch(self, batch_size):
    x = torch.rand((batch_size, self.size, self.size, 3), device=self.device, generator=self.gen)
    x = ((x**self.gamma) * self.scale).to(self.dtype)
    feat = torch.arange(self.count, self.count + batch_size, device=self.device, dtype=torch.float32)
    self.count += batch_size
    return x, feat


class ReplayMemory:

  def __init__(self, cfg, fake_provider, real_provider, seed=0):
    self.cfg = cfg
    self.fake_dataset = fake_provider
    self.real_dataset = real_provider
    self.device = fake_provider.device
    self.target_pool_size = cfg.replay_memory_size
    self.rng = torch.Generator().manual_seed(seed)  # host generator: shuffles / keep decisions / z
    self.images = None  # (P, S, S, 3)
    self.states = None  # (P, num_state_dim)
    self.features = None  # (P,)
    self.fill_pool()

  # -- replay_memory.py:54-63
  def get_initial_states(self, batch_size):
    return torch.zeros((batch_size, self.cfg.num_state_dim), dtype=torch.float32, device=self.device)

  def __len__(self):
    return 0 if self.images is None else self.images.shape[0]

  def _append(self, images, states, features):
    if self.images is None:
      self.images, self.states, self.features = images, states, features
    else:
      self.images = torch.cat([self.images, images], dim=0)
      self.states = torch.cat([self.states, states], dim=0)
      self.features = torch.cat([self.features, features], dim=0)

  def _take(self, idx):
    idx = idx.to(self.device)
    return self.images[idx], self.states[idx], self.features[idx]

  def _shuffle(self):
    perm = torch.randperm(len(self), generator=self.rng).to(self.device)
    self.images, self.states, self.features = self.images[perm], self.states[perm], self.features[perm]

  # -- replay_memory.py:65-77
  def fill_pool(self):
    while len(self) < self.target_pool_size:
      batch, features = self.fake_dataset.get_next_batch(self.cfg.batch_size)
      self._append(batch, self.get_initial_states(batch.shape[0]), features)
    self.images = self.images[:self.target_pool_size]
    self.states = self.states[:self.target_pool_size]
    self.features = self.features[:self.target_pool_size]

  def get_noise(self, batch_size):
    """replay_memory.py:177-185: cfg.z_type 'uniform' (U(0, 1), both shipped configs) or 'normal' (N(0, 1))."""
    z_type = getattr(self.cfg, 'z_type', 'uniform')
    if z_type == 'normal':
      return torch.randn((batch_size, self.cfg.z_dim), generator=self.rng).to(self.device)
    assert z_type == 'uniform', 'Unknown noise type: %s' % z_type
    return torch.rand((batch_size, self.cfg.z_dim), generator=self.rng).to(self.device)

  # -- replay_memory.py:235-252: pop NON-terminated records from the shuffled pool
  def get_next_fake_batch(self, batch_size):
    self._shuffle()
    assert batch_size <= len(self)
    got_i, got_s, got_f, have = [], [], [], 0
    while have < batch_size:
      if len(self) == 0:
        self.fill_pool()
      live = (self.states[:, STATE_STOPPED_DIM] != 1).nonzero().flatten().cpu()
      need = batch_size - have
      if live.numel() >= need:
        # records in front of (and including) the need-th live one are consumed, like the pops
        cut = int(live[need - 1]) + 1
        take = live[:need]
      else:
        cut = len(self)
        take = live
      i, s, f = self._take(take)
      got_i.append(i), got_s.append(s), got_f.append(f)
      have += take.numel()
      self.images, self.states, self.features = self.images[cut:], self.states[cut:], self.features[cut:]
    return torch.cat(got_i), torch.cat(got_s), torch.cat(got_f)

  # -- replay_memory.py:254-279: terminated records only (with repetition if there are few)
  def replay_fake_batch(self, batch_size):
    self.fill_pool()
    self._shuffle()
    assert batch_size <= len(self)
    done = (self.states[:, STATE_STOPPED_DIM] > 0).nonzero().flatten().cpu()
    assert done.numel() > 0, 'No terminated states discovered'
    reps = (batch_size + done.numel() - 1) // done.numel()
    idx = done.repeat(reps)[:batch_size]
    return self._take(idx)

  # -- replay_memory.py:199-209
  def replace_memory(self, images, states, features):
    self._shuffle()
    keep = (states[:, STATE_STEP_DIM].cpu() < self.cfg.maximum_trajectory_length) | \
        (torch.rand(states.shape[0], generator=self.rng) < self.cfg.over_length_keep_prob)
    k = keep.to(self.device)
    self._append(images[k], states[k], features[k])
    self.fill_pool()
    self._shuffle()

  # -- feed-dict builders (replay_memory.py:139-185) as plain dicts of device tensors
  def get_feed_dict_and_states(self, batch_size):
    images, states, features = self.get_next_fake_batch(batch_size)
    real, real_feat = self.real_dataset.get_next_batch(batch_size)
    return dict(fake_input=images, fake_input_feature=features, states=states, real_data=real,
                real_data_feature=real_feat, z=self.get_noise(batch_size)), features

  def get_replay_feed_dict(self, batch_size):
    images, _states, features = self.replay_fake_batch(batch_size)
    real, real_feat = self.real_dataset.get_next_batch(batch_size)
    return dict(fake_output=images, fake_output_feature=features, real_data=real, real_data_feature=real_feat)

  def debug(self):
    avg = float(self.states[:, STATE_STEP_DIM].float().mean())
    return '# Replay memory: size %d, avg. traj. %.2f' % (len(self), avg)
