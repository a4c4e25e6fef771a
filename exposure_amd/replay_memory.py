"""Replay memory of the trainer (``/root/reference/replay_memory.py``) with a device-resident pool.

Same record semantics as the reference -- a pool of ``cfg.replay_memory_size`` (128) records
``(image, state, feature)``; generator batches pop NON-terminated records, the critic replays
TERMINATED records only, finished / over-length trajectories are replaced by fresh RAW images --
but the images never leave the GPU: the reference downloads ``fake_output`` and re-uploads it
through feed dicts every step (``net.py:325-342``), here the pool is three device tensors and a
step moves only indices.  Randomness comes from an explicit ``torch.Generator`` (host) so runs are
reproducible.

Round 4: no host synchronisation in the training loop.  Round 3's pool re-gathered every record at every shuffle and
asked the DEVICE which records had terminated (``nonzero().cpu()``, boolean-mask indexing): 8+ host syncs and ~150
eager launches per iteration, 1.8 ms of a 14.2 ms iteration (``tools/r04/memory_cost.py``).  Now

* the records sit in fixed physical SLOTS of three device buffers (capacity = pool size + one batch); the pool's ORDER is
  a host array of slot ids -- shuffling, popping, cutting and truncating touch no device memory;
* the two state fields the pool's decisions read (``stopped``, ``step``) are mirrored on the host.  They are known
  there without asking the device: fresh records start at 0, and an agent step maps ``step -> step + 1`` and
  ``stopped -> |step + 1 - cfg.test_steps| < 1e-4`` (``agent.py:207-238``) whatever the networks compute.
  ``replace_memory(..., advanced=True)`` (the training loops) applies that rule to the batch popped last;
  without it the two columns are read back from the device (one sync: arbitrary caller-made states);
* index vectors and the selection noise reach the device through a ring of PINNED staging buffers
  (``non_blocking`` copies: a pageable host-to-device copy would wait for the stream);
* what remains on the device per call is the gather / scatter of the batch itself (3 launches each).

The sequence of random decisions (shuffles, keep draws, noise) is the one of the round-3 implementation, draw for draw:
``tests/test_replay_and_loop.py`` runs both side by side and requires identical batches.

Round 6: ~100 of an iteration's 451 launches sat BETWEEN the step graphs (index gathers, the staging copies into the
graphs' static inputs, the synthetic providers' rand / pow / cast, zero states, row selections: ~6 us each on an idle queue).
The feed builders can now hand out :class:`PoolRows` -- (pool tensor, device index vector) -- which the step GATHERS
STRAIGHT INTO its graph's static input (``GAN._replay``: one ``index_select(out=)`` instead of gather + copy);
``replace_memory`` scatters the whole batch with one index vector (dropped rows go to a trash slot: no row selection), fresh
records are written in place (``index_fill_`` for the zero states); :class:`ResidentProvider` keeps the synthetic data set
in HBM and serves batches as views (inputs resident in HBM: nothing to launch).  The host-side decisions are unchanged.
"""
import numpy as np
import torch

from .util import STATE_STEP_DIM, STATE_STOPPED_DIM


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


class ResidentProvider:
  """The synthetic data set of :class:`SyntheticProvider` generated ONCE and kept in HBM (``count`` images: 4 096 proxies of
  64 x 64 x 3 are 100 MB in fp16 -- MIT-Adobe FiveK itself at this resolution is 123 MB); batches are consecutive VIEWS of
  it, epoch after epoch (``data_provider.py:59-69`` walks a shuffled epoch the same way; its crop / flip augmentation has no
  synthetic counterpart).  ``get_next_batch`` launches nothing."""

  def __init__(self, device, size=64, gamma=2.2, scale=1.0, dtype=torch.float32, seed=0, count=4096):
    self.device, self.size, self.dtype, self.count = device, size, dtype, int(count)
    src = SyntheticProvider(device, size=size, gamma=gamma, scale=scale, dtype=dtype, seed=seed)
    parts = [src.get_next_batch(min(256, self.count - at))[0] for at in range(0, self.count, 256)]
    self.images = torch.cat(parts)
    self.features = torch.arange(self.count, device=device, dtype=torch.float32)
    self.at = 0

  def next_rows(self, batch_size):
    """The first row of the next batch (host only: the batch is ``images[lo:lo + batch_size]``)."""
    assert batch_size <= self.count
    if self.at + batch_size > self.count:
      self.at = 0  # next epoch
    lo = self.at
    self.at += batch_size
    return lo

  def get_next_batch(self, batch_size):
    lo = self.next_rows(batch_size)
    return self.images[lo:lo + batch_size], self.features[lo:lo + batch_size]


class PoolRows:
  """Rows ``idx`` (a device int64 vector) of a pool tensor, gathered where they are consumed: ``into(dst)`` writes them
  straight into a step graph's static input, ``materialize()`` returns them as a tensor.  Valid until the pool is
  written again (``ReplayMemory`` counts its writes; a stale gather is an error, not a silent wrong batch)."""

  def __init__(self, pool, idx, owner=None):
    self.pool, self.idx, self.owner = pool, idx, owner
    self.stamp = owner._writes if owner is not None else None
    self.shape = (idx.shape[0],) + tuple(pool.shape[1:])
    self.dtype, self.device = pool.dtype, pool.device

  @property
  def is_cuda(self):
    return self.pool.is_cuda

  def dim(self):
    return self.pool.dim()

  def _check(self):
    assert self.owner is None or self.owner._writes == self.stamp, 'PoolRows used after the pool was written'

  def materialize(self):
    self._check()
    return self.pool.index_select(0, self.idx)

  def into(self, dst):
    self._check()
    assert tuple(dst.shape) == self.shape and dst.dtype == self.dtype and dst.is_contiguous()
    torch.index_select(self.pool, 0, self.idx, out=dst)
    return dst


class HostStaged:
  """A PINNED host tensor on its way to the device: ``into(dst)`` is the one asynchronous copy that puts it into a step
  graph's static input, ``materialize()`` a fresh device tensor.  ``ring`` / ``key``: the staging ring the buffer came
  from, told when the copy has been issued (the buffer is reused only after that copy completed)."""

  def __init__(self, host, device, ring=None, key=None):
    self.host, self.device, self.ring, self.key = host, torch.device(device), ring, key
    self.shape, self.dtype = tuple(host.shape), host.dtype

  def _issued(self):
    if self.ring is not None:
      self.ring.mark(self.key)

  def materialize(self):
    out = self.host.to(self.device, non_blocking=True)
    self._issued()
    return out

  def into(self, dst):
    assert tuple(dst.shape) == self.shape and dst.dtype == self.dtype
    dst.copy_(self.host, non_blocking=True)
    self._issued()
    return dst


def materialize(t):
  """A tensor for a tensor, a :class:`PoolRows` or a :class:`HostStaged`."""
  return t.materialize() if isinstance(t, (PoolRows, HostStaged)) else t


class IterationPlan:
  """``ReplayMemory.plan_iteration``'s result: host index vectors (int64) and the noise (float32)."""

  def __init__(self, g_slots, z, g_scatter, fresh_dst, fresh_src, c_slots, real_rows):
    self.g_slots, self.z, self.g_scatter = g_slots, z, g_scatter
    self.fresh_dst, self.fresh_src, self.c_slots, self.real_rows = fresh_dst, fresh_src, c_slots, real_rows

  def int64_fields(self):
    """The index vectors in record order: G gather, G scatter, fresh destination / source, then per critic step the
    replayed slots and the real rows."""
    out = [self.g_slots, self.g_scatter, self.fresh_dst, self.fresh_src]
    for a, b in zip(self.c_slots, self.real_rows):
      out += [a, b]
    return out


class _PinnedRing:
  """Host staging buffers for small host -> device transfers that must not wait for the stream.  A buffer is reused
  only after the copy that read it has completed (its event; in steady state that was many calls ago)."""

  def __init__(self, device, slots=64):
    self.device = torch.device(device)
    self.cuda = self.device.type == 'cuda'
    self.slots, self.bufs, self.events, self.at = slots, {}, {}, 0

  def take(self, dtype, numel):
    """A pinned buffer of ``numel`` elements whose last copy (if any) has completed -> (view, key); the caller fills it,
    issues its copy and calls ``mark(key)``."""
    key = (dtype, self.at % self.slots)
    self.at += 1
    buf = self.bufs.get(key)
    if buf is None or buf.numel() < numel:
      buf = torch.empty(max(numel, 256), dtype=dtype)
      if self.cuda:
        buf = buf.pin_memory()
      self.bufs[key] = buf
    ev = self.events.pop(key, None)
    if ev is not None:
      ev.synchronize()
    return buf[:numel], key

  def mark(self, key):
    if self.cuda:
      ev = torch.cuda.Event()
      ev.record()
      self.events[key] = ev

  def put(self, host_tensor):
    """Device copy of a (small) host tensor; asynchronous on a GPU."""
    if not self.cuda:
      return host_tensor.clone()
    view, key = self.take(host_tensor.dtype, host_tensor.numel())
    view = view.view(host_tensor.shape)
    view.copy_(host_tensor)
    out = view.to(self.device, non_blocking=True)
    self.mark(key)
    return out


class ReplayMemory:

  def __init__(self, cfg, fake_provider, real_provider, seed=0):
    self.cfg = cfg
    self.fake_dataset = fake_provider
    self.real_dataset = real_provider
    self.device = fake_provider.device
    self.target_pool_size = cfg.replay_memory_size
    self.rng = torch.Generator().manual_seed(seed)  # host generator: shuffles / keep decisions / z
    self._ring = _PinnedRing(self.device)
    self._cap = 0
    self._img = self._st = self._ft = None  # physical slots: (cap, S, S, 3), (cap, D), (cap,)
    self._order = np.zeros((0,), dtype=np.int64)  # logical pool: slot ids, front = next to pop
    self._free = []
    self._h_stopped = np.zeros((0,), dtype=np.float64)  # host mirrors of the two fields the pool reads
    self._h_step = np.zeros((0,), dtype=np.float64)
    self._popped = None  # host (step, stopped) of the batch popped last (replace_memory(advanced=True))
    self._writes = 0  # device writes to the pool so far (PoolRows validity)
    self.fill_pool()

  # -- replay_memory.py:54-63
  def get_initial_states(self, batch_size):
    return torch.zeros((batch_size, self.cfg.num_state_dim), dtype=torch.float32, device=self.device)

  def __len__(self):
    return int(self._order.size)

  # ---- the pool in logical order (tests, debugging; a gather per access)
  def _idx(self, slots):
    return self._ring.put(torch.from_numpy(np.ascontiguousarray(slots, dtype=np.int64)))

  @property
  def images(self):
    return self._img.index_select(0, self._idx(self._order))

  @property
  def states(self):
    return self._st.index_select(0, self._idx(self._order))

  @property
  def features(self):
    return self._ft.index_select(0, self._idx(self._order))

  # ---- slots
  def _ensure_capacity(self, like_images, state_dim, like_features, need):
    """Buffers of ``cap`` slots + ONE trash row behind them (the scatter target of rows that are not kept)."""
    if self._img is not None and self._cap >= need:
      return
    cap = max(need, self.target_pool_size + max(int(self.cfg.batch_size), like_images.shape[0]))
    img = torch.zeros((cap + 1,) + tuple(like_images.shape[1:]), dtype=like_images.dtype, device=self.device)
    st = torch.zeros((cap + 1, state_dim), dtype=torch.float32, device=self.device)
    ft = torch.zeros((cap + 1,) + tuple(like_features.shape[1:]), dtype=like_features.dtype, device=self.device)
    if self._img is not None:  # grow (rare: a caller appending more than one batch)
      img[:self._cap], st[:self._cap], ft[:self._cap] = self._img[:self._cap], self._st[:self._cap], self._ft[:self._cap]
    self._free += list(range(self._cap, cap))
    self._h_stopped = np.concatenate([self._h_stopped, np.zeros(cap - self._cap)])
    self._h_step = np.concatenate([self._h_step, np.zeros(cap - self._cap)])
    self._img, self._st, self._ft, self._cap = img, st, ft, cap
    self._writes += 1

  def _host_append(self, total, rows, h_step, h_stopped):
    """The host half of an append: slots for the kept rows (``rows`` None: all ``total``), mirrors, logical order.
    -> the scatter vector of the ``total`` given rows (a host array: kept rows -> their slots, the others -> the trash row)."""
    n = total if rows is None else int(len(rows))
    assert self._img is not None and len(self) + n <= self._cap
    slots = np.array([self._free.pop() for _ in range(n)], dtype=np.int64)
    if rows is None or n == total:
      scatter = slots
    else:
      scatter = np.full((total,), self._cap, dtype=np.int64)  # the trash row
      scatter[np.asarray(rows)] = slots
    self._h_step[slots] = h_step
    self._h_stopped[slots] = h_stopped
    self._order = np.concatenate([self._order, slots])
    return scatter

  def _append(self, images, states, features, h_step, h_stopped, rows=None):
    """Append records at the back: all rows of the given tensors, or only the rows ``rows`` (a host index array) -- the
    others are scattered to the trash row, so the batch is written with ONE index vector and no row selection.
    ``states`` None: fresh records (zero states, ``replay_memory.py:54-63``), written as a fill."""
    total = images.shape[0]
    n = total if rows is None else int(len(rows))
    if n == 0:
      return
    state_dim = self.cfg.num_state_dim if states is None else states.shape[1]
    self._ensure_capacity(images, state_dim, features, len(self) + n)
    dst = self._idx(self._host_append(total, rows, h_step, h_stopped))
    self._writes += 1
    self._img.index_copy_(0, dst, images.to(self._img.dtype))
    if states is None:
      self._st.index_fill_(0, dst, 0.0)
    else:
      self._st.index_copy_(0, dst, states.to(self._st.dtype))
    self._ft.index_copy_(0, dst, features.to(self._ft.dtype))

  def _drop(self, positions_kept):
    """Keep the logical positions given (a slice result); the others' slots become free."""
    keep = self._order[positions_kept]
    gone = np.setdiff1d(self._order, keep, assume_unique=True)
    self._free += gone.tolist()
    self._order = keep

  def _take(self, slots, lazy=False):
    idx = self._idx(slots)
    if lazy:
      return PoolRows(self._img, idx, self), PoolRows(self._st, idx, self), PoolRows(self._ft, idx, self)
    return self._img.index_select(0, idx), self._st.index_select(0, idx), self._ft.index_select(0, idx)

  def _shuffle(self):
    perm = torch.randperm(len(self), generator=self.rng).numpy()
    self._order = self._order[perm]

  # -- replay_memory.py:65-77
  def fill_pool(self):
    # (the reference appends whole batches and cuts the pool back to its target size: the tail of the last batch is
    # dropped -- here it is not written in the first place; the provider still advances by a whole batch)
    while len(self) < self.target_pool_size:
      batch, features = self.fake_dataset.get_next_batch(self.cfg.batch_size)
      k = min(batch.shape[0], self.target_pool_size - len(self))
      self._append(batch[:k], None, features[:k], 0.0, 0.0)

  def _host_noise(self, batch_size):
    z_type = getattr(self.cfg, 'z_type', 'uniform')
    if z_type == 'normal':
      return torch.randn((batch_size, self.cfg.z_dim), generator=self.rng)
    assert z_type == 'uniform', 'Unknown noise type: %s' % z_type
    return torch.rand((batch_size, self.cfg.z_dim), generator=self.rng)

  def get_noise(self, batch_size):
    """replay_memory.py:177-185: cfg.z_type 'uniform' (U(0, 1), both shipped configs) or 'normal' (N(0, 1))."""
    return self._ring.put(self._host_noise(batch_size))

  # -- replay_memory.py:235-252: pop NON-terminated records from the shuffled pool
  def get_next_fake_batch(self, batch_size, lazy=False):
    """``lazy``: the batch as :class:`PoolRows` (gathered by its consumer) when one pass over the pool yields it; the
    rare multi-pass case refills the pool in between and gathers at once, as before."""
    self._shuffle()
    assert batch_size <= len(self)
    got, taken, have = [], [], 0
    while have < batch_size:
      if len(self) == 0:
        self.fill_pool()
      live = np.nonzero(self._h_stopped[self._order] != 1)[0]
      need = batch_size - have
      if live.size >= need:
        # records in front of (and including) the need-th live one are consumed, like the pops
        cut = int(live[need - 1]) + 1
        take = live[:need]
      else:
        cut = len(self)
        take = live
      slots = self._order[take]
      # (an eager gather happens BEFORE the slots are released: a refill may reuse them; a lazy one is only handed out
      # when no refill follows)
      got.append(self._take(slots, lazy=lazy and not got and take.size == need))
      taken.append(slots)
      have += take.size
      self._drop(slice(cut, None))
    slots = np.concatenate(taken)
    self._popped = (self._h_step[slots].copy(), self._h_stopped[slots].copy())
    if len(got) == 1:
      return got[0]
    return tuple(torch.cat([g[k] for g in got]) for k in range(3))

  # -- replay_memory.py:254-279: terminated records only (with repetition if there are few)
  def replay_fake_batch(self, batch_size, lazy=False):
    self.fill_pool()
    self._shuffle()
    assert batch_size <= len(self)
    done = self._order[self._h_stopped[self._order] > 0]
    assert done.size > 0, 'No terminated states discovered'
    reps = (batch_size + done.size - 1) // done.size
    return self._take(np.tile(done, reps)[:batch_size], lazy=lazy)

  # -- replay_memory.py:199-209
  def replace_memory(self, images, states, features, advanced=False):
    """``advanced=True``: ``states`` are the agent's update (``agent.py:207-238``) of the batch the last
    ``get_next_fake_batch`` returned, in the same order -- the host then KNOWS the two fields it needs
    (step + 1; stopped iff step + 1 == cfg.test_steps) and no device read-back happens.  Otherwise they are read from
    ``states`` (one host sync)."""
    features = materialize(features)  # (rows of the pool itself, popped by get_next_fake_batch: read before any write)
    host = None
    if not advanced:
      host = states[:, [STATE_STOPPED_DIM, STATE_STEP_DIM]].detach().to('cpu', torch.float64).numpy()
    rows, h_step, h_stopped = self._host_replace(states.shape[0], host)
    self._append(images, states, features, h_step, h_stopped, rows=rows)
    self.fill_pool()
    self._shuffle()

  def _host_replace(self, n, host=None):
    """The decisions of ``replace_memory`` (replay_memory.py:199-209): shuffle, the records' new (step, stopped) -- from
    the agent's update rule (``host`` None) or read back -- and the keep draw.  -> (kept rows, their step, their stopped)."""
    self._shuffle()
    if host is None:
      assert self._popped is not None and self._popped[0].shape[0] == n, 'advanced=True follows get_next_fake_batch'
      old_step, _old_stopped = self._popped
      h_step = old_step + 1.0
      h_stopped = (np.abs(old_step + 1.0 - float(self.cfg.test_steps)) < 1e-4).astype(np.float64)
    else:
      h_stopped, h_step = host[:, 0], host[:, 1]
    keep = torch.from_numpy(h_step < self.cfg.maximum_trajectory_length) | \
        (torch.rand(n, generator=self.rng) < self.cfg.over_length_keep_prob)
    rows = np.nonzero(keep.numpy())[0]
    return rows, h_step[rows], h_stopped[rows]

  # ---- one training iteration decided ahead (round 6) ------------------------------------------------------------------
  def plan_iteration(self, batch_size, citers):
    """EVERY pool decision of one regular training iteration (net.py:329-365) -- the generator batch's pop, the replace
    of its results (``advanced``: the host knows the new step / stopped fields without the device), the refill with fresh
    records, ``citers`` critic replays, the providers' batches, the selection noise -- made NOW, on the host, in exactly
    the order (and with exactly the random draws) of

        get_feed_dict_and_states -> [G step] -> replace_memory(advanced=True) -> citers x get_replay_feed_dict -> [C step]

    so that the trainer can run the whole iteration as ONE hipGraph replay fed by ONE host-to-device copy of the plan
    (``GAN.train_iteration``): nothing the host decides depends on what the device computes.  The pool's host state
    (order, free slots, mirrors) is advanced here; the device writes happen inside the graph.  -> :class:`IterationPlan`,
    or None (nothing consumed) when the iteration needs the step-by-step path: providers that are not resident in HBM,
    a pop that would need a refill in the middle, buffers that would have to grow."""
    fd, rd = self.fake_dataset, self.real_dataset
    if not (isinstance(fd, ResidentProvider) and isinstance(rd, ResidentProvider)) or self._img is None:
      return None
    pool_batch = int(self.cfg.batch_size)
    if fd.images.dtype != self._img.dtype or max(batch_size, pool_batch) > min(fd.count, rd.count):
      return None
    if self._cap < self.target_pool_size + batch_size or len(self) != self.target_pool_size:
      return None
    if int(np.count_nonzero(self._h_stopped[self._order] != 1)) < batch_size:
      return None  # the pop would run the pool dry and refill in the middle (get_next_fake_batch's second pass)
    trash = self._cap
    # -- get_feed_dict_and_states: pop, the real batch (drawn like the reference does, unused by the G step), the noise
    self._shuffle()
    live = np.nonzero(self._h_stopped[self._order] != 1)[0]
    g_slots = self._order[live[:batch_size]].copy()
    self._drop(slice(int(live[batch_size - 1]) + 1, None))
    self._popped = (self._h_step[g_slots].copy(), self._h_stopped[g_slots].copy())
    rd.next_rows(batch_size)
    z = self._host_noise(batch_size)
    # -- replace_memory(advanced=True)
    rows, h_step, h_stopped = self._host_replace(batch_size)
    g_scatter = self._host_append(batch_size, rows, h_step, h_stopped) if len(rows) else np.full((batch_size,), trash, np.int64)
    # -- fill_pool: whole provider batches, of which only the rows that fit are written
    fresh_dst = np.full((self.target_pool_size,), trash, dtype=np.int64)
    fresh_src = np.zeros((self.target_pool_size,), dtype=np.int64)
    at = 0
    while len(self) < self.target_pool_size:
      lo = fd.next_rows(pool_batch)
      k = min(pool_batch, self.target_pool_size - len(self))
      fresh_dst[at:at + k] = self._host_append(k, None, 0.0, 0.0)
      fresh_src[at:at + k] = np.arange(lo, lo + k)
      at += k
    self._shuffle()
    # -- citers x get_replay_feed_dict (fill_pool is a no-op: the pool is full)
    c_slots, real_rows = [], []
    for _ in range(citers):
      self._shuffle()
      done = self._order[self._h_stopped[self._order] > 0]
      assert done.size > 0, 'No terminated states discovered'
      reps = (batch_size + done.size - 1) // done.size
      c_slots.append(np.tile(done, reps)[:batch_size])
      lo = rd.next_rows(batch_size)
      real_rows.append(np.arange(lo, lo + batch_size))
    self._writes += 1
    return IterationPlan(g_slots, z, g_scatter, fresh_dst, fresh_src, c_slots, real_rows)

  # the device half of a planned iteration: what ``GAN``'s iteration graph executes with the plan's index vectors on the
  # device (also callable eagerly: tests/test_replay_and_loop.py runs a plan through them on the CPU)
  def planned_generator_batch(self, g_slots):
    return self._img.index_select(0, g_slots), self._st.index_select(0, g_slots), self._ft.index_select(0, g_slots)

  def planned_commit(self, g_scatter, fake_output, new_states, features, fresh_dst, fresh_src):
    """Results of the G step into their slots (dropped rows -> trash row), then the fresh records."""
    self._img.index_copy_(0, g_scatter, fake_output.to(self._img.dtype))
    self._st.index_copy_(0, g_scatter, new_states.to(self._st.dtype))
    self._ft.index_copy_(0, g_scatter, features)
    fd = self.fake_dataset
    self._img.index_copy_(0, fresh_dst, fd.images.index_select(0, fresh_src))
    self._st.index_fill_(0, fresh_dst, 0.0)
    self._ft.index_copy_(0, fresh_dst, fd.features.index_select(0, fresh_src))

  def planned_critic_batch(self, c_slots, real_rows, lazy=False):
    """-> (real_data, fake_output) of one critic step; ``lazy``: as :class:`PoolRows` (the hand-scheduled critic update
    reads the rows straight out of the data set and the pool: ``expo_gp_inputs_rows``)."""
    if lazy:
      return PoolRows(self.real_dataset.images, real_rows), PoolRows(self._img, c_slots)
    return self.real_dataset.images.index_select(0, real_rows), self._img.index_select(0, c_slots)

  def check_host_mirror(self):
    """Debug / tests: the host mirrors equal the device states (synchronises)."""
    st = self.states.detach().to('cpu', torch.float64).numpy()
    return bool(np.array_equal(st[:, STATE_STOPPED_DIM], self._h_stopped[self._order]) and
                np.array_equal(st[:, STATE_STEP_DIM], self._h_step[self._order]))

  # -- feed-dict builders (replay_memory.py:139-185) as plain dicts of device tensors
  def get_feed_dict_and_states(self, batch_size, lazy=False):
    """``lazy``: ``fake_input`` / ``states`` / the features as :class:`PoolRows` (``GAN.generator_step`` gathers them
    straight into its graph's inputs; ``replace_memory`` accepts the features as they are)."""
    images, states, features = self.get_next_fake_batch(batch_size, lazy=lazy)
    real, real_feat = self.real_dataset.get_next_batch(batch_size)
    return dict(fake_input=images, fake_input_feature=features, states=states, real_data=real,
                real_data_feature=real_feat, z=self.get_noise(batch_size)), features

  def get_replay_feed_dict(self, batch_size, lazy=False):
    images, _states, features = self.replay_fake_batch(batch_size, lazy=lazy)
    real, real_feat = self.real_dataset.get_next_batch(batch_size)
    return dict(fake_output=images, fake_output_feature=features, real_data=real, real_data_feature=real_feat)

  def debug(self):
    avg = float(self._h_step[self._order].mean())
    return '# Replay memory: size %d, avg. traj. %.2f' % (len(self), avg)
