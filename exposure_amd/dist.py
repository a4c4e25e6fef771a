"""Image-sharded data parallelism over RCCL/xGMI (new; the reference is single-process).

The filter kernels, the per-image reductions and the conv nets all treat images as independent
units, so a global minibatch is split image-wise across ranks (one process per GPU) and the ONLY
exchange step is the gradient all-reduce of the network weights:

  G/V step: theta_g 6 123 680 + theta_v 1 221 857 fp32  -> two flat buckets (24.5 MB + 4.9 MB)
  C step:   theta_c 1 216 225 fp32                      -> one flat bucket  (4.9 MB)

xGMI is point-to-point (7 links x ~153 GB/s per GPU) so a ring all-reduce is bound by one link:
t ~ 2 (p-1)/p * bytes / 153 GB/s = 0.34 ms for the 29.4 MB of a G/V step at p = 8, i.e. these
messages are latency-dominated -- hence few, large, flat buffers rather than per-tensor calls: the
gradients live in the flat buffers (``p.grad`` are views), and a bucket's all-reduce is started from
a backward hook the moment its last gradient lands, so it runs under the rest of the backward pass.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def world_size(group=None):
  if not dist.is_available() or not dist.is_initialized():
    return 1
  return dist.get_world_size(group)


def rank(group=None):
  if not dist.is_available() or not dist.is_initialized():
    return 0
  return dist.get_rank(group)


def shard_range(n_global, group=None):
  """Rank r owns images [r*N/p, (r+1)*N/p) of the global minibatch."""
  p, r = world_size(group), rank(group)
  assert n_global % p == 0, 'global batch %d not divisible by world size %d' % (n_global, p)
  per = n_global // p
  return r * per, (r + 1) * per


def shard(t, group=None):
  lo, hi = shard_range(t.shape[0], group)
  return t[lo:hi]


class GlobalBatchRng:
  """Per-step random inputs (dropout masks, the gradient penalty's alpha) that do NOT depend on how the global
  minibatch is partitioned: every rank holds a generator with the SAME seed, draws the tensor
  for the GLOBAL batch -- row i belongs to global image i -- and keeps the rows of its own image shard.  The streams
  advance in lock step on every rank, so a run on p ranks consumes exactly the numbers of the single-process run
  (tests/test_dist_gloo.py: two ranks == one process with nothing but the seed in common).  The redundant draw is
  world_size x (N x 4096) uniform numbers per step: negligible next to one convolution."""

  def __init__(self, seed, device=None):
    self.seed = int(seed)
    self.device = self.gen = None
    if device is not None:
      self._bind(device)

  def _bind(self, device):
    self.device = torch.device(device)
    self.gen = torch.Generator(device=self.device)
    self.gen.manual_seed(self.seed)

  def uniform(self, n_local, tail, group=None, device=None, lead=None):
    """(n_local, *tail) uniform [0, 1) numbers: this rank's rows of the (n_local * world, *tail) global draw
    (``lead``: (lead, n_local, *tail) out of ONE draw of ``lead`` such tensors -- several per-image inputs, one launch).
    ``device``: where the consumer lives.  The generator is created on first use on that device and FOLLOWS the
    module when it moves (``GAN(cfg).cuda()``: drawing on the CPU for inputs on the GPU was a device mismatch); a
    move restarts the stream from the seed, which happens before training, not inside it."""
    if device is not None and (self.device is None or torch.device(device).type != self.device.type or
                               (torch.device(device).index is not None and torch.device(device) != self.device)):
      self._bind(device)
    elif self.gen is None:
      self._bind('cpu')
    p, r = world_size(group), rank(group)
    if lead is not None:
      full = torch.rand((int(lead), n_local * p) + tuple(tail), generator=self.gen, device=self.device)
      return full[:, r * n_local:(r + 1) * n_local]
    full = torch.rand((n_local * p,) + tuple(tail), generator=self.gen, device=self.device)
    return full[r * n_local:(r + 1) * n_local]


def nccl_works_retired(timeout_s=5.0, poll_s=0.01, skip_ids=None):
  """Block until ProcessGroupNCCL's watchdog has RETIRED every collective issued so far (removed it from its work
  list), i.e. until it holds no work whose end event it would still poll.  Needed before a hipGraph capture that
  pulls RCCL's stream in: a watchdog poll (hipEventQuery) of an earlier eager work's event while that stream is
  capturing fails with hipErrorCapturedEvent, the watchdog throws in its own thread and the process aborts
  (profiles/r02_rccl_graph_soak.txt).  The watchdog reaps completed works every ~100 ms; instead of sleeping for
  "long enough", this reads the NCCL flight recorder (every collective is logged with the `retired` flag the
  watchdog sets when it drops the work) and returns True once no un-retired entry is left.  Returns False when the
  recorder is unavailable or disabled (TORCH_FR_BUFFER_SIZE=0) or the deadline passes: the caller then
  must not rely on the drain."""
  import pickle
  import time
  try:
    from torch._C._distributed_c10d import _dump_nccl_trace
  except ImportError:
    return False
  deadline = time.monotonic() + timeout_s
  seen_any = False
  while True:
    try:
      dump = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    except Exception:  # recorder compiled out / API drift: undetermined
      return False
    entries = dump.get('entries', []) if isinstance(dump, dict) else []
    seen_any = seen_any or bool(entries)
    if not entries:
      return False  # nothing recorded although collectives were issued: the recorder is off
    # collectives issued DURING an earlier hipGraph capture are never handed to the watchdog; a torch build whose
    # flight recorder logs them anyway would show them un-retired for ever: the caller passes the record ids it saw
    # while capturing (skip_ids) and they are not waited for
    pending = [e for e in entries if not e.get('retired', False) and
               (skip_ids is None or e.get('record_id', e.get('collective_seq_id')) not in skip_ids)]
    if not pending:
      return True
    if time.monotonic() > deadline:
      return False
    time.sleep(poll_s)


def drain_before_capture(works=(), issued_collectives=True):
  """Make it safe to start a hipGraph capture that will contain RCCL collectives: wait for the given Work handles,
  drain the device, and wait until the watchdog has retired every earlier collective.  Returns True when the drain
  is VERIFIED (flight recorder), False when it could only wait a grace period (EXPO_CAPTURE_GRACE_S, default
  0.35 s = 3.5 watchdog periods) -- callers treat the latter as "not proven"."""
  import os
  import time
  for w in works:
    if w is not None:
      w.wait()
  if torch.cuda.is_available():
    torch.cuda.synchronize()
  if not issued_collectives:
    return True
  if nccl_works_retired():
    return True
  time.sleep(float(os.environ.get('EXPO_CAPTURE_GRACE_S', '0.35')))
  return False


def all_reduce_mean_(t, group=None, force=False):
  if world_size(group) > 1 or (force and dist.is_available() and dist.is_initialized()):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.div_(world_size(group))
  return t


def _dense(p):
  """Non-overlapping and dense: the strides are a permutation of a contiguous layout."""
  dims = sorted(((st, sz) for st, sz in zip(p.stride(), p.size()) if sz > 1))
  expect = 1
  for st, sz in dims:
    if st != expect:
      return False
    expect *= sz
  return True


class _Pending:
  """Handle of an in-flight bucket all-reduce; ``wait()`` finishes it (scale by 1/p)."""

  def __init__(self, bucket, work, scale):
    self.bucket, self.work, self.scale = bucket, work, scale

  def wait(self):
    if self.work is not None:
      self.work.wait()  # stream-level wait: the current stream waits for the collective's stream
      self.work = None
    if self.scale != 1.0:
      self.bucket.flat.mul_(self.scale)
      self.scale = 1.0

  wait_and_scatter = wait  # round-1 name


def _padded(numel):
  return (numel + 3) & ~3


class GradBucket:
  """Flat fp32 gradient storage for a parameter list; every ``p.grad`` IS a view into it.

  autograd accumulates straight into the flat buffer (``loss.backward(inputs=bucket.params)`` after
  ``bucket.zero()``), the bucket is all-reduced in place with ONE collective, and the optimiser reads
  the same memory: no gather / scatter copies (round 1 spent ~60 small copy kernels per bucket each
  way).  A view keeps the parameter's own strides (conv weights are channels_last), so fused
  multi-tensor Adam, which walks parameter and gradient storage in lock step, sees matching layouts.
  ``on_ready`` (optional) is called from inside the backward pass as soon as the LAST gradient of the
  bucket has been accumulated -- the hook GAN uses to start a bucket's all-reduce while the rest of
  the backward is still running."""

  def __init__(self, params, on_ready=None):
    self.params = [p for p in params]
    # every parameter starts on a 16-byte boundary of the flat buffer (padding elements stay zero and ride along in the
    # all-reduce): element-wise kernels over (parameter, gradient) pairs -- expo_adam_step -- keep their float4 path
    self.numel = sum(_padded(p.numel()) for p in self.params)
    self.flat = None
    self.on_ready = on_ready
    self.launched = False  # set by the owner once this step's all-reduce has been issued
    self._arrived = 0
    self._armed = False  # zero() arms the hook counting for ONE backward pass; on_ready / disarm() ends it
    self._hooks = []

  def attach(self):
    """(Re)create the flat buffer on the parameters' device and point every ``p.grad`` into it."""
    dev = self.params[0].device
    self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
    off = 0
    for p in self.params:
      assert p.dtype == torch.float32 and _dense(p), 'bucket parameters must be dense fp32'
      p.grad = torch.as_strided(self.flat, p.size(), p.stride(), off)
      off += _padded(p.numel())
    for h in self._hooks:
      h.remove()
    self._hooks = []
    if self.on_ready is not None:
      for p in self.params:
        self._hooks.append(p.register_post_accumulate_grad_hook(self._arrive))
    return self

  def attached(self):
    if self.flat is None or self.flat.device != self.params[0].device:
      return False
    off = 0
    for p in self.params:
      g = p.grad
      # same storage slot AND the parameter's own layout (fused Adam walks both in lock step): a later
      # .to(memory_format=...) of the module leaves stale views behind -> re-attach
      if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off or g.stride() != p.stride() or \
          g.size() != p.size():
        return False
      off += _padded(p.numel())
    return True

  def zero(self):
    if not self.attached():  # first use, after .to(device), or someone replaced a .grad
      self.attach()
    else:
      self.flat.zero_()
    self._arrived = 0
    self._armed = True
    self.launched = False

  def release(self):
    """Single-rank use: drop the views (``p.grad = None``) so that the next backward pass STORES each gradient
    instead of accumulating it into the flat buffer; ``zero()`` re-attaches whenever a collective is wanted again."""
    for p in self.params:
      p.grad = None
    self._armed = False
    self.launched = False

  def disarm(self):
    """End of the backward pass this bucket was zero()ed for: gradient hooks fired by any OTHER backward (user
    code, a pre-training loop, a test) must not count towards -- or trigger -- this bucket's all-reduce: a
    collective issued by one rank only, or in a different order, is a mismatch or a hang."""
    self._armed = False

  def _arrive(self, _param):
    if not self._armed:
      return
    self._arrived += 1
    if self._arrived == len(self.params) and self.on_ready is not None:
      self._armed = False
      self.on_ready(self)

  def all_reduce_mean(self, group=None, async_op=False, force=False):
    """SUM all-reduce of the flat buffer, scaled by 1/p afterwards (each rank's loss is a mean over its
    local shard of equal size, so the global-batch mean gradient is the rank average).  ``force`` issues
    the collective even in a one-rank group (exercises the RCCL path on one GPU)."""
    assert self.flat is not None, 'zero() / attach() first'
    p = world_size(group)
    work = None
    if p > 1 or (force and dist.is_initialized()):
      work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    pend = _Pending(self, work, 1.0 / p)
    if not async_op:
      pend.wait()
    return pend
