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


def per_image_generator(seed, global_index, device):
  """RNG stream keyed by the GLOBAL image index so dropout / alpha / z do not depend on how the
  batch is partitioned."""
  g = torch.Generator(device=device)
  g.manual_seed(int(seed) * 1000003 + int(global_index))
  return g


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
    self.numel = sum(p.numel() for p in self.params)
    self.flat = None
    self.on_ready = on_ready
    self.launched = False  # set by the owner once this step's all-reduce has been issued
    self._arrived = 0
    self._hooks = []

  def attach(self):
    """(Re)create the flat buffer on the parameters' device and point every ``p.grad`` into it."""
    dev = self.params[0].device
    self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
    off = 0
    for p in self.params:
      assert p.dtype == torch.float32 and _dense(p), 'bucket parameters must be dense fp32'
      p.grad = torch.as_strided(self.flat, p.size(), p.stride(), off)
      off += p.numel()
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
      if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off:
        return False
      off += p.numel()
    return True

  def zero(self):
    if not self.attached():  # first use, after .to(device), or someone replaced a .grad
      self.attach()
    else:
      self.flat.zero_()
    self._arrived = 0
    self.launched = False

  def _arrive(self, _param):
    self._arrived += 1
    if self._arrived == len(self.params) and self.on_ready is not None:
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
