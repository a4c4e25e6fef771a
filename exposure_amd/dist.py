"""Image-sharded data parallelism over RCCL/xGMI (new; the reference is single-process).

The filter kernels, the per-image reductions and the conv nets all treat images as independent
units, so a global minibatch is split image-wise across ranks (one process per GPU) and the ONLY
exchange step is the gradient all-reduce of the network weights:

  G/V step: theta_g 6 123 680 + theta_v 1 221 857 fp32  -> two flat buckets (24.5 MB + 4.9 MB)
  C step:   theta_c 1 216 225 fp32                      -> one flat bucket  (4.9 MB)

xGMI is point-to-point (7 links x ~153 GB/s per GPU) so a ring all-reduce is bound by one link:
t ~ 2 (p-1)/p * bytes / 153 GB/s = 0.34 ms for the 29.4 MB of a G/V step at p = 8, i.e. these
messages are latency-dominated -- hence few, large, flat buffers (<= 3 collectives per step,
issued async so the second bucket's copy-in overlaps the first's ring) rather than per-tensor
calls.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
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


class _Pending:

  def __init__(self, bucket, work, scale):
    self.bucket, self.work, self.scale = bucket, work, scale

  def wait_and_scatter(self):
    if self.work is not None:
      self.work.wait()
    self.bucket.scatter(self.scale)


class GradBucket:
  """Flat fp32 gradient buffer for a parameter list: one all-reduce per bucket."""

  def __init__(self, params):
    self.params = [p for p in params]
    self.numel = sum(p.numel() for p in self.params)
    self.flat = None

  def _ensure(self, device):
    if self.flat is None or self.flat.device != device:
      self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)

  def gather(self):
    self._ensure(self.params[0].device)
    off = 0
    for p in self.params:
      n = p.numel()
      if p.grad is None:
        self.flat[off:off + n].zero_()
      else:
        self.flat[off:off + n].copy_(p.grad.reshape(-1))
      off += n
    return self.flat

  def scatter(self, scale=1.0):
    off = 0
    for p in self.params:
      n = p.numel()
      g = self.flat[off:off + n].view_as(p)
      if p.grad is None:
        p.grad = (g * scale).clone()
      else:
        p.grad.copy_(g).mul_(scale)
      off += n

  def all_reduce_mean(self, group=None, async_op=False, force=False):
    """SUM all-reduce of the flat buffer, scaled by 1/p on scatter (each rank's loss is a mean over
    its local shard of equal size, so the global-batch mean gradient is the rank average).
    ``force`` issues the collective even in a one-rank group (exercises the RCCL path on one GPU)."""
    flat = self.gather()
    p = world_size(group)
    work = None
    if p > 1 or (force and dist.is_initialized()):
      work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
      if not async_op:
        work.wait()
        work = None
    pend = _Pending(self, work, 1.0 / p)
    if not async_op:
      pend.wait_and_scatter()
    return pend
