"""r03p33: per-filter cost inside the one-pass kernels: 8 steps of the SAME filter, 64x512x512x3 fp16."""
import numpy as np
import torch
from exposure_amd import _cabi, synthetic

dev = torch.device('cuda:0')
shape = synthetic.SHAPES['C']
n = shape[0]
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(shape, device=dev, generator=g) * 0.9 + 0.02).half()
dy = (torch.randn(shape, device=dev, generator=g) * 0.5).half()
y, dx = torch.empty_like(x), torch.empty_like(x)
rng = np.random.default_rng(1)
NP = (1, 1, 3, 1, 8, 1, 1, 24, 2)
names = ('E', 'G', 'W', 'S+', 'T', 'Ct', 'BW', 'C', 'Le')


def timeit(fn, reps=20):
  for _ in range(3):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


for fid in range(9):
  ids = torch.full((n, 8), fid, dtype=torch.int32, device=dev)
  p = torch.zeros((n, 8, 24), device=dev)
  for st in range(8):
    p[:, st, :NP[fid]] = torch.from_numpy(synthetic.make_params(rng, fid, n)).to(dev)
  dp = torch.empty_like(p)
  tf = timeit(lambda: _cabi.chain_fused_fwd(ids, p, x, y))
  tb = timeit(lambda: _cabi.chain_fused_bwd(ids, p, x, dy, dx, dp))
  print('%-3s fused fwd %6.1f us  one-pass bwd %6.1f us  (per step %5.1f / %5.1f)' % (names[fid], tf, tb, tf / 8, tb / 8))
