"""Throughput of the kernels outside the headline chain (64x512x512x3 fp16 unless --shape):
dispatch (one-hot select + fused penalty), masked apply (cfg.masking=True), critic stats, penalty.
Prints one JSON object; algorithmic bytes as in DESIGN.md section 3."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exposure_amd import _cabi, synthetic  # noqa: E402


def timeit(fn, reps=20):
  for _ in range(3):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--shape', default='C')
  ap.add_argument('--ids', default='cycle', choices=['cycle', 'sorted'],
                  help='dispatch: filter ids cycling over the 8 filters per image, or the same multiset sorted (light '
                  'filters first, curve filters last)')
  args = ap.parse_args()
  shape = synthetic.SHAPES[args.shape]
  dev = torch.device('cuda:0')
  n = shape[0]
  px = shape[0] * shape[1] * shape[2]
  g = torch.Generator(device=dev).manual_seed(0)
  xs = [(torch.rand(shape, device=dev, generator=g)**2.2).half() for _ in range(3)]  # rotate buffers
  dy = torch.randn(shape, device=dev, generator=g).half()
  outs = [torch.empty_like(xs[0]) for _ in range(3)]
  rng = np.random.default_rng(0)
  res = {}

  def gbps(bytes_per_px, ms):
    return bytes_per_px * px / (ms * 1e-3) / 1e9

  # dispatch: images cycle through all 8 filters
  idl = [i % 8 for i in range(n)]
  if args.ids == 'sorted':
    idl = sorted(idl, key=lambda f: (f in (4, 7), f))
  ids = torch.tensor(idl, device=dev, dtype=torch.int32)
  p24 = torch.zeros((n, 24), device=dev)
  for i in range(n):
    fid = idl[i]
    p24[i, :synthetic.NUM_PARAMS[fid]] = torch.from_numpy(synthetic.make_params(rng, fid, 1)[0]).to(dev)
  pen = torch.empty(n, device=dev)
  dp24 = torch.empty_like(p24)
  dpen = torch.ones(n, device=dev)
  k = [0]

  def nxt():
    k[0] = (k[0] + 1) % 3
    return k[0]

  ms = timeit(lambda: _cabi.dispatch_fwd(ids, xs[nxt()], outs[k[0]], p24, pen))
  res['dispatch_fwd+penalty'] = {'ms': ms, 'GBps': gbps(12, ms)}
  ms = timeit(lambda: _cabi.dispatch_bwd(ids, xs[nxt()], dy, outs[k[0]], p24, dp24, dpen))
  res['dispatch_bwd+penalty'] = {'ms': ms, 'GBps': gbps(18, ms)}
  # masked apply
  mp = torch.from_numpy(np.tanh(rng.standard_normal((n, 6))).astype(np.float32) * 5).to(dev)
  dmp = torch.empty_like(mp)
  for fid in (0, 5, 7):
    p = torch.from_numpy(synthetic.make_params(rng, fid, n)).to(dev)
    dp = torch.empty_like(p)
    ms = timeit(lambda: _cabi.apply_fwd(fid, xs[nxt()], outs[k[0]], p, mp, 1.0, 0.3))
    res['apply_fwd_%s' % synthetic.FILTER_NAMES[fid]] = {'ms': ms, 'GBps': gbps(12, ms)}
    ms = timeit(lambda: _cabi.apply_bwd(fid, xs[nxt()], dy, outs[k[0]], p, dp, mp, dmp, 1.0, 0.3))
    res['apply_bwd_%s' % synthetic.FILTER_NAMES[fid]] = {'ms': ms, 'GBps': gbps(18, ms)}
  stats = torch.empty((n, 3), device=dev)
  ms = timeit(lambda: _cabi.critic_stats(xs[nxt()], stats))
  res['critic_stats'] = {'ms': ms, 'GBps': gbps(6, ms)}
  ms = timeit(lambda: _cabi.overexposure_penalty(xs[nxt()], pen))
  res['overexposure_penalty'] = {'ms': ms, 'GBps': gbps(6, ms)}
  # derivatives of the critic statistics (round 3): J^T g (map), J v (reduction), the second-order term (map)
  g3 = torch.randn((n, 3), device=dev)
  jv = torch.empty((n, 3), device=dev)
  ms = timeit(lambda: _cabi.critic_stats_bwd(xs[nxt()], stats, g3, outs[k[0]]))
  res['critic_stats_bwd'] = {'ms': ms, 'GBps': gbps(12, ms)}
  ms = timeit(lambda: _cabi.critic_stats_jvp(xs[nxt()], stats, dy, jv))
  res['critic_stats_jvp'] = {'ms': ms, 'GBps': gbps(12, ms)}
  ms = timeit(lambda: _cabi.critic_stats_hvp(xs[nxt()], g3, jv, dy, outs[k[0]]))
  res['critic_stats_hvp'] = {'ms': ms, 'GBps': gbps(18, ms)}
  ms = timeit(lambda: _cabi.overexposure_penalty_bwd(xs[nxt()], dpen, outs[k[0]]))
  res['overexposure_penalty_bwd'] = {'ms': ms, 'GBps': gbps(12, ms)}
  # VignetFilter.apply (mask evaluation + lerp in one pass; backward with the 5 mask-parameter gradients)
  vmp = torch.from_numpy(np.tanh(rng.standard_normal((n, 5))).astype(np.float32) * 5).to(dev)
  dvmp = torch.empty_like(vmp)
  ms = timeit(lambda: _cabi.vignet_apply_fwd(xs[nxt()], outs[k[0]], vmp, 1.0, True))
  res['vignet_apply_fwd'] = {'ms': ms, 'GBps': gbps(12, ms)}
  ms = timeit(lambda: _cabi.vignet_apply_bwd(xs[nxt()], dy, outs[k[0]], vmp, dvmp, 1.0, True))
  res['vignet_apply_bwd'] = {'ms': ms, 'GBps': gbps(18, ms)}
  print(json.dumps({'shape': list(shape), 'dtype': 'f16', 'ids': args.ids, 'kernels': res}))


if __name__ == '__main__':
  main()
