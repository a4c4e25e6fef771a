"""Decomposition sweep of the in-house convolution kernels at the training iteration's shapes (hipGraph replay of 20
launches, us per launch): python tools/r06/conv_sweep.py [fwd|bwd|wrw|all] [--n 64,192]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402

LAYERS = [(6, 64, 32), (14, 64, 32), (17, 64, 32), (32, 32, 64), (64, 16, 128), (128, 8, 256)]


def timeit(fn, reps=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    for _ in range(reps):
      fn()
  graph.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(4):
    graph.replay()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / (4 * reps) * 1e3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('what', nargs='?', default='all')
  ap.add_argument('--n', default='64,192')
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  g = torch.Generator(device=dev).manual_seed(0)
  for n in [int(v) for v in args.n.split(',')]:
    for cin, h, cout in LAYERS:
      x = torch.randn((n, h, h, cin), device=dev, generator=g)
      w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
      b = torch.zeros((cout,), device=dev)
      y = torch.empty((n, h // 2, h // 2, cout), device=dev)
      gy = torch.randn_like(y)
      dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
      gf = 2.0 * n * (h // 2)**2 * cout * 16 * cin / 1e9
      head = 'n=%3d cin=%3d h=%2d cout=%3d (%.2f GF)' % (n, cin, h, cout, gf)
      if args.what in ('fwd', 'all'):
        res = {}
        for tile, nt, sl in [(0, 0, 0)] + [(5, nt, sl) for nt in (1, 2) for sl in (1, 2, 4, 8, 16)] + [(t, 0, 0) for t in (1, 2, 3, 4)]:
          _cabi.conv_tuning(tile, nt, sl)
          res['t%d n%d s%d' % (tile, nt, sl)] = timeit(lambda: _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2))
        _cabi.conv_tuning(0, 0, 0)
        best = min(res, key=res.get)
        print('%s fwd auto %.1f best %s %.1f (%.0f TF) | %s' % (head, res['t0 n0 s0'], best, res[best], gf / res[best] * 1e3,
                                                              ' '.join('%s=%.1f' % kv for kv in res.items())), flush=True)
      if args.what in ('bwd', 'all'):
        res = {}
        for tile, nt, sl in [(0, 0, 0)] + [(0, nt, sl) for nt in (1, 2) for sl in (1, 2, 4, 8, 16)] + [(t, 0, 0) for t in (1, 2, 3)]:
          _cabi.conv_tuning(tile, nt, sl)
          res['n%d s%d' % (nt, sl) if tile == 0 else 't%d' % tile] = timeit(lambda: _cabi.conv4x4s2_bwd_data_mask(gy, w, x, dx, 0.2))
        _cabi.conv_tuning(0, 0, 0)
        best = min(res, key=res.get)
        print('%s bwd auto %.1f best %s %.1f (%.0f TF) | %s' % (head, res['n0 s0'], best, res[best], gf / res[best] * 1e3,
                                                              ' '.join('%s=%.1f' % kv for kv in res.items())), flush=True)
      if args.what in ('wrw', 'all'):
        res = {}
        for sl, parts in [(0, 0), (4, 8), (4, 16), (4, 32), (4, 64), (4, 128), (4, 256), (2, 64), (2, 128), (2, 256)]:
          try:
            _cabi.conv_wrw_tuning(sl, parts)
            res['s%d p%d' % (sl, parts)] = timeit(lambda: _cabi.conv4x4s2_wrw_bias(x, gy, dw, db))
          except Exception as e:  # noqa: BLE001
            res['s%d p%d' % (sl, parts)] = float('nan')
        _cabi.conv_wrw_tuning(0, 0)
        best = min(res, key=lambda k: res[k] if res[k] == res[k] else 1e9)
        print('%s wrw auto %.1f best %s %.1f (%.0f TF) | %s' % (head, res['s0 p0'], best, res[best], gf / res[best] * 1e3,
                                                              ' '.join('%s=%.1f' % kv for kv in res.items())), flush=True)


if __name__ == '__main__':
  main()
