"""Probe: the first layers with their constant planes folded (expo_conv4x4s2_fwd_planes) against the row-staged kernel on
the same planes_concat inputs, us per launch under hipGraph replay.   usage: python tools/r06/planes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n, cin in [(64, 6), (128, 6), (192, 6), (64, 14), (128, 17), (64, 17)]:
  img = torch.rand((n, 64, 64, 3), device=dev, generator=g)
  vec = torch.randn((n, cin - 3), device=dev, generator=g)
  x = torch.empty((n, 64, 64, cin), device=dev)
  _cabi.planes_concat(img, vec, x, 0.5)
  w = (torch.randn((32, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
  w2 = w.clone()
  b = torch.zeros((32,), device=dev)
  y, y2 = torch.empty((n, 32, 32, 32), device=dev), torch.empty((n, 32, 32, 32), device=dev)
  t_rows = timeit(lambda: _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2))
  t_fold = timeit(lambda: _cabi.conv4x4s2_fwd_planes(x, w, b, y, 1, 0.2))
  t_rows_pair = timeit(lambda: _cabi.conv4x4s2_fwd_pair((x, w, b, y), (x, w2, b, y2), 1, 0.2))
  t_fold_pair = timeit(lambda: _cabi.conv4x4s2_fwd_planes_pair((x, w, b, y), (x, w2, b, y2), 1, 0.2))
  print('n=%3d cin=%2d: rows %.1f  folded %.1f   pair: rows %.1f  folded %.1f' % (n, cin, t_rows, t_fold, t_rows_pair, t_fold_pair),
        flush=True)
