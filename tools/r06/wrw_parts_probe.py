"""Probe: block copies (parts) of the first layers' weight gradient beyond 256 -- their dW is one to three tiles, so the
plan's waves come from the parts alone.  us per launch under hipGraph replay, incl. the reduce launch.
usage: python tools/r06/wrw_parts_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n in (64, 128, 192):
  for cin in (6, 14, 17):
    h, cout = 64, 32
    x = torch.randn((n, h, h, cin), device=dev, generator=g)
    gy = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=g)
    dw = torch.empty((cout, cin, 4, 4), device=dev).contiguous(memory_format=torch.channels_last)
    db = torch.empty((cout,), device=dev)
    res = {}
    for sl, parts in [(0, 0), (4, 128), (4, 170), (4, 256), (4, 384), (4, 512), (4, 768), (4, 1024), (2, 512), (2, 1024)]:
      _cabi.conv_wrw_tuning(sl, parts)
      res['s%d p%d' % (sl, parts)] = timeit(lambda: _cabi.conv4x4s2_wrw_bias(x, gy, dw, db))
    _cabi.conv_wrw_tuning(0, 0)
    print('n=%3d cin=%2d: %s' % (n, cin, '  '.join('%s=%.1f' % kv for kv in res.items())), flush=True)
