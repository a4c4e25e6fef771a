"""Probe: forward timings under the FLAT decomposition (forced) of every layer, batch 64 / 192.  usage: python tools/r06/fwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n in (64, 192):
  for cin, h, cout in ((6, 64, 32), (32, 32, 64), (64, 16, 128), (128, 8, 256)):
    x = torch.randn((n, h, h, cin), device=dev, generator=g)
    w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((n, h // 2, h // 2, cout), device=dev)
    res = []
    for nt in (1, 2):
      _cabi.conv_tuning(5, nt, 0)
      res.append(timeit(lambda: _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2)))
    _cabi.conv_tuning(0, 0, 0)
    gf = 2.0 * n * (h // 2)**2 * cout * 16 * cin / 1e9
    print('n=%3d cin=%3d  fwd flat nt1 %.1f us  nt2 %.1f us (%.0f TF)' % (n, cin, res[0], res[1], gf / min(res) * 1e3), flush=True)
