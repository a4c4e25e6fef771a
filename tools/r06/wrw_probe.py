"""Probe: weight-gradient timings of layers 2-4 (batch 64 / 192), the main kernel alone (one block copy set, reduce included
in the entry point).  usage: python tools/r06/wrw_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n in (64, 192):
  for cin, h, cout in ((32, 32, 64), (64, 16, 128), (128, 8, 256)):
    x = torch.randn((n, h, h, cin), device=dev, generator=g)
    w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=g)
    dw, db = torch.empty_like(w), torch.empty((cout,), device=dev)
    t = timeit(lambda: _cabi.conv4x4s2_wrw_bias(x, gy, dw, db))
    gf = 2.0 * n * (h // 2)**2 * cout * 16 * cin / 1e9
    print('n=%3d cin=%3d  wrw %.1f us (%.0f TF)' % (n, cin, t, gf / t * 1e3), flush=True)
