"""Probe: the first layers' forward / weight gradient with Cin = 6 / 14 / 17 against the same shapes with Cin padded to a
multiple of 4 (the single-load chunk path).  usage: python tools/r06/first_layer_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n in (64, 128, 192):
  for cin in (6, 8, 14, 16, 17, 20):
    h, cout = 64, 32
    x = torch.randn((n, h, h, cin), device=dev, generator=g)
    w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((n, h // 2, h // 2, cout), device=dev)
    gy = torch.randn_like(y)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    f = timeit(lambda: _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2))
    r = timeit(lambda: _cabi.conv4x4s2_wrw_bias(x, gy, dw, db))
    print('n=%3d cin=%2d  fwd %.1f us  wrw %.1f us' % (n, cin, f, r), flush=True)
