"""Probe: the critic's first FC layer (m x 4096 @ 4096 x 128) through the library GEMM in its possible spellings, us per
call under hipGraph replay.  usage: python tools/r06/fc_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
for m in (64, 128, 192):
  flat = torch.randn((m, 4096), device=dev)
  w = torch.randn((128, 4096), device=dev) * 0.02
  wt = w.t().contiguous()
  b = torch.randn((128,), device=dev)
  dh = torch.randn((m, 128), device=dev)
  out = torch.empty((m, 128), device=dev)
  outt = torch.empty((128, m), device=dev)
  dz = torch.empty((m, 4096), device=dev)
  dw = torch.empty((128, 4096), device=dev)
  res = {
      'addmm(b, flat, w.t())': timeit(lambda: torch.addmm(b, flat, w.t())),
      'mm(flat, w.t(), out)': timeit(lambda: torch.mm(flat, w.t(), out=out)),
      'mm(flat, wt, out)': timeit(lambda: torch.mm(flat, wt, out=out)),
      'mm(w, flat.t(), out)': timeit(lambda: torch.mm(w, flat.t(), out=outt)),
      'linear': timeit(lambda: torch.nn.functional.linear(flat, w, b)),
      'dz = mm(dh, w)': timeit(lambda: torch.mm(dh, w, out=dz)),
      'dw = mm(dh.t(), flat)': timeit(lambda: torch.mm(dh.t(), flat, out=dw)),
  }
  print('m=%d: ' % m + '  '.join('%s %.1f' % kv for kv in res.items()), flush=True)
