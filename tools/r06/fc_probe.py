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
  # split-K through the library's strided-batch GEMM: S slabs [S][m][128] that the consumer (critic_head_fwd / _bwd) adds
  for S in (4, 8, 16, 32):
    k = 4096 // S
    a3 = flat.view(m, S, k).transpose(0, 1)      # [S][m][k], strides (k, 4096, 1): no copy
    b3 = w.view(128, S, k).permute(1, 2, 0)      # [S][k][128], strides (k, 1, 4096)
    slabs = torch.empty((S, m, 128), device=dev)
    res['bmm S=%d' % S] = timeit(lambda: torch.bmm(a3, b3, out=slabs))
    if S == 8:
      err = float((slabs.sum(0) - flat @ w.t()).abs().max())
      assert err < 1e-2, err
  print('m=%d: ' % m + '  '.join('%s %.1f' % kv for kv in res.items()), flush=True)
