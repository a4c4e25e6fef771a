"""Round-5 probe: the in-house 4x4 / stride-2 convolution kernels (csrc/conv_ops.hip) against MIOpen (what
torch.ops.aten.convolution / convolution_backward pick on this box, with the shipped find-db) -- correctness first,
then HIP-event timings of the training step's layer shapes.

  python tools/r05/conv_bench.py [fwd|bwd|wrw|all] [--reps 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from exposure_amd import _cabi  # noqa: E402

STRIDE, PAD, DIL = [2, 2], [1, 1], [1, 1]
# (cin, h, cout) of the three trunks (agent.py:21-32, critics.py:13-35): generator 14, critic 6, value 17 input planes
LAYERS = [(14, 64, 32), (6, 64, 32), (17, 64, 32), (32, 32, 64), (64, 16, 128), (128, 8, 256)]


def make(n, h, cin, cout, dev, seed=0):
  g = torch.Generator(device=dev).manual_seed(seed)
  x = torch.randn((n, h, h, cin), device=dev, generator=g)
  w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * (1.0 / (16 * cin)**0.5)).contiguous(
      memory_format=torch.channels_last)
  b = torch.randn((cout,), device=dev, generator=g) * 0.1
  return x, w, b


def ref_fwd(x, w):
  return torch.ops.aten.convolution(x.permute(0, 3, 1, 2), w, None, STRIDE, PAD, DIL, False, [0, 0], 1).permute(
      0, 2, 3, 1).contiguous()


def timeit(fn, reps, warm=5):
  """us per call of fn, launched from a captured hipGraph (the training step replays graphs: no host launch cost in the
  figure; back-to-back dependent-free launches of one kernel)."""
  for _ in range(warm):
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
  return e0.elapsed_time(e1) / (4 * reps) * 1e3  # us


def check_fwd(dev):
  worst = 0.0
  cases = [(3, 8, 5, 7), (2, 16, 14, 32), (5, 12, 6, 32), (2, 64, 17, 32), (9, 8, 32, 64), (4, 16, 64, 128),
           (64, 8, 128, 256), (1, 2, 3, 1), (7, 4, 4, 33), (8, 64, 14, 32), (64, 32, 32, 64)]
  for (n, h, cin, cout) in cases:
    x, w, b = make(n, h, cin, cout, dev, seed=n + h)
    y = torch.empty((n, h // 2, h // 2, cout), device=dev)
    # float64 reference on the CPU (exact enough to rank both GPU results)
    ref = torch.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, 2, 1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    lib = ref_fwd(x, w)
    err_lib = float((lib.double().cpu() - ref).abs().max()) / scale
    # tile 0: the flat kernel (the library's default) under every (column tiles per block, K slices) it can be given;
    # tiles 1-4: the LDS-tiled shapes
    variants = [('0', nt, sl) for nt in ('1', '2') for sl in ('0', '1', '2', '4', '8', '16')] + [(t, '0', '0') for t in '1234']
    for tile, nt, sl in variants:
      _cabi.conv_tuning(int(tile), int(nt), int(sl))
      errs = []
      for act in (0, 1):
        y.fill_(float('nan'))
        _cabi.conv4x4s2_fwd(x, w, b if act else None, y, act, 0.2)
        want = ref + b.double().cpu() if act else ref
        if act:
          want = torch.where(want > 0, want, want * 0.2)
        err = float((y.double().cpu() - want).abs().max()) / scale
        errs.append(err)
        assert err < 2e-6, ('forward mismatch', n, h, cin, cout, tile, nt, sl, act, err)
      worst = max(worst, max(errs))
    print('fwd check n=%d h=%d cin=%d cout=%d: %d variants OK, worst err %.2e (MIOpen %.2e) of max |y| %.2f' %
          (n, h, cin, cout, len(variants), worst, err_lib, scale))
    _cabi.conv_tuning(0, 0, 0)
  print('fwd check OK, worst %.2e' % worst)


def bench_fwd(dev, reps):
  settings = [('auto', {}), ('nt1', {'EXPO_CONV_NT': '1'}), ('nt2', {'EXPO_CONV_NT': '2'})]
  settings += [('nt1 s%d' % s, {'EXPO_CONV_NT': '1', 'EXPO_CONV_SLICES': str(s)}) for s in (1, 2, 4, 8, 16)]
  settings += [('nt2 s%d' % s, {'EXPO_CONV_NT': '2', 'EXPO_CONV_SLICES': str(s)}) for s in (2, 4, 8)]
  settings += [('tile%d' % t, {'EXPO_CONV_TILE': str(t)}) for t in (1, 2, 3, 4)]
  print('forward, us per launch (graph replay).  MIOpen | MIOpen + bias_lrelu | ours fused by setting')
  tot_lib = tot_ours = 0.0
  for n in (64, 128):
    for (cin, h, cout) in LAYERS:
      x, w, b = make(n, h, cin, cout, dev)
      y = torch.empty((n, h // 2, h // 2, cout), device=dev)
      z = torch.empty_like(y)
      t_lib = timeit(lambda: ref_fwd(x, w), reps)
      t_lib_act = timeit(lambda: _cabi.bias_lrelu_fwd(ref_fwd(x, w), b, z, 0.2), reps)
      res = {}
      for name, env in settings:
        _cabi.conv_tuning(int(env.get('EXPO_CONV_TILE', 0)), int(env.get('EXPO_CONV_NT', 0)), int(env.get('EXPO_CONV_SLICES', 0)))
        res[name] = timeit(lambda: _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2), reps)
      _cabi.conv_tuning(0, 0, 0)
      fl = 2.0 * n * (h // 2)**2 * cout * 16 * cin
      best = min(res, key=res.get)
      print('n=%3d cin=%3d h=%2d cout=%3d  %6.1f | %6.1f | auto %6.1f  best %s %6.1f (%.0f TFLOP/s, x%.2f) | %s' %
            (n, cin, h, cout, t_lib, t_lib_act, res['auto'], best, res[best], fl / res[best] / 1e6, t_lib_act / res[best],
             ' '.join('%s=%.1f' % (k, v) for k, v in res.items())))
      tot_lib += t_lib_act
      tot_ours += res['auto']
  print('sum: MIOpen + bias_lrelu %.1f us, ours fused (auto) %.1f us' % (tot_lib, tot_ours))


def ref_bwd(g, w, n, h, cin):
  x_like = torch.empty((n, cin, h, h), device=g.device, memory_format=torch.channels_last)
  return torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x_like, w, None, STRIDE, PAD, DIL, False, [0, 0], 1,
                                             [True, False, False])[0].permute(0, 2, 3, 1).contiguous()


def ref_wrw(x, g, w):
  return torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, STRIDE, PAD, DIL, False,
                                             [0, 0], 1, [False, True, False])[1]


def clear_env():
  _cabi.conv_tuning(0, 0, 0)


def check_bwd_wrw(dev):
  cases = [(3, 8, 5, 8), (2, 16, 14, 32), (5, 12, 6, 32), (2, 64, 17, 32), (9, 8, 32, 64), (4, 16, 64, 128),
           (16, 8, 128, 256), (7, 4, 4, 36), (8, 64, 14, 32), (32, 32, 32, 64), (3, 4, 3, 4)]
  for (n, h, cin, cout) in cases:
    x, w, b = make(n, h, cin, cout, dev, seed=n + h)
    g = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, None, 2, 1)
    gx_ref, gw_ref = torch.autograd.grad(yd, [xd, wd], g.double().cpu().permute(0, 3, 1, 2))
    gx_ref = gx_ref.permute(0, 2, 3, 1)
    sx, sw = float(gx_ref.abs().max()), float(gw_ref.abs().max())
    lib_x = float((ref_bwd(g, w, n, h, cin).double().cpu() - gx_ref).abs().max()) / sx
    lib_w = float((ref_wrw(x, g, w).double().cpu() - gw_ref).abs().max()) / sw
    worst_x = worst_w = 0.0
    dx = torch.empty((n, h, h, cin), device=dev)
    for nt, sl in [(nt, sl) for nt in ('0', '1', '2') for sl in ('0', '1', '2', '4', '8', '16')]:
      _cabi.conv_tuning(0, int(nt), int(sl))
      dx.fill_(float('nan'))
      _cabi.conv4x4s2_bwd_data(g, w, dx)
      err = float((dx.double().cpu() - gx_ref).abs().max()) / sx
      assert err < 3e-6, ('bwd_data mismatch', n, h, cin, cout, sl, err)
      worst_x = max(worst_x, err)
    dw = torch.empty_like(w)
    if (h // 2) % 2 == 0:
      for sl, parts in ((0, 0), (1, 1), (2, 3), (4, 0), (4, 7), (3, 2)):
        _cabi.conv_wrw_tuning(sl, parts)
        for rep in range(2):
          dw.fill_(float('nan'))
          _cabi.conv4x4s2_wrw(x, g, dw)
          err = float((dw.double().cpu() - gw_ref).abs().max()) / sw
          assert err < 1e-5, ('wrw mismatch', n, h, cin, cout, sl, parts, rep, err)
          worst_w = max(worst_w, err)
      _cabi.conv_wrw_tuning(0, 0)
    clear_env()
    print('wrw check: dw err %.2e (MIOpen %.2e)' % (worst_w, lib_w))
    print('bwd check n=%d h=%d cin=%d cout=%d: dx err %.2e (MIOpen %.2e)' % (n, h, cin, cout, worst_x, lib_x))
  print('bwd/wrw check OK')


def bench_bwd_wrw(dev, reps):
  print('data gradient / weight gradient, us per launch (graph replay): MIOpen | ours by setting')
  tot = {'bwd_lib': 0.0, 'bwd': 0.0, 'wrw_lib': 0.0, 'wrw': 0.0}
  for n in (64, 128):
    for (cin, h, cout) in LAYERS:
      x, w, b = make(n, h, cin, cout, dev)
      g = torch.randn((n, h // 2, h // 2, cout), device=dev)
      dx = torch.empty((n, h, h, cin), device=dev)
      dw = torch.empty_like(w)
      fl = 2.0 * n * (h // 2)**2 * cout * 16 * cin
      clear_env()
      t_bl = timeit(lambda: ref_bwd(g, w, n, h, cin), reps)
      t_wl = timeit(lambda: ref_wrw(x, g, w), reps)
      rb = {}
      for nt, sl in ((0, 0), (1, 0), (2, 0), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (2, 2), (2, 4), (2, 8), (2, 16)):
        _cabi.conv_tuning(0, nt, sl)
        rb['n%ds%d' % (nt, sl)] = timeit(lambda: _cabi.conv4x4s2_bwd_data(g, w, dx), reps)
      rw = {}
      for sl, parts in ((0, 0), (4, 0), (2, 0), (1, 0), (4, 8), (4, 32), (4, 128), (4, 256), (2, 64)):
        _cabi.conv_wrw_tuning(sl, parts)
        rw['s%dp%d' % (sl, parts)] = timeit(lambda: _cabi.conv4x4s2_wrw(x, g, dw), reps)
      _cabi.conv_wrw_tuning(0, 0)
      bw = min(rw, key=rw.get)
      clear_env()
      bb = min(rb, key=rb.get)
      print('n=%3d cin=%3d h=%2d cout=%3d  bwd %6.1f | auto %6.1f best %s %6.1f (%.0f TF) | %s' %
            (n, cin, h, cout, t_bl, rb['n0s0'], bb, rb[bb], fl / rb[bb] / 1e6, ' '.join('%s=%.1f' % kv for kv in rb.items())))
      print('%28s  wrw %6.1f | auto %6.1f best %s %6.1f (%.0f TF) | %s' %
            ('', t_wl, rw['s0p0'], bw, rw[bw], fl / rw[bw] / 1e6, ' '.join('%s=%.1f' % kv for kv in rw.items())))
      tot['wrw'] += rw['s0p0']
      tot['bwd_lib'] += t_bl
      tot['bwd'] += rb['n0s0']
      tot['wrw_lib'] += t_wl
  print('sums (auto): bwd MIOpen %.1f ours %.1f | wrw MIOpen %.1f ours %.1f (each incl. everything the call launches: the zero fill of MIOpen, the reduce launch here)' %
        (tot['bwd_lib'], tot['bwd'], tot['wrw_lib'], tot['wrw']))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('what', nargs='?', default='all')
  ap.add_argument('--reps', type=int, default=50)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  torch.backends.cudnn.benchmark = False
  if args.what in ('fwd', 'all'):
    check_fwd(dev)
    bench_fwd(dev, args.reps)
  if args.what in ('bwd', 'all'):
    check_bwd_wrw(dev)
    bench_bwd_wrw(dev, args.reps)


if __name__ == '__main__':
  main()
