"""Weight-gradient kernel under forced (slices, parts) against MIOpen, per layer (tools/r05/conv_bench.py's timing)."""
import sys
import torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tools/r05')
import conv_bench as cb
from exposure_amd import _cabi

dev = torch.device('cuda:0')
SETS = {0: ((0, 0), (4, 128), (4, 256), (4, 384)), 1: ((0, 0), (4, 256), (4, 512), (4, 768)), 2: ((0, 0), (4, 128), (4, 170), (4, 256)),
        3: ((0, 0), (4, 64), (4, 96), (4, 128)), 4: ((0, 0), (4, 16), (4, 24), (4, 32)), 5: ((0, 0), (4, 4), (4, 6), (4, 8))}
for n in (64, 128):
  for li, (cin, h, cout) in enumerate(cb.LAYERS):
    x, w, b = cb.make(n, h, cin, cout, dev)
    g = torch.randn((n, h // 2, h // 2, cout), device=dev)
    dw = torch.empty_like(w)
    res = {}
    for sl, parts in SETS[li]:
      _cabi.conv_wrw_tuning(sl, parts)
      res['s%dp%d' % (sl, parts)] = cb.timeit(lambda: _cabi.conv4x4s2_wrw(x, g, dw), 20)
    _cabi.conv_wrw_tuning(0, 0)
    t = cb.timeit(lambda: cb.ref_wrw(x, g, w), 20)
    print('n=%d cin=%d: MIOpen %.1f | %s' % (n, cin, t, ' '.join('%s=%.1f' % kv for kv in res.items())))
