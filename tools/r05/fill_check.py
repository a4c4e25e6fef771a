import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/r05')
import exposure_amd  # the shipped MIOpen rankings
import conv_bench as cb
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
for (cin, h, cout) in (cb.LAYERS[0], cb.LAYERS[3]):
  x, w, b = cb.make(64, h, cin, cout, dev); g = torch.randn((64, h // 2, h // 2, cout), device=dev)
  for name, fn in (('wrw', lambda: cb.ref_wrw(x, g, w)), ('bwd', lambda: cb.ref_bwd(g, w, 64, h, cin))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
      fn(); torch.cuda.synchronize()
    ks = [(e.name[:60], round(e.device_time_total, 1)) for e in p.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(cin, name, ks)
