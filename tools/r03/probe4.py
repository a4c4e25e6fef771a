"""r03p4: which op launches which kernel in one critic step / one generator step (torch.profiler, eager)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
cfg = make_cfg()
torch.manual_seed(0)
gan = GAN(cfg, device=dev, use_graphs=False)
n = 64
g = torch.Generator(device=dev).manual_seed(1)
real = torch.rand((n, 64, 64, 3), device=dev, generator=g)
fake = torch.rand((n, 64, 64, 3), device=dev, generator=g)
states = torch.zeros((n, 11), device=dev)
z = torch.rand((n, 131), device=dev, generator=g)
for _ in range(3):
  gan.critic_step(real, fake, it=1)
  gan.generator_step(fake.half(), z, states, 0.5, it=1)
torch.cuda.synchronize()
for name, fn in (('critic_step', lambda: gan.critic_step(real, fake, it=1)),
                 ('generator_step', lambda: gan.generator_step(fake.half(), z, states, 0.5, it=1))):
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fn()
    torch.cuda.synchronize()
  evs = prof.events()
  print('=====', name)
  tot = 0
  rows = []
  for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA:
      continue
    ks = [k for k in e.kernels] if hasattr(e, 'kernels') else []
    if not ks:
      continue
    # only leaf ops (those whose children launched nothing are leaves already since kernels attach to the launching op)
    rows.append((e.name, str(e.input_shapes)[:120], [(k.name[:60], k.duration) for k in ks]))
  nk = sum(len(r[2]) for r in rows)
  print('ops with kernels: %d, kernels: %d' % (len(rows), nk))
  # aggregate by op name
  agg = {}
  for name_, shp, ks in rows:
    a = agg.setdefault(name_, [0, 0, 0.0])
    a[0] += 1
    a[1] += len(ks)
    a[2] += sum(k[1] for k in ks)
  for k_, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-60s calls %4d kernels %5d  %9.1f us' % (k_[:60], v[0], v[1], v[2]))
  print('--- convolution-type ops in order')
  for name_, shp, ks in rows:
    if 'conv' in name_.lower():
      print(name_[:40], shp, ' | '.join('%s %.0fus' % k for k in ks))
