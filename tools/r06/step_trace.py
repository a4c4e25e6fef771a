"""Ordered kernel list (name, microseconds) of one eager G / V step and one eager critic update (torch profiler):
which launches a step consists of, in issue order.  usage: python tools/r06/step_trace.py [g|c]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg  # noqa: E402
from exposure_amd.gan import GAN  # noqa: E402
from tests.test_oracle_nets import make_batch  # noqa: E402


def trace(fn):
  from torch.profiler import ProfilerActivity, profile
  torch.cuda.synchronize()
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn()
    torch.cuda.synchronize()
  ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
  ev.sort(key=lambda e: e.time_range.start)
  return [(e.name, e.time_range.end - e.time_range.start) for e in ev]


def main():
  which = sys.argv[1] if len(sys.argv) > 1 else 'gc'
  dev = torch.device('cuda:0')
  n = 64
  fake_input, real, states, z, masks, alpha = make_batch(n, 1)
  t = lambda a: torch.from_numpy(a).to(dev)
  real_t, fake_t, alpha_t, zt, st = t(real).half(), t(fake_input).half(), t(alpha), t(z), t(states)
  torch.manual_seed(0)
  gan = GAN(make_cfg(), device=dev, use_graphs=False)
  steps = {'g': lambda: gan.generator_step(fake_t, zt, st, 0.3, it=1),
           'c': lambda: gan.critic_step(real_t, fake_t, it=1, alpha=alpha_t)}
  for key in which:
    for _ in range(3):
      steps[key]()
    rows = trace(steps[key])
    print('== %s step: %d launches, %.1f us of kernel time' % (key, len(rows), sum(r[1] for r in rows)))
    for name, us in rows:
      print('%8.1f  %s' % (us, name[:130]))


if __name__ == '__main__':
  main()
