"""Per-step timing and launch counts of the training iteration's two steps (hipGraph replay): the critic update
hand-scheduled (exposure_amd/critic_direct.py) vs through autograd, and the G / V step.  usage: python tools/r06/step_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg  # noqa: E402
from exposure_amd.gan import GAN  # noqa: E402
from tests.test_oracle_nets import make_batch  # noqa: E402


def count_launches(fn):
  from torch.profiler import ProfilerActivity, profile
  torch.cuda.synchronize()
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn()
    torch.cuda.synchronize()
  names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
  return len(names), names


def main():
  dev = torch.device('cuda:0')
  n = 64
  fake_input, real, states, z, masks, alpha = make_batch(n, 1)
  t = lambda a: torch.from_numpy(a).to(dev)
  real_t, fake_t, alpha_t = t(real).half(), t(fake_input).half(), t(alpha)
  for direct in (True, False):
    torch.manual_seed(0)
    gan = GAN(make_cfg(), device=dev, use_graphs=True, direct_critic=direct)
    for _ in range(3):
      gan.critic_step(real_t, fake_t, it=1, alpha=alpha_t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
      gan.critic_step(real_t, fake_t, it=1, alpha=alpha_t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    eager = GAN(make_cfg(), device=dev, use_graphs=False, direct_critic=direct)
    eager.critic_step(real_t, fake_t, it=1, alpha=alpha_t)
    cnt, names = count_launches(lambda: eager.critic_step(real_t, fake_t, it=1, alpha=alpha_t))
    print('critic step direct=%s: %.3f ms per replayed step, %d launches (eager count)' % (direct, ms, cnt))
    if direct and '-v' in sys.argv:
      for nm in names:
        print('   ', nm[:110])
  zt, st = t(z), t(states)
  for direct in (True, False):
    torch.manual_seed(0)
    gan = GAN(make_cfg(), device=dev, use_graphs=True, direct_generator=direct)
    for _ in range(3):
      gan.generator_step(fake_t, zt, st, 0.3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
      gan.generator_step(fake_t, zt, st, 0.3, it=1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    eager = GAN(make_cfg(), device=dev, use_graphs=False, direct_generator=direct)
    eager.generator_step(fake_t, zt, st, 0.3, it=1)
    cnt, names = count_launches(lambda: eager.generator_step(fake_t, zt, st, 0.3, it=1))
    print('G / V step direct=%s: %.3f ms per replayed step, %d launches (eager count)' % (direct, ms, cnt))
    if direct and '-g' in sys.argv:
      import collections
      for nm, k in collections.Counter(names).most_common(70):
        print('   %3d %s' % (k, nm[:120]))

if __name__ == '__main__':
  main()
