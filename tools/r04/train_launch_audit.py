"""Which Python lines of a training iteration launch which GPU kernels (eager mode, torch.profiler with stacks).
Usage (GPU box): python tools/r04/train_launch_audit.py [out.txt]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg  # noqa: E402
from exposure_amd.gan import GAN  # noqa: E402
from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider  # noqa: E402


def main():
  out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
  dev = torch.device('cuda:0')
  cfg = make_cfg()
  torch.manual_seed(0)
  gan = GAN(cfg, device=dev, use_graphs=False, seed=0)
  n = cfg.batch_size
  memory = ReplayMemory(cfg, SyntheticProvider(dev, dtype=torch.float16, seed=1),
                        SyntheticProvider(dev, gamma=1.0, dtype=torch.float16, seed=2), seed=0)
  for _ in range(8):
    feed, feats = memory.get_feed_dict_and_states(n)
    o = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
    memory.replace_memory(o['fake_output'], o['new_states'], feats, advanced=True)
  rep = memory.get_replay_feed_dict(n)
  gan.critic_step(rep['real_data'], rep['fake_output'], it=1)
  torch.cuda.synchronize()
  from torch.profiler import ProfilerActivity, profile
  for which in ('generator_step', 'critic_step'):
    feed, feats = memory.get_feed_dict_and_states(n)
    rep = memory.get_replay_feed_dict(n)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
      if which == 'generator_step':
        gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.1, it=1)
      else:
        gan.critic_step(rep['real_data'], rep['fake_output'], it=1)
      torch.cuda.synchronize()
    events = prof.events()
    # map each device kernel to the CPU op that launched it (correlation via the launch event's parent chain)
    by_site = collections.Counter()
    by_kernel = collections.Counter()
    total = 0
    for ev in events:
      if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
        continue
      # only leaf CPU ops that own kernels
      if any(c.kernels for c in ev.cpu_children):
        continue
      site = '?'
      for fr in (ev.stack or []):
        if 'exposure_amd' in fr:
          site = fr.split('exposure_amd/')[-1].split(',')[0] if 'exposure_amd/' in fr else fr
          break
      # climb to a parent with a stack if this op has none
      p = ev
      while site == '?' and p.cpu_parent is not None:
        p = p.cpu_parent
        for fr in (p.stack or []):
          if 'exposure_amd' in fr:
            site = fr.split('exposure_amd/')[-1]
            break
      for k in ev.kernels:
        total += 1
        by_site[(site[:70], ev.name[:40])] += 1
        by_kernel[k.name[:90]] += 1
    print('=====', which, total, 'kernels', file=out)
    for (site, op), c in by_site.most_common(70):
      print('%5d  %-42s %s' % (c, op, site), file=out)
    print('--- by kernel', file=out)
    for k, c in by_kernel.most_common(45):
      print('%5d  %s' % (c, k), file=out)


if __name__ == '__main__':
  main()
