"""Per-iteration wall time (synchronised) of GAN.train_iteration and which path each iteration took.
usage: python tools/r06/iter_probe.py [iterations]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg  # noqa: E402
from exposure_amd.gan import GAN  # noqa: E402
from exposure_amd.replay_memory import ReplayMemory, ResidentProvider  # noqa: E402

dev = torch.device('cuda:0')
cfg = make_cfg()
torch.manual_seed(0)
gan = GAN(cfg, device=dev, use_graphs=True, seed=0)
mem = ReplayMemory(cfg, ResidentProvider(dev, dtype=torch.float16, seed=1), ResidentProvider(dev, gamma=1.0, dtype=torch.float16, seed=2), seed=0)
n = cfg.batch_size
for _ in range(8):
  feed, feats = mem.get_feed_dict_and_states(n, lazy=True)
  out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
  mem.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)
torch.cuda.synchronize()
orig = mem.plan_iteration
state = {}


def plan(*a):
  t0 = time.perf_counter()
  p = orig(*a)
  state['plan_ms'] = (time.perf_counter() - t0) * 1e3
  state['planned'] = p is not None
  return p


mem.plan_iteration = plan
for it in range(1, int(sys.argv[1]) if len(sys.argv) > 1 else 25):
  t0 = time.perf_counter()
  gan.train_iteration(mem, it)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  live = int((mem._h_stopped[mem._order] != 1).sum())
  print('it %2d  host %.2f ms  total %.2f ms  planned %s (plan %.2f ms)  live after %d' %
        (it, (t1 - t0) * 1e3, (t2 - t0) * 1e3, state.get('planned'), state.get('plan_ms', 0), live))
