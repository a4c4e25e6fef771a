"""How much of a training iteration is the replay memory (eager indexing + host syncs) rather than the optimisation steps?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider

dev = torch.device('cuda:0')
cfg = make_cfg()
torch.manual_seed(0)
gan = GAN(cfg, device=dev, use_graphs=True, seed=0)
n = cfg.batch_size
memory = ReplayMemory(cfg, SyntheticProvider(dev, dtype=torch.float16, seed=1), SyntheticProvider(dev, gamma=1.0, dtype=torch.float16, seed=2), seed=0)
for _ in range(8):
  feed, feats = memory.get_feed_dict_and_states(n)
  o = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
  memory.replace_memory(o['fake_output'], o['new_states'], feats, advanced=True)

LAST = [None, None]
def full(it):
  feed, feats = memory.get_feed_dict_and_states(n)
  out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], progress=0.1, it=it)
  LAST[:] = [out['fake_output'].clone(), out['new_states'].clone()]
  memory.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)
  for _ in range(cfg.citers):
    rep = memory.get_replay_feed_dict(n)
    gan.critic_step(rep['real_data'], rep['fake_output'], it=it)

feed0, feats0 = memory.get_feed_dict_and_states(n)
rep0 = memory.get_replay_feed_dict(n)
def steps_only(it):
  gan.generator_step(feed0['fake_input'], feed0['z'], feed0['states'], progress=0.1, it=it)
  for _ in range(cfg.citers):
    gan.critic_step(rep0['real_data'], rep0['fake_output'], it=it)

def mem_only(it):
  feed, feats = memory.get_feed_dict_and_states(n)
  memory.replace_memory(LAST[0], LAST[1], feats)
  for _ in range(cfg.citers):
    memory.get_replay_feed_dict(n)

for name, fn in (('full iteration', full), ('optimisation steps only (fixed batches)', steps_only), ('replay memory only', mem_only), ('full iteration', full)):
  for i in range(3): fn(i + 1)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(20): fn(i + 1)
  torch.cuda.synchronize(); print('%-45s %.3f ms' % (name, (time.perf_counter() - t0) / 20 * 1e3))
