"""Round-5 incident, reproduced on purpose: a DEAD GAN whose step graphs are only reachable through a reference cycle,
and the cyclic collector forced to run inside the next GAN's hipGraph capture.  With exposure_amd.util.capture_without_gc
around the capture (the product) the run finishes; with the guard patched out (argument `unguarded`) destroying the
dead graphs inside the capture aborts the process on ROCm 7 / torch 2.10.

  python tools/r05/gc_capture_repro.py [unguarded]
"""
import gc
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from exposure_amd import gan as gan_mod, synthetic, util  # noqa: E402
from exposure_amd.config import make_cfg  # noqa: E402


class _NoGuard:

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


def main():
  unguarded = len(sys.argv) > 1 and sys.argv[1] == 'unguarded'
  if unguarded:
    gan_mod.capture_without_gc = _NoGuard
  # the collector is switched on the moment the stream starts capturing (and, guarded, switched off again by the guard's
  # own logic: capture_without_gc disables it BEFORE capture_begin, so this hook re-enabling it would defeat the guard --
  # hence the hook only arms the collector in the unguarded run; the guarded run arms it right before the step instead)
  real_begin = torch.cuda.CUDAGraph.capture_begin

  def begin(self, *a, **kw):
    real_begin(self, *a, **kw)
    if unguarded:
      gc.enable()
      gc.set_threshold(1, 1, 1)

  torch.cuda.CUDAGraph.capture_begin = begin
  dev = torch.device('cuda:0')
  cfg = make_cfg()
  rng = np.random.default_rng(1)
  n = 8
  t = lambda a: torch.from_numpy(a).to(dev)
  img = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  real = t(synthetic.make_images(rng, (n, 64, 64, 3), np.float16))
  states = torch.zeros(n, 11, device=dev)
  z = t(rng.random((n, 131), dtype=np.float32))

  def steps(g, its):
    for it in its:
      o = g.generator_step(img, z, states, progress=0.2, it=it)
      g.critic_step(real, o['fake_output'].clone(), it=it)

  gc.disable()  # nothing is collected until the capture below
  for _ in range(3):  # dead GANs with captured graphs, each a reference cycle (module -> buckets -> callback -> module)
    g = gan_mod.GAN(cfg, device=dev, use_graphs=True)
    steps(g, (3, 4, 5))
    assert any(e != 'warm' for e in g._graphs.values())
    del g
  torch.cuda.synchronize()
  live = gan_mod.GAN(cfg, device=dev, use_graphs=True)
  steps(live, (3,))  # warm-up (eager)
  if not unguarded:
    # guarded: the collector is live and eager when the step starts; capture_without_gc collects the dead GANs BEFORE the
    # capture and keeps the collector off inside it
    gc.enable()
    gc.set_threshold(1, 1, 1)
  steps(live, (4, 5))  # captures + replays
  torch.cuda.synchronize()
  gc.set_threshold(700, 10, 10)
  print('OK: captured with dead graph owners around (%s)' % ('unguarded' if gan_mod.capture_without_gc is _NoGuard else 'guarded'))


if __name__ == '__main__':
  main()
