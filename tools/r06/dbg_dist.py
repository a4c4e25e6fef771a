"""Which gradient tensors of the 2-rank step differ from the one-process step by more than 1e-3 of their own largest element (the comparison of tests/test_hip_dist.py, tensor by tensor).  usage: python tools/r06/dbg_dist.py"""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.multiprocessing as mp
from tests.test_hip_dist import _worker, _free_port, _inputs, _run_steps, _snapshot
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
if __name__ == '__main__':
  tmp = tempfile.mkdtemp()
  mp.spawn(_worker, args=(2, _free_port(), tmp, True, 2), nprocs=2, join=True)
  r0 = torch.load(os.path.join(tmp, 'rank0.pt'))
  dev = torch.device('cuda:0')
  torch.manual_seed(123)
  ref = GAN(make_cfg(), device=dev, seed=77)
  inputs = _inputs(dev)
  g, c = _run_steps(ref, *inputs, iters=1)
  torch.cuda.synchronize()
  with torch.no_grad():
    for p, a in zip(ref.parameters(), r0['params_after_first']):
      p.copy_(a.to(p.device))
  g, c = _run_steps(ref, *inputs, iters=1)
  torch.cuda.synchronize()
  want = _snapshot(ref, g, c)
  names = [n for n, _ in ref.named_parameters()]
  gmax = max(float(x.abs().max()) for x in want['grads'])
  print('largest gradient magnitude over all tensors', gmax)
  for name, got, ref_g in zip(names, r0['grads'], want['grads']):
    scale = float(ref_g.abs().max()) + 1e-12
    err = float((got - ref_g).abs().max())
    if err > 1e-3 * scale:
      print('%-40s scale %.3e err %.3e rel %.2e shape %s' % (name, scale, err, err / scale, tuple(ref_g.shape)))
