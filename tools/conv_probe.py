"""Time the convnet building blocks on the GPU to choose the plumbing configuration
(channel padding, memory format, benchmark mode).  Not part of the product."""
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == 'bench'


def timeit(fn, reps=20):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / reps * 1e6


for cin in (6, 8, 14, 16, 17, 24, 32):
  for cl in (True, False):
    x = torch.randn(64, cin, 64, 64, device=dev, requires_grad=True)
    w = torch.randn(32, cin, 4, 4, device=dev, requires_grad=True)
    if cl:
      x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
      w = w.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def fwd():
      return F.conv2d(x, w, None, stride=2, padding=1)

    def fwdbwd():
      y = F.conv2d(x, w, None, stride=2, padding=1)
      gx, gw = torch.autograd.grad(y.sum(), [x, w])
      return gx

    def dbl():
      y = F.conv2d(x, w, None, stride=2, padding=1)
      gx, = torch.autograd.grad(y.pow(2).sum(), [x], create_graph=True)
      gw, = torch.autograd.grad(gx.pow(2).sum(), [w])
      return gw

    print('cin %2d channels_last %d: fwd %7.1f us  fwd+bwd %7.1f us  double-bwd %7.1f us' %
          (cin, cl, timeit(fwd), timeit(fwdbwd), timeit(dbl)))
