"""The convolution launches of the training iteration, one shape at a time (for rocprofv3 --kernel-trace / --pmc):
python tools/r06/conv_cases.py [reps]   -- every primitive at the critic update's batch (192 / 64 images) and the G step's (64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402

LAYERS = [(6, 32, 64), (32, 64, 32), (64, 128, 16), (128, 256, 8), (14, 32, 64), (17, 32, 64)]


def main():
  reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
  dev = torch.device('cuda:0')
  g = torch.Generator(device=dev).manual_seed(0)
  for n in (64, 192):
    for cin, cout, h in LAYERS:
      x = torch.randn((n, h, h, cin), device=dev, generator=g)
      w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
      b = torch.zeros((cout,), device=dev)
      y = torch.empty((n, h // 2, h // 2, cout), device=dev)
      gy = torch.randn_like(y)
      dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
      for _ in range(reps):
        _cabi.conv4x4s2_fwd(x, w, b, y, 1, 0.2)
        _cabi.conv4x4s2_bwd_data_mask(gy, w, x, dx, 0.2)
        _cabi.conv4x4s2_wrw_bias(x, gy, dw, db)
      torch.cuda.synchronize()
      print('n=%d cin=%d cout=%d h=%d done' % (n, cin, cout, h), flush=True)


if __name__ == '__main__':
  main()
