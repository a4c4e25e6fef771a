"""Generates tests/golden/filters_*.npz: committed input/expected-output vectors for every filter.

The reference (TF-1) cannot be imported in this image, so these vectors come from the float64
oracle (oracle/filters_np.py, pinned by tests/test_oracle_filters.py); they freeze its behaviour so
that neither the oracle nor the HIP kernels can drift unnoticed.  Data only: fp16-quantised inputs,
float32 parameters, float64-computed outputs stored as float32.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from exposure_amd import synthetic  # noqa: E402
from oracle import agent_np  # noqa: E402
from oracle import filters_np as fnp  # noqa: E402

CASES = {'small': (2, 16, 16, 3), 'proxy': (1, 64, 64, 3), 'ragged': (2, 5, 7, 3),
         # round 3: channels BELOW 0 and above 1 in every pixel neighbourhood -- SaturationPlus feeds min(x, 1) (which may
         # be negative) into rgb_to_hsv, Gamma's max(x, 0.001), the luminance clamps of Contrast, the curves' clip
         'negative': (2, 12, 10, 3)}


def main():
  only = sys.argv[1:]  # e.g. `make_golden.py negative`: (re)generate the named cases only
  for name, shape in CASES.items():
    if only and name not in only:
      continue
    out = {}
    for fid in range(9):
      rng = np.random.default_rng(9000 + 17 * fid + len(name))
      x = synthetic.make_images(rng, shape, np.float16)
      if name == 'negative':
        x = (x.astype(np.float32) * 1.4 - 0.3).astype(np.float16)
        x.reshape(-1, 3)[::5, fid % 3] = np.float16(-0.25)  # one clearly negative channel in every fifth pixel
      if fid in (4, 7, 8):  # put some samples exactly on knots / clip edges (tie conventions)
        flat = x.reshape(-1)
        flat[::37] = np.float16(0.125) * (np.arange(flat[::37].size) % 9)
      dy = synthetic.make_grad(rng, shape, np.float16)
      p = synthetic.make_params(rng, fid, shape[0])
      x64, dy64, p64 = x.astype(np.float64), dy.astype(np.float64), p.astype(np.float64)
      y = fnp.process_packed(fid, x64, p64)
      dx, dp = fnp.backward_packed(fid, x64, p64, dy64)
      out['x_%d' % fid] = x
      out['dy_%d' % fid] = dy
      out['p_%d' % fid] = p
      out['y_%d' % fid] = y.astype(np.float32)
      out['dx_%d' % fid] = dx.astype(np.float32)
      out['dp_%d' % fid] = dp.astype(np.float32)
      # round 4: the sum of ABSOLUTE terms of every parameter gradient -- the scale the accumulated rounding of dp is
      # judged against (tests/_tol.py: |err| <= 1e-4 |dp| + 2e-6 adp)
      out['adp_%d' % fid] = fnp.param_grad_abs(fid, x64, p64, dy64).astype(np.float32)
    rng = np.random.default_rng(77)
    img = synthetic.make_images(rng, shape, np.float16)
    if name == 'negative':
      img = (img.astype(np.float32) * 1.4 - 0.3).astype(np.float16)
    out['stats_x'] = img
    out['stats'] = agent_np.critic_stats(img.astype(np.float64)).astype(np.float32)
    out['penalty'] = agent_np.overexposure_penalty(img.astype(np.float64) * 1.5).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'filters_%s.npz' % name), **out)
  if only:
    return
  # action sampling: the only reference-supplied known answer (pdf_sample_layer.py:55-78)
  pdf = np.tile(np.array([[2.0, 4.0, 8.0]], dtype=np.float32), (9, 1))
  u = np.linspace(0, 1, 9, dtype=np.float32)[:, None]
  np.savez_compressed(os.path.join(HERE, 'pdf_sample.npz'), pdf=pdf, u=u, ids=agent_np.pdf_sample(pdf, u))


if __name__ == '__main__':
  main()
