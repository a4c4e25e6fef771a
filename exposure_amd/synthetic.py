"""Seeded synthetic inputs for the filter-stack benchmarks and parity tests.

Shapes and distributions follow SURVEY.md section 8(d): linear-RAW-like images
``x ~ U(0,1)**2.2`` scaled so that ~1 % of values exceed 1.0 (exercises the
clamps), rounded to fp16; filter parameters drawn *through the reference
regressors* from ``f ~ N(0,1)`` so they land in the ranges the agent produces
(``/root/reference/filters.py:177-179, 201-203, 224-235, 256-262, 306-310,
411-413, 435-436, 481-482``); upstream gradient ``dy ~ N(0,1)`` rounded to fp16.

NumPy only (host side); callers move the arrays to the device.
"""
import math

import numpy as np

FILTER_NAMES = ('E', 'G', 'W', 'S+', 'T', 'Ct', 'BW', 'C', 'Le')  # 'Le' = LevelFilter (not in cfg.filters)
NUM_PARAMS = (1, 1, 3, 1, 8, 1, 1, 24, 2)

SHAPES = {
    'A': (64, 64, 64, 3),  # BASELINE config 2
    'B': (16, 512, 512, 3),  # BASELINE config 5
    'C': (64, 512, 512, 3),  # the shape the headline metric is quoted on
}


def make_images(rng, shape, dtype=np.float16):
  u = rng.random(shape, dtype=np.float32)
  x = u**np.float32(2.2)
  # scale so that the 99th percentile of U**2.2 maps to 1.0  (0.99**2.2 = 0.978)
  x *= np.float32(1.0 / 0.99**2.2)
  return x.astype(dtype)


def make_grad(rng, shape, dtype=np.float16):
  return rng.standard_normal(shape, dtype=np.float32).astype(dtype)


def _tanh_range(l, r, x):
  return (np.tanh(x) * 0.5 + 0.5) * (r - l) + l


def regress(fid, f):
  """Packed (N,P) float32 parameters from raw features f (N,P); same maths as the
  reference regressors (bias terms are all atanh(0) = 0 for the shipped cfg)."""
  f = np.asarray(f, dtype=np.float64)
  name = FILTER_NAMES[fid]
  if name == 'E':
    p = _tanh_range(-3.5, 3.5, f)
  elif name == 'G':
    p = np.exp(_tanh_range(-math.log(3), math.log(3), f))
  elif name == 'W':
    s = np.exp(_tanh_range(-0.5, 0.5, f * np.array([[0.0, 1.0, 1.0]])))
    p = s / (1e-5 + 0.27 * s[:, 0:1] + 0.67 * s[:, 1:2] + 0.06 * s[:, 2:3])
  elif name in ('S+', 'BW', 'Le'):
    p = 1.0 / (1.0 + np.exp(-f))
  elif name == 'T':
    p = _tanh_range(0.5, 2.0, f)
  elif name == 'Ct':
    p = np.tanh(f)
  elif name == 'C':
    p = _tanh_range(0.90, 1.10, f)
  else:
    raise ValueError(fid)
  return p.astype(np.float32)


def make_params(rng, fid, n):
  f = rng.standard_normal((n, NUM_PARAMS[fid]))
  return regress(fid, f)


def make_case(seed, shape, dtype=np.float16):
  """(x, dy, [params for fid 0..7]) for one benchmark / parity case."""
  rng = np.random.default_rng(seed)
  x = make_images(rng, shape, dtype)
  dy = make_grad(rng, shape, dtype)
  params = [make_params(rng, fid, shape[0]) for fid in range(8)]
  return x, dy, params
