"""Tolerances for HIP-vs-oracle comparisons.

north_star: outputs match the reference within 1e-3 per pixel (fp32).  With fp16 *storage*
the output itself is rounded to the nearest half (relative 2^-11), so the fp16 bound is
1e-3 + half-ulp_fp16(|ref|); with fp32 storage the bound is much tighter."""
import numpy as np


def image_tol(ref, dtype):
  ref = np.abs(np.asarray(ref, dtype=np.float64))
  if dtype == np.float16:
    return 1e-3 + ref * 2.0**-11 + 2.0**-25
  return 2e-5 + ref * 2e-5


def assert_image_close(got, ref, dtype, what=''):
  got = np.asarray(got, dtype=np.float64)
  ref = np.asarray(ref, dtype=np.float64)
  err = np.abs(got - ref)
  tol = image_tol(ref, dtype)
  bad = err > tol
  assert not bad.any(), '%s: %d / %d elements out of tolerance, worst err %.3e (tol %.3e) at ref %.6g' % (
      what, bad.sum(), bad.size, err[bad].max(), tol[bad][err[bad].argmax()], ref[bad][err[bad].argmax()])


def assert_param_grad_close(got, ref, scale, what=''):
  """Parameter gradients are sums over H*W*3 products; `scale` = sum |terms| bound the
  accumulated rounding (fp32 accumulation + fp16-free inputs): rel 2e-4 of that scale."""
  got = np.asarray(got, dtype=np.float64)
  ref = np.asarray(ref, dtype=np.float64)
  tol = 2e-4 * np.maximum(np.abs(ref), scale) + 1e-6
  err = np.abs(got - ref)
  assert (err <= tol).all(), '%s: worst err %.3e vs tol %.3e (ref %.4g)' % (what, err.max(), tol.flat[err.argmax()],
                                                                          ref.flat[err.argmax()])
