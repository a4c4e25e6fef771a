"""Tolerances for HIP-vs-oracle comparisons.

north_star: outputs match the reference within 1e-3 per pixel (fp32).  With fp16 *storage*
the output itself is rounded to the nearest half (relative 2^-11), so the fp16 bound is
1e-3 + half-ulp_fp16(|ref|); with fp32 storage the bound is much tighter."""
import json
import os

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


# Reduced quantities (parameter gradients, mask-parameter gradients, J v): sums of H*W*3 terms accumulated in fp32 by
# 256-thread partials + a tree.  The honest error model is rounding relative to the sum of the ABSOLUTE terms
# A = sum_e |dy_e * dy_e/dp_k| (returned by the oracles: filters_np.param_grad_abs, filters_c.backward_packed(with_abs),
# filters_np.abs_terms_fd), NOT relative to sum |dy| (round 3's bound, which an all-zero gradient passed at 512x512):
#   |got - ref| <= 1e-4 |ref| + 2e-6 A        (2e-6 ~ 33 eps_fp32)
# tests/test_tolerance_mutation.py feeds zeros / 0.99 x / -1 x through this assertion and requires a failure.
PARAM_GRAD_REL = 1e-4
PARAM_GRAD_ABS_TERMS = 2e-6
_RECORD = os.environ.get('EXPO_RECORD_PARAM_ERR')  # path of a .jsonl collecting the worst |err| / A per call


def param_grad_tol(ref, abs_terms, abs_coeff=PARAM_GRAD_ABS_TERMS):
  ref = np.abs(np.asarray(ref, dtype=np.float64))
  return PARAM_GRAD_REL * ref + abs_coeff * np.broadcast_to(np.asarray(abs_terms, dtype=np.float64), ref.shape)


def assert_param_grad_close(got, ref, abs_terms, what='', abs_coeff=PARAM_GRAD_ABS_TERMS):
  """`abs_terms` = A, the oracle's sum of absolute per-element terms of each reduced value (same shape as `ref`).
  `abs_coeff` is raised ONLY where the oracle and the kernel do not see the same inputs (each such call site says why)."""
  got = np.asarray(got, dtype=np.float64)
  ref = np.asarray(ref, dtype=np.float64)
  assert got.shape == ref.shape, (got.shape, ref.shape)
  a = np.broadcast_to(np.asarray(abs_terms, dtype=np.float64), ref.shape)
  assert (a >= np.abs(ref) * (1 - 1e-6) - 1e-300).all(), '%s: A is not a sum of absolute terms of ref' % what
  # entries ten orders of magnitude below the largest scale of the same comparison are below the float64 ORACLE's own
  # resolution (e.g. SaturationPlus on a pixel with a zero channel: full - xc is exactly 0, the restatement's HSV round
  # trip leaves 1e-17): an absolute floor of 1e-10 max(A)
  tol = param_grad_tol(ref, a, abs_coeff) + 1e-10 * (a.max() if a.size else 0.0)
  err = np.abs(got - ref)
  if _RECORD:
    with np.errstate(divide='ignore', invalid='ignore'):
      live = a > 1e-8 * (a.max() if a.size else 0.0)  # entries whose scale is float64-oracle noise: not a measurement
      ra = np.where(live, err / np.where(live, a, 1.0), 0.0)
      rr = np.where(live & (np.abs(ref) > 0), err / np.where(ref != 0, np.abs(ref), 1.0), 0.0)
    with open(_RECORD, 'a') as f:
      f.write(json.dumps({'what': what, 'err_over_A': float(ra.max()), 'err_over_ref': float(rr.max()),
                          'ref_over_A': float((np.abs(ref) / np.where(a > 0, a, 1)).min())}) + '\n')
  ok = err <= tol
  assert ok.all(), '%s: %d / %d values out of tolerance; worst err %.3e vs tol %.3e (ref %.6g, A %.4g, err/A %.2e)' % (
      what, (~ok).sum(), ok.size, err[~ok].max(), tol[~ok][err[~ok].argmax()], ref[~ok][err[~ok].argmax()],
      a[~ok][err[~ok].argmax()], (err[~ok] / np.maximum(a[~ok], 1e-300)).max())
