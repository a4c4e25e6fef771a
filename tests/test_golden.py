"""Committed golden vectors (tests/golden/, made by make_golden.py from the float64 oracle):
CPU: the oracle still reproduces them; GPU: the HIP kernels match them."""
import os

import numpy as np
import pytest
import torch

from oracle import agent_np
from oracle import filters_np as fnp
from tests._tol import assert_image_close, assert_param_grad_close

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['small', 'proxy', 'ragged', 'negative']


def load(name):
  return np.load(os.path.join(HERE, 'filters_%s.npz' % name))


@pytest.mark.parametrize('case', CASES)
def test_oracle_reproduces_golden(case):
  g = load(case)
  for fid in range(9):
    x, dy, p = (g['%s_%d' % (k, fid)].astype(np.float64) for k in ('x', 'dy', 'p'))
    np.testing.assert_allclose(fnp.process_packed(fid, x, p), g['y_%d' % fid], rtol=2e-6, atol=2e-7)
    dx, dp = fnp.backward_packed(fid, x, p, dy)
    np.testing.assert_allclose(dx, g['dx_%d' % fid], rtol=2e-6, atol=2e-6)
    adp = fnp.param_grad_abs(fid, x, p, dy)
    np.testing.assert_allclose(adp, g['adp_%d' % fid], rtol=1e-6)
    assert (np.abs(dp - g['dp_%d' % fid]) <= 1e-6 * np.abs(dp) + 1e-9 * adp).all(), fid  # the fixture stores float32
  np.testing.assert_allclose(agent_np.critic_stats(g['stats_x'].astype(np.float64)), g['stats'], rtol=1e-6)


def test_pdf_sample_golden():
  g = np.load(os.path.join(HERE, 'pdf_sample.npz'))
  assert np.array_equal(agent_np.pdf_sample(g['pdf'], g['u']), g['ids'])
  from exposure_amd.agent import pdf_sample
  assert np.array_equal(pdf_sample(torch.from_numpy(g['pdf']), torch.from_numpy(g['u'])).numpy(), g['ids'])
  assert g['ids'].tolist() == [-1, 0, 1, 1, 2, 2, 2, 2, 2]


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_hip_matches_golden(case, dtype, gpu_device):
  from exposure_amd import _cabi, critics
  g = load(case)
  dev = gpu_device
  npdt = np.float16 if dtype == torch.float16 else np.float32
  for fid in range(9):
    tx = torch.from_numpy(g['x_%d' % fid]).to(dev).to(dtype)
    tdy = torch.from_numpy(g['dy_%d' % fid]).to(dev).to(dtype)
    tp = torch.from_numpy(g['p_%d' % fid]).to(dev)
    y, dx, dp = torch.empty_like(tx), torch.empty_like(tx), torch.empty_like(tp)
    _cabi.filter_fwd(fid, tx, y, tp)
    _cabi.filter_bwd(fid, tx, tdy, dx, tp, dp)
    assert_image_close(y.float().cpu().numpy(), g['y_%d' % fid], npdt, 'golden y %d' % fid)
    assert_image_close(dx.float().cpu().numpy(), g['dx_%d' % fid], npdt, 'golden dx %d' % fid)
    assert_param_grad_close(dp.cpu().numpy(), g['dp_%d' % fid], g['adp_%d' % fid].astype(np.float64) * (1 + 1e-6),
                            'golden dp %d' % fid)
  sx = torch.from_numpy(g['stats_x']).to(dev).to(dtype)
  np.testing.assert_allclose(critics.critic_stats(sx).cpu().numpy(), g['stats'], rtol=2e-4, atol=2e-6)
  pen = torch.empty(sx.shape[0], device=dev)
  _cabi.overexposure_penalty((sx.float() * 1.5).to(dtype), pen)
  np.testing.assert_allclose(pen.cpu().numpy(), g['penalty'], rtol=2e-3, atol=1e-7)
