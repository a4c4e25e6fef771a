"""-m gpu: VignetFilter.apply (filters.py:341-396) through expo_vignet_apply_fwd / _bwd against the float64
restatements (oracle/filters_np.py forward, oracle/filters_torch.py autograd backward)."""
import numpy as np
import pytest
import torch

from exposure_amd import _cabi, filters, synthetic
from exposure_amd.config import make_cfg
from oracle import filters_np as fnp
from oracle import filters_torch as ft
from tests._tol import assert_image_close, assert_param_grad_close

pytestmark = pytest.mark.gpu
NP_DT = {torch.float16: np.float16, torch.float32: np.float32}


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(4, 64, 64, 3), (3, 7, 5, 3), (2, 96, 160, 3), (2, 160, 96, 3)])
@pytest.mark.parametrize('masking', [True, False])
def test_vignet_apply_matches_oracle(dtype, shape, masking, gpu_device):
  dev = gpu_device
  rng = np.random.default_rng(8)
  n = shape[0]
  x = synthetic.make_images(rng, shape, NP_DT[dtype])
  dy = synthetic.make_grad(rng, shape, NP_DT[dtype])
  raw = rng.standard_normal((n, 5)).astype(np.float32) * 1.5
  raw[0] = [0.3, -2.0, 1.5, 2.5, -0.5]
  mp = (np.tanh(raw.astype(np.float64)) * 5).astype(np.float32)  # what the C-ABI takes
  tx, tdy, tmp = (torch.from_numpy(a).to(dev) for a in (x, dy, mp))
  y = torch.empty_like(tx)
  _cabi.vignet_apply_fwd(tx, y, tmp, 1.0, masking)
  raw64 = np.arctanh(mp.astype(np.float64) / 5.0)  # the raw parameters that reproduce the float32 mp exactly
  ref = fnp.vignet_apply(x.astype(np.float64), raw64, 1.0, masking)
  assert_image_close(y.float().cpu().numpy(), ref, NP_DT[dtype], 'vignet fwd')
  if not masking:
    assert float(y.float().abs().max()) == 0.0
  # backward: torch float64 autograd on the restatement, with respect to the squashed parameters the C-ABI sees
  xi = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
  mpi = torch.from_numpy(mp.astype(np.float64)).requires_grad_(True)
  out = ft.vignet_apply(xi, torch.atanh(mpi / 5.0), 1.0, masking)
  gx, gm = torch.autograd.grad(out, [xi, mpi], torch.from_numpy(dy.astype(np.float64)), allow_unused=True)
  gm = gm.numpy() if gm is not None else np.zeros((n, 5))
  dx = torch.empty_like(tx)
  dmp = torch.full((n, 5), float('nan'), device=dev)
  _cabi.vignet_apply_bwd(tx, tdy, dx, tmp, dmp, 1.0, masking)
  assert_image_close(dx.float().cpu().numpy(), gx.numpy(), NP_DT[dtype], 'vignet dx')
  # A = sum of absolute terms of each mask-parameter gradient (central differences of the NumPy restatement)
  a = fnp.abs_terms_fd(lambda q: fnp.vignet_apply(x.astype(np.float64), np.arctanh(q / 5.0), 1.0, masking),
                       mp.astype(np.float64), dy.astype(np.float64))
  assert_param_grad_close(dmp.cpu().numpy(), gm, a, 'vignet dmask')
  if not masking:
    assert float(dmp.abs().max()) == 0.0
  dmp2 = torch.empty_like(dmp)
  _cabi.vignet_apply_bwd(tx, tdy, None, tmp, dmp2, 1.0, masking)  # parameter gradients only
  # (two template instantiations of the kernel: the same sums in the same order, possibly contracted differently)
  assert torch.allclose(dmp, dmp2, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('masking', [True, False])
def test_vignet_filter_class_on_gpu(masking, gpu_device):
  """The Filter protocol end to end: features -> FC heads -> tanh_range -> one kernel; low- and high-resolution
  application share the parameters (filters.py:88-96) and both reach the mask half of fc2 in the backward."""
  dev = gpu_device
  cfg = make_cfg()
  cfg.masking = masking
  torch.manual_seed(1)
  v = filters.VignetFilter((1, 10, 14, 3), cfg).to(dev)
  img = torch.rand(2, 10, 14, 3, device=dev)
  high = torch.rand(2, 20, 12, 3, device=dev)
  feats = torch.randn(2, cfg.feature_extractor_dims, device=dev)
  low, hi, dbg = v.apply(img, img_features=feats, high_res=high)
  with torch.no_grad():
    _, mraw = v.extract_parameters(feats)
  ref = fnp.vignet_apply(img.cpu().numpy().astype(np.float64), mraw.cpu().numpy().astype(np.float64),
                         cfg.maximum_sharpness, masking)
  ref_hi = fnp.vignet_apply(high.cpu().numpy().astype(np.float64), mraw.cpu().numpy().astype(np.float64),
                            cfg.maximum_sharpness, masking)
  assert np.abs(low.detach().cpu().numpy() - ref).max() < 2e-5
  assert np.abs(hi.detach().cpu().numpy() - ref_hi).max() < 2e-5
  assert dbg['mask'].shape[-1] == 1
  (low.sum() + hi.sum()).backward()
  g = v.fc2.weight.grad[1:]  # the 5 mask rows
  assert (float(g.abs().max()) > 0.0) == masking
  assert float(v.fc2.weight.grad[:1].abs().max()) == 0.0  # process() = img * 0: the filter parameter reaches nothing
  # specified_parameter asserts masking off (filters.py:72) and then the output is 0
  if not masking:
    out, _, _ = v.apply(img, specified_parameter=torch.full((1, 1), 0.5, device=dev))
    assert float(out.abs().max()) == 0.0
    assert float(v.process(img, None).abs().max()) == 0.0
