"""The per-image reductions (parameter gradients, penalty, statistics) run without float atomics: one
workspace record per block, a finish launch adds an image's records in a fixed order
(exposure_hip.hip::block_reduce_record / finish_kernel; include/exposure_hip.h "Reduction workspace").
Results must be BIT-identical run to run, correct on every one of many back-to-back launches with fresh
data sharing one workspace (a stale record would show as a wrong sum), and independent of whatever the
workspace held before."""
import numpy as np
import pytest
import torch

from exposure_amd import _cabi, synthetic
from oracle import agent_np
from oracle import filters_np as fnp
from tests._tol import assert_param_grad_close

pytestmark = pytest.mark.gpu


def device_case(shape, seed, dev, dtype=np.float16):
  x, dy, params = synthetic.make_case(seed, shape, dtype)
  return x, dy, params, torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev), [torch.from_numpy(p).to(dev)
                                                                                  for p in params]


@pytest.mark.parametrize('shape', [(8, 256, 256, 3), (64, 64, 64, 3), (2, 1024, 768, 3), (5, 33, 47, 3)])
def test_parameter_gradients_are_bit_reproducible(shape, gpu_device):
  x, dy, params, tx, tdy, tp = device_case(shape, 11, gpu_device)
  for fid in range(8):
    runs = []
    for _ in range(4):
      dp = torch.full_like(tp[fid], float('nan'))  # overwritten, never accumulated into
      dx = torch.empty_like(tx)
      _cabi.filter_bwd(fid, tx, tdy, dx, tp[fid], dp)
      runs.append((dp.cpu().numpy().copy(), dx.cpu().numpy().copy()))
    for dp_i, dx_i in runs[1:]:
      assert np.array_equal(dp_i.view(np.uint32), runs[0][0].view(np.uint32)), fid
      assert np.array_equal(dx_i.view(np.uint16), runs[0][1].view(np.uint16)), fid
    assert np.isfinite(runs[0][0]).all()


def test_back_to_back_launches_with_fresh_data_never_read_stale_records(gpu_device):
  """200 launches sharing one workspace, inputs changing every time: every result is checked against a
  float64 reduction of the SAME inputs (computed by torch on the device from the kernel's own dx)."""
  dev = gpu_device
  shape = (16, 128, 128, 3)
  n = shape[0]
  gen = torch.Generator(device=dev).manual_seed(5)
  ws = _cabi.new_workspace(dev, _cabi.workspace_bytes(n, 128, 128, _cabi.EXPO_F16))
  for it in range(200):
    x = torch.rand(shape, device=dev, generator=gen).half()
    dy = torch.randn(shape, device=dev, generator=gen).half()
    fid = it % 3  # E, G, W: closed-form parameter gradients from x, dy (and dx) alone
    if fid == 0:
      p = torch.rand((n, 1), device=dev, generator=gen) * 2 - 1
      terms = dy.double() * x.double() * (2.0**p.double())[:, :, None, None] * np.log(2.0)
      want, a = terms.sum(dim=(1, 2, 3))[:, None], terms.abs().sum(dim=(1, 2, 3))[:, None]
    elif fid == 1:
      p = torch.rand((n, 1), device=dev, generator=gen) * 2 + 0.4
      xm = x.double().clamp_min(0.001)
      terms = dy.double() * xm**p.double()[:, :, None, None] * torch.log(xm)
      want, a = terms.sum(dim=(1, 2, 3))[:, None], terms.abs().sum(dim=(1, 2, 3))[:, None]
    else:
      p = torch.rand((n, 3), device=dev, generator=gen) + 0.5
      terms = dy.double() * x.double()
      want, a = terms.sum(dim=(1, 2)), terms.abs().sum(dim=(1, 2))
    dp = torch.empty_like(p)
    _cabi.filter_bwd(fid, x, dy, None, p.contiguous(), dp, workspace=ws)
    assert_param_grad_close(dp.cpu().numpy(), want.cpu().numpy(), a.cpu().numpy(), 'launch %d filter %d' % (it, fid))
  torch.cuda.synchronize()


def test_results_do_not_depend_on_what_the_workspace_held(gpu_device):
  """Records are fully overwritten before they are read: the workspace needs no initialisation."""
  shape = (6, 96, 160, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 3, gpu_device)
  nbytes = _cabi.workspace_bytes(6, 96, 160, _cabi.EXPO_F16)
  clean = _cabi.new_workspace(gpu_device, nbytes)
  clean.zero_()
  dirty = _cabi.new_workspace(gpu_device, nbytes)
  dirty.fill_(0x7f)  # NaN-ish garbage everywhere
  for fid in (0, 4, 7):
    a, b = torch.empty_like(tp[fid]), torch.empty_like(tp[fid])
    _cabi.filter_bwd(fid, tx, tdy, None, tp[fid], a, workspace=clean)
    _cabi.filter_bwd(fid, tx, tdy, None, tp[fid], b, workspace=dirty)
    assert torch.equal(a, b), fid


def test_workspace_too_small_or_missing_is_refused(gpu_device):
  shape = (4, 128, 128, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 4, gpu_device)
  tiny = torch.zeros(64, dtype=torch.uint8, device=gpu_device)
  with pytest.raises(_cabi.ExposureHipError, match='workspace'):
    _cabi.filter_bwd(0, tx, tdy, None, tp[0], torch.empty_like(tp[0]), workspace=tiny)
  lib = _cabi.load()
  rc = lib.expo_filter_bwd(0, tx.data_ptr(), tdy.data_ptr(), None, tp[0].data_ptr(), tp[0].data_ptr(), 4, 128, 128, 0,
                           0, None, 0, None)
  assert rc == -1 and b'workspace' in lib.expo_last_error()


def test_chain_backward_one_finish_for_all_steps(gpu_device):
  """expo_chain_bwd keeps every step's records in its own slice of the workspace and finishes them with
  one launch: same dparams, bit for bit, as eight single calls."""
  shape = (5, 96, 96, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 9, gpu_device)
  ids = list(range(8))
  acts = [tx] + [torch.empty_like(tx) for _ in ids]
  _cabi.chain_fwd(ids, acts, tp)
  grads = [torch.empty_like(tx) for _ in ids] + [tdy]
  dps = [torch.full_like(p, float('nan')) for p in tp]
  _cabi.chain_bwd(ids, acts, grads, tp, dps)
  g = tdy
  for i in reversed(ids):
    dp = torch.empty_like(tp[i])
    dx = torch.empty_like(tx)
    _cabi.filter_bwd(i, acts[i], g, dx, tp[i], dp)
    assert torch.equal(dp, dps[i]), i
    assert torch.equal(dx, grads[i]), i
    g = dx
  with pytest.raises(_cabi.ExposureHipError, match='workspace'):  # one step's worth is not enough for a chain
    one = _cabi.new_workspace(gpu_device, _cabi.workspace_bytes(5, 96, 96, _cabi.EXPO_F16))
    _cabi.chain_bwd(ids, acts, grads, tp, dps, workspace=one)


def test_records_pass_plus_finish_equals_filter_bwd(gpu_device):
  """expo_filter_bwd_records + expo_finish_bwd (the two halves a caller may batch) == expo_filter_bwd."""
  shape = (4, 80, 112, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 12, gpu_device)
  nb = _cabi.workspace_bytes(4, 80, 112, _cabi.EXPO_F16)
  ws = _cabi.new_workspace(gpu_device, 3 * nb)
  fids = [7, 2, 5]
  dxs = [torch.empty_like(tx) for _ in fids]
  for k, fid in enumerate(fids):
    _cabi.filter_bwd_records(fid, tx, tdy, dxs[k], tp[fid], workspace=ws[k * nb:])
  dps = [torch.full_like(tp[fid], float('nan')) for fid in fids]
  _cabi.finish_bwd(fids, tx, [tp[f] for f in fids], dps, workspace=ws)
  for k, fid in enumerate(fids):
    dp, dx = torch.empty_like(tp[fid]), torch.empty_like(tx)
    _cabi.filter_bwd(fid, tx, tdy, dx, tp[fid], dp)
    assert torch.equal(dp, dps[k]) and torch.equal(dx, dxs[k]), fid


def test_accumulate_adds_to_the_existing_value(gpu_device):
  shape = (3, 64, 64, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 6, gpu_device)
  for fid in (2, 7):
    fresh = torch.empty_like(tp[fid])
    _cabi.filter_bwd(fid, tx, tdy, None, tp[fid], fresh)
    acc = torch.full_like(tp[fid], 3.0)
    _cabi.filter_bwd(fid, tx, tdy, None, tp[fid], acc, accumulate=True)
    assert torch.equal(acc, fresh + 3.0)
    _cabi.filter_bwd(fid, tx, tdy, None, tp[fid], acc, accumulate=True)
    assert torch.allclose(acc, 2 * fresh + 3.0, rtol=1e-6, atol=1e-6)


def test_dispatch_rows_are_fully_written_without_a_fill(gpu_device):
  """dparams rows come back complete from NaN-poisoned memory: P gradients + zeros for the unused
  slots, an all-zero row for id -1; the fused penalty likewise."""
  dev = gpu_device
  n = 11
  shape = (n, 64, 64, 3)
  x, dy, params, tx, tdy, tp = device_case(shape, 8, dev)
  ids = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 7], dtype=np.int32)
  p24 = np.zeros((n, 24), dtype=np.float32)
  rng = np.random.default_rng(1)
  for i, fid in enumerate(ids):
    if fid >= 0:
      p24[i, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, int(fid), 1)[0]
  tids, tp24 = torch.from_numpy(ids).to(dev), torch.from_numpy(p24).to(dev)
  y = torch.empty_like(tx)
  pen = torch.full((n,), float('nan'), device=dev)
  _cabi.dispatch_fwd(tids, tx, y, tp24, pen)
  dpen = torch.rand(n, device=dev)
  dx = torch.empty_like(tx)
  dp = torch.full((n, 24), float('nan'), device=dev)
  _cabi.dispatch_bwd(tids, tx, tdy, dx, tp24, dp, dpen)
  dp_h, pen_h = dp.cpu().numpy(), pen.cpu().numpy()
  assert np.isfinite(dp_h).all() and np.isfinite(pen_h).all()
  for i, fid in enumerate(ids):
    npar = fnp.NUM_PARAMS[fid] if fid >= 0 else 0
    assert (dp_h[i, npar:] == 0).all(), (i, fid)
    if fid < 0:
      assert pen_h[i] == 0.0 and float(y[i].abs().max()) == 0.0
      continue
    xi = x[i:i + 1].astype(np.float64)
    yi = fnp.process_packed(int(fid), xi, p24[i:i + 1, :npar].astype(np.float64))
    assert abs(pen_h[i] - agent_np.overexposure_penalty(yi)[0]) <= 1e-5 + 1e-4 * abs(pen_h[i])
    g = dy[i:i + 1].astype(np.float64) + 2.0 * np.maximum(yi - 1, 0) * float(dpen[i]) / (64 * 64 * 3)
    _, rdp = fnp.backward_packed(int(fid), xi, p24[i:i + 1, :npar].astype(np.float64), g)
    adp = fnp.param_grad_abs(int(fid), xi, p24[i:i + 1, :npar].astype(np.float64), g)
    assert_param_grad_close(dp_h[i:i + 1, :npar], rdp, adp, 'dispatch row %d (filter %d)' % (i, fid))


def test_dispatch_backward_on_streaming_sized_tensors_and_in_a_graph(gpu_device):
  """Tensors >= 8 MiB take the nt / sc1 instantiations of the light / curve launch pair.  Every image must
  equal the per-filter backward of that image bit for bit (dx) / to summation order (dparams), back-to-back
  calls must not see each other's records, and a hipGraph capture of the call replays to the same bits."""
  from oracle import filters_c as fc
  dev = gpu_device
  n = 10
  shape = (n, 512, 512, 3)
  rng = np.random.default_rng(5)
  ids = np.array([7, 0, 4, 3, -1, 7, 5, 4, 1, 6], dtype=np.int32)
  p24 = np.zeros((n, 24), dtype=np.float32)
  for i, fid in enumerate(ids):
    if fid >= 0:
      p24[i, :fnp.NUM_PARAMS[fid]] = synthetic.make_params(rng, int(fid), 1)[0]
  g = torch.Generator(device=dev).manual_seed(3)
  tx = (torch.rand(shape, device=dev, generator=g)**2.2 * 1.1).half()
  tdy = torch.randn(shape, device=dev, generator=g).half()
  tids, tp24 = torch.from_numpy(ids).to(dev), torch.from_numpy(p24).to(dev)
  assert tx.numel() * 2 >= 8 << 20
  dx = torch.empty_like(tx)
  dp = torch.full((n, 24), float('nan'), device=dev)
  for _ in range(3):  # back to back
    dx.fill_(float('nan'))
    _cabi.dispatch_bwd(tids, tx, tdy, dx, tp24, dp, None)
  torch.cuda.synchronize()
  for i, fid in enumerate(ids):
    if fid < 0:
      assert float(dx[i].abs().max()) == 0.0 and float(dp[i].abs().max()) == 0.0
      continue
    npar = fnp.NUM_PARAMS[fid]
    rdx = torch.empty_like(tx[i:i + 1])
    rdp = torch.empty((1, npar), device=dev)
    _cabi.filter_bwd(int(fid), tx[i:i + 1], tdy[i:i + 1], rdx, tp24[i:i + 1, :npar].contiguous(), rdp)
    assert torch.equal(dx[i:i + 1], rdx), 'dx of image %d (filter %d)' % (i, fid)
    # both summation orders against the float64 C restatement (|err| <= 1e-4 |ref| + 2e-6 A, tests/_tol.py)
    _, odp, adp = fc.backward_packed(int(fid), tx[i:i + 1].cpu().numpy().astype(np.float64),
                                     p24[i:i + 1, :npar].astype(np.float64),
                                     tdy[i:i + 1].cpu().numpy().astype(np.float64), with_abs=True)
    assert_param_grad_close(dp[i:i + 1, :npar].cpu().numpy(), odp, adp, 'dispatch dparams of image %d (filter %d)' % (i, fid))
    assert_param_grad_close(rdp.cpu().numpy(), odp, adp, 'per-filter dparams of image %d (filter %d)' % (i, fid))
    assert npar == 24 or float(dp[i, npar:].abs().max()) == 0.0
  # the same call replayed from a hipGraph
  dx_g = torch.empty_like(tx)
  dp_g = torch.full((n, 24), float('nan'), device=dev)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    _cabi.dispatch_bwd(tids, tx, tdy, dx_g, tp24, dp_g, None)
  torch.cuda.current_stream().wait_stream(side)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  from exposure_amd.util import capture_without_gc
  with capture_without_gc(), torch.cuda.graph(graph):
    _cabi.dispatch_bwd(tids, tx, tdy, dx_g, tp24, dp_g, None)
  for _ in range(3):
    dx_g.fill_(float('nan'))
    dp_g.fill_(float('nan'))
    graph.replay()
  torch.cuda.synchronize()
  assert torch.equal(dx_g, dx) and torch.equal(dp_g, dp)


@pytest.mark.parametrize('shape', [(3, 8, 8, 3), (1, 2048, 2048, 3), (70, 40, 40, 3)])
def test_stats_and_penalty_single_launch(shape, gpu_device):
  """critic statistics are finished by the image's last block (no separate finish kernel, no fill)."""
  rng = np.random.default_rng(2)
  img = synthetic.make_images(rng, shape, np.float16)
  t = torch.from_numpy(img).to(gpu_device)
  stats = torch.full((shape[0], 3), float('nan'), device=gpu_device)
  pen = torch.full((shape[0],), float('nan'), device=gpu_device)
  for _ in range(3):
    _cabi.critic_stats(t, stats)
    _cabi.overexposure_penalty(t, pen)
  ref = agent_np.critic_stats(img.astype(np.float64))
  assert np.abs(stats.cpu().numpy() - ref).max() < 2e-5
  assert np.abs(pen.cpu().numpy() - agent_np.overexposure_penalty(img.astype(np.float64))).max() < 1e-6
