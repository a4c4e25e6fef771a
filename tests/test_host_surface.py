"""Host-side surface around the hot path (CPU; the C-ABI binding mocked by the oracle where a filter
runs): the train -> evaluate weight round trip, evaluate's I/O helpers (net.py:726-747), the Filter
protocol corners the advisor flagged, VignetFilter (filters.py:341-401) and the masking=True agent path."""
import os

import numpy as np
import pytest
import torch

from exposure_amd import agent as xagent
from exposure_amd import evaluate, filters
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from oracle import filters_np as fnp
from tests._fake_hip import fake_hip


def test_train_save_loads_into_evaluate_agent(tmp_path):
  torch.manual_seed(0)
  cfg = make_cfg()
  gan = GAN(cfg)
  path = tmp_path / 'w.pt'
  torch.save(gan.state_dict(), path)  # what `python -m exposure_amd.train --save` writes
  fresh = xagent.Agent(make_cfg())
  evaluate.load_agent_weights(fresh, torch.load(path))
  for (k, a), (_, b) in zip(gan.generator.state_dict().items(), fresh.state_dict().items()):
    assert torch.equal(a, b), k
  # a plain Agent state dict still loads, and so does the {'model', 'optim'} file train --save writes since round 6
  evaluate.load_agent_weights(xagent.Agent(make_cfg()), gan.generator.state_dict())
  both = xagent.Agent(make_cfg())
  evaluate.load_agent_weights(both, {'model': gan.state_dict(), 'optim': {}})
  for (k, a), (_, b) in zip(gan.generator.state_dict().items(), both.state_dict().items()):
    assert torch.equal(a, b), k


def test_output_paths_do_not_collide(tmp_path):
  d = str(tmp_path / 'out') + os.sep
  a = evaluate.output_path(d, '/x/a.tif', True)
  b = evaluate.output_path(d, '/x/b.tif', True)
  assert a != b and os.path.dirname(a) == str(tmp_path / 'out')
  f = str(tmp_path / 'res.npy')
  assert evaluate.output_path(f, '/x/a.tif', False) == f
  assert evaluate.output_path(f, '/x/a.tif', True) != evaluate.output_path(f, '/x/b.tif', True)
  assert evaluate.output_path(None, '/x/a.tif', True) == '/x/a.tif.retouched.npy'


def test_png_output_follows_cv2_imwrite(tmp_path):
  """net.py:769-772, 822-823: img * 255 rounded to nearest (half to even) and saturated; tone mapping = max to white,
  gamma 1 / 2.4."""
  from PIL import Image
  img = np.array([[[0.0, 0.5 / 255, 1.5 / 255], [2.5 / 255, 1.0, 1.7]], [[-0.2, 0.25, 254.5 / 255], [0.999, 0.001, 0.5]]],
                 dtype=np.float32)
  path = evaluate.save_png(str(tmp_path / 'a.png'), img)
  back = np.asarray(Image.open(path).convert('RGB'))
  want = np.array([[[0, 0, 2], [2, 255, 255]], [[0, 64, 254], [255, 0, 128]]], dtype=np.uint8)
  assert np.array_equal(back, want)
  lin = np.array([[[0.1, 0.2, 0.4]]], dtype=np.float32)
  np.testing.assert_allclose(evaluate.tone_mapped_input(lin), (lin / 0.4)**(1 / 2.4), rtol=1e-6)


def test_load_image_non_tif_branch(tmp_path):
  """net.py:739-747: /255, **2.2, / (2 max)."""
  from PIL import Image
  rng = np.random.default_rng(0)
  raw = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
  p = str(tmp_path / 'a.png')
  Image.fromarray(raw).save(p)
  got = evaluate.load_image(p)
  lin = (raw.astype(np.float64) / 255.0)**2.2
  want = lin / (2 * lin.max())
  assert got.shape == (9, 7, 3) and np.abs(got - want).max() < 1e-6 and abs(got.max() - 0.5) < 1e-6


def test_load_image_tif_branch(tmp_path):
  """net.py:729-733: 16-bit TIFF / 65535, ProPhoto linearisation x**1.8 (util.py:495-501)."""
  from exposure_amd.tiff16 import write_tiff
  raw = np.random.default_rng(1).integers(0, 65536, (64, 64, 3), dtype=np.uint16)
  p = str(tmp_path / 'a.tif')
  write_tiff(p, raw)
  got = evaluate.load_image(p)
  assert np.abs(got - (raw.astype(np.float64) / 65535.0)**1.8).max() < 1e-6


def test_module_apply_recurses_through_filters():
  """nn.Module.apply(fn) reaches every Filter although the reference's `apply` shadows it."""
  ag = xagent.Agent(make_cfg())
  seen = []
  ag.apply(lambda m: seen.append(type(m).__name__))
  assert 'ExposureFilter' in seen and 'Agent' in seen and seen.count('Linear') >= 18


def test_specified_parameter_broadcasts_over_the_batch():
  cfg = make_cfg()
  f = filters.ExposureFilter((1, 8, 8, 3), cfg)
  img = torch.rand(3, 8, 8, 3)
  with fake_hip():
    low, high, dbg = f.apply(img, specified_parameter=torch.tensor([[1.0]]))
  assert high is None and torch.allclose(low, img * 2.0, atol=1e-6)
  t = filters.ToneFilter((1, 8, 8, 3), cfg)
  with fake_hip():
    low, _, _ = t.apply(img, specified_parameter=torch.ones(1, 1, 1, 1, 8))
  assert torch.allclose(low, img.clamp(0, 1), atol=1e-6)


@pytest.mark.parametrize('masking', [False, True])
def test_vignet_filter_matches_oracle(masking):
  cfg = make_cfg()
  cfg.masking = masking
  torch.manual_seed(1)
  v = filters.VignetFilter((1, 10, 14, 3), cfg)
  img = torch.rand(2, 10, 14, 3)
  feats = torch.randn(2, cfg.feature_extractor_dims)
  with fake_hip():
    low, high, dbg = v.apply(img, img_features=feats, high_res=torch.rand(2, 20, 12, 3))
    with torch.no_grad():
      _, mraw = v.extract_parameters(feats)
  assert mraw.shape == (2, 5)
  ref = fnp.vignet_apply(img.numpy().astype(np.float64), mraw.numpy().astype(np.float64), cfg.maximum_sharpness,
                         masking)
  assert np.abs(low.detach().numpy() - ref).max() < 1e-6
  assert high.shape == (2, 20, 12, 3)
  with fake_hip():
    (low.sum() + high.sum()).backward()
  assert (float(v.fc2.weight.grad[1:].abs().max()) > 0.0) == masking  # the 5 mask rows of fc2
  if not masking:
    assert float(low.detach().abs().max()) == 0.0  # mask forced to 1, process = img * 0


def test_agent_masking_path_matches_oracle_and_gives_mask_gradients():
  """cfg.masking = True: out = sum_j onehot_j lerp(img, process_j(img), mask_j(img)) (agent.py:58-77,
  119-125; filters.py:86-88, 110-148) and the mask half of every selected head's fc2 gets a gradient."""
  cfg = make_cfg()
  cfg.masking = True
  torch.manual_seed(2)
  ag = xagent.Agent(cfg)
  n = 4
  rng = np.random.default_rng(3)
  img = torch.from_numpy((rng.random((n, 64, 64, 3))**2.2).astype(np.float32))
  states = torch.zeros(n, 11)
  z = torch.from_numpy(rng.random((n, 131)).astype(np.float32))
  masks = [torch.from_numpy((rng.random((n, 4096)) < 0.5).astype(np.float32)) for _ in range(2)]
  with fake_hip():
    (out, new_states, surrogate, penalty), dbg, _ = ag((img, z, states), is_train=1, progress=0.5,
                                                      dropout_masks=masks)
    out.sum().backward()
  ids = dbg['selected_filter_ids'].numpy()
  with fake_hip(), torch.no_grad():
    feats = ag.filter_features(xagent.enrich_image_input(cfg, img, states), masks[0])
  for i in range(n):
    j = int(ids[i])
    filt = ag.filters[j]
    with fake_hip(), torch.no_grad():
      f, mraw = filt.extract_parameters(feats[i:i + 1])
      packed = filt.pack(filt.filter_param_regressor(f)).numpy().astype(np.float64)
    ref = fnp.apply_masked(filt.filter_id, img[i:i + 1].numpy().astype(np.float64), packed,
                           mraw.numpy().astype(np.float64), cfg.maximum_sharpness, cfg.minimum_strength)
    assert np.abs(out[i:i + 1].detach().numpy() - ref).max() < 2e-5
  for j in set(int(v) for v in ids):
    filt = ag.filters[j]
    p = filt.get_num_filter_parameters()
    assert float(filt.fc2.weight.grad[p:].abs().max()) > 0.0  # the 6 mask rows


def test_fused_sequence_host_op_matches_stepwise_composition():
  """filters.fused_sequence (host side, the HIP calls replaced by the oracle): the value and the gradients of a fixed
  per-image sequence equal those of the same filters applied one after the other through pixel_filter; id -1 zeroes
  the image from its step on and blocks the gradient to earlier steps."""
  from exposure_amd import filters as F
  from exposure_amd import synthetic
  rng = np.random.default_rng(3)
  n, steps = 3, 4
  x = torch.from_numpy(synthetic.make_images(rng, (n, 6, 5, 3), np.float32)).double()
  ids = np.array([[0, 4, 3, 7], [5, 1, 2, 6], [1, -1, 0, 4]], dtype=np.int32)
  p = np.zeros((n, steps, 24), dtype=np.float32)
  for i in range(n):
    for st in range(steps):
      if ids[i, st] >= 0:
        p[i, st, :F._cabi.NUM_PARAMS[ids[i, st]]] = synthetic.make_params(rng, int(ids[i, st]), 1)[0]
  w = torch.from_numpy(rng.standard_normal((n, 6, 5, 3)))
  with fake_hip():
    xa = x.clone().requires_grad_(True)
    pa = torch.from_numpy(p).requires_grad_(True)
    ya = F.fused_sequence(xa, pa, torch.from_numpy(ids))
    (ya * w).sum().backward()
    xb = x.clone().requires_grad_(True)
    pb = torch.from_numpy(p).requires_grad_(True)
    rows = []
    for i in range(n):
      cur = xb[i:i + 1]
      for st in range(steps):
        fid = int(ids[i, st])
        cur = cur * 0.0 if fid < 0 else F.pixel_filter(fid, cur, pb[i:i + 1, st, :F._cabi.NUM_PARAMS[fid]])
      rows.append(cur)
    yb = torch.cat(rows)
    (yb * w).sum().backward()
  np.testing.assert_allclose(ya.detach().numpy(), yb.detach().numpy(), rtol=1e-6, atol=1e-7)
  np.testing.assert_allclose(xa.grad.numpy(), xb.grad.numpy(), rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(pa.grad.numpy(), pb.grad.numpy(), rtol=1e-4, atol=1e-5)
  assert float(xa.grad[2].abs().max()) == 0.0 and float(pa.grad[2, :2].abs().max()) == 0.0


