import colorsys

import numpy as np
import pytest
import torch

from exposure_amd import checkpoint, metrics
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN


def test_hls_saturation_matches_colorsys():
  rng = np.random.default_rng(0)
  img = rng.random((2, 5, 7, 3)).astype(np.float32)
  img[0, 0, 0] = 0.5  # grey -> S = 0
  s = metrics.hls_saturation(torch.from_numpy(img)).numpy()
  for idx in np.ndindex(2, 5, 7):
    _h, _l, ref = colorsys.rgb_to_hls(*[float(v) for v in img[idx]])
    assert s[idx] == pytest.approx(ref, abs=1e-6)


def test_histogram_intersection_properties():
  rng = np.random.default_rng(1)
  a = torch.from_numpy(rng.random((200, 16, 16, 3)).astype(np.float32)**2.0)
  b = torch.from_numpy(rng.random((200, 16, 16, 3)).astype(np.float32)**0.5)
  ints, avg = metrics.histogram_intersection(a, a)
  assert all(abs(v - 1.0) < 1e-6 for v in ints) and abs(avg - 1.0) < 1e-6
  ints2, avg2 = metrics.histogram_intersection(a, b)
  assert all(0.0 <= v <= 1.0 for v in ints2) and avg2 < 0.9
  # numpy restatement of histogram_intersection.py:15-33 (cv2 HLS replaced by the formula)
  st = metrics.get_statistics(a).numpy()
  lum = (a.numpy().clip(0, 1) * np.array([0.27, 0.67, 0.06], dtype=np.float32)).sum(-1)
  np.testing.assert_allclose(st[:, 0], lum.reshape(200, -1).mean(1), rtol=1e-5)
  np.testing.assert_allclose(st[:, 1], lum.reshape(200, -1).std(1) * 2, rtol=1e-4)
  h_np, _ = np.histogram(st[:, 0], bins=32, range=(0.0, 1.0))
  np.testing.assert_allclose(metrics.calc_hist(torch.from_numpy(st[:, 0])).numpy(), h_np / 200.0, atol=1e-7)


def test_metric_cli_reads_folders_like_the_script(tmp_path, capsys):
  """histogram_intersection.py:36-76: 16 patches of 64x64 per file (4 square crops -> 80x80 -> 4 patches each); a
  folder against itself with the same sampling scores 100 %, a brighter copy less."""
  import random
  from PIL import Image
  rng = np.random.default_rng(0)
  a, b = tmp_path / 'out', tmp_path / 'target'
  a.mkdir(), b.mkdir()
  for i in range(6):
    img = (rng.random((120 + 8 * i, 160, 3))**2.0 * 255).astype(np.uint8)
    Image.fromarray(img).save(str(a / ('%d.png' % i)))
    Image.fromarray((255 * (img / 255.0)**0.5).astype(np.uint8)).save(str(b / ('%d.png' % i)))
  pa = metrics.read_images(str(a), rng=random.Random(1))
  assert pa.shape == (6 * 16, 64, 64, 3) and pa.dtype == torch.float32 and 0.0 <= float(pa.min()) and float(pa.max()) <= 1.0
  same, avg_same = metrics.histogram_intersection(pa, metrics.read_images(str(a), rng=random.Random(1)))
  assert abs(avg_same - 1.0) < 1e-6
  ints, avg = metrics.main([str(a), str(b)])
  assert 0.0 <= avg < 0.9
  out = capsys.readouterr().out
  assert 'Hist. Inter.:' in out and 'Avg:' in out
  assert metrics.read_images(str(a), tag='3.').shape[0] == 16
  with pytest.raises(SystemExit):
    metrics.main([str(a)])


def test_tf_layout_roundtrip_and_conv_equivalence():
  torch.manual_seed(0)
  gan = GAN(make_cfg())
  d = checkpoint.export_tf_dict(gan)
  assert d['generator/Conv/weights'].shape == (4, 4, 14, 32)  # HWIO, C_in = 3 + 11
  assert d['generator/filter_7/fc2/weights'].shape == (128, 24 + 6)
  assert d['rl_value/critic/Conv/weights'].shape == (4, 4, 17, 32)
  assert d['critic/fully_connected_1/weights'].shape == (128, 1)
  assert len(d) == sum(1 for _ in gan.parameters())
  gan2 = GAN(make_cfg())
  assert checkpoint.load_tf_dict(gan2, d) == []
  for a, b in zip(gan.parameters(), gan2.parameters()):
    assert torch.equal(a, b)
  # the HWIO kernel applied TF-style (NHWC, SAME, stride 2) equals the torch conv with the OIHW copy
  conv = gan.critic.convs[0]
  x = torch.randn(2, 8, 8, conv.in_channels)
  want = conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).detach().numpy()
  w = d['critic/Conv/weights']  # (kh, kw, cin, cout)
  xp = np.pad(x.numpy(), ((0, 0), (1, 1), (1, 1), (0, 0)))
  got = np.zeros_like(want)
  for oy in range(4):
    for ox in range(4):
      patch = xp[:, 2 * oy:2 * oy + 4, 2 * ox:2 * ox + 4, :]
      got[:, oy, ox, :] = np.einsum('nhwc,hwco->no', patch, w) + d['critic/Conv/biases']
  np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
  with pytest.raises(KeyError):
    checkpoint.load_tf_dict(gan2, {})
