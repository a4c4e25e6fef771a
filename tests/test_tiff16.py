import numpy as np
import pytest

from exposure_amd import tiff16


@pytest.mark.parametrize('shape,dtype', [((5, 7, 3), np.uint16), ((4, 4, 3), np.uint8), ((3, 9, 4), np.uint16)])
def test_tiff_roundtrip(tmp_path, shape, dtype):
  rng = np.random.default_rng(0)
  img = rng.integers(0, np.iinfo(dtype).max, shape).astype(dtype)
  p = str(tmp_path / 'a.tif')
  tiff16.write_tiff(p, img)
  back = tiff16.read_tiff(p)
  assert back.dtype == dtype and np.array_equal(back, img)


def test_read_tiff16_scales_to_unit_range(tmp_path):
  img = np.zeros((2, 2, 3), dtype=np.uint16)
  img[0, 0] = 65535
  img[1, 1] = 32768
  p = str(tmp_path / 'b.tif')
  tiff16.write_tiff(p, img)
  f = tiff16.read_tiff16(p)
  assert f.dtype == np.float32 and f[0, 0, 0] == 1.0 and abs(f[1, 1, 2] - 32768 / 65535) < 1e-7


def test_big_endian_and_pil_cross_check(tmp_path):
  PIL = pytest.importorskip('PIL.Image')
  img = (np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3) * 3)
  p = str(tmp_path / 'c.tif')
  PIL.fromarray(img).save(p, compression=None)
  assert np.array_equal(tiff16.read_tiff(p), img)
  p2 = str(tmp_path / 'd.tif')
  tiff16.write_tiff(p2, img)
  assert np.array_equal(np.asarray(PIL.open(p2)), img)
