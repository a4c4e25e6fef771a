"""Dependency-free reader/writer for the subset of TIFF the reference's pipeline uses
(``util.py:311-323`` ``read_tiff16`` via ``tifffile``; the FiveK exports are uncompressed 16-bit RGB):
baseline TIFF, little- or big-endian, uncompressed strips, chunky RGB(A), 8 or 16 bits per sample.
``tifffile`` / ``cv2`` are not installed in this image."""
import struct

import numpy as np

_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8)}


def _read_ifd(buf, off, e):
  n, = struct.unpack_from(e + 'H', buf, off)
  tags = {}
  for i in range(n):
    tag, typ, cnt = struct.unpack_from(e + 'HHI', buf, off + 2 + 12 * i)
    fmt, size = _TYPES.get(typ, (None, 0))
    if fmt is None:
      continue
    total = size * cnt
    pos = off + 2 + 12 * i + 8
    if total > 4:
      pos, = struct.unpack_from(e + 'I', buf, pos)
    if typ == 5:
      vals = [struct.unpack_from(e + 'II', buf, pos + 8 * k) for k in range(cnt)]
    elif typ == 2:
      vals = [buf[pos:pos + cnt]]
    else:
      vals = list(struct.unpack_from(e + fmt * cnt, buf, pos))
    tags[tag] = vals
  return tags


def read_tiff(path):
  """-> uint8 / uint16 array (H, W, C)."""
  buf = open(path, 'rb').read()
  if buf[:2] == b'II':
    e = '<'
  elif buf[:2] == b'MM':
    e = '>'
  else:
    raise ValueError('not a TIFF file: %s' % path)
  magic, ifd = struct.unpack_from(e + 'HI', buf, 2)
  if magic != 42:
    raise ValueError('unsupported TIFF variant (BigTIFF?)')
  t = _read_ifd(buf, ifd, e)
  w, h = t[256][0], t[257][0]
  bps = t.get(258, [1])
  spp = t.get(277, [1])[0]
  if t.get(259, [1])[0] != 1:
    raise ValueError('compressed TIFF is not supported (only uncompressed strips)')
  if t.get(284, [1])[0] != 1 and spp > 1:
    raise ValueError('planar TIFF is not supported')
  if len(set(bps)) != 1 or bps[0] not in (8, 16):
    raise ValueError('only 8 or 16 bits per sample are supported, got %s' % (bps,))
  dt = np.dtype(np.uint8) if bps[0] == 8 else np.dtype(e + 'u2')
  offsets, counts = t[273], t.get(279)
  if counts is None:
    counts = [h * w * spp * dt.itemsize]
  data = b''.join(buf[o:o + c] for o, c in zip(offsets, counts))
  img = np.frombuffer(data, dtype=dt, count=h * w * spp).reshape(h, w, spp)
  return img.astype(dt.newbyteorder('='))


def read_tiff16(path):
  """util.py:311-323 + net.py:731: 16-bit TIFF -> float32 RGB in [0, 1] (``/ 65535``)."""
  img = read_tiff(path)
  if img.dtype != np.uint16:
    raise ValueError('expected a 16-bit TIFF')
  return (img[:, :, :3].astype(np.float32) / 65535.0)


def write_tiff(path, img):
  """Minimal uncompressed little-endian writer (tests / examples)."""
  img = np.ascontiguousarray(img)
  assert img.ndim == 3 and img.dtype in (np.uint8, np.uint16)
  h, w, c = img.shape
  bits = 8 * img.dtype.itemsize
  data = img.astype('<u2' if bits == 16 else np.uint8).tobytes()
  entries = []
  extra = b''
  base = 8 + len(data)
  n_tags = 10
  extra_off = base + 2 + 12 * n_tags + 4

  def add(tag, typ, vals):
    nonlocal extra
    fmt, size = _TYPES[typ]
    cnt = len(vals)
    raw = struct.pack('<' + fmt * cnt, *vals)
    if len(raw) <= 4:
      entries.append(struct.pack('<HHI', tag, typ, cnt) + raw.ljust(4, b'\x00'))
    else:
      entries.append(struct.pack('<HHII', tag, typ, cnt, extra_off + len(extra)))
      extra += raw

  add(256, 4, [w])
  add(257, 4, [h])
  add(258, 3, [bits] * c)
  add(259, 3, [1])
  add(262, 3, [2 if c >= 3 else 1])
  add(273, 4, [8])
  add(277, 3, [c])
  add(278, 4, [h])
  add(279, 4, [len(data)])
  add(284, 3, [1])
  assert len(entries) == n_tags
  with open(path, 'wb') as f:
    f.write(b'II' + struct.pack('<HI', 42, base))
    f.write(data)
    f.write(struct.pack('<H', n_tags) + b''.join(entries) + struct.pack('<I', 0))
    f.write(extra)
