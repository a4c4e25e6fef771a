"""Reader / writer for TensorFlow's V2 checkpoint format ("tensor bundle"), without TensorFlow.

The reference keeps its weights in ``tf.train.Saver`` checkpoints (``net.py:271`` creates the saver,
``net.py:380-384`` saves ``models/<cfg>/<name>/model.ckpt-<iter>``, ``net.py:405-407`` /
``evaluate.py:27-28`` restore iteration 20000).  ``Saver`` has written the V2 format by default since
TF 0.12, i.e. for every TF-1 release the reference can have run on:

``<prefix>.index``
    an SSTable in TensorFlow's copy of the LevelDB table format (``tensorflow/core/lib/io/table*``,
    format constants restated below): sorted string keys -> values.  Key ``""`` holds a
    ``BundleHeaderProto``, every other key is a variable name whose value is a ``BundleEntryProto``
    (dtype, shape, shard, offset, size, masked CRC-32C of the tensor bytes).
``<prefix>.data-SSSSS-of-NNNNN``
    the tensors' raw little-endian bytes, back to back.

Nothing of this is in ``/root/reference`` (TensorFlow is its un-vendored dependency; the format has been
frozen since 2016) -- the layout is restated from the published format: block = entries
``varint shared | varint non_shared | varint value_len | key suffix | value`` followed by the restart
array and its length; every block is followed by a 5-byte trailer (compression type, masked CRC-32C of
contents + type); the 48-byte footer holds the metaindex and index block handles and the magic
``0xdb4775248b80fb57``.  The writer exists so that weights trained here can be handed back to a TF graph through
``tf.train.Saver(var_list=<trainable variables>).restore`` (the reference's default saver also expects Adam slots,
step counters and the EMA shadow, which are not written: exposure_amd/checkpoint.py::save) and so that the reader has files to be tested on (no TF-written bundle
is available in this image: the reader's parity against TensorFlow itself is unpinned, its parity
against the published format constants is tested in ``tests/test_tf_bundle.py``).

Not supported (a clear error, never a silent skip): Snappy-compressed index blocks (``BundleWriter``
writes them uncompressed), partitioned variables (``slices``), big-endian bundles, string tensors, the
pre-0.12 V1 single-file format.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER_LEN = 5
CRC_MASK_DELTA = 0xa282ead8
RESTART_INTERVAL = 16
HEADER_KEY = b''

# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
          6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('bool'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
          22: np.dtype('<u4'), 23: np.dtype('<u8')}
DTYPE_IDS = {v: k for k, v in DTYPES.items()}


class BundleError(ValueError):
  pass


# ------------------------------------------------------------------------------- CRC-32C (Castagnoli)
_POLY = 0x82F63B78  # reflected 0x1EDC6F41


def _make_table():
  t = np.zeros(256, dtype=np.uint32)
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ (_POLY if c & 1 else 0)
    t[i] = c
  return t


_TABLE = _make_table()
_TABLE_LIST = [int(v) for v in _TABLE]


def _crc_serial(data, crc=0):
  c = crc ^ 0xFFFFFFFF
  tab = _TABLE_LIST
  for b in data:
    c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def _shift_matrix(nbytes):
  """32 column images of the operator "append nbytes zero bytes" on the raw CRC register (zlib's crc32_combine
  construction with the Castagnoli polynomial): square the one-zero-bit operator log2(8 nbytes) times."""
  def times(mat, vec):
    s = 0
    i = 0
    while vec:
      if vec & 1:
        s ^= mat[i]
      vec >>= 1
      i += 1
    return s

  def square(mat):
    return [times(mat, mat[i]) for i in range(32)]

  one_bit = [_POLY] + [1 << (i - 1) for i in range(1, 32)]  # one zero BIT
  op = one_bit
  result = None  # identity
  n = nbytes * 8
  while n:
    if n & 1:
      result = op if result is None else [times(op, result[i]) for i in range(32)]
    n >>= 1
    if n:
      op = square(op)
  return result, times


def crc32c(data):
  """CRC-32C of a bytes-like object.  Large inputs are cut into equal lanes whose CRCs advance together as one numpy
  vector (a table step per byte position), then folded with the zero-append operator -- a 25 MB checkpoint takes a
  fraction of a second instead of a pure-Python byte loop's minute."""
  buf = np.frombuffer(memoryview(data).cast('B'), dtype=np.uint8)
  n = buf.size
  if n < 1 << 14:
    return _crc_serial(buf.tolist())
  lanes = 4096
  seg = n // lanes
  body = buf[:seg * lanes].reshape(lanes, seg)
  c = np.full(lanes, 0xFFFFFFFF, dtype=np.uint32)
  cols = np.ascontiguousarray(body.T)
  for j in range(seg):
    c = _TABLE[(c ^ cols[j]) & 0xFF] ^ (c >> 8)
  c ^= 0xFFFFFFFF
  mat, times = _shift_matrix(seg)
  acc = 0
  for v in c.tolist():
    acc = times(mat, acc) ^ v
  return _crc_serial(buf[seg * lanes:].tolist(), acc)


def mask_crc(crc):
  return (((crc >> 15) | (crc << 17)) + CRC_MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
  rot = (masked - CRC_MASK_DELTA) & 0xFFFFFFFF
  return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------- varints / protobuf wire format
def _put_varint(out, v):
  v &= (1 << 64) - 1
  while v >= 0x80:
    out.append((v & 0x7F) | 0x80)
    v >>= 7
  out.append(v)


def _get_varint(buf, pos):
  shift = 0
  v = 0
  while True:
    if pos >= len(buf):
      raise BundleError('truncated varint')
    b = buf[pos]
    pos += 1
    v |= (b & 0x7F) << shift
    if not b & 0x80:
      return v, pos
    shift += 7
    if shift > 63:
      raise BundleError('varint longer than 64 bits')


def _fields(buf):
  """(field number, wire type, value) of one protobuf message; values are ints or bytes."""
  pos = 0
  while pos < len(buf):
    tag, pos = _get_varint(buf, pos)
    num, wt = tag >> 3, tag & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wt == 2:
      ln, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + ln])
      if len(v) != ln:
        raise BundleError('truncated length-delimited field')
      pos += ln
    elif wt == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise BundleError('unsupported protobuf wire type %d' % wt)
    yield num, wt, v


def _signed64(v):
  return v - (1 << 64) if v >= 1 << 63 else v


def _tag(out, num, wt):
  _put_varint(out, (num << 3) | wt)


def _encode_shape(shape):
  out = bytearray()
  for d in shape:
    dim = bytearray()
    _tag(dim, 1, 0)
    _put_varint(dim, int(d))
    _tag(out, 2, 2)
    _put_varint(out, len(dim))
    out += dim
  return bytes(out)


def _decode_shape(buf):
  shape = []
  for num, _wt, v in _fields(buf):
    if num == 2:
      size = 0
      for n2, _w2, v2 in _fields(v):
        if n2 == 1:
          size = _signed64(v2)
      shape.append(size)
    elif num == 3 and v:
      raise BundleError('tensor of unknown rank')
  return tuple(shape)


def _encode_entry(dtype_id, shape, shard, offset, size, crc_masked):
  out = bytearray()
  _tag(out, 1, 0)
  _put_varint(out, dtype_id)
  sh = _encode_shape(shape)
  _tag(out, 2, 2)
  _put_varint(out, len(sh))
  out += sh
  if shard:
    _tag(out, 3, 0)
    _put_varint(out, shard)
  if offset:
    _tag(out, 4, 0)
    _put_varint(out, offset)
  _tag(out, 5, 0)
  _put_varint(out, size)
  _tag(out, 6, 5)
  out += struct.pack('<I', crc_masked)
  return bytes(out)


def _decode_entry(buf):
  e = dict(dtype=0, shape=(), shard=0, offset=0, size=0, crc=None, slices=0)
  for num, _wt, v in _fields(buf):
    if num == 1:
      e['dtype'] = v
    elif num == 2:
      e['shape'] = _decode_shape(v)
    elif num == 3:
      e['shard'] = v
    elif num == 4:
      e['offset'] = v
    elif num == 5:
      e['size'] = v
    elif num == 6:
      e['crc'] = v
    elif num == 7:
      e['slices'] += 1
  return e


def _encode_header(num_shards):
  out = bytearray()
  _tag(out, 1, 0)
  _put_varint(out, num_shards)
  ver = bytearray()
  _tag(ver, 1, 0)
  _put_varint(ver, 1)  # VersionDef.producer = kTensorBundleVersion
  _tag(out, 3, 2)
  _put_varint(out, len(ver))
  out += ver
  return bytes(out)  # endianness LITTLE = 0 is the default and is not serialised


def _decode_header(buf):
  h = dict(num_shards=0, endianness=0)
  for num, _wt, v in _fields(buf):
    if num == 1:
      h['num_shards'] = v
    elif num == 2:
      h['endianness'] = v
  return h


# ------------------------------------------------------------------------------- the table
def _read_block(buf, offset, size, what):
  end = offset + size + BLOCK_TRAILER_LEN
  if end > len(buf):
    raise BundleError('%s block [%d, %d) runs past the end of the index file' % (what, offset, end))
  contents = buf[offset:offset + size]
  ctype = buf[offset + size]
  stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
  if unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
    raise BundleError('%s block at %d: checksum mismatch' % (what, offset))
  if ctype != 0:
    raise BundleError('%s block at %d is compressed (type %d); only uncompressed tables are supported' %
                      (what, offset, ctype))
  return contents


def _block_entries(block):
  if len(block) < 4:
    raise BundleError('block shorter than its restart count')
  n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * n_restarts
  if limit < 0:
    raise BundleError('bad restart array')
  pos = 0
  key = b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise BundleError('corrupt block entry')
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    yield key, bytes(block[pos:pos + vlen])
    pos += vlen


def read_table(path):
  """All (key, value) pairs of an uncompressed TF / LevelDB table file, in key order."""
  with open(path, 'rb') as f:
    buf = f.read()
  if len(buf) < FOOTER_LEN:
    raise BundleError('%s: too short for a table footer' % path)
  footer = buf[-FOOTER_LEN:]
  if struct.unpack_from('<Q', footer, FOOTER_LEN - 8)[0] != TABLE_MAGIC:
    raise BundleError('%s: not a TensorFlow table (bad magic) -- a V1 checkpoint or a data shard?' % path)
  pos = 0
  _mo, pos = _get_varint(footer, pos)
  _ms, pos = _get_varint(footer, pos)
  io, pos = _get_varint(footer, pos)
  isz, pos = _get_varint(footer, pos)
  out = []
  for _last_key, handle in _block_entries(_read_block(buf, io, isz, 'index')):
    bo, p2 = _get_varint(handle, 0)
    bs, _ = _get_varint(handle, p2)
    out.extend(_block_entries(_read_block(buf, bo, bs, 'data')))
  return out


class _BlockBuilder:
  def __init__(self):
    self.buf = bytearray()
    self.restarts = [0]
    self.count = 0
    self.last = b''

  def add(self, key, value):
    shared = 0
    if self.count % RESTART_INTERVAL == 0 and self.count:
      self.restarts.append(len(self.buf))
    elif self.count:
      m = min(len(key), len(self.last))
      while shared < m and key[shared] == self.last[shared]:
        shared += 1
    _put_varint(self.buf, shared)
    _put_varint(self.buf, len(key) - shared)
    _put_varint(self.buf, len(value))
    self.buf += key[shared:]
    self.buf += value
    self.last = key
    self.count += 1

  def finish(self):
    out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
    return out


def _emit_block(out, contents):
  offset = len(out)
  out += contents
  out.append(0)  # kNoCompression
  out += struct.pack('<I', mask_crc(crc32c(contents + b'\0')))
  handle = bytearray()
  _put_varint(handle, offset)
  _put_varint(handle, len(contents))
  return bytes(handle)


def write_table(path, items, block_size=4096):
  """items: (key bytes, value bytes) in strictly increasing key order."""
  out = bytearray()
  index = _BlockBuilder()
  blk = _BlockBuilder()
  prev = None
  for key, value in items:
    if prev is not None and key <= prev:
      raise BundleError('table keys must be strictly increasing')
    blk.add(key, value)
    prev = key
    if len(blk.buf) >= block_size:
      index.add(key, _emit_block(out, blk.finish()))
      blk = _BlockBuilder()
  if blk.count:
    index.add(prev, _emit_block(out, blk.finish()))
  meta = _emit_block(out, _BlockBuilder().finish())
  idx = _emit_block(out, index.finish())
  footer = bytearray(meta + idx)
  footer += b'\0' * (FOOTER_LEN - 8 - len(footer))
  footer += struct.pack('<Q', TABLE_MAGIC)
  out += footer
  with open(path, 'wb') as f:
    f.write(out)


# ------------------------------------------------------------------------------- the bundle
def _shard_name(prefix, shard, num_shards):
  return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def list_variables(prefix):
  """{name: (numpy dtype, shape)} of the bundle ``<prefix>.index``."""
  out = {}
  for key, value in read_table(prefix + '.index'):
    if key == HEADER_KEY:
      continue
    e = _decode_entry(value)
    out[key.decode('utf-8')] = (DTYPES.get(e['dtype']), e['shape'])
  return out


def read_bundle(prefix, names=None, verify=True):
  """{variable name: ndarray} of the checkpoint ``prefix`` (what ``saver.restore(sess, prefix)`` is given,
  e.g. ``models/<cfg>/<name>/model.ckpt-20000``).  ``names``: read only these (missing ones are simply absent
  from the result -- the caller decides what is required).  ``verify`` checks each tensor's CRC-32C."""
  index = prefix + '.index'
  if not os.path.exists(index):
    if os.path.exists(prefix):
      raise BundleError('%s is a single file: the pre-0.12 V1 checkpoint format is not supported' % prefix)
    raise FileNotFoundError(index)
  entries = read_table(index)
  if not entries or entries[0][0] != HEADER_KEY:
    raise BundleError('%s: no bundle header entry' % index)
  header = _decode_header(entries[0][1])
  if header['endianness'] != 0:
    raise BundleError('big-endian bundles are not supported')
  want = None if names is None else set(names)
  shards = {}
  out = {}
  for key, value in entries[1:]:
    name = key.decode('utf-8')
    if want is not None and name not in want:
      continue
    e = _decode_entry(value)
    if e['slices']:
      raise BundleError('%s is a partitioned variable (slices); not supported' % name)
    dt = DTYPES.get(e['dtype'])
    if dt is None:
      raise BundleError('%s: unsupported dtype enum %d' % (name, e['dtype']))
    if e['shard'] >= max(header['num_shards'], 1):
      raise BundleError('%s: shard %d of %d' % (name, e['shard'], header['num_shards']))
    count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
    if count * dt.itemsize != e['size']:
      raise BundleError('%s: %d bytes for shape %s of %s' % (name, e['size'], e['shape'], dt))
    if e['shard'] not in shards:
      shards[e['shard']] = np.memmap(_shard_name(prefix, e['shard'], header['num_shards']), dtype=np.uint8, mode='r')
    data = shards[e['shard']]
    if e['offset'] + e['size'] > data.size:
      raise BundleError('%s: [%d, %d) runs past the end of its data shard' % (name, e['offset'], e['offset'] + e['size']))
    raw = np.array(data[e['offset']:e['offset'] + e['size']])
    if verify and e['crc'] is not None:
      actual = crc32c(raw)
      # BundleWriter stores the MASKED checksum (crc32c::Mask, as the table blocks do); a plain one is accepted too
      if unmask_crc(e['crc']) != actual and e['crc'] != actual:
        raise BundleError('%s: tensor checksum mismatch' % name)
    out[name] = raw.view(dt).reshape(e['shape'])
  return out


def write_bundle(prefix, tensors):
  """Writes ``{name: ndarray}`` as ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``."""
  os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
  items = [(HEADER_KEY, _encode_header(1))]
  offset = 0
  with open(_shard_name(prefix, 0, 1), 'wb') as f:
    for name in sorted(tensors, key=lambda s: s.encode('utf-8')):
      a = np.asarray(tensors[name])
      dt = a.dtype.newbyteorder('<') if a.dtype.itemsize > 1 else a.dtype  # the bundle is little-endian
      if dt not in DTYPE_IDS:
        raise BundleError('%s: dtype %s has no TensorFlow counterpart here' % (name, a.dtype))
      raw = np.ascontiguousarray(a.astype(dt, copy=False)).tobytes()
      if not name:
        raise BundleError('the empty name is the bundle header')
      items.append((name.encode('utf-8'), _encode_entry(DTYPE_IDS[dt], a.shape, 0, offset, len(raw),
                                                        mask_crc(crc32c(raw)))))
      f.write(raw)
      offset += len(raw)
  write_table(prefix + '.index', items)
