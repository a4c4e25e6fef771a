"""TF-1 checkpoint (V2 tensor bundle) reader / writer: format constants, round trips, corruption, the
``GAN.restore`` / ``saver.save`` mirror (net.py:271,380-384,405-407; evaluate.py:27-28)."""
import os
import struct

import numpy as np
import pytest
import torch

from exposure_amd import checkpoint, tf_bundle
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN


def test_crc32c_known_answers():
  # RFC 3720 B.4 / the iSCSI test vectors every CRC-32C implementation quotes
  assert tf_bundle.crc32c(b'123456789') == 0xE3069283
  assert tf_bundle.crc32c(bytes(32)) == 0x8A9136AA
  assert tf_bundle.crc32c(b'\xff' * 32) == 0x62A8AB43
  assert tf_bundle.crc32c(bytes(range(32))) == 0x46DD794E
  assert tf_bundle.crc32c(b'') == 0
  # LevelDB's crc32c_test: Mask(Crc("foo")) differs from the crc, and unmasking inverts it
  c = tf_bundle.crc32c(b'foo')
  assert tf_bundle.mask_crc(c) != c and tf_bundle.unmask_crc(tf_bundle.mask_crc(c)) == c
  assert tf_bundle.unmask_crc(tf_bundle.unmask_crc(tf_bundle.mask_crc(tf_bundle.mask_crc(c)))) == c


def test_crc32c_lane_parallel_path_equals_the_byte_loop():
  rng = np.random.default_rng(0)
  for n in (1 << 14, (1 << 14) + 1, 70001, 4096 * 9 + 4095):
    data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    assert tf_bundle.crc32c(data) == tf_bundle._crc_serial(data), n


def test_table_layout_constants(tmp_path):
  """The bytes a reader written against the published table format must find: footer magic, handles, block trailer."""
  path = str(tmp_path / 't.index')
  tf_bundle.write_table(path, [(b'', b'H'), (b'a/b', b'1'), (b'a/c', b'22')])
  buf = open(path, 'rb').read()
  assert buf[-8:] == bytes([0x57, 0xfb, 0x80, 0x8b, 0x24, 0x75, 0x47, 0xdb])  # kTableMagicNumber, little-endian
  assert len(buf) >= 48
  # first data block starts at 0: entry "" -> "H" is  shared 0 | non_shared 0 | value_len 1 | 'H'
  assert buf[:4] == bytes([0, 0, 1]) + b'H'
  # second entry: no shared prefix with "", third shares "a/" with the second
  assert buf[4:11] == bytes([0, 3, 1]) + b'a/b' + b'1'
  assert buf[11:17] == bytes([2, 1, 2]) + b'c' + b'22'
  # block trailer: type 0 + masked crc of contents + type
  pos = 0
  footer = buf[-48:]
  _mo, pos = tf_bundle._get_varint(footer, pos)
  _ms, pos = tf_bundle._get_varint(footer, pos)
  io, pos = tf_bundle._get_varint(footer, pos)
  isz, pos = tf_bundle._get_varint(footer, pos)
  assert buf[io + isz] == 0
  stored = struct.unpack_from('<I', buf, io + isz + 1)[0]
  assert stored == tf_bundle.mask_crc(tf_bundle.crc32c(buf[io:io + isz + 1]))
  assert tf_bundle.read_table(path) == [(b'', b'H'), (b'a/b', b'1'), (b'a/c', b'22')]


def test_bundle_entry_wire_format():
  """BundleEntryProto / TensorShapeProto bytes spelled out by hand (field numbers of tensor_bundle.proto)."""
  got = tf_bundle._encode_entry(1, (4, 4, 14, 32), 0, 256, 28672, 0xDEADBEEF)
  want = bytes([0x08, 0x01,  # dtype = DT_FLOAT
                0x12, 0x10, 0x12, 0x02, 0x08, 0x04, 0x12, 0x02, 0x08, 0x04, 0x12, 0x02, 0x08, 0x0E, 0x12, 0x02, 0x08, 0x20,
                0x20, 0x80, 0x02,  # offset = 256
                0x28, 0x80, 0xE0, 0x01,  # size = 28672
                0x35, 0xEF, 0xBE, 0xAD, 0xDE])  # crc32c fixed32
  assert got == want
  e = tf_bundle._decode_entry(want)
  assert (e['dtype'], e['shape'], e['shard'], e['offset'], e['size'], e['crc']) == (1, (4, 4, 14, 32), 0, 256, 28672,
                                                                                  0xDEADBEEF)
  assert tf_bundle._decode_header(tf_bundle._encode_header(1)) == dict(num_shards=1, endianness=0)
  assert tf_bundle._encode_header(1) == bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])


def test_bundle_roundtrip_many_variables(tmp_path):
  rng = np.random.default_rng(3)
  tensors = {}
  for i in range(300):  # several data blocks, every restart-interval / prefix-compression case
    name = 'scope_%d/layer_%d/%s' % (i % 7, i, 'weights' if i % 2 else 'biases')
    shape = [(), (5,), (3, 4), (2, 3, 4, 5)][i % 4]
    tensors[name] = rng.standard_normal(shape).astype([np.float32, np.float64, np.float16][i % 3])
  tensors['global_step'] = np.asarray(20000, dtype=np.int64)
  tensors['flags'] = np.array([True, False, True])
  tensors['big'] = rng.standard_normal((4096, 40)).astype(np.float32)  # the lane-parallel CRC path
  prefix = str(tmp_path / 'model.ckpt-7')
  tf_bundle.write_bundle(prefix, tensors)
  assert sorted(os.listdir(tmp_path)) == ['model.ckpt-7.data-00000-of-00001', 'model.ckpt-7.index']
  back = tf_bundle.read_bundle(prefix)
  assert set(back) == set(tensors)
  for k, v in tensors.items():
    assert back[k].dtype == v.dtype and back[k].shape == v.shape
    np.testing.assert_array_equal(back[k], v)
  listing = tf_bundle.list_variables(prefix)
  assert listing['big'] == (np.dtype('<f4'), (4096, 40)) and listing['global_step'] == (np.dtype('<i8'), ())
  some = tf_bundle.read_bundle(prefix, names=['big', 'flags', 'not there'])
  assert set(some) == {'big', 'flags'}


def test_bundle_corruption_is_detected(tmp_path):
  prefix = str(tmp_path / 'm')
  tf_bundle.write_bundle(prefix, {'a': np.arange(12, dtype=np.float32).reshape(3, 4), 'b': np.ones(5)})
  data = prefix + '.data-00000-of-00001'
  raw = bytearray(open(data, 'rb').read())
  raw[5] ^= 1
  open(data, 'wb').write(raw)
  with pytest.raises(tf_bundle.BundleError, match='a: tensor checksum'):
    tf_bundle.read_bundle(prefix)
  assert tf_bundle.read_bundle(prefix, names=['b'])['b'].sum() == 5.0  # the intact tensor still loads
  assert tf_bundle.read_bundle(prefix, verify=False)['a'].shape == (3, 4)
  open(data, 'wb').write(raw[:20])
  with pytest.raises(tf_bundle.BundleError, match='runs past the end'):
    tf_bundle.read_bundle(prefix, verify=False)
  idx = bytearray(open(prefix + '.index', 'rb').read())
  idx[2] ^= 0x40
  open(prefix + '.index', 'wb').write(idx)
  with pytest.raises(tf_bundle.BundleError, match='checksum mismatch'):
    tf_bundle.read_bundle(prefix)
  idx[-1] ^= 1
  open(prefix + '.index', 'wb').write(idx)
  with pytest.raises(tf_bundle.BundleError, match='bad magic'):
    tf_bundle.read_bundle(prefix)
  with pytest.raises(FileNotFoundError):
    tf_bundle.read_bundle(str(tmp_path / 'absent'))
  open(str(tmp_path / 'v1.ckpt'), 'wb').write(b'x' * 100)
  with pytest.raises(tf_bundle.BundleError, match='V1 checkpoint format'):
    tf_bundle.read_bundle(str(tmp_path / 'v1.ckpt'))


def test_unsupported_features_fail_loudly(tmp_path):
  prefix = str(tmp_path / 'm')
  # a partitioned variable: BundleEntryProto.slices (field 7) present
  entry = tf_bundle._encode_entry(1, (2,), 0, 0, 8, 0) + bytes([0x3A, 0x00])
  tf_bundle.write_table(prefix + '.index', [(b'', tf_bundle._encode_header(1)), (b'v', entry)])
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(8))
  with pytest.raises(tf_bundle.BundleError, match='partitioned'):
    tf_bundle.read_bundle(prefix)
  # DT_STRING
  entry = tf_bundle._encode_entry(7, (1,), 0, 0, 8, 0)
  tf_bundle.write_table(prefix + '.index', [(b'', tf_bundle._encode_header(1)), (b'v', entry)])
  with pytest.raises(tf_bundle.BundleError, match='unsupported dtype'):
    tf_bundle.read_bundle(prefix)
  # big-endian header
  tf_bundle.write_table(prefix + '.index', [(b'', tf_bundle._encode_header(1) + bytes([0x10, 0x01]))])
  with pytest.raises(tf_bundle.BundleError, match='big-endian'):
    tf_bundle.read_bundle(prefix)
  # compressed block: flip the type byte and re-checksum
  tf_bundle.write_table(prefix + '.index', [(b'', tf_bundle._encode_header(1))])
  buf = bytearray(open(prefix + '.index', 'rb').read())
  footer = buf[-48:]
  pos = 0
  _mo, pos = tf_bundle._get_varint(footer, pos)
  _ms, pos = tf_bundle._get_varint(footer, pos)
  io, pos = tf_bundle._get_varint(footer, pos)
  isz, pos = tf_bundle._get_varint(footer, pos)
  buf[io + isz] = 1  # kSnappyCompression
  buf[io + isz + 1:io + isz + 5] = struct.pack('<I', tf_bundle.mask_crc(tf_bundle.crc32c(bytes(buf[io:io + isz + 1]))))
  open(prefix + '.index', 'wb').write(buf)
  with pytest.raises(tf_bundle.BundleError, match='compressed'):
    tf_bundle.read_bundle(prefix)
  with pytest.raises(tf_bundle.BundleError, match='strictly increasing'):
    tf_bundle.write_table(prefix + '.index', [(b'b', b''), (b'a', b'')])


def test_restore_mirrors_the_reference_call(tmp_path):
  """save(dir, it) then restore(dir, it) on a fresh GAN reproduces every parameter; an Agent alone restores the
  generator's variables; extra variables in the checkpoint (optimizer slots) are ignored; missing ones raise."""
  torch.manual_seed(0)
  gan = GAN(make_cfg())
  model_dir = str(tmp_path / 'models' / 'example' / 'test')
  prefix = checkpoint.save(gan, model_dir, 20000)
  assert prefix.endswith('model.ckpt-20000')
  assert 'model_checkpoint_path: "model.ckpt-20000"' in open(os.path.join(model_dir, 'checkpoint')).read()
  names = tf_bundle.list_variables(prefix)
  assert names['generator/Conv/weights'] == (np.dtype('<f4'), (4, 4, 14, 32))  # HWIO
  assert names['generator/filter_0/fc1/weights'] == (np.dtype('<f4'), (4096, 128))  # (in, out)
  torch.manual_seed(1)
  gan2 = GAN(make_cfg())
  assert not torch.equal(gan.critic.fc1.weight, gan2.critic.fc1.weight)
  assert checkpoint.restore(gan2, model_dir) == []  # default ckpt = 20000 (evaluate.py:28)
  for a, b in zip(gan.parameters(), gan2.parameters()):
    assert torch.equal(a, b)
  from exposure_amd.agent import Agent
  agent = Agent(make_cfg())
  assert checkpoint.restore(agent, model_dir, 20000) == []
  for a, b in zip(gan.generator.parameters(), agent.parameters()):
    assert torch.equal(a, b)
  # a checkpoint as TF writes it also holds Adam slots etc.; and one that lacks a variable fails by name
  d = checkpoint.export_tf_dict(gan)
  d['generator/Conv/weights/Adam'] = np.zeros((4, 4, 14, 32), np.float32)
  d['beta1_power'] = np.float32(0.5)
  del d['critic/fully_connected_1/biases']
  tf_bundle.write_bundle(checkpoint.checkpoint_prefix(model_dir, 500), d)
  with pytest.raises(KeyError, match='critic/fully_connected_1/biases'):
    checkpoint.restore(gan2, model_dir, 500)
  assert checkpoint.restore(gan2, model_dir, 500, strict=False) == ['critic/fully_connected_1/biases']
  assert checkpoint.restore(agent, model_dir, 500) == []
  d['generator/Conv/biases'] = np.zeros(31, np.float32)
  tf_bundle.write_bundle(checkpoint.checkpoint_prefix(model_dir, 501), d)
  with pytest.raises(ValueError, match='generator/Conv/biases: shape'):
    checkpoint.restore(agent, model_dir, 501)


def test_table_round_trip_property(tmp_path):
  """Random key sets (shared prefixes, empty values, keys longer than a block) through write_table / read_table, and
  random varints through the wire helpers."""
  from hypothesis import given, settings, strategies as st
  import itertools
  counter = itertools.count()

  @settings(max_examples=60, deadline=None)
  @given(st.dictionaries(st.binary(min_size=0, max_size=40), st.binary(min_size=0, max_size=300), min_size=1, max_size=80),
         st.sampled_from([64, 512, 4096]))
  def table(items, block_size):
    path = str(tmp_path / ('t%d.index' % next(counter)))
    ordered = sorted(items.items())
    tf_bundle.write_table(path, ordered, block_size=block_size)
    assert tf_bundle.read_table(path) == ordered

  @settings(max_examples=200, deadline=None)
  @given(st.integers(min_value=0, max_value=(1 << 64) - 1))
  def varint(v):
    out = bytearray()
    tf_bundle._put_varint(out, v)
    got, pos = tf_bundle._get_varint(bytes(out), 0)
    assert got == v and pos == len(out) <= 10

  table()
  varint()
