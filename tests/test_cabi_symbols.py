"""CPU-only: the C-ABI library loads and exports every symbol include/exposure_hip.h declares;
argument validation (no compute without a GPU); product path refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

from exposure_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
  txt = open(os.path.join(ROOT, 'include', 'exposure_hip.h')).read()
  txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
  return sorted(set(re.findall(r'\b(expo_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
  assert header_symbols() == sorted(_cabi.SIGNATURES)


def test_library_exports_every_symbol():
  lib = ctypes.CDLL(_cabi.LIB_PATH)
  for name in header_symbols():
    assert hasattr(lib, name), name


def test_version_and_param_counts():
  lib = _cabi.load()
  assert lib.expo_version() == _cabi.EXPO_ABI_VERSION
  assert [lib.expo_num_filter_params(i) for i in range(9)] == list(_cabi.NUM_PARAMS)
  assert lib.expo_num_filter_params(-1) == -1 and lib.expo_num_filter_params(9) == -1


def test_argument_validation_without_gpu():
  lib = _cabi.load()
  assert lib.expo_filter_fwd(9, None, None, None, 1, 1, 1, 0, None) == -1
  assert b'filter_id' in lib.expo_last_error()
  assert lib.expo_filter_fwd(0, None, None, None, 1, 1, 1, 7, None) == -2
  assert lib.expo_filter_fwd(0, None, None, None, 1, 0, 1, 0, None) == -1
  assert lib.expo_filter_fwd(0, None, None, None, 1, 1, 1, 0, None) == -1  # null pointers
  assert lib.expo_filter_fwd(0, None, None, None, 0, 1, 1, 0, None) == 0  # empty batch is a no-op
  assert lib.expo_filter_bwd(0, None, None, None, None, None, 1, 1, 1, 0, 5, None, 0, None) == -1  # bad hsv mode
  assert lib.expo_critic_stats(None, None, 0, 4, 4, 0, None, 0, None) == 0
  # reduction workspace: sized by the library, monotone, 0 for invalid shapes
  assert lib.expo_workspace_bytes(0, 4, 4, 0) == 0 and lib.expo_workspace_bytes(1, 0, 4, 0) == 0
  small, big = lib.expo_workspace_bytes(64, 64, 64, 0), lib.expo_workspace_bytes(64, 512, 512, 0)
  assert 0 < small <= big and big >= 64 * 32 * 32 * 4  # >= one 32-float record per block, 32 blocks per image
  assert lib.expo_workspace_bytes(64, 512, 512, 1) >= big  # fp32 has more 48-byte groups per image
  assert lib.expo_filter_fwd(0, None, None, None, 1, 40000, 40000, 0, None) == -1  # image >= 2 GiB
  assert b'2 GiB' in lib.expo_last_error()
  assert lib.expo_chain_fwd(None, 1, None, None, 1, 1, 1, 0, None) == -1
  # a bad entry in the LAST step is found before anything is enqueued (so this is safe without a GPU): the chain
  # entry points validate every step first
  vp = ctypes.c_void_p
  ids = (ctypes.c_int * 3)(0, 1, 2)
  fake = 0x1000  # never dereferenced on the host
  acts = (vp * 4)(fake, fake, fake, fake)
  grads = (vp * 4)(fake, fake, fake, fake)
  prm = (vp * 3)(fake, fake, None)
  dprm = (vp * 3)(fake, fake, fake)
  assert lib.expo_chain_fwd(ids, 3, acts, prm, 1, 4, 4, 0, None) == -1 and b'null' in lib.expo_last_error()
  assert lib.expo_chain_bwd(ids, 3, acts, grads, prm, dprm, 1, 4, 4, 0, 0, vp(fake), 1 << 20, None) == -1
  assert b'null' in lib.expo_last_error()
  bad_ids = (ctypes.c_int * 3)(0, 1, 9)
  prm_ok = (vp * 3)(fake, fake, fake)
  assert lib.expo_chain_fwd(bad_ids, 3, acts, prm_ok, 1, 4, 4, 0, None) == -1 and b'filter_id' in lib.expo_last_error()
  # the new entry points validate the same way
  assert lib.expo_critic_stats_bwd(None, None, None, None, 1, 4, 4, 0, None) == -1
  assert lib.expo_critic_stats_jvp(None, None, None, None, 0, 4, 4, 0, None, 0, None) == 0
  assert lib.expo_vignet_apply_fwd(None, None, None, 1.0, 1, 1, 4, 4, 5, None) == -2
  assert lib.expo_bias_lrelu_fwd(None, None, None, 0, 1, 0.2, None) == 0
  assert lib.expo_bias_lrelu_fwd(None, None, None, 8, 1, 0.2, None) == -1


def test_product_path_refuses_cpu_tensors():
  from exposure_amd import filters
  from exposure_amd.config import make_cfg
  cfg = make_cfg()
  f = filters.ExposureFilter((1, 4, 4, 3), cfg)
  img = torch.rand(2, 4, 4, 3)
  with pytest.raises(_cabi.ExposureHipError, match='no CPU fallback'):
    f.process(img, torch.zeros(2, 1))
  with pytest.raises(_cabi.ExposureHipError):
    f.apply(img, specified_parameter=torch.zeros(2, 1))


def test_no_oracle_import_in_product():
  pkg = os.path.join(ROOT, 'exposure_amd')
  for dirpath, _, files in os.walk(pkg):
    for fn in files:
      if fn.endswith(('.py', '.hip', '.h')):
        src = open(os.path.join(dirpath, fn)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_integration_document_covers_every_entry_point():
  """INTEGRATION.md's table must name every function the header declares (as `expo_x`, or folded as
  `expo_x_fwd/bwd` / `expo_x_fwd/_bwd`), so a maintainer binding the library finds what each one replaces."""
  import re
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  header = open(os.path.join(root, 'include', 'exposure_hip.h')).read()
  doc = open(os.path.join(root, 'INTEGRATION.md')).read()
  names = sorted(set(re.findall(r'\b(expo_[a-z0-9_]+)\s*\(', header)))
  assert len(names) >= 30
  for n in names:
    folded = n.endswith('_bwd') and (n[:-4] + '_fwd/bwd' in doc or n[:-4] + '_fwd/_bwd' in doc or n[:-4] + '_fwd / _bwd' in doc)
    assert n in doc or folded, n
