"""Static checks on the gfx950 ISA hipcc emits for the library (CPU-only: cross-compiles with -S, runs nothing).

Why this exists: while rewriting the fused inference kernel (round 3) hipcc 7.2 allocated the ADDRESS register of a
``buffer_store_dwordx3`` inside the 96-bit DATA tuple of the same instruction in the fp32 instantiation -- R and G of
every 64th pixel came out as address bits.  The fp32 parity tests caught it on the GPU; this test catches that class
of miscompile here, for every kernel of every translation unit, before anything travels to a GPU."""
import concurrent.futures
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'exposure_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# (source, extra flags) exactly as exposure_amd/csrc/build.sh compiles them
UNITS = [('exposure_hip.hip', []), ('chain_fused.hip', ['-fno-slp-vectorize', '-fno-honor-nans']), ('nn_ops.hip', []),
         ('chain_fused_bwd.hip', ['-fno-slp-vectorize']), ('curve_generic.hip', [])]


def _listing(unit, tmp):
  src, flags = unit
  out = os.path.join(tmp, src + '.s')
  subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only'] + flags +
                        [os.path.join(CSRC, src), '-o', out], stderr=subprocess.DEVNULL)
  return open(out).read()


@pytest.fixture(scope='module')
def listings(tmp_path_factory):
  if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
    pytest.skip('hipcc not available')
  tmp = str(tmp_path_factory.mktemp('isa'))
  with concurrent.futures.ThreadPoolExecutor(len(UNITS)) as ex:
    return dict(zip([u[0] for u in UNITS], ex.map(lambda u: _listing(u, tmp), UNITS)))


def test_no_store_takes_its_address_from_its_own_data_registers(listings):
  pat = re.compile(r'(?:buffer|global|flat)_store_\w+\s+(?:v(\d+)|v\[(\d+):(\d+)\]), v(\d+)\b[^\[]')
  checked = 0
  for name, txt in listings.items():
    kernel = None
    for line in txt.splitlines():
      m = re.match(r'^(_Z\w+|finish_kernel):', line)
      if m:
        kernel = m.group(1)
      m = re.search(r'buffer_store_dword(?:x(\d))?\s+(?:v(\d+)|v\[(\d+):(\d+)\]), v(\d+), s\[', line)
      if m:
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        addr = int(m.group(5))
        checked += 1
        assert not (lo <= addr <= hi), '%s: %s stores its own address register: %s' % (name, kernel, line.strip())
  assert checked > 100  # the pattern still matches what the compiler prints


def test_streaming_kernels_do_not_spill(listings):
  """Scratch in a streaming kernel costs more than the occupancy it buys (r03p14: a 5-wave register budget made the
  Color backward 7 % slower); nothing in the library may fall back to it silently."""
  for name, txt in listings.items():
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, flags=re.S):
      scratch = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2))
      assert scratch is not None and int(scratch.group(1)) == 0, (name, m.group(1), scratch and scratch.group(1))
