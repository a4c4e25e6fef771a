import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # the C-ABI library is git-ignored: (cross-)compile it if this checkout has not been built yet
  lib = os.path.join(ROOT, 'exposure_amd', 'libexposure_hip.so')
  if not os.path.exists(lib):
    import __graft_entry__
    __graft_entry__.build()


def has_gpu():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:
    return False


@pytest.fixture(scope='session')
def gpu_device():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('test marked gpu but no GPU is visible')
  return torch.device('cuda:0')


def pytest_collection_finish(session):
  """Everything alive once the test modules are imported (torch, numpy, the package, the tests themselves) is moved to
  the collector's permanent generation: the full collections of the fixture below then only look at what the tests
  created (a full collection over a loaded torch costs ~90 ms -- 40 s over the gpu suite; frozen: a few ms)."""
  import gc
  gc.collect()
  gc.freeze()


@pytest.fixture(autouse=True)
def _collect_device_garbage_between_tests(request):
  """After every gpu test: free dead reference cycles that own device resources NOW (a ``GAN`` with captured step graphs
  is such a cycle) instead of whenever the cyclic collector next runs -- possibly inside a later test's hipGraph capture,
  where destroying a graph aborts the process (exposure_amd/util.py::capture_without_gc; one gpu-suite run in fifteen
  died that way in round 5 before this fixture and the guards around the captures existed)."""
  yield
  if request.node.get_closest_marker('gpu') is not None:
    import gc
    gc.collect()
