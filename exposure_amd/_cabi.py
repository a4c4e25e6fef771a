"""ctypes binding of ``libexposure_hip.so`` (C-ABI: ``include/exposure_hip.h``).

This is the ONLY way the package reaches the filter maths: there is no CPU or
eager-PyTorch fallback.  If the library is missing, fails to load, or a tensor is
not on a ROCm device, the call raises -- loudly -- instead of computing elsewhere.

PyTorch is used for plumbing only: device memory (``tensor.data_ptr()``) and the
current HIP stream (``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXPO_HIP_LIB selects an alternative build of the SAME library (kernel-variant A/B runs); it is
# never a non-HIP implementation.
LIB_PATH = os.environ.get('EXPO_HIP_LIB') or os.path.join(_HERE, 'libexposure_hip.so')

EXPO_ABI_VERSION = 7
EXPO_CURVE_MAX_STEPS = 16
EXPO_F16, EXPO_F32 = 0, 1
EXPO_MAX_PARAMS = 24
NUM_PARAMS = (1, 1, 3, 1, 8, 1, 1, 24, 2)  # ids 0..7 = cfg.filters order, 8 = LevelFilter

# every symbol include/exposure_hip.h declares: name -> (restype, argtypes)
_vp, _i, _fp, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    'expo_version': (_i, []),
    'expo_last_error': (ctypes.c_char_p, []),
    'expo_build_info': (ctypes.c_char_p, []),
    'expo_num_filter_params': (_i, [_i]),
    'expo_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'expo_filter_fwd': (_i, [_i, _vp, _vp, _fp, _i, _i, _i, _i, _vp]),
    'expo_filter_bwd': (_i, [_i, _vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_filter_bwd_accumulate': (_i, [_i, _vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_filter_bwd_records': (_i, [_i, _vp, _vp, _vp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_finish_bwd': (_i, [ctypes.POINTER(_i), _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i, _vp, _sz,
                            _vp]),
    'expo_filter_apply_fwd': (_i, [_i, _vp, _vp, _fp, _fp, _f, _f, _i, _i, _i, _i, _vp]),
    'expo_filter_apply_bwd': (_i, [_i, _vp, _vp, _vp, _fp, _fp, _fp, _fp, _f, _f, _i, _i, _i, _i, _i, _vp, _sz,
                                   _vp]),
    'expo_filter_apply_dispatch_fwd': (_i, [_vp, _vp, _vp, _fp, _fp, _f, _f, _i, _i, _i, _i, _vp]),
    'expo_filter_apply_dispatch_bwd': (_i, [_vp, _vp, _vp, _vp, _fp, _fp, _fp, _fp, _f, _f, _i, _i, _i, _i, _i, _vp, _sz,
                                            _vp]),
    'expo_filter_dispatch_fwd': (_i, [_vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_filter_dispatch_bwd': (_i, [_vp, _vp, _vp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_chain_streams': (_i, [_i, _i, _i, _i]),
    'expo_chain_helper_stats': (_i, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'expo_chain_prepare': (_i, [_vp]),
    'expo_conv4x4s2_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_bwd_data': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    'expo_conv_tuning': (_i, [_i, _i, _i]),
    'expo_conv4x4s2_wrw_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'expo_conv4x4s2_wrw': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_conv_wrw_tuning': (_i, [_i, _i]),
    'expo_conv4x4s2_bwd_data_mask': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_fwd_planes': (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_fwd_planes_pair': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_fwd_pair': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_bwd_data_mask_pair': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_fwd_mask': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_conv4x4s2_wrw_bias': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_conv4x4s2_wrw_group': (_i, [_i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                      ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i),
                                      ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_vp), ctypes.POINTER(_sz), _vp]),
    'expo_critic_head_fwd': (_i, [_fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _f, _f, _fp, _fp, _fp, _vp]),
    'expo_critic_report': (_i, [_fp, _fp, _fp, _i, _i, _i, _f, _f, _fp, _fp, _fp, _vp]),
    'expo_critic_head_bwd': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _f, _fp, _fp, _fp, _vp]),
    'expo_fc_fwd_slabs_count': (_i, [_i, _i]),
    'expo_fc_fwd_slabs': (_i, [_fp, _fp, _fp, _i, _i, _i, _vp]),
    'expo_fc_bwd_data_mask': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _f, _vp]),
    'expo_fc_wrw': (_i, [_fp, _fp, _fp, _i, _i, _i, _vp]),
    'expo_plane_sums': (_i, [_fp, _fp, _i, _sz, _i, _i, _vp]),
    'expo_gp_direct': (_i, [_fp, _i, _fp, _f, _fp, _fp, _fp, _i, _sz, _vp]),
    'expo_critic_penalty_tangent': (_i, [_fp, _fp, _fp, _f, _fp, _fp, _fp, _i, _i, _i, _vp]),
    'expo_chain_release': (_i, [_vp]),
    'expo_chain_fwd': (_i, [ctypes.POINTER(_i), _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i,
                            _vp]),
    'expo_chain_bwd': (_i, [ctypes.POINTER(_i), _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                            ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_chain_fused_fwd': (_i, [_vp, _fp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'expo_chain_fused_bwd': (_i, [_vp, _fp, _i, _vp, _vp, _vp, _fp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_critic_stats': (_i, [_vp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_overexposure_penalty': (_i, [_vp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_overexposure_penalty_bwd': (_i, [_vp, _fp, _vp, _i, _i, _i, _i, _vp]),
    'expo_critic_stats_bwd': (_i, [_vp, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    'expo_critic_stats_jvp': (_i, [_vp, _fp, _vp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_critic_stats_hvp': (_i, [_vp, _fp, _fp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'expo_vignet_apply_fwd': (_i, [_vp, _vp, _fp, _f, _i, _i, _i, _i, _i, _vp]),
    'expo_vignet_apply_bwd': (_i, [_vp, _vp, _vp, _fp, _fp, _f, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'expo_bias_lrelu_fwd': (_i, [_fp, _fp, _fp, _sz, _i, _f, _vp]),
    'expo_lrelu_bwd': (_i, [_fp, _fp, _fp, _sz, _f, _vp]),
    'expo_lrelu_bwd_bias_workspace_bytes': (_sz, [_i]),
    'expo_lrelu_bwd_bias': (_i, [_fp, _fp, _fp, _fp, _sz, _i, _f, _vp, _sz, _vp]),
    'expo_heads_regress_fwd': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i), _i, ctypes.POINTER(_f), _vp, _fp, _i, _vp]),
    'expo_heads_regress_bwd': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i), _i,
                                   ctypes.POINTER(_f), _vp, _fp, _i, _vp]),
    'expo_agent_select_fwd': (_i, [_fp, _fp, _i, _fp, _fp, ctypes.POINTER(_f), _i, _i, _i, _fp, _fp, _vp, _fp, _fp, _fp, _fp, _i, _vp]),
    'expo_agent_select_bwd': (_i, [_fp, _vp, _fp, ctypes.POINTER(_f), _i, _i, _fp, _fp, _fp, _i, _vp]),
    'expo_planes_concat': (_i, [_vp, _fp, _fp, _i, _sz, _i, _i, _f, _vp]),
    'expo_generator_losses': (_i, [_fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, ctypes.POINTER(_f), _i, _fp, _fp, _fp, _fp, _i, _fp, _fp, _vp]),
    'expo_adam_step': (_i, [_i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                           ctypes.POINTER(_sz), _fp, _fp, _vp, _f, _f, _f, _i, _vp]),
    'expo_gp_inputs': (_i, [_vp, _vp, _fp, _fp, _fp, _i, _sz, _i, _vp]),
    'expo_net_inputs': (_i, [_vp, _vp, _vp, _vp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'expo_gp_inputs_rows': (_i, [_vp, _vp, _vp, _vp, _fp, _fp, _fp, _i, _sz, _i, _vp]),
    'expo_grad_penalty_fwd': (_i, [_fp, _fp, _fp, _i, _sz, _vp]),
    'expo_grad_penalty_bwd': (_i, [_fp, _fp, _fp, _fp, _i, _sz, _vp]),
    'expo_curve_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'expo_curve_fwd': (_i, [_vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    'expo_curve_bwd': (_i, [_vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
}

_lib = None


class ExposureHipError(RuntimeError):
  pass


def load():
  """Load (once) and return the ctypes library; raises if it is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ExposureHipError(
        'exposure_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(or exposure_amd/csrc/build.sh). There is no CPU fallback.' % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the symbol is missing
    fn.restype = res
    fn.argtypes = args
  ver = lib.expo_version()
  if ver != EXPO_ABI_VERSION:
    raise ExposureHipError('exposure_amd: ABI version mismatch (library %d, binding %d)' % (ver, EXPO_ABI_VERSION))
  # the binary must come from THESE sources (it is git-ignored and travels prebuilt): csrc/build.sh bakes the digest in
  built, tree = (lib.expo_build_info() or b'').decode(), source_digest()
  if tree is not None and built != tree and os.environ.get('EXPO_ALLOW_STALE_LIB') != '1' and not os.environ.get('EXPO_HIP_LIB'):
    raise ExposureHipError(
        'exposure_amd: %s was built from other sources (binary %s, tree %s) -- rebuild with exposure_amd/csrc/build.sh '
        '(EXPO_ALLOW_STALE_LIB=1 loads it anyway)' % (LIB_PATH, built[:16], tree[:16]))
  _lib = lib
  return lib


def source_digest():
  """sha256 over csrc/*.hip, csrc/*.h, csrc/build.sh (byte order of the names) behind include/exposure_hip.h -- what
  csrc/build.sh passes to the compiler as EXPO_SOURCE_DIGEST; None when the sources are not there (an installed copy)."""
  import hashlib
  csrc = os.path.join(_HERE, 'csrc')
  header = os.path.join(_HERE, '..', 'include', 'exposure_hip.h')
  if not (os.path.isdir(csrc) and os.path.exists(header)):
    return None
  names = sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.h')) or f == 'build.sh')
  h = hashlib.sha256()
  for path in [header] + [os.path.join(csrc, f) for f in names]:
    with open(path, 'rb') as fh:
      h.update(fh.read())
  return h.hexdigest()


def build_info():
  """The source digest baked into the loaded binary (expo_build_info)."""
  return (load().expo_build_info() or b'').decode()


def _check(rc, what):
  if rc != 0:
    msg = load().expo_last_error()
    raise ExposureHipError('%s failed (code %d): %s' % (what, rc, (msg or b'').decode()))


def _dtype_code(t):
  if t.dtype == torch.float16:
    return EXPO_F16
  if t.dtype == torch.float32:
    return EXPO_F32
  raise ExposureHipError('exposure_amd: image dtype must be float16 or float32, got %s' % t.dtype)


def _img(t, name):
  if not isinstance(t, torch.Tensor) or not t.is_cuda:
    raise ExposureHipError('exposure_amd: %s must be a tensor on a ROCm device (HIP path only, no CPU fallback)' %
                           name)
  if t.dim() != 4 or t.shape[3] != 3:
    raise ExposureHipError('exposure_amd: %s must be NHWC with C == 3, got %s' % (name, tuple(t.shape)))
  if not t.is_contiguous():
    raise ExposureHipError('exposure_amd: %s must be contiguous NHWC' % name)
  return t


def _f32(t, name, shape):
  if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
    raise ExposureHipError('exposure_amd: %s must be a contiguous float32 device tensor of shape %s, got %s %s' %
                           (name, tuple(shape), t.dtype, tuple(t.shape)))
  return t


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def num_filter_params(fid):
  return load().expo_num_filter_params(fid)


# ---- reduction workspace (include/exposure_hip.h "Reduction workspace") -------------------------
# One scratch buffer per device, grown on demand and never freed or moved while the process lives (a
# captured hipGraph keeps the raw pointer).  It needs no initialisation and carries no state between
# calls.  It serves ONE stream at a time: callers that run filter kernels concurrently on several
# streams of a device pass their own ``workspace=`` tensors (``new_workspace``).  It must exist before
# a hipGraph capture starts: run the step once eagerly first (bench.py and GAN._replay do), or call
# ``reserve_workspace``.
_WORKSPACES = {}
_RETIRED = []
_MIN_WORKSPACE = 4 << 20
_WS_STREAM = {}  # device index -> the torch stream that last used the device's shared workspace


def workspace_bytes(n, h, w, dtype_code, steps=1):
  return int(load().expo_workspace_bytes(int(n), int(h), int(w), int(dtype_code))) * int(steps)


def new_workspace(device, nbytes):
  """A private workspace tensor (uint8) for callers that manage their own."""
  return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def reserve_workspace(device, nbytes):
  """Make the device's shared workspace at least ``nbytes`` large and return it."""
  key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
  ws = _WORKSPACES.get(key)
  if ws is None or ws.numel() < nbytes:
    if torch.cuda.is_current_stream_capturing():
      raise ExposureHipError('exposure_amd: the reduction workspace must exist before a hipGraph capture starts '
                             '(run the step once eagerly, or call _cabi.reserve_workspace)')
    if ws is not None:
      _RETIRED.append(ws)  # earlier captures may still point into it
    ws = new_workspace(torch.device('cuda', key), max(int(nbytes), _MIN_WORKSPACE))
    _WORKSPACES[key] = ws
  return ws


def _order_shared_workspace(device):
  """The shared workspace serves one stream at a time.  When a call arrives on a DIFFERENT stream than the previous
  user's, make it wait for everything that stream has enqueued (the earlier call's records are consumed by its own
  finish launch, so stream order is all that is needed) instead of silently racing on the block records.  No cost
  while the stream does not change; skipped under hipGraph capture (a captured region runs on one stream, and a
  cross-stream wait on a non-capturing stream is not capturable) -- concurrent streams pass their own
  ``workspace=`` tensors."""
  key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
  cur = torch.cuda.current_stream(key)
  prev = _WS_STREAM.get(key)
  if prev is not None and prev != cur and not torch.cuda.is_current_stream_capturing():
    cur.wait_stream(prev)
  _WS_STREAM[key] = cur


def _ws(x, workspace, steps=1):
  """(pointer, size) of the workspace for an image tensor ``x``."""
  n, h, w, _ = x.shape
  need = workspace_bytes(n, h, w, _dtype_code(x), steps)
  if workspace is None:
    ws = reserve_workspace(x.device, need)
    _order_shared_workspace(x.device)
  else:
    ws = workspace
  if not ws.is_cuda or ws.device != x.device or ws.numel() * ws.element_size() < need:
    raise ExposureHipError('exposure_amd: workspace must be a device tensor of at least %d bytes on %s' %
                           (need, x.device))
  return ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel() * ws.element_size())


def filter_fwd(fid, x, y, params):
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  assert y.shape == x.shape and y.dtype == x.dtype
  _f32(params, 'params', (n, NUM_PARAMS[fid]))
  with torch.cuda.device(x.device):
    _check(lib.expo_filter_fwd(fid, _ptr(x), _ptr(y), _ptr(params), n, h, w, _dtype_code(x), _stream()),
           'expo_filter_fwd')


def filter_bwd(fid, x, dy, dx, params, dparams, hsv_grad_mode=0, accumulate=False, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  assert dy.shape == x.shape and dy.dtype == x.dtype
  if dx is not None:
    _img(dx, 'dx')
    assert dx.shape == x.shape and dx.dtype == x.dtype
  _f32(params, 'params', (n, NUM_PARAMS[fid]))
  _f32(dparams, 'dparams', (n, NUM_PARAMS[fid]))
  fn = lib.expo_filter_bwd_accumulate if accumulate else lib.expo_filter_bwd
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(fn(fid, _ptr(x), _ptr(dy), _ptr(dx), _ptr(params), _ptr(dparams), n, h, w, _dtype_code(x),
              hsv_grad_mode, wsp, wsb, _stream()), 'expo_filter_bwd')


def filter_bwd_records(fid, x, dy, dx, params, hsv_grad_mode=0, workspace=None):
  """The streaming pass of filter_bwd alone (dx written, block records left in the workspace, no
  parameter gradients yet); finish with :func:`finish_bwd`.  Per-kernel timing and batched finishes."""
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  if dx is not None:
    _img(dx, 'dx')
  _f32(params, 'params', (n, NUM_PARAMS[fid]))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(lib.expo_filter_bwd_records(fid, _ptr(x), _ptr(dy), _ptr(dx), _ptr(params), n, h, w, _dtype_code(x),
                                       hsv_grad_mode, wsp, wsb, _stream()), 'expo_filter_bwd_records')


def finish_bwd(filter_ids, like, params, dparams, workspace=None):
  """One finish launch for len(filter_ids) record slices of the workspace (``like``: an image tensor of
  the shape / dtype the records were produced for)."""
  lib = load()
  steps = len(filter_ids)
  n, h, w, _ = like.shape
  for fid, p, dp in zip(filter_ids, params, dparams):
    _f32(p, 'params', (n, NUM_PARAMS[fid]))
    _f32(dp, 'dparams', (n, NUM_PARAMS[fid]))
  ids = (ctypes.c_int * steps)(*filter_ids)
  with torch.cuda.device(like.device):
    wsp, wsb = _ws(like, workspace, steps)
    _check(lib.expo_finish_bwd(ids, steps, _ptr_array(params), _ptr_array(dparams), n, h, w, _dtype_code(like), wsp,
                               wsb, _stream()), 'expo_finish_bwd')


def apply_fwd(fid, x, y, params, mask_params, maximum_sharpness, minimum_strength):
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  _f32(params, 'params', (n, NUM_PARAMS[fid]))
  _f32(mask_params, 'mask_params', (n, 6))
  with torch.cuda.device(x.device):
    _check(
        lib.expo_filter_apply_fwd(fid, _ptr(x), _ptr(y), _ptr(params), _ptr(mask_params), float(maximum_sharpness),
                                  float(minimum_strength), n, h, w, _dtype_code(x), _stream()),
        'expo_filter_apply_fwd')


def apply_bwd(fid, x, dy, dx, params, dparams, mask_params, dmask_params, maximum_sharpness, minimum_strength,
              hsv_grad_mode=0, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  if dx is not None:
    _img(dx, 'dx')
  _f32(params, 'params', (n, NUM_PARAMS[fid]))
  _f32(dparams, 'dparams', (n, NUM_PARAMS[fid]))
  _f32(mask_params, 'mask_params', (n, 6))
  _f32(dmask_params, 'dmask_params', (n, 6))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(
        lib.expo_filter_apply_bwd(fid, _ptr(x), _ptr(dy), _ptr(dx), _ptr(params), _ptr(dparams), _ptr(mask_params),
                                  _ptr(dmask_params), float(maximum_sharpness), float(minimum_strength), n, h, w,
                                  _dtype_code(x), hsv_grad_mode, wsp, wsb, _stream()), 'expo_filter_apply_bwd')


def _ids(ids, n):
  if not ids.is_cuda or ids.dtype != torch.int32 or tuple(ids.shape) != (n,) or not ids.is_contiguous():
    raise ExposureHipError('exposure_amd: filter_ids must be a contiguous int32 device tensor of shape (%d,)' % n)
  return ids


def dispatch_fwd(ids, x, y, params, penalty=None, workspace=None):
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  _ids(ids, n)
  _f32(params, 'params', (n, EXPO_MAX_PARAMS))
  if penalty is not None:
    _f32(penalty, 'penalty', (n,))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(
        lib.expo_filter_dispatch_fwd(_ptr(ids), _ptr(x), _ptr(y), _ptr(params), _ptr(penalty), n, h, w,
                                     _dtype_code(x), wsp, wsb, _stream()), 'expo_filter_dispatch_fwd')


def dispatch_bwd(ids, x, dy, dx, params, dparams, dpenalty=None, hsv_grad_mode=0, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  _ids(ids, n)
  if dx is not None:
    _img(dx, 'dx')
  _f32(params, 'params', (n, EXPO_MAX_PARAMS))
  _f32(dparams, 'dparams', (n, EXPO_MAX_PARAMS))
  if dpenalty is not None:
    _f32(dpenalty, 'dpenalty', (n,))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(
        lib.expo_filter_dispatch_bwd(_ptr(ids), _ptr(x), _ptr(dy), _ptr(dx), _ptr(params), _ptr(dparams),
                                     _ptr(dpenalty), n, h, w, _dtype_code(x), hsv_grad_mode, wsp, wsb, _stream()),
        'expo_filter_dispatch_bwd')


def _ptr_array(tensors):
  arr = (ctypes.c_void_p * len(tensors))()
  for i, t in enumerate(tensors):
    arr[i] = t.data_ptr()
  return arr


def chain_fwd(filter_ids, acts, params):
  """acts: list of steps+1 image tensors (acts[0] input); params: list of (N,P_i) float32."""
  lib = load()
  steps = len(filter_ids)
  assert len(acts) == steps + 1 and len(params) == steps
  n, h, w, _ = acts[0].shape
  for a in acts:
    _img(a, 'act')
    assert a.shape == acts[0].shape and a.dtype == acts[0].dtype
  for fid, p in zip(filter_ids, params):
    _f32(p, 'params', (n, NUM_PARAMS[fid]))
  ids = (ctypes.c_int * steps)(*filter_ids)
  with torch.cuda.device(acts[0].device):
    _check(lib.expo_chain_fwd(ids, steps, _ptr_array(acts), _ptr_array(params), n, h, w, _dtype_code(acts[0]),
                              _stream()), 'expo_chain_fwd')


def chain_bwd(filter_ids, acts, grads, params, dparams, hsv_grad_mode=0, workspace=None):
  lib = load()
  steps = len(filter_ids)
  assert len(acts) == steps + 1 and len(grads) == steps + 1 and len(params) == steps and len(dparams) == steps
  n, h, w, _ = acts[0].shape
  for a in list(acts) + list(grads):
    _img(a, 'act/grad')
    assert a.shape == acts[0].shape and a.dtype == acts[0].dtype
  for fid, p, dp in zip(filter_ids, params, dparams):
    _f32(p, 'params', (n, NUM_PARAMS[fid]))
    _f32(dp, 'dparams', (n, NUM_PARAMS[fid]))
  ids = (ctypes.c_int * steps)(*filter_ids)
  with torch.cuda.device(acts[0].device):
    wsp, wsb = _ws(acts[0], workspace, steps)
    _check(
        lib.expo_chain_bwd(ids, steps, _ptr_array(acts), _ptr_array(grads), _ptr_array(params),
                           _ptr_array(dparams), n, h, w, _dtype_code(acts[0]), hsv_grad_mode, wsp, wsb, _stream()),
        'expo_chain_bwd')


def chain_fused_fwd(filter_ids, params, x, y):
  """filter_ids (N, steps) int32, params (N, steps, 24) float32 -> y = all steps applied in registers."""
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  steps = filter_ids.shape[1]
  if not filter_ids.is_cuda or filter_ids.dtype != torch.int32 or not filter_ids.is_contiguous() or \
      tuple(filter_ids.shape) != (n, steps):
    raise ExposureHipError('exposure_amd: filter_ids must be a contiguous int32 device tensor of shape (N, steps)')
  _f32(params, 'params', (n, steps, EXPO_MAX_PARAMS))
  with torch.cuda.device(x.device):
    _check(lib.expo_chain_fused_fwd(_ptr(filter_ids), _ptr(params), steps, _ptr(x), _ptr(y), n, h, w,
                                    _dtype_code(x), _stream()), 'expo_chain_fused_fwd')


FUSED_BWD_MAX_STEPS = 8  # EXPO_FUSED_BWD_MAX_STEPS


def chain_fused_bwd(filter_ids, params, x, dy, dx, dparams, hsv_grad_mode=0, workspace=None):
  """One-pass backward of the fixed per-image sequence of ``chain_fused_fwd``: filter_ids (N, steps) int32, params
  and dparams (N, steps, 24) float32; x, dy -> dx (may alias dy).  steps <= FUSED_BWD_MAX_STEPS."""
  lib = load()
  _img(x, 'x'), _img(dy, 'dy'), _img(dx, 'dx')
  n, h, w, _ = x.shape
  steps = filter_ids.shape[1]
  if dy.shape != x.shape or dx.shape != x.shape or dy.dtype != x.dtype or dx.dtype != x.dtype:
    raise ExposureHipError('exposure_amd: x, dy, dx must agree in shape and dtype')
  if not filter_ids.is_cuda or filter_ids.dtype != torch.int32 or not filter_ids.is_contiguous() or \
      tuple(filter_ids.shape) != (n, steps):
    raise ExposureHipError('exposure_amd: filter_ids must be a contiguous int32 device tensor of shape (N, steps)')
  _f32(params, 'params', (n, steps, EXPO_MAX_PARAMS))
  _f32(dparams, 'dparams', (n, steps, EXPO_MAX_PARAMS))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace, max(steps, 1))
    _check(lib.expo_chain_fused_bwd(_ptr(filter_ids), _ptr(params), steps, _ptr(x), _ptr(dy), _ptr(dx), _ptr(dparams),
                                    n, h, w, _dtype_code(x), hsv_grad_mode, wsp, wsb, _stream()),
           'expo_chain_fused_bwd')


def critic_stats(x, stats, workspace=None):
  lib = load()
  _img(x, 'x')
  n, h, w, _ = x.shape
  _f32(stats, 'stats', (n, 3))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(lib.expo_critic_stats(_ptr(x), _ptr(stats), n, h, w, _dtype_code(x), wsp, wsb, _stream()),
           'expo_critic_stats')


def overexposure_penalty(y, penalty, workspace=None):
  lib = load()
  _img(y, 'y')
  n, h, w, _ = y.shape
  _f32(penalty, 'penalty', (n,))
  with torch.cuda.device(y.device):
    wsp, wsb = _ws(y, workspace)
    _check(lib.expo_overexposure_penalty(_ptr(y), _ptr(penalty), n, h, w, _dtype_code(y), wsp, wsb, _stream()),
           'expo_overexposure_penalty')


def overexposure_penalty_bwd(y, dpenalty, dy):
  lib = load()
  _img(y, 'y'), _img(dy, 'dy')
  n, h, w, _ = y.shape
  assert dy.shape == y.shape and dy.dtype == y.dtype
  _f32(dpenalty, 'dpenalty', (n,))
  with torch.cuda.device(y.device):
    _check(lib.expo_overexposure_penalty_bwd(_ptr(y), _ptr(dpenalty), _ptr(dy), n, h, w, _dtype_code(y), _stream()),
           'expo_overexposure_penalty_bwd')


def critic_stats_bwd(x, stats, dstats, dx):
  """dx = J^T dstats (first derivative of the critic statistics)."""
  lib = load()
  _img(x, 'x'), _img(dx, 'dx')
  n, h, w, _ = x.shape
  assert dx.shape == x.shape and dx.dtype == x.dtype
  _f32(stats, 'stats', (n, 3)), _f32(dstats, 'dstats', (n, 3))
  with torch.cuda.device(x.device):
    _check(lib.expo_critic_stats_bwd(_ptr(x), _ptr(stats), _ptr(dstats), _ptr(dx), n, h, w, _dtype_code(x), _stream()),
           'expo_critic_stats_bwd')


def critic_stats_jvp(x, stats, v, jv, workspace=None):
  """jv = J v (N, 3): the derivative of <critic_stats_bwd(x, g), v> with respect to g."""
  lib = load()
  _img(x, 'x'), _img(v, 'v')
  n, h, w, _ = x.shape
  assert v.shape == x.shape and v.dtype == x.dtype
  _f32(stats, 'stats', (n, 3)), _f32(jv, 'jv', (n, 3))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(lib.expo_critic_stats_jvp(_ptr(x), _ptr(stats), _ptr(v), _ptr(jv), n, h, w, _dtype_code(x), wsp, wsb,
                                     _stream()), 'expo_critic_stats_jvp')


def critic_stats_hvp(x, dstats, jv, v, out):
  """out = d <critic_stats_bwd(x, dstats), v> / dx (jv = critic_stats_jvp(x, stats, v))."""
  lib = load()
  _img(x, 'x'), _img(v, 'v'), _img(out, 'out')
  n, h, w, _ = x.shape
  assert v.shape == x.shape and v.dtype == x.dtype and out.shape == x.shape and out.dtype == x.dtype
  _f32(dstats, 'dstats', (n, 3)), _f32(jv, 'jv', (n, 3))
  with torch.cuda.device(x.device):
    _check(lib.expo_critic_stats_hvp(_ptr(x), _ptr(dstats), _ptr(jv), _ptr(v), _ptr(out), n, h, w, _dtype_code(x),
                                     _stream()), 'expo_critic_stats_hvp')


def _dense_f32(t, name):
  if not isinstance(t, torch.Tensor) or not t.is_cuda:
    raise ExposureHipError('exposure_amd: %s must be a tensor on a ROCm device (HIP path only, no CPU fallback)' %
                           name)
  if t.dtype != torch.float32 or not t.is_contiguous():
    raise ExposureHipError('exposure_amd: %s must be a contiguous float32 tensor, got %s' % (name, t.dtype))
  return t


def bias_lrelu_fwd(y, bias, z, leak=0.2):
  """z = lrelu(y + bias) with the channel as the LAST dimension of y (bias may be None)."""
  lib = load()
  _dense_f32(y, 'y'), _dense_f32(z, 'z')
  assert z.shape == y.shape
  channels = 1
  if bias is not None:
    _dense_f32(bias, 'bias')
    channels = int(y.shape[-1])
    assert bias.numel() == channels
  with torch.cuda.device(y.device):
    _check(lib.expo_bias_lrelu_fwd(_ptr(y), _ptr(bias), _ptr(z), y.numel(), channels, float(leak), _stream()),
           'expo_bias_lrelu_fwd')


def lrelu_bwd(z, dz, dy, leak=0.2):
  """dy = dz * lrelu'(.) with the slope read from the activation's OUTPUT z."""
  lib = load()
  _dense_f32(z, 'z'), _dense_f32(dz, 'dz'), _dense_f32(dy, 'dy')
  assert dz.shape == z.shape and dy.shape == z.shape
  with torch.cuda.device(z.device):
    _check(lib.expo_lrelu_bwd(_ptr(z), _ptr(dz), _ptr(dy), z.numel(), float(leak), _stream()), 'expo_lrelu_bwd')


def vignet_apply_fwd(x, y, mask_params, maximum_sharpness, masking):
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  assert y.shape == x.shape and y.dtype == x.dtype
  _f32(mask_params, 'mask_params', (n, 5))
  with torch.cuda.device(x.device):
    _check(lib.expo_vignet_apply_fwd(_ptr(x), _ptr(y), _ptr(mask_params), float(maximum_sharpness), int(bool(masking)),
                                     n, h, w, _dtype_code(x), _stream()), 'expo_vignet_apply_fwd')


def vignet_apply_bwd(x, dy, dx, mask_params, dmask_params, maximum_sharpness, masking, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  if dx is not None:
    _img(dx, 'dx')
  _f32(mask_params, 'mask_params', (n, 5))
  _f32(dmask_params, 'dmask_params', (n, 5))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(lib.expo_vignet_apply_bwd(_ptr(x), _ptr(dy), _ptr(dx), _ptr(mask_params), _ptr(dmask_params),
                                     float(maximum_sharpness), int(bool(masking)), n, h, w, _dtype_code(x), wsp, wsb,
                                     _stream()), 'expo_vignet_apply_bwd')


def lrelu_bwd_bias(z, dz, dy, dbias, leak=0.2, workspace=None):
  """dy = dz * lrelu'(.) AND dbias = column sums of dy in one pass (channel = last dimension of z)."""
  lib = load()
  channels = z.shape[-1]
  for t in (z, dz, dy):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == z.shape):
      raise ExposureHipError('exposure_amd: lrelu_bwd_bias wants contiguous float32 device tensors of one shape')
  _f32(dbias, 'dbias', (channels,))
  need = int(lib.expo_lrelu_bwd_bias_workspace_bytes(int(channels)))
  with torch.cuda.device(z.device):
    if workspace is None:
      workspace = reserve_workspace(z.device, need)
      _order_shared_workspace(z.device)
    _check(lib.expo_lrelu_bwd_bias(_ptr(z), _ptr(dz), _ptr(dy), _ptr(dbias), z.numel(), int(channels), float(leak),
                                   ctypes.c_void_p(workspace.data_ptr()),
                                   ctypes.c_size_t(workspace.numel() * workspace.element_size()), _stream()),
           'expo_lrelu_bwd_bias')


def lrelu_bwd_bias_supported(z, dz):
  c = z.shape[-1]
  if os.environ.get('EXPO_FUSED_BIAS_GRAD', '1') == '0':  # A/B switch: the separate column reduction of round 3
    return False
  return z.is_cuda and 4 <= c <= 256 and (c & (c - 1)) == 0 and z.numel() > 0 and z.data_ptr() % 16 == 0 and \
      dz.data_ptr() % 16 == 0 and z.dtype == torch.float32 and dz.dtype == torch.float32 and z.is_contiguous()


def _heads_common(raws, abi_ids, ranges):
  n = raws[0].shape[0]
  for r in raws:
    if not (r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and r.dim() == 2 and r.shape[0] == n):
      raise ExposureHipError('exposure_amd: every head output must be a contiguous float32 (N, width) device tensor')
  h = len(raws)
  ptrs = (_vp * h)(*[ctypes.c_void_p(r.data_ptr()) for r in raws])
  widths = (_i * h)(*[int(r.shape[1]) for r in raws])
  abi = (_i * h)(*[int(v) for v in abi_ids])
  rng = (_f * 9)(*[float(v) for v in ranges])
  return n, h, ptrs, widths, abi, rng


def heads_regress_fwd(raws, abi_ids, ranges, selected, params):
  """params[n] = regressor of the head selected[n] applied to that head's raw features (EXPO_MAX_PARAMS wide, zero padded)."""
  lib = load()
  n, h, ptrs, widths, abi, rng = _heads_common(raws, abi_ids, ranges)
  _ids(selected, n)
  _f32(params, 'params', (n, EXPO_MAX_PARAMS))
  with torch.cuda.device(params.device):
    _check(lib.expo_heads_regress_fwd(ptrs, widths, abi, h, rng, _ptr(selected), _ptr(params), n, _stream()),
           'expo_heads_regress_fwd')


def heads_regress_bwd(raws, draws, abi_ids, ranges, selected, dparams):
  lib = load()
  n, h, ptrs, widths, abi, rng = _heads_common(raws, abi_ids, ranges)
  assert len(draws) == h and all(d.shape == r.shape and d.is_contiguous() and d.dtype == torch.float32 for d, r in zip(draws, raws))
  dptrs = (_vp * h)(*[ctypes.c_void_p(d.data_ptr()) for d in draws])
  _ids(selected, n)
  _f32(dparams, 'dparams', (n, EXPO_MAX_PARAMS))
  with torch.cuda.device(dparams.device):
    _check(lib.expo_heads_regress_bwd(ptrs, dptrs, widths, abi, h, rng, _ptr(selected), _ptr(dparams), n, _stream()),
           'expo_heads_regress_bwd')


def agent_select_fwd(logits, noise, states, progress, consts, is_train, pdf, entropy, selected, onehot, surrogate, new_states,
                     penalty_base):
  """agent.py:87-125, 207-252 per image in one launch; ``noise``: (N, z_dim) float32, column 0 is used."""
  lib = load()
  n, k = logits.shape
  _f32(logits, 'logits', (n, k)), _f32(states, 'states', (n, states.shape[1])), _f32(progress, 'progress', tuple(progress.shape))
  assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape[0] == n and progress.numel() == 1
  _f32(pdf, 'pdf', (n, k)), _f32(onehot, 'onehot', (n, k)), _f32(new_states, 'new_states', tuple(states.shape))
  _ids(selected, n)
  for t, nm in ((entropy, 'entropy'), (surrogate, 'surrogate'), (penalty_base, 'penalty_base')):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, nm
  c = (_f * 5)(*[float(v) for v in consts])
  with torch.cuda.device(logits.device):
    _check(lib.expo_agent_select_fwd(_ptr(logits), _ptr(noise), int(noise.shape[1]) if noise.dim() > 1 else 1, _ptr(states),
                                     _ptr(progress), c, k, int(states.shape[1]), int(bool(is_train)), _ptr(pdf), _ptr(entropy),
                                     _ptr(selected), _ptr(onehot), _ptr(surrogate), _ptr(new_states), _ptr(penalty_base), n,
                                     _stream()), 'expo_agent_select_fwd')


def agent_select_bwd(logits, selected, progress, consts, state_dim, d_surrogate, d_penalty_base, d_logits):
  lib = load()
  n, k = logits.shape
  _f32(logits, 'logits', (n, k)), _f32(d_logits, 'd_logits', (n, k))
  _ids(selected, n)
  for t in (d_surrogate, d_penalty_base):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
  c = (_f * 5)(*[float(v) for v in consts])
  with torch.cuda.device(logits.device):
    _check(lib.expo_agent_select_bwd(_ptr(logits), _ptr(selected), _ptr(progress), c, k, int(state_dim), _ptr(d_surrogate),
                                     _ptr(d_penalty_base), _ptr(d_logits), n, _stream()), 'expo_agent_select_bwd')


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, step, ticket, beta1, beta2, eps, step_advanced=False):
  """One Adam update of the listed fp32 tensors (expo_adam_step).  Tensor j of the four lists must share one element
  order (same sizes and strides, dense); ``lr`` / ``step`` are device floats, ``ticket`` a device int32 (zero).
  ``step_advanced``: a kernel in front of this update has moved ``step`` already (``critic_report`` / ``generator_losses``
  with ``adam_step(s)``): the update computes with t = step and launches no advance behind it."""
  lib = load()
  count = len(params)
  assert len(grads) == count and len(exp_avg) == count and len(exp_avg_sq) == count
  if count == 0:
    return
  dev = params[0].device
  for quad in zip(params, grads, exp_avg, exp_avg_sq):
    for t in quad:
      if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.size() == quad[0].size() and
              t.stride() == quad[0].stride()):
        raise ExposureHipError('exposure_amd: adam_step wants fp32 device tensors of one layout per parameter')
  assert lr.is_cuda and lr.dtype == torch.float32 and step.is_cuda and step.dtype == torch.float32
  assert ticket.is_cuda and ticket.dtype == torch.int32
  arr = lambda ts: (_vp * count)(*[t.data_ptr() for t in ts])
  numel = (_sz * count)(*[t.numel() for t in params])
  with torch.cuda.device(dev):
    _check(lib.expo_adam_step(count, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, _ptr(lr), _ptr(step),
                              _ptr(ticket), float(beta1), float(beta2), float(eps), 1 if step_advanced else 0, _stream()),
           'expo_adam_step')


def _step_scalar(t):
  assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1)
  return t


def planes_concat(images, vec, out, offset):
  """out[n, p, c] = (c < 3 ? images[n, p, c] : vec[n, c - 3]) - offset (float32, 3 + V channels): expo_planes_concat."""
  lib = load()
  _img(images, 'images')
  n = images.shape[0]
  pixels = images[0].numel() // 3 if n else 0
  v = 0 if vec is None else int(vec.shape[1])
  if vec is not None:
    _f32(vec, 'vec', (n, v))
  assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
  assert tuple(out.shape) == tuple(images.shape[:-1]) + (3 + v,)
  with torch.cuda.device(images.device):
    _check(lib.expo_planes_concat(_ptr(images), _ptr(vec), _ptr(out), n, pixels, v, _dtype_code(images), float(offset),
                                  _stream()), 'expo_planes_concat')


def generator_losses(fake_logit, fake_input_logit, new_value, old_value, new_states, penalty, surrogate, consts, use_td,
                     losses, reward, q_value, coef, adam_steps=(None, None)):
  """expo_generator_losses: (g_loss, v_loss), reward, q and the five gradient coefficient rows of the G step's loss glue."""
  lib = load()
  n = fake_logit.numel()
  for t in (fake_logit, fake_input_logit, new_value, old_value, surrogate, reward, q_value) + ((penalty,) if penalty is not None else ()):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
  _f32(new_states, 'new_states', (n, new_states.shape[1]))
  assert losses.is_cuda and losses.dtype == torch.float32 and losses.numel() == 2
  _f32(coef, 'coef', (5, n))
  c = (_f * 5)(*[float(v) for v in consts])
  with torch.cuda.device(fake_logit.device):
    _check(lib.expo_generator_losses(_ptr(fake_logit), _ptr(fake_input_logit), _ptr(new_value), _ptr(old_value),
                                     _ptr(new_states), int(new_states.shape[1]), _ptr(penalty), _ptr(surrogate), c,
                                     int(bool(use_td)), _ptr(losses), _ptr(reward), _ptr(q_value), _ptr(coef), n,
                                     _ptr(_step_scalar(adam_steps[0])), _ptr(_step_scalar(adam_steps[1])), _stream()),
           'expo_generator_losses')


def gp_inputs(real, fake, alpha, cat_out, interp, real_rows=None, fake_rows=None):
  """cat_out[:n] = real, cat_out[n:] = fake (float32), interp = real + alpha (fake - real): one launch.  ``interp`` (and
  ``alpha``) may be None: conversion + concatenation only.  ``real_rows`` / ``fake_rows`` (device int64 (n,)): the batch's
  image i is row rows[i] of ``real`` / ``fake`` (a data set, the replay memory's pool) instead of row i."""
  lib = load()
  _img(real, 'real'), _img(fake, 'fake')
  assert fake.shape[1:] == real.shape[1:] and fake.dtype == real.dtype
  for rows in (real_rows, fake_rows):
    assert rows is None or (rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous() and rows.dim() == 1)
  n = real.shape[0] if real_rows is None else real_rows.shape[0]
  assert n == (fake.shape[0] if fake_rows is None else fake_rows.shape[0])
  m = real[0].numel() if real.shape[0] else 0
  assert cat_out.dtype == torch.float32 and cat_out.is_contiguous() and tuple(cat_out.shape) == (2 * n,) + tuple(real.shape[1:])
  if interp is not None:
    assert interp.dtype == torch.float32 and interp.is_contiguous() and tuple(interp.shape) == (n,) + tuple(real.shape[1:])
    assert alpha.is_cuda and alpha.dtype == torch.float32 and alpha.is_contiguous() and alpha.numel() == n
  with torch.cuda.device(real.device):
    _check(lib.expo_gp_inputs_rows(_ptr(real), _ptr(real_rows), _ptr(fake), _ptr(fake_rows),
                                   _ptr(alpha) if interp is not None else None, _ptr(cat_out), _ptr(interp),
                                   n, m, _dtype_code(real), _stream()), 'expo_gp_inputs_rows')


NET_INPUTS_MAX_PIXELS = 4096  # expo_net_inputs holds an image in LDS


def net_inputs(a, b, alpha, planes, stats, x_out=None, x_first=0, vec_a=None, vec_b=None, a_rows=None, b_rows=None,
               offset=0.5):
  """expo_net_inputs: planes = concat([a | b (| a + alpha (b - a))] as float32, per-image values, statistics) - offset,
  stats = the rows' statistics, x_out = the float32 images of rows x_first .. x_first + len(x_out): one launch
  (critics.py:42-76; net.py:170-172).  ``a_rows`` / ``b_rows`` (device int64 (n,)): image j is row rows[j] of a / b."""
  lib = load()
  _img(a, 'a'), _img(b, 'b')
  assert a.shape[1:] == b.shape[1:] and a.dtype == b.dtype and a.dim() == 4
  for rows in (a_rows, b_rows):
    assert rows is None or (rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous() and rows.dim() == 1)
  n = a.shape[0] if a_rows is None else a_rows.shape[0]
  assert n == (b.shape[0] if b_rows is None else b_rows.shape[0])
  h, w = int(a.shape[1]), int(a.shape[2])
  m = (3 if alpha is not None else 2) * n
  v0 = 0 if vec_a is None else int(vec_a.shape[1])
  if alpha is not None:
    assert alpha.is_cuda and alpha.dtype == torch.float32 and alpha.is_contiguous() and alpha.numel() == n
  if v0:
    _f32(vec_a, 'vec_a', (n, v0)), _f32(vec_b, 'vec_b', (n, v0))
  _f32(planes, 'planes', (m, h, w, 6 + v0)), _f32(stats, 'stats', (m, 3))
  x_count = 0
  if x_out is not None:
    x_count = int(x_out.shape[0])
    _f32(x_out, 'x_out', (x_count, h, w, 3))
  with torch.cuda.device(a.device):
    _check(lib.expo_net_inputs(_ptr(a), _ptr(a_rows), _ptr(b), _ptr(b_rows), _ptr(alpha), _ptr(vec_a) if v0 else None,
                               _ptr(vec_b) if v0 else None, v0, _ptr(planes), _ptr(stats), _ptr(x_out), int(x_first), x_count,
                               n, h, w, _dtype_code(a), float(offset), _stream()), 'expo_net_inputs')


def grad_penalty_fwd(g, norm, term):
  lib = load()
  n = g.shape[0]
  assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()
  _f32(norm, 'norm', (n,)), _f32(term, 'term', (n,))
  with torch.cuda.device(g.device):
    _check(lib.expo_grad_penalty_fwd(_ptr(g), _ptr(norm), _ptr(term), n, g[0].numel() if n else 0, _stream()),
           'expo_grad_penalty_fwd')


def grad_penalty_bwd(g, norm, dterm, dg):
  lib = load()
  n = g.shape[0]
  assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and dg.shape == g.shape and dg.is_contiguous()
  _f32(norm, 'norm', (n,)), _f32(dterm, 'dterm', (n,))
  with torch.cuda.device(g.device):
    _check(lib.expo_grad_penalty_bwd(_ptr(g), _ptr(norm), _ptr(dterm), _ptr(dg), n, g[0].numel() if n else 0, _stream()),
           'expo_grad_penalty_bwd')


def curve_fwd(x, y, params, curves, steps):
  """Tone (curves = 1) / Color (curves = 3) with any cfg.curve_steps: params (N, curves * steps)."""
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  assert y.shape == x.shape and y.dtype == x.dtype
  _f32(params, 'params', (n, curves * steps))
  with torch.cuda.device(x.device):
    _check(lib.expo_curve_fwd(_ptr(x), _ptr(y), _ptr(params), n, h, w, _dtype_code(x), int(curves), int(steps), _stream()),
           'expo_curve_fwd')


def curve_bwd(x, dy, dx, params, dparams, curves, steps, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  if dx is not None:
    _img(dx, 'dx')
  _f32(params, 'params', (n, curves * steps))
  _f32(dparams, 'dparams', (n, curves * steps))
  need = int(lib.expo_curve_workspace_bytes(n, h, w, int(curves), int(steps)))
  with torch.cuda.device(x.device):
    if workspace is None:
      workspace = reserve_workspace(x.device, need)
      _order_shared_workspace(x.device)
    if not workspace.is_cuda or workspace.numel() * workspace.element_size() < need:
      raise ExposureHipError('exposure_amd: workspace must be a device tensor of at least %d bytes' % need)
    _check(lib.expo_curve_bwd(_ptr(x), _ptr(dy), _ptr(dx), _ptr(params), _ptr(dparams), n, h, w, _dtype_code(x),
                              int(curves), int(steps), ctypes.c_void_p(workspace.data_ptr()),
                              ctypes.c_size_t(workspace.numel() * workspace.element_size()), _stream()), 'expo_curve_bwd')


def chain_streams(n, h, w, dtype_code):
  """1 or 2: how many streams expo_chain_fwd / _bwd use for a batch of this shape."""
  return int(load().expo_chain_streams(int(n), int(h), int(w), int(dtype_code)))


def chain_helper_stats():
  """(probed, rejected): helper-stream pairings this process probed for a hardware queue of their own, and helpers it
  rejected (expo_chain_helper_stats; diagnostics)."""
  a, b = ctypes.c_int(0), ctypes.c_int(0)
  _check(load().expo_chain_helper_stats(ctypes.byref(a), ctypes.byref(b)), 'expo_chain_helper_stats')
  return a.value, b.value


def conv4x4s2_fwd(x, w, bias, y, act, leak=0.2):
  """y = f(conv(x, w) + bias): NHWC float32 ``x`` (N, H, W, Cin), ``w`` an nn.Conv2d weight (Cout, Cin, 4, 4) in
  channels_last memory order, ``y`` (N, H/2, W/2, Cout); ``act`` 1 -> lrelu(., leak) (expo_conv4x4s2_fwd)."""
  lib = load()
  n, h, wd, cin = x.shape
  cout = w.shape[0]
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  assert w.dtype == torch.float32 and tuple(w.shape) == (cout, cin, 4, 4) and w.permute(0, 2, 3, 1).is_contiguous()
  assert y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (n, h // 2, wd // 2, cout)
  if bias is not None:
    _f32(bias, 'bias', (cout,))
  with torch.cuda.device(x.device):
    _check(lib.expo_conv4x4s2_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, h, wd, cin, cout, int(act), float(leak),
                                  _stream()), 'expo_conv4x4s2_fwd')


def conv4x4s2_bwd_data(dy, w, dx):
  """dx = the data gradient of conv4x4s2 for upstream ``dy`` (N, H/2, W/2, Cout): NHWC float32 ``dx`` (N, H, W, Cin),
  every element written (expo_conv4x4s2_bwd_data)."""
  lib = load()
  n, h, wd, cin = dx.shape
  cout = w.shape[0]
  assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
  assert w.dtype == torch.float32 and tuple(w.shape) == (cout, cin, 4, 4) and w.permute(0, 2, 3, 1).is_contiguous()
  assert dx.dtype == torch.float32 and dx.is_contiguous()
  with torch.cuda.device(dy.device):
    _check(lib.expo_conv4x4s2_bwd_data(_ptr(dy), _ptr(w), _ptr(dx), n, h, wd, cin, cout, _stream()),
           'expo_conv4x4s2_bwd_data')


def conv_tuning(tile=0, nt=0, slices=0):
  """Probes / tests: force the decomposition of the convolution kernels (expo_conv_tuning; 0 = the library's choice)."""
  _check(load().expo_conv_tuning(int(tile), int(nt), int(slices)), 'expo_conv_tuning')


def conv4x4s2_wrw(x, dy, dw):
  """dw = the weight gradient of conv4x4s2: NHWC float32 ``x`` (N, H, W, Cin), ``dy`` (N, H/2, W/2, Cout), ``dw`` a
  (Cout, Cin, 4, 4) tensor in channels_last memory order, every element written (expo_conv4x4s2_wrw)."""
  lib = load()
  n, h, wd, cin = x.shape
  cout = dy.shape[-1]
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
  assert dw.dtype == torch.float32 and tuple(dw.shape) == (cout, cin, 4, 4) and dw.permute(0, 2, 3, 1).is_contiguous()
  need = int(lib.expo_conv4x4s2_wrw_workspace_bytes(n, h, wd, cin, cout))
  # the P block copies of dW: scratch from torch's caching allocator per call (stream-ordered reuse; inside a hipGraph
  # capture it comes from the graph's private pool) -- a buffer cached per device would be shared between streams
  ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
  with torch.cuda.device(x.device):
    _check(lib.expo_conv4x4s2_wrw(_ptr(x), _ptr(dy), _ptr(dw), n, h, wd, cin, cout,
                                  ctypes.c_void_p(ws.data_ptr() if ws is not None else 0),
                                  ctypes.c_size_t(ws.numel() if ws is not None else 0), _stream()), 'expo_conv4x4s2_wrw')


def _conv_args(x_shape, w):
  n, h, wd, cin = x_shape
  cout = w.shape[0]
  assert w.dtype == torch.float32 and tuple(w.shape) == (cout, cin, 4, 4) and w.permute(0, 2, 3, 1).is_contiguous()
  return n, h, wd, cin, cout


def conv4x4s2_bwd_data_mask(dy, w, zmask, dx, leak=0.2):
  """dx = D(dy, w) * slope(zmask): the data gradient with the lrelu backward of the layer below in its epilogue
  (expo_conv4x4s2_bwd_data_mask); ``zmask`` = that layer's activation, shaped like ``dx``."""
  lib = load()
  n, h, wd, cin, cout = _conv_args(dx.shape, w)
  assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
  assert dx.dtype == torch.float32 and dx.is_contiguous()
  assert zmask.dtype == torch.float32 and zmask.is_contiguous() and zmask.shape == dx.shape
  with torch.cuda.device(dy.device):
    _check(lib.expo_conv4x4s2_bwd_data_mask(_ptr(dy), _ptr(w), _ptr(zmask), _ptr(dx), n, h, wd, cin, cout, float(leak),
                                            _stream()), 'expo_conv4x4s2_bwd_data_mask')


def conv_planes_ok(x_shape, cout):
  """``conv4x4s2_fwd_planes`` takes this first layer (64-wide NHWC input of 4 .. 20 planes, at most 32 output channels)."""
  n, h, wd, cin = x_shape
  return wd == 64 and h % 8 == 0 and 4 <= cin <= 20 and cout <= 32


def conv4x4s2_fwd_planes(x, w, bias, y, act, leak=0.2, zmask=None):
  """``conv4x4s2_fwd`` (or, with ``zmask`` and no bias, ``conv4x4s2_fwd_mask``) of a first layer whose input channels 3 .. are
  per-image constants (``planes_concat`` / ``net_inputs`` / ``critic_penalty_tangent`` outputs): K = 48 instead of 16 cin
  (expo_conv4x4s2_fwd_planes).  The caller vouches for the planes being constant."""
  lib = load()
  n, h, wd, cin, cout = _conv_args(x.shape, w)
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  assert y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (n, h // 2, wd // 2, cout)
  if bias is not None:
    _f32(bias, 'bias', (cout,))
  if zmask is not None:
    assert zmask.dtype == torch.float32 and zmask.is_contiguous() and zmask.shape == y.shape
  with torch.cuda.device(x.device):
    _check(lib.expo_conv4x4s2_fwd_planes(_ptr(x), _ptr(w), _ptr(bias), _ptr(zmask), _ptr(y), n, h, wd, cin, cout, int(act),
                                         float(leak), _stream()), 'expo_conv4x4s2_fwd_planes')


def conv4x4s2_fwd_planes_pair(a, b, act, leak=0.2):
  """``conv4x4s2_fwd_planes`` of two problems of one geometry as ONE grid: ``a`` / ``b`` = (x, w, bias, y)."""
  lib = load()
  (xa, wa, ba, ya), (xb, wb, bb, yb) = a, b
  n, h, wd, cin, cout = _conv_args(xa.shape, wa)
  assert _conv_args(xb.shape, wb) == (n, h, wd, cin, cout) and (ba is None) == (bb is None)
  for x, y, bias in ((xa, ya, ba), (xb, yb, bb)):
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    assert y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (n, h // 2, wd // 2, cout)
    if bias is not None:
      _f32(bias, 'bias', (cout,))
  with torch.cuda.device(xa.device):
    _check(lib.expo_conv4x4s2_fwd_planes_pair(_ptr(xa), _ptr(wa), _ptr(ba), _ptr(ya), _ptr(xb), _ptr(wb), _ptr(bb), _ptr(yb), n, h,
                                              wd, cin, cout, int(act), float(leak), _stream()), 'expo_conv4x4s2_fwd_planes_pair')


def conv4x4s2_fwd_pair(a, b, act, leak=0.2):
  """``conv4x4s2_fwd`` of two problems of one geometry as ONE grid (expo_conv4x4s2_fwd_pair): ``a`` / ``b`` = (x, w, bias, y).
  Planned for the grid that runs (twice the blocks): the two calls' results up to the order of the K slices' sum."""
  lib = load()
  (xa, wa, ba, ya), (xb, wb, bb, yb) = a, b
  n, h, wd, cin, cout = _conv_args(xa.shape, wa)
  assert _conv_args(xb.shape, wb) == (n, h, wd, cin, cout) and (ba is None) == (bb is None)
  for x, y, bias in ((xa, ya, ba), (xb, yb, bb)):
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    assert y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (n, h // 2, wd // 2, cout)
    if bias is not None:
      _f32(bias, 'bias', (cout,))
  with torch.cuda.device(xa.device):
    _check(lib.expo_conv4x4s2_fwd_pair(_ptr(xa), _ptr(wa), _ptr(ba), _ptr(ya), _ptr(xb), _ptr(wb), _ptr(bb), _ptr(yb), n, h, wd,
                                       cin, cout, int(act), float(leak), _stream()), 'expo_conv4x4s2_fwd_pair')


def conv4x4s2_bwd_data_mask_pair(a, b, leak=0.2):
  """``conv4x4s2_bwd_data_mask`` of two problems of one geometry as ONE grid: ``a`` / ``b`` = (dy, w, zmask, dx)."""
  lib = load()
  (dya, wa, za, dxa), (dyb, wb, zb, dxb) = a, b
  n, h, wd, cin, cout = _conv_args(dxa.shape, wa)
  assert _conv_args(dxb.shape, wb) == (n, h, wd, cin, cout)
  for dy, z, dx in ((dya, za, dxa), (dyb, zb, dxb)):
    assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
    assert dx.dtype == torch.float32 and dx.is_contiguous()
    assert z.dtype == torch.float32 and z.is_contiguous() and z.shape == dx.shape
  with torch.cuda.device(dya.device):
    _check(lib.expo_conv4x4s2_bwd_data_mask_pair(_ptr(dya), _ptr(wa), _ptr(za), _ptr(dxa), _ptr(dyb), _ptr(wb), _ptr(zb), _ptr(dxb),
                                                 n, h, wd, cin, cout, float(leak), _stream()),
           'expo_conv4x4s2_bwd_data_mask_pair')


def conv4x4s2_fwd_mask(x, w, zmask, y, leak=0.2):
  """y = conv(x, w) * slope(zmask) (expo_conv4x4s2_fwd_mask); ``y`` may be ``zmask`` itself (written in place)."""
  lib = load()
  n, h, wd, cin, cout = _conv_args(x.shape, w)
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  for t in (y, zmask):
    assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n, h // 2, wd // 2, cout)
  with torch.cuda.device(x.device):
    _check(lib.expo_conv4x4s2_fwd_mask(_ptr(x), _ptr(w), _ptr(zmask), _ptr(y), n, h, wd, cin, cout, float(leak),
                                       _stream()), 'expo_conv4x4s2_fwd_mask')


def conv4x4s2_wrw_bias(x, dy, dw, dbias, bias_images=None):
  """conv4x4s2_wrw and, in the same pass, ``dbias[co]`` = the sum of ``dy[..., co]`` over the first ``bias_images``
  images (default: all) -- expo_conv4x4s2_wrw_bias."""
  lib = load()
  n, h, wd, cin = x.shape
  cout = dy.shape[-1]
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
  assert dw.dtype == torch.float32 and tuple(dw.shape) == (cout, cin, 4, 4) and dw.permute(0, 2, 3, 1).is_contiguous()
  _f32(dbias, 'dbias', (cout,))
  need = int(lib.expo_conv4x4s2_wrw_workspace_bytes(n, h, wd, cin, cout))
  ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
  with torch.cuda.device(x.device):
    _check(lib.expo_conv4x4s2_wrw_bias(_ptr(x), _ptr(dy), _ptr(dw), _ptr(dbias), n if bias_images is None else int(bias_images),
                                       n, h, wd, cin, cout, ctypes.c_void_p(ws.data_ptr() if ws is not None else 0),
                                       ctypes.c_size_t(ws.numel() if ws is not None else 0), _stream()),
           'expo_conv4x4s2_wrw_bias')


def conv4x4s2_wrw_group(items):
  """The weight and bias gradients of a stack of layers with one reduce launch (expo_conv4x4s2_wrw_group).  ``items``:
  up to 8 tuples ``(x, dy, dw, dbias or None, bias_images or None)`` as for :func:`conv4x4s2_wrw_bias`."""
  lib = load()
  k = len(items)
  assert 1 <= k <= 8
  xs, dys, dws, dbs, bis, ns, hs, ws_, cis, cos, wss, wsb, keep = [], [], [], [], [], [], [], [], [], [], [], [], []
  for x, dy, dw, db, bias_images in items:
    n, h, wd, cin = x.shape
    cout = dy.shape[-1]
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (n, h // 2, wd // 2, cout)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (cout, cin, 4, 4) and dw.permute(0, 2, 3, 1).is_contiguous()
    if db is not None:
      _f32(db, 'dbias', (cout,))
    need = int(lib.expo_conv4x4s2_wrw_workspace_bytes(n, h, wd, cin, cout))
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    keep.append(ws)
    xs.append(x.data_ptr()), dys.append(dy.data_ptr()), dws.append(dw.data_ptr()), dbs.append(db.data_ptr() if db is not None else 0)
    bis.append(n if bias_images is None else int(bias_images)), ns.append(n), hs.append(h), ws_.append(wd), cis.append(cin), cos.append(cout)
    wss.append(ws.data_ptr() if ws is not None else 0), wsb.append(ws.numel() if ws is not None else 0)
  vp = lambda v: (_vp * k)(*v)
  ia = lambda v: (_i * k)(*v)
  with torch.cuda.device(items[0][0].device):
    _check(lib.expo_conv4x4s2_wrw_group(k, vp(xs), vp(dys), vp(dws), vp(dbs), ia(bis), ia(ns), ia(hs), ia(ws_), ia(cis), ia(cos),
                                        vp(wss), (_sz * k)(*wsb), _stream()), 'expo_conv4x4s2_wrw_group')


def critic_head_fwd(hpre, w2, b2, n_real, n_fake, n_interp, inv_n, logits, h, dh, leak=0.2, b1=None):
  """expo_critic_head_fwd: fc1 activation, fc2 and the rows' upstream gradients of the batched critic pass.  ``hpre``
  (m, hidden) is fc1's pre-activation, or (slabs, m, hidden) partial sums of it (``fc_fwd_slabs``) with fc1's bias ``b1``."""
  lib = load()
  slabs = 1 if hpre.dim() == 2 else int(hpre.shape[0])
  m, hidden = hpre.shape[-2:]
  assert m == n_real + n_fake + n_interp
  _f32(hpre, 'hpre', (m, hidden) if hpre.dim() == 2 else (slabs, m, hidden))
  for t in (h, dh):
    _f32(t, 'h / dh', (m, hidden))
  assert w2.is_cuda and w2.dtype == torch.float32 and w2.is_contiguous() and w2.numel() == hidden
  assert b2.is_cuda and b2.dtype == torch.float32 and b2.numel() == 1
  if b1 is not None:
    _f32(b1, 'b1', (hidden,))
  _f32(logits, 'logits', (m,))
  with torch.cuda.device(hpre.device):
    _check(lib.expo_critic_head_fwd(_ptr(hpre), _ptr(b1), slabs, _ptr(w2), _ptr(b2), int(n_real), int(n_fake), int(n_interp),
                                    hidden, float(inv_n), float(leak), _ptr(logits), _ptr(h), _ptr(dh), _stream()),
           'expo_critic_head_fwd')


def fc_fwd_slabs_count(m, k):
  """How many partial sums ``fc_fwd_slabs`` writes for m rows of k features; 0: take the library GEMM."""
  return int(load().expo_fc_fwd_slabs_count(int(m), int(k)))


def fc_fwd_slabs(x, w, slabs):
  """expo_fc_fwd_slabs: slabs[s] = x[:, K_s] w[:, K_s]^T over the s-th range of features (x (m, k), w (n, k) = nn.Linear's
  weight, slabs (fc_fwd_slabs_count(m, k), m, n)); sum(slabs, 0) + bias is the layer's pre-activation."""
  lib = load()
  m, k = x.shape
  n = w.shape[0]
  _f32(x, 'x', (m, k)), _f32(w, 'w', (n, k))
  _f32(slabs, 'slabs', (fc_fwd_slabs_count(m, k), m, n))
  with torch.cuda.device(x.device):
    _check(lib.expo_fc_fwd_slabs(_ptr(x), _ptr(w), _ptr(slabs), m, n, k, _stream()), 'expo_fc_fwd_slabs')


def fc_bwd_data_mask(dh, w, z, gy, leak=0.2):
  """expo_fc_bwd_data_mask: gy = (dh w) * slope(z) -- fc1's data gradient times the activation gradient of the feature map
  below it (dh (m, j), w (j, c), z / gy any shape of m * c elements)."""
  lib = load()
  m, j = dh.shape
  c = w.shape[1]
  _f32(dh, 'dh', (m, j)), _f32(w, 'w', (j, c))
  assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and z.numel() == m * c
  assert gy.is_cuda and gy.dtype == torch.float32 and gy.is_contiguous() and gy.numel() == m * c
  with torch.cuda.device(dh.device):
    _check(lib.expo_fc_bwd_data_mask(_ptr(dh), _ptr(w), _ptr(z), _ptr(gy), m, j, c, float(leak), _stream()),
           'expo_fc_bwd_data_mask')


def critic_report(logits, norm, term, n_real, n_fake, n_interp, lam, out, ema=None, decay=0.99, adam_step=None):
  """expo_critic_report: out[0..4] = c_loss, emd, mean gradient norm, gradient penalty, c_average; ``ema`` (a device
  scalar, optional) advances in the same launch, and so does ``adam_step`` (the step counter of the Adam update behind this
  launch: ``HipAdam.step(advanced=True)``)."""
  lib = load()
  _f32(logits, 'logits', (n_real + n_fake + n_interp,))
  _f32(norm, 'norm', (n_interp,)), _f32(term, 'term', (n_interp,))
  assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == 5
  assert ema is None or (ema.is_cuda and ema.dtype == torch.float32 and ema.numel() == 1)
  with torch.cuda.device(logits.device):
    _check(lib.expo_critic_report(_ptr(logits), _ptr(norm), _ptr(term), int(n_real), int(n_fake), int(n_interp), float(lam),
                                  float(decay), _ptr(out), _ptr(ema), _ptr(_step_scalar(adam_step)), _stream()),
           'expo_critic_report')


def fc_wrw(dh, x, dw):
  """expo_fc_wrw: dw = dh^T x -- the weight gradient of an nn.Linear (dh (m, j), x (m, c), dw (j, c)) over m rows."""
  lib = load()
  m, j = dh.shape
  c = x.shape[1]
  _f32(dh, 'dh', (m, j)), _f32(x, 'x', (m, c)), _f32(dw, 'dw', (j, c))
  with torch.cuda.device(dh.device):
    _check(lib.expo_fc_wrw(_ptr(dh), _ptr(x), _ptr(dw), m, j, c, _stream()), 'expo_fc_wrw')


def critic_head_bwd(dh, h, thpre, n_real, n_fake, n_interp, inv_n, gb1, gw2, gb2, leak=0.2):
  """expo_critic_head_bwd: fc1 bias / fc2 weight / fc2 bias gradients of the loss rows and of the penalty's tangent
  (``thpre`` (n_interp, hidden), or (slabs, n_interp, hidden) partial sums of it)."""
  lib = load()
  m, hidden = h.shape
  assert m == n_real + n_fake + n_interp
  _f32(dh, 'dh', (m, hidden)), _f32(h, 'h', (m, hidden))
  th_slabs = 1
  if n_interp:
    th_slabs = 1 if thpre.dim() == 2 else int(thpre.shape[0])
    _f32(thpre, 'thpre', (n_interp, hidden) if thpre.dim() == 2 else (th_slabs, n_interp, hidden))
  for t, k in ((gb1, hidden), (gw2, hidden), (gb2, 1)):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == k
  with torch.cuda.device(h.device):
    _check(lib.expo_critic_head_bwd(_ptr(dh), _ptr(h), _ptr(thpre), th_slabs, int(n_real), int(n_fake), int(n_interp),
                                    hidden, float(inv_n), float(leak), _ptr(gb1), _ptr(gw2), _ptr(gb2), _stream()),
           'expo_critic_head_bwd')


def plane_sums(x, sums, first):
  """sums[n, c - first] = x[n, ..., c] summed over the pixels, first <= c < C (expo_plane_sums)."""
  lib = load()
  n, c = x.shape[0], x.shape[-1]
  assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
  _f32(sums, 'sums', (n, c - first))
  with torch.cuda.device(x.device):
    _check(lib.expo_plane_sums(_ptr(x), _ptr(sums), n, x[0].numel() // c if n else 0, c, int(first), _stream()),
           'expo_plane_sums')


def gp_direct(u, ds, scale, v, norm, term):
  """g = u[..., :3] + ds; per image norm / penalty term and v = scale * d term / d g in one launch (expo_gp_direct)."""
  lib = load()
  n, c = u.shape[0], u.shape[-1]
  assert u.is_cuda and u.dtype == torch.float32 and u.is_contiguous() and c >= 3
  for t in (ds, v):
    assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(u.shape[:-1]) + (3,)
  _f32(norm, 'norm', (n,)), _f32(term, 'term', (n,))
  with torch.cuda.device(u.device):
    _check(lib.expo_gp_direct(_ptr(u), c, _ptr(ds), float(scale), _ptr(v), _ptr(norm), _ptr(term), n,
                              u[0].numel() // c if n else 0, _stream()), 'expo_gp_direct')


def critic_penalty_tangent(u, x, stats, scale, t0, norm, term):
  """expo_critic_penalty_tangent: plane sums, J^T, norm / term / penalty gradient, J v and the tangent's 6-plane input in one
  launch per batch of interpolated images."""
  lib = load()
  n, h, w, c = u.shape
  assert c == 6 and u.is_cuda and u.dtype == torch.float32 and u.is_contiguous()
  assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (n, h, w, 3)
  assert t0.dtype == torch.float32 and t0.is_contiguous() and t0.shape == u.shape
  _f32(stats, 'stats', (n, 3)), _f32(norm, 'norm', (n,)), _f32(term, 'term', (n,))
  with torch.cuda.device(u.device):
    _check(lib.expo_critic_penalty_tangent(_ptr(u), _ptr(x), _ptr(stats), float(scale), _ptr(t0), _ptr(norm), _ptr(term), n, h, w,
                                           _stream()), 'expo_critic_penalty_tangent')


def conv_wrw_tuning(slices=0, parts=0):
  """Probes / tests: waves per block and blocks per tile of the weight-gradient kernel (expo_conv_wrw_tuning)."""
  _check(load().expo_conv_wrw_tuning(int(slices), int(parts)), 'expo_conv_wrw_tuning')


def chain_prepare(stream=None):
  """Probe the helper-stream pairing of ``stream`` (default: torch's current stream) now, so that the first eager
  two-stream chain call does not stall it (expo_chain_prepare; idempotent)."""
  s = _stream() if stream is None else ctypes.c_void_p(int(getattr(stream, 'cuda_stream', stream)))
  _check(load().expo_chain_prepare(s), 'expo_chain_prepare')


def chain_release(stream):
  """Forget the helper-stream pairing of a stream that is about to be destroyed (expo_chain_release)."""
  s = ctypes.c_void_p(int(getattr(stream, 'cuda_stream', stream)))
  _check(load().expo_chain_release(s), 'expo_chain_release')


def apply_dispatch_fwd(ids, x, y, params, mask_params, maximum_sharpness, minimum_strength):
  """Per-image masked apply: image n goes through filter ids[n] with params[n] (N, 24) and mask_params[n] (N, 6)."""
  lib = load()
  _img(x, 'x'), _img(y, 'y')
  n, h, w, _ = x.shape
  _ids(ids, n)
  _f32(params, 'params', (n, EXPO_MAX_PARAMS))
  _f32(mask_params, 'mask_params', (n, 6))
  with torch.cuda.device(x.device):
    _check(lib.expo_filter_apply_dispatch_fwd(_ptr(ids), _ptr(x), _ptr(y), _ptr(params), _ptr(mask_params),
                                              float(maximum_sharpness), float(minimum_strength), n, h, w,
                                              _dtype_code(x), _stream()), 'expo_filter_apply_dispatch_fwd')


def apply_dispatch_bwd(ids, x, dy, dx, params, dparams, mask_params, dmask_params, maximum_sharpness, minimum_strength,
                       hsv_grad_mode=0, workspace=None):
  lib = load()
  _img(x, 'x'), _img(dy, 'dy')
  n, h, w, _ = x.shape
  _ids(ids, n)
  if dx is not None:
    _img(dx, 'dx')
  _f32(params, 'params', (n, EXPO_MAX_PARAMS))
  _f32(dparams, 'dparams', (n, EXPO_MAX_PARAMS))
  _f32(mask_params, 'mask_params', (n, 6))
  _f32(dmask_params, 'dmask_params', (n, 6))
  with torch.cuda.device(x.device):
    wsp, wsb = _ws(x, workspace)
    _check(lib.expo_filter_apply_dispatch_bwd(_ptr(ids), _ptr(x), _ptr(dy), _ptr(dx), _ptr(params), _ptr(dparams),
                                              _ptr(mask_params), _ptr(dmask_params), float(maximum_sharpness),
                                              float(minimum_strength), n, h, w, _dtype_code(x), hsv_grad_mode, wsp, wsb,
                                              _stream()), 'expo_filter_apply_dispatch_bwd')
