"""High-resolution inference loop (``/root/reference/net.py:711-821``, ``evaluate.py:8-31``).

Per image: a 64x64 proxy is made from the full-resolution picture (``net.py:779``: bilinear resize
of the centre crop), and for ``cfg.test_steps`` (= 5) steps the agent regresses parameters and picks
a filter on the PROXY while the same per-image parameters are applied to BOTH the proxy and the
full-resolution tensor (``filters.py:88-96``; ``agent.py:124-129``), feeding both outputs back
(``net.py:796-821``).  ``is_train = 0`` -> the action is ``argmax(pdf)`` (``agent.py:114-116``);
dropout stays on, as in the reference (``agent.py:36``), unless masks are passed.

Image decoding (16-bit TIFF / ProPhoto linearisation, ``util.py:311-323, 495-501``) and the PNG /
pickle outputs of ``net.py:825-877`` are file I/O, out of scope: this function works on tensors.
"""
import torch

from .util import STATE_STOPPED_DIM


def linearize_ProPhotoRGB(pp_rgb):
  """util.py:495-501 as used by net.py:733 (``reverse=False``): gamma 1.8 decode."""
  return pp_rgb**1.8


def get_image_center(image):
  """util.py:160-167: the largest centred square (NHWC)."""
  h, w = image.shape[1], image.shape[2]
  if h > w:
    o = (h - w) // 2
    return image[:, o:o + w]
  o = (w - h) // 2
  return image[:, :, o:o + h]


def make_low_res(high_res, size):
  """net.py:779: cv2.resize(get_image_center(hi), (size, size)) -- bilinear, half-pixel centres."""
  c = get_image_center(high_res).permute(0, 3, 1, 2).float()
  low = torch.nn.functional.interpolate(c, size=(size, size), mode='bilinear', align_corners=False,
                                        antialias=False)
  return low.permute(0, 2, 3, 1).contiguous().to(high_res.dtype)


def fused_chain(high_res, filter_ids, params24):
  """Apply a per-image sequence of filters to the full-resolution image in ONE pass
  (``expo_chain_fused_fwd``).  filter_ids: (N, steps) int32 C-ABI ids; params24: (N, steps, 24)."""
  from . import _cabi
  out = torch.empty_like(high_res)
  _cabi.chain_fused_fwd(filter_ids.contiguous().to(torch.int32), params24.contiguous().float(),
                        high_res.contiguous(), out)
  return out


@torch.no_grad()
def retouch(agent, high_res, steps=None, z=None, dropout_masks=None, return_trace=False, fused=True):
  """Run the 5-step retouching loop.  ``high_res``: NHWC device tensor (fp16/fp32), linear RGB.
  Returns (retouched_high_res, retouched_low_res, states[, trace of selected filter ids]).

  ``fused=True`` (default): the agent steps run on the 64x64 proxy only, recording each step's
  (filter id, parameters); the full-resolution image is then read once, pushed through all steps
  in registers and written once.  ``fused=False`` is the reference's schedule (``net.py:796-821``):
  every step also filters the full-resolution tensor and feeds it back -- identical maths, one
  fp16 rounding per step, ``steps`` times the HBM traffic."""
  cfg = agent.cfg
  steps = steps or cfg.test_steps
  n = high_res.shape[0]
  dev = high_res.device
  low = make_low_res(high_res, cfg.source_img_size)
  states = torch.zeros((n, cfg.num_state_dim), dtype=torch.float32, device=dev)  # get_initial_states
  if z is None:
    z = torch.rand((n, cfg.z_dim), device=dev)
  trace, abi_ids, params = [], [], []
  hi = high_res.contiguous()
  for i in range(steps):
    masks = dropout_masks[i] if dropout_masks is not None else None
    if fused:
      (low, states, _s, _p), dbg, _ = agent((low, z, states), is_train=0, progress=0.0, dropout_masks=masks)
      abi_ids.append(dbg['abi_filter_ids'])
      params.append(dbg['params24'])
    else:
      (low, states, hi), dbg, _ = agent((low, z, states), is_train=0, progress=0.0, high_res=hi,
                                        dropout_masks=masks)
    trace.append(dbg['selected_filter_ids'].clone())
    if bool((states[:, STATE_STOPPED_DIM] > 0).all()):
      break
  if fused:
    hi = fused_chain(hi, torch.stack(abi_ids, dim=1), torch.stack(params, dim=1))
  if return_trace:
    return hi, low, states, torch.stack(trace, dim=1)
  return hi, low, states


def load_image(path):
  """net.py:726-747: ``.tif`` -> 16-bit ProPhoto, linearised (x**1.8); anything else readable by
  PIL -> 8-bit sRGB-ish, ``/255``, ``**2.2``, scaled by ``1 / (2 max)``."""
  import numpy as np
  if path.lower().endswith(('.tif', '.tiff')):
    from .tiff16 import read_tiff16
    return linearize_ProPhotoRGB(read_tiff16(path))
  from PIL import Image
  img = (np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0)**2.2
  return img / (2 * img.max())


def main(argv=None):
  """``python -m exposure_amd.evaluate [--weights w.pt] [--out out.npy] img ...`` -- the tensor part
  of ``evaluate.py:8-31``: 5 retouching steps per image on the GPU; writes the linear result."""
  import argparse
  import numpy as np
  from .agent import Agent
  from .config import make_cfg
  ap = argparse.ArgumentParser()
  ap.add_argument('images', nargs='+')
  ap.add_argument('--weights', default=None, help='torch state_dict of exposure_amd.agent.Agent (random init if absent)')
  ap.add_argument('--out', default=None)
  ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
  args = ap.parse_args(argv)
  dev = torch.device('cuda:0')
  cfg = make_cfg()
  agent = Agent(cfg).to(dev)
  if args.weights:
    agent.load_state_dict(torch.load(args.weights, map_location=dev))
  dt = torch.float16 if args.dtype == 'f16' else torch.float32
  for path in args.images:
    hi = torch.from_numpy(np.ascontiguousarray(load_image(path))).to(dev).to(dt)[None]
    out, _low, states, trace = retouch(agent, hi, return_trace=True)
    names = [agent.filters[int(j)].get_short_name() for j in trace[0]]
    print('%s: %dx%d  filters: %s' % (path, hi.shape[2], hi.shape[1], ' '.join(names)))
    np.save(args.out or (path + '.retouched.npy'), out[0].float().cpu().numpy())


if __name__ == '__main__':
  main()
