"""High-resolution inference loop (``/root/reference/net.py:711-821``, ``evaluate.py:8-31``).

Per image: a 64x64 proxy is made from the full-resolution picture (``net.py:779``: bilinear resize
of the centre crop), and for ``cfg.test_steps`` (= 5) steps the agent regresses parameters and picks
a filter on the PROXY while the same per-image parameters are applied to BOTH the proxy and the
full-resolution tensor (``filters.py:88-96``; ``agent.py:124-129``), feeding both outputs back
(``net.py:796-821``).  ``is_train = 0`` -> the action is ``argmax(pdf)`` (``agent.py:114-116``);
dropout stays on, as in the reference (``agent.py:36``), unless masks are passed.

Image decoding (16-bit TIFF / ProPhoto linearisation, ``util.py:311-323, 495-501``) sits in ``load_image``; of the
outputs of ``net.py:825-877`` the CLI writes the linear result (.npy) and, with ``--png``, the reference's 8-bit
``retouched`` / ``input_tone_mapped`` pictures; the debug pickle and the cv2-drawn ``steps`` panel are out of scope.
"""
import numpy as np
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
  if cfg.masking:
    fused = False  # the spatial mask depends on the running image: no parameters-only replay
  generic = any(f.uses_generic_kernels() for f in agent.filters)
  if generic:
    fused = False  # cfg.curve_steps != 8: the one-pass kernel is instantiated for 8-step curves (reference schedule instead)
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
    else:
      (low, states, hi), dbg, _ = agent((low, z, states), is_train=0, progress=0.0, high_res=hi,
                                        dropout_masks=masks)
    if generic:  # no (N, 24) parameter rows exist for this configuration: record the ids only
      abi_ids.append(agent.abi_filter_ids[dbg['selected_filter_ids'].clamp_min(0).long()])
      params.append(torch.zeros((n, 24), dtype=torch.float32, device=dev))
    else:
      abi_ids.append(dbg['abi_filter_ids'])
      params.append(dbg['params24'])
    trace.append(dbg['selected_filter_ids'].clone())
    if bool((states[:, STATE_STOPPED_DIM] > 0).all()):
      break
  if fused:
    hi = fused_chain(hi, torch.stack(abi_ids, dim=1), torch.stack(params, dim=1))
  if return_trace == 'full':  # the per-step operations (what net.py:825-877 pickles as decisions / operations)
    return hi, low, states, dict(selected=torch.stack(trace, dim=1), abi_filter_ids=torch.stack(abi_ids, dim=1),
                                 params24=torch.stack(params, dim=1))
  if return_trace:
    return hi, low, states, torch.stack(trace, dim=1)
  return hi, low, states


def load_image(path):
  """net.py:726-747: ``.tif`` -> 16-bit ProPhoto, linearised (x**1.8); anything else readable by
  PIL -> sRGB-ish: ``/255`` (uint8) or ``/65535`` (uint16), ``**2.2``, scaled by ``1 / (2 max)``."""
  import numpy as np
  if path.lower().endswith(('.tif', '.tiff')):
    from .tiff16 import read_tiff16
    return linearize_ProPhotoRGB(read_tiff16(path))
  from PIL import Image
  pil = Image.open(path)
  raw = np.asarray(pil)
  if raw.dtype == np.uint16:  # net.py:738-739 (16-bit grey / multi-channel arrays PIL can deliver)
    img = raw.astype(np.float32) / 65535.0
    if img.ndim == 2:
      img = np.repeat(img[:, :, None], 3, axis=2)
    img = img[:, :, :3]
  else:
    img = np.asarray(pil.convert('RGB'), dtype=np.float32) / 255.0
  img = img**2.2  # linearise sRGB
  return img / (2 * img.max())  # mimic RAW exposure


def load_agent_weights(agent, state):
  """Accepts an ``Agent`` state dict, a ``GAN.state_dict()`` (keys prefixed 'generator.' / 'critic.' / 'value.') or the
  ``{'model': GAN.state_dict(), 'optim': ...}`` file that ``python -m exposure_amd.train --save`` writes: the
  generator's entries are picked out and the prefix stripped."""
  if 'model' in state and isinstance(state['model'], dict):
    state = state['model']
  if any(k.startswith('generator.') for k in state):
    state = {k[len('generator.'):]: v for k, v in state.items() if k.startswith('generator.')}
  agent.load_state_dict(state)
  return agent


def save_png(path, img):
  """``show_and_save`` of ``net.py:769-772``: ``cv2.imwrite(path, img[:, :, ::-1] * 255.0)`` -- 8 bits per channel,
  rounded to nearest (half to even, as ``cvRound``) and saturated; the file holds RGB."""
  from PIL import Image
  a = np.asarray(img, dtype=np.float32)
  Image.fromarray(np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8), 'RGB').save(path)
  return path


def tone_mapped_input(linear):
  """``net.py:822-823``: max to white, then gamma 1/2.4 -- the ``input_tone_mapped`` picture of ``GAN.eval``."""
  a = np.asarray(linear, dtype=np.float32)
  return (a / a.max())**(1 / 2.4)


def output_path(out, image_path, many):
  """--out: a directory (existing, or ending in a path separator) receives <name>.retouched.npy per
  image; with one image it may also be the file itself; with several images and a plain name the
  image's stem is inserted so results do not overwrite each other."""
  import os
  base = os.path.basename(image_path)
  if out is None:
    return image_path + '.retouched.npy'
  if os.path.isdir(out) or out.endswith(os.sep):
    os.makedirs(out, exist_ok=True)
    return os.path.join(out, base + '.retouched.npy')
  if not many:
    return out
  root, ext = os.path.splitext(out)
  return '%s.%s%s' % (root, os.path.splitext(base)[0], ext or '.npy')


FILTER_BY_SHORT_NAME = {'E': 'ExposureFilter', 'G': 'GammaFilter', 'W': 'ImprovedWhiteBalanceFilter',
                        'S+': 'SaturationPlusFilter', 'T': 'ToneFilter', 'Ct': 'ContrastFilter', 'BW': 'WNBFilter',
                        'C': 'ColorFilter'}


def main(argv=None):
  """``python -m exposure_amd.evaluate [--filters E,G] [--weights w.pt | --tf-checkpoint dir] [--out dir|file] img ...`` -- the
  tensor part of ``evaluate.py:8-31`` / ``GAN.eval`` (``net.py:711-821``): per image, load (16-bit TIFF or
  8-bit sRGB), 5 retouching steps on the GPU, write the linear result.  Returns one record per image."""
  import argparse
  import numpy as np
  from . import filters as F
  from .agent import Agent
  from .config import make_cfg
  ap = argparse.ArgumentParser()
  ap.add_argument('images', nargs='+')
  ap.add_argument('--weights', default=None,
                  help='torch state_dict of exposure_amd.agent.Agent, or the GAN state dict train.py --save writes '
                  '(random init if absent)')
  ap.add_argument('--tf-checkpoint', default=None, metavar='MODEL_DIR',
                  help="directory of a TF-1 checkpoint of the reference (models/<cfg>/<name>): restores "
                  "MODEL_DIR/model.ckpt-<--ckpt> like evaluate.py:27-28 (no TensorFlow needed)")
  ap.add_argument('--ckpt', default='20000', help='checkpoint iteration for --tf-checkpoint (evaluate.py:28: 20000)')
  ap.add_argument('--out', default=None, help='output file (one image) or directory; default <image>.retouched.npy')
  ap.add_argument('--png', action='store_true',
                  help="also write <output>.png (8-bit, the reference's `<name>.retouched.png`, net.py:769-772, 832) "
                  'and, with --show-input, <output>.input_tone_mapped.png (net.py:822-829)')
  ap.add_argument('--show-input', action='store_true')
  ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
  ap.add_argument('--filters', default=None,
                  help="cfg.filters as comma-separated short names, e.g. 'E,G' (BASELINE config 1); default: all 8")
  ap.add_argument('--seed', type=int, default=None, help='seeds the random-init weights / dropout / noise')
  ap.add_argument('--stepwise', action='store_true', help="the reference's schedule: filter the full-resolution "
                  'tensor at every step instead of one fused pass at the end')
  args = ap.parse_args(argv)
  dev = torch.device('cuda:0')
  if args.seed is not None:
    torch.manual_seed(args.seed)
  flt = None
  if args.filters:
    flt = [getattr(F, FILTER_BY_SHORT_NAME[name.strip()]) for name in args.filters.split(',')]
  cfg = make_cfg(filters=flt)
  agent = Agent(cfg).to(dev)
  if args.weights and args.tf_checkpoint:
    ap.error('--weights and --tf-checkpoint are alternatives')
  if args.weights:
    load_agent_weights(agent, torch.load(args.weights, map_location=dev))
  if args.tf_checkpoint:
    from . import checkpoint
    checkpoint.restore(agent, args.tf_checkpoint, args.ckpt)
  dt = torch.float16 if args.dtype == 'f16' else torch.float32
  records = []
  for path in args.images:
    hi = torch.from_numpy(np.ascontiguousarray(load_image(path))).to(dev).to(dt)[None]
    out, _low, states, ops = retouch(agent, hi, return_trace='full', fused=not args.stepwise)
    trace = ops['selected']
    names = [agent.filters[int(j)].get_short_name() for j in trace[0]]
    print('%s: %dx%d  filters: %s' % (path, hi.shape[2], hi.shape[1], ' '.join(names)))
    dst = output_path(args.out, path, len(args.images) > 1)
    result = out[0].float().cpu().numpy()
    np.save(dst, result)
    pngs = {}
    if args.png:
      stem = dst[:-4] if dst.endswith('.npy') else dst
      pngs['retouched'] = save_png(stem + '.png', result)
      if args.show_input:
        pngs['input_tone_mapped'] = save_png(stem + '.input_tone_mapped.png', tone_mapped_input(hi[0].float().cpu().numpy()))
    records.append(dict(image=path, output=dst, png=pngs, filters=names, states=states[0].cpu().tolist(),
                        abi_filter_ids=ops['abi_filter_ids'][0].cpu().tolist(),
                        params24=ops['params24'][0].cpu().numpy()))
  return records


if __name__ == '__main__':
  main()
