"""The paper's evaluation metric (``/root/reference/histogram_intersection.py``): histogram
intersection of per-image luminance mean, contrast (2 x luminance std) and HLS saturation between a
set of retouched images and a set of target images.  The statistics are tensor-in / number-out and run on
whatever device the images live on; ``read_images`` and ``python -m exposure_amd.metrics OUT_DIR TARGET_DIR``
restate the script's file side (``histogram_intersection.py:36-76``: every file -> 4 random square crops ->
80x80 by area averaging -> 4 random 64x64 patches each) with PIL instead of cv2."""
import os
import random

import torch

HIST_BINS = 32  # histogram_intersection.py:8


def hls_saturation(img):
  """S channel of cv2.cvtColor(img, COLOR_RGB2HLS) for float images in [0,1] (NHWC):
  L = (max+min)/2;  S = (max-min)/(max+min) if L < 0.5 else (max-min)/(2-max-min);  0 if max == min."""
  mx = img.amax(dim=-1)
  mn = img.amin(dim=-1)
  d = mx - mn
  l = (mx + mn) * 0.5
  den = torch.where(l < 0.5, mx + mn, 2.0 - mx - mn)
  return torch.where(d > 0, d / den.clamp_min(1e-12), torch.zeros_like(d))


def get_statistics(images):
  """histogram_intersection.py:15-20 for a batch (N,H,W,3) -> (N,3) [lum mean, 2*lum std, mean sat]."""
  img = images.float().clamp(0.0, 1.0)
  lum = img[..., 0] * 0.27 + img[..., 1] * 0.67 + img[..., 2] * 0.06
  sat = hls_saturation(img).mean(dim=(1, 2))
  return torch.stack([lum.mean(dim=(1, 2)), lum.std(dim=(1, 2), unbiased=False) * 2, sat], dim=1)


def calc_hist(arr, nbins=HIST_BINS, xrange=(0.0, 1.0)):
  """histogram_intersection.py:23-25 (np.histogram semantics: values outside the range are dropped,
  the right edge is inclusive)."""
  arr = arr.float()
  inside = (arr >= xrange[0]) & (arr <= xrange[1])
  h = torch.histc(arr[inside], bins=nbins, min=xrange[0], max=xrange[1])
  return h / float(arr.numel())


def hist_intersection(a, b):
  """histogram_intersection.py:11-12."""
  return torch.minimum(a, b).sum()


def histogram_intersection(output_images, target_images):
  """histogram_intersection.py:62-76 -> (three intersections, their average), as floats in [0,1]."""
  so, st = get_statistics(output_images), get_statistics(target_images)
  ints = [float(hist_intersection(calc_hist(so[:, k]), calc_hist(st[:, k]))) for k in range(3)]
  return ints, sum(ints) / len(ints)


def read_images(src, tag=None, rng=None, device='cpu'):
  """histogram_intersection.py:36-59 -> (16 * files, 64, 64, 3) float32 in [0, 1].  ``rng``: a ``random.Random``
  (the script uses the module-level generator, i.e. a different sample every run).  The 80x80 reduction is an area
  average (``cv2.INTER_AREA``; for crops that are not a multiple of 80 the window edges are rounded to whole pixels
  here, where cv2 weights the partial ones)."""
  import numpy as np
  from PIL import Image
  rng = rng or random
  patches = []
  for f in sorted(os.listdir(src)):
    if tag and f.find(tag) == -1:
      continue
    image = np.asarray(Image.open(os.path.join(src, f)).convert('RGB'), dtype=np.float32) / 255.0
    edge = min(image.shape[0], image.shape[1])
    for _ in range(4):
      sx = rng.randrange(0, image.shape[0] - edge + 1)
      sy = rng.randrange(0, image.shape[1] - edge + 1)
      crop = torch.from_numpy(np.ascontiguousarray(image[sx:sx + edge, sy:sy + edge])).permute(2, 0, 1)[None]
      patch = torch.nn.functional.adaptive_avg_pool2d(crop, (80, 80))[0].permute(1, 2, 0)
      for _ in range(4):
        ssx = rng.randrange(0, 80 - 64)
        ssy = rng.randrange(0, 80 - 64)
        patches.append(patch[ssx:ssx + 64, ssy:ssy + 64])
  if not patches:
    raise FileNotFoundError('no images in %s' % src)
  return torch.stack(patches).to(device)


def main(argv=None):
  """``python -m exposure_amd.metrics OUTPUT_DIR TARGET_DIR`` -- histogram_intersection.py:62-76."""
  import sys
  argv = sys.argv[1:] if argv is None else argv
  if len(argv) != 2:
    raise SystemExit('usage: python -m exposure_amd.metrics OUTPUT_DIR TARGET_DIR')
  device = 'cuda:0' if torch.cuda.is_available() else 'cpu'
  ints, avg = histogram_intersection(read_images(argv[0], device=device), read_images(argv[1], device=device))
  print('Hist. Inter.: %.2f%% %.2f%% %.2f%%' % (ints[0] * 100, ints[1] * 100, ints[2] * 100))
  print('         Avg: %.2f%%' % (avg * 100))
  return ints, avg


if __name__ == '__main__':
  main()
