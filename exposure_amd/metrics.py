"""The paper's evaluation metric (``/root/reference/histogram_intersection.py``): histogram
intersection of per-image luminance mean, contrast (2 x luminance std) and HLS saturation between a
set of retouched images and a set of target images.  Tensor-in / number-out (the reference's file
listing, ``cv2.imread`` and random 64x64 crop sampling, ``histogram_intersection.py:36-59``, are
dataset I/O and stay outside).  Runs on whatever device the images live on."""
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
