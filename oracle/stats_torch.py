"""Differentiable torch restatement of the critic's statistics (``/root/reference/critics.py:48-73``), op by op
as the reference graph composes them, so torch autograd supplies the first AND the second derivative.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the product computes the statistics and their
derivatives in HIP kernels (``exposure_amd/critics.py``); tests use this to check the double backward of the
gradient penalty (net.py:174-194) end to end -- the critic-weight gradients of ``c_loss`` must agree between the
HIP path and this autograd path.  PARITY UNPINNED by the reference (no tests, TensorFlow absent); pinned by the
finite-difference checks of ``oracle/nets_np.py`` it must agree with (tests/test_oracle_nets.py)."""
import torch


def stat_features(images):
  """NHWC ``images`` -> (N, 3) [mean lum, population variance of lum, mean saturation]."""
  lum = images[:, :, :, 0] * 0.27 + images[:, :, :, 1] * 0.67 + images[:, :, :, 2] * 0.06 + 1e-5
  luminance = lum.mean(dim=(1, 2))
  contrast = lum.var(dim=(1, 2), unbiased=False)  # tf.nn.moments: population variance
  clipped = images.clamp(0.0, 1.0)
  # amax/amin split the gradient evenly between tied maxima, like tf.reduce_max/min
  i_max = clipped.amax(dim=3)
  i_min = clipped.amin(dim=3)
  a, b = i_max + i_min, 2.0 - i_max - i_min
  sat = (i_max - i_min) / (torch.where(a <= b, a, b) + 1e-2)  # tf.minimum(x, y): ties go to x
  saturation = sat.mean(dim=(1, 2))
  return torch.stack([luminance, contrast, saturation], dim=1)
