"""NumPy restatement of the callers either side of the filter path: action sampling, state
update, penalties (``agent.py:80-125, 207-252``; ``pdf_sample_layer.py:5-10``) and the critic's
per-image statistics (``critics.py:48-62``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); PARITY UNPINNED by the reference.
The only reference-supplied known answer is the ``pdf = (2,4,8)`` sampling frequencies of
``pdf_sample_layer.py:55-78`` (expected 1/7, 2/7, 4/7), restated as interval checks in the tests.
"""
import math

import numpy as np

from . import filters_np as fnp

# util.py:13-16
STATE_REWARD_DIM = 0
STATE_STOPPED_DIM = 1
STATE_STEP_DIM = 2
STATE_DROPOUT_BEGIN = 3


def exclusive_cumsum(pdf):
  """tf.cumsum(pdf, axis=1, exclusive=True) (pdf_sample_layer.py:7): a shifted prefix scan, out[:, 0] = 0
  and out[:, j] = out[:, j-1] + pdf[:, j-1], accumulated left to right in the dtype of ``pdf`` (TF's CPU
  scan functor walks the axis sequentially).  Not ``cumsum(pdf) - pdf``, which differs by an ulp in fp32."""
  out = np.zeros_like(pdf)
  for j in range(1, pdf.shape[1]):
    out[:, j] = out[:, j - 1] + pdf[:, j - 1]
  return out


def row_sum(pdf):
  """tf.reduce_sum(pdf, axis=1, keep_dims=True) (pdf_sample_layer.py:6) with an explicit association
  order (the device/library default tree must not decide an integer output): K == 8 follows Eigen's
  packet reducer as TF-1 CPU kernels run it -- ((p0+p4)+(p2+p6)) + ((p1+p5)+(p3+p7)) -- any other K
  sums left to right.  TensorFlow is absent from the tree: this order is PARITY UNPINNED."""
  k = pdf.shape[1]
  c = [pdf[:, j] for j in range(k)]
  if k == 8:
    total = ((c[0] + c[4]) + (c[2] + c[6])) + ((c[1] + c[5]) + (c[3] + c[7]))
  else:
    total = c[0]
    for j in range(1, k):
      total = total + c[j]
  return total[:, None]


def pdf_sample(pdf, uniform_noise):
  """pdf_sample_layer.py:5-10.  pdf (N,K); uniform_noise (N,1) -> int32 (N,).
  cumsum(exclusive) < u summed, minus one; u == 0 gives -1 (nothing selected)."""
  pdf = pdf / (row_sum(pdf) + pdf.dtype.type(1e-36))
  cdf = exclusive_cumsum(pdf)
  return (cdf < uniform_noise).astype(np.int32).sum(axis=1) - 1


def softmax(x):
  e = np.exp(x - x.max(axis=1, keepdims=True))
  return e / e.sum(axis=1, keepdims=True)


def action_selection(logits, selection_noise, is_train, exploration=0.05):
  """agent.py:101-121 from the selector FC2 output to (pdf, entropy, selected id, one-hot, surrogate)."""
  k = logits.shape[1]
  dt = logits.dtype.type
  pdf = softmax(logits) + dt(1e-37)
  pdf = pdf * dt(1 - exploration) + dt(exploration * 1.0 / k)
  pdf = pdf / (np.sum(pdf, axis=1, keepdims=True) + dt(1e-30))
  entropy = np.sum(-pdf * np.log(pdf), axis=1)[:, None]
  random_filter_id = pdf_sample(pdf, selection_noise)
  max_filter_id = np.argmax(pdf, axis=1).astype(np.int32)
  selected = is_train * random_filter_id + (1 - is_train) * max_filter_id
  one_hot = (selected[:, None] == np.arange(k)[None, :]).astype(logits.dtype)  # tf.one_hot: -1 -> zeros
  surrogate = np.sum(one_hot * np.log(pdf + dt(1e-10)), axis=1, keepdims=True)
  return pdf, entropy, selected.astype(np.int32), one_hot, surrogate


def new_states(states, one_hot, test_steps=5):
  """agent.py:207-238."""
  is_last_step = (np.abs(states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1] + 1 - test_steps) < 1e-4).astype(
      states.dtype)
  submitted = is_last_step
  step = (states[:, STATE_STEP_DIM] + 1)[:, None]
  filter_usage = states[:, STATE_STEP_DIM + 1:]
  usage_penalty = np.sum(filter_usage * one_hot, axis=1, keepdims=True)
  new_usage = np.maximum(filter_usage, one_hot)
  out = np.concatenate([submitted, submitted, step, new_usage], axis=1)
  return out, usage_penalty, is_last_step, submitted


def overexposure_penalty(img):
  """agent.py:249-251: reduce_mean(maximum(net - 1, 0)**2, axis=(1,2,3))."""
  return np.mean(np.maximum(img - 1, 0)**2, axis=(1, 2, 3))


def penalty(img, entropy, usage_penalty, is_last_step, submitted, progress, k=8, exploration_penalty=0.05,
            filter_usage_penalty=1.0, early_stop_penalty=1.0):
  """agent.py:226-252 -> (N,1)."""
  early = (1 - is_last_step) * submitted * early_stop_penalty
  entropy_penalty = (1.0 - progress) * exploration_penalty * (-entropy + math.log(k))
  return overexposure_penalty(img)[:, None] + entropy_penalty + usage_penalty * filter_usage_penalty + early


def select_filtered(filtered_images, one_hot):
  """agent.py:77,124-125: stack on axis 1, multiply by the one-hot, reduce_sum."""
  stack = np.stack(filtered_images, axis=1)
  return np.sum(stack * one_hot[:, :, None, None, None], axis=1)


def apply_all_and_select(img, packed_params, one_hot):
  """The reference's per-step image path: all 8 filters on the same input, then one-hot select."""
  outs = [fnp.process_packed(fid, img, packed_params[fid]) for fid in range(8)]
  return select_filtered(outs, one_hot)


def critic_stats(images):
  """critics.py:48-62 -> (N,3) = [mean lum, variance lum (tf.nn.moments), mean saturation]."""
  dt = images.dtype.type
  lum = images[:, :, :, 0] * dt(0.27) + images[:, :, :, 1] * dt(0.67) + images[:, :, :, 2] * dt(0.06) + dt(1e-5)
  luminance = lum.mean(axis=(1, 2))
  contrast = lum.var(axis=(1, 2))
  clipped = np.clip(images, 0.0, 1.0)
  i_max = clipped.max(axis=3)
  i_min = clipped.min(axis=3)
  sat = (i_max - i_min) / (np.minimum(i_max + i_min, dt(2.0) - i_max - i_min) + dt(1e-2))
  saturation = sat.mean(axis=(1, 2))
  return np.stack([luminance, contrast, saturation], axis=1)
