"""Weight import/export between the reference's TF-1 variable layout and this package's modules.

The reference saves ``tf.train.Saver`` checkpoints (``net.py:271,380-384``); reading a ``.ckpt``
needs TensorFlow, which this image does not have, and the pretrained ``models/`` submodule is empty
(``.gitmodules``).  What CAN be fixed here is the layout contract, so that a checkpoint dumped to
``{variable_name: ndarray}`` (e.g. ``np.savez`` of ``tf.train.load_checkpoint(...).get_tensor``)
loads directly:

* ``ly.conv2d`` kernels are HWIO -> ``nn.Conv2d`` OIHW;
* ``ly.fully_connected`` weights are (in, out) -> ``nn.Linear`` (out, in); the 4096-d feature is
  flattened in (H, W, C) order on both sides (``agent.py:35``, ``FeatureExtractor.forward``);
* variable scopes: ``generator/Conv{,_1,_2,_3}`` (shared feature extractor, ``agent.py:52-56``),
  ``generator/filter_<j>/fc1|fc2`` (``agent.py:59``, ``filters.py:31-42``),
  ``generator/action_selection/Conv*`` + ``selector_fc1|selector_fc2`` (``agent.py:80-99``),
  ``critic/Conv*`` + ``critic/fully_connected{,_1}`` (``critics.py:42-97``), and the value network
  under ``rl_value/critic/...`` (``net.py:76-90``).  TF names end in ``/weights`` and ``/biases``.
"""
import numpy as np
import torch


def _conv_names(prefix, n):
  return [prefix + ('Conv' if i == 0 else 'Conv_%d' % i) for i in range(n)]


def tf_name_map(gan):
  """[(tf_variable_name, torch_parameter, kind)] with kind in {'conv_w', 'fc_w', 'bias'}."""
  out = []

  def conv(prefix, convs):
    for name, m in zip(_conv_names(prefix, len(convs)), convs):
      out.append((name + '/weights', m.weight, 'conv_w'))
      out.append((name + '/biases', m.bias, 'bias'))

  def fc(name, m):
    out.append((name + '/weights', m.weight, 'fc_w'))
    out.append((name + '/biases', m.bias, 'bias'))

  g = gan.generator
  conv('generator/', g.filter_features.convs)
  for j, f in enumerate(g.filters):
    fc('generator/filter_%d/fc1' % j, f.fc1)
    fc('generator/filter_%d/fc2' % j, f.fc2)
  conv('generator/action_selection/', g.selector_features.convs)
  fc('generator/action_selection/selector_fc1', g.selector_fc1)
  fc('generator/action_selection/selector_fc2', g.selector_fc2)
  for scope, net in (('critic/', gan.critic), ('rl_value/critic/', gan.value)):
    conv(scope, net.convs)
    fc(scope + 'fully_connected', net.fc1)
    fc(scope + 'fully_connected_1', net.fc2)
  return out


def to_tf_layout(param, kind):
  a = param.detach().cpu().numpy()
  if kind == 'conv_w':
    return np.ascontiguousarray(a.transpose(2, 3, 1, 0))  # OIHW -> HWIO
  if kind == 'fc_w':
    return np.ascontiguousarray(a.T)  # (out,in) -> (in,out)
  return a.copy()


def from_tf_layout(array, kind):
  a = np.asarray(array)
  if kind == 'conv_w':
    return np.ascontiguousarray(a.transpose(3, 2, 0, 1))  # HWIO -> OIHW
  if kind == 'fc_w':
    return np.ascontiguousarray(a.T)
  return a.copy()


def export_tf_dict(gan):
  return {name: to_tf_layout(p, kind) for name, p, kind in tf_name_map(gan)}


def load_tf_dict(gan, weights, strict=True):
  """Copy ``{tf_name: ndarray}`` into the modules; returns the list of names that were missing."""
  missing = []
  with torch.no_grad():
    for name, p, kind in tf_name_map(gan):
      if name not in weights:
        missing.append(name)
        continue
      a = from_tf_layout(weights[name], kind)
      if tuple(a.shape) != tuple(p.shape):
        raise ValueError('%s: shape %s does not match %s' % (name, a.shape, tuple(p.shape)))
      p.copy_(torch.from_numpy(a).to(p.dtype))
  if strict and missing:
    raise KeyError('missing TF variables: %s' % ', '.join(missing[:5]))
  return missing


def load_tf_npz(gan, path, strict=True):
  with np.load(path) as z:
    return load_tf_dict(gan, {k: z[k] for k in z.files}, strict)
