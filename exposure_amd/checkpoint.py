"""Weight import/export between the reference's TF-1 variable layout and this package's modules.

The reference saves ``tf.train.Saver`` checkpoints (``net.py:271,380-384``) and ``evaluate.py:27-28``
restores ``model.ckpt-20000``.  ``tf_bundle.py`` reads and writes that file format without TensorFlow;
this module is the layout contract between the variables in such a checkpoint (or a
``{variable_name: ndarray}`` dump of one) and the modules here, in both directions
(``restore`` / ``save`` mirror ``GAN.restore(ckpt)`` / the ``saver.save`` call):

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
  """[(tf_variable_name, torch_parameter, kind)] with kind in {'conv_w', 'fc_w', 'bias'}.  ``gan``: a ``GAN``
  (generator + critic + value network) or an ``Agent`` alone (the generator's variables: all that
  ``evaluate.py`` needs from a checkpoint)."""
  out = []

  def conv(prefix, convs):
    for name, m in zip(_conv_names(prefix, len(convs)), convs):
      out.append((name + '/weights', m.weight, 'conv_w'))
      out.append((name + '/biases', m.bias, 'bias'))

  def fc(name, m):
    out.append((name + '/weights', m.weight, 'fc_w'))
    out.append((name + '/biases', m.bias, 'bias'))

  g = getattr(gan, 'generator', gan)
  conv('generator/', g.filter_features.convs)
  for j, f in enumerate(g.filters):
    fc('generator/filter_%d/fc1' % j, f.fc1)
    fc('generator/filter_%d/fc2' % j, f.fc2)
  conv('generator/action_selection/', g.selector_features.convs)
  fc('generator/action_selection/selector_fc1', g.selector_fc1)
  fc('generator/action_selection/selector_fc2', g.selector_fc2)
  if g is gan:
    return out
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


def checkpoint_prefix(model_dir, ckpt):
  """``net.py:406-407``: ``os.path.join(self.dir, "model.ckpt-%s" % ckpt)``."""
  import os
  return os.path.join(model_dir, 'model.ckpt-%s' % ckpt)


def restore(gan, model_dir, ckpt=20000, strict=True):
  """``GAN.restore(ckpt)`` (``net.py:405-407``; ``evaluate.py:28`` passes 20000) from the TF-1 checkpoint
  ``<model_dir>/model.ckpt-<ckpt>.{index,data-*}``.  Whatever else the checkpoint holds (Adam slots, the
  critic's moving averages, step counters) is ignored; returns the names that were missing."""
  from . import tf_bundle
  names = [n for n, _p, _k in tf_name_map(gan)]
  return load_tf_dict(gan, tf_bundle.read_bundle(checkpoint_prefix(model_dir, ckpt), names=names), strict)


def save(gan, model_dir, iteration):
  """The ``saver.save(sess, <dir>/model.ckpt, global_step=iter)`` of ``net.py:380-384``: writes
  ``model.ckpt-<iteration>.index / .data-00000-of-00001`` and the ``checkpoint`` state file TF keeps next to them.
  Only the TRAINABLE variables are written.  The reference's own ``GAN.restore`` builds ``tf.train.Saver(max_to_keep=1)``
  over ALL global variables (Adam slots, beta powers, the counter_g / v / c step variables, the EMA shadow of
  c_average): restoring this bundle with that saver fails with NotFoundError.  A consumer restores it with
  ``tf.train.Saver(var_list=tf.trainable_variables())`` (what ``evaluate.py`` needs) and initialises the rest itself.
  The reference names its files with ``global_step=iter + 1`` (net.py:380-384); pass that as ``iteration`` for
  byte-compatible file names."""
  import os
  from . import tf_bundle
  prefix = checkpoint_prefix(model_dir, iteration)
  tf_bundle.write_bundle(prefix, export_tf_dict(gan))
  base = os.path.basename(prefix)
  with open(os.path.join(model_dir, 'checkpoint'), 'w') as f:
    f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
  return prefix
