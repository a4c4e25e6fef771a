"""COMPOSITION PIN -- TF primitive semantics ASSUMED.

The bodies of the reference's TensorFlow methods, EXECUTED in the build container (never on the GPU box: /root/reference
does not exist there): ``process`` and ``filter_param_regressor`` of all eight filter classes and ``LevelFilter``
(filters.py:177-182, 200-206, 224-238, 256-273, 306-322, 411-419, 435-440, 481-498, 456-464), the helpers they call
(util.py:225-229 lrelu, 271-274 rgb2lum, 277-294 tanh01 / tanh_range, 307-308 lerp), ``pdf_sample``
(pdf_sample_layer.py:5-10) and the statements of ``agent_generator`` that turn the selector's logits into the action pdf,
the sampled / arg-max filter id, the one-hot, the surrogate, the new states and the penalty (agent.py:100-123, 208-252).

TensorFlow is not installed, so the method bodies -- cut out of the files by ``ast`` (by line span where the file does not
parse), nothing else of the reference runs -- see a NumPy FACADE as ``tf``: float64 stand-ins for exactly the primitives
they use (the list is written into the fixture).  What this pins: that the build's restatements COMPOSE those
primitives the way the reference does -- operand order, broadcasting axes, which tensor is clamped before which blend,
knot indexing, the 1e-30 / 1e-36 / 1e-37 / 1e-10 guards, exclusive-cumsum sampling.  What it cannot pin: the primitives
themselves (tf.image.rgb_to_hsv / hsv_to_rgb are matplotlib.colors here, which SURVEY.md 8(c) found equal to TF's
published formulas to 4e-16; tf.clip_by_value / maximum / minimum ties; Eigen's reduction order; every gradient).
A stand-in library pins no primitive, so DESIGN.md section 7 still calls the parity of those filters "pinned by
restatement"; this fixture turns "filters.py:415-419 was read correctly" into a test.

The committed fixture holds inputs, outputs, the facade's op list and the sha256 of each source file -- data only.
tests/test_reference_facade.py checks the three CPU oracles and (gpu) the HIP kernels against it.

  python tests/golden/make_reference_facade.py        # rewrites tests/golden/reference_facade.npz
"""
import ast
import contextlib
import hashlib
import math
import os
import re
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def sha256(path):
  return hashlib.sha256(open(path, 'rb').read()).hexdigest()


class TfFacade:
  """NumPy stand-ins for the TensorFlow-1 primitives the cut-out bodies use; every call is recorded."""

  int32, float32 = np.int32, np.float64  # (the facade computes in float64 throughout)

  def __init__(self):
    import matplotlib.colors as mc
    self.used = set()
    facade = self

    class image:  # tf.image
      # matplotlib.colors where it accepts the pixel (every component in [0, 1]: an implementation independent of this
      # repository); TF's published per-pixel formulas (colorspace_op.h, restated in SURVEY.md 8(c)) for the pixels it
      # refuses -- negative channels, which filters.py:485 lets through (`tf.minimum(img, 1.0)` clamps from above only)
      @staticmethod
      def rgb_to_hsv(x):
        x = np.asarray(x, dtype=np.float64)
        inside = ((x >= 0) & (x <= 1)).all(axis=-1)
        out = np.empty_like(x)
        facade.used.add('image.rgb_to_hsv [matplotlib.colors]')
        out[inside] = mc.rgb_to_hsv(x[inside])
        if (~inside).any():
          facade.used.add('image.rgb_to_hsv [TF formula: pixels outside [0, 1]]')
          r, g, b = (x[~inside][:, k] for k in range(3))
          v = np.maximum(np.maximum(r, g), b)
          rng_ = v - np.minimum(np.minimum(r, g), b)
          sat = np.where(v > 0, rng_ / np.where(v > 0, v, 1.0), 0.0)
          norm = 1.0 / (6.0 * np.where(rng_ > 0, rng_, 1.0))
          hue = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
          hue = np.where(rng_ > 0, hue, 0.0)
          hue = np.where(hue < 0, hue + 1.0, hue)
          out[~inside] = np.stack([hue, sat, v], axis=-1)
        return out

      @staticmethod
      def hsv_to_rgb(x):
        x = np.asarray(x, dtype=np.float64)
        inside = ((x >= 0) & (x <= 1)).all(axis=-1)
        out = np.empty_like(x)
        facade.used.add('image.hsv_to_rgb [matplotlib.colors]')
        out[inside] = mc.hsv_to_rgb(x[inside])
        if (~inside).any():
          facade.used.add('image.hsv_to_rgb [TF formula: pixels outside [0, 1]]')
          hh, ss, vv = (x[~inside][:, k] for k in range(3))
          dh = hh * 6.0
          dr = np.clip(np.abs(dh - 3.0) - 1.0, 0.0, 1.0)
          dg = np.clip(2.0 - np.abs(dh - 2.0), 0.0, 1.0)
          db = np.clip(2.0 - np.abs(dh - 4.0), 0.0, 1.0)
          one_minus_s = 1.0 - ss
          out[~inside] = np.stack([(one_minus_s + ss * dr) * vv, (one_minus_s + ss * dg) * vv, (one_minus_s + ss * db) * vv], axis=-1)
        return out

    class nn:  # tf.nn
      @staticmethod
      def softmax(x):
        facade.used.add('nn.softmax')
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e / e.sum(axis=-1, keepdims=True)

    self.image, self.nn = image, nn

  def _u(self, name):
    self.used.add(name)

  def exp(self, x): self._u('exp'); return np.exp(x)
  def log(self, x): self._u('log'); return np.log(x)
  def pow(self, x, y): self._u('pow'); return np.power(x, y)
  def maximum(self, x, y): self._u('maximum'); return np.maximum(x, y)
  def minimum(self, x, y): self._u('minimum'); return np.minimum(x, y)
  def clip_by_value(self, x, lo, hi): self._u('clip_by_value'); return np.clip(x, lo, hi)
  def cos(self, x): self._u('cos'); return np.cos(x)
  def abs(self, x): self._u('abs'); return np.abs(x)
  def tanh(self, x): self._u('tanh'); return np.tanh(x)
  def sigmoid(self, x): self._u('sigmoid'); return 1.0 / (1.0 + np.exp(-x))
  def reshape(self, x, shape): self._u('reshape'); return np.reshape(x, shape)
  def concat(self, values, axis): self._u('concat'); return np.concatenate(values, axis=axis)
  def less(self, x, y): self._u('less'); return np.less(x, y)
  def cast(self, x, dtype): self._u('cast'); return np.asarray(x).astype(dtype)
  def argmax(self, x, axis): self._u('argmax'); return np.argmax(x, axis=axis)

  def one_hot(self, ids, depth, dtype):
    self._u('one_hot')  # tf.one_hot: an index outside [0, depth) gives an all-zero row
    return (np.asarray(ids)[:, None] == np.arange(depth)[None, :]).astype(dtype)

  def reduce_sum(self, x, axis=None, keep_dims=False):
    self._u('reduce_sum')
    return np.sum(x, axis=axis, keepdims=keep_dims)

  def reduce_mean(self, x, axis=None, keep_dims=False):
    self._u('reduce_mean')
    return np.mean(x, axis=axis, keepdims=keep_dims)

  def cumsum(self, x, axis, exclusive=False):
    self._u('cumsum')
    c = np.cumsum(x, axis=axis)
    if exclusive:  # a SHIFTED inclusive scan, accumulated left to right (not cumsum - x)
      c = np.concatenate([np.zeros_like(np.take(c, [0], axis=axis)), np.delete(c, -1, axis=axis)], axis=axis)
    return c

  @staticmethod
  def variable_scope(name):
    return contextlib.nullcontext()


def cut_span(path, name):
  """A top-level ``def name`` of a file that does not parse as a whole (util.py:658): its ``def`` line to the next
  top-level statement."""
  lines = open(path).read().split('\n')
  start = next(i for i, l in enumerate(lines) if re.match(r'def %s\(' % re.escape(name), l))
  end = start + 1
  while end < len(lines) and (lines[end].strip() == '' or lines[end][0] in ' \t'):
    end += 1
  return ast.parse('\n'.join(lines[start:end]), filename='%s:%d' % (path, start + 1))


def methods_of(tree, cls_name, names):
  cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
  out = []
  for fn in cls.body:
    if isinstance(fn, ast.FunctionDef) and fn.name in names:
      fn = ast.FunctionDef(name='%s__%s' % (cls_name, fn.name), args=fn.args, body=fn.body, decorator_list=[],
                           returns=None, type_comment=None, lineno=fn.lineno, col_offset=0)
      out.append(fn)
  assert len(out) == len(names), (cls_name, [f.name for f in out])
  return out


def no_prints(stmts):
  keep = []
  for s in stmts:
    if isinstance(s, ast.Expr) and isinstance(s.value, ast.Call) and getattr(s.value.func, 'id', None) == 'print':
      continue
    keep.append(s)
  return keep


FILTERS = [('ExposureFilter', 1), ('GammaFilter', 1), ('ImprovedWhiteBalanceFilter', 3), ('SaturationPlusFilter', 1),
           ('ToneFilter', 8), ('ContrastFilter', 1), ('WNBFilter', 1), ('ColorFilter', 24), ('LevelFilter', 2)]


def main():
  rng = np.random.default_rng(20261001)
  tf = TfFacade()
  out, prov = {}, []

  # ---- util.py helpers (the file has a Python-2 print at line 658: cut by span) -----------------------------------
  u_path = os.path.join(REF, 'util.py')
  prov.append(('util.py', sha256(u_path)))
  ns = {'tf': tf, 'np': np, 'math': math}
  for name in ('lrelu', 'rgb2lum', 'tanh01', 'tanh_range', 'lerp'):
    exec(compile(cut_span(u_path, name), '<reference util.%s>' % name, 'exec'), ns)
  consts = {}
  for line in open(u_path).read().split('\n'):
    m = re.match(r'(STATE_[A-Z_]+) = (\d+)\s*$', line)
    if m:
      consts[m.group(1)] = int(m.group(2))
  assert consts == {'STATE_REWARD_DIM': 0, 'STATE_STOPPED_DIM': 1, 'STATE_STEP_DIM': 2, 'STATE_DROPOUT_BEGIN': 3}, consts
  ns.update(consts)
  xs = rng.standard_normal(64) * 2
  xs[:3] = [0.0, -0.0, 1e-30]
  out['util_lrelu_x'], out['util_lrelu_y'] = xs, ns['lrelu'](xs)

  # ---- filters.py: process / filter_param_regressor of every filter class --------------------------------------------
  f_path = os.path.join(REF, 'filters.py')
  prov.append(('filters.py', sha256(f_path)))
  ftree = ast.parse(open(f_path).read(), filename=f_path)
  body = []
  for cls, _p in FILTERS:
    body += methods_of(ftree, cls, ['filter_param_regressor', 'process'])
  mod = ast.Module(body=body, type_ignores=[])
  ast.fix_missing_locations(mod)
  ns['print'] = lambda *a, **k: None  # (ImprovedWhiteBalanceFilter.filter_param_regressor prints a shape)
  exec(compile(mod, '<reference filters.py methods>', 'exec'), ns)
  cfg = types.SimpleNamespace(exposure_range=3.5, gamma_range=3, curve_steps=8, color_curve_range=(0.90, 1.10),
                              tone_curve_range=(0.5, 2))  # config_example.py:27-33
  self_ = types.SimpleNamespace(cfg=cfg, channels=3, curve_steps=8)

  n, h, w = 6, 12, 10
  x = rng.random((n, h, w, 3))**2.2 * 1.25  # linear-RAW-like, a few per cent above 1
  x[0, 0, :8, :] = (np.arange(8) / 8.0)[:, None]  # exactly on the curve knots
  x[0, 1, :4, :] = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.001, 0.001, 0.001], [1.0, 0.0, 0.0]])  # lum 0 / 1, gamma's floor
  x[1, 0, :4, :] = np.array([[-0.05, 0.2, 0.3], [0.4, -0.3, 0.1], [1.5, 0.2, 0.9], [2.0, 2.0, 2.0]])  # negative / > 1 channels
  x[2, 0, :3, :] = np.array([[0.5, 0.5, 0.5], [0.3, 0.3, 0.7], [0.7, 0.7, 0.2]])  # grey (s = 0), two-way channel ties
  out['x'] = x
  for cls, p in FILTERS:
    f = rng.standard_normal((n, p)) * 1.5
    f[0] = 0.0  # the regressors' centre
    params = ns['%s__filter_param_regressor' % cls](self_, f)
    y = ns['%s__process' % cls](self_, x, params)
    assert y.shape == x.shape and np.isfinite(y).all(), cls
    out['%s_features' % cls], out['%s_params' % cls], out['%s_y' % cls] = f, np.asarray(params), y

  # ---- pdf_sample_layer.py:5-10 ------------------------------------------------------------------------------------------
  p_path = os.path.join(REF, 'pdf_sample_layer.py')
  prov.append(('pdf_sample_layer.py', sha256(p_path)))
  ptree = ast.parse(open(p_path).read(), filename=p_path)
  fn = next(nd for nd in ptree.body if isinstance(nd, ast.FunctionDef) and nd.name == 'pdf_sample')
  exec(compile(ast.Module(body=[fn], type_ignores=[]), '<reference pdf_sample>', 'exec'), ns)
  pdf = rng.random((40, 8)) + 0.01
  noise = rng.random((40, 1))
  noise[:4, 0] = [0.0, 1.0, 0.999999, 1e-9]  # 0 -> id -1 (nothing is < 0), 1 -> the last id
  pdf[4] = [0, 0, 1, 0, 0, 0, 0, 0]
  out['pdf_sample_pdf'], out['pdf_sample_noise'] = pdf, noise
  out['pdf_sample_ids'] = ns['pdf_sample'](pdf, noise).astype(np.int32)

  # ---- agent.py:100-123 and 208-252: the statements of agent_generator between the selector's logits and the penalty --
  a_path = os.path.join(REF, 'agent.py')
  prov.append(('agent.py', sha256(a_path)))
  atree = ast.parse(open(a_path).read(), filename=a_path)
  gen = next(nd for nd in atree.body if isinstance(nd, ast.FunctionDef) and nd.name == 'agent_generator')

  def statements(lo, hi):
    found = []
    for node in ast.walk(gen):
      if isinstance(node, ast.stmt) and not isinstance(node, (ast.With, ast.If, ast.For, ast.FunctionDef)) and \
          lo <= node.lineno <= hi:
        found.append(node)
    found.sort(key=lambda s: s.lineno)
    return no_prints(found)

  sel = statements(100, 123)
  assert ast.unparse(sel[0]).startswith('pdf = tf.nn.softmax(pdf) + 1e-37') and ast.unparse(sel[-1]).startswith('surrogate = '), \
      [ast.unparse(s)[:50] for s in sel]
  upd = [s for s in statements(208, 252) if not isinstance(s, ast.Assert) and 'regular_filter_start = 0' != ast.unparse(s)]
  assert ast.unparse(upd[0]).startswith('new_states = [None') and ast.unparse(upd[-1]).startswith('penalty = '), \
      [ast.unparse(s)[:50] for s in upd]
  acfg = types.SimpleNamespace(exploration=0.05, test_steps=5, early_stop_penalty=1.0, filter_usage_penalty=1.0,
                               exploration_penalty=0.05, clamp=False)  # config_example.py:48-69
  m = 24
  logits = rng.standard_normal((m, 8)) * 2
  states = np.zeros((m, 11))
  states[:, 2] = rng.integers(0, 6, m)  # the step counter, incl. the last step (4 -> 5 = test_steps)
  states[:, 3:] = (rng.random((m, 8)) < 0.3).astype(np.float64)  # filters already used
  sel_noise = rng.random((m, 1))
  sel_noise[0, 0] = 0.0
  net = rng.random((m, 6, 5, 3)) * 1.3
  for is_train in (1, 0):
    ans = dict(ns, cfg=acfg, filters=[None] * 8, pdf=logits.copy(), selection_noise=sel_noise, is_train=is_train,
               states=states, progress=0.25, net=net, regular_filter_start=0)
    exec(compile(ast.Module(body=sel + upd, type_ignores=[]), '<reference agent.py:100-123, 208-252>', 'exec'), ans)
    tag = 'train' if is_train else 'eval'
    for key in ('pdf', 'entropy', 'selected_filter_id', 'filter_one_hot', 'surrogate', 'new_states', 'penalty'):
      out['agent_%s_%s' % (tag, key)] = np.asarray(ans[key])
  out['agent_logits'], out['agent_states'], out['agent_noise'], out['agent_net'] = logits, states, sel_noise, net
  out['agent_progress'] = np.array(0.25)

  out['facade_ops'] = np.array(sorted(tf.used))
  out['provenance'] = np.array(['%s sha256=%s' % p for p in prov])
  path = os.path.join(HERE, 'reference_facade.npz')
  np.savez_compressed(path, **out)
  print('wrote %s (%d bytes)' % (path, os.path.getsize(path)))
  print('facade ops:', ', '.join(sorted(tf.used)))


if __name__ == '__main__':
  if not os.path.isdir(REF):
    sys.exit('needs /root/reference (the build container)')
  main()
