"""The reference's configuration, produced by EXECUTING its own config files in the build container (never on the GPU
box).  ``config_example.py`` / ``config_sintel.py`` are plain Python behind six imports (TensorFlow-side modules); the
import statements are dropped and every name they would have bound -- the filter classes of ``filters.py`` (read from its
AST), ``agent_generator``, ``critic``, the data providers -- is bound to an empty class of the same name, which is all the
files need at import time (they only LIST those names); ``Dict`` is cut out of ``util.py``.  The fixture
(tests/golden/reference_config.json) holds every plain value of ``cfg``, the filter order by class name, the two
learning-rate callbacks evaluated at a set of iterations, and the Adam constants read from the optimizer lambda's AST.
tests/test_reference_config.py holds ``exposure_amd.config.make_cfg()`` to it.

  python tests/golden/make_reference_config.py
"""
import ast
import hashlib
import json
import os
import re
import sys

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ITERS = [0, 1, 10, 100, 499, 500, 3333, 6667, 10000, 19999, 20000]


def cut_span(path, pattern):
  lines = open(path).read().split('\n')
  start = next(i for i, l in enumerate(lines) if re.match(pattern, l))
  end = start + 1
  while end < len(lines) and (lines[end].strip() == '' or lines[end][0] in ' \t'):
    end += 1
  return ast.parse('\n'.join(lines[start:end]), filename='%s:%d' % (path, start + 1))


def run_config(name):
  path = os.path.join(REF, name)
  tree = ast.parse(open(path).read(), filename=path)
  ns = {}
  exec(compile(cut_span(os.path.join(REF, 'util.py'), r'class Dict\(dict\):'), '<reference util.Dict>', 'exec'), ns)
  names = []
  for node in tree.body:
    if isinstance(node, ast.ImportFrom):
      if node.module == 'filters':  # from filters import *
        ftree = ast.parse(open(os.path.join(REF, 'filters.py')).read())
        names += [n.name for n in ftree.body if isinstance(n, ast.ClassDef)]
      elif node.module != 'util':
        names += [a.name for a in node.names]
  for n in names:
    ns[n] = type(n, (), {})
  body = [n for n in tree.body if not isinstance(n, (ast.Import, ast.ImportFrom))]
  exec(compile(ast.Module(body=body, type_ignores=[]), '<reference %s>' % name, 'exec'), ns)
  cfg = ns['cfg']
  plain, other = {}, {}
  for k, v in cfg.items():
    if isinstance(v, (bool, int, float, str)):
      plain[k] = v
    elif isinstance(v, tuple) and all(isinstance(x, (int, float)) for x in v):
      plain[k] = list(v)
    elif k == 'filters':
      other[k] = [c.__name__ for c in v]
    elif isinstance(v, type):
      other[k] = v.__name__
    else:
      other[k] = '<callable>'
  adam = {}
  for node in ast.walk(tree):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'AdamOptimizer':
      adam = {kw.arg: kw.value.value for kw in node.keywords if isinstance(kw.value, ast.Constant)}
  return {'plain': plain, 'other': other, 'adam': adam, 'lr_g': [cfg.lr_g(t) for t in ITERS],
          'lr_c': [cfg.lr_c(t) for t in ITERS], 'sha256': hashlib.sha256(open(path, 'rb').read()).hexdigest()}


def main():
  out = {'iterations': ITERS, 'config_example.py': run_config('config_example.py'),
         'config_sintel.py': run_config('config_sintel.py')}
  path = os.path.join(HERE, 'reference_config.json')
  json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
  print('wrote %s: %d plain fields, filters %s' % (path, len(out['config_example.py']['plain']),
                                                   out['config_example.py']['other']['filters']))


if __name__ == '__main__':
  if not os.path.isdir(REF):
    sys.exit('needs /root/reference (the build container)')
  main()
