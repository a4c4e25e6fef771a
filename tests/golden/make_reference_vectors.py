"""Golden vectors produced by RUNNING THE REFERENCE'S OWN CODE in the build container (never on the GPU box:
/root/reference does not exist there) -- the part of the reference that is plain NumPy and can execute without
TensorFlow / cv2 / PyQt5:

  user_study_ui/filters.py   ExposureFilter.apply, GammaFilter.apply, WBFilter.apply (the authors' NumPy statement of
                             three of the eight filters, the same formulas as filters.py:181-182, 205-206, 227-238),
                             rgb2lum, lerp (== util.py:271-274, 307-308)
  util.py                    linearize_ProPhotoRGB (495-501), lerp (307-308)
  histogram_intersection.py  hist_intersection, calc_hist (11-12, 23-25) and the luminance half of
                             get_statistics (15-20; its saturation needs cv2.cvtColor and is NOT covered)
  filters.py                 the NumPy statements inside Filter.get_mask / VignetFilter.get_mask that build the constant
                             coordinate grid of the spatial masks (124-133, 371-380)

Neither module can be IMPORTED here (module-level ``import cv2`` / ``PyQt5`` / ``tensorflow``; util.py:658 is a syntax
error on Python >= 3.7), so the named definitions are cut out of the files -- by ``ast`` where the file parses, by
their ``def`` line span where it does not -- and executed in a namespace that holds numpy and math only.  Nothing else
of the reference runs, nothing of it is written to the repository: the committed file holds inputs, outputs and the
sha256 of each source file the outputs came from.  tests/test_reference_vectors.py checks the oracle (CPU) and the HIP
kernels (GPU) against it.

What this pins and what it does not: the forward arithmetic of Exposure / Gamma (x >= 0.001) /
WhiteBalance incl. its luminance normalisation, rgb2lum's weights, lerp, the ProPhoto linearisation, the metric's
histogram arithmetic.  It does NOT pin the TF-side conventions (HSV ops, clip / maximum ties, Eigen's row-sum order):
no executable statement of those exists in /root/reference.

  python tests/golden/make_reference_vectors.py        # rewrites tests/golden/reference_numpy.npz
"""
import ast
import hashlib
import math
import os
import re
import sys

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def sha256(path):
  return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def cut_by_ast(path, names):
  """The top-level definitions ``names`` of a file that parses, as one module (their source order)."""
  tree = ast.parse(open(path).read(), filename=path)
  keep = [node for node in tree.body
          if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names]
  missing = set(names) - {n.name for n in keep}
  assert not missing, missing
  return ast.Module(body=keep, type_ignores=[])


def cut_by_span(path, name):
  """A top-level ``def name`` of a file that does NOT parse as a whole: from its ``def`` line to the next top-level
  statement."""
  lines = open(path).read().split('\n')
  start = next(i for i, l in enumerate(lines) if re.match(r'def %s\(' % re.escape(name), l))
  end = start + 1
  while end < len(lines) and (lines[end].strip() == '' or lines[end][0] in ' \t'):
    end += 1
  return ast.parse('\n'.join(lines[start:end]), filename='%s:%d' % (path, start + 1))


def run(module, namespace):
  exec(compile(module, '<reference>', 'exec'), namespace)
  return namespace


def main():
  rng = np.random.default_rng(20260927)
  out = {}
  prov = []

  # ---- user_study_ui/filters.py: the authors' NumPy filters -------------------------------------------------------
  ui_path = os.path.join(REF, 'user_study_ui', 'filters.py')
  ui = run(cut_by_ast(ui_path, ['Filter', 'ExposureFilter', 'GammaFilter', 'WBFilter', 'rgb2lum', 'lerp']),
           {'np': np, 'math': math})
  prov.append(('user_study_ui/filters.py', sha256(ui_path)))
  # images H x W x 3 as the UI holds them (float32, linear, mostly dark, some values above 1 and a few negative ones)
  n, h, w = 6, 24, 20
  img = (rng.random((n, h, w, 3))**2.2 * 1.3).astype(np.float32)
  img[0, :2] *= -0.25  # negative pixels (Exposure and WB are linear: defined there too)
  out['ui_images'] = img
  # slider positions 0..100 -> the filters' own get_transformed_parameter
  sl_e = np.array([50, 62, 41, 70, 20, 55])  # 'linear' -5..5 -> EV -0.9 .. +2 (inside cfg.exposure_range 3.5)
  sl_g = np.array([50, 30, 70, 62, 40, 76])  # 'log' 8 .. 1/8 -> gamma in [1/3, 3] (cfg.gamma_range)
  sl_w = np.array([[50, 50], [10, 90], [99, 1], [35, 60], [80, 80], [5, 45]])  # temperature, tint in (-0.5, 0.5)
  ev, gm, temp_tint, y_e, y_g, y_w = [], [], [], [], [], []
  for i in range(n):
    f = ui['ExposureFilter']()
    f.parameters[0] = int(sl_e[i])
    ev.append(f.get_transformed_parameter(0))
    y_e.append(f.apply(img[i].copy()))
    f = ui['GammaFilter']()
    f.parameters[0] = int(sl_g[i])
    gm.append(f.get_transformed_parameter(0))
    pos = np.maximum(img[i], np.float32(0.001))  # the TF path clamps below 0.001, the UI does not: compare above it
    y_g.append(f.apply(pos.copy()))
    f = ui['WBFilter']()
    f.parameters[0], f.parameters[1] = int(sl_w[i, 0]), int(sl_w[i, 1])
    temp_tint.append([f.get_transformed_parameter(0), f.get_transformed_parameter(1)])
    y_w.append(f.apply(img[i].copy()))
  out['ui_exposure_ev'] = np.array(ev, np.float64)
  out['ui_exposure_y'] = np.stack(y_e)
  out['ui_gamma_g'] = np.array(gm, np.float64)
  out['ui_gamma_x'] = np.maximum(img, np.float32(0.001))
  out['ui_gamma_y'] = np.stack(y_g)
  out['ui_wb_temp_tint'] = np.array(temp_tint, np.float64)
  out['ui_wb_y'] = np.stack(y_w)
  out['ui_rgb2lum'] = np.stack([ui['rgb2lum'](img[i]) for i in range(n)])
  alpha = rng.random((n, 1, 1, 1)).astype(np.float32)
  other = rng.random(img.shape).astype(np.float32)
  out['ui_lerp_b'], out['ui_lerp_alpha'] = other, alpha
  out['ui_lerp_y'] = ui['lerp'](img, other, alpha)

  # ---- util.py (does not parse as a whole: cut by line span) -------------------------------------------------------
  util_path = os.path.join(REF, 'util.py')
  ns = {'np': np, 'math': math}
  for name in ('linearize_ProPhotoRGB', 'lerp'):
    run(cut_by_span(util_path, name), ns)
  prov.append(('util.py', sha256(util_path)))
  pp = rng.random((3, 16, 12, 3))
  out['util_pp_rgb'] = pp
  out['util_linearized'] = ns['linearize_ProPhotoRGB'](pp)
  out['util_delinearized'] = ns['linearize_ProPhotoRGB'](pp, reverse=True)
  out['util_lerp_y'] = ns['lerp'](img.astype(np.float64), other.astype(np.float64), alpha.astype(np.float64))

  # ---- histogram_intersection.py -----------------------------------------------------------------------------------
  hi_path = os.path.join(REF, 'histogram_intersection.py')
  hi = run(cut_by_ast(hi_path, ['hist_intersection', 'calc_hist']), {'np': np})
  prov.append(('histogram_intersection.py', sha256(hi_path)))
  # get_statistics calls cv2.cvtColor for the saturation: its luminance statements (lines 16, 18, 20) are executed
  # from the function's own source with the cv2 line and the saturation entry left out
  src = open(hi_path).read().split('\n')
  start = next(i for i, l in enumerate(src) if l.startswith('def get_statistics'))
  body = [l for l in src[start:start + 6] if 'cv2' not in l and 'sat = ' not in l]
  body = [l.replace('return [lum.mean(), lum.std() * 2, sat]', 'return [lum.mean(), lum.std() * 2]') for l in body]
  assert len(body) == 4 and body[-1].strip() == 'return [lum.mean(), lum.std() * 2]', body
  run(ast.parse('\n'.join(body)), hi)
  m = 40
  imgs_a = (rng.random((m, 12, 12, 3))**1.5 * 1.2 - 0.05).astype(np.float32)  # some values outside [0, 1]: clipped
  imgs_b = (rng.random((m, 12, 12, 3))**0.8).astype(np.float32)
  out['hi_images_a'], out['hi_images_b'] = imgs_a, imgs_b
  st_a = np.array([hi['get_statistics'](im) for im in imgs_a], np.float64)
  st_b = np.array([hi['get_statistics'](im) for im in imgs_b], np.float64)
  out['hi_stats_a'], out['hi_stats_b'] = st_a, st_b
  hists_a = np.array([hi['calc_hist'](st_a[:, k], 32, (0.0, 1.0)) for k in range(2)])
  hists_b = np.array([hi['calc_hist'](st_b[:, k], 32, (0.0, 1.0)) for k in range(2)])
  out['hi_hists_a'], out['hi_hists_b'] = hists_a, hists_b
  out['hi_intersections'] = np.array([hi['hist_intersection'](hists_a[k], hists_b[k]) for k in range(2)])
  # np.histogram's edge rules through the reference's calc_hist: values on bin edges, on both range ends, outside
  edge = np.array([0.0, 1.0, 0.5, 0.03125, 0.96875, -0.1, 1.1, 0.999999, 0.25, 0.25])
  out['hi_edge_values'] = edge
  out['hi_edge_hist'] = hi['calc_hist'](edge, 32, (0.0, 1.0))

  # ---- filters.py: the NumPy statements of Filter.get_mask / VignetFilter.get_mask that build the coordinate grid ----
  # (the methods themselves are TensorFlow graph code; the grid is a constant they compute with plain NumPy loops:
  # filters.py:124-133 and 371-380.  Those statements -- the np.zeros, shorter_edge and the nested for loop -- are taken
  # from the method's AST and executed with `size` given.)
  f_path = os.path.join(REF, 'filters.py')
  ftree = ast.parse(open(f_path).read(), filename=f_path)
  prov.append(('filters.py', sha256(f_path)))

  def grid_statements(cls_name):
    cls = next(n for n in ftree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'get_mask')
    stmts = []
    for node in ast.walk(fn):
      if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
        name = node.targets[0].id
        src = ast.unparse(node.value)
        if (name == 'grid' and src.startswith('np.zeros')) or name == 'shorter_edge':
          stmts.append(node)
      elif isinstance(node, ast.For) and isinstance(node.target, ast.Name) and node.target.id == 'i':
        stmts.append(node)
    stmts.sort(key=lambda n: n.lineno)
    assert [type(n).__name__ for n in stmts] == ['Assign', 'Assign', 'For'], [ast.unparse(n)[:40] for n in stmts]
    return ast.Module(body=stmts, type_ignores=[])

  sizes = [(64, 64), (5, 9), (9, 7), (16, 24), (2, 2)]
  out['mask_grid_sizes'] = np.array(sizes)
  for cls_name, key in (('Filter', 'mask_grid'), ('VignetFilter', 'vignet_grid')):
    code = compile(grid_statements(cls_name), '<reference filters.%s.get_mask grid>' % cls_name, 'exec')
    for (gh, gw) in sizes:
      gns = {'np': np, 'size': [gh, gw]}
      exec(code, gns)
      assert gns['grid'].dtype == np.float32 and gns['grid'].shape == (1, gh, gw, 2)
      out['%s_%dx%d' % (key, gh, gw)] = gns['grid']

  out['provenance'] = np.array(['%s sha256=%s' % p for p in prov])
  path = os.path.join(HERE, 'reference_numpy.npz')
  np.savez_compressed(path, **out)
  print('wrote %s (%d bytes): %s' % (path, os.path.getsize(path), ', '.join(sorted(out))))


if __name__ == '__main__':
  if not os.path.isdir(REF):
    sys.exit('needs /root/reference (the build container)')
  main()
