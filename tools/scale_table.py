"""Markdown table of the scaling runs tools/scale_all.sh left in a directory (one bench.py JSON line per file)."""
import glob
import json
import os
import sys


def load(path):
  try:
    lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
    return json.loads(lines[-1]) if lines else None
  except (OSError, ValueError):
    return None


def main(d):
  rows = {}
  for f in sorted(glob.glob(os.path.join(d, '*.json'))):
    name = os.path.basename(f)[:-5]
    kind, n = name.rsplit('_', 1)
    rec = load(f)
    if rec is not None:
      rows.setdefault(kind, {})[int(n)] = rec
  print('| workload | scaling | N | value | unit | ms / step | efficiency vs N = 1 |')
  print('|---|---|---|---|---|---|---|')
  for kind in sorted(rows):
    base = rows[kind].get(1)
    for n in sorted(rows[kind]):
      r = rows[kind][n]
      eff = ''
      if base and base['value'] > 0 and not kind.startswith('allreduce'):
        # `value` is the whole-job aggregate for both scalings (weak: N x the work; strong: the same work, faster)
        eff = '%.2f' % (r['value'] / (n * base['value']))
      print('| %s | %s | %d | %.1f | %s | %.3f | %s |' % (kind, r.get('scaling', ''), n, r['value'], r['unit'],
                                                          r['ms_per_step'], eff))


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else '.')
