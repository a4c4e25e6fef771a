"""Worst |err| / A per kind of reduced quantity from the jsonl the gpu suite writes under EXPO_RECORD_PARAM_ERR
(tests/_tol.py): the table of DESIGN.md section 7.   usage: python tools/r04/param_err_table.py errors.jsonl"""
import collections
import json
import re
import sys


def kind(what):
  w = what
  for pat, name in ((r'^dparams of step (\d) \((\w) (\w+)\)', None), ):
    m = re.match(pat, w)
    if m:
      f = 'E G W S+ T Ct BW C'.split()[int(m.group(1))]
      return 'every value, %sx512x512 %s chain: %s' % ({'C': '64', 'B': '16'}[m.group(2)], m.group(3), 'curves T / C' if f in ('T', 'C') else 'E G W S+ Ct BW')
  for frag, name in (('chain dp step', 'chain 64x64x64 per step'), ('masked dmask', 'masked apply: mask parameters'),
                     ('masked dparams', 'masked apply: filter parameters'), ('vignet', 'VignetFilter mask parameters'),
                     ('dispatch', 'per-image dispatch (agent step)'), ('fused', 'one-pass backward (benchmark construct)'),
                     ('golden', 'golden vectors'), ('stats J v', 'critic statistics J v'), ('autograd J v', 'critic statistics J v'),
                     ('generic curve', 'generic curve kernels (cfg.curve_steps != 8)'), ('launch ', 'back-to-back launches (E, G, W)'),
                     ('level', 'LevelFilter'), ('pair', 'low + high resolution pair node'), ('tone dparams', 'generic curve kernels (cfg.curve_steps != 8)')):
    if frag in w:
      return name
  return 'per-filter kernels, assorted shapes'


def main(path):
  rows = [json.loads(l) for l in open(path)]
  by = collections.defaultdict(list)
  for r in rows:
    if r['err_over_A'] < 1e3:  # (entries whose A is pure float64 noise are covered by the absolute floor)
      by[kind(r['what'])].append(r)
  print('| comparison | calls | worst err / A | worst err / ref | bound |')
  print('|---|---|---|---|---|')
  for k in sorted(by):
    v = by[k]
    print('| %s | %d | %.2e | %.2e | 1e-4 ref + 2e-6 A |' % (k, len(v), max(r['err_over_A'] for r in v), max(r['err_over_ref'] for r in v)))
  print('\n%d comparisons recorded.' % len(rows))


if __name__ == '__main__':
  main(sys.argv[1])
