"""Instruction histogram of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: python tools/isa_hist.py dev.s <substring of the mangled kernel name>"""
import collections
import re
import sys


def main(path, pat):
  lines = open(path).read().split('\n')
  start = None
  for i, l in enumerate(lines):
    if re.match(r'^[A-Za-z_][^\s]*:', l) and pat in l.split(':')[0]:
      start = i
      break
  if start is None:
    raise SystemExit('kernel not found')
  ops = collections.Counter()
  n = 0
  for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end') or '.end_amdhsa_kernel' in l:
      break
    m = re.match(r'\s+([a-z][a-z_0-9]+)\s', l + ' ')
    if m and not l.strip().startswith(('.', ';')):
      ops[m.group(1)] += 1
      n += 1
  cat = collections.Counter()
  for k, v in ops.items():
    key = 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else \
        'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'
    cat[key] += v
  print(lines[start][:110], 'instructions:', n)
  print(dict(cat))
  for k, v in ops.most_common(40):
    print('  %-28s %d' % (k, v))


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2])
