"""Median duration per convolution kernel and shape from a rocprofv3 kernel trace of tools/r06/conv_cases.py."""
import collections
import csv
import re
import sys

LAY = [(6, 32, 64), (32, 64, 32), (64, 128, 16), (128, 256, 8), (14, 32, 64), (17, 32, 64)]


def short(n):
  m = re.search(r'expo::(conv_\w+?)(<[^>]*>)?\(', n)
  return (m.group(1) + (m.group(2) or '')) if m else None


def main(path, reps=10):
  rows = [r for r in csv.DictReader(open(path)) if 'expo::conv' in r['Kernel_Name']]
  seq = [(short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']),
          int(r['Workgroup_Size_X']), int(r['VGPR_Count']), int(r['LDS_Block_Size'])) for r in rows]
  i = 0
  flops = lambda n, cin, cout, h: 2.0 * n * (h // 2)**2 * cout * 16 * cin
  for n in (64, 192):
    for cin, cout, h in LAY:
      stats, cnt = collections.OrderedDict(), 0
      while i < len(seq):
        nm = seq[i][0]
        if nm.startswith('conv_fwd'):
          cnt += 1
          if cnt > reps:
            break
        stats.setdefault(nm, []).append(seq[i])
        i += 1
      f = flops(n, cin, cout, h)
      print('n=%3d cin=%3d cout=%3d h=%2d  %.2f GF' % (n, cin, cout, h, f / 1e9))
      for nm, v in stats.items():
        us = sorted(x[1] for x in v)[len(v) // 2]
        print('    %-44s %6.1f us  %5.1f TF  grid %6d x %4d  vgpr %3d lds %6d' % (nm, us, f / us / 1e6, v[0][2] // v[0][3], v[0][3], v[0][4], v[0][5]))


if __name__ == '__main__':
  main(sys.argv[1])
