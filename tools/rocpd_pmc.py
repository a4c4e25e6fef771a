"""Average PMC counter value per kernel from a rocprofv3 rocpd database (counters_collection view).
usage: python tools/rocpd_pmc.py results.db   -> CSV: kernel, counter, launches, mean, min, max"""
import re
import sqlite3
import sys


def short(name):
  m = re.search(r'filter_(fwd|bwd)_kernel<expo::(\w+?)(F|F<(\d)>),', name)
  if m:
    return 'filter_%s_kernel<%s%s>' % (m.group(1), m.group(2), m.group(4) or '')
  name = re.sub(r'\(.*', '', name)
  return name[:160]


def table(path):
  c = sqlite3.connect(path)
  rows = c.execute('select kernel_name, counter_name, count(*), avg(value), min(value), max(value) '
                   'from counters_collection group by kernel_name, counter_name order by avg(value) desc').fetchall()
  return [(short(k), cn, n, a, mn, mx) for k, cn, n, a, mn, mx in rows]


if __name__ == '__main__':
  print('"Kernel","Counter","Launches","Mean","Min","Max"')
  for r in table(sys.argv[1]):
    print('"%s","%s",%d,%.3f,%.3f,%.3f' % r)
