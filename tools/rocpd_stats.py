"""Per-kernel stats (calls, total, avg, min, max, %) from a rocprofv3 rocpd SQLite database --
the same table `rocprofv3 --stats` prints, for runs whose output format was the default rocpd.
usage: python tools/rocpd_stats.py results.db [> profiles/xxx_kernel_stats.csv]"""
import re
import sqlite3
import sys


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = name.replace('expo::', '').replace('_Float16', 'f16')
  name = re.sub(r'\(.*\)$', '', name)
  name = re.sub(r'^void ', '', name)
  return name


def main(path):
  c = sqlite3.connect(path)
  rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                   'from kernels group by name order by sum(duration) desc').fetchall()
  total = sum(r[2] for r in rows) or 1
  print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
  for n, cnt, tot, avg, mn, mx in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (short(n), cnt, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == '__main__':
  main(sys.argv[1])
