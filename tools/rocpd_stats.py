"""Per-kernel stats (calls, total, avg, min, max, %) from a rocprofv3 rocpd SQLite database --
the same table `rocprofv3 --stats` prints, for runs whose output format was the default rocpd.
usage: python tools/rocpd_stats.py results.db [grid_y=N] [> profiles/xxx_kernel_stats.csv]
grid_y=N[,M..] keeps only launches over N (or M ..) images (the library's grids are (blocks per image, images)): the chain runs its
two half-batches as separate, overlapping launches, whereas the per-kernel roofline is quoted on whole-batch launches
of the same kernels in the same run -- one table per launch shape."""
import re
import sqlite3
import sys


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = name.replace('expo::', '').replace('_Float16', 'f16')
  name = re.sub(r'\(.*\)$', '', name)
  name = re.sub(r'^void ', '', name)
  return name


def main(path, grid_y=None):
  c = sqlite3.connect(path)
  where = '' if grid_y is None else ' where grid_y in (%s)' % ','.join(str(int(v)) for v in str(grid_y).split(','))
  rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                   'from kernels%s group by name order by sum(duration) desc' % where).fetchall()
  total = sum(r[2] for r in rows) or 1
  print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
  for n, cnt, tot, avg, mn, mx in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (short(n), cnt, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == '__main__':
  gy = [a.split('=', 1)[1] for a in sys.argv[2:] if a.startswith('grid_y=')]
  main(sys.argv[1], gy[0] if gy else None)
