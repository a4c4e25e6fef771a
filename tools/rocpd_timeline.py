"""Launch-ordered timeline of the LAST `window_ms` of a rocprofv3 rocpd trace: start offset, duration and the idle gap in
front of every kernel, plus the gaps summed by the kernel that follows them -- where a step-graph replay waits for the host
and which eager launches sit between the graphs.
usage: python tools/rocpd_timeline.py results.db window_ms [max_rows] > gpurun_out/xxx_timeline.txt"""
import collections
import sqlite3
import sys

from rocpd_stats import short


def main(path, window_ms, max_rows=700):
  c = sqlite3.connect(path)
  t1, = c.execute('select max(end) from kernels').fetchone()
  cut = t1 - int(window_ms * 1e6)
  rows = c.execute('select name, start, end from kernels where start >= ? order by start', (cut,)).fetchall()
  busy = sum(e - s for _, s, e in rows)
  gaps = collections.Counter()
  counts = collections.Counter()
  prev_end = rows[0][1]
  lines = []
  for name, s, e in rows:
    gap = max(0, s - prev_end)
    nm = short(name)[:70]
    gaps[nm] += gap
    counts[nm] += 1
    lines.append('%10.1f %8.1f %8.1f  %s' % ((s - rows[0][1]) / 1e3, (e - s) / 1e3, gap / 1e3, nm))
    prev_end = max(prev_end, e)
  span = prev_end - rows[0][1]
  print('# %d launches, span %.3f ms, busy %.3f ms, idle %.3f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
  print('# idle time by the kernel that FOLLOWS the gap (us total, launches, us per launch)')
  for nm, g in gaps.most_common(25):
    print('#   %9.1f %5d %7.2f  %s' % (g / 1e3, counts[nm], g / 1e3 / counts[nm], nm))
  print('# offset_us duration_us gap_us name')
  for ln in lines[:max_rows]:
    print(ln)


if __name__ == '__main__':
  main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 700)
