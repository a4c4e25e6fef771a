"""Kernel stats over the LAST fraction of a rocprofv3 trace (steady state, skipping warm-up/autotune).
usage: python tools/rocpd_tail_stats.py results.db [fraction=0.3] [top=25]"""
import sqlite3
import sys

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
c = sqlite3.connect(path)
t0, t1 = c.execute('select min(start), max(end) from kernels').fetchone()
cut = t1 - (t1 - t0) * frac
rows = c.execute('select name, count(*), sum(duration), avg(duration) from kernels where start >= ? group by name '
                 'order by sum(duration) desc', (cut,)).fetchall()
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print('window %.1f ms: %d launches, kernel time %.1f ms (%.1f%% busy), avg %.1f us/launch' %
      ((t1 - cut) / 1e6, n, tot / 1e6, 100.0 * tot / (t1 - cut), tot / max(n, 1) / 1e3))
for name, cnt, s, a in rows[:top]:
  print('%6d  %9.1f us total  %8.1f us avg  %5.1f%%  %s' % (cnt, s / 1e3, a / 1e3, 100.0 * s / tot, name[:90]))
