"""Per-kernel stats over the LAST `window_ms` of a rocprofv3 rocpd trace -- the timed region of a bench.py run
(its steps x ms_per_step), so warm-up, MIOpen's find-mode trials and hipGraph captures stay out of the table.
usage: python tools/rocpd_window_stats.py results.db window_ms [iterations] > profiles/xxx_kernel_stats_train.csv
The first line is a '#' comment with the window totals (launches, busy time, per-iteration figures)."""
import sqlite3
import sys

from rocpd_stats import short


def main(path, window_ms, iters=1):
  c = sqlite3.connect(path)
  t0, t1 = c.execute('select min(start), max(end) from kernels').fetchone()
  cut = t1 - int(window_ms * 1e6)
  rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels '
                   'where start >= ? group by name order by sum(duration) desc', (cut,)).fetchall()
  total = sum(r[2] for r in rows) or 1
  n = sum(r[1] for r in rows)
  print('# window %.2f ms (of %.0f ms traced), %d iterations: %d launches = %.0f per iteration, kernel time %.2f ms '
        '= %.3f ms per iteration (%.1f %% of the window), %d distinct kernels' %
        (window_ms, (t1 - t0) / 1e6, iters, n, n / iters, total / 1e6, total / 1e6 / iters,
         100.0 * total / (window_ms * 1e6), len(rows)))
  print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
  for name, cnt, tot, avg, mn, mx in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (short(name), cnt, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == '__main__':
  main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
