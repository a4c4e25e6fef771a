"""Pretty-print bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json
import sys

for path in sys.argv[1:]:
  try:
    d = json.load(open(path))
  except Exception as e:  # noqa: BLE001
    print(path, 'ERR', e)
    continue
  pk = d.get('per_kernel', {})
  print('%s: ms/step %.4f  value %.0f %s  chain %.0f GB/s  sum_kernels %.4f ms' %
        (path, d['ms_per_step'], d['value'], d['unit'], d['config'].get('chain_algorithmic_GBps', 0),
         sum(v['ms'] for v in pk.values())))
  if pk:
    print('   fwd us:', ' '.join('%s %.1f' % (k[4:], v['ms'] * 1e3) for k, v in pk.items() if k.startswith('fwd')))
    print('   bwd us:', ' '.join('%s %.1f' % (k[4:], v['ms'] * 1e3) for k, v in pk.items() if k.startswith('bwd')))
  if 'roofline' in d:
    r = d['roofline']
    print('   roofline: %s %.0f GB/s frac %.3f' % (r['kernel'], r['achieved'], r['frac']))
  if 'cpu_baseline' in d:
    print('   cpu:', d['cpu_baseline'])
