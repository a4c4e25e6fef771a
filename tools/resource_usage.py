"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log) per kernel."""
import re
import subprocess
import sys


def demangle(names):
  try:
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.strip().split('\n')
  except FileNotFoundError:
    return names


def main(path):
  txt = open(path).read()
  blocks = re.split(r'remark: [^\n]*Function Name: ', txt)
  rows = []
  for b in blocks[1:]:
    name = b.split('\n')[0].strip()

    def g(k):
      m = re.search(k + r': (\d+)', b)
      return int(m.group(1)) if m else -1

    rows.append((name, g('VGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                 g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
  names = demangle([r[0] for r in rows])
  for r, nm in zip(rows, names):
    nm = nm.replace('expo::', '').replace('_Float16', 'f16').replace('(anonymous namespace)::', '')
    nm = re.sub(r'\(.*$', '', nm)
    print(f'{r[1]:4d} vgpr {r[2]:4d} sgpr scratch {r[3]:3d} occ {r[4]} lds {r[5]:5d}  {nm[:120]}')


if __name__ == '__main__':
  main(sys.argv[1])
