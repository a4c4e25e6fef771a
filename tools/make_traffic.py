"""profiles/traffic.json (HBM bytes per launch, read by bench.py for roofline.traffic) from the
FETCH_SIZE / WRITE_SIZE CSVs that tools/profile_all.sh produced.
usage: python tools/make_traffic.py gpurun_out/final [out.json [workload-key [px]]]

The file is keyed by the workload the counters were collected on ("64x512x512x3:f16", the default);
an existing file's other keys are kept, so passes on several shapes accumulate.  bench.py reports
roofline.traffic = null for a shape / dtype without an entry.

traffic = kf * FETCH_SIZE * 1024 + kw * WRITE_SIZE * 1024.  The factors are calibrated in the same
run on tools/membench kernels that move a known byte count with the same dwordx3 access pattern and
the same cache policy as the filter kernels (cpol<2,16>: 96 MiB read + 96 MiB written; rpol<2,2,16>:
2 x 96 MiB read + 96 MiB written): on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
(MI355X_MICROARCH.md, HBM section), so kf is expected to be 2 and kw 1."""
import csv
import json
import re
import sys

NAMES = {'9ExposureF': 'E', '6GammaF': 'G', '13WhiteBalanceF': 'W', '8SatPlusF': 'S+', '6CurveFILi1': 'T',
         '9ContrastF': 'Ct', '4WnbF': 'BW', '6CurveFILi3': 'C'}
MIB96 = 96 * 1024 * 1024


def read(path):
  return [r for r in csv.DictReader(open(path))]


def calib(rows, counter, kernel, known_bytes):
  for r in rows:
    if r['Counter'] == counter and r['Kernel'].startswith(kernel):
      return known_bytes / (float(r['Mean']) * 1024.0)
  raise SystemExit('calibration kernel %s not found' % kernel)


def main(d, out, wkey='64x512x512x3:f16', suffix=''):
  sfx = ('_' + suffix) if suffix else ''
  cal = '_calibration_512' if suffix == 'cold' else '_calibration'
  buf = (512 if suffix == 'cold' else 96) * 1024 * 1024  # bytes per stream of the calibration kernels
  kf_c = calib(read(d + '/pmc_fetch_size%s.csv' % cal), 'FETCH_SIZE', 'void cpol<2, 16>', buf)
  kf_r = calib(read(d + '/pmc_fetch_size%s.csv' % cal), 'FETCH_SIZE', 'void rpol<2, 2, 16>', 2 * buf)
  kw_c = calib(read(d + '/pmc_write_size%s.csv' % cal), 'WRITE_SIZE', 'void cpol<2, 16>', buf)
  kw_r = calib(read(d + '/pmc_write_size%s.csv' % cal), 'WRITE_SIZE', 'void rpol<2, 2, 16>', buf)
  kf, kw = round(0.5 * (kf_c + kf_r)), round(0.5 * (kw_c + kw_r))
  res = {}
  for kind, rows in (('f', read(d + '/pmc_fetch_size%s.csv' % sfx)), ('w', read(d + '/pmc_write_size%s.csv' % sfx))):
    for r in rows:
      m = re.search(r'filter_(fwd|bwd)_kernelINS_', r['Kernel'])
      if not m or 'DF16_' not in r['Kernel']:
        continue
      tail = r['Kernel'][m.end():]
      name = next((v for k, v in NAMES.items() if tail.startswith(k + 'E')), None)
      if name is None:
        continue
      key = '%s_%s' % (m.group(1), name)
      res.setdefault(key, {})[kind] = float(r['Mean']) * 1024.0
  traffic = {k: int(round(kf * v['f'] + kw * v['w'])) for k, v in sorted(res.items()) if 'f' in v and 'w' in v}
  doc = {'_comment': 'HBM bytes per launch at ' + wkey + ' from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE '
                     '(separate passes, tools/profile_all.sh). traffic = kf*FETCH_SIZE*1024 + kw*WRITE_SIZE*1024 with '
                     'kf = %d, kw = %d calibrated in the same run on tools/membench kernels of known byte count, same '
                     'dwordx3 access pattern and cache policy (measured factors: cpol<2,16> fetch %.4f write %.4f; '
                     'rpol<2,2,16> fetch %.4f write %.4f).'
                     % (kf, kw, kf_c, kw_c, kf_r, kw_r)}
  try:
    old = json.load(open(out))
  except (OSError, ValueError):
    old = {}
  merged = {k: v for k, v in old.items() if isinstance(v, dict)}
  merged[wkey] = dict(traffic, _calibration=doc['_comment'])
  doc = {'_comment': 'HBM bytes per launch by workload key "NxHxWx3:dtype" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in '
                     'separate passes; traffic = kf*FETCH_SIZE*1024 + kw*WRITE_SIZE*1024, factors calibrated in-run: '
                     'see _calibration of each entry)'}
  doc.update(merged)
  json.dump(doc, open(out, 'w'), indent=1)
  print(json.dumps(doc, indent=1))


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'profiles/traffic.json',
       sys.argv[3] if len(sys.argv) > 3 else '64x512x512x3:f16', sys.argv[4] if len(sys.argv) > 4 else '')
