"""The committed measurement artefacts under profiles/ must be mutually consistent: traffic.json is
what tools/make_traffic.py derives from the committed PMC CSVs, every chain kernel appears in the
rocprofv3 kernel table, and the measured HBM traffic equals the algorithmic bytes bench.py uses."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
ALGO = {'fwd': 2 * 3 * 2 * 64 * 512 * 512, 'bwd': 3 * 3 * 2 * 64 * 512 * 512}  # bytes per launch, fp16, shape C


def test_traffic_json_matches_the_committed_counters(tmp_path):
  d = tmp_path / 'final'
  d.mkdir()
  for name in ('pmc_fetch_size', 'pmc_write_size', 'pmc_fetch_size_calibration', 'pmc_write_size_calibration'):
    shutil.copy(os.path.join(PROF, 'r01_final_%s.csv' % name), d / (name + '.csv'))
  out = tmp_path / 'traffic.json'
  subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_traffic.py'), str(d), str(out)], check=True,
                 capture_output=True)
  derived = json.load(open(out))['64x512x512x3:f16']
  committed = json.load(open(os.path.join(PROF, 'traffic.json')))['64x512x512x3:f16']
  keys = sorted(k for k in committed if not k.startswith('_'))
  assert len(keys) == 16 and keys == sorted(k for k in derived if not k.startswith('_'))
  for k in keys:
    assert derived[k] == committed[k], k
    algo = ALGO[k[:3]]
    assert abs(committed[k] - algo) <= 0.005 * algo, (k, committed[k], algo)  # no wasted re-reads


def test_kernel_table_lists_every_chain_kernel():
  rows = list(csv.DictReader(open(os.path.join(PROF, 'r01_final_kernel_stats.csv'))))
  names = [r['Name'] for r in rows]
  for direction in ('fwd', 'bwd'):
    mine = [n for n in names if 'filter_%s_kernel' % direction in n]
    assert len(mine) == 8, (direction, mine)
  # the bench line's dominant kernel must be one of them and its HIP-event time within 10 % of rocprof's
  bench = json.load(open(os.path.join(PROF, 'r01_final_bench_chain.json')))
  dom = bench['roofline']['kernel']
  tag = {'E': '9ExposureF', 'G': '6GammaF', 'W': '13WhiteBalanceF', 'S+': '8SatPlusF', 'T': '6CurveFILi1',
         'Ct': '9ContrastF', 'BW': '4WnbF', 'C': '6CurveFILi3'}[dom[4:]]
  row = next(r for r in rows if 'filter_%s_kernelINS_%sE' % (dom[:3], tag) in r['Name'])
  rocprof_ms = float(row['AverageNs']) * 1e-6
  assert abs(bench['roofline']['avg_launch_ms'] - rocprof_ms) <= 0.10 * rocprof_ms
