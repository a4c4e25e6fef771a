"""`python bench.py --gpus N` must start N ranks by itself (the driver runs the plain command) and print
ONE line with n_gpus = N.  --dry-run exercises the launcher + rendezvous on CPU (gloo) without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
  env = dict(os.environ, OMP_NUM_THREADS='1')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', *extra], env=env,
                       capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout
  return json.loads(lines[0])


@pytest.mark.parametrize('workload', ['chain', 'train'])
def test_bench_self_launches_two_ranks(workload):
  line = run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--workload', workload, '--scaling', 'strong')
  assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
  assert line['scaling'] == 'strong' and line['dry_run'] is True


def test_bench_self_launches_eight_ranks_at_config_4_geometry():
  """BASELINE config 4 (8 ranks, the reference's global batch of 64 split image-wise: 8 images per rank) through the
  launcher and the gloo rendezvous on the CPU."""
  line = run_bench('--gpus', '8', '--steps', '2', '--warmup', '1', '--workload', 'train', '--scaling', 'strong')
  assert line['n_gpus'] == 8 and line['scaling'] == 'strong' and line['dry_run'] is True


def test_bench_single_rank_needs_no_launcher():
  line = run_bench('--gpus', '1')
  assert line['n_gpus'] == 1 and line['scaling'] == 'weak'


def test_strong_scaling_splits_the_global_batch():
  sys.path.insert(0, ROOT)
  import bench
  assert bench.local_shape((64, 512, 512, 3), 8, 'strong') == (8, 512, 512, 3)
  assert bench.local_shape((64, 512, 512, 3), 8, 'weak') == (64, 512, 512, 3)
  with pytest.raises(SystemExit):
    bench.local_shape((10, 8, 8, 3), 4, 'strong')


def test_traffic_is_null_for_a_shape_without_a_pmc_pass():
  sys.path.insert(0, ROOT)
  import bench
  assert bench.load_traffic('bwd_C', (64, 64, 64, 3), 'f16') is None
  assert bench.load_traffic('bwd_C', (64, 512, 512, 3), 'f32') is None
  assert bench.load_traffic('bwd_C', (64, 512, 512, 3), 'f16') > 3e8
