"""-m gpu twin of tests/test_dist_gloo.py: the image-sharded data-parallel training step with TWO ranks on the one
MI355X there is -- every kernel of the step is the real one (libexposure_hip.so, MIOpen, hipBLASLt), the ranks hold
disjoint halves of the global minibatch, draw their random inputs from the shared global-index generator, and all-reduce
their gradient buckets from the backward hooks.  The transport is gloo (RCCL refuses two ranks on one device), so what
this does NOT cover is RCCL itself and the hipGraph capture of collectives: those are
test_training_step_graph_captures_rccl_collectives (one rank, forced RCCL) and the driver's 8-GPU run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _inputs(dev, n=8, s=64):
  rng = np.random.default_rng(0)
  t = lambda a: torch.from_numpy(a).to(dev)
  img = t((rng.random((n, s, s, 3), dtype=np.float32)**2.2).astype(np.float32))
  real = t(rng.random((n, s, s, 3), dtype=np.float32))
  states = torch.zeros(n, 11)
  states[:, 2] = torch.from_numpy(rng.integers(0, 4, n).astype(np.float32))
  z = t(rng.random((n, 131), dtype=np.float32))
  return img, real, states.to(dev), z


def _run_steps(gan, img, real, states, z, iters=1):
  for _ in range(iters):
    g = gan.generator_step(img, z, states, progress=0.1, it=7)
    c = gan.critic_step(real, g['fake_output'].float(), it=7)
  return g, c


def _snapshot(gan, g, c):
  return {'params': [p.detach().cpu().clone() for p in gan.parameters()],
          'grads': [p.grad.detach().cpu().clone() for p in gan.parameters()],
          'ids': g['debug']['selected_filter_ids'].cpu() if 'debug' in g else None,
          'g_loss': float(g['g_loss']), 'c_loss': float(c['c_loss'])}


def _worker(rank, world, port, out_dir, use_graphs, iters):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dev = torch.device('cuda:0')
  torch.cuda.set_device(dev)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from exposure_amd import dist as xdist
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  torch.manual_seed(123)  # identical initial weights on every rank
  gan = GAN(make_cfg(), device=dev, use_graphs=use_graphs, seed=77)
  img, real, states, z = _inputs(dev)
  sh = xdist.shard
  import warnings
  after_first = None
  with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    for k in range(iters):
      g, c = _run_steps(gan, sh(img), sh(real), sh(states), sh(z), 1)
      if k == 0 and iters > 1:
        torch.cuda.synchronize()
        after_first = [p.detach().cpu().clone() for p in gan.parameters()]
  torch.cuda.synchronize()
  snap = _snapshot(gan, g, c)
  snap['params_after_first'] = after_first
  snap['still_graphs'] = bool(gan.use_graphs)
  snap['warned'] = any('stay eager' in str(w.message) for w in caught)
  torch.save(snap, os.path.join(out_dir, 'rank%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('use_graphs,iters', [(False, 1), (True, 2)])
def test_two_ranks_on_one_gpu_match_one_process(gpu_device, tmp_path, use_graphs, iters):
  """use_graphs with a transport whose watchdog drain cannot be verified (gloo has no flight recorder): the second call
  of each step kind must NOT capture -- it warns and stays eager (GAN._replay) -- and the results are the same."""
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), use_graphs, iters), nprocs=2, join=True)
  r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
  r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
  torch.manual_seed(123)
  ref = GAN(make_cfg(), device=gpu_device, seed=77)
  inputs = _inputs(gpu_device)
  g, c = _run_steps(ref, *inputs, iters=1)
  if iters > 1:
    # The gradients of iteration k are a function of the weights after iteration k - 1 (and of the shared noise
    # stream).  After an Adam step -- which moves every weight by ~lr whatever the size of its gradient, so
    # rounding-level differences in near-zero gradients become lr-sized weight differences -- the two runs' weights
    # differ at the 1e-4 level and iteration-2 gradients by up to 8e-3 of a tensor's largest (measured, round 4): that
    # divergence is Adam's, not the exchange's.  So the one-process run CONTINUES FROM THE RANKS' WEIGHTS (checked first
    # to be within Adam's first-step size of its own): iteration 2 is then compared from identical weights under the
    # same bound as iteration 1.
    torch.cuda.synchronize()
    worst = max(float((a - p.detach().cpu()).abs().max()) for a, p in zip(r0['params_after_first'], ref.parameters()))
    assert worst < 3e-4, worst
    with torch.no_grad():
      for p, a in zip(ref.parameters(), r0['params_after_first']):
        p.copy_(a.to(p.device))  # in place: captured graphs and the packed heads keep their storage
    for _ in range(iters - 1):
      g, c = _run_steps(ref, *inputs, iters=1)
  torch.cuda.synchronize()
  want = _snapshot(ref, g, c)
  if use_graphs:
    assert r0['warned'] and not r0['still_graphs']  # the eager fallback was taken, loudly
  for a, b in zip(r0['params'], r1['params']):
    assert torch.equal(a, b)  # the ranks stay in lock-step: same reduced gradients, same update
  for a, b in zip(r0['grads'], r1['grads']):
    assert torch.equal(a, b)
  # all-reduced mean gradients == the full-batch gradients of the single process
  gmax = max(float(ref_g.abs().max()) for ref_g in want['grads'])
  for got, ref_g in zip(r0['grads'], want['grads']):
    scale = float(ref_g.abs().max()) + 1e-12
    # (fp32 convolutions of 4 and of 8 images through different decompositions, and the double backward of the gradient
    # penalty on top: 1.7e-3 of a tensor's largest gradient seen; a wrong reduction -- sum instead of mean, a missing shard --
    # would be off by a factor.  The absolute floor is 2e-5 of the LARGEST gradient in the model: a head whose filter one or
    # two images selected has gradients of ~1e-6 -- five orders below the model's 7e-2 -- that are one per-image scalar, a sum
    # over 4 096 pixels with heavy cancellation, times fixed tensors; a rounding-level change of the image gradient moves that
    # scalar, and with it EVERY tensor of the head by the same factor (seen: 4 % of 2.8e-6 .. 2.2e-5 on all four tensors of
    # the SaturationPlus head when the paired convolution launches changed their K slicing: tools/r06/dbg_dist.py))
    assert float((got - ref_g).abs().max()) <= 5e-3 * scale + 2e-5 * gmax
  worst = max(float((a - b).abs().max()) for a, b in zip(r0['params'], want['params']))
  # Adam's steps are ~lr-sized (tests/test_dist_gloo.py); the moments of iteration 1 differ at rounding level
  assert worst < 3e-4, worst


def test_bench_launches_two_ranks_that_share_the_gpu(gpu_device):
  """`python bench.py --gpus 2` (the plain command the driver runs; no launcher around it) in the test mode where both
  ranks use the one GPU and talk over gloo: the self-launch through torch.distributed.run, the per-rank shards, the
  device-side barrier, the MAX over ranks and the single JSON line of a 2-rank run, with real kernels."""
  import json
  import subprocess
  env = dict(os.environ, EXPO_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  for extra in (['--scaling', 'weak', '--shape', 'B'], ['--scaling', 'strong', '--shape', 'B', '--no-legs']):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                          '--no-cpu-baseline', '--cold-shape', 'none'] + extra, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == extra[1] and d['value'] > 0
    per_gpu = 16 if extra[1] == 'weak' else 8
    assert d['config']['batch_per_gpu'] == per_gpu and d['config']['global_batch'] == 2 * per_gpu
    assert 'gloo' in d['config']['transport']
    if '--no-legs' in extra:
      assert 'legs' not in d
      continue
    # the extra legs of the default line, run by BOTH ranks after the timed region: config 2's shape, one training
    # iteration with its gradient exchange, the gradient buckets' all-reduce alone
    legs = d['legs']
    assert 'error' not in legs, legs
    for name in ('chain_64x64x64x3', 'train', 'allreduce'):
      assert name in legs and 'error' not in legs[name], (name, legs.get(name))
    assert legs['chain_64x64x64x3']['Mpixels_per_s'] > 0
    assert legs['train']['ms_per_iteration'] > 0 and legs['train']['images_per_s'] > 0
    assert 0 < legs['train']['roofline']['frac'] < 1 and legs['train']['roofline']['bound'] == 'mfma_fp32'
    assert legs['allreduce']['bytes_per_iteration'] > 50e6 and legs['allreduce']['bus_GBps'] > 0


@pytest.mark.parametrize('workload', ['train', 'allreduce', 'infer'])
def test_bench_other_workloads_with_two_ranks_sharing_the_gpu(workload, gpu_device):
  """The same for the workloads with a gradient exchange (train: bucketed all-reduce from the backward hooks, both
  ranks' losses are global-batch means; allreduce: the buckets alone) and for the inference workload."""
  import json
  import subprocess
  env = dict(os.environ, EXPO_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', workload, '--steps', '3', '--warmup', '2']
  if workload == 'infer':
    cmd += ['--shape', 'B']
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['steps'] == 3 and d['value'] > 0
