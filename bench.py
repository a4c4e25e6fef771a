#!/usr/bin/env python
"""Headline benchmark: Mpixels/s through the 8-step filter chain, forward + backward.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: the 8 filters of cfg.filters
(E,G,W,S+,T,Ct,BW,C; /root/reference/config_example.py:22-25) applied sequentially, one HIP
kernel per filter step forward and one backward (dx + per-image parameter gradients), on a
64x512x512x3 fp16 NHWC batch already resident in HBM.  Multi-GPU: images are independent, so
ranks hold disjoint replicas of the batch shape (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the
dominant kernel (HIP-event timed, algorithmic bytes: 12 B/px fwd, 18 B/px bwd at fp16) and
`cpu_baseline` (the torch-CPU fp32 op-by-op restatement, oracle/filters_torch.py, timed on the
host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from exposure_amd import _cabi, synthetic  # noqa: E402

FILTER_NAMES = synthetic.FILTER_NAMES
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceilings: DESIGN.md 3.1


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--shape', default=None, help='A|B|C (synthetic.SHAPES) or N,H,W; default C (chain), B (infer)')
  ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
  ap.add_argument('--kernel-reps', type=int, default=50, help='launches per kernel for the roofline timing')
  ap.add_argument('--prewarm-s', type=float, default=0.3, help='untimed clock warm-up before the W warm-up steps')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-per-kernel', action='store_true')
  ap.add_argument('--seed', type=int, default=1234)
  ap.add_argument('--order', default='0,1,2,3,4,5,6,7', help='filter ids of the chain steps (experiments only; the '
                  'metric is defined on the cfg.filters order 0..7)')
  ap.add_argument('--workload', default='chain', choices=['chain', 'train', 'infer', 'allreduce'],
                  help="chain: the headline filter-chain metric; train: one reference training iteration "
                  "(1 generator/value step + cfg.citers critic steps, net.py:307-365) on 64 images per GPU")
  ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                  help='replay the 17 launches of a step from one hipGraph (auto = on; off: eager C-ABI calls)')
  return ap.parse_args()


def parse_shape(name):
  if name in synthetic.SHAPES:
    return synthetic.SHAPES[name]
  return tuple(int(v) for v in name.split(',')) + (3,)


def make_device_case(shape, dtype, dev, seed):
  """Same distributions as synthetic.make_case, generated on the device for the big shapes."""
  g = torch.Generator(device=dev).manual_seed(seed)
  x = torch.rand(shape, device=dev, generator=g, dtype=torch.float32)
  x = (x**2.2) * (1.0 / 0.99**2.2)
  dy = torch.randn(shape, device=dev, generator=g, dtype=torch.float32)
  rng = np.random.default_rng(seed)
  params = [torch.from_numpy(synthetic.make_params(rng, fid, shape[0])).to(dev) for fid in range(8)]
  return x.to(dtype), dy.to(dtype), params


class Chain:
  """Buffers + one-call-per-direction launch of the 8-step chain (expo_chain_fwd / expo_chain_bwd)."""

  def __init__(self, shape, dtype, dev, seed, ids=None):
    self.ids = list(ids) if ids is not None else list(range(8))
    x, dy, params = make_device_case(shape, dtype, dev, seed)
    self.params = [params[i] for i in self.ids]
    self.acts = [x] + [torch.empty_like(x) for _ in range(8)]
    # gradients ping-pong between two buffers; grads[8] = upstream dy
    ga, gb = torch.empty_like(x), torch.empty_like(x)
    self.grads = [ga if (i % 2 == 0) else gb for i in range(8)] + [dy]
    # per-step parameter gradients carved from one flat buffer -> a single zero-fill per backward
    flat = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
    self.dparams, off = [], 0
    for p in self.params:
      self.dparams.append(flat[off:off + p.numel()].view_as(p))
      off += p.numel()

    self.graph = None
    self.unroll = 1

  def launch(self):
    _cabi.chain_fwd(self.ids, self.acts, self.params)
    _cabi.chain_bwd(self.ids, self.acts, self.grads, self.params, self.dparams)

  def capture(self, unroll=1):
    """Capture `unroll` steps (each 8 fwd + 1 fill + 8 bwd launches) into one hipGraph: small shapes
    are bound by the ~5 us host cost per launch, a graph replay pays it once; every replay boundary
    costs ~3 us on the device, so several steps share a replay when the step count allows."""
    self.unroll = unroll
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      self.launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    # thread_local: with a process group alive, RCCL's watchdog thread may poll events while this
    # thread captures; under the default "global" mode such a call aborts the process
    try:
      with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
        for _ in range(unroll):
          self.launch()
    except RuntimeError as e:  # never lose the run over the launch method
      print('warning: hipGraph capture failed (%s); eager launches' % e, file=sys.stderr)
      torch.cuda.synchronize()
      self.graph = None

  def run(self, steps):
    """Exactly `steps` chain steps: whole replays of the captured graph, the remainder eagerly."""
    if self.graph is None:
      for _ in range(steps):
        self.launch()
      return
    for _ in range(steps // self.unroll):
      self.graph.replay()
    for _ in range(steps % self.unroll):
      self.launch()


def time_kernels(chain, reps):
  """Average duration (ms) of each of the 16 kernels measured IN the chain sequence (so every
  kernel sees the cache state it sees in the timed region: each step writes a fresh 96 MiB tensor),
  with a HIP event pair around every launch on the launch stream.  The pairs read ~2.3 us more than
  the per-kernel averages of `rocprofv3 --kernel-trace` for the same command (profiles/), i.e. the
  roofline fraction reported from them is slightly conservative."""
  ids = chain.ids
  nsteps = len(ids)
  names = ['fwd_' + FILTER_NAMES[i] for i in ids] + ['bwd_' + FILTER_NAMES[i] for i in reversed(ids)]
  ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in names]
        for _ in range(reps)]
  for r in range(-2, reps):  # two untimed passes first
    k = 0
    for i in range(nsteps):
      if r >= 0:
        ev[r][k][0].record()
      _cabi.filter_fwd(ids[i], chain.acts[i], chain.acts[i + 1], chain.params[i])
      if r >= 0:
        ev[r][k][1].record()
      k += 1
    for i in reversed(range(nsteps)):
      if r >= 0:
        ev[r][k][0].record()
      _cabi.filter_bwd(ids[i], chain.acts[i], chain.grads[i + 1], chain.grads[i], chain.params[i], chain.dparams[i],
                       accumulate=True)  # kernel only: the chain zero-fills all dparams once per step
      if r >= 0:
        ev[r][k][1].record()
      k += 1
  torch.cuda.synchronize()
  out = {}
  for k, name in enumerate(names):
    out[name] = sum(ev[r][k][0].elapsed_time(ev[r][k][1]) for r in range(reps)) / reps
  return out


def cpu_baseline():
  """torch-CPU fp32 op-by-op restatement (oracle/filters_torch.py) timed on the host cores on a
  bounded sample (4x512x512x3 = 1 Mpixel per pass).  torch's intra-op pool does not scale to every
  core of a big host for this op mix, so a few thread counts are tried (1 warm-up + best of 2 each)
  and the best is reported together with the thread count that produced it."""
  from oracle import filters_torch as ft
  ncpu = os.cpu_count() or 1
  shape = (4, 512, 512, 3)
  x, dy, params = synthetic.make_case(1234, shape, np.float16)
  tx = torch.from_numpy(x.astype(np.float32))
  tdy = torch.from_numpy(dy.astype(np.float32))
  tp = [torch.from_numpy(p) for p in params]
  px = shape[0] * shape[1] * shape[2]
  best, best_threads = float('inf'), 1
  t_start = time.perf_counter()
  for threads in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
    torch.set_num_threads(threads)
    for it in range(3):
      t0 = time.perf_counter()
      ft.chain_fwd_bwd(tx, tp, tdy)
      dt = time.perf_counter() - t0
      if it > 0 and dt < best:
        best, best_threads = dt, threads
    if time.perf_counter() - t_start > 25.0:
      break
  model = ''
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        model = line.split(':', 1)[1].strip()
        break
  except OSError:
    pass
  return {
      'value': px / best / 1e6,
      'unit': 'Mpixels/s',
      'cores': best_threads,
      'kind': 'port',
      'sample': 'CPU restatement (torch fp32 op-by-op, best of {8,16,32,64} threads = %d, host %s with %d '
                'logical CPUs): 8-step chain fwd+bwd on 4x512x512x3, best of 2 after 1 warm-up' %
                (best_threads, model or 'unknown CPU', ncpu),
  }


def load_traffic(kernel):
  """Per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
  path = os.path.join(ROOT, 'profiles', 'traffic.json')
  try:
    return json.load(open(path)).get(kernel)
  except (OSError, ValueError):
    return None


_BARRIER_FLAG = {}


def light_barrier(dist, dev):
  """Cross-rank barrier for the timed region: a 1-element all-reduce enqueued on the device (every
  rank must contribute before any rank's copy completes); the torch.cuda.synchronize() that follows
  it in the timing bracket waits for it.  `dist.barrier()` costs ~0.5 ms of host-side NCCL work per
  call, which is 5 % of a 20-step chain measurement; this costs one small kernel."""
  if dist is None:
    return
  flag = _BARRIER_FLAG.get(dev)
  if flag is None:
    flag = _BARRIER_FLAG[dev] = torch.zeros(1, device=dev)
  dist.all_reduce(flag)


def run_train(args, world, rank, dev, dist):
  """BASELINE configs 3/4: agent rollout step + policy CNN + WGAN-GP critic, batch 64 per GPU,
  random-init weights, synthetic FiveK-shaped inputs, gradients all-reduced over RCCL."""
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  cfg = make_cfg()
  torch.manual_seed(args.seed)  # identical initial weights on every rank
  gan = GAN(cfg, device=dev, use_graphs=(args.graph != 'off'))
  n = cfg.batch_size
  from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
  pool_dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  memory = ReplayMemory(cfg, SyntheticProvider(dev, dtype=pool_dtype, seed=args.seed + 10 * rank + 1),
                        SyntheticProvider(dev, gamma=1.0, dtype=pool_dtype, seed=args.seed + 10 * rank + 2),
                        seed=args.seed + rank)
  # net.py:320-328: the first iteration rolls the generator with lr_g = 0 until terminated
  # trajectories exist for the critic to replay (100 steps in the reference; 8 suffice: 5 steps end one)
  for _ in range(8):
    feed, feats = memory.get_feed_dict_and_states(n)
    out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
    memory.replace_memory(out['fake_output'], out['new_states'], feats)

  def iteration(it):
    feed, feats = memory.get_feed_dict_and_states(n)
    out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], progress=it / cfg.max_iter_step, it=it)
    memory.replace_memory(out['fake_output'], out['new_states'], feats)
    for _ in range(cfg.citers):
      rep = memory.get_replay_feed_dict(n)
      gan.critic_step(rep['real_data'], rep['fake_output'], it=it)
    return out

  def barrier():
    light_barrier(dist, dev)

  for i in range(args.warmup):
    iteration(i + 1)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(args.steps):
    iteration(i + 1)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  if rank == 0:
    ms = elapsed / args.steps * 1e3
    print(json.dumps({
        'metric': 'generator-step images/s (1 G/V step + %d critic steps per iteration)' % cfg.citers,
        'value': world * n / (elapsed / args.steps),
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': '%s images / f32 nets' % args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': 'reference training iteration (net.py:307-365): agent rollout step, policy CNN, value net, '
                        'WGAN-GP critic; batch %d x 64x64x3 per GPU; random-init weights' % n,
            'global_batch': world * n,
            'parallelism': 'dp%d image-sharded, 3 flat gradient buckets over RCCL' % world,
            'launch': 'one hipGraph replay per G/V step and per critic step' if gan._replay_steps else 'eager',
            'reference_note': 'README.md:43: ~0.30 s/iteration on a GTX 1080 Ti (whole run ~100 min / 20000 it)',
        },
    }))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def run_allreduce(args, world, rank, dev, dist):
  """SURVEY.md section 8(e) "all-reduce only": the three flat fp32 gradient buckets of one training
  iteration (theta_g 24.5 MB + theta_v 4.9 MB once, theta_c 4.9 MB x citers) reduced over RCCL, no
  compute.  With one rank the collective degenerates to nothing and the line reports 0 bytes."""
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  cfg = make_cfg()
  torch.manual_seed(args.seed)
  gan = GAN(cfg, device=dev, use_graphs=False)
  sizes = {name: b.numel for name, b in gan.buckets.items()}
  bufs = {k: torch.zeros(v, dtype=torch.float32, device=dev) for k, v in sizes.items()}

  def iteration():
    if dist is None:
      return
    dist.all_reduce(bufs['g'])
    dist.all_reduce(bufs['v'])
    for _ in range(cfg.citers):
      dist.all_reduce(bufs['c'])

  for _ in range(args.warmup):
    iteration()
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    iteration()
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  nbytes = 4 * (sizes['g'] + sizes['v'] + cfg.citers * sizes['c'])
  if rank == 0:
    ms = elapsed / args.steps * 1e3
    print(json.dumps({
        'metric': 'gradient all-reduce GB/s per training iteration (bucket bytes / time)',
        'value': (nbytes / 1e9) / (elapsed / args.steps) if world > 1 else 0.0,
        'unit': 'GB/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': 'all-reduce only: theta_g %d + theta_v %d + %d x theta_c %d fp32 elements per iteration'
                        % (sizes['g'], sizes['v'], cfg.citers, sizes['c']),
            'bytes_per_iteration': nbytes,
            'parallelism': 'dp%d, flat buckets over RCCL' % world,
        },
    }))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def run_infer(args, world, rank, dev, dist):
  """BASELINE config 5: high-resolution inference, 16x512x512x3 fp16, the 8 filters of cfg.filters
  applied to every image -- (a) one kernel per step (8 reads + 8 writes of the image), (b) the fused
  multi-step forward (1 read + 1 write).  Forward only."""
  shape = parse_shape(args.shape or 'B')
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  esz = 2 if args.dtype == 'f16' else 4
  x, _dy, params = make_device_case(shape, dtype, dev, args.seed + rank)
  n = shape[0]
  px = shape[0] * shape[1] * shape[2]
  ids = torch.arange(8, dtype=torch.int32, device=dev)[None, :].repeat(n, 1).contiguous()
  p24 = torch.zeros((n, 8, 24), dtype=torch.float32, device=dev)
  for fid in range(8):
    p24[:, fid, :params[fid].shape[1]] = params[fid]
  y = torch.empty_like(x)
  acts = [x] + [torch.empty_like(x) for _ in range(8)]

  def timed(fn):
    for _ in range(args.warmup):
      fn()
    torch.cuda.synchronize()
    light_barrier(dist, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      fn()
    torch.cuda.synchronize()
    light_barrier(dist, dev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
      t = torch.tensor([el], dtype=torch.float64, device=dev)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      el = float(t.item())
    return el / args.steps

  t_fused = timed(lambda: _cabi.chain_fused_fwd(ids, p24, x, y))
  t_steps = timed(lambda: _cabi.chain_fwd(list(range(8)), acts, params))
  same = float((y.float() - acts[8].float()).abs().max())
  if rank == 0:
    print(json.dumps({
        'metric': 'Mpixels/s through 8-step filter chain fwd (high-res inference)',
        'value': world * px / t_fused / 1e6,
        'unit': 'Mpixels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': t_fused * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': 'fused 8-step chain forward (expo_chain_fused_fwd), %dx%dx%dx3 %s per GPU' %
                        (shape[0], shape[1], shape[2], args.dtype),
            'algorithmic_GBps_at_96B_per_pixel': 8 * 2 * 3 * esz * px / t_fused / 1e9,
            'actual_traffic_GBps_at_12B_per_pixel': 2 * 3 * esz * px / t_fused / 1e9,
            'per_step_kernels_ms': t_steps * 1e3,
            'per_step_kernels_Mpixels_per_s': world * px / t_steps / 1e6,
            'speedup_vs_per_step': t_steps / t_fused,
            'max_abs_diff_fused_vs_per_step': same,
        },
    }))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def main():
  args = parse()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  dist = None
  if world > 1 or 'LOCAL_RANK' in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=dev)
  if args.gpus != world and rank == 0:
    print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
  if args.workload == 'train':
    return run_train(args, world, rank, dev, dist)
  if args.workload == 'infer':
    return run_infer(args, world, rank, dev, dist)
  if args.workload == 'allreduce':
    return run_allreduce(args, world, rank, dev, dist)

  shape = parse_shape(args.shape or 'C')
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  esz = 2 if args.dtype == 'f16' else 4
  chain = Chain(shape, dtype, dev, args.seed + rank, [int(v) for v in args.order.split(',')])
  px = shape[0] * shape[1] * shape[2]
  # One hipGraph replay per step by default.  Small shapes are launch-bound (17 launches in 87 us
  # eagerly vs 57 us replayed at 64x64x64); at 64x512x512 eager launches are ~0.5 % faster in a
  # plain process but 1.5-4 % slower and noisy once a process group exists (torchrun, RCCL's extra
  # queues), while the replay measures the same +-0.3 % either way -- so every N uses the replay.
  use_graph = args.graph != 'off'
  if use_graph:
    # as many steps per replay as divide the timed step count (at most 10): K stays exact
    chain.capture(max(u for u in range(1, 11) if args.steps % u == 0))

  def barrier():
    light_barrier(dist, dev)

  # untimed: leave the idle power state first (the first ~100 ms after idle run at lower clocks and
  # would make a 20-step measurement read ~3 % slow), then the W warm-up steps of the contract
  t_pre = time.perf_counter()
  while time.perf_counter() - t_pre < args.prewarm_s:
    chain.run(10)
    torch.cuda.synchronize()
  chain.run(args.warmup)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  chain.run(args.steps)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  ms_per_step = elapsed / args.steps * 1e3
  value = world * px / (elapsed / args.steps) / 1e6

  result = {
      'metric': 'Mpixels/s through 8-step filter chain fwd+bwd',
      'value': value,
      'unit': 'Mpixels/s',
      'n_gpus': world,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': ms_per_step,
      'higher_is_better': True,
      'scaling': 'weak',
      'vs_baseline': None,
      'dtype': args.dtype,
      'data': 'synthetic',
      'config': {
          'workload': '8-step filter chain (E,G,W,S+,T,Ct,BW,C) fwd+bwd, %dx%dx%dx3 %s NHWC per GPU, one HIP '
                      'kernel per filter step and direction' % (shape[0], shape[1], shape[2], args.dtype),
          'batch_per_gpu': shape[0],
          'height': shape[1],
          'width': shape[2],
          'parallelism': 'image-sharded replicas x%d (no data-path collective)' % world,
          'launch': ('hipGraph replay, %d steps (%d captured launches) per replay' % (chain.unroll, 17 * chain.unroll)) if chain.graph is not None else 'eager (one C-ABI call per direction)',
          'chain_algorithmic_GBps': 8 * 5 * 3 * esz * px / (elapsed / args.steps) / 1e9 * 1.0,
      },
  }

  if rank == 0 and not args.no_per_kernel:
    per = time_kernels(chain, args.kernel_reps)
    dom = max(per, key=per.get)
    bpp = (2 if dom.startswith('fwd') else 3) * 3 * esz  # algorithmic bytes per pixel per launch
    achieved = bpp * px / (per[dom] * 1e-3) / 1e9
    result['roofline'] = {
        'bound': 'hbm',
        'kernel': dom,
        'achieved': achieved,
        'peak': HBM_PEAK_GBPS,
        'unit': 'GB/s',
        'frac': achieved / HBM_PEAK_GBPS,
        'traffic': load_traffic(dom),
        'avg_launch_ms': per[dom],
        'peak_note': 'spec 8.0 TB/s; frac_of_copy_ceiling is against the 6.29 TB/s default-policy float4 copy of '
                     'MI355X_MICROARCH.md (nt loads + sc1 stores reach 7.0-7.7 TB/s: tools/membench, DESIGN.md 3.1)',
        'frac_of_copy_ceiling': achieved / 6290.0,
        'algorithmic_bytes_per_launch': bpp * px,
        # the whole timed region (16 kernels + 1 fill per step): 240 B/pixel/step over the step time
        'chain_achieved': result['config']['chain_algorithmic_GBps'],
        'chain_frac': result['config']['chain_algorithmic_GBps'] / HBM_PEAK_GBPS,
    }
    result['per_kernel'] = {
        k: {
            'ms': v,
            'GBps': (2 if k.startswith('fwd') else 3) * 3 * esz * px / (v * 1e-3) / 1e9
        } for k, v in per.items()
    }
  barrier()
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    result['cpu_baseline'] = cpu_baseline()
  if rank == 0:
    print(json.dumps(result))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
