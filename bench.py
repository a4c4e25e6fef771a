#!/usr/bin/env python
"""Headline benchmark: Mpixels/s through the 8-step filter chain, forward + backward.

  python bench.py --gpus N --steps K --warmup W          (N > 1: re-launches itself as N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W                  (same thing, external launcher)
  --workload chain|train|infer|allreduce|chain_fused   --scaling weak|strong        (SURVEY.md 8(e) scaling report rows)

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
MFMA_FP32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (v_mfma_f32_*_f32), MI355X_MICROARCH.md
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
  ap.add_argument('--rotating-probe', action='store_true',
                  help='also time the dominant kernel back to back on operands rotating through three buffer sets '
                  '(roofline.avg_launch_ms_rotating_buffers)')
  ap.add_argument('--no-per-kernel', action='store_true')
  ap.add_argument('--no-legs', action='store_true', help='skip the extra legs of the default line (64x64x64 chain, training '
                  'iteration, all-reduce): profiling runs that want the chain kernels only')
  ap.add_argument('--seed', type=int, default=1234)
  ap.add_argument('--order', default='0,1,2,3,4,5,6,7', help='filter ids of the chain steps (experiments only; the '
                  'metric is defined on the cfg.filters order 0..7)')
  ap.add_argument('--workload', default='chain', choices=['chain', 'train', 'infer', 'allreduce', 'chain_fused'],
                  help="chain: the headline filter-chain metric; train: one reference training iteration "
                  "(1 generator/value step + cfg.citers critic steps, net.py:307-365) on 64 images per GPU")
  ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                  help='weak: every rank runs the full batch shape (per-GPU work fixed); strong: the global batch '
                  '(--shape N / cfg.batch_size) is split image-wise over the ranks (total work fixed)')
  ap.add_argument('--dry-run', action='store_true',
                  help='launcher / rendezvous check without a GPU: ranks meet over gloo and rank 0 prints the line '
                  'with value 0 (CPU test of the self-launch path)')
  ap.add_argument('--cold-shape', default='256,512,512',
                  help="chain workload: also time the dominant kernel on tensors of this shape (384 MiB each, beyond "
                  "the 256 MiB Infinity Cache) for roofline.hbm_cold; 'none' disables")
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
    # per-step parameter gradients carved from one flat buffer
    flat = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
    self.dparams, off = [], 0
    for p in self.params:
      self.dparams.append(flat[off:off + p.numel()].view_as(p))
      off += p.numel()

    self.graph = None
    self.unroll = 1
    # 8 fwd + 8 bwd per half-batch stream + 1 finish (all parameter gradients)
    self.streams = _cabi.chain_streams(shape[0], shape[1], shape[2], _cabi._dtype_code(x))
    self.launches_per_step = self.streams * 2 * len(self.ids) + 1

  def launch(self):
    _cabi.chain_fwd(self.ids, self.acts, self.params)
    _cabi.chain_bwd(self.ids, self.acts, self.grads, self.params, self.dparams)

  def capture(self, unroll=1):
    """Capture `unroll` steps (each 8 fwd + 8 bwd + 1 finish launches) into one hipGraph: small shapes
    are bound by the ~5 us host cost per launch, a graph replay pays it once; every replay boundary
    costs ~3 us on the device, so several steps share a replay when the step count allows."""
    self.unroll = unroll
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      self.launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # The chain contains no collective, so RCCL's stream never joins this capture and the NCCL watchdog's event
    # polls stay legal under capture_error_mode='thread_local'; draining it first (exposure_amd.dist) is cheap
    # insurance against a collective of the timing bracket still sitting in its list.
    import torch.distributed as _dist
    if _dist.is_available() and _dist.is_initialized():
      from exposure_amd import dist as xdist
      xdist.drain_before_capture()
    self.graph = torch.cuda.CUDAGraph()
    # thread_local: with a process group alive, RCCL's watchdog thread may poll events while this
    # thread captures; under the default "global" mode such a call aborts the process
    try:
      from exposure_amd.util import capture_without_gc  # no cyclic garbage collection inside a capture
      with capture_without_gc(), torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
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
      # the streaming kernel alone, as it runs inside the chain (whose ONE finish launch per step -- 16 launches
      # + 1 -- is part of ms_per_step, not of any per-kernel figure)
      _cabi.filter_bwd_records(ids[i], chain.acts[i], chain.grads[i + 1], chain.grads[i], chain.params[i])
      if r >= 0:
        ev[r][k][1].record()
      k += 1
  torch.cuda.synchronize()
  out = {}
  for k, name in enumerate(names):
    out[name] = sum(ev[r][k][0].elapsed_time(ev[r][k][1]) for r in range(reps)) / reps
  return out


# filter short name -> (demangled, mangled) spelling of its functor in the kernel names rocprofv3 reports
KERNEL_CLASS = {'E': ('ExposureF,', '9ExposureFE'), 'G': ('GammaF,', '6GammaFE'), 'W': ('WhiteBalanceF,', '13WhiteBalanceFE'),
                'S+': ('SatPlusF,', '8SatPlusFE'), 'T': ('CurveF<1>,', '6CurveFILi1EE'), 'Ct': ('ContrastF,', '9ContrastFE'),
                'BW': ('WnbF,', '4WnbFE'), 'C': ('CurveF<3>,', '6CurveFILi3EE')}


def time_back_to_back(chain, name, reps=60):
  """Average duration (ms) of ONE launch of kernel `name` ('fwd_X' / 'bwd_X') from `reps` launches enqueued back to back
  between ONE HIP-event pair, the operands rotating through three buffer sets (no launch re-reads what its predecessor
  just touched, each writes a tensor of its own).  Per-launch event pairs (time_kernels) read 2.3-4.3 us more than
  rocprofv3's kernel durations, and by how much depends on the box; this figure carries no event overhead, only the
  1.7-1.9 us boundary between two dependent streaming kernels that every launch of the chain pays as well."""
  direction, short = name.split('_', 1)
  step = [i for i, fid in enumerate(chain.ids) if FILTER_NAMES[fid] == short][0]
  fid = chain.ids[step]
  nact = len(chain.acts)
  xs = [chain.acts[(step + k) % nact] for k in range(3)]
  outs = [torch.empty_like(xs[0]) for _ in range(3)]
  dys = [chain.grads[-1], chain.grads[0], chain.grads[1]]
  prm = chain.params[step]

  def launch(k):
    if direction == 'fwd':
      _cabi.filter_fwd(fid, xs[k % 3], outs[k % 3], prm)
    else:
      _cabi.filter_bwd_records(fid, xs[k % 3], dys[k % 3], outs[k % 3], prm)

  for k in range(6):
    launch(k)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for k in range(reps):
    launch(k)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def event_pair_overhead_ms(dev, reps=200):
  """What a HIP-event pair around ONE launch adds to the launch it brackets, measured on a 1-element kernel: the mean
  of `reps` per-launch pairs minus the per-launch time of `reps` launches between one pair (same stream)."""
  t = torch.zeros(1, device=dev)
  for _ in range(20):
    t.add_(1.0)
  pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in pairs:
    a.record()
    t.add_(1.0)
    b.record()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    t.add_(1.0)
  e1.record()
  torch.cuda.synchronize()
  per_pair = sum(a.elapsed_time(b) for a, b in pairs) / reps
  return max(per_pair - e0.elapsed_time(e1) / reps, 0.0)


PROFILE_TABLE_OF_SHAPE = {(64, 512, 512, 3): 'chain', (16, 512, 512, 3): 'chain_B', (64, 64, 64, 3): 'chain_A'}


def rocprof_avg_us(kernel, shape, dtype, prefix=None):
  """Average duration (us) of `kernel` ('bwd_C', ...) in the committed rocprofv3 kernel table of the SAME command on the
  SAME shape and storage dtype (profiles/<round>_final_kernel_stats_chain[_A|_B].csv, newest round first); None when no
  table exists for this workload or none lists the kernel (a duration measured on another shape would be wrong, not
  approximate -- like load_traffic)."""
  import csv
  import glob
  table = PROFILE_TABLE_OF_SHAPE.get(tuple(shape))
  if table is None:
    return None
  direction, short = kernel.split('_', 1)
  cls = KERNEL_CLASS.get(short, ('?', '?'))
  wants = ('filter_%s_kernel<%s' % (direction, cls[0]), 'filter_%s_kernelINS_%s' % (direction, cls[1]))
  is_dtype = (lambda name: 'f16' in name or 'DF16_' in name or '_Float16' in name) if dtype == 'f16' else \
      (lambda name: not ('f16' in name or 'DF16_' in name or '_Float16' in name))
  for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '%s_final_kernel_stats_%s.csv' % (prefix or 'r*', table))),
                     reverse=True):
    try:
      # whole-batch launches only where the table mixes them with the half-batch launches of the two-stream chain: the
      # metric shape's table is filtered to grid_y = 64 when it is made (tools/profile_all.sh), the others hold one kind
      for row in csv.reader(l for l in open(path) if not l.startswith('#')):
        if row and any(w in row[0] for w in wants) and is_dtype(row[0]):
          return {'us': float(row[3]) / 1e3, 'calls': int(row[1]), 'file': os.path.relpath(path, ROOT),
                  'shape': 'x'.join(str(v) for v in shape), 'dtype': dtype}
    except (OSError, ValueError, IndexError):
      continue
  return None


def committed_train_launches():
  """Launches per training iteration from the newest committed rocprofv3 table of the training workload
  (profiles/<round>_final_kernel_stats_train.csv: the timed region's window) -> (count, file) or (None, None)."""
  import glob
  import re
  for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_final_kernel_stats_train.csv')), reverse=True):
    try:
      m = re.search(r'= (\d+) per iteration', open(path).readline())
      if m:
        return int(m.group(1)), os.path.relpath(path, ROOT)
    except OSError:
      continue
  return None, None


def _cpu_chain_rates(name, threads, budget_s=4.0):
  """Best-of-5-after-2-warm-ups rates (Mpixels/s) of the torch-CPU restatement on SHAPES[name] with
  `threads` intra-op threads; a configuration that needs more than `budget_s` per kind is cut short."""
  from oracle import filters_torch as ft
  shape = synthetic.SHAPES[name]
  x, dy, params = synthetic.make_case(1234, shape, np.float16)
  tx = torch.from_numpy(x.astype(np.float32))
  tdy = torch.from_numpy(dy.astype(np.float32))
  tp = [torch.from_numpy(p) for p in params]
  px = shape[0] * shape[1] * shape[2]

  def fwd_only():
    with torch.no_grad():
      cur = tx
      for fid, p in enumerate(tp):
        cur = ft.process_packed(fid, cur, p)
    return cur

  torch.set_num_threads(threads)
  best = {}
  for kind, fn in (('fwd_bwd', lambda: ft.chain_fwd_bwd(tx, tp, tdy)), ('fwd', fwd_only)):
    times, t_cfg = [], time.perf_counter()
    for it in range(7):
      t0 = time.perf_counter()
      fn()
      times.append(time.perf_counter() - t0)
      if time.perf_counter() - t_cfg > budget_s and len(times) >= 2:
        break
    best[kind] = px / min(times[min(2, len(times) - 1):]) / 1e6
  return best


def _c_port_rates(name, threads, budget_s=3.0):
  """Best-of-5-after-2-warm-ups rate (Mpixels/s, fwd+bwd) of the C / OpenMP restatement (oracle/filters_c.c,
  float32 -- the reference's dtype -- one fused pass per step and direction) on SHAPES[name], in its own
  process (`python -m oracle.filters_c`): the OpenMP runtime is configured through the environment before it
  starts, and a team left spinning by one thread count cannot disturb the next."""
  import subprocess
  env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY='passive', OMP_PROC_BIND='spread',
             OMP_PLACES='cores' if threads <= (os.cpu_count() or 1) // 2 else 'threads', PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, '-m', 'oracle.filters_c', name, str(threads), str(budget_s)], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=60)
  if out.returncode != 0:
    raise RuntimeError('oracle.filters_c worker failed: %s' % out.stderr.strip()[-300:])
  return json.loads(out.stdout.strip().splitlines()[-1])['Mpixels_per_s']


def cpu_baseline():
  """The CPU side of the comparison, on the host cores, same synthetic workload, best of 5 after 2 warm-ups
  (BASELINE.md section 3), two restatements (both `kind: "port"` -- TF-1 itself cannot run here):

  * `value`: the C / OpenMP restatement oracle/filters_c.c in float32 (the reference's dtype), one fused pass per
    step and direction like the HIP chain, swept over thread counts up to every logical CPU -- what a tuned CPU
    implementation of this path does on this host;
  * `op_by_op`: the torch-CPU fp32 op-by-op restatement (oracle/filters_torch.py, autograd backward) -- the
    granularity at which TF-1 executes the reference graph (one pass over the image per elementwise op) -- on
    shapes A (64x64x64x3) and B (16x512x512x3).
  ~20-30 s of CPU work in total."""
  import subprocess
  try:
    ncpu = len(os.sched_getaffinity(0))  # the CPUs this process may run on, not the host's total
  except (AttributeError, OSError):
    ncpu = os.cpu_count() or 1
  model = ''
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        model = line.split(':', 1)[1].strip()
        break
  except OSError:
    pass
  t_start = time.perf_counter()
  # ---- C / OpenMP port
  c_sweep = sorted({t for t in (ncpu // 8, ncpu // 4, ncpu // 2, ncpu) if t >= 1})
  c_port = {'B': {}, 'A': {}}
  try:
    for threads in c_sweep:
      c_port['B'][str(threads)] = _c_port_rates('B', threads)
      print('cpu_baseline: C port, shape B, %d threads: %.0f Mpixels/s fwd+bwd' % (threads, c_port['B'][str(threads)]),
            file=sys.stderr)
    best_c = int(max(c_port['B'], key=c_port['B'].get))
    c_port['A'][str(best_c)] = _c_port_rates('A', best_c)
  except (OSError, RuntimeError, ValueError, subprocess.SubprocessError) as e:  # no C oracle: op-by-op figure only
    print('cpu_baseline: C port unavailable (%s)' % e, file=sys.stderr)
    c_port, best_c = None, None
  # ---- torch op-by-op
  sweep = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)})
  by_shape = {}
  for name in ('A', 'B'):
    # shape A (0.26 Mpixel) sweeps every thread count; shape B (4.2 Mpixel, ~1.5 s per pass) only the two best of A
    counts = sweep if name == 'A' else sorted(by_shape['A']['sweep_fwd_bwd'], key=by_shape['A']['sweep_fwd_bwd'].get)[-2:]
    rows = {}
    for threads in [int(t) for t in counts]:
      rows[threads] = _cpu_chain_rates(name, threads, budget_s=2.0 if name == 'B' else 3.0)
      print('cpu_baseline: op-by-op, shape %s, %d threads: %.1f Mpixels/s fwd+bwd (%.1f s so far)' %
            (name, threads, rows[threads]['fwd_bwd'], time.perf_counter() - t_start), file=sys.stderr)
    bt = max(rows, key=lambda t: rows[t]['fwd_bwd'])
    by_shape[name] = {
        'shape': 'x'.join(str(v) for v in synthetic.SHAPES[name]),
        'best_threads': bt,
        'fwd_bwd_Mpixels_per_s': rows[bt]['fwd_bwd'],
        'fwd_Mpixels_per_s': rows[bt]['fwd'],
        'sweep_fwd_bwd': {str(t): r['fwd_bwd'] for t, r in rows.items()},
    }
  b = by_shape['B']
  host = '%s with %d logical CPUs' % (model or 'unknown CPU', ncpu)
  if c_port is None:
    value, cores = b['fwd_bwd_Mpixels_per_s'], b['best_threads']
    sample = ('torch fp32 op-by-op restatement (autograd backward; never TF1) on host %s: 8-step chain fwd+bwd on '
              '16x512x512x3, best of 5 after 2 warm-ups, best thread count of %s' % (host, sweep))
  else:
    value, cores = c_port['B'][str(best_c)], best_c
    sample = ('C / OpenMP restatement oracle/filters_c.c (float32, one fused pass per step and direction; never TF1) on '
              'host %s: 8-step chain fwd+bwd on 16x512x512x3 (4.19 Mpixel per pass), best of 5 after 2 warm-ups, best '
              'thread count of %s = %d; op_by_op = the torch fp32 op-by-op restatement (TF-1\'s execution granularity) '
              'on 64x64x64x3 and 16x512x512x3' % (host, c_sweep, best_c))
  return {
      'value': value,
      'unit': 'Mpixels/s',
      'cores': cores,
      'kind': 'port',
      'sample': sample,
      'c_port_sweep': c_port,
      'op_by_op': by_shape,
      'seconds': time.perf_counter() - t_start,
  }


def parity_sample(dev):
  """Part of the cpu_baseline leg (the oracle is the CHECKER here, after the timed region): the 8-step chain fwd+bwd on
  64x64x64x3 fp16 through the HIP library against the float64 restatement evaluated on the inputs each launch read.
  Reports the largest deviation on values below 2.0 -- where north_star's plain 1e-3 per-pixel bound applies to fp16
  storage -- and overall (fp16 storage above 2.0 is itself coarser than 1e-3: bound 1e-3 + half an fp16 ulp)."""
  try:
    from oracle import filters_c as orc
    orc.process_packed(0, np.zeros((1, 2, 2, 3)), np.zeros((1, 1)))
    which = 'oracle/filters_c.c (float64)'
  except Exception:  # no C build on this host: the NumPy restatement
    from oracle import filters_np as orc
    which = 'oracle/filters_np.py (float64)'
  shape = synthetic.SHAPES['A']
  x, dy, params = synthetic.make_case(4321, shape, np.float16)
  acts = [torch.from_numpy(x).to(dev)] + [torch.empty(shape, dtype=torch.float16, device=dev) for _ in range(8)]
  prm = [torch.from_numpy(p).to(dev) for p in params]
  grads = [torch.empty(shape, dtype=torch.float16, device=dev) for _ in range(8)] + [torch.from_numpy(dy).to(dev)]
  dprm = [torch.empty_like(p) for p in prm]
  _cabi.chain_fwd(list(range(8)), acts, prm)
  _cabi.chain_bwd(list(range(8)), acts, grads, prm, dprm)
  torch.cuda.synchronize()
  worst, worst_lt2, count, ok, dp_ok, dp_worst = 0.0, 0.0, 0, True, True, {}
  for i in range(8):
    xin = acts[i].cpu().numpy().astype(np.float64)
    gin = grads[i + 1].cpu().numpy().astype(np.float64)
    p64 = params[i].astype(np.float64)
    ry = orc.process_packed(i, xin, p64)
    if which.startswith('oracle/filters_c'):
      rdx, rdp, adp = orc.backward_packed(i, xin, p64, gin, with_abs=True)
    else:
      rdx, rdp = orc.backward_packed(i, xin, p64, gin)
      adp = orc.param_grad_abs(i, xin, p64, gin)
    # parameter gradients (sums over H*W*3): error relative to A, the sum of the absolute terms; bound 1e-4 |ref| + 2e-6 A
    dp_err = np.abs(dprm[i].cpu().numpy().astype(np.float64) - rdp)
    dp_ok = dp_ok and bool((dp_err <= 1e-4 * np.abs(rdp) + 2e-6 * adp).all())
    dp_worst[FILTER_NAMES[i]] = {'err_over_A': float((dp_err / adp).max()),
                                 'err_over_ref': float((dp_err / np.maximum(np.abs(rdp), 1e-300)).max()),
                                 'min_ref_over_A': float((np.abs(rdp) / adp).min())}
    for got, ref in ((acts[i + 1], ry), (grads[i], rdx)):
      ref = np.clip(ref, -65504.0, 65504.0)
      err = np.abs(got.float().cpu().numpy().astype(np.float64) - ref)
      ok = ok and bool((err <= 1e-3 + np.abs(ref) * 2.0**-11).all())
      worst = max(worst, float(err.max()))
      small = np.abs(ref) < 2.0
      worst_lt2 = max(worst_lt2, float(err[small].max()))
      count += err.size
  return {
      'checker': which,
      'workload': '8-step chain fwd+bwd, %s fp16, every value of all 16 launches' % 'x'.join(str(v) for v in shape),
      'values_checked': count,
      'max_abs_err_values_below_2': worst_lt2,
      'bound_below_2': 1e-3,
      'max_abs_err_all_values': worst,
      'bound_all_values': '1e-3 + |ref| * 2^-11 (half an fp16 ulp of the stored value)',
      'dparams': dp_worst,
      'dparams_bound': '|err| <= 1e-4 |ref| + 2e-6 A, A = sum of the absolute per-element terms of each parameter gradient',
      'dparams_values_checked': int(sum(p.size for p in params)),
      'within_bounds': ok and worst_lt2 <= 1e-3 and dp_ok,
  }


def load_traffic(kernel, shape, dtype):
  """Per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/traffic.json), keyed by
  the workload they were collected on ("64x512x512x3:f16"); None when no PMC pass exists for THIS
  shape / dtype (a number measured on another shape would be wrong, not approximate)."""
  path = os.path.join(ROOT, 'profiles', 'traffic.json')
  key = '%s:%s' % ('x'.join(str(v) for v in shape), dtype)
  try:
    return json.load(open(path)).get(key, {}).get(kernel)
  except (OSError, ValueError, AttributeError):
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


def trace(msg):
  """EXPO_TRACE=1: phase markers on stderr (soak runs: where does a rank die?)."""
  if os.environ.get('EXPO_TRACE') == '1':
    print('[trace rank %s] %s' % (os.environ.get('RANK', '0'), msg), file=sys.stderr, flush=True)


def bracket(fn, dist, dev):
  """The timing bracket of the contract: barrier + synchronize on both sides, MAX over ranks (seconds)."""
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  fn()
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  return elapsed


def conv_stack_macs(c_in, size=64, base=32, out_ch=None):
  """MACs per image of one forward pass of the reference's conv stack (4x4 / stride 2 until 4x4; agent.py:11-37 ends on
  output_dim / 16 channels, critics.py:6-38 keeps doubling) + its FC 4096 -> 128: [(macs, has_input_grad)] per layer."""
  layers, ch, prev, sz = [], base, c_in, size // 2
  while True:
    last = sz == 4
    co = (out_ch if (last and out_ch) else ch)
    layers.append(sz * sz * co * 16 * prev)
    prev = co
    if last:
      break
    ch *= 2
    sz //= 2
  layers.append(prev * 16 * 128)  # FC on the flattened 4x4 map
  return layers


def train_flops_per_iteration(cfg, n):
  """Analytic GEMM flops (2 x MACs) of ONE training iteration on n images per GPU (net.py:307-365): one generator /
  value step + cfg.citers critic steps.  A layer's forward costs its MACs once; a backward costs them once per wanted
  gradient (data, weight); the gradient penalty's double backward costs every critic layer twice more (the tangent pass
  t_l = F(t_{l-1}, W_l) through the data-gradient graph and the weight gradient G(t_{l-1}, u_l): critic_direct.py; rounds 3-5
  counted three, one forward pass of n images per critic step too many -- 7 % of the iteration's total).  First layers have
  no data gradient except where the gradient reaches the IMAGE (critic / value nets on generated images, the penalty)."""
  trunk = conv_stack_macs(3 + cfg.num_state_dim, out_ch=cfg.feature_extractor_dims // 16)
  crit = conv_stack_macs(3 + 3)
  val = conv_stack_macs(3 + cfg.num_state_dim + 3)
  heads = sum(128 * (f_np + 6) for f_np in (1, 1, 3, 1, 8, 1, 1, 24)) + 7 * 4096 * 128 + 128 * 8  # 8 fc2 + 7 more fc1 + selector fc2
  fwd = lambda net: sum(net)
  wgrad = lambda net: sum(net)
  dgrad_to_image = lambda net: sum(net)
  dgrad_inner = lambda net: sum(net[1:])
  g = 2 * (fwd(trunk) + wgrad(trunk) + dgrad_inner(trunk)) + 3 * heads  # two trunks, all their gradients; FC heads x3
  g += 2 * fwd(crit) + dgrad_to_image(crit)  # critic(fake_output) with the image gradient, critic(fake_input) forward only
  g += 2 * fwd(val) + wgrad(val) + dgrad_inner(val) + dgrad_to_image(val)  # old_value (theta_v), new_value (image)
  c = 3 * fwd(crit) + 2 * (wgrad(crit) + dgrad_inner(crit))  # real + fake + interpolated forward; emd backward on 2n
  c += dgrad_to_image(crit) + fwd(crit) + wgrad(crit)  # penalty: d D / d x^, then its double backward (two passes per layer)
  return 2.0 * n * (g + cfg.citers * c)


def measure_train(args, world, rank, dev, dist, steps, warmup):
  """BASELINE configs 3/4: agent rollout step + policy CNN + WGAN-GP critic, batch 64 per GPU, random-init weights,
  synthetic FiveK-shaped inputs, gradients all-reduced over RCCL.  Returns the measurement as a dict (every rank)."""
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  cfg = make_cfg()
  torch.manual_seed(args.seed)  # identical initial weights on every rank
  gan = GAN(cfg, device=dev, use_graphs=(args.graph != 'off'), seed=args.seed)
  # weak: cfg.batch_size (64) images per GPU; strong: the reference's global batch of 64 split image-wise
  n = local_shape((cfg.batch_size,), world, args.scaling)[0]
  from exposure_amd.replay_memory import ReplayMemory, ResidentProvider
  pool_dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  # the synthetic RAW / retouched data sets are resident in HBM (4 096 proxies each, generated before the timed region);
  # batches are views of them and the replay memory's records are gathered straight into the step graphs' inputs
  memory = ReplayMemory(cfg, ResidentProvider(dev, dtype=pool_dtype, seed=args.seed + 10 * rank + 1),
                        ResidentProvider(dev, gamma=1.0, dtype=pool_dtype, seed=args.seed + 10 * rank + 2),
                        seed=args.seed + rank)
  # net.py:320-328: the first iteration rolls the generator with lr_g = 0 until terminated
  # trajectories exist for the critic to replay (100 steps in the reference; 8 suffice: 5 steps end one)
  for _ in range(8):
    feed, feats = memory.get_feed_dict_and_states(n, lazy=True)
    out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
    memory.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)

  def iteration(it):
    # one G / V step, its results back into the replay memory, cfg.citers critic steps: planned ahead on the host and
    # replayed as ONE hipGraph fed by one host-to-device copy (GAN.train_iteration); --graph off: the step-by-step calls
    return gan.train_iteration(memory, it, progress=it / cfg.max_iter_step, batch_size=n)['g']

  trace('pool primed')
  for i in range(warmup):
    iteration(i + 1)
    trace('warm-up iteration %d done' % i)

  def timed():
    for i in range(steps):
      iteration(i + 1)

  trace('timed region starts')
  elapsed = bracket(timed, dist, dev)
  trace('timed region done')
  ms = elapsed / steps * 1e3
  flops = train_flops_per_iteration(cfg, n)
  achieved = flops / (ms * 1e-3) / 1e12
  return {
      'ms_per_iteration': ms,
      'images_per_s': world * n / (elapsed / steps),
      'steps': steps,
      'warmup': warmup,
      'batch_per_gpu': n,
      'critic_steps_per_iteration': cfg.citers,
      'launch': ('one hipGraph replay per iteration, fed by one host-to-device copy of the iteration plan'
                 if any(k[0] == 'it' for k in gan._graphs) else
                 'one hipGraph replay per G/V step and per critic step' if gan._replay_steps else 'eager'),
      'capture_drain_verified': getattr(gan, 'capture_drain_verified', None),
      'roofline': {
          'bound': 'mfma_fp32',
          'flops_per_iteration': flops,
          'flops_note': 'analytic GEMM flops of the conv stacks (agent.py:11-37, critics.py:6-38) and FC layers for every '
                        'forward / data-gradient / weight-gradient pass of the iteration incl. the penalty\'s double '
                        'backward (bench.py::train_flops_per_iteration); element-wise work not counted',
          'achieved': achieved,
          'peak': MFMA_FP32_PEAK_TFLOPS,
          'unit': 'TFLOP/s',
          'frac': achieved / MFMA_FP32_PEAK_TFLOPS,
          'note': 'the nets are fp32 like the reference\'s; every convolution primitive (forward + bias + lrelu, data gradient '
                  'with the activation gradient in its epilogue, weight gradient with the bias sums) runs on the in-house '
                  'kernels (csrc/conv_ops.hip, v_mfma_f32_32x32x2_f32; 6 / 17-plane data gradients on the vector ALUs); the '
                  'critic update is a hand-scheduled launch sequence over [real | fake | interpolated] as one batch '
                  '(exposure_amd/critic_direct.py), the G / V step runs each convnet stack as one autograd node; no MIOpen '
                  'kernel remains (DESIGN.md 3.12)',
      },
  }


def train_cpu_baseline(steps=2):
  """The same iteration (same modules, fp32) on the host cores through torch's CPU kernels, with the filter step
  through the oracle-backed mock of the C-ABI (tests/_fake_hip.py: the checker standing in for the library -- a CPU
  baseline, never shipped).  Bounded sample: one warm-up + `steps` iterations at batch 64."""
  try:
    from exposure_amd.config import make_cfg
    from exposure_amd.gan import GAN
    from exposure_amd.replay_memory import ReplayMemory, SyntheticProvider
    from tests._fake_hip import fake_hip
  except Exception as e:
    return {'error': str(e)[:200]}
  t_start = time.perf_counter()
  cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  threads = min(cores, 64)
  torch.set_num_threads(threads)
  cfg = make_cfg()
  torch.manual_seed(0)
  cpu = torch.device('cpu')
  try:
    with fake_hip():
      gan = GAN(cfg, device=cpu, use_graphs=False, seed=0)
      n = cfg.batch_size
      memory = ReplayMemory(cfg, SyntheticProvider(cpu, dtype=torch.float32, seed=1),
                            SyntheticProvider(cpu, gamma=1.0, dtype=torch.float32, seed=2), seed=0)
      for _ in range(6):
        feed, feats = memory.get_feed_dict_and_states(n)
        out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], 0.0, it=0)
        memory.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)

      def iteration(it):
        feed, feats = memory.get_feed_dict_and_states(n)
        out = gan.generator_step(feed['fake_input'], feed['z'], feed['states'], progress=0.1, it=it)
        memory.replace_memory(out['fake_output'], out['new_states'], feats, advanced=True)
        for _ in range(cfg.citers):
          rep = memory.get_replay_feed_dict(n)
          gan.critic_step(rep['real_data'], rep['fake_output'], it=it)

      iteration(1)
      times = []
      for i in range(steps):
        t0 = time.perf_counter()
        iteration(i + 2)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 40:
          break
  except Exception as e:  # a baseline, not a reason to lose the line
    return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
  return {'value': n / min(times), 'unit': 'images/s', 'ms_per_iteration': min(times) * 1e3, 'cores': threads,
          'kind': 'port', 'sample': '%d iterations at batch %d after one warm-up (best), torch CPU kernels, fp32; filter '
          'step through the oracle' % (len(times), n), 'seconds': time.perf_counter() - t_start}


def run_train(args, world, rank, dev, dist):
  m = measure_train(args, world, rank, dev, dist, args.steps, args.warmup)
  if rank == 0:
    n = m['batch_per_gpu']
    line = {
        'metric': 'generator-step images/s (1 G/V step + %d critic steps per iteration)' % m['critic_steps_per_iteration'],
        'value': m['images_per_s'],
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': m['ms_per_iteration'],
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': '%s images / f32 nets' % args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': 'reference training iteration (net.py:307-365): agent rollout step, policy CNN, value net, '
                        'WGAN-GP critic; batch %d x 64x64x3 per GPU; random-init weights' % n,
            'global_batch': world * n,
            'parallelism': 'dp%d image-sharded; flat gradient buckets (theta_g heads / trunks, theta_v, theta_c) '
                           'all-reduced over RCCL from backward hooks' % world,
            'launch': m['launch'],
            'capture_drain_verified': m['capture_drain_verified'],
            'reference_note': 'README.md:43: ~0.30 s/iteration on a GTX 1080 Ti (whole run ~100 min / 20000 it)',
        },
        'roofline': m['roofline'],
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = train_cpu_baseline()
    print(json.dumps(line), flush=True)
  trace('line printed')
  if dist is not None:
    dist.barrier()
    trace('final barrier done')
    dist.destroy_process_group()
    trace('process group destroyed')


def measure_allreduce(args, world, rank, dev, dist, steps, warmup):
  """SURVEY.md section 8(e) "all-reduce only": the flat fp32 gradient buckets of one training iteration
  (theta_v 4.9 MB, theta_g heads 18.9 MB + trunks 5.6 MB once, theta_c 4.9 MB x citers) reduced over RCCL, no
  compute.  With one rank the collective degenerates to nothing and the rate is 0."""
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  cfg = make_cfg()
  torch.manual_seed(args.seed)
  gan = GAN(cfg, device=dev, use_graphs=False)
  sizes = {name: b.numel for name, b in gan.buckets.items()}
  bufs = {k: torch.zeros(v, dtype=torch.float32, device=dev) for k, v in sizes.items()}
  del gan

  def iteration():
    if dist is None:
      return
    dist.all_reduce(bufs['v'])
    dist.all_reduce(bufs['g_head'])
    dist.all_reduce(bufs['g_trunk'])
    for _ in range(cfg.citers):
      dist.all_reduce(bufs['c'])

  for _ in range(warmup):
    iteration()

  def timed():
    for _ in range(steps):
      iteration()

  elapsed = bracket(timed, dist, dev)
  sizes['g'] = sizes['g_head'] + sizes['g_trunk']
  nbytes = 4 * (sizes['g'] + sizes['v'] + cfg.citers * sizes['c'])
  algo = (nbytes / 1e9) / (elapsed / steps) if world > 1 else 0.0
  return {
      'ms_per_iteration': elapsed / steps * 1e3,
      'bytes_per_iteration': nbytes,
      'collectives_per_iteration': 3 + cfg.citers,
      'algorithm_GBps': algo,  # bucket bytes / time
      'bus_GBps': algo * 2.0 * (world - 1) / world,  # ring all-reduce: each rank moves 2 (p-1)/p of the bytes over its links
      'elements': {'theta_g': sizes['g'], 'theta_v': sizes['v'], 'theta_c': sizes['c'], 'citers': cfg.citers},
      'steps': steps,
      'warmup': warmup,
  }


def run_allreduce(args, world, rank, dev, dist):
  m = measure_allreduce(args, world, rank, dev, dist, args.steps, args.warmup)
  if rank == 0:
    e = m['elements']
    print(json.dumps({
        'metric': 'gradient all-reduce GB/s per training iteration (bucket bytes / time)',
        'value': m['algorithm_GBps'],
        'unit': 'GB/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': m['ms_per_iteration'],
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': 'all-reduce only: theta_g %d (2 buckets) + theta_v %d + %d x theta_c %d fp32 elements per iteration'
                        % (e['theta_g'], e['theta_v'], e['citers'], e['theta_c']),
            'bytes_per_iteration': m['bytes_per_iteration'],
            'bus_GBps': m['bus_GBps'],
            'parallelism': 'dp%d, flat buckets over RCCL' % world,
        },
    }))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def measure_shape(name, args, world, rank, dev, dist, steps=200, warmup=20):
  """The 8-step chain fwd+bwd on another of north_star's shapes (64x64x64x3: BASELINE config 2), hipGraph replay,
  same bracket: every rank runs its own batch, value = units of all ranks / slowest rank's time."""
  shape = synthetic.SHAPES[name]
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  chain = Chain(shape, dtype, dev, args.seed + 1000 + rank, [int(v) for v in args.order.split(',')])
  if args.graph != 'off':
    chain.capture(10)
  chain.run(warmup)
  elapsed = bracket(lambda: chain.run(steps), dist, dev)
  px = shape[0] * shape[1] * shape[2]
  esz = 2 if args.dtype == 'f16' else 4
  return {
      'ms_per_step': elapsed / steps * 1e3,
      'Mpixels_per_s': world * px / (elapsed / steps) / 1e6,
      'steps': steps,
      'chain_algorithmic_GBps_per_gpu': 8 * 5 * 3 * esz * px / (elapsed / steps) / 1e9,
      'note': 'launch-latency bound at this size (17 launches on 1.5 MB tensors): Mpixels/s only, no roofline claim'
              if px * 3 * esz < (8 << 20) else 'tensors inside the 256 MiB Infinity Cache',
  }


def run_legs(result, args, world, rank, dev, dist):
  """Outside the timed region, after the headline measurement: the other numbers north_star names, from the SAME
  driver-run command -- the 64x64x64x3 chain (config 2), one training iteration (configs 3 / 4: ms, images/s, whether the
  hipGraph capture of the collectives was verified) and, for N > 1, the gradient buckets' all-reduce alone (bus GB/s).
  A watchdog thread prints the line as it stands and exits if a leg hangs (a collective on hardware this code has not
  met must not cost the headline); a leg that raises is recorded as {'error': ...}."""
  import threading
  legs = {}
  result['legs'] = legs
  state = {'leg': None}
  deadline = float(os.environ.get('EXPO_BENCH_LEGS_TIMEOUT_S', '300'))

  def fire():
    if rank == 0:
      legs['error'] = 'watchdog: leg %r still running after %.0f s; line printed without it' % (state['leg'], deadline)
      print(json.dumps(result), flush=True)
    os._exit(0)

  timer = threading.Timer(deadline, fire)
  timer.daemon = True
  timer.start()

  def leg(name, fn):
    state['leg'] = name
    t0 = time.perf_counter()
    try:
      legs[name] = fn()
      legs[name]['leg_seconds'] = time.perf_counter() - t0
    except Exception as e:
      legs[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    torch.cuda.empty_cache()

  try:
    leg('chain_64x64x64x3', lambda: measure_shape('A', args, world, rank, dev, dist))
    leg('train', lambda: measure_train(args, world, rank, dev, dist, steps=20, warmup=5))
    if world > 1:
      leg('allreduce', lambda: measure_allreduce(args, world, rank, dev, dist, steps=20, warmup=5))
  finally:
    timer.cancel()
  if rank == 0 and world == 1 and not args.no_cpu_baseline and 'error' not in legs.get('train', {'error': 1}):
    legs['train']['cpu_baseline'] = train_cpu_baseline()
  legs['note'] = ('measured after the timed region with the same barrier + synchronize bracket and MAX over ranks; '
                  'whole-job aggregates (images/s, Mpixels/s over all ranks)')


def run_infer(args, world, rank, dev, dist):
  """BASELINE config 5: high-resolution inference, 16x512x512x3 fp16, the 8 filters of cfg.filters
  applied to every image -- (a) one kernel per step (8 reads + 8 writes of the image), (b) the fused
  multi-step forward (1 read + 1 write).  Forward only."""
  shape = local_shape(parse_shape(args.shape or 'B'), world, args.scaling)
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

  # HIP-event average of the fused launch alone (the wall-clock figure above includes the host's launch cadence)
  def event_avg(fn, reps=200):
    for _ in range(10):
      fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

  t_fused = timed(lambda: _cabi.chain_fused_fwd(ids, p24, x, y))
  t_kernel = event_avg(lambda: _cabi.chain_fused_fwd(ids, p24, x, y))
  t_steps = timed(lambda: _cabi.chain_fwd(list(range(8)), acts, params))
  same = float((y.float() - acts[8].float()).abs().max())
  # The fused kernel is VALU-bound, not HBM-bound: 8 filter bodies on a pixel group that stays in registers.  Its
  # ceiling is the SIMDs' issue rate: VALU wave-instructions per 8-pixel group (tools/isa_hist.py on
  # chain_fused.hip, round 3: 2 166 static VALU for the two unrolled steps of the loop = 1 083 per pass over all
  # case bodies, 92 of them transcendental) at the issue cost tools/valubench measured on gfx950
  # (profiles/r02_valubench.txt: 4.5 clocks per wave instruction, 8.2 for v_exp / v_log / v_rcp / v_sin), 1024 SIMDs
  # at 2.4 GHz.  `achieved` / `peak` are in G wave-instructions/s.
  # (-60: the Level body, compiled in but not part of the 8-filter sequence)
  valu_per_group, transc_per_group = 1083.0 - 60.0, 92.0
  clk_per_group = (valu_per_group - transc_per_group) * 4.5 + transc_per_group * 8.2
  groups = px / (8.0 if args.dtype == 'f16' else 4.0) / 64.0  # wave iterations
  peak_ginstr = 1024 * 2.4e9 / (clk_per_group / valu_per_group) / 1e9
  ach_ginstr = groups * valu_per_group / t_kernel / 1e9
  if rank == 0:
    print(json.dumps({
        'roofline': {
            'bound': 'valu', 'kernel': 'chain_fused_fwd_kernel', 'achieved': ach_ginstr, 'peak': peak_ginstr,
            'unit': 'G wave-instr/s', 'frac': ach_ginstr / peak_ginstr, 'avg_launch_ms': t_kernel * 1e3,
            'valu_per_8px_group': valu_per_group, 'issue_clocks_per_group': clk_per_group,
            'issue_floor_ms': groups * clk_per_group / (1024 * 2.4e9) * 1e3,
            'hbm_GBps_actual_traffic': 2 * 3 * esz * px / t_kernel / 1e9,
            'hbm_frac_of_8TBps': 2 * 3 * esz * px / t_kernel / 1e9 / HBM_PEAK_GBPS,
            'note': 'VALU-issue ceiling, not HBM: one read + one write of the image for all 8 steps (12 B/px at fp16)',
        },
        'metric': 'Mpixels/s through 8-step filter chain fwd (high-res inference)',
        'value': world * px / t_fused / 1e6,
        'unit': 'Mpixels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': t_fused * 1e3,
        'higher_is_better': True,
        'scaling': args.scaling,
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


def run_chain_fused(args, world, rank, dev, dist):
  """A benchmark construct beside the headline, never instead of it: the same 8-step sequence forward + backward with
  the parameters FIXED, so that each direction is ONE pass (expo_chain_fused_fwd: read x, write y; expo_chain_fused_bwd:
  read x and dy, write dx, activations recomputed in registers) -- 30 B/pixel of HBM traffic instead of the 240 B/pixel
  of the 16 per-step launches.  The reference has no caller for this (its step k+1 parameters depend on image k through
  the CNN), which is why the headline stays the per-step chain.  Quoted against the same 240 B/pixel."""
  shape = local_shape(parse_shape(args.shape or 'C'), world, args.scaling)
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  esz = 2 if args.dtype == 'f16' else 4
  x, dy, params = make_device_case(shape, dtype, dev, args.seed + rank)
  n = shape[0]
  px = shape[0] * shape[1] * shape[2]
  ids = torch.arange(8, dtype=torch.int32, device=dev)[None, :].repeat(n, 1).contiguous()
  p24 = torch.zeros((n, 8, 24), dtype=torch.float32, device=dev)
  for fid in range(8):
    p24[:, fid, :params[fid].shape[1]] = params[fid]
  y, dx, dp24 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(p24)
  ws = _cabi.reserve_workspace(dev, _cabi.workspace_bytes(n, shape[1], shape[2], 0 if args.dtype == 'f16' else 1, 8))

  def step():
    _cabi.chain_fused_fwd(ids, p24, x, y)
    _cabi.chain_fused_bwd(ids, p24, x, dy, dx, dp24, workspace=ws)

  def event_avg(fn, reps=50):
    for _ in range(5):
      fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  torch.cuda.synchronize()
  light_barrier(dist, dev)
  torch.cuda.synchronize()
  el = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
  t_step = el / args.steps
  t_fwd = event_avg(lambda: _cabi.chain_fused_fwd(ids, p24, x, y))
  t_bwd = event_avg(lambda: _cabi.chain_fused_bwd(ids, p24, x, dy, dx, dp24, workspace=ws))
  # the per-step construction on the same tensors (the headline's 16 launches + finish), for the ratio
  acts = [x] + [torch.empty_like(x) for _ in range(8)]
  grads = [torch.empty_like(x) for _ in range(8)] + [dy]
  dprm = [torch.empty_like(q) for q in params]

  def per_step():
    _cabi.chain_fwd(list(range(8)), acts, params)
    _cabi.chain_bwd(list(range(8)), acts, grads, params, dprm, workspace=ws)

  t_per_step = event_avg(per_step, reps=20)
  worst = 0.0
  for fid in range(8):
    a, b = dp24[:, fid, :params[fid].shape[1]], dprm[fid]
    # relative to the gradient's own scale sum |dy| (the bound of its accumulated rounding, tests/_tol.py)
    scale = torch.maximum(b.abs(), dy.float().abs().sum(dim=(1, 2, 3))[:, None].expand_as(b))
    worst = max(worst, float(((a - b).abs() / scale).max()))
  if rank == 0:
    actual = (2 + 3) * 3 * esz * px  # x, y | x, dy, dx
    print(json.dumps({
        'roofline': {
            'bound': 'valu', 'kernel': 'chain_fused_bwd_kernel', 'avg_launch_ms': t_bwd * 1e3,
            'achieved': 3 * 3 * esz * px / t_bwd / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': 3 * 3 * esz * px / t_bwd / 1e9 / HBM_PEAK_GBPS, 'traffic': None,
            'note': 'actual traffic of the one-pass backward (18 B/px at fp16) against HBM; the kernel is VALU-bound '
                    '(8 forward + 8 backward filter bodies per pixel group), so this fraction is not a quality measure',
        },
        'metric': 'Mpixels/s through 8-step filter chain fwd+bwd, one pass each way (fixed parameters)',
        'value': world * px / t_step / 1e6,
        'unit': 'Mpixels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': t_step * 1e3,
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': 'fused 8-step chain forward + backward (expo_chain_fused_fwd / _bwd), %dx%dx%dx3 %s per GPU; '
                        'a benchmark construct beside the per-step headline' % (shape[0], shape[1], shape[2], args.dtype),
            'fused_fwd_ms': t_fwd * 1e3, 'fused_bwd_ms': t_bwd * 1e3,
            'per_step_chain_ms': t_per_step * 1e3, 'speedup_vs_per_step': t_per_step / (t_fwd + t_bwd),
            'algorithmic_GBps_at_240B_per_pixel': 8 * 5 * 3 * esz * px / t_step / 1e9,
            'actual_traffic_GBps': actual / t_step / 1e9,
            'dparams_max_diff_vs_per_step_rel_to_scale': worst,
        },
    }))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def self_launch(args):
  """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this same command
  (one process per GPU, `torch.distributed.run`, rendezvous on 127.0.0.1) and return their exit code.  Rank 0
  of the children prints the ONE JSON line; it passes straight through to our stdout."""
  import socket
  import subprocess
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC (RCCL across processes on this driver)
  env.setdefault('OMP_NUM_THREADS', '8')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def run_dry(args, world, rank):
  """--dry-run: no GPU is touched.  The ranks meet over gloo, agree on the slowest rank's (empty) timed
  region exactly like the real path does, and rank 0 prints the line -- the CPU test of the launcher."""
  import torch.distributed as dist
  if world > 1 or 'LOCAL_RANK' in os.environ:
    dist.init_process_group('gloo')
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(t.item()) == world - 1
    dist.barrier()
  if rank == 0:
    print(json.dumps({
        'metric': 'Mpixels/s through 8-step filter chain fwd+bwd', 'value': 0.0, 'unit': 'Mpixels/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 0.0, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'dry_run': True,
        'config': {'workload': 'dry run of the %s workload launcher (no GPU work)' % args.workload},
    }))
  if dist.is_initialized():
    dist.destroy_process_group()


def local_shape(shape, world, scaling):
  """weak: the full batch shape on every rank; strong: the global batch split image-wise."""
  if scaling == 'weak' or world == 1:
    return tuple(shape)
  if shape[0] % world:
    raise SystemExit('--scaling strong: global batch %d is not divisible by %d ranks' % (shape[0], world))
  return (shape[0] // world,) + tuple(shape[1:])


def time_cold(args, dev, dom, ids, ev_over=0.0):
  """roofline.hbm_cold: the dominant kernel's launch time on tensors far beyond the 256 MiB Infinity Cache
  (default 256x512x512x3: 384 MiB per tensor, 4.6 GB for the chain), measured exactly like `per_kernel`."""
  if args.cold_shape in ('none', '', None):
    return None
  shape = parse_shape(args.cold_shape)
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  esz = 2 if args.dtype == 'f16' else 4
  try:
    chain = Chain(shape, dtype, dev, args.seed + 77, ids)
    chain.launch()
    per = {k: max(v - ev_over, 1e-6) for k, v in time_kernels(chain, max(5, args.kernel_reps // 5)).items()}
  except RuntimeError as e:  # e.g. out of memory on a shared device: report nothing rather than die
    print('warning: cold-shape measurement skipped (%s)' % e, file=sys.stderr)
    return None
  finally:
    torch.cuda.empty_cache()
  px = shape[0] * shape[1] * shape[2]
  gbps = lambda k: (2 if k.startswith('fwd') else 3) * 3 * esz * px / (per[k] * 1e-3) / 1e9
  chain_ms = sum(per.values())
  # the chain CALLS at this size (expo_chain_fwd / _bwd run tile-major beyond the Infinity Cache: every tile goes
  # through all the steps while its tensors are cached): whole steps between one event pair
  call_ms = None
  try:
    chain = Chain(shape, dtype, dev, args.seed + 77, ids)
    chain.run(2)
    call_runs = []
    for _ in range(5):  # five brackets of 5 steps; the median is reported, all five are in the line
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      chain.run(5)
      e1.record()
      torch.cuda.synchronize()
      call_runs.append(e0.elapsed_time(e1) / 5)
    call_ms = sorted(call_runs)[len(call_runs) // 2]
    del chain
  except RuntimeError as e:
    print('warning: cold chain-call measurement skipped (%s)' % e, file=sys.stderr)
  finally:
    torch.cuda.empty_cache()
  call = {} if call_ms is None else {
      'chain_call_ms_per_step': call_ms,
      'chain_call_ms_runs': call_runs,
      'chain_call_achieved': 8 * 5 * 3 * esz * px / (call_ms * 1e-3) / 1e9,
      'chain_call_frac': 8 * 5 * 3 * esz * px / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
      'chain_call_note': 'expo_chain_fwd + expo_chain_bwd (16 launches per tile and stream + 1 finish), eager, median of '
                         'five brackets of 5 steps between one event pair: the calls walk the batch tile-major (EXPO_CHAIN_TILE_MIB, default 96 MiB '
                         'per tensor and tile), so a consumer finds its producer\'s tile in the Infinity Cache',
  }
  return dict(call, **{
      'shape': 'x'.join(str(v) for v in shape),
      'tensor_MiB': px * 3 * esz / 2**20,
      'kernel': dom,
      'achieved': gbps(dom),
      'frac': gbps(dom) / HBM_PEAK_GBPS,
      'avg_launch_ms': per[dom],
      'slowest_kernel': min(per, key=gbps),
      'slowest_achieved': min(gbps(k) for k in per),
      'chain_achieved': 8 * 5 * 3 * esz * px / (chain_ms * 1e-3) / 1e9,
      'note': 'HIP-event pairs around every launch minus the calibrated pair overhead; chain_achieved = 240 B/px over '
              'the sum of the 16 launch times',
  })


def main():
  args = parse()
  if args.gpus > 1 and 'LOCAL_RANK' not in os.environ and 'RANK' not in os.environ:
    raise SystemExit(self_launch(args))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.dry_run:
    return run_dry(args, world, rank)
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
  # EXPO_BENCH_SHARE_GPU=1 (tests only): the ranks share the visible GPU(s) and talk over gloo -- RCCL refuses two ranks
  # on one device.  Exercises the launcher, the sharding and the timing bracket of a > 1-rank run on a 1-GPU box; its
  # numbers mean nothing (the line says so: config.transport).
  share = os.environ.get('EXPO_BENCH_SHARE_GPU') == '1'
  if local_rank >= torch.cuda.device_count() and not share:
    raise SystemExit('rank %d: only %d GPU(s) visible' % (local_rank, torch.cuda.device_count()))
  device_index = local_rank % torch.cuda.device_count()
  torch.cuda.set_device(device_index)
  dev = torch.device('cuda', device_index)
  dist = None
  if world > 1 or 'LOCAL_RANK' in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
    import torch.distributed as dist
    if share:
      dist.init_process_group('gloo')
    else:
      # the NCCL flight recorder lets exposure_amd.dist verify the watchdog drain in front of a hipGraph capture
      os.environ.setdefault('TORCH_FR_BUFFER_SIZE', '2000')  # (TORCH_NCCL_TRACE_BUFFER_SIZE before torch 2.8)
      dist.init_process_group('nccl', device_id=dev)
  if args.gpus != world and rank == 0:
    print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
  if args.workload == 'train':
    return run_train(args, world, rank, dev, dist)
  if args.workload == 'infer':
    return run_infer(args, world, rank, dev, dist)
  if args.workload == 'allreduce':
    return run_allreduce(args, world, rank, dev, dist)
  if args.workload == 'chain_fused':
    return run_chain_fused(args, world, rank, dev, dist)

  gshape = parse_shape(args.shape or 'C')
  shape = local_shape(gshape, world, args.scaling)
  dtype = torch.float16 if args.dtype == 'f16' else torch.float32
  esz = 2 if args.dtype == 'f16' else 4
  ids = [int(v) for v in args.order.split(',')]
  chain = Chain(shape, dtype, dev, args.seed + rank, ids)
  px = shape[0] * shape[1] * shape[2]
  # One hipGraph replay per step by default.  Small shapes are launch-bound (17 launches in ~85 us
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
  launches = chain.launches_per_step

  result = {
      'metric': 'Mpixels/s through 8-step filter chain fwd+bwd',
      'value': value,
      'unit': 'Mpixels/s',
      'n_gpus': world,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': ms_per_step,
      'higher_is_better': True,
      'scaling': args.scaling,
      'vs_baseline': None,
      'dtype': args.dtype,
      'data': 'synthetic',
      'config': {
          'workload': '8-step filter chain (E,G,W,S+,T,Ct,BW,C) fwd+bwd, %dx%dx%dx3 %s NHWC per GPU, one HIP '
                      'kernel per filter step and direction' % (shape[0], shape[1], shape[2], args.dtype),
          'batch_per_gpu': shape[0],
          'global_batch': shape[0] * world,
          'height': shape[1],
          'width': shape[2],
          'parallelism': 'image-sharded replicas x%d (no data-path collective), %s scaling' % (world, args.scaling),
          'launch': ('hipGraph replay, %d steps (%d captured launches) per replay' % (chain.unroll, launches * chain.unroll)) if chain.graph is not None else 'eager (one C-ABI call per direction)',
          'chain_streams': chain.streams,
          'transport': ('none (one process)' if dist is None else
                        'gloo, ranks SHARING the GPU (EXPO_BENCH_SHARE_GPU test mode: the value means nothing)'
                        if os.environ.get('EXPO_BENCH_SHARE_GPU') == '1' else 'rccl'),
          'chain_algorithmic_GBps': 8 * 5 * 3 * esz * px / (elapsed / args.steps) / 1e9 * 1.0,
      },
  }

  if rank == 0 and not args.no_per_kernel:
    per = time_kernels(chain, args.kernel_reps)
    dom = max(per, key=per.get)
    bpp = (2 if dom.startswith('fwd') else 3) * 3 * esz  # algorithmic bytes per pixel per launch
    # The dominant kernel's launch duration: the in-sequence HIP-event pair average (the kernel runs in the cache state
    # it has inside the chain) MINUS the event pair's own cost, calibrated in this run on a 1-element kernel (a pair
    # around every launch vs many launches between one pair).  Rounds 1-2 reported the raw pair figure, 2.3-4.3 us above
    # rocprofv3's kernel duration depending on the box.
    ev_over = event_pair_overhead_ms(dev)
    launch_ms = max(per[dom] - ev_over, 1e-6)
    # (optional extra leg; off by default so that the rocprofv3 table of the default command holds in-sequence launches only)
    b2b = time_back_to_back(chain, dom) if args.rotating_probe else None
    achieved = bpp * px / (launch_ms * 1e-3) / 1e9
    tensor_mib = px * 3 * esz / 2**20
    result['roofline'] = {
        'bound': 'hbm',
        'kernel': dom,
        'achieved': achieved,
        'peak': HBM_PEAK_GBPS,
        'unit': 'GB/s',
        'frac': achieved / HBM_PEAK_GBPS,
        'traffic': load_traffic(dom, shape, args.dtype),
        'avg_launch_ms': launch_ms,
        'avg_launch_ms_method': 'HIP-event pair around every launch of the kernel inside the chain sequence (%d reps) on '
                                'the launch stream, minus the event pair overhead measured in the same run' %
                                args.kernel_reps,
        'avg_launch_ms_event_pair_raw': per[dom],
        'event_pair_overhead_ms': ev_over,
        # the same kernel launched back to back between ONE event pair with its operands rotating through three buffer
        # sets (0.86 GB: colder than inside the chain, where the upstream gradient was written by the previous launch
        # and still sits in the 256 MiB Infinity Cache); includes the boundary between two dependent kernels
        'avg_launch_ms_rotating_buffers': b2b,
        # what rocprofv3 --kernel-trace measured for this kernel on the same command (committed table)
        'rocprof_avg_us': rocprof_avg_us(dom, shape, args.dtype),
        # 12 tensors of this size cycle through a 256 MiB Infinity Cache (MALL): below ~256 MiB per tensor the
        # consumer of a just-written tensor is partly served from it, so `achieved` is an EFFECTIVE bandwidth
        # (FETCH_SIZE/WRITE_SIZE count MALL hits too); `hbm_cold` is the same kernel on 384 MiB tensors
        'regime': 'mall_assisted' if tensor_mib < 256 else 'hbm_cold',
        'peak_note': 'spec 8.0 TB/s HBM3E. Measured copy ceilings of this access pattern (tools/membench, '
                     'profiles/r02_membench_*.txt): 7.0-7.7 TB/s on 96 MiB buffers (Infinity-Cache assisted), '
                     '5.4-5.9 TB/s on 512-1024 MiB buffers (HBM-cold); guide float4 copy 6.29 TB/s',
        'frac_of_copy_ceiling': achieved / 6290.0,
        'algorithmic_bytes_per_launch': bpp * px,
        # the whole timed region (16 kernels + 1 finish launch per step): 240 B/pixel/step over the step time
        'chain_achieved': result['config']['chain_algorithmic_GBps'],
        'chain_frac': result['config']['chain_algorithmic_GBps'] / HBM_PEAK_GBPS,
    }
    result['per_kernel_note'] = ('raw HIP-event pairs around whole-batch launches in chain order; each includes the pair '
                                 'overhead roofline.event_pair_overhead_ms')
    result['per_kernel'] = {
        k: {
            'ms': v,
            'GBps': (2 if k.startswith('fwd') else 3) * 3 * esz * px / (v * 1e-3) / 1e9
        } for k, v in per.items()
    }
    if world == 1 and tensor_mib < 256:
      del chain  # free the 1.2 GB of the timed chain before the 4.6 GB cold one
      t_cold = time.perf_counter()
      cold = time_cold(args, dev, dom, ids, ev_over)
      print('hbm_cold leg: %.1f s' % (time.perf_counter() - t_cold), file=sys.stderr)
      if cold is not None:
        result['roofline']['hbm_cold'] = cold
  barrier()
  if not args.no_legs and os.environ.get('EXPO_BENCH_LEGS', '1') != '0':
    try:
      del chain  # (rank 0 may already have dropped it in front of the cold leg)
    except NameError:
      pass
    torch.cuda.empty_cache()
    run_legs(result, args, world, rank, dev, dist)
    # a compact copy where the driver's parser keeps it (`config`): BASELINE configs 2 and 3 / 4 from the same command
    legs = result.get('legs', {})
    chain_a, train = legs.get('chain_64x64x64x3', {}), legs.get('train', {})
    launches, launches_file = committed_train_launches()
    result['config']['legs'] = {
        'chain_64x64x64_Mpixels_per_s': chain_a.get('Mpixels_per_s'),
        'train_ms_per_iteration': train.get('ms_per_iteration'),
        'train_images_per_s': train.get('images_per_s'),
        'train_roofline_frac': train.get('roofline', {}).get('frac'),
        'train_launches_per_iteration': launches,
        'train_launches_source': launches_file,
        'capture_drain_verified': train.get('capture_drain_verified'),
        'allreduce_bus_GBps': legs.get('allreduce', {}).get('bus_GBps'),
        'errors': {k: v['error'] for k, v in legs.items() if isinstance(v, dict) and 'error' in v} or None,
    }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    result['cpu_baseline'] = cpu_baseline()
    try:
      result['cpu_baseline']['parity_check'] = parity_sample(dev)
    except Exception as e:  # the checker is optional equipment of the benchmark, never a reason to lose the line
      result['cpu_baseline']['parity_check'] = {'error': str(e)[:200]}
  if rank == 0:
    print(json.dumps(result))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
