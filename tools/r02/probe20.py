"""Does splitting the batch over concurrent HIP streams fill the ramp / tail bubbles between the chain's
17 dependent launches?  parts = 1 (the product path), 2, 4 independent image sub-batches, each with its own
stream (parallel branches of one hipGraph), same total 64x512x512x3 fp16."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from exposure_amd import _cabi

PARTS = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(","))
dev = torch.device('cuda', 0)
N, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,512,512').split(','))


def build(parts):
  n = N // parts
  chains = [bench.Chain((n, H, W, 3), torch.float16, dev, 100 + i) for i in range(parts)]
  code = _cabi._dtype_code(chains[0].acts[0])
  wss = [_cabi.new_workspace(dev, _cabi.workspace_bytes(n, H, W, code, 8)) for _ in range(parts)]
  streams = [torch.cuda.Stream() for _ in range(parts)]

  def launch():
    cur = torch.cuda.current_stream()
    for s, c, w in zip(streams, chains, wss):
      s.wait_stream(cur)
      with torch.cuda.stream(s):
        _cabi.chain_fwd(c.ids, c.acts, c.params)
        _cabi.chain_bwd(c.ids, c.acts, c.grads, c.params, c.dparams, workspace=w)
    for s in streams:
      cur.wait_stream(s)
  return launch, chains


def timeit(fn, reps=30):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


out = {}
for rep in range(2):
  for parts in PARTS:
    launch, chains = build(parts)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = timeit(launch, 10)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      launch()
    ms = timeit(g.replay)
    out.setdefault(str(parts), []).append({'graph_ms': round(ms, 4), 'eager_ms': round(eager, 4),
                                          'Mpix_s': round(N * H * W / ms / 1e3, 0)})
    print(parts, out[str(parts)][-1], flush=True)
    del g, launch, chains
    torch.cuda.empty_cache()
print(json.dumps(out))
