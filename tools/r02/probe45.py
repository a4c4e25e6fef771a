"""Does it matter that every step of the benchmark rewrites its buffers with the data they already hold?
Two chains share all intermediate buffers; graph 1 replays the same chain twice (every store writes what is already
there), graph 2 alternates two chains with different inputs, upstream gradients and parameters."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from exposure_amd import synthetic

dev = torch.device('cuda', 0)
shape = synthetic.SHAPES[sys.argv[1] if len(sys.argv) > 1 else 'C']
cA = bench.Chain(shape, torch.float16, dev, 1)
cB = bench.Chain(shape, torch.float16, dev, 2)
cB.acts[1:] = cA.acts[1:]
cB.grads[:8] = cA.grads[:8]
for c in (cA, cB):
  c.launch()
torch.cuda.synchronize()


def graph_of(seq):
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for c in seq:
      c.launch()
  return g


def timeit(g, steps_per_replay, reps=20):
  for _ in range(5):
    g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps / steps_per_replay


gs = {'same data (A A A A)': graph_of([cA] * 4), 'alternating data (A B A B)': graph_of([cA, cB] * 2)}
out = {}
for rep in range(3):
  for name, g in gs.items():
    out.setdefault(name, []).append(round(timeit(g, 4), 4))
print(json.dumps(out))
