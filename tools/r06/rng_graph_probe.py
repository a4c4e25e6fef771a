"""Probe: a torch.Generator registered with a captured graph yields the eager sequence, replay after replay, and keeps
working for eager draws in between.  usage: python tools/r06/rng_graph_probe.py"""
import torch

dev = torch.device('cuda:0')


def seq(gen, k):
  out = []
  for _ in range(k):
    u = torch.rand((2, 8, 4096), generator=gen, device=dev)
    a = torch.rand((8, 1, 1, 1), generator=gen, device=dev)
    out.append((u.clone(), a.clone()))
  return out


g1 = torch.Generator(device=dev).manual_seed(5)
want = seq(g1, 6)
g2 = torch.Generator(device=dev).manual_seed(5)
got = seq(g2, 1)  # eager first
graph = torch.cuda.CUDAGraph()
graph.register_generator_state(g2)
with torch.cuda.graph(graph):
  u = torch.rand((2, 8, 4096), generator=g2, device=dev)
  a = torch.rand((8, 1, 1, 1), generator=g2, device=dev)
# does the capture itself consume numbers?
for _ in range(3):
  graph.replay()
  got.append((u.clone(), a.clone()))
got += seq(g2, 1)  # eager again
graph.replay()
got.append((u.clone(), a.clone()))
torch.cuda.synchronize()
for i, ((u1, a1), (u2, a2)) in enumerate(zip(want, got)):
  print(i, torch.equal(u1, u2), torch.equal(a1, a2))
