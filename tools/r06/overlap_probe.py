"""Probe: inside a critic update the weight gradients of the LOSS rows (2n images) do not depend on the penalty's tangent
pass (n images through the four layers under masks).  How long do the two take one after the other, and forked onto two
streams inside one hipGraph?   usage: python tools/r06/overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402

LAYERS = [(6, 32, 64), (32, 64, 32), (64, 128, 16), (128, 256, 8)]
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n = 64


def tensors(rows):
  out = []
  for cin, cout, h in LAYERS:
    x = torch.randn((rows, h, h, cin), device=dev, generator=g)
    w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    z = torch.randn((rows, h // 2, h // 2, cout), device=dev, generator=g)
    gy = torch.randn_like(z)
    out.append((x, w, z, gy, torch.empty_like(w), torch.empty((cout,), device=dev)))
  return out


T = tensors(n)          # the tangent pass: n rows
L2 = tensors(2 * n)     # weight gradients of the loss rows
L3 = tensors(3 * n)     # weight gradients of all rows (today's single launch)
P1 = tensors(n)         # weight gradients of the penalty rows


def tangent():
  for x, w, z, gy, dw, db in T:
    _cabi.conv4x4s2_fwd_mask(x, w, z, z, 0.2)


def wrw(ts):
  _cabi.conv4x4s2_wrw_group([(x, gy, dw, db, None) for x, w, z, gy, dw, db in ts])


def time_graph(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    for _ in range(reps):
      fn()
  graph.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(4):
    graph.replay()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / (4 * reps) * 1e3


side = torch.cuda.Stream()


def today():
  tangent()
  wrw(L3)


def serial_split():
  tangent()
  wrw(L2)
  wrw(P1)


def forked():
  cur = torch.cuda.current_stream()
  side.wait_stream(cur)
  with torch.cuda.stream(side):
    wrw(L2)
  tangent()
  wrw(P1)
  cur.wait_stream(side)


print('tangent pass alone            %.1f us' % time_graph(tangent))
print('wrw 3n alone                  %.1f us' % time_graph(lambda: wrw(L3)))
print('wrw 2n alone                  %.1f us' % time_graph(lambda: wrw(L2)))
print('wrw n alone                   %.1f us' % time_graph(lambda: wrw(P1)))
print('tangent -> wrw 3n (today)     %.1f us' % time_graph(today))
print('tangent -> wrw 2n -> wrw n    %.1f us' % time_graph(serial_split))
print('wrw 2n || (tangent -> wrw n)  %.1f us' % time_graph(forked))
