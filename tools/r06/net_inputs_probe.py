"""Probe: expo_net_inputs against the four launches it replaces, us per call under hipGraph replay (the critic update's
input: 64 real + 64 fake rows of fp16 pools + 64 interpolated, 64 x 64; the value net's: 2 x 64 with 11 state values).
usage: python tools/r06/net_inputs_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd import _cabi  # noqa: E402
from tools.r06.conv_sweep import timeit  # noqa: E402

dev = torch.device('cuda:0')
n, h = 64, 64
g = torch.Generator(device=dev).manual_seed(0)
pool_a = torch.rand((2048, h, h, 3), device=dev, generator=g).half()
pool_b = torch.rand((2048, h, h, 3), device=dev, generator=g).half()
ra = torch.randperm(2048, device=dev, generator=g)[:n].contiguous()
rb = torch.randperm(2048, device=dev, generator=g)[:n].contiguous()
alpha = torch.rand((n,), device=dev, generator=g)
x = torch.empty((3 * n, h, h, 3), device=dev)
stats = torch.empty((3 * n, 3), device=dev)
planes = torch.empty((3 * n, h, h, 6), device=dev)
xi = torch.empty((n, h, h, 3), device=dev)


def separate():
  _cabi.gp_inputs(pool_a, pool_b, alpha, x[:2 * n], x[2 * n:], real_rows=ra, fake_rows=rb)
  _cabi.critic_stats(x, stats)
  _cabi.planes_concat(x, stats, planes, 0.5)


def fused():
  _cabi.net_inputs(pool_a, pool_b, alpha, planes, stats, x_out=xi, x_first=2 * n, a_rows=ra, b_rows=rb)


va, vb = torch.randn((n, 11), device=dev, generator=g), torch.randn((n, 11), device=dev, generator=g)
planes_v = torch.empty((2 * n, h, h, 17), device=dev)
stats_v = torch.empty((2 * n, 3), device=dev)
a, b = pool_a[:n].contiguous(), pool_b[:n].contiguous()


def fused_v():
  _cabi.net_inputs(a, b, None, planes_v, stats_v, x_out=xi, x_first=n, vec_a=va, vec_b=vb)


print('critic update input: separate %.1f us, net_inputs %.1f us;  value net input: net_inputs %.1f us' %
      (timeit(separate), timeit(fused), timeit(fused_v)))
