"""Convolution building block of the policy / critic / value networks.

All convolutions of the reference are 4x4, stride 2, SAME (``agent.py:21-32``, ``critics.py:13-35``)
on small NHWC tensors (64x64 and down, batch 64).  MIOpen's kernels for the *double backward* of
such a convolution (needed by the WGAN-GP gradient penalty, ``net.py:174-194``) take 0.75-2 ms
each (``tools/conv_probe.py``), which made them 2/3 of a training iteration.  Here the same
convolution is an explicit im2col + GEMM in NHWC:

  x (N,H,W,C) -pad 1-> unfold(H,4,2).unfold(W,4,2) -> (N*Ho*Wo, C*16) @ W(O, C*16)^T -> (N,Ho,Wo,O)

so forward, backward and double backward are all plain GEMMs (hipBLASLt, MFMA) plus strided
copies, with no layout change anywhere between the filter kernels' NHWC images and the FC heads.
The parameter keeps ``nn.Conv2d``'s (O, C, 4, 4) layout (checkpoint mapping, Xavier fan-in/out).
"""
import torch
from torch import nn


class Conv4x4S2(nn.Module):
  """4x4 / stride 2 / pad 1 convolution on NHWC input, returning NHWC."""

  def __init__(self, in_channels, out_channels):
    super().__init__()
    self.in_channels, self.out_channels = in_channels, out_channels
    self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 4, 4))
    self.bias = nn.Parameter(torch.zeros(out_channels))
    nn.init.xavier_uniform_(self.weight)

  def forward(self, x):
    n, h, w, c = x.shape
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))  # pad W and H by 1 (NHWC: last dim is C)
    cols = xp.unfold(1, 4, 2).unfold(2, 4, 2)  # (N, Ho, Wo, C, 4, 4) strided view
    ho, wo = cols.shape[1], cols.shape[2]
    cols = cols.reshape(n * ho * wo, c * 16)
    out = torch.addmm(self.bias, cols, self.weight.reshape(self.out_channels, c * 16).t())
    return out.view(n, ho, wo, self.out_channels)
