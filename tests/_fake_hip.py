"""TEST-ONLY stand-in for the C-ABI binding so host logic (agent step, GAN step, gloo data
parallelism) can be exercised in this GPU-less container.  It patches the *python binding*
functions of ``exposure_amd._cabi`` with oracle-backed CPU implementations via
``unittest.mock``; the product package has no knowledge of it and no fallback of its own."""
import contextlib
from unittest import mock

import numpy as np
import torch

from oracle import agent_np
from oracle import nets_np
from oracle import filters_torch as ft

NUM_PARAMS = (1, 1, 3, 1, 8, 1, 1, 24, 2)


def _fwd(fid, x, y, params):
  y.copy_(ft.process_packed(fid, x.double(), params.double()).to(y.dtype))


def _bwd(fid, x, dy, dx, params, dparams, hsv_grad_mode=0, accumulate=False):
  with torch.enable_grad():  # called from inside autograd.Function.backward (grad mode off)
    gx, gp = ft.backward_packed(fid, x.double(), params.double(), dy.double(), hsv_grad_mode)
  if dx is not None:
    dx.copy_(gx.to(dx.dtype))
  if accumulate:
    dparams.add_(gp.float())
  else:
    dparams.copy_(gp.float())


def _dispatch_fwd(ids, x, y, params, penalty=None):
  for n in range(x.shape[0]):
    fid = int(ids[n])
    if fid < 0:
      y[n].zero_()
    else:
      p = params[n:n + 1, :NUM_PARAMS[fid]].contiguous()
      y[n:n + 1].copy_(ft.process_packed(fid, x[n:n + 1].double(), p.double()).to(y.dtype))
  if penalty is not None:
    penalty.copy_(((y.double() - 1).clamp_min(0)**2).mean(dim=(1, 2, 3)).float())


def _dispatch_bwd(ids, x, dy, dx, params, dparams, dpenalty=None, hsv_grad_mode=0):
  dparams.zero_()
  cnt = x.shape[1] * x.shape[2] * 3
  for n in range(x.shape[0]):
    fid = int(ids[n])
    if fid < 0:
      if dx is not None:
        dx[n].zero_()
      continue
    p = params[n:n + 1, :NUM_PARAMS[fid]].contiguous().double()
    xi = x[n:n + 1].double()
    g = dy[n:n + 1].double()
    if dpenalty is not None:
      yi = ft.process_packed(fid, xi, p)
      g = g + 2.0 * (yi - 1).clamp_min(0) * float(dpenalty[n]) / cnt
    with torch.enable_grad():
      gx, gp = ft.backward_packed(fid, xi, p, g, hsv_grad_mode)
    if dx is not None:
      dx[n:n + 1].copy_(gx.to(dx.dtype))
    dparams[n, :NUM_PARAMS[fid]] = gp[0].float()


def _chain_fused_fwd(ids, params, x, y):
  cur = x.double()
  for st in range(ids.shape[1]):
    nxt = torch.zeros_like(cur)
    for n in range(x.shape[0]):
      fid = int(ids[n, st])
      if fid >= 0:
        p = params[n:n + 1, st, :NUM_PARAMS[fid]].contiguous().double()
        nxt[n:n + 1] = ft.process_packed(fid, cur[n:n + 1], p)
    cur = nxt
  y.copy_(cur.to(y.dtype))


def _chain_fused_bwd(ids, params, x, dy, dx, dparams, hsv_grad_mode=0, workspace=None):
  dparams.zero_()
  for n in range(x.shape[0]):
    xi = x[n:n + 1].double().requires_grad_(True)
    ps = []
    cur = xi
    alive = True
    for st in range(ids.shape[1]):
      fid = int(ids[n, st])
      if fid < 0:
        cur = cur * 0.0
        ps.append(None)
        continue
      q = params[n:n + 1, st, :NUM_PARAMS[fid]].double().requires_grad_(True)
      ps.append(q)
      with torch.enable_grad():
        cur = ft.process_packed(fid, cur, q, hsv_grad_mode)
    live = [xi] + [q for q in ps if q is not None]
    with torch.enable_grad():
      grads = torch.autograd.grad(cur, live, dy[n:n + 1].double(), allow_unused=True)
    dx[n:n + 1].copy_((grads[0] if grads[0] is not None else torch.zeros_like(xi)).to(dx.dtype))
    it = iter(grads[1:])
    for st, q in enumerate(ps):
      if q is not None:
        g = next(it)
        if g is not None:
          dparams[n, st, :q.shape[1]] = g[0].float()


def _raw_mask(mp):
  """the C-ABI takes tanh_range(-5, 5)(raw) = 5 tanh(raw); the oracle takes raw"""
  return torch.atanh((mp.double() / 5.0).clamp(-1 + 1e-15, 1 - 1e-15))


def _apply_fwd(fid, x, y, params, mask_params, maximum_sharpness, minimum_strength):
  y.copy_(ft.apply_masked(fid, x.double(), params.double(), _raw_mask(mask_params), maximum_sharpness,
                          minimum_strength).to(y.dtype))


def _apply_bwd(fid, x, dy, dx, params, dparams, mask_params, dmask_params, maximum_sharpness, minimum_strength,
               hsv_grad_mode=0):
  with torch.enable_grad():
    mp = mask_params.double().detach().clone().requires_grad_(True)
    xi = x.double().detach().clone().requires_grad_(True)
    pi = params.double().detach().clone().requires_grad_(True)
    out = ft.apply_masked(fid, xi, pi, torch.atanh((mp / 5.0).clamp(-1 + 1e-15, 1 - 1e-15)), maximum_sharpness,
                          minimum_strength, hsv_grad_mode)
    gx, gp, gm = torch.autograd.grad(out, [xi, pi, mp], dy.double())
  if dx is not None:
    dx.copy_(gx.to(dx.dtype))
  dparams.copy_(gp.float())
  dmask_params.copy_(gm.float())


def _vignet_fwd(x, y, mask_params, maximum_sharpness, masking):
  y.copy_(ft.vignet_apply(x.double(), _raw_mask(mask_params), maximum_sharpness, bool(masking)).to(y.dtype))


def _vignet_bwd(x, dy, dx, mask_params, dmask_params, maximum_sharpness, masking):
  with torch.enable_grad():
    mp = mask_params.double().detach().clone().requires_grad_(True)
    xi = x.double().detach().clone().requires_grad_(True)
    out = ft.vignet_apply(xi, torch.atanh((mp / 5.0).clamp(-1 + 1e-15, 1 - 1e-15)), maximum_sharpness, bool(masking))
    gx, gm = torch.autograd.grad(out, [xi, mp], dy.double(), allow_unused=True)
  if dx is not None:
    dx.copy_(gx.to(dx.dtype))
  dmask_params.copy_(gm.float() if gm is not None else torch.zeros_like(dmask_params))


def _stats(x, stats):
  stats.copy_(torch.from_numpy(agent_np.critic_stats(x.double().numpy())).float())


def _apply_dispatch_fwd(ids, x, y, params, mask_params, maximum_sharpness, minimum_strength):
  for n in range(x.shape[0]):
    fid = int(ids[n])
    if fid < 0:
      y[n].zero_()
    else:
      _apply_fwd(fid, x[n:n + 1], y[n:n + 1], params[n:n + 1, :NUM_PARAMS[fid]].contiguous(), mask_params[n:n + 1],
                 maximum_sharpness, minimum_strength)


def _apply_dispatch_bwd(ids, x, dy, dx, params, dparams, mask_params, dmask_params, maximum_sharpness, minimum_strength,
                        hsv_grad_mode=0):
  dparams.zero_()
  dmask_params.zero_()
  for n in range(x.shape[0]):
    fid = int(ids[n])
    if fid < 0:
      if dx is not None:
        dx[n].zero_()
      continue
    dp = torch.empty((1, NUM_PARAMS[fid]))
    _apply_bwd(fid, x[n:n + 1], dy[n:n + 1], dx[n:n + 1] if dx is not None else None,
               params[n:n + 1, :NUM_PARAMS[fid]].contiguous(), dp, mask_params[n:n + 1], dmask_params[n:n + 1],
               maximum_sharpness, minimum_strength, hsv_grad_mode)
    dparams[n, :NUM_PARAMS[fid]] = dp[0]


def _stats_cache(x):
  return nets_np.stat_features(x.detach().double().numpy())[1]


def _stats_bwd(x, stats, dstats, dx):
  dx.copy_(torch.from_numpy(nets_np.stat_features_backward(_stats_cache(x), dstats.detach().double().numpy())).to(dx.dtype))


def _stats_jvp(x, stats, v, jv, workspace=None):
  jv.copy_(torch.from_numpy(nets_np.stat_features_jvp(_stats_cache(x), v.detach().double().numpy())).float())


def _stats_hvp(x, dstats, jv, v, out):
  out.copy_(torch.from_numpy(nets_np.stat_features_hvp(_stats_cache(x), dstats.detach().double().numpy(),
                                                       v.detach().double().numpy())).to(out.dtype))


def _penalty_bwd(y, dpen, dy):
  cnt = y.shape[1] * y.shape[2] * 3
  dy.copy_((2.0 * (y.double() - 1).clamp_min(0) * dpen.double()[:, None, None, None] / cnt).to(dy.dtype))


def _bias_lrelu_fwd(y, bias, z, leak=0.2):
  v = y if bias is None else y + bias
  z.copy_(torch.where(v > 0, v, v * leak))


def _lrelu_bwd(z, dz, dy, leak=0.2):
  f1 = 0.5 * (1 + leak)
  dy.copy_(dz * torch.where(z > 0, torch.ones_like(z), torch.where(z < 0, torch.full_like(z, leak), torch.full_like(z, f1))))


def _curve_fwd(x, y, params, curves, steps):
  fid = 4 if curves == 1 else 7
  y.copy_(ft.process_packed(fid, x.double(), params.double()).to(y.dtype))


def _curve_bwd(x, dy, dx, params, dparams, curves, steps, workspace=None):
  fid = 4 if curves == 1 else 7
  with torch.enable_grad():
    gx, gp = ft.backward_packed(fid, x.double(), params.double(), dy.double(), 0)
  if dx is not None:
    dx.copy_(gx.to(dx.dtype))
  dparams.copy_(gp.float())


def _lrelu_bwd_bias(z, dz, dy, dbias, leak=0.2, workspace=None):
  _lrelu_bwd(z, dz, dy, leak)
  dbias.copy_(dy.reshape(-1, dy.shape[-1]).sum(dim=0))


def _penalty(y, pen):
  pen.copy_(torch.from_numpy(agent_np.overexposure_penalty(y.double().numpy())).float())


@contextlib.contextmanager
def fake_hip():
  with mock.patch.multiple('exposure_amd._cabi', filter_fwd=_fwd, filter_bwd=_bwd, dispatch_fwd=_dispatch_fwd,
                           dispatch_bwd=_dispatch_bwd, critic_stats=_stats, overexposure_penalty=_penalty,
                           critic_stats_bwd=_stats_bwd, critic_stats_jvp=_stats_jvp, critic_stats_hvp=_stats_hvp,
                           overexposure_penalty_bwd=_penalty_bwd, bias_lrelu_fwd=_bias_lrelu_fwd, lrelu_bwd=_lrelu_bwd,
                           vignet_apply_fwd=_vignet_fwd, vignet_apply_bwd=_vignet_bwd,
                           apply_dispatch_fwd=_apply_dispatch_fwd, apply_dispatch_bwd=_apply_dispatch_bwd,
                           chain_fused_fwd=_chain_fused_fwd, chain_fused_bwd=_chain_fused_bwd, apply_fwd=_apply_fwd, apply_bwd=_apply_bwd,
                           curve_fwd=_curve_fwd, curve_bwd=_curve_bwd, lrelu_bwd_bias=_lrelu_bwd_bias,
                           lrelu_bwd_bias_supported=lambda z, dz: z.shape[-1] % 4 == 0):
    yield
