"""The generator / value step's critic and value-net passes as hand-scheduled launch sequences
(``/root/reference/net.py:56-165, 222-241``; ``critics.py:42-98``).

In one G / V step the reference evaluates the critic on the retouched and on the input images (the reward is the
DIFFERENCE of the two logits, net.py:92-108) and the value net on the input images with the old states and on the
retouched images with the new ones (net.py:76-90, 110-134).  Only the agent's own path needs autograd; the four passes
around it are two fixed nets whose parameters this step either does not update at all (the critic) or updates through ONE
of the two passes (the value net: ``v_loss`` reaches theta_v through ``old_value`` only).  So, like the critic update
(``critic_direct.py``), each net runs its two passes as ONE batch of 2n images and its backward as explicit launches:

  critic  [retouched | input]                 forward 2n; backward for the retouched rows only: data gradients (activation
                                              gradient in their epilogue), first layer on the vector ALUs, the statistics'
                                              J^T -- no weight gradients (frozen)
  value   [input + states | retouched + new]  forward 2n; backward of BOTH row blocks through one data-gradient launch per
                                              layer; weight / bias gradients from the first block's rows (v_loss), the image
                                              gradient from the second block's (g_loss)

The two nets' input sides are one launch each (``expo_net_inputs``), their layers 2 .. 4 run pairwise as one grid
(``expo_conv4x4s2_fwd_pair``: same geometry, different weights), the value net's 17-plane first layer with its constant planes
folded (``expo_conv4x4s2_fwd_planes``), fc1 split-K in-house (``expo_fc_*``).  The loss glue is ``expo_generator_losses`` as before (its coefficient rows ARE the rows' upstream gradients).  The image
gradient of both nets and the coefficients of surrogate / penalty then enter the agent's autograd graph in ONE backward
pass (``torch.autograd.backward([fake_output, surrogate, penalty], ...)``): theta_g sees exactly d g_loss / d theta_g,
theta_v exactly d v_loss / d theta_v -- the two losses share no path (q is a constant inside the advantage, the value and
critic parameters are frozen in the passes g_loss reaches), which is why the autograd step's two backward passes can be one
here.  ``tests/test_generator_direct.py`` holds losses, outputs and every gradient tensor to the autograd step.
"""
import torch

from . import _cabi
from .critic_direct import fc_split
from .nn_ops import once_differentiable_convnets, planes_fold
from .util import STATE_STEP_DIM, STATE_STOPPED_DIM

LEAK = 0.2


def _net_ok(net, h, w):
  for conv in net.convs:
    wt = conv.weight
    if not (wt.is_cuda and wt.dtype == torch.float32 and tuple(wt.shape[2:]) == (4, 4) and
            wt.permute(0, 2, 3, 1).is_contiguous() and conv.bias is not None and wt.shape[0] % 4 == 0):
      return False
    if h % 2 or w % 2 or (w // 2) % 2:
      return False
    h, w = h // 2, w // 2
  return h * w * net.convs[-1].weight.shape[0] == net.flat and net.fc2.weight.shape[0] == 1


def supported(gan, fake_input, states):
  cfg = gan.cfg
  if cfg.gan != 'w' or cfg.supervised or STATE_STOPPED_DIM != 1 or STATE_STEP_DIM != 2:
    return False
  if not (fake_input.is_cuda and fake_input.dim() == 4 and fake_input.shape[-1] == 3 and
          fake_input.dtype in (torch.float16, torch.float32) and states.dtype == torch.float32):
    return False
  h, w = fake_input.shape[1], fake_input.shape[2]
  return (gan.critic.num_state_dim == 0 and gan.value.num_state_dim == states.shape[1] and _net_ok(gan.critic, h, w) and
          _net_ok(gan.value, h, w))


class _PairPass:
  """One net on the batch [a | b] (n images each): the constructor builds its input, ``forward_all`` runs the
  convolutions and the head (-> ``logits`` (2n,)), ``backward(dlogit, ...)`` the explicit backward."""

  def __init__(self, net, img_a, img_b, vec_a=None, vec_b=None, x_rows=None):
    self.net, self.n = net, img_a.shape[0]
    n, m = self.n, 2 * img_a.shape[0]
    dev = img_a.device
    f32 = dict(dtype=torch.float32, device=dev)
    self.stats = torch.empty((m, 3), **f32)
    v0 = 0 if vec_a is None else vec_a.shape[1]
    self.stats_first = 3 + v0  # first statistics plane of the net's input
    a0 = torch.empty((m,) + tuple(img_a.shape[1:3]) + (6 + v0,), **f32)
    x_rows = slice(0, m) if x_rows is None else x_rows
    self.x_first = x_rows.start  # self.x: the float32 images of the rows whose image gradient the backward returns
    if img_a.shape[1] * img_a.shape[2] <= _cabi.NET_INPUTS_MAX_PIXELS:
      self.x = torch.empty((x_rows.stop - x_rows.start,) + tuple(img_a.shape[1:]), **f32)
      _cabi.net_inputs(img_a.contiguous(), img_b.contiguous(), None, a0, self.stats, x_out=self.x, x_first=self.x_first,
                       vec_a=None if vec_a is None else vec_a.float().contiguous(),
                       vec_b=None if vec_b is None else vec_b.float().contiguous())  # one launch
    else:
      x = torch.empty((m,) + tuple(img_a.shape[1:]), **f32)
      _cabi.gp_inputs(img_a.contiguous(), img_b.contiguous(), None, x, None)
      _cabi.critic_stats(x, self.stats)
      vec = self.stats if vec_a is None else torch.cat([torch.cat([vec_a.float(), vec_b.float()], dim=0), self.stats], dim=1)
      _cabi.planes_concat(x, vec.contiguous(), a0, 0.5)
      self.x = x[x_rows]
    self.acts = [a0]
    self.m = m

  def conv_layer(self, l):
    """(input, weight, bias, fresh output) of layer l (0-based); the output joins ``acts``."""
    conv = self.net.convs[l]
    a = self.acts[l]
    z = torch.empty((self.m, a.shape[1] // 2, a.shape[2] // 2, conv.weight.shape[0]), dtype=torch.float32, device=a.device)
    self.acts.append(z)
    return a, conv.weight, conv.bias, z

  def head(self):
    net, m = self.net, self.m
    f32 = dict(dtype=torch.float32, device=self.acts[0].device)
    self.flat = self.acts[-1].reshape(m, net.flat)
    self.hidden = net.fc1.weight.shape[0]
    self.logits = torch.empty((m,), **f32)
    self.h, self.dh_unit = torch.empty((m, self.hidden), **f32), torch.empty((m, self.hidden), **f32)
    # every row as an "interpolated" row: upstream gradient 1, i.e. dh_unit = w2 * slope(h) -- scaled by the rows' real
    # upstream gradients once the loss kernel has produced them
    self.split = fc_split(net.fc1, m)  # fc1 with its K dimension split: the head kernel adds the slabs
    if self.split:
      hpre = torch.empty((_cabi.fc_fwd_slabs_count(m, net.flat), m, self.hidden), **f32)
      _cabi.fc_fwd_slabs(self.flat, net.fc1.weight, hpre)
    else:
      hpre = torch.addmm(net.fc1.bias, self.flat, net.fc1.weight.t())
    _cabi.critic_head_fwd(hpre, net.fc2.weight.reshape(self.hidden), net.fc2.bias, 0, 0, m, 1.0, self.logits, self.h,
                          self.dh_unit, LEAK, b1=net.fc1.bias if self.split else None)

  @staticmethod
  def forward_all(passes):
    """The convolutions and heads of the passes: layer by layer, two passes whose layer has ONE geometry (the critic's and
    the value net's layers 2 .. 4 on 2n images each) as one grid (expo_conv4x4s2_fwd_pair)."""
    depth = len(passes[0].net.convs)
    same_depth = all(len(p.net.convs) == depth for p in passes)
    for l in range(depth if same_depth else 0):
      items = [p.conv_layer(l) for p in passes]
      if len(items) == 2 and items[0][0].shape == items[1][0].shape and items[0][1].shape == items[1][1].shape:
        _cabi.conv4x4s2_fwd_pair(items[0], items[1], 1, LEAK)
      else:
        for a, w, b, z in items:
          # (first layers: channels 3 .. of expo_net_inputs' output are per-image constants -- folded where that pays)
          fold = l == 0 and planes_fold(a.shape, w.shape[0])
          (_cabi.conv4x4s2_fwd_planes if fold else _cabi.conv4x4s2_fwd)(a, w, b, z, 1, LEAK)
    if not same_depth:
      for p in passes:
        for l in range(len(p.net.convs)):
          a, w, b, z = p.conv_layer(l)
          _cabi.conv4x4s2_fwd(a, w, b, z, 1, LEAK)
    for p in passes:
      p.head()

  def backward(self, dlogit, rows, rows_x, grads=None, rows_w=None):
    """``dlogit`` the upstream gradients of the logits of ``rows``, the row range the backward covers (a slice); ``rows_x``
    the rows (inside it) whose IMAGE gradient is wanted, ``rows_w`` the rows whose weight gradients go to ``grads``
    (id(parameter) -> tensor).  -> d image (float32, rows_x)."""
    net, convs = self.net, list(self.net.convs)
    lo = rows.start
    rel = lambda sl: slice(sl.start - lo, sl.stop - lo)
    dh = self.dh_unit[rows] * dlogit[:, None]
    top = self.acts[-1][rows]
    gy = torch.empty_like(top)
    if self.split:
      _cabi.fc_bwd_data_mask(dh, net.fc1.weight, top, gy, LEAK)
    else:
      dz = torch.mm(dh, net.fc1.weight)
      _cabi.lrelu_bwd(top, dz.reshape(top.shape), gy, LEAK)
    gys = [None] * (len(convs) + 1)
    gys[len(convs)] = gy
    for l in range(len(convs), 1, -1):
      below = self.acts[l - 1][rows]
      g = torch.empty_like(below)
      _cabi.conv4x4s2_bwd_data_mask(gys[l], convs[l - 1].weight, below, g, LEAK)
      gys[l - 1] = g
    a0x = self.acts[0][rows_x]
    u0 = torch.empty_like(a0x)
    _cabi.conv4x4s2_bwd_data(gys[1][rel(rows_x)], convs[0].weight, u0)
    gs = torch.empty((u0.shape[0], 3), dtype=torch.float32, device=u0.device)
    _cabi.plane_sums(u0, gs, self.stats_first)
    xr = self.x[rows_x.start - self.x_first:rows_x.stop - self.x_first]
    ds = torch.empty_like(xr)
    _cabi.critic_stats_bwd(xr, self.stats[rows_x], gs, ds)
    d_img = u0[..., :3] + ds
    if grads is not None:
      rw = rel(rows_w)
      _cabi.conv4x4s2_wrw_group([(self.acts[l - 1][rows_w], gys[l][rw], grads[id(conv.weight)], grads[id(conv.bias)], None)
                                 for l, conv in enumerate(convs, start=1)])
      dhw = dh[rw]
      if self.split:
        _cabi.fc_wrw(dhw.contiguous(), self.flat[rows_w], grads[id(net.fc1.weight)])
      else:
        torch.mm(dhw.t(), self.flat[rows_w], out=grads[id(net.fc1.weight)])
      torch.sum(dhw, dim=0, out=grads[id(net.fc1.bias)])
      torch.mv(self.h[rows_w].t(), dlogit[rw], out=grads[id(net.fc2.weight)].reshape(self.hidden))
      torch.sum(dlogit[rw], dim=0, keepdim=True, out=grads[id(net.fc2.bias)])
    return d_img


def _grad_targets(gan, module, bucket):
  """Where a hand-computed gradient goes: the bucket's views when collectives run, fresh tensors otherwise."""
  params = list(module.parameters())
  b = gan.buckets[bucket]
  if gan._collectives():
    b.zero()
    b.disarm()
    return {id(p): p.grad for p in params}
  b.release()
  out = {}
  for p in params:
    p.grad = torch.empty_like(p, memory_format=torch.preserve_format)
    out[id(p)] = p.grad
  return out


def generator_step_losses_and_grads(gan, fake_input, z, states, progress, dropout_masks, adam_steps=(None, None)):
  """Losses of one G / V step and their gradients: theta_v's written to ``p.grad`` by hand, theta_g's through ONE autograd
  backward over the agent.  -> the dict ``GAN.generator_losses`` returns.  ``adam_steps``: the step counters of the
  generator's and the value net's optimisers, advanced by the loss launch (``HipAdam.step(advanced=True)`` follows)."""
  cfg = gan.cfg
  n = fake_input.shape[0]
  dev = fake_input.device
  f32 = dict(dtype=torch.float32, device=dev)
  with once_differentiable_convnets():
    (fake_output, new_states, surrogate, penalty), debug, _ = gan.generator(
        (fake_input, z, states), is_train=1, progress=progress, dropout_masks=dropout_masks)
  A, B = slice(0, n), slice(n, 2 * n)
  with torch.no_grad():
    fo = fake_output.detach()
    critic = _PairPass(gan.critic, fo, fake_input.to(fo.dtype), x_rows=A)
    value = _PairPass(gan.value, fake_input.to(fo.dtype), fo, states, new_states.detach(), x_rows=B)
    _PairPass.forward_all([critic, value])
    losses = torch.empty((2,), **f32)
    reward, q = torch.empty((n,), **f32), torch.empty((n,), **f32)
    coef = torch.empty((5, n), **f32)
    use_pen = bool(cfg.use_penalty)
    _cabi.generator_losses(critic.logits[A], critic.logits[B], value.logits[B], value.logits[A],
                           new_states.detach().contiguous().float(), penalty.detach().reshape(n).contiguous().float() if use_pen else None,
                           surrogate.detach().reshape(n).contiguous().float(),
                           (cfg.all_reward, cfg.critic_logit_multiplier, cfg.discount_factor, cfg.parameter_lr_mul,
                            cfg.maximum_trajectory_length), bool(cfg.use_TD), losses, reward, q, coef, adam_steps=adam_steps)
    # the value net: old_value's rows carry d v_loss / d old_value (-> theta_v), new_value's rows d g_loss / d new_value
    # (-> the retouched image); its all-reduce (if any) then runs under the rest of the step
    v_grads = _grad_targets(gan, gan.value, 'v')
    d_img = value.backward(torch.cat([coef[4], coef[1]]), slice(0, 2 * n), B, grads=v_grads, rows_w=A)
    gan._bucket_ready(gan.buckets['v'])
    # the critic (frozen): d g_loss / d fake_logit on the retouched rows only
    d_img = d_img + critic.backward(coef[0], A, A)
  # ---- the agent: one autograd pass from (retouched image, surrogate, penalty) to theta_g ---------------------------
  tensors = [fake_output, surrogate] + ([penalty] if use_pen else [])
  gtens = [d_img.to(fake_output.dtype), coef[2].reshape(surrogate.shape)] + ([coef[3].reshape(penalty.shape)] if use_pen else [])
  gan._backward_into(tensors, ['g_head', 'g_trunk'], grad_tensors=gtens)
  shape = (n, 1)
  return dict(g_loss=losses[0], v_loss=losses[1], fake_output=fake_output, new_states=new_states, reward=reward.reshape(shape),
              q_value=q.reshape(shape), fake_logit=critic.logits[A].reshape(shape), debug=debug)
