"""The WGAN-GP critic update as a hand-scheduled sequence of HIP launches (``/root/reference/net.py:126-199, 245-251``;
``critics.py:6-38, 42-98``): no autograd graph, 24 launches instead of 160.

    c_loss = mean(D(fake) - D(real)) + lambda * mean(max(||grad_x^ D(x^)|| - 1, 0)^2),   x^ = real + alpha (fake - real)

The critic is per-sample (no normalisation layers), so the real, the fake and the interpolated images run as ONE batch
``[real | fake | interpolated]`` of 3n images.  With z_l = lrelu(conv(z_{l-1}, W_l) + b_l), slope masks m_l = lrelu'(z_l)
and the rows' upstream gradients dlogit = (-1/n, +1/n, 1) -- the third block is the INNER gradient d D(x^) / d x^, which
``tf.gradients`` starts from ones (net.py:174-183) --

  inputs         [real | fake | interpolated] as float32, statistics,   ONE launch (expo_net_inputs: a block holds its image in LDS)
                 the six planes - 0.5
  forward        z_1 .. z_4, fc1, fc2                                   4 conv launches (bias + lrelu fused) + fc1 with its K dimension
                                                                        split (expo_fc_fwd_slabs) + the head kernel that adds the slabs
  backward       gy_4 = (dh W_fc1) m_4;  gy_{l-1} = D(gy_l, W_l) m_{l-1}  the activation gradient sits in the EPILOGUE of the
                 data-gradient kernel above it (expo_fc_bwd_data_mask, expo_conv4x4s2_bwd_data_mask); the first layer's data gradient only for the
                 interpolated block (6 input planes: conv_bwd_small_kernel on the vector ALUs)
  penalty        g = u_0[..., :3] + J^T sum(u_0[..., 3:]) (the statistics planes, critics.py:48-76), norm, term, the
                 penalty's gradient v with respect to g and the tangent's input [v | J v] in ONE launch
                 (expo_critic_penalty_tangent)
  tangent        the double backward of the penalty is a FORWARD pass of v through the same layers under the same masks:
                 t_0 = [v | J v], t_l = F(t_{l-1}, W_l) m_l -- written IN PLACE over the interpolated block of z_l
                 (expo_conv4x4s2_fwd_mask), which nothing reads any more
  weight grads   after that the activation buffers hold [z_{l-1}(real, fake) | t_{l-1}] and the gradient buffers
                 [gy_l(real, fake) | gy_l(interpolated)]: ONE weight-gradient launch per layer over the 3n "images" yields
                 d c_loss / d W_l including the penalty's second-order term, and the bias gradient (column sums over the
                 first 2n images) comes out of the same launch (expo_conv4x4s2_wrw_bias); the four layers' launches share
                 one grid and one reduce (expo_conv4x4s2_wrw_group); fc1's is expo_fc_wrw.  Biases get no gradient from the
                 penalty: the masks are piecewise constant.  The reporting launch also advances Adam's step counter.

Every quantity equals what ``GAN.critic_losses`` + ``backward`` compute through autograd (tests/test_critic_direct.py holds
the two against each other and against finite differences of the float64 oracle); the summation orders are fixed, so a step
is bit-reproducible.  Gradients are written into ``p.grad`` -- fresh tensors on one rank, the views of the flat gradient
bucket when collectives run (means over the LOCAL shard, averaged by the bucket's all-reduce like everywhere else).
"""
import torch

from . import _cabi
from .replay_memory import PoolRows

LEAK = 0.2  # util.py:225 lrelu(x, leak=0.2), every layer of critics.py


def supported(gan, real_data, fake_output):
  """The default configuration on a ROCm device: Wasserstein critic with the gradient penalty, fp32 weights in
  channels_last order, 3-channel NHWC images whose layers keep an even width down to the last convolution."""
  cfg = gan.cfg
  c = gan.critic
  if cfg.gan != 'w' or not (cfg.gradient_penalty_lambda > 0) or c.num_state_dim != 0:
    return False
  if not (real_data.is_cuda and real_data.dim() == 4 and real_data.shape[-1] == 3 and real_data.shape == fake_output.shape and
          real_data.dtype == fake_output.dtype and real_data.dtype in (torch.float16, torch.float32)):
    return False
  h, w = real_data.shape[1], real_data.shape[2]
  for conv in c.convs:
    wt = conv.weight
    if not (wt.is_cuda and wt.dtype == torch.float32 and tuple(wt.shape[2:]) == (4, 4) and
            wt.permute(0, 2, 3, 1).is_contiguous() and conv.bias is not None and wt.shape[0] % 4 == 0):
      return False
    if h % 2 or w % 2 or (w // 2) % 2:
      return False
    h, w = h // 2, w // 2
  return h * w * c.convs[-1].weight.shape[0] == c.flat and c.fc2.weight.shape[0] == 1


def fc_split(fc, rows):
  """expo_fc_fwd_slabs / expo_fc_bwd_data_mask apply to this nn.Linear on `rows` rows."""
  w = fc.weight
  return (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.shape[0] % 16 == 0 and w.shape[1] % 128 == 0 and
          _cabi.fc_fwd_slabs_count(rows, w.shape[1]) > 0)


def _grad_targets(gan):
  """Where the gradients go: the bucket's views when collectives run, fresh tensors (handed to the optimiser) otherwise."""
  params = list(gan.critic.parameters())
  if gan._collectives():
    b = gan.buckets['c']
    b.zero()  # (re)attaches p.grad as views of the flat buffer; the padding between tensors stays zero
    b.disarm()
    return {id(p): p.grad for p in params}
  gan.buckets['c'].release()
  out = {}
  for p in params:
    g = torch.empty_like(p, memory_format=torch.preserve_format)
    p.grad = g
    out[id(p)] = g
  return out


@torch.no_grad()
def critic_losses_and_grads(gan, real_data, fake_output, alpha, ema=None, adam_step=None):
  """One evaluation of c_loss and of its gradient with respect to theta_c, written to ``p.grad`` of the critic's
  parameters.  -> the dict ``GAN.critic_losses`` returns (c_loss, emd, gradient_norm, gradient_penalty, c_average);
  ``ema`` (a device scalar) is advanced by 0.01 (c_average - ema) in the reporting launch (net.py:165-168), ``adam_step`` (the
  step counter of the update the caller runs next, ``HipAdam.step(advanced=True)``) by one."""
  cfg, critic = gan.cfg, gan.critic
  dev = real_data.device
  n = real_data.shape[0]
  m = 3 * n
  f32 = dict(dtype=torch.float32, device=dev)
  lam = float(cfg.gradient_penalty_lambda)
  inv_n = 1.0 / n
  convs = list(critic.convs)
  grads = _grad_targets(gan)

  # ---- inputs: [real | fake | interpolated] as float32, statistics planes, - 0.5 ----------------------------------
  # (PoolRows: the batch's images are rows of a data set / of the replay memory's pool, read in place)
  real_rows = fake_rows = None
  if isinstance(real_data, PoolRows):
    real_data, real_rows = real_data.pool, real_data.idx
  if isinstance(fake_output, PoolRows):
    fake_output, fake_rows = fake_output.pool, fake_output.idx
  real_data, fake_output = real_data.contiguous(), fake_output.detach().contiguous()
  alpha = alpha.contiguous().float().reshape(n)
  stats = torch.empty((m, 3), **f32)
  acts = [torch.empty((m,) + tuple(real_data.shape[1:3]) + (6,), **f32)]
  if real_data.shape[1] * real_data.shape[2] <= _cabi.NET_INPUTS_MAX_PIXELS:
    xi = torch.empty((n,) + tuple(real_data.shape[1:]), **f32)  # the interpolated images: the penalty's J^T / J v read them
    _cabi.net_inputs(real_data, fake_output, alpha, acts[0], stats, x_out=xi, x_first=2 * n, a_rows=real_rows,
                     b_rows=fake_rows)  # one launch: a block holds its image in LDS
  else:
    x = torch.empty((m,) + tuple(real_data.shape[1:]), **f32)
    xi = x[2 * n:]
    _cabi.gp_inputs(real_data, fake_output, alpha, x[:2 * n], xi, real_rows=real_rows, fake_rows=fake_rows)
    _cabi.critic_stats(x, stats)
    _cabi.planes_concat(x, stats, acts[0], 0.5)
  si = stats[2 * n:]

  # ---- forward ----------------------------------------------------------------------------------------------------------
  for conv in convs:
    a = acts[-1]
    z = torch.empty((m, a.shape[1] // 2, a.shape[2] // 2, conv.weight.shape[0]), **f32)
    _cabi.conv4x4s2_fwd(a, conv.weight, conv.bias, z, 1, LEAK)
    acts.append(z)
  flat = acts[-1].reshape(m, critic.flat)
  hidden = critic.fc1.weight.shape[0]
  logits = torch.empty((m,), **f32)
  h, dh = torch.empty((m, hidden), **f32), torch.empty((m, hidden), **f32)
  split = fc_split(critic.fc1, m) and fc_split(critic.fc1, n)  # fc1 with its K dimension split: the head kernels add the slabs
  if split:
    hpre = torch.empty((_cabi.fc_fwd_slabs_count(m, critic.flat), m, hidden), **f32)
    _cabi.fc_fwd_slabs(flat, critic.fc1.weight, hpre)
    _cabi.critic_head_fwd(hpre, critic.fc2.weight.reshape(hidden), critic.fc2.bias, n, n, n, inv_n, logits, h, dh, LEAK,
                          b1=critic.fc1.bias)
  else:
    hpre = torch.addmm(critic.fc1.bias, flat, critic.fc1.weight.t())
    _cabi.critic_head_fwd(hpre, critic.fc2.weight.reshape(hidden), critic.fc2.bias, n, n, n, inv_n, logits, h, dh, LEAK)

  # ---- backward of the three blocks at once (the interpolated block's upstream gradient is 1: the inner gradient) --
  gys = [None] * (len(convs) + 1)  # gys[l]: the gradient in front of layer l's activation (l = 1 .. L)
  gy = torch.empty_like(acts[-1])
  if split:
    _cabi.fc_bwd_data_mask(dh, critic.fc1.weight, acts[-1], gy, LEAK)  # (dh W) slope(z_L): GEMM + activation gradient
  else:
    dz = torch.mm(dh, critic.fc1.weight)  # (3n, flat)
    _cabi.lrelu_bwd(acts[-1], dz.reshape(acts[-1].shape), gy, LEAK)
  gys[len(convs)] = gy
  for l in range(len(convs), 1, -1):
    below = acts[l - 1]
    g = torch.empty_like(below)
    _cabi.conv4x4s2_bwd_data_mask(gys[l], convs[l - 1].weight, below, g, LEAK)
    gys[l - 1] = g
  u0 = torch.empty((n,) + tuple(acts[0].shape[1:]), **f32)
  _cabi.conv4x4s2_bwd_data(gys[1][2 * n:], convs[0].weight, u0)

  # ---- the penalty on g = d D(x^) / d x^, its gradient v with respect to g, and the tangent's input [v | J v] written over
  # the interpolated block of the first activation buffer: one launch (plane sums, J^T, norm / term, v, J v, planes) ------
  norm, term = torch.empty((n,), **f32), torch.empty((n,), **f32)
  _cabi.critic_penalty_tangent(u0, xi, si, lam * inv_n, acts[0][2 * n:], norm, term)

  # ---- tangent pass (the double backward), in place over the interpolated block of every activation ----------------
  for l, conv in enumerate(convs, start=1):
    zi = acts[l][2 * n:]
    _cabi.conv4x4s2_fwd_mask(acts[l - 1][2 * n:], conv.weight, zi, zi, LEAK)
  tflat = acts[-1][2 * n:].reshape(n, critic.flat)
  if split:
    thpre = torch.empty((_cabi.fc_fwd_slabs_count(n, critic.flat), n, hidden), **f32)
    _cabi.fc_fwd_slabs(tflat, critic.fc1.weight, thpre)
  else:
    thpre = torch.mm(tflat, critic.fc1.weight.t())  # (n, hidden)

  # ---- gradients: one launch per layer over [loss rows | penalty rows] -----------------------------------------------
  _cabi.conv4x4s2_wrw_group([(acts[l - 1], gys[l], grads[id(conv.weight)], grads[id(conv.bias)], 2 * n)
                             for l, conv in enumerate(convs, start=1)])  # (one reduce launch for the four layers)
  if split:
    _cabi.fc_wrw(dh, flat, grads[id(critic.fc1.weight)])  # dh^T flat: the batch is this GEMM's K dimension
  else:
    torch.mm(dh.t(), flat, out=grads[id(critic.fc1.weight)])
  _cabi.critic_head_bwd(dh, h, thpre, n, n, n, inv_n, grads[id(critic.fc1.bias)],
                        grads[id(critic.fc2.weight)].reshape(hidden), grads[id(critic.fc2.bias)], LEAK)

  # ---- reported values (net.py:188-199) and, on one rank, the logit centre's moving average in the same launch ----
  rep = torch.empty((5,), **f32)
  _cabi.critic_report(logits, norm, term, n, n, n, lam, rep, ema, 0.99, adam_step=adam_step)
  return dict(c_loss=rep[0], emd=rep[1], gradient_norm=rep[2], gradient_penalty=rep[3], c_average=rep[4])
