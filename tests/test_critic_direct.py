"""The hand-scheduled critic update (exposure_amd/critic_direct.py, csrc/critic_step.hip, the mask / bias variants of the
convolution kernels) -- net.py:126-199, 245-251; critics.py:6-38, 42-98:

* every new kernel against a float64 statement of what it computes;
* the whole update against the autograd path (``GAN(direct_critic=False)``: loss values, every gradient tensor) and
  against finite differences of the float64 NumPy oracle (oracle/nets_np.py), eager and replayed from a hipGraph."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN

pytestmark = pytest.mark.gpu


def _slope(z, leak=0.2):
  return torch.where(z > 0, torch.ones_like(z), torch.where(z < 0, torch.full_like(z, leak), torch.full_like(z, 0.5 * (1 + leak))))


def _case(n, h, cin, cout, dev, seed):
  g = torch.Generator(device=dev).manual_seed(seed)
  x = torch.randn((n, h, h, cin), device=dev, generator=g)
  w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) / (16 * cin)**0.5).contiguous(memory_format=torch.channels_last)
  gy = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=g)
  return x, w, gy


def _ref_dgrad(x_shape, w, gy):
  n, h, _, cin = x_shape
  xd = torch.zeros((n, cin, h, h), dtype=torch.float64, requires_grad=True)
  yd = F.conv2d(xd, w.double().cpu(), None, 2, 1)
  ref, = torch.autograd.grad(yd, [xd], gy.double().cpu().permute(0, 3, 1, 2))
  return ref.permute(0, 2, 3, 1)


CASES = [(3, 8, 5, 8), (2, 16, 14, 32), (5, 12, 6, 32), (2, 64, 17, 32), (64, 64, 6, 32), (9, 8, 32, 64), (4, 16, 64, 128),
         (16, 8, 128, 256), (1, 2, 6, 4), (3, 4, 17, 8), (7, 4, 4, 36)]


@pytest.mark.parametrize('case', CASES)
def test_data_gradient_with_the_activation_gradient_in_its_epilogue(case, gpu_device):
  """expo_conv4x4s2_bwd_data_mask == D(dy, w) * slope(z) with z of either sign and exactly 0 (TF's sub-gradient 0.6);
  6 input planes take the vector-ALU kernel (conv_bwd_small_kernel), the rest the matrix-core kernel (17 planes since the
  end of round 6); the plain entry point agrees (no mask) and every element is written."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, gy = _case(n, h, cin, cout, dev, seed=n + h + cin)
  z = torch.randn(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(n + cin))
  z.view(-1)[::7] = 0.0
  ref = _ref_dgrad(x.shape, w, gy)
  scale = float(ref.abs().max())
  dx = torch.full_like(x, float('nan'))
  _cabi.conv4x4s2_bwd_data(gy, w, dx)
  assert float((dx.double().cpu() - ref).abs().max()) / scale < 3e-6
  dm = torch.full_like(x, float('nan'))
  _cabi.conv4x4s2_bwd_data_mask(gy, w, z, dm, 0.2)
  assert torch.equal(dm, dx * _slope(z))
  if cin in (6, 17):  # the same problem through the matrix-core kernel (a forced plan): both kernels must agree
    _cabi.conv_tuning(0, 1, 1)
    try:
      d2 = torch.full_like(x, float('nan'))
      _cabi.conv4x4s2_bwd_data(gy, w, d2)
    finally:
      _cabi.conv_tuning(0, 0, 0)
    assert float((d2 - dx).abs().max()) / scale < 3e-6


@pytest.mark.parametrize('case', CASES)
def test_forward_with_the_slope_mask_in_place(case, gpu_device):
  """expo_conv4x4s2_fwd_mask == conv(x, w) * slope(z), into a separate tensor and written over z itself, under every
  decomposition of the forward kernels."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, gy = _case(n, h, cin, cout, dev, seed=n + h)
  z = torch.randn(gy.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(n + cout))
  z.view(-1)[::5] = 0.0
  ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, 2, 1).permute(0, 2, 3, 1) * _slope(z).double().cpu()
  scale = float(ref.abs().max())
  try:
    for tile, nt, sl in [(0, 0, 0), (5, 1, 1), (5, 2, 4), (1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0)]:
      _cabi.conv_tuning(tile, nt, sl)
      y = torch.full_like(z, float('nan'))
      _cabi.conv4x4s2_fwd_mask(x, w, z, y, 0.2)
      # (one wave alone sums K = 2048 terms in f32 under the forced single-slice plans: 2.2e-6 of the largest output)
      assert float((y.double().cpu() - ref).abs().max()) / scale < 3e-6, (case, tile, nt, sl)
      zz = z.clone()
      _cabi.conv4x4s2_fwd_mask(x, w, zz, zz, 0.2)
      assert torch.equal(zz, y), (case, tile, nt, sl)
  finally:
    _cabi.conv_tuning(0, 0, 0)


@pytest.mark.parametrize('case', [(3, 8, 5, 8), (2, 16, 14, 32), (6, 64, 6, 32), (2, 64, 17, 32), (9, 8, 32, 64), (6, 16, 64, 128),
                                  (16, 8, 128, 256), (7, 4, 4, 36)])
def test_weight_gradient_with_the_bias_gradient(case, gpu_device):
  """expo_conv4x4s2_wrw_bias: dw as expo_conv4x4s2_wrw, dbias = the column sums of dy over the first k images (k = 0,
  part of the batch, all of it), under several splits of the pixel sum; bit-reproducible."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, gy = _case(n, h, cin, cout, dev, seed=n + h)
  wd = w.double().cpu().requires_grad_(True)
  yd = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wd, None, 2, 1)
  ref, = torch.autograd.grad(yd, [wd], gy.double().cpu().permute(0, 3, 1, 2))
  scale = float(ref.abs().max())
  try:
    for sl, parts in ((0, 0), (1, 1), (2, 3), (4, 0), (4, 7), (3, 2)):
      _cabi.conv_wrw_tuning(sl, parts)
      for k in (n, 0, (2 * n) // 3):
        dw, db = torch.full_like(w, float('nan')), torch.full((cout,), float('nan'), device=dev)
        _cabi.conv4x4s2_wrw_bias(x, gy, dw, db, k)
        assert float((dw.double().cpu() - ref).abs().max()) / scale < 1e-5, (case, sl, parts)
        want = gy[:k].double().sum(dim=(0, 1, 2)).cpu()
        tol = 1e-5 * float(gy[:k].double().abs().sum(dim=(0, 1, 2)).max()) + 1e-30
        assert float((db.double().cpu() - want).abs().max()) <= tol, (case, sl, parts, k)
        dw2, db2 = torch.empty_like(w), torch.empty_like(db)
        _cabi.conv4x4s2_wrw_bias(x, gy, dw2, db2, k)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
  finally:
    _cabi.conv_wrw_tuning(0, 0)


def test_conv_tuning_rejects_slice_counts_the_kernels_cannot_cut(gpu_device):
  """3, 5 .. 7 K slices would silently drop K segments (advisor, round 5): rejected at the boundary."""
  from exposure_amd import _cabi
  for bad in (3, 5, 6, 7, 12, 17):
    with pytest.raises(_cabi.ExposureHipError):
      _cabi.conv_tuning(0, 0, bad)
  _cabi.conv_tuning(0, 0, 0)


@pytest.mark.parametrize('rows', [(4, 4, 4), (64, 64, 64), (3, 5, 0), (0, 0, 7)])
def test_head_kernels(rows, gpu_device):
  from exposure_amd import _cabi
  dev = gpu_device
  nr, nf, ni = rows
  m, hidden = nr + nf + ni, 128
  g = torch.Generator(device=dev).manual_seed(m)
  hpre = torch.randn((m, hidden), device=dev, generator=g)
  hpre.view(-1)[::11] = 0.0
  w2, b2 = torch.randn((hidden,), device=dev, generator=g), torch.randn((1,), device=dev, generator=g)
  inv_n = 1.0 / max(nr, 1)
  logits, h, dh = torch.empty((m,), device=dev), torch.empty_like(hpre), torch.empty_like(hpre)
  _cabi.critic_head_fwd(hpre, w2, b2, nr, nf, ni, inv_n, logits, h, dh)
  hd = torch.where(hpre > 0, hpre, 0.2 * hpre).double()
  assert torch.equal(h, hd.float())
  want_logits = hd @ w2.double() + b2.double()
  assert float((logits.double() - want_logits).abs().max()) < 1e-5 * max(1.0, float(want_logits.abs().max()))
  dl = torch.cat([torch.full((nr,), -inv_n), torch.full((nf,), inv_n), torch.ones(ni)]).to(dev).double()
  want_dh = dl[:, None] * w2.double()[None, :] * _slope(h).double()
  assert float((dh.double() - want_dh).abs().max()) < 1e-6 * max(1.0, float(want_dh.abs().max()))
  norm, term = torch.rand((ni,), device=dev, generator=g) + 0.5, torch.rand((ni,), device=dev, generator=g)
  rep, ema = torch.full((5,), float('nan'), device=dev), torch.full((1,), 0.25, device=dev)
  _cabi.critic_report(logits, norm, term, nr, nf, ni, 10.0, rep, ema, 0.99)
  mean = lambda v: float(v.double().mean()) if v.numel() else 0.0
  mr, mf, gp = mean(logits[:nr]), mean(logits[nr:nr + nf]), 10.0 * mean(term)
  want_rep = [mf - mr + gp, mr - mf, mean(norm), gp, 0.5 * (mf + mr)]
  for a, b in zip(rep.tolist(), want_rep):
    assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (rep.tolist(), want_rep)
  assert abs(float(ema) - (0.25 + 0.01 * (want_rep[4] - 0.25))) < 1e-6
  thpre = torch.randn((ni, hidden), device=dev, generator=g)
  gb1, gw2, gb2 = (torch.full((k,), float('nan'), device=dev) for k in (hidden, hidden, 1))
  _cabi.critic_head_bwd(dh, h, thpre, nr, nf, ni, inv_n, gb1, gw2, gb2)
  nl = nr + nf
  want_gb1 = want_dh[:nl].sum(0)
  want_gw2 = (dl[:nl, None] * hd[:nl]).sum(0) + (thpre.double() * _slope(h[nl:]).double()).sum(0)
  assert float((gb1.double() - want_gb1).abs().max()) <= 1e-5 * max(1e-3, float(want_dh[:nl].abs().sum(0).max()) if nl else 1e-3)
  assert float((gw2.double() - want_gw2).abs().max()) <= 1e-5 * max(1.0, float(want_gw2.abs().max()))
  assert abs(float(gb2) - float(dl[:nl].sum())) < 1e-6


@pytest.mark.parametrize('shape', [(5, 64, 64, 6), (3, 7, 5, 6), (2, 4, 4, 17)])
def test_plane_sums_and_gp_direct(shape, gpu_device):
  from exposure_amd import _cabi
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(shape[1])
  u = torch.randn(shape, device=dev, generator=g) * 0.02
  n, c = shape[0], shape[-1]
  sums = torch.empty((n, c - 3), device=dev)
  _cabi.plane_sums(u, sums, 3)
  want = u.double()[..., 3:].sum(dim=(1, 2))
  assert float((sums.double() - want).abs().max()) <= 1e-5 * float(u.double()[..., 3:].abs().sum(dim=(1, 2)).max())
  ds = torch.randn(shape[:-1] + (3,), device=dev, generator=g) * 0.02
  if n > 1:
    u[1] *= 0.01  # an image whose norm stays below 1: the one-sided penalty and its gradient vanish
    ds[1] *= 0.01
  v, norm, term = torch.empty_like(ds), torch.empty((n,), device=dev), torch.empty((n,), device=dev)
  _cabi.gp_direct(u, ds, 0.37, v, norm, term)
  gd = (u[..., :3] + ds).double().requires_grad_(True)
  nm = torch.sqrt(1e-6 + (gd**2).sum(dim=(1, 2, 3)))
  tm = torch.clamp_min(nm - 1.0, 0.0)**2
  vd, = torch.autograd.grad(0.37 * tm.sum(), [gd])
  assert float((norm.double() - nm).abs().max()) < 1e-5 * float(nm.max())
  assert float((term.double() - tm.detach()).abs().max()) < 1e-4 * max(1e-6, float(tm.max()))
  assert float((v.double() - vd).abs().max()) <= 1e-4 * max(1e-9, float(vd.abs().max()))
  if n > 1:
    assert float(term[1]) == 0.0 and float(v[1].abs().max()) == 0.0


def _make_gan(dev, seed, **kw):
  torch.manual_seed(seed)
  gan = GAN(make_cfg(), device=dev, **kw)
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)  # gradient norm > 1: the one-sided penalty is active
  return gan


@pytest.mark.parametrize('n,dtype', [(8, torch.float32), (5, torch.float16), (64, torch.float16)])
def test_direct_critic_update_matches_autograd(n, dtype, gpu_device):
  """Loss values and EVERY gradient tensor of theta_c: the hand-scheduled update against ``critic_losses`` + backward."""
  from exposure_amd import critic_direct
  from tests.test_oracle_nets import make_batch
  dev = gpu_device
  gan = _make_gan(dev, 3)
  fake_input, real, _s, _z, _m, alpha = make_batch(n, 17)
  t = lambda a: torch.from_numpy(a).to(dev)
  real_t, fake_t, alpha_t = t(real).to(dtype), t(fake_input).to(dtype), t(alpha)
  assert critic_direct.supported(gan, real_t, fake_t)
  out = critic_direct.critic_losses_and_grads(gan, real_t, fake_t, alpha_t)
  got = {name: p.grad.detach().clone() for name, p in gan.critic.named_parameters()}
  for p in gan.critic.parameters():
    p.grad = None
  ref = gan.critic_losses(real_t, fake_t, alpha_t)
  ref['c_loss'].backward()
  assert float(ref['gradient_norm']) > 1.0
  for key in ('c_loss', 'emd', 'gradient_norm', 'gradient_penalty', 'c_average'):
    a, b = float(out[key]), float(ref[key].detach())
    assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (key, a, b)
  for name, p in gan.critic.named_parameters():
    a, b = got[name], p.grad
    assert a.shape == b.shape
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= 2e-4 * scale + 1e-9, (name, err, scale)


def test_direct_critic_gradients_against_finite_differences_of_the_oracle(gpu_device):
  """The gradients the hand-scheduled update hands to Adam, along their own direction, against central differences of
  the float64 NumPy critic loss (oracle/nets_np.py::critic_losses) -- every variable of the critic."""
  from exposure_amd import checkpoint, critic_direct
  from oracle import nets_np as nn_np
  from tests.test_oracle_nets import make_batch
  dev = gpu_device
  gan = _make_gan(dev, 11)
  n = 8
  fake_input, real, _s, _z, _m, alpha = make_batch(n, 11)
  t = lambda a: torch.from_numpy(a).to(dev)
  d = lambda a: a.astype(np.float64)
  out = critic_direct.critic_losses_and_grads(gan, t(real), t(fake_input), t(alpha))
  cfg = dict(nn_np.DEFAULT_CFG)
  weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
  rc = nn_np.critic_losses(d(real), d(fake_input), d(alpha), cfg, weights)
  for key in ('c_loss', 'emd', 'gradient_norm', 'gradient_penalty', 'c_average'):
    assert abs(float(out[key]) - rc[key]) <= 1e-4 * max(1.0, abs(rc[key])), (key, float(out[key]), rc[key])
  loss = lambda w: nn_np.critic_losses(d(real), d(fake_input), d(alpha), cfg, w)['c_loss']
  checked = 0
  for name, p, kind in checkpoint.tf_name_map(gan):
    if not name.startswith('critic/'):
      continue
    g = checkpoint.to_tf_layout(p.grad, kind).astype(np.float64)
    gnorm = float(np.sqrt((g**2).sum()))
    if name.endswith('fully_connected_1/biases'):  # d c_loss / d b2 = sum of dlogit = 0 exactly
      assert gnorm < 1e-6
      continue
    assert gnorm > 0, name
    w0 = weights[name]
    s = max(float(np.abs(w0).std()), 0.02)
    direction = g / gnorm * np.sqrt(g.size)
    got = float((g * direction).sum())
    for hstep in (1e-6 * s, 2.5e-7 * s, 6e-8 * s):
      fd = (loss(dict(weights, **{name: w0 + hstep * direction})) - loss(dict(weights, **{name: w0 - hstep * direction}))) / (2 * hstep)
      if abs(got - fd) <= 1e-3 * abs(fd):
        break
    assert abs(got - fd) <= 1e-3 * abs(fd), (name, got, fd)
    checked += 1
  assert checked >= 11


def test_direct_critic_step_replayed_from_a_graph_trains_like_the_autograd_step(gpu_device):
  """Three critic steps (eager warm-up, capture, replay) with the hand-scheduled update == the same three steps through
  autograd: the weights after Adam agree, and the logit-centre average advances."""
  from tests.test_oracle_nets import make_batch
  dev = gpu_device
  gans = [_make_gan(dev, 5, use_graphs=True, direct_critic=flag) for flag in (True, False)]
  gans[1].load_state_dict(gans[0].state_dict())
  t = lambda a: torch.from_numpy(a).to(dev)
  for step in range(3):
    fake_input, real, _s, _z, _m, alpha = make_batch(8, 30 + step)
    outs = [g.critic_step(t(real).half(), t(fake_input).half(), it=1, alpha=t(alpha)) for g in gans]
    # (identical weights at step 0; afterwards the two runs' weights differ by Adam's response to rounding-level
    # gradient differences, and c_loss -- fc2 scaled by 40 -- is sensitive to them)
    tol = 2e-5 if step == 0 else 2e-3
    assert abs(float(outs[0]['c_loss']) - float(outs[1]['c_loss'])) <= tol * max(1.0, abs(float(outs[1]['c_loss']))), step
  torch.cuda.synchronize()
  for (name, a), (_, b) in zip(gans[0].critic.named_parameters(), gans[1].critic.named_parameters()):
    # Adam normalises the gradient: early steps move every weight by ~lr whatever the gradient's size, so an element whose
    # gradient is rounding-level noise around zero can differ by up to 2 lr per step between the two paths; on average the
    # weights stay within a small fraction of a step
    lr = float(gans[0].cfg.lr_c(1))
    assert float((a - b).abs().max()) <= 2.0 * 3 * lr * 1.01, name
    assert float((a - b).abs().mean()) <= 0.05 * 3 * lr, name  # (measured: <= 0.02 of a step on the 32-element biases)
  assert abs(float(gans[0].c_average_biased) - float(gans[1].c_average_biased)) < 1e-5


@pytest.mark.parametrize('cin,size,n,frozen', [(6, 64, 5, False), (14, 64, 4, False), (17, 32, 3, False), (17, 64, 4, True)])
def test_conv_trunk_node_matches_the_layerwise_path(cin, size, n, frozen, gpu_device):
  """nn_ops.conv_trunk as ONE once-differentiable node (activation gradients in the data-gradient epilogues, bias
  gradients out of the weight-gradient launches) against the same stack layer by layer through ``conv_bias_lrelu``:
  output, input gradient, every weight / bias gradient; frozen parameters receive none."""
  from exposure_amd import nn_ops
  dev = gpu_device
  torch.manual_seed(cin + size)
  chans = [cin, 32, 64, 128, 256]
  convs = torch.nn.ModuleList([torch.nn.Conv2d(chans[i], chans[i + 1], 4, 2, 1) for i in range(4)]).to(dev)
  convs = convs.to(memory_format=torch.channels_last)
  with torch.no_grad():
    for c in convs:
      c.bias.normal_(0.0, 0.1)
  x0 = torch.randn((n, size, size, cin), device=dev)
  gz = torch.randn((n, size // 16, size // 16, 256), device=dev)
  res = []
  for fused in (True, False):
    x = x0.clone().requires_grad_(True)
    for c in convs:
      c.weight.grad = c.bias.grad = None
    if fused:
      with nn_ops.once_differentiable_convnets():
        if frozen:
          with nn_ops.frozen_parameters():
            z = nn_ops.conv_trunk(x, convs)
        else:
          z = nn_ops.conv_trunk(x, convs)
      assert type(z.grad_fn).__name__ == '_ConvTrunkBackward'
    else:
      z = x
      for c in convs:
        z = nn_ops.conv_bias_lrelu(z, c.weight.detach() if frozen else c.weight, c.bias.detach() if frozen else c.bias)
    z.backward(gz)
    res.append([z.detach(), x.grad] + [c.weight.grad for c in convs] + [c.bias.grad for c in convs])
  for i, (a, b) in enumerate(zip(*res)):
    if b is None:
      assert a is None, i
      continue
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= 2e-5 * scale + 1e-9, (i, err, scale)


def test_grouped_weight_gradients_equal_the_separate_calls(gpu_device):
  """expo_conv4x4s2_wrw_group (one reduce launch for a stack of layers) == the same layers one call each, bit for bit,
  including a layer small enough to need no reduce at all and one without a bias gradient."""
  from exposure_amd import _cabi
  dev = gpu_device
  cases = [(6, 64, 6, 32), (6, 32, 32, 64), (6, 16, 64, 128), (6, 8, 128, 256), (1, 4, 4, 8)]
  items, want = [], []
  for k, (n, h, cin, cout) in enumerate(cases):
    x, w, gy = _case(n, h, cin, cout, dev, seed=k)
    dw, db = torch.full_like(w, float('nan')), (torch.full((cout,), float('nan'), device=dev) if k != 2 else None)
    items.append((x, gy, dw, db, n - 1 if k == 0 else None))
    rw, rb = torch.empty_like(w), torch.empty((cout,), device=dev)
    _cabi.conv4x4s2_wrw_bias(x, gy, rw, rb, n - 1 if k == 0 else None)
    want.append((rw, rb))
  _cabi.conv4x4s2_wrw_group(items)
  for (x, gy, dw, db, _), (rw, rb) in zip(items, want):
    assert torch.equal(dw, rw)
    assert db is None or torch.equal(db, rb)


@pytest.mark.parametrize('shape', [(5, 64, 64), (3, 7, 5), (2, 2, 2), (2, 80, 72)])  # (> 4096 pixels: the three-walk kernel)
def test_penalty_tangent_kernel_equals_its_six_launch_composition(shape, gpu_device):
  """expo_critic_penalty_tangent == plane sums -> expo_critic_stats_bwd -> expo_gp_direct -> expo_critic_stats_jvp ->
  expo_planes_concat(offset 0): the same norm / term and the same 6-plane tangent input, one image with its norm below 1
  (no penalty, a zero tangent)."""
  from exposure_amd import _cabi
  dev = gpu_device
  n, h, w = shape
  g = torch.Generator(device=dev).manual_seed(h)
  x = torch.rand((n, h, w, 3), device=dev, generator=g) * 1.2 - 0.05
  u = torch.randn((n, h, w, 6), device=dev, generator=g) * (2.0 / (h * w * 3)**0.5)
  if n > 1:
    u[1] *= 0.01
  stats = torch.empty((n, 3), device=dev)
  _cabi.critic_stats(x, stats)
  gs, ds = torch.empty((n, 3), device=dev), torch.empty_like(x)
  _cabi.plane_sums(u, gs, 3)
  _cabi.critic_stats_bwd(x, stats, gs, ds)
  v, norm, term = torch.empty_like(x), torch.empty((n,), device=dev), torch.empty((n,), device=dev)
  _cabi.gp_direct(u, ds, 0.3, v, norm, term)
  jv = torch.empty((n, 3), device=dev)
  _cabi.critic_stats_jvp(x, stats, v, jv)
  want = torch.empty_like(u)
  _cabi.planes_concat(v, jv, want, 0.0)
  t0, norm2, term2 = torch.full_like(u, float('nan')), torch.empty_like(norm), torch.empty_like(term)
  _cabi.critic_penalty_tangent(u, x, stats, 0.3, t0, norm2, term2)
  assert float((norm2 - norm).abs().max()) <= 2e-6 * float(norm.max())
  assert float((term2 - term).abs().max()) <= 1e-5 * max(1e-6, float(term.max()))
  scale = max(1e-12, float(want.abs().max()))
  assert float((t0 - want).abs().max()) <= 2e-5 * scale, float((t0 - want).abs().max()) / scale
  if n > 1:
    assert float(term2[1]) == 0.0 and float(t0[1].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape,v0,interp,rows', [((5, 64, 64), 0, True, True), ((4, 64, 64), 11, False, False),
                                                    ((3, 7, 5), 0, True, False), ((2, 6, 3), 2, False, True),
                                                    ((64, 64, 64), 0, True, True), ((1, 1, 1), 0, False, False)])
def test_net_inputs_equals_its_four_launch_composition(shape, v0, interp, rows, dtype, gpu_device):
  """expo_net_inputs == expo_gp_inputs_rows -> expo_critic_stats -> cat of the per-image values -> expo_planes_concat: the
  float32 images of the requested rows and the image planes bit for bit, the statistics (another summation order) to float32
  rounding of sums of h w terms -- against the float64 formulas of critics.py:48-62 as well."""
  from exposure_amd import _cabi
  dev = gpu_device
  n, h, w = shape
  g = torch.Generator(device=dev).manual_seed(17 * n + h)
  pool_a = (torch.rand((n + 3, h, w, 3), device=dev, generator=g) * 1.2 - 0.05).to(dtype)
  pool_b = (torch.rand((n + 2, h, w, 3), device=dev, generator=g) * 1.2 - 0.05).to(dtype)
  ra = torch.randperm(n + 3, device=dev, generator=g)[:n].contiguous() if rows else None
  rb = torch.randperm(n + 2, device=dev, generator=g)[:n].contiguous() if rows else None
  a, b = (pool_a, pool_b) if rows else (pool_a[:n].contiguous(), pool_b[:n].contiguous())
  alpha = torch.rand((n,), device=dev, generator=g) if interp else None
  va = torch.randn((n, v0), device=dev, generator=g) if v0 else None
  vb = torch.randn((n, v0), device=dev, generator=g) if v0 else None
  m = (3 if interp else 2) * n
  # the separate launches
  x = torch.empty((m, h, w, 3), device=dev)
  _cabi.gp_inputs(a, b, alpha, x[:2 * n], x[2 * n:] if interp else None, real_rows=ra, fake_rows=rb)
  stats = torch.empty((m, 3), device=dev)
  _cabi.critic_stats(x, stats)
  vec = stats if not v0 else torch.cat([torch.cat([va, vb], dim=0), stats], dim=1).contiguous()
  want = torch.empty((m, h, w, 6 + v0), device=dev)
  _cabi.planes_concat(x, vec, want, 0.5)
  # one launch; the float32 images of the last block and of one row in front of it
  planes = torch.full_like(want, float('nan'))
  stats2 = torch.full_like(stats, float('nan'))
  x_first = m - n - 1 if m > n else 0
  xo = torch.full((m - x_first, h, w, 3), float('nan'), device=dev)
  _cabi.net_inputs(a, b, alpha, planes, stats2, x_out=xo, x_first=x_first, vec_a=va, vec_b=vb, a_rows=ra, b_rows=rb)
  assert torch.equal(xo, x[x_first:])
  assert torch.equal(planes[..., :3 + v0], want[..., :3 + v0])
  xd = x.double().cpu().reshape(m, h * w, 3)
  lum = xd[..., 0] * 0.27 + xd[..., 1] * 0.67 + xd[..., 2] * 0.06 + 1e-5
  c = xd.clamp(0, 1)
  mx, mn = c.max(dim=2).values, c.min(dim=2).values
  sat = (mx - mn) / (torch.minimum(mx + mn, 2.0 - mx - mn) + 1e-2)
  ref = torch.stack([lum.mean(1), lum.var(1, unbiased=False), sat.mean(1)], dim=1)
  tol = torch.tensor([2e-6, 2e-6, 4e-6], dtype=torch.float64)
  assert bool(((stats2.double().cpu() - ref).abs() <= tol).all()), (stats2.double().cpu() - ref).abs().max(0).values
  assert bool(((stats2 - stats).abs().cpu() <= tol.float()).all())
  assert torch.equal(planes[..., 3 + v0:], (stats2 - 0.5)[:, None, None, :].expand(m, h, w, 3))


@pytest.mark.parametrize('m,n,k', [(192, 128, 4096), (64, 128, 4096), (128, 128, 4096), (37, 96, 1024), (5, 32, 128), (200, 128, 384)])
def test_fc_layer_with_its_features_split_and_its_data_gradient(m, n, k, gpu_device):
  """expo_fc_fwd_slabs: the slabs' sum == x w^T (float64) and the head kernels that consume them equal the head kernels on
  the finished pre-activation; expo_fc_bwd_data_mask == (dh w) slope(z), expo_fc_wrw == dh^T x (float64)."""
  from exposure_amd import _cabi
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(m + k)
  x = torch.randn((m, k), device=dev, generator=g)
  w = torch.randn((n, k), device=dev, generator=g) / k**0.5
  b1 = torch.randn((n,), device=dev, generator=g)
  s = _cabi.fc_fwd_slabs_count(m, k)
  assert s >= 1 and k % (128 * s) == 0
  assert _cabi.fc_fwd_slabs_count(m, 4000) == 0
  slabs = torch.full((s, m, n), float('nan'), device=dev)
  _cabi.fc_fwd_slabs(x, w, slabs)
  want = x.double() @ w.double().t()
  assert float((slabs.double().sum(0) - want).abs().max()) <= 2e-6 * float(want.abs().max()) * (k / 128)**0.5
  # the consumers: slabs + b1 against the finished pre-activation
  hpre = (slabs.double().sum(0) + b1.double()).float()
  w2, b2 = torch.randn((n,), device=dev, generator=g), torch.randn((1,), device=dev, generator=g)
  nr = nf = m // 3
  ni = m - nr - nf
  outs = []
  for src, kw in ((hpre, {}), (slabs, dict(b1=b1))):
    logits, h, dh = torch.empty((m,), device=dev), torch.empty((m, n), device=dev), torch.empty((m, n), device=dev)
    _cabi.critic_head_fwd(src, w2, b2, nr, nf, ni, 1.0 / max(nr, 1), logits, h, dh, **kw)
    outs.append((logits, h, dh))
  for a, b in zip(*outs):
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
  th = torch.randn((s, ni, n), device=dev, generator=g)
  res = []
  for src in (th.double().sum(0).float().contiguous(), th):
    gb1, gw2, gb2 = (torch.full((q,), float('nan'), device=dev) for q in (n, n, 1))
    _cabi.critic_head_bwd(outs[0][2], outs[0][1], src, nr, nf, ni, 1.0 / max(nr, 1), gb1, gw2, gb2)
    res.append((gb1, gw2, gb2))
  for a, b in zip(*res):
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
  # data gradient with the activation gradient of the map below (z with exact zeros and negative values)
  if n % 16 == 0 and k % 128 == 0:
    dh = torch.randn((m, n), device=dev, generator=g)
    z = torch.randn((m, k), device=dev, generator=g)
    z.view(-1)[::7] = 0.0
    gy = torch.full((m, k), float('nan'), device=dev)
    _cabi.fc_bwd_data_mask(dh, w, z, gy)
    want = (dh.double() @ w.double()) * _slope(z).double()
    assert float((gy.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * (n / 16)**0.5
  # weight gradient: the batch is the GEMM's K dimension
  dh = torch.randn((m, n), device=dev, generator=g)
  dw = torch.full((n, k), float('nan'), device=dev)
  _cabi.fc_wrw(dh, x, dw)
  want = dh.double().t() @ x.double()
  assert float((dw.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * max(1.0, m / 16)**0.5


def test_direct_critic_update_on_its_separate_launches(gpu_device, monkeypatch):
  """Images beyond 4096 pixels, or an FC layer whose width the split kernels do not take, run the update's input side as
  gp_inputs -> stats -> planes_concat and fc1 through the library GEMM: the same losses and gradients as the fused launches."""
  from exposure_amd import _cabi, critic_direct
  from tests.test_oracle_nets import make_batch
  dev = gpu_device
  gan = _make_gan(dev, 3)
  fake_input, real, _s, _z, _m, alpha = make_batch(8, 17)
  t = lambda a: torch.from_numpy(a).to(dev)
  real_t, fake_t, alpha_t = t(real).half(), t(fake_input).half(), t(alpha)
  outs = []
  for fused in (True, False):
    if not fused:
      monkeypatch.setattr(_cabi, 'NET_INPUTS_MAX_PIXELS', 0)
      monkeypatch.setattr(critic_direct, 'fc_split', lambda fc, rows: False)
    out = critic_direct.critic_losses_and_grads(gan, real_t, fake_t, alpha_t)
    outs.append(({k: float(v) for k, v in out.items()},
                 {name: p.grad.detach().clone() for name, p in gan.critic.named_parameters()}))
  for key, a in outs[0][0].items():
    assert abs(a - outs[1][0][key]) <= 2e-5 * max(1.0, abs(a)), key
  for name, a in outs[0][1].items():
    b = outs[1][1][name]
    assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9, name


@pytest.mark.parametrize('case', [(64, 64, 14, 32), (8, 64, 17, 32), (64, 32, 32, 64), (128, 16, 64, 128), (64, 8, 128, 256),
                                  (5, 12, 6, 32), (3, 8, 5, 8)])
def test_pair_launches_equal_the_two_calls(case, gpu_device):
  """expo_conv4x4s2_fwd_pair / _bwd_data_mask_pair: two problems of one geometry as one grid (gridDim.y = 2) == the two
  separate calls up to the order of the K slices' sum (the pair is planned for the grid that runs: twice the blocks), and
  bit-reproducible -- every forward kernel family (row-staged first layers, LDS-tiled, flat) and the flat data gradient,
  the same and different inputs."""
  from exposure_amd import _cabi
  dev = gpu_device
  n, h, cin, cout = case
  xa, wa, gya = _case(n, h, cin, cout, dev, 5)
  xb, wb, gyb = _case(n, h, cin, cout, dev, 6)
  g = torch.Generator(device=dev).manual_seed(7)
  ba, bb = torch.randn((cout,), device=dev, generator=g), torch.randn((cout,), device=dev, generator=g)
  shape = (n, h // 2, h // 2, cout)

  def close(got, want):
    return float((got - want).abs().max()) <= 4e-6 * float(want.abs().max())

  for xb_ in (xb, xa):  # (the agent's two extractors read the SAME input)
    want = [torch.empty(shape, device=dev) for _ in range(2)]
    _cabi.conv4x4s2_fwd(xa, wa, ba, want[0], 1, 0.2)
    _cabi.conv4x4s2_fwd(xb_, wb, bb, want[1], 1, 0.2)
    runs = []
    for _ in range(2):
      got = [torch.full(shape, float('nan'), device=dev) for _ in range(2)]
      _cabi.conv4x4s2_fwd_pair((xa, wa, ba, got[0]), (xb_, wb, bb, got[1]), 1, 0.2)
      assert close(got[0], want[0]) and close(got[1], want[1])
      runs.append(got)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
  if cout % 4 == 0:
    want = [torch.empty_like(xa) for _ in range(2)]
    _cabi.conv4x4s2_bwd_data_mask(gya, wa, xa, want[0], 0.2)
    _cabi.conv4x4s2_bwd_data_mask(gyb, wb, xb, want[1], 0.2)
    runs = []
    for _ in range(2):
      got = [torch.full_like(xa, float('nan')) for _ in range(2)]
      _cabi.conv4x4s2_bwd_data_mask_pair((gya, wa, xa, got[0]), (gyb, wb, xb, got[1]), 0.2)
      assert close(got[0], want[0]) and close(got[1], want[1])
      runs.append(got)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize('n,cin,h', [(3, 6, 64), (5, 14, 64), (4, 17, 64), (64, 17, 64), (2, 6, 32), (2, 20, 64)])
def test_first_layer_with_its_constant_planes_folded(n, cin, h, gpu_device):
  """expo_conv4x4s2_fwd_planes: a first layer on an input whose channels 3 .. are per-image constants (planes_concat's
  output) == the float64 convolution of that input -- with bias + lrelu, and as the tangent pass (slope mask, no bias) --
  and == the pair launch of two such layers; every border class of the 4 x 4 / stride 2 / pad 1 window."""
  from exposure_amd import _cabi
  dev = gpu_device
  w_img = 64
  g = torch.Generator(device=dev).manual_seed(cin + n)
  img = torch.rand((n, h, w_img, 3), device=dev, generator=g)
  vec = torch.randn((n, cin - 3), device=dev, generator=g)
  x = torch.empty((n, h, w_img, cin), device=dev)
  _cabi.planes_concat(img, vec, x, 0.5)
  ws = [(torch.randn((32, cin, 4, 4), device=dev, generator=g) / (16 * cin)**0.5).contiguous(memory_format=torch.channels_last)
        for _ in range(2)]
  bs = [torch.randn((32,), device=dev, generator=g) for _ in range(2)]
  assert _cabi.conv_planes_ok(x.shape, 32) == (h % 8 == 0)
  if not _cabi.conv_planes_ok(x.shape, 32):
    return
  shape = (n, h // 2, w_img // 2, 32)
  ref = [F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), b.double().cpu(), 2, 1).permute(0, 2, 3, 1)
         for w, b in zip(ws, bs)]
  ys = []
  for w, b, r in zip(ws, bs, ref):
    y = torch.full(shape, float('nan'), device=dev)
    _cabi.conv4x4s2_fwd_planes(x, w, b, y, 1, 0.2)
    want = torch.where(r > 0, r, 0.2 * r)
    assert float((y.double().cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * 4
    ys.append(y)
  got = [torch.full(shape, float('nan'), device=dev) for _ in range(2)]
  _cabi.conv4x4s2_fwd_planes_pair((x, ws[0], bs[0], got[0]), (x, ws[1], bs[1], got[1]), 1, 0.2)
  assert torch.equal(got[0], ys[0]) and torch.equal(got[1], ys[1])
  # the tangent pass: no bias, the slope mask of z in the epilogue, in place over z
  z = torch.randn(shape, device=dev, generator=g)
  z.view(-1)[::9] = 0.0
  lin = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), ws[0].double().cpu(), None, 2, 1).permute(0, 2, 3, 1)
  want = lin * _slope(z).double().cpu()
  t = z.clone()
  _cabi.conv4x4s2_fwd_planes(x, ws[0], None, t, 0, 0.2, zmask=t)
  assert float((t.double().cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * 4
