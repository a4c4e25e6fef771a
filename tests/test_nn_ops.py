"""exposure_amd/nn_ops.py: the convolution family (forward / data gradient / weight gradient, closed under
differentiation) against torch's own conv2d autograd, and the fused bias + lrelu against the reference formula
(util.py:225-229) with TF's sub-gradient at 0.  CPU tests run the library-call structure in float64 (gradcheck /
gradgradcheck); the -m gpu tests run the HIP activation kernels and MIOpen's kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from exposure_amd import nn_ops
from tests._fake_hip import fake_hip
from tests._tol import assert_param_grad_close


def ref_conv(x_nhwc, w):
  return F.conv2d(x_nhwc.permute(0, 3, 1, 2), w, None, stride=2, padding=1).permute(0, 2, 3, 1)


def test_conv_family_first_and_second_derivatives_cpu_float64():
  torch.manual_seed(0)
  x = torch.randn(2, 8, 6, 3, dtype=torch.float64, requires_grad=True)
  w = torch.randn(5, 3, 4, 4, dtype=torch.float64, requires_grad=True)
  assert torch.allclose(nn_ops.conv2d_nhwc(x, w), ref_conv(x, w), atol=1e-12)
  assert torch.autograd.gradcheck(nn_ops.conv2d_nhwc, (x, w), atol=1e-8)
  assert torch.autograd.gradgradcheck(nn_ops.conv2d_nhwc, (x, w), atol=1e-8)


def test_conv_family_matches_torch_double_backward_cpu():
  """The gradient-penalty pattern: d/dW of || d sum(conv(x, W) * c) / dx ||^2, ours vs torch's conv2d."""
  torch.manual_seed(1)
  x0 = torch.randn(3, 16, 16, 6, dtype=torch.float64)
  w0 = torch.randn(8, 6, 4, 4, dtype=torch.float64)
  c = torch.randn(3, 8, 8, 8, dtype=torch.float64)
  out = []
  for conv in (nn_ops.conv2d_nhwc, ref_conv):
    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    gx, = torch.autograd.grad((conv(x, w) * c).sum(), x, create_graph=True)
    out.append(torch.autograd.grad((gx**2).sum(), [w]))
  assert torch.allclose(out[0][0], out[1][0], rtol=1e-10, atol=1e-10)


def lrelu_formula(x, leak=0.2):
  f1, f2 = 0.5 * (1 + leak), 0.5 * (1 - leak)
  return f1 * x + f2 * np.abs(x)


def lrelu_slope(x, leak=0.2):
  return np.where(x > 0, 1.0, np.where(x < 0, leak, 0.5 * (1 + leak)))


def test_bias_lrelu_autograd_wiring_cpu():
  """_BiasLrelu / _LreluGrad with the two C-ABI calls mocked: values, first derivative (incl. the bias reduction) and
  the second derivative with respect to the incoming gradient."""
  rng = np.random.default_rng(0)
  y = rng.standard_normal((3, 4, 5, 8)).astype(np.float32)
  b = rng.standard_normal(8).astype(np.float32)
  y[0, 0, 0, :] = -b  # pre-activation exactly 0: TF's sub-gradient f1 = 0.6
  ty, tb = torch.tensor(y, requires_grad=True), torch.tensor(b, requires_grad=True)
  g = torch.tensor(rng.standard_normal(y.shape).astype(np.float32), requires_grad=True)
  with fake_hip():
    z = nn_ops.bias_lrelu(ty, tb)
    gy, gb = torch.autograd.grad(z, [ty, tb], g, create_graph=True)
    v = torch.tensor(rng.standard_normal(y.shape).astype(np.float32))
    gg, = torch.autograd.grad(gy, g, v, retain_graph=True)
  pre = y + b
  np.testing.assert_allclose(z.detach().numpy(), lrelu_formula(pre), rtol=1e-6, atol=1e-7)
  np.testing.assert_allclose(gy.detach().numpy(), g.detach().numpy() * lrelu_slope(pre), rtol=1e-6)
  assert np.allclose(gy.detach().numpy()[0, 0, 0], 0.6 * g.detach().numpy()[0, 0, 0])
  np.testing.assert_allclose(gb.detach().numpy(), (g.detach().numpy() * lrelu_slope(pre)).reshape(-1, 8).sum(0),
                             rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(gg.numpy(), v.numpy() * lrelu_slope(pre), rtol=1e-6)
  # the fused (dy, dbias) node (the mock reports 8 channels as supported): dbias is differentiable w.r.t. the incoming gradient
  w = torch.tensor(rng.standard_normal(8).astype(np.float32))
  with fake_hip():
    gg_b, = torch.autograd.grad(gb, g, w)
  np.testing.assert_allclose(gg_b.numpy(), np.broadcast_to(w.numpy(), y.shape) * lrelu_slope(pre), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(64, 32, 32, 32), (5, 128), (3, 7, 5, 6), (1, 3), (64, 4, 4, 256), (128, 16, 16, 64), (1, 4)])
def test_bias_lrelu_kernels_match_formula(shape, gpu_device):
  dev = gpu_device
  rng = np.random.default_rng(1)
  y = rng.standard_normal(shape).astype(np.float32)
  b = rng.standard_normal(shape[-1]).astype(np.float32)
  y.reshape(-1, shape[-1])[0] = -b  # exact zeros after the bias add
  for bias in (b, None):
    ty = torch.from_numpy(y).to(dev).requires_grad_(True)
    tb = torch.from_numpy(bias).to(dev).requires_grad_(True) if bias is not None else None
    g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev).requires_grad_(True)
    v = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)
    z = nn_ops.bias_lrelu(ty, tb)
    pre = y + (bias if bias is not None else 0.0)
    # x if x > 0 else 0.2 x: within 1 ulp of the literal 0.6 x + 0.4 |x|
    np.testing.assert_allclose(z.detach().cpu().numpy(), lrelu_formula(pre.astype(np.float64)), rtol=3e-7, atol=1e-30)
    ins = [ty] + ([tb] if tb is not None else [])
    grads = torch.autograd.grad(z, ins, g, create_graph=True)
    ref_gy = g.detach().cpu().numpy() * lrelu_slope(pre).astype(np.float32)
    assert np.array_equal(grads[0].detach().cpu().numpy(), ref_gy)  # one multiply per element: exact
    if tb is not None:
      # the bias gradient (fused into the lrelu backward pass for power-of-two channel counts, expo_lrelu_bwd_bias):
      # |err| <= 1e-4 |ref| + 2e-6 A, A = the column sums of |dy| (tests/_tol.py)
      cols = ref_gy.astype(np.float64).reshape(-1, shape[-1])
      assert_param_grad_close(grads[1].detach().cpu().numpy(), cols.sum(0), np.abs(cols).sum(0), 'bias gradient %r' % (shape,))
      # bit-reproducible, and differentiable once more with respect to the incoming gradient
      again = torch.autograd.grad(nn_ops.bias_lrelu(ty, tb), [tb], g)[0]
      assert torch.equal(again, grads[1].detach())
      w = torch.from_numpy(rng.standard_normal(shape[-1]).astype(np.float32)).to(dev)
      gg_b, = torch.autograd.grad(grads[1], g, w, retain_graph=True)
      assert np.array_equal(gg_b.cpu().numpy(), np.broadcast_to(w.cpu().numpy(), shape) * lrelu_slope(pre).astype(np.float32))
    gg, = torch.autograd.grad(grads[0], g, v)
    assert np.array_equal(gg.cpu().numpy(), v.cpu().numpy() * lrelu_slope(pre).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,size,n', [(6, 32, 64, 16), (17, 32, 64, 8), (32, 64, 32, 8), (128, 256, 8, 8)])
def test_conv_family_matches_torch_on_gpu(cin, cout, size, n, gpu_device):
  """Values, first derivatives and the gradient-penalty double backward of conv2d_nhwc (MIOpen forward / data-gradient
  / weight-gradient kernels) against torch's conv2d autograd on the same device (its generic double backward)."""
  dev = gpu_device
  torch.manual_seed(2)
  x0 = torch.randn(n, size, size, cin, device=dev)
  w0 = (torch.randn(cout, cin, 4, 4, device=dev) * (cin * 16)**-0.5).contiguous(memory_format=torch.channels_last)
  c = torch.randn(n, size // 2, size // 2, cout, device=dev)
  res = []
  for conv in (nn_ops.conv2d_nhwc, ref_conv):
    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    y = conv(x, w)
    gx, gw = torch.autograd.grad((y * c).sum(), [x, w], create_graph=True)
    ggw, = torch.autograd.grad((gx**2).sum(), [w])
    res.append((y.detach(), gx.detach(), gw.detach(), ggw))
  for a, b, what in zip(res[0], res[1], ('y', 'dx', 'dW', 'd/dW |dx|^2')):
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, (what, float((a - b).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(64, 64, 64, 3), (3, 7, 5, 3), (1, 2, 2, 3)])
def test_critic_step_glue_kernels(dtype, shape, gpu_device):
  """expo_gp_inputs / expo_grad_penalty_fwd / _bwd (net.py:126-194's loss glue) against the torch formulas in float64."""
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(3)
  real = torch.rand(shape, device=dev, generator=g).to(dtype)
  fake = (torch.rand(shape, device=dev, generator=g) * 1.5).to(dtype)
  alpha = torch.rand((shape[0], 1, 1, 1), device=dev, generator=g)
  both, interp = nn_ops.critic_step_inputs(real, fake, alpha)
  assert both.dtype == torch.float32 and torch.equal(both[:shape[0]], real.float()) and torch.equal(both[shape[0]:], fake.float())
  ref = real.double() + alpha.double() * (fake.double() - real.double())
  assert float((interp.double() - ref).abs().max()) <= 2e-7 * 1.5
  # the penalty term: gradients with norms on both sides of 1 (the one-sided penalty's kink) and exactly small ones
  grads = torch.randn(shape, device=dev, generator=g) * torch.linspace(0.001, 0.05, shape[0], device=dev)[:, None, None, None]
  grads.requires_grad_(True)
  term, norm = nn_ops.grad_penalty_term(grads)
  gd = grads.detach().double()
  rnorm = torch.sqrt(1e-6 + (gd**2).sum(dim=(1, 2, 3)))
  rterm = torch.clamp_min(rnorm - 1.0, 0.0)**2
  assert float((norm.double() - rnorm).abs().max() / rnorm.max()) <= 2e-6
  assert float((term.double() - rterm).abs().max()) <= 1e-5 * max(1.0, float(rterm.max()))
  w = torch.rand(shape[0], device=dev, generator=g)
  (term * w).sum().backward()
  gref = gd.clone().requires_grad_(True)
  rn = torch.sqrt(1e-6 + (gref**2).sum(dim=(1, 2, 3)))
  ((torch.clamp_min(rn - 1.0, 0.0)**2) * w.double()).sum().backward()
  scale = float(gref.grad.abs().max()) + 1e-12
  assert float((grads.grad.double() - gref.grad).abs().max()) <= 1e-5 * scale
  if shape[0] > 8:
    assert float((rnorm > 1).float().mean()) not in (0.0, 1.0), 'the case must cover both sides of the kink'


def _adam_reference(ps, gs_per_step, lrs, b1, b2, eps):
  """torch.optim.Adam's rule in float64 (no weight decay, no amsgrad)."""
  ps = [p.double().clone() for p in ps]
  ms = [torch.zeros_like(p) for p in ps]
  vs = [torch.zeros_like(p) for p in ps]
  for t, (gs, lr) in enumerate(zip(gs_per_step, lrs), start=1):
    for p, g, m, v in zip(ps, gs, ms, vs):
      g = g.double()
      m.mul_(b1).add_(g, alpha=1 - b1)
      v.mul_(b2).addcmul_(g, g, value=1 - b2)
      denom = v.sqrt() / (1 - b2**t)**0.5 + eps
      p.sub_(lr / (1 - b1**t) * m / denom)
  return ps


@pytest.mark.gpu
def test_hip_adam_with_the_step_counter_advanced_in_front_of_the_update(gpu_device):
  """``HipAdam.step(advanced=True)`` behind a launch that moved the step counter (``critic_report(adam_step=...)``: the
  hand-scheduled steps save the one-thread launch behind every update) == ``step()``, bit for bit, counters included."""
  from exposure_amd import _cabi
  from exposure_amd.optim import HipAdam
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(3)
  init = [torch.randn(sz, device=dev, generator=g) for sz in [(5,), (1030,), (32, 6, 4, 4)]]
  opts = []
  for _ in range(2):
    ps = [t.clone().requires_grad_(True) for t in init]
    opts.append((ps, HipAdam(ps, lr=1e-3, betas=(0.5, 0.9))))
  logits, norm, term = torch.randn((6,), device=dev, generator=g), torch.ones((2,), device=dev), torch.ones((2,), device=dev)
  for step in range(4):
    grads = [torch.randn(t.shape, device=dev, generator=g) for t in init]
    for k, (ps, opt) in enumerate(opts):
      for p, gr in zip(ps, grads):
        p.grad = gr.clone()
      if k == 0:
        opt.step()
      else:
        _cabi.critic_report(logits, norm, term, 2, 2, 2, 10.0, torch.empty((5,), device=dev), adam_step=opt.step_counter())
        assert float(opt.step_counter()) == step + 1
        opt.step(advanced=True)
    assert float(opts[0][1]._step) == float(opts[1][1]._step) == step + 1
    for a, b in zip(opts[0][0], opts[1][0]):
      assert torch.equal(a, b)
  ps, opt = opts[1]
  ps[0].grad = None
  with pytest.raises(AssertionError):
    opt.step(advanced=True)  # a parameter without a gradient: the caller cannot have advanced the counter for it


@pytest.mark.gpu
@pytest.mark.parametrize('count', [7, 70])
def test_hip_adam_matches_the_rule_in_float64(count, gpu_device):
  """expo_adam_step (exposure_amd/optim.py) against torch.optim.Adam's update rule in float64 over six steps with a
  changing learning rate: ragged sizes (1, 3, 1023, 1025 ...), a channels_last conv weight, a parameter without a
  gradient (skipped), more tensors than one launch's table holds (70 > EXPO_ADAM_MAX_TENSORS), and the same six steps
  again as ONE captured launch sequence replayed (device-side step counter and learning rate)."""
  from exposure_amd.optim import HipAdam
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(count)
  sizes = [(1,), (3,), (7, 5), (1023,), (1025,), (64, 6, 4, 4), (100003,)]
  sizes = (sizes * (count // len(sizes) + 1))[:count]
  b1, b2, eps = 0.5, 0.9, 1e-8
  lrs = [2e-3, 2e-3, 1e-3, 5e-4, 0.0, 3e-3]

  def fresh():
    ps = []
    for i, sz in enumerate(sizes):
      p = torch.randn(sz, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + i))
      if sz == (1025,):  # a parameter that does not start on a 16-byte boundary: this tensor takes the scalar path
        base = torch.empty(1026, device=dev)
        base[1:].copy_(p)
        p = base[1:]
        assert p.data_ptr() % 16 == 4
      if len(sz) == 4:
        p = p.contiguous(memory_format=torch.channels_last)
      ps.append(p.requires_grad_(True))
    idle = torch.ones(5, device=dev, requires_grad=True)  # never receives a gradient
    return ps, idle

  grads = [[torch.randn(sz, device=dev, generator=g) * (10.0**float(torch.randint(-6, 2, (1,)).item())) for sz in sizes]
           for _ in lrs]
  ps, idle = fresh()
  want = _adam_reference([p.detach() for p in ps], grads, lrs, b1, b2, eps)

  def check(ps):
    for p, w, sz in zip(ps, want, sizes):
      err = (p.detach().double() - w).abs().max().item()
      assert err <= 4e-7 * w.abs().max().item() + 2e-6 * max(lrs) * len(lrs), (sz, err)

  opt = HipAdam(ps + [idle], lr=lrs[0], betas=(b1, b2), eps=eps)
  for gs, lr in zip(grads, lrs):
    opt.param_groups[0]['lr'].fill_(lr)
    for p, gr in zip(ps, gs):
      p.grad = gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr
    opt.step()
  torch.cuda.synchronize()
  check(ps)
  assert float(opt._step) == len(lrs) and int(opt._ticket) == 0 and torch.equal(idle.detach(), torch.ones(5, device=dev))

  # the same as a captured step, replayed: gradients and learning rate are static inputs of the graph
  ps, idle = fresh()
  opt = HipAdam(ps + [idle], lr=lrs[0], betas=(b1, b2), eps=eps)
  static = [torch.zeros_like(p) for p in ps]
  for p, sg in zip(ps, static):
    p.grad = sg
  for p in ps:
    opt._moments(p)  # allocate the moments outside the capture
  graph = torch.cuda.CUDAGraph()
  from exposure_amd.util import capture_without_gc
  with capture_without_gc(), torch.cuda.graph(graph):
    opt.step()
  for p in ps:  # the capture itself does not execute
    pass
  for gs, lr in zip(grads, lrs):
    opt.param_groups[0]['lr'].fill_(lr)
    for sg, gr in zip(static, gs):
      sg.copy_(gr)
    graph.replay()
  torch.cuda.synchronize()
  check(ps)
  assert float(opt._step) == len(lrs)


@pytest.mark.gpu
def test_hip_adam_state_dict_round_trip_and_torch_interop(gpu_device):
  """HipAdam's ``state_dict`` is torch.optim.Optimizer's: (1) three steps, save, a FRESH HipAdam over copies of the
  parameters loads it and both take three more steps -> identical parameters, moments and step counter; (2) the same
  state loads into ``torch.optim.Adam`` (capturable: tensor step) and the next steps agree to fp32 rounding, and a
  state written by torch's Adam loads back; (3) two parameter groups with their own learning rates, ``add_param_group``;
  (4) a mismatching state is refused; restoring writes the learning-rate scalar IN PLACE (a captured graph reads it)."""
  from exposure_amd.optim import HipAdam
  dev = gpu_device
  sizes = [(5,), (33, 7), (8, 3, 4, 4), (1025,)]

  def fresh(seed=0):
    ps = []
    for i, sz in enumerate(sizes):
      p = torch.randn(sz, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + i))
      if len(sz) == 4:
        p = p.contiguous(memory_format=torch.channels_last)
      ps.append(p.requires_grad_(True))
    return ps

  def grads(k):
    g = torch.Generator(device=dev).manual_seed(1000 + k)
    out = [torch.randn(sz, device=dev, generator=g) for sz in sizes]
    return [t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t for t in out]

  def steps(opt, ps, ks):
    for k in ks:
      for p, g in zip(ps, grads(k)):
        p.grad = g
      opt.step()

  a = fresh()
  opt_a = HipAdam(a, lr=1e-2, betas=(0.5, 0.9))
  steps(opt_a, a, range(3))
  sd = opt_a.state_dict()
  assert sorted(sd['state']) == [0, 1, 2, 3] and float(sd['state'][0]['step']) == 3.0
  assert sd['param_groups'][0]['params'] == [0, 1, 2, 3]
  # (1) resume in a fresh optimiser
  b = [p.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for p in a]
  opt_b = HipAdam(b, lr=123.0, betas=(0.9, 0.999))
  lr_tensor = opt_b.param_groups[0]['lr']
  opt_b.load_state_dict(sd)
  assert opt_b.param_groups[0]['lr'] is lr_tensor and float(lr_tensor) == pytest.approx(1e-2)
  assert opt_b.param_groups[0]['betas'] == (0.5, 0.9) and float(opt_b._step) == 3.0
  steps(opt_a, a, range(3, 6))
  steps(opt_b, b, range(3, 6))
  for p, q in zip(a, b):
    assert torch.equal(p, q)
    assert torch.equal(opt_a.state[p][0], opt_b.state[q][0]) and torch.equal(opt_a.state[p][1], opt_b.state[q][1])
  assert float(opt_b._step) == 6.0
  # (2) torch.optim.Adam takes the same state and continues alike; its own state loads back
  c = [p.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for p in a]
  opt_c = torch.optim.Adam(c, lr=1.0, betas=(0.1, 0.2), capturable=True)
  opt_c.load_state_dict(opt_a.state_dict())
  assert float(opt_c.param_groups[0]['lr']) == pytest.approx(1e-2) and tuple(opt_c.param_groups[0]['betas']) == (0.5, 0.9)
  steps(opt_a, a, range(6, 8))
  steps(opt_c, c, range(6, 8))
  for p, q in zip(a, c):
    assert (p - q).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item())
  d = [p.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for p in c]
  opt_d = HipAdam(d, lr=0.0)
  opt_d.load_state_dict(opt_c.state_dict())
  steps(opt_c, c, range(8, 10))
  steps(opt_d, d, range(8, 10))
  for p, q in zip(c, d):
    assert (p - q).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item())
  assert float(opt_d._step) == 10.0
  # (3) two groups, each with its own learning rate and step counter
  e = fresh(7)
  f = [p.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for p in e]
  opt_e = HipAdam([dict(params=e[:2], lr=1e-2), dict(params=e[2:3], lr=1e-3)], lr=5e-3, betas=(0.5, 0.9))
  opt_e.add_param_group(dict(params=e[3:]))  # -> the default 5e-3
  opt_f = torch.optim.Adam([dict(params=f[:2], lr=1e-2), dict(params=f[2:3], lr=1e-3), dict(params=f[3:], lr=5e-3)],
                           betas=(0.5, 0.9))
  assert [float(g['lr']) for g in opt_e.param_groups] == pytest.approx([1e-2, 1e-3, 5e-3])
  steps(opt_e, e, range(4))
  steps(opt_f, f, range(4))
  for p, q in zip(e, f):
    assert (p - q).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item())
  sd_e = opt_e.state_dict()
  assert [g['params'] for g in sd_e['param_groups']] == [[0, 1], [2], [3]] and len(opt_e.params) == 4
  # (4) refusals
  with pytest.raises(ValueError):
    opt_a.load_state_dict(sd_e)  # three groups into one
  bad = opt_a.state_dict()
  bad['state'][1]['exp_avg'] = bad['state'][1]['exp_avg'][:5]
  with pytest.raises(ValueError):
    opt_b.load_state_dict(bad)
  with pytest.raises(AssertionError):
    opt_e.add_param_group(dict(params=e[:1]))  # already in a group


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('v', [0, 3, 14])
def test_planes_concat_matches_torch(dtype, v, gpu_device):
  """expo_planes_concat == cat([images.float(), vec planes], 3) - 0.5, with autograd's first derivative (channel slice in
  the image's dtype, per-image sum for the planes) and the second one (the penalty path differentiates the input
  gradient again: a function of the upstream gradient only, checked through autograd.grad with create_graph)."""
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(v)
  n, h, w = 5, 16, 12
  img = torch.rand((n, h, w, 3), device=dev, generator=g).to(dtype).requires_grad_(True)
  vec = torch.randn((n, v), device=dev, generator=g).requires_grad_(True) if v else None
  out = nn_ops.planes_concat(img, vec, 0.5)
  ref_img = img.detach().clone().requires_grad_(True)
  ref_vec = vec.detach().clone().requires_grad_(True) if v else None
  ref = ref_img.float()
  if v:
    ref = torch.cat([ref, ref_vec[:, None, None, :].expand(n, h, w, v)], dim=3)
  ref = ref - 0.5
  assert out.dtype == torch.float32 and out.shape == (n, h, w, 3 + v) and torch.equal(out, ref)
  wgt = torch.randn(out.shape, device=dev, generator=g)
  (out * wgt).sum().backward()
  (ref * wgt).sum().backward()
  assert img.grad.dtype == dtype and torch.equal(img.grad, ref_img.grad)
  if v:
    assert float((vec.grad - ref_vec.grad).abs().max()) <= 1e-5 * float(ref_vec.grad.abs().max())
  if dtype == torch.float32 and v:  # second order: d/dw of <grad_img(w), u> + <grad_vec(w), s>
    w2 = wgt.clone().requires_grad_(True)
    img2 = img.detach().clone().requires_grad_(True)
    vec2 = vec.detach().clone().requires_grad_(True)
    gi, gv = torch.autograd.grad((nn_ops.planes_concat(img2, vec2, 0.5) * w2).sum(), [img2, vec2], create_graph=True)
    u, s = torch.randn_like(gi), torch.randn_like(gv)
    gw, = torch.autograd.grad((gi * u).sum() + (gv * s).sum(), w2)
    want = torch.cat([u, s[:, None, None, :].expand(n, h, w, v)], dim=3)
    assert float((gw - want).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('use_td', [True, False])
@pytest.mark.parametrize('use_penalty', [True, False])
def test_generator_loss_glue_matches_the_op_by_op_formula(use_td, use_penalty, gpu_device):
  """expo_generator_losses (nn_ops.generator_losses_fused) against the op-by-op restatement of net.py:92-160 in
  gan.py::generator_losses -- g_loss, v_loss, reward, q and the gradients that reach fake_logit, new_value, old_value,
  penalty and surrogate from the two separate backward passes of a generator step; trajectories past the maximum length
  lose their bootstrap value, stopped ones their continuation."""
  dev = gpu_device
  g = torch.Generator(device=dev).manual_seed(11)
  n, d = 37, 11
  consts = dict(all_reward=0.3, mult=0.05, discount=0.98, plm=1.7, max_len=7)
  mk = lambda: torch.randn((n, 1), device=dev, generator=g)
  base = dict(fake_logit=mk(), fake_input_logit=mk(), new_value=mk(), old_value=mk(), penalty=mk().abs(), surrogate=mk())
  states = torch.zeros((n, d), device=dev)
  states[:, 1] = (torch.rand(n, device=dev, generator=g) < 0.4).float()
  states[:, 2] = torch.randint(1, 10, (n,), device=dev, generator=g).float()
  assert float(states[:, 2].max()) > consts['max_len'] and float(states[:, 1].sum()) > 0

  def leaves():
    return {k: v.clone().requires_grad_(k != 'fake_input_logit') for k, v in base.items()}

  a = leaves()
  g_loss, v_loss, reward, q = nn_ops.generator_losses_fused(
      a['fake_logit'], a['fake_input_logit'], a['new_value'], a['old_value'], states, a['penalty'] if use_penalty else None,
      a['surrogate'], (consts['all_reward'], consts['mult'], consts['discount'], consts['plm'], consts['max_len']), use_td)
  b = leaves()
  stopped = states[:, 1:2]
  clear_final = (states[:, 2:3] > consts['max_len']).float()
  nv = b['new_value'] * (1.0 - clear_final)
  gate = consts['all_reward'] + (1 - consts['all_reward']) * stopped
  raw = gate * (b['fake_logit'] - b['fake_input_logit']) * consts['mult']
  r_ref = raw - b['penalty'] if use_penalty else raw
  q_ref = r_ref + (1.0 - stopped) * consts['discount'] * nv
  adv = q_ref.detach() - b['old_value']
  v_ref = (adv**2).mean()
  routine, weight = (-q_ref * consts['plm'], -adv) if use_td else (-r_ref, -r_ref)
  g_ref = (routine + b['surrogate'] * weight.detach()).mean()
  close = lambda x, y, tol=2e-6: float((x - y).abs().max()) <= tol * max(1.0, float(y.abs().max()))
  assert close(g_loss, g_ref) and close(v_loss, v_ref) and close(reward, r_ref) and close(q, q_ref)
  assert reward.shape == (n, 1) and q.shape == (n, 1) and not reward.requires_grad
  v_loss.backward()  # the value net's backward runs first, on its own (gan.py::_generator_body)
  g_loss.backward()
  v_ref.backward()
  g_ref.backward()
  for k in ('fake_logit', 'new_value', 'old_value', 'penalty', 'surrogate'):
    ga, gb = a[k].grad, b[k].grad
    if gb is None or float(gb.abs().max()) == 0.0:
      assert ga is None or float(ga.abs().max()) == 0.0, k
    else:
      assert ga is not None and close(ga, gb, 3e-6), k


# ---- the in-house convolution kernels (csrc/conv_ops.hip, round 5) ---------------------------------------------------
CONV_CASES = [(3, 8, 5, 7), (2, 16, 14, 32), (5, 12, 6, 32), (2, 64, 17, 32), (9, 8, 32, 64), (4, 16, 64, 128),
              (16, 8, 128, 256), (1, 2, 3, 1), (7, 4, 4, 33), (8, 64, 14, 32), (32, 32, 32, 64)]


def _conv_case(n, h, cin, cout, dev, seed):
  g = torch.Generator(device=dev).manual_seed(seed)
  x = torch.randn((n, h, h, cin), device=dev, generator=g)
  w = (torch.randn((cout, cin, 4, 4), device=dev, generator=g) / (16 * cin)**0.5).contiguous(memory_format=torch.channels_last)
  b = torch.randn((cout,), device=dev, generator=g) * 0.1
  return x, w, b


@pytest.mark.gpu
@pytest.mark.parametrize('case', CONV_CASES)
def test_hip_conv_forward_matches_float64(case, gpu_device, monkeypatch):
  """expo_conv4x4s2_fwd (implicit GEMM on v_mfma_f32_32x32x2_f32) against a float64 convolution on the CPU, plain and
  with the bias + lrelu epilogue, for EVERY decomposition the library can pick -- the flat one under 1 / 2 column tiles
  per wave and 1 .. 16 K slices, the four LDS-tiled shapes, the first layers' row-staged kernel -- on shapes with ragged tiles (M, Cout not multiples of
  32), the first layers' channel counts (14, 6, 17: chunks cut by the image edge), one-pixel outputs.  f32 MFMA is an
  exact fmaf chain: the error is f32 summation rounding, 2e-6 of the largest output; MIOpen's own error on the same
  operands is printed beside it."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, b = _conv_case(n, h, cin, cout, dev, seed=n + h)
  ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, 2, 1).permute(0, 2, 3, 1)
  scale = float(ref.abs().max())
  lib = float((ref_conv(x, w).double().cpu() - ref).abs().max()) / scale
  variants = ([('0', '0', '0')] + [('5', nt, sl) for nt in ('1', '2') for sl in ('0', '1', '2', '4', '8', '16')] +
              [(t, '0', '0') for t in '12346'])
  y = torch.empty((n, h // 2, h // 2, cout), device=dev)
  worst = 0.0
  try:  # (the override is process-wide: a failing assertion must not leak a forced plan into later tests)
    for tile, nt, sl in variants:
      _cabi.conv_tuning(int(tile), int(nt), int(sl))
      for act in (0, 1):
        y.fill_(float('nan'))
        _cabi.conv4x4s2_fwd(x, w, b if act else None, y, act, 0.2)
        want = ref + b.double().cpu() if act else ref
        if act:
          want = torch.where(want > 0, want, want * 0.2)
        err = float((y.double().cpu() - want).abs().max()) / scale
        assert err < 2e-6, (case, tile, nt, sl, act, err)
        worst = max(worst, err)
  finally:
    _cabi.conv_tuning(0, 0, 0)
  print('conv fwd %s: worst %.2e of max |y| over %d variants (MIOpen %.2e)' % (case, worst, len(variants), lib))


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,n,h', [(14, 32, 6, 16), (64, 128, 5, 8)])
def test_fused_conv_layer_autograd_matches_the_library_pair(cin, cout, n, h, gpu_device):
  """nn_ops.conv_bias_lrelu (one launch forward; every derivative on the in-house kernels) == the same layer written with
  torch's own operators (aten convolution through MIOpen, its generic double backward): value, first derivatives with
  respect to input / weight / bias, and the gradient-penalty pattern -- the derivative of ||d out / d x||^2 with respect
  to the weight (double backward through the layer)."""
  dev = gpu_device
  x0, w0, b0 = _conv_case(n, h, cin, cout, dev, seed=3)
  c = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=torch.Generator(device=dev).manual_seed(9))
  res = []
  for hip in (True, False):
    x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    if hip:
      z = nn_ops.conv_bias_lrelu(x, w, b)
    else:
      y = ref_conv(x, w) + b
      z = 0.6 * y + 0.4 * y.abs()  # util.py:225-229
    gx, gw, gb = torch.autograd.grad((z * c).sum(), [x, w, b], create_graph=True)
    ggw, = torch.autograd.grad((gx**2).sum(), [w])
    res.append([t.detach() for t in (z, gx, gw, gb, ggw)])
  for name, a, r in zip(('z', 'gx', 'gw', 'gb', 'ggw'), *res):
    tol = 2e-5 * float(r.abs().max()) + 1e-7
    assert float((a - r).abs().max()) <= tol, (name, float((a - r).abs().max()), tol)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 8, 5, 8), (2, 16, 14, 32), (2, 64, 17, 32), (9, 8, 32, 64), (4, 16, 64, 128),
                                  (16, 8, 128, 256), (7, 4, 4, 36), (32, 32, 32, 64), (3, 4, 3, 4)])
def test_hip_conv_data_gradient_matches_float64(case, gpu_device, monkeypatch):
  """expo_conv4x4s2_bwd_data (four parity-class GEMMs on the f32 matrix cores) against the float64 autograd gradient
  of a convolution on the CPU, under every K' slice count (1 .. 16 waves per tile); ragged pixel / channel tiles,
  channel counts below one tile, the smallest image.  Every element of dx is written (the buffer starts as NaN)."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, _ = _conv_case(n, h, cin, cout, dev, seed=n + h)
  g = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
  xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
  yd = F.conv2d(xd, w.double().cpu(), None, 2, 1)
  ref, = torch.autograd.grad(yd, [xd], g.double().cpu().permute(0, 3, 1, 2))
  ref = ref.permute(0, 2, 3, 1)
  scale = float(ref.abs().max())
  dx = torch.empty((n, h, h, cin), device=dev)
  try:
    for nt in ('0', '1', '2'):  # input-channel tiles per wave (0: the library's choice)
      for sl in ('0', '1', '2', '4', '8', '16'):
        _cabi.conv_tuning(0, int(nt), int(sl))
        dx.fill_(float('nan'))
        _cabi.conv4x4s2_bwd_data(g, w, dx)
        err = float((dx.double().cpu() - ref).abs().max()) / scale
        assert err < 3e-6, (case, nt, sl, err)
  finally:
    _cabi.conv_tuning(0, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 8, 5, 8), (2, 16, 14, 32), (2, 64, 17, 32), (9, 8, 32, 64), (4, 16, 64, 128),
                                  (16, 8, 128, 256), (7, 4, 4, 36), (32, 32, 32, 64), (3, 4, 3, 4)])
def test_hip_conv_weight_gradient_matches_float64(case, gpu_device):
  """expo_conv4x4s2_wrw against the float64 autograd weight gradient on the CPU, under
  several (waves per block, blocks per tile) splits of the pixel sum incl. P = 1 (no reduce launch) and an odd P, twice
  in a row through the same scratch; every element of dw is written (the buffer starts as NaN); the result of a given
  split is bit-reproducible (fixed summation order: no atomics)."""
  from exposure_amd import _cabi
  n, h, cin, cout = case
  dev = gpu_device
  x, w, _ = _conv_case(n, h, cin, cout, dev, seed=n + h)
  g = torch.randn((n, h // 2, h // 2, cout), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
  wd = w.double().cpu().requires_grad_(True)
  yd = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wd, None, 2, 1)
  ref, = torch.autograd.grad(yd, [wd], g.double().cpu().permute(0, 3, 1, 2))
  scale = float(ref.abs().max())
  dw = torch.empty_like(w)
  try:
    for sl, parts in ((0, 0), (1, 1), (2, 3), (4, 0), (4, 7), (3, 2)):
      _cabi.conv_wrw_tuning(sl, parts)
      runs = []
      for _ in range(2):
        dw.fill_(float('nan'))
        _cabi.conv4x4s2_wrw(x, g, dw)
        err = float((dw.double().cpu() - ref).abs().max()) / scale
        assert err < 1e-5, (case, sl, parts, err)  # (P = S = 1: one wave alone sums up to 16 384 terms in f32)
        runs.append(dw.clone())
      assert torch.equal(runs[0], runs[1])
  finally:
    _cabi.conv_wrw_tuning(0, 0)
