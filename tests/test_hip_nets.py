"""-m gpu twin of tests/test_oracle_nets.py: the torch convnets / loss graph of exposure_amd on the
MI355X (MIOpen / hipBLASLt GEMMs, the filter step through libexposure_hip.so) against the float64
NumPy oracle (oracle/nets_np.py) holding the same weights -- rows a-12, a-14, a-15 of SURVEY.md 8(a):
features, logits, g_loss, v_loss, c_loss, gradient norm / penalty within 1e-4 relative (fp32 nets);
selected filter ids and states bit-equal."""
import pytest
import torch

from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
from tests.test_oracle_nets import GRAD_TENSORS_SHORT, compare_gan_with_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [11, 12])
def test_torch_nets_and_losses_match_oracle_gpu(gpu_device, seed):
  torch.manual_seed(seed)
  gan = GAN(make_cfg(), device=gpu_device)
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)  # gradient norm > 1: the one-sided penalty is active
  res = compare_gan_with_oracle(gan, gpu_device, n=8, seed=seed)
  assert res['gradient_norm'] > 1.0
  # weight gradients of g_loss / v_loss / c_loss against finite differences of the float64 oracle: every filter head
  # (the HIP kernels' parameter gradients feed them) must have been reached
  assert not any(k[2] == 'zero' and 'filter_' in k[1] for k in res['grad_report'])


@pytest.mark.parametrize('gan_kind,use_td,gp_lambda', [('ls', False, 10), ('w', True, 0)])
def test_loss_branches_match_oracle_gpu(gpu_device, gan_kind, use_td, gp_lambda):
  """net.py:100-199's other branches on the device: LSGAN with the plain-reward policy gradient; WGAN without the
  penalty term (a training step of the latter also clips theta_c, net.py:252-262)."""
  torch.manual_seed(5)
  cfg = make_cfg()
  cfg.gan, cfg.use_TD, cfg.gradient_penalty_lambda = gan_kind, use_td, gp_lambda
  gan = GAN(cfg, device=gpu_device)
  with torch.no_grad():
    for p in gan.parameters():
      if p.dim() == 1:
        p.normal_(0.0, 0.05)
    gan.critic.fc2.weight.mul_(40.0)
  compare_gan_with_oracle(gan, gpu_device, n=8, seed=13, grad_tensors=GRAD_TENSORS_SHORT)
  from tests.test_oracle_nets import make_batch
  fake_input, real, _s, _z, _m, alpha = make_batch(8, 14)
  t = lambda a: torch.from_numpy(a).to(gpu_device)
  out = gan.critic_step(t(real), t(fake_input), it=1, alpha=t(alpha))
  assert torch.isfinite(out['c_loss'])
  worst = max(float(p.detach().abs().max()) for p in gan.critic.parameters())
  assert (worst <= cfg.clamp_critic + 1e-12) == (gan_kind == 'w' and gp_lambda == 0)


def test_losses_match_oracle_with_f16_image_pool(gpu_device):
  """Same comparison with fp16 image storage feeding the nets (the filter kernels' default dtype)."""
  torch.manual_seed(3)
  gan = GAN(make_cfg(), device=gpu_device)
  import numpy as np
  from exposure_amd import checkpoint
  from oracle import nets_np as nn_np
  from tests.test_oracle_nets import make_batch
  fake_input, real, states, z, masks, alpha = make_batch(8, 21)
  fake_input = fake_input.astype(np.float16)
  t = lambda a: torch.from_numpy(a).to(gpu_device)
  d = lambda a: a.astype(np.float64)
  out = gan.generator_losses(t(fake_input), t(z), t(states), 0.5, 1, [t(m) for m in masks])
  weights = {k: v.astype(np.float64) for k, v in checkpoint.export_tf_dict(gan).items()}
  ref = nn_np.generator_losses(d(fake_input), d(z), d(states), 0.5, nn_np.DEFAULT_CFG, weights, [d(m) for m in masks], 1)
  assert np.array_equal(out['debug']['selected_filter_ids'].cpu().numpy(), ref['debug']['selected_filter_id'])
  assert out['fake_output'].dtype == torch.float16
  # the fp16-stored step output feeds the critic / value nets: loss tolerance follows the 1e-3 pixel bound
  got = out['fake_output'].detach().float().cpu().numpy()
  assert (np.abs(got - ref['fake_output']) <= 1e-3 + np.abs(ref['fake_output']) * 2.0**-11).all()
  assert abs(float(out['g_loss'].detach()) - ref['g_loss']) <= 2e-3 * max(1.0, abs(ref['g_loss']))
  assert abs(float(out['v_loss'].detach()) - ref['v_loss']) <= 2e-3 * max(1.0, abs(ref['v_loss']))
