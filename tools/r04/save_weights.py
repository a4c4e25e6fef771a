import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exposure_amd.config import make_cfg
from exposure_amd.gan import GAN
dev = torch.device('cuda:0')
torch.manual_seed(5)
cfg = make_cfg()
cfg.gan, cfg.use_TD, cfg.gradient_penalty_lambda = 'w', True, 0
gan = GAN(cfg, device=dev)
with torch.no_grad():
  for p in gan.parameters():
    if p.dim() == 1:
      p.normal_(0.0, 0.05)
  gan.critic.fc2.weight.mul_(40.0)
sd = {k: v.cpu() for k, v in gan.state_dict().items() if k.startswith(('generator.filter_features', 'generator.filters.6', 'critic', 'value'))}
torch.save(sd, 'gpurun_out/r04_gpu_seed5.pt')
print(sum(v.numel() for v in sd.values()))
