#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_fused_bwd.py -x -q -k "metric_shape or autograd_node" 2>&1 | tail -8 | cut -c1-250
python - <<'PY'
import torch, numpy as np
from exposure_amd import _cabi, synthetic
from oracle import filters_np as fnp
dev = torch.device('cuda:0')
shape = synthetic.SHAPES['C']; n = shape[0]
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand(shape, device=dev, generator=g) * 0.9 + 0.02).half()
dy = (torch.randn(shape, device=dev, generator=g) * 0.5).half()
rng = np.random.default_rng(17)
params = [torch.from_numpy(synthetic.make_params(rng, fid, n)).to(dev) for fid in range(8)]
p = torch.zeros((n, 8, 24), device=dev)
for fid in range(8): p[:, fid, :fnp.NUM_PARAMS[fid]] = params[fid]
ids = torch.arange(8, dtype=torch.int32, device=dev).repeat(n, 1).contiguous()
dx1, dp1 = torch.empty_like(x), torch.empty_like(p)
_cabi.chain_fused_bwd(ids, p, x, dy, dx1, dp1)
acts = [x] + [torch.empty_like(x) for _ in range(8)]
grads = [torch.empty_like(x) for _ in range(8)] + [dy]
dprm = [torch.empty_like(q) for q in params]
_cabi.chain_fwd(list(range(8)), acts, params)
_cabi.chain_bwd(list(range(8)), acts, grads, params, dprm)
err = (dx1.float() - grads[0].float()).abs()
for k in (1, 2, 4, 8, 16):
  tol = k * 2.0**-10 * grads[0].float().abs() + 2e-3
  print('k', k, 'frac within', (err <= tol).float().mean().item())
print('max err', err.max().item(), 'mean |dx|', grads[0].float().abs().mean().item())
s = dy.float().abs().sum(dim=(1, 2, 3))
for fid in range(8):
  a, b = dp1[:, fid, :fnp.NUM_PARAMS[fid]], dprm[fid]
  print(fid, 'max |a-b| / max(|b|, s)', ((a - b).abs() / torch.maximum(b.abs(), s[:, None].expand_as(b))).max().item())
PY
