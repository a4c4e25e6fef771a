"""Adam for the three optimisers of a training iteration (``/root/reference/net.py:222-251``,
``config_example.py:158``: ``tf.train.AdamOptimizer(lr, beta1=0.5, beta2=0.9)``) as ONE HIP launch per step
(``expo_adam_step``): torch's fused multi-tensor Adam gives 65 536 elements to a block -- 20 to 75 blocks for one of
these networks on 256 CUs, 43 us per step, 8 steps per iteration.

Same update rule and the same surface as the ``torch.optim.Adam`` it replaces on a ROCm device (``param_groups[i]['lr']``
float or device tensor, ``step()``, ``zero_grad``); the step counter and the learning rate live on the device, so a
step is capturable into a hipGraph and replays without host involvement.  Dense parameters only -- a parameter whose
gradient is ``None`` is skipped, as in torch."""
import torch

from . import _cabi


class HipAdam:

  def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
    self.params = [p for p in params]
    assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)
    dev = self.params[0].device
    lr = lr if torch.is_tensor(lr) else torch.tensor(float(lr), device=dev)
    self.param_groups = [dict(params=self.params, lr=lr.to(device=dev, dtype=torch.float32), betas=tuple(betas), eps=eps)]
    self.state = {}
    self._step = torch.zeros((), dtype=torch.float32, device=dev)  # t - 1: advanced by the kernel
    self._ticket = torch.zeros((), dtype=torch.int32, device=dev)

  def zero_grad(self, set_to_none=True):
    for p in self.params:
      if set_to_none:
        p.grad = None
      elif p.grad is not None:
        p.grad.zero_()

  def _moments(self, p):
    st = self.state.get(p)
    if st is None:  # same strides as the parameter (conv weights are channels_last): the kernel walks raw memory
      st = self.state[p] = (torch.zeros_like(p, memory_format=torch.preserve_format),
                            torch.zeros_like(p, memory_format=torch.preserve_format))
    return st

  @torch.no_grad()
  def step(self):
    group = self.param_groups[0]
    lr = group['lr']
    if not torch.is_tensor(lr):  # a float assigned from outside: move it to the device (not capturable, like torch)
      lr = group['lr'] = torch.tensor(float(lr), device=self._step.device)
    ps, gs, ms, vs = [], [], [], []
    for p in self.params:
      g = p.grad
      if g is None:
        continue
      if g.stride() != p.stride():  # autograd normally hands over the parameter's layout; if not, one copy
        g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
      m, v = self._moments(p)
      ps.append(p), gs.append(g), ms.append(m), vs.append(v)
    _cabi.adam_step(ps, gs, ms, vs, lr, self._step, self._ticket, group['betas'][0], group['betas'][1], group['eps'])
