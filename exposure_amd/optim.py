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
  """``param_groups`` / ``state`` / ``state_dict()`` / ``load_state_dict()`` / ``add_param_group()`` follow
  ``torch.optim.Optimizer``: a ``state_dict`` written here loads into ``torch.optim.Adam`` over the same parameters and
  the other way round (entries ``step``, ``exp_avg``, ``exp_avg_sq`` per parameter index).  One launch per group and
  step; a group's step counter is one device scalar shared by its parameters (torch keeps one per parameter, all equal
  unless a gradient was ``None`` in some steps)."""

  def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
    params = list(params)
    assert params, 'HipAdam: empty parameter list'
    self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps)
    self.param_groups = []
    self.state = {}
    for group in (params if isinstance(params[0], dict) else [dict(params=params)]):
      self.add_param_group(group)

  @property
  def params(self):
    return [p for g in self.param_groups for p in g['params']]

  @property
  def _step(self):  # (the first group's counters: what a one-group optimiser's tests and tools look at)
    return self.param_groups[0]['_step']

  @property
  def _ticket(self):
    return self.param_groups[0]['_ticket']

  def add_param_group(self, group):
    group = dict(group)
    ps = group['params'] = [group['params']] if torch.is_tensor(group['params']) else list(group['params'])
    assert ps and all(p.is_cuda and p.dtype == torch.float32 for p in ps), 'HipAdam: fp32 parameters on a ROCm device'
    seen = {id(p) for g in self.param_groups for p in g['params']}
    assert not any(id(p) in seen for p in ps) and len({id(p) for p in ps}) == len(ps), 'parameter in two groups'
    dev = ps[0].device
    for k, v in self.defaults.items():
      group.setdefault(k, v)
    lr = group['lr']
    # every group owns its learning-rate scalar (a tensor passed in is copied, so two groups never alias one)
    group['lr'] = (lr.detach().to(device=dev, dtype=torch.float32).clone() if torch.is_tensor(lr) else
                   torch.tensor(float(lr), device=dev))
    group['betas'] = tuple(group['betas'])
    group['_step'] = torch.zeros((), dtype=torch.float32, device=dev)  # steps taken so far: advanced by the kernel
    group['_ticket'] = torch.zeros((), dtype=torch.int32, device=dev)
    self.param_groups.append(group)

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

  def step_counter(self):
    """The device scalar a kernel in FRONT of ``step(advanced=True)`` advances (``_cabi.critic_report`` /
    ``generator_losses``): one-group optimisers only."""
    assert len(self.param_groups) == 1
    return self.param_groups[0]['_step']

  @torch.no_grad()
  def step(self, advanced=False):
    """``advanced``: the caller's previous launch has moved the step counter (``step_counter()``) already -- the update
    then needs no one-thread launch behind it.  Every parameter must have a gradient in that case."""
    assert not advanced or (len(self.param_groups) == 1 and all(p.grad is not None for p in self.params))
    for group in self.param_groups:
      lr = group['lr']
      if not torch.is_tensor(lr):  # a float assigned from outside: move it to the device (not capturable, like torch)
        lr = group['lr'] = torch.tensor(float(lr), device=group['_step'].device)
      ps, gs, ms, vs = [], [], [], []
      for p in group['params']:
        g = p.grad
        if g is None:
          continue
        if g.stride() != p.stride():  # autograd normally hands over the parameter's layout; if not, one copy
          g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
        m, v = self._moments(p)
        ps.append(p), gs.append(g), ms.append(m), vs.append(v)
      _cabi.adam_step(ps, gs, ms, vs, lr, group['_step'], group['_ticket'], group['betas'][0], group['betas'][1],
                      group['eps'], step_advanced=advanced)

  # -- checkpoint / resume (torch.optim.Optimizer's format)
  def state_dict(self):
    index, groups, state = {}, [], {}
    for group in self.param_groups:
      ids = []
      for p in group['params']:
        ids.append(index.setdefault(id(p), len(index)))
        if p in self.state:
          m, v = self.state[p]
          state[ids[-1]] = dict(step=group['_step'].detach().clone(), exp_avg=m.detach().clone(),
                                exp_avg_sq=v.detach().clone())
      lr = group['lr']
      groups.append(dict(lr=lr.detach().clone() if torch.is_tensor(lr) else lr, betas=group['betas'], eps=group['eps'],
                         weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=True,
                         differentiable=False, fused=None, decoupled_weight_decay=False, params=ids))
    return dict(state=state, param_groups=groups)

  @torch.no_grad()
  def load_state_dict(self, sd):
    groups = sd['param_groups']
    if len(groups) != len(self.param_groups) or any(
        len(a['params']) != len(b['params']) for a, b in zip(groups, self.param_groups)):
      raise ValueError('HipAdam.load_state_dict: parameter groups do not match')
    for saved, group in zip(groups, self.param_groups):
      if saved.get('weight_decay', 0) or saved.get('amsgrad', False) or saved.get('maximize', False):
        raise ValueError('HipAdam implements plain Adam (no weight decay, amsgrad or maximize)')
      dev = group['_step'].device
      # IN PLACE: a captured hipGraph reads these very scalars (learning rate, step counter)
      lr = saved['lr']
      if torch.is_tensor(group['lr']):
        group['lr'].copy_(lr.to(dev) if torch.is_tensor(lr) else torch.tensor(float(lr), device=dev))
      else:
        group['lr'] = float(lr)
      group.pop('_lr_value', None)  # gan.GAN.set_lrs' note of what the scalar holds
      group['betas'], group['eps'] = tuple(saved['betas']), saved['eps']
      steps = []
      for idx, p in zip(saved['params'], group['params']):
        st = sd['state'].get(idx)
        if st is None:
          # no moments in the checkpoint: ZERO the existing buffers in place (a captured hipGraph keeps updating these
          # very tensors; popping them would leave the graph on stale moments while an eager step allocates new ones)
          if p in self.state:
            for t in self.state[p]:
              t.zero_()
          continue
        if tuple(st['exp_avg'].shape) != tuple(p.shape):
          raise ValueError('HipAdam.load_state_dict: moment shape %s for a parameter of shape %s' %
                           (tuple(st['exp_avg'].shape), tuple(p.shape)))
        m, v = self._moments(p)  # existing buffers are kept (the captured graph's pointers stay valid)
        m.copy_(st['exp_avg'])
        v.copy_(st['exp_avg_sq'])
        steps.append(float(st['step']))
      if steps and min(steps) != max(steps):
        raise ValueError('HipAdam keeps one step counter per group; the state holds %s' % sorted(set(steps)))
      group['_step'].fill_(steps[0] if steps else 0.0)
