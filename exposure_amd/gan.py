"""One training iteration of the reference's GAN trainer (``/root/reference/net.py:92-251``):
the callers of the filter hot path.  Losses, stop-gradients, optimiser settings and learning
rate schedules follow the reference line by line; the filter work inside ``Agent`` runs in the
HIP library.

``GAN.generator_step`` == one ``sess.run([opt_g, opt_v, ...])`` (net.py:325-335);
``GAN.critic_step`` == one ``sess.run(opt_c)`` (net.py:358-365).  With a process group the
gradients are all-reduced in flat buckets (``exposure_amd.dist``) before the optimiser steps.
"""
import os

import numpy as np
import torch
from torch import nn

from . import critic_direct, generator_direct
from . import dist as xdist
from .agent import Agent
from .critics import Critic
from .nn_ops import (critic_step_inputs, frozen_parameters, generator_losses_fused, grad_penalty_term,
                     once_differentiable_convnets, skip_parameter_gradients)
from .replay_memory import HostStaged, PoolRows, materialize
from .util import STATE_STEP_DIM, STATE_STOPPED_DIM, capture_without_gc


class GAN(nn.Module):

  def __init__(self, cfg, device=None, process_group=None, use_graphs=False, seed=0, direct_critic=True,
               direct_generator=True):
    super().__init__()
    self.cfg = cfg
    self.use_graphs = bool(use_graphs)
    # the critic update as a hand-scheduled launch sequence (exposure_amd/critic_direct.py) wherever it applies -- the
    # default Wasserstein critic with the gradient penalty on a ROCm device; False keeps every step on autograd (the
    # path the direct schedule is tested against, and the one every other configuration takes)
    self.direct_critic = bool(direct_critic)
    # likewise the critic / value-net passes of the generator step (exposure_amd/generator_direct.py); the agent itself
    # stays on autograd
    self.direct_generator = bool(direct_generator)
    self._graphs = {}
    self.generator = Agent(cfg)
    self.critic = Critic(cfg, num_state_dim=0)
    self.value = Critic(cfg, num_state_dim=cfg.num_state_dim)
    if device is not None:
      self.to(device)
    # the filter heads' parameters move into their packed storage NOW -- before the optimisers, the gradient buckets
    # and any captured step take pointers to them (a lazy pack inside the first forward would be fine, a later re-pack
    # behind captured graphs is not: _check_heads)
    fused_path = not cfg.masking and device is not None and torch.device(device).type == 'cuda'
    self._heads_pack = self.generator.pack_heads() if fused_path else None
    self._heads_generation = self._heads_pack.generation if self._heads_pack is not None else 0
    adam = dict(betas=(cfg.adam_beta1, cfg.adam_beta2), eps=1e-8)  # config_example.py:158
    if device is not None and torch.device(device).type == 'cuda' and os.environ.get('EXPO_FUSED_ADAM', '1') == '1':
      adam['fused'] = True  # one multi-tensor kernel per optimiser step instead of ~10 foreach launches
    if self.use_graphs:
      # hipGraph replay of a whole step: learning rates live in device tensors, Adam is capturable
      assert device is not None and torch.device(device).type == 'cuda', 'graphs need a ROCm device'
      adam['capturable'] = True
      lr = lambda v: torch.tensor(float(v), device=device)
    else:
      lr = float
    if device is not None and torch.device(device).type == 'cuda' and os.environ.get('EXPO_HIP_ADAM', '1') == '1':
      # one launch per optimiser step with a block per 1 024 elements (exposure_amd/optim.py; torch's fused kernel:
      # a block per 65 536 -- 43 us for a network of these sizes); learning rate and step counter on the device
      from .optim import HipAdam
      make = lambda params, value: HipAdam(params, torch.tensor(float(value), device=device), betas=adam['betas'],
                                           eps=adam['eps'])
    else:
      make = lambda params, value: torch.optim.Adam(params, lr=lr(value), **adam)
    self.opt_g = make(self.generator.parameters(), cfg.lr_g(0))
    self.opt_v = make(self.value.parameters(), cfg.value_lr_mul * cfg.lr_g(0))
    self.opt_c = make(self.critic.parameters(), cfg.lr_c(0))
    self.process_group = process_group
    self.world_size = xdist.world_size(process_group)
    # dropout masks and the gradient penalty's alpha: drawn for the GLOBAL batch from a generator every rank seeds
    # identically, each rank keeping its image shard's rows -- results do not depend on the partition
    self.rng = xdist.GlobalBatchRng(seed, device)  # device None: bound to the parameters' device at the first draw
    # EXPO_FORCE_COLLECTIVES=1: issue the gradient all-reduces even in a one-rank group, so the RCCL
    # path (and its hipGraph capture) can be exercised on a single GPU
    self.force_collectives = os.environ.get('EXPO_FORCE_COLLECTIVES', '0') == '1'
    # Steps that contain RCCL collectives are captured and replayed like any other (the collectives are
    # stream-ordered kernels); EXPO_GRAPH_COLLECTIVES=0 falls back to eager launches for multi-rank runs.
    # (Round 1 kept this opt-in because ~1 run in 15 aborted; the cause -- the NCCL watchdog polling an
    # eager work's event while RCCL's stream was being captured -- is handled in _replay.)
    # EXPO_GRAPH_COLLECTIVES: 'auto' (default) captures them once the watchdog drain in front of the capture is
    # VERIFIED (exposure_amd.dist.drain_before_capture) and otherwise keeps multi-rank steps eager; '1' captures after
    # the unverified grace period too; '0' never captures a step that contains collectives.
    collectives = self.world_size > 1 or self.force_collectives
    self._graph_collectives = os.environ.get('EXPO_GRAPH_COLLECTIVES', 'auto')
    self._replay_steps = self.use_graphs and (not collectives or self._graph_collectives != '0')
    # Flat gradient buckets (p.grad are views into them).  theta_g is split where the backward pass splits
    # in time: the FC heads (8 x fc1/fc2 + the selector FCs, 19 MB) receive their gradients first, the two
    # conv trunks (5.6 MB) last -- so the heads' all-reduce runs under the trunks' backward, and theta_v's
    # (whose small backward runs first) under the whole generator backward.
    gen = self.generator
    trunk = list(gen.filter_features.parameters()) + list(gen.selector_features.parameters())
    trunk_ids = {id(p) for p in trunk}
    heads = [p for p in gen.parameters() if id(p) not in trunk_ids]
    self.buckets = {
        'g_head': xdist.GradBucket(heads, on_ready=self._bucket_ready),
        'g_trunk': xdist.GradBucket(trunk, on_ready=self._bucket_ready),
        'v': xdist.GradBucket(self.value.parameters(), on_ready=self._bucket_ready),
        'c': xdist.GradBucket(self.critic.parameters(), on_ready=self._bucket_ready),
    }
    self._pending = []
    self._progress = None  # the step's `progress` input as a device scalar (generator_step)
    self._it_ring = None   # pinned staging buffers of the iteration plans (train_iteration)
    # ExponentialMovingAverage(decay=0.99, zero_debias=True) of c_average (net.py:107-108, 165-168): the biased average
    # lives on the device and is advanced INSIDE the critic step (one in-place lerp, part of the captured graph)
    self._c_ema = None
    self.c_average_steps = 0

  # -- optimiser state (the Adam slots and the logit-centre average a ``tf.train.Saver`` keeps beside the weights,
  #    net.py:271, 380-384): ``torch.save({'model': gan.state_dict(), 'optim': gan.optimizer_state_dict()}, path)``
  def optimizer_state_dict(self):
    return dict(opt_g=self.opt_g.state_dict(), opt_v=self.opt_v.state_dict(), opt_c=self.opt_c.state_dict(),
                c_ema=None if self._c_ema is None else self._c_ema.detach().clone(),
                c_average_steps=self.c_average_steps)

  @torch.no_grad()
  def load_optimizer_state_dict(self, sd):
    """Restores moments, step counters and learning rates IN PLACE (captured step graphs keep reading the same
    buffers); the next ``set_lrs`` refills the learning-rate scalars for its iteration."""
    self.opt_g.load_state_dict(sd['opt_g'])
    self.opt_v.load_state_dict(sd['opt_v'])
    self.opt_c.load_state_dict(sd['opt_c'])
    for opt in (self.opt_g, self.opt_v, self.opt_c):
      for g in opt.param_groups:
        g.pop('_lr_value', None)
    if sd.get('c_ema') is not None:
      if self._c_ema is None:
        self._c_ema = sd['c_ema'].detach().clone().to(next(self.parameters()).device)
      else:
        self._c_ema.copy_(sd['c_ema'])
      self.c_average_steps = int(sd.get('c_average_steps', 0))
    else:
      # no average in the checkpoint: restart it -- the live scalar zeroed IN PLACE (captured steps advance this tensor),
      # the step count with it (a stale value over (1 - 0.99^1) would be a hundredfold overshoot)
      if self._c_ema is not None:
        self._c_ema.zero_()
      self.c_average_steps = 0

  # -- learning rates (config_example.py:134-158; net.py:222-251)
  def set_lrs(self, it, zero_g=False):
    lr_g = 0.0 if zero_g else self.cfg.lr_g(it)  # net.py:327-328: lr_g = 0 at iter 0

    def put(opt, value):
      for g in opt.param_groups:
        lr = g['lr']
        if torch.is_tensor(lr):
          # the five critic steps of an iteration repeat the G step's values: no launch.  The note is only trusted for
          # the tensor it was made for and while nobody else wrote to it (tensor version counter): a scalar replaced or
          # refilled from outside (a restored optimiser state) is filled again.
          if g.get('_lr_value') != (value, id(lr), lr._version):
            lr.fill_(value)  # in place: the captured graph reads this tensor
            g['_lr_value'] = (value, id(lr), lr._version)
        else:
          g['lr'] = value

    put(self.opt_g, lr_g)
    put(self.opt_v, self.cfg.value_lr_mul * lr_g)
    put(self.opt_c, self.cfg.lr_c(it))

  def generator_losses(self, fake_input, z, states, progress, is_train=1, dropout_masks=None):
    """net.py:56-165: reward from the critic (``cfg.gan`` 'w': logit difference; 'ls': 1 - (logit - 1)^2), TD or
    plain-reward policy gradient (``cfg.use_TD``), over-exposure penalty (``cfg.use_penalty``).  ``cfg.supervised``
    (a paired-data critic that is not in the reference's ``critics.py``) is not built."""
    cfg = self.cfg
    if cfg.supervised:
      raise NotImplementedError('cfg.supervised: the paired-data reward path (net.py:100-102) is not built')
    assert cfg.gan in ('w', 'ls'), cfg.gan  # net.py:26
    (fake_output, new_states, surrogate, penalty), debug, _ = self.generator(
        (fake_input, z, states), is_train=is_train, progress=progress, dropout_masks=dropout_masks)
    # this step updates theta_g (through g_loss) and theta_v (through v_loss = f(old_value) only): the critic's and the
    # value net's convolution parameters take no part in differentiating fake_logit / new_value
    with frozen_parameters():
      fake_logit = self.critic(fake_output)
    if cfg.gan != 'ls':
      with torch.no_grad():
        fake_input_logit = self.critic(fake_input)
    old_value = self.value(fake_input, states)
    with frozen_parameters():
      new_value = self.value(fake_output, new_states)
    if (cfg.gan == 'w' and fake_logit.is_cuda and fake_logit.dtype == torch.float32 and STATE_STOPPED_DIM == 1 and
        STATE_STEP_DIM == 2 and os.environ.get('EXPO_FUSED_G_LOSSES', '1') == '1'):
      # the ~25 per-image scalar operations below (and their backward) as one launch each way
      g_loss, v_loss, reward, q_value = generator_losses_fused(
          fake_logit, fake_input_logit, new_value, old_value, new_states, penalty if cfg.use_penalty else None, surrogate,
          (cfg.all_reward, cfg.critic_logit_multiplier, cfg.discount_factor, cfg.parameter_lr_mul,
           cfg.maximum_trajectory_length), cfg.use_TD)
      return dict(g_loss=g_loss, v_loss=v_loss, fake_output=fake_output, new_states=new_states, reward=reward,
                  q_value=q_value, fake_logit=fake_logit, debug=debug)
    stopped = new_states[:, STATE_STOPPED_DIM:STATE_STOPPED_DIM + 1]
    clear_final = (new_states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1] > cfg.maximum_trajectory_length).float()
    new_value = new_value * (1.0 - clear_final)
    gate = cfg.all_reward + (1 - cfg.all_reward) * stopped
    if cfg.gan == 'ls':  # net.py:103-106: the LSGAN discriminator wants 1 for real
      raw_reward = gate * (1.0 - (fake_logit - 1.0)**2)
    else:
      raw_reward = gate * (fake_logit - fake_input_logit) * cfg.critic_logit_multiplier
    reward = raw_reward - penalty if cfg.use_penalty else raw_reward
    q_value = reward + (1.0 - stopped) * cfg.discount_factor * new_value
    advantage = q_value.detach() - old_value
    v_loss = (advantage**2).mean()
    if cfg.use_TD:  # net.py:135-140, 152-157 (the same in both GAN branches)
      routine_loss, weight = -q_value * cfg.parameter_lr_mul, -advantage
    else:
      routine_loss, weight = -reward, -reward
    g_loss = (routine_loss + surrogate * weight.detach()).mean()
    return dict(g_loss=g_loss, v_loss=v_loss, fake_output=fake_output, new_states=new_states, reward=reward,
                q_value=q_value, fake_logit=fake_logit, debug=debug)

  # -- gradient exchange: losses are means over the GLOBAL batch -> average the per-rank gradients
  def _collectives(self):
    return self.world_size > 1 or self.force_collectives

  def _bucket_ready(self, bucket):
    """Backward hook (exposure_amd.dist.GradBucket): the bucket's last gradient has just been accumulated
    -- start its all-reduce now, on RCCL's stream, while autograd keeps running the rest of the backward."""
    if self._collectives() and not bucket.launched:
      bucket.launched = True
      self._pending.append(bucket.all_reduce_mean(self.process_group, async_op=True, force=self.force_collectives))

  def _backward_into(self, loss, names, retain_graph=False, grad_tensors=None):
    """loss.backward restricted to the named buckets' parameters (the reference's optimize_loss
    ``variables=`` lists: theta_g sees only g_loss, theta_v only v_loss, theta_c only c_loss).  ``loss`` may be a list of
    tensors with ``grad_tensors`` (the hand-scheduled generator step enters the agent's graph at the retouched image, the
    surrogate and the penalty with their upstream gradients)."""
    def run(params):
      if isinstance(loss, (list, tuple)):
        torch.autograd.backward(list(loss), grad_tensors=list(grad_tensors), inputs=params, retain_graph=retain_graph)
      else:
        loss.backward(inputs=params, retain_graph=retain_graph)

    params = []
    if not self._collectives():
      # One rank, nothing to exchange: no flat buffer is needed.  With `.grad = None` autograd hands the optimiser the
      # gradient tensors it computed (AccumulateGrad keeps the parameter's layout) instead of ADDING each of them into
      # a zero-filled view: one launch less per parameter and backward pass -- 78 + 26 x citers of the 2 111 launches
      # of an iteration (profiles/r03_final_kernel_stats_train.csv), plus the buckets' fills.
      for name in names:
        b = self.buckets[name]
        b.release()
        params += b.params
      run(params)
      return
    for name in names:
      b = self.buckets[name]
      b.zero()
      params += b.params
    run(params)
    for name in names:  # a bucket whose hook could not fire (a parameter outside the graph): reduce it now
      self.buckets[name].disarm()
      self._bucket_ready(self.buckets[name])

  def _finish_collectives(self):
    for pend in self._pending:
      pend.wait()
    self._pending = []

  @staticmethod
  def _hip_adam(opt):
    return hasattr(opt, 'step_counter') and len(opt.param_groups) == 1

  def _generator_body(self, fake_input, z, states, progress, dropout_masks):
    if self.direct_generator and generator_direct.supported(self, fake_input, states):
      # the critic / value-net passes as two hand-scheduled batches of 2n images, their gradients entering the agent's
      # autograd graph in ONE backward pass (exposure_amd/generator_direct.py)
      # (HipAdam: the loss launch advances both step counters, so neither update needs a launch behind it)
      pre = self._hip_adam(self.opt_g) and self._hip_adam(self.opt_v)
      out = generator_direct.generator_step_losses_and_grads(
          self, fake_input, z, states, progress, dropout_masks,
          adam_steps=(self.opt_g.step_counter(), self.opt_v.step_counter()) if pre else (None, None))
    else:
      pre = False
      # (this step differentiates its convnets once: each stack of layers runs as one node, nn_ops.conv_trunk)
      with once_differentiable_convnets():
        out = self.generator_losses(fake_input, z, states, progress, 1, dropout_masks)
      # the value net's short backward first: its all-reduce then runs under the whole generator backward
      # (v_loss reaches theta_v through old_value only, g_loss through new_value / the critic / the agent: the two
      # backward passes share no graph nodes, so nothing needs to be retained)
      self._backward_into(out['v_loss'], ['v'])
      self._backward_into(out['g_loss'], ['g_head', 'g_trunk'])
    self._finish_collectives()
    if pre:
      self.opt_g.step(advanced=True)
      self.opt_v.step(advanced=True)
    else:
      self.opt_g.step()
      self.opt_v.step()
    return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}

  def _draw_masks(self, n, device=None):
    """The two always-on dropout masks of feature_extractor (agent.py:36), partition-invariant (self.rng)."""
    keep = self.cfg.dropout_keep_prob
    dev = device if device is not None else next(self.parameters()).device
    # both masks out of one draw, thresholded into float32 by one launch (were 2 x (rand, <, cast))
    u = self.rng.uniform(n, (self.cfg.feature_extractor_dims,), self.process_group, device=dev, lead=2)
    m = torch.empty(u.shape, dtype=torch.float32, device=u.device)
    torch.lt(u, keep, out=m)
    return [m[0], m[1]]

  def _draw_alpha(self, n):
    """net.py:170-172: alpha ~ U(0, 1) per image for the interpolation of the gradient penalty."""
    return self.rng.uniform(n, (1, 1, 1), self.process_group, device=next(self.parameters()).device)

  def generator_step(self, fake_input, z, states, progress, it=1, dropout_masks=None):
    """opt_g on g_loss w.r.t. theta_g and opt_v on v_loss w.r.t. theta_v (net.py:222-241).  ``fake_input`` / ``states``
    may be :class:`~exposure_amd.replay_memory.PoolRows` (rows of the replay memory gathered straight into the step
    graph's inputs)."""
    self.set_lrs(it, zero_g=(it == 0))
    masks = dropout_masks or self._draw_masks(fake_input.shape[0])
    if not self._replay_steps:
      fake_input, states = materialize(fake_input), materialize(states)
    if self._replay_steps:
      # (a device scalar filled in place: torch.as_tensor(float, device=...) is a blocking copy from pageable memory --
      # one full drain of the queue per iteration, with the G step's graph launch behind it)
      if self._progress is None or self._progress.device != fake_input.device:
        self._progress = torch.zeros((), dtype=torch.float32, device=fake_input.device)
      prog = self._progress.fill_(float(progress))
      return self._replay('g', self._generator_body_graph, (fake_input, z, states, prog, masks[0], masks[1]))
    return self._generator_body(fake_input, z, states, progress, masks)

  def _generator_body_graph(self, fake_input, z, states, progress, m0, m1):
    return self._generator_body(fake_input, z, states, progress, [m0, m1])

  def _check_heads(self):
    """A replayed step never runs ``Agent.forward``, so nothing would notice head parameters that were re-allocated
    behind the captured graphs (``gan.to(...)``, ``.half()``, a restore that replaces storages): the graphs would go on
    reading and updating the old packed buffers.  Two pointer comparisons per step; on a mismatch the heads are packed
    again (eagerly, here) and every captured step is dropped -- the next calls warm up and capture afresh."""
    pack = self._heads_pack
    if pack is None or (pack.quick_aliased() and pack.generation == self._heads_generation):
      return
    # the last replay may still be RUNNING: neither move the parameters under it nor destroy its graph exec in flight
    torch.cuda.current_stream().synchronize()
    pack.ensure()
    self._heads_generation = pack.generation
    if any(entry != 'warm' for entry in self._graphs.values()):
      import warnings
      warnings.warn('exposure_amd: the filter heads\' parameters were re-allocated after step graphs had been captured; '
                    'the captured steps are dropped and will be captured again')
    self._graphs.clear()

  # ---- hipGraph capture / replay of a whole optimisation step --------------------------------
  def _replay(self, key, body, inputs, extra_sig=(), rng=False):
    """hipGraph execution of one optimisation step.  Per (step kind, input shapes): the FIRST call
    runs eagerly (it is a normal training step and doubles as the warm-up that lets MIOpen /
    hipBLASLt pick kernels and allocate workspaces), the SECOND call captures ``body`` into one
    hipGraph, and from then on the inputs are copied into the static buffers and the graph is
    replayed: ~6 000 eager launches (~10 us of host time each) become one graph launch.  Returned
    tensors are the graph's static outputs (valid until the next replay of the same graph)."""
    self._check_heads()
    # ``extra_sig``: whatever else the captured launches depend on (buffers the body closes over); ``rng``: the body
    # draws from self.rng inside the graph (the generator is registered with it: replays continue the eager sequence)
    sig = (key,) + tuple((tuple(t.shape), t.dtype) for t in inputs) + tuple(extra_sig)
    entry = self._graphs.get(sig)
    if entry is None:
      self._graphs[sig] = 'warm'
      return body(*[materialize(t) for t in inputs])
    if entry == 'warm':
      # (with a process group the gradient all-reduces are captured too: RCCL collectives are
      # stream-ordered kernels, and the eager first call has already initialised the communicator)
      static_in = [t.materialize() if isinstance(t, (PoolRows, HostStaged)) else t.clone() for t in inputs]
      # ROOT CAUSE of round 1's "one run in ~15 aborts" (gpurun r02soak, 3 of 28 runs): the eager first call left
      # WorkNCCL entries in ProcessGroupNCCL's watchdog list; they are complete, but the watchdog only reaps its list
      # every ~100 ms.  If it polls (hipEventQuery on the work's end event) AFTER this thread has pulled RCCL's stream
      # into the capture, HIP answers hipErrorCapturedEvent for an event that belongs to a now-capturing stream, the
      # watchdog thread throws, and the process aborts.  Works issued DURING capture are never enqueued, so the list
      # only has to be EMPTY when the capture starts: drain_before_capture waits until the flight recorder shows every
      # earlier collective retired (deterministic); when it cannot verify that (recorder off) a multi-rank step
      # stays eager unless EXPO_GRAPH_COLLECTIVES=1 accepts the timed grace period of round 2.
      verified = xdist.drain_before_capture(issued_collectives=self._collectives())
      if not verified and self.world_size > 1 and self._graph_collectives != '1':
        import warnings
        warnings.warn('exposure_amd: the NCCL watchdog drain could not be verified (flight recorder off?); steps with '
                      'collectives stay eager (set TORCH_FR_BUFFER_SIZE>0, or EXPO_GRAPH_COLLECTIVES=1)')
        self.use_graphs = self._replay_steps = False
        return body(*[materialize(t) for t in inputs])
      self.capture_drain_verified = verified
      graph = torch.cuda.CUDAGraph()
      if rng and self.rng.gen is not None and self.rng.gen.device.type == 'cuda':
        graph.register_generator_state(self.rng.gen)
      # thread_local: RCCL's watchdog thread polls events while this thread captures; under the
      # default "global" mode such a call from another thread aborts the process
      mode = 'thread_local' if xdist.world_size(self.process_group) > 1 or self.force_collectives else 'global'
      try:
        # (no cyclic garbage collection inside the capture: util.capture_without_gc)
        with capture_without_gc(), torch.cuda.graph(graph, capture_error_mode=mode):
          static_out = body(*static_in)
      except RuntimeError as e:  # e.g. a collective that refuses capture: keep training, eagerly
        import warnings
        warnings.warn('hipGraph capture of the %r step failed (%s); continuing with eager launches' % (key, e))
        torch.cuda.synchronize()
        self.use_graphs = self._replay_steps = False
        return body(*[materialize(t) for t in inputs])
      entry = (graph, static_in, static_out)
      self._graphs[sig] = entry
    graph, static_in, static_out = entry
    for dst, src in zip(static_in, inputs):
      if isinstance(src, (PoolRows, HostStaged)):
        src.into(dst)  # gathered / copied from pinned memory straight into the graph's input: one launch, not two
      else:
        dst.copy_(src)
    graph.replay()
    return static_out

  def critic_losses(self, real_data, fake_output, alpha=None):
    """net.py:126-199.  ``cfg.gan == 'w'``: c_loss = mean(fake - real) + lambda * mean(max(||grad||-1, 0)^2) (without
    the penalty term when lambda <= 0: the weights are clipped after the update instead); ``'ls'``:
    c_loss = mean(fake^2) + mean((real - 1)^2).  ``fake_output`` is FED, as in the reference: the critic update runs on terminated images
    replayed from the memory (``replay_memory.py:168-185`` feeds the ``fake_output`` tensor), the
    generator is not executed."""
    cfg = self.cfg
    n = real_data.shape[0]
    if cfg.gan == 'ls':
      fake_output = fake_output.detach().float()
      real_data = real_data.float()
      fake_output = fake_output.requires_grad_(True)  # for the reported d fake_logit / d fake_output
      logits = self.critic(torch.cat([real_data, fake_output], dim=0))
      real_logit, fake_logit = logits[:n], logits[n:]
      # net.py:129-147, 195-199: least-squares discriminator, no gradient penalty; the reported norm is that of
      # d fake_logit / d fake_output (no epsilon), a summary value only
      c_loss = (fake_logit**2).mean() + ((real_logit - 1.0)**2).mean()
      with skip_parameter_gradients():
        fake_gradients, = torch.autograd.grad(fake_logit.sum(), fake_output, retain_graph=True)
      gradient_norm = torch.sqrt((fake_gradients**2).sum(dim=(1, 2, 3)))
      zero = torch.zeros((), device=c_loss.device)
      return dict(c_loss=c_loss, emd=c_loss.detach(), gradient_norm=gradient_norm.mean().detach(),
                  gradient_penalty=zero, c_average=zero)
    if alpha is None:
      alpha = self._draw_alpha(n)
    # one batched pass for the real and the fake half (the critic is per-sample: no normalisation layers), so their
    # forward and backward are single launches of twice the batch; both critic inputs come out of ONE kernel
    # (nn_ops.critic_step_inputs: dtype conversions, concatenation, interpolation)
    both, interpolated = critic_step_inputs(real_data, fake_output.detach(), alpha)
    interpolated.requires_grad_(True)
    logits = self.critic(both)
    real_logit, fake_logit = logits[:n], logits[n:]
    c_loss = (fake_logit - real_logit).mean()
    use_gp = cfg.gradient_penalty_lambda > 0
    inte_logit = self.critic(interpolated)
    with skip_parameter_gradients():  # only d D / d x^ is wanted here; theta_c is reached by the OUTER backward
      gradients, = torch.autograd.grad(inte_logit, interpolated, torch.ones_like(inte_logit), create_graph=use_gp)
    term, gradient_norm = grad_penalty_term(gradients)  # max(||g|| - 1, 0)^2 and ||g|| = sqrt(1e-6 + sum g^2) per image
    gradient_penalty = cfg.gradient_penalty_lambda * term.mean()
    # net.py:188-199: without the penalty (lambda <= 0) the norm is still reported and theta_c is clipped after the
    # update instead (clip_critic_weights)
    total = c_loss + gradient_penalty if use_gp else c_loss
    with torch.no_grad():
      c_average = (fake_logit + real_logit).mean() * 0.5
    return dict(c_loss=total, emd=-c_loss.detach(), gradient_norm=gradient_norm.mean().detach(),
                gradient_penalty=gradient_penalty.detach(), c_average=c_average)

  def clip_critic_weights(self):
    """net.py:252-262: WGAN without the gradient penalty clamps every critic variable (biases too) to
    +-cfg.clamp_critic after each critic update."""
    cfg = self.cfg
    if cfg.gan == 'w' and cfg.gradient_penalty_lambda <= 0:
      params = [p.data for p in self.critic.parameters()]
      torch._foreach_clamp_min_(params, -float(cfg.clamp_critic))
      torch._foreach_clamp_max_(params, float(cfg.clamp_critic))

  def _critic_body(self, real_data, fake_output, alpha):
    ema_done = False
    if self.direct_critic and critic_direct.supported(self, real_data, fake_output):
      # on one rank the reporting launch also advances the logit centre's average (with collectives c_average is
      # averaged over the ranks first, below)
      ema_done = not self._collectives()
      if ema_done and (self._c_ema is None or self._c_ema.device != real_data.device):
        self._c_ema = torch.zeros((), dtype=torch.float32, device=real_data.device)
      pre = self._hip_adam(self.opt_c)  # the reporting launch advances Adam's step counter as well
      out = critic_direct.critic_losses_and_grads(self, real_data, fake_output, alpha, self._c_ema if ema_done else None,
                                                  adam_step=self.opt_c.step_counter() if pre else None)
      self._bucket_ready(self.buckets['c'])  # (a no-op on one rank: the gradients already sit in p.grad)
    else:
      pre = False
      out = self.critic_losses(materialize(real_data), materialize(fake_output), alpha)
      self._backward_into(out['c_loss'], ['c'])
    if self._collectives():
      ca = out['c_average'].clone()
      xdist.all_reduce_mean_(ca, self.process_group, force=self.force_collectives)
      out['c_average'] = ca
    self._finish_collectives()
    if pre:
      self.opt_c.step(advanced=True)
    else:
      self.opt_c.step()
    self.clip_critic_weights()
    ca = out['c_average'].detach().reshape(())
    if self._c_ema is None or self._c_ema.device != ca.device:
      self._c_ema = torch.zeros((), dtype=ca.dtype, device=ca.device)
    if not ema_done:
      self._c_ema.lerp_(ca, 0.01)  # update_average (net.py:165-168, 267-268): 0.99 * average + 0.01 * c_average
    return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}

  def critic_step(self, real_data, fake_output, it=1, alpha=None):
    """opt_c on c_loss w.r.t. theta_c (net.py:245-251); ``fake_output`` may be
    :class:`~exposure_amd.replay_memory.PoolRows`."""
    self.set_lrs(it)
    if alpha is None:
      alpha = self._draw_alpha(real_data.shape[0])
    if not self._replay_steps:
      real_data, fake_output = materialize(real_data), materialize(fake_output)
    if self._replay_steps:
      out = dict(self._replay('c', self._critic_body, (real_data, fake_output, alpha)))
    else:
      out = self._critic_body(real_data, fake_output, alpha)
    self.c_average_steps += 1
    return out

  @property
  def c_average_biased(self):
    return self._c_ema if self._c_ema is not None else 0.0

  def c_average_smoothed(self):
    """The zero-debiased moving average of the logit centre (net.py:167; a reported value: nothing on the path reads
    it), computed when asked for."""
    if self.c_average_steps == 0:
      return 0.0
    return self._c_ema / (1.0 - 0.99**self.c_average_steps)

  # -- one regular iteration (net.py:329-365): ONE graph replay fed by ONE host-to-device copy ---------------------------
  def _iteration_layout(self, memory, n):
    cfg = self.cfg
    p = memory.target_pool_size
    ni = 2 * n + 2 * p + 2 * n * cfg.citers  # int64: G gather / scatter, fresh dst / src, per critic step slots + real rows
    nf = n * cfg.z_dim + 4                   # float32: z, progress, lr_g, lr_v, lr_c
    return ni, nf

  def _iteration_body(self, rec):
    """The captured iteration: ``rec`` is the plan on the device (``ReplayMemory.plan_iteration`` + the iteration's
    scalars, one uint8 record).  Learning rates and progress are read from it, the pool is gathered from / scattered to with
    its index vectors, dropout masks and alpha are drawn in the graph (self.rng is registered with it)."""
    cfg, mem, n = self.cfg, self._it_memory, self._it_n
    ni, nf = self._iteration_layout(mem, n)
    p = mem.target_pool_size
    i64, f32 = rec[:8 * ni].view(torch.int64), rec[8 * ni:8 * ni + 4 * nf].view(torch.float32)
    at = [0]

    def take(k):
      v = i64[at[0]:at[0] + k]
      at[0] += k
      return v

    g_slots, g_scatter, fresh_dst, fresh_src = take(n), take(n), take(p), take(p)
    z = f32[:n * cfg.z_dim].view(n, cfg.z_dim)
    scal = f32[n * cfg.z_dim:]
    for opt, k in ((self.opt_g, 1), (self.opt_v, 2), (self.opt_c, 3)):
      for g in opt.param_groups:
        g['lr'].copy_(scal[k])
    fake_input, states, feats = mem.planned_generator_batch(g_slots)
    masks = self._draw_masks(n, device=fake_input.device)
    g_out = self._generator_body(fake_input, z, states, scal[0], masks)
    mem.planned_commit(g_scatter, g_out['fake_output'], g_out['new_states'], feats, fresh_dst, fresh_src)
    c_out = None
    for _ in range(cfg.citers):
      # (rows of the pool / the data set, read in place by the hand-scheduled update's first launch)
      fake, real = take(n), take(n)
      real, fake = mem.planned_critic_batch(fake, real, lazy=True)
      c_out = self._critic_body(real, fake, self._draw_alpha(n))
    return dict(g=g_out, c=c_out)

  def _iteration_stepwise(self, memory, it, progress, n):
    cfg = self.cfg
    feed, features = memory.get_feed_dict_and_states(n, lazy=True)
    g_out = self.generator_step(feed['fake_input'], feed['z'], feed['states'], progress, it=it)
    memory.replace_memory(g_out['fake_output'], g_out['new_states'], features, advanced=True)
    c_out = None
    for _ in range(cfg.citers):
      feed = memory.get_replay_feed_dict(n, lazy=True)
      c_out = self.critic_step(feed['real_data'], feed['fake_output'], it=it)
    return dict(g=g_out, c=c_out)

  def train_iteration(self, memory, it, progress=None, batch_size=None):
    """One REGULAR training iteration (net.py:329-365: one generator / value step, its results back into the replay
    memory, ``cfg.citers`` critic steps on replayed terminated records) -> ``dict(g=<generator_step's outputs>,
    c=<the last critic_step's>)``.

    With step graphs on and both data sets resident in HBM, nothing the host decides during an iteration depends on what
    the device computes (the pool's decisions read the host mirror of step / stopped), so the iteration is planned ahead
    (``ReplayMemory.plan_iteration``: same decisions, same random draws as the step-by-step calls), the plan -- index
    vectors, selection noise, progress, the three learning rates -- goes to the device as ONE pinned-memory copy, and the
    whole iteration replays as ONE hipGraph: pool gathers, the G / V step, the scatter of its results and the fresh
    records, the critic steps with their replayed batches, dropout masks and alpha from the graph-registered generator.
    Everything else (eager mode, the warm-up iteration 0 and its 100-step phases, providers that stream from the host)
    takes the step-by-step calls."""
    cfg = self.cfg
    n = int(batch_size or cfg.batch_size)
    progress = float(it) / cfg.max_iter_step if progress is None else float(progress)
    plan = None
    if self._replay_steps and it > 0 and int(cfg.giters) == 1 and cfg.citers >= 1:
      plan = memory.plan_iteration(n, cfg.citers)
    if plan is None:
      return self._iteration_stepwise(memory, it, progress, n)
    dev = memory.device
    ni, nf = self._iteration_layout(memory, n)
    # (a ring of its own, four buffers deep: pinning memory is a device-wide synchronisation, not something to do per
    # iteration; a buffer is rewritten only after the copy that read it has completed)
    if self._it_ring is None or self._it_ring.device != dev:
      from .replay_memory import _PinnedRing
      self._it_ring = _PinnedRing(dev, slots=4)
    host, key = self._it_ring.take(torch.uint8, 8 * ni + 4 * nf)
    fields = plan.int64_fields()
    host[:8 * ni].view(torch.int64).numpy()[:] = np.concatenate([np.asarray(f, dtype=np.int64).ravel() for f in fields])
    f32 = host[8 * ni:].view(torch.float32).numpy()
    f32[:n * cfg.z_dim] = plan.z.numpy().ravel()
    lr_g = cfg.lr_g(it)
    f32[n * cfg.z_dim:] = (progress, lr_g, cfg.value_lr_mul * lr_g, cfg.lr_c(it))
    for opt in (self.opt_g, self.opt_v, self.opt_c):
      for g in opt.param_groups:
        g.pop('_lr_value', None)  # (set_lrs' note of what the scalar holds: the graph writes it behind its back)
    if self._c_ema is None or self._c_ema.device != dev:
      self._c_ema = torch.zeros((), dtype=torch.float32, device=dev)  # (allocated outside any capture)
    self._it_memory, self._it_n = memory, n
    extra = (id(memory), memory._img.data_ptr(), memory._st.data_ptr(), memory.fake_dataset.images.data_ptr(),
             memory.real_dataset.images.data_ptr(), cfg.citers)
    out = self._replay('it', self._iteration_body, (HostStaged(host, dev, self._it_ring, key),), extra_sig=extra, rng=True)
    self.c_average_steps += cfg.citers
    return out

  # -- the training loop (net.py:298-403), minus visualisation / checkpoints / TensorBoard
  def train(self, memory, max_iter_step=None, log_every=0, log=print):
    cfg = self.cfg
    max_iter_step = cfg.max_iter_step if max_iter_step is None else max_iter_step
    history = []
    for it in range(max_iter_step + 1):
      progress = float(it) / cfg.max_iter_step
      if cfg.gan == 'w' and (it < cfg.critic_initialization or it % 500 == 0):
        citers = 100
      else:
        citers = cfg.citers
      giters = 100 if it == 0 else cfg.giters  # make sure there are terminating states
      if giters == 1 and citers == cfg.citers:
        out = self.train_iteration(memory, it, progress)
        g_out, c_out = out['g'], out['c']
      else:
        g_out = None
        for _ in range(giters):
          feed, features = memory.get_feed_dict_and_states(cfg.batch_size, lazy=True)
          g_out = self.generator_step(feed['fake_input'], feed['z'], feed['states'], progress, it=it)
          memory.replace_memory(g_out['fake_output'], g_out['new_states'], features, advanced=True)
        c_out = None
        for _ in range(citers):
          feed = memory.get_replay_feed_dict(cfg.batch_size, lazy=True)
          c_out = self.critic_step(feed['real_data'], feed['fake_output'], it=it)
      # the four reported scalars stay on the device (one small launch; the steps' outputs are static graph buffers that
      # the next replay overwrites): the host reads them when it logs and once at the end, not four times per iteration
      history.append(torch.stack([g_out['g_loss'].reshape(()), g_out['v_loss'].reshape(()), c_out['emd'].reshape(()),
                                  c_out['gradient_norm'].reshape(())]).float())
      if log_every and it % log_every == 0:
        g_loss, v_loss, emd, cgn = history[-1].tolist()
        log('it%6d, g_loss=%.2f, v_loss=%.2f, EMD=%.3f, cgn=%.2f  %s' % (it, g_loss, v_loss, emd, cgn, memory.debug()))
    values = torch.stack(history).cpu().tolist() if history else []
    return [dict(iter=it, g_loss=v[0], v_loss=v[1], emd=v[2], cgn=v[3]) for it, v in enumerate(values)]
