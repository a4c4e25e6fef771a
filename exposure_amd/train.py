"""``python -m exposure_amd.train [--iters N]`` -- the tensor part of the reference's
``train.py:9-14`` / ``GAN.train`` (``net.py:298-403``) on synthetic FiveK-shaped data: the G/V and
critic alternation with the device-resident replay memory, one hipGraph replay per optimisation
step.  Dataset loading, TensorBoard, PNG dashboards and checkpoints of the reference are out of scope
(SURVEY.md section 2); ``--save`` writes weights + optimiser state with ``torch.save`` (``--resume`` reads it), ``--save-tf`` a TF-1 checkpoint
(``checkpoint.py``, ``tf_bundle.py``).

Note on the numbers it prints: with random-init weights the policy can chain Exposure (x11) and
Gamma (power 3) steps, so pixel values -- and with them the over-exposure penalty, the critic logits
and the value targets -- occasionally overflow float32 in the first iterations.  That is the
reference's arithmetic (``filters.py:181-182, 205-206`` have no clamp; ``cfg.clamp`` is off), not
a kernel artefact; ``--clamp`` turns on the reference's own ``clip_by_value(net, 0, 5)``
(``agent.py:240-241``)."""
import argparse
import time

import torch

from .config import make_cfg
from .gan import GAN
from .replay_memory import ReplayMemory, ResidentProvider


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=20, help='training iterations to run (the reference runs 20000)')
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--log-every', type=int, default=10)
  ap.add_argument('--no-graphs', action='store_true')
  ap.add_argument('--save', default=None, help="torch.save of {'model': state_dict, 'optim': Adam slots, step counters, "
                  "learning rates and the logit centre's average} -- what the reference's tf.train.Saver keeps (net.py:271)")
  ap.add_argument('--resume', default=None, help='a file written by --save: weights and optimiser state are restored in place')
  ap.add_argument('--save-tf', default=None, metavar='MODEL_DIR',
                  help="also write MODEL_DIR/model.ckpt-<iters> in TensorFlow's checkpoint format, variable names and "
                  "layouts as the reference's graph declares them (net.py:380-384)")
  ap.add_argument('--clamp', action='store_true', help='cfg.clamp = True (agent.py:240-241)')
  ap.add_argument('--dtype', default='f32', choices=['f32', 'f16'], help='storage type of the image pool')
  args = ap.parse_args(argv)
  dev = torch.device('cuda:0')
  torch.manual_seed(args.seed)
  cfg = make_cfg()
  cfg.clamp = bool(args.clamp)
  gan = GAN(cfg, device=dev, use_graphs=not args.no_graphs, seed=args.seed)  # (--seed also drives dropout / alpha)
  dt = torch.float32 if args.dtype == 'f32' else torch.float16
  # toy task with the statistics of the real one: dark linear-RAW-like inputs, brighter targets
  # (both synthetic data sets resident in HBM: 4 096 images each, served as views)
  memory = ReplayMemory(cfg, ResidentProvider(dev, gamma=2.2, scale=0.35, dtype=dt, seed=args.seed + 1),
                        ResidentProvider(dev, gamma=1.2, scale=0.9, dtype=dt, seed=args.seed + 2), seed=args.seed)
  if args.resume:
    ckpt = torch.load(args.resume, map_location=dev)
    gan.load_state_dict(ckpt['model'] if 'model' in ckpt else ckpt)
    if 'optim' in ckpt:
      gan.load_optimizer_state_dict(ckpt['optim'])
  t0 = time.time()
  hist = gan.train(memory, max_iter_step=args.iters, log_every=args.log_every)
  torch.cuda.synchronize()
  dt = time.time() - t0
  print('%d iterations in %.1f s (iteration 0 = 100 warm-up generator steps + 100 critic steps, net.py:314-323)' %
        (len(hist), dt))
  if args.save:
    torch.save({'model': gan.state_dict(), 'optim': gan.optimizer_state_dict()}, args.save)
  if args.save_tf:
    from . import checkpoint
    print('wrote', checkpoint.save(gan, args.save_tf, args.iters))


if __name__ == '__main__':
  main()
