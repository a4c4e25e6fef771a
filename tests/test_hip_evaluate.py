"""BASELINE config 1 on the HIP path: `evaluate.py` on ONE 64x64 16-bit TIFF with
cfg.filters = [ExposureFilter, GammaFilter] (evaluate.py:8-31; net.py:726-747, 779, 796-821), through
the real CLI entry (`exposure_amd.evaluate.main`): load_image -> retouch (5 steps on the proxy, the
recorded operations applied to the full-resolution tensor by libexposure_hip.so) -> .npy.  The written
result is compared with an oracle replay of the recorded operations on the same loaded image."""
import os

import numpy as np
import pytest
import torch

from exposure_amd import evaluate
from exposure_amd.tiff16 import write_tiff
from oracle import filters_np as fnp
from tests._tol import assert_image_close

pytestmark = pytest.mark.gpu


def replay(image, rec, dtype):
  """Oracle replay of the recorded (filter id, parameters) sequence on the loaded image.  fused: fp32 between
  the steps, ONE rounding to the storage dtype at the end; stepwise: one rounding per step."""
  x = image.astype(dtype).astype(np.float64)[None]
  outs = []
  for fid, p24 in zip(rec['abi_filter_ids'], rec['params24']):
    p = p24[None, :fnp.NUM_PARAMS[fid]].astype(np.float64)
    x = fnp.process_packed(fid, x, p)
    outs.append(x)
  return x[0], outs


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_evaluate_cli_on_a_64x64_tiff_with_exposure_and_gamma(gpu_device, tmp_path, dtype):
  rng = np.random.default_rng(5)
  raw = (rng.random((64, 64, 3))**1.5 * 40000).astype(np.uint16)  # a dark 16-bit ProPhoto "RAW export"
  tif = str(tmp_path / 'a0001.tif')
  write_tiff(tif, raw)
  out_dir = str(tmp_path / 'outputs') + os.sep
  recs = evaluate.main(['--filters', 'E,G', '--seed', '7', '--dtype', dtype, '--out', out_dir, tif])
  assert len(recs) == 1
  rec = recs[0]
  assert rec['filters'] == [{0: 'E', 1: 'G'}[i] for i in rec['abi_filter_ids']] and len(rec['filters']) == 5
  assert rec['states'][:3] == [1.0, 1.0, 5.0]  # submitted after test_steps = 5 (agent.py:210-217)
  got = np.load(rec['output'])
  assert got.shape == (64, 64, 3) and got.dtype == np.float32
  np_dtype = np.float16 if dtype == 'f16' else np.float32
  image = evaluate.load_image(tif)
  assert np.abs(image - (raw / 65535.0)**1.8).max() < 1e-6
  ref, _ = replay(image, rec, np_dtype)
  assert_image_close(got, ref, np_dtype, 'evaluate.main (%s)' % dtype)
  assert np.isfinite(got).all() and float(np.abs(got - image).max()) > 1e-3  # it did retouch something


def test_evaluate_cli_non_tif_branch_and_two_images(gpu_device, tmp_path):
  """net.py:735-747 (8-bit sRGB-ish input) + several inputs with distinct outputs; full 8-filter cfg,
  non-square high-resolution image (the proxy is the centre crop, net.py:779)."""
  from PIL import Image
  rng = np.random.default_rng(6)
  paths = []
  for k, (h, w) in enumerate([(96, 160), (128, 80)]):
    p = str(tmp_path / ('img%d.png' % k))
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
    paths.append(p)
  recs = evaluate.main(['--seed', '8', '--dtype', 'f16', '--out', str(tmp_path / 'res.npy'), *paths])
  assert len({r['output'] for r in recs}) == 2
  for p, rec in zip(paths, recs):
    image = evaluate.load_image(p)
    got = np.load(rec['output'])
    assert got.shape == image.shape
    ref, _ = replay(image, rec, np.float16)
    assert_image_close(got, ref, np.float16, 'evaluate.main non-tif')


def test_png_outputs_of_the_cli(gpu_device, tmp_path):
  """--png / --show-input: the reference's `<name>.retouched.png` and `.input_tone_mapped.png` (net.py:769-772,
  822-832) next to the linear .npy."""
  from PIL import Image
  raw = (np.random.default_rng(9).random((64, 64, 3)) * 40000).astype(np.uint16)
  tif = str(tmp_path / 'p.tif')
  write_tiff(tif, raw)
  rec = evaluate.main(['--filters', 'E,G', '--seed', '2', '--png', '--show-input', '--out', str(tmp_path / 'p.npy'), tif])[0]
  lin = np.load(rec['output'])
  png = np.asarray(Image.open(rec['png']['retouched']).convert('RGB'))
  assert np.array_equal(png, np.clip(np.rint(lin * 255.0), 0, 255).astype(np.uint8))
  src = evaluate.load_image(tif).astype(np.float16).astype(np.float32)  # the CLI's default storage dtype
  tone = np.asarray(Image.open(rec['png']['input_tone_mapped']).convert('RGB')).astype(np.int32)
  want = np.clip(np.rint(evaluate.tone_mapped_input(src) * 255.0), 0, 255).astype(np.int32)
  assert np.abs(tone - want).max() <= 1


def test_stepwise_schedule_matches_fused(gpu_device, tmp_path):
  """--stepwise = the reference's schedule (the high-res tensor filtered at every step, net.py:796-821):
  same operations, one fp16 rounding per step instead of one at the end."""
  rng = np.random.default_rng(9)
  raw = (rng.random((48, 72, 3))**1.5 * 30000).astype(np.uint16)
  tif = str(tmp_path / 'b.tif')
  write_tiff(tif, raw)
  a = evaluate.main(['--filters', 'E,G', '--seed', '3', '--dtype', 'f32', '--out', str(tmp_path / 'f.npy'), tif])[0]
  b = evaluate.main(['--filters', 'E,G', '--seed', '3', '--dtype', 'f32', '--stepwise', '--out',
                     str(tmp_path / 's.npy'), tif])[0]
  assert a['abi_filter_ids'] == b['abi_filter_ids']
  fa, fb = np.load(a['output']), np.load(b['output'])
  assert np.abs(fa - fb).max() <= 2e-5 * max(1.0, np.abs(fa).max())


def test_train_save_then_evaluate_round_trip(gpu_device, tmp_path):
  from exposure_amd import train
  w = str(tmp_path / 'gan.pt')
  train.main(['--iters', '1', '--no-graphs', '--clamp', '--save', w, '--log-every', '0'])
  raw = (np.random.default_rng(2).random((64, 64, 3)) * 30000).astype(np.uint16)
  tif = str(tmp_path / 'c.tif')
  write_tiff(tif, raw)
  rec = evaluate.main(['--weights', w, '--seed', '1', '--out', str(tmp_path / 'o.npy'), tif])[0]
  assert np.isfinite(np.load(rec['output'])).all()


def test_tf_checkpoint_round_trip_through_the_cli(gpu_device, tmp_path):
  """f-4: train -> `saver.save`-format checkpoint (net.py:380-384) -> evaluate restoring it like evaluate.py:27-28
  restores bit-for-bit the parameters of the torch state dict of the same weights and retouches the same; a checkpoint
  with the optimizer's slot variables in it restores the same."""
  from exposure_amd import checkpoint, tf_bundle, train
  w = str(tmp_path / 'gan.pt')
  model_dir = str(tmp_path / 'models' / 'example' / 'run')
  train.main(['--iters', '1', '--no-graphs', '--clamp', '--save', w, '--save-tf', model_dir, '--log-every', '0'])
  assert os.path.exists(os.path.join(model_dir, 'model.ckpt-1.index'))
  raw = (np.random.default_rng(5).random((96, 64, 3)) * 30000).astype(np.uint16)
  tif = str(tmp_path / 'd.tif')
  write_tiff(tif, raw)
  # the restored parameters are bit-for-bit those of the torch state dict
  from exposure_amd.agent import Agent
  from exposure_amd.config import make_cfg
  ag_a = evaluate.load_agent_weights(Agent(make_cfg()), torch.load(w, map_location='cpu'))
  ag_b = Agent(make_cfg())
  assert checkpoint.restore(ag_b, model_dir, 1) == []
  for (name, pa), pb in zip(ag_a.named_parameters(), ag_b.parameters()):
    assert torch.equal(pa, pb), name
  a = evaluate.main(['--weights', w, '--seed', '3', '--out', str(tmp_path / 'a.npy'), tif])[0]
  b = evaluate.main(['--tf-checkpoint', model_dir, '--ckpt', '1', '--seed', '3', '--out', str(tmp_path / 'b.npy'), tif])[0]
  # (two runs of the convnets agree to the last bit or two, not bit for bit: MIOpen picks its kernels per process state)
  assert a['filters'] == b['filters']
  np.testing.assert_allclose(a['params24'], b['params24'], rtol=2e-5, atol=1e-6)
  oa, ob = np.load(a['output']), np.load(b['output'])
  assert np.abs(oa - ob).max() <= 2.0**-9 * max(1.0, np.abs(oa).max())
  # what TF itself writes holds more than the trainable variables
  d = tf_bundle.read_bundle(checkpoint.checkpoint_prefix(model_dir, 1))
  extra = dict(d)
  for k, v in d.items():
    extra[k + '/Adam'] = np.zeros_like(v)
    extra[k + '/Adam_1'] = np.zeros_like(v)
  extra['beta1_power'] = np.float32(0.5)
  tf_bundle.write_bundle(checkpoint.checkpoint_prefix(model_dir, 20000), extra)
  c = evaluate.main(['--tf-checkpoint', model_dir, '--seed', '3', '--out', str(tmp_path / 'c.npy'), tif])[0]
  assert c['filters'] == a['filters']
  assert np.abs(np.load(c['output']) - oa).max() <= 2.0**-9 * max(1.0, np.abs(oa).max())
  with pytest.raises(SystemExit):
    evaluate.main(['--tf-checkpoint', model_dir, '--weights', w, tif])


def test_histogram_intersection_metric_on_device(gpu_device):
  """histogram_intersection.py:11-31,62-76 on GPU tensors (fp16 storage, as the retouched images are) equals the
  CPU evaluation of the same images, and retouched-vs-target behaves like a similarity."""
  from exposure_amd import metrics
  g = torch.Generator().manual_seed(4)
  out = (torch.rand((256, 64, 64, 3), generator=g)**2.0).half()
  tgt = (torch.rand((256, 64, 64, 3), generator=g)**0.7).half()
  ints_cpu, avg_cpu = metrics.histogram_intersection(out, tgt)
  ints_gpu, avg_gpu = metrics.histogram_intersection(out.to(gpu_device), tgt.to(gpu_device))
  st_cpu, st_gpu = metrics.get_statistics(out), metrics.get_statistics(out.to(gpu_device)).cpu()
  np.testing.assert_allclose(st_gpu.numpy(), st_cpu.numpy(), rtol=2e-5, atol=2e-6)
  # a statistic within rounding of a bin edge may change bins between devices: at most one image per histogram
  for a, b in zip(ints_cpu, ints_gpu):
    assert abs(a - b) <= 2.0 / 256 + 1e-6
  assert abs(avg_cpu - avg_gpu) <= 2.0 / 256 + 1e-6
  same, avg_same = metrics.histogram_intersection(out.to(gpu_device), out.to(gpu_device))
  assert all(abs(v - 1.0) < 1e-6 for v in same) and avg_gpu < avg_same


def test_train_resume_restores_optimiser_state_in_place(gpu_device, tmp_path):
  """`train --save` keeps what the reference's tf.train.Saver keeps beside the weights (Adam slots, step counters, the
  logit centre's average; net.py:271) and `--resume` puts it back IN PLACE; a checkpoint without moments / average zeroes
  the live buffers instead of dropping them (captured step graphs keep reading them) -- advisor, round 5."""
  from exposure_amd import train
  from exposure_amd.config import make_cfg
  from exposure_amd.gan import GAN
  w = str(tmp_path / 'gan.pt')
  train.main(['--iters', '1', '--no-graphs', '--clamp', '--save', w, '--log-every', '0'])
  ckpt = torch.load(w, map_location=gpu_device)
  assert set(ckpt) == {'model', 'optim'} and ckpt['optim']['c_average_steps'] > 0
  assert len(ckpt['optim']['opt_c']['state']) == len(list(GAN(make_cfg()).critic.parameters()))
  train.main(['--iters', '1', '--no-graphs', '--clamp', '--resume', w, '--log-every', '0'])
  gan = GAN(make_cfg(), device=gpu_device)
  gan.load_state_dict(ckpt['model'])
  gan.load_optimizer_state_dict(ckpt['optim'])
  p = next(iter(gan.critic.parameters()))
  m, v = gan.opt_c.state[p]
  assert float(m.abs().max()) > 0 and float(gan.opt_c._step) == float(ckpt['optim']['opt_c']['state'][0]['step'])
  ema = gan._c_ema
  assert ema is not None and gan.c_average_steps == ckpt['optim']['c_average_steps']
  ptrs = (m.data_ptr(), v.data_ptr(), ema.data_ptr())
  bare = dict(ckpt['optim'], c_ema=None, opt_c=dict(ckpt['optim']['opt_c'], state={}))
  gan.load_optimizer_state_dict(bare)
  m2, v2 = gan.opt_c.state[p]
  assert (m2.data_ptr(), v2.data_ptr(), gan._c_ema.data_ptr()) == ptrs
  assert float(m2.abs().max()) == 0.0 and float(v2.abs().max()) == 0.0 and float(gan._c_ema) == 0.0
  assert gan.c_average_steps == 0
