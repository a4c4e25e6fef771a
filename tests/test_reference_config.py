"""``exposure_amd.config.make_cfg()`` against the reference's own configuration files EXECUTED in the build container
(tests/golden/reference_config.json, made by tests/golden/make_reference_config.py: the import lines of
``config_example.py`` / ``config_sintel.py`` dropped, the names they would bind replaced by empty classes of the same name).
Every plain field the package's cfg carries must equal the reference's, the filter order must be the reference's, the
learning-rate callbacks must agree at every probed iteration, and the Adam constants must be the ones in the reference's
optimizer lambda.  Fields only one side has are listed explicitly, so a new or a dropped field shows up here."""
import json
import os

import pytest

from exposure_amd.config import make_cfg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# reference fields the package does not carry: data providers / optimizers (TF objects), and display switches of the
# reference's training dashboard (out of scope: SURVEY.md section 2)
REFERENCE_ONLY = {'summary_freq', 'vis_draw_critic_scores', 'vis_step_test', 'realtime_vis', 'write_image_interval'}
# fields of the package's cfg the reference keeps in module-level variables / lambdas instead of cfg
PACKAGE_ONLY = {'lr_decay', 'base_lr', 'lr_segments', 'generator_lr_mul', 'critic_lr_mul', 'adam_beta1', 'adam_beta2',
                'hsv_grad_mode'}


@pytest.fixture(scope='module')
def ref():
  return json.load(open(os.path.join(HERE, 'reference_config.json')))


@pytest.mark.parametrize('name', ['config_example.py', 'config_sintel.py'])
def test_cfg_equals_the_reference_configuration(ref, name):
  r = ref[name]
  cfg = make_cfg()
  mine = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()
          if isinstance(v, (bool, int, float, str, tuple))}
  assert set(r['plain']) - set(mine) == REFERENCE_ONLY
  assert set(mine) - set(r['plain']) == PACKAGE_ONLY
  for k in sorted(set(mine) & set(r['plain'])):
    assert mine[k] == r['plain'][k] and type(mine[k]) is type(r['plain'][k]), (k, mine[k], r['plain'][k])
  assert [f.__name__ for f in cfg.filters] == r['other']['filters']
  assert (cfg.adam_beta1, cfg.adam_beta2) == (r['adam']['beta1'], r['adam']['beta2'])
  for t, g, c in zip(ref['iterations'], r['lr_g'], r['lr_c']):
    assert cfg.lr_g(t) == pytest.approx(g, rel=1e-14) and cfg.lr_c(t) == pytest.approx(c, rel=1e-14), t
  # the values GAN.set_lrs derives (net.py:222-251, config_example.py:151-158)
  assert cfg.value_lr_mul * cfg.lr_g(0) == pytest.approx(10 * r['lr_g'][0])
  assert len(r['sha256']) == 64
