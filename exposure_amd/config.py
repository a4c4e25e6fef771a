"""``cfg`` with the reference's field names and values (``config_example.py:8-176``).

Only plain data: the TF optimiser / data-provider lambdas of the reference config are replaced
by the numbers they encode (Adam betas, LR schedule constants).  ``cfg.filters`` holds the
filter *classes* in the reference order, so ``[F(net, cfg) for F in cfg.filters]``
(``agent.py:45``) works unchanged.
"""
from .util import Dict


def make_cfg(filters=None):
  from . import filters as F
  cfg = Dict()
  # -- filter parameters (config_example.py:22-39)
  cfg.filters = list(filters) if filters is not None else [
      F.ExposureFilter, F.GammaFilter, F.ImprovedWhiteBalanceFilter, F.SaturationPlusFilter, F.ToneFilter,
      F.ContrastFilter, F.WNBFilter, F.ColorFilter
  ]
  cfg.curve_steps = 8
  cfg.gamma_range = 3
  cfg.exposure_range = 3.5
  cfg.wb_range = 1.1
  cfg.color_curve_range = (0.90, 1.10)
  cfg.lab_curve_range = (0.90, 1.10)
  cfg.tone_curve_range = (0.5, 2)
  cfg.masking = False
  cfg.minimum_strength = 0.3
  cfg.maximum_sharpness = 1
  cfg.clamp = False
  # -- RL parameters (config_example.py:44-69)
  cfg.critic_logit_multiplier = 0.05
  cfg.discount_factor = 1.0
  cfg.filter_usage_penalty = 1.0
  cfg.use_TD = True
  cfg.test_random_walk = False
  cfg.replay_memory_size = 128
  cfg.maximum_trajectory_length = 7
  cfg.over_length_keep_prob = 0.5
  cfg.all_reward = 1.0
  cfg.img_include_states = True
  cfg.exploration = 0.05
  cfg.exploration_penalty = 0.05
  cfg.early_stop_penalty = 1.0
  # -- CNN parameters (config_example.py:74-85)
  cfg.source_img_size = 64
  cfg.base_channels = 32
  cfg.dropout_keep_prob = 0.5
  cfg.share_feed_dict = True
  cfg.shared_feature_extractor = True
  cfg.fc1_size = 128
  cfg.bnw = False
  cfg.feature_extractor_dims = 4096
  # -- GAN parameters (config_example.py:90-118)
  cfg.use_penalty = True
  cfg.gan = 'w'
  cfg.giters = 1
  cfg.gradient_penalty_lambda = 10
  cfg.citers = 5
  cfg.critic_initialization = 10
  cfg.clamp_critic = 0.01
  cfg.median_filter_size = 101
  cfg.z_type = 'uniform'
  cfg.z_dim_per_filter = 16
  cfg.num_state_dim = 3 + len(cfg.filters)
  cfg.z_dim = 3 + len(cfg.filters) * cfg.z_dim_per_filter
  cfg.test_steps = 5
  cfg.real_img_size = 64
  cfg.real_img_channels = 1 if cfg.bnw else 3
  # -- training (config_example.py:126-161)
  cfg.supervised = False
  cfg.batch_size = 64
  cfg.max_iter_step = 20000
  cfg.lr_decay = 0.1
  cfg.base_lr = 5e-5
  cfg.lr_segments = 3
  cfg.generator_lr_mul = 0.3
  cfg.parameter_lr_mul = 1
  cfg.value_lr_mul = 10
  cfg.critic_lr_mul = 1
  cfg.adam_beta1 = 0.5
  cfg.adam_beta2 = 0.9
  cfg.lr_g = lambda t: cfg.generator_lr_mul * cfg.base_lr * cfg.lr_decay**(1.0 * t * cfg.lr_segments / cfg.
                                                                         max_iter_step)
  cfg.lr_c = lambda t: cfg.critic_lr_mul * cfg.base_lr * cfg.lr_decay**(1.0 * t * cfg.lr_segments / cfg.
                                                                      max_iter_step)
  cfg.num_samples = 64
  cfg.img_channels = 1 if cfg.bnw else 3
  # -- build-specific switches (not in the reference)
  cfg.hsv_grad_mode = 0  # 0 = TF-1.x faithful (HSV ops not differentiable), 1 = analytic
  return cfg
