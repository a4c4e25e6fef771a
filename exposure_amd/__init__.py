"""exposure_amd -- Exposure's differentiable filter stack on MI355X (gfx950): HIP kernels behind a C-ABI
(include/exposure_hip.h) under the reference's Filter / agent / critic Python surface.  DESIGN.md has the map."""
import os as _os

# MIOpen kernel selection for the convnets around the filter path (agent.py:11-37, critics.py:6-38).  MIOpen's
# immediate mode picks convolution solvers by heuristic unless its user find-db already holds a measured ranking for the
# problem; on a fresh box the heuristic's picks for these 4x4 / stride-2 NHWC fp32 problems cost 1.8 ms of a 12 ms training
# iteration (11.95 vs 10.17 ms, profiles/r04_experiments.md r04p13).  exposure_amd/miopen_db/ is such a find-db -- 28
# rankings measured once on an MI355X with `bench.py --workload train --miopen-find on`, plain text, data only -- for the
# batch-64 / batch-128 problems of the training step.  Pointing MIOpen at it gives find-mode selection WITHOUT running
# find mode (whose trial kernels once faulted inside the gpu test suite, r03p6-12).  Other problem sizes fall back to the
# heuristic as before.  MIOPEN_USER_DB_PATH set by the user wins; EXPO_MIOPEN_DB=0 switches this off.
if _os.environ.get('EXPO_MIOPEN_DB', '1') != '0':
  _db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'miopen_db')
  if _os.path.isdir(_db):
    _os.environ.setdefault('MIOPEN_USER_DB_PATH', _db)
