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
# MIOpen also WRITES its user db (find mode appends rankings, perf-db updates), so the process is pointed at a private
# writable COPY of the shipped files, never at the package directory itself (tracked files stay untouched, a read-only
# install works, ranks do not append to one file): ``$EXPO_MIOPEN_DB_DIR`` or ``~/.cache/exposure_amd/miopen_db``, one
# sub-directory per local rank; a shipped file replaces its copy only when the copy is missing or older.  The user's own
# ``~/.config/miopen`` entries are not consulted while this is active -- set MIOPEN_USER_DB_PATH yourself to keep them.


def _private_miopen_db(shipped):
  import shutil
  import tempfile
  rank = _os.environ.get('LOCAL_RANK', '0')
  base = _os.environ.get('EXPO_MIOPEN_DB_DIR') or _os.path.join(
      _os.environ.get('XDG_CACHE_HOME') or _os.path.join(_os.path.expanduser('~'), '.cache'), 'exposure_amd', 'miopen_db')
  dst = _os.path.join(base, 'rank%s' % rank)
  try:
    _os.makedirs(dst, exist_ok=True)
    probe = _os.path.join(dst, '.writable.%d' % _os.getpid())
    open(probe, 'w').close()
    _os.remove(probe)
  except OSError:
    dst = tempfile.mkdtemp(prefix='exposure_amd_miopen_db_')
  for name in _os.listdir(shipped):
    src, out = _os.path.join(shipped, name), _os.path.join(dst, name)
    try:
      if not _os.path.exists(out) or _os.path.getmtime(out) < _os.path.getmtime(src):
        tmp = '%s.%d.tmp' % (out, _os.getpid())
        shutil.copyfile(src, tmp)
        _os.replace(tmp, out)  # atomic: another process of the same rank id never sees half a file
    except OSError:
      pass
  return dst


if _os.environ.get('EXPO_MIOPEN_DB', '1') != '0' and 'MIOPEN_USER_DB_PATH' not in _os.environ:
  _db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'miopen_db')
  if _os.path.isdir(_db):
    _os.environ['MIOPEN_USER_DB_PATH'] = _private_miopen_db(_db)
    _os.environ['EXPO_MIOPEN_DB_ACTIVE'] = _db  # (what bench.py reports: which shipped set the copy came from)
