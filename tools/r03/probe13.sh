#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p13
rm -rf $OUT; mkdir -p $OUT
cd $R
cat > /tmp/fr.py <<'PY'
import os, pickle, time, torch, torch.distributed as dist
os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
from torch._C._distributed_c10d import _dump_nccl_trace
t = torch.ones(1024, device=dev)
for i in range(3):
  dist.all_reduce(t)
def show(tag):
  d = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
  es = d.get('entries', [])
  print(tag, 'keys', list(d.keys()), 'n', len(es), [(e.get('profiling_name'), e.get('state'), e.get('retired')) for e in es])
show('right after issue')
torch.cuda.synchronize(); show('after sync')
import sys; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from exposure_amd import dist as xdist
t0 = time.time(); ok = xdist.nccl_works_retired(); print('retired?', ok, 'waited %.3f s' % (time.time() - t0)); show('after wait')
dist.destroy_process_group()
PY
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python /tmp/fr.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee $OUT/fr.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
