"""r04p19: would MORE than two lanes help the headline chain?  Python-level prototype (as r02p20 for two lanes): the
batch of 64 images is cut into L slices, each slice's whole chain step (expo_chain_fwd + expo_chain_bwd with
EXPO_CHAIN_STREAMS=1, own workspace) goes to its own stream, ten steps are captured into one hipGraph.  L finish
launches per step instead of one, otherwise the library's kernels.  Usage (GPU box):
EXPO_CHAIN_STREAMS=1 python tools/r04/lanes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from exposure_amd import _cabi  # noqa: E402


def main():
  assert os.environ.get('EXPO_CHAIN_STREAMS') == '1', 'run with EXPO_CHAIN_STREAMS=1 (the lanes are made here)'
  dev = torch.device('cuda:0')
  shape = (64, 512, 512, 3)
  ids = list(range(8))
  x, dy, params = bench.make_device_case(shape, torch.float16, dev, 1234)
  acts = [x] + [torch.empty_like(x) for _ in range(8)]
  ga, gb = torch.empty_like(x), torch.empty_like(x)
  grads = [ga if (i % 2 == 0) else gb for i in range(8)] + [dy]
  dprm = [torch.empty_like(p) for p in params]
  for lanes in (1, 2, 3, 4, 2, 1):
    cuts = [round(i * shape[0] / lanes) for i in range(lanes + 1)]
    sl = [slice(cuts[i], cuts[i + 1]) for i in range(lanes)]
    ws = [_cabi.new_workspace(dev, _cabi.workspace_bytes(s.stop - s.start, 512, 512, _cabi.EXPO_F16, 8)) for s in sl]
    streams = [torch.cuda.Stream() for _ in range(lanes - 1)]

    def step():
      cur = torch.cuda.current_stream()
      for st in streams:
        st.wait_stream(cur)
      for li, s in enumerate(sl):
        ctx = torch.cuda.stream(streams[li - 1]) if li > 0 else torch.cuda.stream(cur)
        with ctx:
          a = [t[s] for t in acts]
          g = [t[s] for t in grads]
          p = [t[s] for t in params]
          d = [t[s] for t in dprm]
          _cabi.chain_fwd(ids, a, p)
          _cabi.chain_bwd(ids, a, g, p, d, workspace=ws[li])
      for st in streams:
        cur.wait_stream(st)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      for _ in range(10):
        step()
    for _ in range(3):
      graph.replay()
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(4):
        graph.replay()
      e1.record()
      e1.synchronize()
      runs.append(e0.elapsed_time(e1) / 40)
    print('lanes %d: %s ms per chain step (median %.4f)' % (lanes, ' '.join('%.4f' % r for r in runs), sorted(runs)[2]))
    del graph


if __name__ == '__main__':
  main()
