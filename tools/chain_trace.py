"""Timeline of a chained launch (tuning aid; library built with FV_HIPCC_FLAGS=-DFV_PAIR_TRACE): HiFi-GAN light, batch 1;
per phase of the chain the blocks' durations (s_memtime ticks; the counters of different XCDs are not synchronised, so
every block is measured from its own start) and the time they spent waiting for flags; a least-squares fit of the phase
durations against the items the schedule gave the block.
    python tools/chain_trace.py [tuning "key=value,..."]"""
import os
import sys

import numpy as np
import torch

dev = torch.device("cuda:0")
NBLK, NPH = 256, 4
GHZ = 2.36      # s_memtime rate observed on MI355X (tools/clock_probe.hip)
trace = torch.zeros(NBLK * NPH * 8 + 64, dtype=torch.int64, device=dev)
os.environ["FV_TUNING"] = "1"
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    _native.tuning_set(k, int(v))
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, 1)).to(dev)
with torch.no_grad():
    for _ in range(4):
        model(mel)
    torch.cuda.synchronize()
    trace.zero_()
    torch.cuda.synchronize()
    model(mel)
    torch.cuda.synchronize()
tr = trace.cpu().numpy()[:NBLK * NPH * 8].reshape(NBLK, NPH, 8).astype(np.int64)
us = lambda x: x / (1e3 * GHZ)  # noqa: E731
base = tr[:, 0, 0]
end = tr[:, :, 1].max(axis=1) - base
print(f"block totals (own start -> own end): min {us(end.min()):.1f} median {us(np.median(end)):.1f} max {us(end.max()):.1f} us")
rows, dur = [], []
for ph in range(NPH):
    act = tr[:, ph, 1] > tr[:, ph, 0]
    if not act.any():
        continue
    s, e = tr[act, ph, 0] - base[act], tr[act, ph, 1] - base[act]
    d = e - s
    st = tr[act, ph, 2] / 8.0
    cnt = (tr[act, ph, 4:7] >> 20).astype(np.float64)
    print(f"phase {ph}: start {us(s.min()):.1f} / {us(np.median(s)):.1f} / {us(s.max()):.1f}  "
          f"end {us(e.min()):.1f} / {us(np.median(e)):.1f} / {us(e.max()):.1f}  duration {us(d.min()):.1f} / "
          f"{us(np.median(d)):.1f} / {us(d.max()):.1f}  waiting {us(st.min()):.1f} / {us(np.median(st)):.1f} / "
          f"{us(st.max()):.1f} us (min / median / max); items per block {cnt.sum(axis=1).mean():.2f}")
    for c, dd, ww in zip(cnt, d, st):
        rows.append(list(c) + [float((c > 0).sum())])
        dur.append(us(dd - ww))
A, y = np.array(rows), np.array(dur)
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print("fit of (phase duration - waiting): us per item of member 0 / 1 / 2, us per member run:", np.round(coef, 2),
      " rms residual %.2f us" % np.sqrt(np.mean((A @ coef - y) ** 2)))
w = (tr[:, :, 2] / 8.0).sum(axis=1)
print(f"waiting per block: median {us(np.median(w)):.1f} max {us(w.max()):.1f} us")
