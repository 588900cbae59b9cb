"""Does a HIP graph of the forward's 15 launches run faster on the GPU than the same launches enqueued one by one?
(batch 1, HiFi-GAN light, 1000 frames; torch.cuda.CUDAGraph capture of Plan.run on a side stream)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, 1)).to(dev)
T = mel.shape[2]


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    plan = model._trunk_plan(T)
    out = plan.run(mel)
    ref = out.clone()
    print(f"stream launches : {timeit(lambda: plan.run(mel, out=out)):.4f} ms per forward")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.run(mel, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            plan.run(mel, out=out)
    torch.cuda.synchronize()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("graph replay == stream launches:", torch.equal(out, ref))
    print(f"graph replay    : {timeit(g.replay):.4f} ms per forward")
    print(f"stream launches : {timeit(lambda: plan.run(mel, out=out)):.4f} ms per forward")
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    host = 1e3 * (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    print(f"graph replay host enqueue: {host:.4f} ms")
