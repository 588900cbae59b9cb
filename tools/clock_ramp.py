"""ms per forward as a function of time since the device left idle (HiFi-GAN light, T = 1000, batch 1): one event every
`every` forwards over `total` forwards, after `idle` seconds of host sleep.  What bench.py's PREWARM_S is sized by.
    python tools/clock_ramp.py [total=4000] [every=25] [idle=1.0]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    idle = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    dev = torch.device("cuda", 0)
    model, cfg, sd = bench.build_model("light", dev, None, 0)
    mel = torch.from_numpy(bench.utterance_mels(0, 1)).to(dev)
    with torch.no_grad():
        model(mel)
    torch.cuda.synchronize()
    for rnd in range(2):
        time.sleep(idle)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(total // every + 1)]
        ev[0].record()
        with torch.no_grad():
            for i in range(total):
                model(mel)
                if (i + 1) % every == 0:
                    ev[(i + 1) // every].record()
        torch.cuda.synchronize()
        assert not model.check_range()
        ms = [ev[i].elapsed_time(ev[i + 1]) / every for i in range(len(ev) - 1)]
        t = 0.0
        print(f"round {rnd}: after {idle} s idle; columns: ms since start, ms per forward over the next {every}")
        line = []
        for m in ms:
            line.append(f"{t:7.0f}:{m:.4f}")
            t += m * every
            if len(line) == 8:
                print("  " + "  ".join(line))
                line = []
        if line:
            print("  " + "  ".join(line))


if __name__ == "__main__":
    main()
