"""Random shapes through the 128-row-tile kernels (convr / convs / convu) against their 64-row forms (convg / convh /
convt): identical bits expected.   python tools/fuzz_rows128.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
S = _native.PAIR_SPLIT_F16
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
bad = 0
for case in range(n_cases):
    kind = rng.choice(["convs", "convr", "convu"])
    C = int(rng.choice([128, 256, 512]))
    B = int(rng.randint(1, 4))
    T = int(rng.choice([1, 7, 127, 128, 129, 255, 300, 513, 1000, 2049]))
    blocks = int(rng.choice([0, 0, 1, 5, 37]))
    outs = []
    if kind == "convs":
        dil = int(rng.choice([1, 3, 5, 9]))
        ks = [3] if dil == 9 else [int(k) for k in rng.choice([3, 7, 11], size=rng.randint(1, 4), replace=False)]
        xs = [t(rng.randn(B, C, T).astype(np.float32)) for _ in ks]
        P = [_native.pack_pair(t((rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)), S) for k in ks]
        bs = [t(rng.randn(C).astype(np.float32)) for _ in ks]
        R = [t(rng.randn(B, C, T).astype(np.float32)) for _ in ks]
        reflect = rng.rand() < 0.5 and (max(ks) - 1) // 2 * dil < T
        key = "convh_rows64"
        run = lambda: _native.conv1d_split_f16(xs, P, bs, ks, dil, pre_slope=0.1, res=R, act_slope=0.3,  # noqa: E731
                                               pad_mode=_native.PAD_REFLECT if reflect else _native.PAD_ZERO)
        desc = f"convs C={C} B={B} T={T} dil={dil} ks={ks} reflect={reflect} blocks={blocks}"
    elif kind == "convr":
        x, x2 = t(rng.randn(B, C, T).astype(np.float32)), t(rng.randn(B, C, T).astype(np.float32))
        P = _native.pack_conv1x1_2src_split(t((rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32)),
                                            t((rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32)))
        b = t(rng.randn(C).astype(np.float32))
        r = t(rng.randn(B, C, T).astype(np.float32)) if rng.rand() < 0.5 else None
        key = "convg_rows64"
        run = lambda: [_native.conv1x1_2src_split_f16(x, x2, P, b, pre_slope=0.2, res=r)]  # noqa: E731
        desc = f"convr C={C} B={B} T={T} res={r is not None} blocks={blocks}"
    else:
        s = int(rng.choice([2, 3, 4, 5, 6, 8, 10, 16]))
        cout = int(rng.choice([16, 20, 32, 64, 128, 256]))
        if cout * s < 64:
            cout = 64
        pad = int(rng.randint(0, s + 1))
        x = t(rng.randn(B, C, T).astype(np.float32))
        P = _native.pack_conv_transpose1d_split(t((rng.randn(C, cout, 2 * s) / np.sqrt(2 * C)).astype(np.float32)), s)
        b = t(rng.randn(cout).astype(np.float32))
        key = "convt_rows64"
        if (T - 1) * s - 2 * pad + 2 * s <= 0:
            continue
        run = lambda: [_native.conv_transpose1d_split_f16(x, P, b, cout, 2 * s, s, pad, 0, pre_slope=0.1)]  # noqa: E731
        desc = f"convu Cin={C} Cout={cout} B={B} T={T} s={s} pad={pad} blocks={blocks}"
    _native.tuning_set("convh_blocks", blocks)
    for rows64 in (1, 0):
        _native.tuning_set(key, rows64)
        outs.append([o.clone() for o in run()])
    _native.tuning_set(key, -1)
    _native.tuning_set("convh_blocks", 0)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b_) for a, b_ in zip(*outs)) and all(bool(torch.isfinite(a).all()) for a in outs[0])
    if not same:
        bad += 1
        print("MISMATCH", desc, flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
