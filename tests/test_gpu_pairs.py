"""GPU parity of the fused ResBlock-pair kernels (csrc/pair_kernels.hpp) through the C ABI
(fv_resblock1_fused, fv_mrf_stage) against the C oracle's conv1d on the same seeded inputs.

Tolerance: 2e-5 relative to the tensor's scale (the generator-level bound is the north star's 1e-4).
"""
import numpy as np
import pytest
import torch

from fastvocoder_amd import _native
from oracle import ops as oo

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


def _pair_ref(x, w1, b1, w2, b2, dil, slope):
    """x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 (reference modules.py:223-230, one loop iteration)."""
    k = w1.shape[2]
    mid = oo.conv1d(x, w1, b1, dil=dil, pad=(k - 1) * dil // 2, pre_slope=slope)
    return oo.conv1d(mid, w2, b2, dil=1, pad=(k - 1) // 2, pre_slope=slope) + x


def _member(rng, B, C, T, k, bias=True):
    x = rng.randn(B, C, T).astype(np.float32)
    w1 = (rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)
    w2 = (rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)
    b1 = rng.randn(C).astype(np.float32) if bias else None
    b2 = rng.randn(C).astype(np.float32) if bias else None
    return x, w1, b1, w2, b2


def _t(a):
    return None if a is None else torch.from_numpy(a).to(_dev())


PAIR_CASES = [
    # B, C, T, dil, taps of the members, bias
    (1, 16, 48, 1, (3,), True),            # one tile, both sequence ends inside it
    (2, 16, 100, 3, (7,), True),
    (1, 16, 1000, 5, (11,), True),         # several 244-column tiles
    (2, 16, 740, 5, (11, 7, 3), True),     # the MRF trio in one launch, ragged last tiles
    (1, 16, 2000, 1, (3, 7, 11), False),   # member order free, no bias
    (1, 32, 64, 1, (11,), True),
    (2, 32, 500, 3, (11, 7, 3), True),     # 116 / 120 / 124-column tiles
    (1, 32, 1204, 5, (7, 3, 11), True),
    (3, 32, 236, 5, (3,), False),
    (1, 16, 4, 5, (11,), True),            # shorter than every halo
]


@pytest.mark.parametrize("case", PAIR_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_resblock_pair_vs_oracle(case):
    B, C, T, dil, ks, bias = case
    rng = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    ms = [_member(rng, B, C, T, k, bias) for k in ks]
    refs = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
    xs = [_t(m[0]) for m in ms]
    w1s = [_native.pack_pair(_t(m[1])) for m in ms]
    w2s = [_native.pack_pair(_t(m[3])) for m in ms]
    ys = _native.resblock1_fused(xs, w1s, w2s, [_t(m[2]) for m in ms], [_t(m[4]) for m in ms], list(ks), dil, 0.1)
    for y, ref in zip(ys, refs):
        assert _rel(y, ref) <= 2e-5
    # activated twin next to the raw output, and the in-place form
    acts = [torch.empty_like(x) for x in xs]
    ys2 = _native.resblock1_fused(xs, w1s, w2s, [_t(m[2]) for m in ms], [_t(m[4]) for m in ms], list(ks), dil, 0.1,
                                  act_slope=0.2, outs_act=acts)
    for y, a, ref in zip(ys2, acts, refs):
        assert _rel(y, ref) <= 2e-5
        assert _rel(a, oo.lrelu(ref, 0.2)) <= 2e-5
    ys3 = _native.resblock1_fused(xs, w1s, w2s, [_t(m[2]) for m in ms], [_t(m[4]) for m in ms], list(ks), dil, 0.1,
                                  act_slope=0.2)
    for y, ref in zip(ys3, refs):
        assert _rel(y, oo.lrelu(ref, 0.2)) <= 2e-5


@pytest.mark.parametrize("case", [(1, 244, 5), (2, 1000, 5), (1, 100, 1), (3, 488, 3), (1, 8, 5), (1, 2444, 5)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_mrf_stage_vs_oracle(case):
    B, T, dil = case
    C = 16
    rng = np.random.RandomState(1000 * T + dil)
    ks = (3, 11, 7)                                       # any member order
    ms = [_member(rng, B, C, T, k, True) for k in ks]
    total = sum(_pair_ref(x, w1, b1, w2, b2, dil, 0.1).astype(np.float64) for x, w1, b1, w2, b2 in ms)
    ref = (total / 3.0).astype(np.float32)
    xs = [_t(m[0]) for m in ms]
    w1s = [_native.pack_pair(_t(m[1])) for m in ms]
    w2s = [_native.pack_pair(_t(m[3])) for m in ms]
    b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
    y = _native.mrf_stage(xs, w1s, w2s, b1s, b2s, list(ks), dil, 0.1, out_div=3.0)
    assert _rel(y, ref) <= 2e-5
    act = torch.empty_like(y)
    y2 = _native.mrf_stage(xs, w1s, w2s, b1s, b2s, list(ks), dil, 0.1, out_div=3.0, act_slope=0.01, out_act=act)
    assert _rel(y2, ref) <= 2e-5 and _rel(act, oo.lrelu(ref, 0.01)) <= 2e-5
    y3 = _native.mrf_stage(xs, w1s, w2s, b1s, b2s, list(ks), dil, 0.1, out_div=3.0, post=_native.POST_TANH)
    assert _rel(y3, np.tanh(ref.astype(np.float64))) <= 2e-5


SPLIT = _native.PAIR_SPLIT_F16


@pytest.mark.parametrize("case", PAIR_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_resblock_pair_split_f16_vs_oracle(case):
    """The split-f16 kernel (csrc/pairh_kernels.hpp) against the double-accumulating oracle: per layer it must be
    as close as the fp32-MFMA kernel is (within 3x of its error + 1e-7), and inside 4e-6 of the tensor's scale."""
    B, C, T, dil, ks, bias = case
    rng = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    ms = [_member(rng, B, C, T, k, bias) for k in ks]
    refs = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
    xs = [_t(m[0]) for m in ms]
    b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
    f1, f2 = [_native.pack_pair(_t(m[1])) for m in ms], [_native.pack_pair(_t(m[3])) for m in ms]
    h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
    y32 = _native.resblock1_fused(xs, f1, f2, b1s, b2s, list(ks), dil, 0.1)
    ys = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
    for y, yf, ref in zip(ys, y32, refs):
        e, ef = _rel(y, ref), _rel(yf, ref)
        assert e <= 4e-6 and e <= 3 * ef + 1e-7, (e, ef)
    acts = [torch.empty_like(x) for x in xs]
    ys2 = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, act_slope=0.2, outs_act=acts, prec=SPLIT)
    for y, a, ref in zip(ys2, acts, refs):
        assert _rel(y, ref) <= 4e-6 and _rel(a, oo.lrelu(ref, 0.2)) <= 4e-6
    ys3 = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, act_slope=0.2, prec=SPLIT)
    for y, ref in zip(ys3, refs):
        assert _rel(y, oo.lrelu(ref, 0.2)) <= 4e-6


@pytest.mark.parametrize("case", [(1, 16, 244, 5), (2, 16, 1000, 5), (1, 16, 100, 1), (3, 16, 488, 3), (1, 16, 8, 5),
                                  (1, 16, 2444, 5), (2, 32, 500, 5), (1, 32, 1204, 3), (1, 32, 12, 1)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_split_f16_stage_end_vs_oracle(case):
    """End of an MRF stage on the split-f16 kernels: the 7- and 11-tap pairs in one launch (r1, r2), then the
    3-tap pair with ((r0 + r1) + r2) / 3 in its epilogue -- the reference's association (hifigan.py:99-103)."""
    B, C, T, dil = case
    rng = np.random.RandomState(2000 * T + dil + C)
    ks = (3, 7, 11)
    ms = [_member(rng, B, C, T, k, True) for k in ks]
    r = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
    ref = ((r[0] + r[1]) + r[2]) / np.float32(3.0)
    xs = [_t(m[0]) for m in ms]
    h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
    b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
    r12 = _native.resblock1_fused(xs[1:], h1[1:], h2[1:], b1s[1:], b2s[1:], [7, 11], dil, 0.1, prec=SPLIT)
    args = ([xs[0]], [h1[0]], [h2[0]], [b1s[0]], [b2s[0]], [3], dil, 0.1)
    y, = _native.resblock1_fused(*args, prec=SPLIT, add1=[r12[0]], add2=[r12[1]], out_div=3.0)
    assert _rel(y, ref) <= 4e-6
    act = [torch.empty_like(y)]
    y2, = _native.resblock1_fused(*args, prec=SPLIT, add1=[r12[0]], add2=[r12[1]], out_div=3.0, act_slope=0.01,
                                  outs_act=act)
    assert _rel(y2, ref) <= 4e-6 and _rel(act[0], oo.lrelu(ref, 0.01)) <= 4e-6
    y3, = _native.resblock1_fused(*args, prec=SPLIT, add1=[r12[0]], add2=[r12[1]], out_div=3.0, post=_native.POST_TANH)
    assert _rel(y3, np.tanh(ref.astype(np.float64))) <= 4e-6
    y4, = _native.resblock1_fused(*args, prec=SPLIT, add1=[r12[0]], out_div=2.0)        # one addend only
    assert _rel(y4, (r[0] + r[1]) / np.float32(2.0)) <= 4e-6


def test_split_f16_results_do_not_depend_on_the_batch():
    rng = np.random.RandomState(11)
    ks = (11, 7, 3)
    for C, T in ((16, 1500), (32, 700)):
        ms = [_member(rng, 3, C, T, k, True) for k in ks]
        xs = [_t(m[0]) for m in ms]
        h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
        b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
        full = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), 3, 0.1, prec=SPLIT)
        for b in range(3):
            one = _native.resblock1_fused([x[b:b + 1].contiguous() for x in xs], h1, h2, b1s, b2s, list(ks), 3, 0.1,
                                          prec=SPLIT)
            for yf, yo in zip(full, one):
                assert torch.equal(yf[b:b + 1], yo)


def test_split_f16_rejects_what_it_is_not_built_for():
    dev = _dev()
    with pytest.raises(_native.NativeError):
        _native.pack_pair(torch.zeros((48, 48, 3), device=dev), SPLIT)
    x = torch.zeros((1, 16, 50), device=dev)               # T % 4 != 0
    w = _native.pack_pair(torch.zeros((16, 16, 3), device=dev), SPLIT)
    with pytest.raises(_native.NativeError, match="multiple of 4"):
        _native.resblock1_fused([x], [w], [w], [None], [None], [3], 1, 0.1, prec=SPLIT)
    x16 = torch.zeros((1, 16, 64), device=dev)
    w16 = _native.pack_pair(torch.zeros((16, 16, 3), device=dev))
    with pytest.raises(_native.NativeError, match="SPLIT_F16 only"):
        _native.resblock1_fused([x16], [w16], [w16], [None], [None], [3], 1, 0.1, add1=[x16.clone()])


WIDE_CASES = [
    # B, C, T, dil, taps, bias -- ResBlock pairs of the 64- / 128-channel stages: two launches of the split-f16 conv
    # kernel with streamed weights (csrc/convh_kernels.hpp)
    (1, 64, 300, 1, (3,), True),             # two 256-column tiles, ragged
    (2, 64, 700, 5, (11, 7, 3), True),
    (1, 64, 1030, 3, (7, 11), False),
    (1, 64, 7, 5, (11,), True),              # shorter than every halo, T % 4 != 0
    (1, 128, 200, 3, (7,), True),            # two row tiles share a column tile's image
    (1, 128, 520, 5, (11, 7, 3), True),
    (2, 128, 130, 1, (3, 11), True),
    (1, 256, 150, 3, (7, 3), True),          # two chunks of 128 input channels, four row tiles
    (1, 512, 70, 1, (3,), True),             # four chunks, eight row tiles
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_wide_resblock_pair_split_f16_vs_oracle(case):
    B, C, T, dil, ks, bias = case
    rng = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    ms = [_member(rng, B, C, T, k, bias) for k in ks]
    refs = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
    xs = [_t(m[0]) for m in ms]
    b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
    h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
    ys = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
    for y, ref in zip(ys, refs):
        assert _rel(y, ref) <= 4e-6
    acts = [torch.empty_like(x) for x in xs]
    ys2 = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, act_slope=0.2, outs_act=acts, prec=SPLIT)
    for y, a, ref in zip(ys2, acts, refs):
        assert _rel(y, ref) <= 4e-6 and _rel(a, oo.lrelu(ref, 0.2)) <= 4e-6
    if len(ks) == 3:                          # the stage end: members 1, 2 first, member 0 carries the merge
        r12 = _native.resblock1_fused(xs[1:], h1[1:], h2[1:], b1s[1:], b2s[1:], list(ks[1:]), dil, 0.1, prec=SPLIT)
        y, = _native.resblock1_fused([xs[0]], [h1[0]], [h2[0]], [b1s[0]], [b2s[0]], [ks[0]], dil, 0.1, prec=SPLIT,
                                     add1=[r12[0]], add2=[r12[1]], out_div=3.0, act_slope=0.1)
        ref = oo.lrelu(((refs[0] + refs[1]) + refs[2]) / np.float32(3.0), 0.1)
        assert _rel(y, ref) <= 4e-6
        # bit-identity of an utterance alone and inside the batch
        if B > 1:
            one = _native.resblock1_fused([x[1:2].contiguous() for x in xs], h1, h2, b1s, b2s, list(ks), dil, 0.1,
                                          prec=SPLIT)
            for yf, yo in zip(ys, one):
                assert torch.equal(yf[1:2], yo)


@pytest.mark.parametrize("case", [(2, 64, 300, 3, (7, 3)), (1, 128, 260, 5, (11, 7, 3)), (1, 64, 5, 1, (3,)),
                                  (3, 128, 129, 1, (11,)), (1, 256, 300, 5, (11, 3)), (2, 512, 40, 3, (7,))],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv1d_split_f16_vs_oracle(case):
    """fv_conv1d_split_f16: every epilogue form against the oracle's conv1d."""
    B, C, T, dil, ks = case
    rng = np.random.RandomState(31 * T + C + dil)
    n = len(ks)
    xs = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    ws = [(rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    bs = [rng.randn(C).astype(np.float32) for _ in ks]
    rs = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    a1 = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    a2 = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    conv = [oo.conv1d(x, w, b, dil=dil, pad=(k - 1) * dil // 2, pre_slope=0.1) for x, w, b, k in zip(xs, ws, bs, ks)]
    X, P, Bi = [_t(x) for x in xs], [_native.pack_pair(_t(w), SPLIT) for w in ws], [_t(b) for b in bs]
    R, A1, A2 = [_t(r) for r in rs], [_t(a) for a in a1], [_t(a) for a in a2]
    ys = _native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.1)
    for y, c in zip(ys, conv):
        assert _rel(y, c) <= 4e-6
    ys = _native.conv1d_split_f16(X, P, [None] * n, list(ks), dil, pre_slope=0.1, res=R)
    for y, c, b, r in zip(ys, conv, bs, rs):
        assert _rel(y, c - b[None, :, None] + r) <= 4e-6
    acts = [torch.empty_like(x) for x in X]
    ys = _native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.1, res=R, add1=A1, add2=A2, out_div=3.0,
                                  act_slope=0.01, outs_act=acts)
    for y, a, c, r, p1, p2 in zip(ys, acts, conv, rs, a1, a2):
        ref = (((c + r) + p1) + p2) / np.float32(3.0)
        assert _rel(y, ref) <= 4e-6 and _rel(a, oo.lrelu(ref, 0.01)) <= 4e-6
    ys = _native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.1, add1=A1, out_div=2.0, post=_native.POST_TANH)
    for y, c, p1 in zip(ys, conv, a1):
        assert _rel(y, np.tanh(((c + p1) / np.float32(2.0)).astype(np.float64))) <= 4e-6
    with pytest.raises(_native.NativeError, match="64, 128, 256 or 512"):
        _native.conv1d_split_f16([torch.zeros((1, 32, 16), device=_dev())], P[:1], [None], [ks[0]], dil)


@pytest.mark.parametrize("case", [(1, 128, 260, 5, (11, 7, 3)), (3, 128, 129, 1, (11,)), (1, 256, 300, 5, (11, 3)), (2, 512, 140, 3, (7,)),
                                  (2, 256, 1000, 9, (3,)), (1, 512, 257, 1, (3, 7, 11)), (1, 128, 1, 3, (3,)),
                                  (2, 256, 1000, 5, (11, 7, 3)), (1, 512, 700, 3, (3, 11)), (3, 256, 64, 1, (7,))],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv1d_split_f16_on_128_row_tiles_gives_the_same_bits(case, tuning):
    """The split-f16 convs at 128 channels and more on 128-row tiles (csrc/convr_kernels.hpp convs_kernel: a chunk's window
    is converted once for both 64-row tiles; at 256 / 512 channels without reflection padding the ring-free convs2_kernel of
    csrc/convs2_kernels.hpp, and convs_kernel under `convs_ringfree = 0`) against the 64-row tiles of convh_kernel: the same K
    order per output, so identical bits -- zero and reflection padding, every epilogue form, few persistent blocks, and the
    oracle."""
    B, C, T, dil, ks = case
    rng = np.random.RandomState(7 * T + C + dil)
    n = len(ks)
    xs = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    ws = [(rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    X, P = [_t(x) for x in xs], [_native.pack_pair(_t(w), SPLIT) for w in ws]
    Bi = [_t(rng.randn(C).astype(np.float32)) for _ in ks]
    R, A1, A2 = [[_t(rng.randn(B, C, T).astype(np.float32)) for _ in ks] for _ in range(3)]
    reflect = (max(ks) - 1) // 2 * dil < T

    def forms():
        out = [_native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.1)]
        acts = [torch.empty_like(x) for x in X]
        out.append(_native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.1, res=R, add1=A1, add2=A2, out_div=3.0,
                                            act_slope=0.01, outs_act=acts))
        out.append(acts)
        out.append(_native.conv1d_split_f16(X, P, [None] * n, list(ks), dil, pre_slope=0.1, add1=A1, out_div=2.0,
                                            post=_native.POST_TANH))
        if reflect:
            out.append(_native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.2, pad_mode=_native.PAD_REFLECT,
                                                act_slope=0.2))
        return [t.clone() for ts in out for t in ts]

    tuning("convh_rows64", 1)
    narrow = forms()
    tuning("convh_rows64", 0)
    wide = forms()
    tuning("convs_ringfree", 0)
    ring = forms()
    tuning("convs_ringfree", 1)
    rf64 = forms()
    tuning("convs_ringfree", 2)
    rf128 = forms()
    tuning("convs_ringfree", -1)
    tuning("convh_blocks", 3)
    few = forms()
    tuning("convh_blocks", 0)
    assert all(torch.equal(a, b) for a, b in zip(narrow, wide)) and all(torch.equal(a, b) for a, b in zip(wide, few))
    assert all(torch.equal(a, b) for a, b in zip(wide, ring))
    assert all(torch.equal(a, b) for a, b in zip(wide, rf64)) and all(torch.equal(a, b) for a, b in zip(wide, rf128))
    for y, x, w, b, k in zip(wide[:n], xs, ws, Bi, ks):
        assert _rel(y, oo.conv1d(x, w, b.cpu().numpy(), dil=dil, pad=(k - 1) * dil // 2, pre_slope=0.1)) <= 4e-6


@pytest.mark.parametrize("case", [(2, 64, 300, 9, (3,)), (1, 128, 130, 9, (3,)), (2, 256, 257, 3, (3,)), (1, 64, 12, 1, (3, 7)),
                                  (1, 512, 50, 9, (3,)), (3, 128, 10, 9, (3,)), (1, 64, 128, 5, (11, 3))],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv1d_split_f16_reflection_padding_vs_oracle(case):
    """MelGAN's ResidualStack conv (reference modules.py:351-359): ReflectionPad1d((k-1)/2 dil) in front of the conv,
    as mirrored rows of the window loader; dilation 9 exists with 3 taps."""
    B, C, T, dil, ks = case
    rng = np.random.RandomState(17 * T + C + dil)
    xs = [rng.randn(B, C, T).astype(np.float32) for _ in ks]
    ws = [(rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    bs = [rng.randn(C).astype(np.float32) for _ in ks]
    conv = [oo.conv1d(x, w, b, dil=dil, pad=(k - 1) * dil // 2, pad_mode=oo.PAD_REFLECT, pre_slope=0.2)
            for x, w, b, k in zip(xs, ws, bs, ks)]
    X, P, Bi = [_t(x) for x in xs], [_native.pack_pair(_t(w), SPLIT) for w in ws], [_t(b) for b in bs]
    ys = _native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.2, pad_mode=_native.PAD_REFLECT)
    for y, c in zip(ys, conv):
        assert _rel(y, c) <= 4e-6
    # the hidden tensor stored activated for the 1x1 behind it (act_slope without a twin)
    ys = _native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.2, pad_mode=_native.PAD_REFLECT, act_slope=0.2)
    for y, c in zip(ys, conv):
        assert _rel(y, oo.lrelu(c, 0.2)) <= 4e-6
    # zero padding with dilation 9 too
    zc = [oo.conv1d(x, w, b, dil=dil, pad=(k - 1) * dil // 2, pre_slope=0.2) for x, w, b, k in zip(xs, ws, bs, ks)]
    for y, c in zip(_native.conv1d_split_f16(X, P, Bi, list(ks), dil, pre_slope=0.2), zc):
        assert _rel(y, c) <= 4e-6


def test_conv1d_split_f16_reflection_rejects():
    x = [torch.zeros((1, 64, 9), device=_dev())]
    w = [_native.pack_pair(torch.zeros((64, 64, 3), device=_dev()), SPLIT)]
    with pytest.raises(_native.NativeError, match="reflection padding 9 needs more than 9"):
        _native.conv1d_split_f16(x, w, [None], [3], 9, pad_mode=_native.PAD_REFLECT)
    w7 = [_native.pack_pair(torch.zeros((64, 64, 7), device=_dev()), SPLIT)]
    with pytest.raises(_native.NativeError, match="dilation 9"):
        _native.conv1d_split_f16(x, w7, [None], [7], 9)
    with pytest.raises(_native.NativeError, match="pad_mode"):
        _native.conv1d_split_f16(x, w, [None], [3], 1, pad_mode=_native.PAD_CAUSAL)


@pytest.fixture
def tuning():
    """fv_tuning_set for the duration of a test (process-wide switches of the launchers: restored afterwards)."""
    defaults = {"sched": 1, "sched_switch": 4, "convh_blocks": 0, "pair_blocks": 0, "pair128_unfused": 0, "convg_rows64": -1,
                "convh_rows64": -1, "convt_rows64": -1, "convp_wide": 20, "convq_wide": 20, "convt_lean": 50, "convs_ringfree": -1, "convu_resident": 1, "convp_pp": 0}
    yield _native.tuning_set
    for k, v in defaults.items():
        _native.tuning_set(k, v)


def test_block_schedule_gives_the_same_bits(tuning):
    """Few, unequal items per block (batch 1): the host's longest-processing-time-first block schedule
    (csrc/convh_launch.hip pair_schedule, handed to the kernel inside its arguments) against the kernels' own
    contiguous partition (sched = 0) -- every item is computed exactly once either way, so the results are
    bit-identical; also with a switch cost that makes blocks take up two and three members."""
    rng = np.random.RandomState(78)
    tuning("sched", 2)                             # every launch with few items per block is scheduled
    for C, T, dil, ks in ((128, 8000, 3, (11, 7, 3)), (64, 9000, 5, (11, 7, 3)), (256, 700, 1, (7, 11, 3)), (128, 5000, 5, (11, 7))):
        ms = [_member(rng, 1, C, T, k, True) for k in ks]
        xs = [_t(m[0]) for m in ms]
        h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
        b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
        sched = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
        tuning("sched_switch", 0)
        sched0 = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
        tuning("sched_switch", 4)
        tuning("sched", 0)
        plain = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
        tuning("sched", 2)
        for a, b, c in zip(sched, sched0, plain):
            assert torch.equal(a, c) and torch.equal(b, c)
        ref = _pair_ref(*ms[0], dil, 0.1)
        assert _rel(sched[0], ref) <= 4e-6


@pytest.mark.parametrize("case", [(1, 1000, 3, 5, True), (2, 484, 3, 5, True), (1, 244, 11, 1, False), (3, 8, 7, 3, True),
                                  (1, 2468, 3, 1, False)], ids=lambda c: "x".join(str(v) for v in c))
def test_pair_with_folded_output_conv_vs_oracle(case):
    """fv_plan_set_pair_output_conv (HiFi-GAN's last launch, hifigan.py:97-106): the carrier pair with the MRF merge,
    lrelu(0.01), conv_post (16 -> 1, 7 taps) and tanh in ONE launch, against the oracle's pair + merge + conv1d."""
    B, T, k, dil, merge = case
    rng = np.random.RandomState(13 * T + k)
    C = 16
    x, w1, b1, w2, b2 = _member(rng, B, C, T, k, True)
    r1, r2 = rng.randn(B, C, T).astype(np.float32), rng.randn(B, C, T).astype(np.float32)
    wp = (rng.randn(1, C, 7) / np.sqrt(C * 7)).astype(np.float32)
    bp = rng.randn(1).astype(np.float32)
    pair = _pair_ref(x, w1, b1, w2, b2, dil, 0.1)
    merged = ((pair + r1) + r2) / np.float32(3.0) if merge else pair
    ref = np.tanh(oo.conv1d(merged, wp, bp, pad=3, pre_slope=0.01).astype(np.float64))
    # x, r1, r2 enter the plan as ONE [B, 3C, T] input; 1-tap convs with selection weights (exact copies) slice it
    X = _t(np.concatenate([x, r1, r2], axis=1))
    sel = [torch.zeros((C, 3 * C, 1), device=_dev()) for _ in range(3)]
    for j in range(3):
        sel[j][:, j * C:(j + 1) * C, 0] = torch.eye(C, device=_dev())
    plan3 = _native.Plan(3 * C)
    for j in range(3):
        plan3.add_conv1d(_native.SLOT_IN, 2 + j, _native.pack_conv1d(sel[j]), None, 3 * C, C, 1)
    plan3.add_resblock_pair(2, 5, _native.pack_pair(_t(w1), SPLIT), _native.pack_pair(_t(w2), SPLIT), _t(b1), _t(b2), C, k, dil,
                            0.1, prec=SPLIT, add1=3 if merge else _native.SLOT_NONE, add2=4 if merge else _native.SLOT_NONE,
                            out_div=3.0 if merge else 1.0)
    plan3.set_pair_output_conv(_t(wp.reshape(C, 7)), _t(bp), _native.SLOT_OUT, 0.01, _native.POST_TANH)
    y = plan3.run(X)
    assert tuple(y.shape) == (B, 1, T) and _rel(y, ref) <= 4e-6


CONVT_CASES = [  # B, Cin, Cout, Tin, stride, pad, out_pad
    (1, 256, 128, 300, 8, 4, 0), (2, 128, 64, 257, 5, 3, 1), (1, 128, 32, 129, 2, 1, 0), (1, 512, 256, 40, 8, 4, 0),
    (3, 128, 64, 1, 10, 5, 0), (2, 256, 64, 127, 6, 3, 0), (1, 128, 64, 128, 3, 2, 1), (1, 256, 256, 200, 4, 2, 0),
    (2, 128, 16, 50, 4, 0, 0), (1, 128, 64, 33, 8, 8, -8), (1, 128, 64, 260, 16, 8, 0),
    (4, 256, 256, 500, 4, 2, 0), (2, 128, 64, 700, 8, 4, 0),
    # 64 input channels (half a chunk) and row counts that are not a multiple of the 64-row tile
    (1, 64, 32, 300, 3, 2, 1), (2, 64, 32, 129, 2, 1, 0), (1, 64, 40, 100, 5, 3, 1), (1, 128, 20, 64, 4, 2, 0),
    # 32 input channels (a quarter of a 128-chunk's image: HiFi-GAN light's last upsampler, 32 -> 16 x 2) and 32 rows
    (1, 32, 16, 1000, 2, 1, 0), (2, 32, 16, 131, 2, 1, 0), (1, 32, 24, 77, 4, 2, 0),
]


@pytest.mark.parametrize("case", CONVT_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_transpose1d_split_f16_vs_oracle(case, tuning):
    """fv_conv_transpose1d_split_f16 (kernel = 2 strides; reference hifigan.py:45-46, melgan.py:37-39) against the
    oracle's transposed conv: every stride the shipped configs use, ragged ends, trimmed tail, activated twin."""
    B, cin, cout, T, s, pad, op = case
    k = 2 * s
    rng = np.random.RandomState(cin + 7 * T + s)
    x = rng.randn(B, cin, T).astype(np.float32)
    w = (rng.randn(cin, cout, k) / np.sqrt(cin * 2)).astype(np.float32)
    b = rng.randn(cout).astype(np.float32)
    ref = oo.conv_transpose1d(x, w, b, s, pad, max(op, 0), pre_slope=0.1)
    if op < 0:
        ref = ref[:, :, :ref.shape[2] + op]
    X, Bi = _t(x), _t(b)
    P = _native.pack_conv_transpose1d_split(_t(w), s)
    y = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1)
    assert tuple(y.shape) == ref.shape and _rel(y, ref) <= 4e-6
    twin = torch.empty_like(y)
    y2 = _native.conv_transpose1d_split_f16(X, P, None, cout, k, s, pad, op, pre_slope=0.1, out_act=twin, act_slope=0.2)
    assert _rel(y2, ref - b[None, :, None]) <= 4e-6 and _rel(twin, oo.lrelu(ref - b[None, :, None], 0.2)) <= 4e-6
    y3 = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1, act_slope=0.2)
    assert _rel(y3, oo.lrelu(ref, 0.2)) <= 4e-6
    # a few persistent blocks walking many tiles (and row tiles of one column tile): same bits
    tuning("convh_blocks", 3)
    few = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1)
    tuning("convh_blocks", 0)
    assert torch.equal(few, y)
    # these launches have few items per CU: they ran on the lean kernel (csrc/convtl_kernels.hpp: 64-column tiles, A operands
    # L2 -> registers).  The ring pipeline (convt_kernel: 128-column tiles, weights through LDS) gives the same bits -- and
    # is what the rest of this test forces
    tuning("convt_lean", 0)
    ring = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1)
    tw_r = torch.empty_like(y)
    ring2 = _native.conv_transpose1d_split_f16(X, P, None, cout, k, s, pad, op, pre_slope=0.1, out_act=tw_r, act_slope=0.2)
    assert torch.equal(ring, y) and torch.equal(ring2, y2) and torch.equal(tw_r, twin)
    tuning("convh_blocks", 3)
    assert torch.equal(_native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1), y)
    tuning("convh_blocks", 0)
    if B > 1:                                      # an utterance alone and inside the batch: same bits
        one = _native.conv_transpose1d_split_f16(X[1:2].contiguous(), P, Bi, cout, k, s, pad, op, pre_slope=0.1)
        assert torch.equal(one, y[1:2])
    if cin in (128, 256) and (cout * s + 63) // 64 * 64 <= 1024:
        # launches with many column tiles run a column tile with ALL its rows on resident images (csrc/convu2_kernels.hpp: A
        # operands L2 -> registers, the window converted once per column tile); forced here: same bits
        tuning("convu_resident", 2)
        tw_u = torch.empty_like(y)
        u1 = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1)
        u2 = _native.conv_transpose1d_split_f16(X, P, None, cout, k, s, pad, op, pre_slope=0.1, out_act=tw_u, act_slope=0.2)
        tuning("convh_blocks", 3)
        u3 = _native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1)
        tuning("convh_blocks", 0)
        tuning("convu_resident", 1)
        assert torch.equal(u1, y) and torch.equal(u2, y2) and torch.equal(tw_u, twin) and torch.equal(u3, y)
    if cin >= 128:
        # 128-row tiles (csrc/convr_kernels.hpp convu_kernel: the window of a chunk converted once for two 64-row tiles,
        # an odd count of row tiles leaves the last pair half empty) against 64-row tiles: the same K order, same bits
        outs = []
        for rows64 in (1, 0):
            tuning("convt_rows64", rows64)
            tw = torch.empty_like(y)
            outs.append((_native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1).clone(),
                         _native.conv_transpose1d_split_f16(X, P, None, cout, k, s, pad, op, pre_slope=0.1, out_act=tw,
                                                            act_slope=0.2).clone(), tw))
            tuning("convh_blocks", 3)
            outs.append((_native.conv_transpose1d_split_f16(X, P, Bi, cout, k, s, pad, op, pre_slope=0.1).clone(),))
            tuning("convh_blocks", 0)
        tuning("convt_rows64", -1)
        tuning("convt_lean", 50)
        assert torch.equal(outs[0][0], y) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))
        assert torch.equal(outs[1][0], y) and torch.equal(outs[3][0], y)


def test_conv_transpose1d_split_f16_rejects():
    with pytest.raises(_native.NativeError, match="not built"):
        _native.pack_conv_transpose1d_split(torch.zeros((48, 32, 16), device=_dev()), 8)       # Cin = 48
    with pytest.raises(_native.NativeError, match="not built"):
        _native.pack_conv_transpose1d_split(torch.zeros((128, 32, 17), device=_dev()), 8)      # k != 2 s
    P = _native.pack_conv_transpose1d_split(torch.zeros((128, 32, 16), device=_dev()), 8)
    x = torch.zeros((1, 128, 10), device=_dev())
    with pytest.raises(_native.NativeError, match="pad="):
        _native.conv_transpose1d_split_f16(x, P, None, 32, 16, 8, 9, 0)
    with pytest.raises(_native.NativeError, match="alias"):
        _native.conv_transpose1d_split_f16(x, P, None, 32, 16, 8, 4, 0, out=x)


@pytest.mark.parametrize("blocks", [1, 3, 7])
def test_persistent_blocks_walk_many_tiles_and_cross_members(tuning, blocks):
    """With few persistent blocks every block walks several tiles and crosses from one member to the next (the small
    cases above give each block one tile): forced grid sizes, same results bit for bit as the full grid."""
    rng = np.random.RandomState(77)
    ks = (7, 11, 3)
    for C, T, dil in ((16, 1500, 3), (32, 900, 5), (64, 1030, 3), (128, 520, 1), (256, 260, 5)):
        ms = [_member(rng, 2, C, T, k, True) for k in ks]
        xs = [_t(m[0]) for m in ms]
        h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
        b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
        full = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
        refs = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
        tuning("pair_blocks", blocks)
        tuning("convh_blocks", blocks)
        few = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
        tuning("pair_blocks", 0)
        tuning("convh_blocks", 0)
        for yf, yw, ref in zip(full, few, refs):
            assert _rel(yw, ref) <= 4e-6
            assert torch.equal(yf, yw)
        if C in (64, 128):                           # fused (convp / convq_kernels.hpp) == two conv launches (convh), bit for bit
            mids = _native.conv1d_split_f16(xs, h1, b1s, list(ks), dil, pre_slope=0.1)
            two = _native.conv1d_split_f16(mids, h2, b2s, list(ks), 1, pre_slope=0.1, res=xs)
            for yf, yt in zip(full, two):
                assert torch.equal(yf, yt)
        if C <= 32:                                  # the fp32 kernels share the partition code
            f1, f2 = [_native.pack_pair(_t(m[1])) for m in ms], [_native.pack_pair(_t(m[3])) for m in ms]
            tuning("pair_blocks", blocks)
            y32 = _native.resblock1_fused(xs, f1, f2, b1s, b2s, list(ks), dil, 0.1)
            tuning("pair_blocks", 0)
            for y, ref in zip(y32, refs):
                assert _rel(y, ref) <= 2e-5


def test_pair_tile_forms_give_the_same_bits(tuning):
    """convq2_kernel (csrc/convq2_kernels.hpp: A operands from L2 straight into registers, no barrier in the K loops) on its
    narrow tiles (16 x 64 wave tiles) and on the wide ones (32 x 64: 256 columns at 64 channels, 128 at 128 channels with
    dilation 1 / 3), on full and three-block grids (long runs of warm tiles), two utterances, every tap count, with the MRF sum:
    the same MFMA order per output, so identical bits -- and the bits of the two conv launches (convh_kernel) a fused pair
    replaces.  (Until round 4 this test also ran the LDS-ring forms convq_kernel / convp_kernel, which gave the same bits and
    were removed in round 5.)"""
    rng = np.random.RandomState(123)
    ks = (11, 3, 7)
    # (the last four: shorter than a tile / than the halo, exactly one cold tile, one column into the first warm tile)
    for C, T, dil in ((128, 520, 1), (128, 1100, 5), (64, 1030, 3), (64, 300, 5), (128, 5, 5), (128, 54, 3), (128, 55, 1),
                      (64, 7, 3), (64, 2100, 1), (64, 247, 5), (128, 1500, 3), (128, 119, 1)):
        ms = [_member(rng, 2, C, T, k, True) for k in ks]
        xs = [_t(m[0]) for m in ms]
        h1, h2 = [_native.pack_pair(_t(m[1]), SPLIT) for m in ms], [_native.pack_pair(_t(m[3]), SPLIT) for m in ms]
        b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
        refs = [_pair_ref(x, w1, b1, w2, b2, dil, 0.1) for x, w1, b1, w2, b2 in ms]
        outs = {}
        # (wide: narrow tiles only / wide tiles wherever they exist; 64 channels, "pp": convq3_kernel -- two wave groups one
        # conv phase apart, each on 64-column tiles of its own share -- on full, three- and one-block grids)
        # ("pp2": convq4_kernel -- the same group pipeline as four-wave blocks of their own, two per CU)
        for wide in (1 << 20, 0, "pp", "pp2"):
            if wide in ("pp", "pp2") and C != 64:
                continue
            tuning("convp_pp", 1 if wide == "pp" else 2 if wide == "pp2" else 0)
            if wide not in ("pp", "pp2"):
                tuning("convp_wide", wide)
                tuning("convq_wide", wide)
            for blocks in (0, 3, 1) if wide in ("pp", "pp2") else (0, 3):
                tuning("convh_blocks", blocks)
                ys = _native.resblock1_fused(xs, h1, h2, b1s, b2s, list(ks), dil, 0.1, prec=SPLIT)
                merged = torch.empty_like(xs[0])
                _native.resblock1_fused([xs[1]], [h1[1]], [h2[1]], [b1s[1]], [b2s[1]], [3], dil, 0.1, outs=[merged], prec=SPLIT,
                                        add1=[ys[0]], add2=[ys[2]], out_div=3.0, act_slope=0.1)
                outs[(wide, blocks)] = ys + [merged]
        tuning("convh_blocks", 0)
        tuning("convp_wide", 20)
        tuning("convq_wide", 20)
        tuning("convp_pp", 0)
        base = outs[(1 << 20, 0)]
        for y, ref in zip(base[:3], refs):
            assert _rel(y, ref) <= 4e-6
        for key, ys in outs.items():
            for y, yb in zip(ys, base):
                assert torch.equal(y, yb), key
        mids = _native.conv1d_split_f16(xs, h1, b1s, list(ks), dil, pre_slope=0.1)
        two = _native.conv1d_split_f16(mids, h2, b2s, list(ks), 1, pre_slope=0.1, res=xs)
        for yf, yt in zip(base[:3], two):
            assert torch.equal(yf, yt)


def test_pair_results_do_not_depend_on_the_batch():
    """Bit-identity: an utterance gives the same bits alone and inside a batch (what lets a batch be
    sharded over GPUs), for the plain and the sum kernels, on both channel counts."""
    rng = np.random.RandomState(7)
    for C, T in ((16, 1500), (32, 700)):
        ks = (11, 7, 3)
        ms = [_member(rng, 3, C, T, k, True) for k in ks]
        xs = [_t(m[0]) for m in ms]
        w1s = [_native.pack_pair(_t(m[1])) for m in ms]
        w2s = [_native.pack_pair(_t(m[3])) for m in ms]
        b1s, b2s = [_t(m[2]) for m in ms], [_t(m[4]) for m in ms]
        full = _native.resblock1_fused(xs, w1s, w2s, b1s, b2s, list(ks), 3, 0.1)
        for b in range(3):
            one = _native.resblock1_fused([x[b:b + 1].contiguous() for x in xs], w1s, w2s, b1s, b2s, list(ks), 3, 0.1)
            for yf, yo in zip(full, one):
                assert torch.equal(yf[b:b + 1], yo)
        if C == 16:
            fs = _native.mrf_stage(xs, w1s, w2s, b1s, b2s, list(ks), 5, 0.1)
            for b in range(3):
                os_ = _native.mrf_stage([x[b:b + 1].contiguous() for x in xs], w1s, w2s, b1s, b2s, list(ks), 5, 0.1)
                assert torch.equal(fs[b:b + 1], os_)


def test_pair_rejects_what_it_is_not_built_for():
    dev = _dev()
    x = torch.zeros((1, 16, 50), device=dev)              # T % 4 != 0
    w = _native.pack_pair(torch.zeros((16, 16, 3), device=dev))
    with pytest.raises(_native.NativeError, match="multiple of 4"):
        _native.resblock1_fused([x], [w], [w], [None], [None], [3], 1, 0.1)
    x64 = torch.zeros((1, 64, 64), device=dev)
    w64 = _native.pack_pair(torch.zeros((64, 64, 3), device=dev))
    with pytest.raises(_native.NativeError, match="C = 64"):
        _native.resblock1_fused([x64], [w64], [w64], [None], [None], [3], 1, 0.1)
    x16 = torch.zeros((1, 16, 64), device=dev)
    with pytest.raises(_native.NativeError, match="taps"):
        _native.resblock1_fused([x16], [w], [w], [None], [None], [5], 1, 0.1)
    with pytest.raises(_native.NativeError, match="dilation"):
        _native.resblock1_fused([x16], [w], [w], [None], [None], [3], 2, 0.1)
    x32 = torch.zeros((1, 32, 64), device=dev)
    w32 = _native.pack_pair(torch.zeros((32, 32, 3), device=dev))
    with pytest.raises(_native.NativeError, match="mrf"):
        _native.mrf_stage([x32] * 3, [w32] * 3, [w32] * 3, [None] * 3, [None] * 3, [3, 7, 11], 5, 0.1)


def test_resblock1_on_the_fused_path_vs_reference_goldens():
    """ResBlock1(x) at a length the fused pair kernels take (T = 52), 16 and 32 channels, 3 / 7 / 11 taps,
    against the REFERENCE module's outputs (tests/golden/blocks_t52.npz, made by make_golden.py)."""
    import os
    from fastvocoder_amd.generator import modules as M
    from tests import cases
    g = np.load(os.path.join(cases.ROOT, "tests", "golden", "blocks_t52.npz"))
    for ch in (16, 32):
        x = torch.from_numpy(g[f"x{ch}"]).to(_dev())
        for k in (3, 7, 11):
            rb = M.ResBlock1(ch, k, (1, 3, 5)).to(_dev())
            flat, off = g[f"rb1_c{ch}_k{k}_params"], 0
            with torch.no_grad():
                for p in rb.parameters():
                    p.copy_(torch.from_numpy(flat[off:off + p.numel()].reshape(tuple(p.shape))))
                    off += p.numel()
            y = rb(x)
            plans = {name: plan for (name, _), (_, plan) in rb._fv_plans.items()}
            assert "forward_fused" in plans and plans["forward_fused"].num_ops() == 3
            assert _rel(y, g[f"rb1_c{ch}_k{k}_out"]) <= 2e-5


# ---------------------------------------------------------------------------
# the domain of the split-f16 arithmetic: activation scales, and what happens outside the f16 range
# ---------------------------------------------------------------------------
def _rel_to(a, ref):
    """max |a - ref| relative to max |ref| (no floor: these tests move the tensors' scale)."""
    a = a.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape
    return float(np.abs(a - ref).max()) / float(np.abs(ref).max())


@pytest.mark.parametrize("scale", [1e-4, 1e-2, 1e2, 1e3])
def test_split_kernels_at_activation_scales(scale):
    """Every split-f16 kernel family (pairh at 16 / 32 channels, convp at 64, convh at 128, convt) with activations and
    biases at scale 1e-4 ... 1e3 against the double-accumulating C oracle.  v = h1 + h2/2048 keeps 22 bits of every
    operand whatever its scale as long as h1 is a normal f16; below 6.1e-5 h1 is a SUBNORMAL f16 and the claim is that
    gfx950 (conversion and MFMA alike) does not flush it -- flushed, the 46 % of a 1e-4-scale tensor below that bound
    would keep 11 bits and the error would be ~1e-4 of the tensor, not the 1e-5 allowed here (operands below the
    smallest normal f16 keep an ABSOLUTE resolution of 2^-35, which is 3e-7 of a 5e-5 value)."""
    tol = 1e-5 if scale < 1e-3 else 4e-6
    rng = np.random.RandomState(int(1000 + np.log10(scale)))
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    for C, T, k, dil in ((16, 700, 7, 3), (32, 500, 3, 5), (64, 300, 3, 1), (128, 200, 3, 3)):
        x, w1, b1, w2, b2 = _member(rng, 1, C, T, k, True)
        x, b1, b2 = (x * scale).astype(np.float32), (b1 * scale).astype(np.float32), (b2 * scale).astype(np.float32)
        ref = _pair_ref(x, w1, b1, w2, b2, dil, 0.1)
        y = _native.resblock1_fused([_t(x)], [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)],
                                    [_t(b1)], [_t(b2)], [k], dil, 0.1, prec=SPLIT, guard=guard)[0]
        assert _rel_to(y, ref) <= tol, (C, scale, _rel_to(y, ref))
    cin, cout, T, s = 128, 64, 150, 4
    x = (rng.randn(1, cin, T) * scale).astype(np.float32)
    w = (rng.randn(cin, cout, 2 * s) / np.sqrt(cin * 2)).astype(np.float32)
    b = (rng.randn(cout) * scale).astype(np.float32)
    ref = oo.conv_transpose1d(x, w, b, s, s // 2, 0, pre_slope=0.1)
    y = _native.conv_transpose1d_split_f16(_t(x), _native.pack_conv_transpose1d_split(_t(w), s), _t(b), cout, 2 * s, s,
                                           s // 2, 0, pre_slope=0.1, guard=guard)
    assert _rel_to(y, ref) <= tol, ("convt", scale, _rel_to(y, ref))
    # inside the domain no guard is raised; a tensor of scale 1e-4 (largest magnitude 4e-4 < 2^-10) is below its LOW side:
    # the kernels flag it (4) -- a plan would repeat the call on the fp32 kernels -- although gfx950 still keeps it at 1e-5
    assert int(guard.item()) == (_native.GUARD_LOW if scale < 1e-3 else 0)


def test_range_guard_of_the_split_kernels():
    """Outside the f16 range (|v| >= 65520) the split-f16 kernels do not apply to ACTIVATIONS: every kernel family
    raises its guard word when an activation (input or the intermediate of a fused pair) overflows -- while operands
    just inside the range leave it clear.  Weights of any finite magnitude are rescaled when they are packed (the pack
    kernels raise their flag for a non-finite one only)."""
    rng = np.random.RandomState(5)
    dev = _dev()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    for C, k in ((16, 3), (32, 7), (64, 11), (128, 3)):
        w = (rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)
        for big in (65000.0, -70000.0, 3.0e30):       # any finite weight: its row is rescaled by a power of two
            w[1, 2, 0] = big
            _native.pack_pair(_t(w), SPLIT, flag)
            assert int(flag.item()) == 0
        w[1, 2, 0] = np.nan                           # ... a non-finite one is flagged
        _native.pack_pair(_t(w), SPLIT, flag)
        assert int(flag.item()) == 1
        flag.zero_()
        w[1, 2, 0] = 0.01
    wt = (rng.randn(128, 32, 8)).astype(np.float32)
    wt[5, 5, 5] = np.inf
    _native.pack_conv_transpose1d_split(_t(wt), 4, flag)
    assert int(flag.item()) == 1
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    for C, T, k, dil in ((16, 600, 3, 1), (32, 600, 7, 3), (64, 400, 3, 5), (128, 300, 3, 1)):
        x, w1, b1, w2, b2 = _member(rng, 2, C, T, k, True)
        ws = [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)]
        args = ([_t(b1)], [_t(b2)], [k], dil, 0.1)
        x[1, C // 2, T // 3] = 60000.0                # inside: finite result, no flag
        y = _native.resblock1_fused([_t(x)], *ws, *args, prec=SPLIT, guard=guard)[0]
        assert bool(torch.isfinite(y).all()) and int(guard.item()) == 0
        assert _rel(y, _pair_ref(x, w1, b1, w2, b2, dil, 0.1)) <= 4e-6
        x[1, C // 2, T // 3] = 1.0e6                  # outside: the guard is raised (and the output is not finite)
        y = _native.resblock1_fused([_t(x)], *ws, *args, prec=SPLIT, guard=guard)[0]
        assert int(guard.item()) == 1 and not bool(torch.isfinite(y).all())
        guard.zero_()
        # ... while the fp32 kernels (16 / 32 channels) take the same input in their stride
        if C <= 32:
            y32 = _native.resblock1_fused([_t(x)], [_native.pack_pair(_t(w1))], [_native.pack_pair(_t(w2))], *args)[0]
            assert bool(torch.isfinite(y32).all()) and _rel_to(y32, _pair_ref(x, w1, b1, w2, b2, dil, 0.1)) <= 2e-5
    # the intermediate of a fused pair overflows although its input does not: large first conv
    C, T, k = 16, 500, 3
    x, w1, b1, w2, b2 = _member(rng, 1, C, T, k, True)
    w1 = (w1 * 3000.0).astype(np.float32)
    x = (x * 100.0).astype(np.float32)
    y = _native.resblock1_fused([_t(x)], [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)],
                                [_t(b1)], [_t(b2)], [k], 1, 0.1, prec=SPLIT, guard=guard)[0]
    assert int(guard.item()) == 1
    guard.zero_()
    x = rng.randn(1, 128, 100).astype(np.float32)
    x[0, 3, 50] = -2.0e5
    wT = (rng.randn(128, 64, 8) / 16).astype(np.float32)
    _native.conv_transpose1d_split_f16(_t(x), _native.pack_conv_transpose1d_split(_t(wT), 4), None, 64, 8, 4, 2, 0,
                                       pre_slope=1.0, guard=guard)
    assert int(guard.item()) == 1


# ---------------------------------------------------------------------------
# two-source 1x1 conv with split-f16 operands (convg_kernel): ResidualStack's tail
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(1, 128, 300), (2, 256, 257), (1, 512, 130), (3, 128, 1), (1, 256, 1000), (2, 128, 129)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv1x1_2src_split_f16_vs_oracle(case, tuning):
    """y = post(W1 lrelu(x, s) + W2 x2 + b + res) (reference modules.py:362-366,382: stack[4](act(h)) + skip_layer(c))
    against the oracle's two 1x1 convs: every channel count, ragged lengths, residual, ReLU, activated twin, and a few
    persistent blocks walking many (row tile, chunk) items -- same bits."""
    B, C, T = case
    rng = np.random.RandomState(C + T)
    x, x2 = rng.randn(B, C, T).astype(np.float32), rng.randn(B, C, T).astype(np.float32)
    w1 = (rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32)
    w2 = (rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32)
    b = rng.randn(C).astype(np.float32)
    res = rng.randn(B, C, T).astype(np.float32)
    ref = oo.conv1d(x, w1, None, pre_slope=0.2) + oo.conv1d(x2, w2, b)
    P = _native.pack_conv1x1_2src_split(_t(w1), _t(w2))
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    tuning("convg_rows64", 0)                          # 128-row tiles whatever the size (the default picks by item count)
    y = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2, guard=guard)
    assert tuple(y.shape) == ref.shape and _rel(y, ref) <= 4e-6
    twin = torch.empty_like(y)
    y2 = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, None, pre_slope=0.2, res=_t(res), post=_native.POST_RELU,
                                        out_act=twin, act_slope=0.1, guard=guard)
    want = np.maximum(ref - b[None, :, None] + res, 0)
    assert _rel(y2, want) <= 4e-6 and _rel(twin, oo.lrelu(want, 0.1)) <= 4e-6
    assert int(guard.item()) == 0
    tuning("convh_blocks", 3)
    few = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2)
    tuning("convh_blocks", 0)
    assert torch.equal(few, y)
    if B > 1:
        one = _native.conv1x1_2src_split_f16(_t(x[1:2]), _t(x2[1:2]), P, _t(b), pre_slope=0.2)
        assert torch.equal(one, y[1:2])
    # 128-row tiles (convr_kernel, the default) against 64-row tiles (convg_kernel): the same K order per output
    tuning("convg_rows64", 1)
    narrow = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2)
    twin64 = torch.empty_like(y)
    narrow2 = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, None, pre_slope=0.2, res=_t(res), post=_native.POST_RELU,
                                             out_act=twin64, act_slope=0.1)
    tuning("convg_rows64", -1)
    auto = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2)
    tuning("convg_rows64", 0)
    assert torch.equal(narrow, y) and torch.equal(narrow2, y2) and torch.equal(twin64, twin) and torch.equal(auto, y)
    x2[0, 1, 0] = 3.0e5                                # the raw branch leaves the f16 range: guard
    _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2, guard=guard)
    assert int(guard.item()) == 1


def test_conv1x1_2src_split_f16_rejects():
    with pytest.raises(_native.NativeError, match="not built"):
        _native.pack_conv1x1_2src_split(torch.zeros((64, 64, 1), device=_dev()), torch.zeros((64, 64, 1), device=_dev()))
    z = torch.zeros((128, 128, 1), device=_dev())
    P = _native.pack_conv1x1_2src_split(z, z)
    x = torch.zeros((1, 128, 10), device=_dev())
    with pytest.raises(_native.NativeError, match="alias"):
        _native.conv1x1_2src_split_f16(x, x, P, None, out=x)


# ---------------------------------------------------------------------------
# the LOW side of the domain: small weights behind large activations, small tensors
# ---------------------------------------------------------------------------
def _pow2(e):
    return np.float32(2.0 ** e)


@pytest.mark.parametrize("wexp,aexp", [(-10, 10), (-14, 13), (-17, 14), (-20, 14), (-40, 12), (12, -6)])
def test_split_kernels_at_weight_scales(wexp, aexp):
    """Weights x 2^wexp (down to far below the smallest normal f16, 2^-14) behind activations x 2^aexp, every split-f16
    kernel family -- pairh (16 / 32 channels), convp (64), convq (128), convh / convs (64 ... 256, 64- and 128-row tiles),
    convt / convu, convg / convr -- against the double-accumulating C oracle at the SAME tolerance as at ordinary scales
    (4e-6 of the tensor's scale; an fp32 FMA chain is at ~1e-6): the pack functions' per-row power-of-two prescale makes
    the weights' magnitude irrelevant.  Round 3's kernels were 10x ... 100x off here (tests/test_split_precision.py)."""
    ws, xs = _pow2(wexp), _pow2(aexp - 2)      # (randn reaches 4.5: x 2^(aexp - 2) stays below 65504 at aexp = 14)
    rng = np.random.RandomState(7000 + wexp * 31 + aexp)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    tol = 4e-6
    # fused pairs: x O(1), conv1 x 2^aexp (the INTERMEDIATE is the large activation), conv2's weights x 2^wexp with rows of
    # unequal scale (weight norm's g).  The pair's branch y - x is what is compared (fp32 rounding of x + branch allowed for).
    for C, T, k, dil in ((16, 700, 7, 3), (32, 500, 11, 5), (64, 300, 3, 1), (128, 200, 7, 3)):
        x, w1, b1, w2, b2 = _member(rng, 2, C, T, k, True)
        rows = (2.0 ** rng.randint(-2, 3, size=(C, 1, 1))).astype(np.float32)
        w1, b1, w2 = w1 * xs, b1 * xs, w2 * ws * rows
        ref = _pair_ref(x, w1, b1, w2, b2, dil, 0.1)
        y = _native.resblock1_fused([_t(x)], [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)],
                                    [_t(b1)], [_t(b2)], [k], dil, 0.1, prec=SPLIT, guard=guard)[0]
        br, br_ref = y.detach().cpu().numpy().astype(np.float64) - x, ref.astype(np.float64) - x
        err = np.abs(br - br_ref).max()
        assert err <= tol * np.abs(br_ref).max() + 2.5e-7 * np.abs(ref).max(), (C, wexp, aexp, err / np.abs(br_ref).max())
    # plain convs on 64-row (convh) and 128-row (convs) tiles, zero and reflection padding
    for C, T, k, dil, refl in ((64, 300, 7, 3, False), (128, 260, 11, 5, False), (256, 200, 3, 9, True), (512, 140, 3, 1, False)):
        x = rng.randn(1, C, T).astype(np.float32) * xs
        w = (rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32) * ws
        b = rng.randn(C).astype(np.float32)
        ref = oo.conv1d(x, w, b, dil=dil, pad=(k - 1) * dil // 2, pad_mode=oo.PAD_REFLECT if refl else oo.PAD_ZERO, pre_slope=0.1)
        P = _native.pack_pair(_t(w), SPLIT)
        for rows64 in (1, 0):
            _native.tuning_set("convh_rows64", rows64)
            y = _native.conv1d_split_f16([_t(x)], [P], [_t(b)], [k], dil, pre_slope=0.1, guard=guard,
                                         pad_mode=_native.PAD_REFLECT if refl else _native.PAD_ZERO)[0]
            _native.tuning_set("convh_rows64", -1)
            assert _rel_to(y, ref) <= tol, ("conv", C, rows64, wexp, aexp, _rel_to(y, ref))
    # transposed convs (64-row and 128-row tiles)
    for cin, cout, T, s in ((64, 32, 150, 3), (128, 64, 150, 4), (256, 128, 90, 8)):
        x = rng.randn(1, cin, T).astype(np.float32) * xs
        w = (rng.randn(cin, cout, 2 * s) / np.sqrt(cin * 2)).astype(np.float32) * ws
        w *= (2.0 ** rng.randint(-2, 3, size=(cin, 1, 1))).astype(np.float32)                  # weight norm's g of a ConvTranspose1d
        b = rng.randn(cout).astype(np.float32)
        ref = oo.conv_transpose1d(x, w, b, s, s // 2 + s % 2, s % 2, pre_slope=0.1)
        P = _native.pack_conv_transpose1d_split(_t(w), s)
        for rows64 in (1, 0):
            _native.tuning_set("convt_rows64", rows64)
            y = _native.conv_transpose1d_split_f16(_t(x), P, _t(b), cout, 2 * s, s, s // 2 + s % 2, s % 2, pre_slope=0.1, guard=guard)
            _native.tuning_set("convt_rows64", -1)
            assert _rel_to(y, ref) <= tol, ("convt", cin, rows64, wexp, aexp, _rel_to(y, ref))
    # ResidualStack's 1x1 + skip GEMM: the two halves of the K range at different scales
    for C, T in ((128, 300), (256, 200)):
        x, x2 = rng.randn(1, C, T).astype(np.float32) * xs, rng.randn(1, C, T).astype(np.float32) * xs
        w1 = (rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32) * ws
        w2 = (rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32) * ws * _pow2(-3)
        b = rng.randn(C).astype(np.float32)
        ref = oo.conv1d(x, w1, None, pre_slope=0.2) + oo.conv1d(x2, w2, b)
        P = _native.pack_conv1x1_2src_split(_t(w1), _t(w2))
        for rows64 in (1, 0):
            _native.tuning_set("convg_rows64", rows64)
            y = _native.conv1x1_2src_split_f16(_t(x), _t(x2), P, _t(b), pre_slope=0.2, guard=guard)
            _native.tuning_set("convg_rows64", -1)
            assert _rel_to(y, ref) <= tol, ("convg", C, rows64, wexp, aexp, _rel_to(y, ref))
    assert int(guard.item()) == 0                     # nothing here leaves the domain on either side


def test_low_side_of_the_range_guard():
    """A tensor that is small as a WHOLE (largest magnitude below 2^-10) is outside the split-f16 domain: every kernel
    family raises guard value 4 for it (input window, or the intermediate of a fused pair); a tensor with small regions,
    an all-zero tensor and one whose maximum is 2^-9 are inside."""
    rng = np.random.RandomState(77)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())

    def fired():
        v = int(guard.item())
        guard.zero_()
        return v

    for C, T, k, dil in ((16, 2000, 3, 1), (32, 900, 7, 3), (64, 400, 3, 5), (128, 300, 3, 1)):
        x, w1, b1, w2, b2 = _member(rng, 2, C, T, k, True)
        ws = [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)]
        args = ([_t(b1)], [_t(b2)], [k], dil, 0.1)
        xn = (x / np.abs(x).max()).astype(np.float32)
        _native.resblock1_fused([_t(xn * _pow2(-9))], *ws, *args, prec=SPLIT, guard=guard)
        assert fired() == 0                           # largest magnitude 2^-9: inside
        _native.resblock1_fused([_t(xn * _pow2(-11))], *ws, *args, prec=SPLIT, guard=guard)
        assert fired() == _native.GUARD_LOW                           # 2^-11: the whole tensor is below the low side (although the
                                                      # intermediate, which the biases dominate, is ordinary)
        quiet = x.copy()
        quiet[:, :, T // 3: T // 3 + 40] *= _pow2(-20)     # silence inside an ordinary signal is not
        y = _native.resblock1_fused([_t(quiet)], *ws, *args, prec=SPLIT, guard=guard)[0]
        assert fired() == 0 and _rel(y, _pair_ref(quiet, w1, b1, w2, b2, dil, 0.1)) <= 4e-6
        # all zeros (the zero-mel pass of a model without biases): exact in f16, nothing to lose
        _native.resblock1_fused([_t(np.zeros_like(x))], *ws, [None], [None], [k], dil, 0.1, prec=SPLIT, guard=guard)
        assert fired() == 0
        # the INTERMEDIATE of the fused pair is small although input and output are not: conv1 x 2^-14, conv2 x 2^14
        small = [_native.pack_pair(_t(w1 * _pow2(-14)), SPLIT)], [_native.pack_pair(_t(w2 * _pow2(14)), SPLIT)]
        _native.resblock1_fused([_t(x)], *small, [_t(b1 * _pow2(-14))], [_t(b2)], [k], dil, 0.1, prec=SPLIT, guard=guard)
        assert fired() == _native.GUARD_LOW
    x = rng.randn(1, 128, 200).astype(np.float32)
    xn = (x / np.abs(x).max()).astype(np.float32)
    w = (rng.randn(128, 128, 3) / 20).astype(np.float32)
    P = _native.pack_pair(_t(w), SPLIT)
    for rows64 in (1, 0):
        _native.tuning_set("convh_rows64", rows64)
        _native.conv1d_split_f16([_t(xn * _pow2(-12))], [P], [None], [3], 1, pre_slope=0.1, guard=guard)
        assert fired() == _native.GUARD_LOW
        _native.conv1d_split_f16([_t(xn * _pow2(-8))], [P], [None], [3], 1, pre_slope=0.1, guard=guard)
        assert fired() == 0
    _native.tuning_set("convh_rows64", -1)
    wT = (rng.randn(128, 64, 8) / 16).astype(np.float32)
    PT = _native.pack_conv_transpose1d_split(_t(wT), 4)
    for rows64 in (1, 0):
        _native.tuning_set("convt_rows64", rows64)
        _native.conv_transpose1d_split_f16(_t(xn * _pow2(-12)), PT, None, 64, 8, 4, 2, 0, pre_slope=1.0, guard=guard)
        assert fired() == _native.GUARD_LOW
        _native.conv_transpose1d_split_f16(_t(xn * _pow2(-8)), PT, None, 64, 8, 4, 2, 0, pre_slope=1.0, guard=guard)
        assert fired() == 0
    _native.tuning_set("convt_rows64", -1)
    w1 = (rng.randn(128, 128, 1) / 11).astype(np.float32)
    PG = _native.pack_conv1x1_2src_split(_t(w1), _t(w1))
    for rows64 in (1, 0):
        _native.tuning_set("convg_rows64", rows64)
        _native.conv1x1_2src_split_f16(_t(xn * _pow2(-12)), _t(xn * _pow2(-13)), PG, None, pre_slope=0.2, guard=guard)
        assert fired() == _native.GUARD_LOW
        # the two sources are two operand tensors: one of them at an ordinary scale does not hide the other
        _native.conv1x1_2src_split_f16(_t(xn * _pow2(-12)), _t(xn), PG, None, pre_slope=0.2, guard=guard)
        assert fired() == _native.GUARD_LOW
        _native.conv1x1_2src_split_f16(_t(xn), _t(xn * _pow2(-12)), PG, None, pre_slope=0.2, guard=guard)
        assert fired() == _native.GUARD_LOW
        _native.conv1x1_2src_split_f16(_t(xn * _pow2(-3)), _t(xn * _pow2(-9)), PG, None, pre_slope=0.2, guard=guard)
        assert fired() == 0
    _native.tuning_set("convg_rows64", -1)


def test_guard_sides_combine_whatever_the_order():
    """The two sides of a guard word are separate bytes (FV_GUARD_HIGH / FV_GUARD_LOW, fastvocoder_hip.h): an overflow
    followed by a quiet launch -- or the reverse -- leaves BOTH raised.  (With one 32-bit value per side the last writer
    won: 'overflow, then low' read as 'low' and the module stayed on the split kernels -- ADVICE r5.)"""
    rng = np.random.RandomState(78)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    for C, T, k, dil in ((16, 1200, 3, 1), (64, 400, 7, 3)):
        x, w1, b1, w2, b2 = _member(rng, 1, C, T, k, True)
        ws = [_native.pack_pair(_t(w1), SPLIT)], [_native.pack_pair(_t(w2), SPLIT)]
        args = ([_t(b1)], [_t(b2)], [k], dil, 0.1)
        xn = (x / np.abs(x).max()).astype(np.float32)
        quiet, loud = xn * _pow2(-12), xn.copy()
        loud[0, 0, T // 2] = 7e4                      # beyond the f16 range: inf in the split, NaN in the sums
        both = _native.GUARD_HIGH | _native.GUARD_LOW
        for first, second in ((loud, quiet), (quiet, loud)):
            guard.zero_()
            _native.resblock1_fused([_t(first)], *ws, *args, prec=SPLIT, guard=guard)
            _native.resblock1_fused([_t(second)], *ws, *args, prec=SPLIT, guard=guard)
            assert int(guard.item()) == both, (C, int(guard.item()))
