"""GPU parity of the one-launch MRF stage (csrc/mrfh_kernels.hpp; 32 channels: csrc/mrfw_kernels.hpp) through the C ABI (fv_mrf_stage_split_f16 and its plan
op): against the C oracle's convs on the same seeded inputs (4e-6 of the tensor's scale, the split-f16 kernels' bar), and
BIT FOR BIT against the pair launches it replaces (fv_resblock1_fused_ex) -- over windows, runs and shares of every shape:
one tile, tiles with history, runs that start inside an utterance, shares that cross utterance ends, both block shapes.
"""
import numpy as np
import pytest
import torch

from fastvocoder_amd import _native
from oracle import ops as oo

pytestmark = pytest.mark.gpu
SPLIT = _native.PAIR_SPLIT_F16
DILS = (1, 3, 5)


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


@pytest.fixture
def tuning():
    yield _native.tuning_set
    for k, v in {"mrf_blocks": 0, "mrf_shape": 0, "mrf_prio": 1}.items():
        _native.tuning_set(k, v)


def _stage_weights(rng, ks, bias=True, C=16):
    """Nine pairs: index 3 j + p = pair p of ResBlock j (taps ks[j], dilation DILS[p])."""
    w1, w2, b1, b2 = [], [], [], []
    for j in range(3):
        for _ in range(3):
            k = ks[j]
            w1.append((rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32))
            w2.append((rng.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32))
            b1.append(rng.randn(C).astype(np.float32) * 0.3 if bias else None)
            b2.append(rng.randn(C).astype(np.float32) * 0.3 if bias else None)
    return w1, w2, b1, b2


def _oracle_stage(x, ws, ks, slope=0.1):
    """((r0 + r1) + r2) / 3 with r_j = ResBlock1_j(x) (reference hifigan.py:97-103, modules.py:223-230), C oracle."""
    w1, w2, b1, b2 = ws
    rs = []
    for j in range(3):
        r = x
        for p in range(3):
            i, k, d = 3 * j + p, ks[j], DILS[p]
            mid = oo.conv1d(r, w1[i], b1[i], dil=d, pad=(k - 1) * d // 2, pre_slope=slope)
            r = oo.conv1d(mid, w2[i], b2[i], dil=1, pad=(k - 1) // 2, pre_slope=slope) + r
        rs.append(r)
    return ((rs[0] + rs[1]) + rs[2]) / np.float32(3.0)


def _pairs_stage(x, ws, ks, slope=0.1):
    """The same stage as round 4 ran it: pair launches (fv_resblock1_fused_ex), the merge in the last one."""
    w1, w2, b1, b2 = ws
    P1 = [_native.pack_pair(_t(w), SPLIT) for w in w1]
    P2 = [_native.pack_pair(_t(w), SPLIT) for w in w2]
    cur = [x, x, x]
    for p in range(2):
        idx = [3 * j + p for j in range(3)]
        cur = _native.resblock1_fused(cur, [P1[i] for i in idx], [P2[i] for i in idx], [_t(b1[i]) for i in idx],
                                      [_t(b2[i]) for i in idx], list(ks), DILS[p], slope, prec=SPLIT)
    idx = [5, 8]
    r12 = _native.resblock1_fused(cur[1:], [P1[i] for i in idx], [P2[i] for i in idx], [_t(b1[i]) for i in idx],
                                  [_t(b2[i]) for i in idx], list(ks[1:]), DILS[2], slope, prec=SPLIT)
    return _native.resblock1_fused(cur[:1], [P1[2]], [P2[2]], [_t(b1[2])], [_t(b2[2])], [ks[0]], DILS[2], slope, prec=SPLIT,
                                   add1=[r12[0]], add2=[r12[1]], out_div=3.0)[0]


def _pack(ws, ks, flag=None):
    w1, w2, b1, b2 = ws
    return _native.pack_mrf_stage([_t(w) for w in w1], [_t(w) for w in w2], [_t(b) for b in b1], [_t(b) for b in b2],
                                  list(ks), flag)


STAGE_CASES = [
    # B, T, taps of the three ResBlocks, bias, blocks (0: one per CU / per 128 columns), shape
    (1, 40, (3, 7, 11), True, 0, 0),        # shorter than the halo: one cold tile, both sequence ends inside
    (2, 200, (3, 7, 11), True, 0, 0),       # a share per 128 columns: every tile cold, runs start inside the utterance
    (1, 516, (3, 7, 11), True, 1, 0),       # exactly one window's final columns ... one block
    (1, 1500, (3, 7, 11), True, 1, 0),      # one block: a cold tile, then two tiles on history
    (3, 1201, (11, 3, 7), False, 2, 0),     # two blocks over three utterances: shares cross utterance ends, no bias
    (2, 2000, (7, 7, 3), True, 3, 0),       # repeated tap counts, odd length, blocks that end inside a window
    (1, 4003, (3, 7, 11), True, 5, 1),      # the 16-wave block shape (512-column windows)
    (2, 1100, (11, 11, 11), True, 1, 1),
    (1, 5, (3, 7, 11), True, 0, 1),
]


@pytest.mark.parametrize("case", STAGE_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_mrf_stage_vs_oracle_and_pair_launches(case, tuning):
    B, T, ks, bias, blocks, shape = case
    rng = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = rng.randn(B, 16, T).astype(np.float32)
    ws = _stage_weights(rng, ks, bias)
    ref = _oracle_stage(x, ws, ks)
    tuning("mrf_blocks", blocks)
    tuning("mrf_shape", shape)
    X, P = _t(x), _pack(ws, ks)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    y = _native.mrf_stage_split_f16(X, P, ks, guard=guard)
    assert _rel(y, ref) <= 4e-6
    assert int(guard.item()) == 0
    if T % 4 == 0:                                        # (the pair kernels want 16-byte aligned rows)
        assert torch.equal(y, _pairs_stage(X, ws, ks)), "one launch and the four pair launches differ"
    # activated twin next to the raw output, and the in-place form
    twin = torch.empty_like(y)
    y2 = _native.mrf_stage_split_f16(X, P, ks, out_act=twin, act_slope=0.1)
    assert torch.equal(y2, y) and _rel(twin, oo.lrelu(ref, 0.1)) <= 4e-6
    y3 = _native.mrf_stage_split_f16(X, P, ks, act_slope=0.1)
    assert torch.equal(y3, twin)


def test_mrf_stage_does_not_depend_on_blocks_or_batch(tuning):
    """Every output element is computed the same way whichever block owns its window and wherever a run starts: block
    counts, block shapes and batch decompositions give the same bits."""
    rng = np.random.RandomState(5)
    ks = (3, 7, 11)
    x = rng.randn(3, 16, 2600).astype(np.float32)
    ws = _stage_weights(rng, ks)
    X, P = _t(x), _pack(ws, ks)
    want = _native.mrf_stage_split_f16(X, P, ks)
    for blocks, shape in ((1, 0), (2, 0), (7, 0), (61, 0), (1, 1), (9, 1)):
        tuning("mrf_blocks", blocks)
        tuning("mrf_shape", shape)
        assert torch.equal(_native.mrf_stage_split_f16(X, P, ks), want), (blocks, shape)
        for b in range(3):
            assert torch.equal(_native.mrf_stage_split_f16(X[b:b + 1].contiguous(), P, ks)[0], want[b]), (blocks, shape, b)


@pytest.mark.parametrize("case", [(1, 37, 0, 0), (2, 1000, 1, 0), (1, 1531, 2, 0), (2, 777, 3, 1), (1, 3000, 0, 0)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_mrf_stage_with_folded_output_conv(case, tuning):
    """HiFi-GAN's last launch (hifigan.py:97-106): stage, lrelu(0.01), conv_post (16 -> 1, 7 taps), tanh -- against the
    oracle, and bit for bit against the stage followed by the narrow conv as a launch of its own (fv_conv1d_fused)."""
    B, T, blocks, shape = case
    rng = np.random.RandomState(31 * T + blocks)
    ks = (3, 7, 11)
    x = rng.randn(B, 16, T).astype(np.float32)
    ws = _stage_weights(rng, ks)
    wp = (rng.randn(1, 16, 7) / np.sqrt(16 * 7)).astype(np.float32)
    bp = rng.randn(1).astype(np.float32)
    ref = np.tanh(oo.conv1d(_oracle_stage(x, ws, ks), wp, bp, pad=3, pre_slope=0.01).astype(np.float64))
    tuning("mrf_blocks", blocks)
    tuning("mrf_shape", shape)
    X, P = _t(x), _pack(ws, ks)
    y = _native.mrf_stage_split_f16(X, P, ks, fold=(_t(wp.reshape(16, 7)), _t(bp)), act_slope=0.01, post=_native.POST_TANH)
    assert tuple(y.shape) == (B, 1, T) and _rel(y, ref) <= 4e-6
    stage = _native.mrf_stage_split_f16(X, P, ks)
    two = _native.conv1d_fused(stage, _native.pack_conv1d(_t(wp)), _t(bp), 1, 7, pad=3, pre_slope=0.01, post=_native.POST_TANH)
    assert torch.equal(y, two), "folded and separate output conv differ"
    # the plan op with the fold, and without a bias
    plan = _native.Plan(16)
    plan.add_mrf_stage(_native.SLOT_IN, 2, P, 16, ks, DILS, 0.1)
    plan.set_pair_output_conv(_t(wp.reshape(16, 7)), None, _native.SLOT_OUT, 0.01, _native.POST_TANH)
    assert plan.num_ops() == 1
    ref0 = np.tanh(oo.conv1d(_oracle_stage(x, ws, ks), wp, None, pad=3, pre_slope=0.01).astype(np.float64))
    assert _rel(plan.run(X), ref0) <= 4e-6


def test_mrf_stage_guards():
    """Both sides of the split-f16 domain: an activation beyond the f16 range raises the guard word (1) and the output is
    not finite; an input that is tiny as a whole raises the low-side guard (4); zeros raise nothing."""
    rng = np.random.RandomState(11)
    ks = (3, 7, 11)
    ws = _stage_weights(rng, ks)
    P = _pack(ws, ks)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    x = rng.randn(2, 16, 900).astype(np.float32)
    y = _native.mrf_stage_split_f16(_t(x), P, ks, guard=guard)
    assert int(guard.item()) == 0 and bool(torch.isfinite(y).all())
    big = x.copy()
    big[1, 7, 450] = 1.0e6
    y = _native.mrf_stage_split_f16(_t(big), P, ks, guard=guard)
    assert int(guard.item()) == 1 and not bool(torch.isfinite(y).all())
    guard.zero_()
    nob = _stage_weights(rng, ks, bias=False)
    y = _native.mrf_stage_split_f16(_t(x * np.float32(2.0 ** -14)), _pack(nob, ks), ks, guard=guard)
    assert int(guard.item()) == _native.GUARD_LOW
    guard.zero_()
    y = _native.mrf_stage_split_f16(_t(np.zeros_like(x)), _pack(nob, ks), ks, guard=guard)
    assert int(guard.item()) == 0 and float(y.abs().max()) == 0.0
    # a non-finite weight raises the pack kernels' flag
    flag = torch.zeros(1, dtype=torch.int32, device=_dev())
    bad = _stage_weights(rng, ks)
    bad[1][4][3, 2, 1] = np.inf
    _pack(bad, ks, flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


def test_mrf_stage_refuses_what_it_is_not_built_for():
    rng = np.random.RandomState(2)
    ks = (3, 7, 11)
    P = _pack(_stage_weights(rng, ks), ks)
    x = _t(rng.randn(1, 16, 64).astype(np.float32))
    with pytest.raises(_native.NativeError, match="dilations"):
        _native.mrf_stage_split_f16(x, P, ks, dils=(1, 3, 9))
    with pytest.raises(_native.NativeError):
        _native.mrf_stage_split_f16(x, P, (3, 5, 11))
    assert not _native.mrf_stage_supported(64, ks, DILS) and not _native.mrf_stage_supported(16, (3, 7), DILS)
    assert _native.mrf_stage_supported(16, (11, 11, 3), DILS) and _native.mrf_stage_supported(32, (7, 3, 11), DILS)


# ---- 32 channels (csrc/mrfw_kernels.hpp): 384-column windows, weights in pieces of four taps, history in the workspace ----
STAGE32_CASES = [
    # B, T, taps of the three ResBlocks, bias, blocks (0: one per CU / per 128 columns)
    (1, 40, (3, 7, 11), True, 0),           # shorter than the halo: one cold tile, both sequence ends inside
    (2, 200, (3, 7, 11), True, 0),          # a share per 128 columns: every tile cold, runs start inside the utterance
    (1, 264, (3, 7, 11), True, 1),          # exactly one cold window's final columns
    (1, 1500, (3, 7, 11), True, 1),         # one block: a cold tile, then four tiles on history (324 columns each)
    (3, 1201, (11, 3, 7), False, 2),        # two blocks over three utterances: shares cross utterance ends, no bias
    (2, 2000, (7, 7, 3), True, 3),          # repeated tap counts, blocks that end inside a window
    (1, 4003, (3, 7, 11), True, 5),         # odd length
    (2, 1100, (11, 11, 11), True, 1),       # three pieces per conv throughout
    (2, 1100, (3, 3, 3), True, 1),          # one piece per conv throughout
    (1, 5, (3, 7, 11), True, 0),
]


@pytest.mark.parametrize("case", STAGE32_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_mrf_stage32_vs_oracle_and_pair_launches(case, tuning):
    B, T, ks, bias, blocks = case
    rng = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = rng.randn(B, 32, T).astype(np.float32)
    ws = _stage_weights(rng, ks, bias, C=32)
    ref = _oracle_stage(x, ws, ks)
    tuning("mrf_blocks", blocks)
    X, P = _t(x), _pack(ws, ks)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    y = _native.mrf_stage_split_f16(X, P, ks, guard=guard)
    assert _rel(y, ref) <= 4e-6
    assert int(guard.item()) == 0
    if T % 4 == 0:
        assert torch.equal(y, _pairs_stage(X, ws, ks)), "one launch and the four pair launches differ"
    twin = torch.empty_like(y)
    y2 = _native.mrf_stage_split_f16(X, P, ks, out_act=twin, act_slope=0.1)
    assert torch.equal(y2, y) and _rel(twin, oo.lrelu(ref, 0.1)) <= 4e-6
    assert torch.equal(_native.mrf_stage_split_f16(X, P, ks, act_slope=0.1), twin)


def test_mrf_stage32_does_not_depend_on_blocks_or_batch(tuning):
    rng = np.random.RandomState(6)
    ks = (3, 7, 11)
    x = rng.randn(3, 32, 2600).astype(np.float32)
    ws = _stage_weights(rng, ks, C=32)
    X, P = _t(x), _pack(ws, ks)
    want = _native.mrf_stage_split_f16(X, P, ks)
    for blocks in (1, 2, 7, 61):
        tuning("mrf_blocks", blocks)
        assert torch.equal(_native.mrf_stage_split_f16(X, P, ks), want), blocks
        for b in range(3):
            assert torch.equal(_native.mrf_stage_split_f16(X[b:b + 1].contiguous(), P, ks)[0], want[b]), (blocks, b)


def test_mrf_stage32_history_survives_a_dirty_workspace(tuning):
    """The workspace's contents do not matter: a run's first tile reads zeros, later tiles what the tile before them
    wrote -- NaNs left in it by whoever had the memory before change nothing."""
    rng = np.random.RandomState(8)
    ks = (11, 7, 3)
    x = rng.randn(2, 32, 1800).astype(np.float32)
    ws = _stage_weights(rng, ks, C=32)
    X, P = _t(x), _pack(ws, ks)
    tuning("mrf_blocks", 3)
    want = _native.mrf_stage_split_f16(X, P, ks)
    n = _native.lib().fv_mrf_stage_workspace_bytes(32)
    assert n > 0 and _native.lib().fv_mrf_stage_workspace_bytes(16) == 0
    work = torch.full((n // 4,), float("nan"), device=_dev())
    out = torch.empty_like(X)
    import ctypes
    karr, darr = (ctypes.c_int * 3)(*ks), (ctypes.c_int * 3)(*DILS)
    rc = _native.lib().fv_mrf_stage_split_f16(X.data_ptr(), P.data_ptr(), out.data_ptr(), None, 2, 32, 1800, karr, darr, 0.1, 3.0,
                                              _native.POST_NONE, 1.0, None, None, None, work.data_ptr(), n, None, None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(out, want)
    # no workspace, or one that is too small: refused, nothing launched
    rc = _native.lib().fv_mrf_stage_split_f16(X.data_ptr(), P.data_ptr(), out.data_ptr(), None, 2, 32, 1800, karr, darr, 0.1, 3.0,
                                              _native.POST_NONE, 1.0, None, None, None, None, 0, None, None)
    assert rc == _native.ERR_WORKSPACE
    rc = _native.lib().fv_mrf_stage_split_f16(X.data_ptr(), P.data_ptr(), out.data_ptr(), None, 2, 32, 1800, karr, darr, 0.1, 3.0,
                                              _native.POST_NONE, 1.0, None, None, None, work.data_ptr(), 1000, None, None)
    assert rc == _native.ERR_WORKSPACE


def test_mrf_stage32_guards_and_refusals():
    rng = np.random.RandomState(12)
    ks = (3, 7, 11)
    ws = _stage_weights(rng, ks, C=32)
    P = _pack(ws, ks)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    x = rng.randn(2, 32, 900).astype(np.float32)
    big = x.copy()
    big[1, 19, 450] = 1.0e6
    y = _native.mrf_stage_split_f16(_t(big), P, ks, guard=guard)
    assert int(guard.item()) == 1 and not bool(torch.isfinite(y).all())
    guard.zero_()
    nob = _stage_weights(rng, ks, bias=False, C=32)
    _native.mrf_stage_split_f16(_t(x * np.float32(2.0 ** -14)), _pack(nob, ks), ks, guard=guard)
    assert int(guard.item()) == _native.GUARD_LOW
    guard.zero_()
    y = _native.mrf_stage_split_f16(_t(np.zeros_like(x)), _pack(nob, ks), ks, guard=guard)
    assert int(guard.item()) == 0 and float(y.abs().max()) == 0.0


def test_mrf_stage32_plan_op():
    rng = np.random.RandomState(13)
    ks = (3, 7, 11)
    x = rng.randn(2, 32, 1000).astype(np.float32)
    ws = _stage_weights(rng, ks, C=32)
    X, P = _t(x), _pack(ws, ks)
    plan = _native.Plan(32)
    plan.add_mrf_stage(_native.SLOT_IN, _native.SLOT_OUT, P, 32, ks, DILS, 0.1)
    assert torch.equal(plan.run(X), _native.mrf_stage_split_f16(X, P, ks))


@pytest.mark.parametrize("case", [(1, 37, 0), (2, 1000, 1), (1, 1531, 2), (2, 777, 3), (1, 3000, 0), (3, 640, 7)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_mrf_stage32_with_folded_output_conv(case, tuning):
    """HiFi-GAN large's last launch (hifigan.py:97-106 at 32 channels): stage, lrelu(0.01), conv_post (32 -> 1, 7 taps), tanh --
    against the oracle, and bit for bit against the stage followed by the narrow conv as a launch of its own."""
    B, T, blocks = case
    rng = np.random.RandomState(37 * T + blocks)
    ks = (3, 7, 11)
    x = rng.randn(B, 32, T).astype(np.float32)
    ws = _stage_weights(rng, ks, C=32)
    wp = (rng.randn(1, 32, 7) / np.sqrt(32 * 7)).astype(np.float32)
    bp = rng.randn(1).astype(np.float32)
    ref = np.tanh(oo.conv1d(_oracle_stage(x, ws, ks), wp, bp, pad=3, pre_slope=0.01).astype(np.float64))
    tuning("mrf_blocks", blocks)
    X, P = _t(x), _pack(ws, ks)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    y = _native.mrf_stage_split_f16(X, P, ks, fold=(_t(wp.reshape(32, 7)), _t(bp)), act_slope=0.01, post=_native.POST_TANH, guard=guard)
    assert tuple(y.shape) == (B, 1, T) and _rel(y, ref) <= 4e-6 and int(guard.item()) == 0
    stage = _native.mrf_stage_split_f16(X, P, ks)
    two = _native.conv1d_fused(stage, _native.pack_conv1d(_t(wp)), _t(bp), 1, 7, pad=3, pre_slope=0.01, post=_native.POST_TANH)
    assert torch.equal(y, two), "folded and separate output conv differ"
    # the plan op with the fold, without a bias
    plan = _native.Plan(32)
    plan.add_mrf_stage(_native.SLOT_IN, 2, P, 32, ks, DILS, 0.1)
    plan.set_pair_output_conv(_t(wp.reshape(32, 7)), None, _native.SLOT_OUT, 0.01, _native.POST_TANH)
    assert plan.num_ops() == 1
    ref0 = np.tanh(oo.conv1d(_oracle_stage(x, ws, ks), wp, None, pad=3, pre_slope=0.01).astype(np.float64))
    assert _rel(plan.run(X), ref0) <= 4e-6
