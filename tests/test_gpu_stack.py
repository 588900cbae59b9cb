"""GPU parity of the one-launch MelGAN ResidualStack (csrc/convk_kernels.hpp) through the C ABI
(fv_residual_stack_split_f16, fv_plan_add_residual_stack_split_f16) against the C oracle's convs on the same seeded
inputs -- reference model/generator/modules.py:351-382:

    y = stack[4](act(stack[2](pad(act(c))))) + skip_layer(c)

Tolerance: 4e-6 relative to the tensor's scale per stack (the generator-level bound is the north star's 1e-4); at 128
channels also bit-identity with the two-launch form (fv_conv1d_split_f16 + fv_conv1x1_2src_split_f16).
"""
import numpy as np
import pytest
import torch

from fastvocoder_amd import _native
from oracle import ops as oo

pytestmark = pytest.mark.gpu
SPLIT = _native.PAIR_SPLIT_F16


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


def _stack(rng, B, C, T, bias=True, xs=1.0, w_scale=1.0):
    x = (rng.randn(B, C, T) * xs).astype(np.float32)
    w1 = (rng.randn(C, C, 3) / np.sqrt(3 * C) * w_scale).astype(np.float32)
    w2 = (rng.randn(C, C, 1) / np.sqrt(C)).astype(np.float32)
    ws = (rng.randn(C, C, 1) / np.sqrt(C) * w_scale).astype(np.float32)
    b1 = (rng.randn(C) * xs * w_scale).astype(np.float32) if bias else None
    b2 = (rng.randn(C) * xs * w_scale).astype(np.float32) if bias else None
    bs = (rng.randn(C) * xs * w_scale).astype(np.float32) if bias else None
    return x, w1, b1, w2, b2, ws, bs


def _ref(x, w1, b1, w2, b2, ws, bs, dil, slope, pad_mode):
    h = oo.conv1d(x, w1, b1, dil=dil, pad=dil, pad_mode=pad_mode, pre_slope=slope)
    return oo.conv1d(h, w2, b2, pre_slope=slope) + oo.conv1d(x, ws, bs)


def _run(x, w1, b1, w2, b2, ws, bs, dil, slope, pad_mode, **kw):
    P = _native.pack_residual_stack_split(_t(w1), _t(w2), _t(ws))
    bo = None if b2 is None else _t(b2 + bs)
    return _native.residual_stack_split_f16(_t(x), P, _t(b1), bo, 3, dil, slope, pad_mode=pad_mode, **kw)


@pytest.fixture
def tuning():
    defaults = {"convh_blocks": 0, "convg_rows64": -1, "convh_rows64": -1, "stack_items": 1 << 20, "stack_wide": 10}
    yield _native.tuning_set
    for k, v in defaults.items():
        _native.tuning_set(k, v)


STACK_CASES = [
    # B, C, T, dil, reflect, bias
    (1, 32, 51200 // 8, 1, True, True),     # MelGAN's last stage (T = 200 frames x 256) cut to an eighth: 25 tiles of 256
    (1, 32, 700, 3, True, True),
    (2, 32, 513, 9, True, True),            # ragged last tile, two utterances
    (1, 32, 10, 9, True, False),            # one more sample than the reflected pad
    (1, 64, 25600 // 8, 1, True, True),
    (2, 64, 300, 3, True, True),
    (1, 64, 129, 9, True, True),
    (3, 64, 40, 9, False, True),            # zero padding
    (1, 128, 12800 // 8, 1, True, True),
    (2, 128, 257, 3, True, True),
    (1, 128, 130, 9, True, False),
    (1, 128, 64, 9, False, True),
    (2, 128, 1, 1, False, True),            # a single sample
    (1, 256, 1600, 1, True, True),          # MelGAN's first stage: 50 tiles of 32 columns (convk2_kernel)
    (2, 256, 257, 3, True, True),
    (1, 256, 130, 9, True, False),
    (3, 256, 33, 9, False, True),
    (1, 256, 10, 9, True, True),
]


@pytest.mark.parametrize("case", STACK_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_residual_stack_vs_oracle(case, tuning):
    B, C, T, dil, reflect, bias = case
    rng = np.random.RandomState(1000 * C + 7 * T + dil)
    m = _stack(rng, B, C, T, bias)
    mode = oo.PAD_REFLECT if reflect else oo.PAD_ZERO
    nmode = _native.PAD_REFLECT if reflect else _native.PAD_ZERO
    ref = _ref(*m, dil, 0.2, mode)
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    y = _run(*m, dil, 0.2, nmode, guard=guard)
    assert tuple(y.shape) == ref.shape and _rel(y, ref) <= 4e-6
    assert int(guard.item()) == 0
    # activated twin next to the raw output; the output stored activated (what a consumer's hoisted activation asks for)
    twin = torch.empty_like(y)
    y2 = _run(*m, dil, 0.2, nmode, out_act=twin, act_slope=0.1)
    assert torch.equal(y2, y) and _rel(twin, oo.lrelu(ref, 0.1)) <= 4e-6
    y3 = _run(*m, dil, 0.2, nmode, act_slope=0.1)
    assert torch.equal(y3, twin)
    # the graph's last stack carries the generator's final activation (Basis-MelGAN's ReLU, basis_melgan.py:99-100)
    y4 = _run(*m, dil, 0.2, nmode, post=_native.POST_RELU)
    assert torch.equal(y4, torch.clamp(y, min=0.0))
    # a few persistent blocks walking many tiles each (ring and window hand-over between tiles): the same bits
    tuning("convh_blocks", 3)
    few = _run(*m, dil, 0.2, nmode)
    tuning("convh_blocks", 0)
    assert torch.equal(few, y)
    if B > 1:                                # utterances are independent
        one = _run(m[0][1:2], *m[1:], dil, 0.2, nmode)
        assert torch.equal(one, y[1:2])
    if C == 256:                             # 32- and 64-column tiles (convk2_kernel<C, DIL, NM>; the launcher picks by size)
        for wide in (0, 1 << 20):
            tuning("stack_wide", wide)
            for blocks in (0, 2):
                tuning("convh_blocks", blocks)
                assert torch.equal(_run(*m, dil, 0.2, nmode), y)
        tuning("convh_blocks", 0)
        tuning("stack_wide", 10)


@pytest.mark.parametrize("case", [(1, 128, 1600, 1), (2, 128, 300, 3), (1, 128, 203, 9), (1, 128, 12, 9),
                                  (1, 256, 1600, 9), (2, 256, 300, 1), (1, 256, 77, 3)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_residual_stack_is_the_two_launch_form_bit_for_bit(case, tuning):
    """At 128 and 256 channels the two-launch form runs on the same arithmetic (convh_kernel / convs_kernel, then convg_kernel
    / convr_kernel): same K order, same split of the hidden tensor, same epilogue -- identical bits."""
    B, C, T, dil = case
    rng = np.random.RandomState(C + T + dil)
    x, w1, b1, w2, b2, ws, bs = _stack(rng, B, C, T)
    fused = _run(x, w1, b1, w2, b2, ws, bs, dil, 0.2, _native.PAD_REFLECT)
    tuning("stack_wide", 0)                  # (256 channels: the 64-column tile; elsewhere no effect)
    wide = _run(x, w1, b1, w2, b2, ws, bs, dil, 0.2, _native.PAD_REFLECT)
    tuning("stack_wide", 10)
    assert torch.equal(wide, fused)
    hid = _native.conv1d_split_f16([_t(x)], [_native.pack_pair(_t(w1), SPLIT)], [_t(b1)], [3], dil, pre_slope=0.2,
                                   pad_mode=_native.PAD_REFLECT)[0]
    two = _native.conv1x1_2src_split_f16(hid, _t(x), _native.pack_conv1x1_2src_split(_t(w2), _t(ws)), _t(b2 + bs), pre_slope=0.2)
    assert torch.equal(fused, two)


@pytest.mark.parametrize("C", [32, 64, 128, 256])
def test_residual_stack_at_weight_and_activation_scales(C):
    """The split-f16 domain: weights of any magnitude (row prescale at pack time), activations
    over the f16 range; the error stays within three times the fp32 chain's."""
    rng = np.random.RandomState(C)
    for xs, w_scale in ((1.0, 1.0), (1e3, 1e-3), (1e-2, 2.0 ** 12), (30.0, 2.0 ** -12), (1e-2, 1.0)):
        m = _stack(rng, 1, C, 700, True, xs=xs, w_scale=w_scale)
        ref = _ref(*m, 3, 0.2, oo.PAD_REFLECT)
        guard = torch.zeros(1, dtype=torch.int32, device=_dev())
        y = _run(*m, 3, 0.2, _native.PAD_REFLECT, guard=guard)
        assert int(guard.item()) == 0, (xs, w_scale)
        scale = float(np.abs(ref).max())
        assert float(np.abs(y.cpu().numpy() - ref).max()) <= 6e-6 * scale, (xs, w_scale)


def test_residual_stack_guards():
    rng = np.random.RandomState(5)
    m = list(_stack(rng, 1, 64, 500))
    guard = torch.zeros(1, dtype=torch.int32, device=_dev())
    m[0] = m[0].copy()
    m[0][0, 3, 17] = 3.0e5                             # an input beyond the f16 range
    _run(*m, 1, 0.2, _native.PAD_REFLECT, guard=guard)
    assert int(guard.item()) == 1
    guard.zero_()
    small = list(_stack(rng, 1, 64, 500, xs=1e-5))     # a tensor that is small as a whole: the low side (value 4)
    _run(*small, 1, 0.2, _native.PAD_REFLECT, guard=guard)
    assert int(guard.item()) == _native.GUARD_LOW
    guard.zero_()
    quiet = list(_stack(rng, 1, 64, 500))              # silence inside an ordinary signal is not "small"
    quiet[0][:, :, 100:300] = 0.0
    _run(*quiet, 1, 0.2, _native.PAD_REFLECT, guard=guard)
    assert int(guard.item()) == 0


def test_residual_stack_rejects():
    z3, z1 = torch.zeros((48, 48, 3), device=_dev()), torch.zeros((48, 48, 1), device=_dev())
    with pytest.raises(_native.NativeError, match="not built"):
        _native.pack_residual_stack_split(z3, z1, z1)
    w3, w1 = torch.zeros((64, 64, 3), device=_dev()), torch.zeros((64, 64, 1), device=_dev())
    P = _native.pack_residual_stack_split(w3, w1, w1)
    x = torch.zeros((1, 64, 9), device=_dev())
    with pytest.raises(_native.NativeError, match="reflection padding 9 needs more than 9"):
        _native.residual_stack_split_f16(x, P, None, None, 3, 9, 0.2, pad_mode=_native.PAD_REFLECT)
    with pytest.raises(_native.NativeError, match="dilation 5"):
        _native.residual_stack_split_f16(x, P, None, None, 3, 5, 0.2)
    with pytest.raises(_native.NativeError, match="pad_mode"):
        _native.residual_stack_split_f16(x, P, None, None, 3, 1, 0.2, pad_mode=_native.PAD_CAUSAL)
    with pytest.raises(_native.NativeError, match="alias"):
        _native.residual_stack_split_f16(x, P, None, None, 3, 1, 0.2, out=x)


def test_melgan_module_runs_its_stacks_fused():
    """generator/modules.py ResidualStack on the one-launch kernel (the default) against the two-launch form, through the
    module surface; the launch counts of a whole MelGAN."""
    from fastvocoder_amd.generator import modules as M
    from fastvocoder_amd.generator.melgan import MelGANGenerator
    torch.manual_seed(3)
    for C, d in ((32, 9), (64, 3), (128, 1)):
        rs = M.ResidualStack(kernel_size=3, channels=C, dilation=d).to(_dev())
        x = torch.randn(2, C, 333, device=_dev())
        fused = rs(x)
        rs.fuse_stack = False
        rs.invalidate_plans()
        plain = rs(x)
        rs.fuse_stack = True
        assert float((fused - plain).abs().max()) <= 4e-6 * max(1.0, float(plain.abs().max()))
        if C == 128:
            assert torch.equal(fused, plain)
    # 256 channels: the op carries both forms and a run picks by its size -- identical bits
    rs = M.ResidualStack(kernel_size=3, channels=256, dilation=3).to(_dev())
    for T in (500, 40000):
        x = torch.randn(1, 256, T, device=_dev())
        picked = rs(x)                                     # one launch (the default at every size)
        assert {k[0]: v[1].num_ops() for k, v in rs._fv_plans.items()}["forward"] == 1
        _native.tuning_set("stack_items", 0)               # the two launches the same op carries
        try:
            other = rs(x)
        finally:
            _native.tuning_set("stack_items", 1 << 20)
        assert torch.equal(picked, other)
    g = MelGANGenerator().to(_dev()).eval()
    mel = torch.randn(1, 80, 50, device=_dev())

    def launches():
        _native.profile_enable(True)
        g(mel)
        n = _native.profile_collect(-1)["launches"]
        _native.profile_enable(False)
        return n

    y = g(mel)
    n_fused = launches()
    M.ResidualStack.fuse_stack = False
    try:
        g.invalidate_plans()
        y2 = g(mel)
        n_plain = launches()
    finally:
        M.ResidualStack.fuse_stack = True
        g.invalidate_plans()
    assert float((y - y2).abs().max()) <= 1e-5
    assert n_plain - n_fused == 12 and n_fused <= 18      # three stacks in each of the four stages
