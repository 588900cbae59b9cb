"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the
same seeded inputs, against the committed reference goldens, and -- at the
BASELINE sizes -- through size-independent properties.

Tolerance: BASELINE.json's north_star asks for <= 1e-4 fp32 max-abs on the
generator output (|output| <= 1 after tanh); op/block-level checks whose outputs
are not O(1) use the same bound relative to the tensor's scale.
"""
import os

import numpy as np
import pytest
import torch

import fastvocoder_amd as fa
from fastvocoder_amd import _native
from fastvocoder_amd.bin.synthesize import build_generator
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict
from oracle import generators as og
from oracle import ops as oo
from oracle import torch_port
from tests import cases

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: outputs match the reference generator within 1e-4 fp32 max-abs


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def _rel(a, b):
    b = np.asarray(b)
    return _err(a, b) / max(1.0, float(np.abs(b).max()))


def _model(name, cfg, seed):
    m = build_generator(name, cfg)
    sd = seeded_state_dict(name, cfg, seed=seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(_dev()).eval(), sd


def test_native_library_is_loaded():
    L = _native.lib()
    assert L.fv_version() == _native.ABI_VERSION
    # the binary is THIS tree's: its embedded build id is the hash of the sources next to it
    assert L.fv_build_id().decode() == _native.source_hash()
    with open("/proc/self/maps") as f:
        assert "libfastvocoder_hip.so" in f.read()


def test_measured_matrix_peak_hook_is_sane():
    """fv_profile_mfma_f16_rate (bench.py's `roofline.peak_measured`): a dense stream of v_mfma_f32_16x16x32_f16 from registers
    cannot beat the nominal 2.5 PFLOP/s and, on an MI355X, does not fall below a third of it."""
    tf = _native.profile_mfma_f16_rate(launches=8, iters=4000)
    assert 800.0 < tf < 2600.0, tf


# ---------------------------------------------------------------------------
# operators vs the C oracle
# ---------------------------------------------------------------------------
CONV_CASES = [
    # B, Cin, Cout, T, k, dil, pad, pad_mode, pre_slope
    (2, 16, 16, 257, 3, 1, 1, 0, 0.1),
    (1, 16, 16, 1000, 11, 5, 25, 0, 0.1),
    (2, 32, 32, 301, 7, 3, 9, 0, 0.1),
    (1, 64, 64, 500, 11, 1, 5, 0, 0.1),
    (1, 128, 128, 130, 3, 5, 5, 0, 0.1),
    (2, 80, 64, 64, 7, 1, 3, 0, 1.0),       # conv_pre shape class
    (1, 80, 96, 33, 7, 1, 3, 1, 1.0),       # reflect first layer, Cout not a multiple of 64
    (2, 32, 32, 200, 3, 9, 9, 1, 0.2),      # ResidualStack dilated conv, reflect
    (1, 64, 64, 77, 1, 1, 0, 0, 0.2),       # 1x1
    (2, 16, 1, 999, 7, 1, 3, 0, 0.01),      # conv_post
    (1, 32, 1, 130, 7, 1, 3, 1, 0.2),       # LastLayer
    (3, 64, 4, 250, 7, 1, 3, 0, 0.01),      # multiband conv_post
    (1, 8, 8, 100, 5, 2, 4, 0, 0.1),        # M = 8 rows (padded tile), generic tap count
    (1, 4, 4, 90, 3, 1, 1, 1, 0.2),         # narrow kernel with Cin = 4
    (1, 24, 40, 75, 3, 2, 2, 0, 0.0),       # odd channel counts, ReLU pre-activation
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv1d_fused_vs_oracle(case):
    B, Cin, Cout, T, k, dil, pad, mode, slope = case
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x = rng.randn(B, Cin, T).astype(np.float32)
    w = (rng.randn(Cout, Cin, k) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    ref = oo.conv1d(x, w, b, dil=dil, pad=pad, pad_mode=mode, pre_slope=slope)
    dev = _dev()
    packed = _native.pack_conv1d(torch.from_numpy(w).to(dev))
    y = _native.conv1d_fused(torch.from_numpy(x).to(dev), packed, torch.from_numpy(b).to(dev), Cout, k,
                             dil=dil, pad=pad, pad_mode=mode, pre_slope=slope)
    assert _rel(y, ref) <= 2e-5
    # fused epilogue: residual + running sum + true division + tanh
    res = rng.randn(*ref.shape).astype(np.float32)
    acc = rng.randn(*ref.shape).astype(np.float32)
    ref2 = np.tanh(((acc + (ref + res)) / np.float32(3.0)).astype(np.float64)).astype(np.float32)
    y2 = _native.conv1d_fused(torch.from_numpy(x).to(dev), packed, torch.from_numpy(b).to(dev), Cout, k,
                              dil=dil, pad=pad, pad_mode=mode, pre_slope=slope,
                              res=torch.from_numpy(res).to(dev), acc_in=torch.from_numpy(acc).to(dev),
                              out_div=3.0, post=_native.POST_TANH)
    assert _err(y2, ref2) <= 2e-5


def test_random_conv_shapes_vs_oracle():
    """Seeded sweep over shapes no generator uses: odd channel counts (row tiles that stick out
    of the output, channel chunks that do not divide Cin), every tap count / dilation / padding
    combination the launcher dispatches differently, pre-activation on and off, with and
    without residual -- each against the C oracle."""
    rng = np.random.RandomState(20260927)
    dev = _dev()
    for n in range(60):
        B = int(rng.choice([1, 2, 3]))
        Cin, Cout = int(rng.randint(1, 72)), int(rng.randint(5, 72))
        k = int(rng.choice([1, 2, 3, 5, 7, 11]))
        dil = int(rng.choice([1, 2, 3, 5])) if k > 1 else 1
        T = int(rng.randint(max(2, dil * (k - 1) + 1), 700))
        mode = int(rng.choice([0, 1]))
        pad = int(rng.choice([0, dil * (k - 1) // 2, dil * (k - 1)]))
        if mode == 1 and pad >= T:
            pad = 0
        if T + 2 * pad - dil * (k - 1) <= 0:
            continue
        slope = float(rng.choice([1.0, 0.1, 0.0]))
        x = rng.randn(B, Cin, T).astype(np.float32)
        w = (rng.randn(Cout, Cin, k) / np.sqrt(Cin * k)).astype(np.float32)
        b = rng.randn(Cout).astype(np.float32) if rng.rand() < 0.8 else None
        ref = oo.conv1d(x, w, b, dil=dil, pad=pad, pad_mode=mode, pre_slope=slope)
        res = rng.randn(*ref.shape).astype(np.float32) if rng.rand() < 0.5 else None
        want = ref + res if res is not None else ref
        t = lambda a: None if a is None else torch.from_numpy(a).to(dev)  # noqa: E731
        y = _native.conv1d_fused(t(x), _native.pack_conv1d(t(w)), t(b), Cout, k, dil=dil, pad=pad,
                                 pad_mode=mode, pre_slope=slope, res=t(res))
        assert _rel(y, want) <= 2e-5, (n, B, Cin, Cout, T, k, dil, pad, mode, slope)


def test_random_transposed_conv_shapes_vs_oracle():
    """Same for ConvTranspose1d: strides 2..10, kernels from s to 3s, paddings / output paddings
    that make Tout not a multiple of the stride (the overflow rows of the last column)."""
    rng = np.random.RandomState(4242)
    dev = _dev()
    for n in range(40):
        B = int(rng.choice([1, 2]))
        Cin, Cout = int(rng.randint(1, 48)), int(rng.randint(1, 40))
        s_ = int(rng.randint(2, 11))
        k = int(rng.randint(s_, 3 * s_ + 1))
        if n % 3 == 0:      # the phase-major form: Cout a multiple of 32, kernel a whole number of strides
            Cout, k = int(rng.choice([32, 64, 96])), int(rng.choice([2, 3])) * s_
        p_ = int(rng.randint(0, min(k // 2, s_) + 1))
        op = int(rng.randint(0, s_))
        T = int(rng.randint(1, 200))
        if (T - 1) * s_ - 2 * p_ + k + op <= 0:
            continue
        slope = float(rng.choice([1.0, 0.2]))
        x = rng.randn(B, Cin, T).astype(np.float32)
        w = (rng.randn(Cin, Cout, k) / np.sqrt(max(1, Cin * k // s_))).astype(np.float32)
        b = rng.randn(Cout).astype(np.float32)
        ref = oo.conv_transpose1d(x, w, b, s_, p_, op, pre_slope=slope)
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        y = _native.conv_transpose1d_fused(t(x), _native.pack_conv_transpose1d(t(w), s_, p_), t(b), Cout, k,
                                           s_, p_, op, pre_slope=slope)
        assert tuple(y.shape) == ref.shape, (n, ref.shape)
        assert _rel(y, ref) <= 2e-5, (n, B, Cin, Cout, T, k, s_, p_, op)


CONVT_CASES = [
    # B, Cin, Cout, T, k, stride   (pad = s//2 + s%2, out_pad = s%2, like the generators)
    (1, 64, 32, 50, 16, 8), (2, 32, 16, 41, 10, 5), (1, 32, 16, 100, 6, 3), (2, 32, 16, 77, 4, 2),
    (1, 64, 32, 30, 20, 10), (1, 32, 16, 55, 12, 6),        # MB light / MelGAN
    (1, 32, 16, 30, 16, 10), (1, 16, 8, 40, 16, 6),         # MB large: k < 2s and k > 2s
    (1, 32, 32, 60, 8, 4),                                  # Basis
    (1, 128, 64, 9, 16, 8),                                 # very short input
]


@pytest.mark.parametrize("case", CONVT_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_transpose1d_fused_vs_oracle(case):
    B, Cin, Cout, T, k, s = case
    p, op = s // 2 + s % 2, s % 2
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x = rng.randn(B, Cin, T).astype(np.float32)
    w = (rng.randn(Cin, Cout, k) / np.sqrt(Cin * k / s)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    ref = oo.conv_transpose1d(x, w, b, s, p, op, pre_slope=0.1)
    dev = _dev()
    packed = _native.pack_conv_transpose1d(torch.from_numpy(w).to(dev), s, p)
    y = _native.conv_transpose1d_fused(torch.from_numpy(x).to(dev), packed, torch.from_numpy(b).to(dev),
                                       Cout, k, s, p, op, pre_slope=0.1)
    assert _rel(y, ref) <= 2e-5


UPCONV_CASES = [
    # B, Cin, Cout, T, k, rate, pad   (HiFi-GAN: pad = k//2; MelGAN/Basis: k = 2*rate+1, pad = rate)
    (1, 64, 32, 50, 16, 8, 8), (2, 32, 16, 41, 10, 5, 5), (1, 32, 16, 33, 7, 3, 3), (2, 16, 8, 40, 4, 2, 2),
    (1, 32, 32, 60, 9, 4, 4), (1, 32, 16, 25, 21, 10, 10),
    (1, 16, 16, 12, 5, 3, 0), (1, 16, 8, 9, 3, 4, 1),         # no padding; kernel shorter than rate
]


@pytest.mark.parametrize("case", UPCONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_upsample_conv1d_fused_vs_oracle(case):
    """UpsampleLayer (nearest repeat x rate + Conv1d) as summed-phase weights vs the literal
    repeat-then-convolve oracle (oracle/generators.py upsample_layer)."""
    B, Cin, Cout, T, k, u, p = case
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x = rng.randn(B, Cin, T).astype(np.float32)
    w = (rng.randn(Cout, Cin, k) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    ref = oo.conv1d(np.repeat(x, u, axis=2), w, b, dil=1, pad=p, pre_slope=0.1)
    dev = _dev()
    packed = _native.pack_upsample_conv1d(torch.from_numpy(w).to(dev), u, p)
    y = _native.upsample_conv1d_fused(torch.from_numpy(x).to(dev), packed, torch.from_numpy(b).to(dev),
                                      Cout, k, u, p, pre_slope=0.1)
    assert tuple(y.shape) == ref.shape
    assert _rel(y, ref) <= 2e-5


def test_upsample_layer_module_standalone():
    from fastvocoder_amd.generator.modules import UpsampleLayer
    torch.manual_seed(3)
    layer = UpsampleLayer(24, 12, upsample_rate=5, kernel_size=11, stride=1, padding=5).to(_dev())
    x = torch.randn(2, 24, 37)
    ref = oo.conv1d(np.repeat(x.numpy(), 5, axis=2), layer.conv.weight.detach().cpu().numpy(),
                    layer.conv.bias.detach().cpu().numpy(), dil=1, pad=5)
    assert _rel(layer(x.to(_dev())), ref) <= 2e-5


@pytest.mark.parametrize("case", [(2, 16, 16, 50, 3, 9, 1), (1, 32, 16, 64, 3, 1, 0), (1, 8, 8, 40, 4, 2, 1),
                                  (1, 64, 64, 300, 3, 3, 1)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_causal_conv_vs_oracle(case):
    """CausalConv1d (modules.py:273-294): pad (k-1)*dil on both sides, valid conv, first T kept."""
    B, Cin, Cout, T, k, dil, mode = case
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x = rng.randn(B, Cin, T).astype(np.float32)
    w = (rng.randn(Cout, Cin, k) / np.sqrt(Cin * k)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    pad = (k - 1) * dil
    ref = oo.conv1d(x, w, b, dil=dil, pad=pad, pad_mode=mode, pre_slope=0.2)[:, :, :T]
    dev = _dev()
    packed = _native.pack_conv1d(torch.from_numpy(w).to(dev))
    y = _native.conv1d_fused(torch.from_numpy(x).to(dev), packed, torch.from_numpy(b).to(dev), Cout, k,
                             dil=dil, pad=pad, pad_mode=mode | _native.PAD_CAUSAL, pre_slope=0.2)
    assert tuple(y.shape) == (B, Cout, T)
    assert _rel(y, ref) <= 2e-5
    with pytest.raises(_native.NativeError, match="causal"):
        _native.conv1d_fused(torch.from_numpy(x).to(dev), packed, None, Cout, k, dil=dil, pad=0,
                             pad_mode=_native.PAD_CAUSAL)


def test_batchnorm_fold_and_last_linear():
    """fv_fold_batchnorm_conv vs the literal eval-mode BatchNorm + conv, then the LastLinear head."""
    from fastvocoder_amd.generator.modules import LastLinear
    torch.manual_seed(5)
    head = LastLinear(24, 12)
    for bn in (head.bn_1, head.bn_2):
        bn.weight.data.uniform_(0.7, 1.3)
        bn.bias.data.uniform_(-0.2, 0.2)
        bn.running_mean.uniform_(-0.3, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    head = head.to(_dev())
    x = torch.randn(2, 24, 50)
    sd = {"h." + k: v for k, v in head.state_dict().items()}
    ref = og.last_linear(x.numpy(), sd, "h")
    with pytest.raises(_native.NativeError, match="eval"):
        head(x.to(_dev()))                      # train mode: batch statistics are training-only
    head.eval()
    assert _rel(head(x.to(_dev())), ref) <= 2e-5
    w2, b2 = _native.fold_batchnorm_conv(head.linear_2.weight, head.linear_2.bias, head.bn_2)
    bn = og.batchnorm_eval(x.numpy(), sd, "h.bn_2")
    lit = oo.conv1d(bn, head.linear_2.weight.detach().cpu().numpy(), head.linear_2.bias.detach().cpu().numpy())
    fold = oo.conv1d(x.numpy(), w2.cpu().numpy(), b2.cpu().numpy())
    assert _rel(torch.from_numpy(fold), lit) <= 2e-5


@pytest.mark.parametrize("case", [(2, 32, 32, 32, 200), (1, 64, 64, 64, 1000), (1, 16, 48, 32, 333),
                                  (3, 256, 256, 256, 61), (1, 8, 8, 16, 7)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_two_source_1x1_conv_vs_oracle(case):
    """fv_conv1d_2src_fused: W1 x + W2 x2 + (b1 + b2) [+ res] as one GEMM over the concatenated K
    range (ResidualStack's 1x1 + skip 1x1, modules.py:362-366,382) vs the two convs of the oracle;
    aligned and unaligned T (the latter takes the per-element staging path)."""
    B, C1, C2, Cout, T = case
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x1 = rng.randn(B, C1, T).astype(np.float32)
    x2 = rng.randn(B, C2, T).astype(np.float32)
    w1 = (rng.randn(Cout, C1, 1) / np.sqrt(C1)).astype(np.float32)
    w2 = (rng.randn(Cout, C2, 1) / np.sqrt(C2)).astype(np.float32)
    b1, b2 = rng.randn(Cout).astype(np.float32), rng.randn(Cout).astype(np.float32)
    res = rng.randn(B, Cout, T).astype(np.float32)
    ref = oo.conv1d(x1, w1, b1) + oo.conv1d(x2, w2, b2)
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    packed = _native.pack_conv1d(torch.cat([t(w1), t(w2)], dim=1).contiguous())
    y = _native.conv1d_2src_fused(t(x1), t(x2), packed, t(b1 + b2), Cout)
    assert _rel(y, ref) <= 2e-5
    y = _native.conv1d_2src_fused(t(x1), t(x2), packed, None, Cout, res=t(res), post=_native.POST_RELU)
    assert _rel(y, np.maximum(ref - (b1 + b2)[None, :, None] + res, 0)) <= 2e-5


def _plans(module):
    """name -> native plan of a module's plan cache (keyed by (name, policy))."""
    return {name: plan for (name, _), (_, plan) in module._fv_plans.items()}


def test_fused_skip_equals_unfused_residual_stack():
    """The K-concatenated ResidualStack tail vs the three-launch form (fuse_skip = False)."""
    from fastvocoder_amd.generator import modules
    torch.manual_seed(1)
    rs = modules.ResidualStack(kernel_size=3, channels=64, dilation=3).to(_dev())
    x = torch.randn(2, 64, 500, device=_dev())
    one = rs(x).cpu().numpy()
    assert _plans(rs)["forward"].num_ops() == 1          # the whole stack as one launch (csrc/convk_kernels.hpp)
    rs.fuse_stack = False
    rs.invalidate_plans()
    fused = rs(x).cpu().numpy()
    assert _plans(rs)["forward"].num_ops() == 2
    assert np.abs(one - fused).max() <= 1e-5 * max(1.0, np.abs(fused).max())
    rs.fuse_skip = False
    rs.invalidate_plans()
    plain = rs(x).cpu().numpy()
    assert _plans(rs)["forward"].num_ops() == 3
    assert np.abs(fused - plain).max() <= 1e-5 * max(1.0, np.abs(plain).max())


def test_activated_twin_outputs():
    """Activation hoisting at operator level: y raw + y_act = lrelu(y, s) in one launch, and the
    in-place form; a consumer reading y_act with pre_slope = 1 equals reading y with pre_slope = s."""
    dev = _dev()
    rng = np.random.RandomState(12)
    x = torch.from_numpy(rng.randn(2, 32, 500).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.randn(32, 32, 7) / 15).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.randn(32).astype(np.float32)).to(dev)
    packed = _native.pack_conv1d(w)
    y = _native.conv1d_fused(x, packed, b, 32, 7, dil=3, pad=9)
    y_act = torch.empty_like(y)
    y2 = _native.conv1d_fused(x, packed, b, 32, 7, dil=3, pad=9, out_act=y_act, act_slope=0.1)
    assert torch.equal(y2, y)
    assert torch.equal(y_act, torch.where(y >= 0, y, y * 0.1))
    y3 = _native.conv1d_fused(x, packed, b, 32, 7, dil=3, pad=9, act_slope=0.1)     # in place
    assert torch.equal(y3, y_act)
    a = _native.conv1d_fused(y, packed, b, 32, 7, dil=1, pad=3, pre_slope=0.1)       # read-time activation
    c = _native.conv1d_fused(y_act, packed, b, 32, 7, dil=1, pad=3, pre_slope=1.0)   # hoisted
    assert float((a - c).abs().max()) <= 1e-5 * float(a.abs().max())
    # transposed conv with a twin
    wt = torch.from_numpy((rng.randn(32, 16, 10) / 8).astype(np.float32)).to(dev)
    pt = _native.pack_conv_transpose1d(wt, 5, 3)
    u = _native.conv_transpose1d_fused(x, pt, None, 16, 10, 5, 3, 1)
    u_act = torch.empty_like(u)
    u2 = _native.conv_transpose1d_fused(x, pt, None, 16, 10, 5, 3, 1, out_act=u_act, act_slope=0.2)
    assert torch.equal(u2, u) and torch.equal(u_act, torch.where(u >= 0, u, u * 0.2))


@pytest.mark.parametrize("tag", ["hifigan_s", "mb_s", "hifigan_up"])
def test_mrf_sum3_kernel_forced_on_small_configs(golden_dir, tag):
    """fv_plan_add_conv1d_sum3 picks its one-launch kernel only for layers with >= 800 tiles; forcing
    it (tuning switch sum3_min = 1) on the shrunken configs -- 64/32/16/8-channel stages, ragged row tiles,
    unaligned lengths -- must still match the reference goldens; channel counts <= 4 keep the
    two-launch form."""
    _native.tuning_set("sum3_min", 1)
    try:
        name, cfg = next((n, c) for t, n, c in cases.SMALL if t == tag)
        g = np.load(os.path.join(golden_dir, f"small_{tag}.npz"))
        m, sd = _model(name, cfg, seed=7)
        m.fuse_pairs = False                        # the conv-by-conv path is where sum3 lives
        y = m.inference(seeded_mel(cases.SMALL_T, seed=5))
        assert _err(y, g["inference"]) <= TOL
        f = m(torch.from_numpy(seeded_mel(cases.SMALL_T, seed=6, batch=cases.SMALL_B)).to(_dev()))
        assert _err(f, g["forward"]) <= TOL
    finally:
        _native.tuning_set("sum3_min", 800)


@pytest.mark.parametrize("prec,pair", [("f32", True), ("f32", False), ("split", False)])
def test_arithmetic_policies_agree(prec, pair):
    """precision = "f32" (exact-fp32 MFMA on every stage) and fuse_pairs = False (no fused ResBlock pairs: the
    conv-by-conv path) give the default path's result up to fp32 summation noise."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    x = torch.from_numpy(seeded_mel(200, seed=23, batch=2)).to(_dev())
    ref_model, _ = _model("hifigan", cfg, seed=0)
    with torch.no_grad():
        ref = ref_model(x).cpu().numpy()
    m, _ = _model("hifigan", cfg, seed=0)
    m.precision, m.fuse_pairs = prec, pair
    with torch.no_grad():
        got = m(x).cpu().numpy()
    assert np.abs(got - ref).max() <= 4e-6
    ops = _plans(m)
    assert len(ops) == 1 and not m.check_range()


def test_causal_conv_transpose_vs_reference_fixture(golden_dir):
    """CausalConvTranspose1d (reference modules.py:297-317) against outputs of the reference module itself."""
    from fastvocoder_amd.generator import modules as M
    g = np.load(os.path.join(golden_dir, "causal_convt.npz"))
    for tag in "abc":
        cin, cout, k, s = (int(v) for v in g[f"{tag}_shape"])
        m = M.CausalConvTranspose1d(cin, cout, k, s).to(_dev())
        with torch.no_grad():
            m.deconv.weight.copy_(torch.from_numpy(g[f"{tag}_weight"]))
            m.deconv.bias.copy_(torch.from_numpy(g[f"{tag}_bias"]))
        y = m(torch.from_numpy(g[f"{tag}_x"]).to(_dev()))
        ref = g[f"{tag}_out"]
        assert tuple(y.shape) == ref.shape and _rel(y, ref) <= 2e-5
        assert sorted(m.state_dict()) == ["deconv.bias", "deconv.weight"]


def test_conv_transpose_no_padding_is_overlap_add():
    """ConvTranspose1d(Cout=1, k=L, stride=L/2, pad=0) == linear + overlap_and_add."""
    rng = np.random.RandomState(5)
    W = rng.uniform(-0.2, 0.2, size=(30, 48)).astype(np.float32)
    wt = np.abs(rng.randn(2, 48, 37)).astype(np.float32)
    ref = oo.basis_ola(wt, W, 15)
    dev = _dev()
    w = torch.from_numpy(W).to(dev).t().contiguous().view(48, 1, 30)
    packed = _native.pack_conv_transpose1d(w, 15, 0)
    y = _native.conv_transpose1d_fused(torch.from_numpy(wt).to(dev), packed, None, 1, 30, 15, 0, 0)
    assert _rel(y[:, 0, :], ref) <= 2e-5


def test_basis_ola_and_generator_run_entries():
    """SURVEY.md section 8(b)'s named C entries.  fv_basis_ola (+ fv_pack_basis of nn.Linear's W [L, C]) against the
    oracle's linear + overlap_and_add and against the reference's own BasisSignalLayer output (blocks.npz `basis_out`);
    fv_generator_run = fv_plan_run on a built plan, same bits as the module's forward."""
    import ctypes
    rng = np.random.RandomState(5)
    W = rng.uniform(-0.2, 0.2, size=(30, 48)).astype(np.float32)
    wt = np.abs(rng.randn(2, 48, 37)).astype(np.float32)
    dev = _dev()
    y = _native.basis_ola(torch.from_numpy(wt).to(dev), _native.pack_basis(torch.from_numpy(W).to(dev)), 30)
    assert tuple(y.shape) == (2, 1, 36 * 15 + 30) and _rel(y[:, 0, :], oo.basis_ola(wt, W, 15)) <= 2e-5
    g = np.load(os.path.join(cases.ROOT, "tests", "golden", "blocks.npz"))
    bw = torch.from_numpy(g["basis_weight"]).to(dev).transpose(1, 2).contiguous()       # [B, F, C] -> [B, C, F]
    y = _native.basis_ola(bw, _native.pack_basis(torch.from_numpy(g["basis_W"]).to(dev)), 30)
    assert _rel(y[:, 0, :], g["basis_out"]) <= 2e-5
    with pytest.raises(_native.NativeError, match="even"):
        _native.pack_basis(torch.zeros((7, 16), device=dev))
    # fv_generator_run: the plan of a small HiFi-GAN, called through the C entry with a caller-owned workspace
    tag, name, cfg = next(c for c in cases.SMALL if c[0] == "hifigan_s")
    m, _ = _model(name, cfg, seed=7)
    x = torch.from_numpy(np.ascontiguousarray(seeded_mel(cases.SMALL_T, seed=5).T[None])).to(dev)
    with torch.no_grad():
        want = m(x)
        plan = m._trunk_plan(x.shape[2])
    L = _native.lib()
    ws_bytes = L.fv_plan_workspace_bytes(plan._h, 1, x.shape[2])
    ws = torch.empty(max(ws_bytes, 4) // 4 + 64, dtype=torch.float32, device=dev)
    out = torch.empty_like(want).unsqueeze(1).contiguous()
    torch.cuda.synchronize()
    _native.check(L.fv_generator_run(plan._h, 1, x.shape[2], ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), ws_bytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(out[:, 0, :], want)


def test_fold_weight_norm_vs_torch():
    rng = np.random.RandomState(9)
    for shape in [(64, 32, 7), (128, 64, 16), (5, 3, 1)]:
        v = torch.from_numpy(rng.randn(*shape).astype(np.float32))
        g = torch.from_numpy((rng.rand(shape[0], 1, 1) + 0.5).astype(np.float32))
        ref = torch._weight_norm(v, g, 0)
        w = _native.fold_weight_norm(v.to(_dev()), g.to(_dev()))
        assert _rel(w, ref.numpy()) <= 1e-6


def test_pqmf_synthesis_vs_golden_and_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    pq = fa.PQMF().to(_dev())
    assert _err(pq.synthesis_filter, g["pqmf_synthesis_filter"]) == 0.0
    assert _err(pq.analysis_filter, g["pqmf_analysis_filter"]) == 0.0
    y = pq.synthesis(torch.from_numpy(g["pqmf_sub"]).to(_dev()))
    assert _err(y, g["pqmf_synth_out"]) <= 1e-5
    # odd sizes vs the oracle
    rng = np.random.RandomState(2)
    sub = rng.uniform(-1, 1, size=(3, 4, 333)).astype(np.float32)
    ref = oo.pqmf_synthesis(sub, g["pqmf_synthesis_filter"][0])
    y = pq.synthesis(torch.from_numpy(sub).to(_dev()))
    assert _err(y[:, 0, :], ref) <= 1e-5
    # analysis (oracle) -> synthesis (GPU) reconstructs the interior
    a = oo.pqmf_analysis(g["pqmf_wav"][:, 0, :], g["pqmf_analysis_filter"][:, 0, :])
    r = pq.synthesis(torch.from_numpy(a).to(_dev())).cpu().numpy()
    assert np.abs(r[0, 0, 200:-200] - g["pqmf_wav"][0, 0, 200:-200]).max() < 2e-3


def test_pqmf_analysis_vs_golden_and_roundtrip(golden_dir):
    """PQMF.analysis (pqmf.py:108-119) on the GPU vs the reference's output, odd lengths vs the
    oracle, and the analysis -> synthesis known-answer check fully on the GPU."""
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    pq = fa.PQMF().to(_dev())
    wav = torch.from_numpy(g["pqmf_wav"]).to(_dev())
    a = pq.analysis(wav)
    assert _err(a, g["pqmf_analysis_out"]) <= 1e-5
    assert _err(pq.synthesis(a), g["pqmf_roundtrip"]) <= 1e-5
    r = pq.synthesis(a).cpu().numpy()
    assert np.abs(r[0, 0, 200:-200] - g["pqmf_wav"][0, 0, 200:-200]).max() < 2e-3
    rng = np.random.RandomState(4)
    for T in (4, 63, 1001, 4098):
        x = rng.uniform(-1, 1, size=(2, 1, T)).astype(np.float32)
        ref = oo.pqmf_analysis(x[:, 0, :], g["pqmf_analysis_filter"][:, 0, :])
        assert _err(pq.analysis(torch.from_numpy(x).to(_dev())), ref) <= 1e-5
    with pytest.raises(_native.NativeError):
        pq.analysis(wav[:, 0, :])


# ---------------------------------------------------------------------------
# blocks vs the reference goldens
# ---------------------------------------------------------------------------
def _fill(mod, flat):
    off = 0
    with torch.no_grad():
        for p in mod.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(flat[off:off + n].reshape(tuple(p.shape))))
            off += n
    assert off == flat.size
    return mod.to(_dev())


def test_blocks_vs_reference_goldens(golden_dir):
    from fastvocoder_amd.generator import modules as M
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    x = torch.from_numpy(g["x16"]).to(_dev())
    for k in (3, 7, 11):
        rb = _fill(M.ResBlock1(16, k, (1, 3, 5)), g[f"rb1_k{k}_params"])
        assert _rel(rb(x), g[f"rb1_k{k}_out"]) <= 2e-5
    rb = _fill(M.ResBlock2(16, 5, (1, 3)), g["rb2_params"])
    assert _rel(rb(x), g["rb2_out"]) <= 2e-5
    for d in (1, 3, 9):
        rs = _fill(M.ResidualStack(kernel_size=3, channels=16, dilation=d), g[f"rs_d{d}_params"])
        assert _rel(rs(x), g[f"rs_d{d}_out"]) <= 2e-5
    ll = _fill(M.LastLayer(16, 1, "LeakyReLU", {"negative_slope": 0.2}, "ReflectionPad1d", 7, {}, True),
               g["last_params"])
    assert _rel(ll(x), g["last_out"]) <= 2e-5
    bs = M.BasisSignalLayer(torch.from_numpy(g["basis_W"]), L=30).to(_dev())
    assert _rel(bs(torch.from_numpy(g["basis_weight"]).to(_dev())), g["basis_out"]) <= 2e-5


# ---------------------------------------------------------------------------
# whole generators: shrunken configs (full tensors) and the shipped yamls
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("tag,name,cfg", cases.SMALL, ids=[c[0] for c in cases.SMALL])
def test_small_configs_vs_reference_golden_and_oracle(golden_dir, tag, name, cfg):
    g = np.load(os.path.join(golden_dir, f"small_{tag}.npz"))
    m, sd = _model(name, cfg, seed=7)
    mel = seeded_mel(cases.SMALL_T, seed=5)
    melb = seeded_mel(cases.SMALL_T, seed=6, batch=cases.SMALL_B)
    with torch.no_grad():
        y = m.inference(mel)
        f = m(torch.from_numpy(melb))
    assert _err(y, g["inference"]) <= TOL
    assert _err(y, og.INFERENCE[name](mel, sd, cfg)) <= TOL
    if name == "basis-melgan":
        assert _err(f[0], g["forward_src"]) <= TOL
        assert _rel(f[1], g["forward_w"]) <= TOL
    else:
        assert _err(f, g["forward"]) <= TOL
    # weight-norm lifecycle: removing it must not change the function
    m.remove_weight_norm()
    assert not any(k.endswith("weight_g") for k in m.state_dict())
    with torch.no_grad():
        y2 = m.inference(mel)
    assert _err(y2, g["inference"]) <= TOL
    # ... and re-applying it keeps the state_dict layout of a training checkpoint
    m.apply_weight_norm()
    if cfg.get("use_weight_norm", True):
        assert set(m.state_dict().keys()) == set(sd.keys())
    with torch.no_grad():
        assert _err(m.inference(mel), g["inference"]) <= TOL


@pytest.mark.parametrize("tag,name,path", cases.SHIPPED, ids=[c[0] for c in cases.SHIPPED])
def test_shipped_configs_vs_reference_golden(golden_dir, tag, name, path):
    g = np.load(os.path.join(golden_dir, f"full_{tag}.npz"))
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    with torch.no_grad():
        y = m.inference(seeded_mel(cases.FULL_T, seed=0))
        assert _err(y, g["inference_T64"]) <= TOL
        f = m(torch.from_numpy(seeded_mel(16, seed=3, batch=2)))
        if name == "basis-melgan":
            assert _err(f[0], g["forward_T16_src"]) <= TOL
            assert _rel(f[1], g["forward_T16_w"]) <= TOL
        else:
            assert _err(f, g["forward_T16"]) <= TOL
        # benchmark length (T=1000): strided samples and float64 sums of the reference
        y = m.inference(seeded_mel(cases.STATS_T, seed=1)).double().cpu().numpy().reshape(-1)
    assert y.size == int(g["T1000_n"])
    assert np.abs(y[g["T1000_idx"]] - g["T1000_samples"]).max() <= TOL
    # distance to the float64 run of the reference: same order as the reference's own fp32 noise
    assert np.abs(y[g["T1000_idx"]] - g["T1000_samples64"]).max() <= TOL
    assert abs(y.sum() - float(g["T1000_sum"])) <= TOL * y.size * 0.05
    assert abs(np.abs(y).sum() - float(g["T1000_abssum"])) <= TOL * y.size * 0.05


@pytest.mark.parametrize("tag,name,path", cases.SHIPPED, ids=[c[0] for c in cases.SHIPPED])
def test_shipped_configs_vs_aten_port_full_length(tag, name, path):
    """Full tensor at T=250 against the validated ATen port on the host."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=4)
    mel = seeded_mel(250, seed=8)
    with torch.no_grad():
        y = m.inference(mel)
    ref = torch_port.inference(name, mel, sd, cfg).numpy()
    assert _err(y, ref) <= TOL


@pytest.mark.parametrize("name,path,T", [("hifigan", "conf/hifigan/light.yaml", 100), ("hifigan", "conf/hifigan/large.yaml", 40),
                                         ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 64)],
                         ids=["hifigan_light", "hifigan_large", "mb_light"])
def test_mrf_merge_inside_the_upsampler_gives_the_same_bits(name, path, T):
    """hifigan.py:99-103 -- xs = r0; xs += r1; xs += r2; x = xs / 3 -- at the end of every MRF stage but the last is formed by
    the split-f16 upsampler BEHIND the stage while it loads its window (fv_plan_set_input_merge): the stage's last pair
    position is one three-member launch instead of two launches.  Same association, same division: bit-identical to the
    plan that merges in the stage's own last launch (`merge_in_upsampler = False`), single utterance and a batch, with
    fewer launches."""
    cfg = cases.load_conf(path)
    m, _ = _model(name, cfg, seed=3)
    x = torch.from_numpy(seeded_mel(T, seed=12, batch=3)).to(_dev())
    def launches():
        torch.cuda.synchronize()
        _native.profile_collect(-1)
        _native.profile_enable(True)
        y = m(x).clone()
        torch.cuda.synchronize()
        _native.profile_enable(False)
        return y, int(_native.profile_collect(-1)["launches"])
    with torch.no_grad():
        m(x)
        a, n_merged = launches()
        m.merge_in_upsampler = False
        m(x)
        b, n_plain = launches()
        one = m(x[1:2].contiguous()).clone()
        m.merge_in_upsampler = True
        one_m = m(x[1:2].contiguous()).clone()
    assert torch.equal(a, b) and torch.equal(one, a[1:2]) and torch.equal(one_m, one)
    assert n_merged < n_plain, (n_merged, n_plain)     # (one launch less per stage whose upsampler is a split-f16 one)
    assert not m.check_range()


def test_one_launch_stage_gives_the_same_bits():
    """HiFi-GAN light's 32- and 16-channel MRF stages (hifigan.py:97-106) as ONE launch each -- nine fused pairs, the mean,
    and at 16 channels conv_post + tanh (csrc/mrfw_kernels.hpp, mrfh_kernels.hpp) -- against the pair launches of round 4
    (`fuse_stage = False`; `(16,)`: only the last stage fused): the same arithmetic per element, so identical bits; a single
    utterance at the headline length, a ragged batch, `inference` and the bias-removal plan (whose conv_post stays a launch
    of its own), with five launches fewer per forward."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    one, _ = _model("hifigan", cfg, seed=3)
    one.fuse_stage = (16, 32)
    four, _ = _model("hifigan", cfg, seed=3)
    four.fuse_stage = False
    last, _ = _model("hifigan", cfg, seed=3)
    last.fuse_stage = (16,)
    auto, _ = _model("hifigan", cfg, seed=3)          # the default: per call, by the shape (hifigan._stage_one_launch)
    assert auto.fuse_stage is True
    with torch.no_grad():
        for T, batch in ((1000, 1), (560, 1), (77, 3), (9, 2)):
            x = torch.from_numpy(seeded_mel(T, seed=15, batch=batch)).to(_dev())
            a, b = one(x), four(x)
            assert torch.equal(a, b) and torch.equal(last(x), a) and torch.equal(auto(x), a), (T, batch)
            assert torch.equal(one(x[:1].contiguous())[0], a[0])
        # 560 frames are 67 200 columns at 32 channels: one window per block -> one launch; 1000 frames: two -> pair launches
        auto._fv_batch = 1
        assert auto._flag_tag(560) == "ffss" and auto._flag_tag(1000) == "fffs" and one._flag_tag(1000) == "ffss"
        mel = seeded_mel(123, seed=16)
        assert torch.equal(one.inference(mel), four.inference(mel))
        bias = four.inference(np.zeros_like(mel))
        (e1, r1), (e2, r2) = one.inference_minus(mel, bias), four.inference_minus(mel, bias)
        assert torch.equal(e1, e2) and torch.equal(r1, r2)
    def launches(m, x):
        torch.cuda.synchronize()
        _native.profile_collect(-1)
        _native.profile_enable(True)
        with torch.no_grad():
            m(x)
        torch.cuda.synchronize()
        _native.profile_enable(False)
        return int(_native.profile_collect(-1)["launches"])
    x = torch.from_numpy(seeded_mel(77, seed=15, batch=1)).to(_dev())
    n1, n4, nl = launches(one, x), launches(four, x), launches(last, x)
    assert nl == n4 - 3, (nl, n4)                      # four launches of the last stage -> one
    assert n1 == n4 - 5, (n1, n4)                      # ... and three of the 32-channel stage -> one
    assert not one.check_range() and not four.check_range() and not last.check_range()


@pytest.mark.parametrize("name,path,T", [("hifigan", "conf/hifigan/light.yaml", 1000), ("hifigan", "conf/hifigan/large.yaml", 60),
                                         ("melgan", "conf/melgan/original.yaml", 200), ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 90)],
                         ids=["hifigan_light", "hifigan_large", "melgan", "mb_light"])
def test_lean_upsamplers_give_the_same_bits(name, path, T):
    """The transposed-conv upsamplers of a small batch run on the lean kernel (csrc/convtl_kernels.hpp: 64-column tiles, A
    operands from L2 straight into registers, no weight ring); `fv_tuning_set("convt_lean", 0)` sends them through the ring
    pipeline (convt_kernel / convu_kernel) instead.  Same GEMM, same K order per output, the MRF merge formed the same way
    while the window loads: whole generators give identical bits -- one utterance and a ragged batch."""
    cfg = cases.load_conf(path)
    m, _ = _model(name, cfg, seed=2)
    xs = [torch.from_numpy(seeded_mel(T, seed=21, batch=1)).to(_dev()), torch.from_numpy(seeded_mel(37, seed=22, batch=3)).to(_dev())]
    try:
        with torch.no_grad():
            lean = [m(x).clone() for x in xs]
            _native.tuning_set("convt_lean", 0)
            ring = [m(x).clone() for x in xs]
    finally:
        _native.tuning_set("convt_lean", 50)
    assert all(torch.equal(a, b) for a, b in zip(lean, ring))
    assert not m.check_range()


def test_three_instruction_division_on_the_device():
    """csrc/pair_kernels.hpp div_exact (the MRF mean's xs / 3, hifigan.py:103) against the device's own IEEE division: every
    fp32 value of four binades (and the negatives), d = 3 -- and 2, 5, 7, 12 for the rule's other divisors: zero mismatches."""
    cnt = torch.zeros(1, dtype=torch.int64, device=_dev())
    L = _native.lib()
    for d in (3.0, 2.0, 5.0, 7.0, 12.0):
        for first in (0x3F800000, 0x00C00000, 0x7E000000, 0x42000000):          # 1.0, tiny, huge, 32.0: 2^23 values each
            _native.check(L.fv_div_probe(first, 1 << 23, d, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0


def test_headline_workload_full_tensor_at_T1000():
    """BASELINE config 2 (HiFi-GAN light, 1000 frames -> 240 000 samples) against the validated ATen port on the host,
    EVERY sample (the golden pins 1024 strided samples and the sums: a tile-boundary fault between two strides would have to
    show in the sums); the same mel and weights as the golden, so the strided samples are checked in the same breath."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    m, sd = _model("hifigan", cfg, seed=0)
    mel = seeded_mel(1000, seed=1)
    with torch.no_grad():
        y = m.inference(mel)
    ref = torch_port.inference("hifigan", mel, sd, cfg).numpy()
    assert y.numel() == 240000 and _err(y, ref) <= TOL
    g = np.load(os.path.join(cases.ROOT, "tests", "golden", "full_hifigan_light.npz"))
    assert np.abs(y.cpu().numpy()[g["T1000_idx"]] - g["T1000_samples"]).max() <= TOL


@pytest.mark.parametrize("name,path,golden", [("melgan", "conf/melgan/original.yaml", "full_melgan"),
                                              ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", "full_mb_light"),
                                              ("basis-melgan", "conf/basis-melgan/light.yaml", "full_basis"),
                                              ("hifigan", "conf/hifigan/large.yaml", "full_hifigan_large")],
                         ids=["config1_melgan", "config3_mb_light", "config4_basis", "config5_hifigan_large"])
def test_other_baseline_workloads_full_tensor_at_T1000(name, path, golden):
    """The other BASELINE configs at the benchmark length, one utterance, EVERY sample against the validated ATen port on the
    host (at T = 1000 they were tied to the reference by 1024 strided samples and sums only) -- and the
    golden's strided samples of the reference's own output for the same mel and weights in the same breath."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    mel = seeded_mel(1000, seed=1)
    with torch.no_grad():
        y = m.inference(mel)
    ref = torch_port.inference(name, mel, sd, cfg).numpy()
    g = np.load(os.path.join(cases.ROOT, "tests", "golden", golden + ".npz"))
    assert y.numel() == ref.size == int(g["T1000_n"]) and _err(y, ref) <= TOL
    assert np.abs(y.cpu().numpy().reshape(-1)[g["T1000_idx"]] - g["T1000_samples"]).max() <= TOL


_PORT_CACHE = {}      # (model, conf, what, T) -> the ATen port's output: the two fuse_stage variants share the reference


def _port(name, path, what, T, make):
    key = (name, path, what, T)
    if key not in _PORT_CACHE:
        _PORT_CACHE[key] = make()
    return _PORT_CACHE[key]


def _sweep_lengths(n_random, t_max, batch_for_policy=(1, 3)):
    """Frame counts for the randomized-length sweep: the corners the one-launch stage kernels cut their work at, plus seeded
    random lengths.  16-channel stage: 240 T columns in 516-column final regions (240 T mod 516 == 0 iff T mod 43 == 0; the
    residues next to it are T mod 43 in {28, 15}: 240 T mod 516 = 12 / 504 -- 240 T is a multiple of 12, so +-1 does not
    exist); 32-channel stage: 120 T columns in 324-column windows (T mod 27 == 0; neighbours T mod 27 in {19, 8}); lengths
    below the stage halo (60 columns: T = 1 is 240 samples, one partial window); and both sides of every flip of the
    per-call policy hifigan.stage32_windows_fit in range (256 CUs; batch 1 and 3)."""
    from fastvocoder_amd.generator.hifigan import stage32_windows_fit
    ts = {1, 2, 3, 5, 9, 43, 86, 43 * 6, 43 * 5 + 28, 43 * 9 + 15, 27 * 4, 27 * 11, 27 * 7 + 19, 27 * 15 + 8, 1000, 27 * 2 + 19, 43 + 15}
    for B in batch_for_policy:
        flips = [t for t in range(2, t_max + 1) if stage32_windows_fit(120 * t * B, 256) != stage32_windows_fit(120 * (t - 1) * B, 256)]
        for t in flips[:: max(1, len(flips) // 4)][:4]:
            ts.update((t - 1, t))
    rng = np.random.RandomState(606)
    ts.update(int(v) for v in rng.randint(1, t_max + 1, size=n_random))
    return sorted(t for t in ts if t <= t_max)


@pytest.mark.parametrize("name,path,t_max,n_random,fuse", [
    ("hifigan", "conf/hifigan/light.yaml", 1500, 10, True),
    ("hifigan", "conf/hifigan/light.yaml", 1500, 10, (16, 32)),
    ("hifigan", "conf/hifigan/large.yaml", 360, 12, True),
    ("hifigan", "conf/hifigan/large.yaml", 360, 12, (16, 32)),
    ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 1500, 10, True),
], ids=["hifigan_light", "hifigan_light_one_launch_stages", "hifigan_large", "hifigan_large_one_launch_stages", "mb_light"])
def test_random_lengths_vs_aten_port(name, path, t_max, n_random, fuse):
    """Whole shipped generators at >= 25 lengths in [1, t_max] against the validated ATen port on the host, EVERY sample, 1e-4
    (the one-launch stage kernels and the per-call policy met the oracle at T in {16, 64, 250, 1000} only;
    everything else was self-comparison with the pair launches).  Batch 1 through `inference` at every length; batch 3 (three
    different utterances) through `forward` at six of them.  `fuse` = (16, 32) forces the one-launch kernels at both widths,
    True is the default per-call policy (16 channels always, 32 where the windows fit)."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    m.fuse_stage = fuse
    folded = torch_port.fold_state_dict(sd)
    lengths = _sweep_lengths(n_random, t_max)
    assert len(lengths) >= 25, lengths
    worst = 0.0
    batched = set(lengths[:3] + lengths[len(lengths) // 3::4][:3])
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))          # (the port's small convs do not scale past a few dozen threads)
    with torch.no_grad():
        for T in lengths:
            mel = seeded_mel(T, seed=3000 + T)
            err = _err(m.inference(mel), _port(name, path, "inference", T, lambda: torch_port.inference(name, mel, folded, cfg).numpy()))
            assert err <= TOL, (T, err)
            worst = max(worst, err)
            if T in batched:
                x = seeded_mel(T, seed=5000 + T, batch=3)
                err = _err(m(torch.from_numpy(x)), _port(name, path, "forward", T, lambda: torch_port.forward(name, x, folded, cfg).numpy()))
                assert err <= TOL, (T, "batch 3", err)
                worst = max(worst, err)
    torch.set_num_threads(threads)
    assert not m.check_range()
    print(f"{name} {path} fuse_stage={fuse}: {len(lengths)} lengths, worst {worst:.2e}")


@pytest.mark.parametrize("name,path,t_max,n_random", [
    ("melgan", "conf/melgan/original.yaml", 700, 22),
    ("basis-melgan", "conf/basis-melgan/light.yaml", 1500, 22),
    ("multiband-hifigan", "conf/multiband-hifigan/large.yaml", 300, 22),
], ids=["melgan", "basis_melgan_light", "mb_large"])
def test_random_lengths_vs_aten_port_other_generators(name, path, t_max, n_random):
    """The randomized-length sweep for the generators the round-6 sweep above left out: MelGAN and Basis-MelGAN (every
    ResidualStack is ONE launch whose reflection padding, `modules.py:296-382`, is cut per tile: the short lengths are where a
    tile holds both reflected ends), MB-HiFi-GAN large (256-channel convs conv by conv + PQMF).  >= 25 lengths, every sample
    against the ATen port, 1e-4; batch 1 through `inference`, batch 3 through `forward` at six of them (Basis-MelGAN's
    `forward` returns the bias-removed pair `(est_source, weight)`, `basis_melgan.py:139-162`: both are compared).  The
    shortest length is the reference's own: ReflectionPad1d(3) in front of the first conv needs T >= 4."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    folded = torch_port.fold_state_dict(sd)
    rng = np.random.RandomState(707)
    lo = 4 if "melgan" in name else 1
    ts = {lo, lo + 1, lo + 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, t_max}
    ts.update(int(v) for v in rng.randint(lo, t_max + 1, size=n_random))
    lengths = sorted(t for t in ts if lo <= t <= t_max)
    assert len(lengths) >= 25, lengths
    batched = set(lengths[:3] + lengths[len(lengths) // 3::4][:3])
    worst = 0.0
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    with torch.no_grad():
        for T in lengths:
            mel = seeded_mel(T, seed=7000 + T)
            err = _err(m.inference(mel), torch_port.inference(name, mel, folded, cfg).numpy())
            assert err <= TOL, (T, err)
            worst = max(worst, err)
            if T in batched:
                x = seeded_mel(T, seed=9000 + T, batch=3)
                got, ref = m(torch.from_numpy(x)), torch_port.forward(name, x, folded, cfg)
                pairs = list(zip(got, ref)) if isinstance(ref, tuple) else [(got, ref)]
                assert not isinstance(ref, tuple) or len(got) == len(ref)
                for g, r in pairs:
                    err = _err(g, r.numpy())
                    assert err <= TOL, (T, "batch 3", err)
                    worst = max(worst, err)
    torch.set_num_threads(threads)
    assert not m.check_range()
    print(f"{name} {path}: {len(lengths)} lengths, worst {worst:.2e}")


@pytest.mark.parametrize("name,path,t_max", [
    ("hifigan", "conf/hifigan/light.yaml", 1200), ("melgan", "conf/melgan/original.yaml", 600),
    ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 1200), ("basis-melgan", "conf/basis-melgan/light.yaml", 1200),
], ids=["hifigan_light", "melgan", "mb_light", "basis_light"])
def test_random_lengths_on_the_exact_fp32_kernels(name, path, t_max):
    """The kernels a call is REPEATED on when the range guard fires (`precision = "f32"`: exact-fp32 MFMA everywhere,
    csrc/conv_kernels.hpp / pair_kernels.hpp) over the same kind of length sweep, every sample against the ATen port: the
    fallback a user never sees unless it is needed has to be right at every length too."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    m.precision = "f32"
    folded = torch_port.fold_state_dict(sd)
    rng = np.random.RandomState(909)
    lo = 4 if "melgan" in name else 1
    ts = {lo, lo + 1, 9, 16, 17, 43, 63, 64, 65, 127, 129, 256, 1000 if t_max >= 1000 else t_max, t_max}
    ts.update(int(v) for v in rng.randint(lo, t_max + 1, size=8))
    lengths = sorted(t for t in ts if lo <= t <= t_max)
    worst = 0.0
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    with torch.no_grad():
        for i, T in enumerate(lengths):
            mel = seeded_mel(T, seed=15000 + T)
            err = _err(m.inference(mel), torch_port.inference(name, mel, folded, cfg).numpy())
            assert err <= TOL, (T, err)
            worst = max(worst, err)
            if i % 5 == 0:
                x = seeded_mel(T, seed=16000 + T, batch=2)
                got, ref = m(torch.from_numpy(x)), torch_port.forward(name, x, folded, cfg)
                for g, r in (list(zip(got, ref)) if isinstance(ref, tuple) else [(got, ref)]):
                    err = _err(g, r.numpy())
                    assert err <= TOL, (T, "batch 2", err)
                    worst = max(worst, err)
    torch.set_num_threads(threads)
    print(f"{name} precision=f32: {len(lengths)} lengths, worst {worst:.2e}")


@pytest.mark.parametrize("name,path,B,T", [
    ("hifigan", "conf/hifigan/light.yaml", 12, 500), ("hifigan", "conf/hifigan/large.yaml", 6, 250),
    ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 12, 500), ("basis-melgan", "conf/basis-melgan/light.yaml", 12, 500),
    ("melgan", "conf/melgan/original.yaml", 8, 300),
], ids=["hifigan_light_B12", "hifigan_large_B6", "mb_light_B12", "basis_light_B12", "melgan_B8"])
def test_saturated_kernel_forms_vs_aten_port(name, path, B, T):
    """Batches large enough for the launchers to pick the forms they use at saturation (256- / 128-column pair tiles at 64 / 128
    channels, the ring-free 256-channel conv, the resident-image upsampler, the one-launch 32-channel stage) against the
    ATen port on EVERY sample of every utterance -- until now those forms met the oracle through bit-identity with the
    batch-1 forms only.  (Which form ran is the launchers' decision by items per CU -- convh_launch.hip `wide`, `form`;
    HiFi-GAN light shows it in its launch count: 13 here against 15 at batch 1.)"""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    folded = torch_port.fold_state_dict(sd)
    x = seeded_mel(T, seed=21000 + T, batch=B)

    def kinds(inp):
        torch.cuda.synchronize()
        _native.profile_collect(-1)
        _native.profile_enable(True)
        with torch.no_grad():
            out = m(inp)
        torch.cuda.synchronize()
        _native.profile_enable(False)
        return out, int(_native.profile_collect(-1)["launches"])

    xd = torch.from_numpy(x).to(_dev())
    got, n_launches = kinds(xd)
    assert n_launches > 0
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 64))
    ref = torch_port.forward(name, x, folded, cfg)
    torch.set_num_threads(threads)
    worst = 0.0
    for g, r in (list(zip(got, ref)) if isinstance(ref, tuple) else [(got, ref)]):
        err = _err(g, r.numpy())
        assert err <= TOL, (name, B, T, err)
        worst = max(worst, err)
    assert not m.check_range()
    print(f"{name} {path} B={B} T={T}: worst {worst:.2e}, {n_launches} launches")


def test_batch_rows_are_independent_and_bit_identical():
    """Utterances never mix: row b of a batched forward equals the single-row call
    bit for bit (this is what makes N-GPU sharding exact)."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    m, _ = _model("hifigan", cfg, seed=0)
    x = torch.from_numpy(seeded_mel(200, seed=11, batch=3)).to(_dev())
    with torch.no_grad():
        full = m(x)
        for b in range(3):
            one = m(x[b:b + 1].contiguous())
            assert torch.equal(one[0], full[b])


@pytest.mark.parametrize("name,path,B", [("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 32),
                                         ("basis-melgan", "conf/basis-melgan/light.yaml", 64),
                                         ("hifigan", "conf/hifigan/large.yaml", 64)],
                         ids=["config3_mb_light_B32", "config4_basis_B64", "config5_hifigan_large_B64"])
def test_baseline_batch_sizes_by_size_independent_properties(name, path, B):
    """BASELINE.json configs 3-5 at their full batch and T = 1000 (too large for the oracle):
    (i) sampled rows of the batch equal the single-utterance run bit for bit -- which the
    shipped-config tests tie to the reference within 1e-4 --, (ii) the run is deterministic
    (checksum of checksums of two runs), (iii) output lengths follow the reference's law."""
    cfg = cases.load_conf(path)
    m, _ = _model(name, cfg, seed=0)
    T = 1000
    x = torch.from_numpy(seeded_mel(T, seed=77, batch=B)).to(_dev())
    run = (lambda z: m.synthesize_batch(z)) if name == "multiband-hifigan" else \
          (lambda z: m._samples(z)) if name == "basis-melgan" else (lambda z: m(z))
    with torch.no_grad():
        full = run(x)
        again = run(x)
        assert full.shape == (B, 240 * T + (15 if name == "basis-melgan" else 0))
        assert torch.equal(full, again)
        assert float(full.double().sum(dim=1).abs().sum()) == float(again.double().sum(dim=1).abs().sum())
        for b in (0, B // 2, B - 1):
            one = run(x[b:b + 1].contiguous())
            assert torch.equal(one[0], full[b]), (name, b)
    assert bool(torch.isfinite(full).all())


def test_conv_is_linear_without_activation():
    """conv(a*x1 + x2) == a*conv(x1) + conv(x2) - bias terms (size-independent property)."""
    dev = _dev()
    rng = np.random.RandomState(4)
    w = torch.from_numpy((rng.randn(64, 64, 7) / 21).astype(np.float32)).to(dev)
    packed = _native.pack_conv1d(w)
    x1 = torch.from_numpy(rng.randn(1, 64, 100000).astype(np.float32)).to(dev)
    x2 = torch.from_numpy(rng.randn(1, 64, 100000).astype(np.float32)).to(dev)
    f = lambda t: _native.conv1d_fused(t, packed, None, 64, 7, dil=3, pad=9)  # noqa: E731
    lhs = f(2.0 * x1 + x2)
    rhs = 2.0 * f(x1) + f(x2)
    assert float((lhs - rhs).abs().max()) <= 2e-5 * float(rhs.abs().max())


def test_synthesize_triple_config1(golden_dir, tmp_path):
    """BASELINE config 1 through the drop-in Synthesizer: checkpoint file with
    weight norm attached -> (est, est - bias, bias)."""
    from fastvocoder_amd.bin.synthesize import Synthesizer
    g = np.load(os.path.join(golden_dir, "synthesize_melgan.npz"))
    cfg_path = os.path.join(cases.ROOT, "conf/melgan/original.yaml")
    sd = seeded_state_dict("melgan", cases.load_conf("conf/melgan/original.yaml"), seed=0)
    ck = str(tmp_path / "ckpt.pth.tar")
    torch.save({"model": {k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
    syn = Synthesizer(ck, cfg_path, "melgan")
    mel = np.random.RandomState(0).rand(80, 200)
    est, rem, bias = syn.synthesize(mel.T)
    assert est.shape == (48000,)
    assert _err(est, g["est"]) <= TOL
    assert _err(bias, g["bias"]) <= TOL
    assert _err(rem, g["remove"]) <= TOL


def test_encode_16bits_vs_reference_fixture(golden_dir):
    """fv_encode_16bits against what the REFERENCE's data/audio.py:12-14 produced (tests/golden/audio.npz,
    made by make_golden.py): the int16 samples AND the in-place scaled argument, bit for bit -- including
    the max(0.01, .) floor case, an all-zero waveform and rescale_out = 0.4 (hparams.rescale_out)."""
    from fastvocoder_amd import audio
    g = np.load(os.path.join(golden_dir, "audio.npz"))
    for tag in ("unit", "quiet_floor", "rescale04", "big", "zeros"):
        x = torch.from_numpy(g[f"{tag}_in"].copy()).to(_dev())
        q = audio.encode_16bits(x, float(g[f"{tag}_rescale"]))
        assert np.array_equal(q if isinstance(q, np.ndarray) else q.cpu().numpy(), g[f"{tag}_int16"]), tag
        assert np.array_equal(x.cpu().numpy(), g[f"{tag}_scaled"]), tag


@pytest.mark.parametrize("n,peak,rescale", [(48000, 0.7, 1.0), (240015, 0.93, 0.4), (1001, 0.004, 1.0),
                                            (7, 1.7, 0.4), (1, 0.5, 1.0)])
def test_encode_16bits_on_device_is_bit_exact(n, peak, rescale):
    """fv_encode_16bits vs the reference arithmetic in numpy (data/audio.py:12-14): int16 samples,
    the in-place scaled waveform and the peak, all bit for bit (integer work: no tolerance)."""
    from fastvocoder_amd import audio
    rng = np.random.RandomState(n)
    x = (rng.uniform(-1, 1, size=n) * peak).astype(np.float32)
    host = x.copy()
    want = audio.encode_16bits(host, rescale)                 # numpy route, mutates host
    dev = torch.from_numpy(x.copy()).to(_dev())
    got = audio.encode_16bits(dev, rescale)
    assert got.dtype == torch.int16 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(dev.cpu().numpy(), host)            # same in-place mutation
    # batched rows are normalised independently; unaligned rows (odd n) take the scalar path
    xb = np.stack([x, 0.5 * x, -2.0 * x]).astype(np.float32)
    q, pk = _native.encode_16bits(torch.from_numpy(xb.copy()).to(_dev()), rescale, scale_in_place=False)
    for r in range(3):
        row = xb[r].copy()
        assert np.array_equal(q[r].cpu().numpy(), audio.encode_16bits(row, rescale))
        assert float(pk[r]) == float(np.abs(xb[r]).max())
    with pytest.raises(_native.NativeError):
        audio.encode_16bits(torch.zeros(4))


@pytest.mark.parametrize("tag,name,cfg", cases.SMALL, ids=[c[0] for c in cases.SMALL])
def test_time_chunked_run_equals_whole_run(tag, name, cfg):
    """inputs longer than ``max_frames_per_run`` are cut into chunks with
    receptive-field halos; the stitched result must equal the whole-utterance run (same
    arithmetic per sample; only the tile shapes, hence fp32 summation order, may differ)."""
    m, _ = _model(name, cfg, seed=7)
    T = 61
    mel = seeded_mel(T, seed=9)
    whole = m.inference(mel).cpu().numpy()
    melb = torch.from_numpy(seeded_mel(T, seed=10, batch=2)).to(_dev())
    whole_f = m(melb)
    whole_f = whole_f[0] if isinstance(whole_f, tuple) else whole_f
    plan_halo = next(iter(m._fv_plans.values()))[1].halo_frames
    assert 0 < plan_halo < 64
    for chunk in (7, 16, 60):
        m.max_frames_per_run = chunk
        got = m.inference(mel).cpu().numpy()
        assert got.shape == whole.shape
        assert np.abs(got - whole).max() <= 1e-5, (tag, chunk)
        f = m(melb)
        f = f[0] if isinstance(f, tuple) else f
        assert _err(f, whole_f.cpu().numpy()) <= 1e-5, (tag, chunk)


def test_time_chunked_shipped_hifigan_light():
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    m, _ = _model("hifigan", cfg, seed=0)
    mel = seeded_mel(700, seed=4)
    whole = m.inference(mel)
    m.max_frames_per_run = 256
    got = m.inference(mel)
    assert got.shape == whole.shape == (700 * 240,)
    assert _err(got, whole.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("name,path,t_hi", [
    ("hifigan", "conf/hifigan/light.yaml", 900), ("melgan", "conf/melgan/original.yaml", 500),
    ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 900), ("basis-melgan", "conf/basis-melgan/light.yaml", 900),
], ids=["hifigan_light", "melgan", "mb_light", "basis_light"])
def test_random_chunking_of_shipped_generators(name, path, t_hi):
    """SURVEY 8 f-4 on the shipped generators at seeded random (length, chunk) pairs: the stitched chunks equal the
    whole-utterance run (1e-5: the same arithmetic per sample in other tile shapes) AND the ATen port of the reference on the
    whole utterance (1e-4) -- chunks shorter than the receptive-field halo, a last chunk of one frame, MelGAN's reflection
    padding (`modules.py:355-356`) applying at the utterance's ends only, never at a chunk boundary."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    folded = torch_port.fold_state_dict(sd)
    rng = np.random.RandomState(808)
    pairs = [(int(rng.randint(40, t_hi + 1)), None) for _ in range(5)]
    pairs = [(T, int(rng.randint(3, T))) for T, _ in pairs] + [(129, 128), (130, 5), (257, 64)]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    worst = 0.0
    with torch.no_grad():
        for T, chunk in pairs:
            mel = seeded_mel(T, seed=12000 + T)
            m.max_frames_per_run = 16384
            whole = m.inference(mel).cpu().numpy()
            m.max_frames_per_run = chunk
            got = m.inference(mel).cpu().numpy()
            assert got.shape == whole.shape, (T, chunk)
            assert np.abs(got - whole).max() <= 1e-5, (T, chunk, float(np.abs(got - whole).max()))
            err = _err(torch.from_numpy(got), torch_port.inference(name, mel, folded, cfg).numpy())
            assert err <= TOL, (T, chunk, err)
            worst = max(worst, err)
    torch.set_num_threads(threads)
    m.max_frames_per_run = 16384
    assert not m.check_range()
    print(f"{name}: {len(pairs)} (length, chunk) pairs, worst vs the port {worst:.2e}")


def test_errors_are_loud():
    with pytest.raises(_native.NativeError):
        fa.HiFiGANGenerator()(torch.zeros(1, 80, 8))      # CPU module: no fallback
    with pytest.raises(Exception, match="no model find"):
        build_generator("wavenet", {})
    dev = _dev()
    x = torch.zeros(1, 8, 4, device=dev)
    packed = _native.pack_conv1d(torch.zeros(8, 8, 3, device=dev))
    with pytest.raises(_native.NativeError):               # reflection pad longer than the input
        _native.conv1d_fused(x, packed, None, 8, 3, dil=9, pad=9, pad_mode=_native.PAD_REFLECT)


def test_edge_inputs_match_the_reference_behaviour():
    """Shortest and odd inputs: a single mel frame, lengths that are not multiples of the DMA
    vector width or of any tile, the empty utterance (the reference's conv raises there too),
    wrong channel count, and a ragged set of utterances run one by one (the reference has no
    padding/masking: ragged batches are separate calls)."""
    cfg = dict(cases.SMALL[0][2])
    m, sd = _model("hifigan", cfg, seed=7)
    for T in (1, 2, 3, 5, 17, 33):
        mel = seeded_mel(T, seed=T)
        want = og.hifigan_inference(mel, sd, cfg) if hasattr(og, "hifigan_inference") else None
        got = m.inference(mel)
        assert got.numel() == T * int(np.prod(cfg["upsample_rates"]))
        if want is not None:
            assert _err(got.reshape(-1), np.asarray(want).reshape(-1)) <= TOL
    with pytest.raises(_native.NativeError):
        m(torch.zeros(1, 80, 0, device=_dev()))               # empty utterance
    with pytest.raises(_native.NativeError):
        m(torch.zeros(1, 79, 8, device=_dev()))               # wrong number of mel channels
    # ragged utterances, one call each, equal the same utterances inside an equal-length batch slice
    mels = [seeded_mel(T, seed=40 + T) for T in (7, 19, 12)]
    outs = [m.inference(x) for x in mels]
    for x, y in zip(mels, outs):
        again = m(torch.from_numpy(x.T[None].copy()).to(_dev()))[0]
        assert torch.equal(again, y)


@pytest.mark.parametrize("tag,name,path", cases.SHIPPED, ids=[c[0] for c in cases.SHIPPED])
def test_inference_minus_is_one_pass_of_the_two_pass_flow(tag, name, path):
    """Bias removal in the epilogue: inference_minus(mel, bias) returns the plain inference
    output and output - bias, both bit-identical to the two-pass form (inference, then a subtraction) -- for
    every generator: conv_post / LastLayer epilogue, PQMF synthesis, Basis overlap-add."""
    cfg = cases.load_conf(path)
    m, _ = _model(name, cfg, seed=0)
    mel = seeded_mel(48, seed=5)
    with torch.no_grad():
        bias = m.inference(np.zeros_like(mel)).clone()
        est = m.inference(mel).clone()
        e2, rem = m.inference_minus(mel, bias)
    assert torch.equal(e2, est)
    assert torch.equal(rem, est - bias)
    assert float((est - bias).abs().max()) > 1e-3          # not vacuous


def test_basis_forward_caches_the_zero_pass():
    cfg = cases.load_conf("conf/basis-melgan/light.yaml")
    m, _ = _model("basis-melgan", cfg, seed=0)
    x = torch.from_numpy(seeded_mel(16, seed=3, batch=2)).to(_dev())
    with torch.no_grad():
        a = m(x)
        zw = m._zero_response(16)[0]
        b = m(x)
        assert m._zero_response(16)[0].data_ptr() == zw.data_ptr()      # served from the cache
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # the two-pass form of the reference (basis_melgan.py:147-159), for the same weights
        w = m._weights(x)
        z = m._weights(torch.zeros_like(x))
        assert torch.equal(a[1], (w - z).transpose(1, 2))
        m.melgan[1].weight.data.mul_(1.0)                                # same values, but "modified": rebuilds
        m.invalidate_plans()
        c = m(x)
        assert torch.equal(a[0], c[0])


# ---------------------------------------------------------------------------
# outside the split-f16 domain: the reference is defined for any finite fp32 -- so is the engine
# ---------------------------------------------------------------------------
def _scaled_hifigan(gain, seed=0):
    """HiFi-GAN light whose conv_pre (an fp32 kernel: no weight limit) is `gain` times too loud and whose conv_post
    undoes it: every split-f16 layer then works at `gain` times its usual scale (beyond the f16 range for gain = 1e6)
    with its weights untouched, while the waveform stays comparable to the unscaled model's (leaky ReLU is homogeneous;
    only the inner biases do not scale)."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    m, _ = _model("hifigan", cfg, seed=seed)
    m.remove_weight_norm()
    with torch.no_grad():
        m.conv_pre.weight.mul_(gain)
        m.conv_pre.bias.mul_(gain)
        m.conv_post.weight.mul_(1.0 / gain)
    return m


def test_generator_beyond_the_f16_range_repeats_on_fp32():
    """Activations beyond 65504 inside the generator: every entry under the default policy (range_guard "auto" =
    checked before the call returns) -- `inference` AND `forward` -- notices, repeats the call on the exact-fp32 kernels
    and returns what a precision = "f32" model returns, bit for bit; the module then stays on the fp32 kernels.  The
    explicit opt-in "lazy" defers the check to the next call / check_range()."""
    mel = seeded_mel(64, seed=31)
    exact = _scaled_hifigan(1e6)
    exact.precision = "f32"
    with torch.no_grad():
        want = exact.inference(mel)
    assert bool(torch.isfinite(want).all()) and float(want.abs().max()) > 1e-3
    m = _scaled_hifigan(1e6)
    with pytest.warns(RuntimeWarning, match="split-f16 range"), torch.no_grad():
        got = m.inference(mel)
    assert torch.equal(got, want)
    assert m._fv_policy()[0] == "f32"
    with torch.no_grad():
        again = m.inference(mel)                    # no second warning, no second split-f16 attempt
    assert torch.equal(again, want)
    # the oracle agrees with the fp32 path on this model (the reference's arithmetic has no such limit)
    sd = {k: v.detach().cpu().numpy() for k, v in exact.state_dict().items()}
    ref = torch_port.inference("hifigan", mel, sd, cases.load_conf("conf/hifigan/light.yaml")).numpy()
    assert _err(want, ref) <= TOL
    # forward under the DEFAULT policy: checked before it returns, with or without torch.no_grad()
    x = torch.from_numpy(np.ascontiguousarray(mel.T[None])).to(_dev())      # the same mel, forward layout [1, 80, T]
    dflt = _scaled_hifigan(1e6)
    assert dflt.range_guard == "auto"
    with pytest.warns(RuntimeWarning, match="split-f16 range"), torch.no_grad():
        assert torch.equal(dflt(x)[0], want)
    assert dflt._fv_policy()[0] == "f32"
    dflt2 = _scaled_hifigan(1e6)
    with pytest.warns(RuntimeWarning, match="split-f16 range"):
        assert torch.equal(dflt2(x)[0], want)
    # forward under the explicit opt-in "lazy": stream-ordered, the check is deferred
    lazy = _scaled_hifigan(1e6)
    lazy.range_guard = "lazy"
    with torch.no_grad():
        first = lazy(x)
        with pytest.warns(RuntimeWarning, match="split-f16 range"):
            assert lazy.check_range()
        second = lazy(x)
    assert not bool(torch.isfinite(first).all())
    assert torch.equal(second[0], want)
    # a model that stays inside the range never leaves the split-f16 kernels
    ok = _scaled_hifigan(100.0)
    with torch.no_grad():
        y = ok.inference(mel)
    assert ok._fv_policy()[0] == "split" and bool(torch.isfinite(y).all()) and not ok.check_range()
    # "sync" on forward: checked before the call returns
    strict = _scaled_hifigan(1e6)
    strict.range_guard = "sync"
    with pytest.warns(RuntimeWarning, match="split-f16 range"), torch.no_grad():
        assert torch.equal(strict(x)[0], want)


def _scaled_melgan(gain, seed=0):
    """MelGAN original whose first conv (an fp32 kernel) is `gain` times too loud and whose last layer undoes it: every
    upsampler and ResidualStack (one-launch split-f16 kernels, csrc/convk_kernels.hpp) then works at `gain` times its usual
    scale with its weights untouched."""
    cfg = cases.load_conf("conf/melgan/original.yaml")
    m, _ = _model("melgan", cfg, seed=seed)
    m.remove_weight_norm()
    from fastvocoder_amd.generator.modules import LastLayer
    first = next(mod for mod in m.melgan if isinstance(mod, torch.nn.Conv1d))
    last = next(mod for mod in m.melgan if isinstance(mod, LastLayer)).conv
    with torch.no_grad():
        first.weight.mul_(gain)
        first.bias.mul_(gain)
        last.weight.mul_(1.0 / gain)
    return m


@pytest.mark.parametrize("gain", [1e7, 1e-7])
def test_melgan_outside_the_split_domain_repeats_on_fp32(gain):
    """Both sides of the split-f16 domain through the one-launch ResidualStack kernels: activations beyond the f16 range (the
    kernels' range guard, value 1) and tensors that are tiny as a whole (the low-side guard, value 4) send `inference` to
    the exact-fp32 kernels -- the two-launch fp32 form of every stack -- and the result is the fp32 model's, bit for bit."""
    mel = seeded_mel(40, seed=5)
    exact = _scaled_melgan(gain)
    exact.precision = "f32"
    with torch.no_grad():
        want = exact.inference(mel)
    assert bool(torch.isfinite(want).all()) and float(want.abs().max()) > 1e-4
    m = _scaled_melgan(gain)
    if gain < 1:
        # the low side alone is first taken for a property of the input: the call is repeated on fp32, the module stays
        with torch.no_grad():
            for _ in range(m.low_range_patience):
                assert torch.equal(m.inference(mel), want) and m._fv_policy()[0] == "split"
    with pytest.warns(RuntimeWarning, match="split-f16 range"), torch.no_grad():
        got = m.inference(mel)
    assert torch.equal(got, want) and m._fv_policy()[0] == "f32"
    ok = _scaled_melgan(30.0 if gain > 1 else 1.0 / 30.0)          # inside the domain: stays on the split kernels
    with torch.no_grad():
        y = ok.inference(mel)
    assert ok._fv_policy()[0] == "split" and not ok.check_range() and bool(torch.isfinite(y).all())


def test_weights_of_any_magnitude_stay_on_the_split_kernels():
    """Weights far outside the f16 range inside a generator (round 3: a weight beyond 65504 meant an fp32 plan, one below
    6e-5 lost bits silently).  Every ResBlock's first convs 2^10 times too large and its second convs 2^-10 times too small
    -- weights of 2e-5, below the smallest normal f16, behind intermediates in the thousands: the same function in exact
    arithmetic (leaky ReLU is homogeneous), and the same output within the fp32 noise on the split-f16 kernels -- the
    pack functions rescale every row by a power of two -- with no guard raised."""
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    mel = seeded_mel(48, seed=33)
    plain, _ = _model("hifigan", cfg, seed=0)
    m, _ = _model("hifigan", cfg, seed=0)
    m.remove_weight_norm()
    with torch.no_grad():
        for rb in m.resblocks:
            for c1, c2 in zip(rb.convs1, rb.convs2):
                c1.weight.mul_(2.0 ** 10)
                c1.bias.mul_(2.0 ** 10)
                c2.weight.mul_(2.0 ** -10)
        want = plain.inference(mel)
        got = m.inference(mel)
    assert m._fv_policy()[0] == "split" and not m.check_range()
    assert float(want.abs().max()) > 0.05 and _err(got, want.cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("path,T,spread", [("conf/hifigan/light.yaml", 200, 8), ("conf/hifigan/light.yaml", 77, 11),
                                           ("conf/hifigan/large.yaml", 40, 8)],
                         ids=["light_2^16", "light_2^22", "large_2^16"])
def test_channel_scales_of_a_trained_model_stay_inside_the_tolerance(path, T, spread):
    """No trained checkpoint exists here (the reference names a URL only), and seeded weights give every channel of a layer
    the same scale -- a trained vocoder does not: a few loud channels beside nearly dead ones.  So: every intermediate
    channel of every ResBlock pair gets its own power-of-two gain, log-uniform over 2^(2 spread) (conv1's row and bias times
    g, conv2's column times 1 / g: the same function in exact arithmetic, leaky ReLU being homogeneous), i.e. inside ONE
    split-f16 operand tile magnitudes differ by up to 2^16 (2^22), rows of conv2 mix weights of that spread, and the row
    prescale sees the largest only.  The split kernels must still be within 1e-4 of the ATen port ON THESE WEIGHTS, agree
    with the unscaled model, and raise no guard."""
    cfg = cases.load_conf(path)
    mel = seeded_mel(T, seed=77)
    plain, _ = _model("hifigan", cfg, seed=0)
    m, _ = _model("hifigan", cfg, seed=0)
    m.remove_weight_norm()
    rng = np.random.RandomState(1234)
    with torch.no_grad():
        for rb in m.resblocks:
            for c1, c2 in zip(rb.convs1, rb.convs2):
                g = torch.from_numpy((2.0 ** rng.randint(-spread, spread + 1, size=c1.weight.shape[0])).astype(np.float32)).to(c1.weight.device)
                c1.weight.mul_(g[:, None, None])
                c1.bias.mul_(g)
                c2.weight.mul_(1.0 / g[None, :, None])
        want = plain.inference(mel)
        got = m.inference(mel)
    assert m._fv_policy()[0] == "split" and not m.check_range()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = torch_port.inference("hifigan", mel, sd, cfg).numpy()
    assert float(np.abs(ref).max()) > 0.05
    assert _err(got, ref) <= TOL and _err(got, want.cpu().numpy()) <= TOL
    print(f"{path} spread 2^{2 * spread}: {_err(got, ref):.2e} from the port, {_err(got, want.cpu().numpy()):.2e} from the unscaled model")


@pytest.mark.parametrize("name,path", [("hifigan", "conf/hifigan/light.yaml"), ("melgan", "conf/melgan/original.yaml"),
                                       ("multiband-hifigan", "conf/multiband-hifigan/light.yaml")])
def test_silence_and_transients_in_the_mel(name, path):
    """What a real utterance looks like and the seeded mels do not: the reference normalises its mels into [0, 1] WITH clipping
    (/root/reference/data/audio.py:159-160), so silence is a stretch of exact zeros and loud frames sit at exactly 1; and a
    caller with un-normalised log-mels feeds a floor of log(1e-5) with isolated peaks.  Blocks of the split kernels then see
    exact zeros, constant columns and transients.  Every sample against the ATen port; whatever the guard decides, the
    call returns the reference's values."""
    cfg = cases.load_conf(path)
    m, sd = _model(name, cfg, seed=0)
    folded = torch_port.fold_state_dict(sd)
    T = 360
    mel = np.array(seeded_mel(T, seed=91), copy=True)
    mel[40:150] = 0.0                                   # silence after the clip
    mel[150:156] = 1.0                                  # an onset at the ceiling
    mel[200:260] = np.float32(np.log(1e-5))             # an un-normalised floor
    mel[230] += 20.0                                    # ... with one frame far above it
    mel[300:, 40:] = 0.0                                # band-limited tail
    with torch.no_grad():
        got = m.inference(mel)
    ref = torch_port.inference(name, mel, folded, cfg).numpy()
    assert bool(torch.isfinite(got).all()) and _err(got, ref) <= TOL
    print(f"{name}: {_err(got, ref):.2e}, policy after the call: {m._fv_policy()[0]}")


def test_generator_below_the_low_side_repeats_on_fp32():
    """The low side of the domain end to end: conv_pre 2^-20 times too quiet (conv_post undoes it) -- the first split-f16
    layer sees a tensor that is small as a whole, the guard fires (4), `inference` repeats the call on the exact-fp32
    kernels and returns the fp32 model's output bit for bit -- call by call at first (a quiet input is not the model's
    fault), for good after `low_range_patience` such calls in a row; at 2^-6 the model stays on the split kernels."""
    mel = seeded_mel(64, seed=31)
    exact = _scaled_hifigan(2.0 ** -20)
    exact.precision = "f32"
    with torch.no_grad():
        want = exact.inference(mel)
    m = _scaled_hifigan(2.0 ** -20)
    with torch.no_grad():
        for _ in range(m.low_range_patience):
            assert torch.equal(m.inference(mel), want) and m._fv_policy()[0] == "split" and not m._fv_overflow
    with pytest.warns(RuntimeWarning, match="split-f16 range"), torch.no_grad():
        got = m.inference(mel)
    assert torch.equal(got, want) and m._fv_policy()[0] == "f32"
    ok = _scaled_hifigan(2.0 ** -6)
    with torch.no_grad():
        y = ok.inference(mel)
    assert ok._fv_policy()[0] == "split" and bool(torch.isfinite(y).all()) and not ok.check_range()


@pytest.mark.parametrize("path", ["conf/multiband-hifigan/light.yaml", "conf/multiband-hifigan/large.yaml"])
def test_conv_post_and_pqmf_in_one_launch_give_the_same_bits(path):
    """Multiband-HiFi-GAN's inference tail (multiband_hifigan.py:113-115,136): conv_post + tanh + PQMF synthesis as ONE
    launch (fv_conv_post_pqmf: the sub-bands stay in LDS) against the two launches it replaces -- same FMA order, so
    identical bits: single utterance, a ragged batch, and the bias-removal plan (output offset in the last epilogue)."""
    cfg = cases.load_conf(path)
    fused, _ = _model("multiband-hifigan", cfg, seed=0)
    plain, _ = _model("multiband-hifigan", cfg, seed=0)
    plain.fuse_pqmf = False
    mel = seeded_mel(61, seed=41)
    with torch.no_grad():
        a, b = fused.inference(mel), plain.inference(mel)
        assert torch.equal(a, b) and a.numel() == plain.inference(mel).numel()
        x = torch.from_numpy(seeded_mel(333, seed=42, batch=3)).to(_dev())
        assert torch.equal(fused.synthesize_batch(x), plain.synthesize_batch(x))
        bias = plain.inference(np.zeros_like(mel))
        (e1, r1), (e2, r2) = fused.inference_minus(mel, bias), plain.inference_minus(mel, bias)
        assert torch.equal(e1, e2) and torch.equal(r1, r2) and torch.equal(e1, a)
    n_f = _plans(fused)
    n_p = _plans(plain)
    assert all(n_f[k].num_ops() == n_p[k].num_ops() - 1 for k in n_f)


@pytest.mark.parametrize("name,path", [("basis-melgan", "conf/basis-melgan/light.yaml"), ("melgan", "conf/melgan/original.yaml"),
                                       ("hifigan", "conf/hifigan/large.yaml")], ids=["basis", "melgan", "hifigan-large"])
def test_128_row_tiles_give_the_same_waveform_bits(name, path):
    """The launchers pick 128-row tiles (csrc/convr_kernels.hpp: convr_kernel for ResidualStack's 1x1 + skip GEMM,
    convs_kernel for the convs at 128+ channels) for launches that fill the chip and 64-row tiles otherwise; both give
    identical bits, so whole generators do too whichever is forced: forward, the bias-removal flow (the output offset of
    the GEMM's epilogue), and a batch."""
    cfg = cases.load_conf(path)
    m, _ = _model(name, cfg, seed=0)
    mel = seeded_mel(40, seed=9)
    x = torch.from_numpy(seeded_mel(33, seed=8, batch=2)).to(_dev())
    outs = []
    try:
        for rows64 in (1, 0):
            _native.tuning_set("convg_rows64", rows64)
            _native.tuning_set("convh_rows64", rows64)
            m.invalidate_plans()
            with torch.no_grad():
                bias = m.inference(np.zeros_like(mel)).clone()
                est, rem = m.inference_minus(mel, bias)
                y = m(x)
                ys = (y,) if isinstance(y, torch.Tensor) else tuple(y)
                outs.append((bias, est.clone(), rem.clone()) + tuple(t.clone() for t in ys))
    finally:
        _native.tuning_set("convg_rows64", -1)
        _native.tuning_set("convh_rows64", -1)
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert not m.check_range()
