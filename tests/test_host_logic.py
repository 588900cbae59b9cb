"""CPU-only tests of the host side: state_dict wire format vs the reference's key
tables, config loading, seeded checkpoints, weight-norm lifecycle bookkeeping,
error behaviour without a GPU, the polyphase ConvTranspose1d packing maths, and
that the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import fastvocoder_amd as fa
from fastvocoder_amd import _native
from fastvocoder_amd.bin.synthesize import build_generator
from fastvocoder_amd.synthetic import seeded_state_dict, state_dict_spec
from oracle import ops as oo
from tests import cases


def test_state_dict_keys_match_reference_tables(golden_dir):
    """keys.json was dumped from the imported reference for the six shipped yamls."""
    with open(os.path.join(golden_dir, "keys.json")) as f:
        ref = json.load(f)
    for tag, name, path in cases.SHIPPED:
        cfg = cases.load_conf(path)
        model = build_generator(name, cfg)
        sd = model.state_dict()
        assert list(sd.keys()) == list(k for k, _, _ in state_dict_spec(name, cfg)), tag
        assert {k: list(v.shape) for k, v in sd.items()} == ref[path], tag


def test_checkpoint_roundtrip_and_weight_norm_bookkeeping():
    name, cfg = "hifigan", cases.SMALL[0][2]
    sd = seeded_state_dict(name, cfg, seed=3)
    m = build_generator(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), sd[k]), k
    # remove_weight_norm folds g*v/||v|| into .weight exactly like torch does
    folded = oo.weight_norm_fold(sd["conv_pre.weight_v"], sd["conv_pre.weight_g"])
    m.remove_weight_norm()
    assert "conv_pre.weight" in m.state_dict() and "conv_pre.weight_g" not in m.state_dict()
    assert np.abs(m.state_dict()["conv_pre.weight"].numpy() - folded).max() <= 1e-6
    m.remove_weight_norm()          # idempotent: modules without weight norm are skipped
    m.apply_weight_norm()
    assert set(m.state_dict().keys()) == set(sd.keys())
    # ConvTranspose1d: g is per INPUT channel (dim 0 of [Cin, Cout, k])
    assert m.state_dict()["ups.0.weight_g"].shape[0] == cfg["upsample_initial_channel"]


def test_training_only_yaml_keys_are_ignored_and_missing_keys_raise():
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    assert "lamda_stft" in cfg and "use_feature_map_loss" in cfg
    build_generator("hifigan", cfg)
    bad = dict(cfg)
    del bad["upsample_rates"]
    with pytest.raises(KeyError):
        build_generator("hifigan", bad)
    with pytest.raises(Exception, match="no model find"):
        build_generator("nhv", cfg)


def test_no_cpu_fallback():
    m = build_generator("melgan", cases.SMALL[4][2])
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        m(torch.zeros(1, 80, 16))
    with pytest.raises(_native.NativeError):
        fa.PQMF().synthesis(torch.zeros(1, 4, 8))


def test_optional_variants_keep_the_reference_checkpoint_layout():
    """transposedconv: False, use_causal_conv and lastlinear are constructor options no shipped
    yaml sets; the containers still follow the reference's key layout (pinned by
    tests/golden/make_golden.py, which asserts state_dict_spec == the reference's keys)."""
    from fastvocoder_amd.synthetic import state_dict_spec
    for tag, name, cfg in cases.SMALL:
        if tag not in ("hifigan_up", "basis_up", "melgan_causal", "basis_causal_ll"):
            continue
        m = build_generator(name, cfg)
        spec = state_dict_spec(name, cfg)
        assert [k for k, _, _ in spec] == list(m.state_dict().keys()), tag
        for k, shp, _ in spec:
            assert tuple(m.state_dict()[k].shape) == tuple(shp), (tag, k)
    m = fa.HiFiGANGenerator(transposedconv=False, upsample_initial_channel=32)
    assert any(k.startswith("ups.0.conv.weight_v") for k in m.state_dict())
    with pytest.raises(AssertionError):          # even kernels are only legal with causal convs
        fa.MelGANGenerator(kernel_size=6)


def test_seeded_checkpoint_is_reproducible_and_gain_calibrated():
    cfg = cases.load_conf("conf/melgan/original.yaml")
    a = seeded_state_dict("melgan", cfg, seed=0)
    b = seeded_state_dict("melgan", cfg, seed=0)
    c = seeded_state_dict("melgan", cfg, seed=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert any(not np.array_equal(a[k], c[k]) for k in a)
    v, g = a["melgan.1.weight_v"], a["melgan.1.weight_g"]
    nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1))
    ratio = g.reshape(-1) / nrm
    assert ratio.min() > 0.8 and ratio.max() < 1.2 and ratio.std() > 0.01   # g != ||v||: fold matters


def _polyphase(k, s, p):
    dmin, dmax = 1 << 30, -(1 << 30)
    for r in range(s):
        a, c = (r + p) % s, (r + p) // s
        m, j = 0, a
        while j < k:
            dmin, dmax = min(dmin, c - m), max(dmax, c - m)
            j += s
            m += 1
    return dmin, dmax


@pytest.mark.parametrize("k,s", [(16, 8), (10, 5), (6, 3), (4, 2), (20, 10), (12, 6), (16, 10), (16, 6),
                                 (8, 4), (30, 15), (7, 3)])
def test_polyphase_form_of_conv_transpose(k, s):
    """The identity the HIP path relies on (csrc/fv_internal.h polyphase(), api.hip
    pack_convT_kernel): ConvTranspose1d == a (dmax-dmin+1)-tap dense conv over
    Cout*s phase rows followed by interleaving the phases in time."""
    p = (s // 2 + s % 2) if k != 30 else 0
    op = s % 2 if k != 30 else 0
    rng = np.random.RandomState(k * 100 + s)
    cin, cout, T = 3, 2, 9
    x = rng.randn(1, cin, T).astype(np.float32)
    w = rng.randn(cin, cout, k).astype(np.float32)
    ref = oo.conv_transpose1d(x, w, None, s, p, op)
    dmin, dmax = _polyphase(k, s, p)
    taps = dmax - dmin + 1
    wp = np.zeros((cout * s, cin, taps), np.float32)          # dense conv weight [M, Cin, taps]
    for co in range(cout):
        for r in range(s):
            a, c = (r + p) % s, (r + p) // s
            for jj in range(taps):
                mi = c - dmin - jj
                j = a + mi * s
                if mi >= 0 and j < k:
                    wp[co * s + r, :, jj] = w[:, co, j]
    Tout = ref.shape[2]
    Tq = (Tout + s - 1) // s
    # conv: y[m, q] = sum wp[m, ci, jj] * x[ci, q + jj + dmin]  == conv1d with pad = -dmin (+ right pad)
    xp = np.zeros((1, cin, Tq + taps + 8 + max(0, -dmin) * 2), np.float32)
    off = max(0, -dmin) + 2
    xp[:, :, off:off + T] = x
    y = oo.conv1d(xp, wp, None)          # valid conv over the padded signal
    out = np.zeros_like(ref)
    for m in range(cout * s):
        co, r = divmod(m, s)
        for q in range(Tq):
            t = q * s + r
            if t < Tout:
                out[0, co, t] = y[0, m, q + off + dmin]
    assert np.abs(out - ref).max() <= 1e-5


def test_c_abi_library_exports_every_declared_symbol():
    """No compute calls (no GPU here): the .so loads and every function declared in
    include/fastvocoder_hip.h resolves."""
    header = open(os.path.join(cases.ROOT, "include", "fastvocoder_hip.h")).read()
    names = set(re.findall(r"\b(fv_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 18, names
    lib = ctypes.CDLL(_native.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert _native.lib().fv_version() == _native.ABI_VERSION
    # build provenance: the binary carries the hash of the sources it was built from
    assert _native.lib().fv_build_id().decode() == _native.source_hash() == _native.built_id()
    assert _native.lib().fv_packed_pair_floats(32, 11) == 32 * 32 * 11
    # split-f16 images: C = 16: (k+1)/2 K steps x 2 KB; C = 32: k steps x 2 row halves; C = 64 / 128: row tiles x k C/32 steps x 8 KB
    L = _native.lib()
    assert L.fv_packed_pair_floats_ex(16, 11, _native.PAIR_SPLIT_F16) == 6 * 512 + 16      # (+ C: the rows' inverse prescales)
    assert L.fv_packed_pair_floats_ex(32, 7, _native.PAIR_SPLIT_F16) == 7 * 2 * 512 + 32
    assert L.fv_packed_pair_floats_ex(64, 11, _native.PAIR_SPLIT_F16) == 22 * 2048 + 64
    assert L.fv_packed_pair_floats_ex(128, 3, _native.PAIR_SPLIT_F16) == 2 * 12 * 2048 + 128
    assert L.fv_packed_pair_floats_ex(512, 7, _native.PAIR_SPLIT_F16) == 8 * 4 * 28 * 2048 + 512      # row tiles x chunks x steps
    assert L.fv_packed_pair_floats_ex(48, 3, _native.PAIR_SPLIT_F16) == 0
    assert L.fv_packed_pair_floats_ex(32, 11, _native.PAIR_F32) == 32 * 32 * 11
    # host-only entry points that need no device
    assert _native.lib().fv_packed_conv1d_floats(128, 128, 11) == 128 * 11 * 128
    # k = 2*stride, Cout % 32 == 0: phase-major form, 2 taps per phase (3 in the co-major form)
    assert _native.lib().fv_packed_conv_transpose1d_floats(256, 128, 16, 8, 4) == 256 * 2 * 1024
    assert _native.lib().fv_packed_conv_transpose1d_floats(32, 16, 4, 2, 1) == 32 * 3 * 32
    assert _native.lib().fv_packed_conv1d_floats(1, 16, 7) == 16 * 7 * 16      # rows padded to 16
    assert _native.lib().fv_last_error() is not None


@pytest.mark.parametrize("k,u,p", [(16, 8, 8), (10, 5, 5), (7, 3, 3), (4, 2, 2), (9, 4, 4), (21, 10, 10),
                                   (5, 3, 0), (3, 4, 1)])
def test_summed_phase_form_of_upsample_layer(k, u, p):
    """The identity behind fv_pack_upsample_conv1d_weight (api.hip pack_upconv_kernel,
    fv_internal.h upsample_phases): nearest-repeat x u then Conv1d(k, padding p) == a
    (dmax-dmin+1)-tap dense conv over Cout*u phase rows whose weights are the sums of the
    taps that read the same input sample, then interleaving the phases in time."""
    rng = np.random.RandomState(k * 100 + u)
    cin, cout, T = 3, 2, 9
    x = rng.randn(1, cin, T).astype(np.float32)
    w = rng.randn(cout, cin, k).astype(np.float32)
    ref = oo.conv1d(np.repeat(x, u, axis=2), w, None, dil=1, pad=p)
    dmin, dmax = (-p) // u, (u - 1 + k - 1 - p) // u          # python // floors, like floor_div()
    taps = dmax - dmin + 1
    wp = np.zeros((cout * u, cin, taps), np.float32)
    for co in range(cout):
        for r in range(u):
            for j in range(k):
                wp[co * u + r, :, (r + j - p) // u - dmin] += w[co, :, j]
    Tout = ref.shape[2]
    assert Tout == u * T + 2 * p - (k - 1)
    Tq = (Tout + u - 1) // u
    out = np.zeros((1, cout, Tq * u), np.float64)
    for q in range(Tq):
        for jj in range(taps):
            i = q + dmin + jj
            if 0 <= i < T:
                out[0, :, q * u:(q + 1) * u] += (wp[:, :, jj].astype(np.float64) @ x[0, :, i]).reshape(cout, u)
    assert np.abs(out[:, :, :Tout] - ref).max() <= 1e-5
    mpad = 16 if cout * u <= 16 else -(-cout * u // 32) * 32
    assert _native.lib().fv_packed_upsample_conv1d_floats(cout, cin, k, u, p) == cin * taps * mpad


def test_plan_shape_inference_without_gpu():
    """fv_plan_* shape logic is host code: HiFi-GAN output length = prod(rates)*T,
    MB-large = 60T-20 sub-band samples (x4 after PQMF), Basis = 240T+15."""
    L = _native.lib()
    h = L.fv_plan_create(80)
    dummy = ctypes.c_void_p(16)     # never dereferenced by the host-side shape walk
    assert L.fv_plan_add_conv1d(h, 0, 2, -1, -1, -1, -1, dummy, None, 80, 32, 7, 1, 3, 0, 1.0, 1.0, 0, 0.1) == 0
    assert L.fv_plan_add_conv_transpose1d(h, 2, 3, 4, dummy, None, 32, 16, 16, 10, 5, 0, 1.0, 0, 0.1) == 0
    assert L.fv_plan_add_conv_transpose1d(h, 4, 2, -1, dummy, None, 16, 8, 16, 6, 3, 0, 1.0, 0, 0.01) == 0
    assert L.fv_plan_add_conv1d(h, 2, 3, -1, -1, -1, -1, dummy, None, 8, 4, 7, 1, 3, 0, 1.0, 1.0, 1, 1.0) == 0
    assert L.fv_plan_add_pqmf_synthesis(h, 3, 1, dummy, 4, 63) == 0
    c, n = ctypes.c_int(), ctypes.c_int64()
    assert L.fv_plan_output_shape(h, 100, ctypes.byref(c), ctypes.byref(n)) == 0
    assert (c.value, n.value) == (1, 4 * (60 * 100 - 20))
    assert L.fv_plan_workspace_bytes(h, 2, 100) > 0
    assert L.fv_plan_num_ops(h) == 5
    # conv_post + PQMF in one launch writes S * T samples per utterance: only a 'same' conv fits its output
    assert L.fv_plan_add_conv_post_pqmf(h, 2, 3, dummy, None, 8, 4, 7, 4, 1.0, 1, dummy, 63) != 0
    assert b"pad = (k - 1) / 2" in L.fv_last_error() and L.fv_plan_num_ops(h) == 5
    # UpsampleLayer: rate*T + 2*pad - (k-1)
    u = L.fv_plan_create(8)
    assert L.fv_plan_add_upsample_conv1d(u, 0, 1, -1, dummy, None, 8, 4, 16, 8, 8, 0.1, 0, 1.0) == 0
    assert L.fv_plan_output_shape(u, 100, ctypes.byref(c), ctypes.byref(n)) == 0
    assert (c.value, n.value) == (4, 801)
    L.fv_plan_destroy(u)
    # channel mismatch is caught at shape-inference time with a message
    assert L.fv_plan_add_conv1d(h, 1, 5, -1, -1, -1, -1, dummy, None, 7, 4, 3, 1, 1, 0, 1.0, 1.0, 0, 1.0) == 0
    assert L.fv_plan_output_shape(h, 100, ctypes.byref(c), ctypes.byref(n)) != 0
    assert b"channels" in L.fv_last_error()
    L.fv_plan_destroy(h)
    # ResBlock pairs / split-f16 convs keep the length; a wide pair needs its scratch slot, and only a wide one
    S = _native.PAIR_SPLIT_F16
    w = L.fv_plan_create(64)
    assert L.fv_plan_add_conv1d_split_f16(w, 0, 2, -1, -1, -1, -1, dummy, None, 64, 11, 5, 0, 0.1, 1.0, 0, 1.0) == 0
    assert L.fv_plan_add_conv1d_split_f16(w, 2, 3, -1, 0, -1, -1, dummy, None, 64, 11, 1, 0, 0.1, 1.0, 0, 1.0) == 0
    assert L.fv_plan_add_resblock_pair_ex(w, 3, 1, -1, 4, 0, 2, dummy, dummy, None, None, 64, 3, 3, 0.1, 3.0, 0, 0.1, S) == 0
    assert L.fv_plan_output_shape(w, 333, ctypes.byref(c), ctypes.byref(n)) == 0
    assert (c.value, n.value) == (64, 333)
    assert L.fv_plan_add_resblock_pair_ex(w, 3, 1, -1, -1, -1, -1, dummy, dummy, None, None, 64, 3, 3, 0.1, 1.0, 0, 1.0, S) != 0
    assert b"scratch" in L.fv_last_error()
    assert L.fv_plan_add_resblock_pair_ex(w, 3, 1, -1, 4, -1, -1, dummy, dummy, None, None, 16, 3, 3, 0.1, 1.0, 0, 1.0, S) != 0
    assert L.fv_plan_add_conv1d_split_f16(w, 0, 2, -1, -1, -1, -1, dummy, None, 32, 11, 5, 0, 0.1, 1.0, 0, 1.0) != 0
    assert b"64, 128, 256 or 512" in L.fv_last_error()
    # MelGAN's ResidualStack convs: reflection padding, dilation 9 with 3 taps only; no causal form
    assert L.fv_plan_add_conv1d_split_f16(w, 0, 2, -1, -1, -1, -1, dummy, None, 64, 3, 9, _native.PAD_REFLECT, 0.2, 1.0, 0, 1.0) == 0
    assert L.fv_plan_add_conv1d_split_f16(w, 0, 2, -1, -1, -1, -1, dummy, None, 64, 7, 9, 0, 0.2, 1.0, 0, 1.0) != 0
    assert b"dilation 9" in L.fv_last_error()
    assert L.fv_plan_add_conv1d_split_f16(w, 0, 2, -1, -1, -1, -1, dummy, None, 64, 3, 1, _native.PAD_CAUSAL, 0.2, 1.0, 0, 1.0) != 0
    assert b"pad_mode" in L.fv_last_error()
    L.fv_plan_destroy(w)
    # conv_post folded into the last pair: a 16-channel split-f16 pair only; the op's output becomes [B, 1, T]
    f = L.fv_plan_create(16)
    assert L.fv_plan_add_resblock_pair_ex(f, 0, 2, -1, -1, -1, -1, dummy, dummy, None, None, 16, 3, 5, 0.1, 1.0, 0, 1.0, S) == 0
    assert L.fv_plan_set_pair_output_conv(f, dummy, None, 1, 0.01, 1) == 0
    assert L.fv_plan_output_shape(f, 400, ctypes.byref(c), ctypes.byref(n)) == 0
    assert (c.value, n.value) == (1, 400)
    assert L.fv_plan_set_pair_output_conv(f, dummy, None, 1, 0.01, 1) != 0          # already folded
    assert L.fv_plan_add_resblock_pair_ex(f, 0, 2, -1, -1, -1, -1, dummy, dummy, None, None, 16, 3, 5, 0.1, 1.0, 0, 1.0,
                                          _native.PAIR_F32) == 0
    assert L.fv_plan_set_pair_output_conv(f, dummy, None, 3, 0.01, 1) != 0          # fp32 pair
    assert b"16-channel split-f16" in L.fv_last_error()
    L.fv_plan_destroy(f)
    # transposed conv with split-f16 operands: kernel = 2 strides, 128+ input channels; same length law as the fp32 op
    t = L.fv_plan_create(128)
    assert L.fv_packed_conv_transpose1d_split_floats(128, 64, 10, 5) == 5 * 1 * 8 * 2048 + 320     # (+ one float per padded row)
    assert L.fv_packed_conv_transpose1d_split_floats(256, 128, 16, 8) == 16 * 2 * 8 * 2048 + 1024
    assert L.fv_packed_conv_transpose1d_split_floats(64, 32, 6, 3) == 2 * 1 * 4 * 2048 + 128     # 96 rows -> 2 tiles; 64-channel chunks
    assert L.fv_packed_conv_transpose1d_split_floats(32, 16, 4, 2) == 2048 + 32                # its own kernel: 8 KB of A operands + 32 row scales
    assert L.fv_packed_conv_transpose1d_split_floats(32, 24, 8, 4) == 2 * 1 * 4 * 2048 + 128     # 32 channels on the general kernel: half a chunk
    assert L.fv_packed_conv_transpose1d_split_floats(48, 16, 4, 2) == 0
    assert L.fv_packed_conv_transpose1d_split_floats(128, 64, 11, 5) == 0
    assert L.fv_plan_add_conv_transpose1d_split_f16(t, 0, 1, -1, dummy, None, 128, 64, 10, 5, 3, 1, 0.1, 1.0) == 0
    assert L.fv_plan_output_shape(t, 100, ctypes.byref(c), ctypes.byref(n)) == 0
    assert (c.value, n.value) == (64, 500)
    assert L.fv_plan_add_conv_transpose1d_split_f16(t, 0, 1, -1, dummy, None, 48, 16, 4, 2, 1, 0, 0.1, 1.0) != 0
    assert b"Cin = 48" in L.fv_last_error()
    # the MRF merge of the stage in front, inside this op's window loader: only on a split-f16 transposed conv
    assert L.fv_plan_set_input_merge(t, 2, 3, 3.0) == 0
    assert L.fv_plan_set_input_merge(t, -1, 3, 3.0) != 0 and L.fv_plan_set_input_merge(t, 2, 3, 0.0) != 0
    assert L.fv_plan_output_shape(t, 100, ctypes.byref(c), ctypes.byref(n)) != 0     # slots 2, 3 are not set: caught by the shape walk
    assert b"merged input" in L.fv_last_error()
    assert L.fv_plan_set_output_offset(t, 28, -1) != 0
    L.fv_plan_destroy(t)
    # MelGAN's ResidualStack as one op: 32 / 64 / 128 channels, 3 taps, dilation 1 / 3 / 9, zero or reflection padding
    assert [L.fv_packed_residual_stack_floats(c_, 3) for c_ in (32, 64, 128, 256, 512, 16)] == \
        [5 * 1 * 32 * 32 + 64, 5 * 2 * 64 * 32 + 128, 5 * 4 * 128 * 32 + 256, 80 * 4096 + 512, 0, 0]
    assert L.fv_packed_residual_stack_floats(64, 7) == 0
    r = L.fv_plan_create(64)
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 64, 3, 9, 0.2, _native.PAD_REFLECT, 0, 1.0) == 0
    assert L.fv_plan_output_shape(r, 321, ctypes.byref(c), ctypes.byref(n)) == 0 and (c.value, n.value) == (64, 321)
    assert L.fv_plan_set_output_offset(r, 28, -1) == 0                      # (the bias-removal flows' offset rides in its epilogue)
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 64, 3, 5, 0.2, 0, 0, 1.0) != 0
    assert b"dilation 5" in L.fv_last_error()
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 512, 3, 1, 0.2, 0, 0, 1.0) != 0
    assert L.fv_plan_set_stack_two_launch(r, 2, dummy, dummy) != 0          # 64 channels: no two-launch form to carry
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 64, 3, 1, 0.2, _native.PAD_CAUSAL, 0, 1.0) != 0
    assert b"pad_mode" in L.fv_last_error()
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 64, 3, 1, 1.5, 0, 0, 1.0) != 0
    assert L.fv_plan_num_ops(r) == 1
    L.fv_plan_destroy(r)
    r = L.fv_plan_create(256)
    assert L.fv_plan_add_residual_stack_split_f16(r, 0, 1, -1, dummy, None, None, 256, 3, 3, 0.2, _native.PAD_REFLECT, _native.POST_RELU, 1.0) == 0
    assert L.fv_plan_set_stack_two_launch(r, 1, dummy, dummy) != 0 and b"aliases" in L.fv_last_error()
    assert L.fv_plan_set_stack_two_launch(r, 2, dummy, dummy) == 0
    assert L.fv_plan_output_shape(r, 100, ctypes.byref(c), ctypes.byref(n)) == 0 and (c.value, n.value) == (256, 100)
    assert L.fv_plan_workspace_bytes(r, 2, 100) >= 2 * 256 * 100 * 4        # the hidden tensor's scratch slot is sized
    L.fv_plan_destroy(r)


def test_encode_16bits_matches_reference_semantics():
    from fastvocoder_amd.audio import encode_16bits
    x = np.array([0.5, -1.0, 0.25], dtype=np.float32)
    y = encode_16bits(x, rescale_out=0.4)
    assert y.dtype == np.int16 and y.tolist() == [6553, -13106, 3276]
    assert abs(x[1] + 13106.8) < 0.1          # scaled in place, like the reference
    z = encode_16bits(np.zeros(4, np.float32))
    assert z.tolist() == [0, 0, 0, 0]         # max(0.01, peak) guard


def test_checkpoint_loader_is_restricted_unless_asked(tmp_path):
    """load_checkpoint reads what the reference's checkpoints hold (tensors, containers, the published numpy
    'pattern', bin/publish.py:71-75) with torch's restricted unpickler, and REFUSES a file that needs arbitrary code
    (no silent fall-back to the unrestricted unpickler); `unsafe=True` is the explicit opt-in."""
    import pickle
    from fastvocoder_amd.bin.synthesize import load_checkpoint
    good = tmp_path / "published.pth.tar"
    torch.save({"model": {"w": torch.arange(6.0).reshape(2, 3)}, "pattern": np.linspace(0, 1, 7).astype(np.float32),
                "step": 3}, good)
    ck = load_checkpoint(str(good), "cpu")
    assert torch.equal(ck["model"]["w"], torch.arange(6.0).reshape(2, 3)) and ck["step"] == 3
    assert isinstance(ck["pattern"], np.ndarray) and ck["pattern"].dtype == np.float32 and ck["pattern"].shape == (7,)

    # the same file as numpy 1.x wrote it (the reference's published checkpoints): the array's rebuild function is
    # spelled numpy.core.multiarray._reconstruct there, numpy._core... in a numpy 2.x file -- both load
    import zipfile
    other = tmp_path / "published_other_numpy.pth.tar"
    with zipfile.ZipFile(good) as zi, zipfile.ZipFile(other, "w", zipfile.ZIP_STORED) as zo:
        swapped = 0
        for item in zi.infolist():
            data = zi.read(item.filename)
            if item.filename.endswith("data.pkl"):
                a, b = b"numpy._core.multiarray", b"numpy.core.multiarray"
                if a not in data:
                    a, b = b, a
                swapped = data.count(a)
                data = data.replace(a, b)
            zo.writestr(item, data)
    assert swapped >= 1
    ck2 = load_checkpoint(str(other), "cpu")
    assert np.array_equal(ck2["pattern"], ck["pattern"]) and torch.equal(ck2["model"]["w"], ck["model"]["w"])

    class Evil:
        def __reduce__(self):
            return (os.path.join, ("never", "called"))
    bad = tmp_path / "evil.pth.tar"
    torch.save({"model": {}, "extra": Evil()}, bad)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(str(bad), "cpu")
    assert load_checkpoint(str(bad), "cpu", unsafe=True)["extra"] == os.path.join("never", "called")


def test_bench_refuses_worlds_it_cannot_place():
    """bench.py --gpus N: N must be the world torch.distributed.run started and every rank needs its own device --
    the driver's first 8-GPU run cannot silently fall back to fewer devices."""
    import bench
    assert bench.world_error(1, 1, 0, 1) is None
    assert bench.world_error(8, 8, 7, 8) is None
    assert "torch.distributed.run" in bench.world_error(8, 1, 0, 8)          # N > 1 started as one process
    assert "visible devices" in bench.world_error(8, 8, 0, 1)                # 8 ranks, one GPU
    assert "visible devices" in bench.world_error(2, 2, 1, 1)
    assert "WORLD_SIZE" in bench.world_error(4, 8, 0, 8)                      # flag and launcher disagree
    assert "WORLD_SIZE" in bench.world_error(1, 2, 0, 2)
    assert bench.world_error(2, 2, 5, 4) is not None                          # a local rank without a device
    assert bench.world_error(0, 1, 0, 1) is not None
    # the declared test mode: ranks share device 0 (tests/test_gpu_multi.py), never the default
    assert bench.world_error(2, 2, 1, 1, one_gpu=True) is None


def test_stage32_policy_follows_the_window_arithmetic():
    """hifigan.stage32_windows_fit: the 32-channel one-launch stage kernel is used where its fixed 384-column windows are at
    least 65 % full -- the shapes measured in DESIGN.md section 4.3 fall on the sides they were measured on."""
    from fastvocoder_amd.generator.hifigan import stage32_windows_fit as fit
    cus = 256
    # HiFi-GAN light's 32-channel stage has 120 samples per mel frame
    assert fit(1 * 560 * 120, cus)           # 263 columns per block: one run-in window (264 final columns)
    assert not fit(1 * 1000 * 120, cus)      # 469: a second window, 63 % empty  -> pair launches
    assert not fit(1 * 700 * 120, cus)       # 329: the second window holds 65 columns
    assert not fit(1 * 250 * 120, cus)       # 235 blocks of 128 columns: a third of a window each
    assert fit(4 * 1000 * 120, cus)          # 1875 = 264 + 5 x 324 - 9: six full windows
    assert not fit(4 * 500 * 120, cus)       # 938: four windows for 2.6
    assert fit(16 * 1000 * 120, cus) and fit(64 * 1000 * 120, cus)
    assert not fit(0, cus) and fit(264, 1) and not fit(265, 1)
    # the limit for many windows: 324 / 384 of every further window is final
    assert fit(10 ** 7, cus)


def test_block_schedules_cover_every_item_exactly_once():
    """The tables the fused-pair kernels index with blockIdx (csrc/convh_launch.hip pair_schedule / pair_cut_schedule, through the
    host-only test hook fv_debug_pair_schedule): seeded random shapes plus the headline's.
    Longest-processing-time-first schedule: every member's items are handed out as consecutive ranges in block order that
    tile [0, n_items) exactly; no block exceeds mean + the largest item + one member switch (the greedy bound).
    Contiguous cut: the share starts are non-decreasing from 0, inside the item sequence, and a share costs at most the
    mean + one item."""
    rng = np.random.RandomState(1234)
    shapes = [([149, 138, 130], [49, 33, 17], 256, True), ([149, 138], [49, 33], 256, False),
              ([339, 328, 318], [49, 33, 17], 256, True), ([1, 1, 1], [49, 33, 17], 3, True), ([5], [17], 4, True)]
    for _ in range(60):
        nm = int(rng.randint(2, 4))
        nblk = int(rng.choice([2, 3, 7, 64, 200, 256]))
        cost = sorted((int(c) for c in rng.randint(5, 100, size=nm)), reverse=True)
        items = [int(v) for v in rng.randint(1, 2 * nblk + 2, size=nm)]
        shapes.append((items, cost, nblk, bool(rng.randint(0, 2))))
    seen_lpt = 0
    for items, cost, nblk, three in shapes:
        on, shares = _native.debug_pair_schedule(items, cost, nblk, mode=0, three_members=three)
        assert on in (0, 1), (items, cost, nblk, on)
        if on == 1:
            seen_lpt += 1
            at = [0] * len(items)
            load = []
            for b in range(nblk):
                t, members = 0, 0
                for m, (lo, cnt) in enumerate(shares[b]):
                    if cnt:
                        assert lo == at[m], (items, cost, nblk, b, m)
                        at[m] += cnt
                        t += cnt * cost[m]
                        members += 1
                load.append((t, members))
            assert at == items, (items, cost, nblk, at)
            mean = sum(n * c for n, c in zip(items, cost)) / nblk
            worst = max(t + 4 * max(0, k - 1) for t, k in load)        # Tuning::sched_switch = 4 per extra member
            assert worst <= mean + max(cost) + 4 * len(items), (items, cost, nblk, worst, mean)
        on, starts = _native.debug_pair_schedule(items, cost, nblk, mode=1)
        assert on == 2
        total_items = sum(items)
        assert starts[0] == 0 and all(a <= b for a, b in zip(starts, starts[1:])) and starts[-1] <= total_items
        seq = [c for n, c in zip(items, cost) for _ in range(n)]
        mean = sum(seq) / nblk
        bounds = starts + [total_items]
        for i in range(nblk):
            assert sum(seq[bounds[i]:bounds[i + 1]]) <= mean + max(cost) + 1e-9, (items, cost, nblk, i)
    assert seen_lpt >= 10
    # the headline's 128-channel launch (HiFi-GAN light, 8000 columns): 417 items on 256 blocks, makespan 70 against a mean of 54.9
    on, shares = _native.debug_pair_schedule([149, 138, 130], [49, 33, 17], 256, mode=0, three_members=True)
    assert on == 1
    worst = max(sum(cnt * c for (lo, cnt), c in zip(sh, [49, 33, 17])) + 4 * max(0, sum(1 for lo, cnt in sh if cnt) - 1) for sh in shares)
    assert worst == 70
    with pytest.raises(_native.NativeError):
        _native.debug_pair_schedule([1, 2, 3, 4], [1, 1, 1, 1], 8)
