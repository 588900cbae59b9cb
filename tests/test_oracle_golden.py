"""Pin the oracle (oracle/fv_oracle.c + oracle/generators.py and the ATen port
oracle/torch_port.py) against the golden vectors produced by the imported
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict
from oracle import generators as og
from oracle import ops as oo
from oracle import torch_port
from tests import cases

TOL = 2e-5  # oracle accumulates in double; the reference's own fp32 noise is ~1.5e-6 on these nets


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag,name,cfg", cases.SMALL, ids=[c[0] for c in cases.SMALL])
def test_small_configs_c_oracle(golden_dir, tag, name, cfg):
    g = _load(golden_dir, f"small_{tag}.npz")
    sd = seeded_state_dict(name, cfg, seed=7)
    y = og.INFERENCE[name](seeded_mel(cases.SMALL_T, seed=5), sd, cfg)
    assert y.shape == g["inference"].shape
    assert np.abs(y - g["inference"]).max() <= TOL
    f = og.FORWARD[name](seeded_mel(cases.SMALL_T, seed=6, batch=cases.SMALL_B), sd, cfg)
    if name == "basis-melgan":
        assert f[0].shape == g["forward_src"].shape and f[1].shape == g["forward_w"].shape
        assert np.abs(f[0] - g["forward_src"]).max() <= TOL
        assert np.abs(f[1] - g["forward_w"]).max() <= 1e-4  # pre-ReLU-scale activations, O(10)
    else:
        assert f.shape == g["forward"].shape
        assert np.abs(f - g["forward"]).max() <= TOL


@pytest.mark.parametrize("tag,name,cfg", cases.SMALL, ids=[c[0] for c in cases.SMALL])
def test_small_configs_torch_port(golden_dir, tag, name, cfg):
    g = _load(golden_dir, f"small_{tag}.npz")
    sd = seeded_state_dict(name, cfg, seed=7)
    y = torch_port.inference(name, seeded_mel(cases.SMALL_T, seed=5), sd, cfg).numpy()
    assert np.abs(y - g["inference"]).max() <= 1e-6


@pytest.mark.parametrize("tag,name,path", cases.SHIPPED, ids=[c[0] for c in cases.SHIPPED])
def test_shipped_configs_torch_port(golden_dir, tag, name, path):
    g = _load(golden_dir, f"full_{tag}.npz")
    cfg = cases.load_conf(path)
    sd = seeded_state_dict(name, cfg, seed=0)
    y = torch_port.inference(name, seeded_mel(cases.FULL_T, seed=0), sd, cfg).numpy()
    assert y.shape == g["inference_T64"].shape
    assert np.abs(y - g["inference_T64"]).max() <= 1e-6
    # output-length quirks (SURVEY 8 trap 5)
    expect = {"mb_large": 4 * (60 * cases.FULL_T - 20), "basis": 240 * cases.FULL_T + 15}
    assert y.shape[0] == expect.get(tag, 240 * cases.FULL_T)


def test_hifigan_light_full_width_c_oracle(golden_dir):
    """One full-width config through the C restatement (T=16 keeps it to seconds)."""
    g = _load(golden_dir, "full_hifigan_light.npz")
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    sd = seeded_state_dict("hifigan", cfg, seed=0)
    f = og.hifigan_forward(seeded_mel(16, seed=3, batch=2), sd, cfg)
    assert f.shape == g["forward_T16"].shape
    assert np.abs(f - g["forward_T16"]).max() <= TOL


def _close(a, b, tol=TOL):
    """max-abs error relative to the tensor's scale (block outputs are O(10))."""
    return np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max()))


def _params(mod_keys, flat):
    out, off = [], 0
    for shape in mod_keys:
        n = int(np.prod(shape))
        out.append(flat[off:off + n].reshape(shape))
        off += n
    assert off == flat.size
    return out


def test_blocks_c_oracle(golden_dir):
    g = _load(golden_dir, "blocks.npz")
    x = g["x16"]
    C = 16
    for k in (3, 7, 11):
        # parameter order of the reference module: convs1.{0,1,2}.{weight,bias}, convs2...
        shapes = [(C, C, k), (C,)] * 6
        p = _params(shapes, g[f"rb1_k{k}_params"])
        sd = {}
        for m in range(3):
            sd[f"rb.convs1.{m}.weight"], sd[f"rb.convs1.{m}.bias"] = p[2 * m], p[2 * m + 1]
            sd[f"rb.convs2.{m}.weight"], sd[f"rb.convs2.{m}.bias"] = p[6 + 2 * m], p[7 + 2 * m]
        y = og.resblock1(x, sd, "rb", k, (1, 3, 5))
        assert _close(y, g[f"rb1_k{k}_out"])
    p = _params([(C, C, 5), (C,)] * 2, g["rb2_params"])
    sd = {f"rb.convs.{m}.weight": p[2 * m] for m in range(2)}
    sd.update({f"rb.convs.{m}.bias": p[2 * m + 1] for m in range(2)})
    assert _close(og.resblock2(x, sd, "rb", 5, (1, 3)), g["rb2_out"])
    for d in (1, 3, 9):
        p = _params([(C, C, 3), (C,), (C, C, 1), (C,), (C, C, 1), (C,)], g[f"rs_d{d}_params"])
        sd = {"rs.stack.2.weight": p[0], "rs.stack.2.bias": p[1], "rs.stack.4.weight": p[2],
              "rs.stack.4.bias": p[3], "rs.skip_layer.weight": p[4], "rs.skip_layer.bias": p[5]}
        assert _close(og.residual_stack(x, sd, "rs", 3, d), g[f"rs_d{d}_out"])
    p = _params([(1, C, 7), (1,)], g["last_params"])
    y = oo.conv1d(x, p[0], p[1], pad=3, pad_mode=oo.PAD_REFLECT, pre_slope=0.2)
    assert _close(y, g["last_out"])
    # basis matmul + overlap-add: the golden input is [B,F,C], the oracle takes [B,C,F]
    y = oo.basis_ola(g["basis_weight"].transpose(0, 2, 1), g["basis_W"], 15)
    assert y.shape == g["basis_out"].shape
    assert _close(y, g["basis_out"])


def test_pqmf_c_oracle(golden_dir):
    g = _load(golden_dir, "blocks.npz")
    ha, hs = og.pqmf_filters()
    assert np.abs(ha.astype(np.float32) - g["pqmf_analysis_filter"][:, 0, :]).max() == 0
    assert np.abs(hs.astype(np.float32) - g["pqmf_synthesis_filter"][0]).max() == 0
    y = oo.pqmf_synthesis(g["pqmf_sub"], g["pqmf_synthesis_filter"][0])
    assert np.abs(y - g["pqmf_synth_out"][:, 0, :]).max() <= 1e-5
    a = oo.pqmf_analysis(g["pqmf_wav"][:, 0, :], g["pqmf_analysis_filter"][:, 0, :])
    assert np.abs(a - g["pqmf_analysis_out"]).max() <= 1e-5
    # analysis -> synthesis is near-perfect reconstruction in the interior
    # (known-answer property of the filter bank, SURVEY 8 f-4: ~8.6e-4)
    r = oo.pqmf_synthesis(a, g["pqmf_synthesis_filter"][0])
    assert np.abs(r - g["pqmf_roundtrip"][:, 0, :]).max() <= 1e-5
    assert np.abs(r[0, 200:-200] - g["pqmf_wav"][0, 0, 200:-200]).max() < 2e-3


def test_synthesize_triple_torch_port(golden_dir):
    """BASELINE config 1: MelGAN original, one 80x200 mel through synthesize()."""
    g = _load(golden_dir, "synthesize_melgan.npz")
    cfg = cases.load_conf("conf/melgan/original.yaml")
    sd = torch_port.fold_state_dict(seeded_state_dict("melgan", cfg, seed=0))
    mel = np.random.RandomState(0).rand(80, 200).T
    bias = torch_port.inference("melgan", np.zeros_like(mel), sd, cfg).numpy()
    est = torch_port.inference("melgan", mel, sd, cfg).numpy()
    assert est.shape == (48000,)
    assert np.abs(est - g["est"]).max() <= 2e-6
    assert np.abs(bias - g["bias"]).max() <= 2e-6
    assert np.abs((est - bias) - g["remove"]).max() <= 4e-6


def test_weight_norm_fold_matches_torch():
    rng = np.random.RandomState(3)
    for shape in [(8, 5, 7), (6, 4, 3)]:
        v = rng.randn(*shape).astype(np.float32)
        gg = rng.rand(shape[0], 1, 1).astype(np.float32) + 0.5
        w = oo.weight_norm_fold(v, gg)
        ref = torch._weight_norm(torch.from_numpy(v), torch.from_numpy(gg), 0).numpy()
        assert np.abs(w - ref).max() <= 1e-6
