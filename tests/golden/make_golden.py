"""Generate the committed golden vectors from the REFERENCE implementation.

Runs only in the build container (needs /root/reference; it is imported
read-only, nothing of it is copied).  Weights and mels are regenerated from
seeds (fastvocoder_amd/synthetic.py), so the fixtures hold only the reference's
OUTPUTS (plus key/shape tables and the PQMF filters):

  keys.json                       state_dict key -> shape for the six shipped yamls
  small_<tag>.npz                 shrunken configs: forward(B=3) and inference outputs, full tensors
  full_<tag>.npz                  shipped yamls: inference(T=64) output in full, T=1000
                                  strided samples + float64 sums, per-stage taps
  blocks.npz                      ResBlock1/2, ResidualStack, LastLayer, BasisSignalLayer, PQMF
  synthesize_melgan.npz           Synthesizer.synthesize triple, BASELINE config 1
  audio.npz                       data/audio.py encode_16bits: int16 samples + the in-place scaled input
  blocks_t52.npz                  ResBlock1 (16 and 32 channels) at T = 52: lengths the fused pair kernels take

Usage:  python tests/golden/make_golden.py [extra]      (extra: only the last two files)
"""
import json
import os
import sys
import types
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import scipy.signal
import scipy.signal.windows
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# the one shim the reference needs on SciPy >= 1.13
scipy.signal.kaiser = scipy.signal.windows.kaiser
for name in ("librosa", "librosa.filters", "tensorflow", "tensorboardX"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["tensorboardX"].SummaryWriter = object
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import model.generator as refgen            # noqa: E402  (the reference)
import model.generator.modules as refmod    # noqa: E402
from model.generator.pqmf import PQMF as RefPQMF  # noqa: E402

from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict, state_dict_spec  # noqa: E402
from oracle import torch_port                # noqa: E402
from tests import cases                      # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def ref_build(name, cfg):
    """bin/synthesize.py:25-68 dispatch."""
    if name in ("hifigan", "multiband-hifigan"):
        keys = ["resblock_kernel_sizes", "upsample_rates", "upsample_initial_channel", "resblock_type",
                "upsample_kernel_sizes", "resblock_dilation_sizes", "transposedconv", "bias"]
        cls = refgen.HiFiGANGenerator if name == "hifigan" else refgen.MultiBandHiFiGANGenerator
        return cls(**{k: cfg[k] for k in keys})
    if name == "melgan":
        keys = ["in_channels", "out_channels", "kernel_size", "channels", "upsample_scales",
                "stack_kernel_size", "stacks", "use_weight_norm", "use_causal_conv"]
        return refgen.MelGANGenerator(**{k: cfg[k] for k in keys})
    if name == "basis-melgan":
        keys = ["L", "in_channels", "out_channels", "kernel_size", "channels", "upsample_scales",
                "stack_kernel_size", "stacks", "use_weight_norm", "use_causal_conv", "transposedconv"]
        return refgen.BasisMelGANGenerator(
            basis_signal_weight=torch.zeros(cfg["L"], cfg["out_channels"]).float(),
            lastlinear=cfg.get("lastlinear", False), **{k: cfg[k] for k in keys})
    raise Exception("no model find!")


def loaded(name, cfg, seed=0):
    m = ref_build(name, cfg).eval()
    sd = seeded_state_dict(name, cfg, seed=seed)
    ref_sd = m.state_dict()
    spec = state_dict_spec(name, cfg)
    assert [k for k, _, _ in spec] == list(ref_sd.keys()), "key list differs from the reference"
    for k, shp, _ in spec:
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, shp, tuple(ref_sd[k].shape))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m, sd


def tonp(t):
    return t.detach().cpu().numpy()


def stats(y):
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, y.size - 1, cases.STRIDE_N).astype(np.int64)
    return dict(sum=np.float64(y.sum()), abssum=np.float64(np.abs(y).sum()),
                idx=idx, samples=y[idx].astype(np.float32), n=np.int64(y.size))


def small_fixtures(out, only=None):
    """Full-tensor fixtures of the shrunken configs (``only``: a set of tags)."""
    for tag, name, cfg in cases.SMALL:
        if only and tag not in only:
            continue
        m, sd = loaded(name, cfg, seed=7)
        rec = {}
        with torch.no_grad():
            mel = seeded_mel(cases.SMALL_T, seed=5)
            rec["inference"] = tonp(m.inference(mel))
            melb = seeded_mel(cases.SMALL_T, seed=6, batch=cases.SMALL_B)
            f = m(torch.from_numpy(melb))
            if name == "basis-melgan":
                rec["forward_src"], rec["forward_w"] = tonp(f[0]), tonp(f[1])
            else:
                rec["forward"] = tonp(f)
            # weight norm removed must not change anything
            m.remove_weight_norm()
            g = m(torch.from_numpy(melb))
            g0 = g[0] if name == "basis-melgan" else g
            f0 = f[0] if name == "basis-melgan" else f
            assert np.abs(tonp(g0) - tonp(f0)).max() <= 1e-5
        print(f"{tag:14s} inference {rec['inference'].shape} std={rec['inference'].std():.3f}")
        np.savez_compressed(os.path.join(out, f"small_{tag}.npz"), **rec)


def main():
    out = HERE
    if len(sys.argv) > 2 and sys.argv[1] == "--small-only":
        # regenerate only the named shrunken-config fixtures (leaves the rest untouched)
        return small_fixtures(out, set(sys.argv[2].split(",")))
    keys = {}
    # ---- shipped yamls ------------------------------------------------------
    for tag, name, path in cases.SHIPPED:
        cfg = cases.load_conf(path)
        m, sd = loaded(name, cfg)
        keys[path] = {k: list(v.shape) for k, v in m.state_dict().items()}
        rec = {}
        with torch.no_grad():
            mel = seeded_mel(cases.FULL_T, seed=0)
            y = tonp(m.inference(mel))
            rec["inference_T64"] = y.astype(np.float32)
            p = tonp(torch_port.inference(name, mel, sd, cfg))
            assert np.abs(p - y).max() <= 1e-6, (tag, np.abs(p - y).max())
            # batched forward, B=2 (pins batch indexing at full width)
            melb = seeded_mel(16, seed=3, batch=2)
            f = m(torch.from_numpy(melb))
            pf = torch_port.forward(name, melb, sd, cfg)
            if name == "basis-melgan":
                rec["forward_T16_src"], rec["forward_T16_w"] = tonp(f[0]), tonp(f[1])
                assert np.abs(tonp(pf[0]) - tonp(f[0])).max() <= 1e-6
            else:
                rec["forward_T16"] = tonp(f)
                assert np.abs(tonp(pf) - tonp(f)).max() <= 1e-6
            # benchmark length: samples + sums (+ input-sensitivity figure)
            mel = seeded_mel(cases.STATS_T, seed=1)
            y = tonp(m.inference(mel))
            y2 = tonp(m.inference(seeded_mel(cases.STATS_T, seed=2)))
            st = stats(y)
            for k, v in st.items():
                rec["T1000_" + k] = v
            rec["T1000_std"] = np.float64(y.std())
            rec["T1000_sens"] = np.float64(np.abs(y - y2).mean())
            # the reference's own fp32 noise floor: same graph in float64
            m64 = ref_build(name, cfg).double().eval()
            m64.load_state_dict({k: torch.from_numpy(v).double() for k, v in sd.items()})
            y64 = tonp(m64.inference(torch.from_numpy(mel).double()))
            rec["T1000_ref_fp32_noise"] = np.float64(np.abs(y64 - y).max())
            rec["T1000_samples64"] = y64.reshape(-1)[st["idx"]]
            print(f"{tag:14s} len={y.size} std={y.std():.3f} sens={rec['T1000_sens']:.4f} "
                  f"ref fp32-vs-fp64 noise={rec['T1000_ref_fp32_noise']:.2e}")
        np.savez_compressed(os.path.join(out, f"full_{tag}.npz"), **rec)
    with open(os.path.join(out, "keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    # ---- shrunken configs: full tensors --------------------------------------
    small_fixtures(out)

    # ---- blocks ---------------------------------------------------------------
    rng = np.random.RandomState(11)
    rec = {}
    with torch.no_grad():
        def fill(mod):
            for p in mod.parameters():
                p.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, size=tuple(p.shape)).astype(np.float32)))
            return mod
        x = rng.randn(2, 16, 50).astype(np.float32)
        rec["x16"] = x
        for k in (3, 7, 11):
            rb = fill(refmod.ResBlock1(16, k, (1, 3, 5)))
            rec[f"rb1_k{k}_params"] = np.concatenate([tonp(p).reshape(-1) for p in rb.parameters()])
            rec[f"rb1_k{k}_out"] = tonp(rb(torch.from_numpy(x)))
        rb = fill(refmod.ResBlock2(16, 5, (1, 3)))
        rec["rb2_params"] = np.concatenate([tonp(p).reshape(-1) for p in rb.parameters()])
        rec["rb2_out"] = tonp(rb(torch.from_numpy(x)))
        for d in (1, 3, 9):
            rs = fill(refmod.ResidualStack(kernel_size=3, channels=16, dilation=d))
            rec[f"rs_d{d}_params"] = np.concatenate([tonp(p).reshape(-1) for p in rs.parameters()])
            rec[f"rs_d{d}_out"] = tonp(rs(torch.from_numpy(x)))
        ll = fill(refmod.LastLayer(16, 1, "LeakyReLU", {"negative_slope": 0.2}, "ReflectionPad1d", 7, {}, True))
        rec["last_params"] = np.concatenate([tonp(p).reshape(-1) for p in ll.parameters()])
        rec["last_out"] = tonp(ll(torch.from_numpy(x)))
        W = rng.uniform(-0.2, 0.2, size=(30, 16)).astype(np.float32)
        bs = refmod.BasisSignalLayer(torch.from_numpy(W), L=30)
        wt = np.abs(rng.randn(2, 40, 16)).astype(np.float32)
        rec["basis_W"], rec["basis_weight"] = W, wt
        rec["basis_out"] = tonp(bs(torch.from_numpy(wt)))
        pq = RefPQMF()
        rec["pqmf_analysis_filter"] = tonp(pq.analysis_filter)
        rec["pqmf_synthesis_filter"] = tonp(pq.synthesis_filter)
        rec["pqmf_updown_filter"] = tonp(pq.updown_filter)
        sub = rng.uniform(-1, 1, size=(2, 4, 100)).astype(np.float32)
        rec["pqmf_sub"] = sub
        rec["pqmf_synth_out"] = tonp(pq.synthesis(torch.from_numpy(sub)))
        wav = rng.uniform(-1, 1, size=(1, 1, 4000)).astype(np.float32)
        rec["pqmf_wav"] = wav
        rec["pqmf_analysis_out"] = tonp(pq.analysis(torch.from_numpy(wav)))
        rec["pqmf_roundtrip"] = tonp(pq.synthesis(pq.analysis(torch.from_numpy(wav))))
    np.savez_compressed(os.path.join(out, "blocks.npz"), **rec)

    # ---- Synthesizer.synthesize, BASELINE config 1 ---------------------------------
    import tempfile
    from bin.synthesize import Synthesizer   # reference class (bin/synthesize.py:17-84)
    cfg = cases.load_conf("conf/melgan/original.yaml")
    sd = seeded_state_dict("melgan", cfg, seed=0)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "ckpt.pth.tar")
        torch.save({"model": {k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
        syn = Synthesizer(ck, os.path.join(ROOT, "conf/melgan/original.yaml"), "melgan")
    mel = np.random.RandomState(0).rand(80, 200)          # float64 [80,T] like resource/test.mel.npy
    est, rem, bias = syn.synthesize(mel.T)
    np.savez_compressed(os.path.join(out, "synthesize_melgan.npz"), est=tonp(est), remove=tonp(rem),
                        bias=tonp(bias))
    print("synthesize", tonp(est).shape, float(tonp(est).std()), float(tonp(bias).std()))


def extra_fixtures(out):
    """Fixtures added in round 2; kept apart so that the round-1 files need not be regenerated."""
    import data.audio as refaudio            # reference data/audio.py (librosa / tensorflow stubbed above)
    rng = np.random.RandomState(21)
    rec = {}
    cases_ = [("unit", rng.uniform(-1, 1, 5000), 1.0), ("quiet_floor", rng.uniform(-0.004, 0.004, 3000), 1.0),
              ("rescale04", rng.randn(4801) * 0.3, 0.4), ("big", rng.randn(2000) * 7.0, 0.4),
              ("zeros", np.zeros(64), 0.4)]
    for tag, x, rescale in cases_:
        x = x.astype(np.float32)
        rec[f"{tag}_in"] = x.copy()
        rec[f"{tag}_rescale"] = np.float32(rescale)
        rec[f"{tag}_int16"] = refaudio.encode_16bits(x, rescale_out=rescale)      # mutates x (data/audio.py:13)
        rec[f"{tag}_scaled"] = x
    np.savez_compressed(os.path.join(out, "audio.npz"), **rec)

    rng = np.random.RandomState(12)
    rec = {}
    with torch.no_grad():
        for ch in (16, 32):
            x = rng.randn(2, ch, 52).astype(np.float32)
            rec[f"x{ch}"] = x
            for k in (3, 7, 11):
                rb = refmod.ResBlock1(ch, k, (1, 3, 5))
                for p in rb.parameters():
                    p.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, size=tuple(p.shape)).astype(np.float32)
                                             / np.float32(np.sqrt(ch / 16.0))))
                rec[f"rb1_c{ch}_k{k}_params"] = np.concatenate([tonp(p).reshape(-1) for p in rb.parameters()])
                rec[f"rb1_c{ch}_k{k}_out"] = tonp(rb(torch.from_numpy(x)))
    np.savez_compressed(os.path.join(out, "blocks_t52.npz"), **rec)

    # CausalConvTranspose1d (reference modules.py:297-317): ConvTranspose1d(k, stride, pad 0) minus its last `stride` samples
    rng = np.random.RandomState(13)
    rec = {}
    with torch.no_grad():
        for tag, cin, cout, k, s, T in (("a", 24, 12, 16, 8, 37), ("b", 64, 32, 6, 3, 50), ("c", 8, 4, 4, 2, 9)):
            m = refmod.CausalConvTranspose1d(cin, cout, k, s)
            for p in m.parameters():
                p.copy_(torch.from_numpy(rng.uniform(-0.2, 0.2, size=tuple(p.shape)).astype(np.float32)))
            x = rng.randn(2, cin, T).astype(np.float32)
            rec[f"{tag}_shape"] = np.array([cin, cout, k, s], dtype=np.int64)
            rec[f"{tag}_x"] = x
            rec[f"{tag}_weight"] = tonp(m.deconv.weight)
            rec[f"{tag}_bias"] = tonp(m.deconv.bias)
            rec[f"{tag}_out"] = tonp(m(torch.from_numpy(x)))
    np.savez_compressed(os.path.join(out, "causal_convt.npz"), **rec)
    print("extra fixtures written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra_fixtures(HERE)
    else:
        main()
        extra_fixtures(HERE)
