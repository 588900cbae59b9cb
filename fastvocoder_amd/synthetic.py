"""Seeded synthetic checkpoints and mels for the four generators.

There is no trained checkpoint in the reference tree (only a URL,
/root/reference/README.md:18) and no network, so benchmarks and parity tests run
on random-init weights of the right architecture.  This module derives the
reference's ``state_dict`` key/shape list from the yaml kwargs alone
(``*.weight_g / *.weight_v / *.bias`` with weight norm
attached, the wire format of bin/train.py:235-247) and fills it from a frozen
``numpy.random.RandomState`` stream, so the same bytes are regenerated on any
box.  Weights are gain-calibrated so the networks stay input-sensitive (with
the default init MelGAN's output moves ~1e-5 when the mel changes, which would
make a 1e-4 parity check vacuous).
"""
import numpy as np

from .generator.pqmf import design_pqmf_filters

# per-model multiplicative gain on the uniform init bound (calibrated so the
# output has std ~0.1-0.5 and depends on the mel; see tests/golden/make_golden.py)
DEFAULT_GAIN = {"hifigan": 1.9, "multiband-hifigan": 1.6, "melgan": 1.7, "basis-melgan": 1.9}


def _conv(spec, prefix, cout, cin, k, bias, wn):
    """torch.nn.Conv1d(cin, cout, k) keys; weight norm g is per OUTPUT channel."""
    if not wn:
        spec.append((prefix + ".weight", (cout, cin, k), ("v", cin * k)))
    if bias:
        spec.append((prefix + ".bias", (cout,), ("b", cin * k)))
    if wn:  # weight norm re-registers g and v after the bias
        spec.append((prefix + ".weight_g", (cout, 1, 1), ("g", prefix)))
        spec.append((prefix + ".weight_v", (cout, cin, k), ("v", cin * k)))


def _batchnorm(spec, prefix, c):
    """torch.nn.BatchNorm1d(c) keys, with non-trivial statistics so the fold is exercised."""
    spec.append((prefix + ".weight", (c,), ("u", 0.7, 1.3)))
    spec.append((prefix + ".bias", (c,), ("u", -0.2, 0.2)))
    spec.append((prefix + ".running_mean", (c,), ("u", -0.3, 0.3)))
    spec.append((prefix + ".running_var", (c,), ("u", 0.5, 1.5)))
    spec.append((prefix + ".num_batches_tracked", (), ("count",)))


def _convT(spec, prefix, cin, cout, k, stride, bias, wn):
    """torch.nn.ConvTranspose1d(cin, cout, k) keys; weight [cin,cout,k], g per INPUT channel."""
    fan = max(1, cin * k // stride)
    if not wn:
        spec.append((prefix + ".weight", (cin, cout, k), ("v", fan)))
    if bias:
        spec.append((prefix + ".bias", (cout,), ("b", fan)))
    if wn:
        spec.append((prefix + ".weight_g", (cin, 1, 1), ("g", prefix)))
        spec.append((prefix + ".weight_v", (cin, cout, k), ("v", fan)))


def state_dict_spec(model_name, cfg, weight_norm=True):
    """[(key, shape, kind)] in the reference's registration order."""
    spec = []
    if model_name in ("hifigan", "multiband-hifigan"):
        bias = cfg.get("bias", True)
        c0 = cfg["upsample_initial_channel"]
        ks, ds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
        _conv(spec, "conv_pre", c0, 80, 7, bias, weight_norm)
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
            if cfg.get("transposedconv", True):
                _convT(spec, f"ups.{i}", cin, cout, k, u, bias, weight_norm)
            else:
                _conv(spec, f"ups.{i}.conv", cout, cin, k, bias, weight_norm)
        ch = c0
        for i in range(len(cfg["upsample_rates"])):
            ch = c0 // 2 ** (i + 1)
            for j, (k, d) in enumerate(zip(ks, ds)):
                p = f"resblocks.{i * len(ks) + j}"
                if str(cfg.get("resblock_type", "1")) == "1":
                    for m in range(3):
                        _conv(spec, f"{p}.convs1.{m}", ch, ch, k, bias, weight_norm)
                    for m in range(3):
                        _conv(spec, f"{p}.convs2.{m}", ch, ch, k, bias, weight_norm)
                else:
                    for m in range(2):
                        _conv(spec, f"{p}.convs.{m}", ch, ch, k, bias, weight_norm)
        _conv(spec, "conv_post", 4 if model_name == "multiband-hifigan" else 1, ch, 7, bias, weight_norm)
        if model_name == "multiband-hifigan":
            spec.append(("pqmf.analysis_filter", (4, 1, 63), ("pqmf", "analysis")))
            spec.append(("pqmf.synthesis_filter", (1, 4, 63), ("pqmf", "synthesis")))
            spec.append(("pqmf.updown_filter", (4, 4, 4), ("pqmf", "updown")))
        return spec
    if model_name in ("melgan", "basis-melgan"):
        causal = cfg.get("use_causal_conv", False)
        bias = cfg.get("bias", True)
        wn = weight_norm and cfg.get("use_weight_norm", True)
        K = cfg.get("kernel_size", 7)
        ch = cfg["channels"]
        sk, stacks = cfg.get("stack_kernel_size", 3), cfg.get("stacks", 3)
        idx = 1
        _conv(spec, f"melgan.{idx}", ch[0], cfg.get("in_channels", 80), K, bias, wn)
        idx += 1
        for i, s in enumerate(cfg["upsample_scales"]):
            idx += 1
            if cfg.get("transposedconv", True):
                _convT(spec, f"melgan.{idx}", ch[i], ch[i + 1], 2 * s, s, bias, wn)
            else:
                _conv(spec, f"melgan.{idx}.conv", ch[i + 1], ch[i], 2 * s + 1, bias, wn)
            idx += 1
            for _ in range(stacks):
                c = ch[i + 1]
                if causal:      # CausalConv1d at stack.1 (owns .conv), 1x1 at stack.3
                    _conv(spec, f"melgan.{idx}.stack.1.conv", c, c, sk, bias, wn)
                    _conv(spec, f"melgan.{idx}.stack.3", c, c, 1, bias, wn)
                else:
                    _conv(spec, f"melgan.{idx}.stack.2", c, c, sk, bias, wn)
                    _conv(spec, f"melgan.{idx}.stack.4", c, c, 1, bias, wn)
                _conv(spec, f"melgan.{idx}.skip_layer", c, c, 1, bias, wn)
                idx += 1
        if model_name == "melgan":
            _conv(spec, f"melgan.{idx}.conv", cfg.get("out_channels", 1), ch[-1], K, bias, wn)
        else:
            if cfg.get("lastlinear", False):     # LastLinear head (modules.py:116-132)
                h, o = ch[-1], cfg.get("out_channels", 256)
                for n, cout in (("1", h), ("2", o)):
                    _batchnorm(spec, f"melgan.{idx}.bn_{n}", h)
                    _conv(spec, f"melgan.{idx}.linear_{n}", cout, h, 1, bias, wn)
            spec.append(("basis_signal.layer.weight", (cfg.get("L", 30), cfg.get("out_channels", 256)),
                         ("basis", cfg.get("out_channels", 256))))
        return spec
    raise Exception("no model find!")


def seeded_state_dict(model_name, cfg, seed=0, gain=None, weight_norm=True):
    """{key: float32 ndarray}; bit-identical for a given (model, cfg, seed, gain)."""
    gain = DEFAULT_GAIN[model_name] if gain is None else gain
    rng = np.random.RandomState(seed)
    sd, vs = {}, {}
    spec = state_dict_spec(model_name, cfg, weight_norm)
    # v / plain weights / biases first, in key order, then g from ||v||
    for key, shape, kind in sorted(spec, key=lambda e: e[0]):
        tag = kind[0]
        if tag in ("v", "b"):
            bound = gain / np.sqrt(kind[1])
            if tag == "b":
                bound *= 0.5
            sd[key] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
            if key.endswith(".weight_v"):
                vs[key[: -len(".weight_v")]] = sd[key]
        elif tag == "u":
            sd[key] = rng.uniform(kind[1], kind[2], size=shape).astype(np.float32)
        elif tag == "count":
            sd[key] = np.array(1000, dtype=np.int64)
        elif tag == "basis":
            bound = 0.35 / np.sqrt(kind[1])
            sd[key] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    for key, shape, kind in sorted(spec, key=lambda e: e[0]):
        if kind[0] == "g":
            v = vs[kind[1]].astype(np.float64)
            nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1))
            jitter = rng.uniform(0.85, 1.15, size=nrm.shape)
            sd[key] = (nrm * jitter).astype(np.float32).reshape(shape)
        elif kind[0] == "pqmf":
            ha, hs = design_pqmf_filters()
            if kind[1] == "analysis":
                sd[key] = ha.astype(np.float32)[:, None, :]
            elif kind[1] == "synthesis":
                sd[key] = hs.astype(np.float32)[None, :, :]
            else:
                u = np.zeros((4, 4, 4), np.float32)
                for k in range(4):
                    u[k, k, 0] = 1.0
                sd[key] = u
    return {k: sd[k] for k, _, _ in spec}


def seeded_mel(T, seed=0, batch=None):
    """U[0,1) fp32 mel, [T,80] (inference layout) or [B,80,T] (forward layout);
    the reference's normalised mel range is [0,1] (data/audio.py:159-160)."""
    rng = np.random.RandomState(1000 + seed)
    if batch is None:
        return rng.rand(T, 80).astype(np.float32)
    return rng.rand(batch, 80, T).astype(np.float32)
