"""numpy/C restatement of FastVocoder's four generator graphs -- TEST
INFRASTRUCTURE ONLY (see oracle/fv_oracle.c header for the import rule and the
parity pin: tests/golden/*.npz generated from the imported reference by
tests/golden/make_golden.py).

Every function takes the reference's constructor kwargs (the conf/*.yaml keys)
and a state dict keyed exactly like the reference's ``state_dict()`` (with or
without weight norm attached: ``*.weight_g``/``*.weight_v`` or ``*.weight``)
holding numpy arrays, and returns numpy fp32.
"""
import numpy as np

from . import ops

LRELU_SLOPE = 0.1          # model/generator/modules.py:9
DEFAULT_LRELU = 0.01       # F.leaky_relu default, hifigan.py:104 / multiband_hifigan.py:113
MELGAN_SLOPE = 0.2         # melgan.py:30, modules.py:329


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def weight_of(sd, prefix):
    """Folded conv weight for ``prefix`` (hifigan.py:58-76 weight-norm lifecycle)."""
    if prefix + ".weight" in sd:
        return _np(sd[prefix + ".weight"]).astype(np.float32)
    return ops.weight_norm_fold(_np(sd[prefix + ".weight_v"]), _np(sd[prefix + ".weight_g"]))


def bias_of(sd, prefix):
    b = sd.get(prefix + ".bias")
    return None if b is None else _np(b).astype(np.float32)


def get_padding(kernel_size, dilation=1):
    """modules.py:186-187"""
    return int((kernel_size * dilation - dilation) / 2)


# --------------------------------------------------------------------------
# HiFi-GAN family
# --------------------------------------------------------------------------

def resblock1(x, sd, prefix, k, dilations):
    """ResBlock1.forward, modules.py:223-230."""
    for m, d in enumerate(dilations):
        xt = ops.conv1d(x, weight_of(sd, f"{prefix}.convs1.{m}"), bias_of(sd, f"{prefix}.convs1.{m}"),
                        dil=d, pad=get_padding(k, d), pre_slope=LRELU_SLOPE)
        xt = ops.conv1d(xt, weight_of(sd, f"{prefix}.convs2.{m}"), bias_of(sd, f"{prefix}.convs2.{m}"),
                        dil=1, pad=get_padding(k, 1), pre_slope=LRELU_SLOPE)
        x = xt + x
    return x


def resblock2(x, sd, prefix, k, dilations):
    """ResBlock2.forward, modules.py:247-252."""
    for m, d in enumerate(dilations):
        xt = ops.conv1d(x, weight_of(sd, f"{prefix}.convs.{m}"), bias_of(sd, f"{prefix}.convs.{m}"),
                        dil=d, pad=get_padding(k, d), pre_slope=LRELU_SLOPE)
        x = xt + x
    return x


def upsample_layer(x, sd, prefix, rate, k, pre_slope):
    """UpsampleLayer.forward (modules.py:160-177): nearest x rate then Conv1d(k, pad k//2).
    The activation the caller applies before it commutes with nearest-repeat."""
    x = np.repeat(x, rate, axis=2)
    return ops.conv1d(x, weight_of(sd, prefix + ".conv"), bias_of(sd, prefix + ".conv"),
                      dil=1, pad=k // 2, pre_slope=pre_slope)


def hifigan_trunk(x, sd, cfg, taps=None):
    """Shared body of HiFiGANGenerator.forward (hifigan.py:92-106) and
    MultiBandHiFiGANGenerator.forward (multiband_hifigan.py:101-115); x [B,80,T].
    Returns the tanh output [B, Cpost, T']."""
    ks = cfg["resblock_kernel_sizes"]
    ds = cfg["resblock_dilation_sizes"]
    rates = cfg["upsample_rates"]
    uks = cfg["upsample_kernel_sizes"]
    rtype = str(cfg.get("resblock_type", "1"))
    x = ops.conv1d(x, weight_of(sd, "conv_pre"), bias_of(sd, "conv_pre"), pad=3)
    nk = len(ks)
    for i, (u, k) in enumerate(zip(rates, uks)):
        if cfg.get("transposedconv", True):
            x = ops.conv_transpose1d(x, weight_of(sd, f"ups.{i}"), bias_of(sd, f"ups.{i}"),
                                     stride=u, pad=u // 2 + u % 2, out_pad=u % 2,
                                     pre_slope=LRELU_SLOPE)
        else:
            x = upsample_layer(x, sd, f"ups.{i}", u, k, LRELU_SLOPE)
        if taps is not None:
            taps.append(x.copy())
        xs = None
        for j in range(nk):
            rb = resblock1 if rtype == "1" else resblock2
            r = rb(x, sd, f"resblocks.{i * nk + j}", ks[j], ds[j])
            xs = r if xs is None else xs + r           # hifigan.py:99-102, in order
        x = (xs / np.float32(nk)).astype(np.float32)    # hifigan.py:103 true division
    x = ops.conv1d(x, weight_of(sd, "conv_post"), bias_of(sd, "conv_post"), pad=3,
                   pre_slope=DEFAULT_LRELU)             # hifigan.py:104-105
    return ops.tanh(x)                                  # hifigan.py:106


def hifigan_forward(x, sd, cfg):
    """HiFiGANGenerator.forward: [B,80,T] -> [B, prod(rates)*T] (hifigan.py:108)."""
    return hifigan_trunk(x, sd, cfg)[:, 0, :]


def hifigan_inference(c, sd, cfg):
    """HiFiGANGenerator.inference: [T,80] -> squeeze()d 1-D (hifigan.py:110-129)."""
    c = np.asarray(c, dtype=np.float32)
    return np.squeeze(hifigan_trunk(c.T[None], sd, cfg))


def multiband_forward(x, sd, cfg):
    """MultiBandHiFiGANGenerator.forward returns the sub-bands [B,4,T'] (multiband_hifigan.py:116)."""
    return hifigan_trunk(x, sd, cfg)


def multiband_inference(c, sd, cfg):
    """MultiBandHiFiGANGenerator.inference (multiband_hifigan.py:118-137): + pqmf.synthesis."""
    c = np.asarray(c, dtype=np.float32)
    sub = hifigan_trunk(c.T[None], sd, cfg)
    h = _np(sd["pqmf.synthesis_filter"]).astype(np.float32)[0] if "pqmf.synthesis_filter" in sd \
        else pqmf_filters()[1].astype(np.float32)
    return np.squeeze(ops.pqmf_synthesis(sub, h))


# --------------------------------------------------------------------------
# PQMF filter design
# --------------------------------------------------------------------------

def design_prototype_filter(taps=62, cutoff_ratio=0.142, beta=9.0):
    """pqmf.py:15-48 (np.kaiser == scipy.signal.kaiser to 3e-17)."""
    n = np.arange(taps + 1) - 0.5 * taps
    omega_c = np.pi * cutoff_ratio
    with np.errstate(invalid="ignore", divide="ignore"):
        h_i = np.sin(omega_c * n) / (np.pi * n)
    h_i[taps // 2] = np.cos(0) * cutoff_ratio
    return h_i * np.kaiser(taps + 1, beta)


def pqmf_filters(subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
    """(analysis [S,taps+1], synthesis [S,taps+1]) float64, pqmf.py:76-88."""
    h = design_prototype_filter(taps, cutoff_ratio, beta)
    n = np.arange(taps + 1) - (taps / 2)
    ha = np.zeros((subbands, taps + 1))
    hs = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        ph = (2 * k + 1) * (np.pi / (2 * subbands)) * n
        ha[k] = 2 * h * np.cos(ph + (-1) ** k * np.pi / 4)
        hs[k] = 2 * h * np.cos(ph - (-1) ** k * np.pi / 4)
    return ha, hs


# --------------------------------------------------------------------------
# MelGAN family
# --------------------------------------------------------------------------

def residual_stack(x, sd, prefix, k, d, causal=False):
    """ResidualStack.forward (modules.py:372-382): stack(c) + skip_layer(c).  With
    ``use_causal_conv`` the dilated conv is a CausalConv1d (modules.py:273-294): the
    configured pad module -- ReflectionPad1d, both sides -- by (k-1)*d, valid conv, then
    ``[:, :, :T]``; it sits at stack.1 (its conv at .conv) and the 1x1 at stack.3."""
    if causal:
        h = ops.conv1d(x, weight_of(sd, prefix + ".stack.1.conv"), bias_of(sd, prefix + ".stack.1.conv"),
                       dil=d, pad=(k - 1) * d, pad_mode=ops.PAD_REFLECT, pre_slope=MELGAN_SLOPE)
        h = np.ascontiguousarray(h[:, :, : x.shape[2]])
        pw = prefix + ".stack.3"
    else:
        h = ops.conv1d(x, weight_of(sd, prefix + ".stack.2"), bias_of(sd, prefix + ".stack.2"),
                       dil=d, pad=(k - 1) // 2 * d, pad_mode=ops.PAD_REFLECT, pre_slope=MELGAN_SLOPE)
        pw = prefix + ".stack.4"
    h = ops.conv1d(h, weight_of(sd, pw), bias_of(sd, pw), pre_slope=MELGAN_SLOPE)
    s = ops.conv1d(x, weight_of(sd, prefix + ".skip_layer"), bias_of(sd, prefix + ".skip_layer"))
    return h + s


def batchnorm_eval(x, sd, prefix, eps=1e-5):
    """torch.nn.BatchNorm1d in eval mode: (x - running_mean) / sqrt(running_var + eps) * weight + bias
    per channel, in float64 then rounded once to fp32."""
    g, b = (_np(sd[prefix + k]).astype(np.float64)[None, :, None] for k in (".weight", ".bias"))
    m, v = (_np(sd[prefix + k]).astype(np.float64)[None, :, None] for k in (".running_mean", ".running_var"))
    return ((x.astype(np.float64) - m) / np.sqrt(v + eps) * g + b).astype(np.float32)


def last_linear(x, sd, prefix):
    """LastLinear.forward (modules.py:125-132): act, bn_1, linear_1, act, bn_2, linear_2."""
    for n in ("1", "2"):
        x = batchnorm_eval(ops.lrelu(x, MELGAN_SLOPE), sd, f"{prefix}.bn_{n}")
        x = ops.conv1d(x, weight_of(sd, f"{prefix}.linear_{n}"), bias_of(sd, f"{prefix}.linear_{n}"))
    return x


def melgan_trunk(x, sd, cfg, with_last=True, taps=None):
    """The ``melgan`` Sequential of MelGANGenerator (melgan.py:66-113) /
    BasisMelGANGenerator (basis_melgan.py:72-124) up to (excluding) the final
    Tanh/ReLU.  Sequential indices follow the reference's layer list."""
    K = cfg.get("kernel_size", 7)
    scales = cfg["upsample_scales"]
    stacks = cfg.get("stacks", 3)
    sk = cfg.get("stack_kernel_size", 3)
    idx = 1
    x = ops.conv1d(x, weight_of(sd, f"melgan.{idx}"), bias_of(sd, f"melgan.{idx}"),
                   pad=(K - 1) // 2, pad_mode=ops.PAD_REFLECT)
    idx += 1
    for s in scales:
        idx += 1  # the activation module
        if cfg.get("transposedconv", True):
            x = ops.conv_transpose1d(x, weight_of(sd, f"melgan.{idx}"), bias_of(sd, f"melgan.{idx}"),
                                     stride=s, pad=s // 2 + s % 2, out_pad=s % 2,
                                     pre_slope=MELGAN_SLOPE)
        else:
            x = upsample_layer(x, sd, f"melgan.{idx}", s, 2 * s + 1, MELGAN_SLOPE)
        if taps is not None:
            taps.append(x.copy())
        idx += 1
        for j in range(stacks):
            x = residual_stack(x, sd, f"melgan.{idx}", sk, sk ** j, cfg.get("use_causal_conv", False))
            idx += 1
    if not with_last and cfg.get("lastlinear", False):
        x = last_linear(x, sd, f"melgan.{idx}")          # Basis-MelGAN's optional head
    if with_last:
        # LastLayer (modules.py:76-89)
        x = ops.conv1d(x, weight_of(sd, f"melgan.{idx}.conv"), bias_of(sd, f"melgan.{idx}.conv"),
                       pad=(K - 1) // 2, pad_mode=ops.PAD_REFLECT, pre_slope=MELGAN_SLOPE)
    return x


def melgan_forward(x, sd, cfg):
    """MelGANGenerator.forward (melgan.py:125-136)."""
    return ops.tanh(melgan_trunk(x, sd, cfg))[:, 0, :]


def melgan_inference(c, sd, cfg):
    """MelGANGenerator.inference (melgan.py:172-185)."""
    c = np.asarray(c, dtype=np.float32)
    return np.squeeze(ops.tanh(melgan_trunk(c.T[None], sd, cfg)))


def basis_weight(x, sd, cfg):
    """trunk + ReLU (basis_melgan.py:120-121), native [B,C,F] layout."""
    return np.maximum(melgan_trunk(x, sd, cfg, with_last=False), 0).astype(np.float32)


def basis_inference(c, sd, cfg):
    """BasisMelGANGenerator.inference (basis_melgan.py:196-208): (F-1)*L/2+L samples."""
    c = np.asarray(c, dtype=np.float32)
    L = cfg.get("L", 30)
    wt = basis_weight(c.T[None], sd, cfg)
    W = _np(sd["basis_signal.layer.weight"]).astype(np.float32)
    return np.squeeze(ops.basis_ola(wt, W, L // 2))


def basis_forward(x, sd, cfg):
    """BasisMelGANGenerator.forward (basis_melgan.py:140-162): two passes, truncation,
    returns (est - zero_est [B, F*L/2], weight - zero_weight [B,F,C])."""
    L = cfg.get("L", 30)
    W = _np(sd["basis_signal.layer.weight"]).astype(np.float32)
    x = np.asarray(x, dtype=np.float32)
    zw = basis_weight(np.zeros_like(x), sd, cfg)
    zs = ops.basis_ola(zw, W, L // 2)[:, : zw.shape[2] * (L // 2)]
    w = basis_weight(x, sd, cfg)
    s = ops.basis_ola(w, W, L // 2)[:, : w.shape[2] * (L // 2)]
    return s - zs, (w - zw).transpose(0, 2, 1)


FORWARD = {"hifigan": hifigan_forward, "multiband-hifigan": multiband_forward,
           "melgan": melgan_forward, "basis-melgan": basis_forward}
INFERENCE = {"hifigan": hifigan_inference, "multiband-hifigan": multiband_inference,
             "melgan": melgan_inference, "basis-melgan": basis_inference}


def synthesize(model_name, mel, sd, cfg):
    """Synthesizer.synthesize (bin/synthesize.py:74-80): (est, est - bias, bias), mel [T,80]."""
    inf = INFERENCE[model_name]
    mel = np.asarray(mel, dtype=np.float32)
    bias = inf(np.zeros_like(mel), sd, cfg)
    est = inf(mel, sd, cfg)
    return est, est - bias, bias
