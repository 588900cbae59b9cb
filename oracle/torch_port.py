"""PyTorch-CPU functional port of the reference generator graphs -- TEST
INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/).

The reference's CPU path IS these ATen ops (conv1d, conv_transpose1d,
leaky_relu, reflection_pad1d, linear, index_add_), so
timing this port on the GPU box's host cores is the "reference CPU path timed
beside it" that BASELINE.md section 4 describes.  It is validated against the
imported reference in tests/golden/make_golden.py (<= 1e-6) and against the C
restatement in tests/test_oracle_golden.py.  The op sequence is deliberately
un-fused, one ATen call per reference call site.
"""
import math

import torch
import torch.nn.functional as F

from .generators import get_padding, LRELU_SLOPE, MELGAN_SLOPE


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(a)


def weight_of(sd, prefix):
    if prefix + ".weight" in sd:
        return _t(sd[prefix + ".weight"]).float()
    v, g = _t(sd[prefix + ".weight_v"]).float(), _t(sd[prefix + ".weight_g"]).float()
    return torch._weight_norm(v, g, 0)


def bias_of(sd, prefix):
    b = sd.get(prefix + ".bias")
    return None if b is None else _t(b).float()


def fold_state_dict(sd):
    """Fold weight norm once (what remove_weight_norm leaves behind)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_v"):
            p = k[: -len(".weight_v")]
            out[p + ".weight"] = weight_of(sd, p)
        elif k.endswith(".weight_g"):
            continue
        else:
            out[k] = _t(v).float() if _t(v).is_floating_point() else _t(v)
    return out


def _resblock1(x, sd, p, k, dil):
    for m, d in enumerate(dil):                                   # modules.py:223-230
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, weight_of(sd, f"{p}.convs1.{m}"), bias_of(sd, f"{p}.convs1.{m}"),
                      padding=get_padding(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, weight_of(sd, f"{p}.convs2.{m}"), bias_of(sd, f"{p}.convs2.{m}"),
                      padding=get_padding(k, 1))
        x = xt + x
    return x


def _resblock2(x, sd, p, k, dil):
    for m, d in enumerate(dil):                                   # modules.py:247-252
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, weight_of(sd, f"{p}.convs.{m}"), bias_of(sd, f"{p}.convs.{m}"),
                      padding=get_padding(k, d), dilation=d)
        x = xt + x
    return x


def _upsample_layer(x, sd, p, rate, k):
    x = F.interpolate(x.unsqueeze(1), scale_factor=(1, rate), mode="nearest").squeeze(1)
    return F.conv1d(x, weight_of(sd, p + ".conv"), bias_of(sd, p + ".conv"), padding=k // 2)


def hifigan_trunk(x, sd, cfg):
    ks, ds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    rb = _resblock1 if str(cfg.get("resblock_type", "1")) == "1" else _resblock2
    x = F.conv1d(x, weight_of(sd, "conv_pre"), bias_of(sd, "conv_pre"), padding=3)
    nk = len(ks)
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        if cfg.get("transposedconv", True):
            x = F.conv_transpose1d(x, weight_of(sd, f"ups.{i}"), bias_of(sd, f"ups.{i}"),
                                   stride=u, padding=u // 2 + u % 2, output_padding=u % 2)
        else:
            x = _upsample_layer(x, sd, f"ups.{i}", u, k)
        xs = None
        for j in range(nk):
            r = rb(x, sd, f"resblocks.{i * nk + j}", ks[j], ds[j])
            if xs is None:
                xs = r
            else:
                xs += r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, weight_of(sd, "conv_post"), bias_of(sd, "conv_post"), padding=3)
    return torch.tanh(x)


def pqmf_synthesis(x, h_syn):
    """pqmf.py:121-135; h_syn [1,S,taps+1]."""
    S = x.shape[1]
    updown = torch.zeros(S, S, S)
    for k in range(S):
        updown[k, k, 0] = 1.0
    x = F.conv_transpose1d(x, updown * S, stride=S)
    half = (h_syn.shape[-1] - 1) // 2
    return F.conv1d(F.pad(x, (half, half)), h_syn)


def overlap_and_add(signal, frame_step):
    """modules.py:34-73 (gcd-subframe + index_add_)."""
    outer = signal.size()[:-2]
    frames, frame_length = signal.size()[-2:]
    sub = math.gcd(frame_length, frame_step)
    sstep = frame_step // sub
    spf = frame_length // sub
    osize = frame_step * (frames - 1) + frame_length
    osub = osize // sub
    ss = signal.reshape(*outer, -1, sub)
    frame = torch.arange(0, osub).unfold(0, spf, sstep).contiguous().view(-1)
    res = signal.new_zeros(*outer, osub, sub)
    res.index_add_(-2, frame, ss)
    return res.view(*outer, -1)


def _residual_stack(x, sd, p, k, d, causal=False):
    h = F.leaky_relu(x, MELGAN_SLOPE)
    if causal:                          # CausalConv1d, modules.py:273-294
        h = F.pad(h, ((k - 1) * d,) * 2, mode="reflect")
        h = F.conv1d(h, weight_of(sd, p + ".stack.1.conv"), bias_of(sd, p + ".stack.1.conv"),
                     dilation=d)[:, :, : x.size(2)]
        pw = p + ".stack.3"
    else:
        h = F.pad(h, ((k - 1) // 2 * d,) * 2, mode="reflect")
        h = F.conv1d(h, weight_of(sd, p + ".stack.2"), bias_of(sd, p + ".stack.2"), dilation=d)
        pw = p + ".stack.4"
    h = F.leaky_relu(h, MELGAN_SLOPE)
    h = F.conv1d(h, weight_of(sd, pw), bias_of(sd, pw))
    return h + F.conv1d(x, weight_of(sd, p + ".skip_layer"), bias_of(sd, p + ".skip_layer"))


def melgan_trunk(x, sd, cfg, with_last=True):
    K = cfg.get("kernel_size", 7)
    sk, stacks = cfg.get("stack_kernel_size", 3), cfg.get("stacks", 3)
    idx = 1
    x = F.conv1d(F.pad(x, ((K - 1) // 2,) * 2, mode="reflect"),
                 weight_of(sd, f"melgan.{idx}"), bias_of(sd, f"melgan.{idx}"))
    idx += 1
    for s in cfg["upsample_scales"]:
        idx += 1
        x = F.leaky_relu(x, MELGAN_SLOPE)
        if cfg.get("transposedconv", True):
            x = F.conv_transpose1d(x, weight_of(sd, f"melgan.{idx}"), bias_of(sd, f"melgan.{idx}"),
                                   stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
        else:
            x = _upsample_layer(x, sd, f"melgan.{idx}", s, 2 * s + 1)
        idx += 1
        for j in range(stacks):
            x = _residual_stack(x, sd, f"melgan.{idx}", sk, sk ** j, cfg.get("use_causal_conv", False))
            idx += 1
    if not with_last and cfg.get("lastlinear", False):   # LastLinear, modules.py:116-132
        for n in ("1", "2"):
            q = f"melgan.{idx}.bn_{n}"
            x = F.batch_norm(F.leaky_relu(x, MELGAN_SLOPE), _t(sd[q + ".running_mean"]).float(),
                             _t(sd[q + ".running_var"]).float(), _t(sd[q + ".weight"]).float(),
                             _t(sd[q + ".bias"]).float(), False, 0.1, 1e-5)
            x = F.conv1d(x, weight_of(sd, f"melgan.{idx}.linear_{n}"), bias_of(sd, f"melgan.{idx}.linear_{n}"))
    if with_last:
        x = F.leaky_relu(x, MELGAN_SLOPE)
        x = F.conv1d(F.pad(x, ((K - 1) // 2,) * 2, mode="reflect"),
                     weight_of(sd, f"melgan.{idx}.conv"), bias_of(sd, f"melgan.{idx}.conv"))
    return x


@torch.no_grad()
def forward(model_name, x, sd, cfg):
    """``Generator.forward`` semantics for x [B,80,T] (torch or numpy)."""
    x = _t(x).float()
    if model_name == "hifigan":
        return hifigan_trunk(x, sd, cfg)[:, 0, :]
    if model_name == "multiband-hifigan":
        return hifigan_trunk(x, sd, cfg)
    if model_name == "melgan":
        return torch.tanh(melgan_trunk(x, sd, cfg))[:, 0, :]
    if model_name == "basis-melgan":
        L = cfg.get("L", 30)
        W = _t(sd["basis_signal.layer.weight"]).float()

        def one(inp):
            w = torch.relu(melgan_trunk(inp, sd, cfg, with_last=False)).contiguous().transpose(1, 2)
            s = overlap_and_add(F.linear(w, W), L // 2)
            return s[:, : w.size(1) * (L // 2)], w
        zs, zw = one(torch.zeros_like(x))
        s, w = one(x)
        return s - zs, w - zw
    raise Exception("no model find!")


@torch.no_grad()
def inference(model_name, c, sd, cfg):
    """``Generator.inference`` semantics for c [T,80]."""
    x = _t(c).float().transpose(1, 0).unsqueeze(0)
    if model_name == "hifigan":
        return hifigan_trunk(x, sd, cfg).squeeze()
    if model_name == "multiband-hifigan":
        sub = hifigan_trunk(x, sd, cfg)
        return pqmf_synthesis(sub, _t(sd["pqmf.synthesis_filter"]).float()).squeeze()
    if model_name == "melgan":
        return torch.tanh(melgan_trunk(x, sd, cfg)).squeeze()
    if model_name == "basis-melgan":
        L = cfg.get("L", 30)
        w = torch.relu(melgan_trunk(x, sd, cfg, with_last=False)).contiguous().transpose(1, 2)
        return overlap_and_add(F.linear(w, _t(sd["basis_signal.layer.weight"]).float()), L // 2).squeeze()
    raise Exception("no model find!")
