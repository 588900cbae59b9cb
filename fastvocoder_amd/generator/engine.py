"""Host-side glue between torch parameter containers and native plans.

A generator here is a tree of ordinary ``torch.nn`` modules that exist ONLY to
own parameters under the reference's names (so ``state_dict()`` /
``load_state_dict()`` / ``.to()`` / weight-norm utilities behave exactly like
the reference's) -- their own ``forward`` is never
called.  Arithmetic happens in libfastvocoder_hip.so: a module *emits* its ops
into a :class:`PlanBuilder`, which folds weight norm and packs every weight on
the GPU once, and the resulting native plan is replayed for every call until a
parameter changes.
"""
import warnings

import torch

from .. import _native
from .._native import (PAD_CAUSAL, PAD_REFLECT, PAD_ZERO, POST_NONE, POST_RELU, POST_TANH,  # noqa: F401
                       SLOT_AUX_IN0, SLOT_AUX_IN1, SLOT_IN, SLOT_NONE, SLOT_OUT, SLOT_OUT2)


def weight_norm(module):
    """torch.nn.utils.weight_norm without its deprecation chatter; keeps the
    ``weight_g`` / ``weight_v`` key names the reference's checkpoints use."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.nn.utils.weight_norm(module)


def effective_weight(conv):
    """Folded weight of a Conv1d / ConvTranspose1d container, computed on the
    GPU (fv_fold_weight_norm) when weight norm is attached."""
    if hasattr(conv, "weight_g") and hasattr(conv, "weight_v"):
        return _native.fold_weight_norm(conv.weight_v, conv.weight_g)
    return conv.weight.detach().contiguous().float()


def pair_precision(precision, channels):
    """Arithmetic of the ResBlock / ResidualStack / upsampler kernels for a layer of ``channels`` channels under the
    policy ``precision``: "split" = split-f16 operands (csrc/pairh_kernels.hpp, convh_kernels.hpp; fp32-class accuracy,
    DESIGN.md section 3.7) wherever that kernel exists, "f32" = the exact-fp32 MFMA kernels everywhere."""
    if precision == "f32":
        return _native.PAIR_F32
    return _native.PAIR_SPLIT_F16 if _native.pair_supported(channels, 3, 1, _native.PAIR_SPLIT_F16) else _native.PAIR_F32


class PlanBuilder:
    """Accumulates ops, runs the activation-hoisting pass, then emits a native
    plan.  Slots are small integers naming tensors (SLOT_IN / SLOT_OUT /
    temporaries handed out by :meth:`tmp`).

    Activation hoisting: on gfx950 plain VALU instructions do not overlap the
    fp32 MFMA (each costs matrix time), so ``conv(leaky_relu(x))`` must not apply
    the activation per operand read.  :meth:`finalize` rewrites the op list so
    that the op PRODUCING ``x`` stores ``leaky_relu(x, slope)`` -- in place when
    every consumer wants the activated tensor, or into a twin slot next to the
    raw tensor when a residual add also needs the raw one -- and the consuming
    convs read it with ``pre_slope = 1``.
    """

    def __init__(self, in_channels, precision="split", fold_post=True, guard=None):
        """``precision``: "split" | "f32" (pair_precision); ``fold_post``: conv_post may ride in the last pair's launch;
        ``guard``: the owning module's _native.GuardWord -- the pack kernels raise its weight word, the split-f16
        launches of the finished plan its activation word."""
        self.plan = _native.Plan(in_channels)
        self._next = _native.SLOT_TMP0
        self.ops = []
        self.group = 0      # non-zero: ops recorded under it are mutually independent (fv_plan_set_group)
        self._groups = 0
        self.precision = precision
        self.fold_post = bool(fold_post)
        self.guard = guard

    def tmp(self):
        s = self._next
        if s >= SLOT_AUX_IN0:
            raise _native.NativeError("plan needs more than %d tensor slots" % (SLOT_AUX_IN0 - _native.SLOT_TMP0))
        self._next += 1
        return s

    @staticmethod
    def _bias(conv):
        return None if conv.bias is None else conv.bias.detach().contiguous().float()

    def begin_group(self):
        """Ops recorded until :meth:`end_group` are mutually independent (no op reads or
        overwrites what another writes): the executor may run them as one launch."""
        self._groups += 1
        self.group = self._groups

    def end_group(self):
        self.group = 0

    def conv(self, conv, src, dst, pad=None, pad_mode=PAD_ZERO, pre_slope=1.0, res=SLOT_NONE,
             acc=SLOT_NONE, out_div=1.0, post=POST_NONE, acc2=SLOT_NONE, batchnorm=None,
             own_first=False):
        """Record ``conv`` (a torch.nn.Conv1d container): dst = epilogue(conv(bn(act(src)))).
        ``batchnorm`` (an eval-mode BatchNorm1d container applied to the conv's input) is
        folded into the weight and bias (fv_fold_batchnorm_conv)."""
        if conv.stride[0] != 1 or conv.groups != 1:
            raise _native.NativeError("only stride-1, groups-1 Conv1d layers exist on this path")
        k, d = conv.kernel_size[0], conv.dilation[0]
        if pad is None:
            pad = conv.padding[0]
        weight, bias = effective_weight(conv), self._bias(conv)
        if batchnorm is not None:
            if pad != 0:
                raise _native.NativeError("BatchNorm folds only into an unpadded conv")
            weight, bias = _native.fold_batchnorm_conv(weight, bias, batchnorm)
        self.ops.append(dict(kind="conv", group=self.group, x=src, y=dst, res=res, acc=acc,
                             acc2=acc2, pre_slope=float(pre_slope), own_first=bool(own_first),
                             packed=_native.pack_conv1d(weight), bias=bias,
                             cin=conv.in_channels, cout=conv.out_channels, k=k, dil=d, pad=pad,
                             pad_mode=pad_mode, out_div=out_div, post=post))

    def conv_sum_1x1(self, conv_a, src_a, conv_b, src_b, dst, pre_slope_a=1.0, res=SLOT_NONE,
                     post=POST_NONE):
        """dst = post(conv_a(act(src_a)) + conv_b(src_b) [+ res]) for two 1x1 Conv1d containers, as
        ONE launch: the K range of the GEMM is the concatenation of the two inputs.
        128 / 256 / 512 channels under the split-f16 policy (and ``split``): fv_conv1x1_2src_split_f16 -- ``src_a`` is
        read raw and activated on chip.  Otherwise fv_conv1d_2src_fused (fp32 MFMA): that kernel applies no input
        activation, so ``pre_slope_a`` must be absorbed by the producer of ``src_a`` -- activation hoisting does that
        whenever ``src_a`` has no consumer that needs it raw (checked in :meth:`finalize`)."""
        for c in (conv_a, conv_b):
            if c.kernel_size[0] != 1 or c.stride[0] != 1 or c.groups != 1 or c.padding[0] != 0:
                raise _native.NativeError("conv_sum_1x1: both layers must be plain 1x1 convs")
        if conv_a.out_channels != conv_b.out_channels:
            raise _native.NativeError("conv_sum_1x1: the two convs must have the same output channels")
        ch = conv_a.out_channels
        if (conv_a.in_channels == ch and conv_b.in_channels == ch and _native.conv1x1_2src_split_supported(ch)
                and self.pair_precision(ch) == _native.PAIR_SPLIT_F16):
            ba, bb = self._bias(conv_a), self._bias(conv_b)
            bias = ba if bb is None else (bb if ba is None else (ba + bb).contiguous())
            self.ops.append(dict(kind="conv2h", x=src_a, x2=src_b, y=dst, res=res, acc=SLOT_NONE, pre_slope=1.0,
                                 slope=float(pre_slope_a), split=True,
                                 packed=_native.pack_conv1x1_2src_split(effective_weight(conv_a), effective_weight(conv_b),
                                                                        self.guard),
                                 bias=bias, channels=ch, post=post))
            return
        w = torch.cat([effective_weight(conv_a), effective_weight(conv_b)], dim=1).contiguous()
        ba, bb = self._bias(conv_a), self._bias(conv_b)
        bias = ba if bb is None else (bb if ba is None else (ba + bb).contiguous())
        self.ops.append(dict(kind="conv2", x=src_a, x2=src_b, y=dst, res=res, acc=SLOT_NONE,
                             pre_slope=float(pre_slope_a), packed=_native.pack_conv1d(w), bias=bias,
                             cin1=conv_a.in_channels, cin2=conv_b.in_channels, cout=conv_a.out_channels,
                             post=post))

    def conv_sum3(self, convs, srcs, ress, tmps, dst, pre_slope=1.0, out_div=1.0, post=POST_NONE):
        """dst = post((sum_j conv_j(act(src_j)) + bias_j + res_j) / out_div) for the three last
        convs of an MRF stage in ONE launch (fv_plan_add_conv1d_sum3).  The kernel applies no
        input activation: ``pre_slope`` must be absorbed by the producers (checked in finalize)."""
        c0 = convs[0]
        for c in convs:
            if (c.stride[0] != 1 or c.groups != 1 or c.dilation[0] != 1 or c.in_channels != c0.in_channels
                    or c.out_channels != c0.in_channels or c.padding[0] != (c.kernel_size[0] - 1) // 2):
                raise _native.NativeError("conv_sum3: three undilated 'same' C->C convs are required")
        biases = [self._bias(c) for c in convs]
        bias = None if all(b is None for b in biases) else sum(b for b in biases if b is not None).contiguous()
        self.ops.append(dict(kind="sum3", x=srcs[0], xb=srcs[1], xc=srcs[2], y=dst,
                             res=ress[0], resb=ress[1], resc=ress[2], tmps=list(tmps), acc=SLOT_NONE,
                             pre_slope=float(pre_slope),
                             packed=[_native.pack_conv1d(effective_weight(c)) for c in convs], bias=bias,
                             channels=c0.in_channels, ks=[c.kernel_size[0] for c in convs],
                             out_div=float(out_div), post=post))

    def pair_precision(self, channels):
        """Arithmetic (PAIR_F32 / PAIR_SPLIT_F16) of a ``channels``-channel layer under this builder's policy."""
        return pair_precision(self.precision, channels)

    @staticmethod
    def pair_fusable(conv1, conv2, prec=None):
        """Can this (dilated conv, conv) pair of a ResBlock1 run on the fused pair kernels?"""
        k, c = conv1.kernel_size[0], conv1.in_channels
        prec = _native.PAIR_F32 if prec is None else prec
        return (all(cv.stride[0] == 1 and cv.groups == 1 and cv.in_channels == c and cv.out_channels == c
                    and cv.kernel_size[0] == k and cv.padding[0] == cv.dilation[0] * (k - 1) // 2
                    for cv in (conv1, conv2))
                and conv2.dilation[0] == 1 and _native.pair_supported(c, k, conv1.dilation[0], prec))

    def _pair_member(self, conv1, conv2, prec=_native.PAIR_F32):
        if not self.pair_fusable(conv1, conv2, prec):
            raise _native.NativeError("resblock pair: shape not built into the fused kernels")
        return dict(w1=_native.pack_pair(effective_weight(conv1), prec, self.guard),
                    w2=_native.pack_pair(effective_weight(conv2), prec, self.guard),
                    b1=self._bias(conv1), b2=self._bias(conv2), k=conv1.kernel_size[0])

    @staticmethod
    def pair_fold_supported(conv1, out_conv, prec, stage=False):
        """Can ``out_conv`` (HiFi-GAN's conv_post: 16 -> 1 channels, 7 taps, 'same' zero padding) be folded into the
        fused pair in front of it (fv_plan_set_pair_output_conv)?  ``stage``: into a one-launch MRF stage, which takes 32
        channels too (HiFi-GAN large / V1: csrc/mrfw_kernels.hpp)."""
        c = conv1.in_channels
        return (prec == _native.PAIR_SPLIT_F16 and (c == 16 or (stage and c == 32))
                and isinstance(out_conv, torch.nn.Conv1d) and out_conv.in_channels == c and out_conv.out_channels == 1
                and out_conv.kernel_size[0] == 7 and out_conv.stride[0] == 1 and out_conv.dilation[0] == 1
                and out_conv.padding[0] == 3 and out_conv.groups == 1)

    def pair(self, conv1, conv2, src, dst, slope, prec=_native.PAIR_F32, add1=SLOT_NONE, add2=SLOT_NONE,
             out_div=1.0, post=POST_NONE, mid=SLOT_NONE, fold=None):
        """dst = src + conv2(lrelu(conv1(lrelu(src)))) as ONE fused op (fv_plan_add_resblock_pair_ex); it reads
        ``src`` raw and applies both activations on chip.  Ops recorded inside one group share a launch.
        With ``add1`` / ``add2`` (split-f16 arithmetic): dst = post(((pair + add1) + add2) / out_div), the MRF
        merge of hifigan.py:99-103 in the reference's association.  ``fold`` = (out_conv, slope, post): the pair's
        result is not stored; dst = post(out_conv(lrelu(result, slope))), a [B, 1, T] tensor (pair_fold_supported)."""
        m = self._pair_member(conv1, conv2, prec)
        if fold is not None:
            out_conv, fslope, fpost = fold
            if not self.pair_fold_supported(conv1, out_conv, prec) or post != POST_NONE or self.group:
                raise _native.NativeError("resblock pair: this output conv cannot be folded into the pair")
            m = dict(m, fold_w=effective_weight(out_conv).detach().float().reshape(16, 7).contiguous(),
                     fold_b=self._bias(out_conv), fold_post=fpost, act_slope=float(fslope))
        wide = conv1.in_channels >= 64      # two conv launches through the scratch slot ``mid`` (csrc/convh_kernels.hpp)
        if wide and mid == SLOT_NONE:
            raise _native.NativeError("resblock pair: a scratch slot (mid) is needed at 64 channels and above")
        self.ops.append(dict(kind="pair", group=self.group, x=src, y=dst, res=SLOT_NONE,
                             acc=add1, acc2=add2, pre_slope=1.0, slope=float(slope),
                             channels=conv1.in_channels, dil=conv1.dilation[0], prec=prec,
                             out_div=float(out_div), post=post, mid=mid if wide else SLOT_NONE,
                             tmps=[mid] if wide else [], **m))

    def mrf_stage_supported(self, blocks):
        """Can these three ResBlock1s run as ONE launch (fv_mrf_stage_split_f16, csrc/mrfh_kernels.hpp)?  16 channels,
        split-f16 arithmetic in force, 3 / 7 / 11 taps, pair dilations (1, 3, 5), every pair fusable."""
        if len(blocks) != 3 or self.pair_precision(blocks[0].channels) != _native.PAIR_SPLIT_F16:
            return False
        if any(len(b.convs1) != 3 or len(b.convs2) != 3 for b in blocks):
            return False
        ks = [b.convs1[0].kernel_size[0] for b in blocks]
        dils = [c.dilation[0] for c in blocks[0].convs1]
        return (_native.mrf_stage_supported(blocks[0].channels, ks, dils)
                and all([c.dilation[0] for c in b.convs1] == dils for b in blocks)
                and all(self.pair_fusable(c1, c2, _native.PAIR_SPLIT_F16) for b in blocks for c1, c2 in zip(b.convs1, b.convs2)))

    def mrf_stage(self, blocks, src, dst, slope, out_div, fold=None):
        """dst = ((r0 + r1) + r2) / out_div, r_j = blocks[j](src): a whole MRF stage (hifigan.py:97-103) as ONE op.
        ``fold`` = (out_conv, slope, post): the stage's result is not stored; dst = post(out_conv(lrelu(result, slope))),
        a [B, 1, T] tensor (pair_fold_supported)."""
        if not self.mrf_stage_supported(blocks):
            raise _native.NativeError("mrf stage: shape not built into the one-launch kernel")
        ks = [b.convs1[0].kernel_size[0] for b in blocks]
        dils = [c.dilation[0] for c in blocks[0].convs1]
        c1s = [c for b in blocks for c in b.convs1]
        c2s = [c for b in blocks for c in b.convs2]
        packed = _native.pack_mrf_stage([effective_weight(c) for c in c1s], [effective_weight(c) for c in c2s],
                                        [self._bias(c) for c in c1s], [self._bias(c) for c in c2s], ks, self.guard)
        op = dict(kind="stage", group=0, x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE, acc2=SLOT_NONE, pre_slope=1.0,
                  slope=float(slope), channels=blocks[0].channels, ks=ks, dils=dils, packed=packed, out_div=float(out_div),
                  post=POST_NONE, prec=_native.PAIR_SPLIT_F16)
        if fold is not None:
            out_conv, fslope, fpost = fold
            if not self.pair_fold_supported(blocks[0].convs1[0], out_conv, _native.PAIR_SPLIT_F16, stage=True):
                raise _native.NativeError("mrf stage: this output conv cannot be folded into the stage")
            op.update(fold_w=effective_weight(out_conv).detach().float().reshape(blocks[0].channels, 7).contiguous(),
                      fold_b=self._bias(out_conv), fold_post=fpost, act_slope=float(fslope))
        self.ops.append(op)

    def conv_split_supported(self, conv, pad, pad_mode=PAD_ZERO):
        """Is this a 'same' conv (zero or reflection padding ``pad`` applied in front of it) the split-f16 conv kernels
        are built for, and is that arithmetic in force?"""
        c, k, d = conv.in_channels, conv.kernel_size[0], conv.dilation[0]
        return (conv.stride[0] == 1 and conv.groups == 1 and conv.out_channels == c and pad == d * (k - 1) // 2
                and pad_mode in (PAD_ZERO, PAD_REFLECT) and c in (64, 128, 256, 512)
                and self.pair_precision(c) == _native.PAIR_SPLIT_F16
                and _native.conv_split_supported(c, k, d))

    def conv_split(self, conv, src, dst, slope, res=SLOT_NONE, add1=SLOT_NONE, add2=SLOT_NONE, out_div=1.0,
                   post=POST_NONE, pad=None, pad_mode=PAD_ZERO):
        """dst = post((conv(pad(lrelu(src, slope))) + bias + res + add1 + add2) / out_div): a 'same' conv of a 64- ...
        512-channel ResBlock (zero padding, the conv's own) or MelGAN ResidualStack (``pad`` reflected samples in
        front of an unpadded conv) with split-f16 operands (fv_plan_add_conv1d_split_f16).  ``src`` is read raw --
        the activation is applied on chip -- so nothing is hoisted into its producer.  Ops recorded inside one group
        share a launch."""
        c, k, d = conv.in_channels, conv.kernel_size[0], conv.dilation[0]
        pad = conv.padding[0] if pad is None else pad + conv.padding[0]
        if not (conv.stride[0] == 1 and conv.groups == 1 and conv.out_channels == c and pad == d * (k - 1) // 2
                and pad_mode in (PAD_ZERO, PAD_REFLECT) and (pad_mode == PAD_ZERO or conv.padding[0] == 0)
                and c in (64, 128, 256, 512) and _native.conv_split_supported(c, k, d)):
            raise _native.NativeError("conv_split: shape not built into the split-f16 conv kernels")
        self.ops.append(dict(kind="convh", group=self.group, x=src, y=dst, res=res, acc=add1, acc2=add2,
                             pre_slope=1.0, slope=float(slope), channels=c, k=k, dil=d, pad_mode=pad_mode,
                             packed=_native.pack_pair(effective_weight(conv), _native.PAIR_SPLIT_F16, self.guard),
                             bias=self._bias(conv), out_div=float(out_div), post=post))

    def residual_stack_supported(self, dilated, pointwise, skip, pad, pad_mode=PAD_ZERO):
        """Can this MelGAN ResidualStack -- ``dilated`` (k taps, 'same' padding ``pad`` applied in front of it, zero or
        reflected), ``pointwise`` and ``skip`` (1x1) -- run as ONE launch (csrc/convk_kernels.hpp), and is split-f16
        arithmetic in force for its channel count?"""
        c, k, d = dilated.in_channels, dilated.kernel_size[0], dilated.dilation[0]
        return (self.precision != "f32" and dilated.stride[0] == 1 and dilated.groups == 1 and dilated.out_channels == c
                and dilated.padding[0] == 0 and pad == d * (k - 1) // 2 and pad_mode in (PAD_ZERO, PAD_REFLECT)
                and all(cv.kernel_size[0] == 1 and cv.stride[0] == 1 and cv.groups == 1 and cv.padding[0] == 0
                        and cv.in_channels == c and cv.out_channels == c for cv in (pointwise, skip))
                and _native.residual_stack_split_supported(c, k, d))

    def residual_stack(self, dilated, pointwise, skip, src, dst, slope, pad_mode=PAD_ZERO, hidden=SLOT_NONE, post=POST_NONE):
        """dst = pointwise(lrelu(dilated(pad(lrelu(src))))) + skip(src), reference modules.py:351-382, as one launch
        (fv_plan_add_residual_stack_split_f16).  ``src`` is read raw; nothing is hoisted into its producer.
        256 channels: the op also carries the two-launch form (scratch slot ``hidden``; fv_plan_set_stack_two_launch,
        ``Tuning::stack_items``) for A/B runs and the bit-identity tests."""
        c, k, d = dilated.in_channels, dilated.kernel_size[0], dilated.dilation[0]
        if not self.residual_stack_supported(dilated, pointwise, skip, d * (k - 1) // 2, pad_mode):
            raise _native.NativeError("residual_stack: shape not built into the one-launch kernel")
        ba, bb = self._bias(pointwise), self._bias(skip)
        bias_out = ba if bb is None else (bb if ba is None else (ba + bb).contiguous())
        w1, w2, ws = effective_weight(dilated), effective_weight(pointwise), effective_weight(skip)
        op = dict(kind="stack", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE, pre_slope=1.0, slope=float(slope),
                  split=True, channels=c, k=k, dil=d, pad_mode=pad_mode,
                  packed=_native.pack_residual_stack_split(w1, w2, ws, self.guard),
                  bias=self._bias(dilated), bias_out=bias_out, post=post)
        if c >= 256:
            if hidden == SLOT_NONE:
                raise _native.NativeError("residual_stack: a scratch slot (hidden) is needed at 256 channels")
            op.update(tmps=[hidden], two_launch=(hidden, _native.pack_pair(w1, _native.PAIR_SPLIT_F16, self.guard),
                                                 _native.pack_conv1x1_2src_split(w2, ws, self.guard)))
        self.ops.append(op)

    def mrf_sum(self, pairs, srcs, dst, slope, out_div, post=POST_NONE):
        """dst = post(sum_j pair_j(srcs[j]) / out_div): the last pairs of the three ResBlocks of an MRF stage
        and the mean, one launch (fv_plan_add_mrf_sum)."""
        ms = [self._pair_member(c1, c2) for c1, c2 in pairs]
        c1 = pairs[0][0]
        if any(p[0].dilation[0] != c1.dilation[0] or p[0].in_channels != c1.in_channels for p in pairs):
            raise _native.NativeError("mrf_sum: the three pairs must share channels and dilation")
        self.ops.append(dict(kind="mrfsum", x=srcs[0], xb=srcs[1], xc=srcs[2], y=dst,
                             res=SLOT_NONE, acc=SLOT_NONE, pre_slope=1.0, slope=float(slope),
                             channels=c1.in_channels, dil=c1.dilation[0], members=ms, out_div=float(out_div),
                             post=post))

    def conv_transpose_takes_merge(self, convt):
        """Can this upsampler form its own input from the three ResBlock results of the stage in front of it
        (``conv_transpose(..., merge=...)``)?  The split-f16 transposed-conv kernel can (csrc/convh_kernels.hpp
        merge_window); the fp32 polyphase kernel and UpsampleLayer cannot."""
        if not isinstance(convt, torch.nn.ConvTranspose1d) or convt.groups != 1 or convt.dilation[0] != 1:
            return False
        k, s = convt.kernel_size[0], convt.stride[0]
        # (up to 128 input channels: with 256 / 512 the merging kernel -- 64-row tiles only -- converts every chunk's window once
        # per 64 rows where the 128-row kernel converts it once per 128: HiFi-GAN large, 64 utterances, 118.7 -> 121.8 ms
        # with the merge in its 256-channel upsampler [measured, tools/bench_configs.py --no-merge])
        return ((self.SPLIT_CONVT_MIN_CIN <= convt.in_channels <= 128
                 or _native.conv_transpose_small(convt.in_channels, convt.out_channels, k, s))
                and self.pair_precision(convt.in_channels) == _native.PAIR_SPLIT_F16
                and _native.conv_transpose_split_supported(convt.in_channels, convt.out_channels, k, s, convt.padding[0],
                                                           convt.output_padding[0]))

    # The split-f16 transposed conv exists from 32 input channels on (fv_conv_transpose1d_split_f16), but with 32 channels and
    # 32 rows the GENERAL kernel pads both K and M to 64: HiFi-GAN light's last upsampler (32 -> 16 x 2) took 29 us on it against
    # 18 us on the fp32-MFMA kernel, and although the 32-channel stage in front of it then ends in one launch (-13 us), the step
    # did not gain at batch 1 and lost 1 % at batch 16 [measured, tools/ab_lib.py].  The plans use the general kernel from 64
    # channels on -- and, for exactly that upsampler, the kernel written for it (csrc/convtn_kernels.hpp).
    SPLIT_CONVT_MIN_CIN = 64

    def conv_transpose(self, convt, src, dst, pre_slope=1.0, post=POST_NONE, trim=0, merge=None):
        """Record a torch.nn.ConvTranspose1d container (polyphase form); ``trim``: samples dropped from the end of
        the output (CausalConvTranspose1d: its stride); ``merge = (slot_b, slot_c, div)``: the input is
        ((src + slot_b) + slot_c) / div (conv_transpose_takes_merge must hold)."""
        if convt.groups != 1 or convt.dilation[0] != 1:
            raise _native.NativeError("only dense, undilated ConvTranspose1d layers exist on this path")
        k, s = convt.kernel_size[0], convt.stride[0]
        p, op = convt.padding[0], convt.output_padding[0]
        cin, cout = convt.in_channels, convt.out_channels
        if merge is not None and not (post == POST_NONE and trim == 0 and self.conv_transpose_takes_merge(convt)):
            raise _native.NativeError("conv_transpose: a merged input exists on the split-f16 kernel only")
        if (post == POST_NONE and (cin >= self.SPLIT_CONVT_MIN_CIN or _native.conv_transpose_small(cin, cout, k, s))
                and self.pair_precision(cin) == _native.PAIR_SPLIT_F16
                and _native.conv_transpose_split_supported(cin, cout, k, s, p, op - int(trim))):
            # kernel = 2 strides, 128+ input channels: split-f16 operands (csrc/convh_kernels.hpp convt_kernel)
            # (src is read raw, the activation is applied on chip: nothing is hoisted into its producer)
            extra = {} if merge is None else dict(xb=merge[0], xc=merge[1], in_div=float(merge[2]))
            self.ops.append(dict(kind="convT", split=True, x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE,
                                 pre_slope=1.0, slope=float(pre_slope),
                                 packed=_native.pack_conv_transpose1d_split(effective_weight(convt), s, self.guard),
                                 bias=self._bias(convt), cin=cin, cout=cout, k=k, stride=s, pad=p,
                                 out_pad=op - int(trim), post=post, **extra))
            return
        self.ops.append(dict(kind="convT", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE,
                             pre_slope=float(pre_slope),
                             packed=_native.pack_conv_transpose1d(effective_weight(convt), s, p),
                             bias=self._bias(convt), cin=cin, cout=cout,
                             k=k, stride=s, pad=p, out_pad=op - int(trim), post=post))

    def upsample_conv(self, layer, src, dst, pre_slope=1.0, post=POST_NONE):
        """Record an UpsampleLayer container (nearest-repeat x rate, then its Conv1d)."""
        conv = layer.conv
        if conv.stride[0] != 1 or conv.dilation[0] != 1 or conv.groups != 1:
            raise _native.NativeError("UpsampleLayer: only a stride-1, undilated, dense conv is supported")
        k, p, u = conv.kernel_size[0], conv.padding[0], layer.upsample_rate
        self.ops.append(dict(kind="upconv", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE,
                             pre_slope=float(pre_slope),
                             packed=_native.pack_upsample_conv1d(effective_weight(conv), u, p),
                             bias=self._bias(conv), cin=conv.in_channels, cout=conv.out_channels,
                             k=k, rate=u, pad=p, post=post))

    def basis_overlap_add(self, basis_weight, src, dst, hop, pre_slope=1.0):
        """frames = act(src)^T @ W^T then overlap-add with hop: a ConvTranspose1d
        with Cout = 1, kernel L, stride hop (weight [C,1,L] = W^T)."""
        L, C = basis_weight.shape
        w = basis_weight.detach().float().t().contiguous().view(C, 1, L)
        self.ops.append(dict(kind="convT", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE,
                             pre_slope=float(pre_slope), packed=_native.pack_conv_transpose1d(w, hop, 0),
                             bias=None, cin=C, cout=1, k=L, stride=hop, pad=0, out_pad=0, post=POST_NONE))

    def subtract_output(self, aux=0, second=True):
        """The op recorded last also subtracts the plan's auxiliary input ``aux`` (0 or 1; a cached
        zero-input response given to ``Plan.run(aux=...)``) after its post op: with ``second`` it keeps its
        own output and writes the difference to SLOT_OUT2, otherwise its output becomes the difference.
        Bias removal without a separate elementwise pass (reference bin/synthesize.py:74-80,
        basis_melgan.py:147-159, bin/test.py:82-91)."""
        op = self.ops[-1]
        if (op["kind"] not in ("conv", "conv2", "conv2h", "stack", "convT", "upconv", "pqmf", "postpqmf") or op.get("group", 0)
                or (op.get("split") and op["kind"] not in ("conv2h", "stack"))):
            raise _native.NativeError("subtract_output: the last op must be a plain conv / transposed conv / two-source "
                                      "1x1 conv / residual stack / pqmf")
        op["sub"] = SLOT_AUX_IN0 + int(aux)
        op["sub_y2"] = SLOT_OUT2 if second else SLOT_NONE

    @staticmethod
    def post_pqmf_supported(conv, synthesis_filter):
        """conv_post + PQMF synthesis as one launch (fv_conv_post_pqmf): 4 sub-bands, 63 taps, a plain 'same' conv."""
        k = conv.kernel_size[0]
        return (isinstance(conv, torch.nn.Conv1d) and conv.out_channels == 4 and conv.stride[0] == 1 and conv.groups == 1
                and conv.dilation[0] == 1 and conv.padding[0] == (k - 1) // 2 and tuple(synthesis_filter.shape[1:]) == (4, 63))

    def conv_post_pqmf(self, conv, synthesis_filter, src, dst, pre_slope=1.0, post=POST_NONE):
        """dst [B,1,4T'] = pqmf_synthesis(post(conv(lrelu(src, pre_slope)))): Multiband-HiFi-GAN's inference tail in one
        launch; the sub-bands stay on chip."""
        if not self.post_pqmf_supported(conv, synthesis_filter):
            raise _native.NativeError("conv_post_pqmf: needs a 4-band 'same' conv and a 63-tap synthesis filter")
        S = synthesis_filter.shape[1]
        h = synthesis_filter.detach().reshape(S, -1).contiguous().float()
        self.ops.append(dict(kind="postpqmf", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE, pre_slope=float(pre_slope),
                             packed=_native.pack_conv1d(effective_weight(conv)), bias=self._bias(conv),
                             cin=conv.in_channels, k=conv.kernel_size[0], pad=conv.padding[0], post=post, h=h))

    def pqmf_synthesis(self, synthesis_filter, src, dst):
        S = synthesis_filter.shape[1]
        h = synthesis_filter.detach().reshape(S, -1).contiguous().float()
        self.ops.append(dict(kind="pqmf", x=src, y=dst, res=SLOT_NONE, acc=SLOT_NONE, pre_slope=1.0, h=h))

    # -- activation hoisting ---------------------------------------------------
    def _hoist_activations(self):
        ops = self.ops
        twin_of = {}
        for i, op in enumerate(ops):
            op.setdefault("y_act", SLOT_NONE)
            op.setdefault("act_slope", 1.0)
        for i, op in enumerate(ops):
            if op["kind"] in ("pqmf", "postpqmf"):
                continue
            y = op["y"]
            # consumers of this definition of slot y: until the slot is written again
            act_uses, raw_needed = {}, y == SLOT_OUT
            for j in range(i + 1, len(ops)):
                c = ops[j]
                for key in ("x", "xb", "xc"):           # activated inputs (sum3 has three)
                    if c.get(key, SLOT_NONE) == y:
                        if c["pre_slope"] != 1.0:
                            act_uses.setdefault(c["pre_slope"], []).append((j, key))
                        else:
                            raw_needed = True
                if y in (c["res"], c["acc"], c.get("acc2", SLOT_NONE), c.get("x2", SLOT_NONE),
                         c.get("resb", SLOT_NONE), c.get("resc", SLOT_NONE)):
                    raw_needed = True
                if c["y"] == y or c.get("y_act") == y or y in c.get("tmps", ()):
                    break
            if not act_uses:
                continue
            slope = max(act_uses, key=lambda s: len(act_uses[s]))
            if len(act_uses) > 1:
                raw_needed = True          # the other slopes keep activating at read time
            if raw_needed:
                if y not in twin_of:
                    twin_of[y] = self.tmp()
                op["y_act"], op["act_slope"], target = twin_of[y], slope, twin_of[y]
            else:
                op["act_slope"], target = slope, y
            for j, key in act_uses[slope]:
                ops[j][key] = target
                ops[j].setdefault("_hoisted", set()).add(key)
        for op in ops:      # an op's read-time activation is off once ALL its activated inputs are hoisted
            keys = [k for k in ("x", "xb", "xc") if op.get(k, SLOT_NONE) != SLOT_NONE]
            if op.get("_hoisted") and set(keys) <= op["_hoisted"]:
                op["pre_slope"] = 1.0

    # -- receptive field (for time-chunked runs) -------------------------------
    def receptive_halo(self):
        """Input frames of context, per side, that every output sample of the recorded graph
        can depend on -- a backward dataflow walk over the ops (slots are reused, so a slot's
        requirement is reset at the op that defines it).  Used by
        :meth:`NativeModule._run_chunked`; slightly generous (transposed convs are bounded by
        ceil(k/stride)+1 taps)."""
        need = {}
        for op in reversed(self.ops):
            h_out = max(need.pop(op["y"], 0), need.pop(op.get("y_act", SLOT_NONE), 0))
            if op["kind"] == "conv":
                reach = op["dil"] * (op["k"] - 1)
                own, rate = max(op["pad"], reach - op["pad"]), 1
            elif op["kind"] == "sum3":
                own, rate = max(op["ks"]) // 2, 1
            elif op["kind"] in ("conv2", "conv2h"):
                own, rate = 0, 1
            elif op["kind"] in ("convh", "stack"):
                own, rate = (op["k"] - 1) // 2 * op["dil"], 1
            elif op["kind"] == "pair":
                own, rate = (op["k"] - 1) // 2 * (op["dil"] + 1) + (3 if "fold_w" in op else 0), 1
            elif op["kind"] == "mrfsum":
                own, rate = max((m["k"] - 1) // 2 * (op["dil"] + 1) for m in op["members"]), 1
            elif op["kind"] == "stage":
                own = max(sum((k - 1) // 2 * (d + 1) for d in op["dils"]) for k in op["ks"]) + (3 if "fold_w" in op else 0)
                rate = 1
            elif op["kind"] == "convT":
                own, rate = -(-op["k"] // op["stride"]) + 1, op["stride"]
            elif op["kind"] == "upconv":
                own, rate = -(-(op["k"] + op["rate"]) // op["rate"]) + 1, op["rate"]
            elif op["kind"] == "postpqmf":                  # conv (k taps) in front of the synthesis filter
                S, ntaps = op["h"].shape
                own, rate = -(-(ntaps // 2) // S) + 1 + op["k"] // 2, S
            else:                                           # pqmf synthesis: S bands, ntaps taps
                S, ntaps = op["h"].shape
                own, rate = -(-(ntaps // 2) // S) + 1, S
            need[op["x"]] = max(need.get(op["x"], 0), -(-h_out // rate) + own)
            for key in ("xb", "xc"):
                if op.get(key, SLOT_NONE) != SLOT_NONE:
                    need[op[key]] = max(need.get(op[key], 0), -(-h_out // rate) + own)
            for aux in (op["res"], op["acc"], op.get("acc2", SLOT_NONE), op.get("x2", SLOT_NONE),
                        op.get("resb", SLOT_NONE), op.get("resc", SLOT_NONE)):
                if aux != SLOT_NONE:
                    need[aux] = max(need.get(aux, 0), h_out)
        return need.get(SLOT_IN, 0)

    def finalize(self):
        """Hoist activations and emit the recorded ops into the native plan."""
        self._hoist_activations()
        self.plan.halo_frames = self.receptive_halo()
        for op in self.ops:
            self.plan.set_group(op.get("group", 0))
            self.plan.set_sum_order(op.get("own_first", False))
            if op["kind"] == "conv":
                self.plan.add_conv1d(op["x"], op["y"], op["packed"], op["bias"], op["cin"], op["cout"],
                                     op["k"], dil=op["dil"], pad=op["pad"], pad_mode=op["pad_mode"],
                                     pre_slope=op["pre_slope"], res=op["res"], acc=op["acc"],
                                     out_div=op["out_div"], post=op["post"], y_act=op["y_act"],
                                     act_slope=op["act_slope"], acc2=op.get("acc2", SLOT_NONE))
            elif op["kind"] == "sum3":
                if op["pre_slope"] != 1.0:
                    raise _native.NativeError("conv_sum3: the activation of its inputs could not be hoisted")
                self.plan.add_conv1d_sum3([op["x"], op["xb"], op["xc"]], [op["res"], op["resb"], op["resc"]],
                                          op["tmps"], op["y"], op["packed"], op["bias"], op["channels"], op["ks"],
                                          out_div=op["out_div"], post=op["post"], y_act=op["y_act"],
                                          act_slope=op["act_slope"])
            elif op["kind"] == "pair":
                self.plan.add_resblock_pair(op["x"], op["y"], op["w1"], op["w2"], op["b1"], op["b2"],
                                            op["channels"], op["k"], op["dil"], op["slope"],
                                            y_act=op["y_act"], act_slope=op["act_slope"], prec=op["prec"],
                                            add1=op["acc"], add2=op["acc2"], out_div=op["out_div"],
                                            post=op["post"], mid=op["mid"])
                if "fold_w" in op:
                    self.plan.set_pair_output_conv(op["fold_w"], op["fold_b"], op["y"], op["act_slope"], op["fold_post"])
            elif op["kind"] == "stage":
                self.plan.add_mrf_stage(op["x"], op["y"], op["packed"], op["channels"], op["ks"], op["dils"], op["slope"],
                                        out_div=op["out_div"], y_act=op["y_act"], act_slope=op["act_slope"])
                if "fold_w" in op:
                    self.plan.set_pair_output_conv(op["fold_w"], op["fold_b"], op["y"], op["act_slope"], op["fold_post"])
            elif op["kind"] == "convh":
                self.plan.add_conv1d_split_f16(op["x"], op["y"], op["packed"], op["bias"], op["channels"], op["k"],
                                               op["dil"], pre_slope=op["slope"], res=op["res"], add1=op["acc"],
                                               add2=op["acc2"], out_div=op["out_div"], post=op["post"],
                                               y_act=op["y_act"], act_slope=op["act_slope"], pad_mode=op["pad_mode"])
            elif op["kind"] == "mrfsum":
                ms = op["members"]
                self.plan.add_mrf_sum([op["x"], op["xb"], op["xc"]], op["y"], [m["w1"] for m in ms],
                                      [m["w2"] for m in ms], [m["b1"] for m in ms], [m["b2"] for m in ms],
                                      op["channels"], [m["k"] for m in ms], op["dil"], op["slope"],
                                      out_div=op["out_div"], post=op["post"], y_act=op["y_act"],
                                      act_slope=op["act_slope"])
            elif op["kind"] == "conv2":
                if op["pre_slope"] != 1.0:
                    raise _native.NativeError("conv_sum_1x1: the activation of the first input could not "
                                              "be hoisted into its producer")
                self.plan.add_conv1d_2src(op["x"], op["x2"], op["y"], op["packed"], op["bias"], op["cin1"],
                                          op["cin2"], op["cout"], res=op["res"], post=op["post"],
                                          y_act=op["y_act"], act_slope=op["act_slope"])
            elif op["kind"] == "conv2h":
                self.plan.add_conv1x1_2src_split_f16(op["x"], op["x2"], op["y"], op["packed"], op["bias"], op["channels"],
                                                     pre_slope=op["slope"], res=op["res"], post=op["post"],
                                                     y_act=op["y_act"], act_slope=op["act_slope"])
            elif op["kind"] == "stack":
                self.plan.add_residual_stack_split_f16(op["x"], op["y"], op["packed"], op["bias"], op["bias_out"],
                                                       op["channels"], op["k"], op["dil"], op["slope"],
                                                       pad_mode=op["pad_mode"], y_act=op["y_act"], act_slope=op["act_slope"],
                                                       two_launch=op.get("two_launch"), post=op["post"])
            elif op["kind"] == "convT" and op.get("split"):
                self.plan.add_conv_transpose1d_split_f16(op["x"], op["y"], op["packed"], op["bias"], op["cin"],
                                                         op["cout"], op["k"], op["stride"], op["pad"], op["out_pad"],
                                                         pre_slope=op["slope"], y_act=op["y_act"],
                                                         act_slope=op["act_slope"],
                                                         merge=(op["xb"], op["xc"], op["in_div"]) if "in_div" in op else None)
            elif op["kind"] == "convT":
                self.plan.add_conv_transpose1d(op["x"], op["y"], op["packed"], op["bias"], op["cin"],
                                               op["cout"], op["k"], op["stride"], op["pad"],
                                               op["out_pad"], pre_slope=op["pre_slope"], post=op["post"],
                                               y_act=op["y_act"], act_slope=op["act_slope"])
            elif op["kind"] == "upconv":
                self.plan.add_upsample_conv1d(op["x"], op["y"], op["packed"], op["bias"], op["cin"],
                                              op["cout"], op["k"], op["rate"], op["pad"],
                                              pre_slope=op["pre_slope"], post=op["post"],
                                              y_act=op["y_act"], act_slope=op["act_slope"])
            elif op["kind"] == "postpqmf":
                self.plan.add_conv_post_pqmf(op["x"], op["y"], op["packed"], op["bias"], op["cin"], op["k"], op["pad"],
                                             op["h"], pre_slope=op["pre_slope"], post=op["post"])
            else:
                self.plan.add_pqmf_synthesis(op["x"], op["y"], op["h"])
            if "sub" in op:
                self.plan.set_output_offset(op["sub"], op["sub_y2"], op["y"])
        # the split-f16 launches of this plan report operands beyond the f16 range through the owner's guard word
        self.plan.guarded = any(op["kind"] == "convh" or op.get("split") or op.get("prec") == _native.PAIR_SPLIT_F16
                                for op in self.ops)
        if self.plan.guarded and self.guard is not None:
            self.plan.set_guard(self.guard)
        return self.plan


# Any parameter / buffer (re-)registration anywhere bumps this epoch: a replaced tensor
# (``conv.weight = nn.Parameter(...)``, torch's remove_weight_norm on a submodule) is not in a
# module's memoised tensor list, so in-place version counters alone would miss it.  Walking
# the module tree on every call instead costs ~350 us for a HiFi-GAN (a quarter of a step).
# The epoch only triggers a RE-SCAN of the module's own tensors: plans are keyed on what the scan
# finds (identity + version of every tensor), so constructing an unrelated module elsewhere in the
# process costs one walk, not a rebuild of every live model's plans.
_registration_epoch = [0]


def _bump_epoch(*_args):
    _registration_epoch[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_epoch)
torch.nn.modules.module.register_module_buffer_registration_hook(_bump_epoch)


class NativeModule(torch.nn.Module):
    """Base of every module on the path: caches native plans keyed by a name and
    rebuilds them when any parameter/buffer was modified, replaced or moved.

    Policy attributes (plain attributes: set them on an instance, or on the class for every model):

    ``precision``    "split" (default): ResBlock / ResidualStack / upsampler layers with 16 ... 512 channels form
                     every fp32 product from split-f16 operand pairs on the f16 matrix cores (fp32-class accuracy,
                     DESIGN.md section 3.7; domain: activations below 65520 in magnitude and not smaller than 2^-10
                     as a whole tensor -- weights of any finite magnitude are rescaled by a power of two per output row
                     when they are packed); "f32": the exact-fp32 MFMA kernels everywhere.
    ``range_guard``  what happens when an activation leaves the split-f16 domain on either side, or a weight is not finite
                     (the reference, fp32 ATen, is defined for any finite fp32).  Weights are checked when a plan is built.  Activations:
                     "sync" -- every call waits for its kernels and reads the guard word they raise; an out-of-range
                         call is repeated on the fp32 kernels before it returns, and the module stays on them (one
                         warning).  Results are always the reference's; calls are synchronous.
                     "auto" (default) -- the same as "sync", for EVERY entry (``forward``, the standalone blocks,
                         ``inference``, ``inference_minus``, the Synthesizer flows): whatever a call returns is the
                         reference's result.  (Until round 5 ``forward`` checked lazily under "auto"; a tensor-in /
                         tensor-out call could then return non-finite values and only the next call noticed.)
                     "lazy" -- an explicit opt-in for pipelined callers: calls stay stream-ordered (asynchronous); the
                         word is looked at when the NEXT call starts (and by ``check_range()``, the explicit barrier):
                         the module then switches to the fp32 kernels with a warning, but the call that overflowed has
                         already returned non-finite values.  ``bench.py`` times its steps under "lazy" (and says so),
                         calls ``check_range()`` after them, and reports the "sync" figure beside.
                     "off"  -- no check.
                     The two sides of the domain are told apart (under "sync" / "auto"): an overflow (or a non-finite value)
                     moves the module to fp32 at once; the LOW side alone -- a block's share of a tensor that is small as a
                     whole, e.g. a quiet stretch of an utterance; the test is per block, not per tensor -- only repeats THAT
                     call on fp32, and moves the module after ``low_range_patience`` such calls in a row.
    ``fuse_pairs``   ResBlock1 pairs as fused launches (default) or conv by conv (round-1 path; A/B runs).
    ``fuse_stage``   a 16- or 32-channel MRF stage (three ResBlock1s of three pairs + the mean) as ONE launch
                     (csrc/mrfh_kernels.hpp, mrfw_kernels.hpp) or as fused-pair launches.  True (default): 16 channels always,
                     32 channels where one window per block covers the batch (hifigan._stage_one_launch); a tuple of channel
                     counts -- e.g. (16,) or (16, 32) -- fuses exactly those stages; False: none (A/B runs, identical bits).
    ``fold_post``    HiFi-GAN's conv_post inside the last pair's / the last stage's launch (default) or as a launch of its own.
    ``merge_in_upsampler``  the MRF merge ((r0 + r1) + r2) / 3 of a fused stage inside the split-f16 upsampler behind it
                     (default: the stage ends in one three-member launch) or in the stage's own last launch (A/B runs;
                     identical bits).
    """

    precision = "split"
    range_guard = "auto"
    low_range_patience = 3        # consecutive low-side alarms answered call by call before the module moves to fp32 for good
    fuse_pairs = True
    fuse_stage = True
    fold_post = True
    merge_in_upsampler = True

    def __init__(self):
        super().__init__()
        self._fv_plans = {}
        self._fv_fast = {}         # (plan getter, batch, frames) -> (state, plan): the per-call route to a plan (_exec)
        self._fv_tensors = None
        self._fv_ids = ()
        self._fv_epoch = -1
        self._fv_key = None
        self._fv_guard = None
        self._fv_overflow = False
        self._fv_low_hits = 0
        self._fv_force_f32 = False

    # -- cache bookkeeping -------------------------------------------------
    def _fv_policy(self):
        """The policy a plan is built under (part of every plan's cache key)."""
        prec = "f32" if (self.precision == "f32" or self._fv_overflow or getattr(self, "_fv_force_f32", False)) else "split"
        # (the guard is part of the key: a plan built under range_guard = "off" carries no guard word)
        return (prec, bool(self.fuse_pairs), bool(self.fold_post), self.range_guard != "off", bool(self.merge_in_upsampler),
                self.fuse_stage if isinstance(self.fuse_stage, tuple) else bool(self.fuse_stage))

    def _fv_state(self):
        """(identity + in-place version of every tensor the plans bake in, policy in force).  The tensor list is
        re-scanned only when some module somewhere registered a parameter or buffer since the last scan."""
        if self._fv_tensors is None or self._fv_epoch != _registration_epoch[0]:
            self._fv_tensors = list(self.parameters()) + list(self.buffers())
            self._fv_ids = tuple(id(t) for t in self._fv_tensors)    # (a replaced tensor is a registration: re-scanned)
            self._fv_epoch = _registration_epoch[0]
        return (self._fv_ids, tuple([t._version for t in self._fv_tensors])), self._fv_policy()

    # A native plan is a raw handle plus pointers into THIS module's packed weights: a copied or
    # unpickled module must not share it (double free, stale device pointers) -- it rebuilds its own.
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_fv_plans"] = {}
        state["_fv_fast"] = {}
        state["_fv_tensors"] = None
        state["_fv_epoch"] = -1
        state["_fv_guard"] = None
        state.pop("_fv_zero", None)       # cached device waveforms (zero-mel responses) are not part of a model
        state.pop("_zero_cache", None)
        return state

    def invalidate_plans(self):
        """Drop cached native plans (packed weights) and everything derived from the weights.  Called automatically on
        load_state_dict / .to() / weight-norm changes / in-place parameter updates; call it by hand after writing
        through ``param.data``."""
        self._fv_plans = {}
        self._fv_fast = {}
        self._fv_tensors = None
        self._fv_overflow = False         # new weights: the split-f16 path gets its chance again
        self._fv_low_hits = 0
        self.__dict__.pop("_fv_zero", None)
        self.__dict__.pop("_zero_cache", None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.invalidate_plans()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.invalidate_plans()
        return out

    def train(self, mode=True):
        changed = any(m.training != mode for m in self.modules())
        out = super().train(mode)
        if changed:                      # eval-mode-only layers (BatchNorm) are baked into plans
            for m in self.modules():
                if isinstance(m, NativeModule):
                    m.invalidate_plans()
        return out

    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _native.NativeError(
                f"{type(self).__name__} lives on {dev}: fastvocoder_amd runs only on a ROCm "
                "device (model.to('cuda')); there is no CPU fallback")
        return dev

    def _guard_word(self):
        if self._fv_guard is None:
            self._fv_guard = _native.GuardWord()
        return self._fv_guard

    def _went_out_of_range(self, what):
        """Sticky: from now on (until the weights change) this module's plans use the exact-fp32 kernels."""
        self._fv_overflow = True
        warnings.warn(f"{type(self).__name__}: {what} outside the split-f16 range (|v| >= 65520 or not finite, or a "
                      "tensor that is smaller than 2^-10 as a whole); this model now runs on the exact-fp32 kernels "
                      "(precision = 'f32')", RuntimeWarning, stacklevel=3)

    def _plan(self, name, emit, in_channels):
        """Return the cached plan ``name`` or build it with ``emit(builder)``.  A split-f16 plan whose pack kernels
        meet a weight beyond the f16 range is discarded and rebuilt with fp32 arithmetic.  ``name`` may be a callable
        (and ``emit`` should then consult the same things): graphs whose shape depends on the policy -- which stages
        run fused -- are named and emitted under the policy in force at that moment."""
        state = self._fv_state()
        name_of, name = name, (name() if callable(name) else name)
        hit = self._fv_plans.get((name, state[1]))
        if hit is not None and hit[0] == state:
            return hit[1]
        prec, _, fold, guarded = state[1][:4]
        guard = self._guard_word() if (prec == "split" and guarded) else None
        with torch.no_grad():
            pb = PlanBuilder(in_channels, precision=prec, fold_post=fold, guard=guard)
            emit(pb)
            plan = pb.finalize()
            if guard is not None and plan.guarded:
                # one-off, at plan build: the pack kernels' verdict (they ran on the MODULE's device, under _on())
                torch.cuda.current_stream(self._device()).synchronize()
                if guard.peek(1):
                    guard.clear(1)
                    self._went_out_of_range("a weight lies")
                    return self._plan(name_of, emit, in_channels)
        self._fv_plans[(name, state[1])] = (state, plan)
        return plan

    def _exec(self, plan_for, x, sync=False, **run_kw):
        """``plan_for(T).run(x, **run_kw)`` under the module's range guard (class docstring); ``sync``: the result is
        bound for the host (informational since round 5: "auto" checks every call before returning).  ``plan_for`` must resolve the plan through
        :meth:`_plan` every time it is called: after an overflow it returns the fp32 plan."""
        mode = self.range_guard
        if mode == "auto":              # safe by default: every call is checked before it returns (class docstring)
            mode = "sync"
        elif mode not in ("sync", "lazy", "off"):
            raise ValueError(f"range_guard = {mode!r}: 'auto', 'sync', 'lazy' or 'off'")
        if mode == "lazy" and not self._fv_overflow and self._fv_guard is not None and self._fv_guard.peek(0):
            self._fv_guard.clear(0)
            self._went_out_of_range("an EARLIER call met an activation (its output holds non-finite values)")
        B, T = int(x.shape[0]), int(x.shape[2])
        self._fv_batch = B                # (graphs whose shape depends on the batch: hifigan._stage_one_launch)
        # The per-call route to the plan: resolving it through plan_for costs the whole policy evaluation (which stages
        # fuse at this length and batch, ~100 us of Python in front of the first launch -- exposed in full when every call
        # is checked before it returns); what the answer depends on is (getter, batch, frames) + the state _plan keys on.
        getter = getattr(plan_for, "__func__", None) if getattr(plan_for, "__self__", None) is self else None
        plan = None
        if getter is not None:
            state = self._fv_state()
            hit = self.__dict__.setdefault("_fv_fast", {}).get((getter, B, T))
            if hit is not None and hit[0] == state:
                plan = hit[1]
        if plan is None:
            plan = plan_for(T)
            if getter is not None:
                if len(self._fv_fast) >= 512:
                    self._fv_fast.clear()
                self._fv_fast[(getter, B, T)] = (self._fv_state(), plan)    # (the state AFTER the build: a weight beyond
                                                                           # the f16 range moves the policy to fp32)
        out = plan.run(x, **run_kw)
        if mode == "sync" and plan.guarded:
            seen = plan.check_range()
            if seen == "low" and getattr(self, "_fv_low_hits", 0) < self.low_range_patience:
                # The LOW side alone: some block's share of a tensor was small as a whole -- a property of THIS input (a quiet
                # stretch), not of the model.  This call is repeated on the exact-fp32 kernels; the module stays on the split
                # kernels (until it has happened `low_range_patience` times in a row: then it is the model).
                self._fv_low_hits = getattr(self, "_fv_low_hits", 0) + 1
                self._fv_force_f32 = True
                try:
                    out = plan_for(x.shape[2]).run(x, **run_kw)
                finally:
                    self._fv_force_f32 = False
            elif seen:
                self._went_out_of_range("an activation lies")
                out = plan_for(x.shape[2]).run(x, **run_kw)
            else:
                self._fv_low_hits = 0
        return out

    def check_range(self):
        """Wait for this module's queued work and report whether a split-f16 kernel has met an out-of-range operand
        since the last check (True: the module has switched to the fp32 kernels; earlier outputs of a "lazy" module
        may hold non-finite values)."""
        if self._fv_guard is None:
            return False
        torch.cuda.current_stream(self._device()).synchronize()
        if not self._fv_guard.peek(0):
            return False
        self._fv_guard.clear(0)
        if not self._fv_overflow:
            self._went_out_of_range("an activation lay")
        return True

    # Longest input (frames) handed to one plan run; longer inputs are cut into chunks with
    # receptive-field halos.  One launch addresses < 1 GiB per
    # tensor row (csrc/conv_mfma.hip kOutOfRange), which the widest shipped layer
    # (MelGAN 256 ch x 10T) reaches near T = 100k frames; 16384 frames (~3 min of audio)
    # keeps the workspace bounded and costs < 1 % in halo recomputation.
    max_frames_per_run = 16384

    def _run_plan(self, plan, x, chunk_frames=None, sync=False):
        """plan.run(x), time-chunked when x is longer than ``chunk_frames``
        (default ``max_frames_per_run``).  ``plan`` is a native plan or a callable ``T -> plan``
        (variants of one graph whose fused ops depend on the length, all with the same
        receptive field and output-length law)."""
        plan_for = plan if callable(plan) else (lambda T: plan)
        chunk = self.max_frames_per_run if chunk_frames is None else int(chunk_frames)
        if chunk <= 0 or x.shape[2] <= chunk:
            return self._exec(plan_for, x, sync=sync)
        return self._run_chunked(plan_for, x, chunk, sync)

    def _run_chunked(self, plan_for, x, chunk, sync=False):
        """Exact chunked evaluation: each chunk of ``chunk`` frames is run with ``halo`` extra
        frames of real input on both sides (clipped at the utterance ends, where the layers'
        own zero / reflection padding applies as in a whole run) and only its interior is
        kept.  Output length law out = hop*T + c is read from the plan: c < 0 is a symmetric
        crop (MB-large), c > 0 a tail (Basis overlap-add); either way a chunk that starts at
        frame ``lo`` produces final samples [lo*hop, lo*hop + len)."""
        B, _, T = x.shape
        self._fv_batch = int(B)           # (plan_for below: graphs whose shape depends on the batch are chosen for THIS batch,
                                          # not for the last call's -- ADVICE r5)
        plan = plan_for(min(T, chunk))            # any length's variant has the same receptive field and output-length law: ask
                                                  # for one of the size the chunks will run at, not for a whole-length plan
        halo = plan.halo_frames
        (_, n1), (cout, n2) = plan.output_shape(halo + 64), plan.output_shape(halo + 65)
        hop = n2 - n1
        total = plan.output_shape(T)[1]
        out = torch.empty((B, cout, total), dtype=torch.float32, device=x.device)
        for a in range(0, T, chunk):
            b = min(T, a + chunk)
            lo, hi = max(0, a - halo), min(T, b + halo)
            y = self._exec(plan_for, x[:, :, lo:hi].contiguous(), sync=sync)
            first = a * hop if a > 0 else 0
            last = b * hop if b < T else total
            out[:, :, first:last] = y[:, :, first - lo * hop: last - lo * hop]
        return out

    def _run_minus(self, plan_for, x, bias):
        """Run ``plan_for(T)`` -- a graph whose last op carries ``PlanBuilder.subtract_output(0, True)`` --
        on x [B,C,T] with ``bias`` (any shape with the output's element count per utterance, or one row
        per utterance) as the offset: returns (out, out - bias), both [B,Cout,Tout], from ONE pass; the
        subtraction happens in the last kernel's epilogue.  Inputs longer than ``max_frames_per_run`` take
        the chunked route and one elementwise subtraction."""
        bias = bias.detach().to(device=x.device, dtype=torch.float32)
        if x.shape[2] > self.max_frames_per_run:
            out = self._run_plan(plan_for, x, sync=True)
            return out, out - bias.reshape((-1,) + tuple(out.shape[1:]))
        c, n = plan_for(x.shape[2]).output_shape(x.shape[2])
        if bias.numel() not in (c * n, x.shape[0] * c * n):
            raise _native.NativeError(f"bias has {bias.numel()} elements, the output {c} x {n} per utterance")
        return self._exec(plan_for, x, sync=True, aux=(bias.reshape(-1, c, n).contiguous(),), out2=True)

    def _prepare(self, x):
        """Any array-like -> contiguous fp32 tensor on this module's device."""
        dev = self._device()
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x, dtype=torch.float)
        # (a host array travels by torch's pageable copy: measured 32 us for a 320 KB mel on the MI355X box, against 56 us
        # through a pinned staging buffer of the module's -- the extra host copy costs more than the DMA set-up saves)
        return x.detach().to(device=dev, dtype=torch.float32).contiguous()

    # -- weight-norm lifecycle (reference hifigan.py:58-90 and siblings) ----
    def remove_weight_norm(self):
        """Remove weight normalization module from all of the layers."""
        def _remove(m):
            try:
                torch.nn.utils.remove_weight_norm(m)
            except ValueError:  # this module didn't have weight norm
                return
        self.apply(_remove)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.invalidate_plans()

    def apply_weight_norm(self):
        """Apply weight normalization module from all of the layers."""
        def _apply(m):
            if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d)) and not hasattr(m, "weight_g"):
                weight_norm(m)
        self.apply(_apply)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.invalidate_plans()

    _RESET_STD = 0.01

    def reset_parameters(self):
        """``m.weight.data.normal_(0, std)`` on every conv, like the reference.
        With weight norm attached ``weight`` is the derived tensor, so -- exactly
        as in the reference -- this has no effect on
        the next forward; after remove_weight_norm it re-initialises for real."""
        def _reset(m):
            if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d)):
                m.weight.data.normal_(0.0, self._RESET_STD)
        self.apply(_reset)
        for m in self.modules():
            if isinstance(m, NativeModule):
                m.invalidate_plans()
