"""Building blocks of the four generators, as parameter containers that emit
native ops (see engine.py).  Names, constructor arguments and ``state_dict``
keys follow /root/reference/model/generator/modules.py so checkpoints and
calling code carry over; the arithmetic is in csrc/.

Each block can also be called on its own (``block(x)``, x ``[B,C,T]`` on a ROCm
device): it then runs a private plan SLOT_IN -> SLOT_OUT.
"""
import torch

from .engine import (NativeModule, PAD_CAUSAL, PAD_REFLECT, PAD_ZERO, POST_NONE, SLOT_IN, SLOT_NONE,
                     SLOT_OUT)
from .. import _native

LRELU_SLOPE = 0.1  # reference modules.py:9


def get_padding(kernel_size, dilation=1):
    """'same' padding of a dilated odd kernel (reference modules.py:186-187)."""
    return int((kernel_size * dilation - dilation) / 2)


def _resblock_conv(channels, kernel_size, dilation, bias):
    """Conv1d container initialised like the reference's Conv1d subclass
    (modules.py:92-103: kaiming-normal weight, zero bias)."""
    conv = torch.nn.Conv1d(channels, channels, kernel_size, 1, dilation=dilation,
                           padding=get_padding(kernel_size, dilation), bias=bias)
    torch.nn.init.kaiming_normal_(conv.weight, nonlinearity="relu")
    if conv.bias is not None:
        torch.nn.init.constant_(conv.bias, 0.0)
    return conv


class _Block(NativeModule):
    """A [B,C,T] -> [B,C,T] block: standalone ``forward`` via a private plan."""

    channels = None

    def emit(self, pb, src, dst, scratch, **epilogue):
        raise NotImplementedError

    def scratch_slots(self):
        return 0

    def forward(self, x):
        x = self._prepare(x)

        def build(pb):
            self.emit(pb, SLOT_IN, SLOT_OUT, [pb.tmp() for _ in range(self.scratch_slots())])
        return self._exec(lambda T: self._plan("forward", build, self.channels), x)


class ResBlock1(_Block):
    """HiFi-GAN residual block, three (dilated conv, conv) pairs
    (reference modules.py:190-230):  x <- x + c2(lrelu(c1(lrelu(x))))."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5), bias=True):
        super().__init__()
        self.channels = channels
        self.convs1 = torch.nn.ModuleList(
            [_resblock_conv(channels, kernel_size, d, bias) for d in dilation])
        self.convs2 = torch.nn.ModuleList(
            [_resblock_conv(channels, kernel_size, 1, bias) for _ in dilation])

    def scratch_slots(self):
        return 3

    def num_steps(self):
        return 2 * len(self.convs1)

    def emit_step(self, pb, step, state, src, dst, scratch, acc=SLOT_NONE, acc2=SLOT_NONE, out_div=1.0,
                  own_first=False):
        """Emit conv number ``step`` (0 .. num_steps()-1) of the block; ``state`` is a dict
        the caller keeps per block between steps.  Lets a generator interleave the steps
        of several independent blocks (one group per step)."""
        mid, ping, pong = scratch
        cur = state.get("cur", src)
        i, second = divmod(step, 2)
        if not second:
            pb.conv(self.convs1[i], cur, mid, pre_slope=LRELU_SLOPE)
            return
        last = i == len(self.convs1) - 1
        nxt = dst if last else (ping if cur != ping else pong)
        pb.conv(self.convs2[i], mid, nxt, pre_slope=LRELU_SLOPE, res=cur,
                acc=acc if last else SLOT_NONE, acc2=acc2 if last else SLOT_NONE,
                out_div=out_div if last else 1.0, own_first=own_first and last)
        state["cur"] = nxt

    def last_conv_inputs(self, state, src, scratch):
        """(conv, input slot, residual slot) of the block's last conv, for a caller that merges
        the last convs of several blocks itself (PlanBuilder.conv_sum3); valid after every
        earlier step has been emitted."""
        return self.convs2[-1], scratch[0], state.get("cur", src)

    def emit(self, pb, src, dst, scratch, acc=SLOT_NONE, out_div=1.0):
        """src -> dst through the pairs; the LAST conv's epilogue also carries the
        caller's running MRF sum (``acc``) and mean (``out_div``)."""
        state = {}
        for step in range(self.num_steps()):
            self.emit_step(pb, step, state, src, dst, scratch, acc=acc, out_div=out_div)

    def pairs_fusable(self, precision="split"):
        """Every (dilated conv, conv) pair has a shape the fused pair kernels are built for."""
        from .engine import PlanBuilder, pair_precision
        prec = pair_precision(precision, self.channels)
        return all(PlanBuilder.pair_fusable(c1, c2, prec) for c1, c2 in zip(self.convs1, self.convs2))

    def emit_fused(self, pb, src, dst, scratch):
        """src -> dst with every pair as ONE fused launch (csrc/pair_kernels.hpp); needs T % 4 == 0."""
        mid, ping, pong = scratch
        cur = src
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            nxt = dst if i == len(self.convs1) - 1 else (ping if cur != ping else pong)
            pb.pair(c1, c2, cur, nxt, LRELU_SLOPE, pb.pair_precision(self.channels), mid=mid)
            cur = nxt

    def forward(self, x):
        x = self._prepare(x)

        def fused():     # (evaluated under the policy in force: after a range overflow the fp32 pair kernels exist at 16 / 32 channels only)
            return x.shape[2] % 4 == 0 and self.fuse_pairs and self.pairs_fusable(self._fv_policy()[0])

        def build(pb):
            (self.emit_fused if fused() else self.emit)(pb, SLOT_IN, SLOT_OUT, [pb.tmp() for _ in range(3)])
        return self._exec(lambda T: self._plan(lambda: "forward_fused" if fused() else "forward", build, self.channels), x)


class ResBlock2(_Block):
    """Two single dilated convs with residual (reference modules.py:233-252)."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3), bias=True):
        super().__init__()
        self.channels = channels
        self.convs = torch.nn.ModuleList(
            [_resblock_conv(channels, kernel_size, d, bias) for d in dilation])

    def scratch_slots(self):
        return 2

    def num_steps(self):
        return len(self.convs)

    def emit_step(self, pb, step, state, src, dst, scratch, acc=SLOT_NONE, acc2=SLOT_NONE, out_div=1.0,
                  own_first=False):
        ping, pong = scratch[:2]
        cur = state.get("cur", src)
        last = step == len(self.convs) - 1
        nxt = dst if last else (ping if cur != ping else pong)
        pb.conv(self.convs[step], cur, nxt, pre_slope=LRELU_SLOPE, res=cur,
                acc=acc if last else SLOT_NONE, acc2=acc2 if last else SLOT_NONE,
                out_div=out_div if last else 1.0, own_first=own_first and last)
        state["cur"] = nxt

    def emit(self, pb, src, dst, scratch, acc=SLOT_NONE, out_div=1.0):
        state = {}
        for step in range(self.num_steps()):
            self.emit_step(pb, step, state, src, dst, scratch, acc=acc, out_div=out_div)


def _activation_slope(name, params):
    """Map the reference's (nonlinear_activation, params) pair onto the fused
    input activation of the conv kernels."""
    if name == "LeakyReLU":
        return float(params.get("negative_slope", 0.01))
    if name == "ReLU":
        return 0.0
    raise _native.NativeError(f"activation {name} is not fused into the HIP conv kernels "
                              "(LeakyReLU / ReLU are)")


def _pad_mode(name, params):
    if name == "ReflectionPad1d":
        return PAD_REFLECT
    if name == "ConstantPad1d" and float(params.get("value", 0.0)) == 0.0:
        return PAD_ZERO
    raise _native.NativeError(f"padding module {name}{params} is not supported by the HIP conv "
                              "kernels (ReflectionPad1d / zero ConstantPad1d are)")


class CausalConv1d(torch.nn.Module):
    """Parameter container with the reference's layout (modules.py:273-294): ``pad`` module +
    ``conv``; checkpoint keys ``<prefix>.conv.*``.  Semantics: pad (k-1)*dil on BOTH sides
    with the configured pad module, valid conv, keep the first T outputs -- one
    FV_PAD_CAUSAL conv launch here."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, bias=True,
                 pad="ConstantPad1d", pad_params={"value": 0.0}):
        super().__init__()
        self.pad_amount = (kernel_size - 1) * dilation
        self.pad_mode = _pad_mode(pad, pad_params)
        self.pad = getattr(torch.nn, pad)(self.pad_amount, **pad_params)
        self.conv = torch.nn.Conv1d(in_channels, out_channels, kernel_size, dilation=dilation,
                                    bias=bias)


class ResidualStack(_Block):
    """MelGAN residual stack (reference modules.py:320-382):
    ``conv1x1(act(conv_k_dilated(pad(act(c))))) + skip1x1(c)``.
    ``stack`` keeps the reference's Sequential indices: conv at .2 and .4, or, with
    ``use_causal_conv``, a CausalConv1d at .1 (keys ``stack.1.conv.*``) and the 1x1 at .3."""

    fuse_skip = True      # stack[4] + skip_layer as ONE GEMM over the concatenated K range (False: three launches; A/B)
    fuse_stack = True     # the whole stack as ONE launch where that kernel exists (32 ... 256 channels; False: A/B)
    fuse_last = True      # ... also the graph's last stack (False: A/B)

    def __init__(self, kernel_size=3, channels=32, dilation=1, bias=True,
                 nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2},
                 pad="ReflectionPad1d", pad_params={}, use_causal_conv=False):
        super().__init__()
        self.channels = channels
        self._slope = _activation_slope(nonlinear_activation, nonlinear_activation_params)
        self._pad_mode = _pad_mode(pad, pad_params)
        act = getattr(torch.nn, nonlinear_activation)
        if not use_causal_conv:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
            self._pad = (kernel_size - 1) // 2 * dilation
            self.stack = torch.nn.Sequential(
                act(**nonlinear_activation_params),
                getattr(torch.nn, pad)(self._pad, **pad_params),
                torch.nn.Conv1d(channels, channels, kernel_size, dilation=dilation, bias=bias),
                act(**nonlinear_activation_params),
                torch.nn.Conv1d(channels, channels, 1, bias=bias),
            )
            self._conv_at = (2, 4)
        else:
            self._pad = (kernel_size - 1) * dilation
            self._pad_mode |= PAD_CAUSAL
            self.stack = torch.nn.Sequential(
                act(**nonlinear_activation_params),
                CausalConv1d(channels, channels, kernel_size, dilation=dilation, bias=bias, pad=pad,
                             pad_params=pad_params),
                act(**nonlinear_activation_params),
                torch.nn.Conv1d(channels, channels, 1, bias=bias),
            )
            self._conv_at = (1, 3)
        self.skip_layer = torch.nn.Conv1d(channels, channels, 1, bias=bias)

    def scratch_slots(self):
        return 2

    def emit(self, pb, src, dst, scratch, post=POST_NONE, last=False):
        """``last``: this stack produces the graph's output (final activation and, in the bias-removal flows, an output
        offset in its epilogue; ``fuse_last = False``: the two-launch form, A/B)."""
        hidden, skip = scratch[:2]
        dilated, pointwise = (self.stack[i] for i in self._conv_at)
        dilated = getattr(dilated, "conv", dilated)               # CausalConv1d wraps its conv
        if (self.fuse_stack and (not last or self.fuse_last)
                and pb.residual_stack_supported(dilated, pointwise, self.skip_layer, self._pad, self._pad_mode)):
            # 32 ... 256 channels: dilated conv, activation, 1x1 conv and skip branch in ONE launch, the hidden
            # tile stays in LDS (csrc/convk_kernels.hpp; 256 channels: only for runs of few tiles, else the two launches below)
            pb.residual_stack(dilated, pointwise, self.skip_layer, src, dst, self._slope, pad_mode=self._pad_mode, hidden=hidden,
                              post=post)
            return
        if pb.conv_split_supported(dilated, self._pad, self._pad_mode):
            # 64 ... 512 channels: the dilated conv with split-f16 operands (csrc/convh_kernels.hpp), the
            # reflected samples are mirrored addresses of its window loader
            pb.conv_split(dilated, src, hidden, self._slope, pad=self._pad, pad_mode=self._pad_mode)
        else:
            pb.conv(dilated, src, hidden, pad=self._pad, pad_mode=self._pad_mode, pre_slope=self._slope)
        if self.fuse_skip and self.channels > 4:
            # stack[4](act(hidden)) + skip_layer(src): one GEMM over the concatenated K range;
            # the skip branch costs no launch and no [B,C,T] round trip (src is read raw)
            pb.conv_sum_1x1(pointwise, hidden, self.skip_layer, src, dst, pre_slope_a=self._slope, post=post)
        else:
            pb.conv(self.skip_layer, src, skip)                   # un-activated input
            pb.conv(pointwise, hidden, dst, pre_slope=self._slope, res=skip, post=post)


class LastLinear(NativeModule):
    """Basis-MelGAN's optional head (reference modules.py:116-132, ``lastlinear: True``):
    LeakyReLU(0.2) -> BatchNorm1d -> conv1x1 -> LeakyReLU(0.2) -> BatchNorm1d -> conv1x1.
    Inference only: in eval mode each BatchNorm is a per-channel affine map and is folded
    into the 1x1 conv behind it when the plan is built (fv_fold_batchnorm_conv), leaving two
    conv launches.  Train-mode BatchNorm (batch statistics) belongs to training, which this
    library does not do: it raises."""

    def __init__(self, hidden_channel, out_channel, bias=True):
        super().__init__()
        self.hidden_channel = hidden_channel
        self.activation = torch.nn.LeakyReLU(negative_slope=0.2)
        self.bn_1 = torch.nn.BatchNorm1d(hidden_channel)
        self.linear_1 = torch.nn.Conv1d(hidden_channel, hidden_channel, 1, bias=bias)
        self.bn_2 = torch.nn.BatchNorm1d(hidden_channel)
        self.linear_2 = torch.nn.Conv1d(hidden_channel, out_channel, 1, bias=bias)
        for conv in (self.linear_1, self.linear_2):                # reference Conv1d1x1 init
            torch.nn.init.kaiming_normal_(conv.weight, nonlinearity="relu")
            if conv.bias is not None:
                torch.nn.init.constant_(conv.bias, 0.0)

    def emit(self, pb, src, dst, scratch, post=POST_NONE):
        if self.bn_1.training or self.bn_2.training:
            raise _native.NativeError(
                "LastLinear: BatchNorm1d in train mode uses batch statistics, which only training "
                "needs; call .eval() (the synthesize/test flows do)")
        pb.conv(self.linear_1, src, scratch[0], pre_slope=0.2, batchnorm=self.bn_1)
        pb.conv(self.linear_2, scratch[0], dst, pre_slope=0.2, batchnorm=self.bn_2, post=post)

    def forward(self, x):
        x = self._prepare(x)
        return self._exec(lambda T: self._plan("forward", lambda pb: self.emit(pb, SLOT_IN, SLOT_OUT, [pb.tmp()]),
                                               self.hidden_channel), x)


class LastLayer(NativeModule):
    """act -> pad -> Conv1d(kernel_size) (reference modules.py:76-89)."""

    def __init__(self, in_channels, out_channels, nonlinear_activation,
                 nonlinear_activation_params, pad, kernel_size, pad_params, bias):
        super().__init__()
        self.in_channels = in_channels
        self._slope = _activation_slope(nonlinear_activation, nonlinear_activation_params)
        self._pad_mode = _pad_mode(pad, pad_params)
        self._pad = (kernel_size - 1) // 2
        self.activation = getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)
        self.pad = getattr(torch.nn, pad)(self._pad, **pad_params)
        self.conv = torch.nn.Conv1d(in_channels, out_channels, kernel_size, bias=bias)

    def emit(self, pb, src, dst, post=POST_NONE):
        pb.conv(self.conv, src, dst, pad=self._pad, pad_mode=self._pad_mode,
                pre_slope=self._slope, post=post)

    def forward(self, x):
        x = self._prepare(x)
        return self._exec(lambda T: self._plan("forward", lambda pb: self.emit(pb, SLOT_IN, SLOT_OUT),
                                               self.in_channels), x)


class BasisSignalLayer(NativeModule):
    """Learned-basis synthesis (reference modules.py:255-267): ``weight [B,F,C]``
    times ``W^T [C,L]`` gives frames, overlap-added with hop L/2.  Here both are
    ONE polyphase transposed-conv launch; no [B,F,L] frame tensor exists."""

    def __init__(self, basis_signal_weight, L=64):
        super().__init__()
        self.layer = torch.nn.Linear(basis_signal_weight.size(0), basis_signal_weight.size(1),
                                     bias=False)
        self.layer.weight = torch.nn.Parameter(basis_signal_weight)
        self.L = L

    def emit(self, pb, src, dst, pre_slope=1.0):
        """src holds the trunk output in its native [B,C,F] layout."""
        pb.basis_overlap_add(self.layer.weight, src, dst, self.L // 2, pre_slope=pre_slope)

    def forward(self, weight):
        """weight [B,F,C] (the reference's layout) -> [B,(F-1)*L/2+L]."""
        w = self._prepare(weight).transpose(1, 2).contiguous()
        out = self._exec(lambda T: self._plan("forward", lambda pb: self.emit(pb, SLOT_IN, SLOT_OUT),
                                              self.layer.weight.shape[1]), w)
        return out[:, 0, :]


class CausalConvTranspose1d(NativeModule):
    """Drop-in for the reference ``CausalConvTranspose1d`` (modules.py:297-317): ``ConvTranspose1d(kernel_size, stride)``
    without its last ``stride`` output samples.  None of the reference's generators instantiates it; it exists
    here for completeness of the module surface (same ``deconv.weight`` / ``deconv.bias`` keys).  The tail is
    never computed: the transposed conv runs with ``out_pad = -stride`` (include/fastvocoder_hip.h)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True):
        super().__init__()
        self.deconv = torch.nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, bias=bias)
        self.stride = stride
        self.in_channels = in_channels

    def forward(self, x):
        x = self._prepare(x)
        return self._exec(lambda T: self._plan("forward", lambda pb: pb.conv_transpose(self.deconv, SLOT_IN, SLOT_OUT,
                                                                                        trim=self.stride),
                                               self.in_channels), x)


class UpsampleLayer(NativeModule):
    """Nearest-repeat + Conv1d upsampler (reference modules.py:135-177: ``Stretch2d`` then
    ``conv``), chosen by ``transposedconv: False``.  The repeated signal is never built:
    taps that read the same input sample are summed when the weight is packed and the
    layer runs as a short dense conv over ``out_channel * upsample_rate`` phase rows
    (csrc/api.hip pack_upconv_kernel)."""

    def __init__(self, in_channel, out_channel, upsample_rate, kernel_size, stride, padding,
                 dilation=1, bias=True):
        super().__init__()
        self.upsample_rate = upsample_rate
        self.in_channel = in_channel
        self.conv = torch.nn.Conv1d(in_channel, out_channel, kernel_size, stride, padding,
                                    dilation=dilation, bias=bias)

    def forward(self, x):
        x = self._prepare(x)
        return self._exec(lambda T: self._plan("forward", lambda pb: pb.upsample_conv(self, SLOT_IN, SLOT_OUT),
                                               self.in_channel), x)
