"""MelGAN and Basis-MelGAN generators on libfastvocoder_hip.so.

Same constructor kwargs (conf/melgan/original.yaml, conf/basis-melgan/light.yaml
keys), methods and ``state_dict`` keys as the reference's ``MelGANGenerator``
(/root/reference/model/generator/melgan.py:17-185) and ``BasisMelGANGenerator``
(basis_melgan.py:19-212).  The ``melgan`` attribute is a ``Sequential`` with the
reference's exact index layout (pad, conv, then per upsample: act, ConvTranspose1d,
``stacks`` x ResidualStack, ...) because checkpoint keys are ``melgan.<index>.*``;
it is a parameter container only -- calls go through native plans.
"""
import torch

from .. import _native
from .engine import NativeModule, POST_NONE, POST_RELU, POST_TANH, SLOT_IN, SLOT_OUT, SLOT_OUT2  # noqa: F401
from .modules import (BasisSignalLayer, LastLayer, LastLinear, ResidualStack, UpsampleLayer,
                      _activation_slope, _pad_mode)


class _MelGANTrunk(NativeModule):
    _RESET_STD = 0.02  # reference melgan.py:166

    def _build_layers(self, in_channels, kernel_size, channels, bias, upsample_scales,
                      stack_kernel_size, stacks, nonlinear_activation,
                      nonlinear_activation_params, pad, pad_params, use_causal_conv,
                      transposedconv=True):
        if not use_causal_conv:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        self._slope = _activation_slope(nonlinear_activation, nonlinear_activation_params)
        self._first_pad = ((kernel_size - 1) // 2, _pad_mode(pad, pad_params))
        self._in_channels = in_channels
        act = getattr(torch.nn, nonlinear_activation)
        layers = [getattr(torch.nn, pad)((kernel_size - 1) // 2, **pad_params),
                  torch.nn.Conv1d(in_channels, channels[0], kernel_size, bias=bias)]
        for i, s in enumerate(upsample_scales):
            layers.append(act(**nonlinear_activation_params))
            if transposedconv:
                layers.append(torch.nn.ConvTranspose1d(
                    channels[i], channels[i + 1], s * 2, stride=s, padding=s // 2 + s % 2,
                    output_padding=s % 2, bias=bias))
            else:
                layers.append(UpsampleLayer(channels[i], channels[i + 1], upsample_rate=s,
                                            kernel_size=s * 2 + 1, stride=1, padding=s, bias=bias))
            for j in range(stacks):
                layers.append(ResidualStack(
                    kernel_size=stack_kernel_size, channels=channels[i + 1],
                    dilation=stack_kernel_size ** j, bias=bias,
                    nonlinear_activation=nonlinear_activation,
                    nonlinear_activation_params=nonlinear_activation_params,
                    pad=pad, pad_params=pad_params, use_causal_conv=use_causal_conv))
        return layers

    def _emit_layers(self, pb, dst, final_post):
        """Walk the Sequential and emit it; the activation modules are folded into
        the next conv's input stage, and ``final_post`` (tanh / ReLU) into the
        last conv's epilogue."""
        mods = list(self.melgan)
        # index of the last module that owns a conv (gets dst + final_post)
        convy = [n for n, m in enumerate(mods)
                 if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d, ResidualStack, LastLayer,
                                   UpsampleLayer, LastLinear))]
        last = convy[-1]
        a, b = pb.tmp(), pb.tmp()
        scratch = [pb.tmp(), pb.tmp()]
        cur, pending_slope = SLOT_IN, 1.0
        for n, m in enumerate(mods):
            is_last = n == last
            nxt = dst if is_last else (a if cur != a else b)
            post = final_post if is_last else POST_NONE
            if isinstance(m, torch.nn.Conv1d):
                pad, mode = self._first_pad
                pb.conv(m, cur, nxt, pad=pad, pad_mode=mode, pre_slope=pending_slope, post=post)
            elif isinstance(m, torch.nn.ConvTranspose1d):
                pb.conv_transpose(m, cur, nxt, pre_slope=pending_slope, post=post)
            elif isinstance(m, ResidualStack):
                m.emit(pb, cur, nxt, scratch, post=post, last=is_last)
            elif isinstance(m, LastLayer):
                m.emit(pb, cur, nxt, post=post)
            elif isinstance(m, UpsampleLayer):
                pb.upsample_conv(m, cur, nxt, pre_slope=pending_slope, post=post)
            elif isinstance(m, LastLinear):
                m.emit(pb, cur, nxt, scratch, post=post)
            elif isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU)):
                pending_slope = 0.0 if isinstance(m, torch.nn.ReLU) else float(m.negative_slope)
                continue
            else:
                continue  # the leading pad module and the trailing Tanh/ReLU are fused
            cur, pending_slope = nxt, 1.0


class MelGANGenerator(_MelGANTrunk):
    """Drop-in for the reference ``MelGANGenerator``."""

    def __init__(self, in_channels=80, out_channels=1, kernel_size=7,
                 channels=[512, 256, 128, 64, 32], bias=True, upsample_scales=[10, 6, 2, 2],
                 stack_kernel_size=3, stacks=3, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d",
                 pad_params={}, use_final_nonlinear_activation=True, use_weight_norm=True,
                 use_causal_conv=False):
        super().__init__()
        layers = self._build_layers(in_channels, kernel_size, channels, bias, upsample_scales,
                                    stack_kernel_size, stacks, nonlinear_activation,
                                    nonlinear_activation_params, pad, pad_params, use_causal_conv)
        layers.append(LastLayer(channels[-1], out_channels, nonlinear_activation,
                                nonlinear_activation_params, pad, kernel_size, pad_params, bias))
        self._final_post = POST_TANH if use_final_nonlinear_activation else POST_NONE
        if use_final_nonlinear_activation:
            layers.append(torch.nn.Tanh())
        self.melgan = torch.nn.Sequential(*layers)
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()
        self.pqmf = None  # attribute kept for parity with the reference (melgan.py:123)

    def _run(self, x, sync=False):
        return self._run_plan(lambda T: self._plan("trunk", lambda pb: self._emit_layers(pb, SLOT_OUT, self._final_post),
                                                   self._in_channels), x, sync=sync)

    def forward(self, c):
        """c [B,in_channels,T] -> [B, T*prod(upsample_scales)] (channel 0)."""
        return self._run(self._prepare(c))[:, 0, :]

    def _minus_plan(self, T):
        def emit(pb):
            self._emit_layers(pb, SLOT_OUT, self._final_post)
            pb.subtract_output(0, second=True)
        return self._plan("minus", emit, self._in_channels)

    def inference_minus(self, c, bias):
        """c [T,in_channels], bias [n] -> (waveform, waveform - bias) from one pass (bin/synthesize.py:74-80)."""
        c = self._prepare(c).transpose(1, 0).unsqueeze(0).contiguous()
        est, rem = self._run_minus(self._minus_plan, c, bias)
        return est.squeeze(), rem.squeeze()

    def inference(self, c):
        """c [T,in_channels] -> squeezed waveform."""
        c = self._prepare(c)
        return self._run(c.transpose(1, 0).unsqueeze(0).contiguous(), sync=True).squeeze()


class BasisMelGANGenerator(_MelGANTrunk):
    """Drop-in for the reference ``BasisMelGANGenerator``: a MelGAN-style trunk
    that ends in ReLU and predicts per-frame weights over a learned basis
    (``basis_signal_weight [L, out_channels]``); samples = overlap-add of
    ``weight @ basis^T`` with hop L/2."""

    def __init__(self, basis_signal_weight, L=30, in_channels=80, out_channels=256, kernel_size=7,
                 channels=[256, 256, 256], bias=True, upsample_scales=[4, 4], stack_kernel_size=3,
                 stacks=3, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d",
                 pad_params={}, use_final_nonlinear_activation=True, use_weight_norm=True,
                 use_causal_conv=False, transposedconv=True, lastlinear=False):
        super().__init__()
        layers = self._build_layers(in_channels, kernel_size, channels, bias, upsample_scales,
                                    stack_kernel_size, stacks, nonlinear_activation,
                                    nonlinear_activation_params, pad, pad_params, use_causal_conv,
                                    transposedconv)
        if lastlinear:
            layers.append(LastLinear(channels[-1], out_channels, bias=bias))
        self._final_post = POST_RELU if use_final_nonlinear_activation else POST_NONE
        if use_final_nonlinear_activation:
            layers.append(torch.nn.ReLU())
        self.melgan = torch.nn.Sequential(*layers)
        self.L = L
        self.basis_signal = BasisSignalLayer(basis_signal_weight, L=L)
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()
        self.pqmf = None

    def _weights(self, x):
        """Trunk + ReLU in its native layout [B, C, F]."""
        return self._run_plan(lambda T: self._plan("trunk", lambda pb: self._emit_layers(pb, SLOT_OUT, self._final_post),
                                                   self._in_channels), x)

    def _emit_full(self, pb):
        w = pb.tmp()
        self._emit_layers(pb, w, self._final_post)
        self.basis_signal.emit(pb, w, SLOT_OUT)

    def _samples(self, x, sync=False):
        """mel [B,C,T] -> [B, (F-1)*L/2 + L] in one plan (weights stay on chip/HBM
        scratch, no [B,F,L] frame tensor)."""
        return self._run_plan(lambda T: self._plan("full", self._emit_full, self._in_channels), x, sync=sync)[:, 0, :]

    def _zero_response(self, T):
        """(zero_weight [1,C,F], zero_est [1,1,n]): the generator's response to an all-zero mel of T frames.
        It depends only on (weights, T) and not on the batch -- the reference recomputes it in every
        ``forward`` (basis_melgan.py:147-152); here it is computed once per (weights, T) and kept."""
        def emit(pb):
            self._emit_layers(pb, SLOT_OUT2, self._final_post)
            self.basis_signal.emit(pb, SLOT_OUT2, SLOT_OUT)
        key = (self._fv_state(), int(T))
        cache = self.__dict__.setdefault("_fv_zero", {})
        if key not in cache:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            dev = self._device()
            zero = torch.zeros((1, self._in_channels, int(T)), dtype=torch.float32, device=dev)
            est, w = self._exec(lambda T: self._plan("zero", emit, self._in_channels), zero, out2=True)
            cache[key] = (w, est)
        return cache[key]

    def forward(self, c):
        """Reference semantics (basis_melgan.py:140-162): (est - zero_est [B, F*L/2], weight - zero_weight
        [B,F,C]).  One generator pass: both differences are formed in the epilogues of the trunk's last conv
        and of the overlap-add, against the cached zero-mel response."""
        c = self._prepare(c)
        hop = self.L // 2
        if c.shape[2] > self.max_frames_per_run:
            # longer than one run: time-chunked passes (mel and zero mel), then the overlap-add and the
            # differences on the whole tensors -- the two-pass form of the reference
            outs = []
            for inp in (torch.zeros_like(c[:1]), c):
                w = self._weights(inp)                                   # [B,C,F]
                src = self._exec(lambda T: self._plan("ola", lambda pb: self.basis_signal.emit(pb, SLOT_IN, SLOT_OUT),
                                                      w.shape[1]), w)[:, 0, :]
                outs.append((src[:, : w.shape[2] * hop], w.transpose(1, 2)))
            (zs, zw), (s, w) = outs
            return s - zs, w - zw
        zw, zs = self._zero_response(c.shape[2])

        def emit(pb):
            w = pb.tmp()
            self._emit_layers(pb, w, self._final_post)
            pb.subtract_output(0, second=True)                       # SLOT_OUT2 = weight - zero_weight
            self.basis_signal.emit(pb, w, SLOT_OUT)
            pb.subtract_output(1, second=False)                      # SLOT_OUT  = est - zero_est
        s, w = self._exec(lambda T: self._plan("forward", emit, self._in_channels), c, aux=(zw, zs), out2=True)
        return s[:, 0, : w.shape[2] * hop], w.transpose(1, 2)

    def _minus_plan(self, T):
        def emit(pb):
            self._emit_full(pb)
            pb.subtract_output(0, second=True)
        return self._plan("minus", emit, self._in_channels)

    def inference_minus(self, c, bias):
        """c [T,in_channels], bias [n] -> (waveform, waveform - bias) from one pass (bin/test.py:82-91)."""
        c = self._prepare(c).transpose(1, 0).unsqueeze(0).contiguous()
        est, rem = self._run_minus(self._minus_plan, c, bias)
        return est.squeeze(), rem.squeeze()

    def inference(self, c):
        """c [T,in_channels] -> squeezed waveform of (F-1)*L/2 + L samples."""
        c = self._prepare(c)
        return self._samples(c.transpose(1, 0).unsqueeze(0).contiguous(), sync=True).squeeze()

    def test(self, weight):
        """weight [B,F,C] -> basis synthesis only (reference basis_melgan.py:210-212)."""
        return self.basis_signal(weight)
