"""HiFi-GAN and Multiband-HiFi-GAN generators on libfastvocoder_hip.so.

Same constructor kwargs (the conf/hifigan/*.yaml and conf/multiband-hifigan/*.yaml
keys), method set and ``state_dict`` keys as the reference's
``HiFiGANGenerator`` (/root/reference/model/generator/hifigan.py:13-129) and
``MultiBandHiFiGANGenerator`` (multiband_hifigan.py:14-137).

Graph (hifigan.py:92-106):
    x = conv_pre(mel)
    per stage i:  x = ConvTranspose1d_i(lrelu(x, 0.1));  x = mean_j ResBlock_{i,j}(x)
    y = tanh(conv_post(lrelu(x, 0.01)))          # default slope 0.01, not 0.1
Every activation, bias, residual add, the MRF sum and its /num_kernels, and
tanh are epilogue/prologue work of the conv kernels: the whole forward is
``num_convs`` launches (78 for the shipped 4-stage configs) and nothing else.
"""
import torch

from .._native import PAIR_SPLIT_F16
from .engine import (NativeModule, PlanBuilder, POST_TANH, SLOT_IN, SLOT_NONE, SLOT_OUT, pair_precision,  # noqa: F401
                     weight_norm)
from .modules import LRELU_SLOPE, ResBlock1, ResBlock2, UpsampleLayer
from .pqmf import PQMF

DEFAULT_LRELU_SLOPE = 0.01  # F.leaky_relu's default, used before conv_post (hifigan.py:104)


_CU_COUNT = {}      # device -> compute units (asked once per device: the policy below runs on every forward)


def stage32_windows_fit(cols, cus, fill=0.65):
    """Do the 32-channel one-launch kernel's fixed windows fit ``cols`` columns (batch x samples of the stage) on ``cus``
    CUs?  The launcher (csrc/mrfh_launch.hip) gives every block -- one per CU, at most one per 128 columns -- an equal share;
    a share takes one run-in window of 384 columns (264 of them final) plus whole windows of 324 final columns.  True when
    at least ``fill`` of the windows' columns are the share's (measured crossover: _HiFiGANBase._stage_one_launch)."""
    cols = int(cols)
    if cols <= 0:
        return False
    nblk = max(1, min(int(cus), -(-cols // 128)))
    share = -(-cols // nblk)
    windows = 1 if share <= 264 else 1 + -(-(share - 264) // 324)
    return share >= fill * 384 * windows


class _HiFiGANBase(NativeModule):
    _post_channels = 1
    fuse_pqmf = True      # Multiband: conv_post + tanh + PQMF synthesis as one launch (False: two; A/B and bit-identity tests)

    def __init__(self, resblock_kernel_sizes, upsample_rates, upsample_initial_channel,
                 resblock_type, upsample_kernel_sizes, resblock_dilation_sizes, transposedconv,
                 bias):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        c0 = upsample_initial_channel
        self.conv_pre = torch.nn.Conv1d(80, c0, 7, 1, padding=3, bias=bias)
        block = ResBlock1 if resblock_type == "1" else ResBlock2

        self.ups = torch.nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
            if transposedconv:
                up = torch.nn.ConvTranspose1d(cin, cout, k, u, padding=(u // 2 + u % 2),
                                              output_padding=u % 2, bias=bias)
            else:
                up = UpsampleLayer(cin, cout, upsample_rate=u, kernel_size=k, stride=1,
                                   padding=k // 2, bias=bias)
            self.ups.append(up)

        self.resblocks = torch.nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(block(ch, k, d, bias=bias))
        self.conv_post = torch.nn.Conv1d(ch, self._post_channels, 7, 1, padding=3, bias=bias)

    def _finish_init(self):
        # same order as the reference: weight norm first, then the (therefore
        # ineffective) normal_(0, 0.01) reset -- SURVEY.md section 8 a-13
        self.apply_weight_norm()
        self.reset_parameters()

    # -- fused ResBlock pairs ------------------------------------------------
    def _stage_fusable(self, i, precision):
        """Stage i can run on the fused ResBlock-pair kernels (csrc/pair_kernels.hpp, pairh_kernels.hpp, convp / convh):
        the classic ResBlock1 trio (3 / 7 / 11 taps) with one dilation per pair position, at a channel count the
        kernels of ``precision`` are built for."""
        nk = self.num_kernels
        blocks = [self.resblocks[i * nk + j] for j in range(nk)]
        if nk != 3 or not all(isinstance(b, ResBlock1) for b in blocks):
            return False
        if sorted(b.convs1[0].kernel_size[0] for b in blocks) != [3, 7, 11]:
            return False
        dils = [[c.dilation[0] for c in b.convs1] for b in blocks]
        if any(d != dils[0] for d in dils):
            return False
        prec = pair_precision(precision, blocks[0].channels)
        return all(PlanBuilder.pair_fusable(c1, c2, prec) for b in blocks for c1, c2 in zip(b.convs1, b.convs2))

    def _fused_flags(self, T):
        """Per stage: run it fused for a mel of T frames?  The pair kernels need 16-byte aligned rows
        (stage length % 4 == 0); ``fuse_pairs = False`` keeps every stage on the conv-by-conv path (A/B runs)."""
        precision, fuse = self._fv_policy()[:2]
        if not fuse:
            return (False,) * self.num_upsamples
        flags, t = [], int(T)
        for i in range(self.num_upsamples):
            up = self.ups[i]
            if isinstance(up, UpsampleLayer):
                t = t * up.upsample_rate + 2 * up.conv.padding[0] - (up.conv.kernel_size[0] - 1)
            else:
                t = (t - 1) * up.stride[0] - 2 * up.padding[0] + up.kernel_size[0] + up.output_padding[0]
            fusable = t > 0 and t % 4 == 0 and self._stage_fusable(i, precision)
            ch = self.resblocks[i * self.num_kernels].channels
            flags.append("s" if fusable and self._stage_one_launch(ch, t) else bool(fusable))
        return tuple(flags)

    def _stage_one_launch(self, channels, t):
        """A fused stage of ``t`` samples as ONE launch (csrc/mrfh_kernels.hpp, mrfw_kernels.hpp) rather than as pair
        launches?  ``fuse_stage`` = a tuple of widths: exactly those; False: none; True (default): 16 channels always; 32
        channels when the kernel's fixed 384-column windows fit the work -- a block's share of the batch's columns takes
        one run-in window (264 final columns) plus whole windows of 324, and the pair launches, whose tiles are cut to the
        share, win when much of the last window is empty.  Measured on whole forwards (tools/stage_policy_bench.py, us,
        pairs / one launch): batch 1, 560 frames (share 263 of one window: 0.68) 457 / 440; 1000 frames (469 of two: 0.61)
        603 / 619; 700 frames (0.43) 494 / 537; batch 4, 1000 frames (1875 of six: 0.81) 1938 / 1850; 500 frames (0.61)
        1093 / 1120; 700 frames (0.68) 1407 / 1403 -- one launch from 0.65 up."""
        fs = self.fuse_stage
        if isinstance(fs, (tuple, list)):
            return channels in fs
        if not fs or channels == 16:
            return bool(fs)
        if channels != 32:
            return False
        dev = self._device()
        cus = _CU_COUNT.get(dev)
        if cus is None:
            cus = _CU_COUNT[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
        return stage32_windows_fit(getattr(self, "_fv_batch", 1) * int(t), cus)

    def _emit_fused_stage(self, pb, blocks, up, x, scratch, parts, fold=None, merge_next=False, one_launch=False):
        """The three ResBlocks of a stage as fused pair launches: every pair position is ONE launch of
        three members; the last position also forms the MRF mean when the three weight sets fit in
        LDS (16 channels), otherwise it runs conv by conv (grouped first convs + the merged last convs).
        ``merge_next`` (split-f16 stages in front of a split-f16 upsampler): the last position is an ordinary three-member
        launch too -- block 0 stores r_0 in ``x``, blocks 1, 2 store r_1, r_2 in ``parts`` -- and the UPSAMPLER forms
        ((r_0 + r_1) + r_2) / nk while it loads its window (PlanBuilder.conv_transpose(merge=...)): one launch less per
        stage, the same bits."""
        nk = len(blocks)
        npairs = len(blocks[0].convs1)
        ch = blocks[0].channels
        curs = [up] * nk
        prec = pb.pair_precision(ch)
        if one_launch and pb.mrf_stage_supported(blocks):
            # 16 / 32 channels: the whole stage -- nine pairs, the mean, and conv_post when it folds -- is ONE launch
            # (csrc/mrfh_kernels.hpp, mrfw_kernels.hpp): the stage's tensor is read once and written once (or not at all)
            if fold is not None:
                pb.mrf_stage(blocks, up, fold[3], LRELU_SLOPE, float(nk), fold=fold[:3])
            else:
                pb.mrf_stage(blocks, up, x, LRELU_SLOPE, float(nk))
            return
        if prec == PAIR_SPLIT_F16 and ch > 128:
            # 256 / 512 channels (HiFi-GAN large): a pair is two launches of the split-f16 conv kernel
            # (csrc/convh_kernels.hpp); 64 and 128 channels run fused (csrc/convp_kernels.hpp, convq_kernels.hpp)
            # below.  Per pair position: the three blocks' first convs, then their second convs
            # (+ residual); at the last position the second convs of blocks 1.. store r_1.., and the first block's
            # runs after them with ((r_0 + r_1) + r_2) / nk in its epilogue -- the reference's order, bit for bit.
            for pi in range(npairs):
                last = pi == npairs - 1
                pb.begin_group()
                for j in range(nk):
                    pb.conv_split(blocks[j].convs1[pi], curs[j], scratch[j][0], LRELU_SLOPE)
                pb.end_group()
                nxt = []
                pb.begin_group()
                for j in range(1 if (last and not merge_next) else 0, nk):
                    ping, pong = scratch[j][1], scratch[j][2]
                    d = (x if j == 0 else parts[j - 1]) if last else (ping if curs[j] != ping else pong)
                    pb.conv_split(blocks[j].convs2[pi], scratch[j][0], d, LRELU_SLOPE, res=curs[j])
                    nxt.append(d)
                pb.end_group()
                if last and not merge_next:
                    pb.conv_split(blocks[0].convs2[pi], scratch[0][0], x, LRELU_SLOPE, res=curs[0],
                                  add1=parts[0], add2=parts[1] if nk > 2 else SLOT_NONE, out_div=float(nk))
                curs = nxt
            return
        for pi in range(npairs - 1):
            pb.begin_group()
            nxt = []
            for j in range(nk):
                ping, pong = scratch[j][1], scratch[j][2]
                d = ping if curs[j] != ping else pong
                pb.pair(blocks[j].convs1[pi], blocks[j].convs2[pi], curs[j], d, LRELU_SLOPE, prec, mid=scratch[j][0])
                nxt.append(d)
            pb.end_group()
            curs = nxt
        if prec == PAIR_SPLIT_F16 and merge_next:
            pb.begin_group()
            for j in range(nk):
                pb.pair(blocks[j].convs1[-1], blocks[j].convs2[-1], curs[j], x if j == 0 else parts[j - 1], LRELU_SLOPE, prec,
                        mid=scratch[j][0])
            pb.end_group()
            return
        if prec == PAIR_SPLIT_F16:
            # the 7- and 11-tap blocks' last pairs share a launch and store r_1, r_2; the first block's last pair
            # runs after them and forms ((r_0 + r_1) + r_2) / nk in its epilogue: the reference's order, bit for bit
            pb.begin_group()
            for j in range(1, nk):
                pb.pair(blocks[j].convs1[-1], blocks[j].convs2[-1], curs[j], parts[j - 1], LRELU_SLOPE, prec,
                        mid=scratch[j][0])
            pb.end_group()
            if fold is not None:        # conv_post inside the stage's last launch: its output goes to fold[3]
                pb.pair(blocks[0].convs1[-1], blocks[0].convs2[-1], curs[0], fold[3], LRELU_SLOPE, prec,
                        add1=parts[0], add2=parts[1], out_div=float(nk), mid=scratch[0][0], fold=fold[:3])
                return
            pb.pair(blocks[0].convs1[-1], blocks[0].convs2[-1], curs[0], x, LRELU_SLOPE, prec,
                    add1=parts[0], add2=parts[1], out_div=float(nk), mid=scratch[0][0])
            return
        if ch == 16:
            pb.mrf_sum([(b.convs1[-1], b.convs2[-1]) for b in blocks], curs, x, LRELU_SLOPE, float(nk))
            return
        states = [dict(cur=c) for c in curs]
        steps = blocks[0].num_steps()
        pb.begin_group()
        for j in range(nk):
            blocks[j].emit_step(pb, steps - 2, states[j], up, x, scratch[j])
        pb.end_group()
        convs, srcs, ress = zip(*[blocks[j].last_conv_inputs(states[j], up, scratch[j]) for j in range(nk)])
        pb.conv_sum3(convs, srcs, ress, parts[:2], x, pre_slope=LRELU_SLOPE, out_div=float(nk))

    # -- op emission ---------------------------------------------------------
    def _fold_post(self, pb, fused):
        """conv_post can run inside the last stage's last launch (PlanBuilder.pair_fold_supported): the classic trio
        of 16-channel ResBlock1s as split-f16 fused pairs, one output channel."""
        if not pb.fold_post or not fused or not fused[-1] or self.num_kernels != 3:
            return False
        blocks = self.resblocks[-3:]
        # (a 32-channel last stage -- HiFi-GAN large -- folds only as the one-launch stage: the pair kernels' fold is 16 channels)
        return PlanBuilder.pair_fold_supported(blocks[0].convs1[-1], self.conv_post, pb.pair_precision(blocks[0].channels),
                                               stage=fused[-1] == "s" and pb.mrf_stage_supported(blocks))

    def _emit_trunk(self, pb, dst, fused=None, fold_post=False, pqmf=None):
        """mel (SLOT_IN) -> tanh(conv_post(...)) in ``dst``; ``fused``: per-stage flags (_fused_flags); ``fold_post``:
        conv_post inside the last pair's launch (not for the plans whose LAST op subtracts an output offset); ``pqmf``:
        a synthesis filter -- ``dst`` then receives the full-band signal, conv_post + tanh + PQMF synthesis being ONE
        launch (PlanBuilder.conv_post_pqmf; two where that kernel does not apply)."""
        fold_post = fold_post and self._fold_post(pb, fused)
        x, up = pb.tmp(), pb.tmp()
        nk = self.num_kernels
        # The nk ResBlocks of a stage are independent given the upsampled input, and at
        # batch 1 one conv cannot fill 256 CUs.  Their steps are therefore emitted
        # interleaved -- conv number s of every block forms one GROUP, which the
        # executor runs as a single launch when the blocks are the classic 3/7/11-tap
        # trio (fv_plan_set_group), else one launch each.  Fused stages: _emit_fused_stage.
        # Conv-by-conv stages (shapes the pair kernels are not built for, rows that are not
        # 16-byte aligned): the three last convs accumulate into one output tile in ONE launch
        # (conv_sum3) when they are the classic trio; otherwise the blocks after the first
        # store r_1, r_2 and the FIRST block's final conv -- the cheapest -- runs after them and
        # forms ((r_0 + r_1) + r_2) / nk in its epilogue (own value first: fv_plan_set_sum_order),
        # the reference's summation order bit for bit (hifigan.py:97-103).
        # Measured on MI355X (HiFi-GAN light, B = 1, per forward): see DESIGN.md section 3.2.
        parts = [pb.tmp() for _ in range(nk - 1)]          # r_1 .. r_{nk-1}
        scratch = [[pb.tmp(), pb.tmp(), pb.tmp()] for _ in range(nk)]
        pb.conv(self.conv_pre, SLOT_IN, x)
        merged = False        # the stage in front left r_0, r_1, r_2 in x, parts[0], parts[1]: this upsampler merges them
        for i in range(self.num_upsamples):
            if isinstance(self.ups[i], UpsampleLayer):
                pb.upsample_conv(self.ups[i], x, up, pre_slope=LRELU_SLOPE)
            else:
                pb.conv_transpose(self.ups[i], x, up, pre_slope=LRELU_SLOPE,
                                  merge=(parts[0], parts[1], float(nk)) if merged else None)
            merged = False
            blocks = [self.resblocks[i * nk + j] for j in range(nk)]
            if fused is not None and fused[i]:
                if fold_post and i == self.num_upsamples - 1:
                    self._emit_fused_stage(pb, blocks, up, x, scratch, parts, one_launch=fused[i] == "s",
                                           fold=(self.conv_post, DEFAULT_LRELU_SLOPE, POST_TANH, dst))
                    return
                # the MRF merge inside the NEXT upsampler (engine.NativeModule.merge_in_upsampler)
                merged = (self.merge_in_upsampler and nk == 3 and i + 1 < self.num_upsamples
                          and pb.pair_precision(blocks[0].channels) == PAIR_SPLIT_F16
                          and pb.conv_transpose_takes_merge(self.ups[i + 1])
                          and not (fused[i] == "s" and pb.mrf_stage_supported(blocks)))
                self._emit_fused_stage(pb, blocks, up, x, scratch, parts, merge_next=merged, one_launch=fused[i] == "s")
                continue
            if nk <= 3:
                steps = blocks[0].num_steps()
                states = [dict() for _ in range(nk)]
                sum3 = (nk == 3 and all(isinstance(b, ResBlock1) for b in blocks)
                        and sorted(b.convs2[-1].kernel_size[0] for b in blocks) == [3, 7, 11])
                if sum3:
                    for st in range(steps - 1):
                        pb.begin_group()
                        for j in range(nk):
                            blocks[j].emit_step(pb, st, states[j], up, x, scratch[j])
                        pb.end_group()
                    convs, srcs, ress = zip(*[blocks[j].last_conv_inputs(states[j], up, scratch[j])
                                              for j in range(nk)])
                    pb.conv_sum3(convs, srcs, ress, parts[:2], x, pre_slope=LRELU_SLOPE, out_div=float(nk))
                    continue
                others = list(range(1, nk))
                for st in range(steps):
                    final = st == steps - 1
                    members = others if final else range(nk)
                    pb.begin_group()
                    for j in members:
                        blocks[j].emit_step(pb, st, states[j], up, parts[others.index(j)] if final else x,
                                            scratch[j])
                    pb.end_group()
                # the first block's final conv: ((r_0 + r_1) + r_2) / nk in the reference's order
                blocks[0].emit_step(pb, steps - 1, states[0], up, x, scratch[0],
                                    acc=parts[0] if nk > 1 else SLOT_NONE,
                                    acc2=parts[1] if nk > 2 else SLOT_NONE, out_div=float(nk), own_first=True)
            else:
                # generic: a running sum chained through the blocks, one after the other
                for j in range(nk):
                    last = j == nk - 1
                    blocks[j].emit(pb, up, x if last else parts[0], scratch[0],
                                   acc=parts[0] if j > 0 else SLOT_NONE,
                                   out_div=float(nk) if last else 1.0)
        if pqmf is not None and pb.post_pqmf_supported(self.conv_post, pqmf) and self.fuse_pqmf:
            pb.conv_post_pqmf(self.conv_post, pqmf, x, dst, pre_slope=DEFAULT_LRELU_SLOPE, post=POST_TANH)
        elif pqmf is not None:
            sub = pb.tmp()
            pb.conv(self.conv_post, x, sub, pre_slope=DEFAULT_LRELU_SLOPE, post=POST_TANH)
            pb.pqmf_synthesis(pqmf, sub, dst)
        else:
            pb.conv(self.conv_post, x, dst, pre_slope=DEFAULT_LRELU_SLOPE, post=POST_TANH)

    def _flag_tag(self, T):
        return "".join("s" if f == "s" else "f" if f else "-" for f in self._fused_flags(T))

    # (which stages run fused depends on the policy in force -- after a range overflow the fp32 pair kernels exist at 16 /
    # 32 channels only -- so plan names and emit functions evaluate _fused_flags when they are used, not here)
    def _trunk_plan(self, T):
        return self._plan(lambda: "trunk" + self._flag_tag(T),
                          lambda pb: self._emit_trunk(pb, SLOT_OUT, self._fused_flags(T), fold_post=True), 80)

    def _emit_inference(self, pb, fused):
        self._emit_trunk(pb, SLOT_OUT, fused)

    def _minus_plan(self, T):
        """The inference graph whose last op also writes (output - auxiliary input) to the second output."""
        def emit(pb):
            self._emit_inference(pb, self._fused_flags(T))
            pb.subtract_output(0, second=True)
        return self._plan(lambda: "minus" + self._flag_tag(T), emit, 80)

    def inference_minus(self, x, bias):
        """x [T,80], bias [n] (e.g. the response to an all-zero mel) -> (waveform, waveform - bias), both 1-D,
        from one generator pass: what bin/synthesize.py:74-80 forms with a second pass and a subtraction."""
        x = self._prepare(x).transpose(1, 0).unsqueeze(0).contiguous()
        est, rem = self._run_minus(self._minus_plan, x, bias)
        return est.squeeze(), rem.squeeze()

    def _trunk(self, x, sync=False):
        return self._run_plan(self._trunk_plan, x, sync=sync)


class HiFiGANGenerator(_HiFiGANBase):
    """Drop-in for the reference ``HiFiGANGenerator``."""

    def __init__(self, resblock_kernel_sizes=[3, 7, 11], upsample_rates=[8, 5, 3, 2],
                 upsample_initial_channel=256, resblock_type="1",
                 upsample_kernel_sizes=[16, 10, 6, 4],
                 resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 transposedconv=True, bias=True):
        super().__init__(resblock_kernel_sizes, upsample_rates, upsample_initial_channel,
                         resblock_type, upsample_kernel_sizes, resblock_dilation_sizes,
                         transposedconv, bias)
        self._finish_init()

    def forward(self, x):
        """x [B,80,T] -> waveform [B, prod(upsample_rates)*T]."""
        return self._trunk(self._prepare(x))[:, 0, :]

    def inference(self, x):
        """x [T,80] (ndarray or tensor) -> 1-D waveform."""
        x = self._prepare(x)
        return self._trunk(x.transpose(1, 0).unsqueeze(0).contiguous(), sync=True).squeeze()


class MultiBandHiFiGANGenerator(_HiFiGANBase):
    """Drop-in for the reference ``MultiBandHiFiGANGenerator``: ``forward``
    returns the 4 sub-bands, ``inference`` adds PQMF synthesis."""

    _post_channels = 4

    def __init__(self, resblock_kernel_sizes=[3, 7, 11], upsample_rates=[10, 6],
                 upsample_initial_channel=256, resblock_type="1",
                 upsample_kernel_sizes=[20, 12],
                 resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 transposedconv=True, bias=True):
        super().__init__(resblock_kernel_sizes, upsample_rates, upsample_initial_channel,
                         resblock_type, upsample_kernel_sizes, resblock_dilation_sizes,
                         transposedconv, bias)
        self.pqmf = PQMF()
        self._finish_init()

    def forward(self, x):
        """x [B,80,T] -> sub-bands [B,4,T'] (the caller applies pqmf.synthesis,
        reference bin/train.py:96)."""
        return self._trunk(self._prepare(x))

    def _emit_full(self, pb, fused=None):
        self._emit_trunk(pb, SLOT_OUT, fused, pqmf=self.pqmf.synthesis_filter)

    def _emit_inference(self, pb, fused):
        self._emit_full(pb, fused)

    def _full_plan(self, T):
        return self._plan(lambda: "inference" + self._flag_tag(T),
                          lambda pb: self._emit_full(pb, self._fused_flags(T)), 80)

    def inference(self, x):
        """x [T,80] -> 1-D full-band waveform (trunk + PQMF synthesis, one plan)."""
        x = self._prepare(x).transpose(1, 0).unsqueeze(0).contiguous()
        return self._run_plan(self._full_plan, x, sync=True).squeeze()

    def synthesize_batch(self, x):
        """x [B,80,T] -> full-band waveforms [B, 4*T'] (batched ``inference``)."""
        return self._run_plan(self._full_plan, self._prepare(x))[:, 0, :]
