"""Configurations shared by the golden generator and the parity tests."""
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the six shipped yaml files: (tag, model_name, conf path)
SHIPPED = [
    ("hifigan_light", "hifigan", "conf/hifigan/light.yaml"),
    ("hifigan_large", "hifigan", "conf/hifigan/large.yaml"),
    ("mb_light", "multiband-hifigan", "conf/multiband-hifigan/light.yaml"),
    ("mb_large", "multiband-hifigan", "conf/multiband-hifigan/large.yaml"),
    ("melgan", "melgan", "conf/melgan/original.yaml"),
    ("basis", "basis-melgan", "conf/basis-melgan/light.yaml"),
]


def load_conf(path):
    with open(os.path.join(ROOT, path)) as f:
        return yaml.safe_load(f)


_H = dict(resblock_kernel_sizes=[3, 7, 11], resblock_type="1",
          resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], transposedconv=True, bias=True)
_M = dict(in_channels=80, kernel_size=7, stack_kernel_size=3, stacks=3, use_weight_norm=True,
          use_causal_conv=False)

# shrunken configs: small enough for the C oracle and for full-tensor fixtures,
# chosen to hit every kernel variant (M = 64/32/16/8/4 rows, odd strides, k < 2s,
# k > 2s, ResBlock2, no bias, no weight norm)
SMALL = [
    ("hifigan_s", "hifigan", dict(_H, upsample_rates=[8, 5, 3, 2], upsample_initial_channel=64,
                                  upsample_kernel_sizes=[16, 10, 6, 4])),
    ("hifigan_rb2", "hifigan", dict(_H, resblock_type="2", upsample_rates=[4, 3],
                                    upsample_initial_channel=128, upsample_kernel_sizes=[8, 7],
                                    resblock_kernel_sizes=[3, 5],
                                    resblock_dilation_sizes=[[1, 3], [2, 4]], bias=False)),
    ("mb_s", "multiband-hifigan", dict(_H, upsample_rates=[10, 6], upsample_initial_channel=64,
                                       upsample_kernel_sizes=[20, 12])),
    ("mb_s_k16", "multiband-hifigan", dict(_H, upsample_rates=[10, 6], upsample_initial_channel=32,
                                           upsample_kernel_sizes=[16, 16])),
    ("melgan_s", "melgan", dict(_M, out_channels=1, channels=[64, 32, 16, 8, 4],
                                upsample_scales=[10, 6, 2, 2])),
    ("melgan_nown", "melgan", dict(_M, out_channels=1, channels=[32, 32, 16], upsample_scales=[5, 3],
                                   use_weight_norm=False)),
    ("basis_s", "basis-melgan", dict(_M, L=30, out_channels=32, channels=[32, 32, 32],
                                     upsample_scales=[4, 4], transposedconv=True)),
    # transposedconv: False -> UpsampleLayer (nearest repeat + conv), even and odd kernels
    ("hifigan_up", "hifigan", dict(_H, transposedconv=False, upsample_rates=[8, 3],
                                   upsample_initial_channel=64, upsample_kernel_sizes=[16, 7])),
    ("basis_up", "basis-melgan", dict(_M, L=30, out_channels=32, channels=[32, 16, 32],
                                      upsample_scales=[4, 3], transposedconv=False)),
    # use_causal_conv: CausalConv1d inside every ResidualStack; lastlinear: BatchNorm head
    ("melgan_causal", "melgan", dict(_M, out_channels=1, channels=[32, 32, 16], upsample_scales=[5, 3],
                                     use_causal_conv=True)),
    ("basis_causal_ll", "basis-melgan", dict(_M, L=30, out_channels=24, channels=[32, 32, 16],
                                             upsample_scales=[4, 4], transposedconv=True,
                                             use_causal_conv=True, lastlinear=True)),
]
SMALL_T = 24
SMALL_B = 3
FULL_T = 64       # shipped configs: full-output fixtures
STATS_T = 1000    # shipped configs: strided samples + sums at the benchmark length
STRIDE_N = 1024
