"""Per-layer proof of the split-f16 arithmetic of csrc/pairh_kernels.hpp / convh_kernels.hpp (SURVEY.md section 7:
"any reduced-precision trick must be proven against the budget per layer"), in numpy, no GPU:

    v = h1 + h2 / 2048 + e,  h1 = f16(v), h2 = f16((v - h1) * 2048);   a * b ~ a1 b1 + (a1 b2 + a2 b1) / 2048

with exact f16 x f16 products accumulated in fp32.  Against a float64 reference the result must be as close as a
plain fp32 accumulation of fp32 products is -- at every activation scale the f16 range admits.
"""
import numpy as np
import pytest


def _split(a):
    h1 = a.astype(np.float16)
    h2 = ((a - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return h1.astype(np.float32), h2.astype(np.float32)


def _conv(x, w, T, dtype):
    C, _, k = w.shape
    y = np.zeros((C, T), dtype)
    for j in range(k):                       # fp32 (or fp64) accumulation over taps; the matmul accumulates in `dtype`
        y += w[:, :, j].astype(dtype) @ x[:, j:j + T].astype(dtype)
    return y


def _split_conv(x, w, T):
    x1, x2 = _split(x)
    w1, w2 = _split(w)
    hi = _conv(x1, w1, T, np.float32)
    lo = _conv(x2, w1, T, np.float32) + _conv(x1, w2, T, np.float32)
    return hi + lo * np.float32(1.0 / 2048.0)


@pytest.mark.parametrize("C,k", [(16, 11), (32, 7), (64, 11), (128, 3)])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 3e2])
def test_split_f16_products_are_fp32_class(C, k, scale):
    rs = np.random.RandomState(C * 100 + k)
    T = 1024
    x = (rs.randn(C, T + k - 1) * 0.5 * scale).astype(np.float32)
    x = np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)          # a leaky-ReLU'd activation
    w = (rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)
    ref = _conv(x, w, T, np.float64)
    e32 = np.abs(_conv(x, w, T, np.float32) - ref)
    esp = np.abs(_split_conv(x, w, T) - ref)
    rms32, rmssp = np.sqrt((e32 ** 2).mean()), np.sqrt((esp ** 2).mean())
    assert rmssp <= 1.6 * rms32 and esp.max() <= 3.0 * e32.max(), (rmssp / rms32, esp.max() / e32.max())
    assert esp.max() <= 4e-6 * np.abs(ref).max()


def test_split_is_exact_to_2_pow_minus_22():
    rs = np.random.RandomState(5)
    v = np.concatenate([rs.randn(100000) * s for s in (1e-4, 1e-2, 1.0, 1e2, 1e4)]).astype(np.float32)
    h1, h2 = _split(v)
    back = h1.astype(np.float64) + h2.astype(np.float64) / 2048.0
    rel = np.abs(back - v.astype(np.float64)) / np.maximum(np.abs(v.astype(np.float64)), 1e-30)
    big = np.abs(v) >= 2.0 ** -13                 # below that h1 is a subnormal f16 and the error is absolute:
    assert rel[big].max() <= 2.0 ** -21           #   2^-22 relative (one ulp of slack for the double rounding)
    assert np.abs(back - v)[~big].max() <= 2.0 ** -36
    # beyond the f16 range the scheme does not apply (documented limit of the kernels; FV_PAIR_PREC=f32)
    assert not np.isfinite(_split(np.array([7e4], np.float32))[0]).all()
