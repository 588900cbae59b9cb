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


def _row_prescale(w):
    """api.hip row_scale_kernel: per output row 2^e with max|w_row| 2^e in [2^13, 2^14) (e = 0 for an all-zero row)."""
    m = np.abs(w.reshape(w.shape[0], -1)).max(axis=1)
    _, x = np.frexp(np.where(m > 0, m, 1.0))          # m = f 2^x, f in [0.5, 1)
    e = np.where(m > 0, np.clip(14 - x, -110, 110), 0)
    return np.ldexp(np.float32(1.0), e).astype(np.float32)


def _split_conv(x, w, T, prescale=True):
    """The kernels' arithmetic: weights scaled per row by a power of two at pack time, split products accumulated in fp32,
    the accumulated sum multiplied by the inverse scale (exact) in the epilogue."""
    s = _row_prescale(w) if prescale else np.ones(w.shape[0], np.float32)
    x1, x2 = _split(x)
    w1, w2 = _split((w * s[:, None, None]).astype(np.float32))
    hi = _conv(x1, w1, T, np.float32)
    lo = _conv(x2, w1, T, np.float32) + _conv(x1, w2, T, np.float32)
    return (hi + lo * np.float32(1.0 / 2048.0)) * (np.float32(1.0) / s)[:, None]


K_SPLIT_LOW = 2.0 ** -10                              # pairh_kernels.hpp kSplitLow


def _low_guard(x):
    """pairh_kernels.hpp low_flag: operands not all zero and all below 2^-10 -> the call is repeated on fp32 kernels."""
    m = float(np.abs(x).max())
    return 0.0 < m < K_SPLIT_LOW


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


def _lrelu(x):
    return np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)


@pytest.mark.parametrize("C,k", [(16, 11), (32, 7), (64, 7), (128, 3)])
@pytest.mark.parametrize("wexp,aexp", [(-10, 10), (-13, 13), (-14, 14), (-17, 14), (-20, 14), (-30, 14), (17, -5), (-14, 0)])
def test_weight_scale_does_not_matter(C, k, wexp, aexp):
    """The LOW side of the weights' domain: weights x 2^-10 ... 2^-30 -- far below the smallest
    normal f16, 6.1e-5 -- with activations x 2^10 ... 2^14, and the other way round.  With the per-row power-of-two prescale of
    the pack kernels the split conv stays within 3x the error of a plain fp32 accumulation (+ 1e-7 of the output's
    scale); WITHOUT it (round 3's kernels) the same inputs are 5x ... 100x worse than fp32 -- which pins that this
    test sees the problem."""
    rs = np.random.RandomState(C * 1000 + k * 10 + (wexp % 7))
    T = 512
    x = _lrelu(rs.randn(C, T + k - 1) * 0.5) * np.float32(2.0 ** aexp)
    w = ((rs.randn(C, C, k) / np.sqrt(C * k)) * 2.0 ** wexp).astype(np.float32)
    w *= (2.0 ** rs.randint(-3, 4, size=(C, 1, 1))).astype(np.float32)      # rows of different scale (weight norm's g)
    assert not _low_guard(x)
    ref = _conv(x, w, T, np.float64)
    scale = np.abs(ref).max()
    e32 = np.abs(_conv(x, w, T, np.float32) - ref).max()
    esp = np.abs(_split_conv(x, w, T) - ref).max()
    assert esp <= 3.0 * e32 + 1e-7 * scale, (esp / e32, esp / scale)
    if wexp <= -17:
        eraw = np.abs(_split_conv(x, w, T, prescale=False) - ref).max()
        assert eraw >= (5.0 if wexp > -20 else 30.0) * e32, eraw / e32


def test_prescale_is_exact_and_changes_nothing_for_ordinary_weights():
    """Scaling by a power of two and back is exact; for weights that were inside the normal f16 range anyway the
    prescaled conv differs from the unscaled one only through second halves (h2) that were subnormal unscaled: far below
    the fp32 rounding of the sums."""
    rs = np.random.RandomState(3)
    C, k, T = 32, 7, 256
    x = _lrelu(rs.randn(C, T + k - 1))
    w = rs.randn(C, C, k).astype(np.float32)
    w = np.where(np.abs(w) < 2.0 ** -9, np.float32(2.0 ** -9), w).astype(np.float32) * np.float32(0.0625)
    a, b = _split_conv(x, w, T), _split_conv(x, w, T, prescale=False)
    assert np.abs(a - b).max() <= 2.0 ** -24 * np.abs(a).max()
    s = _row_prescale(w)
    m = np.abs(w.reshape(C, -1)).max(axis=1) * s
    assert (m >= 2.0 ** 13).all() and (m < 2.0 ** 14).all()
    assert _row_prescale(np.zeros((2, 2, 3), np.float32)).tolist() == [1.0, 1.0]


@pytest.mark.parametrize("aexp", [-2, -6, -9])
def test_small_activations_inside_the_guarded_domain(aexp):
    """The LOW side for activations: a tensor whose largest magnitude is at least 2^-10 passes the guard, and the split
    conv of it (weights 2^-aexp larger, output O(1)) is as accurate as fp32; below that the guard fires and the call
    is repeated on the fp32 kernels (test_gpu_pairs.py::test_low_side_of_the_range_guard)."""
    rs = np.random.RandomState(40 - aexp)
    C, k, T = 64, 7, 512
    x = _lrelu(rs.randn(C, T + k - 1) * 0.5)
    x = (x / np.abs(x).max() * np.float32(2.0 ** aexp)).astype(np.float32)     # max |x| = 2^aexp exactly
    w = ((rs.randn(C, C, k) / np.sqrt(C * k)) * 2.0 ** -aexp).astype(np.float32)
    assert not _low_guard(x)
    ref = _conv(x, w, T, np.float64)
    e32 = np.abs(_conv(x, w, T, np.float32) - ref).max()
    esp = np.abs(_split_conv(x, w, T) - ref).max()
    assert esp <= 3.0 * e32 + 1e-7 * np.abs(ref).max(), esp / e32
    assert _low_guard(x * np.float32(0.25) if aexp == -9 else x * np.float32(2.0 ** (-11 - aexp)))
    assert not _low_guard(np.zeros(4, np.float32))


@pytest.mark.parametrize("d", [3, 5, 6, 7, 9, 11, 12, 15])
def test_three_instruction_division_is_correctly_rounded(d):
    """csrc/pair_kernels.hpp div_exact -- q0 = v r; q = fma(fma(-d, q0, v), r, q0), r = RN(1 / d) -- for the MRF mean
    (hifigan.py:103: xs / num_kernels): equal to the correctly rounded v / d for every significand (all 2^23 of a binade,
    both signs, several binades: correct rounding does not depend on the exponent while the quotient is a normal number).
    float64 emulates the fused multiply-adds exactly here: d q0 and e r are products of a <= 4-bit and a 24-bit, resp. two
    24-bit significands.  (The GPU repeats the check against its own division: test_gpu_parity.py.)"""
    d32 = np.float32(d)
    r = np.float32(1.0) / d32
    m = np.arange(2 ** 23, 2 ** 24, dtype=np.int64).astype(np.float64)
    for e, sgn in ((-40, 1.0), (0, 1.0), (0, -1.0), (60, -1.0)):
        if True:
            v = (sgn * m * 2.0 ** (e - 23)).astype(np.float32)
            q0 = (v * r).astype(np.float32)
            res = (v.astype(np.float64) - np.float64(d) * q0.astype(np.float64)).astype(np.float32)     # fma(-d, q0, v): exact
            q = (q0.astype(np.float64) + res.astype(np.float64) * np.float64(r)).astype(np.float32)
            want = (v.astype(np.float64) / np.float64(d)).astype(np.float32)
            assert np.array_equal(q, want), (d, e, sgn, int((q != want).sum()))
