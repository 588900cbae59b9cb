"""PQMF filter bank: design on the host, synthesis on the GPU.

Mirrors the surface of the reference's ``PQMF`` module
(/root/reference/model/generator/pqmf.py:51-135): same constructor arguments,
same three registered buffers (``analysis_filter [S,1,taps+1]``,
``synthesis_filter [1,S,taps+1]``, ``updown_filter [S,S,S]``) so a
Multiband-HiFi-GAN checkpoint's ``pqmf.*`` entries load unchanged, and
``synthesis(x[B,S,T]) -> [B,1,S*T]``.

The reference realises synthesis as zero-stuffing (a one-hot ConvTranspose1d)
followed by a dense 63-tap FIR over 4 channels; 3 of every 4 products are with
stuffed zeros.  Here it is ONE polyphase HIP kernel (csrc/pqmf.hip): each output
sample touches only the <=16 non-zero taps per band.
"""
import numpy as np
import torch

from .. import _native


def design_prototype_filter(taps=62, cutoff_ratio=0.142, beta=9.0):
    """Kaiser-windowed sinc prototype (reference pqmf.py:15-48).  ``np.kaiser``
    equals ``scipy.signal.kaiser`` to 3e-17, so no SciPy dependency."""
    assert taps % 2 == 0, "The number of taps mush be even number."
    assert 0.0 < cutoff_ratio < 1.0, "Cutoff ratio must be > 0.0 and < 1.0."
    n = np.arange(taps + 1) - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        h = np.sin(np.pi * cutoff_ratio * n) / (np.pi * n)
    h[taps // 2] = cutoff_ratio
    return h * np.kaiser(taps + 1, beta)


def design_pqmf_filters(subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
    """Cosine-modulated analysis / synthesis banks, float64 [S, taps+1]
    (reference pqmf.py:76-88)."""
    proto = design_prototype_filter(taps, cutoff_ratio, beta)
    n = np.arange(taps + 1) - taps / 2
    ana = np.zeros((subbands, taps + 1))
    syn = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        phase = (2 * k + 1) * (np.pi / (2 * subbands)) * n
        sign = (-1) ** k * np.pi / 4
        ana[k] = 2 * proto * np.cos(phase + sign)
        syn[k] = 2 * proto * np.cos(phase - sign)
    return ana, syn


class PQMF(torch.nn.Module):
    """Drop-in for the reference ``PQMF`` (pqmf.py:51-135)."""

    def __init__(self, subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
        super().__init__()
        ana, syn = design_pqmf_filters(subbands, taps, cutoff_ratio, beta)
        self.register_buffer("analysis_filter", torch.from_numpy(ana).float().unsqueeze(1))
        self.register_buffer("synthesis_filter", torch.from_numpy(syn).float().unsqueeze(0))
        updown = torch.zeros((subbands, subbands, subbands)).float()
        for k in range(subbands):
            updown[k, k, 0] = 1.0
        self.register_buffer("updown_filter", updown)
        self.subbands = subbands
        self.taps = taps

    def synthesis(self, x):
        """x [B, subbands, T/subbands] -> [B, 1, T] (reference pqmf.py:121-135)."""
        x = x.contiguous().float()
        B, S, Tsub = x.shape
        assert S == self.subbands
        y = torch.empty((B, 1, S * Tsub), dtype=torch.float32, device=x.device)
        _native.pqmf_synthesis(x, self.synthesis_filter, y)
        return y

    def analysis(self, x):
        """x [B, 1, T] -> [B, subbands, T // subbands] (reference pqmf.py:108-119).  Used by the
        reference only for the multiband training loss; kept so the class is whole and for the
        analysis -> synthesis reconstruction check."""
        x = x.contiguous().float()
        if x.dim() != 3 or x.shape[1] != 1:
            raise _native.NativeError(f"PQMF.analysis expects [B, 1, T], got {tuple(x.shape)}")
        return _native.pqmf_analysis(x, self.analysis_filter)
