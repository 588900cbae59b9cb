"""The wav sink of the synthesize flow (reference data/audio.py:12-26).  Mel
extraction and Griffin-Lim of the reference are librosa/TensorFlow bound and
out of scope.

``encode_16bits`` / ``save_wav`` take what the reference's take (a float numpy
array, scaled IN PLACE) and, additionally, a float32 tensor on the ROCm device:
then the peak reduction, scaling and int16 conversion run on the GPU
(csrc/wav_sink.hip, fv_encode_16bits) and only the int16 samples are copied to
the host -- half the PCIe bytes of the fp32 waveform.
Both routes give the same int16 samples bit for bit.
"""
import numpy as np
import scipy.io.wavfile
import torch

from . import _native


def encode_16bits(x, rescale_out=1.0):
    """Peak-normalise to int16 full scale times ``rescale_out``.  Like the
    reference this scales ``x`` IN PLACE (callers see the mutation).  A device
    tensor ([n] or [B,n], normalised per row) returns a device int16 tensor."""
    if torch.is_tensor(x):
        if not x.is_cuda:
            raise _native.NativeError("encode_16bits: a tensor argument must live on the ROCm device; "
                                      "pass a numpy array for the host route")
        return _native.encode_16bits(x, rescale_out, scale_in_place=True)[0]
    x *= 32767 / max(0.01, np.max(np.abs(x))) * rescale_out
    return x.astype(np.int16)


def save_wav(y, filename, sample_rate, rescale_out=1.0):
    y = encode_16bits(y, rescale_out)
    if torch.is_tensor(y):
        y = y.cpu().numpy()
    scipy.io.wavfile.write(filename, sample_rate, y.astype(np.int16))
