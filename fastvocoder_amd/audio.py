"""The wav sink of the synthesize flow (reference data/audio.py:12-26).  Mel
extraction and Griffin-Lim of the reference are librosa/TensorFlow bound and
out of scope."""
import numpy as np
import scipy.io.wavfile


def encode_16bits(x, rescale_out=1.0):
    """Peak-normalise to int16 full scale times ``rescale_out``.  Like the
    reference this scales ``x`` IN PLACE (callers see the mutation)."""
    x *= 32767 / max(0.01, np.max(np.abs(x))) * rescale_out
    return x.astype(np.int16)


def save_wav(y, filename, sample_rate, rescale_out=1.0):
    y = encode_16bits(y, rescale_out)
    scipy.io.wavfile.write(filename, sample_rate, y.astype(np.int16))
