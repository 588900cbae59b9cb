"""``MODE=synthesize``: checkpoint + mel.npy -> wav files, on the MI355X engine.

Same surface as the reference's bin/synthesize.py: ``Synthesizer(checkpoint_path,
config_path, model_name)`` with ``.synthesize(mel[T,80]) -> (est, est - bias,
bias)`` and ``.test_rtf(mel)``; ``run_synthesizer()`` parses the same flags and
writes ``<wav>``, ``<wav[:-3]>remove.wav`` and ``<wav[:-3]>bias.wav``.  The
reference's fourth output (Griffin-Lim ``gl.wav``) needs librosa and is skipped
with a notice.
"""
import argparse
import os

import numpy as np
import torch
import yaml

from .. import hparams as hp
from ..audio import save_wav
from ..generator import (BasisMelGANGenerator, HiFiGANGenerator, MelGANGenerator,
                         MultiBandHiFiGANGenerator)

_HIFI_KEYS = ("resblock_kernel_sizes", "upsample_rates", "upsample_initial_channel",
              "resblock_type", "upsample_kernel_sizes", "resblock_dilation_sizes",
              "transposedconv", "bias")
_MELGAN_KEYS = ("in_channels", "out_channels", "kernel_size", "channels", "upsample_scales",
                "stack_kernel_size", "stacks", "use_weight_norm", "use_causal_conv")


def build_generator(model_name, config):
    """yaml dict -> generator, the reference's if/elif on ``model_name``
    (bin/synthesize.py:25-68); training-only yaml keys are ignored and a missing
    key raises KeyError like there."""
    if model_name == "melgan":
        return MelGANGenerator(**{k: config[k] for k in _MELGAN_KEYS})
    if model_name == "hifigan":
        return HiFiGANGenerator(**{k: config[k] for k in _HIFI_KEYS})
    if model_name == "multiband-hifigan":
        return MultiBandHiFiGANGenerator(**{k: config[k] for k in _HIFI_KEYS})
    if model_name == "basis-melgan":
        basis = torch.zeros(config["L"], config["out_channels"]).float()
        keys = ("L",) + _MELGAN_KEYS + ("transposedconv",)
        # ``lastlinear`` is a constructor option the reference's launcher never forwards
        # (basis_melgan.py:41); a yaml that names it gets it here
        extra = {"lastlinear": config["lastlinear"]} if "lastlinear" in config else {}
        return BasisMelGANGenerator(basis_signal_weight=basis, **{k: config[k] for k in keys}, **extra)
    raise Exception("no model find!")


def _numpy_globals():
    """What a pickled numpy array needs from the unpickler: published checkpoints carry a numpy ``'pattern'`` entry
    (bin/publish.py:71-75).  Allow-listing these keeps torch.load's restricted (weights_only) unpickler in charge."""
    try:
        from numpy._core.multiarray import _reconstruct
    except ImportError:                                   # numpy 1.x
        from numpy.core.multiarray import _reconstruct
    kinds = (np.float16, np.float32, np.float64, np.int8, np.int16, np.int32, np.int64, np.uint8, np.bool_)
    # torch's allow-list matches on the qualified NAME a pickle spells: numpy 1.x files (the reference's published
    # checkpoints) say numpy.core.multiarray._reconstruct, numpy 2.x files numpy._core.multiarray._reconstruct --
    # the same function; both spellings are registered whatever numpy runs here
    rebuild = [(_reconstruct, "numpy.core.multiarray._reconstruct"), (_reconstruct, "numpy._core.multiarray._reconstruct")]
    return [np.ndarray, np.dtype] + rebuild + sorted({type(np.dtype(k)) for k in kinds}, key=lambda t: t.__name__)


def load_checkpoint(path, device, unsafe=None):
    """torch.load with the restricted (weights_only) unpickler: tensors, plain containers and numpy arrays (allow-listed
    globals) -- everything a training or a published checkpoint of the reference holds.  Anything else in the file is
    refused; the unrestricted unpickler, which EXECUTES what a file tells it to, runs only on explicit request
    (``unsafe=True`` or FV_UNSAFE_LOAD=1 in the environment: a file the caller trusts)."""
    if unsafe is None:
        unsafe = os.environ.get("FV_UNSAFE_LOAD", "0") == "1"
    if unsafe:
        return torch.load(path, map_location=device, weights_only=False)
    with torch.serialization.safe_globals(_numpy_globals()):
        return torch.load(path, map_location=device, weights_only=True)


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("fastvocoder_amd needs a ROCm GPU (MI355X); no CPU inference path exists")
    return torch.device("cuda", torch.cuda.current_device())


class Synthesizer:
    def __init__(self, checkpoint_path, config_path, model_name, device=None) -> None:
        self.device = device if device is not None else default_device()
        self.model = self.load_model(checkpoint_path, config_path, model_name)

    def load_model(self, checkpoint_path, config_path, model_name):
        with open(config_path) as f:
            config = yaml.load(f, Loader=yaml.Loader)
        print(f"Loading Model of {model_name}...")
        model = build_generator(model_name, config).to(self.device)
        ckpt = load_checkpoint(checkpoint_path, self.device)
        model.load_state_dict(ckpt["model"])
        model.eval()
        model.remove_weight_norm()
        self.checkpoint = ckpt
        self.config = config
        return model

    _ZERO_CACHE_ENTRIES = 8     # ~1 MB of device memory per 1000 frames each

    def zero_mel_response(self, frames):
        """The generator's output for an all-zero mel of ``frames`` frames.  It depends only on
        (weights, frames): computed once per (weights version, length) and kept in a small LRU
        (the reference recomputes it for every utterance, bin/synthesize.py:76-77)."""
        cache = self.__dict__.setdefault("_zero_cache", {})
        key = (self.model._fv_state(), int(frames))
        if key in cache:
            cache[key] = cache.pop(key)                      # most recently used last
        else:
            while len(cache) >= self._ZERO_CACHE_ENTRIES:
                cache.pop(next(iter(cache)))
            with torch.no_grad():
                cache[key] = self.model.inference(
                    torch.zeros(frames, self.config.get("in_channels", 80))).clone()
        return cache[key]

    def synthesize(self, mel):
        """mel [T,80] ndarray -> (est_source, est_source - bias, bias), each 1-D fp32 on the device;
        ``bias`` is the generator's response to an all-zero mel (cached) and the difference is formed in
        the last kernel's epilogue: one generator pass per utterance instead of the reference's two."""
        with torch.no_grad():
            bias = self.zero_mel_response(int(np.asarray(mel).shape[0]))
            est_source, est_source_remove_bias = self.model.inference_minus(mel, bias)
        return est_source, est_source_remove_bias, bias

    def test_rtf(self, mel):
        with torch.no_grad():
            self.model.inference(mel)


def run_synthesizer():
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint_path", type=str)
    parser.add_argument("--mel_path", type=str)
    parser.add_argument("--wav_path", type=str)
    parser.add_argument("--model_name", type=str,
                        help="melgan, hifigan, multiband-hifigan and basis-melgan.")
    parser.add_argument("--config", type=str, help="path to model configuration file")
    args = parser.parse_args()

    synthesizer = Synthesizer(args.checkpoint_path, args.config, args.model_name)
    mel = np.load(args.mel_path)
    outs = synthesizer.synthesize(mel.T)
    est_source, est_source_remove_bias, bias = (o.cpu().numpy() for o in outs)
    stem = args.wav_path[:-3]
    save_wav(est_source, args.wav_path, hp.sample_rate, rescale_out=hp.rescale_out)
    save_wav(est_source_remove_bias, stem + "remove.wav", hp.sample_rate, rescale_out=hp.rescale_out)
    save_wav(bias, stem + "bias.wav", hp.sample_rate, rescale_out=hp.rescale_out)
    print(f"[fastvocoder_amd] skipped {stem}gl.wav: Griffin-Lim needs librosa (out of scope)")
