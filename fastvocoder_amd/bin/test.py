"""``MODE=test``: the reference's RTF benchmark over a directory of mels
(bin/test.py:98-132), on the MI355X engine.  Same flags, same printed lines
(``duration is``, ``cost time:``, ``rtf is``) and the same formula
``rtf = cost / (10 * duration)``; unlike the reference the loop is bracketed by
device synchronisation (and preceded by one warm-up pass) so the wall time is
the GPU's, not the launch queue's.  Basis-MelGAN synthesis with the published
``pattern`` (bin/test.py:82-91) is reproduced too.
"""
import argparse
import os
import time

import numpy as np
import torch

from .. import hparams as hp
from ..audio import save_wav
from .synthesize import Synthesizer as _BaseSynthesizer

USE_PATTERN = True
TEST_RTF = True


class Synthesizer(_BaseSynthesizer):
    def load_model(self, checkpoint_path, config_path, model_name):
        model = super().load_model(checkpoint_path, config_path, model_name)
        if model_name == "basis-melgan":
            self.L = self.config["L"]
            pattern = self.checkpoint.get("pattern") if USE_PATTERN else None
            self.pattern = None if pattern is None else \
                torch.as_tensor(np.asarray(pattern), dtype=torch.float32, device=self.device)
        return model

    def synthesize(self, mel):
        """Basis-MelGAN only (like the reference): drop the trailing L/2 samples
        and subtract the stored zero-mel pattern (or a fresh zero-mel pass)."""
        with torch.no_grad():
            frames = int(np.asarray(mel).shape[0])
            n = self.model._minus_plan(frames).output_shape(frames)[1]      # (F-1)*L/2 + L samples
            keep = n - self.L // 2
            if getattr(self, "pattern", None) is not None:
                # the stored zero-mel pattern, padded to the generator's output length (the tail is dropped)
                bias = torch.zeros(n, dtype=torch.float32, device=self.pattern.device)
                bias[:keep] = self.pattern[:keep]
            else:
                bias = self.model.inference(torch.zeros(frames, np.asarray(mel).shape[1]))
            _, removed = self.model.inference_minus(mel, bias)              # one pass; difference in the epilogue
        return removed[:keep]


def run_test():
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint_path", type=str)
    parser.add_argument("--file_path", type=str)
    parser.add_argument("--model_name", type=str,
                        help="melgan, hifigan, multiband-hifigan and basis-melgan.")
    parser.add_argument("--config", type=str, help="path to model configuration file")
    args = parser.parse_args()

    synthesizer = Synthesizer(args.checkpoint_path, args.config, args.model_name)
    mels, names, duration = [], [], 0.0
    for file in sorted(os.listdir(args.file_path)):
        if not file.endswith(".npy"):
            continue
        mel = np.load(os.path.join(args.file_path, file))
        if mel.shape[0] == hp.num_mels:
            mel = mel.T
        mels.append(mel)
        names.append(file)
        duration += (mel.shape[0] * hp.hop_size) / hp.sample_rate
    print(f"duration is {duration}s.")

    if args.model_name == "basis-melgan":
        for mel, filename in zip(mels, names):
            est_source = synthesizer.synthesize(mel)
            # device tensor -> int16 on the GPU (fv_encode_16bits), then 2 B/sample over PCIe
            save_wav(est_source.contiguous(), os.path.join(args.file_path, f"{filename}.wav"),
                     sample_rate=hp.sample_rate)

    if TEST_RTF:
        for mel in mels:                      # warm-up: plan build + first launches
            synthesizer.test_rtf(mel)
        torch.cuda.synchronize()
        s = time.perf_counter()
        for _ in range(10):
            for mel in mels:
                synthesizer.test_rtf(mel)
        torch.cuda.synchronize()
        cost = time.perf_counter() - s
        print(f"cost time: {cost}s.")
        rtf = cost / (10.0 * duration)
        print(f"rtf is {rtf}.")
