"""``MODE=publish``: re-save a checkpoint for deployment (reference
bin/publish.py:18-78).  For Basis-MelGAN the zero-mel response ("pattern", the
waveform of 30000 all-zero frames) is computed once on the GPU and stored beside
``model`` so test-time synthesis subtracts it instead of running a second pass.
"""
import argparse

import numpy as np
import torch

from .synthesize import Synthesizer

PATTERN_FRAMES = 30000  # reference bin/publish.py:69


def publish_model(checkpoint_path, config_path, model_name, save_path, pattern_frames=PATTERN_FRAMES):
    """Like the reference (bin/publish.py:66-75) this writes ``save_path`` for Basis-MelGAN only --
    {'model': the checkpoint's state dict (weight-norm keys kept), 'pattern': zero-mel waveform} -- and is a
    load check for the other models.  Returns the published dict (None when nothing was written)."""
    syn = Synthesizer(checkpoint_path, config_path, model_name)
    if model_name != "basis-melgan":
        print(f"[fastvocoder_amd] {model_name}: checkpoint loads; only basis-melgan is re-saved (with its pattern)")
        return None
    with torch.no_grad():
        zero = np.zeros((pattern_frames, syn.config["in_channels"]), dtype=np.float32)
        pattern = syn.model.inference(zero).cpu().numpy()
    # the state dict as loaded: equal, key for key, to the reference's model.state_dict() before remove_weight_norm
    out = {"model": syn.checkpoint["model"], "pattern": pattern}
    torch.save(out, save_path)
    return out


def run_publisher():
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint_path", type=str)
    parser.add_argument("--save_path", type=str)
    parser.add_argument("--model_name", type=str)
    parser.add_argument("--config", type=str, help="path to model configuration file")
    args = parser.parse_args()
    publish_model(args.checkpoint_path, args.config, args.model_name, args.save_path)
