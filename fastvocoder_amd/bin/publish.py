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
    syn = Synthesizer(checkpoint_path, config_path, model_name)
    out = {"model": syn.checkpoint["model"]}
    if model_name == "basis-melgan":
        with torch.no_grad():
            zero = np.zeros((pattern_frames, syn.config["in_channels"]), dtype=np.float32)
            out["pattern"] = syn.model.inference(zero).cpu().numpy()
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
