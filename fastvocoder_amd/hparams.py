"""Inference-side constants with the reference's names and values
(/root/reference/hparams.py:4-15).  Only what the generator path and its
callers read; the training knobs of the reference are out of scope."""
num_mels = 80
hop_size = 240
sample_rate = 24000
rescale_out = 0.4
