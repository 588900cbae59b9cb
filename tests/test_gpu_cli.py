"""GPU tests of the drop-in command-line surface (SURVEY.md section 8 b / a-14 / a-15 and
the section 8 f flows): MODE=synthesize / test / publish through bin/launcher.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.io.wavfile
import torch

from fastvocoder_amd.synthetic import seeded_state_dict
from oracle import torch_port
from tests import cases

pytestmark = pytest.mark.gpu


def _ckpt(tmp_path, name, path, extra=None):
    cfg = cases.load_conf(path)
    sd = seeded_state_dict(name, cfg, seed=0)
    ck = str(tmp_path / f"{name}.pth.tar")
    obj = {"model": {k: torch.from_numpy(v) for k, v in sd.items()}}
    obj.update(extra or {})
    torch.save(obj, ck)
    return ck, cfg, sd


def _run(mode, *args):
    env = dict(os.environ, MODE=mode)
    r = subprocess.run([sys.executable, os.path.join(cases.ROOT, "bin", "launcher.py"), *args],
                       env=env, cwd=cases.ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_mode_synthesize_writes_the_three_wavs(tmp_path):
    ck, cfg, sd = _ckpt(tmp_path, "hifigan", "conf/hifigan/light.yaml")
    mel = np.random.RandomState(3).rand(80, 120)           # [80, T] float64, like resource/test.mel.npy
    np.save(tmp_path / "in.npy", mel)
    wav = str(tmp_path / "out.wav")
    out = _run("synthesize", "--checkpoint_path", ck, "--mel_path", str(tmp_path / "in.npy"),
               "--wav_path", wav, "--model_name", "hifigan",
               "--config", os.path.join(cases.ROOT, "conf/hifigan/light.yaml"))
    assert "Loading Model of hifigan" in out
    ref = torch_port.inference("hifigan", mel.T, torch_port.fold_state_dict(sd), cfg).numpy()
    zero = torch_port.inference("hifigan", np.zeros_like(mel.T), torch_port.fold_state_dict(sd), cfg).numpy()
    for suffix, expect in (("out.wav", ref), ("out.remove.wav", ref - zero), ("out.bias.wav", zero)):
        sr, data = scipy.io.wavfile.read(str(tmp_path / suffix))
        assert sr == 24000 and data.dtype == np.int16 and data.shape == (120 * 240,)
        x = expect.astype(np.float32).copy()
        x *= 32767 / max(0.01, np.max(np.abs(x))) * 0.4      # encode_16bits, data/audio.py:12-14
        assert np.abs(data.astype(np.int32) - x.astype(np.int16)).max() <= 2
        assert abs(int(np.abs(data).max()) - 13106) <= 2      # peak = 0.4 * 32767


def test_mode_test_prints_the_rtf_lines(tmp_path):
    ck, _, _ = _ckpt(tmp_path, "melgan", "conf/melgan/original.yaml")
    d = tmp_path / "mels"
    d.mkdir()
    for i in range(2):
        np.save(d / f"m{i}.npy", np.random.RandomState(i).rand(80, 100 + 20 * i))
    out = _run("test", "--checkpoint_path", ck, "--file_path", str(d), "--model_name", "melgan",
               "--config", os.path.join(cases.ROOT, "conf/melgan/original.yaml"))
    assert "duration is 2.2s." in out                         # (100 + 120) * 240 / 24000
    assert "cost time:" in out and "rtf is" in out
    rtf = float(out.split("rtf is")[1].strip().rstrip("."))
    assert 0 < rtf < 0.05


def test_publish_pattern_and_basis_test_flow(tmp_path):
    """bin/publish.py:67-75 + bin/test.py:82-91: the stored zero-mel pattern replaces the
    second generator pass of Basis-MelGAN synthesis."""
    from fastvocoder_amd.bin.publish import publish_model
    from fastvocoder_amd.bin.test import Synthesizer
    ck, cfg, sd = _ckpt(tmp_path, "basis-melgan", "conf/basis-melgan/light.yaml")
    conf = os.path.join(cases.ROOT, "conf/basis-melgan/light.yaml")
    pub = str(tmp_path / "pub.pth.tar")
    out = publish_model(ck, conf, "basis-melgan", pub, pattern_frames=300)
    assert out["pattern"].shape == (300 * 240 + 15,)
    syn = Synthesizer(pub, conf, "basis-melgan")
    mel = np.random.RandomState(5).rand(200, 80).astype(np.float32)
    y = syn.synthesize(mel).cpu().numpy()
    folded = torch_port.fold_state_dict(sd)
    ref = torch_port.inference("basis-melgan", mel, folded, cfg).numpy()[:-15]
    zero = torch_port.inference("basis-melgan", np.zeros((300, 80), np.float32), folded, cfg).numpy()
    assert y.shape == (200 * 240,)
    assert np.abs(y - (ref - zero[: ref.shape[0]])).max() <= 2e-4


def test_zero_mel_response_is_cached(tmp_path):
    from fastvocoder_amd.bin.synthesize import Synthesizer
    ck, cfg, sd = _ckpt(tmp_path, "hifigan", "conf/hifigan/light.yaml")
    syn = Synthesizer(ck, os.path.join(cases.ROOT, "conf/hifigan/light.yaml"), "hifigan")
    mel = np.random.RandomState(1).rand(50, 80)
    a = syn.synthesize(mel)
    b = syn.synthesize(mel)
    assert a[2].data_ptr() == b[2].data_ptr()                 # bias served from the cache
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
