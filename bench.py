"""Headline benchmark: HiFi-GAN-light generator inference on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): conf/hifigan/light.yaml, batch = 1 utterance
per GPU, mel 80 x 1000 frames of synthetic U[0,1) data already resident in HBM,
seeded gain-calibrated random-init weights (no checkpoint exists offline),
weight norm folded.  One "step" = one ``Generator.forward`` over the per-GPU
batch -> 240 000 samples per utterance.  N > 1 (launched by torch.distributed.run,
one rank per GPU): weights are built on rank 0 and broadcast over RCCL once,
every rank then runs its own utterances (weak scaling: the path has no exchange
step, so the timed steps contain no collective); one root gather after the timed
region checks the RCCL data path, and ``--gather`` puts an asynchronous per-step
gather inside the timed region instead.

Prints ONE JSON line: value = whole-job audio samples / second, plus RTF at
22.05 kHz and 24 kHz, the roofline of the dominant kernel (fp32-MFMA implicit-GEMM
conv, per-launch HIP-event timing on the launch stream), and the reference's CPU
path (its ATen op sequence, oracle/torch_port.py) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fastvocoder_amd import _native  # noqa: E402
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

MODEL, CONF = "hifigan", "conf/hifigan/light.yaml"
T_FRAMES = 1000
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0


def baseline_metric():
    """BASELINE.json's metric string, verbatim (value = the samples/sec part; RTF rides alongside)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "audio samples/sec + RTF @22.05kHz, HiFi-GAN-light, 1/2/4/8 MI355X"


def load_conf():
    import yaml
    with open(os.path.join(ROOT, CONF)) as f:
        return yaml.safe_load(f)


def cpu_baseline(cfg, sd, mel):
    """The reference's CPU path (un-fused ATen op sequence) on this box's cores:
    one warm-up + best of 3 passes over ONE utterance of the same workload."""
    from oracle import torch_port  # the only oracle use in bench: the CPU baseline leg
    folded = torch_port.fold_state_dict(sd)
    threads = torch.get_num_threads()
    torch_port.inference(MODEL, mel, folded, cfg)
    best = float("inf")
    n = 0
    for _ in range(3):
        t0 = time.perf_counter()
        y = torch_port.inference(MODEL, mel, folded, cfg)
        best = min(best, time.perf_counter() - t0)
        n = int(y.numel())
    return {"value": n / best, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"1 utterance, mel 80x{mel.shape[0]} -> {n} samples, best of 3 after 1 warm-up, "
                      f"ATen port of the reference generator, {threads} threads",
            "seconds": best}


def timed_steps(step, steps, warmup, dist, dev, after=None):
    """``warmup`` untimed calls of ``step()``, then EXACTLY ``steps`` timed ones bracketed by
    (device sync, barrier, device sync) on both sides; returns (seconds = MAX over ranks,
    last step's result).  ``after`` (optional) runs after the last warm-up and after the last
    timed step, inside the bracket.  ``dist`` is torch.distributed or None (single process);
    covered on CPU by tests/test_distributed_cpu.py with the gloo backend."""
    def sync():
        # drain this GPU, meet the other ranks, drain the barrier's own collective
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if dev.type == "cuda":
                torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    if after is not None:
        after()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    if after is not None:
        after()             # e.g. complete asynchronous transfers started by the last step
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather every step's waveforms to rank 0 inside the timed region "
                         "(asynchronously, overlapping the next step); default: utterances stay on "
                         "the rank that made them and one gather after the timed region checks the path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                 "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # FV_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL init, broadcast, gather, barrier,
    # all-reduce) with a single rank too -- a self-test of that path on a 1-GPU box
    force_dist = os.environ.get("FV_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = load_conf()
    from fastvocoder_amd import parallel
    model = build_generator(MODEL, cfg)
    sd = seeded_state_dict(MODEL, cfg, seed=0) if rank == 0 else None
    if rank == 0:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    if dist is not None:
        parallel.broadcast_weights(model, src=0)          # RCCL broadcast, once
    model.remove_weight_norm()

    B = args.batch
    mel = torch.from_numpy(seeded_mel(T_FRAMES, seed=100 + rank, batch=B)).to(dev)
    gather = parallel.WaveformGather(world, rank, dev) if dist is not None else None

    def step():
        with torch.no_grad():
            wav = model(mel)
        if gather is not None and args.gather:
            gather(wav)           # asynchronous: overlaps the next step's forward
        return wav

    def last_step_done():
        if gather is not None and args.gather:
            gather.flush()        # the last gather belongs to the timed region

    # model load, not a step: the first call folds weight norm and packs every layer's weights on
    # the GPU (the plan), which later calls replay -- done here so that even --warmup 0 times steps only
    with torch.no_grad():
        model(mel)
    torch.cuda.synchronize()
    elapsed, wav = timed_steps(step, args.steps, args.warmup, dist, dev, after=last_step_done)

    if gather is not None and not args.gather:
        # outside the timed region: one root gather, so that the RCCL data path is exercised and
        # rank 0 ends up holding every rank's last waveforms, as a serving front-end would
        gather(wav)
        bufs = gather.flush()
        if rank == 0:
            assert len(bufs) == world and all(tuple(b.shape) == tuple(wav.shape) for b in bufs)
            assert torch.equal(bufs[0], wav)
    samples_per_utt = int(wav.shape[-1])
    total_samples = samples_per_utt * B * world * args.steps
    value = total_samples / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    out = None
    if rank == 0:
        # per-launch timing of the dominant kernel family with HIP events on the launch stream.
        # A per-launch duration is only well defined without stream overlap, so this leg pins
        # the replay of the same forward to ONE stream (FV_SINGLE_LANE, read by fv_plan_run at
        # every call; the default grouped plan already is single-stream, FV_MRF=lanes is not).
        os.environ["FV_SINGLE_LANE"] = "1"
        for _ in range(2):
            with torch.no_grad():
                model(mel)
        torch.cuda.synchronize()
        _native.profile_enable(True)
        reps = 5
        for _ in range(reps):
            with torch.no_grad():
                model(mel)
        torch.cuda.synchronize()
        _native.profile_enable(False)
        del os.environ["FV_SINGLE_LANE"]
        p32 = _native.profile_collect(_native.KERNEL_CONV_MFMA32)
        p16 = _native.profile_collect(_native.KERNEL_CONV_MFMA16)
        q16 = _native.profile_collect(_native.KERNEL_PAIR16)
        q32 = _native.profile_collect(_native.KERNEL_PAIR32)
        for k in ("launches", "ms", "flops", "bytes"):
            p16[k] += q16[k]
            p32[k] += q32[k]
        pn = _native.profile_collect(_native.KERNEL_CONV_NARROW)
        mf_launch, mf_ms, mf_flops = (p32["launches"] + p16["launches"], p32["ms"] + p16["ms"],
                                      p32["flops"] + p16["flops"])
        achieved = mf_flops / (mf_ms * 1e-3) / 1e12 if mf_ms > 0 else 0.0
        # HBM bytes per launch of this kernel family from the committed rocprofv3 PMC
        # passes of this same command (tools/pmc_traffic.py: (2*FETCH_SIZE + WRITE_SIZE)*1024)
        traffic, traffic_src = None, None
        for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles"))
                            if f.endswith("_hbm_traffic.json")), reverse=True) \
                if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                traffic = json.load(f)["conv_mfma_family"]["hbm_bytes_per_launch"]
            traffic_src = "profiles/" + cand
            break
        roofline = {
            "kernel": "fv::conv_mfma_kernel / fv::conv_group3_kernel / fv::conv_sum3_kernel (fp32-MFMA "
                      "implicit-GEMM conv1d, csrc/conv_kernels.hpp: every Conv1d / ConvTranspose1d layer with "
                      "Cout > 4 = 77 of the 78 convs of a forward; the 3 ResBlock convs of an MRF position "
                      "share one launch)",
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC, offline pass)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": (p32["bytes"] + p16["bytes"]) / max(mf_launch, 1),
            "measured": "per-launch HIP events, single-stream replay of the same forward",
            # the timed (multi-stream) step as a whole: algorithmic FLOP of one forward / step time
            "achieved_whole_step": mf_flops / reps / (ms_per_step * 1e-3) / 1e12,
            "frac_whole_step": mf_flops / reps / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "launches_per_step": mf_launch // reps,
            "avg_launch_us": 1e3 * mf_ms / max(mf_launch, 1),
            "algorithmic_gflop_per_step": mf_flops / reps / 1e9,
            "kernel_ms_per_step": mf_ms / reps,
            "hbm_GBps_algorithmic": (p32["bytes"] + p16["bytes"]) / (mf_ms * 1e-3) / 1e9 if mf_ms > 0 else 0.0,
            "narrow_conv_ms_per_step": pn["ms"] / reps,
        }
        # The HBM-bound members of the same family, which the north star names ("memory roofline on
        # the dilated-conv kernels"): the C = 16 stage (12-44 FLOP/B, below the 20 FLOP/B ridge for the
        # 3-tap layers) runs in the 16x16x4-MFMA instantiations -- algorithmic bytes / their HIP-event time
        roofline_hbm = {
            "kernel": "the C = 16 stage of the generator (5 grouped + 1 merged launch of dilated / plain "
                      "16-channel convs, 240 000 samples each): 16x16x4-MFMA instantiations of the same conv body",
            "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
            "achieved": p16["bytes"] / (p16["ms"] * 1e-3) / 1e9 if p16["ms"] > 0 else 0.0,
            "frac": p16["bytes"] / (p16["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS if p16["ms"] > 0 else 0.0,
            "launches_per_step": p16["launches"] // reps,
            "tflops": p16["flops"] / (p16["ms"] * 1e-3) / 1e12 if p16["ms"] > 0 else 0.0,
            "measured": "algorithmic bytes (each tensor once) / per-launch HIP events; traffic by PMC: profiles/",
        }
        dur22, dur24 = total_samples / 22050.0, total_samples / 24000.0
        out = {
            "metric": baseline_metric(), "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rtf_22k05": elapsed / dur22, "rtf_24k": elapsed / dur24,
            "config": {"workload": "HiFi-GAN light (conf/hifigan/light.yaml) generator forward, "
                                   f"mel 80x{T_FRAMES}, batch {B} utterance(s) per GPU, "
                                   f"{samples_per_utt} samples each; BASELINE.json configs[1]",
                       "global_batch": B * world, "frames": T_FRAMES,
                       "parallelism": f"utterance-sharded x{world}" if world > 1 else "single GPU",
                       "convs_per_forward": model._trunk_plan(T_FRAMES).num_ops()},
            "roofline": roofline,
            "roofline_hbm_stage": roofline_hbm,
        }
        # PCIe-inclusive rate of the drop-in boundary (never `value`): Generator.inference takes a
        # HOST mel [T,80] and the caller wants a HOST waveform -- pageable numpy in, numpy out,
        # one utterance per call, fully synchronous (H2D + forward + D2H per call)
        if world == 1:
            mel_np = seeded_mel(T_FRAMES, seed=100)
            for _ in range(3):
                model.inference(mel_np).cpu()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps_io = 20
            for _ in range(reps_io):
                y_host = model.inference(mel_np).cpu().numpy()
            dt = (time.perf_counter() - t0) / reps_io
            out["host_to_host"] = {"ms_per_utterance": 1e3 * dt, "samples_per_s": y_host.size / dt,
                                   "what": "Generator.inference(numpy mel) -> numpy waveform, per call: "
                                           "H2D 320 KB + forward + D2H 960 KB, pageable memory, synchronous"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, seeded_mel(T_FRAMES, seed=100))
            out["cpu_baseline"]["rtf_22k05"] = out["cpu_baseline"]["seconds"] / (samples_per_utt / 22050.0)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
