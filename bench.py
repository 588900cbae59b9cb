"""Headline benchmark: HiFi-GAN-light generator inference on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config light|large512]

Default workload (BASELINE.json configs[1]): conf/hifigan/light.yaml, batch = 1 utterance per GPU,
mel 80 x 1000 frames of synthetic U[0,1) data already resident in HBM, seeded gain-calibrated
random-init weights (no checkpoint exists offline), weight norm folded.  One "step" = one
``Generator.forward`` over the per-GPU batch -> 240 000 samples per utterance.  Utterance 0 is the
mel the committed reference golden was made from, and the LAST timed output is checked against that
golden (<= 1e-4, the north star's bound) before anything is printed: the timed forward is the
parity-checked forward.

N > 1: one rank per GPU.  `python bench.py --gpus N` started as ONE process replaces itself with
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
(refused when the node shows fewer than N devices); started by that launcher (WORLD_SIZE in the environment) it is
a rank.  Weights are built on rank 0 and broadcast over RCCL once; every rank runs its own utterances (weak scaling: the path has no exchange step) and
every step's waveforms are gathered to rank 0 INSIDE the timed region (asynchronously, overlapping the
next step's forward); the same steps without the gather are timed too and reported beside it.

The fixed job (BASELINE.json configs[4]): HiFi-GAN large, 512 utterances of 80 x 1000 per step -- strong
scaling: rank 0 holds the mels, every rank synthesises its contiguous block in sub-batches and encodes them
to int16 on the GPU (fv_encode_16bits), rank 0 gathers the int16 waveforms; the scatter of sub-batch i + 1 and the
gather of sub-batch i - 1 are in flight under sub-batch i's forward (parallel.synthesize_pipelined); scatter, forward,
encode and gather are all inside the timed step (the same step without scatter / gather is timed beside it), and rank 0
checks gathered rows against its own single-utterance runs bit for bit.  ``--config large512`` makes it the
headline of the line; the default run appends it as ``strong_scaling_job`` (one timed step; ``--no-job``
skips it), so that one invocation per GPU count gives both scaling curves.

Sustained rate: every leg runs PREWARM_S (0.75) seconds of untimed forwards in front of its W warm-up steps -- an idle MI355X
ramps for ~30 ms, boosts for ~0.5 s and then settles (profiles/r06_clock_ramp.txt); the default W = 5 is 3 ms -- then W warm-up
and EXACTLY K timed steps; the same K steps without it are timed first and reported as ``cold_start``
(FV_BENCH_PREWARM_S=0 switches the prewarm off).

Prints ONE JSON line: value = whole-job audio samples / second, plus RTF at 22.05 kHz and 24 kHz, the
roofline of the dominant kernel family (fp32-MFMA convs, per-launch HIP events on the launch stream),
the C = 16 stage against the memory roofline, and the reference's CPU path (its ATen op sequence,
oracle/torch_port.py) timed on this host's cores.
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

MODEL = "hifigan"
CONFS = {"light": "conf/hifigan/light.yaml", "large512": "conf/hifigan/large.yaml"}
T_FRAMES = 1000
JOB_UTTERANCES = int(os.environ.get("FV_BENCH_JOB", "512"))   # --config large512 (the env override is for tests)
# the arithmetic the path computes in: fp32 tensors, fp32 accumulation; on the ResBlock stages every fp32 product is
# formed from split-f16 operand pairs on the f16 matrix cores (fp32-class accuracy); the
# exact-fp32 figure of the same forward rides in the line as `exact_fp32`
DTYPE = "f32 (products as split-f16 pairs, fp32 accumulate)"
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense f16 matrix peak (same guide; the sparse headline figure is twice that)
PEAK_HBM_GBS = 8000.0
TOL = 1e-4                      # north star: outputs match the reference generator within 1e-4 fp32 max-abs


def baseline_metric():
    """BASELINE.json's metric string, verbatim (value = the samples/sec part; RTF rides alongside)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "audio samples/sec + RTF @22.05kHz, HiFi-GAN-light, 1/2/4/8 MI355X"


def load_conf(config):
    import yaml
    with open(os.path.join(ROOT, CONFS[config])) as f:
        return yaml.safe_load(f)


def utterance_mels(first, count):
    """[count, 80, T] forward-layout mels of the global utterances first .. first+count-1; utterance i is
    ``seeded_mel(T, seed=1+i)``: utterance 0 is the mel of tests/golden/full_hifigan_light.npz (seed 1)."""
    # (C-contiguous: np.stack keeps the transposed views' memory order, and a strided mel would make every forward start with
    # a copy kernel inside the timed region -- the workload is "mel resident in HBM", in the layout forward() takes)
    return np.ascontiguousarray(np.stack([seeded_mel(T_FRAMES, seed=1 + first + i).T for i in range(count)]), dtype=np.float32)


def check_against_golden(wav_row):
    """The generator output for utterance 0 against the reference's own output (strided samples of
    tests/golden/full_hifigan_light.npz, written by tests/golden/make_golden.py from /root/reference)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_hifigan_light.npz"))
    y = wav_row.double().cpu().numpy().reshape(-1)
    assert y.size == int(g["T1000_n"]), (y.size, int(g["T1000_n"]))
    err = float(np.abs(y[g["T1000_idx"]] - g["T1000_samples"]).max())
    assert err <= TOL, f"timed output differs from the reference golden by {err:.3e} > {TOL}"
    return err


def host_cpu():
    """(model string, physical cores, logical CPUs) of this host, from /proc/cpuinfo (the CPU
    baseline states what it ran on)."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    logical += 1
                elif key == "model name" and model == "unknown":
                    model = val
                elif key == "physical id":
                    pid = val
                elif key == "core id":
                    cid = val
                    phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or logical or (os.cpu_count() or 0)), (logical or (os.cpu_count() or 0))


def cpu_baseline(cfg, sd, mel):
    """The reference's CPU path (un-fused ATen op sequence) on this box's cores: one warm-up + best of 3 passes over
    ONE utterance of the same workload with torch's default thread count, and -- SURVEY.md section 8(d) -- one pass on
    ONE thread over the first quarter of the same utterance (bounded: the whole leg stays within ~20 s)."""
    from oracle import torch_port  # the only oracle use in bench: the CPU baseline leg
    folded = torch_port.fold_state_dict(sd)
    threads = torch.get_num_threads()
    cpu_model, physical, logical = host_cpu()
    torch_port.inference(MODEL, mel, folded, cfg)
    best = float("inf")
    n = 0
    for _ in range(3):
        t0 = time.perf_counter()
        y = torch_port.inference(MODEL, mel, folded, cfg)
        best = min(best, time.perf_counter() - t0)
        n = int(y.numel())
    short = mel[:max(mel.shape[0] // 4, 1)]
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        y1 = torch_port.inference(MODEL, short, folded, cfg)
        one = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    n1 = int(y1.numel())
    return {"value": n / best, "unit": "samples/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "physical_cores": physical, "logical_cpus": logical,
            "sample": f"1 utterance, mel 80x{mel.shape[0]} -> {n} samples, best of 3 after 1 warm-up, "
                      f"ATen port of the reference generator, {threads} threads ({physical} physical cores, {cpu_model})",
            "seconds": best,
            "one_thread": {"value": n1 / one, "unit": "samples/s", "cores": 1, "seconds": one,
                           "rtf_22k05": one / (n1 / 22050.0),
                           "sample": f"the first {short.shape[0]} frames of the same mel -> {n1} samples, ONE pass on one "
                                     "thread (torch.set_num_threads(1)), no warm-up"}}


# The other single-GPU BASELINE.json configs (parity-test cases; the headline stays configs[1]): label, model name, yaml,
# batch, frames, golden fixture, timed steps.  Utterance 0 of every batch is the mel its golden was made from.
# The timed steps are pipelined (stream-ordered) forwards: the range guard is the explicit opt-in "lazy" -- the guard word
# is read by check_range() AFTER the steps (asserted clean) instead of by a stream drain inside every step.  The default
# policy ("auto" = checked before every call returns) is timed beside it: range_guard.ms_per_step_sync_checked.
BENCH_RANGE_GUARD = "lazy"

OTHER_CONFIGS = [
    ("config1_melgan_T200_B1", "melgan", "conf/melgan/original.yaml", 1, 200, "synthesize_melgan.npz", 50),
    ("config3_mb_hifigan_light_pqmf_B32", "multiband-hifigan", "conf/multiband-hifigan/light.yaml", 32, 1000, "full_mb_light.npz", 4),
    ("config4_basis_melgan_light_B64", "basis-melgan", "conf/basis-melgan/light.yaml", 64, 1000, "full_basis.npz", 4),
]


def other_configs(dev):
    """BASELINE.json configs 1, 3 and 4 on this GPU, driver-timed in the same line as the headline: ms/step, samples/s,
    RTF, algorithmic TFLOP/s -- each with utterance 0 of its LAST timed output checked against the reference's golden."""
    import yaml
    out = {}
    for label, name, path, B, T, golden, steps in OTHER_CONFIGS:
        with open(os.path.join(ROOT, path)) as f:
            cfg = yaml.safe_load(f)
        m = build_generator(name, cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(name, cfg, seed=0).items()})
        m = m.to(dev).eval()
        m.remove_weight_norm()
        m.range_guard = BENCH_RANGE_GUARD                 # checked after the timed steps (m.check_range() below)
        g = np.load(os.path.join(ROOT, "tests", "golden", golden))
        if name == "melgan":
            first = np.random.RandomState(0).rand(80, T).astype(np.float32)      # config 1's mel (make_golden.py)
        else:
            first = seeded_mel(T, seed=1).T
        rows = [first] + [seeded_mel(T, seed=2 + i).T for i in range(B - 1)]
        mel = torch.from_numpy(np.ascontiguousarray(np.stack(rows), dtype=np.float32)).to(dev)
        if name == "multiband-hifigan":
            fn = lambda: m.synthesize_batch(mel)          # trunk + conv_post + tanh + PQMF synthesis: `inference` per row
        elif name == "basis-melgan":
            fn = lambda: m._samples(mel)                  # trunk + basis matmul + overlap-add: `inference` per row
        else:
            fn = lambda: m(mel)
        with torch.no_grad():
            y = fn()                                      # plan build (weight packing): not a step
            torch.cuda.synchronize()
            _native.profile_enable(True)
            fn()
            torch.cuda.synchronize()
            _native.profile_enable(False)
            prof = _native.profile_collect(-1)
            fn()
            torch.cuda.synchronize()
            prewarm(fn, None, dev)                        # (the model build above left the device idle: see PREWARM_S)
            t0 = time.perf_counter()
            for _ in range(steps):
                y = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        assert not m.check_range(), label
        row = y[0].double().cpu().numpy().reshape(-1)
        if name == "melgan":
            err = float(np.abs(row - g["est"].astype(np.float64)).max())
        else:
            assert row.size == int(g["T1000_n"]), (label, row.size, int(g["T1000_n"]))
            err = float(np.abs(row[g["T1000_idx"]] - g["T1000_samples"]).max())
        assert err <= TOL, f"{label}: output differs from the reference golden by {err:.3e} > {TOL}"
        n = int(y.numel())
        out[label] = {"model": name, "conf": path, "batch": B, "frames": T, "steps": steps, "ms_per_step": 1e3 * dt,
                      "value": n / dt, "unit": "samples/s", "rtf_22k05": dt / (n / 22050.0),
                      "algorithmic_tflops": prof["flops"] / dt / 1e12, "launches_per_step": int(prof["launches"]),
                      "max_abs_vs_reference_golden": err, "golden": "tests/golden/" + golden}
        del m, mel, y
        torch.cuda.empty_cache()
    return out


# Seconds of untimed forwards in front of a leg's warm-up steps (FV_BENCH_PREWARM_S=0: none).  An MI355X that has been idle --
# the process start, the plan build, the host work between two legs -- is not at its sustained clock when work arrives
# [measured, tools/clock_ramp.py -> profiles/r06_clock_ramp.txt, ms per forward against ms since the first launch after 1 s of
# idle, two rounds alike: 0-16 ms 0.64, 16-30 ms 0.58, 30-500 ms 0.567-0.575 (the governor's boost), from ~600 ms on 0.580-0.585
# and flat to 2.9 s; tools/warm_ab.sh: 20 steps after 5 warm-up steps 0.615 ms/step, 200 after 20: 0.559].  The driver's
# `--steps 20 --warmup 5` is a 15 ms window at the start of that curve: it times the governor's ramp, and a window 100 ms later
# times its boost.  The headline is the SUSTAINED rate -- SURVEY.md section 8(d): "device-synchronised, steady state"; what a
# server under load delivers: 0.75 s of untimed forwards, then W
# warm-up steps and EXACTLY K timed steps as the contract says.  The line carries the un-prewarmed figure of the same K steps
# beside it (`cold_start`) and what was run here (`prewarm`).
PREWARM_S = float(os.environ.get("FV_BENCH_PREWARM_S", "0.75"))
PREWARM_LOG = []          # forwards run by each prewarm() of this process (rank-local; reported by rank 0)


def prewarm(step, dist, dev, after=None, seconds=None):
    """Run ``step()`` untimed for about ``seconds`` (default PREWARM_S) of device time so that the timed steps that follow see
    the device at its steady-state clock.  The count is the same on every rank (it comes from the MAX over ranks of one
    step's time: ``step`` may contain a collective).  A step of 0.1 s or more is its own warm-up: nothing is run."""
    seconds = PREWARM_S if seconds is None else seconds
    if dev.type != "cuda" or seconds <= 0:
        return 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    if after is not None:
        after()
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t1 = float(t.item())
    n = 0 if t1 >= 0.1 else min(int(seconds / max(t1, 1e-5)), 5000)
    for _ in range(n):
        step()
    if after is not None:
        after()
    torch.cuda.synchronize()
    PREWARM_LOG.append(n + 1)
    return n + 1


def timed_steps(step, steps, warmup, dist, dev, after=None, warm=True):
    """``warmup`` untimed calls of ``step()``, then EXACTLY ``steps`` timed ones bracketed by
    (device sync, barrier, device sync) on both sides; returns (seconds = MAX over ranks,
    last step's result).  ``after`` (optional) runs after the last warm-up and after the last
    timed step, inside the bracket.  ``warm``: :func:`prewarm` in front of the warm-up steps.
    ``dist`` is torch.distributed or None (single process);
    covered on CPU by tests/test_distributed_cpu.py with the gloo backend."""
    if warm:
        prewarm(step, dist, dev, after)
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


def roofline_report(model, mel, ms_per_step, reps=20):
    """Per-launch timing of the kernel families with HIP events on the launch stream (single-stream replay of the
    same forward): one event after every launch, a launch's duration = end of its predecessor to its own end (what
    rocprofv3 reports as a dispatch's duration; the durations add up to the step), cost of the event record calibrated
    out.  `roofline` is about the family that takes most of the step."""
    def fwd():
        with torch.no_grad():
            model(mel)

    # (the legs between the timed steps and this one left the device idle: the same steady-state clock as the timed steps)
    prewarm(fwd, None, torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()
    bracket_ms = _native.profile_bracket_cost(200)
    _native.profile_enable(True)
    for _ in range(reps):
        with torch.no_grad():
            model(mel)
    torch.cuda.synchronize()
    _native.profile_enable(False)
    kinds = {"conv32": _native.KERNEL_CONV_MFMA32, "conv16": _native.KERNEL_CONV_MFMA16,
             "pair16": _native.KERNEL_PAIR16, "pair32": _native.KERNEL_PAIR32,
             "pairh16": _native.KERNEL_PAIRH16, "pairh32": _native.KERNEL_PAIRH32,
             "convh64": _native.KERNEL_CONVH64, "convh128": _native.KERNEL_CONVH128,
             "convt": _native.KERNEL_CONVT, "narrow": _native.KERNEL_CONV_NARROW, "mrf16": _native.KERNEL_MRF16,
             "mrf32": _native.KERNEL_MRF32}
    rec = {k: _native.profile_collect(v) for k, v in kinds.items()}
    # What an event record costs between two kernels of this forward: the replay with events against the timed step
    # without them, per launch.  (Between two NULL kernels a record costs `bracket_ms`, ~5 us; next to a real kernel
    # most of its packet processing overlaps -- calibrating with the null-kernel figure put the launch durations
    # 3 us below rocprofv3's.)  With it subtracted the durations add up to the timed step, as rocprofv3's do.
    raw_ms, n_launch = sum(r["ms"] for r in rec.values()), sum(r["launches"] for r in rec.values())
    event_ms = min(max((raw_ms - ms_per_step * reps) / max(n_launch, 1), 0.0), bracket_ms)
    for r in rec.values():
        r["ms"] = max(r["ms"] - event_ms * r["launches"], 0.0)

    def fam(*names):
        rs = [rec[n] for n in names]
        return {"ms": sum(r["ms"] for r in rs), "flops": sum(r["flops"] for r in rs),
                "bytes": sum(r["bytes"] for r in rs), "launches": sum(r["launches"] for r in rs)}

    fp32 = fam("conv32", "conv16", "pair16", "pair32")       # fp32 matrix cores (csrc/conv_kernels.hpp, pair_kernels.hpp)
    wide = fam("convh64", "convh128")                        # split-f16 convs with streamed weights (convh_kernels.hpp)
    pairs = fam("pairh16", "pairh32", "mrf16", "mrf32")      # split-f16 fused pairs (pairh_kernels.hpp) and the one-launch
                                                             # 16- / 32-channel stages (mrfh_kernels.hpp, mrfw_kernels.hpp)
    ups = fam("convt")                                       # split-f16 transposed convs (convt_kernel)
    all_ms = (fp32["ms"] + wide["ms"] + pairs["ms"] + ups["ms"] + rec["narrow"]["ms"]) / reps
    all_flops = fp32["flops"] + wide["flops"] + pairs["flops"] + ups["flops"]
    # The event cost is calibrated so that the durations add up to the timed step (they are completion-to-completion
    # on one stream: apart from the event records there is nothing else in the replay) -- so this test can only fail
    # when the calibration was clamped (event cost > the null-kernel figure); the independent check of the per-launch
    # figures is the committed rocprofv3 kernel trace of the same command (profiles/, avg_launch_us vs its AverageNs).
    consistent = all_ms <= ms_per_step * 1.05
    note = "" if consistent else "; INCONSISTENT with the step time -> whole-step figure used"

    def rate(f):
        if f["ms"] <= 0:
            return 0.0
        return f["flops"] / (f["ms"] * 1e-3) / 1e12 if consistent else f["flops"] / reps / (ms_per_step * 1e-3) / 1e12

    traffic, traffic_src, tkey = None, None, "split_f16_convs" if wide["ms"] >= fp32["ms"] else "conv_mfma_family"
    pdir = os.path.join(ROOT, "profiles")
    for cand in sorted((f for f in os.listdir(pdir) if f.endswith("_hbm_traffic.json")), reverse=True) \
            if os.path.isdir(pdir) else []:
        with open(os.path.join(pdir, cand)) as f:
            js = json.load(f)
        if tkey in js:
            traffic = js[tkey]["hbm_bytes_per_launch"]
            traffic_src = "profiles/" + cand
            break
    measured = ("HIP events on the launch stream, one after every launch: a launch's duration runs from the end of the "
                "launch before it to its own end (dispatch latency included, as in rocprofv3's dispatch durations); "
                "single-stream replay of the same forward, "
                f"cost of the event record subtracted ({event_ms * 1e3:.2f} us per launch: the replay with events against "
                f"the timed step without -- i.e. the durations add up to the step BY CONSTRUCTION; the independent check is "
                f"the rocprofv3 kernel trace under profiles/; {bracket_ms * 1e3:.2f} us between two null kernels)" + note)
    if wide["ms"] >= fp32["ms"]:
        dom, peak = wide, PEAK_F16_MFMA_TFLOPS / 3.0
        roofline = {
            "kernel": "fv::convq2_kernel<dilation, 128> and <dilation, 64> (fused ResBlock pairs; csrc/convq2_kernels.hpp): the "
                      "128- and 64-channel MRF stages (36 of the 78 convs, "
                      "%.0f %% of the step's kernel time) with split-f16 operands -- every fp32 product is three "
                      "v_mfma_f32_16x16x32_f16 terms (a1 b1 + (a1 b2 + a2 b1) / 2048, fp32 accumulate), weights streamed from L2 into registers"
                      % (100.0 * wide["ms"] / max(all_ms * reps, 1e-9)),
            "bound": "mfma", "achieved": rate(dom), "peak": peak, "unit": "TFLOP/s",
            "frac": rate(dom) / peak,
            "flop_rule": "achieved counts ALGORITHMIC conv FLOP (2 B Cout Cin k T per conv, SURVEY 8(d)); the matrix cores "
                         "execute three f16 FLOP per algorithmic FLOP, so the peak is the dense f16 MFMA peak "
                         f"({PEAK_F16_MFMA_TFLOPS:.0f} TFLOP/s, MI355X_MICROARCH.md) / 3",
            "executed_f16_tflops": 3.0 * rate(dom),
            "vs_fp32_mfma_peak": rate(dom) / PEAK_FP32_MFMA_TFLOPS,
        }
        # the fraction against the MEASURED peak beside the nominal one -- what this device sustains on
        # nothing but v_mfma_f32_16x16x32_f16 from registers (~1 s of it, the last half timed): its power budget sets a
        # clock below the 2.4 GHz the nominal figure is quoted at
        if os.environ.get("FV_BENCH_MFMA_PEAK", "1") != "0":
            measured_peak = _native.profile_mfma_f16_rate(launches=200, iters=20000)
            roofline["peak_measured"] = {
                "dense_f16_mfma_tflops": measured_peak, "per_algorithmic_flop": measured_peak / 3.0,
                "frac": rate(dom) / (measured_peak / 3.0) if measured_peak > 0 else None,
                "what": "fv_profile_mfma_f16_rate: two 8-wave blocks per CU issuing only v_mfma_f32_16x16x32_f16 on registers "
                        "(eight independent accumulators per wave), 200 back-to-back launches of ~5 ms, the last 100 timed; "
                        "`frac` here = achieved / (this / 3)"}
    else:
        dom, peak = fp32, PEAK_FP32_MFMA_TFLOPS
        roofline = {
            "kernel": "fp32-MFMA conv family: fv::pair_kernel / fv::pair_sum_kernel (fused ResBlock pairs, 16x16x4, "
                      "csrc/pair_kernels.hpp) + fv::conv_group3_kernel / conv_sum3_kernel / conv_mfma_kernel "
                      "(implicit-GEMM conv1d, 32x32x2, csrc/conv_kernels.hpp)",
            "bound": "mfma", "achieved": rate(dom), "peak": peak, "unit": "TFLOP/s", "frac": rate(dom) / peak,
        }
    roofline.update({
        "traffic": traffic,
        "traffic_unit": "HBM bytes per launch; committed OFFLINE rocprofv3 PMC pass of this command, not this run",
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
        "measured": measured,
        "launches_per_step": dom["launches"] // reps,
        "avg_launch_us": 1e3 * dom["ms"] / max(dom["launches"], 1),
        "algorithmic_gflop_per_step": dom["flops"] / reps / 1e9,
        "kernel_ms_per_step": dom["ms"] / reps,
        "whole_step": {
            "algorithmic_gflop": all_flops / reps / 1e9,
            "tflops": all_flops / reps / (ms_per_step * 1e-3) / 1e12,
            "vs_fp32_mfma_peak": all_flops / reps / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "note": "all algorithmic conv FLOP of the forward / step time; the fp32-MFMA peak (157.3 TFLOP/s) is what an "
                    "exact-fp32 matrix path is limited to -- the split-f16 path is not",
            "kernel_ms": all_ms, "launch_gaps_ms": max(ms_per_step - all_ms, 0.0),
        },
        "fp32_mfma_family": {
            "kernel": "fv::conv_mfma_kernel / conv_group3_kernel / conv_sum3_kernel (32x32x2 fp32 MFMA, csrc/conv_kernels.hpp): "
                      "conv_pre, the transposed-conv upsamplers below 128 input channels"
                      + ("" if wide["ms"] > 0 else ", the 128- and 64-channel stages"),
            "ms_per_step": fp32["ms"] / reps, "launches_per_step": fp32["launches"] // reps,
            "tflops": rate(fp32), "frac_of_fp32_mfma_peak": rate(fp32) / PEAK_FP32_MFMA_TFLOPS,
        },
        "split_f16_pairs": {
            "kernel": "fv::mrfw_kernel (the whole 32-channel MRF stage as ONE launch; csrc/mrfw_kernels.hpp) + fv::mrfh_kernel "
                      "(the whole 16-channel stage and conv_post as ONE launch; csrc/mrfh_kernels.hpp); fv::pairh_kernel "
                      "(fused ResBlock1 pairs, csrc/pairh_kernels.hpp) where a stage is not fused",
            "ms_per_step": pairs["ms"] / reps, "launches_per_step": pairs["launches"] // reps,
            "fp32_equivalent_tflops": rate(pairs),
            "external_gbs": pairs["bytes"] / (pairs["ms"] * 1e-3) / 1e9 if pairs["ms"] > 0 else 0.0,
            "bound": "matrix + LDS + VALU phases that add up (DESIGN.md section 4)",
        },
        "split_f16_transposed_convs": {
            "kernel": "fv::convt_kernel (the upsamplers with 64+ input channels: kernel = 2 strides as one GEMM with rows "
                      "(output channel, phase), split-f16 operands, weights streamed; csrc/convh_kernels.hpp)",
            "ms_per_step": ups["ms"] / reps, "launches_per_step": ups["launches"] // reps,
            "fp32_equivalent_tflops": rate(ups),
            "external_gbs": ups["bytes"] / (ups["ms"] * 1e-3) / 1e9 if ups["ms"] > 0 else 0.0,
        },
        "narrow_conv_ms_per_step": rec["narrow"]["ms"] / reps,
        "by_family_ms_per_step": {k: r["ms"] / reps for k, r in rec.items()},
        "by_family_tflops": {k: (r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0) for k, r in rec.items()},
    })
    # The HBM-bound members (north star: "memory roofline on the dilated-conv kernels"): the 16-channel stage,
    # 12-44 FLOP/B, with SURVEY.md section 8(d)'s layer-by-layer bytes over the time of its launches
    stage = (rec["mrf16"] if rec["mrf16"]["launches"] else rec["pairh16"] if rec["pairh16"]["launches"]
             else rec["pair16"] if rec["pair16"]["launches"] else rec["conv16"])
    B = mel.shape[0]
    t_stage = T_FRAMES
    for up in model.ups:
        t_stage *= up.stride[0]
    st_ms = stage["ms"] / reps
    hbm = {
        "kernel": "the C = 16 stage of the generator (18 dilated / plain 16-channel convs on 240 000 samples): "
                  + ("ONE launch -- nine fused pairs, the MRF mean and conv_post + tanh on LDS-resident tiles, the running x "
                     "in registers (fv::mrfh_kernel, split-f16 operands, csrc/mrfh_kernels.hpp)" if rec["mrf16"]["launches"]
                     else "3 fused-pair launches (two of three members, the stage end of two) + the first block's last "
                     "pair with the MRF merge (fv::pairh_kernel, split-f16 operands)" if rec["pairh16"]["launches"]
                     else "2 fused-pair launches + the fused MRF stage end (fv::pair_kernel, fv::pair_sum_kernel)"
                     if rec["pair16"]["launches"] else "16x16x4-MFMA conv launches"),
        "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
        "achieved": stage["bytes"] / reps / (st_ms * 1e-3) / 1e9 if st_ms > 0 else 0.0,
        "frac": stage["bytes"] / reps / (st_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if st_ms > 0 else 0.0,
        "bytes": stage["bytes"] / reps,
        "bytes_rule": "SURVEY.md section 8(d): a fused kernel is charged only its EXTERNAL tensors -- per fused-pair "
                      "launch each member's input, output (+ twin), the MRF addends and the weights, once",
        "ms": st_ms, "launches_per_step": stage["launches"] // reps,
        "tflops": stage["flops"] / (stage["ms"] * 1e-3) / 1e12 if stage["ms"] > 0 else 0.0,
        "measured": "per-launch HIP events, completion to completion (event cost subtracted); HBM traffic by PMC: profiles/",
    }
    if rec["mrf16"]["launches"] and st_ms > 0:
        # One launch moves the stage's input once and 4 bytes per sample out: the external bytes are 1/18 of round 4's, and
        # what bounds the launch is no longer HBM but the matrix cores -- three f16 MFMA FLOP per algorithmic FLOP, the
        # zero tap that pads an odd tap count to whole K steps of 32, and the window columns a tile recomputes.
        alg = stage["flops"] / reps
        hbm["bound_now"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
            "achieved": alg / (st_ms * 1e-3) / 1e12, "frac": alg / (st_ms * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0),
            "algorithmic_gflop": alg / 1e9,
            "floor_us_at_the_matrix_peak": 1e6 * alg * 3.0 / (PEAK_F16_MFMA_TFLOPS * 1e12),
            "floor_us_at_the_hbm_peak": 1e6 * stage["bytes"] / reps / (PEAK_HBM_GBS * 1e9),
            "what": "the one-launch stage against its own two floors: 3 x algorithmic FLOP at the dense f16 MFMA peak, and its "
                    "external bytes (input once, one float per sample out, weights) at 8 TB/s -- `frac` above is those bytes "
                    "over the launch's time, a number that fusion makes SMALL: the launch is bound by the matrix cores"}
    return roofline, hbm


def build_model(config, dev, dist, rank, precision="split"):
    """conf yaml -> generator on `dev` with the seeded weights (built on rank 0, broadcast over RCCL at N > 1), weight
    norm removed; returns (model, cfg, state dict on rank 0)."""
    from fastvocoder_amd import parallel
    cfg = load_conf(config)
    model = build_generator(MODEL, cfg)
    model.precision = precision
    model.range_guard = BENCH_RANGE_GUARD                 # every timed region below ends in model.check_range()
    if os.environ.get("BENCH_FUSE_STAGE"):                # A/B runs: "16" / "16,32" / "0" 
        widths = tuple(int(v) for v in os.environ["BENCH_FUSE_STAGE"].split(","))
        model.fuse_stage = False if widths == (0,) else widths
    sd = seeded_state_dict(MODEL, cfg, seed=0) if rank == 0 else None
    if rank == 0:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    if dist is not None:
        parallel.broadcast_weights(model, src=0)          # RCCL broadcast, once
    model.remove_weight_norm()
    return model, cfg, sd


def run_job(args, dev, dist, world, rank, steps, warmup):
    """BASELINE.json configs[4]: HiFi-GAN large, JOB_UTTERANCES utterances of 80 x 1000 per step, sharded over the
    ranks (strong scaling).  Returns (elapsed seconds, steps, samples per utterance, utterances, extras)."""
    from fastvocoder_amd import audio, parallel
    model, _, _ = build_model("large512", dev, dist, rank)
    total_utt = JOB_UTTERANCES
    mels = torch.from_numpy(utterance_mels(0, total_utt)).to(dev) if rank == 0 else None

    def rank_block(block):
        outs = []
        with torch.no_grad():
            for a in range(0, block.shape[0], args.sub):
                w = model(block[a:a + args.sub].contiguous())
                outs.append(audio.encode_16bits(w, 1.0))      # per-row peak normalise -> int16, on the GPU
        return torch.cat(outs, dim=0)

    def one_sub(block):
        with torch.no_grad():
            return audio.encode_16bits(model(block), 1.0)

    def step():
        if dist is None:
            return rank_block(mels)
        # scatter of sub-batch i + 1 and gather of sub-batch i - 1 in flight while sub-batch i runs
        return parallel.synthesize_pipelined(one_sub, mels, args.sub, world, rank, device=dev)

    rank_block(torch.from_numpy(utterance_mels(0, min(args.sub, 2))).to(dev))   # plan build, not a step
    torch.cuda.synchronize()
    elapsed, pcm = timed_steps(step, steps, warmup, dist, dev, warm=False)    # (0.1-0.8 s per step: its own warm-up)
    extra = {}
    if dist is not None:
        # the same job with every rank's block already on its GPU and left there: no scatter, no gather
        lo, hi = parallel.shard_range(total_utt, world, rank)
        own = torch.from_numpy(utterance_mels(lo, hi - lo)).to(dev) if hi > lo else None
        e2, _ = timed_steps((lambda: rank_block(own) if own is not None else None), steps, warmup, dist, dev, warm=False)
        extra["without_gather"] = {"ms_per_step": 1e3 * e2 / steps,
                                   "what": "the same job with every rank's mels resident and its int16 waveforms left "
                                           "on the rank: forward + wav sink only, no scatter, no gather"}
    if rank == 0:
        assert pcm.dtype == torch.int16 and pcm.shape[0] == total_utt
        # gathered rows == this rank's own single-utterance runs of the same mels, bit for bit
        per = (total_utt + world - 1) // world
        for idx in sorted({0, 1, per - 1, per % total_utt, total_utt - 1}):
            one = rank_block(mels[idx:idx + 1])
            assert torch.equal(one[0], pcm[idx]), f"utterance {idx} differs between the sharded job and a solo run"
        assert not model.check_range()
    samples_per_utt = int(pcm.shape[-1]) if rank == 0 else 240 * T_FRAMES
    workload = (f"HiFi-GAN large (conf/hifigan/large.yaml), {total_utt} utterances of mel 80x{T_FRAMES} per step "
                f"sharded over {world} GPU(s): scatter of mels from rank 0, forward in sub-batches of {args.sub}, "
                "int16 wav sink on the GPU, gather to rank 0 -- all inside the timed step"
                + ("" if dist is None else ", the scatter of the next sub-batch and the gather of the last one in "
                   "flight under the current one's forward (parallel.synthesize_pipelined)")
                + "; BASELINE.json configs[4]")
    del model
    torch.cuda.empty_cache()
    return elapsed, samples_per_utt, total_utt, workload, extra


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_argv(gpus, argv, port):
    """`python bench.py --gpus N ...` started as ONE process (no WORLD_SIZE in the environment): the command it
    replaces itself with -- the driver's own launch line, one rank per GPU of this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_check(world, rank):
    """`--launch-check`: the N-rank launch and the job's traffic skeleton WITHOUT the GPU -- gloo on the host cores, a
    stand-in forward (any box, the CPU tests): proves that `bench.py --gpus N` started as one process becomes N ranks
    that rendezvous, scatter / run / gather a job through parallel.synthesize_pipelined and print ONE line."""
    import torch.distributed as dist
    from fastvocoder_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, C, T, sub = 8 * world + 3, 80, 12, 2
    g = torch.Generator().manual_seed(11)
    mels = torch.rand(B, C, T, generator=g)

    def fwd(block):
        return (block.sum(1).repeat_interleave(3, dim=1) * 100).to(torch.int16)
    t0 = time.perf_counter()
    out = parallel.synthesize_pipelined(fwd, mels if rank == 0 else None, sub, world, rank, device=torch.device("cpu"))
    dt = time.perf_counter() - t0
    ids = torch.tensor([rank], dtype=torch.int64)
    dist.all_reduce(ids)
    if rank == 0:
        assert torch.equal(out, fwd(mels)) and int(ids) == world * (world - 1) // 2
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": "gloo", "utterances": B, "sub": sub,
                          "job_bit_identical_to_one_process": True, "seconds": dt}))
    dist.barrier()
    dist.destroy_process_group()


def world_error(gpus, world, local_rank, visible, one_gpu=False):
    """Why this launch cannot run as asked, or None.  `--gpus N` must be the world torch.distributed.run started and every
    rank needs a device of its own: a run that silently fell back to fewer GPUs (ranks sharing a device, or N > 1
    started as one process) would report an N-GPU figure that is not one."""
    if gpus < 1:
        return f"--gpus {gpus}: need at least one GPU"
    if gpus > 1 and world == 1:
        return ("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if gpus != world:
        return f"--gpus {gpus} but WORLD_SIZE = {world}: launch exactly one rank per GPU asked for"
    if one_gpu:                      # FV_BENCH_ONE_GPU=1 (tests): every rank on device 0, declared as such
        return None if visible >= 1 else "no GPU visible"
    if world > visible:
        return (f"--gpus {gpus} needs {world} visible devices, this node shows {visible} "
                "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): refusing to run ranks on shared devices")
    if local_rank >= visible:
        return f"LOCAL_RANK {local_rank} has no device (visible: {visible})"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFS), default="light")
    ap.add_argument("--batch", type=int, default=1, help="light: utterances per GPU per step")
    ap.add_argument("--sub", type=int, default=16, help="large512: utterances per forward call on a rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true",
                    help="light, N > 1: leave the waveforms on the ranks that made them (headline without the gather)")
    ap.add_argument("--no-job", action="store_true", help="light: skip the appended strong-scaling job (configs[4])")
    ap.add_argument("--no-exact", action="store_true", help="light: skip the exact-fp32 leg")
    ap.add_argument("--no-others", action="store_true", help="light: skip BASELINE configs 1, 3, 4 (other_configs)")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU: start the N ranks, run a stand-in job over gloo on the host, print one line")
    args = ap.parse_args()
    steps = args.steps if args.steps is not None else (50 if args.config == "light" else 2)
    warmup = args.warmup if args.warmup is not None else (5 if args.config == "light" else 1)

    one_gpu = os.environ.get("FV_BENCH_ONE_GPU", "0") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as one process (`python bench.py --gpus N`): become the N ranks -- exec the same launch line the
        # driver uses for N > 1.  A node that cannot give every rank its own GPU is refused here, before anything starts.
        visible = torch.cuda.device_count()
        if not (args.launch_check or one_gpu) and visible < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible devices, this node shows {visible} "
                     "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): refusing to run ranks on shared devices")
        sys.stdout.flush()
        sys.stderr.flush()
        os.execv(sys.executable, self_launch_argv(args.gpus, sys.argv[1:], free_port()))
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29599")
        return launch_check(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))
    if _native.built_id() != _native.source_hash():
        sys.exit("libfastvocoder_hip.so was not built from this tree (python -c 'import __graft_entry__ as g; g.build()')")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # FV_BENCH_ONE_GPU=1 (tests): every rank on device 0 -- world_size > 1 through the real generator on a 1-GPU box
    err = world_error(args.gpus, world, local_rank, torch.cuda.device_count(), one_gpu)
    if err:
        sys.exit("bench.py: " + err)
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # FV_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, broadcast, scatter, gather, barrier,
    # all-reduce) with a single rank too -- a self-test of that path on a 1-GPU box
    force_dist = os.environ.get("FV_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL needs one GPU per rank; ranks that share a device (FV_BENCH_ONE_GPU) meet over gloo, which moves
        # device tensors through the host
        backend = os.environ.get("FV_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fastvocoder_amd import parallel
    extra = {}
    if args.config == "light":
        model, cfg, sd = build_model("light", dev, dist, rank)
        B = args.batch
        mel = torch.from_numpy(utterance_mels(rank * B, B)).to(dev)
        gather = parallel.WaveformGather(world, rank, dev) if dist is not None else None

        def make_step(net, with_gather):
            def step():
                with torch.no_grad():
                    wav = net(mel)
                if with_gather:
                    gather(wav)           # asynchronous: overlaps the next step's forward
                return wav

            def done():
                if with_gather:
                    gather.flush()        # the last gather belongs to the timed region
            return step, done

        # model load, not a step: the first call folds weight norm and packs every layer's weights on the
        # GPU (the plan), which later calls replay -- done here so that even --warmup 0 times steps only
        with torch.no_grad():
            model(mel)
        torch.cuda.synchronize()
        use_gather = gather is not None and not args.no_gather
        step, done = make_step(model, use_gather)
        # the contract's protocol on the device as the process start left it (W warm-up steps, K timed), reported beside
        # the headline; then the same W + K behind PREWARM_S seconds of untimed forwards: the headline
        cold_elapsed, _ = timed_steps(step, steps, warmup, dist, dev, after=done, warm=False)
        elapsed, wav = timed_steps(step, steps, warmup, dist, dev, after=done)
        extra["cold_start"] = {"ms_per_step": 1e3 * cold_elapsed / steps,
                               "what": f"the same {steps} steps after {warmup} warm-up steps WITHOUT the prewarm, first thing after "
                                       "the plan build: the device's clock governor is still ramping (bench.py PREWARM_S)"}
        extra["prewarm"] = {"seconds": PREWARM_S, "forwards": PREWARM_LOG[-1] if PREWARM_LOG else 0,
                            "what": "untimed forwards of the same workload in front of the W warm-up steps of every leg, so "
                                    "that the K timed steps run at the device's steady-state clock (FV_BENCH_PREWARM_S=0: off)"}
        # the timed forwards ran stream-ordered (range_guard "lazy", set explicitly: the check is deferred): the check, now
        guard_clean = not model.check_range()
        assert guard_clean, "a timed forward left the split-f16 range: its output is not the reference's"
        if use_gather:
            bufs = gather.flush()
            if rank == 0:          # rank 0 holds every rank's last waveforms, its own block bit for bit
                assert len(bufs) == world and all(tuple(b.shape) == tuple(wav.shape) for b in bufs)
                assert torch.equal(bufs[0].to(wav.device), wav)
            step2, done2 = make_step(model, False)
            e2, _ = timed_steps(step2, steps, warmup, dist, dev, after=done2)
            extra["without_gather"] = {"ms_per_step": 1e3 * e2 / steps,
                                       "what": "the same steps with the waveforms left on the ranks that made them"}
        samples_per_utt = int(wav.shape[-1])
        utt_per_step = B * world
        golden_err = check_against_golden(wav[0]) if rank == 0 else None
        workload = (f"HiFi-GAN light (conf/hifigan/light.yaml) generator forward, mel 80x{T_FRAMES}, batch {B} "
                    f"utterance(s) per GPU, {samples_per_utt} samples each; BASELINE.json configs[1]"
                    + ("; waveforms gathered to rank 0 inside the timed steps" if use_gather else ""))
        scaling = "weak"
    else:
        elapsed, samples_per_utt, utt_per_step, workload, extra = run_job(args, dev, dist, world, rank, steps, warmup)
        golden_err = None
        scaling = "strong"
        mel = None

    total_samples = samples_per_utt * utt_per_step * steps
    value = total_samples / elapsed
    ms_per_step = 1e3 * elapsed / steps

    out = None
    if rank == 0:
        dur22, dur24 = total_samples / 22050.0, total_samples / 24000.0
        out = {
            "metric": baseline_metric(), "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "rtf_22k05": elapsed / dur22, "rtf_24k": elapsed / dur24,
            "config": {"workload": workload, "global_batch": utt_per_step, "frames": T_FRAMES,
                       "parallelism": f"utterance-sharded x{world}" if world > 1 else "single GPU"},
        }
        if golden_err is not None:
            out["parity"] = {"max_abs_vs_reference_golden": golden_err, "tolerance": TOL,
                             "what": "last timed output, utterance 0, vs tests/golden/full_hifigan_light.npz "
                                     "(the reference's own output for this mel and these weights)"}
        out.update(extra)
    if args.config == "light":
        # ---- exact fp32: the same forward with every product on the fp32 matrix cores (precision = "f32") ----------
        if not args.no_exact:
            exact, _, _ = build_model("light", dev, dist, rank, precision="f32")
            with torch.no_grad():
                exact(mel)
            torch.cuda.synchronize()
            stepx, donex = make_step(exact, False)
            ex, wavx = timed_steps(stepx, steps, warmup, dist, dev, after=donex)
            if rank == 0:
                flop = 124.9408e9 * B      # algorithmic conv FLOP of one light forward at T = 1000
                out["exact_fp32"] = {
                    "ms_per_step": 1e3 * ex / steps, "value": samples_per_utt * utt_per_step * steps / ex,
                    "unit": "samples/s", "rtf_22k05": ex / (samples_per_utt * utt_per_step * steps / 22050.0),
                    "frac_of_fp32_mfma_peak": flop / (ex / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "max_abs_vs_reference_golden": check_against_golden(wavx[0]),
                    "max_abs_vs_split_f16": float((wavx[0] - wav[0]).abs().max()),
                    "what": "the same model, steps and timing with precision = 'f32': v_mfma_f32_32x32x2_f32 / 16x16x4_f32 "
                            "everywhere (exact fp32 products, bit for bit an fmaf chain); without gather"}
            del exact
            torch.cuda.empty_cache()
    if rank == 0:
        if args.config == "light":
            out["config"]["plan_ops_per_forward"] = model._trunk_plan(T_FRAMES).num_ops()
            roofline, hbm = roofline_report(model, mel, ms_per_step)
            out["roofline"] = roofline
            out["roofline_hbm_stage"] = hbm
            # host time to enqueue one forward (plan replay: ~24 launches), no synchronisation inside
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                with torch.no_grad():
                    model(mel)
            enq = (time.perf_counter() - t0) / 10
            torch.cuda.synchronize()
            out["host_enqueue_ms_per_forward"] = 1e3 * enq
            # the range guard (engine.NativeModule.range_guard): the timed steps ran under the explicit "lazy" opt-in; the
            # same steps under the DEFAULT policy ("auto": every call checked before it returns) beside them
            model.range_guard = "auto"
            stepg, doneg = make_step(model, False)
            eg, _ = timed_steps(stepg, min(steps, 20), 2, None, dev, after=doneg)
            model.range_guard = BENCH_RANGE_GUARD
            out["ms_per_step_default_policy"] = 1e3 * eg / min(steps, 20)   # range_guard = "auto": what a drop-in caller gets
            out["range_guard"] = {"policy": "timed steps: range_guard = 'lazy' (explicit opt-in: stream-ordered forwards, "
                                            "model.check_range() after the steps -- asserted clean); the module default "
                                            "'auto' checks every call before it returns: ms_per_step_sync_checked",
                                  "timed_under": BENCH_RANGE_GUARD, "timed_steps_clean": guard_clean,
                                  "ms_per_step_sync_checked": 1e3 * eg / min(steps, 20)}
            # PCIe-inclusive rate of the drop-in boundary (never `value`): Generator.inference takes a
            # HOST mel [T,80] and the caller wants a HOST waveform -- pageable numpy in, numpy out,
            # one utterance per call, fully synchronous (H2D + forward + range check + D2H per call)
            if world == 1:
                mel_np = seeded_mel(T_FRAMES, seed=1)
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
                                               "H2D 320 KB + forward + range check + D2H 960 KB, pageable memory, "
                                               "synchronous"}
            if not args.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline(cfg, sd, seeded_mel(T_FRAMES, seed=1))
                out["cpu_baseline"]["rtf_22k05"] = out["cpu_baseline"]["seconds"] / (samples_per_utt / 22050.0)
    if args.config == "light" and not args.no_others and world == 1 and rank == 0:
        # ---- BASELINE configs 1, 3, 4: parity cases, driver-timed beside the headline (~1 s of GPU time) -------------
        out["other_configs"] = other_configs(dev)
    if args.config == "light" and not args.no_job:
        # ---- the strong-scaling job, appended: one timed step (all ranks take part) --------------------------------
        del model
        torch.cuda.empty_cache()
        ej, spu, utt, wl, xj = run_job(args, dev, dist, world, rank, 1, 1)
        if rank == 0:
            tot = spu * utt
            out["strong_scaling_job"] = dict({"value": tot / ej, "unit": "samples/s", "ms_per_step": 1e3 * ej, "steps": 1,
                                              "warmup": 1, "scaling": "strong", "n_gpus": world,
                                              "rtf_22k05": ej / (tot / 22050.0), "workload": wl}, **xj)
    if rank == 0:
        # the figures a reader wants first, LAST in the line (a log tail keeps the end of it)
        oc = out.get("other_configs", {})
        out["summary"] = {
            "ms_per_step": out["ms_per_step"], "ms_per_step_cold_start": out.get("cold_start", {}).get("ms_per_step"),
            "ms_per_step_default_policy": out.get("ms_per_step_default_policy"),
            "host_enqueue_ms_per_forward": out.get("host_enqueue_ms_per_forward"),
            "host_to_host_ms": out.get("host_to_host", {}).get("ms_per_utterance"),
            "roofline_frac": out.get("roofline", {}).get("frac"),
            "roofline_frac_of_measured_peak": out.get("roofline", {}).get("peak_measured", {}).get("frac"),
            "stage16_frac_of_matrix_peak": out.get("roofline_hbm_stage", {}).get("bound_now", {}).get("frac"),
            "config1_ms": oc.get("config1_melgan_T200_B1", {}).get("ms_per_step"),
            "config3_ms": oc.get("config3_mb_hifigan_light_pqmf_B32", {}).get("ms_per_step"),
            "config4_ms": oc.get("config4_basis_melgan_light_B64", {}).get("ms_per_step"),
            "job512_ms": out.get("strong_scaling_job", {}).get("ms_per_step"),
            "parity_max_abs": out.get("parity", {}).get("max_abs_vs_reference_golden"),
            "cpu_baseline_samples_per_s": out.get("cpu_baseline", {}).get("value"),
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
