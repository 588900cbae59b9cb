"""world_size-2 (and 3) gloo tests of the utterance-sharding layer
(fastvocoder_amd/parallel.py): the N>1 code path of bench.py / serving, on CPU
with a stand-in forward (the HIP generator itself needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastvocoder_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_generator(mels):
    """Deterministic per-utterance 'waveform': depends only on that row."""
    B, C, T = mels.shape
    w = torch.arange(1, C + 1, dtype=torch.float32).view(1, C, 1)
    return (mels * w).sum(1).repeat_interleave(3, dim=1)      # [B, 3T]


def _worker(rank, world, port, B, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1) weights: rank 0 holds the checkpoint, everyone ends up with it
        torch.manual_seed(rank)
        lin = torch.nn.Linear(4, 3)
        parallel.broadcast_weights(lin, src=0)
        ref = torch.nn.Linear(4, 3)
        torch.manual_seed(0)
        ref = torch.nn.Linear(4, 3)
        assert torch.equal(lin.weight, ref.weight) and torch.equal(lin.bias, ref.bias)
        # 2) sharded synthesis, ragged batch, gathered in utterance order on rank 0
        g = torch.Generator().manual_seed(123)
        mels = torch.rand(B, 5, 7, generator=g)
        out = parallel.synthesize_sharded(_fake_generator, mels)
        if rank == 0:
            assert out.shape == (B, 21)
            assert torch.equal(out, _fake_generator(mels))        # bit-identical to one process
            np.save(out_path, out.numpy())
        else:
            assert out is None
        # 2b) the job form (bench.py --config large512): only the root holds the mels, blocks are
        #     scattered, results come back in the generator's dtype (int16 after a wav sink)
        def _sink(block):
            return (_fake_generator(block) * 100).to(torch.int16)
        out = parallel.synthesize_sharded(_sink, mels if rank == 0 else None, scatter=True,
                                          device=torch.device("cpu"))
        if rank == 0:
            assert out.dtype == torch.int16 and torch.equal(out, _sink(mels))
        else:
            assert out is None
        # 3) the fixed-shape gather used by bench.py
        gather = parallel.WaveformGather(world, rank, torch.device("cpu"))
        mine = torch.full((2, 6), float(rank))
        bufs = gather(mine)
        gather.flush()                                   # asynchronous gather: complete it
        if rank == 0:
            assert [float(b[0, 0]) for b in bufs] == [float(r) for r in range(world)]
        # back-to-back gathers reuse the buffers; each call completes the previous one first
        for it in range(3):
            gather(torch.full((2, 6), float(rank + 10 * it)))
        bufs = gather.flush()
        if rank == 0:
            assert [float(b[1, 5]) for b in bufs] == [float(r + 20) for r in range(world)]
        # 4) bench.py's timing bracket: K steps, barrier on both sides, MAX over ranks
        import time

        import bench
        calls = []

        def step():
            calls.append(1)
            time.sleep(0.02 * (rank + 1))          # rank r is (r+1)x slower
            return rank
        elapsed, last = bench.timed_steps(step, steps=3, warmup=1, dist=dist, dev=torch.device("cpu"))
        assert len(calls) == 4 and last == rank
        assert elapsed >= 3 * 0.02 * world - 1e-3    # every rank reports the slowest rank's time
        t = torch.tensor([elapsed], dtype=torch.float64)
        lo = t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        assert float(lo) == float(t)                 # identical on all ranks
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _fake_generator_counting(calls):
    def fn(mels):
        calls.append(int(mels.shape[0]))
        return _fake_generator(mels)
    return fn


def _worker_job(rank, world, port, B, T):
    """BASELINE config 5's shape on CPU: 512 utterances (tiny T) scattered from the root over 8 ranks, int16 sink,
    gathered back; a ragged batch where no rank may run a filler row; utterances of different lengths by length."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        mels = torch.rand(B, 80, T, generator=g)

        def sink(block):
            return (_fake_generator(block) * 50).to(torch.int16)
        out = parallel.synthesize_sharded(sink, mels if rank == 0 else None, scatter=True, device=torch.device("cpu"))
        if rank == 0:
            assert out.shape == (B, 3 * T) and out.dtype == torch.int16 and torch.equal(out, sink(mels))
        # ragged: B' rows over `world` ranks -- every rank runs exactly its own rows (no filler-row forwards)
        for Bp in (world + 3, 3, 1):
            calls = []
            out = parallel.synthesize_sharded(_fake_generator_counting(calls), mels[:Bp] if rank == 0 else None,
                                              scatter=True, device=torch.device("cpu"))
            lo, hi = parallel.shard_range(Bp, world, rank)
            assert calls == ([hi - lo] if hi > lo else []), (rank, Bp, calls)
            if rank == 0:
                assert torch.equal(out, _fake_generator(mels[:Bp]))
        # utterances of different lengths: longest-first assignment, results in utterance order on the root
        lens = [int(v) for v in torch.randint(3, 40, (3 * world + 1,), generator=torch.Generator().manual_seed(3))]
        ragged = [torch.rand(80, n, generator=torch.Generator().manual_seed(100 + i)) for i, n in enumerate(lens)]
        res = parallel.synthesize_ragged(_fake_generator, ragged)
        if rank == 0:
            assert len(res) == len(ragged)
            for r, m in zip(res, ragged):
                assert torch.equal(r, _fake_generator(m[None])[0])
        else:
            assert res is None
        # one flat broadcast for a whole module tree (parameters AND buffers, mixed dtypes)
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Conv1d(3, 4, 3), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 2))
        parallel.broadcast_weights(net, src=0)
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Conv1d(3, 4, 3), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 2))
        for (ka, a), (kb, b) in zip(sorted(net.state_dict().items()), sorted(ref.state_dict().items())):
            assert ka == kb and a.dtype == b.dtype and torch.equal(a, b), ka
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_edges(rank, world, port):
    """The corners round 4's review found: a root without rows, an empty batch, a module whose tensors do not pack to
    aligned offsets, and a forward_fn whose dtype the wire cannot carry (every rank must raise, none may hang)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mels = torch.rand(1, 5, 7, generator=torch.Generator().manual_seed(5))
        # B < world with dst = the rank that has NO rows (shard_range gives the only row to rank 0)
        for dst in range(world):
            out = parallel.synthesize_sharded(_fake_generator, mels, dst=dst)
            assert (torch.equal(out, _fake_generator(mels)) if rank == dst else out is None), (rank, dst)
            out = parallel.synthesize_sharded(lambda b: (_fake_generator(b) * 9).to(torch.int16),
                                              mels if rank == dst else None, dst=dst, scatter=True, device=torch.device("cpu"))
            assert (out.dtype == torch.int16 and out.shape == (1, 21)) if rank == dst else out is None
        # an empty batch: an empty result on the root, no collective left half-entered
        out = parallel.synthesize_sharded(_fake_generator, mels[:0])
        assert (out.shape[0] == 0) if rank == 0 else out is None
        dist.barrier()
        # mixed dtypes at unaligned byte offsets: 13 fp32 (52 bytes), then int64, bool, fp16, fp64
        class Odd(torch.nn.Module):
            def __init__(self, seed):
                super().__init__()
                g = torch.Generator().manual_seed(seed)
                self.a = torch.nn.Parameter(torch.rand(13, generator=g))
                self.register_buffer("n", torch.randint(0, 1 << 40, (3,), generator=g))
                self.register_buffer("m", torch.rand(5, generator=g) > 0.5)
                self.register_buffer("h", torch.rand(3, generator=g).to(torch.float16))
                self.register_buffer("d", torch.rand(2, generator=g).to(torch.float64))
        net, ref = Odd(rank + 1), Odd(1)
        parallel.broadcast_weights(net, src=0)
        for (ka, a), (kb, b) in zip(sorted(net.state_dict().items()), sorted(ref.state_dict().items())):
            assert ka == kb and a.dtype == b.dtype and torch.equal(a, b), ka
        # more wire dtypes: bf16 and int32 come back as they were produced
        for dt in (torch.bfloat16, torch.int32):
            fn = lambda b, dt=dt: (_fake_generator(b) * 64).to(dt)
            m3 = torch.rand(3, 5, 7, generator=torch.Generator().manual_seed(6))
            out = parallel.synthesize_sharded(fn, m3)
            assert (out.dtype == dt and torch.equal(out, fn(m3))) if rank == 0 else out is None
            res = parallel.synthesize_ragged(fn, [m3[0], m3[1][:, :4], m3[2][:, :6]])
            assert (res[1].dtype == dt and torch.equal(res[1], fn(m3[1][None, :, :4])[0])) if rank == 0 else res is None
        # a dtype outside the wire's list: ValueError on EVERY rank, after the collective (nobody hangs)
        bad = lambda b: torch.complex(_fake_generator(b), _fake_generator(b))
        with pytest.raises(ValueError, match="dtype"):
            parallel.synthesize_ragged(bad, [mels[0]])          # one utterance: only rank 0 runs forward_fn
        with pytest.raises(ValueError, match="dtype"):
            parallel.synthesize_sharded(bad, mels)              # B < world: only rank 0 has a row
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_pipelined(rank, world, port):
    """parallel.synthesize_pipelined (bench.py's job at N > 1): sub-batched, scatter of the next sub-batch and gather
    of the last one in flight under the current forward -- the rows must come back exactly as one process makes them,
    whatever the block / sub-batch split, and no rank may run a filler row."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(31)
        mels = torch.rand(6 * world + 5, 8, 6, generator=g)

        def sink(block):
            return (_fake_generator(block) * 50).to(torch.int16)
        cpu = torch.device("cpu")
        for B, sub in ((6 * world + 5, 2), (6 * world + 5, 4), (6 * world + 5, 100), (4 * world, 2), (world + 1, 1),
                       (world, 3), (world - 1, 2), (1, 2)):
            calls = []

            def fn(block):
                calls.append(int(block.shape[0]))
                return sink(block)
            out = parallel.synthesize_pipelined(fn, mels[:B] if rank == 0 else None, sub, device=cpu)
            lo, hi = parallel.shard_range(B, world, rank)
            per = -(-B // world)
            eff = min(sub, per)
            want_calls = [min(eff, hi - a) for a in range(lo, hi, eff)]
            assert calls == want_calls, (rank, B, sub, calls, want_calls)          # its own rows, in sub-batches, no filler
            if rank == 0:
                assert out.dtype == torch.int16 and torch.equal(out, sink(mels[:B])), (B, sub)
            else:
                assert out is None
        # fp32 rows travel as fp32; a root that is not rank 0
        out = parallel.synthesize_pipelined(_fake_generator, mels if rank == world - 1 else None, 3, dst=world - 1, device=cpu)
        assert (torch.equal(out, _fake_generator(mels)) if rank == world - 1 else out is None)
        # an empty job
        out = parallel.synthesize_pipelined(_fake_generator, mels[:0] if rank == 0 else None, 3, device=cpu)
        assert (out.shape[0] == 0) if rank == 0 else out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_pipelined_job_gloo(world):
    mp.spawn(_worker_pipelined, args=(world, _free_port()), nprocs=world, join=True)


@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_started_as_one_process_launches_its_own_ranks(gpus):
    """`python bench.py --gpus N` with no launcher around it (how the driver starts the N = 1 bench): it must become N
    ranks by itself.  `--launch-check` keeps the run on the host cores (gloo, stand-in forward) so that the launch, the
    rendezvous on 127.0.0.1 and the job's scatter / forward / gather pipeline are exercised where there is no GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--launch-check"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["launch_check"] and out["n_gpus"] == gpus and out["job_bit_identical_to_one_process"]


def test_bench_self_launch_command_is_the_drivers():
    import bench
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "3"], 29512)
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert argv[argv.index("--nproc-per-node") + 1] == "8" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "29512" and argv[-4:] == ["--gpus", "8", "--steps", "3"]
    assert os.path.basename(argv[-5]) == "bench.py"


@pytest.mark.parametrize("world", [2, 3])
def test_sharding_edge_cases_gloo(world):
    mp.spawn(_worker_edges, args=(world, _free_port()), nprocs=world, join=True)


def test_job_of_512_utterances_over_8_ranks_gloo():
    """BASELINE.json configs[4]'s control flow at full utterance count and world size (tiny T), on CPU."""
    mp.spawn(_worker_job, args=(8, _free_port(), 512, 4), nprocs=8, join=True)


def test_length_sorted_assignment_balances_ragged_batches():
    rng = np.random.RandomState(0)
    for world in (2, 3, 8):
        for n in (1, 5, 8, 40, 513):
            lens = rng.randint(50, 3000, size=n).tolist()
            plan = parallel.assign_by_length(lens, world)
            assert sorted(i for ix in plan for i in ix) == list(range(n))                 # a partition
            assert all(ix == sorted(ix) for ix in plan)
            loads = [sum(lens[i] for i in ix) for ix in plan]
            opt = max(max(lens), -(-sum(lens) // world))                                   # lower bound of any assignment
            assert max(loads) <= (4.0 / 3.0) * opt + 1, (world, n, max(loads), opt)
            assert plan == parallel.assign_by_length(lens, world)                          # deterministic
    # contiguous blocks of a length-sorted batch would put all the long utterances on one rank
    lens = list(range(100, 900, 100))
    loads = [sum(lens[i] for i in ix) for ix in parallel.assign_by_length(lens, 2)]
    assert abs(loads[0] - loads[1]) <= 100


@pytest.mark.parametrize("world,B", [(2, 5), (2, 4), (3, 7), (2, 1)])
def test_sharded_synthesis_gloo(tmp_path, world, B):
    port = _free_port()
    out = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(world, port, B, out), nprocs=world, join=True)
    got = np.load(out)
    g = torch.Generator().manual_seed(123)
    mels = torch.rand(B, 5, 7, generator=g)
    assert np.array_equal(got, _fake_generator(mels).numpy())


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 5, 8, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
