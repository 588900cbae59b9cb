"""Utterance sharding across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference is single-process / single-device: nothing
is translated here.  The generators have no cross-utterance op, so the batch
partitions exactly: contiguous blocks of utterances per rank, and the only
traffic is
  * once:      broadcast of the checkpoint tensors from rank 0  (14-55 MB)
  * per batch: gather of the waveforms to rank 0 -- a flat root gather, so each
               peer sends over its own direct xGMI link to the root (7 links
               used concurrently) instead of a ring bound by one link.
There is no all-reduce and no collective inside the generator.

Backends: "nccl" (RCCL) moves device tensors directly and needs one GPU per rank.  Under "gloo" device tensors are
staged through the host (gloo's scatter / gather take CPU tensors only): that is how several ranks can share ONE GPU
in tests (tests/test_gpu_multi.py: world_size 2 through the real generator on a 1-GPU box) and how the CPU tests run.
"""
import torch
import torch.distributed as dist


# what a forward_fn may return (wire code = index; new entries go to the END: the codes travel between ranks)
_DTYPES = [torch.float32, torch.int16, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64,
           torch.uint8, torch.int8]
_ALIGN = 16      # byte alignment of every tensor inside the flat broadcast buffer (>= the largest element size)


def _dtype_code(dtype):
    """Wire code of ``dtype``, or -2: not a dtype the gather can carry.  Never raises -- a rank that raised BEFORE a
    collective would leave the others waiting in it; the code travels and every rank raises after the collective."""
    return _DTYPES.index(dtype) if dtype in _DTYPES else -2


def _wire_bytes(t):
    """``t`` as its bytes (RCCL / gloo lack some element types -- int16, bf16 under gloo: everything travels as uint8)."""
    return t.contiguous().reshape(-1).view(torch.uint8)


def shard_range(n_items, world_size, rank):
    """Contiguous [lo, hi) block of ``n_items`` for ``rank``; blocks differ by at
    most one item and concatenate, in rank order, to range(n_items)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _staged(group, *tensors):
    """Device tensors have to travel through the host (gloo)?"""
    return dist.get_backend(group) == "gloo" and any(t is not None and t.is_cuda for t in tensors)


def broadcast_weights(model, src=0, group=None):
    """Broadcast every parameter and buffer of ``model`` from ``src`` in place -- as ONE collective: the tensors'
    bytes are concatenated into a single flat buffer (HiFi-GAN with weight norm: 234 tensors, 14-55 MB; one RCCL
    broadcast over xGMI instead of 234 launches of a few KB each), broadcast, and copied back."""
    tensors = list(model.parameters()) + list(model.buffers())
    if tensors:
        with torch.no_grad():
            dev = tensors[0].device
            staged = _staged(group, *tensors)
            # every tensor starts on a 16-byte boundary of the flat buffer: `.view(dtype)` of a byte range needs an
            # offset that is a multiple of the element size (a module with mixed dtypes -- BatchNorm's int64
            # num_batches_tracked behind an odd count of fp32 elements, bool / fp16 buffers -- would otherwise raise on
            # every rank)
            sizes = [t.numel() * t.element_size() for t in tensors]
            offs, total = [], 0
            for n in sizes:
                offs.append(total)
                total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
            flat = torch.zeros(total, dtype=torch.uint8, device="cpu" if staged else dev)
            if dist.get_rank(group) == src:
                for t, off, n in zip(tensors, offs, sizes):
                    flat[off:off + n].copy_(_wire_bytes(t.data))
            dist.broadcast(flat, src=src, group=group)
            for t, off, n in zip(tensors, offs, sizes):
                t.data.copy_(flat[off:off + n].view(t.dtype).reshape(t.shape))
    if hasattr(model, "invalidate_plans"):
        for m in model.modules():
            if hasattr(m, "invalidate_plans"):
                m.invalidate_plans()
    return model


def assign_by_length(lengths, world_size):
    """Ragged batches (utterances of different length): which utterances each rank synthesises.  Longest first, each
    to the rank with the least work so far (longest-processing-time-first; the generator's cost is linear in the
    frame count), ties to the lower rank: deterministic on every rank.  Returns ``world_size`` index lists, each in
    ascending utterance order; the bound is the classic 4/3 - 1/(3 world) of the optimal makespan."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(ix) for ix in out]


def synthesize_ragged(forward_fn, mels, world_size=None, rank=None, dst=0, group=None):
    """Utterances of DIFFERENT lengths: ``mels`` is a list of [C, T_i] tensors (every rank holds the list);
    rank r runs ``forward_fn(mel[None]) -> [1, n_i]`` on the utterances :func:`assign_by_length` gives it -- no filler
    rows, no padded frames -- and the waveforms come back to ``dst`` as a list in utterance order (None elsewhere).
    Traffic: a tiny gather of the sample counts, a one-word all-reduce (the longest rank's byte total: the gather wants
    equal buffers), one root gather of every rank's concatenated waveforms."""
    world_size = dist.get_world_size(group) if world_size is None else world_size
    rank = dist.get_rank(group) if rank is None else rank
    plan = assign_by_length([m.shape[-1] for m in mels], world_size)
    outs = [forward_fn(mels[i][None].contiguous())[0].contiguous() for i in plan[rank]]
    dev = mels[0].device
    wire_dev = torch.device("cpu") if (dist.get_backend(group) == "gloo" and dev.type == "cuda") else dev
    # [sample count of each of this rank's utterances ..., dtype code (or -1: nothing ran here)] -> root
    slots = max(len(ix) for ix in plan) + 1
    meta = torch.zeros(slots, dtype=torch.int64, device=wire_dev)
    for j, o in enumerate(outs):
        meta[j] = o.numel()
    code = _dtype_code(outs[0].dtype) if outs else -1
    meta[slots - 1] = code
    metas = [torch.empty_like(meta) for _ in range(world_size)] if rank == dst else None
    dist.gather(meta, gather_list=metas, dst=dst, group=group)
    # [this rank's byte total, 1 if its forward_fn returned a dtype the wire cannot carry] -> MAX over ranks: every rank
    # learns of the error inside the collective and raises after it, none is left waiting in the gather below
    total = torch.tensor([sum(o.numel() * o.element_size() for o in outs), 1 if code == -2 else 0], dtype=torch.int64,
                         device=wire_dev)
    dist.all_reduce(total, op=dist.ReduceOp.MAX, group=group)
    if int(total[1]):
        raise ValueError("synthesize_ragged: forward_fn returned a dtype the gather cannot carry"
                         + (f" ({outs[0].dtype})" if code == -2 else " (on another rank)")
                         + f"; supported: {', '.join(str(d) for d in _DTYPES)}")
    wire = torch.zeros(max(int(total[0]), 1), dtype=torch.uint8, device=wire_dev)
    off = 0
    for o in outs:
        b = o.reshape(-1).view(torch.uint8).to(wire_dev)
        wire[off:off + b.numel()] = b
        off += b.numel()
    bufs = [torch.empty_like(wire) for _ in range(world_size)] if rank == dst else None
    dist.gather(wire, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    result = [None] * len(mels)
    for r in range(world_size):
        code = int(metas[r][slots - 1])
        if code < 0:
            continue
        dtype = _DTYPES[code]
        esize = torch.empty(0, dtype=dtype).element_size()
        off = 0
        for j, i in enumerate(plan[r]):
            nbytes = int(metas[r][j]) * esize
            result[i] = bufs[r][off:off + nbytes].clone().view(dtype).to(dev)
            off += nbytes
    return result


class WaveformGather:
    """Root gather of equally-shaped per-rank waveform blocks [b, n] -> [world*b, n] on rank 0
    (None elsewhere).  Buffers are allocated once and reused.

    The gather is issued asynchronously (on the communicator's own stream) and only the
    PREVIOUS one is waited for when the next is issued, so the transfer of batch i overlaps
    the generator's work on batch i+1 -- the generator itself has no collective to wait for.
    ``flush()`` completes the last one; the list returned by a call is valid after the next
    call or ``flush()``."""

    def __init__(self, world_size, rank, device, dst=0, group=None):
        self.world, self.rank, self.dst, self.group = world_size, rank, dst, group
        self.device = device
        self._bufs = None
        self._shape = None
        self._pending = None     # (work handle, tensor kept alive while in flight)

    def flush(self):
        if self._pending is not None:
            self._pending[0].wait()
            self._pending = None
        return self._bufs

    def __call__(self, wav):
        wav = wav.contiguous()
        if _staged(self.group, wav):
            wav = wav.cpu()      # (gloo: through the host; the copy waits for the forward)
        self.flush()             # the receive buffers are about to be reused
        if self.rank == self.dst:
            if self._shape != tuple(wav.shape):
                self._bufs = [torch.empty_like(wav) for _ in range(self.world)]
                self._shape = tuple(wav.shape)
            work = dist.gather(wav, gather_list=self._bufs, dst=self.dst, group=self.group, async_op=True)
            self._pending = (work, wav)
            return self._bufs
        work = dist.gather(wav, gather_list=None, dst=self.dst, group=self.group, async_op=True)
        self._pending = (work, wav)
        return None


def synthesize_sharded(forward_fn, mels, world_size=None, rank=None, dst=0, group=None, scatter=False,
                       device=None):
    """Run ``forward_fn(block) -> [rows, n]`` on this rank's contiguous block of the batch
    ``mels [B, C, T]`` and gather all results, in utterance order, on ``dst``.

    ``scatter=False``: every rank already holds ``mels`` (or makes its own) and slices its block.
    ``scatter=True``: only ``dst`` holds ``mels`` (others pass None): its shape is broadcast and the
    blocks are scattered from ``dst`` -- with the gather that is the whole traffic of a job (SURVEY.md
    section 8e: scatter of mels, gather of waveforms, nothing in between).  ``device``: where the
    received block lives (default: ``mels``' device on dst, required on the other ranks).
    Ragged blocks (B not divisible by the world size) are padded ON THE WIRE only and trimmed on the root: no rank
    runs the generator on a filler row.  The result
    keeps ``forward_fn``'s dtype (fp32 waveforms, or int16 after a GPU wav sink).  Returns [B, n] on
    ``dst`` and None elsewhere."""
    world_size = dist.get_world_size(group) if world_size is None else world_size
    rank = dist.get_rank(group) if rank is None else rank
    if scatter:
        device = mels.device if mels is not None else device
        staged = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
        wire_dev = "cpu" if staged else device
        shape = torch.tensor(list(mels.shape) if rank == dst else [0, 0, 0], dtype=torch.int64, device=wire_dev)
        dist.broadcast(shape, src=dst, group=group)
        B, C, T = (int(v) for v in shape.tolist())
    else:
        B = mels.shape[0]
        staged = _staged(group, mels)
    lo, hi = shard_range(B, world_size, rank)
    per = (B + world_size - 1) // world_size
    if scatter:
        block = torch.empty((per, C, T), dtype=torch.float32, device=wire_dev)
        parts = None
        if rank == dst:
            parts = []
            for r in range(world_size):
                a, b = shard_range(B, world_size, r)
                part = mels[a:b]
                if b - a < per:
                    filler = part[-1:] if b > a else torch.zeros_like(mels[:1])
                    part = torch.cat([part] + [filler] * (per - (b - a)), dim=0)
                parts.append(part.contiguous().to(wire_dev))
        dist.scatter(block, scatter_list=parts, src=dst, group=group)
        block = block.to(device)
    else:
        block = mels[lo:hi]
        if hi - lo < per:   # pad with a copy of the last row (or a zero row for an empty block)
            filler = block[-1:] if hi > lo else torch.zeros_like(mels[:1])
            block = torch.cat([block] + [filler] * (per - (hi - lo)), dim=0)
    # Only this rank's REAL rows go through the generator (the filler rows of a ragged batch exist on the wire only:
    # scatter and gather want equal blocks); a rank without rows learns the row shape from the root.
    rows_here = hi - lo
    if B == 0:                           # nothing to do on any rank (every rank knows B): no collective, an empty result
        return torch.zeros((0, 0), dtype=torch.float32, device=block.device) if rank == dst else None
    wav = forward_fn(block[:rows_here].contiguous()).contiguous() if rows_here > 0 else None
    if B < world_size:
        # Some rank has nothing to run and cannot know the row shape / dtype.  The ranks that DID run something tell the
        # others -- MAX over ranks of [samples per row, dtype code + 1, error flag]; a rank without rows contributes
        # zeros.  (Not a broadcast from dst: dst itself may be a rank without rows.)
        code = _dtype_code(wav.dtype) if wav is not None else -1
        info = torch.tensor([wav.shape[1] if wav is not None else 0, max(code, -1) + 1, 1 if code == -2 else 0],
                            dtype=torch.int64, device="cpu" if dist.get_backend(group) == "gloo" else block.device)
        dist.all_reduce(info, op=dist.ReduceOp.MAX, group=group)
        if int(info[2]):
            raise ValueError("synthesize_sharded: forward_fn returned a dtype the gather cannot carry"
                             + (f" ({wav.dtype})" if code == -2 else " (on another rank)")
                             + f"; supported: {', '.join(str(d) for d in _DTYPES)}")
        if wav is None:
            wav = torch.zeros((0, int(info[0])), dtype=_DTYPES[int(info[1]) - 1], device=block.device)
    if wav.shape[0] < per:
        wav = torch.cat([wav, torch.zeros((per - wav.shape[0], wav.shape[1]), dtype=wav.dtype, device=wav.device)], dim=0)
    # RCCL / gloo have no 16-bit integer type (and gloo no bf16): anything but fp32 / fp64 / fp16 travels as its bytes
    wire = wav if wav.dtype in (torch.float32, torch.float64, torch.float16) else wav.view(torch.uint8)
    if staged:
        wire = wire.cpu()
    bufs = [torch.empty_like(wire) for _ in range(world_size)] if rank == dst else None
    dist.gather(wire, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    rows = []
    for r in range(world_size):
        a, b = shard_range(B, world_size, r)
        rows.append(bufs[r].view(wav.dtype)[: b - a])
    return torch.cat(rows, dim=0).to(wav.device)


def synthesize_pipelined(forward_fn, mels, sub, world_size=None, rank=None, dst=0, group=None, device=None):
    """A job in sub-batches with its traffic UNDER its compute: only ``dst`` holds ``mels [B, C, T]`` (others pass
    None); every rank owns the contiguous block :func:`shard_range` gives it and walks it ``sub`` utterances at a
    time.  While sub-batch i runs through ``forward_fn(block [rows, C, T]) -> [rows, n]``, the scatter of sub-batch
    i + 1 and the gather of sub-batch i - 1 are in flight (asynchronous root scatter / gather on the communicator's
    own stream: each peer over its own xGMI link to the root, nothing ring-shaped) -- the serial terms that are left
    are the first scatter and the last gather.  Every rank issues the same sequence of collectives
    (scatter 0, scatter 1, gather 0, scatter 2, gather 1, ...).  Filler rows exist on the wire only; the rows come
    back in utterance order in ``forward_fn``'s dtype, bit-identical to one process running the same sub-batches
    (:func:`synthesize_sharded` is the unpipelined form: one scatter, one forward_fn call, one gather).
    Returns [B, n] on ``dst`` and None elsewhere."""
    world_size = dist.get_world_size(group) if world_size is None else world_size
    rank = dist.get_rank(group) if rank is None else rank
    sub = max(int(sub), 1)
    device = mels.device if mels is not None else device
    staged = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
    wire_dev = torch.device("cpu") if staged else torch.device(device)
    shape = torch.tensor(list(mels.shape) if rank == dst else [0, 0, 0], dtype=torch.int64, device=wire_dev)
    dist.broadcast(shape, src=dst, group=group)
    B, C, T = (int(v) for v in shape.tolist())
    if B == 0:
        return torch.zeros((0, 0), dtype=torch.float32, device=device) if rank == dst else None
    per = (B + world_size - 1) // world_size
    sub = min(sub, per)
    chunks = (per + sub - 1) // sub
    lo, hi = shard_range(B, world_size, rank)

    def rows_of(r, c):
        """[a, b) = the utterances of rank r's sub-batch c (empty past the end of its block)"""
        a0, b0 = shard_range(B, world_size, r)
        a = min(a0 + c * sub, b0)
        return a, min(a + sub, b0)

    def issue_scatter(c):
        block = torch.empty((sub, C, T), dtype=torch.float32, device=wire_dev)
        parts = None
        if rank == dst:
            parts = []
            for r in range(world_size):
                a, b = rows_of(r, c)
                part = mels[a:b]
                if b - a < sub:
                    part = torch.cat([part, torch.zeros((sub - (b - a), C, T), dtype=mels.dtype, device=mels.device)], dim=0)
                parts.append(part.contiguous().to(wire_dev))
        work = dist.scatter(block, scatter_list=parts, src=dst, group=group, async_op=True)
        return work, block, parts          # (the send buffers stay alive while the scatter is in flight)

    meta = None                            # (samples per row, dtype) once a forward has run here
    gathers = []                           # per sub-batch: (work, receive buffers, wire tensor)
    nxt = issue_scatter(0)
    for c in range(chunks):
        cur, nxt = nxt, (issue_scatter(c + 1) if c + 1 < chunks else None)
        cur[0].wait()
        a, b = rows_of(rank, c)
        block = cur[1].to(device) if staged else cur[1]
        wav = forward_fn(block[: b - a].contiguous()).contiguous() if b > a else None
        if meta is None:
            # the row shape / dtype of the job: known to every rank that ran something; with B < world some rank ran
            # nothing and learns it from the others (MAX over ranks -- the only case with a collective besides
            # scatter / gather, and one word)
            if B < world_size:
                code = _dtype_code(wav.dtype) if wav is not None else -1
                info = torch.tensor([wav.shape[1] if wav is not None else 0, max(code, -1) + 1, 1 if code == -2 else 0],
                                    dtype=torch.int64, device=wire_dev)
                dist.all_reduce(info, op=dist.ReduceOp.MAX, group=group)
                if int(info[2]):
                    raise ValueError("synthesize_pipelined: forward_fn returned a dtype the gather cannot carry")
                meta = (int(info[0]), _DTYPES[int(info[1]) - 1])
            else:
                if _dtype_code(wav.dtype) == -2:
                    raise ValueError(f"synthesize_pipelined: forward_fn returned {wav.dtype}, which the gather cannot carry")
                meta = (int(wav.shape[1]), wav.dtype)
        n, dtype = meta
        if wav is None:
            wav = torch.zeros((sub, n), dtype=dtype, device=device)
        elif wav.shape[0] < sub:
            wav = torch.cat([wav, torch.zeros((sub - wav.shape[0], n), dtype=dtype, device=wav.device)], dim=0)
        wire = wav if dtype in (torch.float32, torch.float64, torch.float16) else wav.view(torch.uint8)
        if staged:
            wire = wire.cpu()
        bufs = [torch.empty_like(wire) for _ in range(world_size)] if rank == dst else None
        gathers.append((dist.gather(wire, gather_list=bufs, dst=dst, group=group, async_op=True), bufs, wire))
    for work, _, _ in gathers:
        work.wait()
    if rank != dst:
        return None
    n, dtype = meta
    out = torch.empty((B, n), dtype=dtype, device=device)
    for c, (_, bufs, _) in enumerate(gathers):
        for r in range(world_size):
            a, b = rows_of(r, c)
            if b > a:
                out[a:b].copy_(bufs[r].view(dtype)[: b - a])
    return out
