"""ctypes binding of libfastvocoder_hip.so (include/fastvocoder_hip.h).

PyTorch-ROCm is plumbing here: it owns device memory (tensors) and the stream;
every function below passes raw device pointers and the current HIP stream to
the C ABI.  There is NO fallback: if the shared library is missing or a tensor
is not a contiguous fp32 tensor on a ROCm device, these raise.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfastvocoder_hip.so")
_CSRC = os.path.join(_HERE, "csrc")
# conv_inst_s*.hip instantiate the conv kernel templates (conv_kernels.hpp) one tile shape each,
# so that the ~170 kernel variants compile in parallel
SOURCES = ["conv_mfma.hip", "api.hip", "plan.hip", "pack.hip", "pqmf.hip", "wav_sink.hip", "conv_inst_narrow.hip",
           "pair_launch.hip", "pair_inst_c16.hip", "pair_inst_c32.hip", "pairh_inst_c16.hip", "pairh_inst_c32.hip",
           "convh_launch.hip", "convh_inst_c64.hip", "convh_inst_c128.hip", "convt_inst.hip",
           "convg_inst.hip", "convr_inst.hip", "convtn_inst.hip", "convk_inst.hip", "convq2_inst.hip",
           "convq3_inst.hip", "mrfh_launch.hip", "mrfh_inst_a.hip", "mrfh_inst_b.hip", "mrfw_inst.hip", "convtl_inst.hip", "convs2_inst.hip", "convu2_inst.hip"] + \
          [f"conv_inst_s{i}.hip" for i in range(6)]
HEADERS = ["fv_internal.h", "conv_kernels.hpp", "pair_kernels.hpp", "pair_inst.hpp", "pairh_kernels.hpp",
           "pairh_inst.hpp", "convh_kernels.hpp", "convh_inst.hpp", "convr_kernels.hpp",
           "convtn_kernels.hpp", "convk_kernels.hpp", "convq2_kernels.hpp", "convq3_kernels.hpp", "mrfh_kernels.hpp", "mrfh_inst.hpp", "mrfw_kernels.hpp", "convtl_kernels.hpp", "convs2_kernels.hpp", "convu2_kernels.hpp", "api_internal.h"]

PAD_ZERO, PAD_REFLECT = 0, 1
PAD_CAUSAL = 2      # flag: pad (k-1)*dil on both sides, keep the first Tin outputs (CausalConv1d)
POST_NONE, POST_TANH, POST_RELU = 0, 1, 2
SLOT_NONE, SLOT_IN, SLOT_OUT, SLOT_TMP0, MAX_SLOTS = -1, 0, 1, 2, 32
SLOT_AUX_IN0, SLOT_AUX_IN1, SLOT_OUT2 = 28, 29, 30    # caller-provided tensors of Plan.run(aux=..., out2=...)
ABI_VERSION = 13
PAIR_F32, PAIR_SPLIT_F16 = 0, 1   # arithmetic of the fused ResBlock-pair kernels (fastvocoder_hip.h)


ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -2, -3
ERR_RANGE = -4                    # fv_plan_check_range: a split-f16 kernel met an operand beyond the f16 range
GUARD_HIGH, GUARD_LOW = 0x001, 0x100   # the two sides of a guard word (separate bytes: fastvocoder_hip.h)
ERR_RANGE_LOW = -5                # ... only the low-side guard fired (a block's share of a tensor was small as a whole)


class NativeError(RuntimeError):
    pass


class RangeError(NativeError):
    """A weight or an activation lies outside the domain of the split-f16 kernels (|v| < 65520, the f16 range;
    include/fastvocoder_hip.h FV_PAIR_SPLIT_F16): the computation has to be repeated with PAIR_F32 arithmetic.
    engine.NativeModule does that by itself; the exception reaches a caller only from the single-operator
    functions of this module."""


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
              "-Wno-unused-value", "-Wno-comment", "-Wno-pass-failed",
              # MFMA results straight in VGPRs (unified register file on gfx950): no
              # v_accvgpr_read/write pairs around every stage -- VALU work costs MFMA time
              "-mllvm", "-amdgpu-mfma-vgpr-form=1",
              # the device code objects compressed inside the fat binary (zstd; the HIP runtime inflates them when the library
              # is loaded): the .so is 2.4 MB instead of 10.9
              "--offload-compress"]


def source_hash():
    """sha256 (first 16 hex digits) over every source of the library and the extra compiler flags: the
    build id the .so carries (fv_build_id), so that a binary can be proven to be built from this tree."""
    import hashlib
    h = hashlib.sha256()
    paths = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(_CSRC, x) for x in HEADERS] + \
        [os.path.join(_HERE, "..", "include", "fastvocoder_hip.h")]
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(BASE_FLAGS[4:]).encode())      # (the flags that were added after round 4: older ids stay comparable)
    h.update(os.environ.get("FV_HIPCC_FLAGS", "").encode())
    return h.hexdigest()[:16]


def built_id():
    """Build id of the .so on disk, or None when it is missing / predates build ids.  Read from the file's bytes
    (the "fv-build-id:" tag in front of the string fv_build_id() returns), not through dlopen: a library this
    process has already loaded would answer for the OLD image after a rebuild."""
    if not os.path.exists(LIB_PATH):
        return None
    with open(LIB_PATH, "rb") as f:
        data = f.read()
    at = data.find(b"fv-build-id:")
    if at < 0:
        return None
    end = data.find(b"\0", at)
    return data[at + 12:end].decode(errors="replace") if 0 < end - at - 12 <= 64 else None


def _local_includes(path, seen=None):
    """Transitive closure of the `#include "..."` files of a source (paths relative to the including file)."""
    import re
    seen = set() if seen is None else seen
    with open(path) as f:
        text = f.read()
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        dep = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if dep not in seen and os.path.exists(dep):
            seen.add(dep)
            _local_includes(dep, seen)
    return seen


def _object_stamp(src, flags):
    """Hash of everything an object file depends on: its source, the headers it includes (transitively), the flags."""
    import hashlib
    h = hashlib.sha256()
    for path in [src] + sorted(_local_includes(src)):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """hipcc the kernels for gfx950 into fastvocoder_amd/libfastvocoder_hip.so
    (cross-compiles without a GPU): one object per source, compiled in parallel, then linked.
    Up to date <=> the library's embedded build id equals the hash of the sources.  Objects are rebuilt only when their
    own source, a header they include or the flags changed (a stamp file next to each object); the build id is compiled
    into api.hip alone."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    want = source_hash()
    if not force and built_id() == want:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = BASE_FLAGS + os.environ.get("FV_HIPCC_FLAGS", "").split()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        own = flags + ([f'-DFV_BUILD_ID="{want}"'] if os.path.basename(src) == "api.hip" else [])
        stamp, stamp_path = _object_stamp(src, own), obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp_path):
            with open(stamp_path) as f:
                if f.read() == stamp:
                    continue
        cmd = [hipcc] + own + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        if os.path.exists(stamp_path):
            os.remove(stamp_path)
        jobs.append((cmd, stamp_path, stamp, subprocess.Popen(cmd)))
    for cmd, stamp_path, stamp, proc in jobs:
        if proc.wait() != 0:
            for _, _, _, other in jobs:
                if other.poll() is None:
                    other.kill()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(stamp_path, "w") as f:
            f.write(stamp)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


_lib = None


def lib():
    """Load the library (never builds implicitly: a GPU box gets the prebuilt .so)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). fastvocoder_amd has no CPU or eager fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i, f, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
    L.fv_version.restype = i
    L.fv_last_error.restype = ctypes.c_char_p
    L.fv_build_id.restype = ctypes.c_char_p
    L.fv_fold_weight_norm.argtypes = [vp, vp, vp, i, i64, vp]
    L.fv_packed_conv1d_floats.argtypes = [i, i, i]
    L.fv_packed_conv1d_floats.restype = i64
    L.fv_packed_conv_transpose1d_floats.argtypes = [i, i, i, i, i]
    L.fv_packed_conv_transpose1d_floats.restype = i64
    L.fv_pack_conv1d_weight.argtypes = [vp, vp, i, i, i, vp]
    L.fv_pack_conv_transpose1d_weight.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.fv_conv1d_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, f, i, f, vp]
    L.fv_conv_transpose1d_fused.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, f, vp]
    L.fv_packed_basis_floats.argtypes = [i, i]
    L.fv_packed_basis_floats.restype = i64
    L.fv_pack_basis.argtypes = [vp, vp, i, i, vp]
    L.fv_basis_ola.argtypes = [vp, vp, vp, i, i, i, i, vp]
    L.fv_generator_run.argtypes = [vp, i, i, vp, vp, vp, i64, vp]
    L.fv_packed_conv_transpose1d_split_floats.argtypes = [i, i, i, i]
    L.fv_packed_conv_transpose1d_split_floats.restype = i64
    L.fv_pack_conv_transpose1d_split_f16.argtypes = [vp, vp, i, i, i, i, vp, vp]
    L.fv_conv_transpose1d_split_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, f, vp, vp]
    L.fv_plan_add_conv_transpose1d_split_f16.argtypes = [vp, i, i, i, vp, vp, i, i, i, i, i, i, f, f]
    L.fv_plan_set_input_merge.argtypes = [vp, i, i, f]
    L.fv_div_probe.argtypes = [ctypes.c_uint, i64, f, vp, vp]
    L.fv_pqmf_synthesis.argtypes = [vp, vp, vp, i, i, i, i, vp]
    L.fv_conv1d_2src_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, f, vp]
    L.fv_plan_add_conv1d_2src.argtypes = [vp, i, i, i, i, i, vp, vp, i, i, i, i, f]
    L.fv_packed_conv1x1_2src_split_floats.argtypes = [i]
    L.fv_packed_conv1x1_2src_split_floats.restype = i64
    L.fv_pack_conv1x1_2src_split_f16.argtypes = [vp, vp, vp, i, vp, vp]
    L.fv_conv1x1_2src_split_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, f, i, f, vp, vp]
    L.fv_plan_add_conv1x1_2src_split_f16.argtypes = [vp, i, i, i, i, i, vp, vp, i, f, i, f]
    L.fv_packed_residual_stack_floats.argtypes = [i, i]
    L.fv_packed_residual_stack_floats.restype = i64
    L.fv_pack_residual_stack_split_f16.argtypes = [vp, vp, vp, vp, i, i, vp, vp]
    L.fv_residual_stack_split_f16.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, i, i, f, vp, vp]
    L.fv_plan_add_residual_stack_split_f16.argtypes = [vp, i, i, i, vp, vp, vp, i, i, i, f, i, i, f]
    L.fv_plan_set_stack_two_launch.argtypes = [vp, i, vp, vp]
    L.fv_conv_post_pqmf.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, f, i, i, vp]
    L.fv_plan_add_conv_post_pqmf.argtypes = [vp, i, i, vp, vp, i, i, i, i, f, i, vp, i]
    L.fv_plan_set_sum_order.argtypes = [vp, i]
    L.fv_plan_add_conv1d_sum3.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i), i, i,
                                          ctypes.POINTER(vp), vp,
                                          i, ctypes.POINTER(i), f, i, f]
    L.fv_encode_16bits.argtypes = [vp, vp, vp, i, i64, f, i, vp]
    L.fv_pqmf_analysis.argtypes = [vp, vp, vp, i, i, i, i64, vp]
    L.fv_fold_batchnorm_conv.argtypes = [vp, vp, vp, vp, vp, vp, f, vp, vp, i, i, i, vp]
    L.fv_packed_upsample_conv1d_floats.argtypes = [i, i, i, i, i]
    L.fv_packed_upsample_conv1d_floats.restype = i64
    L.fv_pack_upsample_conv1d_weight.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.fv_upsample_conv1d_fused.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f, i, f, vp]
    L.fv_plan_add_upsample_conv1d.argtypes = [vp, i, i, i, vp, vp, i, i, i, i, i, f, i, f]
    pp = ctypes.POINTER(vp)
    L.fv_packed_pair_floats.argtypes = [i, i]
    L.fv_packed_pair_floats.restype = i64
    L.fv_pack_pair_weight.argtypes = [vp, vp, i, i, vp]
    L.fv_resblock1_fused.argtypes = [i, pp, pp, pp, pp, pp, pp, pp, ctypes.POINTER(i), i, i, i, i, f, f, vp]
    L.fv_mrf_stage.argtypes = [pp, pp, pp, pp, pp, vp, vp, ctypes.POINTER(i), i, i, i, i, f, f, i, f, vp]
    L.fv_plan_add_resblock_pair.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, i, i, f, f]
    L.fv_packed_pair_floats_ex.argtypes = [i, i, i]
    L.fv_packed_pair_floats_ex.restype = i64
    L.fv_pack_pair_weight_ex.argtypes = [vp, vp, i, i, i, vp, vp]
    L.fv_resblock1_fused_ex.argtypes = [i, pp, pp, pp, pp, pp, pp, pp, pp, pp, pp, ctypes.POINTER(i), i, i, i, i, f, f,
                                        i, f, i, vp, vp]
    L.fv_plan_add_resblock_pair_ex.argtypes = [vp, i, i, i, i, i, i, vp, vp, vp, vp, i, i, i, f, f, i, f, i]
    L.fv_conv1d_split_f16.argtypes = [i, pp, pp, pp, pp, pp, pp, pp, pp, ctypes.POINTER(i), i, i, i, i, i, f, f, i, f, vp, vp]
    L.fv_plan_set_pair_output_conv.argtypes = [vp, vp, vp, i, f, i]
    L.fv_packed_mrf_stage_floats.argtypes = [i, ctypes.POINTER(i)]
    L.fv_packed_mrf_stage_floats.restype = i64
    L.fv_pack_mrf_stage_split_f16.argtypes = [pp, pp, pp, pp, vp, i, ctypes.POINTER(i), vp, vp]
    L.fv_mrf_stage_workspace_bytes.argtypes = [i]
    L.fv_mrf_stage_workspace_bytes.restype = i64
    L.fv_mrf_stage_split_f16.argtypes = [vp, vp, vp, vp, i, i, i, ctypes.POINTER(i), ctypes.POINTER(i), f, f, i, f, vp, vp, vp,
                                         vp, i64, vp, vp]
    L.fv_plan_add_mrf_stage_split_f16.argtypes = [vp, i, i, i, vp, i, ctypes.POINTER(i), ctypes.POINTER(i), f, f, i, f, vp, i64]
    L.fv_plan_add_conv1d_split_f16.argtypes = [vp, i, i, i, i, i, i, vp, vp, i, i, i, i, f, f, i, f]
    L.fv_plan_add_mrf_sum.argtypes = [vp, ctypes.POINTER(i), i, i, pp, pp, pp, pp, i, ctypes.POINTER(i), i, f, f, i, f]
    L.fv_plan_create.argtypes = [i]
    L.fv_plan_create.restype = vp
    L.fv_plan_destroy.argtypes = [vp]
    L.fv_plan_destroy.restype = None
    L.fv_plan_add_conv1d.argtypes = [vp, i, i, i, i, i, i, vp, vp, i, i, i, i, i, i, f, f, i, f]
    L.fv_plan_add_conv_transpose1d.argtypes = [vp, i, i, i, vp, vp, i, i, i, i, i, i, f, i, f]
    L.fv_plan_add_pqmf_synthesis.argtypes = [vp, i, i, vp, i, i]
    L.fv_plan_set_guard.argtypes = [vp, vp]
    L.fv_plan_check_range.argtypes = [vp, vp]
    L.fv_tuning_set.argtypes = [ctypes.c_char_p, i]
    L.fv_debug_pair_schedule.argtypes = [i, ctypes.POINTER(i), ctypes.POINTER(i), i, i, i, ctypes.POINTER(ctypes.c_uint)]
    L.fv_plan_set_group.argtypes = [vp, i]
    L.fv_plan_output_shape.argtypes = [vp, i, ctypes.POINTER(i), ctypes.POINTER(i64)]
    L.fv_plan_workspace_bytes.argtypes = [vp, i, i]
    L.fv_plan_workspace_bytes.restype = i64
    L.fv_plan_run.argtypes = [vp, i, i, vp, vp, vp, i64, vp]
    L.fv_plan_run_aux.argtypes = [vp, i, i, vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i), vp, i64, vp]
    L.fv_plan_set_output_offset.argtypes = [vp, i, i]
    L.fv_plan_slot_shape.argtypes = [vp, i, i, ctypes.POINTER(i), ctypes.POINTER(i64)]
    L.fv_plan_num_ops.argtypes = [vp]
    L.fv_profile_enable.argtypes = [i]
    L.fv_profile_bracket_cost.argtypes = [vp, i, ctypes.POINTER(ctypes.c_double)]
    L.fv_profile_mfma_f16_rate.argtypes = [vp, i64, i, i, vp, ctypes.POINTER(ctypes.c_double)]
    L.fv_profile_collect.argtypes = [i, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    if L.fv_version() != ABI_VERSION:
        raise NativeError(f"ABI mismatch: library reports {L.fv_version()}, binding expects {ABI_VERSION}")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise NativeError(f"libfastvocoder_hip error {rc}: {lib().fv_last_error().decode()}")


def _ptr(t, name="tensor", allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise NativeError(f"{name} is None")
    if not t.is_cuda:
        raise NativeError(f"{name} lives on {t.device}; the HIP kernels need a ROCm device "
                          "tensor (there is no CPU path in fastvocoder_amd)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise NativeError(f"{name} must be contiguous float32, got {t.dtype} "
                          f"contiguous={t.is_contiguous()}")
    return t.data_ptr()


class _on:
    """Context for one native call: makes the operands' device current (the kernels launch on the
    CURRENT device; after ``model.to('cuda:1')`` that need not be the tensors') and yields that
    device's current stream.  All tensor operands must live on one device."""

    def __init__(self, *tensors):
        devs = {t.device for t in tensors if t is not None}
        if len(devs) != 1:
            raise NativeError(f"operands of a native call must share one ROCm device, got {sorted(map(str, devs))}")
        self.dev = devs.pop()
        if self.dev.type != "cuda":
            raise NativeError(f"operands live on {self.dev}; the HIP kernels need a ROCm device tensor "
                              "(there is no CPU path in fastvocoder_amd)")
        self._guard = torch.cuda.device(self.dev)

    def __enter__(self):
        self._guard.__enter__()
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        return self._guard.__exit__(*exc)


class GuardWord:
    """Two int32 words in pinned, device-mapped host memory: [0] raised by the split-f16 kernels of a run when an
    activation left the f16 range (fv_plan_set_guard), [1] by the pack kernels when a weight does.  The kernels write
    them through the host pointer; the host reads them after the stream has drained (or, lazily, whenever)."""

    def __init__(self):
        self.t = torch.zeros(2, dtype=torch.int32).pin_memory()

    def ptr(self, i=0):
        return self.t.data_ptr() + 4 * i

    def peek(self, i=0):
        """The word as it is now (no synchronisation)."""
        return int(self.t[i])

    def clear(self, i=0):
        self.t[i] = 0


def tuning_set(key, value):
    """Test / tuning hook (fv_tuning_set): one of the launchers' switches, process-wide."""
    check(lib().fv_tuning_set(key.encode(), int(value)))


def debug_pair_schedule(n_items, cost, nblk, mode=0, three_members=False):
    """Test hook (fv_debug_pair_schedule, host only): the block schedule of a fused-pair launch.  Returns
    ``(sched_on, shares)``: sched_on 1 -- ``shares[b][m] = (lo, count)``, member m's items of block b (pair_schedule);
    2 -- ``shares[i]`` = first item of share i in the members' concatenated item sequence (pair_cut_schedule); 0 -- no
    table for this shape (``shares`` is None)."""
    n = len(n_items)
    arr = ctypes.c_int * n
    table = (ctypes.c_uint * 512)()
    rc = lib().fv_debug_pair_schedule(n, arr(*[int(v) for v in n_items]), arr(*[int(v) for v in cost]), int(nblk), int(mode),
                                      1 if three_members else 0, table)
    check(rc if rc < 0 else 0)
    if rc == 1:
        out = []
        for b in range(nblk):
            w0, w1 = table[2 * b], table[2 * b + 1]
            e = [w0 & 0xFFFF, w0 >> 16, w1 & 0xFFFF]
            out.append([(v & 2047, v >> 11) for v in e[:n]])
        return 1, out
    if rc == 2:
        return 2, [int(table[i]) for i in range(nblk)]
    return 0, None


def _flag_ptr(flag):
    """Address of a range-flag word: None, a GuardWord (weights: word 1), an int address or an int32 tensor."""
    if flag is None:
        return None
    if isinstance(flag, GuardWord):
        return flag.ptr(1)
    if isinstance(flag, int):
        return flag
    return flag.data_ptr()


def _guard_ptr(guard):
    if guard is None:
        return None
    if isinstance(guard, GuardWord):
        return guard.ptr(0)
    return guard.data_ptr()


# ---------------------------------------------------------------------------
# weight preparation
# ---------------------------------------------------------------------------

def fold_weight_norm(v, g):
    """w = v * g/||v|| over all dims but 0 (torch.nn.utils.weight_norm semantics)."""
    v = v.detach().contiguous().float()
    g = g.detach().contiguous().float()
    w = torch.empty_like(v)
    with _on(v, g) as stream:
        check(lib().fv_fold_weight_norm(_ptr(v, "v"), _ptr(g, "g"), _ptr(w), v.shape[0], v[0].numel(), stream))
    return w


def pack_conv1d(w):
    """Conv1d weight [Cout,Cin,k] -> packed K-major image (flat tensor)."""
    w = w.detach().contiguous().float()
    cout, cin, k = w.shape
    out = torch.empty(lib().fv_packed_conv1d_floats(cout, cin, k), dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(lib().fv_pack_conv1d_weight(_ptr(w, "w"), _ptr(out), cout, cin, k, stream))
    return out


def pack_conv_transpose1d(w, stride, pad):
    """ConvTranspose1d weight [Cin,Cout,k] -> packed polyphase image (flat tensor)."""
    w = w.detach().contiguous().float()
    cin, cout, k = w.shape
    n = lib().fv_packed_conv_transpose1d_floats(cin, cout, k, stride, pad)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(lib().fv_pack_conv_transpose1d_weight(_ptr(w, "w"), _ptr(out), cin, cout, k, stride, pad, stream))
    return out


def conv_transpose_split_supported(cin, cout, k, stride, pad, out_pad):
    """Shapes of the split-f16 transposed conv (csrc/convh_launch.hip launch_convt)."""
    if conv_transpose_small(cin, cout, k, stride):
        return pad == 1 and out_pad == 0
    return (cin in (32, 64, 128, 256, 512) and 2 <= stride <= 16 and k == 2 * stride and cout * stride >= 32
            and 0 <= pad <= stride and -stride <= out_pad < stride)


def conv_transpose_small(cin, cout, k, stride):
    """The shape with a kernel of its own (csrc/convtn_kernels.hpp): HiFi-GAN light's last upsampler, 32 -> 16 channels x 2."""
    return (cin, cout, k, stride) == (32, 16, 4, 2)


def pack_conv_transpose1d_split(w, stride, flag=None):
    """ConvTranspose1d weight [Cin,Cout,2*stride] -> split-f16 stage image of convt_kernel (flat fp32-typed tensor).
    ``flag`` (GuardWord / int32 tensor): raised by the kernel when a weight is outside the f16 range."""
    w = w.detach().contiguous().float()
    cin, cout, k = w.shape
    n = lib().fv_packed_conv_transpose1d_split_floats(cin, cout, k, stride)
    if n <= 0:
        raise NativeError(f"pack_conv_transpose1d_split: Cin={cin} Cout={cout} k={k} stride={stride} is not built")
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(lib().fv_pack_conv_transpose1d_split_f16(_ptr(w, "w"), _ptr(out), cin, cout, k, stride, _flag_ptr(flag),
                                                       stream))
    return out


def conv1x1_2src_split_supported(channels):
    """Channel counts of the split-f16 two-source 1x1 conv (csrc/convh_launch.hip launch_convg)."""
    return channels in (128, 256, 512)


def pack_conv1x1_2src_split(w1, w2, flag=None):
    """Two 1x1 Conv1d weights [C,C,1] -> split-f16 stage image of convg_kernel for y = W1 act(x) + W2 x2 (flat tensor)."""
    w1, w2 = w1.detach().contiguous().float(), w2.detach().contiguous().float()
    c = w1.shape[0]
    if tuple(w1.shape) != (c, c, 1) or tuple(w2.shape) != (c, c, 1):
        raise NativeError(f"pack_conv1x1_2src_split: two [C,C,1] weights expected, got {tuple(w1.shape)}, {tuple(w2.shape)}")
    n = lib().fv_packed_conv1x1_2src_split_floats(c)
    if n <= 0:
        raise NativeError(f"pack_conv1x1_2src_split: C={c} is not built (128, 256, 512)")
    out = torch.empty(n, dtype=torch.float32, device=w1.device)
    with _on(w1, w2) as stream:
        check(lib().fv_pack_conv1x1_2src_split_f16(_ptr(w1, "w1"), _ptr(w2, "w2"), _ptr(out), c, _flag_ptr(flag), stream))
    return out


def residual_stack_split_supported(channels, k, dil):
    """Shapes of the one-launch MelGAN ResidualStack (csrc/convk_kernels.hpp)."""
    return channels in (32, 64, 128, 256) and k == 3 and dil in (1, 3, 9)


def pack_residual_stack_split(w_dilated, w_pointwise, w_skip, flag=None):
    """The three Conv1d weights of a ResidualStack ([C,C,3] dilated, [C,C,1] stack[4], [C,C,1] skip_layer) -> the
    split-f16 stage image of convk_kernel (flat tensor)."""
    ws = [w.detach().contiguous().float() for w in (w_dilated, w_pointwise, w_skip)]
    c, k = ws[0].shape[0], ws[0].shape[2]
    if tuple(ws[0].shape) != (c, c, k) or tuple(ws[1].shape) != (c, c, 1) or tuple(ws[2].shape) != (c, c, 1):
        raise NativeError(f"pack_residual_stack_split: [C,C,k], [C,C,1], [C,C,1] expected, got {[tuple(w.shape) for w in ws]}")
    n = lib().fv_packed_residual_stack_floats(c, k)
    if n <= 0:
        raise NativeError(f"pack_residual_stack_split: C={c}, k={k} is not built (32 / 64 / 128 / 256 channels, 3 taps)")
    out = torch.empty(n, dtype=torch.float32, device=ws[0].device)
    with _on(*ws) as stream:
        check(lib().fv_pack_residual_stack_split_f16(_ptr(ws[0], "w_dilated"), _ptr(ws[1], "w_pointwise"), _ptr(ws[2], "w_skip"),
                                                     _ptr(out), c, k, _flag_ptr(flag), stream))
    return out


def residual_stack_split_f16(x, packed, bias_dilated, bias_out, k, dil, slope, pad_mode=PAD_REFLECT, out=None, out_act=None,
                             act_slope=1.0, guard=None, post=POST_NONE):
    """MelGAN ResidualStack as one launch (fv_residual_stack_split_f16); packed = pack_residual_stack_split(...),
    bias_out = stack[4].bias + skip_layer.bias."""
    B, c, T = x.shape
    if out is None:
        out = torch.empty_like(x)
    with _on(x, packed, bias_dilated, bias_out, out, out_act) as stream:
        check(lib().fv_residual_stack_split_f16(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(bias_dilated, "bias_dilated", True),
                                                _ptr(bias_out, "bias_out", True), _ptr(out, "out"), _ptr(out_act, "out_act", True),
                                                B, c, T, k, dil, float(slope), pad_mode, post, float(act_slope), _guard_ptr(guard), stream))
    return out


def fold_batchnorm_conv(w, b, bn):
    """Fold an eval-mode ``torch.nn.BatchNorm1d`` into the conv weight/bias that follow it;
    returns (w', b') on w's device."""
    w = w.detach().contiguous().float()
    cout, cin, k = w.shape
    w_out = torch.empty_like(w)
    b_out = torch.empty(cout, dtype=torch.float32, device=w.device)
    t = lambda a: None if a is None else a.detach().contiguous().float()  # noqa: E731
    b, gamma, beta, mean, var = t(b), t(bn.weight), t(bn.bias), t(bn.running_mean), t(bn.running_var)
    with _on(w, b, gamma, beta, mean, var) as stream:
        check(lib().fv_fold_batchnorm_conv(_ptr(w, "w"), _ptr(b, "b", True), _ptr(gamma, "gamma", True),
                                           _ptr(beta, "beta", True), _ptr(mean, "running_mean"),
                                           _ptr(var, "running_var"), float(bn.eps), _ptr(w_out), _ptr(b_out),
                                           cout, cin, k, stream))
    return w_out, b_out


def pack_upsample_conv1d(w, rate, pad):
    """UpsampleLayer conv weight [Cout,Cin,k] -> packed phase image (flat tensor)."""
    w = w.detach().contiguous().float()
    cout, cin, k = w.shape
    n = lib().fv_packed_upsample_conv1d_floats(cout, cin, k, rate, pad)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(lib().fv_pack_upsample_conv1d_weight(_ptr(w, "w"), _ptr(out), cout, cin, k, rate, pad, stream))
    return out


def pack_pair(w, prec=PAIR_F32, flag=None):
    """ResBlock Conv1d weight [C,C,k] -> A-operand image of the fused pair kernels (flat tensor); ``prec``
    selects the fp32 (pair_kernels.hpp) or the split-f16 layout (pairh_kernels.hpp).  ``flag`` (GuardWord / int32
    tensor, split-f16 only): raised by the kernel when a weight is outside the f16 range."""
    w = w.detach().contiguous().float()
    c, c2, k = w.shape
    if c != c2:
        raise NativeError(f"pack_pair: square [C,C,k] weight expected, got {tuple(w.shape)}")
    n = lib().fv_packed_pair_floats_ex(c, k, prec)
    if n <= 0:
        raise NativeError(f"pack_pair: no packed layout for C={c} k={k} arithmetic {prec}")
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(lib().fv_pack_pair_weight_ex(_ptr(w, "w"), _ptr(out), c, k, prec, _flag_ptr(flag), stream))
    return out


def pair_supported(channels, k, dil, prec=PAIR_F32):
    """Shapes the fused ResBlock-pair kernels are built for (csrc/pair_launch.hip)."""
    chans = (16, 32) if prec == PAIR_F32 else (16, 32, 64, 128, 256, 512)
    return channels in chans and k in (3, 7, 11) and dil in (1, 3, 5) and prec in (PAIR_F32, PAIR_SPLIT_F16)


def conv_split_supported(channels, k, dil):
    """Shapes of the conv-level split-f16 op (csrc/convh_launch.hip): the ResBlock shapes plus MelGAN's dilation 9."""
    return channels in (64, 128, 256, 512) and ((k in (3, 7, 11) and dil in (1, 3, 5)) or (k == 3 and dil == 9))


def _vp_array(tensors, name, allow_none=False):
    return (ctypes.c_void_p * len(tensors))(*[_ptr(t, name, allow_none) for t in tensors])


# ---------------------------------------------------------------------------
# single fused operators (used by tests and by modules outside a plan)
# ---------------------------------------------------------------------------

def resblock1_fused(xs, w1s, w2s, b1s, b2s, ks, dil, slope, act_slope=1.0, outs=None, outs_act=None,
                    prec=PAIR_F32, add1=None, add2=None, out_div=1.0, post=POST_NONE, mids=None, guard=None):
    """n = len(xs) independent fused ResBlock pairs in one launch (fv_resblock1_fused_ex):
    y_j = x_j + conv2_j(lrelu(conv1_j(lrelu(x_j)) + b1_j)) + b2_j; w1s / w2s from pack_pair(w, prec).
    With add1 / add2 (split-f16 arithmetic): y_j = post(((y_j + add1_j) + add2_j) / out_div)."""
    n = len(xs)
    B, C, T = xs[0].shape
    if outs is None:
        outs = [torch.empty_like(x) for x in xs]
    acts = list(outs_act) if outs_act is not None else [None] * n
    a1 = list(add1) if add1 is not None else [None] * n
    a2 = list(add2) if add2 is not None else [None] * n
    # C >= 64 (split-f16): the pair runs as two conv launches through a scratch tensor per member
    mids = [torch.empty_like(x) if C >= 64 else None for x in xs] if mids is None else list(mids)
    with _on(*xs, *w1s, *w2s, *b1s, *b2s, *outs, *acts, *a1, *a2, *mids) as stream:
        check(lib().fv_resblock1_fused_ex(n, _vp_array(xs, "x"), _vp_array(w1s, "w1"), _vp_array(w2s, "w2"),
                                          _vp_array(b1s, "b1", True), _vp_array(b2s, "b2", True),
                                          _vp_array(outs, "y"), _vp_array(acts, "y_act", True),
                                          _vp_array(mids, "mid", True),
                                          _vp_array(a1, "add1", True), _vp_array(a2, "add2", True),
                                          (ctypes.c_int * n)(*ks), B, C, T, dil, float(slope), float(out_div), post,
                                          float(act_slope), prec, _guard_ptr(guard), stream))
    return outs


def conv1d_split_f16(xs, packed, biases, ks, dil, pre_slope=1.0, res=None, add1=None, add2=None, out_div=1.0,
                     post=POST_NONE, act_slope=1.0, outs=None, outs_act=None, pad_mode=PAD_ZERO, guard=None):
    """n = len(xs) independent 'same' convs with split-f16 operands in one launch (fv_conv1d_split_f16), C = 64 ... 512:
    y_j = post((conv(pad(lrelu(x_j, pre_slope))) + bias_j + res_j + add1_j + add2_j) / out_div), zero or reflection
    padding; packed from pack_pair(w, PAIR_SPLIT_F16)."""
    n = len(xs)
    B, C, T = xs[0].shape
    if outs is None:
        outs = [torch.empty_like(x) for x in xs]
    none = [None] * n
    acts = list(outs_act) if outs_act is not None else none
    rs, a1, a2 = (list(v) if v is not None else none for v in (res, add1, add2))
    with _on(*xs, *packed, *biases, *rs, *a1, *a2, *outs, *acts) as stream:
        check(lib().fv_conv1d_split_f16(n, _vp_array(xs, "x"), _vp_array(packed, "packed"),
                                        _vp_array(biases, "bias", True), _vp_array(rs, "res", True),
                                        _vp_array(a1, "add1", True), _vp_array(a2, "add2", True), _vp_array(outs, "y"),
                                        _vp_array(acts, "y_act", True), (ctypes.c_int * n)(*ks), B, C, T, dil,
                                        pad_mode, float(pre_slope), float(out_div), post, float(act_slope),
                                        _guard_ptr(guard), stream))
    return outs


MRF_STAGE_DILATIONS = (1, 3, 5)


def mrf_stage_supported(channels, ks, dils):
    """Shapes the one-launch MRF stage kernels are built for (csrc/mrfh_launch.hip): 16 or 32 channels, three ResBlocks
    with 3 / 7 / 11 taps (any order), pair dilations (1, 3, 5)."""
    return (channels in (16, 32) and len(ks) == 3 and all(k in (3, 7, 11) for k in ks)
            and tuple(dils) == MRF_STAGE_DILATIONS)


def mrf_stage_workspace(channels, device):
    """The scratch a one-launch MRF stage needs next to its tensors (fv_mrf_stage_workspace_bytes: the 32-channel
    kernel's columns of history; nothing at 16 channels -> None).  One per launch in flight; contents don't matter."""
    n = lib().fv_mrf_stage_workspace_bytes(int(channels))
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device) if n > 0 else None


def _work_args(work):
    return (_ptr(work, "workspace", True), 0 if work is None else work.numel() * 4)


def pack_mrf_stage(w1s, w2s, b1s, b2s, ks, flag=None):
    """The 18 convs of a 16-channel MRF stage -> the packed stage of fv_mrf_stage_split_f16 (flat tensor).  w1s / w2s:
    nine [16, 16, k_j] weights in the order pair p of ResBlock j at index 3 j + p; b1s / b2s: nine [16] biases or None;
    ks: the three ResBlocks' taps.  ``flag``: raised for a non-finite weight (as pack_pair)."""
    w1s = [w.detach().contiguous().float() for w in w1s]
    w2s = [w.detach().contiguous().float() for w in w2s]
    b1s = [None if b is None else b.detach().contiguous().float() for b in b1s]
    b2s = [None if b is None else b.detach().contiguous().float() for b in b2s]
    if not (len(w1s) == len(w2s) == len(b1s) == len(b2s) == 9 and len(ks) == 3):
        raise NativeError("pack_mrf_stage: nine pairs (three ResBlocks x three positions) expected")
    c = w1s[0].shape[0]
    for i, w in enumerate(w1s + w2s):
        if tuple(w.shape) != (c, c, ks[(i % 9) // 3]):
            raise NativeError(f"pack_mrf_stage: weight {i} has shape {tuple(w.shape)}, expected {(c, c, ks[(i % 9) // 3])}")
    karr = (ctypes.c_int * 3)(*ks)
    nfl = lib().fv_packed_mrf_stage_floats(c, karr)
    if nfl <= 0:
        raise NativeError(f"pack_mrf_stage: no packed layout for C={c} taps={list(ks)}")
    out = torch.empty(nfl, dtype=torch.float32, device=w1s[0].device)
    with _on(*w1s, *w2s, *b1s, *b2s, out) as stream:
        check(lib().fv_pack_mrf_stage_split_f16(_vp_array(w1s, "w1"), _vp_array(w2s, "w2"), _vp_array(b1s, "b1", True),
                                                _vp_array(b2s, "b2", True), _ptr(out), c, karr, _flag_ptr(flag), stream))
    return out


def mrf_stage_split_f16(x, packed, ks, dils=MRF_STAGE_DILATIONS, slope=0.1, out_div=3.0, post=POST_NONE, act_slope=1.0,
                        out=None, out_act=None, fold=None, guard=None):
    """A whole 16- or 32-channel MRF stage in one launch (fv_mrf_stage_split_f16): y = post(((r0 + r1) + r2) / out_div)
    with r_j = ResBlock1_j(x); ``packed`` from pack_mrf_stage.  ``fold`` = (w [C, 7], bias [1] or None):
    returns post(conv1d(lrelu(y, act_slope); w, padding 3) + bias), [B, 1, T], instead of y.  The 32-channel kernel's
    scratch is allocated here, per call (stream-ordered by the caching allocator)."""
    B, C, T = x.shape
    karr, darr = (ctypes.c_int * 3)(*ks), (ctypes.c_int * 3)(*dils)
    work = mrf_stage_workspace(C, x.device)
    if fold is not None:
        fw, fb = fold
        fw = fw.detach().contiguous().float()
        fb = None if fb is None else fb.detach().contiguous().float()
        res = torch.empty((B, 1, T), dtype=torch.float32, device=x.device)
        with _on(x, packed, fw, fb, res) as stream:
            check(lib().fv_mrf_stage_split_f16(_ptr(x, "x"), _ptr(packed, "packed"), None, None, B, C, T, karr, darr,
                                               float(slope), float(out_div), post, float(act_slope), _ptr(fw, "fold_w"),
                                               _ptr(fb, "fold_b", True), _ptr(res), *_work_args(work), _guard_ptr(guard), stream))
        return res
    if out is None:
        out = torch.empty_like(x)
    with _on(x, packed, out, out_act, work) as stream:
        check(lib().fv_mrf_stage_split_f16(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(out, "y"), _ptr(out_act, "y_act", True),
                                           B, C, T, karr, darr, float(slope), float(out_div), post, float(act_slope), None,
                                           None, None, *_work_args(work), _guard_ptr(guard), stream))
    return out


def mrf_stage(xs, w1s, w2s, b1s, b2s, ks, dil, slope, out_div=3.0, post=POST_NONE, act_slope=1.0, out=None,
              out_act=None):
    """y = post(sum_j pair_j(x_j) / out_div): the last pairs of three ResBlocks + the MRF mean (fv_mrf_stage)."""
    B, C, T = xs[0].shape
    if out is None:
        out = torch.empty_like(xs[0])
    with _on(*xs, *w1s, *w2s, *b1s, *b2s, out, out_act) as stream:
        check(lib().fv_mrf_stage(_vp_array(xs, "x"), _vp_array(w1s, "w1"), _vp_array(w2s, "w2"),
                                 _vp_array(b1s, "b1", True), _vp_array(b2s, "b2", True), _ptr(out, "y"),
                                 _ptr(out_act, "y_act", True), (ctypes.c_int * 3)(*ks), B, C, T, dil, float(slope),
                                 float(out_div), post, float(act_slope), stream))
    return out


def conv1d_fused(x, packed, bias, cout, k, dil=1, pad=0, pad_mode=PAD_ZERO, pre_slope=1.0,
                 res=None, acc_in=None, out_div=1.0, post=POST_NONE, out=None, out_act=None,
                 act_slope=1.0, acc_in2=None):
    """One fused conv launch; ``out_act`` (optional) receives lrelu(out, act_slope)."""
    B, cin, T = x.shape
    tout = T if pad_mode & PAD_CAUSAL else T + 2 * pad - dil * (k - 1)
    if out is None:
        out = torch.empty((B, cout, tout), dtype=torch.float32, device=x.device)
    with _on(x, packed, bias, res, acc_in, acc_in2, out, out_act) as stream:
        check(lib().fv_conv1d_fused(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(bias, "bias", True),
                                    _ptr(res, "res", True), _ptr(acc_in, "acc_in", True),
                                    _ptr(acc_in2, "acc_in2", True), _ptr(out, "out"),
                                    _ptr(out_act, "out_act", True), B, cin, cout, T, k, dil, pad, pad_mode,
                                    float(pre_slope), float(out_div), post, float(act_slope), stream))
    return out


def conv1d_2src_fused(x, x2, packed, bias, cout, res=None, post=POST_NONE, out=None, out_act=None,
                      act_slope=1.0):
    """y = post(W1 x + W2 x2 + bias + res) with packed = pack_conv1d(cat([W1, W2], 1)); 1-tap convs."""
    B, c1, T = x.shape
    c2 = x2.shape[1]
    if out is None:
        out = torch.empty((B, cout, T), dtype=torch.float32, device=x.device)
    with _on(x, x2, packed, bias, res, out, out_act) as stream:
        check(lib().fv_conv1d_2src_fused(_ptr(x, "x"), _ptr(x2, "x2"), _ptr(packed, "packed"),
                                         _ptr(bias, "bias", True), _ptr(res, "res", True), _ptr(out, "out"),
                                         _ptr(out_act, "out_act", True), B, c1, c2, cout, T, post,
                                         float(act_slope), stream))
    return out


def conv1x1_2src_split_f16(x, x2, packed, bias, pre_slope=1.0, res=None, post=POST_NONE, out=None, out_act=None,
                           act_slope=1.0, guard=None):
    """y = post(W1 lrelu(x, pre_slope) + W2 x2 + bias + res), split-f16 operands (fv_conv1x1_2src_split_f16);
    packed = pack_conv1x1_2src_split(W1, W2)."""
    B, c, T = x.shape
    if out is None:
        out = torch.empty_like(x)
    with _on(x, x2, packed, bias, res, out, out_act) as stream:
        check(lib().fv_conv1x1_2src_split_f16(_ptr(x, "x"), _ptr(x2, "x2"), _ptr(packed, "packed"), _ptr(bias, "bias", True),
                                              _ptr(res, "res", True), _ptr(out, "out"), _ptr(out_act, "out_act", True), B, c,
                                              T, float(pre_slope), post, float(act_slope), _guard_ptr(guard), stream))
    return out


def conv_transpose1d_fused(x, packed, bias, cout, k, stride, pad, out_pad, pre_slope=1.0,
                           post=POST_NONE, out=None, out_act=None, act_slope=1.0):
    B, cin, T = x.shape
    tout = (T - 1) * stride - 2 * pad + k + out_pad
    if out is None:
        out = torch.empty((B, cout, tout), dtype=torch.float32, device=x.device)
    with _on(x, packed, bias, out, out_act) as stream:
        check(lib().fv_conv_transpose1d_fused(_ptr(x, "x"), _ptr(packed, "packed"),
                                              _ptr(bias, "bias", True), _ptr(out, "out"),
                                              _ptr(out_act, "out_act", True), B, cin, cout, T, k, stride,
                                              pad, out_pad, float(pre_slope), post, float(act_slope),
                                              stream))
    return out


def pack_basis(W):
    """BasisSignalLayer weight W [L, C] (nn.Linear.weight) -> fv_pack_basis image (flat tensor)."""
    L_, c = W.shape
    n = lib().fv_packed_basis_floats(L_, c)
    if n <= 0:
        raise NativeError(f"pack_basis: L={L_} (even, >= 2) C={c}")
    out = torch.empty(n, dtype=torch.float32, device=W.device)
    with _on(W, out) as stream:
        check(lib().fv_pack_basis(_ptr(W, "W"), _ptr(out), L_, c, stream))
    return out


def basis_ola(weight, packed_basis, L_):
    """fv_basis_ola: weight [B, C, F] (channel-major trunk output) -> [B, 1, (F - 1) L/2 + L]."""
    B, c, F = weight.shape
    out = torch.empty((B, 1, (F - 1) * (L_ // 2) + L_), dtype=torch.float32, device=weight.device)
    with _on(weight, packed_basis, out) as stream:
        check(lib().fv_basis_ola(_ptr(weight, "weight"), _ptr(packed_basis, "packed_basis"), _ptr(out), B, c, F, L_, stream))
    return out


def conv_transpose1d_split_f16(x, packed, bias, cout, k, stride, pad, out_pad, pre_slope=1.0, out=None, out_act=None,
                               act_slope=1.0, guard=None):
    """ConvTranspose1d (kernel = 2 stride) with split-f16 operands (fv_conv_transpose1d_split_f16)."""
    B, cin, T = x.shape
    tout = (T - 1) * stride - 2 * pad + k + out_pad
    if out is None:
        out = torch.empty((B, cout, tout), dtype=torch.float32, device=x.device)
    with _on(x, packed, bias, out, out_act) as stream:
        check(lib().fv_conv_transpose1d_split_f16(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(bias, "bias", True),
                                                  _ptr(out, "out"), _ptr(out_act, "out_act", True), B, cin, cout, T, k,
                                                  stride, pad, out_pad, float(pre_slope), float(act_slope),
                                                  _guard_ptr(guard), stream))
    return out


def upsample_conv1d_fused(x, packed, bias, cout, k, rate, pad, pre_slope=1.0, post=POST_NONE,
                          out=None, out_act=None, act_slope=1.0):
    B, cin, T = x.shape
    tout = T * rate + 2 * pad - (k - 1)
    if out is None:
        out = torch.empty((B, cout, tout), dtype=torch.float32, device=x.device)
    with _on(x, packed, bias, out, out_act) as stream:
        check(lib().fv_upsample_conv1d_fused(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(bias, "bias", True),
                                             _ptr(out, "out"), _ptr(out_act, "out_act", True), B, cin, cout,
                                             T, k, rate, pad, float(pre_slope), post, float(act_slope),
                                             stream))
    return out


def encode_16bits(x, rescale_out=1.0, scale_in_place=True):
    """x [n] or [B,n] float32 device tensor -> (int16 tensor of the same shape, peaks [B]).
    Row-wise ``encode_16bits`` of the reference; with ``scale_in_place`` x is scaled like
    the reference scales its argument."""
    flat = x if x.dim() == 2 else x.reshape(1, -1)
    B, n = flat.shape
    out = torch.empty((B, n), dtype=torch.int16, device=x.device)
    peak = torch.empty(B, dtype=torch.float32, device=x.device)
    with _on(flat, out, peak) as stream:
        check(lib().fv_encode_16bits(_ptr(flat, "x"), out.data_ptr(), _ptr(peak), B, n, float(rescale_out),
                                     1 if scale_in_place else 0, stream))
    return out.reshape(x.shape), peak


def pqmf_analysis(x, analysis_filter):
    """x [B,1,T] or [B,T] full band, analysis_filter [S,1,ntaps] -> [B,S,(T-S)//S+1]."""
    S, ntaps = analysis_filter.shape[0], analysis_filter.shape[-1]
    x = x.reshape(x.shape[0], -1)
    B, T = x.shape
    h = analysis_filter.detach().reshape(S, ntaps).contiguous().float()
    y = torch.empty((B, S, (T - S) // S + 1), dtype=torch.float32, device=x.device)
    with _on(x, h, y) as stream:
        check(lib().fv_pqmf_analysis(_ptr(x, "x"), _ptr(h, "analysis_filter"), _ptr(y), B, S, ntaps, T, stream))
    return y


def pqmf_synthesis(x, synthesis_filter, y):
    """x [B,S,Tsub], synthesis_filter [1,S,ntaps] -> y [B,1,S*Tsub] (filled in place)."""
    B, S, Tsub = x.shape
    h = synthesis_filter.reshape(S, -1).contiguous().float()
    with _on(x, h, y) as stream:
        check(lib().fv_pqmf_synthesis(_ptr(x, "x"), _ptr(h, "h"), _ptr(y, "y"), B, S, h.shape[1], Tsub,
                                      stream))
    return y


def conv_post_pqmf(x, packed, bias, synthesis_filter, k, pad, pre_slope=1.0, post=POST_TANH):
    """pqmf_synthesis(post(conv1d(lrelu(x, pre_slope)) + bias)) in one launch (fv_conv_post_pqmf): x [B,Cin,T],
    packed = pack_conv1d(w [S,Cin,k]), synthesis_filter [1,S,ntaps] -> [B,1,S*T]."""
    B, cin, T = x.shape
    S = synthesis_filter.shape[1]
    h = synthesis_filter.reshape(S, -1).contiguous().float()
    y = torch.empty((B, 1, S * T), dtype=torch.float32, device=x.device)
    with _on(x, packed, bias, h, y) as stream:
        check(lib().fv_conv_post_pqmf(_ptr(x, "x"), _ptr(packed, "packed"), _ptr(bias, "bias", True), _ptr(h, "h"),
                                      _ptr(y, "y"), B, cin, S, T, k, pad, float(pre_slope), post, h.shape[1], stream))
    return y


# ---------------------------------------------------------------------------
# plans
# ---------------------------------------------------------------------------

class Plan:
    """Owner of a native op list (fv_plan_t) plus the tensors its ops point at."""

    def __init__(self, in_channels):
        self._h = lib().fv_plan_create(in_channels)
        self.in_channels = in_channels
        self._keep = []        # packed weights / biases the native plan references
        self._ws = None
        self._calls = {}       # (B, T, device) -> (output channels, output length, workspace bytes)
        self._guard = None     # GuardWord of the owning module (set_guard)
        self._dev = None       # device of the last run
        self.guarded = False   # the plan holds split-f16 launches (PlanBuilder.finalize)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.fv_plan_destroy(h)

    def keep(self, t):
        self._keep.append(t)
        return t

    # a plan is a raw native handle + device pointers into its owner's tensors: never copied or pickled
    def __deepcopy__(self, memo):
        raise NativeError("a native plan cannot be copied; the owning module rebuilds its plans on demand")

    def __reduce__(self):
        raise NativeError("a native plan cannot be pickled; the owning module rebuilds its plans on demand")

    def add_conv1d(self, x, y, packed, bias, cin, cout, k, dil=1, pad=0, pad_mode=PAD_ZERO,
                   pre_slope=1.0, res=SLOT_NONE, acc=SLOT_NONE, out_div=1.0, post=POST_NONE,
                   y_act=SLOT_NONE, act_slope=1.0, acc2=SLOT_NONE):
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_conv1d(self._h, x, y, y_act, res, acc, acc2, _ptr(packed, "packed"),
                                       _ptr(bias, "bias", True), cin, cout, k, dil, pad, pad_mode,
                                       float(pre_slope), float(out_div), post, float(act_slope)))

    def add_conv_transpose1d(self, x, y, packed, bias, cin, cout, k, stride, pad, out_pad,
                             pre_slope=1.0, post=POST_NONE, y_act=SLOT_NONE, act_slope=1.0):
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_conv_transpose1d(self._h, x, y, y_act, _ptr(packed, "packed"),
                                                 _ptr(bias, "bias", True), cin, cout, k, stride,
                                                 pad, out_pad, float(pre_slope), post,
                                                 float(act_slope)))

    def add_conv_transpose1d_split_f16(self, x, y, packed, bias, cin, cout, k, stride, pad, out_pad, pre_slope=1.0,
                                       y_act=SLOT_NONE, act_slope=1.0, merge=None):
        """``merge = (add1_slot, add2_slot, div)``: the input is ((x + add1) + add2) / div, formed in the kernel's window
        loader (fv_plan_set_input_merge: the MRF merge of hifigan.py:99-103 inside the upsampler behind the stage)."""
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_conv_transpose1d_split_f16(self._h, x, y, y_act, _ptr(packed, "packed"),
                                                           _ptr(bias, "bias", True), cin, cout, k, stride, pad, out_pad,
                                                           float(pre_slope), float(act_slope)))
        if merge is not None:
            check(lib().fv_plan_set_input_merge(self._h, int(merge[0]), int(merge[1]), float(merge[2])))

    def add_conv1d_2src(self, x, x2, y, packed, bias, cin1, cin2, cout, res=SLOT_NONE, post=POST_NONE,
                        y_act=SLOT_NONE, act_slope=1.0):
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_conv1d_2src(self._h, x, x2, y, y_act, res, _ptr(packed, "packed"),
                                            _ptr(bias, "bias", True), cin1, cin2, cout, post,
                                            float(act_slope)))

    def add_conv1x1_2src_split_f16(self, x, x2, y, packed, bias, channels, pre_slope=1.0, res=SLOT_NONE, post=POST_NONE,
                                   y_act=SLOT_NONE, act_slope=1.0):
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_conv1x1_2src_split_f16(self._h, x, x2, y, y_act, res, _ptr(packed, "packed"),
                                                       _ptr(bias, "bias", True), channels, float(pre_slope), post,
                                                       float(act_slope)))

    def add_residual_stack_split_f16(self, x, y, packed, bias_dilated, bias_out, channels, k, dil, slope, pad_mode=PAD_REFLECT,
                                     y_act=SLOT_NONE, act_slope=1.0, two_launch=None, post=POST_NONE):
        """``two_launch`` = (hidden slot, pack_pair image of the dilated conv, pack_conv1x1_2src_split image of the 1x1 pair):
        the form a run with many tiles takes instead (fv_plan_set_stack_two_launch; 256 channels)."""
        self.keep(packed)
        for b in (bias_dilated, bias_out):
            if b is not None:
                self.keep(b)
        check(lib().fv_plan_add_residual_stack_split_f16(self._h, x, y, y_act, _ptr(packed, "packed"),
                                                         _ptr(bias_dilated, "bias_dilated", True), _ptr(bias_out, "bias_out", True),
                                                         channels, k, dil, float(slope), pad_mode, post, float(act_slope)))
        if two_launch is not None:
            hidden, pd, pp = two_launch
            self.keep(pd)
            self.keep(pp)
            check(lib().fv_plan_set_stack_two_launch(self._h, hidden, _ptr(pd, "packed_dilated"), _ptr(pp, "packed_pair")))

    def add_upsample_conv1d(self, x, y, packed, bias, cin, cout, k, rate, pad, pre_slope=1.0,
                            post=POST_NONE, y_act=SLOT_NONE, act_slope=1.0):
        self.keep(packed)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_add_upsample_conv1d(self._h, x, y, y_act, _ptr(packed, "packed"),
                                                _ptr(bias, "bias", True), cin, cout, k, rate, pad,
                                                float(pre_slope), post, float(act_slope)))

    def add_conv1d_sum3(self, xs, ress, tmps, y, packed, bias_sum, channels, ks, out_div=1.0, post=POST_NONE,
                        y_act=SLOT_NONE, act_slope=1.0):
        """The three last convs of an MRF stage into one output (fv_plan_add_conv1d_sum3)."""
        for t in packed:
            self.keep(t)
        if bias_sum is not None:
            self.keep(bias_sum)
        i3 = ctypes.c_int * 3
        wp = (ctypes.c_void_p * 3)(*[_ptr(t, "packed") for t in packed])
        check(lib().fv_plan_add_conv1d_sum3(self._h, i3(*xs), i3(*ress), (ctypes.c_int * 2)(*tmps), y, y_act, wp,
                                            _ptr(bias_sum, "bias_sum", True), channels, i3(*ks),
                                            float(out_div), post, float(act_slope)))

    def add_resblock_pair(self, x, y, packed1, packed2, bias1, bias2, channels, k, dil, slope,
                          y_act=SLOT_NONE, act_slope=1.0, prec=PAIR_F32, add1=SLOT_NONE, add2=SLOT_NONE,
                          out_div=1.0, post=POST_NONE, mid=SLOT_NONE):
        """One fused ResBlock pair (fv_plan_add_resblock_pair_ex); group members share a launch."""
        for t in (packed1, packed2, bias1, bias2):
            if t is not None:
                self.keep(t)
        check(lib().fv_plan_add_resblock_pair_ex(self._h, x, y, y_act, mid, add1, add2, _ptr(packed1, "packed1"),
                                                 _ptr(packed2, "packed2"), _ptr(bias1, "bias1", True),
                                                 _ptr(bias2, "bias2", True), channels, k, dil, float(slope),
                                                 float(out_div), post, float(act_slope), prec))

    def add_mrf_stage(self, x, y, packed, channels, ks, dils, slope, out_div=3.0, post=POST_NONE, y_act=SLOT_NONE,
                      act_slope=1.0):
        """A whole 16- / 32-channel MRF stage as one op / one launch (fv_plan_add_mrf_stage_split_f16); the 32-channel
        kernel's scratch belongs to the op (a plan runs its ops in order on one stream)."""
        self.keep(packed)
        work = mrf_stage_workspace(channels, packed.device)
        if work is not None:
            self.keep(work)
        check(lib().fv_plan_add_mrf_stage_split_f16(self._h, x, y, y_act, _ptr(packed, "packed"), channels,
                                                    (ctypes.c_int * 3)(*ks), (ctypes.c_int * 3)(*dils), float(slope),
                                                    float(out_div), post, float(act_slope), *_work_args(work)))

    def set_pair_output_conv(self, w, bias, y, act_slope, post=POST_NONE):
        """Fold a 16 -> 1 channel, 7-tap conv into the resblock pair appended last (fv_plan_set_pair_output_conv):
        ``y`` becomes post(conv(lrelu(pair result, act_slope)) + bias), [B, 1, T]."""
        self.keep(w)
        if bias is not None:
            self.keep(bias)
        check(lib().fv_plan_set_pair_output_conv(self._h, _ptr(w, "w"), _ptr(bias, "bias", True), y, float(act_slope), post))

    def add_conv1d_split_f16(self, x, y, packed, bias, channels, k, dil, pre_slope=1.0, res=SLOT_NONE, add1=SLOT_NONE,
                             add2=SLOT_NONE, out_div=1.0, post=POST_NONE, y_act=SLOT_NONE, act_slope=1.0,
                             pad_mode=PAD_ZERO):
        """One 'same' conv with split-f16 operands (fv_plan_add_conv1d_split_f16); group members share a launch."""
        for t in (packed, bias):
            if t is not None:
                self.keep(t)
        check(lib().fv_plan_add_conv1d_split_f16(self._h, x, y, y_act, res, add1, add2, _ptr(packed, "packed"),
                                                 _ptr(bias, "bias", True), channels, k, dil, pad_mode,
                                                 float(pre_slope), float(out_div), post, float(act_slope)))

    def add_mrf_sum(self, xs, y, packed1, packed2, bias1, bias2, channels, ks, dil, slope, out_div=3.0,
                    post=POST_NONE, y_act=SLOT_NONE, act_slope=1.0):
        """Last pairs of the three ResBlocks + the MRF mean in one launch (fv_plan_add_mrf_sum)."""
        for t in list(packed1) + list(packed2) + list(bias1) + list(bias2):
            if t is not None:
                self.keep(t)
        i3 = ctypes.c_int * 3
        check(lib().fv_plan_add_mrf_sum(self._h, i3(*xs), y, y_act, _vp_array(packed1, "packed1"),
                                        _vp_array(packed2, "packed2"), _vp_array(bias1, "bias1", True),
                                        _vp_array(bias2, "bias2", True), channels, i3(*ks), dil, float(slope),
                                        float(out_div), post, float(act_slope)))

    def set_sum_order(self, own_first):
        check(lib().fv_plan_set_sum_order(self._h, 1 if own_first else 0))

    def set_guard(self, guard):
        """Range guard of the plan's split-f16 launches (fv_plan_set_guard): a GuardWord, or None."""
        self._guard = self.keep(guard) if guard is not None else None
        check(lib().fv_plan_set_guard(self._h, guard.ptr(0) if guard is not None else None))

    def check_range(self):
        """Drain the current stream, then: did a split-f16 kernel of the runs since the last check meet an operand
        beyond the f16 range?  (fv_plan_check_range; clears the word.)"""
        if self._guard is None:
            return False
        rc = lib().fv_plan_check_range(self._h, torch.cuda.current_stream(self._dev).cuda_stream)
        if rc == ERR_RANGE:
            return True
        if rc == ERR_RANGE_LOW:
            return "low"             # (truthy: the run has to be repeated; the caller may treat it as a property of the input)
        check(rc)
        return False

    def set_group(self, group):
        check(lib().fv_plan_set_group(self._h, group))

    def add_pqmf_synthesis(self, x, y, h):
        """h [S, ntaps] contiguous fp32 device tensor."""
        self.keep(h)
        check(lib().fv_plan_add_pqmf_synthesis(self._h, x, y, _ptr(h, "h"), h.shape[0], h.shape[1]))

    def add_conv_post_pqmf(self, x, y, packed, bias, cin, k, pad, h, pre_slope=1.0, post=POST_TANH):
        """conv_post (cin -> S sub-bands) + post + PQMF synthesis as one op (fv_plan_add_conv_post_pqmf); h [S, ntaps]."""
        for t in (packed, bias, h):
            if t is not None:
                self.keep(t)
        check(lib().fv_plan_add_conv_post_pqmf(self._h, x, y, _ptr(packed, "packed"), _ptr(bias, "bias", True), cin,
                                               h.shape[0], k, pad, float(pre_slope), post, _ptr(h, "h"), h.shape[1]))

    def set_output_offset(self, aux_slot, y2_slot=SLOT_NONE, y_slot=SLOT_OUT):
        """The op added last (output slot ``y_slot``) subtracts auxiliary input ``aux_slot`` in its epilogue
        (fv_plan_set_output_offset)."""
        check(lib().fv_plan_set_output_offset(self._h, aux_slot, y2_slot))
        self.__dict__.setdefault("_aux_of", {})[aux_slot] = y_slot

    def _aux_elems(self, T, aux_slot):
        """Elements per utterance of the tensor auxiliary input ``aux_slot`` is subtracted from (None: unused)."""
        y = self.__dict__.get("_aux_of", {}).get(aux_slot)
        if y is None:
            return None
        c, n = self.slot_shape(T, y)
        return c * n

    def slot_shape(self, T, slot):
        c, n = ctypes.c_int(), ctypes.c_int64()
        check(lib().fv_plan_slot_shape(self._h, T, slot, ctypes.byref(c), ctypes.byref(n)))
        return c.value, n.value

    def output_shape(self, T):
        return self.slot_shape(T, SLOT_OUT)

    def num_ops(self):
        return lib().fv_plan_num_ops(self._h)

    def run(self, x, out=None, aux=(), out2=False):
        """x [B,Cin,T] contiguous fp32 on a ROCm device -> [B,Cout,Tout].  ``aux``: up to two tensors for the
        plan's auxiliary inputs (output offsets: [C,T'], [1,C,T'] or, per utterance, [B,C,T']); ``out2``: also
        return the plan's second output (SLOT_OUT2) -> (out, out2)."""
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise NativeError(f"plan input must be [B, {self.in_channels}, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        if B == 0 or T == 0:
            raise NativeError(f"plan input is empty: {tuple(x.shape)}")
        dev = x.device
        key = (B, T, dev)
        ent = self._calls.get(key)
        if ent is None:
            # once per (batch, frames, device): output shape and workspace size (two native queries), and that the plan's
            # packed weights live where the input does
            _on(x, *self._keep[:1])
            c, n = self.output_shape(T)
            nbytes = lib().fv_plan_workspace_bytes(self._h, B, T)
            if nbytes < 0:
                check(int(nbytes))
            if len(self._calls) >= 256:
                self._calls.clear()
            ent = self._calls[key] = (c, n, max(int(nbytes), 256))
        c, n, nbytes = ent
        if out is None:
            out = torch.empty((B, c, n), dtype=torch.float32, device=dev)
        self._dev = dev
        if self._ws is None or self._ws.device != dev or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # (grows, never shrinks: one arena per plan)
        if not aux and not out2:
            # the common call (no auxiliary inputs, one output): one pointer check per operand, no marshalling of empties
            if not (x.is_cuda and x.dtype is torch.float32 and x.is_contiguous()):
                _ptr(x, "input")
            if not (out.is_cuda and out.dtype is torch.float32 and out.is_contiguous() and out.device == dev):
                _ptr(out, "out")
                raise NativeError(f"out lives on {out.device}, the input on {dev}")
            if torch.cuda.current_device() == dev.index:
                check(lib().fv_plan_run_aux(self._h, B, T, x.data_ptr(), out.data_ptr(), None, None, None,
                                            self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(dev).cuda_stream))
            else:
                with torch.cuda.device(dev):
                    check(lib().fv_plan_run_aux(self._h, B, T, x.data_ptr(), out.data_ptr(), None, None, None,
                                                self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(dev).cuda_stream))
            return out
        second = None
        if out2:
            c2, n2 = self.slot_shape(T, SLOT_OUT2)
            second = torch.empty((B, c2, n2), dtype=torch.float32, device=x.device)
        aux = list(aux) + [None] * (2 - len(aux))
        ptrs = (ctypes.c_void_p * 2)(*[_ptr(a, "aux", True) for a in aux])
        # an auxiliary input is an output offset: exactly the element count of the tensor it is subtracted from
        # (per utterance, or one row per utterance) -- the kernels index it without bounds
        batched = []
        for j, a in enumerate(aux):
            if a is None:
                batched.append(0)
                continue
            per = self._aux_elems(T, SLOT_AUX_IN0 + j)
            if per is None:
                raise NativeError(f"auxiliary input {j} was given, but no op of the plan subtracts it")
            if a.numel() == per:
                batched.append(0)
            elif a.numel() == B * per:
                batched.append(1)
            else:
                raise NativeError(f"auxiliary input {j} has {a.numel()} elements; the tensor it is subtracted from has "
                                  f"{per} per utterance (batch {B})")
        batched = (ctypes.c_int * 2)(*batched)
        with _on(x, out, self._ws, second, *aux, *self._keep[:1]) as stream:
            check(lib().fv_plan_run_aux(self._h, B, T, _ptr(x, "input"), _ptr(out, "out"), _ptr(second, "out2", True),
                                        ptrs, batched, self._ws.data_ptr(), self._ws.numel(), stream))
        return (out, second) if out2 else out


# ---------------------------------------------------------------------------
# measurement hook
# ---------------------------------------------------------------------------

def profile_enable(on):
    check(lib().fv_profile_enable(1 if on else 0))


KERNEL_CONV_MFMA32, KERNEL_CONV_MFMA16, KERNEL_CONV_NARROW, KERNEL_PAIR16, KERNEL_PAIR32 = 0, 1, 2, 3, 4
KERNEL_PAIRH16, KERNEL_PAIRH32, KERNEL_CONVH64, KERNEL_CONVH128, KERNEL_CONVT, KERNEL_CONVG = 5, 6, 7, 8, 9, 10
KERNEL_STACK = 11
KERNEL_MRF16 = 12     # a whole 16-channel MRF stage as one launch (mrfh_kernel)
KERNEL_MRF32 = 13     # ... 32-channel (mrfw_kernel)


def profile_bracket_cost(n=200):
    """Milliseconds the (begin, end) event pair of the measurement hook adds per launch."""
    ms = ctypes.c_double()
    check(lib().fv_profile_bracket_cost(torch.cuda.current_stream().cuda_stream, n, ctypes.byref(ms)))
    return ms.value


def profile_mfma_f16_rate(launches=40, iters=20000):
    """The device's achievable dense f16 matrix rate in TFLOP/s (fv_profile_mfma_f16_rate: nothing but v_mfma_f32_16x16x32_f16 on
    registers, timed over the last half of ``launches`` back-to-back launches on the current stream)."""
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    scratch = torch.empty(512 * 2 * props.multi_processor_count, dtype=torch.float32, device="cuda")
    tf = ctypes.c_double()
    check(lib().fv_profile_mfma_f16_rate(scratch.data_ptr(), scratch.numel(), int(launches), int(iters),
                                         torch.cuda.current_stream().cuda_stream, ctypes.byref(tf)))
    return tf.value


def profile_collect(kind=-1):
    """Drain the per-launch HIP-event records of one kernel family (-1: all)."""
    n, ms = ctypes.c_int64(), ctypes.c_double()
    fl, by = ctypes.c_double(), ctypes.c_double()
    check(lib().fv_profile_collect(kind, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl),
                                   ctypes.byref(by)))
    return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}
