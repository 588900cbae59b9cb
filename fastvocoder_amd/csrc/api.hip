// C-ABI entry points of libfastvocoder_hip.so (include/fastvocoder_hip.h): the fused operators as direct calls, the op core they
// share with the plan executor (plan.hip), error text, tuning switches and the measurement hook.  (Weight preparation: pack.hip.)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <mutex>
#include <utility>
#include <vector>

#include <ctype.h>
#include <string.h>
#include <unistd.h>

#include "api_internal.h"

namespace fv {

static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code ? code : FV_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------
// measurement hook
// ---------------------------------------------------------------------------
// A launch's duration is completion-to-completion on its stream: from the end of the launch before it (or, for the
// first one, from an event recorded in front of it) to its own end -- its dispatch latency is part of it, as it is in
// rocprofv3's dispatch durations, and the durations of a run add up to the run.  One event per launch.
struct ProfRec {
    hipEvent_t a, b;        // a: only where there is no previous launch to measure from
    hipStream_t s;
    double ms;              // < 0: not read back yet
    double flops, bytes;
    int kind;
};
static bool g_prof_on = false;
static bool g_prof_need_start = true;
static std::vector<ProfRec> g_prof;
static hipEvent_t g_prof_start;

void profile_begin(hipStream_t s) {
    if (!g_prof_on) return;
    g_prof_start = nullptr;
    if (g_prof_need_start || g_prof.empty() || g_prof.back().s != s || !g_prof.back().b) {
        (void)hipEventCreate(&g_prof_start);
        (void)hipEventRecord(g_prof_start, s);
        g_prof_need_start = false;
    }
}

void profile_end(hipStream_t s, int kind, double flops, double bytes) {
    if (!g_prof_on) return;
    ProfRec r;
    r.a = g_prof_start;
    g_prof_start = nullptr;
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.b, s);
    r.s = s;
    r.ms = -1.0;
    r.flops = flops;
    r.bytes = bytes;
    r.kind = kind;
    g_prof.push_back(r);
}

// read every pending record back (in order: a record without its own start event is measured from its predecessor's
// end), then drop the events
static int profile_resolve() {
    for (size_t i = 0; i < g_prof.size(); ++i) {
        ProfRec& r = g_prof[i];
        if (r.ms >= 0) continue;
        FV_HIP(hipEventSynchronize(r.b));
        float e = 0.f;
        FV_HIP(hipEventElapsedTime(&e, r.a ? r.a : g_prof[i - 1].b, r.b));
        r.ms = e;
    }
    for (ProfRec& r : g_prof) {
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
        r.a = r.b = nullptr;
    }
    g_prof_need_start = true;
    return 0;
}

}  // namespace fv

namespace fv {

int64_t conv_out_len(const Op& o, int64_t Tin) {
    if (o.type == OP_PAIR || o.type == OP_MRFSUM || o.type == OP_CONVH || o.type == OP_CONVG || o.type == OP_STACK ||
        o.type == OP_STAGE)
        return Tin;
    if (o.type == OP_CONV) {
        const int64_t t = (o.pad_mode & FV_PAD_CAUSAL) ? Tin : Tin + 2LL * o.pad - (int64_t)o.dil * (o.k - 1);
        return o.pq_h ? t * o.Cout : t;      // (conv_post + pqmf: the S sub-bands interleave into S * T' samples)
    }
    if (o.type == OP_CONVT) return (Tin - 1) * o.stride - 2LL * o.pad + o.k + o.out_pad;
    if (o.type == OP_UPCONV) return Tin * o.stride + 2LL * o.pad - (o.k - 1);
    return Tin * o.Cin;  // PQMF: S sub-bands interleave into S*Tsub samples
}


ConvParams make_params(const Op& o, const float* x, float* y, float* y2, const float* res,
                              const float* acc, const float* acc2, int B, int64_t Tin,
                              const float* x2, const float* sub, int sub_batched) {
    ConvParams p = {};
    p.sub = sub;
    p.sub_batched = sub_batched;
    p.x = x;
    p.x2 = x2;
    p.Cin1 = x2 ? o.Cin1 : o.Cin;
    p.wp = o.wp;
    p.bias = o.bias;
    p.res = res;
    p.acc_in = acc;
    p.acc_in2 = acc2;
    p.y = y;
    p.y_act = y2;
    p.act_slope = o.act_slope;
    p.B = B;
    p.Cin = o.Cin;
    p.Cout = o.Cout;
    p.Tin = (int)Tin;
    p.pad_mode = o.pad_mode & ~FV_PAD_CAUSAL;
    p.pre_slope = o.pre_slope;
    p.out_div = o.out_div;
    p.post = o.post;
    p.own_first = o.own_first;
    p.Tout = (int)conv_out_len(o, Tin);
    if (o.type == OP_CONV) {
        p.M = o.Cout;
        p.k = o.k;
        p.dil = o.dil;
        p.pad = o.pad;
        p.ups = 1;
        p.Tq = p.Tout;
    } else {
        const Polyphase ph = o.type == OP_UPCONV ? upsample_phases(o.k, o.stride, o.pad)
                                                 : polyphase(o.k, o.stride, o.pad);
        p.M = o.Cout * o.stride;
        p.k = ph.taps;
        p.dil = 1;
        p.pad = -ph.dmin;
        if (o.type == OP_CONVT && convt_phase_major(o.Cout, o.k, o.stride, o.pad)) {
            p.phase_major = 1;
            p.pad_orig = o.pad;
            p.k = o.k / o.stride;          // taps of every phase
            p.pad = p.k - 1;               // the widest window shift (phases with (r+p)/s == 0)
        }
        p.pad_mode = FV_PAD_ZERO;
        p.ups = o.stride;
        p.Tq = (p.Tout + o.stride - 1) / o.stride;
        // MACs of a ConvTranspose1d = Tin * Cin * Cout * k, whatever the polyphase image pads
        if (o.type == OP_CONVT) p.alg_flops = 2.0 * B * (double)Tin * o.Cin * o.Cout * o.k;
    }
    p.Mpad = pad_rows(p.M);
    return p;
}

int run_op(const Op& o, const float* x, float* y, float* y2, const float* res, const float* acc,
                  const float* acc2, int B, int64_t Tin, hipStream_t s, const float* x2,
                  const float* sub, int sub_batched, int* guard) {
    if (o.type == OP_PQMF) return launch_pqmf(x, o.wp, y, y2, sub, sub_batched, B, o.Cin, o.k, (int)Tin, s);
    if (o.pq_h) {
        Op c = o;                              // the conv in front: [B, S, T'] sub-bands that never leave the CU
        c.pq_h = nullptr;
        return launch_conv_post_pqmf(make_params(c, x, y, nullptr, nullptr, nullptr, nullptr, B, Tin), o.pq_h, o.pq_taps, y, y2,
                                     sub, sub_batched, s);
    }
    if (o.type == OP_CONVG) {
        PairParams pp = {};
        pp.B = B;
        pp.T = (int)Tin;
        pp.slope = o.pre_slope;
        pp.act_slope = o.act_slope;
        pp.post = o.post;
        pp.prec = FV_PAIR_SPLIT_F16;
        pp.guard = guard;
        pp.sub = sub;
        pp.sub_batched = sub_batched;
        pp.m[0].x = x;
        pp.m[0].x2 = x2;
        pp.m[0].w1 = o.wp;
        pp.m[0].b1 = o.bias;
        pp.m[0].res = res;
        pp.m[0].y = y;
        pp.m[0].y_act = y2;
        return launch_convg(pp, o.Cout, s);
    }
    if (o.type == OP_STACK) {
        PairParams pp = {};
        pp.B = B;
        pp.T = (int)Tin;
        pp.slope = o.pre_slope;
        pp.act_slope = o.act_slope;
        pp.prec = FV_PAIR_SPLIT_F16;
        pp.guard = guard;
        pp.reflect = (o.pad_mode & FV_PAD_REFLECT) ? 1 : 0;
        pp.post = o.post;
        pp.sub = sub;
        pp.sub_batched = sub_batched;
        pp.m[0].x = x;
        pp.m[0].w1 = o.wp;
        pp.m[0].b1 = o.bias;
        pp.m[0].b2 = o.bias2;
        pp.m[0].y = y;
        pp.m[0].y_act = y2;
        return launch_convk(pp, o.Cout, o.dil, s);
    }
    if (o.type == OP_CONVT && o.prec == FV_PAIR_SPLIT_F16) {
        PairParams pp = {};
        pp.B = B;
        pp.T = (int)Tin;
        pp.slope = o.pre_slope;
        pp.act_slope = o.act_slope;
        pp.prec = FV_PAIR_SPLIT_F16;
        pp.guard = guard;
        pp.m[0].x = x;
        pp.m[0].w1 = o.wp;
        pp.m[0].b1 = o.bias;
        pp.m[0].y = y;
        pp.m[0].y_act = y2;
        // fv_plan_set_input_merge: the input is ((x + acc) + acc2) / out_div, formed in the kernel's window loader
        pp.m[0].add1 = acc;
        pp.m[0].add2 = acc2;
        pp.out_div = acc ? o.out_div : 1.f;
        if (convtn_shape(o.Cin, o.Cout, o.k, o.stride)) return launch_convtn(pp, (int)conv_out_len(o, Tin), s);
        return launch_convt(pp, o.Cin, o.Cout, o.stride, o.pad, (int)conv_out_len(o, Tin), s);
    }
    return launch_conv(make_params(o, x, y, y2, res, acc, acc2, B, Tin, x2, sub, sub_batched), s);
}

// CausalConv1d keeps the first Tin outputs of a conv padded on both sides: only a pad of
// at least (k-1)*dil - which makes that many outputs exist - is meaningful.
int check_pad_mode(int pad_mode, int pad, int k, int dil) {
    if (pad_mode < 0 || pad_mode > (FV_PAD_REFLECT | FV_PAD_CAUSAL))
        return fail(FV_ERR_INVALID_ARG, "unknown pad_mode %d", pad_mode);
    if ((pad_mode & FV_PAD_CAUSAL) && 2LL * pad < (int64_t)dil * (k - 1))
        return fail(FV_ERR_INVALID_ARG, "causal conv: pad=%d leaves fewer than Tin outputs (k=%d dil=%d)",
                    pad, k, dil);
    return 0;
}

int check_conv_args(int Cin, int Cout, int k, int dil) {
    if (Cin <= 0 || Cout <= 0 || k <= 0 || dil <= 0)
        return fail(FV_ERR_INVALID_ARG, "bad conv shape Cin=%d Cout=%d k=%d dil=%d", Cin, Cout, k, dil);
    return 0;
}

// ---- ConvTranspose1d with split-f16 operands (convt_kernel) ----
int check_convt_split_args(int Cin, int Cout, int k, int stride, int pad, int out_pad) {
    if (Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256 && Cin != 512)
        return fail(FV_ERR_UNSUPPORTED, "conv_transpose1d_split_f16: Cin = %d (32, 64, 128, 256 or 512)", Cin);
    if (stride < 2 || stride > 16 || k != 2 * stride)
        return fail(FV_ERR_UNSUPPORTED, "conv_transpose1d_split_f16: kernel %d, stride %d (kernel = 2 x stride, stride 2..16)", k, stride);
    if (Cout <= 0 || Cout * stride < 32)
        return fail(FV_ERR_UNSUPPORTED, "conv_transpose1d_split_f16: Cout * stride = %d (32 or more)", Cout * stride);
    if (pad < 0 || pad > stride || out_pad < -stride || out_pad >= stride)
        return fail(FV_ERR_INVALID_ARG, "conv_transpose1d_split_f16: pad=%d (0..stride) out_pad=%d", pad, out_pad);
    if (convtn_shape(Cin, Cout, k, stride) && (pad != 1 || out_pad != 0))
        return fail(FV_ERR_UNSUPPORTED, "conv_transpose1d_split_f16: 32 -> 16 channels, kernel 4, stride 2 exists with pad=1, "
                    "out_pad=0 only (got %d, %d)", pad, out_pad);
    return 0;
}

}  // namespace fv

using namespace fv;

namespace fv {
int allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    struct Grant { const void* kernel; int device; size_t bytes; };
    static std::mutex mu;
    static std::vector<Grant> granted;                               // a few dozen kernels per device at most
    int dev = 0;
    FV_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (Grant& g : granted)
        if (g.kernel == kernel && g.device == dev) {
            if (g.bytes >= bytes) return 0;
            FV_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            g.bytes = bytes;
            return 0;
        }
    FV_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    granted.push_back({kernel, dev, bytes});
    return 0;
}

// Tuning switches (fv_internal.h struct Tuning).  Defaults unless the process was started with FV_TUNING=1, in which
// case the FV_* variables of the table below are read ONCE, here; after that the launch path never touches the
// environment.  fv_tuning_set changes an entry at run time (tests: different block counts / schedules must give the
// same bits).
static Tuning g_tuning;
static std::once_flag g_tuning_once;
struct TuningEntry {
    const char* key;      // fv_tuning_set key; the environment variable is "FV_" + upper case
    int Tuning::*field;
};
static const TuningEntry kTuningTable[] = {
    {"pair_dbg", &Tuning::pair_dbg},       {"dbg", &Tuning::conv_dbg},           {"sched", &Tuning::sched},
    {"sched_switch", &Tuning::sched_switch}, {"convh_skel", &Tuning::convh_skel}, {"convp_skel", &Tuning::convp_skel},
    {"convq_skel", &Tuning::convq_skel},   {"pair128_unfused", &Tuning::pair128_unfused},
    {"convg_rows64", &Tuning::convg_rows64}, {"stack_items", &Tuning::stack_items}, {"stack_wide", &Tuning::stack_wide},
    {"convp_wide", &Tuning::convp_wide},   {"convq_wide", &Tuning::convq_wide},  {"convp_pp", &Tuning::convp_pp},
    {"convh_rows64", &Tuning::convh_rows64},  {"convt_rows64", &Tuning::convt_rows64},
    {"pairh_skel", &Tuning::pairh_skel},   {"pair_skel", &Tuning::pair_skel},    {"convh_blocks", &Tuning::convh_blocks},
    {"pair_blocks", &Tuning::pair_blocks}, {"sum3_min", &Tuning::sum3_min},      {"lds_budget", &Tuning::lds_budget},
    {"units", &Tuning::units},             {"shape16", &Tuning::shape16},        {"shape32", &Tuning::shape32},
    {"shape64", &Tuning::shape64},         {"krows", &Tuning::krows},            {"grid_cap", &Tuning::grid_cap},
    {"no_group", &Tuning::no_group},       {"mrf_blocks", &Tuning::mrf_blocks},  {"mrf_shape", &Tuning::mrf_shape},
    {"mrf_prio", &Tuning::mrf_prio},       {"convt_lean", &Tuning::convt_lean},
    {"convs_ringfree", &Tuning::convs_ringfree}, {"convu_resident", &Tuning::convu_resident},
};
static void tuning_from_env() {
    const char* on = getenv("FV_TUNING");
    if (!on || atoi(on) != 1) return;
    for (const TuningEntry& e : kTuningTable) {
        char name[64] = "FV_";
        size_t n = 3;
        for (const char* c = e.key; *c && n + 1 < sizeof(name); ++c) name[n++] = (char)toupper((unsigned char)*c);
        name[n] = 0;
        const char* v = getenv(name);
        if (v && *v) g_tuning.*(e.field) = atoi(v);
    }
    const char* tp = getenv("FV_PAIR_TRACE_PTR");
    if (tp && *tp) g_tuning.trace_ptr = strtoull(tp, nullptr, 0);
}
const Tuning& tuning() {
    std::call_once(g_tuning_once, tuning_from_env);
    return g_tuning;
}

int device_cu_count() {
    static int cus = 0;   // one device model per process (the boxes hold eight identical GPUs)
    if (!cus) {
        int dev = 0, v = 0;
        cus = hipGetDevice(&dev) == hipSuccess &&
                      hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0
                  ? v : 256;
    }
    return cus;
}
}  // namespace fv

extern "C" {

int fv_version(void) { return FV_ABI_VERSION; }


#ifndef FV_BUILD_ID
#define FV_BUILD_ID "unknown"
#endif
// (the "fv-build-id:" tag lets the id be read from the file without loading it: _native.built_id)
const char* fv_build_id(void) { return "fv-build-id:" FV_BUILD_ID + 12; }

const char* fv_last_error(void) { return g_err.c_str(); }

int fv_upsample_conv1d_fused(const float* x, const float* packed, const float* bias, float* y,
                             float* y_act, int B, int Cin, int Cout, int Tin, int k, int rate, int pad,
                             float pre_slope, int post, float act_slope, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (!x || !packed || !y) return fail(FV_ERR_INVALID_ARG, "upsample_conv1d: null tensor");
    if (rate <= 0 || pad < 0) return fail(FV_ERR_INVALID_ARG, "upsample_conv1d: rate=%d pad=%d", rate, pad);
    if (x == y || x == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "upsample_conv1d: y / y_act must not alias x or each other");
    Op o = {};
    o.type = OP_UPCONV;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin;
    o.Cout = Cout;
    o.k = k;
    o.stride = rate;
    o.pad = pad;
    o.pre_slope = pre_slope;
    o.out_div = 1.f;
    o.act_slope = act_slope;
    o.post = post;
    if (conv_out_len(o, Tin) <= 0) return fail(FV_ERR_INVALID_ARG, "upsample_conv1d: empty output");
    return run_op(o, x, y, y_act, nullptr, nullptr, nullptr, B, Tin, (hipStream_t)stream);
}

int fv_conv1d_fused(const float* x, const float* packed, const float* bias, const float* res,
                    const float* acc_in, const float* acc_in2, float* y, float* y_act, int B, int Cin,
                    int Cout, int Tin, int k, int dil, int pad, int pad_mode, float pre_slope,
                    float out_div, int post, float act_slope, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, dil)) return rc;
    if (int rc = check_pad_mode(pad_mode, pad, k, dil)) return rc;
    if (!x || !packed || !y) return fail(FV_ERR_INVALID_ARG, "conv1d: null tensor");
    if (x == y || x == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "conv1d: y / y_act must not alias x or each other");
    Op o = {};
    o.type = OP_CONV;
    o.act_slope = act_slope;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin;
    o.Cout = Cout;
    o.k = k;
    o.dil = dil;
    o.pad = pad;
    o.pad_mode = pad_mode;
    o.pre_slope = pre_slope;
    o.out_div = out_div;
    o.post = post;
    if (conv_out_len(o, Tin) <= 0) return fail(FV_ERR_INVALID_ARG, "conv1d: empty output");
    return run_op(o, x, y, y_act, res, acc_in, acc_in2, B, Tin, (hipStream_t)stream);
}

int fv_conv_transpose1d_fused(const float* x, const float* packed, const float* bias, float* y,
                              float* y_act, int B, int Cin, int Cout, int Tin, int k, int stride,
                              int pad, int out_pad, float pre_slope, int post, float act_slope,
                              void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (!x || !packed || !y) return fail(FV_ERR_INVALID_ARG, "conv_transpose1d: null tensor");
    // out_pad in [-stride, stride): negative values trim the tail (CausalConvTranspose1d, modules.py:297-317: -stride)
    if (stride <= 0 || pad < 0 || out_pad < -stride || out_pad >= stride + (stride == 1))
        return fail(FV_ERR_INVALID_ARG, "conv_transpose1d: stride=%d pad=%d out_pad=%d", stride, pad, out_pad);
    if (x == y || x == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "conv_transpose1d: y / y_act must not alias x or each other");
    Op o = {};
    o.type = OP_CONVT;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin;
    o.Cout = Cout;
    o.k = k;
    o.stride = stride;
    o.pad = pad;
    o.out_pad = out_pad;
    o.pre_slope = pre_slope;
    o.out_div = 1.f;
    o.act_slope = act_slope;
    o.post = post;
    if (conv_out_len(o, Tin) <= 0) return fail(FV_ERR_INVALID_ARG, "conv_transpose1d: empty output");
    return run_op(o, x, y, y_act, nullptr, nullptr, nullptr, B, Tin, (hipStream_t)stream);
}

int fv_basis_ola(const float* weight, const float* packed_basis, float* out, int B, int C, int F, int L, void* stream) {
    if (L < 2 || L % 2 != 0) return fail(FV_ERR_INVALID_ARG, "basis_ola: L=%d (even, >= 2)", L);
    return fv_conv_transpose1d_fused(weight, packed_basis, nullptr, out, nullptr, B, C, 1, F, L, L / 2, 0, 0, 1.f, FV_POST_NONE,
                                     1.f, stream);
}

int fv_generator_run(fv_plan_t* plan, int B, int T, const float* mel, float* out, void* workspace, int64_t workspace_bytes,
                     void* stream) {
    return fv_plan_run(plan, B, T, mel, out, workspace, workspace_bytes, stream);
}

int fv_conv_transpose1d_split_f16(const float* x, const float* packed, const float* bias, float* y, float* y_act, int B,
                                  int Cin, int Cout, int Tin, int k, int stride, int pad, int out_pad, float pre_slope,
                                  float act_slope, int* guard, void* stream) {
    if (!x || !packed || !y) return fail(FV_ERR_INVALID_ARG, "conv_transpose1d_split_f16: null tensor");
    if (int rc = check_convt_split_args(Cin, Cout, k, stride, pad, out_pad)) return rc;
    if (x == y || x == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "conv_transpose1d_split_f16: y / y_act must not alias x or each other");
    Op o = {};
    o.type = OP_CONVT;
    o.prec = FV_PAIR_SPLIT_F16;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin;
    o.Cout = Cout;
    o.k = k;
    o.stride = stride;
    o.pad = pad;
    o.out_pad = out_pad;
    o.pre_slope = pre_slope;
    o.out_div = 1.f;
    o.act_slope = act_slope;
    if (B < 0 || Tin < 0) return fail(FV_ERR_INVALID_ARG, "conv_transpose1d_split_f16: B=%d Tin=%d", B, Tin);
    if (B == 0 || Tin == 0) return 0;
    if (conv_out_len(o, Tin) <= 0) return fail(FV_ERR_INVALID_ARG, "conv_transpose1d_split_f16: empty output");
    return run_op(o, x, y, y_act, nullptr, nullptr, nullptr, B, Tin, (hipStream_t)stream, nullptr, nullptr, 0, guard);
}

int fv_pqmf_synthesis(const float* x, const float* h, float* y, int B, int S, int ntaps, int Tsub,
                      void* stream) {
    if (!x || !h || !y || B < 0 || S <= 0 || ntaps <= 0 || ntaps % 2 == 0 || Tsub < 0)
        return fail(FV_ERR_INVALID_ARG, "pqmf: B=%d S=%d ntaps=%d Tsub=%d", B, S, ntaps, Tsub);
    return launch_pqmf(x, h, y, nullptr, nullptr, 0, B, S, ntaps, Tsub, (hipStream_t)stream);
}

int fv_pqmf_analysis(const float* x, const float* h, float* y, int B, int S, int ntaps, int64_t T,
                     void* stream) {
    if (!x || !h || !y || B < 0 || S <= 0 || ntaps <= 0 || ntaps % 2 == 0 || T < S)
        return fail(FV_ERR_INVALID_ARG, "pqmf analysis: B=%d S=%d ntaps=%d T=%lld", B, S, ntaps, (long long)T);
    return launch_pqmf_analysis(x, h, y, B, S, ntaps, T, (hipStream_t)stream);
}

int fv_encode_16bits(float* x, int16_t* out, float* peak, int B, int64_t n, float rescale_out,
                     int scale_in_place, void* stream) {
    if (!x || !out || !peak || B < 0 || n < 0)
        return fail(FV_ERR_INVALID_ARG, "encode_16bits: null tensor or B=%d n=%lld", B, (long long)n);
    return launch_encode16(x, B, n, rescale_out, reinterpret_cast<short*>(out),
                           reinterpret_cast<unsigned*>(peak), scale_in_place, (hipStream_t)stream);
}

int fv_conv1d_2src_fused(const float* x, const float* x2, const float* packed, const float* bias,
                         const float* res, float* y, float* y_act, int B, int Cin1, int Cin2, int Cout,
                         int T, int post, float act_slope, void* stream) {
    if (Cin1 <= 0 || Cin2 <= 0) return fail(FV_ERR_INVALID_ARG, "conv1d_2src: Cin1=%d Cin2=%d", Cin1, Cin2);
    if (int rc = check_conv_args(Cin1 + Cin2, Cout, 1, 1)) return rc;
    if (!x || !x2 || !packed || !y) return fail(FV_ERR_INVALID_ARG, "conv1d_2src: null tensor");
    if (x == y || x2 == y || x == y_act || x2 == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "conv1d_2src: y / y_act must not alias an input or each other");
    if (T <= 0) return fail(FV_ERR_INVALID_ARG, "conv1d_2src: empty output");
    Op o = {};
    o.type = OP_CONV;
    o.act_slope = act_slope;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin1 + Cin2;
    o.Cin1 = Cin1;
    o.Cout = Cout;
    o.k = 1;
    o.dil = 1;
    o.pre_slope = 1.f;
    o.out_div = 1.f;
    o.post = post;
    return run_op(o, x, y, y_act, res, nullptr, nullptr, B, T, (hipStream_t)stream, x2);
}

int fv_conv1x1_2src_split_f16(const float* x, const float* x2, const float* packed, const float* bias, const float* res,
                              float* y, float* y_act, int B, int C, int T, float pre_slope, int post, float act_slope,
                              int* guard, void* stream) {
    if (!x || !x2 || !packed || !y) return fail(FV_ERR_INVALID_ARG, "conv1x1_2src_split_f16: null tensor");
    if (x == y || x2 == y || x == y_act || x2 == y_act || (y_act && y_act == y) || (res && (res == y || res == y_act)))
        return fail(FV_ERR_INVALID_ARG, "conv1x1_2src_split_f16: y / y_act must not alias an input or each other");
    if (B < 0 || T < 0) return fail(FV_ERR_INVALID_ARG, "conv1x1_2src_split_f16: B=%d T=%d", B, T);
    Op o = {};
    o.type = OP_CONVG;
    o.wp = packed;
    o.bias = bias;
    o.Cin = 2 * C;
    o.Cin1 = C;
    o.Cout = C;
    o.k = 1;
    o.dil = 1;
    o.pre_slope = pre_slope;
    o.out_div = 1.f;
    o.post = post;
    o.act_slope = act_slope;
    return run_op(o, x, y, y_act, res, nullptr, nullptr, B, T, (hipStream_t)stream, x2, nullptr, 0, guard);
}

}  // extern "C"

namespace fv {
int check_stack_args(int C, int k, int dil, int pad_mode, float slope, float act_slope, int post) {
    if (post != FV_POST_NONE && post != FV_POST_TANH && post != FV_POST_RELU)
        return fail(FV_ERR_INVALID_ARG, "residual_stack_split_f16: post op %d", post);
    if (!convk_shape(C, k, dil))
        return fail(FV_ERR_UNSUPPORTED, "residual_stack_split_f16: C = %d, k = %d, dilation %d (32 / 64 / 128 / 256 channels, 3 taps, "
                    "dilation 1, 3 or 9)", C, k, dil);
    if (pad_mode != FV_PAD_ZERO && pad_mode != FV_PAD_REFLECT)
        return fail(FV_ERR_UNSUPPORTED, "residual_stack_split_f16: pad_mode %d (zero or reflection padding of the 'same' conv)", pad_mode);
    if (slope < 0.f || slope > 1.f || act_slope < 0.f || act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "residual_stack_split_f16: activation slope outside [0, 1]");
    return 0;
}
}  // namespace fv

extern "C" {

int fv_residual_stack_split_f16(const float* x, const float* packed, const float* bias_dilated, const float* bias_out, float* y,
                                float* y_act, int B, int C, int T, int k, int dil, float slope, int pad_mode, int post,
                                float act_slope, int* guard, void* stream) {
    if (!x || !packed || !y) return fail(FV_ERR_INVALID_ARG, "residual_stack_split_f16: null tensor");
    if (x == y || x == y_act || (y_act && y_act == y))
        return fail(FV_ERR_INVALID_ARG, "residual_stack_split_f16: y / y_act must not alias x or each other");
    if (B < 0 || T < 0) return fail(FV_ERR_INVALID_ARG, "residual_stack_split_f16: B=%d T=%d", B, T);
    if (int rc = check_stack_args(C, k, dil, pad_mode, slope, act_slope, post)) return rc;
    Op o = {};
    o.type = OP_STACK;
    o.prec = FV_PAIR_SPLIT_F16;
    o.wp = packed;
    o.bias = bias_dilated;
    o.bias2 = bias_out;
    o.Cin = o.Cout = C;
    o.k = k;
    o.dil = dil;
    o.pad = dil * (k - 1) / 2;
    o.pad_mode = pad_mode;
    o.pre_slope = slope;
    o.out_div = 1.f;
    o.post = post;
    o.act_slope = act_slope;
    return run_op(o, x, y, y_act, nullptr, nullptr, nullptr, B, T, (hipStream_t)stream, nullptr, nullptr, 0, guard);
}

int fv_conv_post_pqmf(const float* x, const float* packed, const float* bias, const float* h, float* y, int B, int Cin,
                      int S, int T, int k, int pad, float pre_slope, int post, int ntaps, void* stream) {
    if (!x || !packed || !h || !y) return fail(FV_ERR_INVALID_ARG, "conv_post_pqmf: null tensor");
    if (S != 4 || ntaps != 63) return fail(FV_ERR_UNSUPPORTED, "conv_post_pqmf: S=%d ntaps=%d (4 sub-bands, 63 taps)", S, ntaps);
    if (int rc = check_conv_args(Cin, S, k, 1)) return rc;
    // a 'same' conv: y holds S * T samples per utterance (a larger pad would make the kernel store past it)
    if (k % 2 != 1 || pad != (k - 1) / 2)
        return fail(FV_ERR_INVALID_ARG, "conv_post_pqmf: k=%d pad=%d (odd kernel, pad = (k - 1) / 2)", k, pad);
    Op o = {};
    o.type = OP_CONV;
    o.act_slope = 1.f;
    o.wp = packed;
    o.bias = bias;
    o.Cin = Cin;
    o.Cout = S;
    o.k = k;
    o.dil = 1;
    o.pad = pad;
    o.pre_slope = pre_slope;
    o.out_div = 1.f;
    o.post = post;
    o.pq_h = h;
    o.pq_taps = ntaps;
    if (B <= 0 || T <= 0) return 0;
    if (conv_out_len(o, T) <= 0) return fail(FV_ERR_INVALID_ARG, "conv_post_pqmf: empty output");
    return run_op(o, x, y, nullptr, nullptr, nullptr, nullptr, B, T, (hipStream_t)stream);
}

}  // extern "C"

namespace fv {
// A pair at C >= 64 (split-f16 arithmetic only): conv1 of every member in one launch, then conv2 + residual
// (+ the MRF addends); the members' intermediates go through mid[j]
int launch_wide_pairs(const PairParams& pp, float* const* mid, int C, int dil, hipStream_t s) {
    PairParams c1 = pp, c2 = pp;
    c1.act_slope = 1.f;
    c1.out_div = 1.f;
    c1.post = FV_POST_NONE;
    for (int j = 0; j < pp.n_members; ++j) {
        if (!mid || !mid[j]) return fail(FV_ERR_INVALID_ARG, "resblock pair: C = %d needs a scratch tensor per member (mid)", C);
        if (mid[j] == pp.m[j].x || mid[j] == pp.m[j].y || mid[j] == pp.m[j].y_act || mid[j] == pp.m[j].add1 ||
            mid[j] == pp.m[j].add2)
            return fail(FV_ERR_INVALID_ARG, "resblock pair: mid aliases another tensor of member %d", j);
        PairMember& a = c1.m[j];
        a.y = mid[j];
        a.y_act = nullptr;
        a.res = a.add1 = a.add2 = nullptr;
        PairMember& b = c2.m[j];
        b.x = mid[j];
        b.w1 = pp.m[j].w2;
        b.b1 = pp.m[j].b2;
        b.res = pp.m[j].x;
    }
    if (int rc = launch_convh(c1, C, dil, s)) return rc;
    return launch_convh(c2, C, 1, s);
}


int check_pair_args(int n, int C, const int* k, int dil, int prec) {
    if (n < 1 || n > 3) return fail(FV_ERR_INVALID_ARG, "resblock pair: %d members (1..3)", n);
    if (C != 16 && C != 32 && !(prec == FV_PAIR_SPLIT_F16 && (C == 64 || C == 128 || C == 256 || C == 512)))
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: C = %d (16 or 32; 64 ... 512 with split-f16 operands); use the conv1d ops", C);
    if (dil != 1 && dil != 3 && dil != 5) return fail(FV_ERR_UNSUPPORTED, "resblock pair: dilation %d (1, 3 or 5)", dil);
    for (int j = 0; j < n; ++j)
        if (k[j] != 3 && k[j] != 7 && k[j] != 11) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", k[j]);
    return 0;
}
}  // namespace fv

extern "C" {

int fv_resblock1_fused(int n, const float* const* x, const float* const* w1, const float* const* w2,
                       const float* const* b1, const float* const* b2, float* const* y, float* const* y_act,
                       const int* k, int B, int C, int T, int dil, float slope, float act_slope, void* stream) {
    return fv_resblock1_fused_ex(n, x, w1, w2, b1, b2, y, y_act, nullptr, nullptr, nullptr, k, B, C, T, dil, slope,
                                 1.f, FV_POST_NONE, act_slope, FV_PAIR_F32, nullptr, stream);
}

int fv_resblock1_fused_ex(int n, const float* const* x, const float* const* w1, const float* const* w2,
                          const float* const* b1, const float* const* b2, float* const* y, float* const* y_act,
                          float* const* mid, const float* const* add1, const float* const* add2, const int* k, int B,
                          int C, int T, int dil, float slope, float out_div, int post, float act_slope, int prec,
                          int* guard, void* stream) {
    if (!x || !w1 || !w2 || !y || !k) return fail(FV_ERR_INVALID_ARG, "resblock1_fused: null argument");
    if (int rc = check_pair_args(n, C, k, dil, prec)) return rc;
    PairParams pp = {};
    pp.n_members = n;
    pp.B = B;
    pp.T = T;
    pp.slope = slope;
    pp.act_slope = act_slope;
    pp.out_div = out_div;
    pp.post = post;
    pp.prec = prec;
    pp.guard = prec == FV_PAIR_SPLIT_F16 ? guard : nullptr;
    for (int j = 0; j < n; ++j) {
        if (!x[j] || !y[j] || x[j] == y[j] || (y_act && y_act[j] && (y_act[j] == y[j] || y_act[j] == x[j])))
            return fail(FV_ERR_INVALID_ARG, "resblock1_fused: member %d: null tensor, or y / y_act aliases x or each other", j);
        PairMember& mb = pp.m[j];
        mb.add1 = add1 ? add1[j] : nullptr;
        mb.add2 = add2 ? add2[j] : nullptr;
        if ((mb.add1 && (mb.add1 == y[j] || (y_act && mb.add1 == y_act[j]))) ||
            (mb.add2 && (mb.add2 == y[j] || (y_act && mb.add2 == y_act[j]))))
            return fail(FV_ERR_INVALID_ARG, "resblock1_fused: member %d: add1 / add2 alias an output", j);
        mb.x = x[j];
        mb.w1 = w1[j];
        mb.w2 = w2[j];
        mb.b1 = b1 ? b1[j] : nullptr;
        mb.b2 = b2 ? b2[j] : nullptr;
        mb.y = y[j];
        mb.y_act = y_act ? y_act[j] : nullptr;
        mb.k = k[j];
    }
    if (C == 64) return launch_convp(pp, dil, (hipStream_t)stream);
    if (C == 128 && !tuning().pair128_unfused) return launch_convq(pp, dil, (hipStream_t)stream);
    if (C >= 64) return launch_wide_pairs(pp, mid, C, dil, (hipStream_t)stream);
    return launch_pairs(pp, C, dil, (hipStream_t)stream);
}

int fv_mrf_stage(const float* const* x, const float* const* w1, const float* const* w2, const float* const* b1,
                 const float* const* b2, float* y, float* y_act, const int* k, int B, int C, int T, int dil,
                 float slope, float out_div, int post, float act_slope, void* stream) {
    if (!x || !w1 || !w2 || !y || !k) return fail(FV_ERR_INVALID_ARG, "mrf_stage: null argument");
    if (int rc = check_pair_args(3, C, k, dil)) return rc;
    PairParams pp = {};
    pp.n_members = 3;
    pp.sum = 1;
    pp.B = B;
    pp.T = T;
    pp.slope = slope;
    pp.act_slope = act_slope;
    pp.out_div = out_div;
    pp.post = post;
    for (int j = 0; j < 3; ++j) {
        if (!x[j] || x[j] == y || x[j] == y_act || (y_act && y_act == y))
            return fail(FV_ERR_INVALID_ARG, "mrf_stage: member %d: null input, or y / y_act aliases an input or each other", j);
        PairMember& mb = pp.m[j];
        mb.x = x[j];
        mb.w1 = w1[j];
        mb.w2 = w2[j];
        mb.b1 = b1 ? b1[j] : nullptr;
        mb.b2 = b2 ? b2[j] : nullptr;
        mb.y = y;
        mb.y_act = y_act;
        mb.k = k[j];
    }
    return launch_pairs(pp, C, dil, (hipStream_t)stream);
}

}  // extern "C"

namespace fv {
int check_stage_args(int C, const int* k, const int* dil, float slope, float act_slope, int post) {
    if (!k || !dil) return fail(FV_ERR_INVALID_ARG, "mrf stage: null taps / dilations");
    if (!mrf_stage_shape(C, k, dil))
        return fail(FV_ERR_UNSUPPORTED, "mrf stage: C = %d, taps (%d, %d, %d), dilations (%d, %d, %d): built for 16 / 32 channels, "
                    "taps 3 / 7 / 11, dilations (1, 3, 5)", C, k[0], k[1], k[2], dil[0], dil[1], dil[2]);
    if (slope < 0.f || slope > 1.f || act_slope < 0.f || act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "mrf stage: activation slope outside [0, 1]");
    if (post != FV_POST_NONE && post != FV_POST_TANH && post != FV_POST_RELU) return fail(FV_ERR_INVALID_ARG, "mrf stage: post %d", post);
    return 0;
}
}  // namespace fv

extern "C" {

int64_t fv_mrf_stage_workspace_bytes(int C) { return mrf_workspace_bytes(C); }

int fv_mrf_stage_split_f16(const float* x, const float* packed, float* y, float* y_act, int B, int C, int T, const int* k,
                           const int* dil, float slope, float out_div, int post, float act_slope, const float* fold_w,
                           const float* fold_b, float* fold_y, void* workspace, int64_t workspace_bytes, int* guard,
                           void* stream) {
    if (int rc = check_stage_args(C, k, dil, slope, act_slope, post)) return rc;
    if (B < 0 || T < 0) return fail(FV_ERR_INVALID_ARG, "mrf stage: B=%d T=%d", B, T);
    MrfParams p = {};
    p.x = x;
    p.y = y;
    p.y_act = y_act;
    p.blob = packed;
    for (int j = 0; j < 3; ++j) p.k[j] = k[j];
    p.B = B;
    p.T = T;
    p.slope = slope;
    p.out_div = out_div;
    p.act_slope = act_slope;
    p.post = post;
    p.fold_w = fold_w;
    p.fold_b = fold_b;
    p.fold_y = fold_y;
    p.hist = static_cast<float*>(workspace);
    p.hist_bytes = workspace_bytes;
    p.guard = guard;
    return launch_mrfh(p, C, dil, (hipStream_t)stream);
}

}  // extern "C"

namespace fv {
int check_convh_args(int n, int C, const int* k, int dil, int pad_mode) {
    if (pad_mode != FV_PAD_ZERO && pad_mode != FV_PAD_REFLECT)
        return fail(FV_ERR_UNSUPPORTED, "conv1d_split_f16: pad_mode %d (FV_PAD_ZERO or FV_PAD_REFLECT)", pad_mode);
    if (n < 1 || n > 3) return fail(FV_ERR_INVALID_ARG, "conv1d_split_f16: %d members (1..3)", n);
    if (C != 64 && C != 128 && C != 256 && C != 512)
        return fail(FV_ERR_UNSUPPORTED, "conv1d_split_f16: C = %d (64, 128, 256 or 512)", C);
    if (dil != 1 && dil != 3 && dil != 5 && dil != 9)
        return fail(FV_ERR_UNSUPPORTED, "conv1d_split_f16: dilation %d (1, 3, 5; 9 with 3 taps)", dil);
    for (int j = 0; j < n; ++j)
        if ((k[j] != 3 && k[j] != 7 && k[j] != 11) || (dil == 9 && k[j] != 3))
            return fail(FV_ERR_UNSUPPORTED, "conv1d_split_f16: %d taps at dilation %d (3, 7 or 11; 3 at dilation 9)", k[j], dil);
    return 0;
}
}  // namespace fv

extern "C" {

int fv_conv1d_split_f16(int n, const float* const* x, const float* const* packed, const float* const* bias,
                        const float* const* res, const float* const* add1, const float* const* add2, float* const* y,
                        float* const* y_act, const int* k, int B, int C, int T, int dil, int pad_mode, float pre_slope,
                        float out_div, int post, float act_slope, int* guard, void* stream) {
    if (!x || !packed || !y || !k) return fail(FV_ERR_INVALID_ARG, "conv1d_split_f16: null argument");
    if (int rc = check_convh_args(n, C, k, dil, pad_mode)) return rc;
    PairParams pp = {};
    pp.n_members = n;
    pp.B = B;
    pp.T = T;
    pp.slope = pre_slope;
    pp.act_slope = act_slope;
    pp.out_div = out_div;
    pp.post = post;
    pp.prec = FV_PAIR_SPLIT_F16;
    pp.guard = guard;
    pp.reflect = pad_mode == FV_PAD_REFLECT;
    for (int j = 0; j < n; ++j) {
        PairMember& mb = pp.m[j];
        mb.x = x[j];
        mb.w1 = packed[j];
        mb.b1 = bias ? bias[j] : nullptr;
        mb.res = res ? res[j] : nullptr;
        mb.add1 = add1 ? add1[j] : nullptr;
        mb.add2 = add2 ? add2[j] : nullptr;
        mb.y = y[j];
        mb.y_act = y_act ? y_act[j] : nullptr;
        mb.k = k[j];
        if (!mb.x || !mb.y || mb.x == mb.y || mb.y == mb.res || mb.y == mb.add1 || mb.y == mb.add2 ||
            (mb.y_act && (mb.y_act == mb.y || mb.y_act == mb.x || mb.y_act == mb.res)))
            return fail(FV_ERR_INVALID_ARG, "conv1d_split_f16: member %d: null tensor or an output aliases an input", j);
    }
    return launch_convh(pp, C, dil, (hipStream_t)stream);
}

int fv_div_probe(unsigned first_bits, int64_t n, float d, unsigned long long* mismatches, void* stream) {
    if (!mismatches || n <= 0) return fail(FV_ERR_INVALID_ARG, "div_probe: null counter / empty range");
    return fv::launch_div_probe(first_bits, (long long)n, d, mismatches, (hipStream_t)stream);
}

int fv_tuning_set(const char* key, int value) {
    if (!key) return fail(FV_ERR_INVALID_ARG, "tuning_set: null key");
    (void)tuning();                         // the environment (FV_TUNING=1) is read first, once
    for (const TuningEntry& e : kTuningTable)
        if (strcmp(e.key, key) == 0) {
            g_tuning.*(e.field) = value;
            return 0;
        }
    return fail(FV_ERR_INVALID_ARG, "tuning_set: unknown key '%s'", key);
}

int fv_debug_pair_schedule(int n_members, const int* n_items, const int* cost, int nblk, int mode, int three_members,
                           unsigned* table) {
    if (!n_items || !cost || !table) return fail(FV_ERR_INVALID_ARG, "debug_pair_schedule: null argument");
    if (n_members < 1 || n_members > 3) return fail(FV_ERR_INVALID_ARG, "debug_pair_schedule: %d members", n_members);
    if (mode != 0 && mode != 1) return fail(FV_ERR_INVALID_ARG, "debug_pair_schedule: mode %d", mode);
    if (nblk < 1 || nblk > (mode == 0 ? fv::kSchedBlocks : 2 * fv::kSchedBlocks))
        return fail(FV_ERR_INVALID_ARG, "debug_pair_schedule: %d blocks", nblk);
    fv::PairParams p = {};
    p.n_members = n_members;
    long long n[3] = {0, 0, 0};
    for (int m = 0; m < n_members; ++m) {
        if (n_items[m] < 0 || cost[m] < 1) return fail(FV_ERR_INVALID_ARG, "debug_pair_schedule: member %d", m);
        p.m[m].n_items = n_items[m];
        p.m[m].cost = cost[m];
        n[m] = n_items[m];
    }
    if (mode == 0) fv::pair_schedule(p, nblk, three_members != 0);
    else fv::pair_cut_schedule(p, nblk, n);
    memcpy(table, p.sched, sizeof(p.sched));
    return p.sched_on;
}

int fv_profile_enable(int on) {
    g_prof_on = on != 0;
    g_prof_need_start = true;
    return 0;
}

// What the measurement adds to a launch: the end-of-launch event record is one more packet on the stream.  A chain
// of n (null kernel, event record) pairs against a chain of n null kernels, per launch.
__global__ void profile_null_kernel() {}

int fv_profile_bracket_cost(void* stream, int n, double* ms_per_bracket) {
    if (n <= 0 || !ms_per_bracket) return fail(FV_ERR_INVALID_ARG, "profile_bracket_cost: n=%d", n);
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev((size_t)n + 4);
    for (hipEvent_t& e : ev) FV_HIP(hipEventCreate(&e));
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, s);
    FV_HIP(hipEventRecord(ev[n], s));
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, s);
        FV_HIP(hipEventRecord(ev[i], s));
    }
    FV_HIP(hipEventRecord(ev[n + 1], s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, s);
    FV_HIP(hipEventRecord(ev[n + 2], s));
    FV_HIP(hipEventSynchronize(ev[n + 2]));
    float with = 0.f, without = 0.f;
    FV_HIP(hipEventElapsedTime(&with, ev[n], ev[n - 1]));
    FV_HIP(hipEventElapsedTime(&without, ev[n + 1], ev[n + 2]));
    for (hipEvent_t& e : ev) (void)hipEventDestroy(e);
    const double cost = ((double)with - (double)without) / n;
    *ms_per_bracket = cost > 0 ? cost : 0;
    return 0;
}

// A dense stream of v_mfma_f32_16x16x32_f16 from registers: the matrix rate the device sustains when nothing else is asked of it.
typedef _Float16 peak_f16x8 __attribute__((ext_vector_type(8)));
typedef float peak_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void profile_mfma_f16_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    // eight operand pairs, one per accumulator, all different per lane and element and of both signs (products of order 1e-3):
    // consecutive MFMAs see different operands, as in a GEMM (the same pair every time would leave the multipliers' inputs
    // unchanged from one instruction to the next -- and the device's clock higher than any real kernel sees)
    peak_f16x8 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (_Float16)(0.03125f * (float)(((lane * 7 + i * 13 + j * 17) % 29) - 14) + 0.001f * (float)j);
            b[j][i] = (_Float16)(0.0078125f * (float)(((lane * 11 + i * 5 + j * 23) % 31) - 15) - 0.0007f * (float)j);
        }
    peak_f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = peak_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8) {          // (iters: rounded up to a multiple of 8 by the host)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[(j + r) & 7], acc[j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int fv_profile_mfma_f16_rate(float* scratch, long long scratch_floats, int launches, int iters, void* stream, double* tflops) {
    if (!scratch || !tflops || launches < 2 || iters < 1)
        return fail(FV_ERR_INVALID_ARG, "profile_mfma_f16_rate: launches=%d iters=%d", launches, iters);
    iters = (iters + 7) / 8 * 8;
    const int blocks = 2 * fv::device_cu_count();
    if (scratch_floats < 512LL * blocks)
        return fail(FV_ERR_INVALID_ARG, "profile_mfma_f16_rate: scratch of %lld floats, %lld needed", scratch_floats, 512LL * blocks);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    FV_HIP(hipEventCreate(&e0));
    FV_HIP(hipEventCreate(&e1));
    const int first = launches / 2;
    for (int i = 0; i < launches; ++i) {
        if (i == first) FV_HIP(hipEventRecord(e0, s));
        hipLaunchKernelGGL(profile_mfma_f16_kernel, dim3(blocks), dim3(512), 0, s, scratch, iters);
    }
    FV_HIP(hipEventRecord(e1, s));
    FV_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FV_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flop = (double)(launches - first) * blocks * 8.0 * iters * 8.0 * 16384.0;
    *tflops = ms > 0.f ? flop / ((double)ms * 1e-3) / 1e12 : 0.0;
    return 0;
}

int fv_profile_collect(int kind, int64_t* launches, double* ms, double* flops, double* bytes) {
    if (int rc = profile_resolve()) return rc;
    double tms = 0, tf = 0, tb = 0;
    int64_t n = 0;
    std::vector<ProfRec> rest;
    for (ProfRec& r : g_prof) {
        if (kind >= 0 && r.kind != kind) {
            rest.push_back(r);
            continue;
        }
        tms += r.ms;
        tf += r.flops;
        tb += r.bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (ms) *ms = tms;
    if (flops) *flops = tf;
    if (bytes) *bytes = tb;
    g_prof.swap(rest);
    return 0;
}

}  // extern "C"
