// C-ABI entry points of libfastvocoder_hip.so (include/fastvocoder_hip.h):
// weight preparation kernels, the fused operators, the plan executor and the
// measurement hook.
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

#include "fv_internal.h"

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

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
enum OpType { OP_CONV = 0, OP_CONVT = 1, OP_PQMF = 2, OP_UPCONV = 3, OP_PAIR = 4, OP_MRFSUM = 5, OP_CONVH = 6, OP_CONVG = 7, OP_STACK = 8, OP_STAGE = 9 };

struct Op {
    int type;
    int x, y, res, acc, y2, acc2;
    int group;     // ops with the same non-zero id are mutually independent: one grouped launch
    const float* wp;
    const float* bias;
    const float* bias2 = nullptr;   // OP_STACK: bias of the 1x1 pair (stack[4] + skip_layer); `bias` is the dilated conv's
    // OP_STACK at 256 channels (fv_plan_set_stack_two_launch): the two-launch form for runs with many tiles -- the dilated
    // conv's fv_pack_pair_weight_ex image, the 1x1 pair's fv_pack_conv1x1_2src_split_f16 image, the hidden tensor's slot
    const float* alt_w1 = nullptr;
    const float* alt_w2 = nullptr;
    int alt_mid = FV_SLOT_NONE;
    int Cin, Cout, k, dil, pad, pad_mode, stride, out_pad;
    float pre_slope, out_div, act_slope;
    int post;
    // two-source conv (1 tap): GEMM rows ci >= Cin1 are read from slot x2 (Cin - Cin1 channels)
    int x2 = FV_SLOT_NONE;
    int Cin1 = 0;
    int own_first = 0;   // association of the MRF sum in the epilogue (ConvParams::own_first)
    // sum3 (fv_plan_add_conv1d_sum3): two more (input, residual, weight, taps) members; this op's own
    // x / res / wp / k / bias (summed) / y are member 0
    bool sum3 = false;
    int xb = FV_SLOT_NONE, xc = FV_SLOT_NONE, resb = FV_SLOT_NONE, resc = FV_SLOT_NONE;
    int tmpb = FV_SLOT_NONE, tmpc = FV_SLOT_NONE;   // [B,C,T] scratch for the two-launch form (few tiles)
    const float* wpb = nullptr;
    const float* wpc = nullptr;
    int kb = 0, kc = 0;
    // fused ResBlock pairs (OP_PAIR: member 0 only; OP_MRFSUM: the three members, inputs x / xb / xc)
    const float* pw1[3] = {nullptr, nullptr, nullptr};
    const float* pw2[3] = {nullptr, nullptr, nullptr};
    const float* pb1[3] = {nullptr, nullptr, nullptr};
    const float* pb2[3] = {nullptr, nullptr, nullptr};
    int pk[3] = {0, 0, 0};
    int sdil[3] = {0, 0, 0};  // OP_STAGE: dilations of the three pair positions (pk: taps of the three ResBlocks; wp: the packed stage)
    void* work = nullptr;     // OP_STAGE, 32 channels: the launch's history slots (caller-owned)
    int64_t work_bytes = 0;
    int prec = 0;             // FV_PAIR_F32 / FV_PAIR_SPLIT_F16
    bool in_merge = false;    // fv_plan_set_input_merge (split-f16 transposed conv): the input is ((x + xb) + xc) / out_div
    // fv_plan_set_pair_output_conv: a 16 -> 1 channel, 7-tap conv folded into the pair; y is ITS output [B, 1, T]
    const float* fold_w = nullptr;
    const float* fold_b = nullptr;
    int sub = FV_SLOT_NONE;   // fv_plan_set_output_offset: auxiliary input subtracted in this op's epilogue
    // fv_plan_add_conv_post_pqmf: an OP_CONV (Cout = S sub-bands) whose launch also runs the PQMF synthesis: y is the
    // FULL-BAND output [B, 1, S * T']
    const float* pq_h = nullptr;
    int pq_taps = 0;
};

struct Shape {
    int C;
    int64_t T;
    bool set;
};

}  // namespace fv

struct fv_plan {
    int in_channels;
    std::vector<fv::Op> ops;
    int cur_group = 0;
    int cur_own_first = 0;
    // range guard of the split-f16 launches (fv_plan_set_guard): a caller-owned word in pinned, device-mapped host
    // memory -- the host's and the device's view of it
    int* guard_host = nullptr;
    int* guard_dev = nullptr;
};

namespace fv {

static int64_t conv_out_len(const Op& o, int64_t Tin) {
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

// Propagate shapes through the op list; fills per-slot max element counts.
static int infer(const fv_plan* plan, int B, int T, Shape* sh, int64_t* slot_elems) {
    for (int i = 0; i < FV_MAX_SLOTS; ++i) {
        sh[i].set = false;
        slot_elems[i] = 0;
    }
    sh[FV_SLOT_IN] = {plan->in_channels, T, true};
    for (size_t n = 0; n < plan->ops.size(); ++n) {
        const Op& o = plan->ops[n];
        if (o.x == FV_SLOT_AUX_IN0 || o.x == FV_SLOT_AUX_IN1 || o.y == FV_SLOT_AUX_IN0 || o.y == FV_SLOT_AUX_IN1 ||
            o.y2 == FV_SLOT_AUX_IN0 || o.y2 == FV_SLOT_AUX_IN1)
            return fail(FV_ERR_INVALID_ARG, "op %zu: the auxiliary input slots can only be output offsets", n);
        if (!sh[o.x].set) return fail(FV_ERR_INVALID_ARG, "op %zu reads unset slot %d", n, o.x);
        const int cin_x = o.x2 == FV_SLOT_NONE ? o.Cin : o.Cin1;
        if (sh[o.x].C != cin_x)
            return fail(FV_ERR_INVALID_ARG, "op %zu: slot %d has %d channels, op expects %d", n,
                        o.x, sh[o.x].C, cin_x);
        if (o.x2 != FV_SLOT_NONE &&
            (!sh[o.x2].set || sh[o.x2].C != o.Cin - o.Cin1 || sh[o.x2].T != sh[o.x].T))
            return fail(FV_ERR_INVALID_ARG, "op %zu: second input slot %d must be [%d, T] like the first", n,
                        o.x2, o.Cin - o.Cin1);
        const int64_t Tout = conv_out_len(o, sh[o.x].T);
        if (Tout <= 0) return fail(FV_ERR_INVALID_ARG, "op %zu: empty output (T=%lld)", n, (long long)sh[o.x].T);
        const int Cout = (o.type == OP_PQMF || o.pq_h) ? 1 : o.Cout;
        const int aux[3] = {o.res, o.acc, o.acc2};
        for (int a = 0; a < 3; ++a) {
            if (aux[a] == FV_SLOT_NONE) continue;
            if (!sh[aux[a]].set || sh[aux[a]].C != Cout || sh[aux[a]].T != Tout)
                return fail(FV_ERR_INVALID_ARG, "op %zu: residual/accumulator slot %d shape mismatch", n, aux[a]);
        }
        if (o.in_merge) {
            const int extra[2] = {o.xb, o.xc};
            for (int e = 0; e < 2; ++e)
                if (extra[e] != FV_SLOT_NONE && (!sh[extra[e]].set || sh[extra[e]].C != o.Cin || sh[extra[e]].T != sh[o.x].T ||
                                                 extra[e] == o.y || extra[e] == o.y2))
                    return fail(FV_ERR_INVALID_ARG, "op %zu: merged input slot %d must be [%d, T] like the first and not the output",
                                n, extra[e], o.Cin);
        }
        if (o.type == OP_MRFSUM) {
            const int extra[2] = {o.xb, o.xc};
            for (int e = 0; e < 2; ++e)
                if (!sh[extra[e]].set || sh[extra[e]].C != o.Cin || sh[extra[e]].T != sh[o.x].T || extra[e] == o.y ||
                    extra[e] == o.y2)
                    return fail(FV_ERR_INVALID_ARG, "op %zu: mrf member slot %d must be [%d, T] and not the output", n,
                                extra[e], o.Cin);
        }
        if (o.sum3) {
            const int extra[4] = {o.xb, o.xc, o.resb, o.resc};
            for (int e = 0; e < 4; ++e)
                if (!sh[extra[e]].set || sh[extra[e]].C != o.Cin || sh[extra[e]].T != sh[o.x].T || extra[e] == o.y ||
                    extra[e] == o.y2)
                    return fail(FV_ERR_INVALID_ARG, "op %zu: sum3 member slot %d must be [%d, T] and not the output",
                                n, extra[e], o.Cin);
        }
        if (o.sum3) {   // the scratch tensors of the two-launch form
            const int t2[2] = {o.tmpb, o.tmpc};
            for (int e = 0; e < 2; ++e) {
                if (t2[e] == o.x || t2[e] == o.xb || t2[e] == o.xc || t2[e] == o.res || t2[e] == o.resb ||
                    t2[e] == o.resc || t2[e] == o.y || t2[e] == o.y2 || t2[e] == FV_SLOT_IN)
                    return fail(FV_ERR_INVALID_ARG, "op %zu: sum3 scratch slot %d aliases an operand", n, t2[e]);
                sh[t2[e]] = {o.Cout, sh[o.x].T, true};
                const int64_t es = (int64_t)B * o.Cout * sh[o.x].T;
                if (es > slot_elems[t2[e]]) slot_elems[t2[e]] = es;
            }
        }
        if (o.type == OP_STACK && o.alt_mid != FV_SLOT_NONE) {   // hidden tensor of the two-launch form
            if (o.alt_mid == o.x || o.alt_mid == o.y || o.alt_mid == o.y2 || o.alt_mid == FV_SLOT_IN)
                return fail(FV_ERR_INVALID_ARG, "op %zu: stack scratch slot %d aliases an operand", n, o.alt_mid);
            sh[o.alt_mid] = {o.Cout, sh[o.x].T, true};
            const int64_t es = (int64_t)B * o.Cout * sh[o.x].T;
            if (es > slot_elems[o.alt_mid]) slot_elems[o.alt_mid] = es;
        }
        if (o.type == OP_PAIR && o.tmpb != FV_SLOT_NONE) {   // intermediate of a two-launch (C >= 64) pair
            if (o.tmpb == o.x || o.tmpb == o.y || o.tmpb == o.y2 || o.tmpb == o.acc || o.tmpb == o.acc2 || o.tmpb == FV_SLOT_IN)
                return fail(FV_ERR_INVALID_ARG, "op %zu: pair scratch slot %d aliases an operand", n, o.tmpb);
            sh[o.tmpb] = {o.Cout, sh[o.x].T, true};
            const int64_t es = (int64_t)B * o.Cout * sh[o.x].T;
            if (es > slot_elems[o.tmpb]) slot_elems[o.tmpb] = es;
        }
        if (o.y == o.x || o.y == o.x2) return fail(FV_ERR_INVALID_ARG, "op %zu: output aliases input", n);
        const int Cy = o.fold_w ? 1 : Cout;
        sh[o.y] = {Cy, Tout, true};
        const int64_t e = (int64_t)B * Cy * Tout;
        if (e > slot_elems[o.y]) slot_elems[o.y] = e;
        if (o.y2 != FV_SLOT_NONE) {
            if (o.y2 == o.x || o.y2 == o.y || o.y2 == o.x2)
                return fail(FV_ERR_INVALID_ARG, "op %zu: activated twin aliases another tensor of the op", n);
            sh[o.y2] = {Cout, Tout, true};
            if (e > slot_elems[o.y2]) slot_elems[o.y2] = e;
        }
    }
    return 0;
}

static ConvParams make_params(const Op& o, const float* x, float* y, float* y2, const float* res,
                              const float* acc, const float* acc2, int B, int64_t Tin,
                              const float* x2 = nullptr, const float* sub = nullptr, int sub_batched = 0) {
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

static int run_op(const Op& o, const float* x, float* y, float* y2, const float* res, const float* acc,
                  const float* acc2, int B, int64_t Tin, hipStream_t s, const float* x2 = nullptr,
                  const float* sub = nullptr, int sub_batched = 0, int* guard = nullptr) {
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
static int check_pad_mode(int pad_mode, int pad, int k, int dil) {
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
    {"convp_wide", &Tuning::convp_wide},   {"convq_wide", &Tuning::convq_wide},
    {"convh_rows64", &Tuning::convh_rows64},  {"convt_rows64", &Tuning::convt_rows64},
    {"pairh_skel", &Tuning::pairh_skel},   {"pair_skel", &Tuning::pair_skel},    {"convh_blocks", &Tuning::convh_blocks},
    {"pair_blocks", &Tuning::pair_blocks}, {"sum3_min", &Tuning::sum3_min},      {"lds_budget", &Tuning::lds_budget},
    {"units", &Tuning::units},             {"shape16", &Tuning::shape16},        {"shape32", &Tuning::shape32},
    {"shape64", &Tuning::shape64},         {"krows", &Tuning::krows},            {"grid_cap", &Tuning::grid_cap},
    {"no_group", &Tuning::no_group},       {"mrf_blocks", &Tuning::mrf_blocks},  {"mrf_shape", &Tuning::mrf_shape},
    {"mrf_prio", &Tuning::mrf_prio},       {"convt_lean", &Tuning::convt_lean},
    {"convs_ringfree", &Tuning::convs_ringfree},
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

int fv_plan_add_upsample_conv1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot,
                                const float* packed, const float* bias, int Cin, int Cout, int k,
                                int rate, int pad, float pre_slope, int post, float act_slope) {
    if (int rc = fv_plan_add_conv_transpose1d(plan, x_slot, y_slot, y_act_slot, packed, bias, Cin, Cout, k,
                                              rate, pad, 0, pre_slope, post, act_slope))
        return rc;
    plan->ops.back().type = OP_UPCONV;
    return 0;
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

fv_plan_t* fv_plan_create(int in_channels) {
    fv_plan* p = new fv_plan();
    p->in_channels = in_channels;
    return p;
}

void fv_plan_destroy(fv_plan_t* plan) {
    delete plan;
}

static int check_slot(int s, bool allow_none) {
    if (s == FV_SLOT_NONE && allow_none) return 0;
    if (s < 0 || s >= FV_MAX_SLOTS) return fail(FV_ERR_INVALID_ARG, "slot %d out of range", s);
    return 0;
}

int fv_plan_add_conv1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, int res_slot,
                       int acc_slot, int acc2_slot, const float* packed, const float* bias, int Cin,
                       int Cout, int k, int dil, int pad, int pad_mode, float pre_slope, float out_div,
                       int post, float act_slope) {
    if (!plan || !packed) return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d: null");
    if (int rc = check_conv_args(Cin, Cout, k, dil)) return rc;
    if (int rc = check_pad_mode(pad_mode, pad, k, dil)) return rc;
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(res_slot, true)) return rc;
    if (int rc = check_slot(acc_slot, true)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (int rc = check_slot(acc2_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN)
        return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_CONV;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.act_slope = act_slope;
    o.res = res_slot;
    o.acc = acc_slot;
    o.acc2 = acc2_slot;
    o.group = plan->cur_group;
    o.own_first = plan->cur_own_first;
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
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_add_conv1d_sum3(fv_plan_t* plan, const int* x_slots, const int* res_slots, const int* tmp_slots,
                            int y_slot, int y_act_slot, const float* const* packed, const float* bias_sum,
                            int C, const int* k, float out_div, int post, float act_slope) {
    if (!plan || !x_slots || !res_slots || !tmp_slots || !packed || !k)
        return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d_sum3: null");
    for (int j = 0; j < 2; ++j)
        if (int rc = check_slot(tmp_slots[j], false)) return rc;
    for (int j = 0; j < 3; ++j) {
        if (!packed[j] || k[j] < 1 || k[j] % 2 == 0)
            return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d_sum3: member %d needs packed weights and an odd tap count", j);
        if (int rc = check_slot(x_slots[j], false)) return rc;
        if (int rc = check_slot(res_slots[j], false)) return rc;
    }
    if (int rc = fv_plan_add_conv1d(plan, x_slots[0], y_slot, y_act_slot, res_slots[0], FV_SLOT_NONE, FV_SLOT_NONE,
                                    packed[0], bias_sum, C, C, k[0], 1, (k[0] - 1) / 2, FV_PAD_ZERO, 1.f, out_div,
                                    post, act_slope))
        return rc;
    Op& o = plan->ops.back();
    o.sum3 = true;
    o.group = 0;
    o.xb = x_slots[1];
    o.xc = x_slots[2];
    o.resb = res_slots[1];
    o.resc = res_slots[2];
    o.wpb = packed[1];
    o.wpc = packed[2];
    o.kb = k[1];
    o.kc = k[2];
    o.tmpb = tmp_slots[0];
    o.tmpc = tmp_slots[1];
    return 0;
}

int fv_plan_add_conv1d_2src(fv_plan_t* plan, int x_slot, int x2_slot, int y_slot, int y_act_slot,
                            int res_slot, const float* packed, const float* bias, int Cin1, int Cin2,
                            int Cout, int post, float act_slope) {
    if (Cin1 <= 0 || Cin2 <= 0) return fail(FV_ERR_INVALID_ARG, "conv1d_2src: Cin1=%d Cin2=%d", Cin1, Cin2);
    if (int rc = check_slot(x2_slot, false)) return rc;
    if (int rc = fv_plan_add_conv1d(plan, x_slot, y_slot, y_act_slot, res_slot, FV_SLOT_NONE, FV_SLOT_NONE,
                                    packed, bias, Cin1 + Cin2, Cout, 1, 1, 0, FV_PAD_ZERO, 1.f, 1.f, post,
                                    act_slope))
        return rc;
    Op& o = plan->ops.back();
    o.x2 = x2_slot;
    o.Cin1 = Cin1;
    o.group = 0;
    return 0;
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

int fv_plan_add_conv1x1_2src_split_f16(fv_plan_t* plan, int x_slot, int x2_slot, int y_slot, int y_act_slot, int res_slot,
                                       const float* packed, const float* bias, int C, float pre_slope, int post,
                                       float act_slope) {
    if (fv_packed_conv1x1_2src_split_floats(C) <= 0)
        return fail(FV_ERR_UNSUPPORTED, "plan_add_conv1x1_2src_split_f16: C = %d (128, 256 or 512)", C);
    if (int rc = fv_plan_add_conv1d_2src(plan, x_slot, x2_slot, y_slot, y_act_slot, res_slot, packed, bias, C, C, C, post,
                                         act_slope))
        return rc;
    Op& o = plan->ops.back();
    o.type = OP_CONVG;
    o.pre_slope = pre_slope;
    o.prec = FV_PAIR_SPLIT_F16;
    return 0;
}

// A residual stack that carries its two-launch form (256 channels): one launch up to Tuning::stack_items tiles per CU (in
// tenths), else two launches on 128-row x 128-column tiles.  [measured, Basis-MelGAN light, 1000 frames, batch 1 / 4 / 16 /
// 64, tools/bench_configs.py --only 3 --batch B --tuning stack_items=0 against the default] 0.385 -> 0.304, 1.03 -> 0.94,
// 3.55 -> 3.35, 13.0 -> 12.45 ms: the one-launch kernel wins at every size, so the default limit is "none"; the switch
// stays for A/B runs and the bit-identity tests
static bool stack_two_launch(int C, int B, int64_t T) {
    const int nm = convk_tile_columns(C);
    const int64_t items = (int64_t)B * ((T + nm - 1) / nm);
    return items * 10 > (int64_t)tuning().stack_items * device_cu_count();
}

static int check_stack_args(int C, int k, int dil, int pad_mode, float slope, float act_slope, int post = FV_POST_NONE) {
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

int fv_plan_add_residual_stack_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed,
                                         const float* bias_dilated, const float* bias_out, int C, int k, int dil, float slope,
                                         int pad_mode, int post, float act_slope) {
    if (!plan || !packed) return fail(FV_ERR_INVALID_ARG, "plan_add_residual_stack_split_f16: null");
    if (int rc = check_stack_args(C, k, dil, pad_mode, slope, act_slope, post)) return rc;
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_STACK;
    o.prec = FV_PAIR_SPLIT_F16;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.res = o.acc = o.acc2 = FV_SLOT_NONE;
    o.group = 0;
    o.wp = packed;
    o.bias = bias_dilated;
    o.bias2 = bias_out;
    o.Cin = o.Cout = C;
    o.k = k;
    o.dil = dil;
    o.pad = dil * (k - 1) / 2;
    o.pad_mode = pad_mode;
    o.stride = 1;
    o.pre_slope = slope;
    o.out_div = 1.f;
    o.post = post;
    o.act_slope = act_slope;
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_set_stack_two_launch(fv_plan_t* plan, int hidden_slot, const float* packed_dilated, const float* packed_pair) {
    if (!plan || plan->ops.empty() || !packed_dilated || !packed_pair)
        return fail(FV_ERR_INVALID_ARG, "plan_set_stack_two_launch: no op / null weights");
    if (int rc = check_slot(hidden_slot, false)) return rc;
    Op& o = plan->ops.back();
    if (o.type != OP_STACK || fv_packed_conv1x1_2src_split_floats(o.Cout) <= 0 || o.Cout < 128)
        return fail(FV_ERR_UNSUPPORTED, "plan_set_stack_two_launch: the last op must be a residual stack of 128 or 256 channels");
    if (hidden_slot == FV_SLOT_IN || hidden_slot == o.x || hidden_slot == o.y || hidden_slot == o.y2)
        return fail(FV_ERR_INVALID_ARG, "plan_set_stack_two_launch: the scratch slot aliases an operand");
    o.alt_w1 = packed_dilated;
    o.alt_w2 = packed_pair;
    o.alt_mid = hidden_slot;
    return 0;
}

int fv_plan_add_conv_transpose1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot,
                                 const float* packed, const float* bias, int Cin, int Cout, int k,
                                 int stride, int pad, int out_pad, float pre_slope, int post,
                                 float act_slope) {
    if (!plan || !packed) return fail(FV_ERR_INVALID_ARG, "plan_add_conv_transpose1d: null");
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (stride <= 0 || pad < 0 || out_pad < -stride)
        return fail(FV_ERR_INVALID_ARG, "convT stride=%d pad=%d out_pad=%d", stride, pad, out_pad);
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN)
        return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_CONVT;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.act_slope = act_slope;
    o.res = FV_SLOT_NONE;
    o.acc = FV_SLOT_NONE;
    o.acc2 = FV_SLOT_NONE;
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
    o.post = post;
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_add_conv_transpose1d_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed,
                                           const float* bias, int Cin, int Cout, int k, int stride, int pad,
                                           int out_pad, float pre_slope, float act_slope) {
    if (int rc = check_convt_split_args(Cin, Cout, k, stride, pad, out_pad)) return rc;
    if (int rc = fv_plan_add_conv_transpose1d(plan, x_slot, y_slot, y_act_slot, packed, bias, Cin, Cout, k, stride, pad,
                                              out_pad, pre_slope, FV_POST_NONE, act_slope))
        return rc;
    plan->ops.back().prec = FV_PAIR_SPLIT_F16;
    return 0;
}

int fv_plan_set_input_merge(fv_plan_t* plan, int add1_slot, int add2_slot, float div) {
    if (!plan || plan->ops.empty()) return fail(FV_ERR_INVALID_ARG, "plan_set_input_merge: no op to attach to");
    Op& o = plan->ops.back();
    if (o.type != OP_CONVT || o.prec != FV_PAIR_SPLIT_F16)
        return fail(FV_ERR_UNSUPPORTED, "plan_set_input_merge: the last op is not a split-f16 transposed conv");
    if (add1_slot == FV_SLOT_NONE || add1_slot < 0 || add1_slot >= FV_MAX_SLOTS ||
        (add2_slot != FV_SLOT_NONE && (add2_slot < 0 || add2_slot >= FV_MAX_SLOTS)))
        return fail(FV_ERR_INVALID_ARG, "plan_set_input_merge: slots %d, %d", add1_slot, add2_slot);
    if (!(div > 0.f)) return fail(FV_ERR_INVALID_ARG, "plan_set_input_merge: divisor %g", (double)div);
    o.in_merge = true;
    o.xb = add1_slot;
    o.xc = add2_slot;
    o.out_div = div;
    return 0;
}

int fv_plan_add_conv_post_pqmf(fv_plan_t* plan, int x_slot, int y_slot, const float* packed, const float* bias, int Cin,
                               int S, int k, int pad, float pre_slope, int post, const float* h, int ntaps) {
    if (!h || S != 4 || ntaps != 63)
        return fail(FV_ERR_UNSUPPORTED, "plan_add_conv_post_pqmf: S=%d ntaps=%d (4 sub-bands, 63 taps)", S, ntaps);
    // a 'same' conv: the kernel writes S * T samples per utterance, the output's size (a larger pad would run past it)
    if (k % 2 != 1 || pad != (k - 1) / 2)
        return fail(FV_ERR_INVALID_ARG, "plan_add_conv_post_pqmf: k=%d pad=%d (odd kernel, pad = (k - 1) / 2)", k, pad);
    if (int rc = fv_plan_add_conv1d(plan, x_slot, y_slot, FV_SLOT_NONE, FV_SLOT_NONE, FV_SLOT_NONE, FV_SLOT_NONE, packed, bias,
                                    Cin, S, k, 1, pad, FV_PAD_ZERO, pre_slope, 1.f, post, 1.f))
        return rc;
    Op& o = plan->ops.back();
    o.group = 0;
    o.pq_h = h;
    o.pq_taps = ntaps;
    return 0;
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

int fv_plan_add_pqmf_synthesis(fv_plan_t* plan, int x_slot, int y_slot, const float* h, int S,
                               int ntaps) {
    if (!plan || !h || S <= 0 || ntaps <= 0 || ntaps % 2 == 0)
        return fail(FV_ERR_INVALID_ARG, "plan_add_pqmf: S=%d ntaps=%d", S, ntaps);
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    Op o = {};
    o.type = OP_PQMF;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = FV_SLOT_NONE;
    o.res = FV_SLOT_NONE;
    o.acc = FV_SLOT_NONE;
    o.acc2 = FV_SLOT_NONE;
    o.wp = h;
    o.Cin = S;
    o.Cout = 1;
    o.k = ntaps;
    plan->ops.push_back(o);
    return 0;
}

// A pair at C >= 64 (split-f16 arithmetic only): conv1 of every member in one launch, then conv2 + residual
// (+ the MRF addends); the members' intermediates go through mid[j]
static int launch_wide_pairs(const PairParams& pp, float* const* mid, int C, int dil, hipStream_t s) {
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

static int check_pair_args(int n, int C, const int* k, int dil, int prec = FV_PAIR_F32) {
    if (n < 1 || n > 3) return fail(FV_ERR_INVALID_ARG, "resblock pair: %d members (1..3)", n);
    if (C != 16 && C != 32 && !(prec == FV_PAIR_SPLIT_F16 && (C == 64 || C == 128 || C == 256 || C == 512)))
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: C = %d (16 or 32; 64 ... 512 with split-f16 operands); use the conv1d ops", C);
    if (dil != 1 && dil != 3 && dil != 5) return fail(FV_ERR_UNSUPPORTED, "resblock pair: dilation %d (1, 3 or 5)", dil);
    for (int j = 0; j < n; ++j)
        if (k[j] != 3 && k[j] != 7 && k[j] != 11) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", k[j]);
    return 0;
}

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

static int check_stage_args(int C, const int* k, const int* dil, float slope, float act_slope, int post) {
    if (!k || !dil) return fail(FV_ERR_INVALID_ARG, "mrf stage: null taps / dilations");
    if (!mrf_stage_shape(C, k, dil))
        return fail(FV_ERR_UNSUPPORTED, "mrf stage: C = %d, taps (%d, %d, %d), dilations (%d, %d, %d): built for 16 / 32 channels, "
                    "taps 3 / 7 / 11, dilations (1, 3, 5)", C, k[0], k[1], k[2], dil[0], dil[1], dil[2]);
    if (slope < 0.f || slope > 1.f || act_slope < 0.f || act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "mrf stage: activation slope outside [0, 1]");
    if (post != FV_POST_NONE && post != FV_POST_TANH && post != FV_POST_RELU) return fail(FV_ERR_INVALID_ARG, "mrf stage: post %d", post);
    return 0;
}

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

int fv_plan_add_mrf_stage_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed, int C,
                                    const int* k, const int* dil, float slope, float out_div, int post, float act_slope,
                                    void* workspace, int64_t workspace_bytes) {
    if (!plan || !packed) return fail(FV_ERR_INVALID_ARG, "plan_add_mrf_stage: null");
    if (int rc = check_stage_args(C, k, dil, slope, act_slope, post)) return rc;
    if (workspace_bytes < mrf_workspace_bytes(C) || (mrf_workspace_bytes(C) > 0 && !workspace))
        return fail(FV_ERR_WORKSPACE, "plan_add_mrf_stage: C = %d needs a workspace of %lld bytes (fv_mrf_stage_workspace_bytes)", C,
                    (long long)mrf_workspace_bytes(C));
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_STAGE;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.res = o.acc = o.acc2 = FV_SLOT_NONE;
    o.group = 0;
    o.Cin = o.Cout = C;
    o.k = k[0];
    o.dil = dil[0];
    for (int j = 0; j < 3; ++j) {
        o.pk[j] = k[j];
        o.sdil[j] = dil[j];
    }
    o.wp = packed;
    o.work = workspace;
    o.work_bytes = workspace_bytes;
    o.pre_slope = slope;
    o.act_slope = act_slope;
    o.out_div = out_div;
    o.post = post;
    o.prec = FV_PAIR_SPLIT_F16;
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_add_resblock_pair(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed1,
                              const float* packed2, const float* bias1, const float* bias2, int C, int k, int dil,
                              float slope, float act_slope) {
    return fv_plan_add_resblock_pair_ex(plan, x_slot, y_slot, y_act_slot, FV_SLOT_NONE, FV_SLOT_NONE, FV_SLOT_NONE,
                                        packed1, packed2, bias1, bias2, C, k, dil, slope, 1.f, FV_POST_NONE, act_slope,
                                        FV_PAIR_F32);
}

int fv_plan_add_resblock_pair_ex(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, int mid_slot, int add1_slot,
                                 int add2_slot, const float* packed1, const float* packed2, const float* bias1,
                                 const float* bias2, int C, int k, int dil, float slope, float out_div, int post,
                                 float act_slope, int prec) {
    if (!plan || !packed1 || !packed2) return fail(FV_ERR_INVALID_ARG, "plan_add_resblock_pair: null");
    if (int rc = check_pair_args(1, C, &k, dil, prec)) return rc;
    if ((C >= 64) != (mid_slot != FV_SLOT_NONE))
        return fail(FV_ERR_INVALID_ARG, "plan_add_resblock_pair: a scratch slot is needed at C >= 64 and only there");
    if (int rc = check_slot(mid_slot, true)) return rc;
    if (prec != FV_PAIR_F32 && prec != FV_PAIR_SPLIT_F16)
        return fail(FV_ERR_INVALID_ARG, "plan_add_resblock_pair: unknown arithmetic %d", prec);
    if (prec == FV_PAIR_F32 && (add1_slot != FV_SLOT_NONE || add2_slot != FV_SLOT_NONE || out_div != 1.f || post != FV_POST_NONE))
        return fail(FV_ERR_UNSUPPORTED, "plan_add_resblock_pair: add1 / add2 / out_div / post exist with FV_PAIR_SPLIT_F16 only");
    if (add2_slot != FV_SLOT_NONE && add1_slot == FV_SLOT_NONE) return fail(FV_ERR_INVALID_ARG, "plan_add_resblock_pair: add2 without add1");
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (int rc = check_slot(add1_slot, true)) return rc;
    if (int rc = check_slot(add2_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_PAIR;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.res = FV_SLOT_NONE;
    o.acc = add1_slot;      // the MRF addends travel in the running-sum fields (dependencies, shape checks)
    o.acc2 = add2_slot;
    o.tmpb = mid_slot;
    o.group = plan->cur_group;
    o.Cin = o.Cout = C;
    o.k = k;
    o.dil = dil;
    o.pre_slope = slope;
    o.act_slope = act_slope;
    o.out_div = out_div;
    o.post = post;
    o.prec = prec;
    o.pw1[0] = packed1;
    o.pw2[0] = packed2;
    o.pb1[0] = bias1;
    o.pb2[0] = bias2;
    o.pk[0] = k;
    plan->ops.push_back(o);
    return 0;
}

static int check_convh_args(int n, int C, const int* k, int dil, int pad_mode) {
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

int fv_plan_add_conv1d_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, int res_slot, int add1_slot,
                                 int add2_slot, const float* packed, const float* bias, int C, int k, int dil,
                                 int pad_mode, float pre_slope, float out_div, int post, float act_slope) {
    if (!plan || !packed) return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d_split_f16: null");
    if (int rc = check_convh_args(1, C, &k, dil, pad_mode)) return rc;
    if (add2_slot != FV_SLOT_NONE && add1_slot == FV_SLOT_NONE) return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d_split_f16: add2 without add1");
    if (int rc = check_slot(x_slot, false)) return rc;
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (int rc = check_slot(res_slot, true)) return rc;
    if (int rc = check_slot(add1_slot, true)) return rc;
    if (int rc = check_slot(add2_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    if (y_slot == res_slot || y_slot == add1_slot || y_slot == add2_slot)
        return fail(FV_ERR_INVALID_ARG, "plan_add_conv1d_split_f16: the output aliases an addend");
    Op o = {};
    o.type = OP_CONVH;
    o.x = x_slot;
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.res = res_slot;
    o.acc = add1_slot;
    o.acc2 = add2_slot;
    o.group = plan->cur_group;
    o.Cin = o.Cout = C;
    o.k = k;
    o.dil = dil;
    o.pad_mode = pad_mode;
    o.pre_slope = pre_slope;
    o.act_slope = act_slope;
    o.out_div = out_div;
    o.post = post;
    o.prec = FV_PAIR_SPLIT_F16;
    o.pw1[0] = packed;
    o.pb1[0] = bias;
    o.pk[0] = k;
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_add_mrf_sum(fv_plan_t* plan, const int* x_slots, int y_slot, int y_act_slot,
                        const float* const* packed1, const float* const* packed2, const float* const* bias1,
                        const float* const* bias2, int C, const int* k, int dil, float slope, float out_div,
                        int post, float act_slope) {
    if (!plan || !x_slots || !packed1 || !packed2 || !k) return fail(FV_ERR_INVALID_ARG, "plan_add_mrf_sum: null");
    if (int rc = check_pair_args(3, C, k, dil)) return rc;
    if (C != 16) return fail(FV_ERR_UNSUPPORTED, "plan_add_mrf_sum: C = %d (16)", C);
    for (int j = 0; j < 3; ++j) {
        if (!packed1[j] || !packed2[j]) return fail(FV_ERR_INVALID_ARG, "plan_add_mrf_sum: member %d has no weights", j);
        if (int rc = check_slot(x_slots[j], false)) return rc;
    }
    if (int rc = check_slot(y_slot, false)) return rc;
    if (int rc = check_slot(y_act_slot, true)) return rc;
    if (y_slot == FV_SLOT_IN || y_act_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan: the input slot is read-only");
    Op o = {};
    o.type = OP_MRFSUM;
    o.x = x_slots[0];
    o.xb = x_slots[1];
    o.xc = x_slots[2];
    o.y = y_slot;
    o.y2 = y_act_slot;
    o.res = o.acc = o.acc2 = FV_SLOT_NONE;
    o.Cin = o.Cout = C;
    o.k = k[0];
    o.dil = dil;
    o.pre_slope = slope;
    o.act_slope = act_slope;
    o.out_div = out_div;
    o.post = post;
    for (int j = 0; j < 3; ++j) {
        o.pw1[j] = packed1[j];
        o.pw2[j] = packed2[j];
        o.pb1[j] = bias1 ? bias1[j] : nullptr;
        o.pb2[j] = bias2 ? bias2[j] : nullptr;
        o.pk[j] = k[j];
    }
    plan->ops.push_back(o);
    return 0;
}

int fv_plan_set_output_offset(fv_plan_t* plan, int aux_slot, int y2_slot) {
    if (!plan || plan->ops.empty()) return fail(FV_ERR_INVALID_ARG, "plan_set_output_offset: no op to attach to");
    if (aux_slot != FV_SLOT_AUX_IN0 && aux_slot != FV_SLOT_AUX_IN1)
        return fail(FV_ERR_INVALID_ARG, "plan_set_output_offset: slot %d is not an auxiliary input", aux_slot);
    if (int rc = check_slot(y2_slot, true)) return rc;
    Op& o = plan->ops.back();
    if (o.type == OP_PAIR || o.type == OP_MRFSUM || o.type == OP_CONVH || o.sum3 || o.group != 0 ||
        (o.type == OP_CONVT && o.prec == FV_PAIR_SPLIT_F16))   // (a pair with a folded output conv included)
        return fail(FV_ERR_UNSUPPORTED, "plan_set_output_offset: only plain conv / transposed conv / two-source 1x1 / residual stack / pqmf ops carry an offset");
    if (y2_slot != FV_SLOT_NONE) {
        if (o.y2 != FV_SLOT_NONE) return fail(FV_ERR_INVALID_ARG, "plan_set_output_offset: the op already has a second output");
        if (y2_slot == o.y || y2_slot == o.x || y2_slot == FV_SLOT_IN) return fail(FV_ERR_INVALID_ARG, "plan_set_output_offset: y2 aliases");
        o.y2 = y2_slot;
        if (o.type != OP_PQMF) o.act_slope = 1.f;
    }
    o.sub = aux_slot;
    return 0;
}

int fv_plan_set_pair_output_conv(fv_plan_t* plan, const float* w, const float* bias, int y_slot, float act_slope, int post) {
    if (!plan || plan->ops.empty() || !w) return fail(FV_ERR_INVALID_ARG, "plan_set_pair_output_conv: no op / null weights");
    if (int rc = check_slot(y_slot, false)) return rc;
    Op& o = plan->ops.back();
    if ((o.type != OP_PAIR && o.type != OP_STAGE) || o.prec != FV_PAIR_SPLIT_F16 || o.Cin != 16 || o.group != 0 ||
        o.y2 != FV_SLOT_NONE || o.post != FV_POST_NONE || o.fold_w)
        return fail(FV_ERR_UNSUPPORTED, "plan_set_pair_output_conv: the last op must be an ungrouped 16-channel split-f16 "
                    "resblock pair (or a one-launch MRF stage) without an activated twin or a post op of its own");
    if (y_slot == FV_SLOT_IN || y_slot == o.x || y_slot == o.acc || y_slot == o.acc2 || y_slot == o.tmpb)
        return fail(FV_ERR_INVALID_ARG, "plan_set_pair_output_conv: the output slot aliases an operand");
    if (act_slope < 0.f || act_slope > 1.f) return fail(FV_ERR_INVALID_ARG, "plan_set_pair_output_conv: slope outside [0, 1]");
    o.fold_w = w;
    o.fold_b = bias;
    o.y = y_slot;
    o.act_slope = act_slope;
    o.post = post;
    return 0;
}

int fv_plan_set_group(fv_plan_t* plan, int group) {
    if (!plan || group < 0) return fail(FV_ERR_INVALID_ARG, "plan_set_group: group %d", group);
    plan->cur_group = group;
    return 0;
}

int fv_plan_set_sum_order(fv_plan_t* plan, int own_first) {
    if (!plan) return fail(FV_ERR_INVALID_ARG, "plan_set_sum_order: null plan");
    plan->cur_own_first = own_first ? 1 : 0;
    return 0;
}

int fv_plan_slot_shape(fv_plan_t* plan, int T, int slot, int* channels, int64_t* len) {
    if (!plan) return fail(FV_ERR_INVALID_ARG, "null plan");
    if (int rc = check_slot(slot, false)) return rc;
    Shape sh[FV_MAX_SLOTS];
    int64_t elems[FV_MAX_SLOTS];
    if (int rc = infer(plan, 1, T, sh, elems)) return rc;
    if (!sh[slot].set) return fail(FV_ERR_INVALID_ARG, "plan never writes slot %d", slot);
    if (channels) *channels = sh[slot].C;
    if (len) *len = sh[slot].T;
    return 0;
}

int fv_plan_output_shape(fv_plan_t* plan, int T, int* out_channels, int64_t* out_len) {
    return fv_plan_slot_shape(plan, T, FV_SLOT_OUT, out_channels, out_len);
}

int64_t fv_plan_workspace_bytes(fv_plan_t* plan, int B, int T) {
    if (!plan) return fail(FV_ERR_INVALID_ARG, "null plan");
    Shape sh[FV_MAX_SLOTS];
    int64_t elems[FV_MAX_SLOTS];
    if (int rc = infer(plan, B, T, sh, elems)) return rc < 0 ? rc : -rc;
    int64_t bytes = 0;
    for (int i = FV_SLOT_TMP0; i < FV_SLOT_AUX_IN0; ++i) bytes += (elems[i] * 4 + 255) / 256 * 256;
    return bytes;
}

int fv_plan_run(fv_plan_t* plan, int B, int T, const float* in, float* out, void* workspace,
                int64_t workspace_bytes, void* stream) {
    return fv_plan_run_aux(plan, B, T, in, out, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int fv_plan_run_aux(fv_plan_t* plan, int B, int T, const float* in, float* out, float* out2,
                    const float* const* aux_in, const int* aux_batched, void* workspace,
                    int64_t workspace_bytes, void* stream) {
    if (!plan || !in || !out) return fail(FV_ERR_INVALID_ARG, "plan_run: null argument");
    if (B <= 0 || T <= 0) return fail(FV_ERR_INVALID_ARG, "plan_run: B=%d T=%d", B, T);
    Shape sh[FV_MAX_SLOTS];
    int64_t elems[FV_MAX_SLOTS];
    if (int rc = infer(plan, B, T, sh, elems)) return rc;
    float* base[FV_MAX_SLOTS] = {};
    int64_t off = 0;
    for (int i = FV_SLOT_TMP0; i < FV_SLOT_AUX_IN0; ++i) {
        base[i] = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
        off += (elems[i] * 4 + 255) / 256 * 256;
    }
    base[FV_SLOT_AUX_IN0] = aux_in ? const_cast<float*>(aux_in[0]) : nullptr;
    base[FV_SLOT_AUX_IN1] = aux_in ? const_cast<float*>(aux_in[1]) : nullptr;
    base[FV_SLOT_OUT2] = out2;
    const int aux_b[2] = {aux_batched ? aux_batched[0] : 0, aux_batched ? aux_batched[1] : 0};
    for (const Op& o : plan->ops) {
        if (o.sub != FV_SLOT_NONE && !base[o.sub]) return fail(FV_ERR_INVALID_ARG, "plan_run: the plan subtracts auxiliary input %d, which was not given", o.sub - FV_SLOT_AUX_IN0);
        if ((o.y == FV_SLOT_OUT2 || o.y2 == FV_SLOT_OUT2) && !out2) return fail(FV_ERR_INVALID_ARG, "plan_run: the plan writes a second output, which was not given");
    }
    if (off > workspace_bytes || (off > 0 && !workspace))
        return fail(FV_ERR_WORKSPACE, "plan needs %lld workspace bytes, got %lld", (long long)off,
                    (long long)workspace_bytes);
    base[FV_SLOT_IN] = const_cast<float*>(in);
    base[FV_SLOT_OUT] = out;
    hipStream_t const s = (hipStream_t)stream;
    // shapes again, op by op (a slot may change shape when it is reused)
    for (int i = 0; i < FV_MAX_SLOTS; ++i) sh[i].set = false;
    sh[FV_SLOT_IN] = {plan->in_channels, T, true};
    for (size_t n = 0; n < plan->ops.size(); ++n) {
        const Op& o = plan->ops[n];
        // ---- split-f16 convs of the wide stages: the members of a group in one launch ----
        if (o.type == OP_CONVH) {
            size_t m = n + 1;
            if (o.group != 0)
                while (m < plan->ops.size() && m - n < 3 && plan->ops[m].type == OP_CONVH && plan->ops[m].group == o.group &&
                       plan->ops[m].Cin == o.Cin && plan->ops[m].dil == o.dil &&
                       plan->ops[m].pad_mode == o.pad_mode &&
                       plan->ops[m].pre_slope == o.pre_slope && plan->ops[m].act_slope == o.act_slope &&
                       plan->ops[m].out_div == o.out_div && plan->ops[m].post == o.post)
                    ++m;
            PairParams pp = {};
            pp.B = B;
            pp.T = (int)sh[o.x].T;
            pp.slope = o.pre_slope;
            pp.act_slope = o.act_slope;
            pp.out_div = o.out_div;
            pp.post = o.post;
            pp.prec = FV_PAIR_SPLIT_F16;
            pp.guard = plan->guard_dev;
            pp.reflect = o.pad_mode == FV_PAD_REFLECT;
            pp.n_members = (int)(m - n);
            for (size_t q = n; q < m; ++q) {
                const Op& qo = plan->ops[q];
                PairMember& mb = pp.m[q - n];
                mb.x = base[qo.x];
                mb.w1 = qo.pw1[0];
                mb.b1 = qo.pb1[0];
                mb.k = qo.pk[0];
                mb.y = base[qo.y];
                mb.y_act = qo.y2 == FV_SLOT_NONE ? nullptr : base[qo.y2];
                mb.res = qo.res == FV_SLOT_NONE ? nullptr : base[qo.res];
                mb.add1 = qo.acc == FV_SLOT_NONE ? nullptr : base[qo.acc];
                mb.add2 = qo.acc2 == FV_SLOT_NONE ? nullptr : base[qo.acc2];
            }
            if (int rc = launch_convh(pp, o.Cin, o.dil, s)) return rc;
            for (size_t q = n; q < m; ++q) {
                const Op& qo = plan->ops[q];
                sh[qo.y] = {qo.Cout, sh[qo.x].T, true};
                if (qo.y2 != FV_SLOT_NONE) sh[qo.y2] = sh[qo.y];
            }
            n = m - 1;
            continue;
        }
        // ---- a whole 16-channel MRF stage: one launch ----
        if (o.type == OP_STAGE) {
            MrfParams mp = {};
            mp.x = base[o.x];
            mp.blob = o.wp;
            for (int j = 0; j < 3; ++j) mp.k[j] = o.pk[j];
            mp.B = B;
            mp.T = (int)sh[o.x].T;
            mp.slope = o.pre_slope;
            mp.out_div = o.out_div;
            mp.act_slope = o.act_slope;
            mp.post = o.post;
            mp.guard = plan->guard_dev;
            mp.hist = static_cast<float*>(o.work);
            mp.hist_bytes = o.work_bytes;
            if (o.fold_w) {
                mp.fold_w = o.fold_w;
                mp.fold_b = o.fold_b;
                mp.fold_y = base[o.y];
            } else {
                mp.y = base[o.y];
                mp.y_act = o.y2 == FV_SLOT_NONE ? nullptr : base[o.y2];
            }
            if (int rc = launch_mrfh(mp, o.Cin, o.sdil, s)) return rc;
            sh[o.y] = {o.fold_w ? 1 : o.Cout, sh[o.x].T, true};
            if (o.y2 != FV_SLOT_NONE) sh[o.y2] = sh[o.y];
            continue;
        }
        // ---- fused ResBlock pairs: the members of a group (the three ResBlocks of an MRF stage) in one launch ----
        if (o.type == OP_PAIR || o.type == OP_MRFSUM) {
            // the launch that starts at op n0: its members [n0, end) and its parameters (Tn: samples per utterance)
            auto gather = [&](size_t n0, int Tn, PairParams& pp) -> size_t {
            const Op& o = plan->ops[n0];
            const size_t n = n0;
            size_t m = n + 1;
            if (o.type == OP_PAIR && o.group != 0)
                while (m < plan->ops.size() && m - n < 3 && plan->ops[m].type == OP_PAIR && plan->ops[m].group == o.group &&
                       plan->ops[m].Cin == o.Cin && plan->ops[m].dil == o.dil &&
                       plan->ops[m].pre_slope == o.pre_slope && plan->ops[m].act_slope == o.act_slope && !o.fold_w &&
                       !plan->ops[m].fold_w &&
                       plan->ops[m].prec == o.prec && plan->ops[m].out_div == o.out_div && plan->ops[m].post == o.post)
                    ++m;
            pp = {};
            pp.B = B;
            pp.T = Tn;
            pp.slope = o.pre_slope;
            pp.act_slope = o.act_slope;
            pp.out_div = o.out_div;
            pp.post = o.post;
            pp.prec = o.prec;
            pp.guard = plan->guard_dev;
            if (o.type == OP_MRFSUM) {
                const int xs3[3] = {o.x, o.xb, o.xc};
                pp.sum = 1;
                pp.n_members = 3;
                for (int j = 0; j < 3; ++j) {
                    PairMember& mb = pp.m[j];
                    mb.x = base[xs3[j]];
                    mb.w1 = o.pw1[j];
                    mb.w2 = o.pw2[j];
                    mb.b1 = o.pb1[j];
                    mb.b2 = o.pb2[j];
                    mb.k = o.pk[j];
                    mb.y = base[o.y];
                    mb.y_act = o.y2 == FV_SLOT_NONE ? nullptr : base[o.y2];
                }
            } else {
                pp.n_members = (int)(m - n);
                for (size_t q = n; q < m; ++q) {
                    const Op& qo = plan->ops[q];
                    PairMember& mb = pp.m[q - n];
                    mb.x = base[qo.x];
                    mb.w1 = qo.pw1[0];
                    mb.w2 = qo.pw2[0];
                    mb.b1 = qo.pb1[0];
                    mb.b2 = qo.pb2[0];
                    mb.k = qo.pk[0];
                    mb.y = base[qo.y];
                    mb.y_act = qo.y2 == FV_SLOT_NONE ? nullptr : base[qo.y2];
                    mb.add1 = qo.acc == FV_SLOT_NONE ? nullptr : base[qo.acc];
                    mb.add2 = qo.acc2 == FV_SLOT_NONE ? nullptr : base[qo.acc2];
                    if (qo.fold_w) {            // (never grouped: one member)
                        pp.fold_w = qo.fold_w;
                        pp.fold_b = qo.fold_b;
                        pp.fold_y = base[qo.y];
                        mb.y = nullptr;
                    }
                }
            }
            return m;
            };
            auto set_shapes = [&](size_t n0, size_t m0) {
                for (size_t q = n0; q < m0; ++q) {
                    const Op& qo = plan->ops[q];
                    sh[qo.y] = {qo.fold_w ? 1 : qo.Cout, sh[qo.x].T, true};
                    if (qo.y2 != FV_SLOT_NONE) sh[qo.y2] = sh[qo.y];
                }
            };
            PairParams pp;
            const size_t m = gather(n, (int)sh[o.x].T, pp);
            if (o.Cin == 64 && o.prec == FV_PAIR_SPLIT_F16) {
                if (int rc = launch_convp(pp, o.dil, s)) return rc;
            } else if (o.Cin == 128 && o.prec == FV_PAIR_SPLIT_F16 && !tuning().pair128_unfused) {
                if (int rc = launch_convq(pp, o.dil, s)) return rc;
            } else if (o.Cin >= 64) {
                float* mids[3] = {nullptr, nullptr, nullptr};
                for (size_t q = n; q < m; ++q) mids[q - n] = base[plan->ops[q].tmpb];
                if (int rc = launch_wide_pairs(pp, mids, o.Cin, o.dil, s)) return rc;
            } else if (int rc = launch_pairs(pp, o.Cin, o.dil, s)) return rc;
            set_shapes(n, m);
            n = m - 1;
            continue;
        }
        // ---- a group of mutually independent convs: one launch when possible ----
        if (o.group != 0 && o.type == OP_CONV) {
            size_t m = n;
            ConvParams gp[3];
            int cnt = 0;
            while (m < plan->ops.size() && plan->ops[m].group == o.group && plan->ops[m].type == OP_CONV &&
                   cnt < 3) {
                const Op& q = plan->ops[m];
                gp[cnt++] = make_params(q, base[q.x], base[q.y], q.y2 == FV_SLOT_NONE ? nullptr : base[q.y2],
                                        q.res == FV_SLOT_NONE ? nullptr : base[q.res],
                                        q.acc == FV_SLOT_NONE ? nullptr : base[q.acc],
                                        q.acc2 == FV_SLOT_NONE ? nullptr : base[q.acc2], B, sh[q.x].T,
                                        q.x2 == FV_SLOT_NONE ? nullptr : base[q.x2]);
                ++m;
            }
            if (int rc = launch_conv_group(gp, cnt, s)) return rc;
            for (size_t q = n; q < m; ++q) {
                const Op& qo = plan->ops[q];
                sh[qo.y] = {qo.Cout, conv_out_len(qo, sh[qo.x].T), true};
                if (qo.y2 != FV_SLOT_NONE) sh[qo.y2] = sh[qo.y];
            }
            n = m - 1;
            continue;
        }
        if (o.sum3) {
            hipStream_t s3 = s;
            Op mb = o, mc = o;           // members 1, 2: same layer geometry, their own taps / weights
            mb.k = o.kb; mb.pad = (o.kb - 1) / 2; mb.wp = o.wpb; mb.bias = nullptr;
            mc.k = o.kc; mc.pad = (o.kc - 1) / 2; mc.wp = o.wpc; mc.bias = nullptr;
            float* y2s = o.y2 == FV_SLOT_NONE ? nullptr : base[o.y2];
            // One launch pays when the three K loops in a row still leave enough blocks to fill the
            // GPU (tiles of 32 x 128, or 16 x 128); otherwise the two-launch form: members 1, 2 as a
            // grouped launch into scratch, then member 0 with both as running-sum inputs.
            const int64_t T3 = sh[o.x].T;
            const int Mp = pad_rows(o.Cout);
            const int64_t blocks = (int64_t)(Mp == 16 ? 1 : Mp / 32) * ((T3 + 127) / 128);
            const int min_blocks = tuning().sum3_min;   // 800 -- measured: HiFi-GAN light, B = 1
            int rc3;
            if (blocks >= min_blocks && Mp == o.Cout) {   // (whole row tiles only: the kernel's epilogue is the affine one)
                ConvParams ps[3] = {
                    make_params(o, base[o.x], base[o.y], y2s, base[o.res], nullptr, nullptr, B, T3),
                    make_params(mb, base[o.xb], base[o.y], y2s, base[o.resb], nullptr, nullptr, B, T3),
                    make_params(mc, base[o.xc], base[o.y], y2s, base[o.resc], nullptr, nullptr, B, T3)};
                rc3 = launch_conv_sum3(ps, s3);
            } else {
                Op duo_b = mb, duo_c = mc;                 // r_b, r_c: conv + residual, raw, no mean / activation
                duo_b.out_div = duo_c.out_div = 1.f;
                duo_b.act_slope = duo_c.act_slope = 1.f;
                duo_b.post = duo_c.post = FV_POST_NONE;
                ConvParams duo[2] = {
                    make_params(duo_b, base[o.xb], base[o.tmpb], nullptr, base[o.resb], nullptr, nullptr, B, T3),
                    make_params(duo_c, base[o.xc], base[o.tmpc], nullptr, base[o.resc], nullptr, nullptr, B, T3)};
                rc3 = launch_conv_group(duo, 2, s3);
                if (!rc3) {
                    Op car = o;                            // ((own + r_b) + r_c) / out_div, summed bias on this one
                    car.own_first = 1;
                    rc3 = launch_conv(make_params(car, base[o.x], base[o.y], y2s, base[o.res], base[o.tmpb],
                                                  base[o.tmpc], B, T3), s3);
                }
            }
            if (rc3) return rc3;
            sh[o.y] = {o.Cout, conv_out_len(o, sh[o.x].T), true};
            if (o.y2 != FV_SLOT_NONE) sh[o.y2] = sh[o.y];
            continue;
        }
        if (o.type == OP_STACK && o.alt_w1 && stack_two_launch(o.Cout, B, sh[o.x].T)) {
            // many tiles: dilated conv into the scratch slot (convs_kernel), then the K-concatenated 1x1 pair (convr_kernel)
            PairParams pp = {};
            pp.B = B;
            pp.T = (int)sh[o.x].T;
            pp.slope = o.pre_slope;
            pp.act_slope = 1.f;
            pp.out_div = 1.f;
            pp.prec = FV_PAIR_SPLIT_F16;
            pp.guard = plan->guard_dev;
            pp.reflect = o.pad_mode == FV_PAD_REFLECT;
            pp.n_members = 1;
            pp.m[0].x = base[o.x];
            pp.m[0].w1 = o.alt_w1;
            pp.m[0].b1 = o.bias;
            pp.m[0].k = o.k;
            pp.m[0].y = base[o.alt_mid];
            if (int rc = launch_convh(pp, o.Cin, o.dil, s)) return rc;
            PairParams pg = {};
            pg.B = B;
            pg.T = (int)sh[o.x].T;
            pg.slope = o.pre_slope;
            pg.act_slope = o.act_slope;
            pg.post = o.post;
            pg.prec = FV_PAIR_SPLIT_F16;
            pg.guard = plan->guard_dev;
            pg.m[0].x = base[o.alt_mid];
            pg.m[0].x2 = base[o.x];
            pg.m[0].w1 = o.alt_w2;
            pg.m[0].b1 = o.bias2;
            pg.m[0].y = base[o.y];
            pg.m[0].y_act = o.y2 == FV_SLOT_NONE ? nullptr : base[o.y2];
            pg.sub = o.sub == FV_SLOT_NONE ? nullptr : base[o.sub];
            pg.sub_batched = o.sub == FV_SLOT_NONE ? 0 : aux_b[o.sub - FV_SLOT_AUX_IN0];
            if (int rc = launch_convg(pg, o.Cout, s)) return rc;
            sh[o.y] = {o.Cout, sh[o.x].T, true};
            if (o.y2 != FV_SLOT_NONE) sh[o.y2] = sh[o.y];
            continue;
        }
        const int64_t Tin = sh[o.x].T;
        const int64_t Tout = conv_out_len(o, Tin);
        const float* res = o.res == FV_SLOT_NONE ? nullptr : base[o.res];
        const float* acc = o.acc == FV_SLOT_NONE ? nullptr : base[o.acc];
        const float* acc2 = o.acc2 == FV_SLOT_NONE ? nullptr : base[o.acc2];
        if (o.type == OP_CONVT && o.in_merge) {        // merged INPUT (fv_plan_set_input_merge): xb, xc travel as acc, acc2
            acc = base[o.xb];
            acc2 = o.xc == FV_SLOT_NONE ? nullptr : base[o.xc];
        }
        float* y2 = o.y2 == FV_SLOT_NONE ? nullptr : base[o.y2];
        if (int rc = run_op(o, base[o.x], base[o.y], y2, res, acc, acc2, B, Tin, s,
                            o.x2 == FV_SLOT_NONE ? nullptr : base[o.x2], o.sub == FV_SLOT_NONE ? nullptr : base[o.sub],
                            o.sub == FV_SLOT_NONE ? 0 : aux_b[o.sub - FV_SLOT_AUX_IN0], plan->guard_dev))
            return rc;
        sh[o.y] = {(o.type == OP_PQMF || o.pq_h) ? 1 : o.Cout, Tout, true};
        if (o.y2 != FV_SLOT_NONE) sh[o.y2] = sh[o.y];
    }
    return 0;
}

int fv_plan_num_ops(fv_plan_t* plan) { return plan ? (int)plan->ops.size() : 0; }

int fv_plan_set_guard(fv_plan_t* plan, int* word) {
    if (!plan) return fail(FV_ERR_INVALID_ARG, "plan_set_guard: null plan");
    plan->guard_host = plan->guard_dev = nullptr;
    if (!word) return 0;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, word, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return fail(FV_ERR_INVALID_ARG, "plan_set_guard: the guard word must live in pinned, device-mapped host memory "
                                        "(hipHostMalloc / torch pin_memory)");
    }
    plan->guard_host = word;
    plan->guard_dev = static_cast<int*>(dev);
    return 0;
}

int fv_plan_check_range(fv_plan_t* plan, void* stream) {
    if (!plan) return fail(FV_ERR_INVALID_ARG, "plan_check_range: null plan");
    if (!plan->guard_host) return 0;
    FV_HIP(hipStreamSynchronize((hipStream_t)stream));
    volatile int* w = plan->guard_host;
    const int seen = *w;
    if (seen == 0) return 0;
    *w = 0;
    if (seen == 4)
        return fail(FV_ERR_RANGE_LOW, "a split-f16 kernel met operands that were smaller than 2^-10 throughout a block's share of a "
                                      "tensor (not all zero): the last run may carry fewer than 22 bits; repeat it on an fp32-precision plan");
    return fail(FV_ERR_RANGE, "a split-f16 kernel met an operand outside its domain (|v| >= 65520 or a non-finite value; "
                              "or operands that were smaller than 2^-10 throughout a block's share of a tensor): the "
                              "results of the last run are not valid; repeat it on an fp32-precision plan");
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
