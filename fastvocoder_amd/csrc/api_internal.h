// What api.hip (operators, measurement hook, tuning) and plan.hip (the plan executor) share: the op record, the plan, and the
// argument checks / launch helpers both the direct entry points and their plan ops go through.
#pragma once
#include <stdint.h>

#include <vector>

#include "fv_internal.h"

namespace fv {

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

int64_t conv_out_len(const Op& o, int64_t Tin);
ConvParams make_params(const Op& o, const float* x, float* y, float* y2, const float* res,
                              const float* acc, const float* acc2, int B, int64_t Tin,
                              const float* x2 = nullptr, const float* sub = nullptr, int sub_batched = 0);
int run_op(const Op& o, const float* x, float* y, float* y2, const float* res, const float* acc,
                  const float* acc2, int B, int64_t Tin, hipStream_t s, const float* x2 = nullptr,
                  const float* sub = nullptr, int sub_batched = 0, int* guard = nullptr);
int check_pad_mode(int pad_mode, int pad, int k, int dil);
int check_stack_args(int C, int k, int dil, int pad_mode, float slope, float act_slope, int post = FV_POST_NONE);
int launch_wide_pairs(const PairParams& pp, float* const* mid, int C, int dil, hipStream_t s);
int check_pair_args(int n, int C, const int* k, int dil, int prec = FV_PAIR_F32);
int check_stage_args(int C, const int* k, const int* dil, float slope, float act_slope, int post);
int check_convh_args(int n, int C, const int* k, int dil, int pad_mode);

}  // namespace fv
