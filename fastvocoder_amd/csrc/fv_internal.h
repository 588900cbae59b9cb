// Internal declarations shared by the translation units of libfastvocoder_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>
#include <stdint.h>

#include "../../include/fastvocoder_hip.h"

namespace fv {

int fail(int code, const char* fmt, ...);

#define FV_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return ::fv::fail((int)e_, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// argument checks shared by the operator entries (api.hip) and the pack entries (pack.hip)
int check_conv_args(int Cin, int Cout, int k, int dil);
int check_convt_split_args(int Cin, int Cout, int k, int stride, int pad, int out_pad);

// Rows of the packed weight image are padded so that every M tile is full.
static inline int pad_rows(int M) { return M <= 16 ? 16 : round_up(M, 32); }

// Polyphase view of ConvTranspose1d(k, stride s, padding p): output phase r of
// out[q*s + r] reads input taps x[q + delta], delta in [dmin, dmax].
struct Polyphase {
    int dmin, dmax, taps;  // taps = dmax - dmin + 1
};
static inline Polyphase polyphase(int k, int s, int p) {
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int r = 0; r < s; ++r) {
        int a = (r + p) % s, c = (r + p) / s;
        for (int j = a, m = 0; j < k; j += s, ++m) {
            int d = c - m;
            if (d < dmin) dmin = d;
            if (d > dmax) dmax = d;
        }
    }
    if (dmin > dmax) { dmin = 0; dmax = 0; }
    Polyphase ph = {dmin, dmax, dmax - dmin + 1};
    return ph;
}

// PHASE-MAJOR form of a transposed conv whose kernel is a whole number of strides (k = tp*s,
// every shipped upsampler: k = 2s): every output phase r then has exactly tp taps, the window
// x[q + d0(r) .. q + d0(r) + tp - 1] with d0(r) = (r+p)/s - tp + 1.  With the GEMM rows ordered
// phase-major (m = r*Cout + co) and Cout a multiple of the row tile, a row tile holds ONE
// phase, so it runs a dense tp-tap conv with its own window shift -- none of the zero taps
// of the co-major form (1/3 of its MACs at k = 2s).
static inline bool convt_phase_major(int Cout, int k, int s, int p) {
    return s > 1 && k % s == 0 && k / s >= 2 && p <= s && Cout % 32 == 0;
}

// Nearest-repeat x u followed by Conv1d(k, zero padding p) (the reference's UpsampleLayer,
// model/generator/modules.py:160-177) in the same phase form: output t = q*u + r reads
// x[q + delta], delta = floor((r + j - p) / u); taps j that land on the same input
// sample are summed into one weight at pack time.
static inline int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
static inline Polyphase upsample_phases(int k, int u, int p) {
    Polyphase ph = {floor_div(-p, u), floor_div(u - 1 + k - 1 - p, u), 0};
    ph.taps = ph.dmax - ph.dmin + 1;
    return ph;
}

// Everything one implicit-GEMM conv launch needs.  The GEMM is
//   Y[m, q] = sum_{ci, j} Wp[ci, j, m] * act(X[ci, q + j*dil - pad])
// with M rows (= Cout, or Cout*stride phases for a transposed conv) and Tq
// columns per batch item; the epilogue scatters row m / column q to
// y[co, q*ups + ph] with co = m / ups, ph = m % ups.
struct ConvParams {
    const float* x;
    const float* x2;      // optional second input tensor [B, Cin - Cin1, Tin]: GEMM rows ci >= Cin1
                          // come from it (K-concatenation of two convs that sum, e.g. ResidualStack's
                          // 1x1 + skip 1x1); null otherwise.  1-tap kernels only.
    int Cin1;             // channels read from x (== Cin without x2)
    const float* wp;      // [Cin*k][Mpad]
    const float* bias;    // [Cout] or null
    const float* res;     // [B,Cout,Tout] or null
    const float* acc_in;  // [B,Cout,Tout] or null
    const float* acc_in2; // second running-sum input (added to acc_in first) or null
    float* y;             // [B,Cout,Tout]
    float* y_act;         // optional activated twin of y: act(y, act_slope), or null
    const float* sub;     // optional offset subtracted AFTER post / activation from the stored value -- from the
                          // twin when there is one (y stays raw), else from y: [Cout,Tout], or [B,Cout,Tout]
    int sub_batched;      //   (sub_batched).  The bias-removal flows: y - generator(0) (bin/synthesize.py:74-80)
    int B, Cin, M, Mpad, Cout;
    int Tin, Tq, Tout;
    int k, dil, pad, pad_mode;
    int ups;              // 1 for Conv1d, stride for ConvTranspose1d
    float pre_slope, out_div;
    float act_slope;      // with y_act: slope of the twin; without: y itself is stored as act(y, act_slope)
    int post;
    int own_first;        // 0: y = (acc_in + acc_in2) + own;  1: y = (own + acc_in) + acc_in2  (own = conv + bias + res)
    int phase_major;      // transposed conv in phase-major form (see convt_phase_major): rows m = r*Cout + co,
    int pad_orig;         //   the window shift of a row tile comes from its phase and the layer's padding
    // filled in by the launcher
    int ci_chunk;   // input channels staged per LDS stage
    int nchunks;    // ceil(Cin / ci_chunk)
    int xw;         // LDS row stride of the input tile (floats)
    int ncol4;      // float4 columns of an input row the tile actually reads
    int ncol4c;     // float4 columns per row of the LDS image (>= ncol4; xw = 4*ncol4c)
    unsigned ncol4c_magic;  // ceil(2^32 / ncol4c): idx / ncol4c == umulhi(idx, magic)
    int xbuf, wbuf; // floats per LDS buffer of the x / w image (multiples of 256)
    int n_tiles;    // time tiles per batch item
    int m_tiles, tiles_per_run;   // row tiles; consecutive time tiles per block
    int nx_inst, nw_inst;         // 64-lane DMA instructions per stage for the x / w image
    int red_off;    // float offset of the split-K reduction area in LDS
    int vec_ok;     // rows are 16-byte aligned: float4 global loads allowed
    int dbg;        // ablation switches (Tuning::conv_dbg, timing experiments only): 1 no epilogue, 2 no restaging, 4 no MFMA
    double alg_flops;   // > 0: algorithmic FLOPs of the layer for the measurement hook (a transposed conv in
                        // polyphase form executes zero taps that the reference's MAC count does not contain)
};

// Three (or two: the third grid empty) mutually independent convs of one MRF position in
// one launch: conv_group3_kernel, blockIdx.z picks the problem, largest kernel first.
struct GroupParams {
    ConvParams p[3];   // sorted k = 11, 7, 3
    int grid_x[3];
};

// The last convs of the three ResBlocks of an MRF stage, summed in one accumulator
// (conv_sum3_kernel): p sorted k = 11, 7, 3; p[0] carries the output side and the summed bias.
struct Sum3Params {
    ConvParams p[3];
    int xbuf_max, wbuf_max;   // floats per LDS stage buffer (largest member)
};
int launch_conv_sum3(ConvParams* ps, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size (a driver call per launch otherwise:
// tens of microseconds of host time on a path whose whole forward is under a millisecond); api.hip
int allow_dynamic_lds(const void* kernel, size_t bytes);
int device_cu_count();
// Tuning switches of the launchers.  These are NOT product configuration: the defaults below are what every product run
// uses, and the launch path never reads the environment.  A process started with FV_TUNING=1 reads FV_<NAME> once, at
// the first launch (tools/ sweeps); tests change entries through fv_tuning_set (api.hip) to prove that block counts and
// schedules do not change results.  pair_dbg / conv_dbg are ablation switches for timing experiments: results are WRONG.
struct Tuning {
    int pair_dbg = 0;        // persistent kernels: 1 no x prefetch after a block's first tile, 2 no conversion, 4 no MFMA, 8 no stores, 16 no residual loads
    int conv_dbg = 0;        // fp32 conv kernels (FV_DBG): 1 no epilogue, 2 no restaging, 4 no MFMA
    int sched = 1;           // 0: contiguous cost cut only; 1: host schedule for two-member launches; 2: ... and three
    int sched_switch = 4;    // cost units a block pays for taking up another member (pair_schedule)
    int convh_skel = -1;     // per-tile constant of the partition cost (-1: the launcher's own: 5 at 64 channels, 2 above)
    int convp_skel = 5;
    int convq_skel = -1;     // (-1: 5)
    int pair128_unfused = 0; // 1: 128-channel ResBlock pairs as two conv launches (convh) instead of the fused convq kernel (A/B, bit-identity tests)
    int convh_rows64 = -1;    // the split-f16 convs at 128+ channels on 64-row tiles (1: convh_kernel) or 128-row ones (0: convs_kernel); -1: by size
    int convt_rows64 = -1;    // the split-f16 transposed conv (128+ input channels) on 64-row tiles (1) or 128-row ones (0); -1: by size
    int convs_ringfree = -1;  // 256 / 512-channel convs with many items: convs2_kernel (A operands L2 -> registers, no ring) on 64-column (1) or
                              // 128-column tiles (2; -1: by the number of items) instead of convs_kernel (0)
    int convu_resident = 1;   // transposed convs of 128 / 256 input channels with >= 2 column tiles per CU: convu2_kernel (all rows of a column
                              // tile on resident images, A operands L2 -> registers) instead of convu_kernel (0); 2: whenever the shape allows
    int convt_lean = 50;      // ... on the lean kernel (convtl_kernels.hpp) up to this many (64 x 64 item, chunk) units per CU, in tenths; 0: never
    int convq_wide = 20;         // fused 128-channel pairs, dilation 1 / 3: 128-column tiles per CU (in tenths) from which the wide form runs
    int convp_wide = 20;         // fused 64-channel pairs: 256-column tiles per CU (in tenths) from which the wide no-ring form runs
    int convp_pp = 0;            // fused 64-channel pairs as two wave groups one conv phase apart (convq3_kernels.hpp): 0 never, 1 always (convq3_kernel: one 8-wave block), 2 always as four-wave blocks, two per CU (convq4_kernel).
                                 // [measured, round 6: identical bits, 5-15 % SLOWER than convq2_kernel at batch 1 and 8 -- the VALU
                                 // instructions of one wave do not run beside the MFMAs of another wave of the same SIMD
                                 // (tools/pingpong_probe.hip, profiles/r06_pingpong.txt); kept as the measured form of that experiment]
    int stack_wide = 10;         // residual stacks of 256 channels: tiles of 64 columns per CU (in tenths) from which the wide tile runs
                                 // (Basis-MelGAN, 1000 frames: batch 1 -- 250 such tiles in its second stage -- 0.238 narrow / 0.248 wide,
                                 // batch 2 0.384 / 0.359, batch 3 0.624 / 0.534)
    int stack_items = 1 << 20;   // residual stacks of 256 channels: tiles per CU (in tenths) up to which the one-launch kernel runs
                                 // (api.hip stack_two_launch; measured: it wins at every size -- 0 forces the two launches, A/B)
    int convg_rows64 = -1;    // the two-source 1x1 conv on 64-row tiles (1: convg_kernel) or 128-row ones (0: convr_kernel); -1: by size
    int pairh_skel = -1;     // (-1: 8 at 16 channels, 6 at 32)
    int pair_skel = -1;      // (-1: 4 at 16 channels, 3 at 32)
    int convh_blocks = 0;    // > 0: persistent blocks of the convh / convp / convt launches (default: one per CU)
    int pair_blocks = 0;     // > 0: ... of the pair launches
    int mrf_blocks = 0;      // > 0: ... of the one-launch MRF stage (mrfh_kernel)
    int mrf_shape = 0;       // one-launch MRF stage: 0 -- 12 waves x 3 fragments (576-column windows), 1 -- 16 waves x 2 (512)
    int mrf_prio = 1;        // ... s_setprio inside the K loops (0: none; 1: level 1; 2 / 3: later waves higher -- mrfh_kernels.hpp)
    int sum3_min = 800;      // fewest tiles for which the three last convs of an MRF stage run as ONE fp32 launch
    int lds_budget = 39;     // fp32 conv kernels: KiB of LDS per block
    int units = 500;
    int shape16 = -1, shape32 = -1, shape64 = -1;   // >= 0: force a tile shape of the fp32 conv kernel
    int krows = 176;
    int grid_cap = 1024;
    int no_group = 0;
    unsigned long long trace_ptr = 0;   // FV_PAIR_TRACE_PTR (with -DFV_PAIR_TRACE builds only): device buffer for cycle stamps
};
const Tuning& tuning();

// ---- fused ResBlock1 pairs (pair_kernels.hpp / pair_launch.hip) ---------------------------------------
// one ResBlock's pair (a "member" of the launch)
struct PairMember {
    const float* x;      // [B, C, T] raw input = residual
    const float* x2;     // two-source 1x1 conv (convg_kernel): the second input [B, C, T], read as it is
    const float* w1;     // fv_pack_pair_weight image of conv1 / conv2
    const float* w2;
    const float* b1;     // [C] or null
    const float* b2;
    float* y;            // [B, C, T] raw output (sum mode: member 0's is THE output)
    float* y_act;        // optional activated twin lrelu(y, act_slope), or null
    const float* res;    // conv launches of the split-f16 wide-channel kernels (convh_kernels.hpp): residual, or null
    int n_items;         // ... (utterance, column tile, row tile) items of the member
    const float* add1;   // split-f16 kernels, last launch of an MRF stage: y = post(((x' + add1) + add2) / out_div)
    const float* add2;   //   (the other two ResBlocks' outputs, hifigan.py:99-103), or null
    int k;               // taps: 11, 7 or 3
    int cost;            // relative cost of one tile of this member (partition weights): taps + per-tile overhead
    int n_tiles;         // tiles per utterance
    int w_off;           // float offset of this member's two weight images in dynamic LDS
};

constexpr int kSchedBlocks = 256;     // blocks a launch's schedule can describe (one per CU)

struct PairCore {
    PairMember m[3];
    int n_members;
    int sum;             // members accumulate into m[0].y ( / out_div, activation)
    int B, T;
    int n_out_sum;       // sum mode: output columns per tile (the largest member's NOUT)
    float slope;         // leaky slope of the two in-block activations
    float out_div;       // sum mode: divisor
    float act_slope;     // y_act = lrelu(y, act_slope); without y_act and != 1: y itself is stored activated
    int post;            // FV_POST_* applied to y (sum mode / single)
    int nblk;            // persistent blocks in the grid
    int ctot, nch, nmt;  // convh: channels in and out, chunks of <= 128 input channels, row tiles of 64
    const float* fold_w; // pairh, C = 16, one member (pairh_run_member<G, true>): conv_post folded into the pair -- its
    const float* fold_b; //   weights [C][7], bias [1] or null, output [B, 1, T]; `post` is applied to that output,
    float* fold_y;       //   act_slope to the pair's own (never stored) output in front of it
    const float* sub;    // convg_kernel: optional output offset (fv_plan_set_output_offset): subtracted after post / activation from
    int sub_batched;     //   the twin when there is one (y stays raw), else from y; [C, T] or (sub_batched) [B, C, T]
    int* guard;          // split-f16 kernels: device-visible word set to 1 when a final value is not finite (an operand
                         // left the f16 range): pairh_kernels.hpp range_note; null: no check
    int sched_on;        // convh / convp: sched[] holds this launch's block schedule (pair_schedule); 0: the kernel cuts
                         // the cost-weighted item sequence into nblk contiguous shares itself (pair_share)
    int reflect;         // convh: rows outside [0, T) are the mirrored samples (ReflectionPad1d) instead of zeros
    int ups, pad_t, Tout, cout;   // transposed conv (convt_kernel): stride, padding, output samples, output channels;
                                  // T = input samples, ctot = input channels (64: half a chunk), rows m = co * ups +
                                  // phase in nmt tiles of 64
    int prec;            // FV_PAIR_F32: fp32 MFMA (pair_kernels.hpp); FV_PAIR_SPLIT_F16: pairh_kernels.hpp
    int x_off, mid_off;  // float offsets of the x image / intermediate in dynamic LDS
    int img_off;         // split-f16 kernels: float offset of the x image (x_off: the member's packed weights)
    int bias_off;        // ... of the staged biases: per member [b1[C] | b2[C]]
    unsigned long long* trace;   // tuning aid (FV_PAIR_TRACE_PTR): s_memtime stamps [block < 8][wave][tile < 8][16 events]
    int dbg;             // ablation switches (Tuning::pair_dbg, timing experiments only -- results are wrong):
                         // 1 no x DMA after a block's first tile, 2 no activation pass, 4 no MFMA,
                         // 8 no stores, 16 no residual loads
};

struct PairParams : PairCore {
    // Block schedule (few, unequal items per block): two words per block, member m's items [lo, lo + count) packed as
    // lo (11 bits) | count (5 bits): word 0 = member 0 | member 1 << 16, word 1 = member 2.  Part of the kernel
    // arguments: no device table, no copy, legal under stream capture.
    unsigned sched[2 * kSchedBlocks];
};

// tile geometry of the pair kernels, the run-time mirror of PairGeom<> (pair_kernels.hpp)
struct PairShape {
    int MH, NF, NG;      // row halves, column fragments per wave, column groups
    int NW, NM;          // waves per block, intermediate columns per tile
    int XS, NXI, MS;     // x image: row stride (floats), DMA instructions; intermediate row stride
    int WF;              // floats per packed conv weight
    int NOUT;            // output columns per tile
};
PairShape pair_shape(int C, int k, int dil);
// ... of the split-f16 pair kernels, the run-time mirror of PairHGeom<> (pairh_kernels.hpp)
// Block shapes of the split-f16 pair kernels (tools/pair_bench.py, HiFi-GAN light stage sizes, B = 1):
// C = 32, one block per CU -- column groups (= waves): 8 (128-column tiles, 2 waves per SIMD) 59 us per three-member
// launch, 12 (192 columns, 3 per SIMD) 53 us, 15 (240 columns, 4 per SIMD; 16 would need 162 KB of LDS) 49 us;
// C = 16, two blocks per CU, 256-column tiles -- 4 waves x 4 fragments with the A operands in registers (2 waves per
// SIMD), or 8 waves x 2 fragments with streamed A operands (116 VGPRs: 4 per SIMD)
#ifndef FV_PAIRH32_NG
#define FV_PAIRH32_NG 15
#endif
#ifndef FV_PAIRH16_NF
#define FV_PAIRH16_NF 2
#define FV_PAIRH16_NG 8
#endif
struct PairHShape {
    int MH, NF, NG, NM;
    int KS;              // K steps per conv (32 K values each)
    int XROWS, WB;       // x image rows; bytes of one conv's packed weights
    int MROWS;           // rows of the intermediate image
    int XIMG, MIMG;      // bytes of the x image / the intermediate image
    int NOUT;
};
PairHShape pairh_shape(int C, int k, int dil);
// Conv1d with split-f16 operands, C = 64 / 128 (convh_kernels.hpp): members use x, w1 (fv_pack_convh_weight image),
// b1 (bias), res, add1 / add2, y, y_act, k; 'same' zero padding
struct ConvHShape {
    int CG, NFW, NTC;    // 32-channel groups, fragments per wave, output columns per tile
    int NSTEP, NST;      // K steps of 32, stages of two steps
    int XROWS, XIMG, NMT;   // image rows / bytes, 64-row tiles
    int NCH;                // chunks of input channels (128 each above 128 channels)
};
ConvHShape convh_shape(int C, int k, int dil);
int launch_convh(PairParams p, int C, int dil, hipStream_t stream);
// Few, unequal items per block (batch 1: 378 items of three costs on 256 blocks): a longest-processing-time-first
// assignment instead of the contiguous cut, written into p.sched (host cache per shape); p.sched_on = 0 when every block
// has many items anyway
void pair_schedule(PairParams& p, int nblk, bool three_members = false);
// The kernels' own contiguous, cost-balanced cut (pair_share) computed on the host: p.sched[i] = the global item number
// (members concatenated) share i starts at, i = 0 .. nblk - 1 (nblk <= 2 kSchedBlocks entries), p.sched_on = 2; n[m]: items
// of member m.  Left alone (sched_on unchanged) when the table does not fit.
void pair_cut_schedule(PairParams& p, int nblk, const long long* n);
// fused ResBlock pair at C = 64 with split-f16 operands (convq2_kernels.hpp): members use x, w1, w2
// (fv_pack_pair_weight_ex images), b1, b2, add1 / add2, y, y_act, k
int launch_convp(PairParams p, int dil, hipStream_t stream);
// fused ResBlock pair at C = 128 (convq2_kernels.hpp): 128-row x 64-column tiles; members as launch_convp (w1 / w2: the
// fv_pack_pair_weight_ex images of the conv kernel, [row tile][step][8 KB])
int launch_convq(PairParams p, int dil, hipStream_t stream);
template <int CG, int NFW>
int launch_convh_geom(const PairParams& p, int dil, size_t lds, hipStream_t s);
// ConvTranspose1d with split-f16 operands, k = 2 stride, Cin = 64 or a multiple of 128 (convt_kernel in
// convh_kernels.hpp): member 0 uses x, w1 (fv_pack_conv_transpose1d_split_f16 image), b1, y, y_act
int launch_convt(PairParams p, int Cin, int Cout, int stride, int pad, int Tout, hipStream_t stream);
int launch_convt_geom(const PairParams& p, int cg, size_t lds, hipStream_t s);
int launch_convs2_geom(const PairParams& p, int dil, int nh, size_t lds, hipStream_t s);
int launch_convu2_geom(const PairParams& p, int nch, hipStream_t s);
int launch_convtl_geom(const PairParams& p, int cg, hipStream_t s);     // the lean form for launches with few items (convtl_kernels.hpp)
// ... 32 -> 16 channels, kernel 4, stride 2, padding 1 (HiFi-GAN light's last upsampler): its own kernel (convtn_kernels.hpp),
// its own packed layout (8 KB + 32 inverse row prescales); member 0 as launch_convt (add1 / add2: merged input)
// (the packed layout follows from Cin, Cout, k, stride alone; the op then needs pad = 1, out_pad = 0)
inline bool convtn_shape(int Cin, int Cout, int k, int stride) { return Cin == 32 && Cout == 16 && k == 4 && stride == 2; }
constexpr int kTnPackedFloats = 2048 + 32;
int launch_convtn(const PairParams& p, int Tout, hipStream_t s);
int launch_pack_convtn(const float* w, float* packed, const float* inv, int* range_flag, hipStream_t s);
// MelGAN ResidualStack as one launch (convk_kernels.hpp): member 0 uses x, w1 (fv_pack_residual_stack_split_f16 image),
// b1 (the dilated conv's bias), b2 (stack[4]'s + the skip layer's), y, y_act; p.reflect: ReflectionPad1d
inline bool convk_shape(int C, int k, int dil) { return (C == 32 || C == 64 || C == 128 || C == 256) && k == 3 && (dil == 1 || dil == 3 || dil == 9); }
inline int convk_tile_columns(int C) { return C == 256 ? 32 : 256 / (C / 32); }   // a block holds every row of a column tile
int launch_convk(const PairParams& p, int C, int dil, hipStream_t s);
int launch_pack_convk(const float* w1, const float* w2, const float* ws, float* packed, int C, int* range_flag, hipStream_t s);
// y = post(W1 lrelu(x, slope) + W2 x2 + bias + res), 1-tap convs C -> C with split-f16 operands (convg_kernel): member 0
// uses x, x2, w1 (fv_pack_conv1x1_2src_split_f16 image), b1, res, y, y_act; C = 128, 256 or 512
int launch_convg(PairParams p, int C, hipStream_t stream);
int launch_convg_geom(const PairParams& p, size_t lds, hipStream_t s);
int launch_convr_geom(const PairParams& p, size_t lds, hipStream_t s);     // 128-row tiles (convr_kernels.hpp)
int launch_convs_geom(const PairParams& p, int dil, size_t lds, hipStream_t s);   // convh's convs on 128-row tiles
int launch_convu_geom(const PairParams& p, size_t lds, hipStream_t s);            // convt's GEMM on 128-row tiles
// n (1..3) members, plain (sum = 0: one raw output each) or sum mode (one output: mean of the members)
int launch_pairs(PairParams p, int C, int dil, hipStream_t stream);
template <int MH, int NF, int NG>
int launch_pair_geom(const PairParams& p, int dil, size_t lds, hipStream_t s);
template <int MH, int NF, int NG>
int launch_pairh_geom(const PairParams& p, int dil, size_t lds, hipStream_t s);

// ---- a whole 16-channel MRF stage as one launch (mrfh_kernels.hpp / mrfh_launch.hip) -------------------------------------
struct MrfParams {
    const float* x;          // [B, 16, T] stage input (the upsampler's raw output)
    float* y;                // [B, 16, T] stage output, or null with fold_*
    float* y_act;            // optional activated twin lrelu(y, act_slope), or null
    const float* blob;       // fv_pack_mrf_stage_split_f16 image
    unsigned blob_bytes;
    unsigned blk_off[9];     // byte offset of pair q = 3 j + p inside the blob
    int k[3];                // taps of ResBlock j
    int B, T;
    int halo;                // columns a run's first tile starts early (sum of reaches of the longest ResBlock)
    int ol;                  // FOLD: 3 (the 7-tap output conv's reach), else 0
    float slope;             // leaky slope inside the ResBlocks
    float out_div;           // 3
    float act_slope;         // y_act / FOLD: slope of the activation in front of the output conv; 1 = none
    int post;                // FV_POST_* (FOLD: applied to the output conv's result)
    const float* fold_w;     // [16][7] or null
    const float* fold_b;     // [1] or null
    float* fold_y;           // [B, 1, T]
    int* guard;
    int nblk;
    long long total;         // B * T output columns
    int prio;                // s_setprio inside the K loops (Tuning::mrf_prio)
    unsigned long long* trace;
    float* hist;             // 32 channels: the blocks' history slots (mrf_workspace_bytes), else null
    long long hist_bytes;
};

// packed stage (fv_pack_mrf_stage_split_f16): pair q = 3 j + p of ResBlock j is the block [conv1 image | conv2 image |
// b1[C] | b2[C] | 1 / row prescale of conv1 [C] | of conv2 [C] | padding to a whole KB]; images: fv_pack_pair_weight_ex's
inline int mrf_block_bytes(int C, int k) { return 2 * ((k + (32 / C) - 1) / (32 / C)) * (C / 16) * 2048 + 1024; }
inline bool mrf_stage_shape(int C, const int* k, const int* dil) {
    if ((C != 16 && C != 32) || !k || !dil || dil[0] != 1 || dil[1] != 3 || dil[2] != 5) return false;
    for (int j = 0; j < 3; ++j)
        if (k[j] != 3 && k[j] != 7 && k[j] != 11) return false;
    return true;
}
// columns a run's first tile starts early: the reaches of the longest ResBlock's six convs
inline int mrf_halo(const int* k, const int* dil) {
    int h = 0;
    for (int j = 0; j < 3; ++j) {
        int r = 0;
        for (int q = 0; q < 3; ++q) r += (k[j] - 1) * dil[q] / 2 + (k[j] - 1) / 2;
        if (r > h) h = r;
    }
    return h;
}
// p: x, y / y_act or fold_*, blob, k, B, T, slopes, out_div, post, guard filled in; the rest is set here
int launch_mrfh(MrfParams p, int C, const int* dil, hipStream_t stream);
template <int NF, int NG>
int launch_mrfh_geom(const MrfParams& p, hipStream_t s);
// 32 channels (mrfw_kernels.hpp): bytes of history a launch of at most `blocks` blocks needs, and the launch
constexpr int kMrfwHistBytes = 9 * 30 * 128;
long long mrf_workspace_bytes(int C);
int launch_mrfw_geom(const MrfParams& p, hipStream_t s);

// Shared between the host launcher (conv_mfma.hip) and the kernels (conv_kernels.hpp):
#ifndef FV_RING
#define FV_RING 2          // stage buffers of the LDS-DMA ring in the plain (aligned, zero-padded) kernels.
                           // Measured (HiFi-GAN light, B = 1): 2 -> 1.66 ms/step, 3 -> 1.71 ms (the third
                           // buffer halves the stage size of the 7- and 3-tap kernels under the LDS budget)
#endif
// the SLOW / ACT variants stage some tiles synchronously: always two buffers, full waits
constexpr int kRingStages(bool slow, bool act) { return (slow || act) ? 2 : FV_RING; }
constexpr int kMaxDmaX = 6, kMaxDmaW = 8;   // LDS-DMA instructions per wave per stage (host-checked)

// conv_post + tanh + PQMF synthesis in one launch (conv_post_pqmf_kernel, conv_kernels.hpp): what follows the narrow conv
struct PqmfTail {
    const float* h;       // [S][ntaps] synthesis filter
    float* y;             // [B][S * Tsub] full band
    float* y2;            // optional: y - sub (y stays plain), the bias-removal flows
    const float* sub;     // optional offset [S * Tsub] or [B][S * Tsub]
    int sub_batched;
    int ntaps;
    int tail_off;         // float offset of [tile S x 256 | filter S x ntaps] in dynamic LDS
};
constexpr int kPqmfHalo = 8, kPqmfAdvance = 256 - 2 * kPqmfHalo;   // sub-band samples a block recomputes / advances by
// p: the conv_post as a plain conv (Cout = S = 4 sub-bands, post = tanh); y [B, 1, S * Tq]
int launch_conv_post_pqmf(ConvParams p, const float* h, int ntaps, float* y, float* y2, const float* sub, int sub_batched,
                          hipStream_t stream);

int launch_conv(ConvParams p, hipStream_t stream);
// n mutually independent convs; one grouped launch when they form an MRF position
int launch_conv_group(ConvParams* ps, int n, hipStream_t stream);
int launch_pqmf(const float* x, const float* h, float* y, float* y2, const float* sub, int sub_batched, int B, int S,
                int ntaps, int Tsub, hipStream_t stream);
int launch_encode16(float* x, int B, int64_t n, float rescale, short* out, unsigned* peak_bits,
                    int scale_in_place, hipStream_t s);
int launch_pqmf_analysis(const float* xin, const float* ha, float* x, int B, int S, int ntaps,
                         int64_t T, hipStream_t s);

// self-check of pair_kernels.hpp div_exact against the device's division (fv_div_probe; pair_inst_c16.hip)
int launch_div_probe(unsigned first, long long n, float d, unsigned long long* mismatches, hipStream_t s);

// measurement hook
void profile_begin(hipStream_t stream);
void profile_end(hipStream_t stream, int kind, double flops, double bytes);

}  // namespace fv
