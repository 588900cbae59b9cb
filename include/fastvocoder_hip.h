/*
 * fastvocoder_hip.h -- C ABI of libfastvocoder_hip.so, the MI355X (gfx950)
 * replacement for the ATen calls on FastVocoder's generator forward path.
 *
 * The reference has no FFI on this path: its generators are torch.nn.Modules
 * whose arithmetic is delegated to ATen.  Each entry
 * point below names the reference call site(s) whose arithmetic it replaces
 * (paths relative to /root/reference).  The Python host side
 * (the fastvocoder_amd/generator package) mirrors the reference's module API and is
 * the only caller; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to
 *     contiguous fp32 in the reference's [B, C, T] layout unless stated;
 *   - the library owns no device memory: callers pass inputs, outputs, packed
 *     weights and workspace, and keep them alive until the stream has drained;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every call only enqueues work on it and returns (no device sync);
 *   - return value: 0 on success, otherwise a hipError_t or FV_ERR_* code,
 *     with a thread-local message available from fv_last_error();
 *   - re-entrant per (device, stream); no global mutable state.
 */
#ifndef FASTVOCODER_HIP_H
#define FASTVOCODER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV_ABI_VERSION 13

#define FV_ERR_INVALID_ARG (-1)
#define FV_ERR_UNSUPPORTED (-2)
#define FV_ERR_WORKSPACE (-3)
/* a split-f16 kernel met an operand beyond the f16 range or a non-finite value (fv_plan_check_range): the run's results
 * are not valid; the caller repeats it with FV_PAIR_F32 arithmetic */
#define FV_ERR_RANGE (-4)
/* ... the same for the LOW side of the domain only (operands that were smaller than 2^-10 throughout a block's share of a
 * tensor): the run's results may carry fewer than 22 bits; repeating THIS run with FV_PAIR_F32 is enough -- the condition is
 * a property of the input, not of the model */
#define FV_ERR_RANGE_LOW (-5)
/* the two sides of a `guard` word: separate BYTES of the int32, so that plain stores from different blocks and launches
 * combine (a store to one byte never takes back the other) */
#define FV_GUARD_HIGH 0x001
#define FV_GUARD_LOW 0x100

/* padding of a conv's input: zero "same" padding (torch.nn.Conv1d padding=,
 * model/generator/modules.py:193-221) or reflection padding + valid conv
 * (torch.nn.ReflectionPad1d, modules.py:355-356, melgan.py:68-71) */
#define FV_PAD_ZERO 0
#define FV_PAD_REFLECT 1
/* flag ORed into pad_mode: keep only the first Tin outputs (the reference's CausalConv1d,
 * modules.py:273-294: pad (k-1)*dil on BOTH sides with the configured pad module, valid
 * conv, then [:, :, :T]); pass pad = (k-1)*dil */
#define FV_PAD_CAUSAL 2

/* activation applied after the epilogue adds */
#define FV_POST_NONE 0
#define FV_POST_TANH 1 /* torch.tanh, hifigan.py:106 / melgan.py:108-109 */
#define FV_POST_RELU 2 /* torch.nn.ReLU, basis_melgan.py:120-121 */

int fv_version(void);
/* hash of the sources this binary was built from (fastvocoder_amd/_native.py source_hash()): lets a
 * test prove that the loaded library is the working tree's, not a stale prebuilt one */
const char* fv_build_id(void);
/* thread-local description of the last non-zero return on this thread */
const char* fv_last_error(void);

/* ------------------------------------------------------------------ *
 * one-off weight preparation (at load_state_dict / remove_weight_norm)
 * ------------------------------------------------------------------ */

/* torch.nn.utils.weight_norm reparametrisation w = v * g / ||v||_2, the norm
 * over every dim but 0 (hifigan.py:58-76; for ConvTranspose1d dim 0 is the
 * INPUT channel).  v, w: [dim0, inner]; g: [dim0]. */
int fv_fold_weight_norm(const float* v, const float* g, float* w, int dim0,
                        int64_t inner, void* stream);

/* Eval-mode BatchNorm1d folded into the 1x1/short conv that FOLLOWS it (LastLinear,
 * modules.py:116-132: act -> bn -> conv1x1):  conv(bn(x)) = conv'(x) with
 *   w'[co,ci,j] = w[co,ci,j] * a[ci],  b'[co] = b[co] + sum_{ci,j} w[co,ci,j] * c[ci],
 *   a = gamma / sqrt(var + eps),  c = beta - mean * a.
 * w [Cout,Cin,k]; b [Cout] or NULL (treated as 0); gamma/beta may be NULL (1 / 0);
 * w_out [Cout,Cin,k], b_out [Cout] (always written).  Exact only for valid / unpadded
 * convs (LastLinear's are 1x1). */
int fv_fold_batchnorm_conv(const float* w, const float* b, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, float* w_out,
                           float* b_out, int Cout, int Cin, int k, void* stream);

/* Number of floats of the packed (K-major, M-padded) image of a Conv1d weight
 * [Cout, Cin, k] / of the polyphase image of a ConvTranspose1d weight
 * [Cin, Cout, k] with the given stride and padding. */
int64_t fv_packed_conv1d_floats(int Cout, int Cin, int k);
int64_t fv_packed_conv_transpose1d_floats(int Cin, int Cout, int k, int stride, int pad);

/* w [Cout, Cin, k] (torch.nn.Conv1d.weight) -> packed image */
int fv_pack_conv1d_weight(const float* w, float* packed, int Cout, int Cin, int k,
                          void* stream);
/* w [Cin, Cout, k] (torch.nn.ConvTranspose1d.weight) -> packed polyphase image */
int fv_pack_conv_transpose1d_weight(const float* w, float* packed, int Cin, int Cout,
                                    int k, int stride, int pad, void* stream);

/* ------------------------------------------------------------------ *
 * fused operators
 * ------------------------------------------------------------------ */

/*
 * y     = post( ( (acc_in + acc_in2) + ( conv1d(lrelu(x, pre_slope); w, dil, pad) + bias + res ) ) / out_div )
 * y_act = lrelu(y, act_slope)            (optional second output, see below)
 *
 * Replaces F.leaky_relu + torch.nn.Conv1d (+ the residual add, the MRF running
 * sum / mean and tanh) at modules.py:223-230 (ResBlock1), :247-252 (ResBlock2),
 * :372-382 (ResidualStack), :85-89 (LastLayer), hifigan.py:93,99-106,
 * multiband_hifigan.py:102-115, melgan.py:66-71.
 *   x [B,Cin,Tin]; packed from fv_pack_conv1d_weight; bias [Cout] or NULL;
 *   res, acc_in, acc_in2 [B,Cout,Tout] or NULL (acc_in + acc_in2 is formed first: the
 *   reference's xs = r0; xs += r1; xs += r2 order, hifigan.py:99-102); y [B,Cout,Tout],
 *   Tout = Tin + 2*pad - dil*(k-1) (Tin with FV_PAD_CAUSAL).  pre_slope = 1 disables the input
 *   activation, 0 is ReLU; out_div = 1 disables the division (it is a true
 *   fp32 division, hifigan.py:103).  y may alias res or acc_in, never x.
 *   Activation hoisting: plain VALU work does not overlap the fp32 MFMA on gfx950,
 *   so the fastest way to feed the NEXT conv's F.leaky_relu(x, s) is to have THIS
 *   conv store it.  With y_act != NULL the raw y (for residual consumers) and
 *   y_act = lrelu(y, act_slope) (for conv consumers, which then pass
 *   pre_slope = 1) are both written; with y_act == NULL and act_slope != 1 only
 *   the activated tensor is written, to y.  act_slope = 1 disables both.
 */
int fv_conv1d_fused(const float* x, const float* packed, const float* bias,
                    const float* res, const float* acc_in, const float* acc_in2, float* y,
                    float* y_act, int B, int Cin, int Cout, int Tin, int k, int dil, int pad,
                    int pad_mode, float pre_slope, float out_div, int post, float act_slope,
                    void* stream);

/*
 * Fused ResBlock1 pair (model/generator/modules.py:223-230, one iteration of the loop):
 *
 *     y_j = x_j + conv1d( lrelu( conv1d( lrelu(x_j, slope); w1_j, k_j taps, dilation dil ) + b1_j, slope ); w2_j, k_j taps ) + b2_j
 *
 * for n = 1..3 independent members j in ONE launch -- the pairs at the same position of the three
 * ResBlocks of an MRF stage (hifigan.py:97-103; taps 11 / 7 / 3, same dilation).  Both convs use 'same'
 * zero padding.  The intermediate tensor lives in LDS only; x_j is read raw (the activation is applied
 * on chip) and y_j is written raw, so no activated twin tensors exist between the pairs of a block.
 *   x_j, y_j [B,C,T]; w1_j, w2_j: fv_pack_pair_weight images of the [C,C,k_j] Conv1d weights;
 *   b1_j, b2_j [C] or NULL (the arrays themselves may be NULL); y_act (array or its entries may be
 *   NULL): optional second output lrelu(y_j, act_slope) for a conv1d consumer; with y_act[j] == NULL and
 *   act_slope != 1 only the activated tensor is written, to y_j.
 * Supported: C = 16 or 32, k_j in {3, 7, 11}, dil in {1, 3, 5}, T % 4 == 0 (rows 16-byte aligned);
 * anything else returns FV_ERR_UNSUPPORTED (use two fv_conv1d_fused calls).  Persistent blocks, each
 * with a contiguous cost-balanced share of the (member, utterance, tile) list; every output element is
 * computed the same way whichever block owns its tile: results do not depend on B or on how a batch is
 * split over calls.
 */
int64_t fv_packed_pair_floats(int C, int k);
/* w [C, C, k] (torch.nn.Conv1d.weight of a ResBlock conv) -> the A-fragment image the pair kernels read */
int fv_pack_pair_weight(const float* w, float* packed, int C, int k, void* stream);
int fv_resblock1_fused(int n, const float* const* x, const float* const* w1, const float* const* w2,
                       const float* const* b1, const float* const* b2, float* const* y, float* const* y_act,
                       const int* k, int B, int C, int T, int dil, float slope, float act_slope, void* stream);

/*
 * The same operator with a choice of arithmetic, and -- for the last launch of an MRF stage -- the merge
 * of hifigan.py:99-103 in the epilogue.
 *   prec = FV_PAIR_F32:       v_mfma_f32_16x16x4_f32, exact fp32 products (the functions above).
 *   prec = FV_PAIR_SPLIT_F16: every fp32 operand v is split as v = h1 + h2/2048 (+ <= 2^-22 |v|), h1, h2 f16,
 *       and a product is three v_mfma_f32_16x16x32_f16 terms a1 b1 + (a1 b2 + a2 b1)/2048 accumulated in
 *       fp32.  Per layer the result is as close to the exact sum as an fp32 FMA chain (measured: DESIGN.md
 *       section 3.7, tests/test_split_precision.py); 5.3x fewer matrix-core cycles, which turns the C = 16
 *       layers from matrix-bound into HBM / LDS-bound.  DOMAIN -- where the reference (fp32 ATen) is defined for any
 *       finite fp32 -- enforced, not assumed:
 *         weights: any finite fp32.  The pack functions scale every GEMM row by a power of two so that its largest
 *           magnitude lies in [2^13, 2^14) (exact; the f16 pair keeps 22 bits of every element down to 2^-28 of the
 *           row's maximum) and store the inverse behind the image; the kernels multiply the accumulated sum by it inside
 *           the fused multiply-add that adds the bias (exact).  `range_flag` is raised for a non-finite weight.
 *         activations, high side: |v| < 65520.  An operand beyond the f16 range turns into inf in f16 and every output
 *           it feeds into inf / NaN: every split-f16 kernel raises its `guard` word (FV_GUARD_HIGH: byte 0) when a final value is not
 *           finite.
 *         activations, low side: below 2^-14 the pair keeps an absolute 2^-36 instead of 22 bits -- harmless inside an
 *           ordinary tensor, a loss for one that is small as a whole (with correspondingly large weights behind it).
 *           Every kernel keeps the largest magnitude of the operands it splits; a block whose operands were not all zero
 *           and all below 2^-10 raises `guard` (FV_GUARD_LOW: byte 1; the two sides never overwrite each other).
 *         In both cases the caller repeats the layer / the run with FV_PAIR_F32 (fv_plan_check_range).
 *       range_flag / guard: int32 words any kernel can write (device memory or pinned host memory), NULL = no check.
 *       Weights: fv_pack_pair_weight_ex(prec) images ([K step][row half][split half][lane][8 f16], then one float per
 *       row: the inverse prescale).
 *       C = 16 / 32: one fused launch (intermediate and weights in LDS).  C = 64: one fused launch, the weights of
 *       both convs stream through an LDS ring (csrc/convp_kernels.hpp).  C = 128 / 256 / 512: the pair's two LDS images do not
 *       fit next to the ring; it runs as two launches of the split-f16 conv kernel (csrc/convh_kernels.hpp) and
 *       needs mid[j], a [B,C,T] scratch tensor per member (ignored at C = 64, NULL below).
 *   add1 / add2 (arrays or entries may be NULL; FV_PAIR_SPLIT_F16 only): member j stores
 *       y_j = post( ((x'_j + add1_j) + add2_j) / out_div )  -- with x'_j the first ResBlock's result and
 *       add1 / add2 the second and third this is xs = r0; xs += r1; xs += r2; x = xs / 3 in the
 *       reference's association.  Members without add1 ignore out_div / post.
 */
#define FV_PAIR_F32 0
#define FV_PAIR_SPLIT_F16 1
int64_t fv_packed_pair_floats_ex(int C, int k, int prec);
int fv_pack_pair_weight_ex(const float* w, float* packed, int C, int k, int prec, int* range_flag, void* stream);
int fv_resblock1_fused_ex(int n, const float* const* x, const float* const* w1, const float* const* w2,
                          const float* const* b1, const float* const* b2, float* const* y, float* const* y_act,
                          float* const* mid, const float* const* add1, const float* const* add2, const int* k, int B,
                          int C, int T, int dil, float slope, float out_div, int post, float act_slope, int prec,
                          int* guard, void* stream);

/*
 * A whole MRF stage of 16 channels as ONE launch (reference model/generator/hifigan.py:97-103 around
 * model/generator/modules.py:223-230), split-f16 operands (arithmetic and domain: FV_PAIR_SPLIT_F16 above):
 *
 *     r_j = ResBlock1_j(x),  j = 0, 1, 2:   three times  x <- x + conv1d( lrelu( conv1d( lrelu(x, slope); w1, k_j taps,
 *                                            dilation dil[p] ) + b1, slope ); w2, k_j taps ) + b2      (p = 0, 1, 2)
 *     y   = post( ((r_0 + r_1) + r_2) / out_div ),      y_act = lrelu(y, act_slope)
 *
 * -- the reference's association of the sum; every conv with 'same' zero padding.  What fv_resblock1_fused_ex computes in
 * four dependent launches per stage (three pair positions, the last one split for the merge), each of which reads and
 * writes the stage's tensor, is here one pass: a tile's columns stay on the CU through all 18 convs (running x in
 * registers, operand images and a few columns of history per conv in LDS, csrc/mrfh_kernels.hpp), x is read once and y
 * written once.  Bit-identical to the pair launches.
 *   x, y, y_act [B, C, T] (y_act or NULL; without it and act_slope != 1, y itself is stored activated); any T.
 *   k[3]: taps of the three ResBlocks, each 3, 7 or 11; dil[3] = {1, 3, 5}; C = 16 or 32.  Anything else: FV_ERR_UNSUPPORTED.
 *   C = 32 (csrc/mrfw_kernels.hpp: weights stream through LDS in pieces of four taps, the columns of history live in
 *   `workspace`): workspace = fv_mrf_stage_workspace_bytes(32) bytes of device memory, 16-byte aligned, that no other launch
 *   in flight uses (contents: don't care, before and after); C = 16: workspace may be NULL.
 *   packed: fv_pack_mrf_stage_split_f16 of the 18 Conv1d weights and biases -- w1[3 j + p], w2[3 j + p]: [C, C, k_j];
 *   b1 / b2: arrays of 9 pointers to [C] (entries, or the arrays, may be NULL); fv_packed_mrf_stage_floats floats
 *   (0 = shape not built).  Per pair: [conv1 image | conv2 image | b1 | b2 | 1 / row prescale of conv1 | of conv2].
 *   fold_w [C, 7] != NULL (HiFi-GAN's conv_post, hifigan.py:104-106, inside the launch; needs y = y_act = NULL):
 *       fold_y[B, 1, T] = post( conv1d( lrelu(((r_0 + r_1) + r_2) / out_div, act_slope); fold_w, zero padding 3 ) + fold_b )
 *   -- the same arithmetic, bit for bit, as fv_conv1d_fused on a stored y.
 */
int64_t fv_packed_mrf_stage_floats(int C, const int* k);
int fv_pack_mrf_stage_split_f16(const float* const* w1, const float* const* w2, const float* const* b1,
                                const float* const* b2, float* packed, int C, const int* k, int* range_flag, void* stream);
int64_t fv_mrf_stage_workspace_bytes(int C);
int fv_mrf_stage_split_f16(const float* x, const float* packed, float* y, float* y_act, int B, int C, int T, const int* k,
                           const int* dil, float slope, float out_div, int post, float act_slope, const float* fold_w,
                           const float* fold_b, float* fold_y, void* workspace, int64_t workspace_bytes, int* guard,
                           void* stream);

/*
 * Conv1d of the wide ResBlock stages with split-f16 operands (arithmetic: FV_PAIR_SPLIT_F16 above), n = 1..3
 * independent members in one launch -- the convs at the same position of the three ResBlocks of an MRF stage:
 *
 *     y_j = post( ( conv1d( lrelu(x_j, pre_slope); w_j, k_j taps, dilation dil, 'same' padding ) + bias_j
 *                   + res_j + add1_j + add2_j ) / out_div ),      y_act_j = lrelu(y_j, act_slope)
 *
 * pad_mode: FV_PAD_ZERO (the HiFi-GAN ResBlock convs) or FV_PAD_REFLECT (the dilated convs of MelGAN's
 * ResidualStack, modules.py:351-359: ReflectionPad1d((k-1)/2*dil) in front of a 3-tap conv; needs (k-1)/2*dil < T).
 * (modules.py:223-230: conv1 of a pair is res = add = NULL; conv2 is res = the pair's input; the last conv of a
 * stage's first block carries the MRF merge, hifigan.py:99-103, with add1 / add2 = the other blocks' results.)
 * C = Cin = Cout = 64, 128, 256 or 512, k_j in {3, 7, 11}, dil in {1, 3, 5} (and 9 with k_j = 3); x_j is read RAW (the activation is applied on
 * chip while the operand is split); packed_j: fv_pack_pair_weight_ex(C, k_j, FV_PAIR_SPLIT_F16) -- the packed
 * weights stream L2 -> LDS through a 4-stage ring (csrc/convh_kernels.hpp).  out_div / post apply to members with
 * add1 only.  Any T.
 */
int fv_conv1d_split_f16(int n, const float* const* x, const float* const* packed, const float* const* bias,
                        const float* const* res, const float* const* add1, const float* const* add2, float* const* y,
                        float* const* y_act, const int* k, int B, int C, int T, int dil, int pad_mode, float pre_slope,
                        float out_div, int post, float act_slope, int* guard, void* stream);

/*
 * ConvTranspose1d with split-f16 operands (arithmetic: FV_PAIR_SPLIT_F16 above) for the upsamplers whose kernel is two
 * strides long -- every one the reference builds (hifigan.py:45-46, melgan.py:37-39: kernel 2 s, padding s/2 + s%2,
 * output_padding s%2):
 *
 *     y = conv_transpose1d( lrelu(x, pre_slope); w [Cin, Cout, k], stride, pad, out_pad ) + bias,  y_act = lrelu(y, act_slope)
 *
 * (without y_act and act_slope != 1, y itself is stored activated).  Cin = 64, 128, 256 or 512; k = 2 stride, stride 2..16;
 * Cout * stride >= 64 (rows are padded to a multiple of 64, 64 input channels to 128); 0 <= pad <= stride; out_pad in [-stride, stride) (negative: CausalConvTranspose1d's
 * trim, modules.py:297-317).  Tout = (Tin-1)*stride - 2*pad + k + out_pad.  With n + pad = stride u + phase every
 * output sample has the two taps x[u], x[u-1]: one GEMM with rows (output channel, phase), weights streamed as in
 * fv_conv1d_split_f16 (csrc/convh_kernels.hpp, convt_kernel).  packed: fv_pack_conv_transpose1d_split_f16
 * (fv_packed_conv_transpose1d_split_floats floats; 0 = shape not supported).
 */
int64_t fv_packed_conv_transpose1d_split_floats(int Cin, int Cout, int k, int stride);
int fv_pack_conv_transpose1d_split_f16(const float* w, float* packed, int Cin, int Cout, int k, int stride, int* range_flag,
                                       void* stream);
int fv_conv_transpose1d_split_f16(const float* x, const float* packed, const float* bias, float* y, float* y_act, int B,
                                  int Cin, int Cout, int Tin, int k, int stride, int pad, int out_pad, float pre_slope,
                                  float act_slope, int* guard, void* stream);

/*
 * End of an MRF stage (hifigan.py:97-103): the LAST pairs of the three ResBlocks and the mean, one launch:
 *
 *     y = post( ( sum_{j<3} pair_j(x_j) ) / out_div ),   y_act = lrelu(y, act_slope)   (as fv_conv1d_fused)
 *
 * pair_j as in fv_resblock1_fused (taps 11 / 7 / 3 in any order).  The three results are summed inside
 * the fp32 accumulator rather than as ((r0 + r1) + r2): equal up to fp32 rounding of the additions.
 * Supported: C = 16 (the three weight sets stay resident in LDS); otherwise FV_ERR_UNSUPPORTED --
 * use fv_conv1d_fused for the first convs and the sum3 plan op / running-sum epilogues for the merge.
 */
int fv_mrf_stage(const float* const* x, const float* const* w1, const float* const* w2, const float* const* b1,
                 const float* const* b2, float* y, float* y_act, const int* k, int B, int C, int T, int dil,
                 float slope, float out_div, int post, float act_slope, void* stream);

/*
 * y = post( W1 * x + W2 * x2 + bias + res )    (two 1-tap convs that are summed, as ONE GEMM)
 *
 * Replaces ResidualStack's tail (modules.py:362-366,382): stack[4](act(h)) + skip_layer(c) --
 * the K range of the 1x1 conv is the concatenation of the two input tensors, so the skip
 * branch costs no launch, no [B,C,T] round trip through HBM and no separate add.
 *   x [B,Cin1,T], x2 [B,Cin2,T] (both already activated as the graph requires: there is no
 *   input activation here), packed = fv_pack_conv1d_weight of cat([W1, W2], dim=1)
 *   [Cout, Cin1+Cin2, 1], bias = b1 + b2 (or NULL), res [B,Cout,T] or NULL, Cout > 4.
 */
int fv_conv1d_2src_fused(const float* x, const float* x2, const float* packed, const float* bias,
                         const float* res, float* y, float* y_act, int B, int Cin1, int Cin2,
                         int Cout, int T, int post, float act_slope, void* stream);

/*
 * The same operator with split-f16 operands (arithmetic and domain: FV_PAIR_SPLIT_F16 above) for C -> C stacks,
 * C = 128, 256 or 512 -- MelGAN's and Basis-MelGAN's ResidualStack tails (modules.py:362-366,382):
 *
 *     y = post( W1 * lrelu(x, pre_slope) + W2 * x2 + bias + res ),   y_act = lrelu(y, act_slope)
 *
 * x is read RAW and activated on chip while it is split (nothing is hoisted into its producer), x2 is read as it is.
 * One GEMM over the K range [lrelu(x); x2] on the streamed-weight pipeline of fv_conv1d_split_f16 (csrc/convh_kernels.hpp
 * convg_kernel).  packed: fv_pack_conv1x1_2src_split_f16 of the two [C, C, 1] weights (fv_packed_conv1x1_2src_split_floats
 * floats; 0 = C not supported); bias = b1 + b2 or NULL.
 */
int64_t fv_packed_conv1x1_2src_split_floats(int C);
int fv_pack_conv1x1_2src_split_f16(const float* w1, const float* w2, float* packed, int C, int* range_flag, void* stream);
int fv_conv1x1_2src_split_f16(const float* x, const float* x2, const float* packed, const float* bias, const float* res,
                              float* y, float* y_act, int B, int C, int T, float pre_slope, int post, float act_slope,
                              int* guard, void* stream);

/*
 * MelGAN's ResidualStack (modules.py:351-382) as ONE launch, C = 32, 64, 128 or 256 channels, 3 taps, dilation 1, 3 or 9,
 * split-f16 operands (arithmetic and domain: FV_PAIR_SPLIT_F16 above):
 *
 *     y = post( W2 * lrelu( conv1d( pad( lrelu(x, slope) ); w_dilated, dil ) + bias_dilated, slope ) + Ws * x + bias_out )
 *
 * stack = [act, pad, Conv1d(C, C, 3, dilation), act, Conv1d(C, C, 1)] (modules.py:362-366), skip_layer = Conv1d(C, C, 1)
 * of the raw input (:377, :382); bias_out = stack[4].bias + skip_layer.bias or NULL; pad_mode FV_PAD_ZERO or
 * FV_PAD_REFLECT (`dil` samples on either side: the 'same' padding of the dilated conv).  The hidden tensor never
 * leaves the CU (csrc/convk_kernels.hpp); at 128 and 256 channels the result is bit-identical to fv_conv1d_split_f16 followed
 * by fv_conv1x1_2src_split_f16.  y_act (or NULL): lrelu(y, act_slope); without y_act and act_slope != 1, y itself is
 * stored activated.  packed: fv_pack_residual_stack_split_f16 of the three weights [C, C, 3], [C, C, 1], [C, C, 1]
 * (fv_packed_residual_stack_floats floats; 0 = shape not built).
 */
int64_t fv_packed_residual_stack_floats(int C, int k);
int fv_pack_residual_stack_split_f16(const float* w_dilated, const float* w_pointwise, const float* w_skip, float* packed,
                                     int C, int k, int* range_flag, void* stream);
int fv_residual_stack_split_f16(const float* x, const float* packed, const float* bias_dilated, const float* bias_out, float* y,
                                float* y_act, int B, int C, int T, int k, int dil, float slope, int pad_mode, int post,
                                float act_slope, int* guard, void* stream);

/*
 * y = post( conv_transpose1d(lrelu(x, pre_slope); w, stride, pad, out_pad) + bias )
 *
 * Replaces F.leaky_relu + torch.nn.ConvTranspose1d at hifigan.py:95-96,
 * multiband_hifigan.py:104-105, melgan.py:75-85, basis_melgan.py:86-97, and --
 * with Cout = 1, k = L, stride = L/2, pad = 0, packed from W^T -- the
 * F.linear + overlap_and_add pair of BasisSignalLayer (modules.py:264-267,
 * :34-73).  x [B,Cin,Tin] -> y [B,Cout,Tout],
 * Tout = (Tin-1)*stride - 2*pad + k + out_pad; -stride <= out_pad < stride: a negative out_pad drops
 * the tail of the output -- CausalConvTranspose1d (modules.py:297-317) is pad = 0, out_pad = -stride.
 */
int fv_conv_transpose1d_fused(const float* x, const float* packed, const float* bias,
                              float* y, float* y_act, int B, int Cin, int Cout, int Tin,
                              int k, int stride, int pad, int out_pad, float pre_slope,
                              int post, float act_slope, void* stream);

/*
 * BasisSignalLayer + overlap_and_add (reference model/generator/modules.py:255-267, :34-73; Basis-MelGAN's last step,
 * basis_melgan.py:140-162): frames = weight W^T (F.linear, no bias), out[hop f + j] += frames[f, j], hop = L / 2.
 *   weight [B, C, F]: the trunk's output as it lies in memory (channel-major: the reference's transpose(1, 2) view);
 *   W [L, C]: basis_signal.layer.weight;   out [B, 1, hop (F - 1) + L]   (L = 30, C = 256: [B, 15 F + 15]).
 * One launch: the op is a ConvTranspose1d(C -> 1, kernel L, stride hop, pad 0) with weight W^T on the fp32-MFMA kernel
 * (fv_conv_transpose1d_fused); the [B, F, L] frame tensor is never materialised.  packed_basis: fv_pack_basis of W
 * (fv_packed_basis_floats(L, C) floats: the packed image and, behind it, the transposed weight it was made from).
 */
int64_t fv_packed_basis_floats(int L, int C);
int fv_pack_basis(const float* W, float* packed, int L, int C, void* stream);
int fv_basis_ola(const float* weight, const float* packed_basis, float* out, int B, int C, int F, int L, void* stream);

/*
 * y = post( conv1d( nearest_repeat(lrelu(x, pre_slope), rate); w, padding = pad ) + bias )
 *
 * Replaces the reference's UpsampleLayer (modules.py:160-177: Stretch2d nearest x rate,
 * then Conv1d(k, padding)), selected by `transposedconv: False`.  The repeat is never
 * materialised: taps that read the same input sample are summed at pack time
 * (fv_pack_upsample_conv1d_weight) and the layer runs as a short dense conv over
 * Cout*rate phase rows.  x [B,Cin,Tin] -> y [B,Cout,Tout], Tout = rate*Tin + 2*pad - (k-1).
 */
int64_t fv_packed_upsample_conv1d_floats(int Cout, int Cin, int k, int rate, int pad);
int fv_pack_upsample_conv1d_weight(const float* w, float* packed, int Cout, int Cin, int k,
                                   int rate, int pad, void* stream);
int fv_upsample_conv1d_fused(const float* x, const float* packed, const float* bias, float* y,
                             float* y_act, int B, int Cin, int Cout, int Tin, int k, int rate,
                             int pad, float pre_slope, int post, float act_slope, void* stream);

/*
 * PQMF.synthesis (model/generator/pqmf.py:121-135) in polyphase form.
 *   x [B,S,Tsub] sub-bands, h [S,ntaps] (= synthesis_filter[0]), y [B,S*Tsub].
 */
int fv_pqmf_synthesis(const float* x, const float* h, float* y, int B, int S, int ntaps,
                      int Tsub, void* stream);

/*
 * Multiband-HiFi-GAN's inference tail in ONE launch (multiband_hifigan.py:113-115,136 with pqmf.py:121-135):
 *
 *     y = pqmf_synthesis( post( conv1d( lrelu(x, pre_slope); w [S, Cin, k], zero padding pad ) + bias ) )
 *
 * conv_post's S = 4 activated sub-bands stay in LDS and feed the polyphase synthesis (ntaps = 63) directly: the
 * [B, S, T] sub-band tensor never exists and one launch disappears.  Same FMA order as fv_conv1d_fused followed by
 * fv_pqmf_synthesis: identical bits.  x [B, Cin, T]; packed: fv_pack_conv1d_weight of w; h [S, ntaps]; y [B, S * T].
 */
int fv_conv_post_pqmf(const float* x, const float* packed, const float* bias, const float* h, float* y, int B, int Cin,
                      int S, int T, int k, int pad, float pre_slope, int post, int ntaps, void* stream);

/*
 * PQMF.analysis (pqmf.py:108-119): ntaps-tap FIR per band over the zero-padded signal,
 * decimated by S; only the kept samples are computed.
 *   x [B,T] full band, h [S,ntaps] (= analysis_filter[:,0]), y [B,S,(T-S)/S+1].
 * Not on the inference path (the reference uses it for the multiband training loss); it is
 * here so the PQMF class is whole and for the analysis->synthesis reconstruction check.
 */
int fv_pqmf_analysis(const float* x, const float* h, float* y, int B, int S, int ntaps,
                     int64_t T, void* stream);

/*
 * The wav sink's arithmetic (data/audio.py:12-14 encode_16bits) on the device, per waveform
 * (row) of x [B,n]:  s = 32767 / max(0.01, max|x|) * rescale_out;  out = int16(trunc(x * s)).
 * peak [B] receives max|x| per row (device scratch, also an output).  scale_in_place != 0
 * also writes x * s back to x, the reference's in-place mutation of its argument.
 * Results equal numpy's for a float32 array bit for bit (tests/test_gpu_parity.py).
 */
int fv_encode_16bits(float* x, int16_t* out, float* peak, int B, int64_t n, float rescale_out,
                     int scale_in_place, void* stream);

/* ------------------------------------------------------------------ *
 * whole-generator plans: an op list replayed over a caller-owned arena
 * ------------------------------------------------------------------ */

typedef struct fv_plan fv_plan_t;

/* tensor slots inside a plan: FV_SLOT_IN is the caller's mel [B,Cin0,T],
 * FV_SLOT_OUT the caller's output, slots >= FV_SLOT_TMP0 live in the workspace */
#define FV_SLOT_NONE (-1)
#define FV_SLOT_IN 0
#define FV_SLOT_OUT 1
#define FV_SLOT_TMP0 2
#define FV_MAX_SLOTS 32
/* caller-provided auxiliary tensors of fv_plan_run_aux: two read-only inputs (output offsets) and a second
 * output; temporaries live in [FV_SLOT_TMP0, FV_SLOT_AUX_IN0) */
#define FV_SLOT_AUX_IN0 28
#define FV_SLOT_AUX_IN1 29
#define FV_SLOT_OUT2 30

fv_plan_t* fv_plan_create(int in_channels);
void fv_plan_destroy(fv_plan_t* plan);

/* append ops; argument meaning as in the fused operators above, tensors named
 * by slot.  Weight pointers are captured, not copied. */
int fv_plan_add_conv1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot,
                       int res_slot, int acc_slot, int acc2_slot, const float* packed,
                       const float* bias, int Cin, int Cout, int k, int dil, int pad,
                       int pad_mode, float pre_slope, float out_div, int post,
                       float act_slope);
int fv_plan_add_conv_transpose1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot,
                                 const float* packed, const float* bias, int Cin, int Cout,
                                 int k, int stride, int pad, int out_pad, float pre_slope,
                                 int post, float act_slope);
int fv_plan_add_conv1d_2src(fv_plan_t* plan, int x_slot, int x2_slot, int y_slot, int y_act_slot,
                            int res_slot, const float* packed, const float* bias, int Cin1,
                            int Cin2, int Cout, int post, float act_slope);
int fv_plan_add_conv1x1_2src_split_f16(fv_plan_t* plan, int x_slot, int x2_slot, int y_slot, int y_act_slot, int res_slot,
                                       const float* packed, const float* bias, int C, float pre_slope, int post,
                                       float act_slope);
int fv_plan_add_residual_stack_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed,
                                         const float* bias_dilated, const float* bias_out, int C, int k, int dil, float slope,
                                         int pad_mode, int post, float act_slope);
/* The residual stack recorded last (256 channels) also carries its two-launch form -- packed_dilated: fv_pack_pair_weight_ex
 * (FV_PAIR_SPLIT_F16) of the dilated conv, packed_pair: fv_pack_conv1x1_2src_split_f16 of (stack[4], skip_layer), hidden_slot:
 * scratch for the hidden tensor -- and a run picks by size (fv_tuning_set "stack_items": tiles per CU, in tenths, up to
 * which the one-launch kernel runs; default: always -- it measured faster at every batch size; 0: never).  Identical bits
 * either way: the switch exists for A/B runs and tests. */
int fv_plan_set_stack_two_launch(fv_plan_t* plan, int hidden_slot, const float* packed_dilated, const float* packed_pair);
/*
 * y = act( ( sum_{j<3} ( conv1d(x_j; w_j, k_j taps, 'same' zero padding) + res_j ) + bias_sum ) / out_div )
 *
 * The MRF merge of a HiFi-GAN stage (hifigan.py:97-103 with modules.py:226-229): the last convs of
 * the three ResBlocks (taps 11 / 7 / 3 in any order, undilated, C -> C) accumulate into one
 * output tile in ONE launch; bias_sum = b_0 + b_1 + b_2.  The three partial results are summed
 * inside the fp32 accumulator rather than as ((r0 + r1) + r2): equal up to fp32 rounding of
 * the additions (the conv1d ops with acc / acc2 keep the reference's association exactly).
 * x_slots / res_slots / packed / k: arrays of 3.  When the layer has too few tiles for one launch
 * to fill the GPU (utterance-length dependent, never batch dependent) the op runs as two launches
 * instead -- members 1 and 2 into the two [B,C,T] scratch slots tmp_slots[0..1], then member 0
 * with both as running-sum inputs; same sum.
 */
int fv_plan_add_conv1d_sum3(fv_plan_t* plan, const int* x_slots, const int* res_slots,
                            const int* tmp_slots, int y_slot, int y_act_slot,
                            const float* const* packed, const float* bias_sum, int C, const int* k,
                            float out_div, int post, float act_slope);
/* fv_resblock1_fused / fv_mrf_stage as plan ops.  Consecutive resblock-pair ops appended under the same
 * non-zero group id (fv_plan_set_group) with equal C, dilation and slopes run as ONE launch. */
int fv_plan_add_resblock_pair(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed1,
                              const float* packed2, const float* bias1, const float* bias2, int C, int k, int dil,
                              float slope, float act_slope);
/* fv_resblock1_fused_ex as a plan op (add1_slot / add2_slot: FV_SLOT_NONE or [B,C,T] slots; mid_slot: a scratch
 * slot, needed at C >= 64 with FV_PAIR_SPLIT_F16, FV_SLOT_NONE otherwise) */
int fv_plan_add_resblock_pair_ex(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, int mid_slot, int add1_slot,
                                 int add2_slot, const float* packed1, const float* packed2, const float* bias1,
                                 const float* bias2, int C, int k, int dil, float slope, float out_div, int post,
                                 float act_slope, int prec);
/* The last pair of a HiFi-GAN with conv_post folded in (hifigan.py:97-106): applies to the op appended last, which must
 * be an ungrouped 16-channel FV_PAIR_SPLIT_F16 pair (with or without add1 / add2).  The pair's own output x' is never
 * stored; the op's output becomes
 *     y[B, 1, T] = post( conv1d( lrelu( x' / out_div, act_slope ); w [1, 16, 7], zero padding 3 ) + bias ),
 * computed on the activated tile in LDS (tiles overlap by the conv's 3-sample halo).  One launch and a [B,16,T]
 * round trip less than the pair followed by fv_conv1d_fused. */
int fv_plan_set_pair_output_conv(fv_plan_t* plan, const float* w, const float* bias, int y_slot, float act_slope, int post);
/* fv_mrf_stage_split_f16 as a plan op (y_act_slot may be FV_SLOT_NONE; workspace: as there, owned by the caller for the
 * plan's lifetime, one per op).  fv_plan_set_pair_output_conv (w [1, C, 7]: 16 or 32 channels here) also applies to this
 * op when it was appended last (the stage's own y is then not stored; y_slot receives the folded conv's [B, 1, T]). */
int fv_plan_add_mrf_stage_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed, int C,
                                    const int* k, const int* dil, float slope, float out_div, int post, float act_slope,
                                    void* workspace, int64_t workspace_bytes);
/* fv_conv1d_split_f16 as a plan op; consecutive ops under one non-zero group id with equal C, dilation, padding
 * mode, slopes, out_div and post run as ONE launch */
int fv_plan_add_conv1d_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, int res_slot, int add1_slot,
                                 int add2_slot, const float* packed, const float* bias, int C, int k, int dil,
                                 int pad_mode, float pre_slope, float out_div, int post, float act_slope);
/* fv_conv_transpose1d_split_f16 as a plan op */
int fv_plan_add_conv_transpose1d_split_f16(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot, const float* packed,
                                           const float* bias, int Cin, int Cout, int k, int stride, int pad,
                                           int out_pad, float pre_slope, float act_slope);
/* The input of the split-f16 transposed conv just added becomes ((x + add1) + add2) / div, formed on chip while its window
 * is loaded (add2_slot may be FV_SLOT_NONE; slots shaped like x).  This is the MRF merge of reference hifigan.py:99-103 --
 * xs = r0; xs += r1; xs += r2; x = xs / num_kernels, in that association -- moved from the end of a stage into the upsampler
 * behind it, so that the stage's last pair position is ONE launch of its three ResBlocks (each storing its r_j) instead of
 * two launches.  The values are formed exactly as the stage-end launch forms them: identical bits. */
int fv_plan_set_input_merge(fv_plan_t* plan, int add1_slot, int add2_slot, float div);
int fv_plan_add_mrf_sum(fv_plan_t* plan, const int* x_slots, int y_slot, int y_act_slot,
                        const float* const* packed1, const float* const* packed2, const float* const* bias1,
                        const float* const* bias2, int C, const int* k, int dil, float slope, float out_div,
                        int post, float act_slope);
int fv_plan_add_upsample_conv1d(fv_plan_t* plan, int x_slot, int y_slot, int y_act_slot,
                                const float* packed, const float* bias, int Cin, int Cout,
                                int k, int rate, int pad, float pre_slope, int post,
                                float act_slope);
int fv_plan_add_pqmf_synthesis(fv_plan_t* plan, int x_slot, int y_slot, const float* h,
                               int S, int ntaps);
/* fv_conv_post_pqmf as a plan op: y_slot receives the FULL-BAND signal [B, 1, S * T']; may carry an output offset
 * (fv_plan_set_output_offset) like the pqmf op */
int fv_plan_add_conv_post_pqmf(fv_plan_t* plan, int x_slot, int y_slot, const float* packed, const float* bias, int Cin,
                               int S, int k, int pad, float pre_slope, int post, const float* h, int ntaps);

/*
 * Bias removal in the epilogue (bin/synthesize.py:74-80 est - generator(0); basis_melgan.py:147-159
 * (est - zero_est, weight - zero_weight); bin/test.py:82-91 est - pattern): the op appended LAST (a conv1d,
 * conv_transpose1d / upsample conv or the pqmf synthesis) subtracts the auxiliary input `aux_slot`
 * (FV_SLOT_AUX_IN0/1, given to fv_plan_run_aux: the cached zero-input response, [C,T'] or [B,C,T']) after its
 * post op / activation.  With y2_slot != FV_SLOT_NONE the op keeps y raw and writes y2 = y - aux (y2 may be
 * FV_SLOT_OUT2 or a temporary); otherwise y itself becomes y - aux.
 */
int fv_plan_set_output_offset(fv_plan_t* plan, int aux_slot, int y2_slot);

/* Association of the MRF running sum in the epilogue of the conv1d ops added from now on:
 * 0 (default)  y = ((acc + acc2) + own) / out_div   -- the LAST ResBlock's conv carries the sum
 * 1            y = ((own + acc) + acc2) / out_div   -- the FIRST ResBlock's conv carries it
 * (own = conv + bias + res).  Both reproduce xs = r0; xs += r1; xs += r2 (hifigan.py:99-102)
 * bit for bit; 1 lets the two large-kernel blocks finish as one grouped launch while the
 * cheapest (3-tap) conv forms the sum. */
int fv_plan_set_sum_order(fv_plan_t* plan, int own_first);

/* Grouping: consecutive conv1d ops appended under the same non-zero group id are
 * declared mutually independent by the caller.  When they are the 11/7/3-tap
 * convolutions at one position of the three ResBlocks of an MRF stage (same
 * shapes, dilation and activation state) the executor runs them as ONE launch
 * (blockIdx.z selects the problem); otherwise one launch each.  0 ends a group. */
int fv_plan_set_group(fv_plan_t* plan, int group);

/* shape inference for a (B, T) call: channels / length of the output tensor
 * and the workspace the plan needs (bytes) */
int fv_plan_output_shape(fv_plan_t* plan, int T, int* out_channels, int64_t* out_len);
int fv_plan_slot_shape(fv_plan_t* plan, int T, int slot, int* channels, int64_t* len);
int64_t fv_plan_workspace_bytes(fv_plan_t* plan, int B, int T);

/* enqueue the whole op list: in [B,Cin0,T] -> out [B,Cout,Tout] */
int fv_plan_run(fv_plan_t* plan, int B, int T, const float* in, float* out,
                void* workspace, int64_t workspace_bytes, void* stream);

/* fv_plan_run with the auxiliary tensors: out2 (FV_SLOT_OUT2; NULL if the plan writes none), aux_in[2]
 * (FV_SLOT_AUX_IN0/1; entries NULL when unused) and aux_batched[2] (non-zero: that input has a batch dimension) */
int fv_plan_run_aux(fv_plan_t* plan, int B, int T, const float* in, float* out, float* out2,
                    const float* const* aux_in, const int* aux_batched, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* The whole-graph entry under the name SURVEY.md section 8(b) lists: the generator forward of a built plan,
 * mel [B,Cin0,T] -> out [B,Cout,Tout] (fv_plan_run; workspace: fv_plan_workspace_bytes(plan, B, T)). */
int fv_generator_run(fv_plan_t* plan, int B, int T, const float* mel, float* out, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* Self-check (tests): the MRF mean's divisor is applied as q0 = v r, q = fma(fma(-d, q0, v), r, q0), r = RN(1 / d) for small
 * integer d (csrc/pair_kernels.hpp div_exact) -- claimed to be the correctly rounded v / d.  Counts, over the fp32 values with
 * bit patterns [first_bits, first_bits + n) and their negatives, where that differs from the device's IEEE division (values
 * whose quotient is not a normal number are skipped); *mismatches (device memory, zeroed by the caller) receives the count. */
int fv_div_probe(unsigned first_bits, int64_t n, float d, unsigned long long* mismatches, void* stream);

/* number of kernel launches one fv_plan_run enqueues */
int fv_plan_num_ops(fv_plan_t* plan);

/*
 * Range guard of a plan's split-f16 launches (FV_PAIR_SPLIT_F16 above: operands must lie inside the f16 range).
 * fv_plan_set_guard: `word` is an int32 in PINNED, DEVICE-MAPPED host memory (hipHostMalloc / torch pin_memory), owned by
 *   the caller and zero-initialised; every split-f16 kernel the plan launches raises FV_GUARD_HIGH in it when one of its final
 *   values is not finite, FV_GUARD_LOW for the low side.  NULL removes the guard.
 * fv_plan_check_range: waits for `stream` (the stream of the plan's last run) to drain, then returns 0 when the word is
 *   clear; otherwise clears it and returns FV_ERR_RANGE (FV_ERR_RANGE_LOW when ONLY the low-side byte is set -- whatever the order the blocks wrote in): the outputs of the run(s) since the last check are not valid
 *   (inf / NaN where the fp32 reference is finite) and must be recomputed on a plan built with FV_PAIR_F32 arithmetic --
 *   fastvocoder_amd/generator/engine.py does that automatically (NativeModule.range_guard).
 */
int fv_plan_set_guard(fv_plan_t* plan, int* word);
int fv_plan_check_range(fv_plan_t* plan, void* stream);

/*
 * Test / tuning hook, not product configuration: sets one of the launchers' tuning switches (csrc/fv_internal.h struct
 * Tuning: "convh_blocks", "pair_blocks", "sched", "sched_switch", "sum3_min", ...) for the whole process.  The launch
 * path reads no environment variable; a process started with FV_TUNING=1 reads FV_<KEY> once at its first launch.
 */
int fv_tuning_set(const char* key, int value);
/*
 * Test hook, host only (no device is touched): the block schedule a fused-pair / split-f16 conv launch of n_members (1 .. 3)
 * members with n_items[m] items of cost[m] each hands to its nblk persistent blocks (csrc/convh_launch.hip) -- the
 * tables the kernels index with blockIdx, as the CPU suite checks them (every item to exactly one block, shares in
 * order, the makespan bound).  The reference has no counterpart: it leaves the partition to ATen.
 *   mode 0: pair_schedule -- longest-processing-time-first over indivisible items for launches with few items per block
 *           (three_members != 0: three-member launches are scheduled too, as the 128-channel launcher asks);
 *   mode 1: pair_cut_schedule -- the contiguous cost-balanced cut, shares 0 .. nblk - 1 (nblk <= 512).
 * table: 512 words.  Returns what the kernel would see as sched_on: 1 -- two words per block, member m's items
 * [lo, lo + count) as lo (11 bits) | count (5 bits) << 11, word 0 = member 0 | member 1 << 16, word 1 = member 2;
 * 2 -- table[i] = first item of share i in the members' concatenated item sequence; 0 -- no table for this shape (the
 * kernel cuts by arithmetic); or a negative FV_ERR_* code.
 */
int fv_debug_pair_schedule(int n_members, const int* n_items, const int* cost, int nblk, int mode, int three_members,
                           unsigned* table);

/* ------------------------------------------------------------------ *
 * measurement hook (bench.py): per-launch timing of the dominant kernel
 * with HIP events recorded on the caller's stream
 * ------------------------------------------------------------------ */
/* When enabled, every conv kernel launch is followed by a hipEvent on its
 * stream (and the first one preceded by one); a launch's time runs from the
 * end of the launch before it to its own end.  fv_profile_collect()
 * synchronises and returns the accumulated (launches, milliseconds,
 * algorithmic flops, algorithmic bytes) of one kernel family and removes
 * those records.  kind: FV_KERNEL_* or -1 for all. */
#define FV_KERNEL_CONV_MFMA32 0 /* 32x32x2 fp32-MFMA implicit-GEMM conv (M > 16 rows) */
#define FV_KERNEL_CONV_MFMA16 1 /* 16x16x4 fp32-MFMA implicit-GEMM conv (M <= 16 rows) */
#define FV_KERNEL_CONV_NARROW 2 /* VALU conv for Cout <= 4 */
#define FV_KERNEL_PAIR16 3      /* fused ResBlock pairs / MRF stage end, C = 16 (16x16x4 fp32 MFMA) */
#define FV_KERNEL_PAIR32 4      /* fused ResBlock pairs, C = 32 */
#define FV_KERNEL_PAIRH16 5     /* fused ResBlock pairs, C = 16, split-f16 operands (16x16x32 f16 MFMA) */
#define FV_KERNEL_PAIRH32 6     /* ... C = 32 */
#define FV_KERNEL_CONVH64 7     /* conv1d with split-f16 operands, C = 64 (ResBlock pairs of the wide stages) */
#define FV_KERNEL_CONVH128 8    /* ... C = 128 */
#define FV_KERNEL_CONVT 9       /* transposed conv (kernel = 2 strides) with split-f16 operands (convt_kernel) */
#define FV_KERNEL_CONVG 10      /* two-source 1x1 conv (ResidualStack tail) with split-f16 operands (convg_kernel) */
#define FV_KERNEL_STACK 11      /* MelGAN ResidualStack as one launch, 32 / 64 / 128 channels, split-f16 operands (convk_kernel) */
#define FV_KERNEL_MRF16 12      /* a whole 16-channel MRF stage as one launch, split-f16 operands (mrfh_kernel) */
#define FV_KERNEL_MRF32 13      /* ... 32-channel (mrfw_kernel) */
int fv_profile_enable(int on);
/* what the measurement adds to a launch (subtract it per launch): a launch's duration is measured completion to
 * completion on its stream, from the end event of the launch before it to its own (its dispatch latency included, as
 * in rocprofv3's dispatch durations); the end-event record is one more packet -- a chain of n (null kernel, event)
 * pairs against a chain of n null kernels */
int fv_profile_bracket_cost(void* stream, int n, double* ms_per_bracket);
/* The device's ACHIEVABLE dense f16 matrix rate, for the `roofline` object of bench.py (fractions against the
 * nominal AND the measured peak): `launches` back-to-back launches of a kernel that does nothing but v_mfma_f32_16x16x32_f16 on
 * registers (two blocks of 8 waves per CU, eight independent accumulators per wave, `iters` x 8 MFMAs per wave, operands that
 * differ per lane), timed by events over the LAST half of the launches -- so that with enough of them the figure is the
 * sustained one.  scratch: >= 512 * 2 * (number of CUs) floats of device memory (one value per thread is stored so that the
 * loop cannot be dropped).  *tflops: executed f16 FLOP (16 384 per MFMA) / second / 1e12. */
int fv_profile_mfma_f16_rate(float* scratch, long long scratch_floats, int launches, int iters, void* stream, double* tflops);
int fv_profile_collect(int kind, int64_t* launches, double* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* FASTVOCODER_HIP_H */
