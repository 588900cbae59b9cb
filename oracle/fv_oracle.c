/*
 * fv_oracle.c -- CPU restatement of the arithmetic on FastVocoder's generator
 * forward path.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this.  The product path
 * (fastvocoder_amd/) never links, imports or falls back to it.
 *
 * Parity pin: the reference ships no tests/golden vectors for this path
 * (SURVEY.md section 4), so this restatement is pinned against outputs of the
 * reference itself, generated in the build container by
 * tests/golden/make_golden.py (imports /root/reference read-only) and committed
 * as tests/golden/*.npz.  tests/test_oracle_golden.py checks it.
 *
 * Arithmetic: inputs/outputs fp32 (the reference casts everything to
 * torch.float, model/generator/hifigan.py:111-112); accumulation here is in
 * double and rounded once per output element, so this oracle sits inside the
 * reference's own fp32 noise floor (6e-7..7e-6) rather
 * than adding its own.
 *
 * Layout everywhere: contiguous [B, C, T] (NCW), like the reference.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC fv_oracle.c -o libfv_oracle.so -lm
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define FVO_PAD_ZERO 0
#define FVO_PAD_REFLECT 1

/* torch.nn.ReflectionPad1d index map (model/generator/modules.py:355,
 * melgan.py:69): position -i reads i, position T-1+i reads T-1-i. */
static inline long reflect_index(long i, long T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

/* F.leaky_relu (modules.py:225,227; hifigan.py:95,104).  slope == 1 means
 * "no activation"; slope == 0 is ReLU (basis_melgan.py:120-121). */
static inline float act(float v, float slope) { return v >= 0.f ? v : v * slope; }

/*
 * y[b,co,t] = bias[co] + sum_{ci,j} w[co,ci,j] * act(x[b,ci, t + j*dil - pad])
 *
 * torch.nn.Conv1d as used at modules.py:193-221 (zero "same" padding,
 * get_padding modules.py:186-187), modules.py:353-358 / melgan.py:68-71 /
 * modules.py:82-88 (ReflectionPad1d + valid conv == reflect index map with
 * pad = (k-1)/2*dil), hifigan.py:26,52 (conv_pre / conv_post).
 * The pre-activation is the F.leaky_relu the reference applies to the conv's
 * input just before the call.  Tout = Tin + 2*pad - dil*(k-1).
 */
void fvo_conv1d(const float* x, const float* w, const float* bias, float* y,
                int B, int Cin, int Cout, int Tin, int k, int dil, int pad,
                int pad_mode, float pre_slope) {
    const long Tout = (long)Tin + 2L * pad - (long)dil * (k - 1);
    if (Tout <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float* yr = y + ((size_t)b * Cout + co) * Tout;
            for (long t = 0; t < Tout; ++t) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float* xr = x + ((size_t)b * Cin + ci) * Tin;
                    const float* wr = w + ((size_t)co * Cin + ci) * k;
                    for (int j = 0; j < k; ++j) {
                        long i = t + (long)j * dil - pad;
                        if (pad_mode == FVO_PAD_REFLECT) {
                            i = reflect_index(i, Tin);
                        } else if (i < 0 || i >= Tin) {
                            continue;
                        }
                        acc += (double)wr[j] * (double)act(xr[i], pre_slope);
                    }
                }
                yr[t] = (float)acc;
            }
        }
    }
}

/*
 * torch.nn.ConvTranspose1d (hifigan.py:39-44, multiband_hifigan.py:48-53,
 * melgan.py:77-85, basis_melgan.py:89-97); weight layout [Cin, Cout, k]:
 *   y[b,co, i*stride - pad + j] += act(x[b,ci,i]) * w[ci,co,j]
 * Tout = (Tin-1)*stride - 2*pad + k + out_pad.  Written in gather form.
 */
void fvo_conv_transpose1d(const float* x, const float* w, const float* bias,
                          float* y, int B, int Cin, int Cout, int Tin, int k,
                          int stride, int pad, int out_pad, float pre_slope) {
    const long Tout = ((long)Tin - 1) * stride - 2L * pad + k + out_pad;
    if (Tout <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float* yr = y + ((size_t)b * Cout + co) * Tout;
            for (long t = 0; t < Tout; ++t) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int j = 0; j < k; ++j) {
                    long num = t + pad - j;
                    if (num < 0 || num % stride != 0) continue;
                    long i = num / stride;
                    if (i >= Tin) continue;
                    for (int ci = 0; ci < Cin; ++ci) {
                        acc += (double)w[((size_t)ci * Cout + co) * k + j] *
                               (double)act(x[((size_t)b * Cin + ci) * Tin + i], pre_slope);
                    }
                }
                yr[t] = (float)acc;
            }
        }
    }
}

/*
 * torch.nn.utils.weight_norm fold (hifigan.py:58-76): w = v * g / ||v||, the
 * norm over every dim except 0 (for ConvTranspose1d dim 0 is the INPUT
 * channel).  v is [dim0, inner], g is [dim0].
 */
void fvo_weight_norm_fold(const float* v, const float* g, float* w, int dim0, long inner) {
    for (int r = 0; r < dim0; ++r) {
        double ss = 0.0;
        for (long i = 0; i < inner; ++i) ss += (double)v[r * inner + i] * (double)v[r * inner + i];
        /* torch computes the norm in fp32 and then v * (g / norm) */
        float nrm = (float)sqrt(ss);
        float scale = g[r] / nrm;
        for (long i = 0; i < inner; ++i) w[r * inner + i] = v[r * inner + i] * scale;
    }
}

/*
 * PQMF.synthesis (model/generator/pqmf.py:121-135):
 *   u = conv_transpose1d(x, subbands * updown, stride = subbands)  (zero-stuff, x subbands)
 *   y = conv1d(ConstantPad1d(taps/2)(u), synthesis_filter[1, subbands, taps+1])
 * i.e. y[b,n] = sum_k sum_j h[k][j] * u_k[n + j - taps/2], u_k[S*m] = S*x[b,k,m].
 * conv_transpose1d with kernel [S,S,S] and stride S gives length S*Tsub.
 */
void fvo_pqmf_synthesis(const float* x, const float* h, float* y, int B, int S,
                        int ntaps /* taps+1, e.g. 63 */, int Tsub) {
    const long T = (long)S * Tsub;
    const int half = (ntaps - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (long n = 0; n < T; ++n) {
            double acc = 0.0;
            for (int kb = 0; kb < S; ++kb) {
                const float* xr = x + ((size_t)b * S + kb) * Tsub;
                for (int j = 0; j < ntaps; ++j) {
                    long p = n + j - half;
                    if (p < 0 || p >= T || p % S != 0) continue;
                    acc += (double)h[kb * ntaps + j] * ((double)S * (double)xr[p / S]);
                }
            }
            y[(size_t)b * T + n] = (float)acc;
        }
    }
}

/*
 * PQMF.analysis (pqmf.py:108-119), kept only for the analysis->synthesis
 * known-answer test: x[b,k,m] = sum_j ha[k][j] * xin[b, S*m + j - taps/2].
 */
void fvo_pqmf_analysis(const float* xin, const float* ha, float* x, int B, int S,
                       int ntaps, int T) {
    const int half = (ntaps - 1) / 2;
    const long Tsub = ((long)T - S) / S + 1; /* conv1d(stride=S, kernel S) over length T */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int kb = 0; kb < S; ++kb) {
            for (long m = 0; m < Tsub; ++m) {
                double acc = 0.0;
                for (int j = 0; j < ntaps; ++j) {
                    long p = (long)S * m + j - half;
                    if (p < 0 || p >= T) continue;
                    acc += (double)ha[kb * ntaps + j] * (double)xin[(size_t)b * T + p];
                }
                x[((size_t)b * S + kb) * Tsub + m] = (float)acc;
            }
        }
    }
}

/*
 * BasisSignalLayer.forward (modules.py:264-267): frames = act(wt) @ W^T with
 * W [L, C] (F.linear, no bias) then overlap_and_add(frames, L/2)
 * (modules.py:34-73): out[hop*f + j] += frames[f, j].
 * Input here is the trunk output in its native [B, C, F] layout (the reference
 * transposes to [B, F, C] first, basis_melgan.py:205-206); pre_slope is the
 * trunk's final ReLU (slope 0) when the caller has not applied it yet.
 * Output length (F-1)*hop + L.
 */
void fvo_basis_ola(const float* wt, const float* W, float* y, int B, int C, int F,
                   int L, int hop, float pre_slope) {
    const long N = ((long)F - 1) * hop + L;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (long n = 0; n < N; ++n) {
            double acc = 0.0;
            /* every frame f with 0 <= n - hop*f < L contributes */
            long fhi = n / hop;
            for (long f = fhi; f >= 0 && n - hop * f < L; --f) {
                if (f >= F) continue;
                int j = (int)(n - hop * f);
                /* the reference rounds each frame sample to fp32 before the add */
                double fr = 0.0;
                for (int c = 0; c < C; ++c)
                    fr += (double)W[(size_t)j * C + c] * (double)act(wt[((size_t)b * C + c) * F + f], pre_slope);
                acc += (double)(float)fr;
            }
            y[(size_t)b * N + n] = (float)acc;
        }
    }
}

/* y = tanh(x) (hifigan.py:106), elementwise helpers used by oracle/generators.py */
void fvo_tanh(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = (float)tanh((double)x[i]);
}

int fvo_version(void) { return 1; }
