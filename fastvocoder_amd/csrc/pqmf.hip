// PQMF synthesis filter bank in polyphase form (reference:
// model/generator/pqmf.py:121-135).
//
// The reference zero-stuffs each of the S sub-bands by S (a one-hot
// ConvTranspose1d, scaled by S) and runs a dense (ntaps = 63)-tap FIR over the
// S channels: y[n] = sum_k sum_j h[k][j] * u_k[n + j - half], u_k[S*m] = S*x_k[m].
// Only taps with (n + j - half) % S == 0 see a non-zero sample, so per output
// sample there are at most ceil(ntaps/S) = 16 live taps per band:
//   j = j0 + S*i,  j0 = (half - n) mod S,  m = (n + j0 - half)/S + i.
// HBM-bound (~16 FLOP/B): one thread per output sample, coalesced store; the
// S*ntaps filter (pre-scaled by S) sits in LDS, and the sub-band reads of
// neighbouring lanes hit the same or adjacent words (L1-resident).
#include "fv_internal.h"

namespace fv {

// y2 / sub (optional): the bias-removal flows -- y2 = y - sub when y2 is given (y stays the plain synthesis),
// else y = y - sub; sub is [T] or, sub_batched, [B, T].
__global__ __launch_bounds__(256) void pqmf_synthesis_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ h,
                                                             float* __restrict__ y, float* __restrict__ y2,
                                                             const float* __restrict__ sub, int sub_batched,
                                                             int S, int ntaps, int Tsub) {
    extern __shared__ float hs[];  // [S][ntaps], scaled by S
    for (int i = threadIdx.x; i < S * ntaps; i += 256) hs[i] = h[i] * (float)S;
    __syncthreads();
    const int64_t T = (int64_t)S * Tsub;
    const int b = blockIdx.y;
    const int half = (ntaps - 1) / 2;
    for (int64_t n = blockIdx.x * 256LL + threadIdx.x; n < T; n += (int64_t)gridDim.x * 256) {
        int j0 = (int)((half - n) % S);
        if (j0 < 0) j0 += S;
        const int64_t mbase = (n + j0 - half) / S;  // exact: divisible by construction
        float acc = 0.f;
        for (int kb = 0; kb < S; ++kb) {
            const float* xr = x + ((size_t)b * S + kb) * (size_t)Tsub;
            const float* hr = hs + kb * ntaps;
            for (int j = j0, i = 0; j < ntaps; j += S, ++i) {
                const int64_t m = mbase + i;
                if (m >= 0 && m < Tsub) acc = fmaf(hr[j], xr[m], acc);
            }
        }
        if (sub) {
            const float d = acc - sub[(sub_batched ? (size_t)b * T : 0) + n];
            if (y2) {
                y[(size_t)b * T + n] = acc;
                y2[(size_t)b * T + n] = d;
            } else {
                y[(size_t)b * T + n] = d;
            }
        } else {
            y[(size_t)b * T + n] = acc;
        }
    }
}

// PQMF analysis (reference pqmf.py:108-119): a dense ntaps-tap FIR per band over the
// zero-padded full-band signal, then keep every S-th sample (the one-hot strided conv):
//   x_k[m] = sum_j ha[k][j] * xin[S*m + j - half],  m < (T - S)/S + 1.
// The reference computes all T samples of every band and discards (S-1)/S of them; here
// only the kept ones are formed.  One thread per (b, m): the S bands share the same
// ntaps input samples (read once into registers' worth of L1 traffic), filter in LDS.
__global__ __launch_bounds__(256) void pqmf_analysis_kernel(const float* __restrict__ xin,
                                                            const float* __restrict__ ha,
                                                            float* __restrict__ x, int S, int ntaps,
                                                            int64_t T, int64_t Tsub) {
    extern __shared__ float hs[];  // [S][ntaps]
    for (int i = threadIdx.x; i < S * ntaps; i += 256) hs[i] = ha[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int half = (ntaps - 1) / 2;
    const float* xr = xin + (size_t)b * T;
    for (int64_t m = blockIdx.x * 256LL + threadIdx.x; m < Tsub; m += (int64_t)gridDim.x * 256) {
        const int64_t p0 = (int64_t)S * m - half;
        for (int kb = 0; kb < S; ++kb) {
            const float* hr = hs + kb * ntaps;
            float acc = 0.f;
            for (int j = 0; j < ntaps; ++j) {
                const int64_t p = p0 + j;
                if (p >= 0 && p < T) acc = fmaf(hr[j], xr[p], acc);
            }
            x[((size_t)b * S + kb) * (size_t)Tsub + m] = acc;
        }
    }
}

int launch_pqmf_analysis(const float* xin, const float* ha, float* x, int B, int S, int ntaps, int64_t T,
                         hipStream_t s) {
    const int64_t Tsub = (T - S) / S + 1;
    if (B <= 0 || T < S) return 0;
    int64_t blocks = (Tsub + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pqmf_analysis_kernel, dim3((unsigned)blocks, B), dim3(256),
                       (size_t)S * ntaps * sizeof(float), s, xin, ha, x, S, ntaps, T, Tsub);
    FV_HIP(hipGetLastError());
    return 0;
}

int launch_pqmf(const float* x, const float* h, float* y, float* y2, const float* sub, int sub_batched, int B, int S,
                int ntaps, int Tsub, hipStream_t s) {
    if (B <= 0 || Tsub <= 0) return 0;
    const int64_t T = (int64_t)S * Tsub;
    int64_t blocks = (T + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pqmf_synthesis_kernel, dim3((unsigned)blocks, B), dim3(256),
                       (size_t)S * ntaps * sizeof(float), s, x, h, y, y2, sub, sub_batched, S, ntaps, Tsub);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
