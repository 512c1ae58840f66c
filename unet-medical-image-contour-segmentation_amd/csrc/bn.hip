// bn.hip -- nn.BatchNorm2d (+ fused nn.ReLU) of unet/unet_parts.py:16-17,19-20 for NHWC tensors.
//
// Forward statistics arrive as per-tile (sum, sum of squares) slabs written by the conv epilogue
// (conv3x3.hip), so the forward needs no extra pass over y for the statistics:
//   uh_bn_finalize     slabs -> mean / rstd / scale / shift (+ running stats, unbiased var, momentum)
//   uh_bn_relu_apply   z = max(y*scale + shift, 0)                              (HBM-bound, 16 B/lane)
// Backward (closed form):
//   uh_bn_relu_bwd_reduce   per-block partial sums of dz*[z>0] and dz*[z>0]*xhat  (wave/LDS reductions)
//   uh_bn_relu_bwd_apply    dgamma, dbeta, and dy = scale*(dzm - sum1/n - xhat*sum2/n)
#include "uh_vec.h"

// ------------------------------------------------------------------------------------ finalize
// Slab layout written by the conv epilogue: [nslab][2][C] = per-row (mean, M2) of the stored y, followed
// by [nslab] pixel counts (one row per workgroup of the MFMA kernels, per tile of the others).

// Rows with a zero pixel count are skipped (producers that use fewer rows than the buffer holds zero the counts of the rest).
// One-pass merge in double about a pivot (the first row's mean): N = sum n_i, A = sum n_i (m_i - p), S = sum M2_i,
// Q = sum n_i (m_i - p)^2  ->  mean = p + A / N,  M2 = S + Q - A^2 / N; then the affine coefficients, the running statistics and
// num_batches_tracked.
//
// This kernel is nothing but latency on the critical path of the forward pass (launch, one dependent round trip to rows that
// all eight XCDs wrote, a cross-wave reduction, a handful of stores): 18 launches per train step.  Round 3's form (1024-thread
// blocks of 16 channels, a scan of ALL the per-tile counts for the last live row, then a two-pass merge: five block-wide
// barriers over 16 waves) took 8.5-13 us; now: 256-thread blocks of 4 channels x 64 row lanes, the first 1024 rows fetched
// unconditionally in ONE batch (16 per thread: the MFMA conv kernels write one row per workgroup, at most 768, the recomputed
// stem at most 1024), ONE reduction phase (four sums at once, one barrier over four waves) -- and the rows behind the first
// 1024 are only walked when the pixel count the caller passed does not match what those rows hold (producers with one row
// per tile: the generic / >= 2 GiB kernels).
constexpr int BNF_CH = 4, BNF_RL = 64, BNF_MAXI = 16, BNF_FAST = BNF_RL * BNF_MAXI;      // 1024 rows in the first batch
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int nslab, int ldc, int C, double n,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float momentum, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                          long long* __restrict__ nbt, float* __restrict__ m2_o) {
    __shared__ double red[4][4][BNF_CH];          // [wave][sum][channel]
    const int cl = threadIdx.x & (BNF_CH - 1), sl = threadIdx.x / BNF_CH, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * BNF_CH + cl;
    const int cc = c < C ? c : C - 1;
    const float* cnt = stats + (int64_t)nslab * 2 * ldc;      // ldc: channels per stat row (>= C)
    // the first batch: clamped row index, no branch between the loads, the count tests applied to the VALUES
    float nfr[BNF_MAXI], mvr[BNF_MAXI], m2r[BNF_MAXI];
    const int R0 = nslab < BNF_FAST ? nslab : BNF_FAST;
    const float cnt0 = cnt[0], mean0 = stats[cc];
#pragma unroll
    for (int i = 0; i < BNF_MAXI; ++i) {
        const int f = sl + BNF_RL * i;
        const int fc = f < R0 ? f : R0 - 1;
        nfr[i] = cnt[fc];
        mvr[i] = stats[((int64_t)fc * 2 + 0) * ldc + cc];
        m2r[i] = stats[((int64_t)fc * 2 + 1) * ldc + cc];
        if (f >= R0) nfr[i] = 0.f;
    }
    const double piv = cnt0 > 0.f ? (double)mean0 : 0.0;      // (a dead first row may hold anything)
    double sN = 0.0, sA = 0.0, sS = 0.0, sQ = 0.0;
    auto take = [&](float nf_, float mv_, float m2_) {
        const bool live = nf_ > 0.f;               // rows with a zero count may hold anything
        const double nf = (double)nf_, d = (double)mv_ - piv;
        sN = live ? sN + nf : sN;
        sA = live ? fma(nf, d, sA) : sA;
        sS = live ? sS + (double)m2_ : sS;
        sQ = live ? fma(nf * d, d, sQ) : sQ;
    };
#pragma unroll
    for (int i = 0; i < BNF_MAXI; ++i) take(nfr[i], mvr[i], m2r[i]);
    // the 16 row lanes of a wave by shuffles (lane bits 2..5), the 4 waves through LDS -- sums in a fixed order
    auto block_sums = [&](double& N, double& A, double& S, double& Q) {
#pragma unroll
        for (int o = BNF_CH; o < 64; o <<= 1) {
            sN += __shfl_xor(sN, o, 64); sA += __shfl_xor(sA, o, 64); sS += __shfl_xor(sS, o, 64); sQ += __shfl_xor(sQ, o, 64);
        }
        if ((threadIdx.x & 63) < BNF_CH) { red[wave][0][cl] = sN; red[wave][1][cl] = sA; red[wave][2][cl] = sS; red[wave][3][cl] = sQ; }
        __syncthreads();
        N = (red[0][0][cl] + red[1][0][cl]) + (red[2][0][cl] + red[3][0][cl]);
        A = (red[0][1][cl] + red[1][1][cl]) + (red[2][1][cl] + red[3][1][cl]);
        S = (red[0][2][cl] + red[1][2][cl]) + (red[2][2][cl] + red[3][2][cl]);
        Q = (red[0][3][cl] + red[1][3][cl]) + (red[2][3][cl] + red[3][3][cl]);
    };
    double N, A, S, Q;
    block_sums(N, A, S, Q);
    if (nslab > BNF_FAST && !(n > 0.0 && N == n)) {           // (block-uniform: the counts do not depend on the channel)
        // live rows behind the first batch (one row per TILE): walk them all, batches of BNF_MAXI rows per thread
        __syncthreads();                                       // `red` is read above
        sN = sA = sS = sQ = 0.0;
        for (int base = BNF_FAST; base < nslab; base += BNF_FAST) {
#pragma unroll
            for (int i = 0; i < BNF_MAXI; ++i) {
                const int f = base + sl + BNF_RL * i;
                const int fc = f < nslab ? f : nslab - 1;
                nfr[i] = f < nslab ? cnt[fc] : 0.f;
                mvr[i] = stats[((int64_t)fc * 2 + 0) * ldc + cc];
                m2r[i] = stats[((int64_t)fc * 2 + 1) * ldc + cc];
            }
#pragma unroll
            for (int i = 0; i < BNF_MAXI; ++i) take(nfr[i], mvr[i], m2r[i]);
        }
        double N2, A2, S2, Q2;
        block_sums(N2, A2, S2, Q2);
        N += N2; A += A2; S += S2; Q += Q2;
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;            // unet_parts.py:16: BatchNorm2d bookkeeping
    if (sl != 0 || c >= C) return;
    const double mean = N > 0.0 ? piv + A / N : 0.0;
    double m2 = N > 0.0 ? S + Q - A * A / N : 0.0;
    if (m2 < 0.0) m2 = 0.0;
    if (n <= 0.0) n = N;                        // caller did not know the pixel count: the counted total
    double var = m2 / n;                        // biased
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma[c], bt = beta[c];
    float sc = g * rstd;
    scale[c] = sc;
    shift[c] = bt - (float)mean * sc;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    if (m2_o) m2_o[c] = (float)m2;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) {
        double unb = n > 1.0 ? m2 / (n - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

extern "C" int uh_bn_finalize(const float* stat_partials, int nslab, int C, int64_t n, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                              float* m2_out, uh_stream stream) {
    UH_REQUIRE(stat_partials && gamma && beta && scale && shift && mean && rstd, "uh_bn_finalize: null pointer");
    UH_REQUIRE(nslab > 0 && C > 0 && n >= 0, "uh_bn_finalize: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BNF_CH - 1) / BNF_CH), dim3(256), 0, st, stat_partials, nslab, C, C, (double)n, gamma,
                       beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd,
                       (long long*)num_batches_tracked, m2_out);
    UH_CHECK_LAUNCH("bn_finalize_kernel");
    return UH_OK;
}

// As uh_bn_finalize for the first C channels of stat rows that are ldc >= C channels wide (the conv of a small-width layer
// is computed -- and its statistics are laid out -- for the 64-aligned channel count, uh_conv3x3_fwd_narrow): every
// per-channel array here has C entries.
extern "C" int uh_bn_finalize_ld(const float* stat_partials, int nslab, int ldc, int C, int64_t n, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                                 float* m2_out, uh_stream stream) {
    UH_REQUIRE(stat_partials && gamma && beta && scale && shift && mean && rstd, "uh_bn_finalize_ld: null pointer");
    UH_REQUIRE(nslab > 0 && C > 0 && ldc >= C && n >= 0, "uh_bn_finalize_ld: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BNF_CH - 1) / BNF_CH), dim3(256), 0, st, stat_partials, nslab, ldc, C, (double)n, gamma,
                       beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd,
                       (long long*)num_batches_tracked, m2_out);
    UH_CHECK_LAUNCH("bn_finalize_kernel");
    return UH_OK;
}

__global__ void bn_eval_coeffs_kernel(const float* g, const float* b, const float* rm, const float* rv, float eps, int C,
                                      float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = g[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = b[c] - rm[c] * sc;
}

extern "C" int uh_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 uh_stream stream) {
    UH_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "uh_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    UH_CHECK_LAUNCH("bn_eval_coeffs_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ apply
// HOIST: (gridDim.x * 256) % (C / V) == 0, so a thread keeps the same channel group on every grid-stride step and
// its per-channel coefficients live in registers (otherwise they are re-read from L1 per element).
template <typename T, int V, bool HOIST>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ z, int ldz,
                                                            int64_t npix, int C) {
    const int G = C / V;
    const int64_t total = npix * G;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sc[V], sh[V];
    if constexpr (HOIST) {
        const int c = (int)(first % G) * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
    }
    if constexpr (HOIST) {
        // the grid stride is a multiple of G: this thread keeps its channel group and walks pixels p0, p0 + pstep, ... --
        // no 64-bit division per trip, and U branch-free (clamped) loads are in flight before the first is used
        constexpr int U = 4;
        const int64_t pstep = ((int64_t)gridDim.x * blockDim.x) / G;
        const int c = (int)(first % G) * V;
        // a thread's U pixels are neighbours (slot * U + u), so that the loads of one trip stay within a few KB
        for (int64_t slot = first / G; slot * U < npix; slot += pstep) {
            const int64_t p = slot * U;
            float v[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t q = p + u < npix ? p + u : npix - 1;
                uh_load<T, V>(y + q * ldy + c, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < V; ++i) v[u][i] = uh_relu(fmaf(v[u][i], sc[i], sh[i]));
                if (p + u < npix) uh_store<T, V>(z + (p + u) * ldz + c, v[u]);
            }
        }
        return;
    }
    for (int64_t idx = first; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = idx / G;
        int c = (int)(idx - p * G) * V;
        float v[V];
        uh_load<T, V>(y + p * ldy + c, v);
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = uh_relu(fmaf(v[i], scale[c + i], shift[c + i]));
        uh_store<T, V>(z + p * ldz + c, v);
    }
}

static inline unsigned grid_for(int64_t total, int threads = 256, int cap = 256 * 16) {
    int64_t g = (total + threads - 1) / threads;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

extern "C" int uh_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* z, int ldz,
                                int64_t npix, int C, int dt, uh_stream stream) {
    UH_REQUIRE(y && z && scale && shift && npix > 0 && C > 0 && ldy >= C && ldz >= C, "uh_bn_relu_apply: bad args");
    hipStream_t st = (hipStream_t)stream;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(y, ldy, C) && uh_vec_ok<T>(z, ldz, C)) {
            const unsigned g = grid_for(npix * (C / VEC));
            if (((int64_t)g * 256) % (C / VEC) == 0)
                hipLaunchKernelGGL((bn_relu_apply_kernel<T, VEC, true>), dim3(g), dim3(256), 0, st, (const T*)y, ldy, scale,
                                   shift, (T*)z, ldz, npix, C);
            else
                hipLaunchKernelGGL((bn_relu_apply_kernel<T, VEC, false>), dim3(g), dim3(256), 0, st, (const T*)y, ldy, scale,
                                   shift, (T*)z, ldz, npix, C);
        } else {
            hipLaunchKernelGGL((bn_relu_apply_kernel<T, 1, false>), dim3(grid_for(npix * C)), dim3(256), 0, st, (const T*)y,
                               ldy, scale, shift, (T*)z, ldz, npix, C);
        }
    });
    UH_CHECK_LAUNCH("bn_relu_apply_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ backward
// Thread = (pixel lane, channel group of V).  Block = an interleaved set of 128-pixel chunks (blockIdx.x) x a slab of
// up to 8 channel groups (blockIdx.y), so small feature maps with many channels still fill the chip.
extern "C" int uh_bn_bwd_nblk(int64_t npix, int C) {
    // >= ~2048 workgroups over (pixel chunks) x (64-channel slabs) when the map is large enough
    int64_t slabs = (C + 63) / 64;
    int64_t cap = 2048 / slabs;
    if (cap < 512) cap = 512;
    int64_t n = (npix + 127) / 128;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return (int)n;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y,
                                                                 int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 float* __restrict__ partials, int64_t npix, int C) {
    constexpr int GB = 8;                             // channel groups per block
    constexpr int PL = 256 / GB;                      // 32 pixel lanes
    // 2 KB of LDS only: this kernel is meant to run beside the backward-weights conv of the previous layer (side
    // stream), whose two workgroups per CU leave ~4 KB of the CU's 160 KB
    __shared__ float red[4][2][GB * V];
    const int G = C / V;
    const int g_in = threadIdx.x % GB, pl = threadIdx.x / GB;
    const int g = blockIdx.y * GB + g_in;
    const bool act = g < G;
    const int c = g * V;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    if (act) {
        float sc[V], sh[V], mu[V], rs[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; mu[i] = mean[c + i]; rs[i] = rstd[c + i]; }
        // Workgroups interleave over chunks of U*PL pixels (grid-stride): neighbouring workgroups stream neighbouring
        // addresses, so the concurrent streams spread over all HBM channels (one contiguous range per workgroup made
        // them march in lockstep at a 2^k stride).  U pixels per trip = 2U 16-byte loads in flight per lane.
        if constexpr (sizeof(T) == 2 && V == 8) {
            // bf16: the sixteen bytes of a (pixel, channel group) stay packed until they are used, so FOUR pixels of both
            // tensors (8 loads, 32 registers) are in flight per lane -- the float-array form below had 4 loads in flight at 106
            // registers (four waves per SIMD) and ran at 4.2 TB/s where the apply kernels reach 5.5.  Branch-free: a pixel past
            // the end is clamped and its dz forced to zero.
            constexpr int U4 = 4;
            for (int64_t p = (int64_t)blockIdx.x * (U4 * PL) + pl; p < npix; p += (int64_t)gridDim.x * (U4 * PL)) {
                u32x4 rd[U4], ry[U4];
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int64_t q = p + u * PL < npix ? p + u * PL : npix - 1;
                    rd[u] = *reinterpret_cast<const u32x4*>(dz + q * lddz + c);
                    ry[u] = *reinterpret_cast<const u32x4*>(y + q * ldy + c);
                }
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const bool live = p + u * PL < npix;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y0 = __uint_as_float(ry[u][e] << 16), y1 = __uint_as_float(ry[u][e] & 0xffff0000u);
                        const float d0 = live ? __uint_as_float(rd[u][e] << 16) : 0.f, d1 = live ? __uint_as_float(rd[u][e] & 0xffff0000u) : 0.f;
                        const float m0 = (fmaf(y0, sc[2 * e], sh[2 * e]) > 0.f) ? d0 : 0.f;
                        const float m1 = (fmaf(y1, sc[2 * e + 1], sh[2 * e + 1]) > 0.f) ? d1 : 0.f;
                        s1[2 * e] += m0;
                        s2[2 * e] += m0 * (y0 - mu[2 * e]) * rs[2 * e];
                        s1[2 * e + 1] += m1;
                        s2[2 * e + 1] += m1 * (y1 - mu[2 * e + 1]) * rs[2 * e + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one pixel at a time: unpacking all four at once costs two waves per SIMD
                }
            }
        } else {
        constexpr int U = 2;
        for (int64_t p = (int64_t)blockIdx.x * (U * PL) + pl; p < npix; p += (int64_t)gridDim.x * (U * PL)) {
            float d[U][V], yv[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p + u * PL < npix) {
                    uh_load<T, V>(dz + (p + u * PL) * lddz + c, d[u]);
                    uh_load<T, V>(y + (p + u * PL) * ldy + c, yv[u]);
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) { d[u][i] = 0.f; yv[u][i] = 0.f; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float m = (fmaf(yv[u][i], sc[i], sh[i]) > 0.f) ? d[u][i] : 0.f;
                    s1[i] += m;
                    s2[i] += m * (yv[u][i] - mu[i]) * rs[i];
                }
        }
        }
    }
    // the 8 pixel lanes of a wave (lane bits 3..5) by shuffles, then the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < GB) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            red[wave][0][g_in * V + i] = s1[i];
            red[wave][1][g_in * V + i] = s2[i];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * GB * V; k += 256) {
        const int which = k / (GB * V), cc = k - which * (GB * V);
        const int ch = blockIdx.y * GB * V + cc;
        if (ch < C)
            partials[((int64_t)blockIdx.x * 2 + which) * C + ch] =
                (red[0][which][cc] + red[1][which][cc]) + (red[2][which][cc] + red[3][which][cc]);
    }
}

extern "C" int uh_bn_relu_bwd_reduce(const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                     const float* shift, const float* mean, const float* rstd, float* partials,
                                     int64_t npix, int C, int dt, uh_stream stream) {
    UH_REQUIRE(dz && y && scale && shift && mean && rstd && partials, "uh_bn_relu_bwd_reduce: null pointer");
    UH_REQUIRE(npix > 0 && C > 0 && lddz >= C && ldy >= C, "uh_bn_relu_bwd_reduce: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    int nblk = uh_bn_bwd_nblk(npix, C);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(dz, lddz, C) && uh_vec_ok<T>(y, ldy, C)) {
            int G = C / VEC;
            hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<T, VEC>), dim3(nblk, (G + 7) / 8), dim3(256), 0, st, (const T*)dz,
                               lddz, (const T*)y, ldy, scale, shift, mean, rstd, partials, npix, C);
        } else {
            hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<T, 1>), dim3(nblk, (C + 7) / 8), dim3(256), 0, st, (const T*)dz, lddz,
                               (const T*)y, ldy, scale, shift, mean, rstd, partials, npix, C);
        }
    });
    UH_CHECK_LAUNCH("bn_relu_bwd_reduce_kernel");
    return UH_OK;
}

// 512 threads = 8 channels x 64 row lanes (C/8 blocks, every thread walks nblk/64 rows).  Two waves per SIMD at <= 72 VGPRs:
// this kernel sits on the critical path of the backward pass between the two BatchNorm passes, and with 1024-thread blocks
// (four waves per SIMD, 4 x 72 registers) it could not be placed on a CU while a backward-weights workgroup of the side
// stream held half of the register file -- it waited 50-340 us for that kernel to END (kernel trace, profiles/README.md).
constexpr int BWF_CH = 8, BWF_THREADS = 512;
__global__ __launch_bounds__(BWF_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // double: sum(dz) cancels heavily behind a BatchNorm
    __shared__ double red[2][BWF_THREADS / 64][BWF_CH];
    const int cl = threadIdx.x % BWF_CH, sl = threadIdx.x / BWF_CH, wave = threadIdx.x >> 6;
    constexpr int RL = BWF_THREADS / BWF_CH;          // 64 row lanes
    const int c = blockIdx.x * BWF_CH + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // batches of 8 rows: 16 independent loads in flight per thread, then the adds in row order
        int s = sl;
        for (; s + 7 * RL < nblk; s += 8 * RL) {
            float va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                va[i] = partials[((int64_t)(s + RL * i) * 2 + 0) * C + c];
                vb[i] = partials[((int64_t)(s + RL * i) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += (double)va[i]; b += (double)vb[i]; }
        }
        for (; s < nblk; s += RL) {
            a += (double)partials[((int64_t)s * 2 + 0) * C + c];
            b += (double)partials[((int64_t)s * 2 + 1) * C + c];
        }
    }
    // the eight row lanes of a wave (lane bits 3..5), then the 8 waves through LDS
#pragma unroll
    for (int o = BWF_CH; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) < BWF_CH) { red[0][wave][cl] = a; red[1][wave][cl] = b; }
    __syncthreads();
    if (threadIdx.x < BWF_CH && c < C) {
        double x = 0.0, y = 0.0;
#pragma unroll
        for (int k = 0; k < BWF_THREADS / 64; ++k) { x += red[0][k][cl]; y += red[1][k][cl]; }
        dbeta[c] = (float)x;
        dgamma[c] = (float)y;
    }
}

template <typename T, int V, bool HOIST>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y,
                                                                int ldy, const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, T* __restrict__ dy,
                                                                int lddy, int64_t npix, int C, float inv_n) {
    const int G = C / V;
    const int64_t total = npix * G;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // dy = [z>0]*a*dz + b*y + k  with a = scale, b = -scale*rstd*dgamma/n, k = -scale*dbeta/n - b*mean
    float ca[V], cs[V], cb[V], ck[V];
    auto coeffs = [&](int c) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float sc = scale[c + i];
            ca[i] = sc;
            cs[i] = shift[c + i];
            cb[i] = -sc * rstd[c + i] * dgamma[c + i] * inv_n;
            ck[i] = -sc * dbeta[c + i] * inv_n - cb[i] * mean[c + i];
        }
    };
    if constexpr (HOIST) coeffs((int)(first % G) * V);
    if constexpr (HOIST && sizeof(T) == 2 && V == 8) {
        // bf16, fixed channel group per thread: two pixels per trip, their four 16-byte pieces requested before the first is
        // used (two in flight per lane before: 5.0 TB/s on the 805 MB layer)
        const int c = (int)(first % G) * V;
        const int64_t pstep = ((int64_t)gridDim.x * blockDim.x) / G;
        for (int64_t p = first / G; p < npix; p += 2 * pstep) {
            const int64_t p1 = p + pstep < npix ? p + pstep : p;
            u32x4 rd[2], ry[2];
            rd[0] = *reinterpret_cast<const u32x4*>(dz + p * lddz + c);
            ry[0] = *reinterpret_cast<const u32x4*>(y + p * ldy + c);
            rd[1] = *reinterpret_cast<const u32x4*>(dz + p1 * lddz + c);
            ry[1] = *reinterpret_cast<const u32x4*>(y + p1 * ldy + c);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float o[V];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y0 = __uint_as_float(ry[u][e] << 16), y1 = __uint_as_float(ry[u][e] & 0xffff0000u);
                    const float d0 = __uint_as_float(rd[u][e] << 16), d1 = __uint_as_float(rd[u][e] & 0xffff0000u);
                    const float m0 = (fmaf(y0, ca[2 * e], cs[2 * e]) > 0.f) ? d0 : 0.f;
                    const float m1 = (fmaf(y1, ca[2 * e + 1], cs[2 * e + 1]) > 0.f) ? d1 : 0.f;
                    o[2 * e] = fmaf(ca[2 * e], m0, fmaf(cb[2 * e], y0, ck[2 * e]));
                    o[2 * e + 1] = fmaf(ca[2 * e + 1], m1, fmaf(cb[2 * e + 1], y1, ck[2 * e + 1]));
                }
                if (u == 0 || p + pstep < npix) uh_store<T, V>(dy + (u == 0 ? p : p1) * lddy + c, o);
            }
        }
        return;
    }
    for (int64_t idx = first; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = idx / G;
        int c = (int)(idx - p * G) * V;
        if constexpr (!HOIST) coeffs(c);
        float d[V], yv[V], o[V];
        uh_load<T, V>(dz + p * lddz + c, d);
        uh_load<T, V>(y + p * ldy + c, yv);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float m = (fmaf(yv[i], ca[i], cs[i]) > 0.f) ? d[i] : 0.f;
            o[i] = fmaf(ca[i], m, fmaf(cb[i], yv[i], ck[i]));
        }
        uh_store<T, V>(dy + p * lddy + c, o);
    }
}

extern "C" int uh_bn_bwd_finalize(const float* partials, int nblk, int C, float* dgamma, float* dbeta, uh_stream stream) {
    UH_REQUIRE(partials && dgamma && dbeta && nblk > 0 && C > 0, "uh_bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BWF_CH - 1) / BWF_CH), dim3(BWF_THREADS), 0, (hipStream_t)stream, partials, nblk, C,
                       dgamma, dbeta);
    UH_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    return UH_OK;
}

extern "C" int uh_bn_relu_bwd_apply(const void* dz, int lddz, const void* y, int ldy, const float* scale,
                                    const float* shift, const float* mean, const float* rstd, const float* partials,
                                    int nblk, float* dgamma, float* dbeta, void* dy, int lddy, int64_t npix,
                                    int64_t n_total, int C, int dt, uh_stream stream) {
    UH_REQUIRE(dz && y && scale && shift && mean && rstd && dgamma && dbeta && dy, "uh_bn_relu_bwd_apply: null pointer");
    UH_REQUIRE(npix > 0 && C > 0 && nblk >= 0 && (nblk == 0 || partials) && lddz >= C && ldy >= C && lddy >= C,
               "uh_bn_relu_bwd_apply: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (nblk > 0) {          // nblk == 0: dgamma / dbeta already hold the (possibly cross-rank) sums
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BWF_CH - 1) / BWF_CH), dim3(BWF_THREADS), 0, st, partials, nblk, C, dgamma, dbeta);
        UH_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    }
    float inv_n = (float)(1.0 / (double)(n_total > 0 ? n_total : npix));
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(dz, lddz, C) && uh_vec_ok<T>(y, ldy, C) && uh_vec_ok<T>(dy, lddy, C)) {
            const unsigned g = grid_for(npix * (C / VEC));
            if (((int64_t)g * 256) % (C / VEC) == 0)
                hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<T, VEC, true>), dim3(g), dim3(256), 0, st, (const T*)dz, lddz,
                                   (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma, (const float*)dbeta,
                                   (T*)dy, lddy, npix, C, inv_n);
            else
                hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<T, VEC, false>), dim3(g), dim3(256), 0, st, (const T*)dz, lddz,
                                   (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma, (const float*)dbeta,
                                   (T*)dy, lddy, npix, C, inv_n);
        } else {
            hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<T, 1, false>), dim3(grid_for(npix * C)), dim3(256), 0, st,
                               (const T*)dz, lddz, (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma,
                               (const float*)dbeta, (T*)dy, lddy, npix, C, inv_n);
        }
    });
    UH_CHECK_LAUNCH("bn_relu_bwd_apply_kernel");
    return UH_OK;
}
