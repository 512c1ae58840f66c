// bn_fused.hip -- BatchNorm + ReLU of a DoubleConv's LAST conv fused with what consumes it, so that the tensors between
// them never make a round trip through HBM (all kernels here are HBM-bound; what they save is whole passes):
//
//   * "pool tail"  (unet_parts.py:32 behind unet_parts.py:18-20; every encoder level): the activation is both the skip
//     connection and the input of nn.MaxPool2d(2).
//       forward : one kernel reads y, writes z = relu(bn(y)) AND its 2x2 max-pool      (the pool no longer re-reads z)
//       backward: dz = dskip + route(dpool) is never materialised -- the two BatchNorm backward passes rebuild it from
//                 dskip, dpool and the window's z (recomputed from y) as they go       (one write + two reads of dz less)
//   * "head tail"  (unet_parts.py:103 behind the last DoubleConv): the activation is only the input of the 1x1 OutConv.
//       forward : one kernel reads y and writes the fp32 logits; z is never written
//       backward: dz = dlogits . w (rank n_classes) and the OutConv weight gradient sum_p dlogits[p] z[p] are both formed
//                 inside the BatchNorm backward passes from y and dlogits              (z, dz never exist in HBM)
//
// Every intermediate is rounded exactly as the unfused kernels round what they store (uh_round_as<T>), and sums run in the
// same order per thread, so fused and unfused paths agree bit for bit on dy / z / pooled and to summation order on the
// per-channel sums.  Thread = (pixel or 2x2 window, 16-byte channel group), like bn.hip.
#include "uh_vec.h"

static inline unsigned bf_grid(int64_t total, int cap = 256 * 16) {
    int64_t g = (total + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ================================================================================================ pool tail
extern "C" int uh_bn_relu_pool_ok(int B, int H, int W, int C, int dt) {
    if (dt != UH_BF16 && dt != UH_F32) return 0;
    const int V = dt == UH_BF16 ? 8 : 4;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C % V) return 0;
    return (int64_t)B * H * W * (C / V) < ((int64_t)1 << 31);
}

// z = relu(y * scale + shift) and pooled = max over each 2x2 window of z (NaN wins, like torch)
template <typename T, int V, bool HOIST>
__global__ __launch_bounds__(256) void bn_relu_pool_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ z, int ldz,
                                                                 T* __restrict__ pooled, int ldp, int B, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, G = C / V;
    const int total = B * Ho * Wo * G;
    const int first = blockIdx.x * 256 + threadIdx.x;
    float sc[V], sh[V];
    if constexpr (HOIST) {
        const int c = (first % G) * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
    }
    for (int idx = first; idx < total; idx += gridDim.x * 256) {
        const int win = idx / G, c = (idx - win * G) * V;
        const int ox = win % Wo, t = win / Wo, oy = t % Ho, b = t / Ho;
        if constexpr (!HOIST) {
#pragma unroll
            for (int i = 0; i < V; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
        }
        const int64_t p00 = ((int64_t)(b * H + 2 * oy) * W + 2 * ox);
        float v[4][V];
        uh_load<T, V>(y + p00 * ldy + c, v[0]);
        uh_load<T, V>(y + (p00 + 1) * ldy + c, v[1]);
        uh_load<T, V>(y + (p00 + W) * ldy + c, v[2]);
        uh_load<T, V>(y + (p00 + W + 1) * ldy + c, v[3]);
        float m[V];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                v[j][i] = uh_round_as<T>(uh_relu(fmaf(v[j][i], sc[i], sh[i])));
                m[i] = j == 0 ? v[0][i] : uh_max_nan(m[i], v[j][i]);
            }
        uh_store<T, V>(z + p00 * ldz + c, v[0]);
        uh_store<T, V>(z + (p00 + 1) * ldz + c, v[1]);
        uh_store<T, V>(z + (p00 + W) * ldz + c, v[2]);
        uh_store<T, V>(z + (p00 + W + 1) * ldz + c, v[3]);
        uh_store<T, V>(pooled + (int64_t)win * ldp + c, m);
    }
}

extern "C" int uh_bn_relu_pool_apply(const void* y, int ldy, const float* scale, const float* shift, void* z, int ldz,
                                     void* pooled, int ldp, int B, int H, int W, int C, int dt, uh_stream stream) {
    UH_REQUIRE(y && scale && shift && z && pooled && ldy >= C && ldz >= C && ldp >= C, "uh_bn_relu_pool_apply: bad args");
    UH_REQUIRE(uh_bn_relu_pool_ok(B, H, W, C, dt), "uh_bn_relu_pool_apply: shape not covered (even H, W; C a multiple of 16 bytes)");
    hipStream_t st = (hipStream_t)stream;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(y, ldy, C) && uh_vec_ok<T>(z, ldz, C) && uh_vec_ok<T>(pooled, ldp, C),
                   "uh_bn_relu_pool_apply: tensors must be 16-byte aligned with 16-byte multiple pixel strides");
        const int G = C / VEC;
        const unsigned g = bf_grid((int64_t)B * (H / 2) * (W / 2) * G);
        if (((int64_t)g * 256) % G == 0)
            hipLaunchKernelGGL((bn_relu_pool_apply_kernel<T, VEC, true>), dim3(g), dim3(256), 0, st, (const T*)y, ldy, scale, shift,
                               (T*)z, ldz, (T*)pooled, ldp, B, H, W, C);
        else
            hipLaunchKernelGGL((bn_relu_pool_apply_kernel<T, VEC, false>), dim3(g), dim3(256), 0, st, (const T*)y, ldy, scale, shift,
                               (T*)z, ldz, (T*)pooled, ldp, B, H, W, C);
    });
    UH_CHECK_LAUNCH("bn_relu_pool_apply_kernel");
    return UH_OK;
}

// The gradient of z over one 2x2 window, as maxpool2_bwd_kernel (pool_up.hip) would have stored it:
//   dz[j] = round_T(dskip[j] + (j is the FIRST maximum of z[0..3] in (0,0),(0,1),(1,0),(1,1) order ? dpool : 0))
// with torch's rule for the maximum ((val > max) || isnan(val)); yv: the window's conv outputs, z recomputed from them.
// Returns in d[j][i]; pre[j][i] = y * scale + shift (the ReLU mask is pre > 0, as in bn.hip).
template <typename T, int V>
__device__ __forceinline__ void pool_window_dz(const float (&yv)[4][V], const float (&sc)[V], const float (&sh)[V],
                                               const float (&g)[V], float (&d)[4][V], float (&pre)[4][V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float zz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pre[j][i] = fmaf(yv[j][i], sc[i], sh[i]);
            zz[j] = uh_round_as<T>(uh_relu(pre[j][i]));
        }
        int arg = 0;
        float m = zz[0];
        if (zz[1] > m || zz[1] != zz[1]) { m = zz[1]; arg = 1; }
        if (zz[2] > m || zz[2] != zz[2]) { m = zz[2]; arg = 2; }
        if (zz[3] > m || zz[3] != zz[3]) { m = zz[3]; arg = 3; }
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j][i] = uh_round_as<T>(d[j][i] + (arg == j ? g[i] : 0.f));
    }
}

// backward pass 1 with the pool tail: partials [nblk][2][C] like bn_relu_bwd_reduce_kernel; a block trip covers 32 windows
// (= 128 pixels) x 8 channel groups.  dskip may be NULL (no skip gradient).
template <typename T, int V>
__global__ __launch_bounds__(256, 4) void bn_relu_pool_bwd_reduce_kernel(const T* __restrict__ dskip, int ldskip,
                                                                      const T* __restrict__ dpool, int lddp,
                                                                      const T* __restrict__ y, int ldy,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                      float* __restrict__ partials, int B, int H, int W, int C) {
    constexpr int GB = 8, PL = 256 / GB;
    __shared__ float red[4][2][GB * V];
    const int Ho = H >> 1, Wo = W >> 1, G = C / V;
    const int nwin = B * Ho * Wo;
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
        if constexpr (sizeof(T) == 2 && V == 8) {
            // bf16: the nine 16-byte pieces of a window stay packed and are unpacked two channels at a time (the float-array
            // form below keeps 104 values live: 167 registers, three waves per SIMD, 4.2 TB/s)
            float piv[V];
#pragma unroll
            for (int i = 0; i < V; ++i) piv[i] = uh_round_as<T>(mean[c + i]);
            for (int q = blockIdx.x * PL + pl; q < nwin; q += gridDim.x * PL) {
                const int ox = q % Wo, t = q / Wo, oy = t % Ho, b = t / Ho;
                const int64_t p00 = ((int64_t)(b * H + 2 * oy) * W + 2 * ox);
                const int64_t pp[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
                u32x4 ry[4], rd[4], rg;
#pragma unroll
                for (int j = 0; j < 4; ++j) ry[j] = *reinterpret_cast<const u32x4*>(y + pp[j] * ldy + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) rd[j] = dskip ? *reinterpret_cast<const u32x4*>(dskip + pp[j] * ldskip + c) : u32x4{0u, 0u, 0u, 0u};
                rg = *reinterpret_cast<const u32x4*>(dpool + (int64_t)q * lddp + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        const int i = 2 * e + hl;
                        auto half = [&](unsigned w) { return __uint_as_float(hl ? (w & 0xffff0000u) : (w << 16)); };
                        float yy[4], pre[4], zz[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            yy[j] = half(ry[j][e]);
                            pre[j] = fmaf(yy[j], sc[i], sh[i]);
                            zz[j] = uh_round_as<T>(uh_relu(pre[j]));
                        }
                        int arg = 0;
                        float m = zz[0];
                        if (zz[1] > m || zz[1] != zz[1]) { m = zz[1]; arg = 1; }
                        if (zz[2] > m || zz[2] != zz[2]) { m = zz[2]; arg = 2; }
                        if (zz[3] > m || zz[3] != zz[3]) { m = zz[3]; arg = 3; }
                        const float gv = half(rg[e]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dzj = uh_round_as<T>(half(rd[j][e]) + (arg == j ? gv : 0.f));
                            const float mm = pre[j] > 0.f ? dzj : 0.f;
                            s1[i] += mm;
                            s2[i] = fmaf(mm, yy[j] - piv[i], s2[i]);        // sum m (y - pivot); turned into sum m xhat below
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // sum m xhat = rstd * (sum m (y - pivot) - (mean - pivot) * sum m): mean / rstd are needed once, not per element
            // (16 registers less in the loop: a fourth wave per SIMD).  pivot = the workgroup-independent value mean rounded to
            // bf16 -- |y - pivot| stays of the order of the channel's spread, so nothing cancels in fp32.
#pragma unroll
            for (int i = 0; i < V; ++i) s2[i] = (s2[i] - (mean[c + i] - piv[i]) * s1[i]) * rstd[c + i];
        } else {
        for (int q = blockIdx.x * PL + pl; q < nwin; q += gridDim.x * PL) {
            const int ox = q % Wo, t = q / Wo, oy = t % Ho, b = t / Ho;
            const int64_t p00 = ((int64_t)(b * H + 2 * oy) * W + 2 * ox);
            const int64_t pp[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
            float yv[4][V], d[4][V], gp[V], pre[4][V];
#pragma unroll
            for (int j = 0; j < 4; ++j) uh_load<T, V>(y + pp[j] * ldy + c, yv[j]);
            if (dskip) {
#pragma unroll
                for (int j = 0; j < 4; ++j) uh_load<T, V>(dskip + pp[j] * ldskip + c, d[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < V; ++i) d[j][i] = 0.f;
            }
            uh_load<T, V>(dpool + (int64_t)q * lddp + c, gp);
            pool_window_dz<T, V>(yv, sc, sh, gp, d, pre);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float m = pre[j][i] > 0.f ? d[j][i] : 0.f;
                    s1[i] += m;
                    s2[i] += m * (yv[j][i] - mu[i]) * rs[i];
                }
        }
        }
    }
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

// backward pass 2 with the pool tail: dy = scale*(dz*[z>0]) + b*y + k per pixel of each window (coefficients as
// bn_relu_bwd_apply_kernel)
template <typename T, int V, bool HOIST>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply_kernel(const T* __restrict__ dskip, int ldskip,
                                                                     const T* __restrict__ dpool, int lddp,
                                                                     const T* __restrict__ y, int ldy,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                     T* __restrict__ dy, int lddy, int B, int H, int W, int C,
                                                                     float inv_n) {
    const int Ho = H >> 1, Wo = W >> 1, G = C / V;
    const int total = B * Ho * Wo * G;
    const int first = blockIdx.x * 256 + threadIdx.x;
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
    if constexpr (HOIST) coeffs((first % G) * V);
    for (int idx = first; idx < total; idx += gridDim.x * 256) {
        const int win = idx / G, c = (idx - win * G) * V;
        if constexpr (!HOIST) coeffs(c);
        const int ox = win % Wo, t = win / Wo, oy = t % Ho, b = t / Ho;
        const int64_t p00 = ((int64_t)(b * H + 2 * oy) * W + 2 * ox);
        const int64_t pp[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float yv[4][V], d[4][V], gp[V], pre[4][V];
#pragma unroll
        for (int j = 0; j < 4; ++j) uh_load<T, V>(y + pp[j] * ldy + c, yv[j]);
        if (dskip) {
#pragma unroll
            for (int j = 0; j < 4; ++j) uh_load<T, V>(dskip + pp[j] * ldskip + c, d[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) d[j][i] = 0.f;
        }
        uh_load<T, V>(dpool + (int64_t)win * lddp + c, gp);
        pool_window_dz<T, V>(yv, ca, cs, gp, d, pre);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float m = pre[j][i] > 0.f ? d[j][i] : 0.f;
                o[i] = fmaf(ca[i], m, fmaf(cb[i], yv[j][i], ck[i]));
            }
            uh_store<T, V>(dy + pp[j] * lddy + c, o);
        }
    }
}

extern "C" int uh_bn_relu_pool_bwd_reduce(const void* dskip, int ldskip, const void* dpool, int lddp, const void* y, int ldy,
                                          const float* scale, const float* shift, const float* mean, const float* rstd,
                                          float* partials, int B, int H, int W, int C, int dt, uh_stream stream) {
    UH_REQUIRE(dpool && y && scale && shift && mean && rstd && partials, "uh_bn_relu_pool_bwd_reduce: null pointer");
    UH_REQUIRE(uh_bn_relu_pool_ok(B, H, W, C, dt) && lddp >= C && ldy >= C && (!dskip || ldskip >= C),
               "uh_bn_relu_pool_bwd_reduce: shape not covered");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = uh_bn_bwd_nblk((int64_t)B * H * W, C);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(dpool, lddp, C) && uh_vec_ok<T>(y, ldy, C) && (!dskip || uh_vec_ok<T>(dskip, ldskip, C)),
                   "uh_bn_relu_pool_bwd_reduce: tensors must be 16-byte aligned with 16-byte multiple pixel strides");
        const int G = C / VEC;
        hipLaunchKernelGGL((bn_relu_pool_bwd_reduce_kernel<T, VEC>), dim3(nblk, (G + 7) / 8), dim3(256), 0, st, (const T*)dskip,
                           ldskip, (const T*)dpool, lddp, (const T*)y, ldy, scale, shift, mean, rstd, partials, B, H, W, C);
    });
    UH_CHECK_LAUNCH("bn_relu_pool_bwd_reduce_kernel");
    return UH_OK;
}

extern "C" int uh_bn_relu_pool_bwd_apply(const void* dskip, int ldskip, const void* dpool, int lddp, const void* y, int ldy,
                                         const float* scale, const float* shift, const float* mean, const float* rstd,
                                         const float* partials, int nblk, float* dgamma, float* dbeta, void* dy, int lddy,
                                         int B, int H, int W, int64_t n_total, int C, int dt, uh_stream stream) {
    UH_REQUIRE(dpool && y && scale && shift && mean && rstd && dgamma && dbeta && dy, "uh_bn_relu_pool_bwd_apply: null pointer");
    UH_REQUIRE(uh_bn_relu_pool_ok(B, H, W, C, dt) && nblk >= 0 && (nblk == 0 || partials) && lddp >= C && ldy >= C && lddy >= C &&
                   (!dskip || ldskip >= C), "uh_bn_relu_pool_bwd_apply: shape not covered");
    hipStream_t st = (hipStream_t)stream;
    if (nblk > 0) {
        int rc = uh_bn_bwd_finalize(partials, nblk, C, dgamma, dbeta, stream);
        if (rc != UH_OK) return rc;
    }
    const int64_t npix = (int64_t)B * H * W;
    const float inv_n = (float)(1.0 / (double)(n_total > 0 ? n_total : npix));
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(dpool, lddp, C) && uh_vec_ok<T>(y, ldy, C) && uh_vec_ok<T>(dy, lddy, C) &&
                       (!dskip || uh_vec_ok<T>(dskip, ldskip, C)),
                   "uh_bn_relu_pool_bwd_apply: tensors must be 16-byte aligned with 16-byte multiple pixel strides");
        const int G = C / VEC;
        const unsigned g = bf_grid(npix / 4 * G);
        if (((int64_t)g * 256) % G == 0)
            hipLaunchKernelGGL((bn_relu_pool_bwd_apply_kernel<T, VEC, true>), dim3(g), dim3(256), 0, st, (const T*)dskip, ldskip,
                               (const T*)dpool, lddp, (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma,
                               (const float*)dbeta, (T*)dy, lddy, B, H, W, C, inv_n);
        else
            hipLaunchKernelGGL((bn_relu_pool_bwd_apply_kernel<T, VEC, false>), dim3(g), dim3(256), 0, st, (const T*)dskip, ldskip,
                               (const T*)dpool, lddp, (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma,
                               (const float*)dbeta, (T*)dy, lddy, B, H, W, C, inv_n);
    });
    UH_CHECK_LAUNCH("bn_relu_pool_bwd_apply_kernel");
    return UH_OK;
}

// ================================================================================================ head tail
// C == LPP * V channels (LPP = 8 or 16 lanes per pixel), n_classes <= 4 as a template constant: this lane's OutConv taps
// live in registers, like conv1x1_fwd_nc_kernel (convt_1x1.hip).
extern "C" int uh_bn_relu_head_ok(int C, int ncls, int dt) {
    if (dt != UH_BF16 && dt != UH_F32) return 0;
    const int V = dt == UH_BF16 ? 8 : 4;
    return (C == 8 * V || C == 16 * V) && ncls >= 1 && ncls <= 4;
}

template <typename T, int V, int NC, int LPP>
__global__ __launch_bounds__(256) void bn_relu_head_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ logits,
                                                               int64_t npix) {
    constexpr int C = LPP * V, PPB = 256 / LPP, U = 4;
    const int sub = threadIdx.x % LPP;
    float wr[NC][V], bs[NC], sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { sc[i] = scale[sub * V + i]; sh[i] = shift[sub * V + i]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        bs[k] = bias[k];
#pragma unroll
        for (int i = 0; i < V; ++i) wr[k][i] = w[k * C + sub * V + i];
    }
    for (int64_t p0 = (int64_t)blockIdx.x * (U * PPB) + threadIdx.x / LPP; p0 < npix; p0 += (int64_t)gridDim.x * (U * PPB)) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
            const int64_t pc = p < npix ? p : npix - 1;              // clamped: the tail lanes load a valid pixel, store nothing
            uh_load<T, V>(y + pc * ldy + sub * V, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
#pragma unroll
            for (int i = 0; i < V; ++i) v[u][i] = uh_round_as<T>(uh_relu(fmaf(v[u][i], sc[i], sh[i])));
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < V; ++i) a = fmaf(v[u][i], wr[k][i], a);
#pragma unroll
                for (int o = LPP >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                if (sub == 0 && p < npix) logits[p * NC + k] = a + bs[k];
            }
        }
    }
}

// backward pass 1 with the head tail.  Per block: BatchNorm partials [2][C] and OutConv partials [NC][C + 1] (last column:
// the bias gradient), the latter in the layout conv1x1_wgrad_reduce_kernel (convt_1x1.hip) sums.
template <typename T, int V, int NC, int LPP>
__global__ __launch_bounds__(256) void bn_relu_head_bwd_reduce_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                      const T* __restrict__ y, int ldy,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                      float* __restrict__ partials, float* __restrict__ hpartials,
                                                                      int64_t npix) {
    constexpr int C = LPP * V, PL = 256 / LPP, U = 2, NR = 2 + NC;
    __shared__ float red[4][NR][C];
    __shared__ float redb[4][NC];
    const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const int c = sub * V;
    float sc[V], sh[V], mu[V], rs[V], wr[NC][V];
    float s1[V], s2[V], ha[NC][V], hb[NC];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        sc[i] = scale[c + i]; sh[i] = shift[c + i]; mu[i] = mean[c + i]; rs[i] = rstd[c + i];
        s1[i] = 0.f; s2[i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        hb[k] = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { wr[k][i] = w[k * C + c + i]; ha[k][i] = 0.f; }
    }
    for (int64_t p = (int64_t)blockIdx.x * (U * PL) + pl; p < npix; p += (int64_t)gridDim.x * (U * PL)) {
        float yv[U][V], g[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + (int64_t)u * PL;
            const int64_t qc = q < npix ? q : npix - 1;          // branch-free: clamped load, gradient replaced by 0
            uh_load<T, V>(y + qc * ldy + c, yv[u]);
#pragma unroll
            for (int k = 0; k < NC; ++k) g[u][k] = q < npix ? dl[qc * NC + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < NC; ++k) hb[k] += g[u][k];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float pre = fmaf(yv[u][i], sc[i], sh[i]);
                const float zr = uh_round_as<T>(uh_relu(pre));
                float d = 0.f;                                     // conv1x1_dgrad_kernel's order
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    d = fmaf(g[u][k], wr[k][i], d);
                    ha[k][i] = fmaf(g[u][k], zr, ha[k][i]);
                }
                d = uh_round_as<T>(d);
                const float m = pre > 0.f ? d : 0.f;
                s1[i] += m;
                s2[i] += m * (yv[u][i] - mu[i]) * rs[i];
            }
        }
    }
    // the pixel lanes of a wave (lane bits above the LPP channel-group bits) by shuffles, then the 4 waves through LDS
#pragma unroll
    for (int o = LPP; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            s1[i] += __shfl_xor(s1[i], o, 64);
            s2[i] += __shfl_xor(s2[i], o, 64);
#pragma unroll
            for (int k = 0; k < NC; ++k) ha[k][i] += __shfl_xor(ha[k][i], o, 64);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) hb[k] += __shfl_xor(hb[k], o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < LPP) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            red[wave][0][c + i] = s1[i];
            red[wave][1][c + i] = s2[i];
#pragma unroll
            for (int k = 0; k < NC; ++k) red[wave][2 + k][c + i] = ha[k][i];
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < NC; ++k) redb[wave][k] = hb[k];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NR * C; k += 256) {
        const int which = k / C, cc = k - which * C;
        const float v = (red[0][which][cc] + red[1][which][cc]) + (red[2][which][cc] + red[3][which][cc]);
        if (which < 2) partials[((int64_t)blockIdx.x * 2 + which) * C + cc] = v;
        else hpartials[((int64_t)blockIdx.x * NC + (which - 2)) * (C + 1) + cc] = v;
    }
    if (threadIdx.x < NC)
        hpartials[((int64_t)blockIdx.x * NC + threadIdx.x) * (C + 1) + C] =
            (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
}

// one block per OutConv gradient element: sum over the partial rows in double (conv1x1_wgrad_reduce_kernel's twin)
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ partials, int nblk, int Cin, int ncls,
                                                                float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ double red[256];
    const int idx = blockIdx.x;
    const int n = ncls * (Cin + 1);
    double v = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) v += (double)partials[(int64_t)b * n + idx];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int cls = idx / (Cin + 1), c = idx - cls * (Cin + 1);
        if (c < Cin) dw[cls * Cin + c] = (float)red[0];
        else dbias[cls] = (float)red[0];
    }
}

// backward pass 2 with the head tail
template <typename T, int V, int NC, int LPP>
__global__ __launch_bounds__(256) void bn_relu_head_bwd_apply_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                     const T* __restrict__ y, int ldy,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                     T* __restrict__ dy, int lddy, int64_t npix, float inv_n) {
    constexpr int C = LPP * V, PPB = 256 / LPP, U = 4;
    const int sub = threadIdx.x % LPP;
    const int c = sub * V;
    float ca[V], cs[V], cb[V], ck[V], wr[NC][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float sc = scale[c + i];
        ca[i] = sc;
        cs[i] = shift[c + i];
        cb[i] = -sc * rstd[c + i] * dgamma[c + i] * inv_n;
        ck[i] = -sc * dbeta[c + i] * inv_n - cb[i] * mean[c + i];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) wr[k][i] = w[k * C + c + i];
    for (int64_t p0 = (int64_t)blockIdx.x * (U * PPB) + threadIdx.x / LPP; p0 < npix; p0 += (int64_t)gridDim.x * (U * PPB)) {
        float yv[U][V], g[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
            const int64_t pc = p < npix ? p : npix - 1;
            uh_load<T, V>(y + pc * ldy + c, yv[u]);
#pragma unroll
            for (int k = 0; k < NC; ++k) g[u][k] = dl[pc * NC + k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < NC; ++k) d = fmaf(g[u][k], wr[k][i], d);
                d = uh_round_as<T>(d);
                const float m = (fmaf(yv[u][i], ca[i], cs[i]) > 0.f) ? d : 0.f;
                o[i] = fmaf(ca[i], m, fmaf(cb[i], yv[u][i], ck[i]));
            }
            if (p < npix) uh_store<T, V>(dy + p * lddy + c, o);
        }
    }
}

static inline int head_nblk(int64_t npix, int C) { return uh_bn_bwd_nblk(npix, C); }

#define UH_HEAD_SWITCH(LPPV, ...)                                                 \
    switch (ncls) {                                                               \
        case 1: { constexpr int NC = 1; constexpr int LPP = LPPV; __VA_ARGS__ } break;  \
        case 2: { constexpr int NC = 2; constexpr int LPP = LPPV; __VA_ARGS__ } break;  \
        case 3: { constexpr int NC = 3; constexpr int LPP = LPPV; __VA_ARGS__ } break;  \
        default: { constexpr int NC = 4; constexpr int LPP = LPPV; __VA_ARGS__ } break; \
    }
#define UH_HEAD_DISPATCH(V, ...)                          \
    do {                                                  \
        if (C == 8 * (V)) { UH_HEAD_SWITCH(8, __VA_ARGS__) } \
        else { UH_HEAD_SWITCH(16, __VA_ARGS__) }          \
    } while (0)

extern "C" int uh_bn_relu_head_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* head_w,
                                   const float* head_b, float* logits, int64_t npix, int C, int ncls, int dt, uh_stream stream) {
    UH_REQUIRE(y && scale && shift && head_w && head_b && logits && npix > 0 && ldy >= C, "uh_bn_relu_head_fwd: bad args");
    UH_REQUIRE(uh_bn_relu_head_ok(C, ncls, dt), "uh_bn_relu_head_fwd: shape not covered (C = 8 or 16 channel groups, n_classes <= 4)");
    hipStream_t st = (hipStream_t)stream;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(y, ldy, C), "uh_bn_relu_head_fwd: y must be 16-byte aligned with a 16-byte multiple pixel stride");
        UH_HEAD_DISPATCH(VEC, {
            const int ppb4 = 4 * (256 / LPP);
            const unsigned grid = bf_grid((npix + ppb4 - 1) / ppb4 * 256);
            hipLaunchKernelGGL((bn_relu_head_fwd_kernel<T, VEC, NC, LPP>), dim3(grid), dim3(256), 0, st, (const T*)y, ldy, scale,
                               shift, head_w, head_b, logits, npix);
        });
    });
    UH_CHECK_LAUNCH("bn_relu_head_fwd_kernel");
    return UH_OK;
}

extern "C" size_t uh_bn_relu_head_bwd_ws_bytes(int64_t npix, int C, int ncls) {
    return (size_t)head_nblk(npix, C) * ncls * (C + 1) * sizeof(float) + 16;
}

extern "C" int uh_bn_relu_head_bwd_reduce(const float* dlogits, const float* head_w, const void* y, int ldy, const float* scale,
                                          const float* shift, const float* mean, const float* rstd, float* partials,
                                          float* dhead_w, float* dhead_b, void* ws, size_t ws_bytes, int64_t npix, int C,
                                          int ncls, int dt, uh_stream stream) {
    UH_REQUIRE(dlogits && head_w && y && scale && shift && mean && rstd && partials && dhead_w && dhead_b && ws && npix > 0 &&
                   ldy >= C, "uh_bn_relu_head_bwd_reduce: bad args");
    UH_REQUIRE(uh_bn_relu_head_ok(C, ncls, dt), "uh_bn_relu_head_bwd_reduce: shape not covered");
    const int nblk = head_nblk(npix, C);
    const size_t need = (size_t)nblk * ncls * (C + 1) * sizeof(float);
    if (ws_bytes < need) {
        uh_set_error("uh_bn_relu_head_bwd_reduce: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(y, ldy, C), "uh_bn_relu_head_bwd_reduce: y must be 16-byte aligned with a 16-byte multiple pixel stride");
        UH_HEAD_DISPATCH(VEC, {
            hipLaunchKernelGGL((bn_relu_head_bwd_reduce_kernel<T, VEC, NC, LPP>), dim3(nblk), dim3(256), 0, st, dlogits, head_w,
                               (const T*)y, ldy, scale, shift, mean, rstd, partials, (float*)ws, npix);
        });
    });
    UH_CHECK_LAUNCH("bn_relu_head_bwd_reduce_kernel");
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(ncls * (C + 1)), dim3(256), 0, st, (const float*)ws, nblk, C, ncls, dhead_w,
                       dhead_b);
    UH_CHECK_LAUNCH("head_wgrad_reduce_kernel");
    return UH_OK;
}

extern "C" int uh_bn_relu_head_bwd_apply(const float* dlogits, const float* head_w, const void* y, int ldy, const float* scale,
                                         const float* shift, const float* mean, const float* rstd, const float* partials,
                                         int nblk, float* dgamma, float* dbeta, void* dy, int lddy, int64_t npix,
                                         int64_t n_total, int C, int ncls, int dt, uh_stream stream) {
    UH_REQUIRE(dlogits && head_w && y && scale && shift && mean && rstd && dgamma && dbeta && dy && npix > 0 && ldy >= C &&
                   lddy >= C && nblk >= 0 && (nblk == 0 || partials), "uh_bn_relu_head_bwd_apply: bad args");
    UH_REQUIRE(uh_bn_relu_head_ok(C, ncls, dt), "uh_bn_relu_head_bwd_apply: shape not covered");
    hipStream_t st = (hipStream_t)stream;
    if (nblk > 0) {
        int rc = uh_bn_bwd_finalize(partials, nblk, C, dgamma, dbeta, stream);
        if (rc != UH_OK) return rc;
    }
    const float inv_n = (float)(1.0 / (double)(n_total > 0 ? n_total : npix));
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE(uh_vec_ok<T>(y, ldy, C) && uh_vec_ok<T>(dy, lddy, C),
                   "uh_bn_relu_head_bwd_apply: tensors must be 16-byte aligned with 16-byte multiple pixel strides");
        UH_HEAD_DISPATCH(VEC, {
            const int ppb4 = 4 * (256 / LPP);
            const unsigned grid = bf_grid((npix + ppb4 - 1) / ppb4 * 256);
            hipLaunchKernelGGL((bn_relu_head_bwd_apply_kernel<T, VEC, NC, LPP>), dim3(grid), dim3(256), 0, st, dlogits, head_w,
                               (const T*)y, ldy, scale, shift, mean, rstd, (const float*)dgamma, (const float*)dbeta, (T*)dy, lddy,
                               npix, inv_n);
        });
    });
    UH_CHECK_LAUNCH("bn_relu_head_bwd_apply_kernel");
    return UH_OK;
}
