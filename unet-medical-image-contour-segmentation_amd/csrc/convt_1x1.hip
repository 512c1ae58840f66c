// convt_1x1.hip -- nn.ConvTranspose2d(Cin, Cin/2, kernel_size=2, stride=2) + F.pad
// (unet_parts.py:73,85-88) and OutConv = nn.Conv2d(Cin, n_classes, 1) with bias (unet_parts.py:103).
//
// ConvTranspose k2 s2 never overlaps: output pixel (2h+r, 2w+s) depends on input pixel (h,w) only,
// through the [Cin x Cout] slice W[:, :, r, s]  (SURVEY.md A.3).
// OutConv has 1..4 output channels: pure HBM-bound streaming, reductions by wave shuffles.
#include "uh_vec.h"

static inline unsigned ct_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------ ConvTranspose 2x2
// Three GEMMs over pixels on a shared LDS-tiled SIMT kernel (64x64 tile, 16-deep K steps, 4x4 outputs per thread,
// fp32 accumulate); the operand / result addressing is supplied by small functors:
//   fwd   y[(p,rs)][o] = bias[o] + sum_i x[p][i] * W[i][o][rs]          M = pixels, N = 4*Cout, K = Cin
//   dgrad dx[p][i]     = sum_{rs,o} dy[(p,rs)][o] * W[i][o][rs]         M = pixels, N = Cin,    K = 4*Cout
//   wgrad dW[i][o][rs] = sum_p x[p][i] * dy[(p,rs)][o]                  M = Cin,    N = 4*Cout, K = pixels (split)
// (p,rs) = output pixel (2h+r+pad_top, 2w+s+pad_left) of input pixel p; out-of-range ones (negative pad = crop)
// contribute zeros.  Config 5 only; an MFMA version is the obvious next step.
struct CtGeom { int B, h, w, Ho, Wo, pt, pl, Cin, Cout; };

__device__ __forceinline__ int64_t ct_out_pixel(const CtGeom& g, int64_t p, int rs) {
    int hx = (int)(p % g.w);
    int hy = (int)((p / g.w) % g.h);
    int b = (int)(p / ((int64_t)g.w * g.h));
    int oy = 2 * hy + (rs >> 1) + g.pt, ox = 2 * hx + (rs & 1) + g.pl;
    if (oy < 0 || oy >= g.Ho || ox < 0 || ox >= g.Wo) return -1;
    return ((int64_t)b * g.Ho + oy) * g.Wo + ox;
}

template <typename T> struct CtFwd {
    const T* x; int ldx; const float* w; const float* bias; T* y; int ldy; CtGeom g;
    __device__ float A(int64_t m, int k) const { return uh_to_f32(x[m * ldx + k]); }
    __device__ float Bv(int k, int n) const { int rs = n / g.Cout, o = n - rs * g.Cout; return w[((int64_t)k * g.Cout + o) * 4 + rs]; }
    __device__ void C(int64_t m, int n, float v, int) const {
        int rs = n / g.Cout, o = n - rs * g.Cout;
        int64_t q = ct_out_pixel(g, m, rs);
        if (q >= 0) y[q * ldy + o] = uh_from_f32<T>(v + bias[o]);
    }
};
template <typename T> struct CtDgrad {
    const T* dy; int lddy; const float* w; T* dx; int lddx; CtGeom g;
    __device__ float A(int64_t m, int k) const {
        int rs = k / g.Cout, o = k - rs * g.Cout;
        int64_t q = ct_out_pixel(g, m, rs);
        return q >= 0 ? uh_to_f32(dy[q * lddy + o]) : 0.f;
    }
    __device__ float Bv(int k, int n) const { int rs = k / g.Cout, o = k - rs * g.Cout; return w[((int64_t)n * g.Cout + o) * 4 + rs]; }
    __device__ void C(int64_t m, int n, float v, int) const { dx[m * lddx + n] = uh_from_f32<T>(v); }
};
template <typename T> struct CtWgrad {
    const T* x; int ldx; const T* dy; int lddy; float* slabs; CtGeom g;
    // M index = input channel i, K index = pixel (offset by the split's first pixel in the kernel)
    __device__ float A(int64_t m, int64_t k) const { return uh_to_f32(x[k * ldx + m]); }
    __device__ float Bv(int64_t k, int n) const {
        int rs = n / g.Cout, o = n - rs * g.Cout;
        int64_t q = ct_out_pixel(g, k, rs);
        return q >= 0 ? uh_to_f32(dy[q * lddy + o]) : 0.f;
    }
    __device__ void C(int64_t m, int n, float v, int split) const {
        int rs = n / g.Cout, o = n - rs * g.Cout;
        slabs[(int64_t)split * ((int64_t)g.Cin * g.Cout * 4) + ((int64_t)m * g.Cout + o) * 4 + rs] = v;
    }
};

template <typename F>
__global__ __launch_bounds__(256) void ct_gemm_kernel(F f, int64_t M, int N, int64_t K, int nsplit) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int split = blockIdx.z;
    const int64_t k_begin = K * split / nsplit, k_end = K * (split + 1) / nsplit;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += 16) {
        // 64x16 elements of A and of B: 4 each per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int idx = threadIdx.x + e * 256;
            int kk = idx & 15, mm = idx >> 4;
            int64_t k = k0 + kk, m = m0 + mm;
            As[kk][mm] = (k < k_end && m < M) ? f.A(m, k) : 0.f;
            int n = n0 + mm;
            Bs[kk][mm] = (k < k_end && n < N) ? f.Bv(k, n) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + ty * 4 + i;
            int n = n0 + tx * 4 + j;
            if (m < M && n < N) f.C(m, n, acc[i][j], split);
        }
}

// zero fill (used when F.pad leaves a border around the up-sampled image)
template <typename T>
__global__ void ct_zero_kernel(T* y, int ldy, int64_t npix, int C) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * C) return;
    y[(idx / C) * ldy + (idx % C)] = uh_from_f32<T>(0.f);
}

// dbias partials: sum over the up-sampled region of dy[.., o]; grid = (64-channel slabs, pixel slices); one partial row
// per pixel slice (no atomics: the result must not depend on the schedule), summed in double by ct_dbias_finish_kernel
template <typename T>
__global__ __launch_bounds__(256) void ct_dbias_kernel(const T* __restrict__ dy, int lddy, CtGeom g, float* __restrict__ partials) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + cl;
    float acc = 0.f;
    const int y_lo = max(g.pt, 0), y_hi = min(g.pt + 2 * g.h, g.Ho), x_lo = max(g.pl, 0), x_hi = min(g.pl + 2 * g.w, g.Wo);
    const int rw = x_hi - x_lo, rh = y_hi - y_lo;
    const int64_t n = (int64_t)g.B * rh * rw;
    if (o < g.Cout && rw > 0 && rh > 0)
        for (int64_t p = (int64_t)blockIdx.y * 4 + sl; p < n; p += (int64_t)gridDim.y * 4) {
            int xx = (int)(p % rw) + x_lo;
            int yy = (int)((p / rw) % rh) + y_lo;
            int b = (int)(p / ((int64_t)rw * rh));
            acc += uh_to_f32(dy[(((int64_t)b * g.Ho + yy) * g.Wo + xx) * lddy + o]);
        }
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && o < g.Cout) partials[(int64_t)blockIdx.y * g.Cout + o] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}
__global__ void ct_dbias_finish_kernel(const float* __restrict__ partials, int nslice, int Cout, float* __restrict__ dbias) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Cout) return;
    double t = 0.0;
    for (int k = 0; k < nslice; ++k) t += (double)partials[(int64_t)k * Cout + o];
    dbias[o] = (float)t;
}

__global__ void convt_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int64_t nw, int nsplit) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nw) return;
    float v = 0.f;
    for (int k = 0; k < nsplit; ++k) v += slabs[(int64_t)k * nw + idx];
    dw[idx] = v;
}

static inline CtGeom ct_geom(int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pt, int pl) {
    CtGeom g; g.B = B; g.h = h; g.w = w_; g.Ho = Ho; g.Wo = Wo; g.pt = pt; g.pl = pl; g.Cin = Cin; g.Cout = Cout;
    return g;
}

extern "C" int uh_convt2x2_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int h,
                               int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt,
                               uh_stream stream) {
    UH_REQUIRE(x && w && bias && y && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldy >= Cout,
               "uh_convt2x2_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    CtGeom g = ct_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    const int64_t M = (int64_t)B * h * w_;
    const bool border = !(pad_top == 0 && pad_left == 0 && Ho == 2 * h && Wo == 2 * w_);
    UH_DISPATCH_DT(dt, T, {
        if (border) {
            int64_t tot = (int64_t)B * Ho * Wo * Cout;
            hipLaunchKernelGGL(ct_zero_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (T*)y, ldy,
                               (int64_t)B * Ho * Wo, Cout);
        }
        CtFwd<T> f{(const T*)x, ldx, w, bias, (T*)y, ldy, g};
        hipLaunchKernelGGL(ct_gemm_kernel<CtFwd<T>>, dim3((unsigned)((M + 63) / 64), (4 * Cout + 63) / 64, 1), dim3(256), 0, st,
                           f, M, 4 * Cout, (int64_t)Cin, 1);
    });
    UH_CHECK_LAUNCH("convt2x2_fwd");
    return UH_OK;
}

extern "C" int uh_convt2x2_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int B, int h, int w_,
                                 int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && w && dx && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && lddx >= Cin,
               "uh_convt2x2_dgrad: bad args");
    hipStream_t st = (hipStream_t)stream;
    CtGeom g = ct_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    const int64_t M = (int64_t)B * h * w_;
    UH_DISPATCH_DT(dt, T, {
        CtDgrad<T> f{(const T*)dy, lddy, w, (T*)dx, lddx, g};
        hipLaunchKernelGGL(ct_gemm_kernel<CtDgrad<T>>, dim3((unsigned)((M + 63) / 64), (Cin + 63) / 64, 1), dim3(256), 0, st, f,
                           M, Cin, (int64_t)4 * Cout, 1);
    });
    UH_CHECK_LAUNCH("convt2x2_dgrad");
    return UH_OK;
}

static int convt_nsplit(int B, int h, int w_, int Cin, int Cout) {
    int64_t npix = (int64_t)B * h * w_;
    int tiles = ((Cin + 63) / 64) * ((4 * Cout + 63) / 64);
    int want = (1024 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > (npix + 63) / 64) want = (int)((npix + 63) / 64);
    if (want > 128) want = 128;
    return want;
}

constexpr int CT_DBIAS_SLICES = 512;
extern "C" size_t uh_convt2x2_wgrad_ws_bytes(int B, int h, int w_, int Cin, int Cout) {
    return (size_t)convt_nsplit(B, h, w_, Cin, Cout) * ((size_t)Cin * Cout * 4) * sizeof(float) +
           (size_t)CT_DBIAS_SLICES * Cout * sizeof(float) + 16;
}

extern "C" int uh_convt2x2_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, float* dbias, void* ws,
                                 size_t ws_bytes, int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top,
                                 int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && x && dw && dbias && ws && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && ldx >= Cin,
               "uh_convt2x2_wgrad: bad args");
    int nsplit = convt_nsplit(B, h, w_, Cin, Cout);
    size_t need = (size_t)nsplit * ((size_t)Cin * Cout * 4) * sizeof(float) + (size_t)CT_DBIAS_SLICES * Cout * sizeof(float);
    if (ws_bytes < need) {
        uh_set_error("uh_convt2x2_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    CtGeom g = ct_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    const int64_t K = (int64_t)B * h * w_;
    UH_DISPATCH_DT(dt, T, {
        CtWgrad<T> f{(const T*)x, ldx, (const T*)dy, lddy, (float*)ws, g};
        hipLaunchKernelGGL(ct_gemm_kernel<CtWgrad<T>>, dim3((Cin + 63) / 64, (4 * Cout + 63) / 64, nsplit), dim3(256), 0, st, f,
                           (int64_t)Cin, 4 * Cout, K, nsplit);
        int slices = (int)((K * 4 + 1023) / 1024);
        if (slices > CT_DBIAS_SLICES) slices = CT_DBIAS_SLICES;
        if (slices < 1) slices = 1;
        float* partials = (float*)ws + (size_t)nsplit * ((size_t)Cin * Cout * 4);
        hipLaunchKernelGGL(ct_dbias_kernel<T>, dim3((Cout + 63) / 64, slices), dim3(256), 0, st, (const T*)dy, lddy, g, partials);
        hipLaunchKernelGGL(ct_dbias_finish_kernel, dim3((Cout + 255) / 256), dim3(256), 0, st, (const float*)partials, slices, Cout,
                           dbias);
    });
    UH_CHECK_LAUNCH("convt2x2_wgrad");
    int64_t nw = (int64_t)Cin * Cout * 4;
    hipLaunchKernelGGL(convt_slab_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)ws, dw, nw,
                       nsplit);
    UH_CHECK_LAUNCH("convt_slab_reduce_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ OutConv 1x1
constexpr int MAXCLS = 8;

// LPP lanes per pixel, each holding V channels; reduce across the LPP lanes with shuffles.
template <typename T, int V>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ logits,
                                                          int64_t npix, int Cin, int ncls, int LPP) {
    extern __shared__ float ws[];   // [ncls][Cin]
    for (int k = threadIdx.x; k < ncls * Cin; k += 256) ws[k] = w[k];
    __syncthreads();
    const int sub = threadIdx.x % LPP;
    const int ppb = 256 / LPP;
    constexpr int U = 4;            // pixels per lane per trip: U independent 16-byte loads in flight (HBM-bound kernel)
    const bool one_piece = Cin == LPP * V;      // the common case (64 channels): exactly one piece per lane and pixel
    for (int64_t p0 = (int64_t)blockIdx.x * (U * ppb) + threadIdx.x / LPP; p0 < npix; p0 += (int64_t)gridDim.x * (U * ppb)) {
        float acc[U][MAXCLS];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) acc[u][k] = 0.f;
        if (one_piece) {
            float v[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * ppb;
                if (p < npix) uh_load<T, V>(x + p * ldx + sub * V, v[u]);
                else {
#pragma unroll
                    for (int i = 0; i < V; ++i) v[u][i] = 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k)
                if (k < ncls) {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float wk = ws[k * Cin + sub * V + i];
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[u][k] = fmaf(v[u][i], wk, acc[u][k]);
                    }
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * ppb;
                if (p >= npix) continue;
                for (int c = sub * V; c < Cin; c += LPP * V) {
                    float v[V];
                    uh_load<T, V>(x + p * ldx + c, v);
#pragma unroll
                    for (int k = 0; k < MAXCLS; ++k)
                        if (k < ncls) {
#pragma unroll
                            for (int i = 0; i < V; ++i) acc[u][k] = fmaf(v[i], ws[k * Cin + c + i], acc[u][k]);
                        }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * ppb;
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k)
                if (k < ncls) {
                    for (int o = LPP >> 1; o > 0; o >>= 1) acc[u][k] += __shfl_xor(acc[u][k], o, 64);
                    if (sub == 0 && p < npix) logits[p * ncls + k] = acc[u][k] + bias[k];
                }
        }
    }
}

// The common shape (Cin == LPP * V: one 16-byte piece per lane and pixel; n_classes <= 4 as a template constant): this
// lane's filter taps live in registers (no LDS, no barrier), U pixels per trip, the class loop fully unrolled.  The generic
// kernel above keeps the class count dynamic and re-reads its taps from LDS inside the pixel loop -- 2 477 instructions of
// branches and waits that held it at 4 TB/s.
template <typename T, int V, int NC, int LPP>
__global__ __launch_bounds__(256) void conv1x1_fwd_nc_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ logits,
                                                             int64_t npix) {
    constexpr int Cin = LPP * V;
    constexpr int PPB = 256 / LPP;
    constexpr int U = 4;
    const int sub = threadIdx.x % LPP;
    float wr[NC][V], bs[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        bs[k] = bias[k];
#pragma unroll
        for (int i = 0; i < V; ++i) wr[k][i] = w[k * Cin + sub * V + i];
    }
    for (int64_t p0 = (int64_t)blockIdx.x * (U * PPB) + threadIdx.x / LPP; p0 < npix; p0 += (int64_t)gridDim.x * (U * PPB)) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
            const int64_t pc = p < npix ? p : npix - 1;              // clamped: the tail lanes load a valid pixel, store nothing
            uh_load<T, V>(x + pc * ldx + sub * V, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * PPB;
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

template <typename T, int V, int LPP>
static void c11_fwd_nc_launch(int ncls, const T* x, int ldx, const float* w, const float* bias, float* logits, int64_t npix,
                              hipStream_t st) {
    const int ppb4 = 4 * (256 / LPP);                      // pixels per workgroup trip
    const unsigned grid = ct_grid((npix + ppb4 - 1) / ppb4 * 256);
    switch (ncls) {
        case 1: hipLaunchKernelGGL((conv1x1_fwd_nc_kernel<T, V, 1, LPP>), dim3(grid), dim3(256), 0, st, x, ldx, w, bias, logits, npix); break;
        case 2: hipLaunchKernelGGL((conv1x1_fwd_nc_kernel<T, V, 2, LPP>), dim3(grid), dim3(256), 0, st, x, ldx, w, bias, logits, npix); break;
        case 3: hipLaunchKernelGGL((conv1x1_fwd_nc_kernel<T, V, 3, LPP>), dim3(grid), dim3(256), 0, st, x, ldx, w, bias, logits, npix); break;
        default: hipLaunchKernelGGL((conv1x1_fwd_nc_kernel<T, V, 4, LPP>), dim3(grid), dim3(256), 0, st, x, ldx, w, bias, logits, npix); break;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void conv1x1_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                            T* __restrict__ dx, int lddx, int64_t npix, int Cin, int ncls) {
    const int G = Cin / V;
    const int64_t total = npix * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = idx / G;
        int c = (int)(idx - p * G) * V;
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = 0.f;
        for (int k = 0; k < ncls; ++k) {
            float g = dl[p * ncls + k];
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = fmaf(g, w[k * Cin + c + i], o[i]);
        }
        uh_store<T, V>(dx + p * lddx + c, o);
    }
}

// thread = (pixel lane, channel group of V); block partials [nblk][ncls][Cin + 1] (last column = dbias)
template <typename T, int V, int NC>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float* __restrict__ dl, const T* __restrict__ x, int ldx,
                                                            float* __restrict__ partials, int64_t npix, int Cin) {
    extern __shared__ float red[];   // [PL][NC][GB*V + 1]
    const int G = Cin / V;
    const int GB = G < 256 ? G : 256;
    const int PL = 256 / GB;
    const int RW = GB * V + 1;
    const int gl = threadIdx.x % GB, pl = threadIdx.x / GB;
    // Workgroups interleave over chunks of U*PL pixels (see bn_relu_bwd_reduce_kernel): neighbouring workgroups stream
    // neighbouring addresses, and every lane keeps U independent 16-byte loads in flight.
    constexpr int U = 4;
    for (int gb = 0; gb < G; gb += GB) {
        const int c = (gb + gl) * V;
        float acc[NC][V], bs[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            bs[k] = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
        }
        const bool act = pl < PL && gb + gl < G;
        if (act)
            for (int64_t p = (int64_t)blockIdx.x * (U * PL) + pl; p < npix; p += (int64_t)gridDim.x * (U * PL)) {
                float xv[U][V], g[U][NC];
                // branch-free: the tail loads a clamped (valid) pixel and its gradient is replaced by 0 -- a branch
                // around each load made the compiler wait for it before issuing the next one
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t q = p + (int64_t)u * PL;
                    const int64_t qc = q < npix ? q : npix - 1;
                    uh_load<T, V>(x + qc * ldx + c, xv[u]);
#pragma unroll
                    for (int k = 0; k < NC; ++k) g[u][k] = dl[qc * NC + k];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = p + (int64_t)u * PL < npix;
#pragma unroll
                    for (int k = 0; k < NC; ++k) g[u][k] = in ? g[u][k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        bs[k] += g[u][k];
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[k][i] = fmaf(g[u][k], xv[u][i], acc[k][i]);
                    }
            }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
#pragma unroll
                for (int i = 0; i < V; ++i) red[(pl * NC + k) * RW + gl * V + i] = acc[k][i];
                if (gl == 0) red[(pl * NC + k) * RW + GB * V] = bs[k];
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < NC * RW; k += 256) {
            int cls = k / RW, cc = k - cls * RW;
            float v = 0.f;
            for (int q = 0; q < PL; ++q) v += red[(q * NC + cls) * RW + cc];
            float* row = partials + ((int64_t)blockIdx.x * NC + cls) * (Cin + 1);
            if (cc < GB * V) { if (gb * V + cc < Cin) row[gb * V + cc] = v; }
            else if (gb == 0) row[Cin] = v;
        }
    }
}

// one block per output element: tree reduction over the partial rows in double
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ partials, int nblk, int Cin, int ncls,
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
        int cls = idx / (Cin + 1), c = idx - cls * (Cin + 1);
        if (c < Cin) dw[cls * Cin + c] = (float)red[0];
        else dbias[cls] = (float)red[0];
    }
}

static int c11_nblk(int64_t npix) {
    int64_t n = (npix + 1023) / 1024;
    if (n > 2048) n = 2048;
    if (n < 1) n = 1;
    return (int)n;
}

extern "C" int uh_conv1x1_fwd(const void* x, int ldx, const float* w, const float* bias, float* logits, int64_t npix,
                              int Cin, int ncls, int dt, uh_stream stream) {
    UH_REQUIRE(x && w && bias && logits && npix > 0 && Cin > 0 && ncls > 0 && ncls <= MAXCLS && ldx >= Cin,
               "uh_conv1x1_fwd: bad args (ncls <= %d)", MAXCLS);
    hipStream_t st = (hipStream_t)stream;
    size_t sm = (size_t)ncls * Cin * sizeof(float);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(x, ldx, Cin) && (Cin == 8 * VEC || Cin == 16 * VEC) && ncls <= 4) {      // the UNet head: 64 channels
            if (Cin == 8 * VEC) c11_fwd_nc_launch<T, VEC, 8>(ncls, (const T*)x, ldx, w, bias, logits, npix, st);
            else c11_fwd_nc_launch<T, VEC, 16>(ncls, (const T*)x, ldx, w, bias, logits, npix, st);
        } else if (uh_vec_ok<T>(x, ldx, Cin)) {
            int G = Cin / VEC, LPP = 1;
            while (LPP * 2 <= G && LPP < 64) LPP *= 2;   // power of two lanes per pixel
            int ppb = 256 / LPP;
            hipLaunchKernelGGL((conv1x1_fwd_kernel<T, VEC>), dim3(ct_grid((npix + ppb - 1) / ppb * 256)), dim3(256), sm, st,
                               (const T*)x, ldx, w, bias, logits, npix, Cin, ncls, LPP);
        } else {
            hipLaunchKernelGGL((conv1x1_fwd_kernel<T, 1>), dim3(ct_grid(npix)), dim3(256), sm, st, (const T*)x, ldx, w, bias,
                               logits, npix, Cin, ncls, 1);
        }
    });
    UH_CHECK_LAUNCH("conv1x1_fwd_kernel");
    return UH_OK;
}

extern "C" int uh_conv1x1_dgrad(const float* dlogits, const float* w, void* dx, int lddx, int64_t npix, int Cin, int ncls,
                                int dt, uh_stream stream) {
    UH_REQUIRE(dlogits && w && dx && npix > 0 && Cin > 0 && ncls > 0 && lddx >= Cin, "uh_conv1x1_dgrad: bad args");
    hipStream_t st = (hipStream_t)stream;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(dx, lddx, Cin))
            hipLaunchKernelGGL((conv1x1_dgrad_kernel<T, VEC>), dim3(ct_grid(npix * (Cin / VEC))), dim3(256), 0, st, dlogits,
                               w, (T*)dx, lddx, npix, Cin, ncls);
        else
            hipLaunchKernelGGL((conv1x1_dgrad_kernel<T, 1>), dim3(ct_grid(npix * Cin)), dim3(256), 0, st, dlogits, w, (T*)dx,
                               lddx, npix, Cin, ncls);
    });
    UH_CHECK_LAUNCH("conv1x1_dgrad_kernel");
    return UH_OK;
}

extern "C" size_t uh_conv1x1_wgrad_ws_bytes(int64_t npix, int Cin, int ncls) {
    return (size_t)c11_nblk(npix) * ncls * (Cin + 1) * sizeof(float) + 16;
}

extern "C" int uh_conv1x1_wgrad(const float* dlogits, const void* x, int ldx, float* dw, float* dbias, void* ws,
                                size_t ws_bytes, int64_t npix, int Cin, int ncls, int dt, uh_stream stream) {
    UH_REQUIRE(dlogits && x && dw && dbias && ws && npix > 0 && Cin > 0 && ncls > 0 && ncls <= MAXCLS && ldx >= Cin,
               "uh_conv1x1_wgrad: bad args");
    int nblk = c11_nblk(npix);
    size_t need = (size_t)nblk * ncls * (Cin + 1) * sizeof(float);
    if (ws_bytes < need) {
        uh_set_error("uh_conv1x1_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
#define UH_C11_LAUNCH(T, V, NC)                                                                                      \
    do {                                                                                                             \
        int G = Cin / (V), GB = G < 256 ? G : 256, PL = 256 / GB;                                                    \
        size_t sm = (size_t)PL * (NC) * (GB * (V) + 1) * sizeof(float);                                              \
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<T, V, NC>), dim3(nblk), dim3(256), sm, st, dlogits, (const T*)x, ldx, \
                           (float*)ws, npix, Cin);                                                                   \
    } while (0)
#define UH_C11_NC(T, V)                                                                    \
    switch (ncls) {                                                                        \
        case 1: UH_C11_LAUNCH(T, V, 1); break; case 2: UH_C11_LAUNCH(T, V, 2); break;      \
        case 3: UH_C11_LAUNCH(T, V, 3); break; case 4: UH_C11_LAUNCH(T, V, 4); break;      \
        case 5: UH_C11_LAUNCH(T, V, 5); break; case 6: UH_C11_LAUNCH(T, V, 6); break;      \
        case 7: UH_C11_LAUNCH(T, V, 7); break; default: UH_C11_LAUNCH(T, V, 8); break;     \
    }
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(x, ldx, Cin) && ncls <= 4) { UH_C11_NC(T, VEC) }
        else { UH_C11_NC(T, 1) }
    });
    UH_CHECK_LAUNCH("conv1x1_wgrad_kernel");
    int n = ncls * (Cin + 1);
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(n), dim3(256), 0, st, (const float*)ws, nblk, Cin, ncls, dw, dbias);
    UH_CHECK_LAUNCH("conv1x1_wgrad_reduce_kernel");
    return UH_OK;
}
