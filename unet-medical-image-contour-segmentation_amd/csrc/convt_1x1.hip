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
// thread = (output pixel, 4 consecutive output channels); weights read as W[i][o][r][s] fp32
template <typename T>
__global__ __launch_bounds__(256) void convt2x2_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y, int ldy,
                                                           int B, int h, int wd, int Cin, int Cout, int Ho, int Wo,
                                                           int pt, int pl) {
    const int64_t total = (int64_t)B * Ho * Wo * Cout;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int o = (int)(idx % Cout);
        int64_t p = idx / Cout;
        int ox = (int)(p % Wo);
        int oy = (int)((p / Wo) % Ho);
        int b = (int)(p / ((int64_t)Wo * Ho));
        int uy = oy - pt, ux = ox - pl;
        float acc = 0.f;
        if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * wd) {
            int hy = uy >> 1, r = uy & 1, hx = ux >> 1, s = ux & 1;
            const T* xp = x + ((int64_t)(b * h + hy) * wd + hx) * ldx;
            const float* wp = w + (int64_t)o * 4 + r * 2 + s;
            acc = bias[o];
            for (int i = 0; i < Cin; ++i) acc = fmaf(uh_to_f32(xp[i]), wp[(int64_t)i * Cout * 4], acc);
        }
        y[p * ldy + o] = uh_from_f32<T>(acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void convt2x2_dgrad_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                             T* __restrict__ dx, int lddx, int B, int h, int wd, int Cin,
                                                             int Cout, int Ho, int Wo, int pt, int pl) {
    const int64_t total = (int64_t)B * h * wd * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % Cin);
        int64_t p = idx / Cin;
        int hx = (int)(p % wd);
        int hy = (int)((p / wd) % h);
        int b = (int)(p / ((int64_t)wd * h));
        float acc = 0.f;
        const float* wp = w + (int64_t)i * Cout * 4;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int oy = 2 * hy + r + pt, ox = 2 * hx + s + pl;
                if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                const T* g = dy + ((int64_t)(b * Ho + oy) * Wo + ox) * lddy;
                for (int o = 0; o < Cout; ++o) acc = fmaf(uh_to_f32(g[o]), wp[o * 4 + r * 2 + s], acc);
            }
        dx[p * lddx + i] = uh_from_f32<T>(acc);
    }
}

// block = one (i-range, tap) over a pixel split; threads = output channels; partial slabs then reduce
template <typename T>
__global__ __launch_bounds__(256) void convt2x2_wgrad_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x,
                                                             int ldx, float* __restrict__ slabs, int B, int h, int wd,
                                                             int Cin, int Cout, int Ho, int Wo, int pt, int pl, int nsplit) {
    // grid: x = Cin*4 (i, r, s), y = split
    const int i = blockIdx.x >> 2, rs = blockIdx.x & 3, r = rs >> 1, s = rs & 1;
    const int split = blockIdx.y;
    const int64_t npix = (int64_t)B * h * wd;
    const int64_t p0 = npix * split / nsplit, p1 = npix * (split + 1) / nsplit;
    float* slab = slabs + (int64_t)split * ((int64_t)Cin * Cout * 4 + Cout);
    for (int o = threadIdx.x; o < Cout; o += 256) {
        float acc = 0.f, bsum = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            int hx = (int)(p % wd);
            int hy = (int)((p / wd) % h);
            int b = (int)(p / ((int64_t)wd * h));
            int oy = 2 * hy + r + pt, ox = 2 * hx + s + pl;
            if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
            float g = uh_to_f32(dy[((int64_t)(b * Ho + oy) * Wo + ox) * lddy + o]);
            acc = fmaf(uh_to_f32(x[p * ldx + i]), g, acc);
            bsum += g;
        }
        slab[((int64_t)i * Cout + o) * 4 + rs] = acc;
        if (i == 0) atomicAdd(&slab[(int64_t)Cin * Cout * 4 + o], bsum);   // 4 taps add into the bias slot
    }
}

__global__ void convt_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, float* __restrict__ dbias,
                                         int64_t nw, int Cout, int nsplit) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = nw + Cout;
    if (idx >= stride) return;
    float v = 0.f;
    for (int k = 0; k < nsplit; ++k) v += slabs[(int64_t)k * stride + idx];
    if (idx < nw) dw[idx] = v;
    else dbias[idx - nw] = v;
}

extern "C" int uh_convt2x2_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int h,
                               int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt,
                               uh_stream stream) {
    UH_REQUIRE(x && w && bias && y && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldy >= Cout,
               "uh_convt2x2_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t total = (int64_t)B * Ho * Wo * Cout;
    UH_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL(convt2x2_fwd_kernel<T>, dim3(ct_grid(total)), dim3(256), 0, st, (const T*)x, ldx, w, bias, (T*)y,
                           ldy, B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    });
    UH_CHECK_LAUNCH("convt2x2_fwd_kernel");
    return UH_OK;
}

extern "C" int uh_convt2x2_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, int B, int h, int w_,
                                 int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && w && dx && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && lddx >= Cin,
               "uh_convt2x2_dgrad: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t total = (int64_t)B * h * w_ * Cin;
    UH_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL(convt2x2_dgrad_kernel<T>, dim3(ct_grid(total)), dim3(256), 0, st, (const T*)dy, lddy, w, (T*)dx,
                           lddx, B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    });
    UH_CHECK_LAUNCH("convt2x2_dgrad_kernel");
    return UH_OK;
}

static int convt_nsplit(int B, int h, int w_, int Cin) {
    int64_t npix = (int64_t)B * h * w_;
    int want = (2048 + Cin * 4 - 1) / (Cin * 4);
    if (want < 1) want = 1;
    if (want > npix) want = (int)npix;
    if (want > 64) want = 64;
    return want;
}

extern "C" size_t uh_convt2x2_wgrad_ws_bytes(int B, int h, int w_, int Cin, int Cout) {
    return (size_t)convt_nsplit(B, h, w_, Cin) * ((size_t)Cin * Cout * 4 + Cout) * sizeof(float) + 16;
}

extern "C" int uh_convt2x2_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, float* dbias, void* ws,
                                 size_t ws_bytes, int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top,
                                 int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && x && dw && dbias && ws && B > 0 && h > 0 && w_ > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && ldx >= Cin,
               "uh_convt2x2_wgrad: bad args");
    int nsplit = convt_nsplit(B, h, w_, Cin);
    size_t need = (size_t)nsplit * ((size_t)Cin * Cout * 4 + Cout) * sizeof(float);
    if (ws_bytes < need) {
        uh_set_error("uh_convt2x2_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(ws, 0, need, st);   // bias slots are accumulated with atomics
    UH_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL(convt2x2_wgrad_kernel<T>, dim3(Cin * 4, nsplit), dim3(256), 0, st, (const T*)dy, lddy,
                           (const T*)x, ldx, (float*)ws, B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left, nsplit);
    });
    UH_CHECK_LAUNCH("convt2x2_wgrad_kernel");
    int64_t nw = (int64_t)Cin * Cout * 4;
    hipLaunchKernelGGL(convt_slab_reduce_kernel, dim3((unsigned)((nw + Cout + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, dw, dbias, nw, Cout, nsplit);
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
    for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / LPP; p < npix; p += (int64_t)gridDim.x * ppb) {
        float acc[MAXCLS];
#pragma unroll
        for (int k = 0; k < MAXCLS; ++k) acc[k] = 0.f;
        for (int c = sub * V; c < Cin; c += LPP * V) {
            float v[V];
            uh_load<T, V>(x + p * ldx + c, v);
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k)
                if (k < ncls) {
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[k] = fmaf(v[i], ws[k * Cin + c + i], acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < MAXCLS; ++k)
            if (k < ncls) {
                for (int o = LPP >> 1; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
                if (sub == 0) logits[p * ncls + k] = acc[k] + bias[k];
            }
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
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
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
            for (int64_t p = p0 + pl; p < p1; p += PL) {
                float xv[V];
                uh_load<T, V>(x + p * ldx + c, xv);
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    float g = dl[p * NC + k];
                    bs[k] += g;
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[k][i] = fmaf(g, xv[i], acc[k][i]);
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
    int64_t n = (npix + 4095) / 4096;
    if (n > 512) n = 512;
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
        if (uh_vec_ok<T>(x, ldx, Cin)) {
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
