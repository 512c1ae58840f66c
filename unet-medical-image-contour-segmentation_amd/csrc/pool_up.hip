// pool_up.hip -- nn.MaxPool2d(2) (unet_parts.py:32) and nn.Upsample(scale_factor=2, 'bilinear',
// align_corners=True) + F.pad (unet_parts.py:70,85-88) for NHWC tensors; all HBM-bound, one 16-byte
// channel vector per lane.
#include "uh_vec.h"

static inline unsigned pu_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------ max-pool
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                           int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, G = C / V;
    const int64_t total = (int64_t)B * Ho * Wo * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % G) * V;
        int64_t p = idx / G;
        int ox = (int)(p % Wo);
        int oy = (int)((p / Wo) % Ho);
        int b = (int)(p / ((int64_t)Wo * Ho));
        const T* base = x + ((int64_t)(b * H + 2 * oy) * W + 2 * ox) * ldx + c;
        float a[V], t[V];
        uh_load<T, V>(base, a);
        uh_load<T, V>(base + ldx, t);
#pragma unroll
        for (int i = 0; i < V; ++i) a[i] = uh_max_nan(a[i], t[i]);
        uh_load<T, V>(base + (int64_t)W * ldx, t);
#pragma unroll
        for (int i = 0; i < V; ++i) a[i] = uh_max_nan(a[i], t[i]);
        uh_load<T, V>(base + (int64_t)W * ldx + ldx, t);
#pragma unroll
        for (int i = 0; i < V; ++i) a[i] = uh_max_nan(a[i], t[i]);
        uh_store<T, V>(y + p * ldy + c, a);
    }
}

// dx[h,w] = dskip[h,w] + (x[h,w] is the FIRST maximum of its window in (0,0),(0,1),(1,0),(1,1) order ? dy : 0)
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy,
                                                           int lddy, const T* __restrict__ dskip, int ldskip,
                                                           T* __restrict__ dx, int lddx, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, G = C / V;
    const int64_t total = (int64_t)B * H * W * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % G) * V;
        int64_t p = idx / G;
        int w = (int)(p % W);
        int h = (int)((p / W) % H);
        int b = (int)(p / ((int64_t)W * H));
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = 0.f;
        if (dskip) uh_load<T, V>(dskip + p * ldskip + c, o);
        const int oy = h >> 1, ox = w >> 1;
        if (oy < Ho && ox < Wo) {
            const T* base = x + ((int64_t)(b * H + 2 * oy) * W + 2 * ox) * ldx + c;
            float v0[V], v1[V], v2[V], v3[V], g[V];
            uh_load<T, V>(base, v0);
            uh_load<T, V>(base + ldx, v1);
            uh_load<T, V>(base + (int64_t)W * ldx, v2);
            uh_load<T, V>(base + (int64_t)W * ldx + ldx, v3);
            uh_load<T, V>(dy + ((int64_t)(b * Ho + oy) * Wo + ox) * lddy + c, g);
            const int me = (h & 1) * 2 + (w & 1);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                int arg = 0;
                float m = v0[i];
                // torch's rule: (val > max) || isnan(val) -- a NaN takes the window, the last NaN wins
                if (v1[i] > m || v1[i] != v1[i]) { m = v1[i]; arg = 1; }
                if (v2[i] > m || v2[i] != v2[i]) { m = v2[i]; arg = 2; }
                if (v3[i] > m || v3[i] != v3[i]) { m = v3[i]; arg = 3; }
                if (arg == me) o[i] += g[i];
            }
        }
        uh_store<T, V>(dx + p * lddx + c, o);
    }
}

extern "C" int uh_maxpool2_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int dt,
                               uh_stream stream) {
    UH_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && C > 0 && ldx >= C && ldy >= C, "uh_maxpool2_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t npo = (int64_t)B * (H / 2) * (W / 2);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(x, ldx, C) && uh_vec_ok<T>(y, ldy, C))
            hipLaunchKernelGGL((maxpool2_fwd_kernel<T, VEC>), dim3(pu_grid(npo * (C / VEC))), dim3(256), 0, st, (const T*)x,
                               ldx, (T*)y, ldy, B, H, W, C);
        else
            hipLaunchKernelGGL((maxpool2_fwd_kernel<T, 1>), dim3(pu_grid(npo * C)), dim3(256), 0, st, (const T*)x, ldx,
                               (T*)y, ldy, B, H, W, C);
    });
    UH_CHECK_LAUNCH("maxpool2_fwd_kernel");
    return UH_OK;
}

extern "C" int uh_maxpool2_bwd(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int ldskip, void* dx,
                               int lddx, int B, int H, int W, int C, int dt, uh_stream stream) {
    UH_REQUIRE(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && ldx >= C && lddy >= C && lddx >= C,
               "uh_maxpool2_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t np = (int64_t)B * H * W;
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(x, ldx, C) && uh_vec_ok<T>(dy, lddy, C) && uh_vec_ok<T>(dx, lddx, C) &&
            (!dskip || uh_vec_ok<T>(dskip, ldskip, C)))
            hipLaunchKernelGGL((maxpool2_bwd_kernel<T, VEC>), dim3(pu_grid(np * (C / VEC))), dim3(256), 0, st, (const T*)x,
                               ldx, (const T*)dy, lddy, (const T*)dskip, ldskip, (T*)dx, lddx, B, H, W, C);
        else
            hipLaunchKernelGGL((maxpool2_bwd_kernel<T, 1>), dim3(pu_grid(np * C)), dim3(256), 0, st, (const T*)x, ldx,
                               (const T*)dy, lddy, (const T*)dskip, ldskip, (T*)dx, lddx, B, H, W, C);
    });
    UH_CHECK_LAUNCH("maxpool2_bwd_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ bilinear x2
// PyTorch's align_corners=True rule: scale = (in-1)/(out-1) in float; src = scale*dst; i0 = (int)src;
// i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1.
struct UpCoord { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpCoord up_coord(int dst, float scale, int in) {
    float src = scale * (float)dst;
    int i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    UpCoord u;
    u.i0 = i0;
    u.i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    u.l1 = src - (float)i0;
    u.l0 = 1.f - u.l1;
    return u;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                             int B, int h, int w, int C, int Ho, int Wo, int pt, int pl,
                                                             float sy, float sx) {
    const int G = C / V;
    const int64_t total = (int64_t)B * Ho * Wo * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % G) * V;
        int64_t p = idx / G;
        int ox = (int)(p % Wo);
        int oy = (int)((p / Wo) % Ho);
        int b = (int)(p / ((int64_t)Wo * Ho));
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = 0.f;
        int uy = oy - pt, ux = ox - pl;
        if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w) {
            UpCoord cy = up_coord(uy, sy, h), cx = up_coord(ux, sx, w);
            const T* r0 = x + ((int64_t)(b * h + cy.i0) * w) * ldx + c;
            const T* r1 = x + ((int64_t)(b * h + cy.i1) * w) * ldx + c;
            float v00[V], v01[V], v10[V], v11[V];
            uh_load<T, V>(r0 + (int64_t)cx.i0 * ldx, v00);
            uh_load<T, V>(r0 + (int64_t)cx.i1 * ldx, v01);
            uh_load<T, V>(r1 + (int64_t)cx.i0 * ldx, v10);
            uh_load<T, V>(r1 + (int64_t)cx.i1 * ldx, v11);
#pragma unroll
            for (int i = 0; i < V; ++i)
                o[i] = cy.l0 * (cx.l0 * v00[i] + cx.l1 * v01[i]) + cy.l1 * (cx.l0 * v10[i] + cx.l1 * v11[i]);
        }
        uh_store<T, V>(y + p * ldy + c, o);
    }
}

// Same arithmetic, far fewer instructions per byte written (the grid-stride form above spends ~200 VALU instructions --
// 64-bit div / mod, coordinates, four 8-value unpacks -- on every 16 bytes and runs at 2.9 TB/s: instruction-bound).
// Here a workgroup row (blockIdx.x) is UP_ROWS consecutive output rows of one image: the row decode and the vertical
// coordinates are scalar, a thread = (output column, 16-byte channel group) loads the input rows those output
// rows touch (x 2 columns; at most UP_ROWS / 2 + 2), interpolates them horizontally once, and blends each output row from two of
// the horizontal results -- which two is uniform over the workgroup, so it is a scalar branch, not a per-lane select.
// PRE: x is the RAW output of a conv and the activation that is up-sampled, max(x * scale + shift, 0) rounded to T exactly as
// uh_bn_relu_apply would have stored it, is formed on the way in (uh_bn_relu_upsample2x_fwd: that activation has no other reader).
// Output rows per thread.  The kernel is VALU-bound: SQ counters over the four Up-block shapes of config 2 (round 4, scratch/r4_up_pmc.sh)
// give 748 VALU instructions per thread (four 16-byte stores) = 47.9 k per SIMD on the 512 x 512 x 64 output, x 4 cycles = the whole
// 96 us launch.  Unpacking bf16, BatchNorm + ReLU + rounding of every LOADED value (the PRE form) and the two interpolation stages
// are what the arithmetic is; ~130 of the 748 were register copies / selects of the row picks and the zero-padding tests -- gone since
// the vertical blend reads the two rows where they are (one scalar branch per (a, bb) pair) and padding columns carry zeros through
// it: 188.6 -> 164.3 us over the four Up-block shapes of config 2, bit-identical.  Eight
// rows per thread (six input rows instead of two groups of four: a quarter fewer loads) measured no different (188.3 vs 188.6 us
// over the four shapes): the longer pick chains cost what the loads save -- and after the diet above they lose (161.5 -> 181 us,
// -DUH_UP_ROWS=8: the 48 row registers cost occupancy).
#ifndef UH_UP_ROWS
#define UH_UP_ROWS 4
#endif
constexpr int UP_ROWS = UH_UP_ROWS;
template <typename T, int V, bool PRE = false>
__global__ __launch_bounds__(256) void upsample2x_fwd_rows_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                                  int h, int w, int C, int Ho, int Wo, int pt, int pl,
                                                                  float sy, float sx, int gshift,
                                                                  const float* __restrict__ pre_scale = nullptr,
                                                                  const float* __restrict__ pre_shift = nullptr) {
    constexpr int R = UP_ROWS;
    const int G = C / V;
    const int groups = (Ho + R - 1) / R;
    const int b = blockIdx.x / groups, oy0 = (blockIdx.x - b * groups) * R;
    const int t = blockIdx.y * 256 + threadIdx.x;
    if (t >= Wo * G) return;
    const int ox = gshift >= 0 ? (t >> gshift) : (t / G);
    const int c = (t - ox * G) * V;
    const int ux = ox - pl;
    const bool col_in = ux >= 0 && ux < 2 * w;
    // vertical coordinates of the R output rows (scalar): rows outside the up-sampled image are zero padding (F.pad)
    UpCoord cy[R];
    bool row_in[R];
    int base = h - 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int uy = oy0 + r - pt;
        row_in[r] = (oy0 + r < Ho) && uy >= 0 && uy < 2 * h;
        cy[r] = up_coord(row_in[r] ? uy : 0, sy, h);
        if (row_in[r] && cy[r].i0 < base) base = cy[r].i0;
    }
    // R output rows advance the source row by < (R - 1) / 2, so they touch input rows base .. base + R / 2 + 1 at most
    constexpr int NR = R / 2 + 2;
    float hr[NR][V];                                   // horizontally interpolated input rows base .. base + NR - 1
    // (columns in the zero padding keep zeros here: the vertical stage then needs no per-lane test -- 0 * l + 0 * l is the 0 F.pad
    // writes, whatever the neighbours hold)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int i = 0; i < V; ++i) hr[j][i] = 0.f;
    if (col_in) {
        const UpCoord cx = up_coord(ux, sx, w);
        float sc[PRE ? V : 1], sh[PRE ? V : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < V; ++i) { sc[i] = pre_scale[c + i]; sh[i] = pre_shift[c + i]; }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int iy = min(base + j, h - 1);
            const T* rp = x + ((int64_t)(b * h + iy) * w) * ldx + c;
            float v0[V], v1[V];
            uh_load<T, V>(rp + (int64_t)cx.i0 * ldx, v0);
            uh_load<T, V>(rp + (int64_t)cx.i1 * ldx, v1);
            if constexpr (PRE) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    v0[i] = uh_round_as<T>(uh_relu(fmaf(v0[i], sc[i], sh[i])));
                    v1[i] = uh_round_as<T>(uh_relu(fmaf(v1[i], sc[i], sh[i])));
                }
            }
#pragma unroll
            for (int i = 0; i < V; ++i) hr[j][i] = fmaf(cx.l0, v0[i], __fmul_rn(cx.l1, v1[i]));     // spelled out: both instantiations round alike
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (oy0 + r >= Ho) break;
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = 0.f;
        if (row_in[r]) {
            // a, bb (<= a + 1) are uniform over the workgroup: one scalar branch per (a, bb) pair, the blend reads the two rows where
            // they are (copying them into a pair of row registers first cost sixteen moves per output row)
            const int a = min(cy[r].i0 - base, NR - 1), bb = min(cy[r].i1 - base, NR - 1);
            const float l0 = cy[r].l0, l1 = cy[r].l1;
            auto blend = [&](const float (&ra)[V], const float (&rb)[V]) {
                _Pragma("unroll") for (int i = 0; i < V; ++i) o[i] = fmaf(l0, ra[i], __fmul_rn(l1, rb[i]));
            };
            bool done = false;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (!done && a == j) {
                    if (bb == j || j == NR - 1) blend(hr[j], hr[j]);
                    else blend(hr[j], hr[j + 1 < NR ? j + 1 : j]);
                    done = true;
                }
            }
        }
        uh_store<T, V>(y + (((int64_t)b * Ho + oy0 + r) * Wo + ox) * ldy + c, o);
    }
}

// gather form of the transpose: every input pixel collects from the output pixels that read it
__device__ __forceinline__ void up_range(int i, float scale, int in, int& lo, int& hi) {
    const int out = 2 * in;
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float up_weight(int dst, float scale, int in, int i) {
    UpCoord u = up_coord(dst, scale, in);
    float wgt = 0.f;
    if (u.i0 == i) wgt += u.l0;
    if (u.i1 == i) wgt += u.l1;
    return wgt;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx,
                                                             int lddx, int B, int h, int w, int C, int Ho, int Wo, int pt,
                                                             int pl, float sy, float sx) {
    const int G = C / V;
    const int64_t total = (int64_t)B * h * w * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % G) * V;
        int64_t p = idx / G;
        int ix = (int)(p % w);
        int iy = (int)((p / w) % h);
        int b = (int)(p / ((int64_t)w * h));
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        int ylo, yhi, xlo, xhi;
        up_range(iy, sy, h, ylo, yhi);
        up_range(ix, sx, w, xlo, xhi);
        // the column weights do not depend on the row: evaluate them once (at most MAXC candidates by construction
        // of up_range for a x2 upsample; wider ranges -- only when scale == 0 -- fall back to on-the-fly weights)
        constexpr int MAXC = 8;
        float wxs[MAXC];
        const bool cached = (xhi - xlo) < MAXC;
        if (cached) {
#pragma unroll
            for (int k = 0; k < MAXC; ++k) wxs[k] = (xlo + k <= xhi) ? up_weight(xlo + k, sx, w, ix) : 0.f;
        }
        for (int uy = ylo; uy <= yhi; ++uy) {
            float wy = up_weight(uy, sy, h, iy);
            int oy = uy + pt;
            if (wy == 0.f || oy < 0 || oy >= Ho) continue;
#pragma unroll
            for (int k = 0; k < MAXC; ++k) {
                const int ux = xlo + k;
                if (!cached || ux > xhi) break;
                const float wx = wxs[k];
                const int ox = ux + pl;
                if (wx == 0.f || ox < 0 || ox >= Wo) continue;
                float g[V];
                uh_load<T, V>(dy + ((int64_t)(b * Ho + oy) * Wo + ox) * lddy + c, g);
                const float ww = wy * wx;
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] = fmaf(ww, g[i], acc[i]);
            }
            if (!cached)
                for (int ux = xlo; ux <= xhi; ++ux) {
                    float wx = up_weight(ux, sx, w, ix);
                    int ox = ux + pl;
                    if (wx == 0.f || ox < 0 || ox >= Wo) continue;
                    float g[V];
                    uh_load<T, V>(dy + ((int64_t)(b * Ho + oy) * Wo + ox) * lddy + c, g);
                    float ww = wy * wx;
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[i] = fmaf(ww, g[i], acc[i]);
                }
        }
        uh_store<T, V>(dx + p * lddx + c, acc);
    }
}

// Strip form of the same transpose (the hot variant).  Thread = (image, strip of `ty` input rows, input column ix, 16-byte
// channel vector).  It walks the output rows uy that feed its strip in ascending order; per row it forms the column
// pass t = sum_k wx[k] * dy[uy][2*ix-2+k] (the six candidates 2*ix-2 .. 2*ix+3 contain every output column that reads
// input column ix when out = 2*in, align_corners=True: u*scale in [ix-1, ix+1) <=> u in [2ix-2-1/(n-1), 2ix+3+1/(n-1));
// weights come from the forward's own up_coord, so this is the exact transpose) and adds l0*t / l1*t to the two input
// rows i0(uy), i0(uy)+1 it feeds.  i0 is non-decreasing in uy, so two running accumulators suffice and every input
// row is stored exactly once.  ~2.3 gathered 16-byte loads per stored element instead of the full window scan.
// BUF (tensors below 1 GiB): dy is read through a buffer descriptor with 32-bit offsets = column part + row part.  A column that
// carries no weight for this thread has the column part 2^30 and a row outside the strip's range the row part 2^31: the sum is
// out of range (no wrap: 2^30 + 2^31 < 2^32) and the load returns zeros -- so nothing has to be selected away behind the loads
// (the pointer form clamps such loads to a valid address and replaces what they return: 48 selects per output row in a kernel
// that SQ counters show VALU-bound, scratch/r4_up_pmc.sh) and no 64-bit address arithmetic is left in the row loop.
template <typename T, int V, bool BUF = false>
__global__ __launch_bounds__(256) void upsample2x_bwd_strip_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx,
                                                                   int lddx, int B, int h, int w, int C, int Ho, int Wo,
                                                                   int pt, int pl, float sy, float sx, int ty, unsigned dy_bytes = 0) {
    constexpr int NC = 6;
    constexpr int ES = (int)sizeof(T);
    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)dy_bytes, 0x00020000);
    const int G = C / V;
    const int nstrip = (h + ty - 1) / ty;
    const int64_t total = (int64_t)B * nstrip * w * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t_ = idx;
        const int c = (int)(t_ % G) * V; t_ /= G;
        const int ix = (int)(t_ % w); t_ /= w;
        const int strip = (int)(t_ % nstrip);
        const int b = (int)(t_ / nstrip);
        const int iy0 = strip * ty, iy1 = min(iy0 + ty, h);
        const int uxb = 2 * ix - 2;
        float wxs[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int ux = uxb + k, ox = ux + pl;
            wxs[k] = (ux >= 0 && ux < 2 * w && ox >= 0 && ox < Wo) ? up_weight(ux, sx, w, ix) : 0.f;
        }
        int kval = 2;                       // a candidate column with a non-zero weight (2 and 3 always straddle ix)
#pragma unroll
        for (int k = NC - 1; k >= 0; --k)
            if (wxs[k] != 0.f) kval = k;
        const int uy_lo = max(0, 2 * iy0 - 2), uy_hi = min(2 * h - 1, 2 * (iy1 - 1) + 3);
        float a0[V], a1[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        int cur = up_coord(uy_lo, sy, h).i0;
        T* dxp = dx + ((int64_t)b * h * w + ix) * lddx + c;
        // column pass of one output row: tt = sum_k wxs[k] * dy[oy][uxb + pl + k]  (zero for a padding row).  Branch-free:
        // every load goes to a valid address (row and column clamped to ones this thread does use) and a value that
        // must not count is replaced by 0 AFTER the load -- so the loads of two rows (<= 12) are all in flight before the
        // first one is consumed, and a NaN in a pixel this input does not depend on cannot leak in through a zero weight.
        unsigned colpart[BUF ? NC : 1];
        if constexpr (BUF) {
#pragma unroll
            for (int k = 0; k < NC; ++k) colpart[k] = wxs[k] != 0.f ? (unsigned)(((uxb + pl + k) * lddy + c) * ES) : 0x40000000u;
        }
        auto row_sum = [&](int uy, float (&tt)[V]) {
            const int oy = uy + pt;
            const bool live = uy <= uy_hi && oy >= 0 && oy < Ho;
            float g[NC][V];
            if constexpr (BUF) {
                const unsigned rowpart = live ? (unsigned)((b * Ho + oy) * Wo * lddy * ES) : 0x80000000u;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsd, colpart[k] + rowpart, 0, 0);
                    if constexpr (ES == 2) {
                        const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                        for (int i = 0; i < V; ++i) g[k][i] = (float)v[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < V; ++i) g[k][i] = __uint_as_float(raw[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) tt[i] = 0.f;
                // (a zero weight meets a zero: fma(0, 0, t) = t exactly -- t starts at +0 and +0 + -0 = +0, so no sign of zero changes)
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int i = 0; i < V; ++i) tt[i] = fmaf(wxs[k], g[k][i], tt[i]);
            } else {
                const int oyc = min(max(oy, 0), Ho - 1);
                const T* row = dy + ((int64_t)(b * Ho + oyc) * Wo + uxb + pl) * lddy + c;
#pragma unroll
                for (int k = 0; k < NC; ++k) uh_load<T, V>(row + (int64_t)(wxs[k] != 0.f ? k : kval) * lddy, g[k]);
#pragma unroll
                for (int i = 0; i < V; ++i) tt[i] = 0.f;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const bool use = live && wxs[k] != 0.f;
#pragma unroll
                    for (int i = 0; i < V; ++i) tt[i] = use ? fmaf(wxs[k], g[k][i], tt[i]) : tt[i];
                }
            }
        };
        // row pass: fold the column sums of output row uy into the two running input-row accumulators
        auto row_acc = [&](int uy, const float (&tt)[V]) {
            if (uy > uy_hi) return;
            const UpCoord cy = up_coord(uy, sy, h);
            if (cy.i0 > cur) {              // row `cur` is complete (i0 advances by at most one per output row)
                if (cur >= iy0 && cur < iy1) uh_store<T, V>(dxp + (int64_t)cur * w * lddx, a0);
#pragma unroll
                for (int i = 0; i < V; ++i) { a0[i] = a1[i]; a1[i] = 0.f; }
                cur = cy.i0;
            }
            const int oy = uy + pt;
            if (oy < 0 || oy >= Ho) return;
            if (cy.i1 == cy.i0) {
                const float wsum = cy.l0 + cy.l1;
#pragma unroll
                for (int i = 0; i < V; ++i) a0[i] = fmaf(wsum, tt[i], a0[i]);
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) { a0[i] = fmaf(cy.l0, tt[i], a0[i]); a1[i] = fmaf(cy.l1, tt[i], a1[i]); }
            }
        };
        // two output rows per trip: the (up to 12) loads of both rows are independent of the accumulators
        for (int uy = uy_lo; uy <= uy_hi; uy += 2) {
            float t0[V], t1[V];
            row_sum(uy, t0);
            row_sum(uy + 1, t1);
            row_acc(uy, t0);
            row_acc(uy + 1, t1);
        }
        if (cur >= iy0 && cur < iy1) uh_store<T, V>(dxp + (int64_t)cur * w * lddx, a0);
        if (cur + 1 >= iy0 && cur + 1 < iy1) uh_store<T, V>(dxp + (int64_t)(cur + 1) * w * lddx, a1);
    }
}

static inline float up_scale(int in) { return (2 * in > 1) ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }

extern "C" int uh_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int B, int h, int w, int C, int Ho, int Wo,
                                 int pad_top, int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && C > 0 && ldx >= C && ldy >= C, "uh_upsample2x_fwd: bad args");
    UH_REQUIRE(Ho > 0 && Wo > 0, "uh_upsample2x_fwd: bad output size");
    hipStream_t st = (hipStream_t)stream;
    int64_t np = (int64_t)B * Ho * Wo;
    float sy = up_scale(h), sx = up_scale(w);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(x, ldx, C) && uh_vec_ok<T>(y, ldy, C) && (int64_t)Wo * (C / VEC) < (1 << 23) &&
            (int64_t)B * Ho < (1ll << 31)) {
            const int G = C / VEC;
            int gshift = -1;
            for (int k = 0; k < 24; ++k)
                if ((1 << k) == G) gshift = k;
            const unsigned gy = (unsigned)(((int64_t)Wo * G + 255) / 256);
            hipLaunchKernelGGL((upsample2x_fwd_rows_kernel<T, VEC>), dim3((unsigned)(B * ((Ho + UP_ROWS - 1) / UP_ROWS)), gy), dim3(256), 0, st, (const T*)x,
                               ldx, (T*)y, ldy, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx, gshift);
        } else if (uh_vec_ok<T>(x, ldx, C) && uh_vec_ok<T>(y, ldy, C))
            hipLaunchKernelGGL((upsample2x_fwd_kernel<T, VEC>), dim3(pu_grid(np * (C / VEC))), dim3(256), 0, st,
                               (const T*)x, ldx, (T*)y, ldy, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx);
        else
            hipLaunchKernelGGL((upsample2x_fwd_kernel<T, 1>), dim3(pu_grid(np * C)), dim3(256), 0, st, (const T*)x, ldx,
                               (T*)y, ldy, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx);
    });
    UH_CHECK_LAUNCH("upsample2x_fwd_kernel");
    return UH_OK;
}

// BatchNorm + ReLU + bilinear x2 (+ F.pad) in one pass: z = max(x * scale + shift, 0) of the last DoubleConv layer below an Up
// block is read by nothing but the Up block's nn.Upsample (unet_model.py:34-37 / unet_parts.py:70,80), so it is never stored: the
// raw conv output goes in, the up-sampled activation comes out.  Bit-identical to uh_bn_relu_apply followed by uh_upsample2x_fwd.
extern "C" int uh_bn_relu_upsample2x_ok(int B, int h, int w, int C, int Ho, int Wo, int dt) {
    if (dt != UH_F32 && dt != UH_BF16) return 0;
    const int vec = dt == UH_BF16 ? 8 : 4;
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || C % vec != 0) return 0;
    if (Ho < 2 * h || Wo < 2 * w) return 0;
    return ((int64_t)Wo * (C / vec) < (1 << 23) && (int64_t)B * Ho < (1ll << 31)) ? 1 : 0;
}

extern "C" int uh_bn_relu_upsample2x_fwd(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy, int B,
                                         int h, int w, int C, int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(x && y && scale && shift && ldx >= C && ldy >= C, "uh_bn_relu_upsample2x_fwd: bad args");
    UH_REQUIRE(uh_bn_relu_upsample2x_ok(B, h, w, C, Ho, Wo, dt), "uh_bn_relu_upsample2x_fwd: shape outside the fused path (uh_bn_relu_upsample2x_ok)");
    hipStream_t st = (hipStream_t)stream;
    float sy = up_scale(h), sx = up_scale(w);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        UH_REQUIRE((uh_vec_ok<T>(x, ldx, C) && uh_vec_ok<T>(y, ldy, C)), "uh_bn_relu_upsample2x_fwd: tensors must be 16-byte aligned with 16-byte pixel strides");
        const int G = C / VEC;
        int gshift = -1;
        for (int k = 0; k < 24; ++k)
            if ((1 << k) == G) gshift = k;
        const unsigned gy = (unsigned)(((int64_t)Wo * G + 255) / 256);
        hipLaunchKernelGGL((upsample2x_fwd_rows_kernel<T, VEC, true>), dim3((unsigned)(B * ((Ho + UP_ROWS - 1) / UP_ROWS)), gy), dim3(256), 0, st, (const T*)x,
                           ldx, (T*)y, ldy, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx, gshift, scale, shift);
    });
    UH_CHECK_LAUNCH("upsample2x_fwd_rows_kernel (BatchNorm + ReLU input)");
    return UH_OK;
}

extern "C" int uh_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int B, int h, int w, int C, int Ho, int Wo,
                                 int pad_top, int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && dx && B > 0 && h > 0 && w > 0 && C > 0 && lddy >= C && lddx >= C, "uh_upsample2x_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t np = (int64_t)B * h * w;
    float sy = up_scale(h), sx = up_scale(w);
    UH_DISPATCH_DT(dt, T, {
        constexpr int VEC = 16 / (int)sizeof(T);
        if (uh_vec_ok<T>(dy, lddy, C) && uh_vec_ok<T>(dx, lddx, C) && h >= 2 && w >= 2) {
            // strips of 16 input rows; shorter strips on small maps keep >= ~1024 workgroups in flight
            int ty = 16;
            while (ty > 2 && (int64_t)B * ((h + ty - 1) / ty) * w * (C / VEC) < 256 * 1024) ty >>= 1;
            const int64_t nthr = (int64_t)B * ((h + ty - 1) / ty) * w * (C / VEC);
            const int64_t dyb = (int64_t)B * Ho * Wo * lddy * (int64_t)sizeof(T);
            // (UH_UP_BWD_PTR=1, read per call: the pointer form that tensors of 1 GiB and more take, for the test that holds the two
            // forms bit-identical)
            const char* force_ptr = getenv("UH_UP_BWD_PTR");
            if (dyb < (1ll << 30) && !(force_ptr && force_ptr[0] == '1'))
                hipLaunchKernelGGL((upsample2x_bwd_strip_kernel<T, VEC, true>), dim3(pu_grid(nthr)), dim3(256), 0, st, (const T*)dy,
                                   lddy, (T*)dx, lddx, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx, ty, (unsigned)dyb);
            else
                hipLaunchKernelGGL((upsample2x_bwd_strip_kernel<T, VEC>), dim3(pu_grid(nthr)), dim3(256), 0, st, (const T*)dy,
                                   lddy, (T*)dx, lddx, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx, ty);
        } else if (uh_vec_ok<T>(dy, lddy, C) && uh_vec_ok<T>(dx, lddx, C))
            hipLaunchKernelGGL((upsample2x_bwd_kernel<T, VEC>), dim3(pu_grid(np * (C / VEC))), dim3(256), 0, st,
                               (const T*)dy, lddy, (T*)dx, lddx, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx);
        else
            hipLaunchKernelGGL((upsample2x_bwd_kernel<T, 1>), dim3(pu_grid(np * C)), dim3(256), 0, st, (const T*)dy, lddy,
                               (T*)dx, lddx, B, h, w, C, Ho, Wo, pad_top, pad_left, sy, sx);
    });
    UH_CHECK_LAUNCH("upsample2x_bwd_kernel");
    return UH_OK;
}
