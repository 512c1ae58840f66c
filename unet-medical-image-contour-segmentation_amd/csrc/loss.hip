// loss.hip -- the loss reductions of train.py:118-142 as single-pass HBM-bound kernels.
//
//   uh_bce_dice_sums / _grad   BCEWithLogits (train.py:85,121) + dice_loss (utils/dice_score.py:33-36,
//                              reduce_batch_first=True: ONE ratio of batch-global sums) in one pass over the
//                              logits: {sum BCE, sum s*t, sum s, sum t}; the gradient pass takes the (possibly
//                              all-reduced) sums, so a sharded batch reproduces the single-process gradient.
//   uh_ce_dice_sums / _grad    CrossEntropy + multiclass Dice over softmax (train.py:136-142).
//   uh_dice_sums               per-group sums for dice_coeff(reduce_batch_first=False) (evaluate.py:65).
//   uh_boundary_loss           utils/boundary_loss.py:5-118, literal behaviour: value only.
// Reductions: per-thread -> wave shuffle -> LDS -> per-block partial row -> one finishing block in double.
#include "uh_common.h"

constexpr int LOSS_MAXBLK = 1024;
constexpr int LOSS_ROW = 32;     // floats per partial row

extern "C" size_t uh_loss_ws_bytes(int64_t n) {
    (void)n;
    return (size_t)LOSS_MAXBLK * LOSS_ROW * sizeof(float) + 4096;
}

static inline int loss_nblk(int64_t n) {
    int64_t b = (n + 2047) / 2048;
    if (b > LOSS_MAXBLK) b = LOSS_MAXBLK;
    if (b < 1) b = 1;
    return (int)b;
}

// block-reduce K values held per thread; thread 0 writes them to row[]
template <int K>
__device__ __forceinline__ void block_reduce_store(float (&v)[K], float* row) {
    __shared__ float red[4][K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = uh_wave_sum(v[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < K) row[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one block of 256 threads: column sums of partials[nblk][LOSS_ROW] in double (tree reduction)
__device__ __forceinline__ double column_sum_256(const float* __restrict__ partials, int nblk, int k, double* red) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partials[(int64_t)b * LOSS_ROW + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ partials, int nblk, int K,
                                                          float* __restrict__ out) {
    __shared__ double red[256];
    for (int k = 0; k < K; ++k) {
        double s = column_sum_256(partials, nblk, k, red);
        if (threadIdx.x == 0) out[k] = (float)s;
    }
}

// mask / mask_div as the reference's `true_masks //= 2` leaves it for the class indices a mask holds (train.py:119).  A 64-bit
// integer division is ~100 instructions on this chip and these kernels are instruction-bound (2 M elements, a few bytes each):
// values in [0, 2^31) -- every real mask -- take a shift (divisor 2) or a 32-bit division; anything else the full division.
__device__ __forceinline__ float uh_mask_quot(int64_t m, int mask_div) {
    if ((uint64_t)m < (1ull << 31)) {
        const int v = (int)m;
        return (float)(mask_div == 2 ? (v >> 1) : (mask_div == 1 ? v : v / mask_div));
    }
    return (float)(m / mask_div);
}
__device__ __forceinline__ float bin_target(const int64_t* mask, int mask_div, const float* tf, int64_t i) {
    return mask ? uh_mask_quot(mask[i], mask_div) : tf[i];
}

// K column sums at once, each with column_sum_256's own association (thread t adds rows t, t + 256, ... in order, then the same
// binary tree over the 256 threads): bit-identical to K calls, one set of barriers instead of K.  red: K * 256 doubles.
template <int K>
__device__ __forceinline__ void column_sums_256(const float* __restrict__ partials, int nblk, int k0, double* red, double (&out)[K]) {
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < K; ++k) s[k] += (double)partials[(int64_t)b * LOSS_ROW + k0 + k];
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * 256 + threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < K; ++k) red[k * 256 + threadIdx.x] += red[k * 256 + threadIdx.x + o];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = red[k * 256];
    __syncthreads();
}

// ------------------------------------------------------------------------------------ binary path
__global__ __launch_bounds__(256) void bce_dice_sums_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                            int mask_div, const float* __restrict__ tf, int64_t n,
                                                            float* __restrict__ partials) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = logits[i];
        float t = bin_target(mask, mask_div, tf, i);
        float e = expf(-fabsf(x));
        float sp = log1pf(e);
        float s = (x >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
        v[0] += fmaxf(x, 0.f) - x * t + sp;
        v[1] += s * t;
        v[2] += s;
        v[3] += t;
    }
    block_reduce_store<4>(v, partials + (int64_t)blockIdx.x * LOSS_ROW);
}

extern "C" int uh_bce_dice_sums(const float* logits, const int64_t* mask, int mask_div, const float* target_f, int64_t n,
                                float* sums, void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(logits && (mask || target_f) && sums && ws && n > 0, "uh_bce_dice_sums: bad args");
    UH_REQUIRE(!mask || mask_div > 0, "uh_bce_dice_sums: mask_div must be positive");
    UH_REQUIRE(ws_bytes >= uh_loss_ws_bytes(n), "uh_bce_dice_sums: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nblk = loss_nblk(n);
    hipLaunchKernelGGL(bce_dice_sums_kernel, dim3(nblk), dim3(256), 0, st, logits, mask, mask_div, target_f, n, (float*)ws);
    UH_CHECK_LAUNCH("bce_dice_sums_kernel");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nblk, 4, sums);
    UH_CHECK_LAUNCH("loss_finish_kernel");
    return UH_OK;
}

__global__ __launch_bounds__(256) void bce_dice_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                            int mask_div, const float* __restrict__ tf, int64_t n,
                                                            const float* __restrict__ sums, float inv_n, float w_bce,
                                                            float w_dice, const float* __restrict__ gscale,
                                                            float* __restrict__ dl) {
    // dice = (2I + eps) / (S + eps), S = sum s + sum t  (S == 0 -> S := 2I, gradient 0)
    const float eps = 1e-6f;
    const float I2 = 2.f * sums[1];
    const float S = sums[2] + sums[3];
    const float gs = gscale ? gscale[0] : 1.f;
    float ka = 0.f, kb = 0.f;     // d(1 - dice)/ds_i = ka * t_i + kb
    if (S != 0.f) {
        float den = S + eps;
        ka = -2.f / den;
        kb = (I2 + eps) / (den * den);
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = logits[i];
        float t = bin_target(mask, mask_div, tf, i);
        float e = expf(-fabsf(x));
        float s = (x >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
        float g = w_bce * (s - t) * inv_n + w_dice * (ka * t + kb) * s * (1.f - s);
        dl[i] = gs * g;
    }
}

extern "C" int uh_bce_dice_grad(const float* logits, const int64_t* mask, int mask_div, const float* target_f, int64_t n,
                                const float* sums, double n_mean, float w_bce, float w_dice, const float* gscale,
                                float* dlogits, uh_stream stream) {
    UH_REQUIRE(logits && (mask || target_f) && sums && dlogits && n > 0 && n_mean > 0, "uh_bce_dice_grad: bad args");
    int64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(bce_dice_grad_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, logits, mask, mask_div,
                       target_f, n, sums, (float)(1.0 / n_mean), w_bce, w_dice, gscale, dlogits);
    UH_CHECK_LAUNCH("bce_dice_grad_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ multi-class path
constexpr int MAXC = 8;

template <int NC>
__global__ __launch_bounds__(256) void ce_dice_sums_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                           int64_t npix, float* __restrict__ partials) {
    float v[1 + 3 * NC];
#pragma unroll
    for (int k = 0; k < 1 + 3 * NC; ++k) v[k] = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        float x[NC], xl[NC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) { xl[c] = logits[p * NC + c]; m = fmaxf(m, xl[c]); }
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { xl[c] -= m; x[c] = expf(xl[c]); den += x[c]; }
        const int t = (int)mask[p];
        const float inv = 1.f / den;
        const float lse = logf(den);     // logsumexp - m
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float pc = x[c] * inv;
            float tc = (c == t) ? 1.f : 0.f;
            if (c == t) v[0] += lse - xl[c];               // -log softmax_t
            v[1 + c] += pc * tc;
            v[1 + NC + c] += pc;
            v[1 + 2 * NC + c] += tc;
        }
    }
    block_reduce_store<1 + 3 * NC>(v, partials + (int64_t)blockIdx.x * LOSS_ROW);
}

template <int NC>
__global__ __launch_bounds__(256) void ce_dice_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                           int64_t npix, const float* __restrict__ sums, float inv_n,
                                                           float w_ce, float w_dice, const float* __restrict__ gscale,
                                                           float* __restrict__ dl) {
    const float eps = 1e-6f;
    float I = 0.f, S = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { I += sums[1 + c]; S += sums[1 + NC + c] + sums[1 + 2 * NC + c]; }
    const float gs = gscale ? gscale[0] : 1.f;
    float ka = 0.f, kb = 0.f;
    if (S != 0.f) {
        float den = S + eps;
        ka = -2.f / den;
        kb = (2.f * I + eps) / (den * den);
    }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        float x[NC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) { x[c] = logits[p * NC + c]; m = fmaxf(m, x[c]); }
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { x[c] = expf(x[c] - m); den += x[c]; }
        const int t = (int)mask[p];
        const float inv = 1.f / den;
        float gd[NC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            x[c] *= inv;                                   // softmax
            gd[c] = ka * ((c == t) ? 1.f : 0.f) + kb;      // d(1-dice)/dp_c
            dot += x[c] * gd[c];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float g = w_ce * (x[c] - ((c == t) ? 1.f : 0.f)) * inv_n + w_dice * x[c] * (gd[c] - dot);
            dl[p * NC + c] = gs * g;
        }
    }
}

#define UH_NC_SWITCH(ncls, ...)                \
    switch (ncls) {                            \
        case 2: { constexpr int NC = 2; __VA_ARGS__ } break; \
        case 3: { constexpr int NC = 3; __VA_ARGS__ } break; \
        case 4: { constexpr int NC = 4; __VA_ARGS__ } break; \
        case 5: { constexpr int NC = 5; __VA_ARGS__ } break; \
        case 6: { constexpr int NC = 6; __VA_ARGS__ } break; \
        case 7: { constexpr int NC = 7; __VA_ARGS__ } break; \
        case 8: { constexpr int NC = 8; __VA_ARGS__ } break; \
        default: uh_set_error("multi-class losses support 2..8 classes, got %d", ncls); return UH_EINVAL; \
    }

extern "C" int uh_ce_dice_sums(const float* logits, const int64_t* mask, int64_t npix, int ncls, float* sums, void* ws,
                               size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(logits && mask && sums && ws && npix > 0, "uh_ce_dice_sums: bad args");
    UH_REQUIRE(ws_bytes >= uh_loss_ws_bytes(npix), "uh_ce_dice_sums: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nblk = loss_nblk(npix);
    UH_NC_SWITCH(ncls, hipLaunchKernelGGL(ce_dice_sums_kernel<NC>, dim3(nblk), dim3(256), 0, st, logits, mask, npix, (float*)ws););
    UH_CHECK_LAUNCH("ce_dice_sums_kernel");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nblk, 1 + 3 * ncls, sums);
    UH_CHECK_LAUNCH("loss_finish_kernel");
    return UH_OK;
}

extern "C" int uh_ce_dice_grad(const float* logits, const int64_t* mask, int64_t npix, int ncls, const float* sums,
                               double n_mean, float w_ce, float w_dice, const float* gscale, float* dlogits,
                               uh_stream stream) {
    UH_REQUIRE(logits && mask && sums && dlogits && npix > 0 && n_mean > 0, "uh_ce_dice_grad: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t g = (npix + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    float inv_n = (float)(1.0 / n_mean);
    UH_NC_SWITCH(ncls, hipLaunchKernelGGL(ce_dice_grad_kernel<NC>, dim3((unsigned)g), dim3(256), 0, st, logits, mask, npix, sums,
                                          inv_n, w_ce, w_dice, gscale, dlogits););
    UH_CHECK_LAUNCH("ce_dice_grad_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ generic dice sums
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        int64_t group_len, float* __restrict__ partials, int bpg) {
    const int64_t g = blockIdx.x / bpg;
    const int sub = blockIdx.x - (int)(g * bpg);
    const float* xg = x + g * group_len;
    const float* tg = t + g * group_len;
    float v[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)sub * 256 + threadIdx.x; i < group_len; i += (int64_t)bpg * 256) {
        float a = xg[i], b = tg[i];
        v[0] += a * b;
        v[1] += a;
        v[2] += b;
    }
    block_reduce_store<3>(v, partials + (int64_t)blockIdx.x * 4);
}

__global__ void dice_sums_finish_kernel(const float* __restrict__ partials, int bpg, int64_t ngroups, float* __restrict__ sums) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ngroups * 3) return;
    int64_t g = idx / 3;
    int k = (int)(idx - g * 3);
    double s = 0.0;
    for (int b = 0; b < bpg; ++b) s += (double)partials[(g * bpg + b) * 4 + k];
    sums[idx] = (float)s;
}

extern "C" int uh_dice_sums(const float* x, const float* t, int64_t ngroups, int64_t group_len, float* sums, void* ws,
                            size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(x && t && sums && ws && ngroups > 0 && group_len > 0, "uh_dice_sums: bad args");
    int64_t cap = (int64_t)(ws_bytes / (4 * sizeof(float)));
    int bpg = (int)((group_len + 4095) / 4096);
    if (bpg < 1) bpg = 1;
    while (bpg > 1 && ngroups * bpg > cap) bpg >>= 1;
    UH_REQUIRE(ngroups * bpg <= cap, "uh_dice_sums: workspace too small for %lld groups", (long long)ngroups);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_sums_kernel, dim3((unsigned)(ngroups * bpg)), dim3(256), 0, st, x, t, group_len, (float*)ws, bpg);
    UH_CHECK_LAUNCH("dice_sums_kernel");
    hipLaunchKernelGGL(dice_sums_finish_kernel, dim3((unsigned)((ngroups * 3 + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, bpg, ngroups, sums);
    UH_CHECK_LAUNCH("dice_sums_finish_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ boundary_loss
// Literal behaviour of utils/boundary_loss.py.  For each region R in {interior, edge}
// the reference gathers R's pixels of each image in row-major order and dilates the thresholded map
// with a 3-tap window ALONG THAT GATHER ORDER (zero padded per image).  Here every pixel finds its
// predecessor / successor inside its own region analytically:
//   interior = rows [ew, H-ew) x cols [ew, W-ew);   edge = the rest.
struct BRegion { int H, W, ew; bool interior_empty; };

__device__ __forceinline__ bool b_is_edge(const BRegion& g, int h, int w) {
    if (g.ew == 0) return false;
    return h < g.ew || h >= g.H - g.ew || w < g.ew || w >= g.W - g.ew;
}
// linear index (h*W + w) of the previous / next pixel of the same region, or -1
__device__ __forceinline__ int b_prev(const BRegion& g, int h, int w, bool edge) {
    if (!edge) {
        const int c0 = g.ew, c1 = g.W - g.ew;           // interior cols [c0, c1)
        if (w - 1 >= c0) return h * g.W + w - 1;
        if (h - 1 >= g.ew) return (h - 1) * g.W + c1 - 1;
        return -1;
    }
    int ph = h, pw = w - 1;
    if (pw < 0) { ph = h - 1; pw = g.W - 1; }
    if (ph < 0) return -1;
    if (!g.interior_empty && !b_is_edge(g, ph, pw)) pw = g.ew - 1;   // jump over the interior run of that row
    return ph * g.W + pw;
}
__device__ __forceinline__ int b_next(const BRegion& g, int h, int w, bool edge) {
    if (!edge) {
        const int c0 = g.ew, c1 = g.W - g.ew;
        if (w + 1 < c1) return h * g.W + w + 1;
        if (h + 1 < g.H - g.ew) return (h + 1) * g.W + c0;
        return -1;
    }
    int nh = h, nw = w + 1;
    if (nw >= g.W) { nh = h + 1; nw = 0; }
    if (nh >= g.H) return -1;
    if (!g.interior_empty && !b_is_edge(g, nh, nw)) nw = g.W - g.ew;
    return nh * g.W + nw;
}

__global__ __launch_bounds__(256) void boundary_minmax_kernel(const float* __restrict__ pred, int64_t pstride, int64_t n, int HW,
                                                              int64_t bstride, float* __restrict__ partials) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = i / HW, r = i - b * HW;
        float v = pred[b * bstride + r * pstride];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    __shared__ float smn[4], smx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        partials[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}

// target: float [B][H][W], or (mask != NULL) the int64 class-index mask itself with target = mask / mask_div -- what
// train.py:119 / 134 hands to boundary_loss -- so that no float copy of the mask has to be made
template <typename IDX>
__global__ __launch_bounds__(256) void boundary_count_kernel(const float* __restrict__ pred, int64_t pstride, int64_t bstride,
                                                             const float* __restrict__ target, const int64_t* __restrict__ mask,
                                                             int mask_div, int B, BRegion g,
                                                             const float* __restrict__ mm, int nmm, int mm_stride,
                                                             float* __restrict__ partials) {
    // global min / max of the prediction from the <= 512 partial pairs, by the whole block (one thread walking them was
    // 512 dependent-latency loads in every block: 45 of this kernel's 56 us)
    __shared__ float s_mn[4], s_mx[4];
    {
        float mn = INFINITY, mx = -INFINITY;
        // (pair k at mm[k * mm_stride], mm[k * mm_stride + 1]: packed pairs from boundary_minmax_kernel, or columns 4 / 5 of
        // the fused pass's partial rows)
        for (int k = threadIdx.x; k < nmm; k += 256) { mn = fminf(mn, mm[(int64_t)k * mm_stride]); mx = fmaxf(mx, mm[(int64_t)k * mm_stride + 1]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    }
    __syncthreads();
    const float gmn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    const float gmx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const bool sig = (gmn < -10.f || gmx > 10.f);         // boundary_loss.py:28
    // IDX = int when B * H * W (and the launch's stride past it) fits 31 bits: the per-pixel div / mod chain in 32-bit arithmetic
    const int HW = g.H * g.W;
    const IDX n = (IDX)B * HW;
    // thresholded prediction / target at (b, linear r)
    auto pbin = [&](int64_t b, int r) -> float {
        float v = pred[b * bstride + (int64_t)r * pstride];
        if (sig) v = 1.f / (1.f + expf(-v));
        return v > 0.5f ? 1.f : 0.f;
    };
    auto tbin = [&](int64_t b, int r) -> float {                                                      // boundary_loss.py:37
        if (mask) return uh_mask_quot(mask[b * HW + r], mask_div) == 255.f ? 1.f : 0.f;
        return target[b * HW + r] == 255.f ? 1.f : 0.f;
    };
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // {inter, psum, tsum} x {interior, edge}
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (IDX)gridDim.x * blockDim.x) {
        const IDX bq = i / HW;
        const int64_t b = bq;
        int r = (int)(i - bq * HW);
        int h = r / g.W, w = r - h * g.W;
        bool edge = g.interior_empty ? true : b_is_edge(g, h, w);
        if (g.ew == 0) edge = false;
        int pr = b_prev(g, h, w, edge), nx = b_next(g, h, w, edge);
        float pd = pbin(b, r), td = tbin(b, r);
        if (pr >= 0) { pd = fmaxf(pd, pbin(b, pr)); td = fmaxf(td, tbin(b, pr)); }
        if (nx >= 0) { pd = fmaxf(pd, pbin(b, nx)); td = fmaxf(td, tbin(b, nx)); }
        int o = edge ? 3 : 0;
        v[o + 0] += pd * td;
        v[o + 1] += pd;
        v[o + 2] += td;
    }
    block_reduce_store<6>(v, partials + (int64_t)blockIdx.x * LOSS_ROW);
}

__device__ float boundary_region_loss(double inter, double psum, double tsum, double N, float smooth) {
    if (N <= 0.0) return 0.f;                                  // boundary_loss.py:64-65
    float fi = (float)inter, fp = (float)psum, ft = (float)tsum;
    float uni = fp + ft - fi;
    float iou = (fi + smooth) / (uni + smooth);
    // BCE-with-logits of logit(clamp(pb, 1e-6, 1-1e-6)) against tb, summed, / N   (boundary_loss.py:92-93)
    float p1 = fminf(fmaxf(1.f, 1e-6f), 1.f - 1e-6f), p0 = fminf(fmaxf(0.f, 1e-6f), 1.f - 1e-6f);
    float l1 = logf(p1 / (1.f - p1)), l0 = logf(p0 / (1.f - p0));
    double n11 = inter, n10 = psum - inter, n01 = tsum - inter, n00 = N - psum - tsum + inter;
    double bce = n11 * (double)uh_bce_logits(l1, 1.f) + n10 * (double)uh_bce_logits(l1, 0.f) +
                 n01 * (double)uh_bce_logits(l0, 1.f) + n00 * (double)uh_bce_logits(l0, 0.f);
    return (1.f - iou) + 0.5f * (float)(bce / N);
}

__global__ __launch_bounds__(256) void boundary_finish_kernel(const float* __restrict__ partials, int nblk, int B, BRegion g,
                                                              float edge_weight, float smooth, float* __restrict__ out) {
    __shared__ double red[256];
    double s[6];
    for (int k = 0; k < 6; ++k) s[k] = column_sum_256(partials, nblk, k, red);
    if (threadIdx.x != 0) return;
    double n_int = 0.0, n_edge = 0.0;
    const double tot = (double)g.H * g.W;
    if (g.ew == 0) { n_int = tot; n_edge = 0.0; }
    else if (g.interior_empty) { n_int = 0.0; n_edge = tot; }
    else { n_int = (double)(g.H - 2 * g.ew) * (g.W - 2 * g.ew); n_edge = tot - n_int; }
    n_int *= B; n_edge *= B;
    float normal = boundary_region_loss(s[0], s[1], s[2], n_int, smooth);
    float edge = boundary_region_loss(s[3], s[4], s[5], n_edge, smooth);
    out[0] = (normal + edge_weight * edge) / (1.f + edge_weight);     // boundary_loss.py:44
}

static int boundary_loss_impl(const float* pred, int64_t pstride, int64_t bstride, const float* target, const int64_t* mask,
                              int mask_div, int B, int H, int W, int edge_width, float edge_weight, float smooth, float* out,
                              void* ws, size_t ws_bytes, uh_stream stream);

extern "C" int uh_boundary_loss(const float* pred, int64_t pstride, int64_t bstride, const float* target, int B, int H, int W, int edge_width,
                                float edge_weight, float smooth, float* out, void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(target, "uh_boundary_loss: null target");
    return boundary_loss_impl(pred, pstride, bstride, target, nullptr, 1, B, H, W, edge_width, edge_weight, smooth, out, ws, ws_bytes, stream);
}

extern "C" int uh_boundary_loss_mask(const float* pred, int64_t pstride, int64_t bstride, const int64_t* mask, int mask_div, int B,
                                     int H, int W, int edge_width, float edge_weight, float smooth, float* out, void* ws,
                                     size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(mask && mask_div >= 1, "uh_boundary_loss_mask: null mask / bad divisor");
    return boundary_loss_impl(pred, pstride, bstride, nullptr, mask, mask_div, B, H, W, edge_width, edge_weight, smooth, out, ws, ws_bytes, stream);
}

static int boundary_loss_impl(const float* pred, int64_t pstride, int64_t bstride, const float* target, const int64_t* mask,
                              int mask_div, int B, int H, int W, int edge_width, float edge_weight, float smooth, float* out,
                              void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(pred && (target || mask) && out && ws && B > 0 && H > 0 && W > 0 && pstride > 0 && edge_width >= 0,
               "uh_boundary_loss: bad args");
    UH_REQUIRE(bstride > 0, "uh_boundary_loss: bad batch stride");
    UH_REQUIRE(ws_bytes >= uh_loss_ws_bytes((int64_t)B * H * W), "uh_boundary_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BRegion g;
    g.H = H; g.W = W; g.ew = edge_width;
    g.interior_empty = (edge_width > 0) && (2 * edge_width >= H || 2 * edge_width >= W);
    int64_t n = (int64_t)B * H * W;
    int nblk = loss_nblk(n);
    float* partials = (float*)ws;
    float* mmbuf = (float*)((char*)ws + (size_t)LOSS_MAXBLK * LOSS_ROW * sizeof(float));   // 4096-byte tail: 512 pairs
    int nmm = nblk < 512 ? nblk : 512;
    hipLaunchKernelGGL(boundary_minmax_kernel, dim3(nmm), dim3(256), 0, st, pred, pstride, n, H * W, bstride, mmbuf);
    UH_CHECK_LAUNCH("boundary_minmax_kernel");
    if (n + (int64_t)nblk * 256 < (1ll << 31))
        hipLaunchKernelGGL(boundary_count_kernel<int>, dim3(nblk), dim3(256), 0, st, pred, pstride, bstride, target, mask, mask_div, B, g,
                           (const float*)mmbuf, nmm, 2, partials);
    else
        hipLaunchKernelGGL(boundary_count_kernel<int64_t>, dim3(nblk), dim3(256), 0, st, pred, pstride, bstride, target, mask, mask_div, B, g,
                           (const float*)mmbuf, nmm, 2, partials);
    UH_CHECK_LAUNCH("boundary_count_kernel");
    hipLaunchKernelGGL(boundary_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)partials, nblk, B, g, edge_weight,
                       smooth, out);
    UH_CHECK_LAUNCH("boundary_finish_kernel");
    return UH_OK;
}

// ------------------------------------------------------------------------------------ scalar assembly
__global__ void dice_from_sums_kernel(const float* __restrict__ sums, int64_t ngroups, float eps, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double acc = 0.0;
    for (int64_t g = 0; g < ngroups; ++g) {
        float inter = 2.f * sums[g * 3 + 0];
        float sets = sums[g * 3 + 1] + sums[g * 3 + 2];
        if (sets == 0.f) sets = inter;                       // dice_score.py:16
        acc += (double)((inter + eps) / (sets + eps));
    }
    out[0] = (float)(acc / (double)ngroups);
}

extern "C" int uh_dice_from_sums(const float* sums, int64_t ngroups, float eps, float* out, uh_stream stream) {
    UH_REQUIRE(sums && out && ngroups > 0, "uh_dice_from_sums: bad args");
    hipLaunchKernelGGL(dice_from_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, ngroups, eps, out);
    UH_CHECK_LAUNCH("dice_from_sums_kernel");
    return UH_OK;
}

__global__ void seg_loss_finish_kernel(const float* __restrict__ sums, int ncls, float inv_n, const float* __restrict__ boundary,
                                       float w_boundary, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const float eps = 1e-6f;
    float I = 0.f, S = 0.f;
    if (ncls <= 1) { I = sums[1]; S = sums[2] + sums[3]; }
    else for (int c = 0; c < ncls; ++c) { I += sums[1 + c]; S += sums[1 + ncls + c] + sums[1 + 2 * ncls + c]; }
    float inter = 2.f * I;
    if (S == 0.f) S = inter;
    float dice_loss = 1.f - (inter + eps) / (S + eps);
    float ce = sums[0] * inv_n;
    float bl = boundary ? boundary[0] : 0.f;
    out[0] = ce + dice_loss + w_boundary * bl;
    out[1] = ce;
    out[2] = dice_loss;
    out[3] = bl;
}

extern "C" int uh_seg_loss_binary_finish(const float* sums, double n_mean, const float* boundary, float w_boundary,
                                         float* out, uh_stream stream) {
    UH_REQUIRE(sums && out && n_mean > 0, "uh_seg_loss_binary_finish: bad args");
    hipLaunchKernelGGL(seg_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, 1, (float)(1.0 / n_mean),
                       boundary, w_boundary, out);
    UH_CHECK_LAUNCH("seg_loss_finish_kernel");
    return UH_OK;
}

extern "C" int uh_seg_loss_multiclass_finish(const float* sums, int ncls, double n_mean, const float* boundary,
                                             float w_boundary, float* out, uh_stream stream) {
    UH_REQUIRE(sums && out && n_mean > 0 && ncls >= 2 && ncls <= MAXC, "uh_seg_loss_multiclass_finish: bad args");
    hipLaunchKernelGGL(seg_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, ncls, (float)(1.0 / n_mean),
                       boundary, w_boundary, out);
    UH_CHECK_LAUNCH("seg_loss_finish_kernel");
    return UH_OK;
}


// ------------------------------------------------------------------------------------ train.py:119-134 in three launches
// The single-process binary loss (BCE mean + Dice + w * boundary_loss) used to be six launches plus the torch glue around
// them (isnan, two fills, two copies): every one of them a 5-10 us step on the critical path between the forward and the
// backward pass.  Here: ONE pass over the logits forms the BCE / Dice partial sums AND the prediction's min / max (the
// sigmoid decision of boundary_loss.py:28), the boundary counts follow, and one finishing block writes the four sums the
// backward pass needs, the four loss terms and the NaN flag of train.py:149.  Arithmetic, block counts and reduction orders
// are those of uh_bce_dice_sums / uh_boundary_loss_mask / uh_seg_loss_binary_finish: the results are bit-identical.
__global__ __launch_bounds__(256) void bce_dice_mm_sums_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                               int mask_div, int64_t n, float* __restrict__ partials) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = logits[i];
        float t = uh_mask_quot(mask[i], mask_div);
        float e = expf(-fabsf(x));
        float sp = log1pf(e);
        float s = (x >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
        v[0] += fmaxf(x, 0.f) - x * t + sp;
        v[1] += s * t;
        v[2] += s;
        v[3] += t;
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
    }
    __shared__ float smn[4], smx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    float* row = partials + (int64_t)blockIdx.x * LOSS_ROW;
    block_reduce_store<4>(v, row);                        // (its barrier also covers smn / smx)
    if (threadIdx.x == 0) {
        row[4] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        row[5] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}

__global__ __launch_bounds__(256) void seg_loss_fused_finish_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                                    int nblk, int B, BRegion g, float edge_weight, float smooth,
                                                                    float w_boundary, float inv_n, float* __restrict__ sums,
                                                                    float* __restrict__ out) {
    __shared__ double red[6 * 256];
    float fs[4];
    {
        double d[4];
        column_sums_256<4>(pa, nblk, 0, red, d);
        for (int k = 0; k < 4; ++k) fs[k] = (float)d[k];
    }
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (pb) column_sums_256<6>(pb, nblk, 0, red, s);
    if (threadIdx.x != 0) return;
    float bl = 0.f;
    if (pb) {
        double n_int = 0.0, n_edge = 0.0;
        const double tot = (double)g.H * g.W;
        if (g.ew == 0) { n_int = tot; n_edge = 0.0; }
        else if (g.interior_empty) { n_int = 0.0; n_edge = tot; }
        else { n_int = (double)(g.H - 2 * g.ew) * (g.W - 2 * g.ew); n_edge = tot - n_int; }
        n_int *= B; n_edge *= B;
        const float normal = boundary_region_loss(s[0], s[1], s[2], n_int, smooth);
        const float edge = boundary_region_loss(s[3], s[4], s[5], n_edge, smooth);
        bl = (normal + edge_weight * edge) / (1.f + edge_weight);     // boundary_loss.py:44
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sums[k] = fs[k];
    const float eps = 1e-6f;
    const float inter = 2.f * fs[1];
    float S = fs[2] + fs[3];
    if (S == 0.f) S = inter;                                          // dice_score.py:16
    const float dice_loss = 1.f - (inter + eps) / (S + eps);
    const float ce = fs[0] * inv_n;
    const float total = ce + dice_loss + w_boundary * bl;
    out[0] = total;
    out[1] = ce;
    out[2] = dice_loss;
    out[3] = bl;
    out[4] = (total != total) ? 1.f : 0.f;                            // train.py:149 `torch.isnan(loss)`
}

extern "C" size_t uh_seg_loss_fused_ws_bytes(void) { return (size_t)2 * LOSS_MAXBLK * LOSS_ROW * sizeof(float); }

extern "C" int uh_seg_loss_binary_fused(const float* logits, const int64_t* mask, int mask_div, int B, int H, int W,
                                        int edge_width, float edge_weight, float smooth, float w_boundary, double n_mean,
                                        float* sums, float* out, void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(logits && mask && sums && out && ws, "uh_seg_loss_binary_fused: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && mask_div >= 1 && edge_width >= 0 && n_mean > 0, "uh_seg_loss_binary_fused: bad args");
    UH_REQUIRE(ws_bytes >= uh_seg_loss_fused_ws_bytes(), "uh_seg_loss_binary_fused: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    const int nblk = loss_nblk(n);
    float* pa = (float*)ws;
    float* pb = pa + (size_t)LOSS_MAXBLK * LOSS_ROW;
    BRegion g;
    g.H = H; g.W = W; g.ew = edge_width;
    g.interior_empty = (edge_width > 0) && (2 * edge_width >= H || 2 * edge_width >= W);
    hipLaunchKernelGGL(bce_dice_mm_sums_kernel, dim3(nblk), dim3(256), 0, st, logits, mask, mask_div, n, pa);
    UH_CHECK_LAUNCH("bce_dice_mm_sums_kernel");
    const bool with_boundary = w_boundary != 0.f;
    if (with_boundary) {
        if (n + (int64_t)nblk * 256 < (1ll << 31))
            hipLaunchKernelGGL(boundary_count_kernel<int>, dim3(nblk), dim3(256), 0, st, logits, (int64_t)1, (int64_t)H * W, (const float*)nullptr,
                               mask, mask_div, B, g, (const float*)(pa + 4), nblk, LOSS_ROW, pb);
        else
            hipLaunchKernelGGL(boundary_count_kernel<int64_t>, dim3(nblk), dim3(256), 0, st, logits, (int64_t)1, (int64_t)H * W, (const float*)nullptr,
                               mask, mask_div, B, g, (const float*)(pa + 4), nblk, LOSS_ROW, pb);
        UH_CHECK_LAUNCH("boundary_count_kernel");
    }
    hipLaunchKernelGGL(seg_loss_fused_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)pa,
                       with_boundary ? (const float*)pb : (const float*)nullptr, nblk, B, g, edge_weight, smooth, w_boundary,
                       (float)(1.0 / n_mean), sums, out);
    UH_CHECK_LAUNCH("seg_loss_fused_finish_kernel");
    return UH_OK;
}
