// conv3x3.hip -- nn.Conv2d(k=3, padding=1, bias=False) forward / backward-data / backward-weights
// for gfx950 (reference call sites: unet/unet_parts.py:15,18).
//
// Kernels
//   conv3x3_fwd_mfma    im2col-free implicit GEMM on MFMA.  One workgroup = a 16x16 pixel tile x
//                       BN output channels.  Per K-chunk (64 bytes of channels per pixel) the 18x18
//                       halo tile is staged ONCE into LDS (register staged, zero padded, double
//                       buffered) and the 9 taps are 9 shifted fragment reads of it; a row fragment
//                       is read once per column shift and reused by the 3 row taps.  Filter
//                       fragments go global->VGPR (L2 resident, 16 B per lane).  bf16 uses
//                       v_mfma_f32_16x16x32_bf16, fp32 uses the exact v_mfma_f32_16x16x4_f32.
//                       Epilogue: store y, and per-tile per-channel sum / sum of squares for BatchNorm.
//   conv3x3_fwd_stem    Cin <= 4 (HBM-bound): lane = output channel, halo tile in LDS, broadcast reads.
//   conv3x3_fwd_generic any shape (small-width parity vehicles UNet_T / UNet_S).
//   conv3x3_wgrad_mfma  dW = dy^T (x) x on MFMA 32x32 tiles; pixels are the contraction index, so both
//                       operands are read from [pixel][channel] LDS tiles with ds_read_b64_tr_b16 (bf16)
//                       or ds_read_b32 (fp32); split over pixel tiles, fp32 slabs + a reduce kernel
//                       (deterministic, no atomics).
//   conv3x3_wgrad_stem / _generic
// Backward-data is conv3x3_fwd with the flipped/transposed filter produced by uh_pack_w3x3.
#include "uh_vec.h"

// =====================================================================================
// weight (un)packing
// =====================================================================================
// Two layouts of a packed filter [rows O][9 taps][K input channels]:
//   KRSC            element (o, tap, k) at (o * 9 + tap) * K + k                          (every kernel reads it)
//   fragment-major  (dt | UH_WFRAG) 1 KiB blocks = one MFMA A-fragment as the LDS-DMA MFMA kernel consumes it: block
//                   ((g * 9 + tap) * (K / CK) + chunk), lane = kpart * 16 + m holds the 16 bytes
//                   (row(g, m), tap, chunk * CK + kpart * VEC ...): a fragment load of the conv kernel is 1 KiB of
//                   CONTIGUOUS memory (8 whole cache lines) instead of 16 rows x 64 bytes (16 half lines) -- the vector
//                   memory path (TA / L1), not the matrix pipe, was what the forward kernel waited for.
// row(g, m) = which filter row MFMA row m of 16-row group g holds; the conv kernel hands rows to the MFMAs in an order that
// makes a lane's accumulators 16-byte pieces of the NHWC output (conv3x3_fwd_mfma_v2):
//   bf16, O % 128 == 0:  32-row groups G = g >> 1, n = g & 1:  row = 32 G + (m >> 2) * 8 + n * 4 + (m & 3)
//   bf16, otherwise:     row = 16 g + cg(m >> 2) * 4 + (m & 3),  cg = {0, 2, 1, 3}
//   fp32:                row = 16 g + m
__host__ __device__ __forceinline__ int uh_wfrag_mode(int es, int rows) { return es == 2 ? ((rows % 128 == 0) ? 2 : 1) : 0; }
template <int ES>
__device__ __forceinline__ int64_t uh_wfrag_index(int o, int tap, int k, int K, int mode) {
    constexpr int CK = 64 / ES, VEC = 16 / ES;
    int g, m;
    if (mode == 2) {
        const int c32 = o & 31;
        g = ((o >> 5) << 1) | ((c32 >> 2) & 1);
        m = ((c32 >> 3) << 2) | (c32 & 3);
    } else if (mode == 1) {
        const int c16 = o & 15, q = c16 >> 2;
        g = o >> 4;
        m = ((((q & 1) << 1) | (q >> 1)) << 2) | (c16 & 3);
    } else {
        g = o >> 4;
        m = o & 15;
    }
    const int chunk = k / CK, e = k - chunk * CK, kp = e / VEC, v = e - kp * VEC;
    return ((((int64_t)g * 9 + tap) * (K / CK) + chunk) * 64 + (kp * 16 + m)) * VEC + v;
}

template <typename T>
__global__ void pack_w3x3_kernel(const float* __restrict__ w, int64_t sO, int64_t sI, int64_t sH, int64_t sW,
                                 int Cout, int Cin, T* __restrict__ wf, T* __restrict__ wd, int frag_f, int frag_d) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)Cout * 9 * Cin;
    if (idx >= total) return;
    int i = (int)(idx % Cin);
    int t = (int)((idx / Cin) % 9);
    int o = (int)(idx / (9 * (int64_t)Cin));
    int r = t / 3, s = t - 3 * r;
    float v = w[o * sO + i * sI + r * sH + s * sW];
    wf[frag_f ? uh_wfrag_index<(int)sizeof(T)>(o, t, i, Cin, uh_wfrag_mode(sizeof(T), Cout)) : idx] = uh_from_f32<T>(v);
    if (wd) {
        const int td = (2 - r) * 3 + (2 - s);
        const int64_t k = frag_d ? uh_wfrag_index<(int)sizeof(T)>(i, td, o, Cout, uh_wfrag_mode(sizeof(T), Cin))
                                 : ((int64_t)i * 9 + td) * Cout + o;
        wd[k] = uh_from_f32<T>(v);
    }
}

__global__ void unpack_dw3x3_kernel(const float* __restrict__ dwk, float* __restrict__ dw, int64_t sO, int64_t sI,
                                    int64_t sH, int64_t sW, int Cout, int Cin) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)Cout * 9 * Cin;
    if (idx >= total) return;
    int i = (int)(idx % Cin);
    int t = (int)((idx / Cin) % 9);
    int o = (int)(idx / (9 * (int64_t)Cin));
    int r = t / 3, s = t - 3 * r;
    dw[o * sO + i * sI + r * sH + s * sW] = dwk[idx];
}

// bf16x3 pack: hi = bf16(w), lo = bf16(w - hi); forward copy [hi | lo] x [o][tap][i], backward-data copy [hi | lo] x [i][tap'][o]
__global__ void pack_w3x3_split_kernel(const float* __restrict__ w, int64_t sO, int64_t sI, int64_t sH, int64_t sW, int Cout,
                                       int Cin, bf16_t* __restrict__ wf, bf16_t* __restrict__ wd) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)Cout * 9 * Cin;
    if (idx >= total) return;
    int i = (int)(idx % Cin);
    int t = (int)((idx / Cin) % 9);
    int o = (int)(idx / (9 * (int64_t)Cin));
    int r = t / 3, s = t - 3 * r;
    const float v = w[o * sO + i * sI + r * sH + s * sW];
    const bf16_t h = (bf16_t)v, l = (bf16_t)(v - (float)h);
    wf[idx] = h;
    wf[total + idx] = l;
    if (wd) {
        const int64_t k = (((int64_t)i * 3 + (2 - r)) * 3 + (2 - s)) * Cout + o;
        wd[k] = h;
        wd[total + k] = l;
    }
}

// can a [rows][9][K] filter be packed fragment-major for element size es?
static inline bool uh_wfrag_shape_ok(int rows, int K, int es) { return rows % 16 == 0 && K % (64 / es) == 0; }

extern "C" int uh_pack_w3x3(const float* w, int64_t sO, int64_t sI, int64_t sH, int64_t sW, int Cout, int Cin,
                            void* w_fwd, void* w_dgrad, int dt, uh_stream stream) {
    UH_REQUIRE(w && w_fwd && Cout > 0 && Cin > 0, "uh_pack_w3x3: bad arguments");
    // dt | UH_WFRAG: fragment-major forward copy; dt | UH_WFRAG_D: fragment-major backward-data copy
    const int frag_f = (dt & UH_WFRAG) ? 1 : 0, frag_d = (dt & UH_WFRAG_D) ? 1 : 0;
    dt &= ~(UH_WFRAG | UH_WFRAG_D);
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_pack_w3x3: bad dtype %d", dt);
    UH_REQUIRE(!(frag_f || frag_d) || dt != UH_F32X3, "uh_pack_w3x3: bf16x3 packs are KRSC only");
    const int es_ = dt == UH_BF16 ? 2 : 4;
    UH_REQUIRE(!frag_f || uh_wfrag_shape_ok(Cout, Cin, es_), "uh_pack_w3x3: fragment-major forward pack needs Cout %% 16 == 0 and Cin %% %d == 0", 64 / es_);
    UH_REQUIRE(!frag_d || uh_wfrag_shape_ok(Cin, Cout, es_), "uh_pack_w3x3: fragment-major backward-data pack needs Cin %% 16 == 0 and Cout %% %d == 0", 64 / es_);
    int64_t total = (int64_t)Cout * 9 * Cin;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_F32X3) {
        hipLaunchKernelGGL(pack_w3x3_split_kernel, grid, block, 0, st, w, sO, sI, sH, sW, Cout, Cin, (bf16_t*)w_fwd,
                           (bf16_t*)w_dgrad);
        UH_CHECK_LAUNCH("pack_w3x3_split_kernel");
        return UH_OK;
    }
    if (dt == UH_BF16)
        hipLaunchKernelGGL(pack_w3x3_kernel<bf16_t>, grid, block, 0, st, w, sO, sI, sH, sW, Cout, Cin,
                           (bf16_t*)w_fwd, (bf16_t*)w_dgrad, frag_f, frag_d);
    else
        hipLaunchKernelGGL(pack_w3x3_kernel<float>, grid, block, 0, st, w, sO, sI, sH, sW, Cout, Cin,
                           (float*)w_fwd, (float*)w_dgrad, frag_f, frag_d);
    UH_CHECK_LAUNCH("uh_pack_w3x3");
    return UH_OK;
}

// Zero-padded pack for the small-width layers (uh_conv3x3_fwd_narrow): the filter [Cout][C0 + C1][3][3] is packed as the
// 64-aligned layer [Coutp][Cp0 + Cp1]: source block 0 (C0 channels) goes to padded channels 0.., block 1 (C1) to Cp0..;
// everything else is zero.  Backward-data copy: [Cp0 + Cp1][3][3][Coutp], i.e. the per-source packs are its row blocks.
template <typename T>
__global__ void pack_w3x3_padded_kernel(const float* __restrict__ w, int64_t sO, int64_t sI, int64_t sH, int64_t sW,
                                        int Cout, int C0, int C1, int Coutp, int Cp0, int Cp1, T* __restrict__ wf,
                                        T* __restrict__ wd) {
    const int Cinp = Cp0 + Cp1;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)Coutp * 9 * Cinp;
    if (idx >= total) return;
    int ip = (int)(idx % Cinp);
    int t = (int)((idx / Cinp) % 9);
    int o = (int)(idx / (9 * (int64_t)Cinp));
    int r = t / 3, sx = t - 3 * r;
    int ci = -1;
    if (ip < Cp0) { if (ip < C0) ci = ip; }
    else if (ip - Cp0 < C1) ci = C0 + (ip - Cp0);
    float v = (o < Cout && ci >= 0) ? w[o * sO + ci * sI + r * sH + sx * sW] : 0.f;
    wf[idx] = uh_from_f32<T>(v);
    if (wd) wd[(((int64_t)ip * 3 + (2 - r)) * 3 + (2 - sx)) * Coutp + o] = uh_from_f32<T>(v);
}

extern "C" int uh_pack_w3x3_padded(const float* w, int64_t sO, int64_t sI, int64_t sH, int64_t sW, int Cout, int C0, int C1,
                                   int Coutp, int Cp0, int Cp1, void* w_fwd, void* w_dgrad, int dt, uh_stream stream) {
    UH_REQUIRE(w && w_fwd && Cout > 0 && C0 > 0 && C1 >= 0, "uh_pack_w3x3_padded: bad arguments");
    UH_REQUIRE(Coutp >= Cout && Cp0 >= C0 && Cp1 >= C1, "uh_pack_w3x3_padded: padded counts below the real ones");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16, "uh_pack_w3x3_padded: bad dtype %d", dt);
    int64_t total = (int64_t)Coutp * 9 * (Cp0 + Cp1);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_BF16)
        hipLaunchKernelGGL(pack_w3x3_padded_kernel<bf16_t>, grid, block, 0, st, w, sO, sI, sH, sW, Cout, C0, C1, Coutp, Cp0, Cp1,
                           (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
    else
        hipLaunchKernelGGL(pack_w3x3_padded_kernel<float>, grid, block, 0, st, w, sO, sI, sH, sW, Cout, C0, C1, Coutp, Cp0, Cp1,
                           (float*)w_fwd, (float*)w_dgrad);
    UH_CHECK_LAUNCH("uh_pack_w3x3_padded");
    return UH_OK;
}

// All 3x3 filters of a model in ONE launch (the per-layer form costs ~10 us of launch + tail per layer, 18 layers per
// step).  table[l] = {w pointer, sO, sI, sH, sW, Cout, Cin, first element of layer l in the flat outputs, first TILE of
// layer l, 0}.  Workgroup = one 32(o) x 32(i) tile of one tap, transposed through LDS so that BOTH packed copies are
// written in 64-byte runs (the backward-data copy [i][tap'][o] is a transpose of the forward copy [o][tap][i]; written
// straight from registers it would be 2-byte stores 9*Cout elements apart).
template <typename T>
__global__ __launch_bounds__(256) void pack_w3x3_batched_kernel(const long long* __restrict__ table, int nlayers,
                                                                long long ntiles, T* __restrict__ wf, T* __restrict__ wd) {
    __shared__ float tile[32][33];
    __shared__ __attribute__((aligned(16))) float vtile[32][36];       // the 16-byte form's tile: rows 16-byte aligned
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        int lo = 0, hi = nlayers - 1;                 // last layer whose first tile <= tl
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid * 10 + 8] <= tl) lo = mid; else hi = mid - 1;
        }
        const long long* e = table + lo * 10;
        const float* w = reinterpret_cast<const float*>(e[0]);
        const long long sO = e[1], sI = e[2], sH = e[3], sW = e[4];
        const int Cout = (int)e[5], Cin = (int)e[6];
        const long long base = e[7];
        const int frag_f = (int)(e[9] & 1), frag_d = (int)((e[9] >> 1) & 1);      // fragment-major copies (see uh_wfrag_index)
        const int ti = (Cin + 31) >> 5, to = (Cout + 31) >> 5;
        long long k = tl - e[8];
        const int it = (int)(k % ti); k /= ti;
        const int ot = (int)(k % to);
        const int t = (int)(k / to);                   // tap 0..8
        const int r = t / 3, s_ = t - 3 * r;
        const int o0 = ot * 32, i0 = it * 32;
        __syncthreads();                               // the previous tile's readers are done
        if constexpr (sizeof(T) == 2) {
            // Whole 32 x 32 tiles of channels_last bf16 layers (every 3x3 layer of the UNets but the stem): one 16-byte load per
            // thread, and both copies leave as 16-byte pieces -- eight consecutive input channels of one filter row are contiguous
            // in the forward copy (KRSC and fragment-major alike), eight consecutive output channels in the backward-data copy.
            // The 2-byte stores of the element-wise form below ran this kernel at 2.7 TB/s (50 us per step).
            if (sI == 1 && o0 + 32 <= Cout && i0 + 32 <= Cin && (sO & 3) == 0 && (sH & 3) == 0 && (sW & 3) == 0 &&
                (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
                {
                    const int row = threadIdx.x >> 3, q = threadIdx.x & 7;           // 32 rows x 8 float4
                    const f32x4 v = *reinterpret_cast<const f32x4*>(w + (long long)(o0 + row) * sO + (i0 + 4 * q) + r * sH + s_ * sW);
                    *reinterpret_cast<f32x4*>(&vtile[row][4 * q]) = v;
                }
                __syncthreads();
                const int half = threadIdx.x >> 7, t7 = threadIdx.x & 127;
                const int row = t7 >> 2, part = t7 & 3;                              // 32 rows x four 8-element pieces
                if (half == 0) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&vtile[row][8 * part]);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&vtile[row][8 * part + 4]);
                    const bf16x8 o8 = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
                    const int o = o0 + row, i = i0 + 8 * part;
                    const long long k = frag_f ? uh_wfrag_index<2>(o, t, i, Cin, uh_wfrag_mode(2, Cout)) : ((long long)o * 9 + t) * Cin + i;
                    *reinterpret_cast<bf16x8*>(wf + base + k) = o8;
                } else if (wd) {
                    const int td = (2 - r) * 3 + (2 - s_);
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)vtile[8 * part + e][row];      // column `row` = input channel i0 + row
                    const int i = i0 + row, o = o0 + 8 * part;
                    const long long k = frag_d ? uh_wfrag_index<2>(i, td, o, Cout, uh_wfrag_mode(2, Cin)) : ((long long)i * 9 + td) * Cout + o;
                    *reinterpret_cast<bf16x8*>(wd + base + k) = o8;
                }
                continue;
            }
        }
#pragma unroll
        for (int rr = ty; rr < 32; rr += 8) {
            const int o = o0 + rr, i = i0 + tx;
            float v = 0.f;
            if (o < Cout && i < Cin) {
                v = w[o * sO + i * sI + r * sH + s_ * sW];
                const long long k = frag_f ? uh_wfrag_index<(int)sizeof(T)>(o, t, i, Cin, uh_wfrag_mode(sizeof(T), Cout))
                                           : ((long long)o * 9 + t) * Cin + i;
                wf[base + k] = uh_from_f32<T>(v);
            }
            tile[rr][tx] = v;
        }
        __syncthreads();
        if (wd) {
            const int td = (2 - r) * 3 + (2 - s_);
#pragma unroll
            for (int rr = ty; rr < 32; rr += 8) {
                const int i = i0 + rr, o = o0 + tx;
                if (i < Cin && o < Cout) {
                    const long long k = frag_d ? uh_wfrag_index<(int)sizeof(T)>(i, td, o, Cout, uh_wfrag_mode(sizeof(T), Cin))
                                               : ((long long)i * 9 + td) * Cout + o;
                    wd[base + k] = uh_from_f32<T>(tile[tx][rr]);
                }
            }
        }
    }
}

extern "C" int uh_pack_w3x3_batched(const int64_t* table, int nlayers, int64_t ntiles, void* w_fwd_flat, void* w_dgrad_flat,
                                    int dt, uh_stream stream) {
    UH_REQUIRE(table && w_fwd_flat && nlayers > 0 && ntiles > 0, "uh_pack_w3x3_batched: bad arguments");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16, "uh_pack_w3x3_batched: bad dtype %d", dt);
    int64_t g = ntiles;
    if (g > 256 * 64) g = 256 * 64;
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_BF16)
        hipLaunchKernelGGL(pack_w3x3_batched_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const long long*)table,
                           nlayers, (long long)ntiles, (bf16_t*)w_fwd_flat, (bf16_t*)w_dgrad_flat);
    else
        hipLaunchKernelGGL(pack_w3x3_batched_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const long long*)table,
                           nlayers, (long long)ntiles, (float*)w_fwd_flat, (float*)w_dgrad_flat);
    UH_CHECK_LAUNCH("uh_pack_w3x3_batched");
    return UH_OK;
}

extern "C" int uh_unpack_dw3x3(const float* dw_krsc, float* dw, int64_t sO, int64_t sI, int64_t sH, int64_t sW,
                               int Cout, int Cin, uh_stream stream) {
    UH_REQUIRE(dw_krsc && dw, "uh_unpack_dw3x3: null pointer");
    int64_t total = (int64_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(unpack_dw3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dw_krsc, dw, sO, sI, sH, sW, Cout, Cin);
    UH_CHECK_LAUNCH("uh_unpack_dw3x3");
    return UH_OK;
}

// =====================================================================================
// forward, MFMA implicit GEMM
// =====================================================================================
constexpr int TILE = 16;                 // 16x16 output pixels per workgroup
constexpr int HALO_W = TILE + 2;
constexpr int HALO_PIX = HALO_W * HALO_W;   // 324
constexpr int PSTR = 80;                 // LDS bytes per halo pixel: 64 B of channels + 16 B pad
                                         // (5 x 16 B: consecutive pixels rotate through all 16-B slots)
constexpr int HALO_BYTES = HALO_PIX * PSTR;  // 25920

template <typename T, int NB>
__global__ __launch_bounds__(256, 2) void conv3x3_fwd_mfma(
    const T* __restrict__ x0, int C0, int ld0, const T* __restrict__ x1, int C1, int ld1,
    const T* __restrict__ w, T* __restrict__ y, int ldy, int Cout, float* __restrict__ stats,
    int B, int H, int W, int tilesX, int tilesY) {
    constexpr int ES = sizeof(T);
    constexpr int CK = 64 / ES;        // channels per K-chunk
    constexpr int VEC = 16 / ES;       // channels per 16-byte piece
    constexpr int BN = NB * 32;        // output channels per workgroup (2 waves along N)
    constexpr int NPIECE = HALO_PIX * 4;                // 1296 16-byte pieces per chunk
    constexpr int NLOAD = (NPIECE + 255) / 256;         // 6

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HALO_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lx = lane & 15, kg = lane >> 4;

    int t = blockIdx.x;
    const int txt = t % tilesX; t /= tilesX;
    const int tyt = t % tilesY;
    const int b = t / tilesY;
    const int y0 = tyt * TILE, x0p = txt * TILE;
    const int co_blk = blockIdx.y * BN;
    const int co_base = co_blk + wn * (NB * 16);
    const int Cin = C0 + C1;
    const int nchunk = Cin / CK;

    // ---- staging bookkeeping: which halo pixel / 16-B part each of my pieces is
    int pix_idx[NLOAD];   // global pixel index or -1 (zero padding / beyond the piece list)
    int lds_off[NLOAD];
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
        int idx = tid + k * 256;
        int q = idx >> 2, part = idx & 3;
        int hy = q / HALO_W, hx = q - hy * HALO_W;
        int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
        bool ok = (idx < NPIECE) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        pix_idx[k] = ok ? ((b * H + gy) * W + gx) : -1;
        lds_off[k] = (idx < NPIECE) ? (q * PSTR + part * 16) : -1;
    }
    const int part_c = (tid & 3) * VEC;    // my channel offset inside the chunk (idx & 3 == tid & 3)

    u32x4 stage[NLOAD];
    auto stage_load = [&](int c) {
        int cc = c * CK;
        const T* src; int ld;
        if (cc < C0) { src = x0 + cc; ld = ld0; } else { src = x1 + (cc - C0); ld = ld1; }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (pix_idx[k] >= 0) v = *reinterpret_cast<const u32x4*>(src + (int64_t)pix_idx[k] * ld + part_c);
            stage[k] = v;
        }
    };
    auto stage_store = [&](int bufi) {
        unsigned char* buf = lds + bufi * HALO_BYTES;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k)
            if (lds_off[k] >= 0) *reinterpret_cast<u32x4*>(buf + lds_off[k]) = stage[k];
    };

    f32x4 acc[8][NB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // filter fragment base for this lane: row (co_base + nb*16 + lx), 16-B part kg
    const T* wl = w + (int64_t)(co_base + lx) * 9 * Cin + kg * VEC;
    const int64_t wnb_stride = (int64_t)16 * 9 * Cin;

    stage_load(0);
    stage_store(0);
    __syncthreads();

    // Filter fragments are software-pipelined one tap ahead in registers (global -> VGPR, L2 resident): the
    // loads of tap t+1 are issued before the 8*NB MFMAs of tap t, so their latency hides under ~0.5k MFMA cycles.
    u32x4 wcur[NB], wnxt[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) wcur[n] = *reinterpret_cast<const u32x4*>(wl + n * wnb_stride);

    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) stage_load(c + 1);
        const unsigned char* buf = lds + (c & 1) * HALO_BYTES;
        const T* wc = wl + c * CK;
        const T* wc_next = wl + ((c + 1 < nchunk) ? (c + 1) : c) * CK;      // clamped: the last prefetch is unused
        const unsigned char* xrow = buf + ((wm * 8) * HALO_W + lx) * PSTR + kg * 16;
#pragma unroll 1
        for (int s = 0; s < 3; ++s) {
            u32x4 xf[10];
#pragma unroll
            for (int k = 0; k < 10; ++k)
                xf[k] = *reinterpret_cast<const u32x4*>(xrow + (k * HALO_W + s) * PSTR);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // next tap in issue order: (s, r+1) -> (s+1, 0) -> next chunk (0, 0)
                const T* wn = (r < 2) ? (wc + ((r + 1) * 3 + s) * Cin)
                                      : ((s < 2) ? (wc + (s + 1) * Cin) : wc_next);
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    wnxt[n] = *reinterpret_cast<const u32x4*>(wn + n * wnb_stride);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE this tap's MFMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        if constexpr (ES == 2) {
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wcur[n]), __builtin_bit_cast(bf16x8, xf[i + r]), acc[i][n], 0, 0, 0);
                        } else {
                            f32x4 a = __builtin_bit_cast(f32x4, wcur[n]);
                            f32x4 bb = __builtin_bit_cast(f32x4, xf[i + r]);
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bb[q], acc[i][n], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < NB; ++n) wcur[n] = wnxt[n];
            }
        }
        if (c + 1 < nchunk) stage_store((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: acc[i][n][j] = y[pixel (row wm*8+i, col lx)][channel co_base + n*16 + kg*4 + j]
    // BatchNorm statistics per tile and channel as (mean, M2 = sum (v - mean)^2) of the STORED values:
    // two passes over the accumulator registers, combined across tiles with Chan's formula in
    // uh_bn_finalize (sum / sum-of-squares would cancel catastrophically when mean^2 >> var).
    float ssum[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) ssum[n][j] = 0.f;
    const int gx = x0p + lx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gy = y0 + wm * 8 + i;
        const bool ok = (gy < H) && (gx < W);
        T* yp = y + (int64_t)((b * H + gy) * W + gx) * ldy + co_base + kg * 4;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][n][j] = uh_round_as<T>(acc[i][n][j]);
                ssum[n][j] += ok ? acc[i][n][j] : 0.f;
            }
            if (ok) {
                if constexpr (ES == 2) {
                    bf16x4 o = {(bf16_t)acc[i][n][0], (bf16_t)acc[i][n][1], (bf16_t)acc[i][n][2], (bf16_t)acc[i][n][3]};
                    *reinterpret_cast<bf16x4*>(yp + n * 16) = o;
                } else {
                    *reinterpret_cast<f32x4*>(yp + n * 16) = acc[i][n];
                }
            }
        }
    }
    if (stats) {
        float* red = reinterpret_cast<float*>(lds);   // [2 wm][BN] then [BN] means; main loop ended with a barrier
        const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
        const float inv_cnt = 1.f / (float)(vy * vx);
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) ssum[n][j] += __shfl_xor(ssum[n][j], o, 64);
        if (lx == 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wm * BN + wn * (NB * 16) + n * 16 + kg * 4 + j] = ssum[n][j];
        }
        __syncthreads();
        if (tid < BN) red[2 * BN + tid] = (red[tid] + red[BN + tid]) * inv_cnt;     // tile mean
        __syncthreads();
        float sm2[NB][4];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mu = red[2 * BN + wn * (NB * 16) + n * 16 + kg * 4 + j];
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = (y0 + wm * 8 + i < H) && (gx < W);
                    float d = acc[i][n][j] - mu;
                    a += ok ? d * d : 0.f;
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                sm2[n][j] = a;
            }
        __syncthreads();
        if (lx == 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wm * BN + wn * (NB * 16) + n * 16 + kg * 4 + j] = sm2[n][j];
        }
        __syncthreads();
        if (tid < BN) {
            stats[((int64_t)blockIdx.x * 2 + 0) * Cout + co_blk + tid] = red[2 * BN + tid];
            stats[((int64_t)blockIdx.x * 2 + 1) * Cout + co_blk + tid] = red[tid] + red[BN + tid];
        }
        if (blockIdx.y == 0 && tid == 0) stats[(int64_t)gridDim.x * 2 * Cout + blockIdx.x] = (float)(vy * vx);
    }
}

// =====================================================================================
// forward, MFMA implicit GEMM, v2: the hot variant.
//   * wave w of the workgroup owns ALL 256 pixels of the tile x its own NBW*16 output channels, so filter
//     fragments (global -> VGPR, 16 B per lane) are loaded by exactly one wave each: no redundant L2 traffic
//     (v1's 2x2 wave grid re-read every filter element twice and ran into the L2 bandwidth roof);
//   * the 18x18 halo tile goes global -> LDS by DMA (buffer_load ... lds, zero VGPRs, out-of-image pixels read
//     as zeros through the descriptor's range check), double buffered, 64-byte pixel pitch with an XOR swizzle
//     of the four 16-byte parts applied on the SOURCE address;
//   * a halo row fragment is read from LDS once per column shift and used by the three row taps (rolling
//     window of 3 rows); the filter fragments of the next column shift are prefetched under the current one;
//   * BatchNorm (mean, M2) per tile and channel come straight from the accumulator registers + 4 shuffles.
// =====================================================================================
// Diagnostic builds (scratch/r5_mklib.py NAME -DUH_ABL_X=1, scratch/r5_ablate.sh): each switch removes ONE class of work from the
// forward / backward-data kernel's tile loop -- results are garbage, launch times say what that work costs where it sits.
//   UH_ABL_NOW     no filter-fragment loads inside the chunk loop       UH_ABL_NODMA   no halo DMA inside the loop
//   UH_ABL_NOLDS   no pixel-fragment LDS reads (opaque stale registers) UH_ABL_NOMFMA  no MFMAs
//   UH_ABL_NOSTORE no output stores                                     UH_ABL_NOSTATS no BatchNorm statistics
#ifndef UH_ABL_NOW
#define UH_ABL_NOW 0
#endif
#ifndef UH_ABL_NODMA
#define UH_ABL_NODMA 0
#endif
#ifndef UH_ABL_NOLDS
#define UH_ABL_NOLDS 0
#endif
#ifndef UH_ABL_NOMFMA
#define UH_ABL_NOMFMA 0
#endif
#ifndef UH_ABL_NOSTORE
#define UH_ABL_NOSTORE 0
#endif
#ifndef UH_ABL_NOSTATS
#define UH_ABL_NOSTATS 0
#endif
constexpr int HALO2_BYTES = HALO_PIX * 64;   // 20736
constexpr int HALO2_STRIDE = 6 * 4096;       // LDS bytes per halo buffer: six 4 KiB DMA rounds of 256 threads (the tail is padding)
constexpr unsigned OOB_OFFSET = 0xF0000000u;

// One LDS-DMA piece (buffer_load_dwordx4 ... lds: 16 bytes per lane, LDS address = M0 + 16 * lane) as inline asm: the
// compiler does not see a memory operation, so it neither counts it in its own vmcnt bookkeeping nor drains it in front of
// the next ds_read (which it does for the builtin: every LDS read after an LDS-DMA issue waits for that DMA).  The waits are
// placed by hand (see the main loop).  desc = the four descriptor words in SGPRs.
__device__ __forceinline__ void uh_dma16(u32x4 desc, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory", "m0");
}
// 16-byte / 8-byte loads into registers, invisible to the compiler's wait insertion as well: the destination is valid only
// after the hand-placed s_waitcnt that covers it.  "+v": the variable keeps its register across the loop.
__device__ __forceinline__ void uh_ld16_async(u32x4& dst, u32x4 desc, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ void uh_ld8_async(u32x2& dst, u32x4 desc, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 uh_desc_words(const void* p, unsigned bytes) {
    const uint64_t a = (uint64_t)p;
    return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
#define UH_WAIT_VM(n) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// XOR applied to the 16-byte part index of a halo pixel in column hx: with part' = part ^ (2 * ((hx >> 2) & 1)) the 16 lanes of
// every ds_read_b128 lane group (8 lanes of part k, 8 of part k^1, 16 consecutive pixels at ANY start) hit 16
// distinct 16-byte slots of the 256-byte bank row -> conflict-free fragment reads for all nine taps.
__device__ __forceinline__ int halo_swz(int hx) { return ((hx >> 2) & 1) << 1; }   // hx = halo COLUMN (0..17)

// SPLIT (T = float only, "bf16x3"): fp32 tensors, but the products run on the bf16 matrix pipe, 16x faster than the fp32
// one on this chip: x = xh + xl and w = wh + wl with bf16 halves (filters pre-split by uh_pack_w3x3, pixels split in
// registers as their fragment arrives), and  w*x ~= wh*xh + wh*xl + wl*xh  (the dropped wl*xl term and the second
// roundings are ~2^-17 relative: 1e-5, against a 1e-3 parity bar), accumulated in fp32 by v_mfma_f32_16x16x16_bf16.
// WRES (bf16, NBW = 1, exactly two K-chunks = 64 input channels): the wave's whole filter slice (18 fragments = 72 VGPRs)
// is loaded ONCE and stays in registers for every tile of the persistent workgroup.  The 64 -> 64 layers at 512 x 512 have
// only 2 chunks per tile, so every 16x16 tile used to re-read the whole 74 KB filter through L1 (590 MB per launch, more
// than the activations; TA 82 % busy): with it resident the main loop's only vector-memory traffic is the halo DMA.
// PRE (bf16, source 0 only): x0 is the RAW output of the previous conv and the BatchNorm + ReLU that follows it
// (unet_parts.py:16-17) is applied here, by the consumer: z = max(y * scale + shift, 0) per channel, rounded to bf16 exactly as
// uh_bn_relu_apply would have stored it -- so the activation between the two convs of a DoubleConv never exists in HBM.  The
// halo tile still travels HBM -> LDS by DMA; once a wave's own pieces have landed (its covering vmcnt) every thread rewrites
// the pieces IT issued in place (ds_read_b128 -> 8 x fma / max -> ds_write_b128) in front of the chunk's barrier.  Pieces
// outside the image stay zero (the padding of the ACTIVATION is zero, not max(shift, 0)).  The 2 * C0 coefficients sit in LDS.
// BSUM (bf16 backward-data of the SECOND conv of a DoubleConv): the tensor this launch produces is dz, the gradient of the
// activation z = ReLU(BatchNorm(q)) between the two convs (q = bs_y: the first conv's raw output, same shape as this launch's
// output).  The two per-channel sums BatchNorm's backward starts with (unet_parts.py:16-17 differentiated)
//     sum_p g,   sum_p g * (q - mean) * rstd        with g = dz where q * scale + shift > 0, else 0
// are formed in this kernel's epilogue from the accumulators (rounded to bf16 as stored) and one read of q, like the forward's
// statistics: one partial row per workgroup, [row][2][Cout] -- the layout uh_bn_bwd_finalize / uh_bn_relu_bwd_apply take.  The
// uh_bn_relu_bwd_reduce pass (a read of dz and of q) is not launched for that layer.
#ifndef UH_BUILD_PRE
#define UH_BUILD_PRE 0        // build.py: UH_BUILD_PRE=1 compiles the PRE instantiations (A/B and tests/test_gpu_pre_fusion.py)
#endif
constexpr int PRE_MAX_C = 512;
// KS = 2 (bf16, NBW = 1, exactly ONE tile per workgroup): the deep layers of a small batch -- 512 channels at 32 x 32 -- have
// fewer (tile, channel slab) pairs than the chip has CUs and 16+ K-chunks per tile: one 4-wave workgroup per CU, i.e. ONE wave
// per SIMD working through a serial chain of chunks with nothing to hide its LDS / MFMA latencies behind (40 us for 19 GFLOP
// at batch 4).  With KS = 2 the workgroup has EIGHT waves: waves 0-3 contract the first half of the K-chunks, waves 4-7 the
// second half (own halo double buffer each, same hand-counted pipeline), the two partial accumulator sets meet in LDS once
// at the end (the halo buffers are free by then: nothing follows the only tile) and waves 0-3 run the epilogue unchanged --
// two waves per SIMD, half the chain each.  Sums are added in a fixed order (first half + second half): deterministic.
template <typename T, int NBW, bool SPLIT = false, bool WRES = false, bool PRE = false, bool BSUM = false, int KS = 1>
// (three workgroups per CU -- 168 registers -- for the 16-channel-per-wave form; its BSUM instantiation needs more than that for
// the epilogue's batches and spilled 42 instructions per tile at 168: two per CU, like the other wide-register forms)
// (round 5: the register-resident-filter form does not fit three per CU either -- at 168 registers hipcc 7.2 spills 36 instructions into its
// MFMA stream and parks in-flight filter destinations in scratch; the ISA lint refuses the build)
__global__ __launch_bounds__(256 * KS, ((NBW == 1 && !WRES && !BSUM && KS == 1) ? 3 : 2)) void conv3x3_fwd_mfma_v2(
    const T* __restrict__ x0, int C0, int ld0, const T* __restrict__ x1, int C1, int ld1,
    const T* __restrict__ w, T* __restrict__ y, int ldy, int Cout, float* __restrict__ stats,
    int B, int H, int W, int tilesX, int tilesY, unsigned x0_bytes, unsigned x1_bytes, unsigned y_bytes,
    const float* __restrict__ ep_scale, const float* __restrict__ ep_shift, int C0v, int C1v, int Coutv, int wfrag,
    const float* __restrict__ pre_scale = nullptr, const float* __restrict__ pre_shift = nullptr,
    const T* __restrict__ bs_y = nullptr, int bs_ld = 0, unsigned bs_bytes = 0, const float* __restrict__ bs_coef = nullptr) {
    static_assert(!PRE || (sizeof(T) == 2 && !SPLIT), "PRE is the bf16 training path");
    static_assert(!BSUM || (sizeof(T) == 2 && !SPLIT && !PRE), "BSUM is the bf16 backward-data path");
    static_assert(KS == 1 || (KS == 2 && sizeof(T) == 2 && NBW == 1 && !SPLIT && !WRES && !PRE), "KS = 2 is a bf16 NBW = 1 form");
    if constexpr (BSUM) {
        // a plain single-source call (the host checks it): folding the second source, the narrow-tensor counts and the inference
        // epilogue away frees the scalar registers the extra arguments take -- the scalar file is full (the base kernel already
        // parks scalars in vector lanes), and every parked scalar costs the MFMA loop a vector register
        x1 = nullptr; C1 = 0; ld1 = 0; x1_bytes = 0;
        C0v = C0; C1v = 0; Coutv = Cout;
        ep_scale = nullptr; ep_shift = nullptr;
        bs_ld = ldy; bs_bytes = y_bytes;            // q has the geometry of the tensor this launch writes
    }
    // C0 / C1 / Cout are the channel counts the filter pack is laid out for (multiples of a chunk / of 64); C0v / C1v /
    // Coutv (<=) are the channels that exist in memory ("narrow" tensors of the small-width nets): input channels
    // beyond them are fetched as zeros by the DMA, output channels beyond Coutv are computed (zero filters) but not stored.
    constexpr int ES = sizeof(T);
    constexpr int CK = 64 / ES;
    constexpr int VEC = 16 / ES;
    constexpr int BN = NBW * 64;
    constexpr int NPIECE = HALO_PIX * 4;     // 1296
    constexpr int NLOAD = 6;
#ifndef UH_FWD_PF
#define UH_FWD_PF 2
#endif
    // LDS fragment prefetch distance in halo rows.  Two rows ahead (192 cycles of MFMA issue at NBW = 2 instead of 96): +0.3 % on
    // the train step in three interleaved rounds (856.5 -> 858.9 images/s, round 4); the register-resident-filter form has no four
    // registers to spare for it (it spills inside its MFMA stream at distance 2) and keeps one row.
    constexpr int PF = WRES ? 1 : UH_FWD_PF;

#ifndef UH_WRES_TRI
#define UH_WRES_TRI 1
#endif
    // TRI (the register-resident-filter form): THREE halo buffers, the DMA runs two chunks = one whole tile ahead.  With 16 channels
    // per wave a chunk is 144 MFMAs, and the two column shifts a double-buffered DMA has to land under (96 MFMAs, ~1 us beside the
    // SIMD's other wave) are shorter than an HBM round trip under load: the 64 -> 64 layers at 512 x 512 waited at every chunk fence
    // (scratch/r4_bsum_bench.sh with the cached-input variant: 141 -> 115 us per launch).  72 KB per workgroup, two per CU.
    constexpr bool TRI = WRES && !PRE && UH_WRES_TRI;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[(TRI ? 3 : KS * 2) * HALO2_STRIDE];
    // BatchNorm statistics of this workgroup's channels over ALL the tiles it processes, as pivot-shifted sums
    // S1 = sum (v - p), S2 = sum (v - p)^2 with p = one stored value of the channel (so that |mean - p| ~ std and the
    // final M2 = S2 - S1^2 / n does not cancel): one partial row per WORKGROUP (<= 768 rows), written once at the end.
    __shared__ float wg_sum[3][BN];          // [0] = S1, [1] = S2, [2] = pivot; slot = channel - co_blk
    __shared__ __attribute__((aligned(16))) float pre_tab[PRE ? 2 * PRE_MAX_C : 4];     // PRE: (scale, shift) pairs of source 0
    __shared__ __attribute__((aligned(16))) float bs_tab[BSUM ? 4 * BN : 4];            // BSUM: [scale | shift | mean | rstd][slot]
    float n_run = 0.f;

    // KS = 2: `tid` / `wave` are the indices inside the thread's K-group (0 .. 255 / 0 .. 3): DMA slots, filter rows and
    // channel ownership are per group; `grp` picks the group's half of the chunks and its own pair of halo buffers
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = wave_all & 3;
    const int grp = KS == 2 ? (wave_all >> 2) : 0;
    unsigned char* const lds = lds_all + grp * (2 * HALO2_STRIDE);
    const int lx = lane & 15, kg = lane >> 4;
    // XCD-aware block mapping (1-D grid of nlanes * nslab workgroups, nlanes % 8 == 0 or nslab == 1): blocks are dealt
    // round-robin to the 8 XCDs, so id % 8 labels the XCD; all channel slabs of one tile lane get the SAME label and
    // adjacent dispatch slots -> the slabs re-read the same input tile from that XCD's L2 instead of from HBM.
    const int nslab = Cout / BN;
    const int nlanes = gridDim.x / nslab;
    int tile_lane, slab;
#ifndef UH_XCD_BLOCK
#define UH_XCD_BLOCK 0      // experiment (round 5): 1 = an XCD owns a CONTIGUOUS range of tile lanes (horizontally adjacent tiles share its L2), 0 = lanes dealt round-robin
#endif
    if ((nlanes & 7) == 0) {
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
        slab = jj % nslab;
        tile_lane = UH_XCD_BLOCK ? xcd * (nlanes >> 3) + jj / nslab : (jj / nslab) * 8 + xcd;
    } else {
        slab = blockIdx.x / nlanes;
        tile_lane = blockIdx.x - slab * nlanes;
    }
    const int co_blk = slab * BN;
    const int Cin = C0 + C1;
    // K-chunks that hold at least one stored channel: chunks beyond a source's valid count (narrow tensors) would be all
    // zeros, so they are neither fetched nor multiplied.  Ordinary tensors: nch0 + nch1 == Cin / CK, chunk_of(v) == v.
    const int nch0 = (C0v + CK - 1) / CK, nch1 = (C1v + CK - 1) / CK;
    const int nchunk = nch0 + nch1;
    const int skip1 = C0 / CK - nch0;             // chunk index jump between the last valid chunk of x0 and the first of x1
    auto chunk_of = [&](int v) -> int { return v < nch0 ? v : v + skip1; };
    const int ntile = B * tilesX * tilesY;

    // ---- DMA bookkeeping: LDS slot p = tid + k*256 holds (halo pixel q = p >> 2, part' = p & 3) = source part part' ^ f(q).
    // rel0[k] = byte offset of slot k's source piece from the halo tile's top-left pixel at pixel pitch ld0 -- a constant of
    // the thread.  For a tile whose 18x18 halo lies inside the image (and a chunk whose 64 bytes exist in memory at pitch
    // ld0: everything but the narrow tensors of the small nets) the address of a piece is rel0[k] + one wave-uniform base:
    // ONE vector add per piece.  Tiles on the image border add the range tests; anything else takes the generic path.
    const u32x4 rs0 = uh_desc_words(x0, x0_bytes);
    const u32x4 rs1 = uh_desc_words(x1 ? x1 : x0, x1 ? x1_bytes : x0_bytes);
    __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // (BSUM with 128 accumulators: six registers too many -- the offsets are rebuilt at every issue, ~30 VALU operations per chunk
    // of 288 MFMAs, from an opaque copy of the thread id so that they are not hoisted back out of the loop)
    constexpr bool REL_LIVE = !(BSUM && NBW == 2);
    auto rel_of = [&](int t, int k) -> int {
        const int p = t + k * 256, q = p >> 2;
        const int hy = (q * 3641) >> 16, hx = q - hy * HALO_W;          // q / 18 for q < 324
        return (hy * W + hx) * ld0 * ES + (((p & 3) ^ halo_swz(hx)) << 4);
    };
    int rel0[NLOAD];
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) rel0[k] = REL_LIVE ? rel_of(tid, k) : 0;
    bool abl_first_dma = true; (void)abl_first_dma;
    // tile the NEXT DMA reads from (scalars): image, top-left pixel, "halo inside the image"
    int d_b = 0, d_y0 = 0, d_x0 = 0;
    bool d_in = false;
    auto dma_tile = [&](int tile) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        d_b = t / tilesY;
        d_y0 = tyt * TILE; d_x0 = txt * TILE;
        d_in = d_y0 >= 1 && d_y0 + TILE + 1 <= H && d_x0 >= 1 && d_x0 + TILE + 1 <= W;
    };
    // EVERY thread issues exactly NLOAD pieces per call (the hand-placed vmcnt waits count instructions per wave): slots
    // past the halo tile (and `live == false`: nothing left to fetch) read out of range, i.e. write zeros into the padding
    // of the buffer / into a buffer nobody reads.
    auto dma_chunk = [&](int c, int bufi, bool live) {
#if UH_ABL_NODMA
        if (!abl_first_dma) return;
#endif
        const int cc = c * CK;
        const bool first = cc < C0;
        const int ld = first ? ld0 : ld1;
        const int soff = (first ? cc : cc - C0) * ES;
        const int cleft = first ? (C0v - cc) : (C1v - (cc - C0));      // channels of this chunk that exist in memory
        const unsigned dst = lds_base + bufi * HALO2_STRIDE + wave * 1024;
        const int base = ((d_b * H + d_y0 - 1) * W + d_x0 - 1) * ld0 * ES;     // may be negative: only used where in range
        unsigned voff[NLOAD];
        if (!live) {
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) voff[k] = OOB_OFFSET;
        } else if (ld == ld0 && cleft >= CK) {
            if (d_in) {
                int tid_r = tid;
                if constexpr (!REL_LIVE) asm volatile("" : "+v"(tid_r));
#pragma unroll
                for (int k = 0; k < NLOAD; ++k) voff[k] = (unsigned)((REL_LIVE ? rel0[k] : rel_of(tid_r, k)) + base);
                if (tid + (NLOAD - 1) * 256 >= NPIECE) voff[NLOAD - 1] = OOB_OFFSET;
            } else {
                // (the opaque copy of tid keeps the per-slot coordinates from being hoisted out of the tile loop into 12
                // long-lived registers: the main loop runs at the register limit)
                int tid_o = tid;
                asm volatile("" : "+v"(tid_o));
#pragma unroll
                for (int k = 0; k < NLOAD; ++k) {
                    const int q = (tid_o + k * 256) >> 2;
                    const int hy = (q * 3641) >> 16, hx = q - hy * HALO_W;
                    const bool ok = (unsigned)(d_y0 - 1 + hy) < (unsigned)H && (unsigned)(d_x0 - 1 + hx) < (unsigned)W && q < HALO_PIX;
                    voff[k] = ok ? (unsigned)((REL_LIVE ? rel0[k] : rel_of(tid_o, k)) + base) : OOB_OFFSET;
                }
            }
        } else {
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int p = tid_o + k * 256, q = p >> 2;
                const int hy = (q * 3641) >> 16, hx = q - hy * HALO_W;
                const int gy = d_y0 - 1 + hy, gx = d_x0 - 1 + hx;
                const int part = (p & 3) ^ halo_swz(hx);
                const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && part * VEC < cleft && q < HALO_PIX;
                voff[k] = ok ? (unsigned)(((d_b * H + gy) * W + gx) * ld * ES + part * 16) : OOB_OFFSET;
            }
        }
        const u32x4 rs = first ? rs0 : rs1;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) uh_dma16(rs, dst + k * 4096, voff[k], soff);
    };

    // Which output channel an MFMA row holds.  The 16 rows of MFMA n land in the accumulator as 4 consecutive rows per lane
    // group kg (acc[i][n][j] = row kg*4 + j, pixel lx).  The filter rows are handed to the MFMAs in a PERMUTED order so that
    // what a lane holds for one pixel is one 16-byte piece of the NHWC output and the epilogue stores straight from the
    // registers (no LDS bounce, no barrier).  ch(grp, n, j) = channel - co_blk of row grp*4 + j of the wave's MFMA n:
    //   bf16, NBW = 2:                  wave*32 + grp*8 + n*4 + j: a lane's (n, j) values are 8 consecutive channels ->
    //                                   one 16-byte store per lane and pixel row, 64 contiguous bytes per pixel;
    //   bf16, NBW = 1, Cout % 128 != 0: wave*16 + cg(grp)*4 + j with cg = {0, 2, 1, 3}: lanes l and l + 32 hold adjacent
    //                                   4-channel groups, so ONE v_permlane32_swap per dword pairs two pixel rows into
    //                                   16-byte stores;
    //   bf16, NBW = 1, Cout % 128 == 0: (small feature maps of wide layers) the row order of the NBW = 2 case, the wave
    //                                   holding one of its two MFMAs: (wave>>1)*32 + grp*8 + (wave&1)*4 + j, 8-byte stores --
    //                                   so that the row order, and with it the fragment-major filter pack (uh_wfrag_index),
    //                                   depends on the dtype and Cout only, not on which instantiation a launch picks;
    //   fp32:                           the MFMA's own order wave*16*NBW + n*16 + grp*4 + j (16 bytes per lane already).
    constexpr bool PERM2 = (ES == 2 && NBW >= 2);
    const bool half32 = (ES == 2 && NBW == 1) && (Cout % 128 == 0);
    const bool perm1 = (ES == 2 && NBW == 1) && !half32;
    auto ch = [&](int grp, int n, int j) -> int {       // grp = row >> 2 (= kg for accumulators), j = row & 3
        if constexpr (PERM2) return wave * (16 * NBW) + (n >> 1) * 32 + grp * 8 + (n & 1) * 4 + j;
        else if constexpr (ES == 2) return half32 ? ((wave >> 1) * 32 + grp * 8 + (wave & 1) * 4 + j)
                                                  : (wave * 16 + ((((grp & 1) << 1) | (grp >> 1)) << 2) + j);
        else return wave * (16 * NBW) + n * 16 + grp * 4 + j;
    };
    const int wrow0 = co_blk + ch(lx >> 2, 0, lx & 3);
    auto wn_off = [&](int n) -> int64_t { return (int64_t)(PERM2 ? ((n >> 1) * 32 + (n & 1) * 4) : n * 16) * 9 * Cin; };   // filter rows of MFMA n behind MFMA 0's (KRSC packs)
    // Filter fragments come through a buffer descriptor: ONE per-lane byte offset (the lane's filter row and 16-byte part)
    // plus a wave-uniform byte offset in an SGPR (chunk, tap, n) -- no 64-bit address arithmetic in vector registers -- as
    // asynchronous loads whose waits are placed by hand (uh_ld16_async).
    // SPLIT: `w` holds two bf16 arrays [Cout][9][Cin] (hi then lo); a fragment = 4 hi values + 4 lo values (two 8-byte loads).
    const int64_t wlo_off = (int64_t)Cout * 9 * Cin;
    const u32x4 rsw = uh_desc_words(w, (unsigned)((int64_t)Cout * 9 * Cin * (SPLIT ? 4 : ES)));
    const unsigned wvoff = wfrag ? (unsigned)(lane * 16)
                                 : (unsigned)(((int64_t)wrow0 * 9 * Cin + kg * (SPLIT ? 4 : VEC)) * (SPLIT ? 2 : ES));
    // fragment-major packs (uh_wfrag_index): block ((g * 9 + tap) * NCH + chunk) of 1 KiB, g = the wave's 16-row group
    const int wg0 = (co_blk >> 4) + wave * NBW, nch_all = Cin / CK;
    struct WFrag { u32x4 v; u32x2 hi, lo; };     // non-SPLIT: v = the fragment; SPLIT: hi / lo = the bf16 halves of 4 values
    constexpr int NWLOAD = 3 * NBW * (SPLIT ? 2 : 1);      // load instructions per column shift (one weight set)
    auto wfrag_async = [&](WFrag& dst, int64_t off, int n, int tap, int chunk) {    // off = wn_off(n) + tap * Cin + chunk offset (elements)
        // ONE asm statement per destination: with a load in each arm of a branch the compiler merges the two "results"
        // with register copies behind the branch -- executed before the data has arrived (asynchronous destination)
        if constexpr (SPLIT) {
            uh_ld8_async(dst.hi, rsw, wvoff, (int)(off * 2));
            uh_ld8_async(dst.lo, rsw, wvoff, (int)((off + wlo_off) * 2));
        } else {
            const int soff = wfrag ? ((((wg0 + n) * 9 + tap) * nch_all + chunk) << 10) : (int)(off * ES);
            uh_ld16_async(dst.v, rsw, wvoff, soff);
        }
    };
    // fp32 pixel fragment (4 channels) -> {hi pair, hi pair, lo pair, lo pair} as bf16
    auto split4 = [&](u32x4& f) {
        const f32x4 x = __builtin_bit_cast(f32x4, f);
        bf16_t h[4];
        unsigned lo[2], hi[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = (bf16_t)x[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bf16_t l0 = (bf16_t)(x[2 * q] - (float)h[2 * q]), l1 = (bf16_t)(x[2 * q + 1] - (float)h[2 * q + 1]);
            hi[q] = (unsigned)__builtin_bit_cast(unsigned short, h[2 * q]) | ((unsigned)__builtin_bit_cast(unsigned short, h[2 * q + 1]) << 16);
            lo[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        f = u32x4{hi[0], hi[1], lo[0], lo[1]};
    };

    // PRE: BatchNorm + ReLU of the producer, applied to the pieces THIS thread's DMA has just landed in buffer `bufi_t`
    // (chunk `c`, the tile the scalars d_* describe).  Called behind the wave's vmcnt(0), in front of the chunk's barrier.
    auto pre_transform = [&](int c, int bufi_t) {
        if constexpr (PRE) {
            const int cc = c * CK;
            if (cc >= C0) return;                               // second source (skip concatenation): stored activated
            unsigned char* bufp = lds + bufi_t * HALO2_STRIDE;
            int tid_o = tid;                                     // (opaque: keeps the slot coordinates out of long-lived registers)
            asm volatile("" : "+v"(tid_o));
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const int p = tid_o + k * 256, q = p >> 2;
                const int hy = (q * 3641) >> 16, hx = q - hy * HALO_W;
                bool ok = q < HALO_PIX;
                if (!d_in) ok = ok && (unsigned)(d_y0 - 1 + hy) < (unsigned)H && (unsigned)(d_x0 - 1 + hx) < (unsigned)W;
                if (ok) {
                    const int part = (p & 3) ^ halo_swz(hx);
                    u32x4* slot = reinterpret_cast<u32x4*>(bufp + p * 16);
                    const u32x4 v = *slot;
                    const f32x4* t = reinterpret_cast<const f32x4*>(pre_tab + (cc + part * VEC) * 2);
                    const f32x4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];       // (s0, h0, s1, h1) ...
                    float r[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        r[2 * e] = __uint_as_float(v[e] << 16);
                        r[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
                    }
                    r[0] = uh_relu(fmaf(r[0], t0[0], t0[1])); r[1] = uh_relu(fmaf(r[1], t0[2], t0[3]));
                    r[2] = uh_relu(fmaf(r[2], t1[0], t1[1])); r[3] = uh_relu(fmaf(r[3], t1[2], t1[3]));
                    r[4] = uh_relu(fmaf(r[4], t2[0], t2[1])); r[5] = uh_relu(fmaf(r[5], t2[2], t2[3]));
                    r[6] = uh_relu(fmaf(r[6], t3[0], t3[1])); r[7] = uh_relu(fmaf(r[7], t3[2], t3[3]));
                    const bf16x8 o = {(bf16_t)r[0], (bf16_t)r[1], (bf16_t)r[2], (bf16_t)r[3],
                                      (bf16_t)r[4], (bf16_t)r[5], (bf16_t)r[6], (bf16_t)r[7]};
                    *slot = __builtin_bit_cast(u32x4, o);
                }
            }
        }
    };
    // end of a chunk: the next buffer's DMA (and the asynchronous filter loads) have landed, every wave is done reading the
    // current buffer.  PRE: the wave's own pieces are rewritten between its vmcnt(0) and the barrier.
    auto chunk_fence = [&](int c_next, int bufi_next, bool live_next) {
        if constexpr (TRI) {
            // the six pieces of the chunk AFTER the next one stay in flight (vmcnt retires in order: everything older has landed)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NLOAD) : "memory");
        } else if constexpr (PRE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (live_next) pre_transform(c_next, bufi_next);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // Persistent over tiles.  Every vector-memory operation of the main loop (halo DMA, filter fragments) is issued through
    // inline asm and waited for by hand-counted s_waitcnt: vmcnt retires IN ORDER, so what matters is the ORDER of issue.
    // Per chunk (W* = the NWLOAD fragment loads of one column shift, D = the 6 DMA pieces of the NEXT chunk's halo tile):
    //     issue WB                         | shift 0 (fragments WA, waited for at the end of the previous chunk)
    //     issue WC, D ; wait vmcnt(NWLOAD+6) -> WB landed        | shift 1
    //     issue WA'   ; wait vmcnt(6+NWLOAD) -> WC landed        | shift 2   (D and WA' stay in flight)
    //     wait vmcnt(0), lgkmcnt(0) ; s_barrier                  -> D and WA' landed, the buffer just read is free
    // The DMA has two column shifts of MFMAs (2 x 48 x NBW) to land under; the compiler sees no memory operation in the loop
    // except the LDS fragment reads, so it inserts no wait of its own (for the LDS-DMA builtin it drains the DMA in front of
    // every following ds_read).  The epilogue's stores are ordinary buffer stores issued BEFORE the statistics: the first
    // wait that covers them is the WB wait of the next tile's first chunk, one column shift + the statistics later.
    int tile = tile_lane;
    if (tile >= ntile) return;
    WFrag wA[3][NBW], wB[3][NBW], wC[3][NBW];      // filter fragments of column shift 0 / 1 / 2
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            wA[r][n].v = wB[r][n].v = wC[r][n].v = u32x4{0u, 0u, 0u, 0u};
            wA[r][n].hi = wB[r][n].hi = wC[r][n].hi = wA[r][n].lo = wB[r][n].lo = wC[r][n].lo = u32x2{0u, 0u};
        }
    bool abl_first = true; (void)abl_first;
    auto load_w = [&](WFrag (&dst)[3][NBW], int chunk, int sft) {      // the three row taps of column shift `sft`, chunk `chunk`
#if UH_ABL_NOW
        if (!abl_first) return;
#endif
        const int64_t wsrc = (int64_t)chunk * CK + (int64_t)sft * Cin;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int n = 0; n < NBW; ++n) wfrag_async(dst[r][n], wsrc + wn_off(n) + (int64_t)(r * 3) * Cin, n, r * 3 + sft, chunk);
    };
    int bufi = 0;
    // one column shift: rolling window over the 18 halo rows, row k+1 is fetched from LDS while output row k-2 is multiplied
    f32x4 acc[16][NBW];
    auto mma_shift = [&](const unsigned char* buf, int sft, auto&& wget) {       // wget(r, n): filter fragment of row tap r, MFMA n
        u32x4 xf[18];
        // column-only swizzle: the lane part of the address is the same for all 18 rows (immediate offsets)
        int lq = lane;
        if constexpr (!REL_LIVE) asm volatile("" : "+v"(lq));       // (same register shortage: the three column offsets are not kept either)
        const unsigned char* xcol = buf + ((lq & 15) + sft) * 64 + (((lq >> 4) ^ halo_swz((lq & 15) + sft)) << 4);
#if UH_ABL_NOLDS
        u32x4 abl_x = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        asm volatile("" : "+v"(abl_x));
        auto rd = [&](int k) { u32x4 v = abl_x; asm volatile("" : "+v"(v)); (void)xcol; return v; };
#else
        auto rd = [&](int k) { return *reinterpret_cast<const u32x4*>(xcol + k * (HALO_W * 64)); };
#endif
#pragma unroll
        for (int k = 0; k < 2 + PF; ++k) xf[k] = rd(k);
#pragma unroll
        for (int k = 2; k < 18; ++k) {
            if (k + PF < 18) xf[k + PF] = rd(k + PF);     // PF rows ahead of the row being multiplied
            __builtin_amdgcn_sched_barrier(0);            // the read stays above this row's MFMAs
            if constexpr (SPLIT) {                        // row k's fragment is first used now: split it once
                if (k == 2) { split4(xf[0]); split4(xf[1]); }
                split4(xf[k]);
            }
            const int i = k - 2;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    if constexpr (ES == 2) {
#if UH_ABL_NOMFMA
                        asm volatile("" : "+v"(acc[i][n]) : "v"(xf[i + r]), "v"(wget(r, n).v));
#else
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wget(r, n).v), __builtin_bit_cast(bf16x8, xf[i + r]), acc[i][n], 0, 0, 0);
#endif
                    } else if constexpr (SPLIT) {
                        const s16x4 wh = __builtin_bit_cast(s16x4, wget(r, n).hi);
                        const s16x4 wlo = __builtin_bit_cast(s16x4, wget(r, n).lo);
                        const s16x4 xh = __builtin_bit_cast(s16x4, u32x2{xf[i + r][0], xf[i + r][1]});
                        const s16x4 xlo = __builtin_bit_cast(s16x4, u32x2{xf[i + r][2], xf[i + r][3]});
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo, xh, acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, xlo, acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, xh, acc[i][n], 0, 0, 0);
                    } else {
                        f32x4 a = __builtin_bit_cast(f32x4, wget(r, n).v);
                        f32x4 bb = __builtin_bit_cast(f32x4, xf[i + r]);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qq], bb[qq], acc[i][n], 0, 0, 0);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    auto wsel = [&](WFrag (&wv)[3][NBW]) { return [&wv](int r, int n) -> const WFrag& { return wv[r][n]; }; };
    // WRES: the resident filter, [chunk][tap][n]
    WFrag wres[WRES ? 2 : 1][WRES ? 9 : 1][NBW];
    // this thread's share of the K-chunks: all of them, or (KS = 2) its group's half
    const int v_first = KS == 2 ? grp * (nchunk >> 1) : 0;
    const int v_end = KS == 2 ? v_first + (nchunk >> 1) : nchunk;
    dma_tile(tile);
    dma_chunk(chunk_of(v_first), 0, true);
    if constexpr (WRES) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    wres[c][t][n].v = u32x4{0u, 0u, 0u, 0u};
                    wfrag_async(wres[c][t][n], (int64_t)c * CK + wn_off(n) + (int64_t)t * Cin, n, t, c);
                }
        if constexpr (TRI) dma_chunk(1, 1, true);      // (behind the filter: the first fence leaves exactly these six in flight)
    } else {
        load_w(wA, chunk_of(v_first), 0);
    }
    if constexpr (PRE) {
        // the coefficient table must be complete before the first rewrite: one extra barrier, once per kernel
        for (int i = tid; i < C0; i += 256) { pre_tab[2 * i] = pre_scale[i]; pre_tab[2 * i + 1] = pre_shift[i]; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BSUM) {
        // this workgroup's BN channels of [scale | shift | mean | rstd] (Cout entries each); first read in the first epilogue,
        // behind the barriers of the chunk fences
        for (int i = tid; i < 4 * BN; i += 256) bs_tab[i] = bs_coef[(i / BN) * Cout + co_blk + (i % BN)];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    chunk_fence(chunk_of(v_first), 0, true);
    abl_first = false; abl_first_dma = false;

    for (; tile < ntile; tile += nlanes) {
        if (KS == 2 && tile != tile_lane) break;     // (one tile per workgroup: the host sizes the grid that way)
        const int next_tile = KS == 2 ? ntile : tile + nlanes;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int n = 0; n < NBW; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if constexpr (WRES) {
            // two chunks, filter in registers: the only vector-memory traffic is the halo DMA of what follows
            bool live_next = true;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const unsigned char* buf = lds + bufi * HALO2_STRIDE;
                mma_shift(buf, 0, [&](int r, int n) -> const WFrag& { return wres[v][r * 3 + 0][n]; });
                if constexpr (TRI) {
                    // chunk v of the NEXT tile, into the buffer the chunk in front of this one was read from
                    const int b2 = bufi == 0 ? 2 : bufi - 1;
                    if (v == 0) {
                        live_next = next_tile < ntile;
                        if (live_next) dma_tile(next_tile);
                    }
                    dma_chunk(v, b2, live_next);
                } else if (v == 0) {
                    dma_chunk(1, bufi ^ 1, true);
                } else {
                    live_next = next_tile < ntile;
                    if (live_next) dma_tile(next_tile);
                    dma_chunk(0, bufi ^ 1, live_next);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_shift(buf, 1, [&](int r, int n) -> const WFrag& { return wres[v][r * 3 + 1][n]; });
                mma_shift(buf, 2, [&](int r, int n) -> const WFrag& { return wres[v][r * 3 + 2][n]; });
                chunk_fence(v == 0 ? 1 : 0, TRI ? 0 : (bufi ^ 1), live_next);
                if constexpr (TRI) bufi = bufi == 2 ? 0 : bufi + 1; else bufi ^= 1;
            }
        } else {
#pragma unroll 1
        for (int v = v_first; v < v_end; ++v, bufi ^= 1) {
            const int cur_c = chunk_of(v);
            const int next_c = chunk_of((v + 1 < v_end) ? v + 1 : v_first);   // wraps to the first chunk of the next tile
            const unsigned char* buf = lds + bufi * HALO2_STRIDE;
            load_w(wB, cur_c, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_shift(buf, 0, wsel(wA));
            load_w(wC, cur_c, 2);
            // the halo tile of what follows chunk v: the tile's next chunk, else chunk 0 of the workgroup's next tile, else
            // nothing (six out-of-range pieces keep the instruction count of the waits below)
            bool live_next = true;
            if (v + 1 < v_end) {
                dma_chunk(chunk_of(v + 1), bufi ^ 1, true);
            } else {
                live_next = next_tile < ntile;
                if (live_next) dma_tile(next_tile);
                dma_chunk(chunk_of(v_first), bufi ^ 1, live_next);
            }
            UH_WAIT_VM(NWLOAD + NLOAD);               // WB landed (WC and the DMA stay in flight)
            mma_shift(buf, 1, wsel(wB));
            load_w(wA, next_c, 0);
            UH_WAIT_VM(NLOAD + NWLOAD);               // WC landed (the DMA and WA' stay in flight)
            mma_shift(buf, 2, wsel(wC));
            // DMA and WA' landed; every wave has finished reading this buffer
            chunk_fence(next_c, bufi ^ 1, live_next);
        }
        }

        if constexpr (KS == 2) {
            // The two halves of the contraction meet: the last chunk fence has drained every DMA and every wave is done reading
            // (the host launches one tile per workgroup: no next tile was requested), so the four halo buffers are scratch now.
            // Waves 4-7 park their accumulators lane-major (conflict-free 16-byte writes), waves 0-3 add them in a fixed order.
            f32x4* scratch = reinterpret_cast<f32x4*>(lds_all);
            if (grp == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int n = 0; n < NBW; ++n) scratch[(i * NBW + n) * 256 + tid] = acc[i][n];
            }
            __syncthreads();
            if (grp == 1) return;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int n = 0; n < NBW; ++n) acc[i][n] += scratch[(i * NBW + n) * 256 + tid];
        }
        // ---- epilogue: acc[i][n][j] = y[pixel (row i, col lx)][channel co_blk + ch(kg, n, j)]
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TILE, x0p = txt * TILE;
        const int gx = x0p + lx;
        const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
        const bool full = (vy == TILE) && (vx == TILE);          // wave-uniform: interior tiles take the mask-free path
        // BSUM: the 16 x NBW eight-byte pieces of q this lane's accumulators meet (pixel (row i, column lx), channels ch(kg, n, 0..3))
        // come in batches of eight rows (16 registers), two batches in flight: the next one is always requested before the
        // previous one is summed.  The sums run BEFORE the stores: behind them hipcc kept the packed bf16 halves of the store data
        // alive beside the values the sums need.  The kernel runs at the register limit: all 32 x NBW
        // registers at once spilled the DMA offsets / filter fragments into the MFMA loop (the ISA lint refuses that), and the
        // lane constants of this block are rebuilt per tile from an opaque copy of the lane id instead of living through the loop.
        constexpr int BQB = 2 * NBW;                   // batches: (n, row half)
        u32x2 bqA[BSUM ? 8 : 1], bqB[BSUM ? 8 : 1];
        int lane_o = lane;
        if constexpr (BSUM) asm volatile("" : "+v"(lane_o));
        const int lx_o = lane_o & 15, kg_o = lane_o >> 4;
        auto bq_load = [&](u32x2 (&dst)[BSUM ? 8 : 1], int bi) {
            if constexpr (BSUM) {
                __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)bs_y, 0, (int)bs_bytes, 0x00020000);
                const int n = bi >> 1, i0 = (bi & 1) * 8;
                // every request stays inside the image (row / column clamped: what a clamped request returns meets an accumulator
                // that was set to zero above); the row part of the address is wave-uniform and rides in the scalar offset operand --
                // sixteen per-row vector offsets would live from here through the stores (hipcc shares them: same pitch)
                const int gxc = min(x0p + (lane_o & 15), W - 1);
                const unsigned voff = (unsigned)((gxc * bs_ld + co_blk + ch(lane_o >> 4, n, 0)) * ES);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = min(i0 + i, vy - 1);
                    dst[i] = __builtin_amdgcn_raw_buffer_load_b64(rsq, voff, ((b * H + y0 + r) * W) * bs_ld * ES, 0);
                }
            }
        };
        if (ep_scale) {      // inference: eval-mode BatchNorm (per-channel scale/shift) + ReLU applied to the accumulators
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ep_scale + co_blk + ch(kg, n, 0));
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ep_shift + co_blk + ch(kg, n, 0));
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][n][j] = uh_relu(fmaf(acc[i][n][j], sc[j], sh[j]));
            }
        }
        // round to the stored type once: the statistics are those of the STORED values
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][n][j] = uh_round_as<T>(acc[i][n][j]);
        // (BSUM: the rounded fp32 values REPLACE the accumulators here; left to its own schedule hipcc kept the packed bf16 halves
        // for the stores beside the fp32 originals for the sums -- 128 more registers, spilled)
        if constexpr (BSUM) __builtin_amdgcn_sched_barrier(0);

        if constexpr (BSUM) {
            float* S1 = &wg_sum[0][0];
            float* S2 = &wg_sum[1][0];
            if (tile == tile_lane && lx_o == 0) {        // the workgroup's first tile
#pragma unroll
                for (int n = 0; n < NBW; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { S1[ch(kg_o, n, j)] = 0.f; S2[ch(kg_o, n, j)] = 0.f; }
            }
            if (!full) {
                // border tile: the accumulators of pixels outside the image (never stored) must not reach the sums
                const bool inw = x0p + lx_o < W;
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int n = 0; n < NBW; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][n][j] = (i < vy && inw) ? acc[i][n][j] : 0.f;
            }
            constexpr bool BQ2 = NBW == 1;              // two batches in flight (NBW = 2: one at a time -- no time difference in an A/B of the three schedules, fewest spills)
            if constexpr (BQ2) bq_load(bqA, 0);
            float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};
#ifndef UH_BSUM_PK
#define UH_BSUM_PK 1
#endif
            auto bq_sum = [&](const u32x2 (&src)[8], int bi) {
                const int n = bi >> 1, i0 = (bi & 1) * 8;
                const int c0 = ch(kg_o, n, 0);
#if UH_BSUM_PK
                // two channels at a time as packed fp32 pairs (the two halves of a q dword, accumulators j, j + 1 = an aligned
                // register pair): v_pk_fma / v_pk_add for the mask argument, q - mean and both sums -- 10 VALU instructions per
                // pair and row instead of 14, every channel's operations and their order unchanged (bit-identical sums)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int jp = 0; jp < 4; jp += 2) {
                    const f32x2 sc = *reinterpret_cast<const f32x2*>(&bs_tab[0 * BN + c0 + jp]);
                    const f32x2 sh = *reinterpret_cast<const f32x2*>(&bs_tab[1 * BN + c0 + jp]);
                    const f32x2 mu = *reinterpret_cast<const f32x2*>(&bs_tab[2 * BN + c0 + jp]);
                    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                    if (bi & 1) { s1 = f32x2{p1[jp], p1[jp + 1]}; s2 = f32x2{p2[jp], p2[jp + 1]}; }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned pk = src[i][jp >> 1];
                        const f32x2 qv = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
                        const f32x2 pre = __builtin_elementwise_fma(qv, sc, sh);       // the ReLU mask, as uh_bn_relu_apply decided it
                        const f32x2 g = {pre[0] > 0.f ? acc[i0 + i][n][jp] : 0.f, pre[1] > 0.f ? acc[i0 + i][n][jp + 1] : 0.f};
                        s1 += g;
                        s2 = __builtin_elementwise_fma(g, qv - mu, s2);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (bi & 1) {
                            const float a1 = uh_row16_sum(s1[e]), a2 = uh_row16_sum(s2[e]);
                            if (lx_o == 0) {
                                S1[c0 + jp + e] += a1;
                                S2[c0 + jp + e] += a2;
                            }
                        } else {
                            p1[jp + e] = s1[e];
                            p2[jp + e] = s2[e];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);       // one channel pair at a time
                }
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sc = bs_tab[0 * BN + c0 + j], sh = bs_tab[1 * BN + c0 + j], mu = bs_tab[2 * BN + c0 + j];
                    float s1 = (bi & 1) ? p1[j] : 0.f, s2 = (bi & 1) ? p2[j] : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned pk = src[i][j >> 1];
                        const float qv = __uint_as_float((j & 1) ? (pk & 0xffff0000u) : (pk << 16));
                        const bool on = fmaf(qv, sc, sh) > 0.f;                  // the ReLU mask, as uh_bn_relu_apply decided it
                        const float g = on ? acc[i0 + i][n][j] : 0.f;            // acc: already rounded to the stored type
                        s1 += g;
                        s2 = fmaf(g, qv - mu, s2);
                    }
                    if (bi & 1) {
                        const float a1 = uh_row16_sum(s1), a2 = uh_row16_sum(s2);
                        if (lx_o == 0) {
                            S1[c0 + j] += a1;
                            S2[c0 + j] += a2;
                        }
                    } else {
                        p1[j] = s1;
                        p2[j] = s2;
                    }
                    __builtin_amdgcn_sched_barrier(0);       // one channel at a time (left free, the scheduler interleaves all four: +40 registers)
                }
#endif
            };
#pragma unroll
            for (int bi = 0; bi < BQB; ++bi) {
                // request batch bi + 1, then sum batch bi (even batches sit in bqA, odd ones in bqB)
                if constexpr (BQ2) {
                    if (bi + 1 < BQB) {
                        if (bi & 1) bq_load(bqA, bi + 1); else bq_load(bqB, bi + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (bi & 1) bq_sum(bqB, bi); else bq_sum(bqA, bi);
                } else {
                    bq_load(bqA, bi);
                    __builtin_amdgcn_sched_barrier(0);
                    bq_sum(bqA, bi);
                }
                __builtin_amdgcn_sched_barrier(0);
                // batch boundary, enforced: hipcc otherwise runs the sums channel-major over BOTH row halves (the partial sums
                // chain them), i.e. with two batches of q and all twelve coefficients in registers.  The partial sums pass through
                // an opaque statement, and the next request's address depends on an opaque copy made behind it.
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(p1[j]), "+v"(p2[j]));
                asm volatile("" : "+v"(lane_o));
            }
            __builtin_amdgcn_sched_barrier(0);
            // The stores below convert the (already rounded) values to bf16 a second time.  hipcc knows that this is the
            // conversion the rounding made and keeps its 128 packed results alive from there to here, beside the 128 fp32 values
            // the sums read: opaque copies make it convert again (two registers at a time).
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int n = 0; n < NBW; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][n][j]));
        }

        // ---- stores, straight from the registers through a buffer descriptor: per-lane byte offset (column, channel
        // piece; out of range for columns / channels that do not exist, which drops the store) + the row offset.  The row
        // offset is ADDED to the vector offset instead of riding in the scalar-offset operand: a 16-byte buffer store with a
        // scalar offset register reads its data over several cycles, and hipcc (ROCm 7.2) let the next VALU instruction
        // overwrite the first data register behind the last store of the loop (wrong bf16 pairs in 8 lanes of one row pair).
        if (!UH_ABL_NOSTORE) {
            const int rbytes = W * ldy * ES;                                   // one image row of y
            const int sbase = ((b * H + y0) * W) * ldy * ES;                   // row 0 of the tile, column 0
            if constexpr (PERM2) {
#pragma unroll
                for (int hp = 0; hp < NBW / 2; ++hp) {             // one 16-byte piece (8 channels) per pair of MFMAs
                const int c0 = co_blk + ch(kg, 2 * hp, 0);
                const bool inr = gx < W && c0 < Coutv;
                const unsigned voff = (unsigned)((gx * ldy + c0) * ES);
                if (full) {
                    // interior tile (wave-uniform): one running offset, no per-row test / select / multiply -- the store section was
                    // 9 % of a wave's lifetime on the K = 1152 layers (profiles/r03_conv_phase_stamps.txt).  An out-of-range lane
                    // (narrow tensors) stays out of range: OOB_OFFSET + 16 rows is far from wrapping.
                    unsigned vo = inr ? voff + (unsigned)sbase : OOB_OFFSET;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        bf16x8 o = {(bf16_t)acc[i][2 * hp][0], (bf16_t)acc[i][2 * hp][1], (bf16_t)acc[i][2 * hp][2], (bf16_t)acc[i][2 * hp][3],
                                    (bf16_t)acc[i][2 * hp + 1][0], (bf16_t)acc[i][2 * hp + 1][1], (bf16_t)acc[i][2 * hp + 1][2], (bf16_t)acc[i][2 * hp + 1][3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsy, vo, 0, 0);
                        vo += (unsigned)rbytes;
                    }
                } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i < vy) {
                        bf16x8 o = {(bf16_t)acc[i][2 * hp][0], (bf16_t)acc[i][2 * hp][1], (bf16_t)acc[i][2 * hp][2], (bf16_t)acc[i][2 * hp][3],
                                    (bf16_t)acc[i][2 * hp + 1][0], (bf16_t)acc[i][2 * hp + 1][1], (bf16_t)acc[i][2 * hp + 1][2], (bf16_t)acc[i][2 * hp + 1][3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsy, inr ? voff + (unsigned)(sbase + i * rbytes) : OOB_OFFSET, 0, 0);
                    }
                }
                }
                }
            } else if (perm1) {
                // rows i, i+1: after the half-wave swap lanes 0..31 hold 8 channels of row i, lanes 32..63 of row i+1
                const int c0 = co_blk + wave * 16 + (kg & 1) * 8;
                const int rsel = kg >> 1;
                const bool inr = gx < W && c0 < Coutv;
                const unsigned voff0 = (unsigned)((gx * ldy + c0) * ES + rsel * rbytes);
                unsigned vo_run = inr ? voff0 + (unsigned)sbase : OOB_OFFSET;      // interior tiles: one running offset (see above)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    bf16x4 a = {(bf16_t)acc[i][0][0], (bf16_t)acc[i][0][1], (bf16_t)acc[i][0][2], (bf16_t)acc[i][0][3]};
                    bf16x4 bq = {(bf16_t)acc[i + 1][0][0], (bf16_t)acc[i + 1][0][1], (bf16_t)acc[i + 1][0][2], (bf16_t)acc[i + 1][0][3]};
                    u32x2 au = __builtin_bit_cast(u32x2, a), bu = __builtin_bit_cast(u32x2, bq);
                    auto r0 = __builtin_amdgcn_permlane32_swap(au[0], bu[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(au[1], bu[1], false, false);
                    const u32x4 o = u32x4{r0[0], r1[0], r0[1], r1[1]};
                    if (full) {
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsy, vo_run, 0, 0);
                        vo_run += 2u * (unsigned)rbytes;
                    } else if (i < vy) {          // (vy odd: the row i+1 half of the last pair is dropped per lane)
                        const unsigned voff = (inr && i + rsel < vy) ? voff0 + (unsigned)(sbase + i * rbytes) : OOB_OFFSET;
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsy, voff, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const int c0 = co_blk + ch(kg, n, 0);
                    const bool inr = gx < W && c0 < Coutv;
                    const unsigned voff = (unsigned)((gx * ldy + c0) * ES);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (i < vy) {
                            const unsigned vo = inr ? voff + (unsigned)(sbase + i * rbytes) : OOB_OFFSET;
                            if constexpr (ES == 2) {
                                bf16x4 o = {(bf16_t)acc[i][n][0], (bf16_t)acc[i][n][1], (bf16_t)acc[i][n][2], (bf16_t)acc[i][n][3]};
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsy, vo, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][n]), rsy, vo, 0, 0);
                            }
                        }
                    }
                }
            }
        }

        // ---- BatchNorm statistics: pivot-shifted sums per lane, 4 DPP adds per channel, accumulated in LDS by the lane that
        // owns the channel's slot (the same lane every tile: no barrier, a wave only touches its own channels)
        if (!BSUM && stats && !UH_ABL_NOSTATS) {
            float* S1 = &wg_sum[0][0];          // slot = channel - co_blk: a wave touches its own channels only
            float* S2 = &wg_sum[1][0];
            float* PV = &wg_sum[2][0];
            if (n_run == 0.f) {
                // pivot = one value of each channel from INSIDE the workgroup's first tile (pixel (2,8); column 0 only when the
                // tile is that narrow).  The first tile of workgroup 0 is the image corner, where zero padding makes pixel (0,0)
                // the least typical value of a channel: on a near-constant channel (flat background) a corner pivot costs
                // eps * (corner - level)^2 of noise per pixel in S2 - S1^2 / n, an interior one costs nothing.  Row 2 and not
                // the tile centre: reading acc[8] here makes the weight-resident instantiation spill (two reloads per tile;
                // no time difference in an A/B, but the ISA lint then has something to report); rows 1 and 2 allocate without.  On an
                // image under three rows high the value comes from the zero-filled halo: still a valid pivot, ~plain sums.
                if (lx == (vx > 8 ? 8 : 0)) {
#pragma unroll
                    for (int n = 0; n < NBW; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int cl = ch(kg, n, j);
                            PV[cl] = acc[2][n][j];
                            S1[cl] = 0.f;
                            S2[cl] = 0.f;
                        }
                }
            }
            // Two channels at a time, as packed fp32 pairs (v_pk_add_f32 / v_pk_fma_f32: the accumulators of channels j, j + 1 of a
            // pixel are an aligned register pair): half the VALU instructions of the scalar form -- the statistics were 10 % of a
            // wave's lifetime on the K = 1152 layers (profiles/r03_conv_phase_stamps.txt) -- at six live temporaries.
            if constexpr (!WRES) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(&PV[ch(kg, n, 0)]);      // same-wave LDS write -> read: in order
#pragma unroll
                    for (int jp = 0; jp < 4; jp += 2) {
                        const f32x2 pv = {q[jp], q[jp + 1]};
                        f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                        if (full) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const f32x2 v = {acc[i][n][jp], acc[i][n][jp + 1]};
                                const f32x2 d = v - pv;
                                s1 += d;
                                s2 = __builtin_elementwise_fma(d, d, s2);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const bool in = (i < vy) && (gx < W);
                                const f32x2 v = {acc[i][n][jp], acc[i][n][jp + 1]};
                                f32x2 d = v - pv;
                                d[0] = in ? d[0] : 0.f;
                                d[1] = in ? d[1] : 0.f;
                                s1 += d;
                                s2 = __builtin_elementwise_fma(d, d, s2);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float a1 = uh_row16_sum(s1[e]), a2 = uh_row16_sum(s2[e]);   // lanes of one kg = one DPP row
                            if (lx == 0) {
                                const int cl = ch(kg, n, jp + e);
                                S1[cl] += a1;
                                S2[cl] += a2;
                            }
                        }
                    }
                }
            } else {
                // (the weight-resident instantiation has no six registers to spare here: one channel at a time, 3 live temporaries)
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(&PV[ch(kg, n, 0)]);      // same-wave LDS write -> read: in order
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pv = q[j];
                        float s1 = 0.f, s2 = 0.f;
                        if (full) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const float d = acc[i][n][j] - pv;
                                s1 += d;
                                s2 = fmaf(d, d, s2);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const float d = ((i < vy) && (gx < W)) ? acc[i][n][j] - pv : 0.f;
                                s1 += d;
                                s2 = fmaf(d, d, s2);
                            }
                        }
                        const float a1 = uh_row16_sum(s1), a2 = uh_row16_sum(s2);   // lanes of one kg = one DPP row
                        if (lx == 0) {
                            const int cl = ch(kg, n, j);
                            S1[cl] += a1;
                            S2[cl] += a2;
                        }
                    }
                }
            }
            n_run += (float)(vy * vx);
        }
    }
    if constexpr (BSUM) {
        // one partial row per workgroup: [row][0][c] = sum g, [row][1][c] = rstd * sum g * (q - mean)
        if (lx == 0) {
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = ch(kg, n, j);
                    stats[((int64_t)tile_lane * 2 + 0) * Cout + co_blk + cl] = wg_sum[0][cl];
                    stats[((int64_t)tile_lane * 2 + 1) * Cout + co_blk + cl] = wg_sum[1][cl] * bs_tab[3 * BN + cl];
                }
        }
        return;
    }
    if (stats) {
        // row = tile lane of this workgroup; rows nlanes .. ntile-1 of the (per-tile sized) buffer get a zero pixel
        // count, which uh_bn_finalize skips
        if (lx == 0) {
            const float inv_n = 1.f / n_run;
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = ch(kg, n, j);
                    const float a1 = wg_sum[0][cl], a2 = wg_sum[1][cl];
                    stats[((int64_t)tile_lane * 2 + 0) * Cout + co_blk + cl] = wg_sum[2][cl] + a1 * inv_n;      // mean
                    stats[((int64_t)tile_lane * 2 + 1) * Cout + co_blk + cl] = fmaxf(a2 - a1 * a1 * inv_n, 0.f); // M2
                }
        }
        float* counts = stats + (int64_t)ntile * 2 * Cout;
        if (slab == 0) {
            if (tid == 0) counts[tile_lane] = n_run;
            if (tile_lane == 0)
                for (int r = nlanes + tid; r < ntile; r += 256) counts[r] = 0.f;
        }
    }
}

// =====================================================================================
// forward, stem (Cin <= 4): lane = output channel, halo tile broadcast from LDS
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_fwd_stem(const T* __restrict__ x, int Cin, int ldx,
                                                        const T* __restrict__ w, T* __restrict__ y, int ldy, int Cout,
                                                        float* __restrict__ stats, int B, int H, int W, int tilesX,
                                                        int tilesY) {
    __shared__ float xs[HALO_PIX * 4];
    __shared__ float red[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int txt = t % tilesX; t /= tilesX;
    const int tyt = t % tilesY;
    const int b = t / tilesY;
    const int y0 = tyt * TILE, x0p = txt * TILE;
    for (int idx = tid; idx < HALO_PIX * 4; idx += 256) {
        int q = idx >> 2, ci = idx & 3;
        int hy = q / HALO_W, hx = q - hy * HALO_W;
        int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
        float v = 0.f;
        if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = uh_to_f32(x[(int64_t)((b * H + gy) * W + gx) * ldx + ci]);
        xs[idx] = v;
    }
    __syncthreads();
    for (int cg = 0; cg < Cout; cg += 64) {
        const int co = cg + lane;
        const bool cok = co < Cout;
        float wr[36];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
                wr[tap * 4 + ci] = (cok && ci < Cin) ? uh_to_f32(w[((int64_t)co * 9 + tap) * Cin + ci]) : 0.f;
        const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
        float mu = 0.f;
        for (int pass = 0; pass < (stats ? 2 : 1); ++pass) {
            float s1 = 0.f;
            for (int rr = 0; rr < 4; ++rr) {
                const int ty = wave * 4 + rr;
                const int gy = y0 + ty;
                if (gy >= H) break;
                for (int tx = 0; tx < TILE; ++tx) {
                    const int gx = x0p + tx;
                    if (gx >= W) break;
                    float a = 0.f;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            const float* xp = &xs[((ty + r) * HALO_W + tx + s) * 4];
#pragma unroll
                            for (int ci = 0; ci < 4; ++ci) a = fmaf(xp[ci], wr[(r * 3 + s) * 4 + ci], a);
                        }
                    T o = uh_from_f32<T>(a);
                    float v = uh_to_f32(o);
                    if (cok) {
                        if (pass == 0) { y[(int64_t)((b * H + gy) * W + gx) * ldy + co] = o; s1 += v; }
                        else s1 += (v - mu) * (v - mu);
                    }
                }
            }
            if (stats) {
                __syncthreads();
                red[wave * 64 + lane] = s1;
                __syncthreads();
                float tot = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
                if (pass == 0) mu = tot / (float)(vy * vx);
                else if (wave == 0 && cok) {
                    stats[((int64_t)blockIdx.x * 2 + 0) * Cout + co] = mu;
                    stats[((int64_t)blockIdx.x * 2 + 1) * Cout + co] = tot;
                }
            }
        }
        if (stats && cg == 0 && tid == 0) stats[(int64_t)gridDim.x * 2 * Cout + blockIdx.x] = (float)(vy * vx);
    }
}

// =====================================================================================
// forward, stem v2 (Cin <= 4, Cout % V == 0): thread = (pixel lane, V consecutive output channels) so every
// store is a 16-byte piece of a full pixel row; 16x16 tile per workgroup, halo + filters in LDS.
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_fwd_stem_v2(const T* __restrict__ x, int Cin, int ldx,
                                                           const T* __restrict__ w, T* __restrict__ y, int ldy, int Cout,
                                                           float* __restrict__ stats, int B, int H, int W, int tilesX,
                                                           int tilesY) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int PPT = 8;                       // pixels per thread per pass (256 px / 32 pixel lanes)
    extern __shared__ float sm[];                // xs[324*4] | ws[9*4*CG] | red[CG] | part[256][V]
    const int G = Cout / V;                      // channel groups
    const int GB = G < 8 ? G : 8;                // groups per pass (8 x V channels)
    const int CG = GB * V;
    float* xs = sm;
    float* ws = sm + HALO_PIX * 4;
    float* red = ws + 36 * CG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int txt = t % tilesX; t /= tilesX;
    const int tyt = t % tilesY;
    const int b = t / tilesY;
    const int y0 = tyt * TILE, x0p = txt * TILE;
    const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
    for (int idx = tid; idx < HALO_PIX * 4; idx += 256) {
        int q = idx >> 2, ci = idx & 3;
        int hy = q / HALO_W, hx = q - hy * HALO_W;
        int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
        float v = 0.f;
        if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = uh_to_f32(x[(int64_t)((b * H + gy) * W + gx) * ldx + ci]);
        xs[idx] = v;
    }
    const int g = tid % GB, pl = tid / GB;       // pl in [0, 256/GB)
    const int PL = 256 / GB;
    for (int cb = 0; cb < Cout; cb += CG) {
        __syncthreads();
        for (int idx = tid; idx < 36 * CG; idx += 256) {          // ws[(tap*4+ci)*CG + c]
            int c = idx % CG, tc = idx / CG, tap = tc >> 2, ci = tc & 3;
            ws[idx] = (ci < Cin && cb + c < Cout) ? uh_to_f32(w[((int64_t)(cb + c) * 9 + tap) * Cin + ci]) : 0.f;
        }
        __syncthreads();
        const bool cok = cb + g * V < Cout;
        float acc[PPT][V];
#pragma unroll
        for (int j = 0; j < PPT; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[j][i] = 0.f;
        if (pl < PL) {
            for (int tap = 0; tap < 9; ++tap) {
                const int r = tap / 3, s = tap - 3 * r;
                for (int ci = 0; ci < Cin; ++ci) {
                    float wv[V];
#pragma unroll
                    for (int i = 0; i < V; ++i) wv[i] = ws[(tap * 4 + ci) * CG + g * V + i];
#pragma unroll
                    for (int j = 0; j < PPT; ++j) {
                        const int px = pl + j * PL;             // PL * PPT >= 256 when GB == 8
                        if (px < 256) {
                            const float xv = xs[(((px >> 4) + r) * HALO_W + (px & 15) + s) * 4 + ci];
#pragma unroll
                            for (int i = 0; i < V; ++i) acc[j][i] = fmaf(xv, wv[i], acc[j][i]);
                        }
                    }
                }
            }
        }
        float s1[V];
#pragma unroll
        for (int i = 0; i < V; ++i) s1[i] = 0.f;
        if (pl < PL && cok) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int px = pl + j * PL;
                const int ty = px >> 4, tx = px & 15;
                const bool ok = px < 256 && ty < vy && tx < vx;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    acc[j][i] = uh_round_as<T>(acc[j][i]);
                    s1[i] += ok ? acc[j][i] : 0.f;
                }
                if (ok) uh_store<T, V>(y + (int64_t)((b * H + y0 + ty) * W + x0p + tx) * ldy + cb + g * V, acc[j]);
            }
        }
        if (stats) {
            // per-channel tile statistics: reduce over the pixel lanes (threads with equal g) through LDS
            for (int pass = 0; pass < 2; ++pass) {
                // every thread parks its partial in LDS, then one thread per channel adds the pixel lanes in a FIXED
                // order (no atomics: the statistics, and with them the whole step, must not depend on the schedule)
                __syncthreads();
                {
                    float* part = red + CG;                       // [256][V]
#pragma unroll
                    for (int i = 0; i < V; ++i) part[tid * V + i] = (pl < PL && cok) ? s1[i] : 0.f;
                    __syncthreads();
                    for (int c = tid; c < CG; c += 256) {
                        const int gg = c / V, ii = c - gg * V;
                        float t = 0.f;
                        for (int k = 0; k < PL; ++k) t += part[(k * GB + gg) * V + ii];
                        red[c] = t;
                    }
                }
                __syncthreads();
                if (pass == 0) {
                    const float inv = 1.f / (float)(vy * vx);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float mu = red[g * V + i] * inv;
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < PPT; ++j) {
                            const int px = pl + j * PL;
                            const bool ok = px < 256 && (px >> 4) < vy && (px & 15) < vx;
                            const float d = acc[j][i] - mu;
                            a += ok ? d * d : 0.f;
                        }
                        s1[i] = a;
                    }
                    if (tid < CG && cb + tid < Cout) stats[((int64_t)blockIdx.x * 2 + 0) * Cout + cb + tid] = red[tid] * inv;
                } else if (tid < CG && cb + tid < Cout) {
                    stats[((int64_t)blockIdx.x * 2 + 1) * Cout + cb + tid] = red[tid];
                }
            }
        }
    }
    if (stats && tid == 0) stats[(int64_t)gridDim.x * 2 * Cout + blockIdx.x] = (float)(vy * vx);
}

// =====================================================================================
// forward, stem v3 (Cin <= 4, Cout == 64): persistent workgroups walk the 16x16 tiles; thread = (pixel lane of 32,
// 8 consecutive output channels) so every store is a 16-byte piece of a full pixel row; the 9*Cin*8 filter taps
// live in registers for the whole kernel; BatchNorm statistics are shifted sums (shift K = the workgroup's first
// output pixel) accumulated in registers over ALL tiles of the workgroup and reduced once: one slab per workgroup.
// =====================================================================================
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv3x3_fwd_stem_v3(const T* __restrict__ x, int ldx, const T* __restrict__ w,
                                                           T* __restrict__ y, int ldy, float* __restrict__ stats, int B,
                                                           int H, int W, int tilesX, int tilesY,
                                                           const float* __restrict__ ep_scale,
                                                           const float* __restrict__ ep_shift) {
    constexpr int V = 8, COUT = 64;
    __shared__ float xs[HALO_PIX * CIN];
    __shared__ float kshift[COUT];
    __shared__ float red[4][2][COUT];
    const int tid = threadIdx.x, g = tid & 7, pl = tid >> 3;      // 8 channel groups x 32 pixel lanes
    const int ntile = B * tilesX * tilesY;
    float wr[9 * CIN][V];
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) wr[k][i] = uh_to_f32(w[((int64_t)(g * V + i) * 9 + k / CIN) * CIN + (k % CIN)]);
    float s1[V], s2[V], ks[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s1[i] = 0.f; s2[i] = 0.f; ks[i] = 0.f; }
    float esc[V], esh[V];                 // inference epilogue: eval-mode BatchNorm scale/shift + ReLU
#pragma unroll
    for (int i = 0; i < V; ++i) { esc[i] = ep_scale ? ep_scale[g * V + i] : 1.f; esh[i] = ep_scale ? ep_shift[g * V + i] : 0.f; }
    float cnt = 0.f;
    bool have_k = false;
    // the image halo of the NEXT tile is fetched into registers while the current tile is computed, so a workgroup does
    // not sit through an HBM round trip between its two barriers on every tile
    constexpr int NX = (HALO_PIX * CIN + 255) / 256;
    float xr[NX];
    auto fetch_halo = [&](int tile_) {
        int t = tile_;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TILE, x0p = txt * TILE;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + k * 256;
            const int q = idx / CIN, ci = idx - q * CIN;
            const int hy = q / HALO_W, hx = q - hy * HALO_W;
            const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
            const bool in = idx < HALO_PIX * CIN && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1);
            const float v = uh_to_f32(x[(int64_t)((b * H + gyc) * W + gxc) * ldx + (idx < HALO_PIX * CIN ? ci : 0)]);
            xr[k] = in ? v : 0.f;
        }
    };
    if ((int)blockIdx.x < ntile) fetch_halo(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TILE, x0p = txt * TILE;
        const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (tid + k * 256 < HALO_PIX * CIN) xs[tid + k * 256] = xr[k];
        __syncthreads();
        if (tile + (int)gridDim.x < ntile) fetch_halo(tile + gridDim.x);
        float out[8][V];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = pl + j * 32, ty = px >> 4, tx = px & 15;
            float a[V];
#pragma unroll
            for (int i = 0; i < V; ++i) a[i] = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int ss = 0; ss < 3; ++ss)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        const float xv = xs[((ty + r) * HALO_W + tx + ss) * CIN + ci];
#pragma unroll
                        for (int i = 0; i < V; ++i) a[i] = fmaf(xv, wr[(r * 3 + ss) * CIN + ci][i], a[i]);
                    }
            if (ep_scale) {
#pragma unroll
                for (int i = 0; i < V; ++i) a[i] = uh_relu(fmaf(a[i], esc[i], esh[i]));
            }
#pragma unroll
            for (int i = 0; i < V; ++i) out[j][i] = uh_round_as<T>(a[i]);
            if (ty < vy && tx < vx) uh_store<T, V>(y + (int64_t)((b * H + y0 + ty) * W + x0p + tx) * ldy + g * V, out[j]);
        }
        if (stats) {
            if (!have_k) {
                // shift = one output pixel of this workgroup's first tile: (8,8) when the tile has it, else (0,0) -- an interior
                // pixel, because the first tile of workgroup 0 is the image corner, the least typical value of a flat channel
                const bool mid = vy > 8 && vx > 8;                 // pixel (8,8) = px 136 = pl 8, j 4
                if (pl == (mid ? 8 : 0)) {
#pragma unroll
                    for (int i = 0; i < V; ++i) kshift[g * V + i] = mid ? out[4][i] : out[0][i];
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < V; ++i) ks[i] = kshift[g * V + i];
                have_k = true;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = pl + j * 32;
                if ((px >> 4) < vy && (px & 15) < vx) {
#pragma unroll
                    for (int i = 0; i < V; ++i) { const float d = out[j][i] - ks[i]; s1[i] += d; s2[i] += d * d; }
                }
            }
            cnt += (float)(vy * vx);
        }
    }
    if (stats) {
        // reduce over the 32 pixel lanes: lanes of a wave with equal g differ in bits 3..5; then across the 4 waves
#pragma unroll
        for (int i = 0; i < V; ++i)
            for (int o = 8; o < 64; o <<= 1) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
        __syncthreads();
        if ((tid & 63) < 8) {
#pragma unroll
            for (int i = 0; i < V; ++i) { red[tid >> 6][0][g * V + i] = s1[i]; red[tid >> 6][1][g * V + i] = s2[i]; }
        }
        __syncthreads();
        if (tid < COUT) {
            const float a = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            const float q = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            const float n = cnt > 0.f ? cnt : 1.f;
            const float kk = have_k ? kshift[tid] : 0.f;
            stats[((int64_t)blockIdx.x * 2 + 0) * COUT + tid] = kk + a / n;                   // mean
            stats[((int64_t)blockIdx.x * 2 + 1) * COUT + tid] = fmaxf(q - a * a / n, 0.f);     // M2 (shifted-data formula)
        }
        // slab layout is sized for one row per TILE: rows beyond the workgroup count get a zero pixel count (skipped)
        float* counts = stats + (int64_t)ntile * 2 * COUT;
        if (tid == 0) counts[blockIdx.x] = cnt;
        for (int r = gridDim.x + blockIdx.x * 256 + tid; r < ntile; r += gridDim.x * 256) counts[r] = 0.f;
    }
}

// =====================================================================================
// forward, generic (any channel counts)
// =====================================================================================
template <typename T>
__global__ void conv3x3_fwd_generic(const T* __restrict__ x0, int C0, int ld0, const T* __restrict__ x1, int C1, int ld1,
                                    const T* __restrict__ w, T* __restrict__ y, int ldy, int Cout, int B, int H, int W) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)B * H * W * Cout;
    if (idx >= total) return;
    int co = (int)(idx % Cout);
    int64_t p = idx / Cout;
    int wx = (int)(p % W);
    int hy = (int)((p / W) % H);
    int b = (int)(p / ((int64_t)W * H));
    const int Cin = C0 + C1;
    float acc = 0.f;
    for (int r = 0; r < 3; ++r) {
        int yy = hy + r - 1;
        if (yy < 0 || yy >= H) continue;
        for (int s = 0; s < 3; ++s) {
            int xx = wx + s - 1;
            if (xx < 0 || xx >= W) continue;
            int64_t pix = ((int64_t)b * H + yy) * W + xx;
            const T* wr = w + ((int64_t)co * 9 + r * 3 + s) * Cin;
            const T* p0 = x0 + pix * ld0;
            for (int i = 0; i < C0; ++i) acc = fmaf(uh_to_f32(p0[i]), uh_to_f32(wr[i]), acc);
            if (C1) {
                const T* p1 = x1 + pix * ld1;
                for (int i = 0; i < C1; ++i) acc = fmaf(uh_to_f32(p1[i]), uh_to_f32(wr[C0 + i]), acc);
            }
        }
    }
    y[p * ldy + co] = uh_from_f32<T>(acc);
}

// per-16x16-tile channel statistics of a stored tensor (used after the generic forward)
template <typename T>
__global__ __launch_bounds__(256) void tile_stats_kernel(const T* __restrict__ y, int ldy, int C, float* __restrict__ stats,
                                                         int B, int H, int W, int tilesX, int tilesY) {
    int t = blockIdx.x;
    const int txt = t % tilesX; t /= tilesX;
    const int tyt = t % tilesY;
    const int b = t / tilesY;
    const int y0 = tyt * TILE, x0p = txt * TILE;
    const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int ty = 0; ty < vy; ++ty)
            for (int tx = 0; tx < vx; ++tx) s1 += uh_to_f32(y[(int64_t)((b * H + y0 + ty) * W + x0p + tx) * ldy + c]);
        const float mu = s1 / (float)(vy * vx);
        for (int ty = 0; ty < vy; ++ty)
            for (int tx = 0; tx < vx; ++tx) {
                float d = uh_to_f32(y[(int64_t)((b * H + y0 + ty) * W + x0p + tx) * ldy + c]) - mu;
                s2 += d * d;
            }
        stats[((int64_t)blockIdx.x * 2 + 0) * C + c] = mu;
        stats[((int64_t)blockIdx.x * 2 + 1) * C + c] = s2;
    }
    if (threadIdx.x == 0) stats[(int64_t)gridDim.x * 2 * C + blockIdx.x] = (float)(vy * vx);
}

// =====================================================================================
// host dispatch: forward
// =====================================================================================
extern "C" int uh_conv3x3_stat_slabs(int B, int H, int W, int Cin, int Cout, int dt) {
    (void)Cin; (void)Cout; (void)dt;
    return B * ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
}

// UH_NO_WRES=1 (A/B runs): the 64-input-channel layers take the streaming-filter instantiation instead of the
// register-resident one.  Read once per process, not per launch.
static bool uh_no_wres() {
    static const bool off = getenv("UH_NO_WRES") != nullptr;
    return off;
}

// Which bf16 instantiation of conv3x3_fwd_mfma_v2 a plain call takes and how many tile lanes (= workgroups per channel slab =
// statistics / partial rows written) it is launched with.  Mirrors the branches of conv3x3_fwd_dispatch below.
// K split inside the workgroup (KS = 2, see the kernel): bf16, 16 channels per wave, at most one (tile, 64-channel slab) pair
// per CU and an even number (>= 8) of 32-channel K-chunks.  UH_NO_KSPLIT=1 turns it off (A/B runs).
// UH_WRES_WIDE=1 (experiment, round 5): 64 input channels and 128+ output channels (down1.0 forward, backward-data of up4.0 / up3.3)
// take the register-resident-filter form, one launch of 64-channel slabs, instead of the 32-channel-per-wave streaming form
static bool uh_wres_wide() {
    static const bool on = getenv("UH_WRES_WIDE") != nullptr && getenv("UH_WRES_WIDE")[0] == '1';
    return on;
}

static bool fwd_ksplit_ok(int ntile, int Cin, int Cout) {
    static const bool off = getenv("UH_NO_KSPLIT") != nullptr && getenv("UH_NO_KSPLIT")[0] == '1';
    const int nchunk = Cin / 32;
    return !off && Cin % 32 == 0 && Cout % 64 == 0 && (int64_t)ntile * (Cout / 64) <= 256 && nchunk >= 8 && nchunk % 2 == 0;
}
struct FwdSel { int nbw; bool wres; int slabs; int gx; int ks; };
static FwdSel fwd_select(int ntile, int Cin, int Cout, bool bf16_plain, bool bsum = false) {
    auto lanes_for = [&](int per_cu, int slabs) {
        int gx = (per_cu * 256 + slabs - 1) / slabs;
        gx = (gx + 7) & ~7;
        if (gx > ntile) gx = ntile;
        return gx;
    };
    FwdSel r;
    r.ks = 1;
    const bool wres_first = bf16_plain && Cin == 64 && !uh_no_wres() && uh_wres_wide();
    if (!wres_first && Cout % 128 == 0 && (int64_t)ntile * (Cout / 128) >= 512) { r.nbw = 2; r.wres = false; r.slabs = Cout / 128; r.gx = lanes_for(2, r.slabs); }
    else if (bf16_plain && Cin == 64 && !uh_no_wres()) { r.nbw = 1; r.wres = true; r.slabs = Cout / 64; r.gx = lanes_for(2, r.slabs); }
    else {
        r.nbw = 1; r.wres = false; r.slabs = Cout / 64; r.gx = lanes_for(bsum ? 2 : 3, r.slabs);
        if (bf16_plain && fwd_ksplit_ok(ntile, Cin, Cout)) { r.ks = 2; r.gx = ntile; }      // one tile per (8-wave) workgroup
    }
    return r;
}

template <typename T>
static int conv3x3_fwd_dispatch(const T* x0, int C0, int ld0, const T* x1, int C1, int ld1, const T* w, T* y, int ldy,
                                int Cout, float* stats, int B, int H, int W, hipStream_t st, const float* ep_scale,
                                const float* ep_shift, bool* ep_done, bool split = false, int C0v = -1, int C1v = -1,
                                int Coutv = -1, bool wfrag = false, const float* pre_scale = nullptr,
                                const float* pre_shift = nullptr, const T* bs_y = nullptr, int bs_ld = 0,
                                const float* bs_coef = nullptr) {
    const bool narrow = C0v >= 0;          // narrow tensors: only the LDS-DMA MFMA kernel implements the channel masks
    const bool pre = pre_scale != nullptr; // BatchNorm + ReLU of the producer applied to source 0 by this kernel's loader
    const bool bsum = bs_y != nullptr;     // backward-data + the BatchNorm-backward sums of the tensor it differentiates (stats = the partial rows)
    if (!narrow) { C0v = C0; C1v = C1; Coutv = Cout; }
    // ep_scale/ep_shift (inference): kernels that apply them in their epilogue set *ep_done; for the others the caller
    // runs the separate scale/shift/ReLU pass
    *ep_done = false;
    constexpr int ES = sizeof(T);
    constexpr int CK = 64 / ES;
    const int tilesX = (W + TILE - 1) / TILE, tilesY = (H + TILE - 1) / TILE;
    const int ntile = B * tilesX * tilesY;
    const int Cin = C0 + C1;
    const bool mfma_ok = (C0 % CK == 0) && (C1 % CK == 0) && (Cout % 64 == 0) && uh_aligned16(x0) &&
                         (C1 == 0 || uh_aligned16(x1)) && uh_aligned16(w) && uh_aligned16(y) &&
                         ((ld0 * ES) % 16 == 0) && (C1 == 0 || (ld1 * ES) % 16 == 0) && ((ldy * ES) % 16 == 0);
    if (mfma_ok) {
        const int64_t b0 = (int64_t)B * H * W * ld0 * ES, b1 = C1 ? (int64_t)B * H * W * ld1 * ES : 0;
        // y and the filter pack are addressed through buffer descriptors as well (32-bit offsets)
        const int64_t by = (int64_t)B * H * W * ldy * ES, bw = (int64_t)Cout * 9 * Cin * (split ? 4 : ES);
        if (b0 < (1ll << 31) - 4096 && b1 < (1ll << 31) - 4096 && by < (1ll << 31) - 4096 && bw < (1ll << 31) - 4096) {
            // 128-channel slabs halve the halo re-reads, but small feature maps need the extra workgroups
            // persistent workgroups: 2 (NBW=2, VGPR-bound) or 3 (NBW=1) per CU, spread over the channel slabs; the
            // number of tile lanes is rounded to a multiple of 8 for the XCD-aware mapping inside the kernel
            auto lanes_for = [&](int per_cu, int slabs) {
                int gx = (per_cu * 256 + slabs - 1) / slabs;
                gx = (gx + 7) & ~7;
                if (gx > ntile) gx = ntile;
                return gx;
            };
            if (ep_scale && !(uh_aligned16(ep_scale) && uh_aligned16(ep_shift))) ep_scale = ep_shift = nullptr;   // 16-B loads
            constexpr bool CAN_SPLIT = (ES == 4);
            if (split && !CAN_SPLIT) { uh_set_error("conv3x3_fwd: bf16x3 needs fp32 tensors"); return UH_EINVAL; }
            if (split && wfrag) { uh_set_error("conv3x3_fwd: bf16x3 filters are KRSC packs"); return UH_EINVAL; }
            if (narrow && wfrag) { uh_set_error("conv3x3_fwd: narrow-tensor calls take KRSC packs"); return UH_EINVAL; }
            if (pre) {
#if !UH_BUILD_PRE
                uh_set_error("conv3x3_fwd: the consumer-side BatchNorm+ReLU instantiations are not in this build (UH_BUILD_PRE=1); uh_conv3x3_pre_ok says so");
                return UH_EINVAL;
#else
                if constexpr (ES == 2) {
                    if (split || narrow || C0 > PRE_MAX_C) { uh_set_error("conv3x3_fwd: the fused BatchNorm+ReLU input needs a plain bf16 call with at most %d channels in source 0; ask uh_conv3x3_pre_ok first", PRE_MAX_C); return UH_EINVAL; }
                    const int wf = wfrag ? 1 : 0;
                    if (Cout % 128 == 0 && (int64_t)ntile * (Cout / 128) >= 512) {
                        int slabs = Cout / 128, gx = lanes_for(2, slabs);
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 2, false, false, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wf, pre_scale, pre_shift);
                    } else if (C0 + C1 == 2 * CK && !uh_no_wres()) {
                        int slabs = Cout / 64, gx = lanes_for(2, slabs);
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, true, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wf, pre_scale, pre_shift);
                    } else {
                        int slabs = Cout / 64, gx = lanes_for(3, slabs);
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, false, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wf, pre_scale, pre_shift);
                    }
                    UH_CHECK_LAUNCH("conv3x3_fwd_mfma_v2 (BatchNorm+ReLU input)");
                    *ep_done = ep_scale != nullptr;
                    return UH_OK;
                } else {
                    uh_set_error("conv3x3_fwd: the fused BatchNorm+ReLU input is a bf16 path; ask uh_conv3x3_pre_ok first");
                    return UH_EINVAL;
                }
#endif
            }
            if (bsum) {
                if constexpr (ES == 2) {
                    const int64_t bq = (int64_t)B * H * W * bs_ld * ES;
                    if (split || narrow || C1 != 0 || !stats || !bs_coef || bs_ld != ldy || !uh_aligned16(bs_y) || bq >= (1ll << 31) - 4096) {
                        uh_set_error("uh_conv3x3_dgrad_bnsum: needs a plain single-source bf16 call and a 16-byte aligned BatchNorm input below 2 GiB; ask uh_conv3x3_dgrad_bnsum_rows first");
                        return UH_EINVAL;
                    }
                    const FwdSel sel = fwd_select(ntile, Cin, Cout, true, true);
                    const int wf = wfrag ? 1 : 0, grid = sel.gx * sel.slabs;
                    if (sel.ks == 2)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, false, false, true, 2>), dim3(grid), dim3(512), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, nullptr, nullptr, C0v, C1v, Coutv, wf,
                                           nullptr, nullptr, bs_y, bs_ld, (unsigned)bq, bs_coef);
                    else if (sel.nbw == 2)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 2, false, false, false, true>), dim3(grid), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, nullptr, nullptr, C0v, C1v, Coutv, wf,
                                           nullptr, nullptr, bs_y, bs_ld, (unsigned)bq, bs_coef);
                    else if (sel.wres)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, true, false, true>), dim3(grid), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, nullptr, nullptr, C0v, C1v, Coutv, wf,
                                           nullptr, nullptr, bs_y, bs_ld, (unsigned)bq, bs_coef);
                    else
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, false, false, true>), dim3(grid), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, nullptr, nullptr, C0v, C1v, Coutv, wf,
                                           nullptr, nullptr, bs_y, bs_ld, (unsigned)bq, bs_coef);
                    UH_CHECK_LAUNCH("conv3x3_fwd_mfma_v2 (backward-data + BatchNorm sums)");
                    return UH_OK;
                } else {
                    uh_set_error("uh_conv3x3_dgrad_bnsum: bf16 only; ask uh_conv3x3_dgrad_bnsum_rows first");
                    return UH_EINVAL;
                }
            }
            const bool wres_first = ES == 2 && !split && !narrow && C0 + C1 == 2 * CK && !uh_no_wres() && uh_wres_wide();
            if (!wres_first && Cout % 128 == 0 && (int64_t)ntile * (Cout / 128) >= 512) {
                int slabs = Cout / 128, gx = lanes_for(2, slabs);
                if (split) {
                    if constexpr (CAN_SPLIT)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 2, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
                } else
                    hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 2>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1,
                                       C1, ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
            } else if (ES == 2 && !split && !narrow && C0 + C1 == 2 * CK && !uh_no_wres()) {
                // 64 input channels: the filter stays in registers (2 workgroups per CU: 72 more VGPRs)
                int slabs = Cout / 64, gx = lanes_for(2, slabs);
                if constexpr (ES == 2)
                    hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1,
                                       C1, ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
            } else {
                int slabs = Cout / 64, gx = lanes_for(3, slabs);
                if (split) {
                    if constexpr (CAN_SPLIT)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, true>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1, C1, ld1,
                                           w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
                } else if (ES == 2 && !narrow && fwd_ksplit_ok(ntile, Cin, Cout)) {
                    // few tiles, long contraction: eight waves per workgroup, each half of them half of the K-chunks, one tile each
                    if constexpr (ES == 2)
                        hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1, false, false, false, false, 2>), dim3(ntile * slabs), dim3(512), 0, st, x0, C0, ld0, x1,
                                           C1, ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
                } else
                    hipLaunchKernelGGL((conv3x3_fwd_mfma_v2<T, 1>), dim3(gx * slabs), dim3(256), 0, st, x0, C0, ld0, x1,
                                       C1, ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY, (unsigned)b0, (unsigned)b1, (unsigned)by, ep_scale, ep_shift, C0v, C1v, Coutv, wfrag ? 1 : 0);
            }
            UH_CHECK_LAUNCH("conv3x3_fwd_mfma_v2");
            *ep_done = ep_scale != nullptr;
            return UH_OK;
        }
        if (pre) { uh_set_error("conv3x3_fwd: the fused BatchNorm+ReLU input needs the LDS-DMA MFMA kernel (tensors below 2 GiB); ask uh_conv3x3_pre_ok first"); return UH_EINVAL; }
        if (bsum) { uh_set_error("uh_conv3x3_dgrad_bnsum: needs the LDS-DMA MFMA kernel (tensors below 2 GiB); ask uh_conv3x3_dgrad_bnsum_rows first"); return UH_EINVAL; }
        if (wfrag) { uh_set_error("conv3x3_fwd: the filter is packed fragment-major (UH_WFRAG) but this call cannot take the LDS-DMA MFMA kernel (a tensor of 2 GiB or more); ask uh_conv3x3_wfrag_ok first"); return UH_EINVAL; }
        if (split) { uh_set_error("conv3x3_fwd: bf16x3 is implemented for tensors below 2 GiB only"); return UH_EINVAL; }
        if (narrow) { uh_set_error("conv3x3_fwd: narrow tensors are implemented for tensors below 2 GiB only"); return UH_EINVAL; }
        if (Cout % 128 == 0) {
            hipLaunchKernelGGL((conv3x3_fwd_mfma<T, 4>), dim3(ntile, Cout / 128), dim3(256), 0, st, x0, C0, ld0, x1, C1,
                               ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY);
        } else {
            hipLaunchKernelGGL((conv3x3_fwd_mfma<T, 2>), dim3(ntile, Cout / 64), dim3(256), 0, st, x0, C0, ld0, x1, C1,
                               ld1, w, y, ldy, Cout, stats, B, H, W, tilesX, tilesY);
        }
        UH_CHECK_LAUNCH("conv3x3_fwd_mfma");
        return UH_OK;
    }
    if (pre) { uh_set_error("conv3x3_fwd: the fused BatchNorm+ReLU input needs an MFMA-aligned shape; ask uh_conv3x3_pre_ok first"); return UH_EINVAL; }
    if (bsum) { uh_set_error("uh_conv3x3_dgrad_bnsum: needs an MFMA-aligned shape; ask uh_conv3x3_dgrad_bnsum_rows first"); return UH_EINVAL; }
    if (wfrag) { uh_set_error("conv3x3_fwd: the filter is packed fragment-major (UH_WFRAG) but the shape / alignment is outside the MFMA path; ask uh_conv3x3_wfrag_ok first"); return UH_EINVAL; }
    if (split) { uh_set_error("conv3x3_fwd: bf16x3 needs an MFMA-aligned shape (Cin %% 16 == 0, Cout %% 64 == 0, 16-byte strides)"); return UH_EINVAL; }
    if (narrow) { uh_set_error("conv3x3_fwd: narrow tensors need padded counts that are MFMA-aligned and 16-byte strides"); return UH_EINVAL; }
    if (Cin <= 4 && C1 == 0) {
        constexpr int V = 16 / ES;
        if constexpr (ES == 2) {
            if (Cout == 64 && uh_aligned16(y) && (ldy * ES) % 16 == 0) {
                int grid = ntile < 1024 ? ntile : 1024;
                switch (Cin) {
                    case 1: hipLaunchKernelGGL((conv3x3_fwd_stem_v3<T, 1>), dim3(grid), dim3(256), 0, st, x0, ld0, w, y, ldy, stats, B, H, W, tilesX, tilesY, ep_scale, ep_shift); break;
                    case 2: hipLaunchKernelGGL((conv3x3_fwd_stem_v3<T, 2>), dim3(grid), dim3(256), 0, st, x0, ld0, w, y, ldy, stats, B, H, W, tilesX, tilesY, ep_scale, ep_shift); break;
                    case 3: hipLaunchKernelGGL((conv3x3_fwd_stem_v3<T, 3>), dim3(grid), dim3(256), 0, st, x0, ld0, w, y, ldy, stats, B, H, W, tilesX, tilesY, ep_scale, ep_shift); break;
                    default: hipLaunchKernelGGL((conv3x3_fwd_stem_v3<T, 4>), dim3(grid), dim3(256), 0, st, x0, ld0, w, y, ldy, stats, B, H, W, tilesX, tilesY, ep_scale, ep_shift); break;
                }
                UH_CHECK_LAUNCH("conv3x3_fwd_stem_v3");
                *ep_done = ep_scale != nullptr;
                return UH_OK;
            }
        }
        if (Cout % V == 0 && uh_aligned16(y) && (ldy * ES) % 16 == 0) {
            int G = Cout / V, GB = G < 8 ? G : 8, CG = GB * V;
            size_t sm = (size_t)(HALO_PIX * 4 + 36 * CG + CG + 256 * V) * sizeof(float);
            hipLaunchKernelGGL(conv3x3_fwd_stem_v2<T>, dim3(ntile), dim3(256), sm, st, x0, Cin, ld0, w, y, ldy, Cout, stats,
                               B, H, W, tilesX, tilesY);
            UH_CHECK_LAUNCH("conv3x3_fwd_stem_v2");
            return UH_OK;
        }
        hipLaunchKernelGGL(conv3x3_fwd_stem<T>, dim3(ntile), dim3(256), 0, st, x0, Cin, ld0, w, y, ldy, Cout, stats, B, H,
                           W, tilesX, tilesY);
        UH_CHECK_LAUNCH("conv3x3_fwd_stem");
        return UH_OK;
    }
    int64_t total = (int64_t)B * H * W * Cout;
    hipLaunchKernelGGL(conv3x3_fwd_generic<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x0, C0, ld0, x1,
                       C1, ld1, w, y, ldy, Cout, B, H, W);
    UH_CHECK_LAUNCH("conv3x3_fwd_generic");
    if (stats) {
        hipLaunchKernelGGL(tile_stats_kernel<T>, dim3(ntile), dim3(256), 0, st, (const T*)y, ldy, Cout, stats, B, H, W,
                           tilesX, tilesY);
        UH_CHECK_LAUNCH("tile_stats_kernel");
    }
    return UH_OK;
}

extern "C" int uh_conv3x3_fwd(const void* x0, int C0, int ld0, const void* x1, int C1, int ld1, const void* w, void* y,
                              int ldy, int Cout, float* stat_partials, int B, int H, int W, int dt, uh_stream stream) {
    UH_REQUIRE(x0 && w && y, "uh_conv3x3_fwd: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_fwd: bad shape");
    UH_REQUIRE(ld0 >= C0 && ldy >= Cout && (C1 == 0 || (x1 && ld1 >= C1)), "uh_conv3x3_fwd: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_fwd: pixel count overflows int32");
    const bool wfrag = (dt & UH_WFRAG) != 0;          // the filter pack is fragment-major (uh_pack_w3x3 with the same flag)
    dt &= ~UH_WFRAG;
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_fwd: bad dtype %d", dt);
    hipStream_t st = (hipStream_t)stream;
    bool done;
    if (dt == UH_BF16)
        return conv3x3_fwd_dispatch<bf16_t>((const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1, ld1, (const bf16_t*)w,
                                            (bf16_t*)y, ldy, Cout, stat_partials, B, H, W, st, nullptr, nullptr, &done, false,
                                            -1, -1, -1, wfrag);
    return conv3x3_fwd_dispatch<float>((const float*)x0, C0, ld0, (const float*)x1, C1, ld1, (const float*)w, (float*)y,
                                       ldy, Cout, stat_partials, B, H, W, st, nullptr, nullptr, &done, dt == UH_F32X3, -1, -1,
                                       -1, wfrag);
}

// Will uh_conv3x3_fwd take the LDS-DMA MFMA kernel for this call (pointers assumed 16-byte aligned)?  Only then may the
// filter be packed fragment-major (dt | UH_WFRAG in uh_pack_w3x3 and in the conv call): 8 whole cache lines per fragment
// load instead of 16 half lines.
extern "C" int uh_conv3x3_wfrag_ok(int B, int H, int W, int C0, int C1, int Cout, int ld0, int ld1, int ldy, int dt) {
    if (dt != UH_F32 && dt != UH_BF16) return 0;
    const int es = dt == UH_BF16 ? 2 : 4, ck = 64 / es;
    if (B <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    if (C0 % ck || C1 % ck || Cout % 64) return 0;
    if ((ld0 * es) % 16 || (C1 && (ld1 * es) % 16) || (ldy * es) % 16) return 0;
    const int64_t px = (int64_t)B * H * W, lim = (1ll << 31) - 4096;
    if (px * ld0 * es >= lim || (C1 && px * ld1 * es >= lim) || px * ldy * es >= lim) return 0;
    if ((int64_t)Cout * 9 * (C0 + C1) * es >= lim) return 0;
    return 1;
}

// Training forward whose input is the RAW output y_prev of the previous conv: the BatchNorm + ReLU between the two convs of a
// DoubleConv (unet_parts.py:16-17) is applied by this kernel's loader, x = max(y_prev * pre_scale + pre_shift, 0) rounded to the
// activation dtype exactly as uh_bn_relu_apply stores it -- the activation itself is never written.  bf16, single source of
// <= 512 channels, LDS-DMA MFMA kernel only: uh_conv3x3_pre_ok says whether a call qualifies (else run uh_bn_relu_apply).
extern "C" int uh_conv3x3_pre_ok(int B, int H, int W, int C0, int Cout, int ld0, int ldy, int dt) {
#if !UH_BUILD_PRE
    return 0;            // the default library does not carry the PRE instantiations (measured a net loss: DESIGN.md section 3)
#endif
    if (dt != UH_BF16 || C0 > PRE_MAX_C) return 0;
    if (C0 % 64 || Cout % 64) return 0;                  // (backward-weights of the same layer works on 64-channel slabs)
    return uh_conv3x3_wfrag_ok(B, H, W, C0, 0, Cout, ld0, 0, ldy, dt);
}

extern "C" int uh_conv3x3_fwd_pre(const void* x0, int C0, int ld0, const float* pre_scale, const float* pre_shift, const void* w,
                                  void* y, int ldy, int Cout, float* stat_partials, int B, int H, int W, int dt,
                                  uh_stream stream) {
    UH_REQUIRE(x0 && w && y && pre_scale && pre_shift, "uh_conv3x3_fwd_pre: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && Cout > 0, "uh_conv3x3_fwd_pre: bad shape");
    UH_REQUIRE(ld0 >= C0 && ldy >= Cout, "uh_conv3x3_fwd_pre: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_fwd_pre: pixel count overflows int32");
    const bool wfrag = (dt & UH_WFRAG) != 0;
    dt &= ~UH_WFRAG;
    UH_REQUIRE(dt == UH_BF16, "uh_conv3x3_fwd_pre: bf16 only (dtype %d)", dt);
    UH_REQUIRE(UH_BUILD_PRE, "uh_conv3x3_fwd_pre: the consumer-side BatchNorm+ReLU instantiations are not in this build (UH_BUILD_PRE=1); uh_conv3x3_pre_ok says so");
    UH_REQUIRE(uh_conv3x3_pre_ok(B, H, W, C0, Cout, ld0, ldy, dt), "uh_conv3x3_fwd_pre: shape outside the fused path (uh_conv3x3_pre_ok)");
    bool done;
    return conv3x3_fwd_dispatch<bf16_t>((const bf16_t*)x0, C0, ld0, nullptr, 0, 0, (const bf16_t*)w, (bf16_t*)y, ldy, Cout,
                                        stat_partials, B, H, W, (hipStream_t)stream, nullptr, nullptr, &done, false, -1, -1, -1,
                                        wfrag, pre_scale, pre_shift);
}

// Backward-data of the second conv of a DoubleConv together with the first half of the BatchNorm backward of the layer in
// front of it (see BSUM above).  dy [B,H,W,Cdy] -> dx [B,H,W,Cdx] with the backward-data filter pack; q = the raw output of the
// first conv (what BatchNorm normalised), coef = its [scale | shift | mean | rstd] (Cdx floats each).  partials receives
// uh_conv3x3_dgrad_bnsum_rows() rows of [2][Cdx]: feed them to uh_bn_relu_bwd_apply / uh_bn_bwd_finalize as (partials, nblk = rows)
// in place of uh_bn_relu_bwd_reduce's.  rows == 0: the shape is outside the fused path, run the two kernels separately.
extern "C" int uh_conv3x3_dgrad_bnsum_rows(int B, int H, int W, int Cdy, int Cdx, int lddy, int lddx, int ldq, int dt) {
    if (dt != UH_BF16) return 0;
    if (!uh_conv3x3_wfrag_ok(B, H, W, Cdy, 0, Cdx, lddy, 0, lddx, dt)) return 0;
    if (ldq != lddx) return 0;                 // q is addressed with the offsets of the tensor being written
    const int ntile = B * ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
    return fwd_select(ntile, Cdy, Cdx, true, true).gx;
}

extern "C" int uh_conv3x3_dgrad_bnsum(const void* dy, int Cdy, int lddy, const void* w_dgrad, void* dx, int lddx, int Cdx,
                                      const void* q, int ldq, const float* coef, float* partials, int B, int H, int W, int dt,
                                      uh_stream stream) {
    UH_REQUIRE(dy && w_dgrad && dx && q && coef && partials, "uh_conv3x3_dgrad_bnsum: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && Cdy > 0 && Cdx > 0, "uh_conv3x3_dgrad_bnsum: bad shape");
    UH_REQUIRE(lddy >= Cdy && lddx >= Cdx && ldq >= Cdx, "uh_conv3x3_dgrad_bnsum: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_dgrad_bnsum: pixel count overflows int32");
    const bool wfrag = (dt & UH_WFRAG) != 0;
    dt &= ~UH_WFRAG;
    UH_REQUIRE(dt == UH_BF16, "uh_conv3x3_dgrad_bnsum: bf16 only (dtype %d)", dt);
    UH_REQUIRE(uh_aligned16(dy) && uh_aligned16(w_dgrad) && uh_aligned16(dx) && uh_aligned16(q) && uh_aligned16(coef),
               "uh_conv3x3_dgrad_bnsum: pointers must be 16-byte aligned");
    UH_REQUIRE(uh_conv3x3_dgrad_bnsum_rows(B, H, W, Cdy, Cdx, lddy, lddx, ldq, dt) > 0,
               "uh_conv3x3_dgrad_bnsum: shape outside the fused path (uh_conv3x3_dgrad_bnsum_rows)");
    bool done;
    return conv3x3_fwd_dispatch<bf16_t>((const bf16_t*)dy, Cdy, lddy, nullptr, 0, 0, (const bf16_t*)w_dgrad, (bf16_t*)dx, lddx, Cdx,
                                        partials, B, H, W, (hipStream_t)stream, nullptr, nullptr, &done, false, -1, -1, -1, wfrag,
                                        nullptr, nullptr, (const bf16_t*)q, ldq, coef);
}

// Inference forward: z = max(conv(x, w) * scale + shift, 0) with the eval-mode BatchNorm coefficients of
// uh_bn_eval_coeffs.  The MFMA and stem kernels apply them to the accumulators (no intermediate tensor, one
// rounding); the fallback kernels are followed by the in-place uh_bn_relu_apply pass.
extern "C" int uh_conv3x3_fwd_affine_relu(const void* x0, int C0, int ld0, const void* x1, int C1, int ld1, const void* w,
                                          void* z, int ldz, int Cout, const float* scale, const float* shift, int B,
                                          int H, int W, int dt, uh_stream stream) {
    UH_REQUIRE(x0 && w && z && scale && shift, "uh_conv3x3_fwd_affine_relu: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_fwd_affine_relu: bad shape");
    UH_REQUIRE(ld0 >= C0 && ldz >= Cout && (C1 == 0 || (x1 && ld1 >= C1)), "uh_conv3x3_fwd_affine_relu: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_fwd_affine_relu: pixel count overflows int32");
    const bool wfrag = (dt & UH_WFRAG) != 0;
    dt &= ~UH_WFRAG;
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_fwd_affine_relu: bad dtype %d", dt);
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    int rc;
    if (dt == UH_BF16)
        rc = conv3x3_fwd_dispatch<bf16_t>((const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1, ld1, (const bf16_t*)w,
                                          (bf16_t*)z, ldz, Cout, nullptr, B, H, W, st, scale, shift, &done, false, -1, -1, -1, wfrag);
    else
        rc = conv3x3_fwd_dispatch<float>((const float*)x0, C0, ld0, (const float*)x1, C1, ld1, (const float*)w, (float*)z,
                                         ldz, Cout, nullptr, B, H, W, st, scale, shift, &done, dt == UH_F32X3, -1, -1, -1, wfrag);
    if (rc != UH_OK || done) return rc;
    return uh_bn_relu_apply(z, ldz, scale, shift, z, ldz, (int64_t)B * H * W, Cout, dt == UH_BF16 ? UH_BF16 : UH_F32, stream);
}

// Narrow tensors (small-width nets: 8..32 channels): the filter pack and the K loop use channel counts rounded up to the
// MFMA granularity (C0 / C1 multiples of a 64-byte chunk, Cout multiple of 64) while x0 / x1 / y hold only C0v / C1v /
// Coutv channels (multiples of a 16-byte piece) at their own pixel strides: nothing padded ever reaches HBM.
// scale == NULL: plain forward (+ statistics, Cout columns per slab row); else the fused inference epilogue.
extern "C" int uh_conv3x3_fwd_narrow(const void* x0, int C0, int C0v, int ld0, const void* x1, int C1, int C1v, int ld1,
                                     const void* w, void* y, int ldy, int Cout, int Coutv, float* stat_partials,
                                     const float* scale, const float* shift, int B, int H, int W, int dt, uh_stream stream) {
    UH_REQUIRE(x0 && w && y, "uh_conv3x3_fwd_narrow: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_fwd_narrow: bad shape");
    UH_REQUIRE(C0v > 0 && C0v <= C0 && C1v >= 0 && C1v <= C1 && Coutv > 0 && Coutv <= Cout, "uh_conv3x3_fwd_narrow: bad valid counts");
    UH_REQUIRE(ld0 >= C0v && ldy >= Coutv && (C1 == 0 || (x1 && ld1 >= C1v)), "uh_conv3x3_fwd_narrow: bad strides");
    UH_REQUIRE((scale == nullptr) == (shift == nullptr), "uh_conv3x3_fwd_narrow: scale and shift come together");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_fwd_narrow: pixel count overflows int32");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_fwd_narrow: bad dtype %d", dt);
    const int vec = dt == UH_BF16 ? 8 : 4;
    UH_REQUIRE(C0v % vec == 0 && C1v % vec == 0 && Coutv % vec == 0, "uh_conv3x3_fwd_narrow: valid counts must be multiples of a 16-byte piece");
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    int rc;
    if (dt == UH_BF16)
        rc = conv3x3_fwd_dispatch<bf16_t>((const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1, ld1, (const bf16_t*)w, (bf16_t*)y,
                                          ldy, Cout, stat_partials, B, H, W, st, scale, shift, &done, false, C0v, C1v, Coutv);
    else
        rc = conv3x3_fwd_dispatch<float>((const float*)x0, C0, ld0, (const float*)x1, C1, ld1, (const float*)w, (float*)y, ldy,
                                         Cout, stat_partials, B, H, W, st, scale, shift, &done, dt == UH_F32X3, C0v, C1v, Coutv);
    if (rc != UH_OK || !scale || done) return rc;
    return uh_bn_relu_apply(y, ldy, scale, shift, y, ldy, (int64_t)B * H * W, Coutv, dt == UH_BF16 ? UH_BF16 : UH_F32, stream);
}

// =====================================================================================
// backward-weights, MFMA.  Workgroup = (64 out-ch) x (64 in-ch) x 9 taps over a range of pixel tiles.
// Wave (wr, wc) owns a 32x32 (co, ci) block for all 9 taps: 9 x 16 = 144 accumulator registers.
//   A = dy^T : rows = co, k = 16 pixels of one tile row
//   B = x    : k = the same 16 pixels shifted by the tap, cols = ci
// Pixel is the contraction index and channel-contiguous in memory, so fragments are transposed
// reads of [pixel][channel] LDS tiles: ds_read_b64_tr_b16 for bf16, ds_read_b32 for fp32.
// The x row fragment for (halo row hy, shift s) is read once and used by the three row taps r with
// dy row hy - r.
// =====================================================================================
template <typename T> struct WgradCfg;
template <> struct WgradCfg<bf16_t> { static constexpr int TH = 8; };    // pixel-tile rows
template <> struct WgradCfg<float> { static constexpr int TH = 8; };

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// SPLIT (T = float, "bf16x3"): both operands are split into bf16 halves as their fragments leave LDS and the three
// products dh*xh + dh*xl + dl*xh run on v_mfma_f32_32x32x8_bf16 (see conv3x3_fwd_mfma_v2).
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_mfma(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x0, int C0, int ld0, const T* __restrict__ x1, int C1,
    int ld1, float* __restrict__ slabs, int Cout, int B, int H, int W, int tilesX, int tilesY, int nsplit,
    int C0v, int C1v, int Coutv) {              // channels that exist in memory (narrow tensors), see conv3x3_fwd_mfma_v2
    constexpr int ES = sizeof(T);
    constexpr int TH = WgradCfg<T>::TH;
    constexpr int PB = 64 * ES;                 // bytes per pixel in LDS (64 channels)
    constexpr int XPIX = (TH + 2) * HALO_W;
    constexpr int DPIX = TH * TILE;
    constexpr int VEC = 16 / ES;
    constexpr int PPP = PB / 16;                // 16-B pieces per pixel (8 bf16 / 16 fp32)
    __shared__ __attribute__((aligned(16))) unsigned char lds[(XPIX + DPIX) * PB];
    unsigned char* xs = lds;
    unsigned char* ds = lds + XPIX * PB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int Cin = C0 + C1;
    const int nci = Cin / 64;
    const int cot = blockIdx.y / nci, cit = blockIdx.y - cot * nci;
    const int co0 = cot * 64, ci0 = cit * 64;
    const T* xsrc; int ldx;
    int xleft;                                   // channels of this 64-channel input slab that exist in memory
    if (ci0 < C0) { xsrc = x0 + ci0; ldx = ld0; xleft = C0v - ci0; } else { xsrc = x1 + (ci0 - C0); ldx = ld1; xleft = C1v - (ci0 - C0); }
    const T* dsrc = dy + co0;
    const int dleft = Coutv - co0;

    const int ntile = B * tilesX * tilesY;
    const int split = blockIdx.x;
    const int t_begin = (int)(((int64_t)ntile * split) / nsplit);
    const int t_end = (int)(((int64_t)ntile * (split + 1)) / nsplit);

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

    // swizzle of 64-byte halves inside a 128-byte bf16 pixel row so that the 4 pixel rows of one
    // transposed read land on distinct banks: half ^= (q >> 1) & 1
    auto lds_addr = [&](int q, int byte) -> int {
        if constexpr (ES == 2) return q * PB + (byte ^ (((q >> 1) & 1) << 6));
        else return q * PB + byte;
    };

    for (int tile = t_begin; tile < t_end; ++tile) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TH, x0p = txt * TILE;
        __syncthreads();   // previous tile's reads finished
        // Staging loads are branch-free (clamped to a valid pixel / piece, zeroed afterwards) and go in batches of four:
        // with the load inside `if (in range)` each one was waited for before the next was issued.
        constexpr int SB = 4;
        for (int base = 0; base < XPIX * PPP; base += SB * 256) {
            u32x4 v[SB];
            bool ok[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int idx = min(base + tid + k * 256, XPIX * PPP - 1);
                const int q = idx / PPP, part = idx - q * PPP;
                const int hy = q / HALO_W, hx = q - hy * HALO_W;
                const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
                ok[k] = gy >= 0 && gy < H && gx >= 0 && gx < W && part * VEC < xleft;
                const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1), pc = part * VEC < xleft ? part : 0;
                v[k] = *reinterpret_cast<const u32x4*>(xsrc + (int64_t)((b * H + gyc) * W + gxc) * ldx + pc * VEC);
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int idx = base + tid + k * 256;
                if (idx < XPIX * PPP) {
                    const int q = idx / PPP, part = idx - q * PPP;
                    *reinterpret_cast<u32x4*>(xs + lds_addr(q, part * 16)) = ok[k] ? v[k] : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        for (int base = 0; base < DPIX * PPP; base += SB * 256) {
            u32x4 v[SB];
            bool ok[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int idx = min(base + tid + k * 256, DPIX * PPP - 1);
                const int q = idx / PPP, part = idx - q * PPP;
                const int ty = q / TILE, tx = q - ty * TILE;
                const int gy = y0 + ty, gx = x0p + tx;
                ok[k] = gy < H && gx < W && part * VEC < dleft;
                const int gyc = min(gy, H - 1), gxc = min(gx, W - 1), pc = part * VEC < dleft ? part : 0;
                v[k] = *reinterpret_cast<const u32x4*>(dsrc + (int64_t)((b * H + gyc) * W + gxc) * lddy + pc * VEC);
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int idx = base + tid + k * 256;
                if (idx < DPIX * PPP) {
                    const int q = idx / PPP, part = idx - q * PPP;
                    *reinterpret_cast<u32x4*>(ds + lds_addr(q, part * 16)) = ok[k] ? v[k] : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        __syncthreads();

        if constexpr (ES == 2) {
            // 32x32x16 operand: lane l holds row/col (l & 31), k = 8*(l >> 5) + j.  One transposed read
            // returns 4 consecutive pixels of one channel: group g = (l >> 4) & 1 selects channels
            // 16g..16g+15, lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3.
            const int l16 = lane & 15;
            const int grp = (lane >> 4) & 1;
            const int kh = lane >> 5;                        // pixels 8*kh .. 8*kh+7 of the row
            const int rq = l16 >> 2, cp = l16 & 3;
            const int a_cbyte = (wr * 32 + grp * 16 + cp * 4) * 2;   // dy channel byte offset
            const int b_cbyte = (wc * 32 + grp * 16 + cp * 4) * 2;   // x channel byte offset
            bf16x8 dfrag[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dfrag[k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (int hy = 0; hy < TH + 2; ++hy) {
                // rotate dy row fragments: dfrag[r] = dy row (hy - r)
                dfrag[2] = dfrag[1];
                dfrag[1] = dfrag[0];
                if (hy < TH) {
                    int qa = hy * TILE + kh * 8 + rq;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(ds + lds_addr(qa, a_cbyte)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(ds + lds_addr(qa + 4, a_cbyte)));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    dfrag[0] = __builtin_bit_cast(bf16x8, both);
                } else {
                    dfrag[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    int qb = hy * HALO_W + s + kh * 8 + rq;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xs + lds_addr(qb, b_cbyte)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xs + lds_addr(qb + 4, b_cbyte)));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf16x8 xf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        // tap (r, s): x halo row hy pairs with dy row hy - r (valid 0 <= hy - r < TH)
                        if (hy - r >= 0 && hy - r < TH)
                            acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfrag[r], xf, acc[r * 3 + s], 0, 0, 0);
                    }
                }
            }
        } else if constexpr (SPLIT) {
            // 32x32x8 bf16 operand: lane l holds row/col (l & 31), k = 4*(l >> 5) + j: 4 consecutive pixels of one channel.
            // A tile row of 16 pixels = two K groups of 8.  fragment = {hi01, hi23, lo01, lo23} (bf16 pairs).
            const int l32 = lane & 31, kh = lane >> 5;
            auto load_split = [&](const unsigned char* base) -> u32x4 {      // base -> pixel 0 of this lane's 4 pixels
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(base + j * PB);
                bf16_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = (bf16_t)v[j]; l[j] = (bf16_t)(v[j] - (float)h[j]); }
                auto pk = [](bf16_t a, bf16_t b) -> unsigned {
                    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
                };
                return u32x4{pk(h[0], h[1]), pk(h[2], h[3]), pk(l[0], l[1]), pk(l[2], l[3])};
            };
            u32x4 dfrag[3][2];                                 // dy rows hy, hy-1, hy-2  x  K group
#pragma unroll
            for (int k = 0; k < 3; ++k) dfrag[k][0] = dfrag[k][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll 1
            for (int hy = 0; hy < TH + 2; ++hy) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    dfrag[2][g] = dfrag[1][g];
                    dfrag[1][g] = dfrag[0][g];
                    dfrag[0][g] = (hy < TH) ? load_split(ds + (hy * TILE + g * 8 + kh * 4) * PB + (wr * 32 + l32) * 4)
                                            : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const u32x4 xfr = load_split(xs + (hy * HALO_W + s + g * 8 + kh * 4) * PB + (wc * 32 + l32) * 4);
                        const s16x4 xh = __builtin_bit_cast(s16x4, u32x2{xfr[0], xfr[1]});
                        const s16x4 xl = __builtin_bit_cast(s16x4, u32x2{xfr[2], xfr[3]});
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            if (hy - r >= 0 && hy - r < TH) {
                                const s16x4 dh = __builtin_bit_cast(s16x4, u32x2{dfrag[r][g][0], dfrag[r][g][1]});
                                const s16x4 dl = __builtin_bit_cast(s16x4, u32x2{dfrag[r][g][2], dfrag[r][g][3]});
                                acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(dl, xh, acc[r * 3 + s], 0, 0, 0);
                                acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(dh, xl, acc[r * 3 + s], 0, 0, 0);
                                acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(dh, xh, acc[r * 3 + s], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        } else {
            // fp32: v_mfma_f32_32x32x2_f32, lane l: row/col (l & 31), k = l >> 5
            const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll 1
            for (int hy = 0; hy < TH + 2; ++hy) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int ty = hy - r;
                        if (ty < 0 || ty >= TH) continue;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            const int px = kk * 2 + kh;
                            float a = *reinterpret_cast<const float*>(ds + (ty * TILE + px) * PB + (wr * 32 + l32) * 4);
                            float bb = *reinterpret_cast<const float*>(xs + (hy * HALO_W + px + s) * PB + (wc * 32 + l32) * 4);
                            acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[r * 3 + s], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- write this split's slab: slabs[split][co][tap][ci] fp32
    // 32x32 C layout: col = lane & 31 (ci), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (co)
    float* slab = slabs + (int64_t)split * Cout * 9 * Cin;
    const int ci = ci0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int co = co0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            slab[((int64_t)co * 9 + tap) * Cin + ci] = acc[tap][reg];
        }
}

// out[i] = sum_k slabs[k][i]; block = 64 float4 columns x 4 split lanes, 4 independent loads in flight per lane
// (device functions: the same code serves the one-layer kernels and the batched one; `blk` = block index inside the layer,
// `red` = 8 KB of LDS)
__device__ __forceinline__ void slab_reduce_block_f32(const float* __restrict__ slabs, float* __restrict__ out, int64_t n,
                                                      int nsplit, int64_t blk, f32x4 (*red)[64]) {
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i4 = blk * 64 + cl;
    const int64_t n4 = n >> 2;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (i4 < n4) {
        const f32x4* base = reinterpret_cast<const f32x4*>(slabs) + i4;
        int k = sl;
        for (; k + 12 < nsplit; k += 16) {
            a0 += base[(int64_t)k * n4];
            a1 += base[(int64_t)(k + 4) * n4];
            a2 += base[(int64_t)(k + 8) * n4];
            a3 += base[(int64_t)(k + 12) * n4];
        }
        for (; k < nsplit; k += 4) a0 += base[(int64_t)k * n4];
    }
    red[sl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && i4 < n4) reinterpret_cast<f32x4*>(out)[i4] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n,
                                                          int nsplit) {
    __shared__ f32x4 red[8][64];
    slab_reduce_block_f32(slabs, out, n, nsplit, blockIdx.x, red);
}

// Many splits of a SMALL result (the recomputed stem: 1024 slabs of 576 floats): the form above would run three blocks whose
// threads each walk 256 slabs (24 us).  Here a block is 4 float4 columns x 64 split lanes: 16 slabs per thread, four loads in
// flight, the 16 split lanes of a wave by shuffles and the four waves through LDS -- a fixed order again.
__global__ __launch_bounds__(256) void slab_reduce_tall_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n,
                                                               int nsplit) {
    __shared__ f32x4 red[4][4];
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2, wave = threadIdx.x >> 6;
    const int64_t n4 = n >> 2;
    const int64_t i4 = (int64_t)blockIdx.x * 4 + cl;
    const int64_t ic = i4 < n4 ? i4 : n4 - 1;
    const f32x4* base = reinterpret_cast<const f32x4*>(slabs) + ic;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int k = sl;
    for (; k + 192 < nsplit; k += 256) {
        a0 += base[(int64_t)k * n4];
        a1 += base[(int64_t)(k + 64) * n4];
        a2 += base[(int64_t)(k + 128) * n4];
        a3 += base[(int64_t)(k + 192) * n4];
    }
    for (; k < nsplit; k += 64) a0 += base[(int64_t)k * n4];
    f32x4 v = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
    if ((threadIdx.x & 63) < 4) red[wave][cl] = v;
    __syncthreads();
    if (threadIdx.x < 4 && i4 < n4) reinterpret_cast<f32x4*>(out)[i4] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// The same for slabs of block-scaled fp16 pairs (conv3x3_wgrad_mfma_v2<.., SLAB16>): dword (cp, tap, ci) of a slab = rows 2 cp,
// 2 cp + 1 of one (tap, ci) as fp16 values v * 2^s, with ONE power-of-two scale per (split, workgroup block of coblk x 64 filter
// rows / input channels); the inverse scales [nsplit][nby] (nby = (Cout / coblk) * (Cin / 64), block = cot * (Cin / 64) + cit) sit
// behind the slabs.  npair = (Cout / 2) * row dwords per slab, row = 9 * Cin (a multiple of 4: a 16-byte piece stays inside one
// row pair and one 64-channel block).
//
// Shape of a block: CL = 256 / SL columns (16-byte pieces) x SL split lanes.  The layers with FEW filter rows are the ones with MANY
// splits (a 64 x 64-channel layer: 4 608 pieces x 512 splits; 512 x 512: 294 912 x 8), so a fixed 64 x 4 block left the shallow layers
// with 72 workgroups whose threads each walked 128 slabs -- 41 us for 37 MB, against 9 us for the same bytes of a deep layer
// (profiles/r05_slab_reduce_shapes.md).  SL follows the split count (uh_slab16_lanes: about eight slabs per thread, at least 128
// contiguous bytes per split row of a wave); the sums stay in a fixed order: per thread as before, the SL lanes pairwise through LDS.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline int uh_slab16_lanes(int nsplit) { return nsplit >= 256 ? 32 : nsplit >= 128 ? 16 : nsplit >= 64 ? 8 : 4; }
__host__ __device__ inline int64_t uh_slab16_blocks(int64_t npair, int nsplit) {
    const int cl = 256 / uh_slab16_lanes(nsplit);
    return ((npair >> 2) + cl - 1) / cl;
}
template <int SL>
__device__ __forceinline__ void slab_reduce_block_f16pair(const unsigned* __restrict__ slabs, float* __restrict__ out,
                                                          int64_t npair, int row, int nsplit, int coblk, int64_t blk,
                                                          f32x4* red, float* sc_tab) {
    constexpr int CL = 256 / SL;
    const int cl = threadIdx.x % CL, sl = threadIdx.x / CL;
    const int64_t i4 = blk * CL + cl;
    const int64_t n4 = npair >> 2;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo, lo2 = lo, hi2 = lo;
    auto add = [](f32x4& l, f32x4& h, const u32x4 v, const float sc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (the two halves through 16-bit scalars: with __builtin_bit_cast(f16x2, v[e]) hipcc 7.2 loaded ONE dword of the piece and
            // fed the other three lanes of the packed fma from registers nobody had written)
            const unsigned w = v[e];
            const float a = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
            const float b = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
            l[e] = fmaf(a, sc, l[e]);                       // (a power of two: the product is exact, the fma rounds the sum once)
            h[e] = fmaf(b, sc, h[e]);
        }
    };
    const int Cin = row / 9, nci = Cin >> 6;
    const int nby = nci * (int)((npair * 2 / row) / coblk);
    const float* scales_all = reinterpret_cast<const float*>(slabs + (int64_t)nsplit * npair);
    // The inverse scales of the workgroup's output-channel block -- [nsplit][nci] floats, at most 512 -- are staged in LDS once: read
    // one by one from memory they doubled the kernel's load instructions (15 us per launch against the bf16 form's 9.8).  A
    // workgroup's 256 dwords lie in one or two row pairs; a piece whose row pair belongs to the NEXT block takes the memory path.
    const int64_t cp_first = ((blk * CL) << 2) / row;
    const int cot0 = (int)(2 * cp_first) / coblk;
    const bool staged = nsplit * nci <= 512;
    if (staged) {
        for (int i = threadIdx.x; i < nsplit * nci; i += 256) {
            const int k = i / nci, c = i - k * nci;
            sc_tab[i] = scales_all[(int64_t)k * nby + cot0 * nci + c];
        }
    }
    __syncthreads();
    const int64_t e0 = (i4 < n4 ? i4 : 0) << 2;           // first dword of the piece
    const int64_t cp = e0 / row, rem = e0 - cp * row;
    if (i4 < n4) {
        const int tap = (int)(rem / Cin), ci = (int)(rem - (int64_t)tap * Cin);
        const int cot = (int)(2 * cp) / coblk, cit = ci >> 6;
        const bool from_lds = staged && cot == cot0;
        const float* scales = scales_all + cot * nci + cit;
        const float* tab = sc_tab + cit;
        auto scale_of = [&](int k) -> float { return from_lds ? tab[k * nci] : scales[(int64_t)k * nby]; };
        const u32x4* base = reinterpret_cast<const u32x4*>(slabs) + i4;
        int k = sl;
        for (; k + 3 * SL < nsplit; k += 4 * SL) {   // four independent loads in flight per lane, added in a fixed order
            const u32x4 v0 = base[(int64_t)k * n4], v1 = base[(int64_t)(k + SL) * n4];
            const u32x4 v2 = base[(int64_t)(k + 2 * SL) * n4], v3 = base[(int64_t)(k + 3 * SL) * n4];
            const float s0 = scale_of(k), s1 = scale_of(k + SL), s2 = scale_of(k + 2 * SL), s3 = scale_of(k + 3 * SL);
            add(lo, hi, v0, s0); add(lo2, hi2, v1, s1); add(lo, hi, v2, s2); add(lo2, hi2, v3, s3);
        }
        for (; k < nsplit; k += SL) add(lo, hi, base[(int64_t)k * n4], scale_of(k));
    }
    // red[half][sl][cl]: the SL partial sums of a column meet pairwise (lane s += lane s + h, h = SL/2 ... 1: a fixed tree)
    red[sl * CL + cl] = lo + lo2;
    red[(SL + sl) * CL + cl] = hi + hi2;
    __syncthreads();
#pragma unroll
    for (int h = SL / 2; h >= 2; h >>= 1) {
        if (sl < h) {
            red[sl * CL + cl] += red[(sl + h) * CL + cl];
            red[(SL + sl) * CL + cl] += red[(SL + sl + h) * CL + cl];
        }
        __syncthreads();
    }
    if (sl < 2 && i4 < n4) {                          // split lane 0 writes the even rows, lane 1 the odd ones
        const f32x4 v = red[(sl * SL) * CL + cl] + red[(sl * SL + 1) * CL + cl];
        *reinterpret_cast<f32x4*>(out + (2 * cp + sl) * row + rem) = v;
    }
}
template <typename F>
__device__ __forceinline__ void slab16_dispatch(int nsplit, F&& f) {
    switch (uh_slab16_lanes(nsplit)) {          // (block-uniform)
        case 32: f(std::integral_constant<int, 32>{}); break;
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 8: f(std::integral_constant<int, 8>{}); break;
        default: f(std::integral_constant<int, 4>{}); break;
    }
}

__global__ __launch_bounds__(256) void slab_reduce_f16pair_kernel(const unsigned* __restrict__ slabs, float* __restrict__ out,
                                                                  int64_t npair, int row, int nsplit, int coblk) {
    __shared__ f32x4 red[512];
    __shared__ float sc_tab[512];
    slab16_dispatch(nsplit, [&](auto sl) {
        slab_reduce_block_f16pair<decltype(sl)::value>(slabs, out, npair, row, nsplit, coblk, blockIdx.x, red, sc_tab);
    });
}

// Every pending slab reduction of a backward pass in ONE launch (uh_slab_reduce_batched): the filter gradients only feed the
// optimizer, so the per-layer reduce launches -- eighteen 8-14 us kernels in the middle of the backward stream -- can wait until
// the gradients are needed.  table[r] = { slabs, out, n, nsplit, format (0 fp32, 1 scaled fp16 pairs), row, first block, coblk } (int64).
constexpr int SLAB_TAB = 8;
__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(const int64_t* __restrict__ table, int nrows) {
    __shared__ f32x4 red[512];
    __shared__ float sc_tab[512];
    const int64_t b = blockIdx.x;
    int r = 0;
    for (int k = 1; k < nrows; ++k)                   // (a few dozen rows at most; scalar loads)
        if (table[k * SLAB_TAB + 6] <= b) r = k;
    const int64_t* t = table + r * SLAB_TAB;
    const int64_t blk = b - t[6];
    if (t[4] == 0)
        slab_reduce_block_f32(reinterpret_cast<const float*>(t[0]), reinterpret_cast<float*>(t[1]), t[2], (int)t[3], blk,
                              reinterpret_cast<f32x4(*)[64]>(red));
    else
        slab16_dispatch((int)t[3], [&](auto sl) {
            slab_reduce_block_f16pair<decltype(sl)::value>(reinterpret_cast<const unsigned*>(t[0]), reinterpret_cast<float*>(t[1]),
                                                           t[2] / 2, (int)t[5], (int)t[3], (int)t[7], blk, red, sc_tab);
        });
}

// =====================================================================================
// backward-weights v2 (bf16): same MFMA mapping as conv3x3_wgrad_mfma, but 8-row pixel tiles whose two
// [pixel][64 channel] LDS images (x halo 10x18, dy 8x16) are filled by LDS-DMA (buffer_load ... lds, zero
// padding through the descriptor range check) and double buffered: the DMA of tile t+1 is issued right after
// the barrier that retires tile t-1 and lands under the 72 MFMAs per wave of tile t.
// =====================================================================================
// NWR = number of 32-row output-channel blocks of the workgroup tile: 2 -> 64 co x 64 ci, 4 waves, two workgroups per CU;
// 4 -> 128 co x 64 ci, 8 waves, one workgroup per CU: the x halo image (the operand with the 1.4x halo) is staged once for
// twice the MFMA work, 30 % fewer DMA bytes / pieces per MFMA -- the kernel is bound by the rate at which L2 / Infinity
// Cache fill the LDS images (TCC hit rate 26 %, 36 GB/s per CU needed at the full MFMA rate), not by the matrix pipe.
// PRE: x0 is the RAW output of the previous conv; its BatchNorm + ReLU (the layer's real input, never stored: see
// conv3x3_fwd_mfma_v2) is applied to the x image in LDS by the thread that issued the DMA piece, behind the wave's vmcnt(0) and in
// front of the tile's barrier.  LDS is full (two workgroups x two stages = 160 KiB), so the 64 (scale, shift) pairs of the
// workgroup's input-channel slab live in ONE register pair spread over the lanes (lane = channel) and reach the lane that needs
// them through ds_bpermute (the LDS crossbar, no LDS memory).
// SLAB16: the per-split partial results go to the workspace as 16-bit PAIRS (u32 = rows co, co + 1 of one (tap, ci)) instead of
// fp32: half the 75 MB a launch writes and slab_reduce reads back.  Each partial is a sum over >= one tile of pixels accumulated
// in fp32 by the MFMAs and rounded ONCE, to fp16 (11 significant bits, rms relative error 2^-12.8) after multiplication by a power
// of two chosen per workgroup so that its largest |value| lands in [2^14, 2^15): block-scaled fp16.  slab_reduce multiplies back
// and adds in fp32; the result stays fp32.  (Round 4 stored bf16: 8 significant bits.  The partials of a filter gradient can be
// an order of magnitude larger than their sum -- a gradient whose sign follows image regions sums to zero over the batch after
// BatchNorm backward, while a tile lies inside one region -- and the error then was that much above the single bf16 rounding of
// the TOTAL the reference's autocast backward performs (train.py:116; conv backward returns the filter gradient in the dtype of
// the bf16 filter copy): 8-13 x on synthetic region-signed gradients, up to 2.7 x on one layer of a trained UNet, 0.1-0.9 x
// elsewhere (scratch/r5_slab_structured.py).  Three more bits put all of them at or below the reference's own rounding.)
// fp32 slabs: UH_WGRAD_SLAB_F32=1.
template <typename T, int NWR, bool PRE = false, bool SLAB16 = false>
__global__ __launch_bounds__(128 * NWR, 2) void conv3x3_wgrad_mfma_v2(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x0, int C0, int ld0, const T* __restrict__ x1, int C1,
    int ld1, float* __restrict__ slabs, int Cout, int B, int H, int W, int tilesX, int tilesY, int nsplit,
    unsigned dy_bytes, unsigned x_bytes, int C0v, int C1v, int Coutv, const float* __restrict__ pre_scale = nullptr,
    const float* __restrict__ pre_shift = nullptr) {
    static_assert(sizeof(T) == 2, "v2 is the bf16 kernel");
    constexpr int TH = 8;
    constexpr int NT = 128 * NWR;               // threads
    constexpr int PB = 128;                     // x bytes per pixel: 64 bf16 channels
    constexpr int DPB = 64 * NWR;               // dy bytes per pixel: 32 * NWR bf16 channels
    constexpr int DU = DPB / 16;                // 16-byte units per dy pixel (8 / 16)
    constexpr int XPIX = (TH + 2) * HALO_W;     // 180
    constexpr int DPIX = TH * TILE;             // 128
    constexpr int XUNITS = XPIX * 8;            // 16-byte units of the x image
    constexpr int XR = (XUNITS + NT - 1) / NT;  // DMA rounds: 6 (256 threads) / 3 (512 threads)
    constexpr int DR = DPIX * DU / NT;          // 4
    // LDS stage = x halo image (6 DMA rounds of 256 threads x 16 B, the tail of the sixth is padding) + dy image (4 rounds):
    // every thread issues exactly 10 pieces per tile, so that the hand-placed vmcnt waits count the same on every wave
    constexpr int RB = NT * 16;                 // bytes per DMA round
    constexpr int XBYTES = XR * RB, STAGE = XBYTES + DR * RB;       // 24576 + 16384 (NWR 2: 2 stages x 2 workgroups = all 160 KiB) / 24576 + 32768
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;     // wr: 32-row co block (0 .. NWR-1), wc: 32-column ci block
    const int Cin = C0 + C1;
    const int nci = Cin / 64;
    const int cot = blockIdx.y / nci, cit = blockIdx.y - cot * nci;
    const int co0 = cot * (32 * NWR), ci0 = cit * 64;
    const T* xsrc; int ldx;
    int xleft;                                   // channels of this slab that exist in memory (narrow tensors)
    if (ci0 < C0) { xsrc = x0 + ci0; ldx = ld0; xleft = C0v - ci0; } else { xsrc = x1 + (ci0 - C0); ldx = ld1; xleft = C1v - (ci0 - C0); }
    const int dleft = Coutv - co0;
    const u32x4 rsx = uh_desc_words(xsrc, x_bytes);
    const u32x4 rsd = uh_desc_words(dy + co0, dy_bytes);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    const int ntile = B * tilesX * tilesY;
    const int split = blockIdx.x;
    const int t_begin = (int)(((int64_t)ntile * split) / nsplit);
    const int t_end = (int)(((int64_t)ntile * (split + 1)) / nsplit);

    // per-thread DMA geometry that does not depend on the tile: halo coordinates of each 16-byte unit and its byte
    // offset relative to the tile's top-left halo pixel (tile-dependent part is one scalar base + 4 range checks)
    int xg[XR], xo[XR];
#pragma unroll
    for (int k = 0; k < XR; ++k) {
        const int p = tid + k * NT;
        const int q = p >> 3;
        const int hy = q / HALO_W, hx = q - hy * HALO_W;
        const int u = (p & 7) ^ (((hx >> 1) & 1) << 2);          // swizzle by halo COLUMN: row independent
        xg[k] = (p < XUNITS && u * 8 < xleft) ? ((hy << 8) | hx) : -1;       // units beyond the valid channels read as zeros
        xo[k] = (hy * W + hx) * ldx * 2 + u * 16;
    }
    // dy image [pixel][DPB]: the 64-byte groups of a pixel row are XOR-ed with the column so that the 4 consecutive pixels
    // of a transposed read hit 4 distinct bank groups (128-byte pitch: bit 1 of the column flips the two groups; 256-byte
    // pitch: the two low column bits permute the four groups)
    auto dswz = [&](int col) -> int { return NWR == 2 ? (((col >> 1) & 1) << 2) : ((col & 3) << 2); };     // in 16-byte units
    // A DMA round covers NT / DU = 32 pixels = two whole tile rows, so a thread's DR dy pieces sit in one column, 2 rows apart:
    // piece k = piece 0 + k * (a wave-uniform stride).  One (coordinate, offset) pair per thread instead of DR of them -- the
    // NWR = 2 instantiation spilled one of them and reloaded it, behind a vmcnt(0), in the middle of every tile's DMA issue
    // (an A/B of the two builds shows no time difference: the reload hid behind the other resident workgroup's MFMAs).
    static_assert(NT / DU == 2 * TILE, "dy DMA rounds are two tile rows apart");
    int dg0, dof0;
    {
        const int q = tid / DU, u = (tid % DU) ^ dswz(q & 15);
        dg0 = (u * 8 < dleft) ? (((q >> 4) << 8) | (q & 15)) : -1;
        dof0 = ((q >> 4) * W + (q & 15)) * lddy * 2 + u * 16;
    }
    const int dof_round = 2 * W * lddy * 2;       // bytes between rounds
    auto issue = [&](int tile, int bufi) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TH, x0p = txt * TILE;
        const int xbase = ((b * H + y0 - 1) * W + x0p - 1) * ldx * 2;     // may be "negative": only used when in range
        const int dbase = ((b * H + y0) * W + x0p) * lddy * 2;
        const unsigned xb = lds_base + bufi * STAGE + wave * 1024;
        const unsigned db = xb + XBYTES;
        // the 10 x 18 halo inside the image and the 8 x 16 dy tile complete: no per-piece range tests (wave-uniform)
        const bool inside = y0 >= 1 && y0 + TH + 1 <= H && x0p >= 1 && x0p + TILE + 1 <= W;
#pragma unroll
        for (int k = 0; k < XR; ++k) {
            bool ok = xg[k] >= 0;                     // (slots past the halo image / past the stored channels: out of range)
            if (!inside) {
                const int gy = y0 - 1 + (xg[k] >> 8), gx = x0p - 1 + (xg[k] & 255);
                ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            uh_dma16(rsx, xb + k * RB, ok ? (unsigned)(xbase + xo[k]) : OOB_OFFSET, 0);
        }
#pragma unroll
        for (int k = 0; k < DR; ++k) {
            bool ok = dg0 >= 0;
            if (!inside) {
                const int gy = y0 + 2 * k + (dg0 >> 8), gx = x0p + (dg0 & 255);
                ok = ok && gy < H && gx < W;
            }
            uh_dma16(rsd, db + k * RB, ok ? (unsigned)(dbase + k * dof_round + dof0) : OOB_OFFSET, 0);
        }
    };

#ifndef UH_WGRAD_M16
#define UH_WGRAD_M16 1      // 1: v_mfma_f32_16x16x32_bf16 (k = the 16 pixels of rows p and p+4), 0: v_mfma_f32_32x32x16_bf16 (k = one row)
#endif
    const int l16 = lane & 15;
    const int rq = l16 >> 2, cp = l16 & 3;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#if UH_WGRAD_M16
    // The chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md,
    // DVFS give-back item 7), and this kernel runs at the power limit.  A wave's 32 x 32 (co, ci) block = 2 x 2 MFMA tiles;
    // the contraction index of one MFMA = 32 pixels = the 16 columns of tile rows p and p + 4 (so that the x operand of
    // tap row r is the row pair (p + r, p + r + 4): six pairs cover the ten halo rows).  Lane group g = lane >> 4 holds
    // the 8 pixels (row p + 4 * (g >> 1), columns 8 * (g & 1) ...) of channel lane & 15 -- two transposed LDS reads.
    f32x4 acc[9][2][2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) acc[k][h][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g4 = lane >> 4;
    const int rowsel = g4 >> 1, colh = g4 & 1;
    const int a_cbyte = (wr * 32 + cp * 4) * 2;        // + 32 bytes for the second 16-channel half (bit 5: below the swizzle bits)
    const int b_cbyte = (wc * 32 + cp * 4) * 2;
#else
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    const int grp = (lane >> 4) & 1;
    const int kh = lane >> 5;
    const int a_cbyte = (wr * 32 + grp * 16 + cp * 4) * 2;
    const int b_cbyte = (wc * 32 + grp * 16 + cp * 4) * 2;
#endif
    // Lane-constant LDS byte offsets of the two transposed reads of a fragment (pixels c, c+4 of a row); the halves
    // of a 128-byte pixel row are swapped by ((column >> 1) & 1), so 4 consecutive pixels hit 4 distinct bank groups.
    auto col_off = [&](int col, int cbyte) -> int { return col * PB + (cbyte ^ (((col >> 1) & 1) << 6)); };
    auto dcol_off = [&](int col, int cbyte) -> int { return col * DPB + (cbyte ^ (dswz(col) << 4)); };
#if UH_WGRAD_M16
    const int d_lo = rowsel * 4 * (TILE * DPB) + dcol_off(colh * 8 + rq, a_cbyte);
    const int d_hi = rowsel * 4 * (TILE * DPB) + dcol_off(colh * 8 + rq + 4, a_cbyte);
    int x_lo[3], x_hi[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        x_lo[s] = rowsel * 4 * (HALO_W * PB) + col_off(s + colh * 8 + rq, b_cbyte);
        x_hi[s] = rowsel * 4 * (HALO_W * PB) + col_off(s + colh * 8 + rq + 4, b_cbyte);
    }
#else
    const int d_lo = dcol_off(kh * 8 + rq, a_cbyte), d_hi = dcol_off(kh * 8 + rq + 4, a_cbyte);
    int x_lo[3], x_hi[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        x_lo[s] = col_off(s + kh * 8 + rq, b_cbyte);
        x_hi[s] = col_off(s + kh * 8 + rq + 4, b_cbyte);
    }
#endif

    // PRE: lane l of every wave holds (scale, shift) of channel ci0 + l of source 0 (loaded -- and waited for -- before the first
    // asynchronous DMA is in flight, so that the compiler's own wait for these two loads cannot miscount)
    float pre_s = 0.f, pre_h = 0.f;
    const bool pre_on = PRE && ci0 < C0;
    if constexpr (PRE) {
        if (pre_on) { pre_s = pre_scale[ci0 + lane]; pre_h = pre_shift[ci0 + lane]; }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre_s), "+v"(pre_h) :: "memory");
    }
    auto pre_x = [&](int tile, int bufi_t) {
        if constexpr (PRE) {
            if (!pre_on) return;
            int t = tile;
            const int txt = t % tilesX; t /= tilesX;
            const int tyt = t % tilesY;
            const int y0 = tyt * TH, x0p = txt * TILE;
            const bool inside = y0 >= 1 && y0 + TH + 1 <= H && x0p >= 1 && x0p + TILE + 1 <= W;
            unsigned char* xs_t = lds + bufi_t * STAGE;
#pragma unroll
            for (int k = 0; k < XR; ++k) {
                bool ok = xg[k] >= 0;
                const int hy = xg[k] >> 8, hx = xg[k] & 255;
                if (!inside) ok = ok && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0p - 1 + hx) < (unsigned)W;
                // every lane takes part in the permutes (they are cross-lane), only valid pieces are rewritten
                const int p = tid + k * NT;
                const int u = (p & 7) ^ (((hx >> 1) & 1) << 2);
                // (two channel quads in turn: 8 coefficients + 4 values live at a time -- the kernel runs at the register limit)
                u32x4* slot = reinterpret_cast<u32x4*>(xs_t + p * 16);
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) v = *slot;
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    float sc[4], sh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sc[e] = __int_as_float(__builtin_amdgcn_ds_bpermute((u * 8 + hq * 4 + e) * 4, __float_as_int(pre_s)));
                        sh[e] = __int_as_float(__builtin_amdgcn_ds_bpermute((u * 8 + hq * 4 + e) * 4, __float_as_int(pre_h)));
                    }
                    const float r0 = uh_relu(fmaf(__uint_as_float(v[2 * hq] << 16), sc[0], sh[0]));
                    const float r1 = uh_relu(fmaf(__uint_as_float(v[2 * hq] & 0xffff0000u), sc[1], sh[1]));
                    const float r2 = uh_relu(fmaf(__uint_as_float(v[2 * hq + 1] << 16), sc[2], sh[2]));
                    const float r3 = uh_relu(fmaf(__uint_as_float(v[2 * hq + 1] & 0xffff0000u), sc[3], sh[3]));
                    const bf16x4 o = {(bf16_t)r0, (bf16_t)r1, (bf16_t)r2, (bf16_t)r3};
                    const u32x2 ou = __builtin_bit_cast(u32x2, o);
                    v[2 * hq] = ou[0];
                    v[2 * hq + 1] = ou[1];
                }
                if (ok) *slot = v;
            }
        }
    };
    // end of a tile: the DMA of the next tile has landed and every wave has finished reading this buffer
    auto tile_fence = [&](int next_tile, int bufi_next, bool live_next) {
        if constexpr (PRE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (live_next) pre_x(next_tile, bufi_next);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // The LDS-DMA is issued through inline asm (uh_dma16): the compiler does not see it, so it does not drain it in front
    // of the tile's first ds_read (which it does for the builtin -- the DMA of tile t+1 then overlapped nothing); it is
    // waited for by hand at the END of tile t, behind its 72 MFMAs per wave.
    auto tr_pair = [&](const unsigned char* row, int lo, int hi) -> bf16x8 {
        s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(row + lo));
        s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(row + hi));
        s16x8 both = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(bf16x8, both);
    };
#ifndef UH_WGRAD_ROT
#define UH_WGRAD_ROT 1      // 1: the tile loop is rotated (see below); 0: one barrier at the very end of a tile (round 3)
#endif
#if UH_WGRAD_M16
    // (the consumer-side BatchNorm variants -- UH_BUILD_PRE -- keep the round-3 loop: their rewrite of the x image sits inside the
    // fence, and inside the rotated MFMA stream it spills)
    if constexpr (UH_WGRAD_ROT && !PRE) {
    // ROTATED tile loop.  A tile = six x row pairs a (fragments of pair a + 1 are fetched while pair a is multiplied; dy pair
    // p = a - r meets tap row r); the last LDS reads of a tile are those of pair 5, requested in front of pair 4's MFMAs.  The
    // end-of-tile fence (next tile's DMA landed, everyone done READING this buffer) therefore sits in front of pair 5's twelve
    // MFMAs, not behind them, and what used to open the next tile -- its first sixteen transposed reads -- is requested between
    // the fence and those twelve MFMAs: when the barrier releases the workgroup, every SIMD has two waves with matrix work in
    // hand instead of two waves waiting for LDS
    // (profiles/r03_wgrad_phase_stamps.txt: the pipe was busy 0.67 of the tile loop; the start-of-tile reads of eight waves
    // alone are ~500 LDS cycles).  Fragments of the next tile cross the loop back edge in dfr[0] / xfr[0].
    bf16x8 dfr[4][2];
    bf16x8 xfr[6][3][2];
    auto ld_x = [&](const unsigned char* xs_, int a) {
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                xfr[a][sft][hh] = tr_pair(xs_ + a * (HALO_W * PB) + hh * 32, x_lo[sft], x_hi[sft]);
    };
    auto ld_d = [&](const unsigned char* ds_, int pr) {
#pragma unroll
        for (int h = 0; h < 2; ++h) dfr[pr][h] = tr_pair(ds_ + pr * (TILE * DPB) + h * 32, d_lo, d_hi);
    };
    auto mma_pair = [&](int a) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int pr = a - r;
            if (pr >= 0 && pr < 4) {
#pragma unroll
                for (int sft = 0; sft < 3; ++sft)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            acc[r * 3 + sft][h][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[pr][h], xfr[a][sft][hh],
                                                                                             acc[r * 3 + sft][h][hh], 0, 0, 0);
            }
        }
    };
    if (t_begin < t_end) issue(t_begin, 0);
    tile_fence(t_begin, 0, t_begin < t_end);
    ld_d(lds + XBYTES, 0);
    ld_x(lds, 0);
    int bufi = 0;
    for (int tile = t_begin; tile < t_end; ++tile, bufi ^= 1) {
        const unsigned char* xs = lds + bufi * STAGE;
        const unsigned char* ds = xs + XBYTES;
        // the next tile's DMA, into the buffer the last fence released (behind this tile's first fragment reads, which were
        // issued in front of pair 5 of the previous tile; issuing it there as well -- with pair 5's fragments still live --
        // spilled ~30 registers into the MFMA stream)
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 1 < t_end) issue(tile + 1, bufi ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            ld_x(xs, a + 1);
            if (a + 1 < 4) ld_d(ds, a + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_pair(a);
        }
        // every LDS read of this tile has been issued (pair 5's in front of pair 4's MFMAs)
        tile_fence(tile + 1, bufi ^ 1, tile + 1 < t_end);
        {
            // (unconditional: behind the last tile these sixteen reads fetch stale LDS contents nobody uses -- a branch here
            // makes the fragments phi nodes of the loop and costs ~20 spilled registers around the fence)
            const unsigned char* xn = lds + (bufi ^ 1) * STAGE;
            ld_d(xn + XBYTES, 0);
            ld_x(xn, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(5);
    }
    } else
#endif
    {
    if (t_begin < t_end) issue(t_begin, 0);
    tile_fence(t_begin, 0, t_begin < t_end);
    int bufi = 0;
    for (int tile = t_begin; tile < t_end; ++tile, bufi ^= 1) {
#if !UH_WGRAD_M16
        if (tile + 1 < t_end) issue(tile + 1, bufi ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        const unsigned char* xs = lds + bufi * STAGE;
        const unsigned char* ds = xs + XBYTES;
#if UH_WGRAD_M16
        // fully unrolled over the six x row pairs (a, a + 4): the fragments of pair a + 1 are fetched while pair a is
        // multiplied; dy pair p = a - r meets tap row r
        bf16x8 dfr[4][2];
        bf16x8 xfr[6][3][2];
        auto ld_x = [&](int a) {
#pragma unroll
            for (int sft = 0; sft < 3; ++sft)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    xfr[a][sft][hh] = tr_pair(xs + a * (HALO_W * PB) + hh * 32, x_lo[sft], x_hi[sft]);
        };
        auto ld_d = [&](int pr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) dfr[pr][h] = tr_pair(ds + pr * (TILE * DPB) + h * 32, d_lo, d_hi);
        };
        ld_d(0);
        ld_x(0);
        // The next tile's DMA (ten pieces per thread, ~150 VALU instructions of address arithmetic) is issued HERE, behind the tile's
        // first fragment reads: at the top of the tile that arithmetic ran in front of them, with every wave of the workgroup just
        // released from the barrier and the matrix pipe empty; now it runs under the LDS latency of the reads the first MFMAs wait
        // for (profiles/r03_wgrad_phase_stamps.txt).  The buffer it writes was released by the barrier at the end of the last tile.
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 1 < t_end) issue(tile + 1, bufi ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (a + 1 < 6) ld_x(a + 1);
            if (a + 1 < 4) ld_d(a + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int pr = a - r;
                if (pr >= 0 && pr < 4) {
#pragma unroll
                    for (int sft = 0; sft < 3; ++sft)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh)
                                acc[r * 3 + sft][h][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[pr][h], xfr[a][sft][hh],
                                                                                                 acc[r * 3 + sft][h][hh], 0, 0, 0);
                }
            }
        }
#else
        const bf16x8 zero8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        // fully unrolled over the 10 halo rows: row addresses are immediates, the dy fragments rotate by renaming,
        // and the fragments of row hy+1 are fetched while row hy is multiplied
        bf16x8 dfr[TH + 2];
        bf16x8 xfr[TH + 2][3];
        dfr[0] = tr_pair(ds, d_lo, d_hi);
#pragma unroll
        for (int s = 0; s < 3; ++s) xfr[0][s] = tr_pair(xs, x_lo[s], x_hi[s]);
#pragma unroll
        for (int hy = 0; hy < TH + 2; ++hy) {
            if (hy + 1 < TH + 2) {
                dfr[hy + 1] = (hy + 1 < TH) ? tr_pair(ds + (hy + 1) * (TILE * DPB), d_lo, d_hi) : zero8;
#pragma unroll
                for (int s = 0; s < 3; ++s) xfr[hy + 1][s] = tr_pair(xs + (hy + 1) * (HALO_W * PB), x_lo[s], x_hi[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (hy - r >= 0 && hy - r < TH)
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[hy - r], xfr[hy][s], acc[r * 3 + s], 0, 0, 0);
        }
#endif
        tile_fence(tile + 1, bufi ^ 1, tile + 1 < t_end);
    }
    }

#if UH_WGRAD_M16
    if constexpr (SLAB16) {
        // ---- the workgroup's largest |partial| -> one power-of-two scale (the tile loop is over: LDS is scratch now; the only LDS
        // operations still in flight are the rotated loop's look-ahead reads, whose results nobody uses)
        float m = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(acc[tap][h][hh][j]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float* wmax = reinterpret_cast<float*>(lds);
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * NWR; ++k) m = fmaxf(m, wmax[k]);
        // biased exponent eb of the maximum, kept inside [15, 254] (zero / denormal maxima and infinities included): the values are
        // multiplied by 2^(14 - (eb - 127)), the reduce kernel multiplies by 2^(eb - 127 - 14).  A NaN partial stays a NaN.
        unsigned eb = (__float_as_uint(m) >> 23) & 0xffu;
        eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
        const float sc = __uint_as_float((268u - eb) << 23);
        unsigned* slab16 = reinterpret_cast<unsigned*>(slabs) + (int64_t)split * (Cout / 2) * 9 * Cin;
        if (tid == 0) {
            float* inv = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(slabs) + (int64_t)nsplit * (Cout / 2) * 9 * Cin);
            inv[(int64_t)split * gridDim.y + blockIdx.y] = __uint_as_float((eb - 14u) << 23);
        }
        // [co / 2][tap][ci] dwords: low half = row co (even), high half = row co + 1; a lane's accumulators j, j + 1 are such a pair
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int jp = 0; jp < 4; jp += 2) {
                        const int co = co0 + wr * 32 + h * 16 + (lane >> 4) * 4 + jp;
                        const int ci = ci0 + wc * 32 + hh * 16 + (lane & 15);
                        const f16x2 pr = {(_Float16)(acc[tap][h][hh][jp] * sc), (_Float16)(acc[tap][h][hh][jp + 1] * sc)};
                        slab16[((int64_t)(co >> 1) * 9 + tap) * Cin + ci] = __builtin_bit_cast(unsigned, pr);
                    }
        return;
    }
#endif
    float* slab = slabs + (int64_t)split * Cout * 9 * Cin;
#if UH_WGRAD_M16
    // acc[tap][h][hh][j] = dW[co0 + wr*32 + h*16 + (lane >> 4)*4 + j][tap][ci0 + wc*32 + hh*16 + (lane & 15)]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = co0 + wr * 32 + h * 16 + (lane >> 4) * 4 + j;
                    const int ci = ci0 + wc * 32 + hh * 16 + (lane & 15);
                    slab[((int64_t)co * 9 + tap) * Cin + ci] = acc[tap][h][hh][j];
                }
#else
    const int ci = ci0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int co = co0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            slab[((int64_t)co * 9 + tap) * Cin + ci] = acc[tap][reg];
        }
#endif
}

__global__ void slab_reduce_scalar_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n, int nsplit) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slabs[(int64_t)k * n + i];
    out[i] = s;
}

// ---- stem wgrad (Cin <= 4): lane = output channel; per-block partial [Cout][9][Cin]
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wgrad_stem(const T* __restrict__ dy, int lddy, const T* __restrict__ x,
                                                          int Cin, int ldx, float* __restrict__ slabs, int Cout, int B,
                                                          int H, int W, int tilesX, int tilesY, int nsplit) {
    __shared__ float xs[HALO_PIX * 4];
    __shared__ float red[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = B * tilesX * tilesY;
    const int split = blockIdx.x;
    const int t_begin = (int)(((int64_t)ntile * split) / nsplit);
    const int t_end = (int)(((int64_t)ntile * (split + 1)) / nsplit);
    const int cg = blockIdx.y * 64;
    const int co = cg + lane;
    const bool cok = co < Cout;
    float acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.f;
    for (int tile = t_begin; tile < t_end; ++tile) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TILE, x0p = txt * TILE;
        __syncthreads();
        for (int idx = tid; idx < HALO_PIX * 4; idx += 256) {
            int q = idx >> 2, ci = idx & 3;
            int hy = q / HALO_W, hx = q - hy * HALO_W;
            int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
            float v = 0.f;
            if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = uh_to_f32(x[(int64_t)((b * H + gy) * W + gx) * ldx + ci]);
            xs[idx] = v;
        }
        __syncthreads();
        for (int rr = 0; rr < 4; ++rr) {
            const int ty = wave * 4 + rr, gy = y0 + ty;
            if (gy >= H) break;
            for (int tx = 0; tx < TILE; ++tx) {
                const int gx = x0p + tx;
                if (gx >= W) break;
                float g = cok ? uh_to_f32(dy[(int64_t)((b * H + gy) * W + gx) * lddy + co]) : 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const float* xp = &xs[((ty + r) * HALO_W + tx + s) * 4];
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) acc[(r * 3 + s) * 4 + ci] = fmaf(g, xp[ci], acc[(r * 3 + s) * 4 + ci]);
                    }
            }
        }
    }
    float* slab = slabs + (int64_t)split * Cout * 9 * Cin;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        __syncthreads();
        red[wave * 64 + lane] = acc[k];
        __syncthreads();
        if (wave == 0 && cok) {
            int tap = k >> 2, ci = k & 3;
            if (ci < Cin) slab[((int64_t)co * 9 + tap) * Cin + ci] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        }
    }
}

// ---- stem wgrad v2 (Cin <= 4, Cout % V == 0): thread = (pixel lane, V output channels); dy is streamed once with
// 16-byte loads, the 9 input taps are scalar (cache resident); one input channel per blockIdx.z.
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wgrad_stem_v2(const T* __restrict__ dy, int lddy, const T* __restrict__ x,
                                                             int Cin, int ldx, float* __restrict__ slabs, int Cout, int B,
                                                             int H, int W, int nsplit) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ float red[4][9][64];              // [wave][tap][channel in pass]
    const int G = Cout / V;
    const int GB = G < 8 ? G : 8;
    const int PL = 256 / GB;
    const int tid = threadIdx.x, g = tid % GB, pl = tid / GB;
    const int ci = blockIdx.z;
    const int64_t npix = (int64_t)B * H * W;
    const int64_t p0 = npix * blockIdx.x / nsplit, p1 = npix * (blockIdx.x + 1) / nsplit;
    float* slab = slabs + (int64_t)blockIdx.x * Cout * 9 * Cin;
    for (int cb = 0; cb < Cout; cb += GB * V) {
        const bool cok = pl < PL && cb + g * V < Cout;
        float acc[9][V];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
        if (cok)
            for (int64_t p = p0 + pl; p < p1; p += PL) {
                const int wx = (int)(p % W);
                const int hy = (int)((p / W) % H);
                float d[V];
                uh_load<T, V>(dy + p * lddy + cb + g * V, d);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const int yy = hy + r - 1, xx = wx + s - 1;
                        float xv = 0.f;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                            xv = uh_to_f32(x[(p + (int64_t)(r - 1) * W + (s - 1)) * ldx + ci]);
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[r * 3 + s][i] = fmaf(d[i], xv, acc[r * 3 + s][i]);
                    }
            }
        // reduce over pixel lanes: lanes of a wave with equal g differ in bits >= log2(GB)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float v = acc[k][i];
                for (int o = GB; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                acc[k][i] = v;
            }
        __syncthreads();
        if ((tid & 63) < GB && pl < PL) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int i = 0; i < V; ++i) red[tid >> 6][k][g * V + i] = acc[k][i];
        }
        __syncthreads();
        for (int idx = tid; idx < 9 * GB * V; idx += 256) {
            const int k = idx / (GB * V), c = idx - k * (GB * V);
            if (cb + c < Cout)
                slab[((int64_t)(cb + c) * 9 + k) * Cin + ci] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
        }
    }
}

// ---- stem wgrad v3 (Cin <= 4, Cout == 64): persistent workgroups over 16x16 tiles, x halo in LDS, thread = (pixel lane
// of 32, 8 output channels) with 9*CIN*8 accumulators in registers; one slab per workgroup.  CIN is the number of input
// channels ONE workgroup handles (1 keeps the 72 accumulators in registers); blockIdx.y selects the input channel
// `ci0 = blockIdx.y * CIN` of the cin_total the layer has, so Cin = 2..4 re-reads dy per channel instead of spilling.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv3x3_wgrad_stem_v3(const T* __restrict__ dy, int lddy, const T* __restrict__ x,
                                                             int ldx, float* __restrict__ slabs, int B, int H, int W,
                                                             int tilesX, int tilesY, int cin_total) {
    constexpr int V = 8, COUT = 64;
    const int ci0 = blockIdx.y * CIN;
    __shared__ float xs[HALO_PIX * CIN];
    __shared__ float red[4][9 * CIN][COUT];
    const int tid = threadIdx.x, g = tid & 7, pl = tid >> 3;
    const int ntile = B * tilesX * tilesY;
    float acc[9 * CIN][V];
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int t = tile;
        const int txt = t % tilesX; t /= tilesX;
        const int tyt = t % tilesY;
        const int b = t / tilesY;
        const int y0 = tyt * TILE, x0p = txt * TILE;
        const int vy = min(TILE, H - y0), vx = min(TILE, W - x0p);
        // this thread's eight 16-byte pieces of dy are requested FIRST, branch-free (clamped to a pixel of the tile, masked
        // below), so they travel while the image tile is staged and the two barriers pass: a load inside each
        // `if (pixel in tile)` was waited for before the next one was issued -- eight exposed HBM round trips per tile
        static_assert(sizeof(T) == 2, "stem_v3 is the bf16 kernel");
        constexpr int DEPTH = 4;            // pieces in flight (a rolling window: 16 VGPRs; all eight cost an occupancy step)
        auto dy_piece = [&](int j) -> u32x4 {
            const int px = pl + j * 32, tyc = min(px >> 4, vy - 1), txc = min(px & 15, vx - 1);
            return *reinterpret_cast<const u32x4*>(dy + (int64_t)((b * H + y0 + tyc) * W + x0p + txc) * lddy + g * V);
        };
        u32x4 raw[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) raw[j] = dy_piece(j);
        __syncthreads();
        for (int idx = tid; idx < HALO_PIX * CIN; idx += 256) {
            int q = idx / CIN, ci = idx - q * CIN;
            int hy = q / HALO_W, hx = q - hy * HALO_W;
            int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = uh_to_f32(x[(int64_t)((b * H + gy) * W + gx) * ldx + ci0 + ci]);
            xs[idx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = pl + j * 32, ty = px >> 4, tx = px & 15;
            {
                const bool in = ty < vy && tx < vx;
                float d[V];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d[2 * q] = in ? __builtin_bit_cast(float, raw[j % DEPTH][q] << 16) : 0.f;
                    d[2 * q + 1] = in ? __builtin_bit_cast(float, raw[j % DEPTH][q] & 0xffff0000u) : 0.f;
                }
                if (j + DEPTH < 8) raw[j % DEPTH] = dy_piece(j + DEPTH);      // refill the slot just consumed
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int ss = 0; ss < 3; ++ss)
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) {
                            const float xv = xs[((ty + r) * HALO_W + tx + ss) * CIN + ci];
#pragma unroll
                            for (int i = 0; i < V; ++i) acc[(r * 3 + ss) * CIN + ci][i] = fmaf(d[i], xv, acc[(r * 3 + ss) * CIN + ci][i]);
                        }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float v = acc[k][i];
            for (int o = 8; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            acc[k][i] = v;
        }
    __syncthreads();
    if ((tid & 63) < 8) {
#pragma unroll
        for (int k = 0; k < 9 * CIN; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) red[tid >> 6][k][g * V + i] = acc[k][i];
    }
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * COUT * 9 * cin_total;
    for (int idx = tid; idx < 9 * CIN * COUT; idx += 256) {
        const int k = idx / COUT, c = idx - k * COUT;
        // slab layout [co][tap][ci]
        slab[((int64_t)c * 9 + k / CIN) * cin_total + ci0 + (k % CIN)] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// ---- generic wgrad: one block per (co, tap); threads stride over ci; serial over pixels
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wgrad_generic(const T* __restrict__ dy, int lddy, const T* __restrict__ x0,
                                                             int C0, int ld0, const T* __restrict__ x1, int C1, int ld1,
                                                             float* __restrict__ dw, int Cout, int B, int H, int W) {
    const int co = blockIdx.x / 9, tap = blockIdx.x - co * 9;
    const int r = tap / 3, s = tap - 3 * r;
    const int Cin = C0 + C1;
    // threads: (pixel lane, ci lane).  ci lanes = min(Cin, 256) rounded to a divisor layout
    int cl = Cin < 256 ? Cin : 256;
    int pl = 256 / cl;                 // pixel lanes
    int my_c = threadIdx.x % cl, my_p = threadIdx.x / cl;
    __shared__ float red[256];
    for (int cbase = 0; cbase < Cin; cbase += cl) {
        int ci = cbase + my_c;
        float acc = 0.f;
        if (my_p < pl && ci < Cin) {
            const T* src; int ld, cc;
            if (ci < C0) { src = x0; ld = ld0; cc = ci; } else { src = x1; ld = ld1; cc = ci - C0; }
            int64_t npix = (int64_t)B * H * W;
            for (int64_t p = my_p; p < npix; p += pl) {
                int wx = (int)(p % W);
                int hy = (int)((p / W) % H);
                int yy = hy + r - 1, xx = wx + s - 1;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                int64_t q = p + (int64_t)(r - 1) * W + (s - 1);
                acc = fmaf(uh_to_f32(dy[p * lddy + co]), uh_to_f32(src[q * ld + cc]), acc);
            }
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (my_p == 0 && ci < Cin) {
            float v = 0.f;
            for (int k = 0; k < pl; ++k) v += red[k * cl + my_c];
            dw[((int64_t)co * 9 + tap) * Cin + ci] = v;
        }
    }
}

// =====================================================================================
// host dispatch: wgrad
// =====================================================================================
struct WgradPlan { int kind; int nsplit; int tilesX, tilesY, ntile; int nwr; };   // kind 0 = mfma, 1 = stem, 2 = generic

// wide = the bf16 LDS-DMA kernel with 128-row output-channel tiles (8 waves, one workgroup per CU) may be used
template <typename T>
static WgradPlan wgrad_plan(int B, int H, int W, int Cin, int Cout, bool aligned, bool wide) {
    constexpr int TH = WgradCfg<T>::TH;
    WgradPlan p;
    p.nwr = 2;
    if (aligned && Cin % 64 == 0 && Cout % 64 == 0) {
        p.kind = 0;
        p.tilesX = (W + TILE - 1) / TILE; p.tilesY = (H + TH - 1) / TH;
        p.ntile = B * p.tilesX * p.tilesY;
        // UH_WGRAD_HALF_CU=1 (experiment): 4-wave workgroups, one per CU -- backward-weights then holds half of every CU and the
        // kernels of the main stream can run beside it
        static const bool half_cu = getenv("UH_WGRAD_HALF_CU") != nullptr;
        if (wide && sizeof(T) == 2 && Cout % 128 == 0 && !half_cu) p.nwr = 4;
        int ctiles = (Cin / 64) * (Cout / (32 * p.nwr));
        int total = (p.nwr == 4 || half_cu) ? 256 : 512;           // one 8-wave / two 4-wave workgroups per CU in flight (LDS)
        int want = (total + ctiles - 1) / ctiles;
        p.nsplit = want < 1 ? 1 : (want > p.ntile ? p.ntile : want);
    } else if (Cin <= 4) {
        p.kind = 1;
        p.tilesX = (W + TILE - 1) / TILE; p.tilesY = (H + TILE - 1) / TILE;
        p.ntile = B * p.tilesX * p.tilesY;
        int want = 1024;
        p.nsplit = want > p.ntile ? p.ntile : want;
    } else {
        p.kind = 2; p.nsplit = 0; p.tilesX = p.tilesY = p.ntile = 0;
    }
    return p;
}

extern "C" size_t uh_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt) {
    // alignment is unknown here: size for the slab paths (the generic path needs no workspace)
    WgradPlan p = (dt == UH_BF16) ? wgrad_plan<bf16_t>(B, H, W, Cin, Cout, true, false) : wgrad_plan<float>(B, H, W, Cin, Cout, true, false);
    if (p.kind == 2) return 16;
    size_t n = (size_t)p.nsplit * Cout * 9 * Cin * sizeof(float) + 16;
    if (dt == UH_BF16) {      // the wide-tile plan of the LDS-DMA kernel may use a different split count: the larger of the two
        WgradPlan q = wgrad_plan<bf16_t>(B, H, W, Cin, Cout, true, true);
        size_t m = (size_t)q.nsplit * Cout * 9 * Cin * sizeof(float) + 16;
        if (m > n) n = m;
    }
    return n;
}

template <typename T>
static int conv3x3_wgrad_dispatch(const T* dy, int lddy, const T* x0, int C0, int ld0, const T* x1, int C1, int ld1,
                                  float* dw, int Cout, void* ws, size_t ws_bytes, int B, int H, int W, hipStream_t st,
                                  bool split = false, int C0v = -1, int C1v = -1, int Coutv = -1,
                                  const float* pre_scale = nullptr, const float* pre_shift = nullptr, int64_t* defer = nullptr) {
    // defer != NULL (uh_conv3x3_wgrad_partials): the slab reduction is NOT launched; defer[0..7] describes it as a row of
    // uh_slab_reduce_batched's table (defer[3] = 0: the path taken has no slabs, dw is final)
    constexpr int ES = sizeof(T);
    const int Cin = C0 + C1;
    const bool narrow = C0v >= 0;                 // tensors hold fewer channels than the filter is padded to
    const bool pre = pre_scale != nullptr;        // x0 = raw conv output, BatchNorm + ReLU applied by the loader (bf16 DMA kernel only)
    if (!narrow) { C0v = C0; C1v = C1; Coutv = Cout; }
    const bool aligned = uh_aligned16(dy) && uh_aligned16(x0) && (C1 == 0 || uh_aligned16(x1)) && (lddy * ES) % 16 == 0 &&
                         (ld0 * ES) % 16 == 0 && (C1 == 0 || (ld1 * ES) % 16 == 0) && (C0 % 64 == 0);
    // the bf16 LDS-DMA kernel (and with it the 128-row tile) needs every tensor addressable through a buffer descriptor
    bool dma = false;
    if constexpr (ES == 2) {
        const int64_t npx_ = (int64_t)B * H * W;
        const int64_t ldmax_ = ld0 > ld1 ? ld0 : ld1;
        dma = npx_ * ldmax_ * 2 < (1ll << 31) - 4096 && npx_ * lddy * 2 < (1ll << 31) - 4096;
    }
    WgradPlan p = wgrad_plan<T>(B, H, W, Cin, Cout, aligned, dma && !narrow);
    if (p.kind == 1 && C1 != 0) p.kind = 2;
    if (narrow && p.kind != 0) {
        uh_set_error("uh_conv3x3_wgrad_narrow: needs the MFMA path (padded channel counts multiples of 64, 16-byte strides)");
        return UH_EINVAL;
    }
    if (split && p.kind != 0) {
        uh_set_error("uh_conv3x3_wgrad: bf16x3 needs an MFMA-aligned shape (channel counts multiples of 64, 16-byte strides)");
        return UH_EINVAL;
    }
#if !UH_BUILD_PRE
    if (pre) {
        uh_set_error("uh_conv3x3_wgrad_pre: the consumer-side BatchNorm+ReLU instantiations are not in this build (UH_BUILD_PRE=1); uh_conv3x3_pre_ok says so");
        return UH_EINVAL;
    }
#endif
    if (pre && !(ES == 2 && p.kind == 0 && dma && !narrow && !split)) {
        uh_set_error("uh_conv3x3_wgrad_pre: needs the bf16 LDS-DMA kernel (64-aligned channels, tensors below 2 GiB); ask uh_conv3x3_pre_ok first");
        return UH_EINVAL;
    }
    if (defer)
        for (int k = 0; k < 8; ++k) defer[k] = 0;
    if (p.kind == 2) {
        hipLaunchKernelGGL(conv3x3_wgrad_generic<T>, dim3(Cout * 9), dim3(256), 0, st, dy, lddy, x0, C0, ld0, x1, C1, ld1,
                           dw, Cout, B, H, W);
        UH_CHECK_LAUNCH("conv3x3_wgrad_generic");
        return UH_OK;
    }
    size_t need = (size_t)p.nsplit * Cout * 9 * Cin * sizeof(float);
    if (ws_bytes < need || !ws) {
        uh_set_error("uh_conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    float* slabs = (float*)ws;
    // block-scaled fp16-pair slabs: the bf16 LDS-DMA kernel's default (see SLAB16); the inverse scales follow the slabs in `ws`
    // (nsplit * n * 2 + nsplit * blocks * 4 bytes <= the nsplit * n * 4 the workspace is sized for)
    static const bool slab_f32 = getenv("UH_WGRAD_SLAB_F32") != nullptr && getenv("UH_WGRAD_SLAB_F32")[0] == '1';
    const bool slab16 = ES == 2 && p.kind == 0 && dma && !slab_f32 && UH_WGRAD_M16 && uh_aligned16(dw);
    if (p.kind == 0) {
        const int64_t npx = (int64_t)B * H * W;
        const int64_t ldmax = ld0 > ld1 ? ld0 : ld1;
        if constexpr (ES == 2) {
            if (dma) {
                // byte extents of the (sliced) source views as seen from their base pointers
                unsigned xb = (unsigned)(npx * ldmax * 2), db = (unsigned)(npx * lddy * 2);
#define UH_LAUNCH_WGRAD_V2(NWR_, PRE_, S16_)                                                                                      \
    hipLaunchKernelGGL((conv3x3_wgrad_mfma_v2<T, NWR_, PRE_, S16_>), dim3(p.nsplit, (Cin / 64) * (Cout / (32 * NWR_))),           \
                       dim3(128 * NWR_), 0, st, dy, lddy, x0, C0, ld0, x1, C1, ld1, slabs, Cout, B, H, W, p.tilesX, p.tilesY,      \
                       p.nsplit, db, xb, C0v, C1v, Coutv, pre_scale, pre_shift)
#if UH_BUILD_PRE
                if (pre) {
                    if (p.nwr == 4) { if (slab16) UH_LAUNCH_WGRAD_V2(4, true, true); else UH_LAUNCH_WGRAD_V2(4, true, false); }
                    else { if (slab16) UH_LAUNCH_WGRAD_V2(2, true, true); else UH_LAUNCH_WGRAD_V2(2, true, false); }
                } else
#endif
                if (p.nwr == 4) { if (slab16) UH_LAUNCH_WGRAD_V2(4, false, true); else UH_LAUNCH_WGRAD_V2(4, false, false); }
                else { if (slab16) UH_LAUNCH_WGRAD_V2(2, false, true); else UH_LAUNCH_WGRAD_V2(2, false, false); }
#undef UH_LAUNCH_WGRAD_V2
                UH_CHECK_LAUNCH("conv3x3_wgrad_mfma_v2");
            }
        }
        if (!dma) {
            if (split) {
                if constexpr (ES == 4)
                    hipLaunchKernelGGL((conv3x3_wgrad_mfma<T, true>), dim3(p.nsplit, (Cin / 64) * (Cout / 64)), dim3(256), 0, st, dy,
                                       lddy, x0, C0, ld0, x1, C1, ld1, slabs, Cout, B, H, W, p.tilesX, p.tilesY, p.nsplit, C0v,
                                       C1v, Coutv);
            } else
                hipLaunchKernelGGL(conv3x3_wgrad_mfma<T>, dim3(p.nsplit, (Cin / 64) * (Cout / 64)), dim3(256), 0, st, dy, lddy,
                                   x0, C0, ld0, x1, C1, ld1, slabs, Cout, B, H, W, p.tilesX, p.tilesY, p.nsplit, C0v, C1v, Coutv);
            UH_CHECK_LAUNCH("conv3x3_wgrad_mfma");
        }
    } else {
        constexpr int V = 16 / ES;
        bool done = false;
        if constexpr (ES == 2) {
            if (Cin <= 4 && Cout == 64 && uh_aligned16(dy) && (lddy * ES) % 16 == 0) {      // 72 accumulators per lane
                hipLaunchKernelGGL((conv3x3_wgrad_stem_v3<T, 1>), dim3(p.nsplit, Cin), dim3(256), 0, st, dy, lddy, x0, ld0, slabs,
                                   B, H, W, p.tilesX, p.tilesY, Cin);
                UH_CHECK_LAUNCH("conv3x3_wgrad_stem_v3");
                done = true;
            }
        }
        if (done) {
        } else if (Cout % V == 0 && uh_aligned16(dy) && (lddy * ES) % 16 == 0) {
            hipLaunchKernelGGL(conv3x3_wgrad_stem_v2<T>, dim3(p.nsplit, 1, Cin), dim3(256), 0, st, dy, lddy, x0, Cin, ld0,
                               slabs, Cout, B, H, W, p.nsplit);
            UH_CHECK_LAUNCH("conv3x3_wgrad_stem_v2");
        } else {
            hipLaunchKernelGGL(conv3x3_wgrad_stem<T>, dim3(p.nsplit, (Cout + 63) / 64), dim3(256), 0, st, dy, lddy, x0, Cin,
                               ld0, slabs, Cout, B, H, W, p.tilesX, p.tilesY, p.nsplit);
            UH_CHECK_LAUNCH("conv3x3_wgrad_stem");
        }
    }
    int64_t n = (int64_t)Cout * 9 * Cin;      // multiple of 4 on every slab path (Cout % 64 == 0 or 9*... stem: Cout*9*Cin)
    if (defer && n % 4 == 0 && uh_aligned16(dw) && uh_aligned16(slabs)) {
        defer[0] = (int64_t)(uintptr_t)slabs; defer[1] = (int64_t)(uintptr_t)dw; defer[2] = n; defer[3] = p.nsplit;
        defer[4] = slab16 ? 1 : 0; defer[5] = 9 * Cin; defer[7] = 32 * p.nwr;
        defer[6] = slab16 ? uh_slab16_blocks(n / 2, p.nsplit) : (n / 4 + 63) / 64;      // blocks of the reduction (the caller turns it into an offset)
        return UH_OK;
    }
    if (slab16) {
        const int64_t npair = n / 2;
        hipLaunchKernelGGL(slab_reduce_f16pair_kernel, dim3((unsigned)uh_slab16_blocks(npair, p.nsplit)), dim3(256), 0, st,
                           (const unsigned*)slabs, dw, npair, 9 * Cin, p.nsplit, 32 * p.nwr);
        UH_CHECK_LAUNCH("slab_reduce_f16pair_kernel");
        return UH_OK;
    }
    if (n % 4 != 0 || !uh_aligned16(dw) || !uh_aligned16(slabs))
        hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const float*)slabs, dw, n, p.nsplit);
    else
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, st, (const float*)slabs,
                           dw, n, p.nsplit);
    UH_CHECK_LAUNCH("slab_reduce_kernel");
    return UH_OK;
}

extern "C" int uh_conv3x3_wgrad(const void* dy, int lddy, const void* x0, int C0, int ld0, const void* x1, int C1, int ld1,
                                float* dw_krsc, int Cout, void* ws, size_t ws_bytes, int B, int H, int W, int dt,
                                uh_stream stream) {
    UH_REQUIRE(dy && x0 && dw_krsc, "uh_conv3x3_wgrad: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_wgrad: bad shape");
    UH_REQUIRE(lddy >= Cout && ld0 >= C0 && (C1 == 0 || (x1 && ld1 >= C1)), "uh_conv3x3_wgrad: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_wgrad: pixel count overflows int32");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_wgrad: bad dtype %d", dt);
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_BF16)
        return conv3x3_wgrad_dispatch<bf16_t>((const bf16_t*)dy, lddy, (const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1,
                                              ld1, dw_krsc, Cout, ws, ws_bytes, B, H, W, st);
    return conv3x3_wgrad_dispatch<float>((const float*)dy, lddy, (const float*)x0, C0, ld0, (const float*)x1, C1, ld1,
                                         dw_krsc, Cout, ws, ws_bytes, B, H, W, st, dt == UH_F32X3);
}

// uh_conv3x3_wgrad without its last step: the contraction runs, the per-split partial results stay in `ws` (which must live until
// the reduction has run) and desc[0..7] (HOST memory) receives the row of uh_slab_reduce_batched's table that finishes the job:
// { slabs, dw, n, nsplit, format, row, number of blocks, 0 } -- the caller replaces desc[6] by the row's first block (running sum)
// when it builds the table.  desc[3] == 0: this shape took a path without slabs, dw_krsc is already final.
extern "C" int uh_conv3x3_wgrad_partials(const void* dy, int lddy, const void* x0, int C0, int ld0, const void* x1, int C1,
                                         int ld1, float* dw_krsc, int Cout, void* ws, size_t ws_bytes, int B, int H, int W,
                                         int dt, int64_t* desc, uh_stream stream) {
    UH_REQUIRE(dy && x0 && dw_krsc && desc, "uh_conv3x3_wgrad_partials: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_wgrad_partials: bad shape");
    UH_REQUIRE(lddy >= Cout && ld0 >= C0 && (C1 == 0 || (x1 && ld1 >= C1)), "uh_conv3x3_wgrad_partials: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_wgrad_partials: pixel count overflows int32");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_wgrad_partials: bad dtype %d", dt);
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_BF16)
        return conv3x3_wgrad_dispatch<bf16_t>((const bf16_t*)dy, lddy, (const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1,
                                              ld1, dw_krsc, Cout, ws, ws_bytes, B, H, W, st, false, -1, -1, -1, nullptr, nullptr, desc);
    return conv3x3_wgrad_dispatch<float>((const float*)dy, lddy, (const float*)x0, C0, ld0, (const float*)x1, C1, ld1,
                                         dw_krsc, Cout, ws, ws_bytes, B, H, W, st, dt == UH_F32X3, -1, -1, -1, nullptr, nullptr, desc);
}

// table: DEVICE memory, nrows x 8 int64 (rows as uh_conv3x3_wgrad_partials fills them, [6] = first block of the row);
// total_blocks = sum of the rows' block counts.  One launch reduces every row: dw = sum over the splits, in the fixed order of
// the per-layer kernels (bit-identical results).
extern "C" int uh_slab_reduce_batched(const int64_t* table, int nrows, int64_t total_blocks, uh_stream stream) {
    UH_REQUIRE(table && nrows > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "uh_slab_reduce_batched: bad arguments");
    hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, nrows);
    UH_CHECK_LAUNCH("slab_reduce_batched_kernel");
    return UH_OK;
}

// Backward-weights of a layer whose forward was uh_conv3x3_fwd_pre: x0 is the RAW output of the previous conv and the layer's
// real input max(x0 * pre_scale + pre_shift, 0) is rebuilt by the loader (it was never stored).  Same result, bit for bit, as
// uh_conv3x3_wgrad on the stored activation.  Workspace: uh_conv3x3_wgrad_ws_bytes.
extern "C" int uh_conv3x3_wgrad_pre(const void* dy, int lddy, const void* x0, int C0, int ld0, const float* pre_scale,
                                    const float* pre_shift, float* dw_krsc, int Cout, void* ws, size_t ws_bytes, int B, int H,
                                    int W, int dt, uh_stream stream) {
    UH_REQUIRE(dy && x0 && dw_krsc && pre_scale && pre_shift, "uh_conv3x3_wgrad_pre: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && Cout > 0, "uh_conv3x3_wgrad_pre: bad shape");
    UH_REQUIRE(lddy >= Cout && ld0 >= C0, "uh_conv3x3_wgrad_pre: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_wgrad_pre: pixel count overflows int32");
    UH_REQUIRE(dt == UH_BF16, "uh_conv3x3_wgrad_pre: bf16 only (dtype %d)", dt);
    UH_REQUIRE(C0 % 64 == 0 && Cout % 64 == 0, "uh_conv3x3_wgrad_pre: channel counts must be multiples of 64");
    return conv3x3_wgrad_dispatch<bf16_t>((const bf16_t*)dy, lddy, (const bf16_t*)x0, C0, ld0, nullptr, 0, 0, dw_krsc, Cout, ws,
                                          ws_bytes, B, H, W, (hipStream_t)stream, false, -1, -1, -1, pre_scale, pre_shift);
}

// ---- the stem with a recomputed output (kernels: stem_mfma.hip): 1 -> 64 channels, bf16, w = KRSC pack [64][9][1]
// (single-channel images: the conv is a GEMM with K = 9, one K = 16 MFMA; an RGB stem keeps the stored path)
extern "C" int uh_stem_ok(int Cin, int Cout, int dt) { return (dt == UH_BF16 && Cin == 1 && Cout == 64) ? 1 : 0; }
extern "C" int uh_stem_nblk(int B, int H, int W) {
    const int64_t ntile = (int64_t)B * ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
    return (int)(ntile < 1024 ? ntile : 1024);
}

#define UH_STEM_COMMON(fn)                                                                                                    \
    UH_REQUIRE(x && w, fn ": null pointer");                                                                                  \
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && ldx >= Cin, fn ": bad shape");                                                      \
    UH_REQUIRE(uh_stem_ok(Cin, 64, dt), fn ": bf16, one input channel, 64 output channels (uh_stem_ok)");                   \
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), fn ": pixel count overflows int32")

extern "C" int uh_stem_stats(const void* x, int Cin, int ldx, const void* w, float* stat_partials, int B, int H, int W, int dt,
                             uh_stream stream) {
    UH_STEM_COMMON("uh_stem_stats");
    UH_REQUIRE(stat_partials, "uh_stem_stats: null statistics buffer");
    return uh_stem_mfma_launch(0, x, ldx, w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, nullptr, 0,
                               stat_partials, B, H, W, uh_stem_nblk(B, H, W), stream);
}

extern "C" int uh_stem_bn_relu_fwd(const void* x, int Cin, int ldx, const void* w, const float* scale, const float* shift, void* z,
                                   int ldz, int B, int H, int W, int dt, uh_stream stream) {
    UH_STEM_COMMON("uh_stem_bn_relu_fwd");
    UH_REQUIRE(scale && shift && z && ldz >= 64, "uh_stem_bn_relu_fwd: null pointer / bad stride");
    UH_REQUIRE(uh_aligned16(z) && (ldz * 2) % 16 == 0, "uh_stem_bn_relu_fwd: z must be 16-byte aligned with a 16-byte pixel pitch");
    return uh_stem_mfma_launch(1, x, ldx, w, scale, shift, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, z, ldz, nullptr, B, H, W,
                               uh_stem_nblk(B, H, W), stream);
}

extern "C" int uh_stem_bn_relu_bwd_reduce(const void* dz, int lddz, const void* x, int Cin, int ldx, const void* w,
                                          const float* scale, const float* shift, const float* mean, const float* rstd,
                                          float* partials, int B, int H, int W, int dt, uh_stream stream) {
    UH_STEM_COMMON("uh_stem_bn_relu_bwd_reduce");
    UH_REQUIRE(dz && scale && shift && mean && rstd && partials && lddz >= 64, "uh_stem_bn_relu_bwd_reduce: null pointer / bad stride");
    UH_REQUIRE(uh_aligned16(dz) && (lddz * 2) % 16 == 0, "uh_stem_bn_relu_bwd_reduce: dz must be 16-byte aligned with a 16-byte pixel pitch");
    return uh_stem_mfma_launch(2, x, ldx, w, scale, shift, mean, rstd, nullptr, nullptr, 0.f, dz, lddz, nullptr, 0, partials, B, H, W,
                               uh_stem_nblk(B, H, W), stream);
}

extern "C" size_t uh_stem_bwd_wgrad_ws_bytes(int B, int H, int W, int Cin) {
    return (size_t)uh_stem_nblk(B, H, W) * 64 * 9 * (Cin > 0 ? Cin : 1) * sizeof(float) + 16;
}

extern "C" int uh_stem_bn_relu_bwd_wgrad(const void* dz, int lddz, const void* x, int Cin, int ldx, const void* w,
                                         const float* scale, const float* shift, const float* mean, const float* rstd,
                                         const float* dgamma, const float* dbeta, int64_t n_total, float* dw_krsc, void* ws,
                                         size_t ws_bytes, int B, int H, int W, int dt, uh_stream stream) {
    UH_STEM_COMMON("uh_stem_bn_relu_bwd_wgrad");
    UH_REQUIRE(dz && scale && shift && mean && rstd && dgamma && dbeta && dw_krsc && ws && lddz >= 64,
               "uh_stem_bn_relu_bwd_wgrad: null pointer / bad stride");
    UH_REQUIRE(uh_aligned16(dz) && (lddz * 2) % 16 == 0, "uh_stem_bn_relu_bwd_wgrad: dz must be 16-byte aligned with a 16-byte pixel pitch");
    UH_REQUIRE(ws_bytes >= uh_stem_bwd_wgrad_ws_bytes(B, H, W, Cin), "uh_stem_bn_relu_bwd_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const float inv_n = (float)(1.0 / (double)(n_total > 0 ? n_total : (int64_t)B * H * W));
    float* slabs = (float*)ws;
    const int rc = uh_stem_mfma_launch(3, x, ldx, w, scale, shift, mean, rstd, dgamma, dbeta, inv_n, dz, lddz, nullptr, 0, slabs, B, H, W,
                                       uh_stem_nblk(B, H, W), stream);
    if (rc != UH_OK) return rc;
    const int64_t n = (int64_t)64 * 9 * Cin;
    const int nsplit = uh_stem_nblk(B, H, W);
    if (n % 4 != 0 || !uh_aligned16(dw_krsc) || !uh_aligned16(slabs))
        hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)slabs, dw_krsc, n, nsplit);
    else if (n <= 4096 && nsplit >= 256)
        hipLaunchKernelGGL(slab_reduce_tall_kernel, dim3((unsigned)((n / 4 + 3) / 4)), dim3(256), 0, st, (const float*)slabs, dw_krsc, n, nsplit);
    else
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, st, (const float*)slabs, dw_krsc, n, nsplit);
    UH_CHECK_LAUNCH("slab_reduce_kernel");
    return UH_OK;
}

// Narrow variant (see uh_conv3x3_fwd_narrow): the filter gradient is that of the PADDED layer [Cout][3][3][C0 + C1]; x0 / x1 /
// dy hold only their first C0v / C1v / Coutv channels per pixel, the rest count as zeros (their dW rows come out 0).
extern "C" int uh_conv3x3_wgrad_narrow(const void* dy, int lddy, int Cout, int Coutv, const void* x0, int C0, int C0v, int ld0,
                                       const void* x1, int C1, int C1v, int ld1, float* dw_krsc, void* ws, size_t ws_bytes,
                                       int B, int H, int W, int dt, uh_stream stream) {
    UH_REQUIRE(dy && x0 && dw_krsc, "uh_conv3x3_wgrad_narrow: null pointer");
    UH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "uh_conv3x3_wgrad_narrow: bad shape");
    UH_REQUIRE(C0v > 0 && C0v <= C0 && C1v >= 0 && C1v <= C1 && Coutv > 0 && Coutv <= Cout, "uh_conv3x3_wgrad_narrow: bad valid counts");
    UH_REQUIRE(lddy >= Coutv && ld0 >= C0v && (C1 == 0 || (x1 && ld1 >= C1v)), "uh_conv3x3_wgrad_narrow: bad strides");
    UH_REQUIRE((int64_t)B * H * W < (1ll << 31), "uh_conv3x3_wgrad_narrow: pixel count overflows int32");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16 || dt == UH_F32X3, "uh_conv3x3_wgrad_narrow: bad dtype %d", dt);
    const int vec = dt == UH_BF16 ? 8 : 4;
    UH_REQUIRE(C0v % vec == 0 && C1v % vec == 0 && Coutv % vec == 0, "uh_conv3x3_wgrad_narrow: valid counts must be multiples of a 16-byte piece");
    hipStream_t st = (hipStream_t)stream;
    if (dt == UH_BF16)
        return conv3x3_wgrad_dispatch<bf16_t>((const bf16_t*)dy, lddy, (const bf16_t*)x0, C0, ld0, (const bf16_t*)x1, C1,
                                              ld1, dw_krsc, Cout, ws, ws_bytes, B, H, W, st, false, C0v, C1v, Coutv);
    return conv3x3_wgrad_dispatch<float>((const float*)dy, lddy, (const float*)x0, C0, ld0, (const float*)x1, C1, ld1,
                                         dw_krsc, Cout, ws, ws_bytes, B, H, W, st, dt == UH_F32X3, C0v, C1v, Coutv);
}
