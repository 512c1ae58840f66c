// stem_mfma.hip -- the network's first layer, Conv2d(1 -> 64, 3x3, pad 1) -> BatchNorm2d -> ReLU (unet_parts.py:15-17 as
// instantiated by unet_model.py:15 on a single-channel image), with the conv output RECOMPUTED by every consumer instead of
// stored (see conv3x3.hip "Stem with a RECOMPUTED output"), the recomputation on the matrix pipe.
//
// y[pixel][channel] = sum over 9 taps is a GEMM with K = 9: one v_mfma_f32_16x16x16_bf16 (K padded to 16) produces 16 pixels
// x 16 channels, four of them a pixel row segment x all 64 channels.  With the filter rows handed to the MFMAs in the order
// channel(n, m) = 16 (m >> 2) + 4 n + (m & 3), lane (pixel lx, group kg) ends up with the 16 CONTIGUOUS channels 16 kg .. 16 kg + 15
// of its pixel: two 16-byte pieces of the NHWC tensor, so dz is loaded and z is stored in whole 2 KiB runs per wave.  What the
// VALU form spent on the conv (72 multiply-adds + 9 LDS reads per lane and pixel) becomes 4 MFMAs per 16 pixels.
//   MODE 0  statistics rows of round_bf16(y)                       (feeds uh_bn_finalize; nothing else is written)
//   MODE 1  z = max(round_bf16(y) * scale + shift, 0)
//   MODE 2  partial sums {sum m, sum m * xhat}, m = dz [z > 0]     (finish with uh_bn_bwd_finalize)
//   MODE 3  dy = round_bf16(a m + b y + k) in registers; dW[c][tap] = sum_pixels dy[pixel][c] * x[pixel + tap] is a GEMM too
//           (M = channels, N = 9 taps, K = the 16 pixels of a row): dy goes through a per-wave LDS tile to become the A operand
//           (pixels along K), the B operand is gathered from the image halo -> 16 accumulator registers; slabs, reduced by the caller
// The MFMA adds the nine products in an order of its own: y can differ from the serial-FMA kernels (conv3x3_fwd_stem_v3) in the
// last fp32 bit, i.e. in one bf16 ulp for about one element in 10^4.  All four modes use the SAME arithmetic, so the ReLU mask
// and xhat of the backward pass are exactly those of the forward pass.
#include "uh_common.h"
#include <type_traits>

namespace {

constexpr int ST_TILE = 16, ST_HALO = 18, ST_PITCH = 20;       // halo rows of 18 bf16 at a 20-halfword pitch
constexpr int ST_C = 64;

struct StemArgs {
    const bf16_t* x; int ldx;                 // image [B][H][W][>= 1], channel 0 is used
    const bf16_t* w;                          // KRSC pack [64][9][1]
    const float* scale; const float* shift; const float* mean; const float* rstd;
    const float* dgamma; const float* dbeta; float inv_n;
    const bf16_t* dz; int lddz;               // MODE 2 / 3
    bf16_t* z; int ldz;                       // MODE 1
    float* out;                               // MODE 0: statistics rows + counts; MODE 2: partials; MODE 3: slabs
    int B, H, W, tilesX, tilesY;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(StemArgs a) {
    __shared__ unsigned short xs[ST_HALO * ST_PITCH + 4];
    __shared__ float red[4][MODE == 3 ? 9 : 2][ST_C];
    __shared__ float pivot[ST_C];
    __shared__ __attribute__((aligned(16))) float cf[4][ST_C];
    constexpr int DP = 68;                     // halfwords per pixel of the dy tile: 64 channels + 4 (8-byte aligned rows, the four
                                               // lane groups of a transposed read land in four different bank octets)
    __shared__ __attribute__((aligned(16))) unsigned short dyt[MODE == 3 ? 4 * 16 * DP : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 15, kg = lane >> 4;
    const int ntile = a.B * a.tilesX * a.tilesY;
    const int H = a.H, W = a.W;

    // ---- MFMA operands that do not change: the filter (A) and the halo offsets of this lane's four taps (B)
    s16x4 wA[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int c = 16 * (lx >> 2) + 4 * n + (lx & 3);
        short v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * kg + e;
            v[e] = k < 9 ? __builtin_bit_cast(short, a.w[c * 9 + k]) : (short)0;
        }
        wA[n] = s16x4{v[0], v[1], v[2], v[3]};
    }
    int toff[4];                               // halfword offset of tap k = 4 kg + e from (row ty, column lx) of the halo; -1: k >= 9
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * kg + e;
        toff[e] = k < 9 ? (k / 3) * ST_PITCH + (k % 3) : -1;
    }
    const int c0 = 16 * kg;                    // this lane's channels: c0 + 4 n + j  <->  acc[n][j]
    // MODE 3, B operand of the dW MFMA: lane (tap n = lx, kg) holds x at pixels 4 kg .. 4 kg + 3 of the row, shifted by tap n
    const int woff = lx < 9 ? (lx / 3) * ST_PITCH + (lx % 3) + 4 * kg : -1;

    // ---- per-channel coefficients
    //   MODE 1: scale / shift in registers.  MODE 2: scale, shift, mean, rstd in registers (2 x 16 accumulators only).
    //   MODE 3: scale, shift, b, k in LDS (144 accumulators in registers).
    float r_a[16], r_s[16], r_b[16], r_k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r_a[i] = r_s[i] = r_b[i] = r_k[i] = 0.f; }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            r_a[i] = a.scale[c0 + i]; r_s[i] = a.shift[c0 + i];
            if (MODE == 2) { r_b[i] = a.mean[c0 + i]; r_k[i] = a.rstd[c0 + i]; }
        }
    }
    if constexpr (MODE == 3) {
        if (tid < ST_C) {
            const float sc = a.scale[tid];
            const float b_ = -sc * a.rstd[tid] * a.dgamma[tid] * a.inv_n;
            cf[0][tid] = sc;
            cf[1][tid] = a.shift[tid];
            cf[2][tid] = b_;
            cf[3][tid] = -sc * a.dbeta[tid] * a.inv_n - b_ * a.mean[tid];
        }
    }

    // ---- accumulators over all the tiles of this workgroup
    constexpr int NACC = MODE == 3 ? 1 : (MODE == 1 ? 1 : 2);      // MODE 3: acc[0][4 g + j] = dW[channel 16 g + 4 kg + j][tap lx]
    float acc[NACC][16];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pv[i] = 0.f;
    bool have_pivot = false;
    float cnt = 0.f;

    // the image halo of the NEXT tile travels in registers while the current tile is computed
    unsigned short xr[2];
    auto fetch_halo = [&](int tile_) {
        int t = tile_;
        const int txt = t % a.tilesX; t /= a.tilesX;
        const int tyt = t % a.tilesY;
        const int b = t / a.tilesY;
        const int y0 = tyt * ST_TILE, x0p = txt * ST_TILE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + k * 256;
            const int hy = q / ST_HALO, hx = q - hy * ST_HALO;
            const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
            const bool in = q < ST_HALO * ST_HALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1);
            const unsigned short v = __builtin_bit_cast(unsigned short, a.x[(int64_t)((b * H + gyc) * W + gxc) * a.ldx]);
            xr[k] = in ? v : (unsigned short)0;
        }
    };
    if ((int)blockIdx.x < ntile) fetch_halo(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int t = tile;
        const int txt = t % a.tilesX; t /= a.tilesX;
        const int tyt = t % a.tilesY;
        const int b = t / a.tilesY;
        const int y0 = tyt * ST_TILE, x0p = txt * ST_TILE;
        const int vy = min(ST_TILE, H - y0), vx = min(ST_TILE, W - x0p);
        // dz of this wave's four pixel rows (2 x 16 bytes per lane and row) is requested FIRST, branch-free (clamped to a pixel
        // of the tile, masked where used), so that it travels while the halo is staged and the two barriers pass
        u32x4 draw[4][2];
        if constexpr (MODE >= 2) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int gp = (b * H + y0 + min(4 * wave + rr, vy - 1)) * W + x0p + min(lx, vx - 1);
                const bf16_t* dp = a.dz + (int64_t)gp * a.lddz + c0;
                draw[rr][0] = *reinterpret_cast<const u32x4*>(dp);
                draw[rr][1] = *reinterpret_cast<const u32x4*>(dp + 8);
            }
        }
        __syncthreads();                                   // the previous tile's readers are done
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + k * 256;
            if (q < ST_HALO * ST_HALO) xs[(q / ST_HALO) * ST_PITCH + (q % ST_HALO)] = xr[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntile) fetch_halo(tile + gridDim.x);

        // wave w owns tile rows 4 w .. 4 w + 3: one MFMA group (16 pixels x 64 channels) per row
        auto do_row = [&](auto rr_c) {
            constexpr int rr = decltype(rr_c)::value;
            const int ty = 4 * wave + rr;
            const bool in = ty < vy && lx < vx;
            const int gpix = (b * H + y0 + min(ty, vy - 1)) * W + x0p + min(lx, vx - 1);      // clamped: loads stay in range
            const int base = ty * ST_PITCH + lx;
            short bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = toff[e] >= 0 ? (short)xs[base + toff[e]] : (short)0;
            const s16x4 bB = {bv[0], bv[1], bv[2], bv[3]};
            float y[16];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wA[n], bB, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[4 * n + j] = uh_round_as<bf16_t>(d[j]);              // y as it would have been stored
            }
            if constexpr (MODE == 0) {
                if (!have_pivot) {
                    // shift of the sums = the values at one pixel of the workgroup's first tile: (8, 8) when the tile has it (an
                    // interior pixel: the first tile of workgroup 0 is the image corner), else (0, 0)
                    const bool mid = vy > 8 && vx > 8;
                    const int prow = mid ? 8 : 0, pcol = mid ? 8 : 0;
                    if (ty == prow && lx == pcol) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) pivot[c0 + i] = y[i];
                    }
                }
                if (have_pivot && in) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float d = y[i] - pv[i]; acc[0][i] += d; acc[1][i] = fmaf(d, d, acc[1][i]); }
                }
                if (!have_pivot) {
                    // first tile: park the rounded outputs of this row, the sums are taken below once the pivot is known
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (rr == 0) r_a[i] = y[i];
                        if (rr == 1) r_s[i] = y[i];
                        if (rr == 2) r_b[i] = y[i];
                        if (rr == 3) r_k[i] = y[i];
                    }
                }
            } else if constexpr (MODE == 1) {
                bf16x8 o0, o1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    o0[i] = (bf16_t)uh_relu(fmaf(y[i], r_a[i], r_s[i]));
                    o1[i] = (bf16_t)uh_relu(fmaf(y[8 + i], r_a[8 + i], r_s[8 + i]));
                }
                if (in) {
                    bf16_t* zp = a.z + (int64_t)gpix * a.ldz + c0;
                    *reinterpret_cast<bf16x8*>(zp) = o0;
                    *reinterpret_cast<bf16x8*>(zp + 8) = o1;
                }
            } else {
                float d[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d[2 * q] = in ? __builtin_bit_cast(float, draw[rr][0][q] << 16) : 0.f;
                    d[2 * q + 1] = in ? __builtin_bit_cast(float, draw[rr][0][q] & 0xffff0000u) : 0.f;
                    d[8 + 2 * q] = in ? __builtin_bit_cast(float, draw[rr][1][q] << 16) : 0.f;
                    d[8 + 2 * q + 1] = in ? __builtin_bit_cast(float, draw[rr][1][q] & 0xffff0000u) : 0.f;
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float m = (fmaf(y[i], r_a[i], r_s[i]) > 0.f) ? d[i] : 0.f;          // d is 0 outside the image
                        acc[0][i] += m;
                        acc[1][i] += m * (y[i] - r_b[i]) * r_k[i];
                    }
                } else {
                    // dy of this pixel, 16 channels, as uh_bn_relu_bwd_apply would have stored it
                    float dyv[16];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f32x4 ca = *reinterpret_cast<const f32x4*>(&cf[0][c0 + 4 * h]), cs = *reinterpret_cast<const f32x4*>(&cf[1][c0 + 4 * h]);
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(&cf[2][c0 + 4 * h]), ck = *reinterpret_cast<const f32x4*>(&cf[3][c0 + 4 * h]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * h + e;
                            const float m = (fmaf(y[i], ca[e], cs[e]) > 0.f) ? d[i] : 0.f;
                            const float o = uh_round_as<bf16_t>(fmaf(ca[e], m, fmaf(cb[e], y[i], ck[e])));
                            dyv[i] = in ? o : 0.f;
                        }
                    }
                    // dy of the row -> this wave's LDS tile [pixel][channel] ...
                    unsigned short* tile_w = dyt + wave * 16 * DP;
                    {
                        bf16x4 p4[4];
#pragma unroll
                        for (int h = 0; h < 4; ++h) p4[h] = bf16x4{(bf16_t)dyv[4 * h], (bf16_t)dyv[4 * h + 1], (bf16_t)dyv[4 * h + 2], (bf16_t)dyv[4 * h + 3]};
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            *reinterpret_cast<u32x2*>(tile_w + lx * DP + c0 + 4 * h) = __builtin_bit_cast(u32x2, p4[h]);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same wave: LDS executes its accesses in order
                    // ... and back as the A operand (pixels along K) of four MFMAs, one per 16-channel group
                    short xb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xb[e] = woff >= 0 ? (short)xs[ty * ST_PITCH + woff + e] : (short)0;
                    const s16x4 bW = {xb[0], xb[1], xb[2], xb[3]};
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        short av[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[e] = (short)tile_w[(4 * kg + e) * DP + 16 * gq + lx];
                        const s16x4 aW = {av[0], av[1], av[2], av[3]};
                        f32x4 c4 = {acc[0][4 * gq], acc[0][4 * gq + 1], acc[0][4 * gq + 2], acc[0][4 * gq + 3]};
                        c4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aW, bW, c4, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[0][4 * gq + j] = c4[j];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile is rewritten by the next row
                    __builtin_amdgcn_sched_barrier(0);       // one pixel row at a time (register pressure)
                }
            }
        };
        // (MODE 2 / 3: one row at a time -- interleaving the four rows' unpacked values costs a wave per SIMD)
        do_row(std::integral_constant<int, 0>{});
        if constexpr (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
        do_row(std::integral_constant<int, 1>{});
        if constexpr (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
        do_row(std::integral_constant<int, 2>{});
        if constexpr (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
        do_row(std::integral_constant<int, 3>{});
        if constexpr (MODE == 0) {
            if (!have_pivot) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[i] = pivot[c0 + i];
                have_pivot = true;
                // the first tile's four rows, parked in r_a / r_s / r_b / r_k
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ty = 4 * wave + rr;
                    if (ty < vy && lx < vx) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float v = rr == 0 ? r_a[i] : (rr == 1 ? r_s[i] : (rr == 2 ? r_b[i] : r_k[i]));
                            const float d = v - pv[i];
                            acc[0][i] += d;
                            acc[1][i] = fmaf(d, d, acc[1][i]);
                        }
                    }
                }
            }
            cnt += (float)(vy * vx);
        }
    }

    if constexpr (MODE == 1) return;
    if constexpr (MODE == 3) {
        // acc[0][4 g + j] = dW[channel 16 g + 4 kg + j][tap lx] of this wave's rows: add the four waves through LDS
        __syncthreads();
        if (lx < 9) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wave][lx][16 * gq + 4 * kg + j] = acc[0][4 * gq + j];
        }
        __syncthreads();
        float* slab = a.out + (int64_t)blockIdx.x * ST_C * 9;
        for (int idx = tid; idx < 9 * ST_C; idx += 256) {
            const int k = idx / ST_C, c = idx - k * ST_C;
            slab[c * 9 + k] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);     // [channel][tap] = KRSC with Cin = 1
        }
        return;
    }
    // ---- reduce over the 16 pixel lanes of a lane group (one DPP row), then over the 4 waves through LDS
    constexpr int NK = 2;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = uh_row16_sum(acc[k][i]);
    __syncthreads();
    if (lx == 0) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wave][k][c0 + i] = acc[k][i];
    }
    __syncthreads();
    if constexpr (MODE == 0) {
        if (tid < ST_C) {
            const float s1 = (red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid]);
            const float s2 = (red[0][1][tid] + red[1][1][tid]) + (red[2][1][tid] + red[3][1][tid]);
            const float n = cnt > 0.f ? cnt : 1.f;
            const float kk = have_pivot ? pivot[tid] : 0.f;
            a.out[((int64_t)blockIdx.x * 2 + 0) * ST_C + tid] = kk + s1 / n;                        // mean
            a.out[((int64_t)blockIdx.x * 2 + 1) * ST_C + tid] = fmaxf(s2 - s1 * s1 / n, 0.f);       // M2 (shifted-data formula)
        }
        // the statistics buffer is sized for one row per TILE: rows beyond the workgroup count get a zero pixel count
        float* counts = a.out + (int64_t)ntile * 2 * ST_C;
        if (tid == 0) counts[blockIdx.x] = cnt;
        for (int r = gridDim.x + blockIdx.x * 256 + tid; r < ntile; r += gridDim.x * 256) counts[r] = 0.f;
    } else if constexpr (MODE == 2) {
        for (int idx = tid; idx < 2 * ST_C; idx += 256) {
            const int k = idx / ST_C, c = idx - k * ST_C;
            a.out[((int64_t)blockIdx.x * 2 + k) * ST_C + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
        }
    }
}

}  // namespace

int uh_stem_mfma_launch(int mode, const void* x, int ldx, const void* w, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, const float* dgamma, const float* dbeta, float inv_n,
                                   const void* dz, int lddz, void* z, int ldz, float* out, int B, int H, int W, int grid,
                                   uh_stream stream) {
    StemArgs a;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.w = (const bf16_t*)w;
    a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.dgamma = dgamma; a.dbeta = dbeta; a.inv_n = inv_n;
    a.dz = (const bf16_t*)dz; a.lddz = lddz; a.z = (bf16_t*)z; a.ldz = ldz; a.out = out;
    a.B = B; a.H = H; a.W = W; a.tilesX = (W + ST_TILE - 1) / ST_TILE; a.tilesY = (H + ST_TILE - 1) / ST_TILE;
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL(stem_mfma_kernel<0>, dim3(grid), dim3(256), 0, st, a); break;
        case 1: hipLaunchKernelGGL(stem_mfma_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(stem_mfma_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(stem_mfma_kernel<3>, dim3(grid), dim3(256), 0, st, a); break;
        default: uh_set_error("uh_stem_mfma_launch: bad mode %d", mode); return UH_EINVAL;
    }
    UH_CHECK_LAUNCH("stem_mfma_kernel");
    return UH_OK;
}
