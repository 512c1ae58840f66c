// convt_mfma.hip -- nn.ConvTranspose2d(Cin, Cin/2, kernel_size=2, stride=2) + F.pad (unet_parts.py:73,85-88) on the
// matrix cores.  k2 s2 never overlaps: output pixel (2h+r, 2w+s) depends on input pixel (h,w) only, through the
// [Cin x Cout] slice W[:, :, r, s], so all three directions are GEMMs over pixels.  One kernel, three operand plans:
//
//   fwd    y[(m,q)][co] = bias[co] + sum_ci  Wf[(q,co)][ci] * x[m][ci]            rows = 4*Cout, K = Cin
//   dgrad  dx[m][ci]    =            sum_qco Wd[ci][(q,co)] * dy[(m,q)][co]       rows = Cin,    K = 4*Cout
//   wgrad  dW[(q,co)][ci] =          sum_m   dyT[(q,co)][m] * xT[ci][m]           rows = 4*Cout, K = pixels (split)
//
// (m,q) = output pixel (2h + q/2 + pad_top, 2w + q%2 + pad_left) of input pixel m.  The kernel computes
// C[row][col] = sum_k A[row][k] * B[col][k] with BOTH operands K-contiguous ("NT"): A = packed filters (or dyT),
// B = pixels (or xT).  Workgroup = 128 rows x 128 columns, 4 waves of 64x64 (16 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 / 4 x v_mfma_f32_16x16x4_f32), 64-byte K chunks staged HBM -> LDS by LDS-DMA with the same
// XOR swizzle as the 3x3 conv (conflict-free ds_read_b128 fragments), double buffered.  The MFMA puts 4 consecutive
// ROWS (= channels) of one column (= pixel) in a lane, so the fwd / dgrad epilogues store 8 / 16 contiguous bytes.
// wgrad gets its K-contiguous operands from two tiled transposes (pixels are the contraction index but channels are
// the contiguous one in NHWC) and is split over pixel ranges into fp32 slabs, reduced deterministically.
#include "uh_vec.h"

namespace {

constexpr unsigned CT_OOB = 0xF0000000u;      // beyond every buffer range -> the DMA returns zeros
enum { CT_FWD = 0, CT_DGRAD = 1, CT_WGRAD = 2 };

struct CtmGeom { int B, h, w, Ho, Wo, pt, pl, Cin, Cout; };

template <typename T>
struct CtmArgs {
    const T* A; int lda; int arows;            // row-major [arows][lda >= K], K contiguous
    const T* Bm; int ldb; long long bcols;     // FWD: x (ld = pixel stride), WGRAD: xT [Cin][pixels]; DGRAD: dy
    unsigned a_bytes, b_bytes;
    int K;                                     // total contraction length
    int kchunks_per_split;                     // WGRAD: chunks per blockIdx.z (others: all)
    CtmGeom g;
    const float* bias;                         // FWD
    T* out; int ldo;                           // FWD: y, DGRAD: dx
    float* slabs;                              // WGRAD: [split][arows][bcols]
};

__device__ __forceinline__ int ctm_swz(int row) { return ((row >> 2) & 1) << 1; }

// output pixel of (input pixel m, quadrant q) or -1 when F.pad crops it away
__device__ __forceinline__ int ctm_out_pixel(const CtmGeom& g, int m, int q) {
    const int hx = m % g.w, t = m / g.w, hy = t % g.h, b = t / g.h;
    const int oy = 2 * hy + (q >> 1) + g.pt, ox = 2 * hx + (q & 1) + g.pl;
    if (oy < 0 || oy >= g.Ho || ox < 0 || ox >= g.Wo) return -1;
    return (b * g.Ho + oy) * g.Wo + ox;
}

// SPLIT (T = float, "bf16x3", dtype code UH_F32X3): both fp32 fragments are split into bf16 hi/lo halves as they leave
// LDS and the three products ah*bh + ah*bl + al*bh run on v_mfma_f32_16x16x16_bf16 (fp32 accumulate, ~1e-5 relative).
template <typename T, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void ctm_gemm_kernel(CtmArgs<T> a) {
    constexpr int ES = sizeof(T);
    constexpr int CK = 64 / ES;                // K elements per 64-byte chunk
    constexpr int VEC = 16 / ES;
    constexpr int TILE_BYTES = 128 * 64;       // one operand tile: 128 rows x 64 bytes
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TILE_BYTES];   // [buffer][A | B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & 15, kg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;                 // wave = 64 rows x 64 columns
    const int row0 = blockIdx.y * 128;
    const long long col0 = (long long)blockIdx.x * 128;

    const int nchunk_total = a.K / CK;
    int kc_begin = 0, kc_end = nchunk_total;
    if (MODE == CT_WGRAD) {
        kc_begin = blockIdx.z * a.kchunks_per_split;
        kc_end = min(kc_begin + a.kchunks_per_split, nchunk_total);
    }

    // ---- DMA plan: LDS slot p = tid + k*256 (k = 0,1: A tile, k = 2,3: B tile) holds (row = (p & 511) >> 2,
    // part' = p & 3) = source part part' ^ swz(row)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)a.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.Bm, 0, (int)a.b_bytes, 0x00020000);
    unsigned offA[2], offB[2];                 // byte offset of (row, source part) at K = 0, or CT_OOB
    int bbase[2], bvalid[2];                   // DGRAD: output pixel of quadrant 0 for the B rows of this thread + 4 valid bits
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = tid + k * 256, row = p >> 2, part = (p & 3) ^ ctm_swz(row);
        const int ar = row0 + row;
        offA[k] = ar < a.arows ? (unsigned)(((long long)ar * a.lda) * ES + part * 16) : CT_OOB;
        const long long bc = col0 + row;
        bbase[k] = 0; bvalid[k] = 0;
        if (MODE == CT_DGRAD) {
            offB[k] = (unsigned)(part * 16);
            if (bc < a.bcols) {
                const int m = (int)bc, hx = m % a.g.w, t = m / a.g.w, hy = t % a.g.h, b = t / a.g.h;
                const int oy = 2 * hy + a.g.pt, ox = 2 * hx + a.g.pl;
                bbase[k] = (b * a.g.Ho + oy) * a.g.Wo + ox;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int yy = oy + (q >> 1), xx = ox + (q & 1);
                    if (yy >= 0 && yy < a.g.Ho && xx >= 0 && xx < a.g.Wo) bvalid[k] |= 1 << q;
                }
            }
        } else {
            offB[k] = bc < a.bcols ? (unsigned)((bc * a.ldb) * ES + part * 16) : CT_OOB;
        }
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma_chunk = [&](int kc, int bufi) {
        unsigned char* dstA = lds + bufi * (2 * TILE_BYTES) + wave * 1024;
        unsigned char* dstB = dstA + TILE_BYTES;
        const unsigned kbyte = (unsigned)(kc * 64);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned vo = offA[k] == CT_OOB ? CT_OOB : offA[k] + kbyte;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dstA + k * 4096), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned vo;
            if (MODE == CT_DGRAD) {
                // K index = q*Cout + co: a 64-byte chunk lies inside one quadrant (Cout % CK == 0)
                const int k0 = kc * CK, q = k0 / a.g.Cout, co0 = k0 - q * a.g.Cout;
                const int op = bbase[k] + (q >> 1) * a.g.Wo + (q & 1);
                vo = ((bvalid[k] >> q) & 1) ? (unsigned)(((long long)op * a.ldb + co0) * ES) + offB[k] : CT_OOB;
            } else {
                vo = offB[k] == CT_OOB ? CT_OOB : offB[k] + kbyte;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(dstB + k * 4096), 16, vo, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kc_begin < kc_end) {
        dma_chunk(kc_begin, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this wave's pieces have landed BEFORE it signals the barrier
        __syncthreads();                       // chunk 0 is in LDS
    }
    int bufi = 0;
    for (int kc = kc_begin; kc < kc_end; ++kc, bufi ^= 1) {
        const unsigned char* bufA = lds + bufi * (2 * TILE_BYTES);
        const unsigned char* bufB = bufA + TILE_BYTES;
        u32x4 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wr * 64 + i * 16 + lx;
            fa[i] = *reinterpret_cast<const u32x4*>(bufA + r * 64 + ((kg ^ ctm_swz(r)) << 4));
            const int c = wc * 64 + i * 16 + lx;
            fb[i] = *reinterpret_cast<const u32x4*>(bufB + c * 64 + ((kg ^ ctm_swz(c)) << 4));
        }
        // the next chunk's DMA goes out AFTER this chunk's fragment reads (a ds_read that follows a DMA in program
        // order makes the compiler drain the DMA first) and lands under the 16 MFMAs
        if (kc + 1 < kc_end) dma_chunk(kc + 1, bufi ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPLIT) {                 // 4 floats -> {hi01, hi23, lo01, lo23}
            auto split4 = [](u32x4& f) {
                const f32x4 x = __builtin_bit_cast(f32x4, f);
                unsigned hi[2], lo[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bf16_t h0 = (bf16_t)x[2 * q], h1 = (bf16_t)x[2 * q + 1];
                    const bf16_t l0 = (bf16_t)(x[2 * q] - (float)h0), l1 = (bf16_t)(x[2 * q + 1] - (float)h1);
                    hi[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    lo[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
                f = u32x4{hi[0], hi[1], lo[0], lo[1]};
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) { split4(fa[i]); split4(fb[i]); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SPLIT) {
                    const s16x4 ah = __builtin_bit_cast(s16x4, u32x2{fa[i][0], fa[i][1]}), al = __builtin_bit_cast(s16x4, u32x2{fa[i][2], fa[i][3]});
                    const s16x4 bh = __builtin_bit_cast(s16x4, u32x2{fb[j][0], fb[j][1]}), bl = __builtin_bit_cast(s16x4, u32x2{fb[j][2], fb[j][3]});
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc[i][j], 0, 0, 0);
                } else if constexpr (ES == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                        __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
                } else {
                    const f32x4 x = __builtin_bit_cast(f32x4, fa[i]), y = __builtin_bit_cast(f32x4, fb[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[q], acc[i][j], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);     // keep the wait + barrier BELOW the MFMAs (they only touch registers)
        __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0) BEFORE the barrier: every wave reads pieces fetched by the others
        __syncthreads();                       // next chunk landed and this buffer is free again
    }

    // ---- epilogue: acc[i][j][e] = C[row0 + wr*64 + i*16 + kg*4 + e][col0 + wc*64 + j*16 + lx]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long col = col0 + wc * 64 + j * 16 + lx;
        if (col >= a.bcols) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + wr * 64 + i * 16 + kg * 4;
            if (row >= a.arows) continue;              // arows % 4 == 0 (host check)
            if (MODE == CT_FWD) {
                const int q = row / a.g.Cout, co = row - q * a.g.Cout;
                const int op = ctm_out_pixel(a.g, (int)col, q);
                if (op < 0) continue;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e] + bv[e];
                T* dst = a.out + (long long)op * a.ldo + co;
                if constexpr (ES == 2) {
                    bf16x4 v = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(dst) = v;
                } else {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                }
            } else if (MODE == CT_DGRAD) {
                T* dst = a.out + col * a.ldo + row;
                if constexpr (ES == 2) {
                    bf16x4 v = {(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
                    *reinterpret_cast<bf16x4*>(dst) = v;
                } else {
                    *reinterpret_cast<f32x4*>(dst) = acc[i][j];
                }
            } else {
                float* dst = a.slabs + ((long long)blockIdx.z * a.arows + row) * a.bcols + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(long long)e * a.bcols] = acc[i][j][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------ packing / transposes
// reference weight [Cin][Cout][2][2] fp32 -> Wf[(q,co)][ci] and Wd[ci][(q,co)] in T
template <typename T>
__global__ void ctm_pack_kernel(const float* __restrict__ w, int Cin, int Cout, T* __restrict__ wf, T* __restrict__ wd) {
    const long long total = (long long)Cin * Cout * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx & 3);
        const long long t = idx >> 2;
        const int co = (int)(t % Cout), ci = (int)(t / Cout);
        const T v = uh_from_f32<T>(w[idx]);
        wf[((long long)q * Cout + co) * Cin + ci] = v;
        wd[(long long)ci * (4 * Cout) + q * Cout + co] = v;
    }
}

// out[c][m] = in[row(m)][c] for c < C: 32x32 tiles through LDS.  QUAD: row(m) = output pixel (m, q) of a strided
// ConvTranspose gradient (zeros when cropped), out row = q*C + c; otherwise row(m) = m.
template <typename T, bool QUAD>
__global__ __launch_bounds__(256) void ctm_transpose_kernel(const T* __restrict__ in, int ld, int C, long long M, CtmGeom g,
                                                            T* __restrict__ out) {
    __shared__ float tile[32][33];
    const int q = QUAD ? blockIdx.z : 0;
    const long long m0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const long long m = m0 + r;
        float v = 0.f;
        if (m < M && c0 + tx < C) {
            long long src = m;
            if (QUAD) src = ctm_out_pixel(g, (int)m, q);
            if (src >= 0) v = uh_to_f32(in[src * ld + c0 + tx]);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long m = m0 + tx;
        if (c < C && m < M) out[((long long)q * C + c) * M + m] = uh_from_f32<T>(tile[tx][r]);
    }
}

// dw[ci][co][q] = sum over splits of slabs[split][(q,co)][ci]   (reference layout [Cin][Cout][2][2])
__global__ void ctm_wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, int Cin, int Cout, float* __restrict__ dw) {
    const long long total = (long long)Cin * Cout * 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ci = (int)(idx % Cin);
    const long long t = idx / Cin;                 // (q, co) row
    const int co = (int)(t % Cout), q = (int)(t / Cout);
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slabs[(long long)s * total + idx];
    dw[((long long)ci * Cout + co) * 4 + q] = v;
}

// dbias[co] = sum of dy over the up-sampled region: partial rows (interleaved 64-pixel chunks) + double finish
template <typename T>
__global__ __launch_bounds__(256) void ctm_dbias_partial_kernel(const T* __restrict__ dy, int lddy, CtmGeom g,
                                                                float* __restrict__ partials) {
    // thread = (pixel lane of 256/G, 16-byte channel vector); G = Cout / VEC <= 64 vectors
    constexpr int V = 16 / (int)sizeof(T);
    const int G = g.Cout / V, PLN = 256 / G;
    const int gi = threadIdx.x % G, pl = threadIdx.x / G;
    const int y_lo = max(g.pt, 0), y_hi = min(g.pt + 2 * g.h, g.Ho), x_lo = max(g.pl, 0), x_hi = min(g.pl + 2 * g.w, g.Wo);
    const int rw = x_hi - x_lo, rh = y_hi - y_lo;
    const long long n = (rw > 0 && rh > 0) ? (long long)g.B * rh * rw : 0;
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    if (pl < PLN)
        for (long long p = (long long)blockIdx.x * PLN + pl; p < n; p += (long long)gridDim.x * PLN) {
            const int xx = (int)(p % rw) + x_lo, yy = (int)((p / rw) % rh) + y_lo, b = (int)(p / ((long long)rw * rh));
            float v[V];
            uh_load<T, V>(dy + (((long long)b * g.Ho + yy) * g.Wo + xx) * lddy + gi * V, v);
#pragma unroll
            for (int i = 0; i < V; ++i) s[i] += v[i];
        }
    __shared__ float red[256][V + 1];
#pragma unroll
    for (int i = 0; i < V; ++i) red[threadIdx.x][i] = s[i];
    __syncthreads();
    for (int c = threadIdx.x; c < g.Cout; c += 256) {
        const int gg = c / V, i = c - gg * V;
        float t = 0.f;
        for (int k = 0; k < PLN; ++k) t += red[k * G + gg][i];
        partials[(long long)blockIdx.x * g.Cout + c] = t;
    }
}
// block = 16 channels x 64 row lanes over the partial rows, double accumulation
__global__ __launch_bounds__(1024) void ctm_dbias_finish_kernel(const float* __restrict__ partials, int nblk, int Cout,
                                                                float* __restrict__ dbias) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 16 + cl;
    double t = 0.0;
    if (c < Cout)
        for (int k = sl; k < nblk; k += 64) t += (double)partials[(long long)k * Cout + c];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if ((threadIdx.x & 63) < 16) red[wave][cl] = t;
    __syncthreads();
    if (threadIdx.x < 16 && c < Cout) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][cl];
        dbias[c] = (float)v;
    }
}

template <typename T>
static bool ctm_ok(int B, int h, int w, int Cin, int Cout, const void* p0, int ld0, const void* p1, int ld1) {
    constexpr int ES = sizeof(T), CK = 64 / ES;
    const long long M = (long long)B * h * w;
    return Cin % CK == 0 && Cout % CK == 0 && Cout % 4 == 0 && M < (1ll << 30) && uh_aligned16(p0) && uh_aligned16(p1) &&
           (ld0 * ES) % 16 == 0 && (ld1 * ES) % 16 == 0;
}

static inline CtmGeom ctm_geom(int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pt, int pl) {
    CtmGeom g; g.B = B; g.h = h; g.w = w_; g.Ho = Ho; g.Wo = Wo; g.pt = pt; g.pl = pl; g.Cin = Cin; g.Cout = Cout;
    return g;
}

constexpr long long CT_MAX_BYTES = (1ll << 31) - 4096;

}  // namespace

// ------------------------------------------------------------------------------------ C ABI (MFMA path)
extern "C" int uh_convt2x2_pack(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dt, uh_stream stream) {
    UH_REQUIRE(w && w_fwd && w_dgrad && Cin > 0 && Cout > 0, "uh_convt2x2_pack: bad args");
    if (dt == UH_F32X3) dt = UH_F32;                 // bf16x3 splits both operands in registers: plain fp32 copies
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16, "uh_convt2x2_pack: bad dtype %d", dt);
    long long total = (long long)Cin * Cout * 4, g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    UH_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL(ctm_pack_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (T*)w_fwd,
                           (T*)w_dgrad);
    });
    UH_CHECK_LAUNCH("ctm_pack_kernel");
    return UH_OK;
}

// 1 when the MFMA kernels take this problem (else the callers use the SIMT path of convt_1x1.hip)
extern "C" int uh_convt2x2_mfma_ok(int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int dt) {
    if (dt == UH_F32X3) dt = UH_F32;                 // same tensors, same shape rules
    const int ES = dt == UH_BF16 ? 2 : 4, CK = 64 / ES;
    const long long M = (long long)B * h * w_;
    if (Cin % CK || Cout % CK || M >= (1ll << 30) || M % CK) return 0;
    if ((long long)B * Ho * Wo * Cout * ES >= CT_MAX_BYTES || M * Cin * ES >= CT_MAX_BYTES) return 0;
    if (4ll * Cout * M * ES >= CT_MAX_BYTES) return 0;              // dyT workspace of wgrad
    return 1;
}

extern "C" int uh_convt2x2_fwd_mfma(const void* x, int ldx, const void* w_fwd, const float* bias, void* y, int ldy, int B,
                                    int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt,
                                    uh_stream stream) {
    UH_REQUIRE(x && w_fwd && bias && y && ldx >= Cin && ldy >= Cout, "uh_convt2x2_fwd_mfma: bad args");
    UH_REQUIRE(uh_convt2x2_mfma_ok(B, h, w_, Cin, Cout, Ho, Wo, dt), "uh_convt2x2_fwd_mfma: shape not supported");
    UH_REQUIRE(uh_aligned16(x) && uh_aligned16(y) && uh_aligned16(w_fwd) && uh_aligned16(bias), "uh_convt2x2_fwd_mfma: alignment");
    hipStream_t st = (hipStream_t)stream;
    const bool split = dt == UH_F32X3;
    if (split) dt = UH_F32;
    const long long M = (long long)B * h * w_;
    const bool border = !(pad_top == 0 && pad_left == 0 && Ho == 2 * h && Wo == 2 * w_);
    UH_DISPATCH_DT(dt, T, {
        UH_REQUIRE((ldx * (int)sizeof(T)) % 16 == 0 && (ldy * (int)sizeof(T)) % 8 == 0, "uh_convt2x2_fwd_mfma: strides");
        if (border) {
            UH_REQUIRE(ldy == Cout, "uh_convt2x2_fwd_mfma: a padded output must be pixel-dense");
            (void)hipMemsetAsync(y, 0, (size_t)B * Ho * Wo * Cout * sizeof(T), st);
        }
        CtmArgs<T> a{};
        a.A = (const T*)w_fwd; a.lda = Cin; a.arows = 4 * Cout;
        a.Bm = (const T*)x; a.ldb = ldx; a.bcols = M;
        a.a_bytes = (unsigned)((size_t)4 * Cout * Cin * sizeof(T));
        a.b_bytes = (unsigned)((size_t)M * ldx * sizeof(T));
        a.K = Cin; a.kchunks_per_split = 0;
        a.g = ctm_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
        a.bias = bias; a.out = (T*)y; a.ldo = ldy; a.slabs = nullptr;
        if (split) {
            if constexpr (sizeof(T) == 4)
                hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_FWD, true>), dim3((unsigned)((M + 127) / 128), (4 * Cout + 127) / 128, 1),
                                   dim3(256), 0, st, a);
        } else
            hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_FWD>), dim3((unsigned)((M + 127) / 128), (4 * Cout + 127) / 128, 1), dim3(256),
                               0, st, a);
    });
    UH_CHECK_LAUNCH("ctm_gemm_kernel<fwd>");
    return UH_OK;
}

extern "C" int uh_convt2x2_dgrad_mfma(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, int B, int h, int w_,
                                      int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left, int dt,
                                      uh_stream stream) {
    UH_REQUIRE(dy && w_dgrad && dx && lddy >= Cout && lddx >= Cin, "uh_convt2x2_dgrad_mfma: bad args");
    UH_REQUIRE(uh_convt2x2_mfma_ok(B, h, w_, Cin, Cout, Ho, Wo, dt), "uh_convt2x2_dgrad_mfma: shape not supported");
    UH_REQUIRE(uh_aligned16(dy) && uh_aligned16(dx) && uh_aligned16(w_dgrad), "uh_convt2x2_dgrad_mfma: alignment");
    hipStream_t st = (hipStream_t)stream;
    const bool split = dt == UH_F32X3;
    if (split) dt = UH_F32;
    const long long M = (long long)B * h * w_;
    UH_DISPATCH_DT(dt, T, {
        UH_REQUIRE((lddy * (int)sizeof(T)) % 16 == 0 && (lddx * (int)sizeof(T)) % 8 == 0, "uh_convt2x2_dgrad_mfma: strides");
        UH_REQUIRE((long long)B * Ho * Wo * lddy * (long long)sizeof(T) < CT_MAX_BYTES, "uh_convt2x2_dgrad_mfma: dy too large");
        CtmArgs<T> a{};
        a.A = (const T*)w_dgrad; a.lda = 4 * Cout; a.arows = Cin;
        a.Bm = (const T*)dy; a.ldb = lddy; a.bcols = M;
        a.a_bytes = (unsigned)((size_t)4 * Cout * Cin * sizeof(T));
        a.b_bytes = (unsigned)((size_t)B * Ho * Wo * lddy * sizeof(T));
        a.K = 4 * Cout; a.kchunks_per_split = 0;
        a.g = ctm_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
        a.bias = nullptr; a.out = (T*)dx; a.ldo = lddx; a.slabs = nullptr;
        if (split) {
            if constexpr (sizeof(T) == 4)
                hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_DGRAD, true>), dim3((unsigned)((M + 127) / 128), (Cin + 127) / 128, 1),
                                   dim3(256), 0, st, a);
        } else
            hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_DGRAD>), dim3((unsigned)((M + 127) / 128), (Cin + 127) / 128, 1), dim3(256), 0,
                               st, a);
    });
    UH_CHECK_LAUNCH("ctm_gemm_kernel<dgrad>");
    return UH_OK;
}

static int ctm_wgrad_plan(int B, int h, int w_, int Cin, int Cout, int dt, int* chunks_per_split) {
    const int ES = dt == UH_BF16 ? 2 : 4, CK = 64 / ES;       // UH_F32X3 plans like UH_F32
    const long long M = (long long)B * h * w_;
    const int nchunk = (int)(M / CK);
    const int tiles = ((4 * Cout + 127) / 128) * ((Cin + 127) / 128);
    int want = (1024 + tiles - 1) / tiles;               // ~4 workgroups per CU over the whole launch
    if (want > nchunk) want = nchunk;
    if (want > 256) want = 256;
    if (want < 1) want = 1;
    const int cps = (nchunk + want - 1) / want;
    *chunks_per_split = cps;
    return (nchunk + cps - 1) / cps;
}

// workspace: dyT [4*Cout][M] + xT [Cin][M] in the activation dtype, fp32 slabs [nsplit][4*Cout][Cin], dbias partials
extern "C" size_t uh_convt2x2_wgrad_mfma_ws_bytes(int B, int h, int w_, int Cin, int Cout, int dt) {
    const size_t ES = dt == UH_BF16 ? 2 : 4;
    const size_t M = (size_t)B * h * w_;
    int cps;
    const int nsplit = ctm_wgrad_plan(B, h, w_, Cin, Cout, dt, &cps);
    size_t n = (4 * (size_t)Cout + Cin) * M * ES;
    n = (n + 255) & ~(size_t)255;
    n += (size_t)nsplit * 4 * Cout * Cin * sizeof(float);
    n += (size_t)1024 * Cout * sizeof(float);
    return n + 256;
}

extern "C" int uh_convt2x2_wgrad_mfma(const void* dy, int lddy, const void* x, int ldx, float* dw, float* dbias, void* ws,
                                      size_t ws_bytes, int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top,
                                      int pad_left, int dt, uh_stream stream) {
    UH_REQUIRE(dy && x && dw && dbias && ws && lddy >= Cout && ldx >= Cin, "uh_convt2x2_wgrad_mfma: bad args");
    UH_REQUIRE(uh_convt2x2_mfma_ok(B, h, w_, Cin, Cout, Ho, Wo, dt), "uh_convt2x2_wgrad_mfma: shape not supported");
    const size_t need = uh_convt2x2_wgrad_mfma_ws_bytes(B, h, w_, Cin, Cout, dt);
    if (ws_bytes < need) {
        uh_set_error("uh_convt2x2_wgrad_mfma: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    UH_REQUIRE(uh_aligned16(ws), "uh_convt2x2_wgrad_mfma: workspace alignment");
    hipStream_t st = (hipStream_t)stream;
    const bool split = dt == UH_F32X3;
    if (split) dt = UH_F32;
    const long long M = (long long)B * h * w_;
    int cps;
    const int nsplit = ctm_wgrad_plan(B, h, w_, Cin, Cout, dt, &cps);
    const CtmGeom g = ctm_geom(B, h, w_, Cin, Cout, Ho, Wo, pad_top, pad_left);
    UH_DISPATCH_DT(dt, T, {
        constexpr int V = 16 / (int)sizeof(T);
        T* dyT = (T*)ws;
        T* xT = dyT + (size_t)4 * Cout * M;
        size_t off = ((4 * (size_t)Cout + Cin) * (size_t)M * sizeof(T) + 255) & ~(size_t)255;
        float* slabs = (float*)((unsigned char*)ws + off);
        float* partials = slabs + (size_t)nsplit * 4 * Cout * Cin;
        hipLaunchKernelGGL((ctm_transpose_kernel<T, true>), dim3((unsigned)((M + 31) / 32), (Cout + 31) / 32, 4), dim3(256), 0, st,
                           (const T*)dy, lddy, Cout, M, g, dyT);
        hipLaunchKernelGGL((ctm_transpose_kernel<T, false>), dim3((unsigned)((M + 31) / 32), (Cin + 31) / 32, 1), dim3(256), 0, st,
                           (const T*)x, ldx, Cin, M, g, xT);
        CtmArgs<T> a{};
        a.A = dyT; a.lda = (int)M; a.arows = 4 * Cout;
        a.Bm = xT; a.ldb = (int)M; a.bcols = Cin;
        a.a_bytes = (unsigned)((size_t)4 * Cout * M * sizeof(T));
        a.b_bytes = (unsigned)((size_t)Cin * M * sizeof(T));
        a.K = (int)M; a.kchunks_per_split = cps;
        a.g = g; a.bias = nullptr; a.out = nullptr; a.ldo = 0; a.slabs = slabs;
        if (split) {
            if constexpr (sizeof(T) == 4)
                hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_WGRAD, true>), dim3((Cin + 127) / 128, (4 * Cout + 127) / 128, nsplit),
                                   dim3(256), 0, st, a);
        } else
            hipLaunchKernelGGL((ctm_gemm_kernel<T, CT_WGRAD>), dim3((Cin + 127) / 128, (4 * Cout + 127) / 128, nsplit), dim3(256), 0, st,
                               a);
        const long long nw = (long long)Cin * Cout * 4;
        hipLaunchKernelGGL(ctm_wgrad_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)slabs, nsplit,
                           Cin, Cout, dw);
        // dbias
        if (Cout % V == 0 && Cout / V <= 256 && uh_aligned16(dy) && (lddy * (int)sizeof(T)) % 16 == 0) {
            int nblk = (int)((4 * M + 255) / 256);
            if (nblk > 1024) nblk = 1024;
            hipLaunchKernelGGL(ctm_dbias_partial_kernel<T>, dim3(nblk), dim3(256), 0, st, (const T*)dy, lddy, g, partials);
            hipLaunchKernelGGL(ctm_dbias_finish_kernel, dim3((Cout + 15) / 16), dim3(1024), 0, st, (const float*)partials, nblk, Cout,
                               dbias);
        } else {
            uh_set_error("uh_convt2x2_wgrad_mfma: dbias needs 16-byte channel vectors");
            return UH_EINVAL;
        }
    });
    UH_CHECK_LAUNCH("uh_convt2x2_wgrad_mfma");
    return UH_OK;
}
