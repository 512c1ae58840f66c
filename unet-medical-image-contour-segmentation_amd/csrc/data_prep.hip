// data_prep.hip -- the device stage of the input pipeline: what BasicDataset.__getitem__ does to a
// decoded image / mask pair AFTER PIL has produced uint8 pixels (/root/reference/utils/data_loading.py):
//   :100-121  the x4 augmentation: item i = file i // 4 rotated by (i % 4) quarter turns counter-clockwise, canvas following
//             the image (Image.rotate(angle, expand=True) for right angles = a lossless transpose);
//   :74-78    mask grey levels -> class indices: 255 -> 2, 128 -> 1, everything else -> 0;
//   :86-87    `if (img > 1).any(): img = img / 255.0` -- the division is decided PER IMAGE;
//   :129-132  image float32 [C,H,W], mask int64 [H,W];  train.py:113 then moves the batch to channels_last.
// Here the batch arrives as uint8 [B][Hin][Win][C] + uint8 [B][Hin][Win] (pinned host memory copied once, 1 + C bytes per pixel
// instead of 4C + 8) and leaves as NHWC fp32 / bf16 in [0,1] + int64 labels, rotated per item.  HBM-bound byte work: a
// workgroup moves one 64x64 output tile; its source block is read row-wise (whole 64-byte lines for every turn count) into
// LDS and read back in the rotated order, so neither side of the transpose touches memory with a stride.
// Decode and the BICUBIC / NEAREST rescale (scale < 1) stay on the host.
#include "uh_common.h"

namespace {

constexpr int DP_TILE = 64;

// flags[b] = 1 when image b holds a byte > 1 (16 bytes per thread; bytes past the image are not read)
__global__ __launch_bounds__(256) void u8_any_gt1_kernel(const uint8_t* __restrict__ img, int64_t per_image, int B,
                                                          int* __restrict__ flags) {
    const int b = blockIdx.y;
    const uint8_t* p = img + (int64_t)b * per_image;
    bool any = false;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16; i < per_image; i += (int64_t)gridDim.x * 256 * 16) {
        if (i + 16 <= per_image && ((uintptr_t)(p + i) & 15) == 0) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + i);
            any |= ((v[0] | v[1] | v[2] | v[3]) & 0xFEFEFEFEu) != 0u;
        } else {
            for (int64_t k = i; k < per_image && k < i + 16; ++k) any |= p[k] > 1;
        }
    }
    if (__any(any) && (threadIdx.x & 63) == 0) flags[b] = 1;          // every writer stores the same value
}

// source pixel of output pixel (r, c) after `t` quarter turns counter-clockwise of an Hin x Win image
__device__ __forceinline__ void dp_source(int t, int r, int c, int Hin, int Win, int& gy, int& gx) {
    switch (t & 3) {
        case 0: gy = r; gx = c; break;
        case 1: gy = c; gx = Win - 1 - r; break;
        case 2: gy = Hin - 1 - r; gx = Win - 1 - c; break;
        default: gy = Hin - 1 - c; gx = r; break;
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void batch_prepare_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                             const int* __restrict__ turns, const int* __restrict__ flags,
                                                             T* __restrict__ out, int ld_out, int64_t* __restrict__ labels,
                                                             int Hin, int Win, int Ho, int Wo, int tilesX) {
    __shared__ uint8_t s_img[DP_TILE][DP_TILE * C + 4];
    __shared__ uint8_t s_mask[DP_TILE][DP_TILE + 4];
    const int b = blockIdx.y;
    const int t = turns ? (turns[b] & 3) : 0;
    const int R0 = (blockIdx.x / tilesX) * DP_TILE, C0 = (blockIdx.x % tilesX) * DP_TILE;
    // the source block of this output tile: a 64 x 64 window whose origin is the smallest source coordinate the tile touches
    int y_a, x_a, y_b, x_b;
    dp_source(t, R0, C0, Hin, Win, y_a, x_a);
    dp_source(t, R0 + DP_TILE - 1, C0 + DP_TILE - 1, Hin, Win, y_b, x_b);
    const int iy0 = min(y_a, y_b), ix0 = min(x_a, x_b);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const uint8_t* ib = img ? img + (int64_t)b * Hin * Win * C : nullptr;
    const uint8_t* mb = mask ? mask + (int64_t)b * Hin * Win : nullptr;
#pragma unroll 4
    for (int k = 0; k < DP_TILE / 4; ++k) {
        const int iy = ty + 4 * k, gy = iy0 + iy, gx = ix0 + tx;
        const bool ok = (unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win;
        if (ib) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                // C consecutive 64-byte runs per row: lane tx reads byte tx of run ch
                const int gxe = ix0 * C + ch * DP_TILE + tx;             // element index inside the row
                const bool oke = (unsigned)gy < (unsigned)Hin && gxe >= 0 && gxe < Win * C && gxe < (ix0 + DP_TILE) * C;
                s_img[iy][ch * DP_TILE + tx] = oke ? ib[((int64_t)gy * Win) * C + gxe] : (uint8_t)0;
            }
        }
        if (mb) s_mask[iy][tx] = ok ? mb[(int64_t)gy * Win + gx] : (uint8_t)0;
    }
    __syncthreads();
    const bool div = flags ? flags[b] != 0 : true;
#pragma unroll 4
    for (int k = 0; k < DP_TILE / 4; ++k) {
        const int r = R0 + ty + 4 * k, c = C0 + tx;
        if (r >= Ho || c >= Wo) continue;
        int gy, gx;
        dp_source(t, r, c, Hin, Win, gy, gx);
        const int iy = gy - iy0, ix = gx - ix0;
        const int64_t opix = ((int64_t)b * Ho + r) * Wo + c;
        if (out) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float u = (float)s_img[iy][ix * C + ch];
                // numpy: img.astype(float32) / 255.0 (correctly rounded fp32 division), else the raw 0 / 1 value
                out[opix * ld_out + ch] = uh_from_f32<T>(div ? __fdiv_rn(u, 255.0f) : u);
            }
        }
        if (labels) {
            const uint8_t g = s_mask[iy][ix];
            labels[opix] = g == 255 ? 2 : (g == 128 ? 1 : 0);
        }
    }
}

template <typename T>
int launch_prepare(int C, dim3 grid, hipStream_t st, const uint8_t* img, const uint8_t* mask, const int* turns, const int* flags,
                   T* out, int ld_out, int64_t* labels, int Hin, int Win, int Ho, int Wo, int tilesX) {
    switch (C) {
        case 1: hipLaunchKernelGGL((batch_prepare_kernel<T, 1>), grid, dim3(256), 0, st, img, mask, turns, flags, out, ld_out, labels, Hin, Win, Ho, Wo, tilesX); break;
        case 2: hipLaunchKernelGGL((batch_prepare_kernel<T, 2>), grid, dim3(256), 0, st, img, mask, turns, flags, out, ld_out, labels, Hin, Win, Ho, Wo, tilesX); break;
        case 3: hipLaunchKernelGGL((batch_prepare_kernel<T, 3>), grid, dim3(256), 0, st, img, mask, turns, flags, out, ld_out, labels, Hin, Win, Ho, Wo, tilesX); break;
        default: hipLaunchKernelGGL((batch_prepare_kernel<T, 4>), grid, dim3(256), 0, st, img, mask, turns, flags, out, ld_out, labels, Hin, Win, Ho, Wo, tilesX); break;
    }
    return 0;
}

}  // namespace

extern "C" int uh_batch_prepare(const uint8_t* img_u8, int C, const uint8_t* mask_u8, const int* turns, int odd_turns,
                                void* image_out, int ld_out, int64_t* labels_out, int* flags_ws, int B, int Hin, int Win,
                                int dt, uh_stream stream) {
    UH_REQUIRE(img_u8 || mask_u8, "uh_batch_prepare: neither an image nor a mask batch");
    UH_REQUIRE(B > 0 && Hin > 0 && Win > 0, "uh_batch_prepare: bad sizes B=%d H=%d W=%d", B, Hin, Win);
    UH_REQUIRE(!img_u8 || (C >= 1 && C <= 4 && image_out && ld_out >= C && flags_ws), "uh_batch_prepare: image batch needs 1..4 channels, an output with ld >= C and a B-int workspace");
    UH_REQUIRE(!mask_u8 || labels_out, "uh_batch_prepare: mask batch without a label output");
    UH_REQUIRE(dt == UH_F32 || dt == UH_BF16, "uh_batch_prepare: bad dtype %d", dt);
    UH_REQUIRE(odd_turns == 0 || odd_turns == 1, "uh_batch_prepare: odd_turns is 0 or 1");
    UH_REQUIRE(odd_turns == 0 || turns, "uh_batch_prepare: odd_turns without a turn table");
    UH_REQUIRE((int64_t)B * Hin * Win * (C > 0 ? C : 1) < (1ll << 40), "uh_batch_prepare: batch too large");
    hipStream_t st = (hipStream_t)stream;
    // every item of a batch has the same output shape: Hin x Win for even turn counts, Win x Hin for odd ones (the caller
    // states which; square images may mix them)
    const int Ho = odd_turns ? Win : Hin, Wo = odd_turns ? Hin : Win;
    const int tilesX = (Wo + DP_TILE - 1) / DP_TILE, tilesY = (Ho + DP_TILE - 1) / DP_TILE;
    if (img_u8) {
        if (hipMemsetAsync(flags_ws, 0, sizeof(int) * B, st) != hipSuccess) { uh_set_error("uh_batch_prepare: memset failed"); return UH_ELAUNCH; }
        const int64_t per = (int64_t)Hin * Win * C;
        int gx = (int)((per + 256 * 16 - 1) / (256 * 16));
        if (gx > 256) gx = 256;
        hipLaunchKernelGGL(u8_any_gt1_kernel, dim3(gx, B), dim3(256), 0, st, img_u8, per, B, flags_ws);
        UH_CHECK_LAUNCH("u8_any_gt1_kernel");
    }
    dim3 grid(tilesX * tilesY, B);
    if (dt == UH_BF16)
        launch_prepare<bf16_t>(C > 0 ? C : 1, grid, st, img_u8, mask_u8, turns, img_u8 ? flags_ws : nullptr, (bf16_t*)image_out, ld_out, labels_out, Hin, Win, Ho, Wo, tilesX);
    else
        launch_prepare<float>(C > 0 ? C : 1, grid, st, img_u8, mask_u8, turns, img_u8 ? flags_ws : nullptr, (float*)image_out, ld_out, labels_out, Hin, Win, Ho, Wo, tilesX);
    UH_CHECK_LAUNCH("batch_prepare_kernel");
    return UH_OK;
}
