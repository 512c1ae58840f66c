// post_process.hip -- device-side mask post-processing: /root/reference/utils/post_process.py
//   postprocess_mask(mask in {0,1,2}, min_area=15000, morph_kernel_size=3)                     post_process.py:51-88
//     1. remove_internal_regions (:5-48): every non-foreground pixel enclosed by the outer border of a foreground
//        (== 2) component becomes foreground  ==  fill the background components (4-connected, the dual of the
//        8-connected foreground borders cv2.findContours follows) that do not reach the image border
//     2. 3x3 opening of the foreground (cv2.morphologyEx MORPH_OPEN; OpenCV's border rule: outside counts as
//        foreground for the erosion and as background for the dilation)
//     3. 8-connected components, keep those with >= min_area pixels (cv2.connectedComponentsWithStats)
//     4. kept pixels -> 2, EVERYTHING else -> 0 (post_process.py:84-86 zeroes the class-1 background as well)
// The reference does this per image on the host with OpenCV after a device->host copy (evaluate.py:71-78); here the
// whole batch stays on the device.  Connected components = union-find over pixel indices (atomicMin links towards the
// smaller index, so the result -- the minimum pixel index of each component -- does not depend on the schedule).
// PARITY UNPINNED: OpenCV is not available in this image; the oracle is the scipy restatement oracle/post_process_ref.py.
#include "uh_common.h"

namespace {

__device__ __forceinline__ int pp_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int pp_find(const int* L, int x) {
    int p = pp_load(L + x);
    while (p != x) { x = p; p = pp_load(L + x); }
    return x;
}
__device__ __forceinline__ void pp_unite(int* L, int a, int b) {
    for (;;) {
        a = pp_find(L, a);
        b = pp_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }          // link the larger root under the smaller
        const int old = atomicMin(L + b, a);
        if (old == b) return;
        b = old;                                               // somebody re-linked b meanwhile: retry from there
    }
}

// sel[p] = 1 where the pixel belongs to the set being labelled
__global__ void pp_init_kernel(const unsigned char* __restrict__ sel, int* __restrict__ L, int* __restrict__ aux, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    L[p] = sel[p] ? (int)p : -1;
    aux[p] = 0;
}
// CONN8 = false: 4-connectivity (W, N); true: 8-connectivity (W, NW, N, NE)
template <bool CONN8>
__global__ void pp_union_kernel(const unsigned char* __restrict__ sel, int* __restrict__ L, int H, int W, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n || !sel[p]) return;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    if (x > 0 && sel[p - 1]) pp_unite(L, (int)p, (int)p - 1);
    if (y > 0) {
        if (sel[p - W]) pp_unite(L, (int)p, (int)p - W);
        if (CONN8) {
            if (x > 0 && sel[p - W - 1]) pp_unite(L, (int)p, (int)p - W - 1);
            if (x + 1 < W && sel[p - W + 1]) pp_unite(L, (int)p, (int)p - W + 1);
        }
    }
}
// root[p] = representative; MODE 0: mark components that touch the image border, MODE 1: count pixels per component
template <int MODE>
__global__ void pp_flatten_kernel(const int* __restrict__ L, int* __restrict__ root, int* __restrict__ aux, int H, int W,
                                  long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int r = -1;
    if (L[p] >= 0) {
        r = pp_find(L, (int)p);
        if (MODE == 0) {
            const int x = (int)(p % W), y = (int)((p / W) % H);
            if (x == 0 || y == 0 || x == W - 1 || y == H - 1) aux[r] = 1;
        } else {
            atomicAdd(aux + r, 1);
        }
    }
    root[p] = r;
}

__global__ void pp_select_kernel(const unsigned char* __restrict__ mask, unsigned char* __restrict__ fg,
                                 unsigned char* __restrict__ bg, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const unsigned char f = mask[p] == 2;
    fg[p] = f;
    bg[p] = !f;
}
// filled foreground: original foreground + background components that never reach the border
__global__ void pp_fill_kernel(const unsigned char* __restrict__ fg, const int* __restrict__ root, const int* __restrict__ border,
                               unsigned char* __restrict__ out, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int r = root[p];
    out[p] = fg[p] || (r >= 0 && !border[r]);
}
// ERODE: all of the k x k window set (outside the image counts as set); else DILATE: any set (outside = clear)
template <bool ERODE>
__global__ void pp_morph_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, int rad,
                                long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long img = p - (long long)y * W - x;
    bool v = ERODE;
    for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -rad; dx <= rad; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const bool s = in[img + (long long)yy * W + xx] != 0;
            if (ERODE) v = v && s; else v = v || s;
        }
    }
    out[p] = v;
}
__global__ void pp_keep_kernel(const int* __restrict__ root, const int* __restrict__ area, int min_area,
                               unsigned char* __restrict__ out, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int r = root[p];
    out[p] = (r >= 0 && area[r] >= min_area) ? 2 : 0;
}

}  // namespace

extern "C" size_t uh_postprocess_ws_bytes(int B, int H, int W) {
    const size_t n = (size_t)B * H * W;
    return 3 * n * sizeof(int) + 4 * ((n + 15) & ~(size_t)15) + 256;
}

extern "C" int uh_postprocess_masks(const uint8_t* mask, uint8_t* out, int B, int H, int W, int min_area, int morph_kernel_size,
                                    void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(mask && out && ws && B > 0 && H > 0 && W > 0, "uh_postprocess_masks: bad args");
    UH_REQUIRE(morph_kernel_size >= 1 && (morph_kernel_size & 1), "uh_postprocess_masks: the structuring element must be odd");
    UH_REQUIRE((long long)B * H * W < (1ll << 31), "uh_postprocess_masks: pixel count overflows int32");
    const size_t need = uh_postprocess_ws_bytes(B, H, W);
    if (ws_bytes < need) {
        uh_set_error("uh_postprocess_masks: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    const size_t nb = ((size_t)n + 15) & ~(size_t)15;
    int* L = (int*)ws;
    int* root = L + n;
    int* aux = root + n;
    unsigned char* fg = (unsigned char*)(aux + n);
    unsigned char* bg = fg + nb;
    unsigned char* t0 = bg + nb;
    unsigned char* t1 = t0 + nb;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    const int rad = morph_kernel_size / 2;
    // 1. hole filling = background components (4-connected) that do not touch the border
    hipLaunchKernelGGL(pp_select_kernel, grid, blk, 0, st, mask, fg, bg, n);
    hipLaunchKernelGGL(pp_init_kernel, grid, blk, 0, st, (const unsigned char*)bg, L, aux, n);
    hipLaunchKernelGGL(pp_union_kernel<false>, grid, blk, 0, st, (const unsigned char*)bg, L, H, W, n);
    hipLaunchKernelGGL(pp_flatten_kernel<0>, grid, blk, 0, st, (const int*)L, root, aux, H, W, n);
    hipLaunchKernelGGL(pp_fill_kernel, grid, blk, 0, st, (const unsigned char*)fg, (const int*)root, (const int*)aux, t0, n);
    // 2. opening
    hipLaunchKernelGGL(pp_morph_kernel<true>, grid, blk, 0, st, (const unsigned char*)t0, t1, H, W, rad, n);
    hipLaunchKernelGGL(pp_morph_kernel<false>, grid, blk, 0, st, (const unsigned char*)t1, t0, H, W, rad, n);
    // 3. 8-connected components of the opened foreground, area filter; 4. {0, 2} output
    hipLaunchKernelGGL(pp_init_kernel, grid, blk, 0, st, (const unsigned char*)t0, L, aux, n);
    hipLaunchKernelGGL(pp_union_kernel<true>, grid, blk, 0, st, (const unsigned char*)t0, L, H, W, n);
    hipLaunchKernelGGL(pp_flatten_kernel<1>, grid, blk, 0, st, (const int*)L, root, aux, H, W, n);
    hipLaunchKernelGGL(pp_keep_kernel, grid, blk, 0, st, (const int*)root, (const int*)aux, min_area, out, n);
    UH_CHECK_LAUNCH("uh_postprocess_masks");
    return UH_OK;
}
