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

// ---- connected_component_loss on the device (utils/connected_component_loss.py:20-59) -------------
// cv2.findContours(RETR_EXTERNAL) returns the outer border of every 8-connected foreground component that is not nested
// inside another one; what the loss uses of it:
//   * cv2.contourArea = area of the polygon through the border pixels' centres.  That polygon bounds exactly the unit
//     squares whose 4 corner pixels belong to the (hole-filled) component plus half a square for every 2x2 block with 3
//     of them (8-connectivity cuts the corner diagonally; blocks with 2 or fewer contribute a line): twice the area is
//     an INTEGER sum over 2x2 blocks -- no border following needed;
//   * cv2.boundingRect = bounding box of the component's pixels.
// "Not nested" + "outer border only" == label the HOLE-FILLED foreground: background components (4-connected) that do not
// reach the image border are filled first, which also swallows any blob sitting inside such a hole.
__global__ void cc_select_kernel(const unsigned char* __restrict__ mask, unsigned char* __restrict__ fg,
                                 unsigned char* __restrict__ bg, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const unsigned char f = mask[p] != 0;
    fg[p] = f;
    bg[p] = !f;
}
__global__ void cc_clear_kernel(int* __restrict__ x0, int* __restrict__ y0, int* __restrict__ x1, int* __restrict__ y1,
                                int* __restrict__ area2, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    x0[p] = 0x7fffffff; y0[p] = 0x7fffffff; x1[p] = -1; y1[p] = -1; area2[p] = 0;
}
// per component (indexed by its root pixel): bounding box and twice the contour area; integer atomics only
__global__ void cc_stats_kernel(const unsigned char* __restrict__ sel, const int* __restrict__ L, int* __restrict__ root,
                                int* __restrict__ x0, int* __restrict__ y0, int* __restrict__ x1, int* __restrict__ y1,
                                int* __restrict__ area2, int H, int W, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    int r = -1;
    if (sel[p]) {
        r = pp_find(L, (int)p);
        atomicMin(x0 + r, x); atomicMax(x1 + r, x);
        atomicMin(y0 + r, y); atomicMax(y1 + r, y);
    }
    root[p] = r;
    if (x + 1 < W && y + 1 < H) {                                  // the 2x2 block whose top-left pixel is p
        const int a = sel[p] != 0, b = sel[p + 1] != 0, c = sel[p + W] != 0, d = sel[p + W + 1] != 0;
        const int cnt = a + b + c + d;
        if (cnt >= 3) {                                            // its foreground pixels are mutually 8-connected
            const int q = a ? (int)p : (int)p + 1;                 // at most one corner is missing
            const int rr = a ? r : pp_find(L, q);
            atomicAdd(area2 + rr, cnt == 4 ? 2 : 1);
        }
    }
}
// one block: penalties of all components (connected_component_loss.py:35-56), summed in a fixed order in double
__global__ __launch_bounds__(1024) void cc_penalty_kernel(const int* __restrict__ root, const int* __restrict__ x0,
                                                          const int* __restrict__ y0, const int* __restrict__ x1,
                                                          const int* __restrict__ y1, const int* __restrict__ area2, int B,
                                                          int H, int W, int edge_distance, int min_area, long long n,
                                                          double* __restrict__ out) {
    __shared__ double red[2][1024];
    double pen = 0.0, cnt = 0.0;
    for (long long p = threadIdx.x; p < n; p += 1024) {
        if (root[p] != (int)p) continue;                           // one thread per component: its root pixel
        cnt += 1.0;
        const double area = 0.5 * (double)area2[p];
        if (area < (double)min_area) { pen += 1.0 - area / (double)min_area; continue; }      // :37-41
        const int wc = x1[p] - x0[p] + 1, hc = y1[p] - y0[p] + 1;  // cv2.boundingRect
        const int cx = x0[p] + wc / 2, cy = y0[p] + hc / 2;        // :45-46
        int d = cx;
        if (W - cx < d) d = W - cx;
        if (cy < d) d = cy;
        if (H - cy < d) d = H - cy;                                // :49-51
        if (d < edge_distance) pen += 1.0 - (double)d / (double)edge_distance;                // :53-56
    }
    red[0][threadIdx.x] = pen;
    red[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0] / (double)B; out[1] = red[1][0]; }
}

}  // namespace

extern "C" size_t uh_cc_loss_ws_bytes(int B, int H, int W) {
    const size_t n = (size_t)B * H * W;
    return 8 * n * sizeof(int) + 3 * ((n + 15) & ~(size_t)15) + 256;
}

// masks: DEVICE uint8 [B][H][W] (non-zero = foreground = p > 0.5, connected_component_loss.py:25); out: DEVICE double[2] =
// {sum of penalties / B (the caller multiplies by penalty_weight, :59), number of external contours}.  Same values as
// uh_cc_loss_host (which follows the borders on the host) without the device -> host copy of the masks.
extern "C" int uh_cc_loss_device(const uint8_t* masks, int B, int H, int W, int edge_distance, int min_area, void* ws,
                                 size_t ws_bytes, double* out, uh_stream stream) {
    UH_REQUIRE(masks && out && ws && B > 0 && H > 0 && W > 0 && edge_distance > 0 && min_area > 0, "uh_cc_loss_device: bad args");
    UH_REQUIRE((long long)B * H * W < (1ll << 31), "uh_cc_loss_device: pixel count overflows int32");
    const size_t need = uh_cc_loss_ws_bytes(B, H, W);
    if (ws_bytes < need) {
        uh_set_error("uh_cc_loss_device: workspace %zu < %zu bytes", ws_bytes, need);
        return UH_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    const size_t nb = ((size_t)n + 15) & ~(size_t)15;
    int* L = (int*)ws;
    int* root = L + n;
    int* aux = root + n;
    int* bx0 = aux + n;
    int* by0 = bx0 + n;
    int* bx1 = by0 + n;
    int* by1 = bx1 + n;
    int* area2 = by1 + n;
    unsigned char* fg = (unsigned char*)(area2 + n);
    unsigned char* bg = fg + nb;
    unsigned char* filled = bg + nb;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(cc_select_kernel, grid, blk, 0, st, masks, fg, bg, n);
    // holes (and whatever sits in them) become foreground: only outermost outer borders count
    hipLaunchKernelGGL(pp_init_kernel, grid, blk, 0, st, (const unsigned char*)bg, L, aux, n);
    hipLaunchKernelGGL(pp_union_kernel<false>, grid, blk, 0, st, (const unsigned char*)bg, L, H, W, n);
    hipLaunchKernelGGL(pp_flatten_kernel<0>, grid, blk, 0, st, (const int*)L, root, aux, H, W, n);
    hipLaunchKernelGGL(pp_fill_kernel, grid, blk, 0, st, (const unsigned char*)fg, (const int*)root, (const int*)aux, filled, n);
    // 8-connected components of the filled foreground, their bounding boxes and contour areas
    hipLaunchKernelGGL(pp_init_kernel, grid, blk, 0, st, (const unsigned char*)filled, L, aux, n);
    hipLaunchKernelGGL(pp_union_kernel<true>, grid, blk, 0, st, (const unsigned char*)filled, L, H, W, n);
    hipLaunchKernelGGL(cc_clear_kernel, grid, blk, 0, st, bx0, by0, bx1, by1, area2, n);
    hipLaunchKernelGGL(cc_stats_kernel, grid, blk, 0, st, (const unsigned char*)filled, (const int*)L, root, bx0, by0, bx1, by1, area2,
                       H, W, n);
    hipLaunchKernelGGL(cc_penalty_kernel, dim3(1), dim3(1024), 0, st, (const int*)root, (const int*)bx0, (const int*)by0,
                       (const int*)bx1, (const int*)by1, (const int*)area2, B, H, W, edge_distance, min_area, n, out);
    UH_CHECK_LAUNCH("uh_cc_loss_device");
    return UH_OK;
}

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
