// Inference-side kernels: class-index masks from the logits.
//   predict.py:27   mask_pred.argmax(dim=1)                      -> uh_argmax_classes
//   evaluate.py:111 mask_pred.argmax(dim=1); :60-62 sigmoid > 0.5 -> uh_argmax_classes / uh_threshold_mask
#include "uh_common.h"

namespace {

// logits fp32 [npix][ncls] (NHWC of the 1x1 head) -> int64 index of the FIRST maximum (torch.argmax's CPU/GPU
// rule for ties; a NaN counts as the maximum, like torch).  One thread per pixel; a wave reads ncls*256 B contiguous.
__global__ __launch_bounds__(256) void argmax_classes_kernel(const float* __restrict__ logits, int ncls,
                                                              int64_t npix, int64_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const float* l = logits + p * ncls;
    float best = l[0];
    int idx = 0;
    for (int c = 1; c < ncls; ++c) {
        const float v = l[c];
        const bool take = (v > best) || (v != v && best == best);
        if (take) { best = v; idx = c; }
    }
    out[p] = idx;
}

// binary head: mask = (sigmoid(logit) > 0.5) == (logit > 0), as 0.0/1.0 floats (evaluate.py:60-62)
__global__ __launch_bounds__(256) void threshold_mask_kernel(const float* __restrict__ logits, int64_t n,
                                                              float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(logits + i);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? 1.f : 0.f;
        *reinterpret_cast<f32x4*>(out + i) = o;
    } else {
        for (int64_t k = i; k < n; ++k) out[k] = logits[k] > 0.f ? 1.f : 0.f;
    }
}

}  // namespace

extern "C" int uh_argmax_classes(const float* logits, int64_t npix, int ncls, int64_t* out, uh_stream stream) {
    UH_REQUIRE(logits && out, "uh_argmax_classes: null pointer");
    UH_REQUIRE(npix >= 0 && ncls >= 1, "uh_argmax_classes: bad sizes npix=%lld ncls=%d", (long long)npix, ncls);
    if (npix == 0) return UH_OK;
    hipLaunchKernelGGL(argmax_classes_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, ncls, npix, out);
    UH_CHECK_LAUNCH("argmax_classes_kernel");
    return UH_OK;
}

extern "C" int uh_threshold_mask(const float* logits, int64_t n, float* out, uh_stream stream) {
    UH_REQUIRE(logits && out, "uh_threshold_mask: null pointer");
    UH_REQUIRE(n >= 0, "uh_threshold_mask: bad size");
    UH_REQUIRE(uh_aligned16(logits) && uh_aligned16(out), "uh_threshold_mask: pointers must be 16-byte aligned");
    if (n == 0) return UH_OK;
    hipLaunchKernelGGL(threshold_mask_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       logits, n, out);
    UH_CHECK_LAUNCH("threshold_mask_kernel");
    return UH_OK;
}
