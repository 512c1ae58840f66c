// optim.hip -- torch.nn.utils.clip_grad_norm_ (train.py:157) and torch.optim.RMSprop with momentum
// (train.py:80-81,158) over ONE flat fp32 buffer holding every parameter: two HBM-bound passes
// (sum of squares, fused clip + update) instead of ~250 foreach launches.
#include "uh_common.h"

constexpr int OPT_MAXBLK = 2048;

extern "C" size_t uh_optim_ws_bytes(int64_t n) {
    (void)n;
    return (size_t)OPT_MAXBLK * sizeof(float) + 16;
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials) {
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = g4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += g[i] * g[i];
    __shared__ float red[4];
    acc = uh_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void grad_norm_finish_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}

extern "C" int uh_grad_sumsq(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes, uh_stream stream) {
    UH_REQUIRE(g && norm_out && ws && n > 0, "uh_grad_sumsq: bad args");
    UH_REQUIRE(ws_bytes >= uh_optim_ws_bytes(n), "uh_grad_sumsq: workspace too small");
    UH_REQUIRE(uh_aligned16(g), "uh_grad_sumsq: gradient buffer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n / 4 + 255) / 256;
    if (nb > OPT_MAXBLK) nb = OPT_MAXBLK;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, n, (float*)ws);
    UH_CHECK_LAUNCH("grad_sumsq_kernel");
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)nb, norm_out);
    UH_CHECK_LAUNCH("grad_norm_finish_kernel");
    return UH_OK;
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq,
                                                      float* __restrict__ buf, int64_t n, const float* __restrict__ total_norm,
                                                      float max_norm, float lr, float alpha, float eps, float wd, float mu) {
    float coef = 1.f;
    // A NaN / infinite gradient norm means the loss was not finite (train.py:149-151 aborts before it gets here): leave the
    // parameters and the optimizer state untouched, so that a step replayed from a captured graph -- where the host can
    // only look at the loss afterwards -- cannot poison them either.
    if (total_norm && !(fabsf(total_norm[0]) < __builtin_huge_valf())) return;
    if (max_norm > 0.f && total_norm) coef = fminf(max_norm / (total_norm[0] + 1e-6f), 1.f);
    const int64_t n4 = n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    f32x4* s4 = reinterpret_cast<f32x4*>(sq);
    f32x4* b4 = reinterpret_cast<f32x4*>(buf);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = p4[i], gv = g4[i], sv = s4[i], bv = b4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gc = gv[k] * coef;
            gv[k] = gc;                                  // clip_grad_norm_ scales .grad in place
            float ge = gc + wd * pv[k];
            sv[k] = alpha * sv[k] + (1.f - alpha) * ge * ge;
            bv[k] = mu * bv[k] + ge / (sqrtf(sv[k]) + eps);
            pv[k] = pv[k] - lr * bv[k];
        }
        p4[i] = pv; g4[i] = gv; s4[i] = sv; b4[i] = bv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gc = g[i] * coef;
        g[i] = gc;
        float ge = gc + wd * p[i];
        float s = alpha * sq[i] + (1.f - alpha) * ge * ge;
        float b = mu * buf[i] + ge / (sqrtf(s) + eps);
        sq[i] = s; buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

extern "C" int uh_rmsprop_step(float* p, float* g, float* square_avg, float* momentum_buf, int64_t n,
                               const float* total_norm, float max_norm, float lr, float alpha, float eps,
                               float weight_decay, float momentum, uh_stream stream) {
    UH_REQUIRE(p && g && square_avg && momentum_buf && n > 0, "uh_rmsprop_step: bad args");
    UH_REQUIRE(uh_aligned16(p) && uh_aligned16(g) && uh_aligned16(square_avg) && uh_aligned16(momentum_buf),
               "uh_rmsprop_step: buffers must be 16-byte aligned");
    int64_t nb = (n / 4 + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, momentum_buf,
                       n, total_norm, max_norm, lr, alpha, eps, weight_decay, momentum);
    UH_CHECK_LAUNCH("rmsprop_kernel");
    return UH_OK;
}
