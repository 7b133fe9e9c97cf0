// SplitAttn merge of the local and global branches (architect_mode == 'parallel'):
// opencood/models/fusion_modules/split_attn.py:32-67 as used by hetero_fusion.py:459-470.
//   gap = mean over H, W of (a + b)   ->  fc1 (no bias) -> LayerNorm -> ReLU -> fc2 (no bias)
//   -> softmax over the two branches per channel -> out = a * w0 + b * w1
// Three HBM-bound kernels; the reduction is two-pass (fixed summation order: bit-reproducible).
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

constexpr int GAP_TOKENS = 256;   // tokens reduced by one workgroup

// partial[slot][chunk][c] = sum over the chunk's tokens of a + b
__global__ void k_gap_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial,
                              SplitSlots slots, int P, int C, int n_chunks) {
    const int slot = slots.s[blockIdx.y], chunk = blockIdx.x;
    const size_t base = (size_t)slot * P * C;
    const int t0 = chunk * GAP_TOKENS, t1 = min(P, t0 + GAP_TOKENS);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int t = t0; t < t1; ++t) acc += a[base + (size_t)t * C + c] + b[base + (size_t)t * C + c];
        partial[((size_t)blockIdx.y * n_chunks + chunk) * C + c] = acc;
    }
}

// one workgroup (C threads) per slot: finish the mean, fc1, LayerNorm, ReLU, fc2, radix softmax
__global__ void k_split_weights(const float* __restrict__ partial, const float* __restrict__ fc1,
                                const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                const float* __restrict__ fc2, float* __restrict__ w, int P, int C, int n_chunks) {
    extern __shared__ float sh[];   // gap[C], g[C], red[2]
    float* gap = sh;
    float* g = sh + C;
    float* red = sh + 2 * C;
    const int c = threadIdx.x, j = blockIdx.x;
    float acc = 0.f;
    for (int k = 0; k < n_chunks; ++k) acc += partial[((size_t)j * n_chunks + k) * C + c];
    gap[c] = acc / (float)P;
    __syncthreads();
    float v = 0.f;
    for (int i = 0; i < C; ++i) v = fmaf(fc1[(size_t)c * C + i], gap[i], v);
    g[c] = v;
    __syncthreads();
    if (c == 0) {
        float m = 0.f;
        for (int i = 0; i < C; ++i) m += g[i];
        m /= C;
        float q = 0.f;
        for (int i = 0; i < C; ++i) q += (g[i] - m) * (g[i] - m);
        red[0] = m;
        red[1] = rsqrtf(q / C + 1e-5f);
    }
    __syncthreads();
    const float h = fmaxf((v - red[0]) * red[1] * ln_g[c] + ln_b[c], 0.f);
    __syncthreads();
    g[c] = h;
    __syncthreads();
    float z0 = 0.f, z1 = 0.f;
    for (int i = 0; i < C; ++i) {
        z0 = fmaf(fc2[(size_t)c * C + i], g[i], z0);
        z1 = fmaf(fc2[(size_t)(C + c) * C + i], g[i], z1);
    }
    const float mx = fmaxf(z0, z1);
    const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
    w[((size_t)j * 2 + 0) * C + c] = e0 / (e0 + e1);
    w[((size_t)j * 2 + 1) * C + c] = e1 / (e0 + e1);
}

__global__ void k_split_combine(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                                float* __restrict__ out, SplitSlots slots, int P, int C) {
    const int j = blockIdx.y, slot = slots.s[j];
    const size_t n4 = (size_t)P * C / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)slot * P * C);
    const float4* b4 = reinterpret_cast<const float4*>(b + (size_t)slot * P * C);
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)slot * P * C);
    const int c4n = C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const float4 w0 = *reinterpret_cast<const float4*>(w + ((size_t)j * 2 + 0) * C + c4 * 4);
        const float4 w1 = *reinterpret_cast<const float4*>(w + ((size_t)j * 2 + 1) * C + c4 * 4);
        const float4 x = a4[i], y = b4[i];
        o4[i] = make_float4(x.x * w0.x + y.x * w1.x, x.y * w0.y + y.y * w1.y, x.z * w0.z + y.z * w1.z,
                            x.w * w0.w + y.w * w1.w);
    }
}

int launch_split_attn(const float* a, const float* b, float* out, const SplitSlots& slots, int n_slots,
                      const SplitWeights& sw, float* partial, float* w, int P, int C, hipStream_t st) {
    if (n_slots == 0) return HMVIT_OK;
    const int n_chunks = cdiv(P, GAP_TOKENS);
    hipLaunchKernelGGL(k_gap_partial, dim3(n_chunks, n_slots), dim3(C < 256 ? C : 256), 0, st, a, b, partial, slots, P, C,
                       n_chunks);
    HMVIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_split_weights, dim3(n_slots), dim3(C), (2 * C + 2) * sizeof(float), st, partial, sw.fc1, sw.ln_g,
                       sw.ln_b, sw.fc2, w, P, C, n_chunks);
    HMVIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_split_combine, dim3(1024, n_slots), dim3(256), 0, st, a, b, w, out, slots, P, C);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
