// Load-time algebra for the decode step: LayerNorm folded through the projection that follows it.
//
// The post-norm decoder (model/rpr.py:59-69) computes, per sub-block,  u = A.Wo^T + bo + r  and then feeds
// LayerNorm(u) to the next projection  y = LayerNorm(u).W^T + b.  With W' = W o gamma (columns scaled) this is
//      y_n = (raw_n - mu * g_n) * rstd + c_n ,   raw = u . W'^T ,  g_n = sum_k W'[n][k] ,  c_n = W[n].beta + b_n
// and  raw = A.(W'.Wo)^T + r.W'^T + W'.bo  is linear in the producer's *inputs* (A, r): the producing skinny GEMM
// emits raw next to u in one launch, and the consumer only needs u's row statistics (mu, rstd).  That removes one
// dependent kernel per LayerNorm from the decode chain.  The kernels here build W', the transposed Wo, and the
// vectors g, c, dv = W'.bo (accumulated in fp64, rounded once); the matrix product W'.Wo runs on the dense GEMM.
#include "amt_common.h"
#include "kernels.h"

namespace {

__global__ void scale_cols_kernel(const float* __restrict__ W, const float* __restrict__ gamma, float* __restrict__ out,
                                  size_t total, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = W[i] * gamma[i % K];
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < C && r < R) out[(size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

// one wave per output row n: g = sum_k Ws[n][k], c = sum_k W[n][k]*beta[k] + b[n], dv = sum_k Ws[n][k]*bo[k]
__global__ __launch_bounds__(64) void fold_vectors_kernel(const float* __restrict__ W, const float* __restrict__ Ws,
                                                          const float* __restrict__ beta, const float* __restrict__ b,
                                                          const float* __restrict__ bo, float* __restrict__ g,
                                                          float* __restrict__ c, float* __restrict__ dv, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double sg = 0.0, sc = 0.0, sd = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double ws = (double)Ws[(size_t)n * K + k];
        sg += ws;
        sc += (double)W[(size_t)n * K + k] * (double)beta[k];
        sd += ws * (double)bo[k];
    }
    for (int off = 32; off > 0; off >>= 1) {
        sg += __shfl_xor(sg, off, 64); sc += __shfl_xor(sc, off, 64); sd += __shfl_xor(sd, off, 64);
    }
    if (lane == 0) { g[n] = (float)sg; c[n] = (float)(sc + (double)(b ? b[n] : 0.f)); dv[n] = (float)sd; }
}

}  // namespace

int32_t amt_launch_scale_cols(const float* W, const float* gamma, float* out, int N, int K, hipStream_t stream) {
    const size_t total = (size_t)N * K;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, gamma, out, total, K);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_transpose(const float* in, float* out, int R, int C, hipStream_t stream) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream, in, out, R, C);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_fold_vectors(const float* W, const float* Ws, const float* beta, const float* b, const float* bo,
                                float* g, float* c, float* dv, int N, int K, hipStream_t stream) {
    hipLaunchKernelGGL(fold_vectors_kernel, dim3(N), dim3(64), 0, stream, W, Ws, beta, b, bo, g, c, dv, K);
    AMT_LAUNCH_CHECK();
    return 0;
}
