// LSTM / GRU recurrences of the regression head `VideoRegression(regModel='lstm' | 'bilstm' | 'gru' | 'bigru')`
// (model/video_regression.py:124-135: torch.nn.LSTM / nn.GRU, batch_first, hidden = d_model; the arithmetic lives in torch,
// whose documented cell equations are restated here):
//   LSTM  i,f,g,o = split(W_ih x + b_ih + W_hh h + b_hh);  c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')
//   GRU   r = sigmoid(x_r + h_r);  z = sigmoid(x_z + h_z);  n = tanh(x_n + r * h_n);  h' = (1 - z) n + z h
//         with x_* = W_i* x + b_i* and h_* = W_h* h + b_h*
// The input projections of all time steps are one dense GEMM (amt_linear_ex_fwd) per layer and direction; this kernel is
// the part that cannot be batched over time.  One workgroup per (clip, direction) walks the sequence; the hidden-to-hidden
// matrix (G*d rows x d columns, G = 4 or 3 gates) stays in registers for the whole walk -- two threads per row, d/2 columns
// each (64 registers at d = 128) --, h lives in LDS and is read as broadcast float4s, so a step costs one half-row dot
// product per thread, one lane exchange, two barriers and d cell updates: ~1 us, no global traffic besides the step's
// G*d projected inputs and d outputs.
#include "../../include/amt_hip.h"
#include "amt_common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + __expf(-v)); }

constexpr int MAXD = 128;               // hidden size limit: d/2 weights per thread in registers

// xp: [B][L][ldxp] projected inputs of this direction (columns [0, G*d): W_ih x + b_ih); whh [G*d][d]; bhh [G*d];
// y: [B][L][ldy], this direction writes columns [0, d) of its y pointer.  reverse: walk t = L-1 .. 0.
// gridDim.y == 2 (both directions of a bidirectional layer at once): direction dir = blockIdx.y reads columns
// [dir*G*d, (dir+1)*G*d) of xp, the dir-th of the stacked whh / bhh, writes columns [dir*d, (dir+1)*d) of y; dir 1 walks backwards.
template <int G>
__global__ __launch_bounds__(2 * 4 * MAXD) void rnn_seq_kernel(const float* __restrict__ xp, int ldxp, const float* __restrict__ whh,
                                                               const float* __restrict__ bhh, float* __restrict__ y, int ldy,
                                                               int L, int d, int reverse) {
    __shared__ __attribute__((aligned(16))) float sh[MAXD];          // h_{t-1}
    __shared__ float sa[4 * MAXD];                                  // W_hh h + b_hh per gate row
    const int tid = threadIdx.x, b = blockIdx.x;
    const int row = tid >> 1, half = tid & 1, R = G * d, hc = d >> 1;
    if (gridDim.y == 2) {
        const int dir = blockIdx.y;
        xp += dir * R; whh += (size_t)dir * R * d; bhh += dir * R; y += dir * d; reverse = dir;
    }
    // this thread's half row of W_hh
    float w[MAXD / 2];
#pragma unroll
    for (int j = 0; j < MAXD / 2; ++j) w[j] = (row < R && j < hc) ? whh[(size_t)row * d + half * hc + j] : 0.f;
    const float bh = row < R ? bhh[row] : 0.f;
    float c = 0.f;                                                  // LSTM cell state of unit `tid` (tid < d)
    if (tid < d) sh[tid] = 0.f;
    __syncthreads();
    xp += (size_t)b * L * ldxp;
    y += (size_t)b * L * ldy;
    for (int s = 0; s < L; ++s) {
        const int t = reverse ? L - 1 - s : s;
        const float* xr = xp + (size_t)t * ldxp;
        // issue this step's projected inputs early: they do not depend on the recurrence
        float xg[G];
        if (tid < d) {
#pragma unroll
            for (int g = 0; g < G; ++g) xg[g] = xr[g * d + tid];
        }
        float acc = 0.f;
        const float4* h4 = reinterpret_cast<const float4*>(sh + half * hc);
#pragma unroll
        for (int j = 0; j < MAXD / 8; ++j) {
            if (4 * j < hc) {
                const float4 hv = h4[j];
                acc += w[4 * j] * hv.x + w[4 * j + 1] * hv.y + w[4 * j + 2] * hv.z + w[4 * j + 3] * hv.w;
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        if (row < R && half == 0) sa[row] = acc + bh;
        __syncthreads();
        if (tid < d) {
            float hn;
            if constexpr (G == 4) {
                const float ig = sigm(xg[0] + sa[tid]), fg = sigm(xg[1] + sa[d + tid]);
                const float gg = tanhf(xg[2] + sa[2 * d + tid]), og = sigm(xg[3] + sa[3 * d + tid]);
                c = fg * c + ig * gg;
                hn = og * tanhf(c);
            } else {
                const float r = sigm(xg[0] + sa[tid]), z = sigm(xg[1] + sa[d + tid]);
                const float n = tanhf(xg[2] + r * sa[2 * d + tid]);
                hn = (1.0f - z) * n + z * sh[tid];
            }
            y[(size_t)t * ldy + tid] = hn;
            sh[tid] = hn;           // every reader of the old h has passed the barrier above
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int32_t amt_rnn_seq_fwd(const float* xproj, int32_t ldxp, const float* w_hh, const float* b_hh, float* y, int32_t ldy,
                                   int32_t B, int32_t L, int32_t d, int32_t gates, int32_t reverse, int32_t n_dirs, void* stream) {
    AMT_CHECK_ARG(xproj && w_hh && b_hh && y, "amt_rnn_seq_fwd: null pointer");
    AMT_CHECK_ARG(gates == 3 || gates == 4, "amt_rnn_seq_fwd: gates=%d (4 = LSTM, 3 = GRU)", gates);
    AMT_CHECK_ARG(B > 0 && L > 0 && d >= 8 && d <= MAXD && d % 8 == 0, "amt_rnn_seq_fwd: hidden size %d must be a multiple of 8, at most %d", d, MAXD);
    AMT_CHECK_ARG(n_dirs == 1 || n_dirs == 2, "amt_rnn_seq_fwd: n_dirs=%d", n_dirs);
    AMT_CHECK_ARG(ldxp >= n_dirs * gates * d && ldy >= n_dirs * d, "amt_rnn_seq_fwd: bad leading dimensions");
    hipStream_t s = (hipStream_t)stream;
    const int threads = 2 * gates * d;                       // two threads per gate row
    if (gates == 4) hipLaunchKernelGGL(rnn_seq_kernel<4>, dim3(B, n_dirs), dim3(threads), 0, s, xproj, ldxp, w_hh, b_hh, y, ldy, L, d, reverse);
    else hipLaunchKernelGGL(rnn_seq_kernel<3>, dim3(B, n_dirs), dim3(threads), 0, s, xproj, ldxp, w_hh, b_hh, y, ldy, L, d, reverse);
    AMT_LAUNCH_CHECK();
    return 0;
}
