// Single-query attention for the autoregressive decode step (HBM-bound K/V streaming).
//
// One 512-thread workgroup per (clip, head).  The head's K and V rows ([cap][hd] fp32, contiguous)
// are streamed straight to registers with 16-byte lanes: hd/4 lanes cover one key row, so one
// wave-instruction fetches 64/(hd/4) whole rows = 1 KiB fully coalesced.  Each lane group keeps an
// online-softmax state (m, l, o[4]) for the keys it has seen; groups and waves are merged once at
// the end (shuffles, then LDS).  No score row is ever materialised.
//
// Relative-position self-attention (model/rpr.py:391-394 in closed form): for the
// query at position t and key j the score is q.k_j + q.Er[er_len-1-(t-j)], i.e. q.(k_j + e_j) with
// the Er row read from L2 (the table is shared by every clip and head of a layer).
// Cross-attention (torch MultiheadAttention at model/rpr.py:62-63) is the same kernel with Er=null
// and a fixed key count.
#include <stdlib.h>

#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int NW = 8;            // waves per workgroup
constexpr int UNROLL = 4;        // key groups per wave and batch

template <int HD>
struct Batch {                   // one register-resident batch of UNROLL key groups
    float4 k[UNROLL], v[UNROLL], e[UNROLL];
};

__device__ __forceinline__ float4 ld4_stream(const float* p, bool nt) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    if (nt) { v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
    return ld4(p);
}

template <int HD, bool NT>
__device__ __forceinline__ void load_kv(Batch<HD>& bt, const float* kb, const float* vb, int j0, int sub, int n_keys) {
    constexpr int KPW = 64 / (HD / 4);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int j = j0 + u * NW * KPW + sub;
        const int jj = j < n_keys ? j : 0;
        bt.k[u] = ld4_stream(kb + (size_t)jj * HD, NT);
        bt.v[u] = ld4_stream(vb + (size_t)jj * HD, NT);
    }
}
template <int HD>
__device__ __forceinline__ void load_er(Batch<HD>& bt, const float* eb, int j0, int sub, int n_keys) {
    constexpr int KPW = 64 / (HD / 4);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int j = j0 + u * NW * KPW + sub;
        bt.e[u] = ld4(eb + (size_t)(j < n_keys ? j : 0) * HD);
    }
}
template <int HD, bool RPR, bool NT>
__device__ __forceinline__ void load_batch(Batch<HD>& bt, const float* kb, const float* vb, const float* eb,
                                           int j0, int sub, int n_keys) {
    load_kv<HD, NT>(bt, kb, vb, j0, sub, n_keys);
    if (RPR) load_er<HD>(bt, eb, j0, sub, n_keys);
}

template <int HD, bool RPR>
__device__ __forceinline__ void consume_batch(const Batch<HD>& bt, const float4 q4, int j0, int sub, int n_keys,
                                              float& m, float& l, float4& o) {
    constexpr int LPK = HD / 4, KPW = 64 / LPK;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        float4 k4 = bt.k[u];
        if (RPR) { k4.x += bt.e[u].x; k4.y += bt.e[u].y; k4.z += bt.e[u].z; k4.w += bt.e[u].w; }
        float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        s = group_sum<LPK>(s);
        if (j0 + u * NW * KPW + sub < n_keys) {
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn), pj = __expf(s - mn);
            l = l * alpha + pj;
            o.x = o.x * alpha + pj * bt.v[u].x; o.y = o.y * alpha + pj * bt.v[u].y;
            o.z = o.z * alpha + pj * bt.v[u].z; o.w = o.w * alpha + pj * bt.v[u].w;
            m = mn;
        }
    }
}

template <int HD, bool RPR, bool NT>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(AttnDecodeParams p) {
    constexpr int LPK = HD / 4;          // lanes per key row
    constexpr int KPW = 64 / LPK;        // keys per wave-instruction
    constexpr int STRIDE = NW * KPW * UNROLL;     // keys per workgroup batch
    __shared__ float sm_m[NW], sm_l[NW];
    __shared__ __attribute__((aligned(16))) float sm_o[NW][HD];

    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPK, sub = lane / LPK;
    const float* kb = p.k + ((size_t)b * p.H + h) * p.cap * HD + c * 4;
    const float* vb = p.v + ((size_t)b * p.H + h) * p.cap * HD + c * 4;

    // wave w takes key groups w, w+NW, ...; two batches are kept in flight (load i+1 before using i).
    // The first K/V batch and q do not depend on the step position: they are issued before `pos` is
    // read (rows past the current length are fetched but never used; they lie inside the cache).
    Batch<HD> b0, b1;
    int j0 = wave * KPW;
    load_kv<HD, NT>(b0, kb, vb, j0, sub, p.cap);
    const float4 q4 = ld4(p.q + ((size_t)b * p.H + h) * HD + c * 4);
    const int t = p.pos ? *p.pos : (p.n_keys - 1);
    const int n_keys = t + 1;
    const float* eb = RPR ? p.Er + (size_t)(p.er_len - 1 - t) * HD + c * 4 : nullptr;   // Er row of key 0
    if (RPR) load_er<HD>(b0, eb, j0, sub, n_keys);
    float m = -INFINITY, l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    while (j0 < n_keys) {
        if (j0 + STRIDE < n_keys) load_batch<HD, RPR, NT>(b1, kb, vb, eb, j0 + STRIDE, sub, n_keys);
        consume_batch<HD, RPR>(b0, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
        if (j0 >= n_keys) break;
        if (j0 + STRIDE < n_keys) load_batch<HD, RPR, NT>(b0, kb, vb, eb, j0 + STRIDE, sub, n_keys);
        consume_batch<HD, RPR>(b1, q4, j0, sub, n_keys, m, l, o);
        j0 += STRIDE;
    }

    // merge the KPW lane groups of the wave (lanes with equal c)
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        float4 o2;
        o2.x = __shfl_xor(o.x, off, 64); o2.y = __shfl_xor(o.y, off, 64);
        o2.z = __shfl_xor(o.z, off, 64); o2.w = __shfl_xor(o.w, off, 64);
        const float mn = fmaxf(m, m2);
        const float a1 = (m == -INFINITY) ? 0.f : __expf(m - mn);
        const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        l = l * a1 + l2 * a2;
        o.x = o.x * a1 + o2.x * a2; o.y = o.y * a1 + o2.y * a2;
        o.z = o.z * a1 + o2.z * a2; o.w = o.w * a1 + o2.w * a2;
        m = mn;
    }
    if (sub == 0) {
        if (c == 0) { sm_m[wave] = m; sm_l[wave] = l; }
        st4(&sm_o[wave][c * 4], o);
    }
    __syncthreads();
    if (wave == 0 && sub == 0) {
        float mn = sm_m[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mn = fmaxf(mn, sm_m[w]);
        float lt = 0.f;
        float4 ot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float a = (sm_m[w] == -INFINITY) ? 0.f : __expf(sm_m[w] - mn);
            const float4 ow = ld4(&sm_o[w][c * 4]);
            lt += sm_l[w] * a;
            ot.x += ow.x * a; ot.y += ow.y * a; ot.z += ow.z * a; ot.w += ow.w * a;
        }
        const float inv = 1.0f / lt;
        ot.x *= inv; ot.y *= inv; ot.z *= inv; ot.w *= inv;
        st4(p.o + ((size_t)b * p.H + h) * HD + c * 4, ot);
    }
}

template <int HD>
void launch_decode(const AttnDecodeParams& p, hipStream_t stream) {
    dim3 grid(p.H, p.B);
    // K/V are streamed once per launch and exceed the 256 MiB Infinity Cache per step: non-temporal loads keep
    // the step's re-used bytes (63 MB of weights, activations) resident instead (measured +8 % tokens/s at
    // config 2).  AMT_NT overrides for experiments: bit 0 = self-attention, bit 1 = cross-attention.
    static int nt_mask = -1;
    if (nt_mask < 0) { const char* e = getenv("AMT_NT"); nt_mask = e ? atoi(e) : 3; }
    if (p.Er) {
        if (nt_mask & 1) hipLaunchKernelGGL((attn_decode_kernel<HD, true, true>), grid, dim3(NW * 64), 0, stream, p);
        else hipLaunchKernelGGL((attn_decode_kernel<HD, true, false>), grid, dim3(NW * 64), 0, stream, p);
    } else {
        if (nt_mask & 2) hipLaunchKernelGGL((attn_decode_kernel<HD, false, true>), grid, dim3(NW * 64), 0, stream, p);
        else hipLaunchKernelGGL((attn_decode_kernel<HD, false, false>), grid, dim3(NW * 64), 0, stream, p);
    }
}

}  // namespace

int32_t amt_launch_attn_decode(const AttnDecodeParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.H > 0 && p.cap > 0, "attn_decode: bad shape B=%d H=%d cap=%d", p.B, p.H, p.cap);
    AMT_CHECK_ARG(p.pos != nullptr || (p.n_keys > 0 && p.n_keys <= p.cap), "attn_decode: n_keys=%d outside (0,%d]", p.n_keys, p.cap);
    AMT_CHECK_ARG(p.Er == nullptr || p.er_len >= p.cap, "attn_decode: er_len=%d smaller than the key capacity %d", p.er_len, p.cap);
    switch (p.hd) {
        case 16: launch_decode<16>(p, stream); break;
        case 32: launch_decode<32>(p, stream); break;
        case 64: launch_decode<64>(p, stream); break;
        case 128: launch_decode<128>(p, stream); break;
        default: AMT_CHECK_ARG(false, "attn_decode: head_dim %d not in {16,32,64,128}", p.hd);
    }
    AMT_LAUNCH_CHECK();
    return 0;
}
