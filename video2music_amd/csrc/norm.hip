// Row-wise normalisation and elementwise kernels (HBM-bound; one wavefront per row, float4 lanes).
//
// LayerNorm follows torch.nn.LayerNorm (biased variance, eps inside the sqrt; two passes over the
// row held in registers), used by the reference at model/rpr.py:48-50,59-69 and inside torch's
// TransformerEncoderLayer.  RMSNorm follows model/custom_transformer.py:38-45.
// RoPE follows model/rotate_operation.py:111-165 (interleaved pairs, cached cos/sin).
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int MAX_CHUNKS = 8;   // dim <= 64 lanes * 4 floats * 8 = 2048

template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   float* __restrict__ y, int rows, int dim, float eps,
                                                   const float* __restrict__ post) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    float4 v[MAX_CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
        int i = (c * 64 + lane) * 4;
        if (i < dim) {
            v[c] = ld4(xr + i);
            if (resid) {
                float4 r = ld4(resid + (size_t)row * dim + i);
                v[c].x += r.x; v[c].y += r.y; v[c].z += r.z; v[c].w += r.w;
            }
            s += RMS ? (v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w)
                     : (v[c].x + v[c].y + v[c].z + v[c].w);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / dim + eps);
    } else {
        mean = s / dim;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            int i = (c * 64 + lane) * 4;
            if (i < dim) {
                float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
                q += dx * dx + dy * dy + dz * dz + dw * dw;
            }
        }
        q = wave_sum(q);
        rstd = 1.0f / sqrtf(q / dim + eps);
    }
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
        int i = (c * 64 + lane) * 4;
        if (i < dim) {
            float4 o;
            o.x = (v[c].x - mean) * rstd; o.y = (v[c].y - mean) * rstd;
            o.z = (v[c].z - mean) * rstd; o.w = (v[c].w - mean) * rstd;
            if (w) { float4 g = ld4(w + i); o.x *= g.x; o.y *= g.y; o.z *= g.z; o.w *= g.w; }
            if (b) { float4 g = ld4(b + i); o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w; }
            v[c] = o;
        }
    }
    if (w2) {   // stacked second LayerNorm (last decoder layer's norm3 followed by decoder.norm)
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            int i = (c * 64 + lane) * 4;
            if (i < dim) s2 += v[c].x + v[c].y + v[c].z + v[c].w;
        }
        float m2 = wave_sum(s2) / dim, q2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            int i = (c * 64 + lane) * 4;
            if (i < dim) {
                float dx = v[c].x - m2, dy = v[c].y - m2, dz = v[c].z - m2, dw = v[c].w - m2;
                q2 += dx * dx + dy * dy + dz * dz + dw * dw;
            }
        }
        float r2 = 1.0f / sqrtf(wave_sum(q2) / dim + eps);
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            int i = (c * 64 + lane) * 4;
            if (i < dim) {
                float4 g = ld4(w2 + i), h = ld4(b2 + i);
                v[c].x = (v[c].x - m2) * r2 * g.x + h.x; v[c].y = (v[c].y - m2) * r2 * g.y + h.y;
                v[c].z = (v[c].z - m2) * r2 * g.z + h.z; v[c].w = (v[c].w - m2) * r2 * g.w + h.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
        int i = (c * 64 + lane) * 4;
        if (i < dim) {
            if (post) { const float4 a = ld4(post + (size_t)row * dim + i); v[c].x += a.x; v[c].y += a.y; v[c].z += a.z; v[c].w += a.w; }
            st4(y + (size_t)row * dim + i, v[c]);
        }
    }
}

// x viewed as [n0][seq][n2][hd]; cache flat [seq*cache_half*2] reinterpreted as [n0'][seq][hd/2][2]
__global__ void rope_kernel(const float* __restrict__ x, const float* __restrict__ cache, float* __restrict__ y,
                            int n0, int seq, int n2, int hd, long total_pairs, int bcast) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_pairs) return;
    const int half = hd / 2;
    int p = (int)(i % half);
    long r = i / half;
    r /= n2;                                 // drop n2 (cache broadcasts over it)
    int s = (int)(r % seq);
    int a = bcast ? 0 : (int)(r / seq);     // a single cache slab broadcasts over the leading axis
    size_t ci = (((size_t)a * seq + s) * half + p) * 2;
    float c = cache[ci], sn = cache[ci + 1];
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    y[2 * i] = x0 * c - x1 * sn;
    y[2 * i + 1] = x1 * c + x0 * sn;
}

__global__ void concat_features_kernel(const float* __restrict__ sem, int sem_dim, const float* __restrict__ scene,
                                       const float* __restrict__ motion, int motion_dim,
                                       const float* __restrict__ emotion, int emo_dim,
                                       float* __restrict__ out, int rows, int ld_out) {
    const int row = blockIdx.x;
    float* o = out + (size_t)row * ld_out;
    const int o_scene = sem_dim, o_motion = sem_dim + 1, o_emo = o_motion + motion_dim, o_end = o_emo + emo_dim;
    for (int c = threadIdx.x; c < ld_out; c += blockDim.x) {
        float v;
        if (c < o_scene) v = sem[(size_t)row * sem_dim + c];
        else if (c < o_motion) v = scene[row];
        else if (c < o_emo) v = motion[(size_t)row * motion_dim + (c - o_motion)];
        else if (c < o_end) v = emotion[(size_t)row * emo_dim + (c - o_emo)];
        else v = 0.f;
        o[c] = v;
    }
}

__global__ void chord_embed_kernel(const int64_t* __restrict__ root, const int64_t* __restrict__ attr,
                                   const float* __restrict__ key, const float* __restrict__ PR, const float* __restrict__ PA,
                                   const float* __restrict__ wkey, const float* __restrict__ bias,
                                   const float* __restrict__ pe, float* __restrict__ out, int L, int d) {
    const int row = blockIdx.x;          // b*L + l
    const int b = row / L, l = row - b * L;
    const int r = (int)root[row], a = (int)attr[row];
    const float kv = key[b];
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        float4 pr = ld4(PR + (size_t)r * d + c), pa = ld4(PA + (size_t)a * d + c);
        float4 wk = ld4(wkey + c), bb = ld4(bias + c), pp = ld4(pe + (size_t)l * d + c);
        float4 o;
        o.x = ((pr.x + pa.x) + kv * wk.x + bb.x) + pp.x;
        o.y = ((pr.y + pa.y) + kv * wk.y + bb.y) + pp.y;
        o.z = ((pr.z + pa.z) + kv * wk.z + bb.z) + pp.z;
        o.w = ((pr.w + pa.w) + kv * wk.w + bb.w) + pp.w;
        st4(out + (size_t)row * d + c, o);
    }
}

// Differential attention tail (custom_transformer.py:818-826): y = RMSNorm_hd(o1 - lambda * o2) * w * out_scale per (clip,
// head, position) row of hd <= 128 values; one wavefront per row.
__global__ void diff_subln_kernel(const float* __restrict__ o1, const float* __restrict__ o2, const float* __restrict__ w,
                                  float* __restrict__ y, int rows, int hd, float lambda, float out_scale, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * hd;
    float v[2], ss = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < hd ? o1[base + c] - lambda * o2[base + c] : 0.f;
        ss += v[k] * v[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float r = rsqrtf(ss / (float)hd + eps);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        if (c < hd) y[base + c] = v[k] * r * w[c] * out_scale;
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const float4 u = ld4(a + 4 * i), v = ld4(b + 4 * i);
        st4(y + 4 * i, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
    }
}

// y[r][:] = x[r][:] * row_scale[r] (+ add[r][:]): the dropped video rows of the V1/V2/V3 classes (dropTokenRate)
__global__ void row_scale_add_kernel(const float* __restrict__ x, const float* __restrict__ row_scale, const float* __restrict__ add,
                                     float* __restrict__ y, long n4, int dim4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const float m = row_scale[i / dim4];
        const float4 u = ld4(x + 4 * i);
        float4 v = make_float4(u.x * m, u.y * m, u.z * m, u.w * m);
        if (add) { const float4 a = ld4(add + 4 * i); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        st4(y + 4 * i, v);
    }
}

}  // namespace

int32_t amt_launch_row_scale_add(const float* x, const float* row_scale, const float* add, float* y, int rows, int dim, hipStream_t stream) {
    AMT_CHECK_ARG(rows > 0 && dim > 0 && dim % 4 == 0, "row_scale_add: bad shape rows=%d dim=%d", rows, dim);
    const long n4 = (long)rows * (dim / 4);
    hipLaunchKernelGGL(row_scale_add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, row_scale, add, y, n4, dim / 4);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_diff_subln(const float* o1, const float* o2, const float* w, float* y, int rows, int hd, float lambda,
                              float out_scale, float eps, hipStream_t stream) {
    AMT_CHECK_ARG(rows > 0 && hd > 0 && hd <= 128, "diff_subln: bad shape rows=%d hd=%d", rows, hd);
    hipLaunchKernelGGL(diff_subln_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, o1, o2, w, y, rows, hd, lambda, out_scale, eps);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_add(const float* a, const float* b, float* y, long n, hipStream_t stream) {
    AMT_CHECK_ARG(n > 0 && n % 4 == 0, "add: element count must be a positive multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a, b, y, n / 4);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_layernorm(const float* x, const float* resid, const float* w, const float* b,
                             const float* w2, const float* b2, float* y, int rows, int dim, float eps,
                             hipStream_t stream, const float* post) {
    AMT_CHECK_ARG(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 64 * 4 * MAX_CHUNKS, "layernorm: bad shape rows=%d dim=%d", rows, dim);
    hipLaunchKernelGGL(norm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, resid, w, b, w2, b2, y, rows, dim, eps, post);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_rmsnorm(const float* x, const float* w, float* y, int rows, int dim, float eps, hipStream_t stream,
                           const float* resid) {
    AMT_CHECK_ARG(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 64 * 4 * MAX_CHUNKS, "rmsnorm: bad shape rows=%d dim=%d", rows, dim);
    hipLaunchKernelGGL(norm_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, resid, w, nullptr, nullptr, nullptr, y, rows, dim, eps, nullptr);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_rope(const float* x, const float* cache, float* y, int n0, int seq, int n2, int hd,
                        int cache_half, hipStream_t stream) {
    AMT_CHECK_ARG(n0 > 0 && seq > 0 && n2 > 0 && hd > 0 && hd % 2 == 0, "rope: bad shape");
    // the [n0'][seq][hd/2][2] reinterpretation of the [seq][cache_half][2] cache (rotate_operation.py:148-149)
    // must either cover the n0 leading slabs or consist of exactly one slab, which then broadcasts
    const int bcast = (cache_half * 2 == hd) ? 1 : 0;
    AMT_CHECK_ARG(bcast || (long)cache_half * 2 >= (long)n0 * hd, "rope: cache has %d pairs per position, need %d (or exactly %d)",
                  cache_half, n0 * hd / 2, hd / 2);
    long total = (long)n0 * seq * n2 * (hd / 2);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, cache, y, n0, seq, n2, hd, total, bcast);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_concat_features(const float* sem, int sem_dim, const float* scene, const float* motion, int motion_dim,
                                   const float* emotion, int emo_dim, float* out, int rows, int ld_out, hipStream_t stream) {
    AMT_CHECK_ARG(ld_out >= sem_dim + 1 + motion_dim + emo_dim, "concat: ld_out too small");
    hipLaunchKernelGGL(concat_features_kernel, dim3(rows), dim3(256), 0, stream, sem, sem_dim, scene, motion, motion_dim,
                       emotion, emo_dim, out, rows, ld_out);
    AMT_LAUNCH_CHECK();
    return 0;
}

int32_t amt_launch_chord_embed(const int64_t* root, const int64_t* attr, const float* key, const float* PR, const float* PA,
                               const float* wkey, const float* bias, const float* pe, float* out,
                               int B, int L, int d, hipStream_t stream) {
    AMT_CHECK_ARG(d % 4 == 0, "chord_embed: d must be a multiple of 4");
    hipLaunchKernelGGL(chord_embed_kernel, dim3(B * L), dim3(128), 0, stream, root, attr, key, PR, PA, wkey, bias, pe, out, L, d);
    AMT_LAUNCH_CHECK();
    return 0;
}
