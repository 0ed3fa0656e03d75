// Flash-style fp32 attention on the CDNA4 matrix cores for the teacher-forced / encoder paths:
//   O = softmax(Q K^T (+ skewed relative-position bias) (+ causal mask)) V      per (clip, head)
// without materialising the L x L scores the reference builds (model/rpr.py:387-414) or the
// (B*H, L, L) skew temporaries of model/rpr.py:439-455.
//
// Workgroup = 4 waves = 128 query rows of one (clip, head); each wave owns 32 rows.  Keys/values
// are visited in tiles of 32 staged in LDS ([32][hd+4] floats -> conflict-free ds_read_b128).
// Everything is computed TRANSPOSED so the softmax row reduction never crosses lanes:
//   S^T[key][query] = K . Q^T     v_mfma_f32_32x32x2_f32, A = K tile (LDS), B = Q^T (registers)
// puts one query per lane (column) and its 32 keys in 16 registers x 2 lane halves; the running
// max / sum and the rescale of O^T[d][query] are then per-lane scalars, and P^T is already in the
// B-operand layout of the second product
//   O^T[d][query] += V^T . P^T    A = V^T read from the LDS V tile, B = P^T accumulator registers.
//
// Relative positions (bias[i][j] = q_i . Er[er_len-1-(i-j)], j <= i): the distances of a
// 32x32 tile lie in two aligned chunks of 32; R^T[m][query] = Er_chunk . Q^T is one MFMA tile per chunk
// (A = Er rows straight from L2).  Consecutive key tiles share a chunk, so each tile computes ONE new
// chunk into a 2-slot per-wave LDS ring and reads the bias back along the skew diagonal.
#include "amt_common.h"
#include "kernels.h"

namespace {

constexpr int QB = 128;      // query rows per workgroup
constexpr int KT = 32;       // keys per tile

// NOMASK: relative positions WITHOUT the causal mask (forward(mask=False)); a separate instantiation so that the causal kernel
// of the hot path keeps its exact instruction stream
template <int HD, bool RPR, bool NOMASK = false>
// without the relative-position term (cross-attention, the encoder, GQA) and head_dim <= 64 the kernel is held at 168 registers = THREE
// waves per SIMD (three 51 KB workgroups per CU): two resident workgroups drift into lockstep -- both in their MFMA phases, then both in
// their softmax -- and the matrix pipe idles meanwhile; a third one fills the gaps
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((!RPR && HD <= 64) ? 3 : 1, (!RPR && HD <= 64) ? 3 : 8)))
void attn_prefill_kernel(AttnParams p) {
    constexpr int HDP = HD < 32 ? 32 : HD;   // head_dim 16: the O^T tile is still 32 rows of d; V columns 16..31 are zeros in LDS
    constexpr int LD = HDP + 4;
    constexpr int NS = HD / 8;           // ds_read_b128 k-groups per operand row
    constexpr int ND = HDP / 32;         // 32-wide d tiles of O^T
    constexpr int SCR = 2 * 32 * 33 > 32 * (HD + 1) ? 2 * 32 * 33 : 32 * (HD + 1);   // two 32x33 distance chunks / the O transpose
    __shared__ __attribute__((aligned(16))) float Ks[KT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LD];
    __shared__ __attribute__((aligned(16))) float scr_all[4 * SCR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid, query block slowest: all (clip, head) pairs of the heaviest (last) causal query block are dispatched first, then
    // the next lighter one, ... (longest-processing-time order: the tail of the launch is made of 4-tile blocks instead of
    // 32-tile ones), and the linear id of a (clip, head) is the same in every group, so under round-robin block->XCD
    // placement all query blocks of a (clip, head) read its K/V through the same XCD's L2
    const int n_bh = p.H * p.B, n_qb = (p.Lq + QB - 1) / QB;
    const int qblk = n_qb - 1 - (int)(blockIdx.x / n_bh);
    const int bh = blockIdx.x % n_bh;
    const int h = bh % p.H, b = bh / p.H;
    const int hk = h / p.kv_group;
    const int I0 = qblk * QB, i0 = I0 + wave * 32, iq = i0 + li;
    float* scr = scr_all + wave * SCR;

    const float* qp = p.q + (size_t)b * p.q_bs + (size_t)h * p.q_hs;
    const float* kp = p.k + (size_t)b * p.k_bs + (size_t)hk * p.k_hs;
    const float* vp = p.v + (size_t)b * p.v_bs + (size_t)hk * p.v_hs;

    // Q^T fragments: qreg[4s+e] = Q[iq][8s + 4*lh + e] (* q_scale)
    const float qs = p.q_scale == 0.f ? 1.f : p.q_scale;
    float qreg[HD / 2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float4 t = (iq < p.Lq) ? ld4(qp + (size_t)iq * p.q_ls + 8 * s + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
        qreg[4 * s + 0] = t.x * qs; qreg[4 * s + 1] = t.y * qs; qreg[4 * s + 2] = t.z * qs; qreg[4 * s + 3] = t.w * qs;
    }

    f32x16 oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int k_end = p.Lk;
    if (p.causal) k_end = min(p.Lk, I0 + QB);
    const int n_tiles = (k_end + KT - 1) / KT;

    // K/V tile staging: 32 rows x HD/4 float4 per tensor over 256 threads
    constexpr int F4_ROW = HD / 4, F4_TILE = KT * F4_ROW, PER_T = (F4_TILE + 255) / 256;
    static_assert(F4_TILE % 256 == 0 || F4_TILE < 256, "a K/V tile is a whole number of 256-thread passes (or part of one)");
    float4 kst[PER_T], vst[PER_T];
    auto gload = [&](int j0) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int f = tid + i * 256;
            const int r = f / F4_ROW, c = (f - r * F4_ROW) * 4;
            // no guarded load: keys past the end re-read the last row (their scores are masked to -inf, so P = 0 meets a finite
            // V row); a load under a branch hides the number of loads in flight from the compiler, which then drains them all
            // (s_waitcnt vmcnt(0)) in front of the relative-position prefetch
            const int j = min(j0 + r, p.Lk - 1);
            kst[i] = ld4(kp + (size_t)j * p.k_ls + c);
            vst[i] = ld4(vp + (size_t)j * p.v_ls + c);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int f = tid + i * 256;
            if (f < F4_TILE) {
                const int r = f / F4_ROW, c = (f - r * F4_ROW) * 4;
                st4(&Ks[r * LD + c], kst[i]);
                st4(&Vs[r * LD + c], vst[i]);
            }
        }
    };

    float4 er_next[NS];                    // Er rows of the chunk the next tile will need (relative positions only)
    bool er_primed = false;
    // (double-buffering the tiles in LDS for one barrier per tile was measured: 587 vs 584 us at config 2 — no gain, more LDS)
    if constexpr (HD < HDP) {              // the padding columns of the V tile: written once, never touched by lstore
        for (int f = tid; f < KT * (HDP - HD); f += 256) Vs[(f / (HDP - HD)) * LD + HD + f % (HDP - HD)] = 0.f;
    }
    gload(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int j0 = kt * KT;
        __syncthreads();                 // previous tile fully consumed
        lstore();
        __syncthreads();
        if (kt + 1 < n_tiles) gload(j0 + KT);
        if (p.causal && j0 > i0 + 31) continue;          // whole tile above this wave's diagonal

        // ---- S^T = K . Q^T ----
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a = ld4(&Ks[li * LD + 8 * s + 4 * lh]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * s + 0], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * s + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * s + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * s + 3], sacc, 0, 0, 0);
        }

        // ---- relative-position term.  With D = i0 - j0 = 32k the distances of this tile, D-31 .. D+31, lie in the
        //      aligned chunks k-1 and k of 32 distances each; chunk c holds R^T[m][query] = Er[er_len-1-(32c+m)] . q.
        //      Tiles are visited with k descending, so chunk k was produced by the previous tile (as its k-1) and
        //      only chunk k-1 is new: one extra MFMA tile per key tile, kept in a 2-slot per-wave LDS ring. ----
        // Without the causal mask (forward(mask=False), model/video_music_transformer.py:978-982) the tiles above the
        // diagonal are visible too; `_skew` (model/rpr.py:439-455) leaves the relative term ZERO for every key j > i, so those
        // tiles (k < 0) and the upper half of the diagonal tile (k == 0, distance < 0) take the plain Q.K score.
        if constexpr (RPR) if (!NOMASK || i0 >= j0) {
            const int k = (i0 - j0) / 32;
            // The Er rows of a chunk come straight from L2 (the table is shared by every clip and head).  The rows of the NEXT
            // tile's chunk (k-2) are requested as soon as this tile's fragments are in the matrix pipe, so their latency
            // overlaps the softmax and the PV product instead of stalling the next tile.
            auto er_load = [&](int c, float4 (&ef)[NS]) {
                int row = p.er_len - 1 - (32 * c + li);
                row = max(0, min(p.er_len - 1, row));      // out-of-range rows belong to masked pairs
                const float* ep = p.Er + (size_t)row * HD + 4 * lh;
#pragma unroll
                for (int s = 0; s < NS; ++s) ef[s] = ld4(ep + 8 * s);
            };
            auto chunk = [&](int c, const float4 (&ef)[NS]) {
                f32x16 racc;
#pragma unroll
                for (int e = 0; e < 16; ++e) racc[e] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    racc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[s].x, qreg[4 * s + 0], racc, 0, 0, 0);
                    racc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[s].y, qreg[4 * s + 1], racc, 0, 0, 0);
                    racc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[s].z, qreg[4 * s + 2], racc, 0, 0, 0);
                    racc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[s].w, qreg[4 * s + 3], racc, 0, 0, 0);
                }
                float* slot = scr + (c & 1) * (32 * 33);
#pragma unroll
                for (int e = 0; e < 16; ++e) slot[li * 33 + (e & 3) + 8 * (e >> 2) + 4 * lh] = racc[e];
            };
            if (!er_primed) {                                // first tile this wave computes: both chunks are new
                float4 e0[NS];
                er_load(k, e0);
                chunk(k, e0);
                er_load(k - 1, er_next);
                er_primed = true;
            }
            if (k >= 1) chunk(k - 1, er_next);
            if (k >= 2) er_load(k - 2, er_next);             // for the next tile (it reads chunk k-2 as its k-1)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int krow = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int dlt = li - krow;                  // distance - D
                const int c = dlt >= 0 ? k : k - 1;
                const float r = scr[(c & 1) * (32 * 33) + li * 33 + (dlt & 31)];
                if constexpr (NOMASK) sacc[e] += (k == 0 && dlt < 0) ? 0.f : r;
                else sacc[e] += r;
            }
        }

        // ---- mask + online softmax (one query per lane, keys split over the two lane halves) ----
        float tmax = -INFINITY;
        if (j0 + KT <= p.Lk && (!p.causal || j0 + KT - 1 <= i0)) {      // the whole tile is visible to every query of the wave
#pragma unroll
            for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (j >= p.Lk || (p.causal && j > iq)) sacc[e] = -INFINITY;
                tmax = fmaxf(tmax, sacc[e]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = __expf(sacc[e] - m_use);
            psum += sacc[e];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;

        // ---- O^T += V^T . P^T : k-step e pairs key krow(e,0) (lanes 0-31) with krow(e,1) ----
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int krow = (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const float a = Vs[krow * LD + dt * 32 + li];
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sacc[e], oacc[dt], 0, 0, 0);
            }
        }
    }

    // ---- normalise, transpose through the wave's scratch, store rows coalesced ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    __syncthreads();                     // all waves are done with their scratch as R band
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dd = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (HD >= 32 || dd < HD) scr[li * (HD + 1) + dd] = oacc[dt][e] * inv;
        }
    float* op = p.o + (size_t)b * p.o_bs + (size_t)h * p.o_hs;
    for (int r = 0; r < 32; ++r) {
        const int i = i0 + r;
        if (i >= p.Lq) break;
        for (int c = lane; c < HD; c += 64) op[(size_t)i * p.o_ls + c] = scr[r * (HD + 1) + c];
    }
}

// Split-key variant for small grids (few clips x heads x query blocks: MultiheadGQA at config 4 has 128 of the 128-row blocks
// for 256 CUs, and the last causal block alone walks 64 key tiles).  A workgroup owns ONE 32-query row block; its 4 waves
// take the key tiles round-robin (wave w: tiles w, w+4, ...), each staging its own tiles in its own LDS region -- no barrier
// in the loop --, and the 4 partial softmax states (m, l, O) are merged through LDS at the end.  Grid = 4x the blocks of the
// 128-row kernel and a critical path 4x shorter; K/V tiles are read once per 32 query rows instead of once per 128 (L2
// traffic the large-grid shapes would not want).  No relative-position term.
template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_splitk_kernel(AttnParams p) {
    constexpr int LD = HD + 4, NS = HD / 8, ND = HD / 32;
    constexpr int TILE = KT * LD;
    __shared__ __attribute__((aligned(16))) float KV[4][2 * TILE];      // per wave: K tile | V tile; reused for the merge

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n_bh = p.H * p.B, n_qb = (p.Lq + 31) / 32;
    const int qblk = n_qb - 1 - (int)(blockIdx.x / n_bh);               // heaviest causal row blocks first
    const int bh = blockIdx.x % n_bh;
    const int h = bh % p.H, b = bh / p.H;
    const int hk = h / p.kv_group;
    const int i0 = qblk * 32, iq = i0 + li;
    float* Ks = KV[wave];
    float* Vs = KV[wave] + TILE;

    const float* qp = p.q + (size_t)b * p.q_bs + (size_t)h * p.q_hs;
    const float* kp = p.k + (size_t)b * p.k_bs + (size_t)hk * p.k_hs;
    const float* vp = p.v + (size_t)b * p.v_bs + (size_t)hk * p.v_hs;
    const float qs = p.q_scale == 0.f ? 1.f : p.q_scale;
    float qreg[HD / 2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float4 t = (iq < p.Lq) ? ld4(qp + (size_t)iq * p.q_ls + 8 * s + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
        qreg[4 * s + 0] = t.x * qs; qreg[4 * s + 1] = t.y * qs; qreg[4 * s + 2] = t.z * qs; qreg[4 * s + 3] = t.w * qs;
    }
    f32x16 oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int k_end = p.Lk;
    if (p.causal) k_end = min(p.Lk, i0 + 32);
    const int n_tiles = (k_end + KT - 1) / KT;

    // one wave stages a whole tile: 32 rows x HD/4 float4 per tensor over 64 lanes
    constexpr int F4_ROW = HD / 4, PER_W = KT * F4_ROW / 64;
    float4 kst[PER_W], vst[PER_W];
    auto gload = [&](int j0) {
#pragma unroll
        for (int i = 0; i < PER_W; ++i) {
            const int f = lane + i * 64;
            const int r = f / F4_ROW, c = (f - r * F4_ROW) * 4;
            const int j = min(j0 + r, p.Lk - 1);          // (unguarded: see attn_prefill_kernel)
            kst[i] = ld4(kp + (size_t)j * p.k_ls + c);
            vst[i] = ld4(vp + (size_t)j * p.v_ls + c);
        }
    };
    if (wave < n_tiles) gload(wave * KT);
    for (int kt = wave; kt < n_tiles; kt += 4) {
        const int j0 = kt * KT;
#pragma unroll
        for (int i = 0; i < PER_W; ++i) {
            const int f = lane + i * 64;
            const int r = f / F4_ROW, c = (f - r * F4_ROW) * 4;
            st4(&Ks[r * LD + c], kst[i]);
            st4(&Vs[r * LD + c], vst[i]);
        }
        // the tile is staged by this wave for this wave alone: LDS operations of one wave execute in order, so no s_barrier is
        // needed -- but other LANES read these words below, which the compiler must not reorder around the stores
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kt + 4 < n_tiles) gload(j0 + 4 * KT);
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a = ld4(&Ks[li * LD + 8 * s + 4 * lh]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * s + 0], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * s + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * s + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * s + 3], sacc, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (j >= p.Lk || (p.causal && j > iq)) sacc[e] = -INFINITY;
            tmax = fmaxf(tmax, sacc[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = __expf(sacc[e] - m_use);
            psum += sacc[e];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int krow = (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const float a = Vs[krow * LD + dt * 32 + li];
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sacc[e], oacc[dt], 0, 0, 0);
            }
        }
        // (and the next tile's stores stay behind this tile's fragment reads)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- merge the 4 waves' states: partial O (un-normalised, [query][d]) | m | l in the wave's own LDS region ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float* mg = KV[wave];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[li * (HD + 1) + dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh] = oacc[dt][e];
    if (lh == 0) { mg[32 * (HD + 1) + li] = m_run; mg[32 * (HD + 1) + 32 + li] = l_tot; }
    __syncthreads();
    float* op = p.o + (size_t)b * p.o_bs + (size_t)h * p.o_hs;
    for (int f = tid; f < 32 * HD; f += 256) {
        const int r = f / HD, c = f - r * HD;
        if (i0 + r >= p.Lq) continue;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mx = fmaxf(mx, KV[w][32 * (HD + 1) + r]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = KV[w][32 * (HD + 1) + r];
            const float a = (mw == -INFINITY) ? 0.f : __expf(mw - mx);
            num += a * KV[w][r * (HD + 1) + c];
            den += a * KV[w][32 * (HD + 1) + 32 + r];
        }
        op[(size_t)(i0 + r) * p.o_ls + c] = den > 0.f ? num / den : 0.f;
    }
}

template <int HD>
int32_t launch_hd(const AttnParams& p, hipStream_t stream) {
    dim3 grid(cdiv(p.Lq, QB) * p.H * p.B);
    // fewer than two 128-row blocks per CU and a long key range: one 32-row block per workgroup, keys split over its waves
    if (!p.Er && HD >= 32 && HD <= 64 && grid.x < 512 && p.Lk >= 256) {
        hipLaunchKernelGGL((attn_prefill_splitk_kernel<(HD >= 32 && HD <= 64 ? HD : 64)>), dim3(cdiv(p.Lq, 32) * p.H * p.B), dim3(256), 0, stream, p);
        return 0;
    }
    if (p.Er && !p.causal) hipLaunchKernelGGL((attn_prefill_kernel<HD, true, true>), grid, dim3(256), 0, stream, p);
    else if (p.Er) hipLaunchKernelGGL((attn_prefill_kernel<HD, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((attn_prefill_kernel<HD, false>), grid, dim3(256), 0, stream, p);
    return 0;
}

}  // namespace

int32_t amt_launch_attn_prefill(const AttnParams& p, hipStream_t stream) {
    AMT_CHECK_ARG(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, "attn_prefill: bad shape");
    AMT_CHECK_ARG(p.kv_group >= 1 && p.H % p.kv_group == 0, "attn_prefill: bad kv_group %d", p.kv_group);
    AMT_CHECK_ARG(p.Er == nullptr || (p.Lq == p.Lk && p.Lq <= p.er_len),
                  "attn_prefill: relative positions need self-attention with L=%d <= er_len=%d", p.Lq, p.er_len);
    AMT_CHECK_ARG(p.q_ls % 4 == 0 && p.k_ls % 4 == 0 && p.v_ls % 4 == 0, "attn_prefill: row strides must be multiples of 4 floats");
    switch (p.hd) {
        case 16: launch_hd<16>(p, stream); break;
        case 32: launch_hd<32>(p, stream); break;
        case 64: launch_hd<64>(p, stream); break;
        case 128: launch_hd<128>(p, stream); break;
        default: AMT_CHECK_ARG(false, "attn_prefill: head_dim %d not in {16,32,64,128}", p.hd);
    }
    AMT_LAUNCH_CHECK();
    return 0;
}
