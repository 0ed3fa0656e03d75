// In-kernel timeline of the fused decode phase (G1 + cross-attention in one launch) against the separate launches.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAMT_STAMPS tools/ubench_phase.cpp \
//        tools/experiments/decode_phase.hip video2music_amd/csrc/{decode_gemm,attn_decode}.hip -o tools/ubench_phase.bin
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>
#include "decode_phase.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
void amt_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }

static double bench(const char* name, std::function<void(hipStream_t, int)> body, int reps, int per_rep) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    body(s, 0); CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) body(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-64s %8.3f us per layer-slice (%d launches each)\n", name, ms * 1e3 / reps, per_rep);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return ms * 1e3 / reps;
}

int main() {
    const int B = 32, d = 512, dff = 1024, H = 8, hd = 64, S = 300, nl = 6;
    auto falloc = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    float *ob = falloc(B * d), *xa = falloc(B * d), *xb = falloc(B * d), *u1 = falloc(B * d), *u2 = falloc(B * d);
    float *qraw = falloc(B * d), *hraw = falloc(B * dff), *vecs = falloc(16384);
    unsigned* sync; CK(hipMalloc(&sync, 64 * 4)); CK(hipMemset(sync, 0, 64 * 4));
    CK(amt_decode_gemm_init() ? hipErrorUnknown : hipSuccess);
    std::vector<DecodeGemmParams> G1(nl), G2(nl);
    std::vector<AttnDecodeParams> X(nl);
    for (int l = 0; l < nl; ++l) {
        float* p_sao = falloc((size_t)d * d); float* pf_a = falloc((size_t)d * 2 * d);
        float* p_cao = falloc((size_t)d * d); float* pf_b = falloc((size_t)dff * 2 * d);
        float* kx = falloc((size_t)B * H * S * hd); float* vx = falloc((size_t)B * H * S * hd);
        DecodeGemmParams g1{};
        g1.B = B; g1.eps = 1e-5f; g1.scale = 1.f; g1.x = ob; g1.ldx = d; g1.x2 = xa; g1.ldx2 = d; g1.K1 = d; g1.K = 2 * d;
        g1.Wp = p_sao; g1.bias = vecs; g1.resid = xa; g1.ldr = d; g1.y = u1; g1.ldy = d;
        g1.n_split = d; g1.N = 2 * d; g1.Wp2 = pf_a; g1.bias2 = vecs; g1.y2 = qraw; g1.ldy2 = d;
        G1[l] = g1;
        DecodeGemmParams g2 = g1;
        g2.x2 = xb; g2.Wp = p_cao; g2.resid = xb; g2.y = u2; g2.N = d + dff; g2.Wp2 = pf_b; g2.y2 = hraw; g2.ldy2 = dff;
        g2.y = xa;      // keeps the chain dependent: the next G1 reads ob (attention) and xa
        G2[l] = g2;
        AttnDecodeParams x{};
        x.k = kx; x.v = vx; x.o = ob; x.B = B; x.H = H; x.hd = hd; x.cap = S; x.n_keys = S;
        x.q = qraw; x.ldq = d; x.d = d; x.fold_u = u1; x.fold_g = vecs; x.fold_c = vecs;
        x.fold_lnw = vecs; x.fold_lnb = vecs; x.xn = xb; x.eps = 1e-5f; x.q_scale = 0.125f;
        X[l] = x;
    }
    // separate launches (the round-1 chain): G1, cross-attention, G2
    bench("separate: decode_gemm G1 -> attn_decode (cross) -> decode_gemm G2", [&](hipStream_t s, int i) {
        const int l = i % nl;
        amt_launch_decode_gemm(G1[l], s); amt_launch_attn_decode(X[l], s); amt_launch_decode_gemm(G2[l], s);
    }, 120, 3);
    bench("GEMM-only phase kernel: G1 -> attn_decode (cross) -> G2", [&](hipStream_t s, int i) {
        const int l = i % nl;
        DecodePhaseParams p1{}; p1.g = G1[l]; amt_launch_decode_phase(p1, s);
        amt_launch_attn_decode(X[l], s);
        DecodePhaseParams p2{}; p2.g = G2[l]; amt_launch_decode_phase(p2, s);
    }, 120, 3);
    const int n1 = (2 * d / 16) * 2, na = B * H, n2 = ((d + dff) / 16) * 2;
    unsigned long long *st1, *st2;
    CK(hipMalloc(&st1, (size_t)(n1 + na) * 64)); CK(hipMalloc(&st2, (size_t)n2 * 64));
    const int delay = getenv("PHASE_DELAY") ? atoi(getenv("PHASE_DELAY")) : 0;
    printf("attention role start delay: %d x 0.27 us\n", delay);
    bench("fused: phase(G1 + cross-attention) -> phase(G2)", [&](hipStream_t s, int i) {
        const int l = i % nl;
        DecodePhaseParams p1{}; p1.g = G1[l]; p1.a = X[l]; p1.sync = sync + 4 * l; p1.stamps = st1; p1.pf_rows = delay; amt_launch_decode_phase(p1, s);
        DecodePhaseParams p2{}; p2.g = G2[l]; p2.stamps = st2; amt_launch_decode_phase(p2, s);
    }, 120, 2);
    unsigned err[64]; CK(hipMemcpy(err, sync, 64 * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < nl; ++l) if (err[4 * l + 3] || err[4 * l] || err[4 * l + 2]) printf("layer %d sync words: %u %u %u err %u\n", l, err[4 * l], err[4 * l + 1], err[4 * l + 2], err[4 * l + 3]);
    // timeline of the last fused launch
    std::vector<unsigned long long> h1((size_t)(n1 + na) * 8), h2((size_t)n2 * 8);
    CK(hipMemcpy(h1.data(), st1, h1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), st2, h2.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < n1 + na; ++w) t0 = std::min(t0, h1[(size_t)w * 8]);
    auto stat = [&](const char* name, const std::vector<unsigned long long>& h, int w0, int w1, int idx) {
        std::vector<double> v;
        for (int w = w0; w < w1; ++w) v.push_back((double)((long long)(h[(size_t)w * 8 + idx] - t0)) * 0.01);
        std::sort(v.begin(), v.end());
        printf("  %-46s min %6.2f  median %6.2f  max %6.2f us after the launch's first workgroup started\n", name, v[0], v[v.size() / 2], v.back());
    };
    printf("fused launch, GEMM tiles (%d workgroups):\n", n1);
    stat("start", h1, 0, n1, 0); stat("loads issued", h1, 0, n1, 1); stat("MFMA done, partials in LDS", h1, 0, n1, 2);
    stat("arrived on the counter (end)", h1, 0, n1, 4);
    printf("fused launch, attention workgroups (%d):\n", na);
    stat("start", h1, n1, n1 + na, 0); stat("K/V prefetch issued", h1, n1, n1 + na, 1); stat("counter satisfied", h1, n1, n1 + na, 2);
    stat("q/u rows read (K/V landed)", h1, n1, n1 + na, 3); stat("scores + PV done", h1, n1, n1 + na, 4); stat("output stored", h1, n1, n1 + na, 5);
    stat("left", h1, n1, n1 + na, 6);
    {   // placement: HW_ID (reg 4): cu_id bits 8-11, sh_id bit 12, se_id bits 13-15(+); XCC_ID (reg 20) bits 0-3
        std::vector<int> per_cu_attn(8 * 64, 0), per_cu_gemm(8 * 64, 0);
        for (int w = 0; w < n1 + na; ++w) {
            const unsigned long long v = h1[(size_t)w * 8 + 7];
            const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 0xF;
            const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            const int id = (int)(xcc * 64 + se * 16 + sh * 0 + cu) % 512;
            (w < n1 ? per_cu_gemm : per_cu_attn)[id] += 1;
        }
        int hist[4][4] = {};
        int used = 0;
        for (int i = 0; i < 512; ++i) { if (per_cu_attn[i] + per_cu_gemm[i]) ++used; hist[std::min(per_cu_gemm[i], 3)][std::min(per_cu_attn[i], 3)]++; }
        printf("placement: %d distinct (xcc,se,cu) ids used; ids with (gemm WGs, attention WGs): ", used);
        for (int g = 0; g < 4; ++g) for (int a = 0; a < 4; ++a) if (hist[g][a] && (g || a)) printf("(%d,%d)x%d ", g, a, hist[g][a]);
        printf("\n");
    }
    unsigned long long t2 = ~0ull, e1 = 0;
    for (int w = 0; w < n1 + na; ++w) e1 = std::max(e1, h1[(size_t)w * 8 + 6]);
    for (int w = 0; w < n1; ++w) e1 = std::max(e1, h1[(size_t)w * 8 + 4]);
    for (int w = 0; w < n2; ++w) t2 = std::min(t2, h2[(size_t)w * 8]);
    printf("gap to the following GEMM-only launch: %.2f us\n", (double)((long long)(t2 - e1)) * 0.01);
    t0 = t2;
    printf("GEMM-only launch G2 (%d workgroups):\n", n2);
    stat("start", h2, 0, n2, 0); stat("loads issued", h2, 0, n2, 1); stat("MFMA done, partials in LDS", h2, 0, n2, 2); stat("stored (end)", h2, 0, n2, 4);
    return 0;
}
