// In-kernel timeline of the decode step's skinny GEMMs (diagnostic build of decode_gemm.hip with -DAMT_STAMPS).
// A 6-layer chain G1 -> G2 -> G3 with the decode step's real shapes (config 2: B=32, d=512, dff=1024), every launch
// reading what the previous one wrote, is captured in a hipGraph and replayed; the stamps of the last replay give, per
// launch: gap to the previous kernel's last workgroup, first-workgroup start spread, and the median time of each phase.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAMT_STAMPS tools/ubench_chain.cpp \
//        video2music_amd/csrc/decode_gemm.hip video2music_amd/csrc/tuning.hip -o tools/ubench_chain.bin
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../video2music_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
void amt_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }

struct Launch { const char* name; DecodeGemmParams p; int wgs; unsigned long long* stamps; };

int main() {
    const int B = 32, d = 512, dff = 1024, nl = 6, VS = 160;
    auto falloc = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    float *ob = falloc(B * d), *xa = falloc(B * d), *xb = falloc(B * d), *u1 = falloc(B * d), *u2 = falloc(B * d), *u3 = falloc(B * d);
    float *qraw = falloc(B * d), *hraw = falloc(B * dff), *qkvraw = falloc(B * 3 * d), *vecs = falloc(16384);
    CK(amt_decode_gemm_init() ? hipErrorUnknown : hipSuccess);
    std::vector<Launch> L;
    for (int l = 0; l < nl; ++l) {
        // distinct packed weights per layer (as in the model: 116 MB per step cycle through the caches)
        float* p_sao = falloc((size_t)d * d); float* pf_a = falloc((size_t)d * 2 * d);
        float* p_cao = falloc((size_t)d * d); float* pf_b = falloc((size_t)dff * 2 * d);
        float* p_l2 = falloc((size_t)d * dff); float* pf_c = falloc((size_t)3 * d * (dff + d));
        DecodeGemmParams g1{};
        g1.B = B; g1.eps = 1e-5f; g1.scale = 1.f; g1.x = ob; g1.ldx = d; g1.x2 = xa; g1.ldx2 = d; g1.K1 = d; g1.K = 2 * d;
        g1.Wp = p_sao; g1.bias = vecs; g1.resid = xa; g1.ldr = d; g1.y = u1; g1.ldy = d;
        g1.n_split = d; g1.N = 2 * d; g1.Wp2 = pf_a; g1.bias2 = vecs; g1.y2 = qraw; g1.ldy2 = d;
        L.push_back({"G1 K=1024 N=1024", g1, (2 * d / 16) * 2, nullptr});
        DecodeGemmParams g2 = g1;
        g2.x2 = xb; g2.Wp = p_cao; g2.resid = xb; g2.y = u2; g2.N = d + dff; g2.Wp2 = pf_b; g2.y2 = hraw; g2.ldy2 = dff;
        L.push_back({"G2 K=1024 N=1536", g2, ((d + dff) / 16) * 2, nullptr});
        DecodeGemmParams g3{};
        g3.B = B; g3.eps = 1e-5f; g3.scale = 1.f; g3.pro = 1; g3.x = hraw; g3.ldx = dff; g3.x2 = u2; g3.ldx2 = d;
        g3.K1 = dff; g3.K = dff + d; g3.fold_g = vecs; g3.fold_c = vecs; g3.ln_w = vecs; g3.ln_b = vecs;
        g3.Wp = p_l2; g3.bias = vecs; g3.y = u3; g3.ldy = d; g3.n_split = d; g3.N = 4 * d; g3.Wp2 = pf_c; g3.bias2 = vecs;
        g3.y2 = qkvraw; g3.ldy2 = 3 * d;
        // the next layer's G1 reads `ob` and `xa`: let this G3 write them so that the chain stays dependent
        g3.y = ob;
        L.push_back({"G3 K=1536 N=2048", g3, (4 * d / 16) * 2, nullptr});
    }
    (void)VS;
    for (auto& l : L) { CK(hipMalloc(&l.stamps, (size_t)l.wgs * 8 * 8)); CK(hipMemset(l.stamps, 0, (size_t)l.wgs * 8 * 8)); l.p.stamps = l.stamps; }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& l : L) if (amt_launch_decode_gemm(l.p, s)) return 1;
    CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    const int steps = 8;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < steps; ++i) for (auto& l : L) if (amt_launch_decode_gemm(l.p, s)) return 1;
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("chain of %zu skinny GEMMs: %.3f us per launch (graph replay, stamped build)\n", L.size(), ms * 1e3 / (20.0 * steps * L.size()));

    // stamps of the last step of the last replay (each launch overwrote its buffer every step)
    unsigned long long prev_end = 0;
    const char* ph[] = {"issue loads", "rows+prologue->LDS", "barrier", "weights+MFMA", "barrier+LDS partials", "reduce+store"};
    printf("%-18s %6s %9s %9s |", "launch", "WGs", "gap(us)", "span(us)");
    for (auto n : ph) printf(" %20s", n);
    printf(" | start spread\n");
    for (auto& l : L) {
        std::vector<unsigned long long> h((size_t)l.wgs * 8);
        CK(hipMemcpy(h.data(), l.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long s0 = ~0ull, s0max = 0, e = 0;
        std::vector<std::vector<double>> phs(6);
        for (int w = 0; w < l.wgs; ++w) {
            const unsigned long long* t = &h[(size_t)w * 8];
            s0 = std::min(s0, t[0]); s0max = std::max(s0max, t[0]); e = std::max(e, t[6]);
            for (int i = 0; i < 6; ++i) phs[i].push_back((double)(t[i + 1] - t[i]) * 0.01);
        }
        printf("%-18s %6d %9.2f %9.2f |", l.name, l.wgs, prev_end ? (double)((long long)(s0 - prev_end)) * 0.01 : 0.0, (double)(e - s0) * 0.01);
        for (int i = 0; i < 6; ++i) {
            std::sort(phs[i].begin(), phs[i].end());
            printf("      %5.2f /%5.2f /%5.2f", phs[i][0], phs[i][phs[i].size() / 2], phs[i].back());
        }
        printf(" | %.2f\n", (double)(s0max - s0) * 0.01);
        prev_end = e;
    }
    printf("(phase columns: min / median / max over the launch's workgroups, microseconds; gap = first start - previous launch's last end)\n");
    return 0;
}
