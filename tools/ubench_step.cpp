// In-kernel timeline of ONE decode step's launch chain at config 2 (B=32, d=512, dff=1024, H=8, S=300): per layer
// self-attention (folded prologue, new K/V row) -> G1 -> cross-attention (folded prologue) -> G2 -> G3, six layers, captured in a
// hipGraph and replayed; diagnostic build (-DAMT_STAMPS) of the library's own kernel files.  Prints, per launch of the last
// replay: the gap to the previous launch, the span, and the median time of each phase over the launch's workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAMT_STAMPS tools/ubench_step.cpp \
//        video2music_amd/csrc/{decode_gemm,attn_decode,tuning}.hip -o tools/ubench_step.bin
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../video2music_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
void amt_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }

struct Launch { std::string name; int kind; DecodeGemmParams g; AttnDecodeParams a; int wgs; unsigned long long* stamps; };

int main(int argc, char** argv) {
    const int B = 32, d = 512, dff = 1024, H = 8, hd = 64, S = 300, nl = 6;
    const int t = argc > 1 ? atoi(argv[1]) : 511;
    const int cap = 1024 + (argc > 2 ? atoi(argv[2]) : 0);      // rows per (clip, head) of the self-attention cache: 1024 + padding
    auto falloc = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    float *ob = falloc(B * d), *xa = falloc(B * d), *xb = falloc(B * d), *u1 = falloc(B * d), *u2 = falloc(B * d), *u3 = falloc(B * d);
    float *qraw = falloc(B * d), *hraw = falloc(B * dff), *qkvraw = falloc(B * 3 * d), *vecs = falloc(16384);
    int* pos; CK(hipMalloc(&pos, 64)); CK(hipMemcpy(pos, &t, 4, hipMemcpyHostToDevice));
    CK(amt_decode_gemm_init() ? hipErrorUnknown : hipSuccess);
    std::vector<Launch> L;
    for (int l = 0; l < nl; ++l) {
        float* p_sao = falloc((size_t)d * d); float* pf_a = falloc((size_t)d * 2 * d);
        float* p_cao = falloc((size_t)d * d); float* pf_b = falloc((size_t)dff * 2 * d);
        float* p_l2 = falloc((size_t)d * dff); float* pf_c = falloc((size_t)3 * d * (dff + d));
        float* kc = falloc((size_t)B * H * cap * hd); float* vc = falloc((size_t)B * H * cap * hd);
        float* kx = falloc((size_t)B * H * S * hd); float* vx = falloc((size_t)B * H * S * hd);
        float* Er = falloc((size_t)cap * hd);
        AttnDecodeParams a{};
        a.k = kc; a.v = vc; a.o = ob; a.B = B; a.H = H; a.hd = hd; a.cap = cap; a.pos = pos; a.Er = Er; a.er_len = cap;
        a.q = qkvraw; a.ldq = 3 * d; a.d = d; a.fold_u = u3; a.fold_g = vecs; a.fold_c = vecs; a.fold_lnw = vecs; a.fold_lnb = vecs;
        a.xn = xa; a.new_kv = 1; a.k_new = kc; a.v_new = vc; a.eps = 1e-5f; a.q_scale = 0.125f;
        L.push_back({"SA  self-attention", 1, {}, a, B * H, nullptr});
        DecodeGemmParams g1{};
        g1.B = B; g1.eps = 1e-5f; g1.scale = 1.f; g1.x = ob; g1.ldx = d; g1.x2 = xa; g1.ldx2 = d; g1.K1 = d; g1.K = 2 * d;
        g1.Wp = p_sao; g1.bias = vecs; g1.resid = xa; g1.ldr = d; g1.y = u1; g1.ldy = d;
        g1.n_split = d; g1.N = 2 * d; g1.Wp2 = pf_a; g1.bias2 = vecs; g1.y2 = qraw; g1.ldy2 = d;
        L.push_back({"G1  K=1024 N=1024", 0, g1, {}, (2 * d / 16) * 2, nullptr});
        AttnDecodeParams x{};
        x.k = kx; x.v = vx; x.o = ob; x.B = B; x.H = H; x.hd = hd; x.cap = S; x.n_keys = S;
        x.q = qraw; x.ldq = d; x.d = d; x.fold_u = u1; x.fold_g = vecs; x.fold_c = vecs; x.fold_lnw = vecs; x.fold_lnb = vecs;
        x.xn = xb; x.eps = 1e-5f; x.q_scale = 0.125f;
        L.push_back({"CA  cross-attention", 1, {}, x, B * H, nullptr});
        DecodeGemmParams g2 = g1;
        g2.x2 = xb; g2.Wp = p_cao; g2.resid = xb; g2.y = u2; g2.N = d + dff; g2.Wp2 = pf_b; g2.y2 = hraw; g2.ldy2 = dff;
        L.push_back({"G2  K=1024 N=1536", 0, g2, {}, ((d + dff) / 16) * 2, nullptr});
        DecodeGemmParams g3{};
        g3.B = B; g3.eps = 1e-5f; g3.scale = 1.f; g3.pro = 1; g3.x = hraw; g3.ldx = dff; g3.x2 = u2; g3.ldx2 = d;
        g3.K1 = dff; g3.K = dff + d; g3.fold_g = vecs; g3.fold_c = vecs; g3.ln_w = vecs; g3.ln_b = vecs;
        g3.Wp = p_l2; g3.bias = vecs; g3.y = u3; g3.ldy = d; g3.n_split = d; g3.N = 4 * d; g3.Wp2 = pf_c; g3.bias2 = vecs;
        g3.y2 = qkvraw; g3.ldy2 = 3 * d;
        L.push_back({"G3  K=1536 N=2048", 0, g3, {}, (4 * d / 16) * 2, nullptr});
    }
    for (auto& l : L) {
        CK(hipMalloc(&l.stamps, (size_t)l.wgs * 64)); CK(hipMemset(l.stamps, 0, (size_t)l.wgs * 64));
        if (l.kind) l.a.stamps = l.stamps; else l.g.stamps = l.stamps;
    }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto enqueue = [&]() { for (auto& l : L) { if (l.kind ? amt_launch_attn_decode(l.a, s) : amt_launch_decode_gemm(l.g, s)) exit(1); } };
    enqueue(); CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    const int steps = 8;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < steps; ++i) enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("t = %d: %.1f us per step of %zu launches (stamped build; no sampling head)\n", t, ms * 1e3 / (10.0 * steps), L.size());
    const char* gph[] = {"issue loads", "rows+prologue", "barrier", "weights+MFMA", "partials", "reduce+store"};
    const char* aph[] = {"prologue: query ready", "keys streamed (wave 0)", "group merge", "barrier", "final combine+store"};
    unsigned long long prev_end = 0;
    for (size_t li = 0; li < 10 && li < L.size(); ++li) {          // layers 0 and 1
        auto& l = L[li];
        std::vector<unsigned long long> h((size_t)l.wgs * 8);
        CK(hipMemcpy(h.data(), l.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        const int last = l.kind ? 5 : 6, nph = l.kind ? 5 : 6;
        unsigned long long s0 = ~0ull, s0max = 0, e = 0;
        std::vector<std::vector<double>> phs(nph);
        for (int w = 0; w < l.wgs; ++w) {
            const unsigned long long* st = &h[(size_t)w * 8];
            s0 = std::min(s0, st[0]); s0max = std::max(s0max, st[0]); e = std::max(e, st[last]);
            for (int i = 0; i < nph; ++i) phs[i].push_back((double)((long long)(st[i + 1] - st[i])) * 0.01);
        }
        printf("%-20s gap %5.2f  span %6.2f  start spread %4.2f |", l.name.c_str(), prev_end ? (double)((long long)(s0 - prev_end)) * 0.01 : 0.0,
               (double)(e - s0) * 0.01, (double)(s0max - s0) * 0.01);
        for (int i = 0; i < nph; ++i) { std::sort(phs[i].begin(), phs[i].end()); printf("  %s %.2f", l.kind ? aph[i] : gph[i], phs[i][phs[i].size() / 2]); }
        printf("\n");
        prev_end = e;
    }
    return 0;
}
