// Graph-replay micro-benchmark of the decode-step kernels (links the library's kernel files directly).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench_kernels.cpp video2music_amd/csrc/{decode_gemm,attn_decode,sample,tuning}.hip -o tools/ubench_kernels.bin
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include "../video2music_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
void amt_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }

static double bench(const char* name, std::function<void(hipStream_t)> launch, int reps = 500) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    launch(s); CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-60s %7.3f us/kernel\n", name, ms * 1e3 / reps);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int B = 32, d = 512, H = 8, hd = 64, cap = 1024, S = 300, dff = 1024;
    float *x, *w, *wp, *bias, *lnw, *lnb, *res, *y, *xn, *kc, *vc, *Er, *q, *o;
    int* pos;
    size_t kvn = (size_t)B * H * cap * hd;
    CK(hipMalloc(&x, B * 1024 * 4)); CK(hipMalloc(&w, 1536 * 1024 * 4)); CK(hipMalloc(&wp, 1536 * 1024 * 4));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&lnw, 4096)); CK(hipMalloc(&lnb, 4096)); CK(hipMalloc(&res, B * 1536 * 4));
    CK(hipMalloc(&y, B * 1536 * 4)); CK(hipMalloc(&xn, B * 1024 * 4)); CK(hipMalloc(&kc, kvn * 4)); CK(hipMalloc(&vc, kvn * 4));
    CK(hipMalloc(&Er, cap * hd * 4)); CK(hipMalloc(&q, B * d * 4)); CK(hipMalloc(&o, B * d * 4)); CK(hipMalloc(&pos, 64));
    CK(hipMemset(x, 0, B * 1024 * 4)); CK(hipMemset(w, 0, 1536 * 1024 * 4)); CK(hipMemset(bias, 0, 4096 * 4));
    CK(hipMemset(lnw, 0, 4096)); CK(hipMemset(lnb, 0, 4096)); CK(hipMemset(res, 0, B * 1536 * 4));
    CK(hipMemset(kc, 0, kvn * 4)); CK(hipMemset(vc, 0, kvn * 4)); CK(hipMemset(Er, 0, cap * hd * 4)); CK(hipMemset(q, 0, B * d * 4));
    struct { int N, K, ln; } shapes[] = {{1536, 512, 1}, {512, 512, 0}, {512, 512, 1}, {1024, 512, 1}, {512, 1024, 0}};
    for (auto sh : shapes) {
        amt_launch_pack_weight(w, wp, sh.N, sh.K, nullptr); CK(hipDeviceSynchronize());
        for (int dbg : {0}) {
            DecodeGemmParams g{};
            g.x = x; g.ldx = sh.K; g.Wp = wp; g.bias = bias; g.B = B; g.N = sh.N; g.K = sh.K;
            if (sh.ln) { g.ln_w = lnw; g.ln_b = lnb; g.xn = xn; }
            g.eps = 1e-5f; g.resid = res; g.ldr = sh.N; g.scale = 1.f; g.y = y; g.ldy = sh.N;
            char n[128]; snprintf(n, 128, "decode_gemm N=%d K=%d ln=%d dbg=%d", sh.N, sh.K, sh.ln, dbg);
            bench(n, [&](hipStream_t s) { amt_launch_decode_gemm(g, s); });
        }
    }
    for (int t : {0, 15, 127, 299, 511, 1023}) {
        int hp = t; CK(hipMemcpy(pos, &hp, 4, hipMemcpyHostToDevice));
        AttnDecodeParams a{};
        a.q = q; a.k = kc; a.v = vc; a.o = o; a.B = B; a.H = H; a.hd = hd; a.cap = cap; a.pos = pos; a.Er = Er; a.er_len = cap;
        char n[128]; snprintf(n, 128, "attn_decode self t=%d", t);
        double us = bench(n, [&](hipStream_t s) { amt_launch_attn_decode(a, s); }, 200);
        printf("    -> %.0f GB/s algorithmic\n", (double)B * (t + 1) * d * 8 / us / 1e3);
        a.Er = nullptr;
        snprintf(n, 128, "attn_decode no-rpr t=%d", t);
        bench(n, [&](hipStream_t s) { amt_launch_attn_decode(a, s); }, 200);
    }
    return 0;
}
