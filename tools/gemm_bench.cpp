// Standalone timing of mpa_gemm_f32 at the layer shapes of the cls model (no Python overhead).
// Build: hipcc --offload-arch=gfx950 -O2 -I include tools/gemm_bench.cpp -L markov-process-analysis-on-point-cloud_amd -lmpa_hip -o gpurun_out/gemm_bench
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wimplicit-const-int-float-conversion"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mpa_hip.h"

struct Shape { int M, N, K, tA, tB, stats; const char *what; };

int main()
{
    std::vector<Shape> shapes = {
        // the cls-fp32 step's tiled launches, by time (tools/gemm_tags.py)
        {4096, 2048, 256, 0, 1, 0, "fan-out fwd"},    {4096, 256, 2048, 0, 0, 0, "fan-out dX"},
        {65536, 64, 256, 0, 0, 0, "la1 dX"},          {65536, 256, 64, 0, 1, 0, "la1 fwd"},
        {8192, 1024, 128, 0, 1, 0, "fan-out fwd"},    {2048, 512, 1024, 0, 1, 1, "fc fwd"},
        {8192, 128, 1024, 0, 0, 0, "fan-out dX"},     {2048, 512, 1024, 0, 0, 0, "fc dX"},
        {2048, 1024, 512, 0, 1, 1, "fc fwd"},         {2048, 1024, 512, 0, 0, 0, "fc dX"},
        {8192, 128, 128, 0, 1, 1, "ffn fwd"},         {4096, 256, 128, 0, 1, 1, "ffn fwd"},
        {2048, 256, 1024, 0, 0, 0, "dX"},             {4096, 256, 512, 0, 1, 1, "fwd"},
        {32768, 64, 64, 0, 1, 1, "ffn fwd"},          {2048, 512, 512, 0, 1, 1, "fwd"},
        {32768, 256, 64, 0, 1, 0, "fwd"},             {32768, 64, 256, 0, 0, 0, "dX"},
        {8192, 128, 256, 0, 1, 1, "fwd"},             {16384, 512, 64, 0, 1, 0, "fwd"},
        {16384, 64, 512, 0, 0, 0, "dX"},              {4096, 512, 256, 0, 0, 0, "dX"},
        {64, 1024, 2048, 0, 1, 1, "head fwd"},        {64, 2048, 1024, 0, 0, 0, "head dX"},
        {8192, 8192, 4096, 0, 1, 0, "large"},
    };
    size_t maxA = 0, maxB = 0, maxC = 0;
    for (auto &s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.N * s.K);
        maxC = std::max(maxC, (size_t)s.M * s.N);
    }
    float *A, *B, *C, *bias, *st, *ws;
    hipMalloc(&ws, (size_t)64 << 20);
    hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4); hipMalloc(&C, maxC * 4); hipMalloc(&bias, 4096 * 4);
    hipMalloc(&st, 1024 * 2 * 1024 * 4);
    std::vector<float> h(std::max(maxA, maxB));
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), maxB * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096 * 4); hipMemset(st, 0, 1024 * 2 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0;
    for (auto &s : shapes) {
        int lda = s.tA ? s.M : s.K, ldb = s.tB ? s.K : s.N;
        auto run = [&]() {
            return mpa_gemm_f32(A, lda, s.tA, B, ldb, s.tB, s.tA ? nullptr : bias, C, s.N, s.M, s.N, s.K, 0,
                                (s.stats && !getenv("NOSTATS")) ? st : nullptr, 0, nullptr, getenv("NOWS") ? nullptr : ws,
                                (size_t)64 << 20, nullptr);
        };
        for (int i = 0; i < 3; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
        hipDeviceSynchronize();
        const int it = 50;
        hipEventRecord(e0, 0);
        for (int i = 0; i < it; ++i) run();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / it;
        double fl = 2.0 * s.M * s.N * s.K, by = 4.0 * ((double)s.M * s.K + (double)s.N * s.K + (double)s.M * s.N);
        printf("%-18s M=%6d N=%5d K=%6d tA=%d tB=%d : %8.1f us  %6.1f TFLOP/s  %5.2f TB/s\n", s.what, s.M, s.N, s.K,
               s.tA, s.tB, us, fl / us / 1e6, by / us / 1e6);
        if (s.M * (double)s.N * s.K < 1e11) total += us;
    }
    printf("sum over the step's shapes: %.1f us\n", total);
    return 0;
}
