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
        {65536, 64, 64, 0, 1, 1, "la0 ffn fwd"},      {65536, 64, 3, 0, 1, 1, "la0 conv_res fwd"},
        {65536, 128, 64, 0, 1, 0, "la1 kv fwd"},      {32768, 64, 64, 0, 1, 1, "la1 ffn fwd"},
        {32768, 64, 128, 0, 1, 1, "la1 fc2 fwd"},     {16384, 256, 64, 0, 1, 0, "la3 kv fwd"},
        {8192, 128, 128, 0, 1, 1, "la3 ffn fwd"},     {8192, 512, 128, 0, 1, 0, "la4 kv fwd"},
        {4096, 256, 256, 0, 1, 1, "la4 ffn fwd"},     {4096, 1024, 256, 0, 1, 0, "la5 kv fwd"},
        {2048, 512, 512, 0, 1, 1, "la5 ffn fwd"},     {2048, 512, 1024, 0, 1, 1, "la5 fc2 fwd"},
        {2048, 1024, 512, 0, 1, 1, "conv4 fwd"},
        {65536, 64, 64, 0, 0, 0, "la0 ffn dX"},       {65536, 64, 128, 0, 0, 0, "la1 kv dX"},
        {4096, 256, 1024, 0, 0, 0, "la5 kv dX"},      {2048, 512, 1024, 0, 0, 0, "conv4 dX"},
        {64, 64, 65536, 1, 0, 0, "la0 ffn dW"},       {128, 64, 65536, 1, 0, 0, "la1 kv dW"},
        {64, 128, 32768, 1, 0, 0, "la1 fc2 dW"},      {1024, 256, 4096, 1, 0, 0, "la5 kv dW"},
        {1024, 512, 2048, 1, 0, 0, "conv4 dW"},       {512, 512, 2048, 1, 0, 0, "la5 ffn dW"},
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
    }
    return 0;
}
