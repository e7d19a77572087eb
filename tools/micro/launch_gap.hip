// What does a dependent kernel launch cost inside a replayed hipGraph, and does the size of its kernel-argument block
// matter?  The grouped kernels of the step take their problems BY VALUE (up to ~3.7 KB of arguments per launch).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/launch_gap.hip -o tools/micro/launch_gap && tools/micro/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N> struct Big { int v[N]; };
template <int N> __global__ void touch(Big<N> b, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.v[0] == 12345) *out = b.v[N - 1]; }
// a launch with some work in it: every workgroup reads 16 KB
template <int N> __global__ void work(Big<N> b, const float4* src, float* out) {
    float4 a = src[blockIdx.x * 1024 + threadIdx.x];
    float4 c = src[blockIdx.x * 1024 + 256 + threadIdx.x];
    if (a.x + c.y == 1234.5f + b.v[0]) out[0] = a.x;
}

template <int N> int run(hipStream_t st, int* d, const float4* src, float* fo, int launches, bool with_work, int grid) {
    Big<N> b{};
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < launches; ++i) {
        if (with_work) hipLaunchKernelGGL(work<N>, dim3(grid), dim3(256), 0, st, b, src, fo);
        else hipLaunchKernelGGL(touch<N>, dim3(grid), dim3(256), 0, st, b, d);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%4d-byte arguments, grid %4d, %s: %.2f us per launch\n", (int)sizeof(Big<N>), grid, with_work ? "16 KB read per workgroup" : "empty", us / reps / launches);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int* d;
    float4* src;
    float* fo;
    CK(hipMalloc(&d, 4));
    CK(hipMalloc(&src, 1024 * 1024 * 16));
    CK(hipMalloc(&fo, 4));
    CK(hipMemset(src, 0, 1024 * 1024 * 16));
    for (int grid : {1, 256, 768}) {
        for (int w = 0; w < 2; ++w) {
            if (run<4>(st, d, src, fo, 400, w, grid)) return 1;
            if (run<64>(st, d, src, fo, 400, w, grid)) return 1;
            if (run<256>(st, d, src, fo, 400, w, grid)) return 1;
            if (run<920>(st, d, src, fo, 400, w, grid)) return 1;
        }
    }
    return 0;
}
