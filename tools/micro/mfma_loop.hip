// Calibration of the grouped GEMM's inner iteration on gfx950: cycles per k-tile (64 deep, bf16) of a 4-wave workgroup
// whose waves own 64 x 64 of a 128 x 128 tile -- (a) the 32 MFMAs alone, (b) with the 16 ds_read_b128 fragment reads,
// (c) with the barrier, (d) with the 8 LDS-DMA loads per wave of the next tile, (e) the DMA spread between the MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_loop.hip -o /tmp/mfma_loop && /tmp/mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const int4v& rsrc, unsigned lds, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// MODE bits: 1 fragment reads from LDS, 2 barrier per k-tile, 4 LDS-DMA of the next tile right after the barrier,
// 8 LDS-DMA spread between the MFMAs (one per four), 16 32x32x16 MFMAs instead of 16x16x32
template <int MODE, int WAVES, bool SHARE = true>
__global__ __launch_bounds__(64 * WAVES, 2) void loop_kernel(const __bf16* src, long long* out, float* sink, int nk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    constexpr int NT = 64 * WAVES;
    constexpr int STAGE = 256 * 64;  // elements: A 128 x 64 + B 128 x 64
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 1) * (128 / (WAVES / 2)), wn = (wid & 1) * 64;
    constexpr int TM = 128 / (16 * (WAVES / 2)), TN = 4;
    for (int i = tid; i < 2 * STAGE / 8; i += NT) reinterpret_cast<short8_t*>(lds)[i] = (short8_t){1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    float4_t acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // the workgroups of one XCD (block id mod 8) stream the SAME 512 KB: L2 hits, as the tiles of a GEMM that share panels
    const uint64_t a = reinterpret_cast<uint64_t>(src + (size_t)(SHARE ? (blockIdx.x & 7) : blockIdx.x) * 256 * 1024);
    int4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff;
    rs[2] = 256 * 1024 * 2;
    rs[3] = 0x00020000;
    constexpr int G = 256 * 8 / NT;  // DMA instructions per thread and k-tile
    int voff[G];
    for (int i = 0; i < G; ++i) voff[i] = (tid + NT * i) * 16;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds) + wid * 1024);
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8_t fa[TM], fb[TN];
    for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8_t, (short8_t){1, 1, 1, 1, 1, 1, 1, 1});
    for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8_t, (short8_t){1, 1, 1, 1, 1, 1, 1, 1});
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        if (MODE & (4 | 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 2) __builtin_amdgcn_s_barrier();
        const int soff = __builtin_amdgcn_readfirstlane((t & 15) * 32768);  // 16 tiles of 32 KB: the 512 KB the descriptor covers
        const unsigned dst = lds0 + (stage ^ 1) * (STAGE * 2);
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < G; ++i) glds16(rs, dst + i * NT * 16, voff[i], soff);
        }
        const __bf16* Ac = lds + stage * STAGE;
        const __bf16* Bc = Ac + 128 * 64;
        int dma = 0;
        if (MODE & 16) {  // every fragment of the k-tile requested before the first MFMA (64 registers of fragments)
            bf16x8_t ga[2][TM], gb[2][TN];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm + i * 16 + fr, chunk = ((h * 4) + fq) ^ ((row >> 1) & 7);
                    ga[h][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Ac + row * 64 + (chunk << 3)));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn + j * 16 + fr, chunk = ((h * 4) + fq) ^ ((row >> 1) & 7);
                    gb[h][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Bc + row * 64 + (chunk << 3)));
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[h][j], ga[h][i], acc[i][j], 0, 0, 0);
                        if ((MODE & 8) && ((i * TN + j) & 3) == 3 && dma < G) {
                            glds16(rs, dst + dma * NT * 16, voff[dma], soff);
                            ++dma;
                        }
                    }
            stage ^= 1;
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 64; ks += 32) {
            if (MODE & 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm + i * 16 + fr, chunk = ((ks >> 3) + fq) ^ ((row >> 1) & 7);
                    fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Ac + row * 64 + (chunk << 3)));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn + j * 16 + fr, chunk = ((ks >> 3) + fq) ^ ((row >> 1) & 7);
                    fb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Bc + row * 64 + (chunk << 3)));
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    if ((MODE & 8) && ((i * TN + j) & 3) == 3 && dma < G) {
                        glds16(rs, dst + dma * NT * 16, voff[dma], soff);
                        ++dma;
                    }
                }
        }
        stage ^= 1;
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int WAVES> void run(const char* what, const __bf16* src, long long* out, float* sink, int wgs_per_cu) {
    const int nk = 48, grid = 256 * wgs_per_cu;
    const size_t lds = 2 * 256 * 64 * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_kernel<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, out, sink, nk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, out, sink, 4096);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hl(grid);
    hipMemcpy(hl.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double ml = 0;
    for (auto v : hl) ml += (double)v;
    const double ghz = ml / grid / (ms * 1e6);  // counter ticks per ns over a 4096-tile run
    hipLaunchKernelGGL((loop_kernel<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, out, sink, nk);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    printf("%-64s waves %d, %d WG/CU: %7.0f cycles per k-tile  (long run: %.0f ticks per k-tile, %.2f ticks/ns, %.0f TFLOP/s)\n", what, WAVES,
           wgs_per_cu, m / grid / nk, ml / grid / 4096, ghz, 2.0 * 128 * 128 * 64 * 4096 * grid / (ms * 1e-3) / 1e12);
}

// Producer / consumer split: WAVES compute waves (fragment reads + MFMAs, never a vector-memory instruction in the
// loop) and ONE loader wave that issues all 32 LDS-DMA instructions of the next k-tile and waits for them; one
// s_barrier per k-tile joins the two roles.
template <int WAVES>
__global__ __launch_bounds__(64 * (WAVES + 1), 2) void split_kernel(const __bf16* src, long long* out, float* sink, int nk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    constexpr int NT = 64 * (WAVES + 1);
    constexpr int STAGE = 256 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * STAGE / 8; i += NT) reinterpret_cast<short8_t*>(lds)[i] = (short8_t){1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    const uint64_t a = reinterpret_cast<uint64_t>(src + (size_t)(blockIdx.x & 7) * 256 * 1024);
    int4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff;
    rs[2] = 256 * 1024 * 2;
    rs[3] = 0x00020000;
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds));
    const long long t0 = __builtin_readcyclecounter();
    if (wid == WAVES) {  // the loader
        const int voff = lane * 16;
        int stage = 0;
        for (int t = 0; t < nk; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int soff = __builtin_amdgcn_readfirstlane((t & 15) * 32768);
            const unsigned dst = ldsb + (stage ^ 1) * (STAGE * 2);
#pragma unroll
            for (int q = 0; q < 32; ++q) glds16(rs, dst + q * 1024, voff, soff + q * 1024);
            stage ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int wm = (wid >> 1) * (128 / (WAVES / 2)), wn = (wid & 1) * 64;
    constexpr int TM = 128 / (16 * (WAVES / 2)), TN = 4;
    float4_t acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        __builtin_amdgcn_s_barrier();
        const __bf16* Ac = lds + stage * STAGE;
        const __bf16* Bc = Ac + 128 * 64;
#pragma unroll
        for (int ks = 0; ks < 64; ks += 32) {
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm + i * 16 + fr, chunk = ((ks >> 3) + fq) ^ ((row >> 1) & 7);
                fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Ac + row * 64 + (chunk << 3)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn + j * 16 + fr, chunk = ((ks >> 3) + fq) ^ ((row >> 1) & 7);
                fb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Bc + row * 64 + (chunk << 3)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        stage ^= 1;
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int WAVES> void run_split(const char* what, const __bf16* src, long long* out, float* sink, int wgs_per_cu) {
    const int nk = 4096, grid = 256 * wgs_per_cu;
    const size_t lds = 2 * 256 * 64 * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&split_kernel<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((split_kernel<WAVES>), dim3(grid), dim3(64 * (WAVES + 1)), lds, 0, src, out, sink, 48);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((split_kernel<WAVES>), dim3(grid), dim3(64 * (WAVES + 1)), lds, 0, src, out, sink, nk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), out, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    printf("%-64s waves %d+1, %d WG/CU: %7.0f cycles per k-tile  (%.2f ticks/ns, %.0f TFLOP/s)\n", what, WAVES, wgs_per_cu, m / grid / nk,
           m / grid / (ms * 1e6), 2.0 * 128 * 128 * 64 * nk * grid / (ms * 1e-3) / 1e12);
}

int main() {
    __bf16* src;
    long long* out;
    float* sink;
    hipMalloc(&src, (size_t)512 * 256 * 1024 * 2);
    hipMemset(src, 0, (size_t)512 * 256 * 1024 * 2);
    hipMalloc(&out, 1024 * sizeof(long long));
    hipMalloc(&sink, 4096);
    for (int w = 1; w <= 2; ++w) {
        run<0, 4>("32 MFMAs per wave, operands in registers", src, out, sink, w);
        run<1, 4>("+ 16 ds_read_b128 fragment reads", src, out, sink, w);
        run<3, 4>("+ barrier per k-tile", src, out, sink, w);
        run<7, 4>("+ 8 LDS-DMA per wave after the barrier (32 KB per k-tile)", src, out, sink, w);
        run<11, 4>("  the same DMA spread between the MFMAs", src, out, sink, w);
        run<6, 4>("barrier + DMA, MFMAs on registers (no fragment reads)", src, out, sink, w);
        run<4, 4>("DMA + MFMAs on registers, no barrier", src, out, sink, w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<16 | 1, 4>("all 16 fragment reads up front, then the MFMAs", src, out, sink, w);
        run<16 | 3, 4>("  + barrier", src, out, sink, w);
        run<16 | 7, 4>("  + barrier + DMA after the barrier", src, out, sink, w);
        run<16 | 11, 4>("  + barrier + DMA spread between the MFMAs", src, out, sink, w);
    }
    run<0, 8>("8 waves: 16 MFMAs per wave, registers", src, out, sink, 1);
    run<3, 8>("8 waves: + reads + barrier", src, out, sink, 1);
    run<7, 8>("8 waves: + DMA after the barrier", src, out, sink, 1);
    run<11, 8>("8 waves: DMA spread", src, out, sink, 1);
    run_split<4>("loader wave + 4 compute waves (reads, barrier, MFMAs)", src, out, sink, 1);
    run_split<4>("loader wave + 4 compute waves", src, out, sink, 2);
    run_split<8>("loader wave + 8 compute waves", src, out, sink, 1);
    return 0;
}
