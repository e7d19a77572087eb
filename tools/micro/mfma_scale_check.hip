// v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e4m3 operands and unit block scales: which k does a lane's byte e of its
// 32-byte fragment stand for?  For a reduction only the PAIRING matters: lane (row r = l & 15, group q = l >> 4), byte e of
// the A fragment must meet lane (column c, group q), byte e of the B fragment.  Checked here with A = 16 x 128 and
// B = 16 x 128 (k contiguous) of small exact values, each lane loading bytes [32 q, 32 q + 32) of its row, against the
// host's dot products; the output block is read with the usual 16 x 16 map (col = l & 15, row = 4 (l >> 4) + reg).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_scale_check.hip -o tools/micro/mfma_scale_check && tools/micro/mfma_scale_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned char* A, const unsigned char* B, float* C) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    v8i a, b;
    memcpy(&a, A + r * 128 + 32 * q, 32);
    memcpy(&b, B + r * 128 + 32 * q, 32);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // first operand = "A" (rows of the output), second = "B" (columns)
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int e = 0; e < 4; ++e) C[(4 * q + e) * 16 + r] = acc[e];  // row 4 q + e, column r
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, ex = (v >> 3) & 15, m = v & 7;
    float x = ex == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, ex - 7);
    return s ? -x : x;
}

int main() {
    unsigned char hA[16 * 128], hB[16 * 128];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 16; };
    // values from {0, +-0.5, +-1, +-1.5, +-2, +-3}: exact products, exact fp32 sums
    const unsigned char vals[] = {0x00, 0x30, 0xb0, 0x38, 0xb8, 0x3c, 0xbc, 0x40, 0xc0, 0x44, 0xc4};
    for (auto& v : hA) v = vals[rnd() % 11];
    for (auto& v : hB) v = vals[rnd() % 11];
    unsigned char *dA, *dB;
    float* dC;
    hipMalloc(&dA, sizeof hA);
    hipMalloc(&dB, sizeof hB);
    hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    float hC[256];
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float ref = 0.f;
            for (int kk = 0; kk < 128; ++kk) ref += e4m3(hA[i * 128 + kk]) * e4m3(hB[j * 128 + kk]);
            if (hC[i * 16 + j] != ref) ++bad;
            if (hC[j * 16 + i] != ref) ++badT;
        }
    printf("C[i][j] = sum_k A[i][k] B[j][k]: %d of 256 wrong (transposed reading: %d wrong)\n", bad, badT);
    return bad && badT ? 1 : 0;
}
