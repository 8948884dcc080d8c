// Probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands and unit scales.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned char* A, const unsigned char* B, float* C, int scale) {
    // A: [16 rows][128 k] bytes, B: [128 k][16 cols] bytes; hypothesis: lane l holds A[l&15][32*(l>>4) + j], B[32*(l>>4)+j][l&15]
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    union { i32x8 v; unsigned char b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        a.b[j] = A[r * 128 + 32 * g + j];
        b.b[j] = B[(32 * g + j) * 16 + r];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, scale, 0, scale);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];   // row = 4*(lane>>4)+i, col = lane&15
}

static unsigned char enc(int v) {   // small integers / halves in e4m3fn
    switch (v) {
        case 0: return 0x00; case 1: return 0x38; case 2: return 0x40; case 3: return 0x44; case 4: return 0x48;
        case -1: return 0xB8; case -2: return 0xC0; case -3: return 0xC4; case -4: return 0xC8;
    }
    return 0;
}
int main() {
    unsigned char hA[16 * 128], hB[128 * 16];
    int iA[16 * 128], iB[128 * 16];
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; hA[i] = enc(iA[i]); }
    for (int i = 0; i < 128 * 16; ++i) { iB[i] = rand() % 9 - 4; hB[i] = enc(iB[i]); }
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int scale : {0x7f7f7f7f, 0x7f, 0, (int)0x80808080}) {
        probe<<<1, 64>>>(dA, dB, dC, scale);
        float hC[256];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        int bad = 0; double ratio = 0; int nr = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            int ref = 0;
            for (int k = 0; k < 128; ++k) ref += iA[m * 128 + k] * iB[k * 16 + n];
            if (hC[m * 16 + n] != (float)ref) ++bad;
            if (ref != 0) { ratio += hC[m * 16 + n] / ref; ++nr; }
        }
        printf("scale 0x%08x: %d of 256 mismatches, mean C/ref %.6g, C[0][0..3] = %g %g %g %g\n", scale, bad, ratio / nr, hC[0], hC[1], hC[2], hC[3]);
    }
    return 0;
}
