// Micro-benchmark: accuracy of split-precision MFMA products against an fp64 reference (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 split_mfma_accuracy.hip -o split_mfma_accuracy && ./split_mfma_accuracy
// C[16][16] = A[16][K] * B[16][K]^T with
//   f32     v_mfma_f32_16x16x4_f32                      (what csrc/lstm.hip used in round 1)
//   h3      fp16 hi/lo split, 3 products (hh + hl + lh), lo kept scaled by 2^11, two accumulators
//   h3u     fp16 hi/lo split, 3 products, lo unscaled, ONE accumulator (operands pre-scaled by a power of two)
//   b3      bf16 hi/lo split, 3 products
//   b6      bf16 hi/mid/lo split, 6 products
//   h1/b1   plain fp16 / bf16 operands (1 product)
// Reported: max and rms of |C - C64| / (|A| |B|^T) (error relative to the magnitude of the dot product's terms).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

__device__ inline float bf16_round(float x, __bf16* out) {
    const __bf16 b = (__bf16)x;
    *out = b;
    return (float)b;
}

// mode: 0 f32, 1 h3, 2 h3u, 3 b3, 4 b6, 5 h1, 6 b1
__global__ void kern(const float* A, const float* B, float* C, int K, int mode, float sa, float sb) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + g], B[r * K + k + g], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            float a[8], b[8];
            for (int j = 0; j < 8; ++j) {
                a[j] = A[r * K + k + g * 8 + j] * sa;
                b[j] = B[r * K + k + g * 8 + j] * sb;
            }
            if (mode == 1 || mode == 2 || mode == 5) {
                h16x8 ah, al, bh, bl;
                const float ls = mode == 1 ? 2048.f : 1.f;
                for (int j = 0; j < 8; ++j) {
                    ah[j] = (_Float16)a[j];
                    al[j] = (_Float16)((a[j] - (float)ah[j]) * ls);
                    bh[j] = (_Float16)b[j];
                    bl[j] = (_Float16)((b[j] - (float)bh[j]) * ls);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
                if (mode == 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc1, 0, 0, 0);
                } else if (mode == 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
                }
            } else {
                b16x8 ah, am, al, bh, bm, bl;
                for (int j = 0; j < 8; ++j) {
                    __bf16 t;
                    float ra = a[j] - bf16_round(a[j], &t);
                    ah[j] = t;
                    ra -= bf16_round(ra, &t);
                    am[j] = t;
                    bf16_round(ra, &t);
                    al[j] = t;
                    float rb = b[j] - bf16_round(b[j], &t);
                    bh[j] = t;
                    rb -= bf16_round(rb, &t);
                    bm[j] = t;
                    bf16_round(rb, &t);
                    bl[j] = t;
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                if (mode == 3 || mode == 4) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc1, 0, 0, 0);
                }
                if (mode == 4) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc1, 0, 0, 0);
                }
            }
        }
    }
    const float inv = 1.f / (sa * sb);
    for (int q = 0; q < 4; ++q) {
        float v = acc[q];
        if (mode == 1) v += acc1[q] * (1.f / 2048.f);
        if (mode == 3 || mode == 4) v += acc1[q];
        C[(g * 4 + q) * 16 + r] = v * inv;       // C layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    }
}

static double urand() { return rand() / (double)RAND_MAX * 2. - 1.; }

int main() {
    const char* names[] = {"f32", "h3 (lo x 2^11, 2 acc)", "h3u (1 acc)", "b3", "b6", "h1", "b1"};
    for (int K : {608, 2400, 8096}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A(16 * K), B(16 * K);
            srand(1 + dist);
            for (auto& v : A) v = (float)(dist == 2 ? urand() * exp(8 * urand()) * 1e-3 : urand());      // dist 2: wide dynamic range (gradients)
            for (auto& v : B) v = (float)(0.05 * urand() * (dist == 1 ? 20. : 1.));
            double amax = 0, bmax = 0;
            for (auto v : A) amax = fmax(amax, fabs(v));
            for (auto v : B) bmax = fmax(bmax, fabs(v));
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4);
            hipMalloc(&dB, B.size() * 4);
            hipMalloc(&dC, 256 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("K=%d dist=%d amax=%.3g bmax=%.3g\n", K, dist, amax, bmax);
            for (int mode = 0; mode < 7; ++mode) {
                // power-of-two prescale for the fp16 paths: largest magnitude -> [2^13, 2^14)
                float sa = 1.f, sb = 1.f;
                if (mode == 1 || mode == 2 || mode == 5) {
                    sa = exp2f(13.f - floorf(log2f((float)amax)));
                    sb = exp2f(13.f - floorf(log2f((float)bmax)));
                }
                kern<<<1, 64>>>(dA, dB, dC, K, mode, sa, sb);
                float C[256];
                hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
                double emax = 0, e2 = 0;
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) {
                        double ref = 0, mag = 0;
                        for (int k = 0; k < K; ++k) {
                            ref += (double)A[i * K + k] * B[j * K + k];
                            mag += fabs((double)A[i * K + k] * B[j * K + k]);
                        }
                        const double e = fabs(C[i * 16 + j] - ref) / mag;
                        emax = fmax(emax, e);
                        e2 += e * e;
                    }
                printf("  %-24s max %.3e  rms %.3e\n", names[mode], emax, sqrt(e2 / 256));
            }
            hipFree(dA);
            hipFree(dB);
            hipFree(dC);
        }
    }
    return 0;
}
