// In-register FFT codelets for gfx950 (wave64).  Everything is fully unrolled so that every
// array index and every twiddle is a compile-time constant and the arrays live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace ptmi {

struct cpx {
    float x, y;
};
__device__ __forceinline__ cpx mk(float a, float b) { return cpx{a, b}; }
__device__ __forceinline__ cpx operator+(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx operator-(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
    return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) {  // a * conj(b)
    return cpx{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
__device__ __forceinline__ cpx cconj(cpx a) { return cpx{a.x, -a.y}; }

// cos/sin(2*pi*j/32), j = 0..15 (covers every radix <= 32: W_R^j = W_32^(j*32/R)).
__device__ constexpr float kCos32[16] = {
    1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
    0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f,
    0.0f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f,
    -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f};
__device__ constexpr float kSin32[16] = {
    0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f,
    0.70710678118654746f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f,
    1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f,
    0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f};

template <int R>
__host__ __device__ constexpr int bitrev(int v) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

// d * W_32^idx (forward, W = exp(-2*pi*i/32)) or d * conj(W_32^idx) (inverse); 0 <= idx < 16.
// idx is a compile-time constant after unrolling, so the branches fold away.
template <bool INV>
__device__ __forceinline__ cpx twiddle_mul32(cpx d, int idx) {
    if (idx == 0) return d;
    if (idx == 8) return INV ? cpx{-d.y, d.x} : cpx{d.y, -d.x};  // * (-i) fwd, * (+i) inv
    const float c = kCos32[idx];
    const float s = INV ? kSin32[idx] : -kSin32[idx];
    return cpx{d.x * c - d.y * s, d.x * s + d.y * c};
}

// Radix-2 decimation-in-frequency FFT of R points held in registers.
// On return a[bitrev<R>(k)] holds bin k.  Unnormalised in both directions.
template <int R, bool INV>
__device__ __forceinline__ void fft_dif(cpx (&a)[R]) {
#pragma unroll
    for (int half = R / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int g = 0; g < R; g += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const cpx u = a[g + j];
                const cpx v = a[g + j + half];
                a[g + j] = u + v;
                a[g + j + half] = twiddle_mul32<INV>(u - v, j * (16 / half));  // W_{2*half}^j
            }
        }
    }
}

}  // namespace ptmi
