// In-register FFT codelets for gfx950 (wave64).  Everything is fully unrolled so that every
// array index and every twiddle is a compile-time constant and the arrays live in VGPRs.
//
// The STFT kernels are VALU-bound (a wave64 VALU op takes 4 cycles on the 16-lane SIMD), so the
// complex arithmetic is written for the PACKED fp32 pipe: a complex number is one even-aligned
// VGPR pair and add / sub / multiply-by-(+-i) / complex multiply are one or two v_pk_*_f32 each,
// with the (re, im) swizzles and sign flips carried by the op_sel / neg_lo / neg_hi modifiers
// instead of extra moves.
#pragma once
#include <hip/hip_runtime.h>

namespace ptmi {

typedef float cpx __attribute__((ext_vector_type(2)));   // .x = re, .y = im

__device__ __forceinline__ cpx mk(float a, float b) { return cpx{a, b}; }
__device__ __forceinline__ cpx cconj(cpx a) { return cpx{a.x, -a.y}; }

// a * w,  w in VGPRs
__device__ __forceinline__ cpx cmul(cpx a, cpx w) {
    cpx t, r;   // t = (-a.y w.y, a.y w.x);  r = (a.x w.x, a.x w.y) + t
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w),  w in VGPRs
__device__ __forceinline__ cpx cmulc(cpx a, cpx w) {
    cpx t, r;   // t = (a.y w.y, a.y w.x);  r = (a.x w.x, -a.x w.y) + t
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * w with a compile-time constant w (lives in an SGPR pair)
__device__ __forceinline__ cpx cmul_k(cpx a, cpx w) {
    cpx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// (u - v) * (-i) = (u.y - v.y, v.x - u.x)
__device__ __forceinline__ cpx sub_mul_mi(cpx u, cpx v) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(u), "v"(v));
    return r;
}
// (u - v) * (+i) = (v.y - u.y, u.x - v.x)
__device__ __forceinline__ cpx sub_mul_pi(cpx u, cpx v) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(v));
    return r;
}
// a + conj(b),  a - conj(b),  a + i b,  a - i b
__device__ __forceinline__ cpx add_conj(cpx a, cpx b) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cpx sub_conj(cpx a, cpx b) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cpx add_i(cpx a, cpx b) {   // (a.x - b.y, a.y + b.x)
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cpx sub_i(cpx a, cpx b) {   // (a.x + b.y, a.y - b.x)
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

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

// (u - v) * W_32^idx (forward, W = exp(-2*pi*i/32)) or (u - v) * conj(W_32^idx) (inverse);
// 0 <= idx < 16 is a compile-time constant after unrolling, so the branches fold away.
template <bool INV>
__device__ __forceinline__ cpx sub_twiddle32(cpx u, cpx v, int idx) {
    if (idx == 0) return u - v;
    if (idx == 8) return INV ? sub_mul_pi(u, v) : sub_mul_mi(u, v);
    const cpx w = cpx{kCos32[idx], INV ? kSin32[idx] : -kSin32[idx]};
    return cmul_k(u - v, w);
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
                a[g + j + half] = sub_twiddle32<INV>(u, v, j * (16 / half));  // W_{2*half}^j
            }
        }
    }
}

}  // namespace ptmi
