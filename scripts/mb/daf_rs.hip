// Micro-benchmark (round 5): the backward recurrence's per-step hand-off as it is - ALL-GATHER of the chain's
// dgates planes, 154 KB per workgroup and step - against the REDUCE-SCATTER form - every workgroup multiplies its own 64 gate
// columns by its [64 x 600] slice of W_hh, writes a 16 x 16 fp32 partial per consumer (38 x 1 KB, 16-byte write-through stores) and
// sums the 38 partials of its own 16 units (38 KB in) -, both under the data-as-flag protocol of csrc/lstm_split.hip (planes pre-filled
// with a pattern no value has, consumers re-request until none of their words is the pattern, first request held back `hold` ticks of
// the 100 MHz clock after the workgroup's barrier), with the step's real MFMA count, LDS traffic and barrier, on the kernels'
// placement (4 chains x 38 workgroups of 512 threads, a chain on two neighbouring XCDs).  Every word a consumer uses is checked
// against what its producer must have written (check = 1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 daf_rs.hip -o daf_rs
//   ./daf_rs <mode 0 = all-gather | 1 = reduce-scatter> <hold ticks> <T> <check> [span = 2] [P = 38] [plain stores = 0]
// span 1: a chain on ONE XCD (P <= 32); plain stores 1: the hand-off stores stay in that XCD's L2 (only valid with span 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

struct Args {
    unsigned* buf;       // hand-off scratch, pre-filled with 0xFFFFFFFF
    unsigned* err;       // [4]: timeouts, mismatches
    float* sink;
    int T, P, span, mode, check, plain;
    unsigned hold;
    unsigned max_polls;
    unsigned long long* phase_out;
};

constexpr int KB = 75;            // 32-wide k blocks of the all-gather operand (4H = 2400 columns)
constexpr unsigned kFill = 0xffffffffu;

__device__ __forceinline__ unsigned ag_val(int t, int chain, unsigned gd) {
    return ((unsigned)(t * 131 + chain * 17) + gd * 3u) % 65000u | ((gd * 7u + (unsigned)t) % 65000u) << 16;
}
__device__ __forceinline__ float rs_val(int t, int chain, int c, int p, int idx) {
    return (float)((t * 131 + chain * 17 + c * 29 + p * 7 + idx * 3) % 4093) + 0.25f;
}
__device__ __forceinline__ unsigned fold_max16(const uint4 v, unsigned m) {
    us2 x = __builtin_elementwise_max(__builtin_bit_cast(us2, v.x), __builtin_bit_cast(us2, v.y));
    x = __builtin_elementwise_max(x, __builtin_bit_cast(us2, v.z));
    x = __builtin_elementwise_max(x, __builtin_bit_cast(us2, v.w));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, __builtin_bit_cast(us2, m)));
}
__device__ __forceinline__ bool has_fill16(unsigned m) { return (m & 0xffffu) == 0xffffu || (m >> 16) == 0xffffu; }
__device__ __forceinline__ int handoff_index(int col, int plane, int row) {
    return (((col >> 5) * 2 + plane) * 64 + ((col & 31) >> 3) * 16 + row) * 8 + (col & 7);
}
__device__ __forceinline__ f32x4 mma(const uint4 a, const uint4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void step_kernel(const Args A) {
    const int L = blockIdx.x, xcd = L & 7;
    const int chain = xcd / A.span;
    const int idx = (L >> 3) * A.span + (xcd - chain * A.span);          // this workgroup's place in its chain
    if (idx >= A.P || chain >= 4) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = A.P;
    __shared__ float red[2][8][16][20];
    __shared__ uint4 afrag[2][4][64];        // reduce-scatter: this step's own 16 x 64 gate gradients as MFMA A fragments (2 k blocks x (hi, lo))
    unsigned long long ref = __builtin_amdgcn_s_memrealtime();
    auto hold_wait = [&]() { while ((unsigned)(__builtin_amdgcn_s_memrealtime() - ref) < A.hold) __builtin_amdgcn_s_sleep(1); };
    unsigned bad = 0, timeouts = 0;
    float sinkv = 0.f;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // resident "weights": 80 registers per lane in either form
    uint4 w[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) w[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u);
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, last = 0;
    const bool prof = A.phase_out && tid == 0 && chain == 0 && idx == 1;
    auto mark = [&](int k) {
        if (prof) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            ph[k] += now - last;
            last = now;
        }
    };
    if (prof) last = __builtin_amdgcn_s_memrealtime();

    if (A.mode == 0) {
        // ---------------------------------------------------------------- all-gather (the kernel as it is)
        const size_t blk = (size_t)KB * 2 * 256;                        // dwords per (time, chain)
        const int base = KB / 8, extra = KB - base * 8;
        const int kb0 = wave * base + min(wave, extra), nbw = base + (wave < extra ? 1 : 0);
        const int bl_ = (tid >> 4) & 15, jl = tid & 15;
        for (int s = 0; s < A.T; ++s) {
            float dh = 0.f;
            if (s > 0) {
                mark(0);
                const unsigned* tb = A.buf + ((size_t)(s - 1) * 4 + chain) * blk;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(tb), 0, (int)(blk * 4), 0x00020000);
                uint4 a[20];
                hold_wait();
                mark(1);
                unsigned polls = 0;
                for (;;) {
#pragma unroll
                    for (int f = 0; f < 20; ++f) {
                        const int i = min(f >> 1, nbw - 1);
                        a[f] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((kb0 + i) * 2 + (f & 1)) * 1024 + lane * 16), 0, 16));
                    }
                    unsigned m = 0u;
#pragma unroll
                    for (int f = 0; f < 20; ++f) m = fold_max16(a[f], m);
                    if (!__any(has_fill16(m))) break;
                    if (++polls >= A.max_polls) {
                        ++timeouts;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                mark(2);
                if (A.check) {
#pragma unroll
                    for (int f = 0; f < 20; ++f) {
                        const int i = min(f >> 1, nbw - 1);
                        const unsigned gd0 = (unsigned)(((kb0 + i) * 2 + (f & 1)) * 256 + lane * 4);
                        const unsigned v[4] = {a[f].x, a[f].y, a[f].z, a[f].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) bad += v[q] != ag_val(s - 1, chain, gd0 + q);
                    }
                }
                f32x4 acc[3] = {zero, zero, zero};
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    acc[0] = mma(a[2 * i + 1], w[2 * i], acc[0]);
                    acc[1] = mma(a[2 * i], w[2 * i + 1], acc[1]);
                    acc[2] = mma(a[2 * i], w[2 * i], acc[2]);
                }
                const int g4 = lane >> 4, r = lane & 15;
#pragma unroll
                for (int q = 0; q < 4; ++q) red[s & 1][wave][g4 * 4 + q][r] = (acc[0][q] + acc[1][q]) + acc[2][q];
                mark(3);
                __syncthreads();
                ref = __builtin_amdgcn_s_memrealtime();
                mark(4);
                if (tid < 256) {
                    float part[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) part[k] = red[s & 1][k][bl_][jl];
#pragma unroll
                    for (int k = 0; k < 8; ++k) dh += part[k];
                }
            }
            sinkv += dh;
            // gate arithmetic stand-in (~40 VALU operations, as the kernel's), then the four 4-byte hand-off stores of an owner
            float gsum = dh;
#pragma unroll
            for (int k = 0; k < 12; ++k) gsum = __builtin_fmaf(gsum, 0.999f, 1e-3f * (float)k);
            sinkv += gsum * 1e-30f;
            if (tid < 256) {
                unsigned* tq = A.buf + ((size_t)s * 4 + chain) * blk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = g * 600 + idx * 16 + jl;
                    if (idx * 16 + jl < 600) {
                        const unsigned gd = (unsigned)(handoff_index(col & ~1, lane & 1, bl_) / 2);
                        const unsigned v = ag_val(s, chain, gd) + (sinkv == 12345.f ? 1u : 0u);
                        if (A.plain) tq[gd] = v;
                        else __hip_atomic_store(tq + gd, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            mark(5);
        }
    } else {
        // ---------------------------------------------------------------- reduce-scatter
        const size_t blk = (size_t)P * P * 256;                         // dwords per (time, chain): [consumer][producer][256]
        const int e = lane >> 1, h = lane & 1;
        const int mine = wave * 32 + e;                                 // float index of this lane pair's element in a partial tile
        for (int s = 0; s < A.T; ++s) {
            float dh = 0.f;
            if (s > 0) {
                mark(0);
                const unsigned* tb = A.buf + ((size_t)(s - 1) * 4 + chain) * blk + (size_t)idx * P * 256;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(tb), 0, P * 1024, 0x00020000);
                unsigned v[19];
                hold_wait();
                mark(1);
                unsigned polls = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < 19; ++i) {
                        const int p = 2 * i + h;
                        v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, p < P ? (unsigned)(p * 1024 + mine * 4) : 0x80000000u, 0, 16);
                    }
                    unsigned m = 0u;
#pragma unroll
                    for (int i = 0; i < 19; ++i) m = max(m, v[i]);
                    if (!__any(m == kFill)) break;
                    if (++polls >= A.max_polls) {
                        ++timeouts;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                mark(2);
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 19; ++i) {
                    const int p = 2 * i + h;
                    const float f = __uint_as_float(v[i]);
                    if (A.check && p < P) bad += f != rs_val(s - 1, chain, idx, p, mine);
                    sum += p < P ? f : 0.f;
                }
                const float other = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(sum), 0xB1, 0xF, 0xF, true));
                dh = h ? other + sum : sum + other;          // producers in a fixed order: (even ones) + (odd ones)
            }
            sinkv += dh;
            float gsum = dh;
#pragma unroll
            for (int k = 0; k < 12; ++k) gsum = __builtin_fmaf(gsum, 0.999f, 1e-3f * (float)k);
            // this step's own gate gradients -> LDS as A fragments: every lane pair owns an element; lane h writes 2 of its 4 gates'
            // (hi, lo) halves: 4 x ds_write_b16-equivalent work as one 8-byte store each
            {
                unsigned long long* a8 = reinterpret_cast<unsigned long long*>(&afrag[s & 1][0][0]);
                a8[(h * 2 + 0) * 128 + (mine >> 1)] = (unsigned long long)__float_as_uint(gsum) * 0x100000001ull;
                a8[(h * 2 + 1) * 128 + (mine >> 1)] = (unsigned long long)__float_as_uint(gsum + 1.f) * 0x100000001ull;
            }
            mark(3);
            __syncthreads();
            ref = __builtin_amdgcn_s_memrealtime();
            mark(4);
            uint4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = afrag[s & 1][k][lane];
            unsigned* tq = A.buf + ((size_t)s * 4 + chain) * blk;
            const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(tq, 0, (int)(blk * 4), 0x00020000);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c = wave + 8 * k;                 // consumer = N tile of 16 hidden units
                f32x4 acc[3] = {zero, zero, zero};
                acc[0] = mma(a[1], w[4 * k], acc[0]);
                acc[1] = mma(a[0], w[4 * k + 1], acc[1]);
                acc[2] = mma(a[0], w[4 * k], acc[2]);
                acc[0] = mma(a[3], w[4 * k + 2], acc[0]);
                acc[1] = mma(a[2], w[4 * k + 3], acc[1]);
                acc[2] = mma(a[2], w[4 * k + 2], acc[2]);
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float real = (acc[0][q] + acc[1][q]) + acc[2][q];
                    o[q] = rs_val(s, chain, c, idx, lane * 4 + q) + (real == 12345.f ? 1.f : 0.f);
                }
                if (c < P) {
                    const unsigned off = (unsigned)((((size_t)c * P + idx) * 256 + lane * 4) * 4);
                    if (A.plain) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ws, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ws, off, 0, 16);
                }
            }
            mark(5);
        }
    }
    if (bad) atomicAdd(A.err + 1, bad);
    if (timeouts) atomicAdd(A.err, timeouts);
    if (sinkv == 3.14159f) A.sink[0] = sinkv;
    if (prof) for (int k = 0; k < 6; ++k) A.phase_out[k] = ph[k];
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 5) {
        printf("usage: mode hold T check [span] [P] [plain] [phases]\n");
        return 2;
    }
    Args A{};
    A.mode = atoi(argv[1]);
    A.hold = (unsigned)atoi(argv[2]);
    A.T = atoi(argv[3]);
    A.check = atoi(argv[4]);
    A.span = argc > 5 ? atoi(argv[5]) : 2;
    A.P = argc > 6 ? atoi(argv[6]) : 38;
    A.plain = argc > 7 ? atoi(argv[7]) : 0;
    const int phases = argc > 8 ? atoi(argv[8]) : 0;
    A.max_polls = 1u << 16;
    if (A.P > 38 || (A.span == 1 && A.P > 32)) {
        printf("P too large\n");
        return 2;
    }
    const size_t blk = A.mode == 0 ? (size_t)KB * 2 * 256 : (size_t)A.P * A.P * 256;
    const size_t bytes = (size_t)A.T * 4 * blk * 4;
    CK(hipMalloc(&A.buf, bytes));
    CK(hipMalloc(&A.err, 16));
    CK(hipMalloc(&A.sink, 16));
    CK(hipMalloc(&A.phase_out, 64));
    CK(hipMemset(A.phase_out, 0, 64));
    if (!phases) A.phase_out = nullptr;
    const int per_xcd = (A.P + A.span - 1) / A.span;
    const int grid = 8 * per_xcd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(A.buf, 0xff, bytes));
        CK(hipMemset(A.err, 0, 16));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(512), 0, 0, A);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err[4];
        CK(hipMemcpy(err, A.err, 16, hipMemcpyDeviceToHost));
        printf("mode %d (%s) hold %u span %d P %d plain %d T %d check %d: %.3f us/step  timeouts %u mismatches %u  scratch %.1f MB/step\n", A.mode,
               A.mode ? "reduce-scatter" : "all-gather", A.hold, A.span, A.P, A.plain, A.T, A.check, ms * 1e3 / A.T, err[0], err[1],
               4.0 * blk * 4 / 1e6);
        if (rep > 0 && ms < best) best = ms;
        if (phases && A.phase_out) {
            unsigned long long ph[8];
            CK(hipMemcpy(ph, A.phase_out, 48, hipMemcpyDeviceToHost));
            const char* nm[6] = {"loop head", "hold", "loads until clean", "check+mma+lds", "barrier", "reduce/mma+stores"};
            printf("   ns/step:");
            for (int k = 0; k < 6; ++k) printf(" %s %.0f;", nm[k], ph[k] * 10.0 / A.T);
            printf("\n");
        }
    }
    printf("BEST mode %d hold %u span %d P %d plain %d: %.3f us/step\n", A.mode, A.hold, A.span, A.P, A.plain, best * 1e3 / A.T);
    return 0;
}
