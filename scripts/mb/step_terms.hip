// Micro-benchmark: the TERMS of one time step of the persistent recurrence kernels (csrc/lstm_split.hip), each alone, with NO dependency
// between workgroups - the speed-of-light of one workgroup's own serial work per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 step_terms.hip -o step_terms
//   ./step_terms            -> a table: ns per step and per term, forward (39 KB gather, 27 MFMAs / wavefront of 4) and backward shape
//                              (154 KB gather, 30 MFMAs / wavefront of 8), 152 workgroups of 512 threads like the backward kernel
// Terms:
//   gather   every wavefront requests its k blocks of the chain's row tile (1 KB per wavefront load: 16 rows x 32 k x 2 B, hi and lo
//            plane) with sc1 buffer loads from a tile ALL workgroups read (the previous step's hand-off, resident in L2 / MALL) and waits
//            for them: the CU's vector-memory path (64 B per clock)
//   mfma     the wavefront's MFMA chain on registers: blocks x 3 products of v_mfma_f32_16x16x32_bf16 into 3 accumulators
//   reduce   accumulators -> LDS, workgroup barrier, NW LDS reads per owner thread, the gate arithmetic's transcendentals
//   store    the step's hand-off stores, write-through (4 x 4-byte stores per owner thread), NOT waited for
//   all      gather + mfma (overlapped as the kernels do: next pass requested before the current one is multiplied) + reduce + store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const float* tile;        // [T][planes 2][KB][1 KB] floats viewed as bytes: the row tile every workgroup gathers
    float* out;               // [grid][512] sink
    unsigned* hand;           // [T][grid][512 * 4] hand-off words
    unsigned long long* ticks;     // [grid]: s_memrealtime ticks of the loop (100 MHz)
    int T, KB, mode, nw;      // KB: k blocks of the tile (75 at H = 600); nw: wavefronts that take part in the product
    int ring;                 // tiles the gather walks through in turn (8: they stay in the XCD's L2; 256: every step's tile comes over the fabric)
    int aux;                  // cache-policy bits of the gather loads (16 = sc1 as in the kernels, 0 = plain)
};

template <int CB>
__global__ __launch_bounds__(512) void terms_kernel(const Args A) {
    __shared__ float red[2][8][16][20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (A.KB + A.nw - 1) / A.nw;
    const int kb0 = min(wave * per, A.KB - 1);
    const int nb = wave < A.nw ? min(per, A.KB - wave * per) : 0;
    f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    u32x4 bh[CB], bl[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        bh[i] = u32x4{0x3f803f80u + (unsigned)(lane + i), 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        bl[i] = u32x4{0x3b003b00u, 0x3b003b00u + (unsigned)i, 0x3b003b00u, 0x3b003b00u};
    }
    const bool do_gather = A.mode == 0 || A.mode == 4, do_mfma = A.mode == 1 || A.mode == 4, do_reduce = A.mode == 2 || A.mode == 4,
               do_store = A.mode == 3 || A.mode == 4;
    float state = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < A.T; ++s) {
        u32x4 fh[CB], fl[CB];
        if (do_gather) {
            const float* tbase = A.tile + (size_t)(s % A.ring) * A.KB * 512;       // `ring` tiles in turn (2 planes x KB x 1 KB each)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tbase), 0, A.KB * 2048, 0x00020000);
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const unsigned off = i < nb ? (unsigned)((kb0 + i) * 2048 + lane * 16) : 0x80000000u;
                if (A.aux) {
                    fh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                    fl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 1024, 16);
                } else {
                    fh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    fl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 1024, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                fh[i] = u32x4{0x3f803f80u, 0x3f803f80u ^ (unsigned)s, 0x3f803f80u, 0x3f803f80u};
                fl[i] = u32x4{0x3b003b00u, 0x3b003b00u, 0x3b003b00u ^ (unsigned)s, 0x3b003b00u};
            }
        }
        if (do_mfma || do_gather) {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                if (i < nb || !do_gather) {
                    if (do_mfma) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[i]), __builtin_bit_cast(bf16x8, bh[i]), acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fh[i]), __builtin_bit_cast(bf16x8, bl[i]), acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fh[i]), __builtin_bit_cast(bf16x8, bh[i]), acc[2], 0, 0, 0);
                    } else {
                        acc[0][0] += __builtin_bit_cast(f32x4, fh[i])[0] + __builtin_bit_cast(f32x4, fl[i])[3];      // consume the loads
                    }
                }
            }
        }
        if (do_reduce) {
            const int g4 = lane >> 4, r = lane & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) red[s & 1][wave][g4 * 4 + q][r] = (acc[0][q] + acc[1][q]) + acc[2][q];
            __syncthreads();
            if (tid < 256) {
                const int bl_ = (tid >> 4) & 15, jl = tid & 15;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sum += red[s & 1][w][bl_][jl];
                // the gate arithmetic of a backward step: 1 tanh + a handful of products (forward: 3 sigmoids + 2 tanh)
                const float tc = tanhf(sum * 1e-3f + state);
                state = state * 0.5f + tc * (1.f - tc * tc);
            }
        }
        if (do_store && tid < 256) {
            unsigned* tq = A.hand + ((size_t)(s & 7) * gridDim.x + blockIdx.x) * 2048 + tid;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __hip_atomic_store(tq + g * 512, __float_as_uint(state) + (unsigned)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (A.mode == 5) {           // the cost of reading the 100 MHz clock itself (DafHold::mark / wait of the kernels): A.T dependent reads
        unsigned long long x = t1;
        for (int s = 0; s < A.T; ++s) {
            const unsigned long long y = __builtin_amdgcn_s_memrealtime();
            x = y > x ? y : x + 1;
            asm volatile("" : "+s"(x));
        }
        state += (float)(x & 7);
        t1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) A.ticks[blockIdx.x] = t1 - x + (x - t0 >= 0 ? 0 : 0);
    }
    A.out[(size_t)blockIdx.x * 512 + tid] = state + acc[0][0] + acc[1][1] + acc[2][2];
    if (tid == 0 && A.mode != 5) A.ticks[blockIdx.x] = t1 - t0;
    if (tid == 0 && A.mode == 5) A.ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int T = 2000, grid = argc > 1 ? atoi(argv[1]) : 152;
    const int ring = argc > 2 ? atoi(argv[2]) : 8, aux = argc > 3 ? atoi(argv[3]) : 16;
    const int KBmax = 75;
    float* tile;
    float* out;
    unsigned* hand;
    unsigned long long* ticks;
    hipMalloc(&tile, (size_t)ring * KBmax * 2048);
    hipMemset(tile, 0x3b, (size_t)ring * KBmax * 2048);
    hipMalloc(&out, (size_t)grid * 512 * 4);
    hipMalloc(&hand, (size_t)8 * grid * 2048 * 4);
    hipMalloc(&ticks, grid * 8);
    const char* names[6] = {"gather", "mfma", "reduce", "store", "all", "clock"};
    struct Shape { const char* what; int KB, nw, cb; } shapes[] = {
        {"forward  (H = 600: 19 k blocks of h, 2 planes = 39 KB; 4 of 8 wavefronts multiply, 5 blocks x 3 products)", 19, 4, 5},
        {"backward (4H = 2400: 75 k blocks of dgates, 2 planes = 154 KB; 8 wavefronts, 10 blocks x 3 products)", 75, 8, 10},
    };
    printf("# %d workgroups x 512 threads, %d steps, no dependency between workgroups; gather walks %d tiles, loads %s; ns per step = median over the workgroups (max)\n", grid, T, ring, aux ? "sc1" : "plain");
    for (const Shape& sh : shapes) {
        printf("%s\n", sh.what);
        for (int mode = 0; mode < 6; ++mode) {
            if (ring != 8 && mode != 0 && mode != 4) continue;
            Args A{tile, out, hand, ticks, T, sh.KB, mode, sh.nw, ring, aux};
            for (int rep = 0; rep < 2; ++rep) {
                if (sh.cb == 5) hipLaunchKernelGGL(terms_kernel<5>, dim3(grid), dim3(512), 0, 0, A);
                else hipLaunchKernelGGL(terms_kernel<10>, dim3(grid), dim3(512), 0, 0, A);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> srt(h);
            std::sort(srt.begin(), srt.end());
            printf("  %-7s %7.0f ns per step (slowest workgroup %7.0f)\n", names[mode], srt[grid / 2] * 10.0 / T, srt[grid - 1] * 10.0 / T);
        }
    }
    return 0;
}
