// Micro-benchmark: per-step cost of the recurrence hand-off (all-gather of a 16 x KP fp32 row tile among the P
// workgroups of a chain, once per time step) for three placements / protocols (gfx950, 8 XCDs with private L2s).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 xcd_chain_handoff.hip -o xcd_chain_handoff
//   ./xcd_chain_handoff <mode> <nchains> <P> <JT> <grid> <lds_kb> <T> <work> <check>
// mode 0  chain members spread over all XCDs (consecutive block ids), payload + flag stored write-through (sc1),
//         read with sc1 loads: the round-1 protocol of csrc/lstm.hip
// mode 1  chain c = the workgroups that FIND THEMSELVES on XCD c (HW_REG_XCC_ID + a ticket per XCD), payload and flag
//         stored PLAIN (they stay in that XCD's L2), read with sc1 loads (bypass the CU's L1, served by the shared L2)
// mode 2  placement of mode 1, stores of mode 0
// mode 3  mode 1 with PLAIN payload loads (every address is written once and first read after its flag: no stale copy)
// mode 4  reduce-scatter (backward form): every workgroup writes P slices of 16 x JT partial sums (one per consumer),
//         reads its slice from all P producers; placement and flavours of mode 3
// mode 5  mode 4 with write-through stores and sc1 loads, chain members spread over all XCDs
// Every consumer checks every word it reads against the value the producer must have written (stale data = error).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    float* hbuf;          // [T][nchains][KB][16 rows][16]
    unsigned* flags;      // [nchains][64]
    unsigned* tickets;    // [8]
    unsigned* err;        // [4]: timeouts, mismatches, -, -
    unsigned* info;       // [grid]: xcc | ticket << 8 | role << 20
    int T, P, JT, H, KB, mode, nchains, work, check, phases;
    unsigned long long* phase_out;
};

__device__ __forceinline__ float val(int t, int chain, int row, int unit) {
    return (float)((t * 131 + chain * 17 + row * 29 + unit * 3) % 4093) + 0.25f;
}

extern "C" __global__ __launch_bounds__(256) void chain_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sh = reinterpret_cast<unsigned*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
        const unsigned ticket = atomicAdd(A.tickets + xcc, 1u);
        sh[0] = ticket;
    }
    __syncthreads();
    const unsigned ticket = sh[0];
    int chain, idx;
    if (A.mode == 0 || A.mode == 5) {
        chain = blockIdx.x / A.P;
        idx = blockIdx.x % A.P;
    } else {
        chain = (int)xcc;
        idx = (int)ticket;
    }
    const bool live = chain < A.nchains && idx < A.P;
    if (tid == 0) A.info[blockIdx.x] = xcc | (ticket << 8) | ((live ? 1u : 0u) << 20);
    if (!live) return;
    const bool wt = A.mode == 0 || A.mode == 2 || A.mode == 5;      // write-through stores
    const bool plain_ld = A.mode == 3 || A.mode == 4;
    const bool rs = A.mode >= 4;
    unsigned* const myflags = A.flags + chain * 64;
    const size_t tile_elems = A.mode >= 4 ? (size_t)A.P * A.P * 16 * A.JT : (size_t)A.KB * 256;
    const int g4 = lane >> 4, r = lane & 15;
    const int per = (A.KB + 3) / 4, kb0 = wave * per, kb1 = min(A.KB, kb0 + per);
    float sink = 0.f;
    unsigned bad = 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    const bool prof = A.phases && tid == 0 && chain == 0 && idx == (A.P > 1 ? 1 : 0);
    auto mark = [&](int k) {
        if (prof) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            ph[k] += now - last;
            last = now;
        }
    };
    if (prof) last = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < A.T; ++s) {
        mark(7);
        if (s > 0) {
            if (wave == 0) {
                unsigned it = 0;
                for (;; ++it) {
                    const unsigned v = lane < A.P ? __hip_atomic_load(myflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
                    if (__all(v >= (unsigned)s)) break;
                    if (it > (1u << 22)) {
                        if (lane == 0) atomicAdd(A.err, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            mark(0);
            __syncthreads();
            mark(1);
            const float* tbase = A.hbuf + ((size_t)(s - 1) * A.nchains + chain) * tile_elems;
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tbase), 0, (int)(tile_elems * 4), 0x00020000);
            f32x4 a[10];
            if (!rs) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int kb = min(kb0 + i, kb1 - 1);
                    const unsigned off = (unsigned)(kb * 1024 + r * 64 + g4 * 16);
                    a[i] = plain_ld ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 0))
                                    : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 16 /* sc1 */));
                }
            } else {
                // slab layout [producer][consumer][16 rows][JT] floats; this workgroup = consumer idx: its slice of producer p is
                // 16 * JT contiguous floats; 256 threads x 10 float4 cover P * 16 * JT floats (host: P * 16 * JT <= 10240)
                const int slice = 16 * A.JT;             // floats
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int e4 = (i * 256 + tid);      // float4 index over [P][slice / 4]
                    const int p = e4 / (slice / 4), w4 = e4 - p * (slice / 4);
                    const unsigned off = p < A.P ? (unsigned)((((size_t)p * A.P + idx) * slice + 4 * w4) * 4) : 0x80000000u;
                    a[i] = plain_ld ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 0))
                                    : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 16));
                }
            }
            if (A.phases) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            mark(2);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int kb = min(kb0 + i, kb1 - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sink += a[i][q];
                    if (A.check) {
                        if (!rs) {
                            const int unit = kb * 16 + g4 * 4 + q;
                            if (unit < A.H && a[i][q] != val(s - 1, chain, r, unit)) ++bad;
                        } else {
                            const int slice = 16 * A.JT, e4 = i * 256 + tid, p = e4 / (slice / 4), w4 = e4 - p * (slice / 4);
                            if (p < A.P && a[i][q] != val(s - 1, chain, p, idx * slice + 4 * w4 + q)) ++bad;
                        }
                    }
                }
            }
            for (int w = 0; w < A.work; ++w) __builtin_amdgcn_s_sleep(1);      // stand-in for the step's MFMA work (64 cycles each)
            mark(3);
            __syncthreads();
            mark(4);
        }
        // produce
        float* tq = A.hbuf + ((size_t)s * A.nchains + chain) * tile_elems;
        if (!rs) {          // this workgroup's JT units of all 16 rows
            for (int e = tid; e < 16 * A.JT; e += 256) {
                const int row = e / A.JT, u = idx * A.JT + (e - row * A.JT);
                if (u < A.H) {
                    float* p = tq + (u >> 4) * 256 + row * 16 + (u & 15);
                    const float v = val(s, chain, row, u) + (sink == 12345.678f ? 1.f : 0.f);
                    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *p = v;
                }
            }
        } else {            // partial sums for every consumer: P * 16 * JT floats, 16-byte stores
            const int slice = 16 * A.JT, total4 = A.P * slice / 4;
            const __amdgpu_buffer_rsrc_t ws_ = __builtin_amdgcn_make_buffer_rsrc(tq, 0, (int)(tile_elems * 4), 0x00020000);
            for (int e4 = tid; e4 < total4; e4 += 256) {
                const int c = e4 / (slice / 4), w4 = e4 - c * (slice / 4);
                f32x4 v;
                for (int q = 0; q < 4; ++q) v[q] = val(s, chain, idx, c * slice + 4 * w4 + q) + (sink == 12345.678f ? 1.f : 0.f);
                const unsigned off = (unsigned)((((size_t)idx * A.P + c) * slice + 4 * w4) * 4);
                if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ws_, off, 0, 16); else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ws_, off, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(5);
        __syncthreads();
        mark(6);
        if (tid == 0) {
            if (wt) __hip_atomic_store(myflags + idx, (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *(volatile unsigned*)(myflags + idx) = (unsigned)s + 1u;
        }
    }
    if (bad) atomicAdd(A.err + 1, bad);
    if (prof) for (int k = 0; k < 8; ++k) A.phase_out[k] = ph[k];
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
    if (argc < 10) {
        printf("usage: mode nchains P JT grid lds_kb T work check\n");
        return 2;
    }
    Args A{};
    A.mode = atoi(argv[1]);
    A.nchains = atoi(argv[2]);
    A.P = atoi(argv[3]);
    A.JT = atoi(argv[4]);
    int grid = atoi(argv[5]);
    const int lds = atoi(argv[6]) * 1024;
    A.T = atoi(argv[7]);
    A.work = atoi(argv[8]);
    A.check = atoi(argv[9]);
    A.phases = argc > 10 ? atoi(argv[10]) : 0;
    A.H = 600;
    A.KB = 38;
    if (A.mode == 0 || A.mode == 5) grid = A.nchains * A.P;
    const size_t hb = (size_t)A.T * A.nchains * (A.mode >= 4 ? (size_t)A.P * A.P * 16 * A.JT : (size_t)A.KB * 256) * 4;
    CK(hipMalloc(&A.hbuf, hb));
    CK(hipMalloc(&A.flags, 8 * 64 * 4));
    CK(hipMalloc(&A.tickets, 8 * 4));
    CK(hipMalloc(&A.err, 16));
    CK(hipMalloc(&A.info, grid * 4));
    CK(hipMalloc(&A.phase_out, 64));
    CK(hipMemset(A.phase_out, 0, 64));
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(A.hbuf, 0xff, hb));          // poison: stale reads show as mismatches
        CK(hipMemset(A.flags, 0, 8 * 64 * 4));
        CK(hipMemset(A.tickets, 0, 32));
        CK(hipMemset(A.err, 0, 16));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), lds, 0, A);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err[4];
        CK(hipMemcpy(err, A.err, 16, hipMemcpyDeviceToHost));
        std::vector<unsigned> info(grid);
        CK(hipMemcpy(info.data(), A.info, grid * 4, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0}, live = 0, mism = 0;
        for (int b = 0; b < grid; ++b) {
            per_xcc[info[b] & 15]++;
            live += (info[b] >> 20) & 1;
            mism += ((info[b] & 15) != (unsigned)(b % 8));
        }
        printf("mode %d chains %d P %d JT %d grid %d lds %d work %d check %d: %.3f us/step  timeouts %u mismatches %u live %d  wg/xcc",
               A.mode, A.nchains, A.P, A.JT, grid, lds, A.work, A.check, ms * 1e3 / A.T, err[0], err[1], live);
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("  (b%%8 != xcc: %d)\n", mism);
        if (A.phases) {
            unsigned long long ph[8];
            CK(hipMemcpy(ph, A.phase_out, 64, hipMemcpyDeviceToHost));
            const char* nm[8] = {"poll", "barrier", "loads", "check+work", "barrier", "stores+drain", "barrier", "flag+loop"};
            printf("   ns/step:");
            for (int k = 0; k < 8; ++k) printf(" %s %.0f", nm[k], ph[k] * 10.0 / A.T);
            printf("\n");
        }
    }
    return 0;
}
