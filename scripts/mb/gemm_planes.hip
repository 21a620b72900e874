// Prototype: C[M, N] (fp32) = A B^T with BOTH operands already split into fp16 (hi, lo) planes in fragment-tile order
//   operand X [R rows][K]:  tiles [R / 16][K / 32][plane hi | lo][16 rows][32 k] of fp16  (1 KB per plane tile)
// (the layout of the recurrence kernels' hand-off copy), three MFMA products per fragment pair, LDS filled by
// global_load_lds (16 B per lane, no VALU on the way), one barrier per k-step, STAGES LDS buffers.
// Answers: what does the dense-layer GEMM reach when no split arithmetic is left inside it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_planes.hip -o gemm_planes && ./gemm_planes [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int BM = 128, BN = 128;          // workgroup tile; 4 wavefronts of 64 x 64
constexpr int FR = 64;                     // uint4 per plane tile (1 KB)

template <int STAGES, bool LINEAR>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ C,
                                                             int M, int N, int KB, int ldc, float inv_scale, int tiles_n) {
#if __HIP_DEVICE_COMPILE__        // (the host pass cannot parse the LDS-DMA builtin; it only needs the stub)
    // one LDS object: [stage][operand][row tile 0..7][plane][64 x 16 B]
    __shared__ uint4 lds[STAGES * 2 * 8 * 2 * FR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int wm = wave >> 1, wn = wave & 1;
    // this wave copies 8 of the 32 plane tiles of a stage: f = wave * 8 + i; operand = f >> 4, row tile = (f >> 1) & 7, plane = f & 1
    // source chunk of lane l inside the 1 KB tile: row (l & 15), k group (l >> 4)  ->  LDS slot l  (conflict-free fragment reads)
    const int src = LINEAR ? lane : (lane & 15) * 4 + (lane >> 4);
    const uint4* gsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = wave * 8 + i, op = f >> 4, rt = (f >> 1) & 7, p = f & 1;
        const uint4* base = op == 0 ? A : B;
        const long long row_tile = (op == 0 ? tm : tn) * 8 + rt;
        const long long max_tile = ((op == 0 ? M : N) + 15) / 16 - 1;
        gsrc[i] = base + ((min(row_tile, max_tile) * KB) * 2 + p) * FR + src;
    }
    auto stage_load = [&](int kb, int st) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = wave * 8 + i;
            __builtin_amdgcn_global_load_lds(gsrc[i] + (long long)kb * 2 * FR, &lds[(st * 32 + f) * FR], 16, 0, 0);
        }
    };
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KB) stage_load(s, s);
    for (int kb = 0; kb < KB; ++kb) {
        const int st = kb % STAGES;
        // stage kb has landed when at most the loads of the (STAGES - 2) younger stages are outstanding
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 3) { if (kb + 1 < KB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { if (kb + 2 < KB) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if (kb + 1 < KB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (kb + STAGES - 1 < KB) stage_load(kb + STAGES - 1, (kb + STAGES - 1) % STAGES);
        const uint4* sa = &lds[(st * 32 + 0 + wm * 8) * FR + lane];
        const uint4* sb = &lds[(st * 32 + 16 + wn * 8) * FR + lane];
        h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = __builtin_bit_cast(h8, sa[(i * 2 + 0) * FR]);
            al[i] = __builtin_bit_cast(h8, sa[(i * 2 + 1) * FR]);
            bh[i] = __builtin_bit_cast(h8, sb[(i * 2 + 0) * FR]);
            bl[i] = __builtin_bit_cast(h8, sb[(i * 2 + 1) * FR]);
        }
        // D[n = 4 (lane >> 4) + q][m = lane & 15]: operand "a" = the B fragment, so a lane holds 4 consecutive columns of C;
        // product kind outermost: 16 independent MFMAs between two that share an accumulator
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al[i] : ah[i], acc[i][j], 0, 0, 0);
    }
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tm * BM + wm * 64 + i * 16 + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * BN + wn * 64 + j * 16 + g * 4;
            if (m < M && n + 3 < N) {
                f4 v = acc[i][j] * inv_scale;
                *reinterpret_cast<f4*>(C + (long long)m * ldc + n) = v;
            }
        }
    }
#endif
}


// Generalised tile: WM x WN wavefronts, each (16 MT) x (16 NT); 2 stages; the LDS-DMA pieces of the next stage are dealt out
// between the row-tile iterations (one per iteration), A fragments double-buffered in registers.  Tiles in the linear order.
template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void gemm_planes_big_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B,
                                                                          float* __restrict__ C, int M, int N, int KB, int ldc,
                                                                          float inv_scale, int tiles_n) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NWAVE = WM * WN, RA = WM * MT, RB = WN * NT, PIECES = 2 * (RA + RB), PW = PIECES / NWAVE;
    static_assert(PIECES % NWAVE == 0, "pieces per wave");
    __shared__ uint4 lds[2 * PIECES * FR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int wm = wave / WN, wn = wave - wm * WN;
    const uint4* gsrc[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int f = wave * PW + i;                       // piece: A pieces first (row tile, plane), then B
        const bool isa = f < 2 * RA;
        const int rt = (isa ? f : f - 2 * RA) >> 1, p = f & 1;
        const long long row_tile = (long long)(isa ? tm * RA : tn * RB) + rt;
        const long long max_tile = ((isa ? M : N) + 15) / 16 - 1;
        gsrc[i] = (isa ? A : B) + ((min(row_tile, max_tile) * KB) * 2 + p) * FR + lane;
    }
    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PW; ++i) __builtin_amdgcn_global_load_lds(gsrc[i], &lds[(wave * PW + i) * FR], 16, 0, 0);
    for (int kb = 0; kb < KB; ++kb) {
        const int st = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint4* sa = &lds[(st * PIECES + wm * MT * 2) * FR + lane];
        const uint4* sb = &lds[(st * PIECES + 2 * RA + wn * NT * 2) * FR + lane];
        const bool more = kb + 1 < KB;
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = __builtin_bit_cast(h8, sb[(j * 2 + 0) * FR]);
            bl[j] = __builtin_bit_cast(h8, sb[(j * 2 + 1) * FR]);
        }
        h8 ah = __builtin_bit_cast(h8, sa[0]), al = __builtin_bit_cast(h8, sa[FR]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            h8 nh = ah, nl = al;
            if (i + 1 < MT) {
                nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                nl = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
            }
#pragma unroll
            for (int q = 0; q < (PW + MT - 1) / MT; ++q) {
                const int pc = i * ((PW + MT - 1) / MT) + q;
                if (pc < PW && more)
                    __builtin_amdgcn_global_load_lds(gsrc[pc] + (long long)(kb + 1) * 2 * FR, &lds[((st ^ 1) * PIECES + wave * PW + pc) * FR], 16, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
            ah = nh;
            al = nl;
        }
    }
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (tm * WM + wm) * MT * 16 + i * 16 + r;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (tn * WN + wn) * NT * 16 + j * 16 + g * 4;
            if (m < M && n + 3 < N) *reinterpret_cast<f4*>(C + (long long)m * ldc + n) = acc[i][j] * inv_scale;
        }
    }
#endif
}

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void gemm_planes_big3_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B,
                                                                          float* __restrict__ C, int M, int N, int KB, int ldc,
                                                                          float inv_scale, int tiles_n) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NWAVE = WM * WN, RA = WM * MT, RB = WN * NT, PIECES = 2 * (RA + RB), PW = PIECES / NWAVE;
    static_assert(PIECES % NWAVE == 0, "pieces per wave");
    __shared__ uint4 lds[3 * PIECES * FR];      // three stages: the pieces of stage kb + 2 are issued during step kb
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int wm = wave / WN, wn = wave - wm * WN;
    const uint4* gsrc[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int f = wave * PW + i;                       // piece: A pieces first (row tile, plane), then B
        const bool isa = f < 2 * RA;
        const int rt = (isa ? f : f - 2 * RA) >> 1, p = f & 1;
        const long long row_tile = (long long)(isa ? tm * RA : tn * RB) + rt;
        const long long max_tile = ((isa ? M : N) + 15) / 16 - 1;
        gsrc[i] = (isa ? A : B) + ((min(row_tile, max_tile) * KB) * 2 + p) * FR + lane;
    }
    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PW; ++i) __builtin_amdgcn_global_load_lds(gsrc[i], &lds[(wave * PW + i) * FR], 16, 0, 0);
    if (KB > 1) {
#pragma unroll
        for (int i = 0; i < PW; ++i) __builtin_amdgcn_global_load_lds(gsrc[i] + 2 * FR, &lds[(PIECES + wave * PW + i) * FR], 16, 0, 0);
    }
    int st = 0;
    for (int kb = 0; kb < KB; ++kb) {
        // stage kb has landed when at most the PW pieces of stage kb + 1 (issued during step kb - 1) are still in flight
        if (kb + 1 < KB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int st2 = st >= 1 ? st - 1 : 2;           // (st + 2) % 3: the stage read in step kb - 1, free again behind the barrier
        const uint4* sa = &lds[(st * PIECES + wm * MT * 2) * FR + lane];
        const uint4* sb = &lds[(st * PIECES + 2 * RA + wn * NT * 2) * FR + lane];
        const bool more = kb + 2 < KB;
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = __builtin_bit_cast(h8, sb[(j * 2 + 0) * FR]);
            bl[j] = __builtin_bit_cast(h8, sb[(j * 2 + 1) * FR]);
        }
        h8 ah = __builtin_bit_cast(h8, sa[0]), al = __builtin_bit_cast(h8, sa[FR]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            h8 nh = ah, nl = al;
            if (i + 1 < MT) {
                nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                nl = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
            }
#pragma unroll
            for (int q = 0; q < (PW + MT - 1) / MT; ++q) {
                const int pc = i * ((PW + MT - 1) / MT) + q;
                if (pc < PW && more)
                    __builtin_amdgcn_global_load_lds(gsrc[pc] + (long long)(kb + 2) * 2 * FR, &lds[(st2 * PIECES + wave * PW + pc) * FR], 16, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
            ah = nh;
            al = nl;
        }
        st = st == 2 ? 0 : st + 1;
    }
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (tm * WM + wm) * MT * 16 + i * 16 + r;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (tn * WN + wn) * NT * 16 + j * 16 + g * 4;
            if (m < M && n + 3 < N) *reinterpret_cast<f4*>(C + (long long)m * ldc + n) = acc[i][j] * inv_scale;
        }
    }
#endif
}

// host-side packing of a row-major fp32 matrix [R][K] (scale s) into the plane-tile layout
void pack(const std::vector<float>& x, int R, int K, float s, std::vector<_Float16>& out, bool linear) {
    const int RT = (R + 15) / 16, KB = (K + 31) / 32;
    out.assign((size_t)RT * KB * 2 * 512, (_Float16)0.f);
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < K; ++k) {
            const float v = x[(size_t)r * K + k] * s;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const size_t t = (((size_t)(r / 16) * KB + k / 32) * 2) * 512 + (linear ? ((k % 32) / 8 * 16 + r % 16) * 8 + k % 8 : (r % 16) * 32 + k % 32);
            out[t] = hi;
            out[t + 512] = lo;
        }
}

typedef void (*launch_fn)(dim3, const uint4*, const uint4*, float*, int, int, int, int, float, int);
template <bool LIN>
void launch2(dim3 g, const uint4* a, const uint4* b, float* c, int M, int N, int KB, int ldc, float inv, int tn) {
    hipLaunchKernelGGL((gemm_planes_kernel<2, LIN>), g, dim3(256), 0, 0, a, b, c, M, N, KB, ldc, inv, tn);
}
template <bool LIN>
void launch3(dim3 g, const uint4* a, const uint4* b, float* c, int M, int N, int KB, int ldc, float inv, int tn) {
    hipLaunchKernelGGL((gemm_planes_kernel<3, LIN>), g, dim3(256), 0, 0, a, b, c, M, N, KB, ldc, inv, tn);
}

template <int MT, int NT, int WM, int WN>
void launch_big(dim3, const uint4* a, const uint4* b, float* c, int M, int N, int KB, int ldc, float inv, int) {
    const int tn = (N + WN * NT * 16 - 1) / (WN * NT * 16), tmm = (M + WM * MT * 16 - 1) / (WM * MT * 16);
    hipLaunchKernelGGL((gemm_planes_big_kernel<MT, NT, WM, WN>), dim3(tmm * tn), dim3(WM * WN * 64), 0, 0, a, b, c, M, N, KB, ldc, inv, tn);
}

template <int MT, int NT, int WM, int WN>
void launch_big3(dim3, const uint4* a, const uint4* b, float* c, int M, int N, int KB, int ldc, float inv, int) {
    const int tn = (N + WN * NT * 16 - 1) / (WN * NT * 16), tmm = (M + WM * MT * 16 - 1) / (WM * MT * 16);
    static bool once = false;
    if (!once) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_big3_kernel<MT, NT, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 0));
        once = true;
    }
    hipLaunchKernelGGL((gemm_planes_big3_kernel<MT, NT, WM, WN>), dim3(tmm * tn), dim3(WM * WN * 64), 0, 0, a, b, c, M, N, KB, ldc, inv, tn);
}

float run(launch_fn fn, const uint4* dA, const uint4* dB, float* dC, int M, int N, int K, float inv, int iters) {
    const int KB = (K + 31) / 32, tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fn(dim3(tiles_m * tiles_n), dA, dB, dC, M, N, KB, N, inv, tiles_n);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fn(dim3(tiles_m * tiles_n), dA, dB, dC, M, N, KB, N, inv, tiles_n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 8096, N = argc > 3 ? atoi(argv[2]) : 4800, K = argc > 3 ? atoi(argv[3]) : 1200;
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    srand(1);
    for (auto& v : a) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : b) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    const float sa = 1024.f, sb = 8192.f * 16.f;
    const float inv = 1.f / (sa * sb);
    const double flop = 2.0 * M * N * K;
    uint4 *dA, *dB;
    float* dC;
    CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    for (int linear = 0; linear < 2; ++linear) {
        std::vector<_Float16> pa, pb;
        pack(a, M, K, sa, pa, linear);
        pack(b, N, K, sb, pb, linear);
        CHECK(hipMalloc(&dA, pa.size() * 2));
        CHECK(hipMalloc(&dB, pb.size() * 2));
        CHECK(hipMemcpy(dA, pa.data(), pa.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, pb.data(), pb.size() * 2, hipMemcpyHostToDevice));
        const float t2 = run(linear ? launch2<true> : launch2<false>, dA, dB, dC, M, N, K, inv, 20);
        const float t3 = run(linear ? launch3<true> : launch3<false>, dA, dB, dC, M, N, K, inv, 20);
        if (linear) {
            const float b1 = run(launch_big<8, 4, 2, 4>, dA, dB, dC, M, N, K, inv, 20);
            const float b2 = run(launch_big<4, 4, 4, 2>, dA, dB, dC, M, N, K, inv, 20);
            const float b3 = run(launch_big<4, 4, 2, 4>, dA, dB, dC, M, N, K, inv, 20);
            const float b4 = run(launch_big<4, 4, 2, 2>, dA, dB, dC, M, N, K, inv, 20);
            const float c2 = run(launch_big3<4, 4, 4, 2>, dA, dB, dC, M, N, K, inv, 20);
            const float c3 = run(launch_big3<4, 4, 2, 4>, dA, dB, dC, M, N, K, inv, 20);
            printf("   3 stages, counted vmcnt, 8 waves: 256x128: %.1f us (%.0f)   128x256: %.1f us (%.0f)\n", c2, flop / c2 * 1e-6, c3, flop / c3 * 1e-6);
            printf("   8 waves 256x256: %.1f us (%.0f)   256x128: %.1f us (%.0f)   128x256: %.1f us (%.0f)   4 waves 128x128: %.1f us (%.0f)\n", b1, flop / b1 * 1e-6,
                   b2, flop / b2 * 1e-6, b3, flop / b3 * 1e-6, b4, flop / b4 * 1e-6);
        }
        printf("M=%d N=%d K=%d %s  2 stages: %.1f us (%.0f fp32-equivalent TFLOP/s)   3 stages: %.1f us (%.0f)\n", M, N, K,
               linear ? "tile = [k / 8][row][8] (linear copy)" : "tile = [row][32] (permuted copy)   ", t2, flop / t2 * 1e-6, t3, flop / t3 * 1e-6);
    }
    // check a sample of entries against double
    std::vector<float> c((size_t)M * N);
    CHECK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int s = 0; s < 4000; ++s) {
        const int m = (int)((size_t)rand() % M), n = ((int)((size_t)rand() % N)) & ~0;
        if (n + 3 >= (N / 4) * 4 && N % 4) continue;
        double ref = 0.0, mag = 0.0;
        for (int k = 0; k < K; ++k) {
            ref += (double)a[(size_t)m * K + k] * b[(size_t)n * K + k];
            mag += fabs((double)a[(size_t)m * K + k] * b[(size_t)n * K + k]);
        }
        worst = fmax(worst, fabs(c[(size_t)m * N + n] - ref) / mag);
    }
    printf("max |err| / sum |a b| over 4000 samples: %.3g\n", worst);
    return 0;
}
