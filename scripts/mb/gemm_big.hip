// Prototype of round 3's dense-layer GEMM: C[M, N] (fp32) = A B^T on operands pre-split into 16-bit (hi, lo) planes in MFMA-fragment
// order (the layout of csrc/gemm_planes.hip), as a PERSISTENT big-tile kernel:
//   * workgroup = 8 wavefronts (2 x 4), wave tile (16 MT) x (16 NT), workgroup tile (32 MT) x (64 NT): 256 x 256, 256 x 320, 128 x 320 ...
//     chosen per problem so that the tile count fills whole rounds of the 256 CUs (N = 4800 = 15 x 320);
//   * grid = min(tiles, CUs); every workgroup walks ITS tiles (XCD-aware, band-major order) in one flat (tile, k-step) loop: the
//     LDS-DMA of the next tile's first stage is in flight while the finished tile's accumulators are stored - the pipeline never
//     drains between tiles, no workgroup waits for another (safe next to the persistent recurrence kernels);
//   * LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): one VGPR of lane offset, per-piece offsets in SGPRs.
// Times it against the 128 x 128 kernel of the product at the step's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_big.hip -o gemm_big && ./gemm_big [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int FR = 64;                     // uint4 per plane tile (1 KB)

struct BigArgs {
    const uint4* A;
    const uint4* B;
    float* C;
    const float* bias;
    int M, N, KB;
    long long ldc;
    int accumulate;
    int tiles_m, tiles_n, band;
    unsigned a_bytes, b_bytes;
    float inv;
};

// ---------------------------------------------------------------------------------------------------------------- the new kernel
template <bool BF16, int MT, int NT, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const BigArgs G) {
#if __HIP_DEVICE_COMPILE__
    constexpr int WM = 2, WN = 4, RA = WM * MT, RB = WN * NT, PIECES = 2 * (RA + RB);
    constexpr int PA = 2 * RA / 8, PB = 2 * RB / 8, PW = PA + PB;
    static_assert((2 * RA) % 8 == 0 && (2 * RB) % 8 == 0, "pieces per wave");
    __shared__ uint4 lds[2 * PIECES * FR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.A), 0, G.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.B), 0, G.b_bytes, 0x00020000);
    const int voff = lane * 16;
    const int KB = G.KB;
    const int rta = (G.M + 15) / 16, rtb = (G.N + 15) / 16;

    // this workgroup's tiles: workgroup id b runs on XCD b % 8; every XCD owns one contiguous range of the band-major tile list
    const int T = G.tiles_m * G.tiles_n;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q = T / 8, r8 = T % 8;
    const int cnt = xcd < r8 ? q + 1 : q;
    const int first = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    if (j0 >= cnt) return;
    const int band_tiles = G.band * G.tiles_n;
    auto tile_rc = [&](int idx, int& tm, int& tn) {
        const int b = idx / band_tiles, rem = idx - b * band_tiles;
        const int h = min(G.band, G.tiles_m - b * G.band);
        tn = rem / h;
        tm = b * G.band + (rem - tn * h);
    };
    auto issue = [&](int i, int tm, int tn, int kb, int st) {            // piece i of this wave for (tile, k-step) into stage st
        if (i < PA) {
            const int f = wave * PA + i;
            const int rt = min(tm * RA + (f >> 1), rta - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, &lds[(st * PIECES + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        } else {
            const int f = wave * PB + (i - PA);
            const int rt = min(tn * RB + (f >> 1), rtb - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, &lds[(st * PIECES + 2 * RA + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        }
    };

    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    if ((VAR & 16) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    int idx = j0, tm, tn;
    tile_rc(first + idx, tm, tn);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(i, tm, tn, 0, 0);
    int st = 0;
    constexpr int PPI = (PW + MT - 1) / MT;             // pieces issued per row-tile iteration
    while (true) {
        const int nidx = idx + per;
        const bool has_next_tile = nidx < cnt;
        int ntm = tm, ntn = tn;
        if (has_next_tile) tile_rc(first + nidx, ntm, ntn);
        for (int kb = 0; kb < KB; ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of stage `st` have landed
            if (!(VAR & 4)) __builtin_amdgcn_s_barrier();             // everybody's have; everybody is done reading stage st ^ 1
            const bool last = kb + 1 == KB;
            const int ptm = last ? ntm : tm, ptn = last ? ntn : tn, pkb = last ? 0 : kb + 1;
            const uint4* sa = &lds[(st * PIECES + wm * MT * 2) * FR + lane];
            const uint4* sb = &lds[(st * PIECES + 2 * RA + wn * NT * 2) * FR + lane];
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
                if (i + 1 < MT && !(VAR & 8)) {
                    nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                    nl = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
                }
#pragma unroll
                for (int pc = i * PPI; pc < (i + 1) * PPI; ++pc)
                    if (pc < PW && !(VAR & 2)) issue(pc, ptm, ptn, pkb, st ^ 1);
                if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = BF16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, p == 0 ? bl[j] : bh[j]),
                                                                                   __builtin_bit_cast(b8, p == 1 ? al : ah), acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
                if (VAR & 1) __builtin_amdgcn_s_setprio(0);
                ah = nh;
                al = nl;
            }
            st ^= 1;
        }
        // epilogue of tile (tm, tn): the next tile's first stage is in flight meanwhile
        {
            const int r = lane & 15, g = lane >> 4;
            const int n0 = (tn * WN + wn) * NT * 16 + g * 4;
            const bool vec = (G.ldc & 3) == 0 && (reinterpret_cast<unsigned long long>(G.C) & 15) == 0;
            const bool full = vec && (tn * WN + wn + 1) * NT * 16 <= G.N;           // wave-uniform: every column of this wave's tile exists
            f4 bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bv[j] = f4{0.f, 0.f, 0.f, 0.f};
                if (G.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[j][e] = G.bias[min(n0 + j * 16 + e, G.N - 1)];
                }
            }
            float* const crow = G.C + (long long)((tm * WM + wm) * MT * 16 + r) * G.ldc + n0;
            const int mrow = (tm * WM + wm) * MT * 16 + r;
            if (full && !G.accumulate) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * G.inv + bv[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        if (mrow + i * 16 < G.M) *reinterpret_cast<f4*>(crow + (long long)i * 16 * G.ldc + j * 16) = v;
                    }
                }
            } else if (full) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
                    f4 old[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) old[j] = ok ? *reinterpret_cast<const f4*>(crow + (long long)i * 16 * G.ldc + j * 16) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * G.inv + bv[j] + old[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        if (ok) *reinterpret_cast<f4*>(crow + (long long)i * 16 * G.ldc + j * 16) = v;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * G.inv + bv[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        float* o = crow + (long long)i * 16 * G.ldc + j * 16;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ok && n0 + j * 16 + e < G.N) o[e] = G.accumulate ? o[e] + v[e] : v[e];
                    }
                }
            }
        }
        if (!has_next_tile) break;          // (the stage requested during the last step is never read; the wait below retires it)
        idx = nidx;
        tm = ntm;
        tn = ntn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---------------------------------------------------------------------------------------------------------------- 32 x 32 x 16 MFMA variant
// The same tile loop on v_mfma_f32_32x32x16_f16 (half the operand-register reads per MAC): does the power-limited rate change?
// 256 x 256 tile, 8 wavefronts (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32; fragments from the same 16 x 32 plane tiles:
// lane l = (row l % 32 of two adjacent row tiles, k group l / 32) reads chunk (g = 2 h + l / 32, r = l % 16) of row tile (l % 32) / 16.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm_big32_kernel(const BigArgs G) {
#if __HIP_DEVICE_COMPILE__
    constexpr int MT = 8, NT = 4, WN = 4, RA = 16, RB = 16, PIECES = 64, PA = 4, PB = 4, PW = 8;
    __shared__ uint4 lds[2 * PIECES * FR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.A), 0, G.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.B), 0, G.b_bytes, 0x00020000);
    const int voff = lane * 16;
    const int KB = G.KB;
    const int rta = (G.M + 15) / 16, rtb = (G.N + 15) / 16;
    const int T = G.tiles_m * G.tiles_n;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q = T / 8, r8 = T % 8;
    const int cnt = xcd < r8 ? q + 1 : q;
    const int first = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    if (j0 >= cnt) return;
    const int band_tiles = G.band * G.tiles_n;
    auto tile_rc = [&](int idx, int& tm, int& tn) {
        const int b = idx / band_tiles, rem = idx - b * band_tiles;
        const int h = min(G.band, G.tiles_m - b * G.band);
        tn = rem / h;
        tm = b * G.band + (rem - tn * h);
    };
    auto issue = [&](int i, int tm, int tn, int kb, int st) {
        if (i < PA) {
            const int f = wave * PA + i;
            const int rt = min(tm * RA + (f >> 1), rta - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, &lds[(st * PIECES + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        } else {
            const int f = wave * PB + (i - PA);
            const int rt = min(tn * RB + (f >> 1), rtb - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, &lds[(st * PIECES + 2 * RA + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        }
    };
    f16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int idx = j0, tm, tn;
    tile_rc(first + idx, tm, tn);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(i, tm, tn, 0, 0);
    int st = 0;
    // lane's chunk inside a 32-row pair of plane tiles: row tile (l % 32) / 16 (pieces are (row tile, plane): + 2 pieces), slot 16 g + r
    const int lofs = ((lane & 31) >> 4) * 2 * FR + (lane >> 5) * 16 + (lane & 15);       // in uint4; + h * 32 for the k half, + FR for lo
    while (true) {
        const int nidx = idx + per;
        const bool has_next_tile = nidx < cnt;
        int ntm = tm, ntn = tn;
        if (has_next_tile) tile_rc(first + nidx, ntm, ntn);
        for (int kb = 0; kb < KB; ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool last = kb + 1 == KB;
            const int ptm = last ? ntm : tm, ptn = last ? ntn : tn, pkb = last ? 0 : kb + 1;
            const uint4* sa = &lds[(st * PIECES + wm * MT * 2) * FR + lofs];
            const uint4* sb = &lds[(st * PIECES + 2 * RA + wn * NT * 2) * FR + lofs];
            h8 bh[2][2], bl[2][2];          // [32-column tile][k half]
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bh[j][h] = __builtin_bit_cast(h8, sb[j * 4 * FR + h * 32]);
                    bl[j][h] = __builtin_bit_cast(h8, sb[j * 4 * FR + FR + h * 32]);
                }
            h8 ah[2], al[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ah[h] = __builtin_bit_cast(h8, sa[h * 32]);
                al[h] = __builtin_bit_cast(h8, sa[FR + h * 32]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h8 nh[2] = {ah[0], ah[1]}, nl[2] = {al[0], al[1]};
                if (i + 1 < 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        nh[h] = __builtin_bit_cast(h8, sa[(i + 1) * 4 * FR + h * 32]);
                        nl[h] = __builtin_bit_cast(h8, sa[(i + 1) * 4 * FR + FR + h * 32]);
                    }
                }
#pragma unroll
                for (int pc = i * 2; pc < i * 2 + 2; ++pc) issue(pc, ptm, ptn, pkb, st ^ 1);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p == 0 ? bl[j][h] : bh[j][h], p == 1 ? al[h] : ah[h], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ah[h] = nh[h];
                    al[h] = nl[h];
                }
            }
            st ^= 1;
        }
        {
            // D[n = 8 (e / 4) + 4 (lane / 32) + e % 4][m = lane % 32]: four consecutive columns of C per 4 accumulator registers
            const int ml = lane & 31, nq = (lane >> 5) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (tm * 2 + wm) * 128 + i * 32 + ml;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int n = (tn * 4 + wn) * 64 + j * 32 + e4 * 8 + nq;
                        const f4 v = f4{acc[i][j][e4 * 4], acc[i][j][e4 * 4 + 1], acc[i][j][e4 * 4 + 2], acc[i][j][e4 * 4 + 3]} * G.inv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e4 * 4 + e] = 0.f;
                        if (m < G.M && n + 3 < G.N) *reinterpret_cast<f4*>(G.C + (long long)m * G.ldc + n) = v;
                    }
            }
        }
        if (!has_next_tile) break;
        idx = nidx;
        tm = ntm;
        tn = ntn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---------------------------------------------------------------------------------------------------------------- the product's 128 x 128 kernel (baseline)
template <bool BF16>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ C,
                                                             int M, int N, int KB, long long ldc, float inv, int tiles_m, int tiles_n) {
#if __HIP_DEVICE_COMPILE__
    constexpr int PIECES = 32;
    __shared__ uint4 lds[2 * PIECES * FR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = tiles_m * tiles_n;
    const int q = T / 8, r8 = T % 8, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= (xcd < r8 ? q + 1 : q)) return;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int wm = wave >> 1, wn = wave & 1;
    const int rta = (M + 15) / 16, rtb = (N + 15) / 16;
    const uint4* gsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = wave * 8 + i;
        const bool isa = f < 16;
        const int rt = (f & 15) >> 1, p = f & 1;
        const long long row_tile = min((long long)(isa ? tm : tn) * 8 + rt, (long long)(isa ? rta : rtb) - 1);
        gsrc[i] = (isa ? A : B) + ((row_tile * KB) * 2 + p) * FR + lane;
    }
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds(gsrc[i], &lds[(wave * 8 + i) * FR], 16, 0, 0);
    for (int kb = 0; kb < KB; ++kb) {
        const int st = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint4* sa = &lds[(st * PIECES + wm * 8) * FR + lane];
        const uint4* sb = &lds[(st * PIECES + 16 + wn * 8) * FR + lane];
        const bool more = kb + 1 < KB;
        h8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = __builtin_bit_cast(h8, sb[(j * 2 + 0) * FR]);
            bl[j] = __builtin_bit_cast(h8, sb[(j * 2 + 1) * FR]);
        }
        h8 ah = __builtin_bit_cast(h8, sa[0]), al = __builtin_bit_cast(h8, sa[FR]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h8 nh = ah, nl = al;
            if (i + 1 < 4) {
                nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                nl = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
            }
            if (more) {
#pragma unroll
                for (int pc = 2 * i; pc < 2 * i + 2; ++pc)
                    __builtin_amdgcn_global_load_lds(gsrc[pc] + (long long)(kb + 1) * 2 * FR, &lds[((st ^ 1) * PIECES + wave * 8 + pc) * FR], 16, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = BF16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, p == 0 ? bl[j] : bh[j]),
                                                                               __builtin_bit_cast(b8, p == 1 ? al : ah), acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
            ah = nh;
            al = nl;
        }
    }
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tm * 128 + wm * 64 + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * 128 + wn * 64 + j * 16 + g * 4;
            if (n + 3 < N) *reinterpret_cast<f4*>(C + (long long)m * ldc + n) = acc[i][j] * inv;
        }
    }
#endif
}

// host-side packing of a row-major fp32 matrix [R][K] (scale s) into the plane-tile layout (fragment order)
void pack(const std::vector<float>& x, int R, int K, float s, std::vector<_Float16>& out) {
    const int RT = (R + 15) / 16, KB = (K + 31) / 32;
    out.assign((size_t)RT * KB * 2 * 512, (_Float16)0.f);
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < K; ++k) {
            const float v = x[(size_t)r * K + k] * s;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const size_t t = (((size_t)(r / 16) * KB + k / 32) * 2) * 512 + ((k % 32) / 8 * 16 + r % 16) * 8 + k % 8;
            out[t] = hi;
            out[t + 512] = lo;
        }
}

struct Problem {
    int M, N, K, KB;
    uint4 *dA, *dB;
    float* dC;
    float inv;
    size_t a_bytes, b_bytes;
};

template <int MT, int NT, int VAR>
void launch_big(const Problem& P, int band, int cus) {
    BigArgs G{P.dA, P.dB, P.dC, nullptr, P.M, P.N, P.KB, (long long)P.N, 0, (P.M + 32 * MT - 1) / (32 * MT), (P.N + 64 * NT - 1) / (64 * NT),
              band, (unsigned)P.a_bytes, (unsigned)P.b_bytes, P.inv};
    const int T = G.tiles_m * G.tiles_n;
    const int grid = std::min((T + 7) / 8 * 8, cus);
    hipLaunchKernelGGL((gemm_big_kernel<false, MT, NT, VAR>), dim3(grid), dim3(512), 0, 0, G);
}

void launch_big32(const Problem& P, int band, int cus) {
    BigArgs G{P.dA, P.dB, P.dC, nullptr, P.M, P.N, P.KB, (long long)P.N, 0, (P.M + 255) / 256, (P.N + 255) / 256,
              band, (unsigned)P.a_bytes, (unsigned)P.b_bytes, P.inv};
    const int T = G.tiles_m * G.tiles_n;
    const int grid = std::min((T + 7) / 8 * 8, cus);
    hipLaunchKernelGGL((gemm_big32_kernel<0>), dim3(grid), dim3(512), 0, 0, G);
}

void launch_old(const Problem& P, int, int) {
    const int tm = (P.M + 127) / 128, tn = (P.N + 127) / 128;
    hipLaunchKernelGGL((gemm_planes_kernel<false>), dim3((tm * tn + 7) / 8 * 8), dim3(256), 0, 0, P.dA, P.dB, P.dC, P.M, P.N, P.KB, (long long)P.N,
                       P.inv, tm, tn);
}

typedef void (*launch_fn)(const Problem&, int, int);

float run(launch_fn fn, const Problem& P, int band, int cus, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fn(P, band, cus);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fn(P, band, cus);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

double check(const Problem& P, const std::vector<float>& a, const std::vector<float>& b) {
    std::vector<float> c((size_t)P.M * P.N);
    CHECK(hipMemcpy(c.data(), P.dC, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    srand(7);
    for (int s = 0; s < 6000; ++s) {
        int m = (int)((size_t)rand() % P.M), n = (int)((size_t)rand() % P.N);
        if (s < 600) m = P.M - 1 - (s % 40);          // the edges
        if (s >= 600 && s < 1200) n = P.N - 1 - (s % 40);
        double ref = 0.0, mag = 0.0;
        for (int k = 0; k < P.K; ++k) {
            ref += (double)a[(size_t)m * P.K + k] * b[(size_t)n * P.K + k];
            mag += fabs((double)a[(size_t)m * P.K + k] * b[(size_t)n * P.K + k]);
        }
        worst = fmax(worst, fabs(c[(size_t)m * P.N + n] - ref) / mag);
    }
    return worst;
}

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 8096, N = argc > 3 ? atoi(argv[2]) : 4800, K = argc > 3 ? atoi(argv[3]) : 1200;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    srand(1);
    const bool zero = getenv("ZERO") != nullptr;          // zero-filled operands: what the clock does without data toggling (DVFS)
    for (auto& v : a) v = zero ? 0.f : (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : b) v = zero ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    const float sa = 1024.f, sb = 8192.f * 16.f;
    Problem P{M, N, K, (K + 31) / 32, nullptr, nullptr, nullptr, 1.f / (sa * sb), 0, 0};
    const double flop = 2.0 * M * N * K;
    CHECK(hipMalloc(&P.dC, (size_t)M * N * 4));
    std::vector<_Float16> pa, pb;
    pack(a, M, K, sa, pa);
    pack(b, N, K, sb, pb);
    P.a_bytes = pa.size() * 2;
    P.b_bytes = pb.size() * 2;
    CHECK(hipMalloc(&P.dA, P.a_bytes));
    CHECK(hipMalloc(&P.dB, P.b_bytes));
    CHECK(hipMemcpy(P.dA, pa.data(), P.a_bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(P.dB, pb.data(), P.b_bytes, hipMemcpyHostToDevice));
    printf("M=%d N=%d K=%d  (%d CUs)   us (fp32-equivalent TFLOP/s; of 833)   max |err| / sum |a b|\n", M, N, K, cus);
    struct V { const char* name; launch_fn fn; int band; };
    const V vs[] = {
        {"old 128x128 (product r2)      ", launch_old, 0},
        {"big 256x256 band 4            ", launch_big<8, 4, 0>, 4},
        {"big 256x256 band 4 setprio    ", launch_big<8, 4, 1>, 4},
        {"big 256x256 32x32x16 MFMA     ", launch_big32, 4},
        {"big 256x320 band 4            ", launch_big<8, 5, 0>, 4},
        {"big 256x320 band 4 setprio    ", launch_big<8, 5, 1>, 4},
        {"big 256x320 waves 4-7 prio 1  ", launch_big<8, 5, 16>, 4},
        {"256x320 ABLATION no DMA       ", launch_big<8, 5, 2>, 4},
        {"256x320 ABLATION no barrier   ", launch_big<8, 5, 4>, 4},
        {"256x320 ABLATION no DMA no bar", launch_big<8, 5, 6>, 4},
        {"256x320 ABLATION no A reads   ", launch_big<8, 5, 8>, 4},
        {"256x320 ABLATION no A rd/DMA  ", launch_big<8, 5, 10>, 4},
        {"256x320 ABLATION none of them ", launch_big<8, 5, 14>, 4},
        {"big 128x320 band 4            ", launch_big<4, 5, 0>, 4},
        {"big 128x320 band 8            ", launch_big<4, 5, 0>, 8},
        {"big 128x256 band 8            ", launch_big<4, 4, 0>, 8},
        {"big 256x192 band 4            ", launch_big<8, 3, 0>, 4},
    };
    for (const V& v : vs) {
        CHECK(hipMemset(P.dC, 0xff, (size_t)M * N * 4));
        const float t = run(v.fn, P, v.band, cus, 20);
        const double err = check(P, a, b);
        printf("  %s %8.1f us  (%5.0f; %.3f)   %.3g\n", v.name, t, flop / t * 1e-6, flop / t * 1e-6 / 833.3, err);
    }
    return 0;
}
