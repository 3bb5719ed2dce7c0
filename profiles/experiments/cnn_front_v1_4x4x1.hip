// conv blocks 0 and 1 of the detector in one kernel (see cnn.hip for the rest of the network and for the two separate
// stride-2 kernels this replaces when the detector is created with AXT_FUSE_S2=1).
#include "axt_common.h"

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef AXT_FUSED_ABLATE
#define AXT_FUSED_ABLATE 0      // timing-only builds: 1 no input DMA | 2 no chunk writes | 4 no block-1 MFMAs | 8 no block-0 MFMAs | 16 no stores
#endif

namespace {

constexpr unsigned kOobOffset = 0x80000000u;       // byte offset beyond every descriptor's range: the load returns 0
constexpr int kBufRecords = 0x7fffffff;

// ------------------------------------------------------------------------------------------------
// conv_s2_fused: conv blocks 0 AND 1 in one kernel -- block 0's output (0.67 GB per 128 tile-forwards, written and read
// back once by the two separate kernels, which that traffic bounds) never leaves the CU.
//   * a workgroup (8 waves, one per CU) owns an 8 x 32 tile of block 1's output. That needs 17 rows x 65 columns of
//     block 0's output (all 20 channels), which need 35 rows x 131 columns of the 5 input frames.
//   * the input arrives by LDS-DMA (buffer_load_dword ... lds), one channel = 80 pieces of 64 floats, de-interleaved on
//     the fly by the per-lane SOURCE offsets into an even-column and an odd-column half per row (unit-stride operand
//     reads); elements outside the tile or the frame carry an out-of-range offset, for which the DMA writes zeros
//     (measured: profiles/experiments/lds_dma_oob.hip). Four channel buffers: channels 0-3 of the NEXT tile are issued
//     when block 1's phase of the current tile starts, channel 4 as soon as channel 0's buffer is free again.
//   * phase A (per input channel, one barrier each): block 0 on v_mfma_f32_4x4x1, all 20 channels of a pixel tile
//     accumulate in registers over the 5 channels x 9 taps (k ascending: the same sums, in the same order, as
//     conv3x3_s2_k1<5,20>). The region is cut into 9 row pairs x 2 column blocks of 2 x 32 pixels plus the single
//     column X = 2 x0 - 1 (19 units over 8 waves: 3,3,3,2,2,2,2,2 -- 5,5,5,4 per SIMD).
//   * phase C (per chunk of 4 block-0 channels, one barrier each): the chunk's LeakyReLU'd values go to LDS
//     (de-interleaved like conv3x3_s2_k1's patches, zero where block 1 pads), then block 1's 36 k-steps on it: wave =
//     one 2 x 32 pixel tile x 5 of the 10 channel groups (groups 4h..4h+3 and 8+h: one b128 + one b32 weight read).
//   Results are bit-identical to the two separate kernels (tests/test_gpu_parity.py).
// ------------------------------------------------------------------------------------------------
struct GeoF {
    static constexpr int TH = 8, TW = 32;                                               // block-1 tile of a workgroup
    static constexpr int R1 = 18, HALF1 = 34, RW1 = 2 * HALF1, PLANE1 = R1 * RW1, CHUNK = 4 * PLANE1;
    static constexpr int RIN = 35, HALF0 = 68, RW0 = 2 * HALF0, NPIECE = 80, PLANE0 = NPIECE * 64, NBUF = 4;
    static constexpr int W0 = 45 * 4 * 8, W1 = 180 * 4 * 12;
    static constexpr int LDS_FLOATS = NBUF * PLANE0 + 2 * CHUNK + W1 + W0 + 64;
};
static_assert((GeoF::RIN + 2) * GeoF::RW0 <= GeoF::PLANE0, "the garbage rows of the last row pair must stay inside the plane");

__global__ __launch_bounds__(512, 1) void conv_s2_fused(
    const float *__restrict__ in, const float *__restrict__ wpk0, const float *__restrict__ bias0,
    const float *__restrict__ wpk1, const float *__restrict__ bias1, float *__restrict__ out, int B,
    int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, TileList tl)
{
    using G = GeoF;
    constexpr int RW0 = G::RW0, HALF0 = G::HALF0, RW1 = G::RW1, HALF1 = G::HALF1, PLANE1 = G::PLANE1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *IN = smem;                                  // [NBUF][35 rows (+2)][E 68 | O 68]
    float *CH = IN + G::NBUF * G::PLANE0;              // [2][4 channels][18 rows][E 34 | O 34]
    float *w1 = CH + 2 * G::CHUNK;                     // [180][4][12]
    float *w0 = w1 + G::W1;                            // [45][4][8]
    float *bl = w0 + G::W0;                            // bias0 [j][5], bias1 [j][10]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jch = lane & 3, blk = lane >> 2;

    {
        const f32x4 *s1 = reinterpret_cast<const f32x4 *>(wpk1), *s0 = reinterpret_cast<const f32x4 *>(wpk0);
        f32x4 *l1 = reinterpret_cast<f32x4 *>(w1), *l0 = reinterpret_cast<f32x4 *>(w0);
        for (int e = tid; e < G::W1 / 4; e += 512) l1[e] = s1[e];
        for (int e = tid; e < G::W0 / 4; e += 512) l0[e] = s0[e];
        if (tid < 20) bl[(tid & 3) * 5 + (tid >> 2)] = bias0[tid];
        if (tid >= 64 && tid < 104) bl[20 + ((tid - 64) & 3) * 10 + ((tid - 64) >> 2)] = bias1[tid - 64];
    }
    __syncthreads();

    // work list: as conv3x3_s2_k1<FIRST>: every XCD a contiguous range of items, walked one band of tiles at a time
    constexpr int TILES_X = 128 / G::TW, NTILE = TILES_X * (128 / G::TH);
    const int xcd = blockIdx.x & 7;
    const int it_begin = (int)((long)xcd * B / 8);
    const int ni = (int)((long)(xcd + 1) * B / 8) - it_begin;
    const int wstep = gridDim.x >> 3, wend = ni * NTILE;
    int w = blockIdx.x >> 3;
    if (w >= wend) return;

    // ---- roles ----
    const bool three = wave < 3, special = wave == 7;
    const int u0 = three ? 3 * wave : 9 + 2 * (wave - 3);
    int a0_base[3], wr_base[3];
    bool wr_top[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int id = u0 + u;
        if (id < 18) {
            const int rp = id >> 1, cb = id & 1;
            a0_base[u] = 2 * (2 * rp + (lane >> 5)) * RW0 + 2 + 32 * cb + (lane & 31);
            const int rw = 2 * rp + (blk >> 3);
            wr_base[u] = jch * PLANE1 + rw * RW1 + 1 + 16 * cb + 2 * (blk & 7);
            wr_top[u] = rw == 0;
        } else {                                       // the column X = 2 x0 - 1 (h = 1): lane = row
            a0_base[u] = 2 * min(lane, 17) * RW0 + 1;
            wr_base[u] = jch * PLANE1 + HALF1;
            wr_top[u] = false;
        }
    }
    const int pt = wave & 3, half = wave >> 2;
    const int a1_base = 2 * (2 * pt + (lane >> 5)) * RW1 + (lane & 31);
    auto gidx = [&](int gi) { return gi < 4 ? 4 * half + gi : 8 + half; };

    // ---- the input DMA: piece p = wave + 8 k covers floats [64 p, 64 p + 64) of a channel plane ----
    const int cstride = Hf * Wf, rstride = Wf;
    unsigned rc[10], eff[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int q = (wave + 8 * k) * 64 + lane;
        const int row = q / RW0, wq = q - row * RW0;
        const bool ok = row < G::RIN && (wq < 66 || (wq >= HALF0 && wq < HALF0 + 66));
        const int c = wq < HALF0 ? 2 * wq : 2 * (wq - HALF0) + 1;
        rc[k] = ok ? (unsigned)(row | (c << 8)) : 0xffffffffu;
    }
    __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 0, 0x00020000);
    int nb = 0, ny0 = 0, nx0 = 0;
    auto decode = [&](int ww) {                        // tile ww -> nb, ny0, nx0, src_rsrc, eff[]
        const int per_band = ni * TILES_X;
        const int band = ww / per_band, r = ww - band * per_band;
        const int ib = r / TILES_X;
        nb = it_begin + ib;
        ny0 = band * G::TH;
        nx0 = (r - ib * TILES_X) * G::TW;
        const int item = item0 + nb;
        const int t = t0 + (item / n_tiles) * tstep, kt = item % n_tiles;
        const int oy = tl.yx[2 * kt] * AXT_TILE, ox = tl.yx[2 * kt + 1] * AXT_TILE;
        const int lim_y = min(AXT_TILE, Hf - oy), lim_x = min(AXT_TILE, Wf - ox);
        const int iy0 = 4 * ny0 - 3, jx0 = 4 * nx0 - 4;
        const long src = ((long)t * Hf + oy + iy0) * Wf + ox + jx0;
        src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in) + src, 0, kBufRecords, 0x00020000);
        const unsigned r_lo = (unsigned)max(0, -iy0), r_n = (unsigned)max(0, min(G::RIN, lim_y - iy0) - (int)r_lo);
        const unsigned c_lo = (unsigned)max(0, -jx0), c_n = (unsigned)max(0, min(132, lim_x - jx0) - (int)c_lo);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const unsigned row = rc[k] & 0xffu, c = (rc[k] >> 8) & 0xffu;
            const bool ok = rc[k] != 0xffffffffu && row - r_lo < r_n && c - c_lo < c_n;
            eff[k] = ok ? (row * (unsigned)rstride + c) * 4u : kOobOffset;
        }
    };
    auto issue = [&](int ci) {                         // channel ci of the tile decoded last -> IN[ci % NBUF]
        if (AXT_FUSED_ABLATE & 1) return;
        float *dst = IN + (ci % G::NBUF) * G::PLANE0 + wave * 64;
        const int soff = ci * cstride * 4;
#pragma unroll
        for (int k = 0; k < 10; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (__attribute__((address_space(3))) void *)(dst + k * 512), 4,
                                                     (int)eff[k], soff, 0, 0);
    };
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- accumulators ----
    f32x4 acc0[3][5], acc1[5];
    auto reset0 = [&](int u, int g) { const float bv = bl[jch * 5 + g]; acc0[u][g] = f32x4{bv, bv, bv, bv}; };
    auto reset1 = [&](int gi) { const float bv = bl[20 + jch * 10 + gidx(gi)]; acc1[gi] = f32x4{bv, bv, bv, bv}; };
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int g = 0; g < 5; ++g) reset0(u, g);
#pragma unroll
    for (int gi = 0; gi < 5; ++gi) reset1(gi);

    // block 0, input channel ci: 9 taps on this wave's NU pixel tiles
    auto phase_a = [&](auto nu_tag, int ci) {
        constexpr int NU = decltype(nu_tag)::value;
        constexpr int PF = 2;
        const float *src = IN + (ci % G::NBUF) * G::PLANE0;
        const float *wc = w0 + ci * 9 * 32 + jch * 8;
        float a[PF + 1][NU], bs[PF + 1];
        f32x4 bq[PF + 1];
        auto fetch = [&](int k) {
            const int ky = k / 3, kx = k % 3, slot = k % (PF + 1);
            const int off = ky * RW0 + (kx == 0 ? HALF0 - 1 : kx == 1 ? 0 : HALF0);
#pragma unroll
            for (int u = 0; u < NU; ++u) a[slot][u] = src[a0_base[u] + off];
            bq[slot] = *reinterpret_cast<const f32x4 *>(wc + k * 32);
            bs[slot] = wc[k * 32 + 4];
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) fetch(k);
        __builtin_amdgcn_sched_group_barrier(0x100, PF * (NU + 2), 0);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k + PF < 9) fetch(k + PF);
            const int slot = k % (PF + 1);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int g = 0; g < 5; ++g)
                    acc0[u][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[slot][u], g < 4 ? bq[slot][g] : bs[slot], acc0[u][g], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NU + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NU * 5, 0);
        }
    };
    // block 1, chunk g (block-0 channels 4g..4g+3) from CH[g & 1]
    auto phase_c = [&](int g) {
        constexpr int PF = 3;
        const float *src = CH + (g & 1) * G::CHUNK + a1_base;
        const float *wc = w1 + g * 36 * 48 + jch * 12;
        float a[PF + 1], bs[PF + 1];
        f32x4 bq[PF + 1];
        auto fetch = [&](int k) {
            const int c = k / 9, ky = (k % 9) / 3, kx = k % 3, slot = k % (PF + 1);
            a[slot] = src[c * PLANE1 + ky * RW1 + (kx == 0 ? HALF1 : kx == 1 ? 1 : HALF1 + 1)];
            bq[slot] = *reinterpret_cast<const f32x4 *>(wc + k * 48 + 4 * half);
            bs[slot] = wc[k * 48 + 8 + half];
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) fetch(k);
        __builtin_amdgcn_sched_group_barrier(0x100, PF * 3, 0);
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            if (k + PF < 36) fetch(k + PF);
            const int slot = k % (PF + 1);
#pragma unroll
            for (int gi = 0; gi < 5; ++gi)
                acc1[gi] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[slot], gi < 4 ? bq[slot][gi] : bs[slot], acc1[gi], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        }
    };
    int cy0 = 0, cx0 = 0, cb_ = 0;
    // block-0 channels 4g..4g+3 of the wave's units: LeakyReLU, zero where block 1 pads, to LDS; accumulators back to the bias
    auto write_chunk = [&](auto g_tag, float *dst) {
        constexpr int g = decltype(g_tag)::value;
        if (AXT_FUSED_ABLATE & 2) return;
        const bool top = cy0 == 0, left = cx0 == 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u == 2 && !three) break;
            f32x4 v = acc0[u][g];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * 0.1f);
            if (special && u == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = min(4 * blk + i, 17);
                    dst[wr_base[u] + row * RW1] = (left || (top && row == 0)) ? 0.f : v[i];
                }
            } else {
                if (top && wr_top[u]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                float *p = dst + wr_base[u];
                p[0] = v[0];
                p[1] = v[2];
                p[HALF1] = v[1];
                p[HALF1 + 1] = v[3];
            }
            reset0(u, g);
        }
    };
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, B * 40 * 128 * 128 * 4, 0x00020000);
    auto store_tile = [&]() {
        if (AXT_FUSED_ABLATE & 16) return;
        const int base = ((cb_ * 40 * 128 + cy0 + 2 * pt + (blk >> 3)) * 128 + cx0 + (blk & 7) * 4) * 4;
#pragma unroll
        for (int gi = 0; gi < 5; ++gi) {
            f32x4 v = acc1[gi];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * 0.1f);
            // tile-dependent part in the VECTOR offset: see conv3x3_s2_k1's write_tile
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst_rsrc,
                                                   base + (4 * gidx(gi) + jch) * 128 * 128 * 4, 0, 0);
            reset1(gi);
        }
    };

    decode(w);
    cb_ = nb; cy0 = ny0; cx0 = nx0;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) issue(ci);
#pragma unroll 1
    for (;;) {
        // ---- phase A: vmcnt counts the pieces issued after the channel that is needed (loads retire in order; stores
        // in flight only make the wait longer) ----
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            if (ci == 0) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            else if (ci == 1 || ci == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (ci == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier();
            if (ci == 1) issue(4);                     // channel 0's buffer: every wave is past its phase
            if (!(AXT_FUSED_ABLATE & 8)) {
                if (three) phase_a(std::integral_constant<int, 3>{}, ci);
                else phase_a(std::integral_constant<int, 2>{}, ci);
            }
        }
        write_chunk(std::integral_constant<int, 0>{}, CH);
        const bool has_next = w + wstep < wend;
#pragma unroll 1
        for (int g = 0; g < 5; ++g) {
            barrier();
            if (g == 0 && has_next) {
                decode(w + wstep);
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) issue(ci);
            }
            if (!(AXT_FUSED_ABLATE & 4)) phase_c(g);
            switch (g) {
            case 0: write_chunk(std::integral_constant<int, 1>{}, CH + G::CHUNK); break;
            case 1: write_chunk(std::integral_constant<int, 2>{}, CH); break;
            case 2: write_chunk(std::integral_constant<int, 3>{}, CH + G::CHUNK); break;
            case 3: write_chunk(std::integral_constant<int, 4>{}, CH); break;
            default: break;
            }
        }
        store_tile();
        if (!has_next) break;
        w += wstep;
        cb_ = nb; cy0 = ny0; cx0 = nx0;
    }
}

}  // namespace

// conv blocks 0 + 1 in one launch (conv_s2_fused): frames -> block 1's output [B,40,128,128]
int axt_launch_conv_fused01(const float *in, const float *w0, const float *b0, const float *w1, const float *b1, float *out,
                        int B, hipStream_t st, int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, const TileList &tl)
{
    constexpr size_t lds = (size_t)GeoF::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)conv_s2_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    AXT_REQUIRE((double)Hf * Wf * 5 * 4 < 2.0e9, "conv: frames of %d x %d are too large", Hf, Wf);
    AXT_REQUIRE((double)B * 40 * 128 * 128 * 4 < 2.0e9, "conv: batch of %d is too large for one launch", B);
    const int nwork = B * (128 / GeoF::TH) * (128 / GeoF::TW);
    const int grid = nwork < 256 ? (nwork + 7) / 8 * 8 : 256;        // persistent: one workgroup per CU, a multiple of 8 (one slice per XCD)
    hipLaunchKernelGGL(conv_s2_fused, dim3(grid), dim3(512), lds, st, in, w0, b0, w1, b1, out, B, Hf, Wf,
                       t0, tstep, item0, n_tiles, tl);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

