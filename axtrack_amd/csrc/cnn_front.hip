// conv blocks 0 and 1 of the detector in one kernel (see cnn.hip for the rest of the network and for the two separate
// stride-2 kernels this replaces by default: axt_detector_set_fused_front; AXT_FUSE_S2=0 at create keeps them).
#include "axt_common.h"

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
//   * the input arrives by LDS-DMA (buffer_load_dwordx4 ... lds): one channel = 35 rows x 34 16-byte segments = 20 pieces
//     of 1 KiB, three per wave (four of them twice, so that every wave counts the same), rows as they lie in the frame.
//     Segments outside the tile or the frame carry an out-of-range offset, for which the DMA writes zeros (measured:
//     profiles/experiments/lds_dma_oob.hip). Four channel buffers, and the second chunk buffer for the fifth channel: the
//     input of the NEXT tile is issued one piece at a time behind the MFMAs of block 1's phase (back to back a DMA
//     instruction holds its wave for 100+ cycles) -- channel g behind chunk g, whose buffers are free by then.
//   * phase A (one barrier in front: all five channels are in LDS; none inside): block 0 on v_mfma_f32_4x4x1, all 20
//     channels of a pixel tile accumulate in registers over the 5 channels x 9 taps (k ascending: the same sums, in the
//     same order, as conv3x3_s2_k1<5,20>). The region is cut into 9 row pairs x 2 column blocks of 2 x 32 pixels plus the
//     single column X = 2 x0 - 1 (19 units over 8 waves: 3,3,3,2,2,2,2,2 -- 5,5,5,4 per SIMD). Per kernel row a pixel's
//     three taps are one 4-byte and one aligned 8-byte LDS read. Measured: 0.67 of the 4x4x1 rate (what two waves per SIMD
//     reach with this instruction elsewhere, too); on four waves with five units each 0.62.
//   * phase C (per chunk of 4 block-0 channels, one barrier each): the chunk's LeakyReLU'd values go to LDS -- behind the
//     MFMAs of the chunk before --, split into an even-column and an odd-column half per row (unit-stride reads), zero
//     where block 1 pads; then block 1 on
//     v_mfma_f32_16x16x4 -- K = the chunk's 4 channels, one MFMA per tap: wave = one output row = 2 pixel tiles of 16 x 3
//     channel tiles of 16 (40 channels padded to 48). With the 4x4x1 shape a wave has 5 MFMAs per 3 operand reads here and
//     the LDS, not the matrix pipe, bounds the phase (measured: 0.59 ms against 0.38 ms of MFMAs for 252 tile-forwards);
//     the 16x16x4 shape spends 20 % more MFMA cycles on the padding and needs a sixth of the operand bytes.
//   A tile's results leave after the first barrier of the next tile (the vmcnt(0) in front of that barrier covers the DMA
//   pieces and must not wait for a young store).
//   Block 1 sums its products in another order than conv3x3_s2_k1<20,40> (chunk, tap, then the 4 channels inside the
//   instruction): results agree to f32 rounding, not bit for bit (tests/test_gpu_parity.py).
// ------------------------------------------------------------------------------------------------
struct GeoF {
    static constexpr int TH = 8, TW = 32;                                               // block-1 tile of a workgroup
    static constexpr int R1 = 18, HALF1 = 34, RW1 = 2 * HALF1, PLANE1 = R1 * RW1 + 8;   // + 8: plane stride == 16 (mod 64)
    static constexpr int CHUNK = 4 * PLANE1;
    static constexpr int RIN = 35, SEGS = 34, RW0 = 4 * SEGS, NPIECE = 20, PLANE0 = NPIECE * 256, NBUF = 4;
    static constexpr int W0 = 45 * 4 * 8, W1 = 45 * 4 * 48;
    static constexpr int LDS_FLOATS = NBUF * PLANE0 + 2 * CHUNK + W1 + W0;
};
static_assert((GeoF::RIN + 2) * GeoF::RW0 <= GeoF::PLANE0, "the garbage rows of the last row pair must stay inside the plane");
static_assert(GeoF::PLANE1 % 64 == 16, "plane stride of the chunk buffers");
static_assert(GeoF::LDS_FLOATS * 4 <= 160 * 1024, "LDS");

__global__ __launch_bounds__(512, 1) void conv_s2_fused(
    const float *__restrict__ in, const float *__restrict__ wpk0, const float *__restrict__ bias0,
    const float *__restrict__ wpk1, const float *__restrict__ bias1, float *__restrict__ out, int B,
    int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, TileList tl)
{
    using G = GeoF;
    constexpr int RW0 = G::RW0, RW1 = G::RW1, HALF1 = G::HALF1, PLANE1 = G::PLANE1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *IN = smem;                                  // [NBUF][35 rows (+2)][136]
    float *CH = IN + G::NBUF * G::PLANE0;              // [2][4 channels][18 rows][E 34 | O 34]
    float *w1 = CH + 2 * G::CHUNK;                     // [chunk 5][tap 9][channel 4][48]
    float *w0 = w1 + G::W1;                            // [45][4][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jch = lane & 3, blk = lane >> 2;

    {
        // block 1's weights: from conv3x3_s2_k1's image [k = ci*9 + tap][j][12] (channel 4g + j at g) to
        // [chunk][tap][ci % 4][48 output channels], zero beyond 40
        for (int e = tid; e < G::W1; e += 512) {
            const int n = e % 48, c = (e / 48) & 3, tap = (e / 192) % 9, g = e / 1728;
            const int k = (4 * g + c) * 9 + tap;
            w1[e] = n < 40 ? wpk1[(k * 4 + (n & 3)) * 12 + (n >> 2)] : 0.f;
        }
        const f32x4 *s0 = reinterpret_cast<const f32x4 *>(wpk0);
        f32x4 *l0 = reinterpret_cast<f32x4 *>(w0);
        for (int e = tid; e < G::W0 / 4; e += 512) l0[e] = s0[e];
    }
    __syncthreads();

    // work list: as conv3x3_s2_k1<FIRST>: every XCD a contiguous range of items, walked one band of tiles at a time
    constexpr int TILES_X = 128 / G::TW, NTILE = TILES_X * (128 / G::TH);
    // (fewer than 8 items: one list for all workgroups -- an XCD's share of the items would be empty for most XCDs)
    const int xcd = blockIdx.x & 7;
    const int it_begin = B >= 8 ? (int)((long)xcd * B / 8) : 0;
    const int ni = B >= 8 ? (int)((long)(xcd + 1) * B / 8) - it_begin : B;
    const int wstep = B >= 8 ? gridDim.x >> 3 : gridDim.x, wend = ni * NTILE;
    int w = B >= 8 ? blockIdx.x >> 3 : blockIdx.x;
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
            a0_base[u] = 2 * (2 * rp + (lane >> 5)) * RW0 + 2 * (2 + 32 * cb + (lane & 31)) - 1;
            const int rw = 2 * rp + (blk >> 3);
            wr_base[u] = jch * PLANE1 + rw * RW1 + 1 + 16 * cb + 2 * (blk & 7);
            wr_top[u] = rw == 0;
        } else {                                       // the column X = 2 x0 - 1 (h = 1): lane = row
            a0_base[u] = 2 * min(lane, 17) * RW0 + 1;
            wr_base[u] = jch * PLANE1 + HALF1;
            wr_top[u] = false;
        }
    }
    // block 1: wave = output row `wave`; lane = (pixel lane % 16, chunk channel lane / 16) for A, (channel lane % 16,
    // chunk channel lane / 16) for B; results: channel 16 ct + lane % 16, pixels 16 xh + 4 (lane / 16) + i
    const int a1_base = (lane >> 4) * PLANE1 + 2 * wave * RW1 + (lane & 15);
    const int b1_base = (lane >> 4) * 48 + (lane & 15);
    float b0v[5], b1v[3];
#pragma unroll
    for (int g = 0; g < 5; ++g) b0v[g] = bias0[4 * g + jch];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) b1v[ct] = 16 * ct + (lane & 15) < 40 ? bias1[16 * ct + (lane & 15)] : 0.f;

    // ---- the input DMA: piece p covers segments [64 p, 64 p + 64) of a channel plane (34 segments of 4 floats per row) ----
    const int cstride = Hf * Wf, rstride = Wf;
    unsigned rc[3], eff[3], rc18, eff18 = 0;
    int piece[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        piece[k] = k < 2 ? wave + 8 * k : 16 + (wave & 3);
        const int q = piece[k] * 64 + lane;
        const int row = q / G::SEGS, sg = q - row * G::SEGS;
        rc[k] = (row < G::RIN && sg < 33) ? (unsigned)(row | (sg << 8)) : 0xffffffffu;
    }
    {
        const int q = 18 * 64 + lane, row = q / G::SEGS, sg = q - row * G::SEGS;
        rc18 = (row < G::RIN && sg < 33) ? (unsigned)(row | (sg << 8)) : 0xffffffffu;
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
        // rows with 0 <= iy0 + row < lim_y; segments entirely inside [0, lim_x) (widths are multiples of 4)
        const unsigned r_lo = (unsigned)max(0, -iy0), r_n = (unsigned)max(0, min(G::RIN, lim_y - iy0) - (int)r_lo);
        const unsigned s_lo = jx0 < 0 ? 1u : 0u, s_n = (unsigned)max(0, min(33, (lim_x - jx0) / 4) - (int)s_lo);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned row = rc[k] & 0xffu, sg = (rc[k] >> 8) & 0xffu;
            const bool ok = rc[k] != 0xffffffffu && row - r_lo < r_n && sg - s_lo < s_n;
            eff[k] = ok ? (row * (unsigned)rstride + 4u * sg) * 4u : kOobOffset;
        }
        {
            const unsigned row = rc18 & 0xffu, sg = (rc18 >> 8) & 0xffu;
            const bool ok = rc18 != 0xffffffffu && row - r_lo < r_n && sg - s_lo < s_n;
            eff18 = ok ? (row * (unsigned)rstride + 4u * sg) * 4u : kOobOffset;
        }
    };
    auto issue = [&](int ci, int k) {                  // piece k of channel ci of the tile decoded last -> IN[ci] (channel 4: CH[1])
        if (AXT_FUSED_ABLATE & 1) return;
        // channel 4 lands in the second chunk buffer, which is 192 floats shorter than a plane: its last piece (rows 35.7+,
        // read only for the unused 18th row of block 0) is not fetched -- piece 18 is, twice, so that every wave counts alike
        const int p = (ci == 4 && k == 2 && (wave & 3) == 3) ? 18 : piece[k];
        float *dst = (ci < 4 ? IN + ci * G::PLANE0 : CH + G::CHUNK) + p * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (__attribute__((address_space(3))) void *)dst, 16,
                                                 (int)(p == piece[k] ? eff[k] : eff18), ci * cstride * 4, 0, 0);
    };
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- accumulators ----
    f32x4 acc0[3][5], acc1[2][3] = {};
    auto reset0 = [&](int u, int g) { acc0[u][g] = f32x4{b0v[g], b0v[g], b0v[g], b0v[g]}; };
    auto reset1 = [&](int xh, int ct) { acc1[xh][ct] = f32x4{b1v[ct], b1v[ct], b1v[ct], b1v[ct]}; };

    // block 0, input channel ci: 9 taps on this wave's NU pixel tiles
    auto phase_a = [&](auto nu_tag, int ci) {
        constexpr int NU = decltype(nu_tag)::value;
        const float *src = ci < 4 ? IN + ci * G::PLANE0 : CH + G::CHUNK;
        const float *wc = w0 + ci * 9 * 32 + jch * 8;
        // operands of one kernel row (3 taps) per fetch: the pixel's columns 2h - 1 | 2h, 2h + 1 -- the pair is one aligned
        // 8-byte read (lanes 8 bytes apart: no bank conflict; the 4-byte reads at stride 2 are 2-way conflicts)
        float a0[2][NU];
        f32x2 a12[2][NU];
        f32x4 bq[2][3];
        float bs[2][3];
        auto fetch = [&](int ky) {
            const int slot = ky & 1;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                a0[slot][u] = src[a0_base[u] + ky * RW0];
                a12[slot][u] = *reinterpret_cast<const f32x2 *>(src + a0_base[u] + ky * RW0 + 1);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bq[slot][kx] = *reinterpret_cast<const f32x4 *>(wc + (3 * ky + kx) * 32);
                bs[slot][kx] = wc[(3 * ky + kx) * 32 + 4];
            }
        };
        fetch(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NU + 6, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (ky < 2) fetch(ky + 1);
            const int slot = ky & 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int g = 0; g < 5; ++g)
                        acc0[u][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(kx == 0 ? a0[slot][u] : a12[slot][u][kx - 1],
                                                                       g < 4 ? bq[slot][kx][g] : bs[slot][kx], acc0[u][g], 0, 0, 0);
            if (ky < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NU + 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NU * 15, 0);
        }
    };
    int cy0 = 0, cx0 = 0, cb_ = 0;
    // block-0 channels 4g..4g+3 of unit u: LeakyReLU, zero where block 1 pads, to LDS; accumulators back to the bias
    auto write_unit = [&](auto g_tag, int u, float *dst) {
        constexpr int g = decltype(g_tag)::value;
        if (AXT_FUSED_ABLATE & 2) return;
        if (u == 2 && !three) return;
        const bool top = cy0 == 0, left = cx0 == 0;
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
    };
    // block 1, chunk g (block-0 channels 4g..4g+3) from CH[g & 1]. Behind its MFMAs (32 cycles each: the wave has issue
    // slots to spare): chunk g + 1 on its way to the other buffer, and three of the next tile's input pieces.
    auto phase_c = [&](auto g_tag, bool prefetch) {
        constexpr int g = decltype(g_tag)::value;
        constexpr int PF = 2;
        const float *src = CH + (g & 1) * G::CHUNK + a1_base;
        const float *wc = w1 + g * 36 * 48 + b1_base;
        float *nxt = CH + ((g + 1) & 1) * G::CHUNK;
        if (g == 0) {
#pragma unroll
            for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) reset1(xh, ct);
        }
        float a[PF + 1][2], b[PF + 1][3];
        auto fetch = [&](int k) {
            const int ky = k / 3, kx = k % 3, slot = k % (PF + 1);
            const int off = ky * RW1 + (kx == 0 ? HALF1 : kx == 1 ? 1 : HALF1 + 1);
            a[slot][0] = src[off];
            a[slot][1] = src[off + 16];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) b[slot][ct] = wc[k * 192 + 16 * ct];
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) fetch(k);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k + PF < 9) fetch(k + PF);
            const int slot = k % (PF + 1);
            if (!(AXT_FUSED_ABLATE & 4)) {
#pragma unroll
                for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
                        acc1[xh][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[slot][xh], b[slot][ct], acc1[xh][ct], 0, 0, 0);
            }
            if (g < 4 && k >= 1 && k <= 3) write_unit(std::integral_constant<int, (g < 4 ? g + 1 : 4)>{}, k - 1, nxt);
            if (prefetch && k >= 4 && k <= 6) issue(g, k - 4);
        }
    };
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, B * 40 * 128 * 128 * 4, 0x00020000);
    auto store_tile = [&]() {                          // the tile whose block-1 sums are in acc1
        if (AXT_FUSED_ABLATE & 16) return;
        const int base = ((cb_ * 40 * 128 + cy0 + wave) * 128 + cx0 + 4 * (lane >> 4)) * 4;
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                f32x4 v = acc1[xh][ct];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * 0.1f);
                const int ch = 16 * ct + (lane & 15);
                // tile-dependent part in the VECTOR offset: see conv3x3_s2_k1's write_tile. Channels 40..47 are padding.
                const unsigned off = ch < 40 ? (unsigned)(base + ch * 128 * 128 * 4 + 16 * xh * 4) : kOobOffset;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst_rsrc, (int)off, 0, 0);
            }
    };

    // One tile = one barrier + phase A (5 channels, no barrier between them: all five are in LDS before it starts) + 5 x
    // (barrier, phase C). The input of tile t + 1 streams in behind the MFMAs of tile t's phase C; the results of tile t
    // leave after the first barrier of tile t + 1, so that the vmcnt(0) before that barrier -- which has to cover the DMA
    // pieces, and loads and stores do not retire in one order -- never waits for a store younger than a whole tile.
    decode(w);
    int nb_cur = nb, ny_cur = ny0, nx_cur = nx0;
#pragma unroll
    for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int k = 0; k < 3; ++k) issue(ci, k);
    bool pending = false;
#pragma unroll 1
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();                                     // the input is visible; CH[0] (chunk 4 of the last tile) is read
        if (pending) store_tile();
        cb_ = nb_cur; cy0 = ny_cur; cx0 = nx_cur;
        // the accumulators start at the folded bias (set here, not where they are consumed: they are dead in between)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int g = 0; g < 5; ++g) reset0(u, g);
        if (!(AXT_FUSED_ABLATE & 8)) {
            if (three) {
#pragma unroll
                for (int ci = 0; ci < 5; ++ci) phase_a(std::integral_constant<int, 3>{}, ci);
            } else {
#pragma unroll
                for (int ci = 0; ci < 5; ++ci) phase_a(std::integral_constant<int, 2>{}, ci);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) write_unit(std::integral_constant<int, 0>{}, u, CH);
        const bool has_next = w + wstep < wend;
        // every barrier: chunk g is visible; chunk g - 1 (at g = 0: the input; at g = 4: CH[1], where channel 4 goes) is read
        barrier();
        if (has_next) decode(w + wstep);
        phase_c(std::integral_constant<int, 0>{}, has_next);
        barrier();
        phase_c(std::integral_constant<int, 1>{}, has_next);
        barrier();
        phase_c(std::integral_constant<int, 2>{}, has_next);
        barrier();
        phase_c(std::integral_constant<int, 3>{}, has_next);
        barrier();
        phase_c(std::integral_constant<int, 4>{}, has_next);
        pending = true;
        if (!has_next) break;
        w += wstep;
        nb_cur = nb; ny_cur = ny0; nx_cur = nx0;
    }
    store_tile();
}

}  // namespace

// conv blocks 0 + 1 in one launch (conv_s2_fused): frames -> block 1's output [B,40,128,128]
int axt_launch_conv_fused01(const float *in, const float *w0, const float *b0, const float *w1, const float *b1, float *out,
                        int B, hipStream_t st, int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, const TileList &tl)
{
    constexpr size_t lds = (size_t)GeoF::LDS_FLOATS * sizeof(float);
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(conv_s2_fused, (int)lds, once)) return rc;
    AXT_REQUIRE((double)Hf * Wf * 5 * 4 < 2.0e9, "conv: frames of %d x %d are too large", Hf, Wf);
    AXT_REQUIRE((double)B * 40 * 128 * 128 * 4 < 2.0e9, "conv: batch of %d is too large for one launch", B);
    const int nwork = B * (128 / GeoF::TH) * (128 / GeoF::TW);
    const int grid = nwork < 256 ? (nwork + 7) / 8 * 8 : 256;        // persistent: one workgroup per CU, a multiple of 8 (one slice per XCD)
    hipLaunchKernelGGL(conv_s2_fused, dim3(grid), dim3(512), lds, st, in, w0, b0, w1, b1, out, B, Hf, Wf,
                       t0, tstep, item0, n_tiles, tl);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

