// Detector forward pass for gfx950 (MI355X): YOLO_AXTrack (reference axtrack/machinelearning/model.py:20-125)
// as f32-in / f32-accumulate MFMA kernels.
//
//   conv3x3_mfma     stride-1 implicit-GEMM 3x3 convolution + folded BatchNorm + LeakyReLU(0.1) (+ fused
//                    MaxPool2d(2,2)) for conv blocks 2..10: the input patch and a K-chunk of the weights are staged
//                    in LDS, v_mfma_f32_16x16x4_f32 accumulates 16 pixels x 16 channels per instruction, the next
//                    chunk's global loads are prefetched into registers behind the MFMAs.
//   conv3x3_s2_mfma  the two stride-2 blocks: persistent workgroups, every wave its own barrier-free pipeline,
//                    weights LDS-resident. Block 0 reads the 5-frame temporal stack straight from the timelapse
//                    (fuses Timelapse.get_frametiles_stack, Timelapse.py:111-125,150-157).
//   gemm_mfma        split-K GEMM for the three linear layers, partial slabs reduced in a fixed order
//                    (bit-reproducible) by reduce_bias_act (+ Sigmoid).
//
// Numerics: f32 MFMA on gfx950 is a k-ordered chain of f32 FMAs (exact f32, no reduced precision).
// BatchNorm is folded in f64 at pack time and rounded once to f32.
#include "axt_common.h"

#include <math.h>
#include <stdlib.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// network description (deployed ARCHITECTURE, deployed_model/params.txt:34)
// ------------------------------------------------------------------------------------------------
struct ConvSpec { int cin, cout, stride, pool, hin; };
static const ConvSpec kConv[8] = {
    {5, 20, 2, 0, 512}, {20, 40, 2, 0, 256}, {40, 80, 1, 1, 128}, {80, 80, 1, 0, 64},
    {80, 80, 1, 1, 64}, {80, 80, 1, 0, 32},  {80, 80, 1, 1, 32},  {80, 160, 1, 0, 16}};
constexpr int kFeat = 160 * 16 * 16;   // 40960
constexpr int kFc = 1024;
constexpr int kOut = AXT_YOLO_FLOATS;  // 432
constexpr int kOutPad = 448;           // 7 x 64

// ------------------------------------------------------------------------------------------------
// conv kernels: shared pieces
// ------------------------------------------------------------------------------------------------
constexpr int npadw(int nt) { return (nt % 2) ? nt * 16 : nt * 16 + 16; }   // row stride == 16 (mod 32)

struct TileList { int n; short yx[2 * 256]; };

// Persistent workgroups: the grid is a multiple of 8 and every workgroup walks a strided list of work items
// (batch item, channel group, 16x16 output tile). Workgroups are dealt round-robin over the 8 XCDs
// (block b -> XCD b % 8), so the list is cut into 8 contiguous ranges, one per XCD: neighbouring tiles -- which
// share halos -- and neighbouring batch items -- which share 4 of their 5 input frames -- meet in one L2.
struct WorkRange { int begin, end, step; };
__device__ __forceinline__ WorkRange my_work(int nwork)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_wg = gridDim.x >> 3;
    const int per_xcd = (nwork + 7) >> 3;
    WorkRange r;
    r.begin = xcd * per_xcd + slot;
    r.end = min(nwork, (xcd + 1) * per_xcd);
    r.step = per_xcd_wg;
    return r;
}

// epilogue shared by the conv kernels: + folded bias, LeakyReLU(0.1), optional 2x2 max, store NCHW.
// acc[m][n][j]: pixel (row y0 + wave*MT + m, x0 + q*4 + j), channel grp*NT*16 + n*16 + p.
// The lane's NT bias values are loaded once, before the main loop: a load here would make the compiler drain
// the vector-memory queue (prefetches and earlier stores included) in every epilogue.
template <int COUT, int NT>
__device__ __forceinline__ void load_bias(const float *__restrict__ bias, int grp, int p, float (&bias_v)[NT])
{
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ch = grp * NT * 16 + n * 16 + p;
        bias_v[n] = bias[ch < COUT ? ch : 0];
    }
}

template <int COUT, bool POOL, int MT, int NT, bool BIAS_IN_ACC = false>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[MT][NT], const float (&bias_v)[NT],
                                              float *__restrict__ out, int b, int grp, int y0, int x0, int wave, int p,
                                              int q, int Hout, int Wout)
{
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ch = grp * NT * 16 + n * 16 + p;
        if (ch >= COUT) continue;
        const float bv = BIAS_IN_ACC ? 0.f : bias_v[n];
        float *och = out + ((long)b * COUT + ch) * Hout * Wout;
        if constexpr (!POOL) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v = acc[m][n];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = BIAS_IN_ACC ? v[j] : v[j] + bv;
                    v[j] = fmaxf(t, t * 0.1f);          // LeakyReLU(0.1): t > 0 ? t : 0.1 t
                }
                const int row = y0 + wave * MT + m, x = x0 + q * 4;
                *reinterpret_cast<f32x4 *>(och + (long)row * Wout + x) = v;
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; m += 2) {
                float v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[m][n][j] + bv;
                    v0[j] = t > 0.f ? t : t * 0.1f;
                    t = acc[m + 1][n][j] + bv;
                    v1[j] = t > 0.f ? t : t * 0.1f;
                }
                float2 r;
                r.x = fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v1[0], v1[1]));
                r.y = fmaxf(fmaxf(v0[2], v0[3]), fmaxf(v1[2], v1[3]));
                const int row = (y0 + wave * MT + m) >> 1, x = (x0 + q * 4) >> 1;
                *reinterpret_cast<float2 *>(och + (long)row * Wout + x) = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stride-1 conv3x3 + folded BN + LeakyReLU (+ maxpool 2x2): conv blocks 2,4,5,7,8,10
//   work item = (batch item, channel group of NT*16, 16x16 output tile); block: 256 threads = 4 waves, wave w owns
//   rows [4w, 4w+4) of the tile and all NT*16 channels of the group: 4 x NT accumulator tiles of 16 px x 16 ch.
//   K = 9*CIN runs in chunks of CCH channels: the (18 x 18 x CCH) input patch and the chunk's weights are staged in
//   LDS. The chunks are software-pipelined: the global loads of chunk k+1 are issued into registers before the
//   MFMAs of chunk k and written to LDS after them (async-STAGE split).
//   LDS patch layout: channel-planar, plane stride == 16 (mod 32) dwords: the four k of an MFMA step are four
//   channels, lanes 16..31 then read 16 banks away from lanes 0..15 (conflict-free ds_read_b32).
// ------------------------------------------------------------------------------------------------
template <int CCH>
struct GeoS1 {
    static constexpr int MT = 4, TH = 16, PH = 18, PW = 18, RAW = PH * PW;
    static constexpr int PLANE = RAW + ((16 - RAW % 32) + 32) % 32;        // 336
    static constexpr int KSTEPS = 9 * CCH / 4, KROWS = KSTEPS * 4;
};

template <int CIN, int COUT, bool POOL, int CCH, int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma(
    const float *__restrict__ in,       // activations [B,CIN,Hin,Hin]
    const float *__restrict__ wpk,      // packed weights [ngroup][nchunk][KROWS][NPADW]
    const float *__restrict__ bias,     // folded bias [COUT]
    float *__restrict__ out,            // [B,COUT,Hout,Hout]
    int Hin, int ngroups, int B)
{
    using G = GeoS1<CCH>;
    constexpr int MT = G::MT, PH = G::PH, PW = G::PW, PLANE = G::PLANE, KSTEPS = G::KSTEPS, KROWS = G::KROWS;
    constexpr int NPADW = npadw(NT);
    constexpr int NCHUNK = CIN / CCH;
    static_assert(CIN % CCH == 0 && CCH % 4 == 0, "CIN must be a multiple of CCH, CCH of 4");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                            // [CCH][PLANE]
    float *wl = smem + ((CCH * PLANE + 3) & ~3);     // [KROWS][NPADW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int tiles_x = Hin / 16, ntile = tiles_x * tiles_x;
    // one work item per workgroup; blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
    // range of items (bijective remap): neighbouring tiles, which share halos and weights, meet in one L2
    int w;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    }
    const int tile = w % ntile, rest = w / ntile;
    const int grp = rest % ngroups, b = rest / ngroups;
    const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    const long cstride = (long)Hin * Hin;
    const int Hout = POOL ? Hin / 2 : Hin;

    // staging plan: element e = tid + k*256 of the patch is (channel c, row r, col). Branch-free: elements past
    // the patch go to a dummy LDS word, elements outside the image load from a clamped (valid) address and are
    // zeroed by a select. Per chunk all loads are issued back to back into registers and written to LDS after the
    // barrier, while the next chunk's loads are already in flight behind the MFMAs (async-STAGE split).
    constexpr int NPE = (CCH * PH * PW + 255) / 256;
    constexpr int NW4 = KROWS * NPADW / 4;
    constexpr int NWE = (NW4 + 255) / 256;
    constexpr int DUMMY = CCH * PLANE - 1;       // last (padding) word of the last plane, never read by the MFMAs
    static_assert(PLANE > PH * PW, "the plane padding provides the dummy word");
    int loff[NPE], goff[NPE];
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
        const int e = tid + k * 256;
        const int c = e / (PH * PW), rem = e - c * (PH * PW);
        const int r = rem / PW, col = rem - r * PW;
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        const bool in_patch = e < CCH * PH * PW;
        const bool ok = in_patch && gy >= 0 && gy < Hin && gx >= 0 && gx < Hin;
        loff[k] = in_patch ? c * PLANE + r * PW + col : DUMMY;
        const int g = c * (int)cstride + gy * Hin + gx;
        goff[k] = ok ? g : -1;
    }
    float pv[NPE];
    f32x4 wv[NWE];
    const float *isrc = in + (long)b * CIN * cstride;
    const float *wsrc = wpk + (long)grp * NCHUNK * KROWS * NPADW;
    auto load_chunk = [&](int chunk) {
        const float *csrc = isrc + (long)chunk * CCH * cstride;
#pragma unroll
        for (int k = 0; k < NPE; ++k) pv[k] = csrc[goff[k] < 0 ? 0 : goff[k]];     // masked at store time
        const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wsrc + (long)chunk * KROWS * NPADW);
#pragma unroll
        for (int k = 0; k < NWE; ++k) {
            const int e = tid + k * 256;
            wv[k] = w4[(NW4 % 256 == 0 || e < NW4) ? e : 0];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < NPE; ++k) patch[loff[k]] = goff[k] < 0 ? 0.f : pv[k];
        f32x4 *l4 = reinterpret_cast<f32x4 *>(wl);
#pragma unroll
        for (int k = 0; k < NWE; ++k) {
            const int e = tid + k * 256;
            if (NW4 % 256 == 0 || k + 1 < NWE || e < NW4) l4[e] = wv[k];
        }
    };
    float bias_v[NT];
    load_bias<COUT, NT>(bias, grp, p, bias_v);

    const int a_base = q * PLANE + (wave * MT) * PW + p;
    const int b_base = q * NPADW + p;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk) __syncthreads();          // every wave is done reading the previous chunk
        store_chunk();
        __syncthreads();
        if (chunk + 1 < NCHUNK) load_chunk(chunk + 1);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            float a[MT], bw[NT];
            constexpr int CG = CCH / 4;
            const int kk = s / CG, cg = s % CG;      // (ky,kx) major, channel group minor
            const int ky = kk / 3, kx = kk % 3;
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = patch[a_base + cg * 4 * PLANE + (m + ky) * PW + kx];
#pragma unroll
            for (int n = 0; n < NT; ++n) bw[n] = wl[b_base + s * 4 * NPADW + n * 16];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bw[n], acc[m][n], 0, 0, 0);
        }
    }
    conv_epilogue<COUT, POOL, MT, NT>(acc, bias_v, out, b, grp, y0, x0, wave, p, q, Hout, Hout);
}

// ------------------------------------------------------------------------------------------------
// stride-2 conv3x3 (conv blocks 0 and 1): low arithmetic intensity, so data movement decides the speed.
//   * every WAVE is its own pipeline: it stages the 9 input rows its 4 output rows need (rows of 36 floats starting
//     3 columns left of the patch, 2*x0 - 4, a multiple of 4: aligned 16-byte global loads and ds_write_b128) into a
//     wave-private LDS region. The main loop therefore has no workgroup barrier; the waves of a CU drift apart and
//     the loads, MFMAs and stores of different waves overlap. Halo rows shared by two waves are staged twice (9 rows
//     instead of 8.25).
//   * the next (tile, chunk)'s loads are issued into registers before the MFMAs of the current one and written to
//     LDS after them; the zero mask for padding is applied at that store, so nothing touches the loaded registers
//     (and forces a wait) while the MFMAs run. Workgroups are persistent and walk an XCD-contiguous tile list.
//   * K is ordered k = c*9 + ky*3 + kx per chunk and addressed through a per-lane offset table, so the four k of one
//     MFMA step mostly differ in kx: with stride-2 pixel addresses (even banks) the next k lands on the odd banks.
//   * ALL the layer's weights stay in LDS for the life of the workgroup (staged once, one barrier).
//   conv block 0 (FIRST) gathers its 5 channels straight from frames t..t+4 of the timelapse at the tile origin
//   (fuses Timelapse.get_frametiles_stack, Timelapse.py:111-125,150-157), zero beyond the tile and the frame.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int NPC, int NT, bool FIRST, int WPS>
__global__ __launch_bounds__(256) void conv3x3_s2_mfma(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ zeros, int Hin, int B,
    int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, TileList tl)
{
    constexpr int MT = 4, PHW = 9, RW = 36, PLANE = PHW * RW;        // per wave: 9 rows x 36 floats per channel
    int rag_jx0 = 0, rag_limx = 0;                                    // only used for ragged (unaligned) frames
    constexpr int WREGION = NPC * PLANE + 4;                          // + one spare float4 (dummy store target)
    constexpr int NCHUNK = CIN / NPC;
    constexpr int KREAL = 9 * NPC, KSTEPS = (KREAL + 3) / 4, KROWS = KSTEPS * 4;
    constexpr int NPADW = npadw(NT);
    static_assert(CIN % NPC == 0, "CIN must be a multiple of the chunk");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *patch = smem + wave * WREGION;     // this wave's [NPC][9][36]
    float *wl = smem + 4 * WREGION;           // [NCHUNK][KROWS][NPADW]  (whole layer, shared)

    const int p = lane & 15, q = lane >> 4;
    const int tiles_x = Hin / 32, ntile = tiles_x * tiles_x;     // output is Hin/2, tiles of 16
    const WorkRange wr = my_work(ntile * B);

    // the layer's weights: once per workgroup
    {
        constexpr int NW4 = NCHUNK * KROWS * NPADW / 4;
        const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wpk);
        f32x4 *l4 = reinterpret_cast<f32x4 *>(wl);
        for (int e = tid; e < NW4; e += 256) l4[e] = w4[e];
    }
    __syncthreads();
    if (wr.begin >= wr.end) return;

    int koff[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int k = s * 4 + q;
        const int kk = k < KREAL ? k : 0;                  // padded k rows carry zero weights
        koff[s] = (kk / 9) * PLANE + ((kk % 9) / 3) * RW + (kk % 3) + 3;
    }
    const int a_base = 2 * p;
    const int b_base = q * NPADW + p;

    // staging plan of the wave (float4 granularity): element e = lane + k*64 is (channel c, row r, 4-column segment).
    // The decomposition is done once; per tile an element costs one add (its offset), a few compares and an
    // address select: lanes whose segment lies outside the image read a 16-byte block of zeros instead, so zero
    // padding needs no masking of the loaded data (nothing consumes the loaded registers before the MFMAs).
    constexpr int NP4 = NPC * PHW * (RW / 4);
    constexpr int NPE = (NP4 + 63) / 64;
    constexpr int DUMMY = NPC * PLANE;                    // the wave's spare float4
    int loff[NPE], crs[NPE];                 // LDS offset; packed (c << 16 | r << 8 | seg), c = 255 for "no element"
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
        const int e = lane + k * 64;
        const int c = e / (PHW * (RW / 4)), rem = e - c * (PHW * (RW / 4));
        const int r = rem / (RW / 4), seg = rem - r * (RW / 4);
        loff[k] = e < NP4 ? (c * PLANE + r * RW + 4 * seg) : DUMMY;
        crs[k] = e < NP4 ? ((c << 16) | (r << 8) | seg) : (255 << 16);
    }
    const float *pl[NPE];                    // this step's load addresses
    f32x4 pv[NPE];

    int cur_b = 0, cur_y0 = 0, cur_x0 = 0;
    int nxt_b = 0, nxt_y0 = 0, nxt_x0 = 0, nxt_cstride = 0, nxt_aligned = 0;
    auto decode_plan = [&](int w) {
        const int tile = w % ntile;
        nxt_b = w / ntile;
        nxt_y0 = (tile / tiles_x) * 16;
        nxt_x0 = (tile % tiles_x) * 16;
        int lim_y, lim_x, rstride;
        long src;
        if constexpr (FIRST) {
            const int item = item0 + nxt_b;
            const int t = t0 + (item / n_tiles) * tstep, k = item % n_tiles;
            const int oy = tl.yx[2 * k] * AXT_TILE, ox = tl.yx[2 * k + 1] * AXT_TILE;
            src = ((long)t * Hf + oy) * Wf + ox;
            nxt_cstride = Hf * Wf;
            rstride = Wf;
            lim_y = min(AXT_TILE, Hf - oy);
            lim_x = min(AXT_TILE, Wf - ox);
            nxt_aligned = (Wf % 4 == 0) && (lim_x % 4 == 0);
        } else {
            src = (long)nxt_b * CIN * Hin * Hin;
            nxt_cstride = Hin * Hin;
            rstride = Hin;
            lim_y = Hin;
            lim_x = Hin;
            nxt_aligned = 1;
        }
        // first staged row of this wave / first staged column (a multiple of 4)
        const int iy0 = (nxt_y0 + wave * MT) * 2 - 1, jx0 = nxt_x0 * 2 - 4;
        const float *base = in + src + (long)iy0 * rstride + jx0;
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
            const int c = crs[k] >> 16, r = (crs[k] >> 8) & 255, seg = crs[k] & 255;
            const int gy = iy0 + r, gx = jx0 + 4 * seg;
            // aligned images: a segment is entirely inside or entirely outside. Ragged ones (width not a multiple
            // of 4) take the per-float path in load_chunk and only need the row test here.
            const bool ok = c < NPC && gy >= 0 && gy < lim_y && (!nxt_aligned || (gx >= 0 && gx + 3 < lim_x));
            pl[k] = ok ? base + (c * nxt_cstride + r * rstride + 4 * seg) : zeros;
        }
        if (!nxt_aligned) { rag_jx0 = jx0; rag_limx = lim_x; }
    };
    auto load_chunk = [&](int chunk) {
        const long coff = (long)chunk * NPC * nxt_cstride;
        if (nxt_aligned) {
#pragma unroll
            for (int k = 0; k < NPE; ++k)
                pv[k] = *reinterpret_cast<const f32x4 *>(pl[k] == zeros ? zeros : pl[k] + coff);
        } else {
#pragma unroll
            for (int k = 0; k < NPE; ++k) {
                const int gx = rag_jx0 + 4 * (crs[k] & 255);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = pl[k] != zeros && gx + j >= 0 && gx + j < rag_limx;
                    v[j] = ok ? pl[k][coff + j] : 0.f;
                }
                pv[k] = v;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < NPE; ++k) *reinterpret_cast<f32x4 *>(patch + loff[k]) = pv[k];
    };
    float bias_v[NT];
    load_bias<COUT, NT>(bias, 0, p, bias_v);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int w = wr.begin, chunk = 0;
    decode_plan(w);
    cur_b = nxt_b; cur_y0 = nxt_y0; cur_x0 = nxt_x0;
    load_chunk(0);
    const int Hout = Hin / 2;
#pragma unroll 1
    for (;;) {
        // LDS accesses of one wave execute in order: the reads of the previous step are done before these writes
        store_chunk();
        const bool last_chunk = (chunk == NCHUNK - 1);
        const bool has_next = w + wr.step < wr.end;
        if (!last_chunk) {
            load_chunk(chunk + 1);
        } else if (has_next) {
            decode_plan(w + wr.step);
            load_chunk(0);
        }
        const float *wc = wl + chunk * KROWS * NPADW;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            float a[MT], bw[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = patch[a_base + koff[s] + m * 2 * RW];
#pragma unroll
            for (int n = 0; n < NT; ++n) bw[n] = wc[b_base + s * 4 * NPADW + n * 16];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bw[n], acc[m][n], 0, 0, 0);
        }
        if (last_chunk) {
            conv_epilogue<COUT, false, MT, NT>(acc, bias_v, out, cur_b, 0, cur_y0, cur_x0, wave, p, q, Hout, Hout);
            if (!has_next) break;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            w += wr.step;
            cur_b = nxt_b; cur_y0 = nxt_y0; cur_x0 = nxt_x0;
            chunk = 0;
        } else {
            ++chunk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split-K GEMM  slab[z][M][N] = A[M][k0:k1] * B[k0:k1][N]      (A row-major lda, B row-major ldb = N)
//   block 256 threads = 2x2 waves, block tile 64 x 64, wave tile 32 x 32 (2 x 2 MFMA tiles), BK = 32
// ------------------------------------------------------------------------------------------------
constexpr int GBM = 64, GBN = 64, GBK = 32, GLDA = 34, GLDB = 80;

__global__ __launch_bounds__(256) void gemm_mfma(const float *__restrict__ A, int lda,
                                                 const float *__restrict__ Bm, int N,
                                                 float *__restrict__ slab, int M, int kper)
{
    __shared__ __attribute__((aligned(16))) float As[GBM * GLDA];
    __shared__ __attribute__((aligned(16))) float Bs[GBK * GLDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * GBN, m0 = blockIdx.y * GBM;
    const int k0 = blockIdx.z * kper;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging assignments: A tile 64 rows x 32 k as float2 (1024 float2 -> 4 per thread)
    //                      B tile 32 k x 64 n as float4 (512 float4 -> 2 per thread)
    for (int k = k0; k < k0 + kper; k += GBK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            const int r = e >> 4, c2 = e & 15;
            int gm = m0 + r;
            gm = gm < M ? gm : M - 1;
            const float2 v = *reinterpret_cast<const float2 *>(A + (long)gm * lda + k + c2 * 2);
            *reinterpret_cast<float2 *>(&As[r * GLDA + c2 * 2]) = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            const int r = e >> 4, c4 = e & 15;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Bm + (long)(k + r) * N + n0 + c4 * 4);
            *reinterpret_cast<f32x4 *>(&Bs[r * GLDB + c4 * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GBK / 4; ++s) {
            float a[2], bw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + p) * GLDA + s * 4 + q];
#pragma unroll
            for (int j = 0; j < 2; ++j) bw[j] = Bs[(s * 4 + q) * GLDB + wn * 32 + j * 16 + p];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bw[j], acc[i][j], 0, 0, 0);
        }
    }
    float *dst = slab + (long)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + q * 4 + r, n = n0 + wn * 32 + j * 16 + p;
                if (m < M) dst[(long)m * N + n] = acc[i][j][r];
            }
}

// out[m][n] = act(bias[n] + sum_z slab[z][m][n]), n < Nout (slab rows are N wide), fixed z order
__global__ void reduce_bias_act(const float *__restrict__ slab, int S, int M, int N, int Nout,
                                const float *__restrict__ bias, int sigmoid, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * Nout) return;
    const int m = i / Nout, n = i % Nout;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slab[((long)z * M + m) * N + n];
    s += bias[n];
    if (sigmoid) s = 1.0f / (1.0f + expf(-s));
    out[(long)m * Nout + n] = s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct ConvPlan { int cch, nt, ngroups; };
static const ConvPlan kPlan[8] = {{5, 2, 1}, {5, 3, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 2}};

constexpr int kChunkA = 16;       // tile-forwards per launch for conv blocks 0-2 (activations fit the Infinity Cache)
constexpr int kChunkB = 32;       // tile-forwards per launch for conv blocks 3-4 (64x64 maps: 1024 workgroups per launch)
constexpr int kFc1Split = 8, kFc2Split = 4, kFc3Split = 4;

}  // namespace

struct axt_detector {
    int max_batch = 0;
    float *d_wconv[8] = {};     // packed conv weights
    float *d_bconv[8] = {};     // folded bias
    float *d_wfc[3] = {};       // [K][Npad]
    float *d_bfc[3] = {};
    float *d_act[8] = {};       // activations after conv block i (chunk-sized for i < 4)
    float *d_slab = nullptr, *d_fc1 = nullptr, *d_fc2 = nullptr;
    float *d_zeros = nullptr;   // 64 bytes of zeros: the address out-of-image lanes load from
    size_t bytes = 0;
    // optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
    bool profiling = false;
    struct Span { hipEvent_t a, b; int kernel; int items; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
};

namespace {

hipEvent_t take_event(axt_detector *d)
{
    if (!d->free_events.empty()) {
        hipEvent_t e = d->free_events.back();
        d->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// RAII: records an event before and after one kernel launch when profiling is on
struct ProfSpan {
    axt_detector *d;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int kernel, items;
    ProfSpan(axt_detector *d_, hipStream_t st_, int kernel_, int items_) : d(d_), st(st_), kernel(kernel_), items(items_)
    {
        if (!d->profiling) return;
        a = take_event(d);
        b = take_event(d);
        (void)hipEventRecord(a, st);
    }
    ~ProfSpan()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        d->spans.push_back({a, b, kernel, items});
    }
};

template <typename T>
int dev_alloc(axt_detector *d, T **p, size_t n)
{
    if (hipMalloc((void **)p, n * sizeof(T)) != hipSuccess) {
        axt_set_error("hipMalloc of %zu bytes failed", n * sizeof(T));
        return AXT_ENOMEM;
    }
    d->bytes += n * sizeof(T);
    return AXT_OK;
}

// Packs one conv block: folds BN (f64), lays weights out as [group][chunk][krow][NPADW].
// krow order inside a chunk, stride-1 layers: ((ky*3+kx) * CCH/4 + cg) * 4 + kk  <->  channel chunk*CCH + cg*4 + kk
// stride-2 layers (conv3x3_s2_mfma): krow = c*9 + ky*3 + kx with c the channel inside the chunk (45 rows, padded to 48).
void pack_conv(int li, const float *w, const float *b, const float *gamma, const float *beta,
               const float *mean, const float *var, std::vector<float> &wp, std::vector<float> &bp)
{
    const ConvSpec &cs = kConv[li];
    const ConvPlan &pl = kPlan[li];
    const int NPADW = npadw(pl.nt);
    const bool first = cs.stride == 2;          // table-ordered K
    const int nchunk = cs.cin / pl.cch;
    const int krows = first ? ((9 * pl.cch + 3) / 4) * 4 : 9 * pl.cch;
    wp.assign((size_t)pl.ngroups * nchunk * krows * NPADW, 0.f);
    bp.assign(cs.cout, 0.f);
    for (int co = 0; co < cs.cout; ++co) {
        const double sc = (double)gamma[co] / sqrt((double)var[co] + 1e-5);
        bp[co] = (float)(((double)b[co] - (double)mean[co]) * sc + (double)beta[co]);
        const int grp = co / (pl.nt * 16), col = co % (pl.nt * 16);
        for (int ci = 0; ci < cs.cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = (float)((double)w[(((size_t)co * cs.cin + ci) * 3 + ky) * 3 + kx] * sc);
                    int chunk, krow;
                    if (first) {
                        chunk = ci / pl.cch;
                        krow = (ci % pl.cch) * 9 + ky * 3 + kx;
                    } else {
                        chunk = ci / pl.cch;
                        const int c = ci % pl.cch;
                        krow = ((ky * 3 + kx) * (pl.cch / 4) + c / 4) * 4 + c % 4;
                    }
                    wp[(((size_t)grp * nchunk + chunk) * krows + krow) * NPADW + col] = v;
                }
    }
}

// persistent grids: a multiple of 8 (one slice per XCD), at most `per_cu` workgroups per CU
int persistent_grid(int nwork, int per_cu)
{
    int g = 256 * per_cu;
    if (g > nwork) g = nwork;
    g = ((g + 7) / 8) * 8;
    return g;
}

template <int CIN, int COUT, bool POOL, int CCH, int NT>
int launch_conv(const float *in, const float *w, const float *bias, float *out, int Hin, int ngroups, int B,
                hipStream_t st)
{
    auto kern = conv3x3_mfma<CIN, COUT, POOL, CCH, NT>;
    using G = GeoS1<CCH>;
    constexpr size_t lds = (size_t)(((CCH * G::PLANE + 3) & ~3) + G::KROWS * npadw(NT)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    AXT_REQUIRE(Hin % 16 == 0, "conv: map size %d not a multiple of the 16x16 tile", Hin);
    const int nwork = (Hin / 16) * (Hin / 16) * ngroups * B;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), lds, st, in, w, bias, out, Hin, ngroups, B);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

template <int CIN, int COUT, int NPC, int NT, bool FIRST, int WPS>
int launch_conv_s2(const float *in, const float *w, const float *bias, float *out, const float *zeros, int Hin, int B,
                   hipStream_t st, int Hf = 0, int Wf = 0, int t0 = 0, int tstep = 1, int item0 = 0, int n_tiles = 1,
                   const TileList *tl = nullptr)
{
    auto kern = conv3x3_s2_mfma<CIN, COUT, NPC, NT, FIRST, WPS>;
    constexpr int KROWS = ((9 * NPC + 3) / 4) * 4;
    constexpr size_t lds = (size_t)(4 * (NPC * 9 * 36 + 4) + (CIN / NPC) * KROWS * npadw(NT)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AXT_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    AXT_REQUIRE(Hin % 32 == 0, "conv: map size %d not a multiple of the tile", Hin);
    TileList dummy;
    dummy.n = 0;
    const int nwork = (Hin / 32) * (Hin / 32) * B;
    static int per_cu = 0;                      // resident workgroups per CU for this kernel (registers + LDS)
    if (per_cu == 0) {
        int n = 0;
        AXT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kern, 256, lds));
        per_cu = n < 1 ? 1 : (n > 8 ? 8 : n);
    }
    hipLaunchKernelGGL(kern, dim3(persistent_grid(nwork, per_cu)), dim3(256), lds, st, in, w, bias, out, zeros, Hin, B, Hf, Wf,
                       t0, tstep, item0, n_tiles, tl ? *tl : dummy);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

int launch_gemm(const float *A, int lda, const float *Bm, int N, float *slab, int M, int K, int split,
                hipStream_t st)
{
    AXT_REQUIRE(N % GBN == 0 && K % (split * GBK) == 0, "gemm: bad shape N=%d K=%d split=%d", N, K, split);
    dim3 grid(N / GBN, axt_cdiv(M, GBM), split);
    hipLaunchKernelGGL(gemm_mfma, grid, dim3(256), 0, st, A, lda, Bm, N, slab, M, K / split);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

int launch_reduce(const float *slab, int S, int M, int N, int Nout, const float *bias, int sig, float *out,
                  hipStream_t st)
{
    const long n = (long)M * Nout;
    hipLaunchKernelGGL(reduce_bias_act, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, S, M, N, Nout,
                       bias, sig, out);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

// conv blocks 0..2 for up to kChunkA items: frames -> d_act[2] slot `slot0`
int run_front_a(axt_detector *d, const float *frames, int Hf, int Wf, int t0, int tstep, int item0, int n_tiles,
                const TileList &tl, int nb, int slot0, hipStream_t st)
{
    int rc;
    {
        ProfSpan ps(d, st, 0, nb);
        if ((rc = launch_conv_s2<5, 20, 5, 2, true, 4>(frames, d->d_wconv[0], d->d_bconv[0], d->d_act[0], d->d_zeros, 512, nb, st, Hf, Wf,
                                                    t0, tstep, item0, n_tiles, &tl))) return rc;
    }
    {
        ProfSpan ps(d, st, 1, nb);
        if ((rc = launch_conv_s2<20, 40, 5, 3, false, 3>(d->d_act[0], d->d_wconv[1], d->d_bconv[1], d->d_act[1], d->d_zeros, 256, nb,
                                                      st))) return rc;
    }
    {
        ProfSpan ps(d, st, 2, nb);
        if ((rc = launch_conv<40, 80, true, 8, 5>(d->d_act[1], d->d_wconv[2], d->d_bconv[2],
                                                               d->d_act[2] + (size_t)slot0 * 80 * 64 * 64, 128, 1, nb,
                                                               st))) return rc;
    }
    return AXT_OK;
}

// conv blocks 3..4 for up to kChunkB items: d_act[2] -> act4_out
int run_front_b(axt_detector *d, int nb, float *act4_out, hipStream_t st)
{
    int rc;
    {
        ProfSpan ps(d, st, 3, nb);
        if ((rc = launch_conv<80, 80, false, 8, 5>(d->d_act[2], d->d_wconv[3], d->d_bconv[3], d->d_act[3],
                                                                64, 1, nb, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 4, nb);
        if ((rc = launch_conv<80, 80, true, 8, 5>(d->d_act[3], d->d_wconv[4], d->d_bconv[4], act4_out, 64,
                                                               1, nb, st))) return rc;
    }
    return AXT_OK;
}

// layers 5..7 + the three linear layers for nb items whose block-4 output is in d_act[4]
int run_back(axt_detector *d, int nb, float *d_yolo, hipStream_t st)
{
    int rc;
    {
        ProfSpan ps(d, st, 5, nb);
        if ((rc = launch_conv<80, 80, false, 8, 5>(d->d_act[4], d->d_wconv[5], d->d_bconv[5], d->d_act[5],
                                                                32, 1, nb, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 6, nb);
        if ((rc = launch_conv<80, 80, true, 8, 5>(d->d_act[5], d->d_wconv[6], d->d_bconv[6], d->d_act[6],
                                                               32, 1, nb, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 7, nb);
        if ((rc = launch_conv<80, 160, false, 8, 5>(d->d_act[6], d->d_wconv[7], d->d_bconv[7], d->d_act[7],
                                                                 16, 2, nb, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 8, nb);
        if ((rc = launch_gemm(d->d_act[7], kFeat, d->d_wfc[0], kFc, d->d_slab, nb, kFeat, kFc1Split, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 9, nb);
        if ((rc = launch_reduce(d->d_slab, kFc1Split, nb, kFc, kFc, d->d_bfc[0], 1, d->d_fc1, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 10, nb);
        if ((rc = launch_gemm(d->d_fc1, kFc, d->d_wfc[1], kFc, d->d_slab, nb, kFc, kFc2Split, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 11, nb);
        if ((rc = launch_reduce(d->d_slab, kFc2Split, nb, kFc, kFc, d->d_bfc[1], 1, d->d_fc2, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 12, nb);
        if ((rc = launch_gemm(d->d_fc2, kFc, d->d_wfc[2], kOutPad, d->d_slab, nb, kFc, kFc3Split, st))) return rc;
    }
    {
        ProfSpan ps(d, st, 13, nb);
        if ((rc = launch_reduce(d->d_slab, kFc3Split, nb, kOutPad, kOut, d->d_bfc[2], 0, d_yolo, st))) return rc;
    }
    return AXT_OK;
}

int forward_items(axt_detector *d, const float *frames, int Hf, int Wf, int t0, int tstep, int n_items, int n_tiles,
                  const TileList &tl, float *d_yolo, hipStream_t st)
{
    for (int base = 0; base < n_items; base += d->max_batch) {
        const int nb = (n_items - base < d->max_batch) ? n_items - base : d->max_batch;
        for (int cb = 0; cb < nb; cb += kChunkB) {
            const int nbb = (nb - cb < kChunkB) ? nb - cb : kChunkB;
            for (int c = 0; c < nbb; c += kChunkA) {
                const int nc = (nbb - c < kChunkA) ? nbb - c : kChunkA;
                const int rc = run_front_a(d, frames, Hf, Wf, t0, tstep, base + cb + c, n_tiles, tl, nc, c, st);
                if (rc) return rc;
            }
            const int rc = run_front_b(d, nbb, d->d_act[4] + (size_t)cb * 80 * 32 * 32, st);
            if (rc) return rc;
        }
        const int rc = run_back(d, nb, d_yolo + (size_t)base * kOut, st);
        if (rc) return rc;
    }
    return AXT_OK;
}

}  // namespace

extern "C" {

double axt_cnn_flops_per_tile(void)
{
    double f = 0;
    for (const ConvSpec &c : kConv) {
        const double ho = c.hin / c.stride;
        f += 2.0 * ho * ho * c.cout * c.cin * 9;
    }
    f += 2.0 * kFeat * kFc + 2.0 * kFc * kFc + 2.0 * kFc * kOut;
    return f;
}

int axt_detector_create(const float *const *h_tensors, int n_tensors, int max_batch, axt_detector **out)
{
    AXT_REQUIRE(h_tensors && out, "null argument");
    AXT_REQUIRE(n_tensors == AXT_N_WEIGHT_TENSORS, "expected %d tensors, got %d", AXT_N_WEIGHT_TENSORS, n_tensors);
    AXT_REQUIRE(max_batch >= 1 && max_batch <= 65535, "max_batch %d out of range", max_batch);
    for (int i = 0; i < n_tensors; ++i) AXT_REQUIRE(h_tensors[i] != nullptr, "tensor %d is null", i);
    axt_detector *d = new (std::nothrow) axt_detector();
    if (!d) return AXT_ENOMEM;
    d->max_batch = max_batch;
    int rc = AXT_OK;
    std::vector<float> wp, bp;
    for (int li = 0; li < 8 && !rc; ++li) {
        const float *const *t = h_tensors + li * 6;
        pack_conv(li, t[0], t[1], t[2], t[3], t[4], t[5], wp, bp);
        if ((rc = dev_alloc(d, &d->d_wconv[li], wp.size()))) break;
        if ((rc = dev_alloc(d, &d->d_bconv[li], bp.size()))) break;
        if (hipMemcpy(d->d_wconv[li], wp.data(), wp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d->d_bconv[li], bp.data(), bp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            axt_set_error("weight upload failed (conv block %d)", li);
            rc = AXT_EHIP;
        }
    }
    // linear layers: reference stores [out,in]; the GEMM wants [in][out_padded]
    const int fin[3] = {kFeat, kFc, kFc}, fout[3] = {kFc, kFc, kOut}, fpad[3] = {kFc, kFc, kOutPad};
    for (int l = 0; l < 3 && !rc; ++l) {
        const float *w = h_tensors[48 + 2 * l], *b = h_tensors[48 + 2 * l + 1];
        std::vector<float> wt((size_t)fin[l] * fpad[l], 0.f);
        for (int o = 0; o < fout[l]; ++o)
            for (int i = 0; i < fin[l]; ++i) wt[(size_t)i * fpad[l] + o] = w[(size_t)o * fin[l] + i];
        if ((rc = dev_alloc(d, &d->d_wfc[l], wt.size()))) break;
        if ((rc = dev_alloc(d, &d->d_bfc[l], (size_t)fout[l]))) break;
        if (hipMemcpy(d->d_wfc[l], wt.data(), wt.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d->d_bfc[l], b, (size_t)fout[l] * 4, hipMemcpyHostToDevice) != hipSuccess) {
            axt_set_error("weight upload failed (linear %d)", l);
            rc = AXT_EHIP;
        }
    }
    // activations
    const size_t per_item[8] = {20u * 256 * 256, 40u * 128 * 128, 80u * 64 * 64, 80u * 64 * 64,
                                80u * 32 * 32,   80u * 32 * 32,   80u * 16 * 16, 160u * 16 * 16};
    for (int i = 0; i < 8 && !rc; ++i)
        rc = dev_alloc(d, &d->d_act[i], per_item[i] * (size_t)(i < 2 ? kChunkA : i < 4 ? kChunkB : max_batch));
    if (!rc) rc = dev_alloc(d, &d->d_slab, (size_t)kFc1Split * max_batch * kFc);
    if (!rc) rc = dev_alloc(d, &d->d_fc1, (size_t)max_batch * kFc);
    if (!rc) rc = dev_alloc(d, &d->d_fc2, (size_t)max_batch * kFc);
    if (!rc) rc = dev_alloc(d, &d->d_zeros, (size_t)16);
    if (!rc && hipMemset(d->d_zeros, 0, 64) != hipSuccess) {
        axt_set_error("hipMemset failed");
        rc = AXT_EHIP;
    }
    if (!rc && hipDeviceSynchronize() != hipSuccess) {
        axt_set_error("device synchronize failed after upload");
        rc = AXT_EHIP;
    }
    if (rc) {
        axt_detector_destroy(d);
        return rc;
    }
    *out = d;
    return AXT_OK;
}

void axt_detector_destroy(axt_detector *d)
{
    if (!d) return;
    for (int i = 0; i < 8; ++i) {
        (void)hipFree(d->d_wconv[i]);
        (void)hipFree(d->d_bconv[i]);
        (void)hipFree(d->d_act[i]);
    }
    for (int i = 0; i < 3; ++i) {
        (void)hipFree(d->d_wfc[i]);
        (void)hipFree(d->d_bfc[i]);
    }
    (void)hipFree(d->d_slab);
    (void)hipFree(d->d_fc1);
    (void)hipFree(d->d_fc2);
    (void)hipFree(d->d_zeros);
    for (auto &sp : d->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (hipEvent_t e : d->free_events) (void)hipEventDestroy(e);
    delete d;
}

size_t axt_detector_device_bytes(const axt_detector *d) { return d ? d->bytes : 0; }

int axt_detector_set_profiling(axt_detector *d, int on)
{
    AXT_REQUIRE(d, "null argument");
    d->profiling = on != 0;
    return AXT_OK;
}

int axt_detector_read_profile(axt_detector *d, double *ms, int64_t *launches, int64_t *items, int n)
{
    AXT_REQUIRE(d && ms && launches && items && n >= AXT_N_CNN_KERNELS, "bad argument");
    for (int i = 0; i < n; ++i) { ms[i] = 0; launches[i] = 0; items[i] = 0; }
    for (auto &sp : d->spans) {
        AXT_CHECK_HIP(hipEventSynchronize(sp.b));
        float t = 0.f;
        AXT_CHECK_HIP(hipEventElapsedTime(&t, sp.a, sp.b));
        ms[sp.kernel] += t;
        launches[sp.kernel] += 1;
        items[sp.kernel] += sp.items;
        d->free_events.push_back(sp.a);
        d->free_events.push_back(sp.b);
    }
    d->spans.clear();
    return AXT_OK;
}

double axt_cnn_kernel_flops_per_tile(int kernel)
{
    if (kernel < 0 || kernel >= AXT_N_CNN_KERNELS) return 0;
    if (kernel < 8) {
        const ConvSpec &c = kConv[kernel];
        const double ho = c.hin / c.stride;
        return 2.0 * ho * ho * c.cout * c.cin * 9;
    }
    const double f[6] = {2.0 * kFeat * kFc, (double)kFc1Split * kFc, 2.0 * kFc * kFc, (double)kFc2Split * kFc,
                         2.0 * kFc * kOut, (double)kFc3Split * kOut};
    return f[kernel - 8];
}

int axt_cnn_forward(axt_detector *det, const float *d_x, int B, float *d_yolo, void *stream)
{
    AXT_REQUIRE(det && d_x && d_yolo, "null argument");
    AXT_REQUIRE(B >= 0, "negative batch");
    if (B == 0) return AXT_OK;
    // X[B,5,512,512] is a timelapse of 5*B frames of 512x512 in which item b reads frames 5b..5b+4:
    // the frames path with a frame step of 5 and a single tile at the origin.
    TileList tl;
    tl.n = 1;
    tl.yx[0] = 0;
    tl.yx[1] = 0;
    return forward_items(det, d_x, AXT_TILE, AXT_TILE, 0, AXT_IN_CH, B, 1, tl, d_yolo, (hipStream_t)stream);
}

int axt_cnn_forward_frames(axt_detector *det, const float *d_frames, int T_all, int H, int W, int t0, int n_frames,
                           const int32_t *h_tile_yx, int n_tiles, float *d_yolo, void *stream)
{
    AXT_REQUIRE(det && d_frames && d_yolo && h_tile_yx, "null argument");
    AXT_REQUIRE(n_tiles >= 1 && n_tiles <= 256, "n_tiles %d out of range [1,256]", n_tiles);
    AXT_REQUIRE(t0 >= 0 && n_frames >= 0 && t0 + n_frames + 4 <= T_all, "frames [%d,%d) + context exceed T_all=%d",
                t0, t0 + n_frames, T_all);
    TileList tl;
    tl.n = n_tiles;
    for (int k = 0; k < n_tiles; ++k) {
        const int ty = h_tile_yx[2 * k], tx = h_tile_yx[2 * k + 1];
        AXT_REQUIRE(ty >= 0 && tx >= 0 && ty * AXT_TILE < H && tx * AXT_TILE < W, "tile %d (%d,%d) outside %dx%d", k,
                    ty, tx, H, W);
        tl.yx[2 * k] = (short)ty;
        tl.yx[2 * k + 1] = (short)tx;
    }
    if (n_frames == 0) return AXT_OK;
    return forward_items(det, d_frames, H, W, t0, 1, n_frames * n_tiles, n_tiles, tl, d_yolo, (hipStream_t)stream);
}

}  // extern "C"
