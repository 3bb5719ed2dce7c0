// Detector forward pass for gfx950 (MI355X): YOLO_AXTrack (reference axtrack/machinelearning/model.py:20-125)
// as f32-in / f32-accumulate MFMA kernels.
//
//   conv3x3_mfma     stride-1 implicit-GEMM 3x3 convolution + folded BatchNorm + LeakyReLU(0.1) (+ fused
//                    MaxPool2d(2,2)) for conv blocks 2..10: the input patch and a K-chunk of the weights are staged
//                    in LDS, v_mfma_f32_16x16x4_f32 accumulates 16 pixels x 16 channels per instruction, the next
//                    chunk's global loads are prefetched into registers behind the MFMAs.
//   conv3x3_s2_k1    the two stride-2 blocks (HBM-bound) on v_mfma_f32_4x4x1_16b_f32: persistent workgroups, every
//                    wave its own barrier-free pipeline over a wave-private LDS patch, weights LDS-resident, global
//                    traffic through buffer loads / stores (zero padding from the range check). Block 0 reads the
//                    5-frame temporal stack straight from the timelapse (fuses Timelapse.get_frametiles_stack,
//                    Timelapse.py:111-125,150-157).
//   conv3x3_wino     the same stride-1 blocks as Winograd F(2x2,3x3) on the same f32 MFMA: 16/36 of the multiplications,
//                    every operation f32 -- the DEFAULT for conv blocks 2,4,5,7,8,10 (output channels in groups of 80;
//                    conv3x3_mfma stays selectable: axt_detector_set_arith).
//   gemm_mfma        split-K GEMM for the three linear layers, partial slabs reduced in a fixed order
//                    (bit-reproducible) by reduce_bias_act (+ Sigmoid).
//
// Numerics: f32 MFMA on gfx950 is a k-ordered chain of f32 FMAs (exact f32, no reduced precision).
// BatchNorm is folded in f64 at pack time and rounded once to f32.
#include "axt_common.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
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

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;       // byte offset beyond every descriptor's range: the load returns 0
constexpr int kBufRecords = 0x7fffffff;


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

template <int CIN, int COUT, bool POOL, int CCH, int NT, int PF = 1>
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
    const int cstride = Hin * Hin;
    const int Hout = POOL ? Hin / 2 : Hin;

    // staging plan: element e = tid + k*256 of the patch is (channel c, row r, col). Branch-free: per element one
    // word, (offset inside the chunk << 12) | LDS offset; elements past the patch go to a dummy LDS word, elements
    // outside the image carry the offset kOut and are loaded with an out-of-range buffer offset (the range check
    // returns the zero padding). Both streams are buffer loads: descriptor bases and chunk offsets are scalars, so a
    // chunk costs no 64-bit address arithmetic. Per chunk all loads are issued back to back into registers and
    // written to LDS after the barrier, while the next chunk's loads are already in flight behind the MFMAs
    // (async-STAGE split).
    constexpr int NPE = (CCH * PH * PW + 255) / 256;
    constexpr int NW4 = KROWS * NPADW / 4;
    constexpr int NWE = (NW4 + 255) / 256;
    constexpr int DUMMY = CCH * PLANE - 1;       // last (padding) word of the last plane, never read by the MFMAs
    static_assert(PLANE > PH * PW, "the plane padding provides the dummy word");
    static_assert(CCH * PLANE <= 4096, "LDS offsets are packed into 12 bits");
    constexpr int kOut = (1 << 20) - 1;
    int plan[NPE];
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
        const int e = tid + k * 256;
        const int c = e / (PH * PW), rem = e - c * (PH * PW);
        const int r = rem / PW, col = rem - r * PW;
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        const bool in_patch = e < CCH * PH * PW;
        const bool ok = in_patch && gy >= 0 && gy < Hin && gx >= 0 && gx < Hin;
        const int g = c * cstride + gy * Hin + gx;
        plan[k] = ((ok ? g : kOut) << 12) | (in_patch ? c * PLANE + r * PW + col : DUMMY);
    }
    float pv[NPE];
    f32x4 wv[NWE];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in) + (long)b * CIN * cstride, 0, kBufRecords, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wpk) + (long)grp * NCHUNK * KROWS * NPADW, 0, kBufRecords, 0x00020000);
    auto load_chunk = [&](int chunk) {
        const int isoff = chunk * CCH * cstride * 4;
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
            const unsigned g = (unsigned)plan[k] >> 12;
            pv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(g == kOut ? kOobOffset : g * 4u), isoff, 0));
        }
        const int wsoff = chunk * KROWS * NPADW * 4;
#pragma unroll
        for (int k = 0; k < NWE; ++k) {
            const int e = tid + k * 256;
            wv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  w_rsrc, (NW4 % 256 == 0 || e < NW4) ? e * 16 : (int)kOobOffset, wsoff, 0));
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < NPE; ++k) patch[plan[k] & 4095] = pv[k];
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
        // operands of k-step s + PF are fetched from LDS while the MFMAs of k-step s issue (register ring of PF + 1
        // sets); the group barriers pin that interleave (left alone the scheduler sinks the reads to their first use)
        float a[PF + 1][MT], bw[PF + 1][NT];
        auto fetch = [&](int s) {
            constexpr int CG = CCH / 4;
            const int kk = s / CG, cg = s % CG;      // (ky,kx) major, channel group minor
            const int ky = kk / 3, kx = kk % 3, slot = s % (PF + 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) a[slot][m] = patch[a_base + cg * 4 * PLANE + (m + ky) * PW + kx];
#pragma unroll
            for (int n = 0; n < NT; ++n) bw[slot][n] = wl[b_base + s * 4 * NPADW + n * 16];
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) fetch(s);
        __builtin_amdgcn_sched_group_barrier(0x100, PF * (MT + NT), 0);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if (s + PF < KSTEPS) fetch(s + PF);
            const int slot = s % (PF + 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[slot][m], bw[slot][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
        }
    }
    conv_epilogue<COUT, POOL, MT, NT>(acc, bias_v, out, b, grp, y0, x0, wave, p, q, Hout, Hout);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_bf16x3: the stride-1 blocks with 80 output channels on the bf16 matrix pipe -- OPT-IN arithmetic
// (parameters['CNN_ARITH'] = 'bf16x3'; the default and every headline number stay on the f32 kernels above).
//   Every f32 operand is split exactly into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 mantissa bits), and a product
//   x*w is taken as the six partial products of weight >= 2^-16: hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi. Each
//   bf16 x bf16 product is exact in f32 and the accumulation is f32, so one product carries a relative error of about
//   3 * 2^-24 -- the size of an f32 rounding -- at 6/16 of the f32 pipe's cycles per multiply-add.
//   Shape and occupancy (profiles/micro/mfma_bf16_shapes.hip, random operands): v_mfma_f32_16x16x32_bf16 fits 80 = 5 x 16
//   channels exactly, but ONE wave per SIMD issues it only every 27.5 cycles (1.41 PFLOP/s); two waves per SIMD reach
//   17.7 cycles (1.84 PFLOP/s at the 2.0 GHz the chip holds under that load). The 32x32x16 shape issues every 33 cycles
//   from one wave (1.85 PFLOP/s) but pads 80 channels to 96 and pulls the clock to 1.86 GHz for the whole kernel: built
//   and measured slower than this one. Hence 8 waves per workgroup, one workgroup per CU.
//   Tile: 32 x 16 output pixels per workgroup, wave w owns rows 4w..4w+3 and all 80 channels (4 x 5 accumulator tiles of
//   16 x 16). K runs in chunks of 16 input channels. One MFMA k-step (32 k) = four "pairs" of (tap, 8-channel half):
//   lane (p, q) holds pixel p's 8 channels of pair q as ONE 16-byte LDS read. A chunk has 9 taps x 2 halves = 18 pairs
//   = 4.5 k-steps, padded to 5 with zero weights (10 %).
//   LDS (134 KB): patch [plane hi|mid|lo][half][34 rows][18 cols][8 ch] bf16, halves padded to a multiple of 256 B so that
//   the two halves a 16-lane read group touches fall on disjoint banks; weights [k-step][plane][q][80 n][8 k] bf16, copied
//   verbatim from the packed image (pack_bf16x3). The next chunk's global loads (24 + 40 registers) wait in registers
//   behind the MFMAs; the f32 -> 3 x bf16 split of the activations happens on the way from those registers into LDS
//   (v_cvt_pk_bf16_f32, round to nearest even; the residuals are exact in f32).
//   Input and output stay f32 NCHW, so the layer is interchangeable with conv3x3_mfma block by block.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct GeoB3 {
    static constexpr int CCH = 16, MT = 4, NT = 5, TH = 32, TW = 16, PH = TH + 2, PW = TW + 2, KS = 5, WAVES = 8;
    static constexpr int HALF_B = ((PH * PW * 16 + 255) / 256) * 256;      // 9984 bytes
    static constexpr int PLANE_B = 2 * HALF_B;
    static constexpr int PATCH_B = 3 * PLANE_B;                             // 59904
    static constexpr int WQ_B = NT * 16 * 16;                               // one (k-step, plane, q) block: 80 n x 16 B
    static constexpr int WSTEP_B = 3 * 4 * WQ_B;
    static constexpr int WCHUNK_B = KS * WSTEP_B;                           // 76800
    static constexpr int LDS_B = PATCH_B + WCHUNK_B;                        // 136704
    static constexpr int DUMMY = PH * PW * 16;                              // first padding byte of half 0: never read
};

// x[0..7] -> the three bf16 planes, 8 values each (exact: x = hi + mid + lo unless lo underflows)
__device__ __forceinline__ void split_bf16x3(const float (&x)[8], u32x4 &H, u32x4 &M, u32x4 &L)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const bf16x2 h = {(__bf16)a, (__bf16)b};
        const float ra = a - (float)h[0], rb = b - (float)h[1];
        const bf16x2 m = {(__bf16)ra, (__bf16)rb};
        const float sa = ra - (float)m[0], sb = rb - (float)m[1];
        const bf16x2 l = {(__bf16)sa, (__bf16)sb};
        H[i] = __builtin_bit_cast(unsigned, h);
        M[i] = __builtin_bit_cast(unsigned, m);
        L[i] = __builtin_bit_cast(unsigned, l);
    }
}

template <int CIN, bool POOL>
__global__ __launch_bounds__(512, 1) void conv3x3_bf16x3(
    const float *__restrict__ in,       // activations [B,CIN,Hin,Hin] f32
    const unsigned *__restrict__ wpk,   // packed weights: per chunk the LDS image [KS][3][4][80][8] bf16
    const float *__restrict__ bias,     // folded bias [80]
    float *__restrict__ out,            // [B,80,Hout,Hout] f32
    int Hin, int B)
{
    using G = GeoB3;
    constexpr int COUT = 80, MT = G::MT, NT = G::NT, PH = G::PH, PW = G::PW, NTHR = 64 * G::WAVES;
    constexpr int NCHUNK = (CIN + G::CCH - 1) / G::CCH;

    extern __shared__ __attribute__((aligned(256))) unsigned char smem_b[];
    unsigned char *patch = smem_b;
    unsigned char *wl = smem_b + G::PATCH_B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int tiles_x = Hin / G::TW, ntile = tiles_x * (Hin / G::TH);
    int w;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    }
    const int tile = w % ntile, b = w / ntile;
    const int y0 = (tile / tiles_x) * G::TH, x0 = (tile % tiles_x) * G::TW;
    const int cstride = Hin * Hin;
    const int Hout = POOL ? Hin / 2 : Hin;

    // staging plan of the patch: item e = tid + k*512 is (half h, row r, col): 8 channels of one pixel. The global side
    // is eight buffer loads (one per channel, coalesced along the row); pixels outside the image carry an
    // out-of-range offset and channels beyond CIN fall behind the descriptor's range: both read as the zero padding.
    constexpr int NITEM = 2 * PH * PW;                       // 1224
    constexpr int NPE = (NITEM + NTHR - 1) / NTHR;           // 3
    unsigned gofs[NPE], lofs[NPE];
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
        const int e = tid + k * NTHR;
        const int h = e / (PH * PW), rem = e - h * (PH * PW);
        const int r = rem / PW, col = rem - r * PW;
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        const bool in_patch = e < NITEM;
        const bool ok = in_patch && gy >= 0 && gy < Hin && gx >= 0 && gx < Hin;
        gofs[k] = ok ? (unsigned)((8 * h * cstride + gy * Hin + gx) * 4) : kOobOffset;
        lofs[k] = in_patch ? (unsigned)(h * G::HALF_B + (r * PW + col) * 16) : (unsigned)G::DUMMY;
    }
    constexpr int NW16 = G::WCHUNK_B / 16;                   // 4800 16-byte elements per chunk
    constexpr int NWE = (NW16 + NTHR - 1) / NTHR;            // 10
    float pv[NPE][8];
    u32x4 wv[NWE];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in) + (long)b * CIN * cstride, 0, CIN * cstride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned *>(wpk), 0, NCHUNK * G::WCHUNK_B, 0x00020000);
    auto load_chunk = [&](int chunk) {
        const unsigned cbase = (unsigned)(chunk * G::CCH * cstride * 4);
#pragma unroll
        for (int k = 0; k < NPE; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned off = gofs[k] == kOobOffset ? kOobOffset : gofs[k] + cbase + (unsigned)(i * cstride * 4);
                pv[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)off, 0, 0));
            }
        const unsigned wbase = (unsigned)(chunk * G::WCHUNK_B);
#pragma unroll
        for (int k = 0; k < NWE; ++k) {
            const int e = tid + k * NTHR;
            wv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, e < NW16 ? (int)(wbase + e * 16) : (int)kOobOffset, 0, 0));
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
            u32x4 H, M, L;
            split_bf16x3(pv[k], H, M, L);
            *reinterpret_cast<u32x4 *>(patch + lofs[k]) = H;
            *reinterpret_cast<u32x4 *>(patch + G::PLANE_B + lofs[k]) = M;
            *reinterpret_cast<u32x4 *>(patch + 2 * G::PLANE_B + lofs[k]) = L;
        }
#pragma unroll
        for (int k = 0; k < NWE; ++k) {
            const int e = tid + k * NTHR;
            if (k + 1 < NWE || e < NW16) *reinterpret_cast<u32x4 *>(wl + e * 16) = wv[k];
        }
    };
    const unsigned a_lane = (unsigned)((q & 1) * G::HALF_B + ((wave * MT) * PW + p) * 16);
    const unsigned b_lane = (unsigned)(q * G::WQ_B + p * 16);

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
        // Operands run ahead of the MFMAs in register double buffers, so that no block starts with a wait: a k-step is done
        // in two halves of two rows each -- block (s, half, n) = 2 rows x 6 partial products = 12 MFMAs; the three weight
        // fragments of the next block are fetched before them, the six pixel fragments of the next half during block
        // (s, half, 1) -- and the group barriers pin that interleave. (With all twelve pixel fragments of a k-step loaded at
        // its head, the two waves of a SIMD, which leave every barrier in step, stalled on them together: matrix pipe
        // busy 62 % of the time.)
        bf16x8 A[2][3][2], Bv[2][3];
        auto load_a = [&](int hidx, int buf) {
            // pairs 4s + q: tap (4s + q) / 2 (the padding pairs of the last step re-read tap 8: finite values, zero weights)
            constexpr int kLast = 8;
            const int s = hidx >> 1, h = hidx & 1;
            const int tap_a = 2 * s < kLast ? 2 * s : kLast, tap_b = 2 * s + 1 < kLast ? 2 * s + 1 : kLast;
            const unsigned off_a = (unsigned)(((tap_a / 3) * PW + tap_a % 3) * 16);
            const unsigned off_b = (unsigned)(((tap_b / 3) * PW + tap_b % 3) * 16);
            const unsigned toff = a_lane + ((q >> 1) ? off_b : off_a) + (unsigned)(2 * h * PW * 16);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    A[buf][pl][m] = *reinterpret_cast<const bf16x8 *>(patch + pl * G::PLANE_B + toff + m * PW * 16);
        };
        auto load_b = [&](int s, int n, int buf) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                Bv[buf][pl] = *reinterpret_cast<const bf16x8 *>(wl + (s * 3 + pl) * 4 * G::WQ_B + b_lane + n * 256);
        };
        load_a(0, 0);
        load_b(0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        constexpr int NBLK = G::KS * 2 * NT;
#pragma unroll
        for (int hidx = 0; hidx < 2 * G::KS; ++hidx) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int blk = hidx * NT + n, cur = blk & 1, ab = hidx & 1, h = hidx & 1;
                const bool more = blk + 1 < NBLK;
                const bool next_a = n == 1 && hidx + 1 < 2 * G::KS;
                if (more) load_b((n + 1 < NT ? hidx : hidx + 1) >> 1, n + 1 < NT ? n + 1 : 0, cur ^ 1);
                if (next_a) load_a(hidx + 1, ab ^ 1);
                // the six partial products, small terms first
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][0][m], Bv[cur][2], acc[2 * h + m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][2][m], Bv[cur][0], acc[2 * h + m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][1][m], Bv[cur][1], acc[2 * h + m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][0][m], Bv[cur][1], acc[2 * h + m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][1][m], Bv[cur][0], acc[2 * h + m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[2 * h + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ab][0][m], Bv[cur][0], acc[2 * h + m][n], 0, 0, 0);
                if (more) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                if (next_a) {
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                }
            }
        }
    }
    float bias_v[NT];                     // loaded here: nothing is in flight any more, and five registers fewer live in the loop
    load_bias<COUT, NT>(bias, 0, p, bias_v);
    conv_epilogue<COUT, POOL, MT, NT>(acc, bias_v, out, b, 0, y0, x0, wave, p, q, Hout, Hout);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_wino: the stride-1 blocks (80 output channels, or NG groups of 80) as Winograd F(2x2, 3x3) on the f32 matrix pipe.
//   Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile (Lavin & Gray 2016): 16 multiplies per tile, input and output
//   channel instead of 36 -- the 16 transform positions are 16 independent GEMMs  M_pos[tile][cout] = V_pos[tile][cin] *
//   U_pos[cin][cout]  on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; all arithmetic stays f32: the transforms are
//   additions, the weights G g G^T are formed in f64 and rounded once). Measured on post-activation data the result is as
//   close to an f64 convolution as the direct f32 kernel's.
//   Workgroup = 16 x 16 output pixels (8 x 8 tiles = 4 MFMA row blocks of 16 tiles) x 80 channels, 8 waves, one workgroup
//   per CU, persistent. Wave (m, h): row block m, ALL 16 positions, output-channel blocks {0,1} (h = 0) or {2,3,4}
//   (h = 1): a lane holds every position of its (tile, channel) pairs, so the output transform, bias, LeakyReLU and the
//   2x2 max -- one Winograd tile is one pooled pixel -- never leave its registers. Waves w and w + 4 share a SIMD, so each
//   SIMD carries 2 + 3 channel blocks; the h = 0 waves, with a third less MFMA work, also run the input transform.
//   K runs in chunks of 8 input channels = 2 MFMA k-steps. LDS (144 KB): two buffers of
//     V [pos 16][row block 4][lane 64][k-step 2]   (32 KB)   and   U [pos 16][channel block 5][lane 64][k-step 2]   (40 KB):
//   lane-linear 8-byte elements, one conflict-free ds_read_b64 (256 B/clk) fetches a lane's operand for both k-steps.
//   U is packed in exactly that order (pack_wino), so a chunk is 40 LDS-DMA pieces of 1 KiB (global_load_lds_dwordx4), five
//   per wave, no registers and no ds_write. One barrier per chunk. Behind the MFMAs of every position of chunk g one slice
//   of the loads of chunk g + 1 is issued (issued in one block they hold the wave, and half of its SIMD's MFMA supply, for
//   300-500 cycles per DMA instruction): the input patches first (h = 0 waves; per patch row ONE aligned 8-byte load per
//   lane, the neighbouring columns from the neighbouring lanes by DPP, a sparse load for the halo column at the ends of the
//   16-pixel row), then the DMA pieces of U into the other buffer. After the MFMAs the h = 0 waves transform the patches
//   (32 add/sub per patch) and write V for g + 1. The chunk stream runs across tiles, so a tile's epilogue and stores
//   overlap the next tile's loads. The result of a tile does not depend on the launch shape.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef AXT_WINO_XFORM_PRIO
#define AXT_WINO_XFORM_PRIO 2
#endif
// Diagnostic build only (-DAXT_WINO_STAMPS, profiles/wino_stamps.py): s_memtime stamps at the segment boundaries of a chunk,
// summed per wave in scalar registers and stored once after the loop to a buffer nothing else reads.
#ifdef AXT_WINO_STAMPS
__device__ unsigned long long g_wino_stamps[1024 * 8 * 8];
#define WINO_STAMP(i)                                                                                          \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        st_sum[i] += t_ - st_prev;                                                                             \
        st_prev = t_;                                                                                          \
    } while (0)
#else
#define WINO_STAMP(i)
#endif

struct GeoW {
    static constexpr int CCH = 8;
    static constexpr int VBUF = 16 * 4 * 64 * 2;          // floats per V buffer
    static constexpr int UBUF = 16 * 5 * 64 * 2;          // floats per U buffer = one packed chunk
    static constexpr int LDS_B = 2 * (VBUF + UBUF) * 4;   // 147456
};

template <int CIN, bool POOL, int NTW, bool XFORM, int NG>
__device__ __forceinline__ void wino_body(const float *__restrict__ in, const float *__restrict__ upk,
                                          const float *__restrict__ bias, float *__restrict__ out, int H, int B, float *smem)
{
    using G = GeoW;
    constexpr int NCH = CIN / G::CCH;
    constexpr int N0 = XFORM ? 0 : 2;              // first output-channel block of this wave
    static_assert(CIN % G::CCH == 0 && NCH >= 2, "CIN must be a multiple of 8, at least 16");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
    const int m = wave & 3;
    const int tiles_x = H >> 4, ntile = tiles_x * tiles_x;
    // work item w = ((batch item * NG) + group of 80 output channels) * ntile + tile
    const WorkRange wr = my_work(ntile * B * NG);
    if (wr.begin >= wr.end) return;
    const int cstride = H * H;
    const int Hout = POOL ? H / 2 : H;
    float *Vs = smem, *Us = smem + 2 * G::VBUF;
    auto group_of = [&](int w) { return NG == 1 ? 0 : (w / ntile) % NG; };

    // Which operand is which decides what a lane holds and hence how its results leave:
    //   pooled blocks:   A = V, B = U: D[tile 4q + r][channel p] -- a lane holds one channel and four x-adjacent tiles = four
    //                    adjacent POOLED pixels, one 16-byte store per channel block;
    //   unpooled blocks: A = U, B = V: D[channel 4q + r][tile p] -- a lane holds ONE tile (row 2m + p/8, column p%8 of the
    //                    workgroup's 8 x 8 tiles) and four consecutive channels; the 8 lanes of a tile row own 8 x-adjacent
    //                    tiles = one whole 64-byte line of an output row per channel, so a store instruction writes 8 whole
    //                    lines instead of 64 separate 16-byte pieces.
    // Interleaved A/B on one box (profiles/r04f_ab_stores.log, r04g_*): the second layout is 2 % faster on the unpooled blocks
    // and 2 % slower on the pooled ones (four 4-byte stores per channel block instead of one 16-byte store), so each kind keeps
    // the layout that suits it. Same products in the same k order either way: the results do not depend on it.
    constexpr bool TILE_LANES = !POOL;              // lane = tile, registers = four channels
    f32x4 bias_v[NTW];
    auto load_bias = [&](int grp) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            if constexpr (TILE_LANES) bias_v[n] = *reinterpret_cast<const f32x4 *>(bias + grp * 80 + (N0 + n) * 16 + 4 * q);
            else { const float bv = bias[grp * 80 + (N0 + n) * 16 + p]; bias_v[n] = f32x4{bv, bv, bv, bv}; }
        }
    };
    load_bias(group_of(wr.begin));

    // ---- raw patches (transform waves): fetched one chunk ahead of their use ----
    // A patch row = [left][mid.x mid.y][right] at columns x0 + 2 tx - 1 .. + 2. Every lane loads only its aligned middle
    // pair (8 lanes = 64 contiguous bytes per row and channel); left / right are the neighbouring lanes' pairs (DPP),
    // except at the ends of the 16-pixel row, where one sparse load brings the halo column (zero outside the image).
    // Rows outside the image take the out-of-range offset (zeros).
    float d[1][2][4][3];                            // [set][k-step][row][mid.x, mid.y, halo]
    int vo_mid[4], vo_halo[4];
    __amdgpu_buffer_rsrc_t ld_rsrc;
    const bool first_col = (p & 7) == 0, last_col = (p & 7) == 7;
    int lw = wr.begin, lc = 0;                      // load cursor: tile and chunk of the next patches to fetch
    auto plan_tile = [&](int w) {
        const int tile = w % ntile, b = w / (ntile * NG);
        ld_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in) + (long)b * CIN * cstride, 0, kBufRecords, 0x00020000);
        const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
        const int ty = 2 * m + (p >> 3), tx = p & 7;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = y0 + 2 * ty - 1 + r;
            const bool ok = gy >= 0 && gy < H;
            const int rowb = q * cstride + gy * H;
            vo_mid[r] = ok ? (rowb + x0 + 2 * tx) * 4 : (int)kOobOffset;
            const int hx = first_col ? x0 - 1 : x0 + 16;
            vo_halo[r] = (ok && (first_col || last_col) && hx >= 0 && hx < H) ? (rowb + hx) * 4 : (int)kOobOffset;
        }
    };
    // patch row k of 8 (k-step, row) of the chunk at the load cursor -> set
    auto raw_load = [&](int set, int k) {
        const int s = k / 4, r = k % 4;
        const int soff = (lc * G::CCH + s * 4) * cstride * 4;
        const f32x2 mid = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ld_rsrc, vo_mid[r], soff, 0));
        d[set][s][r][0] = mid.x;
        d[set][s][r][1] = mid.y;
        d[set][s][r][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ld_rsrc, vo_halo[r], soff, 0));
    };
    // the cursor moves on by one chunk; past the last chunk of the last tile it stays (that chunk is fetched again, unused)
    auto advance_cursor = [&]() {
        if (lc + 1 < NCH) ++lc;
        else if (lw + wr.step < wr.end) {
            lw += wr.step;
            lc = 0;
            plan_tile(lw);
        }
    };
    // DMA piece k of this wave's 5 of U chunk c
    // every workgroup starts its round through a chunk's 40 pieces elsewhere: the CUs of an XCD run in step, and without
    // the rotation they all ask the L2 for the same lines at the same moment
    const int rot = (blockIdx.x * 7) % 40;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // The stream goes through a buffer descriptor that covers exactly the packed image (NG * NCH chunks): a chunk index
    // outside it -- which the clamps of step() never produce -- would read zeros instead of leaving the allocation
    // (the patches are range-checked the same way; round 2's experimental builds faulted on exactly such a load).
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(upk), 0, NG * NCH * G::UBUF * 4, 0x00020000);
    // The U stream is issued by the waves with three channel blocks (h = 1) alone, ten pieces each: in-kernel stamps show
    // them waiting ~3 000 cycles per chunk at the barrier for their partners, which carry the transform and the patch
    // loads on top of their MFMAs (profiles/r03c_stamps_*.log) -- a DMA instruction costs its wave 100-400 cycles of issue.
    auto dma_piece = [&](int c, int buf, int k) {               // c: chunk index in the packed image (group * NCH + chunk)
        int piece = (wave_u & 3) * 10 + k + rot;                // k = 0 .. 9; wave-uniform: it travels in the scalar offset
        piece = piece >= 40 ? piece - 40 : piece;
        float *dst = Us + buf * G::UBUF + piece * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, (__attribute__((address_space(3))) void *)dst, 16,
                                                 lane * 16, c * (G::UBUF * 4) + piece * 1024, 0, 0);
    };
    // B^T d B of the two patches in `set` -> V[buf][pos][m][lane][k-step]
    auto xform_store = [&](int set, int buf) {
        float v[2][16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float row[4][4], t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float halo = d[set][s][r][2];
                const int xi = __builtin_bit_cast(int, d[set][s][r][0]), yi = __builtin_bit_cast(int, d[set][s][r][1]);
                const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, yi, 0x111, 0xf, 0xf, false));    // row_shr:1
                const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x101, 0xf, 0xf, false));   // row_shl:1
                row[r][0] = first_col ? halo : left;
                row[r][1] = d[set][s][r][0];
                row[r][2] = d[set][s][r][1];
                row[r][3] = last_col ? halo : right;
            }
            // one plain v_add / v_sub each: left to itself hipcc pairs these into v_pk_add_f32 plus the moves that line the
            // pairs up (63 + 45 instructions per chunk), and packed f32 arithmetic is slow beside the partner wave's MFMAs --
            // the transform took 2 100 cycles per chunk (profiles/r03e_stamps_40p.log)
            auto fadd = [](float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
            auto fsub = [](float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                t[0][x] = fsub(row[0][x], row[2][x]);
                t[1][x] = fadd(row[1][x], row[2][x]);
                t[2][x] = fsub(row[2][x], row[1][x]);
                t[3][x] = fsub(row[1][x], row[3][x]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[s][i * 4 + 0] = fsub(t[i][0], t[i][2]);
                v[s][i * 4 + 1] = fadd(t[i][1], t[i][2]);
                v[s][i * 4 + 2] = fsub(t[i][2], t[i][1]);
                v[s][i * 4 + 3] = fsub(t[i][1], t[i][3]);
            }
        }
        float *dstv = Vs + buf * G::VBUF + (m * 64 + lane) * 2;
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) *reinterpret_cast<f32x2 *>(dstv + pos * 512) = f32x2{v[0][pos], v[1][pos]};
    };

    f32x4 acc[16][NTW];
    // The MFMAs of the chunk in buffer `buf`. Behind the MFMAs of a position one slice of the loads is issued (in one
    // block at the top of the chunk they would hold the wave, and with it half of the SIMD's MFMA supply, for their whole
    // issue time): the DMA pieces of the next chunk's U into the other buffer, then (transform waves) the raw patches at
    // the load cursor into register set `set`.
    auto mfma_chunk = [&](auto first_tag, int buf, int set, int c_next) {
        constexpr bool FIRST = decltype(first_tag)::value;          // first chunk of a tile: the accumulators start at zero
        const float *va = Vs + buf * G::VBUF + (m * 64 + lane) * 2;
        const float *ub = Us + buf * G::UBUF + (N0 * 64 + lane) * 2;
        f32x2 a[2], bq[2][NTW];
        auto fetch = [&](int pos) {
            const int slot = pos & 1;
            a[slot] = *reinterpret_cast<const f32x2 *>(va + pos * 512);
#pragma unroll
            for (int n = 0; n < NTW; ++n) bq[slot][n] = *reinterpret_cast<const f32x2 *>(ub + (pos * 5 + n) * 128);
        };
        fetch(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + NTW, 0);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            if (pos + 1 < 16) fetch(pos + 1);
            const int slot = pos & 1;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    acc[pos][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(TILE_LANES ? bq[slot][n][s] : a[slot][s], TILE_LANES ? a[slot][s] : bq[slot][n][s],
                                                                       (FIRST && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pos][n], 0, 0, 0);
            int nvm = 0;
            if constexpr (XFORM) {
                // the patches of the chunk after next: they have until the transform at the start of the next chunk
                if (pos >= 2 && pos < 10) {
                    raw_load(set, pos - 2);
                    nvm = 2;
                }
            } else if (pos >= 2 && pos < 12) {          // U of the next chunk: has to land by the end of this one
                dma_piece(c_next, buf ^ 1, pos - 2);
                nvm = 1;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + NTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);
            if (nvm == 2) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
            else if (nvm == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
    };
    // A^T M A, bias, LeakyReLU(0.1) (+ 2x2 max) and the stores of tile w
    auto epilogue = [&](int w) {
        const int tile = w % ntile, b = w / (ntile * NG), grp = group_of(w);
        const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
        // TILE_LANES: this lane's tile (ty, tx), channels ch .. ch + 3 in the registers; else: its channel ch, tiles (ty, tx .. tx + 3)
        const int ty = TILE_LANES ? 2 * m + (p >> 3) : 2 * m + (q >> 1), tx = TILE_LANES ? (p & 7) : 4 * (q & 1);
        const long plane = (long)Hout * Hout;
#ifndef AXT_WINO_PK_EPILOGUE
        // one plain v_add / v_sub per element, as in the input transform: packed f32 arithmetic is slow beside the partner
        // wave's MFMAs (profiles/r04g_ab_epilogue.log)
        auto fadd = [](float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
        auto fsub = [](float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
#endif
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int ch = (N0 + n) * 16 + (TILE_LANES ? 4 * q : p);
            float *och = out + ((long)b * (80 * NG) + grp * 80 + ch) * plane;
            // the four elements of an accumulator register quadruple (four channels of a tile, or four tiles of a channel) go
            // through A^T M A, the bias and LeakyReLU(0.1) side by side
            f32x4 y[2][2];
#ifdef AXT_WINO_PK_EPILOGUE
            f32x4 t[2][4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                t[0][jj] = acc[jj][n] + acc[4 + jj][n] + acc[8 + jj][n];
                t[1][jj] = acc[4 + jj][n] - acc[8 + jj][n] - acc[12 + jj][n];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                y[a][0] = t[a][0] + t[a][1] + t[a][2];
                y[a][1] = t[a][1] - t[a][2] - t[a][3];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const f32x4 tb = y[a][bb] + bias_v[n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[a][bb][j] = fmaxf(tb[j], tb[j] * 0.1f);      // LeakyReLU(0.1) = max(t, 0.1 t)
                }
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t[2][4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    t[0][jj] = fadd(fadd(acc[jj][n][j], acc[4 + jj][n][j]), acc[8 + jj][n][j]);
                    t[1][jj] = fsub(fsub(acc[4 + jj][n][j], acc[8 + jj][n][j]), acc[12 + jj][n][j]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float y0v = fadd(fadd(t[a][0], t[a][1]), t[a][2]), y1v = fsub(fsub(t[a][1], t[a][2]), t[a][3]);
                    const float tb0 = fadd(y0v, bias_v[n][j]), tb1 = fadd(y1v, bias_v[n][j]);
                    y[a][0][j] = fmaxf(tb0, tb0 * 0.1f);                                        // LeakyReLU(0.1) = max(t, 0.1 t)
                    y[a][1][j] = fmaxf(tb1, tb1 * 0.1f);
                }
            }
#endif
            if constexpr (POOL) {
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = fmaxf(fmaxf(y[0][0][j], y[0][1][j]), fmaxf(y[1][0][j], y[1][1][j]));
#ifdef AXT_WINO_NO_STORE              // diagnostic build (timing only, wrong results): the epilogue without its stores
                if (r[0] == 12345.678f)
#endif
                *reinterpret_cast<f32x4 *>(och + (long)(y0 / 2 + ty) * Hout + x0 / 2 + tx) = r;
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float *orow = och + (long)(y0 + 2 * ty + a) * Hout + x0 + 2 * tx;
#ifdef AXT_WINO_NO_STORE
                    if (y[a][0][0] == 12345.678f)
#endif
                    {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x2 *>(orow + j * plane) = f32x2{y[a][0][j], y[a][1][j]};
                    }
                }
            }
        }
    };

    // ---- prologue: U and V of the first chunk; the patches of the second chunk on their way ----
    if constexpr (XFORM) {
        plan_tile(lw);
#pragma unroll
        for (int k = 0; k < 8; ++k) raw_load(0, k);
        advance_cursor();
        xform_store(0, 0);                                       // (waits for the patches)
#pragma unroll
        for (int k = 0; k < 8; ++k) raw_load(0, k);
        advance_cursor();
    } else {
#pragma unroll
        for (int k = 0; k < 10; ++k) dma_piece(group_of(wr.begin) * NCH, 0, k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- the chunk stream (LDS buffer = parity of the chunk count). During chunk g a transform wave first turns the
    // patches of chunk g + 1 (fetched behind the MFMAs of chunk g - 1) into V[other buffer] -- while its partner on the
    // SIMD, which has half again as many MFMAs per chunk, has the matrix pipe to itself -- and then joins the MFMAs of
    // chunk g, behind which the U of chunk g + 1 (LDS-DMA into the other U buffer) and the patches of chunk g + 2 (into the
    // registers the transform has just emptied) are issued. Nothing is left for the end of the chunk but the wait for
    // the DMA pieces. (Round 2 transformed at the END of the chunk, after the wave's MFMAs: the partner then ran its last
    // third of the chunk alone and every stall of its DMA issue idled the matrix pipe -- busy 0.65.)
    int w_cur = wr.begin, c = 0;
    bool done = false;
#ifdef AXT_WINO_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    // (Tried and dropped in round 3: the epilogue of a finished tile at the HEAD of the next chunk, so that the lighter wave's
    // next MFMAs run beside the heavier wave's output transform -- no difference within 0.5 %: vector instructions of one
    // wave make little headway beside the other wave's f32 MFMAs, which is also why the input transform is scalar code.)
    auto step = [&](int PAR) {
        const bool last = c + 1 == NCH;
        // U of the chunk after this one: the next chunk of this tile's group, or the first chunk of the next tile's group
        const int w_after = w_cur + wr.step < wr.end ? w_cur + wr.step : w_cur;
        const int c_next = last ? group_of(w_after) * NCH : group_of(w_cur) * NCH + c + 1;
        if constexpr (XFORM) {
#ifdef AXT_WINO_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WINO_STAMP(6);
#endif
            // the transform is the serial head of this wave's chunk: its VALU goes ahead of the partner's MFMA stream
            __builtin_amdgcn_s_setprio(AXT_WINO_XFORM_PRIO);
            xform_store(0, PAR ^ 1);                              // (after the WG's last chunk: of refetched patches, unused)
            __builtin_amdgcn_s_setprio(0);
            // the patch registers are free from here on: without this tie the compiler may issue the next loads into them
            // before the transform has read them all
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    asm volatile("" : "+v"(d[0][s][r][0]), "+v"(d[0][s][r][1]), "+v"(d[0][s][r][2]));
        }
        WINO_STAMP(0);
        if (c == 0) mfma_chunk(std::true_type{}, PAR, 0, c_next);
        else mfma_chunk(std::false_type{}, PAR, 0, c_next);
        WINO_STAMP(1);
        if constexpr (XFORM) advance_cursor();                    // (its patch loads stay in flight across the barrier)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces have landed
        WINO_STAMP(2);
        if (last) {
            epilogue(w_cur);
            c = 0;
            w_cur += wr.step;
            done = w_cur >= wr.end;
            if (NG > 1 && !done) load_bias(group_of(w_cur));
        } else ++c;
        WINO_STAMP(3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        WINO_STAMP(4);
    };
    int g = 0;
    do step(g++ & 1);
    while (!done);
#ifdef AXT_WINO_STAMPS
    if (CIN == AXT_WINO_STAMP_CIN && POOL == (bool)AXT_WINO_STAMP_POOL && NG == 1 && lane == 0 && blockIdx.x < 1024) {
        st_sum[5] = (unsigned long long)g;
        for (int i = 0; i < 8; ++i) g_wino_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_sum[i];
    }
#endif
}

template <int CIN, bool POOL, int NG>
__global__ __launch_bounds__(512, 1) void conv3x3_wino(const float *__restrict__ in,      // activations [B,CIN,H,H]
                                                       const float *__restrict__ upk,     // pack_wino image
                                                       const float *__restrict__ bias,    // folded bias [80*NG]
                                                       float *__restrict__ out,           // [B,80*NG,Hout,Hout]
                                                       int H, int B)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // INVARIANT: the two halves of the workgroup run different instantiations of wino_body and meet at the SAME hardware
    // barrier (s_barrier counts arriving waves, not program counters). That only works while both instantiations execute
    // exactly the same number of barriers: one per step() -- the step count depends on the tile range and CIN alone, never
    // on NB or XFORM -- plus the fixed ones of the prologue. Anything that adds a barrier to one half only (an extra
    // __syncthreads() under `if (XFORM)`, an early return) hangs the workgroup or races on LDS.
    if ((threadIdx.x >> 8) == 0) wino_body<CIN, POOL, 2, true, NG>(in, upk, bias, out, H, B, smem);
    else wino_body<CIN, POOL, 3, false, NG>(in, upk, bias, out, H, B, smem);
}

// ------------------------------------------------------------------------------------------------
// stride-2 conv3x3 on v_mfma_f32_4x4x1_16b_f32 (conv blocks 0 and 1).
//   These two layers have narrow N (20 / 40 channels) and, block 0, K = 45: on 16x16x4 tiles 20 pads to 32 and 45 to
//   48. The 4x4x1 shape (16 independent 4-pixel x 4-channel outer products per instruction, 8 cycles, the same
//   64 FLOP/clk/SIMD) fits exactly: M = 64 pixels per instruction, N in groups of 4 channels, K one at a time -- a
//   plain k-ordered FMA chain per output.
//     A (1 VGPR): lane l = the input value of pixel l for this k; B (1 VGPR): lane l = w[k][4g + l%4];
//     D (4 VGPRs): lane l, register i = pixel 4*(l/4) + i, channel 4g + l%4  -> 4 consecutive pixels, one float4 store.
//   * wave tile = 4 output rows x 32 columns (two instructions of 2 rows x 32 px), workgroup tile 16 x 32.
//   * every WAVE is its own pipeline: it stages the 9 input rows x 68 columns (from 2*x0 - 4, a multiple of 4: aligned
//     16-byte loads) of its tile into a wave-private LDS region, de-interleaved into an even-column and an odd-column
//     half per row, so that the stride-2 pixel reads become unit-stride (conflict-free ds_read_b32). No workgroup
//     barrier in the main loop; the waves of a CU drift apart and their loads, MFMAs and stores overlap.
//   * the next (tile, chunk)'s loads are issued into registers before the MFMAs of the current one and written to
//     LDS after them. The loads are buffer loads: segments outside the image get an out-of-range offset and the
//     range check returns the zero padding -- nothing touches the loaded registers before the LDS write, and inside
//     the image a tile costs no address arithmetic at all. Workgroups are persistent and walk an XCD-contiguous
//     tile list.
//   * ALL the layer's weights stay in LDS, [k][j][g] so that a lane fetches its NG weights of one k with b128 reads
//     (4 distinct addresses per instruction: broadcast, no conflicts).
//   conv block 0 (FIRST) gathers its 5 channels straight from frames t..t+4 of the timelapse at the tile origin
//   (fuses Timelapse.get_frametiles_stack, Timelapse.py:111-125,150-157), zero beyond the tile and the frame.
// ------------------------------------------------------------------------------------------------
template <int COUT>
struct GeoS2 {
    static constexpr int NG = COUT / 4, NGP = (NG + 3) / 4 * 4, NB4 = NGP / 4;
    static constexpr int PHW = 9, SEGS = 17, RW = 4 * SEGS, HALF = RW / 2, PLANE = PHW * RW;
    static constexpr int TH = 16, TW = 32;
};

template <int CIN, int COUT, int NPC, bool FIRST, int PF, int WGS>
__global__ __launch_bounds__(256, WGS) void conv3x3_s2_k1(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ bias,
    float *__restrict__ out, int Hin, int B,
    int Hf, int Wf, int t0, int tstep, int item0, int n_tiles, TileList tl)
{
    using G = GeoS2<COUT>;
    constexpr int NG = G::NG, NGP = G::NGP, NB4 = G::NB4, PHW = G::PHW, SEGS = G::SEGS, RW = G::RW, HALF = G::HALF,
                  PLANE = G::PLANE;
    constexpr int MT = 2;
    constexpr int WREGION = NPC * PLANE + 40;                         // + a spare row head (dummy store target)
    constexpr int NCHUNK = CIN / NPC, KPC = 9 * NPC;
    static_assert(CIN % NPC == 0 && COUT % 4 == 0, "CIN must be a multiple of the chunk, COUT of 4");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and known to be
    float *patch = smem + wave * WREGION;     // this wave's [NPC][9][even 34 | odd 34]
    float *wl = smem + 4 * WREGION;           // [CIN*9][4][NGP]  (whole layer, shared)
    float *bl = wl + CIN * 9 * 4 * NGP;       // [4][NG] folded bias, channel 4g + j at [j][g]

    const int Hout = Hin / 2;
    const int tiles_x = Hout / G::TW, ntile = tiles_x * (Hout / G::TH);
    // Work order. Block 1: the XCD-contiguous (item, tile) list of my_work. Block 0 reads frames t..t+4 for item t, so
    // consecutive items share four of their five frames -- but only while those are still in the XCD's 4 MB L2, and a
    // whole item streams 5 MB through it. Every XCD therefore takes a contiguous range of items and walks it one
    // row of tiles at a time: (row of tiles, item, tile in the row). The band of a frame that a row of tiles needs
    // (35 input rows, 70 KB) is then fetched once per XCD, for all the items that use it (20 frames x 70 KB in flight).
    int it_begin = 0, ni = B;
    WorkRange wr;
    if constexpr (FIRST) {
        const int xcd = blockIdx.x & 7;
        it_begin = (int)((long)xcd * B / 8);
        ni = (int)((long)(xcd + 1) * B / 8) - it_begin;
        wr.begin = blockIdx.x >> 3;
        wr.end = ni * ntile;
        wr.step = gridDim.x >> 3;
    } else {
        wr = my_work(ntile * B);
    }

    {
        constexpr int NW4 = CIN * 9 * 4 * NGP / 4;
        const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wpk);
        f32x4 *l4 = reinterpret_cast<f32x4 *>(wl);
        for (int e = tid; e < NW4; e += 256) l4[e] = w4[e];
        if (tid < COUT) bl[(tid & 3) * NG + (tid >> 2)] = bias[tid];
    }
    __syncthreads();
    if (wr.begin >= wr.end) return;

    const int a_base = (lane >> 5) * 2 * RW + (lane & 31);           // pixel (row lane/32, column lane%32) of an instruction
    const int b_base = (lane & 3) * NGP;
    const int blk = lane >> 2, jch = lane & 3;                       // result: pixels 4*blk..4*blk+3, channel 4g + jch

    // geometry of the source: channel / row strides in floats, constant for the whole launch
    const int cstride = FIRST ? Hf * Wf : Hin * Hin, rstride = FIRST ? Wf : Hin;

    // staging plan of the wave (float4 granularity): element e = lane + k*64 is (channel c, row r, 4-column segment).
    // The global side is a buffer load: descriptor base = the tile's first staged float (wave-uniform, rebuilt per
    // tile), per-lane byte offset voff[k] (constant for the launch), chunk offset in the scalar offset. Elements
    // outside the image get the offset kOobOffset instead: the range check returns zeros (the padding) for them.
    constexpr int NP4 = NPC * PHW * SEGS;
    constexpr int NPE = (NP4 + 63) / 64;
    constexpr int DUMMY = NPC * PLANE;
    unsigned lds_off[NPE];                   // LDS byte address of the element's even half
    unsigned ebits[NPE];                     // (1 << r) | (1 << (PHW + seg)): matched against the tile's valid rows / segments
    unsigned voff[NPE];
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
        const int e = lane + k * 64;
        const int c = e / (PHW * SEGS), rem = e - c * (PHW * SEGS);
        const int r = rem / SEGS, seg = rem - r * SEGS;
        lds_off[k] = (unsigned)(size_t)(patch + (e < NP4 ? (c * PLANE + r * RW + 2 * seg) : DUMMY));
        ebits[k] = e < NP4 ? (1u << r) | (1u << (PHW + seg)) : 0x80000000u;      // bit 31: never valid
        voff[k] = (unsigned)(c * cstride + r * rstride + 4 * seg) * 4u;
    }
    f32x4 pv[NPE];

    int cur_b = 0, cur_y0 = 0, cur_x0 = 0;
    int nxt_b = 0, nxt_y0 = 0, nxt_x0 = 0;
    unsigned nxt_valid = 0;
    __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 0, 0x00020000);
    auto decode_plan = [&](int w) {
        int tile;
        if constexpr (FIRST) {
            const int per_band = ni * tiles_x;
            const int band = w / per_band, r = w - band * per_band;
            const int ib = r / tiles_x;
            nxt_b = it_begin + ib;
            tile = band * tiles_x + (r - ib * tiles_x);
        } else {
            tile = w % ntile;
            nxt_b = w / ntile;
        }
        nxt_y0 = (tile / tiles_x) * G::TH;
        nxt_x0 = (tile % tiles_x) * G::TW;
        int lim_y, lim_x;
        long src;
        if constexpr (FIRST) {
            const int item = item0 + nxt_b;
            const int t = t0 + (item / n_tiles) * tstep, k = item % n_tiles;
            const int oy = tl.yx[2 * k] * AXT_TILE, ox = tl.yx[2 * k + 1] * AXT_TILE;
            src = ((long)t * Hf + oy) * Wf + ox;
            lim_y = min(AXT_TILE, Hf - oy);
            lim_x = min(AXT_TILE, Wf - ox);
        } else {
            src = (long)nxt_b * CIN * Hin * Hin;
            lim_y = Hin;
            lim_x = Hin;
        }
        // first staged row of this wave / first staged column (a multiple of 4)
        const int iy0 = (nxt_y0 + wave * 4) * 2 - 1, jx0 = nxt_x0 * 2 - 4;
        src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in) + (src + (long)iy0 * rstride + jx0), 0,
                                                     kBufRecords, 0x00020000);
        // rows r with 0 <= iy0 + r < lim_y and segments with 0 <= jx0 + 4 seg, jx0 + 4 seg + 3 < lim_x (widths are
        // multiples of 4: a segment is entirely inside or entirely outside), as one scalar bit set
        const int r_lo = max(0, -iy0), r_hi = min(PHW, lim_y - iy0);
        const int s_lo = max(0, -jx0 / 4), s_hi = min(SEGS, (lim_x - jx0) / 4);
        const unsigned rows = r_hi > r_lo ? ((1u << r_hi) - 1u) & ~((1u << r_lo) - 1u) : 0u;
        const unsigned segs = s_hi > s_lo ? ((1u << s_hi) - 1u) & ~((1u << s_lo) - 1u) : 0u;
        nxt_valid = rows | (segs << PHW);
    };
    auto load_chunk = [&](int chunk) {
        const int soff = chunk * NPC * cstride * 4;
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
            const unsigned off = (ebits[k] & nxt_valid) == ebits[k] ? voff[k] : kOobOffset;
            pv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, (int)off, soff, 0));
        }
    };
    auto store_chunk = [&]() {
        // The compiler does not count the LDS operations issued from asm; lgkmcnt has 4 bits, so they are drained
        // before more than 12 are in flight and before compiler-generated LDS reads follow.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
            // two ds_write2_b32 straight from the loaded registers (the compiler would first shuffle them into
            // pairs for ds_write2_b64: 4 moves per element). Even columns, then odd columns.
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds_off[k]), "v"(pv[k][0]), "v"(pv[k][2]) : "memory");
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(lds_off[k]), "v"(pv[k][1]), "v"(pv[k][3]),
                         "n"(HALF), "n"(HALF + 1) : "memory");
            if (k % 6 == 5 || k == NPE - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    // results leave through a buffer store: per-lane offset constant, everything tile-dependent in the scalar offset
    const __amdgpu_buffer_rsrc_t dst_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, B * COUT * Hout * Hout * 4, 0x00020000);
    const int st_voff = (jch * Hout * Hout + (blk >> 3) * Hout + (blk & 7) * 4) * 4;
    // the accumulators start at the folded bias (kept in LDS, [jch][g]): out = lrelu(bias + sum_k w_k x_k), k ascending
    f32x4 acc[MT][NG];
    auto reset_acc = [&](int g) {
        const float bv = bl[jch * NG + g];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][g] = f32x4{bv, bv, bv, bv};
    };
#pragma unroll
    for (int g = 0; g < NG; ++g) reset_acc(g);

    // epilogue of a finished tile: LeakyReLU on the accumulators, one 16-byte store per (row pair, channel group),
    // accumulators back to the bias
    auto write_tile = [&](int b, int y0, int x0) {
        const int tile_soff = (((b * COUT) * Hout + y0 + wave * 4) * Hout + x0) * 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v = acc[m][g];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * 0.1f);          // LeakyReLU(0.1)
                // The tile-dependent part of the address goes into the VECTOR offset on purpose. With it in the scalar
                // offset hipcc (ROCm 7.2) inserts no wait state between the store and the next write of its data
                // registers, and on gfx950 a 16-byte buffer store then picks up part of the NEXT group's values
                // whenever another workgroup shares the CU (measured: ~0.05 % of the outputs wrong).
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst_rsrc,
                                                       st_voff + tile_soff + (4 * g * Hout + 2 * m) * Hout * 4, 0, 0);
            }
            reset_acc(g);
        }
    };

    // Order of one step: [wait for this step's loads] LDS writes -> results of the tile finished in the previous
    // step out -> next step's loads issued -> MFMAs. The stores are thus always OLDER than the loads a step waits
    // for (vmcnt counts both): no step waits for a store to be acknowledged, and the stores drain behind the MFMAs.
    int w = wr.begin, chunk = 0;
    decode_plan(w);
    cur_b = nxt_b; cur_y0 = nxt_y0; cur_x0 = nxt_x0;
    load_chunk(0);
    bool out_pending = false;
    int out_b = 0, out_y0 = 0, out_x0 = 0;
#pragma unroll 1
    for (;;) {
        // LDS accesses of one wave execute in order: the reads of the previous step are done before these writes
        store_chunk();
        if (out_pending) write_tile(out_b, out_y0, out_x0);
        const bool last_chunk = (chunk == NCHUNK - 1);
        const bool has_next = w + wr.step < wr.end;
        if (last_chunk && has_next) decode_plan(w + wr.step);
        if (!last_chunk || has_next) load_chunk(last_chunk ? 0 : chunk + 1);     // ONE load site: one register set
        const float *wc = wl + chunk * KPC * 4 * NGP + b_base;
        {
            // operands of k-step k + PF are fetched from LDS while the MFMAs of k-step k issue (register ring of PF + 1
            // sets); the group barriers pin that interleave -- left alone the scheduler sinks every read to just
            // before its first use and each k-step then waits out the LDS latency.
            float a[PF + 1][MT];
            f32x4 bq[PF + 1][NB4];
            auto fetch = [&](int k) {
                const int c = k / 9, ky = (k % 9) / 3, kx = k % 3, slot = k % (PF + 1);
                // input column 2x + kx - 1: kx = 0 -> odd half at x - 1, kx = 1 -> even half at x, kx = 2 -> odd half
                // at x (both halves start two columns left of the tile)
                const int kxoff = kx == 0 ? HALF + 1 : kx == 1 ? 2 : HALF + 2;
#pragma unroll
                for (int m = 0; m < MT; ++m) a[slot][m] = patch[a_base + c * PLANE + (4 * m + ky) * RW + kxoff];
#pragma unroll
                for (int i = 0; i < NB4; ++i) bq[slot][i] = *reinterpret_cast<const f32x4 *>(wc + k * 4 * NGP + 4 * i);
            };
#pragma unroll
            for (int k = 0; k < PF; ++k) fetch(k);
            __builtin_amdgcn_sched_group_barrier(0x100, PF * (MT + NB4), 0);   // the ring's head start
#pragma unroll
            for (int k = 0; k < KPC; ++k) {
                if (k + PF < KPC) fetch(k + PF);
                const int slot = k % (PF + 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[slot][m], bq[slot][g / 4][g % 4], acc[m][g], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MT + NB4, 0);      // the DS reads of k-step k + PF
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NG, 0);       // the MFMAs of k-step k
            }
        }
        out_pending = last_chunk;
        if (last_chunk) {
            out_b = cur_b; out_y0 = cur_y0; out_x0 = cur_x0;
            if (!has_next) break;
            w += wr.step;
            cur_b = nxt_b; cur_y0 = nxt_y0; cur_x0 = nxt_x0;
            chunk = 0;
        } else {
            ++chunk;
        }
    }
    write_tile(out_b, out_y0, out_x0);
}

// rows [n rows of w floats] -> rows of `pitch` floats (a multiple of 4), zero-filled beyond w
__global__ void repitch_kernel(const float *__restrict__ in, int w, int pitch, long n4, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int p4 = pitch / 4;
    const long row = i / p4;
    const int x = (int)(i - row * p4) * 4;
    const float *src = in + row * w + x;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = x + j < w ? src[j] : 0.f;
    *reinterpret_cast<f32x4 *>(out + row * pitch + x) = v;
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
    // The next tile's global loads are issued into registers before the MFMAs of the current one and written to LDS
    // after them (the weight stream of the first linear layer is 168 MB: its latency must hide behind the MFMAs).
    float2 av[4];
    f32x4 bv[2];
    const float *ap[4];
    const float *bp[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256;
        const int r = e >> 4, c2 = e & 15;
        int gm = m0 + r;
        gm = gm < M ? gm : M - 1;
        ap[i] = A + (long)gm * lda + c2 * 2;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256;
        bp[i] = Bm + (long)(e >> 4) * N + n0 + (e & 15) * 4;
    }
    auto load_tile = [&](int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const float2 *>(ap[i] + k);
#pragma unroll
        for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(bp[i] + (long)k * N);
    };
    load_tile(k0);
    for (int k = k0; k < k0 + kper; k += GBK) {
        if (k != k0) __syncthreads();            // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float2 *>(&As[(e >> 4) * GLDA + (e & 15) * 2]) = av[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<f32x4 *>(&Bs[(e >> 4) * GLDB + (e & 15) * 4]) = bv[i];
        }
        __syncthreads();
        if (k + GBK < k0 + kper) load_tile(k + GBK);
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

// The same GEMM with a 128 x 128 block tile (wave tile 64 x 64 = 4 x 4 MFMA tiles) for the first linear layer, whose
// 168 MB of weights and 41 MB of activations are each streamed once per block row / column: a 64 x 64 tile reads the
// weights 4 times and the activations 16 times (1.3 GB), this one 2 and 8 times.
constexpr int HBM_ = 128, HBN = 128, HLDA = 34, HLDB = 144;

__global__ __launch_bounds__(256) void gemm_mfma_128(const float *__restrict__ A, int lda, const float *__restrict__ Bm, int N,
                                                     float *__restrict__ slab, int M, int kper)
{
    __shared__ __attribute__((aligned(16))) float As[HBM_ * HLDA];
    __shared__ __attribute__((aligned(16))) float Bs[GBK * HLDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * HBN, m0 = blockIdx.y * HBM_;
    const int k0 = blockIdx.z * kper;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: A tile 128 rows x 32 k as float2 (2048 -> 8 per thread), B tile 32 k x 128 n as float4 (1024 -> 4)
    float2 av[8];
    f32x4 bv[4];
    const float *ap[8];
    const float *bp[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256;
        int gm = m0 + (e >> 4);
        gm = gm < M ? gm : M - 1;
        ap[i] = A + (long)gm * lda + (e & 15) * 2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256;
        bp[i] = Bm + (long)(e >> 5) * N + n0 + (e & 31) * 4;
    }
    auto load_tile = [&](int k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const float2 *>(ap[i] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(bp[i] + (long)k * N);
    };
    load_tile(k0);
    for (int k = k0; k < k0 + kper; k += GBK) {
        if (k != k0) __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float2 *>(&As[(e >> 4) * HLDA + (e & 15) * 2]) = av[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<f32x4 *>(&Bs[(e >> 5) * HLDB + (e & 31) * 4]) = bv[i];
        }
        __syncthreads();
        if (k + GBK < k0 + kper) load_tile(k + GBK);
#pragma unroll
        for (int s = 0; s < GBK / 4; ++s) {
            float a[4], bw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wm * 64 + i * 16 + p) * HLDA + s * 4 + q];
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[j] = Bs[(s * 4 + q) * HLDB + wn * 64 + j * 16 + p];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bw[j], acc[i][j], 0, 0, 0);
        }
    }
    float *dst = slab + (long)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + q * 4 + r, n = n0 + wn * 64 + j * 16 + p;
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
static const ConvPlan kPlan[8] = {{5, 2, 1}, {4, 3, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 1}, {8, 5, 2}};

// Tile-forwards per launch of the front layers. Measured on MI355X (252 tile-forwards): 16/32 -> 6.16 ms, 64/64 ->
// 5.9 ms, 128/128 -> 5.8 ms, 256/256 -> 6.1 ms for the whole CNN: what matters is that every persistent workgroup of
// the stride-2 kernels gets several tiles and that launch gaps amortise, not that the activations of one chunk fit
// the Infinity Cache. 128 items of block-0 output are 0.67 GB. Measured again with the fused front kernel (round 3, -DAXT_CHUNK_A/B):
// 64/128 -> 3.69 ms, 128/128 -> 3.65 ms, 256/256 -> 3.61-3.68 ms: no difference beyond the run-to-run spread.
#ifndef AXT_CHUNK_A
#define AXT_CHUNK_A 128
#endif
#ifndef AXT_CHUNK_B
#define AXT_CHUNK_B 128
#endif
constexpr int kChunkA = AXT_CHUNK_A;      // conv blocks 0-2
constexpr int kChunkB = AXT_CHUNK_B;      // conv blocks 3-4
constexpr int kFc1Split = 32, kFc2Split = 4, kFc3Split = 4;

}  // namespace

struct axt_detector {
    int max_batch = 0;
    float *d_wconv[8] = {};     // packed conv weights
    unsigned *d_wb3[8] = {};    // conv blocks 2..6 packed for conv3x3_bf16x3 (allocated on the first switch to that arithmetic)
    std::vector<float> h_wfold[8];   // their BN-folded f32 weights [cout][cin][3][3], kept on the host for that packing
    float *d_wwino[8] = {};     // conv blocks 2..6 packed for conv3x3_wino (allocated on the first switch to that arithmetic)
    int fuse01 = 1;             // conv blocks 0 and 1 in one kernel (conv_s2_fused); axt_detector_set_fused_front, AXT_FUSE_S2=0 at create
    int arith = 0;              // stride-1 blocks: 0 direct f32 MFMA | 1 bf16x3 (blocks 2..8) | 2 f32 Winograd F(2x2,3x3) (set by create)
    float *d_bconv[8] = {};     // folded bias
    float *d_wfc[3] = {};       // [K][Npad]
    float *d_bfc[3] = {};
    float *d_act[8] = {};       // activations after conv block i (chunk-sized for i < 4)
    float *d_slab = nullptr, *d_fc1 = nullptr, *d_fc2 = nullptr;
    float *d_pad = nullptr;     // frames re-pitched to a multiple of 4 floats per row (only for such timelapses)
    size_t pad_cap = 0;
    size_t bytes = 0;
    // optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
    bool profiling = false;
    int profile_only = -1;      // >= 0: only this kernel index is bracketed with events
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
        if (!d->profiling || (d->profile_only >= 0 && d->profile_only != kernel_)) return;
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

// Packs one conv block: folds BN (f64).
// stride-1 layers (conv3x3_mfma): [group][chunk][krow][NPADW], krow inside a chunk =
//   ((ky*3+kx) * CCH/4 + cg) * 4 + kk  <->  channel chunk*CCH + cg*4 + kk
// stride-2 layers (conv3x3_s2_k1): [k = ci*9 + ky*3 + kx][j][NGP], channel 4g + j at position g.
void pack_conv(int li, const float *w, const float *b, const float *gamma, const float *beta,
               const float *mean, const float *var, std::vector<float> &wp, std::vector<float> &bp,
               std::vector<float> *wfold = nullptr)
{
    const ConvSpec &cs = kConv[li];
    const ConvPlan &pl = kPlan[li];
    const int NPADW = npadw(pl.nt);
    const bool s2 = cs.stride == 2;
    const int nchunk = cs.cin / pl.cch;
    const int krows = 9 * pl.cch;
    const int ngp = (cs.cout / 4 + 3) / 4 * 4;
    wp.assign(s2 ? (size_t)cs.cin * 9 * 4 * ngp : (size_t)pl.ngroups * nchunk * krows * NPADW, 0.f);
    bp.assign(cs.cout, 0.f);
    if (wfold) wfold->assign((size_t)cs.cout * cs.cin * 9, 0.f);
    for (int co = 0; co < cs.cout; ++co) {
        const double sc = (double)gamma[co] / sqrt((double)var[co] + 1e-5);
        bp[co] = (float)(((double)b[co] - (double)mean[co]) * sc + (double)beta[co]);
        const int grp = co / (pl.nt * 16), col = co % (pl.nt * 16);
        for (int ci = 0; ci < cs.cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = (float)((double)w[(((size_t)co * cs.cin + ci) * 3 + ky) * 3 + kx] * sc);
                    if (wfold) (*wfold)[(((size_t)co * cs.cin + ci) * 3 + ky) * 3 + kx] = v;
                    if (s2) {
                        wp[(((size_t)ci * 9 + ky * 3 + kx) * 4 + co % 4) * ngp + co / 4] = v;
                    } else {
                        const int chunk = ci / pl.cch, c = ci % pl.cch;
                        const int krow = ((ky * 3 + kx) * (pl.cch / 4) + c / 4) * 4 + c % 4;
                        wp[(((size_t)grp * nchunk + chunk) * krows + krow) * NPADW + col] = v;
                    }
                }
    }
}

// f32 -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)
inline uint16_t bf16_rne(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// Packs the BN-folded weights [80][cin][3][3] of a stride-1 block for conv3x3_bf16x3: per chunk of 16 input channels the
// LDS image [k-step 5][plane hi|mid|lo][q 4][n 80][8] bf16; pair 4*step + q = (tap, 8-channel half), pairs >= 18 are zero.
void pack_bf16x3(int cin, const std::vector<float> &wfold, std::vector<uint16_t> &out)
{
    using G = GeoB3;
    const int nchunk = (cin + G::CCH - 1) / G::CCH;
    out.assign((size_t)nchunk * G::WCHUNK_B / 2, 0);
    for (int chunk = 0; chunk < nchunk; ++chunk)
        for (int s = 0; s < G::KS; ++s)
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * s + q;
                if (j >= 18) continue;
                const int tap = j >> 1, half = j & 1, ky = tap / 3, kx = tap % 3;
                for (int n = 0; n < 80; ++n)
                    for (int i = 0; i < 8; ++i) {
                        const int c = chunk * G::CCH + 8 * half + i;
                        if (c >= cin) continue;
                        const float v = wfold[(((size_t)n * cin + c) * 3 + ky) * 3 + kx];
                        const uint16_t hi = bf16_rne(v);
                        const float r1 = v - bf16_to_f32(hi);
                        const uint16_t mid = bf16_rne(r1);
                        const uint16_t lo = bf16_rne(r1 - bf16_to_f32(mid));
                        const uint16_t pl[3] = {hi, mid, lo};
                        for (int k = 0; k < 3; ++k)
                            out[((((size_t)(chunk * G::KS + s) * 3 + k) * 4 + q) * 80 + n) * 8 + i] = pl[k];
                    }
            }
}

template <int CIN, bool POOL>
int launch_conv_b3(const float *in, const unsigned *w, const float *bias, float *out, int Hin, int B, hipStream_t st)
{
    auto kern = conv3x3_bf16x3<CIN, POOL>;
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(kern, (int)(GeoB3::LDS_B), once)) return rc;
    AXT_REQUIRE(Hin % GeoB3::TH == 0 && w != nullptr, "conv (bf16x3): map size %d not a multiple of the 32x16 tile, or weights not packed", Hin);
    AXT_REQUIRE((double)CIN * Hin * Hin * 4 < 2.0e9, "conv (bf16x3): map too large");
    const int nwork = (Hin / GeoB3::TH) * (Hin / GeoB3::TW) * B;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(64 * GeoB3::WAVES), GeoB3::LDS_B, st, in, w, bias, out, Hin, B);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

// persistent grids: a multiple of 8 (one slice per XCD), at most `per_cu` workgroups per CU
int persistent_grid(int nwork, int per_cu)
{
    int g = 256 * per_cu;
    if (g > nwork) g = nwork;
    g = ((g + 7) / 8) * 8;
    return g;
}

// Packs the BN-folded weights [80][cin][3][3] of a stride-1 block for conv3x3_wino: U = G g G^T (f64, rounded once) in
// the LDS image order [group of 80 output channels][chunk of 8 channels][pos 16][channel block 5][lane = q*16 + p][k-step 2]:
// input channel chunk*8 + s*4 + q, output channel group*80 + block*16 + p.
void pack_wino(int cin, int cout, const std::vector<float> &wfold, std::vector<float> &out)
{
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nchunk = cin / GeoW::CCH;
    out.assign((size_t)(cout / 80) * nchunk * GeoW::UBUF, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = &wfold[((size_t)co * cin + ci) * 9];
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = Gm[i][0] * g[0 * 3 + b] + Gm[i][1] * g[1 * 3 + b] + Gm[i][2] * g[2 * 3 + b];
            const int chunk = (co / 80) * nchunk + ci / 8, s = (ci % 8) / 4, q = ci % 4, n = (co % 80) / 16, p = co % 16;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = tmp[i][0] * Gm[j][0] + tmp[i][1] * Gm[j][1] + tmp[i][2] * Gm[j][2];
                    out[((((size_t)chunk * 16 + i * 4 + j) * 5 + n) * 64 + q * 16 + p) * 2 + s] = (float)u;
                }
        }
}

template <int CIN, bool POOL, int NG = 1>
int launch_conv_wino(const float *in, const float *u, const float *bias, float *out, int Hin, int B, hipStream_t st)
{
    auto kern = conv3x3_wino<CIN, POOL, NG>;
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(kern, (int)(GeoW::LDS_B), once)) return rc;
    AXT_REQUIRE(Hin % 16 == 0 && u != nullptr, "conv (winograd): map size %d not a multiple of the 16x16 tile, or weights not packed", Hin);
    AXT_REQUIRE((double)CIN * Hin * Hin * 4 < 2.0e9, "conv (winograd): map too large");
    const int nwork = (Hin / 16) * (Hin / 16) * B * NG;
    hipLaunchKernelGGL(kern, dim3(persistent_grid(nwork, 1)), dim3(512), GeoW::LDS_B, st, in, u, bias, out, Hin, B);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

template <int CIN, int COUT, bool POOL, int CCH, int NT>
int launch_conv(const float *in, const float *w, const float *bias, float *out, int Hin, int ngroups, int B,
                hipStream_t st)
{
    auto kern = conv3x3_mfma<CIN, COUT, POOL, CCH, NT>;
    using G = GeoS1<CCH>;
    constexpr size_t lds = (size_t)(((CCH * G::PLANE + 3) & ~3) + G::KROWS * npadw(NT)) * sizeof(float);
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(kern, (int)(lds), once)) return rc;
    AXT_REQUIRE(Hin % 16 == 0, "conv: map size %d not a multiple of the 16x16 tile", Hin);
    const int nwork = (Hin / 16) * (Hin / 16) * ngroups * B;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), lds, st, in, w, bias, out, Hin, ngroups, B);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

template <int CIN, int COUT, int NPC, bool FIRST, int PF, int WGS = 2>
int launch_conv_s2(const float *in, const float *w, const float *bias, float *out, int Hin, int B,
                   hipStream_t st, int Hf = 0, int Wf = 0, int t0 = 0, int tstep = 1, int item0 = 0, int n_tiles = 1,
                   const TileList *tl = nullptr)
{
    auto kern = conv3x3_s2_k1<CIN, COUT, NPC, FIRST, PF, WGS>;
    using G = GeoS2<COUT>;
    constexpr size_t lds = (size_t)(4 * (NPC * G::PLANE + 40) + CIN * 9 * 4 * G::NGP + COUT) * sizeof(float);
    static AxtOncePerDevice once;                 // (per device: see axt_common.h)
    if (int rc = axt_max_dynamic_lds(kern, (int)(lds), once)) return rc;
    AXT_REQUIRE(Hin % 64 == 0, "conv: map size %d not a multiple of the tile", Hin);
    AXT_REQUIRE(!FIRST || Wf % 4 == 0, "conv: frame pitch %d not a multiple of 4", Wf);
    // buffer addressing: offsets inside one tile's source window and inside the output of one launch are 32-bit
    AXT_REQUIRE(!FIRST || (double)Hf * Wf * 5 * 4 < 2.0e9, "conv: frames of %d x %d are too large", Hf, Wf);
    AXT_REQUIRE((double)B * COUT * (Hin / 2) * (Hin / 2) * 4 < 2.0e9, "conv: batch of %d is too large for one launch", B);
    TileList dummy;
    dummy.n = 0;
    const int nwork = (Hin / 2 / G::TH) * (Hin / 2 / G::TW) * B;
    // resident workgroups per CU for this kernel (registers + LDS), asked of the device the launch goes to
    static std::atomic<int> per_cu_of[64];
    int dev = 0;
    AXT_CHECK_HIP(hipGetDevice(&dev));
    int per_cu = dev >= 0 && dev < 64 ? per_cu_of[dev].load(std::memory_order_relaxed) : 0;
    if (per_cu == 0) {
        int n = 0;
        AXT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kern, 256, lds));
        per_cu = n < 1 ? 1 : (n > 8 ? 8 : n);
        if (dev >= 0 && dev < 64) per_cu_of[dev].store(per_cu, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(persistent_grid(nwork, per_cu)), dim3(256), lds, st, in, w, bias, out, Hin, B, Hf, Wf,
                       t0, tstep, item0, n_tiles, tl ? *tl : dummy);
    AXT_LAUNCH_CHECK();
    return AXT_OK;
}

int launch_gemm(const float *A, int lda, const float *Bm, int N, float *slab, int M, int K, int split,
                hipStream_t st)
{
    AXT_REQUIRE(N % GBN == 0 && K % (split * GBK) == 0, "gemm: bad shape N=%d K=%d split=%d", N, K, split);
    if (N % HBN == 0 && M > GBM && K >= 8192) {          // the first linear layer at batch sizes beyond one 64-row tile
        dim3 grid(N / HBN, axt_cdiv(M, HBM_), split);
        hipLaunchKernelGGL(gemm_mfma_128, grid, dim3(256), 0, st, A, lda, Bm, N, slab, M, K / split);
        AXT_LAUNCH_CHECK();
        return AXT_OK;
    }
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
    if (d->fuse01 && Wf % 4 == 0) {
        // blocks 0 and 1 in one kernel (its 16-byte input pieces need aligned rows: other widths take the separate kernels)
        ProfSpan ps(d, st, 0, nb);
        if ((rc = axt_launch_conv_fused01(frames, d->d_wconv[0], d->d_bconv[0], d->d_wconv[1], d->d_bconv[1], d->d_act[1], nb, st,
                                      Hf, Wf, t0, tstep, item0, n_tiles, tl))) return rc;
    } else {
    {
        ProfSpan ps(d, st, 0, nb);
        const float *src = frames;
        int pitch = Wf, t_first = 0;
        if (Wf % 4 != 0) {
            // rows that are not 16-byte aligned: copy the frames this launch reads into a buffer whose pitch is
            // (the zero columns it adds are what the tile padding would read anyway)
            t_first = t0 + (item0 / n_tiles) * tstep;
            const int t_last = t0 + ((item0 + nb - 1) / n_tiles) * tstep + AXT_IN_CH - 1;
            const int nf = t_last - t_first + 1;
            pitch = (Wf + 3) / 4 * 4;
            const size_t need = (size_t)nf * Hf * pitch;
            if (need > d->pad_cap) {
                AXT_CHECK_HIP(hipStreamSynchronize(st));
                (void)hipFree(d->d_pad);
                d->d_pad = nullptr;
                d->pad_cap = 0;
                AXT_CHECK_HIP(hipMalloc((void **)&d->d_pad, need * sizeof(float)));
                d->pad_cap = need;
            }
            const long n4 = (long)nf * Hf * (pitch / 4);
            hipLaunchKernelGGL(repitch_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                               frames + (size_t)t_first * Hf * Wf, Wf, pitch, n4, d->d_pad);
            AXT_LAUNCH_CHECK();
            src = d->d_pad;
        }
        rc = launch_conv_s2<5, 20, 5, true, 3>(src, d->d_wconv[0], d->d_bconv[0], d->d_act[0], 512, nb, st, Hf, pitch,
                                            t0 - t_first, tstep, item0, n_tiles, &tl);
        if (rc) return rc;
    }
    {
        ProfSpan ps(d, st, 1, nb);
        if ((rc = launch_conv_s2<20, 40, 4, false, 2>(d->d_act[0], d->d_wconv[1], d->d_bconv[1], d->d_act[1], 256, nb, st))) return rc;
    }
    }
    {
        ProfSpan ps(d, st, 2, nb);
        float *dst = d->d_act[2] + (size_t)slot0 * 80 * 64 * 64;
        rc = d->arith == 2 ? launch_conv_wino<40, true>(d->d_act[1], d->d_wwino[2], d->d_bconv[2], dst, 128, nb, st)
             : d->arith ? launch_conv_b3<40, true>(d->d_act[1], d->d_wb3[2], d->d_bconv[2], dst, 128, nb, st)
                      : launch_conv<40, 80, true, 8, 5>(d->d_act[1], d->d_wconv[2], d->d_bconv[2], dst, 128, 1, nb, st);
        if (rc) return rc;
    }
    return AXT_OK;
}

// conv blocks 3..4 for up to kChunkB items: d_act[2] -> act4_out
int run_front_b(axt_detector *d, int nb, float *act4_out, hipStream_t st)
{
    int rc;
    {
        ProfSpan ps(d, st, 3, nb);
        rc = d->arith == 2 ? launch_conv_wino<80, false>(d->d_act[2], d->d_wwino[3], d->d_bconv[3], d->d_act[3], 64, nb, st)
             : d->arith ? launch_conv_b3<80, false>(d->d_act[2], d->d_wb3[3], d->d_bconv[3], d->d_act[3], 64, nb, st)
                      : launch_conv<80, 80, false, 8, 5>(d->d_act[2], d->d_wconv[3], d->d_bconv[3], d->d_act[3], 64, 1, nb, st);
        if (rc) return rc;
    }
    {
        ProfSpan ps(d, st, 4, nb);
        rc = d->arith == 2 ? launch_conv_wino<80, true>(d->d_act[3], d->d_wwino[4], d->d_bconv[4], act4_out, 64, nb, st)
             : d->arith ? launch_conv_b3<80, true>(d->d_act[3], d->d_wb3[4], d->d_bconv[4], act4_out, 64, nb, st)
                      : launch_conv<80, 80, true, 8, 5>(d->d_act[3], d->d_wconv[4], d->d_bconv[4], act4_out, 64, 1, nb, st);
        if (rc) return rc;
    }
    return AXT_OK;
}

// layers 5..7 + the three linear layers for nb items whose block-4 output is in d_act[4]
int run_back(axt_detector *d, int nb, float *d_yolo, hipStream_t st)
{
    int rc;
    {
        ProfSpan ps(d, st, 5, nb);
        rc = d->arith == 2 ? launch_conv_wino<80, false>(d->d_act[4], d->d_wwino[5], d->d_bconv[5], d->d_act[5], 32, nb, st)
             : d->arith ? launch_conv_b3<80, false>(d->d_act[4], d->d_wb3[5], d->d_bconv[5], d->d_act[5], 32, nb, st)
                      : launch_conv<80, 80, false, 8, 5>(d->d_act[4], d->d_wconv[5], d->d_bconv[5], d->d_act[5], 32, 1, nb, st);
        if (rc) return rc;
    }
    {
        ProfSpan ps(d, st, 6, nb);
        rc = d->arith == 2 ? launch_conv_wino<80, true>(d->d_act[5], d->d_wwino[6], d->d_bconv[6], d->d_act[6], 32, nb, st)
             : d->arith ? launch_conv_b3<80, true>(d->d_act[5], d->d_wb3[6], d->d_bconv[6], d->d_act[6], 32, nb, st)
                      : launch_conv<80, 80, true, 8, 5>(d->d_act[5], d->d_wconv[6], d->d_bconv[6], d->d_act[6], 32, 1, nb, st);
        if (rc) return rc;
    }
    {
        ProfSpan ps(d, st, 7, nb);
        rc = d->arith == 2 ? launch_conv_wino<80, false, 2>(d->d_act[6], d->d_wwino[7], d->d_bconv[7], d->d_act[7], 16, nb, st)
                           : launch_conv<80, 160, false, 8, 5>(d->d_act[6], d->d_wconv[7], d->d_bconv[7], d->d_act[7], 16, 2, nb, st);
        if (rc) return rc;
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
        pack_conv(li, t[0], t[1], t[2], t[3], t[4], t[5], wp, bp, (li >= 2 && li <= 7) ? &d->h_wfold[li] : nullptr);
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
        rc = dev_alloc(d, &d->d_act[i], per_item[i] * (size_t)std::min(max_batch, i < 2 ? kChunkA : i < 4 ? kChunkB : max_batch));
    if (!rc) rc = dev_alloc(d, &d->d_slab, (size_t)kFc1Split * max_batch * kFc);
    if (!rc) rc = dev_alloc(d, &d->d_fc1, (size_t)max_batch * kFc);
    if (!rc) rc = dev_alloc(d, &d->d_fc2, (size_t)max_batch * kFc);
    if (!rc && hipDeviceSynchronize() != hipSuccess) {
        axt_set_error("device synchronize failed after upload");
        rc = AXT_EHIP;
    }
    if (const char *e = getenv("AXT_FUSE_S2")) d->fuse01 = atoi(e) != 0;
    if (!rc) rc = axt_detector_set_arith(d, 2);        // default arithmetic: f32 Winograd for the stride-1 blocks
    if (rc) {
        axt_detector_destroy(d);
        return rc;
    }
    *out = d;
    return AXT_OK;
}

int axt_detector_set_arith(axt_detector *d, int mode)
{
    AXT_REQUIRE(d != nullptr && mode >= 0 && mode <= 2, "axt_detector_set_arith: mode must be 0 (direct f32), 1 (bf16x3) or 2 (f32 Winograd)");
    if (mode == 1 && !d->d_wb3[2]) {
        std::vector<uint16_t> pk;
        for (int li = 2; li <= 6; ++li) {
            pack_bf16x3(kConv[li].cin, d->h_wfold[li], pk);
            const int rc = dev_alloc(d, &d->d_wb3[li], pk.size() / 2);
            if (rc) return rc;
            AXT_CHECK_HIP(hipMemcpy(d->d_wb3[li], pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        }
    }
    if (mode == 2 && !d->d_wwino[2]) {
        std::vector<float> pk;
        for (int li = 2; li <= 7; ++li) {
            pack_wino(kConv[li].cin, kConv[li].cout, d->h_wfold[li], pk);
            const int rc = dev_alloc(d, &d->d_wwino[li], pk.size());
            if (rc) return rc;
            AXT_CHECK_HIP(hipMemcpy(d->d_wwino[li], pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        }
    }
    d->arith = mode;
    return AXT_OK;
}

int axt_detector_set_fused_front(axt_detector *d, int fused)
{
    AXT_REQUIRE(d != nullptr && (fused == 0 || fused == 1), "axt_detector_set_fused_front: fused must be 0 or 1");
    d->fuse01 = fused;
    return AXT_OK;
}

void axt_detector_destroy(axt_detector *d)
{
    if (!d) return;
    for (int i = 0; i < 8; ++i) {
        (void)hipFree(d->d_wb3[i]);
        (void)hipFree(d->d_wwino[i]);
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
    (void)hipFree(d->d_pad);
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
    d->profile_only = on >= 2 ? on - 2 : -1;
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

int axt_cnn_front_frames(axt_detector *det, const float *d_frames, int T_all, int H, int W, int t0, int n_frames,
                         const int32_t *h_tile_yx, int n_tiles, int item0, void *stream)
{
    AXT_REQUIRE(det && d_frames && h_tile_yx, "null argument");
    AXT_REQUIRE(n_tiles >= 1 && n_tiles <= 256, "n_tiles %d out of range [1,256]", n_tiles);
    AXT_REQUIRE(t0 >= 0 && n_frames >= 0 && t0 + n_frames + 4 <= T_all, "frames [%d,%d) + context exceed T_all=%d",
                t0, t0 + n_frames, T_all);
    const int n_items = n_frames * n_tiles;
    AXT_REQUIRE(item0 >= 0 && item0 + n_items <= det->max_batch, "items [%d,%d) exceed the detector's max_batch %d", item0,
                item0 + n_items, det->max_batch);
    TileList tl;
    tl.n = n_tiles;
    for (int k = 0; k < n_tiles; ++k) {
        const int ty = h_tile_yx[2 * k], tx = h_tile_yx[2 * k + 1];
        AXT_REQUIRE(ty >= 0 && tx >= 0 && ty * AXT_TILE < H && tx * AXT_TILE < W, "tile %d (%d,%d) outside %dx%d", k,
                    ty, tx, H, W);
        tl.yx[2 * k] = (short)ty;
        tl.yx[2 * k + 1] = (short)tx;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int cb = 0; cb < n_items; cb += kChunkB) {
        const int nbb = (n_items - cb < kChunkB) ? n_items - cb : kChunkB;
        for (int c = 0; c < nbb; c += kChunkA) {
            const int nc = (nbb - c < kChunkA) ? nbb - c : kChunkA;
            const int rc = run_front_a(det, d_frames, H, W, t0, 1, cb + c, n_tiles, tl, nc, c, st);
            if (rc) return rc;
        }
        const int rc = run_front_b(det, nbb, det->d_act[4] + (size_t)(item0 + cb) * 80 * 32 * 32, st);
        if (rc) return rc;
    }
    return AXT_OK;
}

int axt_cnn_back(axt_detector *det, int n_items, float *d_yolo, void *stream)
{
    AXT_REQUIRE(det && d_yolo, "null argument");
    AXT_REQUIRE(n_items >= 0 && n_items <= det->max_batch, "%d items exceed the detector's max_batch %d", n_items, det->max_batch);
    if (n_items == 0) return AXT_OK;
    return run_back(det, n_items, d_yolo, (hipStream_t)stream);
}

}  // extern "C"

#ifdef AXT_WINO_STAMPS
// diagnostic build: the stamp sums of the last Winograd launch, u64 [1024 workgroups][8 waves][8]
extern "C" int axt_debug_wino_stamps(unsigned long long *h_out)
{
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_wino_stamps), sizeof(unsigned long long) * 1024 * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif
