// 256 x 256 x 64 phased bf16 GEMM tile for the large convolutions of the `precision="bf16"` path (conv3 forward,
// conv3 data gradient, conv2 forward at the bench shape): the structure cdna_hip_programming.md §5 calls the
// "256^2 8-phase template", rebuilt here on this engine's conventions (32x32x16 MFMA, 128-B swizzled K-contiguous LDS
// rows, LDS-DMA with the swizzle on the source address, zero-padded im2col). The 128 x 128 kernel of gemm_bf16.hip
// spends 44 % of its wave time in s_waitcnt / barrier (MfmaUtil 39 %): each K-tile is a serial
// [wait DMA -> barrier -> ds_read latency -> 16 MFMA]. Here:
//
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = acc[4][2] of 32 x 32 (128 accumulator VGPRs), one block per CU;
//   * a K-tile is four half-tiles of 128 rows x 128 B (A0, A1, B0, B1; half h of A holds, for each wave row wr, the 64
//     rows wr*128 + h*64 ..; half h of B, for each wave column wc, the 32 columns wc*64 + h*32 ..), two K-tiles of
//     them in LDS (128 KB);
//   * a K-tile is four phases, one output quadrant (A-half x B-half) each:   (A0,B0) (A0,B1) (A1,B1) (A1,B0);
//     a phase = [ds_read the fragments that changed | issue ONE half-tile of LDS-DMA for a later K-tile |
//     s_waitcnt lgkmcnt(0) | s_barrier | 8 MFMA | s_barrier];
//   * the DMA is never drained in the loop: one counted `s_waitcnt vmcnt(6)` per K-tile (three half-tiles stay in
//     flight across the barriers), raw `s_barrier` (a __syncthreads() would emit vmcnt(0));
//   * the two wave rows run one barrier apart (`if (wr == 1) s_barrier`), so that while one group of four waves is in
//     its MFMA section the other is in its read/stage section: each SIMD holds one wave of either group.
//
// Hazards (P = phase index, both groups; barriers are whole-block):
//   RAW  a half-tile issued for K-tile t+2 is retired by the vmcnt(6) of phase (t+1, q3), placed BEFORE that phase's
//        first barrier; its first ds_read is in phase (t+2, q0), i.e. after a barrier every waiter has passed — also
//        for the staggered group. B0 of K-tile t+1 is read one phase early (in (t, q3), to balance the LDS reads) and
//        is retired by an extra vmcnt(8) in (t, q2).
//   WAR  a slot read in phase P is restaged in phase P+1: the readers' lgkmcnt(0) sits before the first barrier of
//        phase P, which both groups pass before anyone issues phase P+1's DMA.
// Staging order (slot freed one phase earlier): (t,q0) A1 of t+1 | (t,q1) A0 of t+2 | (t,q2) B0 of t+2 | (t,q3) B1 of t+2.
// K-tiles past the end are still "staged" (out-of-range addresses return zeros or unused data) so the counts stay
// uniform; they are never multiplied.
//
// Tile sequence (round 3). One block per CU is launched and WALKS the tiles (virtual block L, L + G, ...; G a multiple of 8,
// so every tile stays on the XCD — the L2 — a one-block-per-tile launch would have given it). After a tile's last phase
// the LDS is free, so the next tile's prologue (14 DMA loads per wave) is issued BEFORE the epilogue and flies under it;
// the bias of the next tile's columns is fetched there too. In-kernel stamps (scripts/stamps_conv.py, conv3 forward, random
// operands): tile period 66.7 us = main loop 57.5 + prologue issue / MFMA drain 2.1 + epilogue 4.3 + waves re-joining 1.9
// + bias 0.7 + 0.2; the one-block-per-tile form paid a block dispatch, the loaders' set-up and the first DMA latency
// in front of every tile (~3 us more). The convolution forward / data gradient feed the MFMA (B, A) instead of (A, B): the
// accumulators hold C^T blocks, a lane owns 4 consecutive columns of one output row, and the epilogue packs and stores
// without transposing (SWAP below; bit-identical results, scripts/probe/swap_hash.py). What is left of the epilogue is
// the stores themselves: the CUs run in step, so 256 x 128 KB leave at the same time (a store pattern of whole 128-B lines
// per 8 lanes, tried as a timing experiment, moved the total by < 1 us).
#include "gemm_bf16_common.h"

using namespace goalnet;

namespace {

constexpr int T = 256;                  // block tile edge
constexpr int HALF_BYTES = 128 * ROWB;  // 16 KB
constexpr int LDS_BYTES = 2 * 4 * HALF_BYTES;

// local row lr (0..127) of half h -> row of the 256-row tile; GROUP = rows a wave owns per half (A: 64, B: 32)
template <int GROUP>
__device__ __forceinline__ int tile_row_of(int h, int lr) { return (lr / GROUP) * (2 * GROUP) + h * GROUP + (lr % GROUP); }

// ---- split operands ("bf16x6", fp32 products from 16-bit MFMAs) --------------------------------------------------------
// An fp32 value v is stored as three bf16 values hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): v = hi + mid + lo
// exactly (3 x 8 significand bits). A product a b is then the sum of nine exact 16-bit products; the three smallest
// (mid lo, lo mid, lo lo: < 2^-23 |a b|) are dropped and the other six are six K-segments of ONE 16-bit GEMM with fp32
// accumulation — the same kernel, 6 x the K-tiles, on operands stored [hi | mid | lo] along the channel axis. Segment s reads
// part (map >> 4 s) & 15 of its operand; the small terms come first:
//        s:   0          1         2         3          4          5
//   A part:  mid        hi        lo        hi         mid        hi          SEGMAP_A = 0x010201
//   B part:  mid        lo        hi        mid        hi         hi          SEGMAP_B = 0x001021
constexpr unsigned SEGMAP_A = 0x010201u, SEGMAP_B = 0x001021u;
// "fp16x3": two fp16 parts [hi | mid] of the value scaled by a power of two (11 + 11 significand bits), three partial products
//        s:   0          1         2
//   A part:  hi         mid       hi           SEGMAP_A3 = 0x010
//   B part:  mid        hi        hi           SEGMAP_B3 = 0x001
constexpr unsigned SEGMAP_A3 = 0x010u, SEGMAP_B3 = 0x001u;
// The segments are the FAST index of the reduction: virtual chunk = NSEG * (64-channel chunk) + s (convolutions), K-tile =
// NSEG * (64-pixel tile) + s (weight gradient), so that the products of one piece of the operands run back to back and its
// parts are fetched from HBM once (with the segment as the slow index every part was streamed two to three times:
// conv3's weight gradient 78.9 ms).
// first stored channel of virtual 64-channel chunk `chunk` (segC channels per part); past the end: >= the stored channels or harmless
template <int NSEG>
__device__ __forceinline__ int seg_channel(int chunk, int segC, unsigned segmap) {
    const int cc = chunk / NSEG, seg = chunk - NSEG * cc;
    return (int)((segmap >> (4 * seg)) & 15u) * segC + cc * BKH;
}

// K-contiguous bf16 matrix X[rows][K] (weights [Cout][9*Cin]); wave w issues pieces 2w, 2w+1 of a half-tile
template <int GROUP, int NSEG = 0>
struct KCLoader256 {
    // convC > 0: K-tiles in (channel chunk, tap) order over rows of 9 x convC values. segC > 0 (split operands, SegMap below):
    // convC = 3 segC stored channels [hi | mid | lo] per tap, 6 segC virtual ones
    struct P { const __hip_bfloat16* x; int64_t ld; int rows; int convC; int segC; unsigned segmap; };
    static constexpr bool TR = false;
    static constexpr int NSEGS = NSEG;
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[2][2];
    int wave, convC, segC;
    unsigned segmap;
    __device__ KCLoader256(const P& p, int row0, int tid) {
        convC = p.convC; segC = p.segC; segmap = p.segmap;
        const int nrows = p.rows - row0 < T ? p.rows - row0 : T;
        rx = make_rsrc(p.x + (int64_t)row0 * p.ld, clamp_u32((int64_t)nrows * p.ld * 2));
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = (wave * 2 + i) * 8 + (lane >> 3);
                const int lc = (lane & 7) ^ ((rl >> 1) & 7);
                const int tr = tile_row_of<GROUP>(h, rl);
                voff[h][i] = tr < nrows ? (unsigned)(((int64_t)tr * p.ld + lc * 8) * 2) : OOB;
            }
    }
    __device__ __forceinline__ void issue(int kt, int h, char* l) const {
        // the k range of K-tile kt: plain = [64 kt, +64); convolution order = tap (kt % 9), channels 64 (kt / 9) .. +63
        unsigned koff = convC > 0 ? (unsigned)((kt % 9) * convC + (kt / 9) * BKH) : (unsigned)kt * BKH;
        // split operands: convolution order as above over 6 x the chunks; plain order = rows [hi K | mid K | lo K], K-tile = 6 * (64-column tile) + segment
        if constexpr (NSEG > 0) koff = convC > 0 ? (unsigned)((kt % 9) * convC + seg_channel<NSEG>(kt / 9, segC, segmap)) : (unsigned)seg_channel<NSEG>(kt, segC, segmap);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 8 * ROWB, voff[h][i], koff * 2);
    }
};

// im2col of a zero-padded bf16 NHWC tensor (buffer convention of gemm_bf16.hip: `x` = padded pixel 0, W+3 zero guard pixels
// in front and behind): row m = (n,h,w) of the output grid, k = (kh,kw,ci); no validity masks.
template <int GROUP, int NSEG = 0>
struct ConvAPadLoader256 {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; int segC; unsigned segmap; };   // segC > 0: C = 2 or 3 segC stored channels (split operands)
    static constexpr bool TR = false;
    static constexpr int NSEGS = NSEG;
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[2][2];
    int Wp2, C, wave, segC;
    unsigned segmap;
    __device__ ConvAPadLoader256(const P& p, int row0, int tid) {
        Wp2 = p.W + 2; C = p.C; segC = p.segC; segmap = p.segmap;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        const int hw = p.H * p.W;
        const double inv_hw = 1.0 / (double)hw, inv_w = 1.0 / (double)p.W;       // rows and pixels stay below 2^31 (launcher)
        const int n0 = fast_div(row0, inv_hw), r0 = row0 - n0 * hw, h0 = fast_div(r0, inv_w), w0 = r0 - h0 * p.W;
        const int64_t pm0 = ((int64_t)n0 * (p.H + 2) + h0 + 1) * Wp2 + w0 + 1;      // padded index of the tile's first pixel
        const int G = Wp2 + 1;
        // 256 consecutive output pixels span < 256 + 2 * (rows crossed + frames crossed * (W+2)) padded pixels. Only an
        // upper bound is needed: valid rows stay inside the tensor by construction, invalid ones use OOB offsets.
        rx = make_rsrc(p.x + (pm0 - G) * p.C, clamp_u32((int64_t)(T + 2 * (T / p.W + 2) + 2 * Wp2 * (T / hw + 2) + 2 * G) * p.C * 2));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = (wave * 2 + i) * 8 + (lane >> 3);
                const int lc = (lane & 7) ^ ((rl >> 1) & 7);
                const int64_t m = (int64_t)row0 + tile_row_of<GROUP>(h, rl);
                if (m < p.M) {
                    const int n = fast_div((int)m, inv_hw), rr = (int)m - n * hw, hh = fast_div(rr, inv_w), w = rr - hh * p.W;
                    const int64_t pm = ((int64_t)n * (p.H + 2) + hh + 1) * Wp2 + w + 1;
                    voff[h][i] = (unsigned)(((pm - pm0) * p.C + lc * 8) * 2);
                } else {
                    voff[h][i] = OOB;
                }
            }
    }
    __device__ __forceinline__ void issue(int kt, int h, char* l) const {
        // K-tile order = (64-channel chunk, tap): the nine taps of one chunk touch almost the same pixels, so they re-hit
        // L2. With taps outermost the reuse distance was a whole sweep over the channels of every resident block (8 MB per
        // XCD at C = 512) and the data gradient re-fetched its input 5x from the fabric (profiles/r01_bf16_traffic.md).
        const int chunk = kt / 9, tap = kt - 9 * chunk;
        int ci = chunk * BKH;
        if constexpr (NSEG > 0) ci = seg_channel<NSEG>(chunk, segC, segmap);
        const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;                  // tap / 3 for tap < 16
        // past the last K-tile (ci >= C) the offset only has to stay harmless: the range check turns it into zeros
        const unsigned s0 = ci < C ? (unsigned)(((kh * Wp2 + kw) * C + ci) * 2) : OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 8 * ROWB, voff[h][i], s0);
    }
};

// ---- row-contiguous ("transposed") operands for the weight gradient: the reduction index (pixel) is the slow index in
// memory. Half-tile image = [64 k-rows][128 columns] bf16 (256-B rows, 16-B chunks XOR-swizzled by 2 * (krow & 7) on the DMA
// source address), read with ds_read_b64_tr_b16 exactly as in gemm_bf16.hip. Half h holds tile columns h*128 .. h*128+127;
// fragment u (0..3) of a half is made of the 16-column units {u, u + 4}: columns 16u + (i & 15) + 64 (i >> 4), i = 0..31.
constexpr int TROWB = 256;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ int tr_chunk(int krow, int chunk) { return chunk ^ (2 * (krow & 7)); }

__device__ __forceinline__ bf16x8 read_frag_tr(const char* half, int u, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = 2 * (u + 4 * (g & 1)) + (p >> 1);
    s16x4 v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int kr = 16 * ks + 8 * (g >> 1) + 4 * t + q;
        const char* a = half + kr * TROWB + tr_chunk(kr, chunk) * 16 + (p & 1) * 8;
        v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
    }
    return bf16x8{v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
}

// The same fragment through inline asm, for the phased loop below. hipcc cannot tell what a `ds_read_tr` builtin may alias
// and puts `s_waitcnt vmcnt(0)` in front of every group of them while LDS-DMA is in flight — a full drain of the staging
// pipeline in every phase (ISA of the weight-gradient kernel: 30 drains per two K-tiles). As asm the reads carry no such
// wait; what the compiler no longer provides is done by hand: `s_waitcnt lgkmcnt(0)` before the phase's MFMAs (GN_PHASE),
// all statements volatile (issued in source order). Per-lane byte address of fragment u, read t (t = 0, 1: k rows 4t + q
// of each group of 8), K-step and slot as immediates; the K-tile parity (64 KB apart) is flipped INTO the address registers
// so that every immediate stays below 64 KB.
__device__ __forceinline__ unsigned tr_lane_addr(unsigned lds0, int u, int t, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = 2 * (u + 4 * (g & 1)) + (p >> 1);
    const int kr = 8 * (g >> 1) + 4 * t + q;
    return lds0 + (unsigned)(kr * TROWB + tr_chunk(kr, chunk) * 16 + (p & 1) * 8);
}
template <int OFF>
__device__ __forceinline__ bf16x8 rd_tr(const unsigned (&ad)[2]) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    s16x4 v0, v1;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v0) : "v"(ad[0]), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v1) : "v"(ad[1]), "n"(OFF) : "memory");
    return bf16x8{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
}
#define GN_RD_TR4(DST, AD, S)                                                                               \
    DST[0] = rd_tr<(S) * HALF_BYTES + 0 * 16 * TROWB>(AD); DST[1] = rd_tr<(S) * HALF_BYTES + 1 * 16 * TROWB>(AD);  \
    DST[2] = rd_tr<(S) * HALF_BYTES + 2 * 16 * TROWB>(AD); DST[3] = rd_tr<(S) * HALF_BYTES + 3 * 16 * TROWB>(AD);

// plain matrix X[kred][cols] (dy_pad [pixels][Cout]): re-based every K-tile; rows past kred / columns past `cols` read 0
template <int NSEG>
struct MCLoader256T {
    // NSEG > 0 (split operands, segKT > 0): K-tile kt = 6 * (64-row tile) + segment s; segment s reads the columns of part
    // (segmap >> 4 s) & 15, segC columns further right each (ld = 3 segC)
    struct P { const __hip_bfloat16* x; int64_t ld; int cols; int64_t kred; int segKT; int segC; unsigned segmap; };
    static constexpr bool TR = true;
    static constexpr int NSEGS = NSEG;
    const __hip_bfloat16* x;
    int64_t ld, kred;
    int wave, segKT, segC;
    unsigned segmap;
    unsigned voff[2][2];
    __device__ MCLoader256T(const P& p, int col0, int tid) {
        x = p.x; ld = p.ld; kred = p.kred; segKT = p.segKT; segC = p.segC; segmap = p.segmap;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = (wave * 2 + i) * 4 + (lane >> 4);
                const int col = col0 + h * 128 + tr_chunk(kr, lane & 15) * 8;
                voff[h][i] = col < p.cols ? (unsigned)(((int64_t)kr * p.ld + col) * 2) : OOB;
            }
    }
    __device__ __forceinline__ void issue(int kt, int h, char* l) const {
        int kk = kt;
        unsigned soff = 0u;
        if constexpr (NSEG > 0) { kk = kt / NSEG; soff = ((segmap >> (4 * (kt - NSEG * kk))) & 15u) * (unsigned)segC * 2u; }
        const int64_t kbase = (int64_t)kk * BKH;                             // past the last tile: empty range below
        const int64_t nk = kred - kbase < BKH ? kred - kbase : BKH;          // <= 0 past the end: empty range, zeros
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + kbase * ld, clamp_u32(nk * ld * 2));
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 4 * TROWB, voff[h][i], soff);
    }
};

typedef MCLoader256T<0> MCLoader256;

// B operand of the weight gradient on the zero-padded pixel grid: B(col = (tap, ci), k = pm) = x_pad[pm + shift(tap)][ci]
// (constant pixel shift per tap; pad pixels contribute nothing because dy_pad is zero there). `x` = padded pixel 0 of a
// buffer with W+3 zero guard pixels in front and behind.
template <int NSEG>
struct ConvWgradBLoader256T {
    // NSEG > 0 (split operands, segKT > 0): Cs = 3 C stored channels per pixel, K-tile kt = 6 * (64-pixel tile) + segment
    struct P { const __hip_bfloat16* x; int Wp2, C; int64_t Mp; int Cs; int segKT; unsigned segmap; };
    static constexpr bool TR = true;
    static constexpr int NSEGS = NSEG;
    const __hip_bfloat16* x;
    int64_t Mp;
    int C, G, wave, Cs, segKT;
    unsigned segmap;
    unsigned voff[2][2];
    __device__ ConvWgradBLoader256T(const P& p, int col0, int tid) {
        x = p.x; C = p.C; G = p.Wp2 + 1; Mp = p.Mp; Cs = p.Cs; segKT = p.segKT; segmap = p.segmap;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = (wave * 2 + i) * 4 + (lane >> 4);
                const int col = col0 + h * 128 + tr_chunk(kr, lane & 15) * 8;
                if (col < 9 * p.C) {
                    const int tap = col / p.C, ci = col - tap * p.C;
                    const int kh = tap / 3, kw = tap - 3 * kh;
                    voff[h][i] = (unsigned)(((kr + kh * p.Wp2 + kw) * p.Cs + ci) * 2);      // relative to pixel (kbase - G)
                } else {
                    voff[h][i] = OOB;
                }
            }
    }
    __device__ __forceinline__ void issue(int kt, int h, char* l) const {
        int kk = kt;
        unsigned soff = 0u;
        if constexpr (NSEG > 0) { kk = kt / NSEG; soff = ((segmap >> (4 * (kt - NSEG * kk))) & 15u) * (unsigned)C * 2u; }
        const int64_t kbase = (int64_t)kk * BKH;
        // K-tiles past the pixel grid (staged only to keep the DMA counts uniform) get an empty range: zeros, no access
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (kbase - G) * Cs, kbase < Mp ? (uint32_t)((BKH + 2 * G) * Cs * 2) : 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 4 * TROWB, voff[h][i], soff);
    }
};
typedef ConvWgradBLoader256T<0> ConvWgradBLoader256;

__device__ __forceinline__ bf16x8 read_frag(const char* half, int row0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(half + kc_boff(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// The compiler may move MFMA builtins (pure register operations) across a barrier builtin, which collapses the phase
// structure (seen in the ISA: 4 / 3 / 17 / 8 MFMAs per phase instead of 8 each); pinning them with data dependencies on
// the accumulators would make every wave wait for its last MFMA to COMPLETE before it reaches the barrier (+70..100
// cycles per phase). The MFMAs and the barriers are therefore `asm volatile`: issued exactly where written, and a wave
// reaches the closing barrier while its last MFMAs drain. Hazards the compiler no longer covers are handled here:
// accumulating back to back into the same registers needs no wait states (CDNA3/4 ISA, XDL write -> XDL SrcC, same
// vDst); the epilogue reads the accumulators only after an explicit drain (s_nop block after the loop).
template <bool F16>
__device__ __forceinline__ void mfma_pinned(f32x16& c, const bf16x8& a, const bf16x8& b) {
    // the two opcodes share operand / accumulator layouts and issue rate; the 16-bit patterns in a, b are read as fp16 or bf16
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define GN_PHASE(ACC0, ACC1, AF, BF)                                                                        \
    if constexpr (TRA || TRB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* the asm fragment reads (rd_tr) */ \
    asm volatile("s_barrier\n\ts_setprio 1" ::: "memory");                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                      \
        if constexpr (SWAP) { mfma_pinned<F16>(ACC0, BF[ks], AF[0][ks]); mfma_pinned<F16>(ACC1, BF[ks], AF[1][ks]); } \
        else { mfma_pinned<F16>(ACC0, AF[0][ks], BF[ks]); mfma_pinned<F16>(ACC1, AF[1][ks], BF[ks]); }      \
    }                                                                                                       \
    asm volatile("s_setprio 0\n\ts_barrier" ::: "memory");

// roles whose launch is persistent (one block per CU walking the tiles); see launch_256_t for the measurements
constexpr bool walks_tiles(int role) { return role == 0 || role == 1 || role == 4; }

#ifndef GN_SWAP_OPERANDS
#define GN_SWAP_OPERANDS 1          // 0: the (A, B) operand order with the transposing epilogue (A/B builds only)
#endif
#ifdef GN_STAMPS
// Diagnostic build only (-DGN_STAMPS, loaded through GOALNET_LIB_PATH by scripts/ablate_conv.py): wave 0 of every block
// stamps the 100 MHz real-time counter at six points of every tile (0 tile start, 1 bias staged, 2 first operands landed,
// 3 main loop done, 4 next prologue issued + MFMAs drained, 5 stores issued); the stamps go to a buffer nothing else reads.
__device__ unsigned long long gn_stamps[6 * 65536];
#define GN_STAMP(I) if (tid == 0 && vb < 65536) gn_stamps[6 * vb + (I)] = __builtin_amdgcn_s_memrealtime();
#else
#define GN_STAMP(I)
#endif

// ROLE only separates the symbols (0: convolution forward, 1: data gradient) so that profiles list them apart
template <class AL, class BL, int ROLE, bool F16>
__global__ __launch_bounds__(512, 1) void gemm_bf16_256_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                              int tiles_m, int tiles_n, int m_fast, int ktiles_total,
                                                              int ktiles_per_split, int total_blocks) {
    constexpr bool TRA = AL::TR, TRB = BL::TR;      // operand forms: K-contiguous rows (false) or row-contiguous / transposed-read (true)
    // SWAP: the MFMA takes (B fragment, A fragment), i.e. it computes the 32 x 32 block of C^T. Products and the order of the
    // K sum are those of (A, B) — the result is bit-identical — but a lane now holds 4 consecutive COLUMNS of one row of C per
    // register group (row = lane & 31) instead of 4 consecutive rows of one column: the epilogue stores straight from the
    // accumulators, without the quad transposes that were half of its ~1100 VALU instructions per wave.
    // Transposed-read operands keep their row / column permutation: it applies to the index inside the 32-block, which SWAP
    // moves from the lane to the register (columns) and from the register to the lane (rows).
    // Only the roles with a 16-bit result use it (same box, bench step: conv forward -5.8 %, data gradient -3.0 %, linear5 dX
    // -2 % together with the tile walk). The fp32 slab roles keep (A, B): their transposed stores are whole 128-B lines per 8
    // lanes, the swapped form writes 32-B pieces of 32 rows per instruction, and linear5's dW (256 KB of fp32 per 16 K-tiles)
    // lost 7 % with it; the weight gradient and linear5's forward did not move.
    constexpr bool SWAP = GN_SWAP_OPERANDS && (ROLE == 0 || ROLE == 1 || ROLE == 4);
    // WALK: the block goes on to the virtual blocks L + gridDim.x, ... (launch_256_t launches one block per CU for these roles);
    // the other roles run one tile per block and compile without the hand-over (no register pressure from it, no spills).
    constexpr bool WALK = walks_tiles(ROLE);
    // fp16x3 (three segments): the operands were scaled by powers of two; *ep.oscale = the exponent that undoes both (ldexpf: exact, and
    // no product of scales that could overflow). The bias then cannot ride in the accumulators: it is added after the unscaling (LATE_BIAS).
    constexpr bool OSC = AL::NSEGS == 3;
    constexpr bool LATE_BIAS = OSC && ROLE == 0;
    int osc_k = 0;
    if constexpr (OSC) { if (ep.oscale) osc_k = *ep.oscale; }
    extern __shared__ __attribute__((aligned(16))) char lds[];          // [2 K-tiles][A0, A1, B0, B1][128 rows][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // (tm, tn, split) of virtual block L of `total_blocks`. The launch is PERSISTENT: one block per CU walks the virtual
    // blocks L = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8, so a virtual block keeps the XCD the
    // plain launch gave it, and with it the L2 reuse the mappings below are built for).
    auto coords = [&](int L, int& tm_, int& tn_, int& split_) {
        split_ = 0;
        if (ROLE == 2 || ROLE == 3) {
            // weight gradient (and the linear5 forward): few output tiles (18 / 8), many K splits. All tiles of a split read the same pixels, so they run on
            // ONE XCD back to back (blocks L, L + 8, ... share an XCD): XCD x takes splits x, x + 8, ... and walks their tiles.
            // Spread over all XCDs, every XCD fetched every pixel: 72 GB/launch from the fabric against 9 GB of operands.
            const int tiles = tiles_m * tiles_n, nsplit = total_blocks / tiles;
            const int xcd = L & 7, j = L >> 3;
            if ((nsplit & 7) == 0) { split_ = xcd + 8 * (j / tiles); const int t = j % tiles; tm_ = t % tiles_m; tn_ = t / tiles_m; }
            else { split_ = L / tiles; const int t = L % tiles; tm_ = t % tiles_m; tn_ = t / tiles_m; }
        } else {
            const unsigned v = xcd_remap((unsigned)L, (unsigned)(tiles_m * tiles_n));
            if (m_fast) { tm_ = (int)(v % (unsigned)tiles_m); tn_ = (int)(v / (unsigned)tiles_m); }
            else        { tn_ = (int)(v % (unsigned)tiles_n); tm_ = (int)(v / (unsigned)tiles_n); }
        }
    };
    auto slot = [&](int t, int s) -> char* { return lds + ((t & 1) * 4 + s) * HALF_BYTES; };     // s: 0 A0, 1 A1, 2 B0, 3 B1
    // prologue of a tile: K-tile 0 complete, K-tile 1 without its A1 (phase (0,q0) stages that)
#define GN_PROLOGUE()                                                                                       \
    al.issue(kt0 + 0, 0, slot(0, 0)); bl.issue(kt0 + 0, 0, slot(0, 2)); bl.issue(kt0 + 0, 1, slot(0, 3)); al.issue(kt0 + 0, 1, slot(0, 1)); \
    bl.issue(kt0 + 1, 0, slot(1, 2)); al.issue(kt0 + 1, 0, slot(1, 0)); bl.issue(kt0 + 1, 1, slot(1, 3));

    // bias of the tile's 256 columns (convolution forward). SWAP: register e of acc[.][j] is output column
    // 32 j + (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the wave's 64 — 32 different values per lane: thread t fetches column t
    // while the previous tile's epilogue runs (one register), the tile start puts it into LDS (1 KB beside the 128 KB of
    // operands) and the accumulators are initialised from there. Otherwise all 16 registers of acc[.][j] are column
    // 32 j + (lane & 31): two registers.
    // two copies, used alternately by consecutive tiles: a wave that is already at the top of the next tile writes the OTHER copy while a
    // slower wave may still be reading this tile's in its epilogue (LATE_BIAS); waves are never more than one tile apart (the phases'
    // barriers keep them together inside a tile)
    __shared__ __attribute__((aligned(16))) float bias_lds[2][T];
    int bias_par = 0;
    float bias_t = 0.f, bias_col[2] = {0.f, 0.f};
#define GN_LOAD_BIAS()                                                                                      \
    if constexpr (ROLE == 0) {                                                                              \
        if constexpr (SWAP) {                                                                               \
            bias_t = 0.f;                                                                                   \
            if (ep.bias && tid < T && tn * T + tid < ep.cols) bias_t = ep.bias[tn * T + tid];               \
        } else {                                                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                 \
                const int colj = tn * T + wc * 64 + j * 32 + (lane & 31);                                   \
                bias_col[j] = ep.bias && colj < ep.cols ? ep.bias[colj] : 0.f;                              \
            }                                                                                               \
        }                                                                                                   \
    }

    int vb = (int)blockIdx.x;
    int tm, tn, split;
    coords(vb, tm, tn, split);
    GN_LOAD_BIAS()
    AL al(ap, tm * T, tid);
    BL bl(bp, tn * T, tid);
    int kt0 = split * ktiles_per_split;                                 // split-K: this block reduces K-tiles [kt0, kt0 + ktiles)
    int ktiles = min(ktiles_total, kt0 + ktiles_per_split) - kt0;
    GN_PROLOGUE()
#define GN_WAIT_KTILE0() asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (SWAP) GN_WAIT_KTILE0()
  for (;;) {
    GN_STAMP(0)
    if constexpr (SWAP && ROLE == 0) {
        if (tid < T) bias_lds[bias_par][tid] = bias_t;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    GN_STAMP(1)
    // K-tile 0 = the 8 oldest of the 14 DMA loads of the prologue: it has landed when at most 6 vector-memory operations
    // are outstanding (they complete in issue order; anything issued later — spill traffic, epilogue stores — only makes the
    // counted wait stricter). SWAP: the wave has already waited (GN_WAIT_KTILE0 before the loop / before the previous tile's
    // stores, so that it does not wait for those stores' acknowledgements here: 1.5 us per tile in the stamps).
    if constexpr (!SWAP) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    // The accumulators start at the bias, so the epilogue has no add left.
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if constexpr (SWAP) {
            f32x16 bj;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (ROLE == 0 && !LATE_BIAS) b4 = *reinterpret_cast<const float4*>(&bias_lds[bias_par][wc * 64 + j * 32 + 8 * g + 4 * (lane >> 5)]);
                bj[4 * g] = b4.x; bj[4 * g + 1] = b4.y; bj[4 * g + 2] = b4.z; bj[4 * g + 3] = b4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = bj;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = bias_col[j];
        }
    }
    GN_STAMP(2)
    if (wr == 1) asm volatile("s_barrier" ::: "memory");                 // stagger the two wave rows by one barrier

    // fragment registers: a = the current A half; bx / by = B0 / B1 of even K-tiles and B1 / B0 of odd ones: the B0
    // fragments of the NEXT K-tile are read in phase q3 into the registers B1 has just left, so every phase issues at most
    // 8 ds_read_b128 per wave (8 | 4 | 8 | 4) and the LDS time of a phase stays within one MFMA section.
    bf16x8 a[2][4], bx[4], by[4];
    // fragment reads: K-contiguous operands by pointer (compiler-scheduled ds_read_b128), row-contiguous ones by rd_tr
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    unsigned adA[2][2], adB[2];                      // [f][t], [t]: K-tile parity folded in (GN_FLIP)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        adA[0][t] = tr_lane_addr(lds0, 2 * wr, t, lane); adA[1][t] = tr_lane_addr(lds0, 2 * wr + 1, t, lane);
        adB[t] = tr_lane_addr(lds0, wc, t, lane);
    }
#define GN_FLIP_A() { adA[0][0] ^= 4u * HALF_BYTES; adA[0][1] ^= 4u * HALF_BYTES; adA[1][0] ^= 4u * HALF_BYTES; adA[1][1] ^= 4u * HALF_BYTES; }
#define GN_FLIP_B() { adB[0] ^= 4u * HALF_BYTES; adB[1] ^= 4u * HALF_BYTES; }
    // S = slot of the half inside its K-tile (0 A0, 1 A1, 2 B0, 3 B1); TT = the K-tile (pointer form only)
#define GN_RD_A(TT, S)                                                                                      \
    if constexpr (TRA) { GN_RD_TR4(a[0], adA[0], S) GN_RD_TR4(a[1], adA[1], S) }                            \
    else {                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                    \
            _Pragma("unroll") for (int f = 0; f < 2; ++f) a[f][ks] = read_frag(slot(TT, S), wr * 64 + f * 32, ks, lane); \
    }
#define GN_RD_B(DST, TT, S)                                                                                 \
    if constexpr (TRB) { GN_RD_TR4(DST, adB, S) }                                                           \
    else { _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) DST[ks] = read_frag(slot(TT, S), wc * 32, ks, lane); }
    GN_RD_B(bx, 0, 2)

#define GN_KTILE(TT, B0R, B1R)                                                                              \
    {                                                                                                       \
        const int t_ = (TT);                                                                                \
        /* q0: (A0, B0) */                                                                                  \
        GN_RD_A(t_, 0)                                                                                      \
        al.issue(kt0 + t_ + 1, 1, slot(t_ + 1, 1));                                                               \
        GN_PHASE(acc[0][0], acc[1][0], a, B0R);                                                             \
        /* q1: (A0, B1) */                                                                                  \
        GN_RD_B(B1R, t_, 3)                                                                                 \
        bl.issue(kt0 + t_ + 2, 0, slot(t_, 2));                                                                   \
        GN_PHASE(acc[0][1], acc[1][1], a, B1R);                                                             \
        /* q2: (A1, B1); B0 of the next K-tile must have landed one phase before q3 reads it */             \
        GN_RD_A(t_, 1)                                                                                      \
        if constexpr (TRA) GN_FLIP_A()                                                                      \
        al.issue(kt0 + t_ + 2, 0, slot(t_, 0));                                                                   \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                                    \
        GN_PHASE(acc[2][1], acc[3][1], a, B1R);                                                             \
        /* q3: (A1, B0); the counted wait that retires K-tile t+1 */                                        \
        if constexpr (TRB) GN_FLIP_B()                                                                      \
        GN_RD_B(B1R, t_ + 1, 2)                                                                             \
        bl.issue(kt0 + t_ + 2, 1, slot(t_, 3));                                                                   \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                    \
        GN_PHASE(acc[2][0], acc[3][0], a, B0R);                                                             \
    }

#pragma unroll 1
    for (int t = 0; t < ktiles; t += 2) {
        GN_KTILE(t, bx, by);
        if (t + 1 < ktiles) GN_KTILE(t + 1, by, bx);
    }
    GN_STAMP(3)
    if (wr == 0) asm volatile("s_barrier" ::: "memory");
    // Every wave has passed the closing barrier of the last phase: all fragment reads of this tile are done and the LDS is
    // free. The NEXT virtual block's prologue is issued now, so that its DMA flies under this tile's epilogue (the stray
    // stagings past this tile's last K-tile were issued earlier by the same wave into the same rows: they land first).
    const int tm_out = tm, tn_out = tn, split_out = split;
    const int vb_next = vb + (int)gridDim.x;
    const bool more = WALK && vb_next < total_blocks;
    if (more) {
        coords(vb_next, tm, tn, split);
        al = AL(ap, tm * T, tid);
        bl = BL(bp, tn * T, tid);
        kt0 = split * ktiles_per_split;
        ktiles = min(ktiles_total, kt0 + ktiles_per_split) - kt0;
        GN_PROLOGUE()
        GN_LOAD_BIAS()
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // no DMA may land after the block has released its LDS
    }
    // the last MFMAs (16 passes) must have written back before the epilogue reads the accumulators
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));                    // no DMA may land after the block has released its LDS

    GN_STAMP(4)
    do {                                            // the epilogue of tile (tm_out, tn_out, split_out); `break` = done
    const int tm = tm_out, tn = tn_out, split = split_out;
    // epilogue (conv forward: bias + ReLU; data gradient: raw; weight gradient / linear5 forward: raw split-K slab).
    // A lane of the 32x32 accumulator holds 4 consecutive ROWS of one column per register group; quad_transpose4
    // (gemm_common.h) turns that into 4 consecutive COLUMNS of one row, i.e. one 16-byte store per lane and whole 128-B lines
    // per quad row: 32 store instructions per wave instead of 128 (the tail of a 256 x 256 fp32 tile is store-issue-bound).
    const int r = lane & 31, hh = lane >> 5;
    // what is left to do per value: ROLE 0 (conv forward) clamps at 0 when ReLU is on (its bias is already in); the
    // other roles store the accumulator as it is (their callers pass no bias and no ReLU)
    auto fin = [&](float v, float b = 0.f) -> float {
        if constexpr (OSC) v = ldexpf(v, osc_k);
        if constexpr (LATE_BIAS) v += b;
        if constexpr (ROLE == 0) return ep.relu ? fmaxf(v, 0.f) : v;
        else return v;
    };
    // LATE_BIAS: the four bias values of register group g of acc[.][ni] (this tile's copy of bias_lds; the next tile writes the other one)
    auto bias4 = [&](int ni, int g) -> float4 {
        if constexpr (LATE_BIAS) return *reinterpret_cast<const float4*>(&bias_lds[bias_par][wc * 64 + ni * 32 + 8 * g + 4 * (lane >> 5)]);
        else return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float* outp = ep.out + (ep.slab_stride > 0 ? (int64_t)split * ep.slab_stride : 0);
    if constexpr (SWAP) {
        // lane (r, hh) holds, for register group g of acc[mi][ni], columns 32 ni + 8 g + 4 hh .. + 3 of row 32 (mi) + r.
        // 16-bit result: lanes l and l + 32 hold the two halves of the same 8 columns; ONE v_permlane32_swap per packed
        // dword gives the lower lane the whole of group 2 gp and the upper lane the whole of group 2 gp + 1: a 16-byte
        // store per lane, 16 store instructions per wave, ~250 VALU instructions instead of ~1100.
        __hip_bfloat16* o16 = reinterpret_cast<__hip_bfloat16*>(ep.out16);
        if (o16) {
            // all values first (registers only), then the wait for the next tile's first operands, then the stores
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 pk[4][2][2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned x[2], y[2];
                        const float4 b0 = bias4(ni, 2 * gp), b1 = bias4(ni, 2 * gp + 1);
                        const float bb0[4] = {b0.x, b0.y, b0.z, b0.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const unsigned p0 = pack2_h16<F16>(fin(acc[mi][ni][8 * gp + 2 * d], bb0[2 * d]), fin(acc[mi][ni][8 * gp + 2 * d + 1], bb0[2 * d + 1]));
                            const unsigned p1 = pack2_h16<F16>(fin(acc[mi][ni][8 * gp + 4 + 2 * d], bb1[2 * d]), fin(acc[mi][ni][8 * gp + 4 + 2 * d + 1], bb1[2 * d + 1]));
                            // lower lane: (its group 2gp, the upper lane's 2gp); upper lane: (the lower lane's 2gp+1, its 2gp+1)
                            const auto sw = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
                            x[d] = sw[0]; y[d] = sw[1];
                        }
                        pk[mi][ni][gp] = u32x4{x[0], x[1], y[0], y[1]};
                    }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) asm volatile("" : "+v"(pk[mi][ni][gp]));          // the packing stays above the wait
            GN_WAIT_KTILE0()
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int64_t row = (int64_t)tm * T + (TRA ? (mi >> 1) * 128 + 16 * (2 * wr + (mi & 1)) + (r & 15) + 64 * (r >> 4)
                                                           : wr * 128 + (mi >> 1) * 64 + (mi & 1) * 32 + r);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int col8 = tn * T + (TRB ? ni * 128 + 16 * wc + 64 * gp + 8 * hh : wc * 64 + ni * 32 + 16 * gp + 8 * hh);
                        if (row < ep.rows && col8 < ep.cols)                                           // cols % 8 == 0
                            *reinterpret_cast<u32x4*>(o16 + row * ep.ld + col8) = pk[mi][ni][gp];
                    }
            }
        } else {
            GN_WAIT_KTILE0()
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int64_t row = (int64_t)tm * T + (TRA ? (mi >> 1) * 128 + 16 * (2 * wr + (mi & 1)) + (r & 15) + 64 * (r >> 4)
                                                           : wr * 128 + (mi >> 1) * 64 + (mi & 1) * 32 + r);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = tn * T + (TRB ? ni * 128 + 16 * wc + 8 * (g & 1) + 4 * hh + 64 * (g >> 1) : wc * 64 + ni * 32 + 8 * g + 4 * hh);
                        const float4 b4 = bias4(ni, g);
                        if (row < ep.rows && col < ep.cols)                                            // cols % 4 == 0
                            *reinterpret_cast<float4*>(outp + row * ep.ld + col) =
                                make_float4(fin(acc[mi][ni][4 * g], b4.x), fin(acc[mi][ni][4 * g + 1], b4.y), fin(acc[mi][ni][4 * g + 2], b4.z),
                                            fin(acc[mi][ni][4 * g + 3], b4.w));
                    }
            }
        }
        GN_STAMP(5)
        break;
    }
    const int r4 = r & ~3;
    if constexpr (ROLE == 0 || ROLE == 1 || ROLE == 4) {
        // bf16 result (a conv output on its way to the max-pool, or the gradient wrt a BatchNorm output; both are only read
        // by HBM-bound passes): after the quad transpose
        // lanes l and l ^ 4 hold columns c..c+3 and c+4..c+7 of the same rows for two register groups (rows 8 apart);
        // they swap one packed group so that each stores 8 consecutive bf16 (16 B) of ONE row: 16 store instructions
        // per wave for the tile (the fp32 form needs 32) and half the bytes.
        if (ep.out16) {
            __hip_bfloat16* o16 = reinterpret_cast<__hip_bfloat16*>(ep.out16);
            const int hi = (lane >> 2) & 1;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col4 = tn * T + (TRB ? ni * 128 + 16 * wc + (r4 & 15) + 64 * (r4 >> 4) : wc * 64 + ni * 32 + r4);
                    const int col8 = col4 - 4 * hi;                                                    // first of this lane's 8 columns
                    const bool colok = col8 < ep.cols;                                                 // cols % 8 == 0
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int g = 2 * gp + k;
                            float n0 = acc[mi][ni][4 * g], n1 = acc[mi][ni][4 * g + 1], n2 = acc[mi][ni][4 * g + 2], n3 = acc[mi][ni][4 * g + 3];
                            quad_transpose4(n0, n1, n2, n3, lane);
                            pk[k][0] = pack2_h16<F16>(fin(n0), fin(n1));
                            pk[k][1] = pack2_h16<F16>(fin(n2), fin(n3));
                        }
                        // lane hi = 0 keeps group 2gp and gets the partner's 2gp (its columns + 4); hi = 1 keeps 2gp + 1
                        const unsigned s0 = hi ? pk[0][0] : pk[1][0], s1 = hi ? pk[0][1] : pk[1][1];
                        const unsigned q0 = (unsigned)__shfl_xor((int)s0, 4, 64), q1 = (unsigned)__shfl_xor((int)s1, 4, 64);
                        const unsigned m0 = hi ? pk[1][0] : pk[0][0], m1 = hi ? pk[1][1] : pk[0][1];
                        const int i = (lane & 3) + 8 * (2 * gp + hi) + 4 * hh;
                        const int64_t row = (int64_t)tm * T + (TRA ? (mi >> 1) * 128 + 16 * (2 * wr + (mi & 1)) + (i & 15) + 64 * (i >> 4)
                                                                   : wr * 128 + (mi >> 1) * 64 + (mi & 1) * 32 + i);
                        if (colok && row < ep.rows)
                            *reinterpret_cast<uint4*>(o16 + row * ep.ld + col8) = hi ? make_uint4(q0, q1, m0, m1) : make_uint4(m0, m1, q0, q1);
                    }
                }
            GN_STAMP(5)
            break;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = tn * T + (TRB ? ni * 128 + 16 * wc + (r4 & 15) + 64 * (r4 >> 4) : wc * 64 + ni * 32 + r4);   // first of 4 columns
            const bool colok = col < ep.cols;                                                                          // cols % 4 == 0
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float n0 = acc[mi][ni][4 * g], n1 = acc[mi][ni][4 * g + 1], n2 = acc[mi][ni][4 * g + 2], n3 = acc[mi][ni][4 * g + 3];
                quad_transpose4(n0, n1, n2, n3, lane);
                const int i = (lane & 3) + 8 * g + 4 * hh;
                const int64_t row = (int64_t)tm * T + (TRA ? (mi >> 1) * 128 + 16 * (2 * wr + (mi & 1)) + (i & 15) + 64 * (i >> 4)
                                                           : wr * 128 + (mi >> 1) * 64 + (mi & 1) * 32 + i);
                if (colok && row < ep.rows)
                    *reinterpret_cast<float4*>(outp + row * ep.ld + col) =
                        make_float4(fin(n0), fin(n1), fin(n2), fin(n3));
            }
        }
    GN_STAMP(5)
    } while (0);
    if (!more) break;
    vb = vb_next;
    bias_par ^= 1;
  }
#undef GN_PROLOGUE
#undef GN_WAIT_KTILE0
#undef GN_LOAD_BIAS
}

}  // namespace

#ifdef GN_STAMPS
extern "C" int goalnet_debug_stamps(unsigned long long* host_dst, int nblocks) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gn_stamps), sizeof(unsigned long long) * 6 * (size_t)nblocks);
}
#endif

namespace goalnet {

// one launch of gemm_bf16_256_kernel<AL, BL, ROLE, F16>: reserves the 128 KB of dynamic LDS once per instantiation
template <class AL, class BL, int ROLE, bool F16>
static int launch_256_t(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep, int64_t tiles_m,
                        int64_t tiles_n, int nsplit, int m_fast, int ktiles, int kps, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<AL, BL, ROLE, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { set_error("%s: cannot reserve %d bytes of LDS: %s", name, LDS_BYTES, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    GN_REQUIRE(tiles_m * tiles_n * nsplit < (1ll << 31), GOALNET_E_SHAPE, "%s: too many blocks", name);
    // persistent launch: one block per CU (128 KB of LDS: one fits) walks the virtual blocks; a multiple of 8 blocks keeps every
    // virtual block on the XCD a plain launch would have given it. GOALNET_PERSISTENT=0: one block per tile (A/B runs).
    static const int persistent_blocks = [] {
        const char* e = getenv("GOALNET_PERSISTENT");
        if (e && atoi(e) == 0) return 0;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return (e && atoi(e) > 1 ? atoi(e) : cus) & ~7;
    }();
    const int total = (int)(tiles_m * tiles_n * nsplit);
    // Measured per role (rocprofv3, bench step): the conv forward gains 4-5 %, linear5's dX 2-4 %, the data gradient nothing;
    // the split-K roles run few, long blocks (nothing to hide) and linear5's dW lost 3-6 %: those keep one block per tile.
    constexpr bool walk = walks_tiles(ROLE);
    const int grid = walk && persistent_blocks > 0 && total > persistent_blocks ? persistent_blocks : total;
    hipLaunchKernelGGL((gemm_bf16_256_kernel<AL, BL, ROLE, F16>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st,
                       ap, bp, ep, (int)tiles_m, (int)tiles_n, m_fast, ktiles, kps, total);
    GN_LAUNCH_CHECK(name);
    return 0;
}
template <class AL, class BL, int ROLE>
static int launch_256(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep, int64_t tiles_m,
                      int64_t tiles_n, int nsplit, int m_fast, int ktiles, int kps, bool f16, hipStream_t st) {
    return f16 ? launch_256_t<AL, BL, ROLE, true>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, m_fast, ktiles, kps, st)
               : launch_256_t<AL, BL, ROLE, false>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, m_fast, ktiles, kps, st);
}

template <class AL, class BL, int ROLE, bool F16> static const char* gemm_bf16_256_kernel_name() { return __PRETTY_FUNCTION__; }
const char* conv_bf16_256_kernel_name(int role) {
    return role == 0 ? gemm_bf16_256_kernel_name<ConvAPadLoader256<64>, KCLoader256<32>, 0, false>()
                     : gemm_bf16_256_kernel_name<ConvAPadLoader256<64>, KCLoader256<32>, 1, false>();
}

// conv 3x3 forward (bias + ReLU epilogue) / data gradient (raw epilogue) on zero-padded 16-bit activations, 256^2 phased tile
int launch_conv_bf16_256(const char* name, const __hip_bfloat16* x_pad, int H, int W, int Cin, int64_t M, const __hip_bfloat16* w,
                         int Cout, const EpiP& ep, bool f16, hipStream_t st) {
    typedef ConvAPadLoader256<64> AL;
    typedef KCLoader256<32> BL;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (Cout + T - 1) / T;
    AL::P ap{x_pad, H, W, Cin, M, 0, 0u};
    BL::P bp{w, (int64_t)9 * Cin, Cout, Cin, 0, 0u};
    const int kt = 9 * Cin / BKH;
    // ROLE 0 carries the bias (in its accumulator start) and the optional ReLU; ROLE 1 stores raw accumulators
    return ep.mode == EPI_BIAS_RELU && (ep.relu || ep.bias) ? launch_256<AL, BL, 0>(name, ap, bp, ep, tiles_m, tiles_n, 1, 0, kt, kt, f16, st)
                                                            : launch_256<AL, BL, 1>(name, ap, bp, ep, tiles_m, tiles_n, 1, 0, kt, kt, f16, st);
}

// split count of the weight gradient on 256 x 256 tiles: ~2048 blocks, a whole number of 256-CU rounds where possible
int wgrad_splits_256(int64_t Mp, int Cin, int Cout) {
    const int64_t tiles = (int64_t)((Cout + T - 1) / T) * ((9 * Cin + T - 1) / T);
    const int ktiles = (int)((Mp + BKH - 1) / BKH);
    int64_t s = (2048 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 64 > 1 ? ktiles / 64 : 1;             // >= 64 K-tiles per split: the 7-half-tile prologue stays < 3 %
    if (s > smax) s = smax;
    for (int64_t c = s; c < s + 32 && c <= smax; ++c)
        if ((tiles * c) % 256 == 0) { s = c; break; }
    if (s > 1024) s = 1024;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;
}

static int splits_for_256(int64_t tiles, int ktiles) {
    int64_t s = (1024 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 64 > 1 ? ktiles / 64 : 1;             // >= 64 K-tiles per split
    if (s > smax) s = smax;
    if (s >= 8) s = (s + 7) / 8 * 8;                                    // whole XCD groups
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;
}

int linear_fwd_splits_256(int M, int64_t K, int J) {
    return splits_for_256((int64_t)((M + T - 1) / T) * ((J + T - 1) / T), (int)(K / BKH));
}

// y_slabs[split][M][J] (fp32) = partial sums of x[M][K] . w[J][K]^T over the split's K range; the caller reduces + epilogue
int launch_linear_fwd_bf16_256(const char* name, const __hip_bfloat16* x, int64_t ldx, const __hip_bfloat16* w, int M, int64_t K,
                               int J, float* slabs, int nsplit, bool f16, hipStream_t st) {
    typedef KCLoader256<64> AL;
    typedef KCLoader256<32> BL;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (J + T - 1) / T;
    const int ktiles = (int)(K / BKH);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    AL::P ap{x, ldx, M, 0, 0, 0u};
    BL::P bp{w, K, J, 0, 0, 0u};
    EpiP ep{EPI_RAW, slabs, J, M, J, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)M * J};
    return launch_256<AL, BL, 3>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, 1, ktiles, kps, f16, st);
}

// dx[M][K] (fp32, or 16-bit into dx16) = dy[M][J] . w[J][K]: A K-contiguous (reduction index j), B row-contiguous; no split
int launch_linear_dx_bf16_256(const char* name, const __hip_bfloat16* dy, int64_t lddy, const __hip_bfloat16* w, int M, int64_t K,
                              int J, float* dx, __hip_bfloat16* dx16, int64_t lddx, bool f16, hipStream_t st) {
    typedef KCLoader256<64> AL;
    typedef MCLoader256 BL;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (K + T - 1) / T;
    AL::P ap{dy, lddy, M, 0, 0, 0u};
    BL::P bp{w, K, (int)K, J, 0, 0, 0u};
    EpiP ep{EPI_RAW, dx, lddx, M, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0, dx16};
    return launch_256<AL, BL, 4>(name, ap, bp, ep, tiles_m, tiles_n, 1, 1, J / BKH, J / BKH, f16, st);
}

// dw[J][K] (fp32) = dy[M][J]^T . x[M][K]: both operands row-contiguous (the reduction index m is the slow one), no split:
// a short reduction (M / 64 K-tiles) into a huge output — the 256^2 tile quarters the operand traffic per output byte
int launch_linear_dw_bf16_256(const char* name, const __hip_bfloat16* dy, int64_t lddy, const __hip_bfloat16* x, int64_t ldx, int M,
                              int64_t K, int J, float* dw, bool f16, hipStream_t st) {
    typedef MCLoader256 AL;
    typedef MCLoader256 BL;
    const int64_t tiles_m = (J + T - 1) / T, tiles_n = (K + T - 1) / T;
    const int ktiles = (M + BKH - 1) / BKH;
    AL::P ap{dy, lddy, J, M, 0, 0, 0u};
    BL::P bp{x, ldx, (int)K, M, 0, 0, 0u};
    EpiP ep{EPI_RAW, dw, K, J, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    return launch_256<AL, BL, 5>(name, ap, bp, ep, tiles_m, tiles_n, 1, 1, ktiles, ktiles, f16, st);
}

// conv 3x3 weight gradient on the zero-padded pixel grid: slabs[split][Cout][9*Cin] (fp32), reduced by the caller
int launch_wgrad_bf16_256(const char* name, const __hip_bfloat16* x_pad, const __hip_bfloat16* dy_pad, int Wp2, int Cin, int Cout,
                          int64_t Mp, float* slabs, int nsplit, bool f16, hipStream_t st) {
    typedef MCLoader256 AL;
    typedef ConvWgradBLoader256 BL;
    const int64_t tiles_m = (Cout + T - 1) / T, tiles_n = (9 * Cin + T - 1) / T;
    const int ktiles = (int)((Mp + BKH - 1) / BKH);
    GN_REQUIRE(nsplit >= 1 && nsplit <= 65535, GOALNET_E_SHAPE, "%s: bad split count %d", name, nsplit);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    AL::P ap{dy_pad, Cout, Cout, Mp, 0, 0, 0u};
    BL::P bp{x_pad, Wp2, Cin, Mp, Cin, 0, 0u};
    EpiP ep{EPI_RAW, slabs, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)Cout * 9 * Cin};
    return launch_256<AL, BL, 2>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, 1, ktiles, kps, f16, st);
}

// ---- split operands: the same kernels over NSEG K-segments of [hi | mid | lo] (bf16x6: 3 parts, 6 segments) or [hi | mid]
// (fp16x3: 2 parts of the power-of-two-scaled value, 3 segments, ep.oscale) operands (comment at SEGMAP_A) ----------------------
template <int NSEG> struct SegCfg;
template <> struct SegCfg<6> { static constexpr int PARTS = 3; static constexpr unsigned A = SEGMAP_A, B = SEGMAP_B; static constexpr bool F16 = false; };
template <> struct SegCfg<3> { static constexpr int PARTS = 2; static constexpr unsigned A = SEGMAP_A3, B = SEGMAP_B3; static constexpr bool F16 = true; };

// conv 3x3 forward / data gradient: x_pads = zero-padded [pixels][PARTS Cin], ws = [Cout][9][PARTS Cin]; fp32 result
template <int NSEG>
static int launch_conv_split_t(const char* name, const __hip_bfloat16* x_pads, int H, int W, int Cin, int64_t M, const __hip_bfloat16* ws,
                               int Cout, const EpiP& ep, hipStream_t st) {
    typedef ConvAPadLoader256<64, NSEG> AL;
    typedef KCLoader256<32, NSEG> BL;
    typedef SegCfg<NSEG> S;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (Cout + T - 1) / T;
    typename AL::P ap{x_pads, H, W, S::PARTS * Cin, M, Cin, S::A};
    typename BL::P bp{ws, (int64_t)9 * S::PARTS * Cin, Cout, S::PARTS * Cin, Cin, S::B};
    const int kt = 9 * NSEG * Cin / BKH;
    return ep.mode == EPI_BIAS_RELU && (ep.relu || ep.bias)
        ? launch_256_t<AL, BL, 0, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, 1, 0, kt, kt, st)
        : launch_256_t<AL, BL, 1, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, 1, 0, kt, kt, st);
}
int launch_conv_split_256(const char* name, int parts, const __hip_bfloat16* x_pads, int H, int W, int Cin, int64_t M,
                          const __hip_bfloat16* ws, int Cout, const EpiP& ep, hipStream_t st) {
    return parts == 3 ? launch_conv_split_t<6>(name, x_pads, H, W, Cin, M, ws, Cout, ep, st)
                      : launch_conv_split_t<3>(name, x_pads, H, W, Cin, M, ws, Cout, ep, st);
}

int wgrad_split_splits_256(int parts, int64_t Mp, int Cin, int Cout) {
    const int nseg = parts == 3 ? 6 : 3;
    const int64_t tiles = (int64_t)((Cout + T - 1) / T) * ((9 * Cin + T - 1) / T);
    const int ktiles = nseg * (int)((Mp + BKH - 1) / BKH);
    int64_t s = (2048 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 64 > 1 ? ktiles / 64 : 1;
    if (s > smax) s = smax;
    for (int64_t c = s; c < s + 32 && c <= smax; ++c)
        if ((tiles * c) % 256 == 0) { s = c; break; }
    if (s > 1024) s = 1024;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;
}

// conv 3x3 weight gradient: dy_pads = [padded pixels][PARTS Cout], x_pads = [padded pixels][PARTS Cin]; slabs[split][Cout][9 Cin]
template <int NSEG>
static int launch_wgrad_split_t(const char* name, const __hip_bfloat16* x_pads, const __hip_bfloat16* dy_pads, int Wp2, int Cin, int Cout,
                                int64_t Mp, float* slabs, int nsplit, hipStream_t st) {
    typedef MCLoader256T<NSEG> AL;
    typedef ConvWgradBLoader256T<NSEG> BL;
    typedef SegCfg<NSEG> S;
    const int64_t tiles_m = (Cout + T - 1) / T, tiles_n = (9 * Cin + T - 1) / T;
    const int seg_kt = (int)((Mp + BKH - 1) / BKH), ktiles = NSEG * seg_kt;
    GN_REQUIRE(nsplit >= 1 && nsplit <= 65535, GOALNET_E_SHAPE, "%s: bad split count %d", name, nsplit);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    typename AL::P ap{dy_pads, (int64_t)S::PARTS * Cout, Cout, Mp, seg_kt, Cout, S::A};
    typename BL::P bp{x_pads, Wp2, Cin, Mp, S::PARTS * Cin, seg_kt, S::B};
    EpiP ep{EPI_RAW, slabs, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)Cout * 9 * Cin};
    return launch_256_t<AL, BL, 2, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, 1, ktiles, kps, st);
}
int launch_wgrad_split_256(const char* name, int parts, const __hip_bfloat16* x_pads, const __hip_bfloat16* dy_pads, int Wp2, int Cin,
                           int Cout, int64_t Mp, float* slabs, int nsplit, hipStream_t st) {
    return parts == 3 ? launch_wgrad_split_t<6>(name, x_pads, dy_pads, Wp2, Cin, Cout, Mp, slabs, nsplit, st)
                      : launch_wgrad_split_t<3>(name, x_pads, dy_pads, Wp2, Cin, Cout, Mp, slabs, nsplit, st);
}

// linear5 on split operands: xs [M][PARTS K], ws [J][PARTS K], dys [M][PARTS J] (parts side by side along the row)
int linear_fwd_split_splits_256(int parts, int M, int64_t K, int J) {
    return splits_for_256((int64_t)((M + T - 1) / T) * ((J + T - 1) / T), (int)((parts == 3 ? 6 : 3) * K / BKH));
}

template <int NSEG>
static int launch_linear_fwd_split_t(const char* name, const __hip_bfloat16* xs, const __hip_bfloat16* ws, int M, int64_t K, int J,
                                     float* slabs, int nsplit, hipStream_t st) {
    typedef KCLoader256<64, NSEG> AL;
    typedef KCLoader256<32, NSEG> BL;
    typedef SegCfg<NSEG> S;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (J + T - 1) / T;
    const int ktiles = (int)(NSEG * K / BKH);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    typename AL::P ap{xs, S::PARTS * K, M, 0, (int)K, S::A};
    typename BL::P bp{ws, S::PARTS * K, J, 0, (int)K, S::B};
    EpiP ep{EPI_RAW, slabs, J, M, J, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)M * J};
    return launch_256_t<AL, BL, 3, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, nsplit, 1, ktiles, kps, st);
}
int launch_linear_fwd_split_256(const char* name, int parts, const __hip_bfloat16* xs, const __hip_bfloat16* ws, int M, int64_t K, int J,
                                float* slabs, int nsplit, hipStream_t st) {
    return parts == 3 ? launch_linear_fwd_split_t<6>(name, xs, ws, M, K, J, slabs, nsplit, st)
                      : launch_linear_fwd_split_t<3>(name, xs, ws, M, K, J, slabs, nsplit, st);
}

template <int NSEG>
static int launch_linear_dx_split_t(const char* name, const __hip_bfloat16* dys, const __hip_bfloat16* ws, int M, int64_t K, int J, float* dx,
                                    int64_t lddx, const int* oscale, hipStream_t st) {
    typedef KCLoader256<64, NSEG> AL;
    typedef MCLoader256T<NSEG> BL;
    typedef SegCfg<NSEG> S;
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (K + T - 1) / T;
    const int ktiles = NSEG * J / BKH;
    typename AL::P ap{dys, (int64_t)S::PARTS * J, M, 0, J, S::A};
    typename BL::P bp{ws, S::PARTS * K, (int)K, J, 1, (int)K, S::B};
    EpiP ep{EPI_RAW, dx, lddx, M, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr};
    ep.oscale = oscale;
    return launch_256_t<AL, BL, 4, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, 1, 1, ktiles, ktiles, st);
}
int launch_linear_dx_split_256(const char* name, int parts, const __hip_bfloat16* dys, const __hip_bfloat16* ws, int M, int64_t K, int J,
                               float* dx, int64_t lddx, const int* oscale, hipStream_t st) {
    return parts == 3 ? launch_linear_dx_split_t<6>(name, dys, ws, M, K, J, dx, lddx, oscale, st)
                      : launch_linear_dx_split_t<3>(name, dys, ws, M, K, J, dx, lddx, oscale, st);
}

template <int NSEG>
static int launch_linear_dw_split_t(const char* name, const __hip_bfloat16* dys, const __hip_bfloat16* xs, int M, int64_t K, int J, float* dw,
                                    const int* oscale, hipStream_t st) {
    typedef MCLoader256T<NSEG> AL;
    typedef MCLoader256T<NSEG> BL;
    typedef SegCfg<NSEG> S;
    const int64_t tiles_m = (J + T - 1) / T, tiles_n = (K + T - 1) / T;
    const int ktiles = NSEG * ((M + BKH - 1) / BKH);
    typename AL::P ap{dys, (int64_t)S::PARTS * J, J, M, 1, J, S::A};
    typename BL::P bp{xs, S::PARTS * K, (int)K, M, 1, (int)K, S::B};
    EpiP ep{EPI_RAW, dw, K, J, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    ep.oscale = oscale;
    return launch_256_t<AL, BL, 5, S::F16>(name, ap, bp, ep, tiles_m, tiles_n, 1, 1, ktiles, ktiles, st);
}
int launch_linear_dw_split_256(const char* name, int parts, const __hip_bfloat16* dys, const __hip_bfloat16* xs, int M, int64_t K, int J,
                               float* dw, const int* oscale, hipStream_t st) {
    return parts == 3 ? launch_linear_dw_split_t<6>(name, dys, xs, M, K, J, dw, oscale, st)
                      : launch_linear_dw_split_t<3>(name, dys, xs, M, K, J, dw, oscale, st);
}

}  // namespace goalnet
