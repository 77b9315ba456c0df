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
#include "gemm_bf16_common.h"

using namespace goalnet;

namespace {

constexpr int T = 256;                  // block tile edge
constexpr int HALF_BYTES = 128 * ROWB;  // 16 KB
constexpr int LDS_BYTES = 2 * 4 * HALF_BYTES;

// local row lr (0..127) of half h -> row of the 256-row tile; GROUP = rows a wave owns per half (A: 64, B: 32)
template <int GROUP>
__device__ __forceinline__ int tile_row_of(int h, int lr) { return (lr / GROUP) * (2 * GROUP) + h * GROUP + (lr % GROUP); }

// K-contiguous bf16 matrix X[rows][K] (weights [Cout][9*Cin]); wave w issues pieces 2w, 2w+1 of a half-tile
template <int GROUP>
struct KCLoader256 {
    struct P { const __hip_bfloat16* x; int64_t ld; int rows; };
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[2][2];
    int wave;
    __device__ KCLoader256(const P& p, int row0, int tid) {
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
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 8 * ROWB, voff[h][i], (unsigned)kt * ROWB);
    }
};

// im2col of a zero-padded bf16 NHWC tensor (buffer convention of gemm_bf16.hip: `x` = padded pixel 0, W+3 zero guard pixels
// in front and behind): row m = (n,h,w) of the output grid, k = (kh,kw,ci); no validity masks.
template <int GROUP>
struct ConvAPadLoader256 {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; };
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[2][2];
    int Wp2, C, lgC, wave;
    __device__ ConvAPadLoader256(const P& p, int row0, int tid) {
        Wp2 = p.W + 2; C = p.C;
        lgC = (p.C & (p.C - 1)) == 0 ? 31 - __builtin_clz(p.C) : -1;       // channel counts here are powers of two: no division per K-tile
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        const int hw = p.H * p.W;
        const int n0 = row0 / hw, r0 = row0 - n0 * hw, h0 = r0 / p.W, w0 = r0 - h0 * p.W;
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
                    const int n = (int)(m / hw), rr = (int)(m - (int64_t)n * hw), hh = rr / p.W, w = rr - hh * p.W;
                    const int64_t pm = ((int64_t)n * (p.H + 2) + hh + 1) * Wp2 + w + 1;
                    voff[h][i] = (unsigned)(((pm - pm0) * p.C + lc * 8) * 2);
                } else {
                    voff[h][i] = OOB;
                }
            }
    }
    __device__ __forceinline__ void issue(int kt, int h, char* l) const {
        const int k = kt * BKH;
        const int tap = lgC >= 0 ? k >> lgC : k / C;
        const int ci = k - tap * C;
        const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;                  // tap / 3 for tap < 16
        // past the last K-tile (tap >= 9) the offset only has to stay harmless: the range check turns it into zeros
        const unsigned s0 = tap < 9 ? (unsigned)(((kh * Wp2 + kw) * C + ci) * 2) : OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, l + (wave * 2 + i) * 8 * ROWB, voff[h][i], s0);
    }
};

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
__device__ __forceinline__ void mfma_pinned(f32x16& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define GN_PHASE(ACC0, ACC1, AF, BF)                                                                        \
    asm volatile("s_barrier\n\ts_setprio 1" ::: "memory");                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                      \
        mfma_pinned(ACC0, AF[0][ks], BF[ks]);                                                               \
        mfma_pinned(ACC1, AF[1][ks], BF[ks]);                                                               \
    }                                                                                                       \
    asm volatile("s_setprio 0\n\ts_barrier" ::: "memory");

// ROLE only separates the symbols (0: convolution forward, 1: data gradient) so that profiles list them apart
template <class AL, class BL, int ROLE>
__global__ __launch_bounds__(512, 1) void gemm_bf16_256_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                              int tiles_m, int tiles_n, int m_fast, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];          // [2 K-tiles][A0, A1, B0, B1][128 rows][128 B]
    const int tid = threadIdx.x;
    int tm, tn;
    tile_of_block(tiles_m, tiles_n, m_fast, tm, tn);
    const AL al(ap, tm * T, tid);
    const BL bl(bp, tn * T, tid);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto slot = [&](int t, int s) -> char* { return lds + ((t & 1) * 4 + s) * HALF_BYTES; };     // s: 0 A0, 1 A1, 2 B0, 3 B1

    // prologue: K-tile 0 complete, K-tile 1 without its A1 (phase (0,q0) stages that)
    al.issue(0, 0, slot(0, 0)); bl.issue(0, 0, slot(0, 2)); bl.issue(0, 1, slot(0, 3)); al.issue(0, 1, slot(0, 1));
    bl.issue(1, 0, slot(1, 2)); al.issue(1, 0, slot(1, 0)); bl.issue(1, 1, slot(1, 3));
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");                 // stagger the two wave rows by one barrier

    // fragment registers: a = the current A half; bx / by = B0 / B1 of even K-tiles and B1 / B0 of odd ones: the B0
    // fragments of the NEXT K-tile are read in phase q3 into the registers B1 has just left, so every phase issues at most
    // 8 ds_read_b128 per wave (8 | 4 | 8 | 4) and the LDS time of a phase stays within one MFMA section.
    bf16x8 a[2][4], bx[4], by[4];
    const int arow = wr * 64, brow = wc * 32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bx[ks] = read_frag(slot(0, 2), brow, ks, lane);

#define GN_KTILE(TT, B0R, B1R)                                                                              \
    {                                                                                                       \
        const int t_ = (TT);                                                                                \
        /* q0: (A0, B0) */                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                    \
            _Pragma("unroll") for (int f = 0; f < 2; ++f) a[f][ks] = read_frag(slot(t_, 0), arow + f * 32, ks, lane); \
        al.issue(t_ + 1, 1, slot(t_ + 1, 1));                                                               \
        GN_PHASE(acc[0][0], acc[1][0], a, B0R);                                                             \
        /* q1: (A0, B1) */                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) B1R[ks] = read_frag(slot(t_, 3), brow, ks, lane);  \
        bl.issue(t_ + 2, 0, slot(t_, 2));                                                                   \
        GN_PHASE(acc[0][1], acc[1][1], a, B1R);                                                             \
        /* q2: (A1, B1); B0 of the next K-tile must have landed one phase before q3 reads it */             \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                    \
            _Pragma("unroll") for (int f = 0; f < 2; ++f) a[f][ks] = read_frag(slot(t_, 1), arow + f * 32, ks, lane); \
        al.issue(t_ + 2, 0, slot(t_, 0));                                                                   \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                                    \
        GN_PHASE(acc[2][1], acc[3][1], a, B1R);                                                             \
        /* q3: (A1, B0); the counted wait that retires K-tile t+1 */                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) B1R[ks] = read_frag(slot(t_ + 1, 2), brow, ks, lane); \
        bl.issue(t_ + 2, 1, slot(t_, 3));                                                                   \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                    \
        GN_PHASE(acc[2][0], acc[3][0], a, B0R);                                                             \
    }

#pragma unroll 1
    for (int t = 0; t < ktiles; t += 2) {
        GN_KTILE(t, bx, by);
        if (t + 1 < ktiles) GN_KTILE(t + 1, by, bx);
    }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");
    // drain: no DMA may land after the block has released its LDS, and the last MFMAs (16 passes) must have written back
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));                    // no DMA may land after the block has released its LDS

    // epilogue (conv forward: bias + ReLU; data gradient: raw)
    const int r = lane & 31, hh = lane >> 5;
    const bool brelu = ep.mode == EPI_BIAS_RELU;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = tn * T + wc * 64 + ni * 32 + r;
            const bool colok = col < ep.cols;
            const float bv = (brelu && ep.bias && colok) ? ep.bias[col] : 0.f;
            const float lo = (brelu && ep.relu) ? 0.f : -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = (int64_t)tm * T + wr * 128 + (mi >> 1) * 64 + (mi & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (colok && row < ep.rows) ep.out[row * ep.ld + col] = fmaxf(acc[mi][ni][e] + bv, lo);
            }
        }
}

}  // namespace

namespace goalnet {

// conv 3x3 forward / data gradient on zero-padded bf16 activations with the 256^2 phased tile; ep.mode RAW or BIAS_RELU
template <int ROLE>
static int launch_conv_bf16_256_role(const char* name, const __hip_bfloat16* x_pad, int H, int W, int Cin, int64_t M,
                                     const __hip_bfloat16* w, int Cout, const EpiP& ep, hipStream_t st) {
    typedef ConvAPadLoader256<64> AL;
    typedef KCLoader256<32> BL;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<AL, BL, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { set_error("%s: cannot reserve %d bytes of LDS: %s", name, LDS_BYTES, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int64_t tiles_m = (M + T - 1) / T, tiles_n = (Cout + T - 1) / T;
    GN_REQUIRE(tiles_m * tiles_n < (1ll << 31), GOALNET_E_SHAPE, "%s: too many tiles", name);
    AL::P ap{x_pad, H, W, Cin, M};
    BL::P bp{w, (int64_t)9 * Cin, Cout};
    hipLaunchKernelGGL((gemm_bf16_256_kernel<AL, BL, ROLE>), dim3((unsigned)(tiles_m * tiles_n)), dim3(512), LDS_BYTES, st, ap, bp, ep,
                       (int)tiles_m, (int)tiles_n, 0, 9 * Cin / BKH);
    GN_LAUNCH_CHECK(name);
    return 0;
}

// conv 3x3 forward (bias + ReLU epilogue) / data gradient (raw epilogue) on zero-padded bf16 activations, 256^2 phased tile
int launch_conv_bf16_256(const char* name, const __hip_bfloat16* x_pad, int H, int W, int Cin, int64_t M, const __hip_bfloat16* w,
                         int Cout, const EpiP& ep, hipStream_t st) {
    return ep.mode == EPI_BIAS_RELU && ep.relu ? launch_conv_bf16_256_role<0>(name, x_pad, H, W, Cin, M, w, Cout, ep, st)
                                               : launch_conv_bf16_256_role<1>(name, x_pad, H, W, Cin, M, w, Cout, ep, st);
}

}  // namespace goalnet
