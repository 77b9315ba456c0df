// The fusion MLP + head (/root/reference/utils.py:242-258, 269-270) and its backward (autograd of main.py:192) at the
// reference's operating point — sub-batches of <= 16 frames (main.py:44, 177-184) — as ONE launch per direction.
//
// With 10 rows the five layers (640 -> 512 -> 512 -> 256 -> 128 -> 1, 3 MB of weights) are a chain of latency-bound weight
// streams: as separate launches (csrc/skinny.hip: 6 forward, 12 backward) each costs its own launch gap and ramp although it
// runs 4 - 8 us. Here 64 blocks stay resident and walk the layers together; between layers they meet at a grid barrier
// (an arrival counter in device memory; what crosses the barrier is written and read with agent-scope relaxed atomics,
// common.h "inter-block hand-over": the other XCDs' L2s are not coherent with this one's and a fence would flush the whole
// L2). The barrier is bounded: a block that does not see the others
// arrive within ~1 s sets an error word and every block leaves — no wave can spin forever. 64 blocks of 256 threads are
// always co-resident on 256 CUs, also beside other kernels of the step (side-stream work, graph branches).
//
// All sums run in a fixed order (deterministic); fp32 throughout, as the reference.
#include <stdlib.h>

#include "common.h"
#include "mse.h"

using namespace goalnet;

namespace {

constexpr int MLP_BLOCKS = 64;            // backward
constexpr int MLP_FWD_BLOCKS = 64;        // forward (16 | 32 | 64 instantiated; measured 813 | 785 | 771 us per 10-frame step)
constexpr int KMAX = 640;                 // widest layer input (fusion.0 with audio: 128 + 512)

struct MlpFwdP {
    const float* x0; int64_t ldx0; int K0;          // cat (n, K0), row stride ldx0
    const float* w[5]; const float* b[5];           // fusion.0, .3, .6, .9, .12
    const float* mask[4]; int64_t ldmask[4];        // dropout multipliers of fusion.2, .5, .8, .11 (nullable)
    float* h[4];                                    // layer outputs (n, J[l]), contiguous
    float* mult[4];                                 // (pre-activation > 0) * mask, saved for backward (nullable)
    float* logit; float* out;                       // (n)
    const float* labels; float* loss; float* dout;  // optional tail: the broadcast MSE of `out` against labels (n) and dL/dout
    int n; int J[4];
    int* sync;                                      // [0] arrivals, [1] departures, [2] error flag; zero on entry / exit
};

struct MlpBwdP {
    const float* dout; const float* out;            // (n)
    const float* x[5]; int64_t ldx0;                // layer inputs: cat (n, K0; row stride ldx0), h1 .. h4 (contiguous)
    const float* m[5]; int64_t ldm0;                // saved multipliers of those inputs: mcat (row stride ldm0), m1 .. m4 (nullable)
    const float* w[5];
    float* dw[5]; float* db[5];
    float* dz[4];                                   // scratch: gradients wrt the pre-activations of layers 0..3's OUTPUTS' inputs, see kernel
    float* dcat; int64_t lddcat;                    // (n, K0): gradient wrt the pre-activations behind `cat` (linear5 | audbl.linear3)
    float* db5; int voff;                           // visbl.linear5.bias gradient = column sums of dcat[:, voff:] (nullable)
    int n; int K0; int J[4];
    int* sync;
};

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// grid_arrive / grid_wait / grid_barrier / grid_leave: common.h (bounded grid barrier over sc1 hand-overs)

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// NB blocks = 4 NB waves; wave gw takes columns gw, gw + 4 NB, ... of every layer (512 | 512 | 256 | 128 columns: with NB = 32 that is
// 4 | 4 | 2 | 1 per wave). All of a wave's weights (25 float4 per lane at NB = 32) are requested at kernel start — they depend on
// nothing — so that between two barriers only the previous layer's activations have to arrive.
template <int MR, int NB>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpFwdP P) {
    __shared__ __attribute__((aligned(16))) float xs[MR * KMAX];
    constexpr int NW = NB * 4;
    constexpr int C0 = 512 / NW > 0 ? 512 / NW : 1, C2 = 256 / NW > 0 ? 256 / NW : 1, C3 = 128 / NW > 0 ? 128 / NW : 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int gw = blockIdx.x * 4 + wv;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w0[C0][3], w1[C0][2], w2[C2][2], w3[C3];
#pragma unroll
    for (int c = 0; c < C0; ++c) {
        const int j = gw + c * NW;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = lane * 4 + 256 * i;
            w0[c][i] = (j < P.J[0] && k < P.K0) ? *reinterpret_cast<const float4*>(P.w[0] + (int64_t)j * P.K0 + k) : z4;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = lane * 4 + 256 * i;
            w1[c][i] = (j < P.J[1] && k < P.J[0]) ? *reinterpret_cast<const float4*>(P.w[1] + (int64_t)j * P.J[0] + k) : z4;
        }
    }
#pragma unroll
    for (int c = 0; c < C2; ++c) {
        const int j = gw + c * NW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = lane * 4 + 256 * i;
            w2[c][i] = (j < P.J[2] && k < P.J[1]) ? *reinterpret_cast<const float4*>(P.w[2] + (int64_t)j * P.J[1] + k) : z4;
        }
    }
#pragma unroll
    for (int c = 0; c < C3; ++c) {
        const int j = gw + c * NW;
        w3[c] = (j < P.J[3] && lane * 4 < P.J[2]) ? *reinterpret_cast<const float4*>(P.w[3] + (int64_t)j * P.J[2] + lane * 4) : z4;
    }

    // one layer: columns j = gw + c NW from the prefetched fragments WF[c][i] (NI float4 per column), epilogue, sc1 store
#define GN_MLP_LAYER(L, NC, NI, WF)                                                                                     \
    _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                                    \
        const int j = gw + c * NW;                                                                                      \
        if (j < P.J[L]) {                                                                                               \
            float acc[MR];                                                                                              \
            _Pragma("unroll") for (int m = 0; m < MR; ++m) acc[m] = 0.f;                                                \
            _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                            \
                const int k = lane * 4 + 256 * i;                                                                       \
                if (k < K) {                                                                                            \
                    _Pragma("unroll") for (int m = 0; m < MR; ++m)                                                      \
                        acc[m] += dot4(WF, *reinterpret_cast<const float4*>(&xs[m * KMAX + k]));                        \
                }                                                                                                       \
            }                                                                                                           \
            float mine = 0.f;                                                                                           \
            _Pragma("unroll") for (int m = 0; m < MR; ++m) {                                                            \
                const float v = wave_sum_dpp(acc[m]);                                                                   \
                if (lane == m) mine = v;                                                                                \
            }                                                                                                           \
            if (lane < P.n) {                                                                                           \
                float v = mine + P.b[L][j];                                                                             \
                float g = v > 0.f ? 1.f : 0.f;                                                                          \
                v = v > 0.f ? v : 0.f;                                                                                  \
                if (P.mask[L]) { const float mk = P.mask[L][(int64_t)lane * P.ldmask[L] + j]; v *= mk; g *= mk; }       \
                st_dev(&P.h[L][(int64_t)lane * P.J[L] + j], v);                                                         \
                if (P.mult[L]) P.mult[L][(int64_t)lane * P.J[L] + j] = g;                                               \
            }                                                                                                           \
        }                                                                                                               \
    }

    for (int l = 0; l < 4; ++l) {
        const int K = l == 0 ? P.K0 : P.J[l - 1];
        if (l == 0) {
            const int kq = K >> 2;
            for (int i = tid; i < MR * kq; i += 256) {
                const int m = i / kq, q = i - m * kq;
                *reinterpret_cast<float4*>(&xs[m * KMAX + q * 4]) = m < P.n ? *reinterpret_cast<const float4*>(P.x0 + (int64_t)m * P.ldx0 + q * 4) : z4;
            }
        } else {
            __syncthreads();                                   // xs of the previous layer has been read by every wave
            const float* x = P.h[l - 1];                       // written by other blocks of this launch: ld_dev
            for (int i = tid; i < MR * K; i += 256) {
                const int m = i / K, k = i - m * K;
                xs[m * KMAX + k] = m < P.n ? ld_dev(x + (int64_t)m * K + k) : 0.f;
            }
        }
        __syncthreads();
        if (l == 0) { GN_MLP_LAYER(0, C0, 3, w0[c][i]) }
        else if (l == 1) { GN_MLP_LAYER(1, C0, 2, w1[c][i]) }
        else if (l == 2) { GN_MLP_LAYER(2, C2, 2, w2[c][i]) }
        else { GN_MLP_LAYER(3, C3, 1, w3[c]) }
        grid_arrive(P.sync);
        if (l == 3 && blockIdx.x != 0) break;                  // the head runs in block 0 only
        if (!grid_wait(P.sync, l + 1, gridDim.x)) return;
    }
#undef GN_MLP_LAYER
    if (blockIdx.x == 0) {
        // head (utils.py:255-256, 270): z = h4 . w12 + b12; out = 4 sigmoid(z) + 1. One wave per row.
        const int K = P.J[3];
        for (int m = wv; m < P.n; m += 4) {
            float acc = 0.f;
            for (int k = lane; k < K; k += 64) acc = fmaf(ld_dev(&P.h[3][(int64_t)m * K + k]), P.w[4][k], acc);
            acc = wave_sum_dpp(acc);
            if (lane == 0) {
                const float z = acc + P.b[4][0];
                P.logit[m] = z;
                P.out[m] = 4.f / (1.f + expf(-z)) + 1.f;
            }
        }
        if (P.labels) {
            // nn.MSELoss on (n,1) x (n,) (main.py:191) in the same launch: this block wrote every out[m] itself
            __syncthreads();
            mse_bcast_block(P.out, P.labels, P.n, P.loss, P.dout);
        }
    }
    grid_leave(P.sync, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------
// backward. Layer l (weights w[l]: J_l x K_l) has input x[l] (n, K_l) and output gradient g_l (n, J_l) wrt its pre-activation:
//   dW_l[j][k] = sum_m g_l[m][j] x[l][m][k],   db_l[j] = sum_m g_l[m][j],   g_{l-1}[m][k] = (sum_j g_l[m][j] w[l][j][k]) m[l][m][k]
// g_4 (the head's input gradient) is n x 128 values computed from dout / out by every block for itself; g_3 .. g_0 travel
// through dz[3] .. dz[0] in global memory with a grid barrier in between; the gradient behind `cat` (linear5's and
// audbl.linear3's pre-activations) is dcat.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dlogit_of(float dout, float out) {
    const float s = (out - 1.f) * 0.25f;                     // d/dz (4 sigmoid(z) + 1) = 4 s (1 - s)
    return dout * 4.f * s * (1.f - s);
}

constexpr int KL = 4;                 // dX: k-float4s per block item
constexpr int JG = 256 / KL;          // dX: j-groups per block

template <int MR>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpBwdP P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // max(512 * MR floats, 256 * MR float4)
    __shared__ float s_dl[MR];
    const int tid = threadIdx.x;
    const int NB = gridDim.x;
    if (tid < MR) s_dl[tid] = tid < P.n ? dlogit_of(P.dout[tid], P.out[tid]) : 0.f;
    __syncthreads();
    if (blockIdx.x == NB - 1) {
        // head's own parameters: dw12[k] = sum_m dlogit[m] h4[m][k], db12 = sum_m dlogit[m] (rows in index order)
        const int K = P.J[3];
        for (int k = tid; k <= K; k += 256) {
            float t = 0.f;
            for (int m = 0; m < P.n; ++m) t += k < K ? s_dl[m] * P.x[4][(int64_t)m * K + k] : s_dl[m];
            if (k < K) P.dw[4][k] = t; else P.db[4][0] = t;
        }
    }
    for (int l = 3; l >= 0; --l) {
        const int J = P.J[l];
        const int K = l == 0 ? P.K0 : P.J[l - 1];
        const float* x = P.x[l];
        const int64_t ldx = l == 0 ? P.ldx0 : (int64_t)K;
        const float* W = P.w[l];
        // ---- stage g_l as gs[j][m] (one b128 read gives four rows)
        __syncthreads();
        if (l == 3) {
            for (int i = tid; i < J * MR; i += 256) {
                const int j = i / MR, m = i - j * MR;
                float v = 0.f;
                if (m < P.n) { v = s_dl[m] * P.w[4][j]; if (P.m[4]) v *= P.m[4][(int64_t)m * J + j]; }
                sm[i] = v;
            }
        } else {
            for (int i = tid; i < J * MR; i += 256) {
                const int j = i / MR, m = i - j * MR;
                sm[i] = m < P.n ? ld_dev(&P.dz[l][(int64_t)m * J + j]) : 0.f;       // written by other blocks of this launch
            }
        }
        __syncthreads();
        // ---- dW_l, db_l: items = (k-float4, group of 8 output rows); the thread keeps its MR x-values in registers
        const int kq = K >> 2, jch = (J + 7) >> 3;
        for (int it = blockIdx.x * 256 + tid; it < kq * jch; it += NB * 256) {
            const int q = it % kq, jc = it / kq;
            float4 x4[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) x4[m] = m < P.n ? *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int j1 = jc * 8 + 8 < J ? jc * 8 + 8 : J;
            for (int j = jc * 8; j < j1; ++j) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mq = 0; mq < MR / 4; ++mq) {
                    const float4 d = *reinterpret_cast<const float4*>(&sm[j * MR + mq * 4]);
                    s.x += d.x * x4[mq * 4].x + d.y * x4[mq * 4 + 1].x + d.z * x4[mq * 4 + 2].x + d.w * x4[mq * 4 + 3].x;
                    s.y += d.x * x4[mq * 4].y + d.y * x4[mq * 4 + 1].y + d.z * x4[mq * 4 + 2].y + d.w * x4[mq * 4 + 3].y;
                    s.z += d.x * x4[mq * 4].z + d.y * x4[mq * 4 + 1].z + d.z * x4[mq * 4 + 2].z + d.w * x4[mq * 4 + 3].z;
                    s.w += d.x * x4[mq * 4].w + d.y * x4[mq * 4 + 1].w + d.z * x4[mq * 4 + 2].w + d.w * x4[mq * 4 + 3].w;
                }
                *reinterpret_cast<float4*>(P.dw[l] + (int64_t)j * K + q * 4) = s;
            }
        }
        for (int j = blockIdx.x * 256 + tid; j < J; j += NB * 256) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < MR; ++m) t += sm[j * MR + m];
            P.db[l][j] = t;
        }
        // ---- g_{l-1} (or dcat): block items of KL k-float4s; 64 j-groups per item, reduced through LDS in a fixed order
        float* dst = l == 0 ? P.dcat : P.dz[l - 1];
        const int64_t lddst = l == 0 ? P.lddcat : (int64_t)K;
        const float* mul = P.m[l];
        const int64_t ldmul = l == 0 ? P.ldm0 : (int64_t)K;
        const int nitems = (kq + KL - 1) / KL;
        const int kl = tid % KL, jg = tid / KL;
        float4* red = reinterpret_cast<float4*>(sm);                  // red[jg][m][kl], aliases gs
        for (int it = blockIdx.x; it < nitems; it += NB) {
            const int k = (it * KL + kl) * 4;
            const bool kok = k < K;
            float4 acc[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int j = jg; j < J; j += JG) {
                const float4 w4 = kok ? *reinterpret_cast<const float4*>(W + (int64_t)j * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mq = 0; mq < MR / 4; ++mq) {
                    const float4 d = *reinterpret_cast<const float4*>(&sm[j * MR + mq * 4]);
                    acc[mq * 4 + 0].x += d.x * w4.x; acc[mq * 4 + 0].y += d.x * w4.y; acc[mq * 4 + 0].z += d.x * w4.z; acc[mq * 4 + 0].w += d.x * w4.w;
                    acc[mq * 4 + 1].x += d.y * w4.x; acc[mq * 4 + 1].y += d.y * w4.y; acc[mq * 4 + 1].z += d.y * w4.z; acc[mq * 4 + 1].w += d.y * w4.w;
                    acc[mq * 4 + 2].x += d.z * w4.x; acc[mq * 4 + 2].y += d.z * w4.y; acc[mq * 4 + 2].z += d.z * w4.z; acc[mq * 4 + 2].w += d.z * w4.w;
                    acc[mq * 4 + 3].x += d.w * w4.x; acc[mq * 4 + 3].y += d.w * w4.y; acc[mq * 4 + 3].z += d.w * w4.z; acc[mq * 4 + 3].w += d.w * w4.w;
                }
            }
            __syncthreads();                                          // every thread has read gs: the buffer becomes red[]
#pragma unroll
            for (int m = 0; m < MR; ++m) red[(jg * MR + m) * KL + kl] = acc[m];
            __syncthreads();
            const int m = tid / KL;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < MR) {
#pragma unroll 16
                for (int g = 0; g < JG; ++g) {
                    const float4 v = red[(g * MR + m) * KL + kl];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                if (m < P.n && kok) {
                    if (mul) {
                        const float4 mv = *reinterpret_cast<const float4*>(mul + (int64_t)m * ldmul + k);
                        s.x *= mv.x; s.y *= mv.y; s.z *= mv.z; s.w *= mv.w;
                    }
                    float* o = dst + (int64_t)m * lddst + k;
                    if (l > 0) { st_dev(o, s.x); st_dev(o + 1, s.y); st_dev(o + 2, s.z); st_dev(o + 3, s.w); }      // the next phase's g, read by every block
                    else *reinterpret_cast<float4*>(o) = s;
                } else {
                    s = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (l == 0 && P.db5) {
                // visbl.linear5.bias gradient: column sums of dcat[:, voff:], rows added in index order
                __syncthreads();
                if (m < MR) red[m * KL + kl] = s;
                __syncthreads();
                if (tid < KL && k >= P.voff && kok) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int mm = 0; mm < MR; ++mm) { const float4 v = red[mm * KL + kl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                    *reinterpret_cast<float4*>(P.db5 + (k - P.voff)) = t;
                }
            }
            // the next item re-stages nothing (gs is gone): one item per block and layer is the design point (nitems <= NB)
            break;
        }
        if (l > 0) {
            if (!grid_barrier(P.sync, 4 - l, NB)) return;
        }
    }
    grid_leave(P.sync, NB);
}

int rows_class(int M) { return M <= 4 ? 4 : M <= 8 ? 8 : M <= 12 ? 12 : 16; }

}  // namespace

extern "C" {

int goalnet_mlp_blocks(void) { return MLP_BLOCKS; }

int goalnet_mlp_fwd(const float* cat, int64_t ldcat, int K0, const float* const* w, const float* const* b,
                    const float* const* mask, const int64_t* ldmask, float* const* h, float* const* mult,
                    float* logit, float* out, const float* labels, float* loss, float* dout, int n, int* sync, void* stream) {
    GN_REQUIRE(cat && w && b && mask && ldmask && h && mult && logit && out && sync, GOALNET_E_NULL, "mlp_fwd: null pointer");
    GN_REQUIRE(n >= 1 && n <= 16, GOALNET_E_SHAPE, "mlp_fwd: 1..16 rows (the reference's sub-batches)");
    GN_REQUIRE(K0 > 0 && K0 <= KMAX && K0 % 4 == 0 && ldcat % 4 == 0 && aligned16(cat), GOALNET_E_SHAPE, "mlp_fwd: K0 must be a multiple of 4, <= 640");
    MlpFwdP P;
    P.x0 = cat; P.ldx0 = ldcat; P.K0 = K0;
    const int J[4] = {512, 512, 256, 128};
    for (int l = 0; l < 5; ++l) {
        GN_REQUIRE(w[l] && b[l] && aligned16(w[l]), GOALNET_E_NULL, "mlp_fwd: null / misaligned weight %d", l);
        P.w[l] = w[l]; P.b[l] = b[l];
    }
    for (int l = 0; l < 4; ++l) {
        GN_REQUIRE(h[l] && aligned16(h[l]), GOALNET_E_NULL, "mlp_fwd: null / misaligned output %d", l);
        P.mask[l] = mask[l]; P.ldmask[l] = ldmask[l]; P.h[l] = h[l]; P.mult[l] = mult[l]; P.J[l] = J[l];
    }
    GN_REQUIRE(!labels || (loss || dout), GOALNET_E_NULL, "mlp_fwd: labels without a loss / dout destination");
    P.logit = logit; P.out = out; P.labels = labels; P.loss = loss; P.dout = dout; P.n = n; P.sync = sync;
    hipStream_t st = (hipStream_t)stream;
    static const int fb = getenv("GOALNET_MLP_FWD_BLOCKS") ? atoi(getenv("GOALNET_MLP_FWD_BLOCKS")) : MLP_FWD_BLOCKS;      // 16 | 32 | 64 (A/B runs)
#define GN_MLP_FWD(MRV)                                                                                       \
    if (fb == 16) hipLaunchKernelGGL((mlp_fwd_kernel<MRV, 16>), dim3(16), dim3(256), 0, st, P);              \
    else if (fb == 32) hipLaunchKernelGGL((mlp_fwd_kernel<MRV, 32>), dim3(32), dim3(256), 0, st, P);        \
    else hipLaunchKernelGGL((mlp_fwd_kernel<MRV, 64>), dim3(64), dim3(256), 0, st, P);
    switch (rows_class(n)) {
        case 4: GN_MLP_FWD(4) break;
        case 8: GN_MLP_FWD(8) break;
        case 12: GN_MLP_FWD(12) break;
        default: GN_MLP_FWD(16) break;
    }
#undef GN_MLP_FWD
    GN_LAUNCH_CHECK("mlp_fwd");
    return 0;
}

size_t goalnet_mlp_bwd_ws_bytes(int n) { return (size_t)n * (512 + 512 + 256 + 128) * sizeof(float); }

int goalnet_mlp_bwd(const float* dout, const float* out, const float* const* x, int64_t ldcat, const float* const* m, int64_t ldmcat,
                    const float* const* w, float* const* dw, float* const* db, float* dcat, int64_t lddcat, float* db5, int voff,
                    int n, int K0, void* ws, size_t ws_bytes, int* sync, void* stream) {
    GN_REQUIRE(dout && out && x && m && w && dw && db && dcat && ws && sync, GOALNET_E_NULL, "mlp_bwd: null pointer");
    GN_REQUIRE(n >= 1 && n <= 16, GOALNET_E_SHAPE, "mlp_bwd: 1..16 rows (the reference's sub-batches)");
    GN_REQUIRE(K0 > 0 && K0 <= KMAX && K0 % 4 == 0 && ldcat % 4 == 0 && lddcat % 4 == 0 && ldmcat % 4 == 0 && voff % 4 == 0 && voff >= 0 && voff < K0,
               GOALNET_E_SHAPE, "mlp_bwd: K0 / leading dims / voff must be multiples of 4");
    GN_REQUIRE(ws_bytes >= goalnet_mlp_bwd_ws_bytes(n) && aligned16(ws) && aligned16(dcat) && aligned16(db5), GOALNET_E_WORKSPACE, "mlp_bwd: workspace too small or misaligned");
    MlpBwdP P;
    P.dout = dout; P.out = out; P.ldx0 = ldcat; P.ldm0 = ldmcat;
    const int J[4] = {512, 512, 256, 128};
    for (int l = 0; l < 5; ++l) {
        GN_REQUIRE(x[l] && w[l] && dw[l] && db[l] && aligned16(x[l]) && aligned16(w[l]) && aligned16(dw[l]) && aligned16(m[l]), GOALNET_E_NULL,
                   "mlp_bwd: null / misaligned pointer of layer %d", l);
        P.x[l] = x[l]; P.m[l] = m[l]; P.w[l] = w[l]; P.dw[l] = dw[l]; P.db[l] = db[l];
    }
    float* z = (float*)ws;
    for (int l = 0; l < 4; ++l) { P.J[l] = J[l]; P.dz[l] = z; z += (size_t)n * J[l]; }
    P.dcat = dcat; P.lddcat = lddcat; P.db5 = db5; P.voff = voff; P.n = n; P.K0 = K0; P.sync = sync;
    GN_REQUIRE((K0 / 4 + KL - 1) / KL <= MLP_BLOCKS, GOALNET_E_SHAPE, "mlp_bwd: K0 too wide for one item per block");
    hipStream_t st = (hipStream_t)stream;
#define GN_MLP_BWD(MRV)                                                                                            \
    {                                                                                                              \
        const size_t a = (size_t)512 * MRV * sizeof(float), bb = (size_t)256 * MRV * sizeof(float4);               \
        hipLaunchKernelGGL(mlp_bwd_kernel<MRV>, dim3(MLP_BLOCKS), dim3(256), a > bb ? a : bb, st, P);              \
    }
    switch (rows_class(n)) {
        case 4: GN_MLP_BWD(4) break;
        case 8: GN_MLP_BWD(8) break;
        case 12: GN_MLP_BWD(12) break;
        default: GN_MLP_BWD(16) break;
    }
#undef GN_MLP_BWD
    GN_LAUNCH_CHECK("mlp_bwd");
    return 0;
}

}  // extern "C"
