// Audio pre-processing on the device: what /root/reference/utils.py:313-349 does to a decoded waveform —
// per frame slot an MFCC matrix (librosa.feature.mfcc, utils.py:333) resampled along time to B columns with a cubic
// spline (scipy interp1d(kind='cubic'), utils.py:337-343).
//
//   logmel_slots_kernel       one block per (slot, STFT frame): 2048-point FFT of the Hann-windowed, zero-padded frame in LDS
//                             (fp64, radix 2), power spectrum, 128 Slaney mel bands (sparse triangles), 10 log10(max(1e-10, .))
//   mfcc_from_logmel_kernel   one block per slot: top_db clip at (slot maximum - 80), orthonormal DCT-II to 30 coefficients,
//                             cubic not-a-knot resample as a (B x T) matrix product
//   cubic_resample_kernel     the resample alone (rows x T -> rows x B): the part that is pinned against scipy fixtures
//
// Everything is a fixed sequence of double operations on float32 samples (HBM-bound, microseconds per video); the window,
// twiddles, mel weights, DCT and spline matrices are constants computed once on the host (cvml_goalnet_amd/preprocess.py).
// PARITY UNPINNED for the MFCC part: librosa is not installed in the build image; oracle/audio_ref.py restates its
// documented defaults (librosa itself computes in float32).
#include "common.h"

using namespace goalnet;

namespace {

constexpr int NFFT = 2048, HOP = 512, NBINS = NFFT / 2 + 1, NMELS = 128, LOGN = 11;

__device__ __forceinline__ int bitrev11(int v) { return (int)(__brev((unsigned)v) >> (32 - LOGN)); }

__global__ __launch_bounds__(256) void logmel_slots_kernel(const float* __restrict__ y, const int64_t* __restrict__ start,
                                                          const int* __restrict__ len, int Tmax,
                                                          const double* __restrict__ window, const double* __restrict__ twiddle,
                                                          const int* __restrict__ mel_start, const int* __restrict__ mel_count,
                                                          const int* __restrict__ mel_off, const double* __restrict__ mel_w,
                                                          double* __restrict__ logmel) {
    __shared__ double re[NFFT], im[NFFT];
    const int slot = blockIdx.y, t = blockIdx.x;
    const int n = len[slot];
    if (t >= 1 + n / HOP) return;                           // frames this slot does not have (whole block)
    const float* seg = y + start[slot];
    const int base = t * HOP - NFFT / 2;                    // center = True: frame t is centred at sample t * hop, zeros outside
    for (int j = threadIdx.x; j < NFFT; j += 256) {
        const int i = base + j;
        const double v = (i >= 0 && i < n) ? (double)seg[i] * window[j] : 0.0;
        const int r = bitrev11(j);
        re[r] = v; im[r] = 0.0;
    }
    __syncthreads();
    for (int s = 1; s <= LOGN; ++s) {
        const int half = 1 << (s - 1), stride = NFFT >> s;
        for (int b = threadIdx.x; b < NFFT / 2; b += 256) {
            const int pos = b & (half - 1), i = ((b >> (s - 1)) << s) + pos, j = i + half;
            const double wr = twiddle[2 * (pos * stride)], wi = twiddle[2 * (pos * stride) + 1];      // exp(-2 pi i k / NFFT)
            const double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
            const double ur = re[i], ui = im[i];
            re[i] = ur + xr; im[i] = ui + xi;
            re[j] = ur - xr; im[j] = ui - xi;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < NBINS; k += 256) re[k] = re[k] * re[k] + im[k] * im[k];       // power spectrum (k <= 1024 only)
    __syncthreads();
    if (threadIdx.x < NMELS) {
        const int m = threadIdx.x, k0 = mel_start[m], cnt = mel_count[m];
        const double* w = mel_w + mel_off[m];
        double acc = 0.0;
        for (int k = 0; k < cnt; ++k) acc += w[k] * re[k0 + k];
        logmel[((int64_t)slot * Tmax + t) * NMELS + m] = 10.0 * log10(acc > 1e-10 ? acc : 1e-10);
    }
}

__global__ __launch_bounds__(256) void mfcc_from_logmel_kernel(const double* __restrict__ logmel, const int* __restrict__ len, int Tmax,
                                                              const double* __restrict__ dct, const double* __restrict__ r_all,
                                                              const int64_t* __restrict__ r_off, float* __restrict__ out,
                                                              int n_mfcc, int B, double top_db) {
    extern __shared__ double sm[];
    const int slot = blockIdx.x;
    const int T = 1 + len[slot] / HOP;
    double* sdb = sm;                         // [T][128]
    double* mf = sm + (size_t)Tmax * NMELS;   // [n_mfcc][T]
    __shared__ double red[256];
    const double* src = logmel + (int64_t)slot * Tmax * NMELS;
    double mx = -1e300;
    for (int i = threadIdx.x; i < T * NMELS; i += 256) { const double v = src[i]; sdb[i] = v; mx = v > mx ? v : mx; }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + o] ? red[threadIdx.x] : red[threadIdx.x + o];
        __syncthreads();
    }
    const double floor_db = red[0] - top_db;                // librosa.power_to_db: np.maximum(S_db, S_db.max() - top_db)
    for (int i = threadIdx.x; i < n_mfcc * T; i += 256) {
        const int f = i / T, t = i - f * T;
        const double* d = dct + f * NMELS;
        const double* col = sdb + t * NMELS;
        double acc = 0.0;
        for (int m = 0; m < NMELS; ++m) { const double v = col[m]; acc += d[m] * (v > floor_db ? v : floor_db); }
        mf[f * T + t] = acc;
    }
    __syncthreads();
    const double* R = r_all + r_off[slot];                  // (B, T) for this slot's T
    for (int i = threadIdx.x; i < n_mfcc * B; i += 256) {
        const int f = i / B, b = i - f * B;
        const double* rr = R + (int64_t)b * T;
        const double* row = mf + f * T;
        double acc = 0.0;
        for (int t = 0; t < T; ++t) acc += rr[t] * row[t];
        out[((int64_t)slot * n_mfcc + f) * B + b] = (float)acc;
    }
}

__global__ __launch_bounds__(256) void cubic_resample_kernel(const float* __restrict__ x, const double* __restrict__ R,
                                                            float* __restrict__ yo, int64_t rows, int T, int B) {
    const int64_t total = rows * B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / B;
        const int b = (int)(i - r * B);
        const float* row = x + r * T;
        const double* rr = R + (int64_t)b * T;
        double acc = 0.0;
        for (int t = 0; t < T; ++t) acc += rr[t] * (double)row[t];
        yo[i] = (float)acc;
    }
}

}  // namespace

extern "C" {

int goalnet_cubic_resample(const float* x, const double* R, float* y, int64_t rows, int T, int B, void* stream) {
    GN_REQUIRE(x && R && y, GOALNET_E_NULL, "cubic_resample: null pointer");
    GN_REQUIRE(rows > 0 && T >= 4 && B > 0, GOALNET_E_SHAPE, "cubic_resample: rows > 0, T >= 4 (a cubic spline needs 4 points), B > 0");
    int64_t blocks = (rows * B + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cubic_resample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, R, y, rows, T, B);
    GN_LAUNCH_CHECK("cubic_resample");
    return 0;
}

int goalnet_logmel_slots(const float* y, const int64_t* start, const int* len, int n_slots, int Tmax, const double* window,
                         const double* twiddle, const int* mel_start, const int* mel_count, const int* mel_off, const double* mel_w,
                         double* logmel, void* stream) {
    GN_REQUIRE(y && start && len && window && twiddle && mel_start && mel_count && mel_off && mel_w && logmel, GOALNET_E_NULL,
               "logmel_slots: null pointer");
    GN_REQUIRE(n_slots > 0 && n_slots <= 65535 && Tmax > 0, GOALNET_E_SHAPE, "logmel_slots: 1..65535 slots, Tmax > 0");
    hipLaunchKernelGGL(logmel_slots_kernel, dim3((unsigned)Tmax, (unsigned)n_slots), dim3(256), 0, (hipStream_t)stream, y, start, len,
                       Tmax, window, twiddle, mel_start, mel_count, mel_off, mel_w, logmel);
    GN_LAUNCH_CHECK("logmel_slots");
    return 0;
}

int goalnet_mfcc_from_logmel(const double* logmel, const int* len, int n_slots, int Tmax, const double* dct, const double* r_all,
                             const int64_t* r_off, float* out, int n_mfcc, int B, double top_db, void* stream) {
    GN_REQUIRE(logmel && len && dct && r_all && r_off && out, GOALNET_E_NULL, "mfcc_from_logmel: null pointer");
    GN_REQUIRE(n_slots > 0 && Tmax >= 4 && n_mfcc > 0 && n_mfcc <= NMELS && B > 0, GOALNET_E_SHAPE, "mfcc_from_logmel: bad dims");
    const size_t lds = ((size_t)Tmax * NMELS + (size_t)n_mfcc * Tmax) * sizeof(double);
    GN_REQUIRE(lds <= 150 * 1024, GOALNET_E_SHAPE, "mfcc_from_logmel: %d STFT frames per slot do not fit in LDS", Tmax);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)mfcc_from_logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { set_error("mfcc_from_logmel: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(mfcc_from_logmel_kernel, dim3((unsigned)n_slots), dim3(256), lds, (hipStream_t)stream, logmel, len, Tmax, dct,
                       r_all, r_off, out, n_mfcc, B, top_db);
    GN_LAUNCH_CHECK("mfcc_from_logmel");
    return 0;
}

}  // extern "C"
