// Device-resident step state for the graph-captured sub-batch loop (the caller's loop is
// /root/reference/main.py:169-198). A captured HIP graph bakes every host scalar into its kernel arguments, so the
// three quantities that change from one sub-batch to the next live in device memory instead:
//   * the Adam step count t (bias corrections, main.py:193 -> torch.optim.Adam),
//   * the dropout stream index (which Bernoulli draw this forward uses, utils.py:170, 245-254),
//   * the frame cursor / sub-batch index of the per-video loop (main.py:177-183).
// The kernels here read them from int64 device counters; goalnet_counter_add advances them inside the graph.
#include "common.h"
#include "rng.h"

using namespace goalnet;

namespace {

__global__ void counter_add_kernel(int64_t* ctr, int64_t delta) { *ctr += delta; }
__global__ void counters_add4_kernel(int64_t* ctr, int64_t d0, int64_t d1, int64_t d2, int64_t d3) {
    const int i = threadIdx.x;
    ctr[i] += i == 0 ? d0 : i == 1 ? d1 : i == 2 ? d2 : d3;
}

// precision = "fp16": when goalnet_grad_finite_check stamped this step (*bad_step == the 1-based count this step ran under) the
// fused Adam left everything untouched; the step count then does NOT advance either (torch's GradScaler does not count skipped
// steps: the next step's bias corrections are those of the step that was skipped) and the stamp is cleared, so that the
// retry — which runs under the same count — is not taken for the overflowed one. The other three counters always advance.
__global__ void counters_add4_guarded_kernel(int64_t* ctr, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int64_t* bad_step) {
    const int i = threadIdx.x;
    if (i == 0) {
        if (*bad_step == ctr[0] + d0) *bad_step = 0;
        else ctr[0] += d0;
    } else {
        ctr[i] += i == 1 ? d1 : i == 2 ? d2 : d3;
    }
}

struct RowCopies { goalnet_rowcopy seg[GOALNET_ROWCOPY_MAX]; };

// several gathers / scatters in one launch: blockIdx.y selects the segment
__global__ __launch_bounds__(256) void rows_copy_batch_kernel(RowCopies rc) {
    const goalnet_rowcopy sg = rc.seg[blockIdx.y];
    const int64_t row_words = sg.row_bytes / 4, words = row_words * sg.nrows, base = (*sg.cursor + sg.cursor_bias) * row_words;
    const uint32_t* s = (const uint32_t*)sg.src + (sg.gather ? base : 0);
    uint32_t* d = (uint32_t*)sg.dst + (sg.gather ? 0 : base);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// the step's last launch: scatter the results (<= 4 small segments, ONE block) and then advance the four counters. One block reads
// every cursor before it moves any: predictions.extend(...), losses.append(...), subbatch_offset += n (main.py:195-198) in one launch.
__global__ __launch_bounds__(256) void rows_scatter_tick_kernel(RowCopies rc, int count, int64_t* ctr, int64_t d0, int64_t d1, int64_t d2,
                                                               int64_t d3, int64_t* bad_step) {
    for (int k = 0; k < count; ++k) {
        const goalnet_rowcopy sg = rc.seg[k];
        const int64_t row_words = sg.row_bytes / 4, words = row_words * sg.nrows, base = (*sg.cursor + sg.cursor_bias) * row_words;
        const uint32_t* s = (const uint32_t*)sg.src + (sg.gather ? base : 0);
        uint32_t* d = (uint32_t*)sg.dst + (sg.gather ? 0 : base);
        for (int64_t i = threadIdx.x; i < words; i += blockDim.x) d[i] = s[i];
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i == 0) {
        if (bad_step && *bad_step == ctr[0] + d0) *bad_step = 0;       // precision = "fp16": a skipped step is not counted (counters_add4_guarded)
        else ctr[0] += d0;
    } else if (i < 4) {
        ctr[i] += i == 1 ? d1 : i == 2 ? d2 : d3;
    }
}

struct Widths { int w[8]; int64_t off[9]; };

__global__ __launch_bounds__(256) void dropout_masks_dev_kernel(float* dst, int n, Widths ws, int layers, uint64_t seed,
                                                                 uint32_t tid_base, uint32_t tid_stride,
                                                                 const int64_t* __restrict__ step, float p, float scale,
                                                                 int64_t row_offset) {
    const uint32_t s = (uint32_t)*step;
    const int64_t total = ws.off[layers];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < layers && i >= ws.off[l + 1]) ++l;
        const uint64_t key = stream_key(seed, tid_base + tid_stride * s + (uint32_t)l);
        dst[i] = unit24(key, i - ws.off[l] + row_offset * ws.w[l]) >= p ? scale : 0.f;
    }
}

// rows [cursor, cursor + nrows) of a row-major table <-> a compact block of nrows rows; 4-byte words
template <bool GATHER>
__global__ __launch_bounds__(256) void rows_copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                       int64_t row_words, int64_t words, const int64_t* __restrict__ cursor) {
    const int64_t base = *cursor * row_words;
    const uint32_t* s = GATHER ? src + base : src;
    uint32_t* d = GATHER ? dst : dst + base;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

unsigned grid1(int64_t n, int cap) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int goalnet_counter_add(int64_t* counter, int64_t delta, void* stream) {
    GN_REQUIRE(counter, GOALNET_E_NULL, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
    GN_LAUNCH_CHECK("counter_add");
    return 0;
}

int goalnet_counters_add4(int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3, void* stream) {
    GN_REQUIRE(counters, GOALNET_E_NULL, "counters_add4: null pointer");
    hipLaunchKernelGGL(counters_add4_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, counters, d0, d1, d2, d3);
    GN_LAUNCH_CHECK("counters_add4");
    return 0;
}

int goalnet_counters_add4_guarded(int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int64_t* bad_step, void* stream) {
    GN_REQUIRE(counters && bad_step, GOALNET_E_NULL, "counters_add4_guarded: null pointer");
    hipLaunchKernelGGL(counters_add4_guarded_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, counters, d0, d1, d2, d3, bad_step);
    GN_LAUNCH_CHECK("counters_add4_guarded");
    return 0;
}

int goalnet_rows_copy_batch(const goalnet_rowcopy* segs, int count, void* stream) {
    GN_REQUIRE(segs, GOALNET_E_NULL, "rows_copy_batch: null pointer");
    GN_REQUIRE(count >= 1 && count <= GOALNET_ROWCOPY_MAX, GOALNET_E_SHAPE, "rows_copy_batch: 1..%d segments", GOALNET_ROWCOPY_MAX);
    RowCopies rc;
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        const goalnet_rowcopy& g = segs[i];
        GN_REQUIRE(g.src && g.dst && g.cursor, GOALNET_E_NULL, "rows_copy_batch: null pointer in segment %d", i);
        GN_REQUIRE(g.row_bytes > 0 && (g.row_bytes & 3) == 0 && g.nrows > 0, GOALNET_E_SHAPE,
                   "rows_copy_batch: segment %d: rows must be a positive multiple of 4 bytes", i);
        rc.seg[i] = g;
        const int64_t words = g.row_bytes / 4 * g.nrows;
        if (words > most) most = words;
    }
    hipLaunchKernelGGL(rows_copy_batch_kernel, dim3(grid1(most, 2048), (unsigned)count), dim3(256), 0, (hipStream_t)stream, rc);
    GN_LAUNCH_CHECK("rows_copy_batch");
    return 0;
}

int goalnet_rows_scatter_tick(const goalnet_rowcopy* segs, int count, int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3,
                              int64_t* bad_step, void* stream) {
    GN_REQUIRE(segs && counters, GOALNET_E_NULL, "rows_scatter_tick: null pointer");
    GN_REQUIRE(count >= 1 && count <= GOALNET_ROWCOPY_MAX, GOALNET_E_SHAPE, "rows_scatter_tick: 1..%d segments", GOALNET_ROWCOPY_MAX);
    RowCopies rc;
    for (int i = 0; i < count; ++i) {
        const goalnet_rowcopy& g = segs[i];
        GN_REQUIRE(g.src && g.dst && g.cursor, GOALNET_E_NULL, "rows_scatter_tick: null pointer in segment %d", i);
        GN_REQUIRE(g.row_bytes > 0 && (g.row_bytes & 3) == 0 && g.nrows > 0 && g.row_bytes / 4 * g.nrows <= (1 << 20), GOALNET_E_SHAPE,
                   "rows_scatter_tick: segment %d: rows must be a positive multiple of 4 bytes, <= 4 MB in all (one block copies them)", i);
        rc.seg[i] = g;
    }
    hipLaunchKernelGGL(rows_scatter_tick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rc, count, counters, d0, d1, d2, d3, bad_step);
    GN_LAUNCH_CHECK("rows_scatter_tick");
    return 0;
}

int goalnet_dropout_masks_dev(float* dst, int n, const int* widths, int layers, uint64_t seed, uint32_t tid_base,
                              uint32_t tid_stride, const int64_t* step, float p, int64_t row_offset, void* stream) {
    GN_REQUIRE(dst && widths && step, GOALNET_E_NULL, "dropout_masks_dev: null pointer");
    GN_REQUIRE(n > 0 && layers >= 1 && layers <= 8 && p >= 0.f && p < 1.f && row_offset >= 0, GOALNET_E_SHAPE, "dropout_masks_dev: bad dims or p");
    Widths ws;
    ws.off[0] = 0;
    for (int l = 0; l < layers; ++l) {
        GN_REQUIRE(widths[l] > 0, GOALNET_E_SHAPE, "dropout_masks_dev: bad width");
        ws.w[l] = widths[l];
        ws.off[l + 1] = ws.off[l] + (int64_t)n * widths[l];
    }
    hipLaunchKernelGGL(dropout_masks_dev_kernel, dim3(grid1(ws.off[layers], 8192)), dim3(256), 0, (hipStream_t)stream, dst, n, ws,
                       layers, seed, tid_base, tid_stride, step, p, 1.0f / (1.0f - p), row_offset);
    GN_LAUNCH_CHECK("dropout_masks_dev");
    return 0;
}

int goalnet_rows_gather(const void* table, void* block, int64_t row_bytes, int nrows, const int64_t* cursor, void* stream) {
    GN_REQUIRE(table && block && cursor, GOALNET_E_NULL, "rows_gather: null pointer");
    GN_REQUIRE(row_bytes > 0 && (row_bytes & 3) == 0 && nrows > 0, GOALNET_E_SHAPE, "rows_gather: rows must be a positive multiple of 4 bytes");
    const int64_t words = row_bytes / 4 * nrows;
    hipLaunchKernelGGL(rows_copy_kernel<true>, dim3(grid1(words, 2048)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)table,
                       (uint32_t*)block, row_bytes / 4, words, cursor);
    GN_LAUNCH_CHECK("rows_gather");
    return 0;
}

int goalnet_rows_scatter(const void* block, void* table, int64_t row_bytes, int nrows, const int64_t* cursor, void* stream) {
    GN_REQUIRE(table && block && cursor, GOALNET_E_NULL, "rows_scatter: null pointer");
    GN_REQUIRE(row_bytes > 0 && (row_bytes & 3) == 0 && nrows > 0, GOALNET_E_SHAPE, "rows_scatter: rows must be a positive multiple of 4 bytes");
    const int64_t words = row_bytes / 4 * nrows;
    hipLaunchKernelGGL(rows_copy_kernel<false>, dim3(grid1(words, 2048)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)block,
                       (uint32_t*)table, row_bytes / 4, words, cursor);
    GN_LAUNCH_CHECK("rows_scatter");
    return 0;
}

}  // extern "C"
