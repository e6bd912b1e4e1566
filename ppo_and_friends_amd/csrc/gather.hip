// K4: mini-batch gather / scatter through the shuffle permutation.
// Replaces DataLoader(shuffle=True) + PPODataset.__getitem__ + collate
// (ppo.py:2181-2184,2292-2295; utils/episode_info.py:922-952) and the value
// write-back dataset.values[batch_idxs] = values (ppo.py:2340).
//
// All fields of the 13-tuple that the update consumes are gathered in ONE
// launch: a thread copies one 4-byte word; consecutive threads walk the words
// of one row (coalesced within a field), rows are picked by perm[b] and,
// optionally, redirected from the reference's flattened episode-major order to
// the time-major [T*E] buffer through row_map.
#include "common.hpp"

namespace ppoaf {

struct GatherArgs {
    const uint32_t* src[PPOAF_MAX_GATHER_FIELDS];
    uint32_t* dst[PPOAF_MAX_GATHER_FIELDS];
    int words[PPOAF_MAX_GATHER_FIELDS];       // row width in 4-byte words
    int word_end[PPOAF_MAX_GATHER_FIELDS];    // inclusive prefix of words[]
    int n_fields;
    int words_total;
};

__global__ __launch_bounds__(256) void minibatch_gather_kernel(
    GatherArgs a, const int64_t* __restrict__ perm, const int32_t* __restrict__ row_map,
    long n_rows, long B) {
    const long total = B * (long)a.words_total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long b = i / a.words_total;
        int w = (int)(i - b * a.words_total);
        long row = perm[b];
        if (row < 0 || row >= n_rows) continue;          // host validates; never fault
        if (row_map) row = row_map[row];
        int f = 0, base = 0;
#pragma unroll
        for (int k = 0; k < PPOAF_MAX_GATHER_FIELDS; ++k) {
            if (k < a.n_fields && w >= a.word_end[k]) { f = k + 1; base = a.word_end[k]; }
        }
        w -= base;
        a.dst[f][b * a.words[f] + w] = a.src[f][row * a.words[f] + w];
    }
}

__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(
    const float* __restrict__ src, const int64_t* __restrict__ perm,
    const int32_t* __restrict__ row_map, long n_rows, long B, float* __restrict__ dst) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long row = perm[b];
    if (row < 0 || row >= n_rows) return;
    if (row_map) row = row_map[row];
    dst[row] = src[b];
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_minibatch_gather(const ppoaf_gather_field_t* fields, int32_t n_fields,
                                      const int64_t* perm, const int32_t* row_map, int64_t n_rows,
                                      int64_t B, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(fields && perm, "minibatch_gather: null pointer");
    PPOAF_REQUIRE(n_fields >= 1 && n_fields <= PPOAF_MAX_GATHER_FIELDS,
                  "minibatch_gather: n_fields=%d out of [1,%d]", n_fields, PPOAF_MAX_GATHER_FIELDS);
    PPOAF_REQUIRE(B >= 0 && n_rows >= 0, "minibatch_gather: negative shape");
    if (B == 0) return PPOAF_OK;
    GatherArgs a;
    int acc = 0;
    for (int k = 0; k < PPOAF_MAX_GATHER_FIELDS; ++k) {
        if (k < n_fields) {
            PPOAF_REQUIRE(fields[k].src && fields[k].dst, "minibatch_gather: field %d null", k);
            PPOAF_REQUIRE(fields[k].row_bytes > 0 && fields[k].row_bytes % 4 == 0,
                          "minibatch_gather: field %d row_bytes=%d must be a positive multiple of 4",
                          k, fields[k].row_bytes);
            a.src[k] = (const uint32_t*)fields[k].src;
            a.dst[k] = (uint32_t*)fields[k].dst;
            a.words[k] = fields[k].row_bytes / 4;
            acc += a.words[k];
            a.word_end[k] = acc;
        } else {
            a.src[k] = nullptr; a.dst[k] = nullptr; a.words[k] = 0; a.word_end[k] = 0x7fffffff;
        }
    }
    a.n_fields = n_fields;
    a.words_total = acc;
    const long total = B * (long)acc;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(minibatch_gather_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, a, perm, row_map, (long)n_rows, (long)B);
    return check_launch("minibatch_gather");
}

extern "C" int ppoaf_scatter_rows_f32(const float* src, const int64_t* perm,
                                      const int32_t* row_map, int64_t n_rows, int64_t B,
                                      float* dst, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(src && perm && dst, "scatter_rows_f32: null pointer");
    PPOAF_REQUIRE(B >= 0 && n_rows >= 0, "scatter_rows_f32: negative shape");
    if (B == 0) return PPOAF_OK;
    const long blocks = (B + 255) / 256;
    PPOAF_REQUIRE(blocks <= 0x7fffffffL, "scatter_rows_f32: B too large");
    hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, src, perm, row_map, (long)n_rows, (long)B, dst);
    return check_launch("scatter_rows_f32");
}
