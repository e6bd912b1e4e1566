// K5: running mean / variance trackers and (de)normalisation.
// Replaces RunningMeanStd (utils/stats.py:9-94) and
// RunningStatNormalizer.normalize/denormalize (utils/misc.py:84-128).
//
// The reference all-gathers the RAW data of every rank and takes np.mean/np.var
// of the concatenation (stats.py:47-54).  Here each rank reduces its own batch
// to (n, mean, M2) in float64, the records are all-gathered (3 doubles per
// tracked column instead of the data), merged with Chan's formula -- equal to
// the statistics of the concatenation -- and integrated into the float32
// running state with the reference's own update (stats.py:73-94).
#include "common.hpp"

namespace ppoaf {

// data [n, W] row-major.  One workgroup per block of columns; W is small
// (1 for values/rewards, obs_dim for observations).  Each wave owns a column
// group so loads along a row stay contiguous: thread (r, c) walks rows
// r, r+R, ... of column c.
__global__ __launch_bounds__(1024) void batch_moments_kernel(const float* __restrict__ data,
                                                             long n, int W,
                                                             double* __restrict__ out) {
    __shared__ double red[17];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0 && blockIdx.x == 0) out[0] = (double)n;
    for (int c = blockIdx.x; c < W; c += gridDim.x) {
        double s = 0.0;
        for (long i = tid; i < n; i += nt) s += (double)data[i * W + c];
        const double mean = block_sum(s, red) / (double)n;
        double q = 0.0;
        for (long i = tid; i < n; i += nt) { const double d = (double)data[i * W + c] - mean; q += d * d; }
        q = block_sum(q, red);
        if (tid == 0) { out[1 + c] = mean; out[1 + W + c] = q; }
    }
}

// moments [R, 1+2W]; state mean/var float32[W], count float64[1].
__global__ __launch_bounds__(256) void running_moments_integrate_kernel(
    const double* __restrict__ moments, int R, int W, float* __restrict__ mean,
    float* __restrict__ var, double* __restrict__ count) {
    const int stride = 1 + 2 * W;
    const double old_count = count[0];
    double n_tot = 0.0;
    for (int r = 0; r < R; ++r) n_tot += moments[(long)r * stride];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) {
        // Chan merge of the R records -> batch (n, mean, M2)
        double n = 0.0, m = 0.0, M2 = 0.0;
        for (int r = 0; r < R; ++r) {
            const double nb = moments[(long)r * stride];
            if (nb <= 0.0) continue;
            const double mb = moments[(long)r * stride + 1 + c];
            const double qb = moments[(long)r * stride + 1 + W + c];
            const double d = mb - m, nn = n + nb;
            m += d * (nb / nn);
            M2 += qb + d * d * n * nb / nn;
            n = nn;
        }
        if (n <= 0.0) continue;
        // stats.py:52-54: batch mean / population variance in the data's dtype
        const float batch_mean = (float)m;
        const float batch_var = (float)(M2 / n);
        // stats.py:73-94, same expression order; mean/var float32, counts float64
        const float old_mean = mean[c], old_var = var[c];
        const float delta = batch_mean - old_mean;
        const double new_count = old_count + n;
        const float new_mean = (float)((double)old_mean + (double)delta * (n / new_count));
        const double m_a = (double)old_var * old_count;
        const double m_b = (double)batch_var * n;
        const double dsq = (double)(delta * delta);
        const double m_2 = m_a + m_b + dsq * old_count * n / (old_count + n);
        mean[c] = new_mean;
        var[c] = (float)(m_2 / (old_count + n));
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_tot > 0.0) count[0] = old_count + n_tot;
}

__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, long total,
                                                        int W, const float* __restrict__ mean,
                                                        const float* __restrict__ var, float eps,
                                                        float lo, float hi, int has_clip,
                                                        float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int c = (W == 1) ? 0 : (int)(i % W);
        float y = (x[i] - mean[c]) / sqrtf(var[c] + eps);
        if (has_clip) y = fminf(fmaxf(y, lo), hi);
        out[i] = y;
    }
}

__global__ __launch_bounds__(256) void denormalize_kernel(const float* __restrict__ x, long total,
                                                          int W, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float eps,
                                                          float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int c = (W == 1) ? 0 : (int)(i % W);
        out[i] = mean[c] + x[i] * sqrtf(var[c] + eps);
    }
}

static unsigned ew_grid(long total) {
    long b = (total + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_batch_moments(const float* data, int64_t n, int32_t W, double* moments_out,
                                   ppoaf_stream_t stream) {
    PPOAF_REQUIRE(data && moments_out, "batch_moments: null pointer");
    PPOAF_REQUIRE(n >= 1 && W >= 1, "batch_moments: need n >= 1, W >= 1 (n=%ld W=%d)", (long)n, W);
    int threads = 64;
    while (threads < 1024 && threads < n) threads <<= 1;
    const unsigned grid = (unsigned)(W < 1024 ? W : 1024);
    hipLaunchKernelGGL(batch_moments_kernel, dim3(grid), dim3(threads), 0, (hipStream_t)stream, data,
                       (long)n, W, moments_out);
    return check_launch("batch_moments");
}

extern "C" int ppoaf_running_moments_integrate(const double* moments, int32_t R, int32_t W,
                                               float* mean, float* var, double* count,
                                               ppoaf_stream_t stream) {
    PPOAF_REQUIRE(moments && mean && var && count, "running_moments_integrate: null pointer");
    PPOAF_REQUIRE(R >= 1 && W >= 1, "running_moments_integrate: R=%d W=%d", R, W);
    // one block: the count update must follow every column's read of the old count
    PPOAF_REQUIRE(W <= 65536, "running_moments_integrate: W=%d too large", W);
    hipLaunchKernelGGL(running_moments_integrate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       moments, R, W, mean, var, count);
    return check_launch("running_moments_integrate");
}

extern "C" int ppoaf_normalize(const float* x, int64_t n, int32_t W, const float* mean,
                               const float* var, float eps, float clip_lo, float clip_hi,
                               int has_clip, float* out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 0 && W >= 1, "normalize: bad shape");
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(x && mean && var && out, "normalize: null pointer");
    const long total = n * (long)W;
    hipLaunchKernelGGL(normalize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x,
                       total, W, mean, var, eps, clip_lo, clip_hi, has_clip, out);
    return check_launch("normalize");
}

extern "C" int ppoaf_denormalize(const float* x, int64_t n, int32_t W, const float* mean,
                                 const float* var, float eps, float* out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 0 && W >= 1, "denormalize: bad shape");
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(x && mean && var && out, "denormalize: null pointer");
    const long total = n * (long)W;
    hipLaunchKernelGGL(denormalize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       x, total, W, mean, var, eps, out);
    return check_launch("denormalize");
}
