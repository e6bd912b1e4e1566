// K6: action distributions for rollout sampling and for the update's
// log-prob / entropy evaluation, forward and backward.
// Replaces networks/distributions.py:199-269 (Categorical over softmax probs,
// output_func :1043-1045) and :441-694 (Gaussian + tanh squashing), as driven by
// policies/ppo_policy.py:758-794 (rollout) and :930-952 (evaluate).
//
// One thread per row; K (number of discrete actions) and D (action dims) are
// tiny in every configuration (2..6), so a row lives in registers and rows are
// read with consecutive lanes on consecutive rows.
#include "common.hpp"
#include <cfloat>

namespace ppoaf {

constexpr int kMaxK = 64;     // per-thread register row; larger K is rejected on the host

// softmax -> torch.distributions.Categorical(probs=p):
//   n_i = p_i / sum(p);  l_i = log(clamp(n_i, eps, 1-eps)), eps = FLT_EPSILON
struct CatRow {
    float n[kMaxK];      // renormalised probs
    float s;             // sum of softmax outputs (~1)
};

template <int K_MAX>
__device__ __forceinline__ void softmax_row(const float* __restrict__ z, int K, float* p, float& s_out) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, z[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) { p[k] = expf(z[k] - m); s += p[k]; }
    const float inv = 1.0f / s;
    float s2 = 0.f;
    for (int k = 0; k < K; ++k) { p[k] *= inv; s2 += p[k]; }     // F.softmax
    s_out = s2;                                                  // Categorical's own renormaliser
}

__device__ __forceinline__ float clamp_prob(float n) {
    return fminf(fmaxf(n, FLT_EPSILON), 1.0f - FLT_EPSILON);
}

__global__ __launch_bounds__(256) void categorical_sample_kernel(
    const float* __restrict__ logits, long n, int K, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ action_out, float* __restrict__ logp_out, float* __restrict__ probs_out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float p[kMaxK];
    float s;
    softmax_row<kMaxK>(logits + row * K, K, p, s);
    const Philox4 rnd = philox4x32_10(seed, offset + (uint64_t)row, 0u);
    const float u = u32_to_unit(rnd.x) * s;         // inverse-CDF over the unnormalised mass
    int a = K - 1;
    float c = 0.f;
    for (int k = 0; k < K; ++k) {
        c += p[k];
        if (u < c) { a = k; break; }
    }
    action_out[row] = a;
    logp_out[row] = logf(clamp_prob(p[a] / s));
    if (probs_out)
        for (int k = 0; k < K; ++k) probs_out[row * K + k] = p[k];
}

__global__ __launch_bounds__(256) void categorical_eval_fwd_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ actions, long n, int K,
    float* __restrict__ logp_out, float* __restrict__ entropy_out, float* __restrict__ probs_out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float p[kMaxK];
    float s;
    softmax_row<kMaxK>(logits + row * K, K, p, s);
    long a = actions[row];
    a = a < 0 ? 0 : (a >= K ? K - 1 : a);           // host validates; never index out of the row
    float H = 0.f;
    for (int k = 0; k < K; ++k) {
        const float nk = p[k] / s;
        H -= nk * logf(clamp_prob(nk));             // Categorical.entropy: -(probs * logits).sum
    }
    logp_out[row] = logf(clamp_prob(p[a] / s));
    entropy_out[row] = H;
    if (probs_out)
        for (int k = 0; k < K; ++k) probs_out[row * K + k] = p[k];
}

// Chain: z -softmax-> p -(/sum)-> n -clamp,log-> l ;  logp = l_a ; H = -sum n_i l_i
__global__ __launch_bounds__(256) void categorical_eval_bwd_kernel(
    const float* __restrict__ probs, const int64_t* __restrict__ actions,
    const float* __restrict__ d_logp, const float* __restrict__ d_entropy, long n, int K,
    float* __restrict__ d_logits) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float p[kMaxK], gn[kMaxK];
    float s = 0.f;
    for (int k = 0; k < K; ++k) { p[k] = probs[row * K + k]; s += p[k]; }
    long a = actions[row];
    a = a < 0 ? 0 : (a >= K ? K - 1 : a);
    const float glp = d_logp ? d_logp[row] : 0.f;
    const float gH = d_entropy ? d_entropy[row] : 0.f;
    // gradient wrt n_i
    float dot_gn_n = 0.f;
    for (int k = 0; k < K; ++k) {
        const float nk = p[k] / s;
        const float ck = clamp_prob(nk);
        const float in_range = (nk >= FLT_EPSILON && nk <= 1.0f - FLT_EPSILON) ? 1.f : 0.f;
        float g = gH * (-logf(ck) - nk * in_range / ck);
        if (k == a) g += glp * in_range / ck;
        gn[k] = g;
        dot_gn_n += g * nk;
    }
    // n = p / s  ->  gp_j = (gn_j - sum_i gn_i n_i) / s
    // p = softmax(z) -> gz_k = p_k (gp_k - sum_j gp_j p_j)
    float dot_gp_p = 0.f;
    for (int k = 0; k < K; ++k) { gn[k] = (gn[k] - dot_gn_n) / s; dot_gp_p += gn[k] * p[k]; }
    for (int k = 0; k < K; ++k) d_logits[row * K + k] = p[k] * (gn[k] - dot_gp_p);
}

// ---- Gaussian + tanh -------------------------------------------------------
constexpr int kMaxD = 64;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

__device__ __forceinline__ float softplus_f(float x) {
    // torch softplus: beta=1, threshold=20
    return x > 20.f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float gauss_tanh_logp_row(const float* mean, const float* log_std,
                                                     const float* x, int D, float min_std) {
    float lp = 0.f, slog = 0.f;
    for (int d = 0; d < D; ++d) {
        const float sd = fmaxf(softplus_f(log_std[d]), min_std);
        const float zz = (x[d] - mean[d]);
        // Normal.log_prob: -((x-mu)^2)/(2 var) - log(sd) - log(sqrt(2 pi))
        float l = -(zz * zz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;
        l = fminf(fmaxf(l, -100.f), 100.f);
        lp += l;
        const float th = tanhf(x[d]);
        slog += logf(fmaxf(1.0f - th * th, 1e-6f));
    }
    return lp - slog;
}

// GaussianDistribution.get_entropy as PPOPolicy.evaluate calls it (policies/ppo_policy.py:950,
// networks/distributions.py:672-694): entropy = -get_log_probs(dist, action_pred), the tanh-corrected log-density
// of the distribution's own MEAN (not of the logged raw action): sum_d [ -clamp(-log sd - log sqrt(2 pi), +-100)
// + log(max(1 - tanh(mean)^2, 1e-6)) ].  Pinned by fixtures g8_distributions / g12_c3_*.
__device__ __forceinline__ float gauss_tanh_entropy_row(const float* mean, const float* log_std, int D, float min_std) {
    float e = 0.f;
    for (int d = 0; d < D; ++d) {
        const float sd = fmaxf(softplus_f(log_std[d]), min_std);
        const float l0 = fminf(fmaxf(-logf(sd) - kHalfLog2Pi, -100.f), 100.f);
        const float th = tanhf(mean[d]);
        e += logf(fmaxf(1.0f - th * th, 1e-6f)) - l0;
    }
    return e;
}

__global__ __launch_bounds__(256) void gaussian_eval_fwd_kernel(
    const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ x,
    long n, int D, float min_std, float* __restrict__ logp_out, float* __restrict__ entropy_out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const float lp = gauss_tanh_logp_row(mean + row * D, log_std, x + row * D, D, min_std);
    logp_out[row] = lp;
    if (entropy_out) entropy_out[row] = gauss_tanh_entropy_row(mean + row * D, log_std, D, min_std);
}

// d_mean[n,D]; d_log_std[D] = sums over ALL rows, by ONE workgroup walking the rows (mini-batch sized n): every thread
// adds its rows in ascending order, the workgroup folds in a fixed association, plain stores -- no zero-fill before the
// launch and no atomics.  (Rounds 1-2 zeroed d_log_std with hipMemsetAsync and added per-workgroup atomics.  A memset
// node captured into a hipGraph does not reliably write its value when the graph is replayed on this stack -- from the
// second replay on the destination held junk, tools/probes/memset_capture_probe.py -- so under graph replay the gradient
// of log_std started from junk: the "graph-replay drift" of the torch update path.  No captured path uses a memset now.)
constexpr int kGaussBwdThreads = 1024;
__global__ __launch_bounds__(kGaussBwdThreads) void gaussian_eval_bwd_kernel(
    const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ x,
    const float* __restrict__ d_logp, const float* __restrict__ d_entropy, long n, int D,
    float min_std, float* __restrict__ d_mean, float* __restrict__ d_log_std) {
    __shared__ float red[17];
    for (int d = 0; d < D; ++d) {
        float gls = 0.f;
        for (long row = threadIdx.x; row < n; row += kGaussBwdThreads) {
            const float g = d_logp ? d_logp[row] : 0.f, gH = d_entropy ? d_entropy[row] : 0.f;
            const float ls = log_std[d];
            const float sp = softplus_f(ls);
            const float sd = fmaxf(sp, min_std);
            const float zz = x[row * D + d] - mean[row * D + d];
            const float l = -(zz * zz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;
            const float pass = (l >= -100.f && l <= 100.f) ? 1.f : 0.f;   // clamp gradient
            // entropy (at the mean): d/dmean log(max(1 - th^2, 1e-6)) = -2 th inside the clamp; d/dsd (+log sd) = 1/sd
            const float thm = tanhf(mean[row * D + d]);
            const float pass_t = (1.0f - thm * thm >= 1e-6f) ? 1.f : 0.f;
            const float l0 = -logf(sd) - kHalfLog2Pi;
            const float pass0 = (l0 >= -100.f && l0 <= 100.f) ? 1.f : 0.f;
            // dl/dmean = zz / sd^2 ; dl/dsd = zz^2/sd^3 - 1/sd
            d_mean[row * D + d] = g * pass * zz / (sd * sd) - gH * pass_t * 2.0f * thm;
            // torch.max(std, min_std): gradient to std where std > min_std, split on ties
            const float dmax = sp > min_std ? 1.f : (sp == min_std ? 0.5f : 0.f);
            const float dsp = ls > 20.f ? 1.f : 1.0f / (1.0f + expf(-ls));  // softplus' = sigmoid
            gls += (g * pass * (zz * zz / (sd * sd * sd) - 1.0f / sd) + gH * pass0 / sd) * dmax * dsp;
        }
        const float tot = block_sum(gls, red);
        if (threadIdx.x == 0) d_log_std[d] = tot;
    }
}

__global__ __launch_bounds__(256) void gaussian_sample_kernel(
    const float* __restrict__ mean, const float* __restrict__ log_std, long n, int D, float min_std,
    const float* __restrict__ act_lo, const float* __restrict__ act_hi, uint64_t seed, uint64_t offset,
    float* __restrict__ raw_out,
    float* __restrict__ action_out, float* __restrict__ logp_out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float xr[kMaxD];
    for (int d0 = 0; d0 < D; d0 += 4) {
        // Box-Muller on one Philox block: 4 uniforms -> 4 normals
        const Philox4 r = philox4x32_10(seed, offset + (uint64_t)row, (uint32_t)(d0 >> 2));
        const float u0 = u32_to_unit_open0(r.x), u1 = u32_to_unit(r.y);
        const float u2 = u32_to_unit_open0(r.z), u3 = u32_to_unit(r.w);
        const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
        float sa, ca, sb, cb;
        sincosf(6.28318530717958647692f * u1, &sa, &ca);
        sincosf(6.28318530717958647692f * u3, &sb, &cb);
        const float z[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
        for (int j = 0; j < 4 && d0 + j < D; ++j) {
            const int d = d0 + j;
            const float sd = fmaxf(softplus_f(log_std[d]), min_std);
            xr[d] = mean[row * D + d] + sd * z[j];
        }
    }
    const bool rescale = act_lo != nullptr;        // per-dimension bounds (distributions.py:476-483); NULL: [-1, 1]
    for (int d = 0; d < D; ++d) {
        raw_out[row * D + d] = xr[d];
        float a = tanhf(xr[d]);
        if (rescale) a = ((a + 1.0f) / 2.0f) * (act_hi[d] - act_lo[d]) + act_lo[d];   // distributions.py:580-609
        action_out[row * D + d] = a;
    }
    logp_out[row] = gauss_tanh_logp_row(mean + row * D, log_std, xr, D, min_std);
}

static unsigned row_grid(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace ppoaf

using namespace ppoaf;

#define ROWS_OK(n, what)                                                          \
    PPOAF_REQUIRE((n) >= 0 && ((n) + 255) / 256 <= 0x7fffffffL, what ": bad row count %ld", (long)(n))

extern "C" int ppoaf_categorical_sample(const float* logits, int64_t n, int32_t K, uint64_t seed,
                                        uint64_t offset, int64_t* action_out, float* logp_out,
                                        float* probs_out, ppoaf_stream_t stream) {
    ROWS_OK(n, "categorical_sample");
    PPOAF_REQUIRE(K >= 1 && K <= kMaxK, "categorical_sample: K=%d out of [1,%d]", K, kMaxK);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(logits && action_out && logp_out, "categorical_sample: null pointer");
    hipLaunchKernelGGL(categorical_sample_kernel, dim3(row_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, logits, (long)n, K, seed, offset, action_out, logp_out,
                       probs_out);
    return check_launch("categorical_sample");
}

extern "C" int ppoaf_categorical_eval_fwd(const float* logits, const int64_t* actions, int64_t n,
                                          int32_t K, float* logp_out, float* entropy_out,
                                          float* probs_out, ppoaf_stream_t stream) {
    ROWS_OK(n, "categorical_eval_fwd");
    PPOAF_REQUIRE(K >= 1 && K <= kMaxK, "categorical_eval_fwd: K=%d out of [1,%d]", K, kMaxK);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(logits && actions && logp_out && entropy_out, "categorical_eval_fwd: null pointer");
    hipLaunchKernelGGL(categorical_eval_fwd_kernel, dim3(row_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, logits, actions, (long)n, K, logp_out, entropy_out,
                       probs_out);
    return check_launch("categorical_eval_fwd");
}

extern "C" int ppoaf_categorical_eval_bwd(const float* probs, const int64_t* actions,
                                          const float* d_logp, const float* d_entropy, int64_t n,
                                          int32_t K, float* d_logits, ppoaf_stream_t stream) {
    ROWS_OK(n, "categorical_eval_bwd");
    PPOAF_REQUIRE(K >= 1 && K <= kMaxK, "categorical_eval_bwd: K=%d out of [1,%d]", K, kMaxK);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(probs && actions && d_logits, "categorical_eval_bwd: null pointer");
    hipLaunchKernelGGL(categorical_eval_bwd_kernel, dim3(row_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, probs, actions, d_logp, d_entropy, (long)n, K, d_logits);
    return check_launch("categorical_eval_bwd");
}

extern "C" int ppoaf_gaussian_tanh_eval_fwd(const float* mean, const float* log_std, const float* x,
                                            int64_t n, int32_t D, float min_std, float* logp_out,
                                            float* entropy_out, ppoaf_stream_t stream) {
    ROWS_OK(n, "gaussian_tanh_eval_fwd");
    PPOAF_REQUIRE(D >= 1 && D <= kMaxD, "gaussian_tanh_eval_fwd: D=%d out of [1,%d]", D, kMaxD);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(mean && log_std && x && logp_out, "gaussian_tanh_eval_fwd: null pointer");
    hipLaunchKernelGGL(gaussian_eval_fwd_kernel, dim3(row_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, mean, log_std, x, (long)n, D, min_std, logp_out,
                       entropy_out);
    return check_launch("gaussian_tanh_eval_fwd");
}

extern "C" int ppoaf_gaussian_tanh_eval_bwd(const float* mean, const float* log_std, const float* x,
                                            const float* d_logp, const float* d_entropy, int64_t n,
                                            int32_t D, float min_std, float* d_mean,
                                            float* d_log_std, ppoaf_stream_t stream) {
    ROWS_OK(n, "gaussian_tanh_eval_bwd");
    PPOAF_REQUIRE(D >= 1 && D <= kMaxD, "gaussian_tanh_eval_bwd: D=%d out of [1,%d]", D, kMaxD);
    PPOAF_REQUIRE(d_log_std, "gaussian_tanh_eval_bwd: null d_log_std");
    PPOAF_REQUIRE(n == 0 || (mean && log_std && x && d_mean), "gaussian_tanh_eval_bwd: null pointer");
    // one workgroup (n = 0: it stores the zeros): no memset node, no atomics -- see the kernel
    hipLaunchKernelGGL(gaussian_eval_bwd_kernel, dim3(1), dim3(kGaussBwdThreads), 0,
                       (hipStream_t)stream, mean, log_std, x, d_logp, d_entropy, (long)n, D, min_std,
                       d_mean, d_log_std);
    return check_launch("gaussian_tanh_eval_bwd");
}

extern "C" int ppoaf_gaussian_tanh_sample(const float* mean, const float* log_std, int64_t n,
                                          int32_t D, float min_std, const float* act_lo,
                                          const float* act_hi,
                                          uint64_t seed, uint64_t offset, float* raw_out,
                                          float* action_out, float* logp_out,
                                          ppoaf_stream_t stream) {
    ROWS_OK(n, "gaussian_tanh_sample");
    PPOAF_REQUIRE(D >= 1 && D <= kMaxD, "gaussian_tanh_sample: D=%d out of [1,%d]", D, kMaxD);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(mean && log_std && raw_out && action_out && logp_out,
                  "gaussian_tanh_sample: null pointer");
    PPOAF_REQUIRE((act_lo == nullptr) == (act_hi == nullptr), "gaussian_tanh_sample: give both bounds or neither");
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(row_grid(n)), dim3(256), 0, (hipStream_t)stream,
                       mean, log_std, (long)n, D, min_std, act_lo, act_hi, seed, offset, raw_out,
                       action_out, logp_out);
    return check_launch("gaussian_tanh_sample");
}
