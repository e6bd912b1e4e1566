// K11: flat-bucket gradient averaging scale + global-norm clip + Adam step.
// Replaces, per network, PPOPolicy.update_weights' tail
//   mpi_avg_gradients (the /num_procs of utils/mpi_utils.py:86)
//   nn.utils.clip_grad_norm_(params, gradient_clip)   policies/ppo_policy.py:1037-1040
//   Adam(lr, eps=1e-5).step()                          policies/ppo_policy.py:336-339,1042
// on ONE contiguous float32 bucket per optimizer (P <= 146k parameters), so the
// per-tensor Python loops become two launches: a squared-norm reduction and the
// fused clip+Adam update.  The Adam step counter and the learning rate live on
// the device, which keeps the pair replayable inside a hipGraph.
#include "common.hpp"

namespace ppoaf {

// Squared-norm partials, one per workgroup, into scratch[2 + blockIdx.x]: plain stores, no atomics -- the consumer
// adds them in a fixed association (ordered_partial_sum), so the clip coefficient is the same in every run and on
// every rank of a DD-PPO job (replicas must stay bitwise identical).
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, long n,
                                                          float scale, double* __restrict__ scratch,
                                                          int64_t* __restrict__ step_count) {
    __shared__ double red[17];
    double s = 0.0;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        const float a = v.x * scale, b = v.y * scale, c = v.z * scale, d = v.w * scale;
        s += (double)a * a + (double)b * b + (double)c * c + (double)d * d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float a = g[(n4 << 2) + threadIdx.x] * scale;
        s += (double)a * a;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        scratch[2 + blockIdx.x] = s;
        if (blockIdx.x == 0) step_count[0] += 1;      // consumed by the next kernel on the stream
    }
}

struct AdamParams {
    float beta1, beta2, eps, grad_scale, max_norm;
};

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p,
                                                        const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long n, const int64_t* __restrict__ step_count,
                                                        const float* __restrict__ lr_dev, AdamParams a,
                                                        double* __restrict__ scratch, unsigned n_partials,
                                                        float* __restrict__ grad_norm_out) {
    // n_partials > 0: the squared norm arrives as per-workgroup partials scratch[2 ..] (every wave adds them in the
    // same fixed association); 0: scratch[0] already holds it
    const double sq = n_partials ? ordered_partial_sum(scratch + 2, n_partials) : scratch[0];
    if (n_partials && blockIdx.x == 0 && threadIdx.x == 0) scratch[0] = sq;
    const float total_norm = (float)sqrt(sq);
    float coef = 1.0f;
    if (a.max_norm > 0.f) coef = fminf(a.max_norm / (total_norm + 1e-6f), 1.0f);
    const float gs = a.grad_scale * coef;
    const double t = (double)step_count[0];
    const float lr = lr_dev[0];
    const double bc1 = 1.0 - pow((double)a.beta1, t);
    const double bc2 = 1.0 - pow((double)a.beta2, t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    if (grad_norm_out && blockIdx.x == 0 && threadIdx.x == 0) grad_norm_out[0] = total_norm;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;
        const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_clip_adam_step(float* params, const float* grads, float* exp_avg,
                                    float* exp_avg_sq, int64_t n, int64_t* step_count,
                                    const float* lr, float beta1, float beta2, float eps,
                                    float grad_scale, float max_norm, double* norm_scratch,
                                    float* grad_norm_out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 1, "clip_adam_step: n must be >= 1");
    PPOAF_REQUIRE(params && grads && exp_avg && exp_avg_sq && step_count && lr && norm_scratch,
                  "clip_adam_step: null pointer");
    PPOAF_REQUIRE(((uintptr_t)grads & 15) == 0, "clip_adam_step: grads must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > PPOAF_NORM_SCRATCH_DOUBLES - 2) blocks = PPOAF_NORM_SCRATCH_DOUBLES - 2;
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grads, (long)n,
                       grad_scale, norm_scratch, step_count);
    int rc = check_launch("clip_adam_step/sqnorm");
    if (rc) return rc;
    AdamParams a{beta1, beta2, eps, grad_scale, max_norm};
    long blocks2 = (n + 255) / 256;
    if (blocks2 > 2048) blocks2 = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks2), dim3(256), 0, s, params, grads,
                       exp_avg, exp_avg_sq, (long)n, step_count, lr, a, norm_scratch, (unsigned)blocks, grad_norm_out);
    return check_launch("clip_adam_step/adam");
}

extern "C" int ppoaf_adam_step_prenormed(float* params, const float* grads, float* exp_avg,
                                         float* exp_avg_sq, int64_t n, const int64_t* step_count,
                                         const float* lr, float beta1, float beta2, float eps,
                                         float grad_scale, float max_norm, double* norm_scratch,
                                         int32_t n_norm_partials, float* grad_norm_out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 1, "adam_step_prenormed: n must be >= 1");
    PPOAF_REQUIRE(n_norm_partials >= 0 && n_norm_partials <= (1 << 20), "adam_step_prenormed: n_norm_partials=%d", n_norm_partials);
    PPOAF_REQUIRE(params && grads && exp_avg && exp_avg_sq && step_count && lr && norm_scratch,
                  "adam_step_prenormed: null pointer");
    AdamParams a{beta1, beta2, eps, grad_scale, max_norm};
    long blocks2 = (n + 255) / 256;
    if (blocks2 > 2048) blocks2 = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks2), dim3(256), 0, (hipStream_t)stream, params,
                       grads, exp_avg, exp_avg_sq, (long)n, step_count, lr, a, norm_scratch, (unsigned)n_norm_partials,
                       grad_norm_out);
    return check_launch("adam_step_prenormed");
}
