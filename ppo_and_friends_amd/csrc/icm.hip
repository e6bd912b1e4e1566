// K8: ICM forward-model loss and intrinsic reward.
// Replaces the tail of ICM.forward (networks/ppo_networks/icm.py:421-430):
//   f_loss_elem      = (obs_2_pred - enc_obs_2)^2                      nn.MSELoss(reduction="none")
//   intrinsic_reward = (reward_scale / 2) * f_loss_elem.sum(dim=-1)    [n]
//   f_loss           = 0.5 * f_loss_elem.mean()                        scalar
// used per env step (policies/ppo_policy.py:999-1005) and per mini-batch (ppo.py:2549-2553).
// One wave per row (D = 128 by default: two 16-B loads per lane pair), row sums reduced by shuffles;
// the scalar loss is reduced by a single workgroup in a fixed order (reproducible).
#include "common.hpp"

namespace ppoaf {

__global__ __launch_bounds__(256) void icm_rowsum_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ enc2, long n, int D,
                                                         float scale, float* __restrict__ rowsum,
                                                         float* __restrict__ intr) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;                                    // wave-uniform
    const float* p = pred + row * D;
    const float* q = enc2 + row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = p[i] - q[i]; s += d * d; }
    s = wave_sum(s);
    if (lane == 0) { rowsum[row] = s; if (intr) intr[row] = scale * 0.5f * s; }
}

__global__ __launch_bounds__(1024) void icm_floss_kernel(const float* __restrict__ rowsum, long n, int D,
                                                         float* __restrict__ f_loss) {
    __shared__ double red[17];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += (double)rowsum[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) f_loss[0] = (float)(0.5 * s / ((double)n * (double)D));
}

// d f_loss / d pred = (pred - enc2) / (n * D) ; d / d enc2 = -that ; scaled by the upstream gradient
__global__ __launch_bounds__(256) void icm_floss_bwd_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ enc2, long total,
                                                            float inv_nd, const float* __restrict__ g,
                                                            float* __restrict__ d_pred,
                                                            float* __restrict__ d_enc2) {
    const float gs = g[0] * inv_nd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float d = (pred[i] - enc2[i]) * gs;
        d_pred[i] = d;
        if (d_enc2) d_enc2[i] = -d;
    }
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_icm_forward_loss_fwd(const float* pred, const float* enc2, int64_t n, int32_t D,
                                          float reward_scale, float* rowsum_scratch, float* intr_out,
                                          float* f_loss_out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 1 && D >= 1, "icm_forward_loss_fwd: n=%ld D=%d", (long)n, D);
    PPOAF_REQUIRE(pred && enc2 && rowsum_scratch, "icm_forward_loss_fwd: null pointer");
    PPOAF_REQUIRE((n + 3) / 4 <= 0x7fffffffL, "icm_forward_loss_fwd: n too large");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(icm_rowsum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, pred, enc2, (long)n, D,
                       reward_scale, rowsum_scratch, intr_out);
    int rc = check_launch("icm_forward_loss_fwd/rowsum");
    if (rc || !f_loss_out) return rc;
    hipLaunchKernelGGL(icm_floss_kernel, dim3(1), dim3(1024), 0, s, rowsum_scratch, (long)n, D, f_loss_out);
    return check_launch("icm_forward_loss_fwd/floss");
}

extern "C" int ppoaf_icm_forward_loss_bwd(const float* pred, const float* enc2, int64_t n, int32_t D,
                                          const float* grad_f_loss, float* d_pred, float* d_enc2,
                                          ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n >= 1 && D >= 1, "icm_forward_loss_bwd: n=%ld D=%d", (long)n, D);
    PPOAF_REQUIRE(pred && enc2 && grad_f_loss && d_pred, "icm_forward_loss_bwd: null pointer");
    const long total = n * (long)D;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(icm_floss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, enc2,
                       total, 1.0f / ((float)n * (float)D), grad_f_loss, d_pred, d_enc2);
    return check_launch("icm_forward_loss_bwd");
}
