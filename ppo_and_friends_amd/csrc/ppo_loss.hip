// K2+K3: mini-batch advantage normalisation + clipped-surrogate / value /
// entropy loss, forward and backward in one launch.
// Replaces ppo.py:2325-2333 and ppo.py:2352-2438 of the reference.
//
// B is a mini-batch (256 by default, B*A for MAT): the whole problem is one
// workgroup's worth of data, so a single 1024-thread workgroup does
//   pass 0: mean(adv)                       (block reduction, wave shuffles)
//   pass 1: sum (adv-mean)^2 -> unbiased std (two-pass, as accurate as torch's)
//   pass 2: ratios / surrogates / kl / entropy / critic loss partial sums,
//           and the three gradient streams
// Algorithmic traffic: 24 B read + 12 B written per sample; the re-reads of
// adv in passes 1-2 hit L1/L2.
#include "common.hpp"

namespace ppoaf {

struct LossParams {
    int normalize_adv, use_huber;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta;
};

__global__ __launch_bounds__(1024) void ppo_loss_kernel(
    const float* __restrict__ cur_logp, const float* __restrict__ old_logp,
    const float* __restrict__ adv, const float* __restrict__ entropy,
    const float* __restrict__ values, const float* __restrict__ rtg, long B, LossParams p,
    float* __restrict__ scalars, float* __restrict__ d_logp, float* __restrict__ d_entropy,
    float* __restrict__ d_values) {
    __shared__ double red[17];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;

    double mean = 0.0, stdv = 1.0;
    if (p.normalize_adv) {
        double s = 0.0;
        for (long i = tid; i < B; i += nt) s += (double)adv[i];
        mean = block_sum(s, red) / (double)B;
        double q = 0.0;
        for (long i = tid; i < B; i += nt) { const double d = (double)adv[i] - mean; q += d * d; }
        q = block_sum(q, red);
        // torch.std: Bessel-corrected; B == 1 gives NaN there too (0/0) and the
        // reference aborts on it (ppo.py:2328-2331) -- the host checks scalars[6].
        stdv = sqrt(q / (double)(B - 1));
    }
    const float mean_f = (float)mean, std_f = (float)stdv;
    const float inv_B = 1.0f / (float)B;
    const float lo = 1.0f - p.surr_clip, hi = 1.0f + p.surr_clip;

    double s_surr = 0.0, s_ent = 0.0, s_kl = 0.0, s_crit = 0.0;
    float bad = 0.f;
    for (long i = tid; i < B; i += nt) {
        float a = adv[i];
        if (p.normalize_adv) a = (a - mean_f) / (std_f + 1e-8f);
        const float lp = cur_logp[i], lpo = old_logp[i];
        const float ratio = expf(lp - lpo);
        if (isnan(ratio) || isinf(ratio)) bad = 1.f;
        const float surr1 = ratio * a;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float surr2 = rc * a;
        s_surr += (double)(-fminf(surr1, surr2));
        s_kl += (double)(lpo - lp);
        const float h = entropy ? entropy[i] : 0.f;
        s_ent += (double)h;
        // d(-min(s1,s2))/d lp.  torch.min(a,b) splits the gradient evenly on
        // ties; surr1 == surr2 exactly whenever the ratio is inside the clip
        // range (the clamp is then the identity with gradient 1), so both
        // halves add up to -a*ratio.  surr2's gradient is 0 where the clamp
        // is active.
        float g;
        if (surr1 <= surr2) g = -a * ratio;
        else g = (ratio >= lo && ratio <= hi) ? -a * ratio : 0.f;
        if (d_logp) d_logp[i] = g * inv_B;
        if (d_entropy) d_entropy[i] = (p.entropy_weight != 0.0f) ? -p.entropy_weight * inv_B : 0.f;
        const float diff = values[i] - rtg[i];
        float l, dl;
        if (p.use_huber) {
            const float ad = fabsf(diff);
            if (ad < p.huber_delta) { l = 0.5f * diff * diff; dl = diff; }
            else { l = p.huber_delta * (ad - 0.5f * p.huber_delta); dl = diff > 0.f ? p.huber_delta : -p.huber_delta; }
        } else {
            l = diff * diff; dl = 2.0f * diff;
        }
        s_crit += (double)l;
        if (d_values) d_values[i] = dl * inv_B;
    }
    s_surr = block_sum(s_surr, red);
    s_ent = block_sum(s_ent, red);
    s_kl = block_sum(s_kl, red);
    s_crit = block_sum(s_crit, red);
    const double badsum = block_sum((double)bad, red);
    if (tid == 0) {
        const double n = (double)B;
        const float surr = (float)(s_surr / n);
        const float ent = (float)(s_ent / n);
        const float kl = (float)(s_kl / n);
        float total = surr;
        if (p.entropy_weight != 0.0f) total -= p.entropy_weight * ent;
        if (p.kl_loss_weight > 0.0f) total += p.kl_loss_weight * kl;
        scalars[0] = surr;
        scalars[1] = total;
        scalars[2] = (float)(s_crit / n);
        scalars[3] = ent;
        scalars[4] = kl;
        scalars[5] = mean_f;
        scalars[6] = std_f;
        scalars[7] = badsum > 0.0 ? 1.f : 0.f;
    }
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_loss_fwd_bwd(const float* cur_logp, const float* old_logp,
                                      const float* adv, const float* entropy, const float* values,
                                      const float* rtg, int64_t B, int normalize_adv,
                                      float surr_clip, float entropy_weight, float kl_loss_weight,
                                      int use_huber, float huber_delta, float* scalars,
                                      float* d_logp, float* d_entropy, float* d_values,
                                      ppoaf_stream_t stream) {
    PPOAF_REQUIRE(B >= 1, "ppo_loss: B must be >= 1 (got %ld)", (long)B);
    PPOAF_REQUIRE(cur_logp && old_logp && adv && values && rtg && scalars, "ppo_loss: null pointer");
    PPOAF_REQUIRE(entropy || entropy_weight == 0.0f, "ppo_loss: entropy stream required");
    LossParams p;
    p.normalize_adv = normalize_adv; p.use_huber = use_huber; p.surr_clip = surr_clip;
    p.entropy_weight = entropy_weight; p.kl_loss_weight = kl_loss_weight; p.huber_delta = huber_delta;
    int threads = 64;
    while (threads < 1024 && threads < B) threads <<= 1;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, cur_logp,
                       old_logp, adv, entropy, values, rtg, (long)B, p, scalars, d_logp, d_entropy,
                       d_values);
    return check_launch("ppo_loss_fwd_bwd");
}
