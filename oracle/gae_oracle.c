/*
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- plain-C restatement of the
 * reference's trajectory scans, used to cross-check the numpy oracle and as a
 * fast CPU baseline for the GAE kernel alone.  Not part of the product.
 *
 * Follows /root/reference/utils/episode_info.py:
 *   discounted_sums   <- EpisodeInfo.compute_discounted_sums   :223-262
 *   gae_rtg_episode   <- EpisodeInfo.end_episode               :419-465
 *                        + _compute_gae_advantages             :264-293
 * Rounding points as in oracle/episode_info_oracle.py (pinned by tests/golden/g1):
 *   delta = r + (double)(float)(gamma * V[t+1]) - V[t], scans in double,
 *   ending reward clipped then rounded to float, ending value rounded to float.
 */
#include <stddef.h>
#include <stdint.h>

static float clipf(double x, int has_clip, double lo, double hi) {
    if (has_clip) { if (x < lo) x = lo; if (x > hi) x = hi; }
    return (float)x;
}

/* one episode: rewards[L] (double), values[L] (float) -> adv[L], rtg[L] (double) */
void ppoaf_oracle_gae_rtg_episode(const double* rewards, const float* values, int64_t L,
                                  double ending_value, double ending_reward,
                                  double gamma, double lambd, int has_clip, double clip_lo,
                                  double clip_hi, int use_gae, double* adv, double* rtg) {
    const float er = clipf(ending_reward, has_clip, clip_lo, clip_hi);
    const float ev = (float)ending_value;
    const float gamma_f = (float)gamma;
    const double gl = gamma * lambd;
    double d = (double)er;                 /* padded_rewards[L] */
    for (int64_t t = L - 1; t >= 0; --t) {
        d = (double)(float)rewards[t] + gamma * d;
        rtg[t] = d;
    }
    if (use_gae) {
        double a = 0.0;
        for (int64_t t = L - 1; t >= 0; --t) {
            const float vn = (t == L - 1) ? ev : values[t + 1];
            const double delta = rewards[t] + (double)(gamma_f * vn) - (double)values[t];
            a = delta + gl * a;
            adv[t] = a;
        }
    } else {
        for (int64_t t = 0; t < L; ++t) adv[t] = rtg[t] - (double)values[t];
    }
}

/* dense time-major [T,E] buffer with end flags (the build's layout); float32 in / out.
 * end_kind: 0 continue, 1 terminal (0,0), 2 bootstrapped (boot_value/boot_reward [T,E]). */
void ppoaf_oracle_gae_rtg_tmajor(const float* rewards, const float* values,
                                 const float* boot_value, const float* boot_reward,
                                 const int8_t* end_kind, int32_t T, int64_t E,
                                 double gamma, double lambd, int has_clip, double clip_lo,
                                 double clip_hi, int use_gae, float* adv_out, float* rtg_out) {
    const float gamma_f = (float)gamma;
    const double gl = gamma * lambd;
    for (int64_t e = 0; e < E; ++e) {
        double a = 0.0, r = 0.0;
        for (int32_t t = T - 1; t >= 0; --t) {
            const int64_t i = (int64_t)t * E + e;
            const int k = end_kind[i];
            float vn;
            /* terminal: end_episode(ending_value = 0, ending_reward = 0) -- the clip runs on that 0 as on any
             * ending reward (episode_info.py:450-454): visible only for clip ranges that exclude zero */
            if (k == 1) { vn = 0.f; a = 0.0; r = (double)clipf(0.f, has_clip, clip_lo, clip_hi); }
            else if (k == 2) { vn = boot_value[i]; a = 0.0; r = (double)clipf(boot_reward[i], has_clip, clip_lo, clip_hi); }
            else vn = (t + 1 < T) ? values[i + E] : 0.f;
            const double delta = (double)rewards[i] + (double)(gamma_f * vn) - (double)values[i];
            a = delta + gl * a;
            r = (double)rewards[i] + gamma * r;
            rtg_out[i] = (float)r;
            adv_out[i] = use_gae ? (float)a : (float)(r - (double)values[i]);
        }
    }
}
