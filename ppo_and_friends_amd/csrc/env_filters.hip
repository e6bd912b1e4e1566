// K13: environment filters -- running observation / reward normalisation + clipping of one
// env step, two launches (moments -> [all-gather across ranks] -> apply).
// Replaces ObservationNormalizer / RewardNormalizer / ObservationClipper / RewardClipper
// (environments/filter_wrappers.py:113-719) in the order wrapper_utils.py:81-111 wires them.
//
// One workgroup per tracked column: (agent, feature) for the two observation streams, (agent)
// for the reward stream.  The env batch of a step is small (n * W floats, L2 resident), so a
// column walk per workgroup is launch-latency bound, not bandwidth bound.
//
// Reward stream (quirk Q3 of SURVEY.md): the reference advances the running reward of env 0,
// updates the stats with the WHOLE vector, advances env 1, updates again, ... -- n Chan merges of
// half-updated vectors per step (filter_wrappers.py:412-418).  Chan merges are exact pooled
// statistics, so the n merges equal ONE merge of the n*n pooled values; with m the running
// mean before the step
//     S1 = sum_i (n-i)(new_i - m) + i (old_i - m)        S2 = same with squares
//     count' = count + n*n_all,  mean' = m + S1/count',  var' = (count*var + S2 - S1^2/count')/count'
// (n_all = rows of all ranks: each of the n updates gathers every rank's vector, stats.py:47-50).
#include "common.hpp"

namespace ppoaf {

struct ObsF {
    const float* x; float* out; float* mean; float* var; double* count;
    int W, normalize, update, has_clip;
    float lo, hi, eps;
    int ncols;                 // G * W (0: stream absent)
    int off_mean, off_m2;      // record offsets
};
struct RewF {
    const float* reward; const uint8_t* done; const uint8_t* done2; float* out; double* rr;
    double* mean; double* var; double* count;
    int normalize, update, has_clip;
    float lo, hi;
    double gamma, eps;
    int ncols;                 // G (0: stream absent)
    int off_s1, off_s2;
};
struct FilterArgs {
    ObsF o, c;
    RewF r;
    long n;
    int rec_len;
};

__device__ __forceinline__ void obs_column_moments(const ObsF& f, int col, long n, double* rec,
                                                   double* red) {
    if (!(f.normalize && f.update)) return;
    const int g = col / f.W, c = col - g * f.W;
    const float* base = f.x + (long)g * n * f.W + c;
    const int tid = threadIdx.x, nt = blockDim.x;
    double s = 0.0;
    for (long i = tid; i < n; i += nt) s += (double)base[i * f.W];
    const double mean = block_sum(s, red) / (double)n;
    double q = 0.0;
    for (long i = tid; i < n; i += nt) { const double d = (double)base[i * f.W] - mean; q += d * d; }
    q = block_sum(q, red);
    if (tid == 0) { rec[f.off_mean + col] = mean; rec[f.off_m2 + col] = q; }
}

__global__ __launch_bounds__(256) void env_filter_moments_kernel(FilterArgs a, double* __restrict__ rec) {
    __shared__ double red[17];
    int b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0) rec[0] = (double)a.n;
    if (b < a.o.ncols) { obs_column_moments(a.o, b, a.n, rec, red); return; }
    b -= a.o.ncols;
    if (b < a.c.ncols) { obs_column_moments(a.c, b, a.n, rec, red); return; }
    b -= a.c.ncols;
    if (b >= a.r.ncols || !(a.r.normalize && a.r.update)) return;
    const long n = a.n;
    const double m = a.r.mean[b];
    const float* rw = a.r.reward + (long)b * n;
    const double* rr = a.r.rr + (long)b * n;
    double s1 = 0.0, s2 = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const double old = rr[i];
        const double nw = old * a.r.gamma + (double)rw[i];
        const double d0 = old - m, d1 = nw - m;
        const double w1 = (double)(n - i), w0 = (double)i;
        s1 += w1 * d1 + w0 * d0;
        s2 += w1 * d1 * d1 + w0 * d0 * d0;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { rec[a.r.off_s1 + b] = s1; rec[a.r.off_s2 + b] = s2; }
}

__device__ __forceinline__ void obs_column_apply(const ObsF& f, int col, long n,
                                                 const double* __restrict__ recs, int R, int rec_len,
                                                 float* sh) {
    const int g = col / f.W, c = col - g * f.W;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (f.normalize) {
        if (tid == 0) {
            float mean = f.mean[col], var = f.var[col];
            if (f.update) {
                // Chan merge of the R rank records == moments of the concatenated batch (stats.py:47-54)
                double nb = 0.0, mb = 0.0, M2 = 0.0;
                for (int r = 0; r < R; ++r) {
                    const double* rec = recs + (long)r * rec_len;
                    const double nr = rec[0];
                    if (nr <= 0.0) continue;
                    const double d = rec[f.off_mean + col] - mb, nn = nb + nr;
                    mb += d * (nr / nn);
                    M2 += rec[f.off_m2 + col] + d * d * nb * nr / nn;
                    nb = nn;
                }
                if (nb > 0.0) {
                    // stats.py:73-94, expression order of moments.hip (float32 state, float64 count)
                    const float batch_mean = (float)mb, batch_var = (float)(M2 / nb);
                    const double old_count = f.count[col];
                    const float delta = batch_mean - mean;
                    const double new_count = old_count + nb;
                    const float new_mean = (float)((double)mean + (double)delta * (nb / new_count));
                    const double m_2 = (double)var * old_count + (double)batch_var * nb +
                                       (double)(delta * delta) * old_count * nb / new_count;
                    mean = new_mean;
                    var = (float)(m_2 / new_count);
                    f.mean[col] = mean; f.var[col] = var; f.count[col] = new_count;
                }
            }
            sh[0] = mean;
            sh[1] = sqrtf(var + f.eps);
        }
        __syncthreads();
    }
    const float mean = f.normalize ? sh[0] : 0.0f, sd = f.normalize ? sh[1] : 1.0f;
    const long base = (long)g * n * f.W + c;
    for (long i = tid; i < n; i += nt) {
        float y = f.x[base + i * f.W];
        if (f.normalize) y = (y - mean) / sd;
        if (f.has_clip) y = fminf(fmaxf(y, f.lo), f.hi);
        f.out[base + i * f.W] = y;
    }
}

__global__ __launch_bounds__(256) void env_filter_apply_kernel(FilterArgs a, const double* __restrict__ recs,
                                                               int R) {
    __shared__ float sh[2];
    __shared__ double shd[1];
    int b = blockIdx.x;
    if (b < a.o.ncols) { obs_column_apply(a.o, b, a.n, recs, R, a.rec_len, sh); return; }
    b -= a.o.ncols;
    if (b < a.c.ncols) { obs_column_apply(a.c, b, a.n, recs, R, a.rec_len, sh); return; }
    b -= a.c.ncols;
    if (b >= a.r.ncols) return;
    const RewF& f = a.r;
    const long n = a.n;
    if (f.normalize) {
        if (threadIdx.x == 0) {
            double var = f.var[b];
            if (f.update) {
                double s1 = 0.0, s2 = 0.0, n_all = 0.0;
                for (int r = 0; r < R; ++r) {
                    const double* rec = recs + (long)r * a.rec_len;
                    n_all += rec[0]; s1 += rec[f.off_s1 + b]; s2 += rec[f.off_s2 + b];
                }
                const double c0 = f.count[b];
                const double cnt = c0 + (double)n * n_all;
                const double M2 = c0 * var + s2 - s1 * s1 / cnt;
                f.mean[b] = f.mean[b] + s1 / cnt;
                var = M2 / cnt;
                f.var[b] = var; f.count[b] = cnt;
            }
            shd[0] = sqrt(var + f.eps);
        }
        __syncthreads();
    }
    const double sd = f.normalize ? shd[0] : 1.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const long k = (long)b * n + i;
        const float r = f.reward[k];
        if (f.normalize) {
            const bool dn = (f.done[k] != 0) || (f.done2 && f.done2[k] != 0);
            const double old = f.rr[k];
            const double nw = f.update ? old * f.gamma + (double)r : old;   // :412-416
            f.rr[k] = dn ? 0.0 : nw;                                        // :420-424
        }
        float y = f.normalize ? (float)((double)r / sd) : r;               // :455-458
        if (f.has_clip) y = fminf(fmaxf(y, f.lo), f.hi);
        f.out[k] = y;
    }
}

static int fill_obs(const ppoaf_obs_filter_t* p, int G, int& off, ObsF& d, const char* what) {
    d = ObsF{};
    if (!p) return PPOAF_OK;
    PPOAF_REQUIRE(p->W >= 1 && p->x && p->out, "env_filter %s: bad W=%d or null data", what, p->W);
    PPOAF_REQUIRE(!p->normalize || (p->mean && p->var && p->count), "env_filter %s: null stats", what);
    PPOAF_REQUIRE(!p->has_clip || p->clip_lo <= p->clip_hi, "env_filter %s: clip range", what);
    d.x = p->x; d.out = p->out; d.mean = p->mean; d.var = p->var; d.count = p->count;
    d.W = p->W; d.normalize = p->normalize != 0; d.update = p->update != 0; d.has_clip = p->has_clip != 0;
    d.lo = p->clip_lo; d.hi = p->clip_hi; d.eps = p->eps;
    d.ncols = G * p->W;
    d.off_mean = off; d.off_m2 = off + d.ncols;
    off += 2 * d.ncols;
    return PPOAF_OK;
}

static int fill_args(const ppoaf_obs_filter_t* obs, const ppoaf_obs_filter_t* cobs,
                     const ppoaf_reward_filter_t* rew, int G, int64_t n, FilterArgs& a) {
    PPOAF_REQUIRE(G >= 1 && n >= 1, "env_filter: G=%d n=%ld", G, (long)n);
    PPOAF_REQUIRE(obs || cobs || rew, "env_filter: nothing to filter");
    int off = 1;
    int rc = fill_obs(obs, G, off, a.o, "obs");
    if (rc) return rc;
    rc = fill_obs(cobs, G, off, a.c, "critic_obs");
    if (rc) return rc;
    a.r = RewF{};
    if (rew) {
        PPOAF_REQUIRE(rew->reward && rew->out, "env_filter reward: null data");
        PPOAF_REQUIRE(!rew->normalize || (rew->done && rew->running_reward && rew->mean && rew->var && rew->count),
                      "env_filter reward: null state");
        PPOAF_REQUIRE(!rew->has_clip || rew->clip_lo <= rew->clip_hi, "env_filter reward: clip range");
        a.r.reward = rew->reward; a.r.done = rew->done; a.r.done2 = rew->done2; a.r.out = rew->out; a.r.rr = rew->running_reward;
        a.r.mean = rew->mean; a.r.var = rew->var; a.r.count = rew->count;
        a.r.normalize = rew->normalize != 0; a.r.update = rew->update != 0; a.r.has_clip = rew->has_clip != 0;
        a.r.lo = rew->clip_lo; a.r.hi = rew->clip_hi; a.r.gamma = rew->gamma; a.r.eps = rew->eps;
        a.r.ncols = G;
        a.r.off_s1 = off; a.r.off_s2 = off + G;
        off += 2 * G;
    }
    a.n = n;
    a.rec_len = off;
    return PPOAF_OK;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_env_filter_moments(const ppoaf_obs_filter_t* obs, const ppoaf_obs_filter_t* critic_obs,
                                        const ppoaf_reward_filter_t* reward, int32_t G, int64_t n,
                                        double* record, ppoaf_stream_t stream) {
    FilterArgs a;
    const int rc = fill_args(obs, critic_obs, reward, G, n, a);
    if (rc) return rc;
    PPOAF_REQUIRE(record, "env_filter_moments: null record");
    const unsigned grid = (unsigned)(a.o.ncols + a.c.ncols + a.r.ncols);
    hipLaunchKernelGGL(env_filter_moments_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, record);
    return check_launch("env_filter_moments");
}

extern "C" int ppoaf_env_filter_apply(const ppoaf_obs_filter_t* obs, const ppoaf_obs_filter_t* critic_obs,
                                      const ppoaf_reward_filter_t* reward, int32_t G, int64_t n,
                                      const double* records, int32_t R, ppoaf_stream_t stream) {
    FilterArgs a;
    const int rc = fill_args(obs, critic_obs, reward, G, n, a);
    if (rc) return rc;
    const bool updating = (a.o.normalize && a.o.update) || (a.c.normalize && a.c.update) ||
                          (a.r.normalize && a.r.update);
    PPOAF_REQUIRE(!updating || (records && R >= 1), "env_filter_apply: updating stats needs R >= 1 records");
    const unsigned grid = (unsigned)(a.o.ncols + a.c.ncols + a.r.ncols);
    hipLaunchKernelGGL(env_filter_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, records,
                       (int)R);
    return check_launch("env_filter_apply");
}
