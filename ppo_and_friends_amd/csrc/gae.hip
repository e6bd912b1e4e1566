// K1: GAE advantages + rewards-to-go reverse scans (HBM-bound, 16 B/transition).
// Replaces utils/episode_info.py:223-293,419-465 of the reference.
//
// Layouts
//   tmajor: rewards/values/adv/rtg are [T,E] with E contiguous.  Lanes run along
//           E (coalesced 256 B per wave-instruction), time is walked backwards.
//           The T axis is split over the waves of a workgroup: each wave first
//           reduces its time chunk to the affine map  X_in -> B + M * X_in  of
//           the recurrence (X = A for GAE, R for rtg), the maps are composed
//           through LDS, then every wave replays its chunk from the true carry.
//           Inputs stay in registers between the two passes (one HBM read).
//   traj:   flat episode-major [N]; one wave per trajectory, 64 elements per
//           step from the tail, wave-level suffix scan of the same affine maps.
#include "common.hpp"

namespace ppoaf {

// One reverse step of both recurrences for one env.
//   adv: A_t = delta_t + (gamma*lambda) * A_{t+1},   delta_t = r + f64(fl32(gamma*Vn)) - V
//   rtg: R_t = f32(r_t) + gamma * R_{t+1}
// `ends` cuts the dependency on t+1: Vn is then the ending value and the rtg
// carry is the (clipped, float32-rounded) ending reward.
struct GaeParams {
    double gamma, gl;      // gamma, gamma*lambda
    float gamma_f;
    float clip_lo, clip_hi;
    int has_clip, use_gae;
};

__device__ __forceinline__ float clip_reward(float er, const GaeParams& p) {
    // np.clip on the python float, then float32 rounding (episode_info.py:450-457).
    // er is already float32 on the device; clip bounds are applied in float32
    // after rounding the bounds themselves, which is order-equivalent.
    if (p.has_clip) er = fminf(fmaxf(er, p.clip_lo), p.clip_hi);
    return er;
}

constexpr int TC = 16;     // timesteps per wave chunk (register resident)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gae_rtg_tmajor_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const float* __restrict__ boot_value, const float* __restrict__ boot_reward,
    const int8_t* __restrict__ end_kind, int T, long E, GaeParams p,
    float* __restrict__ adv_out, float* __restrict__ rtg_out) {
    // LDS: per (tile-chunk wave, lane): affine maps (Ma,Ba) for adv and (Mr,Br) for rtg,
    // plus the carries that cross tiles.
    __shared__ double sMa[WAVES][64], sBa[WAVES][64], sMr[WAVES][64], sBr[WAVES][64];
    __shared__ double sCarryA[64], sCarryR[64];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    const bool live = e < E;
    const bool dense_flags = end_kind != nullptr;

    if (w == 0) { sCarryA[lane] = 0.0; sCarryR[lane] = 0.0; }

    const int tile = WAVES * TC;
    // tiles walk backwards from the end of the rollout; tile k covers
    // [T - (k+1)*tile, T - k*tile) clipped at 0.
    for (int t_hi = T; t_hi > 0; t_hi -= tile) {
        // this wave's chunk: [c_lo, c_hi) ; wave 0 owns the latest chunk
        const int c_hi = t_hi - w * TC;
        const int c_lo = max(c_hi - TC, 0);
        const int n = max(c_hi - c_lo, 0);        // may be 0 for leading waves of the first tile

        float r[TC], v[TC];
        float vn_end[TC];       // ending value where the episode ends, else unused
        float er_end[TC];
        int8_t ek[TC];
        float v_after = 0.f;    // V[c_hi] when the step c_hi-1 continues
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int t = c_hi - 1 - i;           // i = 0 is the latest step of the chunk
            r[i] = 0.f; v[i] = 0.f; ek[i] = 0; vn_end[i] = 0.f; er_end[i] = 0.f;
            if (live && i < n) {
                const long idx = (long)t * E + e;
                r[i] = rewards[idx];
                v[i] = values[idx];
                if (dense_flags) {
                    ek[i] = end_kind[idx];
                    if (ek[i] == 2) { vn_end[i] = boot_value[idx]; er_end[i] = boot_reward[idx]; }
                } else if (t == T - 1) {
                    ek[i] = 2; vn_end[i] = boot_value[e]; er_end[i] = boot_reward[e];
                }
            }
        }
        if (live && n > 0 && c_hi < T) v_after = values[(long)c_hi * E + e];

        // pass 1: affine map of the chunk  X_in -> B + M*X_in  (X_in = state at t = c_hi)
        double Ma = 1.0, Ba = 0.0, Mr = 1.0, Br = 0.0;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            if (i < n) {
                const bool ends = ek[i] != 0;
                const float vn = ends ? vn_end[i] : (i == 0 ? v_after : v[i - 1]);
                const double delta = (double)r[i] + (double)(p.gamma_f * vn) - (double)v[i];
                // A = delta + gl * (ends ? 0 : A_next)
                Ba = delta + (ends ? 0.0 : p.gl * Ba);
                Ma = ends ? 0.0 : p.gl * Ma;
                const double rr = (double)r[i];
                if (ends) { Br = rr + p.gamma * (double)clip_reward(er_end[i], p); Mr = 0.0; }
                else      { Br = rr + p.gamma * Br; Mr = p.gamma * Mr; }
            }
        }
        sMa[w][lane] = Ma; sBa[w][lane] = Ba; sMr[w][lane] = Mr; sBr[w][lane] = Br;
        __syncthreads();

        // carry into this wave's chunk = carry of the tile composed through the
        // later chunks (waves 0..w-1, wave 0 latest).
        double A = sCarryA[lane], R = sCarryR[lane];
        for (int k = 0; k < w; ++k) {
            A = sBa[k][lane] + sMa[k][lane] * A;
            R = sBr[k][lane] + sMr[k][lane] * R;
        }
        // the last wave also publishes the carry for the next (earlier) tile
        double A_tile = 0.0, R_tile = 0.0;
        if (w == WAVES - 1) { A_tile = Ba + Ma * A; R_tile = Br + Mr * R; }

        // pass 2: replay the chunk sequentially from the true carry (same op
        // order as the reference's scalar loop within a chunk).
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            if (i < n) {
                const bool ends = ek[i] != 0;
                const float vn = ends ? vn_end[i] : (i == 0 ? v_after : v[i - 1]);
                const double delta = (double)r[i] + (double)(p.gamma_f * vn) - (double)v[i];
                A = delta + (ends ? 0.0 : p.gl * A);
                R = (double)r[i] + p.gamma * (ends ? (double)clip_reward(er_end[i], p) : R);
                if (live) {
                    const long idx = (long)(c_hi - 1 - i) * E + e;
                    rtg_out[idx] = (float)R;
                    adv_out[idx] = p.use_gae ? (float)A : (float)(R - (double)v[i]);
                }
            }
        }
        __syncthreads();                 // everyone has read sM*/sB*/sCarry*
        if (w == WAVES - 1) { sCarryA[lane] = A_tile; sCarryR[lane] = R_tile; }
        __syncthreads();
    }
}

// ---- ragged trajectory list: one wave per trajectory ----------------------
__global__ __launch_bounds__(256) void gae_rtg_traj_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const float* __restrict__ ending_value, const float* __restrict__ ending_reward,
    const int64_t* __restrict__ traj_start, const int32_t* __restrict__ traj_len,
    long n_traj, GaeParams p, float* __restrict__ adv_out, float* __restrict__ rtg_out) {
    const int lane = threadIdx.x & 63;
    const long traj = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (traj >= n_traj) return;                       // wave-uniform
    const long start = traj_start[traj];
    const int L = traj_len[traj];
    if (L <= 0) return;
    const float ev = ending_value[traj];
    const float er = clip_reward(ending_reward[traj], p);

    double carryA = 0.0;            // A at the position just after the current 64-chunk
    double carryR = (double)er;     // rtg carry: the ending reward seeds the scan
    float v_after = ev;             // V just after the chunk (ending value for the tail chunk)
    // chunks from the tail: chunk c covers [L - (c+1)*64, L - c*64)
    for (int hi = L; hi > 0; hi -= 64) {
        const int pos = hi - 64 + lane;               // element handled by this lane
        const bool ok = pos >= 0;
        float r = 0.f, v = 0.f;
        if (ok) { r = rewards[start + pos]; v = values[start + pos]; }
        // V[t+1]: neighbour lane, or v_after for the last lane of the chunk
        float vn = __shfl_down(v, 1, 64);
        if (lane == 63) vn = v_after;
        const double delta = ok ? ((double)r + (double)(p.gamma_f * vn) - (double)v) : 0.0;
        // affine maps; lanes that hold no element are the identity
        double Ma = ok ? p.gl : 1.0, Ba = delta;
        double Mr = ok ? p.gamma : 1.0, Br = ok ? (double)r : 0.0;
        // inclusive suffix scan (Hillis-Steele towards higher lanes)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double Ma2 = __shfl_down(Ma, d, 64), Ba2 = __shfl_down(Ba, d, 64);
            const double Mr2 = __shfl_down(Mr, d, 64), Br2 = __shfl_down(Br, d, 64);
            if (lane + d < 64) {
                Ba = Ba + Ma * Ba2; Ma = Ma * Ma2;
                Br = Br + Mr * Br2; Mr = Mr * Mr2;
            }
        }
        const double A = Ba + Ma * carryA;
        const double R = Br + Mr * carryR;
        if (ok) {
            if (rtg_out) rtg_out[start + pos] = (float)R;
            adv_out[start + pos] = p.use_gae ? (float)A : (float)(R - (double)v);
        }
        // next (earlier) chunk: carries come from this chunk's first real lane = lane 0
        // (all 64 lanes are real except in the head chunk, which is the last one processed)
        carryA = __shfl(A, 0, 64);
        carryR = __shfl(R, 0, 64);
        v_after = __shfl(v, 0, 64);
    }
}

static GaeParams make_params(double gamma, double lambd, int has_clip, double lo, double hi, int use_gae) {
    GaeParams p;
    p.gamma = gamma; p.gl = gamma * lambd; p.gamma_f = (float)gamma;
    p.has_clip = has_clip; p.clip_lo = (float)lo; p.clip_hi = (float)hi; p.use_gae = use_gae;
    return p;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_gae_rtg_tmajor(const float* rewards, const float* values,
                                    const float* boot_value, const float* boot_reward,
                                    const int8_t* end_kind, int32_t T, int64_t E,
                                    double gamma, double lambd, int has_clip, double clip_lo,
                                    double clip_hi, int use_gae, float* adv_out, float* rtg_out,
                                    ppoaf_stream_t stream) {
    PPOAF_REQUIRE(rewards && values && boot_value && boot_reward && adv_out && rtg_out,
                  "gae_rtg_tmajor: null pointer");
    PPOAF_REQUIRE(T >= 0 && E >= 0, "gae_rtg_tmajor: negative shape T=%d E=%ld", T, (long)E);
    if (T == 0 || E == 0) return PPOAF_OK;
    PPOAF_REQUIRE((E + 63) / 64 <= 0x7fffffffL, "gae_rtg_tmajor: E too large");
    const GaeParams p = make_params(gamma, lambd, has_clip, clip_lo, clip_hi, use_gae);
    const unsigned grid = (unsigned)((E + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    // more waves along T when there are few env columns (latency-bound sizes),
    // fewer when E alone fills the chip.
    const int chunks = (T + TC - 1) / TC;
    if (grid >= 4096 || chunks <= 1)
        hipLaunchKernelGGL(gae_rtg_tmajor_kernel<1>, dim3(grid), dim3(64), 0, s, rewards, values,
                           boot_value, boot_reward, end_kind, T, (long)E, p, adv_out, rtg_out);
    else if (chunks <= 4 || grid >= 1024)
        hipLaunchKernelGGL(gae_rtg_tmajor_kernel<4>, dim3(grid), dim3(256), 0, s, rewards, values,
                           boot_value, boot_reward, end_kind, T, (long)E, p, adv_out, rtg_out);
    else
        hipLaunchKernelGGL(gae_rtg_tmajor_kernel<8>, dim3(grid), dim3(512), 0, s, rewards, values,
                           boot_value, boot_reward, end_kind, T, (long)E, p, adv_out, rtg_out);
    return check_launch("gae_rtg_tmajor");
}

extern "C" int ppoaf_gae_rtg_traj(const float* rewards, const float* values,
                                  const float* ending_value, const float* ending_reward,
                                  const int64_t* traj_start, const int32_t* traj_len,
                                  int64_t n_traj, double gamma, double lambd, int has_clip,
                                  double clip_lo, double clip_hi, int use_gae, float* adv_out,
                                  float* rtg_out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(n_traj >= 0, "gae_rtg_traj: negative n_traj");
    if (n_traj == 0) return PPOAF_OK;
    PPOAF_REQUIRE(rewards && values && ending_value && ending_reward && traj_start && traj_len &&
                      adv_out,
                  "gae_rtg_traj: null pointer");
    PPOAF_REQUIRE(use_gae || rtg_out, "gae_rtg_traj: rtg_out required when use_gae == 0");
    PPOAF_REQUIRE((n_traj + 3) / 4 <= 0x7fffffffL, "gae_rtg_traj: too many trajectories");
    const GaeParams p = make_params(gamma, lambd, has_clip, clip_lo, clip_hi, use_gae);
    const unsigned grid = (unsigned)((n_traj + 3) / 4);
    hipLaunchKernelGGL(gae_rtg_traj_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rewards,
                       values, ending_value, ending_reward, traj_start, traj_len, (long)n_traj, p,
                       adv_out, rtg_out);
    return check_launch("gae_rtg_traj");
}
