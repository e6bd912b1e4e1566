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
#include <hip/hip_ext.h>

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

// ---------------------------------------------------------------------------------------------
// tmajor, large E: streaming kernel.  A lane owns VEC adjacent env columns (16 B loads / stores
// when VEC = 4), walks time backwards with U steps of loads in flight, state in registers.
// Shipped shape for 16-B aligned rows: <VEC 4, U 1, 1024 threads> (round 2; round 1 ran <4, 8, 256>).
// Algorithmic traffic 16 B per transition (+1 B of end flags in the dense form).
// ---------------------------------------------------------------------------------------------
template <int VEC> struct VecF;
template <> struct VecF<1> { using type = float; };
template <> struct VecF<2> { using type = float2; };
template <> struct VecF<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vec_to_arr(const typename VecF<VEC>::type& v, float (&a)[VEC]) {
    const float* p = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) a[k] = p[k];
}

template <int VEC, int U, int TPB = 256, bool NT = false, int G = 1>
__global__ __launch_bounds__(TPB) void gae_rtg_stream_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const float* __restrict__ boot_value, const float* __restrict__ boot_reward,
    const int8_t* __restrict__ end_kind, int T, long E, GaeParams p,
    float* __restrict__ adv_out, float* __restrict__ rtg_out) {
    // A workgroup covers G * TPB * VEC adjacent env columns: G column groups of TPB * VEC, a lane owning VEC
    // adjacent columns in each group.  Large workgroups matter: their waves walk time together, so every row of
    // every array is touched in bursts of TPB * 16 B (16 KB at TPB = 1024) instead of 1 KB pieces that drift apart
    // -- +24 % HBM rate at 2^28 transitions, independent of where the four arrays sit (tools/gae_sweep.py).
    using V = typename VecF<VEC>::type;
    const long base = (long)blockIdx.x * (TPB * VEC * G) + (long)threadIdx.x * VEC;
    if (base >= E) return;
    const bool dense = end_kind != nullptr;
    double A[G][VEC], R[G][VEC];
    float vnext[G][VEC];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < VEC; ++k) { A[g][k] = 0.0; R[g][k] = 0.0; vnext[g][k] = 0.f; }

    for (int t_hi = T; t_hi > 0; t_hi -= U) {
        V rv[U][G], vv[U][G];
        unsigned ek[U][G];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = t_hi - 1 - i;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long e0 = base + (long)g * (TPB * VEC);
                ek[i][g] = 0u;
                if (t >= 0 && e0 < E) {
                    const long idx = (long)t * E + e0;
                    if (NT && VEC == 4) {
                        typedef float nt4 __attribute__((ext_vector_type(4)));
                        const nt4 a = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(rewards + idx));
                        const nt4 c = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(values + idx));
                        rv[i][g] = *reinterpret_cast<const V*>(&a);
                        vv[i][g] = *reinterpret_cast<const V*>(&c);
                    } else {
                        rv[i][g] = *reinterpret_cast<const V*>(rewards + idx);
                        vv[i][g] = *reinterpret_cast<const V*>(values + idx);
                    }
                    if (dense) {
                        if (VEC == 4) ek[i][g] = *reinterpret_cast<const unsigned*>(end_kind + idx);
                        else if (VEC == 2) ek[i][g] = *reinterpret_cast<const unsigned short*>(end_kind + idx);
                        else ek[i][g] = (unsigned char)end_kind[idx];
                    } else if (t == T - 1) {
                        ek[i][g] = 0x02020202u;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = t_hi - 1 - i;
            if (t < 0) break;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long e0 = base + (long)g * (TPB * VEC);
                if (e0 >= E) continue;
                const long idx = (long)t * E + e0;
                float r[VEC], v[VEC], ao[VEC], ro[VEC];
                vec_to_arr<VEC>(rv[i][g], r);
                vec_to_arr<VEC>(vv[i][g], v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const unsigned kind = (ek[i][g] >> (8 * k)) & 0xffu;
                    float vn = vnext[g][k];
                    // terminal: ending value 0, ending reward clip(0) -- the reference clips it like any other ending
                    // reward (episode_info.py:450-454), which matters only for clip ranges that exclude zero
                    if (kind == 1u) { vn = 0.f; A[g][k] = 0.0; R[g][k] = (double)clip_reward(0.f, p); }
                    else if (kind == 2u) {
                        const long b = dense ? idx + k : e0 + k;
                        vn = boot_value[b]; A[g][k] = 0.0;
                        R[g][k] = (double)clip_reward(boot_reward[b], p);
                    }
                    const double delta = (double)r[k] + (double)(p.gamma_f * vn) - (double)v[k];
                    A[g][k] = delta + p.gl * A[g][k];
                    R[g][k] = (double)r[k] + p.gamma * R[g][k];
                    ro[k] = (float)R[g][k];
                    ao[k] = p.use_gae ? (float)A[g][k] : (float)(R[g][k] - (double)v[k]);
                    vnext[g][k] = v[k];
                }
                if (NT && VEC == 4) {
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(*reinterpret_cast<const nt4*>(ao), reinterpret_cast<nt4*>(adv_out + idx));
                    __builtin_nontemporal_store(*reinterpret_cast<const nt4*>(ro), reinterpret_cast<nt4*>(rtg_out + idx));
                } else {
                    *reinterpret_cast<V*>(adv_out + idx) = *reinterpret_cast<const V*>(ao);
                    *reinterpret_cast<V*>(rtg_out + idx) = *reinterpret_cast<const V*>(ro);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tmajor, small E (latency-bound; the 4096 x 128 configuration is 8.4 MB): the work has to be
// spread over all CUs -- a CU pulls only a few tens of GB/s -- so a workgroup takes just 16 env
// columns and splits TIME over its 512 lanes: lane = (time sub-chunk 0..3, env 0..15), 8 waves
// -> 32 chunks of TC steps per tile.  Each lane folds its chunk to the affine map of the
// recurrence (X_in -> B + M * X_in), the maps are composed (shuffles inside a wave, LDS across
// waves), then each lane replays its chunk from the true carry with its inputs still in registers.
// ---------------------------------------------------------------------------------------------
// EW env columns per workgroup, CW waves per workgroup, TC timesteps per lane chunk
// (64 / EW time sub-chunks per wave; one tile = CW * TS * TC steps).

struct Affine { double Ma, Ba, Mr, Br; };   // adv: X -> Ba + Ma X ; rtg: X -> Br + Mr X
// apply `first` (later in time), then `second` (earlier in time)
__device__ __forceinline__ Affine then(const Affine& first, const Affine& second) {
    Affine o;
    o.Ba = second.Ba + second.Ma * first.Ba; o.Ma = second.Ma * first.Ma;
    o.Br = second.Br + second.Mr * first.Br; o.Mr = second.Mr * first.Mr;
    return o;
}
__device__ __forceinline__ Affine shfl_affine(const Affine& a, int src_lane) {
    Affine o;
    o.Ma = __shfl(a.Ma, src_lane, 64); o.Ba = __shfl(a.Ba, src_lane, 64);
    o.Mr = __shfl(a.Mr, src_lane, 64); o.Br = __shfl(a.Br, src_lane, 64);
    return o;
}

template <int EW, int CW, int TC>
__global__ __launch_bounds__(64 * CW) void gae_rtg_chunked_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const float* __restrict__ boot_value, const float* __restrict__ boot_reward,
    const int8_t* __restrict__ end_kind, int T, long E, GaeParams p,
    float* __restrict__ adv_out, float* __restrict__ rtg_out) {
    constexpr int TS = 64 / EW;             // time sub-chunks per wave
    __shared__ Affine sWave[CW][EW];        // per-wave aggregate map
    __shared__ double sCarryA[EW], sCarryR[EW];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int es = lane & (EW - 1), ts = lane / EW;
    const long e = (long)blockIdx.x * EW + es;
    const bool live = e < E;
    const bool dense = end_kind != nullptr;
    if (threadIdx.x < EW) { sCarryA[threadIdx.x] = 0.0; sCarryR[threadIdx.x] = 0.0; }

    constexpr int tile = CW * TS * TC;
    for (int t_hi = T; t_hi > 0; t_hi -= tile) {
        const int c = w * TS + ts;                      // chunk 0 is the latest in time
        const int c_hi = t_hi - c * TC;
        const int c_lo = max(c_hi - TC, 0);
        const int n = max(c_hi - c_lo, 0);
        float r[TC], v[TC], vn_end[TC], er_end[TC];
        int ek[TC];
        float v_after = 0.f;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int t = c_hi - 1 - i;
            r[i] = 0.f; v[i] = 0.f; ek[i] = 0; vn_end[i] = 0.f; er_end[i] = 0.f;
            if (live && i < n) {
                const long idx = (long)t * E + e;
                r[i] = rewards[idx];
                v[i] = values[idx];
                if (dense) {
                    ek[i] = end_kind[idx];
                    if (ek[i] == 2) { vn_end[i] = boot_value[idx]; er_end[i] = boot_reward[idx]; }
                } else if (t == T - 1) {
                    ek[i] = 2; vn_end[i] = boot_value[e]; er_end[i] = boot_reward[e];
                }
            }
        }
        if (live && n > 0 && c_hi < T) v_after = values[(long)c_hi * E + e];

        // pass 1: this lane's chunk as an affine map of the state at t = c_hi
        Affine m{1.0, 0.0, 1.0, 0.0};
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            if (i < n) {
                const bool ends = ek[i] != 0;
                const float vn = ends ? vn_end[i] : (i == 0 ? v_after : v[i - 1]);
                const double delta = (double)r[i] + (double)(p.gamma_f * vn) - (double)v[i];
                m.Ba = delta + (ends ? 0.0 : p.gl * m.Ba);
                m.Ma = ends ? 0.0 : p.gl * m.Ma;
                const double rr = (double)r[i];
                if (ends) { m.Br = rr + p.gamma * (double)clip_reward(er_end[i], p); m.Mr = 0.0; }
                else      { m.Br = rr + p.gamma * m.Br; m.Mr = p.gamma * m.Mr; }
            }
        }
        // composition inside the wave: prefix over the later sub-chunks (ts' < ts), and the
        // wave aggregate (all TS sub-chunks)
        Affine prefix{1.0, 0.0, 1.0, 0.0};
        Affine agg = m;                                  // running: chunks ts..0 applied latest-first
        {
            Affine run{1.0, 0.0, 1.0, 0.0};
#pragma unroll
            for (int k = 0; k < TS; ++k) {
                const Affine mk = shfl_affine(m, es + EW * k);
                if (k == ts) prefix = run;
                run = then(run, mk);
            }
            agg = run;
        }
        if (ts == 0) sWave[w][es] = agg;
        __syncthreads();
        double A = sCarryA[es], R = sCarryR[es];
        for (int k = 0; k < w; ++k) {
            const Affine a = sWave[k][es];
            A = a.Ba + a.Ma * A; R = a.Br + a.Mr * R;
        }
        double A_tile = 0.0, R_tile = 0.0;
        if (w == CW - 1) { A_tile = agg.Ba + agg.Ma * A; R_tile = agg.Br + agg.Mr * R; }
        A = prefix.Ba + prefix.Ma * A;
        R = prefix.Br + prefix.Mr * R;

        // pass 2: replay from the true carry
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
        __syncthreads();
        if (w == CW - 1 && ts == 0) { sCarryA[es] = A_tile; sCarryR[es] = R_tile; }
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

// Kernel-exact timing: hipExtLaunchKernelGGL stamps the start / stop events with the kernel's own
// begin / end (a plain event pair around a launch also counts the dispatch gaps, ~30 us here).
extern "C" int ppoaf_gae_rtg_tmajor_timed(const float* rewards, const float* values,
                                          const float* boot_value, const float* boot_reward,
                                          const int8_t* end_kind, int32_t T, int64_t E,
                                          double gamma, double lambd, int has_clip, double clip_lo,
                                          double clip_hi, int use_gae, float* adv_out, float* rtg_out,
                                          void* start_event, void* stop_event,
                                          ppoaf_stream_t stream) {
    PPOAF_REQUIRE(rewards && values && boot_value && boot_reward && adv_out && rtg_out,
                  "gae_rtg_tmajor: null pointer");
    PPOAF_REQUIRE(T >= 0 && E >= 0, "gae_rtg_tmajor: negative shape T=%d E=%ld", T, (long)E);
    if (T == 0 || E == 0) return PPOAF_OK;
    PPOAF_REQUIRE((E + 15) / 16 <= 0x7fffffffL, "gae_rtg_tmajor: E too large");
    const GaeParams p = make_params(gamma, lambd, has_clip, clip_lo, clip_hi, use_gae);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    // Enough env columns to fill 256 CUs with several waves each -> streaming kernel (16-B lanes
    // when rows are 16-B aligned); otherwise time is split across lanes as well (chunked kernel).
    const bool aligned4 = (E % 4 == 0) && (((uintptr_t)rewards | (uintptr_t)values | (uintptr_t)adv_out |
                                            (uintptr_t)rtg_out | (uintptr_t)end_kind) % 16 == 0);
    if (E >= (1L << 20) && aligned4) {
        const long thr = E / 4;
#ifdef PPOAF_GAE_SWEEP
        // diagnostic build only (tools/gae_sweep.py): launch shape / unroll / cache policy variants of the same kernel
        int variant = 0;
        if (const char* ev = getenv("PPOAF_GAE_VARIANT")) variant = atoi(ev);
#define PPOAF_GAE_V(ID, U_, TPB_, NT_, G_)                                                                               \
        if (variant == ID) {                                                                                             \
            const long per_wg = (long)TPB_ * 4 * G_;                                                                     \
            hipExtLaunchKernelGGL((gae_rtg_stream_kernel<4, U_, TPB_, NT_, G_>), dim3((unsigned)((E + per_wg - 1) / per_wg)), \
                                  dim3(TPB_), 0, s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind,       \
                                  (int)T, (long)E, p, adv_out, rtg_out);                                                 \
            return check_launch("gae_rtg_tmajor(variant)");                                                              \
        }
        PPOAF_GAE_V(1, 4, 256, false, 1) PPOAF_GAE_V(2, 8, 256, false, 1) PPOAF_GAE_V(4, 8, 512, false, 1)
        PPOAF_GAE_V(13, 4, 1024, false, 1) PPOAF_GAE_V(14, 2, 1024, false, 1) PPOAF_GAE_V(15, 1, 1024, false, 1)
        PPOAF_GAE_V(16, 2, 1024, true, 1) PPOAF_GAE_V(17, 2, 1024, false, 2) PPOAF_GAE_V(18, 1, 1024, false, 2)
        PPOAF_GAE_V(19, 2, 512, false, 2) PPOAF_GAE_V(20, 1, 1024, false, 4) PPOAF_GAE_V(21, 3, 1024, false, 1)
        PPOAF_GAE_V(22, 2, 768, false, 1)
#undef PPOAF_GAE_V
#endif
        hipExtLaunchKernelGGL((gae_rtg_stream_kernel<4, 1, 1024>), dim3((unsigned)((thr + 1023) / 1024)), dim3(1024), 0,
                              s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind, (int)T,
                              (long)E, p, adv_out, rtg_out);
    } else if (E >= (1L << 17)) {
        hipExtLaunchKernelGGL((gae_rtg_stream_kernel<1, 8>), dim3((unsigned)((E + 255) / 256)), dim3(256), 0,
                              s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind, (int)T,
                              (long)E, p, adv_out, rtg_out);
    } else {
#ifdef PPOAF_GAE_SWEEP
        int chunk = 0;
        if (const char* ev = getenv("PPOAF_GAE_CHUNK")) chunk = atoi(ev);
#define PPOAF_GAE_C(ID, EW_, CW_, TC_)                                                                                   \
        if (chunk == ID) {                                                                                               \
            hipExtLaunchKernelGGL((gae_rtg_chunked_kernel<EW_, CW_, TC_>), dim3((unsigned)((E + EW_ - 1) / EW_)),        \
                                  dim3(64 * CW_), 0, s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind,   \
                                  (int)T, (long)E, p, adv_out, rtg_out);                                                 \
            return check_launch("gae_rtg_tmajor(chunk variant)");                                                        \
        }
        PPOAF_GAE_C(1, 32, 8, 8) PPOAF_GAE_C(2, 32, 16, 4) PPOAF_GAE_C(3, 64, 8, 16) PPOAF_GAE_C(4, 64, 16, 8)
        PPOAF_GAE_C(5, 16, 16, 2) PPOAF_GAE_C(6, 32, 4, 16) PPOAF_GAE_C(7, 16, 4, 8) PPOAF_GAE_C(8, 8, 8, 2)
        PPOAF_GAE_C(9, 32, 8, 4) PPOAF_GAE_C(10, 16, 8, 2)
#undef PPOAF_GAE_C
#endif
        // lane mapping by measurement (tools/gae_chunk_sweep.py, profiles/r04_gae_chunk_sweep.txt): 16 columns per
        // workgroup while that still leaves at most ~3 workgroups per CU, 32 columns (128-B row segments) beyond
        if (E >= 8192)
            hipExtLaunchKernelGGL((gae_rtg_chunked_kernel<32, 8, 8>), dim3((unsigned)((E + 31) / 32)), dim3(512), 0,
                                  s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind, (int)T,
                                  (long)E, p, adv_out, rtg_out);
        else
            hipExtLaunchKernelGGL((gae_rtg_chunked_kernel<16, 8, 4>), dim3((unsigned)((E + 15) / 16)), dim3(512), 0,
                                  s, e0, e1, 0, rewards, values, boot_value, boot_reward, end_kind, (int)T,
                                  (long)E, p, adv_out, rtg_out);
    }
    return check_launch("gae_rtg_tmajor");
}

extern "C" int ppoaf_gae_rtg_tmajor(const float* rewards, const float* values,
                                    const float* boot_value, const float* boot_reward,
                                    const int8_t* end_kind, int32_t T, int64_t E,
                                    double gamma, double lambd, int has_clip, double clip_lo,
                                    double clip_hi, int use_gae, float* adv_out, float* rtg_out,
                                    ppoaf_stream_t stream) {
    return ppoaf_gae_rtg_tmajor_timed(rewards, values, boot_value, boot_reward, end_kind, T, E, gamma,
                                      lambd, has_clip, clip_lo, clip_hi, use_gae, adv_out, rtg_out,
                                      nullptr, nullptr, stream);
}

// Events for the *_timed entry points (hipEvent_t behind a void*).
extern "C" void* ppoaf_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); return nullptr; }
    return (void*)e;
}
extern "C" int ppoaf_event_destroy(void* e) {
    return hipEventDestroy((hipEvent_t)e) == hipSuccess ? PPOAF_OK : PPOAF_E_LAUNCH;
}
// Blocks until `stop` has completed; returns the kernel time in ms through *ms.
extern "C" int ppoaf_event_elapsed_ms(void* start, void* stop, float* ms) {
    PPOAF_REQUIRE(start && stop && ms, "event_elapsed_ms: null pointer");
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (e != hipSuccess) { set_error("event_elapsed_ms: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
    return PPOAF_OK;
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
