// K6+K7: one rollout step for every env of a rank in ONE launch:
//   actor MLP forward -> distribution -> sample (Philox) -> log-prob      policies/ppo_policy.py:758-794
//   critic MLP forward -> value -> denormalise                            ppo.py:1052-1075, utils/misc.py:113-128
//   + the step's rows of the rollout buffer (observation copies, actions, log-probs, values)
//                                                                         policies/ppo_policy.py:638-651
// replacing, per step, two small-batch torch forwards with host round trips, torch.distributions on the
// CPU and E `EpisodeInfo.add_info` calls.
//
// grid = 2 * ceil(E/16) workgroups (XCD-grouped like the update kernel: blocks 0-3 mod 8 actor, 4-7
// critic), 512 threads: 16 env rows per workgroup, hidden layers on v_mfma_f32_16x16x4_f32 with the same
// tile code as the update's forward, so a row's log-prob here and in the first mini-batch agree bit for bit.
#include "mlp_device.hpp"

namespace ppoaf {

constexpr int kNWs = 8;
constexpr int kThreadsS = 64 * kNWs;

struct StepDev {
    NetDev net[2];
    const float* params;
    const float* obs; const float* critic_obs; long E;
    int head_kind; float min_std;
    const float* act_lo; const float* act_hi;      // per action dimension, NULL: [-1, 1] (no rescale)
    const void* forced_raw_action;                 // NULL: sample; else the raw actions to log (replay / teacher forcing)
    unsigned long long seed, offset;
    int normalize_values; const float* vn_mean; const float* vn_var;
    void* raw_action_out; void* action_out; float* logp_out; float* value_out;
    float* obs_out; float* critic_obs_out;     // buffer rows for the observation copies (may be NULL)
    int n_wg;
};

extern __shared__ __attribute__((aligned(16))) unsigned char policy_step_smem[];

template <int HT>
__device__ __forceinline__ void policy_step_body(const StepDev& u, const int which, const int g) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const NetDev& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim, act = nd.act;
    const int NT0 = (in_dim + 15) >> 4, INP = 16 * NT0 + 4;
    const float* P = u.params + nd.offset;
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * (H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    const long e0 = (long)g * kRows;

    float* smem = reinterpret_cast<float*>(policy_step_smem);
    float* sBias = smem;                                   // [(depth+1), H]
    float* sWout = sBias + (depth + 1) * H;                // [8, H]
    float* sX = sWout + 8 * H;                             // [16, INP]
    float* sH0 = sX + kRows * INP;                         // [16, HS] ping
    float* sH1 = sH0 + kRows * HS;                         // [16, HS] pong
    float* sOut = sH1 + kRows * HS;                        // [16, 16]

    for (int l = 0; l <= depth; ++l) {
        const int n = (l == depth) ? out_dim : H;
        const float* bb = P + offB(l);
        for (int i = tid; i < n; i += kThreadsS) sBias[l * H + i] = bb[i];
    }
    for (int i = tid; i < out_dim * H; i += kThreadsS) sWout[i] = P[offW(depth) + i];
    {
        const float* src = which == 0 ? u.obs : u.critic_obs;
        float* cpy = which == 0 ? u.obs_out : u.critic_obs_out;
        for (int idx = tid; idx < kRows * INP; idx += kThreadsS) {
            const int s = idx / INP, i = idx - s * INP;
            float x = 0.f;
            if (i < in_dim && e0 + s < u.E) {
                x = src[(e0 + s) * in_dim + i];
                if (cpy) cpy[(e0 + s) * in_dim + i] = x;
            }
            sX[idx] = x;
        }
    }
    float4 fr[HT];
    if (depth > 1 && wave < HT) load_fwd_frags<HT>(P + offW(1), wave * 16, lane, fr);
    __syncthreads();

    // first layer (K = in_dim, zero padded)
    for (int nt = wave; nt < HT; nt += kNWs) {
        const int o = nt * 16 + (lane & 15);
        const float bv = sBias[o];
        f32x4 acc = {bv, bv, bv, bv};
        const float* w = P + offW(0) + (long)o * in_dim;
        const float* arow = sX + (lane & 15) * INP;
        for (int k0 = 0; k0 < in_dim; k0 += 16) {
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 4 * j + (lane >> 4);
                bq[j] = k < in_dim ? w[k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k0 + 4 * j + (lane >> 4)], bq[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sH0[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
    }
    __syncthreads();
    float* Hp = sH0;
    float* Hc = sH1;
    for (int l = 1; l < depth; ++l) {
        for (int nt = wave; nt < HT; nt += kNWs) {
            if (nt != wave) load_fwd_frags<HT>(P + offW(l), nt * 16, lane, fr);
            const int o = nt * 16 + (lane & 15);
            const f32x4 acc = mfma_rows_x_frags<HT>(Hp, HS, lane, fr, sBias[l * H + o]);
            if (nt + kNWs >= HT && l + 1 < depth) load_fwd_frags<HT>(P + offW(l + 1), wave * 16, lane, fr);
#pragma unroll
            for (int r = 0; r < 4; ++r) Hc[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
        }
        __syncthreads();
        float* t = Hp; Hp = Hc; Hc = t;
    }
    // output layer
    if (tid < 256) {
        const int s = tid >> 4, part = tid & 15;
        for (int k = 0; k < out_dim; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < HT; ++i) acc = fmaf(Hp[s * HS + part + 16 * i], sWout[k * H + part + 16 * i], acc);
            acc = group16_sum(acc);
            if (part == 0) sOut[s * kMaxOut + k] = acc + sBias[depth * H + k];
        }
    }
    __syncthreads();

    // heads: one lane per env row
    if (tid < kRows && e0 + tid < u.E) {
        const int s = tid;
        const long e = e0 + s;
        if (which == 1) {
            float v = sOut[s * kMaxOut];
            if (u.normalize_values) v = u.vn_mean[0] + v * sqrtf(u.vn_var[0] + 1e-8f);   // misc.py:124-128
            u.value_out[e] = v;
        } else if (u.head_kind == PPOAF_HEAD_CATEGORICAL) {
            float p[8];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < out_dim) m = fmaxf(m, sOut[s * kMaxOut + k]);
            float ssum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { p[k] = k < out_dim ? expf(sOut[s * kMaxOut + k] - m) : 0.f; ssum += p[k]; }
            const float inv = 1.0f / ssum;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { p[k] *= inv; s2 += p[k]; }
            int a = out_dim - 1;
            float c = 0.f, pa = p[0];
            if (u.forced_raw_action) {
                const long fa = reinterpret_cast<const int64_t*>(u.forced_raw_action)[e];
                a = fa < 0 ? 0 : (fa >= out_dim ? out_dim - 1 : (int)fa);
            } else {
                const Philox4 rnd = philox4x32_10(u.seed, u.offset + (unsigned long long)e, 0u);
                const float uu = u32_to_unit(rnd.x) * s2;
                bool found = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < out_dim) {
                        c += p[k];
                        if (!found && uu < c) { a = k; found = true; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k == a) pa = p[k];
            reinterpret_cast<int64_t*>(u.raw_action_out)[e] = a;
            reinterpret_cast<int64_t*>(u.action_out)[e] = a;
            u.logp_out[e] = logf(clamp_prob_u(pa / s2));
        } else {
            const float* log_std = P + nd.log_std_off;
            float* raw = reinterpret_cast<float*>(u.raw_action_out) + e * out_dim;
            float* ac = reinterpret_cast<float*>(u.action_out) + e * out_dim;
            const bool rescale = u.act_lo != nullptr;
            const float* forced = reinterpret_cast<const float*>(u.forced_raw_action);
            float lp = 0.f, slog = 0.f;
            for (int d0 = 0; d0 < out_dim; d0 += 4) {
                const Philox4 r = philox4x32_10(u.seed, u.offset + (unsigned long long)e, (uint32_t)(d0 >> 2));
                const float u0 = u32_to_unit_open0(r.x), u1 = u32_to_unit(r.y);
                const float u2 = u32_to_unit_open0(r.z), u3 = u32_to_unit(r.w);
                const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
                float sa, ca, sb, cb;
                sincosf(6.28318530717958647692f * u1, &sa, &ca);
                sincosf(6.28318530717958647692f * u3, &sb, &cb);
                const float z[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
                for (int j = 0; j < 4 && d0 + j < out_dim; ++j) {
                    const int d = d0 + j;
                    const float sd = fmaxf(softplus_u(log_std[d]), u.min_std);
                    const float mean = sOut[s * kMaxOut + d];
                    const float x = forced ? forced[e * out_dim + d] : mean + sd * z[j];
                    raw[d] = x;
                    float a = tanhf(x);
                    slog += logf(fmaxf(1.0f - a * a, 1e-6f));
                    if (rescale) a = ((a + 1.0f) / 2.0f) * (u.act_hi[d] - u.act_lo[d]) + u.act_lo[d];   // distributions.py:580-581
                    ac[d] = a;
                    const float zz = x - mean;
                    float l = -(zz * zz) / (2.0f * sd * sd) - logf(sd) - 0.91893853320467274178f;
                    lp += fminf(fmaxf(l, -100.f), 100.f);
                }
            }
            u.logp_out[e] = lp - slog;
        }
    }
}

template <int HTA, int HTC>
__global__ __launch_bounds__(kThreadsS) void policy_step_kernel(StepDev u) {
    const int b = blockIdx.x;
    const int which = (b >> 2) & 1;
    const int g = ((b >> 3) << 2) | (b & 3);
    if (g >= u.n_wg) return;
    if (which == 0) policy_step_body<HTA>(u, 0, g);
    else policy_step_body<HTC>(u, 1, g);
}

static size_t step_lds_bytes(const StepDev& u) {
    size_t worst = 0;
    for (int w = 0; w < 2; ++w) {
        const NetDev& n = u.net[w];
        const size_t HS = n.H + 4, INP = 16 * ((n.in_dim + 15) / 16) + 4;
        const size_t f = (size_t)(n.depth + 1) * n.H + 8 * (size_t)n.H + kRows * INP + 2 * kRows * HS + kRows * kMaxOut;
        if (f * 4 > worst) worst = f * 4;
    }
    return (worst + 15) / 16 * 16;
}

template <int HTA, int HTC>
static int launch_step(const StepDev& u, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(policy_step_kernel<HTA, HTC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    const unsigned grid = 8u * (unsigned)((u.n_wg + 3) / 4);
    hipLaunchKernelGGL((policy_step_kernel<HTA, HTC>), dim3(grid), dim3(kThreadsS), lds, s, u);
    return check_launch("policy_step");
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_policy_step(const ppoaf_policy_step_args_t* a, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(a, "policy_step: null args");
    StepDev u;
    int rc = fill_net(a->actor, u.net[0], "actor");
    if (rc) return rc;
    rc = fill_net(a->critic, u.net[1], "critic");
    if (rc) return rc;
    PPOAF_REQUIRE(a->E >= 0, "policy_step: negative E");
    if (a->E == 0) return PPOAF_OK;
    PPOAF_REQUIRE(a->E <= (1L << 27), "policy_step: E too large");
    PPOAF_REQUIRE(a->params && a->obs && a->critic_obs && a->raw_action_out && a->action_out &&
                      a->logp_out && a->value_out,
                  "policy_step: null pointer");
    PPOAF_REQUIRE(a->critic.out_dim == 1, "policy_step: critic out_dim must be 1");
    PPOAF_REQUIRE(a->head_kind == PPOAF_HEAD_CATEGORICAL || a->head_kind == PPOAF_HEAD_GAUSSIAN,
                  "policy_step: head_kind=%d", a->head_kind);
    PPOAF_REQUIRE((a->head_kind == PPOAF_HEAD_GAUSSIAN) == (a->actor.log_std_offset >= 0),
                  "policy_step: log_std offset must be given exactly for the Gaussian head");
    PPOAF_REQUIRE(!a->normalize_values || (a->vn_mean && a->vn_var), "policy_step: normaliser state missing");
    u.params = a->params; u.obs = a->obs; u.critic_obs = a->critic_obs; u.E = a->E;
    PPOAF_REQUIRE((a->act_lo == nullptr) == (a->act_hi == nullptr), "policy_step: give both action bounds or neither");
    u.head_kind = a->head_kind; u.min_std = a->min_std; u.act_lo = a->act_lo; u.act_hi = a->act_hi;
    u.forced_raw_action = a->forced_raw_action;
    u.seed = a->seed; u.offset = a->offset; u.normalize_values = a->normalize_values;
    u.vn_mean = a->vn_mean; u.vn_var = a->vn_var; u.raw_action_out = a->raw_action_out;
    u.action_out = a->action_out; u.logp_out = a->logp_out; u.value_out = a->value_out;
    u.obs_out = a->obs_copy_out; u.critic_obs_out = a->critic_obs_copy_out;
    u.n_wg = (int)((a->E + kRows - 1) / kRows);
    const size_t lds = step_lds_bytes(u);
    PPOAF_REQUIRE(lds <= 160 * 1024, "policy_step: needs %zu B of LDS", lds);
    hipStream_t s = (hipStream_t)stream;
    const int ha = u.net[0].H, hc = u.net[1].H;
    if (ha == 32 && hc == 32) return launch_step<2, 2>(u, lds, s);
    if (ha == 64 && hc == 64) return launch_step<4, 4>(u, lds, s);
    if (ha == 128 && hc == 128) return launch_step<8, 8>(u, lds, s);
    if (ha == 256 && hc == 256) return launch_step<16, 16>(u, lds, s);
    if (ha == 128 && hc == 256) return launch_step<8, 16>(u, lds, s);
    if (ha == 64 && hc == 128) return launch_step<4, 8>(u, lds, s);
    set_error("policy_step: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}
