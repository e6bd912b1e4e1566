// Row tiles of a 256-wide network on TWO workgroups (K12's fwd_bwd launch of the split-wgrad chain, args->row_pairs).
//
// A 16-row tile of a 256-wide network costs its ONE workgroup 8.2 k cycles of f32-MFMA pipe time and 256 KB (forward) /
// 512 KB of lines (dgrad) through the CU's 64 B/clk L1 per 256 x 256 layer pass, two output tiles per wave with the second
// tile's fragments requested only when the first is done (ppo_update_rowtile.hpp; DESIGN.md section 3: 93 k cycles, hidden
// forward 40 k, hidden backward 31 k).  Here the tile's OUTPUT COLUMNS are split over a pair of workgroups on one XCD
// (workgroup b runs on XCD b % 8: partners are 8 apart in the grid): each computes 128 of the 256 outputs of every hidden
// layer pass -- one tile per wave, K over all 256 inputs, the next pass's fragments requested as soon as this pass's have
// been consumed -- and the halves are exchanged through memory after every pass that feeds another one:
//
//   forward   h_l   = act(h_{l-1} W_l^T + b_l)          halves exchanged for l = 1 .. depth-1
//   backward  dz_{l-1} = (dz_l W_l) * act'(h_{l-1})     halves exchanged for l = depth-1 .. 2
//
// (2 depth - 3 exchanges per mini-batch; layer 0, the output layer, the distribution head / losses and dz of the last
// hidden layer are computed by both partners, identically).  An exchange is one round of data-tagged 16-byte records
// (MI355X_MICROARCH.md: {32 data bits, 32-bit tag} granules, `sc1` stores, `sc1` polling loads -- no flag, no fence, no
// ordering): the consumer polls the data itself, so a hand-over costs one store -> load latency instead of data, drain,
// flag and data again.  The tag is the mini-batch index + 1: unique inside an epoch; the caller zeroes the workspace when
// an epoch begins.  Every accumulation runs in the order of the one-workgroup body (same K order per output tile), so all
// published panels -- and with them the chain's gradients and parameters -- are BITWISE those of the one-workgroup form.
// Placement on one XCD is for speed only (the records are written through and read at agent scope: correct wherever the
// partners run); a wait is bounded (2 s) and ends in the region's error word.
#pragma once
#include "ppo_update_rowtile.hpp"

namespace ppoaf {

typedef unsigned pair_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pair_rsrc(unsigned char* block) {
    return __builtin_amdgcn_make_buffer_rsrc(block, 0, kPairRecBytes, 0x00020000);
}
// this workgroup's half panel (LDS, row stride HS, first column col0) -> its record block: thread t sends columns
// 2 (t & 63), +1 of rows t >> 6 and 8 + (t >> 6)
__device__ __forceinline__ void pair_send(unsigned char* block, const float* panel, const int HS, const int col0,
                                          const unsigned tag, const int tid) {
    const __amdgpu_buffer_rsrc_t rs = pair_rsrc(block);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = tid + kThreadsU * k;
        const float2 v = *reinterpret_cast<const float2*>(panel + (r >> 6) * HS + col0 + 2 * (r & 63));
        const pair_u32x4 q = {__float_as_uint(v.x), tag, __float_as_uint(v.y), tag};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, (unsigned)(16 * r), 0, 16 /* sc1 */);
    }
}
// the partner's half panel -> LDS; every wave waits for exactly the records it copies
__device__ __forceinline__ void pair_recv(unsigned char* block, float* panel, const int HS, const int col0,
                                          const unsigned tag, const int tid, unsigned* err) {
    const __amdgpu_buffer_rsrc_t rs = pair_rsrc(block);
    pair_u32x4 q0, q1;
    // a wait that ran out before (this launch or an earlier one of the epoch): nobody waits again, the host redoes the epoch
    const long long budget = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 0 : kPairWaitTicks;
    const long long t0 = (long long)wall_clock64();
    for (unsigned polls = 1;; ++polls) {
        q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(16 * tid), 0, 16 /* sc1 */);
        q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(16 * (tid + kThreadsU)), 0, 16 /* sc1 */);
        const bool ok = q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag;
        if (__all((int)ok)) break;
        if ((polls & 15u) == 0u && (long long)wall_clock64() - t0 > budget) {
            if ((tid & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    *reinterpret_cast<float2*>(panel + (tid >> 6) * HS + col0 + 2 * (tid & 63)) = make_float2(__uint_as_float(q0.x), __uint_as_float(q0.z));
    *reinterpret_cast<float2*>(panel + (8 + (tid >> 6)) * HS + col0 + 2 * (tid & 63)) = make_float2(__uint_as_float(q1.x), __uint_as_float(q1.z));
}

// HT = 16; hf: which half of the hidden columns this workgroup computes.  Split-wgrad chain only (the weight gradients are
// the next launch's), separate launches only (plain loads of everything a previous launch wrote).
template <int HT>
__device__ __forceinline__ void ppo_update_fwd_bwd_pair_body(const UpdateDev& u, const int which, const int g, const int hf,
                                                             const PairDev& pd) {
    constexpr int H = 16 * HT, HS = H + 4, HH = H / 2;
    static_assert(HT == 2 * kNW, "one output tile per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const auto& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim, act = nd.act;
    const int NT0 = (in_dim + 15) >> 4;
    const int INP = 16 * NT0 + 4;
    const float* P = u.params + nd.offset;
    const long B = u.B;
    const long mb = u.cursor[0] + u.mb_offset;
    const long base = mb * u.batch_stride;
    const unsigned tag = (unsigned)(mb + 1);
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * (H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    float* slab = u.sp.outpart[which] + (long)g * (nd.size - offW(depth)) - offW(depth);
    const long sp_plane = (long)u.sp.Bp * H;
    const int col0 = HH * hf, pcol0 = HH * (hf ^ 1);          // my columns, the partner's
    const int n0 = col0 + 16 * wave;                          // this wave's output tile in every hidden pass
    unsigned* err = reinterpret_cast<unsigned*>(pd.base);
    auto block = [&](int phase, int half) -> unsigned char* {
        return pd.base + pd.net_off[which] + (((long)phase * kPairMaxTiles + g) * 2 + half) * (long)kPairRecBytes;
    };
    // my 128 columns of 16 LDS rows -> rows [16 g, +16) of a [Bp][H] panel (the wgrad launch's operand): one float4 per thread
    auto publish_half = [&](const float* src, float* panel) {
        const int r = tid >> 5, c = col0 + 4 * (tid & 31);
        *reinterpret_cast<float4*>(panel + ((long)g * kRows + r) * H + c) = *reinterpret_cast<const float4*>(src + r * HS + c);
    };

    PPOAF_STAMP(0);
    float* smem = reinterpret_cast<float*>(ppo_update_smem);
    int* sRow = reinterpret_cast<int*>(smem);
    float* sMisc = smem + 16;
    float* sRowF = smem + 32;
    float* sActF = smem + 80;
    float* sBias = smem + 208;
    float* sWout = sBias + (depth + 1) * H;
    float* sX = sWout + 8 * H;
    float* sH = sX + kRows * INP;
    float* sD0 = sH + (long)depth * kRows * HS;
    float* sD1 = sD0 + kRows * HS;
    float* sOut = sD1 + kRows * HS;
    float* sDOut = sOut + kRows * kMaxOut;
    float* sScr = sDOut + kRows * kMaxOut + wave * 2 * kLineSlot;      // this wave's two line slots (mfma_rows_x_lines)

    // Request order = arrival order (a wave's loads return in order): what the first phases wait for goes out FIRST -- the
    // row indices and per-row scalars, the statistics, the input rows, layer 0's weights (staged through LDS: coalesced,
    // every line asked for once) -- then biases / output weights, and only then the 128 KB of the first hidden pass's
    // fragments, which one CU pulls at ~30 GB/s and which nothing needs before layer 0 is done.
    if (tid < kRows) {
        const long s = (long)g * kRows + tid;
        int row = -1;
        long di = -1;
        if (s < B) {
            const long p = u.perm[base + s];
            if (p >= 0 && p < u.n_rows) row = u.row_map ? u.row_map[p] : (int)p;
            di = u.pregathered ? base + s : row;
        }
        float av = 0.f, lpo = 0.f, rt = 0.f;
        if (di >= 0) {
            if (which == 0) {
                av = u.adv[di]; lpo = u.old_lp[di];
                if (u.head_kind == PPOAF_HEAD_CATEGORICAL)
                    reinterpret_cast<int*>(sActF)[tid * 8] = (int)reinterpret_cast<const int64_t*>(u.raw_actions)[di];
                else
                    for (int d = 0; d < out_dim; ++d)
                        sActF[tid * 8 + d] = reinterpret_cast<const float*>(u.raw_actions)[(long)di * out_dim + d];
            } else {
                rt = u.rtg[di];
            }
        }
        sRow[tid] = row;
        sRowF[tid] = av; sRowF[16 + tid] = lpo; sRowF[32 + tid] = rt;
    }
    if (tid == 64) {                                        // mini-batch statistics (ppo_update_rowtile.hpp, S0)
        if (which == 0) {
            float mean_f = 0.f, std_f = 1.f;
            if (u.normalize_adv) {
                const double* rec = u.adv_records + mb * 3;
                mean_f = (float)rec[1];
                std_f = (float)sqrt(rec[2] / (rec[0] - 1.0));
            }
            sMisc[0] = mean_f; sMisc[1] = std_f;
        } else {
            const int slot = (int)(mb & 1);
            float m = u.vn_mean[slot], v = u.vn_var[slot];
            double cnt = u.vn_count[slot];
            if (u.normalize_values) {
                double n = 0.0, bm = 0.0, M2 = 0.0;
                for (int r = 0; r < u.n_ranks; ++r) {
                    const double* rec = u.vn_records + (mb * u.n_ranks + r) * 3;
                    const double nb = rec[0];
                    if (nb <= 0.0) continue;
                    const double d = rec[1] - bm, nn = n + nb;
                    bm += d * (nb / nn);
                    M2 += rec[2] + d * d * n * nb / nn;
                    n = nn;
                }
                if (n > 0.0) {
                    const float batch_mean = (float)bm, batch_var = (float)(M2 / n);
                    const float delta = batch_mean - m;
                    const double new_count = cnt + n;
                    const float new_mean = (float)((double)m + (double)delta * (n / new_count));
                    const double m_2 = (double)v * cnt + (double)batch_var * n +
                                       (double)(delta * delta) * cnt * n / (cnt + n);
                    m = new_mean; v = (float)(m_2 / (cnt + n)); cnt = new_count;
                }
            }
            sMisc[2] = m; sMisc[3] = v;
            if (g == 0 && hf == 0) { u.vn_mean[slot ^ 1] = m; u.vn_var[slot ^ 1] = v; u.vn_count[slot ^ 1] = cnt; }
        }
    }
    const float* src = which == 0 ? u.obs : u.critic_obs;
    // per-epoch tables in shuffled order: a row's address depends on the cursor only (16 x in_dim <= 1024 values: two per thread)
    float xr[2] = {0.f, 0.f};
    if (u.pregathered) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * kThreadsU;
            const int s2 = idx / in_dim;
            if (idx < kRows * in_dim && (long)g * kRows + s2 < B) xr[q] = src[(base + (long)g * kRows) * in_dim + idx];
        }
    }
    // layer 0's weights -> LDS, over the planes nothing touches before the first hidden pass is stored (h_1 .., dz)
    float* sW0 = sH + kRows * HS;
    const int n4_w0 = (int)(szW0 >> 2);
    const bool stage_w0 = szW0 <= (long)(depth + 1) * kRows * HS;
    constexpr int kW0Regs = 8;                               // H * in_dim / 4 <= 8 * threads (in_dim <= 64)
    float4 w0r[kW0Regs];
    if (stage_w0) {
        const __amdgpu_buffer_rsrc_t rs0 = frag_rsrc(P);
#pragma unroll
        for (int r = 0; r < kW0Regs; ++r) {
            const int i = tid + r * kThreadsU;
            w0r[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4_w0) w0r[r] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs0, (unsigned)(16 * i), 0, 0));
        }
    }
    constexpr int kCopyRegs = 4;
    float bias_reg[kCopyRegs], wout_reg[kCopyRegs];
    const int n_bias = (depth + 1) * H, n_wout = out_dim * H;
#pragma unroll
    for (int r = 0; r < kCopyRegs; ++r) {
        const int i = tid + r * kThreadsU;
        bias_reg[r] = 0.f; wout_reg[r] = 0.f;
        if (i < n_bias) { const int l = i / H, j = i - l * H; if (l < depth || j < out_dim) bias_reg[r] = P[offB(l) + j]; }
        if (i < n_wout) wout_reg[r] = P[offW(depth) + i];
    }
    float4 fr[HT];
    load_fwd_lines_buf<HT>(P + offW(1), n0, lane, fr);

    if (stage_w0) {
#pragma unroll
        for (int r = 0; r < kW0Regs; ++r) {
            const int i = tid + r * kThreadsU;
            if (i < n4_w0) *reinterpret_cast<float4*>(sW0 + 4 * i) = w0r[r];
        }
    }
#pragma unroll
    for (int r = 0; r < kCopyRegs; ++r) {
        const int i = tid + r * kThreadsU;
        if (i < n_bias) sBias[i] = bias_reg[r];
        if (i < n_wout) sWout[i] = wout_reg[r];
    }
    for (int i = tid; i < kRows * INP; i += kThreadsU) sX[i] = 0.f;
    __syncthreads();
    PPOAF_STAMP(1);

    if (u.pregathered) {                                     // the input rows: already in registers, dropped where the row is padding
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * kThreadsU;
            const int s2 = idx / in_dim, i = idx - s2 * in_dim;
            if (idx < kRows * in_dim && sRow[s2] >= 0) sX[s2 * INP + i] = xr[q];
        }
    } else {
        for (int idx = tid; idx < kRows * in_dim; idx += kThreadsU) {
            const int s2 = idx / in_dim, i = idx - s2 * in_dim;
            const int row = sRow[s2];
            if (row >= 0) sX[s2 * INP + i] = src[(long)row * in_dim + i];
        }
    }
    __syncthreads();
    PPOAF_STAMP(2);
    if (hf == 0) {
        float* xb = u.sp.xbuf[which] + (long)g * kRows * 64;
        for (int i = tid; i < kRows * 64; i += kThreadsU) {
            const int r = i >> 6, c = i & 63;
            xb[i] = c < 16 * NT0 ? sX[r * INP + c] : 0.f;
        }
    }

    // ---- layer 0, all 256 columns by both partners (K = in_dim <= 64: cheaper than an exchange).  The B operands of both
    //      of this wave's tiles are requested first; behind them (loads return in order) the SECOND hidden pass's
    //      fragments: 128 KB that one CU pulls at ~30 GB/s -- asked for one pass ahead they arrived 18 k cycles late (stamps)
    const bool deep = depth == 3;
    float4 fr2[HT];
    if (stage_w0) {
        if (deep) load_fwd_lines_buf<HT>(P + offW(2), n0, lane, fr2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int o = (wave + kNW * t) * 16 + (lane & 15);
            const float bv = sBias[o];
            f32x4 acc = {bv, bv, bv, bv};
            const float* arow = sX + (lane & 15) * INP;
            const float* w = sW0 + o * in_dim;
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
            for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
        }
    } else {
        // (a shallow network with a wide input: no LDS to stage through -- operands of both tiles from memory, then the set)
        float bq[2][4][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* w = P + offW(0) + (long)((wave + kNW * t) * 16 + (lane & 15)) * in_dim;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * ch + 4 * j + (lane >> 4);
                    bq[t][ch][j] = k < in_dim ? w[k] : 0.f;
                }
        }
        if (deep) load_fwd_lines_buf<HT>(P + offW(2), n0, lane, fr2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int o = (wave + kNW * t) * 16 + (lane & 15);
            const float bv = sBias[o];
            f32x4 acc = {bv, bv, bv, bv};
            const float* arow = sX + (lane & 15) * INP;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                if (16 * ch < in_dim) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[16 * ch + 4 * j + (lane >> 4)], bq[t][ch][j], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
        }
    }
    __syncthreads();
    PPOAF_STAMP(3);
    publish_half(sH, u.sp.hbuf[which]);                      // h_0: the K-panel of dW_1

    // ---- hidden layers forward: my 128 columns, then the halves change hands
    for (int l = 1; l < depth; ++l) {
        const float* Hp = sH + (long)(l - 1) * kRows * HS;
        float* Hc = sH + (long)l * kRows * HS;
        const int o = n0 + (lane & 15);
        f32x4 acc;
        if (deep) {      // both forward sets were requested early; as each is consumed its registers take a backward set
            if (l == 1) { acc = mfma_rows_x_lines<HT>(Hp, HS, lane, fr, sBias[l * H + o], sScr); load_dgrad_frags_buf<HT>(P + offW(2), n0, lane, fr); }
            else { acc = mfma_rows_x_lines<HT>(Hp, HS, lane, fr2, sBias[l * H + o], sScr); load_dgrad_frags_buf<HT>(P + offW(1), n0, lane, fr2); }
        } else {
            acc = mfma_rows_x_lines<HT>(Hp, HS, lane, fr, sBias[l * H + o], sScr);
            if (l + 1 < depth) load_fwd_lines_buf<HT>(P + offW(l + 1), n0, lane, fr);
            else load_dgrad_frags_buf<HT>(P + offW(l), n0, lane, fr);              // first backward pass
        }
        if (l == 1) PPOAF_STAMP(8);
#pragma unroll
        for (int r = 0; r < 4; ++r) Hc[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
        if (l == 1) PPOAF_STAMP(11);
        __syncthreads();
        if (l == 1) PPOAF_STAMP(10);
        pair_send(block(l - 1, hf), Hc, HS, col0, tag, tid);
        if (l + 1 < depth) publish_half(Hc, u.sp.hbuf[which] + (long)l * sp_plane);       // h_l: the K-panel of dW_{l+1}
        pair_recv(block(l - 1, hf ^ 1), Hc, HS, pcol0, tag, tid, err);
        if (l == 1) PPOAF_STAMP(12);
        __syncthreads();
        if (l == 1) PPOAF_STAMP(13);
    }
    PPOAF_STAMP(4);
    const float* Hlast = sH + (long)(depth - 1) * kRows * HS;

    // ---- output layer, head and losses: both partners, identically
    if (tid < 256) {
        const int s = tid >> 4, part = tid & 15;
        for (int k = 0; k < out_dim; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < HT; ++i) acc = fmaf(Hlast[s * HS + part + 16 * i], sWout[k * H + part + 16 * i], acc);
            acc = group16_sum(acc);
            if (part == 0) sOut[s * kMaxOut + k] = acc + sBias[depth * H + k];
        }
    }
    __syncthreads();
    PPOAF_STAMP(5);
    if (wave == 0) {
        ppo_head_loss<false>(u, which, g, out_dim, P + nd.log_std_off, sRow, sRowF, sMisc, sActF, sOut, sDOut, lane, B);
    }
    __syncthreads();
    PPOAF_STAMP(6);

    // ---- output layer backward; its gradient partials leave through the first partner only
    {
        if (hf == 0 && tid < H) {
            const int i = tid;
            float h[kRows];
#pragma unroll
            for (int s = 0; s < kRows; ++s) h[s] = Hlast[s * HS + i];
            for (int k = 0; k < out_dim; ++k) {
                float d[kRows];
#pragma unroll
                for (int s = 0; s < kRows; ++s) d[s] = sDOut[s * kMaxOut + k];
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc = fmaf(d[s], h[s], acc);
                slab[offW(depth) + (long)k * H + i] = acc;
            }
        }
        const int out_pad = (out_dim + 3) & ~3;
        if (hf == 0 && tid >= 256 && tid < 256 + out_pad) {
            const int k = tid - 256;
            float acc = 0.f;
            if (k < out_dim) {
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += sDOut[s * kMaxOut + k];
            }
            slab[offB(depth) + k] = acc;
        }
        if (hf == 0 && which == 0 && u.head_kind == PPOAF_HEAD_GAUSSIAN && tid >= 320 && tid < 320 + out_pad) {
            const int d = tid - 320;
            float acc = 0.f;
            if (d < out_dim) {
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += sOut[s * kMaxOut + 8 + d];
            }
            slab[nd.log_std_off + d] = acc;
        }
        if (tid >= 256) {                                    // dz of the last hidden layer, all 256 columns
            const int t2 = tid - 256;
            const int s = t2 >> 4, ig = t2 & 15;
            float d[8];
            const float4 d0 = *reinterpret_cast<const float4*>(sDOut + s * kMaxOut);
            const float4 d1 = *reinterpret_cast<const float4*>(sDOut + s * kMaxOut + 4);
            d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
            for (int i0 = 0; i0 < HT; i0 += HT / 2) {              // (two halves: both fragment sets are live here)
                float hv[HT / 2], acc[HT / 2];
#pragma unroll
                for (int ii = 0; ii < HT / 2; ++ii) { hv[ii] = Hlast[s * HS + ig + 16 * (i0 + ii)]; acc[ii] = 0.f; }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < out_dim) {
#pragma unroll
                        for (int ii = 0; ii < HT / 2; ++ii) acc[ii] = fmaf(d[k], sWout[k * H + ig + 16 * (i0 + ii)], acc[ii]);
                    }
                }
#pragma unroll
                for (int ii = 0; ii < HT / 2; ++ii) sD0[s * HS + ig + 16 * (i0 + ii)] = acc[ii] * act_bwd(hv[ii], act);
            }
        }
    }
    __syncthreads();
    PPOAF_STAMP(7);
    publish_half(sD0, u.sp.dbuf[which] + (long)(depth - 1) * sp_plane);        // dz_{depth-1}

    // ---- hidden layers backward: dz_{l-1}, my 128 columns; exchanged while another dgrad follows
    float* Dc = sD0;
    float* Dn = sD1;
    for (int l = depth - 1; l >= 1; --l) {
        const float* Hin = sH + (long)(l - 1) * kRows * HS;
        f32x4 acc;
        if (deep) {
            acc = l == 2 ? mfma_rows_x_frags<HT>(Dc, HS, lane, fr, 0.f) : mfma_rows_x_frags<HT>(Dc, HS, lane, fr2, 0.f);
        } else {
            acc = mfma_rows_x_frags<HT>(Dc, HS, lane, fr, 0.f);
            if (l - 1 >= 1) load_dgrad_frags_buf<HT>(P + offW(l - 1), n0, lane, fr);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * (lane >> 4) + r, i = n0 + (lane & 15);
            Dn[s * HS + i] = acc[r] * act_bwd(Hin[s * HS + i], act);
        }
        __syncthreads();
        if (l == depth - 1) PPOAF_STAMP(14);
        if (l - 1 >= 1) pair_send(block(depth - 1 + (depth - 1 - l), hf), Dn, HS, col0, tag, tid);
        publish_half(Dn, u.sp.dbuf[which] + (long)(l - 1) * sp_plane);         // dz_{l-1}: the other panel of dW_{l-1}, db_{l-1}
        if (l - 1 >= 1) {
            pair_recv(block(depth - 1 + (depth - 1 - l), hf ^ 1), Dn, HS, pcol0, tag, tid, err);
            __syncthreads();
        }
        if (l == depth - 1) PPOAF_STAMP(15);
        float* t = Dc; Dc = Dn; Dn = t;
    }
    PPOAF_STAMP(9);
}

}  // namespace ppoaf
