// The row-tiled forward + backward of one 16-row block of one network (K12's fwd_bwd kernel body; ppo_update.hip).
#pragma once
#include "ppo_update_dev.hpp"

namespace ppoaf {

extern __shared__ __attribute__((aligned(16))) unsigned char ppo_update_smem[];

// Diagnostic build only (-DPPOAF_STAMPS): s_memtime per phase of workgroup (0, which), wave 0,
// into a buffer nothing else reads.  The shipped library executes no stamp.
#ifdef PPOAF_STAMPS
#ifndef PPOAF_STAMP_BLOCK
#define PPOAF_STAMP_BLOCK 0        /* 0..3: an actor workgroup, 4..7: a critic workgroup */
#endif
static __device__ unsigned long long g_ppo_update_stamps[2][16];
#define PPOAF_STAMP(k)                                                                   \
    do {                                                                                 \
        if (blockIdx.x == PPOAF_STAMP_BLOCK && threadIdx.x == 0) {                       \
            unsigned long long t_;                                                       \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            g_ppo_update_stamps[0][k] = t_;                                              \
        }                                                                                \
    } while (0)
#else
#define PPOAF_STAMP(k) do {} while (0)
#endif


// SPLIT = true (split-wgrad chain): the body keeps forward, losses and dgrad, but computes NO weight gradient of a hidden
// layer.  It publishes what a complete-K wgrad launch needs instead -- its 16 rows of the input x, of every hidden
// activation h_l and of every dLoss/dz_l (u.sp: hbuf / dbuf / xbuf planes, 16 KB per layer per workgroup at H = 128
// against a 135 KB slab) -- and the output layer's partials go to u.sp.outpart (folded in block order by that launch).

// A forward weight set of one output tile: requested as whole lines and turned into fragment order through the wave's LDS
// slots when consumed (mlp_device.hpp: mfma_rows_x_lines; one CU's L1 delivers the fragment pattern itself at a third of
// that rate).  (NT = true: agent-scope loads in fragment order -- the removed persistent forms.)
template <int HT, bool NT>
__device__ __forceinline__ void load_fwd_set(const float* __restrict__ W, int n0, int lane, float4 (&fr)[HT]) {
    if constexpr (NT) load_fwd_frags<HT, true>(W, n0, lane, fr);
    else load_fwd_lines_buf<HT>(W, n0, lane, fr);
}
template <int HT, bool NT>
__device__ __forceinline__ f32x4 mfma_fwd_set(const float* __restrict__ A, int HS, int lane, const float4 (&fr)[HT], float init,
                                              float* __restrict__ scratch) {
    if constexpr (NT) return mfma_rows_x_frags<HT>(A, HS, lane, fr, init);
    else return mfma_rows_x_lines<HT>(A, HS, lane, fr, init, scratch);
}

template <int HT, bool SPLIT = false>
__device__ __forceinline__ void ppo_update_fwd_bwd_body(const UpdateDev& u, const int which, const int g) {
    // (NT, mb_extra: what the persistent forms of rounds 2-3 set -- loads at agent scope, the mini-batch's position inside a
    //  launch; the forms are gone, the constants keep the body's text as it was measured)
    constexpr bool NT = false;
    constexpr long mb_extra = 0;
    constexpr int H = 16 * HT, HS = H + 4;                 // which: 0 actor, 1 critic; g: 16-row block
    int tid_ = threadIdx.x;
    if (NT) {     // persistent form: nothing derived from the lane id may be hoisted out of the caller's mini-batch loop
        asm volatile("" : "+v"(tid_));
        __builtin_assume(tid_ >= 0 && tid_ < kThreadsU);
    }
    const int tid = tid_, lane = tid & 63, wave = NT ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const auto& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim, act = nd.act;
    const int NT0 = (in_dim + 15) >> 4;                    // 16-column tiles of the input
    const int INP = 16 * NT0 + 4;
    const float* P = u.params + nd.offset;
    const long B = u.B;
    const long mb = u.cursor[0] + u.mb_offset + mb_extra;
    const long base = mb * u.batch_stride;
    // Layer offsets inside the bucket, by arithmetic: indexing a table in the kernel arguments with a
    // loop variable compiles to a vector load of kernarg memory plus a full vmcnt drain (~6k cycles).
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * (H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    // where this block's gradient partials go: its slab of the whole bucket, or (SPLIT: only the output layer's segment is
    // produced here) its row of the output-layer partials, addressed with the same bucket offsets
    float* slab = SPLIT ? u.sp.outpart[which] + (long)g * (nd.size - offW(depth)) - offW(depth)
                        : u.slabs + (long)g * u.bucket_total + nd.offset;
    // 16 rows x H floats of LDS (row stride HS) -> rows [16 g, +16) of a [Bp][H] panel: one float4 per thread at H = 128
    auto publish_rows = [&](const float* src, float* panel) {
        float* dst = panel + (long)g * kRows * H;
        for (int i = tid; i < kRows * (H / 4); i += kThreadsU) {
            const int r = i / (H / 4), c4 = i - r * (H / 4);
            *reinterpret_cast<float4*>(dst + (long)r * H + 4 * c4) = *reinterpret_cast<const float4*>(src + r * HS + 4 * c4);
        }
    };
    const long sp_plane = SPLIT ? (long)u.sp.Bp * H : 0;

    PPOAF_STAMP(0);
    // ---- LDS carve (all offsets multiples of 16 B)
    // persistent form: an opaque zero in the LDS base keeps the per-lane LDS addresses of the body from being hoisted
    // out of the caller's mini-batch loop (hoisted, all of them are live at once)
    int zl = 0;
    if (NT) asm volatile("" : "+s"(zl));
    float* smem = reinterpret_cast<float*>(ppo_update_smem + zl);
    int* sRow = reinterpret_cast<int*>(smem);                 // [16]
    float* sMisc = smem + 16;                                 // [16]: adv mean/std, vn mean/var
    float* sRowF = smem + 32;                                 // [3][16]: adv, old log-prob, rewards-to-go
    float* sActF = smem + 80;                                 // [16][8] raw actions (float or int bits)
    float* sBias = smem + 208;                                // [(depth+1), H]
    float* sWout = sBias + (depth + 1) * H;                   // [8, H]
    float* sX = sWout + 8 * H;                                // [16, INP], zero padded
    float* sH = sX + kRows * INP;                             // depth x [16, HS]
    float* sD0 = sH + (long)depth * kRows * HS;               // [16, HS]
    float* sD1 = sD0 + kRows * HS;                            // [16, HS]
    float* sOut = sD1 + kRows * HS;                           // [16, 16]
    float* sDOut = sOut + kRows * kMaxOut;                    // [16, 16]
    float* sScr = sDOut + kRows * kMaxOut + wave * 2 * kLineSlot;   // this wave's two line slots (separate launches only)

    if (!NT && g == 0 && which == 0 && tid == 0) { u.norm_scratch[0] = 0.0; u.norm_scratch[1] = 0.0; }

    // Deep prefetch (three hidden layers, one output tile per wave): the weights depend on nothing, so the
    // fragments of BOTH hidden-to-hidden layers are requested before anything else; as each set is consumed
    // its registers are refilled with the dgrad fragments of the backward pass (W2 first, then W1).  Every
    // weight load is then in flight for at least two phases -- including the cold first touch after the
    // Adam kernel rewrote the bucket -- instead of one barrier.
    const bool has_tile = wave < HT;          // waves beyond the tile count idle in MFMA phases (H < 128)
    const bool deep = depth == 3 && HT <= kNW;
    float4 fr[HT], fr2[HT];
    // first layer with at most 16 inputs: its 4 weight values per lane are requested early as well
    const bool l0_pre = in_dim <= 16 && HT <= kNW && has_tile;
    float l0w[4] = {0.f, 0.f, 0.f, 0.f};
    float l2_touch = 0.f;
    // biases and output-layer weights: requested into registers now, stored to LDS after the row loads
    // below have been issued too -- one wait covers all of them (a store in between would serialise
    // the cold misses: this bucket was rewritten by the Adam kernel a moment ago)
    constexpr int kCopyRegs = 4;                          // (depth + 1) * H and out_dim * H are <= 4 * threads
    float bias_reg[kCopyRegs], wout_reg[kCopyRegs];
    const int n_bias = (depth + 1) * H, n_wout = out_dim * H;
    const bool copy_fits = n_bias <= kCopyRegs * kThreadsU && n_wout <= kCopyRegs * kThreadsU;
    auto request_weights = [&]() {
        // separate launches: request order = arrival order (a wave's loads return in order) -- layer 0's operands, biases and
        // output weights go out BEFORE the 2 x 8 KB per wave of hidden-layer sets, so the first layer starts after one
        // memory round trip instead of behind the whole stream (round 4)
        if (NT && deep && has_tile) {
            load_fwd_set<HT, NT>(P + offW(1), wave * 16, lane, fr);
            load_fwd_set<HT, NT>(P + offW(2), wave * 16, lane, fr2);
        }
        if (l0_pre) {
            const float* w = P + offW(0) + (long)(wave * 16 + (lane & 15)) * in_dim;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int k = 4 * j + (lane >> 4); if (k < in_dim) l0w[j] = ld1<NT>(w + k); }
        }
        // The weights were rewritten by the Adam kernel a moment ago, so this XCD's L2 does not hold them:
        // the first touch of every 128-B line of the network is requested here, before anything else, so
        // the misses overlap the index / gather / first-layer phases instead of stalling the hidden layers.
        // Not for 256-wide networks: the touches pull the whole 786 KB network through the CU's 64 B/clk vector L1 a third
        // time (after them the forward and the dgrad fragments), and every load of the prologue queues behind them (vmcnt
        // retires in order): critic workgroup 103 k -> 93 k cycles without them (stamps), hidden forward unchanged.
#ifndef PPOAF_L2_TOUCH_MAX_HT
#define PPOAF_L2_TOUCH_MAX_HT 8
#endif
        if (!deep && !NT && HT <= PPOAF_L2_TOUCH_MAX_HT) for (long i = (long)tid * 32; i < nd.size; i += (long)kThreadsU * 32) l2_touch += P[i];
        if (copy_fits) {
#pragma unroll
            for (int r = 0; r < kCopyRegs; ++r) {
                const int i = tid + r * kThreadsU;
                bias_reg[r] = 0.f; wout_reg[r] = 0.f;
                if (i < n_bias) { const int l = i / H, j = i - l * H; if (l < depth || j < out_dim) bias_reg[r] = ld1<NT>(P + offB(l) + j); }
                if (i < n_wout) wout_reg[r] = ld1<NT>(P + offW(depth) + i);
            }
        }
        if (!NT && deep && has_tile) {
            load_fwd_set<HT, NT>(P + offW(1), wave * 16, lane, fr);
            load_fwd_set<HT, NT>(P + offW(2), wave * 16, lane, fr2);
        }
    };
    // three-launch chain: the weights first (cold misses, nothing to wait for).  Persistent forms: the row loads
    // first, then the hook (the wait for the previous Adam phase), then the weights (L2 hits).
    // per-epoch tables in shuffled order: an input row's address depends on the cursor only -- requested now (16 x in_dim <=
    // 1024 values: two per thread), dropped below where the row turns out to be padding
    const float* x_src = which == 0 ? u.obs : u.critic_obs;
    const bool x_pre = !NT && u.pregathered && kRows * in_dim <= 2 * kThreadsU;
    float xr[2] = {0.f, 0.f};
    if (x_pre) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * kThreadsU;
            if (idx < kRows * in_dim && (long)g * kRows + idx / in_dim < B) xr[q] = x_src[(base + (long)g * kRows) * in_dim + idx];
        }
    }
    if (!NT) request_weights();

    if (tid < kRows) {
        const long s = (long)g * kRows + tid;
        int row = -1;
        long di = -1;                                      // where this row's inputs are read from
        if (s < B) {
            const long p = u.perm[base + s];
            if (p >= 0 && p < u.n_rows) row = u.row_map ? u.row_map[p] : (int)p;
            // per-epoch tables in shuffled order: the address depends on the cursor only, so these loads
            // go out together with the perm load instead of after it
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

    // ---- S0: everything that does not depend on the rows is requested first: rows / statistics,
    //      biases + output weights -> LDS, and this wave's first-layer weight fragments -> registers.
    if (tid == 64) {                                        // a lane of wave 1: mini-batch statistics
        if (which == 0) {
            float mean_f = 0.f, std_f = 1.f;
            if (u.normalize_adv) {                           // ppo.py:2326-2333, from the per-epoch table
                const double* rec = u.adv_records + mb * 3;
                mean_f = (float)rec[1];
                std_f = (float)sqrt(rec[2] / (rec[0] - 1.0));
            }
            sMisc[0] = mean_f; sMisc[1] = std_f;
        } else {
            const int slot = (int)(mb & 1);
            float m = ld1<NT>(u.vn_mean + slot), v = ld1<NT>(u.vn_var + slot);
            double cnt = ld1<NT>(u.vn_count + slot);
            if (u.normalize_values) {
                // Chan merge of the R per-rank records of this mini-batch, then the reference's
                // integrate (utils/stats.py:73-94) -- same arithmetic as running_moments_integrate_kernel.
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
            if (g == 0) { u.vn_mean[slot ^ 1] = m; u.vn_var[slot ^ 1] = v; u.vn_count[slot ^ 1] = cnt; }
        }
    }
    if (copy_fits) {
#pragma unroll
        for (int r = 0; r < kCopyRegs; ++r) {
            const int i = tid + r * kThreadsU;
            if (i < n_bias) sBias[i] = bias_reg[r];
            if (i < n_wout) sWout[i] = wout_reg[r];
        }
    } else {
        for (int l = 0; l <= depth; ++l) {
            const int n = (l == depth) ? out_dim : H;
            const float* bb = P + offB(l);
            for (int i = tid; i < n; i += kThreadsU) sBias[l * H + i] = ld1<NT>(bb + i);
        }
        for (int i = tid; i < out_dim * H; i += kThreadsU) sWout[i] = ld1<NT>(P + offW(depth) + i);
    }
    for (int i = tid; i < kRows * INP; i += kThreadsU) sX[i] = 0.f;
    __syncthreads();
    PPOAF_STAMP(1);

    // ---- S1: gather the input rows (K4)
    if (x_pre) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * kThreadsU;
            const int s = idx / in_dim, i = idx - s * in_dim;
            if (idx < kRows * in_dim && sRow[s] >= 0) sX[s * INP + i] = xr[q];
        }
    } else {
        const float* src = x_src;
        for (int idx = tid; idx < kRows * in_dim; idx += kThreadsU) {
            const int s = idx / in_dim, i = idx - s * in_dim;
            const int row = sRow[s];
            const long di = u.pregathered ? base + (long)g * kRows + s : row;
            if (row >= 0) sX[s * INP + i] = src[di * in_dim + i];
        }
    }
    // prefetch: fragments of the first hidden-to-hidden layer (or nothing if depth == 1)
    if (!deep && depth > 1 && has_tile) load_fwd_set<HT, NT>(P + offW(1), wave * 16, lane, fr);
    __syncthreads();
    PPOAF_STAMP(2);
    if (SPLIT) {            // the block's input rows, zero padded to 64 columns: the layer-0 wgrad's K-panel
        float* xb = u.sp.xbuf[which] + (long)g * kRows * 64;
        for (int i = tid; i < kRows * 64; i += kThreadsU) {
            const int r = i >> 6, c = i & 63;
            xb[i] = c < 16 * NT0 ? sX[r * INP + c] : 0.f;
        }
    }

    // ---- L0: first layer on MFMA, K = in_dim padded to a multiple of 4 (sX is zero padded)
    for (int nt = wave; nt < HT; nt += kNW) {
        const int o = nt * 16 + (lane & 15);
        const float bv = sBias[o];
        f32x4 acc = {bv, bv, bv, bv};
        const float* w = P + offW(0) + (long)o * in_dim;
        const float* arow = sX + (lane & 15) * INP;
        for (int k0 = 0; k0 < in_dim; k0 += 16) {            // 4 MFMA steps per 16 input columns
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 4 * j + (lane >> 4);
                bq[j] = l0_pre ? l0w[j] : (k < in_dim ? ld1<NT>(w + k) : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k0 + 4 * j + (lane >> 4)], bq[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
    }
    __syncthreads();
    PPOAF_STAMP(3);

    // ---- hidden layers forward; the next phase's fragments are requested before each barrier
    for (int l = 1; l < depth; ++l) {
        const float* Hp = sH + (long)(l - 1) * kRows * HS;
        float* Hc = sH + (long)l * kRows * HS;
        if (SPLIT) publish_rows(Hp, u.sp.hbuf[which] + (long)(l - 1) * sp_plane);      // h_{l-1}: the K-panel of dW_l
        for (int nt = wave; nt < HT; nt += kNW) {
            const int o = nt * 16 + (lane & 15);
            f32x4 acc;
            if (deep) {
                if (l == 1) {
                    acc = mfma_fwd_set<HT, NT>(Hp, HS, lane, fr, sBias[l * H + o], sScr);
                    load_dgrad_frags<HT, NT>(P + offW(2), wave * 16, lane, fr);  // first backward phase, two phases early
                } else {
                    acc = mfma_fwd_set<HT, NT>(Hp, HS, lane, fr2, sBias[l * H + o], sScr);
                    load_dgrad_frags<HT, NT>(P + offW(1), wave * 16, lane, fr2); // second backward phase
                }
            } else {
            if (nt != wave) load_fwd_set<HT, NT>(P + offW(l), nt * 16, lane, fr);
            acc = mfma_fwd_set<HT, NT>(Hp, HS, lane, fr, sBias[l * H + o], sScr);
            if (nt + kNW >= HT) {                           // last tile of this wave in this layer
                if (l + 1 < depth) load_fwd_set<HT, NT>(P + offW(l + 1), wave * 16, lane, fr);
                else load_dgrad_frags<HT, NT>(P + offW(l), wave * 16, lane, fr);   // first backward phase
            }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Hc[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
            
        }
        __syncthreads();
        
    }
    PPOAF_STAMP(4);
    const float* Hlast = sH + (long)(depth - 1) * kRows * HS;

    // ---- output layer (out_dim <= 8): VALU from LDS + 16-lane reductions
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

    // ---- distribution head + loss terms for this workgroup's rows (K6 + K3)
    if (wave == 0) {
        ppo_head_loss<NT>(u, which, g, out_dim, P + nd.log_std_off, sRow, sRowF, sMisc, sActF, sOut, sDOut, lane, B);
    }
    __syncthreads();
    PPOAF_STAMP(6);

    // ---- output layer backward (weights from LDS).  Operands are pulled into registers with
    //      independent LDS reads first; a read-per-FMA loop is LDS-latency bound (~64 cycles each).
    {
        if (tid < H) {
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
        // (SPLIT: the padding slots of the segment are written too, as zeros -- the partials row lives in a workspace whose
        //  layout changes with the mini-batch size, so nothing in it may be assumed to be zero; a slab's padding is never touched)
        const int out_pad = SPLIT ? ((out_dim + 3) & ~3) : out_dim;
        if (tid >= 256 && tid < 256 + out_pad) {
            const int k = tid - 256;
            float acc = 0.f;
            if (k < out_dim) {
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += sDOut[s * kMaxOut + k];
            }
            slab[offB(depth) + k] = acc;
        }
        if (which == 0 && u.head_kind == PPOAF_HEAD_GAUSSIAN && tid >= 320 && tid < 320 + out_pad) {
            const int d = tid - 320;
            float acc = 0.f;
            if (d < out_dim) {
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += sOut[s * kMaxOut + 8 + d];
            }
            slab[nd.log_std_off + d] = acc;
        }
        // dz_last = (dOut . W_out) * act'(Hlast): waves 4..7 (the others store dW_out above)
        if (tid >= 256) {
            const int t2 = tid - 256;
            const int s = t2 >> 4, ig = t2 & 15;
            float d[8];
            const float4 d0 = *reinterpret_cast<const float4*>(sDOut + s * kMaxOut);
            const float4 d1 = *reinterpret_cast<const float4*>(sDOut + s * kMaxOut + 4);
            d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
            float hv[HT], acc[HT];
#pragma unroll
            for (int ii = 0; ii < HT; ++ii) { hv[ii] = Hlast[s * HS + ig + 16 * ii]; acc[ii] = 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < out_dim) {
#pragma unroll
                    for (int ii = 0; ii < HT; ++ii) acc[ii] = fmaf(d[k], sWout[k * H + ig + 16 * ii], acc[ii]);
                }
            }
#pragma unroll
            for (int ii = 0; ii < HT; ++ii) sD0[s * HS + ig + 16 * ii] = acc[ii] * act_bwd(hv[ii], act);
        }
    }
    __syncthreads();
    PPOAF_STAMP(7);

    // ---- hidden layers backward: wgrad + bias grad + dgrad; `fr` holds this wave's dgrad fragments
    float* Dc = sD0;
    float* Dn = sD1;
    for (int l = depth - 1; l >= 1; --l) {
        const float* Hin = sH + (long)(l - 1) * kRows * HS;
        if (SPLIT) publish_rows(Dc, u.sp.dbuf[which] + (long)l * sp_plane);            // dz_l: the other panel of dW_l, db_l
        // dgrad first (its operands were prefetched): dh[s][i] = sum_o dz[s][o] * W[o][i]
        for (int nt = wave; nt < HT; nt += kNW) {
            f32x4 acc;
            if (deep) {
                acc = l == 2 ? mfma_rows_x_frags<HT>(Dc, HS, lane, fr, 0.f) : mfma_rows_x_frags<HT>(Dc, HS, lane, fr2, 0.f);
            } else {
            if (nt != wave) load_dgrad_frags<HT, NT>(P + offW(l), nt * 16, lane, fr);
            acc = mfma_rows_x_frags<HT>(Dc, HS, lane, fr, 0.f);
            if (nt + kNW >= HT && l - 1 >= 1) load_dgrad_frags<HT, NT>(P + offW(l - 1), wave * 16, lane, fr);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * (lane >> 4) + r, i = nt * 16 + (lane & 15);
                Dn[s * HS + i] = acc[r] * act_bwd(Hin[s * HS + i], act);
            }
        }
        if (l == depth - 1) PPOAF_STAMP(10);
        // wgrad: dW[o][i] = sum_s dz[s][o] * Hin[s][i]   (SPLIT: left to the wgrad launch, over all rows at once)
        if (!SPLIT) {
            for (int mt = wave; mt < HT; mt += kNW) {
                if (HT <= 8) wgrad_mtile_full<(HT <= 8 ? HT : 1)>(Dc, HS, Hin, HS, mt * 16, lane, slab + offW(l), H);
                else wgrad_mtile(Dc, HS, Hin, HS, mt * 16, HT, H, lane, slab + offW(l), H);
            }
            if (l == depth - 1) PPOAF_STAMP(11);
            for (int o = tid; o < H; o += kThreadsU) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += Dc[s * HS + o];
                slab[offB(l) + o] = acc;
            }
        }
        if (l == depth - 1) PPOAF_STAMP(12);
        __syncthreads();
        if (l == depth - 1) PPOAF_STAMP(13);
        float* t = Dc; Dc = Dn; Dn = t;
    }
    PPOAF_STAMP(8);

    // ---- first layer backward: dW0[o][i] = sum_s dz0[s][o] * x[s][i] on MFMA against the padded sX
    if (SPLIT) {
        publish_rows(Dc, u.sp.dbuf[which]);                   // dz_0 (its K-panel is x, published after the gather)
    } else {
        for (int mt = wave; mt < HT; mt += kNW)
            wgrad_mtile(Dc, HS, sX, INP, mt * 16, NT0, in_dim, lane, slab + offW(0), in_dim);
        for (int o = tid; o < H; o += kThreadsU) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc += Dc[s * HS + o];
            slab[offB(0) + o] = acc;
        }
    }
    if (l2_touch == 1.2345e38f) u.loss_partials[0] = l2_touch;   // keeps the early line touches alive
    PPOAF_STAMP(9);
}

// dynamic LDS the body needs for one network
constexpr size_t kRowtileLineFloats = (size_t)kNW * 2 * kLineSlot;         // the waves' line slots, behind the carve below
inline size_t rowtile_lds_floats(const NetDev& n) {
    const size_t HS = n.H + 4, INP = 16 * ((n.in_dim + 15) / 16) + 4;
    return 208 + (size_t)(n.depth + 1) * n.H + 8 * (size_t)n.H + kRows * INP + (size_t)n.depth * kRows * HS + 2 * kRows * HS +
           2 * kRows * kMaxOut;
}

}  // namespace ppoaf
