// K12, split-wgrad chain: the weight-gradient launch of its three-launch form (fwd_bwd<SPLIT> -> ppoaf_ppo_update_wgrad ->
// ppoaf_ppo_update_adam(3); the default chain ends in ONE launch instead: ppo_update_tail.hip) and the host-side layout
// queries of the split workspace (panels, the row pairs' record region).
#include "ppo_update_rowtile.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace ppoaf {

// ------------------------------------------------------------------------------------------------------------
// Split-wgrad chain (the default single-rank chain): the launch between fwd_bwd<SPLIT> and Adam.  fwd_bwd's workgroups
// have published the mini-batch's inputs, hidden activations and dLoss/dz as [Bp][H] panels (u.sp); here ONE 4-wave
// workgroup forms one 16 x 16 tile of dW_l = dz_l^T . h_{l-1} over ALL B rows: every lane requests its MFMA operands
// straight from the panels (element (k, o) of dz and (k, i) of h for its k = 4 c + lane / 16: no LDS staging, all loads
// of the tile in flight at once -- the panels were written by other XCDs a moment ago, so the launch is one cold round
// trip long), wave w takes K chunks w, w + 4, ...; the four partial tiles are folded through LDS in wave order; db_l =
// the column sums of dz_l, on the tiles of input tile 0.  One workgroup per network folds the output layer's per-block
// partials in block order; the last one does the per-mini-batch bookkeeping.  Every workgroup leaves a pair of
// squared-norm partials (scaled gradients, double) for the Adam launch to add in workgroup order.  No slabs, no atomics:
// 8.7 MB of slab round trip per mini-batch at C2 become 1.6 MB of panels, and the hidden-layer wgrad MFMAs leave
// fwd_bwd's dependent chain.
// ------------------------------------------------------------------------------------------------------------
// MAXC = chunks of 16 rows a wave may own (B <= 512: 32 chunks over 4 waves)
template <int H>
__device__ __forceinline__ double split_wgrad_job(const UpdateDev& u, const int which, const int job, float* sFold /* [3][2][256] + [4][16] */) {
    constexpr int MAXC = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the chunk offsets below stay in scalar registers
    const auto& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim;
    const int B = (int)u.B;
    const long plane = (long)u.sp.Bp * H;
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * ((long)H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    float* G = u.grads + nd.offset;
    constexpr int t = H / 16, t2 = (t + 1) / 2;
    const int n_it0 = (in_dim + 15) / 16, p0 = (n_it0 + 1) / 2;
    const int n_hidden = (depth - 1) * t * t2, n_l0 = t * p0;
    const float sc = u.grad_scale;
    double q = 0.0;
    if (job < n_hidden + n_l0) {
        // 16 output rows x 32 input columns: input tiles itile and itile + 1 (the second may not exist: odd tile counts)
        int l, ot, itile, n_it;
        if (job < n_hidden) { l = 1 + job / (t * t2); const int jj = job % (t * t2); ot = jj / t2; itile = 2 * (jj % t2); n_it = t; }
        else { l = 0; const int jj = job - n_hidden; ot = jj / p0; itile = 2 * (jj % p0); n_it = n_it0; }
        const bool two = itile + 1 < n_it;                    // uniform per workgroup
        // buffer loads: resource = the layer's panel, scalar offset = chunk + row quad, vector offset = the lane's constant
        // byte offset (no vector address arithmetic per load).  Rows of the last chunk beyond B are dead rows of their
        // tile: their dz is zero and their activations finite, exactly as the slab form sums them.
        const long ldx = l >= 1 ? H : 64;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(u.sp.dbuf[which] + (long)l * plane, 0, 0xFFFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            l >= 1 ? u.sp.hbuf[which] + (long)(l - 1) * plane : u.sp.xbuf[which], 0, 0xFFFFFFFF, 0x00020000);
        const unsigned dl = 4u * (unsigned)((lane >> 4) * H + ot * 16 + (lane & 15));
        const unsigned xl = 4u * (unsigned)((lane >> 4) * (int)ldx + itile * 16 + (lane & 15));
        const int nc = (B + 15) >> 4;                         // 16-row chunks of the mini-batch
        float a[MAXC][4], x0[MAXC][4], x1[MAXC][4];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = wave + 4 * c;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[c][j] = 0.f; x0[c][j] = 0.f; x1[c][j] = 0.f; }
            if (ch < nc) {                                    // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sd = 4u * (unsigned)((16 * ch + 4 * j) * H), sx = 4u * (unsigned)((16 * ch + 4 * j) * (int)ldx);
                    a[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dl, sd, 0));
                    x0[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl, sx, 0));
                    if (two) x1[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl + 64u, sx, 0));
                }
            }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (wave + 4 * c < nc) {                          // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x0[c][j], acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x1[c][j], acc1, 0, 0, 0);
                    bsum += a[c][j];
                }
            }
        }
        // fold the four waves' partial tiles in wave order (waves 1..3 park theirs in LDS)
        if (wave > 0) {
            *reinterpret_cast<f32x4*>(sFold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
            *reinterpret_cast<f32x4*>(sFold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
        }
        // bias: column o = lane & 15 summed over this lane group's rows, then over the 4 lane groups, then over the waves
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 16) sFold[1536 + wave * 16 + lane] = bsum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                acc0 += *reinterpret_cast<const f32x4*>(sFold + ((w * 2 + 0) * 64 + lane) * 4);
                acc1 += *reinterpret_cast<const f32x4*>(sFold + ((w * 2 + 1) * 64 + lane) * 4);
            }
        }
        const long ldw = l >= 1 ? H : in_dim;
        const int i = itile * 16 + (lane & 15);               // C layout: column = lane & 15, rows 4 (lane >> 4) + r
        float bg = 0.f;
        if (wave == 0 && itile == 0 && lane < 16)
            bg = sFold[1536 + lane] + sFold[1536 + 16 + lane] + sFold[1536 + 32 + lane] + sFold[1536 + 48 + lane];
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ot * 16 + 4 * (lane >> 4) + r;
                if (i < ldw) { G[offW(l) + (long)o * ldw + i] = acc0[r]; q += (double)(acc0[r] * sc) * (acc0[r] * sc); }
                if (two && i + 16 < ldw) { G[offW(l) + (long)o * ldw + i + 16] = acc1[r]; q += (double)(acc1[r] * sc) * (acc1[r] * sc); }
            }
            if (itile == 0 && lane < 16) {
                G[offB(l) + ot * 16 + lane] = bg;
                q += (double)(bg * sc) * (bg * sc);
            }
        }
    } else {
        // output layer (+ log_std): row-block partials -> gradient, in block order
        const long seg_off = offW(depth), seg_len = nd.size - seg_off;
        const float* outpart = u.sp.outpart[which];
        const int n_hb = (B + 15) >> 4;
        for (long idx = tid; idx < seg_len; idx += kWgradThreads) {
            float acc = 0.f;
            for (int g0 = 0; g0 < n_hb; g0 += 8) {
                float pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = outpart[(long)(g0 + k < n_hb ? g0 + k : 0) * seg_len + idx];
#pragma unroll
                for (int k = 0; k < 8; ++k) if (g0 + k < n_hb) acc += pv[k];
            }
            G[seg_off + idx] = acc;
            q += (double)(acc * sc) * (acc * sc);
        }
    }
    return q;
}

template <int HA, int HC>
__global__ __launch_bounds__(kWgradThreads) void ppo_update_wgrad_kernel(UpdateDev u, int jobs_a, int jobs_c, int per_xcd) {
    __shared__ double s_red[17];
    __shared__ __attribute__((aligned(16))) float s_fold[6 * 256 + 64];
    const int b = blockIdx.x;
    if (b == 8 * per_xcd) { ppo_update_bookkeeping_split(u); return; }           // uniform per workgroup
    const int job = (b & 7) * per_xcd + (b >> 3);             // XCD b % 8 works on one run of the layer-major job list
    double q = 0.0;
    if (job < jobs_a) q = split_wgrad_job<HA>(u, 0, job, s_fold);
    else if (job < jobs_a + jobs_c) q = split_wgrad_job<HC>(u, 1, job - jobs_a, s_fold);
    const bool actor = job < jobs_a;
    q = block_sum(q, s_red);
    if (threadIdx.x == 0) {
        u.norm_scratch[6 + 2 * b] = actor ? q : 0.0;
        u.norm_scratch[7 + 2 * b] = actor ? 0.0 : q;
    }
}

template <int HA, int HC>
static int wgrad_launch(const UpdateDev& u, hipStream_t s) {
    const int ja = split_wgrad_jobs(u.net[0]), jc = split_wgrad_jobs(u.net[1]), px = split_wgrad_per_xcd(u);
    hipLaunchKernelGGL((ppo_update_wgrad_kernel<HA, HC>), dim3((unsigned)(8 * px + 1)), dim3(kWgradThreads), 0, s, u, ja, jc, px);
    return check_launch("ppo_update_wgrad");
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_update_split_workspace_bytes(const ppoaf_ppo_update_args_t* args, int64_t* bytes_out) {
    PPOAF_REQUIRE(args && bytes_out, "ppo_update_split_workspace_bytes: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    UpdateDev u;
    int rc = make_update_dev(&a, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.net[0].in_dim <= 64 && u.net[1].in_dim <= 64 && u.B <= 512,
                  "ppo_update_split_workspace_bytes: the split-wgrad chain covers in_dim <= 64 and B <= 512 (got %d / %d, %ld)",
                  u.net[0].in_dim, u.net[1].in_dim, u.B);
    size_t need = ws_layout(u, nullptr, nullptr);
    if (a.row_pairs) need = pair_region_offset(u) + pair_region_layout(u, nullptr, nullptr);
    *bytes_out = (int64_t)need;
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_row_pairs_error_offset(const ppoaf_ppo_update_args_t* args, int64_t* offset_out) {
    PPOAF_REQUIRE(args && offset_out, "ppo_update_row_pairs_error_offset: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    UpdateDev u;
    int rc = make_update_dev(&a, u);
    if (rc) return rc;
    const int ha = u.net[0].H, hc = u.net[1].H;
    const bool pairs = a.row_pairs && ((ha == 128 && hc == 256 && pair_eligible(u.net[1])) ||
                                       (ha == 256 && hc == 256 && pair_eligible(u.net[0]) && pair_eligible(u.net[1])));
    *offset_out = pairs ? (int64_t)pair_region_offset(u) : -1;
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_split_blocks(const ppoaf_ppo_update_args_t* args) {
    UpdateDev u;
    ppoaf_ppo_update_args_t a;
    if (!args) return -1;
    a = *args;
    a.split_workspace = nullptr;
    if (make_update_dev(&a, u)) return -1;
    return split_wgrad_blocks(u);
}

extern "C" int ppoaf_ppo_update_wgrad(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.split, "ppo_update_wgrad: args->split_workspace is not set (the slab chain uses ppoaf_ppo_update_reduce)");
    hipStream_t s = (hipStream_t)stream;
    const int ha = u.net[0].H, hc = u.net[1].H;
    if (ha == 32 && hc == 32) return wgrad_launch<32, 32>(u, s);
    if (ha == 64 && hc == 64) return wgrad_launch<64, 64>(u, s);
    if (ha == 128 && hc == 128) return wgrad_launch<128, 128>(u, s);
    if (ha == 256 && hc == 256) return wgrad_launch<256, 256>(u, s);
    if (ha == 128 && hc == 256) return wgrad_launch<128, 256>(u, s);
    if (ha == 64 && hc == 128) return wgrad_launch<64, 128>(u, s);
    set_error("ppo_update_wgrad: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}

