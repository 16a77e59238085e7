// mlp1 + mlp2 of one FGNN block backward in ONE launch, on 16-pixel tiles (round 6; the 32-pixel original: mlp_bwd_pair.hip).
// Replaces the autograd of models/blocks_emb.py:16-27 (two MlpBlock_Real, models/layers.py:109-131, on the SAME input).
//
// Same pairing as mlp_bwd_pair.hip -- the two waves of a SIMD work on the same pixels, wave p runs mlp1, wave p + 4 mlp2, each
// holds only its own MLP's weight-gradient accumulators -- and the same interface (work unit and S1/S2 record = a 32-pixel
// tile, fgnn_mlp_bwd_args, partial layout), so it is a drop-in for fgnn_mlp_bwd_pair.  What changes is the tile CODE:
//   * a 32-pixel tile is processed as two 16-pixel halves on v_mfma_f32_16x16x4_f32 (fgnn_t16.h): half the registers per
//     tile, so the per-graph records {mean, a, beta} of the input and {mean, ca, cb, cc} of the output stay in REGISTERS for
//     all tiles of a graph (32-pixel kernel: 32 ds_read_b128 per tile), the hidden activations stay in registers for the ReLU
//     masks (32 ds_read_b32 per tile), the normalised input is staged ONCE (it was normalised twice), and the first MFMA of a
//     launch waits for half as many bytes;
//   * the gradient of the shared input is summed by the matrix pipe: the mlp1 wave starts its layer-0 dgrad chain from the
//     gradient mlp3 left in HBM, hands the fragment over through LDS, the mlp2 wave starts ITS chain from that fragment and
//     stores -- no vector adds, no zero-initialised accumulators (d_in = fma chain over (d_in3, mlp1 terms, mlp2 terms); the
//     32-pixel kernel rounds the three sums separately, both are exact-fp32 evaluations of the same sum);
//   * the padding mask is applied to dz only (a wave-uniform branch that full tiles skip): every weight-gradient operand and
//     every dx of a padding pixel is then 0 whatever its column of the recomputed chain holds.
// The recomputed hidden activations are bit-identical to the forward's (fgnn_t16.h), so the ReLU masks are the forward's.
// Constant-size and ragged batches (nvalid, optionally with fgnn_ragged_tile_ranges); depth 3, 32-channel input slab.
#include "fgnn_t16.h"
#include "fgnn_pack.h"

#ifdef FGNN_PHASES
// Debug build only (-DFGNN_PHASES): per-wave cycle stamps of the half-tile phases, summed over the wave's halves (tools/gpu_phases_t16.py)
__device__ unsigned long long *g_phase16_buf = nullptr;
#define PH_DECL unsigned long long ph_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last_ = __builtin_amdgcn_s_memtime();
#define PH(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[k] += t_ - ph_last_; ph_last_ = t_; }
#define PH_FLUSH if (g_phase16_buf && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; ++k_) g_phase16_buf[((long long)blockIdx.x * NW + wv) * 16 + k_] = ph_[k_]; }
#else
#define PH_DECL
#define PH(k)
#define PH_FLUSH
#endif

#ifndef FGNN_PRIO
#define FGNN_PRIO 1         // static priority of the mlp2 waves (0: none)
#endif

namespace {

using namespace t16;

constexpr int BWD_WG = 256;          // persistent workgroups (one per CU) = rows of each wpart
constexpr int NW = 8;                // waves per workgroup: 4 pairs
constexpr int NP = 4;

DEVI float4 coef_from_sums(const float4 n, const float2 sv, float nv) {
    const float m = nv * nv;
    float4 k;
    k.x = n.x;
    k.y = n.y;
    k.z = m > 0.f ? -n.y * sv.y * n.w / m : 0.f;
    k.w = m > 0.f ? -n.y * sv.x / m : 0.f;
    return k;
}
DEVI float4 coef_record(const fgnn_mlp_bwd_args &A, int g, int ch, int nv) {
    if (A.coef) return reinterpret_cast<const float4 *>(A.coef)[(long long)g * FGNN_H + ch];
    const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + ch];
    const float2 sv = reinterpret_cast<const float2 *>(A.s12)[(long long)g * FGNN_H + ch];
    return coef_from_sums(n, sv, (float)nv);
}
// vertex count of graph g through a buffer descriptor (an empty one without nvalid: returns 0, no access).  `nvalid ? nvalid[g] : N`
// compiles to ONE load through a select of the two addresses -- a FLAT load, which makes every later s_waitcnt of the tile loop
// a full drain (flat operations complete out of order with respect to buffer loads)
DEVI int graph_nv(const rsrc_t &rnv, bool ragged, int g, int N) {
    const int v = __builtin_amdgcn_raw_buffer_load_b32(rnv, g * 4, 0, 0);
    return __builtin_amdgcn_readfirstlane(ragged ? v : N);
}

struct PairLayout16 {
    static constexpr int DEPTH = 3;
    static constexpr PkBwd PK = pk_bwd(32, 0, DEPTH);                      // one image per MLP: fgnn_pack.h (kind 5)
    static constexpr int OFF_W0 = PK.off_w1a, OFF_W1 = PK.off_wh, OFF_WT1 = PK.off_wt, OFF_WT2 = PK.off_wt + 16, OFF_WT0 = PK.off_wt0a;
    static constexpr int BIAS_F = PK.bias_f;
    static constexpr int WEIGHT_F = pk_pad_floats(PK.floats);             // floats per image (whole KiB: global_load_lds)
    static constexpr int NSLOT = 4;                                       // per wave: x_a, h1, h2 / dpre_1, dz / dpre_0
    static constexpr int PCOUNT = 32 * 32 + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int TILE_OFF = 2 * WEIGHT_F;
    static constexpr int XCH_OFF = TILE_OFF + NW * NSLOT * TILE_F;        // per pair: the handed-over dx fragment [s][lane]
    static constexpr int FLAG_OFF = XCH_OFF + NP * 1024;                  // (two slots per pair); per pair: ready, consumed counters (+ padding)
    static constexpr int REC_OFF = FLAG_OFF + 4 * NP;                     // per wave: {coef[32], nrm[32]} float4 (graph changes only)
    static constexpr int LIVE_OFF = REC_OFF + NW * 256;                   // SKIP: the range's live tiles (build_live_list) + NW counters
    static constexpr int MAIN_F = LIVE_OFF + LIVE_LIST_CAP + NW;
    static constexpr int RED_F = NW * PCOUNT;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

struct PairArgs {
    fgnn_mlp_bwd_args m[2];
};

// SKIP (ragged batches with ranges): work-balanced tile range from fgnn_ragged_tile_ranges, padding-only tiles are stepped over
// (the two waves of a pair walk the same tile sequence, so the hand-over protocol is unchanged)
// HAS_DX: the input gradient exists (every block but one whose input is the model input)
template <bool SKIP, bool HAS_DX>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_pair_t16_kernel(const PairArgs P, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = PairLayout16;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FGNN_SWAP_ROLES      // measurement switch: the older waves (0..3) run mlp2
    const int role = 1 - (wv >> 2), pair = wv & 3;
#else
    const int role = wv >> 2, pair = wv & 3;
#endif            // role 0: mlp1 (starts the dx chain), role 1: mlp2 (finishes it, stores, emits)
    const int px = lane & 15, q = lane >> 4;
    const fgnn_mlp_bwd_args &A = P.m[role];
    const int P2 = A.N * A.N;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vdy = make_view(A.dy, A.dgstride, A.ldd, A.G);
    const View vz = make_view(A.z, A.zgstride, A.ldz, A.G);
    const View vdx = make_view(P.m[1].dxa, P.m[1].dxa_gstride, P.m[1].dxa_ld, P.m[1].G);

    PH_DECL
    if (FGNN_PRIO > 0 && role == 1) __builtin_amdgcn_s_setprio(FGNN_PRIO);
    float *wl = smem + role * L::WEIGHT_F;              // this wave's MLP image
    float *my = smem + L::TILE_OFF + wv * (L::NSLOT * TILE_F);
    float *XA = my, *S0 = my + TILE_F, *S1 = my + 2 * TILE_F, *S2 = my + 3 * TILE_F;
    float *XCH = smem + L::XCH_OFF + pair * 1024;
    int *flags = reinterpret_cast<int *>(smem + L::FLAG_OFF) + 4 * pair;     // [0] = fragments handed over, [1] = fragments consumed
    const int lane_base = tile_lane_base(px, q);

    f32x4 dW0[4], dW1[4], dW2[4];
    float db0[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, db2[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) dW0[k] = dW1[k] = dW2[k] = zero4();

    const int nwg = gridDim.x;
    const int qq = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * qq + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + qq + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const bool normA = A.a.nrm != nullptr;
    constexpr bool has_dx = HAS_DX;
    const bool emit = role == 1 && normA && has_dx && P.m[1].s12part != nullptr;
    const bool rmw = has_dx && P.m[1].accumulate_a;
    const bool ragged = A.nvalid != nullptr;
    const rsrc_t rnv = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.nvalid), 0, ragged ? A.G * 4 : 0, 0x00020000);
    const float rcpN = 1.f / (float)A.N;

    // Prologue = ONE memory round trip: both operand images (straight into LDS), the first half tile, the per-graph records.
    PH(14)              // kernel arguments arrived (descriptors built)
    pk_glds<NW>(smem, P.m[0].packed, L::WEIGHT_F, wv, lane);
    pk_glds<NW>(smem + L::WEIGHT_F, P.m[1].packed, L::WEIGHT_F, wv, lane);
    PH(15)              // image loads issued

    // per-graph records of this lane's 8 channels, kept in registers for all tiles of a graph
    float mean[8], av[8], beta[8], kx[8], ky[8], kz[8], kw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        mean[s] = 0.f;
        av[s] = 1.f;
        beta[s] = normA && A.a.beta ? A.a.beta[chan(s, 0) + chan_q(q)] : 0.f;
        kx[s] = ky[s] = kz[s] = kw[s] = 0.f;
    }
    // at a graph change: one channel per lane (lanes 0..31) fetches / derives the records, a wave-private LDS copy hands every lane
    // its 8 channels (the derivation -- two divisions per channel -- is then one instance per lane instead of eight)
    float4 *rec = reinterpret_cast<float4 *>(smem + L::REC_OFF) + wv * 64;
    auto fetch_records = [&](int g, int nv) {
        if (lane < 32) {
            rec[lane] = coef_record(A, g, lane, nv);
            if (normA) rec[32 + lane] = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)g * A.a.C + lane];
        }
    };
    auto read_records = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 k4 = rec[chan_s(s) + chan_q(q)];
            kx[s] = k4.x;
            ky[s] = k4.y;
            kz[s] = k4.z;
            kw[s] = k4.w;
            if (normA) {
                const float4 n = rec[32 + chan_s(s) + chan_q(q)];
                mean[s] = n.x;
                av[s] = n.y;
            }
        }
    };

    // loop-carried loads: every slab of a half is requested while the PREVIOUS half computes (x after its layer-1 stage, dy / z as
    // soon as dz has consumed their registers, the old d_in after the hand-over), so no wave waits for HBM inside a half
    float xa[8], dyr[8], zr[8], oldr[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) oldr[s] = 0.f;
    int cached_g = -1, cur_nv = A.N;
    int tile = T0 + pair;
    int live_cnt = 0;                                // SKIP: the pairs take the range's LIVE tiles in turn (fgnn_common.h)
    if constexpr (SKIP) tile = __builtin_amdgcn_readfirstlane(next_owned_live_tile_p(T0, T1, live_cnt, pair, NP, tpg, FGNN_TILE, A.N, A.nvalid));
    int hf = 0;
    {
        const bool act = tile < T1;
        const int g = __builtin_amdgcn_readfirstlane(act ? tile / tpg : 0);
        const int p = (act ? tile - g * tpg : 0) * 32 + px;
        const bool inb0 = act && p < P2;
        load8(xa, va, lane_voff(va, q, p, inb0), g * va.gs4);
        if (act) {
            cur_nv = graph_nv(rnv, ragged, g, A.N);
            fetch_records(g, cur_nv);
            cached_g = g;
        }
    }
    PH(12)              // prologue a: kernel arguments, descriptors, every load of the prologue issued
    if (threadIdx.x < 4 * NP) reinterpret_cast<int *>(smem + L::FLAG_OFF)[threadIdx.x] = 0;
    PH(13)              // prologue b: images (and everything requested before them) arrived, written to LDS
    // SKIP: the pairs' later tiles come from the list of the range's live tiles (the first one was found by scanning)
    int *live_list = reinterpret_cast<int *>(smem + L::LIVE_OFF);
    int nlive = 0, li = pair;
    if constexpr (SKIP) nlive = build_live_list(live_list, live_list + LIVE_LIST_CAP, T0, T1, tpg, FGNN_TILE, A.N, A.nvalid, threadIdx.x, 64 * NW);
    const bool use_list = SKIP && nlive <= LIVE_LIST_CAP;
    __syncthreads();
    // The first half's dy / z / old d_in are requested only now: the barrier above drains every outstanding load, and at a kernel's start
    // all CUs fetch at once (~ 10 B per cycle and CU) -- the first MFMA should wait for the images and x (57 KB per CU), not for these 48 KB
    {
        const bool act = tile < T1;
        const int g = __builtin_amdgcn_readfirstlane(act ? tile / tpg : 0);
        const int p = (act ? tile - g * tpg : 0) * 32 + px;
        const bool inb0 = act && p < P2;
        load8(dyr, vdy, lane_voff(vdy, q, p, inb0), g * vdy.gs4);
        load8(zr, vz, lane_voff(vz, q, p, inb0), g * vz.gs4);
        if (has_dx) load8(oldr, vdx, lane_voff(vdx, q, p, inb0 && rmw && role == 0), g * vdx.gs4);
    }
    if (cached_g >= 0) read_records();
    PH(9)               // prologue

#ifdef FGNN_STAGGER         // measurement switch: the mlp2 waves start late by FGNN_STAGGER x 64 cycles
    if (role == 1) __builtin_amdgcn_s_sleep(FGNN_STAGGER);
#endif
    // Static priority for the mlp2 waves: they are the younger half of the workgroup (waves 4..7) -- the loser of every arbitration for the
    // SIMD's matrix / vector issue -- AND carry the longer half (hand-over wait, dgrad at the very end, store, sums), so they set the
    // kernel's duration while the mlp1 waves idle ~ 15 % at the hand-over slots.  One s_setprio before the loop: 62.7 -> 59.3 us.
    // (set at the top of the kernel: the prologue is arbitrated the same way)
    int hk = 0;                                      // fragments this pair has handed over so far (two slots: the mlp1 wave may run a half ahead)
    // role 1: S1 / S2 of the input slab's producer, accumulated PER LANE (pixel column, 8 channels) over the pair's consecutive tiles of
    // a graph and reduced over the pixels once, at the pair's last tile of the graph -- the consumers (fgnn_gn_bwd_coef_tiles, the
    // prologue of fgnn_mlp_bwd) only sum a graph's tile records, so the other tiles of the run carry zeros
    float es1[8], es2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) es1[s] = es2[s] = 0.f;
    while (tile < T1) {
        int tnext = tile + NP;
        if constexpr (SKIP) {
            if (hf == 1) {
                if (use_list) {
                    li += NP;
                    tnext = __builtin_amdgcn_readfirstlane(li < nlive ? live_list[li] : T1);
                } else {
                    tnext = __builtin_amdgcn_readfirstlane(next_owned_live_tile_p(tile + 1, T1, live_cnt, pair, NP, tpg, FGNN_TILE, A.N, A.nvalid));
                }
            }
        }
        const int g = __builtin_amdgcn_readfirstlane(tile / tpg), tt = tile - g * tpg;
        const int p = tt * 32 + 16 * hf + px;
        const bool inb = p < P2;
        if (g != cached_g) {
            cur_nv = graph_nv(rnv, ragged, g, A.N);
            fetch_records(g, cur_nv);
            read_records();
            cached_g = g;
        }
        bool valid = inb;
        if (ragged) {
            int i, jj;
            row_col(p, A.N, rcpN, i, jj);
            valid = inb && i < cur_nv && jj < cur_nv;
        }
        const unsigned long long vmask = __ballot(valid);
        const bool full = vmask == ~0ull;       // (a half without a valid pixel -- rare -- runs masked like any other: no branch around the accumulators)

        // the next half's input slab: the other half of this tile, or the first half of the pair's next tile
        const int ntile = hf == 0 ? tile : tnext, nhf = hf ^ 1;
        int np, ng;
        bool ninb;
        {
            const bool act = ntile < T1;
            ng = __builtin_amdgcn_readfirstlane(act ? ntile / tpg : 0);
            np = (act ? ntile - ng * tpg : 0) * 32 + 16 * nhf + px;
            ninb = act && np < P2;
        }

        {
            f32x4 dx[2];
            const int dvoff = lane_voff(vdx, q, p, inb), ds0 = g * vdx.gs4;

            // ---- forward recompute of the hidden activations (bit-identical to the forward's chain) ----
            float h1[8], h2[8], u[8];
            {
                float ya[8];        // (without a record: mean = 0, a = 1, beta = 0, i.e. x itself)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    u[s] = xa[s] - mean[s];             // (kept for the S2 sums; xa itself is free for the next half's loads)
                    ya[s] = u[s] * av[s] + beta[s];
                }
                stage8(XA, lane_base, ya);
                f32x4 acc[2];
                load_bias(acc, wl + L::BIAS_F, 0, q);
                gemm32<L::OFF_W0>(acc, wl, ya, lane);
#pragma unroll
                for (int s = 0; s < 8; ++s) h1[s] = relu1(acc[s >> 2][s & 3]);
                stage8(S0, lane_base, h1);
                load_bias(acc, wl + L::BIAS_F, 1, q);
                gemm32<L::OFF_W1>(acc, wl, h1, lane);
#pragma unroll
                for (int s = 0; s < 8; ++s) h2[s] = relu1(acc[s >> 2][s & 3]);
                stage8(S1, lane_base, h2);
            }
            PH(1)       // x arrived, normalised, two layers recomputed and staged
            // ---- dz from (dy, z, coef); the ONLY place the padding mask is applied ----
            float dpre[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) dpre[s] = fmaf(kz[s], zr[s] - kx[s], fmaf(ky[s], dyr[s], kw[s]));
            if (!full) {
#pragma unroll
                for (int s = 0; s < 8; ++s) dpre[s] = valid ? dpre[s] : 0.f;
            }
            stage8(S2, lane_base, dpre);
            load8(dyr, vdy, lane_voff(vdy, q, np, ninb), ng * vdy.gs4);      // the next half's dy / z into the registers just consumed
            load8(zr, vz, lane_voff(vz, q, np, ninb), ng * vz.gs4);
            PH(2)       // dz (waits for dy, z)
            __builtin_amdgcn_sched_barrier(0);
            // ---- layer 2: dgrad, weight gradient (dz x h2), ReLU mask of h2 ----
            {
                f32x4 a2[2];
                a2[0] = a2[1] = zero4();
                gemm32<L::OFF_WT2>(a2, wl, dpre, lane);
                wgrad16(dW2, db2, S2, S1, lane);
#pragma unroll
                for (int s = 0; s < 8; ++s) dpre[s] = h2[s] > 0.f ? a2[s >> 2][s & 3] : 0.f;
            }
            PH(3)       // layer 2 dgrad + wgrad + mask
            __builtin_amdgcn_sched_barrier(0);
            stage8(S1, lane_base, dpre);         // h2's tile is dead: its reads were issued above (LDS is in order within a wave)
            // ---- layer 1 ----
            {
                f32x4 a2[2];
                a2[0] = a2[1] = zero4();
                gemm32<L::OFF_WT1>(a2, wl, dpre, lane);
                wgrad16(dW1, db1, S1, S0, lane);
#pragma unroll
                for (int s = 0; s < 8; ++s) dpre[s] = h1[s] > 0.f ? a2[s >> 2][s & 3] : 0.f;
            }
            PH(4)       // layer 1
            __builtin_amdgcn_sched_barrier(0);
            stage8(S2, lane_base, dpre);         // dz's tile is dead
            __builtin_amdgcn_sched_barrier(0);
            // the dx chain starts from the gradient mlp3 left in d_in (mlp1 wave; zeros otherwise)
#pragma unroll
            for (int s = 0; s < 8; ++s) dx[s >> 2][s & 3] = oldr[s];
            // The next half's x and old d_in.  Both are issued HERE, in straight-line code and by both waves (the mlp2 wave's d_in loads are
            // out of range: no traffic, zeros): memory operations issued behind the spin loops below, or on one side of a branch, make the
            // compiler lose count of what is in flight, and the wait for x at the top of the loop becomes a full drain (stores included)
            load8(xa, va, lane_voff(va, q, np, ninb), ng * va.gs4);
            if (has_dx) load8(oldr, vdx, lane_voff(vdx, q, np, ninb && rmw && role == 0), ng * vdx.gs4);
            // ---- layer 0: the dx chain runs through both waves of the pair ----
            // (ONE call site per weight gradient: accumulators that are updated on both sides of a branch cost a second register set)
            if (role == 0 && has_dx) gemm32<L::OFF_WT0>(dx, wl, dpre, lane);
            wgrad16(dW0, db0, S2, XA, lane);
            PH(5)       // layer 0 weight gradient (+ mlp1's dgrad)
            if (role == 0) {
                if (has_dx) {
                    // slot hk & 1 still holds fragment hk - 2: wait until the mlp2 wave has read it
                    if (hk >= 2) {
#if !(FGNN_ABL & 1)
                        while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < hk - 1) __builtin_amdgcn_s_sleep(1);
#endif
                    }
                    PH(6)   // waiting for the partner wave
                    float *slot = XCH + (hk & 1) * 512;
#pragma unroll
                    for (int s = 0; s < 8; ++s) slot[s * 64 + lane] = dx[s >> 2][s & 3];
                    __hip_atomic_store(&flags[0], hk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                if (has_dx) {
#if !(FGNN_ABL & 1)
                    while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < hk + 1) __builtin_amdgcn_s_sleep(1);
#endif
                    PH(6)   // waiting for the partner wave
                    const float *slot = XCH + (hk & 1) * 512;
#pragma unroll
                    for (int s = 0; s < 8; ++s) dx[s >> 2][s & 3] = slot[s * 64 + lane];
                    __hip_atomic_store(&flags[1], hk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    gemm32<L::OFF_WT0>(dx, wl, dpre, lane);
                    float v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = dx[s >> 2][s & 3];
                    store8(v, vdx, dvoff, ds0);
                    if (emit) {
                        // GraphNorm-backward sums of the producer of the input slab: S1 = sum v, S2 = sum v (z_in - mean_in) over the valid pixels
                        if (!full) {
#pragma unroll
                            for (int s = 0; s < 8; ++s) v[s] = valid ? v[s] : 0.f;
                        }
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            es1[s] += v[s];
                            es2[s] = fmaf(v[s], u[s], es2[s]);
                        }
                    }
                }
            }
            ++hk;
        }
        PH(7)           // hand-over / mlp2's dgrad, store, emission
        if (emit && hf == 1) {
            // the pair's last tile of this graph: reduce the lane sums over the pixels (transposed through the dead h1 / dpre_1 tiles:
            // lane (ch, hh) owns 8 pixel columns of a channel); any other tile: an empty record
            float t1 = 0.f, t2 = 0.f;
            if (ntile >= T1 || ng != g) {
                stage8(S0, lane_base, es1);
                stage8(S1, lane_base, es2);
                const int ch = lane & 31, hh = lane >> 5;
                const float4 *vp = reinterpret_cast<const float4 *>(S0 + ch * TLD + 8 * hh);
                const float4 *up = reinterpret_cast<const float4 *>(S1 + ch * TLD + 8 * hh);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 a = vp[k], b = up[k];
                    t1 += (a.x + a.y) + (a.z + a.w);
                    t2 += (b.x + b.y) + (b.z + b.w);
                }
                t1 += __shfl_xor(t1, 32);
                t2 += __shfl_xor(t2, 32);
#pragma unroll
                for (int s = 0; s < 8; ++s) es1[s] = es2[s] = 0.f;
            }
            if (lane < 32) reinterpret_cast<float2 *>(P.m[1].s12part)[((long long)g * tpg + tt) * FGNN_H + row_chan(lane)] = make_float2(t1, t2);
        }
        tile = ntile;
        hf = nhf;
        PH(8)           // record store, loop bookkeeping
    }

    if constexpr (SKIP) {
        // padding-only tiles of this pair's share: empty S1/S2 records (their dx is not written: consumers skip the same tiles)
        if (emit) {
            for (int t = T0 + pair; t < T1; t += NP) {
                const int g = t / tpg, tt = t - g * tpg;
                if (tile_live(tt, A.N, A.nvalid[g])) continue;
                if (lane < 32) reinterpret_cast<float2 *>(P.m[1].s12part)[((long long)g * tpg + tt) * FGNN_H + lane] = make_float2(0.f, 0.f);
            }
        }
    }

    // ---- workgroup reduction: each MLP's partial = fixed-order sum of its four waves ----
    // layout per MLP: [W0 (32*32) | b0 (32) | W1 (1024) | b1 (32) | W2 (1024) | b2 (32)]
    constexpr int PCOUNT = L::PCOUNT;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        db0[b] += __shfl_xor(db0[b], 16);
        db0[b] += __shfl_xor(db0[b], 32);
        db1[b] += __shfl_xor(db1[b], 16);
        db1[b] += __shfl_xor(db1[b], 32);
        db2[b] += __shfl_xor(db2[b], 16);
        db2[b] += __shfl_xor(db2[b], 32);
    }
    __syncthreads();                       // everyone done with the operand images and the tile buffers
    PH(10)              // waiting for the slowest wave of the workgroup
    {
        float *red = smem + wv * PCOUNT;
        auto put = [&](int off, const f32x4 (&dW)[4], const float (&db)[2]) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[off + row_chan(16 * mb + 4 * q + r) * 32 + row_chan(16 * nb + px)] = dW[2 * mb + nb][r];
            if (q == 0) {
                red[off + 1024 + row_chan(px)] = db[0];
                red[off + 1024 + row_chan(16 + px)] = db[1];
            }
        };
        put(0, dW0, db0);
        put(1056, dW1, db1);
        put(2112, dW2, db2);
    }
    __syncthreads();
    static_assert(PCOUNT % 4 == 0 && PCOUNT == 3 * 1056, "partials are summed four at a time");
    const float4 *part4 = reinterpret_cast<const float4 *>(smem);
    for (int e = threadIdx.x; e < 2 * (PCOUNT / 4); e += 64 * NW) {
        const int m = e >= PCOUNT / 4 ? 1 : 0, ee = e - m * (PCOUNT / 4);
        float4 a = part4[(4 * m) * (PCOUNT / 4) + ee];
#pragma unroll
        for (int w = 1; w < NP; ++w) {                                  // fixed order over the MLP's four waves
            const float4 b = part4[(4 * m + w) * (PCOUNT / 4) + ee];
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        reinterpret_cast<float4 *>(P.m[m].wpart + (long long)blockIdx.x * PCOUNT)[ee] = a;
    }
    PH(11)              // workgroup reduction + partial store
    PH_FLUSH
}

template <bool SKIP, bool HAS_DX>
int launch_pair16(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, int tpg, int total, hipStream_t st) {
    constexpr int LDS = PairLayout16::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_pair_t16_kernel<SKIP, HAS_DX>, LDS);
    PairArgs P;
    P.m[0] = *a1;
    P.m[1] = *a2;
    hipLaunchKernelGGL((mlp_bwd_pair_t16_kernel<SKIP, HAS_DX>), dim3(a1->cu_share == 2 && !SKIP ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st, P, tpg,
                       total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#ifdef FGNN_PHASES
extern "C" int fgnn_debug_phase_buffer_t16(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_phase16_buf), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#endif

extern "C" int fgnn_mlp_bwd_pair_t16_supported(int ca, int depth) { return (depth == 3 && ca == 32) ? 1 : 0; }

// Same contract as fgnn_mlp_bwd_pair (see there); `packed` of both argument blocks must be images of kind 5 (fgnn_pack_operands).
extern "C" int fgnn_mlp_bwd_pair_t16(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, void *stream) {
    FGNN_CHECK(a1 && a2, "fgnn_mlp_bwd_pair_t16: null args");
    FGNN_CHECK(BWD_WG == fgnn_mlp_bwd_num_workgroups() && BWD_WG == FGNN_RANGE_WG, "fgnn_mlp_bwd_pair_t16: workgroup count differs from fgnn_mlp_bwd");
    FGNN_CHECK(a1->G > 0 && a1->N > 0 && a1->G == a2->G && a1->N == a2->N && a1->depth == a2->depth,
               "fgnn_mlp_bwd_pair_t16: the two MLPs must share G, N and depth");
    FGNN_CHECK(fgnn_mlp_bwd_pair_t16_supported(a1->a.C, a1->depth) && a1->b.C == 0 && a2->b.C == 0 && !a1->xbits && !a2->xbits,
               "fgnn_mlp_bwd_pair_t16: built for depth 3 and ONE dense input slab of 32 channels (got depth %d, %d + %d); use fgnn_mlp_bwd_pair",
               a1->depth, a1->a.C, a1->b.C);
    FGNN_CHECK(a1->a.ptr && a1->a.ptr == a2->a.ptr && a1->a.C == a2->a.C && a1->a.gstride == a2->a.gstride && a1->a.ldp == a2->a.ldp &&
               a1->a.nrm == a2->a.nrm && a1->a.beta == a2->a.beta && a1->nvalid == a2->nvalid,
               "fgnn_mlp_bwd_pair_t16: the two MLPs must read the same input slab");
    FGNN_CHECK(a1->ranges == a2->ranges && (!a1->ranges || a1->nvalid), "fgnn_mlp_bwd_pair_t16: both MLPs take the same ranges (with nvalid)");
    FGNN_CHECK(a1->packed && a2->packed, "fgnn_mlp_bwd_pair_t16: needs both operand images (fgnn_pack_operands, kind 5)");
    FGNN_CHECK(!a1->dxa && !a1->s12part, "fgnn_mlp_bwd_pair_t16: the input gradient and its tile sums belong to the SECOND argument block");
    FGNN_CHECK(!a1->s12tiles && !a2->s12tiles, "fgnn_mlp_bwd_pair_t16: s12tiles is an mlp3 feature");
    FGNN_CHECK(a1->N <= 256, "fgnn_mlp_bwd_pair_t16: N <= 256 (division-free pixel decode)");
    for (const fgnn_mlp_bwd_args *a : {a1, a2}) {
        FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd_pair_t16: missing dy/z/wpart");
        FGNN_CHECK(a->coef || (a->s12 && a->znrm), "fgnn_mlp_bwd_pair_t16: need coef, or s12 + znrm");
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim && G * a->dxa_gstride < lim,
                   "fgnn_mlp_bwd_pair_t16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a1->N);
    const long long total = (long long)a1->G * tpg;
    FGNN_CHECK(total < (1ll << 29), "fgnn_mlp_bwd_pair_t16: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (a2->dxa) return a1->ranges ? launch_pair16<true, true>(a1, a2, tpg, (int)total, st) : launch_pair16<false, true>(a1, a2, tpg, (int)total, st);
    return a1->ranges ? launch_pair16<true, false>(a1, a2, tpg, (int)total, st) : launch_pair16<false, false>(a1, a2, tpg, (int)total, st);
}
