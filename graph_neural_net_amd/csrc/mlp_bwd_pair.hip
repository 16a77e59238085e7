// mlp1 + mlp2 of one FGNN block (models/blocks_emb.py:16-27: two MlpBlock_Real on the SAME input) backward in ONE launch.
//
// The algorithm per MLP and per tile is mlp_bwd.hip's (recompute, dz, three weight gradients, dgrad chain: see there); what
// this kernel changes is who does it and what travels through HBM:
//   * the two waves that share a SIMD form a PAIR working on the same tile: wave p (p = 0..3) runs mlp1, wave p + 4 runs
//     mlp2.  Each holds only its own MLP's weight-gradient accumulators, so the register budget is the single-MLP kernel's;
//   * the gradient of the shared input is summed in the pair instead of in memory: the mlp1 wave leaves its 32 x 32 dx
//     fragment in one of its (by then dead) LDS tile slots, the mlp2 wave adds it to its own and to the gradient mlp3 left
//     in HBM -- d_in = (d_in3 + dx1) + dx2, the association of the two read-modify-write launches it replaces, so d_in is
//     bit-identical to theirs -- and stores once.  Two LDS words per pair order the hand-over (release / acquire at workgroup
//     scope; the mlp1 wave waits for "consumed" only before it overwrites that slot one tile later, so neither wave waits in
//     the common case).  Per block: 7 slab passes instead of 10 (x read once by the CU, no re-read / re-write of d_in between
//     the two MLPs), one prologue / tail instead of two, and 4.94 tiles per pair (five rounds) instead of 2.47 per wave
//     (three rounds of which the third is half empty).
// Operand images, tile statistics emission (S1/S2 of the previous block's mlp3) and the partial layout are unchanged:
// each MLP's partial is the fixed-order sum of its four waves, so results are bit-reproducible run to run.
// Depth 3, input slab of 32 channels (blocks > 1) or 2 channels (block 1, dense or bit-packed; no input gradient there, the
// pair only shares the launch); constant-size and ragged batches (nvalid, optionally with the padding-tile skipping of
// fgnn_ragged_tile_ranges).
#include "fgnn_tile.h"
#include "fgnn_pack.h"

namespace {

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
DEVI float4 coef_record(const fgnn_mlp_bwd_args &A, int g, int ch) {
    if (A.coef) return reinterpret_cast<const float4 *>(A.coef)[(long long)g * FGNN_H + ch];
    const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + ch];
    const float2 sv = reinterpret_cast<const float2 *>(A.s12)[(long long)g * FGNN_H + ch];
    return coef_from_sums(n, sv, (float)nvalid_of(A.nvalid, g, A.N));
}

// dW += Dt (rows = out channel) x In (rows = in channel), contraction over the 32 pixels; db from the same LDS reads
template <bool WITH_DB>
DEVI f32x16 wgrad_tile(const float *Dt, const float *In, f32x16 acc, float &db, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float4 *dp = reinterpret_cast<const float4 *>(Dt + i * TLD + 4 * h);
    const float4 *ip = reinterpret_cast<const float4 *>(In + i * TLD + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = dp[2 * q];
        const float4 b = ip[2 * q];
        if (WITH_DB) db += (a.x + a.y) + (a.z + a.w);
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
    }
    return acc;
}

DEVI void zero16(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

template <int OFF, int CNT>
DEVI void load_ops(float (&dst)[CNT > 0 ? CNT : 1], const float *wl, int lane) {
    static_assert(OFF % 4 == 0, "operand sets are float4 aligned");
    const float4 *p = reinterpret_cast<const float4 *>(wl) + (OFF / 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < (CNT + 3) / 4; ++q) {
        const float4 v = p[q * 64];
        if (4 * q + 0 < CNT) dst[4 * q + 0] = v.x;
        if (4 * q + 1 < CNT) dst[4 * q + 1] = v.y;
        if (4 * q + 2 < CNT) dst[4 * q + 2] = v.z;
        if (4 * q + 3 < CNT) dst[4 * q + 3] = v.w;
    }
}
DEVI void load_bias(float (&dst)[16], const float *tail, int layer, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(tail + layer * 32 + h * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        dst[4 * q + 0] = v.x;
        dst[4 * q + 1] = v.y;
        dst[4 * q + 2] = v.z;
        dst[4 * q + 3] = v.w;
    }
}

template <int CA>
struct PairLayout {
    static constexpr int DEPTH = 3;
    static constexpr PkBwd PK = pk_bwd(CA, 0, DEPTH);                     // one image per MLP: fgnn_pack.h
    static constexpr int OFF_W1A = PK.off_w1a, OFF_WH = PK.off_wh, BIAS_F = PK.bias_f, OFF_WT = PK.off_wt, OFF_WT0A = PK.off_wt0a;
    static constexpr int WEIGHT_F = PK.floats;                            // floats per image
    static constexpr int REC_F = 2 * 32 * 4;                              // per wave: nrm a, coef
    static constexpr int NSLOT = 3;
    static constexpr int PCOUNT = 32 * CA + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int FLAG_F = 4 * NP;                                 // per pair: ready, consumed (+ padding)
    static constexpr int TILE_OFF = 2 * WEIGHT_F + NW * REC_F;
    static constexpr int FLAG_OFF = TILE_OFF + NW * NSLOT * TILE_F;
    static constexpr int MAIN_F = FLAG_OFF + FLAG_F;
    static constexpr int RED_F = NW * PCOUNT;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

struct PairArgs {
    fgnn_mlp_bwd_args m[2];
};

// SKIP (ragged batches with ranges): work-balanced tile range from fgnn_ragged_tile_ranges, padding-only tiles are stepped over
// (the two waves of a pair walk the same tile sequence, so the hand-over protocol is unchanged)
template <int CA, bool PK, bool SKIP>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_pair_kernel(const PairArgs P, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = PairLayout<CA>;
    constexpr int DEPTH = 3, SA = CA / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv >> 2, pair = wv & 3;            // role 0: mlp1 (hands its dx over), role 1: mlp2 (sums, stores, emits)
    const int j = lane & 31, h = lane >> 5;
    const fgnn_mlp_bwd_args &A = P.m[role];
    const int P2 = A.N * A.N;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PK) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    const View vdy = make_view(A.dy, A.dgstride, A.ldd, A.G);
    const View vz = make_view(A.z, A.zgstride, A.ldz, A.G);
    const View vdxa = make_view(P.m[1].dxa, P.m[1].dxa_gstride, P.m[1].dxa_ld, P.m[1].G);

    float *wl = smem + role * L::WEIGHT_F;              // this wave's MLP image
    float *rec = smem + 2 * L::WEIGHT_F + wv * L::REC_F;
    float *recA = rec, *recK = rec + 128;
    float *tiles = smem + L::TILE_OFF;
    float *my = tiles + wv * (L::NSLOT * TILE_F);
    float *S0 = my, *S1 = my + TILE_F, *S2 = my + 2 * TILE_F;
    float *XA = S0;
    // hand-over slot: the mlp1 wave's S1 (dpre_1 is dead after the layer-1 weight gradient; the next tile re-uses it for h2)
    float *XCH = tiles + pair * (L::NSLOT * TILE_F) + TILE_F;
    int *flags = reinterpret_cast<int *>(smem + L::FLAG_OFF) + 4 * pair;     // [0] = tile whose dx is ready, [1] = tile consumed

    constexpr bool VW0 = (CA == 2);                     // 2-channel input: layer-0 weight gradient on the VALU, no dx
    float w0v[VW0 ? 32 : 1], b0v[VW0 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (VW0 ? 32 : 1); ++r) w0v[r] = 0.f;
#pragma unroll
    for (int r = 0; r < (VW0 ? 16 : 1); ++r) b0v[r] = 0.f;
    f32x16 dW0a, dWh[2];
    float db[DEPTH];
    zero16(dW0a);
    zero16(dWh[0]);
    zero16(dWh[1]);
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] = 0.f;

    const int nwg = gridDim.x;
    const int q = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const bool normA = A.a.nrm != nullptr;
    const bool has_dx = (CA == 32) && P.m[1].dxa != nullptr;
    const bool emit = (CA == 32) && role == 1 && normA && has_dx && P.m[1].s12part != nullptr;
    const bool rmw = has_dx && role == 1 && P.m[1].accumulate_a;

    // Prologue = ONE memory round trip: both operand images (into registers), the first tile, the per-graph records.
    // The two images are separate buffers: element e of the combined index space [0, N4PAD + N4) belongs to image 0 below N4
    // and to image 1 from N4PAD on (N4PAD = N4 rounded up to a wave, so the choice of the descriptor is wave-uniform).
    constexpr int N4 = L::WEIGHT_F / 4, N4PAD = (N4 + 63) & ~63, IMG_PER = (N4PAD + N4 + 64 * NW - 1) / (64 * NW);
    float4 img[IMG_PER];
    {
        const rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.m[0].packed), 0, N4 * 16, 0x00020000);
        const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.m[1].packed), 0, N4 * 16, 0x00020000);
#pragma unroll
        for (int k = 0; k < IMG_PER; ++k) {
            const int e = threadIdx.x + k * (64 * NW);
            const bool second = __builtin_amdgcn_readfirstlane(e) >= N4PAD;
            const rsrc_t rs = second ? r1 : r0;
            const int off = (second ? e - N4PAD : e) * 16;                                    // past the end: returns 0
            img[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            img[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 4, 0));
            img[k].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 8, 0));
            img[k].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 12, 0));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float xa[SA > 0 ? SA : 1];
    float4 rk = make_float4(0.f, 0.f, 0.f, 0.f), ra = rk;
    int cached_g = -1, cur_nv = A.N;
    int first = T0 + pair;
    if constexpr (SKIP) first = __builtin_amdgcn_readfirstlane(next_live_tile(first, T1, NP, tpg, A.N, A.nvalid));
    {
        const int t = first;
        const TileCtx c = decode_tile(t, t < T1, tpg, A.N, P2, j);
        load_slab<SA, PK>(xa, va, ps, c, h);
        if (t < T1 && lane < 32) {
            rk = coef_record(A, c.g, lane);
            if (normA && lane < CA) {
                ra = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                ra.z = A.a.beta ? A.a.beta[lane] : 0.f;
            }
        }
        if (t < T1) {
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
    }
#pragma unroll
    for (int k = 0; k < IMG_PER; ++k) {
        const int e = threadIdx.x + k * (64 * NW);
        if (e < N4) reinterpret_cast<float4 *>(smem)[e] = img[k];
        else if (e >= N4PAD && e < N4PAD + N4) reinterpret_cast<float4 *>(smem)[N4 + e - N4PAD] = img[k];
    }
    if (lane < 32) {
        reinterpret_cast<float4 *>(recK)[lane] = rk;
        reinterpret_cast<float4 *>(recA)[lane] = ra;
    }
    if (threadIdx.x < 4 * NP) reinterpret_cast<int *>(smem + L::FLAG_OFF)[threadIdx.x] = -1;
    __syncthreads();

#ifndef FGNN_PRIO32
#define FGNN_PRIO32 1
#endif
    // static priority for the mlp2 waves (the younger half of the workgroup AND the longer half of the pair): see mlp_bwd_pair_t16.hip
    if (FGNN_PRIO32 > 0 && role == 1) __builtin_amdgcn_s_setprio(FGNN_PRIO32);
    int tnext = 0, prev_tile = -1;
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NP;
        if constexpr (SKIP) tnext = __builtin_amdgcn_readfirstlane(next_live_tile(tnext, T1, NP, tpg, A.N, A.nvalid));
        const TileCtx c = decode_tile(tile, true, tpg, A.N, P2, j);
        if (c.g != cached_g) {
            if (lane < 32) {
                reinterpret_cast<float4 *>(recK)[lane] = coef_record(A, c.g, lane);
                if (normA && lane < CA) {
                    float4 n = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                    n.z = A.a.beta ? A.a.beta[lane] : 0.f;
                    reinterpret_cast<float4 *>(recA)[lane] = n;
                }
            }
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
        const bool c_valid = tile_valid(c, cur_nv);
        float dyr[16], zr[16], old[16];

        // ---- forward recompute of the hidden activations (S0 = h1, S1 = h2) ----
        f32x16 acc;
        {
            float ya[SA > 0 ? SA : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            float b0[16];
            load_bias(b0, wl + L::BIAS_F, 0, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b0[r];
            float w1a[SA > 0 ? SA : 1];
            load_ops<L::OFF_W1A, SA>(w1a, wl, lane);
#pragma unroll
            for (int s = 0; s < SA; ++s) acc = mfma32(w1a[s], ya[s], acc);
        }
        load_rows16(dyr, vdy, c, h);
        load_rows16(zr, vz, c, h);
        if (rmw) load_rows16(old, vdxa, c, h);
        {
            float hid[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hid[r] = relu1(acc[r]);
                S0[ch_of(r, h) * TLD + j] = hid[r];
            }
            float bl[16], wf[16];
            load_bias(bl, wl + L::BIAS_F, 1, h);
            load_ops<L::OFF_WH, 16>(wf, wl, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bl[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = mfma32(wf[r], hid[r], acc);
        }
        if (has_dx && role == 0 && tile != first) {
            // S1 still holds the dx handed over one tile ago: wait until the mlp2 wave has read it
            while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != prev_tile) __builtin_amdgcn_s_sleep(1);
        }
        prev_tile = tile;
#pragma unroll
        for (int r = 0; r < 16; ++r) S1[ch_of(r, h) * TLD + j] = relu1(acc[r]);

        // ---- dz from (dy, z, coef) ----
        float dpre[16];
        {
            const float4 *kp = reinterpret_cast<const float4 *>(recK) + 4 * h;
            const float vf = c_valid ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 k = kp[(r & 3) + 8 * (r >> 2)];
                dpre[r] = (k.y * dyr[r] + k.z * (zr[r] - k.x) + k.w) * vf;
            }
        }
        // ---- hidden layers: l = 2 (D = S2, In = S1), l = 1 (D = S1, In = S0) ----
#pragma unroll
        for (int l = DEPTH - 1; l >= 1; --l) {
            float *Dt = (l == DEPTH - 1) ? S2 : S1;
            const float *In = (l == 2) ? S1 : S0;
#pragma unroll
            for (int r = 0; r < 16; ++r) Dt[ch_of(r, h) * TLD + j] = dpre[r];
            float wt[16], hsv[16];
            if (l == 1) load_ops<L::OFF_WT, 16>(wt, wl, lane);
            if (l == 2) load_ops<L::OFF_WT + 16, 16>(wt, wl, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) hsv[r] = In[ch_of(r, h) * TLD + j];
            f32x16 a2;
            zero16(a2);
#pragma unroll
            for (int r = 0; r < 16; ++r) a2 = mfma32(wt[r], dpre[r], a2);
            dWh[l - 1] = wgrad_tile<true>(Dt, In, dWh[l - 1], db[l], lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = hsv[r] > 0.f ? a2[r] : 0.f;
        }
        // ---- layer 0 (D = S2, x_a re-staged into S0) ----
        if constexpr (VW0) {
            float ya[1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            const float other = __shfl_xor(ya[0], 32);
            const float x0 = h ? other : ya[0], x1 = h ? ya[0] : other;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                w0v[2 * r] = fmaf(dpre[r], x0, w0v[2 * r]);
                w0v[2 * r + 1] = fmaf(dpre[r], x1, w0v[2 * r + 1]);
                b0v[r] += dpre[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) S2[ch_of(r, h) * TLD + j] = dpre[r];
            float ya[SA > 0 ? SA : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
#pragma unroll
            for (int s = 0; s < SA; ++s) XA[slab_ch<SA>(s, h) * TLD + j] = ya[s];
        }
        // next tile's input slab (the emitting wave still needs xa for z - mean)
        float nxa[CA == 32 ? 16 : 1];
        {
            const TileCtx cn = decode_tile(tnext, tnext < T1, tpg, A.N, P2, j);
            if constexpr (CA == 32) load_raw<SA>(nxa, va, cn, h);
            else load_slab<SA, PK>(xa, va, ps, cn, h);
        }
        if constexpr (CA == 32) {
            f32x16 dx;
            zero16(dx);
            if (has_dx) {
                float wt[16];
                load_ops<L::OFF_WT0A, 16>(wt, wl, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) dx = mfma32(wt[r], dpre[r], dx);
            }
            dW0a = wgrad_tile<true>(S2, XA, dW0a, db[0], lane);
            if (has_dx) {
                if (role == 0) {
                    // hand the fragment over: same [channel][pixel] staging as every tile, into the dead dpre_1 slot
#pragma unroll
                    for (int r = 0; r < 16; ++r) S1[ch_of(r, h) * TLD + j] = dx[r];
                    __hip_atomic_store(&flags[0], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != tile) __builtin_amdgcn_s_sleep(1);
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = XCH[ch_of(r, h) * TLD + j];
                    __hip_atomic_store(&flags[1], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // (d_in3 + dx1) + dx2: the association of the two accumulating launches this replaces
                    if (rmw) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = old[r] + v[r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += dx[r];
                    const int voff = lane_off<4>(vdxa, c, h);
                    const int s0 = c.g * vdxa.gs4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_store(v[r], vdxa, voff, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                    if (emit) {
                        // GraphNorm-backward sums of the producer of the input slab over this tile (mlp_bwd.hip)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ch = ch_of(r, h);
                            const float mean = reinterpret_cast<const float4 *>(recA)[ch].x;
                            S1[ch * TLD + j] = c_valid ? v[r] : 0.f;
                            S2[ch * TLD + j] = xa[r] - mean;
                        }
                        const float4 *vp = reinterpret_cast<const float4 *>(S1 + j * TLD + 16 * h);
                        const float4 *up = reinterpret_cast<const float4 *>(S2 + j * TLD + 16 * h);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float4 a = vp[k], b = up[k];
                            s1 += (a.x + a.y) + (a.z + a.w);
                            s2 += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
                        }
                        s1 += __shfl_xor(s1, 32);
                        s2 += __shfl_xor(s2, 32);
                        if (h == 0) {
                            float2 o;
                            o.x = s1;
                            o.y = s2;
                            reinterpret_cast<float2 *>(P.m[1].s12part)[((long long)c.g * tpg + c.tt) * FGNN_H + j] = o;
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < SA; ++s) xa[s] = nxa[s];
        }
    }

    if constexpr (SKIP) {
        // padding-only tiles of this pair's share: empty S1/S2 records (their dx is not written: consumers skip the same tiles)
        if (emit) {
            for (int t = T0 + pair; t < T1; t += NP) {
                const TileCtx c = decode_tile(t, true, tpg, A.N, P2, j);
                if (tile_live(c.tt, A.N, A.nvalid[c.g])) continue;
                if (h == 0)
                    reinterpret_cast<float2 *>(P.m[1].s12part)[((long long)c.g * tpg + c.tt) * FGNN_H + j] = make_float2(0.f, 0.f);
            }
        }
    }

    // ---- workgroup reduction: each MLP's partial = fixed-order sum of its four waves ----
    // layout per MLP: [W0 (32*CA) | b0 (32) | W1 (1024) | b1 (32) | W2 (1024) | b2 (32)]
    constexpr int PCOUNT = L::PCOUNT;
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] += __shfl_xor(db[l], 32);
    if constexpr (VW0) {
#pragma unroll
        for (int r = 0; r < 32; ++r) w0v[r] = half_sum(w0v[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) b0v[r] = half_sum(b0v[r]);
    }
    __syncthreads();                       // everyone done with the operand images and the tile buffers
    {
        float *red = smem + wv * PCOUNT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ch_of(r, h);
            if constexpr (VW0) {
                if (j == 0) {
                    red[o * CA] = w0v[2 * r];
                    red[o * CA + 1] = w0v[2 * r + 1];
                    red[32 * CA + o] = b0v[r];
                }
            } else {
                if (j < CA) red[o * CA + j] = dW0a[r];
            }
        }
        int off = 32 * CA;
#pragma unroll
        for (int l = 0; l < DEPTH; ++l) {
            if (l > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[off + ch_of(r, h) * 32 + j] = dWh[l - 1][r];
                off += 1024;
            }
            if (h == 0 && !(VW0 && l == 0)) red[off + j] = db[l];
            off += 32;
        }
    }
    __syncthreads();
    static_assert(PCOUNT % 4 == 0, "partials are summed four at a time");
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
}

template <int CA, bool PK, bool SKIP>
int launch_pair_impl(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, int tpg, int total, hipStream_t st) {
    constexpr int LDS = PairLayout<CA>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_pair_kernel<CA, PK, SKIP>, LDS);
    PairArgs P;
    P.m[0] = *a1;
    P.m[1] = *a2;
    hipLaunchKernelGGL((mlp_bwd_pair_kernel<CA, PK, SKIP>), dim3(a1->cu_share == 2 && !SKIP ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st, P,
                       tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, bool PK>
int launch_pair(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, int tpg, int total, hipStream_t st) {
    static_assert(BWD_WG == FGNN_RANGE_WG, "fgnn_ragged_tile_ranges splits for the backward grid");
    if (a1->ranges) return launch_pair_impl<CA, PK, true>(a1, a2, tpg, total, st);
    return launch_pair_impl<CA, PK, false>(a1, a2, tpg, total, st);
}

}  // namespace

extern "C" int fgnn_mlp_bwd_pair_supported(int ca, int depth) { return (depth == 3 && (ca == 2 || ca == 32)) ? 1 : 0; }

extern "C" int fgnn_mlp_bwd_pair(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, void *stream) {
    FGNN_CHECK(a1 && a2, "fgnn_mlp_bwd_pair: null args");
    FGNN_CHECK(BWD_WG == fgnn_mlp_bwd_num_workgroups(), "fgnn_mlp_bwd_pair: workgroup count differs from fgnn_mlp_bwd");
    FGNN_CHECK(a1->G > 0 && a1->N > 0 && a1->G == a2->G && a1->N == a2->N && a1->depth == a2->depth,
               "fgnn_mlp_bwd_pair: the two MLPs must share G, N and depth");
    FGNN_CHECK(fgnn_mlp_bwd_pair_supported(a1->a.C, a1->depth) && a1->b.C == 0 && a2->b.C == 0,
               "fgnn_mlp_bwd_pair: built for depth 3 and ONE input slab of 2 or 32 channels (got depth %d, %d + %d); use fgnn_mlp_bwd",
               a1->depth, a1->a.C, a1->b.C);
    FGNN_CHECK(a1->a.ptr == a2->a.ptr && a1->a.C == a2->a.C && a1->a.gstride == a2->a.gstride && a1->a.ldp == a2->a.ldp &&
               a1->a.nrm == a2->a.nrm && a1->a.beta == a2->a.beta && a1->xbits == a2->xbits && a1->xdeg == a2->xdeg &&
               a1->nvalid == a2->nvalid, "fgnn_mlp_bwd_pair: the two MLPs must read the same input slab");
    FGNN_CHECK(a1->ranges == a2->ranges && (!a1->ranges || a1->nvalid), "fgnn_mlp_bwd_pair: both MLPs take the same ranges (with nvalid)");
    FGNN_CHECK(a1->packed && a2->packed, "fgnn_mlp_bwd_pair: needs both operand images (fgnn_pack_operands, kind 1)");
    FGNN_CHECK(!a1->dxa && !a1->s12part, "fgnn_mlp_bwd_pair: the input gradient and its tile sums belong to the SECOND argument block");
    FGNN_CHECK(!a1->s12tiles && !a2->s12tiles, "fgnn_mlp_bwd_pair: s12tiles is an mlp3 feature");
    const bool pk_a = a1->xbits && a1->a.C == 2;
    FGNN_CHECK((a1->a.ptr || pk_a), "fgnn_mlp_bwd_pair: slab a missing");
    FGNN_CHECK(!a1->xbits || a1->xdeg, "fgnn_mlp_bwd_pair: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(!(a2->dxa && a2->a.C != 32), "fgnn_mlp_bwd_pair: the input gradient exists for the 32-channel slab only; use fgnn_mlp_bwd");
    for (const fgnn_mlp_bwd_args *a : {a1, a2}) {
        FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd_pair: missing dy/z/wpart");
        FGNN_CHECK(a->coef || (a->s12 && a->znrm), "fgnn_mlp_bwd_pair: need coef, or s12 + znrm");
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim && G * a->dxa_gstride < lim,
                   "fgnn_mlp_bwd_pair: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a1->N);
    const long long total = (long long)a1->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd_pair: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (a1->xbits) return launch_pair<2, true>(a1, a2, tpg, (int)total, st);
    if (a1->a.C == 2) return launch_pair<2, false>(a1, a2, tpg, (int)total, st);
    return launch_pair<32, false>(a1, a2, tpg, (int)total, st);
}
