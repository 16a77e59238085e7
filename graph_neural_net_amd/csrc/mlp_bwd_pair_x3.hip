// mlp1 + mlp2 of one FGNN block (models/blocks_emb.py:16-27: two MlpBlock_Real on the SAME input) backward in ONE launch,
// with every contraction on the bf16 matrix cores through the exact three-way operand split of fgnn_x3.h.
//
// The algorithm per MLP and per tile is mlp_bwd.hip's (autograd of models/layers.py:126-131 with the GraphNorm backward of
// :68-80 folded into the load of dz); the pairing is mlp_bwd_pair.hip's (the two waves of a SIMD take one MLP each on the
// same tile, the mlp1 wave hands its dx fragment over through LDS, the mlp2 wave stores (d_in3 + dx1) + dx2 once); the
// arithmetic is mlp_bwd_x3.hip's (fp32 tensors, operands split exactly into 3 bf16 parts, 8 partial products in the forward
// recompute -- bit-identical to mlp_fwd_x3.hip, so the ReLU masks are the forward's -- and 6 in the gradient GEMMs, fp32
// accumulation).  What is new here is how the weight-gradient GEMMs get their operands: they contract over the tile's 32
// pixels, so both operands are needed with lane = channel.  The fp32 kernels (and mlp_bwd_x3.hip) stage every operand in a
// wave-private 32 x 36 LDS tile and read it back transposed -- 24 tile slots = 110 KB, which next to two x3 operand images
// (62 KB) does not fit 160 KB.  Here an operand that is already split for its dgrad / forward GEMM (lane = pixel, slots =
// channels) is transposed PART BY PART ON THE MATRIX PIPE: two MFMAs against -I per bf16 part (exact), re-packed (exact).
// No tile slots, no LDS round trip in the dependency chain of a tile; LDS holds the two images (62 KB), the per-wave records
// (8 KB) and per pair one hand-over slot + two slots for the S1/S2 emission (55 KB).  Bias gradients fall out of the
// transposed parts (lane = channel: a register sum).
// Depth 3, input slab of 32 channels (blocks > 1) or 2 channels (block 1, dense or bit-packed; no input gradient there),
// constant-size batches.
#include "fgnn_tile.h"
#include "fgnn_pack.h"
#include "fgnn_x3.h"

namespace {

#ifdef FGNN_PHASES
// Debug build only (make phases): per-wave cycle stamps of the tile phases, summed over the wave's tiles (tools/gpu_phases_px3.py)
__device__ unsigned long long *g_px3_phase_buf = nullptr;
#define PH_DECL unsigned long long ph_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last_ = __builtin_amdgcn_s_memtime();
#define PH(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[k] += t_ - ph_last_; ph_last_ = t_; }
#define PH_FLUSH if (g_px3_phase_buf && CA == 32 && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; ++k_) g_px3_phase_buf[((long long)blockIdx.x * NW + wv) * 16 + k_] = ph_[k_]; }
#else
#define PH_DECL
#define PH(k)
#define PH_FLUSH
#endif

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

template <int CA>
struct PairX3Layout {
    static constexpr int DEPTH = 3;
    static constexpr PkX3 PK = pkx3_layout(1, CA, 0, DEPTH);              // one x3 image (kind 1) per MLP
    static constexpr int PD = PK.part_dw;
    static constexpr int OFF_W0A = PK.p.off_w0a, OFF_W1 = PK.p.off_wh;
    static constexpr int OFF_W2T = PK.p.off_wt, OFF_W1T = PK.p.off_wt + 2, OFF_WT0A = PK.p.off_wt0a;
    static constexpr int STEPS_A = pk16_steps(CA);
    static constexpr int BIAS_F = PK.bias_off;
    static constexpr int WEIGHT_F = PK.floats;
    static constexpr int REC_F = 2 * 32 * 4;                              // per wave: nrm a, coef
    static constexpr int PCOUNT = 32 * CA + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int PAIR_F = TILE_F;                                 // per pair: ONE slot -- dx hand-over, then the S1/S2 emission
    static constexpr int PARK_F = 2 * 16 * 64;                            // per wave: two weight-gradient accumulators [r][lane]
    static constexpr int REC_OFF = 2 * WEIGHT_F;
    static constexpr int PAIR_OFF = REC_OFF + NW * REC_F;
    static constexpr int PARK_OFF = PAIR_OFF + NP * PAIR_F;
    static constexpr int FLAG_OFF = PARK_OFF + NW * PARK_F;
    static constexpr int MAIN_F = FLAG_OFF + 4 * NP;
    static constexpr int RED_F = NW * PCOUNT;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

struct PairArgs {
    fgnn_mlp_bwd_args m[2];
};

// the three bf16 parts of an input slab as B operand (mlp_fwd_x3.hip)
template <int S>
DEVI void split_slab(X3 &x, const float (&v)[S > 0 ? S : 1], int h, const F16 &negI) {
    if constexpr (S == 16) {
        split16m(x, v, negI);
    } else if constexpr (S == 1) {
        const float other = __shfl_xor(v[0], 32);
        const float a = h == 0 ? v[0] : 0.f, b = h == 0 ? other : 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int d = 0; d < 8; ++d) x.p[q].d[d] = 0u;
        split_pair(a, b, x.p[0].d[0], x.p[1].d[0], x.p[2].d[0]);
    }
}

// rows ch_of(r, h) of a (G, 32, ld) tensor with a lane offset computed once per tile (all slabs of a launch share the channel stride)
DEVI void load_rows16_at(float (&x)[16], const View &v, int voff, int g) {
    const int s0 = g * v.gs4;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = buf_load(v, voff, s0 + ((r & 3) + 8 * (r >> 2)) * v.ld4);
}

// rows ch_of(r, h), r = 0..15, of a (G, 32, ld) tensor straight into LDS as [r][lane] (buffer_load ... lds: no VGPR holds the
// data while it is in flight); out-of-range lanes deposit 0.  `dst` is wave-uniform.
DEVI void load_rows16_to_lds(float *dst, const View &v, int voff, int g) {
    const int s0 = g * v.gs4;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v.r, (__attribute__((address_space(3))) void *)(unsigned long long)(unsigned)(unsigned long long)(dst + r * 64), 4,
                                                 voff, s0 + ((r & 3) + 8 * (r >> 2)) * v.ld4, 0, 0);
}

// acc (parked in LDS as [register][lane]: conflict-free, wave-private) += A (x) v over the tile's pixels (fgnn_x3.h)
DEVI void wgrad_parked(float *park, const X3 &a, const float (&v)[16], const F16 &negI, int lane) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = park[r * 64 + lane];
    t = wgrad_stream_x3(t, a, v, negI);
#pragma unroll
    for (int r = 0; r < 16; ++r) park[r * 64 + lane] = t[r];
}

template <int CA, bool PK>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_pair_x3_kernel(const PairArgs P, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = PairX3Layout<CA>;
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
    float *rec = smem + L::REC_OFF + wv * L::REC_F;
    float *recA = rec, *recK = rec + 128;
    float *XCH = smem + L::PAIR_OFF + pair * L::PAIR_F; // the mlp1 wave's dx fragment of the current tile; then the emission's transposer
    float *PARK = smem + L::PARK_OFF + wv * L::PARK_F;  // dW_1, dW_2 of this wave between their GEMMs
    int *flags = reinterpret_cast<int *>(smem + L::FLAG_OFF) + 4 * pair;     // [0] = tile whose dx is ready, [1] = tile consumed

    // Weight-gradient accumulators: dW_0 in registers, dW_1 / dW_2 parked in wave-private LDS between their GEMMs (the
    // matrix-pipe transposes need the registers: three resident 16-register accumulators push the tile loop into scratch)
    f32x16 dW0a;
    float ndb[DEPTH];                                   // NEGATED bias-gradient sums (transpose_x3)
    zero16f(dW0a);
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) ndb[l] = 0.f;

    const int nwg = gridDim.x;
    const int q = total_tiles / nwg, rem = total_tiles % nwg;
    const int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    const int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);
    const bool normA = A.a.nrm != nullptr;
    const bool has_dx = (CA == 32) && P.m[1].dxa != nullptr;
    const bool emit = (CA == 32) && role == 1 && normA && has_dx && P.m[1].s12part != nullptr;
    const bool rmw = has_dx && P.m[1].accumulate_a;

    PH_DECL
    // Prologue = ONE memory round trip: both operand images (into registers), the first tile, the per-graph records
    // (the two images are separate buffers: see mlp_bwd_pair.hip)
    constexpr int N4 = L::WEIGHT_F / 4, N4PAD = (N4 + 63) & ~63, IMG_PER = (N4PAD + N4 + 64 * NW - 1) / (64 * NW);
    static_assert(L::WEIGHT_F % 4 == 0, "images are copied 16 bytes at a time");
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
    const int first = T0 + pair;
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
#pragma unroll
    for (int r = 0; r < 32; ++r) PARK[r * 64 + lane] = 0.f;
    __syncthreads();

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const F16 negI = make_neg_identity(lane);
    PH(9)
    int tnext = 0, prev_tile = -1;
    if (role == 1) __builtin_amdgcn_s_setprio(1);      // static priority for the mlp2 waves: see mlp_bwd_pair_t16.hip
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NP;
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
        const int voff16 = lane_off<4>(vdy, c, h);       // lane offset into every 32-channel slab of this launch (equal channel strides)
        float dyr[16], zr[16];
        if (has_dx && role == 0) {
            // The pair's slot still holds this wave's previous fragment until the mlp2 wave has consumed it.  Then the gradient
            // mlp3 left in d_in is requested INTO the slot (LDS-direct: it waits there, not in registers, for the end of the tile)
            if (tile != first) {
                while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != prev_tile) __builtin_amdgcn_s_sleep(1);
            }
            if (rmw) {
                asm volatile("" ::: "memory");
                load_rows16_to_lds(XCH, vdxa, voff16, c.g);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);       // issued BEFORE this tile's dy / z loads (see the hand-over below)
            }
        }

        PH(0)               // records, consumed-wait, LDS-direct request
        // ---- forward recompute of the hidden activations: the arithmetic of mlp_fwd_x3.hip ----
        f32x16 acc;
        load_bias16(acc, wl + L::BIAS_F, 0, h);
        {
            float ya[SA > 0 ? SA : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            X3 X;
            split_slab<SA>(X, ya, h, negI);
            acc = gemm_x3<L::STEPS_A, X3_FWD_TERMS>(acc, wl, L::PD, L::OFF_W0A, X, lane);
        }
        PH(1)               // x arrival, normalise, split, layer-0 products issued
        // this tile's dy / z fly behind the recompute
        load_rows16_at(dyr, vdy, voff16, c.g);
        load_rows16_at(zr, vz, voff16, c.g);
        float h1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[r] = relu1(acc[r]);
        {
            X3 H;
            split16m(H, h1, negI);
            load_bias16(acc, wl + L::BIAS_F, 1, h);
            acc = gemm_x3<2, X3_FWD_TERMS>(acc, wl, L::PD, L::OFF_W1, H, lane);
        }
        float h2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[r] = relu1(acc[r]);
        __builtin_amdgcn_sched_barrier(0);

        PH(2)               // h1, h2
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
        PH(3)               // dz (waits for dy, z)
        // ---- layer 2: dgrad W_2^T dz, mask with h2, weight gradient dz (x) h2 from the matrix-pipe transposes ----
        // (the order below keeps at most three 24-register operands alive at a time; the barriers pin it)
        {
            X3 D;
            split16m(D, dpre, negI);
            acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_W2T, D, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = h2[r] > 0.f ? acc[r] : 0.f;
            __builtin_amdgcn_sched_barrier(0);
            X3 TD;
            transpose_x3<true>(TD, D, negI, ndb[2]);
            __builtin_amdgcn_sched_barrier(0);
            wgrad_parked(PARK + 1024, TD, h2, negI, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        PH(4)
        // ---- layer 1 ----
        {
            X3 D;
            split16m(D, dpre, negI);
            acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_W1T, D, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = h1[r] > 0.f ? acc[r] : 0.f;
            __builtin_amdgcn_sched_barrier(0);
            X3 TD;
            transpose_x3<true>(TD, D, negI, ndb[1]);
            __builtin_amdgcn_sched_barrier(0);
            wgrad_parked(PARK, TD, h1, negI, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        PH(5)
        // ---- layer 0: input gradient first (hand-over / store / emission), then the weight gradient ----
        {
            X3 D;
            split16m(D, dpre, negI);
            // next tile's input slab: requested here, a whole hand-over + weight-gradient phase ahead of its first use
            float nxa[SA > 0 ? SA : 1];
            {
                const TileCtx cn = decode_tile(tnext, tnext < T1, tpg, A.N, P2, j);
                load_slab<SA, PK>(nxa, va, ps, cn, h);
            }
            PH(6)           // split D0, next slab requested
            if constexpr (CA == 32) {
                if (has_dx) {
                    const f32x16 dx = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_WT0A, D, lane);
                    if (role == 0) {
                        // d_in3 + dx1 (the first of the two accumulating launches this replaces), handed over as [r][lane]
                        float v[16];
                        if (rmw) {
                            // The LDS-direct loads of the tile start have landed long ago: vector memory returns in order, and the
                            // dy / z loads requested AFTER them were waited for at dz.  (The compiler does not track LDS-direct
                            // loads -- it emits no wait before the ds_read below -- so the order of issue at the tile start is pinned
                            // there; this wait, which leaves only the slab prefetch above in flight, is a second line of defence.)
                            __builtin_amdgcn_s_waitcnt(0x0F70 | (SA & 15) | ((SA >> 4) << 14));
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] = XCH[r * 64 + lane] + dx[r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] = dx[r];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) XCH[r * 64 + lane] = v[r];
                        __hip_atomic_store(&flags[0], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != tile) __builtin_amdgcn_s_sleep(1);
                        // (d_in3 + dx1) + dx2: the association of the two accumulating launches this replaces
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = XCH[r * 64 + lane] + dx[r];
                        const int s0 = c.g * vdxa.gs4;
#pragma unroll
                        for (int r = 0; r < 16; ++r) buf_store(v[r], vdxa, voff16, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                        if (emit) {
                            // GraphNorm-backward sums of the producer of the input slab over this tile (mlp_bwd.hip): S1 = sum v,
                            // S2 = sum v (z_in - mean) per channel, two passes through the (consumed) hand-over slot as transposer
                            const float4 *tp = reinterpret_cast<const float4 *>(XCH + j * TLD + 16 * h);
#pragma unroll
                            for (int r = 0; r < 16; ++r) XCH[ch_of(r, h) * TLD + j] = c_valid ? v[r] : 0.f;
                            float4 vt[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) vt[k] = tp[k];
#pragma unroll
                            for (int r = 0; r < 16; ++r) XCH[ch_of(r, h) * TLD + j] = xa[r] - reinterpret_cast<const float4 *>(recA)[ch_of(r, h)].x;
                            float s1 = 0.f, s2 = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float4 a = vt[k], b = tp[k];
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
                        // the slot is free for the mlp1 wave's next fragment
                        __hip_atomic_store(&flags[1], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            PH(7)           // dx products, hand-over (flag waits), store, emission
            X3 TD;
            transpose_x3<true>(TD, D, negI, ndb[0]);
            __builtin_amdgcn_sched_barrier(0);
            {
                float ya[SA > 0 ? SA : 1];
                norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
                float y16[16];            // the slab in fragment order (a 2-channel slab: slots 0, 1 of half-wave 0, as split_slab lays it out)
                if constexpr (SA == 16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) y16[r] = ya[r];
                } else {
                    const float other = __shfl_xor(ya[0], 32);
#pragma unroll
                    for (int r = 0; r < 16; ++r) y16[r] = 0.f;
                    y16[0] = h == 0 ? ya[0] : 0.f;
                    y16[1] = h == 0 ? other : 0.f;
                }
                dW0a = wgrad_stream_x3(dW0a, TD, y16, negI);
            }
#pragma unroll
            for (int s_ = 0; s_ < (SA > 0 ? SA : 1); ++s_) xa[s_] = nxa[s_];
        }
        prev_tile = tile;
        PH(8)               // layer-0 weight gradient
    }

    // ---- workgroup reduction: each MLP's partial = fixed-order sum of its four waves ----
    // layout per MLP: [W0 (32*CA) | b0 (32) | W1 (1024) | b1 (32) | W2 (1024) | b2 (32)]
    constexpr int PCOUNT = L::PCOUNT;
    float db[DEPTH];
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) {
        db[l] = -ndb[l];
        db[l] += __shfl_xor(db[l], 32);
    }
    f32x16 dWh[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dWh[0][r] = PARK[r * 64 + lane];
        dWh[1][r] = PARK[1024 + r * 64 + lane];
    }
    PH(12)
    __syncthreads();                       // everyone done with the operand images, the pair slots and the parked accumulators
    PH(10)
    {
        float *red = smem + wv * PCOUNT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ch_of(r, h);
            if (j < CA) red[o * CA + j] = dW0a[r];
        }
        int off = 32 * CA;
#pragma unroll
        for (int l = 0; l < DEPTH; ++l) {
            if (l > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[off + ch_of(r, h) * 32 + j] = dWh[l - 1][r];
                off += 1024;
            }
            if (h == 0) red[off + j] = db[l];
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
    PH(11)
    PH_FLUSH
}

template <int CA, bool PK>
int launch_pair_x3(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, int tpg, int total, hipStream_t st) {
    constexpr int LDS = PairX3Layout<CA>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_pair_x3_kernel<CA, PK>, LDS);
    PairArgs P;
    P.m[0] = *a1;
    P.m[1] = *a2;
    hipLaunchKernelGGL((mlp_bwd_pair_x3_kernel<CA, PK>), dim3(a1->cu_share == 2 ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st, P, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#ifdef FGNN_PHASES
extern "C" int fgnn_debug_phase_buffer_px3(void *p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_px3_phase_buf), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int fgnn_mlp_bwd_pair_x3(const fgnn_mlp_bwd_args *a1, const fgnn_mlp_bwd_args *a2, void *stream) {
    FGNN_CHECK(a1 && a2, "fgnn_mlp_bwd_pair_x3: null args");
    FGNN_CHECK(BWD_WG == fgnn_mlp_bwd_num_workgroups(), "fgnn_mlp_bwd_pair_x3: workgroup count differs from fgnn_mlp_bwd");
    FGNN_CHECK(a1->G > 0 && a1->N > 0 && a1->G == a2->G && a1->N == a2->N && a1->depth == a2->depth,
               "fgnn_mlp_bwd_pair_x3: the two MLPs must share G, N and depth");
    FGNN_CHECK(fgnn_mlp_bwd_pair_supported(a1->a.C, a1->depth) && a1->b.C == 0 && a2->b.C == 0,
               "fgnn_mlp_bwd_pair_x3: built for depth 3 and ONE input slab of 2 or 32 channels (got depth %d, %d + %d); use fgnn_mlp_bwd",
               a1->depth, a1->a.C, a1->b.C);
    FGNN_CHECK(a1->a.ptr == a2->a.ptr && a1->a.C == a2->a.C && a1->a.gstride == a2->a.gstride && a1->a.ldp == a2->a.ldp &&
               a1->a.nrm == a2->a.nrm && a1->a.beta == a2->a.beta && a1->xbits == a2->xbits && a1->xdeg == a2->xdeg &&
               a1->nvalid == a2->nvalid, "fgnn_mlp_bwd_pair_x3: the two MLPs must read the same input slab");
    FGNN_CHECK(!a1->ranges && !a2->ranges, "fgnn_mlp_bwd_pair_x3: no padding-tile skipping (ranges); use fgnn_mlp_bwd_pair for ragged batches");
    FGNN_CHECK(a1->packed && a2->packed, "fgnn_mlp_bwd_pair_x3: needs both operand images (fgnn_pack_x3_operands, kind 1)");
    FGNN_CHECK(!a1->dxa && !a1->s12part, "fgnn_mlp_bwd_pair_x3: the input gradient and its tile sums belong to the SECOND argument block");
    FGNN_CHECK(!a1->s12tiles && !a2->s12tiles, "fgnn_mlp_bwd_pair_x3: s12tiles is an mlp3 feature");
    const bool pk_a = a1->xbits && a1->a.C == 2;
    FGNN_CHECK((a1->a.ptr || pk_a), "fgnn_mlp_bwd_pair_x3: slab a missing");
    FGNN_CHECK(!a1->xbits || a1->xdeg, "fgnn_mlp_bwd_pair_x3: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(!(a2->dxa && a2->a.C != 32), "fgnn_mlp_bwd_pair_x3: the input gradient exists for the 32-channel slab only; use fgnn_mlp_bwd");
    FGNN_CHECK(a1->ldd == a1->ldz && a2->ldd == a1->ldd && a2->ldz == a1->ldd && (!a2->dxa || a2->dxa_ld == a1->ldd) &&
               (a1->a.C != 32 || a1->a.ldp == a1->ldd),
               "fgnn_mlp_bwd_pair_x3: dy, z, d_in (and a 32-channel input slab) must share one channel stride");
    for (const fgnn_mlp_bwd_args *a : {a1, a2}) {
        FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd_pair_x3: missing dy/z/wpart");
        FGNN_CHECK(a->coef || (a->s12 && a->znrm), "fgnn_mlp_bwd_pair_x3: need coef, or s12 + znrm");
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim && G * a->dxa_gstride < lim,
                   "fgnn_mlp_bwd_pair_x3: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a1->N);
    const long long total = (long long)a1->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd_pair_x3: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (a1->xbits) return launch_pair_x3<2, true>(a1, a2, tpg, (int)total, st);
    if (a1->a.C == 2) return launch_pair_x3<2, false>(a1, a2, tpg, (int)total, st);
    return launch_pair_x3<32, false>(a1, a2, tpg, (int)total, st);
}
