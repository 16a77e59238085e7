// Fused MlpBlock_Real backward for gfx950 (autograd of models/layers.py:126-131 plus the
// GraphNorm backward of :68-80 folded into the load of dz).
//
// Per 32-pixel tile (one wave, everything in registers / wave-private LDS):
//   1. load the input slabs (normalising on load), recompute the hidden activations with
//      the same in-register MFMA chain as the forward kernel;
//   2. dz = ca*dy + cb*(z-mean) + cc              (coef from fgnn_gn_bwd_coef);
//   3. for l = depth-1 .. 0:
//        dW_l += dpre_l (x) in_l       MFMA over pixels; both operands are transposed
//                                      through a 32x36 LDS tile (conflict-free b128 reads)
//        db_l += dpre_l
//        d in_l = W_l^T dpre_l          in-register chain again (D fragment -> B operand)
//        dpre_{l-1} = d in_l * [h_{l-1} > 0]
//   4. dx written (or accumulated) per input slab; optionally the per-tile sums
//      S1 = sum dx, S2 = sum dx*(z_in - mean_in) that the GraphNorm backward of the MLP
//      which PRODUCED slab a needs (saves a separate reduction pass over dx and z).
// dW/db accumulate in registers over the wave's (statically assigned) tiles; at the end
// the eight waves of a workgroup are summed through LDS in a fixed order and one partial
// per workgroup is written, to be reduced in fixed order by fgnn_grad_finalize: results
// are bit-reproducible run to run.
// Execution shape: one 512-thread workgroup per CU (2 waves per SIMD, <= 256 VGPRs), all
// MFMA A-operands in workgroup-shared LDS, three aliased 32x36 LDS tiles per wave.
#include "fgnn_common.h"
#include "fgnn_pack.h"

#ifdef FGNN_PHASES
// Debug build only (make phases): per-wave cycle stamps of the tile phases, summed over the wave's
// tiles and written to a buffer set by fgnn_debug_phase_buffer().  Not part of the shipped library.
__device__ unsigned long long *g_phase_buf = nullptr;
__device__ int g_phase_sel = 0;       // CA * 100 + CB of the variant that records
#define PH_DECL unsigned long long ph_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last_ = __builtin_amdgcn_s_memtime();
#define PH(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[k] += t_ - ph_last_; ph_last_ = t_; }
#define PH_FLUSH if (g_phase_buf && g_phase_sel == CA * 100 + CB && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; ++k_) g_phase_buf[((long long)blockIdx.x * NW + wv) * 16 + k_] = ph_[k_]; }
#else
#define PH_DECL
#define PH(k) __builtin_amdgcn_sched_barrier(0);        // phase boundaries double as scheduling barriers: <32,32,3> 256 VGPRs + 2 spilled -> 251, none spilled (same time)
#define PH_FLUSH
#endif

namespace {

constexpr int TLD = 36;              // LDS tile row stride (floats): 144 B rows, 16-B aligned
constexpr int TILE_F = 32 * TLD;     // floats per 32-row tile
constexpr int BWD_WG = 256;          // persistent workgroups (one per CU)

struct TileCtx {
    int g, tt, p, i, jj;
    bool inb;
};

// The valid-vertex count of the graph is NOT read here: a global load at the top of every tile makes
// the compiler drain the whole memory pipeline (s_waitcnt vmcnt(0): next-tile prefetch AND the previous
// tile's stores).  It is fetched once per graph change, next to the per-graph records.
DEVI TileCtx decode_tile(int tile, bool active, int tpg, int N, int P, int j) {
    TileCtx c;
    c.g = __builtin_amdgcn_readfirstlane(active ? tile / tpg : 0);
    c.tt = active ? tile - c.g * tpg : 0;
    c.p = c.tt * FGNN_TILE + j;
    c.inb = active && c.p < P;
    c.i = c.p / N;
    c.jj = c.p - c.i * N;
    return c;
}
DEVI bool tile_valid(const TileCtx &c, int nv) { return c.inb && c.i < nv && c.jj < nv; }

// channel contracted by k-step k in half-wave h (same convention as mlp_fwd.hip)
template <int S>
DEVI constexpr int slab_ch(int k, int h) { return S == 16 ? ch_of(k, h) : 2 * k + h; }

// h-independent part of slab_ch, and the row multiplier of the h part
template <int S>
DEVI constexpr int slab_kbase(int k) { return S == 16 ? (k & 3) + 8 * (k >> 2) : 2 * k; }
template <int S>
DEVI constexpr int slab_hmul() { return S == 16 ? 4 : 1; }

// per-lane byte offset of pixel c.p in the half-wave's first row (OOB_OFF when out of range)
template <int HMUL>
DEVI int lane_off(const View &v, const TileCtx &c, int h) {
    return c.inb ? HMUL * h * v.ld4 + 4 * c.p : OOB_OFF;
}

template <int S>
DEVI void load_raw(float (&x)[S > 0 ? S : 1], const View &v, const TileCtx &c, int h) {
    if constexpr (S > 0) {
        const int voff = lane_off<slab_hmul<S>()>(v, c, h);
        const int s0 = c.g * v.gs4;
#pragma unroll
        for (int k = 0; k < S; ++k) x[k] = buf_load(v, voff, s0 + slab_kbase<S>(k) * v.ld4);
    }
}

// slab load: from memory, or (PK, 2-channel slabs only) from the packed adjacency
template <int S, bool PK>
DEVI void load_slab(float (&x)[S > 0 ? S : 1], const View &v, const PackedSrc &ps, const TileCtx &c, int h) {
    if constexpr (PK && S == 1) load_packed(x, ps, c, h);
    else load_raw<S>(x, v, c, h);
}

// rows ch_of(r,h) of a (G,32,ld) tensor
DEVI void load_rows16(float (&x)[16], const View &v, const TileCtx &c, int h) {
    const int voff = lane_off<4>(v, c, h);
    const int s0 = c.g * v.gs4;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = buf_load(v, voff, s0 + ((r & 3) + 8 * (r >> 2)) * v.ld4);
}

// y = (x - mean) * a + beta with the per-graph records {mean, a, beta, -} read from LDS
template <int S>
DEVI void norm_from_lds(float (&y)[S > 0 ? S : 1], const float (&x)[S > 0 ? S : 1], const float *rec, bool on,
                        bool valid, int h) {
    if constexpr (S > 0) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rec);
        if (on) {                       // wave-uniform: ONE branch, not one per element
            // multiply by a 0/1 mask instead of `valid ? .. : 0`: the ternary is turned into divergent
            // control flow around the LDS reads, with a full-array phi copy per element
            const float vf = valid ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float4 n = r4[slab_ch<S>(k, h)];
                y[k] = ((x[k] - n.x) * n.y + n.z) * vf;
            }
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) y[k] = x[k];
        }
    }
}

// dz coefficients {mean, ca, cb, cc} of channel `ch` of graph g: precomputed (A.coef) or derived here
// from the GraphNorm-backward sums S1,S2 and the output's GraphNorm record:
//   dz = a*dy - a*S2*r2/m * (z - mean) - a*S1/m
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

// dW += Dt (rows = out channel) x In (rows = in channel), contraction over the 32 pixels.
// The Dt fragments of lane (o, h) are 16 pixels of channel o, so the bias gradient
// db[o] = sum_px Dt[o][px] falls out of the same LDS reads (WITH_DB).
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

constexpr int NW = 8;                // waves per workgroup (2 per SIMD)

// Operand sets kept in workgroup-shared LDS (one float per lane per k-step, stored as
// [step/4][lane][4] so a ds_read_b128 returns four consecutive k-steps of a lane), the
// wave-private per-graph records and the wave-private 32x36 transposition tiles.
// Tile slots per wave: S0 = h1 / later x_a, S1 = h2 / dpre_1 / later x_b, S2 = dz / dpre_0
// (a slab with fewer than 32 channels zero-fills its slot before staging its rows).
template <int CA, int CB, int DEPTH>
struct BwdLayout {
    static constexpr int SA = CA / 2, SB = CB / 2;
    static constexpr int pad4(int x) { return (x + 3) & ~3; }
    static constexpr PkBwd PK = pk_bwd(CA, CB, DEPTH);                    // single source of truth: fgnn_pack.h
    static constexpr int OFF_W1A = PK.off_w1a;                            // SA steps: forward layer 0, slab a
    static constexpr int OFF_W1B = PK.off_w1b;                            // SB steps: forward layer 0, slab b
    static constexpr int OFF_WH = PK.off_wh;                              // 16*(DEPTH-2): forward layers 1..DEPTH-2
    static constexpr int BIAS_F = PK.bias_f;                              // compact bias tail b_0..b_{DEPTH-2}: [layer][h][16]
    static constexpr int OFF_WT = PK.off_wt;                              // 16*(DEPTH-1): W_l^T, l=1..DEPTH-1
    static constexpr int OFF_WT0A = PK.off_wt0a;                          // 16: W_0^T slab a
    static constexpr int OFF_WT0B = PK.off_wt0b;                          // 16: W_0^T slab b
    static constexpr int NSTEPS = PK.steps;
    static constexpr int WEIGHT_F = PK.floats;                            // floats
    static constexpr int REC_F = 3 * 32 * 4;                              // per wave: nrm a, nrm b, coef
    static constexpr int SLOT_XA = 0, SLOT_XB = 1, NSLOT = 3;             // x tiles alias the dead h1 / dpre_1 slots
    static constexpr int PCOUNT = 32 * (CA + CB) + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int TILE_F_ALL = NW * NSLOT * TILE_F;
    static constexpr int RED_F = NW * PCOUNT;                             // final reduction reuses the whole allocation
    static constexpr int WGK_F = FGNN_BWD_COEF_GRAPHS * 128;              // workgroup cache of dz coefficient records
    static constexpr int MAIN_F = WEIGHT_F + NW * REC_F + TILE_F_ALL + WGK_F;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

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

// bias[ch_of(r, h)], r = 0..15, of one layer from the compact tail (broadcast reads)
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

// SKIP (ragged batches with A.ranges): the workgroup's tile range comes from fgnn_ragged_tile_ranges (equal work), the
// waves step over tiles without a single valid pixel (they contribute nothing to the parameter gradients); the only thing
// such a tile still delivers is an empty S1/S2 record, stored after the main loop.
template <int CA, int CB, int DEPTH, bool PK = false, bool SKIP = false>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_kernel(const fgnn_mlp_bwd_args A, const int tpg,
                                                              const int total_tiles) {
    static_assert(DEPTH >= 1 && DEPTH <= 3, "tile-slot plan covers depth <= 3");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BwdLayout<CA, CB, DEPTH>;
    constexpr int CIN = CA + CB, SA = CA / 2, SB = CB / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int P = A.N * A.N;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PK) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    const View vdy = make_view(A.dy, A.dgstride, A.ldd, A.G);
    const View vz = make_view(A.z, A.zgstride, A.ldz, A.G);
    const View vdxa = make_view(A.dxa, A.dxa_gstride, A.dxa_ld, A.G);
    const View vdxb = make_view(A.dxb, A.dxb_gstride, A.dxb_ld, A.G);

    PH_DECL
    float *wl = smem;                                   // shared operand sets
    float *rec = smem + L::WEIGHT_F + wv * L::REC_F;    // wave-private per-graph records
    float *recA = rec, *recB = rec + 128, *recK = rec + 256;
    float *tiles = smem + L::WEIGHT_F + NW * L::REC_F;
    float *my = tiles + wv * (L::NSLOT * TILE_F);
    float *S0 = my, *S1 = my + TILE_F, *S2 = my + 2 * TILE_F;
    float *XA = my + L::SLOT_XA * TILE_F;
    float *XB = my + L::SLOT_XB * TILE_F;

    // (operand image copy moved below: the first tile's loads are issued first)

    // ---- persistent accumulators ----
    // Two-channel input (block 1: adjacency + degree): the layer-0 weight gradient has 2 useful columns, so an MFMA
    // over a zero-padded 32-channel tile (and its LDS staging) is replaced by 48 per-lane FMAs / adds per tile,
    // reduced over the pixels once at the end of the kernel.
    constexpr bool VW0 = (CA == 2 && CB == 0);
    float w0v[VW0 ? 32 : 1], b0v[VW0 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (VW0 ? 32 : 1); ++r) w0v[r] = 0.f;
#pragma unroll
    for (int r = 0; r < (VW0 ? 16 : 1); ++r) b0v[r] = 0.f;
    f32x16 dW0a, dW0b, dWh[DEPTH > 1 ? DEPTH - 1 : 1];
    float db[DEPTH];
    zero16(dW0a);
    zero16(dW0b);
#pragma unroll
    for (int l = 0; l + 1 < DEPTH; ++l) zero16(dWh[l]);
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] = 0.f;

    // static, strided tile assignment inside the workgroup's contiguous range: the order in
    // which a wave accumulates its weight gradients is fixed -> bit-reproducible results
    const int nwg = gridDim.x;
    const int q = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const bool normA = A.a.nrm != nullptr, normB = (CB > 0) && A.b.nrm != nullptr;
    const bool emit = (CA == 32) && (CB == 0) && normA && A.dxa != nullptr && A.s12part != nullptr;
    // dz coefficients from the per-tile sums its consumer left behind (the work of fgnn_gn_bwd_coef_tiles,
    // done per workgroup for the <= FGNN_BWD_COEF_GRAPHS graphs its tile range touches)
    // compiled into the two-slab kernels only (its user is mlp3): in the single-slab variants the extra prologue code
    // perturbs the register allocation of the tile loop (0 -> 40 spilled SGPRs in <32,0,3>)
    const bool from_tiles = (CB > 0) && A.s12tiles != nullptr;
    float *wgK = tiles + L::TILE_F_ALL;
    const int g0 = T0 / tpg;

    // Prologue = ONE memory round trip: the operand image is requested first (into registers), then the first
    // tile's input slabs and the per-graph records; only then come the loads whose values are needed at once
    // (dz coefficients derived from s12 / znrm, nvalid) and the image's way into LDS.
    PkRegs<L::WEIGHT_F / 4, 64 * NW> img;
    if (A.packed) pk_load_regs(img, A.packed);
    __builtin_amdgcn_sched_barrier(0);      // keep these loads first (the scheduler would sink them to their use)
    float xa[SA > 0 ? SA : 1], xb[SB > 0 ? SB : 1];
    float4 rk = make_float4(0.f, 0.f, 0.f, 0.f), ra = rk, rb = rk;
    int cached_g = -1, cur_nv = A.N;
    int first = T0 + wv;
    if constexpr (SKIP) first = __builtin_amdgcn_readfirstlane(next_live_tile(first, T1, NW, tpg, A.N, A.nvalid));
    {
        const int t = first;
        const TileCtx c = decode_tile(t, t < T1, tpg, A.N, P, j);
        load_slab<SA, PK>(xa, va, ps, c, h);
        load_slab<SB, PK>(xb, vb, ps, c, h);
        if (t < T1 && lane < 32) {
            if (!from_tiles) rk = coef_record(A, c.g, lane);
            if (normA && lane < CA) {
                ra = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                ra.z = A.a.beta ? A.a.beta[lane] : 0.f;
            }
            if (normB && lane < CB) {
                rb = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)c.g * A.b.C + lane];
                rb.z = A.b.beta ? A.b.beta[lane] : 0.f;
            }
        }
        if (t < T1) {
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
    }
    PH(12)              // prologue a: kernel arguments, views, first tile's loads + records issued
    if (from_tiles) {
        const int g1 = T1 > T0 ? (T1 - 1) / tpg : g0 - 1;
        float2 *scr = reinterpret_cast<float2 *>(tiles);          // [16 slices][32 channels]; tiles are free here
        const int cc = threadIdx.x & 31, sl = threadIdx.x >> 5;
        for (int g = g0; g <= g1; ++g) {
            float p1 = 0.f, p2 = 0.f;
            // eight tile records in flight per thread and pass: rolled, this loop pays one (cold) memory round trip per
            // record -- five in a row for N = 50 -- at the very start of the kernel
            constexpr int TS = (64 * NW) / 32, U = 8;
            for (int t0 = sl; t0 < tpg; t0 += TS * U) {
                float2 v[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int t = t0 + TS * k;
                    v[k] = reinterpret_cast<const float2 *>(A.s12tiles)[((long long)g * tpg + (t < tpg ? t : 0)) * FGNN_H + cc];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (t0 + TS * k < tpg) {
                        p1 += v[k].x;
                        p2 += v[k].y;
                    }
                }
            }
            scr[sl * 32 + cc] = make_float2(p1, p2);
            __syncthreads();
            if (threadIdx.x < 32) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < (64 * NW) / 32; ++k) {               // fixed order
                    s1 += scr[k * 32 + cc].x;
                    s2 += scr[k * 32 + cc].y;
                }
                const float2 sv = make_float2(s1, s2);
                if (A.s12_out) reinterpret_cast<float2 *>(A.s12_out)[(long long)g * FGNN_H + cc] = sv;
                const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + cc];
                reinterpret_cast<float4 *>(wgK)[(g - g0) * 32 + cc] = coef_from_sums(n, sv, (float)nvalid_of(A.nvalid, g, A.N));
            }
            __syncthreads();
        }
        if (cached_g >= 0 && lane < 32) rk = reinterpret_cast<const float4 *>(wgK)[(cached_g - g0) * 32 + lane];
    }
    // ---- operand image -> LDS: straight copy of the pre-packed image, or build it here ----
    if (A.packed) {
        pk_store_regs(wl, img);
    } else {
        const float *Wp[FGNN_MAX_DEPTH] = {A.W[0], A.W[1], A.W[2]};
        const float *Bp[FGNN_MAX_DEPTH] = {A.bias[0], A.bias[1], A.bias[2]};
        constexpr PkBwd pk = L::PK;
        for (int e = threadIdx.x; e < L::WEIGHT_F; e += 64 * NW) {
            if (e < L::BIAS_F) {
                const int t = e >> 6, l = e & 63;
                wl[(t >> 2) * 256 + l * 4 + (t & 3)] = pk_bwd_value(pk, CA, CB, Wp, t, l);
            } else {
                wl[e] = pk_bias_value(Bp, e - L::BIAS_F);
            }
        }
    }
    PH(13)              // prologue b: operand image loaded and written to LDS
    if (lane < 32) {
        reinterpret_cast<float4 *>(recK)[lane] = rk;
        reinterpret_cast<float4 *>(recA)[lane] = ra;
        reinterpret_cast<float4 *>(recB)[lane] = rb;
    }
    __syncthreads();
    PH(9)               // prologue: first loads issued, operand image copied, barrier
    int tnext = 0;
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NW;
        if constexpr (SKIP) tnext = __builtin_amdgcn_readfirstlane(next_live_tile(tnext, T1, NW, tpg, A.N, A.nvalid));
        const TileCtx c = decode_tile(tile, true, tpg, A.N, P, j);
        if (c.g != cached_g) {
            // per-graph records -> wave-private LDS.  Issued (and waited for) BEFORE the loads
            // below so that the in-order vmcnt wait does not drain them.
            if (lane < 32) {
                const float4 k4 = from_tiles ? reinterpret_cast<const float4 *>(wgK)[(c.g - g0) * 32 + lane]
                                             : coef_record(A, c.g, lane);
                reinterpret_cast<float4 *>(recK)[lane] = k4;
                if (normA && lane < CA) {
                    float4 n = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                    n.z = A.a.beta ? A.a.beta[lane] : 0.f;
                    reinterpret_cast<float4 *>(recA)[lane] = n;
                }
                if (normB && lane < CB) {
                    float4 n = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)c.g * A.b.C + lane];
                    n.z = A.b.beta ? A.b.beta[lane] : 0.f;
                    reinterpret_cast<float4 *>(recB)[lane] = n;
                }
            }
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
        const bool c_valid = tile_valid(c, cur_nv);
        // early read-modify-write prefetch of dx only exists in the single-slab variants
        // (mlp1 / mlp2 accumulating into d_in); two-slab kernels re-read at the store
        constexpr bool EARLY_RMW = (CB == 0);
        float dyr[16], zr[16], old[EARLY_RMW ? 16 : 1];
        const bool rmw = EARLY_RMW && A.dxa != nullptr && A.accumulate_a;

        // ---- forward recompute of the hidden activations (kept in the LDS tiles S0 / S1) ----
        if constexpr (DEPTH > 1) {
            f32x16 acc;
            {
                float ya[SA > 0 ? SA : 1], yb[SB > 0 ? SB : 1];
                norm_from_lds<SA>(ya, xa, recA, normA, c_valid, h);
                norm_from_lds<SB>(yb, xb, recB, normB, c_valid, h);
                PH(0)   // tile decode, records, x arrived + normalised
                float b0[16];
                load_bias(b0, wl + L::BIAS_F, 0, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = b0[r];
                float w1a[SA > 0 ? SA : 1];
                load_ops<L::OFF_W1A, SA>(w1a, wl, lane);
#pragma unroll
                for (int s = 0; s < SA; ++s) acc = mfma32(w1a[s], ya[s], acc);
                if constexpr (SB > 0) {
                    float w1b[SB > 0 ? SB : 1];
                    load_ops<L::OFF_W1B, SB>(w1b, wl, lane);
#pragma unroll
                    for (int s = 0; s < SB; ++s) acc = mfma32(w1b[s], yb[s], acc);
                }
            }
            // this tile's dy / z (consumed after the recompute) and, when accumulating, the
            // current dx values (consumed at the very end) fly behind the recompute MFMAs
            load_rows16(dyr, vdy, c, h);
            load_rows16(zr, vz, c, h);
            if constexpr (EARLY_RMW) {
                if (rmw) load_rows16(old, vdxa, c, h);
            }
            PH(1)       // layer-0 recompute issued, dy / z / old requested
#pragma unroll
            for (int l = 1; l < DEPTH; ++l) {
                float hid[16];
                float *Hs = (l == 1) ? S0 : S1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hid[r] = relu1(acc[r]);
                    Hs[ch_of(r, h) * TLD + j] = hid[r];
                }
                if (l + 1 < DEPTH) {
                    float bl[16], wf[16];
                    load_bias(bl, wl + L::BIAS_F, 1, h);
                    load_ops<L::OFF_WH, 16>(wf, wl, lane);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = bl[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc = mfma32(wf[r], hid[r], acc);
                }
            }
        }

        if constexpr (DEPTH == 1) {
            load_rows16(dyr, vdy, c, h);
            load_rows16(zr, vz, c, h);
            if constexpr (EARLY_RMW) {
                if (rmw) load_rows16(old, vdxa, c, h);
            }
        }
        PH(2)           // hidden activations recomputed and staged
        // ---- dz from (dy, z, coef) ----
        float dpre[16];
        {
            const float4 *kp = reinterpret_cast<const float4 *>(recK) + 4 * h;
            const float vf = c_valid ? 1.f : 0.f;      // mask multiply, see norm_from_lds
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 k = kp[(r & 3) + 8 * (r >> 2)];
                dpre[r] = (k.y * dyr[r] + k.z * (zr[r] - k.x) + k.w) * vf;
            }
        }

        PH(3)           // dz (waits for dy, z)
        // ---- backward through the hidden layers: l = DEPTH-1 .. 1 ----
        //   l == DEPTH-1 : D tile = S2, input tile = (DEPTH == 3 ? S1 : S0)
        //   l == 1 (DEPTH == 3): D tile = S1 (h2 is dead), input tile = S0
#pragma unroll
        for (int l = DEPTH - 1; l >= 1; --l) {
            float *Dt = (l == DEPTH - 1) ? S2 : S1;
            const float *In = (l == 2) ? S1 : S0;
#pragma unroll
            for (int r = 0; r < 16; ++r) Dt[ch_of(r, h) * TLD + j] = dpre[r];
            float wt[16], hsv[16];
            if (l == 1) load_ops<L::OFF_WT, 16>(wt, wl, lane);
            if (l == 2) load_ops<L::OFF_WT + (DEPTH > 2 ? 16 : 0), 16>(wt, wl, lane);
            // h_{l-1} of this lane's (channel, pixel) pairs for the ReLU mask: read before the MFMA chain
#pragma unroll
            for (int r = 0; r < 16; ++r) hsv[r] = In[ch_of(r, h) * TLD + j];
            f32x16 acc;
            zero16(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = mfma32(wt[r], dpre[r], acc);
            dWh[l - 1] = wgrad_tile<true>(Dt, In, dWh[l - 1], db[l], lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = hsv[r] > 0.f ? acc[r] : 0.f;
            PH(4 + (DEPTH - 1 - l))   // hidden layer backward (dgrad + wgrad + mask)
        }

        // ---- layer 0: D tile = (DEPTH == 1 ? S2 : (DEPTH == 2 ? S1 : S2)); x tiles re-staged now ----
        {
            // depth 2: S1 is free in the single-slab kernels; with a second slab it is the x_b slot, so dpre_0 goes to S2 (its
            // last readers, the layer-1 weight-gradient operands, were issued before: LDS is in order within a wave)
            float *Dt = (DEPTH == 2 && CB == 0) ? S1 : S2;
            if constexpr (VW0) {
                float ya[1];
                norm_from_lds<SA>(ya, xa, recA, normA, c_valid, h);      // channel h of this lane's pixel
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
            for (int r = 0; r < 16; ++r) Dt[ch_of(r, h) * TLD + j] = dpre[r];
            }
            if constexpr (!VW0) {
                float ya[SA > 0 ? SA : 1];
                norm_from_lds<SA>(ya, xa, recA, normA, c_valid, h);
                if constexpr (CA < 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) XA[ch_of(r, h) * TLD + j] = 0.f;
                }
#pragma unroll
                for (int s = 0; s < SA; ++s) XA[slab_ch<SA>(s, h) * TLD + j] = ya[s];
            }
            if constexpr (CB > 0) {
                float yb[SB > 0 ? SB : 1];
                norm_from_lds<SB>(yb, xb, recB, normB, c_valid, h);
                if constexpr (CB < 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) XB[ch_of(r, h) * TLD + j] = 0.f;
                }
#pragma unroll
                for (int s = 0; s < SB; ++s) XB[slab_ch<SB>(s, h) * TLD + j] = yb[s];
            }
            // x of this tile is staged; prefetch the next tile's input slabs now (register pressure
            // is lowest here).  The S1/S2-emitting variant still needs xa for (z - mean).
            float nxa[(CA == 32 && CB == 0) ? 16 : 1];
            {
                const int tn = tnext;
                const TileCtx cn = decode_tile(tn, tn < T1, tpg, A.N, P, j);
                if constexpr (CA == 32 && CB == 0) {
                    load_raw<SA>(nxa, va, cn, h);
                } else {
                    load_slab<SA, PK>(xa, va, ps, cn, h);
                    load_slab<SB, PK>(xb, vb, ps, cn, h);
                }
            }
            PH(6)       // layer-0 staging, next x requested
            f32x16 dxa_acc, dxb_acc;
            if (A.dxa) {
                float wt[16];
                load_ops<L::OFF_WT0A, 16>(wt, wl, lane);
                zero16(dxa_acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) dxa_acc = mfma32(wt[r], dpre[r], dxa_acc);
            }
            if constexpr (CB > 0) {
                if (A.dxb) {
                    float wt[16];
                    load_ops<L::OFF_WT0B, 16>(wt, wl, lane);
                    zero16(dxb_acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dxb_acc = mfma32(wt[r], dpre[r], dxb_acc);
                }
            }
            if constexpr (!VW0) dW0a = wgrad_tile<true>(Dt, XA, dW0a, db[0], lane);
            if constexpr (CB > 0) {
                float dummy = 0.f;
                dW0b = wgrad_tile<false>(Dt, XB, dW0b, dummy, lane);
            }
            PH(7)       // layer-0 dgrad + wgrad issued
            if (A.dxa) {
                const int voff = lane_off<4>(vdxa, c, h);
                const int s0 = c.g * vdxa.gs4;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = dxa_acc[r];
                if constexpr (EARLY_RMW) {
                    if (rmw) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] += old[r];
                    }
                } else {
                    if (A.accumulate_a) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            v[r] += buf_load(vdxa, ch_of(r, h) < CA ? voff : OOB_OFF, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store(v[r], vdxa, ch_of(r, h) < CA ? voff : OOB_OFF, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                if constexpr (CA == 32 && CB == 0) {
                    if (emit) {
                        // GraphNorm-backward sums of the producer of slab a over this tile:
                        // S1 = sum v, S2 = sum v * (z_a - mean_a); transposed through S1 / S2 so that
                        // lane (ch, h) sums 16 pixels of its channel.  v is 0 on invalid pixels.
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
                            reinterpret_cast<float2 *>(A.s12part)[((long long)c.g * tpg + c.tt) * FGNN_H + j] = o;
                        }
                    }
                }
            }
            if constexpr (CB > 0) {
                if (A.dxb) {
                    const int voff = lane_off<4>(vdxb, c, h);
                    const int s0 = c.g * vdxb.gs4;
                    float vb2[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) vb2[r] = dxb_acc[r];
                    if (A.accumulate_b) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            vb2[r] += buf_load(vdxb, ch_of(r, h) < CB ? voff : OOB_OFF, s0 + ((r & 3) + 8 * (r >> 2)) * vdxb.ld4);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        buf_store(vb2[r], vdxb, ch_of(r, h) < CB ? voff : OOB_OFF, s0 + ((r & 3) + 8 * (r >> 2)) * vdxb.ld4);
                }
            }
            if constexpr (CA == 32 && CB == 0) {
#pragma unroll
                for (int s = 0; s < SA; ++s) xa[s] = nxa[s];
            }
            PH(8)       // dx stores (+ S1/S2 emission)
        }
    }

    if constexpr (SKIP) {
        // padding-only tiles of this wave's share: empty S1/S2 sums.  dx is NOT written there: its consumers step over the
        // same tiles (MLP kernels) or read the valid corner only (the matmul backward).
        if (emit) {
            for (int t = T0 + wv; t < T1; t += NW) {
                const TileCtx c = decode_tile(t, true, tpg, A.N, P, j);
                if (tile_live(c.tt, A.N, A.nvalid[c.g])) continue;
                if (h == 0)
                    reinterpret_cast<float2 *>(A.s12part)[((long long)c.g * tpg + c.tt) * FGNN_H + j] = make_float2(0.f, 0.f);
            }
        }
    }

    // ---- workgroup reduction of the parameter gradients (fixed order over the waves) ----
    // layout: [W0 (32*CIN) | b0 (32) | W1 (1024) | b1 (32) | ...]
    constexpr int PCOUNT = L::PCOUNT;
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] += __shfl_xor(db[l], 32);
    if constexpr (VW0) {                    // sum the per-pixel partials over the 32 lanes of each half-wave
#pragma unroll
        for (int r = 0; r < 32; ++r) w0v[r] = half_sum(w0v[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) b0v[r] = half_sum(b0v[r]);
    }
    auto put_partials = [&](float *red) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ch_of(r, h);
            if constexpr (VW0) {
                if (j == 0) {
                    red[o * CIN] = w0v[2 * r];
                    red[o * CIN + 1] = w0v[2 * r + 1];
                    red[32 * CIN + o] = b0v[r];
                }
            } else {
                if (j < CA) red[o * CIN + j] = dW0a[r];
                if (CB > 0 && j < CB) red[o * CIN + CA + j] = dW0b[r];
            }
        }
        int off = 32 * CIN;
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
    };
    __syncthreads();                       // everyone done with the operand sets and the tile buffers
    PH(10)              // waiting for the slowest wave of the workgroup
    put_partials(smem + wv * PCOUNT);      // the whole LDS allocation is free now
    __syncthreads();
    static_assert(PCOUNT % 4 == 0, "partials are summed four at a time");
    float4 *out = reinterpret_cast<float4 *>(A.wpart + (long long)blockIdx.x * PCOUNT);
    const float4 *part4 = reinterpret_cast<const float4 *>(smem);
    for (int e = threadIdx.x; e < PCOUNT / 4; e += 64 * NW) {
        float4 a = part4[e];
#pragma unroll
        for (int w = 1; w < NW; ++w) {                                  // fixed order
            const float4 b = part4[w * (PCOUNT / 4) + e];
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        out[e] = a;
    }
    PH(11)              // workgroup reduction + partial store
    PH_FLUSH
}

template <int CA, int CB, int DEPTH, bool PK = false, bool SKIP = false>
int launch_bwd_impl(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    constexpr int LDS = BwdLayout<CA, CB, DEPTH>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_kernel<CA, CB, DEPTH, PK, SKIP>, LDS);
    // BWD_WG workgroups = rows of the partials buffer (fgnn_grad_finalize); cu_share == 2: half of the CUs, half the rows
    hipLaunchKernelGGL((mlp_bwd_kernel<CA, CB, DEPTH, PK, SKIP>), dim3(a->cu_share == 2 && !SKIP ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st,
                       *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, int CB, int DEPTH, bool PK = false>
int launch_bwd(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    static_assert(BWD_WG == FGNN_RANGE_WG, "fgnn_ragged_tile_ranges splits for the backward grid");
    if (a->ranges) return launch_bwd_impl<CA, CB, DEPTH, PK, true>(a, tpg, total, st);
    return launch_bwd_impl<CA, CB, DEPTH, PK, false>(a, tpg, total, st);
}

template <int DEPTH>
int dispatch_c(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    const int ca = a->a.C, cb = a->b.C;
    if (a->xbits) {          // the 2-channel slab comes from the bit-packed adjacency (built for depth 3)
        if constexpr (DEPTH == 3) {
            if (ca == 2 && cb == 0) return launch_bwd<2, 0, DEPTH, true>(a, tpg, total, st);
            if (ca == 32 && cb == 2) return launch_bwd<32, 2, DEPTH, true>(a, tpg, total, st);
        }
        fgnn_set_error("fgnn_mlp_bwd: xbits needs depth 3 and a 2-channel slab (2 or 32+2 input channels), got depth %d, %d + %d",
                       DEPTH, ca, cb);
        return 1;
    }
#define FGNN_CASE(A_, B_) \
    if (ca == A_ && cb == B_) return launch_bwd<A_, B_, DEPTH>(a, tpg, total, st);
    FGNN_CASE(2, 0)
    FGNN_CASE(16, 0)
    FGNN_CASE(32, 0)
    FGNN_CASE(32, 2)
    FGNN_CASE(32, 32)
#undef FGNN_CASE
    fgnn_set_error("fgnn_mlp_bwd: unsupported input channels (%d + %d); built for 2, 16, 32, 32+2, 32+32", ca, cb);
    return 1;
}

}  // namespace

#ifdef FGNN_PHASES
extern "C" int fgnn_debug_phase_buffer(void *p, int ca, int cb) {
    const int sel = ca * 100 + cb;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_sel), &sel, sizeof(sel)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int fgnn_mlp_bwd_num_workgroups(void) { return BWD_WG; }

extern "C" int fgnn_mlp_bwd_coef_tiles_supported(int G, int N) {
    const long long tpg = fgnn_tiles_per_graph(N), total = (long long)G * tpg;
    const long long per_wg = (total + BWD_WG - 1) / BWD_WG;
    // a range of per_wg consecutive tiles touches at most (per_wg + tpg - 2) / tpg + 1 graphs
    return (per_wg + tpg - 2) / tpg + 1 <= FGNN_BWD_COEF_GRAPHS ? 1 : 0;
}

extern "C" int fgnn_mlp_param_count(int Cin, int depth) { return 32 * Cin + 32 + (depth - 1) * (32 * 32 + 32); }

extern "C" int fgnn_mlp_bwd(const fgnn_mlp_bwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_bwd: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_bwd: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(a->depth >= 1 && a->depth <= FGNN_MAX_DEPTH, "fgnn_mlp_bwd: depth %d not in 1..%d", a->depth, FGNN_MAX_DEPTH);
    const bool pk_a = a->xbits && a->a.C == 2, pk_b = a->xbits && a->b.C == 2;     // that slab's memory is never touched
    FGNN_CHECK((a->a.ptr || pk_a) && a->a.C > 0, "fgnn_mlp_bwd: slab a missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr || pk_b, "fgnn_mlp_bwd: slab b has channels but no pointer");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_bwd: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(!(pk_a && a->dxa) && !(pk_b && a->dxb), "fgnn_mlp_bwd: no gradient with respect to the packed adjacency");
    FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd: missing dy/z/wpart");
    FGNN_CHECK(a->coef || (a->s12 && a->znrm) || (a->s12tiles && a->znrm), "fgnn_mlp_bwd: need coef, or s12 + znrm, or s12tiles + znrm");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim &&
                   G * a->dxa_gstride < lim && G * a->dxb_gstride < lim,
                   "fgnn_mlp_bwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    for (int l = 0; l < a->depth; ++l) FGNN_CHECK(a->W[l] && a->bias[l], "fgnn_mlp_bwd: missing weights layer %d", l);
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd: too many tiles");
    FGNN_CHECK(!a->ranges || (a->nvalid && !a->s12tiles), "fgnn_mlp_bwd: ranges need nvalid and exclude s12tiles");
    FGNN_CHECK(!a->s12tiles || a->b.C > 0, "fgnn_mlp_bwd: s12tiles is built into the two-slab kernels only; use fgnn_gn_bwd_coef_tiles");
    FGNN_CHECK(!a->s12tiles || fgnn_mlp_bwd_coef_tiles_supported(a->G, a->N),
               "fgnn_mlp_bwd: s12tiles needs a workgroup's tile range to span <= %d graphs (G=%d N=%d); "
               "use fgnn_gn_bwd_coef_tiles", FGNN_BWD_COEF_GRAPHS, a->G, a->N);
    hipStream_t st = (hipStream_t)stream;
    if (a->depth == 1) return dispatch_c<1>(a, tpg, (int)total, st);
    if (a->depth == 2) return dispatch_c<2>(a, tpg, (int)total, st);
    return dispatch_c<3>(a, tpg, (int)total, st);
}
