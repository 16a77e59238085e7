// Fused MlpBlock_Real forward (conv1x1 + ReLU chain, last conv linear) for gfx950.
//
// Replaces models/layers.py:126-131 of the reference (the GraphNorm reductions of
// :72-73 are emitted as per-tile partials; the normalisation itself is applied by the
// consumer on load).  Formulation: out[o][p] = sum_c W[o][c] in[c][p] with the 32 output
// channels as MFMA rows and 32 pixels as MFMA columns, v_mfma_f32_32x32x2_f32 (exact
// fp32).  The D fragment of one layer (lane = pixel, register = channel) is directly the
// B operand of the next layer, so the whole conv/ReLU chain stays in registers; HBM sees
// one read of the input slabs and one write of z per MLP.
//
// Execution shape (from the PMC profile of the first version: one wave per SIMD left the
// MFMA pipe idle >60 % of the time behind exposed VALU/LDS/VMEM latency):
//   * one 512-thread workgroup per CU = 2 waves per SIMD, so one wave's epilogue / waits
//     overlap the other's MFMA chain;
//   * all MFMA A-operands (weights, biases) live in workgroup-shared LDS, laid out
//     [k-step/4][lane][4] so a ds_read_b128 returns four k-steps of a lane;
//   * a tile = 32 consecutive pixels of one graph; every workgroup owns a contiguous tile
//     range, wave w takes tiles w, w+NW, ... of it (waves w, w+4, ... share a SIMD, so the
//     SIMDs stay balanced); the raw inputs of a wave's next tile are prefetched while the
//     current one computes;
//   * per-graph GraphNorm records of the input slabs are cached in registers and reloaded
//     BEFORE the prefetch is issued (vmcnt is in-order);
//   * tile statistics: the z tile is transposed through a wave-private LDS tile so that a
//     lane owns one channel and sums its 16 pixels in registers (two-pass mean / M2).
#include "fgnn_common.h"
#include "fgnn_pack.h"

#ifdef FGNN_PHASES
// debug build only: absolute s_memtime stamps per wave {start, prologue done, loop done, tiles}
__device__ unsigned long long *g_fwd_stamp = nullptr;
__device__ int g_fwd_sel = 0;
#endif

namespace {

constexpr int TLD = 36;              // LDS tile row stride (floats)
constexpr int TILE_F = 32 * TLD;

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

// channel contracted by k-step k in half-wave h for a slab with S = C/2 steps: 32-channel
// slabs use the accumulator-fragment pairing (same as the hidden layers), narrower ones (2k, 2k+1)
template <int S>
DEVI constexpr int slab_ch(int k, int h) { return S == 16 ? ch_of(k, h) : 2 * k + h; }
template <int S>
DEVI constexpr int slab_kbase(int k) { return S == 16 ? (k & 3) + 8 * (k >> 2) : 2 * k; }
template <int S>
DEVI constexpr int slab_hmul() { return S == 16 ? 4 : 1; }

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

// y = (x - mean) * a + beta with the per-graph records {mean, a, beta, -} read from wave-private LDS
template <int S>
DEVI void apply_norm(float (&x)[S > 0 ? S : 1], const float *rec, bool on, bool valid, int h) {
    if constexpr (S > 0) {
        if (on) {
            const float4 *r4 = reinterpret_cast<const float4 *>(rec);
            // 0/1 mask multiply instead of `valid ? .. : 0` (the ternary becomes divergent control flow
            // around the LDS reads with a full-array phi copy per element)
            const float vf = valid ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float4 n = r4[slab_ch<S>(k, h)];
                x[k] = ((x[k] - n.x) * n.y + n.z) * vf;
            }
        }
    }
}

// LDS operand sets of one MLP (units: k-steps; one float per lane per step)
// waves per workgroup: 4 per SIMD (the forward kernels need <= 128 VGPRs); one workgroup per CU
template <int CA, int CB>
constexpr int fwd_waves() { return 16; }

template <int CA, int CB, int NMLP, int DEPTH>
struct FwdLayout {
    static constexpr int NW = fwd_waves<CA, CB>();
    static constexpr int SA = CA / 2, SB = CB / 2;
    static constexpr int pad4(int x) { return (x + 3) & ~3; }
    static constexpr PkFwd PK = pk_fwd(CA, CB, DEPTH);             // single source of truth: fgnn_pack.h
    static constexpr int OFF_W1A = PK.off_w1a;
    static constexpr int OFF_W1B = PK.off_w1b;
    static constexpr int OFF_WH = PK.off_wh;                       // 16*(DEPTH-1)
    static constexpr int BIAS_F = PK.bias_f;                       // compact bias tail: [layer][h][16]
    static constexpr int MLP_STEPS = PK.steps;
    static constexpr int MLP_F = PK.floats;                        // floats per MLP image
    static constexpr int WEIGHT_F = NMLP * MLP_F;
    static constexpr int REC_F = 2 * 32 * 4;                       // per wave: records of slab a, slab b
    static constexpr int LDS_F = WEIGHT_F + NW * (TILE_F + REC_F);
};

template <int CNT>
DEVI void load_ops(float (&dst)[CNT > 0 ? CNT : 1], const float *wl, int off_steps, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(wl) + (off_steps / 4) * 64 + lane;
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
// waves step over tiles without a single valid pixel; those tiles only get empty statistics records after the main loop.
// DBG (fgnn_debug_mlp_fwd_masks, test-only): the same tile code also exports the ReLU decisions it takes -- one bit per hidden
// pre-activation, dbg[m][((g * (DEPTH-1) + layer) * 32 + channel) * tpg + tile], bit j = pixel 32 * tile + j.
template <int CA, int CB, int NMLP, int DEPTH, bool PK, bool SKIP, bool DBG>
DEVI void mlp_fwd_body(const fgnn_mlp_fwd_args A, const int tpg, const int total_tiles, unsigned *const dbg0, unsigned *const dbg1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = FwdLayout<CA, CB, NMLP, DEPTH>;
    constexpr int NW = L::NW;
#ifdef FGNN_PHASES
    const unsigned long long st0_ = __builtin_amdgcn_s_memtime();
    unsigned long long st1_ = 0;
    int ntl_ = 0;
#endif
    constexpr int SA = CA / 2, SB = CB / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    young_prio(2, wv, blockDim.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int P = A.N * A.N;
    float *wl = smem;
    float *tl = smem + L::WEIGHT_F + wv * TILE_F;
    float *recA = smem + L::WEIGHT_F + NW * TILE_F + wv * L::REC_F, *recB = recA + 128;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PK) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    View vz[NMLP];
#pragma unroll
    for (int m = 0; m < NMLP; ++m) vz[m] = make_view(A.z[m], FGNN_H * A.ldz, A.ldz, A.G);

    // contiguous tile range of this workgroup; wave w takes tiles T0 + w, T0 + w + NW, ... (static)
    const int nwg = gridDim.x;
    const int q = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }

    // The first tile's input slabs and per-graph records are requested BEFORE the operand image
    // is copied, so the three dependent round trips of the prologue overlap into one.
    const bool normA = A.a.nrm != nullptr, normB = (CB > 0) && A.b.nrm != nullptr;
    // the operand image is requested first (into registers): its round trip overlaps the first tile's
    // SKIP: the waves take the range's LIVE tiles in turn (fgnn_common.h build_live_list: dealing raw indices and skipping the dead ones
    // left most of the 16 waves of a ragged batch's workgroup with 0 or 2 tiles); the list sits behind the layout's LDS, built before
    // anything is in flight
    int *live_list = reinterpret_cast<int *>(smem + L::LDS_F);
    int nlive = 0, li = wv, live_cnt = 0;
    bool use_list = false;
    if constexpr (SKIP) {
        nlive = build_live_list(live_list, live_list + LIVE_LIST_CAP, T0, T1, tpg, FGNN_TILE, A.N, A.nvalid, threadIdx.x, 64 * NW);
        use_list = nlive <= LIVE_LIST_CAP;
    }
    auto next_tile = [&](int cur, bool first) {        // the wave's tile after `cur` (first: its first one)
        if constexpr (SKIP) {
            if (use_list) {
                if (!first) li += NW;
                return __builtin_amdgcn_readfirstlane(li < nlive ? live_list[li] : T1);
            }
            return __builtin_amdgcn_readfirstlane(next_owned_live_tile_p(first ? T0 : cur + 1, T1, live_cnt, wv, NW, tpg, FGNN_TILE, A.N, A.nvalid));
        } else {
            return first ? T0 + wv : cur + NW;
        }
    };
    PkRegs<L::WEIGHT_F / 4, 64 * NW> img;
    if (A.packed) pk_load_regs(img, A.packed);
    __builtin_amdgcn_sched_barrier(0);      // keep these loads first (the scheduler would sink them to their use)
    int tile = next_tile(0, true);
    float xa[SA > 0 ? SA : 1], xb[SB > 0 ? SB : 1];
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    int cached_g = -1, cur_nv = A.N;
    {
        const TileCtx c = decode_tile(tile, tile < T1, tpg, A.N, P, j);
        load_slab<SA, PK>(xa, va, ps, c, h);
        load_slab<SB, PK>(xb, vb, ps, c, h);
        if (tile < T1 && lane < 32) {
            if (normA && lane < CA) {
                ra = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                ra.z = A.a.beta ? A.a.beta[lane] : 0.f;
            }
            if (normB && lane < CB) {
                rb = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)c.g * A.b.C + lane];
                rb.z = A.b.beta ? A.b.beta[lane] : 0.f;
            }
        }
        if (tile < T1) {
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
    }
    // ---- operand image -> LDS: straight copy of the pre-packed image, or build it here ----
    if (A.packed) {
        pk_store_regs(wl, img);
    } else {
        constexpr PkFwd pk = L::PK;
#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            const float *Wp[FGNN_MAX_DEPTH] = {A.W[m][0], A.W[m][1], A.W[m][2]};
            const float *Bp[FGNN_MAX_DEPTH] = {A.bias[m][0], A.bias[m][1], A.bias[m][2]};
            for (int e = threadIdx.x; e < L::MLP_F; e += 64 * NW) {
                if (e < L::BIAS_F) {
                    const int t = e >> 6, l = e & 63;
                    wl[m * L::MLP_F + (t >> 2) * 256 + l * 4 + (t & 3)] = pk_fwd_value(pk, CA, CB, Wp, t, l);
                } else {
                    wl[m * L::MLP_F + e] = pk_bias_value(Bp, e - L::BIAS_F);
                }
            }
        }
    }
    if (lane < 32) {
        reinterpret_cast<float4 *>(recA)[lane] = ra;
        reinterpret_cast<float4 *>(recB)[lane] = rb;
    }
    __syncthreads();
#ifdef FGNN_PHASES
    st1_ = __builtin_amdgcn_s_memtime();
#endif

    while (tile < T1) {
#ifdef FGNN_PHASES
        ++ntl_;
#endif
        const TileCtx c = decode_tile(tile, true, tpg, A.N, P, j);
        if (c.g != cached_g) {        // wave-uniform; issued before the prefetch (vmcnt is in-order)
            if (lane < 32) {
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
        // prefetch this wave's next tile (static strided assignment)
        const int next = next_tile(tile, false);
        float na[SA > 0 ? SA : 1], nb[SB > 0 ? SB : 1];
        {
            const TileCtx cn = decode_tile(next, next < T1, tpg, A.N, P, j);
            load_slab<SA, PK>(na, va, ps, cn, h);
            load_slab<SB, PK>(nb, vb, ps, cn, h);
        }
        apply_norm<SA>(xa, recA, normA, c_valid, h);
        apply_norm<SB>(xb, recB, normB, c_valid, h);

        const unsigned vmask = (unsigned)__ballot(c_valid);      // bit px = pixel valid (low half-wave)
        const float cnt = (float)__popc(vmask);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            const float *wm = wl + m * L::MLP_F;
            f32x16 acc;
            {
                float b0[16];
                load_bias(b0, wm + L::BIAS_F, 0, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = b0[r];
                float w1a[SA > 0 ? SA : 1];
                load_ops<SA>(w1a, wm, L::OFF_W1A, lane);
#pragma unroll
                for (int s = 0; s < SA; ++s) acc = mfma32(w1a[s], xa[s], acc);
                if constexpr (SB > 0) {
                    float w1b[SB > 0 ? SB : 1];
                    load_ops<SB>(w1b, wm, L::OFF_W1B, lane);
#pragma unroll
                    for (int s = 0; s < SB; ++s) acc = mfma32(w1b[s], xb[s], acc);
                }
            }
#pragma unroll
            for (int l = 1; l < DEPTH; ++l) {
                float hid[16], bl[16], wh[16];
                load_bias(bl, wm + L::BIAS_F, l, h);
                load_ops<16>(wh, wm, L::OFF_WH + 16 * (l - 1), lane);
                if constexpr (DBG) {      // the decision relu1 takes: bit pattern > 0 as a signed integer
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned long long mk = __ballot(__float_as_int(acc[r]) > 0);
                        if (lane == 0) {
                            unsigned *o = (m == 0 ? dbg0 : dbg1) + (((long long)c.g * (DEPTH - 1) + (l - 1)) * 32 + ch_of(r, 0)) * tpg + c.tt;
                            o[0] = (unsigned)mk;
                            o[4ll * tpg] = (unsigned)(mk >> 32);       // channel ch_of(r, 1) = ch_of(r, 0) + 4
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hid[r] = relu1(acc[r]);
                    acc[r] = bl[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc = mfma32(wh[r], hid[r], acc);
            }
            // epilogue: mask, store z, transpose through LDS, per-tile {mean, M2} with lane = channel
            const int zoff = lane_off<4>(vz[m], c, h);
            const int zs0 = c.g * vz[m].gs4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int chl = (r & 3) + 8 * (r >> 2);   // channel minus 4*h
                const float v = c_valid ? acc[r] : 0.f;
                buf_store(v, vz[m], zoff, zs0 + chl * vz[m].ld4);
                tl[(chl + 4 * h) * TLD + j] = v;
            }
            // lane (ch = j, h) owns pixels 16h .. 16h+15 of channel ch
            const float4 *rp = reinterpret_cast<const float4 *>(tl + j * TLD + 16 * h);
            float4 qv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) qv[k] = rp[k];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (qv[k].x + qv[k].y) + (qv[k].z + qv[k].w);
            s += __shfl_xor(s, 32);
            const float mean = s * inv;
            const unsigned mh = vmask >> (16 * h);
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = ((mh >> (4 * k + 0)) & 1u) ? qv[k].x - mean : 0.f;
                const float d1 = ((mh >> (4 * k + 1)) & 1u) ? qv[k].y - mean : 0.f;
                const float d2 = ((mh >> (4 * k + 2)) & 1u) ? qv[k].z - mean : 0.f;
                const float d3 = ((mh >> (4 * k + 3)) & 1u) ? qv[k].w - mean : 0.f;
                m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            m2 += __shfl_xor(m2, 32);
            if (h == 0) {
                float2 o;
                o.x = mean;
                o.y = m2;
                reinterpret_cast<float2 *>(A.part[m])[((long long)c.g * tpg + c.tt) * FGNN_H + j] = o;
            }
        }
        if (lane == 0) A.cnt[(long long)c.g * tpg + c.tt] = cnt;

#pragma unroll
        for (int s = 0; s < SA; ++s) xa[s] = na[s];
#pragma unroll
        for (int s = 0; s < SB; ++s) xb[s] = nb[s];
        tile = next;
    }
    if constexpr (SKIP) {
        // padding-only tiles of this wave's share: empty statistics.  Their z pixels are NOT written: every consumer of a
        // ragged slab either steps over the same tiles (the MLP kernels) or reads the valid corner only (matmul, pooling).
        for (int t = T0 + wv; t < T1; t += NW) {
            const TileCtx c = decode_tile(t, true, tpg, A.N, P, j);
            if (tile_live(c.tt, A.N, A.nvalid[c.g])) continue;
#pragma unroll
            for (int m = 0; m < NMLP; ++m) {
                if (h == 0)
                    reinterpret_cast<float2 *>(A.part[m])[((long long)c.g * tpg + c.tt) * FGNN_H + j] = make_float2(0.f, 0.f);
            }
            if (lane == 0) A.cnt[(long long)c.g * tpg + c.tt] = 0.f;
        }
    }
#ifdef FGNN_PHASES
    if (g_fwd_stamp && g_fwd_sel == CA * 1000 + CB * 10 + NMLP && (threadIdx.x & 63) == 0) {
        unsigned long long *o = g_fwd_stamp + ((long long)blockIdx.x * NW + wv) * 4;
        o[0] = st0_;
        o[1] = st1_;
        o[2] = __builtin_amdgcn_s_memtime();
        o[3] = ntl_;
    }
#endif
}

template <int CA, int CB, int NMLP, int DEPTH, bool PK = false, bool SKIP = false>
__global__ __launch_bounds__((64 * fwd_waves<CA, CB>()), (fwd_waves<CA, CB>() / 4)) void mlp_fwd_kernel(
    const fgnn_mlp_fwd_args A, const int tpg, const int total_tiles) {
    mlp_fwd_body<CA, CB, NMLP, DEPTH, PK, SKIP, false>(A, tpg, total_tiles, nullptr, nullptr);
}
template <int CA, int CB, int NMLP, int DEPTH, bool PK = false, bool SKIP = false>
__global__ __launch_bounds__((64 * fwd_waves<CA, CB>()), (fwd_waves<CA, CB>() / 4)) void mlp_fwd_dbg_kernel(
    const fgnn_mlp_fwd_args A, const int tpg, const int total_tiles, unsigned *d0, unsigned *d1) {
    mlp_fwd_body<CA, CB, NMLP, DEPTH, PK, SKIP, true>(A, tpg, total_tiles, d0, d1);
}

struct DbgOut {
    unsigned *m[2];
};

template <int CA, int CB, int NMLP, int DEPTH, bool PK = false, bool SKIP = false, bool DBG = false>
int launch_fwd_impl(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st, const DbgOut &d) {
    using L = FwdLayout<CA, CB, NMLP, DEPTH>;
    constexpr int NW = L::NW;
    constexpr int LDS = (L::LDS_F + (SKIP ? LIVE_LIST_CAP + NW : 0)) * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    int grid = (total + NW - 1) / NW;
    const int cap = a->cu_share == 2 ? 128 : 256;      // 2: half of the CUs (two launches on two streams side by side)
    if (grid > cap) grid = cap;
    if (SKIP) grid = FGNN_RANGE_WG;
    if constexpr (DBG) {
        (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd_dbg_kernel<CA, CB, NMLP, DEPTH, PK, SKIP>, LDS);
        hipLaunchKernelGGL((mlp_fwd_dbg_kernel<CA, CB, NMLP, DEPTH, PK, SKIP>), dim3(grid), dim3(64 * NW), LDS, st, *a, tpg, total, d.m[0],
                           d.m[1]);
    } else {
        (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd_kernel<CA, CB, NMLP, DEPTH, PK, SKIP>, LDS);
        hipLaunchKernelGGL((mlp_fwd_kernel<CA, CB, NMLP, DEPTH, PK, SKIP>), dim3(grid), dim3(64 * NW), LDS, st, *a, tpg, total);
    }
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, int CB, int NMLP, int DEPTH, bool PK = false, bool DBG = false>
int launch_fwd(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st, const DbgOut &d) {
    if (a->ranges) return launch_fwd_impl<CA, CB, NMLP, DEPTH, PK, true, DBG>(a, tpg, total, st, d);
    return launch_fwd_impl<CA, CB, NMLP, DEPTH, PK, false, DBG>(a, tpg, total, st, d);
}

template <int NMLP, int DEPTH, bool DBG = false>
int dispatch_c(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st, const DbgOut &d = DbgOut{}) {
    const int ca = a->a.C, cb = a->b.C;
    if (a->xbits) {          // the 2-channel slab comes from the bit-packed adjacency (built for depth 3)
        if constexpr (DEPTH == 3) {
            if (ca == 2 && cb == 0) return launch_fwd<2, 0, NMLP, DEPTH, true, DBG>(a, tpg, total, st, d);
            if constexpr (NMLP == 1) {
                if (ca == 32 && cb == 2) return launch_fwd<32, 2, NMLP, DEPTH, true, DBG>(a, tpg, total, st, d);
            }
        }
        fgnn_set_error("fgnn_mlp_fwd: xbits needs depth 3 and a 2-channel slab (2 or 32+2 input channels), got depth %d, %d + %d",
                       DEPTH, ca, cb);
        return 1;
    }
#define FGNN_CASE(A_, B_) \
    if (ca == A_ && cb == B_) return launch_fwd<A_, B_, NMLP, DEPTH, false, DBG>(a, tpg, total, st, d);
    FGNN_CASE(2, 0)
    if constexpr (!DBG) {
        FGNN_CASE(16, 0)
    }
    FGNN_CASE(32, 0)
    if constexpr (NMLP == 1) {      // two-slab inputs (mlp3 = [mult ; in]) only come with a single MLP
        FGNN_CASE(32, 2)
        FGNN_CASE(32, 32)
    }
#undef FGNN_CASE
    fgnn_set_error("fgnn_mlp_fwd: unsupported input channels (%d + %d) for nmlp=%d; built for 2, 16, 32 and, with nmlp=1, 32+2, 32+32", ca, cb, NMLP);
    return 1;
}

}  // namespace

#ifdef FGNN_PHASES
extern "C" int fgnn_debug_fwd_stamps(void *p, int ca, int cb, int nmlp) {
    const int sel = ca * 1000 + cb * 10 + nmlp;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_sel), &sel, sizeof(sel)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stamp), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int fgnn_tiles_per_graph(int N) { return (N * N + FGNN_TILE - 1) / FGNN_TILE; }

static int mlp_fwd_entry(const fgnn_mlp_fwd_args *a, void *stream, unsigned *const *dbg) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_fwd: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_fwd: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(a->nmlp == 1 || a->nmlp == 2, "fgnn_mlp_fwd: nmlp must be 1 or 2 (got %d)", a->nmlp);
    FGNN_CHECK(a->depth >= 1 && a->depth <= FGNN_MAX_DEPTH, "fgnn_mlp_fwd: depth %d not in 1..%d", a->depth, FGNN_MAX_DEPTH);
    const bool pk_a = a->xbits && a->a.C == 2, pk_b = a->xbits && a->b.C == 2;     // that slab's memory is never touched
    FGNN_CHECK((a->a.ptr || pk_a) && a->a.C > 0, "fgnn_mlp_fwd: slab a missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr || pk_b, "fgnn_mlp_fwd: slab b has channels but no pointer");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_fwd: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK((long long)a->N * a->N <= a->ldz && (pk_a || (long long)a->N * a->N <= a->a.ldp), "fgnn_mlp_fwd: channel stride < N*N");
    for (int m = 0; m < a->nmlp; ++m) {
        FGNN_CHECK(a->z[m] && a->part[m], "fgnn_mlp_fwd: missing output %d", m);
        for (int l = 0; l < a->depth; ++l) FGNN_CHECK(a->W[m][l] && a->bias[m][l], "fgnn_mlp_fwd: missing weights mlp %d layer %d", m, l);
    }
    FGNN_CHECK(a->cnt, "fgnn_mlp_fwd: missing cnt");
    FGNN_CHECK(!a->ranges || a->nvalid, "fgnn_mlp_fwd: ranges (fgnn_ragged_tile_ranges) only make sense with nvalid");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * FGNN_H * a->ldz < lim,
                   "fgnn_mlp_fwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_fwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (dbg) {          // the decision-exporting twins exist for the fused engine's shapes: depth 3
        FGNN_CHECK(a->depth == 3 && dbg[0] && (a->nmlp == 1 || dbg[1]), "fgnn_debug_mlp_fwd_masks: depth 3 and one mask buffer per MLP");
        const DbgOut d = {{dbg[0], a->nmlp == 2 ? dbg[1] : nullptr}};
        if (a->nmlp == 1) return dispatch_c<1, 3, true>(a, tpg, (int)total, st, d);
        return dispatch_c<2, 3, true>(a, tpg, (int)total, st, d);
    }
#define FGNN_ND(NM_, D_) \
    if (a->nmlp == NM_ && a->depth == D_) return dispatch_c<NM_, D_>(a, tpg, (int)total, st);
    FGNN_ND(1, 1) FGNN_ND(1, 2) FGNN_ND(1, 3) FGNN_ND(2, 1) FGNN_ND(2, 2) FGNN_ND(2, 3)
#undef FGNN_ND
    fgnn_set_error("fgnn_mlp_fwd: unsupported nmlp/depth");
    return 1;
}

extern "C" int fgnn_mlp_fwd(const fgnn_mlp_fwd_args *a, void *stream) { return mlp_fwd_entry(a, stream, nullptr); }

// Test-only (tests/test_gpu_grad_pinned.py): fgnn_mlp_fwd once more -- same tile code, same outputs -- that also writes the ReLU
// decisions of the two hidden layers, masks[m]: (G, 2, 32, tiles per graph) words, bit j of word (g, layer, channel, t) = hidden
// pre-activation of pixel 32 t + j > 0.  Words of tiles a ragged launch steps over are not written.
extern "C" int fgnn_debug_mlp_fwd_masks(const fgnn_mlp_fwd_args *a, unsigned *masks0, unsigned *masks1, void *stream) {
    unsigned *const dbg[2] = {masks0, masks1};
    return mlp_fwd_entry(a, stream, dbg);
}
