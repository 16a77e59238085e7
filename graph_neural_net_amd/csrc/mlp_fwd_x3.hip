// Fused MlpBlock_Real forward (models/layers.py:126-131: conv1x1 + ReLU chain, last conv linear) with the channel
// contraction on the bf16 matrix cores through the exact three-way operand split of fgnn_x3.h.
//
// Same interface, tile geometry, slab layout, statistics epilogue and results (up to fp32 reassociation noise) as
// mlp_fwd.hip; what changes is the inside of the conv chain: the D fragment of a layer (fp32, lane = pixel, register r
// <-> channel ch_of(r, h)) is ReLU'd, split into three packed bf16 fragments (which ARE the B operand of the next
// layer's v_mfma_f32_32x32x16_bf16, fgnn_bf16.h) and multiplied with the pre-split weight images from LDS: 12 bf16 MFMAs
// (384 matrix-pipe cycles, running beside the VALU) replace 16 fp32 MFMAs (1024 cycles that block the SIMD's fp32 VALU).
// Built for the reference's depth 3 and input slabs of 2 / 32 / 32+2 / 32+32 channels; constant-size batches.
#include "fgnn_tile.h"
#include "fgnn_pack.h"
#include "fgnn_x3.h"

namespace {

template <int CA, int CB, int NMLP, int NWT>
struct FwdX3Layout {
    static constexpr int DEPTH = 3;
    static constexpr int NW = NWT;
    static constexpr PkX3 PK = pkx3_layout(0, CA, CB, DEPTH);
    static constexpr int PD = PK.part_dw;
    static constexpr int OFF_W0A = PK.p.off_w0a, OFF_W0B = PK.p.off_w0b, OFF_WH = PK.p.off_wh;
    static constexpr int STEPS_A = pk16_steps(CA), STEPS_B = pk16_steps(CB);
    static constexpr int BIAS_F = PK.bias_off;
    static constexpr int MLP_F = PK.floats;
    static constexpr int WEIGHT_F = NMLP * MLP_F;
    static constexpr int REC_F = 2 * 32 * 4;
    static constexpr int LDS_F = WEIGHT_F + NW * (TILE_F + REC_F);
};

// the three bf16 parts of an input slab as B operand: 32 channels = the lane's 16 values in fragment order; 2 channels = one
// zero-padded k-step whose slots 0, 1 (half-wave 0) are the two channels
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

// DBG (fgnn_debug_mlp_fwd_x3_masks, test-only): also exports the ReLU decisions, in the layout of fgnn_debug_mlp_fwd_masks
template <int CA, int CB, int NMLP, bool PK, int NWT, bool DBG>
DEVI void mlp_fwd_x3_body(const fgnn_mlp_fwd_args A, const int tpg, const int total_tiles, unsigned *const dbg0, unsigned *const dbg1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = FwdX3Layout<CA, CB, NMLP, NWT>;
    constexpr int NW = NWT, DEPTH = 3;
    constexpr int SA = CA / 2, SB = CB / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    const int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);

    const bool normA = A.a.nrm != nullptr, normB = (CB > 0) && A.b.nrm != nullptr;
    // one memory round trip: the operand image is requested first (into registers), then the first tile and its records
    PkRegs<L::WEIGHT_F / 4, 64 * NW> img;
    pk_load_regs(img, A.packed);
    __builtin_amdgcn_sched_barrier(0);
    int tile = T0 + wv;
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
    pk_store_regs(wl, img);
    if (lane < 32) {
        reinterpret_cast<float4 *>(recA)[lane] = ra;
        reinterpret_cast<float4 *>(recB)[lane] = rb;
    }
    __syncthreads();
    const F16 negI = make_neg_identity(lane);

    while (tile < T1) {
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
        // normalise + split this tile's input, then prefetch the wave's next tile into the freed registers
        X3 XA, XB;
        {
            float ya[SA > 0 ? SA : 1], yb[SB > 0 ? SB : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            norm_slab<SB>(yb, xb, recB, normB, c_valid, h);
            split_slab<SA>(XA, ya, h, negI);
            if constexpr (SB > 0) split_slab<SB>(XB, yb, h, negI);
        }
        const int next = tile + NW;
        {
            const TileCtx cn = decode_tile(next, next < T1, tpg, A.N, P, j);
            load_slab<SA, PK>(xa, va, ps, cn, h);
            load_slab<SB, PK>(xb, vb, ps, cn, h);
        }

        const unsigned vmask = (unsigned)__ballot(c_valid);      // bit px = pixel valid (low half-wave)
        const float cnt = (float)__popc(vmask);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            const float *wm = wl + m * L::MLP_F;
            f32x16 acc;
            load_bias16(acc, wm + L::BIAS_F, 0, h);
            acc = gemm_x3<L::STEPS_A, X3_FWD_TERMS>(acc, wm, L::PD, L::OFF_W0A, XA, lane);
            if constexpr (SB > 0) acc = gemm_x3<L::STEPS_B, X3_FWD_TERMS>(acc, wm, L::PD, L::OFF_W0B, XB, lane);
#pragma unroll
            for (int l = 1; l < DEPTH; ++l) {
                X3 H;
                {
                    float hid[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) hid[r] = relu1(acc[r]);
                    if constexpr (DBG) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned long long mk = __ballot(__float_as_int(acc[r]) > 0);
                            if (lane == 0) {
                                unsigned *o = (m == 0 ? dbg0 : dbg1) + (((long long)c.g * (DEPTH - 1) + (l - 1)) * 32 + ch_of(r, 0)) * tpg + c.tt;
                                o[0] = (unsigned)mk;
                                o[4ll * tpg] = (unsigned)(mk >> 32);
                            }
                        }
                    }
                    split16m(H, hid, negI);
                }
                load_bias16(acc, wm + L::BIAS_F, l, h);
                acc = gemm_x3<2, X3_FWD_TERMS>(acc, wm, L::PD, L::OFF_WH + 2 * (l - 1), H, lane);
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
        tile = next;
    }
}

template <int CA, int CB, int NMLP, bool PK, int NWT>
__global__ __launch_bounds__(64 * NWT, NWT / 4) void mlp_fwd_x3_kernel(const fgnn_mlp_fwd_args A, const int tpg,
                                                                      const int total_tiles) {
    mlp_fwd_x3_body<CA, CB, NMLP, PK, NWT, false>(A, tpg, total_tiles, nullptr, nullptr);
}
template <int CA, int CB, int NMLP, bool PK, int NWT>
__global__ __launch_bounds__(64 * NWT, NWT / 4) void mlp_fwd_x3_dbg_kernel(const fgnn_mlp_fwd_args A, const int tpg,
                                                                          const int total_tiles, unsigned *d0, unsigned *d1) {
    mlp_fwd_x3_body<CA, CB, NMLP, PK, NWT, true>(A, tpg, total_tiles, d0, d1);
}

template <int CA, int CB, int NMLP, bool PK, int NWT, bool DBG = false>
int launch_fwd_x3(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st, unsigned *d0 = nullptr, unsigned *d1 = nullptr) {
    using L = FwdX3Layout<CA, CB, NMLP, NWT>;
    constexpr int LDS = L::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    int grid = (total + NWT - 1) / NWT;
    const int cap = a->cu_share == 2 ? 128 : 256;
    if (grid > cap) grid = cap;
    if constexpr (DBG) {
        (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd_x3_dbg_kernel<CA, CB, NMLP, PK, NWT>, LDS);
        hipLaunchKernelGGL((mlp_fwd_x3_dbg_kernel<CA, CB, NMLP, PK, NWT>), dim3(grid), dim3(64 * NWT), LDS, st, *a, tpg, total, d0, d1);
    } else {
        (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd_x3_kernel<CA, CB, NMLP, PK, NWT>, LDS);
        hipLaunchKernelGGL((mlp_fwd_x3_kernel<CA, CB, NMLP, PK, NWT>), dim3(grid), dim3(64 * NWT), LDS, st, *a, tpg, total);
    }
    FGNN_LAUNCH_CHECK();
    return 0;
}

constexpr int FWD_X3_WAVES = 16;

}  // namespace

extern "C" int fgnn_mlp_x3_supported(int ca, int cb, int depth, int nmlp) {
    if (depth != 3) return 0;
    if (nmlp == 2) return (cb == 0 && (ca == 2 || ca == 32)) ? 1 : 0;
    return ((ca == 2 || ca == 32) && cb == 0) || (ca == 32 && (cb == 2 || cb == 32)) ? 1 : 0;
}

static int mlp_fwd_x3_entry(const fgnn_mlp_fwd_args *a, void *stream, unsigned *const *dbg) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_fwd_x3: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_fwd_x3: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(a->nmlp == 1 || a->nmlp == 2, "fgnn_mlp_fwd_x3: nmlp must be 1 or 2 (got %d)", a->nmlp);
    FGNN_CHECK(fgnn_mlp_x3_supported(a->a.C, a->b.C, a->depth, a->nmlp),
               "fgnn_mlp_fwd_x3: built for depth 3 and 2, 32, 32+2, 32+32 input channels (got depth %d, %d + %d, nmlp %d); use fgnn_mlp_fwd",
               a->depth, a->a.C, a->b.C, a->nmlp);
    FGNN_CHECK(a->packed, "fgnn_mlp_fwd_x3: needs the operand image of fgnn_pack_x3_operands");
    FGNN_CHECK(!a->ranges, "fgnn_mlp_fwd_x3: no padding-tile skipping (ranges); use fgnn_mlp_fwd for ragged batches");
    const bool pk_a = a->xbits && a->a.C == 2, pk_b = a->xbits && a->b.C == 2;
    FGNN_CHECK((a->a.ptr || pk_a) && (a->b.C == 0 || a->b.ptr || pk_b), "fgnn_mlp_fwd_x3: slab pointer missing");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_fwd_x3: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK((long long)a->N * a->N <= a->ldz && (pk_a || (long long)a->N * a->N <= a->a.ldp), "fgnn_mlp_fwd_x3: channel stride < N*N");
    for (int m = 0; m < a->nmlp; ++m) FGNN_CHECK(a->z[m] && a->part[m], "fgnn_mlp_fwd_x3: missing output %d", m);
    FGNN_CHECK(a->cnt, "fgnn_mlp_fwd_x3: missing cnt");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * FGNN_H * a->ldz < lim,
                   "fgnn_mlp_fwd_x3: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_fwd_x3: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const int ca = a->a.C, cb = a->b.C;
    constexpr int W = FWD_X3_WAVES;
    if (dbg) {          // decision-exporting twins of the launches the engine issues (mlp1 + mlp2 of a block)
        FGNN_CHECK(a->nmlp == 2 && dbg[0] && dbg[1], "fgnn_debug_mlp_fwd_x3_masks: built for the two-MLP launches (mlp1 + mlp2)");
        if (a->xbits) return launch_fwd_x3<2, 0, 2, true, W, true>(a, tpg, (int)total, st, dbg[0], dbg[1]);
        if (ca == 2) return launch_fwd_x3<2, 0, 2, false, W, true>(a, tpg, (int)total, st, dbg[0], dbg[1]);
        return launch_fwd_x3<32, 0, 2, false, W, true>(a, tpg, (int)total, st, dbg[0], dbg[1]);
    }
    if (a->xbits) {
        if (a->nmlp == 2 && ca == 2) return launch_fwd_x3<2, 0, 2, true, W>(a, tpg, (int)total, st);
        if (a->nmlp == 1 && ca == 32 && cb == 2) return launch_fwd_x3<32, 2, 1, true, W>(a, tpg, (int)total, st);
        if (a->nmlp == 1 && ca == 2) return launch_fwd_x3<2, 0, 1, true, W>(a, tpg, (int)total, st);
        fgnn_set_error("fgnn_mlp_fwd_x3: xbits needs a 2-channel slab");
        return 1;
    }
    if (a->nmlp == 2) {
        if (ca == 2) return launch_fwd_x3<2, 0, 2, false, W>(a, tpg, (int)total, st);
        return launch_fwd_x3<32, 0, 2, false, W>(a, tpg, (int)total, st);
    }
    if (cb == 0) {
        if (ca == 2) return launch_fwd_x3<2, 0, 1, false, W>(a, tpg, (int)total, st);
        return launch_fwd_x3<32, 0, 1, false, W>(a, tpg, (int)total, st);
    }
    if (cb == 2) return launch_fwd_x3<32, 2, 1, false, W>(a, tpg, (int)total, st);
    return launch_fwd_x3<32, 32, 1, false, W>(a, tpg, (int)total, st);
}

extern "C" int fgnn_mlp_fwd_x3(const fgnn_mlp_fwd_args *a, void *stream) { return mlp_fwd_x3_entry(a, stream, nullptr); }

// Test-only twin of fgnn_debug_mlp_fwd_masks for the x3 forward (same outputs + the ReLU decisions of the two hidden layers)
extern "C" int fgnn_debug_mlp_fwd_x3_masks(const fgnn_mlp_fwd_args *a, unsigned *masks0, unsigned *masks1, void *stream) {
    unsigned *const dbg[2] = {masks0, masks1};
    return mlp_fwd_x3_entry(a, stream, dbg);
}

// ---- operand images of the x3 kernels (once per step, like fgnn_pack_operands) -----------------------------------------
namespace {
struct PackX3Jobs {
    fgnn_pack_job job[FGNN_MAX_PACK_JOBS];
};
// grid (blocks per job, njobs): one thread per (step, lane) splits the lane's 8 weights and writes 4 dwords to each of the
// three part-images; the bias tail is plain fp32 in the compact [layer][h][16] order
__global__ __launch_bounds__(256) void pack_x3_kernel(const PackX3Jobs J) {
    const fgnn_pack_job &jb = J.job[blockIdx.y];
    if (jb.kind >= 2) {          // an fp32-MFMA image (the layout of fgnn_pack_operands) riding in the same launch
        if (jb.kind == 2) {
            const PkFwd p = pk_fwd(jb.ca, jb.cb, jb.depth);
            const int per = p.floats;
            for (int e = blockIdx.x * 256 + threadIdx.x; e < per * jb.nmlp; e += gridDim.x * 256) {
                const int m = e / per, r = e - m * per;
                if (r < p.bias_f) {
                    const int t = r >> 6, l = r & 63;
                    jb.out[m * per + (t >> 2) * 256 + l * 4 + (t & 3)] = pk_fwd_value(p, jb.ca, jb.cb, jb.W[m], t, l);
                } else {
                    jb.out[m * per + r] = pk_bias_value(jb.bias[m], r - p.bias_f);
                }
            }
        } else {
            const PkBwd p = pk_bwd(jb.ca, jb.cb, jb.depth);
            for (int e = blockIdx.x * 256 + threadIdx.x; e < p.floats; e += gridDim.x * 256) {
                if (e < p.bias_f) {
                    const int t = e >> 6, l = e & 63;
                    jb.out[(t >> 2) * 256 + l * 4 + (t & 3)] = pk_bwd_value(p, jb.ca, jb.cb, jb.W[0], t, l);
                } else {
                    jb.out[e] = pk_bias_value(jb.bias[0], e - p.bias_f);
                }
            }
        }
        return;
    }
    const PkX3 x = pkx3_layout(jb.kind, jb.ca, jb.cb, jb.depth);
    const Pk16 &p = x.p;
    const int nm = jb.kind == 0 ? jb.nmlp : 1;
    for (int m = 0; m < nm; ++m) {
        unsigned *om = reinterpret_cast<unsigned *>(jb.out) + (long long)m * x.floats;
        const float *const *W = jb.W[m];
        const float *const *Bv = jb.bias[m];
        for (int e = blockIdx.x * 256 + threadIdx.x; e < p.steps * 64; e += gridDim.x * 256) {
            const int step = e >> 6, l = e & 63;
            float w8[8];
            pk16_values8(jb.kind, p, jb.ca, jb.cb, jb.depth, W, step, l, w8);
            unsigned d[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split_pair(w8[2 * q], w8[2 * q + 1], d[0][q], d[1][q], d[2][q]);
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                uint4 v;
                v.x = d[part][0];
                v.y = d[part][1];
                v.z = d[part][2];
                v.w = d[part][3];
                reinterpret_cast<uint4 *>(om + part * x.part_dw)[e] = v;
            }
        }
        float *tail = reinterpret_cast<float *>(om + x.bias_off);
        for (int e = blockIdx.x * 256 + threadIdx.x; e < 32 * p.nbias; e += gridDim.x * 256) {
            const int layer = e >> 5, r = e & 15, h = (e >> 4) & 1;
            tail[e] = Bv[layer][pk16_ch(r, h)];
        }
    }
}
}  // namespace

extern "C" int fgnn_pack_x3_floats(int kind, int ca, int cb, int depth, int nmlp) {
    if (kind >= 2) return fgnn_pack_floats(kind - 2, ca, cb, depth, nmlp);
    return pkx3_layout(kind, ca, cb, depth).floats * (kind == 0 ? nmlp : 1);
}

extern "C" int fgnn_pack_x3_operands(const fgnn_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(jobs && njobs > 0 && njobs <= FGNN_MAX_PACK_JOBS, "fgnn_pack_x3_operands: bad arguments (njobs=%d)", njobs);
    PackX3Jobs J;
    for (int i = 0; i < njobs; ++i) {
        FGNN_CHECK(jobs[i].out && jobs[i].kind >= 0 && jobs[i].kind <= 3 && (jobs[i].nmlp == 1 || jobs[i].nmlp == 2),
                   "fgnn_pack_x3_operands: job %d malformed", i);
        if (jobs[i].kind < 2) {
            FGNN_CHECK(jobs[i].depth == 3, "fgnn_pack_x3_operands: job %d: x3 images are built for depth 3", i);
            FGNN_CHECK((jobs[i].ca == 2 || jobs[i].ca == 32) && (jobs[i].cb == 0 || jobs[i].cb == 2 || jobs[i].cb == 32),
                       "fgnn_pack_x3_operands: job %d: slab widths must be 2 or 32", i);
        }
        J.job[i] = jobs[i];
    }
    hipLaunchKernelGGL(pack_x3_kernel, dim3(32, njobs), dim3(256), 0, (hipStream_t)stream, J);
    FGNN_LAUNCH_CHECK();
    return 0;
}
