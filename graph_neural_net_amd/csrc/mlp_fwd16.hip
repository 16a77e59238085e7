// bf16 variant of the fused MlpBlock_Real forward (models/layers.py:126-131, reductions of :72-73) for gfx950:
// activations stored as bf16, channel contraction on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 biases and
// fp32 GraphNorm tile statistics (the reference trains under 16-bit AMP, commander_explore.py:120-122).
//
// Work unit: a tile = 64 consecutive elements of every channel of one graph = 32 pixel pairs; lane (j, h) loads ONE
// dword (two bf16 pixels) per channel row, so every load instruction moves two 128-byte row segments.  The two pixels
// of a pair form two independent 32-column MFMA problems ("E" = even pixels, "O" = odd pixels) that share every
// weight operand.  As in the fp32 kernels the D fragment of a layer (lane = pixel, register = channel) packed
// pairwise IS the next layer's B operand, so the conv / ReLU chain never leaves the register file.  The tile
// statistics need a lane = channel view of z: it comes from the last layer evaluated a second time with the operand
// roles swapped (D' = h^T W^T: lane = channel, register = pixel) -- two more MFMAs per 32 pixels on an otherwise idle
// matrix pipe instead of an LDS transpose.
//
// Rounding points (mirrored by oracle/fgnn_oracle_bf16.py): operands R(W), R((z - mean) a + beta), R(relu(pre));
// outputs R(z); statistics {mean, M2} from the fp32 z.
#include "fgnn_bf16.h"

namespace {

constexpr int NWF = 8;       // waves per workgroup (2 per SIMD)

template <int CA, int CB, int NMLP, int DEPTH>
struct Fwd16Layout {
    static constexpr Pk16 PK = pk16_layout(0, CA, CB, DEPTH);
    static constexpr int MLP_F = PK.floats;
    static constexpr int WEIGHT_F = NMLP * MLP_F;
    static constexpr int REC_F = 2 * 32 * 2;                       // per wave: {a, b'} of slab a, slab b
    static constexpr int LDS_F = WEIGHT_F + NWF * REC_F;
};

// per-graph input records {a, b' = beta - mean * a} of one slab -> wave-private LDS (lanes 0..31, one channel each)
DEVI void fetch_rec(float *rec, const fgnn_slab16 &s, int g, int lane) {
    if (lane < 32) {
        float2 o = make_float2(1.f, 0.f);
        if (s.nrm && lane < s.C) {
            const float4 n = reinterpret_cast<const float4 *>(s.nrm)[(long long)g * s.C + lane];
            const float be = s.beta ? s.beta[lane] : 0.f;
            o.x = n.y;
            o.y = be - n.x * n.y;
        }
        reinterpret_cast<float2 *>(rec)[lane] = o;
    }
}

// loaded pixel-pair dwords of a 32-channel slab -> the two normal fragments (even / odd pixel), normalising on the way
DEVI void operands32(F16 &e, F16 &o, const unsigned (&x)[16], const float *rec, bool norm, int h) {
    if (norm) {
        const float2 *r2 = reinterpret_cast<const float2 *>(rec);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 n0 = r2[ch_of(2 * q, h)], n1 = r2[ch_of(2 * q + 1, h)];
            e.d[q] = cvt_pk(fmaf(bf_lo(x[2 * q]), n0.x, n0.y), fmaf(bf_lo(x[2 * q + 1]), n1.x, n1.y));
            o.d[q] = cvt_pk(fmaf(bf_hi(x[2 * q]), n0.x, n0.y), fmaf(bf_hi(x[2 * q + 1]), n1.x, n1.y));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            e.d[q] = pack_lo(x[2 * q], x[2 * q + 1]);
            o.d[q] = pack_hi(x[2 * q], x[2 * q + 1]);
        }
    }
}
// 2-channel slab (raw input): one zero-padded k-step, channels in slots 0, 1 of the h == 0 half (x is 0 for h == 1)
DEVI void operands2(i32x4 &e, i32x4 &o, const unsigned (&x)[2]) {
    e[0] = (int)pack_lo(x[0], x[1]);
    o[0] = (int)pack_hi(x[0], x[1]);
    e[1] = e[2] = e[3] = 0;
    o[1] = o[2] = o[3] = 0;
}

// SKIP (ragged batches with A.ranges): work-balanced tile range from fgnn_ragged_tile_ranges16, the waves step over tiles
// without a valid element; those only get empty statistics records (their z elements are not written: consumers step over
// the same tiles or read the valid corner only).
// DBG (fgnn_debug_mlp_fwd16_masks, test-only): the same tile code also exports the ReLU decisions it takes -- one bit per hidden
// pre-activation: dbg[m][((((g * (DEPTH-1) + layer) * 32 + channel) * tpg + tile) * 2 + parity], bit j = element 64 * tile + 2 j + parity
// of the (ldr-pitched) plane.
template <int CA, int CB, int NMLP, int DEPTH, bool SKIP, bool DBG>
DEVI void mlp_fwd16_body(const fgnn_mlp_fwd16_args A, const int tpg, const int total_tiles, unsigned *const dbg0, unsigned *const dbg1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Fwd16Layout<CA, CB, NMLP, DEPTH>;
    constexpr Pk16 PK = L::PK;
    constexpr int SA = pk16_steps(CA), SB = pk16_steps(CB);
    constexpr int XA = CA >= 32 ? 16 : 2, XB = CB >= 32 ? 16 : 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    young_prio(1, wv, blockDim.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int PP = A.N * A.ldr;
    float *wl = smem;
    float *recA = smem + L::WEIGHT_F + wv * L::REC_F, *recB = recA + 64;
    const View16 va = make_view16(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View16 vb = make_view16(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    View16 vz[NMLP];
#pragma unroll
    for (int m = 0; m < NMLP; ++m) vz[m] = make_view16(A.z[m], FGNN_H * A.ldz, A.ldz, A.G);

    const int nwg = gridDim.x;
    const int q_ = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * q_ + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + q_ + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const bool normA = A.a.nrm != nullptr, normB = (CB > 0) && A.b.nrm != nullptr;

    // operand image -> LDS (straight copy of the packed image)
    {
        const float4 *src = reinterpret_cast<const float4 *>(A.packed);
        float4 *dst = reinterpret_cast<float4 *>(wl);
        for (int e = threadIdx.x; e < L::WEIGHT_F / 4; e += 64 * NWF) dst[e] = src[e];
    }
    int tile = T0 + wv;
    if constexpr (SKIP) tile = __builtin_amdgcn_readfirstlane(next_live_tile_p(tile, T1, NWF, tpg, 64, A.ldr, A.nvalid));
    unsigned xa[XA], xb[CB > 0 ? XB : 1];
    int cached_g = -1, cur_nv = A.N;
    {
        const Tile16 c = decode16(tile, tile < T1, tpg, A.ldr, PP, j);
        load_slab16<CA>(xa, va, c, h);
        if constexpr (CB > 0) load_slab16<CB>(xb, vb, c, h);
        if (tile < T1) {
            fetch_rec(recA, A.a, c.g, lane);
            if constexpr (CB > 0) fetch_rec(recB, A.b, c.g, lane);
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
    }
    __syncthreads();

    while (tile < T1) {
        const Tile16 c = decode16(tile, true, tpg, A.ldr, PP, j);
        if (c.g != cached_g) {
            fetch_rec(recA, A.a, c.g, lane);
            if constexpr (CB > 0) fetch_rec(recB, A.b, c.g, lane);
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
        const bool v0 = c.inb && c.i < cur_nv && c.jj < cur_nv;
        const bool v1 = c.inb && c.i < cur_nv && c.jj + 1 < cur_nv;
        const unsigned mE = (unsigned)__ballot(v0), mO = (unsigned)__ballot(v1);     // bit jp = pixel pair jp valid
        const bool full = (mE & mO) == 0xffffffffu;
        const float cnt = (float)(__popc(mE) + __popc(mO));
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;

        // ---- input operands ----
        F16 yaE, yaO, ybE, ybO;
        i32x4 ya2E, ya2O, yb2E, yb2O;
        if constexpr (CA >= 32) operands32(yaE, yaO, xa, recA, normA, h);
        else operands2(ya2E, ya2O, xa);
        if constexpr (CB >= 32) operands32(ybE, ybO, xb, recB, normB, h);
        else if constexpr (CB > 0) operands2(yb2E, yb2O, xb);
        // prefetch the wave's next tile
        int next = tile + NWF;
        if constexpr (SKIP) next = __builtin_amdgcn_readfirstlane(next_live_tile_p(next, T1, NWF, tpg, 64, A.ldr, A.nvalid));
        {
            const Tile16 cn = decode16(next, next < T1, tpg, A.ldr, PP, j);
            load_slab16<CA>(xa, va, cn, h);
            if constexpr (CB > 0) load_slab16<CB>(xb, vb, cn, h);
        }

#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            const float *wm = wl + m * L::MLP_F;
            const float *tail = wm + PK.bias_f;
            f32x16 aE, aO;
            load_bias16(aE, tail, 0, h);
            aO = aE;
#pragma unroll
            for (int t = 0; t < SA; ++t) {
                const i32x4 w = lds_step(wm, PK.off_w0a + t, lane);
                if constexpr (CA >= 32) {
                    aE = mfma16(w, step_of(yaE, t), aE);
                    aO = mfma16(w, step_of(yaO, t), aO);
                } else {
                    aE = mfma16(w, ya2E, aE);
                    aO = mfma16(w, ya2O, aO);
                }
            }
#pragma unroll
            for (int t = 0; t < SB; ++t) {
                const i32x4 w = lds_step(wm, PK.off_w0b + t, lane);
                if constexpr (CB >= 32) {
                    aE = mfma16(w, step_of(ybE, t), aE);
                    aO = mfma16(w, step_of(ybO, t), aO);
                } else {
                    aE = mfma16(w, yb2E, aE);
                    aO = mfma16(w, yb2O, aO);
                }
            }
            F16 hE, hO;
            static_assert(DEPTH >= 2, "the swapped last layer takes a hidden fragment");
#pragma unroll
            for (int l = 1; l < DEPTH; ++l) {
                if constexpr (DBG) {      // the decision the packed ReLU takes: the sign of the fp32 pre-activation
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned long long mkE = __ballot(aE[r] > 0.f), mkO = __ballot(aO[r] > 0.f);
                        if (lane == 0) {
                            unsigned *o = (m == 0 ? dbg0 : dbg1) + ((((long long)c.g * (DEPTH - 1) + (l - 1)) * 32 + ch_of(r, 0)) * tpg + c.tt) * 2;
                            o[0] = (unsigned)mkE;
                            o[1] = (unsigned)mkO;
                            o[8ll * tpg] = (unsigned)(mkE >> 32);          // channel ch_of(r, 1) = ch_of(r, 0) + 4
                            o[8ll * tpg + 1] = (unsigned)(mkO >> 32);
                        }
                    }
                }
                pack_acc_relu(hE, aE);
                pack_acc_relu(hO, aO);
                load_bias16(aE, tail, l, h);
                aO = aE;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const i32x4 w = lds_step(wm, PK.off_wh + 2 * (l - 1) + t, lane);
                    aE = mfma16(w, step_of(hE, t), aE);
                    aO = mfma16(w, step_of(hO, t), aO);
                }
            }
            // the last layer once more with swapped operand roles: lane = channel j, register r <-> pixel pair ch_of(r, h)
            f32x16 tE, tO;
            {
                const float bl = tail[DEPTH * 32 + (DEPTH - 1) * 32 + j];
#pragma unroll
                for (int r = 0; r < 16; ++r) tE[r] = bl;
                tO = tE;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const i32x4 w = lds_step(wm, PK.off_wh + 2 * (DEPTH - 2) + t, lane);
                    tE = mfma16(step_of(hE, t), w, tE);
                    tO = mfma16(step_of(hO, t), w, tO);
                }
            }
            // ---- store z (bf16 pixel pairs), exact zeros in the padding ----
            {
                const int zoff = lane_off16<4>(vz[m], c, h);
                const int zs0 = c.g * vz[m].gs2;
                if (!full) {
                    const float f0 = v0 ? 1.f : 0.f, f1 = v1 ? 1.f : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        aE[r] *= f0;
                        aO[r] *= f1;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_u32(cvt_pk(aE[r], aO[r]), vz[m], zoff, zs0 + ((r & 3) + 8 * (r >> 2)) * vz[m].ld2);
            }
            // ---- tile statistics {mean, M2} of channel j over the valid pixels ----
            float s = 0.f, m2 = 0.f;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += tE[r] + tO[r];
                s += __shfl_xor(s, 32);
                const float mean = s * inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d0 = tE[r] - mean, d1 = tO[r] - mean;
                    m2 = fmaf(d0, d0, m2);
                    m2 = fmaf(d1, d1, m2);
                }
                s = mean;
            } else {
                float w0[16], w1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jp = ch_of(r, h);
                    w0[r] = ((mE >> jp) & 1u) ? 1.f : 0.f;
                    w1[r] = ((mO >> jp) & 1u) ? 1.f : 0.f;
                    s += tE[r] * w0[r] + tO[r] * w1[r];
                }
                s += __shfl_xor(s, 32);
                const float mean = s * inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d0 = (tE[r] - mean) * w0[r], d1 = (tO[r] - mean) * w1[r];
                    m2 = fmaf(d0, d0, m2);
                    m2 = fmaf(d1, d1, m2);
                }
                s = mean;
            }
            m2 += __shfl_xor(m2, 32);
            // (G, 32, tpg, 2): contiguous over the tiles of one channel -- every consumer walks a (g, c) column
            if (h == 0) reinterpret_cast<float2 *>(A.part[m])[((long long)c.g * FGNN_H + j) * tpg + c.tt] = make_float2(s, m2);
        }
        if (lane == 0) A.cnt[(long long)c.g * tpg + c.tt] = cnt;
        tile = next;
    }
    if constexpr (SKIP) {       // padding-only tiles of this wave's share: empty statistics records
        for (int t = T0 + wv; t < T1; t += NWF) {
            const int g = __builtin_amdgcn_readfirstlane(t / tpg), tt = t - g * tpg;
            if (tile_live_p(tt, 64, A.ldr, A.nvalid[g])) continue;
#pragma unroll
            for (int m = 0; m < NMLP; ++m) {
                if (h == 0) reinterpret_cast<float2 *>(A.part[m])[((long long)g * FGNN_H + j) * tpg + tt] = make_float2(0.f, 0.f);
            }
            if (lane == 0) A.cnt[(long long)g * tpg + tt] = 0.f;
        }
    }
}

template <int CA, int CB, int NMLP, int DEPTH, bool SKIP = false>
__global__ __launch_bounds__(64 * NWF, 2) void mlp_fwd16_kernel(const fgnn_mlp_fwd16_args A, const int tpg, const int total_tiles) {
    mlp_fwd16_body<CA, CB, NMLP, DEPTH, SKIP, false>(A, tpg, total_tiles, nullptr, nullptr);
}
template <int CA, int CB, int NMLP, int DEPTH>
__global__ __launch_bounds__(64 * NWF, 2) void mlp_fwd16_dbg_kernel(const fgnn_mlp_fwd16_args A, const int tpg, const int total_tiles,
                                                                     unsigned *d0, unsigned *d1) {
    mlp_fwd16_body<CA, CB, NMLP, DEPTH, false, true>(A, tpg, total_tiles, d0, d1);
}
template <int CA, int CB, int NMLP, int DEPTH>
int launch_fwd16_dbg(const fgnn_mlp_fwd16_args *a, int tpg, int total, hipStream_t st, unsigned *d0, unsigned *d1) {
    using L = Fwd16Layout<CA, CB, NMLP, DEPTH>;
    constexpr int LDS = L::LDS_F * 4;
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd16_dbg_kernel<CA, CB, NMLP, DEPTH>, LDS);
    int grid = (total + NWF - 1) / NWF;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL((mlp_fwd16_dbg_kernel<CA, CB, NMLP, DEPTH>), dim3(grid), dim3(64 * NWF), LDS, st, *a, tpg, total, d0, d1);
    FGNN_LAUNCH_CHECK();
    return 0;
}

template <int CA, int CB, int NMLP, int DEPTH, bool SKIP>
int launch_fwd16_impl(const fgnn_mlp_fwd16_args *a, int tpg, int total, hipStream_t st) {
    using L = Fwd16Layout<CA, CB, NMLP, DEPTH>;
    constexpr int LDS = L::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd16_kernel<CA, CB, NMLP, DEPTH, SKIP>, LDS);
    int grid = (total + NWF - 1) / NWF;
    if (grid > 256) grid = 256;
    if (SKIP) grid = FGNN_RANGE_WG;
    hipLaunchKernelGGL((mlp_fwd16_kernel<CA, CB, NMLP, DEPTH, SKIP>), dim3(grid), dim3(64 * NWF), LDS, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, int CB, int NMLP, int DEPTH>
int launch_fwd16(const fgnn_mlp_fwd16_args *a, int tpg, int total, hipStream_t st) {
    if (a->ranges) return launch_fwd16_impl<CA, CB, NMLP, DEPTH, true>(a, tpg, total, st);
    return launch_fwd16_impl<CA, CB, NMLP, DEPTH, false>(a, tpg, total, st);
}

}  // namespace

extern "C" int fgnn_tiles_per_graph16(int N, int ldr) { return (N * ldr + 63) / 64; }

static int mlp_fwd16_entry(const fgnn_mlp_fwd16_args *a, void *stream, unsigned *const *dbg) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_fwd16: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0 && a->ldr >= a->N && a->ldr % 8 == 0, "fgnn_mlp_fwd16: bad G=%d N=%d ldr=%d (ldr: multiple of 8, >= N)",
               a->G, a->N, a->ldr);
    FGNN_CHECK(a->nmlp == 1 || a->nmlp == 2, "fgnn_mlp_fwd16: nmlp must be 1 or 2 (got %d)", a->nmlp);
    FGNN_CHECK(a->depth == 3, "fgnn_mlp_fwd16: built for depth_of_mlp = 3 (got %d)", a->depth);
    FGNN_CHECK(a->a.ptr && a->a.C > 0 && a->packed, "fgnn_mlp_fwd16: slab a / operand image missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr, "fgnn_mlp_fwd16: slab b has channels but no pointer");
    const long long PP = (long long)a->N * a->ldr;
    FGNN_CHECK(PP <= a->ldz && PP <= a->a.ldp && a->ldz % 2 == 0 && a->a.ldp % 2 == 0, "fgnn_mlp_fwd16: channel stride < N*ldr or odd");
    for (int m = 0; m < a->nmlp; ++m) FGNN_CHECK(a->z[m] && a->part[m], "fgnn_mlp_fwd16: missing output %d", m);
    FGNN_CHECK(a->cnt, "fgnn_mlp_fwd16: missing cnt");
    FGNN_CHECK(!a->ranges || a->nvalid, "fgnn_mlp_fwd16: ranges (fgnn_ragged_tile_ranges16) only make sense with nvalid");
    {
        const long long lim = 0x7fffffffll / 2, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * FGNN_H * a->ldz < lim,
                   "fgnn_mlp_fwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph16(a->N, a->ldr);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_fwd16: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const int ca = a->a.C, cb = a->b.C;
    if (dbg) {          // decision-exporting twins: the fused engine's constant-size shapes
        FGNN_CHECK(!a->ranges && dbg[0] && (a->nmlp == 1 || dbg[1]), "fgnn_debug_mlp_fwd16_masks: no ranges, one mask buffer per MLP");
        if (a->nmlp == 2 && ca == 2 && cb == 0) return launch_fwd16_dbg<2, 0, 2, 3>(a, tpg, (int)total, st, dbg[0], dbg[1]);
        if (a->nmlp == 2 && ca == 32 && cb == 0) return launch_fwd16_dbg<32, 0, 2, 3>(a, tpg, (int)total, st, dbg[0], dbg[1]);
        if (a->nmlp == 1 && ca == 32 && cb == 2) return launch_fwd16_dbg<32, 2, 1, 3>(a, tpg, (int)total, st, dbg[0], nullptr);
        if (a->nmlp == 1 && ca == 32 && cb == 32) return launch_fwd16_dbg<32, 32, 1, 3>(a, tpg, (int)total, st, dbg[0], nullptr);
        fgnn_set_error("fgnn_debug_mlp_fwd16_masks: built for the fused engine's shapes (2 / 32 inputs with nmlp = 2; 32 + 2 / 32 + 32 with nmlp = 1)");
        return 1;
    }
    if (a->nmlp == 2 && ca == 2 && cb == 0) return launch_fwd16<2, 0, 2, 3>(a, tpg, (int)total, st);
    if (a->nmlp == 2 && ca == 32 && cb == 0) return launch_fwd16<32, 0, 2, 3>(a, tpg, (int)total, st);
    if (a->nmlp == 1 && ca == 32 && cb == 0) return launch_fwd16<32, 0, 1, 3>(a, tpg, (int)total, st);
    if (a->nmlp == 1 && ca == 2 && cb == 0) return launch_fwd16<2, 0, 1, 3>(a, tpg, (int)total, st);
    if (a->nmlp == 1 && ca == 32 && cb == 2) return launch_fwd16<32, 2, 1, 3>(a, tpg, (int)total, st);
    if (a->nmlp == 1 && ca == 32 && cb == 32) return launch_fwd16<32, 32, 1, 3>(a, tpg, (int)total, st);
    fgnn_set_error("fgnn_mlp_fwd16: unsupported input channels (%d + %d) for nmlp=%d; built for 2, 32 and, with nmlp=1, 32+2, 32+32",
                   ca, cb, a->nmlp);
    return 1;
}

extern "C" int fgnn_mlp_fwd16(const fgnn_mlp_fwd16_args *a, void *stream) { return mlp_fwd16_entry(a, stream, nullptr); }

// Test-only (tests/test_gpu_grad_pinned.py, bf16 case): fgnn_mlp_fwd16 once more -- same tile code, same outputs -- that also writes the
// ReLU decisions of the two hidden layers: masks[m] is (G, 2, 32, tiles per graph, 2) words, bit j of word (g, layer, channel, t, parity)
// = hidden pre-activation of element 64 t + 2 j + parity of the ldr-pitched plane > 0.  Constant-size batches.
extern "C" int fgnn_debug_mlp_fwd16_masks(const fgnn_mlp_fwd16_args *a, unsigned *masks0, unsigned *masks1, void *stream) {
    unsigned *const dbg[2] = {masks0, masks1};
    return mlp_fwd16_entry(a, stream, dbg);
}
