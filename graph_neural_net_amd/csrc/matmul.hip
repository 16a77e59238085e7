// Per-channel N x N matrix products (Matmul, models/layers.py:161-162) and their
// backward for gfx950.  One workgroup (4 waves) owns one 64x64 output tile of one (g,c)
// matrix: operands are normalised on load ((z-mean)*a+beta, padding -> 0), staged in LDS
// with a 65-float row stride (conflict-free column reads), and multiplied with
// v_mfma_f32_32x32x2_f32, one 32x32 quadrant per wave.
//
//   forward : M  = Ya @ Yb
//   backward: dA = dM @ Yb^T,   dB = Ya^T @ dM
#include <cstdlib>
#include "fgnn_common.h"
#include "fgnn_norm.h"

namespace {

constexpr int TM = 64;        // output tile edge
constexpr int LDS_LD = 65;    // LDS row stride (floats)

struct NormRec {
    float mean, a, beta;
    bool on;
};

DEVI NormRec norm_of(const fgnn_slab &s, int g, int c) {
    NormRec r;
    r.on = s.nrm != nullptr;
    if (r.on) {
        const float4 n = reinterpret_cast<const float4 *>(s.nrm)[(long long)g * s.C + c];
        r.mean = n.x;
        r.a = n.y;
        r.beta = s.beta ? s.beta[c] : 0.f;
    } else {
        r.mean = 0.f;
        r.a = 1.f;
        r.beta = 0.f;
    }
    return r;
}

// Stage the [row0,row0+64) x [col0,col0+64) window of an N x N matrix into LDS (zero fill
// outside the valid nv x nv region), optionally normalising.
DEVI void stage_tile(float *lds, const float *mat, int N, int nv, int row0, int col0, const NormRec &nr, int tid) {
    if (row0 == 0 && col0 == 0 && N <= TM) {
        // whole matrix is one tile: walk it linearly (fully coalesced)
        for (int e = tid; e < TM * TM; e += 256) lds[(e >> 6) * LDS_LD + (e & 63)] = 0.f;
        __syncthreads();
        const int P = N * N;
        for (int e = tid; e < P; e += 256) {
            const int r = e / N, c = e - r * N;
            float v = mat[e];
            const bool ok = r < nv && c < nv;
            v = ok ? (nr.on ? (v - nr.mean) * nr.a + nr.beta : v) : 0.f;
            lds[r * LDS_LD + c] = v;
        }
    } else {
        for (int e = tid; e < TM * TM; e += 256) {
            const int r = e >> 6, c = e & 63;
            const int gr = row0 + r, gc = col0 + c;
            float v = 0.f;
            if (gr < nv && gc < nv) {
                v = mat[(long long)gr * N + gc];
                if (nr.on) v = (v - nr.mean) * nr.a + nr.beta;
            }
            lds[r * LDS_LD + c] = v;
        }
    }
}

// D-fragment store of one 32x32 quadrant.
DEVI void store_quadrant(float *out, int N, int row0, int col0, const f32x16 &acc, int lane) {
    const int j = lane & 31, h = lane >> 5;
    const int col = col0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + ch_of(r, h);
        if (row < N && col < N) out[(long long)row * N + col] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------
// forward: grid (tiles_n, tiles_m, G*C)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_matmul_fwd_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                              const int *nvalid, int N, float *out,
                                                              long long ogstride, long long ldo) {
    __shared__ float As[TM * LDS_LD];
    __shared__ float Bs[TM * LDS_LD];
    const int C = ya.C;
    const int gc = blockIdx.z;
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int j = lane & 31, h = lane >> 5;
    const float *A = ya.ptr + (long long)g * ya.gstride + (long long)c * ya.ldp;
    const float *B = yb.ptr + (long long)g * yb.gstride + (long long)c * yb.ldp;
    const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
    float *O = out + (long long)g * ogstride + (long long)c * ldo;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool live = (row0 + 32 * qi < N) && (col0 + 32 * qj < N);

    for (int k0 = 0; k0 < N; k0 += TM) {
        if (k0) __syncthreads();
        stage_tile(As, A, N, nv, row0, k0, na, tid);
        stage_tile(Bs, B, N, nv, k0, col0, nb, tid);
        __syncthreads();
        if (live) {
            const int kmax = (N - k0 < TM ? N - k0 : TM);
            const float *ap = As + (32 * qi + j) * LDS_LD + h;
            const float *bp = Bs + h * LDS_LD + 32 * qj + j;
            for (int k = 0; k < kmax; k += 2) acc = mfma32(ap[k], bp[k * LDS_LD], acc);
        }
    }
    if (live) store_quadrant(O, N, row0 + 32 * qi, col0 + 32 * qj, acc, lane);
}

// ---------------------------------------------------------------------------------------
// backward: grid (tiles, tiles, G*C); each workgroup produces the (ti,tj) tile of dA and of dB.
//   dA[i][k] = sum_j dM[i][j] Yb[k][j]      dB[k][j] = sum_i Ya[i][k] dM[i][j]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_matmul_bwd_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                              const float *dm, long long dmg, long long ldm,
                                                              const int *nvalid, int N, float *da, float *db,
                                                              long long ogstride, long long ldo, float *s12a,
                                                              float *s12b) {
    __shared__ float Xs[TM * LDS_LD];
    __shared__ float Ds[TM * LDS_LD];
    __shared__ float red[4][4];
    const int C = ya.C;
    const int gc = blockIdx.z;
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int j = lane & 31, h = lane >> 5;
    const float *A = ya.ptr + (long long)g * ya.gstride + (long long)c * ya.ldp;
    const float *B = yb.ptr + (long long)g * yb.gstride + (long long)c * yb.ldp;
    const float *D = dm + (long long)g * dmg + (long long)c * ldm;
    const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
    NormRec none;
    none.on = false; none.mean = 0.f; none.a = 1.f; none.beta = 0.f;
    float *OA = da + (long long)g * ogstride + (long long)c * ldo;
    float *OB = db + (long long)g * ogstride + (long long)c * ldo;
    const bool live = (row0 + 32 * qi < N) && (col0 + 32 * qj < N);

    // ---- dA tile (rows row0.., cols col0..): A-operand = dM[row0+i][j0+..], B-operand = Yb[col0+k][j0+..]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int j0 = 0; j0 < N; j0 += TM) {
        if (j0) __syncthreads();
        stage_tile(Ds, D, N, nv, row0, j0, none, tid);
        stage_tile(Xs, B, N, nv, col0, j0, nb, tid);
        __syncthreads();
        if (live) {
            const int kmax = (N - j0 < TM ? N - j0 : TM);
            const float *ap = Ds + (32 * qi + j) * LDS_LD + h;
            const float *bp = Xs + (32 * qj + j) * LDS_LD + h;
            for (int k = 0; k < kmax; k += 2) acc = mfma32(ap[k], bp[k], acc);
        }
    }
    if (live) store_quadrant(OA, N, row0 + 32 * qi, col0 + 32 * qj, acc, lane);
    // GraphNorm-backward sums of the producer of Ya: S1 = sum dA, S2 = sum dA * (z_a - mean_a)
    float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
    if (s12a && live) {
        const int col = col0 + 32 * qj + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + 32 * qi + ch_of(r, h);
            if (row < nv && col < nv) {
                sa1 += acc[r];
                sa2 += acc[r] * (A[(long long)row * N + col] - na.mean);
            }
        }
    }

    // ---- dB tile (rows row0.., cols col0..): A-operand = Ya^T[row0+k'][i0+..] = Ya[i0+..][row0+k'],
    //      B-operand = dM[i0+..][col0+j]
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i0 = 0; i0 < N; i0 += TM) {
        __syncthreads();
        stage_tile(Xs, A, N, nv, i0, row0, na, tid);
        stage_tile(Ds, D, N, nv, i0, col0, none, tid);
        __syncthreads();
        if (live) {
            const int kmax = (N - i0 < TM ? N - i0 : TM);
            const float *ap = Xs + h * LDS_LD + 32 * qi + j;
            const float *bp = Ds + h * LDS_LD + 32 * qj + j;
            for (int k = 0; k < kmax; k += 2) acc = mfma32(ap[k * LDS_LD], bp[k * LDS_LD], acc);
        }
    }
    if (live) store_quadrant(OB, N, row0 + 32 * qi, col0 + 32 * qj, acc, lane);
    if (s12a) {     // only launched with a 1x1 tile grid (whole matrix in this workgroup)
        if (live) {
            const int col = col0 + 32 * qj + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 32 * qi + ch_of(r, h);
                if (row < nv && col < nv) {
                    sb1 += acc[r];
                    sb2 += acc[r] * (B[(long long)row * N + col] - nb.mean);
                }
            }
        }
        sa1 = wave_sum(sa1);
        sa2 = wave_sum(sa2);
        sb1 = wave_sum(sb1);
        sb2 = wave_sum(sb2);
        if (lane == 0) {
            red[wave][0] = sa1;
            red[wave][1] = sa2;
            red[wave][2] = sb1;
            red[wave][3] = sb2;
        }
        __syncthreads();
        if (tid < 4) {
            const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            float *dst = (tid < 2 ? s12a : s12b) + (long long)gc * 2 + (tid & 1);
            *dst = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Single-tile fast path (N <= 64, i.e. the whole matrix is one 64x64 tile): every operand is
// staged exactly once -- all global loads of the workgroup are issued back to back into
// registers, written to LDS, ONE barrier -- and the four waves then run their MFMA chains
// and epilogues independently.
// ---------------------------------------------------------------------------------------
// Unconditional (address-clamped) loads + selects: a predicated load would become one branch
// region per element, each waiting for its own round trip.
// v = operand value (normalised, 0 in padding); u = z - mean (0 in padding) when WANT_U.
template <bool WANT_U>
DEVI void load_tile_regs(float (&v)[16], float (&u)[16], const float *mat, int N, int nv, const NormRec &nr, int tid) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = tid + 256 * k;
        const int r = e >> 6, c = e & 63;
        const bool ok = r < nv && c < nv;
        const float x = mat[ok ? r * N + c : 0];
        const float d = x - nr.mean;
        v[k] = ok ? (nr.on ? d * nr.a + nr.beta : x) : 0.f;
        if (WANT_U) u[k] = ok ? d : 0.f;
    }
}
DEVI void store_tile_lds(float *lds, const float (&v)[16], int tid) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = tid + 256 * k;
        lds[(e >> 6) * LDS_LD + (e & 63)] = v[k];
    }
}
// D fragment of one quadrant -> LDS tile (row stride LDS_LD)
DEVI void frag_to_lds(float *lds, int row0, int col0, const f32x16 &acc, int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[(row0 + ch_of(r, h)) * LDS_LD + col0 + j] = acc[r];
}
// LDS tile -> global (coalesced rows, buffer stores: out-of-matrix lanes use OOB_OFF and are dropped),
// optionally accumulating S1 = sum t, S2 = sum t * u
template <bool WANT_S>
DEVI void tile_to_global(const View &ov, int mat_off4, const float *lds, int N, const float (&u)[16], float &s1,
                         float &s2, int tid) {
    const int c = tid & 63, r0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int voff = c < N ? (r0 * N + c) * 4 : OOB_OFF;
    const float *lp = lds + r0 * LDS_LD + c;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float t = lp[4 * k * LDS_LD];
        buf_store(t, ov, r0 + 4 * k < N ? voff : OOB_OFF, mat_off4 + k * 16 * N);
        if (WANT_S) {
            s1 += t;
            s2 += t * u[k];
        }
    }
}

// raw (un-normalised) loads of one matrix: issued early, consumed one iteration later.
// Buffer addressing: one VGPR offset per thread, the matrix / row-block offsets in SGPRs; padding
// lanes use OOB_OFF (load returns 0) -- no 64-bit address registers, no branches.
DEVI void load_tile_raw(float (&x)[16], const View &v, int mat_off4, int N, int nv, int tid) {
    const int c = tid & 63, r0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int voff = c < nv ? (r0 * N + c) * 4 : OOB_OFF;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + 4 * k;                      // wave-uniform
        x[k] = buf_load(v, r < nv ? voff : OOB_OFF, mat_off4 + k * 16 * N);
    }
}
// raw -> operand value v (normalised, 0 in padding) and optionally u = z - mean
template <bool WANT_U>
DEVI void finish_tile(float (&v)[16], float (&u)[16], const float (&x)[16], int nv, const NormRec &nr, int tid) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = tid + 256 * k;
        const int r = e >> 6, c = e & 63;
        const bool ok = r < nv && c < nv;
        const float d = x[k] - nr.mean;
        v[k] = ok ? (nr.on ? d * nr.a + nr.beta : x[k]) : 0.f;
        if (WANT_U) u[k] = ok ? d : 0.f;
    }
}

// One workgroup per (g,c) matrix.  (A persistent, software-pipelined variant was measured slower:
// the kernel is instruction-issue bound, not load-latency bound, and the prefetch registers spilled.)
// FIN: the GraphNorm records of both operands do not exist yet -- the workgroup finalizes them itself from
// the tile statistics its producer (the mlp1 + mlp2 forward launch) left behind, while its tile loads are in
// flight (wave 0: operand a, wave 1: operand b), and publishes them for every later consumer.  This is the
// work of fgnn_gn_finalize2, without its launch.
struct FinArgs {
    const float *part_a, *part_b, *cnt, *gw_a, *gw_b;
    float *nrm_a, *nrm_b;
    float eps;
    int tpg;
};
template <bool FIN>
__global__ __launch_bounds__(256, 4) void chan_matmul_fwd1_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                               const int *nvalid, int N, int M, float *out,
                                                               long long ogstride, long long ldo, const FinArgs F) {
    __shared__ float As[TM * LDS_LD];
    __shared__ float Bs[TM * LDS_LD];
    __shared__ float4 fin_rec[2];
    const int C = ya.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int j = lane & 31, h = lane >> 5;
    float xa[16], xb[16], dummy[16];
    const int G = M / C;
    const View vA = make_view(ya.ptr, ya.gstride, ya.ldp, G), vB = make_view(yb.ptr, yb.gstride, yb.ldp, G);
    const View vO = make_view(out, ogstride, ldo, G);
    const int gc = blockIdx.x;
    {
        const int g = gc / C, c = gc - g * C;
        const int nv = nvalid_of(nvalid, g, N);
        TilePartials tp;
        if (FIN && wave < 2) tp = finalize_load(wave ? F.part_b : F.part_a, F.cnt, g, c, C, F.tpg, lane);   // first in flight
        load_tile_raw(xa, vA, g * vA.gs4 + c * vA.ld4, N, nv, tid);
        load_tile_raw(xb, vB, g * vB.gs4 + c * vB.ld4, N, nv, tid);
        NormRec na, nb;
        if (FIN) {
            if (wave < 2) {
                const float4 r = finalize_reduce(tp, (float)nv, (wave ? F.gw_b : F.gw_a) ? (wave ? F.gw_b : F.gw_a)[c] : 1.f, F.eps);
                if (lane == 0) {
                    fin_rec[wave] = r;
                    reinterpret_cast<float4 *>(wave ? F.nrm_b : F.nrm_a)[gc] = r;
                }
            }
            __syncthreads();
            const float4 ra = fin_rec[0], rb = fin_rec[1];
            na.on = true; na.mean = ra.x; na.a = ra.y; na.beta = ya.beta ? ya.beta[c] : 0.f;
            nb.on = true; nb.mean = rb.x; nb.a = rb.y; nb.beta = yb.beta ? yb.beta[c] : 0.f;
        } else {
            na = norm_of(ya, g, c);
            nb = norm_of(yb, g, c);
        }
        const int o_off = g * vO.gs4 + c * vO.ld4;
        {
            float va[16], vb[16];
            finish_tile<false>(va, dummy, xa, nv, na, tid);
            finish_tile<false>(vb, dummy, xb, nv, nb, tid);
            store_tile_lds(As, va, tid);
            store_tile_lds(Bs, vb, tid);
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (32 * qi < N && 32 * qj < N) {
            const float *ap = As + (32 * qi + j) * LDS_LD + h;
            const float *bp = Bs + h * LDS_LD + 32 * qj + j;
            for (int k = 0; k < N; k += 2) acc = mfma32(ap[k], bp[k * LDS_LD], acc);
        }
        __syncthreads();                                   // all waves done reading As
        frag_to_lds(As, 32 * qi, 32 * qj, acc, lane);
        __syncthreads();
        float s1 = 0.f, s2 = 0.f;
        tile_to_global<false>(vO, o_off, As, N, dummy, s1, s2, tid);
    }
}

// Two LDS tiles, not three: Ya waits in registers while dA = dM Yb^T runs and takes Yb's tile afterwards -- 35 KB instead of
// 52 KB per workgroup, i.e. four workgroups per CU instead of three (same MFMA sequence: results bit-identical).
__global__ __launch_bounds__(256, 4) void chan_matmul_bwd1_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                               const float *dm, long long dmg, long long ldm,
                                                               const int *nvalid, int N, int M, float *da, float *db,
                                                               long long ogstride, long long ldo, float *s12a,
                                                               float *s12b) {
    __shared__ float Bs[TM * LDS_LD];        // Yb, then Ya, then dB on its way out
    __shared__ float Ds[TM * LDS_LD];        // dM, then dA on its way out
    __shared__ float red[4][4];
    const int C = ya.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int j = lane & 31, h = lane >> 5;
    NormRec none;
    none.on = false; none.mean = 0.f; none.a = 1.f; none.beta = 0.f;
    float xa[16], xb[16], xd[16], dummy[16];
    const int G = M / C;
    const View vA = make_view(ya.ptr, ya.gstride, ya.ldp, G), vB = make_view(yb.ptr, yb.gstride, yb.ldp, G);
    const View vD = make_view(dm, dmg, ldm, G);
    const View vOA = make_view(da, ogstride, ldo, G), vOB = make_view(db, ogstride, ldo, G);
    const int gc = blockIdx.x;
    {
        const int g = gc / C, c = gc - g * C;
        const int nv = nvalid_of(nvalid, g, N);
        load_tile_raw(xa, vA, g * vA.gs4 + c * vA.ld4, N, nv, tid);
        load_tile_raw(xb, vB, g * vB.gs4 + c * vB.ld4, N, nv, tid);
        load_tile_raw(xd, vD, g * vD.gs4 + c * vD.ld4, N, nv, tid);
        const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
        const int o_off = g * vOA.gs4 + c * vOA.ld4;
        float ua[16], ub[16], va[16];
        finish_tile<true>(va, ua, xa, nv, na, tid);
        {
            float vb[16], vd[16];
            finish_tile<true>(vb, ub, xb, nv, nb, tid);
            finish_tile<false>(vd, dummy, xd, nv, none, tid);
            store_tile_lds(Bs, vb, tid);
            store_tile_lds(Ds, vd, tid);
        }
        __syncthreads();
        const bool live = 32 * qi < N && 32 * qj < N;
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accA[r] = 0.f;
            accB[r] = 0.f;
        }
        if (live) {
            // dA[i][k] = sum_j dM[i][j] Yb[k][j]
            const float *ap = Ds + (32 * qi + j) * LDS_LD + h;
            const float *bp = Bs + (32 * qj + j) * LDS_LD + h;
            for (int k = 0; k < N; k += 2) accA = mfma32(ap[k], bp[k], accA);
        }
        __syncthreads();                                   // Yb consumed: its tile takes Ya
        store_tile_lds(Bs, va, tid);
        __syncthreads();
        if (live) {
            // dB[k][j] = sum_i Ya[i][k] dM[i][j]
            const float *ap = Bs + h * LDS_LD + 32 * qi + j;
            const float *bp = Ds + h * LDS_LD + 32 * qj + j;
            for (int k = 0; k < N; k += 2) accB = mfma32(ap[k * LDS_LD], bp[k * LDS_LD], accB);
        }
        __syncthreads();                                   // all products done: both tiles are free
        frag_to_lds(Ds, 32 * qi, 32 * qj, accA, lane);     // dA -> Ds
        frag_to_lds(Bs, 32 * qi, 32 * qj, accB, lane);     // dB -> Bs
        __syncthreads();
        float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
        if (s12a) {
            tile_to_global<true>(vOA, o_off, Ds, N, ua, sa1, sa2, tid);
            tile_to_global<true>(vOB, o_off, Bs, N, ub, sb1, sb2, tid);
            sa1 = wave_sum(sa1);
            sa2 = wave_sum(sa2);
            sb1 = wave_sum(sb1);
            sb2 = wave_sum(sb2);
            if (lane == 0) {
                red[wave][0] = sa1;
                red[wave][1] = sa2;
                red[wave][2] = sb1;
                red[wave][3] = sb2;
            }
            __syncthreads();
            if (tid < 4) {
                const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                float *dst = (tid < 2 ? s12a : s12b) + (long long)gc * 2 + (tid & 1);
                *dst = v;
            }
        } else {
            tile_to_global<false>(vOA, o_off, Ds, N, dummy, sa1, sa2, tid);
            tile_to_global<false>(vOB, o_off, Bs, N, dummy, sb1, sb2, tid);
        }
    }
}


// ---------------------------------------------------------------------------------------
// One WAVE per (g,c) matrix (N <= 64): no workgroup barrier, every operand element is loaded and normalised by
// exactly one lane, about a fifth of the instructions per matrix of the workgroup-per-matrix kernels above.
//   k-step s of half-wave h contracts k = 2 s + h, s = 0 .. 4*KQ - 1 (the summation order of the kernels above)
//   A operand (lane = output row): the matrix is loaded row by row (lane = column: coalesced), normalised and
//       staged in a wave-private LDS tile with the even columns in floats 0..31 and the odd ones in 32..63 of a
//       row (row stride 68 floats), so that four consecutive k-steps of a lane are one conflict-free ds_read_b128;
//   B operand (lane = output column, row k): loaded from global memory straight into the fragment layout
//       (a half-wave reads 32 consecutive floats of row k);
//   output: the D fragments are stored directly (a half-wave writes 32 consecutive floats of one row).
// Masking without per-element selects: a padding COLUMN gets the per-lane constants a_eff = beta_eff = 0 (its
// load is out of range and returns 0), a padding ROW of B is out of range of the per-matrix buffer descriptor
// (reads 0; its value only meets the zero columns of A), a padding row of A (ragged batches only) is selected to 0.
// ---------------------------------------------------------------------------------------
int mm_variant_from_env() {         // FGNN_MM_VARIANT: the measurement switch of fgnn_debug_matmul_variant for whole-step runs
    const char *e = getenv("FGNN_MM_VARIANT");
    return e ? atoi(e) : 1;
}
int g_mm_wave_variant = mm_variant_from_env();
inline bool mm_wave_variant() { return (g_mm_wave_variant & 1) != 0; }
inline bool mm_no_split() { return (g_mm_wave_variant & 2) != 0; }
inline bool mm_no_order() { return (g_mm_wave_variant & 4) != 0; }
inline bool mm_narrow() { return (g_mm_wave_variant & 8) != 0; }      // the four-byte-access form of the wave-per-matrix kernel
inline bool mm_pair_per_wave() { return (g_mm_wave_variant & 16) != 0; }      // two matrices per wave (chan_matmul_fwd_wp_kernel, N = 49 ... 56)
constexpr int WLD = 68;            // floats per row of the wave-private A tile
constexpr int W_WAVES = 2;         // matrices (waves) per workgroup

DEVI rsrc_t mat_rsrc(const float *p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
DEVI float rsrc_load(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
DEVI void rsrc_store(float x, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, voff, soff, 0);
}

// rows 0 .. 8*KQ-1 of one matrix, lane = column; rows >= N re-read row 0 (never used)
template <int KQ>
DEVI void wave_rows_load(float (&x)[8 * KQ], rsrc_t r, int N, int voff) {
#pragma unroll
    for (int i = 0; i < 8 * KQ; ++i) x[i] = rsrc_load(r, voff, i < N ? i * N * 4 : 0);
}
// normalise (a_eff, beta_eff are 0 in padding columns) and stage as [row][pos(col)]; ragged: rows >= nv become 0
template <int KQ>
DEVI void wave_rows_stage(float *T, const float (&x)[8 * KQ], float mean, float a_eff, float b_eff, int nv, int N, int lane) {
    if (nv < N) {
#pragma unroll
        for (int i = 0; i < 8 * KQ; ++i) {
            const float v = (x[i] - mean) * a_eff + b_eff;
            T[i * WLD + lane] = i < nv ? v : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8 * KQ; ++i) T[i * WLD + lane] = (x[i] - mean) * a_eff + b_eff;
    }
}

template <int KQ, bool FIN>
__global__ __launch_bounds__(64 * W_WAVES, 2) void chan_matmul_fwd_w_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                                           const int *nvalid, int N, int M, float *out,
                                                                           long long ogstride, long long ldo,
                                                                           const FinArgs F) {
    constexpr int KH = 4 * KQ;
    constexpr bool TWO = KQ > 4;                       // N > 32: two row blocks x two column blocks
    constexpr int NB = TWO ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float lds[W_WAVES * 32 * NB * WLD];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gc = xcd_swizzle(blockIdx.x, gridDim.x) * W_WAVES + wv;
    if (gc >= M) return;
    const int C = ya.C;
    const int g = gc / C, c = gc - g * C;
    const int nv = __builtin_amdgcn_readfirstlane(nvalid_of(nvalid, g, N));
    const int j = lane & 31, h = lane >> 5;
    float *As = lds + wv * (32 * NB * WLD);
    const rsrc_t rA = mat_rsrc(ya.ptr + (long long)g * ya.gstride + (long long)c * ya.ldp, nv * N * 4);
    const rsrc_t rB = mat_rsrc(yb.ptr + (long long)g * yb.gstride + (long long)c * yb.ldp, nv * N * 4);
    const rsrc_t rO = mat_rsrc(out + (long long)g * ogstride + (long long)c * ldo, N * N * 4);

    TilePartials ta, tb;
    if (FIN) {
        ta = finalize_load(F.part_a, F.cnt, g, c, C, F.tpg, lane);
        tb = finalize_load(F.part_b, F.cnt, g, c, C, F.tpg, lane);
    }
    // ---- every load of the matrix pair goes out before anything waits ----
    float xa[8 * KQ], xb[NB][KH];
    wave_rows_load<KQ>(xa, rA, N, lane < nv ? lane * 4 : OOB_OFF);
    int voffB[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const int col = 32 * cb + j;
        voffB[cb] = col < nv ? (h * N + col) * 4 : OOB_OFF;
    }
#pragma unroll
    for (int s = 0; s < KH; ++s)         // in the order the k-steps consume them (vector memory returns in order)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) xb[cb][s] = rsrc_load(rB, voffB[cb] + s * 2 * N * 4, 0);
    // ---- GraphNorm records ----
    float meanA, aA, meanB, aB;
    if (FIN) {
        const float4 ra = finalize_reduce(ta, (float)nv, F.gw_a ? F.gw_a[c] : 1.f, F.eps);
        const float4 rb = finalize_reduce(tb, (float)nv, F.gw_b ? F.gw_b[c] : 1.f, F.eps);
        if (lane == 0) {
            reinterpret_cast<float4 *>(F.nrm_a)[gc] = ra;
            reinterpret_cast<float4 *>(F.nrm_b)[gc] = rb;
        }
        meanA = ra.x; aA = ra.y; meanB = rb.x; aB = rb.y;
    } else {
        const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
        meanA = na.mean; aA = na.a; meanB = nb.mean; aB = nb.a;
    }
    const bool onA = FIN || ya.nrm != nullptr, onB = FIN || yb.nrm != nullptr;
    const float betaA = (onA && ya.beta) ? ya.beta[c] : 0.f, betaB = (onB && yb.beta) ? yb.beta[c] : 0.f;
    // ---- A: normalise, stage ----
    {
        const bool okc = lane < nv;
        wave_rows_stage<KQ>(As, xa, meanA, okc ? aA : 0.f, okc ? betaA : 0.f, nv, N, (lane & 1) * 32 + (lane >> 1));
    }
    // ---- B: normalised chunk by chunk right before its MFMAs (the later rows are still in flight) ----
    float aeB[NB], beB[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const bool okc = 32 * cb + j < nv;
        aeB[cb] = okc ? aB : 0.f;
        beB[cb] = okc ? betaB : 0.f;
    }
    // ---- products: NB x NB independent accumulator chains ----
    f32x16 acc[NB][NB];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        float4 af[NB];
#pragma unroll
        for (int rb = 0; rb < NB; ++rb)
            af[rb] = *reinterpret_cast<const float4 *>(As + (32 * rb + j) * WLD + h * 32 + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // k-steps past the matrix (k = 2 (4 q + t) >= N: up to three of the last quartet) would add 0 * 0: skipped (uniform)
            if (q == KQ - 1 && 2 * (4 * q + t) >= N) break;
            float bv[NB];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) bv[cb] = (xb[cb][4 * q + t] - meanB) * aeB[cb] + beB[cb];
#pragma unroll
            for (int rb = 0; rb < NB; ++rb) {
                const float a = t == 0 ? af[rb].x : (t == 1 ? af[rb].y : (t == 2 ? af[rb].z : af[rb].w));
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
#ifdef FGNN_MM_NOMFMA       // measurement builds only (tools/gpu_mm_ablate.py): a VALU fma in place of every MFMA
                    acc[rb][cb][(4 * q + t) & 15] += a * bv[cb];
#else
                    acc[rb][cb] = mfma32(a, bv[cb], acc[rb][cb]);
#endif
                }
            }
        }
    }
    // ---- output: row 32 rb + ch_of(r, h), column 32 cb + j; rows >= N are out of range of the descriptor ----
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const int col = 32 * cb + j;
        const int voffO = col < N ? (4 * h * N + col) * 4 : OOB_OFF;
#pragma unroll
        for (int rb = 0; rb < NB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#ifdef FGNN_MM_NOSTORE      // measurement builds only: the stores stay in the code but never execute
                if (acc[rb][cb][r] == 12345.678f)
#endif
                rsrc_store(acc[rb][cb][r], rO, voffO + (32 * rb + (r & 3) + 8 * (r >> 2)) * N * 4, 0);
            }
    }
}

// ---- the same kernel with 8-byte accesses (32 < N <= 64; round 5) ------------------------------------------------------------
// A wave of the kernel above issues 56 + 56 four-byte loads and 64 four-byte stores per matrix; a wave can have 64 vector-memory
// instructions outstanding, so its operands arrive in TWO dependent round trips.  Here every access moves 8 bytes per lane:
//   A: lanes 0..31 read the column pair (2 l, 2 l + 1) of row i, lanes 32..63 of row i + 1 -> 28 loads for 56 rows; the pair lands in
//      the de-interleaved LDS row (even columns at l, odd ones at 32 + l) with two conflict-free ds_write_b32;
//   B: lane (j, h) reads the column pair (2 j, 2 j + 1) of row k = 2 s + h -> 28 loads; the MFMA column blocks are the EVEN and the ODD
//      columns (block q, lane j <-> column 2 j + q) instead of the left and the right half;
//   output: block (rb, 0) and (rb, 1) of a lane are the adjacent columns 2 j, 2 j + 1 of a row -> 32 eight-byte stores.
// 56 loads in flight: one round trip.  Which lane computes which element changes, the arithmetic of an element does not (k order,
// normalisation expression): bit-identical to the kernel above (tests/test_gpu_kernels.py).  An odd column that is padding is
// selected to 0 after the load (its partner may be valid), an odd column beyond N is never stored.
DEVI float2 rsrc_load2(rsrc_t r, int voff, int soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    const unsigned a = v.x, b = v.y;            // (copy the elements to scalars first: see the note on vector elements in DESIGN.md)
    return make_float2(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
DEVI void rsrc_store2(float x, float y, rsrc_t r, int voff, int soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 v;
    v.x = __builtin_bit_cast(unsigned, x);
    v.y = __builtin_bit_cast(unsigned, y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

template <int KQ, bool FIN>
__global__ __launch_bounds__(64 * W_WAVES, 2) void chan_matmul_fwd_w2_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                                            const int *nvalid, int N, int M, float *out,
                                                                            long long ogstride, long long ldo,
                                                                            const FinArgs F) {
    static_assert(KQ > 4, "two column blocks: 32 < N <= 64");
    constexpr int KH = 4 * KQ;                         // k-steps; also: row PAIRS of A
    __shared__ __attribute__((aligned(16))) float lds[W_WAVES * 64 * WLD];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gc = xcd_swizzle(blockIdx.x, gridDim.x) * W_WAVES + wv;
    if (gc >= M) return;
    const int C = ya.C;
    const int g = gc / C, c = gc - g * C;
    const int nv = __builtin_amdgcn_readfirstlane(nvalid_of(nvalid, g, N));
    const int j = lane & 31, h = lane >> 5;
    float *As = lds + wv * (64 * WLD);
    const rsrc_t rA = mat_rsrc(ya.ptr + (long long)g * ya.gstride + (long long)c * ya.ldp, nv * N * 4);
    const rsrc_t rB = mat_rsrc(yb.ptr + (long long)g * yb.gstride + (long long)c * yb.ldp, nv * N * 4);
    const rsrc_t rO = mat_rsrc(out + (long long)g * ogstride + (long long)c * ldo, N * N * 4);

    TilePartials ta, tb;
    if (FIN) {
        ta = finalize_load(F.part_a, F.cnt, g, c, C, F.tpg, lane);
        tb = finalize_load(F.part_b, F.cnt, g, c, C, F.tpg, lane);
    }
    // ---- every load of the matrix pair goes out before anything waits: 28 + 28 eight-byte loads ----
    const bool ok0 = 2 * j < nv, ok1 = 2 * j + 1 < nv;          // this lane's column pair (the same for A and B)
    float2 xa[KH], xb[KH];
    {
        const int voffA = ok0 ? (h * N + 2 * j) * 4 : OOB_OFF;   // rows 2 p + h, p = 0 .. KH - 1 (rows >= nv are out of range: 0)
#pragma unroll
        for (int p = 0; p < KH; ++p) xa[p] = rsrc_load2(rA, voffA, 2 * p < N ? 2 * p * N * 4 : 0);
        const int voffB = ok0 ? (h * N + 2 * j) * 4 : OOB_OFF;   // row k = 2 s + h
#pragma unroll
        for (int s = 0; s < KH; ++s) xb[s] = rsrc_load2(rB, voffB, 2 * s < N ? 2 * s * N * 4 : 0);
    }
    // ---- GraphNorm records ----
    float meanA, aA, meanB, aB;
    if (FIN) {
        const float4 ra = finalize_reduce(ta, (float)nv, F.gw_a ? F.gw_a[c] : 1.f, F.eps);
        const float4 rb = finalize_reduce(tb, (float)nv, F.gw_b ? F.gw_b[c] : 1.f, F.eps);
        if (lane == 0) {
            reinterpret_cast<float4 *>(F.nrm_a)[gc] = ra;
            reinterpret_cast<float4 *>(F.nrm_b)[gc] = rb;
        }
        meanA = ra.x; aA = ra.y; meanB = rb.x; aB = rb.y;
    } else {
        const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
        meanA = na.mean; aA = na.a; meanB = nb.mean; aB = nb.a;
    }
    const bool onA = FIN || ya.nrm != nullptr, onB = FIN || yb.nrm != nullptr;
    const float betaA = (onA && ya.beta) ? ya.beta[c] : 0.f, betaB = (onB && yb.beta) ? yb.beta[c] : 0.f;
    // ---- A: normalise, stage rows 2 p + h: even column -> position j, odd column -> position 32 + j ----
    {
        const float a0 = ok0 ? aA : 0.f, b0 = ok0 ? betaA : 0.f, a1 = ok1 ? aA : 0.f, b1 = ok1 ? betaA : 0.f;
#pragma unroll
        for (int p = 0; p < KH; ++p) {
            const int row = 2 * p + h;
            const bool rok = row < nv;                            // (ragged: rows >= nv are zero; dense: nv == N, rows >= N never used)
            const float v0 = (xa[p].x - meanA) * a0 + b0;
            const float v1 = ((ok1 ? xa[p].y : 0.f) - meanA) * a1 + b1;
            As[row * WLD + j] = rok ? v0 : 0.f;
            As[row * WLD + 32 + j] = rok ? v1 : 0.f;
        }
    }
    const float aeB[2] = {ok0 ? aB : 0.f, ok1 ? aB : 0.f}, beB[2] = {ok0 ? betaB : 0.f, ok1 ? betaB : 0.f};
    // ---- products: 2 x 2 independent accumulator chains; column block q = columns 2 j + q ----
    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        float4 af[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            af[rb] = *reinterpret_cast<const float4 *>(As + (32 * rb + j) * WLD + h * 32 + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (q == KQ - 1 && 2 * (4 * q + t) >= N) break;      // k-steps past the matrix: skipped (uniform), as above
            float bv[2];
            bv[0] = (xb[4 * q + t].x - meanB) * aeB[0] + beB[0];
            bv[1] = ((ok1 ? xb[4 * q + t].y : 0.f) - meanB) * aeB[1] + beB[1];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float a = t == 0 ? af[rb].x : (t == 1 ? af[rb].y : (t == 2 ? af[rb].z : af[rb].w));
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma32(a, bv[cb], acc[rb][cb]);
            }
        }
    }
    // ---- output: row 32 rb + ch_of(r, h), columns (2 j, 2 j + 1); rows >= N are out of range of the descriptor ----
    {
        const bool two = 2 * j + 1 < N;
        const int voffO = 2 * j < N ? (4 * h * N + 2 * j) * 4 : OOB_OFF;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (32 * rb + (r & 3) + 8 * (r >> 2)) * N * 4;
                if (two) rsrc_store2(acc[rb][0][r], acc[rb][1][r], rO, voffO, soff);
                else rsrc_store(acc[rb][0][r], rO, voffO, soff);
            }
    }
}

// ---- two matrices per wave (round 5 measurement; fgnn_debug_matmul_variant bit 4) ------------------------------------------------
// Half as many waves (one per SIMD, up to 512 registers), each owning the matrix pairs gc and gc + 1: the operands of BOTH are requested
// before the first product starts, so the 112 MFMAs of the first matrix (and its stores) run while the second matrix' rows are still
// arriving -- the overlap that one matrix per resident wave cannot have.  Same element arithmetic as
// chan_matmul_fwd_w_kernel: bit-identical results.
template <int KQ, bool FIN>
__global__ __launch_bounds__(64 * W_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void chan_matmul_fwd_wp_kernel(
    const fgnn_slab ya, const fgnn_slab yb, const int *nvalid, int N, int M, float *out, long long ogstride, long long ldo, const FinArgs F) {
    constexpr int KH = 4 * KQ;
    constexpr bool TWO = KQ > 4;
    constexpr int NB = TWO ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float lds[W_WAVES * 32 * NB * WLD];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gc0 = (xcd_swizzle(blockIdx.x, gridDim.x) * W_WAVES + wv) * 2;
    if (gc0 >= M) return;
    const int C = ya.C;
    const int j = lane & 31, h = lane >> 5;
    float *As = lds + wv * (32 * NB * WLD);
    float xa[2][8 * KQ], xb[2][NB][KH];
    TilePartials ta[2], tb[2];
    int gq[2], cq[2], nvq[2];
    rsrc_t rO[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int gc = gc0 + m < M ? gc0 + m : gc0;               // (an odd matrix count: the last wave repeats its matrix, same values)
        gq[m] = gc / C;
        cq[m] = gc - gq[m] * C;
        nvq[m] = __builtin_amdgcn_readfirstlane(nvalid_of(nvalid, gq[m], N));
        const int g = gq[m], c = cq[m], nv = nvq[m];
        const rsrc_t rA = mat_rsrc(ya.ptr + (long long)g * ya.gstride + (long long)c * ya.ldp, nv * N * 4);
        const rsrc_t rB = mat_rsrc(yb.ptr + (long long)g * yb.gstride + (long long)c * yb.ldp, nv * N * 4);
        rO[m] = mat_rsrc(out + (long long)g * ogstride + (long long)c * ldo, N * N * 4);
        if (FIN) {
            ta[m] = finalize_load(F.part_a, F.cnt, g, c, C, F.tpg, lane);
            tb[m] = finalize_load(F.part_b, F.cnt, g, c, C, F.tpg, lane);
        }
        wave_rows_load<KQ>(xa[m], rA, N, lane < nv ? lane * 4 : OOB_OFF);
        int voffB[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = 32 * cb + j;
            voffB[cb] = col < nv ? (h * N + col) * 4 : OOB_OFF;
        }
#pragma unroll
        for (int s = 0; s < KH; ++s)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) xb[m][cb][s] = rsrc_load(rB, voffB[cb] + s * 2 * N * 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);          // every load of both matrices is issued before the first product starts
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int g = gq[m], c = cq[m], nv = nvq[m], gc = g * C + c;
        float meanA, aA, meanB, aB;
        if (FIN) {
            const float4 ra = finalize_reduce(ta[m], (float)nv, F.gw_a ? F.gw_a[c] : 1.f, F.eps);
            const float4 rb = finalize_reduce(tb[m], (float)nv, F.gw_b ? F.gw_b[c] : 1.f, F.eps);
            if (lane == 0) {
                reinterpret_cast<float4 *>(F.nrm_a)[gc] = ra;
                reinterpret_cast<float4 *>(F.nrm_b)[gc] = rb;
            }
            meanA = ra.x; aA = ra.y; meanB = rb.x; aB = rb.y;
        } else {
            const NormRec na = norm_of(ya, g, c), nb = norm_of(yb, g, c);
            meanA = na.mean; aA = na.a; meanB = nb.mean; aB = nb.a;
        }
        const bool onA = FIN || ya.nrm != nullptr, onB = FIN || yb.nrm != nullptr;
        const float betaA = (onA && ya.beta) ? ya.beta[c] : 0.f, betaB = (onB && yb.beta) ? yb.beta[c] : 0.f;
        {
            const bool okc = lane < nv;
            wave_rows_stage<KQ>(As, xa[m], meanA, okc ? aA : 0.f, okc ? betaA : 0.f, nv, N, (lane & 1) * 32 + (lane >> 1));
        }
        float aeB[NB], beB[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const bool okc = 32 * cb + j < nv;
            aeB[cb] = okc ? aB : 0.f;
            beB[cb] = okc ? betaB : 0.f;
        }
        f32x16 acc[NB][NB];
#pragma unroll
        for (int rb = 0; rb < NB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            float4 af[NB];
#pragma unroll
            for (int rb = 0; rb < NB; ++rb)
                af[rb] = *reinterpret_cast<const float4 *>(As + (32 * rb + j) * WLD + h * 32 + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (q == KQ - 1 && 2 * (4 * q + t) >= N) break;
                float bv[NB];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) bv[cb] = (xb[m][cb][4 * q + t] - meanB) * aeB[cb] + beB[cb];
#pragma unroll
                for (int rb = 0; rb < NB; ++rb) {
                    const float a = t == 0 ? af[rb].x : (t == 1 ? af[rb].y : (t == 2 ? af[rb].z : af[rb].w));
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = mfma32(a, bv[cb], acc[rb][cb]);
                }
            }
        }
        if (m == 1 && gc0 + 1 >= M) break;                       // the repeated matrix of an odd count is not stored twice
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = 32 * cb + j;
            const int voffO = col < N ? (4 * h * N + col) * 4 : OOB_OFF;
#pragma unroll
            for (int rb = 0; rb < NB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rsrc_store(acc[rb][cb][r], rO[m], voffO + (32 * rb + (r & 3) + 8 * (r >> 2)) * N * 4, 0);
        }
    }
}

template <bool FIN>
int launch_fwd_w(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int N, int M, float *out, long long ogstride,
                 long long ldo, const FinArgs &F, hipStream_t st) {
    if (mm_pair_per_wave() && (N + 7) / 8 == 7) {             // (measurement build: the N = 49 ... 56 instantiation)
        const int grid2 = ((M + 1) / 2 + W_WAVES - 1) / W_WAVES;
        hipLaunchKernelGGL((chan_matmul_fwd_wp_kernel<7, FIN>), dim3(grid2), dim3(64 * W_WAVES), 0, st, *ya, *yb, nvalid, N, M, out, ogstride, ldo, F);
        return 0;
    }
    const int KQ = (N + 7) / 8, grid = (M + W_WAVES - 1) / W_WAVES;
    // Measured (tools/gpu_mm_wide_probe.py, profiles/archive/r05_mm_wide_probe.txt: 30 back-to-back launches in a replayed graph, 64 x 32 matrices):
    // N = 64: 21.8 - 22.4 us against 25.0 for the four-byte kernel; N = 50: 16.4 - 17.3 against 16.0 - 16.2; N = 40: 13.3 against 12.7 --
    // the instruction count is not what limits the kernel below N ~ 58 (it moves its 61 MB at the ~4 TB/s this part reads at), so
    // the eight-byte form runs where it wins: KQ = 8, even N (an odd N would let a column pair straddle two rows).
    if (KQ == 8 && !mm_narrow() && (N & 1) == 0) {
#define FGNN_W2(K_)                                                                                                       \
    case K_:                                                                                                             \
        hipLaunchKernelGGL((chan_matmul_fwd_w2_kernel<K_, FIN>), dim3(grid), dim3(64 * W_WAVES), 0, st, *ya, *yb, nvalid, N, \
                           M, out, ogstride, ldo, F);                                                                    \
        break;
        switch (KQ) {
            FGNN_W2(8)
        }
#undef FGNN_W2
        return 0;
    }
#define FGNN_W(K_)                                                                                                       \
    case K_:                                                                                                             \
        hipLaunchKernelGGL((chan_matmul_fwd_w_kernel<K_, FIN>), dim3(grid), dim3(64 * W_WAVES), 0, st, *ya, *yb, nvalid, N, \
                           M, out, ogstride, ldo, F);                                                                    \
        break;
    switch (KQ) {
        FGNN_W(1) FGNN_W(2) FGNN_W(3) FGNN_W(4) FGNN_W(5) FGNN_W(6) FGNN_W(7) FGNN_W(8)
    }
#undef FGNN_W
    return 0;
}

// =======================================================================================
// Whole matrix per workgroup, 64 < N <= 256 (cfg4 N = 200, ragged batches padded past 64).
// One 512-thread workgroup owns one (g,c) matrix: every operand element is read from HBM once,
// streamed through double-buffered LDS panels in 32-wide k chunks, and all ceil(nv/32)^2 output
// tiles of 32x32 live in the accumulators of the eight waves (<= NT*NT/8 tiles per wave).  Work is
// limited to the valid nv x nv part of a ragged graph; the padding of the output is zero-filled.
//   Out[m][n] = sum_k OpA(m,k) OpB(k,n),  OpA(m,k) = A_KC ? MA[m][k] : MA[k][m],
//                                         OpB(k,n) = B_KC ? MB[n][k] : MB[k][n]
//   forward M = Ya Yb: (KC, KR);   dA = dM Yb^T: (KC, KC);   dB = Ya^T dM: (KR, KR)
// A K-contiguous (KC) panel is stored [x][kk] with a 36-float row stride (conflict-free
// ds_read_b128 of 4 k-steps), a K-row (KR) panel [kk][x] (conflict-free ds_read_b32).
// Inside a chunk k-step s of half-wave h contracts k = 16h + s.
// NT (4 or 8) is the compile-time bound on tiles per side: panel shapes, sweep counts and all LDS
// offsets are constants, every staging load / store is unconditional (out-of-range lanes read 0
// through the buffer descriptor and land in panel rows that are never consumed).
// =======================================================================================
constexpr int BIG_NW = 8, BIG_THREADS = 64 * BIG_NW, BIG_KC = 32, BIG_LDK = 36;
constexpr int BIG_SPLIT_MAX_MATRICES = 8 * 256;         // four rounds of two workgroups on each of the 256 CUs

struct BigSrc {
    View v;
    int off4;          // byte offset of the (g,c) matrix inside the view
    NormRec nr;
};

template <int NT>
struct BigCfg {
    static constexpr int XM = 32 * NT;                       // panel extent (rows of a KC panel, columns of a KR one)
    static constexpr int MAXT = NT * NT / BIG_NW;            // tiles per wave
    static constexpr int SWEEPS = XM * BIG_KC / BIG_THREADS; // elements per thread per panel
    static constexpr int KC_F = XM * BIG_LDK, KR_F = BIG_KC * XM;
    static constexpr int BUF_F = 2 * KC_F;                   // one buffer holds the largest pair (KC + KC)
    static constexpr int LDS_BYTES = 2 * BUF_F * 4;
};

// global -> registers: the k0 chunk of one operand
template <int NT, bool KC>
DEVI void big_load(float (&x)[BigCfg<NT>::SWEEPS], const BigSrc &s, int N, int nv, int k0, int tid) {
    using Cf = BigCfg<NT>;
    if (KC) {
        const int kk = tid & 31, xr = tid >> 5;                      // 16 rows per sweep
        const int base = (k0 + kk) < nv ? (xr * N + k0 + kk) * 4 : OOB_OFF;
#pragma unroll
        for (int i = 0; i < Cf::SWEEPS; ++i)
            x[i] = buf_load(s.v, (xr + 16 * i) < nv ? base : OOB_OFF, s.off4 + 16 * i * N * 4);
    } else {
        constexpr int RPS = BIG_THREADS / Cf::XM;                    // k rows per sweep
        const int kk = tid / Cf::XM, c = tid % Cf::XM;
        const int base = c < nv ? ((k0 + kk) * N + c) * 4 : OOB_OFF;
#pragma unroll
        for (int i = 0; i < Cf::SWEEPS; ++i)
            x[i] = buf_load(s.v, (k0 + kk + RPS * i) < nv ? base : OOB_OFF, s.off4 + RPS * i * N * 4);
    }
}

// registers -> LDS panel, normalising on the way (padding stays exactly 0)
template <int NT, bool KC>
DEVI void big_stage(float *P, const float (&x)[BigCfg<NT>::SWEEPS], const BigSrc &s, int nv, int k0, int tid) {
    using Cf = BigCfg<NT>;
    if (KC) {
        const int kk = tid & 31, xr = tid >> 5;
        const bool kok = (k0 + kk) < nv;
        float *dst = P + xr * BIG_LDK + kk;
#pragma unroll
        for (int i = 0; i < Cf::SWEEPS; ++i) {
            const float m = (kok && (xr + 16 * i) < nv) ? 1.f : 0.f;
            dst[16 * i * BIG_LDK] = s.nr.on ? ((x[i] - s.nr.mean) * s.nr.a + s.nr.beta) * m : x[i];
        }
    } else {
        constexpr int RPS = BIG_THREADS / Cf::XM;
        const int kk = tid / Cf::XM, c = tid % Cf::XM;
        const bool cok = c < nv;
        float *dst = P + kk * Cf::XM + c;
#pragma unroll
        for (int i = 0; i < Cf::SWEEPS; ++i) {
            const float m = (cok && (k0 + kk + RPS * i) < nv) ? 1.f : 0.f;
            dst[RPS * i * Cf::XM] = s.nr.on ? ((x[i] - s.nr.mean) * s.nr.a + s.nr.beta) * m : x[i];
        }
    }
}

// the 16 k-steps of one 32-row (or 32-column) strip `t` of a panel, for lane (j, h)
template <int NT, bool KC>
DEVI void big_operand(float (&o)[16], const float *P, int t, int j, int h) {
    if (KC) {
        const float4 *p = reinterpret_cast<const float4 *>(P + (32 * t + j) * BIG_LDK + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = p[q];
            o[4 * q + 0] = v.x;
            o[4 * q + 1] = v.y;
            o[4 * q + 2] = v.z;
            o[4 * q + 3] = v.w;
        }
    } else {
        const float *p = P + (16 * h) * BigCfg<NT>::XM + 32 * t + j;
#pragma unroll
        for (int s = 0; s < 16; ++s) o[s] = p[s * BigCfg<NT>::XM];
    }
}

// k-steps 4q .. 4q+3 of the same strip
template <int NT, bool KC>
DEVI void big_operand4(float (&o)[4], const float *P, int t, int j, int h, int q) {
    if (KC) {
        const float4 v = reinterpret_cast<const float4 *>(P + (32 * t + j) * BIG_LDK + 16 * h)[q];
        o[0] = v.x;
        o[1] = v.y;
        o[2] = v.z;
        o[3] = v.w;
    } else {
        const float *p = P + (16 * h + 4 * q) * BigCfg<NT>::XM + 32 * t + j;
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = p[s * BigCfg<NT>::XM];
    }
}

// acc[ti] (tile wv + 8*ti of the ntv x ntv valid tiles) += OpA OpB over all k chunks
template <int NT, bool A_KC, bool B_KC>
DEVI void big_gemm(f32x16 (&acc)[BigCfg<NT>::MAXT], const BigSrc &A, const BigSrc &B, float *lds, int N, int nv,
                   int ntv, int tid) {
    using Cf = BigCfg<NT>;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int T = ntv * ntv;
    constexpr int FA = A_KC ? Cf::KC_F : Cf::KR_F;
    // (plain offsets from the __shared__ base: a runtime-indexed array of pointers makes the compiler
    //  lose the LDS address space and emit flat_load instead of ds_read)
#pragma unroll
    for (int ti = 0; ti < Cf::MAXT; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
    const int nkc = (nv + BIG_KC - 1) / BIG_KC;
    int tms[Cf::MAXT], tns[Cf::MAXT];                       // wave-uniform tile coordinates (rolled form)
    if constexpr (NT != 8) {
#pragma unroll
        for (int ti = 0; ti < Cf::MAXT; ++ti) {
            const int t = wv + BIG_NW * ti;
            tms[ti] = t / ntv;
            tns[ti] = t - tms[ti] * ntv;
        }
    }
    float xa[Cf::SWEEPS], xb[Cf::SWEEPS];
    big_load<NT, A_KC>(xa, A, N, nv, 0, tid);
    big_load<NT, B_KC>(xb, B, N, nv, 0, tid);
    big_stage<NT, A_KC>(lds, xa, A, nv, 0, tid);
    big_stage<NT, B_KC>(lds + FA, xb, B, nv, 0, tid);
    __syncthreads();
    for (int c = 0; c < nkc; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nkc;
        if (more) {
            big_load<NT, A_KC>(xa, A, N, nv, (c + 1) * BIG_KC, tid);
            big_load<NT, B_KC>(xb, B, N, nv, (c + 1) * BIG_KC, tid);
        }
        const float *pa = lds + cur * Cf::BUF_F, *pb = pa + FA;
        // k-steps in groups of four (k = k0 + 16 h + s, s = 4q .. 4q+3); in the last chunk of a matrix only the groups
        // with 4q < kv hold a valid k -- the skipped products are exact zeros (padding is staged as 0).  The group loop is
        // rolled: 4 waves per SIMD cover the LDS latency of a group, and the registers stay under the 128 of that occupancy.
        const int kv = nv - c * BIG_KC;
        const int nq = kv >= 13 ? 4 : (kv + 3) >> 2;
        if constexpr (NT == 8) {
            // one workgroup per CU (LDS), two waves per SIMD: the straight-line form (no trimming) is 2 % faster at N = 200
#pragma unroll
            for (int ti = 0; ti < Cf::MAXT; ++ti) {
                const int t = wv + BIG_NW * ti;
                if (t < T) {
                    const int tm = t / ntv, tn = t - tm * ntv;
                    float a[16], b[16];
                    big_operand<NT, A_KC>(a, pa, tm, j, h);
                    big_operand<NT, B_KC>(b, pb, tn, j, h);
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[ti] = mfma32(a[s], b[s], acc[ti]);
                }
            }
        } else {
        for (int q = 0; q < nq; ++q) {
#pragma unroll
            for (int ti = 0; ti < Cf::MAXT; ++ti) {
                if (wv + BIG_NW * ti < T) {
                    float a[4], b[4];
                    big_operand4<NT, A_KC>(a, pa, tms[ti], j, h, q);
                    big_operand4<NT, B_KC>(b, pb, tns[ti], j, h, q);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[ti] = mfma32(a[s], b[s], acc[ti]);
                }
            }
        }
        }
        if (more) {
            float *nx = lds + (cur ^ 1) * Cf::BUF_F;
            big_stage<NT, A_KC>(nx, xa, A, nv, (c + 1) * BIG_KC, tid);
            big_stage<NT, B_KC>(nx + FA, xb, B, nv, (c + 1) * BIG_KC, tid);
        }
        __syncthreads();
    }
}

// accumulators -> global (rows/cols >= N dropped), optionally S1 = sum t, S2 = sum t * (raw - mean) over
// the valid entries, `raw` re-read from the un-normalised slab
template <int NT, bool WANT_S>
DEVI void big_store(const f32x16 (&acc)[BigCfg<NT>::MAXT], const View &ov, int o_off4, const BigSrc &raw, int N,
                    int nv, int ntv, float &s1, float &s2, int tid) {
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int T = ntv * ntv;
#pragma unroll
    for (int ti = 0; ti < BigCfg<NT>::MAXT; ++ti) {
        const int t = wv + BIG_NW * ti;
        if (t < T) {
            const int tm = t / ntv, tn = t - tm * ntv;
            const int col = 32 * tn + j;
            const int rowb = 32 * tm + 4 * h;
            const int base = col < N ? (rowb * N + col) * 4 : OOB_OFF;
            const int vbase = col < nv ? (rowb * N + col) * 4 : OOB_OFF;
            float u[16];
            if (WANT_S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    u[r] = buf_load(raw.v, (rowb + dr) < nv ? vbase : OOB_OFF, raw.off4 + dr * N * 4);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                buf_store(acc[ti][r], ov, (rowb + dr) < N ? base : OOB_OFF, o_off4 + dr * N * 4);
                if (WANT_S) {
                    const float m = ((rowb + dr) < nv && col < nv) ? 1.f : 0.f;
                    s1 += acc[ti][r] * m;
                    s2 += acc[ti][r] * ((u[r] - raw.nr.mean) * m);
                }
            }
        }
    }
}

// zero the part of an N x N output outside the first X = 32*ntv rows / columns.  tiles_only: the consumers step over the tiles
// (32 consecutive pixels of the plane) that hold no pixel of the valid nv x nv corner and read only the corner otherwise, so what
// can be read outside the X x X block is the tail [X, N) of the rows i < nv and -- when nv is a multiple of 32 -- the first 32
// pixels of row nv, which share a tile with the last valid pixel: 2-3 x fewer bytes than the whole frame, and no N x N index loop
DEVI void big_zero_fill(const View &ov, int o_off4, int N, int nv, int X, int tid, bool tiles_only) {
    if (X >= N) return;
    if (!tiles_only) {
        for (int e = tid; e < N * N; e += BIG_THREADS) {
            const int r = e / N, c = e - r * N;
            if (r >= X || c >= X) buf_store(0.f, ov, e * 4, o_off4);
        }
        return;
    }
    const int w = N - X;
    for (int e = tid; e < nv * w; e += BIG_THREADS) {
        const int r = e / w, c = X + (e - r * w);
        buf_store(0.f, ov, (r * N + c) * 4, o_off4);
    }
    if (nv == X && tid < 32 && tid < N) buf_store(0.f, ov, (nv * N + tid) * 4, o_off4);
}

DEVI BigSrc big_src(const fgnn_slab &s, int G, int g, int c) {
    BigSrc b;
    b.v = make_view(s.ptr, s.gstride, s.ldp, G);
    b.off4 = g * b.v.gs4 + c * b.v.ld4;
    b.nr = norm_of(s, g, c);
    return b;
}
DEVI BigSrc big_src_plain(const float *p, long long gs, long long ld, int G, int g, int c) {
    BigSrc b;
    b.v = make_view(p, gs, ld, G);
    b.off4 = g * b.v.gs4 + c * b.v.ld4;
    b.nr.on = false;
    b.nr.mean = 0.f;
    b.nr.a = 1.f;
    b.nr.beta = 0.f;
    return b;
}

// FIN: as chan_matmul_fwd1_kernel<true> -- waves 0 / 1 finalize the GraphNorm records of the two operands from the tile statistics
// of the mlp1 + mlp2 forward launch and publish them (the work of fgnn_gn_finalize2, without its launch)
template <int NT, bool FIN>
__global__ __launch_bounds__(BIG_THREADS) void chan_matmul_fwd_big_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                                           const int *nvalid, int N, int G, float *out,
                                                                           long long ogstride, long long ldo,
                                                                           const int *order, const FinArgs F, const int fill) {
    extern __shared__ __attribute__((aligned(16))) float big_lds[];
    __shared__ float4 fin_rec[2];
    const int C = ya.C, gi = blockIdx.x / C, c = blockIdx.x - gi * C, tid = threadIdx.x;
    const int g = order ? order[gi] : gi;
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    BigSrc A = FIN ? big_src_plain(ya.ptr, ya.gstride, ya.ldp, G, g, c) : big_src(ya, G, g, c);
    BigSrc B = FIN ? big_src_plain(yb.ptr, yb.gstride, yb.ldp, G, g, c) : big_src(yb, G, g, c);
    if (FIN) {
        const int wv = tid >> 6, lane = tid & 63;
        if (wv < 2) {
            const float *gw = wv ? F.gw_b : F.gw_a;
            const float4 r = finalize_wave_any(wv ? F.part_b : F.part_a, F.cnt, g, c, C, F.tpg, (float)nv, gw ? gw[c] : 1.f, F.eps, lane);
            if (lane == 0) {
                fin_rec[wv] = r;
                reinterpret_cast<float4 *>(wv ? F.nrm_b : F.nrm_a)[g * C + c] = r;
            }
        }
        __syncthreads();
        const float4 ra = fin_rec[0], rb = fin_rec[1];
        A.nr.on = true; A.nr.mean = ra.x; A.nr.a = ra.y; A.nr.beta = ya.beta ? ya.beta[c] : 0.f;
        B.nr.on = true; B.nr.mean = rb.x; B.nr.a = rb.y; B.nr.beta = yb.beta ? yb.beta[c] : 0.f;
    }
    const View vO = make_view(out, ogstride, ldo, G);
    const int o_off = g * vO.gs4 + c * vO.ld4;
    big_zero_fill(vO, o_off, N, nv, 32 * ntv, tid, fill != 0);
    if (ntv == 0) return;
    f32x16 acc[BigCfg<NT>::MAXT];
    big_gemm<NT, true, false>(acc, A, B, big_lds, N, nv, ntv, tid);
    float s1 = 0.f, s2 = 0.f;
    big_store<NT, false>(acc, vO, o_off, A, N, nv, ntv, s1, s2, tid);
}

template <int NT>
__global__ __launch_bounds__(BIG_THREADS) void chan_matmul_bwd_big_kernel(const fgnn_slab ya, const fgnn_slab yb,
                                                                           const float *dm, long long dmg, long long ldm,
                                                                           const int *nvalid, int N, int G, float *da,
                                                                           float *db, long long ogstride, long long ldo,
                                                                           float *s12a, float *s12b, const int *order,
                                                                           int split, const int fill) {
    extern __shared__ __attribute__((aligned(16))) float big_lds[];
    __shared__ float red[BIG_NW][4];
    // split: two workgroups per matrix, one per product (the longest-job-first schedule of ragged batches wants the finer
    // grain: the largest matrix of a batch is otherwise a quarter of its CU's whole share of the launch)
    const int item = split ? blockIdx.x >> 1 : blockIdx.x;
    const bool do_a = !split || (blockIdx.x & 1) == 0, do_b = !split || (blockIdx.x & 1) == 1;
    const int C = ya.C, gi = item / C, c = item - gi * C, tid = threadIdx.x;
    const int g = order ? order[gi] : gi, gc = g * C + c;
    const int lane = tid & 63, wv = tid >> 6;
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    const BigSrc A = big_src(ya, G, g, c), B = big_src(yb, G, g, c);
    const BigSrc D = big_src_plain(dm, dmg, ldm, G, g, c);
    const View vOA = make_view(da, ogstride, ldo, G), vOB = make_view(db, ogstride, ldo, G);
    const int o_off = g * vOA.gs4 + c * vOA.ld4;
    if (do_a) big_zero_fill(vOA, o_off, N, nv, 32 * ntv, tid, fill != 0);
    if (do_b) big_zero_fill(vOB, o_off, N, nv, 32 * ntv, tid, fill != 0);
    float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
    if (ntv > 0) {
        f32x16 acc[BigCfg<NT>::MAXT];
        if (do_a) {
            big_gemm<NT, true, true>(acc, D, B, big_lds, N, nv, ntv, tid);          // dA = dM Yb^T
            if (s12a) big_store<NT, true>(acc, vOA, o_off, A, N, nv, ntv, sa1, sa2, tid);
            else big_store<NT, false>(acc, vOA, o_off, A, N, nv, ntv, sa1, sa2, tid);
        }
        if (do_b) {
            big_gemm<NT, false, false>(acc, A, D, big_lds, N, nv, ntv, tid);        // dB = Ya^T dM
            if (s12a) big_store<NT, true>(acc, vOB, o_off, B, N, nv, ntv, sb1, sb2, tid);
            else big_store<NT, false>(acc, vOB, o_off, B, N, nv, ntv, sb1, sb2, tid);
        }
    }
    if (s12a) {
        sa1 = wave_sum(sa1);
        sa2 = wave_sum(sa2);
        sb1 = wave_sum(sb1);
        sb2 = wave_sum(sb2);
        if (lane == 0) {
            red[wv][0] = sa1;
            red[wv][1] = sa2;
            red[wv][2] = sb1;
            red[wv][3] = sb2;
        }
        __syncthreads();
        if (tid < 4 && (tid < 2 ? do_a : do_b)) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < BIG_NW; ++w) v += red[w][tid];                      // fixed order
            float *dst = (tid < 2 ? s12a : s12b) + (long long)gc * 2 + (tid & 1);
            *dst = v;
        }
    }
}

inline bool big_path(int N) { return N > TM && N <= 256; }

// debug only (tests/diag/gpu_mm_variants_equal.py): 0 selects the workgroup-per-matrix forward kernel for N <= 64, whose
// results the wave-per-matrix kernel reproduces bit for bit
// bit 1: keep both products of a matrix in one workgroup even when an order is given; bit 2: ignore the order (tools/gpu_mm_big_probe.py)

}  // namespace

template <int MT, bool FIN>
static void launch_fwd_big(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int N, int G, float *out, long long ogstride,
                           long long ldo, const int *order, const FinArgs &F, int fill, hipStream_t st) {
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)chan_matmul_fwd_big_kernel<MT, FIN>, BigCfg<MT>::LDS_BYTES);
    hipLaunchKernelGGL((chan_matmul_fwd_big_kernel<MT, FIN>), dim3(G * ya->C), dim3(BIG_THREADS), BigCfg<MT>::LDS_BYTES, st, *ya, *yb, nvalid, N, G, out,
                       ogstride, ldo, order, F, fill);
}

extern "C" int fgnn_chan_matmul_fwd(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int G, int N,
                                    float *out, long long ogstride, long long ldo, void *stream) {
    return fgnn_chan_matmul_fwd_ord(ya, yb, nvalid, G, N, out, ogstride, ldo, nullptr, 0, stream);
}

extern "C" int fgnn_chan_matmul_fwd_ord(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int G, int N,
                                        float *out, long long ogstride, long long ldo, const int *order, int fill, void *stream) {
    FGNN_CHECK(!order || nvalid, "fgnn_chan_matmul_fwd_ord: an order needs the nvalid it was derived from");
    if (mm_no_order()) order = nullptr;
    FGNN_CHECK(ya && yb && out && ya->ptr && yb->ptr, "fgnn_chan_matmul_fwd: null argument");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0, "fgnn_chan_matmul_fwd: bad shapes");
    FGNN_CHECK((long long)G * ya->C <= 65535 * 1024ll, "fgnn_chan_matmul_fwd: G*C too large");
    const int t = (N + TM - 1) / TM;
    if (t == 1) {
        FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 4 && (long long)G * yb->gstride < 0x7fffffffll / 4,
                   "fgnn_chan_matmul_fwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
        FGNN_CHECK((long long)G * ogstride < 0x7fffffffll / 4, "fgnn_chan_matmul_fwd: output exceeds 2 GiB");
        const int M = G * ya->C;
        if (mm_wave_variant()) {
            launch_fwd_w<false>(ya, yb, nvalid, N, M, out, ogstride, ldo, FinArgs{}, (hipStream_t)stream);
            FGNN_LAUNCH_CHECK();
            return 0;
        }
        hipLaunchKernelGGL(chan_matmul_fwd1_kernel<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, *ya, *yb, nvalid,
                           N, M, out, ogstride, ldo, FinArgs{});
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    if (big_path(N)) {
        FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 4 && (long long)G * yb->gstride < 0x7fffffffll / 4 &&
                   (long long)G * ogstride < 0x7fffffffll / 4,
                   "fgnn_chan_matmul_fwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
        if (N <= 128) launch_fwd_big<4, false>(ya, yb, nvalid, N, G, out, ogstride, ldo, order, FinArgs{}, fill, (hipStream_t)stream);
        else launch_fwd_big<8, false>(ya, yb, nvalid, N, G, out, ogstride, ldo, order, FinArgs{}, fill, (hipStream_t)stream);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(chan_matmul_fwd_kernel, dim3(t, t, G * ya->C), dim3(256), 0, (hipStream_t)stream, *ya, *yb,
                       nvalid, N, out, ogstride, ldo);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_chan_matmul_fwd_fin_supported(int N) { return (N <= TM || big_path(N)) ? 1 : 0; }

extern "C" int fgnn_debug_matmul_variant(int wave_per_matrix) {
    g_mm_wave_variant = wave_per_matrix;
    return 0;
}

extern "C" int fgnn_chan_matmul_fwd_fin(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                                        const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                        const int *nvalid, int G, int N, float *out, long long ogstride, long long ldo,
                                        void *stream) {
    return fgnn_chan_matmul_fwd_fin_ord(ya, yb, part_a, part_b, cnt, gn_weight_a, gn_weight_b, eps, nvalid, G, N, out, ogstride, ldo, nullptr, 0, stream);
}

extern "C" int fgnn_chan_matmul_fwd_fin_ord(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                                            const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                            const int *nvalid, int G, int N, float *out, long long ogstride, long long ldo,
                                            const int *order, int fill, void *stream) {
    return fgnn_chan_matmul_fwd_fin_ord_r(ya, yb, part_a, part_b, cnt, gn_weight_a, gn_weight_b, eps, nvalid, G, N, fgnn_tiles_per_graph(N), out,
                                          ogstride, ldo, order, fill, stream);
}
// _r: `recs` statistics records per graph (fgnn_mlp_fwd_t16: one per 16-pixel half)
extern "C" int fgnn_chan_matmul_fwd_fin_ord_r(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                                              const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                              const int *nvalid, int G, int N, int recs, float *out, long long ogstride, long long ldo,
                                              const int *order, int fill, void *stream) {
    FGNN_CHECK(!order || nvalid, "fgnn_chan_matmul_fwd_fin_ord: an order needs the nvalid it was derived from");
    if (mm_no_order()) order = nullptr;
    FGNN_CHECK(ya && yb && out && ya->ptr && yb->ptr && part_a && part_b && cnt, "fgnn_chan_matmul_fwd_fin: null argument");
    FGNN_CHECK(ya->nrm && yb->nrm, "fgnn_chan_matmul_fwd_fin: the slabs must carry the record buffers to fill");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0, "fgnn_chan_matmul_fwd_fin: bad shapes");
    FGNN_CHECK(N <= TM || big_path(N), "fgnn_chan_matmul_fwd_fin: N=%d > 256 (use fgnn_gn_finalize2 + fgnn_chan_matmul_fwd)", N);
    FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 4 && (long long)G * yb->gstride < 0x7fffffffll / 4 &&
               (long long)G * ogstride < 0x7fffffffll / 4,
               "fgnn_chan_matmul_fwd_fin: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    const int M = G * ya->C;
    FGNN_CHECK(recs > 0, "fgnn_chan_matmul_fwd_fin: recs");
    FinArgs F = {part_a, part_b, cnt, gn_weight_a, gn_weight_b, const_cast<float *>(ya->nrm), const_cast<float *>(yb->nrm), eps, recs};
    if (big_path(N)) {
        if (N <= 128) launch_fwd_big<4, true>(ya, yb, nvalid, N, G, out, ogstride, ldo, order, F, fill, (hipStream_t)stream);
        else launch_fwd_big<8, true>(ya, yb, nvalid, N, G, out, ogstride, ldo, order, F, fill, (hipStream_t)stream);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    if (mm_wave_variant()) {
        launch_fwd_w<true>(ya, yb, nvalid, N, M, out, ogstride, ldo, F, (hipStream_t)stream);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(chan_matmul_fwd1_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, *ya, *yb, nvalid, N, M, out,
                       ogstride, ldo, F);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_chan_matmul_bwd(const fgnn_slab *ya, const fgnn_slab *yb, const float *dm, long long dmgstride,
                                    long long ldm, const int *nvalid, int G, int N, float *da, float *db,
                                    long long ogstride, long long ldo, float *s12a, float *s12b, void *stream) {
    return fgnn_chan_matmul_bwd_ord(ya, yb, dm, dmgstride, ldm, nvalid, G, N, da, db, ogstride, ldo, s12a, s12b, nullptr,
                                    0, stream);
}

extern "C" int fgnn_chan_matmul_bwd_ord(const fgnn_slab *ya, const fgnn_slab *yb, const float *dm, long long dmgstride,
                                        long long ldm, const int *nvalid, int G, int N, float *da, float *db,
                                        long long ogstride, long long ldo, float *s12a, float *s12b, const int *order,
                                        int fill, void *stream) {
    FGNN_CHECK(!order || nvalid, "fgnn_chan_matmul_bwd_ord: an order needs the nvalid it was derived from");
    if (mm_no_order()) order = nullptr;
    FGNN_CHECK(ya && yb && dm && da && db && ya->ptr && yb->ptr, "fgnn_chan_matmul_bwd: null argument");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0, "fgnn_chan_matmul_bwd: bad shapes");
    FGNN_CHECK((s12a == nullptr) == (s12b == nullptr), "fgnn_chan_matmul_bwd: s12a and s12b come together");
    FGNN_CHECK(!s12a || (ya->nrm && yb->nrm), "fgnn_chan_matmul_bwd: s12 outputs need normalised slabs");
    const int t = (N + TM - 1) / TM;
    const bool fused = s12a && t == 1;
    if (t == 1) {
        FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 4 && (long long)G * yb->gstride < 0x7fffffffll / 4 &&
                   (long long)G * dmgstride < 0x7fffffffll / 4,
                   "fgnn_chan_matmul_bwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
        FGNN_CHECK((long long)G * ogstride < 0x7fffffffll / 4, "fgnn_chan_matmul_bwd: output exceeds 2 GiB");
        const int M = G * ya->C;
        hipLaunchKernelGGL(chan_matmul_bwd1_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, *ya, *yb, dm,
                           dmgstride, ldm, nvalid, N, M, da, db, ogstride, ldo, s12a, s12b);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    if (big_path(N)) {
        FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 4 && (long long)G * yb->gstride < 0x7fffffffll / 4 &&
                   (long long)G * dmgstride < 0x7fffffffll / 4 && (long long)G * ogstride < 0x7fffffffll / 4,
                   "fgnn_chan_matmul_bwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
#define FGNN_BIG_BWD(MT)                                                                                              \
    {                                                                                                                 \
        static LdsAttrCache attr_cache;                                                                           \
        (void)fgnn_raise_lds(attr_cache, (const void *)chan_matmul_bwd_big_kernel<MT>, BigCfg<MT>::LDS_BYTES);                   \
        hipLaunchKernelGGL(chan_matmul_bwd_big_kernel<MT>, dim3(G * ya->C * (split ? 2 : 1)), dim3(BIG_THREADS),      \
                           BigCfg<MT>::LDS_BYTES, (hipStream_t)stream, *ya, *yb, dm, dmgstride, ldm, nvalid, N, G, da, \
                           db, ogstride, ldo, s12a, s12b, order, split, fill);                                            \
    }
        // fewer than four rounds of workgroups (two per CU): schedule the products separately
        const int split = (order && !mm_no_split() && (long long)G * ya->C <= BIG_SPLIT_MAX_MATRICES) ? 1 : 0;
        if (N <= 128) FGNN_BIG_BWD(4)
        else FGNN_BIG_BWD(8)
#undef FGNN_BIG_BWD
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(chan_matmul_bwd_kernel, dim3(t, t, G * ya->C), dim3(256), 0, (hipStream_t)stream, *ya, *yb, dm,
                       dmgstride, ldm, nvalid, N, da, db, ogstride, ldo, fused ? s12a : nullptr, fused ? s12b : nullptr);
    FGNN_LAUNCH_CHECK();
    if (s12a && !fused) {   // matrices span several workgroups: separate reduction passes
        int rc = fgnn_gn_bwd_stats(da, ogstride, ldo, ya->ptr, ya->gstride, ya->ldp, ya->nrm, nvalid, G, ya->C, N, s12a, stream);
        if (rc) return rc;
        rc = fgnn_gn_bwd_stats(db, ogstride, ldo, yb->ptr, yb->gstride, yb->ldp, yb->nrm, nvalid, G, yb->C, N, s12b, stream);
        if (rc) return rc;
    }
    return 0;
}
