// ColumnMaxPooling (models/layers.py:194-203, masked variant maskedtensor.py:213-228),
// siamese scoring (models/trainers.py:67) and triplet_loss (toolbox/losses.py:20-34)
// forward/backward for gfx950.  These are small (O(B*C*N^2)) next to the blocks.
#include <float.h>
#include "fgnn_common.h"
#include "fgnn_norm.h"

namespace {

// generic (N > 64): one wave per (g,c,i) row, lanes stride the columns (coalesced); the arg-max keeps the
// FIRST maximum (torch.max semantics): each lane scans its columns in ascending order, the wave reduction
// prefers the larger value and, on equal values, the smaller column.
__global__ __launch_bounds__(256) void colmax_fwd_kernel(const fgnn_slab y, const int *nvalid, int G, int N, float *e,
                                                         int *idx) {
    const long long t = (long long)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    const int C = y.C;
    if (t >= (long long)G * C * N) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(t % N);
    const int gc = (int)(t / N);
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    float best = 0.f;
    int bi = 0;
    if (i < nv) {
        const float *row = y.ptr + (long long)g * y.gstride + (long long)c * y.ldp + (long long)i * N;
        float mean = 0.f, a = 1.f, be = 0.f;
        if (y.nrm) {
            const float4 n = reinterpret_cast<const float4 *>(y.nrm)[gc];
            mean = n.x;
            a = n.y;
            be = y.beta ? y.beta[c] : 0.f;
        }
        best = -FLT_MAX;
        bi = 0x7fffffff;
        for (int j = lane; j < nv; j += WAVE) {
            float v = row[j];
            if (y.nrm) v = (v - mean) * a + be;
            if (v > best) {
                best = v;
                bi = j;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
    }
    if (lane == 0) {
        e[t] = best;
        idx[t] = bi;
    }
}

// 64 < N <= 256: 16 lanes per row, four rows per wave, four waves per workgroup; lane l of a row group scans the KC consecutive
// columns KC l .. KC l + KC - 1 (one or two 16-byte loads), the arg-max meets inside the row of 16 lanes by DPP row operations (no
// LDS): a quarter of the waves of the row-per-wave kernel and a third of its instructions per row.  Same tie rule.
template <int KC>
struct __attribute__((packed, aligned(4))) ColFloats {
    float v[KC];
};
template <int KC>
__global__ __launch_bounds__(256) void colmax_fwd_rows16_kernel(const fgnn_slab y, const int *nvalid, int G, int N, float *e, int *idx) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const long long t = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);      // row index over (g, c, i)
    const int C = y.C;
    const bool live = t < (long long)G * C * N;
    const long long tc = live ? t : 0;
    const int i = (int)(tc % N);
    const int gc = (int)(tc / N);
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    float best = 0.f;
    int bi = 0;
    if (live && i < nv) {                      // (uniform inside a row group of 16 lanes)
        const float *row = y.ptr + (long long)g * y.gstride + (long long)c * y.ldp + (long long)i * N;
        float mean = 0.f, a = 1.f, be = 0.f;
        if (y.nrm) {
            const float4 n = reinterpret_cast<const float4 *>(y.nrm)[gc];
            mean = n.x;
            a = n.y;
            be = y.beta ? y.beta[c] : 0.f;
        }
        const int j0 = KC * sub;
        float v[KC];
        if (j0 + KC <= N) {
            const ColFloats<KC> f = *reinterpret_cast<const ColFloats<KC> *>(row + j0);
#pragma unroll
            for (int k = 0; k < KC; ++k) v[k] = f.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < KC; ++k) v[k] = j0 + k < N ? row[j0 + k] : 0.f;
        }
        best = -FLT_MAX;
        bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const float x = y.nrm ? (v[k] - mean) * a + be : v[k];
            if (j0 + k < nv && x > best) {
                best = x;
                bi = j0 + k;
            }
        }
    }
    // arg-max over the 16 lanes of the row group: larger value wins, equal values -> smaller column
#define FGNN_COLMAX_STEP(CTRL)                                                                                         \
    {                                                                                                                  \
        const float ov = dpp_mov<CTRL>(best);                                                                          \
        const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, true);                                       \
        if (ov > best || (ov == best && oi < bi)) {                                                                    \
            best = ov;                                                                                                 \
            bi = oi;                                                                                                   \
        }                                                                                                              \
    }
    FGNN_COLMAX_STEP(0xB1)      // quad_perm [1,0,3,2]
    FGNN_COLMAX_STEP(0x4E)      // quad_perm [2,3,0,1]
    FGNN_COLMAX_STEP(0x141)     // row_half_mirror
    FGNN_COLMAX_STEP(0x140)     // row_mirror
#undef FGNN_COLMAX_STEP
    if (live && sub == 0) {
        e[t] = i < nv ? best : 0.f;
        idx[t] = i < nv ? bi : 0;
    }
}

// N <= 64: one wave per (g,c); the matrix is read with coalesced loads, normalised and staged in a
// wave-private LDS tile (row stride N+1: conflict-free row scans), then lane i scans row i.
// FIN: the record of the input does not exist yet; the wave finalizes it from the producer's tile statistics
// (the work of fgnn_gn_finalize for this (g,c), without its launch) and publishes it in y.nrm.
struct ColmaxFin {
    const float *part, *cnt, *gw;
    float eps;
    int tpg;
};
template <bool FIN>
__global__ __launch_bounds__(256) void colmax_fwd_lds_kernel(const fgnn_slab y, const int *nvalid, int G, int N,
                                                             float *e, int *idx, const ColmaxFin F) {
    __shared__ float sm[4][64 * 65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gc = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave;
    const int C = y.C;
    if (gc >= G * C) return;
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const float *mat = y.ptr + (long long)g * y.gstride + (long long)c * y.ldp;
    float mean = 0.f, a = 1.f, be = 0.f;
    TilePartials tp;
    if (FIN) tp = finalize_load(F.part, F.cnt, g, c, C, F.tpg, lane);
    // the whole matrix is requested before anything waits (lane = column, one load per row): a rolled loop over the
    // elements pays one memory round trip per eight loads (five for N = 50)
    float x[64];
    const int col = lane < N ? lane : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (8 * q < N) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = 8 * q + r;
                x[i] = mat[(i < N ? i : 0) * N + col];
            }
        }
    }
    if (FIN) {
        const float4 n = finalize_reduce(tp, (float)nv, F.gw ? F.gw[c] : 1.f, F.eps);
        if (lane == 0) reinterpret_cast<float4 *>(const_cast<float *>(y.nrm))[gc] = n;
        mean = n.x;
        a = n.y;
        be = y.beta ? y.beta[c] : 0.f;
    } else if (y.nrm) {
        const float4 n = reinterpret_cast<const float4 *>(y.nrm)[gc];
        mean = n.x;
        a = n.y;
        be = y.beta ? y.beta[c] : 0.f;
    }
    float *t = sm[wave];
    const int ld = N + 1;
    const bool norm = FIN || y.nrm;
    if (lane < N) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (8 * q < N) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = 8 * q + r;
                    if (i < N) t[i * ld + lane] = norm ? (x[i] - mean) * a + be : x[i];
                }
            }
        }
    }
    // same wave wrote and reads: LDS operations of a wave execute in order
    float best = 0.f;
    int bi = 0;
    if (lane < nv) {
        best = -FLT_MAX;
        const float *row = t + lane * ld;
        for (int jj = 0; jj < nv; ++jj) {
            const float v = row[jj];
            if (v > best) {
                best = v;
                bi = jj;
            }
        }
    }
    if (lane < N) {
        e[(long long)gc * N + lane] = best;
        idx[(long long)gc * N + lane] = bi;
    }
}

// one workgroup per (g,c): the rows' (argmax, d e) pairs are staged in LDS, then the workgroup walks the
// matrix linearly (coalesced stores: zeros + one scattered value per row) and accumulates the GraphNorm-
// backward sums S1 = sum de, S2 = sum de * (z[i, idx] - mean) (fixed-order reduction).
constexpr int CMB_MAXN = 1024;    // rows staged per workgroup; larger N falls back to per-element global reads
__global__ __launch_bounds__(256) void colmax_bwd_kernel(const float *de, const int *idx, const int *nvalid, int G, int C,
                                                         int N, float *dy, long long gstride, long long ldp,
                                                         const fgnn_slab y, float *s12) {
    __shared__ int sidx[CMB_MAXN];
    __shared__ float sde[CMB_MAXN];
    __shared__ float red[4][2];
    const int gc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    float *mat = dy + (long long)g * gstride + (long long)c * ldp;
    const float *zm = s12 ? y.ptr + (long long)g * y.gstride + (long long)c * y.ldp : nullptr;
    const float mean = s12 ? y.nrm[(long long)gc * 4] : 0.f;
    const bool staged = N <= CMB_MAXN;
    float s1 = 0.f, s2 = 0.f;
    // N <= 256 (one row per thread): the z value at the arg-max position -- a second, dependent memory round trip -- is
    // only requested here and consumed after the store sweep below, which does not depend on it
    const bool late = N <= 256;
    float zlate = 0.f, dlate = 0.f;
    for (int i = tid; i < N; i += 256) {
        const long long t = (long long)gc * N + i;
        const int bi = idx[t];
        const float d = i < nv ? de[t] : 0.f;
        if (staged) {
            sidx[i] = bi;
            sde[i] = d;
        }
        if (s12 && i < nv) {
            if (late) {
                zlate = zm[(long long)i * N + bi];
                dlate = d;
            } else {
                s1 += d;
                s2 += d * (zm[(long long)i * N + bi] - mean);
            }
        }
    }
    __syncthreads();
    const float invN = 1.f / (float)N;
    if (staged && (ldp & 3) == 0 && (gstride & 3) == 0 && (reinterpret_cast<unsigned long long>(dy) & 15ull) == 0) {
        // 16-byte stores: four consecutive elements of the channel (they may straddle a row boundary)
        const int P = N * N;
        float4 *mat4 = reinterpret_cast<float4 *>(mat);
        for (int q = tid; 4 * q < P; q += 256) {
            const int p0 = 4 * q;
            int i = (int)(((float)p0 + 0.5f) * invN);
            int j = p0 - i * N;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ic = i < N ? i : 0;
                v[k] = (i < N && j == sidx[ic]) ? sde[ic] : 0.f;
                if (++j == N) {
                    j = 0;
                    ++i;
                }
            }
            if (p0 + 3 < P) {
                mat4[q] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (p0 + k < P) mat[p0 + k] = v[k];
            }
        }
    } else {
        for (int p = tid; p < N * N; p += 256) {
            const int i = (int)(((float)p + 0.5f) * invN);
            const int j = p - i * N;
            const int bi = staged ? sidx[i] : idx[(long long)gc * N + i];
            const float d = staged ? sde[i] : (i < nv ? de[(long long)gc * N + i] : 0.f);
            mat[p] = (j == bi) ? d : 0.f;
        }
    }
    if (s12) {
        if (late && tid < nv) {
            s1 += dlate;
            s2 += dlate * (zlate - mean);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            red[wave][0] = s1;
            red[wave][1] = s2;
        }
        __syncthreads();
        if (tid < 2) s12[(long long)gc * 2 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// Contiguous global -> LDS copy by a 256-thread workgroup with up to eight 16-byte loads in flight per thread (a rolled
// element loop pays one memory round trip per element).
DEVI void copy_to_lds256(float *dst, const float *src, int n, int tid) {
    constexpr int U = 8;
    int done = 0;
    if (((reinterpret_cast<unsigned long long>(src) | reinterpret_cast<unsigned long long>(dst)) & 15ull) == 0) {
        const int n4 = n >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int e0 = tid; e0 < n4; e0 += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                v[k] = s4[e < n4 ? e : 0];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                if (e < n4) d4[e] = v[k];
            }
        }
        done = n4 << 2;
    }
    for (int e0 = done + tid; e0 < n; e0 += 256 * U) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + 256 * k;
            v[k] = src[e < n ? e : 0];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + 256 * k;
            if (e < n) dst[e] = v[k];
        }
    }
}

// grid (B, row_blocks): workgroup (b, rp) owns a contiguous block of rows of pair b.
// e1,e2: (C, N) each.  scores[i][j] = sum_c e1[c][i] e2[c][j]; lse_i; partial loss of its rows.
// LDS: e2 whole (C x N) and the workgroup's own rows of e1 (C x rows).
__global__ __launch_bounds__(256) void score_ce_fwd_kernel(const float *e1, const float *e2, const int *nvalid,
                                                           int C, int N, int rows, float *scores, float *lse,
                                                           float *pair_loss) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s2 = sm, *s1 = sm + (size_t)C * N;
    float *sS = s1 + (size_t)C * rows;   // the workgroup's score rows, [r][j]
    float *red = sS + (size_t)rows * N;  // 4 floats
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = nvalid_of(nvalid, b, N);
    const int i0 = blockIdx.y * rows;
    const int i1 = (i0 + rows < N) ? i0 + rows : N;
    const int w = i1 > i0 ? i1 - i0 : 0;
    const float *p1 = e1 + (long long)b * C * N, *p2 = e2 + (long long)b * C * N;
    copy_to_lds256(s2, p2, C * N, tid);
    for (int e = tid; e < C * w; e += 256) {
        const int c = e / w, r = e - c * w;
        s1[c * rows + r] = p1[c * N + i0 + r];
    }
    __syncthreads();
    float *S = scores + (long long)b * N * N;
    const float invN = 1.f / (float)N;
    for (int e = tid; e < w * N; e += 256) {
        const int r = (int)(((float)e + 0.5f) * invN), j = e - r * N;
        float acc = 0.f;
        if (i0 + r < nv && j < nv)
            for (int c = 0; c < C; ++c) acc = fmaf(s1[c * rows + r], s2[c * N + j], acc);
        S[(long long)i0 * N + e] = acc;
        sS[e] = acc;
    }
    __syncthreads();
    float wl = 0.f;
    for (int i = i0 + wave; i < i1; i += 4) {
        float l = 0.f;
        const float *Sr = sS + (size_t)(i - i0) * N;      // the row pass reads the LDS copy, not the global one
        if (i < nv) {
            float mx = -FLT_MAX;
            for (int j = lane; j < nv; j += WAVE) mx = fmaxf(mx, Sr[j]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < nv; j += WAVE) se += expf(Sr[j] - mx);
            se = wave_sum(se);
            l = mx + logf(se);
            wl += l - Sr[i];
        }
        if (lane == 0 && lse) lse[(long long)b * N + i] = l;
    }
    if (lane == 0) red[wave] = wl;
    __syncthreads();
    if (tid == 0 && pair_loss) pair_loss[b * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dS[i][j] = (exp(S-lse_i) - [i==j]) * gscale (CE mode) or the given dscores (plain mode);
// de1[c][i] = sum_j e2[c][j] dS[i][j];  de2[c][j] = sum_i e1[c][i] dS[i][j].
// grid (B, splits): every workgroup stages dS (N x N) once in LDS and handles C / gridDim.y channels.
constexpr int CSPLIT = 4;          // channel splits of the blocked kernel and of large batches
constexpr int CSPLIT_SMALL = 8;    // whole-dS staging at small batch (B * 4 < 256 workgroups)
template <bool CE, bool STAGE>
__global__ __launch_bounds__(256) void score_bwd_kernel(const float *e1, const float *e2, const float *scores,
                                                        const float *lse, const float *dscores, const int *nvalid,
                                                        const float *gscale, int C, int N, float *de1, float *de2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cper = (C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c0 = blockIdx.y * cper;
    const int cn = (c0 + cper <= C ? cper : (C > c0 ? C - c0 : 0));
    float *s1 = sm, *s2 = sm + (size_t)cper * N;
    float *dS = s2 + (size_t)cper * N;         // [i][j], row stride N + 1
    const int ld = N + 1;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, b, N);
    const float *p1 = e1 + ((long long)b * C + c0) * N, *p2 = e2 + ((long long)b * C + c0) * N;
    for (int e = tid; e < cn * N; e += 256) {
        s1[e] = p1[e];
        s2[e] = p2[e];
    }
    const float gs = CE ? *gscale : 1.f;
    const float *S = (CE ? scores : dscores) + (long long)b * N * N;
    const float *Lr = CE ? lse + (long long)b * N : nullptr;
    const float invN = 1.f / (float)N;
    auto ds_at = [&](int i, int jj) -> float {     // valid i, jj only
        float d = S[(long long)i * N + jj];
        if (CE) d = (expf(d - Lr[i]) - (i == jj ? 1.f : 0.f)) * gs;
        return d;
    };
    if (STAGE) {
        // eight score loads in flight per thread and pass (the row's lse comes from the L1 / L2 resident vector)
        constexpr int U = 8;
        for (int e0 = tid; e0 < N * N; e0 += 256 * U) {
            float v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                v[k] = S[e < N * N ? e : 0];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                if (e < N * N) {
                    const int i = (int)(((float)e + 0.5f) * invN);
                    const int jj = e - i * N;
                    float d = 0.f;
                    if (i < nv && jj < nv) {
                        d = v[k];
                        if (CE) d = (expf(d - Lr[i]) - (i == jj ? 1.f : 0.f)) * gs;
                    }
                    dS[i * ld + jj] = d;
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < cn * N; e += 256) {
        const int c = e / N, i = e - c * N;
        float a1 = 0.f, a2 = 0.f;
        if (i < nv) {
            const float *r2 = s2 + c * N, *r1 = s1 + c * N;
            for (int jj = 0; jj < nv; ++jj) {
                a1 = fmaf(r2[jj], STAGE ? dS[i * ld + jj] : ds_at(i, jj), a1);
                a2 = fmaf(r1[jj], STAGE ? dS[jj * ld + i] : ds_at(jj, i), a2);
            }
        }
        de1[((long long)b * C + c0) * N + e] = a1;
        de2[((long long)b * C + c0) * N + e] = a2;
    }
}

// Scoring forward + triplet loss + their backward in ONE launch (a training step issues them back to back; N <= 64, 4 | C): grid
// (B, CSPLIT_SMALL).  Every workgroup of pair b stages e1, e2, forms the full score matrix S in LDS (N^2 C fma: cheaper than a second
// launch), the row log-sum-exps and dS = (exp(S - lse) - I) gscale, then produces de1 / de2 of ITS channels; workgroup y = 0 also
// writes scores, lse and the pair-loss partials.  Arithmetic and summation orders are those of score_ce_fwd_kernel +
// score_bwd_kernel<true, true>: results are bit-identical to the two launches (tests/test_gpu_kernels.py).
constexpr int STEP_THREADS = 1024;       // 16 waves: the row pass (one wave per row, as in score_ce_fwd_kernel) is 4 rows deep at N = 50
__global__ __launch_bounds__(STEP_THREADS) void score_ce_step_kernel(const float *e1, const float *e2, const int *nvalid,
                                                                     const float *gscale, int C, int N, int row_blocks, float *scores,
                                                                     float *lse, float *pair_loss, float *de1, float *de2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s1 = sm, *s2 = sm + (size_t)C * N;
    float *dS = s2 + (size_t)C * N;            // S, then dS: [i][j], row stride N + 1
    const int ld = N + 1;
    float *term = dS + (size_t)N * ld;         // per row: lse_i - S_ii (0 beyond nv)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = STEP_THREADS, NWV = STEP_THREADS / 64;
    const int nv = nvalid_of(nvalid, b, N);
    const bool writer = blockIdx.y == 0;
    {
        const float *p1 = e1 + (long long)b * C * N, *p2 = e2 + (long long)b * C * N;
        for (int e = tid; e < C * N; e += NT) {
            s1[e] = p1[e];
            s2[e] = p2[e];
        }
    }
    const float gs = *gscale;
    __syncthreads();
    float *S = scores + (long long)b * N * N;
    const float invN = 1.f / (float)N;
    for (int e = tid; e < N * N; e += NT) {
        const int i = (int)(((float)e + 0.5f) * invN), j = e - i * N;
        float acc = 0.f;
        if (i < nv && j < nv)
            for (int c = 0; c < C; ++c) acc = fmaf(s1[c * N + i], s2[c * N + j], acc);
        if (writer) S[e] = acc;
        dS[i * ld + j] = acc;
    }
    __syncthreads();
    for (int i = wave; i < N; i += NWV) {      // one wave per row, the arithmetic of score_ce_fwd_kernel
        float l = 0.f, t = 0.f;
        float *Sr = dS + (size_t)i * ld;
        if (i < nv) {
            float mx = -FLT_MAX;
            for (int j = lane; j < nv; j += WAVE) mx = fmaxf(mx, Sr[j]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < nv; j += WAVE) se += expf(Sr[j] - mx);
            se = wave_sum(se);
            l = mx + logf(se);
            t = l - Sr[i];
        }
        // the row becomes dS in place (every lane has finished reading it: wave_sum is a full-wave exchange)
        for (int j = lane; j < N; j += WAVE) {
            float d = 0.f;
            if (i < nv && j < nv) d = (expf(Sr[j] - l) - (i == j ? 1.f : 0.f)) * gs;
            Sr[j] = d;
        }
        if (lane == 0) {
            term[i] = t;
            if (writer && lse) lse[(long long)b * N + i] = l;
        }
    }
    __syncthreads();
    if (writer && pair_loss && tid < row_blocks) {
        // the partial sums of score_ce_fwd_kernel's workgroup (b, tid): wave w of it summed the rows i0 + w, i0 + w + 4, ...
        const int rows = (N + row_blocks - 1) / row_blocks;
        const int i0 = tid * rows, i1 = (i0 + rows < N) ? i0 + rows : N;
        float wl[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wl[w] = 0.f;
            for (int i = i0 + w; i < i1; i += 4) wl[w] += term[i];
        }
        pair_loss[b * row_blocks + tid] = (wl[0] + wl[1]) + (wl[2] + wl[3]);
    }
    // ---- de1, de2 of this workgroup's channels (score_bwd_kernel) ----
    const int cper = (C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c0 = blockIdx.y * cper;
    const int cn = (c0 + cper <= C ? cper : (C > c0 ? C - c0 : 0));
    for (int e = tid; e < 2 * cn * N; e += NT) {       // (the two gradients of an element on two threads)
        const int which = e >= cn * N, ee = which ? e - cn * N : e;
        const int c = ee / N, i = ee - c * N;
        float a = 0.f;
        if (i < nv) {
            const float *r = (which ? s1 : s2) + (c0 + c) * N;
            if (which) for (int jj = 0; jj < nv; ++jj) a = fmaf(r[jj], dS[jj * ld + i], a);
            else for (int jj = 0; jj < nv; ++jj) a = fmaf(r[jj], dS[i * ld + jj], a);
        }
        (which ? de2 : de1)[((long long)b * C + c0) * N + ee] = a;
    }
}

// Large N (dS does not fit LDS): grid (B, CSPLIT, ceil(N / SB_BLK)).  Workgroup (b, cs, rb) stages the row block
// dS[i0:i1, :] and produces de1 for those rows, then stages the column block dS[:, i0:i1] and produces de2 for
// those columns -- no cross-workgroup reduction, nothing recomputed inside the inner loops.
template <bool CE>
__global__ __launch_bounds__(256) void score_bwd_blocked_kernel(const float *e1, const float *e2, const float *scores,
                                                                const float *lse, const float *dscores,
                                                                const int *nvalid, const float *gscale, int C, int N,
                                                                int SB_BLK, float *de1, float *de2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cper = (C + CSPLIT - 1) / CSPLIT;
    const int c0 = blockIdx.y * cper;
    const int cn = (c0 + cper <= C ? cper : (C > c0 ? C - c0 : 0));
    float *s1 = sm, *s2 = sm + (size_t)cper * N;
    float *dS = s2 + (size_t)cper * N;         // phase 1: [SB_BLK][N + 1], phase 2: [N][SB_BLK + 1]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, b, N);
    const int x0 = blockIdx.z * SB_BLK, x1 = (x0 + SB_BLK < N) ? x0 + SB_BLK : N, w = x1 - x0;
    const float *p1 = e1 + ((long long)b * C + c0) * N, *p2 = e2 + ((long long)b * C + c0) * N;
    // every staging loop below keeps U loads in flight per thread: rolled, each element pays its own memory round trip
    // (N = 200, B = 8: about thirty in a row per workgroup, the whole 20 us of the launch)
    constexpr int U = 8;
    copy_to_lds256(s1, p1, cn * N, tid);
    copy_to_lds256(s2, p2, cn * N, tid);
    const float gs = CE ? *gscale : 1.f;
    const float *S = (CE ? scores : dscores) + (long long)b * N * N;
    const float *Lr = CE ? lse + (long long)b * N : nullptr;
    // dS[i][jj] from the loaded score (and the row's lse); i, jj < N
    auto ds_of = [&](float sv, float lv, int i, int jj) -> float {
        if (i >= nv || jj >= nv) return 0.f;
        return CE ? (expf(sv - lv) - (i == jj ? 1.f : 0.f)) * gs : sv;
    };
    // ---- phase 1: rows x0..x1 ----
    {
        const int ld = N + 1;
        for (int e0 = tid; e0 < w * N; e0 += 256 * U) {
            float sv[U], lv[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k, ec = e < w * N ? e : 0;
                const int r = ec / N;
                sv[k] = S[(long long)x0 * N + ec];
                lv[k] = CE ? Lr[x0 + r] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                if (e < w * N) {
                    const int r = e / N, jj = e - r * N;
                    dS[r * ld + jj] = ds_of(sv[k], lv[k], x0 + r, jj);
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < cn * w; e += 256) {
            const int c = e / w, r = e - c * w;
            float a1 = 0.f;
            const float *r2 = s2 + c * N, *drow = dS + r * ld;
            for (int jj = 0; jj < nv; ++jj) a1 = fmaf(r2[jj], drow[jj], a1);
            de1[((long long)b * C + c0 + c) * N + x0 + r] = (x0 + r) < nv ? a1 : 0.f;
        }
        __syncthreads();
    }
    // ---- phase 2: columns x0..x1 ----
    {
        const int ld = SB_BLK + 1;
        for (int e0 = tid; e0 < N * w; e0 += 256 * U) {
            float sv[U], lv[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k, ec = e < N * w ? e : 0;
                const int i = ec / w, r = ec - i * w;
                sv[k] = S[(long long)i * N + x0 + r];
                lv[k] = CE ? Lr[i] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + 256 * k;
                if (e < N * w) {
                    const int i = e / w, r = e - i * w;
                    dS[i * ld + r] = ds_of(sv[k], lv[k], i, x0 + r);
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < cn * w; e += 256) {
            const int c = e / w, r = e - c * w;
            float a2 = 0.f;
            const float *r1 = s1 + c * N;
            for (int i = 0; i < nv; ++i) a2 = fmaf(r1[i], dS[i * ld + r], a2);
            de2[((long long)b * C + c0 + c) * N + x0 + r] = (x0 + r) < nv ? a2 : 0.f;
        }
    }
}

// triplet_loss pieces on a given score tensor (module-level API): one workgroup per pair.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float *scores, const int *nvalid, int N, float *lse,
                                                     float *pair_loss) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = nvalid_of(nvalid, b, N);
    const float *S = scores + (long long)b * N * N;
    float wl = 0.f;
    for (int i = wave; i < N; i += 4) {
        float l = 0.f;
        if (i < nv) {
            float mx = -FLT_MAX;
            for (int j = lane; j < nv; j += WAVE) mx = fmaxf(mx, S[(long long)i * N + j]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < nv; j += WAVE) se += expf(S[(long long)i * N + j] - mx);
            se = wave_sum(se);
            l = mx + logf(se);
            wl += l - S[(long long)i * N + i];
        }
        if (lane == 0 && lse) lse[(long long)b * N + i] = l;
    }
    if (lane == 0) red[wave] = wl;
    __syncthreads();
    if (tid == 0 && pair_loss) pair_loss[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void ce_bwd_kernel(const float *scores, const float *lse, const int *nvalid, const float *gscale, int N,
                              float *dscores) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int nv = nvalid_of(nvalid, b, N);
    const int i = e / N, j = e - i * N;
    float d = 0.f;
    if (i < nv && j < nv)
        d = (expf(scores[(long long)b * N * N + e] - lse[(long long)b * N + i]) - (i == j ? 1.f : 0.f)) * (*gscale);
    dscores[(long long)b * N * N + e] = d;
}

}  // namespace

extern "C" int fgnn_colmax_fwd(const fgnn_slab *y, const int *nvalid, int G, int N, float *e, int *idx, void *stream) {
    FGNN_CHECK(y && y->ptr && e && idx && G > 0 && N > 0 && y->C > 0, "fgnn_colmax_fwd: bad arguments");
    const long long tot = (long long)G * y->C * N;
    if (N <= 64) {
        hipLaunchKernelGGL(colmax_fwd_lds_kernel<false>, dim3((unsigned)((G * y->C + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, *y, nvalid, G, N, e, idx, ColmaxFin{});
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    if (N <= 128) {
        hipLaunchKernelGGL(colmax_fwd_rows16_kernel<8>, dim3((unsigned)((tot + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *y, nvalid, G, N, e, idx);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    if (N <= 256) {
        hipLaunchKernelGGL(colmax_fwd_rows16_kernel<16>, dim3((unsigned)((tot + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *y, nvalid, G, N, e, idx);
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(colmax_fwd_kernel, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *y,
                       nvalid, G, N, e, idx);
    FGNN_LAUNCH_CHECK();
    return 0;
}

// (a workgroup-per-plane kernel with the finalize in its prologue was measured for 64 < N <= 256: 29.2 us against 15.7 + 4.9 us for
// the row-per-wave kernel below plus fgnn_gn_finalize at N = 120, 16 graphs -- the looped finalize of 450 tiles serialises in front
// of every plane)
extern "C" int fgnn_colmax_fwd_fin_supported(int N) { return N <= 64 ? 1 : 0; }

extern "C" int fgnn_colmax_fwd_fin(const fgnn_slab *y, const float *part, const float *cnt, const float *gn_weight, float eps,
                                   const int *nvalid, int G, int N, float *e, int *idx, void *stream) {
    return fgnn_colmax_fwd_fin_r(y, part, cnt, gn_weight, eps, nvalid, G, N, fgnn_tiles_per_graph(N), e, idx, stream);
}
// _r: `recs` statistics records per graph (fgnn_mlp_fwd_t16: one per 16-pixel half)
extern "C" int fgnn_colmax_fwd_fin_r(const fgnn_slab *y, const float *part, const float *cnt, const float *gn_weight, float eps,
                                     const int *nvalid, int G, int N, int recs, float *e, int *idx, void *stream) {
    FGNN_CHECK(y && y->ptr && y->nrm && part && cnt && e && idx && G > 0 && N > 0 && y->C > 0, "fgnn_colmax_fwd_fin: bad arguments");
    FGNN_CHECK(N <= 64, "fgnn_colmax_fwd_fin: N=%d > 64 (use fgnn_gn_finalize + fgnn_colmax_fwd)", N);
    FGNN_CHECK(recs > 0, "fgnn_colmax_fwd_fin: recs");
    const ColmaxFin F = {part, cnt, gn_weight, eps, recs};
    hipLaunchKernelGGL(colmax_fwd_lds_kernel<true>, dim3((unsigned)((G * y->C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *y,
                       nvalid, G, N, e, idx, F);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_colmax_bwd(const float *de, const int *idx, const int *nvalid, int G, int C, int N, float *dy,
                               long long gstride, long long ldp, const fgnn_slab *y, float *s12, void *stream) {
    FGNN_CHECK(de && idx && dy && G > 0 && C > 0 && N > 0, "fgnn_colmax_bwd: bad arguments");
    FGNN_CHECK(!s12 || (y && y->ptr && y->nrm && y->C == C), "fgnn_colmax_bwd: s12 needs the normalised input slab");
    fgnn_slab ys = {};
    if (s12) ys = *y;
    const int tot = G * C;
    hipLaunchKernelGGL(colmax_bwd_kernel, dim3((unsigned)tot), dim3(256), 0, (hipStream_t)stream, de,
                       idx, nvalid, G, C, N, dy, gstride, ldp, ys, s12);
    FGNN_LAUNCH_CHECK();
    return 0;
}

static int score_lds_bytes(int C, int N) { return (2 * C * N + 4) * (int)sizeof(float); }
static int score_bwd_lds_bytes(int C, int N, bool stage, int csplit = CSPLIT) {
    const int cper = (C + csplit - 1) / csplit;
    return (2 * cper * N + (stage ? N * (N + 1) : 0)) * (int)sizeof(float);
}
template <bool CE>
static int launch_score_bwd(const float *e1, const float *e2, const float *scores, const float *lse,
                            const float *dscores, const int *nvalid, const float *gscale, int B, int C, int N,
                            float *de1, float *de2, hipStream_t st) {
    // whole-dS staging gives only B x CSPLIT workgroups: with few large pairs (N > 64) the blocked kernel fills the chip better
    const bool stage = score_bwd_lds_bytes(C, N, true) <= 160 * 1024 && (N <= 64 || (long long)B * CSPLIT >= 256);
    const int csplit = (long long)B * CSPLIT < 256 ? CSPLIT_SMALL : CSPLIT;
    const int lds = score_bwd_lds_bytes(C, N, stage, stage ? csplit : CSPLIT);
    FGNN_CHECK(lds <= 160 * 1024, "score backward: C*N=%d too large for LDS staging", C * N);
    if (stage) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)score_bwd_kernel<CE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((score_bwd_kernel<CE, true>), dim3(B, csplit), dim3(256), lds, st, e1, e2, scores, lse, dscores,
                           nvalid, gscale, C, N, de1, de2);
    } else {
        const int cper = (C + CSPLIT - 1) / CSPLIT;
        // 16-wide blocks when the batch is small (N = 200, B = 8: 8 x 4 x 13 workgroups instead of 8 x 4 x 4)
        const int SB_BLK = (long long)B * CSPLIT * ((N + 63) / 64) >= 512 ? 64 : 16;
        const int big = (N + 1) * SB_BLK > N * (SB_BLK + 1) ? (N + 1) * SB_BLK : N * (SB_BLK + 1);
        const int lds2 = (2 * cper * N + big) * (int)sizeof(float);
        FGNN_CHECK(lds2 <= 160 * 1024, "score backward: N=%d too large for the blocked LDS staging", N);
        if (lds2 > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)score_bwd_blocked_kernel<CE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        hipLaunchKernelGGL((score_bwd_blocked_kernel<CE>), dim3(B, CSPLIT, (N + SB_BLK - 1) / SB_BLK), dim3(256), lds2, st, e1, e2,
                           scores, lse, dscores, nvalid, gscale, C, N, SB_BLK, de1, de2);
    }
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_score_row_blocks(int B, int N) {
    // few large pairs: more row blocks than FGNN_SCORE_SPLIT so that the launch fills the chip (8 rows per workgroup)
    if (B * FGNN_SCORE_SPLIT >= 512) return FGNN_SCORE_SPLIT;
    return (N + 3) / 4;                    // small batch: one row per wave, B * N / 4 workgroups
}

extern "C" int fgnn_score_ce_fwd_blocks(const float *e1, const float *e2, const int *nvalid, int B, int C, int N,
                                        int row_blocks, float *scores, float *lse, float *pair_loss, void *stream) {
    FGNN_CHECK(e1 && e2 && scores && B > 0 && C > 0 && N > 0 && row_blocks > 0, "fgnn_score_ce_fwd: bad arguments");
    const int rows = (N + row_blocks - 1) / row_blocks;
    const int lds = (C * N + C * rows + rows * N + 4) * (int)sizeof(float);
    FGNN_CHECK(lds <= 160 * 1024, "fgnn_score_ce_fwd: C*N=%d too large for LDS staging", C * N);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)score_ce_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(score_ce_fwd_kernel, dim3(B, row_blocks), dim3(256), lds, (hipStream_t)stream, e1, e2, nvalid, C, N, rows,
                       scores, lse, pair_loss);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_score_ce_fwd(const float *e1, const float *e2, const int *nvalid, int B, int C, int N,
                                 float *scores, float *lse, float *pair_loss, void *stream) {
    return fgnn_score_ce_fwd_blocks(e1, e2, nvalid, B, C, N, FGNN_SCORE_SPLIT, scores, lse, pair_loss, stream);
}

extern "C" int fgnn_score_ce_bwd(const float *e1, const float *e2, const float *scores, const float *lse,
                                 const int *nvalid, const float *gscale, int B, int C, int N, float *de1, float *de2,
                                 void *stream) {
    FGNN_CHECK(e1 && e2 && scores && lse && gscale && de1 && de2 && B > 0, "fgnn_score_ce_bwd: bad arguments");
    return launch_score_bwd<true>(e1, e2, scores, lse, nullptr, nvalid, gscale, B, C, N, de1, de2, (hipStream_t)stream);
}

extern "C" int fgnn_score_ce_step_supported(int B, int C, int N) {
    // the shapes where score_bwd_kernel stages the whole dS with CSPLIT_SMALL channel splits (small batches of small graphs)
    return (N <= 64 && (long long)B * CSPLIT < 256 && C % CSPLIT_SMALL == 0) ? 1 : 0;
}

extern "C" int fgnn_score_ce_step(const float *e1, const float *e2, const int *nvalid, const float *gscale, int B, int C, int N,
                                  int row_blocks, float *scores, float *lse, float *pair_loss, float *de1, float *de2, void *stream) {
    FGNN_CHECK(e1 && e2 && gscale && scores && lse && pair_loss && de1 && de2 && B > 0 && row_blocks > 0 && row_blocks <= 256,
               "fgnn_score_ce_step: bad arguments");
    FGNN_CHECK(fgnn_score_ce_step_supported(B, C, N), "fgnn_score_ce_step: built for N <= 64, B < 64, C %% 8 == 0 (got B=%d C=%d N=%d); use "
               "fgnn_score_ce_fwd_blocks + fgnn_score_ce_bwd", B, C, N);
    const int lds = (2 * C * N + N * (N + 1) + N + 4) * (int)sizeof(float);
    hipLaunchKernelGGL(score_ce_step_kernel, dim3(B, CSPLIT_SMALL), dim3(STEP_THREADS), lds, (hipStream_t)stream, e1, e2, nvalid, gscale, C, N,
                       row_blocks, scores, lse, pair_loss, de1, de2);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_score_bwd(const float *e1, const float *e2, const float *dscores, const int *nvalid, int B, int C,
                              int N, float *de1, float *de2, void *stream) {
    FGNN_CHECK(e1 && e2 && dscores && de1 && de2 && B > 0, "fgnn_score_bwd: bad arguments");
    return launch_score_bwd<false>(e1, e2, nullptr, nullptr, dscores, nvalid, nullptr, B, C, N, de1, de2, (hipStream_t)stream);
}

extern "C" int fgnn_ce_fwd(const float *scores, const int *nvalid, int B, int N, float *lse, float *pair_loss,
                           void *stream) {
    FGNN_CHECK(scores && lse && pair_loss && B > 0 && N > 0, "fgnn_ce_fwd: bad arguments");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, nvalid, N, lse, pair_loss);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_ce_bwd(const float *scores, const float *lse, const int *nvalid, const float *gscale, int B, int N,
                           float *dscores, void *stream) {
    FGNN_CHECK(scores && lse && gscale && dscores && B > 0 && N > 0, "fgnn_ce_bwd: bad arguments");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((N * N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, scores, lse,
                       nvalid, gscale, N, dscores);
    FGNN_LAUNCH_CHECK();
    return 0;
}
