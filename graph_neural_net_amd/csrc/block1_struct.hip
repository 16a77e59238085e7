// Block 1 of the 2-FGNN on its STRUCTURED input (bit-packed adjacency; constant-size and ragged batches, N <= 128).
//
// The reference feeds block 1 the tensor representation of a graph (loaders/data_generator.py:118-125): channel 0 = the 0/1
// adjacency W, channel 1 = diag(row sums).  So the input pixel (i, j) takes one of few values -- off the diagonal (w_ij, 0)
// with w in {0, 1}, on it (w_ii, deg_i) -- and so do the outputs of mlp1 / mlp2 (models/blocks_emb.py:16-27: two
// MlpBlock_Real on that input): a table of 2 + 2 (N + 1) rows per model instead of G N^2 pixels through three convs.  With
// the normalised class values u0 / u1 / e_i (mlp1) and v0 / v1 / f_j (mlp2) of a graph,
//     Y1_c = u0 J + p W + diag(q),   p = u1 - u0, q_i = e_i - u0 - p w_ii        (and Y2_c = v0 J + r W + diag(s) likewise)
// and the per-channel product mult_c = Y1_c Y2_c (models/layers.py:161-162) has the closed form
//     mult_c[i][j] = u0 v0 n + u0 r degc_j + u0 s_j + p v0 degr_i + p r (W^2)_ij + p w_ij s_j + v0 q_i + r q_i w_ij + [i = j] q_i s_i
// with (W^2)_ij = popcount(row_i & column_j) on the bit rows: ONE integer product per pixel instead of 32 N x N products per
// graph.  GraphNorm statistics (models/layers.py:68-80) follow from the class counts.  In the backward direction the gradient
// reaches the parameters of mlp1 / mlp2 only through the class values, so what is needed of dY1_c = dM_c Y2_c^T and dY2_c =
// Y1_c^T dM_c are their CLASS SUMS, which reduce to row / column / diagonal / masked sums of dM_c and <dM_c, W^2>: one pass
// over d(mult).  The GraphNorm backward and the conv / ReLU chain then run per class (the ReLU masks are class constants, the
// chain is linear in the summed gradient), a few dozen 32-vectors per graph.
//
// Replaces, for block 1 only: fgnn_mlp_fwd (mlp1 + mlp2) + fgnn_chan_matmul_fwd in the forward direction and
// fgnn_chan_matmul_bwd + fgnn_mlp_bwd_pair in the backward direction.  Same function, another evaluation order: results agree
// with the generic kernels to fp32 rounding (tests/test_gpu_struct.py), not bit for bit.
// Ragged batches: graph g has nv = nvalid[g] vertices inside the N x N padded planes; everything below runs on the valid corner
// (rows / bits >= nv are cleared on load, class counts and the GraphNorm n use nv), mult is written as 0 outside it.
// The kernels are instantiated for bit rows of one (N <= 64) and two (N <= 128) 64-bit words.
#include "fgnn_common.h"
#include "fgnn_norm.h"

namespace {

constexpr int SB_CG = 8;                 // channel groups of the forward kernel (4 channels each)
constexpr int SB_CPG = FGNN_H / SB_CG;   // channels per group
constexpr int SB_TAB = 3 * FGNN_H;       // floats per table row: h1 | h2 | z
constexpr int SB_NMAX = 128;

// class of a pixel's input value: 0 = off-diagonal, w = 0; 1 = off-diagonal, w = 1; 2 + 2 deg + w_ii = diagonal
DEVI int sb_classes(int N) { return 2 + 2 * (N + 1); }

typedef unsigned long long u64;

// ---- K0: the class tables of one model, graph independent: tab[m][class][h1 | h2 | z][32] -------------------------------
struct TabArgs {
    const float *W[2][3];
    const float *b[2][3];
};
__global__ __launch_bounds__(64) void sb_tables_kernel(const TabArgs A, const int N, float *tab) {
    __shared__ float hbuf[2][FGNN_H];
    const int cls = blockIdx.x, m = blockIdx.y, o = threadIdx.x;
    float x0, x1;
    if (cls < 2) {
        x0 = (float)cls;
        x1 = 0.f;
    } else {
        x0 = (float)((cls - 2) & 1);
        x1 = (float)((cls - 2) >> 1);
    }
    float *out = tab + ((long long)m * sb_classes(N) + cls) * SB_TAB;
    if (o < FGNN_H) {
        const float *W0 = A.W[m][0];
        float a = A.b[m][0][o];
        a = fmaf(W0[o * 2 + 0], x0, a);
        a = fmaf(W0[o * 2 + 1], x1, a);
        a = fmaxf(a, 0.f);
        hbuf[0][o] = a;
        out[o] = a;
    }
    __syncthreads();
    if (o < FGNN_H) {
        const float *W1 = A.W[m][1] + o * FGNN_H;
        float a = A.b[m][1][o];
#pragma unroll
        for (int c = 0; c < FGNN_H; ++c) a = fmaf(W1[c], hbuf[0][c], a);
        a = fmaxf(a, 0.f);
        hbuf[1][o] = a;
        out[FGNN_H + o] = a;
    }
    __syncthreads();
    if (o < FGNN_H) {
        const float *W2 = A.W[m][2] + o * FGNN_H;
        float a = A.b[m][2][o];
#pragma unroll
        for (int c = 0; c < FGNN_H; ++c) a = fmaf(W2[c], hbuf[1][c], a);
        out[2 * FGNN_H + o] = a;
    }
}

// per-graph bit structure in LDS (NWD 64-bit words per row: N <= 64 NWD), filled by the first 64 NWD threads of a group
template <int NWD>
struct GraphBits {
    static constexpr int MAXN = 64 * NWD;
    u64 row[MAXN][NWD], col[MAXN][NWD];
    float degr[MAXN], degc[MAXN];
    int wii[MAXN], cls[MAXN];
};
// row t of graph g restricted to the valid corner: bits >= nv cleared, rows >= nv empty
template <int NWD>
DEVI void sb_bits_rows(GraphBits<NWD> &B, const unsigned *bits, int g, int t, int N, int nv) {
    if (t < 64 * NWD) {
        const int words = (N + 31) / 32;
        const unsigned *r = bits + ((long long)g * N + (t < N ? t : 0)) * words;
#pragma unroll
        for (int w = 0; w < NWD; ++w) {
            u64 v = 0ull;
            if (t < nv) {
                if (2 * w < words) v = r[2 * w];
                if (2 * w + 1 < words) v |= (u64)r[2 * w + 1] << 32;
                const int left = nv - 64 * w;                      // valid bits in this word
                v = left >= 64 ? v : (left <= 0 ? 0ull : (v & ((1ull << left) - 1ull)));
            }
            B.row[t][w] = v;
        }
    }
}
template <int NWD>
DEVI void sb_bits_vertex(GraphBits<NWD> &B, int t) {           // what needs the vertex's own row only
    int dr = 0;
#pragma unroll
    for (int w = 0; w < NWD; ++w) dr += __popcll(B.row[t][w]);
    B.degr[t] = (float)dr;
    const int wself = (int)((B.row[t][t >> 6] >> (t & 63)) & 1ull);
    B.wii[t] = wself;
    B.cls[t] = 2 + 2 * dr + wself;
}
template <int NWD>
DEVI void sb_bits_cols(GraphBits<NWD> &B, int t, int nv) {
    if (t < 64 * NWD) {
        u64 c[NWD];
#pragma unroll
        for (int w = 0; w < NWD; ++w) c[w] = 0ull;
        const int tw = t >> 6, tb = t & 63;
        for (int i0 = 0; i0 < nv; i0 += 8) {              // eight rows per LDS round trip (rows >= nv are empty)
            u64 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = B.row[(i0 + k) & (64 * NWD - 1)][tw];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = (i0 + k) & (64 * NWD - 1);
                c[NWD == 1 ? 0 : (i >> 6)] |= ((r[k] >> tb) & 1ull) << (i & 63);
            }
        }
        int dc = 0;
#pragma unroll
        for (int w = 0; w < NWD; ++w) {
            B.col[t][w] = c[w];
            dc += __popcll(c[w]);
        }
        B.degc[t] = (float)dc;
        sb_bits_vertex(B, t);
    }
}
template <int NWD>
DEVI int sb_w2(const GraphBits<NWD> &B, int i, int j) {        // (W^2)_ij = |{k: w_ik = w_kj = 1}|
    int s = 0;
#pragma unroll
    for (int w = 0; w < NWD; ++w) s += __popcll(B.row[i][w] & B.col[j][w]);
    return s;
}
template <int NWD>
DEVI bool sb_w(const GraphBits<NWD> &B, int i, int j) { return (B.row[i][j >> 6] >> (j & 63)) & 1ull; }

// The normalised class values of channel c of one model in one graph.  The 64 lanes of a wave call this together; lane l
// owns the vertices l + 64 k.  Statistics from the class counts (two-pass: mean, then squared deviations), record as
// fgnn_norm.h, y = (z - mean) a + beta as every consumer of a slab evaluates it.
template <int NWD>
struct ClassVals {
    float u0, p, q[NWD];     // Y = u0 J + p W + diag(q): q[k] belongs to vertex lane + 64 k
    float4 rec;              // GraphNorm record of (g, c)
};
template <int NWD>
DEVI ClassVals<NWD> sb_class_values(const float *tab_m, const GraphBits<NWD> &B, int nv, int c, float gnw, float beta, float eps, int lane) {
    const float z0 = tab_m[0 * SB_TAB + 2 * FGNN_H + c], z1 = tab_m[1 * SB_TAB + 2 * FGNN_H + c];
    float zd[NWD], on1 = 0.f, szd = 0.f;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const int v = lane + 64 * k;
        const bool on = v < nv;
        zd[k] = on ? tab_m[B.cls[on ? v : 0] * SB_TAB + 2 * FGNN_H + c] : 0.f;
        on1 += on ? B.degr[v] - (float)B.wii[v] : 0.f;
        szd += zd[k];
    }
    const float ones = wave_sum(on1);                                    // off-diagonal ones
    const float fN = (float)nv, m = fN * fN;
    const float n1 = ones, n0 = m - fN - ones;
    const float mean = m > 0.f ? (n0 * z0 + n1 * z1 + wave_sum(szd)) / m : 0.f;
    const float d0 = z0 - mean, d1 = z1 - mean;
    float sdd = 0.f;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const float dd = lane + 64 * k < nv ? zd[k] - mean : 0.f;
        sdd += dd * dd;
    }
    const float m2 = n0 * d0 * d0 + n1 * d1 * d1 + wave_sum(sdd);
    ClassVals<NWD> v;
    v.rec = nrm_record(mean, m2, m, fN, gnw, eps);
    const float u0 = (z0 - mean) * v.rec.y + beta, u1 = (z1 - mean) * v.rec.y + beta;
    v.u0 = u0;
    v.p = u1 - u0;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const int vx = lane + 64 * k;
        const float e = (zd[k] - mean) * v.rec.y + beta;
        v.q[k] = vx < nv ? e - u0 - v.p * (float)B.wii[vx] : 0.f;
    }
    return v;
}

// ---- K1: GraphNorm records of mlp1 / mlp2 + mult = Y1 Y2 in closed form ---------------------------------------------------
// grid (G, SB_CG), 256 threads: a workgroup writes SB_CPG channels of one graph
template <int NWD>
__global__ __launch_bounds__(256) void sb_fwd_kernel(const unsigned *bits, const int *nvalid, const int N, const float *tab, const float *gnw1,
                                                     const float *gnb1, const float *gnw2, const float *gnb2, const float eps, float *nrm1,
                                                     float *nrm2, float *mult, const long long gstride, const long long ldp, float *xdeg) {
    constexpr int MAXN = 64 * NWD;
    __shared__ GraphBits<NWD> B;
    __shared__ float sc[SB_CPG][8];                    // u0, p, v0, r per channel
    __shared__ float Q[SB_CPG][MAXN], S[SB_CPG][MAXN];
    const int g = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = nvalid_of(nvalid, g, N);
    sb_bits_rows(B, bits, g, tid, N, nv);
    __syncthreads();
    sb_bits_cols(B, tid, nv);
    __syncthreads();
    if (xdeg && cg == 0 && tid < N) xdeg[(long long)g * N + tid] = tid < MAXN ? B.degr[tid] : 0.f;      // what fgnn_adjacency_degree would write
    const int NC = sb_classes(N);
    {   // wave wv owns channel cg * SB_CPG + wv of both models
        const int c = cg * SB_CPG + wv;
        const ClassVals<NWD> a = sb_class_values(tab, B, nv, c, gnw1[c], gnb1[c], eps, lane);
        const ClassVals<NWD> b = sb_class_values(tab + (long long)NC * SB_TAB, B, nv, c, gnw2[c], gnb2[c], eps, lane);
        if (lane == 0) {
            reinterpret_cast<float4 *>(nrm1)[(long long)g * FGNN_H + c] = a.rec;
            reinterpret_cast<float4 *>(nrm2)[(long long)g * FGNN_H + c] = b.rec;
            sc[wv][0] = a.u0;
            sc[wv][1] = a.p;
            sc[wv][2] = b.u0;
            sc[wv][3] = b.p;
        }
#pragma unroll
        for (int k = 0; k < NWD; ++k) {
            Q[wv][lane + 64 * k] = a.q[k];
            S[wv][lane + 64 * k] = b.q[k];
        }
    }
    __syncthreads();
    const int P = N * N;
    const float fN = (float)nv;
    float k0[SB_CPG], k1[SB_CPG], k2[SB_CPG], k3[SB_CPG], u0[SB_CPG], pp[SB_CPG], v0[SB_CPG], rr[SB_CPG];
#pragma unroll
    for (int k = 0; k < SB_CPG; ++k) {
        u0[k] = sc[k][0];
        pp[k] = sc[k][1];
        v0[k] = sc[k][2];
        rr[k] = sc[k][3];
        k0[k] = u0[k] * v0[k] * fN;
        k1[k] = u0[k] * rr[k];
        k2[k] = pp[k] * v0[k];
        k3[k] = pp[k] * rr[k];
    }
    float *out = mult + (long long)g * gstride + (long long)cg * SB_CPG * ldp;
    for (int p = tid; p < P; p += 256) {
        const int i = p / N, j = p - i * N;
        if (i < nv && j < nv) {
            const float w = sb_w(B, i, j) ? 1.f : 0.f;
            const float w2 = (float)sb_w2(B, i, j);
            const float dc = B.degc[j], dr = B.degr[i];
            const bool dg = i == j;
#pragma unroll
            for (int k = 0; k < SB_CPG; ++k) {
                const float q = Q[k][i], s = S[k][j];
                float v = k0[k] + k1[k] * dc + u0[k] * s + k2[k] * dr + k3[k] * w2 + v0[k] * q;
                v += w * (pp[k] * s + rr[k] * q);
                if (dg) v += q * s;
                out[(long long)k * ldp + p] = v;
            }
        } else {                      // padding of a ragged graph: exact zeros, as the generic product leaves them
#pragma unroll
            for (int k = 0; k < SB_CPG; ++k) out[(long long)k * ldp + p] = 0.f;
        }
    }
}

// ---- K2: class sums of dY1 = dM Y2^T and dY2 = Y1^T dM from one pass over dM = d(mult) ------------------------------------
// One 256-thread workgroup per (g, c) plane.  Wave w takes rows w, w + 4, ...: lanes = columns (lane + 64 k), the row sums are
// wave reductions, the column sums stay per lane and are combined over the four waves in a fixed order.  csum[m][g][c][0] =
// off-diagonal w = 0, [1] = off-diagonal w = 1, [2 + i] = (i, i).
template <int NWD>
__global__ __launch_bounds__(256) void sb_bwd_reduce_kernel(const unsigned *bits, const int *nvalid, const int G, const int N, const float *tab,
                                                            const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                                                            const float *dm, const long long gstride, const long long ldp, float *csum) {
    constexpr int MAXN = 64 * NWD, CS = 2 + MAXN;
    __shared__ GraphBits<NWD> B;
    __shared__ float colp[4][3][MAXN];                  // per wave: C, Qm, U per column
    __shared__ float Rs[MAXN], Ps[MAXN], Dg[MAXN];      // per row: sum, W-masked sum, diagonal entry
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int pl = blockIdx.x, g = pl / FGNN_H, c = pl - g * FGNN_H;
    const int nv = nvalid_of(nvalid, g, N);
    // the records and class rows this plane needs are requested before the bit rows are processed
    const float4 ra = reinterpret_cast<const float4 *>(nrm1)[(long long)g * FGNN_H + c];
    const float4 rb = reinterpret_cast<const float4 *>(nrm2)[(long long)g * FGNN_H + c];
    const float ba = gnb1[c], bb = gnb2[c];
    sb_bits_rows(B, bits, g, tid, N, nv);
    __syncthreads();
    sb_bits_cols(B, tid, nv);
    __syncthreads();
    const float *src = dm + (long long)g * gstride + (long long)c * ldp;
    float C[NWD], Qm[NWD], U[NWD];
#pragma unroll
    for (int k = 0; k < NWD; ++k) C[k] = Qm[k] = U[k] = 0.f;
    for (int i0 = wv; i0 < nv; i0 += 16) {               // four rows of this wave per memory round trip
        float v[4][NWD];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NWD; ++k) {
                const int i = i0 + 4 * u, j = lane + 64 * k;
                v[u][k] = (i < nv && j < nv) ? src[i * N + j] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u;
            if (i < nv) {
                float rs = 0.f, rm = 0.f;
#pragma unroll
                for (int k = 0; k < NWD; ++k) {
                    const int j = lane + 64 * k;
                    const float x = v[u][k];
                    const bool wij = (B.row[i][k] >> lane) & 1ull;
                    C[k] += x;
                    rs += x;
                    if (wij) {
                        Qm[k] += x;
                        rm += x;
                    }
                    U[k] += x * (float)sb_w2(B, i, j < MAXN ? j : 0);
                    if (j == i) Dg[i] = x;
                }
                rs = wave_sum(rs);
                rm = wave_sum(rm);
                if (lane == 0) {
                    Rs[i] = rs;
                    Ps[i] = rm;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        colp[wv][0][lane + 64 * k] = C[k];
        colp[wv][1][lane + 64 * k] = Qm[k];
        colp[wv][2][lane + 64 * k] = U[k];
    }
    __syncthreads();
    if (wv != 0) return;
    // wave 0: lane l owns the vertices l + 64 k
    const int NC = sb_classes(N);
    const float *ta = tab, *tb = tab + (long long)NC * SB_TAB;
    const float u0 = (ta[2 * FGNN_H + c] - ra.x) * ra.y + ba, u1 = (ta[SB_TAB + 2 * FGNN_H + c] - ra.x) * ra.y + ba;
    const float v0 = (tb[2 * FGNN_H + c] - rb.x) * rb.y + bb, v1 = (tb[SB_TAB + 2 * FGNN_H + c] - rb.x) * rb.y + bb;
    const float p = u1 - u0, r = v1 - v0;
    const float fN = (float)nv;
    float sC = 0.f, sU = 0.f, sCdc = 0.f, sCs = 0.f, sRdr = 0.f, sQs = 0.f, sqR = 0.f, sqP = 0.f, sd1 = 0.f, swd1 = 0.f, sd2 = 0.f, swd2 = 0.f;
    float d1[NWD], d2[NWD];
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const int vx = lane + 64 * k;
        const bool on = vx < nv;
        const int cl = B.cls[on ? vx : 0];
        const float wii = on ? (float)B.wii[vx] : 0.f;
        const float q = on ? ((ta[cl * SB_TAB + 2 * FGNN_H + c] - ra.x) * ra.y + ba) - u0 - p * wii : 0.f;
        const float s = on ? ((tb[cl * SB_TAB + 2 * FGNN_H + c] - rb.x) * rb.y + bb) - v0 - r * wii : 0.f;
        const float Cj = (colp[0][0][vx] + colp[1][0][vx]) + (colp[2][0][vx] + colp[3][0][vx]);
        const float Qj = (colp[0][1][vx] + colp[1][1][vx]) + (colp[2][1][vx] + colp[3][1][vx]);
        const float Uj = (colp[0][2][vx] + colp[1][2][vx]) + (colp[2][2][vx] + colp[3][2][vx]);
        const float R = on ? Rs[vx] : 0.f, Pm = on ? Ps[vx] : 0.f, dg = on ? Dg[vx] : 0.f;
        const float degr = on ? B.degr[vx] : 0.f, degc = on ? B.degc[vx] : 0.f;
        sC += Cj;
        sU += Uj;
        sCdc += Cj * degc;
        sCs += Cj * s;
        sRdr += R * degr;
        sQs += Qj * s;
        sqR += q * R;
        sqP += q * Pm;
        d1[k] = on ? v0 * R + r * Pm + dg * s : 0.f;             // dA_ii = v0 R_i + r (dM W^T)_ii + dM_ii s_i
        d2[k] = on ? u0 * Cj + p * Qj + q * dg : 0.f;            // dB_ii = u0 C_i + p (W^T dM)_ii + q_i dM_ii
        sd1 += d1[k];
        swd1 += wii * d1[k];
        sd2 += d2[k];
        swd2 += wii * d2[k];
    }
    const float Tt = wave_sum(sC), Ut = wave_sum(sU);
    sCdc = wave_sum(sCdc);
    sRdr = wave_sum(sRdr);
    // mlp1: dA = dM Y2^T,  dA_ij = v0 R_i + r (dM W^T)_ij + dM_ij s_j
    const float tot1 = v0 * fN * Tt + r * sCdc + wave_sum(sCs);
    const float sw1 = v0 * sRdr + r * Ut + wave_sum(sQs);
    sd1 = wave_sum(sd1);
    swd1 = wave_sum(swd1);
    // mlp2: dB = Y1^T dM,  dB_ij = u0 C_j + p (W^T dM)_ij + q_i dM_ij
    const float tot2 = u0 * fN * Tt + p * sRdr + wave_sum(sqR);
    const float sw2 = u0 * sCdc + p * Ut + wave_sum(sqP);
    sd2 = wave_sum(sd2);
    swd2 = wave_sum(swd2);
    float *o1 = csum + ((long long)g * FGNN_H + c) * CS;
    float *o2 = csum + (((long long)G + g) * FGNN_H + c) * CS;
    if (lane == 0) {
        const float off11 = sw1 - swd1, off12 = sw2 - swd2;
        o1[0] = tot1 - sd1 - off11;
        o1[1] = off11;
        o2[0] = tot2 - sd2 - off12;
        o2[1] = off12;
    }
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        o1[2 + lane + 64 * k] = d1[k];
        o2[2 + lane + 64 * k] = d2[k];
    }
}

#ifdef SB_STAMPS
__device__ unsigned long long *g_sb_stamps = nullptr;
#define SB_STAMP(i) if (threadIdx.x == 0 && g_sb_stamps) g_sb_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define SB_STAMP(i)
#endif
// ---- K3: per graph and model, class sums -> GraphNorm backward -> conv / ReLU chain per class -> parameter gradients --------
// grid (G, 2), 256 threads; writes row g of wpart[m] ([W0 (32x2) | b0 | W1 | b1 | W2 | b2], the layout of fgnn_mlp_bwd) and
// s12[m][g][c] = {S1, S2} (the GraphNorm-backward sums fgnn_grad_finalize turns into the affine gradients)
struct ParArgs {
    const float *W[2][3];
    float *wpart[2];
    float *s12[2];
    const float *nrm[2];
};
template <int NWD>
__global__ __launch_bounds__(256) void sb_bwd_params_kernel(const unsigned *bits, const int *nvalid, const int G, const int N, const float *tab,
                                                            const float *csum, const ParArgs A) {
    constexpr int MAXN = 64 * NWD, CS = 2 + MAXN;
    constexpr int KMAX = MAXN + 2;                      // class instances of a graph: the two off-diagonal classes + one per vertex
    extern __shared__ __attribute__((aligned(16))) float sb_lds[];
    __shared__ GraphBits<NWD> B;
    __shared__ float coef[FGNN_H][4];                   // ca, cb, cc per channel
    __shared__ float cnt[KMAX];                         // pixels per instance
    float *Wt = sb_lds;                                 // [2][32 * 32]: W1, W2 of this model
    float (*CSs)[CS + 1] = reinterpret_cast<float (*)[CS + 1]>(Wt + 2 * FGNN_H * FGNN_H);           // [32][CS + 1]: class sums [channel][instance]
    float (*ZV)[FGNN_H] = reinterpret_cast<float (*)[FGNN_H]>(&CSs[FGNN_H][0]);                      // [KMAX][32]: z - mean of the instance's class
    float (*HB)[2 * FGNN_H] = reinterpret_cast<float (*)[2 * FGNN_H]>(&ZV[KMAX][0]);                 // [KMAX][64]: h1 | h2
    float (*DZ)[FGNN_H + 4] = reinterpret_cast<float (*)[FGNN_H + 4]>(&HB[KMAX][0]);                 // [KMAX][36] each: rows 16-byte aligned
    float (*D2)[FGNN_H + 4] = DZ + KMAX;
    float (*D1)[FGNN_H + 4] = D2 + KMAX;
    const int g = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, g, N);
    const int NC = sb_classes(N), K = nv + 2;
    const float *tm = tab + (long long)m * NC * SB_TAB;
    SB_STAMP(0)
    // Everything this workgroup reads from memory is requested up front in explicitly unrolled batches (a rolled staging loop
    // pays one memory round trip per iteration: DESIGN.md section 7, "Rolled staging loops")
    sb_bits_rows(B, bits, g, tid, N, nv);
    {
        float w[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = A.W[m][1][tid + 256 * q];
            w[4 + q] = A.W[m][2][tid + 256 * q];
        }
        constexpr int CSN = FGNN_H * CS, CSI = (CSN + 255) / 256;       // the (model, graph) block of csum is contiguous
        const float *cs = csum + ((long long)m * G + g) * CSN;
        float v[CSI];
#pragma unroll
        for (int q = 0; q < CSI; ++q) v[q] = cs[min(tid + 256 * q, CSN - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Wt[tid + 256 * q] = w[q];
            Wt[FGNN_H * FGNN_H + tid + 256 * q] = w[4 + q];
        }
#pragma unroll
        for (int q = 0; q < CSI; ++q) {
            const int e = tid + 256 * q;
            if (e < CSN) CSs[e / CS][e % CS] = v[q];
        }
    }
    SB_STAMP(1)
    __syncthreads();
    SB_STAMP(2)
    if (tid < MAXN) {
        sb_bits_vertex(B, tid);
        cnt[2 + tid] = 1.f;
    }
    __syncthreads();
    if (tid < 64) {
        float on1 = 0.f;
#pragma unroll
        for (int k = 0; k < NWD; ++k) on1 += tid + 64 * k < nv ? B.degr[tid + 64 * k] - (float)B.wii[tid + 64 * k] : 0.f;
        const float ones = wave_sum(on1);      // off-diagonal ones (threads 0..63 = wave 0)
        const float fN = (float)nv;
        if (tid == 0) {
            cnt[0] = fN * fN - fN - ones;
            cnt[1] = ones;
        }
    }
    SB_STAMP(3)
    {
        // h1 | h2 | z of every instance's class: (instance, channel) pairs, KMAX * 32 / 256 per thread and array, loads first
        constexpr int IT = (KMAX * FGNN_H + 255) / 256;
        float v[3][IT];
        const int c = tid & 31;
        const float mean = reinterpret_cast<const float4 *>(A.nrm[m])[(long long)g * FGNN_H + c].x;
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int k = (tid >> 5) + 8 * q;
            const int cl = k < 2 ? k : B.cls[k - 2 < nv ? k - 2 : 0];
            const float *row = tm + (long long)cl * SB_TAB + c;
            v[0][q] = row[0];
            v[1][q] = row[FGNN_H];
            v[2][q] = row[2 * FGNN_H];
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int k = (tid >> 5) + 8 * q;
            if (k < K) {
                HB[k][c] = v[0][q];
                HB[k][FGNN_H + c] = v[1][q];
                ZV[k][c] = v[2][q] - mean;
            }
        }
    }
    SB_STAMP(4)
    __syncthreads();
    SB_STAMP(5)
    {
        // S1 = sum_k S_k, S2 = sum_k S_k (z_k - mean): 8 threads per channel, fixed partition and tree
        const int c = tid >> 3, s8 = tid & 7;
        float s1 = 0.f, s2 = 0.f;
        for (int k = s8; k < K; k += 8) {
            s1 += CSs[c][k];
            s2 += CSs[c][k] * ZV[k][c];
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            s1 += __shfl_xor(s1, d);
            s2 += __shfl_xor(s2, d);
        }
        if (s8 == 0) {
            const float4 rec = reinterpret_cast<const float4 *>(A.nrm[m])[(long long)g * FGNN_H + c];
            const float fN = (float)nv, mm = fN * fN;
            reinterpret_cast<float2 *>(A.s12[m])[(long long)g * FGNN_H + c] = make_float2(s1, s2);
            // dz_p = ca dy_p + cb (z_p - mean) + cc  
            coef[c][0] = rec.y;
            coef[c][1] = mm > 0.f ? -rec.y * s2 * rec.w / mm : 0.f;
            coef[c][2] = mm > 0.f ? -rec.y * s1 / mm : 0.f;
        }
    }
    __syncthreads();
    SB_STAMP(6)
    // dz summed over the pixels of an instance: ca S_k + n_k (cb (z_k - mean) + cc)
    for (int e = tid; e < K * FGNN_H; e += 256) {
        const int k = e >> 5, c = e & 31;
        DZ[k][c] = coef[c][0] * CSs[c][k] + cnt[k] * (coef[c][1] * ZV[k][c] + coef[c][2]);
    }
    __syncthreads();
    SB_STAMP(7)
    // dpre2 = (W2^T dz) masked by h2, then dpre1 = (W1^T dpre2) masked by h1, every instance at once: thread (k mod 8, c) keeps
    // column c of the transposed weight in registers and reads an instance's vector with 128-bit broadcast loads
    {
        const int c = tid & 31, k0 = tid >> 5;
        float wc[FGNN_H];
#pragma unroll
        for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[FGNN_H * FGNN_H + oo * FGNN_H + c];
        for (int k = k0; k < K; k += 8) {
            const float4 *dz = reinterpret_cast<const float4 *>(DZ[k]);
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = dz[q];
                a = fmaf(wc[4 * q + 0], v.x, a);
                a = fmaf(wc[4 * q + 1], v.y, a);
                a = fmaf(wc[4 * q + 2], v.z, a);
                a = fmaf(wc[4 * q + 3], v.w, a);
            }
            D2[k][c] = HB[k][FGNN_H + c] > 0.f ? a : 0.f;
        }
        SB_STAMP(8)
        __syncthreads();
#pragma unroll
        for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[oo * FGNN_H + c];
        for (int k = k0; k < K; k += 8) {
            const float4 *d2 = reinterpret_cast<const float4 *>(D2[k]);
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = d2[q];
                a = fmaf(wc[4 * q + 0], v.x, a);
                a = fmaf(wc[4 * q + 1], v.y, a);
                a = fmaf(wc[4 * q + 2], v.z, a);
                a = fmaf(wc[4 * q + 3], v.w, a);
            }
            D1[k][c] = HB[k][c] > 0.f ? a : 0.f;
        }
    }
    __syncthreads();
    SB_STAMP(9)
    // Gradients = sums over the instances.  Wave 0: dW2 = sum_k dz_k (x) h2_k and wave 1: dW1 = sum_k dpre2_k (x) h1_k as 32 x 32 x K
    // products on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains in instance order; lane (j, h) supplies row k = 2 s + h of both
    // operands, an odd K is padded with a zero row); waves 2, 3: the 2-column dW0 and two of the bias gradients, eight instances
    // per LDS round trip; the third bias gradient after wave 0's product.
    constexpr int PC = 32 * 2 + 32 + 2 * (32 * 32 + 32);
    float *row = A.wpart[m] + (long long)g * PC;
    const int wv = tid >> 6, lane = tid & 63, jj = lane & 31, hh = lane >> 5;
    if (wv < 2) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float (*L)[FGNN_H + 4] = wv == 0 ? DZ : D2;
        const int hoff = wv == 0 ? FGNN_H : 0;
        constexpr int SB8 = 8;                           // k-steps requested per LDS round trip
        for (int s0 = 0; 2 * s0 < K; s0 += SB8) {
            float av[SB8], bv[SB8];
#pragma unroll
            for (int u = 0; u < SB8; ++u) {
                const int k = 2 * (s0 + u) + hh;
                const bool ok = k < K;
                av[u] = ok ? L[ok ? k : 0][jj] : 0.f;
                bv[u] = ok ? HB[ok ? k : 0][hoff + jj] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SB8; ++u)
                if (2 * (s0 + u) < K) acc = mfma32(av[u], bv[u], acc);
        }
        float *dst = row + (wv == 0 ? 96 + 1024 + 32 : 96);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[ch_of(r, hh) * FGNN_H + jj] = acc[r];
    } else if (tid < 128 + 64) {                         // dW0[t >> 1][t & 1] = sum_k dpre1_k x_k,  t = tid - 128
        const int t = tid - 128;
        float a = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {
            float d[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                const bool ok = k < K;
                d[u] = ok ? D1[ok ? k : 0][t >> 1] : 0.f;
                x[u] = (!ok || k == 0) ? 0.f : (k == 1 ? (t & 1 ? 0.f : 1.f) : (t & 1 ? B.degr[k - 2] : (float)B.wii[k - 2]));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fmaf(d[u], x[u], a);
        }
        row[t] = a;
    } else {                                             // threads 192 .. 255: b0 = sum_k dpre1_k, b1 = sum_k dpre2_k
        const int t = tid - 192, l = t >> 5, cc2 = t & 31;
        const float (*V)[FGNN_H + 4] = l == 0 ? D1 : D2;
        float a = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = k0 + u < K ? V[k0 + u < K ? k0 + u : 0][cc2] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += d[u];
        }
        row[l == 0 ? 64 + cc2 : 96 + 1024 + cc2] = a;
    }
    SB_STAMP(10)
    if (wv == 0 && lane < 32) {                           // b2 = sum_k dz_k (wave 0, after its product)
        float a = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = k0 + u < K ? DZ[k0 + u < K ? k0 + u : 0][lane] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += d[u];
        }
        row[96 + 1024 + 32 + 1024 + lane] = a;
    }
    SB_STAMP(11)
}

template <int NWD>
constexpr int sb_params_lds_bytes() {
    constexpr int MAXN = 64 * NWD, CS = 2 + MAXN, KMAX = MAXN + 2;
    return 4 * (2 * FGNN_H * FGNN_H + FGNN_H * (CS + 1) + KMAX * FGNN_H + KMAX * 2 * FGNN_H + 3 * KMAX * (FGNN_H + 4));
}

}  // namespace

#ifdef SB_STAMPS
extern "C" int fgnn_debug_sb_stamps(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sb_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#endif

extern "C" int fgnn_block1_struct_supported(int N, int depth, int c0) { return (N >= 1 && N <= SB_NMAX && depth == 3 && c0 == 2) ? 1 : 0; }
extern "C" int fgnn_block1_struct_table_floats(int N) { return 2 * (2 + 2 * (N + 1)) * SB_TAB; }
extern "C" int fgnn_block1_struct_csum_floats(int G, int N) { return 2 * G * FGNN_H * (2 + (N <= 64 ? 64 : 128)); }

extern "C" int fgnn_block1_struct_tables(const float *const *W1, const float *const *b1, const float *const *W2, const float *const *b2, int N,
                                         float *tables, void *stream) {
    FGNN_CHECK(W1 && b1 && W2 && b2 && tables && N >= 1 && N <= SB_NMAX, "fgnn_block1_struct_tables: bad arguments (N=%d)", N);
    TabArgs A;
    for (int l = 0; l < 3; ++l) {
        A.W[0][l] = W1[l];
        A.b[0][l] = b1[l];
        A.W[1][l] = W2[l];
        A.b[1][l] = b2[l];
        FGNN_CHECK(W1[l] && b1[l] && W2[l] && b2[l], "fgnn_block1_struct_tables: layer %d missing", l);
    }
    hipLaunchKernelGGL(sb_tables_kernel, dim3(2 + 2 * (N + 1), 2), dim3(64), 0, (hipStream_t)stream, A, N, tables);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_block1_struct_fwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *gnw1,
                                      const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult,
                                      long long gstride, long long ldp, float *xdeg, void *stream) {
    FGNN_CHECK(bits && tables && gnw1 && gnb1 && gnw2 && gnb2 && nrm1 && nrm2 && mult && G > 0, "fgnn_block1_struct_fwd: bad arguments");
    FGNN_CHECK(N >= 1 && N <= SB_NMAX, "fgnn_block1_struct_fwd: N = %d (built for N <= %d)", N, SB_NMAX);
    FGNN_CHECK(ldp >= (long long)N * N && gstride >= FGNN_H * ldp, "fgnn_block1_struct_fwd: strides smaller than the planes");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64)
        hipLaunchKernelGGL(sb_fwd_kernel<1>, dim3(G, SB_CG), dim3(256), 0, st, bits, nvalid, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult,
                           gstride, ldp, xdeg);
    else
        hipLaunchKernelGGL(sb_fwd_kernel<2>, dim3(G, SB_CG), dim3(256), 0, st, bits, nvalid, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult,
                           gstride, ldp, xdeg);
    FGNN_LAUNCH_CHECK();
    return 0;
}

namespace {
template <int NWD>
int sb_bwd_launch(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const ParArgs &A, const float *gnb1, const float *gnb2,
                  const float *dmult, long long gstride, long long ldp, float *csum, hipStream_t st) {
    hipLaunchKernelGGL(sb_bwd_reduce_kernel<NWD>, dim3(G * FGNN_H), dim3(256), 0, st, bits, nvalid, G, N, tables, A.nrm[0], A.nrm[1], gnb1, gnb2, dmult,
                       gstride, ldp, csum);
    FGNN_LAUNCH_CHECK();
    constexpr int LDS = sb_params_lds_bytes<NWD>();
    static_assert(LDS <= 150 * 1024, "LDS budget of the class-algebra kernel");
    static LdsAttrCache attr_cache;
    FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)sb_bwd_params_kernel<NWD>, LDS), "fgnn_block1_struct_bwd: %d bytes of LDS refused", LDS);
    hipLaunchKernelGGL(sb_bwd_params_kernel<NWD>, dim3(G, 2), dim3(256), LDS, st, bits, nvalid, G, N, tables, csum, A);
    FGNN_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int fgnn_block1_struct_bwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *const *W1,
                                      const float *const *W2, const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                                      const float *dmult, long long gstride, long long ldp, float *csum, float *wpart1, float *wpart2,
                                      float *s12_1, float *s12_2, void *stream) {
    FGNN_CHECK(bits && tables && W1 && W2 && nrm1 && nrm2 && gnb1 && gnb2 && dmult && csum && wpart1 && wpart2 && s12_1 && s12_2 && G > 0,
               "fgnn_block1_struct_bwd: bad arguments");
    FGNN_CHECK(N >= 1 && N <= SB_NMAX, "fgnn_block1_struct_bwd: N = %d (built for N <= %d)", N, SB_NMAX);
    FGNN_CHECK(G <= fgnn_mlp_bwd_num_workgroups(), "fgnn_block1_struct_bwd: one partial row per graph: G = %d exceeds the %d rows of wpart", G,
               fgnn_mlp_bwd_num_workgroups());
    ParArgs A;
    for (int l = 0; l < 3; ++l) {
        A.W[0][l] = W1[l];
        A.W[1][l] = W2[l];
    }
    A.wpart[0] = wpart1;
    A.wpart[1] = wpart2;
    A.s12[0] = s12_1;
    A.s12[1] = s12_2;
    A.nrm[0] = nrm1;
    A.nrm[1] = nrm2;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return sb_bwd_launch<1>(bits, nvalid, G, N, tables, A, gnb1, gnb2, dmult, gstride, ldp, csum, st);
    return sb_bwd_launch<2>(bits, nvalid, G, N, tables, A, gnb1, gnb2, dmult, gstride, ldp, csum, st);
}
