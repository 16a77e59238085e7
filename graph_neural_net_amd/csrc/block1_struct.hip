// Block 1 of the 2-FGNN on its STRUCTURED input (bit-packed adjacency, constant-size batches, N <= 64).
//
// The reference feeds block 1 the tensor representation of a graph (loaders/data_generator.py:118-125): channel 0 = the 0/1
// adjacency W, channel 1 = diag(row sums).  So the input pixel (i, j) takes one of few values -- off the diagonal (w_ij, 0)
// with w in {0, 1}, on it (w_ii, deg_i) -- and so do the outputs of mlp1 / mlp2 (models/blocks_emb.py:16-27: two
// MlpBlock_Real on that input): a table of 2 + 2 (N + 1) rows per model instead of G N^2 pixels through three convs.  With
// the normalised class values u0 / u1 / e_i (mlp1) and v0 / v1 / f_j (mlp2) of a graph,
//     Y1_c = u0 J + p W + diag(q),   p = u1 - u0, q_i = e_i - u0 - p w_ii        (and Y2_c = v0 J + r W + diag(s) likewise)
// and the per-channel product mult_c = Y1_c Y2_c (models/layers.py:161-162) has the closed form
//     mult_c[i][j] = u0 v0 N + u0 r degc_j + u0 s_j + p v0 degr_i + p r (W^2)_ij + p w_ij s_j + v0 q_i + r q_i w_ij + [i = j] q_i s_i
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
#include "fgnn_common.h"
#include "fgnn_norm.h"

namespace {

constexpr int SB_MAXN = 64;
constexpr int SB_CG = 8;                 // channel groups of the forward kernel (4 channels each)
constexpr int SB_CPG = FGNN_H / SB_CG;   // channels per group
constexpr int SB_TAB = 3 * FGNN_H;       // floats per table row: h1 | h2 | z

// class of a pixel's input value: 0 = off-diagonal, w = 0; 1 = off-diagonal, w = 1; 2 + 2 deg + w_ii = diagonal
DEVI int sb_classes(int N) { return 2 + 2 * (N + 1); }

typedef unsigned long long u64;

// the graph's bit rows as 64-bit words (bits >= N cleared) and what follows from them, for lane / thread t < N
DEVI u64 sb_row(const unsigned *bits, int g, int t, int N) {
    const int words = (N + 31) / 32;
    const unsigned *r = bits + ((long long)g * N + t) * words;
    u64 v = r[0];
    if (words > 1) v |= (u64)r[1] << 32;
    return N >= 64 ? v : (v & ((1ull << N) - 1ull));
}

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

// per-graph bit structure in LDS, shared by the kernels below (filled by the first N threads of a group of >= 64 threads)
struct GraphBits {
    u64 row[SB_MAXN], col[SB_MAXN];
    float degr[SB_MAXN], degc[SB_MAXN];
    int wii[SB_MAXN], cls[SB_MAXN];
};
// step 1 (then a barrier / wave sync), step 2
DEVI void sb_bits_rows(GraphBits &B, const unsigned *bits, int g, int t, int N) {
    if (t < SB_MAXN) B.row[t] = t < N ? sb_row(bits, g, t, N) : 0ull;
}
DEVI void sb_bits_vertex(GraphBits &B, int t) {           // what needs the vertex's own row only
    const int dr = __popcll(B.row[t]);
    B.degr[t] = (float)dr;
    const int w = (int)((B.row[t] >> t) & 1ull);
    B.wii[t] = w;
    B.cls[t] = 2 + 2 * dr + w;
}
DEVI void sb_bits_cols(GraphBits &B, int t, int N) {
    if (t < SB_MAXN) {
        u64 c = 0ull;
        for (int i0 = 0; i0 < N; i0 += 8) {               // eight rows per LDS round trip (rows >= N are 0)
            u64 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = B.row[(i0 + k) & (SB_MAXN - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k) c |= ((r[k] >> t) & 1ull) << ((i0 + k) & (SB_MAXN - 1));
        }
        B.col[t] = c;
        B.degc[t] = (float)__popcll(c);
        sb_bits_vertex(B, t);
    }
}

// The normalised class values of channel c of one model in one graph, for lane i (< N, else zeros): statistics from the
// class counts (two-pass: mean, then squared deviations), record as fgnn_norm.h, y = (z - mean) a + beta as every consumer of
// a slab evaluates it.  All 64 lanes of a wave call this together.
struct ClassVals {
    float u0, p, q;          // Y = u0 J + p W + diag(q): q is the lane's q_i
    float4 rec;              // GraphNorm record of (g, c)
    float z0, z1, zd;        // raw class values (zd: the lane's diagonal class)
};
DEVI ClassVals sb_class_values(const float *tab_m, const GraphBits &B, int N, int c, float gnw, float beta, float eps, int lane) {
    const float z0 = tab_m[0 * SB_TAB + 2 * FGNN_H + c], z1 = tab_m[1 * SB_TAB + 2 * FGNN_H + c];
    const bool on = lane < N;
    const float zd = on ? tab_m[B.cls[on ? lane : 0] * SB_TAB + 2 * FGNN_H + c] : 0.f;
    const float ones = wave_sum(on ? B.degr[lane] - (float)B.wii[lane] : 0.f);      // off-diagonal ones
    const float fN = (float)N, m = fN * fN;
    const float n1 = ones, n0 = m - fN - ones;
    const float mean = (n0 * z0 + n1 * z1 + wave_sum(zd)) / m;
    const float d0 = z0 - mean, d1 = z1 - mean, dd = on ? zd - mean : 0.f;
    const float m2 = n0 * d0 * d0 + n1 * d1 * d1 + wave_sum(dd * dd);
    ClassVals v;
    v.rec = nrm_record(mean, m2, m, fN, gnw, eps);
    v.z0 = z0;
    v.z1 = z1;
    v.zd = zd;
    const float u0 = (z0 - mean) * v.rec.y + beta, u1 = (z1 - mean) * v.rec.y + beta;
    const float e = (zd - mean) * v.rec.y + beta;
    v.u0 = u0;
    v.p = u1 - u0;
    v.q = on ? e - u0 - v.p * (float)B.wii[lane] : 0.f;
    return v;
}

// ---- K1: GraphNorm records of mlp1 / mlp2 + mult = Y1 Y2 in closed form ---------------------------------------------------
// grid (G, SB_CG), 256 threads: a workgroup writes SB_CPG channels of one graph
__global__ __launch_bounds__(256) void sb_fwd_kernel(const unsigned *bits, const int N, const float *tab, const float *gnw1, const float *gnb1,
                                                     const float *gnw2, const float *gnb2, const float eps, float *nrm1, float *nrm2,
                                                     float *mult, const long long gstride, const long long ldp, float *xdeg) {
    __shared__ GraphBits B;
    __shared__ float sc[SB_CPG][8];                    // u0, p, v0, r per channel
    __shared__ float Q[SB_CPG][SB_MAXN], S[SB_CPG][SB_MAXN];
    const int g = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    sb_bits_rows(B, bits, g, tid, N);
    __syncthreads();
    sb_bits_cols(B, tid, N);
    __syncthreads();
    if (xdeg && cg == 0 && tid < N) xdeg[(long long)g * N + tid] = B.degr[tid];      // what fgnn_adjacency_degree would write
    const int NC = sb_classes(N);
    {   // wave wv owns channel cg * SB_CPG + wv of both models
        const int c = cg * SB_CPG + wv;
        const ClassVals a = sb_class_values(tab, B, N, c, gnw1[c], gnb1[c], eps, lane);
        const ClassVals b = sb_class_values(tab + (long long)NC * SB_TAB, B, N, c, gnw2[c], gnb2[c], eps, lane);
        if (lane == 0) {
            reinterpret_cast<float4 *>(nrm1)[(long long)g * FGNN_H + c] = a.rec;
            reinterpret_cast<float4 *>(nrm2)[(long long)g * FGNN_H + c] = b.rec;
            sc[wv][0] = a.u0;
            sc[wv][1] = a.p;
            sc[wv][2] = b.u0;
            sc[wv][3] = b.p;
        }
        Q[wv][lane] = a.q;
        S[wv][lane] = b.q;
    }
    __syncthreads();
    const int P = N * N;
    const float fN = (float)N;
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
        const u64 ri = B.row[i];
        const float w = (float)((ri >> j) & 1ull);
        const float w2 = (float)__popcll(ri & B.col[j]);
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
    }
}

// ---- K2: class sums of dY1 = dM Y2^T and dY2 = Y1^T dM from one pass over dM = d(mult) ------------------------------------
// one wave per (g, c) plane, 4 waves per workgroup; csum[m][g][c][0] = off-diagonal w = 0, [1] = off-diagonal w = 1, [2 + i] = (i, i)
constexpr int SB_CS = 2 + SB_MAXN;
__global__ __launch_bounds__(256) void sb_bwd_reduce_kernel(const unsigned *bits, const int G, const int N, const float *tab, const float *nrm1,
                                                            const float *nrm2, const float *gnb1, const float *gnb2, const float *dm,
                                                            const long long gstride, const long long ldp, float *csum) {
    __shared__ GraphBits B4[4];
    __shared__ float plane[4][SB_MAXN * (SB_MAXN + 1)];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pl = blockIdx.x * 4 + wv;                 // (g, c); G * 32 is a multiple of 4
    const int g = pl / FGNN_H, c = pl - g * FGNN_H;
    GraphBits &B = B4[wv];
    float *T = plane[wv];
    // ONE memory round trip: the plane (lane = column), the two GraphNorm records and the bit rows are requested together
    const float *src = dm + (long long)g * gstride + (long long)c * ldp;
    float v[SB_MAXN];
#pragma unroll
    for (int i = 0; i < SB_MAXN; ++i) v[i] = (lane < N && i < N) ? src[i * N + lane] : 0.f;
    const float4 ra = reinterpret_cast<const float4 *>(nrm1)[(long long)g * FGNN_H + c];
    const float4 rb = reinterpret_cast<const float4 *>(nrm2)[(long long)g * FGNN_H + c];
    const float ba = gnb1[c], bb = gnb2[c];
    // (wave-private LDS: in-order within the wave, no barrier needed; __syncthreads keeps the compiler honest about visibility)
    sb_bits_rows(B, bits, g, lane, N);
    __syncthreads();
    sb_bits_cols(B, lane, N);
    __syncthreads();
    const int NC = sb_classes(N);
    const bool on = lane < N;
    // the class values of this channel in both models, from the forward's records
    const float *ta = tab, *tb = tab + (long long)NC * SB_TAB;
    const int cl = B.cls[on ? lane : 0];
    const float u0 = (ta[2 * FGNN_H + c] - ra.x) * ra.y + ba, u1 = (ta[SB_TAB + 2 * FGNN_H + c] - ra.x) * ra.y + ba;
    const float v0 = (tb[2 * FGNN_H + c] - rb.x) * rb.y + bb, v1 = (tb[SB_TAB + 2 * FGNN_H + c] - rb.x) * rb.y + bb;
    const float p = u1 - u0, r = v1 - v0;
    const float wii = on ? (float)B.wii[lane] : 0.f;
    const float q = on ? ((ta[cl * SB_TAB + 2 * FGNN_H + c] - ra.x) * ra.y + ba) - u0 - p * wii : 0.f;
    const float s = on ? ((tb[cl * SB_TAB + 2 * FGNN_H + c] - rb.x) * rb.y + bb) - v0 - r * wii : 0.f;
    // pass 1, lane = column j: column sums C, masked column sums Qm, <dM, W^2> share, diagonal; the plane goes to LDS
    const u64 cj = B.col[lane];
    float C = 0.f, Qm = 0.f, U = 0.f, dg = 0.f;
    {
#pragma unroll
        for (int i = 0; i < SB_MAXN; ++i) {
            if (i < N) {
                const u64 ri = B.row[i];
                C += v[i];
                if ((ri >> lane) & 1ull) Qm += v[i];
                U += v[i] * (float)__popcll(ri & cj);
                if (i == lane) dg = v[i];
                T[i * (SB_MAXN + 1) + lane] = v[i];
            }
        }
    }
    __syncthreads();
    // pass 2, lane = row i: row sums R, masked row sums Pm
    float R = 0.f, Pm = 0.f;
    const u64 rl = B.row[lane];
    if (on) {
        for (int j0 = 0; j0 < N; j0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = T[lane * (SB_MAXN + 1) + ((j0 + k) & (SB_MAXN - 1))];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (j0 + k < N) {
                    R += v[k];
                    if ((rl >> (j0 + k)) & 1ull) Pm += v[k];
                }
            }
        }
    }
    const float degr = on ? B.degr[lane] : 0.f, degc = on ? B.degc[lane] : 0.f;
    const float fN = (float)N;
    const float Tt = wave_sum(C), Ut = wave_sum(U);
    // mlp1: dA = dM Y2^T,  dA_ij = v0 R_i + r (dM W^T)_ij + dM_ij s_j
    const float tot1 = v0 * fN * Tt + r * wave_sum(C * degc) + wave_sum(C * s);
    const float sw1 = v0 * wave_sum(R * degr) + r * Ut + wave_sum(Qm * s);
    const float d1 = on ? v0 * R + r * Pm + dg * s : 0.f;
    const float sd1 = wave_sum(d1), swd1 = wave_sum(wii * d1);
    // mlp2: dB = Y1^T dM,  dB_ij = u0 C_j + p (W^T dM)_ij + q_i dM_ij
    const float tot2 = u0 * fN * Tt + p * wave_sum(R * degr) + wave_sum(q * R);
    const float sw2 = u0 * wave_sum(C * degc) + p * Ut + wave_sum(q * Pm);
    const float d2 = on ? u0 * C + p * Qm + q * dg : 0.f;
    const float sd2 = wave_sum(d2), swd2 = wave_sum(wii * d2);
    float *o1 = csum + ((long long)g * FGNN_H + c) * SB_CS;
    float *o2 = csum + (((long long)G + g) * FGNN_H + c) * SB_CS;
    if (lane == 0) {
        const float off11 = sw1 - swd1, off12 = sw2 - swd2;
        o1[0] = tot1 - sd1 - off11;
        o1[1] = off11;
        o2[0] = tot2 - sd2 - off12;
        o2[1] = off12;
    }
    o1[2 + lane] = d1;
    o2[2 + lane] = d2;
}

#ifdef SB_STAMPS
__device__ unsigned long long *g_sb_stamps = nullptr;
#define SB_STAMP(i) if (threadIdx.x == 0 && g_sb_stamps) g_sb_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define SB_STAMP(i)
#endif
#ifdef SB_STOP
#define SB_STOP_AT(k) if (SB_STOP == k) return;
#else
#define SB_STOP_AT(k)
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
__global__ __launch_bounds__(256) void sb_bwd_params_kernel(const unsigned *bits, const int G, const int N, const float *tab, const float *csum,
                                                            const ParArgs A) {
    constexpr int KMAX = SB_MAXN + 2;                   // class instances of a graph: the two off-diagonal classes + one per vertex
    __shared__ GraphBits B;
    __shared__ float Wt[2][FGNN_H * FGNN_H];            // W1, W2 of this model
    __shared__ float CS[FGNN_H][SB_CS + 1];             // class sums of this (model, graph): [channel][instance]
    __shared__ float ZV[KMAX][FGNN_H];                  // z - mean of the instance's class
    __shared__ __attribute__((aligned(16))) float HB[KMAX][2 * FGNN_H];   // h1 | h2 of the instance's class
    __shared__ __attribute__((aligned(16))) float DZ[KMAX][FGNN_H + 4], D2[KMAX][FGNN_H + 4], D1[KMAX][FGNN_H + 4];   // rows 16-byte aligned
    __shared__ float coef[FGNN_H][4];                   // ca, cb, cc per channel
    __shared__ float cnt[KMAX + 2];                     // pixels per instance
    const int g = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    const int NC = sb_classes(N), K = N + 2;
    const float *tm = tab + (long long)m * NC * SB_TAB;
    SB_STAMP(0)
    // Everything this workgroup reads from memory is requested up front in explicitly unrolled batches (a rolled staging loop
    // pays one memory round trip per iteration: DESIGN.md section 7, "Rolled staging loops")
    sb_bits_rows(B, bits, g, tid, N);
    {
        float w[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = A.W[m][1][tid + 256 * q];
            w[4 + q] = A.W[m][2][tid + 256 * q];
        }
        constexpr int CSN = FGNN_H * SB_CS, CSI = (CSN + 255) / 256;       // the (model, graph) block of csum is contiguous
        const float *cs = csum + ((long long)m * G + g) * CSN;
        float v[CSI];
#pragma unroll
        for (int q = 0; q < CSI; ++q) v[q] = cs[min(tid + 256 * q, CSN - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Wt[0][tid + 256 * q] = w[q];
            Wt[1][tid + 256 * q] = w[4 + q];
        }
#pragma unroll
        for (int q = 0; q < CSI; ++q) {
            const int e = tid + 256 * q;
            if (e < CSN) CS[e / SB_CS][e % SB_CS] = v[q];
        }
    }
    SB_STAMP(1)
    __syncthreads();
    SB_STAMP(2)
    SB_STOP_AT(1)
    if (tid < SB_MAXN) {
        sb_bits_vertex(B, tid);
        const float ones = wave_sum(tid < N ? B.degr[tid] - (float)B.wii[tid] : 0.f);      // off-diagonal ones (threads 0..63 = wave 0)
        const float fN = (float)N;
        if (tid == 0) {
            cnt[0] = fN * fN - fN - ones;
            cnt[1] = ones;
        }
        cnt[2 + tid] = 1.f;
    }
    __syncthreads();
    SB_STAMP(3)
    SB_STOP_AT(2)
    {
        // h1 | h2 | z of every instance's class: (instance, channel) pairs, KMAX * 32 / 256 per thread and array, loads first
        constexpr int IT = (KMAX * FGNN_H + 255) / 256;
        float v[3][IT];
        const int c = tid & 31;
        const float mean = reinterpret_cast<const float4 *>(A.nrm[m])[(long long)g * FGNN_H + c].x;
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int k = (tid >> 5) + 8 * q;
            const int cl = k < 2 ? k : B.cls[k - 2 < N ? k - 2 : 0];
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
            s1 += CS[c][k];
            s2 += CS[c][k] * ZV[k][c];
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            s1 += __shfl_xor(s1, d);
            s2 += __shfl_xor(s2, d);
        }
        if (s8 == 0) {
            const float4 rec = reinterpret_cast<const float4 *>(A.nrm[m])[(long long)g * FGNN_H + c];
            const float fN = (float)N, mm = fN * fN;
            reinterpret_cast<float2 *>(A.s12[m])[(long long)g * FGNN_H + c] = make_float2(s1, s2);
            // dz_p = ca dy_p + cb (z_p - mean) + cc  
            coef[c][0] = rec.y;
            coef[c][1] = -rec.y * s2 * rec.w / mm;
            coef[c][2] = -rec.y * s1 / mm;
        }
    }
    __syncthreads();
    SB_STAMP(6)
    SB_STOP_AT(3)
    // dz summed over the pixels of an instance: ca S_k + n_k (cb (z_k - mean) + cc)
    for (int e = tid; e < K * FGNN_H; e += 256) {
        const int k = e >> 5, c = e & 31;
        DZ[k][c] = coef[c][0] * CS[c][k] + cnt[k] * (coef[c][1] * ZV[k][c] + coef[c][2]);
    }
    __syncthreads();
    SB_STAMP(7)
    // dpre2 = (W2^T dz) masked by h2, then dpre1 = (W1^T dpre2) masked by h1, every instance at once: thread (k mod 8, c) keeps
    // column c of the transposed weight in registers and reads an instance's vector with 128-bit broadcast loads
    {
        const int c = tid & 31, k0 = tid >> 5;
        float wc[FGNN_H];
#pragma unroll
        for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[1][oo * FGNN_H + c];
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
        for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[0][oo * FGNN_H + c];
#if defined(SB_STOP) && SB_STOP == 6
        if (false)
#endif
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
#if defined(SB_STOP) && SB_STOP >= 5
    if (SB_STOP == 5 || SB_STOP == 6) {     // debug: keep the class vectors alive, skip the gradient sums
        float a = 0.f;
        for (int k = tid >> 5; k < K; k += 8) a += D1[k][tid & 31] + D2[k][tid & 31] + DZ[k][tid & 31];
        A.wpart[m][(long long)g * 2208 + tid] = a;
        return;
    }
#endif
    SB_STAMP(9)
    SB_STOP_AT(4)
    // Gradients = sums over the instances.  Wave 0: dW2 = sum_k dz_k (x) h2_k and wave 1: dW1 = sum_k dpre2_k (x) h1_k as 32 x 32 x K
    // products on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains in instance order; lane (j, h) supplies row k = 2 s + h of both
    // operands, an odd K is padded with a zero row); waves 2, 3: the 2-column dW0 and the three bias gradients, eight instances
    // per LDS round trip.
    constexpr int PC = 32 * 2 + 32 + 2 * (32 * 32 + 32);
    float *row = A.wpart[m] + (long long)g * PC;
    const int wv = tid >> 6, lane = tid & 63, jj = lane & 31, hh = lane >> 5;
    if (wv < 2) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float (*L)[FGNN_H + 4] = wv == 0 ? DZ : D2;
        const int hoff = wv == 0 ? FGNN_H : 0;
        constexpr int SMAX = KMAX / 2;                   // k-steps of two instances; every operand is requested before the first product
        float av[SMAX], bv[SMAX];
#pragma unroll
        for (int u = 0; u < SMAX; ++u) {
            const int k = 2 * u + hh;
            const bool ok = k < K;
            av[u] = ok ? L[ok ? k : 0][jj] : 0.f;
            bv[u] = ok ? HB[ok ? k : 0][hoff + jj] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SMAX; ++u)
            if (2 * u < K) acc = mfma32(av[u], bv[u], acc);
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

}  // namespace

#ifdef SB_STAMPS
extern "C" int fgnn_debug_sb_stamps(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sb_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#endif

extern "C" int fgnn_block1_struct_supported(int N, int depth, int c0) { return (N >= 1 && N <= SB_MAXN && depth == 3 && c0 == 2) ? 1 : 0; }
extern "C" int fgnn_block1_struct_table_floats(int N) { return 2 * (2 + 2 * (N + 1)) * SB_TAB; }
extern "C" int fgnn_block1_struct_csum_floats(int G) { return 2 * G * FGNN_H * SB_CS; }

extern "C" int fgnn_block1_struct_tables(const float *const *W1, const float *const *b1, const float *const *W2, const float *const *b2, int N,
                                         float *tables, void *stream) {
    FGNN_CHECK(W1 && b1 && W2 && b2 && tables && N >= 1 && N <= SB_MAXN, "fgnn_block1_struct_tables: bad arguments (N=%d)", N);
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

extern "C" int fgnn_block1_struct_fwd(const unsigned *bits, int G, int N, const float *tables, const float *gnw1, const float *gnb1,
                                      const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult, long long gstride,
                                      long long ldp, float *xdeg, void *stream) {
    FGNN_CHECK(bits && tables && gnw1 && gnb1 && gnw2 && gnb2 && nrm1 && nrm2 && mult && G > 0, "fgnn_block1_struct_fwd: bad arguments");
    FGNN_CHECK(N >= 1 && N <= SB_MAXN, "fgnn_block1_struct_fwd: N = %d (built for N <= %d)", N, SB_MAXN);
    FGNN_CHECK(ldp >= (long long)N * N && gstride >= FGNN_H * ldp, "fgnn_block1_struct_fwd: strides smaller than the planes");
    hipLaunchKernelGGL(sb_fwd_kernel, dim3(G, SB_CG), dim3(256), 0, (hipStream_t)stream, bits, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2,
                       mult, gstride, ldp, xdeg);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_block1_struct_bwd(const unsigned *bits, int G, int N, const float *tables, const float *const *W1, const float *const *W2,
                                      const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2, const float *dmult,
                                      long long gstride, long long ldp, float *csum, float *wpart1, float *wpart2, float *s12_1, float *s12_2,
                                      void *stream) {
    FGNN_CHECK(bits && tables && W1 && W2 && nrm1 && nrm2 && gnb1 && gnb2 && dmult && csum && wpart1 && wpart2 && s12_1 && s12_2 && G > 0,
               "fgnn_block1_struct_bwd: bad arguments");
    FGNN_CHECK(N >= 1 && N <= SB_MAXN, "fgnn_block1_struct_bwd: N = %d (built for N <= %d)", N, SB_MAXN);
    FGNN_CHECK(G <= fgnn_mlp_bwd_num_workgroups(), "fgnn_block1_struct_bwd: one partial row per graph: G = %d exceeds the %d rows of wpart", G,
               fgnn_mlp_bwd_num_workgroups());
    FGNN_CHECK((G * FGNN_H) % 4 == 0, "fgnn_block1_struct_bwd: internal: planes per workgroup");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sb_bwd_reduce_kernel, dim3(G * FGNN_H / 4), dim3(256), 0, st, bits, G, N, tables, nrm1, nrm2, gnb1, gnb2, dmult, gstride, ldp,
                       csum);
    FGNN_LAUNCH_CHECK();
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
    hipLaunchKernelGGL(sb_bwd_params_kernel, dim3(G, 2), dim3(256), 0, st, bits, G, N, tables, csum, A);
    FGNN_LAUNCH_CHECK();
    return 0;
}
