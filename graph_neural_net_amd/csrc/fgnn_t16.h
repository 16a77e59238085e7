// 16-pixel tiles on v_mfma_f32_16x16x4_f32 (round 6): the tile geometry, fragment layout and GEMM helpers shared by the
// *_t16 MLP kernels.  Same arithmetic as the 32-pixel kernels (mlp_fwd.hip / mlp_bwd.hip: models/layers.py:126-131 and its
// autograd): out[o][p] = sum_c W[o][c] in[c][p] with 16 output channels as MFMA rows, 16 pixels as MFMA columns, 4 channels
// contracted per MFMA in the order k = 0..3 (an fmaf chain, tools/ubench_mfma16.hip).
//
// Fragment layout.  Lane l = (px = l & 15, q = l >> 4).  A 32-channel slab is 8 registers per lane, register s holding channel
//     chan(s, q) = 8 (s >> 1) + 2 (s & 1) + (q >> 1) + 4 (q & 1)
// so that k-step s (one MFMA per 16-row block) contracts channels chan(s, 0..3) = seq[4 s .. 4 s + 3] of the 32-pixel kernels'
// summation order seq = 0,4,1,5,2,6,3,7, 8,12,9,13, ...: a conv chain started from the bias runs through the SAME sequence of
// fused multiply-adds as v_mfma_f32_32x32x2_f32 does there, i.e. the recomputed hidden activations (and with them every ReLU
// decision) are bit-identical to the 32-pixel forward's.  The D fragment of a layer -- block b, register r of lane (px, q) =
// output row 4 q + r of the block -- is made to BE that layout by permuting the rows of the weight image (row m of block b holds
// output channel chan(4 b + (m & 3), m >> 2)), so the chain never leaves the register file here either.
// Per lane a tile costs half the registers of a 32-pixel tile; per-graph records stay in registers across the tiles of a graph.
#pragma once
#include "fgnn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace t16 {

constexpr int TPX = 16;                 // pixels per tile
constexpr int TLD = 20;                 // LDS tile row stride (floats): 16 lanes x ds_read_b128 cover all 64 banks
constexpr int TILE_F = 32 * TLD;        // floats per 32-channel LDS tile

DEVI constexpr int chan(int s, int q) { return 8 * (s >> 1) + 2 * (s & 1) + (q >> 1) + 4 * (q & 1); }
DEVI constexpr int chan_s(int s) { return 8 * (s >> 1) + 2 * (s & 1); }        // register part
DEVI int chan_q(int q) { return (q >> 1) + 4 * (q & 1); }                       // lane part

#ifndef FGNN_ABL
#define FGNN_ABL 0          // measurement switch (tools/build_variant.sh ... -DFGNN_ABL=mask): ablated builds, never shipped.  1 no partner waits, 2 operands
                            // from registers instead of the LDS image, 4 no global traffic in the tile loop, 8 no staging writes, 16 no weight gradients, 32 no MFMAs
#endif
DEVI f32x4 mfma16(float a, float b, f32x4 c) {
#if FGNN_ABL & 32           // no matrix instructions: everything but the MFMA pipe
    c[0] += a * b;
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

DEVI f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// ---- one 32 x 32 operand set from the LDS image: acc[b] += Mx[rows of block b][.] x bop ---------------------------------
// Image element (t, lane), t = 2 s + b, stored [t / 4][lane][4] (one ds_read_b128 = k-steps s, s + 1 of both row blocks).
// The two row blocks are independent accumulator chains, so consecutive MFMAs never wait for each other's result.
template <int OFF>
DEVI void gemm32(f32x4 (&acc)[2], const float *wl, const float (&bop)[8], int lane) {
    static_assert(OFF % 4 == 0, "operand sets are float4 aligned");
    const float4 *p = reinterpret_cast<const float4 *>(wl) + (OFF / 4) * 64 + lane;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#if FGNN_ABL & 2           // operands from registers instead of the LDS image
        const float4 w = make_float4(bop[0], bop[1], bop[2], bop[3 + (u & 1)]);
        (void)p;
#else
        const float4 w = p[u * 64];
#endif
        acc[0] = mfma16(w.x, bop[2 * u], acc[0]);
        acc[1] = mfma16(w.y, bop[2 * u], acc[1]);
        acc[0] = mfma16(w.z, bop[2 * u + 1], acc[0]);
        acc[1] = mfma16(w.w, bop[2 * u + 1], acc[1]);
    }
}
// a 2-channel slab: ONE k-step (lanes q = 0, 1 carry channels 0, 1; the image holds zeros for q = 2, 3)
template <int OFF>
DEVI void gemm2(f32x4 (&acc)[2], const float *wl, float bop, int lane) {
    static_assert(OFF % 4 == 0, "operand sets are float4 aligned");
    const float4 w = (reinterpret_cast<const float4 *>(wl) + (OFF / 4) * 64)[lane];
    acc[0] = mfma16(w.x, bop, acc[0]);
    acc[1] = mfma16(w.y, bop, acc[1]);
}

// bias[chan(s, q)], s = 0..7, of one layer from the compact tail [layer][q][8] (broadcast reads)
DEVI void load_bias(f32x4 (&acc)[2], const float *tail, int layer, int q) {
    const float4 *p = reinterpret_cast<const float4 *>(tail + layer * 32 + q * 8);
    const float4 v0 = p[0], v1 = p[1];
    acc[0][0] = v0.x; acc[0][1] = v0.y; acc[0][2] = v0.z; acc[0][3] = v0.w;
    acc[1][0] = v1.x; acc[1][1] = v1.y; acc[1][2] = v1.z; acc[1][3] = v1.w;
}

// ---- LDS tiles [row][pixel] --------------------------------------------------------------------------------------------------
// Register s of lane (px, q) goes to ROW 16 (s >> 2) + (s & 3) + 4 q, i.e. row m of 16-row block b holds channel row_chan(16 b + m) =
// chan(4 b + (m & 3), m >> 2) -- the row permutation of the weight images.  The four rows one ds_write touches (q = 0..3) are 4 apart:
// 80 floats = 16 banks, so the 64 lanes cover all 64 banks (rows in channel order: q = 1 and q = 2 overlapped on 12 banks; PMC:
// SQ_LDS_BANK_CONFLICT 1.17 M per launch of the pair backward).  The weight-gradient GEMMs read whole 16-row blocks, so their result
// fragments are simply indexed by row_chan(); nothing else reads a tile by channel.
// lane_base = tile_lane_base(px, q) (floats); the register part is a compile-time offset
#ifndef FGNN_ROWMAP
#define FGNN_ROWMAP 1       // measurement switch: 0 = rows in channel order (the first version; 2-way conflicts on 12 banks per write)
#endif
#if FGNN_ROWMAP
DEVI int tile_lane_base(int px, int q) { return 4 * q * TLD + px; }
DEVI constexpr int tile_row_s(int s) { return 16 * (s >> 2) + (s & 3); }
DEVI int row_chan(int row) { return chan(4 * (row >> 4) + (row & 3), (row & 15) >> 2); }
#else
DEVI int tile_lane_base(int px, int q) { return chan_q(q) * TLD + px; }
DEVI constexpr int tile_row_s(int s) { return chan_s(s); }
DEVI int row_chan(int row) { return row; }
#endif
DEVI void stage8(float *T, int lane_base, const float (&v)[8]) {
#if FGNN_ABL & 8           // no staging writes
    if (v[0] != 1.2345e-30f) return;
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) T[lane_base + tile_row_s(s) * TLD] = v[s];
}

// dW[2 mb + nb] += Dt (rows 16 mb ..) x In (rows 16 nb ..), contraction over the 16 pixels: register r of lane (n, q) is
// dW[row_chan(16 mb + 4 q + r)][row_chan(16 nb + n)]; db[mb] from the same reads (lane (i, q) holds pixels 4 q .. 4 q + 3 of row 16 mb + i)
DEVI void wgrad16(f32x4 (&dW)[4], float (&db)[2], const float *Dt, const float *In, int lane) {
#if FGNN_ABL & 16           // no weight gradients
    return;
#endif
    const int i = lane & 15, q = lane >> 4;
    const float4 a0 = *reinterpret_cast<const float4 *>(Dt + i * TLD + 4 * q);
    const float4 a1 = *reinterpret_cast<const float4 *>(Dt + (16 + i) * TLD + 4 * q);
    const float4 b0 = *reinterpret_cast<const float4 *>(In + i * TLD + 4 * q);
    const float4 b1 = *reinterpret_cast<const float4 *>(In + (16 + i) * TLD + 4 * q);
    db[0] += (a0.x + a0.y) + (a0.z + a0.w);
    db[1] += (a1.x + a1.y) + (a1.z + a1.w);
#define FGNN_T16_KS(e)                          \
    dW[0] = mfma16(a0.e, b0.e, dW[0]);          \
    dW[1] = mfma16(a0.e, b1.e, dW[1]);          \
    dW[2] = mfma16(a1.e, b0.e, dW[2]);          \
    dW[3] = mfma16(a1.e, b1.e, dW[3]);
    FGNN_T16_KS(x)
    FGNN_T16_KS(y)
    FGNN_T16_KS(z)
    FGNN_T16_KS(w)
#undef FGNN_T16_KS
}
// ... with a 2-channel input (rows 0, 1 of In; the other rows of the tile must be zero): not used, block 1 keeps the VALU form

// ---- global slabs ------------------------------------------------------------------------------------------------------------
// 32-channel slab, 16 pixels from p0: register s <- channel chan(s, q) of pixel p.  voff = per-lane byte offset (OOB_OFF when the
// pixel does not exist), s0 = graph offset in bytes
DEVI int lane_voff(const View &v, int q, int p, bool inb) { return inb ? chan_q(q) * v.ld4 + 4 * p : OOB_OFF; }
DEVI void load8(float (&x)[8], const View &v, int voff, int s0) {
#if FGNN_ABL & 4           // no global traffic in the tile loop
    for (int s = 0; s < 8; ++s) x[s] = __builtin_bit_cast(float, voff + s0 + s);
    return;
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = buf_load(v, voff, s0 + chan_s(s) * v.ld4);
}
#ifndef FGNN_STORE_AUX
#define FGNN_STORE_AUX 2    // cache policy of the slab stores: 2 = nt (streamed output: measured -0.9 us per launch against 0); 16 = sc1, 17 = sc0 sc1
#endif
DEVI void store8(const float (&x)[8], const View &v, int voff, int s0) {
#if FGNN_ABL & 4
    if (x[0] != 1.2345e-30f) return;
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[s]), v.r, voff, s0 + chan_s(s) * v.ld4, FGNN_STORE_AUX);
}

// row / column of pixel p in an N x N plane without an integer division: (p + 0.5) / N is at least 0.5 / N away from an
// integer, the float product is within 2^-22 (p + 0.5) / N of it; exact for p < 2^16 (N <= 256)
DEVI void row_col(int p, int N, float rcpN, int &i, int &jj) {
    i = (int)(((float)p + 0.5f) * rcpN);
    jj = p - i * N;
}

}  // namespace t16
