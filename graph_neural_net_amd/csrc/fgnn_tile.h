// Tile geometry and slab loads shared by the x3 MLP kernels (the same conventions as mlp_fwd.hip / mlp_bwd.hip):
// a tile = 32 consecutive pixels of one graph; lane (j, h) = pixel j of the tile, half-wave h; a 32-channel slab gives
// the lane its 16 channels ch_of(r, h), a 2-channel slab gives half-wave h channel h.
#pragma once
#include "fgnn_common.h"

namespace {

constexpr int TLD = 36;              // LDS tile row stride (floats): 144 B rows, 16-B aligned
constexpr int TILE_F = 32 * TLD;     // floats per 32-row tile

struct TileCtx {
    int g, tt, p, i, jj;
    bool inb;
};

// (the valid-vertex count of the graph is fetched once per graph change, not here: see mlp_fwd.hip)
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

// rows ch_of(r,h) of a (G,32,ld) tensor
DEVI void load_rows16(float (&x)[16], const View &v, const TileCtx &c, int h) {
    const int voff = lane_off<4>(v, c, h);
    const int s0 = c.g * v.gs4;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = buf_load(v, voff, s0 + ((r & 3) + 8 * (r >> 2)) * v.ld4);
}

// y = (x - mean) * a + beta with the per-graph records {mean, a, beta, -} read from wave-private LDS; 0 on invalid pixels
// (a 0/1 mask multiply: `valid ? .. : 0` becomes divergent control flow around the LDS reads)
template <int S>
DEVI void norm_slab(float (&y)[S > 0 ? S : 1], const float (&x)[S > 0 ? S : 1], const float *rec, bool on, bool valid, int h) {
    if constexpr (S > 0) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rec);
        if (on) {                       // wave-uniform
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

}  // namespace
