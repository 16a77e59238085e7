// GraphNorm record arithmetic shared by the finalize kernels (norm.hip) and by kernels that finalize a
// producer's tile statistics in their own prologue (matmul.hip).
#pragma once
#include "fgnn_common.h"

// nrm record {mean, a = w*q, q = 1/(2 sqrt(n (var+eps))), r2 = 1/(var+eps)} (models/layers.py:71-80)
DEVI float4 nrm_record(float mean, float m2, float m, float nv, float w, float eps) {
    const float var = m > 0.f ? m2 / m : 0.f;
    const float ve = var + eps;
    const float q = nv > 0.f ? 1.f / (2.f * sqrtf(nv * ve)) : 0.f;      // an empty (filler) graph normalises to 0, not inf
    float4 o;
    o.x = mean;
    o.y = w * q;
    o.z = q;
    o.w = 1.f / ve;
    return o;
}

// One wave combines the <= 256 tile partials {mean_t, M2_t} (+ counts n_t) of channel c of graph g:
//   mean = sum_t n_t mean_t / sum_t n_t,   M2 = sum_t [ M2_t + n_t (mean_t - mean)^2 ]
// exact two-level decomposition with plain (fixed-tree) wave sums; every lane returns the record.
// Split in a load half and a reduce half so that a caller can put other loads in flight between them.
struct TilePartials {
    float nb[4], mb[4], qb[4];
};
// `tr`: the partials are stored (G, C, tpg, 2) -- contiguous over the tiles of one channel (the bf16 kernels write them that
// way: every consumer walks one (g, c) column, and with the (G, tpg, C, 2) layout each 8-byte read pulls in a whole line) --
// instead of (G, tpg, C, 2)
DEVI TilePartials finalize_load(const float *part, const float *cnt, int g, int c, int C, int tpg, int lane, bool tr = false) {
    TilePartials p;
    const int ts = tr ? 1 : C, cs = tr ? tpg : 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = lane + WAVE * k;
        const bool ok = t < tpg;
        const int tc = ok ? t : 0;
        const float n = cnt[(long long)g * tpg + tc];
        const float2 pm = reinterpret_cast<const float2 *>(part)[(long long)g * tpg * C + (long long)tc * ts + (long long)c * cs];
        p.nb[k] = ok ? n : 0.f;
        p.mb[k] = pm.x;
        p.qb[k] = ok ? pm.y : 0.f;
    }
    return p;
}
DEVI float4 finalize_reduce(const TilePartials &p, float nv, float w, float eps) {
    float sn = 0.f, sm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sn += p.nb[k];
        sm += p.nb[k] * p.mb[k];
    }
    sn = wave_sum(sn);
    sm = wave_sum(sm);
    const float mean = sn > 0.f ? sm / sn : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = p.mb[k] - mean;
        m2 += p.qb[k] + p.nb[k] * d * d;
    }
    m2 = wave_sum(m2);
    return nrm_record(mean, m2, sn, nv, w, eps);
}
DEVI float4 finalize_wave(const float *part, const float *cnt, int g, int c, int C, int tpg, float nv, float w, float eps,
                          int lane, bool tr = false) {
    return finalize_reduce(finalize_load(part, cnt, g, c, C, tpg, lane, tr), nv, w, eps);
}
// any number of tiles: more than 256 (N > 90) are walked in passes of eight loads per lane (a plain loop pays one memory round
// trip per 64 tiles); the same arithmetic as gn_finalize_kernel
DEVI float4 finalize_wave_any(const float *part, const float *cnt, int g, int c, int C, int tpg, float nv, float w, float eps, int lane,
                              bool tr = false) {
    if (tpg <= 4 * WAVE) return finalize_wave(part, cnt, g, c, C, tpg, nv, w, eps, lane, tr);
    constexpr int U = 8;
    float sn = 0.f, sm = 0.f;
    const float *cg = cnt + (long long)g * tpg;
    const int ts = tr ? 1 : C;              // tile stride of the partials (float2 units)
    const float2 *pg = reinterpret_cast<const float2 *>(part) + (long long)g * tpg * C + (long long)c * (tr ? tpg : 1);
    for (int t0 = lane; t0 < tpg; t0 += U * WAVE) {
        float n[U], x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int t = t0 + k * WAVE, tc = t < tpg ? t : 0;
            n[k] = t < tpg ? cg[tc] : 0.f;
            x[k] = pg[(long long)tc * ts].x;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            sn += n[k];
            sm += n[k] * x[k];
        }
    }
    sn = wave_sum(sn);
    sm = wave_sum(sm);
    const float mean = sn > 0.f ? sm / sn : 0.f;
    float m2 = 0.f;
    for (int t0 = lane; t0 < tpg; t0 += U * WAVE) {
        float n[U];
        float2 pm[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int t = t0 + k * WAVE, tc = t < tpg ? t : 0;
            n[k] = t < tpg ? cg[tc] : 0.f;
            pm[k] = pg[(long long)tc * ts];
            if (t >= tpg) pm[k].y = 0.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float d = pm[k].x - mean;
            m2 += pm[k].y + n[k] * d * d;
        }
    }
    m2 = wave_sum(m2);
    return nrm_record(mean, m2, sn, nv, w, eps);
}
