// GEMM pieces for 16-pixel-tile kernels that run ONE wave per SIMD (csrc/mlp64.hip, csrc/mlp_bwd_pair_a.hip): nothing hides an LDS
// read's latency there but the wave's own MFMAs, so the operands of set i + 1 are requested BEFORE the 16 MFMAs of set i are issued
// (LDS returns in order; the compiler's s_waitcnt lgkmcnt(n) then waits for set i only).  The callers pin the order with
// __builtin_amdgcn_sched_barrier(0): left alone the scheduler sinks the reads back to their first use.
#pragma once
#include "fgnn_t16.h"

namespace t16 {

struct Set4 {                   // one 32 x 32 operand set in the MFMA operand order of fgnn_t16.h gemm32 (4 x ds_read_b128)
    float4 w[4];
};
DEVI Set4 load_set(const float *img, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(img) + lane;
    Set4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o.w[u] = p[u * 64];
    return o;
}
DEVI void mfma_set(f32x4 (&acc)[2], const Set4 &W, const float (&bop)[8]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        acc[0] = mfma16(W.w[u].x, bop[2 * u], acc[0]);
        acc[1] = mfma16(W.w[u].y, bop[2 * u], acc[1]);
        acc[0] = mfma16(W.w[u].z, bop[2 * u + 1], acc[0]);
        acc[1] = mfma16(W.w[u].w, bop[2 * u + 1], acc[1]);
    }
}

// rows i and 16 + i of a staged tile, pixels 4 q .. 4 q + 3 (lane (i, q)): the A / B operand of a weight-gradient set
struct Pair4 {
    float4 lo, hi;
};
DEVI Pair4 load_rows(const float *T, int lane) {
    const int i = lane & 15, q = lane >> 4;
    Pair4 o;
    o.lo = *reinterpret_cast<const float4 *>(T + i * TLD + 4 * q);
    o.hi = *reinterpret_cast<const float4 *>(T + (16 + i) * TLD + 4 * q);
    return o;
}
// dW[2 mb + nb] += Dt (rows 16 mb ..) x In (rows 16 nb ..) over the 16 pixels (fgnn_t16.h wgrad16)
DEVI void wgrad_mfma(f32x4 (&dW)[4], const Pair4 &a, const Pair4 &b) {
#define FGNN_T16P_KS(e)                              \
    dW[0] = mfma16(a.lo.e, b.lo.e, dW[0]);           \
    dW[1] = mfma16(a.lo.e, b.hi.e, dW[1]);           \
    dW[2] = mfma16(a.hi.e, b.lo.e, dW[2]);           \
    dW[3] = mfma16(a.hi.e, b.hi.e, dW[3]);
    FGNN_T16P_KS(x)
    FGNN_T16P_KS(y)
    FGNN_T16P_KS(z)
    FGNN_T16P_KS(w)
#undef FGNN_T16P_KS
}
// pixel sums of the rows of a gradient tile (the bias gradient), from the operand the weight-gradient set has loaded anyway
DEVI void bias_add(float (&db)[2], const Pair4 &a) {
    db[0] += (a.lo.x + a.lo.y) + (a.lo.z + a.lo.w);
    db[1] += (a.hi.x + a.hi.y) + (a.hi.z + a.hi.w);
}

}  // namespace t16
