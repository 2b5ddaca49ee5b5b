// Register-level butterflies of the NTT (device only): one wavefront transforms 128 rows with no LDS traffic and no
// barrier between its seven radix-2 stages (north_star: "wavefront-shuffle butterflies").
//
// A lane owns one butterfly, i.e. TWO elements (x0, x1) -- with one element per lane the twiddle product of a stage would
// run on half the lanes.  Between stages a lane trades ONE of its two elements with lane ^ m (ds_bpermute_b32: the LDS
// crossbar, no LDS memory), after which its pair is again (position p, position p + d) of the next stage:
//
//   DIF (natural in, bit-reversed out)   in : x0 = row L, x1 = row L + 64             (L = lane)
//        d = 64 needs no trade; then m = 32, 16, ..., 1: trade with lane ^ m, butterfly at distance m, twiddle w_(2m)^(L & (m-1))
//                                         out: x0 = position 2L, x1 = position 2L + 1
//   DIT (bit-reversed in, natural out)   in : x0 = position 2L, x1 = position 2L + 1
//        d = 1 needs no trade and no product; then m = 1, 2, ..., 32: trade with lane ^ m, butterfly at distance d = 2m,
//                                         twiddle w_(2d)^(L & (d-1));   out: x0 = row L, x1 = row L + 64
//   trade rule (both directions): a lane with bit m set sends x0 and receives into x0, the others send x1 and receive into x1.
//
// Position bookkeeping: DIF starts with lane bits = position bits p5..p0 and the local index = p6; every trade swaps the local
// index with one lane bit, so after the trade with m = 2^i the local index is p_i and lane bit i holds p_(i+1).  The twiddle
// index p mod d then always equals L & (d - 1).  Same results as ntt_tile_stage (ntt_tile.cuh), stage for stage.
#pragma once
#include "ntt_tile.cuh"

MI_D Fr wave_trade(const Fr &v, u32 src_lane) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v.l[i]);
    return r;
}
MI_D void wave_select_trade(Fr &x0, Fr &x1, u32 lane, u32 m) {
    const bool hi = (lane & m) != 0;
    Fr send;
#pragma unroll
    for (int i = 0; i < 8; i++) send.l[i] = hi ? x0.l[i] : x1.l[i];
    const Fr recv = wave_trade(send, lane ^ m);
#pragma unroll
    for (int i = 0; i < 8; i++) { x0.l[i] = hi ? recv.l[i] : x0.l[i]; x1.l[i] = hi ? x1.l[i] : recv.l[i]; }
}
// small[j] = w_4096^j (forward or inverse root): w_(2d)^j = small[j << (11 - log2 d)]
MI_D void wave_ntt128(Fr &x0, Fr &x1, u32 lane, bool dit, const Fr *small) {
    if (!dit) {
#pragma unroll 1
        for (int log_d = 6; log_d >= 0; log_d--) {
            const u32 d = 1u << log_d;
            if (log_d < 6) wave_select_trade(x0, x1, lane, d);
            ntt_bfly_dif(x0, x1, log_d ? &small[(lane & (d - 1)) << (11 - log_d)] : nullptr);   // d = 1: every twiddle is 1
        }
    } else {
#pragma unroll 1
        for (int log_d = 0; log_d <= 6; log_d++) {
            const u32 d = 1u << log_d;
            if (log_d) wave_select_trade(x0, x1, lane, d >> 1);
            ntt_bfly_dit(x0, x1, log_d ? &small[(lane & (d - 1)) << (11 - log_d)] : nullptr);
        }
    }
}
