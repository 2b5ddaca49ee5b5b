// Internal: the cross-rank half of a transform sharded over the ranks of a device group (csrc/ntt_cross.hip; used by csrc/group.hip).
#pragma once
#include "ctx.h"
#include "field.cuh"

struct CrossNttTables {
    void *tw_inv = nullptr, *tw_fwd = nullptr;   // M / W elements: w^(-j2), w^(+i2) for this rank's columns
    void *s_fwd = nullptr, *s_inv = nullptr;     // M elements: g^k, den g^-k for the coefficient at every local position
    u32 log_n = 0, log_w = 0, rank = 0;
    Fr w_inv_scale, den;                         // 1 / W, den = 1 / (g^N - 1)
};
int32_t mi_cross_tables_build(mi_ctx *ctx, u32 log_n, u32 log_w, u32 rank, CrossNttTables *t);
void mi_cross_tables_free(CrossNttTables *t);
// in place on Y = W rows (source ranks) of M / W columns: mode 0 = the inverse transform's cross-rank step (rows in, natural; rows out,
// bit-reversed; every output times 1 / W, or den / W), mode 1 = the forward transform's (rows in, bit-reversed; rows out, natural)
int32_t mi_cross_dft(mi_ctx *ctx, hipStream_t st, void *Y, const CrossNttTables &t, int mode, bool den_scale);
// computeH's middle on the columns: Ya <- inverse step(forward step(Ya) o forward step(Yb)) -- the coset transforms' cross-rank steps
// of a and b, their product and the last transform's cross-rank step in one kernel
int32_t mi_cross_mid(mi_ctx *ctx, hipStream_t st, void *Ya, const void *Yb, const CrossNttTables &t);
int32_t mi_cross_mul(mi_ctx *ctx, hipStream_t st, void *z, const void *x, const void *y, const void *w_or_null, size_t n);
