// The host-only walk over gnark's ProvingKey.WriteRawTo stream (layout notes in csrc/pk_raw.hip): offsets and counts of
// every section, every count cross-checked.  No HIP in this file: it reads bytes an outside party wrote, and the CPU build with
// -fsanitize=address,undefined (make sanitize; tests/test_parsers_sanitized.py) compiles it together with csrc/whir_ingest.hip.
#include "../../include/mi355x_groth16.h"
#include <cstring>

namespace {
struct Cursor {
    const uint8_t *p;
    size_t len, off = 0;
    bool ok = true;
    const uint8_t *take(size_t n) {
        if (!ok || n > len - off) { ok = false; return nullptr; }
        const uint8_t *r = p + off;
        off += n;
        return r;
    }
    uint64_t be(size_t n) {
        const uint8_t *b = take(n);
        uint64_t v = 0;
        if (b) for (size_t i = 0; i < n; i++) v = (v << 8) | b[i];
        return v;
    }
};
}  // namespace

extern "C" {

// Host-only walk over the stream: offsets and counts of every section (no device needed; what the CPU tests check).
int32_t mi_pk_raw_inspect(const uint8_t *buf, size_t len, mi_pk_raw_info *info) {
    if (!buf || !info) return MI_EINVAL;
    std::memset(info, 0, sizeof(*info));
    Cursor c{buf, len};
    const uint64_t card = c.be(8);
    c.take(5 * 32);                       // CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen, FrMultiplicativeGenInv
    const uint64_t with_pre = c.be(1);    // withPrecompute
    if (!c.ok || card == 0 || (card & (card - 1)) || card > ((uint64_t)1 << 28) || with_pre > 1) return MI_EINVAL;
    uint32_t log_n = 0;
    while (((uint64_t)1 << log_n) < card) log_n++;
    info->log_n = log_n;
    info->off_alpha1 = c.off; c.take(3 * 64);
    auto g1s = [&](uint64_t *off, uint64_t *cnt) { *cnt = c.be(4); *off = c.off; c.take((size_t)*cnt * 64); };
    g1s(&info->off_g1_a, &info->n_g1_a);
    g1s(&info->off_g1_b, &info->n_g1_b);
    g1s(&info->off_g1_z, &info->n_g1_z);
    g1s(&info->off_g1_k, &info->n_g1_k);
    info->off_beta2 = c.off; c.take(2 * 128);
    info->n_g2_b = c.be(4); info->off_g2_b = c.off; c.take((size_t)info->n_g2_b * 128);
    info->nb_wires = c.be(8);
    const uint64_t n_inf_a = c.be(8), n_inf_b = c.be(8);
    const uint64_t la = c.be(4); info->off_infinity_a = c.off; c.take((size_t)((la + 7) / 8));
    const uint64_t lb = c.be(4); info->off_infinity_b = c.off; c.take((size_t)((lb + 7) / 8));
    info->n_commitment_keys = (uint32_t)c.be(4);
    if (!c.ok || la != info->nb_wires || lb != info->nb_wires || info->n_commitment_keys > MI_PK_RAW_MAX_COMMITMENTS) return MI_EINVAL;
    for (uint32_t k = 0; k < info->n_commitment_keys; k++) {
        uint64_t n1, n2;
        g1s(&info->off_basis[k], &n1);
        g1s(&info->off_basis_exp_sigma[k], &n2);
        if (!c.ok || n1 != n2) return MI_EINVAL;
        info->n_basis[k] = n1;
    }
    if (!c.ok || c.off != len) return MI_EINVAL;   // trailing bytes = not the layout this parser knows
    // cross-checks that tie the sections together
    if (info->n_g1_a > info->nb_wires || info->n_g1_b > info->nb_wires) return MI_EINVAL;   // (the sums below are mod 2^64)
    if (info->n_g1_a + n_inf_a != info->nb_wires || info->n_g1_b + n_inf_b != info->nb_wires || info->n_g2_b != info->n_g1_b) return MI_EINVAL;
    if (info->n_g1_z + 1 < card || info->n_g1_k > info->nb_wires) return MI_EINVAL;
    return MI_OK;
}


}  // extern "C"
