// Mutation driver for the library's host-only parsers of OUTSIDE bytes: mi_whir_proof_decode (+ the accessors and mi_whir_parse_paths on
// whatever decodes), mi_whir_interner_decode, mi_whir_config_parse, mi_whir_matrix_cells, mi_pk_raw_inspect.  Compiled TOGETHER with
// csrc/whir_ingest.hip and csrc/pk_raw_inspect.hip as plain C++ under -fsanitize=address,undefined (gnark-whir_amd/Makefile `sanitize`;
// tests/test_parsers_sanitized.py builds and runs it on the CPU -- sanitizers belong on this build only).  Test infrastructure.
//
//   parser_fuzz <proof.bin> <config.json> <interner.bin> <pk_raw.bin> <iterations> <seed>
//
// Every seed file is a VALID input (the driver first checks that each parses).  Mutations: every truncation (short inputs) or sampled
// ones, 1-4 byte flips, 8-byte little-endian / 4-byte big-endian length fields overwritten with 0, 1, 2^32 - 1, 2^63, 2^64 - 1 and
// values near the input's own length, random blobs, splices of two mutants; for the JSON text also token splices (literals cut short,
// lone surrogates, bad escapes, 9223372036854775808, leading zeros, control characters, 20000-deep nesting).  Outputs are written into
// buffers sized from the shapes the library reports, so an out-of-bounds write on the library's side is the sanitizer's to find.
// Exit status 0 = no sanitizer report and no contract violation (a mutant may be accepted or refused; it may not crash).
#include "../../include/mi355x_groth16.h"
#include "../../include/mi355x_whir_ingest.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef std::vector<uint8_t> Bytes;
static uint64_t rng_state = 1;
static uint64_t rnd() { uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static Bytes read_file(const char *path) {
    Bytes b;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "parser_fuzz: cannot open %s\n", path); exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}

static unsigned long n_accept[5], n_refuse[5];

static void run_proof(const Bytes &in) {
    mi_whir_proof *p = nullptr;
    size_t used = 0;
    const int32_t rc = mi_whir_proof_decode(in.data(), in.size(), &p, &used);
    if (rc != MI_OK) { n_refuse[0]++; if (p) { fprintf(stderr, "decode failed but returned a handle\n"); exit(3); } return; }
    n_accept[0]++;
    if (used > in.size()) { fprintf(stderr, "decode consumed more than it was given\n"); exit(3); }
    const uint64_t ns = mi_whir_proof_statement_values(p, nullptr);
    std::vector<uint64_t> sv(4 * ns + 1);
    mi_whir_proof_statement_values(p, sv.data());
    for (int which = 0; which < 2; which++) {
        const uint64_t ne = mi_whir_proof_elements(p, which);
        for (uint64_t i = 0; i < ne && i < 64; i++) {
            mi_whir_shape sh;
            if (mi_whir_element_shape(p, which, i, &sh) != MI_OK) { fprintf(stderr, "shape of an existing element refused\n"); exit(3); }
            const unsigned __int128 cells = (unsigned __int128)sh.n_leaves * sh.tree_height;
            if (cells > (1u << 21) || sh.total_leaf_values > (1u << 21)) continue;   // the caller's buffers would be > 64 MB: skipped here
            std::vector<uint8_t> paths((size_t)cells * 32 + 1), sib(sh.n_leaves * 32 + 1);
            std::vector<uint64_t> idx(sh.n_leaves + 1), lens(sh.n_leaves + 1), leaves(4 * sh.total_leaf_values + 1);
            (void)mi_whir_parse_paths(p, which, i, paths.data(), sib.data(), idx.data(), lens.data(), leaves.data());
            (void)mi_whir_parse_paths(p, which, i, nullptr, nullptr, nullptr, nullptr, nullptr);
        }
        mi_whir_shape sh;
        if (mi_whir_element_shape(p, which, ne, &sh) == MI_OK) { fprintf(stderr, "shape of element [count] accepted\n"); exit(3); }
    }
    mi_whir_proof_free(p);
}
static void run_interner(const Bytes &in) {
    uint64_t n = 0;
    size_t used = 0;
    if (mi_whir_interner_decode(in.data(), in.size(), nullptr, &n, &used) != MI_OK) { n_refuse[1]++; return; }
    n_accept[1]++;
    if (n > in.size() / 32 || used > in.size()) { fprintf(stderr, "interner count exceeds the input\n"); exit(3); }
    std::vector<uint64_t> limbs(4 * n + 1);
    if (mi_whir_interner_decode(in.data(), in.size(), limbs.data(), &n, &used) != MI_OK) { fprintf(stderr, "interner: second pass refused\n"); exit(3); }
    // CSR arrays drawn from the same bytes: rows, columns, value indexes (any of them may be out of range: refused, not trusted)
    const size_t nnz = below(40), rows = below(12);
    std::vector<uint64_t> ri(rows + 1), ci(nnz + 1), vi(nnz + 1), ro(nnz + 1), co(nnz + 1), vo(4 * nnz + 4);
    for (auto &x : ri) x = below(4) ? below(nnz + 2) : rnd();
    for (auto &x : ci) x = rnd();
    for (auto &x : vi) x = below(5) ? below((size_t)n + 1) : rnd();
    (void)mi_whir_matrix_cells(ri.data(), rows, ci.data(), vi.data(), nnz, limbs.data(), (size_t)n, ro.data(), co.data(), vo.data());
}
static void run_config(const Bytes &in) {
    mi_whir_config *c = nullptr;
    if (mi_whir_config_parse((const char *)in.data(), in.size(), &c) != MI_OK) { n_refuse[2]++; if (c) { fprintf(stderr, "config failed but returned a handle\n"); exit(3); } return; }
    n_accept[2]++;
    // everything the handle points at must be readable
    volatile uint64_t sink = 0;
    for (size_t i = 0; i < c->io_pattern_len; i++) sink += (uint8_t)c->io_pattern[i];
    for (size_t i = 0; i < c->n_transcript; i++) sink += c->transcript[i];
    for (size_t i = 0; i < 4 * c->n_statement_evaluations; i++) sink += c->statement_evaluations[i];
    if (c->n_folding_factor > MI_WHIR_MAX_ROUNDS || c->n_ood_samples > MI_WHIR_MAX_ROUNDS || c->n_num_queries > MI_WHIR_MAX_ROUNDS || c->n_pow_bits > MI_WHIR_MAX_ROUNDS) { fprintf(stderr, "config list longer than its array\n"); exit(3); }
    mi_whir_config_free(c);
}
static void run_pk_raw(const Bytes &in) {
    mi_pk_raw_info info;
    if (mi_pk_raw_inspect(in.data(), in.size(), &info) != MI_OK) { n_refuse[3]++; return; }
    n_accept[3]++;
    // every section the info names must lie inside the input
    auto inside = [&](uint64_t off, unsigned __int128 bytes) { if ((unsigned __int128)off + bytes > in.size()) { fprintf(stderr, "pk raw: a section leaves the input\n"); exit(3); } };
    inside(info.off_alpha1, 192); inside(info.off_g1_a, (unsigned __int128)info.n_g1_a * 64); inside(info.off_g1_b, (unsigned __int128)info.n_g1_b * 64);
    inside(info.off_g1_z, (unsigned __int128)info.n_g1_z * 64); inside(info.off_g1_k, (unsigned __int128)info.n_g1_k * 64); inside(info.off_beta2, 256);
    inside(info.off_g2_b, (unsigned __int128)info.n_g2_b * 128); inside(info.off_infinity_a, (info.nb_wires + 7) / 8); inside(info.off_infinity_b, (info.nb_wires + 7) / 8);
    for (uint32_t k = 0; k < info.n_commitment_keys; k++) { inside(info.off_basis[k], (unsigned __int128)info.n_basis[k] * 64); inside(info.off_basis_exp_sigma[k], (unsigned __int128)info.n_basis[k] * 64); }
}
static void run_helpers() {
    const size_t n = below(9), eb = 1 + below(40), pl = below(12), ns = below(9);
    Bytes a(n * eb + 1), b(n * eb + 1), suf(ns * eb + 1), out((pl + ns) * eb + 1);
    for (auto &x : a) x = (uint8_t)rnd();
    (void)mi_whir_reverse(a.data(), n, eb, b.data());
    size_t cnt = 0;
    (void)mi_whir_prefix_decode_path(a.data(), n, pl, suf.data(), ns, eb, out.data(), &cnt);   // pl > n must be refused before anything is copied
    uint64_t in4[4] = {rnd(), rnd(), rnd(), rnd()}, out4[4];
    mi_whir_limbs_to_fr(in4, out4);
}

static const char *TOKENS[] = {"{", "}", "[", "]", ",", ":", "\"", "\\", "\\u", "\\ud800", "\\udc00\\ud800", "\\ud83c\\udf2a", "\\x", "null", "nul", "nXYZ", "n", "true", "tru", "false",
                               "0", "-", "-0", "01", "1e9", "1.5", "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809", "1e", "\x01", "\n", " ",
                               "\"transcript\"", "\"folding_factor\"", "\"N_VARS\"", "\"statement_evaluations\"", "\"domain_generator\"", "\"io_pattern\"", "AAEC", "A===", "=", "\xff\xfe"};
static Bytes mutate(const Bytes &seed, bool text) {
    Bytes m = seed;
    const int kind = (int)below(text ? 9 : 7);
    switch (kind) {
        case 0: m.resize(below(m.size() + 1)); break;                                             // truncation
        case 1: for (size_t k = 0, f = 1 + below(4); k < f && !m.empty(); k++) m[below(m.size())] ^= (uint8_t)(1u << below(8)); break;   // bit flips
        case 2: for (size_t k = 0, f = 1 + below(4); k < f && !m.empty(); k++) m[below(m.size())] = (uint8_t)rnd(); break;             // byte writes
        case 3: {                                                                                  // an 8-byte little-endian length field
            if (m.size() < 8) break;
            const uint64_t vals[] = {0, 1, 0xffffffffull, 1ull << 63, ~0ull, (uint64_t)m.size(), (uint64_t)m.size() / 8, (uint64_t)m.size() / 32 + 1, rnd()};
            const uint64_t v = vals[below(sizeof vals / sizeof vals[0])];
            const size_t off = below(m.size() / 8) * 8;
            for (int k = 0; k < 8; k++) m[off + k] = (uint8_t)(v >> (8 * k));
            break;
        }
        case 4: {                                                                                  // a 4- or 8-byte big-endian count (the key file's)
            if (m.size() < 8) break;
            const uint64_t vals[] = {0, 1, 0xffffffffull, 0x80000000ull, (uint64_t)m.size(), (uint64_t)m.size() / 64, rnd()};
            const uint64_t v = vals[below(sizeof vals / sizeof vals[0])];
            const int w = below(2) ? 4 : 8;
            const size_t off = below(m.size() - 8);
            for (int k = 0; k < w; k++) m[off + k] = (uint8_t)(v >> (8 * (w - 1 - k)));
            break;
        }
        case 5: m.assign(below(300), 0); for (auto &x : m) x = (uint8_t)rnd(); break;              // a random blob
        case 6: { Bytes o = seed; o.resize(below(o.size() + 1)); const size_t cut = below(m.size() + 1); m.resize(cut); m.insert(m.end(), o.begin(), o.end()); break; }   // a splice
        case 7: {                                                                                  // text: token splices
            for (size_t k = 0, f = 1 + below(3); k < f; k++) {
                const char *t = TOKENS[below(sizeof TOKENS / sizeof TOKENS[0])];
                const size_t at = below(m.size() + 1), del = below(6);
                m.erase(m.begin() + at, m.begin() + (at + del < m.size() ? at + del : m.size()));
                m.insert(m.begin() + at, t, t + strlen(t));
            }
            break;
        }
        default: {                                                                                 // text: nesting inside an unknown key
            const size_t depth = below(3) == 0 ? 20000 : below(12000);
            std::string s = "{\"k\": ";
            s.append(depth, below(2) ? '[' : '{');
            if (below(2)) s.append(depth, ']');
            s += "}";
            m.assign(s.begin(), s.end());
        }
    }
    return m;
}

int main(int argc, char **argv) {
    if (argc != 7) { fprintf(stderr, "usage: parser_fuzz proof.bin config.json interner.bin pk_raw.bin iterations seed\n"); return 2; }
    const Bytes seeds[4] = {read_file(argv[1]), read_file(argv[2]), read_file(argv[3]), read_file(argv[4])};
    const unsigned long iters = strtoul(argv[5], nullptr, 10);
    rng_state = strtoull(argv[6], nullptr, 10);
    void (*const run[4])(const Bytes &) = {run_proof, run_config, run_interner, run_pk_raw};
    for (int k = 0; k < 4; k++) {   // the seeds themselves must parse
        const unsigned long before = n_accept[k == 1 ? 2 : k == 2 ? 1 : k];
        run[k](seeds[k]);
        if (n_accept[k == 1 ? 2 : k == 2 ? 1 : k] != before + 1) { fprintf(stderr, "parser_fuzz: seed file %d is not accepted\n", k); return 2; }
    }
    // every truncation of the short seeds, sampled ones of the long
    for (int k = 0; k < 4; k++) {
        const size_t n = seeds[k].size(), step = n > 6000 ? n / 3000 : 1;
        for (size_t cut = 0; cut < n; cut += step) { Bytes m(seeds[k].begin(), seeds[k].begin() + cut); run[k](m); }
    }
    for (unsigned long it = 0; it < iters; it++) {
        const int k = (int)below(4);
        Bytes m = mutate(seeds[k], k == 1);
        if (below(8) == 0) m = mutate(m, k == 1);
        run[k](m);
        if (below(16) == 0) run[below(4)](m);   // the wrong parser for the bytes
        if ((it & 15) == 0) run_helpers();
    }
    printf("{\"proof\": [%lu, %lu], \"interner\": [%lu, %lu], \"config\": [%lu, %lu], \"pk_raw\": [%lu, %lu]}\n", n_accept[0], n_refuse[0], n_accept[1], n_refuse[1], n_accept[2],
           n_refuse[2], n_accept[3], n_refuse[3]);
    return 0;
}
