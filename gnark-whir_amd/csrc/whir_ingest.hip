// ProveKit artefact ingestion: the host-side step BEFORE the prove path -- what /root/reference/main.go and the top
// of mt.go do with the files the Rust prover wrote -- as pure host code behind the C-ABI (no device, no HIP call).
//
//   mi_whir_proof_decode      the arkworks canonical stream of ProofObject (main.go:35-39), read at main.go:101 through
//                             go_ark_serialize.CanonicalDeserializeWithMode(proofFile, &proof, false, false)
//   mi_whir_parse_paths       ParsePathsObject, mt.go:229-304: prefix-compressed Merkle multipaths -> one authentication path per leaf
//   mi_whir_reverse           utilities.Reverse, utilities/utilities.go:58-65
//   mi_whir_prefix_decode_path utilities.PrefixDecodePath, utilities/utilities.go:67-78
//   mi_whir_limbs_to_fr       typeConverters.LimbsToBigIntMod, typeConverters/typeConverters.go:26-44
//   mi_whir_interner_decode   Interner{Values []Fp256} (main.go:74-76) read at main.go:146
//   mi_whir_matrix_cells      the CSR -> MatrixCell loops of verify_circuit, mt.go:358-401
//   mi_whir_config_parse      Config (main.go:41-58) as json.Unmarshal fills it at main.go:115
//
// go-ark-serialize (go.mod:10) is third-party and absent from /root/reference: the wire format is restated from the published
// ark-serialize rules (u64 / usize = 8 bytes little-endian; Vec<T> = u64 length + elements; [u8; 32] = 32 raw bytes; Fp256 = 4 x u64
// limbs, limb 0 first, canonical value; structs = fields in declaration order).  oracle/whir_ingest.py is the Python restatement the
// tests compare with; nothing reference-held pins either (no ProveKit artefact in the container): parity unpinned.
#include "../../include/mi355x_whir_ingest.h"
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {
typedef unsigned __int128 u128;
struct Digest { uint8_t b[32]; };
struct Fp256 { uint64_t l[4]; };
struct MultiPath {   // main.go:23-28
    std::vector<Digest> leaf_sibling_hashes;
    std::vector<uint64_t> prefix_lengths;
    std::vector<std::vector<Digest>> suffixes;
    std::vector<uint64_t> leaf_indexes;
};
struct ProofElement { MultiPath a; std::vector<std::vector<Fp256>> b; };   // main.go:30-33

struct Rd {
    const uint8_t *p; size_t n, i = 0; bool ok = true;
    bool take(void *dst, size_t k) { if (!ok || k > n - i) { ok = false; return false; } std::memcpy(dst, p + i, k); i += k; return true; }
    uint64_t u64() { uint8_t t[8]; if (!take(t, 8)) return 0; uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | t[k]; return v; }
    // a vector's length: every element takes at least min_elem bytes, so a count the rest of the input cannot hold is refused BEFORE anything is reserved
    uint64_t len(size_t min_elem) { const uint64_t v = u64(); if (ok && v > (n - i) / (min_elem ? min_elem : 1)) ok = false; return ok ? v : 0; }
};
bool rd_digests(Rd &r, std::vector<Digest> &v) { const uint64_t k = r.len(32); v.resize(k); for (auto &d : v) r.take(d.b, 32); return r.ok; }
bool rd_u64s(Rd &r, std::vector<uint64_t> &v) { const uint64_t k = r.len(8); v.resize(k); for (auto &x : v) x = r.u64(); return r.ok; }
bool rd_fps(Rd &r, std::vector<Fp256> &v) { const uint64_t k = r.len(32); v.resize(k); for (auto &x : v) for (int j = 0; j < 4; j++) x.l[j] = r.u64(); return r.ok; }
bool rd_element(Rd &r, ProofElement &e) {
    if (!rd_digests(r, e.a.leaf_sibling_hashes) || !rd_u64s(r, e.a.prefix_lengths)) return false;
    const uint64_t ns = r.len(8);
    e.a.suffixes.resize(ns);
    for (auto &s : e.a.suffixes) if (!rd_digests(r, s)) return false;
    if (!rd_u64s(r, e.a.leaf_indexes)) return false;
    const uint64_t nb = r.len(8);
    e.b.resize(nb);
    for (auto &leaf : e.b) if (!rd_fps(r, leaf)) return false;
    return r.ok;
}
bool rd_elements(Rd &r, std::vector<ProofElement> &v) {
    const uint64_t k = r.len(40);   // (five length words at least)
    v.resize(k);
    for (auto &e : v) if (!rd_element(r, e)) return false;
    return r.ok;
}

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (typeConverters/typeConverters.go:28)
const uint64_t R_LIMBS[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
bool geq_r(const uint64_t x[4]) { for (int i = 3; i >= 0; i--) { if (x[i] > R_LIMBS[i]) return true; if (x[i] < R_LIMBS[i]) return false; } return true; }
void sub_r(uint64_t x[4]) { u128 b = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)x[i] - R_LIMBS[i] - b; x[i] = (uint64_t)d; b = (d >> 64) & 1; } }
// x mod r for any 256-bit x: 2^256 < 6 r, so at most five subtractions
void reduce_mod_r(const uint64_t in[4], uint64_t out[4]) { std::memcpy(out, in, 32); while (geq_r(out)) sub_r(out); }

// decimal string -> 256-bit integer (big.Int SetString(s, 10), mt.go:310,352); false on a non-digit or a value >= 2^256
bool dec_to_u256(const std::string &s, uint64_t out[4]) {
    std::memset(out, 0, 32);
    if (s.empty()) return false;
    for (char ch : s) {
        if (ch < '0' || ch > '9') return false;
        u128 carry = (u128)(ch - '0');
        for (int i = 0; i < 4; i++) { const u128 t = (u128)out[i] * 10 + carry; out[i] = (uint64_t)t; carry = t >> 64; }
        if (carry) return false;
    }
    return true;
}

// ---- a JSON reader for the flat Config object (main.go:41-58): numbers, strings, arrays of numbers / strings.
// It accepts what encoding/json's Unmarshal accepts for that struct and nothing else that would change a field: strict literals (true,
// false, null spelled out), strict numbers (no leading zeros; an int field refuses fractions, exponents and values outside int64), strict
// string escapes (\" \\ \/ \b \f \n \r \t \uXXXX; control characters refused), an unpaired surrogate escape becomes U+FFFD, null leaves a
// field as it is, keys match ASCII-case-insensitively, the last of duplicate keys wins, nothing but white space may follow the value,
// nesting deeper than 10000 is refused (Go's limit; skip() keeps its own stack, so the depth of the input costs no call stack).
// Two differences remain, both on the refusing side and both documented in the header: invalid UTF-8 inside a string is copied as it is
// (Go substitutes U+FFFD), and Go's fold of the Kelvin sign / long s onto k / s in key names is not reproduced.
struct Js {
    const char *p, *e; bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool eat(char c) { ws(); if (p < e && *p == c) { p++; return true; } return false; }
    bool lit(const char *word) {   // the exact literal, followed by a delimiter or the end
        ws();
        const size_t n = std::strlen(word);
        if ((size_t)(e - p) < n || std::memcmp(p, word, n) != 0) return false;
        if ((size_t)(e - p) > n) { const char c = p[n]; if (!(c == ',' || c == '}' || c == ']' || c == ' ' || c == '\n' || c == '\t' || c == '\r')) return false; }
        p += n;
        return true;
    }
    static void utf8(std::string &out, unsigned v) {
        if (v < 0x80) out += (char)v;
        else if (v < 0x800) { out += (char)(0xc0 | (v >> 6)); out += (char)(0x80 | (v & 63)); }
        else if (v < 0x10000) { out += (char)(0xe0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 63)); out += (char)(0x80 | (v & 63)); }
        else { out += (char)(0xf0 | (v >> 18)); out += (char)(0x80 | ((v >> 12) & 63)); out += (char)(0x80 | ((v >> 6) & 63)); out += (char)(0x80 | (v & 63)); }
    }
    // q points at the 'u' of \uXXXX; needs q[1..4] inside the input
    bool hex4(const char *q, unsigned &v) const {
        if (e - q < 5) return false;
        v = 0;
        for (int k = 1; k <= 4; k++) {
            const char h = q[k];
            const unsigned d = h >= '0' && h <= '9' ? (unsigned)(h - '0') : h >= 'a' && h <= 'f' ? (unsigned)(h - 'a' + 10) : h >= 'A' && h <= 'F' ? (unsigned)(h - 'A' + 10) : 99u;
            if (d > 15) return false;
            v = v * 16 + d;
        }
        return true;
    }
    bool str(std::string &out) {
        ws();
        if (p >= e || *p != '"') return ok = false;
        p++; out.clear();
        while (p < e && *p != '"') {
            if ((unsigned char)*p < 0x20) return ok = false;   // a control character inside a string literal
            if (*p != '\\') { out += *p++; continue; }
            if (++p >= e) return ok = false;
            switch (*p) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': {   // \uXXXX -> UTF-8; a surrogate pair is one code point, an unpaired surrogate U+FFFD -- as encoding/json does
                    unsigned v = 0, lo = 0;
                    if (!hex4(p, v)) return ok = false;
                    p += 4;   // on the last hex digit
                    if (v >= 0xd800 && v < 0xdc00 && e - p > 6 && p[1] == '\\' && p[2] == 'u' && hex4(p + 2, lo) && lo >= 0xdc00 && lo < 0xe000) {
                        v = 0x10000 + ((v - 0xd800) << 10) + (lo - 0xdc00);
                        p += 6;
                    } else if (v >= 0xd800 && v < 0xe000) v = 0xfffd;
                    utf8(out, v);
                    break;
                }
                default: return ok = false;   // no other escape exists
            }
            p++;
        }
        if (p >= e) return ok = false;
        p++;
        return true;
    }
    // a JSON number at p: [-] (0 | [1-9][0-9]*) [. digits] [(e|E) [+-] digits]; *plain = it has no fraction and no exponent
    bool number_span(const char *&end, bool *plain) const {
        const char *q = p;
        if (q < e && *q == '-') q++;
        if (q >= e) return false;
        if (*q == '0') q++;
        else if (*q >= '1' && *q <= '9') { while (q < e && *q >= '0' && *q <= '9') q++; }
        else return false;
        *plain = true;
        if (q < e && *q == '.') { *plain = false; q++; if (q >= e || *q < '0' || *q > '9') return false; while (q < e && *q >= '0' && *q <= '9') q++; }
        if (q < e && (*q == 'e' || *q == 'E')) {
            *plain = false; q++;
            if (q < e && (*q == '+' || *q == '-')) q++;
            if (q >= e || *q < '0' || *q > '9') return false;
            while (q < e && *q >= '0' && *q <= '9') q++;
        }
        if (q < e && !(*q == ',' || *q == '}' || *q == ']' || *q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) return false;
        end = q;
        return true;
    }
    // an int field: *is_null = the value was null (Go leaves the field alone)
    bool integer(int64_t &v, bool *is_null = nullptr) {
        ws();
        if (is_null) { *is_null = false; if (lit("null")) { *is_null = true; return true; } }
        const char *end = nullptr;
        bool plain = false;
        if (!number_span(end, &plain) || !plain) return ok = false;   // not a number, or a fraction / exponent for an int
        const bool neg = *p == '-';
        u128 acc = 0;
        for (const char *q = p + (neg ? 1 : 0); q < end; q++) {
            acc = acc * 10 + (unsigned)(*q - '0');
            if (acc > ((u128)1 << 63)) return ok = false;
        }
        if (acc == ((u128)1 << 63) && !neg) return ok = false;   // 9223372036854775808 does not fit an int64
        v = neg ? (int64_t)(0 - (uint64_t)acc) : (int64_t)acc;
        p = end;
        return true;
    }
    bool skip() {   // any value; the nesting is kept on a stack of our own
        std::vector<char> open;
        for (;;) {
            ws();
            if (p >= e) return ok = false;
            bool value_done = false;
            if (*p == '"') { std::string t; if (!str(t)) return false; value_done = true; }
            else if (*p == '{' || *p == '[') {
                if (open.size() >= 10000) return ok = false;
                const char o = *p++;
                open.push_back(o);
                if (eat(o == '{' ? '}' : ']')) { open.pop_back(); value_done = true; }
                else if (o == '{') { std::string k; if (!str(k) || !eat(':')) return ok = false; }
            } else if (lit("true") || lit("false") || lit("null")) value_done = true;
            else { const char *end = nullptr; bool plain; if (!number_span(end, &plain)) return ok = false; p = end; value_done = true; }
            while (value_done) {   // what follows a complete value
                if (open.empty()) return true;
                if (eat(',')) { if (open.back() == '{') { std::string k; if (!str(k) || !eat(':')) return ok = false; } value_done = false; }
                else if (eat(open.back() == '{' ? '}' : ']')) open.pop_back();
                else return ok = false;
            }
        }
    }
};
// ASCII-case-insensitive key match (encoding/json prefers an exact match and falls back to a case-insensitive one; the tags are unique
// under folding, so the two rules pick the same field)
bool key_is(const std::string &key, const char *name) {
    const size_t n = std::strlen(name);
    if (key.size() != n) return false;
    for (size_t i = 0; i < n; i++) {
        char a = key[i], b = name[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}
int b64v(char c) { return c >= 'A' && c <= 'Z' ? c - 'A' : c >= 'a' && c <= 'z' ? c - 'a' + 26 : c >= '0' && c <= '9' ? c - '0' + 52 : c == '+' ? 62 : c == '/' ? 63 : -1; }
}  // namespace

struct mi_whir_proof {
    std::vector<ProofElement> rounds[2];   // 0: round0_merkle_paths (FirstRoundPaths), 1: merkle_paths
    std::vector<Fp256> statement_values;
};
struct mi_whir_config_store { std::string io_pattern; std::vector<uint8_t> transcript; std::vector<uint64_t> evals; };

extern "C" {

int32_t mi_whir_proof_decode(const uint8_t *buf, size_t len, mi_whir_proof **out, size_t *consumed) {
    if ((!buf && len) || !out) return MI_EINVAL;
    *out = nullptr;
    mi_whir_proof *p = new (std::nothrow) mi_whir_proof();
    if (!p) return MI_ENOMEM;
    try {
        Rd r{buf, len};
        if (!rd_elements(r, p->rounds[0]) || !rd_elements(r, p->rounds[1]) || !rd_fps(r, p->statement_values)) { delete p; return MI_EINVAL; }
        if (consumed) *consumed = r.i;
    } catch (...) { delete p; return MI_ENOMEM; }
    *out = p;
    return MI_OK;
}
void mi_whir_proof_free(mi_whir_proof *p) { delete p; }
uint64_t mi_whir_proof_elements(const mi_whir_proof *p, int which) { return p && (which == 0 || which == 1) ? p->rounds[which].size() : 0; }
uint64_t mi_whir_proof_statement_values(const mi_whir_proof *p, uint64_t *limbs_out) {
    if (!p) return 0;
    if (limbs_out) for (size_t i = 0; i < p->statement_values.size(); i++) std::memcpy(limbs_out + 4 * i, p->statement_values[i].l, 32);
    return p->statement_values.size();
}
int32_t mi_whir_element_shape(const mi_whir_proof *p, int which, uint64_t i, mi_whir_shape *out) {
    if (!p || !out || (which != 0 && which != 1) || i >= p->rounds[which].size()) return MI_EINVAL;
    const ProofElement &e = p->rounds[which][i];
    out->n_leaves = e.a.leaf_indexes.size();
    out->tree_height = e.a.suffixes.empty() ? 0 : e.a.suffixes[0].size();   // mt.go:243
    out->total_leaf_values = 0;
    for (uint64_t z = 0; z < out->n_leaves && z < e.b.size(); z++) out->total_leaf_values += e.b[z].size();
    return MI_OK;
}
void mi_whir_limbs_to_fr(const uint64_t limbs[4], uint64_t out[4]) { reduce_mod_r(limbs, out); }
int32_t mi_whir_reverse(const void *in, size_t n, size_t elem_bytes, void *out) {
    if ((!in || !out) && n && elem_bytes) return MI_EINVAL;
    if (in == out) return MI_EINVAL;   // (the reference returns a fresh slice)
    for (size_t i = 0; i < n; i++) std::memcpy((char *)out + i * elem_bytes, (const char *)in + (n - 1 - i) * elem_bytes, elem_bytes);
    return MI_OK;
}
int32_t mi_whir_prefix_decode_path(const void *prev, size_t n_prev, uint64_t prefix_len, const void *suffix, size_t n_suffix, size_t elem_bytes,
                                   void *out, size_t *n_out) {
    if (!out || !n_out || (!suffix && n_suffix) || (!prev && prefix_len)) return MI_EINVAL;
    if (prefix_len > n_prev) return MI_EINVAL;   // Go: slice bounds out of range
    // prefixLen == 0: the suffix alone; else prevPath[:prefixLen] followed by the suffix (the two branches of utilities.go:68-77 write the same bytes)
    if (prefix_len) std::memcpy(out, prev, (size_t)prefix_len * elem_bytes);
    if (n_suffix) std::memcpy((char *)out + (size_t)prefix_len * elem_bytes, suffix, n_suffix * elem_bytes);
    *n_out = (size_t)prefix_len + n_suffix;
    return MI_OK;
}
int32_t mi_whir_parse_paths(const mi_whir_proof *p, int which, uint64_t i, uint8_t *auth_paths, uint8_t *leaf_sibling_hashes, uint64_t *leaf_indexes,
                            uint64_t *leaf_lengths, uint64_t *leaves) {
    if (!p || (which != 0 && which != 1) || i >= p->rounds[which].size()) return MI_EINVAL;
    const ProofElement &e = p->rounds[which][i];
    const size_t n = e.a.leaf_indexes.size();
    // what the Go code indexes without checking (it would panic): one suffix, prefix length, sibling hash and leaf per proved leaf
    if (!n || e.a.suffixes.size() < n || e.a.prefix_lengths.size() < n || e.a.leaf_sibling_hashes.size() < n || e.b.size() < n) return MI_EINVAL;
    const size_t height = e.a.suffixes[0].size();
    try {
        std::vector<Digest> prev = e.a.suffixes[0], next;   // mt.go:268: the first path is stored whole, root end first
        for (size_t j = 0; j < n; j++) {
            if (j) {   // mt.go:276: prevPath = PrefixDecodePath(prevPath, AuthPathsPrefixLengths[j], AuthPathsSuffixes[j])
                const uint64_t pl = e.a.prefix_lengths[j];
                const std::vector<Digest> &suf = e.a.suffixes[j];
                if (pl > prev.size()) return MI_EINVAL;
                next.resize((size_t)pl + suf.size());
                size_t cnt = 0;
                const int32_t rc = mi_whir_prefix_decode_path(prev.data(), prev.size(), pl, suf.data(), suf.size(), 32, next.data(), &cnt);
                if (rc != MI_OK) return rc;
                if (cnt != height) return MI_EINVAL;   // mt.go:278-280 reads treeHeight nodes of it
                prev.swap(next);
            }
            // mt.go:269,277: authPathsTemp[j] = Reverse(prevPath): leaf end first
            if (auth_paths) mi_whir_reverse(prev.data(), height, 32, auth_paths + j * height * 32);
        }
    } catch (...) { return MI_ENOMEM; }
    size_t off = 0;
    for (size_t z = 0; z < n; z++) {   // mt.go:284-291
        if (leaf_sibling_hashes) std::memcpy(leaf_sibling_hashes + 32 * z, e.a.leaf_sibling_hashes[z].b, 32);
        if (leaf_indexes) leaf_indexes[z] = e.a.leaf_indexes[z];
        if (leaf_lengths) leaf_lengths[z] = e.b[z].size();
        if (leaves) for (size_t j = 0; j < e.b[z].size(); j++) reduce_mod_r(e.b[z][j].l, leaves + 4 * (off + j));
        off += e.b[z].size();
    }
    return MI_OK;
}
int32_t mi_whir_interner_decode(const uint8_t *buf, size_t len, uint64_t *limbs_out, uint64_t *n_out, size_t *consumed) {
    if ((!buf && len) || !n_out) return MI_EINVAL;
    Rd r{buf, len};
    const uint64_t k = r.len(32);
    if (!r.ok) return MI_EINVAL;
    *n_out = k;
    if (limbs_out) for (uint64_t i = 0; i < k; i++) for (int j = 0; j < 4; j++) limbs_out[4 * i + j] = r.u64();
    else r.i += (size_t)k * 32;
    if (consumed) *consumed = r.i;
    return r.ok ? MI_OK : MI_EINVAL;
}
int32_t mi_whir_matrix_cells(const uint64_t *row_indices, size_t n_rows, const uint64_t *col_indices, const uint64_t *values, size_t nnz,
                             const uint64_t *interner_limbs, size_t n_interner, uint64_t *rows_out, uint64_t *cols_out, uint64_t *values_out) {
    if ((!row_indices && n_rows) || ((!col_indices || !values || !rows_out || !cols_out || !values_out) && nnz) || (!interner_limbs && n_interner)) return MI_EINVAL;
    std::vector<char> seen;
    try { seen.assign(nnz, 0); } catch (...) { return MI_ENOMEM; }
    for (size_t i = 0; i < n_rows; i++) {   // mt.go:359-372 (A; B and C are the same loop)
        if (!nnz) break;
        uint64_t end = nnz - 1;
        if (i + 1 < n_rows) { if (row_indices[i + 1] == 0) continue; end = row_indices[i + 1] - 1; }   // (row_indices[i+1] - 1 with int arithmetic: an empty range)
        for (uint64_t j = row_indices[i]; j <= end; j++) {
            if (j >= nnz || values[j] >= n_interner) return MI_EINVAL;   // Go: index out of range
            rows_out[j] = i; cols_out[j] = col_indices[j];
            reduce_mod_r(interner_limbs + 4 * values[j], values_out + 4 * j);
            seen[j] = 1;
        }
    }
    for (size_t j = 0; j < nnz; j++) if (!seen[j]) { rows_out[j] = 0; cols_out[j] = 0; std::memset(values_out + 4 * j, 0, 32); }   // Go's zero MatrixCell
    return MI_OK;
}

void mi_whir_config_free(mi_whir_config *c) {
    if (!c) return;
    delete (mi_whir_config_store *)c->store;
    delete c;
}
int32_t mi_whir_config_parse(const char *json, size_t len, mi_whir_config **out) {
    if ((!json && len) || !out) return MI_EINVAL;
    *out = nullptr;
    mi_whir_config *c = new (std::nothrow) mi_whir_config();
    mi_whir_config_store *st = new (std::nothrow) mi_whir_config_store();
    if (!c || !st) { delete c; delete st; return MI_ENOMEM; }
    std::memset(c, 0, sizeof(*c));
    c->store = st;
    auto fail = [&](int32_t rc) { mi_whir_config_free(c); return rc; };
    try {
        Js j{json, json + len};
        if (j.lit("null")) {   // Unmarshal of a top-level null leaves the struct as it is
            j.ws();
            if (j.p != j.e) return fail(MI_EINVAL);
            c->io_pattern = st->io_pattern.c_str();
            *out = c;
            return MI_OK;
        }
        if (!j.eat('{')) return fail(MI_EINVAL);
        struct IntField { const char *name; int64_t *dst; } ints[] = {
            {"log_num_constraints", &c->log_num_constraints}, {"n_rounds", &c->n_rounds}, {"n_vars", &c->n_vars}, {"final_queries", &c->final_queries},
            {"final_pow_bits", &c->final_pow_bits}, {"final_folding_pow_bits", &c->final_folding_pow_bits}, {"rate", &c->rate}, {"transcript_len", &c->transcript_len}};
        struct ListField { const char *name; int64_t *dst; uint32_t *n; } lists[] = {
            {"folding_factor", c->folding_factor, &c->n_folding_factor}, {"ood_samples", c->ood_samples, &c->n_ood_samples},
            {"num_queries", c->num_queries, &c->n_num_queries}, {"pow_bits", c->pow_bits, &c->n_pow_bits}};
        if (!j.eat('}')) for (;;) {
            std::string key;
            if (!j.str(key) || !j.eat(':')) return fail(MI_EINVAL);
            bool done = false;
            for (auto &f : ints) if (!done && key_is(key, f.name)) { bool nul; if (!j.integer(*f.dst, &nul)) return fail(MI_EINVAL); done = true; }
            for (auto &f : lists) if (!done && key_is(key, f.name)) {
                done = true;
                if (j.lit("null")) break;   // null: Go leaves the slice as it is
                *f.n = 0;
                if (!j.eat('[')) return fail(MI_EINVAL);
                if (!j.eat(']')) for (;;) {
                    if (*f.n >= MI_WHIR_MAX_ROUNDS) return fail(MI_EINVAL);
                    int64_t v = 0; bool nul;
                    if (!j.integer(v, &nul)) return fail(MI_EINVAL);   // (a null element is Go's zero value)
                    f.dst[(*f.n)++] = v;
                    if (j.eat(',')) continue;
                    if (!j.eat(']')) return fail(MI_EINVAL);
                    break;
                }
            }
            if (!done && key_is(key, "domain_generator")) {
                done = true;
                if (!j.lit("null")) {
                    std::string s;
                    if (!j.str(s)) return fail(MI_EINVAL);
                    if (!dec_to_u256(s, c->domain_generator)) return fail(MI_EINVAL);   // mt.go:310: big.Int SetString(s, 10)
                }
            }
            if (!done && key_is(key, "io_pattern")) { done = true; if (!j.lit("null") && !j.str(st->io_pattern)) return fail(MI_EINVAL); }
            if (!done && key_is(key, "transcript")) {   // []byte: a JSON array of numbers (serde_json's Vec<u8>), a base64 string, or null
                done = true;
                j.ws();
                if (j.lit("null")) { }
                else if (j.p < j.e && *j.p == '"') {
                    st->transcript.clear();
                    std::string s;
                    if (!j.str(s)) return fail(MI_EINVAL);
                    // base64.StdEncoding.DecodeString as encoding/json applies it: \r and \n are skipped, the rest must be whole padded quanta
                    std::string b;
                    for (char ch : s) if (ch != '\r' && ch != '\n') b += ch;
                    if (b.size() % 4 != 0) return fail(MI_EINVAL);
                    for (size_t q = 0; q < b.size(); q += 4) {
                        int v[4], pad = 0;
                        for (int k = 0; k < 4; k++) {
                            const char ch = b[q + k];
                            if (ch == '=') { if (q + 4 != b.size() || k < 2) return fail(MI_EINVAL); v[k] = 0; pad++; }
                            else { if (pad) return fail(MI_EINVAL); v[k] = b64v(ch); if (v[k] < 0) return fail(MI_EINVAL); }
                        }
                        const unsigned w = ((unsigned)v[0] << 18) | ((unsigned)v[1] << 12) | ((unsigned)v[2] << 6) | (unsigned)v[3];
                        st->transcript.push_back((uint8_t)(w >> 16));
                        if (pad < 2) st->transcript.push_back((uint8_t)(w >> 8));
                        if (pad < 1) st->transcript.push_back((uint8_t)w);
                    }
                } else if (j.eat('[')) {
                    st->transcript.clear();
                    if (!j.eat(']')) for (;;) {
                        int64_t v = 0; bool nul;
                        if (!j.integer(v, &nul) || v < 0 || v > 255) return fail(MI_EINVAL);
                        st->transcript.push_back((uint8_t)v);
                        if (j.eat(',')) continue;
                        if (!j.eat(']')) return fail(MI_EINVAL);
                        break;
                    }
                } else return fail(MI_EINVAL);   // a number, an object, true / false: not a []byte
            }
            if (!done && key_is(key, "statement_evaluations")) {
                done = true;
                if (j.lit("null")) { }
                else if (!j.eat('[')) return fail(MI_EINVAL);
                else if (st->evals.clear(), !j.eat(']')) for (;;) {
                    std::string s;
                    uint64_t v[4];
                    if (!j.str(s) || !dec_to_u256(s, v)) return fail(MI_EINVAL);   // mt.go:352: big.Int SetString(s, 10)
                    st->evals.insert(st->evals.end(), v, v + 4);
                    if (j.eat(',')) continue;
                    if (!j.eat(']')) return fail(MI_EINVAL);
                    break;
                }
            }
            if (!done && !j.skip()) return fail(MI_EINVAL);   // unknown keys are ignored, as encoding/json does
            if (j.eat(',')) continue;
            if (!j.eat('}')) return fail(MI_EINVAL);
            break;
        }
        j.ws();
        if (!j.ok || j.p != j.e) return fail(MI_EINVAL);   // nothing but white space may follow the object
    } catch (...) { return fail(MI_ENOMEM); }
    c->io_pattern = st->io_pattern.c_str(); c->io_pattern_len = st->io_pattern.size();
    c->transcript = st->transcript.data(); c->n_transcript = st->transcript.size();
    c->statement_evaluations = st->evals.data(); c->n_statement_evaluations = st->evals.size() / 4;
    *out = c;
    return MI_OK;
}

}  // extern "C"
