/* mi355x_whir_ingest.h -- host-only decoding of the files ProveKit writes (reference main.go:92-152, mt.go:229-401).
 * Part of libmi355x_groth16.so; no device needed; conventions as in mi355x_groth16.h. */
#ifndef MI355X_WHIR_INGEST_H
#define MI355X_WHIR_INGEST_H
#include "mi355x_groth16.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ProveKit artefact ingestion: what /root/reference/main.go:92-152 and the top of verify_circuit (mt.go:306-401) do
 * with the files the Rust prover wrote, BEFORE frontend.Compile -- pure host code, no device needed.  Their consumers are gnark's
 * frontend and solver (Go), so a Go caller binds these only to replace go-ark-serialize + the decoding loops; nothing of the prove path
 * depends on them.  The arkworks wire format is restated from the published ark-serialize rules (go-ark-serialize, go.mod:10, is absent
 * from the reference tree): parity unpinned until a real ProveKit `proof` file is decoded.
 *   mi_whir_proof_decode        go_ark_serialize.CanonicalDeserializeWithMode(proofFile, &proof, false, false), main.go:101, into
 *                               ProofObject (main.go:35-39: round0_merkle_paths, merkle_paths, statement_values_at_random_point)
 *   mi_whir_element_shape       leaves proved, tree height (= len(AuthPathsSuffixes[0]), mt.go:243), leaf values of one ProofElement
 *   mi_whir_parse_paths         ParsePathsObject, mt.go:229-304, for one ProofElement: auth_paths[j][z] = node z (leaf end first) of
 *                               leaf j's authentication path after PrefixDecodePath + Reverse; leaves reduced mod r (LimbsToBigIntMod)
 *   mi_whir_reverse             utilities.Reverse, utilities/utilities.go:58-65 (out must not alias in)
 *   mi_whir_prefix_decode_path  utilities.PrefixDecodePath, utilities/utilities.go:67-78 (MI_EINVAL where Go would panic: prefix_len > n_prev)
 *   mi_whir_limbs_to_fr         typeConverters.LimbsToBigIntMod, typeConverters/typeConverters.go:26-44: 4 x u64 little-endian limbs
 *                               -> the canonical value mod r, same limb order (NOT Montgomery: multiply by R for an mi_fr)
 *   mi_whir_interner_decode     Interner{Values []Fp256}, main.go:74-76,146
 *   mi_whir_matrix_cells        the CSR -> MatrixCell loops of mt.go:358-401 (row i owns entries [row_indices[i], row_indices[i+1] - 1],
 *                               the last row runs to the end; value = LimbsToBigIntMod(interner[values[j]]))
 *   mi_whir_config_parse        json.Unmarshal into Config, main.go:41-58,115: unknown keys ignored, missing keys zero, null leaves a field as
 *                               it is (a top-level null too), keys match ASCII-case-insensitively, the last duplicate wins; strict literals,
 *                               numbers (no leading zeros; an int refuses fractions, exponents, values outside int64) and string escapes
 *                               (an unpaired \uD800-\uDFFF escape becomes U+FFFD); nothing but white space may follow the object; nesting
 *                               deeper than 10000 is refused.  `transcript` as a JSON array of numbers or a padded base64 string (\r, \n
 *                               skipped); decimal strings -> 4 x u64 limbs.  Refused although Go would accept: more than
 *                               MI_WHIR_MAX_ROUNDS list entries, a decimal string that is empty / not a number / >= 2^256 (Go keeps the string
 *                               and fails later, mt.go:310,352).  Copied as they are although Go substitutes U+FFFD: invalid UTF-8 bytes
 *                               inside a string.
 * These readers take bytes an outside party wrote: they run under AddressSanitizer / UBSan with a mutation driver on every CPU test run
 * (gnark-whir_amd/Makefile `sanitize`, tests/test_parsers_sanitized.py), together with mi_pk_raw_inspect. ---- */
typedef struct mi_whir_proof mi_whir_proof;
typedef struct mi_whir_shape { uint64_t n_leaves, tree_height, total_leaf_values; } mi_whir_shape;
int32_t mi_whir_proof_decode(const uint8_t *buf, size_t len, mi_whir_proof **out, size_t *consumed_or_null);
void mi_whir_proof_free(mi_whir_proof *p);
uint64_t mi_whir_proof_elements(const mi_whir_proof *p, int which /* 0 = round0_merkle_paths, 1 = merkle_paths */);
uint64_t mi_whir_proof_statement_values(const mi_whir_proof *p, uint64_t *limbs_out /* count x 4 raw limbs, may be NULL */);
int32_t mi_whir_element_shape(const mi_whir_proof *p, int which, uint64_t i, mi_whir_shape *out);
int32_t mi_whir_parse_paths(const mi_whir_proof *p, int which, uint64_t i, uint8_t *auth_paths /* n_leaves x tree_height x 32 */,
                            uint8_t *leaf_sibling_hashes /* n_leaves x 32 */, uint64_t *leaf_indexes /* n_leaves */,
                            uint64_t *leaf_lengths /* n_leaves */, uint64_t *leaves /* total_leaf_values x 4 */);   /* any output may be NULL */
int32_t mi_whir_reverse(const void *in, size_t n, size_t elem_bytes, void *out);
int32_t mi_whir_prefix_decode_path(const void *prev, size_t n_prev, uint64_t prefix_len, const void *suffix, size_t n_suffix,
                                   size_t elem_bytes, void *out /* (prefix_len + n_suffix) elements */, size_t *n_out);
void mi_whir_limbs_to_fr(const uint64_t limbs[4], uint64_t out[4]);
int32_t mi_whir_interner_decode(const uint8_t *buf, size_t len, uint64_t *limbs_out /* count x 4, may be NULL */, uint64_t *n_out, size_t *consumed_or_null);
int32_t mi_whir_matrix_cells(const uint64_t *row_indices, size_t n_rows, const uint64_t *col_indices, const uint64_t *values, size_t nnz,
                             const uint64_t *interner_limbs, size_t n_interner, uint64_t *rows_out, uint64_t *cols_out, uint64_t *values_out /* nnz x 4 */);
#define MI_WHIR_MAX_ROUNDS 64
typedef struct mi_whir_config {   /* Config, main.go:41-58; pointers are owned by the config (mi_whir_config_free) */
    int64_t log_num_constraints, n_rounds, n_vars, final_queries, final_pow_bits, final_folding_pow_bits, rate, transcript_len;
    int64_t folding_factor[MI_WHIR_MAX_ROUNDS], ood_samples[MI_WHIR_MAX_ROUNDS], num_queries[MI_WHIR_MAX_ROUNDS], pow_bits[MI_WHIR_MAX_ROUNDS];
    uint32_t n_folding_factor, n_ood_samples, n_num_queries, n_pow_bits;
    uint64_t domain_generator[4];              /* the decimal string as an integer (mt.go:310), little-endian limbs */
    const char *io_pattern; size_t io_pattern_len;
    const uint8_t *transcript; size_t n_transcript;
    const uint64_t *statement_evaluations; size_t n_statement_evaluations;   /* decimal strings (mt.go:352) -> 4 limbs each */
    void *store;
} mi_whir_config;
int32_t mi_whir_config_parse(const char *json, size_t len, mi_whir_config **out);
void mi_whir_config_free(mi_whir_config *c);


#ifdef __cplusplus
}
#endif
#endif
