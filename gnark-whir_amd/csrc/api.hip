// Lifecycle, device-memory helpers and the host-only pieces of the C-ABI (point encoding,
// partial-sum combine).  See include/mi355x_groth16.h for the contract of each entry point.
#include "ctx.h"
#include "curve.cuh"
#include <cstring>
#include <dlfcn.h>

thread_local std::string *mi_err_sink = nullptr;
std::atomic<int> mi_fault_countdown{0};

// ---------------------------------------------------------------- roctx ranges ("roctx ranges around NTT/MSM phases")
std::atomic<int> mi_ranges_on{0};
static int (*g_range_push)(const char *) = nullptr;
static int (*g_range_pop)() = nullptr;
void mi_range_push(const char *name) { if (g_range_push) (void)g_range_push(name); }
void mi_range_pop() { if (g_range_pop) (void)g_range_pop(); }
static bool ranges_load() {
    static std::once_flag once;
    std::call_once(once, [] {
        // rocprofv3's marker trace reads the SDK's roctx; roctracer's libroctx64 is the older name of the same two entry points
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_range_push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            g_range_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (g_range_push && g_range_pop) return;
            g_range_push = nullptr; g_range_pop = nullptr;
        }
    });
    return g_range_push && g_range_pop;
}

int32_t mi_copy_stream(mi_ctx *ctx, hipStream_t *out) {
    if (!ctx->copy_stream) MI_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    *out = ctx->copy_stream;
    return MI_OK;
}

extern "C" {

int32_t mi_set_trace_ranges(int32_t on) {
    if (on && !ranges_load()) return MI_ENODEV;   // no roctx library on this machine
    mi_ranges_on.store(on ? 1 : 0);
    return MI_OK;
}
int32_t mi_debug_set_trace_ranges(int32_t on) { return mi_set_trace_ranges(on); }
int32_t mi_debug_inject_hip_failure(int32_t nth) { mi_fault_countdown.store(nth > 0 ? nth : 0); return MI_OK; }
int32_t mi_init(int device_id, mi_ctx **out) { return mi_init_prio(device_id, MI_PRIO_SOLO, out); }
int32_t mi_init_prio(int device_id, int prio_scheme, mi_ctx **out) {
    if (!out || prio_scheme < MI_PRIO_SOLO || prio_scheme > MI_PRIO_POOL_REST) return MI_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return MI_ENODEV;
    if (hipSetDevice(device_id) != hipSuccess) return MI_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return MI_ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MI_ENODEV;  // gfx950-only code object
    mi_ctx *ctx = new (std::nothrow) mi_ctx();
    if (!ctx) return MI_ENOMEM;
    ctx->dev = device_id;
    ctx->cu_count = prop.multiProcessorCount;
    {   // the context's stream carries computeH, the head of a proof's longest chain (h -> Z MSM): high priority (see msm.hip)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        ctx->prio_scheme = prio_scheme;
        int ph = prio_hi;   // MI_PRIO_SOLO, MI_PRIO_POOL_FIRST
        if (prio_scheme == MI_PRIO_POOL_SECOND) ph = 0;
        if (prio_scheme == MI_PRIO_POOL_REST) ph = prio_lo;
        if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, ph) != hipSuccess) { (void)hipGetLastError(); delete ctx; return MI_EHIP; }
    }
    ctx->own_stream = true;
    // (ctx->copy_stream is created by its first user, mi_copy_stream: a stream takes a share of a hardware queue from the moment it
    //  exists, and the contexts of a prover pool never copy through theirs)
    mi_ntt_state_init(ctx);
    // from here on a failure unwinds through mi_shutdown: it frees exactly what exists (null handles are skipped)
    int32_t rc = mi_msm_state_init(ctx);
    for (auto &e : ctx->ev)
        if (rc == MI_OK && hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); rc = MI_EHIP; }
    if (rc != MI_OK) { mi_shutdown(ctx); return rc; }
    *out = ctx;
    return MI_OK;
}
int32_t mi_shutdown(mi_ctx *ctx) {
    if (!ctx) return MI_EINVAL;
    (void)hipSetDevice(ctx->dev);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->ws) if (b.p) (void)hipFree(b.p);
    mi_ntt_state_free(ctx);
    mi_msm_state_free(ctx);
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    delete ctx;
    return MI_OK;
}
// Gives the grow-only workspaces of an IDLE context back to the device: scratch buffers, every MSM slot's sort / partial-sum / bucket
// arrays, the NTT tables (rebuilt in a few ms by the next transform of a size).  Streams, events and pinned result buffers stay.  The
// next call grows what it needs again (hipMalloc, as on first use).  The caller guarantees that no call is running on ctx.
int32_t mi_ctx_trim(mi_ctx *ctx) {
    if (!ctx) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipSetDevice(ctx->dev));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (auto &sl : ctx->msm) if (sl.stream) MI_CHECK_HIP(ctx, hipStreamSynchronize(sl.stream));
    for (auto &b : ctx->ws) if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    for (auto &sl : ctx->msm) for (auto &b : sl.buf) if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    mi_ntt_state_trim(ctx);
    return MI_OK;
}
const char *mi_last_error(mi_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int32_t mi_set_stream(mi_ctx *ctx, void *hip_stream) {
    if (!ctx) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return MI_OK;
}
int32_t mi_get_stats(mi_ctx *ctx, mi_stats *out) {
    if (!ctx || !out) return MI_EINVAL;
    *out = ctx->stats;
    return MI_OK;
}
int32_t mi_dev_alloc(mi_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipSetDevice(ctx->dev));
    MI_CHECK_HIP(ctx, hipMalloc(out, bytes ? bytes : 32));
    return MI_OK;
}
int32_t mi_dev_free(mi_ctx *ctx, void *dev) {
    if (!ctx) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipFree(dev));
    return MI_OK;
}
int32_t mi_dev_upload(mi_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || ((!dst || !src) && bytes)) return MI_EINVAL;
    if (bytes) MI_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
int32_t mi_dev_download(mi_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || ((!dst || !src) && bytes)) return MI_EINVAL;
    if (bytes) MI_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}
int32_t mi_dev_sync(mi_ctx *ctx) {
    if (!ctx) return MI_EINVAL;
    MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

// ---------------------------------------------------------------- encoding (Proof.WriteTo, marshal.go by behaviour)
static void fp_be_bytes(const Fp &mont, uint8_t out[32]) {
    Fp c = fe_from_mont(mont);
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++) out[31 - (4 * i + k)] = (uint8_t)(c.l[i] >> (8 * k));
}
static bool fp_lex_largest(const Fp &mont) {  // canonical y > (q-1)/2
    static const u32 half[8] = {0x6c3e7ea3u, 0x9e10460bu, 0xb438e546u, 0xcbc0b548u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
    Fp c = fe_from_mont(mont);
    for (int i = 7; i >= 0; i--) {
        if (c.l[i] > half[i]) return true;
        if (c.l[i] < half[i]) return false;
    }
    return false;
}
void mi_g1_compress(const mi_g1_affine *p, uint8_t out[32]) {
    G1Aff a;
    std::memcpy(&a, p, sizeof(a));
    if (a.is_inf()) { std::memset(out, 0, 32); out[0] = 0x40; return; }
    fp_be_bytes(a.x, out);
    out[0] |= fp_lex_largest(a.y) ? 0xC0 : 0x80;
}
void mi_g2_compress(const mi_g2_affine *p, uint8_t out[64]) {
    G2Aff a;
    std::memcpy(&a, p, sizeof(a));
    if (a.is_inf()) { std::memset(out, 0, 64); out[0] = 0x40; return; }
    fp_be_bytes(a.x.a1, out);
    fp_be_bytes(a.x.a0, out + 32);
    bool largest = a.y.a1.is_zero() ? fp_lex_largest(a.y.a0) : fp_lex_largest(a.y.a1);
    out[0] |= largest ? 0xC0 : 0x80;
}
size_t mi_proof_write(const mi_proof_out *proof, const mi_g1_affine *commitments, uint32_t n_commitments,
                      const mi_g1_affine *pok, uint8_t *out) {
    uint8_t *p = out;
    mi_g1_compress(&proof->ar, p); p += 32;
    mi_g2_compress(&proof->bs, p); p += 64;
    mi_g1_compress(&proof->krs, p); p += 32;
    p[0] = (uint8_t)(n_commitments >> 24); p[1] = (uint8_t)(n_commitments >> 16);
    p[2] = (uint8_t)(n_commitments >> 8); p[3] = (uint8_t)n_commitments; p += 4;
    for (uint32_t i = 0; i < n_commitments; i++) { mi_g1_compress(&commitments[i], p); p += 32; }
    if (pok) mi_g1_compress(pok, p);
    else { std::memset(p, 0, 32); p[0] = 0x40; }
    p += 32;
    return (size_t)(p - out);
}

}  // extern "C"
// ---------------------------------------------------------------- partial-sum combine (host)
template <class F, class JacT>
static void sum_parts(const JacT *parts, size_t n, JacT *out) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = 0; i < n; i++) {
        Jac<F> j;
        std::memcpy(&j, &parts[i], sizeof(j));
        if (j.z.is_zero()) continue;
        // Jacobian (X,Y,Z) -> XYZZ (X, Y, Z^2, Z^3)
        F zz = fe_sqr(j.z);
        XYZZ<F> q{j.x, j.y, zz, zz * j.z};
        xyzz_add(acc, q);
    }
    Jac<F> r;
    if (acc.is_inf()) r = Jac<F>{F::one(), F::one(), F::zero()};
    else { Affine<F> a = xyzz_to_affine(acc); r = Jac<F>{a.x, a.y, F::one()}; }
    std::memcpy(out, &r, sizeof(r));
}
extern "C" {
int32_t mi_g1_sum(const mi_g1_jac *parts, size_t n, mi_g1_jac *out) {
    if ((!parts && n) || !out) return MI_EINVAL;
    sum_parts<Fp>(parts, n, out);
    return MI_OK;
}
int32_t mi_g2_sum(const mi_g2_jac *parts, size_t n, mi_g2_jac *out) {
    if ((!parts && n) || !out) return MI_EINVAL;
    sum_parts<Fp2>(parts, n, out);
    return MI_OK;
}
}
