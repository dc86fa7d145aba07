/*
 * dpfhe.h - C ABI of libdpfhe_hip.so: the MI355X (gfx950) implementation of DeepPowers' FHE
 * ciphertext-arithmetic hot path (negacyclic NTT / inverse NTT over Z_q, coefficient-wise modular
 * mul/add, ciphertext x ciphertext and ciphertext x plaintext multiply).
 *
 * What each entry point replaces in the reference: NOTHING EXISTS THERE.  deeppowers/deeppowers has no
 * Ciphertext/Evaluator/NTT code, so there is no FFI surface to bind to.  These are
 * the entry points SURVEY.md section 8(b) specifies for the path, shaped by the reference's plugin seam
 * and conventions:
 *   - device placement + raw device pointers + optional stream last:
 *       /root/reference/src/core/hal/hal.hpp:36-57 (Device::allocate/memcpy), :95 (launch(cfg, Stream*))
 *   - error numbers are deeppowers::common::ErrorCode values:
 *       /root/reference/src/common/error.hpp:10-40 (SUCCESS=0, OUT_OF_MEMORY=1001, DEVICE_ERROR=1002,
 *       INVALID_ARGUMENT=2000, INVALID_STATE=2002, RUNTIME_ERROR=3000)
 *   - the encrypted matmul sites an Evaluator would serve (callers of dpfhe_matvec_plain):
 *       /root/reference/src/core/execution/models/gpt_model.cpp:793 (QKV), :848 (FFN), :883 (logits)
 *   - the one collective kept (dpfhe_comm_allgather):
 *       /root/reference/src/core/distributed/distributed_context.cpp:97-122 (ncclAllGather), without its
 *       per-call cudaMalloc/stream-create (:109, :88).
 *
 * Conventions
 *   - All data buffers are DEVICE pointers owned by the caller, 16-byte aligned, holding little-endian
 *     u64 words that are canonical residues in [0, q_limb).  Layout: [batch][component][limb][N], contiguous.
 *     "n_rns_polys" counts RNS polynomials (L limbs x N words each); a 2-component ciphertext is 2 of them.
 *   - forward NTT: natural order in, BIT-REVERSED order out, ahat[k] = sum_j a[j] psi^((2 brv(k)+1) j);
 *     inverse NTT: bit-reversed in, natural out, including N^-1.  Outputs are canonical.
 *   - `stream` is a hipStream_t (NULL = the null stream).  Compute calls only enqueue work and never synchronise.  At log2_n <= 13 (every
 *     BASELINE configuration) they never allocate either: all scratch is caller-provided (d_work / d_digits / ... arguments).  The ONE
 *     exception is spelled out at dpfhe_ctx_create below: at log2_n >= 14 dpfhe_ct_mul, dpfhe_relinearize, dpfhe_switch_key and the two
 *     hybrid entries take their scratch from an arena the context keeps PER STREAM: allocated (hipMalloc) the first time that stream runs such an
 *     operation, grown - after synchronising that stream only - when a larger slice than ever before arrives, kept until dpfhe_ctx_destroy; in steady
 *     state they neither allocate nor synchronise.  dpfhe_ctx_create (uploads the tables: one allocation, one copy, no kernel), dpfhe_ctx_autotune
 *     (times kernels on caller scratch) and dpfhe_comm_create are set-up calls: they may allocate and synchronise.  After set-up a dpfhe_ctx is immutable: concurrent calls from different host threads on
 *     different streams are allowed.
 *   - No C++ types and no exceptions cross this boundary.
 */
#ifndef DPFHE_H
#define DPFHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dpfhe_ctx dpfhe_ctx;
typedef struct dpfhe_comm dpfhe_comm;

/* deeppowers::common::ErrorCode numbers (/root/reference/src/common/error.hpp:10-40) */
enum {
    DPFHE_SUCCESS = 0,
    DPFHE_OUT_OF_MEMORY = 1001,
    DPFHE_DEVICE_ERROR = 1002,
    DPFHE_INVALID_ARGUMENT = 2000,
    DPFHE_INVALID_STATE = 2002,
    DPFHE_RUNTIME_ERROR = 3000
};

/* dpfhe_ct_mul flags: which domain the inputs are in / the output is wanted in */
enum { DPFHE_IN_NTT = 1u, DPFHE_OUT_NTT = 2u };

/* -- A0: context ----------------------------------------------------------------------------------
 * log2_n in [8, 16] (N > 16384 runs a two-kernel split transform; the fused ct x ct / key-switch kernels stop at 13: above it dpfhe_ct_mul,
 * dpfhe_relinearize and dpfhe_switch_key compose the batched transforms with one-pass streaming kernels and take their scratch from the
 * context's scratch arena of the caller's stream (see "Conventions"; kept until dpfhe_ctx_destroy; large batches run in slices of at most
 * 1 GiB of scratch, or what dpfhe_ctx_set_scratch_limit said); so do the hybrid entries - dpfhe_relinearize_hybrid / dpfhe_switch_key_hybrid, and since round 5 the per-item-key ones:
 * dpfhe_rotate_hybrid_batch / dpfhe_rotate_hybrid_grouped and dpfhe_switch_key_qp; dpfhe_rotate_hoisted_qp and dpfhe_ntt_inv_galois run up to log2_n = 14
 * (the whole packed-layer pipeline at N = 16384), and so does dpfhe_rotate_hybrid_hoisted (there: the deferred-division pipeline + one inverse transform
 * and division by P per rotation, scratch from the arena)); n_limbs >= 1; moduli[i] prime < 2^60 with q = 1 (mod 2N); psi[i] a primitive
 * 2N-th root of unity mod q_i (psi^N = -1).  Builds twiddle / Shoup / Barrett tables on device_id. */
int dpfhe_ctx_create(dpfhe_ctx** out, uint32_t log2_n, uint32_t n_limbs, const uint64_t* moduli,
                     const uint64_t* psi, int device_id);
int dpfhe_ctx_destroy(dpfhe_ctx* ctx);
/* set-up call (before the context is shared between threads): the composed large-ring operations (log2_n >= 14) slice their batches so that
 * one slice's scratch stays below `mib` MiB (default 1024).  The library reads NO environment variable. */
int dpfhe_ctx_set_scratch_limit(dpfhe_ctx* ctx, size_t mib);
/* The composed operations take their scratch from one ARENA per (context, stream): a device allocation made at that stream's first composed call (<= the
 * scratch limit above, in 16 MiB steps), grown only for a larger slice than ever before (growth synchronises that stream), reused without allocation or
 * synchronisation from then on.  A context keeps at most 16 arenas - the least recently used one is released (after synchronising ITS stream) when a
 * 17th stream arrives - and all of them until dpfhe_ctx_destroy unless told otherwise:
 *   dpfhe_ctx_release_scratch(ctx, stream, 0)                  releases the arena of `stream` (NULL = the default stream), after synchronising it - call it
 *                                                               before destroying a stream that ran composed operations, or when a phase is over;
 *   dpfhe_ctx_release_scratch(ctx, NULL, DPFHE_SCRATCH_ALL)    releases every arena of the context.
 * dpfhe_ctx_scratch_bytes: device bytes the context's arenas hold right now.  Thread-safe (one mutex per context). */
enum { DPFHE_SCRATCH_ALL = 1 };
int dpfhe_ctx_release_scratch(dpfhe_ctx* ctx, void* stream, uint32_t flags);
size_t dpfhe_ctx_scratch_bytes(const dpfhe_ctx* ctx);
uint32_t dpfhe_ctx_log2n(const dpfhe_ctx* ctx);
uint32_t dpfhe_ctx_limbs(const dpfhe_ctx* ctx);
/* 1 if every limb is of the form 2^60 - d, d < 2^24, and the fold-reduction kernels are in use */
int dpfhe_ctx_uses_fold(const dpfhe_ctx* ctx);
/* Which arithmetic the batched transforms and the fused multiply run limb `limb` on (round 6: chosen PER LIMB, not per context):
 *   DPFHE_ARITH_FOLD 2^60 - d, d < 2^24;  DPFHE_ARITH_F64 any prime below 2^47 (residues as IEEE doubles inside a transform, error-free FMA products);
 *   DPFHE_ARITH_FOLD_SCALED 2^k - d0 with 48 <= k < 60 and d0 2^(60-k) < 2^24 (carried scaled to 2^60 - d);  DPFHE_ARITH_F64_WIDE any other prime below 2^50
 *   (doubles again, with reductions inside the transforms);  DPFHE_ARITH_SHOUP every other prime (51 ... 59 bits far from a power of two: Harvey / Shoup
 *   butterflies, 128-bit Barrett products).  Results are the same words whatever the class.  A context of more than 16 limbs, or
 *   with log2_n > 14, is uniform: fold if every limb is, Shoup otherwise.  -1: null context or no such limb.
 * The reference's widest integer type is INT32 (src/core/hal/hal.hpp:27-33): its own parameter sets would land in DPFHE_ARITH_F64. */
enum { DPFHE_ARITH_SHOUP = 0, DPFHE_ARITH_FOLD = 1, DPFHE_ARITH_F64 = 2, DPFHE_ARITH_FOLD_SCALED = 3, DPFHE_ARITH_F64_WIDE = 4 };
int dpfhe_ctx_limb_class(const dpfhe_ctx* ctx, uint32_t limb);

/* -- A0, continued: which FORM of the fused multiply a context launches -------------------------------------------------
 * dpfhe_ct_mul(flags = 0) has two forms at N = 4096 / 8192 on fold-reduction contexts - "quad" (all four forward and all three inverse
 * transforms of a workgroup share twiddle fetches; the default at N = 4096) and "dual" (transforms in pairs; the default at N = 8192) -
 * with identical results and identical HBM traffic.
 * dpfhe_ctx_create takes the default (or the result an earlier dpfhe_ctx_autotune in this process found for the same device, log2_n and
 * n_limbs): it measures nothing, launches nothing and allocates nothing besides the tables.
 * dpfhe_ctx_autotune is the explicit opt-in: it times both forms on CALLER-provided scratch (work_words >= 7 L N; pairs = work_words /
 * (7 L N) synthetic ciphertext pairs; contents are overwritten; synchronises `stream`) and keeps the default unless the other form is
 * >= 3 % faster.  It changes the context: call it before the context is shared between threads.  Other contexts (generic primes, other
 * ring degrees) have one form; the call is a no-op there. */
/* DPFHE_TUNE_CACHED: a PROCESS-WIDE cache, keyed (device, log2_n, n_limbs), remembers what the last explicit dpfhe_ctx_autotune of that shape found;
 * contexts of the same shape created LATER in the process start from it (so behaviour depends on call order: both forms give the same words, only the
 * speed can differ).  dpfhe_tune_cache_clear() empties it.  (Value 1 was DPFHE_TUNE_AT_CREATE until round 4 and is retired: the cached state got a
 * fresh number so that an old caller never misreads it.) */
enum { DPFHE_TUNE_DEFAULT = 0, DPFHE_TUNE_EXPLICIT = 2, DPFHE_TUNE_FORCED = 3, DPFHE_TUNE_CACHED = 4 };
void dpfhe_tune_cache_clear(void);
typedef struct dpfhe_tune_info {
    int32_t chosen;        /* form in use (index for dpfhe_ct_mul_variant_name) */
    int32_t n_variants;    /* forms this context can run (0: one form, nothing measured) */
    int32_t source;        /* DPFHE_TUNE_* : how `chosen` was decided */
    uint32_t probe_pairs;  /* ciphertext pairs per probe launch */
    uint32_t probe_reps;   /* launches per form and pass */
    float probe_us[8];     /* best-pass microseconds per launch of each form (< 0: not measured) */
} dpfhe_tune_info;
int dpfhe_ctx_autotune(dpfhe_ctx* ctx, uint64_t* d_work, size_t work_words, uint32_t reps, void* stream);
int dpfhe_ctx_tune_info(const dpfhe_ctx* ctx, dpfhe_tune_info* out);
int dpfhe_ctx_set_ct_mul_variant(dpfhe_ctx* ctx, int variant);
const char* dpfhe_ct_mul_variant_name(int variant);

/* -- A1 / A2: batched negacyclic NTT, in place and out of place -------------------------------------- */
int dpfhe_ntt_fwd(dpfhe_ctx* ctx, uint64_t* d_io, size_t n_rns_polys, void* stream);
int dpfhe_ntt_inv(dpfhe_ctx* ctx, uint64_t* d_io, size_t n_rns_polys, void* stream);
int dpfhe_ntt_fwd_oop(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, void* stream);
int dpfhe_ntt_inv_oop(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, void* stream);

/* -- A3: coefficient-wise modular ops (either domain); out may alias a or b ----------------------------- */
int dpfhe_dyadic_mul(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n_rns_polys, void* stream);
int dpfhe_dyadic_mul_add(dpfhe_ctx* ctx, uint64_t* d_acc, const uint64_t* d_a, const uint64_t* d_b, size_t n_rns_polys, void* stream);
int dpfhe_add(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n_rns_polys, void* stream);
int dpfhe_sub(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n_rns_polys, void* stream);
int dpfhe_negate(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, size_t n_rns_polys, void* stream);
/* -- A7 (Evaluator::multiply_plain): d_out[i] = d_a[i] (.) d_pt for the n_rns_polys RNS polynomials of d_a (the components of a batch of
 *    ciphertexts, NTT domain) and ONE plaintext d_pt ([L][N], NTT domain).  One launch; the plaintext's L tiles are re-read from L2.
 *    d_out may alias d_a. */
int dpfhe_multiply_plain(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_pt, size_t n_rns_polys, void* stream);

/* -- A6: ciphertext x ciphertext tensor product, no relinearisation (THE METRIC OP) -----------------
 * d_a2, d_b2: [batch][2][L][N];  d_out3: [batch][3][L][N] = (a0 b0, a0 b1 + a1 b0, a1 b1) in R_q.
 * flags = 0: coefficient domain in and out (4 NTT + 4 dyadic mul + 1 add + 3 inverse NTT per limb, one
 * fused kernel: 7 residue polynomials of HBM traffic per limb).  d_a2 and d_b2 may be the same buffer (squaring); d_out3 must
 * not overlap either operand (DPFHE_INVALID_ARGUMENT). */
int dpfhe_ct_mul(dpfhe_ctx* ctx, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch,
                 uint32_t flags, void* stream);

/* diagnostics (tools/ctmul_trace.py): the "quad" form of dpfhe_ct_mul(flags = 0) with timestamps - thread 0 of workgroup w (= pair w / L, limb w % L)
 * writes d_trace[12 w .. 12 w + 11]: s_memrealtime (100 MHz) at start / first operand word arrived / forward transforms done / tensor
 * product done / inverse transforms done / stores issued / stores drained, then HW_ID | XCC_ID << 32, then prologue done / first operand's
 * loads issued / all loads issued / whole first operand arrived.  d_trace: batch * L * 12 words.  N = 4096 fold contexts only. */
int dpfhe_debug_ct_mul_trace(dpfhe_ctx* ctx, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch, uint64_t* d_trace, void* stream);

/* -- N1 (first "next" row): relinearisation 3 -> 2 components with RNS-digit evaluation keys ----
 * d_in3: [batch][3][L][N], d_out2: [batch][2][L][N], both coefficient domain.
 * d_evk: [L digits][2][L][N] in the NTT domain: evk_j = (-(a_j s) + e_j + g_j s^2, a_j) with g_j the CRT basis
 * element of limb j (1 mod q_j, 0 mod the others).  (c0', c1') = (c0, c1) + sum_j [c2]_{q_j} (.) evk_j. */
int dpfhe_relinearize(dpfhe_ctx* ctx, uint64_t* d_out2, const uint64_t* d_in3, const uint64_t* d_evk, size_t batch, void* stream);

/* -- N1, second half: rescale (exact RNS "divide by the last prime and round"), coefficient domain.
 * d_in: [n_rns_polys][L][N]  ->  d_out: [n_rns_polys][L-1][N] = round(x / q_{L-1}) limb by limb; requires L >= 2.
 * The result lives at the next level: use a context created with the first L-1 moduli for further work on it. */
int dpfhe_rescale(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, void* stream);

/* -- N1, hybrid key switching with ONE special prime P: ctx is the EXTENDED context whose last limb is P (L = Ld + 1
 * limbs); ciphertext data live on the first Ld limbs.  Keys: [Ld digits][2][L][N], NTT domain,
 *   key_j = (-(a_j s) + e_j + P g_j T, a_j)  over all L limbs (T = s^2 for relinearisation, sigma_g(s) for rotations).
 * d_work: caller-provided scratch of batch * 2 * L * N words.  Result (coefficient domain, Ld limbs):
 *   relinearize_hybrid : d_in3 [batch][3][Ld][N] -> d_out2 [batch][2][Ld][N] = (c0, c1) + round(sum_j [c2]_{q_j} key_j / P)
 *   switch_key_hybrid  : d_in2 [batch][2][Ld][N] -> d_out2 = (c0, 0) + round(sum_j [c1]_{q_j} key_j / P)
 * The noise added is ~ Ld N q sigma / P instead of ~ Ld N q sigma. */
int dpfhe_relinearize_hybrid(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in3, const uint64_t* d_key, uint64_t* d_work,
                             size_t batch, void* stream);
int dpfhe_switch_key_hybrid(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in2, const uint64_t* d_key, uint64_t* d_work,
                            size_t batch, void* stream);

/* -- N3, batched rotations (the baby / giant steps of a packed matrix-vector product): item i of the output is the
 * hybrid-key-switched sigma_{galois_elts[i]} of input item i (n_in == batch) or of THE input item (n_in == 1).
 * galois_elts: HOST array of `batch` odd elements < 2N.  d_keys: `batch` keys back to back, each [Ld][2][L][N] as for
 * dpfhe_switch_key_hybrid, key i for element i.  d_work: batch * 2 * L * N words; d_rotated: batch * 2 * Ld * N words
 * (scratch, must not alias the input).  One automorphism launch per 64 items + one key-switch pass for all of them. */
int dpfhe_rotate_hybrid_batch(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in2, size_t n_in, const uint32_t* galois_elts,
                              const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated, size_t batch, void* stream);

/* -- N3, HOISTED rotations: `batch` rotations of ONE ciphertext (d_in2: [2][Ld][N], coefficient domain) sharing the digit
 * decomposition: the Ld digits of c1 are lifted to the L limbs and transformed once (d_digits: Ld * L * N words of scratch), every
 * rotation is then a permutation of those words in the NTT domain, its key inner product and two inverse transforms per limb
 * (instead of Ld + 2), followed by the divide-by-P pass.  Keys and d_work as for dpfhe_rotate_hybrid_batch; d_rotated0:
 * batch * Ld * N words of scratch (sigma_g(c0)).  The result is a valid key switch of sigma_g(ct) but NOT word-identical to
 * dpfhe_rotate_hybrid_batch: here the automorphism acts on the lifted digits (sigma_g after the lift), there on c1 before it. */
int dpfhe_rotate_hybrid_hoisted(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts,
                                const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated0, uint64_t* d_digits, size_t batch, void* stream);
/*    n_items input ciphertexts (tokens) share the `batch` rotations and their keys: d_in2 is [n_items][2][Ld][N], the output is
 *    ROTATION-MAJOR, item r * n_items + t = sigma_{galois_elts[r]}(input t); scratch sizes scale with n_items (d_work, d_rotated0:
 *    batch * n_items items; d_digits: n_items * Ld * L * N words).  Workgroups that read the same key tile run next to each other. */

/* -- N3, grouped rotations (giant steps of several tokens): batch = n_elts * group items, item i rotated by galois_elts[i / group]
 *    with key i / group; otherwise as dpfhe_rotate_hybrid_batch with n_in == batch. */
int dpfhe_rotate_hybrid_grouped(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in2, const uint32_t* galois_elts, size_t n_elts, size_t group,
                                const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated, void* stream);

/* -- N3, round 3: baby-step / giant-step sums with the division by P DEFERRED ("double hoisting", Bossuat-Mouchet-Troncoso-Pastoriza-
 * Hubaux 2021).  A rotated term is kept in the NTT domain over the extended basis Q P as  P sigma_g(ct) + key-switching noise;
 * plaintext products and sums are taken there, and one inverse transform + divide-by-P is paid per SUM, not per rotation.
 * Stages (ctx_ext = data moduli + special prime P as its last limb; QP buffers are [..][2][L][N], data buffers [..][2][Ld][N]):
 *
 * dpfhe_rotate_hoisted_qp: n_items coefficient-domain inputs d_in2 [n_items][2][Ld][N] -> d_out_qp [(1 + batch)][n_items][2][L][N], NTT
 *   domain (forward-output order), canonical:  block 0 = (P c0, P c1) (the input itself);  block 1 + r, r < batch =
 *     ( sum_j NTT(sigma_g lift([c1]_{q_j})) (.) key_{g,j,0} + P NTT(sigma_g c0),  sum_j NTT(sigma_g lift([c1]_{q_j})) (.) key_{g,j,1} ),  g = galois_elts[r]
 *   (the P c0 term is 0 on the special limb).  Keys as for dpfhe_rotate_hybrid_hoisted.  Scratch: d_in_ntt n_items * 2 * Ld * N words,
 *   d_digits n_items * Ld * L * N words.  round(block / P) after an inverse transform equals dpfhe_rotate_hybrid_hoisted's output.
 * dpfhe_ntt_inv_galois (any context): block e of rns_polys_per_elt RNS polynomials:  d_out = sigma_{galois_elts[e]}(INTT(d_in)), the
 *   automorphism applied as a gather in the NTT domain (no separate pass).  d_out == d_in allowed.  log2_n <= 14.
 * dpfhe_switch_key_qp: batch = n_keys * group coefficient-domain items d_in2 [batch][2][Ld][N], item i with key i / group ->
 *   d_out_qp [batch][2][L][N] = sum_j NTT(lift([c1]_{q_j})) (.) key_j, NTT domain, nothing added, no division: the giant steps'
 *   terms, to be summed (dpfhe_reduce_sum on ctx_ext), inverse-transformed once and finished by
 * dpfhe_rescale_bsgs: d_in_qp [batch][2][L][N] (coefficient domain) -> d_out2 [batch][2][Ld][N] = round(in / P) + addends, where
 *   d_addends is [n_add][batch][2][Ld][N]: component 0 adds component 0 of ALL n_add items, component 1 adds component 1 of item 0
 *   only (the rotated inner sums of a baby-step / giant-step product: item 0 is not rotated; the c1 of the others went into
 *   dpfhe_switch_key_qp). */
int dpfhe_rotate_hoisted_qp(dpfhe_ctx* ctx_ext, uint64_t* d_out_qp, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts,
                            const uint64_t* d_keys, uint64_t* d_in_ntt, uint64_t* d_digits, size_t batch, void* stream);
int dpfhe_ntt_inv_galois(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t rns_polys_per_elt, const uint32_t* galois_elts, size_t n_elts,
                         void* stream);
int dpfhe_switch_key_qp(dpfhe_ctx* ctx_ext, uint64_t* d_out_qp, const uint64_t* d_in2, const uint64_t* d_keys, size_t n_keys, size_t group, void* stream);
int dpfhe_rescale_bsgs(dpfhe_ctx* ctx_ext, uint64_t* d_out2, const uint64_t* d_in_qp, const uint64_t* d_addends, size_t n_add, size_t batch,
                       void* stream);

/* -- round 4 (what an EXACT ciphertext x ciphertext multiply needs around dpfhe_ct_mul): limb ranges of one context ------------
 * dpfhe_base_extend: per coefficient, the integer X in (-Qs/2, Qs/2] whose residues modulo the source limbs [src_limb0, src_limb0 + n_src)
 *   are given (Qs their product, n_src <= 10) is reduced modulo the destination limbs [dst_limb0, dst_limb0 + n_dst) (n_dst <= 20; the ranges may
 *   overlap - a destination limb that is also a source limb gets its own residue back).  Exact (mixed-radix reconstruction), not approximate.
 *   d_in: item p's source residues at d_in + (p * in_stride_limbs + i) * N, i < n_src;  d_out: item p's results at d_out + (p * out_stride_limbs + j) * N.
 * dpfhe_scale_round: d_in [n_polys][L][N] holds ALL limbs of the context;  d_out (item stride out_stride_limbs) receives, on the kept limbs
 *   [keep_limb0, keep_limb0 + n_keep),  round(multiplier * X / Qd)  where X is the (centred) integer the L limbs represent and Qd the product of the
 *   dropped limbs [drop_limb0, drop_limb0 + n_drop) (n_drop <= 10, disjoint from the kept ones) - exact as long as |multiplier * X| < Q / 2.
 *   With multiplier = the plaintext modulus t and Qd = the operands' ciphertext modulus this is the scale-and-round of a BFV-style multiply:
 *   extend both operands to the whole context, dpfhe_ct_mul there, dpfhe_scale_round, dpfhe_base_extend back (Evaluator::multiply_exact). */
int dpfhe_base_extend(dpfhe_ctx* ctx, uint64_t* d_out, size_t out_stride_limbs, const uint64_t* d_in, size_t in_stride_limbs, uint32_t src_limb0, uint32_t n_src,
                      uint32_t dst_limb0, uint32_t n_dst, size_t n_polys, void* stream);
int dpfhe_scale_round(dpfhe_ctx* ctx, uint64_t* d_out, size_t out_stride_limbs, const uint64_t* d_in, uint32_t drop_limb0, uint32_t n_drop, uint32_t keep_limb0,
                      uint32_t n_keep, uint64_t multiplier, size_t n_polys, void* stream);

/* -- N3: Galois automorphism a(X) -> a(X^galois_elt) (galois_elt odd, < 2N), coefficient domain, d_out != d_in;
 *        and the key switch that follows it:  (c0', c1') = (c0 + sum_j [c1]_{q_j} (.) key_j[0], sum_j [c1]_{q_j} (.) key_j[1]),
 *        key_j = (-(a_j s) + e_j + g_j sigma(s), a_j) in the NTT domain, layout [L][2][L][N] like the relinearisation keys. */
int dpfhe_apply_galois(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, uint32_t galois_elt, void* stream);
int dpfhe_switch_key(dpfhe_ctx* ctx, uint64_t* d_out2, const uint64_t* d_in2, const uint64_t* d_key, size_t batch, void* stream);

/* -- A7: ciphertext x plaintext matrix-vector product, everything in the NTT domain ------------------
 * d_W: [rows][cols][L][N] plaintext polys; d_x: [cols][2][L][N]; d_y: [rows][2][L][N],
 * y_i = sum_j W_ij (.) x_j  with 128-bit lazy accumulation and one reduction at the end. */
int dpfhe_matvec_plain(dpfhe_ctx* ctx, uint64_t* d_y, const uint64_t* d_W, const uint64_t* d_x, size_t rows,
                       size_t cols, void* stream);

/* -- A7 with n_rhs right-hand sides (tokens): x is [cols][n_rhs][2][L][N], y is [rows][n_rhs][2][L][N] (the token index sits between
 *    the column / row and the component); y[i][t] = sum_j W[i][j] (.) x[j][t].  W is streamed from HBM once: 2 right-hand sides (4 polynomials) x 4 rows per workgroup, the groups of a W tile adjacent on one XCD. */
int dpfhe_matvec_plain_multi(dpfhe_ctx* ctx, uint64_t* d_y, const uint64_t* d_W, const uint64_t* d_x, size_t rows, size_t cols, size_t n_rhs,
                             void* stream);

/* scalar-weight variant (the realistic plaintext linear layer: W_ij in Z_q, given as one residue per limb):
 * d_w: [rows][cols][L] words;  y_i = sum_j w_ij * x_j.  Works in either domain (scalars commute with the NTT). */
int dpfhe_matvec_scalar(dpfhe_ctx* ctx, uint64_t* d_y, const uint64_t* d_w, const uint64_t* d_x, size_t rows,
                        size_t cols, void* stream);

/* -- A8: modular sum of `count` ciphertexts of `components` RNS polys each into one ------------------
 * d_in: [count][components][L][N] -> d_out: [components][L][N].  (shard-local reduce before the all-gather) */
int dpfhe_reduce_sum(dpfhe_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, size_t count, size_t components, void* stream);

/* -- measurement aid (SURVEY.md 8(d) "also report a measured device-copy bandwidth as the practical ceiling"): a plain device-to-device
 * copy of n_words (even) u64 words with the access shape of the library's streaming kernels (16 bytes per lane, eight loads in
 * flight per thread).  bench.py reads the transforms' HBM rates against it. */
int dpfhe_copy(dpfhe_ctx* ctx, uint64_t* d_dst, const uint64_t* d_src, size_t n_words, void* stream);

/* -- (e): the one collective - RCCL all-gather of one partial ciphertext per rank over xGMI ----------
 * One process per GPU.  Rank 0 calls dpfhe_comm_unique_id, ships the 128 bytes to the other ranks by any
 * means (the host program's own rendezvous), every rank calls dpfhe_comm_create.  d_recv holds
 * world_size * words_per_rank words; rank r's contribution lands at offset r * words_per_rank. */
int dpfhe_comm_unique_id(uint8_t out_id[128]);
int dpfhe_comm_create(dpfhe_comm** out, const uint8_t id[128], int rank, int world_size, int device_id);
int dpfhe_comm_destroy(dpfhe_comm* comm);
int dpfhe_comm_allgather(dpfhe_comm* comm, uint64_t* d_recv, const uint64_t* d_send, size_t words_per_rank, void* stream);
/* (SURVEY.md section 8(b) lists this exchange as `dpfhe_allgather_partials`; it is exported under the dpfhe_comm_* names above, with the communicator's
 *  life cycle next to it.)
 * The alternative SURVEY.md section 8(e) asks to have measured: ncclAllReduce(ncclUint64, ncclSum) of the ranks' partial ciphertexts IN PLACE, then one
 * mod-q pass (dpfhe_canonicalize_sum) - world_size <= 15, since 15 q < 2^64 for q < 2^60.  The result is what all-gather + the local sum give.
 * dpfhe_canonicalize_sum alone: words that are sums of at most 15 canonical residues -> canonical residues, in place (the pass after a
 * torch.distributed all_reduce of the partials). */
int dpfhe_comm_allreduce_sum(dpfhe_comm* comm, dpfhe_ctx* ctx, uint64_t* d_io, size_t n_rns_polys, void* stream);
int dpfhe_canonicalize_sum(dpfhe_ctx* ctx, uint64_t* d_io, size_t n_rns_polys, void* stream);

const char* dpfhe_strerror(int code);
/* text of the last HIP/RCCL failure on the calling thread ("" if none) */
const char* dpfhe_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DPFHE_H */
