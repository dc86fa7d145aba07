// launch_impl.h - geometry dispatch shared by the per-policy translation units.
#pragma once
#include <cstdlib>

#include "kernels.h"
#include "launch.h"
#include "kernels_halves.h"
#include "kernels_quarters.h"
#ifdef DPFHE_DIAGNOSTICS   // diagnostic builds only (tools/ab_variant.sh diag -DDPFHE_DIAGNOSTICS): per-workgroup timestamp kernels, their buffers and read-back entries
#include "kernels_trace.h"
#endif

namespace dpfhe {

// gfx950 ONLY: the paired / quad fused kernels (ct_mul_dual_kernel, ct_mul_quad_kernel, relin_shared_kernel, hoisted_ks2_kernel) hold
// TWO padded LDS images of a polynomial per workgroup - 76 KiB at N = 4096, 152 KiB at N = 8192 of CDNA4's 160 KiB per CU (gfx942
// has 64 KiB: this library does not build for it, by design - no dual paths).  A padding change that breaks the budget fails here,
// not at launch time.
constexpr size_t kLdsBytesPerCu = 160 * 1024;
static_assert(2 * sizeof(u64) * Geo<13, kFusedLoge>::lds_words() <= kLdsBytesPerCu, "two LDS images of an N = 8192 polynomial must fit one CU's 160 KiB");
// ... and today they fill it to the byte (2 x 81 920 B = 163 840 B: ct_mul_dual_kernel / hoisted_ks2_kernel at N = 8192 leave NO room for another
// __shared__ word or a wider wave stride).  Pinned on purpose: whoever changes Geo's padding sees the exact figure change here, not a launch failure.
static_assert(2 * sizeof(u64) * Geo<13, kFusedLoge>::lds_words() == 163840 && 2 * sizeof(u64) * Geo<12, kFusedLoge>::lds_words() == 77824,
              "LDS footprint of the paired fused kernels changed: re-check the 160 KiB budget at N = 8192 (zero margin) and two workgroups per CU at N = 4096");
static_assert(sizeof(u64) * Geo<14, 4>::lds_words() <= kLdsBytesPerCu, "the N = 16384 transform's LDS image must fit one CU's 160 KiB");

// (log2n -> LOGE) pairs proven by tests/test_emulated_kernels.py
#define DPFHE_GEO_SWITCH(log2n, MACRO) \
    switch (log2n) {                   \
        case 8: MACRO(8, 4); break;    \
        case 9: MACRO(9, 4); break;    \
        case 10: MACRO(10, 4); break;  \
        case 11: MACRO(11, 4); break;  \
        case 12: MACRO(12, 4); break;  \
        case 13: MACRO(13, 4); break;  \
        default: return -1;            \
    }
// the batched NTT kernels additionally cover N = 16384
#define DPFHE_NTT_GEO_SWITCH(log2n, MACRO)  \
    switch (log2n) {                        \
        case 8: MACRO(8, 4); break;         \
        case 9: MACRO(9, 4); break;         \
        case 10: MACRO(10, 4); break;       \
        case 11: MACRO(11, 4); break;       \
        case 12: MACRO(12, 4); break;       \
        case 13: MACRO(13, 4); break;       \
        case 14: MACRO(14, 4); break;       \
        default: return -1;                 \
    }
static_assert(ntt_loge(12) == 4 && ntt_loge(13) == 4 && ntt_loge(8) == 4 && ntt_loge(14) == 4, "launch.h ntt_loge must match DPFHE_NTT_GEO_SWITCH");

template <class Arith, int LOG_N1>
static void launch_ntt_split(bool inverse, u64* out, const u64* in, size_t npolys, const DevTables<Arith>& tb, hipStream_t s) {
    constexpr int LN2 = kSplitLog2N2;
    const unsigned top_grid = (unsigned)(npolys * ((size_t)1 << LN2) / 256), sub_grid = (unsigned)(npolys << LOG_N1);
    if (!inverse) {
        hipLaunchKernelGGL((ntt_top_kernel<Arith, LOG_N1, true>), dim3(top_grid), dim3(256), 0, s, out, in, tb, LN2);
        hipLaunchKernelGGL((ntt_fwd_kernel<Arith, LN2, 4>), dim3(sub_grid), dim3(Geo<LN2, 4>::T), 0, s, out, out, tb);
    } else {
        hipLaunchKernelGGL((ntt_inv_kernel<Arith, LN2, 4>), dim3(sub_grid), dim3(Geo<LN2, 4>::T), 0, s, out, in, tb);
        hipLaunchKernelGGL((ntt_top_kernel<Arith, LOG_N1, false>), dim3(top_grid), dim3(256), 0, s, out, out, tb, LN2);
    }
}

template <class Arith>
int launch_ntt(int log2n, bool inverse, u64* out, const u64* in, size_t npolys, const DevTables<Arith>& tb, hipStream_t s) {
    constexpr bool kClassPolicy = Arith::kF64 || (Arith::kFoldCore && !Arith::kFold);   // round 6's per-limb classes: single-kernel transforms only
    if constexpr (!kClassPolicy) {
        if (log2n == 15) { launch_ntt_split<Arith, 3>(inverse, out, in, npolys, tb, s); return 0; }
        if (log2n == 16) { launch_ntt_split<Arith, 4>(inverse, out, in, npolys, tb, s); return 0; }
    }
    // batches whose input + output cannot stay in the 256 MiB Infinity Cache stream around it (FoldArith, the two production ring degrees)
    const size_t touched = (npolys << log2n) * sizeof(u64) * (out == in ? 1 : 2);
    const bool nt = Arith::kFold && (log2n == 12 || log2n == 13) && touched > ((size_t)256 << 20);
#ifdef DPFHE_DIAGNOSTICS   // the forward transform at N = 4096 / 8192 with per-workgroup timestamps (kernels_trace.h, tools/ntt_trace.py)
    if constexpr (Arith::kFold) {
        if (!inverse && (log2n == 12 || log2n == 13) && npolys <= 65536 && !tb.n_active) {
            if (!g_ntt_trace) { if (hipMalloc(&g_ntt_trace, sizeof(u64) * 8 * 65536) != hipSuccess) return -1; }
            g_ntt_trace_blocks = (unsigned)npolys;
            if (log2n == 13 && tb.hfwd && npolys >= kHalvesMinPolys) { hipLaunchKernelGGL((ntt_fwd_halves_trace_kernel<Arith>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb, g_ntt_trace); return 0; }
            if (log2n == 12) hipLaunchKernelGGL((ntt_fwd_trace_kernel<Arith, 12, 4>), dim3((unsigned)npolys), dim3(Geo<12, 4>::T), 0, s, out, in, tb, g_ntt_trace);
            else hipLaunchKernelGGL((ntt_fwd_trace_kernel<Arith, 13, 4>), dim3((unsigned)npolys), dim3(Geo<13, 4>::T), 0, s, out, in, tb, g_ntt_trace);
            return 0;
        }
    }
#endif
    // N = 16384, FoldArith: 256-thread workgroups on the N = 4096 body, two to a CU (ntt_quarters.h)
    if constexpr (Arith::kFold) {
        if (log2n == 14 && tb.qfwd && !tb.n_active && npolys >= kQuartersMinPolys) {
            const bool nt14 = touched > ((size_t)256 << 20);
            if (inverse) {
                if (nt14) hipLaunchKernelGGL((ntt_inv_quarters_kernel<true>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
                else hipLaunchKernelGGL((ntt_inv_quarters_kernel<false>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
            } else {
                if (nt14) hipLaunchKernelGGL((ntt_fwd_quarters_kernel<true>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
                else hipLaunchKernelGGL((ntt_fwd_quarters_kernel<false>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
            }
            return 0;
        }
    }
    // N = 8192, large batches, FoldArith: 256-thread workgroups on the N = 4096 body (launch.h)
    if constexpr (Arith::kFold) {
        if (log2n == 13 && tb.hfwd && !tb.n_active && npolys >= kHalvesMinPolys) {
            if (inverse) {
                if (nt) hipLaunchKernelGGL((ntt_inv_halves_kernel<Arith, true>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
                else hipLaunchKernelGGL((ntt_inv_halves_kernel<Arith, false>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
            } else {
                if (nt) hipLaunchKernelGGL((ntt_fwd_halves_kernel<Arith, true>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
                else hipLaunchKernelGGL((ntt_fwd_halves_kernel<Arith, false>), dim3((unsigned)npolys), dim3(256), 0, s, out, in, tb);
            }
            return 0;
        }
    }
#define NTT_CASE(LN, LE)                                                                                                              \
    if constexpr (Arith::kFold && (LN == 12 || LN == 13)) {                                                                           \
        if (nt) {                                                                                                                     \
            if (inverse) hipLaunchKernelGGL((ntt_inv_kernel<Arith, LN, LE, true>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb); \
            else hipLaunchKernelGGL((ntt_fwd_kernel<Arith, LN, LE, true>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb);         \
            break;                                                                                                                    \
        }                                                                                                                             \
    }                                                                                                                                 \
    if (inverse) hipLaunchKernelGGL((ntt_inv_kernel<Arith, LN, LE>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb); \
    else hipLaunchKernelGGL((ntt_fwd_kernel<Arith, LN, LE>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb)
    DPFHE_NTT_GEO_SWITCH(log2n, NTT_CASE)
#undef NTT_CASE
    return 0;
}

template <class Arith, bool IN_NTT, bool OUT_NTT>
static int launch_ct_mul_dom(int log2n, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, hipStream_t s) {
    // the fused kernel keeps up to four transformed polynomials in registers: always E = 16 words per thread.
    // Coefficient domain in and out, N <= 4096: all four forward and all three inverse transforms share their twiddle fetches
    // (ct_mul_quad_kernel); N = 8192 or NTT-domain output: transforms in pairs (ct_mul_dual_kernel; at N = 8192 the quad form
    // measured equal to slightly slower: one 8-wave workgroup per CU, three barriers per all-to-all exchange - profiles/r03_ab_quad13.txt)
    constexpr int kQuadMaxLogN = 12, kDualMaxLogN = 13;
#define CT_CASE(LN, LE)                                                                                                                              \
    if constexpr (Arith::kFoldCore && !IN_NTT && !OUT_NTT && LN <= kQuadMaxLogN)   /* (F64Arith's quad form spills: pairs) */                                           \
        hipLaunchKernelGGL((ct_mul_quad_kernel<Arith, LN, kFusedLoge>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb, (u64*)nullptr); \
    else if constexpr (Arith::kFold && !IN_NTT && OUT_NTT && LN <= kQuadMaxLogN)   /* round 6: four shared forward transforms + the lazy tensor step, products left in the NTT domain */ \
        hipLaunchKernelGGL((ct_mul_quad_kernel<Arith, LN, kFusedLoge, false, true>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb, (u64*)nullptr); \
    else if constexpr ((Arith::kFoldCore || Arith::kF64) && !IN_NTT && LN <= kDualMaxLogN)                                                                                                     \
        hipLaunchKernelGGL((ct_mul_dual_kernel<Arith, LN, kFusedLoge, OUT_NTT>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb); \
    else                                                                                                                                             \
        hipLaunchKernelGGL((ct_mul_kernel<Arith, LN, kFusedLoge, IN_NTT, OUT_NTT>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb)
    DPFHE_GEO_SWITCH(log2n, CT_CASE)
#undef CT_CASE
    return 0;
}

template <class Arith>
int launch_ct_mul(int log2n, unsigned flags, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, hipStream_t s) {
    switch (flags & 3u) {
        case 0: return launch_ct_mul_dom<Arith, false, false>(log2n, out3, a2, b2, blocks, tb, s);
        case 1: return launch_ct_mul_dom<Arith, true, false>(log2n, out3, a2, b2, blocks, tb, s);
        case 2: return launch_ct_mul_dom<Arith, false, true>(log2n, out3, a2, b2, blocks, tb, s);
        default: return launch_ct_mul_dom<Arith, true, true>(log2n, out3, a2, b2, blocks, tb, s);
    }
}

// The fused multiply in a NAMED form (coefficient domain in and out, FoldArith, N = 4096 / 8192): what dpfhe_ctx_autotune probes and
// what a context then launches.  kCtMulQuad / kCtMulDual differ in how many transforms share twiddle fetches and LDS
// buffers (kernels.h), not in results (bit-identical) or HBM traffic (7 residue polynomials per limb).  -1: not compiled for this ring.
template <class Arith>
int launch_ct_mul_variant(int log2n, int variant, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, hipStream_t s) {
    if constexpr (!Arith::kFold) return -1;
    else {
#define CTV_CASE(LN)                                                                                                                                  \
    case LN:                                                                                                                                          \
        if (variant == kCtMulQuad) hipLaunchKernelGGL((ct_mul_quad_kernel<Arith, LN, kFusedLoge>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb, (u64*)nullptr); \
        else if (variant == kCtMulDual) hipLaunchKernelGGL((ct_mul_dual_kernel<Arith, LN, kFusedLoge, false>), dim3((unsigned)blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, out3, a2, b2, tb); \
        else return -1;                                                                                                                               \
        return 0
        switch (log2n) {
            CTV_CASE(12);
            CTV_CASE(13);
            default: return -1;
        }
#undef CTV_CASE
    }
}
// diagnostics: the quad form with per-workgroup timestamps (kernels.h trace_stamp); N = 4096 only
template <class Arith>
int launch_ct_mul_trace(int log2n, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, u64* trace, hipStream_t s) {
    if constexpr (!Arith::kFold) return -1;
    else {
        if (log2n != 12) return -1;
        hipLaunchKernelGGL((ct_mul_quad_kernel<Arith, 12, kFusedLoge, true>), dim3((unsigned)blocks), dim3(Geo<12, kFusedLoge>::T), 0, s, out3, a2, b2, tb, trace);
        return 0;
    }
}

#ifdef DPFHE_DIAGNOSTICS
inline u64* g_relin_trace = nullptr;
inline unsigned g_relin_trace_blocks = 0;
#endif
template <class Arith>
int launch_relin(int log2n, int mode, u64* out2, const u64* in3, const u64* evk, size_t key_stride, unsigned key_group, size_t blocks,
                 const DevTables<Arith>& tb, hipStream_t s) {
    // 4..7 digits at N <= 4096: digit transforms side by side (relin_shared_kernel: -10 % on relinearize at N=4096, L=4; at N=8192,
    // where one workgroup owns the CU and the key tiles are what it streams, the same form measured -1 %: not instantiated)
    if (mode < 0 || mode > 4) return -1;
    const unsigned La = tb.n_active ? (unsigned)tb.n_active : (unsigned)tb.n_limbs;   // limbs this launch works on (one class of a mixed context, or all): blocks = items x La
    const int n_digits = mode >= 2 ? tb.n_limbs - 1 : tb.n_limbs;
    // items that share a key (key_group > 1) are laid out per XCD: kernels.h relin_kernel
    const unsigned kg = key_group ? key_group : 1u;
    // (round 6: items with a key each and NO sharing - one token's rotations - take the key-major order too: all limbs of an item on one XCD, so that its
    //  digits cross the fabric once, not once per XCD - profiles/r06_giant_traffic.txt, shape (15, 1).  Items that share ONE key (key_stride 0) keep the plain
    //  order, in which an XCD only ever touches the key tiles of its own limbs.)
    unsigned n_outer = ((kg > 1 || key_stride != 0) && blocks % ((size_t)kg * La) == 0) ? (unsigned)(blocks / kg) : 0u;   // whole groups only
    unsigned grid = n_outer ? ((n_outer + 7u) / 8u) * 8u * kg : (unsigned)blocks;
    // eight keys or more: one key (all its limbs and items) per XCD at a time, so that the items' digits are fetched once, not once per limb
    if (n_outer && n_outer / La >= 8u) {
        const unsigned n_keys = n_outer / La;
        grid = ((n_keys + 7u) / 8u) * 8u * La * kg;
        n_outer |= kRelinRotMajor;
    }
#ifdef DPFHE_DIAGNOSTICS   // (tools/relin_trace.py reads the buffer back)
    if constexpr (Arith::kFold) {
        if (log2n == 13 && mode == 4) {
            if (!g_relin_trace) { if (hipMalloc(&g_relin_trace, sizeof(u64) * 8 * 65536) != hipSuccess) return -1; }
            g_relin_trace_blocks = grid;
            hipLaunchKernelGGL((relin_kernel<Arith, 13, kFusedLoge, 4, true>), dim3(grid), dim3(Geo<13, kFusedLoge>::T), 0, s, out2, in3, evk, key_stride, kg, n_outer, tb, g_relin_trace);
            return 0;
        }
    }
#endif
#define RL_ONE(LN, M)                                                                                                                                    \
    if constexpr (Arith::kFold && LN >= 10 && LN <= 12) {                                                                          \
        if (n_digits >= 4 && n_digits <= 7 && !tb.n_active) {                                                                                                            \
            hipLaunchKernelGGL((relin_shared_kernel<Arith, LN, kFusedLoge, M>), dim3(grid), dim3(Geo<LN, kFusedLoge>::T), 0, s, out2, in3, evk,          \
                               key_stride, kg, n_outer, tb);                                                                                             \
            break;                                                                                                                                       \
        }                                                                                                                                                \
    }                                                                                                                                                    \
    hipLaunchKernelGGL((relin_kernel<Arith, LN, kFusedLoge, M>), dim3(grid), dim3(Geo<LN, kFusedLoge>::T), 0, s, out2, in3, evk, key_stride, kg, n_outer, tb)
#define RL_CASE(LN, LE)                \
    if (mode == 0) { RL_ONE(LN, 0); }      \
    else if (mode == 1) { RL_ONE(LN, 1); } \
    else if (mode == 2) { RL_ONE(LN, 2); } \
    else if (mode == 3) { RL_ONE(LN, 3); } \
    else { RL_ONE(LN, 4); }
    DPFHE_GEO_SWITCH(log2n, RL_CASE)
#undef RL_CASE
#undef RL_ONE
    return 0;
}

template <class Arith>
int launch_hoisted_ks(int log2n, u64* work, const u64* digits, const u64* keys, size_t key_stride, const unsigned* elts, size_t count,
                      size_t n_items, const DevTables<Arith>& tb, hipStream_t s) {
    if (count > (size_t)kMaxGaloisBatch) return -1;   // the elements travel as kernel arguments: callers chunk by kMaxGaloisBatch
    GaloisElts ge{};
    for (size_t i = 0; i < count; ++i) ge.v[i] = elts[i];
    // few workgroups (one token): one per (rotation, limb, key component), half the serial chain each; many (several tokens, Ld <= 7 so
    // that the lazy sums fit): one per (rotation, limb) doing both components - the digit words are gathered once and the two inverse
    // transforms share their twiddles (kernels.h hoisted_ks2_kernel)
    constexpr size_t kHoistedMergeMin = 512;
    const unsigned La = tb.n_active ? (unsigned)tb.n_active : (unsigned)tb.n_limbs;  // limbs this launch works on (one class of a mixed context, or all)
    const unsigned tiles1 = (unsigned)(count * (size_t)La);                         // (rotation, limb)
    const bool merged = Arith::kFold && !tb.n_active && tb.n_limbs - 1 <= 7 && (size_t)tiles1 * n_items >= kHoistedMergeMin;
    const unsigned tiles = merged ? tiles1 : tiles1 * 2u;                           // ... x key component when split
    const unsigned blocks = ((tiles + 7u) / 8u) * 8u * (unsigned)n_items;           // x token, ids laid out per XCD (kernels.h)
#define HK_CASE(LN, LE)                                                                                                                                  \
    if constexpr (Arith::kFold) {                                                                                                                        \
        if (merged) {                                                                                                                                    \
            hipLaunchKernelGGL((hoisted_ks2_kernel<Arith, LN, kFusedLoge>), dim3(blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, work, digits, keys, key_stride, ge, \
                               (unsigned)n_items, tiles, tb);                                                                                            \
            break;                                                                                                                                       \
        }                                                                                                                                                \
    }                                                                                                                                                    \
    hipLaunchKernelGGL((hoisted_ks_kernel<Arith, LN, kFusedLoge>), dim3(blocks), dim3(Geo<LN, kFusedLoge>::T), 0, s, work, digits, keys, key_stride, ge, (unsigned)n_items, tiles, tb)
    DPFHE_GEO_SWITCH(log2n, HK_CASE)
#undef HK_CASE
    return 0;
}

template <class Arith>
int launch_ntt_inv_galois(int log2n, u64* out, const u64* in, const unsigned* elts, size_t n_elts, size_t polys_per_elt, const DevTables<Arith>& tb, hipStream_t s) {
    if (tb.n_sub != 1 || n_elts > (size_t)kMaxGaloisBatch) return -1;   // split transforms (N > 16384) have no gather form; callers chunk by kMaxGaloisBatch
    GaloisElts ge{};
    for (size_t i = 0; i < n_elts; ++i) ge.v[i] = elts[i];
    // polys_per_elt counts residue polynomials over ALL limbs; a class launch covers n_active of every n_limbs of them
    if (tb.n_active) polys_per_elt = polys_per_elt / (size_t)tb.n_limbs * (size_t)tb.n_active;
    const unsigned grid = (unsigned)(n_elts * polys_per_elt);
#define NG_CASE(LN, LE) \
    hipLaunchKernelGGL((ntt_inv_galois_kernel<Arith, LN, LE>), dim3(grid), dim3(Geo<LN, LE>::T), 0, s, out, in, ge, (unsigned)polys_per_elt, tb)
    DPFHE_NTT_GEO_SWITCH(log2n, NG_CASE)
#undef NG_CASE
    return 0;
}

}  // namespace dpfhe
