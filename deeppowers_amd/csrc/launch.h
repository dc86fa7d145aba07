// launch.h - host launchers of the templated kernels; each arithmetic policy is instantiated in its own
// translation unit (k_ntt_*.hip, k_ctmul_*.hip) so the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "devtables.h"

namespace dpfhe {

// words-per-thread exponent of the batched NTT kernels (launch_impl.h DPFHE_GEO_SWITCH); the fused kernels use 4.
// dpfhe_ctx_create builds a second table layout whenever the two differ.
// (N = 8192 was measured with 32 words per thread / 3 phases and with 16 / 4 phases: same time, the kernels are VALU-bound;
// 16 everywhere keeps one twiddle layout per context)
// N = 16384 (128 KiB of LDS per polynomial): 1024 threads, one workgroup per CU (16 words per thread measured 5 % faster
// than 32 on the forward transform); the fused kernels stop at N = 8192.
constexpr int ntt_loge(int /*log2n*/) { return 4; }   // (32 words per thread at N = 4096 measured equal to slower - round 4 A/B, closed)
constexpr int kMaxLog2N = 16, kMaxFusedLog2N = 13;
// N > 16384: split transform - log2(N1) top stages in ntt_top_kernel, then N1 transforms of N2 = 4096 points each
constexpr int kSplitLog2N2 = 12;
constexpr int split_log_n1(int log2n) { return log2n > 14 ? log2n - kSplitLog2N2 : 0; }
constexpr int kFusedLoge = 4;   // words-per-thread exponent of the fused kernels (8 per thread measured slower - round 3 A/B, closed)
// N = 8192 on the N = 4096 body ("halves": ntt_halves.h, kernels_halves.h - a register column stage + two 4096-point sub-transforms through one LDS
// buffer, 256-thread workgroups, three to a CU).  Round 5, measured (profiles/r05_ntt13_batch_sweep.txt, r05_ntt_workgroup_timelines.txt, r05_halves_*.txt):
//  * batched transforms: 2.5-9 % FASTER than the 512-thread kernels from 384 RNS polynomials (2304 workgroups) up - the 512-thread kernel keeps only 1.7 of
//    its 2 workgroups per CU resident -, 4 % slower at 256 and below (two lockstep generations): launch_ntt picks the form by batch size, FoldArith contexts;
//  * giant-step key inner products as one workgroup per (item, limb, half): 3 % slower at the packed layers' size (profiles/r05_halves_relin_ab.txt) - removed in round 6.
constexpr size_t kHalvesMinPolys = 2304;   // residue polynomials per launch from which the halves form of the N = 8192 transforms wins (measured crossover: 1536 loses, 2304 wins)
// N = 16384 on the N = 4096 body ("quarters": ntt_quarters.h, kernels_quarters.h - two register column stages + four 4096-point sub-transforms through one LDS
// buffer, 256-thread workgroups, two to a CU).  Round 6, measured against the 1024-thread kernel on one box, alternated (profiles/r06_ntt14_quarters_sweep.txt):
// 5-13 % faster from 1536 workgroups up (forward 122 against 138 us at 512 RNS polynomials x 3 limbs, 458 against 493-502 at 2048; inverse 5-7 %), equal at 768,
// slower below (a workgroup lives four sub-transforms long: 28.7 against 21.4 us at 192).  launch_ntt picks the form by batch size, FoldArith contexts.
constexpr size_t kQuartersMinPolys = 768;
constexpr int kMaxGaloisBatch = 64;   // Galois elements travel as kernel arguments, this many per launch

// return 0, or -1 when log2n has no compiled geometry.  Launch errors are left in hipGetLastError().
template <class Arith>
int launch_ntt(int log2n, bool inverse, u64* out, const u64* in, size_t npolys, const DevTables<Arith>& tb, hipStream_t s);
template <class Arith>
int launch_ct_mul(int log2n, unsigned flags, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, hipStream_t s);

// round 6: the batched transforms of a context with per-limb arithmetic classes, one launch (kernels.h ntt_classes_kernel; k_ntt_classes.hip)
int launch_ntt_classes(int log2n, bool inverse, u64* out, const u64* in, size_t npolys, const MixedTables& tb, hipStream_t s);

// named forms of the fused multiply (launch_impl.h launch_ct_mul_variant; instantiated in k_ctmul_var.hip)
// (round 5: the single-transform, prefetching and two-pair forms never won on any of 19 boxes - profiles/r04_box_fingerprints.txt - and are gone)
enum CtMulVariant { kCtMulQuad = 0, kCtMulDual = 1, kCtMulVariants = 2 };
constexpr int ct_mul_default_variant(int log2n) { return log2n <= 12 ? kCtMulQuad : kCtMulDual; }   // what launch_ct_mul runs for flags = 0
template <class Arith>
int launch_ct_mul_variant(int log2n, int variant, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, hipStream_t s);
template <class Arith>
int launch_ct_mul_trace(int log2n, u64* out3, const u64* a2, const u64* b2, size_t blocks, const DevTables<Arith>& tb, u64* trace, hipStream_t s);

template <class Arith>
int launch_relin(int log2n, int mode, u64* out2, const u64* in3, const u64* evk, size_t key_stride, unsigned key_group, size_t blocks,
                 const DevTables<Arith>& tb, hipStream_t s);

// hoisted rotations: work[item][2][L][N] = INTT(sum_j perm_{g_item}(digits[j]) (.) keys[item][j]); g as kernel arguments (<= 64 items)
template <class Arith>
int launch_hoisted_ks(int log2n, u64* work, const u64* digits, const u64* keys, size_t key_stride, const unsigned* elts, size_t count,
                      size_t n_items, const DevTables<Arith>& tb, hipStream_t s);

// out = sigma_g(INTT(in)), the automorphism applied as a gather in the NTT domain; `polys_per_elt` residue polynomials per element (<= 64 elements)
template <class Arith>
int launch_ntt_inv_galois(int log2n, u64* out, const u64* in, const unsigned* elts, size_t n_elts, size_t polys_per_elt, const DevTables<Arith>& tb, hipStream_t s);

}  // namespace dpfhe
