// kernels_misc.h - streaming (HBM-bound) kernels: A3 dyadic ops, A7 ct x pt matvec, A8 shard-local reduce.
// Included by dpfhe_cabi.hip only.  No reference counterpart.
#pragma once
#include <hip/hip_runtime.h>

#include "launch.h"
#include "modarith.h"

namespace dpfhe {

// ------------------------------------------------------------------------------------------------
// A3: coefficient-wise ops.  One workgroup per residue polynomial (limb constants are scalar loads),
// 16-byte accesses, every load of an iteration issued before its first use.
// ------------------------------------------------------------------------------------------------
enum DyOp { DY_MUL = 0, DY_MUL_ADD = 1, DY_ADD = 2, DY_SUB = 3, DY_NEG = 4 };

template <class Arith, int OP>
__device__ __forceinline__ u64 dy_apply(u64 a, u64 b, u64 acc, const LimbConst& lc) {
    if (OP == DY_MUL) return Arith::mul_var(a, b, lc);
    if (OP == DY_MUL_ADD) return add_mod(acc, Arith::mul_var(a, b, lc), lc.q);
    if (OP == DY_ADD) return add_mod(a, b, lc.q);
    if (OP == DY_SUB) return sub_mod(a, b, lc.q);
    return neg_mod(a, lc.q);
}

// b_period = 0: b is shaped like a; b_period = L: ONE RNS polynomial (a plaintext), broadcast over the residue polynomials of a (A7 multiply_plain:
// its L tiles stay in L2, the launch moves two polynomials per polynomial instead of three)
// non-temporal 16-byte accesses (two 8-byte halves: the builtin takes scalars) for streams that cannot stay in the 256 MiB Infinity Cache
template <bool NT>
__device__ __forceinline__ U64x2 ld_vec(const U64x2* p) {
    if (NT) { U64x2 v; v.a = __builtin_nontemporal_load(&p->a); v.b = __builtin_nontemporal_load(&p->b); return v; }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st_vec(U64x2* p, const U64x2& v) {
    if (NT) { __builtin_nontemporal_store(v.a, &p->a); __builtin_nontemporal_store(v.b, &p->b); }
    else *p = v;
}

// NT: operands and result go around the Infinity Cache (chosen by the launcher when the launch touches more than the cache holds: +4-8 % there,
// slower below it, like the copy kernel)
template <class Arith, int OP, bool NT = false>
__global__ __launch_bounds__(256) void dyadic_kernel(u64* out, const u64* a, const u64* b, const LimbConst* lcs, int n_limbs, int n, int b_period = 0) {
    const size_t p = blockIdx.x;
    const LimbConst lc = lcs[p % (size_t)n_limbs];
    const U64x2* pa = reinterpret_cast<const U64x2*>(a + p * n);
    const U64x2* pb = reinterpret_cast<const U64x2*>(b + (b_period ? p % (size_t)b_period : p) * n);
    U64x2* po = reinterpret_cast<U64x2*>(out + p * n);
    const int nv = n >> 1;
    constexpr int UN = 4;
    for (int base = threadIdx.x; base < nv; base += 256 * UN) {
        U64x2 va[UN], vb[UN], vc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * 256;
            if (i < nv) {
                va[u] = ld_vec<NT>(pa + i);
                if (OP != DY_NEG) vb[u] = b_period ? pb[i] : ld_vec<NT>(pb + i);   // a broadcast plaintext is re-read: it stays cached
                if (OP == DY_MUL_ADD) vc[u] = ld_vec<NT>(po + i);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * 256;
            if (i < nv) {
                U64x2 r;
                r.a = dy_apply<Arith, OP>(va[u].a, vb[u].a, vc[u].a, lc);
                r.b = dy_apply<Arith, OP>(va[u].b, vb[u].b, vc[u].b, lc);
                st_vec<NT>(po + i, r);
            }
        }
    }
}

// The practical HBM ceiling the transforms are read against ("also report a measured device-copy bandwidth"):
// a plain copy with the streaming kernels' access shape - 16 bytes per lane, 4 KiB-contiguous per wave instruction, eight
// independent loads in flight per thread, one 32 KiB tile (a residue polynomial at N = 4096) per workgroup.
template <bool NT>   // NT: non-temporal loads and stores (a stream far beyond the 256 MiB Infinity Cache gains from not allocating there)
__global__ __launch_bounds__(256) void copy_kernel(U64x2* __restrict__ dst, const U64x2* __restrict__ src, size_t n_vec) {
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    if ((size_t)blockIdx.x * 2048 + 2048 <= n_vec) {   // whole tile (workgroup-uniform)
        U64x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NT) { v[u].a = __builtin_nontemporal_load(&src[base + u * 256].a); v[u].b = __builtin_nontemporal_load(&src[base + u * 256].b); }
            else v[u] = src[base + u * 256];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NT) { __builtin_nontemporal_store(v[u].a, &dst[base + u * 256].a); __builtin_nontemporal_store(v[u].b, &dst[base + u * 256].b); }
            else dst[base + u * 256] = v[u];
        }
    } else {
        for (size_t i = base; i < n_vec; i += 256) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------
// A8: out[c][l][:] = sum_i in[i][c][l][:]  (HBM-bound: one read per term).
// grid (residue-poly chunks, splits): each workgroup reduces its share of the batch to a canonical partial
// and adds it into `out` (zeroed by the caller on the same stream) with 64-bit atomics - at most 15
// partials < q < 2^60 per word, so the lazy sum cannot wrap.  reduce_final_kernel then canonicalises.
// No workspace, no per-call allocation.
// ------------------------------------------------------------------------------------------------
constexpr int kReduceSplits = 15;

// Work item = (512-word chunk of a residue polynomial, batch split); a workgroup walks the items with stride gridDim.x,
// so the launch size caps how much of the chip (and of the HBM bandwidth) the reduction takes at once.
__global__ __launch_bounds__(256) void reduce_partial_kernel(u64* out, const u64* in, const LimbConst* lcs, int n_limbs, int n, int chunks,
                                                             size_t count, size_t words_per_item, unsigned poly_chunks, unsigned nsplit) {
    for (unsigned work = blockIdx.x; work < poly_chunks * nsplit; work += gridDim.x) {
        const unsigned pc = work % poly_chunks;
        const size_t split = work / poly_chunks;
        const size_t p = pc / (unsigned)chunks;  // residue polynomial within one item
        const int w = (int)(pc % (unsigned)chunks) * 512 + threadIdx.x * 2;
        if (w >= n) continue;
        const u64 q = lcs[p % (size_t)n_limbs].q;
        const size_t lo = count * split / nsplit, hi = count * (split + 1) / nsplit;
        const u64* src = in + p * n + w;
        u64 s0 = 0, s1 = 0;
        size_t it = lo;
        for (; it + 4 <= hi; it += 4) {
            U64x2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const U64x2*>(src + (it + u) * words_per_item);   // (non-temporal loads measured no faster here: plain)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s0 = csub(s0 + v[u].a, q); s1 = csub(s1 + v[u].b, q); }
        }
        for (; it < hi; ++it) {
            const U64x2 v = *reinterpret_cast<const U64x2*>(src + it * words_per_item);
            s0 = csub(s0 + v.a, q); s1 = csub(s1 + v.b, q);
        }
        if (hi > lo) {
            atomicAdd(reinterpret_cast<unsigned long long*>(out + p * n + w), (unsigned long long)s0);
            atomicAdd(reinterpret_cast<unsigned long long*>(out + p * n + w + 1), (unsigned long long)s1);
        }
    }
}

__global__ __launch_bounds__(256) void reduce_final_kernel(u64* out, const LimbConst* lcs, int n_limbs, int n, int chunks) {
    const size_t p = blockIdx.x / chunks;
    const int w = (int)(blockIdx.x % chunks) * 512 + threadIdx.x * 2;
    if (w >= n) return;
    const u64 q = lcs[p % (size_t)n_limbs].q;
    U64x2* o = reinterpret_cast<U64x2*>(out + p * n + w);
    U64x2 v = *o;  // < 15 q
    v.a = csub(csub(csub(csub(v.a, 8 * q), 4 * q), 2 * q), q);
    v.b = csub(csub(csub(csub(v.b, 8 * q), 4 * q), 2 * q), q);
    *o = v;
}

// ------------------------------------------------------------------------------------------------
// A7: y[i][c][l][:] = sum_j W[i][j][l][:] (.) x[j][c][l][:],  c in {0,1}.  128-bit lazy accumulation,
// one reduction at the end.  One workgroup per (row, limb); W is streamed once (the HBM-bound term),
// x (cols x 2 RNS polys) is re-read by every row through L2 / Infinity Cache.
// ------------------------------------------------------------------------------------------------
struct Acc128 {
    u64 lo, hi;
};
__device__ __forceinline__ void acc_mac(Acc128& acc, u64 a, u64 b) {
    const u64 lo = a * b, hi = mulhi64(a, b);
    acc.lo += lo;
    acc.hi += hi + (acc.lo < lo);
}
template <class Arith>
__device__ __forceinline__ u64 acc_reduce(const Acc128& acc, const LimbConst& lc, u64 two64_mod_q) {
    if (Arith::kFold) {  // hi * 2^64 + lo  =  hi * (2^64 mod q) + lo
        const u64 h = FoldArith::mul60_full(acc.hi, two64_mod_q, (u32)lc.d);  // < 2q
        const u64 l = FoldArith::reduce(acc.lo, lc);                      // < 2q
        return FoldArith::canon(h + l, lc);
    } else {
        const u64 h = ShoupArith::mul_var(acc.hi % lc.q, two64_mod_q, lc);
        return add_mod(h, acc.lo % lc.q, lc.q);
    }
}

// RT rows per workgroup: x_j is loaded once per column and re-used from registers for RT rows, so the
// L2/Infinity-Cache traffic for x drops RT-fold and the W stream (read exactly once) owns the HBM pipe.
// grid = (row tiles * L * chunks); every thread owns 2 consecutive words of RT x 2 output polynomials.
template <class Arith, int RT>
__global__ __launch_bounds__(256) void matvec_kernel(u64* y, const u64* W, const u64* x, const LimbConst* lcs, int n_limbs, int n,
                                                     int chunks, size_t rows, size_t cols) {
    const size_t L = (size_t)n_limbs;
    // XCD-aware block ids (see matvec_multi_kernel): slab p = (limb, chunk) on XCD p mod 8, its row tiles adjacent there - the x tile of
    // a slab is fetched from HBM once and re-read from that XCD's L2 by the other row tiles
    const unsigned n_slabs = (unsigned)L * (unsigned)chunks, R = (unsigned)((rows + RT - 1) / RT);
    const unsigned id = blockIdx.x, q = id >> 3, slab = (q / R) * 8u + (id & 7u);
    if (slab >= n_slabs) return;
    const int chunk = (int)(slab % chunks);
    const int limb = (int)(slab / chunks);
    const size_t row0 = (size_t)(q % R) * RT;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const u64 two64 = lc.two64;
    const size_t wstride = L * n, xstride = 2 * L * n, rstride = cols * L * n;
    Acc128 acc[RT][2][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c][0] = acc[r][c][1] = Acc128{0, 0};
    const u64* wp = W + (row0 * cols * L + limb) * n + w0;
    const u64* xp = x + (size_t)limb * n + w0;
    size_t since = 0;
    // (requesting column j + 1 before the products of column j measured SLOWER here - 1.46 against 1.38 ms at configs[2]: the 10 extra registers cost the 4th wave per SIMD - removed)
    U64x2 x0 = *reinterpret_cast<const U64x2*>(xp), x1 = *reinterpret_cast<const U64x2*>(xp + L * n), w[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) w[r] = (row0 + r < rows) ? *reinterpret_cast<const U64x2*>(wp + r * rstride) : U64x2{0, 0};
    for (size_t j = 0; j < cols; ++j) {
        if (j > 0) {
            x0 = *reinterpret_cast<const U64x2*>(xp + j * xstride);
            x1 = *reinterpret_cast<const U64x2*>(xp + j * xstride + L * n);
#pragma unroll
            for (int r = 0; r < RT; ++r) w[r] = (row0 + r < rows) ? *reinterpret_cast<const U64x2*>(wp + r * rstride + j * wstride) : U64x2{0, 0};
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            acc_mac(acc[r][0][0], w[r].a, x0.a); acc_mac(acc[r][0][1], w[r].b, x0.b);
            acc_mac(acc[r][1][0], w[r].a, x1.a); acc_mac(acc[r][1][1], w[r].b, x1.b);
        }
        if (++since == 128) {  // 128 products of < 2^120 stay below 2^128 next to a reduced value
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 2; ++k) acc[r][c][k] = Acc128{acc_reduce<Arith>(acc[r][c][k], lc, two64), 0};
            since = 0;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (row0 + r >= rows) break;
        u64* yp = y + (((row0 + r) * 2) * L + limb) * n + w0;
        *reinterpret_cast<U64x2*>(yp) = U64x2{acc_reduce<Arith>(acc[r][0][0], lc, two64), acc_reduce<Arith>(acc[r][0][1], lc, two64)};
        *reinterpret_cast<U64x2*>(yp + L * n) = U64x2{acc_reduce<Arith>(acc[r][1][0], lc, two64), acc_reduce<Arith>(acc[r][1][1], lc, two64)};
    }
}

// A7 with several right-hand sides (tokens): y[i][t] = sum_j W[i][j] (.) x[j][t], x: [cols][C/2][2][L][N], y: [rows][C/2][2][L][N]; the
// C = 2 * tokens polynomials of a column are just more "components".  A weight tile is loaded once per RT rows and used for all
// tokens: the W stream (the 335 MB of diagonals of a packed GPT-2 layer) is read once per launch instead of once per token.
template <class Arith, int RT, int C>
__global__ __launch_bounds__(256) void matvec_multi_kernel(u64* y, const u64* W, const u64* x, const LimbConst* lcs, int n_limbs, int n,
                                                           int chunks, size_t rows, size_t cols, size_t polys_per_col /* >= C: the full [n_rhs][2] extent */,
                                                           unsigned n_groups, unsigned n_tiles) {
    // `n_groups` groups of C / 2 right-hand sides and all row tiles in ONE launch, block ids laid out for the 8 XCDs (workgroups are
    // dealt to them round-robin by id): slab p = (limb, chunk of 512 words) lives on XCD p mod 8, and its row tiles x groups are
    // adjacent ids on that XCD - id = ((p / 8) * R * G + r * G + g) * 8 + p % 8.  The G workgroups of a row tile read the same W tile and
    // the R workgroups of a group the same x tile at about the same time: each is fetched from HBM once, the rest are L2 hits.
    const size_t L = (size_t)n_limbs;
    const unsigned n_slabs = (unsigned)L * (unsigned)chunks, R = n_tiles / n_slabs;   // n_tiles = row tiles * slabs
    const unsigned id = blockIdx.x, lane = id & 7u, q = id >> 3;
    const unsigned group = q % n_groups, rt = (q / n_groups) % R, slab = (q / (n_groups * R)) * 8u + lane;
    if (slab >= n_slabs) return;
    const int chunk = (int)(slab % chunks);
    const int limb = (int)(slab / chunks);
    const size_t row0 = (size_t)rt * RT;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    x += (size_t)group * C * L * n;
    y += (size_t)group * C * L * n;
    const LimbConst lc = lcs[limb];
    const u64 two64 = lc.two64;
    const size_t wstride = L * n, xstride = polys_per_col * L * n, rstride = cols * L * n;
    Acc128 acc[RT][C][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[r][c][0] = acc[r][c][1] = Acc128{0, 0};
    const u64* wp = W + (row0 * cols * L + limb) * n + w0;
    const u64* xp = x + (size_t)limb * n + w0;
    size_t since = 0;
    // software pipeline: the operands of column j + 1 are requested before the RT * C multiply-accumulates of column j (there are
    // only `cols` = 32 iterations in a packed layer and two waves per SIMD: nothing else hides the HBM latency of the W tiles)
    U64x2 w[RT], xv[C];
#pragma unroll
    for (int r = 0; r < RT; ++r) w[r] = (row0 + r < rows) ? *reinterpret_cast<const U64x2*>(wp + r * rstride) : U64x2{0, 0};
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = *reinterpret_cast<const U64x2*>(xp + (size_t)c * L * n);
    for (size_t j = 0; j < cols; ++j) {
        const size_t jn = j + 1 < cols ? j + 1 : j;
        U64x2 wn[RT], xn[C];
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = (row0 + r < rows) ? *reinterpret_cast<const U64x2*>(wp + r * rstride + jn * wstride) : U64x2{0, 0};
#pragma unroll
        for (int c = 0; c < C; ++c) xn[c] = *reinterpret_cast<const U64x2*>(xp + jn * xstride + (size_t)c * L * n);
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int r = 0; r < RT; ++r) { acc_mac(acc[r][c][0], w[r].a, xv[c].a); acc_mac(acc[r][c][1], w[r].b, xv[c].b); }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) w[r] = wn[r];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = xn[c];
        if (++since == 128) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int k = 0; k < 2; ++k) acc[r][c][k] = Acc128{acc_reduce<Arith>(acc[r][c][k], lc, two64), 0};
            since = 0;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (row0 + r >= rows) break;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            u64* yp = y + (((row0 + r) * polys_per_col + c) * L + limb) * n + w0;
            *reinterpret_cast<U64x2*>(yp) = U64x2{acc_reduce<Arith>(acc[r][c][0], lc, two64), acc_reduce<Arith>(acc[r][c][1], lc, two64)};
        }
    }
}

// A7 for FoldArith contexts (round 3): the same product with the split-at-bit-30 column accumulators of modarith.h
// (FoldArith::dot30_mac: FOUR multiply-adds per term instead of the 21 instructions of a 128-bit accumulate; one 18-instruction
// fold per 8 columns).  W tile (RT rows) and x tile (C polynomials) of a column are split once and used RT x C times.  Same
// XCD-aware block ids, same software pipeline; WPT words per thread (chunks of 256 WPT words).  Serves dpfhe_matvec_plain
// (C = 2, one group) and dpfhe_matvec_plain_multi.
template <int WPT> struct WordVec;
template <> struct WordVec<1> { u64 v[1]; };
template <> struct __attribute__((aligned(16))) WordVec<2> { u64 v[2]; };
template <int WPT, bool NT = false>
__device__ __forceinline__ WordVec<WPT> load_words(const u64* p, bool in_range) {   // rows beyond the matrix read as zero
    WordVec<WPT> r;
    if (in_range) {
        if (NT) {
#pragma unroll
            for (int k = 0; k < WPT; ++k) r.v[k] = __builtin_nontemporal_load(p + k);
        } else r = *reinterpret_cast<const WordVec<WPT>*>(p);
    } else {
#pragma unroll
        for (int k = 0; k < WPT; ++k) r.v[k] = 0;
    }
    return r;
}

// NTW: the W tiles are read once by ONE workgroup (a single group of right-hand sides) and W does not fit the Infinity Cache: non-temporal loads
// WD: how many columns AHEAD the W words are requested (a register ring of WD x RT words; WD divides the period).  The W stream is the one
// operand that comes from HBM (one workgroup group reads each tile once), x comes from L2: round 4 measured ~3.6 us from issue to first word for
// an HBM burst under load, against ~0.5 us of products per column - one column of lookahead cannot cover it.
// FULL: rows is a multiple of RT and cols a multiple of the period (what a packed layer's products are) - no row predicate on the W loads and no
// ragged last period, i.e. NO control flow inside the period: its 8 columns are one basic block, the double-buffered operands are renamed at
// compile time and the loads of column j + 1 are waited for where column j + 1 first uses them.  With the checks in place every column was its
// own block that ended in register copies of the next operands behind s_waitcnt vmcnt(0): the one-column lookahead the source asks for
// did not exist in the binary.
constexpr int kMatvecFullDepth = 1;   // columns of lookahead of the FULL form (both operands)
template <int RT, int C, int WPT, bool NTW = false, int WD = 1, bool FULL = false>
__global__ __launch_bounds__(256) void matvec_fold_kernel(u64* y, const u64* W, const u64* x, const LimbConst* lcs, int n_limbs, int n,
                                                          int chunks, size_t rows, size_t cols, size_t polys_per_col, unsigned n_groups,
                                                          unsigned n_tiles) {
    typedef WordVec<WPT> V;
    typedef FoldArith::Half30 H;
    constexpr int P = FoldArith::kDot30Period;
    const size_t L = (size_t)n_limbs;
    const unsigned n_slabs = (unsigned)L * (unsigned)chunks, R = n_tiles / n_slabs;
    const unsigned id = blockIdx.x, lane = id & 7u, q = id >> 3;
    const unsigned group = q % n_groups, rt = (q / n_groups) % R, slab = (q / (n_groups * R)) * 8u + lane;
    if (slab >= n_slabs) return;
    const int chunk = (int)(slab % chunks);
    const int limb = (int)(slab / chunks);
    const size_t row0 = (size_t)rt * RT;
    const unsigned w0 = (unsigned)(chunk * 256 * WPT + (int)threadIdx.x * WPT);   // the ONLY per-thread part of every address
    if ((int)w0 >= n) return;
    x += (size_t)group * C * L * n;
    y += (size_t)group * C * L * n;
    const LimbConst lc = lcs[limb];
    const size_t wstride = L * n, xstride = polys_per_col * L * n, rstride = cols * L * n;
    // uniform bases (scalar registers, scalar adds) + one 32-bit lane offset: no per-lane 64-bit address arithmetic
    const u64* const wu = W + (row0 * cols * L + limb) * n;
    const u64* const xu = x + (size_t)limb * n;
    // (uniform 64-bit base) + (32-bit BYTE offset of the lane): the shape the global_load "saddr" form takes - base in scalar registers, one
    // 32-bit vector offset shared by every load of the thread; indexing the u64 pointer with w0 instead makes hipcc scale in 64 bits per lane
    // (one v_lshl_add_u64 and a 64-bit vector address per load: 12 % of the kernel's vector instructions)
    auto ld_w = [&](int r, size_t j) { return load_words<WPT, NTW>(&(wu + (r * rstride + j * wstride))[w0], FULL || row0 + r < rows); };
    auto ld_x = [&](int c, size_t j) { return *reinterpret_cast<const V*>(&(xu + (j * xstride + (size_t)c * L * n))[w0]); };
    u64 run[RT][C][WPT];   // folded running words (reduced); the first products of every period chain onto them
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int k = 0; k < WPT; ++k) run[r][c][k] = 0;
    if constexpr (FULL) {
        // Both operands of column j + FD are requested at the top of column j into a register ring of FD columns, and a scheduling barrier
        // closes every column: without it the scheduler pulls the (cheap) splits of the next column's operands up into the current one and
        // the wait for them with it - half a column of lookahead instead of FD columns.
        constexpr int FD = kMatvecFullDepth;
        static_assert(P % FD == 0, "the operand ring must divide the period");
        // Every load of the thread is (uniform 64-bit base) + (ONE 32-bit byte offset of the lane).  Written as pointer arithmetic hipcc folds the lane part
        // into a 64-bit vector base and adds the uniform part per load (one v_lshl_add_u64 and an address register pair each: 12 % of the kernel's vector
        // instructions); a raw buffer load takes the base in scalar registers (the descriptor, rebuilt per load by the scalar unit - W may exceed a 32-bit
        // offset) and the lane offset as is.
        const unsigned b0 = w0 * 8u;
        auto bld = [&](const u64* uniform) {
            V r;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(uniform), 0, 0x7fffffff, 0x00020000);
            if constexpr (WPT == 1) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)b0, 0, NTW ? 2 : 0);
                r.v[0] = ((u64)t.y << 32) | t.x;
            } else {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)b0, 0, NTW ? 2 : 0);
                r.v[0] = ((u64)t.y << 32) | t.x; r.v[1] = ((u64)t.w << 32) | t.z;
            }
            return r;
        };
        auto ld_w = [&](int r, size_t j) { return bld(wu + (r * rstride + j * wstride)); };
        auto ld_x = [&](int c, size_t j) { return bld(xu + (j * xstride + (size_t)c * L * n)); };
        V wr[FD][RT], xr[FD][C];
#pragma unroll
        for (int d = 0; d < FD; ++d) {
            const size_t jd = (size_t)d < cols ? (size_t)d : cols - 1;
#pragma unroll
            for (int r = 0; r < RT; ++r) wr[d][r] = ld_w(r, jd);
#pragma unroll
            for (int c = 0; c < C; ++c) xr[d][c] = ld_x(c, jd);
        }
        for (size_t j0 = 0; j0 < cols; j0 += P) {
            FoldArith::Dot30 acc[RT][C][WPT];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                const size_t j = j0 + u, jp = j + FD < cols ? j + FD : cols - 1;
                V w[RT], xv[C];
#pragma unroll
                for (int r = 0; r < RT; ++r) { w[r] = wr[u % FD][r]; wr[u % FD][r] = ld_w(r, jp); }
#pragma unroll
                for (int c = 0; c < C; ++c) { xv[c] = xr[u % FD][c]; xr[u % FD][c] = ld_x(c, jp); }
                __builtin_amdgcn_sched_barrier(0);   // ... and the requests stay up here (the scheduler otherwise sinks them to the end of the column to save registers)
#pragma unroll
                for (int k = 0; k < WPT; ++k) {
                    H wh[RT], xh[C];
#pragma unroll
                    for (int r = 0; r < RT; ++r) wh[r] = FoldArith::split30(w[r].v[k]);
#pragma unroll
                    for (int c = 0; c < C; ++c) xh[c] = FoldArith::split30(xv[c].v[k]);
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            if (u == 0) acc[r][c][k] = FoldArith::Dot30{run[r][c][k], 0, 0};
                            FoldArith::dot30_mac(acc[r][c][k], wh[r], xh[c]);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int k = 0; k < WPT; ++k) run[r][c][k] = FoldArith::dot30_fold0(acc[r][c][k], lc);
                __builtin_amdgcn_sched_barrier(0);   // one row of folds at a time: interleaving all sixteen costs ten more registers (and the third wave per SIMD)
            }
        }
    } else {
    static_assert(WD >= 1 && FoldArith::kDot30Period % WD == 0, "the W ring must divide the period");
    V wq[WD][RT], xv[C];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int r = 0; r < RT; ++r) wq[d][r] = ld_w(r, (size_t)d < cols ? (size_t)d : cols - 1);
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = ld_x(c, 0);
    // Periods of P columns, unrolled: the accumulators of a period start from {running word, 0, 0} as multiply-add addends (no
    // zeroing moves), the operands of column j + 1 are requested before the RT x C products of column j, one fold ends the period.
    for (size_t j0 = 0; j0 < cols; j0 += P) {
        FoldArith::Dot30 acc[RT][C][WPT];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const size_t j = j0 + u;
            if constexpr (!FULL) {
                if (j >= cols) {   // ragged last period: nothing to add (uniform branch)
                    if (u == 0) break;
                    continue;
                }
            }
            const size_t jn = j + 1 < cols ? j + 1 : j, jw = j + WD < cols ? j + WD : cols - 1;
            V w[RT], xn[C];
#pragma unroll
            for (int r = 0; r < RT; ++r) { w[r] = wq[u % WD][r]; wq[u % WD][r] = ld_w(r, jw); }   // slot of column j is free: column j + WD goes there
#pragma unroll
            for (int c = 0; c < C; ++c) xn[c] = ld_x(c, jn);
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                H wh[RT], xh[C];
#pragma unroll
                for (int r = 0; r < RT; ++r) wh[r] = FoldArith::split30(w[r].v[k]);
#pragma unroll
                for (int c = 0; c < C; ++c) xh[c] = FoldArith::split30(xv[c].v[k]);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (u == 0) acc[r][c][k] = FoldArith::Dot30{run[r][c][k], 0, 0};
                        FoldArith::dot30_mac(acc[r][c][k], wh[r], xh[c]);
                    }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) xv[c] = xn[c];
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < WPT; ++k) run[r][c][k] = FoldArith::dot30_fold(acc[r][c][k], 0, lc);
    }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (!FULL && row0 + r >= rows) break;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            V o;
#pragma unroll
            for (int k = 0; k < WPT; ++k) o.v[k] = FoldArith::canon_small(run[r][c][k], lc);
            *reinterpret_cast<V*>(&(y + (((row0 + r) * polys_per_col + c) * L + limb) * n)[w0]) = o;
        }
    }
}

// A7, scalar weights (SURVEY.md section 8a A7 "cheap special case"): W_ij are residues in Z_q (one word per limb),
// y_i = sum_j w_ij * x_j.  The weights of a row tile are workgroup-uniform (scalar loads -> SGPR multiplier operands),
// x_j is loaded once per column and used for RT rows: RT*4 multiply-accumulates per 32 bytes loaded - ALU-bound.
// d_w: [rows][cols][L];  x: [cols][2][L][N];  y: [rows][2][L][N].
template <class Arith, int RT>
__global__ __launch_bounds__(256) void matvec_scalar_kernel(u64* y, const u64* w, const u64* x, const LimbConst* lcs, int n_limbs, int n,
                                                            int chunks, size_t rows, size_t cols) {
    const size_t L = (size_t)n_limbs;
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % L);
    const size_t row0 = (blockIdx.x / chunks / L) * RT;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const u64 two64 = lc.two64;
    const size_t xstride = 2 * L * n;
    Acc128 acc[RT][2][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c][0] = acc[r][c][1] = Acc128{0, 0};
    const u64* xp = x + (size_t)limb * n + w0;
    size_t since = 0;
    for (size_t j = 0; j < cols; ++j) {
        const U64x2 x0 = *reinterpret_cast<const U64x2*>(xp + j * xstride);
        const U64x2 x1 = *reinterpret_cast<const U64x2*>(xp + j * xstride + L * n);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const size_t row = row0 + r < rows ? row0 + r : rows - 1;   // clamp: out-of-range rows are computed, not stored
            const u64 wr = w[(row * cols + j) * L + limb];              // uniform -> scalar load
            acc_mac(acc[r][0][0], wr, x0.a); acc_mac(acc[r][0][1], wr, x0.b);
            acc_mac(acc[r][1][0], wr, x1.a); acc_mac(acc[r][1][1], wr, x1.b);
        }
        if (++since == 128) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 2; ++k) acc[r][c][k] = Acc128{acc_reduce<Arith>(acc[r][c][k], lc, two64), 0};
            since = 0;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (row0 + r >= rows) break;
        u64* yp = y + (((row0 + r) * 2) * L + limb) * n + w0;
        *reinterpret_cast<U64x2*>(yp) = U64x2{acc_reduce<Arith>(acc[r][0][0], lc, two64), acc_reduce<Arith>(acc[r][0][1], lc, two64)};
        *reinterpret_cast<U64x2*>(yp + L * n) = U64x2{acc_reduce<Arith>(acc[r][1][0], lc, two64), acc_reduce<Arith>(acc[r][1][1], lc, two64)};
    }
}

// N1 (second half): rescale / modulus switch to the next level, coefficient domain.
//   out_i = ((in_i + h) - ((in_last + h) mod q_last)) * q_last^-1   (mod q_i),  h = floor(q_last / 2),  i < L-1
// i.e. round(x / q_last) of the CRT-composed value, limb by limb (no big integers).  HBM-bound.
struct RescaleConst {
    u64 h_mod;     // floor(q_last / 2) mod q_i
    u64 inv;       // q_last^-1 mod q_i
    u64 q_last;
    u64 h;         // floor(q_last / 2)
};

// `addend` (optional) is the last step of hybrid key switching: polynomial p = 2*b + comp of the rescaled pair gets
// component comp of ciphertext b of `addend` ([batch][add_in_comps][L-1][N]) added when bit comp of add_mask is set.
template <class Arith>
__global__ __launch_bounds__(256) void rescale_kernel(u64* out, const u64* in, const u64* addend, int add_in_comps, int add_mask,
                                                      const LimbConst* lcs, const RescaleConst* rcs, int n_limbs, int n, int chunks) {
    const int Lo = n_limbs - 1;
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % Lo);
    const size_t poly = blockIdx.x / chunks / Lo;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const RescaleConst rc = rcs[limb];
    const U64x2 x = *reinterpret_cast<const U64x2*>(in + (poly * n_limbs + limb) * n + w0);
    const U64x2 last = *reinterpret_cast<const U64x2*>(in + (poly * n_limbs + Lo) * n + w0);
    U64x2 r;
    {
        const u64 t = csub(last.a + rc.h, rc.q_last);             // (in_last + h) mod q_last
        const u64 tm = Arith::kFold ? FoldArith::canon(t, lc) : ShoupArith::mul_var(t, 1, lc);   // ... mod q_i
        const u64 d = sub_mod(add_mod(x.a, rc.h_mod, lc.q), tm, lc.q);
        r.a = Arith::mul_var(d, rc.inv, lc);
    }
    {
        const u64 t = csub(last.b + rc.h, rc.q_last);
        const u64 tm = Arith::kFold ? FoldArith::canon(t, lc) : ShoupArith::mul_var(t, 1, lc);
        const u64 d = sub_mod(add_mod(x.b, rc.h_mod, lc.q), tm, lc.q);
        r.b = Arith::mul_var(d, rc.inv, lc);
    }
    if (addend && ((add_mask >> (poly & 1)) & 1)) {
        const size_t ap = (poly >> 1) * (size_t)add_in_comps + (poly & 1);
        const U64x2 ad = *reinterpret_cast<const U64x2*>(addend + (ap * Lo + limb) * n + w0);
        r.a = add_mod(r.a, ad.a, lc.q);
        r.b = add_mod(r.b, ad.b, lc.q);
    }
    *reinterpret_cast<U64x2*>(out + (poly * Lo + limb) * n + w0) = r;
}

// N3 (round 3, deferred divide-by-P): the last step of a baby-step / giant-step sum whose key-switched terms were summed over Q P.
//   out[b][comp][i] = round(in[b][comp] / P)_i  +  sum_{a < n_add} addends[a][b][0][i]   (comp 0)
//                                               +  addends[0][b][1][i]                    (comp 1)
// in: [batch][2][L][N] (coefficient domain, Q P), addends: [n_add][batch][2][Ld][N] (the rotated inner sums: item 0 is the
// un-rotated one and keeps both components, of the others only c0 survives - their c1 went through the key switch), out:
// [batch][2][Ld][N].  One pass; the rounding is orc_rescale's.
template <class Arith>
__global__ __launch_bounds__(256) void rescale_bsgs_kernel(u64* out, const u64* in, const u64* addends, size_t n_add, size_t batch, const LimbConst* lcs,
                                                           const RescaleConst* rcs, int n_limbs, int n, int chunks) {
    const int Lo = n_limbs - 1;
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % Lo);
    const size_t poly = blockIdx.x / chunks / Lo;   // = b * 2 + comp
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const RescaleConst rc = rcs[limb];
    const U64x2 x = *reinterpret_cast<const U64x2*>(in + (poly * n_limbs + limb) * n + w0);
    const U64x2 last = *reinterpret_cast<const U64x2*>(in + (poly * n_limbs + Lo) * n + w0);
    U64x2 r;
    {
        const u64 t = csub(last.a + rc.h, rc.q_last);
        const u64 tm = Arith::kFold ? FoldArith::canon(t, lc) : ShoupArith::mul_var(t, 1, lc);
        r.a = Arith::mul_var(sub_mod(add_mod(x.a, rc.h_mod, lc.q), tm, lc.q), rc.inv, lc);
    }
    {
        const u64 t = csub(last.b + rc.h, rc.q_last);
        const u64 tm = Arith::kFold ? FoldArith::canon(t, lc) : ShoupArith::mul_var(t, 1, lc);
        r.b = Arith::mul_var(sub_mod(add_mod(x.b, rc.h_mod, lc.q), tm, lc.q), rc.inv, lc);
    }
    const size_t terms = (poly & 1) ? (n_add ? 1 : 0) : n_add;
    const u64* ap = addends + (poly * Lo + limb) * n + w0;
    const size_t astride = batch * 2 * Lo * (size_t)n;
    size_t a = 0;
    for (; a + 4 <= terms; a += 4) {
        U64x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const U64x2*>(ap + (a + u) * astride);
#pragma unroll
        for (int u = 0; u < 4; ++u) { r.a = add_mod(r.a, v[u].a, lc.q); r.b = add_mod(r.b, v[u].b, lc.q); }
    }
    for (; a < terms; ++a) {
        const U64x2 v = *reinterpret_cast<const U64x2*>(ap + a * astride);
        r.a = add_mod(r.a, v.a, lc.q); r.b = add_mod(r.b, v.b, lc.q);
    }
    *reinterpret_cast<U64x2*>(out + (poly * Lo + limb) * n + w0) = r;
}

// N3 (round 3): a ciphertext in the NTT domain on the data limbs, lifted to the extended basis as P * ct: word * (P mod q_i) on the
// data limbs, 0 on the special limb (P = 0 mod P).  in: [n_polys][Ld][N] -> out: [n_polys][L][N].  The un-rotated baby step.
template <class Arith>
__global__ __launch_bounds__(256) void lift_qp_kernel(u64* out, const u64* in, const LimbConst* lcs, u64 p_special, int n_limbs, int n, int chunks) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t poly = blockIdx.x / chunks / n_limbs;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    U64x2 r{0, 0};
    if (limb < n_limbs - 1) {
        const LimbConst lc = lcs[limb];
        const u64 pmod = Arith::kFold ? FoldArith::canon(p_special, lc) : ShoupArith::mul_var(p_special, 1, lc);
        const U64x2 v = *reinterpret_cast<const U64x2*>(in + (poly * (n_limbs - 1) + limb) * n + w0);
        r.a = Arith::mul_var(v.a, pmod, lc);
        r.b = Arith::mul_var(v.b, pmod, lc);
    }
    *reinterpret_cast<U64x2*>(out + (poly * n_limbs + limb) * n + w0) = r;
}

// (the exact base extension / scale-and-round kernels live in kernels_bx.h: one kernel per source-limb count, compiled in their own
// translation units k_bx_fold.hip / k_bx_shoup.hip)

// ------------------------------------------------------------------------------------------------
// N3, round 4: the baby-step pass of the double-hoisted packed products as a STREAM (replaces kernels.h hoisted_qp_kernel, whose one
// workgroup per (rotation, limb, token) re-read the digit images once per rotation: 2.3 x its algorithmic bytes in L2 misses).
//   out[(r, t)] = ( sum_j perm_g(digit_{t,j}) (.) key_{r,j,0} + P perm_g(NTT(c0_t)),  sum_j perm_g(digit_{t,j}) (.) key_{r,j,1} )   over all L limbs.
// No transform, so no NTT geometry: a workgroup is 256 threads x PP 16-byte pairs = one SEGMENT of 512 PP words of one (rotation, limb, token).
// Keys and results are read / written in natural (forward-output) order, lane-contiguous; only the digit words are permuted, and sigma_g
// in forward-output order maps every aligned pair ONTO an aligned pair (swapped or not) and every aligned segment onto an aligned segment:
//   pair m -> pair m' = brv((c - 1) / 2),  c = g (2 brv(m) + 1) mod 2N reduced mod N,  words swapped iff that product is >= N
// (brv over log2 N - 1 bits) - one 16-byte gather per pair inside ONE source segment, indices computed once per thread and reused for
// all digits.  What makes the traffic algorithmic is WHERE workgroups run: ids are dealt to the 8 XCDs round-robin, and XCD x gets, one
// block of 16 rotations x n_items tokens at a time, all workgroups of one (limb, SOURCE segment): the block's workgroups are resident
// together (5 per CU), every key segment is fetched from HBM once for its n_items tokens and every digit segment once for its 16 rotations.
// ------------------------------------------------------------------------------------------------
constexpr int kQpPairs = 2;          // 16-byte pairs per thread: segment = 512 pairs = 1024 words
constexpr int kQpRotGroup = 16;      // rotations that share a digit segment in one XCD block
struct QpElts { unsigned v[kMaxGaloisBatch]; };

__device__ __forceinline__ unsigned qp_brev(unsigned x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <class Arith, int PP>
__global__ __launch_bounds__(256) void hoisted_qp_stream_kernel(u64* __restrict__ out, const u64* __restrict__ digits, const u64* __restrict__ xntt,
                                                                const u64* __restrict__ keys, size_t key_stride, QpElts elts, unsigned n_rot, unsigned n_items,
                                                                u64 p_special, const LimbConst* __restrict__ lcs, int n_limbs, int log2n) {
    const int L = n_limbs, Ld = L - 1, n1 = log2n - 1;
    const unsigned half = 1u << n1, n = 2u << n1, seg_pairs = 256u * PP;
    const unsigned nseg = half > seg_pairs ? half / seg_pairs : 1u, tbits = 31u - (unsigned)__clz((int)nseg);
    const unsigned combos = (unsigned)L * nseg, n_rg = (n_rot + kQpRotGroup - 1) / kQpRotGroup, bs = (unsigned)kQpRotGroup * n_items;
    // id -> (XCD x, block of one (limb, source segment, rotation group), rotation in group, token)
    const unsigned xcd = blockIdx.x & 7u, q = blockIdx.x >> 3, within = q % bs, t1 = q / bs, rg = t1 % n_rg, combo = (t1 / n_rg) * 8u + xcd;
    const unsigned rot = rg * kQpRotGroup + within / n_items, token = within % n_items;
    if (combo >= combos || rot >= n_rot) return;
    const int limb = (int)(combo % (unsigned)L);
    const unsigned sseg = combo / (unsigned)L;
    const unsigned g = elts.v[rot];
    // output segment whose sources are segment sseg:  2 rev(oseg) + 1 = g^-1 (2 rev(sseg) + 1)  mod 2 nseg
    unsigned ginv = g;
#pragma unroll
    for (int i = 0; i < 4; ++i) ginv *= 2u - g * ginv;                  // g^-1 mod 2^32
    const unsigned uo = (ginv * (2u * qp_brev(sseg, (int)tbits) + 1u)) & (2u * nseg - 1u);
    const unsigned oseg = qp_brev((uo - 1u) >> 1, (int)tbits);
    const LimbConst lc = lcs[limb];
    const size_t N = n;
    unsigned ms[PP], mo[PP];
    bool sw[PP], ok[PP];
#pragma unroll
    for (int c = 0; c < PP; ++c) {
        mo[c] = oseg * seg_pairs + (unsigned)c * 256u + threadIdx.x;
        ok[c] = mo[c] < half;
        const unsigned cf = (g * (2u * qp_brev(mo[c], n1) + 1u)) & (2u * n - 1u);
        sw[c] = cf >= n;
        ms[c] = qp_brev(((cf & (n - 1u)) - 1u) >> 1, n1);
    }
    const size_t item = (size_t)rot * n_items + token;
    const u64* dig = digits + ((size_t)token * Ld * L + limb) * N;      // digit j at + j L N
    const u64* evk = keys + (size_t)rot * key_stride + (size_t)limb * N; // key (j, comp) at + (j 2 + comp) L N
    auto gather = [&](U64x2 (&x)[PP], const u64* tile) {
#pragma unroll
        for (int c = 0; c < PP; ++c)
            if (ok[c]) x[c] = reinterpret_cast<const U64x2*>(tile)[ms[c]];
    };
    auto unswap = [&](U64x2 (&x)[PP]) {
#pragma unroll
        for (int c = 0; c < PP; ++c) { const u64 a = x[c].a, b = x[c].b; x[c].a = sw[c] ? b : a; x[c].b = sw[c] ? a : b; }
    };
    auto stream = [&](U64x2 (&e)[PP], const u64* tile) {
#pragma unroll
        for (int c = 0; c < PP; ++c)
            if (ok[c]) e[c] = reinterpret_cast<const U64x2*>(tile)[mo[c]];
    };
    u64* o0 = out + ((item * 2 + 0) * L + limb) * N;
    u64* o1 = out + ((item * 2 + 1) * L + limb) * N;
    U64x2 x[PP], e0[PP], e1[PP];
#pragma unroll
    for (int c = 0; c < PP; ++c) { x[c] = U64x2{0, 0}; e0[c] = U64x2{0, 0}; e1[c] = U64x2{0, 0}; }
    if constexpr (Arith::kFold) {
        // Lazy inner products on 30-bit halves (modarith.h Dot30: FOUR multiply-adds per term, the split of a digit word shared by both key
        // components, one fold per kDot30Period terms) instead of one mul60 (15 instructions) per product: the pass is VALU-bound once its
        // traffic is algorithmic.  All operands are canonical residues (< 2^60), which is what split30 needs.
        typedef FoldArith::Dot30 Dot;
        typedef FoldArith::Half30 Half;
        Dot s0[PP][2], s1[PP][2];
        u64 r0[PP][2], r1[PP][2];
#pragma unroll
        for (int c = 0; c < PP; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) { s0[c][h] = Dot{0, 0, 0}; s1[c][h] = Dot{0, 0, 0}; r0[c][h] = 0; r1[c][h] = 0; }
        int terms = 0;
        auto fold_all = [&]() {
#pragma unroll
            for (int c = 0; c < PP; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    r0[c][h] = FoldArith::dot30_fold(s0[c][h], r0[c][h], lc); s0[c][h] = Dot{0, 0, 0};
                    r1[c][h] = FoldArith::dot30_fold(s1[c][h], r1[c][h], lc); s1[c][h] = Dot{0, 0, 0};
                }
            terms = 0;
        };
        if (limb < Ld) {   // component 0 starts with P perm_g(NTT(c0)) on the data limbs (P = 0 on the special limb)
            U64x2 c0[PP];
#pragma unroll
            for (int c = 0; c < PP; ++c) c0[c] = U64x2{0, 0};
            gather(c0, xntt + ((size_t)token * 2 * Ld + limb) * N);
            gather(x, dig);
            stream(e0, evk);
            stream(e1, evk + (size_t)L * N);
            unswap(c0);
            const Half hp = FoldArith::split30(FoldArith::canon(p_special, lc));
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                FoldArith::dot30_mac(s0[c][0], FoldArith::split30(c0[c].a), hp);
                FoldArith::dot30_mac(s0[c][1], FoldArith::split30(c0[c].b), hp);
            }
            terms = 1;
        } else {
            gather(x, dig);
            stream(e0, evk);
            stream(e1, evk + (size_t)L * N);
        }
#pragma unroll 1
        for (int j = 0; j < Ld; ++j) {
            U64x2 xn[PP], e0n[PP], e1n[PP];
            const int jn = j + 1 < Ld ? j + 1 : j;   // the last iteration re-requests its own segments (cache hits) instead of branching
#pragma unroll
            for (int c = 0; c < PP; ++c) { xn[c] = x[c]; e0n[c] = e0[c]; e1n[c] = e1[c]; }
            gather(xn, dig + (size_t)jn * L * N);
            stream(e0n, evk + (size_t)(jn * 2) * L * N);
            stream(e1n, evk + (size_t)(jn * 2 + 1) * L * N);
            unswap(x);
            if (terms == FoldArith::kDot30Period) fold_all();   // workgroup-uniform
            ++terms;
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                const Half xa = FoldArith::split30(x[c].a), xb = FoldArith::split30(x[c].b);
                FoldArith::dot30_mac(s0[c][0], xa, FoldArith::split30(e0[c].a));
                FoldArith::dot30_mac(s0[c][1], xb, FoldArith::split30(e0[c].b));
                FoldArith::dot30_mac(s1[c][0], xa, FoldArith::split30(e1[c].a));
                FoldArith::dot30_mac(s1[c][1], xb, FoldArith::split30(e1[c].b));
            }
#pragma unroll
            for (int c = 0; c < PP; ++c) { x[c] = xn[c]; e0[c] = e0n[c]; e1[c] = e1n[c]; }
        }
        fold_all();
#pragma unroll
        for (int c = 0; c < PP; ++c) {
            if (!ok[c]) continue;
            // written once, read by the next launch: around the caches the keys and digits live in
            st_vec<true>(reinterpret_cast<U64x2*>(o0) + mo[c], U64x2{FoldArith::canon_small(r0[c][0], lc), FoldArith::canon_small(r0[c][1], lc)});
            st_vec<true>(reinterpret_cast<U64x2*>(o1) + mo[c], U64x2{FoldArith::canon_small(r1[c][0], lc), FoldArith::canon_small(r1[c][1], lc)});
        }
    } else {
        U64x2 acc0[PP], acc1[PP];
#pragma unroll
        for (int c = 0; c < PP; ++c) { acc0[c] = U64x2{0, 0}; acc1[c] = U64x2{0, 0}; }
        auto accw = [&](u64 acc, u64 a, u64 b) -> u64 { return add_mod(acc, ShoupArith::mul_var(a, b, lc), lc.q); };
        if (limb < Ld) {
            U64x2 c0[PP];
#pragma unroll
            for (int c = 0; c < PP; ++c) c0[c] = U64x2{0, 0};
            gather(c0, xntt + ((size_t)token * 2 * Ld + limb) * N);
            unswap(c0);
            const u64 pmod = ShoupArith::mul_var(p_special, 1, lc);
#pragma unroll
            for (int c = 0; c < PP; ++c) { acc0[c].a = ShoupArith::mul_var(c0[c].a, pmod, lc); acc0[c].b = ShoupArith::mul_var(c0[c].b, pmod, lc); }
        }
#pragma unroll 1
        for (int j = 0; j < Ld; ++j) {
            gather(x, dig + (size_t)j * L * N);
            stream(e0, evk + (size_t)(j * 2) * L * N);
            stream(e1, evk + (size_t)(j * 2 + 1) * L * N);
            unswap(x);
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                acc0[c].a = accw(acc0[c].a, x[c].a, e0[c].a); acc0[c].b = accw(acc0[c].b, x[c].b, e0[c].b);
                acc1[c].a = accw(acc1[c].a, x[c].a, e1[c].a); acc1[c].b = accw(acc1[c].b, x[c].b, e1[c].b);
            }
        }
#pragma unroll
        for (int c = 0; c < PP; ++c) {
            if (!ok[c]) continue;
            st_vec<true>(reinterpret_cast<U64x2*>(o0) + mo[c], acc0[c]);
            st_vec<true>(reinterpret_cast<U64x2*>(o1) + mo[c], acc1[c]);
        }
    }
}
// The same pass with EVERY operand segment of the workgroup requested up front (round 4, after the workgroup timeline of the multiply put the
// cost of a first HBM touch at ~3.6 us): the loop form above keeps one digit's segments in flight while it multiplies the previous one, i.e. it
// pays that latency once per digit; here a thread owns ONE pair of words, the digit count LD is a template parameter, and the 3 LD + 1 loads of
// the thread are all issued before the first product (straight-line code: hipcc counts vmcnt exactly).  FoldArith, LD <= 6.
template <int LD>
__global__ __launch_bounds__(256) void hoisted_qp_upfront_kernel(u64* __restrict__ out, const u64* __restrict__ digits, const u64* __restrict__ xntt,
                                                                 const u64* __restrict__ keys, size_t key_stride, QpElts elts, unsigned n_rot, unsigned n_items,
                                                                 u64 p_special, const LimbConst* __restrict__ lcs, int log2n) {
    constexpr int L = LD + 1;
    const int n1 = log2n - 1;
    const unsigned half = 1u << n1, n = 2u << n1, seg_pairs = 256u;
    const unsigned nseg = half > seg_pairs ? half / seg_pairs : 1u, tbits = 31u - (unsigned)__clz((int)nseg);
    const unsigned combos = (unsigned)L * nseg, n_rg = (n_rot + kQpRotGroup - 1) / kQpRotGroup, bs = (unsigned)kQpRotGroup * n_items;
    const unsigned xcd = blockIdx.x & 7u, q = blockIdx.x >> 3, within = q % bs, t1 = q / bs, rg = t1 % n_rg, combo = (t1 / n_rg) * 8u + xcd;
    const unsigned rot = rg * kQpRotGroup + within / n_items, token = within % n_items;
    if (combo >= combos || rot >= n_rot) return;
    const int limb = (int)(combo % (unsigned)L);
    const unsigned sseg = combo / (unsigned)L;
    const unsigned g = elts.v[rot];
    unsigned ginv = g;
#pragma unroll
    for (int i = 0; i < 4; ++i) ginv *= 2u - g * ginv;
    const unsigned uo = (ginv * (2u * qp_brev(sseg, (int)tbits) + 1u)) & (2u * nseg - 1u);
    const unsigned oseg = qp_brev((uo - 1u) >> 1, (int)tbits);
    const LimbConst lc = lcs[limb];
    const size_t N = n;
    const unsigned mo = oseg * seg_pairs + threadIdx.x;
    if (mo >= half) return;
    const unsigned cf = (g * (2u * qp_brev(mo, n1) + 1u)) & (2u * n - 1u);
    const bool sw = cf >= n;
    const unsigned ms = qp_brev(((cf & (n - 1u)) - 1u) >> 1, n1);
    const size_t item = (size_t)rot * n_items + token;
    const U64x2* dig = reinterpret_cast<const U64x2*>(digits + ((size_t)token * LD * L + limb) * N) + ms;      // digit j at + j L N words
    const U64x2* evk = reinterpret_cast<const U64x2*>(keys + (size_t)rot * key_stride + (size_t)limb * N) + mo;  // key (j, comp) at + (j 2 + comp) L N words
    const size_t tile = (size_t)L * N / 2;   // in 16-byte units
    U64x2 x[LD], e0[LD], e1[LD], c0{0, 0};
    const bool data_limb = limb < LD;
    if (data_limb) c0 = (reinterpret_cast<const U64x2*>(xntt + ((size_t)token * 2 * LD + limb) * N))[ms];
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        x[j] = dig[(size_t)j * tile];
        e0[j] = evk[(size_t)(2 * j) * tile];
        e1[j] = evk[(size_t)(2 * j + 1) * tile];
    }
    typedef FoldArith::Dot30 Dot;
    typedef FoldArith::Half30 Half;
    static_assert(LD + 1 <= FoldArith::kDot30Period, "one fold per accumulator");
    Dot s0[2] = {Dot{0, 0, 0}, Dot{0, 0, 0}}, s1[2] = {Dot{0, 0, 0}, Dot{0, 0, 0}};
    if (data_limb) {   // P perm_g(NTT(c0)) on the data limbs
        const Half hp = FoldArith::split30(FoldArith::canon(p_special, lc));
        FoldArith::dot30_mac(s0[0], FoldArith::split30(sw ? c0.b : c0.a), hp);
        FoldArith::dot30_mac(s0[1], FoldArith::split30(sw ? c0.a : c0.b), hp);
    }
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        const Half xa = FoldArith::split30(sw ? x[j].b : x[j].a), xb = FoldArith::split30(sw ? x[j].a : x[j].b);
        FoldArith::dot30_mac(s0[0], xa, FoldArith::split30(e0[j].a));
        FoldArith::dot30_mac(s0[1], xb, FoldArith::split30(e0[j].b));
        FoldArith::dot30_mac(s1[0], xa, FoldArith::split30(e1[j].a));
        FoldArith::dot30_mac(s1[1], xb, FoldArith::split30(e1[j].b));
    }
    U64x2 r0, r1;
    r0.a = FoldArith::canon_small(FoldArith::dot30_fold(s0[0], 0, lc), lc); r0.b = FoldArith::canon_small(FoldArith::dot30_fold(s0[1], 0, lc), lc);
    r1.a = FoldArith::canon_small(FoldArith::dot30_fold(s1[0], 0, lc), lc); r1.b = FoldArith::canon_small(FoldArith::dot30_fold(s1[1], 0, lc), lc);
    st_vec<true>(reinterpret_cast<U64x2*>(out + ((item * 2 + 0) * L + limb) * N) + mo, r0);
    st_vec<true>(reinterpret_cast<U64x2*>(out + ((item * 2 + 1) * L + limb) * N) + mo, r1);
}

// grid of the stream kernel (dpfhe_cabi.hip): 8 XCDs x blocks of (16 rotations x n_items) x rotation groups x ceil(L nseg / 8)
inline size_t qp_stream_grid(int log2n, int n_limbs, size_t n_rot, size_t n_items, int pairs = kQpPairs) {
    const size_t half = (size_t)1 << (log2n - 1), seg_pairs = 256 * (size_t)pairs, nseg = half > seg_pairs ? half / seg_pairs : 1;
    const size_t combos = (size_t)n_limbs * nseg, n_rg = (n_rot + kQpRotGroup - 1) / kQpRotGroup;
    return 8 * ((combos + 7) / 8) * n_rg * (size_t)kQpRotGroup * n_items;
}

// N3, hoisted rotations: digit j of the key-switched component (limb j of c1, coefficient domain, values < q_j) lifted to every
// limb i of the extended basis: out[j][i][k] = c1[j][k] mod q_i (canonical).  One workgroup per (digit, limb, 512-word chunk).
template <class Arith>
// Several input ciphertexts: item t reads c1 + t * in_item_stride and writes out + t * (n_limbs - 1) * n_limbs * n.
__global__ __launch_bounds__(256) void lift_digits_kernel(u64* out, const u64* c1, size_t in_item_stride, const LimbConst* lcs, int n_limbs, int n, int chunks) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t dg = blockIdx.x / chunks / n_limbs, item = dg / (unsigned)(n_limbs - 1), digit = dg % (unsigned)(n_limbs - 1);
    c1 += item * in_item_stride;
    out += item * (size_t)(n_limbs - 1) * n_limbs * n;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const U64x2 v = *reinterpret_cast<const U64x2*>(c1 + digit * n + w0);
    U64x2 r;
    r.a = Arith::kFold ? FoldArith::canon(v.a, lc) : ShoupArith::mul_var(v.a, 1, lc);
    r.b = Arith::kFold ? FoldArith::canon(v.b, lc) : ShoupArith::mul_var(v.b, 1, lc);
    *reinterpret_cast<U64x2*>(out + (digit * n_limbs + limb) * n + w0) = r;
}

// N3: Galois automorphism a(X) -> a(X^g), g odd, coefficient domain (a signed permutation; HBM-bound).
// Gather form: out[k] = +in[j] if j = k g^-1 mod 2N < N, else -in[j - N]  (coalesced writes, scattered 8-byte reads).
__global__ __launch_bounds__(256) void galois_kernel(u64* out, const u64* in, const LimbConst* lcs, int n_limbs, int n, unsigned g_inv) {
    const size_t p = blockIdx.x;
    const u64 q = lcs[p % (size_t)n_limbs].q;
    const unsigned mask2n = 2u * (unsigned)n - 1u;
    for (int k = threadIdx.x; k < n; k += 256) {
        const unsigned j = ((unsigned)k * g_inv) & mask2n;
        const u64 v = in[p * n + (j & ((unsigned)n - 1u))];
        out[p * n + k] = (j < (unsigned)n) ? v : neg_mod(v, q);
    }
}

// batched rotations: item i applies its own element (g_inv.v[i]) to input item i (or to the single input item when
// in_item_stride == 0); polys_per_item residue polynomials per item
struct GaloisInvs { unsigned v[kMaxGaloisBatch]; };
// in_mod > 0: output item i reads input item (item0 + i) % in_mod (several rotations of each of in_mod inputs: rotation-major order)
// LDS_STAGE (N <= 8192: the polynomial fits 64 KiB of dynamic LDS): the source polynomial is read with coalesced 16-byte loads
// into LDS and gathered from there - the odd multiplier g^-1 spreads consecutive k over distinct banks - instead of 8-byte
// global gathers that touch a different cache line per lane (packed GPT-2 layer, 8 tokens per application: 0.278 -> 0.265 ms per token).
template <bool LDS_STAGE>
__global__ __launch_bounds__(256) void galois_multi_kernel(u64* out, const u64* in, size_t in_item_stride, const LimbConst* lcs, int n_limbs, int n,
                                                           int polys_per_item, GaloisInvs g_inv, unsigned in_mod, unsigned item0) {
    extern __shared__ __attribute__((aligned(16))) u64 stage[];
    const size_t item = blockIdx.x / (unsigned)polys_per_item, p = blockIdx.x % (unsigned)polys_per_item;
    const u64 q = lcs[p % (size_t)n_limbs].q;
    const unsigned mask2n = 2u * (unsigned)n - 1u, gi = g_inv.v[item];
    const size_t in_item = in_mod ? (item0 + item) % in_mod : item;
    const u64* src = in + in_item * in_item_stride + p * n;
    u64* dst = out + (item * polys_per_item + p) * n;
    if (LDS_STAGE) {
        for (int k = threadIdx.x * 2; k < n; k += 512) *reinterpret_cast<U64x2*>(stage + k) = *reinterpret_cast<const U64x2*>(src + k);
        __syncthreads();
        for (int k = threadIdx.x * 2; k < n; k += 512) {
            const unsigned j0 = ((unsigned)k * gi) & mask2n, j1 = (j0 + gi) & mask2n;
            const u64 v0 = stage[j0 & ((unsigned)n - 1u)], v1 = stage[j1 & ((unsigned)n - 1u)];
            *reinterpret_cast<U64x2*>(dst + k) = U64x2{(j0 < (unsigned)n) ? v0 : neg_mod(v0, q), (j1 < (unsigned)n) ? v1 : neg_mod(v1, q)};
        }
        return;
    }
    for (int k = threadIdx.x; k < n; k += 256) {
        const unsigned j = ((unsigned)k * gi) & mask2n;
        const u64 v = src[j & ((unsigned)n - 1u)];
        dst[k] = (j < (unsigned)n) ? v : neg_mod(v, q);
    }
}

}  // namespace dpfhe
