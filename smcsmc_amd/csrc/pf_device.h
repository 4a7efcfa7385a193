// smcsmc_amd/csrc/pf_device.h -- device-side building blocks of the gfx950 particle filter.
//
// Compiled only by hipcc for gfx950 with -ffp-contract=off: every double operation below is
// one IEEE-754 operation, so the results are a pure function of the inputs and of the
// operation order written here (DESIGN.md "canonical arithmetic").
//
// Reference behaviour restated here (paths relative to /root/reference/src):
//   fastexp                       particle.cpp:30-40
//   site likelihood (pruning)     particle.cpp:625-680
//   tracked branch length         particle.cpp:699-730
//   next recombination position   particle.cpp:1195-1254
//   recombination point sampling  particle.cpp:1060-1126 (unbiased case)
//   SMC' re-coalescence           scrm fork Forest::sampleCoalescences, mirrored by particle.cpp:1266-1384
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PF_BS
#define PF_BS 256          // threads per workgroup of the per-particle kernels (4 wavefronts) = LDS stride of per-lane columns
#endif
#define PF_NMAX 16         // maximum number of haplotypes
#define PF_INF (__longlong_as_double(0x7ff0000000000000LL))

namespace pf {

// ------------------------------------------------------------------ libm-free exp / log
// fdlibm e_exp.c / e_log.c argument reduction + minimax polynomials (published constants).
__device__ __forceinline__ double dexp(double x) {
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (x > 709.782712893383973096) return PF_INF;
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    int k = 0;
    double hi = x, lo = 0.0;
    if (ax > 0.34657359027997264) {
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) {
        return __longlong_as_double(__double_as_longlong(y) + ((long long)k << 52));
    } else {
        y = __longlong_as_double(__double_as_longlong(y) + ((long long)(k + 1000) << 52));
        return y * 9.33263618503218878990e-302;
    }
}

__device__ __forceinline__ double dlog(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    int k = 0;
    unsigned long long ux = (unsigned long long)__double_as_longlong(x);
    int hx = (int)(ux >> 32);
    if (hx < 0x00100000) {
        k -= 54;
        x *= 18014398509481984.0;
        ux = (unsigned long long)__double_as_longlong(x);
        hx = (int)(ux >> 32);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int i = (hx + 0x95f64) & 0x100000;
    ux = (ux & 0x00000000ffffffffULL) | ((unsigned long long)(unsigned)(hx | (i ^ 0x3ff00000)) << 32);
    x = __longlong_as_double((long long)ux);
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}

// reference: particle.cpp:30-40
// particle.cpp:65-74
__device__ __forceinline__ double exp_digamma(double x) {
    if (x > 10) return x - 0.5 + (x + 0.5) / (24 * x * x);
    double f = 0.0;
    while (x < 6) {
        f = f + 1.0 / x;
        x = x + 1.0;
    }
    double psi = dlog(x) - 1 / (2 * x) - 1 / (12 * x * x);
    return dexp(psi - f);
}

// particle.cpp:45-55
__device__ __forceinline__ double fastexp_approx(double x) {
    double xx = x * x;
    if (xx < 2.099166) return 1 + 2 * x / (2 - x + xx * (1.0 / 6));
    return dexp(x);
}

__device__ __forceinline__ double fastexp(double x) {
    double xx = x * x;
    if (xx < 0.516167859) {
        return 1 + 2 * x / (2 - x + xx / (6 + xx * 0.1));
    } else {
        return dexp(x);
    }
}

// ------------------------------------------------------------------ Philox4x32-10, one lane = one stream
__device__ __forceinline__ double philox_uniform(unsigned long long seed, unsigned slot, unsigned stream,
                                                 unsigned long long draw) {
    unsigned c0 = (unsigned)draw, c1 = (unsigned)(draw >> 32), c2 = slot, c3 = stream;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low half (integer multiplies are
        // the slow instructions of this loop)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        unsigned n0 = hi1 ^ c1 ^ k0;
        unsigned n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    unsigned long long bits = (((unsigned long long)c0 << 32) | c1) >> 11;
    return ((double)bits + 0.5) * 1.1102230246251565e-16;
}

// Both halves of the Philox block at one draw index (the first equals philox_uniform).  A genealogy update of the
// one-population engine takes its four uniforms (cut point, waiting-time refresh, re-attachment slot, next
// recombination position) from two consecutive blocks instead of four (a block is ~100 instructions; measured
// ~300 cycles with one wavefront per SIMD).
__device__ __forceinline__ void philox_pair(unsigned long long seed, unsigned slot, unsigned stream, unsigned long long draw,
                                            double& u0, double& u1) {
    unsigned c0 = (unsigned)draw, c1 = (unsigned)(draw >> 32), c2 = slot, c3 = stream;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low half (integer multiplies are
        // the slow instructions of this loop)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        unsigned n0 = hi1 ^ c1 ^ k0;
        unsigned n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    unsigned long long b0 = (((unsigned long long)c0 << 32) | c1) >> 11;
    unsigned long long b1 = (((unsigned long long)c2 << 32) | c3) >> 11;
    u0 = ((double)b0 + 0.5) * 1.1102230246251565e-16;
    u1 = ((double)b1 + 0.5) * 1.1102230246251565e-16;
}

// The index of a workgroup within its chunk, and its chunk: the row kernels of the sweep (k_sweep*, k_sweep_blc) are launched with the grid
// (workgroups per chunk, chunks), every other kernel with a grid of one dimension or (workgroups, epochs).  The dispatcher hands
// workgroups out x fastest: one chunk after the other.  Other orders were built on these two functions and measured in round 4
// (profiles/round4/wg_trace.md: one workgroup per chunk in turn, with and without a rotation over the XCDs; slabs of 24 ... 162
// workgroups per chunk, e.g. every chunk's extend role before any chunk's counts): all slower with counting, and the index arithmetic
// alone cost one chunk 1 %.
__device__ __forceinline__ int pf_bx() { return (int)blockIdx.x; }
__device__ __forceinline__ int pf_chunk() { return (int)blockIdx.y; }

// ------------------------------------------------------------------ wavefront (64-lane) primitives
// xor-butterfly sum: every lane ends with the same pairwise-tree total (matches oracle tree64).
__device__ __forceinline__ double wave_tree_sum(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
// Sum over the wavefront without LDS traffic: row shifts and row broadcasts of the data-parallel primitives (four steps inside the
// rows of sixteen lanes, two across them), total read from lane 63.  A different association from wave_tree_sum's butterfly --
// for sums whose grouping is not part of the canonical arithmetic (the lagged counts) -- and the same one in every run.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wave_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(shi, slo);
}
__device__ __forceinline__ double wave_total_dpp(double v) {
    v = wave_dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = wave_dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = wave_dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = wave_dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row's sum
    v = wave_dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = wave_dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// Hillis-Steele inclusive scan across the wavefront (matches oracle hs64).
__device__ __forceinline__ double wave_hs_scan(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d, 64);
        if (lane >= d) v = o + v;
    }
    return v;
}
__device__ __forceinline__ double wave_max_scan_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d, 64);
        if (lane >= d) v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max_scan_i(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v = max(v, o);
    }
    return v;
}

// ------------------------------------------------------------------ per-lane state held in LDS
// A particle's local tree lives in LDS for the duration of a kernel so that it can be indexed
// with run-time ranks without falling to scratch memory: S[r] at sS[r*PF_BS + tid] (stride-BS:
// lane i hits bank (2*i) mod 64 -> conflict-free ds_read_b64), children at sC[(2r+s)*PF_BS + tid].
struct Lane {
    double* S;        // &sS[tid]
    int8_t* C;        // &sC[tid]
    const double* T;  // epoch start times   (LDS, shared by the block)
    const double* I;  // 1/(2 N_e)           (LDS)
    const double* H;  // cumulative coalescence intensity at the epoch starts (global, read-only)
    double uq0, uq1, uq2, uq3;   // uniforms of the genealogy update in progress
    int uqn = 0;
    const int* RF;    // record flags        (LDS)
    int E, n;
    double L, mu, rho;
    unsigned long long seed;
    unsigned slot;
    unsigned stream;          // Philox stream id: 0 particle filter, 2 lag calibration
    unsigned long long ctr;   // draws consumed by this slot
    double ebuf;              // buffered unit exponential (RandomGenerator::sampleExpoLimit)
    double Ltree;
    const double* vbc;        // variational-Bayes factor per (epoch[, population]) of a coalescence, or null
    double upd_fac;           // product of the factors of the events of the current walk
};

#define LS(ln, r) ((ln).S[(r) * PF_BS])
#define LC(ln, r, s) ((ln).C[((r) * 2 + (s)) * PF_BS])

// the uniforms of the update in progress (see philox_pair) are handed out first
__device__ __forceinline__ void prefetch_update_uniforms(Lane& ln) {
    philox_pair(ln.seed, ln.slot, ln.stream, ln.ctr, ln.uq0, ln.uq1);
    philox_pair(ln.seed, ln.slot, ln.stream, ln.ctr + 1, ln.uq2, ln.uq3);
    ln.ctr += 2;
    ln.uqn = 4;
}
__device__ __forceinline__ double uni(Lane& ln) {
    if (ln.uqn > 0) {
        const double u = ln.uqn == 4 ? ln.uq0 : (ln.uqn == 3 ? ln.uq1 : (ln.uqn == 2 ? ln.uq2 : ln.uq3));
        --ln.uqn;
        return u;
    }
    return philox_uniform(ln.seed, ln.slot, ln.stream, ln.ctr++);
}

__device__ __forceinline__ int epoch_of(const Lane& ln, double t) {
    int e = 0;
    while (e + 1 < ln.E && ln.T[e + 1] <= t) ++e;
    return e;
}
__device__ __forceinline__ double epoch_end(const Lane& ln, int e) { return e + 1 < ln.E ? ln.T[e + 1] : PF_INF; }
__device__ __forceinline__ double node_h(const Lane& ln, int id) { return id < ln.n ? 0.0 : LS(ln, id - ln.n); }

__device__ __forceinline__ double tree_length(const Lane& ln, int nleaves) {
    double acc = 0.0, prev = 0.0;
    for (int r = 0; r < nleaves - 1; ++r) {
        double s = LS(ln, r);
        acc += (double)(nleaves - r) * (s - prev);
        prev = s;
    }
    return acc;
}

// lineages of the stored tree (ni internal nodes) crossing `time`, canonical order
// (parent rank ascending, child 0 then 1); returns the count, writes the want-th slot.
__device__ __forceinline__ int lineages_at(const Lane& ln, int ni, double time, int want, int* pr, int* ps) {
    int R = 0;
    while (R < ni && LS(ln, R) <= time) ++R;
    int cnt = 0;
    for (int r = R; r < ni; ++r)
        for (int s = 0; s < 2; ++s) {
            int id = LC(ln, r, s);
            if (id < ln.n || id - ln.n < R) {
                if (cnt == want) { *pr = r; *ps = s; }
                ++cnt;
            }
        }
    return cnt;
}

__device__ __forceinline__ void remove_rank(Lane& ln, int ni, int rp, int sib, int* a, int* b) {
    const int pid = ln.n + rp;
    for (int r = rp + 1; r < ni; ++r)
        for (int s = 0; s < 2; ++s)
            if (LC(ln, r, s) == pid) LC(ln, r, s) = (int8_t)sib;
    for (int r = rp; r + 1 < ni; ++r) {
        LS(ln, r) = LS(ln, r + 1);
        LC(ln, r, 0) = LC(ln, r + 1, 0);
        LC(ln, r, 1) = LC(ln, r + 1, 1);
    }
    for (int r = 0; r < ni - 1; ++r)
        for (int s = 0; s < 2; ++s)
            if (LC(ln, r, s) > pid) LC(ln, r, s) -= 1;
    if (*a > pid) *a -= 1;
    if (*b > pid) *b -= 1;
}

__device__ __forceinline__ void insert_node(Lane& ln, int ni, double h, int fl, int pr, int ps, int root_id) {
    int rn = 0;
    while (rn < ni && LS(ln, rn) <= h) ++rn;
    const int nid = ln.n + rn;
    for (int r = 0; r < ni; ++r)
        for (int s = 0; s < 2; ++s)
            if (LC(ln, r, s) >= nid) LC(ln, r, s) += 1;
    if (fl >= nid) fl += 1;
    if (root_id >= nid) root_id += 1;
    for (int r = ni; r > rn; --r) {
        LS(ln, r) = LS(ln, r - 1);
        LC(ln, r, 0) = LC(ln, r - 1, 0);
        LC(ln, r, 1) = LC(ln, r - 1, 1);
    }
    int target;
    if (pr >= 0) {
        if (pr >= rn) pr += 1;
        target = LC(ln, pr, ps);
        LC(ln, pr, ps) = (int8_t)nid;
    } else {
        target = root_id;
    }
    LS(ln, rn) = h;
    LC(ln, rn, 0) = (int8_t)fl;
    LC(ln, rn, 1) = (int8_t)target;
}

// Walk the floating lineage upwards from height h through the intervals delimited by the
// heights Sh[0..ns) (read through the functor) and the epoch boundaries; lineage count
// k(t) = nl - #{Sh <= t}, 1 above the top.  Returns the coalescence time.
template <class HeightAt>
__device__ __forceinline__ double coalesce_up(Lane& ln, HeightAt Sh, int ns, int nl, double h) {
    // The waiting time is found on the cumulative intensity Hc(t) = int_0^t 1/(2N(s)) ds (piecewise linear, tabulated at
    // the epoch starts) instead of epoch by epoch: between two nodes the lineage count k is constant, so the
    // unit-exponential budget is compared with k (Hc(next node) - Hc(t)) once per node, and the event time is the
    // inverse of Hc at Hc(t) + budget / k.  Same distribution as the interval walk, O(nodes + log E) steps.
    int e = epoch_of(ln, h);
    double Hc = ln.H[e] + (h - ln.T[e]) * ln.I[e];
    int i = 0;
    while (i < ns && Sh(i) <= h) ++i;
    double lower = h, kd;
    for (;;) {
        if (i >= ns) { kd = 1.0; break; }
        kd = (double)(nl - i);
        const double sn = Sh(i);
        const int en = epoch_of(ln, sn);
        const double Hn = ln.H[en] + (sn - ln.T[en]) * ln.I[en];
        const double need = (Hn - Hc) * kd;
        if (!(ln.ebuf > need)) break;
        ln.ebuf -= need;
        Hc = Hn; lower = sn; ++i;
    }
    const double C = Hc + ln.ebuf / kd;
    int es = 0;
    while (es + 1 < ln.E && ln.H[es + 1] <= C) ++es;
    double t1 = ln.T[es] + (C - ln.H[es]) / ln.I[es];
    if (t1 < lower) t1 = lower;
    if (i < ns) { const double sn = Sh(i); if (t1 > sn) t1 = sn; }
    ln.ebuf = -dlog(uni(ln));
    if (ln.vbc) ln.upd_fac *= ln.vbc[es];
    return t1;
}

// particle.cpp:1195-1254 with multiplicity 1 and one recombination-rate segment
__device__ __forceinline__ double sample_next_base(Lane& ln, double x) {
    double rate = ln.rho * ln.Ltree;
    double limit = ln.L - x;
    double need = limit * rate;
    if (ln.ebuf > need) {
        ln.ebuf -= need;
        return ln.L;
    }
    double nb = x + ln.ebuf / rate;
    ln.ebuf = -dlog(uni(ln));
    if (nb == x) {   // particle.cpp:1238-1244: nextafter(x, x*2+1)
        nb = __longlong_as_double(__double_as_longlong(x) + 1);
        if (x == 0.0) nb = 4.9406564584124654e-324;
    }
    if (nb > ln.L) nb = ln.L;
    return nb;
}

// particle.cpp:699-730 on the rank-sorted tree; `tmp` is a per-lane LDS scratch of 2n doubles
__device__ __forceinline__ double tracked_length(const Lane& ln, const int8_t* data, double* tmp) {
    const int n = ln.n;
    for (int i = 0; i < n; ++i) tmp[i * PF_BS] = data[i] >= 0 ? 0.0 : -1.0;
    double total = 0.0;
    for (int r = 0; r < n - 1; ++r) {
        int c0 = LC(ln, r, 0), c1 = LC(ln, r, 1);
        double sr = LS(ln, r);
        double l = tmp[c0 * PF_BS], rr = tmp[c1 * PF_BS];
        if (l >= 0.0) l += sr - node_h(ln, c0);
        if (rr >= 0.0) rr += sr - node_h(ln, c1);
        double v;
        if (l >= 0.0 && rr >= 0.0) { total = l + rr; v = total; }
        else if (l >= 0.0) v = l;
        else v = rr;
        tmp[(n + r) * PF_BS] = v;
    }
    return total;
}

// particle.cpp:625-680; hap bits: per leaf two bits packed in `m0mask`/`m1mask` (bit i set = likelihood 1)
__device__ __forceinline__ double site_likelihood(const Lane& ln, unsigned m0mask, unsigned m1mask, bool ancestral_aware,
                                                  double* t0, double* t1) {
    const int n = ln.n;
    for (int i = 0; i < n; ++i) {
        t0[i * PF_BS] = (m0mask >> i) & 1 ? 1.0 : 0.0;
        t1[i * PF_BS] = (m1mask >> i) & 1 ? 1.0 : 0.0;
    }
    for (int r = 0; r < n - 1; ++r) {
        int c0 = LC(ln, r, 0), c1 = LC(ln, r, 1);
        double sr = LS(ln, r);
        double tl = sr - node_h(ln, c0);
        double trr = sr - node_h(ln, c1);
        double pl = fastexp(-tl * ln.mu);
        double pr = fastexp(-trr * ln.mu);
        double a0 = t0[c0 * PF_BS], a1 = t1[c0 * PF_BS], b0 = t0[c1 * PF_BS], b1 = t1[c1 * PF_BS];
        t0[(n + r) * PF_BS] = (a0 * pl + a1 * (1 - pl)) * (b0 * pr + b1 * (1 - pr));
        t1[(n + r) * PF_BS] = (a1 * pl + a0 * (1 - pl)) * (b1 * pr + b0 * (1 - pr));
    }
    int root = n + n - 2;
    double p0 = ancestral_aware ? 1.0 : 0.5, p1 = ancestral_aware ? 0.0 : 0.5;
    return t0[root * PF_BS] * p0 + t1[root * PF_BS] * p1;
}

}  // namespace pf
