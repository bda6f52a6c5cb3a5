/*
 * gl_oracle.c -- CPU ORACLE for the STARK hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is a plain-C restatement of the algorithms that the reference
 * (okx/xlayer-prover, snapshot 2024_10_08) runs on its Starks::genProof path:
 * Goldilocks field, cubic extension, NTT/INTT/extendPol, Poseidon permutation,
 * linear hash, Merkle tree, Fiat-Shamir transcript, FRI fold/transpose/query.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (xlayer_prover_amd/) never links or calls it.
 *
 * Provenance.  The arithmetic library the reference calls
 * (github.com/0xPolygonHermez/goldilocks, git submodule `src/goldilocks`,
 * .gitmodules:1-3) is NOT vendored in the snapshot and its pinned commit is
 * unknown.  Each function below therefore restates the *published algorithm*
 * and is anchored to the reference's own call sites and in-tree restatements
 * (cited per function).  Parity is PINNED by golden data held in the
 * reference's testvectors (see tests/golden/make_fixtures.py and
 * tests/test_oracle_golden.py):
 *   - 1243 Poseidon known answers (`db` objects of testvectors JSON files),
 *   - two complete STARK proofs (recursive1 / recursive2 zkin files): every
 *     Merkle path, the transcript-derived query indices and every FRI fold.
 * The LDE *values* are not pinned by any golden file (no committed trace is in
 * the tree): extendPol is checked against a naive O(n^2) evaluation instead,
 * and its evaluation-domain convention (shift = 7, natural order, W32 below)
 * is pinned through the FRI fold relations of the golden proofs.
 *
 * All cited paths are relative to /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL

/* ------------------------------------------------------------------ field */
/* Goldilocks::Element arithmetic (absent lib; semantics = Z/pZ, canonical
 * outputs).  Used everywhere e.g. src/starkpil/polinomial.hpp:140-229. */
static inline u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }
static inline u64 gl_add(u64 a, u64 b)
{
    a = gl_canon(a); b = gl_canon(b);
    u64 s = a + b;
    if (s < a || s >= GL_P) s -= GL_P;
    return s;
}
static inline u64 gl_sub(u64 a, u64 b)
{
    a = gl_canon(a); b = gl_canon(b);
    return a >= b ? a - b : a + (GL_P - b);
}
static inline u64 gl_mul(u64 a, u64 b)
{
    return (u64)(((u128)a * (u128)b) % (u128)GL_P);
}
static u64 gl_pow(u64 a, u64 e)
{
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

u64 orc_add(u64 a, u64 b) { return gl_add(a, b); }
u64 orc_sub(u64 a, u64 b) { return gl_sub(a, b); }
u64 orc_mul(u64 a, u64 b) { return gl_mul(a, b); }
u64 orc_inv(u64 a) { return gl_inv(a); }
u64 orc_pow(u64 a, u64 e) { return gl_pow(a, e); }

/* Goldilocks::w(n): primitive 2^n-th root of unity, w(n) = W32^(2^(32-n)).
 * W32 is pinned by the golden proofs' FRI folds: with
 * 7^((p-1)/2^32) every fold check fails.  Call sites: starks.hpp:149-160,
 * friProve.cpp:36, zhInv.cpp:12-26. */
#define GL_W32 7277203076849721926ULL
#define GL_SHIFT 7ULL /* Goldilocks::shift(), friProve.cpp:12-13 */
u64 orc_w(unsigned n)
{
    u64 w = GL_W32;
    for (unsigned i = 32; i > n; i--) w = gl_mul(w, w);
    return w;
}
u64 orc_shift(void) { return GL_SHIFT; }

/* -------------------------------------------------------- cubic extension */
/* Fp[x]/(x^3 - x - 1); Karatsuba form exactly as polinomial.hpp:195-205. */
static inline void gl3_mul(u64 o[3], const u64 a[3], const u64 b[3])
{
    u64 A = gl_mul(gl_add(a[0], a[1]), gl_add(b[0], b[1]));
    u64 B = gl_mul(gl_add(a[0], a[2]), gl_add(b[0], b[2]));
    u64 C = gl_mul(gl_add(a[1], a[2]), gl_add(b[1], b[2]));
    u64 D = gl_mul(a[0], b[0]);
    u64 E = gl_mul(a[1], b[1]);
    u64 F = gl_mul(a[2], b[2]);
    u64 G = gl_sub(D, E);
    u64 r0 = gl_sub(gl_add(C, G), F);
    u64 r1 = gl_sub(gl_sub(gl_sub(gl_add(A, C), E), E), D);
    u64 r2 = gl_sub(B, G);
    o[0] = r0; o[1] = r1; o[2] = r2;
}
void orc_mul3(u64 o[3], const u64 a[3], const u64 b[3]) { gl3_mul(o, a, b); }

/* ---------------------------------------------------------------- Poseidon */
/* Permutation restated from src/sm/poseidon_g/poseidon_g_executor.cpp:201-230
 * (round loop), :325-331 (pow7), constants :11-101, MDS
 * poseidon_g_executor.hpp:37-50: M[i][j] = MCIRC[(j-i) mod 12] + (i==j)*MDIAG[i]. */
static const u64 POSEIDON_C[360] = {
#include "poseidon_constants.inc"
};
static const u64 MCIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline u64 pow7(u64 a)
{
    u64 a2 = gl_mul(a, a), a4 = gl_mul(a2, a2), a3 = gl_mul(a, a2);
    return gl_mul(a3, a4);
}

static void poseidon_perm(u64 st[12])
{
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], POSEIDON_C[12 * r + i]);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        } else {
            st[0] = pow7(st[0]);
        }
        u64 acc[12];
        for (int x = 0; x < 12; x++) {
            u64 a = 0;
            for (int y = 0; y < 12; y++) {
                u64 m = MCIRC[(y - x + 12) % 12] + (x == y ? MDIAG[x] : 0);
                a = gl_add(a, gl_mul(st[y], m));
            }
            acc[x] = a;
        }
        memcpy(st, acc, sizeof(acc));
    }
}

/* PoseidonGoldilocks::hash_full_result(out12,in12): transcript.cpp:23,46 */
void orc_poseidon_perm(u64 out[12], const u64 in[12])
{
    u64 st[12];
    for (int i = 0; i < 12; i++) st[i] = gl_canon(in[i]);
    poseidon_perm(st);
    memcpy(out, st, sizeof(st));
}

/* linear_hash: leaf rule of PoseidonGoldilocks::merkletree* (called at
 * merkleTreeGL.cpp:39-43): n==0 -> zeros; n<=4 -> copy,
 * zero-pad, no permutation; else sponge rate 8 / capacity 4, capacity starts
 * at 0^4, last block zero-padded, output = state[0..3]. */
void orc_linear_hash(u64 out[4], const u64 *in, u64 n)
{
    if (n <= 4) {
        for (u64 i = 0; i < 4; i++) out[i] = i < n ? gl_canon(in[i]) : 0;
        return;
    }
    u64 st[12];
    u64 cap[4] = {0, 0, 0, 0};
    u64 rem = n;
    while (rem) {
        u64 take = rem < 8 ? rem : 8;
        for (u64 i = 0; i < 8; i++) st[i] = i < take ? gl_canon(in[i]) : 0;
        memcpy(st + 8, cap, sizeof(cap));
        poseidon_perm(st);
        memcpy(cap, st, sizeof(cap));
        in += take;
        rem -= take;
    }
    memcpy(out, cap, sizeof(cap));
}

/* PoseidonGoldilocks::merkletree_avx(tree,input,ncols,nrows)
 * (merkleTreeGL.cpp:37-44): level 0 = linear_hash of each row, then binary
 * levels node = perm(left|right|0^4)[0..3]; buffer = level0 | level1 | ... |
 * root, (2h-1)*4 elements (merkleTreeGL.hpp:61-68). */
void orc_merkletree(u64 *nodes, const u64 *src, u64 w, u64 h)
{
#pragma omp parallel for schedule(static)
    for (u64 i = 0; i < h; i++) orc_linear_hash(nodes + 4 * i, src + i * w, w);
    u64 off = 0, cnt = h;
    while (cnt > 1) {
        u64 *cur = nodes + off, *nxt = nodes + off + 4 * cnt;
#pragma omp parallel for schedule(static)
        for (u64 j = 0; j < cnt / 2; j++) {
            u64 st[12];
            memcpy(st, cur + 8 * j, 8 * sizeof(u64));
            st[8] = st[9] = st[10] = st[11] = 0;
            poseidon_perm(st);
            memcpy(nxt + 4 * j, st, 4 * sizeof(u64));
        }
        off += 4 * cnt;
        cnt /= 2;
    }
}

/* MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35): proof = row ||
 * siblings bottom-up, sibling at level k = nodes[off_k + ((idx>>k)^1)*4]. */
void orc_merkle_proof(u64 *proof, const u64 *nodes, const u64 *src, u64 w, u64 h, u64 idx)
{
    memcpy(proof, src + idx * w, w * sizeof(u64));
    proof += w;
    u64 off = 0, cnt = h;
    while (cnt > 1) {
        memcpy(proof, nodes + off + (idx ^ 1) * 4, 4 * sizeof(u64));
        proof += 4;
        off += 4 * cnt;
        cnt /= 2;
        idx >>= 1;
    }
}

/* Verifier-side fold of a Merkle path (used only to pin the oracle against the
 * golden proofs): returns the root implied by leaf `row[0..w)` at `idx`. */
void orc_merkle_root_from_path(u64 root[4], const u64 *row, u64 w, const u64 *sib, u64 nsib, u64 idx)
{
    u64 cur[4];
    orc_linear_hash(cur, row, w);
    for (u64 k = 0; k < nsib; k++) {
        u64 st[12];
        if (idx & 1) { memcpy(st, sib + 4 * k, 32); memcpy(st + 4, cur, 32); }
        else         { memcpy(st, cur, 32); memcpy(st + 4, sib + 4 * k, 32); }
        st[8] = st[9] = st[10] = st[11] = 0;
        poseidon_perm(st);
        memcpy(cur, st, 32);
        idx >>= 1;
    }
    memcpy(root, cur, 32);
}

/* DFS over the 2^nsib left/right choices of one path until it reproduces
 * `root`; returns the leaf index or (u64)-1.  SURVEY.md Appendix B step 3:
 * how a query index is recovered from a golden proof without the verkey. */
static int dfs_idx(const u64 cur[4], const u64 *sib, u64 k, u64 nsib, const u64 root[4], u64 acc, u64 *found)
{
    if (k == nsib) {
        if (memcmp(cur, root, 32) == 0) { *found = acc; return 1; }
        return 0;
    }
    for (u64 bit = 0; bit < 2; bit++) {
        u64 st[12];
        if (bit) { memcpy(st, sib + 4 * k, 32); memcpy(st + 4, cur, 32); }
        else     { memcpy(st, cur, 32); memcpy(st + 4, sib + 4 * k, 32); }
        st[8] = st[9] = st[10] = st[11] = 0;
        poseidon_perm(st);
        if (dfs_idx(st, sib, k + 1, nsib, root, acc | (bit << k), found)) return 1;
    }
    return 0;
}
u64 orc_find_merkle_index(const u64 *row, u64 w, const u64 *sib, u64 nsib, const u64 root[4])
{
    u64 leaf[4], found = (u64)-1;
    orc_linear_hash(leaf, row, w);
    dfs_idx(leaf, sib, 0, nsib, root, 0, &found);
    return found;
}

/* --------------------------------------------------------------------- NTT */
static unsigned ilog2(u64 n) { unsigned l = 0; while ((1ULL << l) < n) l++; return l; }
static inline u64 bitrev(u64 x, unsigned bits)
{
    u64 r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* One column, in place, natural in / natural out, iterative radix-2 DIT.
 * X[k] = sum_j x[j] w^(jk), w = w(log2 n) (forward) or its inverse, inverse
 * also scales by 1/n.  Semantics of NTT_Goldilocks::NTT/INTT as used at
 * starks.cpp:262,285,326-327 and friProve.cpp:100-102; an independent scalar
 * statement of the same transform is tools/starkpil/bctree/build_const_tree.cpp:242-288. */
static void ntt_col(u64 *a, u64 n, int inverse)
{
    unsigned lg = ilog2(n);
    for (u64 i = 0; i < n; i++) {
        u64 j = bitrev(i, lg);
        if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 1; s <= lg; s++) {
        u64 m = 1ULL << s, h = m >> 1;
        u64 wm = orc_w(s);
        if (inverse) wm = gl_inv(wm);
        for (u64 k = 0; k < n; k += m) {
            u64 w = 1;
            for (u64 j = 0; j < h; j++) {
                u64 t = gl_mul(w, a[k + j + h]);
                u64 u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + h] = gl_sub(u, t);
                w = gl_mul(w, wm);
            }
        }
    }
    if (inverse) {
        u64 ninv = gl_inv(n % GL_P);
        for (u64 i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
    }
}

/* Row-major [n][ncols] matrices, one transform per column (dst may alias src:
 * `ntt.INTT(LEv, LEv, N, 3)` starks.cpp:326). */
void orc_ntt(u64 *dst, const u64 *src, u64 n, u64 ncols, int inverse)
{
    if (n == 0 || ncols == 0) return;
#pragma omp parallel
    {
        u64 *col = (u64 *)malloc(n * sizeof(u64));
#pragma omp for schedule(dynamic, 1)
        for (u64 c = 0; c < ncols; c++) {
            for (u64 i = 0; i < n; i++) col[i] = gl_canon(src[i * ncols + c]);
            ntt_col(col, n, inverse);
            /* all reads of column c precede its writes; other columns untouched */
            for (u64 i = 0; i < n; i++) dst[i * ncols + c] = col[i];
        }
        free(col);
    }
}

/* Naive O(n^2) DFT of one column: the independent check for ntt_col. */
void orc_dft_naive(u64 *dst, const u64 *src, u64 n, int inverse)
{
    u64 w = orc_w(ilog2(n));
    if (inverse) w = gl_inv(w);
    u64 ninv = inverse ? gl_inv(n % GL_P) : 1;
    for (u64 k = 0; k < n; k++) {
        u64 wk = gl_pow(w, k), acc = 0, x = 1;
        for (u64 j = 0; j < n; j++) { acc = gl_add(acc, gl_mul(src[j], x)); x = gl_mul(x, wk); }
        dst[k] = gl_mul(acc, ninv);
    }
}

/* NTT_Goldilocks::extendPol(out,in,N_ext,N,ncols,buf) (starks.cpp:53,134,215):
 * coeff = INTT_N(col); coeff[i] *= shift^i; zero-pad to N_ext; out = NTT_Next.
 * => out[k][c] = P_c(shift * w(log2 N_ext)^k).  Scalar restatement in the
 * reference: build_const_tree.cpp:360-396 (x shift^i / N) and :398-531. */
void orc_extend_pol(u64 *out, const u64 *in, u64 n_ext, u64 n, u64 ncols)
{
    if (n == 0 || ncols == 0) return;
#pragma omp parallel
    {
        u64 *col = (u64 *)malloc(n_ext * sizeof(u64));
#pragma omp for schedule(dynamic, 1)
        for (u64 c = 0; c < ncols; c++) {
            for (u64 i = 0; i < n; i++) col[i] = gl_canon(in[i * ncols + c]);
            ntt_col(col, n, 1);
            u64 s = 1;
            for (u64 i = 0; i < n; i++) { col[i] = gl_mul(col[i], s); s = gl_mul(s, GL_SHIFT); }
            for (u64 i = n; i < n_ext; i++) col[i] = 0;
            ntt_col(col, n_ext, 0);
            for (u64 i = 0; i < n_ext; i++) out[i * ncols + c] = col[i];
        }
        free(col);
    }
}

/* The same extension with EVERY host thread at work (bench.py's cpu_baseline on the GPU box's cores): orc_extend_pol above runs one thread
 * per column, which leaves 120 of 128 cores idle on an 8-column trace.  Here all columns are transformed together by the whole team:
 * the first 16 stages of a transform stay inside blocks of 2^16 points (one block = one task, in cache), the remaining stages run over
 * all butterflies of all columns; twiddles come from a table w(log2 n)^i built in parallel chunks.  Same arithmetic, same results as
 * orc_extend_pol (tests/test_oracle_golden.py compares the two). */
/* The team transform multiplies with the special-form reduction (2^64 = 2^32 - 1, 2^96 = -1 mod p) instead of gl_mul's `% p` on 128 bits (a library
 * call of ~40 ns): what any CPU implementation of this field does, and what makes the cpu_baseline of bench.py a fair scalar figure (round 5: the `%` form
 * was ~10x off a competent scalar transform).  Only here: every other function of the oracle keeps the plain `%`, and tests/test_oracle_golden.py
 * compares this transform with orc_extend_pol word for word. */
static inline u64 gl_mul_fast(u64 a, u64 b)
{
    const u128 x = (u128)a * (u128)b;
    const u64 lo = (u64)x, hi = (u64)(x >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    u64 t0, r;
    if (__builtin_sub_overflow(lo, hh, &t0)) t0 -= 0xFFFFFFFFULL;     /* lo - hh < 0: + p = - EPS (mod 2^64), no second borrow */
    if (__builtin_add_overflow(t0, hl * 0xFFFFFFFFULL, &r)) r += 0xFFFFFFFFULL;   /* carried: 2^64 = EPS, no second carry */
    return r >= GL_P ? r - GL_P : r;
}
u64 orc_mul_fast(u64 a, u64 b) { return gl_mul_fast(a, b); }        /* (exported for the test that pins it to gl_mul on the reduction's edge cases) */
#define TEAM_BLOCK_LOG 16
static void ntt_cols_team(u64 *a /* [ncols][n] */, u64 ncols, u64 n, int inverse, const u64 *tw /* w(log2 n)^(+-i), i < n/2 */)
{
    const unsigned lg = ilog2(n);
    const unsigned lb = lg < TEAM_BLOCK_LOG ? lg : TEAM_BLOCK_LOG;
    const u64 nblk = n >> lb;
#pragma omp for schedule(static)
    for (u64 t = 0; t < ncols * n; t++) {
        const u64 c = t >> lg, i = t & (n - 1), j = bitrev(i, lg);        /* (n is a power of two: shifts, not the divisions `t / n` compiles to) */
        if (i < j) { u64 *col = a + c * n; u64 x = col[i]; col[i] = col[j]; col[j] = x; }
    }
#pragma omp for schedule(dynamic, 1)
    for (u64 t = 0; t < ncols * nblk; t++) {
        u64 *b = a + (t >> (lg - lb)) * n + ((t & (nblk - 1)) << lb);
        for (unsigned s = 1; s <= lb; s++) {
            const u64 h = 1ULL << (s - 1), step = n >> s;
            for (u64 k = 0; k < (1ULL << lb); k += 2 * h)
                for (u64 j = 0; j < h; j++) {
                    const u64 x = gl_mul_fast(tw[j * step], b[k + j + h]), u = b[k + j];
                    b[k + j] = gl_add(u, x);
                    b[k + j + h] = gl_sub(u, x);
                }
        }
    }
    for (unsigned s = lb + 1; s <= lg; s++) {
        const u64 h = 1ULL << (s - 1), step = n >> s;
#pragma omp for schedule(static)
        for (u64 t = 0; t < ncols * (n / 2); t++) {
            u64 *col = a + (t >> (lg - 1)) * n;
            const u64 i = t & (n / 2 - 1), j = i & (h - 1), k = (i >> (s - 1)) << s;
            const u64 x = gl_mul_fast(tw[j * step], col[k + j + h]), u = col[k + j];
            col[k + j] = gl_add(u, x);
            col[k + j + h] = gl_sub(u, x);
        }
    }
    if (inverse) {
        const u64 ninv = gl_inv(n % GL_P);
#pragma omp for schedule(static)
        for (u64 t = 0; t < ncols * n; t++) a[t] = gl_mul_fast(a[t], ninv);
    }
}
static void pow_table_team(u64 *t, u64 count, u64 base)       /* t[i] = base^i */
{
#pragma omp for schedule(static)
    for (u64 c = 0; c < (count + 4095) / 4096; c++) {
        u64 v = gl_pow(base, c * 4096);
        for (u64 i = c * 4096; i < count && i < (c + 1) * 4096; i++) { t[i] = v; v = gl_mul_fast(v, base); }
    }
}
void orc_extend_pol_team(u64 *out, const u64 *in, u64 n_ext, u64 n, u64 ncols)
{
    if (n == 0 || ncols == 0) return;
    u64 *cols = (u64 *)malloc(n_ext * ncols * sizeof(u64));           /* [ncols][n_ext]: the base transform uses the first n words of a column */
    u64 *twi = (u64 *)malloc((n / 2 + 1) * sizeof(u64)), *twf = (u64 *)malloc((n_ext / 2 + 1) * sizeof(u64)), *sh = (u64 *)malloc(n * sizeof(u64));
    u64 *base = (u64 *)malloc(n * ncols * sizeof(u64));               /* [ncols][n] */
#pragma omp parallel
    {
        pow_table_team(twi, n / 2, gl_inv(orc_w(ilog2(n))));
        pow_table_team(twf, n_ext / 2, orc_w(ilog2(n_ext)));
        pow_table_team(sh, n, GL_SHIFT);
#pragma omp for schedule(static)
        for (u64 i = 0; i < n; i++)
            for (u64 c = 0; c < ncols; c++) base[c * n + i] = gl_canon(in[i * ncols + c]);
        ntt_cols_team(base, ncols, n, 1, twi);
#pragma omp for schedule(static)
        for (u64 t = 0; t < ncols * n_ext; t++) {
            const u64 c = t >> ilog2(n_ext), i = t & (n_ext - 1);
            cols[t] = i < n ? gl_mul_fast(base[c * n + i], sh[i]) : 0;
        }
        ntt_cols_team(cols, ncols, n_ext, 0, twf);
#pragma omp for schedule(static)
        for (u64 i = 0; i < n_ext; i++)
            for (u64 c = 0; c < ncols; c++) out[i * ncols + c] = cols[c * n_ext + i];
    }
    free(cols); free(twi); free(twf); free(sh); free(base);
}

/* Naive evaluation: out[k] = P(shift*w_ext^k) by Horner over INTT coefficients
 * computed with the naive DFT.  Independent check of orc_extend_pol. */
void orc_extend_naive(u64 *out, const u64 *in, u64 n_ext, u64 n)
{
    u64 *coef = (u64 *)malloc(n * sizeof(u64));
    orc_dft_naive(coef, in, n, 1);
    u64 wext = orc_w(ilog2(n_ext));
    for (u64 k = 0; k < n_ext; k++) {
        u64 x = gl_mul(GL_SHIFT, gl_pow(wext, k)), acc = 0;
        for (u64 i = n; i-- > 0;) acc = gl_add(gl_mul(acc, x), coef[i]);
        out[k] = acc;
    }
    free(coef);
}

/* P(z) of the interpolant of evals[i * stride] on the coset {shift * w(log2 n)^i}, z outside it, by the BARYCENTRIC formula
 *     P(z) = (z^n - shift^n) / (n shift^n) * sum_i f_i x_i / (z - x_i),      x_i = shift w^i
 * (Lagrange with Z(x) = x^n - shift^n, Z'(x_i) = n shift^n / x_i).  No transform of any kind: an INDEPENDENT check of extendPol's values at any size
 * (tests: the input column on <w_N> and the output column on shift <w_Next> must give the same P(z) at random z -- two polynomials of degree < N that
 * agree at a random point of a 2^64 field).  The inverses by Montgomery's trick in chunks; OpenMP over chunks. */
u64 orc_barycentric_eval(const u64 *evals, u64 n, u64 stride, u64 shift, u64 z)
{
    const u64 w = orc_w(ilog2(n));
    const u64 CH = 4096, nch = (n + CH - 1) / CH;
    u64 total = 0;
#pragma omp parallel
    {
        u64 acc = 0;
        u64 *d = (u64 *)malloc(CH * sizeof(u64)), *pre = (u64 *)malloc(CH * sizeof(u64)), *x = (u64 *)malloc(CH * sizeof(u64));
#pragma omp for schedule(static)
        for (u64 c = 0; c < nch; c++) {
            const u64 i0 = c * CH, m = n - i0 < CH ? n - i0 : CH;
            u64 xi = gl_mul(shift, gl_pow(w, i0)), run = 1;
            for (u64 k = 0; k < m; k++) {
                x[k] = xi; d[k] = gl_sub(z, xi);              /* z is not on the coset: d != 0 */
                pre[k] = run; run = gl_mul(run, d[k]);
                xi = gl_mul(xi, w);
            }
            u64 inv = gl_inv(run);
            for (u64 k = m; k-- > 0;) {
                const u64 dinv = gl_mul(inv, pre[k]);          /* 1 / (z - x_k) */
                inv = gl_mul(inv, d[k]);
                acc = gl_add(acc, gl_mul(gl_mul(gl_canon(evals[(i0 + k) * stride]), x[k]), dinv));
            }
        }
#pragma omp critical
        total = gl_add(total, acc);
        free(d); free(pre); free(x);
    }
    const u64 sn = gl_pow(shift, n);
    const u64 lead = gl_mul(gl_sub(gl_pow(z, n), sn), gl_inv(gl_mul(n % GL_P, sn)));
    return gl_mul(lead, total);
}

/* -------------------------------------------------------------- Transcript */
/* src/starkpil/transcript/transcript.cpp:4-87, transcript.hpp:13-37. */
typedef struct {
    u64 state[4];
    u64 pending[8];
    u64 out[12];
    u64 pending_cursor;
    u64 out_cursor;
} orc_transcript;

void orc_transcript_init(orc_transcript *t) { memset(t, 0, sizeof(*t)); }
static void transcript_absorb(orc_transcript *t)
{
    u64 in[12];
    memcpy(in, t->pending, 64);
    memcpy(in + 8, t->state, 32);
    orc_poseidon_perm(t->out, in);
    t->out_cursor = 12;
    memset(t->pending, 0, 64);
    t->pending_cursor = 0;
    memcpy(t->state, t->out, 32);
}
void orc_transcript_put(orc_transcript *t, const u64 *in, u64 n)
{
    for (u64 i = 0; i < n; i++) {
        t->pending[t->pending_cursor++] = in[i];
        t->out_cursor = 0;
        if (t->pending_cursor == 8) transcript_absorb(t);
    }
}
u64 orc_transcript_get1(orc_transcript *t)
{
    if (t->out_cursor == 0) transcript_absorb(t);
    u64 r = t->out[(12 - t->out_cursor) % 12];
    t->out_cursor--;
    return r;
}
void orc_transcript_get_field(orc_transcript *t, u64 out[3])
{
    for (int i = 0; i < 3; i++) out[i] = orc_transcript_get1(t);
}
/* transcript.cpp:57-87: 63 bits per field element, LSB first. */
void orc_transcript_get_permutations(orc_transcript *t, u64 *res, u64 n, u64 nbits)
{
    u64 total = n * nbits;
    u64 nfields = (total - 1) / 63 + 1;
    u64 *f = (u64 *)malloc(nfields * sizeof(u64));
    for (u64 i = 0; i < nfields; i++) f[i] = orc_transcript_get1(t);
    u64 cf = 0, cb = 0;
    for (u64 i = 0; i < n; i++) {
        u64 a = 0;
        for (u64 j = 0; j < nbits; j++) {
            u64 bit = (gl_canon(f[cf]) >> cb) & 1;
            if (bit) a += (1ULL << j);
            if (++cb == 63) { cb = 0; cf++; }
        }
        res[i] = a;
    }
    free(f);
}

/* --------------------------------------------------------------------- FRI */
/* One folding step of FRIProve::prove, friProve.cpp:44-108 (+polMulAxi :192,
 * evalPol :201).  pol: 2^bits ext-3 elements; pol2: 2^(bits-red).
 *   g-th output = Horner_x( INTT_nX({pol[i*M+g]}_i) (.) (sinv*w(bits)^-g)^i ),
 * M = 2^(bits-red), nX = 2^red.  red == 0 is the copy at :87-90. */
void orc_fri_fold(u64 *pol2, const u64 *pol, unsigned bits, unsigned red, const u64 x[3], u64 shift_inv)
{
    u64 M = 1ULL << (bits - red), nX = 1ULL << red;
    if (red == 0) { memcpy(pol2, pol, M * 3 * sizeof(u64)); return; }
    u64 wi = gl_inv(orc_w(bits));
#pragma omp parallel
    {
        u64 *col = (u64 *)malloc(nX * 3 * sizeof(u64));
        u64 *tmp = (u64 *)malloc(nX * sizeof(u64));
#pragma omp for schedule(static)
        for (u64 g = 0; g < M; g++) {
            for (u64 i = 0; i < nX; i++) memcpy(col + 3 * i, pol + 3 * (i * M + g), 24);
            for (int d = 0; d < 3; d++) {
                for (u64 i = 0; i < nX; i++) tmp[i] = gl_canon(col[3 * i + d]);
                ntt_col(tmp, nX, 1);
                for (u64 i = 0; i < nX; i++) col[3 * i + d] = tmp[i];
            }
            u64 acc = gl_mul(shift_inv, gl_pow(wi, g)), r = 1;
            for (u64 i = 0; i < nX; i++) {
                for (int d = 0; d < 3; d++) col[3 * i + d] = gl_mul(col[3 * i + d], r);
                r = gl_mul(r, acc);
            }
            u64 res[3];
            memcpy(res, col + 3 * (nX - 1), 24);
            for (u64 i = nX - 1; i-- > 0;) {
                u64 t[3];
                gl3_mul(t, res, x);
                for (int d = 0; d < 3; d++) res[d] = gl_add(t[d], col[3 * i + d]);
            }
            memcpy(pol2 + 3 * g, res, 24);
        }
        free(col);
        free(tmp);
    }
}

/* FRIProve::getTransposed, friProve.cpp:252-271: w = 2^tbits, h = n/w,
 * dst[i*h + j] = src[j*w + i], ext-3 elements. */
void orc_transpose3(u64 *dst, const u64 *src, u64 n, unsigned tbits)
{
    u64 w = 1ULL << tbits, h = n / w;
    for (u64 i = 0; i < w; i++)
        for (u64 j = 0; j < h; j++) memcpy(dst + 3 * (i * h + j), src + 3 * (j * w + i), 24);
}

/* Verifier-side fold of one FRI query (pins orc_fri_fold's conventions against
 * the golden proofs): vals = the nX ext-3 values of tree row `g`, laid out as
 * the reference's transposed group; returns the folded value at index g. */
void orc_fri_fold_one(u64 out[3], const u64 *vals, unsigned bits, unsigned red, u64 g, const u64 x[3], u64 shift_inv)
{
    u64 nX = 1ULL << red;
    u64 *col = (u64 *)malloc(nX * 3 * sizeof(u64));
    u64 *tmp = (u64 *)malloc(nX * sizeof(u64));
    memcpy(col, vals, nX * 24);
    for (int d = 0; d < 3; d++) {
        for (u64 i = 0; i < nX; i++) tmp[i] = gl_canon(col[3 * i + d]);
        ntt_col(tmp, nX, 1);
        for (u64 i = 0; i < nX; i++) col[3 * i + d] = tmp[i];
    }
    u64 acc = gl_mul(shift_inv, gl_pow(gl_inv(orc_w(bits)), g)), r = 1;
    for (u64 i = 0; i < nX; i++) {
        for (int d = 0; d < 3; d++) col[3 * i + d] = gl_mul(col[3 * i + d], r);
        r = gl_mul(r, acc);
    }
    u64 res[3];
    memcpy(res, col + 3 * (nX - 1), 24);
    for (u64 i = nX - 1; i-- > 0;) {
        u64 t[3];
        gl3_mul(t, res, x);
        for (int d = 0; d < 3; d++) res[d] = gl_add(t[d], col[3 * i + d]);
    }
    memcpy(out, res, 24);
    free(col);
    free(tmp);
}

/* Synthetic trace generator shared by tests and bench:
 * x[i] = splitmix64(seed ^ i) mod p. */
static inline u64 splitmix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_fill(u64 *dst, u64 n, u64 seed, u64 start)
{
#pragma omp parallel for schedule(static)
    for (u64 i = 0; i < n; i++) dst[i] = splitmix64(seed ^ (start + i)) % GL_P;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------ "next" tier */
/* Inverse in Fp[x]/(x^3 - x - 1): adjugate of the multiplication matrix
 * [[a0,a2,a1],[a1,a0+a2,a1+a2],[a2,a1,a0+a2]] (columns a*1, a*x, a*x^2) over its determinant.
 * Goldilocks3::inv as used by Polinomial::batchInverse (polinomial.hpp:698-720); inv(0) = 0. */
void orc_inv3(u64 o[3], const u64 a_[3])
{
    u64 a0 = gl_canon(a_[0]), a1 = gl_canon(a_[1]), a2 = gl_canon(a_[2]);
    u64 m00 = a0, m01 = a2, m02 = a1;
    u64 m10 = a1, m11 = gl_add(a0, a2), m12 = gl_add(a1, a2);
    u64 m20 = a2, m21 = a1, m22 = gl_add(a0, a2);
    /* cofactors of the first row -> first column of the inverse (solves M b = e0) */
    u64 c00 = gl_sub(gl_mul(m11, m22), gl_mul(m12, m21));
    u64 c01 = gl_sub(gl_mul(m12, m20), gl_mul(m10, m22));   /* = -(m10 m22 - m12 m20) */
    u64 c02 = gl_sub(gl_mul(m10, m21), gl_mul(m11, m20));
    u64 det = gl_add(gl_add(gl_mul(m00, c00), gl_mul(m01, c01)), gl_mul(m02, c02));
    u64 di = gl_inv(det);
    /* b = adj(M) e0 / det ; adj(M)[i][0] = cofactor C[0][i] */
    o[0] = gl_mul(c00, di); o[1] = gl_mul(c01, di); o[2] = gl_mul(c02, di);
}

/* Polinomial::batchInverse / batchInverseParallel (polinomial.hpp:635-720) for dim 3 (and dim 1):
 * res[i] = 1 / src[i].  (The reference batches with Montgomery's trick; for non-zero inputs the
 * results are the exact inverses, which is what is restated.  A zero input maps to zero.) */
void orc_batch_inverse(u64 *res, const u64 *src, u64 n, unsigned dim)
{
#pragma omp parallel for schedule(static)
    for (u64 i = 0; i < n; i++) {
        if (dim == 3) orc_inv3(res + 3 * i, src + 3 * i);
        else res[i] = gl_inv(gl_canon(src[i]));
    }
}

/* Polinomial::calculateZ (polinomial.hpp:586-607): z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1]. */
void orc_calculate_z(u64 *z, const u64 *num, const u64 *den, u64 n)
{
    if (!n) return;
    z[0] = 1; z[1] = 0; z[2] = 0;
    for (u64 i = 1; i < n; i++) {
        u64 di[3], t[3];
        orc_inv3(di, den + 3 * (i - 1));
        gl3_mul(t, num + 3 * (i - 1), di);
        gl3_mul(z + 3 * i, z + 3 * (i - 1), t);
    }
}

/* Starks::evmap (starks.cpp:556-669): evals[i] = sum_k L_i[k] * pol_i[(k << extend_bits) * stride_i + 0..dim_i),
 * L_i = LpEv if prime_i else LEv (ext-3, N rows); mulAddElement_adim3 (polinomial.hpp:722-744). */
void orc_evmap(u64 *evals, const u64 *LEv, const u64 *LpEv, u64 n, unsigned extend_bits, unsigned nev,
               const u64 *const *pols, const u64 *strides, const unsigned *dims, const unsigned *primes)
{
    for (unsigned i = 0; i < nev; i++) {
        u64 acc[3] = {0, 0, 0};
        const u64 *L = primes[i] ? LpEv : LEv;
        for (u64 k = 0; k < n; k++) {
            const u64 *b = pols[i] + (k << extend_bits) * strides[i];
            const u64 *a = L + 3 * k;
            if (dims[i] == 1) {
                for (int d = 0; d < 3; d++) acc[d] = gl_add(acc[d], gl_mul(a[d], b[0]));
            } else {
                u64 t[3];
                gl3_mul(t, a, b);
                for (int d = 0; d < 3; d++) acc[d] = gl_add(acc[d], t[d]);
            }
        }
        memcpy(evals + 3 * i, acc, 24);
    }
}

/* Polinomial::calculateH1H2_opt1 / _opt3 (polinomial.hpp:349-585), dim = 1 or 3.
 * Every value of f must occur in t.  Each t index starts with count 1; every f element adds 1 to the
 * LAST index of t holding its value (the reference's hash chains overwrite the stored index, :378,:496);
 * the sequence "t[id] repeated count[id] times, id ascending" (length 2n) is dealt alternately to h1, h2.
 * Returns 0, or 1 if some f value is missing from t (the reference calls exitProcess()). */
typedef struct { u64 k[3]; u64 idx; } h1h2_ent;
static int h1h2_cmp(const void *a, const void *b)
{
    const h1h2_ent *x = (const h1h2_ent *)a, *y = (const h1h2_ent *)b;
    for (int d = 0; d < 3; d++) if (x->k[d] != y->k[d]) return x->k[d] < y->k[d] ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
int orc_calculate_h1h2(u64 *h1, u64 *h2, const u64 *f, const u64 *t, u64 n, unsigned dim)
{
    h1h2_ent *e = (h1h2_ent *)malloc(n * sizeof(h1h2_ent));
    u64 *cnt = (u64 *)malloc(n * sizeof(u64));
    for (u64 i = 0; i < n; i++) {
        for (unsigned d = 0; d < 3; d++) e[i].k[d] = d < dim ? gl_canon(t[i * dim + d]) : 0;
        e[i].idx = i;
        cnt[i] = 1;
    }
    qsort(e, n, sizeof(h1h2_ent), h1h2_cmp);
    /* last index per distinct key = idx of the last entry of each run (sorted by idx within a key) */
    int missing = 0;
    for (u64 j = 0; j < n && !missing; j++) {
        h1h2_ent q;
        for (unsigned d = 0; d < 3; d++) q.k[d] = d < dim ? gl_canon(f[j * dim + d]) : 0;
        u64 lo = 0, hi = n;      /* upper bound of the key run */
        q.idx = (u64)-1;
        while (lo < hi) { u64 m = (lo + hi) / 2; if (h1h2_cmp(&e[m], &q) <= 0) lo = m + 1; else hi = m; }
        if (lo == 0 || memcmp(e[lo - 1].k, q.k, sizeof(q.k)) != 0) { missing = 1; break; }
        cnt[e[lo - 1].idx]++;
    }
    if (!missing) {
        u64 id = 0;
        for (u64 i = 0; i < n; i++) {
            if (cnt[id] == 0) id++;
            cnt[id]--;
            for (unsigned d = 0; d < dim; d++) h1[i * dim + d] = gl_canon(t[id * dim + d]);
            if (cnt[id] == 0) id++;
            cnt[id]--;
            for (unsigned d = 0; d < dim; d++) h2[i * dim + d] = gl_canon(t[id * dim + d]);
        }
    }
    free(e); free(cnt);
    return missing;
}

/* ---------------------------------------------------------------------------------------------
 * Constraint-expression programs: CPU restatement of what one pass of
 * ZkevmSteps::step*_parser_first_avx does per row (zkevm.chelpers.step3.parser.cpp:11-973 and
 * siblings), on the orthogonal instruction encoding documented in include/xlp_gl.h.  Operand
 * addressing follows the reference's expressions:
 *   pols[args + i * args]                                   (step3.parser.cpp:33, case 1)
 *   pols[args + (((i + j) + args) % args) * args]            (:52-55, case 5: "prime" rows)
 *   pConstPols->getElement(col, i) = const[col + i * numPols] (:45, case 3)
 *   challenges[k] (3 words), publicInputs[k], evals[k] (3 words), x_n[i], x_2ns[i],
 *   zi.zhInv(i) = ZHInv[i % size] (zhInv.hpp:20-23), xDivXSubXi[i], q_2ns[i*3], f_2ns[i*3].
 * Base (x) meets extension as (x,0,0): add13/sub13c/sub31c/mul13/mul1c3c of the reference.
 * Not runnable against the reference binary (the generated parsers need the absent goldilocks library and a
 * starkinfo/witness archive).  Pinned instead through the reference's two generated forms of the same steps: the
 * table form and the straight-line form of step2prev / step3prev / step52ns, converted independently, must give
 * identical outputs under this interpreter, and on the golden recursive1 / recursive2 proofs: the converted step52ns
 * program run on the opened query rows reproduces the committed FRI-polynomial values, the converted step42ns program
 * at xi satisfies C(xi) = Q(xi) Zh(xi) with the proofs' evaluations (tests/test_expr_vm.py). */
typedef struct {
    u64 *pols; u64 pols_size;
    const u64 *const_pols; u64 n_const_pols; u64 const_pols_size;
    const u64 *const_pols_2ns; u64 n_const_pols_2ns; u64 const_pols_2ns_size;
    const u64 *challenges; u64 challenges_size;
    const u64 *public_inputs; u64 public_inputs_size;
    const u64 *evals; u64 evals_size;
    const u64 *x_n; u64 x_n_size;
    const u64 *x_2ns; u64 x_2ns_size;
    const u64 *zi; u64 zi_size;
    const u64 *x_div_x_sub_xi; u64 x_div_x_sub_xi_size;
    const u64 *x_div_x_sub_wxi; u64 x_div_x_sub_wxi_size;
    u64 *q_2ns; u64 q_2ns_size;
    u64 *f_2ns; u64 f_2ns_size;
} orc_steps_params;

static const int ORC_KPARAMS[20] = {0, 1, 1, 2, 4, 1, 3, 1, 3, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};

/* address of element 0 of an operand for row i (NULL for temporaries / numbers) */
static u64 *orc_expr_addr(const orc_steps_params *P, unsigned kind, const u64 *a, u64 i)
{
    switch (kind) {
    case 3: return P->pols + a[0] + i * a[1];
    case 4: return P->pols + a[0] + ((i + a[1]) % a[2]) * a[3];
    case 5: return (u64 *)P->const_pols + a[0] + i * P->n_const_pols;
    case 6: return (u64 *)P->const_pols + a[0] + ((i + a[1]) % a[2]) * P->n_const_pols;
    case 7: return (u64 *)P->const_pols_2ns + a[0] + i * P->n_const_pols_2ns;
    case 8: return (u64 *)P->const_pols_2ns + a[0] + ((i + a[1]) % a[2]) * P->n_const_pols_2ns;
    case 10: return (u64 *)P->challenges + 3 * a[0];
    case 11: return (u64 *)P->public_inputs + a[0];
    case 12: return (u64 *)P->evals + 3 * a[0];
    case 13: return (u64 *)P->x_n + i;
    case 14: return (u64 *)P->x_2ns + i;
    case 15: return (u64 *)P->zi + (i % P->zi_size);
    case 16: return (u64 *)P->x_div_x_sub_xi + 3 * i;
    case 17: return (u64 *)P->x_div_x_sub_wxi + 3 * i;
    case 18: return P->q_2ns + 3 * i;
    case 19: return P->f_2ns + 3 * i;
    default: return NULL;
    }
}

/* returns 0, or 1 + index of the first malformed instruction */
u64 orc_expr_run(const u64 *code, u64 ncode, u64 ntmp1, u64 ntmp3, const orc_steps_params *P, u64 row0, u64 nrows)
{
    u64 *tmp1 = (u64 *)calloc(ntmp1 ? ntmp1 : 1, sizeof(u64));
    u64 *tmp3 = (u64 *)calloc(ntmp3 ? 3 * ntmp3 : 1, sizeof(u64));
    u64 bad = 0;
    for (u64 i = row0; i < row0 + nrows && !bad; i++) {
        u64 pc = 0, k = 0;
        while (pc < ncode) {
            const u64 h = code[pc];
            const unsigned op = h & 0xff, kd = (h >> 8) & 0xff, ka = (h >> 16) & 0xff, kb = (h >> 24) & 0xff;
            const unsigned dd = (h >> 32) & 3, da = (h >> 34) & 3, db = (h >> 36) & 3;
            if (op > 3 || kd >= 20 || ka >= 20 || kb >= 20) { bad = k + 1; break; }
            const u64 *pd = code + pc + 1, *pa = pd + ORC_KPARAMS[kd], *pb = pa + ORC_KPARAMS[ka];
            pc += 1 + ORC_KPARAMS[kd] + ORC_KPARAMS[ka] + ORC_KPARAMS[kb];
            if (pc > ncode) { bad = k + 1; break; }
            u64 a[3] = {0, 0, 0}, b[3] = {0, 0, 0}, r[3] = {0, 0, 0};
            const unsigned kinds[2] = {ka, kb}, dims[2] = {da, db};
            const u64 *prm[2] = {pa, pb};
            u64 *val[2] = {a, b};
            for (int s = 0; s < 2; s++) {
                if (kinds[s] == 0) continue;
                if (kinds[s] == 1) val[s][0] = tmp1[prm[s][0]];
                else if (kinds[s] == 2) memcpy(val[s], tmp3 + 3 * prm[s][0], 3 * sizeof(u64));
                else if (kinds[s] == 9) val[s][0] = gl_canon(prm[s][0]);
                else {
                    const u64 *src = orc_expr_addr(P, kinds[s], prm[s], i);
                    for (unsigned c = 0; c < dims[s]; c++) val[s][c] = gl_canon(src[c]);
                }
            }
            switch (op) {
            case 0: memcpy(r, a, sizeof(r)); break;
            case 1: for (int c = 0; c < 3; c++) r[c] = gl_add(a[c], b[c]); break;
            case 2: for (int c = 0; c < 3; c++) r[c] = gl_sub(a[c], b[c]); break;
            default:
                if (da == 3 && db == 3) gl3_mul(r, a, b);
                else if (da == 3) for (int c = 0; c < 3; c++) r[c] = gl_mul(a[c], b[0]);
                else if (db == 3) for (int c = 0; c < 3; c++) r[c] = gl_mul(b[c], a[0]);
                else r[0] = gl_mul(a[0], b[0]);
            }
            if (kd == 1) tmp1[pd[0]] = r[0];
            else if (kd == 2) memcpy(tmp3 + 3 * pd[0], r, 3 * sizeof(u64));
            else {
                u64 *dst = orc_expr_addr(P, kd, pd, i);
                if (!dst || !(kd == 3 || kd == 4 || kd == 18 || kd == 19)) { bad = k + 1; break; }
                for (unsigned c = 0; c < dd; c++) dst[c] = r[c];
            }
            k++;
        }
    }
    free(tmp1); free(tmp3);
    return bad;
}

/* genProof tables, restated as the reference computes them (sequential products):
 *   x_n / x_2ns / x   starks.hpp:149-183      out[0] = first, out[k] = out[k-1] * ratio
 *   LEv / LpEv        starks.cpp:318-322      (same, ext-3, before their INTT)
 *   xDivXSubXi        starks.cpp:340-362      x[k] * (x[k] - xi)^-1, element by element */
void orc_geometric(u64 *out, u64 n, unsigned dim, const u64 *first, const u64 *ratio)
{
    if (!n) return;
    u64 cur[3] = {0, 0, 0}, r[3] = {0, 0, 0};
    for (unsigned c = 0; c < dim; c++) { cur[c] = gl_canon(first[c]); r[c] = gl_canon(ratio[c]); }
    for (u64 k = 0; k < n; k++) {
        for (unsigned c = 0; c < dim; c++) out[k * dim + c] = cur[c];
        if (dim == 1) cur[0] = gl_mul(cur[0], r[0]);
        else { u64 t[3]; gl3_mul(t, cur, r); memcpy(cur, t, sizeof t); }
    }
}
void orc_x_div_x_sub_xi(u64 *out, const u64 *x, const u64 *xi, u64 n)
{
    for (u64 k = 0; k < n; k++) {
        const u64 xv = gl_canon(x[k]);
        u64 d[3] = {gl_sub(xv, xi[0]), gl_sub(0, xi[1]), gl_sub(0, xi[2])}, inv[3];
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) { out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = 0; continue; }
        orc_inv3(inv, d);
        for (int c = 0; c < 3; c++) out[3 * k + c] = gl_mul(inv[c], xv);
    }
}

/* starks.cpp:267-283 (the block loop over ii is only a work partition) */
void orc_q_split(u64 *qq2, const u64 *qq1, u64 n, u64 n_ext, unsigned qdeg, unsigned qdim, u64 shift_in)
{
    memset(qq2, 0, n_ext * qdeg * qdim * sizeof(u64));
    u64 curS = 1;
    for (unsigned p = 0; p < qdeg; p++) {
        for (u64 k = 0; k < n; k++)
            for (unsigned c = 0; c < qdim; c++) qq2[(k * qdeg + p) * qdim + c] = gl_mul(qq1[(p * n + k) * qdim + c], curS);
        curS = gl_mul(curS, shift_in);
    }
}
