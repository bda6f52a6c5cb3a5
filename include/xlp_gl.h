/*
 * xlp_gl.h -- C ABI of libxlp_gl.so: the MI355X (gfx950) replacement for the
 * Goldilocks field library calls made on the Starks::genProof hot path of
 * okx/xlayer-prover.  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - field elements are uint64_t; inputs may be any value in [0, 2^64)
 *     (canonical or not), outputs are canonical (< p = 2^64 - 2^32 + 1);
 *   - matrices are row-major [rows][ncols] exactly like the reference slab
 *     (src/starkpil/stark_info.cpp:473-482);
 *   - every function returns 0 on success, non-zero on error;
 *     xlp_last_error() then describes it.  The reference's functions return
 *     void and call exitProcess() on failure (src/utils/exit_process.cpp:7);
 *     the C++ shim (xlayer_prover_amd/host/) converts non-zero into that;
 *   - the caller owns every buffer; nothing is retained past return except
 *     caches owned by the process-wide context (twiddle tables, scratch);
 *   - there is NO CPU fallback: without a usable HIP device every call fails;
 *   - threading: every entry point may be called from any thread, concurrently.  The library keeps one process-wide context (stream,
 *     twiddle-table cache, grow-only scratch slots, the host-range device-copy cache) and ONE recursive lock that every entry point holds
 *     for its duration: callers are serialised, which costs nothing the GPU would have overlapped (all work goes to one stream).  This is
 *     what the reference needs: its proverThread issues the big calls one at a time (src/prover/prover.cpp:182-255), but FRIProve::prove
 *     calls NTT_Goldilocks(nX, 1).INTT from inside an OpenMP region (src/starkpil/fri/friProve.cpp:44,100).  xlp_last_error() is per
 *     thread.  tests/test_host_shim.py::test_host_boundary_is_reentrant_from_openmp_threads;
 *   - xlp_dev_* calls enqueue on the context's stream and return without waiting for the GPU, EXCEPT: a call whose scratch
 *     slot or table has to be created or grown (hipMalloc, and a stream synchronise when an old slot is freed: first call of a
 *     shape), xlp_dev_calculate_h1h2 (reads its error flag back), xlp_dev_evmap / xlp_dev_dot_columns (upload their entry
 *     tables), xlp_dev_expr_run with a plugin (none needed) or a first-time staging plan, xlp_dev_merkle_proofs and every
 *     xlp_copy_* (synchronous by definition).
 *
 * Two layers:
 *   xlp_*      host pointers  (drop-in for today's call sites: the data lives
 *              in the prover's host slab, src/prover/prover.cpp:94-115)
 *   xlp_dev_*  device pointers, asynchronous on the context's stream (for a
 *              GPU-resident genProof; used by bench.py and the tests)
 */
#ifndef XLP_GL_H
#define XLP_GL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- context ------------------------------------------------------------ */
/* Bind the library to HIP device `device` (creates the stream and caches).
 * Idempotent.  Replaces nothing in the reference (CPU library needs no init);
 * closest analogue: NTT_Goldilocks constructor building twiddles, starks.hpp:81-82. */
int xlp_init(int device);
/* Use an existing hipStream_t (e.g. torch's current stream) for all xlp_dev_* work. */
int xlp_set_stream(void* hip_stream);
int xlp_sync(void);
/* Two streams for a caller that overlaps exchanges (and the copies around them) with kernels -- host/starks.hpp, the row-sharded commit:
 * band k + 1's all-to-all runs beside band k's extendPol (the split of starks.cpp:53-60).  Stream 0 is the context's stream, stream 1 a second
 * one created on first use.  xlp_stream_select(s): every later xlp_dev_* / xlp_comm_* call enqueues on stream s.  xlp_stream_wait(a, b): work
 * enqueued on stream a from now on starts after everything enqueued on stream b so far.  xlp_sync with stream 0 selected waits for both. */
int xlp_stream_select(int which);
int xlp_stream_wait(int waiter, int signaller);
/* Finer ordering: xlp_stream_mark(s, m) records "everything enqueued on stream s so far" in mark m (0..7, reusable);
 * xlp_stream_wait_mark(s, m): later work of stream s waits for what mark m recorded (never recorded: no wait). */
int xlp_stream_mark(int stream, int mark);
int xlp_stream_wait_mark(int stream, int mark);
void xlp_shutdown(void);
const char* xlp_last_error(void);
/* 0xMMmmpp */
int xlp_version(void);

/* ---- host-pointer API (drop-in) ----------------------------------------- */
/* NTT_Goldilocks::NTT / ::INTT(dst, src, size, ncols, buffer, nphase, nblock)
 * -- starks.cpp:262,285,326-327; friProve.cpp:100-102.  Natural order in and
 * out; inverse != 0 applies w^-1 and the 1/n scaling.  dst may equal src. */
int xlp_ntt(uint64_t* dst, const uint64_t* src, uint64_t n, uint64_t ncols, int inverse);

/* NTT_Goldilocks::extendPol(out, in, N_ext, N, ncols, buffer) -- starks.cpp:53,134,215.
 * out[k][c] = P_c(shift * w(log2 n_ext)^k), shift = 7.  `scratch` is accepted
 * for signature parity and ignored by the host-pointer entry point. */
int xlp_extend(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ncols, uint64_t* scratch);

/* PoseidonGoldilocks::merkletree_avx / merkletree_avx512 / merkletree
 * (tree, input, num_cols, num_rows) -- merkleTreeGL.cpp:39-43, starks.cpp:688-692,
 * tools/starkpil/bctree/build_const_tree.cpp:582.
 * nodes: (2h-1)*4 elements = level0 | level1 | ... | root (merkleTreeGL.hpp:61-68). */
int xlp_merkle(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h);

/* PoseidonGoldilocks::hash_full_result(out12, in12) -- transcript.cpp:23,46. */
int xlp_perm(uint64_t out[12], const uint64_t in[12]);
/* Transcript::put of a long message (src/starkpil/transcript/transcript.cpp:12-29: every 8th element triggers _updateState = one
 * hash_full_result on pending[8] | state[4]): nblocks FULL blocks absorbed in one call.  out = the last permutation's 12 words (the transcript's
 * `out`, whose words 0..3 are its new `state`); block b = elems[8b .. 8b + 8) | capacity, capacity = state for b = 0, then words 0..3 of the
 * previous output.  Host pointers; one launch instead of nblocks round trips (739 for the evaluations of a zkEVM proof). */
int xlp_perm_chain(uint64_t out[12], const uint64_t state[4], const uint64_t* elems, uint64_t nblocks);
/* Batched form of the above: `count` independent 12-word states. */
int xlp_perm_batch(uint64_t* out, const uint64_t* in, uint64_t count);

/* PoseidonGoldilocks::linear_hash(out4, in, n) -- the leaf rule of merkletree;
 * here for `h` rows of width `w` at once (out: h*4). */
int xlp_linear_hash(uint64_t* out, const uint64_t* src, uint64_t w, uint64_t h);

/* One folding step of FRIProve::prove -- friProve.cpp:44-108 (+polMulAxi :192,
 * evalPol :201).  pol: 2^bits ext-3 elements; pol2: 2^(bits-red);
 * x: the challenge special_x; shift_inv: current polShiftInv (friProve.cpp:33,142-148). */
int xlp_fri_fold(uint64_t* pol2, const uint64_t* pol, uint32_t bits, uint32_t red, const uint64_t x[3], uint64_t shift_inv);

/* FRIProve::getTransposed(aux, pol, trasposeBits) -- friProve.cpp:252-271.
 * n ext-3 elements; dst[i*h + j] = src[j*w + i], w = 2^tbits, h = n / w. */
int xlp_transpose3(uint64_t* dst, const uint64_t* src, uint64_t n, uint32_t tbits);

/* Polinomial::batchInverse / calculateZ / calculateH1H2 on host buffers (dense: n elements of `dim` words, the
 * C++ mirror packs the reference's strided Polinomial views): same semantics as the xlp_dev_* forms below. */
int xlp_batch_inverse(uint64_t* res, const uint64_t* src, uint64_t n, uint32_t dim);
int xlp_calculate_z(uint64_t* z, const uint64_t* num, const uint64_t* den, uint64_t n);
int xlp_calculate_h1h2(uint64_t* h1, uint64_t* h2, const uint64_t* f, const uint64_t* t, uint64_t n, uint32_t dim);

/* Device copies of host buffers for the host-pointer API above (off by default; max_bytes = 0 turns it off and frees the
 * copies).  The drop-in call sites name the same slab sections in consecutive calls -- extendPol WRITES p_cm1_2ns, merkelize
 * READS it (starks.cpp:53-56) -- and staging each call separately would move the 100.8 GB extended section over PCIe three
 * times.  With the cache on, the device buffer a call produced or uploaded stays registered under its host address range
 * (least-recently-used eviction beyond max_bytes); a later call that reads exactly that range uses the device copy, a call
 * that writes into a range drops every copy overlapping it.  Contract: between two xlp_* calls the host does not modify a
 * range it handed to the library, or tells it so with xlp_cache_invalidate(ptr, bytes) (ptr NULL: everything).
 * xlp_extend / xlp_merkle on large inputs are pipelines either way: column bands (row chunks) cross PCIe on their own
 * streams while the previous band is extended (hashed); page-lock the slab (xlp_host_register) for the full PCIe rate. */
int xlp_cache_enable(size_t max_bytes);
int xlp_cache_invalidate(const void* host_ptr, size_t bytes);
size_t xlp_cache_bytes(void);

/* ---- device-pointer API -------------------------------------------------- */
/* Same semantics as above; pointers are device memory on the bound device;
 * calls enqueue on the context stream and return without synchronising. */
int xlp_dev_ntt(uint64_t* dst, const uint64_t* src, uint64_t n, uint64_t ncols, int inverse);
/* scratch: accepted for signature parity with extendPol's `buffer` (starks.cpp:53,134,215 pass a dead section).  The three-kernel
 * path (2^17 <= n <= 2^23, blow-up <= 8, ncols >= 4: every production call) ignores it and keeps a context-owned scratch of
 * (1 + n_ext / n) * n * min(ncols, 64) words -- 12.9 GB at n = 2^23, blow-up 2 -- alive between calls; the generic path uses
 * `scratch` when given (>= 2 * n * ncols words) and the context's otherwise. */
int xlp_dev_extend(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ncols, uint64_t* scratch);
int xlp_dev_merkle(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h);
int xlp_dev_perm_batch(uint64_t* out, const uint64_t* in, uint64_t count);
int xlp_dev_linear_hash(uint64_t* out, const uint64_t* src, uint64_t w, uint64_t h);
int xlp_dev_fri_fold(uint64_t* pol2, const uint64_t* pol, uint32_t bits, uint32_t red, const uint64_t x[3], uint64_t shift_inv);
int xlp_dev_transpose3(uint64_t* dst, const uint64_t* src, uint64_t n, uint32_t tbits);
/* MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35) for nq indices in one call: out (HOST, nq x (w + 4 log2 h) words)
 * receives, per index (HOST array idxs), the row followed by the sibling nodes bottom-up; src / nodes are the device
 * buffers of xlp_dev_merkle.  Used by the query phase of FRIProve::prove (friProve.cpp:150-177). */
int xlp_dev_merkle_proofs(uint64_t* out, const uint64_t* src, const uint64_t* nodes, uint64_t w, uint64_t h, const uint64_t* idxs, uint64_t nq);
/* Synthetic trace: dst[i] = splitmix64(seed ^ (start + i)) mod p. */
int xlp_dev_fill(uint64_t* dst, uint64_t n, uint64_t seed, uint64_t start);

/* Synthetic plookup selector for runs on a synthetic witness (benchmarks / tests): dst[i][0..dim) = src row (splitmix64(seed ^ (i >> 4)) mod n),
 * src of row stride src_stride, dst dense -- an f polynomial all of whose values occur in the table (runs of 16 rows pick the same row). */
int xlp_dev_gather_rows(uint64_t* dst, const uint64_t* src, uint64_t src_stride, uint64_t n, uint32_t dim, uint64_t seed);

/* ---- row-scale helpers between the commit stages, device pointers ------------ */
/* Polinomial::batchInverse / batchInverseParallel(res, src) -- polinomial.hpp:635-720, starks.cpp:358-359.
 * n elements of dimension dim (1 or 3); res may equal src.  Inputs must be non-zero as in the
 * reference's use (x - xi on the extended domain); a zero element yields zero. */
int xlp_dev_batch_inverse(uint64_t* res, const uint64_t* src, uint64_t n, uint32_t dim);
/* Polinomial::calculateZ(z, num, den) -- polinomial.hpp:586-607: n ext-3 elements each;
 * z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1]. */
int xlp_dev_calculate_z(uint64_t* z, const uint64_t* num, const uint64_t* den, uint64_t n);
/* Polinomial::calculateH1H2_opt1 / _opt3(h1, h2, fPol, tPol, ...) -- polinomial.hpp:349-585, starks.cpp:108-125.
 * n elements of dimension dim (1 or 3) each, dense.  Fails ("number not included") if a value of f does not
 * occur in t (the reference calls exitProcess()). */
int xlp_dev_calculate_h1h2(uint64_t* h1, uint64_t* h2, const uint64_t* f, const uint64_t* t, uint64_t n, uint32_t dim);
/* The same two on the reference's Polinomial VIEWS (polinomial.hpp:24-40: Polinomial(address, degree, dim, offset), element i at
 * address[i * offset]): every polynomial is a column group of a row-major section, addressed by its first word and the section width
 * (stride >= dim, in words).  This is how Starks::genProof hands them over (starks.cpp:108-125, 150-167 through transposeH1H2Columns /
 * transposeZColumns, starks.cpp:406-555): with strides the transposed copies around the calls disappear. */
int xlp_dev_calculate_z_strided(uint64_t* z, uint64_t z_stride, const uint64_t* num, uint64_t num_stride, const uint64_t* den, uint64_t den_stride, uint64_t n);
/* z[i] *= s (Fp^3, strided view): the second half of a ROW-SHARDED calculateZ (polinomial.hpp:586-607 is a prefix product over the rows) --
 * every rank scans its own rows with xlp_dev_calculate_z_strided, the ranks all-gather their total products (three words each) and
 * scale their rows by the product of the ranks before them. */
int xlp_dev_scale3_strided(uint64_t* z, uint64_t z_stride, uint64_t n, const uint64_t s[3]);
int xlp_dev_calculate_h1h2_strided(uint64_t* h1, uint64_t h1_stride, uint64_t* h2, uint64_t h2_stride, const uint64_t* f, uint64_t f_stride,
                                   const uint64_t* t, uint64_t t_stride, uint64_t n, uint32_t dim);
/* out[k] = first * ratio^k, k < n, elements of dimension dim (1 or 3; `first`, `ratio`: dim host words each).
 * Replaces the sequential table loops of Starks: x_n, x_2ns, x (starks.hpp:149-183: ratio = w(nBits) / w(nBitsExt),
 * first = 1 / shift) and LEv, LpEv before their INTT (starks.cpp:318-322: ratio = xi/shift, w*xi/shift). */
int xlp_dev_geometric(uint64_t* out, uint64_t n, uint32_t dim, const uint64_t* first, const uint64_t* ratio);
/* xDivXSubXi / xDivXSubWXi of genProof step 5 (starks.cpp:340-362): out[k] (ext-3) = x[k] / (x[k] - xi), x base
 * field (n words), xi 3 host words.  x[k] == xi gives zero (the reference would divide by zero). */
int xlp_dev_x_div_x_sub_xi(uint64_t* out, const uint64_t* x, const uint64_t* xi, uint64_t n);
/* Polinomial::copy between strided views -- what transposeH1H2Columns / transposeZColumns / ...Rows (starks.cpp:406-555)
 * do column by column: dst[i * dst_stride + c] = src[i * src_stride + c] for i < n, c < dim. */
int xlp_dev_copy_cols(uint64_t* dst, uint64_t dst_stride, const uint64_t* src, uint64_t src_stride, uint64_t n, uint32_t dim);
/* Step-4 coefficient split between the INTT and the NTT of the quotient (starks.cpp:267-283):
 * qq2[k * q_deg + p] = qq1[p * n + k] * shift_in^p (k < n, p < q_deg, elements of q_dim words), rows n .. n_ext-1
 * of qq2 (n_ext rows x q_deg * q_dim columns) zero; shift_in = shift^-n.  Needs q_deg * n <= n_ext. */
int xlp_dev_q_split(uint64_t* qq2, const uint64_t* qq1, uint64_t n, uint64_t n_ext, uint32_t q_deg, uint32_t q_dim, uint64_t shift_in);
/* Starks::evmap -- starks.cpp:556-669: evals[i] = sum_{k<n} L_i[k] * pol_i[(k << extend_bits) * row_stride_i],
 * L_i = LpEv if prime else LEv (ext-3, n rows), pol_i of dimension 1 or 3 (mulAddElement_adim3,
 * polinomial.hpp:722-744).  `entries` is a HOST array of nev descriptors whose `pol` are device pointers
 * to (row 0, column) of the extended section; evals: nev x 3 device words. */
typedef struct xlp_evmap_entry { const uint64_t* pol; uint64_t row_stride; uint32_t dim; uint32_t prime; } xlp_evmap_entry;
int xlp_dev_evmap(uint64_t* evals, const uint64_t* LEv, const uint64_t* LpEv, uint64_t n, uint32_t extend_bits, uint32_t nev,
                  const xlp_evmap_entry* entries);

/* Row-wise linear combinations of columns with row-independent Fp^3 coefficients -- the linear part of the FRI
 * polynomial (Starks::genProof step 5, starks.cpp:371-380 / zkevm.chelpers.step52ns.parser.cpp: Horner chains in the
 * challenges over every committed column) once the chains are written as dot products (xlayer_prover_amd/expr_opt.py):
 *   out[row * 3 * ngroups + 3 * g + c] = ( sum_{e : group_e == g} coefs[coef_e] * pol_e[row * row_stride_e .. + dim_e) )_c
 * for row < nrows; coefficient coef_e = the three words coefs[3 * coef_e ..].  `entries` is a HOST array; `pol` are device
 * pointers to (row 0, column) of row-major sections; dim 1 or 3; ngroups <= 4; results canonical.  The sections are
 * read once, coalesced, and transposed through LDS; products are accumulated unreduced (one reduction per row and sum). */
typedef struct xlp_dot_entry { const uint64_t* pol; uint64_t row_stride; uint32_t dim; uint32_t group; uint64_t coef; } xlp_dot_entry;
int xlp_dev_dot_columns(uint64_t* out, const xlp_dot_entry* entries, uint64_t nentries, uint32_t ngroups,
                        const uint64_t* coefs, uint64_t coefs_size, uint64_t nrows);

/* ---- constraint-expression VM, device pointers ------------------------------------
 * Replaces ZkevmSteps::step{2prev,3prev,3,42ns,52ns}_parser_first_avx(StepsParams &params, nrows, nrowsBatch)
 * (src/starkpil/zkevm/chelpers/zkevm.chelpers.*.parser.cpp; called at starks.cpp:73,155,193,241,371): one
 * straight-line program of field operations evaluated on every row, temporaries per row, operands taken
 * from the StepsParams buffers (src/starkpil/steps.hpp:4-19).  The reference numbers its ~120 fused
 * (function, operand-kind) cases per generated file; here ONE orthogonal encoding covers them all (every
 * reference case is one or two instructions, tools/chelpers_to_xlp.py converts the checked-in tables):
 *
 *   code[] (uint64 words) = instruction*;  instruction = head, dst params, a params, b params
 *   head = op | dst_kind << 8 | a_kind << 16 | b_kind << 24 | dst_dim << 32 | a_dim << 34 | b_dim << 36
 *   op: 0 COPY (dst = a), 1 ADD, 2 SUB (a - b), 3 MUL.  dim 1 = Goldilocks, 3 = cubic extension; a base
 *   operand combined with an extension one is embedded as (x, 0, 0) (add13/sub31c/mul13... of the reference);
 *   dst_dim = max(a_dim, b_dim).  Results are canonical.
 *
 *   kind                params                      element c of row i
 *   XLP_K_TMP1          idx                         tmp1[idx]
 *   XLP_K_TMP3          idx                         tmp3[idx][c]
 *   XLP_K_POLS          offset, stride              pols[offset + i * stride + c]
 *   XLP_K_POLS_PRIME    offset, next, n, stride     pols[offset + ((i + next) % n) * stride + c]
 *   XLP_K_CONST         col                         const_pols[col + i * n_const_pols]
 *   XLP_K_CONST_PRIME   col, next, n                const_pols[col + ((i + next) % n) * n_const_pols]
 *   XLP_K_CONST2NS / _PRIME                         same on const_pols_2ns
 *   XLP_K_NUMBER        value                       value
 *   XLP_K_CHALLENGE k / XLP_K_EVAL k                challenges[3k + c] / evals[3k + c]
 *   XLP_K_PUBLIC        k                           public_inputs[k]
 *   XLP_K_X_N / XLP_K_X_2NS / XLP_K_ZI              x_n[i] / x_2ns[i] / zi[i % zi_size]   (ZhInv::zhInv, zhInv.hpp:20-23)
 *   XLP_K_XDIVXSUBXI / XLP_K_XDIVXSUBWXI            x_div_x_sub_xi[3i + c] / x_div_x_sub_wxi[3i + c]
 *   XLP_K_Q_2NS / XLP_K_F_2NS                       q_2ns[3i + c] / f_2ns[3i + c]
 *   destinations: TMP1, TMP3, POLS, POLS_PRIME, Q_2NS, F_2NS.
 *
 * xlp_expr_create validates the program (kinds, dims, temp indices) and uploads it; xlp_dev_expr_run checks
 * every operand's address range against the *_size members (elements) for rows [row0, row0 + nrows) before
 * launching, so a malformed program or short buffer is an error, never a GPU fault. */
enum {
    XLP_OP_COPY = 0, XLP_OP_ADD = 1, XLP_OP_SUB = 2, XLP_OP_MUL = 3,
    XLP_K_NONE = 0, XLP_K_TMP1 = 1, XLP_K_TMP3 = 2, XLP_K_POLS = 3, XLP_K_POLS_PRIME = 4, XLP_K_CONST = 5,
    XLP_K_CONST_PRIME = 6, XLP_K_CONST2NS = 7, XLP_K_CONST2NS_PRIME = 8, XLP_K_NUMBER = 9, XLP_K_CHALLENGE = 10,
    XLP_K_PUBLIC = 11, XLP_K_EVAL = 12, XLP_K_X_N = 13, XLP_K_X_2NS = 14, XLP_K_ZI = 15, XLP_K_XDIVXSUBXI = 16,
    XLP_K_XDIVXSUBWXI = 17, XLP_K_Q_2NS = 18, XLP_K_F_2NS = 19, XLP_K_COUNT = 20
};
typedef struct xlp_steps_params {       /* StepsParams, steps.hpp:4-19; all device pointers, sizes in elements */
    uint64_t* pols;                 uint64_t pols_size;
    const uint64_t* const_pols;     uint64_t n_const_pols;     uint64_t const_pols_size;
    const uint64_t* const_pols_2ns; uint64_t n_const_pols_2ns; uint64_t const_pols_2ns_size;
    const uint64_t* challenges;     uint64_t challenges_size;
    const uint64_t* public_inputs;  uint64_t public_inputs_size;
    const uint64_t* evals;          uint64_t evals_size;
    const uint64_t* x_n;            uint64_t x_n_size;
    const uint64_t* x_2ns;          uint64_t x_2ns_size;
    const uint64_t* zi;             uint64_t zi_size;
    const uint64_t* x_div_x_sub_xi; uint64_t x_div_x_sub_xi_size;
    const uint64_t* x_div_x_sub_wxi; uint64_t x_div_x_sub_wxi_size;
    uint64_t* q_2ns;                uint64_t q_2ns_size;
    uint64_t* f_2ns;                uint64_t f_2ns_size;
    /* column-major programs on the extended domain only (XLP_EXPR_COLMAJOR below; 0 otherwise): rows per coset, and the distance in rows
     * between the cosets of a column (0 = coset_rows; larger when every coset block is followed by halo rows: a rank's window) */
    uint64_t coset_rows;            uint64_t coset_stride;
} xlp_steps_params;
typedef struct xlp_expr_program xlp_expr_program;
int xlp_expr_create(xlp_expr_program** out, const uint64_t* code, uint64_t ncode, uint64_t ntmp1, uint64_t ntmp3);
/* COLUMN-MAJOR BUFFERS (flags = XLP_EXPR_COLMAJOR).  The reference's slab is row-major (stark_info.cpp:473-482) because its interpreters walk a
 * row at a time; here a LANE is a row, so a column stored contiguously is read and written coalesced as it is and nothing has to be staged.
 * A program created with this flag addresses
 *   XLP_K_POLS        offset, cstride            pols[offset + c * cstride + i]                  (cstride = words between the columns of a section)
 *   XLP_K_POLS_PRIME  offset, next, n, cstride   pols[offset + c * cstride + wrap(i + next)],    wrap(j) = j - cbase >= n ? j - n : j
 *   XLP_K_CONST col / _PRIME / CONST2NS ...      const_pols[col * (const_pols_size / n_const_pols) + i]   (and with wrap(i + next))
 * where i is the STORAGE row.  The extended domain is stored COSET-MAJOR: natural row k = (q << extendBits) | r sits at storage row
 * r * coset_stride + q (xlp_steps_params: coset_rows = the domain's rows per coset, coset_stride >= coset_rows), so an extended column is the
 * 2^extendBits cosets of the base domain one after the other -- what extendPol produces coset by coset, what evmap reads (coset 0), and what makes
 * a next-row operand (next = 1 after the conversion, n = the base domain) the neighbouring word.  xlp_dev_expr_run(prog, params, row0, nrows)
 * then takes row0 / nrows as dense coset-major indices and launches coset by coset (cbase = r * coset_stride); within coset r
 *   XLP_K_ZI = zi[r];  XLP_K_X_2NS / XDIVXSUB* = table[i - r * (coset_stride - coset_rows)] (tables dense, coset-major);
 *   XLP_K_Q_2NS / XLP_K_F_2NS = q_2ns[3 * (((i - cbase) << extendBits) | r) + c]: these two stay row-major in NATURAL order (3 words per row: the
 *   INTT of the quotient and FRI read them).
 * Base-domain programs: coset_rows = 0, one coset, cbase = 0.  tools/zkevm_geometry.py to_colmajor converts a program. */
#define XLP_EXPR_COLMAJOR 1ULL
/* XLP_EXPR_CANONICAL (with XLP_EXPR_COLMAJOR): the caller promises that every word of pols / const_pols / const_pols_2ns the program reads is CANONICAL
 * (< p).  The ahead-of-time kernels then use the words as loaded (xlp_expr_codegen leaves out one canonicalisation per column load: 4 vector
 * instructions of ~60 a load is followed by); the interpreter does not care.  host/starks.hpp keeps that promise: what enters a column-major section is
 * canonical -- the witness and the constants are canonicalised as they are transposed (xlp_dev_rows_to_cm), every kernel of this library stores
 * canonical words.  A plugin generated under the promise only attaches to a program created with it. */
#define XLP_EXPR_CANONICAL 2ULL
int xlp_expr_create_ex(xlp_expr_program** out, const uint64_t* code, uint64_t ncode, uint64_t ntmp1, uint64_t ntmp3, uint64_t flags);
int xlp_expr_destroy(xlp_expr_program* prog);
/* Ahead-of-time path (the counterpart of the reference's generated straight-line chelpers,
 * src/starkpil/zkevm/chelpers/zkevm.chelpers.step3prev.cpp): write the program as HIP source (segments of
 * straight-line kernels, temporaries in registers), build it with
 *     hipcc -O3 --offload-arch=gfx950 -shared -fPIC -I xlayer_prover_amd/csrc <src> -o <plugin.so>
 * and attach the plugin; xlp_dev_expr_run then launches the compiled kernels instead of the interpreter.
 * xlp_expr_attach refuses a plugin generated from a different program. */
int xlp_expr_codegen(const xlp_expr_program* prog, const char* hip_source_path);
int xlp_expr_attach(xlp_expr_program* prog, const char* plugin_so_path);
/* Evaluate the program on rows row0 .. row0 + nrows - 1 (asynchronous on the context's stream). */
int xlp_dev_expr_run(const xlp_expr_program* prog, const xlp_steps_params* params, uint64_t row0, uint64_t nrows);

/* ---- BN128 side of StarkRecursiveF ------------------------------------------------------------------
 * An Fr element (BN254 scalar field) is 4 x uint64_t, little-endian limbs, MONTGOMERY form (R = 2^256): the in-memory
 * representation of the reference's RawFr::Element (src/ffiasm/fr.hpp), so node buffers are byte-compatible.
 *   xlp_bn128_init             load the Poseidon-BN128 tables (circomlib's optimized constants for t = 2..17,
 *                              src/poseidon_opt/constants_opt.hpp as data; tools/gen_poseidon_bn128.py).  NULL = the file
 *                              chelpers/poseidon_bn128.bin next to the library.  Called implicitly by the entry points below.
 *   xlp_merkle_bn128_num_nodes MerkleTreeBN128::getNumNodes (merkleTreeBN128.cpp:62-84)
 *   xlp_[dev_]merkle_bn128     MerkleTreeBN128::merkelize (merkleTreeBN128.cpp:89-186): src = h rows x w Goldilocks words (row-major),
 *                              nodes = num_nodes(h) Fr elements: leaves (linear hash, 3 Goldilocks words per Fr, 16 Fr per
 *                              permutation, last chunk hashed with t = pending + 1; w <= 4: the words themselves, no hash), then
 *                              the 16-ary levels, each padded to 16 slots, root last.  h a power of two.
 *   xlp_[dev_]poseidon_bn128   Poseidon_opt::hash(vector&) (poseidon_opt.cpp:9-66) on `count` states of t elements, in place
 *                              (TranscriptBN128::_updateState, transcriptBN128.cpp:82-99, is one t = 17 call). */
int xlp_bn128_init(const char* constants_path);
uint64_t xlp_merkle_bn128_num_nodes(uint64_t height);
int xlp_dev_merkle_bn128(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h);
int xlp_merkle_bn128(uint64_t* nodes, const uint64_t* src, uint64_t w, uint64_t h);
int xlp_dev_poseidon_bn128(uint64_t* states, uint32_t t, uint64_t count);
/* MerkleTreeBN128::getGroupProof (merkleTreeBN128.cpp:210-260) for nq indices in one call: src / nodes are the device buffers of a tree
 * built by xlp_dev_merkle_bn128; out (HOST, nq x (w + 64 * ceil(log16 h)) words) = the row, then per level the 16-node group (4 words per
 * node); idxs (HOST) < h. */
int xlp_dev_merkle_bn128_proofs(uint64_t* out, const uint64_t* src, const uint64_t* nodes, uint64_t w, uint64_t h, const uint64_t* idxs, uint64_t nq);
int xlp_poseidon_bn128(uint64_t* states, uint32_t t, uint64_t count);
/*   xlp_[dev_]fft_bn254        FFT<RawFr>::fft / ifft (src/ffiasm/fft.c.hpp:181-221; the transform of the fflonk prover,
 *                              src/rapidsnark/fflonk_prover.c.hpp) of n = 2^k <= 2^28 Fr elements in place, natural order in and out:
 *                              A[k] = sum_i a[i] w^(i k), w = 5^((q - 1) / n); inverse: 1/n sum_k A[k] w^(-i k). */
int xlp_dev_fft_bn254(uint64_t* data, uint64_t n, int inverse);
int xlp_fft_bn254(uint64_t* data, uint64_t n, int inverse);
/*   xlp_[dev_]msm_bn254        ParallelMultiexp<G1>::multiexp (src/ffiasm/multiexp.c.hpp:118-164, Curve::multiMulByScalar as called at
 *                              fflonk_prover.c.hpp:1700): out = sum_i scalars[i] * bases[i] on BN254 G1.  bases: n affine points, 8 words each
 *                              (x, y as Fq Montgomery limbs; (0, 0) = infinity, skipped); scalars: n x 4 words, raw little-endian 256-bit
 *                              integers; out: 8 words, the sum as an affine point ((0, 0) = infinity).  n < 2^27.  The call returns with the sum
 *                              in place: its last step (Horner over the window sums, one inversion) runs on the host. */
int xlp_dev_msm_bn254(uint64_t* out, const uint64_t* bases, const uint64_t* scalars, uint64_t n);
int xlp_msm_bn254(uint64_t* out, const uint64_t* bases, const uint64_t* scalars, uint64_t n);
/* Measurement aids for the multi-exponentiation (SURVEY.md 8d applied to row f4): the window width xlp_[dev_]msm_bn254 picks for n points
 * (n x ceil(256 / width) mixed additions in its bucket pass), and the ALU ceiling of that pass -- `lanes` lanes each add the affine point
 * `point` (device, 8 words: a cache hit for every lane) to a register-resident Jacobian accumulator `iters` times, no HBM traffic; out (device, lanes x 12 words)
 * keeps the sums.  bench.py reports the multi-exponentiation's additions/s against what this reaches. */
int xlp_msm_bn254_window_bits(uint64_t n);
int xlp_dev_msm_resident(uint64_t* out, const uint64_t* point, uint32_t lanes, uint32_t iters);

/* ---- the fflonk prover's vectors (src/rapidsnark/fflonk_prover.c.hpp, src/rapidsnark/polynomial/polynomial.c.hpp) -------
 * DEVICE vectors of Fr elements (4 x uint64 Montgomery limbs each); scalars and small tables are HOST arrays of such elements.  Asynchronous on
 * the current stream except xlp_dev_fr_eval (its result is on the host when it returns).  host/fflonk_prover.hpp drives them.
 *   xlp_dev_fr_gather_mont     out[i] = toMontgomery(values[map[i]]), 0 past the table: FflonkProver::getWitness + the toMontgomery of
 *                              computeWirePolynomial (fflonk_prover.c.hpp:615-629,702-706); values = 32-byte little-endian integers (.wtns form)
 *   xlp_dev_fr_from_mont       FflonkProver::polynomialFromMontgomery (:1683-1698): the scalars of a multi-exponentiation
 *   xlp_dev_fr_mul             out[i] = a[i] * b[i]
 *   xlp_dev_fr_lincomb         out[i] = sum_t coefs[t] * pols[t][i], i < len (a term reads 0 past its lens[t]); at most 6 terms:
 *                              Polynomial::add / sub / mulScalar / subScalar chains (polynomial.c.hpp:216-283) in one pass
 *   xlp_dev_fr_interleave      CPolynomial::getPolynomial (cpolynomial.c.hpp:39-63): out[i * k + j] = pols[j][i] (0 past lens[j]), k <= 8
 *   xlp_dev_fr_div_zh          Polynomial::divZh (polynomial.c.hpp:537-569): pol / (X^n - 1) in place, len a multiple of n
 *   xlp_dev_fr_batch_inverse   FflonkProver::batchInverse (:1531-1555), out of place
 *   xlp_dev_fr_prefix_product  out[i] = in[0] * ... * in[i] (the running products of computeZ :915-930), as a scan
 *   xlp_dev_fr_div_xm_minus    Polynomial::divByZerofier (polynomial.c.hpp:571-634): out = in / (X^m - beta) for a divisible polynomial of len
 *                              coefficients, out of place; the recurrence of every residue class mod m as a scan (the reference: one thread
 *                              per class -- a single thread for X - y)
 *   xlp_dev_fr_eval            Polynomial::fastEvaluate (polynomial.c.hpp:184-214) at npoints <= 8 points at once; values / points: HOST
 *   xlp_dev_fflonk_t0          the loop of computeT0 (:730-778) over the n4 = 4n evaluations (lagrange: npub blocks of n4; pub: device, npub values)
 *   xlp_dev_fflonk_z_numden    the first loop of computeZ (:880-913): scalars = beta, gamma, k1, k2; wpow[k] = omega_n^(2^k), k < nw
 *   xlp_dev_fflonk_z_rotate    z[(i + 1) % n] = run[i] (:915-946)
 *   xlp_dev_fflonk_t1          the loop of computeT1 (:988-1011) over n2 = 2n points: blinding = b7, b8, b9; wpow of omega_2n
 *   xlp_dev_fflonk_t2          the loop of computeT2 (:1041-1113) over 4n points: scalars = beta, gamma, k1, k2, omega_n, b7, b8, b9; wpow of omega_4n */
int xlp_dev_fr_gather_mont(uint64_t* out, const uint32_t* map, uint64_t nmap, const uint64_t* values, uint64_t nvalues);
int xlp_dev_fr_from_mont(uint64_t* out, const uint64_t* in, uint64_t n);
int xlp_dev_fr_mul(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n);
int xlp_dev_fr_lincomb(uint64_t* out, uint64_t len, uint32_t nterms, const uint64_t* const* pols, const uint64_t* lens, const uint64_t* coefs);
int xlp_dev_fr_interleave(uint64_t* out, uint64_t out_len, uint32_t k, const uint64_t* const* pols, const uint64_t* lens);
int xlp_dev_fr_div_zh(uint64_t* pol, uint64_t len, uint64_t n);
int xlp_dev_fr_batch_inverse(uint64_t* out, const uint64_t* in, uint64_t n);
int xlp_dev_fr_prefix_product(uint64_t* out, const uint64_t* in, uint64_t n);
int xlp_dev_fr_div_xm_minus(uint64_t* out, const uint64_t* in, uint64_t len, uint32_t m, const uint64_t* beta);
int xlp_dev_fr_eval(uint64_t* values, const uint64_t* pol, uint64_t len, const uint64_t* points, uint32_t npoints);
int xlp_dev_fflonk_t0(uint64_t* out, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* ql, const uint64_t* qr, const uint64_t* qm, const uint64_t* qo,
                      const uint64_t* qc, const uint64_t* lagrange, const uint64_t* pub, uint32_t npub, uint64_t n4);
int xlp_dev_fflonk_z_numden(uint64_t* num, uint64_t* den, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* s1, const uint64_t* s2, const uint64_t* s3, uint64_t n,
                            const uint64_t* scalars, const uint64_t* wpow, uint32_t nw);
int xlp_dev_fflonk_z_rotate(uint64_t* z, const uint64_t* run, uint64_t n);
int xlp_dev_fflonk_t1(uint64_t* t1, uint64_t* t1z, const uint64_t* z, const uint64_t* lagrange1, uint64_t n2, const uint64_t* blinding, const uint64_t* wpow, uint32_t nw);
int xlp_dev_fflonk_t2(uint64_t* t2, uint64_t* t2z, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* z, const uint64_t* s1, const uint64_t* s2, const uint64_t* s3,
                      uint64_t n4, const uint64_t* scalars, const uint64_t* wpow, uint32_t nw);

/* Test hook: out[i] = op(a[i], b[i]) using the device field functions the kernels are built
 * from (op codes: csrc/fri.hip field_op_kernel).  Host pointers.  Not part of the reference surface. */
int xlp_field_op(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n, int op);

/* Per-kernel timing of the extendPol fast path, measured with HIP events on the context's stream
 * (the stream the kernels are launched on).  enable(1) clears and starts recording; read() synchronises
 * and returns the summed milliseconds of {lde_top_dif, lde_mid, lde_top_dit} over `count` recorded
 * column chunks.  Measurement aid for bench.py; the reference's analogue is TimerStart/TimerStopAndLog
 * around STARK_STEP_n_LDE (src/starkpil/starks.cpp:49-55). */
int xlp_profile_enable(int on);
/* Measurement aid: the ALU ceiling of the Poseidon leaf sponge -- `count` lanes each walk `iters` chained hash_full_result
 * permutations of a state that never leaves its registers (no HBM traffic; out[i] = a checksum of lane i's final state so that the work is
 * kept).  bench.py reports the sponge's permutations/s against what this kernel reaches on the same grid. */
int xlp_dev_perm_resident(uint64_t* out, uint64_t count, uint64_t iters);
/* Measurement aid: the ALU ceiling of extendPol -- the three kernels of xlp_dev_extend(n_ext, n, ncols) in their RESIDENT form: same
 * butterfly networks, twiddle tables, LDS tiles, barriers and grids, the global data traffic removed (an input word is made from its index, an output
 * word folded into a per-lane checksum; csrc/ntt_lde_common.hpp).  n = 2^22 or 2^23, 4..64 columns.  sink: >= 1024 words of device memory (written only
 * if a checksum takes an impossible value).  Recorded by xlp_profile_enable / xlp_profile_read like an extendPol.  bench.py reports the headline's
 * kernel time against what these take (roofline.alu_ceiling). */
int xlp_dev_lde_resident(uint64_t* sink, uint64_t n_ext, uint64_t n, uint64_t ncols);
int xlp_profile_read(double ms_sum[3], uint64_t* count);
/* PLAN MODE: a dry run of the host orchestration that touches no GPU (none needs to exist).  After xlp_plan_begin -- or with XLP_PLAN=1 in the environment
 * before the first call -- device allocations (xlp_dev_alloc, the library's scratch slots and tables) are ACCOUNTED against `hbm_bytes` (0 = 288 GiB): each
 * returns a unique fake address that is never dereferenced; every other device operation returns 0 without doing anything (kernels, copies, exchanges;
 * xlp_copy_d2h zero-fills its destination, xlp_comm_init only records rank and world, xlp_dev_mem_info answers from the account).  The entry points that
 * SIZE memory run their host side as always, so a scratch slot grows to what the real call would make it.  xlp_plan_report: peak and live bytes, the
 * number of allocations, the `nlargest` largest allocations live at the peak, the nine scratch slots.  What it is for: host/zkprover_batch under
 * XLP_PLAN=1 --world G --rank r walks Starks::genProofSharded at 2^23 rows and prints what rank r would allocate (tests/test_memory_plan.py: fits 288 GiB
 * with the margin the one-GPU path keeps, G = 2, 4, 8) -- the code that allocates is the code that is counted. */
int xlp_plan_begin(uint64_t hbm_bytes);
int xlp_plan_active(void);
int xlp_plan_report(uint64_t* peak_bytes, uint64_t* live_bytes, uint64_t* allocations, uint64_t* largest_at_peak, uint32_t nlargest, uint64_t scratch_slot_bytes[9]);

/* ---- transport between the ranks of a row-sharded proof: one process per GPU -----------------------------------------
 * The reference commits a section with ntt.extendPol + treesGL[i]->merkelize() on one host (starks.cpp:53-60,134-141,215-222) and runs
 * the step programs over all rows (starks.cpp:73,155,193,241,371).  Over G ranks the rows are partitioned; these entry points are the
 * exchanges that partition needs (host/starks.hpp drives them): two all-to-alls per committed section (rows -> columns for the
 * per-column LDE, columns -> rows for the row-wise leaf sponge), ring halo rows for the next-row operands, all-gathers of the 3-column
 * quotient / FRI polynomials and plookup tables, and a few host words per rank (sub-tree roots: the G x 32-byte "cap" all-gather of
 * BASELINE's north_star, grand-product totals, partial evaluations).  Device buffers; everything is enqueued on the library's stream.
 *   backend "rccl": RCCL over xGMI (librccl.so.1 loaded on first use); `rendezvous` = a file path through which rank 0 publishes the
 *                   ncclUniqueId.  Call xlp_init(device) first: one process per GPU, no GPU call of another kind before it is needed.
 *   backend "shm":  ranks sharing one GPU or without a peer path (test box, rehearsal): POSIX shared memory named `rendezvous`, host-staged.
 *   backend NULL / "self": world must be 1; exchanges are device copies.
 * xlp_comm_exchange: recv[p] (recv_words[p] words) <- the block rank p sends me; send[p] (send_words[p] words) -> rank p; arrays of
 *   `world` entries; my own block is copied on the device.  Messages travel in rounds of <= 256 MiB per (sender, receiver); all_max_words =
 *   the longest (sender, receiver) message of the whole exchange, the SAME number on every rank (it fixes the number of rounds): a caller that
 *   knows every rank's sizes passes it and the exchange is enqueued without waiting for the GPU; 0 = unknown, the ranks agree on it through
 *   one host word each (synchronises the stream).
 * xlp_comm_ranks_seen: the number of ranks the transport itself reports (ncclCommCount / processes attached to the segment); 0 before init.
 * xlp_comm_abort: tell the other ranks that this one is giving up (the host's exit path calls it): their pending and later xlp_comm_* calls
 *   fail with the reason instead of waiting for a peer that is gone; ranks inside RCCL (no timeout there) are ended by a watchdog thread.
 *   A rank that dies without calling it is found by its heartbeat (20 s).  No-op in a world of one.
 * xlp_comm_all_gather: out[p * words ..] <- rank p's `in`.   xlp_comm_sendrecv: one ring step (halo rows).
 * xlp_comm_all_gather_host: the same for a few HOST words (synchronises the stream).
 * Everything is enqueued on the CURRENT stream (xlp_stream_select).  Environment (tests): XLP_COMM_MAX_MESSAGE_WORDS = words per message
 * and round; XLP_COMM_SELF_VIA_RCCL = the own block of an exchange travels through ncclSend / ncclRecv too (the multi-round path in a world of one).
 * xlp_comm_stats: seconds / bytes sent to peers / calls of the exchanges so far; mode bit 0 resets, bit 1 turns the timing on (each
 *   exchange then synchronises the stream before and after), bit 2 turns it off. */
int xlp_comm_init(int rank, int world, const char* backend, const char* rendezvous);
void xlp_comm_shutdown(void);
int xlp_comm_rank(void);
int xlp_comm_world(void);
int xlp_comm_ranks_seen(void);
void xlp_comm_abort(const char* why);
int xlp_comm_exchange(uint64_t* const* recv, const uint64_t* recv_words, const uint64_t* const* send, const uint64_t* send_words, uint64_t all_max_words);
int xlp_comm_all_gather(uint64_t* out, const uint64_t* in, uint64_t words);
int xlp_comm_sendrecv(uint64_t* recv, int from, const uint64_t* send, int to, uint64_t words);
int xlp_comm_all_gather_host(uint64_t* out, const uint64_t* in, uint64_t words);
int xlp_comm_barrier(void);
int xlp_comm_stats(double* seconds, uint64_t* bytes_sent, uint64_t* calls, int mode);

/* ---- band-pipelined commit of a section (device pointers) ------------------------------------------
 * A Merkle leaf is a sponge over the row in column order, so the leaf state (4 words per row = level 0 of the node buffer)
 * can absorb a section band by band, right behind the LDE of each band.  These three are the pieces; the two commit calls
 * below string them together.
 *   xlp_dev_extend_cols         extendPol (starks.cpp:53,134,215) of the columns [col0, col0 + ncols) of row-major sections of
 *                               row stride ld (in: n rows, out: n_ext rows);
 *   xlp_dev_linear_hash_absorb  continue PoseidonGoldilocks::linear_hash of every row over the band [col0, col0 + ncols) of src
 *                               (row stride ld): state (h x 4) holds the capacity between calls and the leaf after the last;
 *                               col0 % 8 == 0; ncols % 8 == 0 unless `last`; `first` starts from capacity 0;
 *   xlp_dev_merkle_levels       the levels above the leaves of merkletree (nodes[0 .. 4h) already holds level 0). */
int xlp_dev_extend_cols(uint64_t* out, const uint64_t* in, uint64_t n_ext, uint64_t n, uint64_t ld, uint64_t col0, uint64_t ncols);
int xlp_dev_linear_hash_absorb(uint64_t* state, const uint64_t* src, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t h, int first, int last);
int xlp_dev_merkle_levels(uint64_t* nodes, uint64_t h);
/* ntt.extendPol(cmX_2ns, cmX_n, ...) followed by treesGL[i]->merkelize() (starks.cpp:53-56,134-137,215-218) as one call:
 * sec_2ns (n_ext x ncols) and nodes ((2 n_ext - 1) * 4) from sec_n (n x ncols), in 64-column bands. */
int xlp_dev_commit(uint64_t* sec_2ns, uint64_t* nodes, const uint64_t* sec_n, uint64_t n_ext, uint64_t n, uint64_t ncols);
/* The two halves of the last xlp_dev_commit / xlp_dev_commit_from_host, which the reference times apart (TimerStart(STARK_STEP_n_LDE) and
 * (STARK_STEP_n_MERKLETREE), starks.cpp:51-58,133-139): summed HIP-event times of the extendPol kernels and of the leaf sponge + tree
 * levels on the compute stream (the bands interleave them).  Synchronises. */
int xlp_dev_commit_last_times(double* lde_ms, double* merkle_ms);
/* The witness upload of step 1 (the executor's committed polynomials, prover.cpp:435 -> starks.cpp:49-60) in column bands on the copy stream, for a
 * caller that interleaves it with stages of its own (host/starks.hpp, row-sharded commit: band k + 1 crosses PCIe while band k is exchanged and
 * extended).  begin: the copy stream waits for the work enqueued so far on the compute stream (it may still read the target).  cols_async: columns
 * [col0, col0 + ncols) of `rows` rows of a row-major host matrix of row stride ld -> the same place of the device matrix (page-locked host memory for
 * the full rate).  fence: work enqueued on the compute stream from now on waits for every upload enqueued so far. */
int xlp_dev_upload_begin(void);
int xlp_dev_upload_cols_async(uint64_t* dst, const uint64_t* src_host, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t rows);
int xlp_dev_upload_fence(void);
/* The same for a section whose base-domain rows are still in HOST memory -- the witness the executor leaves at pAddress
 * (prover.cpp:435 -> starks.cpp:53): h_src (n x ncols, row-major; pinned with xlp_host_alloc / xlp_host_register for full PCIe
 * rate) is uploaded into sec_n band by band on a second stream while the previous band is extended and absorbed, so the
 * 50.4 GB of the zkEVM witness cross PCIe behind the LDE and the leaf hashing instead of in front of them.  *upload_ms (may be
 * NULL) receives the time the copy stream was busy; asking for it synchronises. */
int xlp_dev_commit_from_host(uint64_t* sec_n, uint64_t* sec_2ns, uint64_t* nodes, const uint64_t* h_src, uint64_t n_ext, uint64_t n, uint64_t ncols,
                             double* upload_ms);
/* ---- column-major sections (device pointers) ------------------------------------------------------------------------------------------
 * The reference's slab is row-major (stark_info.cpp:473-482: element (row, col) of a section at row * width + col) because its loops walk a
 * row at a time.  On the GPU a lane is a row, and a section stored as COLUMNS is read and written coalesced by every stage of
 * Starks::genProof: the step programs need no staging copy (XLP_EXPR_COLMAJOR above), extendPol moves whole lines, the leaf sponge reads
 * 512 contiguous bytes per wave and word.  These entry points are the stages of the commit path on that layout; host/starks.hpp uses them
 * when the starkinfo says "xlp": {"colMajor": true}.  The row-major form stays at the host boundary: the witness is transposed band by band as
 * it arrives over PCIe, opened rows are gathered.
 *   layout: column c of a section = `rows` consecutive words at base + c * cstride.  Base-domain sections: natural row order.  Extended
 *   sections: COSET-MAJOR -- natural row k = (q << ext_bits) | r at storage row r * coset_stride + q (coset_stride = 0 means rows >> ext_bits):
 *   the 2^ext_bits cosets of the base domain one after the other, which is how extendPol produces them and how evmap (coset 0) and the
 *   next-row operands (the neighbouring word) read them.
 *   xlp_dev_rows_to_cm / xlp_dev_cm_to_rows   columns [col0, col0 + ncols) of a row-major matrix (row stride ld, natural order, `rows` rows)
 *                                             <-> ncols columns of the other form (dst / src = the first of them)
 *   xlp_dev_copy_view                          dst[i * drs + c * dcs] = src[i * srs + c * scs], i < n, c < dim: a Polinomial view
 *                                             (polinomial.hpp:24-40) of either layout copied into the other (the dense ext-3 operands of
 *                                             calculateZ / calculateH1H2)
 *   xlp_dev_extend_cm                          extendPol (starks.cpp:53,134,215): in = ncols columns of n rows, out = ncols columns of n_ext rows
 *                                             coset-major.  2^17 <= n <= 2^23, blow-up <= 8: three kernels working on whole lines of both
 *   xlp_dev_linear_hash_absorb_cm / _merkle_cm the leaf sponge / whole tree (merkleTreeGL.cpp:37-44) over a coset-major section; leaves and nodes in
 *                                             NATURAL row order (the buffer of xlp_dev_merkle)
 *   xlp_dev_merkle_proofs_cm                   xlp_dev_merkle_proofs with the opened rows gathered from the columns
 *   xlp_dev_commit_cm / _commit_from_host_cm   xlp_dev_commit / _commit_from_host: bands of columns; the host form uploads each band as a dense
 *                                             row-major [n][band] matrix into `stage` (>= n * ncols words of dead device memory) and transposes
 *                                             it into its columns ahead of its extension */
/*   xlp_dev_evmap_cm                           Starks::evmap (starks.cpp:556-669) with the polynomials as columns: entries[i].pol = the column's first
 *                                             word (coset 0 of an extended column holds its values at the rows k << extendBits), .row_stride = the
 *                                             words between the components of an extension column
 *   xlp_dev_dot_columns_cm                     xlp_dev_dot_columns on columns: out column 3 g + c (out_cstride words apart) = component c of group
 *                                             g's sum over the storage rows [0, nrows); entries as above; results canonical */
int xlp_dev_evmap_cm(uint64_t* evals, const uint64_t* LEv, const uint64_t* LpEv, uint64_t n, uint32_t nev, const xlp_evmap_entry* entries);
int xlp_dev_dot_columns_cm(uint64_t* out, uint64_t out_cstride, const xlp_dot_entry* entries, uint64_t nentries, uint32_t ngroups,
                           const uint64_t* coefs, uint64_t coefs_size, uint64_t nrows);
int xlp_dev_rows_to_cm(uint64_t* dst, uint64_t dst_cstride, uint64_t dst_coset_stride, const uint64_t* src, uint64_t ld, uint64_t col0, uint64_t ncols, uint64_t rows, uint32_t ext_bits);
int xlp_dev_cm_to_rows(uint64_t* dst, uint64_t ld, uint64_t col0, const uint64_t* src, uint64_t src_cstride, uint64_t src_coset_stride, uint64_t ncols, uint64_t rows, uint32_t ext_bits);
int xlp_dev_copy_view(uint64_t* dst, uint64_t dst_row_stride, uint64_t dst_comp_stride, const uint64_t* src, uint64_t src_row_stride, uint64_t src_comp_stride, uint64_t n, uint32_t dim);
/*   xlp_dev_copy_3d                            dst[a * ds0 + b * ds1 + i] = src[a * ss0 + b * ss1 + i], a < n0, b < n1, i < n: runs of contiguous words regrouped (the packing
 *                                             around the exchanges of a row-sharded proof on column-major windows, halo rows, windows of replicated tables) */
int xlp_dev_copy_3d(uint64_t* dst, uint64_t dst_stride0, uint64_t dst_stride1, const uint64_t* src, uint64_t src_stride0, uint64_t src_stride1, uint64_t n0, uint64_t n1, uint64_t n);
int xlp_dev_extend_cm(uint64_t* out, uint64_t out_cstride, uint64_t out_coset_stride, const uint64_t* in, uint64_t in_cstride, uint64_t n_ext, uint64_t n, uint64_t ncols);
int xlp_dev_linear_hash_absorb_cm(uint64_t* state, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, uint64_t col0, uint64_t ncols, uint64_t h, uint32_t ext_bits, int first, int last);
int xlp_dev_merkle_cm(uint64_t* nodes, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, uint64_t w, uint64_t h, uint32_t ext_bits);
int xlp_dev_merkle_proofs_cm(uint64_t* out, const uint64_t* src, uint64_t cstride, uint64_t coset_stride, const uint64_t* nodes, uint64_t w, uint64_t h, uint32_t ext_bits,
                             const uint64_t* idxs, uint64_t nq);
int xlp_dev_commit_cm(uint64_t* sec_2ns, uint64_t cs2, uint64_t* nodes, const uint64_t* sec_n, uint64_t csn, uint64_t n_ext, uint64_t n, uint64_t ncols);
int xlp_dev_commit_from_host_cm(uint64_t* sec_n, uint64_t csn, uint64_t* sec_2ns, uint64_t cs2, uint64_t* nodes, const uint64_t* h_src, uint64_t* stage, uint64_t stage_words,
                                int clear_stage /* != 0: every band's staging words are zeroed once transposed */, uint64_t n_ext, uint64_t n, uint64_t ncols, double* upload_ms);

/* Page-locked host memory for the witness slab (hipHostMalloc / hipHostRegister behind a HIP-free signature). */
int xlp_host_alloc(void** ptr, size_t bytes);
int xlp_host_free(void* ptr);
int xlp_host_register(void* ptr, size_t bytes);
int xlp_host_unregister(void* ptr);

/* Plain device memory helpers so a C++/cgo/ctypes host needs no HIP headers. */
int xlp_dev_alloc(void** ptr, size_t bytes);
/* Free / total bytes of the bound device's memory (hipMemGetInfo), for callers that decide at run time what stays resident
 * (host/starks.hpp: the extended constant polynomials, 31 GB at the zkEVM geometry -- starks.hpp:138-146 keeps them loaded). */
int xlp_dev_mem_info(size_t* free_bytes, size_t* total_bytes);
int xlp_dev_free(void* ptr);
/* Give back grow-only scratch of the context (bit k of mask = slot k, k < 9; 0x1FF = all): for a stage that grew much of it and shares the card
 * with others -- the fflonk prover releases its transform / scan / multi-exponentiation scratch (slots 8, 4, 5, 7) after a final proof.  Slot 8 holds
 * the twiddle table xlp_[dev_]fft_bn254 keeps from call to call; releasing or lending it drops the table. */
int xlp_scratch_release(uint32_t mask);
/* Let a scratch slot use a caller-owned DEVICE region instead of growing its own (dev_ptr = NULL takes it back); a request the region cannot
 * hold falls through to the slot's own scratch.  The fflonk prover in reserved-memory mode lends the rest of its region to the transform /
 * scan / multi-exponentiation slots for the duration of a proof (no allocation inside the proof, nothing left behind). */
int xlp_scratch_lend(uint32_t slot, void* dev_ptr, size_t bytes);
int xlp_dev_memset(void* dst_dev, int value, size_t bytes);          /* asynchronous on the context's stream */
int xlp_copy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int xlp_copy_d2h(void* dst_host, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* XLP_GL_H */
