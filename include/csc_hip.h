/* csc_hip.h -- C ABI of libcsc_hip.so: batched convolutional OMP / MMP-DF on MI355X (gfx950).
 *
 * The reference (yanisgomes/csc) is pure Python and has no FFI; its boundary for this path is the
 * Python class API.  Each entry point below states which reference call site(s) it replaces
 * (paths relative to the reference root).  The Python mirror of that API (csc_amd/) binds these
 * symbols through ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (hipMalloc / torch tensor);
 *     the library allocates no device memory and keeps no state between calls (except an attached
 *     csc_timer, see below);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it
 *     and the calls return without synchronising;
 *   - return value 0 = success, negative = argument error (see csc_last_error()), positive = a
 *     hipError_t from a launch;
 *   - activation index convention is the reference's: idx = pos * K + atom
 *     (src/csc/dictionary.py:436, src/csc/mmp.py:74).
 */
#ifndef CSC_HIP_H
#define CSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSC_ABI_VERSION 8
#define CSC_TILE_COLS 144      /* atoms per column tile of the correlation kernel (9 x 16)   */
#define CSC_TILE_ROWS 256      /* positions per row tile (4 waves x 4 x 16)                   */
#define CSC_MAX_SPARSITY 16    /* largest tree depth / support size                           */
#define CSC_MAX_CONNECTIONS 8  /* largest number of siblings per node                         */
#define CSC_FLAG_FP32_ONLY 1   /* csc_mmpdf_batch flags: skip the fp16 matrix-core filter      */
#define CSC_FLAG_NO_LEAN 2     /* keep single-child levels on the register-resident kernel      */
#define CSC_FLAG_RECORDS_AOS 4 /* per-sparsity records as one contiguous record per signal      */
#define CSC_FLAG_LOWRANK 8     /* the bank image holds a low-rank section (csc_bank_lowrank)     */
#define CSC_FLAG_SIMATOMS 16   /* ... and the atom summary of d_sim_bits (csc_bank_simatoms)     */
#define CSC_STATIC_SCRATCH_BYTES (64u << 20)   /* static device scratch of csc_mmpdf_batch_static      */

int csc_abi_version(void);
const char* csc_last_error(void);

/* Sizes of the packed dictionary images for K atoms of length L and signals of length N. */
size_t csc_bank32_elems(int K, int L);          /* floats   in d_bank32t                        */
int    csc_sim_words(int N, int L);             /* uint32 per (k_sel, k) row of d_sim_bits      */

/* Atom-bank materialisation in HBM.  Replaces the per-call rebuild of the atom matrix in
 * src/csc/dictionary.py:390,404 (np.array([atom() for atom in self.atoms])).
 *   d_bank64  [K][L] f64 (row = atom, as getLocalDictionary(), src/csc/dictionary.py:112-113)
 *   d_bank32t csc_bank32_elems(K, L) floats, opaque to the caller: the bank as MFMA operand blocks in fp32
 *             ([column tile of 144][slab of 32 rows][16-column block][32][16], columns sorted by support so
 *             that all-zero blocks can be skipped), the column permutation and block masks, the same blocks
 *             in fp16 and max_k ||a_k||^2. */
int csc_bank_pack(const double* d_bank64, int K, int L, float* d_bank32t, void* stream);

/* Low-rank screen of large dictionaries (K > 144; no counterpart in the reference, whose correlations are one FFT
 * convolution per atom, src/csc/dictionary.py:395-407).  d_basis: [rank][L] f64, ORTHONORMAL vectors (any basis: e.g.
 * the leading eigenvectors of the bank's Gram matrix B^T B); rank must equal csc_lowrank_rank().  Fills the low-rank
 * section of the packed bank (csc_bank32_elems reserves it when K > 144): the basis as an fp32 tile, the atoms'
 * coefficients in fp16 and, per atom, the rigorous bound of what the rank-`rank` product can miss of a correlation.
 * csc_mmpdf_batch with CSC_FLAG_LOWRANK then screens a node's correlations through the basis first (34 matrix
 * instructions per 16 atoms instead of ~190) and multiplies in full only the twentieth of the dictionary that bound
 * cannot exclude; results are identical with or without it. */
int csc_lowrank_rank(void);
int csc_bank_lowrank(float* d_bank32t, const double* d_bank64, int K, int L, const double* d_basis, int rank, void* stream);

/* Atom summary of a similarity table for the low-rank screen (K > 144; no counterpart in the reference, which masks the
 * correlations of every sibling in full, src/csc/mmp.py:196-201).  d_sim_bits / W: a table of csc_similarity_table[_packed]
 * for this bank and its row length csc_sim_words(N, L).  Records in the packed bank, per pair of atoms, whether their
 * similarity row has a bit set at any shift (K x ceil(K / 32) words) and, per atom, the lowest and the highest shift at
 * which anything is similar to it (2 K words).  With CSC_FLAG_LOWRANK | CSC_FLAG_SIMATOMS and THAT table, the low-rank
 * screen of a node's second and later children reads similarity words only for the columns and the shifts the summary
 * names instead of 6 words per column and earlier pick; results are identical with or without it.  Call it again whenever the table passed to csc_mmpdf_batch changes (another threshold or signal
 * length). */
int csc_bank_simatoms(float* d_bank32t, int K, int L, const uint32_t* d_sim_bits, int W, void* stream);

/* Atom-vs-dictionary similarity table.  Replaces, for every node, the
 * computeCorrelations(atom_signal)/||atom_signal|| <= thr mask of src/csc/mmp.py:92-93,196-201:
 * bit (dp + P-1) of row (k_sel, k) is SET where the mask is 0, i.e. where the correlation of atom
 * k_sel with atom k shifted by dp = pos - pos_sel exceeds `threshold` (shift invariant because
 * every placement is 'valid').  d_sim_bits [K][K][csc_sim_words(N, L)] uint32. */
int csc_similarity_table(const double* d_bank64, int K, int L, int N, double threshold,
                         uint32_t* d_sim_bits, void* stream);

/* The same table, bit for bit, from the PACKED bank (csc_bank_pack's image) on the fp32 matrix cores: an entry is
 * decided by the fp32 cross-correlation wherever that is further from the threshold than its rounding bound and
 * recomputed in float64, as csc_similarity_table computes it, where it is not.  6.2 s -> well under a second for the
 * 9570-atom dictionary of BASELINE config 3 (same reference lines: src/csc/mmp.py:92-93, 196-201).  Falls back to
 * csc_similarity_table for atoms longer than the packed bank's slabs cover. */
int csc_similarity_table_packed(const float* d_bank32t, const double* d_bank64, int K, int L, int N, double threshold,
                                uint32_t* d_sim_bits, void* stream);

/* Batched valid cross-correlation on fp32 MFMA.  Replaces CSCDictionary.computeCorrelations
 * (src/csc/dictionary.py:395-407) for B signals at once.
 *   d_signals [B][N] f64 ; d_corr [B][P][K] f32, P = N - L + 1. */
int csc_correlate(const float* d_bank32t, int K, int L, const double* d_signals, int B, int N,
                  float* d_corr, void* stream);

/* The same correlations in float64 (fp64 FMA, no matrix cores): what the public computeCorrelations and the
 * rmatvec of getMaskedConvOperator return (src/csc/dictionary.py:395-407, 409-425 are float64 arithmetic; a caller
 * running its own lsqr on the operator relies on ~1e-12).  d_corr [B][P][K] f64; B <= 65535. */
int csc_correlate64(const double* d_bank64, int K, int L, const double* d_signals, int B, int N, double* d_corr,
                    void* stream);

/* Tree shape of runMMPDF (src/csc/mmp.py:303-313, 388-424): level d (1..max_sparsity) holds
 * F_d = min(branches, connections^d) nodes; node j of level d has parent j mod connections^(d-1)
 * and sibling rank j div connections^(d-1).  Writes F_1..F_max into level_sizes[max_sparsity]
 * (may be NULL) and returns the number of non-root nodes per signal, or a negative error. */
int csc_tree_nodes(int max_sparsity, int connections, int branches, int* level_sizes);

/* Bytes of scratch csc_mmpdf_batch needs for S signals (two generations of node residuals). */
size_t csc_mmpdf_workspace_bytes(int S, int N, int max_sparsity, int connections, int branches);

/* Batched MMP-DF / OMP: all signals x frontier nodes per level, one launch per level.
 * Replaces the joblib fan-out over MMPTree(...).runMMPDF(...) of
 * src/csc/dictionary.py:1047-1071,1147 (sparVar), :1012-1045 (mmpdf, tree dict) and, with
 * connections = branches = 1, the omp loops of :492-552, :574-649.
 *
 * Node id of (level d, index j) is off_d + j with off_1 = 0, off_{d+1} = off_d + F_d; NN = total.
 *   flags       0, or CSC_FLAG_FP32_ONLY: correlate in fp32 only (v_mfma_f32_16x16x4_f32).  By default a
 *               node's correlations are first taken on the fp16 matrix cores as a filter and every entry
 *               within the fp16 rounding bound of the maximum is re-scored in float64; the selections are
 *               the float64 arg-max either way (identical results, the flag exists to check that).
 *               CSC_FLAG_NO_LEAN: tree levels on which every node has one child normally run on a kernel
 *               that takes the fp16 product in wave-private (128 position x 16 column) tasks (four nodes per CU
 *               instead of two) and walks each chain of such nodes down to its leaf in one launch; the flag keeps
 *               them on the register-resident kernel, one launch per level (identical results, same purpose)
 *   d_signals   [S][N] f64
 *   d_node_pos  [S][NN] int32   selected position of the node      (src/csc/mmp.py:74-76)
 *   d_node_atom [S][NN] int32   selected atom index
 *   d_node_mse  [S][NN] f64     mean(residual^2) of the node       (src/csc/mmp.py:108-109)
 *   d_node_coef [S][NN][max_sparsity] f64 or NULL: LS coefficients of the node's support against
 *               the ORIGINAL signal, in root->leaf order (lsqr(masked_conv_op, signal),
 *               src/csc/dictionary.py:528)
 *   d_rec_mse   [S][max_sparsity] f64     per sparsity k: min leaf MSE of the depth-k tree
 *   d_rec_leaf  [S][max_sparsity][2] int32  leaf index j of the LAST minimum ('<=',
 *               src/csc/mmp.py:586) and of the FIRST minimum (np.argmin, src/csc/mmp.py:464)
 *   d_rec_sel   [S][max_sparsity][max_sparsity][2] int32  (pos, atom) root->leaf of the last-min
 *               leaf, -1 padded                                      (src/csc/mmp.py:257-265)
 *               With CSC_FLAG_RECORDS_AOS the three d_rec_* arrays are ONE buffer of contiguous per-signal
 *               records instead: d_rec_mse points to [S][(16 + 8 K) K] bytes, record s = K f64 mse | K x 2 i32
 *               leaf | K x K x 2 i32 (pos, atom), K = max_sparsity (d_rec_leaf / d_rec_sel are ignored) -- the
 *               rows the multi-GPU result gather sends as they are (replaces the list the joblib fan-out
 *               returns, src/csc/dictionary.py:1147-1148)
 *   d_stats     [4] int32 or NULL (must be zeroed by the caller): [0] selections that fell back
 *               to the fp32 arg-max (candidate overflow), [1] re-selected activations (dead Cholesky
 *               columns), [2] number of fp64 re-scored candidates (a diagnostic: it may vary by a few per
 *               thousand from run to run, the outputs do not), [3] nodes the fp16 filter could not
 *               settle (multiplied again in fp32; on the single-child levels the rest of the node's chain is
 *               then done in fp32 too)
 *   workspace   csc_mmpdf_workspace_bytes(...) bytes, 16-byte aligned (hipMalloc's are): two generations of node
 *               residuals + one flag byte per frontier node (dictionaries beyond one register tile: a child's row
 *               also carries its parent's per-unit screen bounds from the screen kernel to the three-product
 *               kernel, until the child is written)
 */
int csc_mmpdf_batch(const float* d_bank32t, const double* d_bank64, const uint32_t* d_sim_bits,
                    int K, int L, const double* d_signals, int S, int N,
                    int max_sparsity, int connections, int branches, int flags,
                    void* d_workspace, size_t workspace_bytes,
                    int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse,
                    double* d_node_coef, double* d_rec_mse, int32_t* d_rec_leaf,
                    int32_t* d_rec_sel, int32_t* d_stats, void* stream);

/* csc_mmpdf_batch without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE
 * (pure Python, it allocates freely): this entry point exists for callers that cannot call hipMalloc -- code
 * under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments, results and
 * kernels as csc_mmpdf_batch; the node workspace (two generations of node residuals + the chain flags) lives in
 * a file-scope `static __device__` array of CSC_STATIC_SCRATCH_BYTES (64 MiB) that every GPU the library runs
 * on holds until the process ends.  The call allocates and frees nothing (no hipMalloc / hipFree, no library
 * allocator); a batch whose csc_mmpdf_workspace_bytes(...) exceeds the array is refused with -10 (config 2:
 * up to 1844 signals per call).  Every call on one GPU shares that GPU's array: calls that overlap on one GPU
 * (two streams, two host threads) race -- serialise them or use csc_mmpdf_batch with separate workspaces; the
 * library keeps no busy flag (it could not see a captured graph's replays).  The array costs 64 MiB of HBM on every
 * GPU the process touches through this library, used or not.  Its device address is looked up by csc_bank_pack (the
 * set-up call) -- or by the first call, if no bank was packed on that device -- so a call under stream capture
 * issues launches and one memset only: tests/test_zz_no_alloc.py captures a batch into a HIP graph after one
 * ordinary call (which loads the code objects and sets the kernels' LDS attributes) and replays it on new signals.
 * csc_static_scratch_bytes() returns the array's size. */
size_t csc_static_scratch_bytes(void);
int csc_mmpdf_batch_static(const float* d_bank32t, const double* d_bank64, const uint32_t* d_sim_bits,
                           int K, int L, const double* d_signals, int S, int N,
                           int max_sparsity, int connections, int branches, int flags,
                           int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse,
                           double* d_node_coef, double* d_rec_mse, int32_t* d_rec_leaf,
                           int32_t* d_rec_sel, int32_t* d_stats, void* stream);

/* Test aid: the fp16 matrix-core filter product that csc_mmpdf_batch only ever holds in registers, so that the
 * rounding bound its exactness rests on can be checked directly (no counterpart in the reference; its
 * correlations are float64, src/csc/dictionary.py:395-407).
 *   d_prod [B][P][K] f32: product of signal b, scaled to unit norm and rounded to fp16, with the fp16 bank, in
 *          units of ||signal_b||; variant 0 = the loop of the register-resident level kernel, 1 = the
 *          wave-private task loop of the chain kernel
 *   d_eps  [B] f32: the bound on |d_prod - float64 correlation / ||signal||| the kernels assume (their
 *          candidate band is twice this);   d_norm [B] f64: ||signal_b||
 * Covers what the filter covers: K <= 144, P <= 256, L <= 512, N <= 1024. */
int csc_filter_product(const float* d_bank32t, int K, int L, const double* d_signals, int B, int N,
                       int variant, float* d_prod, float* d_eps, double* d_norm, void* stream);

/* Batched plain convolutional matching pursuit.  Replaces CSCDictionary.mp / mpFromDict /
 * mpSparVarFromDict (src/csc/dictionary.py:764-921): per iteration, correlations of the residual
 * (matrix cores + fp64 re-score as in csc_mmpdf_batch), signed arg-max, activation_mask[idx] += <atom, r>,
 * residual = signal - conv(mask * mask) -- the reference multiplies the mask by itself (:803), kept.
 * One node per iteration (NN = iterations, node i = state after iteration i + 1):
 *   d_node_pos / d_node_atom [S][iterations]   activation selected at that iteration
 *   d_node_mse  [S][iterations]                mean((signal - approx)^2) after it
 *   d_node_mask [S][iterations][iterations]    mask value per support slot after it (slot q = q-th
 *               selection; a re-selected activation accumulates in its first slot, the later slot is 0)
 *   workspace: csc_mmpdf_workspace_bytes(S, N, iterations, 1, 1) bytes. */
int csc_mp_batch(const float* d_bank32t, const double* d_bank64, int K, int L, const double* d_signals,
                 int S, int N, int iterations, void* d_workspace, size_t workspace_bytes,
                 int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse, double* d_node_mask,
                 int32_t* d_stats, void* stream);

/* Batched least-squares projection on GIVEN supports (fp64 Gram / Cholesky, one workgroup per
 * problem).  Replaces CSCDictionary.getSignalProjectionFromAtoms (src/csc/dictionary.py:449-473:
 * lsqr(masked_conv_op, signal) then op @ activations) and its callers
 * MMPTree.getSubTreeFromMMPTreeDict / mmpdfCandidateFromMMPTreeDict / ompCandidateFromMMPTreeDict
 * (src/csc/mmp.py:641-733) for M problems at once.
 *   d_signals   [S][N] f64
 *   d_sig_index [M] int32 or NULL: signal row of problem m (NULL: m itself, then M <= S); values must
 *               lie in [0, S)
 *   d_support   [M][kmax][2] int32 (pos, atom); the leading run of in-range pairs is the support,
 *               pad the rest with -1.  An activation listed twice gets coefficient 0 the second
 *               time (the reference's mask holds it once) and counts in d_stats[1].
 *   d_coef      [M][kmax] f64 or NULL   LS coefficients in support order (0 beyond the support)
 *   d_mse       [M] f64 or NULL         mean((signal - approx)^2)
 *   d_approx    [M][N] f64 or NULL      the projection                                           */
int csc_project_batch(const double* d_bank64, int K, int L, const double* d_signals, int S, int N,
                      const int32_t* d_sig_index, const int32_t* d_support, int M, int kmax,
                      double* d_coef, double* d_mse, double* d_approx, int32_t* d_stats, void* stream);

/* Batched atom matching.  Replaces utils.computeMaxTruePositives (src/csc/utils.py:23-69) for M
 * (truth, recovered) pairs at once: the maximum number of true positives = size of a maximum matching
 * between true and recovered atoms whose positions differ by <= pos_err_threshold AND whose placed
 * atoms have |inner product| >= corr_err_threshold (the optimum of the reference's Hungarian
 * assignment on the same 0/1 cost).  Positions may be any integers (the inner product of
 * non-overlapping placements is 0).
 *   d_true   [M][kt][2] int32 (pos, atom), pad with -1      d_approx [M][ka][2] int32, pad with -1
 *   d_tp     [M] int32                                       kt, ka <= 16 */
int csc_match_batch(const double* d_bank64, int K, int L, const int32_t* d_true, int kt,
                    const int32_t* d_approx, int ka, int M, double pos_err_threshold,
                    double corr_err_threshold, int32_t* d_tp, void* stream);

/* On-device constrained signal generator (benchmark input).  Replaces the per-signal loop of
 * generateConstrainedSignalsDB / generateConstrainedTestSignal (src/csc/dictionary.py:271-324, 326-365): signal s
 * = d_sparsity[s] unit-amplitude atoms at uniformly drawn (atom, position), a draw rejected when it lies closer
 * than pos_err_threshold samples to (:305 `<`), or has a normalised inner product above corr_err_threshold with
 * (:306 `>`), an atom already placed; plus white Gaussian noise of variance var(clean[|clean| > 1e-3]) /
 * 10^(snr/10) (:318-322).  Philox-4x32-10 keyed by `seed`, counter = (draw / sample, first_index + s, stream): a
 * signal depends on (seed, first_index + s) only.  The reference draws from NumPy's global RNG, so the STREAM is not
 * reproduced -- parity is statistical (the constraints, the SNR rule), and the tests check exactly those.
 *   d_sparsity [S] int32 (<= kmax), d_snr_db [S] f64, d_signals [S][N] f64 out, d_truth [S][kmax][2] int32 out
 *   (pos, atom) in placement order, -1 padded; d_fail [S] int32 or NULL: 1 where 65536 draws did not place all atoms */
int csc_generate_signals(const float* d_bank32t, const double* d_bank64, int K, int L, int N, int S,
                         const int32_t* d_sparsity, const double* d_snr_db, int kmax, double pos_err_threshold,
                         double corr_err_threshold, uint64_t seed, uint64_t first_index, double* d_signals,
                         int32_t* d_truth, int32_t* d_fail, void* stream);

/* Streaming writer of the sparVar result list (HOST function: every pointer is host memory).  Replaces the
 * json.dump(results, f, indent=4, default=handle_non_serializable) that ends mmpdfSparVarPipelineFromDB
 * (src/csc/dictionary.py:1148-1150, src/csc/utils.py:7-15) for the list under 'mmp': the SAME BYTES, written
 * straight from the packed per-signal records of CSC_FLAG_RECORDS_AOS (K f64 mse | K x 2 i32 leaf | K x K x 2 i32
 * (pos, atom)): per signal {"id", "snr", "results": [{"mse", "path", "atoms": [{"x","b","y","s"}], "delay"} for
 * sparsity 1..K]} (src/csc/dictionary.py:1066-1071, src/csc/mmp.py:594-604).  Floats are formatted like Python's
 * float.__repr__, 'path' is the base-`connections` digit string of the '<=' leaf (src/csc/mmp.py:303-313,586).
 *   path, append      file to write ("wb") or to append to ("ab": the caller has written the dict's other members
 *                     and the list's key, ending with `"mmp": `)
 *   ids [S] int64; snr [S] f64, printed as an integer where snr_is_int[s] (NULL: always as a float)
 *   delay             the 'delay' value of every result (the amortised per-signal time of the batched call)
 *   atom_b/y/s [n_atoms] f64: the dictionary's parameters ('s' = the normalised sigma, src/csc/mmp.py:262)
 *   indent_level      nesting depth of the list's key (1 for the pipelines' top-level dict)
 *   close_object      also write the closing braces of the `indent_level` enclosing dicts
 * Returns 0, or a negative code with csc_json_last_error().  csc_format_double writes repr(x) (NUL-terminated)
 * and returns its length, or -1 if `cap` is too small. */
int csc_sparvar_json_write(const char* path, int append, const void* records, int64_t S, int max_sparsity,
                           int connections, const int64_t* ids, const double* snr, const uint8_t* snr_is_int,
                           double delay, const double* atom_b, const double* atom_y, const double* atom_s,
                           int n_atoms, int indent_level, int close_object);
int csc_format_double(double x, char* out, int cap);
const char* csc_json_last_error(void);

/* Optional per-launch HIP-event timer.  No counterpart in the reference (its time_decorator,
 * src/csc/dictionary.py:25-45, logs host wall-clock per method); bench.py uses it to measure the
 * level kernel's launch durations live, on the stream the kernel runs on.  While a timer is
 * attached (per host thread), csc_mmpdf_batch records one span per level launch, tagged with the
 * level (the launch that walks the single-child chains covers several levels and is tagged with the
 * first of them), until `capacity` spans are held.  Recording never synchronises; csc_timer_read must be
 * called after the stream has been synchronised and returns the number of spans written
 * (elapsed milliseconds + tags), or a negative / hipError_t code. */
typedef struct csc_timer csc_timer;
csc_timer* csc_timer_create(int capacity);
void csc_timer_destroy(csc_timer* t);
void csc_timer_attach(csc_timer* t);            /* NULL detaches */
void csc_timer_reset(csc_timer* t);
int  csc_timer_count(const csc_timer* t);
int  csc_timer_read(csc_timer* t, float* ms, int* tag, int n);

#ifdef __cplusplus
}
#endif
#endif /* CSC_HIP_H */
