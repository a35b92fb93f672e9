// csc_hip.hip -- gfx950 (MI355X) kernels + C ABI for batched convolutional OMP / MMP-DF.
//
// Hot kernel: mmp_level_kernel.  One 256-thread workgroup owns ONE tree node (signal s, frontier
// index j of level d).  It
//   1. loads the node's fp64 residual into LDS (fp64 + an fp32 copy),
//   2. computes the node's full (P x K) correlation matrix on the fp32 matrix cores: the signal
//      operand is the Hankel matrix A[p][l] = r[p + l], read straight out of the LDS-resident
//      residual (one ds_read_b32 per fragment, no materialisation); the atom bank streams
//      L2 -> LDS in 32-row slabs; 4 waves x (4 x 9) v_mfma_f32_16x16x4_f32 accumulators keep the
//      whole matrix in registers (reference: src/csc/dictionary.py:395-407, src/csc/mmp.py:69).
//      The bank's columns are sorted by support so that (32-row x 16-atom) blocks that hold only
//      the atoms' zero padding are neither staged nor multiplied (exact: they contribute +0),
//   3. selects the node's children with the reference's multiplicative sibling suppression
//      (src/csc/mmp.py:210-231): fp32 arg-max by wavefront reductions, then every entry within a
//      guaranteed fp32 error band of the maximum is re-scored in fp64 so the selection equals the
//      float64 reference's,
//   4. solves each child's least squares on its support in fp64 by extending the parent's Cholesky
//      factor with one row (replaces lsqr on the masked operator, src/csc/mmp.py:65-68) and writes
//      the child's residual, factor, MSE, coefficients and (pos, atom).
// Levels are launched one after the other (level-synchronous frontier, closed form of
// src/csc/mmp.py:303-313,388-424); reduce_kernel then extracts the per-sparsity winners
// (src/csc/mmp.py:577-606, 458-467).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>

#include "../../include/csc_hip.h"

#define WG 256
#define SLAB 32                        // bank rows per LDS slab
#define MT 4                           // 16-row MFMA tiles per wave
#define NT 9                           // 16-column MFMA tiles per workgroup (144 atoms)
#define BLKF (SLAB * 16)               // floats of one (slab, column-tile) block, contiguous in memory
#define SLABF (NT * BLKF)              // floats of one slab of a 144-atom bank tile
#define MAX_SLABS 64                   // L <= 2048
#define META_INTS (CSC_TILE_COLS + MAX_SLABS)   // per bank tile: perm[144] + slab masks[64]
#define CAND_CAP 128
#define SUPP_BITS 0x7fc00001u          // quiet-NaN payload marking a suppressed entry (value 0)
#define KMAX CSC_MAX_SPARSITY
#define TRI(n) ((n) * ((n) + 1) / 2)
#define DELTA_REL 6.103515625e-05f      // 2^-14: fp32 candidate band, in units of ||residual||

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 8-byte access
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define R16_COPIES 8                   // fp16 residual window: copy c holds r[i + c], so every 8-element Hankel
#define R16_CS 800                     // fragment is one aligned 16-byte LDS read.  Copy stride = 64 mod 256 bytes:
                                       // the 12 distinct addresses of each 16-lane group of a ds_read_b128 (window
                                       // elements e..e+3, e+12..e+19) then fall on 12 different 16-byte bank slots

// Development aid (make prof): per-phase shader-clock totals of the level kernels, summed over workgroups
// (slots 0-15: mmp_level_kernel / the tiled kernel, 16-31: mmp_level_lean_kernel; scripts/phase_prof.py names them).
#ifdef CSC_PHASE_PROF
// in-loop stamps of the tiled fp16 product (wave 0 only; s_memtime: each stamp also waits for the wave's LDS reads)
#define T3_CLK() ((unsigned)__builtin_amdgcn_s_memtime())
#define T3_STAMP(i) do { const unsigned n_ = T3_CLK(); t3acc[i] += n_ - t3last; t3last = n_; } while (0)
#define PROF_BUCKETS 256                // the counters are spread over buckets (block index) to keep the atomics apart
__device__ unsigned long long g_prof[32][PROF_BUCKETS];
// per-node trace of the chain kernel's product phase: {HW_ID, start, end (s_memtime, low 32 bits), block | XCC_ID << 24 | depth << 28}
#define TRACE_CAP (1 << 18)
__device__ unsigned g_trace[TRACE_CAP][4];
__device__ unsigned g_trace_n;
#define PROF_FIELDS unsigned long long prof_t; int prof_base;
#define PROF_INIT(base) do { if (threadIdx.x == 0) { sl->prof_base = (base); sl->prof_t = __builtin_readcyclecounter(); } } while (0)
#define PROF_MARK(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) { \
    unsigned long long n_ = __builtin_readcyclecounter(); \
    atomicAdd(&g_prof[sl->prof_base + (i)][blockIdx.x % PROF_BUCKETS], n_ - sl->prof_t); sl->prof_t = n_; } } while (0)
#else
#define PROF_FIELDS
#define PROF_INIT(base)
#define PROF_MARK(i)
#define T3_STAMP(i)
#endif

// Development aid (make candcheck): the chain kernel built with a forced publication order of the running maximum
// (CSC_CAND_ORDER 0 = as shipped, 1 = loosest: a task's threshold sees only the task's own maximum, 2 = tightest: every
// task sees the final maximum, i.e. the second pass is forced) and a dump of every node's raw count and final candidate
// list, for tests/test_gpu_cand_order.py.
#ifdef CSC_CAND_ORDER
#define CAND_DUMP_NODES 16384
#define CAND_DUMP_W 260                 // raw count | final count | up to 256 indices | spare
__device__ int g_cand_dump[CAND_DUMP_NODES][CAND_DUMP_W];
#endif

static thread_local char g_err[256] = "";
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

struct TreeShape {
    int F[KMAX + 1];      // F[d] nodes at level d (F[0] = 1)
    int off[KMAX + 2];    // node-id offset of level d (off[1] = 0)
    int pw[KMAX + 1];     // min(connections^d, 2^30)
};

struct LevelArgs {
    const float* bank32t; const int32_t* meta; const double* bank64; const uint32_t* sim_bits;
    int K, L, Lp, N, P, W;
    int depth;            // level d of the parents expanded by this launch
    int Kmax, connections, NN;
    int NS;               // doubles per node in the workspace: N residual + TRI(Kmax) Cholesky factor + Kmax coefficients
    int mode;             // 0: orthogonal (OMP / MMP-DF least squares); 1: plain matching pursuit
    const double* resid_in;   // d == 0: signals [S][N]; else [S][F[d]][NS]
    double* resid_out;        // [S][F[d+1]][NS] or nullptr (children are final leaves)
    int32_t* node_pos; int32_t* node_atom; double* node_mse; double* node_coef;
    int32_t* stats;
    const _Float16* bank16;  // the bank in fp16, blocked like bank32t with (k-group, column, 8 rows) fragments
    const float* amax2;   // max_k ||a_k||^2 (scales the fp16 error bound)
    const double* anorm2; // [K] ||a_k||^2
    uint8_t* flags;       // [S][F[d]] lean kernel -> follow-up kernel: e + 1 = the chain below this node has to be
                          // taken up at level e (in fp32), 0 = done
    int S;                // signals
    const double* signals;    // [S][N]            } the chain kernels walk several levels: level e reads
    double* ws[2];            // [S][F][NS] each   } signals (e == 0) or ws[e & 1] and writes ws[(e + 1) & 1]
    const _Float16* lr_u16;   // low-rank screen (CSC_FLAG_LOWRANK): the basis x 64 as fp16 B fragments [hi | lo][k-step][16-column block][lane][8] | nullptr
    const _Float16* lr_v16;   //   basis coefficients of the bank's columns, fp16 B fragments [16-column block][k-step][lane][8]
    const float* lr_slack;    //   per bank column: what the low-rank product can miss of a correlation, in units of ||r||
    const uint32_t* simatoms; //   [K][sa_w] bit k' of row k: atom k' is similar to atom k at some shift (CSC_FLAG_SIMATOMS) | nullptr
    int sa_w;                 //   words per row of simatoms
    const int* simrange;      //   [2][K] lowest / highest bit set anywhere in atom k's similarity rows (with simatoms) | nullptr
    int sibling;          // screen16_kernel / mmp_level_tiled16_kernel: the one child (sibling) of each node this launch selects
    int picks_only;       // level 0 ahead of the chain kernel: write the children's (pos, atom) and <atom, signal> only
    int from_root;        // chain kernel: build the level-1 node from the signal and the root level's pick
    TreeShape ts;
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// Wave reductions on the DPP network (no LDS traffic, ~2 VALU per step).  After the last step the
// result is in lane 63 (wave) or in lanes 31 and 63 (the two 32-lane halves); the other lanes hold
// partial values.
#define DPP_QUAD_1032 0xB1
#define DPP_QUAD_2301 0x4E
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x) {       // lanes outside ROWMASK receive 0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double half_wave_sum(double v) {  // totals in lanes 31 and 63
    v += dpp_f64<DPP_QUAD_1032, 0xF>(v);
    v += dpp_f64<DPP_QUAD_2301, 0xF>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(v);
    v += dpp_f64<DPP_ROW_MIRROR, 0xF>(v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xA>(v);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {       // total in lane 63
    v = half_wave_sum(v);
    v += dpp_f64<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o);
        int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ bool is_supp(float x) { return __float_as_uint(x) == SUPP_BITS; }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_max(float x) {          // lanes outside ROWMASK keep x
    const int y = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROWMASK, 0xF, false);
    return fmaxf(x, __int_as_float(y));
}
__device__ __forceinline__ float wave_max(float v) {         // maximum in lane 63
    v = dpp_max<DPP_QUAD_1032, 0xF>(v);
    v = dpp_max<DPP_QUAD_2301, 0xF>(v);
    v = dpp_max<DPP_ROW_HALF_MIRROR, 0xF>(v);
    v = dpp_max<DPP_ROW_MIRROR, 0xF>(v);
    v = dpp_max<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_max<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// sum_{i in [lo, hi)} a[i] * b[i] over LANES cooperating lanes (lane = index within the group); the
// 8 load pairs of a chunk are issued together so the L2 round trips overlap.  B is global or LDS.
template <int LANES, typename PB>
__device__ __forceinline__ double dot_range(const double* __restrict__ a, PB b, int lo, int hi, int lane) {
    double acc = 0.0;
    for (int base = lo; base < hi; base += LANES * 8) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + lane + LANES * u;
            const bool ok = i < hi;
            av[u] = ok ? a[i] : 0.0;
            bv[u] = ok ? b[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(av[u], bv[u], acc);
    }
    return acc;
}

// two such sums at once over a full wave (both sets of loads in flight together)
template <typename PB>
__device__ __forceinline__ void dot_pair(const double* __restrict__ a0, PB b0, const double* __restrict__ a1, PB b1,
                                         int n, int lane, double& s0, double& s1) {
    double acc0 = 0.0, acc1 = 0.0;
    for (int base = 0; base < n; base += 64 * 8) {
        double av0[8], bv0[8], av1[8], bv1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + lane + 64 * u;
            const bool ok = i < n;
            av0[u] = ok ? a0[i] : 0.0;
            av1[u] = ok ? a1[i] : 0.0;
            bv0[u] = ok ? b0[i] : 0.0;
            bv1[u] = ok ? b1[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc0 = fma(av0[u], bv0[u], acc0); acc1 = fma(av1[u], bv1[u], acc1); }
    }
    s0 = acc0; s1 = acc1;
}

// ------------------------------------------------------------------------------------------------
// (256 x 144) x Lp correlation tile on v_mfma_f32_16x16x4_f32
// ------------------------------------------------------------------------------------------------
//   tile   : global, packed bank tile [nslab][NT][SLAB][16] f32 (rows >= L and pad columns zero)
//   masks  : LDS, masks[s] bit ni set <=> block (slab s, column tile ni) holds a non-zero
//   r32row : LDS, &r32[row_block + 64*wave + (lane & 15) + (lane >> 4)]
//   slab   : LDS, 2 x SLABF floats
// acc[mi][ni][reg] = C[row = 64*wave + 16*mi + 4*(lane>>4) + reg][col = 16*ni + (lane&15)]
//
// Staging: the blocks of slab s+1 are copied global -> LDS by global_load_lds_dwordx4 (1 KiB per
// wave-instruction, no VGPRs), issued before the MFMAs of slab s and retired by the fence of the
// slab's closing barrier.  Column tiles are sorted by support, so for slab s only tiles
// ni >= ctz(masks[s]) are staged and multiplied.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void stage_blocks(const float* gslab, float* lslab, int ni0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (2 * NT + 3) / 4; ++i) {
        const int piece = 2 * ni0 + wave + 4 * i;                     // wave-uniform, 1 KiB each
        if (piece < 2 * NT)
            __builtin_amdgcn_global_load_lds((gptr_t)(gslab + piece * 256 + lane * 4),
                                             (lptr_t)(lslab + piece * 256), 16, 0, 0);
    }
}

// One slab = 32 bank rows = 8 k-steps.  The wave's Hankel fragments of the slab are 20 distinct LDS
// words (fragment (mi, ks) is word 16*mi + 4*ks of the residual window), loaded once and reused by
// every column tile; the B fragments of column tile ni+1 are prefetched under the 32 MFMAs of tile ni.
// Column tiles below ni0 (all zero padding in this slab) are skipped with wave-uniform branches; the
// 4 accumulators of a column tile rotate, so no MFMA waits on its own result.
__device__ __forceinline__ void corr_tile_gemm(const float* __restrict__ tile, const uint32_t* masks, int nslab,
                                               const float* r32row, float* slab, f32x4 (&acc)[MT][NT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint32_t m = __builtin_amdgcn_readfirstlane(masks[0]);
    if (m) stage_blocks(tile, slab, __builtin_ctz(m), wave, lane);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        uint32_t mn = 0;
        if (s + 1 < nslab) {
            mn = __builtin_amdgcn_readfirstlane(masks[s + 1]);
            if (mn) stage_blocks(tile + (size_t)(s + 1) * SLABF, slab + ((s + 1) & 1) * SLABF, __builtin_ctz(mn), wave, lane);
        }
        if (m) {
            const int ni0 = __builtin_ctz(m);
            const float* ra = r32row + s * SLAB;
            const float* sb = slab + (s & 1) * SLABF + lq * 16 + lr;
            float af[4 * MT + 4], bf[2][8];
#pragma unroll
            for (int q = 0; q < 4 * MT + 4; ++q) af[q] = ra[4 * q];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                if (ni >= ni0) {
                    if (ni == ni0) {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) bf[ni & 1][ks] = sb[ni * BLKF + 64 * ks];
                    }
                    if (ni + 1 < NT) {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) bf[(ni + 1) & 1][ks] = sb[(ni + 1) * BLKF + 64 * ks];
                    }
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * mi + ks], bf[ni & 1][ks],
                                                                              acc[mi][ni], 0, 0, 0);
                }
            }
        }
        m = mn;
        __syncthreads();
    }
}

// Remainder row block (at most 16 valid positions, e.g. rows 256..259 of P = 260): one 16-row tile,
// the column tiles dealt round-robin to the 4 waves (wave w owns ni = w, w+4, w+8), so the pass costs
// about 1/16 of a full unit instead of a whole one.  Only acc[0][owned ni] is written.
__device__ __forceinline__ void corr_rem_gemm(const float* __restrict__ tile, const uint32_t* masks, int nslab,
                                              const float* r32row, float* slab, f32x4 (&acc)[MT][NT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[0][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t m = __builtin_amdgcn_readfirstlane(masks[0]);
    if (m) stage_blocks(tile, slab, __builtin_ctz(m), wave, lane);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        uint32_t mn = 0;
        if (s + 1 < nslab) {
            mn = __builtin_amdgcn_readfirstlane(masks[s + 1]);
            if (mn) stage_blocks(tile + (size_t)(s + 1) * SLABF, slab + ((s + 1) & 1) * SLABF, __builtin_ctz(mn), wave, lane);
        }
        if (m) {
            const int ni0 = __builtin_ctz(m);
            const float* ra = r32row + s * SLAB;
            const float* sb = slab + (s & 1) * SLABF + lq * 16 + lr;
            float af[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) af[ks] = ra[4 * ks];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                if ((ni & 3) == wave && ni >= ni0) {
                    float bf[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) bf[ks] = sb[ni * BLKF + 64 * ks];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks], acc[0][ni], 0, 0, 0);
                }
            }
        }
        m = mn;
        __syncthreads();
    }
}

// ---- the same product on the fp16 matrix cores (16x the fp32 rate): a FILTER, not the answer ---------------
// acc = sum_l a16[k][l] * r16[p + l] with r16 = fl16(r / ||r||), v_mfma_f32_16x16x32_f16 (one per slab, row tile and
// column tile), fp32 accumulation, same accumulator layout as corr_tile_gemm.  The caller widens its candidate
// band by the fp16 rounding bound and re-scores in float64.
// LDS: the residual window in R16_COPIES shifted fp16 copies (build_r16), the slab blocks double-buffered
// (SLABF halves each) at slab16.
// `first` holds elements tid + 256 u (u < 4) of the window, zero beyond N, still in registers from the prologue
// (whole window when N <= 1024); longer signals re-read the LDS copy.
__device__ __forceinline__ void build_r16(const double* r64, int N, float inv, const double (&first)[4], _Float16* r16) {
    if (N <= 4 * WG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = threadIdx.x + WG * u;
            const _Float16 h = (_Float16)((float)first[u] * inv);
#pragma unroll
            for (int c = 0; c < R16_COPIES; ++c)
                if (e >= c && e - c < R16_CS) r16[c * R16_CS + e - c] = h;
        }
        return;
    }
    for (int i = threadIdx.x; i < R16_COPIES * R16_CS; i += WG) {
        const int c = i / R16_CS, e = i - c * R16_CS + c;
        r16[i] = (_Float16)(e < N ? (float)r64[e] * inv : 0.f);
    }
}

// Staging: a ring of RING16 slabs (9 KiB each).  The nine blocks of a slab are copied global -> LDS by
// global_load_lds_dwordx4, one block per wave instruction: wave w copies blocks w, w + 4 (and 8 for w = 0), the
// same number for every slab, so "slab s has landed" is an s_waitcnt vmcnt with a constant: the copies of the
// two slabs issued after it may still be in flight.  Issued three slabs ahead, a copy has ~3 slabs of MFMAs to
// cross the L2 latency (one slab ahead, the loop ran at that latency, not at the MFMA rate).
#define RING16 4
__device__ __forceinline__ void stage_slab16(const _Float16* gslab, _Float16* lslab, int wave, int lane) {
    __builtin_amdgcn_global_load_lds((gptr_t)(gslab + wave * BLKF + lane * 8), (lptr_t)(lslab + wave * BLKF), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gslab + (wave + 4) * BLKF + lane * 8),
                                     (lptr_t)(lslab + (wave + 4) * BLKF), 16, 0, 0);
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(gslab + 8 * BLKF + lane * 8), (lptr_t)(lslab + 8 * BLKF), 16, 0, 0);
}
// s_waitcnt vmcnt(n) only (expcnt, lgkmcnt left at their maxima): simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8
// | vmcnt[5:4] << 14
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))

template <int NI0>
__device__ __forceinline__ void slab_mfma16(const _Float16* rs, const _Float16* sb, f32x4 (&acc)[MT][NT]) {
    constexpr int AHEAD = 3;                                          // B fragments in flight ahead of the MFMAs
    f16x8 af[MT], bf[AHEAD + 1];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) af[mi] = *reinterpret_cast<const f16x8*>(rs + 16 * mi);
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
        if (NI0 + q < NT) bf[q] = *reinterpret_cast<const f16x8*>(sb + (NI0 + q) * BLKF);
#pragma unroll
    for (int ni = NI0; ni < NT; ++ni) {
        if (ni + AHEAD < NT)
            bf[(ni - NI0 + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f16x8*>(sb + (ni + AHEAD) * BLKF);
        __builtin_amdgcn_sched_barrier(0);                            // keep the reads where they are written
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[(ni - NI0) % (AHEAD + 1)], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the first RING16 - 1 slabs; issued at the very start of the node, under the prologue
__device__ __forceinline__ void gemm16_prefetch(const _Float16* __restrict__ tile, int nslab, _Float16* ring) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < RING16 - 1; ++q)
        if (q < nslab) stage_slab16(tile + (size_t)q * SLABF, ring + q * SLABF, wave, lane);
}

// one slab: wait for its copy, barrier, issue the copy three slabs ahead, multiply column tiles NI0 .. 8
template <int NI0>
__device__ __forceinline__ void slab_step16(const _Float16* __restrict__ tile, int nslab, int s, const _Float16* ra,
                                            _Float16* ring, int wave, int lane, f32x4 (&acc)[MT][NT]) {
    // slab s has landed (this wave's copies), then everybody's: the barrier also says that every wave is done
    // reading slab s - 1, whose ring slot the copy issued below overwrites
    const int later = min(RING16 - 2, nslab - 1 - s);                // slabs issued after s and still allowed in flight
    if (wave == 0) { if (later == 2) WAIT_VM(6); else if (later == 1) WAIT_VM(3); else WAIT_VM(0); }
    else           { if (later == 2) WAIT_VM(4); else if (later == 1) WAIT_VM(2); else WAIT_VM(0); }
    __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));   // lgkmcnt(0): own LDS reads done
    __builtin_amdgcn_s_barrier();
    if (s + RING16 - 1 < nslab)
        stage_slab16(tile + (size_t)(s + RING16 - 1) * SLABF, ring + ((s + RING16 - 1) % RING16) * SLABF, wave, lane);
    slab_mfma16<NI0>(ra + 32 * s, ring + (s % RING16) * SLABF + lane * 8, acc);
}

// gemm16_prefetch must have been issued (and no other vector-memory traffic may be in flight).
// Zero-padding skip: bank columns are sorted by support, so in slab s only column tiles >= ctz(masks[s]) hold
// non-zeros, and with centred supports that bound is high at both ends of the atom and 0 in the middle.  The slab
// range is cut into five runs -- tiles 6..8 only / 3..8 / all / 3..8 / 6..8 -- each a loop of straight-line code
// (choosing the variant inside one loop body costs the register allocator ~150 spills).
__device__ __forceinline__ void corr_tile_gemm16(const _Float16* __restrict__ tile, const uint32_t* masks, int nslab,
                                                 const _Float16* r16, _Float16* ring, f32x4 (&acc)[MT][NT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment (mi, slab s) of this lane: elements 64 wave + 16 mi + lr + 8 lq + 32 s ... + 7 of the window
    const _Float16* ra = r16 + (lr & 7) * R16_CS + 64 * wave + (lr & 8) + 8 * lq;

    // first tile that holds a non-zero, per slab (9 = none); runs from both ends
    int p6 = 0, p3, q3, q6 = nslab;
    auto first_tile = [&](int s) {
        const uint32_t m = __builtin_amdgcn_readfirstlane(masks[s]);
        return m ? (int)__builtin_ctz(m) : NT;
    };
    while (p6 < nslab && first_tile(p6) >= 6) ++p6;
    p3 = p6;
    while (p3 < nslab && first_tile(p3) >= 3) ++p3;
    while (q6 > p3 && first_tile(q6 - 1) >= 6) --q6;
    q3 = q6;
    while (q3 > p3 && first_tile(q3 - 1) >= 3) --q3;

    int s = 0;
    for (; s < p6; ++s) slab_step16<6>(tile, nslab, s, ra, ring, wave, lane, acc);
    for (; s < p3; ++s) slab_step16<3>(tile, nslab, s, ra, ring, wave, lane, acc);
    for (; s < q3; ++s) slab_step16<0>(tile, nslab, s, ra, ring, wave, lane, acc);
    for (; s < q6; ++s) slab_step16<3>(tile, nslab, s, ra, ring, wave, lane, acc);
    for (; s < nslab; ++s) slab_step16<6>(tile, nslab, s, ra, ring, wave, lane, acc);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// bank packing: [K][L] f64 -> column tiles sorted by support, blocked [tile][slab][ni][32][16] f32
// ------------------------------------------------------------------------------------------------
// keys[k] = first non-zero row of atom k (L for an all-zero atom); parked in the data region that
// bank_pack_kernel overwrites afterwards
__global__ void bank_keys_kernel(const double* __restrict__ bank64, int K, int L, int32_t* __restrict__ keys) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int first = L;
    for (int l = 0; l < L; ++l)
        if (bank64[(size_t)k * L + l] != 0.0) { first = l; break; }
    keys[k] = first;
}

// sorted position: later first row (= narrower centred support) first, index as tie-break; the
// ntile*144 - K padding columns go in front.  perm[t][c] = atom of column c of tile t, or -1.
__global__ void bank_rank_kernel(const int32_t* __restrict__ keys, int K, int npad, int32_t* __restrict__ meta) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int key = keys[k];
    int rank = 0;
    for (int j = 0; j < K; ++j) {
        const int kj = keys[j];
        rank += (kj > key || (kj == key && j < k)) ? 1 : 0;
    }
    const int col = npad + rank;
    meta[(col / CSC_TILE_COLS) * META_INTS + col % CSC_TILE_COLS] = k;
    for (int q = k; q < npad; q += K) meta[q] = -1;                   // pads sit in tile 0 (there can be more of them than atoms)
}

__global__ void bank_pack_kernel(const double* __restrict__ bank64, int K, int L, int nslab,
                                 const int32_t* __restrict__ meta, float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int col16 = (int)(i & 15), row = (int)((i >> 4) & (SLAB - 1));
    size_t t = i / BLKF;
    const int ni = (int)(t % NT); t /= NT;
    const int slab = (int)(t % nslab);
    const int tile = (int)(t / nslab);
    const int k = meta[tile * META_INTS + 16 * ni + col16];
    const int l = slab * SLAB + row;
    out[i] = (k >= 0 && l < L) ? (float)bank64[(size_t)k * L + l] : 0.f;
}

// the same blocks in fp16 for v_mfma_f32_16x16x32_f16: [tile][slab][ni][k-group 4][column 16][8 rows], so that a
// lane's B fragment (column lane & 15, rows 8 (lane >> 4) ... + 7) is 16 contiguous bytes and a wave reads a block
// front to back
__global__ void bank_pack16_kernel(const double* __restrict__ bank64, int K, int L, int nslab,
                                   const int32_t* __restrict__ meta, _Float16* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), col16 = (int)((i >> 3) & 15), kg = (int)((i >> 7) & 3);
    size_t t = i / BLKF;
    const int ni = (int)(t % NT); t /= NT;
    const int slab = (int)(t % nslab);
    const int tile = (int)(t / nslab);
    const int k = meta[tile * META_INTS + 16 * ni + col16];
    const int l = slab * SLAB + 8 * kg + j;
    out[i] = (k >= 0 && l < L) ? (_Float16)(float)bank64[(size_t)k * L + l] : (_Float16)0.f;
}

// The bank as a two-term fp16 number for the three-product filter of the tiled kernel (a ~ hi + lo to ~2^-22):
// hi = fl16(a 2^sa), lo = fl16(a 2^sa - hi), same block layout as bank_pack16_kernel.  The power of two 2^sa (split_scale)
// brings max_k ||a_k|| to [32, 64), which keeps the lo parts of all but the atoms' far tails out of the fp16
// subnormals; the same function gives the level kernel the scale to undo.
__device__ __host__ __forceinline__ float split_scale(float amax2) {
    const float am = sqrtf(amax2);
    if (!(am > 1e-3f && am < 1e2f)) return 1.f;
    return exp2f(floorf(log2f(64.f / am)));
}
__global__ void bank_pack16s_kernel(const double* __restrict__ bank64, int K, int L, int nslab,
                                    const int32_t* __restrict__ meta, const float* __restrict__ amax2,
                                    _Float16* __restrict__ hi, _Float16* __restrict__ lo, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), col16 = (int)((i >> 3) & 15), kg = (int)((i >> 7) & 3);
    size_t t = i / BLKF;
    const int ni = (int)(t % NT); t /= NT;
    const int slab = (int)(t % nslab);
    const int tile = (int)(t / nslab);
    const int k = meta[tile * META_INTS + 16 * ni + col16];
    const int l = slab * SLAB + 8 * kg + j;
    const double x = (k >= 0 && l < L) ? bank64[(size_t)k * L + l] * (double)split_scale(*amax2) : 0.0;
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (double)h);
}

// anorm2[k] = ||a_k||^2, amax2 = their maximum (one wave per atom)
__global__ __launch_bounds__(64) void bank_norm_kernel(const double* __restrict__ bank64, int K, int L,
                                                       float* __restrict__ amax2, double* __restrict__ anorm2) {
    const int k = blockIdx.x, lane = threadIdx.x;
    double v = 0.0;
    for (int l = lane; l < L; l += 64) { const double x = bank64[(size_t)k * L + l]; v = fma(x, x, v); }
    v = wave_sum(v);
    if (lane == 63) {
        anorm2[k] = v;
        atomicMax(reinterpret_cast<unsigned*>(amax2), __float_as_uint((float)v * 1.0000002f));   // >= 0, rounded up
    }
}

// masks[tile][slab]: bit ni set <=> the block holds a non-zero value
__global__ void bank_mask_kernel(const float* __restrict__ tiles, int nslab, int32_t* __restrict__ meta) {
    __shared__ uint32_t m;
    const int slab = blockIdx.x, tile = blockIdx.y;
    if (slab >= nslab) {                                               // the unused mask words of the tile: defined, zero
        if (threadIdx.x == 0) meta[tile * META_INTS + CSC_TILE_COLS + slab] = 0;
        return;
    }
    if (threadIdx.x == 0) m = 0;
    __syncthreads();
    const float* blk = tiles + ((size_t)tile * nslab + slab) * SLABF;
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < SLABF; i += blockDim.x)
        if (blk[i] != 0.f) mine |= 1u << (i / BLKF);
    if (mine) atomicOr(&m, mine);
    __syncthreads();
    if (threadIdx.x == 0) meta[tile * META_INTS + CSC_TILE_COLS + slab] = (int32_t)m;
}

// ------------------------------------------------------------------------------------------------
// similarity table: one workgroup per (k_sel, k); fp64 cross-correlation at every shift
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void sim_table_kernel(const double* __restrict__ bank64, int K, int L,
                                                       int P, int W, double thr,
                                                       uint32_t* __restrict__ bits) {
    extern __shared__ double sm[];
    double* a_sel = sm;            // [L]
    double* a_k = sm + L;          // [L]
    __shared__ double wpart[WG / 64];
    const int ksel = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    double ss = 0.0;
    for (int i = tid; i < L; i += WG) {
        const double v = bank64[(size_t)ksel * L + i];
        a_sel[i] = v;
        a_k[i] = bank64[(size_t)k * L + i];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 63) wpart[tid >> 6] = ss;
    __syncthreads();
    const double nrm = sqrt(wpart[0] + wpart[1] + wpart[2] + wpart[3]);
    uint32_t* row = bits + ((size_t)ksel * K + k) * W;
    const int nshift = 2 * P - 1;
    for (int base = 0; base < W * 32; base += WG) {
        const int dpi = base + tid;              // dp + P - 1
        bool sup = false;
        if (dpi < nshift) {
            const int dp = dpi - (P - 1);
            // corr = sum_j a_k[j] * a_sel[j + dp]
            const int j0 = max(0, -dp), j1 = min(L, L - dp);
            double acc = 0.0;
            for (int j = j0; j < j1; ++j) acc = fma(a_k[j], a_sel[j + dp], acc);
            sup = (acc / nrm) > thr;
        }
        const unsigned long long m = __ballot(sup);
        const int w0 = (base + (tid & ~63)) >> 5;
        if ((tid & 63) == 0) {
            if (w0 < W) row[w0] = (uint32_t)m;
            if (w0 + 1 < W) row[w0 + 1] = (uint32_t)(m >> 32);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stand-alone batched correlation (API parity: computeCorrelations)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG, 2) void correlate_kernel(const float* __restrict__ bank32t,
                                                          const int32_t* __restrict__ meta, int K, int L,
                                                          int Lp, const double* __restrict__ signals,
                                                          int N, int P, float* __restrict__ out) {
    extern __shared__ float lds[];
    __shared__ uint32_t masks[MAX_SLABS];
    const int R32 = CSC_TILE_ROWS * ((P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS) + Lp + 4;   // Lp = whole slabs
    float* slab = lds;
    float* r32 = lds + 2 * SLABF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const size_t b = blockIdx.x;
    for (int i = tid; i < R32; i += WG) r32[i] = (i < N) ? (float)signals[b * N + i] : 0.f;
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nslab = (L + SLAB - 1) / SLAB;
    const int nrb = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    for (int tl = 0; tl < ntile; ++tl) {
        __syncthreads();
        if (tid < nslab) masks[tid] = (uint32_t)meta[tl * META_INTS + CSC_TILE_COLS + tid];
        int permv[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) permv[ni] = meta[tl * META_INTS + 16 * ni + lr];
        __syncthreads();
        for (int rb = 0; rb < nrb; ++rb) {
            f32x4 acc[MT][NT];
            corr_tile_gemm(bank32t + (size_t)tl * nslab * SLABF, masks, nslab,
                           r32 + rb * CSC_TILE_ROWS + 64 * wave + lr + lq, slab, acc);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int p = rb * CSC_TILE_ROWS + 64 * wave + 16 * mi + 4 * lq + rg;
                        if (p < P && permv[ni] >= 0) out[(b * P + p) * K + permv[ni]] = acc[mi][ni][rg];
                    }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// similarity table on the matrix cores (csc_similarity_table_packed): the same bits as sim_table_kernel
// ------------------------------------------------------------------------------------------------
// sim_table_kernel takes every cross-correlation in float64 on the vector ALUs: 25 TMAC for the 9570-atom dictionary of
// config 3, 6.2 s.  Here one workgroup per selected atom treats that atom, zero-padded by P - 1 samples on both sides,
// as a signal and correlates it with the packed bank exactly like correlate_kernel does (fp32 matrix cores, the
// (256 shifts x 144 columns) tile in registers); an entry is decided by the fp32 value wherever that value is further
// from the threshold than the product's rounding bound, and recomputed the way sim_table_kernel computes it -- the same
// float64 sum in the same order -- where it is not (a few hundred of the 5 M entries of an atom).  The bits of a bank
// tile are collected in LDS and leave as whole rows.
#define SIMQ 2048                      // undecided entries of one bank tile (queue in LDS; more: the whole tile in float64)
__global__ __launch_bounds__(WG, 2) void sim_table_mfma_kernel(const float* __restrict__ bank32t, const int32_t* __restrict__ meta,
                                                               const float* __restrict__ amax2, const double* __restrict__ bank64,
                                                               int K, int L, int Lp, int P, int W, double thr,
                                                               uint32_t* __restrict__ bits) {
    extern __shared__ float lds[];
    __shared__ uint32_t masks[MAX_SLABS];
    __shared__ double wpart[WG / 64];
    __shared__ int qn;
    const int nshift = 2 * P - 1;
    const int nrb = (nshift + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const int R32 = CSC_TILE_ROWS * nrb + Lp + 4;
    float* slab = lds;                                                 // [2][SLABF]
    float* r32 = lds + 2 * SLABF;                                      // [R32]: the selected atom at offset P - 1, zeros around
    uint32_t* tbits = reinterpret_cast<uint32_t*>(r32 + ((R32 + 3) & ~3));   // [144][W]: the bits of the current bank tile
    uint32_t* queue = tbits + CSC_TILE_COLS * W;                       // [SIMQ]: column << 16 | shift index
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ksel = blockIdx.x;
    const double* asel = bank64 + (size_t)ksel * L;
    double ss = 0.0;
    for (int i = tid; i < L; i += WG) {                               // (sim_table_kernel's norm, the same sum)
        const double v = asel[i];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 63) wpart[wave] = ss;
    for (int i = tid; i < R32; i += WG) {
        const int o = i - (P - 1);
        r32[i] = ((unsigned)o < (unsigned)L) ? (float)asel[o] : 0.f;
    }
    __syncthreads();
    const double nrm = sqrt(wpart[0] + wpart[1] + wpart[2] + wpart[3]);
    // decided by the fp32 value: |fp32 product - float64 sum| <= (L + 8) 2^-23 ||a_sel|| max_k ||a_k|| (both operands
    // rounded to fp32, L fp32 accumulations), doubled
    const float T = (float)(thr * nrm);
    const float dlt = 2.f * (float)(L + 8) * 1.1920929e-07f * (float)nrm * sqrtf(*amax2) + 2.3841858e-07f * fabsf(T);
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nslab = (L + SLAB - 1) / SLAB;
    // sim_table_kernel's decision for (bank column c of the tile, shift index dpi), the same float64 sum in the same order
    auto exact = [&](int tl, int c, int dpi) {
        const int k = meta[tl * META_INTS + c];
        if (k < 0 || dpi >= nshift) return;
        const int dp = dpi - (P - 1);
        const int j0 = max(0, -dp), j1 = min(L, L - dp);
        const double* ak = bank64 + (size_t)k * L;
        double e = 0.0;
        for (int j = j0; j < j1; ++j) e = fma(ak[j], asel[j + dp], e);
        if ((e / nrm) > thr) atomicOr(&tbits[c * W + (dpi >> 5)], 1u << (dpi & 31));
    };
    for (int tl = 0; tl < ntile; ++tl) {
        __syncthreads();
        if (tid < nslab) masks[tid] = (uint32_t)meta[tl * META_INTS + CSC_TILE_COLS + tid];
        for (int i = tid; i < CSC_TILE_COLS * W; i += WG) tbits[i] = 0u;
        if (tid == 0) qn = 0;
        int permv[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) permv[ni] = meta[tl * META_INTS + 16 * ni + lr];
        __syncthreads();
        for (int rb = 0; rb < nrb; ++rb) {
            f32x4 acc[MT][NT];
            corr_tile_gemm(bank32t + (size_t)tl * nslab * SLABF, masks, nslab,
                           r32 + rb * CSC_TILE_ROWS + 64 * wave + lr + lq, slab, acc);
            // this lane's entries of (row tile mi, column tile ni): shifts dpi0 + rg, four bits of the column's row
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                if (permv[ni] < 0) continue;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int dpi0 = rb * CSC_TILE_ROWS + 64 * wave + 16 * mi + 4 * lq;
                    unsigned four = 0u, open4 = 0u;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float v = acc[mi][ni][rg];
                        four |= (v > T + dlt) ? 1u << rg : 0u;
                        open4 |= !(v > T + dlt || v < T - dlt) ? 1u << rg : 0u;       // too close to call (or not a number)
                    }
                    if (dpi0 >= nshift) continue;
                    if (dpi0 + 3 >= nshift) {                          // (the table's last shifts: nothing beyond them)
                        four &= (1u << (nshift - dpi0)) - 1u;
                        open4 &= (1u << (nshift - dpi0)) - 1u;
                    }
                    if (four) atomicOr(&tbits[(16 * ni + lr) * W + (dpi0 >> 5)], four << (dpi0 & 31));
                    while (open4) {
                        const int rg = __builtin_ctz(open4);
                        open4 &= open4 - 1;
                        const int slot = atomicAdd(&qn, 1);
                        if (slot < SIMQ) queue[slot] = ((unsigned)(16 * ni + lr) << 16) | (unsigned)(dpi0 + rg);
                    }
                }
            }
        }
        __syncthreads();
        const int nq = qn;
        if (nq <= SIMQ) {
            for (int i = tid; i < nq; i += WG) exact(tl, (int)(queue[i] >> 16), (int)(queue[i] & 0xffffu));
        } else {
            // (degenerate: thousands of entries at the threshold) the whole tile in float64
            __syncthreads();
            for (int i = tid; i < CSC_TILE_COLS * W; i += WG) tbits[i] = 0u;
            __syncthreads();
            for (int i = tid; i < CSC_TILE_COLS * nshift; i += WG) exact(tl, i / nshift, i % nshift);
        }
        __syncthreads();
        // the tile's rows out: W words per (selected atom, column's atom), contiguous
        for (int i = tid; i < CSC_TILE_COLS * W; i += WG) {
            const int c = i / W, w = i - c * W;
            const int k = meta[tl * META_INTS + c];
            if (k >= 0) bits[((size_t)ksel * K + k) * W + w] = tbits[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// float64 correlations for the public API (computeCorrelations / the rmatvec of getMaskedConvOperator)
// ------------------------------------------------------------------------------------------------
// The matrix-core kernel above returns fp32-accurate values; a caller that runs its own solver on the operator (lsqr's
// orthogonality) needs the reference's float64 (src/csc/dictionary.py:395-407 is float64 FFT arithmetic, ~1e-12).
// One workgroup per (signal, position): the window sits in LDS, one wave per atom at a time, fp64 FMA + wave sum.
__global__ __launch_bounds__(WG) void correlate64_kernel(const double* __restrict__ bank64, int K, int L,
                                                         const double* __restrict__ signals, int N, int P,
                                                         double* __restrict__ out) {
    extern __shared__ double win[];                                   // [L]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t b = blockIdx.y;
    const int p = blockIdx.x;
    for (int i = tid; i < L; i += WG) win[i] = signals[b * N + p + i];
    __syncthreads();
    for (int k = wave; k < K; k += WG / 64) {
        double v = dot_range<64>(bank64 + (size_t)k * L, win, 0, L, lane);
        v = wave_sum(v);
        if (lane == 63) out[(b * P + p) * K + k] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// batched projection on given supports (API parity: getSignalProjectionFromAtoms)
// ------------------------------------------------------------------------------------------------
struct ProjLds {
    double G[KMAX][KMAX + 1];
    double b[KMAX], x[KMAX];
    double wsum[WG / 64];
    int sup_pos[KMAX], sup_atom[KMAX];
    int cand_cnt, bad;
};

// Least squares of the LDS-resident target r64 on the placed atoms sup_pos/sup_atom[0..n): fp64 Gram
// and right-hand side by wave-wide dot products (one per wave at a time), in-place Cholesky
// G = C C^T on one lane, C y = b, C^T x = y -> sl->x.  Replaces lsqr on the masked operator
// (src/csc/mmp.py:65-67, src/csc/dictionary.py:471,528).  A column that is numerically dependent on
// the earlier ones (an activation listed twice) is decoupled and gets coefficient 0; sl->bad flags it.
// Contains workgroup barriers: every thread of the workgroup must call it.
__device__ __forceinline__ void support_ls(ProjLds* sl, int n, const double* r64,
                                           const double* __restrict__ bank64, int L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npair = n * (n + 1) / 2;
    for (int q = wave; q < n + npair; q += WG / 64) {
        double v = 0.0;
        if (q < n) {
            const double* at = bank64 + (size_t)sl->sup_atom[q] * L;
            const int p = sl->sup_pos[q];
            for (int i = lane; i < L; i += 64) v = fma(at[i], r64[p + i], v);
            v = wave_sum(v);
            if (lane == 63) sl->b[q] = v;
        } else {
            int r = q - n, i1 = 0;
            while (r > i1) { r -= i1 + 1; ++i1; }               // (i1, i2 = r), i2 <= i1
            const int i2 = r;
            const double* a1 = bank64 + (size_t)sl->sup_atom[i1] * L;
            const double* a2 = bank64 + (size_t)sl->sup_atom[i2] * L;
            const int dp = sl->sup_pos[i1] - sl->sup_pos[i2];    // a2 index = i + dp
            const int lo = max(0, -dp), hi = min(L, L - dp);
            for (int i = lo + lane; i < hi; i += 64) v = fma(a1[i], a2[i + dp], v);
            v = wave_sum(v);
            if (lane == 63) { sl->G[i1][i2] = v; sl->G[i2][i1] = v; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        bool bad = false;
        for (int c = 0; c < n; ++c) {
            const double g0 = sl->G[c][c];
            double dg = g0;
            for (int m = 0; m < c; ++m) dg -= sl->G[c][m] * sl->G[c][m];
            if (!(dg > 1e-13 * g0)) {
                bad = true;
                for (int m = 0; m < c; ++m) sl->G[c][m] = 0.0;
                for (int r = c + 1; r < n; ++r) sl->G[r][c] = 0.0;
                sl->G[c][c] = 1.0;
                sl->b[c] = 0.0;
                continue;
            }
            const double piv = sqrt(dg);
            sl->G[c][c] = piv;
            for (int r = c + 1; r < n; ++r) {
                double v = sl->G[r][c];
                for (int m = 0; m < c; ++m) v -= sl->G[r][m] * sl->G[c][m];
                sl->G[r][c] = v / piv;
            }
        }
        for (int r = 0; r < n; ++r) {
            double v = sl->b[r];
            for (int m = 0; m < r; ++m) v -= sl->G[r][m] * sl->x[m];
            sl->x[r] = v / sl->G[r][r];
        }
        for (int r = n - 1; r >= 0; --r) {
            double v = sl->x[r];
            for (int m = r + 1; m < n; ++m) v -= sl->G[m][r] * sl->x[m];
            sl->x[r] = v / sl->G[r][r];
        }
        sl->bad = bad ? 1 : 0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// batched projection on given supports (API parity: getSignalProjectionFromAtoms)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void project_kernel(const double* __restrict__ bank64, int K, int L, int P,
                                                     const double* __restrict__ signals, int N,
                                                     const int32_t* __restrict__ sig_index,
                                                     const int32_t* __restrict__ support, int kmax,
                                                     double* __restrict__ coef, double* __restrict__ mse,
                                                     double* __restrict__ approx, int32_t* stats) {
    extern __shared__ double lds_raw[];
    double* r64 = lds_raw;                                            // [N]
    ProjLds* sl = reinterpret_cast<ProjLds*>(r64 + ((N + 1) & ~1));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t m = blockIdx.x;
    const size_t s = sig_index ? (size_t)sig_index[m] : m;
    for (int i = tid; i < N; i += WG) r64[i] = signals[s * N + i];
    if (tid == 0) {
        // leading run of valid (pos, atom) pairs; an out-of-range pair ends the support
        int n = 0;
        for (; n < kmax; ++n) {
            const int p = support[(m * kmax + n) * 2], k = support[(m * kmax + n) * 2 + 1];
            if (p < 0 || p >= P || k < 0 || k >= K) break;
            sl->sup_pos[n] = p;
            sl->sup_atom[n] = k;
        }
        sl->cand_cnt = n;
    }
    __syncthreads();
    const int n = sl->cand_cnt;
    support_ls(sl, n, r64, bank64, L);
    if (tid == 0 && sl->bad && stats) atomicAdd(&stats[1], 1);
    double cs = 0.0;
    for (int i = tid; i < N; i += WG) {
        double ap = 0.0;
        for (int q = 0; q < n; ++q) {
            const int o = i - sl->sup_pos[q];
            if (o >= 0 && o < L) ap = fma(sl->x[q], bank64[(size_t)sl->sup_atom[q] * L + o], ap);
        }
        if (approx) approx[m * N + i] = ap;
        const double v = r64[i] - ap;
        cs = fma(v, v, cs);
    }
    cs = wave_sum(cs);
    if (lane == 63) sl->wsum[wave] = cs;
    __syncthreads();
    if (tid == 0 && mse) mse[m] = (sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3]) / (double)N;
    if (coef && tid < kmax) coef[m * kmax + tid] = tid < n ? sl->x[tid] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// the level kernel
// ------------------------------------------------------------------------------------------------
struct NodeLds {
    double Lf[TRI(KMAX)];          // parent's Cholesky factor, packed lower triangle, row i at TRI(i)
    double g[KMAX + 2];            // <a_new, a_q> q < d | ||a_new||^2 | <a_new, r_parent>
    double y[KMAX + 1];            // new factor row (y_0..y_{d-1}, l_dd)
    double x[KMAX + 1];            // LS solution of the parent residual on the child's support
    double coefpar[KMAX];
    double cand_val[CAND_CAP];
    double wsum[WG / 64];
    int cand_idx[CAND_CAP];
    int sup_pos[KMAX + 1], sup_atom[KMAX + 1];
    uint32_t masks[MAX_SLABS];
    int perm[CSC_TILE_COLS];       // atom of each bank column (-1 = padding)
    float wmax[WG / 64];
    int widx[WG / 64], wsupp[WG / 64];
    int cand_cnt, chosen, i32, bad, beta_ok;      // beta_ok: `beta` = <a_chosen, r> from the re-score
    double beta;
    float m32;
    PROF_FIELDS
};

// Block -> (signal, frontier index).  Workgroups whose blockIdx agree mod 8 share an XCD (and its L2):
// the F_d nodes of one signal are given consecutive such blocks, so the signal's correlations C_y, which
// each of them reads, are fetched from HBM once and found in L2 by the siblings.  Pure placement hint:
// any mapping is correct.
__device__ __forceinline__ void node_of_block(int Fd, size_t& s, int& j) {
    const unsigned b = blockIdx.x;
    const unsigned S = gridDim.x / (unsigned)Fd;
    const unsigned nfull = (S & ~7u) * (unsigned)Fd;
    if (b < nfull) {
        const unsigned xcd = b & 7u, t = b >> 3;
        s = (size_t)(t / (unsigned)Fd) * 8u + xcd;
        j = (int)(t % (unsigned)Fd);
    } else {
        s = b / (unsigned)Fd;
        j = (int)(b - (unsigned)s * (unsigned)Fd);
    }
}

// ---- pieces of a node's work shared by the register-resident and the tiled level kernels ------------

// residual -> LDS (fp64 + fp32 window copy), ancestors' (pos, atom), parent's coefficients and Cholesky
// factor.  Returns the fp32 candidate band delta = 2^-14 * ||residual|| * max_k ||a_k||.  Ends with a barrier.
__device__ __forceinline__ float node_prologue(const LevelArgs& a, NodeLds* sl, double* r64, float* r32,
                                               int Npad, int R32, const double* __restrict__ rin, size_t s, int j,
                                               double& rnorm2_out, double (&first)[4], int d) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N;
    double ss = 0.0;
    for (int i0 = 0; i0 < max(R32, Npad); i0 += 4 * WG) {              // one sweep fills both copies
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tid + WG * u;
            v[u] = (i < N) ? rin[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tid + WG * u;
            if (i < Npad) r64[i] = v[u];
            if (i < R32) r32[i] = (float)v[u];
            ss = fma(v[u], v[u], ss);
            if (i0 == 0) first[u] = v[u];                             // elements tid + 256 u, kept for build_r16
        }
    }
    if (tid < d) {
        const int e = tid + 1;                                        // level of this ancestor
        const size_t id = s * a.NN + a.ts.off[e] + (j % a.ts.pw[e]);
        sl->sup_pos[tid] = a.node_pos[id];
        sl->sup_atom[tid] = a.node_atom[id];
        sl->coefpar[tid] = rin[N + TRI(a.Kmax) + tid];               // vs the original signal (mode 1: mask values)
    }
    if (tid >= 64 && tid < 64 + TRI(d)) sl->Lf[tid - 64] = rin[N + tid - 64];
    ss = wave_sum(ss);
    if (lane == 63) sl->wsum[wave] = ss;
    __syncthreads();
    const double rnorm2 = sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3];
    rnorm2_out = rnorm2;
    return DELTA_REL * (float)sqrt(rnorm2) * sqrtf(*a.amax2);          // fp32 product: error ~ ||a|| ||r||
}

// (c) fp64 re-score of the candidates in sl->cand_idx (one wave per candidate), (d) the choice: fp64
// maximum, lowest flattened index on ties; with zero_band the first suppressed index (sl->wsupp) competes
// with value 0; on candidate overflow the fp32 arg-max (sl->widx) is taken.  Writes sl->sup_pos/atom[d].
__device__ __forceinline__ void rescore_and_choose(const LevelArgs& a, NodeLds* sl, const double* r64,
                                                   int ncand_raw, bool zero_band, int d) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                                   // see child_solve_and_write
    const int lane = tid & 63, wave = tid >> 6;
    const int K = a.K, L = a.L;
    const int ncand = min(ncand_raw, CAND_CAP);
    for (int c = 2 * wave; c < ncand; c += 2 * (WG / 64)) {          // two candidates per wave and round trip
        const int i0 = sl->cand_idx[c], i1 = sl->cand_idx[min(c + 1, ncand - 1)];
        const int p0 = i0 / K, k0 = i0 - p0 * K, p1 = i1 / K, k1 = i1 - p1 * K;
        double v0, v1;
        dot_pair(a.bank64 + (size_t)k0 * L, r64 + p0, a.bank64 + (size_t)k1 * L, r64 + p1, L, lane, v0, v1);
        v0 = wave_sum(v0);
        v1 = wave_sum(v1);
        if (lane == 63) {
            sl->cand_val[c] = v0;
            if (c + 1 < ncand) sl->cand_val[c + 1] = v1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int chosen;
        sl->beta_ok = 0;
        if (ncand_raw > CAND_CAP) {
            chosen = min(min(sl->widx[0], sl->widx[1]), min(sl->widx[2], sl->widx[3]));
            if (a.stats) atomicAdd(&a.stats[0], 1);
        } else {
            double bestv = ncand ? sl->cand_val[0] : -INFINITY; int besti = ncand ? sl->cand_idx[0] : 0x7fffffff;
            for (int c = 1; c < ncand; ++c) {
                const double v = sl->cand_val[c]; const int i = sl->cand_idx[c];
                if (v > bestv || (v == bestv && i < besti)) { bestv = v; besti = i; }
            }
            if (zero_band) {
                int z = 0x7fffffff;
                for (int w = 0; w < WG / 64; ++w) z = min(z, sl->wsupp[w]);
                if (z != 0x7fffffff && (0.0 > bestv || (0.0 == bestv && z < besti))) { besti = z; bestv = NAN; }
            }
            chosen = besti;
            if (ncand && bestv == bestv) { sl->beta = bestv; sl->beta_ok = 1; }   // <a_chosen, r>, reused by the solve
        }
        if (a.stats) atomicAdd(&a.stats[2], ncand);
        // all-NaN correlations (NaN / Inf in the signal): np.argmax of an all-NaN array is 0 (src/csc/mmp.py:227)
        if ((unsigned)chosen >= (unsigned)(a.P * K)) { chosen = 0; sl->beta_ok = 0; }
        sl->chosen = chosen;
        sl->sup_pos[d] = chosen / K;
        sl->sup_atom[d] = chosen - (chosen / K) * K;
    }
    __syncthreads();
}

// Child jc of the node: least squares of the parent residual on the child's support by extending the
// parent's Cholesky factor with one row, child residual + factor -> workspace (`rout_base`, the level's output
// buffer, or nullptr), MSE, coefficients, (pos, atom) -> node tables.
// `carry` (the chain kernel: the workgroup goes on with this child): the child's residual replaces the parent's in
// r64 (in place: every element is read and written by the same thread), its factor row and coefficients join
// sl->Lf / sl->coefpar; the child's sum of squares is left in sl->wsum[0..3].
__device__ __forceinline__ void child_solve_and_write(const LevelArgs& a, NodeLds* sl, double* r64,
                                                      size_t s, int jc, size_t cid, int d, double* rout_base,
                                                      bool carry) {
    // the thread index goes through an opaque move: otherwise the 64-bit per-lane address parts derived from it are
    // hoisted above the node's GEMM loops and live (spilled to scratch, i.e. to HBM) until here
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int N = a.N, L = a.L;
    const int p_sel = sl->sup_pos[d], k_sel = sl->sup_atom[d];
    const int Fc = a.ts.F[d + 1];
    if (a.mode == 1) {
        // ---- plain matching pursuit (src/csc/dictionary.py:787-805) ----------------------------------------
        // activation_mask[idx] += <a_sel, r>; approx = conv(mask * mask): the reference applies the mask to
        // itself (:803), so an activation weighs mask^2.  node_coef holds the mask value per support slot
        // (first occurrence of a re-selected activation accumulates, later slots stay 0).
        {
            const double* an = a.bank64 + (size_t)k_sel * L;
            double v = dot_range<64>(an, r64 + p_sel, 0, L, lane);
            v = wave_sum(v);
            if (tid == 63) {
                int slot = d;
                for (int q = d - 1; q >= 0; --q)
                    if (sl->sup_pos[q] == p_sel && sl->sup_atom[q] == k_sel) slot = q;
                const double m_old = slot < d ? sl->coefpar[slot] : 0.0;
                const double m_new = m_old + v;
                for (int q = 0; q <= d; ++q) sl->x[q] = q < d ? sl->coefpar[q] : 0.0;   // mask values out
                sl->x[slot] = m_new;
                sl->g[0] = m_new * m_new - m_old * m_old;            // weight change of the placed atom
            }
        }
        __syncthreads();
        const double dw = sl->g[0];
        double* rout = rout_base ? rout_base + (s * Fc + jc) * (size_t)a.NS : nullptr;
        const double* an = a.bank64 + (size_t)k_sel * L;
        double cs = 0.0;
        for (int n = tid; n < N; n += WG) {
            const int o = n - p_sel;
            const double v = r64[n] - (((unsigned)o < (unsigned)L) ? dw * an[o] : 0.0);
            if (rout) rout[n] = v;
            if (carry) r64[n] = v;
            cs = fma(v, v, cs);
        }
        cs = wave_sum(cs);
        if (lane == 63) sl->wsum[wave] = cs;
        __syncthreads();
        if (tid == 0) {
            a.node_pos[cid] = p_sel;
            a.node_atom[cid] = k_sel;
            a.node_mse[cid] = (sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3]) / (double)N;
        }
        if (tid < a.Kmax) {
            const double cv = tid <= d ? sl->x[tid] : 0.0;
            if (a.node_coef) a.node_coef[cid * a.Kmax + tid] = cv;
            if (rout) rout[N + TRI(a.Kmax) + tid] = cv;
            if (carry && tid <= d) sl->coefpar[tid] = cv;
        }
        return;
    }
    // ---- one new Gram column: <a_new, a_q> for the d ancestors (fp64 dots, one per wave); ||a_new||^2 comes from
    // the bank's table and <a_new, r_parent> from the re-score that chose the atom (recomputed here when the choice
    // was not a re-scored candidate) -----------------------------------------------------------------------------
    {
        const double* an = a.bank64 + (size_t)k_sel * L;
        const int nd = d + (sl->beta_ok ? 0 : 1);
        for (int q = wave; q < nd; q += WG / 64) {
            double v;
            if (q < d) {                                             // <a_new @ p_sel, a_q @ p_q>
                const double* aq = a.bank64 + (size_t)sl->sup_atom[q] * L;
                const int dp = p_sel - sl->sup_pos[q];               // a_q index = i + dp
                v = dot_range<64>(an, aq + dp, max(0, -dp), min(L, L - dp), lane);
            } else {                                                 // <a_new, r_parent>
                v = dot_range<64>(an, r64 + p_sel, 0, L, lane);
            }
            v = wave_sum(v);
            if (lane == 63) sl->g[q < d ? q : d + 1] = v;
        }
        if (tid == 0) {
            sl->g[d] = a.anorm2[k_sel];
            if (sl->beta_ok) sl->g[d + 1] = sl->beta;
        }
    }
    __syncthreads();
    PROF_MARK(6);
    if (tid == 0) {
        // factor row: Lf y = g, l_dd^2 = g_dd - y.y.  A dead column (an activation selected twice,
        // src/csc/mmp.py:56 sets the same mask bit again) has diagonal 0 and is skipped everywhere.
        double yy = 0.0;
        for (int i = 0; i < d; ++i) {
            const double dg = sl->Lf[TRI(i) + i];
            double v = sl->g[i];
            for (int m = 0; m < i; ++m) v = fma(-sl->Lf[TRI(i) + m], sl->y[m], v);
            v = (dg != 0.0) ? v * (1.0 / dg) : 0.0;                 // (reciprocal first: bit for bit chain_tail_fast's stored 1 / l_ii)
            sl->y[i] = v;
            yy = fma(v, v, yy);
        }
        const double gdd = sl->g[d], sdd = gdd - yy;
        const bool bad = !(sdd > 1e-13 * gdd);
        double xd = 0.0;
        if (bad) {
            for (int i = 0; i < d; ++i) sl->y[i] = 0.0;
            sl->y[d] = 0.0;
        } else {
            sl->y[d] = sqrt(sdd);
            xd = sl->g[d + 1] / sdd;                             // (beta / l_dd) / l_dd
        }
        // Lf^T x = z with z = (0, ..., 0, beta / l_dd): the parent residual is orthogonal to the
        // ancestors' atoms, so only the new right-hand side entry is non-zero
        sl->x[d] = xd;
        for (int r = d - 1; r >= 0; --r) {
            const double dg = sl->Lf[TRI(r) + r];
            double v = -sl->y[r] * xd;
            for (int m = r + 1; m < d; ++m) v = fma(-sl->Lf[TRI(m) + r], sl->x[m], v);
            sl->x[r] = (dg != 0.0) ? v * (1.0 / dg) : 0.0;
        }
        if (bad && a.stats) atomicAdd(&a.stats[1], 1);
    }
    __syncthreads();
    PROF_MARK(7);
    // ---- child residual, factor, MSE ------------------------------------------------------------------
    double* rout = rout_base ? rout_base + (s * Fc + jc) * (size_t)a.NS : nullptr;
    double cs = 0.0;
    for (int n0 = 0; n0 < N; n0 += 3 * WG) {
        double v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int n = n0 + tid + WG * u;
            v[u] = n < N ? r64[n] : 0.0;
        }
        for (int i0 = 0; i0 <= d; i0 += 4) {                          // four atoms' loads in flight together
            double av[4][3], xi[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = min(i0 + ii, d);
                const double* ai = a.bank64 + (size_t)sl->sup_atom[i] * L;
                const int pi = sl->sup_pos[i];
                xi[ii] = (i0 + ii <= d) ? sl->x[i] : 0.0;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int o = n0 + tid + WG * u - pi;
                    av[ii][u] = ((unsigned)o < (unsigned)L && i0 + ii <= d) ? ai[o] : 0.0;
                }
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int u = 0; u < 3; ++u) v[u] = fma(-xi[ii], av[ii][u], v[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int n = n0 + tid + WG * u;
            if (n < N) {
                if (rout) rout[n] = v[u];
                if (carry) r64[n] = v[u];
                cs = fma(v[u], v[u], cs);
            }
        }
    }
    if (rout && tid < TRI(d + 1)) rout[N + tid] = tid < TRI(d) ? sl->Lf[tid] : sl->y[tid - TRI(d)];
    if (carry && tid <= d) sl->Lf[TRI(d) + tid] = sl->y[tid];
    cs = wave_sum(cs);
    if (lane == 63) sl->wsum[wave] = cs;
    __syncthreads();
    if (tid == 0) {
        a.node_pos[cid] = p_sel;
        a.node_atom[cid] = k_sel;
        a.node_mse[cid] = (sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3]) / (double)N;
    }
    if (tid < a.Kmax) {
        const double cv = tid <= d ? sl->x[tid] + (tid < d ? sl->coefpar[tid] : 0.0) : 0.0;
        if (a.node_coef) a.node_coef[cid * a.Kmax + tid] = cv;
        if (rout) rout[N + TRI(a.Kmax) + tid] = cv;
        if (carry && tid <= d) sl->coefpar[tid] = cv;
    }
    PROF_MARK(8);
}

// ---- the correlations of a node held in registers --------------------------------------------------------
// f32x4 acc[MT][NT] (144 values per lane) holds the (256 positions x 144 bank columns) correlation tile as the
// 16x16 MFMAs leave it: position 64*wave + 16*mi + 4*(lane>>4) + rg, bank column 16*ni + (lane&15).

// entries outside the (P x K) matrix never win: positions >= P (last row tile only) and padding columns
__device__ __forceinline__ void mask_outside(f32x4 (&acc)[MT][NT], const int (&permv)[NT], int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prow = 64 * wave + 4 * (lane >> 4);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
        if (64 * wave + 16 * mi + 15 >= P) {                         // wave-uniform
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (prow + 16 * mi + rg >= P) acc[mi][ni][rg] = -INFINITY;
        }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
        if (__ballot(permv[ni] < 0)) {                               // wave-uniform
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (permv[ni] < 0) acc[mi][ni][rg] = -INFINITY;
        }
}

// multiply-by-zero suppression of the similarity disk of the placed atom (p_sel, k_sel), src/csc/mmp.py:214-218:
// the atom's bit row of the similarity table is staged in LDS (aliasing the slab buffers), hit entries become
// the value-0 marker.  Contains a barrier.
__device__ __forceinline__ void suppress_similar(const LevelArgs& a, NodeLds* sl, uint32_t* simrow, int p_sel, int k_sel,
                                                 const int (&permv)[NT], f32x4 (&acc)[MT][NT], bool& lane_supp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.W, K = a.K, P = a.P;
    int W1 = a.W + 1;                                                 // LDS rows padded by one word
    asm volatile("" : "+s"(W1));          // opaque: the nine row offsets are not hoisted out of the sibling loop (spills)
    const uint32_t* g = a.sim_bits + (size_t)k_sel * K * W;
    __syncthreads();
    PROF_MARK(10);
    // nine words per lane requested together (the asm keeps them live at one point: next to the 144 accumulators the
    // compiler would otherwise request one, wait, store, nine L2 round trips in a row)
    for (int base = 0; base < K * W; base += 9 * WG) {
        uint32_t v[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = base + tid + WG * u;
            v[u] = i < K * W ? g[i] : 0u;
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                          "+v"(v[8]));
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = base + tid + WG * u;
            if (i < K * W) simrow[(i / W) * W1 + (i % W)] = v[u];
        }
    }
    __syncthreads();
    PROF_MARK(11);
    // this lane's 16 entries of a column tile are bits dp0 + 16*mi + rg of the column's row, dp0 >= 0: a window of
    // 52 bits, i.e. three words per column tile (one LDS read per entry cost a full LDS latency each, 144 times).
    // The similarity disk of an atom covers a few hundred of the 35 k entries: a column tile in which no lane's window
    // holds a set bit is skipped as a whole (wave-uniform), and where there are hits an entry's test is one shift of
    // the lane's window by a constant.
    const int dp0 = 64 * wave + 4 * (lane >> 4) - p_sel + P - 1;
    const int w0 = dp0 >> 5, b0 = dp0 & 31;
    bool any = false;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const uint32_t* row = simrow + max(permv[ni], 0) * W1;
        const uint32_t wd0 = row[min(w0, W - 1)], wd1 = row[min(w0 + 1, W - 1)], wd2 = row[min(w0 + 2, W - 1)];
        const uint64_t lo64 = (uint64_t)wd0 | ((uint64_t)wd1 << 32);
        const uint64_t win = ((lo64 >> b0) | (b0 ? (uint64_t)wd2 << (64 - b0) : 0ull)) & ((1ull << 52) - 1);   // bits 16 mi + rg
        if (__ballot(win != 0ull) == 0ull) continue;                 // nothing of this column tile is suppressed
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const bool hit = ((win >> (16 * mi + rg)) & 1ull) && acc[mi][ni][rg] > -INFINITY;
                acc[mi][ni][rg] = hit ? __uint_as_float(SUPP_BITS) : acc[mi][ni][rg];
                any |= hit;
            }
    }
    lane_supp |= any;
}

// Selection of the node's children + their least squares, on the correlations held in acc
// (reference src/csc/mmp.py:210-231 and :61-70).  `band` is the width below the maximum inside which an entry
// may be the float64 maximum: every such entry is re-scored in float64.
// EXACT = false: acc comes from the fp16 filter (wide band); returns false if the node cannot be settled that way
// (more than CAND_CAP entries in the band, or the value 0 of suppressed entries inside it): the caller then
// repeats the node from the start on the fp32 correlations, with identical results.
template <bool EXACT>
__device__ __forceinline__ bool select_children(const LevelArgs& a, NodeLds* sl, double* r64, uint32_t* simrow,
                                                size_t s, int j, int nch, float band, f32x4 (&acc)[MT][NT], int d,
                                                double* rout_base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15;
    const int K = a.K, P = a.P;
    const int stride = a.ts.pw[d];
    const int prow = 64 * wave + 4 * (lane >> 4);  // + 16*mi + rg: position of an entry
    int permv[NT];                                // atom of this lane's column in tile ni (-1 = padding)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) permv[ni] = sl->perm[16 * ni + lr];
    mask_outside(acc, permv, P);

    bool lane_supp = false;                       // this lane holds a suppressed (value 0) entry
    for (int t = 0; t < nch; ++t) {
        // multiply-by-zero suppression by the sibling selected last
        if (t > 0) suppress_similar(a, sl, simrow, sl->sup_pos[d], sl->sup_atom[d], permv, acc, lane_supp);
        __syncthreads();
        PROF_MARK(2);
        // (a) fp32 maximum.  Suppressed entries carry a quiet NaN, which fmaxf (maxNum) skips; their
        //     value 0 enters through lane_supp.
        float lmax = -INFINITY;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) lmax = fmaxf(lmax, acc[mi][ni][rg]);
        float bv = wave_max(lane_supp ? fmaxf(lmax, 0.f) : lmax);
        if (lane == 63) sl->wmax[wave] = bv;
        if (tid == 0) sl->cand_cnt = 0;
        __syncthreads();
        const float m32 = fmaxf(fmaxf(sl->wmax[0], sl->wmax[1]), fmaxf(sl->wmax[2], sl->wmax[3]));
        const float tau = m32 - band;
        const bool zero_band = (0.f >= tau);
        // the value 0 of suppressed entries may compete: exact path (nothing is suppressed for the first child)
        if (!EXACT && zero_band && t > 0) return false;
        // (b) every unsuppressed entry inside the error band of the maximum is a candidate (NaN >= tau is
        //     false).  Only the few lanes whose own maximum reaches the band look at their 144 entries: a
        //     dense, branch-free bit mask first (straight-line code with a branch per entry made the
        //     instruction fetch, not the ALU, the cost of this phase), then a short loop over the set bits.
        //     Entry e = (4*mi + rg) * 9 + ni.
        int Kb = K;                                // opaque copy: keeps index arithmetic out of the hot path
        asm volatile("" : "+v"(Kb));
        if (lmax >= tau) {
            uint32_t mk[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const int e = (4 * mi + rg) * NT + ni;
                        mk[e >> 5] |= (acc[mi][ni][rg] >= tau) ? (1u << (e & 31)) : 0u;
                    }
#pragma unroll
            for (int w = 0; w < 5; ++w) {
                uint32_t mm = mk[w];
                while (mm) {
                    const int e = 32 * w + __builtin_ctz(mm);
                    mm &= mm - 1;
                    const int r = e / NT, ni = e - r * NT;
                    const int idx = (prow + 16 * (r >> 2) + (r & 3)) * Kb + sl->perm[16 * ni + lr];
                    const int slot = atomicAdd(&sl->cand_cnt, 1);
                    if (slot < CAND_CAP) sl->cand_idx[slot] = idx;
                }
            }
        }
        if (EXACT && zero_band) {                  // degenerate: the value 0 of the suppressed entries competes
            int first_supp = 0x7fffffff;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        if (is_supp(acc[mi][ni][rg]))
                            first_supp = min(first_supp, (prow + 16 * mi + rg) * Kb + permv[ni]);
            first_supp = wave_min(first_supp);
            if (lane == 0) sl->wsupp[wave] = first_supp;
        }
        __syncthreads();
        PROF_MARK(3);
        const int ncand_raw = sl->cand_cnt;
        if (!EXACT && ncand_raw > CAND_CAP) return false;
        if (EXACT && ncand_raw > CAND_CAP) {
            // overflow (never seen on real data): fall back to the fp32 arg-max, lowest index
            int bi = 0x7fffffff;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const float x = acc[mi][ni][rg];
                        const float e = is_supp(x) ? 0.f : x;
                        if (e == m32) bi = min(bi, (prow + 16 * mi + rg) * Kb + permv[ni]);
                    }
            bi = wave_min(bi);
            if (lane == 0) sl->widx[wave] = bi;
        }
        rescore_and_choose(a, sl, r64, ncand_raw, EXACT && zero_band, d);   // the filter gets here only with nothing suppressed
        PROF_MARK(5);
        const int jc = j + t * stride;                               // child's index in level d+1
        const size_t cid = s * a.NN + a.ts.off[d + 1] + jc;
        if (a.picks_only) {
            // the chain kernel, launched behind, builds this child itself (residual, factor, MSE) from the signal: only
            // the pick goes to HBM -- (pos, atom) and <atom, signal>, the latter parked in the child's MSE slot.  That
            // inner product is the re-scored value; a choice that was not a re-scored candidate computes it the way
            // child_solve_and_write does (one wave, same sum)
            if (!sl->beta_ok && wave == 0) {
                double v = dot_range<64>(a.bank64 + (size_t)sl->sup_atom[d] * a.L, r64 + sl->sup_pos[d], 0, a.L, lane);
                v = wave_sum(v);
                if (lane == 63) sl->beta = v;
            }
            __syncthreads();
            if (tid == 0) {
                a.node_pos[cid] = sl->sup_pos[d];
                a.node_atom[cid] = sl->sup_atom[d];
                a.node_mse[cid] = sl->beta;
            }
        } else {
            child_solve_and_write(a, sl, r64, s, jc, cid, d, rout_base, false);
        }
        __syncthreads();
    }
    return true;
}

__device__ __forceinline__ int children_of(const LevelArgs& a, int j) {
    const int d = a.depth, Fc = a.ts.F[d + 1], stride = a.ts.pw[d];
    int nch = 0;
    for (int r = 0; r < a.connections; ++r) {
        if ((long long)j + (long long)r * stride < Fc) ++nch; else break;
    }
    return nch;
}

// ---- the level kernel ---------------------------------------------------------------------------------------
// One workgroup per node of the level: correlations of the node's residual with every shifted atom on the
// matrix cores, the children's selection on the register-resident result, their least squares.
//
// The correlations are first taken in fp16 (v_mfma_f32_16x16x32_f16, 16x the fp32 rate) from the residual scaled
// to unit norm.  That product is only a FILTER: every entry within twice the fp16 rounding bound of the computed
// maximum (typically 5-15 entries) is re-scored in float64 against the residual, so the selection is the
// float64 arg-max, as it is when the fp32 product is re-scored over its own (narrow) band.  A node the filter
// cannot settle -- more than CAND_CAP entries in the band, the value 0 of suppressed entries inside the band,
// NaN / Inf / extreme scales -- multiplies again in fp32 (v_mfma_f32_16x16x4_f32) in the same workgroup and
// repeats the selection from the start, with identical results.
__global__ __launch_bounds__(WG, 2) void mmp_level_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = a.N, L = a.L, Lp = a.Lp, d = a.depth;

    // ---- LDS carve-up ---------------------------------------------------------------------------
    const int Npad = (N + 3) & ~3;
    const int R32 = CSC_TILE_ROWS + Lp + 4;
    float* slab = reinterpret_cast<float*>(lds_raw);                  // [2][SLABF] floats = [RING16][SLABF] halves
    uint32_t* simrow = reinterpret_cast<uint32_t*>(slab);             // aliases the slabs
    double* r64 = lds_raw + SLABF;                                    // [Npad]   (2*SLABF floats = SLABF doubles)
    float* r32 = reinterpret_cast<float*>(r64 + Npad);                // [R32]
    _Float16* r16 = reinterpret_cast<_Float16*>(r32 + ((R32 + 3) & ~3));           // [R16_COPIES][R16_CS]
    NodeLds* sl = reinterpret_cast<NodeLds*>(r16 + R16_COPIES * R16_CS);

    const int Fd = a.ts.F[d];
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    const int nch = children_of(a, j);
    if (tid >= 192 && tid < 192 + MAX_SLABS) sl->masks[tid - 192] = (uint32_t)a.meta[CSC_TILE_COLS + tid - 192];
    if (tid < CSC_TILE_COLS) sl->perm[tid] = a.meta[tid];

    PROF_INIT(0);
    // the first bank slabs of the fp16 product start their way L2 -> LDS now, under the prologue
    if (a.bank16) gemm16_prefetch(a.bank16, (L + SLAB - 1) / SLAB, reinterpret_cast<_Float16*>(slab));
    const double* rin = a.resid_in + (d == 0 ? s * (size_t)N : (s * Fd + j) * (size_t)a.NS);
    double rnorm2, first[4];
    const float delta = node_prologue(a, sl, r64, r32, Npad, R32, rin, s, j, rnorm2, first, d);
    PROF_MARK(0);
    f32x4 acc[MT][NT];
    // ---- correlations on the fp16 matrix cores, in units of ||r||, as a filter ------------------------------
    // Rounding bound of an entry, with am = max_k ||a_k||:
    //   both operands rounded to fp16 (2 x 2^-11 relative), fp32 accumulation over L terms (2^-14), cross terms:
    //       (2^-10 + 2^-14 + 2^-20) am
    //   an operand that is subnormal in fp16 is off by up to 2^-25 absolute instead, over a window of L terms:
    //       2^-25 sqrt(L) (1 + am)
    const float am2 = *a.amax2, am = sqrtf(am2);
    const float eps16 = (9.765625e-04f + DELTA_REL + 1e-6f) * am + 2.9802322e-08f * sqrtf((float)L) * (1.f + am);
    if (a.bank16 && am2 > 1e-6f && am2 < 1e4f && rnorm2 > 1e-30 && rnorm2 < 1e30) {
        _Float16* ring = reinterpret_cast<_Float16*>(slab);
        build_r16(r64, N, (float)(1.0 / sqrt(rnorm2)), first, r16);
        __syncthreads();
        corr_tile_gemm16(a.bank16, sl->masks, (L + SLAB - 1) / SLAB, r16, ring, acc);
        PROF_MARK(1);
        if (select_children<false>(a, sl, r64, simrow, s, j, nch, 2.f * eps16, acc, d, a.resid_out)) {
            __syncthreads();
            PROF_MARK(9);
            return;
        }
        if (tid == 0 && a.stats) atomicAdd(&a.stats[3], 1);
        __syncthreads();
    }
    // ---- correlations on the fp32 matrix cores ------------------------------------------------------
    corr_tile_gemm(a.bank32t, sl->masks, (L + SLAB - 1) / SLAB, r32 + 64 * wave + lr + lq, slab, acc);
    PROF_MARK(1);
    select_children<true>(a, sl, r64, simrow, s, j, nch, delta, acc, d, a.resid_out);
    __syncthreads();
    PROF_MARK(9);
}

// ---- follow-up of the lean kernel: the chains it gave up, in fp32 -----------------------------------------------
// Usually nothing is flagged, so the launch should cost as little as possible: a workgroup looks at FOLLOW_NODES
// consecutive chains (a sixteenth of the workgroups to dispatch, their flags in one load) and runs the flagged
// ones one after the other: flag e + 1 = from level e down to the leaves, level by level through the workspace
// like the per-level launches (the lean kernel left the residual, factor and coefficients of the level-e node in
// its row of ws[e & 1]).
#define FOLLOW_NODES 16
__global__ __launch_bounds__(WG, 2) void mmp_level_followup_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = a.N, L = a.L, Lp = a.Lp;
    const int Npad = (N + 3) & ~3;
    const int R32 = CSC_TILE_ROWS + Lp + 4;
    float* slab = reinterpret_cast<float*>(lds_raw);
    uint32_t* simrow = reinterpret_cast<uint32_t*>(slab);
    double* r64 = lds_raw + SLABF;
    float* r32 = reinterpret_cast<float*>(r64 + Npad);
    _Float16* r16 = reinterpret_cast<_Float16*>(r32 + ((R32 + 3) & ~3));           // unused here; same carve-up
    NodeLds* sl = reinterpret_cast<NodeLds*>(r16 + R16_COPIES * R16_CS);
    const int Fd = a.ts.F[a.depth];                                   // the same on every level of the chain
    const size_t total = (size_t)a.S * Fd;
    bool meta_loaded = false;
    // the workgroup's 16 flag bytes in one round trip (the buffer is padded to a multiple of 16 bytes)
    const uint64_t* fw = reinterpret_cast<const uint64_t*>(a.flags + (size_t)blockIdx.x * FOLLOW_NODES);
    const uint64_t f0 = fw[0], f1 = fw[1];
    if (!(f0 | f1)) return;
    for (int q = 0; q < FOLLOW_NODES; ++q) {
        const size_t nl = (size_t)blockIdx.x * FOLLOW_NODES + q;
        if (nl >= total) break;
        const int fl = (int)(((q < 8 ? f0 : f1) >> (8 * (q & 7))) & 0xff);
        if (!fl) continue;                                            // workgroup-uniform
        if (!meta_loaded) {
            if (tid >= 192 && tid < 192 + MAX_SLABS) sl->masks[tid - 192] = (uint32_t)a.meta[CSC_TILE_COLS + tid - 192];
            if (tid < CSC_TILE_COLS) sl->perm[tid] = a.meta[tid];
            meta_loaded = true;
        }
        const size_t s = nl / Fd;
        const int j = (int)(nl - s * Fd);
        for (int d = fl - 1; d < a.Kmax; ++d) {
            PROF_INIT(0);
            const double* rin = d == 0 ? a.signals + s * (size_t)N : a.ws[d & 1] + (s * Fd + j) * (size_t)a.NS;
            double rnorm2, first[4];
            const float delta = node_prologue(a, sl, r64, r32, Npad, R32, rin, s, j, rnorm2, first, d);
            f32x4 acc[MT][NT];
            corr_tile_gemm(a.bank32t, sl->masks, (L + SLAB - 1) / SLAB, r32 + 64 * wave + lr + lq, slab, acc);
            select_children<true>(a, sl, r64, simrow, s, j, 1, delta, acc, d, d + 1 < a.Kmax ? a.ws[(d + 1) & 1] : nullptr);
            // the next level reads what this one wrote (residual row, node tables): make the stores visible to
            // this workgroup's own later loads (release + acquire at device scope: L1 is not coherent)
            __threadfence();
            __syncthreads();
        }
    }
}

// ---- the chain kernel: single-child levels, wave-private product ------------------------------------------------
// mmp_level_kernel keeps the whole (256 x 144) correlation matrix in registers, 144 VGPRs per lane, which allows
// two nodes per CU -- and a node is a chain of short dependent steps, so the CU idles.  A node does not need the
// matrix at once: only its maximum and the few entries within the band of it.  Here the fp16 product is cut into
// TASKS of (128 positions x 16 bank columns), each owned by ONE wave from its first slab to its last: 32 accumulator
// VGPRs, no barrier and no LDS traffic for the bank at all --
//   B fragments   the bank's fp16 blocks are stored fragment-major (one 16-byte piece per lane), so a wave's B fragment
//                 of (slab, column tile) is ONE global_load_dwordx4 straight from L2 into registers, requested BD = 5
//                 slabs ahead (a ring of four registers, counted vmcnt waits);
//   A fragments   the Hankel windows of the residual, from the eight shifted fp16 copies in LDS: a task's slab uses
//                 window fragments u = 2 s .. 2 s + 7 (row tile mi, slab s -> u = mi + 2 s), EIGHT registers in rotation:
//                 the two fragments a slab uses first (and last: u = 2 s, 2 s + 1) are overwritten, right behind the two
//                 MFMAs that read them, by the two the NEXT slab uses last (u = 2 s + 8, 2 s + 9) -- two 16-byte LDS
//                 reads per slab with twelve MFMAs to land under;
//   per slab      8 MFMAs (v_mfma_f32_16x16x32_f16) on one B fragment.
// A column tile is multiplied only over ITS live slab range (the bank's columns are sorted by support; outside the
// range the tile's blocks are zero padding): 101 of 144 (slab, tile) blocks on the canonical bank, where the
// three-group loop this replaces multiplied 111.  The 18 tasks of a node are dealt to the four waves longest first.
// A finished task is complete in registers, so its candidates are taken at once against the node's RUNNING maximum (the
// tasks finished before, by an LDS atomicMax, and this task's own maximum: a lower bound of the final maximum, so the
// entries kept are a superset), value beside index; the final band filters them behind the product's one barrier.
// The next task's first B fragments and its window are requested BEFORE the finished task's epilogue.
// (scripts/probe/prod_probe.hip times this loop against the three-group loop: 0.74 against 0.86 ms per 40 960 node
// products, matrix pipe 98 % full at four workgroups per CU, where the chip holds 1.45 GHz.)
// A node the fp16 filter cannot settle is flagged (a.flags) and left to mmp_level_followup_kernel (fp32), launched behind.
#define NG 3                           // (column tiles per group of the large-dictionary kernels below)
#define BD 4                           // B fragments in flight per wave (= the unroll of the slab loop: 8 window registers / 2)
#define RAW_CAP 512
#define TASK_SLOTS 8                   // descriptors per wave: slots 0 .. 4 tasks (18 tasks dealt to 4 waves), slot 7 their number
struct ChainLds {
    unsigned desc[WG / 64][TASK_SLOTS];   // task = column tile | row half << 4 | first live slab << 5 | one past the last << 10 (16 bits)
    unsigned short range[NT + 1];      // per column tile: first live slab | one past the last << 8
    int zero_atoms;                    // an all-zero column tile holds a real atom (no task computes it)
    unsigned gkey;                     // running maximum of the node's filter product (fkey)
    int raw_cnt;
    int raw_idx[RAW_CAP];
    float raw_val[RAW_CAP];
    double Linv[KMAX];                 // 1 / diagonal of the Cholesky factor (0 for a dead column): chain_tail_fast's solve
};

// float <-> unsigned key with the same order (for LDS atomicMax on maxima)
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// v_max3_f32 as it stands (fmaxf through the compiler comes with a canonicalising v_max x, x per operand)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The compiler counts the wait states between an MFMA and the first instruction that reads its result, but it does not
// look inside asm statements: a vmax3 scheduled right behind the MFMA that feeds it reads the register before the
// matrix core has written it (seen in screen_lr_kernel: bounds that changed from run to run).  Accumulators that go to
// vmax3 pass through here first: 16 wait states (a 16x16x32 product needs 11), tied to the registers so that neither the
// products nor the vmax3 chain can move across.
__device__ __forceinline__ void mfma_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6,
                                            f32x4& a7) {
    asm volatile("s_nop 15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
}
__device__ __forceinline__ void mfma_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6,
                                            f32x4& a7, f32x4& a8) {
    asm volatile("s_nop 15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8));
}

// The node's tasks, built by 9 + 18 threads: (1) thread c < 9: live slab range of column tile c from the bank's block
// masks (global: no wait for the LDS copy); (2) behind a barrier, thread t < 18 = (tile, row half): the tile's rank by
// falling length, task number 2 rank + half, dealt to the four waves in snake order (0 1 2 3 3 2 1 0 0 1 ...): the
// two halves of a tile start together on two waves (the second reader of a B block finds it in L1) and the waves'
// slab totals differ by a few per cent (54 / 54 / 50 / 50 on the canonical bank).  All-padding tiles get no task.
__device__ __forceinline__ void chain_tile_range(const int32_t* __restrict__ meta, int nslab, ChainLds* cl, int c) {
    int lo = nslab, hi = 0;
    for (int q = 0; q < nslab; ++q)
        if (((uint32_t)meta[CSC_TILE_COLS + q] >> c) & 1u) { lo = min(lo, q); hi = q + 1; }
    cl->range[c] = (unsigned short)(min(lo, hi) | (hi << 8));
    if (c < WG / 64) cl->desc[c][TASK_SLOTS - 1] = 0u;                  // the waves' task counts, raised below
    if (c == 0) cl->zero_atoms = 0;
}
__device__ __forceinline__ void chain_deal_task(ChainLds* cl, const int* perm, int t, int P) {
    const int c = t >> 1, rh = t & 1;
    const int pk = cl->range[c], lo = pk & 0xff, hi = pk >> 8, len = hi - lo;
    if (len <= 0) {
        // an all-zero column tile has no task; should it hold a real atom (an all-zero atom: its correlations are
        // exact zeros and may win when everything else is negative) the chain is left to the fp32 kernels
        if (rh == 0) {
            bool real = false;
            for (int q = 0; q < 16; ++q) real |= perm[16 * c + q] >= 0;
            if (real) cl->zero_atoms = 1;
        }
        return;
    }
    if (128 * rh >= P) return;
    int rank = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int pq = cl->range[q], lq = (pq >> 8) - (pq & 0xff);
        rank += (lq > len || (lq == len && q < c)) ? 1 : 0;
    }
    const int n = (P > 128 ? 2 * rank + rh : rank);                   // task number, longest first (no gaps: a wave
                                                                      // runs slots 0 .. count - 1)
    const int w = (n & 4) ? 3 - (n & 3) : (n & 3);
    cl->desc[w][n >> 2] = (unsigned)c | ((unsigned)rh << 4) | ((unsigned)lo << 5) | ((unsigned)hi << 10);
    atomicMax(&cl->desc[w][TASK_SLOTS - 1], (unsigned)(n >> 2) + 1u);
}

// The wave's tasks of one node.  FINAL: candidates by the given threshold `tau_fix` (the rare second pass behind a
// raw-list overflow) instead of the running maximum.  DUMP (test aid, csc_filter_product): the finished tasks'
// products go to dump[p * K + atom] as they are.
// No vector-memory operation of this wave may be in flight on entry (the loop's vmcnt counts assume its own loads).
template <bool FINAL, bool DUMP>
__device__ __forceinline__ void chain_product16(const _Float16* __restrict__ bank16, const int* perm, ChainLds* cl,
                                                const _Float16* r16, unsigned desc01, unsigned desc23, unsigned desc4,
                                                int nt, int nslab, int K, int P, float band, float tau_fix,
                                                float* __restrict__ dump) {
    // opaque copies: the descriptors and the lane index are the same on every level of a chain, and the compiler
    // otherwise hoists the first task's addresses out of the level loop and spills them to scratch
    asm volatile("" : "+s"(desc01), "+s"(desc23), "+s"(desc4), "+s"(nt));
    // (uniform values the epilogues use, pinned to scalar registers: as vector registers they were spilled, and every
    // epilogue's re-load waited for the next task's B fragments with it)
    band = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, band)));
    tau_fix = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tau_fix)));
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lq = lane >> 4;
    const _Float16* pa0 = r16 + (lr & 7) * R16_CS + (lr & 8) + 8 * lq;       // + 128 row half + 16 u
    // B addresses: the (uniform) base of the fp16 bank + a 32-bit byte offset per lane
    const char* bbase = reinterpret_cast<const char*>(bank16);
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned tile_off = 0u;
    f16x8 W[8], Bf[BD];
    f32x4 acc[8];
    int cur_c = 0, cur_rh = 0, s0 = 0, s1 = 1;
    const _Float16* pa = pa0;
    auto begin_task = [&](int it) __attribute__((always_inline)) {
        // the wave's descriptors ride in scalar registers (16 bits each; a VGPR copy was spilled and re-loaded here,
        // and the wait for that re-load also waited for the B fragments just requested)
        const unsigned dw = it < 2 ? desc01 : (it < 4 ? desc23 : desc4);
        const unsigned d = (it & 1) && it < 4 ? dw >> 16 : dw & 0xffffu;
        cur_c = (int)(d & 15u); cur_rh = (int)((d >> 4) & 1u); s0 = (int)((d >> 5) & 31u); s1 = (int)((d >> 10) & 31u);
        // (whatever a descriptor holds, the addresses below stay inside the bank and the window copies)
        cur_c = min(cur_c, NT - 1); s1 = max(1, min(s1, min(nslab, 16))); s0 = min(s0, s1 - 1);
        pa = pa0 + 128 * cur_rh;
        tile_off = (unsigned)cur_c * (BLKF * 2u);
#pragma unroll
        for (int i = 0; i < 8; ++i) W[i] = *reinterpret_cast<const f16x8*>(pa + 16 * (2 * s0 + i));
        // (a task shorter than BD slabs re-loads its last one: the load count stays fixed)
#pragma unroll
        for (int q = 0; q < BD; ++q)
            Bf[q] = *reinterpret_cast<const f16x8*>(bbase + (tile_off + (unsigned)min(s0 + q, s1 - 1) * (SLABF * 2u)) + lane16);
    };
    // LDS byte offset of the running maximum (for the hand-written ds_max_u32 below)
    const unsigned key_off = (unsigned)(size_t)(lptr_t)&cl->gkey;
    auto epilogue = [&](int c, int rh) __attribute__((always_inline)) {
        // The finished task's 128 x 16 correlations: entries outside the matrix out, maximum, running maximum, candidates.
        // Kept short on purpose: 18 of these per node share the SIMD's vector issue with the other waves' MFMAs.
        const int atom = perm[16 * c + lr];
        const int prow = 128 * rh + 4 * lq;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (128 * rh + 16 * mi + 15 >= P) {                          // (uniform; the last row tile only)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (prow + 16 * mi + rg >= P) acc[mi][rg] = -INFINITY;
            }
        if (DUMP) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int p = prow + 16 * mi + rg;
                    if (p < P && atom >= 0) dump[(size_t)p * K + atom] = acc[mi][rg];
                }
        }
        // the lane's maximum: two chains of v_max3_f32 (NaN operands are skipped, like fmaxf)
        mfma_settle(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
        float m0 = vmax3(acc[0][0], acc[0][1], acc[0][2]), m1 = vmax3(acc[0][3], acc[1][0], acc[1][1]);
        m0 = vmax3(m0, acc[1][2], acc[1][3]); m1 = vmax3(m1, acc[2][0], acc[2][1]);
        m0 = vmax3(m0, acc[2][2], acc[2][3]); m1 = vmax3(m1, acc[3][0], acc[3][1]);
        m0 = vmax3(m0, acc[3][2], acc[3][3]); m1 = vmax3(m1, acc[4][0], acc[4][1]);
        m0 = vmax3(m0, acc[4][2], acc[4][3]); m1 = vmax3(m1, acc[5][0], acc[5][1]);
        m0 = vmax3(m0, acc[5][2], acc[5][3]); m1 = vmax3(m1, acc[6][0], acc[6][1]);
        m0 = vmax3(m0, acc[6][2], acc[6][3]); m1 = vmax3(m1, acc[7][0], acc[7][1]);
        float lmax = vmax3(m0, m1, vmax3(acc[7][2], acc[7][3], -INFINITY));
        if (atom < 0) lmax = -INFINITY;
        float thr = tau_fix;
        if (!FINAL) {
            // Every lane raises the node's running maximum by its own (one ds_max_u32: the LDS serialises the 64 lanes,
            // no wave reduction on the vector pipe) and reads it back: the tasks finished before and all of this one.
            // (A NaN product gives the key of a NaN, above +inf: nothing is a candidate.)
            // (hand-written, wait included: a volatile read through the generic pointer became a FLAT load that waited
            // for the next task's B fragments, vmcnt(0), in every epilogue)
            unsigned cur;
            asm volatile("ds_max_u32 %1, %2\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(cur) : "v"(key_off), "v"(fkey(lmax)) : "memory");
            thr = fkey_inv(cur) - band;
#if defined(CSC_CAND_ORDER) && CSC_CAND_ORDER == 1
            {   // loosest order: nothing of the other tasks has been published yet
                const float bm = wave_max(lmax);
                thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bm == bm ? bm : INFINITY), 63)) - band;
            }
#endif
        }
        // candidates: one compare per entry, and a SCALAR branch over the (rare) entry some lane keeps -- a per-lane bit
        // mask and a select chain per kept entry cost five times the vector instructions
        if (lmax >= thr) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const bool hit = acc[mi][rg] >= thr;
                    if (__builtin_amdgcn_ballot_w64(hit)) {
                        if (hit) {
                            const int at = atomicAdd(&cl->raw_cnt, 1);
                            if (at < RAW_CAP) {
                                cl->raw_idx[at] = (prow + 16 * mi + rg) * K + atom;
                                cl->raw_val[at] = acc[mi][rg];
                            }
                        }
                    }
                }
        }
    };
    nt = min(nt, 5);
    if (nt <= 0) return;
#ifdef CSC_PHASE_PROF
    // (two stamps per product only: a stamp is an s_memtime + a full LDS-counter wait, three per task distorted the loop)
    const unsigned long long pt_start = __builtin_readcyclecounter();
    unsigned long long pt_steps = 0;
#endif
    begin_task(0);
    for (int it = 0; it < nt; ++it) {
#ifdef CSC_PHASE_PROF
        pt_steps += (unsigned long long)(s1 - s0);
#endif
        int s = s0;
        const int sl = s1 - 1;
        auto step = [&](auto phc, auto firstc) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value;
            constexpr bool FIRST = decltype(firstc)::value;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(2 * PH + mi) % 8], Bf[PH], FIRST ? zero : acc[mi], 0, 0, 0);
            // the next slab's last two fragments, into the registers those two MFMAs have just read
            W[(2 * PH) % 8] = *reinterpret_cast<const f16x8*>(pa + 16 * (2 * s + 8));
            W[(2 * PH + 1) % 8] = *reinterpret_cast<const f16x8*>(pa + 16 * (2 * s + 9));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 2; mi < 8; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(2 * PH + mi) % 8], Bf[PH], FIRST ? zero : acc[mi], 0, 0, 0);
            // The fragment BD slabs ahead.  Past the task's end the SAME number of loads goes out (the vmcnt counts are
            // static): the task's last fragment again, which this wave has just brought into the vector L1.
            {
                // (uniform base in scalar registers + the lane's fixed 32-bit offset: no vector ALU work for the address
                // and no address temporaries -- the allocator had put those into ring registers with loads in flight)
                const char* sb = bbase + (tile_off + (unsigned)min(s + BD, sl) * (SLABF * 2u));
                Bf[PH] = *reinterpret_cast<const f16x8*>(sb + lane16);
            }
            ++s;
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{}, std::true_type{});
        // whole rounds of four slabs in a loop with ONE way round (a `break` inside the round lets the compiler merge
        // the exits' load states into the loop header: a full vmcnt(0) drain every round), then the rest
        const int nrest = s1 - s;
        const int rounds = nrest / BD, rem = nrest - rounds * BD;
        for (int q = 0; q < rounds; ++q) {
            step(std::integral_constant<int, 1>{}, std::false_type{});
            step(std::integral_constant<int, 2>{}, std::false_type{});
            step(std::integral_constant<int, 3>{}, std::false_type{});
            step(std::integral_constant<int, 0>{}, std::false_type{});
        }
        if (rem >= 1) {
            step(std::integral_constant<int, 1>{}, std::false_type{});
            if (rem >= 2) {
                step(std::integral_constant<int, 2>{}, std::false_type{});
                if (rem >= 3) step(std::integral_constant<int, 3>{}, std::false_type{});
            }
        }
        const int done_c = cur_c, done_rh = cur_rh;
        if (it + 1 < nt) begin_task(it + 1);
        epilogue(done_c, done_rh);
    }
#ifdef CSC_PHASE_PROF
    if (threadIdx.x == 0) {                                           // wave 0's own clocks: slots 28, 31
        atomicAdd(&g_prof[28][blockIdx.x % PROF_BUCKETS], __builtin_readcyclecounter() - pt_start);
        atomicAdd(&g_prof[31][blockIdx.x % PROF_BUCKETS], pt_steps);
    }
#endif
}

// build_r16 for the chain kernel, from the LDS copy of the residual: thread t <= R16_CS / 4 converts elements
// 4t .. 4t + 6 once and writes, for each shift c = 4a + b, the four halves h[4t + b .. 4t + b + 3] as ONE 8-byte store
// to copy c at index 4 (t - a)  (copy_c[i] = h[i + c]): 8 stores per thread where the element-wise version has 32.
// (r64 is followed by the r16 copies and the node's structs in LDS, so the reads past the residual's end stay inside
// the workgroup's allocation; what they return is replaced by zero.)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void build_r16_quads(const double* r64, int N, float inv, _Float16* r16) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    if (t <= R16_CS / 4) {
        _Float16 h[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double v = r64[4 * t + u];
            h[u] = (4 * t + u < N) ? (_Float16)((float)v * inv) : (_Float16)0.f;
        }
#pragma unroll
        for (int c = 0; c < R16_COPIES; ++c) {
            const int a = c >> 2, b = c & 3, m = t - a;
            if (m >= 0 && m < R16_CS / 4) {
                f16x4 q = {h[b], h[b + 1], h[b + 2], h[b + 3]};
                *reinterpret_cast<f16x4*>(r16 + c * R16_CS + 4 * m) = q;
            }
        }
    }
}

// The usual end of a chain level -- at most 8 entries in the fp16 band, at most 4 ancestors -- in five barriers
// instead of nine (rescore_and_choose + child_solve_and_write remain for everything else, and give the same
// selections; the solve below multiplies by stored reciprocals where theirs divides, so the last bits of coefficients
// and MSE may differ between the two, which is why the choice between them depends on deterministic quantities only):
//   (A) every wave scans the raw list against the final band by itself (same list, same order in all four), takes
//       candidates 2w, 2w + 1 and re-scores them in float64; barrier; every THREAD takes the choice from the <= 8 scores;
//   (B) the new Gram column (one dot per wave) while every thread's operands of the residual update -- its elements of
//       the residual and of the <= 5 placed atoms -- are already on their way from L2; barrier;
//   (C) thread 0 extends the factor and solves, operands loaded up front, registers only; barrier;
//   (D) the child's residual from the prefetched operands, its sum of squares; barrier.
// Returns false when the float64 scores are all NaN (the caller takes the general path; the choice is in sl->...).
__device__ __forceinline__ bool chain_tail_fast(const LevelArgs& a, NodeLds* sl, ChainLds* cl, double* r64, size_t cid,
                                                int d, int ncand, int c0, int c1, int wave_s) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int K = a.K, L = a.L, N = a.N;
    // ---- (A)
    if (2 * wave_s < ncand) {
        const bool two = 2 * wave_s + 1 < ncand;
        const int i0 = c0, i1 = two ? c1 : c0;
        const int p0 = i0 / K, k0 = i0 - p0 * K, p1 = i1 / K, k1 = i1 - p1 * K;
        double v0, v1;
        dot_pair(a.bank64 + (size_t)k0 * L, r64 + p0, a.bank64 + (size_t)k1 * L, r64 + p1, L, lane, v0, v1);
        v0 = wave_sum(v0);
        v1 = wave_sum(v1);
        if (lane == 63) {
            sl->cand_val[2 * wave_s] = v0;
            sl->cand_idx[2 * wave_s] = i0;
            if (two) { sl->cand_val[2 * wave_s + 1] = v1; sl->cand_idx[2 * wave_s + 1] = i1; }
        }
    }
    PROF_MARK(2);
    __syncthreads();
    double bestv = sl->cand_val[0];
    int chosen = sl->cand_idx[0];
    for (int c = 1; c < ncand; ++c) {
        const double v = sl->cand_val[c];
        const int i = sl->cand_idx[c];
        if (v > bestv || (v == bestv && i < chosen)) { bestv = v; chosen = i; }
    }
    const bool ok = bestv == bestv && (unsigned)chosen < (unsigned)(a.P * K);
    if ((unsigned)chosen >= (unsigned)(a.P * K)) chosen = 0;
    const int p_sel = chosen / K, k_sel = chosen - p_sel * K;
    if (tid == 0) {
        if (a.stats) atomicAdd(&a.stats[2], ncand);
        sl->chosen = chosen;
        sl->sup_pos[d] = p_sel;
        sl->sup_atom[d] = k_sel;
        sl->beta = bestv;
        sl->beta_ok = ok ? 1 : 0;
    }
    if (!ok) { __syncthreads(); return false; }
    PROF_MARK(5);
    // ---- (B)
    const double* an = a.bank64 + (size_t)k_sel * L;
    const double gdd = a.anorm2[k_sel];
    // (elements tid, tid + 256, tid + 512 of the window; a fourth, N > 768, is not prefetched: the registers it would
    // hold across the solve were spilled, 300 MB of scratch writes per launch)
    double av[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int ii = min(i, d);
        const int pi = ii == d ? p_sel : sl->sup_pos[ii];
        const double* ai = ii == d ? an : a.bank64 + (size_t)sl->sup_atom[ii] * L;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int o = tid + WG * u - pi;
            av[i][u] = ((unsigned)o < (unsigned)L && i <= d) ? ai[o] : 0.0;
        }
    }
    for (int q = wave_s; q < d; q += WG / 64) {
        const double* aq = a.bank64 + (size_t)sl->sup_atom[q] * L;
        const int dp = p_sel - sl->sup_pos[q];
        double w = dot_range<64>(an, aq + dp, max(0, -dp), min(L, L - dp), lane);
        w = wave_sum(w);
        if (lane == 63) sl->g[q] = w;
    }
    PROF_MARK(4);
    __syncthreads();
    PROF_MARK(6);
    // ---- (C)
    if (tid == 0) {
        double l[TRI(4)], li[4], g[4], cp[4];
#pragma unroll
        for (int i = 0; i < TRI(4); ++i) l[i] = i < TRI(d) ? sl->Lf[i] : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            li[i] = i < d ? cl->Linv[i] : 0.0;
            g[i] = i < d ? sl->g[i] : 0.0;
            cp[i] = i < d ? sl->coefpar[i] : 0.0;
        }
        double y[4], x[5], yy = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double w = g[i];
#pragma unroll
            for (int m = 0; m < i; ++m) w = fma(-l[TRI(i) + m], y[m], w);   // (explicit: the same roundings as the general solve)
            w = (li[i] != 0.0) ? w * li[i] : 0.0;
            y[i] = w;
            yy = fma(w, w, yy);
        }
        const double sdd = gdd - yy;
        const bool bad = !(sdd > 1e-13 * gdd);
        double ydd = 0.0, xd = 0.0, lid = 0.0;
        if (bad) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = 0.0;
        } else {
            ydd = sqrt(sdd);
            xd = bestv / sdd;
            lid = 1.0 / ydd;
        }
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            double w = -y[r] * xd;
#pragma unroll
            for (int m = r + 1; m < 4; ++m) w = fma(-l[TRI(m) + r], x[m], w);
            x[r] = (li[r] != 0.0) ? w * li[r] : 0.0;
        }
        // (rows >= d of l, li, y are zero, so x[r] = 0 there and the sums above skip them)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < d) {
                sl->x[i] = x[i];
                sl->Lf[TRI(d) + i] = y[i];
                sl->coefpar[i] = x[i] + cp[i];
            }
        sl->x[d] = xd;
        sl->Lf[TRI(d) + d] = ydd;
        sl->coefpar[d] = xd;
        cl->Linv[d] = lid;
        if (bad && a.stats) atomicAdd(&a.stats[1], 1);
    }
    __syncthreads();
    PROF_MARK(7);
    // ---- (D)
    double cs = 0.0, v[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int n = tid + WG * u;
        v[u] = n < N ? r64[n] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double xi = i <= d ? sl->x[min(i, d)] : 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u) v[u] = fma(-xi, av[i][u], v[u]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int n = tid + WG * u;
        if (n < N) {
            r64[n] = v[u];
            cs = fma(v[u], v[u], cs);
        }
    }
    if (N > 3 * WG) {                                                 // (uniform) the window's fourth quarter, loaded here
        const int n = tid + 3 * WG;
        if (n < N) {
            double w = r64[n];
            for (int i = 0; i <= d; ++i) {
                const int o = n - (i == d ? p_sel : sl->sup_pos[i]);
                const double* ai = i == d ? an : a.bank64 + (size_t)sl->sup_atom[i] * L;
                w = fma(-sl->x[i], ((unsigned)o < (unsigned)L) ? ai[o] : 0.0, w);
            }
            r64[n] = w;
            cs = fma(w, w, cs);
        }
    }
    cs = wave_sum(cs);
    if (lane == 63) sl->wsum[wave_s] = cs;
    if (tid < a.Kmax && a.node_coef) a.node_coef[cid * a.Kmax + tid] = tid <= d ? sl->coefpar[tid] : 0.0;
    PROF_MARK(10);
    __syncthreads();
    if (tid == 0) {
        a.node_pos[cid] = p_sel;
        a.node_atom[cid] = k_sel;
        a.node_mse[cid] = (sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3]) / (double)N;
    }
    PROF_MARK(8);
    return true;
}

__device__ __forceinline__ void lean_chain(const LevelArgs& a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, K = a.K, L = a.L, d0 = a.depth;
    const int nslab = (L + SLAB - 1) / SLAB;

    // ---- LDS carve-up: r64 [Npad] | r16 [R16_COPIES][R16_CS] | NodeLds | ChainLds
    const int Npad = (N + 3) & ~3;
    double* r64 = lds_raw;
    _Float16* r16 = reinterpret_cast<_Float16*>(r64 + Npad);
    NodeLds* sl = reinterpret_cast<NodeLds*>(r16 + R16_COPIES * R16_CS);
    ChainLds* cl = reinterpret_cast<ChainLds*>(sl + 1);

    const int Fd = a.ts.F[d0];                                        // = F of every level below (one child per node)
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    if (tid >= 192 && tid < 192 + MAX_SLABS) sl->masks[tid - 192] = (uint32_t)a.meta[CSC_TILE_COLS + tid - 192];
    if (tid < CSC_TILE_COLS) sl->perm[tid] = a.meta[tid];
    if (tid >= 160 && tid < 160 + NT) chain_tile_range(a.meta, nslab, cl, tid - 160);

    PROF_INIT(16);
    const double* rin = (d0 == 0 || a.from_root) ? a.signals + s * (size_t)N : a.ws[d0 & 1] + (s * Fd + j) * (size_t)a.NS;
    double rnorm2, first[4];
    node_prologue(a, sl, r64, nullptr, Npad, 0, rin, s, j, rnorm2, first, a.from_root ? 0 : d0);
    if (tid >= 64 && tid < 64 + 2 * NT) chain_deal_task(cl, sl->perm, tid - 64, a.P);   // (behind the prologue's barrier)
    if (a.from_root) {
        // d0 = 1 and the root level left only its pick: this node is the signal's child (pos, atom) with <atom, signal>
        // parked in its MSE slot.  Its least squares, residual, factor and MSE are child_solve_and_write's at depth 0,
        // bit for bit what the register-resident kernel writes; they stay in LDS.
        const size_t cid = s * a.NN + a.ts.off[1] + j;
        if (tid == 0) {
            sl->sup_pos[0] = a.node_pos[cid];
            sl->sup_atom[0] = a.node_atom[cid];
            sl->beta = a.node_mse[cid];
            sl->beta_ok = 1;
        }
        __syncthreads();
        child_solve_and_write(a, sl, r64, s, j, cid, 0, nullptr, true);
        __syncthreads();
        rnorm2 = sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3];
    }
    PROF_MARK(0);
    const float am2 = *a.amax2, am = sqrtf(am2);
    const float band = 2.f * ((9.765625e-04f + DELTA_REL + 1e-6f) * am + 2.9802322e-08f * sqrtf((float)L) * (1.f + am));
    if (tid == 0) a.flags[s * Fd + j] = 0;
    if (tid < (a.from_root ? 1 : d0)) {                               // reciprocal diagonal of the factor brought along
        const double dg = sl->Lf[TRI(tid) + tid];
        cl->Linv[tid] = dg != 0.0 ? 1.0 / dg : 0.0;
    }
    __syncthreads();                                                  // the task table
    // the wave's task descriptors, packed into scalar registers
    const unsigned desc01 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cl->desc[wave][0] | (cl->desc[wave][1] << 16)));
    const unsigned desc23 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cl->desc[wave][2] | (cl->desc[wave][3] << 16)));
    const unsigned desc4 = (unsigned)__builtin_amdgcn_readfirstlane((int)cl->desc[wave][4]);
    const int nt = __builtin_amdgcn_readfirstlane((int)cl->desc[wave][TASK_SLOTS - 1]);

    // The chain cannot go on in fp16 at level d: the node's state goes to its workspace row (level d0's is still
    // there) and the flag tells the follow-up kernel where to take over.  Workgroup-uniform.
    auto give_up = [&](int d) {
        int tq = threadIdx.x;
        asm volatile("" : "+v"(tq));                                  // keeps the address parts out of the loop above
        if (d > d0 || a.from_root) {                                  // (from the root: nobody wrote level d0's row)
            double* row = a.ws[d & 1] + (s * Fd + j) * (size_t)a.NS;
            for (int n = tq; n < N; n += WG) row[n] = r64[n];
            if (tq < TRI(d)) row[N + tq] = sl->Lf[tq];
            if (tq < d) row[N + TRI(a.Kmax) + tq] = sl->coefpar[tq];
        }
        if (tq == 0) { a.flags[s * Fd + j] = (uint8_t)(d + 1); if (a.stats) atomicAdd(&a.stats[3], 1); }
    };

    // ---- the chain: this node, its child, its grandchild ... down to the leaves; the residual, the Cholesky
    // factor and the coefficients stay in LDS from one level to the next, only the node tables go to HBM
    const bool zero_atoms = cl->zero_atoms != 0;                      // (uniform; read behind the task table's barrier)
    for (int d = d0; d < a.Kmax; ++d) {
        // NaN / Inf / extreme scales, an all-zero atom: fp32
        if (zero_atoms || !(am2 > 1e-6f && am2 < 1e4f && rnorm2 > 1e-30 && rnorm2 < 1e30)) { give_up(d); return; }
        build_r16_quads(r64, N, (float)(1.0 / sqrt(rnorm2)), r16);
        if (tid == 0) { sl->cand_cnt = 0; cl->raw_cnt = 0; cl->gkey = fkey(-INFINITY); }
        WAIT_VM(0);                                                   // (the product counts its own loads)
        __syncthreads();
#ifdef CSC_PHASE_PROF
        const unsigned trace_t0 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        // ---- ONE sweep; every task's candidates against the running maximum ------------------------------------------
        chain_product16<false, false>(a.bank16, sl->perm, cl, r16, desc01, desc23, desc4, nt, nslab, K, a.P, band, 0.f, nullptr);
        __syncthreads();
        PROF_MARK(1);
#ifdef CSC_PHASE_PROF
        if (tid == 0) {
            const unsigned at = atomicAdd(&g_trace_n, 1u);
            if (at < TRACE_CAP) {
                g_trace[at][0] = __builtin_amdgcn_s_getreg(63492);
                g_trace[at][1] = trace_t0;
                g_trace[at][2] = (unsigned)__builtin_amdgcn_s_memtime();
                g_trace[at][3] = blockIdx.x | ((__builtin_amdgcn_s_getreg(6164) & 15u) << 24) | ((unsigned)d << 28);
            }
        }
#endif
        int nraw = cl->raw_cnt;
        const float tau = fkey_inv(cl->gkey) - band;
#if defined(CSC_CAND_ORDER) && CSC_CAND_ORDER == 2
        const bool force_final = true;                                // tightest order: every task sees the final maximum
#else
        const bool force_final = false;
#endif
        const int nraw_first = nraw;
        if (nraw > RAW_CAP || force_final) {
            // (uniform, rare) the early thresholds let too many entries through: once more, with the final band
            __syncthreads();
            if (tid == 0) cl->raw_cnt = 0;
            __syncthreads();
            chain_product16<true, false>(a.bank16, sl->perm, cl, r16, desc01, desc23, desc4, nt, nslab, K, a.P, band, tau, nullptr);
            __syncthreads();
            nraw = cl->raw_cnt;
            if (nraw > RAW_CAP) { give_up(d); return; }               // a flat residual: fp32
        }
        // the usual case -- a handful of entries in the final band, a shallow support -- goes through chain_tail_fast.
        // Every wave scans the raw list by itself: the same list in the same order, so the four agree on the number of
        // candidates and on who re-scores which without a barrier.  (The choice of path rests on that NUMBER, which does
        // not depend on the order the tasks finished in; the list's order does.)
        const size_t cid = s * a.NN + a.ts.off[d + 1] + j;
        bool fast = false;
        if (a.mode == 0 && d <= 4) {
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);
            int ncand = 0, c0 = 0, c1 = 0;
            for (int base = 0; base < nraw; base += 64) {
                const int i = base + lane;
                const bool pass = i < nraw && cl->raw_val[i] >= tau;
                const int idx = i < nraw ? cl->raw_idx[i] : 0;
                unsigned long long m = __ballot(pass);
                while (m) {
                    const int pos = __builtin_ctzll(m);
                    m &= m - 1;
                    const int id = __builtin_amdgcn_readlane(idx, pos);
                    if (ncand == 2 * wave_s) c0 = id;
                    if (ncand == 2 * wave_s + 1) c1 = id;
                    ++ncand;
                }
            }
            if (ncand >= 1 && ncand <= 8) {
                PROF_MARK(3);
#ifdef CSC_CAND_ORDER
                const size_t node = ((size_t)s * Fd + j) * a.Kmax + d;
#endif
                fast = chain_tail_fast(a, sl, cl, r64, cid, d, ncand, c0, c1, wave_s);
#ifdef CSC_CAND_ORDER
                if (node < CAND_DUMP_NODES) {                         // (the list is in sl->cand_idx behind the tail's barriers)
                    if (tid == 0) { g_cand_dump[node][0] = force_final ? nraw : nraw_first; g_cand_dump[node][1] = ncand; }
                    if (tid < ncand) g_cand_dump[node][2 + tid] = sl->cand_idx[tid];
                }
#endif
                if (!fast) {                                          // all-NaN scores: the general solve on the choice made
                    child_solve_and_write(a, sl, r64, s, j, cid, d, nullptr, true);
                    __syncthreads();
                    if (tid == 0) { const double dg = sl->Lf[TRI(d) + d]; cl->Linv[d] = dg != 0.0 ? 1.0 / dg : 0.0; }
                }
                rnorm2 = sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3];
                continue;
            }
        }
        {
            int tq = threadIdx.x;
            asm volatile("" : "+v"(tq));
            for (int i = tq; i < nraw; i += WG)
                if (cl->raw_val[i] >= tau) {
                    const int slot = atomicAdd(&sl->cand_cnt, 1);
                    // candidates 128 .. 255 wait in the space of the float64 scores, which nobody has written yet
                    if (slot < CAND_CAP) sl->cand_idx[slot] = cl->raw_idx[i];
                    else if (slot < 2 * CAND_CAP) reinterpret_cast<int*>(sl->cand_val)[slot - CAND_CAP] = cl->raw_idx[i];
                }
        }
        __syncthreads();
        PROF_MARK(3);
        const int ncand_raw = sl->cand_cnt;
#ifdef CSC_CAND_ORDER
        {
            const size_t node = ((size_t)s * Fd + j) * a.Kmax + d;
            if (node < CAND_DUMP_NODES) {
                if (tid == 0) { g_cand_dump[node][0] = force_final ? nraw : nraw_first; g_cand_dump[node][1] = ncand_raw; }
                for (int i = tid; i < min(ncand_raw, 2 * CAND_CAP); i += WG)
                    g_cand_dump[node][2 + i] = i < CAND_CAP ? sl->cand_idx[i] : reinterpret_cast<const int*>(sl->cand_val)[i - CAND_CAP];
            }
        }
        __syncthreads();
#endif
        if (ncand_raw > 2 * CAND_CAP) { give_up(d); return; }         // too many entries in the fp16 band: fp32
        if (ncand_raw <= CAND_CAP) {
            rescore_and_choose(a, sl, r64, ncand_raw, false, d);
        } else {
            // a flat residual deep in a tree: 129 .. 256 entries in the band.  The re-score takes them in two halves (the
            // second half's indices ride in registers while the first half's scores use their space); the choice is the
            // float64 maximum over both, lowest index first
            int tq = threadIdx.x;
            asm volatile("" : "+v"(tq));
            const int extra = tq < ncand_raw - CAND_CAP ? reinterpret_cast<const int*>(sl->cand_val)[tq] : 0;
            __syncthreads();
            rescore_and_choose(a, sl, r64, CAND_CAP, false, d);
            double v0 = 0.0;
            int i0 = 0, ok0 = 0;
            if (tq == 0) { v0 = sl->beta; i0 = sl->chosen; ok0 = sl->beta_ok; }
            if (tq < ncand_raw - CAND_CAP) sl->cand_idx[tq] = extra;
            __syncthreads();
            rescore_and_choose(a, sl, r64, ncand_raw - CAND_CAP, false, d);
            if (tq == 0 && ok0 && (!sl->beta_ok || v0 > sl->beta || (v0 == sl->beta && i0 < sl->chosen))) {
                sl->chosen = i0;
                sl->sup_pos[d] = i0 / K;
                sl->sup_atom[d] = i0 - (i0 / K) * K;
                sl->beta = v0;
                sl->beta_ok = 1;
            }
            __syncthreads();
        }
        PROF_MARK(5);
        child_solve_and_write(a, sl, r64, s, j, cid, d, nullptr, true);
        __syncthreads();
        if (tid == 0) { const double dg = sl->Lf[TRI(d) + d]; cl->Linv[d] = dg != 0.0 ? 1.0 / dg : 0.0; }
        rnorm2 = sl->wsum[0] + sl->wsum[1] + sl->wsum[2] + sl->wsum[3];   // the child's ||residual||^2
    }
    PROF_MARK(9);
}

__global__ __launch_bounds__(WG, 5) void mmp_level_lean_kernel(const LevelArgs a) { lean_chain(a); }

// ------------------------------------------------------------------------------------------------
// test aid: the fp16 filter product itself (what the level kernels only ever hold in registers)
// ------------------------------------------------------------------------------------------------
// out[b][p][k] = the fp16 matrix-core product of signal b scaled to unit norm with atom k at shift p, in units of
// ||signal||, taken by the very loops the level kernels use: variant 0 = corr_tile_gemm16 (mmp_level_kernel),
// variant 1 = chain_product16, the wave-private tasks of the chain kernel (mmp_level_lean_kernel).  eps[b] = the rounding bound the
// kernels assume for an entry (half their candidate band); norm[b] = ||signal||.  tests/test_gpu_filter_bound.py
// checks |out - float64| <= eps on golden residuals and adversarial operands.
__global__ __launch_bounds__(WG, 2) void filter_product_kernel(const _Float16* __restrict__ bank16,
                                                               const int32_t* __restrict__ meta, const float* amax2,
                                                               int K, int L, const double* __restrict__ signals, int N,
                                                               int P, int variant, float* __restrict__ out,
                                                               float* __restrict__ eps, double* __restrict__ norm) {
    extern __shared__ double lds_raw[];
    __shared__ uint32_t masks[MAX_SLABS];
    __shared__ int perm[CSC_TILE_COLS];
    __shared__ double wsum[WG / 64];
    __shared__ ChainLds cls;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int nslab = (L + SLAB - 1) / SLAB;
    const int Npad = (N + 3) & ~3;
    _Float16* ring = reinterpret_cast<_Float16*>(lds_raw);            // [RING16][SLABF] halves
    double* r64 = lds_raw + (RING16 * SLABF) / 4;
    _Float16* r16 = reinterpret_cast<_Float16*>(r64 + Npad);
    const size_t b = blockIdx.x;
    if (tid < MAX_SLABS) masks[tid] = (uint32_t)meta[CSC_TILE_COLS + tid];
    if (tid < CSC_TILE_COLS) perm[tid] = meta[tid];
    double first[4], ss = 0.0;
    for (int i0 = 0; i0 < Npad; i0 += 4 * WG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tid + WG * u;
            const double v = i < N ? signals[b * N + i] : 0.0;
            if (i < Npad) r64[i] = v;
            ss = fma(v, v, ss);
            if (i0 == 0) first[u] = v;
        }
    }
    ss = wave_sum(ss);
    if (lane == 63) wsum[wave] = ss;
    __syncthreads();
    const double rnorm2 = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tid >= 160 && tid < 160 + NT) chain_tile_range(meta, nslab, &cls, tid - 160);
    if (tid == 0) { cls.raw_cnt = 0; cls.gkey = fkey(-INFINITY); }
    __syncthreads();
    if (tid >= 64 && tid < 64 + 2 * NT) chain_deal_task(&cls, perm, tid - 64, P);
    const float am = sqrtf(*amax2);
    if (tid == 0) {
        eps[b] = (9.765625e-04f + DELTA_REL + 1e-6f) * am + 2.9802322e-08f * sqrtf((float)L) * (1.f + am);
        norm[b] = sqrt(rnorm2);
    }
    build_r16(r64, N, (float)(1.0 / sqrt(rnorm2)), first, r16);
    __syncthreads();
    const int prow = 64 * wave + 4 * lq;
    if (variant == 0) {
        f32x4 acc[MT][NT];
        WAIT_VM(0);
        gemm16_prefetch(bank16, nslab, ring);
        corr_tile_gemm16(bank16, masks, nslab, r16, ring, acc);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int p = prow + 16 * mi + rg, k = perm[16 * ni + lr];
                    if (p < P && k >= 0) out[(b * P + p) * K + k] = acc[mi][ni][rg];
                }
    } else {
        // column tiles that are all padding belong to no task: their (valid) entries are exact zeros
        for (int c = 0; c < NT; ++c) {
            bool live = false;
            for (int q = 0; q < nslab; ++q) live |= ((masks[q] >> c) & 1u) != 0u;
            const int k = perm[16 * c + lr];
            if (!live && k >= 0)
                for (int p = tid >> 4; p < P; p += WG / 16) out[(b * P + p) * K + k] = 0.f;
        }
        const unsigned desc01 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cls.desc[wave][0] | (cls.desc[wave][1] << 16)));
        const unsigned desc23 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cls.desc[wave][2] | (cls.desc[wave][3] << 16)));
        const unsigned desc4 = (unsigned)__builtin_amdgcn_readfirstlane((int)cls.desc[wave][4]);
        const int nt = __builtin_amdgcn_readfirstlane((int)cls.desc[wave][TASK_SLOTS - 1]);
        WAIT_VM(0);
        chain_product16<true, true>(bank16, perm, &cls, r16, desc01, desc23, desc4, nt, nslab, K, P, 0.f, INFINITY, out + b * (size_t)P * K);
    }
}

// ------------------------------------------------------------------------------------------------
// the tiled level kernel: dictionaries with more than 144 atoms and/or more than 256 positions
// ------------------------------------------------------------------------------------------------
// Same node semantics as mmp_level_kernel, but the (P x K) correlation matrix no longer fits one
// workgroup's registers: it is produced unit by unit (unit = 256 positions x one 144-atom bank tile)
// and never stored.  Per sibling: sweep 1 keeps the fp32 maximum of every unit (after suppression by
// the earlier siblings, read from the similarity table), sweep 2 recomputes only the units whose
// maximum reaches the candidate band and collects the candidates; then the shared fp64 re-score /
// least-squares path runs.  For siblings after the first, the unsuppressed unit maxima of the first
// sweep are upper bounds, so units that cannot reach the running maximum are skipped.
#define MAX_UNITS 2048
struct TiledLds {
    float tmax0[MAX_UNITS];        // unsuppressed unit maxima (first sibling's sweep)
    float tcur[MAX_UNITS];         // unit maxima of the current sibling's sweep (-inf = skipped)
    int sib_pos[CSC_MAX_CONNECTIONS], sib_atom[CSC_MAX_CONNECTIONS];
};

__device__ __forceinline__ void load_tile_meta(const LevelArgs& a, NodeLds* sl, int tl) {
    const int tid = threadIdx.x;
    const int32_t* m = a.meta + (size_t)tl * META_INTS;
    if (tid >= 192 && tid < 192 + MAX_SLABS) sl->masks[tid - 192] = (uint32_t)m[CSC_TILE_COLS + tid - 192];
    if (tid < CSC_TILE_COLS) sl->perm[tid] = m[tid];
    __syncthreads();
}

// GEMM of one unit + masking of the entries outside the matrix + suppression by the first t siblings.
// Returns the lane's maximum over its unsuppressed entries; lane_supp tells whether it holds a
// suppressed (value 0) one.
__device__ __forceinline__ float unit_correlations(const LevelArgs& a, NodeLds* sl, const TiledLds* tl2,
                                                   const float* r32, float* slab, int rb, int tile, int t,
                                                   int (&permv)[NT], f32x4 (&acc)[MT][NT], bool& lane_supp,
                                                   int& rowbase_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int K = a.K, P = a.P, W = a.W;
    const int nslab = (a.L + SLAB - 1) / SLAB;
    load_tile_meta(a, sl, tile);
    const bool rem = (P - rb * CSC_TILE_ROWS) <= 16;                 // uniform: a one-tile remainder block
    const int rowbase = rb * CSC_TILE_ROWS + (rem ? 0 : 64 * wave);
    if (rem) {
        corr_rem_gemm(a.bank32t + (size_t)tile * nslab * SLABF, sl->masks, nslab,
                      r32 + rb * CSC_TILE_ROWS + lr + lq, slab, acc);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if ((ni & 3) != wave) acc[0][ni] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int mi = 1; mi < MT; ++mi) acc[mi][ni] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    } else {
        corr_tile_gemm(a.bank32t + (size_t)tile * nslab * SLABF, sl->masks, nslab,
                       r32 + rowbase + lr + lq, slab, acc);
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) permv[ni] = sl->perm[16 * ni + lr];
    const int prow = rowbase + 4 * lq;                               // + 16*mi + rg
    rowbase_out = rowbase;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
        if (rowbase + 16 * mi + 15 >= P) {                           // wave-uniform
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (prow + 16 * mi + rg >= P) acc[mi][ni][rg] = -INFINITY;
        }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
        if (__ballot(permv[ni] < 0)) {                               // wave-uniform
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (permv[ni] < 0) acc[mi][ni][rg] = -INFINITY;
        }
    lane_supp = false;
    for (int q = 0; q < t; ++q) {
        const int dpb = prow - tl2->sib_pos[q] + P - 1;              // bit index of (mi, rg) = (0, 0); >= 0
        const int w0 = dpb >> 5, b0 = dpb & 31;
        const uint32_t* srow = a.sim_bits + (size_t)tl2->sib_atom[q] * K * W;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const uint32_t* row = srow + (size_t)max(permv[ni], 0) * W;
            const uint32_t wd0 = row[min(w0, W - 1)], wd1 = row[min(w0 + 1, W - 1)], wd2 = row[min(w0 + 2, W - 1)];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int off = b0 + 16 * mi + rg;               // < 32 + 52
                    const uint32_t wd = off < 32 ? wd0 : (off < 64 ? wd1 : wd2);
                    const bool hit = ((wd >> (off & 31)) & 1u) && acc[mi][ni][rg] > -INFINITY;
                    acc[mi][ni][rg] = hit ? __uint_as_float(SUPP_BITS) : acc[mi][ni][rg];
                    lane_supp |= hit;
                }
        }
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) lmax = fmaxf(lmax, acc[mi][ni][rg]);
    return lmax;
}

__global__ __launch_bounds__(WG, 2) void mmp_level_tiled_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = a.N, K = a.K, Lp = a.Lp, P = a.P, d = a.depth;
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nrb = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const int nunits = ntile * nrb;

    const int Npad = (N + 3) & ~3;
    const int R32 = nrb * CSC_TILE_ROWS + Lp + 4;
    float* slab = reinterpret_cast<float*>(lds_raw);                  // [2][SLABF]
    double* r64 = lds_raw + SLABF;                                    // [Npad]
    float* r32 = reinterpret_cast<float*>(r64 + Npad);                // [R32]
    NodeLds* sl = reinterpret_cast<NodeLds*>(r32 + ((R32 + 3) & ~3));
    PROF_INIT(16);
    TiledLds* tl2 = reinterpret_cast<TiledLds*>(sl + 1);

    const int Fd = a.ts.F[d], Fc = a.ts.F[d + 1];
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    // launched behind mmp_level_tiled16_kernel: only the nodes its filter could not settle (workgroup-uniform)
    if (a.flags && !a.flags[s * Fd + j]) return;
    const int stride = a.ts.pw[d];
    int nch = 0;
    for (int r = 0; r < a.connections; ++r) {
        if ((long long)j + (long long)r * stride < Fc) ++nch; else break;
    }
    const double* rin = a.resid_in + (d == 0 ? s * (size_t)N : (s * Fd + j) * (size_t)a.NS);
    double rnorm2, first[4];
    const float delta = node_prologue(a, sl, r64, r32, Npad, R32, rin, s, j, rnorm2, first, d);

    f32x4 acc[MT][NT];
    int permv[NT];
    for (int t = 0; t < nch; ++t) {
        // ---- sweep 1: unit maxima ------------------------------------------------------------------
        float gmax = -INFINITY;                                       // workgroup-uniform
        for (int u = 0; u < nunits; ++u) {
            if (t > 0 && fmaxf(tl2->tmax0[u], 0.f) < gmax - delta) {   // cannot reach the band (uniform)
                if (tid == 0) tl2->tcur[u] = -INFINITY;
                continue;
            }
            bool lane_supp;
            int rowbase;
            const float lmax = unit_correlations(a, sl, tl2, r32, slab, u / ntile, u % ntile, t, permv, acc, lane_supp, rowbase);
            const float bv = wave_max(lane_supp ? fmaxf(lmax, 0.f) : lmax);
            if (lane == 63) sl->wmax[wave] = bv;
            __syncthreads();
            const float m = fmaxf(fmaxf(sl->wmax[0], sl->wmax[1]), fmaxf(sl->wmax[2], sl->wmax[3]));
            if (tid == 0) {
                tl2->tcur[u] = m;
                if (t == 0) tl2->tmax0[u] = m;
            }
            gmax = fmaxf(gmax, m);
            __syncthreads();                                          // wmax is rewritten by the next unit
        }
        // ---- sweep 2: candidates from the units that reach the band --------------------------------------
        const float tau = gmax - delta;
        const bool zero_band = (0.f >= tau);
        if (tid == 0) sl->cand_cnt = 0;
        __syncthreads();
        int first_supp = 0x7fffffff, bi = 0x7fffffff;
        for (int u = 0; u < nunits; ++u) {
            if (!(tl2->tcur[u] >= tau) && !zero_band) continue;        // uniform
            bool lane_supp;
            int rowbase;
            const float lmax = unit_correlations(a, sl, tl2, r32, slab, u / ntile, u % ntile, t, permv, acc, lane_supp, rowbase);
            const int rowK = (rowbase + 4 * lq) * K;
            if (lmax >= tau) {
                uint32_t mk[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni) {
                            const int e = (4 * mi + rg) * NT + ni;
                            mk[e >> 5] |= (acc[mi][ni][rg] >= tau) ? (1u << (e & 31)) : 0u;
                        }
#pragma unroll
                for (int w = 0; w < 5; ++w) {
                    uint32_t mm = mk[w];
                    while (mm) {
                        const int e = 32 * w + __builtin_ctz(mm);
                        mm &= mm - 1;
                        const int r = e / NT, ni = e - r * NT;
                        const int idx = rowK + (16 * (r >> 2) + (r & 3)) * K + sl->perm[16 * ni + lr];
                        const int slot = atomicAdd(&sl->cand_cnt, 1);
                        if (slot < CAND_CAP) sl->cand_idx[slot] = idx;
                    }
                }
                if (lmax == gmax) {                                   // overflow fallback: entries equal to the maximum
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                            for (int ni = 0; ni < NT; ++ni)
                                if (acc[mi][ni][rg] == gmax) bi = min(bi, rowK + (16 * mi + rg) * K + permv[ni]);
                }
            }
            if (zero_band) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni)
                            if (is_supp(acc[mi][ni][rg]))
                                first_supp = min(first_supp, rowK + (16 * mi + rg) * K + permv[ni]);
                if (gmax == 0.f) bi = min(bi, first_supp);            // the value 0 of suppressed entries is the maximum
            }
            __syncthreads();                                          // perm / masks are rewritten by the next unit
        }
        if (zero_band) {
            first_supp = wave_min(first_supp);
            if (lane == 0) sl->wsupp[wave] = first_supp;
        }
        // overflow fallback (never seen on real data): lowest index among the entries equal to the fp32 maximum,
        // as in mmp_level_kernel (suppressed entries count with their value 0)
        bi = wave_min(bi);
        if (lane == 0) sl->widx[wave] = bi;
        __syncthreads();
        const int ncand_raw = sl->cand_cnt;
        rescore_and_choose(a, sl, r64, ncand_raw, zero_band, d);
        const int jc = j + t * stride;
        const size_t cid = s * a.NN + a.ts.off[d + 1] + jc;
        if (tid == 0) { tl2->sib_pos[t] = sl->sup_pos[d]; tl2->sib_atom[t] = sl->sup_atom[d]; }
        child_solve_and_write(a, sl, r64, s, jc, cid, d, a.resid_out, false);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// the tiled level kernel on the fp16 matrix cores: a three-product filter
// ------------------------------------------------------------------------------------------------
// Same node, same unit-by-unit sweeps as mmp_level_tiled_kernel, but the correlations come from
// v_mfma_f32_16x16x32_f16 (16x the fp32 rate) and still carry ~fp32 accuracy: both operands are two-term fp16
// numbers -- residual r 2^6 / ||r|| = rh + rl, bank a 2^sa = ah + al, each to ~2^-22 relative (bank_pack16s_kernel) --
// and a product is three MFMAs, rh ah + rh al + rl ah (rl al ~ 2^-22 is dropped), accumulated in fp32.  An entry, in
// units of ||r||, is then off by at most
//     eps3 = (2^-13 + 2^-20) am + 2^-29 sqrt(L) (1 + am),     am = max_k ||a_k||:
// 2^-13 am for the fp32 accumulation of 3 L products (3 L 2^-24 <= 2^-13 up to L = 682; checked at launch), 3 x 2^-22 am
// for the two operand splits and the dropped product, the last term for operands that are subnormal in fp16 after
// scaling.  Every entry within 2 eps3 of the computed maximum is re-scored in float64, so the selection is the
// float64 arg-max -- what the single-term filter cannot offer a finely sampled dictionary (its 2^-10 band holds
// ~200 of 9570 atoms).  A unit here is (256 positions + a 16-row remainder tile) x 48 atoms (a column GROUP of a
// bank tile): 48 accumulator VGPRs; the remainder tile (positions 256..271 of the row block, e.g. rows 256..259 of
// P = 260) rides along in the same pass, its three column tiles dealt to waves 0..2 (3 MFMAs per slab and wave
// instead of a second pass over the bank).
// A node the filter cannot settle (more than CAND_CAP entries in the band, the value 0 of suppressed entries inside
// it, NaN / Inf / extreme scales) is flagged; mmp_level_tiled_kernel, launched behind on the same level with the flag
// array, redoes exactly the flagged nodes in fp32.
// halves per shifted copy of the residual: the whole window (row blocks + remainder tile + atom slabs + one slab of
// look-ahead), rounded up to 32 mod 128 so that the copy stride is 64 mod 256 bytes (conflict-free ds_read_b128, as R16_CS)
static __host__ __device__ inline int r16t_stride(int P, int Lp) {
    const int nrb_all = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const bool rem = nrb_all > 1 && (P - (nrb_all - 1) * CSC_TILE_ROWS) <= 16;
    const int need = (rem ? nrb_all - 1 : nrb_all) * CSC_TILE_ROWS + 16 + Lp + SLAB;
    return ((need - 32 + 127) / 128) * 128 + 32;
}
// The residual, scaled by sc, as fp16 (rh) or as a two-term fp16 number (rh + rl) in R16_COPIES shifted copies of CS halves:
// copy c holds element m + c at m.  Every thread converts its elements once and stores them to the eight places.
__device__ __forceinline__ void build_r16t_copies(const double* r, int N, double sc, _Float16* rh, _Float16* rl, int CS) {
    for (int e = threadIdx.x; e < CS + R16_COPIES - 1; e += WG) {
        const double x = e < N ? r[e] * sc : 0.0;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (double)h);
#pragma unroll
        for (int c = 0; c < R16_COPIES; ++c) {
            const int m = e - c;
            if (m >= 0 && m < CS) {
                rh[c * CS + m] = h;
                if (rl) rl[c * CS + m] = l;
            }
        }
    }
}
#define TRING 4                        // group-slabs in the ring: one being multiplied, one landed, two copies in flight
#define TG_BLK (2 * NG * BLKF)         // halves of one group-slab: 3 hi blocks, 3 lo blocks
#define EPS3_ACC 1.220703125e-04f      // 2^-13

#define CAND_CAP16 (2 * CAND_CAP)
#define MAX_UNITS16 1024               // (row block, bank tile, column group) units per node, and at most 2048 16-column
                                       // blocks in the bank: K up to 32 768 at P <= 272
struct Tiled16Lds {
    float tmax0[MAX_UNITS16];      // upper bound of each unit's three-product entries (screen maximum + eps1 + eps3)
    int sib_pos[CSC_MAX_CONNECTIONS], sib_atom[CSC_MAX_CONNECTIONS];
    unsigned short urange[MAX_UNITS16];   // slab range s0 | s1 << 8 of column group g of bank tile: [g + NG tile]
    int permbuf[2][CSC_TILE_COLS]; // atom of each column of the current / the next unit's bank tile (begin_unit16x3)
    short order[MAX_UNITS16];      // units by descending bound (the sweep order of every sibling)
    // candidates taken during the sweep: flattened index and three-product value.  Twice the capacity of the shared
    // re-score (a flat residual at the last level puts 130 - 200 entries of a finely sampled dictionary in the band);
    // the re-score then runs over the list in two halves
    int cand_i[CAND_CAP16];
    float cand_v[CAND_CAP16];
};

// group-slab s of group g -> one ring slot: 6 KiB = 3 hi blocks + 3 lo blocks; every wave issues two copies (one
// whole block, one half block), so "my copies of slab s have landed" is the same vmcnt on every wave
__device__ __forceinline__ void stage_group16x3(const _Float16* hi_slab, const _Float16* lo_slab, _Float16* lslot, int g,
                                                int wave, int lane) {
    const _Float16* src = wave < NG ? hi_slab + (NG * g + wave) * BLKF : lo_slab + NG * g * BLKF;    // pieces 0..3
    __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 8), (lptr_t)(lslot + wave * BLKF), 16, 0, 0);
    const int p = 4 + (wave >> 1), half = (wave & 1) * (BLKF / 2);                                  // pieces 4, 5 in halves
    if (lane < 32)
        __builtin_amdgcn_global_load_lds((gptr_t)(lo_slab + (NG * g + p - NG) * BLKF + half + lane * 8),
                                         (lptr_t)(lslot + p * BLKF + half), 16, 0, 0);
}

// What a unit needs before its product can start, issued as early as possible -- for the sweep's NEXT unit right
// behind the current unit's last slab, so that the L2 / MALL latency (the units of a node are a scattered handful of
// a 20 MB bank: 70 % L2 misses) runs under the current unit's epilogue: the column -> atom map of the unit's bank tile
// -> permbuf[buf] (LDS-DMA, 48 ints per wave on waves 0..2; issued BEFORE the ring copies, so the product's vmcnt
// waits cover it), and the group's first three slabs -> ring slots 0..2.  The ring must be idle (behind the barrier
// that ends gemm16x3_group).
__device__ __forceinline__ void begin_unit16x3(const LevelArgs& a, const _Float16* hi_all, const _Float16* lo_all,
                                               Tiled16Lds* tl, int u, int buf, int ntile, int nslab, _Float16* ring) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = u % NG, tile = (u / NG) % ntile;
    if (wave < 3 && lane < 48)
        __builtin_amdgcn_global_load_lds((gptr_t)(a.meta + (size_t)tile * META_INTS + wave * 48 + lane),
                                         (lptr_t)(&tl->permbuf[buf][wave * 48]), 4, 0, 0);
    const int pk = __builtin_amdgcn_readfirstlane((int)tl->urange[g + NG * tile]);
    const int s0 = pk & 0xff, s1 = pk >> 8;
    const _Float16* hi = hi_all + (size_t)tile * nslab * SLABF;
    const _Float16* lo = lo_all + (size_t)tile * nslab * SLABF;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (s0 + q < s1)
            stage_group16x3(hi + (size_t)(s0 + q) * SLABF, lo + (size_t)(s0 + q) * SLABF, ring + q * TG_BLK, g, wave, lane);
}

// acc[mi][nj]: positions rowbase + 64 wave + 16 mi + ..., column tile 3 g + nj, over slabs [s0, s1) of the tile at
// (hi, lo); REM: accr = the remainder row tile (positions rowbase + 256 + ...), column tile 3 g + wave (waves 0..2)
template <bool REM>
__device__ __forceinline__ void gemm16x3_group(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, int g,
                                               int s0, int s1, const _Float16* rh, const _Float16* rl, int R16T_CS,
                                               int rowbase, _Float16* ring, f32x4 (&acc)[MT][NG], f32x4& accr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nj = 0; nj < NG; ++nj) acc[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accr = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment (row tile mi, slab s) of this lane: elements row + 8 lq + 32 s ... + 7 of the window, row = rowbase +
    // 64 wave + 16 mi + lr, i.e. one aligned 16-byte read of shifted copy row mod 8
    const int aoff = (lr & 7) * R16T_CS + rowbase + 64 * wave + (lr & 8) + 8 * lq;
    const int roff = (lr & 7) * R16T_CS + rowbase + CSC_TILE_ROWS + (lr & 8) + 8 * lq;
    // the copies of the group's first three slabs (slots 0..2) are in flight already: begin_unit16x3 issued them, as
    // this wave's most recent ring copies (vector loads issued since only make the waits below stricter)
    f16x8 ah[2][MT], al[2][MT], arh[2], arl[2], bh[2][NG], bl[2][NG];
#ifdef CSC_PHASE_PROF
    unsigned t3acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t3last = T3_CLK();
#endif
    auto load_a = [&](auto setc, int s) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            ah[set][mi] = *reinterpret_cast<const f16x8*>(rh + aoff + 16 * mi + 32 * s);
            al[set][mi] = *reinterpret_cast<const f16x8*>(rl + aoff + 16 * mi + 32 * s);
        }
        if (REM) {
            arh[set] = *reinterpret_cast<const f16x8*>(rh + roff + 32 * s);
            arl[set] = *reinterpret_cast<const f16x8*>(rl + roff + 32 * s);
        }
    };
    auto load_b = [&](auto setc, int slot) {
        constexpr int set = decltype(setc)::value;
        const _Float16* sb = ring + slot * TG_BLK + lane * 8;
#pragma unroll
        for (int nj = 0; nj < NG; ++nj) {
            bh[set][nj] = *reinterpret_cast<const f16x8*>(sb + nj * BLKF);
            bl[set][nj] = *reinterpret_cast<const f16x8*>(sb + (NG + nj) * BLKF);
        }
    };
#define T3_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
    // the remainder tile's three MFMAs use the B fragments of column tile `wave` (waves 0..2; wave 3 multiplies tile 2
    // again and its result is masked): picked with uniform bit masks -- a branch would split the scheduling region of a
    // slab, and indexing the fragment array with the wave number would put the array in scratch
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned m0 = wave == 0 ? ~0u : 0u, m1 = wave == 1 ? ~0u : 0u, m2 = wave >= 2 ? ~0u : 0u;
    auto pick = [&](const f16x8 (&bv)[NG]) -> f16x8 {
        const u32x4 r = (__builtin_bit_cast(u32x4, bv[0]) & m0) | (__builtin_bit_cast(u32x4, bv[1]) & m1) |
                        (__builtin_bit_cast(u32x4, bv[2]) & m2);
        return __builtin_bit_cast(f16x8, r);
    };
    // One slab.  Its fragments were requested during the slab before, so the first twelve MFMAs go out at once and
    // execute while the wave waits at the barrier; behind the barrier slab s + 1 has landed for everybody (its
    // fragments can be requested) and the slot slab s - 1 was read from is free for the copy of slab s + 3.  The copies
    // and the fragment requests sit between the blocks of MFMAs, in the shadow of the matrix pipe.
    auto body = [&](auto setc, int s, int slot) {
        constexpr int set = decltype(setc)::value;
        __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));   // lgkmcnt(0)
        T3_STAMP(0);
        f16x8 bhw, blw;
        if (REM) { bhw = pick(bh[set]); blw = pick(bl[set]); }
#pragma unroll
        for (int nj = 0; nj < NG; ++nj)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) T3_MFMA(ah[set][mi], bh[set][nj], acc[mi][nj]);
        if (REM) T3_MFMA(arh[set], bhw, accr);
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(1);
        if (s + 2 < s1) WAIT_VM(2); else WAIT_VM(0);
        T3_STAMP(2);
        __builtin_amdgcn_s_barrier();
        T3_STAMP(3);
        if (s + 3 < s1)
            stage_group16x3(hi + (size_t)(s + 3) * SLABF, lo + (size_t)(s + 3) * SLABF, ring + ((slot + 3) & (TRING - 1)) * TG_BLK,
                            g, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(4);
        // the fragments of the next slab (the last slab re-reads itself: no branch inside the scheduling region) are
        // requested one by one in the shadows of the remaining MFMAs
        {
            const int sn = s + 1 < s1 ? s + 1 : s;
            load_a(std::integral_constant<int, 1 - set>{}, sn);
            load_b(std::integral_constant<int, 1 - set>{}, (slot + (s + 1 < s1 ? 1 : 0)) & (TRING - 1));
        }
#pragma unroll
        for (int nj = 0; nj < NG; ++nj)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) T3_MFMA(ah[set][mi], bl[set][nj], acc[mi][nj]);
        if (REM) T3_MFMA(arh[set], blw, accr);
#pragma unroll
        for (int nj = 0; nj < NG; ++nj)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) T3_MFMA(al[set][mi], bh[set][nj], acc[mi][nj]);
        if (REM) T3_MFMA(arl[set], bhw, accr);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
        }
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(7);
    };
    if (s0 < s1) {
        // the group's first slab: its A fragments now, its B fragments behind the barrier that says it has landed
        load_a(std::integral_constant<int, 0>{}, s0);
        const int later = min(2, s1 - 1 - s0);
        if (later == 2) WAIT_VM(4); else if (later == 1) WAIT_VM(2); else WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        load_b(std::integral_constant<int, 0>{}, 0);
    }
    int slot = 0;
    for (int s = s0; s < s1; s += 2) {
        body(std::integral_constant<int, 0>{}, s, slot);
        if (s + 1 < s1) body(std::integral_constant<int, 1>{}, s + 1, (slot + 1) & (TRING - 1));
        slot = (slot + 2) & (TRING - 1);
    }
#undef T3_MFMA
#ifdef CSC_PHASE_PROF
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&g_prof[8 + i][blockIdx.x % PROF_BUCKETS], (unsigned long long)t3acc[i]);
        atomicAdd(&g_prof[7][blockIdx.x % PROF_BUCKETS], (unsigned long long)(s1 - s0));
    }
#endif
    __syncthreads();
}

#ifdef CSC_SCREEN_CHECK
// Development aid (-DCSC_SCREEN_CHECK): the screen of ONE unit through the workgroup ring of the three-product loop; the level
// kernel then prints every unit whose maximum differs from what screen_blocks16 found.
// The screen of a unit: the hi x hi product alone (one MFMA per tile and slab instead of three), good to
//     eps1 = (2^-10 + 2^-14 + 1e-6) am + 2^-25 sqrt(L) (1 + am)
// like the single-term filter of the untiled kernels (two operand roundings of 2^-11 each, fp32 accumulation of L <= 682
// products, fp16 subnormals).  Only the hi blocks of a group-slab are staged (3 KiB, contiguous in the image: 768 B per
// wave, one copy each), and the Hankel structure halves the A reads: the fragment of row tile mi + 2 at slab s IS the
// fragment of row tile mi at slab s + 1 (rows 32 further = one slab further along the window), so the six fragments
// w[0..5] rotate with period 3 and each slab requests only two new ones.
__device__ __forceinline__ void stage_group16x1(const _Float16* hi_slab, _Float16* lslot, int g, int wave, int lane) {
    if (lane < 48)
        __builtin_amdgcn_global_load_lds((gptr_t)(hi_slab + NG * g * BLKF + wave * 384 + lane * 8),
                                         (lptr_t)(lslot + wave * 384), 16, 0, 0);
}

template <bool REM>
__device__ __forceinline__ void gemm16x1_group(const _Float16* __restrict__ hi, int g, int s0, int s1, const _Float16* rh,
                                               int R16T_CS, int rowbase, _Float16* ring, f32x4 (&acc)[MT][NG], f32x4& accr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nj = 0; nj < NG; ++nj) acc[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accr = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16* pa = rh + (lr & 7) * R16T_CS + rowbase + 64 * wave + (lr & 8) + 8 * lq;    // W(i) = pa[16 i ..]
    const _Float16* pr = rh + (lr & 7) * R16T_CS + rowbase + CSC_TILE_ROWS + (lr & 8) + 8 * lq;
    WAIT_VM(0);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (s0 + q < s1) stage_group16x1(hi + (size_t)(s0 + q) * SLABF, ring + q * TG_BLK, g, wave, lane);
    f16x8 w[6], ar[3], b[3][NG];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned m0 = wave == 0 ? ~0u : 0u, m1 = wave == 1 ? ~0u : 0u, m2 = wave >= 2 ? ~0u : 0u;
#define T1_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
    // slab s with the fragments of phase ph: w[2 ph .. 2 ph + 3 (mod 6)], ar[ph], b[ph]; requests those of slab s + 1
    // (the window has one slab of look-ahead, so the last slab's requests stay inside it and are simply not used)
    auto body = [&](auto phc, int s, int slot) {
        constexpr int ph = decltype(phc)::value, pn = (ph + 1) % 3;
        constexpr int i0 = (2 * ph) % 6, i1 = (2 * ph + 1) % 6, i2 = (2 * ph + 2) % 6, i3 = (2 * ph + 3) % 6;
        constexpr int l0 = (2 * ph + 4) % 6, l1 = (2 * ph + 5) % 6;
        __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));   // lgkmcnt(0)
        w[l0] = *reinterpret_cast<const f16x8*>(pa + 32 * s + 64);
        w[l1] = *reinterpret_cast<const f16x8*>(pa + 32 * s + 80);
        if (REM) ar[pn] = *reinterpret_cast<const f16x8*>(pr + 32 * s + 32);
        f16x8 bw;
        if (REM) {
            const u32x4 r = (__builtin_bit_cast(u32x4, b[ph][0]) & m0) | (__builtin_bit_cast(u32x4, b[ph][1]) & m1) |
                            (__builtin_bit_cast(u32x4, b[ph][2]) & m2);
            bw = __builtin_bit_cast(f16x8, r);
        }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            T1_MFMA(w[i0], b[ph][nj], acc[0][nj]);
            T1_MFMA(w[i1], b[ph][nj], acc[1][nj]);
            T1_MFMA(w[i2], b[ph][nj], acc[2][nj]);
            T1_MFMA(w[i3], b[ph][nj], acc[3][nj]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < s1) WAIT_VM(1); else WAIT_VM(0);                 // my copy of slab s + 1 has landed
        __builtin_amdgcn_s_barrier();                                // ... everybody's; slab s - 1's slot is free
        if (s + 3 < s1) stage_group16x1(hi + (size_t)(s + 3) * SLABF, ring + ((slot + 3) & (TRING - 1)) * TG_BLK, g, wave, lane);
        {
            const _Float16* sb = ring + ((slot + (s + 1 < s1 ? 1 : 0)) & (TRING - 1)) * TG_BLK + lane * 8;
#pragma unroll
            for (int nj = 0; nj < NG; ++nj) b[pn][nj] = *reinterpret_cast<const f16x8*>(sb + nj * BLKF);
        }
        __builtin_amdgcn_sched_barrier(0);
        T1_MFMA(w[i0], b[ph][2], acc[0][2]);
        T1_MFMA(w[i1], b[ph][2], acc[1][2]);
        T1_MFMA(w[i2], b[ph][2], acc[2][2]);
        T1_MFMA(w[i3], b[ph][2], acc[3][2]);
        if (REM) T1_MFMA(ar[ph], bw, accr);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (s0 < s1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const f16x8*>(pa + 32 * s0 + 16 * i);
        if (REM) ar[0] = *reinterpret_cast<const f16x8*>(pr + 32 * s0);
        const int later = min(2, s1 - 1 - s0);
        if (later == 2) WAIT_VM(2); else if (later == 1) WAIT_VM(1); else WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int nj = 0; nj < NG; ++nj) b[0][nj] = *reinterpret_cast<const f16x8*>(ring + lane * 8 + nj * BLKF);
    }
    int s = s0, slot = 0;
    while (s < s1) {
        body(std::integral_constant<int, 0>{}, s, slot);
        ++s; slot = (slot + 1) & (TRING - 1);
        if (s >= s1) break;
        body(std::integral_constant<int, 1>{}, s, slot);
        ++s; slot = (slot + 1) & (TRING - 1);
        if (s >= s1) break;
        body(std::integral_constant<int, 2>{}, s, slot);
        ++s; slot = (slot + 1) & (TRING - 1);
    }
#undef T1_MFMA
    __syncthreads();
}
#endif

// The screen of a whole node, without a barrier: every wave works on its own -- wave w takes the 16-column blocks
// q = w, w + 4, ... (q = block + 9 (bank tile + ntile * row block)) against ALL the row tiles of the row block (16, or 17
// with the remainder tile), one MFMA per row tile and slab on one B fragment.  Nobody else in the workgroup needs that
// fragment, so each wave streams its own blocks L2 -> LDS (global_load_lds, one 1 KiB fragment-major block per copy)
// through a private ring of SCREEN_AHEAD slots, across block boundaries, and waits only for its own copies (vmcnt).
// The A side is the Hankel window: slab s uses the fragments W(2 s) .. W(2 s + 16) (W(j) = window elements 16 j + ...),
// i.e. the set moves on by two per slab -- twenty registers in rotation (period 10, the phases below), two
// ds_read_b128 per slab and wave.  A block's maximum is folded into key[] of its (row block, tile, column group) unit
// with an LDS atomicMax (keys from fkey, preset to fkey(-inf)); masks are arithmetic: rows >= P, and the
// ntile * 144 - K padding columns, which the packed bank puts in front.
#define SCREEN_AHEAD 6                 // ring slots per wave (4 x 6 KiB = the 24 KiB ring of the three-product loop)
template <bool REM>
__device__ __forceinline__ void screen_blocks16(const LevelArgs& a, const _Float16* __restrict__ hi_all, unsigned* key,
                                                const unsigned short* brange, int ntile, int nslab, int nrb,
                                                const _Float16* rh, int R16T_CS, _Float16* ring, int nsup, const int* spick,
                                                const unsigned char* need = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int P = a.P, npad = ntile * CSC_TILE_COLS - a.K;
    constexpr int MTW = REM ? 17 : 16;
    _Float16* myring = ring + wave * (SCREEN_AHEAD * BLKF);
    // block position (block of the tile, tile, row block), moved on by the wave's stride of four blocks
    struct Pos { int blk, tile, rb; };
    auto advance = [&](Pos& p) {
        p.blk += 4;
        if (p.blk >= NT) { p.blk -= NT; if (++p.tile == ntile) { p.tile = 0; ++p.rb; } }
    };
    // slab range of a block: outside it the block is zero padding (brange[block + 9 tile] = s0 | s1 << 8, built by the caller)
    // `need` (behind the low-rank screen): only the 16-column blocks marked there are multiplied; the others keep the bound they have
    auto skipped = [&](const Pos& p) -> bool {
        return need && !__builtin_amdgcn_readfirstlane((int)need[p.blk + NT * (p.tile + ntile * p.rb)]);
    };
    auto block_range = [&](const Pos& p, int& s0, int& s1) {
        const int pk = __builtin_amdgcn_readfirstlane((int)brange[p.blk + NT * p.tile]);
        s0 = pk & 0xff;
        s1 = pk >> 8;
        if (skipped(p)) s0 = s1 = 0;
    };
    // the wave's stream of B blocks: copy cursor (starts one stride before the wave's first block)
    Pos cp = {wave - 4, 0, 0};
    int cs = 0, ce = 0, cslot = 0;
    bool more = true;                                                // the cursor has not run out (uniform)
    const _Float16* cb = hi_all;
    auto cursor_copy = [&]() {
        while (cs >= ce) {
            if (!more) return;
            advance(cp);
            if (cp.rb >= nrb) { more = false; return; }
            block_range(cp, cs, ce);
            cb = hi_all + (size_t)cp.tile * nslab * SLABF + cp.blk * BLKF + lane * 8;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(cb + (size_t)cs * SLABF), (lptr_t)(myring + cslot * BLKF), 16, 0, 0);
        ++cs;
        cslot = cslot == SCREEN_AHEAD - 1 ? 0 : cslot + 1;
    };
    WAIT_VM(0);
#pragma unroll
    for (int i = 0; i < SCREEN_AHEAD; ++i) cursor_copy();
    f16x8 W[20], b[2];
    f32x4 acc[MTW];
    int slot = 0;                                                    // ring slot of the slab being multiplied
    // the stream's first fragment (the later ones are read one slab ahead, in body)
    if (more) WAIT_VM(SCREEN_AHEAD - 1); else WAIT_VM(0);
    b[0] = *reinterpret_cast<const f16x8*>(myring + lane * 8);
    const _Float16* pa = rh;
    // one slab; FIRST: the block's first slab starts the accumulators from zero (an inline-constant C operand)
    auto body = [&](auto phc, auto firstc, int s) {
        constexpr int PH = decltype(phc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        // outstanding here: the copies of the next SCREEN_AHEAD - 1 stream slabs; the oldest is the one read now
        if (more) WAIT_VM(SCREEN_AHEAD - 2); else WAIT_VM(0);
        const int nslot = slot == SCREEN_AHEAD - 1 ? 0 : slot + 1;
        b[(PH + 1) & 1] = *reinterpret_cast<const f16x8*>(myring + nslot * BLKF + lane * 8);
        W[(2 * PH + 18) % 20] = *reinterpret_cast<const f16x8*>(pa + 32 * s + 16 * 18);
        W[(2 * PH + 19) % 20] = *reinterpret_cast<const f16x8*>(pa + 32 * s + 16 * 19);
#pragma unroll
        for (int mi = 0; mi < MTW; ++mi)
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(2 * PH + mi) % 20], b[PH & 1],
                                                             FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                           // the slot's fragment is in registers: refill it
        cursor_copy();
        slot = nslot;
    };
    Pos p = {wave, 0, 0};
    for (; p.rb < nrb; advance(p)) {
        int s0, s1;
        block_range(p, s0, s1);
        if (skipped(p)) continue;
        const int rowbase = p.rb * CSC_TILE_ROWS;
        const int rows = min(P - rowbase, CSC_TILE_ROWS + ((REM && p.rb == nrb - 1) ? 16 : 0));   // valid rows of the block
        const bool colok = p.tile * CSC_TILE_COLS + 16 * p.blk + lr >= npad;
        float lmax = -INFINITY;
        if (s0 >= s1) {
            if (4 * lq < rows) lmax = 0.f;
        } else {
            pa = rh + (lr & 7) * R16T_CS + rowbase + (lr & 8) + 8 * lq;
#pragma unroll
            for (int i = 0; i < 18; ++i) W[i] = *reinterpret_cast<const f16x8*>(pa + 32 * s0 + 16 * i);
            int s = s0;
            // a block ends at any phase; the fragment read ahead for the next block's first slab moves to b[0]
#define SCREEN_STEP(PH) body(std::integral_constant<int, PH>{}, std::false_type{}, s); \
            if (++s >= s1) { if (((PH) & 1) == 0) b[0] = b[1]; break; }
            body(std::integral_constant<int, 0>{}, std::true_type{}, s);
            if (++s >= s1) {
                b[0] = b[1];
            } else {
                while (true) {
                    SCREEN_STEP(1) SCREEN_STEP(2) SCREEN_STEP(3) SCREEN_STEP(4) SCREEN_STEP(5)
                    SCREEN_STEP(6) SCREEN_STEP(7) SCREEN_STEP(8) SCREEN_STEP(9) SCREEN_STEP(0)
                }
            }
#undef SCREEN_STEP
            if (nsup > 0 && colok) {
                // a later sibling: the entries similar to an earlier pick (bit (row - pick position + P - 1) of the pick's
                // similarity row for this column's atom) take the value 0, as in the reference's mask multiplication
                const int atom = a.meta[(size_t)p.tile * META_INTS + 16 * p.blk + lr];
                const int Wd = a.W;
                for (int q = 0; q < nsup; ++q) {
                    const uint32_t* srow = a.sim_bits + ((size_t)spick[CSC_MAX_CONNECTIONS + q] * a.K + atom) * Wd;
                    const int dpb = rowbase + 4 * lq - spick[q] + P - 1;         // bit of (mi, rg) = (0, 0); >= 0
                    const int w0 = dpb >> 5, b0 = dpb & 31;
                    uint32_t wd[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) wd[i] = srow[min(w0 + i, Wd - 1)];
                    // the lane's 272 bits, shifted down so that the bit of (mi, rg) sits at 16 mi + rg
#pragma unroll
                    for (int i = 0; i < 9; ++i) wd[i] = __funnelshift_r(wd[i], wd[i + 1], b0);
#pragma unroll
                    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg)
                            if ((wd[mi >> 1] >> (16 * (mi & 1) + rg)) & 1u) acc[mi][rg] = 0.f;
                }
            }
            if (rows >= CSC_TILE_ROWS) {                             // (uniform) the 16 main row tiles are all inside
#pragma unroll
                for (int mi = 0; mi < 16; mi += 2)
                    lmax = fmaxf(lmax, fmaxf(fmaxf(fmaxf(acc[mi][0], acc[mi][1]), fmaxf(acc[mi][2], acc[mi][3])),
                                             fmaxf(fmaxf(acc[mi + 1][0], acc[mi + 1][1]), fmaxf(acc[mi + 1][2], acc[mi + 1][3]))));
                if (REM) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        lmax = fmaxf(lmax, 256 + 4 * lq + rg < rows ? acc[MTW - 1][rg] : -INFINITY);
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        lmax = fmaxf(lmax, 16 * mi + 4 * lq + rg < rows ? acc[mi][rg] : -INFINITY);
            }
        }
        lmax = wave_max(colok ? lmax : -INFINITY);
        if (lane == 63) atomicMax(&key[p.blk / NG + NG * (p.tile + ntile * p.rb)], fkey(lmax == lmax ? lmax : INFINITY));
    }
    WAIT_VM(0);
}

// masking of the entries outside the matrix + suppression by the first t siblings (bits from the similarity table);
// returns the lane's maximum over its unsuppressed entries (remainder tile included), lane_supp = it holds a
// suppressed one
template <bool REM>
__device__ __forceinline__ float group16x3_finish(const LevelArgs& a, const int* perm, const Tiled16Lds* tl, int g, int t,
                                                  int rowbase, int (&permv)[NG], int& permr, f32x4 (&acc)[MT][NG],
                                                  f32x4& accr, bool& lane_supp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int K = a.K, P = a.P, W = a.W;
    const int prow = rowbase + 64 * wave + 4 * lq;
    const int rrow = rowbase + CSC_TILE_ROWS + 4 * lq;
#pragma unroll
    for (int nj = 0; nj < NG; ++nj) permv[nj] = perm[16 * (NG * g + nj) + lr];
    permr = wave < NG ? perm[16 * (NG * g + wave) + lr] : -1;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
        if (rowbase + 64 * wave + 16 * mi + 15 >= P) {               // wave-uniform
#pragma unroll
            for (int nj = 0; nj < NG; ++nj)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (prow + 16 * mi + rg >= P) acc[mi][nj][rg] = -INFINITY;
        }
#pragma unroll
    for (int nj = 0; nj < NG; ++nj)
        if (permv[nj] < 0) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][nj] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    if (REM) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            if (permr < 0 || rrow + rg >= P) accr[rg] = -INFINITY;
    } else {
        accr = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    lane_supp = false;
    for (int q = 0; q < t; ++q) {
        const uint32_t* srow = a.sim_bits + (size_t)tl->sib_atom[q] * K * W;
        {
            const int dpb = prow - tl->sib_pos[q] + P - 1;           // bit index of (mi, rg) = (0, 0); >= 0
            const int w0 = dpb >> 5, b0 = dpb & 31;
#pragma unroll
            for (int nj = 0; nj < NG; ++nj) {
                const uint32_t* row = srow + (size_t)max(permv[nj], 0) * W;
                const uint32_t wd0 = row[min(w0, W - 1)], wd1 = row[min(w0 + 1, W - 1)], wd2 = row[min(w0 + 2, W - 1)];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int off = b0 + 16 * mi + rg;           // < 32 + 52
                        const uint32_t wd = off < 32 ? wd0 : (off < 64 ? wd1 : wd2);
                        const bool hit = ((wd >> (off & 31)) & 1u) && acc[mi][nj][rg] > -INFINITY;
                        acc[mi][nj][rg] = hit ? __uint_as_float(SUPP_BITS) : acc[mi][nj][rg];
                        lane_supp |= hit;
                    }
            }
        }
        if (REM && permr >= 0) {
            const int dpb = rrow - tl->sib_pos[q] + P - 1;
            const int w0 = dpb >> 5, b0 = dpb & 31;
            const uint32_t* row = srow + (size_t)permr * W;
            const uint32_t wd0 = row[min(w0, W - 1)], wd1 = row[min(w0 + 1, W - 1)];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int off = b0 + rg;                             // < 32 + 4
                const uint32_t wd = off < 32 ? wd0 : wd1;
                const bool hit = ((wd >> (off & 31)) & 1u) && accr[rg] > -INFINITY;
                accr[rg] = hit ? __uint_as_float(SUPP_BITS) : accr[rg];
                lane_supp |= hit;
            }
        }
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nj = 0; nj < NG; ++nj)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) lmax = fmaxf(lmax, acc[mi][nj][rg]);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) lmax = fmaxf(lmax, accr[rg]);
    return lmax;
}

// Where a node's screen bounds for its child t wait between screen16_kernel and mmp_level_tiled16_kernel: in the
// workspace row of that child, which is written only at the end of the launch that read them, or, on the last level,
// which has no child rows, in the node's own row of the idle generation (the siblings' launches follow each other).
__device__ __forceinline__ float* screen_table(const LevelArgs& a, size_t s, int j, int d, int t) {
    const size_t row = d + 1 < a.Kmax ? s * a.ts.F[d + 1] + j + (size_t)t * a.ts.pw[d] : s * a.ts.F[d] + j;
    return reinterpret_cast<float*>(a.ws[(d + 1) & 1] + row * (size_t)a.NS);
}

// Stage 1 of the tiled fp16 path as a kernel of its own: it needs 14.5 KB + the ring of LDS and ~150 VGPRs, the
// three-product stage 75 KB and 256, so apart the screen runs three workgroups per CU instead of two.  One workgroup per
// node: residual -> ||r||, fp16 copies scaled to norm 64 (the same numbers mmp_level_tiled16_kernel builds), the
// single-term maximum of every (row block, bank tile, column group) unit -> screen_table.
__global__ __launch_bounds__(WG, 3) void screen16_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, K = a.K, L = a.L, P = a.P, d = a.depth;
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nslab = (L + SLAB - 1) / SLAB;
    const int nrb_all = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const bool rem = nrb_all > 1 && (P - (nrb_all - 1) * CSC_TILE_ROWS) <= 16;
    const int nrb = rem ? nrb_all - 1 : nrb_all;
    const int nunits = nrb * ntile * NG;
    const int R16T_CS = r16t_stride(P, nslab * SLAB);
    // LDS: red [4] doubles | rh [8][R16T_CS] halves | ring [TRING][TG_BLK] halves | key [nunits] | brange [9 ntile] shorts
    double* red = lds_raw;
    _Float16* rh = reinterpret_cast<_Float16*>(red + 4);
    _Float16* ring = rh + R16_COPIES * R16T_CS;
    unsigned* key = reinterpret_cast<unsigned*>(ring + TRING * TG_BLK);
    unsigned short* brange = reinterpret_cast<unsigned short*>(key + nunits);
    int* spick = reinterpret_cast<int*>(brange + ((NT * ntile + 1) & ~1));       // [2][CSC_MAX_CONNECTIONS]: pos, atom
#if defined(CSC_PHASE_PROF) && defined(SCREEN16_PROF)
    unsigned long long s16_t = __builtin_readcyclecounter();
#define S16_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); \
        atomicAdd(&g_prof[i][blockIdx.x % PROF_BUCKETS], n_ - s16_t); s16_t = n_; } } while (0)
#else
#define S16_STAMP(i)
#endif
    const int Fd = a.ts.F[d];
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    // this launch bounds what child t of every node can select: the entries similar to the children picked before it
    // (launches 0 .. t - 1 of this level) count as 0, as they will in the three-product kernel
    const int t = a.sibling;
    if ((long long)j + (long long)t * a.ts.pw[d] >= a.ts.F[d + 1]) return;       // the node has no child t
    if (t > 0 && a.flags[s * Fd + j]) return;                                    // the node went to the fp32 kernel
    if (tid < t) {
        const size_t cid = s * a.NN + a.ts.off[d + 1] + j + (size_t)tid * a.ts.pw[d];
        spick[tid] = a.node_pos[cid];
        spick[CSC_MAX_CONNECTIONS + tid] = a.node_atom[cid];
    }
    const double* rin = a.resid_in + (d == 0 ? s * (size_t)N : (s * Fd + j) * (size_t)a.NS);
    double v = 0.0;
    for (int i = tid; i < N; i += WG) { const double x = rin[i]; v = fma(x, x, v); }
    v = wave_sum(v);
    if (lane == 63) red[wave] = v;
    for (int u = tid; u < nunits; u += WG) key[u] = fkey(-INFINITY);
    // slab range of every 16-column block (s0 | s1 << 8), from the tiles' slab masks: the masks first, all loads at once,
    // into the ring (idle until the blocks are swept) -- read one by one inside the loop over the slabs each was an L2
    // round trip of its own, 17 in a row
    {
        uint32_t* lmask = reinterpret_cast<uint32_t*>(ring);
        for (int i = tid; i < ntile * nslab; i += WG)
            lmask[i] = (uint32_t)a.meta[(size_t)(i / nslab) * META_INTS + CSC_TILE_COLS + i % nslab];
        __syncthreads();
        for (int e = tid; e < NT * ntile; e += WG) {
            const uint32_t* mk = lmask + (e / NT) * nslab;
            const int blk = e % NT;
            int lo = nslab, hi = 0;
            for (int q = 0; q < nslab; ++q)
                if ((mk[q] >> blk) & 1u) { lo = min(lo, q); hi = q + 1; }
            brange[e] = (unsigned short)(min(lo, hi) | (hi << 8));
        }
    }
    __syncthreads();
    const double rnorm2 = red[0] + red[1] + red[2] + red[3];
    const float am2 = *a.amax2;
    // the cases mmp_level_tiled16_kernel hands to fp32 without looking at the table (its ||r||^2 is summed in another
    // order: the window here is a little wider, so that a node it takes always finds its table written)
    if (!(am2 > 1e-6f && am2 < 1e4f && rnorm2 > 0.5e-30 && rnorm2 < 2e30)) return;
    const double sc = 64.0 / sqrt(rnorm2);
    build_r16t_copies(rin, N, sc, rh, nullptr, R16T_CS);
    __syncthreads();
    S16_STAMP(28);
    const _Float16* hi_all = reinterpret_cast<const _Float16*>(a.bank16);
    float* tab = screen_table(a, s, j, d, t);
    const unsigned char* need = nullptr;
    if (a.lr_u16) {
        // behind screen_lr_kernel: its table holds, per 16-column block, an upper bound (less e1) of what the block can hold
        // and, behind the blocks, a lower bound LB of the node's maximum.  A block whose bound + e1 + band / 2 (the most the
        // three-product sweep can find there) stays 2 bands below LB - band / 2 (the least its running maximum ends at)
        // cannot make its unit visited and keeps its bound; the single-term product tightens the others (a quarter of the
        // blocks of the units in reach, a twentieth of all).  (One row block: block b = 9 tile + blk, unit b / 3.)
        unsigned char* nd = reinterpret_cast<unsigned char*>(spick + 2 * CSC_MAX_CONNECTIONS);
        const float am = sqrtf(am2);
        const float toacc = 64.f * split_scale(am2);
        const float e1 = (9.765625e-4f + 6.103515625e-05f + 1e-6f) * am + 2.98023223876953125e-08f * sqrtf((float)L) * (1.f + am);
        const float band = 2.f * ((EPS3_ACC + 9.5367431640625e-07f) * am + 1.862645149230957e-09f * sqrtf((float)L) * (1.f + am));
        const int nblk = NT * ntile;
        const float lb = tab[nblk];
        for (int u = tid; u < nunits; u += WG) {
            float kept = -INFINITY;                                              // the unit's blocks that are not multiplied
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const float ub = tab[NG * u + i];
#ifdef CSC_LR_CHECK
                const bool skip = false;
#else
                const bool skip = ub + (e1 + 0.5f * band) * toacc < lb - 2.5f * band * toacc; // (false on NaN: multiplied)
#endif
                nd[NG * u + i] = skip ? 0 : 1;
                if (skip) kept = fmaxf(kept, ub);
            }
            key[u] = fkey(kept);
        }
        need = nd;
        __syncthreads();
    }
    S16_STAMP(29);
    if (rem) screen_blocks16<true>(a, hi_all, key, brange, ntile, nslab, nrb, rh, R16T_CS, ring, t, spick, need);
    else screen_blocks16<false>(a, hi_all, key, brange, ntile, nslab, nrb, rh, R16T_CS, ring, t, spick, need);
    S16_STAMP(30);
    __syncthreads();
#ifdef CSC_LR_CHECK
    if (a.lr_u16) {
        // development aid: every unit multiplied; its single-term maximum against the low-rank bounds
        const float am = sqrtf(am2);
        const float toacc = 64.f * split_scale(am2);
        const float e1 = (9.765625e-4f + 6.103515625e-05f + 1e-6f) * am + 2.98023223876953125e-08f * sqrtf((float)L) * (1.f + am);
        float mx = -INFINITY;
        for (int u = 0; u < nunits; ++u) mx = fmaxf(mx, fkey_inv(key[u]));
        const float lbc = tab[NT * ntile];
        for (int u = tid; u < nunits; u += WG) {
            const float got = fkey_inv(key[u]), ub = fmaxf(fmaxf(tab[NG * u], tab[NG * u + 1]), tab[NG * u + 2]);
            if (got > ub + 2.f * e1 * toacc) printf("LRCHK ub: s %d j %d d %d t %d unit %d single %g bound %g (e1 %g)\n", (int)s, j, d, t, u, got, ub, e1 * toacc);
        }
        if (tid == 0 && s < 24) {
            // units with a block in reach of the lower bound, and the blocks themselves
            const float band = 2.f * ((EPS3_ACC + 9.5367431640625e-07f) * am + 1.862645149230957e-09f * sqrtf((float)L) * (1.f + am));
            int n0 = 0, n1 = 0;
            for (int u = 0; u < nunits; ++u) {
                bool un = false;
                for (int i = 0; i < NG; ++i) {
                    const bool nb = !(tab[NG * u + i] + (e1 + 0.5f * band) * toacc < lbc - 2.5f * band * toacc);
                    n1 += nb; un |= nb;
                }
                n0 += un;
            }
            printf("LRNEED d %d t %d: units in reach %d, blocks multiplied %d of %d (lb %g max %g)\n", d, t, n0, n1, NG * n0, lbc, mx);
        }
        if (tid == 0 && mx < lbc - e1 * toacc) printf("LRCHK lb: s %d j %d d %d t %d max %g lb %g\n", (int)s, j, d, t, mx, lbc);
    }
    __syncthreads();
#endif
    for (int u = tid; u < nunits; u += WG) tab[u] = fkey_inv(key[u]);
}

// ------------------------------------------------------------------------------------------------
// low-rank screen of large dictionaries (CSC_FLAG_LOWRANK; csc_bank_lowrank installs the section it reads)
// ------------------------------------------------------------------------------------------------
// A finely sampled dictionary spans little: the 9570 atoms of BASELINE config 3 lie within 1e-3 (median; 4e-3 at most) of
// the span of 64 orthonormal vectors U.  With V = U^T a_k (coefficients, rounded to fp16: V16) and res_k = a_k - U V16_k
//     <r_window, a_k> = (r_window U) V16_k + <r_window, res_k>,     |<r_window, res_k>| <= ||r|| ||res_k||
// so C~ = T V16 with T = R U -- the Hankel product against 64 columns instead of K, then a GEMM of depth 64: 34 MFMAs per
// 16-column block where the single-term screen issues ~190 -- bounds every correlation of a column to within
//     slack_k = ||res_k|| + (2^-11 + sqrt(64) ((L + 40) 2^-24 + 2^-20 + ..) + 64 2^-24 + 2^-20) ||V16_k||   (units of ||r||)
// (T rounded to fp16; T itself from the fp16 matrix cores with two-term operands: the accumulation of L products per
// entry, the operand splits, 64 entries per row; the fp32 accumulation of the second product; fp16 subnormals of T:
// lr_slack_kernel).  Any basis is valid -- the slack is
// computed from the V16 actually used -- as long as its vectors are orthonormal (||T_p|| <= ||r_window||).
// screen_lr_kernel leaves per unit  max(C~ + slack) - e1  in the screen table (mmp_level_tiled16_kernel adds e1 back) and,
// behind the units, a LOWER bound of the node's maximum: max(C~ - slack).  screen16_kernel, launched behind, then takes
// the single-term product only of the units whose low-rank bound can still reach that maximum (a tenth of them).
#define LR_R 64                        // rank of the screen
#define LR_PLANE_H (272 * 32)          // halves of one plane of T16 in LDS: [row][32 columns] (k-step of the second product)
#define LR_CHUNK_U 51                  // later siblings: units per chunk of the sweep (the chunk's similarity bands wait in LDS)
#define LR_CHUNK_C (LR_CHUNK_U * 3 * 16)   // = 2448 columns, at most 10 per thread

// U16 [hi | lo][k-step][16-column block][lane][j] = the two fp16 terms of 64 U[l][i], l = 32 ks + 8 (lane >> 4) + j,
// i = 16 block + (lane & 15): B fragments of v_mfma_f32_16x16x32_f16 (the scale keeps the lo terms of all but the
// smallest entries out of the fp16 subnormals, as split_scale does for the bank)
__global__ void lr_pack_u16_kernel(const double* __restrict__ basis, int L, int nks, _Float16* __restrict__ u16) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)nks * (LR_R / 16) * 512;
    if (e >= total) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), nb = (int)((e >> 9) % (LR_R / 16)), ks = (int)(e / (512 * (LR_R / 16)));
    const int l = 32 * ks + 8 * (lane >> 4) + j, i = 16 * nb + (lane & 15);
    const double x = l < L ? 64.0 * basis[(size_t)i * L + l] : 0.0;
    const _Float16 h = (_Float16)x;
    u16[e] = h;
    u16[total + e] = (_Float16)(x - (double)h);
}
// V16[block][ks][lane][j] = fl16(2^sa <U_i, a_k>), i = 32 ks + 8 (lane >> 4) + j, k = atom of column 16 block + (lane & 15)
__global__ void lr_pack_v_kernel(const double* __restrict__ basis, const double* __restrict__ bank64, int K, int L,
                                 const int32_t* __restrict__ meta, const float* __restrict__ amax2,
                                 _Float16* __restrict__ v16, size_t total) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) % (LR_R / 32));
    const size_t blk = e / (512 * (LR_R / 32));
    const int tile = (int)(blk / NT), ni = (int)(blk % NT);
    const int k = meta[tile * META_INTS + 16 * ni + (lane & 15)];
    const int i = 32 * ks + 8 * (lane >> 4) + j;
    double v = 0.0;
    if (k >= 0) {
        const double* u = basis + (size_t)i * L;
        const double* ak = bank64 + (size_t)k * L;
        for (int l = 0; l < L; ++l) v = fma(u[l], ak[l], v);
    }
    v16[e] = (_Float16)(v * (double)split_scale(*amax2));
}
// slack[column] (see above), from the V16 just written; one wave per column
__global__ __launch_bounds__(64) void lr_slack_kernel(const double* __restrict__ basis, const double* __restrict__ bank64,
                                                      int K, int L, const int32_t* __restrict__ meta,
                                                      const float* __restrict__ amax2, const _Float16* __restrict__ v16,
                                                      float* __restrict__ slack) {
    const int col = blockIdx.x, lane = threadIdx.x;
    const int tile = col / CSC_TILE_COLS, c = col % CSC_TILE_COLS;
    const int k = meta[tile * META_INTS + c];
    if (k < 0) { if (lane == 0) slack[col] = 0.f; return; }
    __shared__ double v[LR_R];
    const double inv = 1.0 / (double)split_scale(*amax2);
    if (lane < LR_R) {
        const int i = lane, ks = i >> 5, lq = (i >> 3) & 3, j = i & 7;
        const size_t blk = (size_t)tile * NT + c / 16;
        v[i] = (double)v16[((blk * (LR_R / 32) + ks) * 64 + 16 * lq + (c & 15)) * 8 + j] * inv;
    }
    __syncthreads();
    double rr = 0.0, vv = 0.0;
    for (int l = lane; l < L; l += 64) {
        double x = bank64[(size_t)k * L + l];
        for (int i = 0; i < LR_R; ++i) x = fma(-v[i], basis[(size_t)i * L + l], x);
        rr = fma(x, x, rr);
    }
    if (lane < LR_R) vv = v[lane] * v[lane];
    rr = wave_sum(rr);
    vv = wave_sum(vv);
    if (lane == 63) {
        // T to fp16: 2^-11; T itself: sqrt(64) entries per row, each off by (L + 40) 2^-24 (fp32 accumulation of the
        // rh uh chain over the padded window, the cross chain and the final sum) + 2^-20 (operand splits, the dropped
        // product) + 2^-28 sqrt(L + 32) (fp16 subnormals of the lo terms); 64 2^-24: fp32 accumulation of the second
        // product; 2^-20: fp16 subnormals of T
        const double relT = (double)(L + 40) * 5.9604644775390625e-08 + 9.5367431640625e-07 + 3.725290298461914e-09 * sqrt((double)(L + 32));
        const double rel = 4.8828125e-04 + 8.0 * relT + 64.0 * 5.9604644775390625e-08 + 9.5367431640625e-07;
        slack[col] = (float)((sqrt(rr) + rel * sqrt(vv)) * 1.000001 + 1e-30);
    }
}

// Atom summary of a similarity table: bit k' of row k <=> the similarity row (k, k') has a bit set at any shift; and, per
// atom k, the lowest and the highest bit set in any of its rows (range[k], range[K + k]; preset to INT_MAX / -1): similar
// means nearly aligned, so the band of shifts is narrow (config 3: 29 of 519 at most).
// One thread per (k, k'), one word per half wave.
__global__ __launch_bounds__(WG) void bank_simatoms_kernel(const uint32_t* __restrict__ sim_bits, int W, int K, int sa_w,
                                                           uint32_t* __restrict__ out, int* __restrict__ range) {
    const int k = blockIdx.y, kp = blockIdx.x * WG + threadIdx.x;
    uint32_t any = 0;
    int lo = 0x7fffffff, hi = -1;
    if (kp < K) {
        const uint32_t* row = sim_bits + ((size_t)k * K + kp) * W;
        for (int w = 0; w < W; ++w) {
            const uint32_t x = row[w];
            if (x) {
                if (!any) lo = 32 * w + __builtin_ctz(x);
                hi = 32 * w + 31 - __builtin_clz(x);
            }
            any |= x;
        }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(any != 0);
    if ((threadIdx.x & 31) == 0 && (kp >> 5) < sa_w) out[(size_t)k * sa_w + (kp >> 5)] = (uint32_t)(m >> (threadIdx.x & 32));
    if (m) {                                                         // (uniform per wave)
        lo = wave_min(lo);
        hi = -wave_min(-hi);
        if ((threadIdx.x & 63) == 0) { atomicMin(&range[k], lo); atomicMax(&range[K + k], hi); }
    }
}

// (SUPP: a later sibling's launch, for callers without the atom summary of their similarity table.  Its similarity-row
// loads sit behind a per-lane branch, and a loop with such a branch in it waits for every load in flight at each turn; a
// kernel of its own, so that the first sibling's loop keeps its registers and waits by count)
template <bool SUPP>
__global__ __launch_bounds__(WG, SUPP ? 2 : 3) void screen_lr_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int N = a.N, K = a.K, L = a.L, P = a.P, d = a.depth;
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nslab = (L + SLAB - 1) / SLAB;
    const int nunits = ntile * NG;                                    // (one row block: the host checks P <= 272)
    const int CS = r16t_stride(P, nslab * SLAB);                      // halves per shifted copy of the residual window
    // LDS: red [4] doubles | rh, rl [8][CS] halves each, later plane 1 of T16 | plane 0 of T16 [272][32] halves |
    // key [nblk + 1] (upper bound per 16-column block, the node's lower bound) | spick [5][8]: the earlier picks' position,
    // atom, lowest and highest row they can suppress, first similarity word in use | simw [2][LR_CHUNK_C] (later siblings):
    // the similarity band of every column of the current chunk, per earlier pick
    double* red = lds_raw;
    _Float16* rh = reinterpret_cast<_Float16*>(red + 4);
    _Float16* rl = rh + R16_COPIES * CS;
    _Float16* t16p1 = rh;
    _Float16* t16p0 = rl + R16_COPIES * CS;
    unsigned* key = reinterpret_cast<unsigned*>(t16p0 + LR_PLANE_H);
    const int nblk = ntile * NT;
    int* spick = reinterpret_cast<int*>(key + nblk + 1);
    uint32_t* simw = reinterpret_cast<uint32_t*>(spick + 5 * CSC_MAX_CONNECTIONS);    // [2][LR_CHUNK_C] (later siblings, mode 2)
    const int Fd = a.ts.F[d];
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    const int t = a.sibling;
    if ((long long)j + (long long)t * a.ts.pw[d] >= a.ts.F[d + 1]) return;       // the node has no child t
    if (t > 0 && a.flags[s * Fd + j]) return;                                    // the node went to the fp32 kernel
    if (tid < t) {
        const size_t cid = s * a.NN + a.ts.off[d + 1] + j + (size_t)tid * a.ts.pw[d];
        const int pos = a.node_pos[cid], atom = a.node_atom[cid];
        spick[tid] = pos;
        spick[CSC_MAX_CONNECTIONS + tid] = atom;
        // the band of shifts at which anything is similar to this atom (bit dp of a similarity row: shift dp - (P - 1)):
        // the rows this pick can suppress, and the word its band starts in
        int blo = 0, bhi = 2 * P - 2;
        if (a.simrange && atom >= 0 && atom < K) { blo = max(a.simrange[atom], 0); bhi = min(a.simrange[K + atom], 2 * P - 2); }
        spick[2 * CSC_MAX_CONNECTIONS + tid] = pos + blo - (P - 1);
        spick[3 * CSC_MAX_CONNECTIONS + tid] = pos + bhi - (P - 1);
        spick[4 * CSC_MAX_CONNECTIONS + tid] = ((bhi >> 5) - (blo >> 5) < 2) ? (blo >> 5) : -1;     // -1: a wide band, all words
    }
    const double* rin = a.resid_in + (d == 0 ? s * (size_t)N : (s * Fd + j) * (size_t)a.NS);
    double v = 0.0;
    for (int i = tid; i < N; i += WG) { const double x = rin[i]; v = fma(x, x, v); }
    v = wave_sum(v);
    if (lane == 63) red[wave] = v;
    for (int u = tid; u <= nblk; u += WG) key[u] = fkey(-INFINITY);
    __syncthreads();
    const double rnorm2 = red[0] + red[1] + red[2] + red[3];
    const float am2 = *a.amax2, am = sqrtf(am2);
    if (!(am2 > 1e-6f && am2 < 1e4f && rnorm2 > 0.5e-30 && rnorm2 < 2e30)) return;   // (as screen16_kernel)
    {
        // the residual as a two-term fp16 number, scaled to norm 64, in 8 shifted copies each (as mmp_level_tiled16_kernel)
        const double sc = 64.0 / sqrt(rnorm2);
        build_r16t_copies(rin, N, sc, rh, rl, CS);
    }
    __syncthreads();
#if defined(CSC_PHASE_PROF) && !defined(SCREEN16_PROF)
    unsigned long long lr_t = __builtin_readcyclecounter();
#define LR_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); \
        atomicAdd(&g_prof[i][blockIdx.x % PROF_BUCKETS], n_ - lr_t); lr_t = n_; } } while (0)
#else
#define LR_STAMP(i)
#endif
    // ---- T = R U on the fp16 matrix cores, both operands two-term fp16 numbers (the window copies above; the basis
    // scaled by 64 as B fragments hi | lo in the bank image, straight from L2 a k-step ahead): per k-step and 16 x 16 tile
    // rh uh into one accumulator, rh ul + rl uh into another (rl ul ~ 2^-22 is dropped) -- the cross terms are 2^-11 of the
    // product, so the fp32 accumulation error is that of the first chain alone: L 2^-24 instead of 3 L 2^-24.
    // Wave w: row tiles 4 w .. 4 w + 3 x the basis' four 16-column blocks, and block w of the remainder tile (rows 256 ..);
    // two passes of two column blocks (72 accumulators); pass p writes plane p of T16 [k-step of the second product][row][32]:
    // plane 0 beside the window copies, plane 1 over them once every wave has left the product --------------------------
    {
        const int nks = nslab;                                         // k-steps of 32
        const _Float16* uh = a.lr_u16 + lane * 8;
        const _Float16* ul = uh + (size_t)nks * (LR_R / 16) * 512;
        const int c8 = lr & 7;
        const _Float16* ah0 = rh + c8 * CS + (64 * wave + lr - c8) + 8 * lq;      // (tile i: + 16 i; the remainder tile: below)
        const _Float16* ahr = rh + c8 * CS + (CSC_TILE_ROWS + lr - c8) + 8 * lq;
        const int dlo = R16_COPIES * CS;                               // rl - rh
        auto pass = [&](int p, bool extra, _Float16* plane) __attribute__((always_inline)) {
            f32x4 hh[9], xx[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { hh[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; xx[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            const bool odd = wave & 1;                                 // the remainder tile's column block (extra): 2 p + odd
            // (no second buffer for the B fragments: the lo pair is re-loaded behind its one use, a third of a k-step
            // early, the hi pair at the end of the k-step -- the other waves of the SIMD cover what latency is left)
            f16x8 bh[2], bl[2];
            auto load_u = [&](f16x8 (&B)[2], const _Float16* src, int ks) __attribute__((always_inline)) {
                const int kk = min(ks, nks - 1);                       // (past the end: the last k-step again, unused)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    B[n] = *reinterpret_cast<const f16x8*>(src + ((size_t)kk * (LR_R / 16) + 2 * p + n) * 512);
            };
            load_u(bh, uh, 0);
            load_u(bl, ul, 0);
            for (int ks = 0; ks < nks; ++ks) {
                f16x8 Ah[5], Al[5];
#pragma unroll
                for (int i = 0; i < 4; ++i) Ah[i] = *reinterpret_cast<const f16x8*>(ah0 + 16 * i + 32 * ks);
                Ah[4] = *reinterpret_cast<const f16x8*>(ahr + 32 * ks);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) xx[2 * i + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[i], bl[n], xx[2 * i + n], 0, 0, 0);
                if (extra) xx[8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[4], (odd ? bl[1] : bl[0]), xx[8], 0, 0, 0);
                load_u(bl, ul, ks + 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) Al[i] = *reinterpret_cast<const f16x8*>(ah0 + dlo + 16 * i + 32 * ks);
                Al[4] = *reinterpret_cast<const f16x8*>(ahr + dlo + 32 * ks);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) hh[2 * i + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[i], bh[n], hh[2 * i + n], 0, 0, 0);
                if (extra) hh[8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[4], (odd ? bh[1] : bh[0]), hh[8], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) xx[2 * i + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[i], bh[n], xx[2 * i + n], 0, 0, 0);
                if (extra) xx[8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[4], (odd ? bh[1] : bh[0]), xx[8], 0, 0, 0);
                load_u(bh, uh, ks + 1);
            }
            if (p == 1) __syncthreads();                               // every wave has left the window copies: plane 1 goes there
            // accumulators (64 x 64 <r window, u> / ||r||) -> T16 = fl16(64 <r window, u> / ||r||)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        plane[(64 * wave + 16 * i + 4 * lq + rg) * 32 + 16 * n + lr] =
                            (_Float16)((hh[2 * i + n][rg] + xx[2 * i + n][rg]) * 0.015625f);
            if (extra) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    plane[(CSC_TILE_ROWS + 4 * lq + rg) * 32 + 16 * (wave & 1) + lr] = (_Float16)((hh[8][rg] + xx[8][rg]) * 0.015625f);
            }
        };
        pass(0, wave < 2, t16p0);
        pass(1, wave >= 2, t16p1);
    }
    __syncthreads();
    LR_STAMP(29);
    // ---- C~ = T V16.  Two waves share a unit's rows -- wave w takes row tiles 0 .. 8 (w even) or 9 .. 16 (w odd) of the
    // units u = (w >> 1), + 2, + 4, ... (three 16-column blocks each): 9 x 2 A fragments stay in registers (72 VGPRs; all
    // 17 x 2 with their 68 accumulators spilled at 256 registers, and every reload drained the fragments in flight), three
    // workgroups per CU.  B fragments (and the block's slack, one float per lane: the lane's column) come straight from L2,
    // a block ahead, in two buffers.  A block's upper bound is one ds_max_u32 of 16 lanes on the block's key (no full wave
    // reduction on the vector pipe, which the products' issue shares); the lower bound stays a running value per lane ----
    const int npad = ntile * CSC_TILE_COLS - K;
    const float toacc = 64.f * split_scale(am2);                      // units of ||r|| -> accumulator units
    const float e1 = (9.765625e-4f + 6.103515625e-05f + 1e-6f) * am + 2.98023223876953125e-08f * sqrtf((float)L) * (1.f + am);
    const int rows = min(P, 272);
    float lbmax = -INFINITY;
    const unsigned key_off = (unsigned)(size_t)(lptr_t)key;           // LDS byte offset of the block keys
    static_assert(NT == 3 * NG, "a unit is three 16-column blocks");
    auto sweep = [&](auto half_tag, auto mode_tag, int ubeg, int uend) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_tag)::value;
        // 0: no suppression (first sibling, or no earlier pick reaches this half's rows); 1: similarity words looked up block
        // by block (three dependent loads a block: any number of picks, any band of shifts); 2: the bands of the chunk's
        // columns wait in LDS (gathered with every load in flight at once: at most two earlier picks, bands of at most 32
        // shifts, the summary installed)
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool SUPPNOW = MODE == 1;
        constexpr int ROWLO = 144 * HALF, ROWHI = HALF ? 271 : 143;
        constexpr int NS = HALF ? 8 : 9;                               // row tiles 0 .. 8 | 9 .. 16
        f16x8 A[NS][LR_R / 32];
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int ks = 0; ks < LR_R / 32; ++ks)
                A[i][ks] = *reinterpret_cast<const f16x8*>((ks ? t16p1 : t16p0) + (16 * (9 * HALF + i) + lr) * 32 + 8 * lq);
        // rows past the matrix (the remainder tile: the second half's last) start from -1e30 instead of 0 and stay there:
        // no masks behind the products
        f32x4 initl = {0.f, 0.f, 0.f, 0.f};
        if (HALF) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                if (CSC_TILE_ROWS + 4 * lq + rg >= rows) initl[rg] = -1e30f;
        }
        // (mode 2) per earlier pick, as scalars: the rows it can suppress, and whether any is in this half
        int pk_lo[2] = {0, 0}, pk_hi[2] = {-1, -1};
        bool pk_on[2] = {false, false};
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q >= t) continue;
                pk_lo[q] = __builtin_amdgcn_readfirstlane(spick[2 * CSC_MAX_CONNECTIONS + q]);
                pk_hi[q] = __builtin_amdgcn_readfirstlane(spick[3 * CSC_MAX_CONNECTIONS + q]);
                pk_on[q] = !(pk_hi[q] < ROWLO || pk_lo[q] > ROWHI);
            }
        }
        f16x8 B0[LR_R / 32], B1[LR_R / 32];
        float S0, S1;
        auto load_b = [&](f16x8 (&Bx)[LR_R / 32], float& Sx, int b) __attribute__((always_inline)) {
            const int bb = min(b, nblk - 1);                           // (past the end: the last block again, unused)
#pragma unroll
            for (int ks = 0; ks < LR_R / 32; ++ks)
                Bx[ks] = *reinterpret_cast<const f16x8*>(a.lr_v16 + ((size_t)bb * (LR_R / 32) + ks) * 512 + lane * 8);
            Sx = a.lr_slack[16 * bb + lr];                             // (column 144 tile + 16 blk + lr = 16 b + lr)
        };
        auto block = [&](const f16x8 (&Bcur)[LR_R / 32], float Scur, int b, float& lbl) __attribute__((always_inline)) {
            f32x4 acc[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][0], Bcur[0], (HALF && i == NS - 1) ? initl : (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < LR_R / 32; ++ks)
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][ks], Bcur[ks], acc[i], 0, 0, 0);
            const bool colok = 16 * b + lr >= npad;
            const float slk = Scur * toacc;
            if constexpr (HALF) mfma_settle(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
            else mfma_settle(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], acc[NS - 1]);
            auto lane_max = [&]() __attribute__((always_inline)) -> float {
                float m0 = vmax3(acc[0][0], acc[0][1], acc[0][2]), m1 = vmax3(acc[0][3], acc[1][0], acc[1][1]);
                m0 = vmax3(m0, acc[1][2], acc[1][3]); m1 = vmax3(m1, acc[2][0], acc[2][1]);
                m0 = vmax3(m0, acc[2][2], acc[2][3]); m1 = vmax3(m1, acc[3][0], acc[3][1]);
                m0 = vmax3(m0, acc[3][2], acc[3][3]); m1 = vmax3(m1, acc[4][0], acc[4][1]);
                m0 = vmax3(m0, acc[4][2], acc[4][3]); m1 = vmax3(m1, acc[5][0], acc[5][1]);
                m0 = vmax3(m0, acc[5][2], acc[5][3]); m1 = vmax3(m1, acc[6][0], acc[6][1]);
                m0 = vmax3(m0, acc[6][2], acc[6][3]); m1 = vmax3(m1, acc[7][0], acc[7][1]);
                if constexpr (HALF) return vmax3(m0, m1, vmax3(acc[7][2], acc[7][3], -INFINITY));
                m0 = vmax3(m0, acc[7][2], acc[7][3]); m1 = vmax3(m1, acc[NS - 1][0], acc[NS - 1][1]);
                return vmax3(m0, m1, vmax3(acc[NS - 1][2], acc[NS - 1][3], -INFINITY));
            };
            float lo = lane_max(), hi;
            hi = lo;
            bool touched = SUPPNOW && colok;
            int atom = 0;
            if (SUPPNOW) {
                // with the atom summary of the similarity table: only columns similar to an earlier pick at some shift
                // have rows to look up, and most blocks hold none
                const int tile = b / NT, blk = b - tile * NT;
                atom = max(a.meta[(size_t)tile * META_INTS + 16 * blk + lr], 0);
                if (a.simatoms) {
                    unsigned hit = 0;
                    for (int q = 0; q < t; ++q)
                        hit |= (a.simatoms[(size_t)spick[CSC_MAX_CONNECTIONS + q] * a.sa_w + (atom >> 5)] >> (atom & 31)) & 1u;
                    touched = touched && hit;
                }
            }
            if (SUPPNOW && touched) {
                // a later sibling: entries similar to an earlier pick take the value 0 (as in screen_blocks16)
                const int Wd = a.W;
                bool any = false;
                for (int q = 0; q < t; ++q) {
                    const int qlo = spick[2 * CSC_MAX_CONNECTIONS + q], qhi = spick[3 * CSC_MAX_CONNECTIONS + q];
                    if (qhi < ROWLO || qlo > ROWHI) continue;          // (uniform) this pick reaches none of this half's rows
                    const uint32_t* srow = a.sim_bits + ((size_t)spick[CSC_MAX_CONNECTIONS + q] * K + atom) * Wd;
                    const int w0 = spick[4 * CSC_MAX_CONNECTIONS + q];
                    if (w0 >= 0) {
                        // the band sits in two words: 64 bits from bit 32 w0; entry (slot i, rg) is bit sh0 + 16 i + rg of them
                        const unsigned long long bits = (unsigned long long)srow[w0] |
                                                        ((unsigned long long)(w0 + 1 < Wd ? srow[w0 + 1] : 0u) << 32);
                        const int sh0 = ROWLO + 4 * lq - spick[q] + P - 1 - 32 * w0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            if (ROWLO + 16 * i + 15 < qlo || ROWLO + 16 * i > qhi) continue;        // (uniform)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const int sh = sh0 + 16 * i + rg;
                                if ((unsigned)sh < 64u && ((bits >> sh) & 1ull)) { if (acc[i][rg] > -1e29f) acc[i][rg] = 0.f; any = true; }
                            }
                        }
                    } else {
                        const int dpb = ROWLO + 4 * lq - spick[q] + P - 1;   // bit of (slot 0, rg 0) of this half; >= 0
                        const int wf = dpb >> 5, b0 = dpb & 31;
                        uint32_t wd[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) wd[i] = srow[min(wf + i, Wd - 1)];
#pragma unroll
                        for (int i = 0; i < 5; ++i) wd[i] = __funnelshift_r(wd[i], wd[i + 1], b0);
#pragma unroll
                        for (int i = 0; i < NS; ++i)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg)
                                if ((wd[i >> 1] >> (16 * (i & 1) + rg)) & 1u) { if (acc[i][rg] > -1e29f) acc[i][rg] = 0.f; any = true; }
                    }
                }
                if (any) {
                    hi = lane_max();
                    lo = fminf(lo, hi);                               // (the unsuppressed maximum is no lower bound any more)
                }
            }
            if (MODE == 2) {
                const int cl = 16 * b + lr - 3 * 16 * ubeg;            // the column within the chunk
                const uint32_t band0 = simw[cl], band1 = t > 1 ? simw[LR_CHUNK_C + cl] : 0u;
                if (__builtin_amdgcn_ballot_w64(colok && (band0 | band1))) {    // (uniform) a column of the block is similar to a pick
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (!pk_on[q]) continue;                       // (uniform) this pick reaches none of this half's rows
                        const uint32_t band = colok ? (q ? band1 : band0) : 0u;  // bit i: row pk_lo + i of this column is suppressed
                        const int rel0 = ROWLO + 4 * lq - pk_lo[q];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            if (ROWLO + 16 * i + 15 < pk_lo[q] || ROWLO + 16 * i > pk_hi[q]) continue;   // (uniform: scalars)
                            // the slot's four bits: band >> (rel0 + 16 i), for shifts of -3 .. 35
                            const int sft = rel0 + 16 * i + 4;
                            const uint32_t sb = (unsigned)sft < 36u ? (uint32_t)(((unsigned long long)band << 4) >> sft) : 0u;
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg)
                                if ((sb >> rg) & 1u) { if (acc[i][rg] > -1e29f) acc[i][rg] = 0.f; any = true; }
                        }
                    }
                    if (any) {
                        hi = lane_max();
                        lo = fminf(lo, hi);
                    }
                }
            }
            // the lane's bounds: largest (suppressed) entry + slack (e1 comes off when the table is written; a NaN keeps
            // the key of a NaN, above +inf: the block is multiplied); lower bound: - slack
            // (two quad steps on the DPP network first, then one lane in four: the LDS serialises the lanes of a ds_max_u32
            // on one address, and 64 of them per block kept it busy half the time.  Written out: atomicMax on one address
            // would come back from the compiler as a loop over the lanes)
            float ubq;
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                         : "=&v"(ubq) : "v"(colok ? hi + slk : -INFINITY));
            if ((lane & 3) == 0)
                asm volatile("ds_max_u32 %0, %1" :: "v"(key_off + 4u * (unsigned)b), "v"(fkey(ubq)) : "memory");
            lbl = fmaxf(lbl, colok ? lo - slk : -INFINITY);
        };
        const int ufirst = ubeg + (((wave >> 1) ^ ubeg) & 1);          // the wave pair's first unit of [ubeg, uend)
        load_b(B0, S0, 3 * ufirst);
        load_b(B1, S1, 3 * ufirst + 1);
        for (int u = ufirst; u < uend; u += 4) {                       // two units a turn: the buffers come round after six blocks
            block(B0, S0, 3 * u, lbmax);
            load_b(B0, S0, 3 * u + 2);
            block(B1, S1, 3 * u + 1, lbmax);
            load_b(B1, S1, 3 * u + 6);
            block(B0, S0, 3 * u + 2, lbmax);
            load_b(B0, S0, 3 * u + 7);
            if (u + 2 < uend) {
                block(B1, S1, 3 * u + 6, lbmax);
                load_b(B1, S1, 3 * u + 8);
                block(B0, S0, 3 * u + 7, lbmax);
                load_b(B0, S0, 3 * u + 12);
                block(B1, S1, 3 * u + 8, lbmax);
                load_b(B1, S1, 3 * u + 13);
            }
        }
    };
    bool reach = false;                                                // (uniform) can an earlier pick suppress a row of this wave's half?
    if (SUPP)
        for (int q = 0; q < t; ++q)
            reach |= !(spick[3 * CSC_MAX_CONNECTIONS + q] < 144 * (wave & 1) || spick[2 * CSC_MAX_CONNECTIONS + q] > ((wave & 1) ? 271 : 143));
    bool gathered = SUPP && a.simatoms && t <= 2;                      // (uniform) mode 2's conditions
    if (SUPP)
        for (int q = 0; q < t; ++q)
            gathered = gathered && spick[3 * CSC_MAX_CONNECTIONS + q] - spick[2 * CSC_MAX_CONNECTIONS + q] < 32;
    const bool reaches = SUPP && __builtin_amdgcn_readfirstlane((int)reach);
    if (SUPP && __builtin_amdgcn_readfirstlane((int)gathered)) {
        // chunk by chunk: the similarity band (the <= 32 shifts from the first one anything is similar at) of every column
        // of the chunk, per earlier pick, gathered with all loads of a stage in flight at once -- atoms, summary bits,
        // similarity words: three round trips per chunk instead of three per block
        for (int ubeg = 0; ubeg < nunits; ubeg += LR_CHUNK_U) {
            const int uend = min(nunits, ubeg + LR_CHUNK_U);
            const int cbeg = 3 * 16 * ubeg, cend = 3 * 16 * uend;
            constexpr int PER = (LR_CHUNK_C + WG - 1) / WG;
            int atoms[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = cbeg + tid + WG * i;
                atoms[i] = c < cend ? a.meta[(size_t)(c / CSC_TILE_COLS) * META_INTS + c % CSC_TILE_COLS] : -1;
            }
            uint32_t sa[PER][2];
#pragma unroll
            for (int i = 0; i < PER; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    sa[i][q] = q < t ? a.simatoms[(size_t)spick[CSC_MAX_CONNECTIONS + q] * a.sa_w + (max(atoms[i], 0) >> 5)] : 0u;
            uint32_t w[PER][2][2];
#pragma unroll
            for (int i = 0; i < PER; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool hit = q < t && atoms[i] >= 0 && ((sa[i][q] >> (atoms[i] & 31)) & 1u);
                    w[i][q][0] = 0u; w[i][q][1] = 0u;
                    if (hit) {
                        const int blo = spick[2 * CSC_MAX_CONNECTIONS + q] - spick[q] + P - 1;   // the band's first bit
                        const uint32_t* srow = a.sim_bits + ((size_t)spick[CSC_MAX_CONNECTIONS + q] * K + atoms[i]) * a.W;
                        w[i][q][0] = srow[blo >> 5];
                        w[i][q][1] = (blo >> 5) + 1 < a.W ? srow[(blo >> 5) + 1] : 0u;
                    }
                }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int cl = tid + WG * i;
                if (cl < LR_CHUNK_C)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int blo = q < t ? spick[2 * CSC_MAX_CONNECTIONS + q] - spick[q] + P - 1 : 0;
                        simw[q * LR_CHUNK_C + cl] = (uint32_t)((((unsigned long long)w[i][q][1] << 32) | w[i][q][0]) >> (blo & 31));
                    }
            }
            __syncthreads();
            if (reaches) {
                if (wave & 1) sweep(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, ubeg, uend);
                else sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, ubeg, uend);
            } else {
                if (wave & 1) sweep(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, ubeg, uend);
                else sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ubeg, uend);
            }
            __syncthreads();                                           // (the next chunk's bands go where this one's are)
        }
    } else if (reaches) {
        if (wave & 1) sweep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 0, nunits);
        else sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0, nunits);
    } else {
        if (wave & 1) sweep(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, 0, nunits);
        else sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, nunits);
    }
    if (!(lbmax == lbmax)) lbmax = -INFINITY;
    lbmax = wave_max(lbmax);
    if (lane == 63) atomicMax(&key[nblk], fkey(lbmax));
    LR_STAMP(30);
    __syncthreads();
    // the table: per 16-column block max(C~ + slack) - e1 | the node's lower bound
    float* tab = screen_table(a, s, j, d, t);
    for (int u = tid; u <= nblk; u += WG) tab[u] = fkey_inv(key[u]) - (u < nblk ? e1 * toacc : 0.f);
}

__global__ __launch_bounds__(WG, 2) void mmp_level_tiled16_kernel(const LevelArgs a) {
    extern __shared__ double lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = a.N, K = a.K, L = a.L, P = a.P, d = a.depth;
    const int ntile = (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS;
    const int nslab = (L + SLAB - 1) / SLAB;
    // row blocks of 256 positions; a last block of at most 16 positions rides along with the block before it
    const int nrb_all = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const bool rem = nrb_all > 1 && (P - (nrb_all - 1) * CSC_TILE_ROWS) <= 16;
    const int nrb = rem ? nrb_all - 1 : nrb_all;
    const int nunits = nrb * ntile * NG;
    const int R16T_CS = r16t_stride(P, nslab * SLAB);

    // ---- LDS carve-up: r64 [Npad] | rh, rl [8][R16T_CS] halves each | ring [TRING][TG_BLK] halves | NodeLds | Tiled16Lds
    const int Npad = (N + 3) & ~3;
    double* r64 = lds_raw;
    _Float16* rh = reinterpret_cast<_Float16*>(r64 + Npad);
    _Float16* rl = rh + R16_COPIES * R16T_CS;
    _Float16* ring = rl + R16_COPIES * R16T_CS;
    NodeLds* sl = reinterpret_cast<NodeLds*>(ring + TRING * TG_BLK);
    PROF_INIT(16);                                                   // (the shared phases' marks: slots 16.., 0..15 are this kernel's own counters)
    Tiled16Lds* tl = reinterpret_cast<Tiled16Lds*>(sl + 1);

    const int Fd = a.ts.F[d], Fc = a.ts.F[d + 1];
    size_t s;
    int j;
    node_of_block(Fd, s, j);
    const int stride = a.ts.pw[d];
    int nch = 0;
    for (int r = 0; r < a.connections; ++r) {
        if ((long long)j + (long long)r * stride < Fc) ++nch; else break;
    }
    uint8_t* flag = a.flags + s * Fd + j;
    // one launch per child: this one selects child t0 of every node that has one; a node an earlier launch handed to the
    // fp32 kernel stays there
    const int t0 = a.sibling;
    if (t0 >= nch) return;
    if (t0 > 0 && *flag) return;
    const double* rin = a.resid_in + (d == 0 ? s * (size_t)N : (s * Fd + j) * (size_t)a.NS);
    double rnorm2, first[4];
    node_prologue(a, sl, r64, nullptr, Npad, 0, rin, s, j, rnorm2, first, d);
    if (tid == 0 && t0 == 0) *flag = 0;
    if (tid < t0) {                                                  // the children picked before: suppression
        const size_t cq = s * a.NN + a.ts.off[d + 1] + j + (size_t)tid * stride;
        tl->sib_pos[tid] = a.node_pos[cq];
        tl->sib_atom[tid] = a.node_atom[cq];
    }
    const float am2 = *a.amax2, am = sqrtf(am2);
    // NaN / Inf / extreme scales: fp32 (workgroup-uniform)
    if (!(am2 > 1e-6f && am2 < 1e4f && rnorm2 > 1e-30 && rnorm2 < 1e30)) {
        if (tid == 0) { *flag = 1; if (a.stats) atomicAdd(&a.stats[3], 1); }
        return;
    }
    const float band = 2.f * ((EPS3_ACC + 9.5367431640625e-07f) * am + 1.862645149230957e-09f * sqrtf((float)L) * (1.f + am));
    const float unscale = 1.f / (64.f * split_scale(am2));           // accumulators -> units of ||r||
    {
        // the residual as a two-term fp16 number, scaled to norm 64, in 8 shifted copies each
        const double sc = 64.0 / sqrt(rnorm2);
        build_r16t_copies(r64, N, sc, rh, rl, R16T_CS);
    }
    __syncthreads();
    PROF_MARK(0);
    const _Float16* hi_all = reinterpret_cast<const _Float16*>(a.bank16);      // scaled tiles: hi, then lo
    const _Float16* lo_all = hi_all + (size_t)ntile * nslab * SLABF;

    f32x4 acc[MT][NG], accr;
    int permv[NG], permr;
    // slab range of every (bank tile, column group): outside it the group's blocks are zero padding
    {
        // (the tiles' slab masks first, all loads at once, into the idle ring: see screen16_kernel)
        uint32_t* lmask = reinterpret_cast<uint32_t*>(ring);
        for (int i = tid; i < ntile * nslab; i += WG)
            lmask[i] = (uint32_t)a.meta[(size_t)(i / nslab) * META_INTS + CSC_TILE_COLS + i % nslab];
        __syncthreads();
        for (int e = tid; e < NG * ntile; e += WG) {
            const uint32_t* mk = lmask + (e / NG) * nslab;
            const int g = e % NG;
            int lo = nslab, hi = 0;
            for (int q = 0; q < nslab; ++q)
                if ((mk[q] >> (NG * g)) & 7u) { lo = min(lo, q); hi = q + 1; }
            tl->urange[e] = (unsigned short)(min(lo, hi) | (hi << 8));
        }
    }
    __syncthreads();
    // One unit: product + masks + suppression -> the unscaled lane maximum; the entries stay in acc / accr.
    // begin_unit16x3(u, buf) has been issued (by the unit before, or by the caller); `next` >= 0: the unit the sweep is
    // likely to take next, begun into the other buffer as soon as the ring is idle.
    int unit_rowbase = 0;
    const int* unit_perm = tl->permbuf[0];
#ifdef CSC_PHASE_PROF
    // unit-level clocks (thread 0): slots 0 product, 1 begin-next + masks, 2 maximum + barriers, 3 candidates + loop
    unsigned long long unit_t = __builtin_readcyclecounter();
#define UNIT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); \
        atomicAdd(&g_prof[i][blockIdx.x % PROF_BUCKETS], n_ - unit_t); unit_t = n_; } } while (0)
#else
#define UNIT_STAMP(i)
#endif
    auto unit = [&](int u, int t, bool screen, int buf, int next, bool& lane_supp) -> float {
        const int g = u % NG, tile = (u / NG) % ntile, rb = u / (NG * ntile);
        const bool with_rem = rem && rb == nrb - 1;                  // uniform
        const int rowbase = rb * CSC_TILE_ROWS;
        const int pk = tl->urange[g + NG * tile];
        const int s0 = pk & 0xff, s1 = pk >> 8;
        const _Float16* th = hi_all + (size_t)tile * nslab * SLABF;
        const _Float16* tlo = lo_all + (size_t)tile * nslab * SLABF;
        unit_rowbase = rowbase;
        unit_perm = tl->permbuf[buf];
        if (s0 >= s1) {                                              // nothing was staged: the map has to land by itself
            WAIT_VM(0);
            __syncthreads();
        }
        float lmax;
        UNIT_STAMP(3);
        if (with_rem) {
#ifdef CSC_SCREEN_CHECK
            if (screen) gemm16x1_group<true>(th, g, s0, s1, rh, R16T_CS, rowbase, ring, acc, accr); else
#endif
            gemm16x3_group<true>(th, tlo, g, s0, s1, rh, rl, R16T_CS, rowbase, ring, acc, accr);
            UNIT_STAMP(0);
            if (next >= 0) begin_unit16x3(a, hi_all, lo_all, tl, next, buf ^ 1, ntile, nslab, ring);
            lmax = group16x3_finish<true>(a, unit_perm, tl, g, t, rowbase, permv, permr, acc, accr, lane_supp);
        } else {
#ifdef CSC_SCREEN_CHECK
            if (screen) gemm16x1_group<false>(th, g, s0, s1, rh, R16T_CS, rowbase, ring, acc, accr); else
#endif
            gemm16x3_group<false>(th, tlo, g, s0, s1, rh, rl, R16T_CS, rowbase, ring, acc, accr);
            UNIT_STAMP(0);
            if (next >= 0) begin_unit16x3(a, hi_all, lo_all, tl, next, buf ^ 1, ntile, nslab, ring);
            lmax = group16x3_finish<false>(a, unit_perm, tl, g, t, rowbase, permv, permr, acc, accr, lane_supp);
        }
        UNIT_STAMP(1);
        return lmax;
    };
    // the maximum of a unit over the workgroup (suppressed entries count as 0)
    auto unit_max = [&](int u, int t, bool screen, int buf, int next) -> float {
        bool lane_supp;
        const float lmax = unit(u, t, screen, buf, next, lane_supp);
        const float bv = wave_max(lane_supp ? fmaxf(lmax, 0.f) : lmax);
        __syncthreads();                                              // (the maxima of the unit before have been read)
        if (lane == 63) sl->wmax[wave] = bv;
        __syncthreads();
        UNIT_STAMP(2);
        return fmaxf(fmaxf(sl->wmax[0], sl->wmax[1]), fmaxf(sl->wmax[2], sl->wmax[3]));
    };
    // a unit on its own (no unit before it to begin it)
    auto unit_max_cold = [&](int u, int t, bool screen) -> float {
        __syncthreads();                                              // the ring and permbuf[0] are idle
        begin_unit16x3(a, hi_all, lo_all, tl, u, 0, ntile, nslab, ring);
        return unit_max(u, t, screen, 0, -1);
    };
    // the entries of the unit in acc / accr that reach tauv -> candidate list (flattened index, value)
    auto collect = [&](int u, float tauv) {
        const int g = u % NG, rowbase = unit_rowbase;
        const int rowK = (rowbase + 64 * wave + 4 * lq) * K;
        uint32_t mk[2] = {0u, 0u};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int nj = 0; nj < NG; ++nj) {
                    const int e = (4 * mi + rg) * NG + nj;
                    mk[e >> 5] |= (acc[mi][nj][rg] >= tauv) ? (1u << (e & 31)) : 0u;
                }
        if (mk[0] | mk[1]) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int nj = 0; nj < NG; ++nj) {
                        const int e = (4 * mi + rg) * NG + nj;
                        if ((mk[e >> 5] >> (e & 31)) & 1u) {
                            const int slot = atomicAdd(&sl->cand_cnt, 1);
                            if (slot < CAND_CAP16) {
                                tl->cand_i[slot] = rowK + (16 * mi + rg) * K + unit_perm[16 * (NG * g + nj) + lr];
                                tl->cand_v[slot] = acc[mi][nj][rg];
                            }
                        }
                    }
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            if (accr[rg] >= tauv) {
                const int slot = atomicAdd(&sl->cand_cnt, 1);
                if (slot < CAND_CAP16) {
                    tl->cand_i[slot] = (rowbase + CSC_TILE_ROWS + 4 * lq + rg) * K + permr;
                    tl->cand_v[slot] = accr[rg];
                }
            }
    };
    const float bandu = band / unscale;                              // the band in accumulator units
    {
        // ---- screen: the single-term maximum of every unit, + eps1 + eps3 = an upper bound of what the three-product
        // sweep can find there; the units in descending order of that bound
        const float e1 = (9.765625e-4f + 6.103515625e-05f + 1e-6f) * am + 2.98023223876953125e-08f * sqrtf((float)L) * (1.f + am);
        const float slack = (e1 + 0.5f * band) / unscale;
        const float* tab = screen_table(a, s, j, d, t0);             // written by screen16_kernel, launched in front
        for (int u = tid; u < nunits; u += WG) {
            const float m = tab[u];
            tl->tmax0[u] = m == m ? m + slack : INFINITY;            // (a NaN sorts first and is visited)
        }
#ifdef CSC_SCREEN_CHECK
        __syncthreads();
        for (int u = 0; u < nunits; ++u) {                            // (equal up to the last bit of the two kernels' ||r||)
            const float m = unit_max_cold(u, t0, true);
            if (tid == 0 && m != tab[u] && blockIdx.x < 64)
                printf("block %d depth %d unit %d of %d: screen kernel %g, per-unit product %g\n", blockIdx.x, d, u, nunits, tab[u], m);
        }
#endif
        __syncthreads();
        for (int u = tid; u < nunits; u += WG) {
            const float v = tl->tmax0[u];
            int rank = 0;
            for (int w = 0; w < nunits; ++w) {
                const float x = tl->tmax0[w];
                rank += (x > v || (x == v && w < u)) ? 1 : 0;
            }
            tl->order[rank] = (short)u;
        }
        PROF_MARK(1);
    }

    for (int t = t0; t <= t0; ++t) {
        // ---- the three-product sweep over the units that can reach the band of the running maximum: by descending
        // bound, the first few hold the maximum, and once a unit cannot reach the band no later one can.  Candidates are
        // taken on the way, against the RUNNING maximum: it only rises, so the list is a superset of the entries within
        // the band of the final one and is cut down to them below -- no unit is multiplied twice
        float gmax = -INFINITY;                                       // workgroup-uniform
        int visited = 0;
        if (tid == 0) sl->cand_cnt = 0;
        __syncthreads();
        begin_unit16x3(a, hi_all, lo_all, tl, tl->order[0], 0, ntile, nslab, ring);
        for (int i = 0; i < nunits; ++i, ++visited) {
            const int u = tl->order[i];
            if ((t > 0 ? fmaxf(tl->tmax0[u], 0.f) : tl->tmax0[u]) < gmax - bandu) break;       // (uniform)
#ifdef CSC_PHASE_PROF
            if (tid == 0) atomicAdd(&g_prof[5][blockIdx.x % PROF_BUCKETS], 1ull);
#endif
            const float m = unit_max(u, t, false, i & 1, i + 1 < nunits ? (int)tl->order[i + 1] : -1);
            gmax = fmaxf(gmax, m);
            if (m > -INFINITY && m >= gmax - bandu) collect(u, gmax - bandu);
        }
        WAIT_VM(0);                                                   // the copies begun for a unit the sweep did not take
        __syncthreads();
        PROF_MARK(2);
        const float tau = gmax - bandu;
        // the value 0 of suppressed entries inside the band, or no finite maximum: the fp32 kernel decides
        if ((0.f >= tau && t > 0) || !(gmax > -INFINITY) || !(gmax < INFINITY)) {
            if (tid == 0) { *flag = 1; if (a.stats) atomicAdd(&a.stats[3], 1); }
            return;
        }
        if (sl->cand_cnt > CAND_CAP16) {
            // the list overflowed, perhaps only with entries above an early, lower threshold: once more over the visited
            // units with the final one (rare); more than CAND_CAP16 entries in the band itself: fp32
            __syncthreads();
            if (tid == 0) sl->cand_cnt = 0;
            for (int i = 0; i < visited; ++i) {
                const int u = tl->order[i];
                const float m = unit_max_cold(u, t, false);
                if (m >= tau) collect(u, tau);
            }
            __syncthreads();
            if (sl->cand_cnt > CAND_CAP16) {
                if (tid == 0) { *flag = 1; if (a.stats) atomicAdd(&a.stats[3], 1); }
                return;
            }
        }
        // keep the entries within the band of the final maximum (in place; at most 256: one thread)
        __syncthreads();
        if (tid == 0) {
            const int n = sl->cand_cnt;
            int k = 0;
            for (int q = 0; q < n; ++q)
                if (tl->cand_v[q] >= tau) tl->cand_i[k++] = tl->cand_i[q];
            sl->cand_cnt = k;
        }
        __syncthreads();
        const int ncand_all = sl->cand_cnt;
        PROF_MARK(3);
        // float64 re-score, CAND_CAP candidates at a time; the choice is the float64 maximum, lowest index first,
        // over all of them (thread 0 carries the best of the halves)
        double bestv = -INFINITY;
        int besti = 0x7fffffff;
        for (int c0 = 0; c0 < max(ncand_all, 1); c0 += CAND_CAP) {      // (an empty list still goes through the shared code)
            const int m = min(CAND_CAP, ncand_all - c0);
            __syncthreads();
            for (int q = tid; q < m; q += WG) sl->cand_idx[q] = tl->cand_i[c0 + q];
            __syncthreads();
            rescore_and_choose(a, sl, r64, m, false, d);
            if (tid == 0 && sl->beta_ok && (sl->beta > bestv || (sl->beta == bestv && sl->chosen < besti))) {
                bestv = sl->beta;
                besti = sl->chosen;
            }
        }
        if (ncand_all > CAND_CAP) {
            if (tid == 0 && besti != 0x7fffffff) {
                sl->chosen = besti;
                sl->sup_pos[d] = besti / K;
                sl->sup_atom[d] = besti - (besti / K) * K;
                sl->beta = bestv;
                sl->beta_ok = 1;
            }
            __syncthreads();
        }
        const int jc = j + t * stride;
        const size_t cid = s * a.NN + a.ts.off[d + 1] + jc;
        if (tid == 0) { tl->sib_pos[t] = sl->sup_pos[d]; tl->sib_atom[t] = sl->sup_atom[d]; }
        child_solve_and_write(a, sl, r64, s, jc, cid, d, a.resid_out, false);
        __syncthreads();
        PROF_MARK(9);
    }
}

// ------------------------------------------------------------------------------------------------
// batched atom matching (API parity: utils.computeMaxTruePositives)
// ------------------------------------------------------------------------------------------------
// One wave per problem.  cost[i][j] = 0 where |pos_i - pos_j| <= pos_thr and |<a_i @ pos_i, a_j @ pos_j>|
// >= corr_thr (src/csc/utils.py:34-58); the optimum of linear_sum_assignment on that 0/1 cost is the size
// of a maximum bipartite matching of the zero entries, found here with augmenting paths on bit masks.
#define MATCH_MAX 16
__global__ __launch_bounds__(64) void match_kernel(const double* __restrict__ bank64, int K, int L,
                                                   const int32_t* __restrict__ truth, int kt,
                                                   const int32_t* __restrict__ approx, int ka,
                                                   double pos_thr, double corr_thr, int32_t* __restrict__ tp) {
    __shared__ uint32_t adj[MATCH_MAX];            // adj[i] bit j: true atom i may match approx atom j
    const int lane = threadIdx.x;
    const size_t m = blockIdx.x;
    const int32_t* T = truth + m * kt * 2;
    const int32_t* A = approx + m * ka * 2;
    if (lane < MATCH_MAX) adj[lane] = 0;
    __syncthreads();
    for (int i = 0; i < kt; ++i) {
        const int pi = T[2 * i], ki = T[2 * i + 1];
        if (pi < 0 || ki < 0 || ki >= K) continue;                   // padding
        for (int j = 0; j < ka; ++j) {
            const int pj = A[2 * j], kj = A[2 * j + 1];
            if (pj < 0 || kj < 0 || kj >= K) continue;
            if (fabs((double)(pi - pj)) > pos_thr) continue;           // wave-uniform
            const int dp = pi - pj;                                   // a_j index = l + dp
            double v = dot_range<64>(bank64 + (size_t)ki * L, bank64 + (size_t)kj * L + dp, max(0, -dp),
                                     min(L, L - dp), lane);
            v = wave_sum(v);
            if (lane == 63 && fabs(v) >= corr_thr) adj[i] |= 1u << j;
        }
    }
    __syncthreads();
    if (lane == 0) {
        int owner[MATCH_MAX];                                         // true atom matched to approx atom j
        for (int j = 0; j < MATCH_MAX; ++j) owner[j] = -1;
        int matched = 0;
        for (int i0 = 0; i0 < kt; ++i0) {
            // iterative augmenting-path search from i0 (depth <= kt)
            int stack_i[MATCH_MAX], stack_j[MATCH_MAX], prev_owner_j[MATCH_MAX];
            uint32_t visited = 0;
            int depth = 0;
            stack_i[0] = i0; stack_j[0] = -1;
            bool found = false;
            while (depth >= 0 && !found) {
                const int i = stack_i[depth];
                int j = stack_j[depth] + 1;
                for (; j < ka; ++j) if (((adj[i] >> j) & 1u) && !((visited >> j) & 1u)) break;
                if (j >= ka) { --depth; continue; }
                stack_j[depth] = j;
                visited |= 1u << j;
                if (owner[j] < 0) { found = true; break; }
                stack_i[depth + 1] = owner[j];
                stack_j[depth + 1] = -1;
                ++depth;
            }
            if (found) {
                for (int q = depth; q >= 0; --q) owner[stack_j[q]] = stack_i[q];
                ++matched;
            }
            (void)prev_owner_j;
        }
        tp[m] = matched;
    }
}

// ------------------------------------------------------------------------------------------------
// on-device constrained signal generator (benchmark input: src/csc/dictionary.py:271-324, 326-365)
// ------------------------------------------------------------------------------------------------
// One workgroup per signal: unit-amplitude atoms drawn uniformly (atom, position), a draw rejected when it lies
// closer than pos_thr samples to, or correlates more than corr_thr with, an atom already placed (:302-311); white
// Gaussian noise whose variance is var(clean[|clean| > 1e-3]) / 10^(snr/10) (:318-322).  Random numbers: Philox-4x32-10,
// counter = (draw or sample index, signal index, stream), key = seed -- the result depends on (seed, signal index)
// only, not on the launch.  The reference draws from NumPy's global Mersenne Twister, which a device generator cannot
// reproduce: parity here is statistical (the constraints, the SNR), not the stream.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define GEN_MAX_TRIALS 65536
__global__ __launch_bounds__(WG) void generate_kernel(const double* __restrict__ bank64, const double* __restrict__ anorm2,
                                                      int K, int L, int N, int P, const int32_t* __restrict__ sparsity,
                                                      const double* __restrict__ snr_db, int kmax, double pos_thr,
                                                      double corr_thr, uint64_t seed, uint64_t first_index,
                                                      double* __restrict__ signals, int32_t* __restrict__ truth,
                                                      int32_t* __restrict__ fail) {
    __shared__ int kept_pos[KMAX], kept_atom[KMAX];
    __shared__ int verdict;
    __shared__ double red[3][WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t s = blockIdx.x;
    const uint64_t gid = first_index + s;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int want = min(max(sparsity[s], 0), min(kmax, KMAX));
    int have = 0;
    if (tid == 0) verdict = 0;
    __syncthreads();
    for (uint32_t trial = 0; have < want && trial < GEN_MAX_TRIALS; ++trial) {
        uint32_t r[4];
        philox4x32(trial, (uint32_t)gid, (uint32_t)(gid >> 32), 0u, k0, k1, r);          // stream 0: placements
        const int atom = (int)(((uint64_t)r[0] * (uint64_t)K) >> 32), pos = (int)(((uint64_t)r[1] * (uint64_t)P) >> 32);
        bool near = false;
        for (int q = 0; q < have; ++q) near |= fabs((double)(pos - kept_pos[q])) < pos_thr;
        if (near) continue;                                            // uniform: every thread drew the same numbers
        for (int q = wave; q < have; q += WG / 64) {
            const int ka = kept_atom[q], dp = pos - kept_pos[q];      // a_q index = i + dp
            double v = dot_range<64>(bank64 + (size_t)atom * L, bank64 + (size_t)ka * L + dp, max(0, -dp), min(L, L - dp), lane);
            v = wave_sum(v);
            if (lane == 63 && v / sqrt(anorm2[atom] * anorm2[ka]) > corr_thr) atomicOr(&verdict, 1);
        }
        __syncthreads();
        const bool rejected = verdict != 0;
        __syncthreads();
        if (tid == 0) {
            verdict = 0;
            if (!rejected) { kept_pos[have] = pos; kept_atom[have] = atom; }
        }
        __syncthreads();
        if (!rejected) ++have;
    }
    if (tid == 0 && fail) fail[s] = have < want ? 1 : 0;
    if (tid < kmax) {
        truth[(s * kmax + tid) * 2] = tid < have ? kept_pos[tid] : -1;
        truth[(s * kmax + tid) * 2 + 1] = tid < have ? kept_atom[tid] : -1;
    }
    // clean signal, statistics of its active samples
    double cnt = 0.0, sum = 0.0, sq = 0.0;
    for (int n = tid; n < N; n += WG) {
        double c = 0.0;
        for (int q = 0; q < have; ++q) {
            const int o = n - kept_pos[q];
            if ((unsigned)o < (unsigned)L) c += bank64[(size_t)kept_atom[q] * L + o];
        }
        if (fabs(c) > 1e-3) { cnt += 1.0; sum += c; sq = fma(c, c, sq); }
    }
    cnt = wave_sum(cnt); sum = wave_sum(sum); sq = wave_sum(sq);
    if (lane == 63) { red[0][wave] = cnt; red[1][wave] = sum; red[2][wave] = sq; }
    __syncthreads();
    double tc = 0.0, ts = 0.0, tq = 0.0;
    for (int w = 0; w < WG / 64; ++w) { tc += red[0][w]; ts += red[1][w]; tq += red[2][w]; }
    const double mean = tc > 0.0 ? ts / tc : 0.0;
    const double var = tc > 0.0 ? fmax(tq / tc - mean * mean, 0.0) : 0.0;             // np.var: population variance
    const double sigma = sqrt(var / pow(10.0, snr_db[s] / 10.0));
    for (int n = tid; n < N; n += WG) {
        double c = 0.0;
        for (int q = 0; q < have; ++q) {
            const int o = n - kept_pos[q];
            if ((unsigned)o < (unsigned)L) c += bank64[(size_t)kept_atom[q] * L + o];
        }
        uint32_t r[4];
        philox4x32((uint32_t)n, (uint32_t)gid, (uint32_t)(gid >> 32), 1u, k0, k1, r);      // stream 1: noise
        const double u1 = ((double)r[0] * 4294967296.0 + (double)r[1] + 0.5) * 5.421010862427522e-20;   // (0, 1)
        const double u2 = ((double)r[2] * 4294967296.0 + (double)r[3] + 0.5) * 5.421010862427522e-20;
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        signals[s * (size_t)N + n] = c + sigma * z;
    }
}

// ------------------------------------------------------------------------------------------------
// per-sparsity winners (src/csc/mmp.py:577-606 '<=' = last minimum; :464 np.argmin = first)
// ------------------------------------------------------------------------------------------------
__global__ void reduce_kernel(int S, int Kmax, int NN, TreeShape ts, const int32_t* __restrict__ node_pos,
                              const int32_t* __restrict__ node_atom, const double* __restrict__ node_mse,
                              double* __restrict__ rec_mse, int32_t* __restrict__ rec_leaf,
                              int32_t* __restrict__ rec_sel, int aos) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)S * Kmax) return;
    const size_t s = i / Kmax;
    const int k = (int)(i % Kmax) + 1;
    if (aos) {
        // CSC_FLAG_RECORDS_AOS: one contiguous record per signal, K f64 mse | K x 2 i32 leaf | K x K x 2 i32 (pos, atom);
        // rec_mse is the base of the [S][(16 + 8 K) K] byte buffer (rec_leaf / rec_sel are ignored)
        char* rec = reinterpret_cast<char*>(rec_mse) + s * (size_t)(16 + 8 * Kmax) * Kmax;
        const size_t back = s * Kmax;                                 // the dense indexing below is per batch
        rec_mse = reinterpret_cast<double*>(rec) - back;
        rec_leaf = reinterpret_cast<int32_t*>(rec + 8 * Kmax) - 2 * back;
        rec_sel = reinterpret_cast<int32_t*>(rec + 16 * Kmax) - back * Kmax * 2;
    }
    const double* mse = node_mse + s * NN + ts.off[k];
    double m = INFINITY;
    int last = 0, first = 0, first_nan = -1;
    for (int j = 0; j < ts.F[k]; ++j) {
        const double v = mse[j];
        if (v != v && first_nan < 0) first_nan = j;
        if (v <= m) { if (v < m) first = j; m = v; last = j; }
    }
    // NaN rule.  First minimum: np.argmin (src/csc/mmp.py:464) returns the first NaN when there is one.  Last
    // minimum: the reference's '<=' loop (:583-587) never fires on a NaN, so NaNs are skipped; on a NaN-only level
    // the reference raises (argmin_leaf stays None) -- here that level reports leaf 0 with its NaN MSE.
    if (first_nan >= 0) first = first_nan;
    rec_mse[i] = node_mse[s * NN + ts.off[k] + last];
    rec_leaf[2 * i] = last;
    rec_leaf[2 * i + 1] = first;
    int32_t* sel = rec_sel + i * Kmax * 2;
    for (int e = 1; e <= Kmax; ++e) {
        if (e <= k) {
            const size_t id = s * NN + ts.off[e] + (last % ts.pw[e]);
            sel[2 * (e - 1)] = node_pos[id];
            sel[2 * (e - 1) + 1] = node_atom[id];
        } else {
            sel[2 * (e - 1)] = -1;
            sel[2 * (e - 1) + 1] = -1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// hipGetLastError() is sticky per host thread: drop whatever an earlier, unrelated HIP call of the
// process left behind so a launch is judged by its own status only.
static inline void clear_stale_error() { (void)hipGetLastError(); }
static int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL((int)e, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

static int ceil4(int v) { return (v + 3) & ~3; }
static int ceil8(int v) { return (v + 7) & ~7; }
static int n_tiles(int K) { return (K + CSC_TILE_COLS - 1) / CSC_TILE_COLS; }
static int n_slabs(int L) { return (L + SLAB - 1) / SLAB; }
static size_t tile_floats(int K, int L) { return (size_t)n_tiles(K) * n_slabs(L) * SLABF; }
// packed bank buffer: [fp32 tiles][meta per tile][fp16 tiles][amax2 + spare (16 floats)][anorm2: K doubles, padded to an
// even count][scaled fp16 tiles hi][scaled fp16 tiles lo]
static size_t bank16_offset(int K, int L) { return tile_floats(K, L) + (size_t)n_tiles(K) * META_INTS; }
static const _Float16* bank_tiles16(const float* d_bank32t, int K, int L) {
    return reinterpret_cast<const _Float16*>(d_bank32t + bank16_offset(K, L));
}
static const float* bank_amax2(const float* d_bank32t, int K, int L) {
    return d_bank32t + bank16_offset(K, L) + tile_floats(K, L) / 2;
}
static const double* bank_anorm2(const float* d_bank32t, int K, int L) {
    return reinterpret_cast<const double*>(bank_amax2(d_bank32t, K, L) + 16);
}
// behind anorm2: the scaled two-term fp16 tiles (hi, lo) of the tiled kernel's filter
static const _Float16* bank_tiles16s(const float* d_bank32t, int K, int L, int lo) {
    const double* end = bank_anorm2(d_bank32t, K, L) + K + (K & 1);          // 16-byte aligned
    return reinterpret_cast<const _Float16*>(end) + (lo ? tile_floats(K, L) : 0);
}
// behind the scaled tiles, for dictionaries of more than one bank tile: the low-rank section (csc_bank_lowrank):
// [basis x 64 as fp16 B fragments, hi | lo: 2 * n_slabs * 4 * 512 halves][V16: n_tiles * 9 * (LR_R / 32) * 512 halves][slack: n_tiles * 144 floats]
static size_t bank_base_elems(int K, int L) {
    return bank16_offset(K, L) + tile_floats(K, L) / 2 + 16 + 2 * (size_t)(K + (K & 1)) + tile_floats(K, L);
}
static size_t bank_lr_elems(int K, int L) {
    if (K <= CSC_TILE_COLS) return 0;
    return (size_t)n_slabs(L) * (LR_R / 16) * 512 + (size_t)n_tiles(K) * NT * (LR_R / 32) * 256 + (size_t)n_tiles(K) * CSC_TILE_COLS;
}
// behind it: the atom summary of a similarity table (csc_bank_simatoms): K rows of sa_words(K) words
static int sa_words(int K) { return (K + 31) / 32; }
static size_t bank_sa_elems(int K) { return K <= CSC_TILE_COLS ? 0 : (size_t)K * sa_words(K) + 2 * (size_t)K; }   // + the shift ranges
static uint32_t* bank_simatoms(const float* d_bank32t, int K, int L) {
    return reinterpret_cast<uint32_t*>(const_cast<float*>(d_bank32t) + bank_base_elems(K, L) + bank_lr_elems(K, L));
}
static float* bank_lr_u32(const float* d_bank32t, int K, int L) { return const_cast<float*>(d_bank32t) + bank_base_elems(K, L); }
static size_t bank_lr_u_floats(int L) { return (size_t)n_slabs(L) * (LR_R / 16) * 512; }   // hi + lo fragments, as floats
static _Float16* bank_lr_v16(const float* d_bank32t, int K, int L) {
    return reinterpret_cast<_Float16*>(bank_lr_u32(d_bank32t, K, L) + bank_lr_u_floats(L));
}
static float* bank_lr_slack(const float* d_bank32t, int K, int L) {
    return bank_lr_u32(d_bank32t, K, L) + bank_lr_u_floats(L) + (size_t)n_tiles(K) * NT * (LR_R / 32) * 256;
}
static const int32_t* bank_meta(const float* d_bank32t, int K, int L) {
    return reinterpret_cast<const int32_t*>(d_bank32t + tile_floats(K, L));
}

static int make_shape(int Kmax, int c, int B, TreeShape* ts) {
    if (Kmax < 1 || Kmax > KMAX) FAIL(-2, "max_sparsity %d outside [1, %d]", Kmax, KMAX);
    if (c < 1 || c > CSC_MAX_CONNECTIONS) FAIL(-3, "connections %d outside [1, %d]", c, CSC_MAX_CONNECTIONS);
    if (B < 1) FAIL(-4, "branches must be > 0 (got %d)", B);   // src/csc/mmp.py:409
    long long pw = 1;
    int off = 0;
    ts->F[0] = 1; ts->pw[0] = 1; ts->off[0] = 0;
    for (int d = 1; d <= Kmax; ++d) {
        pw = (pw * c > (1LL << 30)) ? (1LL << 30) : pw * c;
        ts->pw[d] = (int)pw;
        ts->F[d] = (int)(pw < (long long)B ? pw : (long long)B);
        ts->off[d] = off;
        off += ts->F[d];
        if (off > (1 << 24)) FAIL(-5, "tree too large (%d nodes)", off);
    }
    ts->off[Kmax + 1] = off;
    for (int d = Kmax + 1; d <= KMAX; ++d) { ts->F[d] = 0; ts->pw[d] = 1; }
    return off;
}

static size_t level_lds_bytes(int N, int Lp) {
    const int Npad = ceil4(N);
    const int R32 = CSC_TILE_ROWS + Lp + 4;
    return (size_t)Npad * 8 + sizeof(NodeLds) + (size_t)ceil4(R32) * 4 + 2 * SLABF * 4 + R16_COPIES * R16_CS * 2;
}

// ------------------------------------------------------------------------------------------------
// optional per-launch HIP-event timer (bench.py's live roofline measurement)
// ------------------------------------------------------------------------------------------------
struct csc_timer {
    int capacity, count;
    hipEvent_t* ev;      // [2 * capacity]
    int* tag;            // [capacity]
};
static thread_local csc_timer* g_timer = nullptr;

static inline int timer_begin(hipStream_t st, int tag) {
    csc_timer* t = g_timer;
    if (!t || t->count >= t->capacity) return -1;
    const int i = t->count++;
    t->tag[i] = tag;
    hipEventRecord(t->ev[2 * i], st);
    return i;
}
static inline void timer_end(hipStream_t st, int i) {
    if (i >= 0) hipEventRecord(g_timer->ev[2 * i + 1], st);
}

extern "C" {

int csc_abi_version(void) { return CSC_ABI_VERSION; }
const char* csc_last_error(void) { return g_err; }

size_t csc_bank32_elems(int K, int L) {
    if (K <= 0 || L <= 0) return 0;
    return bank_base_elems(K, L) + bank_lr_elems(K, L) + bank_sa_elems(K);
}

int csc_lowrank_rank(void) { return LR_R; }

int csc_bank_simatoms(float* d_bank32t, int K, int L, const uint32_t* d_sim_bits, int W, void* stream) {
    if (!d_bank32t || !d_sim_bits || K <= 0 || L <= 0 || W <= 0) FAIL(-1, "csc_bank_simatoms: bad arguments");
    if (K <= CSC_TILE_COLS) FAIL(-13, "csc_bank_simatoms: dictionaries of one bank tile (K <= %d) have no low-rank screen", CSC_TILE_COLS);
    if (K > 65535) FAIL(-11, "csc_bank_simatoms: dictionary too large (%d atoms)", K);
    hipStream_t st = (hipStream_t)stream;
    clear_stale_error();
    uint32_t* out = bank_simatoms(d_bank32t, K, L);
    int* range = reinterpret_cast<int*>(out + (size_t)K * sa_words(K));
    hipError_t e = hipMemsetAsync(range, 0x7f, (size_t)K * 4, st);                 // (0x7f7f7f7f: above any bit index)
    if (e == hipSuccess) e = hipMemsetAsync(range + K, 0xff, (size_t)K * 4, st);    // -1
    if (e != hipSuccess) FAIL((int)e, "csc_bank_simatoms: hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(bank_simatoms_kernel, dim3((unsigned)((K + WG - 1) / WG), (unsigned)K), dim3(WG), 0, st, d_sim_bits, W, K,
                       sa_words(K), out, range);
    return launch_status("bank_simatoms_kernel");
}

int csc_bank_lowrank(float* d_bank32t, const double* d_bank64, int K, int L, const double* d_basis, int rank, void* stream) {
    if (!d_bank32t || !d_bank64 || !d_basis || K <= 0 || L <= 0) FAIL(-1, "csc_bank_lowrank: bad arguments");
    if (rank != LR_R) FAIL(-13, "csc_bank_lowrank: the screen takes a basis of %d vectors, not %d", LR_R, rank);
    if (K <= CSC_TILE_COLS) FAIL(-13, "csc_bank_lowrank: dictionaries of one bank tile (K <= %d) have no low-rank screen", CSC_TILE_COLS);
    if (n_slabs(L) > MAX_SLABS) FAIL(-8, "atom length %d exceeds %d", L, MAX_SLABS * SLAB);
    hipStream_t st = (hipStream_t)stream;
    const int nslab = n_slabs(L), ntile = n_tiles(K);
    clear_stale_error();
    const size_t nu = (size_t)nslab * (LR_R / 16) * 512;
    hipLaunchKernelGGL(lr_pack_u16_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, d_basis, L, nslab,
                       reinterpret_cast<_Float16*>(bank_lr_u32(d_bank32t, K, L)));
    const size_t nv = (size_t)ntile * NT * (LR_R / 32) * 512;
    hipLaunchKernelGGL(lr_pack_v_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, d_basis, d_bank64, K, L,
                       bank_meta(d_bank32t, K, L), bank_amax2(d_bank32t, K, L), bank_lr_v16(d_bank32t, K, L), nv);
    hipLaunchKernelGGL(lr_slack_kernel, dim3((unsigned)(ntile * CSC_TILE_COLS)), dim3(64), 0, st, d_basis, d_bank64, K, L,
                       bank_meta(d_bank32t, K, L), bank_amax2(d_bank32t, K, L), bank_lr_v16(d_bank32t, K, L),
                       bank_lr_slack(d_bank32t, K, L));
    return launch_status("csc_bank_lowrank kernels");
}

int csc_sim_words(int N, int L) {
    const int P = N - L + 1;
    if (P <= 0) return 0;
    return 2 * ((2 * P - 1 + 63) / 64);
}

static int scratch_ptr(void** out);                                 // csc_static_scratch.inc

int csc_bank_pack(const double* d_bank64, int K, int L, float* d_bank32t, void* stream) {
    if (!d_bank64 || !d_bank32t || K <= 0 || L <= 0) FAIL(-1, "csc_bank_pack: bad arguments");
    {   // this is the library's set-up call on a device: look the static scratch of csc_mmpdf_batch_static up now, so
        // that the entry point itself never has to (it may run under stream capture)
        void* p = nullptr;
        const int rc = scratch_ptr(&p);
        if (rc) return rc;
    }
    if (n_slabs(L) > MAX_SLABS) FAIL(-8, "atom length %d exceeds %d", L, MAX_SLABS * SLAB);
    const size_t total = tile_floats(K, L);
    const int ntile = n_tiles(K), nslab = n_slabs(L), npad = ntile * CSC_TILE_COLS - K;
    int32_t* meta = reinterpret_cast<int32_t*>(d_bank32t + total);
    int32_t* keys = reinterpret_cast<int32_t*>(d_bank32t);           // parked in the data region
    hipStream_t st = (hipStream_t)stream;
    clear_stale_error();
    hipLaunchKernelGGL(bank_keys_kernel, dim3((K + 255) / 256), dim3(256), 0, st, d_bank64, K, L, keys);
    hipLaunchKernelGGL(bank_rank_kernel, dim3((K + 255) / 256), dim3(256), 0, st, keys, K, npad, meta);
    hipLaunchKernelGGL(bank_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_bank64, K, L,
                       nslab, meta, d_bank32t, total);
    hipLaunchKernelGGL(bank_mask_kernel, dim3(MAX_SLABS, ntile), dim3(256), 0, st, d_bank32t, nslab, meta);
    hipLaunchKernelGGL(bank_pack16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_bank64, K, L,
                       nslab, meta, const_cast<_Float16*>(bank_tiles16(d_bank32t, K, L)), total);
    float* amax2 = const_cast<float*>(bank_amax2(d_bank32t, K, L));
    if (hipMemsetAsync(amax2, 0, 16 * sizeof(float), st) != hipSuccess) FAIL(-1, "csc_bank_pack: hipMemsetAsync failed");
    if ((K & 1) && hipMemsetAsync(const_cast<double*>(bank_anorm2(d_bank32t, K, L)) + K, 0, sizeof(double), st) != hipSuccess)
        FAIL(-1, "csc_bank_pack: hipMemsetAsync failed");                // the pad behind anorm2: every byte of the image is defined
    hipLaunchKernelGGL(bank_norm_kernel, dim3(K), dim3(64), 0, st, d_bank64, K, L, amax2,
                       const_cast<double*>(bank_anorm2(d_bank32t, K, L)));
    hipLaunchKernelGGL(bank_pack16s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_bank64, K, L,
                       nslab, meta, amax2, const_cast<_Float16*>(bank_tiles16s(d_bank32t, K, L, 0)),
                       const_cast<_Float16*>(bank_tiles16s(d_bank32t, K, L, 1)), total);
    return launch_status("bank_pack kernels");
}

int csc_similarity_table(const double* d_bank64, int K, int L, int N, double threshold,
                         uint32_t* d_sim_bits, void* stream) {
    const int P = N - L + 1;
    if (!d_bank64 || !d_sim_bits || K <= 0 || L <= 0) FAIL(-1, "csc_similarity_table: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if (K > 65535) FAIL(-7, "K = %d too large for the similarity table launch", K);
    const int W = csc_sim_words(N, L);
    clear_stale_error();
    hipLaunchKernelGGL(sim_table_kernel, dim3(K, K), dim3(WG), (size_t)2 * L * sizeof(double),
                       (hipStream_t)stream, d_bank64, K, L, P, W, threshold, d_sim_bits);
    return launch_status("sim_table_kernel");
}

int csc_similarity_table_packed(const float* d_bank32t, const double* d_bank64, int K, int L, int N, double threshold,
                                uint32_t* d_sim_bits, void* stream) {
    const int P = N - L + 1;
    if (!d_bank32t || !d_bank64 || !d_sim_bits || K <= 0 || L <= 0) FAIL(-1, "csc_similarity_table_packed: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    const int W = csc_sim_words(N, L);
    const int Lp = n_slabs(L) * SLAB;
    const int nrb = (2 * P - 1 + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    const int R32 = CSC_TILE_ROWS * nrb + Lp + 4;
    const size_t lds = (size_t)2 * SLABF * 4 + (size_t)ceil4(R32) * 4 + (size_t)CSC_TILE_COLS * W * 4 + SIMQ * 4;
    // long atoms or very long signals (the shift index travels in 16 bits): the float64 kernel (same bits)
    if (n_slabs(L) > MAX_SLABS || lds > 72 * 1024 || 2 * P - 1 > 65535)
        return csc_similarity_table(d_bank64, K, L, N, threshold, d_sim_bits, stream);
    clear_stale_error();
    hipFuncSetAttribute((const void*)sim_table_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sim_table_mfma_kernel, dim3(K), dim3(WG), lds, (hipStream_t)stream, d_bank32t, bank_meta(d_bank32t, K, L),
                       bank_amax2(d_bank32t, K, L), d_bank64, K, L, Lp, P, W, threshold, d_sim_bits);
    return launch_status("sim_table_mfma_kernel");
}

int csc_correlate(const float* d_bank32t, int K, int L, const double* d_signals, int B, int N,
                  float* d_corr, void* stream) {
    const int P = N - L + 1;
    if (!d_bank32t || !d_signals || !d_corr || K <= 0 || L <= 0 || B < 0) FAIL(-1, "csc_correlate: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if (n_slabs(L) > MAX_SLABS) FAIL(-8, "atom length %d exceeds %d", L, MAX_SLABS * SLAB);
    if (B == 0) return 0;
    const int Lp = n_slabs(L) * SLAB;                      // residual window zero-padded to whole slabs
    const int R32 = CSC_TILE_ROWS * ((P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS) + Lp + 4;   // Lp = whole slabs
    const size_t lds = (size_t)ceil4(R32) * 4 + 2 * SLABF * 4;
    if (lds > 150 * 1024) FAIL(-8, "signal too long for the LDS-resident correlation (%zu B)", lds);
    clear_stale_error();
    hipFuncSetAttribute((const void*)correlate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(correlate_kernel, dim3(B), dim3(WG), lds, (hipStream_t)stream, d_bank32t,
                       bank_meta(d_bank32t, K, L), K, L, Lp, d_signals, N, P, d_corr);
    return launch_status("correlate_kernel");
}

int csc_correlate64(const double* d_bank64, int K, int L, const double* d_signals, int B, int N, double* d_corr,
                    void* stream) {
    const int P = N - L + 1;
    if (!d_bank64 || !d_signals || !d_corr || K <= 0 || L <= 0 || B < 0) FAIL(-1, "csc_correlate64: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if ((size_t)L * 8 > 64 * 1024) FAIL(-8, "atom length %d exceeds the LDS window of csc_correlate64", L);
    if (B > 65535) FAIL(-11, "csc_correlate64: at most 65535 signals per call");
    if (B == 0) return 0;
    clear_stale_error();
    hipLaunchKernelGGL(correlate64_kernel, dim3(P, B), dim3(WG), (size_t)L * 8, (hipStream_t)stream, d_bank64, K, L,
                       d_signals, N, P, d_corr);
    return launch_status("correlate64_kernel");
}

int csc_tree_nodes(int max_sparsity, int connections, int branches, int* level_sizes) {
    TreeShape ts;
    const int nn = make_shape(max_sparsity, connections, branches, &ts);
    if (nn < 0) return nn;
    if (level_sizes) for (int d = 1; d <= max_sparsity; ++d) level_sizes[d - 1] = ts.F[d];
    return nn;
}

static int frontier_max(int max_sparsity, const TreeShape& ts) {
    int fmax = 1;
    for (int d = 1; d < max_sparsity; ++d) fmax = ts.F[d] > fmax ? ts.F[d] : fmax;
    return fmax;
}
static size_t node_ws_bytes(int S, int N, int max_sparsity, const TreeShape& ts) {
    return 2 * (size_t)S * frontier_max(max_sparsity, ts) * (N + TRI(max_sparsity) + max_sparsity) * sizeof(double);
}
static size_t flag_ws_bytes(int S, int max_sparsity, const TreeShape& ts) {      // one byte per frontier node
    return ((size_t)S * frontier_max(max_sparsity, ts) + 15) & ~(size_t)15;
}
static size_t lean_lds_bytes(int N) {
    return (size_t)ceil4(N) * 8 + R16_COPIES * R16_CS * 2 + sizeof(NodeLds) + sizeof(ChainLds);
}

size_t csc_mmpdf_workspace_bytes(int S, int N, int max_sparsity, int connections, int branches) {
    TreeShape ts;
    if (make_shape(max_sparsity, connections, branches, &ts) < 0 || S <= 0 || N <= 0) return 0;
    return node_ws_bytes(S, N, max_sparsity, ts) + flag_ws_bytes(S, max_sparsity, ts);
}

static int run_levels(int mode, const float* d_bank32t, const double* d_bank64, const uint32_t* d_sim_bits,
                      int K, int L, const double* d_signals, int S, int N,
                      int max_sparsity, int connections, int branches, int flags,
                      void* d_workspace, size_t workspace_bytes,
                      int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse,
                      double* d_node_coef, double* d_rec_mse, int32_t* d_rec_leaf,
                      int32_t* d_rec_sel, int32_t* d_stats, void* stream) {
    TreeShape ts;
    const int NN = make_shape(max_sparsity, connections, branches, &ts);
    if (NN < 0) return NN;
    const int P = N - L + 1;
    if (S == 0) return 0;
    if (!d_bank32t || !d_bank64 || !d_signals || !d_node_pos || !d_node_atom || !d_node_mse)
        FAIL(-1, "csc_mmpdf_batch / csc_mp_batch: null pointer");
    if (connections > 1 && !d_sim_bits) FAIL(-1, "csc_mmpdf_batch: similarity table required when connections > 1");
    if (K <= 0 || L <= 0 || S < 0) FAIL(-1, "csc_mmpdf_batch: bad sizes");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    const bool tiled = (K > CSC_TILE_COLS || P > CSC_TILE_ROWS);      // correlation matrix larger than one register tile
    const int nrb = (P + CSC_TILE_ROWS - 1) / CSC_TILE_ROWS;
    if (tiled && (long long)n_tiles(K) * nrb > MAX_UNITS)
        FAIL(-9, "dictionary (K = %d, P = %d) exceeds %d (256 x 144) units per node", K, P, MAX_UNITS);
    if (n_slabs(L) > MAX_SLABS) FAIL(-8, "atom length %d exceeds %d", L, MAX_SLABS * SLAB);
    if (max_sparsity > P * K) FAIL(-2, "max_sparsity exceeds the number of activations");
    const size_t node_bytes = node_ws_bytes(S, N, max_sparsity, ts);
    const size_t need = node_bytes + flag_ws_bytes(S, max_sparsity, ts);
    uint8_t* node_flags = max_sparsity > 1 ? reinterpret_cast<uint8_t*>(static_cast<char*>(d_workspace) + node_bytes) : nullptr;
    // the lean kernel (fp16 filter, four nodes per CU) takes the level first; needs the flag buffer (max_sparsity = 1
    // passes no workspace), the canonical window length and a tree level that fits the flag buffer
    const bool lean = !tiled && !(flags & CSC_FLAG_FP32_ONLY) && node_flags && N <= 4 * WG && n_slabs(L) * SLAB <= 512 &&
                      !(flags & CSC_FLAG_NO_LEAN);
    if (max_sparsity > 1 && (workspace_bytes < need || !d_workspace))
        FAIL(-10, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if (((uintptr_t)d_workspace & 15) != 0) FAIL(-10, "workspace must be 16-byte aligned");
    const int Lp = n_slabs(L) * SLAB;                      // residual window zero-padded to whole slabs
    const int W = csc_sim_words(N, L);
    size_t lds = level_lds_bytes(N, Lp);
    if (tiled) lds += (size_t)(nrb - 1) * CSC_TILE_ROWS * 4 + sizeof(TiledLds);
    if (lds > 150 * 1024) FAIL(-8, "signal too long for the LDS-resident node kernel (%zu B)", lds);
    if (!tiled && (size_t)K * (W + 1) * 4 > 2 * SLABF * 4) FAIL(-8, "similarity row does not fit the slab buffer");
    // dictionaries beyond one register tile: the fp16 three-product filter in front of the fp32 kernel (needs the flag
    // buffer; its accumulation bound 3 L 2^-24 <= 2^-13 holds up to L = 682).  One row block (P <= 272) takes 75 KB of
    // LDS: two workgroups per CU; the window copies of a second and third row block bring it to 85 - 95 KB and one
    // workgroup per CU, which is still 5 - 6x the fp32 kernel
    const size_t lds16 = (size_t)ceil4(N) * 8 + 2 * (size_t)R16_COPIES * r16t_stride(P, Lp) * 2 + (size_t)TRING * TG_BLK * 2 +
                         sizeof(NodeLds) + sizeof(Tiled16Lds);
    // its screen is a kernel of its own (three workgroups per CU); the bounds travel in a workspace row per node
    const int nrb16 = (nrb > 1 && P - (nrb - 1) * CSC_TILE_ROWS <= 16) ? nrb - 1 : nrb;
    const size_t units16 = (size_t)n_tiles(K) * nrb16 * NG;
    const size_t lds_screen = 4 * 8 + (size_t)R16_COPIES * r16t_stride(P, Lp) * 2 + (size_t)TRING * TG_BLK * 2 + units16 * 4 +
                              (((size_t)n_tiles(K) * NT * 2 + 15) & ~(size_t)15) + 2 * CSC_MAX_CONNECTIONS * 4 +
                              (((size_t)n_tiles(K) * NT + 15) & ~(size_t)15);
    const size_t lds_lr = 32 + 2 * (size_t)R16_COPIES * r16t_stride(P, Lp) * 2 + (size_t)LR_PLANE_H * 2 + ((size_t)n_tiles(K) * NT + 1) * 4 +
                          5 * CSC_MAX_CONNECTIONS * 4;
    const size_t lds_lr_supp = lds_lr + 2 * (size_t)LR_CHUNK_C * 4;     // later siblings: + the chunk's similarity bands
    const bool tiled16 = tiled && !(flags & CSC_FLAG_FP32_ONLY) && node_flags && mode == 0 && L <= 682 &&
                         lds16 <= 150 * 1024 && (long long)n_tiles(K) * nrb * NG <= MAX_UNITS16 && n_tiles(K) * NT <= 2 * MAX_UNITS16 &&
                         lds_screen <= 100 * 1024 && (units16 + 1) * 4 <= (size_t)(N + TRI(max_sparsity) + max_sparsity) * 8;
    const bool lr_table_fits = ((size_t)n_tiles(K) * NT + 1) * 4 <= (size_t)(N + TRI(max_sparsity) + max_sparsity) * 8;
    clear_stale_error();
    // the low-rank screen in front (the caller vouches for an installed section: csc_bank_lowrank): one row block only
    const bool lowrank = tiled16 && (flags & CSC_FLAG_LOWRANK) && P >= CSC_TILE_ROWS && P <= 272 && lds_lr_supp <= 100 * 1024 && lr_table_fits;
    // ... and the atom summary of d_sim_bits (csc_bank_simatoms, vouched for likewise): the later siblings' low-rank product
    // looks up similarity rows only for the columns the summary names
    const bool simatoms = lowrank && (flags & CSC_FLAG_SIMATOMS) && d_sim_bits;
    if (tiled16) {
        hipFuncSetAttribute((const void*)mmp_level_tiled16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipFuncSetAttribute((const void*)screen16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_screen);
        if (lowrank) {
            hipFuncSetAttribute((const void*)screen_lr_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lr);
            hipFuncSetAttribute((const void*)screen_lr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lr_supp);
        }
    }
    hipFuncSetAttribute(tiled ? (const void*)mmp_level_tiled_kernel : (const void*)mmp_level_kernel,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (!tiled) hipFuncSetAttribute((const void*)mmp_level_followup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds);

    double* ws[2] = {static_cast<double*>(d_workspace), static_cast<double*>(d_workspace) + node_bytes / 16};
    hipStream_t st = (hipStream_t)stream;
    bool picked = false;                                              // level 0 wrote picks only
    for (int d = 0; d < max_sparsity; ++d) {
        LevelArgs a;
        a.bank32t = d_bank32t; a.meta = bank_meta(d_bank32t, K, L); a.bank64 = d_bank64; a.sim_bits = d_sim_bits;
        a.K = K; a.L = L; a.Lp = Lp; a.N = N; a.P = P; a.W = W;
        a.depth = d; a.Kmax = max_sparsity; a.connections = connections; a.NN = NN;
        a.NS = N + TRI(max_sparsity) + max_sparsity; a.mode = mode;
        // the fp16 filter's LDS copies hold a window of R16_CS = 256 + 512 + ... elements: atoms of up to 512 samples
        // (whole slabs); longer atoms correlate in fp32
        a.bank16 = (!tiled && !(flags & CSC_FLAG_FP32_ONLY) && Lp <= 512) ? bank_tiles16(d_bank32t, K, L) : nullptr;
        a.amax2 = bank_amax2(d_bank32t, K, L);
        a.anorm2 = bank_anorm2(d_bank32t, K, L);
        a.resid_in = (d == 0) ? d_signals : ws[d & 1];
        a.resid_out = (d + 1 < max_sparsity) ? ws[(d + 1) & 1] : nullptr;
        a.node_pos = d_node_pos; a.node_atom = d_node_atom; a.node_mse = d_node_mse; a.node_coef = d_node_coef;
        a.stats = d_stats; a.ts = ts;
        const size_t grid = (size_t)S * ts.F[d];
        if (grid > 0x7fffffffULL) FAIL(-11, "batch too large for one launch (%zu workgroups)", grid);
        const int span = timer_begin(st, d);
        a.flags = nullptr; a.S = S; a.signals = d_signals; a.ws[0] = ws[0]; a.ws[1] = ws[1];
        a.picks_only = 0; a.from_root = 0; a.sibling = -1;
        a.lr_u16 = nullptr; a.lr_v16 = nullptr; a.lr_slack = nullptr; a.simatoms = nullptr; a.sa_w = 0; a.simrange = nullptr;
        // level 0 in front of the chain kernel (every node of level 1 has one child): it only has to pick -- the chain
        // kernel builds a pick's node from the signal, so no child residual crosses HBM between the two launches
        const bool root_picks = lean && mode == 0 && d == 0 && max_sparsity > 1 && connections > 1 && ts.pw[1] >= ts.F[2];
        bool chain = false;
        if (tiled16) {
            // the three-product fp16 filter takes the level; the fp32 kernel behind it redoes the flagged nodes
            a.flags = node_flags;
            a.bank16 = bank_tiles16s(d_bank32t, K, L, 0);
            // one pair of launches per child: the screen bounds child t under the suppression by the children picked
            // before it, which the launches before have written to the node tables
            int nsib = 0;
            for (int r = 0; r < connections; ++r)
                if ((long long)r * ts.pw[d] < ts.F[d + 1]) ++nsib;
            a.lr_u16 = lowrank ? reinterpret_cast<const _Float16*>(bank_lr_u32(d_bank32t, K, L)) : nullptr;
            a.lr_v16 = lowrank ? bank_lr_v16(d_bank32t, K, L) : nullptr;
            a.lr_slack = lowrank ? bank_lr_slack(d_bank32t, K, L) : nullptr;
            a.simatoms = simatoms ? bank_simatoms(d_bank32t, K, L) : nullptr;
            a.sa_w = sa_words(K);
            a.simrange = simatoms ? reinterpret_cast<const int*>(bank_simatoms(d_bank32t, K, L) + (size_t)K * sa_words(K)) : nullptr;
            for (int t = 0; t < nsib; ++t) {
                a.sibling = t;
                if (lowrank && t == 0) hipLaunchKernelGGL(screen_lr_kernel<false>, dim3((unsigned)grid), dim3(WG), lds_lr, st, a);
                else if (lowrank) hipLaunchKernelGGL(screen_lr_kernel<true>, dim3((unsigned)grid), dim3(WG), lds_lr_supp, st, a);

                hipLaunchKernelGGL(screen16_kernel, dim3((unsigned)grid), dim3(WG), lds_screen, st, a);
                hipLaunchKernelGGL(mmp_level_tiled16_kernel, dim3((unsigned)grid), dim3(WG), lds16, st, a);
            }
            a.bank16 = nullptr;
            hipLaunchKernelGGL(mmp_level_tiled_kernel, dim3((unsigned)grid), dim3(WG), lds, st, a);
        } else if (tiled) {
            hipLaunchKernelGGL(mmp_level_tiled_kernel, dim3((unsigned)grid), dim3(WG), lds, st, a);
        } else if (root_picks) {
            a.picks_only = 1;
            a.resid_out = nullptr;
            hipLaunchKernelGGL(mmp_level_kernel, dim3((unsigned)grid), dim3(WG), lds, st, a);
            picked = true;
        } else if (lean && (connections == 1 || ts.pw[d] >= ts.F[d + 1])) {
            // every node of this level has one child (node j's second child j + pw[d] falls outside F[d + 1]), and
            // so has every node below (pw grows, F stays at `branches`): ONE launch walks each of these chains down
            // to its leaf, the residual never leaving LDS (batches too small to fill the chip gain as well: 170 signals
            // of config 2 take 187 instead of 231 us).  With siblings the lean kernel would repeat its sweeps per
            // sibling and the register-resident kernel wins
            a.flags = node_flags;
            a.from_root = picked ? 1 : 0;
            chain = true;
            hipLaunchKernelGGL(mmp_level_lean_kernel, dim3((unsigned)grid), dim3(WG), lean_lds_bytes(N), st, a);
            hipLaunchKernelGGL(mmp_level_followup_kernel, dim3((unsigned)((grid + FOLLOW_NODES - 1) / FOLLOW_NODES)),
                               dim3(WG), lds, st, a);                // the chains it gave up: fp32
        } else {
            hipLaunchKernelGGL(mmp_level_kernel, dim3((unsigned)grid), dim3(WG), lds, st, a);
        }
        timer_end(st, span);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) FAIL((int)e, "mmp_level_kernel launch failed at level %d: %s", d, hipGetErrorString(e));
        if (chain) break;                                             // levels d .. max_sparsity - 1 are done
    }
    if (d_rec_mse && ((d_rec_leaf && d_rec_sel) || (flags & CSC_FLAG_RECORDS_AOS))) {
        const size_t tot = (size_t)S * max_sparsity;
        hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, S, max_sparsity,
                           NN, ts, d_node_pos, d_node_atom, d_node_mse, d_rec_mse, d_rec_leaf, d_rec_sel,
                           (flags & CSC_FLAG_RECORDS_AOS) ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) FAIL((int)e, "reduce_kernel launch failed: %s", hipGetErrorString(e));
    }
    return 0;
}

int csc_mmpdf_batch(const float* d_bank32t, const double* d_bank64, const uint32_t* d_sim_bits,
                    int K, int L, const double* d_signals, int S, int N,
                    int max_sparsity, int connections, int branches, int flags,
                    void* d_workspace, size_t workspace_bytes,
                    int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse,
                    double* d_node_coef, double* d_rec_mse, int32_t* d_rec_leaf,
                    int32_t* d_rec_sel, int32_t* d_stats, void* stream) {
    return run_levels(0, d_bank32t, d_bank64, d_sim_bits, K, L, d_signals, S, N, max_sparsity, connections,
                      branches, flags, d_workspace, workspace_bytes, d_node_pos, d_node_atom, d_node_mse, d_node_coef,
                      d_rec_mse, d_rec_leaf, d_rec_sel, d_stats, stream);
}

int csc_filter_product(const float* d_bank32t, int K, int L, const double* d_signals, int B, int N,
                       int variant, float* d_prod, float* d_eps, double* d_norm, void* stream) {
    const int P = N - L + 1;
    if (!d_bank32t || !d_signals || !d_prod || !d_eps || !d_norm || K <= 0 || L <= 0 || B < 0)
        FAIL(-1, "csc_filter_product: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if (K > CSC_TILE_COLS || P > CSC_TILE_ROWS || n_slabs(L) * SLAB > 512 || N > 4 * WG)
        FAIL(-9, "csc_filter_product: the fp16 filter covers K <= %d, P <= %d, L <= 512, N <= %d", CSC_TILE_COLS,
             CSC_TILE_ROWS, 4 * WG);
    if (variant != 0 && variant != 1) FAIL(-1, "csc_filter_product: variant must be 0 or 1");
    if (B == 0) return 0;
    const size_t lds = (size_t)RING16 * SLABF * 2 + (size_t)ceil4(N) * 8 + R16_COPIES * R16_CS * 2;
    clear_stale_error();
    hipFuncSetAttribute((const void*)filter_product_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(filter_product_kernel, dim3(B), dim3(WG), lds, (hipStream_t)stream, bank_tiles16(d_bank32t, K, L),
                       bank_meta(d_bank32t, K, L), bank_amax2(d_bank32t, K, L), K, L, d_signals, N, P, variant, d_prod,
                       d_eps, d_norm);
    return launch_status("filter_product_kernel");
}

int csc_mp_batch(const float* d_bank32t, const double* d_bank64, int K, int L, const double* d_signals,
                 int S, int N, int iterations, void* d_workspace, size_t workspace_bytes,
                 int32_t* d_node_pos, int32_t* d_node_atom, double* d_node_mse, double* d_node_mask,
                 int32_t* d_stats, void* stream) {
    if (!d_node_mask && S > 0) FAIL(-1, "csc_mp_batch: d_node_mask is required");
    return run_levels(1, d_bank32t, d_bank64, nullptr, K, L, d_signals, S, N, iterations, 1, 1, 0, d_workspace,
                      workspace_bytes, d_node_pos, d_node_atom, d_node_mse, d_node_mask, nullptr, nullptr,
                      nullptr, d_stats, stream);
}

int csc_match_batch(const double* d_bank64, int K, int L, const int32_t* d_true, int kt,
                    const int32_t* d_approx, int ka, int M, double pos_err_threshold,
                    double corr_err_threshold, int32_t* d_tp, void* stream) {
    if (!d_bank64 || !d_true || !d_approx || !d_tp || K <= 0 || L <= 0 || M < 0)
        FAIL(-1, "csc_match_batch: bad arguments");
    if (kt < 1 || kt > MATCH_MAX || ka < 1 || ka > MATCH_MAX)
        FAIL(-2, "csc_match_batch: %d x %d atoms outside [1, %d]", kt, ka, MATCH_MAX);
    if (M == 0) return 0;
    clear_stale_error();
    hipLaunchKernelGGL(match_kernel, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, d_bank64, K, L, d_true, kt,
                       d_approx, ka, pos_err_threshold, corr_err_threshold, d_tp);
    return launch_status("match_kernel");
}

int csc_generate_signals(const float* d_bank32t, const double* d_bank64, int K, int L, int N, int S,
                         const int32_t* d_sparsity, const double* d_snr_db, int kmax, double pos_err_threshold,
                         double corr_err_threshold, uint64_t seed, uint64_t first_index, double* d_signals,
                         int32_t* d_truth, int32_t* d_fail, void* stream) {
    const int P = N - L + 1;
    if (!d_bank32t || !d_bank64 || !d_sparsity || !d_snr_db || !d_signals || !d_truth || K <= 0 || L <= 0 || S < 0)
        FAIL(-1, "csc_generate_signals: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if (kmax < 1 || kmax > KMAX) FAIL(-2, "kmax %d outside [1, %d]", kmax, KMAX);
    if (S == 0) return 0;
    clear_stale_error();
    hipLaunchKernelGGL(generate_kernel, dim3((unsigned)S), dim3(WG), 0, (hipStream_t)stream, d_bank64,
                       bank_anorm2(d_bank32t, K, L), K, L, N, P, d_sparsity, d_snr_db, kmax, pos_err_threshold,
                       corr_err_threshold, seed, first_index, d_signals, d_truth, d_fail);
    return launch_status("generate_kernel");
}

int csc_project_batch(const double* d_bank64, int K, int L, const double* d_signals, int S, int N,
                      const int32_t* d_sig_index, const int32_t* d_support, int M, int kmax,
                      double* d_coef, double* d_mse, double* d_approx, int32_t* d_stats, void* stream) {
    const int P = N - L + 1;
    if (!d_bank64 || !d_signals || !d_support || K <= 0 || L <= 0 || S <= 0 || M < 0)
        FAIL(-1, "csc_project_batch: bad arguments");
    if (P <= 0) FAIL(-6, "signal length %d shorter than atom length %d", N, L);
    if (kmax < 1 || kmax > KMAX) FAIL(-2, "support size %d outside [1, %d]", kmax, KMAX);
    if (M == 0) return 0;
    const size_t lds = (size_t)((N + 1) & ~1) * 8 + sizeof(ProjLds);
    if (lds > 64 * 1024) FAIL(-8, "signal too long for the LDS-resident projection (%zu B)", lds);
    clear_stale_error();
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)M), dim3(WG), lds, (hipStream_t)stream, d_bank64, K, L, P,
                       d_signals, N, d_sig_index, d_support, kmax, d_coef, d_mse, d_approx, d_stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL((int)e, "project_kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

csc_timer* csc_timer_create(int capacity) {
    if (capacity <= 0) return nullptr;
    csc_timer* t = new csc_timer;
    t->capacity = capacity; t->count = 0;
    t->ev = new hipEvent_t[2 * (size_t)capacity];
    t->tag = new int[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&t->ev[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) hipEventDestroy(t->ev[j]);
            delete[] t->ev; delete[] t->tag; delete t;
            return nullptr;
        }
    return t;
}

void csc_timer_destroy(csc_timer* t) {
    if (!t) return;
    if (g_timer == t) g_timer = nullptr;
    for (int i = 0; i < 2 * t->capacity; ++i) hipEventDestroy(t->ev[i]);
    delete[] t->ev; delete[] t->tag; delete t;
}

void csc_timer_attach(csc_timer* t) { g_timer = t; }
void csc_timer_reset(csc_timer* t) { if (t) t->count = 0; }
int csc_timer_count(const csc_timer* t) { return t ? t->count : 0; }

int csc_timer_read(csc_timer* t, float* ms, int* tag, int n) {
    if (!t || !ms || n < 0) FAIL(-1, "csc_timer_read: bad arguments");
    const int m = n < t->count ? n : t->count;
    for (int i = 0; i < m; ++i) {
        hipError_t e = hipEventElapsedTime(&ms[i], t->ev[2 * i], t->ev[2 * i + 1]);
        if (e != hipSuccess) FAIL((int)e, "csc_timer_read: span %d not complete: %s", i, hipGetErrorString(e));
        if (tag) tag[i] = t->tag[i];
    }
    return m;
}

#ifdef CSC_CAND_ORDER
// the dump of the chain kernel's candidate lists: rows [0, nodes) of g_cand_dump -> out (nodes x CAND_DUMP_W ints);
// the dump is cleared to -1 afterwards (call once before the first run)
int csc_dbg_cand_read(int* out, int nodes) {
    hipDeviceSynchronize();
    if (nodes > CAND_DUMP_NODES) nodes = CAND_DUMP_NODES;
    if (nodes > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cand_dump), (size_t)nodes * CAND_DUMP_W * sizeof(int)) != hipSuccess) return -1;
    void* p = nullptr;                                                // every row back to "not written" (-1)
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_cand_dump)) != hipSuccess || hipMemset(p, 0xff, sizeof(int) * CAND_DUMP_NODES * CAND_DUMP_W) != hipSuccess) return -1;
    hipDeviceSynchronize();
    return nodes;
}
int csc_dbg_cand_width(void) { return CAND_DUMP_W; }
#endif

#ifdef CSC_PHASE_PROF
int csc_prof_read(unsigned long long* out32, int reset) {
    hipDeviceSynchronize();
    static unsigned long long host[32][PROF_BUCKETS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof host) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) {
        out32[i] = 0;
        for (int b = 0; b < PROF_BUCKETS; ++b) out32[i] += host[i][b];
    }
    if (reset) { memset(host, 0, sizeof host); hipMemcpyToSymbol(HIP_SYMBOL(g_prof), host, sizeof host); }
    return 0;
}
// the chain kernel's trace since the last call: up to cap entries of 4 words -> out; returns the count
int csc_prof_trace_read(unsigned* out, int cap) {
    hipDeviceSynchronize();
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), sizeof n) != hipSuccess) return -1;
    if (n > TRACE_CAP) n = TRACE_CAP;
    if ((int)n > cap) n = (unsigned)cap;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)n * 16) != hipSuccess) return -1;
    const unsigned zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, sizeof zero);
    return (int)n;
}
#endif

}  // extern "C"

#include "csc_static_scratch.inc"   // csc_mmpdf_batch_static: the entry point that allocates nothing
