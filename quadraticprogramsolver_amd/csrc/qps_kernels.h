// qps_kernels.h -- host-callable launchers of the gfx950 kernels (internal; the public boundary is include/qps.h).
//
// Device data layout (dense path), all in HBM, T = double or float:
//   A   : m x n, ROW-major, leading dimension NP = roundup(n, 64), MP = roundup(m, 64) rows allocated, zero padded.
//   P   : n x n symmetric, row-major (== column-major), ld NP, zero padded.
//   S   : NP x NP "sweep matrix": lower triangle = W, upper triangle = W' where W is the Cholesky factor L of
//         M = P + sigma I + rho A'A with its nb x nb diagonal blocks replaced by their inverses.  Both triangular
//         sweeps are then row-dot-product GEMVs over S (forward reads c <= r, backward reads c >= r).
//   vectors of length n / m are allocated NP / MP long and zero padded.
#pragma once
#ifndef QPS_NT_SLABS
#define QPS_NT_SLABS 1   // non-temporal (write-through) stores for the per-workgroup slabs of the pass and sweep kernels
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <thread>

namespace qps {

// ---- per-device facts (qps_capi.hip).  A process may hold handles on several devices (include/qps.h:19), so nothing that
// describes or configures a DEVICE is cached in a plain function-local static: it is keyed by the device ordinal. ----------------------
constexpr int kMaxDevices = 64;
int current_device();                 // hipGetDevice (the entry points of a handle make its device current first)
int device_cu_count(int device);      // multiProcessorCount, asked once per ordinal
// Per-device one-time settings such as hipFuncSetAttribute(MaxDynamicSharedMemorySize): `once(dev, f)` runs f exactly once per device ordinal and returns,
// on EVERY calling thread, only after f has returned -- a second host thread (distinct handles may be driven from distinct threads, include/qps.h:19) must
// not launch the kernel before the attribute is in place (its launch with 144-160 KiB of dynamic LDS would fail).
struct PerDeviceOnce {
    std::atomic<unsigned char> state[kMaxDevices] = {};          // 0 not started, 1 running, 2 done
    template <typename F> void once(int dev, F&& f) {
        if (dev < 0 || dev >= kMaxDevices) { f(); return; }
        if (state[dev].load(std::memory_order_acquire) == 2) return;
        unsigned char expect = 0;
        if (state[dev].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) { f(); state[dev].store(2, std::memory_order_release); return; }
        while (state[dev].load(std::memory_order_acquire) != 2) std::this_thread::yield();
    }
};

// A launcher that finds a pending event pair here attaches it to its kernel dispatch (hipExtLaunchKernelGGL): the events then carry the
// kernel's own begin / end timestamps -- what rocprofv3 reports -- instead of bracketing the launch from outside (+3-4 us of
// launch gap and event handling).  Thread-local: handles may be driven from different host threads.
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchTiming g_launch_timing;


// Batched launches (BASELINE config 4: many independent QPs of one shape per GPU): blockIdx.y = QP index, element strides
// between consecutive QPs for the matrix / input-vector / output-vector operands, optional per-QP active mask.
// Default-constructed = the single-QP launch (count 1, no offsets).
struct BatchStride { int count = 1; int64_t mat = 0, vin = 0, vout = 0; const int* active = nullptr; };

template <typename T> struct VecOf;
template <> struct VecOf<double> { using type = double2; static constexpr int N = 2; };
template <> struct VecOf<float>  { using type = float4;  static constexpr int N = 4; };

// ---- loop kernels (k_loop.hip) -------------------------------------------------------------------------------
// out[r] = alpha * sum_{c in [c0,c1), tri} S[r][c] v[c] + beta * out0[r]   for r in [r0, r1); tri: 0 none, 1 c<=r, 2 c>=r
template <typename T>
void gemv_rows(hipStream_t st, const T* S, int64_t ld, const T* v, T* out, const T* out0, T alpha, T beta,
               int r0, int r1, int c0, int c1, int tri, BatchStride bs = BatchStride());

// part[t][c] = sum_{r in row tile t} S[r][c] * (ca*va[r] + cb*vb[r])  for c in [0, ncols); returns number of tiles
template <typename T>
int gemv_cols_partial(hipStream_t st, const T* S, int64_t ld, const T* va, const T* vb, T ca, T cb,
                      T* part, int64_t part_ld, int nrows, int ncols);
int gemv_cols_tiles(int nrows);

// out[c] = s0*a0[c] + s1*a1[c] + sum_{t<ntiles} part[t][c]
template <typename T>
void colsum(hipStream_t st, const T* part, int64_t part_ld, int ntiles, const T* a0, T s0, const T* a1, T s1,
            T* out, int ncols, BatchStride bs = BatchStride());   // bs.mat = stride between slab sets

// SolveQuadraticProgram.jl:56-61 fused: xp=x; x=alpha*xx+(1-alpha)*x; zp=z; z=clamp(...); y=y+rho*(...)
template <typename T>
void admm_update(hipStream_t st, int NP, int MP, const T* xx, const T* zz, T* x, T* xp, T* z, T* zp, T* y,
                 const T* l, const T* u, T alpha, T rho);

// SolveQuadraticProgram.jl:79-112: norms + decision.  Ax, Px, Aty are precomputed by GEMVs.  res = 16 doubles:
//  [0] resPrim [1] resDual [2] maxNormPrim [3] maxNormDual [4] rhorho [5] convFlag [6] dx [7] dz
struct CheckScalars { double epsAbs, epsRel, epsAdmm, rho, rhorho; int adptRho; int convFlag; };
template <typename T>
void check_convergence(hipStream_t st, int n, int m, const T* Ax, const T* Px, const T* Aty, const T* q, const T* x,
                       const T* xp, const T* z, const T* zp, unsigned long long* scratch /*>=16 u64 per QP*/, double* res_dev,
                       CheckScalars cs, int dual_only = 0, BatchStride bs = BatchStride(), const double* rho_arr = nullptr,
                       const double* rhorho_arr = nullptr);   // batched: bs.vin = n-vector stride, bs.vout = m-vector stride
// dual_only = 1: the primal-side slots (0,2,3,7,8) were already filled by the fused pass (k_pass.hip); scratch is not
// cleared and only the n-length norms are added before the decision.

// ---- fused single pass over A (k_pass.hip) -----------------------------------------------------------------------
// z~ = A x~, z/y update (SolveQuadraticProgram.jl:59-61), x_new = alpha x~ + (1-alpha) x_old (:57), slabs of
// A'(rho z_new - y_new) for the next right-hand side; check = true adds A x_new norms, slabs of A'y_new, |x_new-x_old|.
template <typename T> int apass_plan(int NP, int MP, int* rows_per_wg, int count = 1);   // slabs per QP (0: unsupported)
template <typename T> int apass_max_np();
// Batched form: blockIdx.y = QP; strides are MP*ld (A), NP (n-vectors), MP (m-vectors), slabs*part_ld (slabs), 16 (slots);
// rho_arr (double per QP) overrides the scalar rho; active masks finished QPs.
struct PassBatch { int count = 1; int slabs = 0; const double* rho_arr = nullptr; const int* active = nullptr; int pq_me = 0; };
template <typename T>
void apass(hipStream_t st, bool check, const T* A, int64_t ld, int NP, int MP, const T* xx, const T* x_old, T* x_new, T* z,
           T* y, const T* l, const T* u, T alpha, T rho, T* part, T* part2, int64_t part_ld, unsigned long long* slots,
           PassBatch pb = PassBatch());
// ProxQP.jl rows over G = [A; C] (k_pass_pq.hip): v = G x, slack/dual updates (ProxQP.jl:227-249) and the slabs of
// G'[rho b - y ; rho (d - s) - z] for the next right-hand side (:212-216) in one read of G.  Returns the slab count (0: unsupported).
template <typename T> int apass_proxqp_slabs(int NP, int MP);
// polishing (SolveQuadraticProgram.m:304-305): out_lam = mask (A vx) - delta mask vlam, slabs of A'(mask vlam); 0: unsupported shape
template <typename T>
int apass_kkt(hipStream_t st, const T* A, int64_t ld, int NP, int MP, const T* vx, const T* vlam, const T* mask, T delta, T* out_lam,
              T* part, int64_t part_ld);
template <typename T>
int apass_proxqp(hipStream_t st, const T* G, int64_t ld, int NP, int MP, int me, const T* x, T* x_scratch, T* slack, T* dual,
                 const T* g, T rho, T* part, int64_t part_ld);

// ---- fused triangular sweeps (k_trsv.hip) -----------------------------------------------------------------------------
template <typename T> bool sweep_fused_supported(int NP);
// Both sweeps fused into one pass over the lower triangle (single inverted block): slabs of x~ = W' (W t); the caller adds
// the slabs with colsum().  bs.mat = stride between sweep matrices, bs.vin = stride of t, bs.vout = stride between the slab sets of two QPs.
template <typename T> int sweep_fused_slabs(int NP, int count = 1);
template <typename T>
int sweep_fused(hipStream_t st, const T* S, int64_t ld, int NP, const T* v, T* part, int64_t part_ld, BatchStride bs = BatchStride());

// ---- small-problem path (k_small.hip): the whole loop in one single-workgroup launch ---------------------------------------
template <typename T> bool admm_small_supported(int n, int m, int NP, int MP);
template <typename T> void transpose_small(hipStream_t st, const T* A, int NP, int MP, T* At);
// batch of small QPs, one workgroup per QP (register-resident kernel): per-QP argument slots are filled on the host with
// admm_small_args_set (an empty iteration window parks a QP), reports come back as admm_small_out_bytes() records
template <typename T> bool admm_small_batch_supported(int NP, int MP);
size_t admm_small_args_bytes();
void admm_small_args_set(void* host_array, int idx, int n, int m, int NP, int MP, int it_begin, int it_end, int numItrConv, int adptRho, double rho,
                         double rhorho, double sigma, double alpha, double epsAbs, double epsRel, double epsAdmm, double fctrRho);
template <typename T>
void admm_small_batch(hipStream_t st, int count, int NP, int MP, const void* args_dev, const T* A, const T* P, const T* S, const T* q, const T* l,
                      const T* u, T* x, T* xp, T* z, T* y, void* outs_dev);
// runs iterations it_begin+1 .. it_end (or until a termination test fires / the proposed rho leaves the fctrRho band);
// out_dev receives {last iteration, convFlag, need_rho, res[8]} (admm_small_out_bytes / admm_small_read)
template <typename T>
void admm_small(hipStream_t st, int n, int m, int NP, int MP, int it_begin, int it_end, int numItrConv, int adptRho, double rho,
                double rhorho, double sigma, double alpha, double epsAbs, double epsRel, double epsAdmm, double fctrRho, const T* A,
                const T* At, const T* P, const T* S, const T* q, const T* l, const T* u, T* x, T* xp, T* z, T* y, void* out_dev);
size_t admm_small_out_bytes();
void admm_small_read(const void* host_copy, int* last_it, int* convFlag, int* need_rho, double* res8);

template <typename T> void fill(hipStream_t st, T* p, int64_t n, T v);
template <typename T> void convert_copy(hipStream_t st, const double* src, T* dst, int64_t n);   // dst[i] = (T)src[i]
template <typename T> void convert_back(hipStream_t st, const T* src, double* dst, int64_t n);   // dst[i] = (double)src[i]

// ---- setup kernels (k_setup.hip) -----------------------------------------------------------------------------
// dst (row-major rows x NPc, zero padded to rowsP x NPc) = transpose-of-column-major src (rows x cols, ld lds) as T
template <typename T>
void import_colmajor(hipStream_t st, const double* src, int64_t lds, int rows, int cols, T* dst, int64_t ldd);

// C[i][j] (+)= alpha * sum_k opA(i,k) opB(k,j); all dims multiples of 64 (K multiple of 16); row-major C with ldc.
// a_kcontig: opA(i,k) = A[i*lda + k] else A[k*lda + i].  b_kcontig: opB(k,j) = B[j*ldb + k] else B[k*ldb + j].
// lower_only: skip tiles strictly above the block diagonal.  batch: blockIdx.z with element strides sA,sB,sC.
template <typename T>
void gemm(hipStream_t st, int M, int N, int K, T alpha, const T* A, int64_t lda, bool a_kcontig, const T* B, int64_t ldb,
          bool b_kcontig, T beta, T* C, int64_t ldc, bool lower_only, int batch = 1, int64_t sA = 0, int64_t sB = 0,
          int64_t sC = 0, int ktri = 0);
// ktri: 1 = opB lower triangular (k loop starts at the tile's first column), 2 = opA lower triangular (k loop stops
// after the tile's last row): structurally zero k-tiles are skipped.

// M = PI + rho * AA (lower triangle incl. diagonal tiles; NP x NP)   (LinearSystemSolvers.jl:114,128)
template <typename T> void assemble_M(hipStream_t st, int NP, const T* PI, const T* AA, T rho, T* M, int batch = 1,
                                      const double* rho_arr = nullptr);   // rho_arr: per-QP rho (device) for a batch
// PI = P + sigma I on the n x n part, identity on the padding diagonal (keeps the padded factor well defined)
template <typename T> void make_PI(hipStream_t st, int n, int NP, const T* P, T sigma, T* PI, int batch = 1);

// Blocked right-looking Cholesky of the NP x NP row-major matrix M (lower), in place.  dinv receives the inverses of
// the 64 x 64 diagonal blocks (NP/64 blocks of 64*64).  fail_dev: int32, 0 or 1+index of the failing pivot.
// scratch != nullptr: chol_scratch_elems(NP) elements per matrix (consecutive); selects the 128-column steps (one panel launch + one
// GEMM launch per 128 columns).  Any buffer that is dead during the factorisation will do (the sweep matrix S, for instance).
inline int64_t chol_scratch_elems(int NP) { return (int64_t)((NP / 64 + 1) / 2) * 3 * 4096 + 64; }   // stash tiles + the hand-off counter of k_chol_update_diag
inline bool chol_scratch_fits(int NP) { return chol_scratch_elems(NP) <= (int64_t)NP * NP; }   // an NP x NP buffer is large enough (NP >= 128)
template <typename T> void cholesky(hipStream_t st, int NP, T* M, T* dinv, int* fail_dev, int batch = 1, T* scratch = nullptr);   // batch: matrices NP*NP apart

// Build the sweep matrix S from L (lower of M) and the 64-block inverses: diagonal nb-blocks inverted by recursive
// doubling, then mirrored into the upper triangle.  tmp: NP x NP scratch.
// premul (single matrix, nb < NP): every block row is additionally multiplied by its inverted diagonal block and negated off the diagonal
// (lower blocks -W_JJ L_JK, upper blocks -(L_KJ W_JJ)'): the form the single-launch blocked sweeps read (k_trsv_blocked.hip).
template <typename T> void build_sweep_matrix(hipStream_t st, int NP, int nb, const T* L, const T* dinv, T* S, T* tmp, int batch = 1, bool premul = false);

// ---- blocked triangular substitution, one launch per sweep (k_trsv_blocked.hip) ----------------------------------------------------
// out = forward (bwd = false: rows of S left of and inside the diagonal block) or backward (bwd = true) sweep over the PREMULTIPLIED sweep
// matrix with nb x nb blocks.  pub: trsv_blocked_pub_words<T>(NP) 64-bit words, zero-filled once (hand-off granules); epoch: a value that grows
// with every launch on this pub buffer (never 0); abort_word: set to 1 by a launch that gave up waiting (its `out` is garbage then).
template <typename T> bool trsv_blocked_supported(int NP, int nb);   // on the CURRENT device (co-residency of the 256 workgroups checked per device)
template <typename T> int64_t trsv_blocked_pub_words(int NP);
// Scope around the launches of one forward + backward pair: chains the pair behind the previous blocked-sweep launches of the device
// when those went to another stream, so that two of these persistent launches never share the chip (k_trsv_blocked.hip, "Co-residency").
struct TrsvBlockedPair { int dev; hipStream_t stream; TrsvBlockedPair(int device, hipStream_t st); ~TrsvBlockedPair(); TrsvBlockedPair(const TrsvBlockedPair&) = delete; };
void trsv_blocked_forget_stream(int device, hipStream_t st);   // a handle's stream is about to be recycled / destroyed (it is idle)
template <typename T>
void trsv_blocked(hipStream_t st, bool bwd, const T* S, int64_t ld, int NP, int nb, const T* v, T* out, unsigned long long* pub,
                  unsigned epoch, unsigned* abort_word, int mode = 0);

}  // namespace qps
