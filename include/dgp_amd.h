/*
 * dgp_amd.h -- C-ABI of the MI355X-native stochastic-imputation engine.
 *
 * This is the drop-in boundary for dgpsi's SI training / imputed-GP prediction
 * hot path (mingdeyu/DGP).  Each entry point replaces one njit/LAPACK operator
 * of the reference; the reference interface is cited as file:line relative to
 * the reference repository.  A dgpsi maintainer binds these with ctypes (see
 * INTEGRATION.md); dgp_amd/ is the Python host that does exactly that.
 *
 * Conventions
 *   - every array argument is a DEVICE pointer unless its name ends in `_h`
 *     (host) ; f64 row-major (C order) and int64 indices, like numpy's;
 *   - all launches go to the HIP stream given to dgpamd_create(); nothing
 *     synchronises unless documented; workspaces are caller-allocated with the
 *     size returned by the matching *_workspace() query (bytes);
 *   - kernel kind: 0 = 'sexp', 1 = 'matern2.5' (reference passes these strings);
 *   - every function returns 0 on success, DGPAMD_BAD_ARG, or DGPAMD_HIP_ERROR
 *     (message via dgpamd_last_error).  Loss of positive-definiteness is
 *     reported through device-side `info` words (LAPACK convention: 0 = ok,
 *     j>0 = leading minor j not positive) which the host shim turns into
 *     numpy.linalg.LinAlgError (reference: scipy/numpy cholesky raising,
 *     dgp.py:1402, kernel_class.py:749).
 *
 * "Augmented" matrices.  A factorisation buffer is an Np x Np row-major array,
 * Np = dgpamd_padded_dim(n) (a multiple of 64, > n).  Rows/cols [0,n) hold the
 * SPD matrix (lower triangle is referenced), rows [n, n+r) hold r right-hand
 * sides y_q^T in columns [0,n) and the bottom-right corner starts at zero.
 * dgpamd_potrf factors the leading n columns only, so on exit row n+q holds
 * w_q = L^-1 y_q and the corner entry (n+q, n+q') holds -w_q . w_q'
 * (the quadratic forms y^T K^-1 y come out of the trailing update for free).
 */
#ifndef DGP_AMD_H
#define DGP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGPAMD_OK 0
#define DGPAMD_NOT_PD 1
#define DGPAMD_BAD_ARG 2
#define DGPAMD_HIP_ERROR 3

#define DGPAMD_SEXP 0
#define DGPAMD_MATERN25 1

#define DGPAMD_MAXD 64   /* max columns of [input | global_input] of one GP node */
#define DGPAMD_MAXB 64   /* max matrices in one batched call */

typedef struct dgpamd_ctx dgpamd_ctx;

/* ---- context ------------------------------------------------------------ */
/* stream: a hipStream_t created by the caller (e.g. a torch side stream), or
 * NULL for the device's default (null) stream -- torch's default current stream. */
int dgpamd_create(int device, void *stream, dgpamd_ctx **out);
int dgpamd_destroy(dgpamd_ctx *ctx);
const char *dgpamd_last_error(const dgpamd_ctx *ctx);
int dgpamd_sync(dgpamd_ctx *ctx);
/* Results to the host: copy `bytes` from device memory through the context's pinned staging buffer, ordered after
 * everything queued on the context's stream, and return when they have landed (one stream synchronisation). */
int dgpamd_fetch(dgpamd_ctx *ctx, const void *device_src, void *host_dst, size_t bytes);
/* The two halves of dgpamd_fetch for results the host does not need yet.  dgpamd_post queues the copy of `bytes` from device
 * memory into mailbox `slot` (0 .. DGPAMD_MAILBOXES-1) behind everything queued so far and returns at once; work queued
 * afterwards does not delay it.  dgpamd_collect waits for that one copy (not for the stream) and hands the bytes over
 * (host_dst NULL: wait and discard).  A mailbox holds one result at a time.  The lock-step M-step keeps two groups of
 * optimisers in flight this way (one group's objective evaluations run while the host advances the other group's
 * optimisers); the I-step's latents travel to the host while the M-step's first evaluations run. */
#define DGPAMD_MAILBOXES 8
int dgpamd_post(dgpamd_ctx *ctx, const void *device_src, size_t bytes, int slot);
int dgpamd_collect(dgpamd_ctx *ctx, int slot, void *host_dst, size_t bytes);
/* A sum over ranks in the middle of a queued sequence.  With the rows of a Vecchia likelihood split over several processes
 * (one per GPU), dgpamd_ess_queue leaves every batch's B x 2 sums of THIS rank's rows in a device buffer and calls
 * hook(user, device_buf, count) -- on the calling thread, while it is queueing, between the launch that produced the sums and
 * the one that reads them.  The hook queues an in-place all-reduce of the count doubles on the context's stream (RCCL:
 * ncclAllReduce on that stream; the Python host passes torch.distributed.all_reduce) and returns 0, or non-zero to abort the
 * call (DGPAMD_BAD_ARG).  Every rank makes the same calls in the same order whatever the accept decisions are (launches of
 * an update that is already closed are predicated away, the hook is not).  NULL: no hook (one process). */
typedef int (*dgpamd_reduce_hook)(void *user, double *device_buf, int count);
int dgpamd_set_reduce_hook(dgpamd_ctx *ctx, dgpamd_reduce_hook hook, void *user);
/* Two device regions (e.g. log-likelihoods and their info words) into one host buffer, back to back, one sync. */
int dgpamd_fetch2(dgpamd_ctx *ctx, const void *src_a, size_t bytes_a, const void *src_b, size_t bytes_b, void *host_dst);
const char *dgpamd_version(void);
int64_t dgpamd_padded_dim(int64_t n); /* Np for an n x n problem (>= n+1, multiple of 64) */

/* Static launch sequences (the ~100 launches of one blocked factorisation / inverse) are captured once per
 * (shape, buffers) and replayed as hipGraphs; enable = 0 turns that off (direct launches).  Graphs need a
 * real stream: with the null stream the library always launches directly.                              */
int dgpamd_set_graphs(dgpamd_ctx *ctx, int enable);

/* Timing on the library's stream (bench.py: HIP events around the timed region). */
int dgpamd_event_create(dgpamd_ctx *ctx, void **ev);
int dgpamd_event_record(dgpamd_ctx *ctx, void *ev);
int dgpamd_event_elapsed_ms(dgpamd_ctx *ctx, void *start, void *stop, float *ms_h); /* syncs on stop */
int dgpamd_event_destroy(dgpamd_ctx *ctx, void *ev);

/* Launch timing of ONE kernel class with HIP events on the launching stream (bench.py's roofline):
 * classes 1 kmatrix, 2 potrf diagonal block, 3 panel TRSM, 4 trailing SYRK, 5 trtri, 6 lauum,
 * 7 grad_reduce, 8 linked-GP J, 9 gp quadratic form.  collect() syncs and returns the number of
 * timed launches, their summed duration and their summed algorithmic work (flops; bytes for 1). */
int dgpamd_prof_enable(dgpamd_ctx *ctx, int kernel_class);
int dgpamd_prof_collect(dgpamd_ctx *ctx, int64_t *launches_h, double *total_ms_h, double *work_h);
int dgpamd_prof_event_overhead_us(dgpamd_ctx *ctx, double *us_h); /* duration an EMPTY event pair reports */

/* ---- a1/a2  kernel-matrix assembly ---------------------------------------
 * kernel.k_matrix()  kernel_class.py:304-359 (pdist/squareform + functions.py:16-34).
 * X = [Xloc[:, colmap] | Xglob]: Xloc is (n x ldloc) with batch stride
 * `stride_loc` (elements) and Dl gathered columns colmap_h[0..Dl); Xglob is
 * (n x Dg), shared by the batch (may be NULL iff Dg == 0).
 * length_h: 1 (shared) or Dl+Dg lengthscales.  Diagonal = 1 + nugget*W[i]
 * (W = NULL -> 1)  kernel_class.py:352-355.
 * Output, per batch b: K + b*stride_k, ld = ldk.
 *   full != 0 : the n x n matrix, both triangles (ldk >= n)            [k_matrix()]
 *   full == 0 : augmented factorisation buffer (ldk = Np): lower tiles of K,
 *               rows [n, n+r) <- Y (r x n, ld = ldy, batch stride stride_y; may
 *               be NULL iff r == 0), zero corner.                                */
int dgpamd_kmatrix(dgpamd_ctx *ctx, int kind, int64_t n,
                   const double *Xloc, int64_t ldloc, int64_t stride_loc, const int32_t *colmap_h, int Dl,
                   const double *Xglob, int Dg,
                   const double *length_h, int nlen, double nugget, const double *W,
                   double *K, int64_t ldk, int64_t stride_k, int full,
                   const double *Y, int64_t ldy, int64_t stride_y, int r,
                   int batch);

/* ---- LAPACK potrf (+ the forward solves and log-determinant it feeds) -----
 * scipy.linalg.cholesky / np.linalg.cholesky call sites kernel_class.py:417,483,746,
 * functions.py:109,119 ; logdet_nb functions.py:220-222 ; cho_solve with y
 * kernel_class.py:423,487.
 * In place on `batch` augmented buffers (see top).  logdet[b] = 2 sum log L_ii;
 * info[b] as LAPACK.  work: dgpamd_potrf_workspace(n,batch) bytes of device memory starting on a
 * 128-byte line (any allocator's base address does; the one-launch kernel's synchronisation words
 * sit on cache lines of their own inside it); it keeps the inverses of the 64x64 diagonal blocks
 * for dgpamd_potri.                                                                            */
size_t dgpamd_potrf_workspace(int64_t n, int batch);
int dgpamd_potrf(dgpamd_ctx *ctx, int64_t n, double *A, int64_t stride_a, int batch,
                 double *logdet, int32_t *info, void *work);
/* How dgpamd_potrf / dgpamd_potrf_inv (and everything built on them) run the blocked factorisation:
 * mode 1 (default): ONE persistent launch -- a pivot-chain workgroup per matrix plus workers that pull tile tasks
 * from a queue, ordered by per-tile version words (csrc/chol.hip, potrf_mega_kernel); mode 0: one launch per 64-column
 * block step, replayed as a hipGraph (no waits between workgroups inside a launch: the mode for a device shared with other
 * processes).  The results are bit-identical (every tile applies its panels in the same order).
 * info[b] = -1 reports a lost in-kernel hand-off (a bounded spin gave up), never a numerical failure: the caller
 * rebuilds the matrices and runs the call again in mode 0 (dgp_amd.dgp.train does, once per iteration). */
int dgpamd_set_potrf_mode(dgpamd_ctx *ctx, int mode);

/* Read the quadratic forms out of factored buffers: quad[b*r*r + q*r + q'] =
 * y_q^T K^-1 y_q'  (= -corner).                                               */
int dgpamd_aug_quad(dgpamd_ctx *ctx, int64_t n, const double *A, int64_t stride_a, int batch, int r, double *quad);

/* The closing arithmetic of dgpamd_loglik (kernel_class.py:486-488) for buffers that were assembled with their y row and
 * factored by another call (e.g. as extra matrices of a batched dgpamd_potrf):
 * ll[b] = -0.5 (n log scale + logdet[b] + y'K^-1y / scale). */
int dgpamd_loglik_finish(dgpamd_ctx *ctx, int64_t n, const double *A, int64_t stride_a, int batch, const double *logdet,
                         double scale, double *ll);

/* Dense row-major matrix-vector product out = A x (rows x cols, leading dimension ld): the K alpha + u step of the
 * heteroskedastic exact-posterior draw (likelihood_class.py:184-243), where K is a full kernel matrix. */
int dgpamd_gemv(dgpamd_ctx *ctx, int64_t rows, int64_t cols, const double *A, int64_t ld, const double *x, double *out);

/* ---- a5  ESS target log-likelihood, batched over speculative proposals -----
 * kernel.log_likelihood_func  kernel_class.py:481-492:
 *   ll[b] = -0.5 ( n log(scale) + logdet(K_b) + y^T K_b^-1 y / scale )
 * Fused pipeline: kmatrix (augmented with y) -> potrf -> read-out.  A is scratch
 * for `batch` augmented buffers.  y: (n) shared by the batch.                  */
int dgpamd_loglik(dgpamd_ctx *ctx, int kind, int64_t n,
                  const double *Xloc, int64_t ldloc, int64_t stride_loc, const int32_t *colmap_h, int Dl,
                  const double *Xglob, int Dg,
                  const double *length_h, int nlen, double nugget, const double *W, double scale,
                  const double *y, double *A, int64_t stride_a, int batch,
                  double *ll, int32_t *info, void *work);

/* ---- a3  fmvn: nu = sqrt(scale) * L z  -------------------------------------
 * functions.fmvn  functions.py:113-121 (chol(scale*K) z = sqrt(scale) chol(K) z).
 * L: factored augmented buffers (batch stride stride_a); z, out: (batch x n).   */
int dgpamd_trmv_lower(dgpamd_ctx *ctx, int64_t n, const double *L, int64_t stride_a, const double *scale_h,
                      const double *z, double *out, int batch);

/* ---- a4  ESS proposals: FP[b] = F cos(theta_b) + NU sin(theta_b) -----------
 * functions.update_f  functions.py:203-208.  F, NU: (n x M); FP: (batch x n x M). */
int dgpamd_ess_propose(dgpamd_ctx *ctx, int64_t n, int M, const double *F, const double *NU,
                       const double *theta_h, int batch, double *FP);

/* ---- inverse from the factor (cho_solve(L, I))  kernel_class.py:418,747 -----
 * On entry A holds dgpamd_potrf output (and `work` its workspace).  On exit
 * Ainv (Np x Np buffer) holds K^-1 in BOTH triangles of [0,n)x[0,n) and row n+q
 * of columns [0,n) holds -alpha_q^T = -(K^-1 y_q)^T.  A is overwritten with L^-1. */
int dgpamd_potri(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int r, void *work);
/* Factorisation AND inverse in one sweep (replaces potrf + potri where both are wanted, e.g. kernel.llik
 * kernel_class.py:417-423): the identity rides along as n extra rows, so the right-looking sweep leaves
 *   A: as dgpamd_potrf;   T (Np x Np): L^-T in its first n rows (block upper triangular; column n = -K^-1 y_0);
 *   S (Np x Np): K^-1 in the lower 64x64 tiles of its first n rows/columns (diagonal tiles full), row n = -(K^-1 y_0)^T
 * -- the layout dgpamd_potri leaves in Ainv, minus the strictly upper tiles.  T and S need no initialisation.
 * The inverse's GEMM work runs in the shadow of the factorisation's pivot chain instead of after it. */
int dgpamd_potrf_inv(dgpamd_ctx *ctx, int64_t n, double *A, double *T, double *S, int64_t stride_a, int batch,
                     double *logdet, int32_t *info, void *work);

/* Same on `batch` buffers (A and Ainv share the batch stride stride_a; work = the batched potrf workspace). */
int dgpamd_potri_batched(dgpamd_ctx *ctx, int64_t n, double *A, double *Ainv, int64_t stride_a, int r, int batch,
                         void *work);

/* A GP node as the batched entry points take it (host struct; pointers as commented). */
typedef struct {
    int kind;              /* DGPAMD_SEXP / DGPAMD_MATERN25 */
    int Dl, Dg, nlen;      /* local / global input columns, number of lengthscales */
    int nugget_est;
    int reserved;
    int64_t ldloc;
    const double *Xloc;    /* device, n x ldloc */
    const int32_t *colmap; /* host, Dl entries, or NULL */
    const double *Xglob;   /* device, n x Dg, or NULL */
    const double *length;  /* host, nlen entries */
    double nugget;
    const double *W;       /* device replicate weights or NULL */
    const double *y;       /* device, n */
    /* Vecchia form of the node (read by dgpamd_ess_queue only; vecch_nn == NULL: dense).  kernel.log_likelihood_func_vecch
     * kernel_class.py:494-509 = vecchia_llik (vecchia.py:165-180) on the inputs and outputs permuted by kernel.ord. */
    const int64_t *vecch_ord;   /* device, n: the ordering (row i of the ordered arrays is point vecch_ord[i]) */
    const int64_t *vecch_nn;    /* device, n x (vecch_m + 1): NNarray (ordered coordinates) */
    const double *vecch_nd;     /* device, n: nugget weights in ordered coordinates (ones without replicates) */
    const double *vecch_y;      /* device, n: the outputs in ordered coordinates */
    int vecch_m, reserved2;
    /* This rank's rows of the Vecchia likelihood (dist.split_training(rows=True)): rows vecch_row0 .. vecch_row0 + vecch_rows - 1
     * of vecch_nn; vecch_rows == 0: all n.  The partial sums of a batch are completed by the context's reduce hook
     * (dgpamd_set_reduce_hook) before the accept / shrink decision reads them. */
    int64_t vecch_row0, vecch_rows;
    /* Likelihood node on top of the model (read by dgpamd_ess_queue and dgpamd_lik_loglik only; lik_kind == 0: a GP node).
     * The reference's plugin protocol llik() (likelihood_class.py:30-90 Poisson, :245-292 NegBin, :470-621 ZIP, :624-812 ZINB,
     * the classification likelihood): y = the lik_nobs observations (device; class indices for the categorical kinds),
     * colmap / Dl = the latent columns it reads (one per class for SOFTMAX / ROBUSTMAX), lik_rep = latent row of every
     * observation (device, replicates) or NULL (observation i reads row i), lik_par = robustmax's epsilon. */
    int lik_kind, lik_classes;
    int64_t lik_nobs;
    const int64_t *lik_rep;
    double lik_par;
} dgpamd_node;
enum { DGPAMD_LIK_POISSON = 1, DGPAMD_LIK_NEGBIN = 2, DGPAMD_LIK_ZIP = 3, DGPAMD_LIK_ZINB = 4, DGPAMD_LIK_BIN_LOGIT = 5,
       DGPAMD_LIK_BIN_PROBIT = 6, DGPAMD_LIK_ROBUSTMAX = 7, DGPAMD_LIK_SOFTMAX = 8 };

/* ---- a7  a likelihood node's log-likelihood of `batch` candidate blocks ------------
 * imputation.py:71-78,91-106 -> <likelihood>.llik(): sum_i log p(y_i | f_i) for every block X[b] (n x M, stride_x doubles
 * apart; 0: one block), f_i = row lik_rep[i] (or i), columns colmap.  out (device, batch); work: dgpamd_lik_workspace(batch)
 * bytes.  The sums are formed in a fixed order (the same bits as inside dgpamd_ess_queue). */
size_t dgpamd_lik_workspace(int batch);
int dgpamd_lik_loglik(dgpamd_ctx *ctx, const dgpamd_node *node, int64_t n, int M, const double *X, int64_t stride_x, int batch,
                      double *work, double *out);

/* ---- a7  one elliptical-slice update of a latent block, loop and all ------------
 * imputation.one_sample_block imputation.py:81-119 for the common case of ONE dense GP node upstairs: the
 * shrinking-bracket loop runs here (speculative batches: propose -> batched log-likelihood -> one result copy per
 * batch), so an update costs one library call instead of several host round trips per batch.
 *   F (n x M, device): current latent block, overwritten by the accepted proposal; NU: the prior draw;
 *   node: the upper node (its Xloc / ldloc are ignored: the proposals are its local input through node->colmap);
 *   scale: its variance; log_y: the slice threshold; state[4] = {theta, lo, hi, pending}: the current angle and
 *   bracket (pending != 0: the last batch was rejected and its closing shrink still needs a uniform);
 *   uniforms[nuni]: the next draws of the sampler's uniform stream, consumed exactly as the sequential loop
 *   (imputation.py:115-119) would; batch_first / batch_next: speculative proposals per batch;
 *   FP (batch_first x n x M), A (batch_first x Np x Np), work (dgpamd_potrf_workspace(n, batch_first)),
 *   ll_dev (batch_first doubles), info_dev (batch_first int32): device scratch.
 * out[6] = {status, uniforms consumed, proposals evaluated, batches, accepted log-likelihood, info}: status 0 =
 * accepted, 1 = uniforms used up (call again with more; state is updated), 2 = a proposal's matrix is not positive
 * definite (info = the LAPACK-style index). */
int dgpamd_ess_update(dgpamd_ctx *ctx, int64_t n, int M, double *F, const double *NU, const dgpamd_node *node, double scale,
                      double log_y, double *state, const double *uniforms, int nuni, int batch_first, int batch_next,
                      double *FP, double *A, void *work, double *ll_dev, int32_t *info_dev, double *out);

/* ---- a7  several elliptical-slice updates of a latent block queued without host synchronisation ------------
 * imputation.py:44-119, the accept / shrink loop advanced on the device: after every speculative batch a one-thread
 * kernel compares the batch's log-likelihoods (summed over the `nnodes` dense GP nodes of the layer above) with the
 * threshold, takes the first accepted proposal into F or shrinks the bracket, and moves the cursor into the uniform
 * stream exactly as the sequential loop would; the launches of an update's later batches are predicated on its `done`
 * word.  `nupd` updates (prior draws NU: nupd x n x M) are queued on the context's stream and the call returns at once.
 *   state (device, DGPAMD_ESS_STATE doubles): {theta, lo, hi, pending, cursor, status, info, ll, log_y, proposals,
 *     batches, updates}; the caller zeroes it and sets cursor / ll as needed (compute_ll0 = 1: ll is computed from F first; compute_ll0 = 2 (round 6): the FIRST
 *     update of this call is the update an earlier queue left open -- out of queued batches (status 3) or of uploaded uniforms (1): the caller writes the state back with
 *     status and counters zeroed and the cursor at the start of this call's uniforms, theta / lo / hi / pending / log_y / ll as fetched, and the update goes on where it stopped).
 *     status after the queue: 0 = every update accepted; 1 = uniforms used up; 2 = a proposal the sequential loop reaches
 *     is not positive definite (info); 3 = an update was not accepted within max_batches batches; 4 = uniforms used up
 *     before an update began (nothing of it is in the state).  After a non-zero status
 *     the later updates leave F alone: `updates` tells how many were completed, and {theta, lo, hi, pending, cursor, log_y}
 *     are those of the open update (resume it with dgpamd_ess_update).
 *   uniforms / log_uniforms (device, nuni): the sampler's next uniforms and their logarithms (taken on the host so
 *     that the thresholds are bit-identical to the host loop's).
 *   FP (batch_first x n x M), A (batch_first x Np x Np), work (dgpamd_potrf_workspace(n, batch_first)),
 *   scratch (dgpamd_ess_queue_scratch() bytes): device scratch.
 *   Vecchia nodes upstairs (nodes[k].vecch_nn != NULL, kernel_class.py:494-509): every batch gathers the candidates'
 *     ordered inputs and evaluates vecchia_llik for all of them in one row launch; vwork (dgpamd_ess_queue_vwork(n, max
 *     Dl + Dg, batch_first) bytes, device) holds the gathered inputs and the per-row partials.  A, work may be NULL when
 *     every node upstairs is a Vecchia node; vwork may be NULL when none is.
 *   A likelihood node upstairs (nodes[k].lik_kind != 0; scales_h[k] is ignored): the summed log-density of its observations.
 *   A queue may be continued: a later call on the same `state` that the caller has NOT zeroed carries on from its cursor,
 *     status and counters (ll is recomputed with compute_ll0 when the target changed) -- the layers of a deeper model are
 *     queued one update at a time this way, with one fetch of the state at the end. */
#define DGPAMD_ESS_STATE 16
size_t dgpamd_ess_queue_scratch(void);
size_t dgpamd_ess_queue_vwork(int64_t n, int D, int batch);
int dgpamd_ess_queue(dgpamd_ctx *ctx, int64_t n, int M, double *F, const double *NU, int nupd, const dgpamd_node *nodes,
                     const double *scales_h, int nnodes, double *state, const double *uniforms, const double *log_uniforms,
                     int nuni, int batch_first, int batch_next, int max_batches, int compute_ll0, double *FP, double *A,
                     void *work, void *scratch, void *vwork);
/* Folds the `count` info words of a factorisation queued between two updates of a queue (the prior factors of a deeper
 * layer, imputation.py:54-63) into the queue's state: the first non-zero one stops the queue with status 2. */
int dgpamd_ess_queue_note_info(dgpamd_ctx *ctx, double *state, const int32_t *info, int count);

/* ---- a8  M-step objective pieces -------------------------------------------
 * kernel.llik  kernel_class.py:403-449 restructured:
 *   tr_p   = sum_ij Kinv_ij dK_p,ij          (= trace(K^-1 dK_p))
 *   quad_p = sum_ij alpha_i alpha_j dK_p,ij  (= y^T K^-1 dK_p K^-1 y)
 * with dK_p recomputed in flight (never stored).  p runs over the lengthscales
 * (nlen == 1: the summed derivative) and, iff nugget_est, the nugget last.
 * out (host-visible after sync): out[0..P) = tr_p, out[P..2P) = quad_p.
 * Ainv: from dgpamd_potri (alpha is read from its row n, negated).             */
int dgpamd_grad_reduce(dgpamd_ctx *ctx, int kind, int64_t n,
                       const double *Xloc, int64_t ldloc, const int32_t *colmap_h, int Dl,
                       const double *Xglob, int Dg,
                       const double *length_h, int nlen, double nugget, const double *W, int nugget_est,
                       const double *Ainv, double *out, void *work);

/* kernel.llik's device part (kernel_class.py:403-449) for `batch` GP nodes of the same size n in ONE call -- the
 * lock-step M-step (dgp.py:1391-1398 fits the nodes one after another; their objectives are independent): per node
 * K assembly with its own inputs / hyper-parameters (y as augmented row), ONE batched factorisation and inverse,
 * per-node derivative reductions, ONE device-to-host copy.  The call returns after the results have landed:
 *   host_out[b * stride_out + ...] = { logdet K_b, y' K_b^-1 y, info (0 = PD), tr_p (P_b values), quad_p (P_b values) }
 * with P_b as in dgpamd_grad_reduce; stride_out >= 3 + 2 max P_b.  A, T, Ainv: batch x Np x Np buffers (stride
 * stride_a; see dgpamd_potrf_inv); work: dgpamd_potrf_workspace(n, batch); grad_work: batch x dgpamd_grad_workspace(n, max P_b)
 * (the nodes' reductions run side by side in one launch, like their K assemblies);
 * dev_out: device scratch of batch * (stride_out + 2) doubles. */
int dgpamd_llik_batch(dgpamd_ctx *ctx, int64_t n, int batch, const dgpamd_node *nodes, double *A, double *T, double *Ainv,
                      int64_t stride_a, void *work, void *grad_work, double *dev_out, double *host_out,
                      int64_t stride_out);
/* The same call in two halves (round 6): _launch queues everything and returns, _wait returns when the results have landed in host_out (the layout above).
 * Between the two the caller may do host work that does not touch the evaluations' inputs -- dgp.train's loop refreshes the nodes' numpy attributes from the I-step's
 * device state and runs the R2 diagnostics (dgp.py:1391-1398 would do both BEFORE the first objective evaluation) while the device works on the M-step's first round.
 * One evaluation in flight per context; every other entry point stays usable in between (the results have a pinned staging buffer of their own). */
int dgpamd_llik_batch_launch(dgpamd_ctx *ctx, int64_t n, int batch, const dgpamd_node *nodes, double *A, double *T, double *Ainv,
                             int64_t stride_a, void *work, void *grad_work, double *dev_out, int64_t stride_out);
int dgpamd_llik_batch_wait(dgpamd_ctx *ctx, double *host_out);
size_t dgpamd_grad_workspace(int64_t n, int nparam);

/* ---- a11  GP prediction ------------------------------------------------------
 * functions.gp  functions.py:379-394 (+ K_vec_nb vecchia.py:244-265):
 *   m_t = Rinv_y . r_t ,  v_t = | scale (1 + nugget - r_t^T Rinv r_t) |
 * x: (M x D) test inputs already concatenated [local | global]; Wtr: (n x D)
 * training inputs; Rinv: n x n symmetric with leading dimension ldr.
 * ry: (nry x n) -- several R^-1 y at once: for a first-layer node R^-1 and the
 * variance are identical in every imputation, only y (hence the mean) differs
 * (emulation.py:701-734 recomputes both per imputation).  mean: (nry x M); var: (M).
 * work: dgpamd_gp_workspace(n, M) bytes.                                       */
size_t dgpamd_gp_workspace(int64_t n, int64_t M);
int dgpamd_gp_predict(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int D,
                      const double *x, const double *Wtr, const double *length_h, int nlen,
                      const double *Rinv, int64_t ldr, const double *ry, int nry, double scale, double nugget,
                      double *mean, double *var, void *work);

/* ---- a12-a14  linked-GP prediction ------------------------------------------
 * functions.link_gp  functions.py:396-430 with IJ_sexp :432-451 / IJ_matern
 * :453-494 (Jd, Jd0 vecchia.py:915-988), trace_sum :496-506, quad vecchia.py:990:
 *   mean_t = I_t . ry ,  var_t = | ry^T J_t ry - mean_t^2 + scale (1 + nugget - tr(Rinv J_t)) |
 * m, v: (M x Dw) input moments; z: (M x Dz) deterministic global inputs (NULL iff
 * Dz == 0); Wtr: (n x Dw), Wg: (n x Dz).  Psexp/R2sexp are never materialised. */
size_t dgpamd_linkgp_workspace(int64_t n, int64_t M, int Dw);
/* Matern-2.5: by default the J factor is evaluated through its separable form (every erf/exp depends on one
 * training point: Jd = <S(x_min), T(x_max)> + (erf_hi - erf_lo) <S', T'>, csrc/linkfun.hpp), which agrees with the
 * reference's expression to ~1e-11; enable = 1 evaluates the reference's direct expression (vecchia.py:915-959). */
int dgpamd_set_linkgp_direct(dgpamd_ctx *ctx, int enable);

/* Diagnostics: when device_buf (>= 4096 int64 in device memory) is set, the factorisation kernels of matrix 0 write
 * wall_clock64() stamps of their phases into it (16 slots per launch; tools/gpu_potrf_trace.py decodes them).
 * NULL switches it off (the default). */
int dgpamd_debug_trace(dgpamd_ctx *ctx, long long *device_buf);
/* Diagnostics of the one-launch factorisation (tools/gpu_mega_tasklog.py): when device_buf (`words` int64 of device memory) is
 * set, every chain step and every worker task of every matrix writes its wall_clock64() stamps into it -- chain step k of matrix b
 * at 64 + 8 (b nbk + k): start, factored, panel tile staged, diagonal tile seen, solved, updated; task `slot` of matrix b at
 * 64 + 8 batch nbk + 8 (b ntask + slot): pulled, inputs seen, arithmetic done, W_k seen, stored, published, workgroup.  A log that
 * would not fit is not written.  dgpamd_debug_mega_table copies the task table those slots index (8 int32 per task as MTask, then
 * 2 int32 per block: the visits of A[k+1][k] and A[k+1][k+1] the chain waits for) and returns the number of tasks, or a
 * negative status.  NULL switches the log off (the default).  With DGPAMD_JSEP_LOG=1 in the environment the same buffer takes the
 * step log of the next Matern pair-kernel launch of dgpamd_linkgp_predict instead (per workgroup 8 + 48 x 4 x 2 int64 from word 64:
 * start / end on the 100-MHz clock and in shader cycles, HW_ID, XCC_ID, tile, steps; then per step and wave the cycle counts at the
 * arrival at and the departure from the step's barrier, the order class in the top bits: tools/gpu_pair_steplog.py). */
int dgpamd_debug_tasklog(dgpamd_ctx *ctx, long long *device_buf, long long words);
int dgpamd_debug_mega_table(dgpamd_ctx *ctx, int64_t n, int inv, int batch, int32_t *host_out, int64_t cap_words);
int dgpamd_linkgp_predict(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int Dw, int Dz,
                          const double *m, const double *v, const double *z,
                          const double *Wtr, const double *Wg, const double *length_h, int nlen,
                          const double *Rinv, int64_t ldr, const double *ry, double scale, double nugget,
                          double *mean, double *var, void *work);
/* Leave-one-out form of the same prediction (emulation.py:90-143 with a dense emulator: vecchia.py:23-26 and
 * kernel_class.py:647-664 hand test row k all training points but row k): drop[t] in [0, n) is the training point
 * left out of the conditioning set of test point t (any index: the kernel does not assume drop[t] == t).  Nothing is refactorised: with u = Rinv[:, d], (R_-d)^-1 = Rinv - u u^T / u_d
 * (embedded), applied inside the pair weights.  Same workspace as dgpamd_linkgp_predict. */
int dgpamd_linkgp_loo(dgpamd_ctx *ctx, int kind, int64_t n, int64_t M, int Dw, int Dz,
                      const double *m, const double *v, const double *z,
                      const double *Wtr, const double *Wg, const double *length_h, int nlen,
                      const double *Rinv, int64_t ldr, const double *ry, const int32_t *drop, double scale,
                      double nugget, double *mean, double *var, void *work);

/* ---- a15  imputation-moment accumulation (emulation.py:846-847) --------------
 * sum_mu += mu ; sum_m2 += mu^2 + var   (count elements).  finalize:
 * mu = sum_mu / S ; var = sum_m2 / S - mu^2.                                   */
int dgpamd_moments_accumulate(dgpamd_ctx *ctx, int64_t count, const double *mu, const double *var,
                              double *sum_mu, double *sum_m2);
int dgpamd_moments_finalize(dgpamd_ctx *ctx, int64_t count, double S, double *sum_mu_to_mu, double *sum_m2_to_var);

/* ---- a17  Vecchia neighbour search --------------------------------------------
 * vecchia.nn  vecchia.py:61-109: NNarray (n x (m+1)) int64, row i = i and its <= m
 * nearest EARLIER points (exact, squared-euclidean on x), sorted by index
 * descending, -1 padded.  vecchia.get_pred_nn  vecchia.py:20-40: (M x m) nearest
 * training points, nearest first.  x, q are already divided by the lengthscales. */
int dgpamd_nn_ordered(dgpamd_ctx *ctx, int64_t n, int D, const double *x, int m, int64_t *NNarray);
int dgpamd_nn_query(dgpamd_ctx *ctx, int64_t M, int64_t n, int D, const double *q, const double *x, int m, int64_t *NN);

/* Debugging aid: fill the LDS of every CU with NaNs (a kernel that reads LDS it has not written then shows it in its results
 * whatever ran before).  DGPAMD_POISON_LDS=1 does this before the Vecchia row / prediction launches, =2 makes the Python
 * engine call this entry point before every library call. */
int dgpamd_debug_poison_lds(dgpamd_ctx *ctx);

/* ---- a19-a21  Vecchia likelihoods and sampler ---------------------------------
 * vecchia_llik vecchia.py:164-180 ; vecchia_nllik :182-242 (raw sums; the host
 * finishes the scale_est / replicate branches) ; L_matrix :409-424 ;
 * forward_solve_sp :111-120.  X: (n x D) ordered inputs; y: (n); nugget_diag: (n).
 * out_llik (device, 2 doubles): {quad, logdet}.  out_nllik (device, 2+2P
 * doubles): {quad, logdet, dquad[P], dlogdet[P]}, P = nlen (+1 iff nugget_est). */
int dgpamd_vecchia_llik(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, const double *y,
                        const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                        const double *nugget_diag, double *out_llik);
/* The same for `batch` input sets at once (the candidate blocks of one speculative ESS batch, imputation.py:91-106
 * called once per proposal): X holds them x_stride doubles apart, y / NNarray / nugget_diag are shared; out_llik
 * (device, batch x 2 doubles).  One row launch and one reduction for the whole batch. */
int dgpamd_vecchia_llik_batch(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, int64_t x_stride,
                              int batch, const double *y, const int64_t *NNarray, const double *length_h, int nlen,
                              double nugget, const double *nugget_diag, double *out_llik);
int dgpamd_vecchia_nllik(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X, const double *y,
                         const int64_t *NNarray, const double *length_h, int nlen, double nugget,
                         const double *nugget_diag, int nugget_est, double *out_nllik);
int dgpamd_vecchia_lmatrix(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X,
                           const int64_t *NNarray, const double *length_h, int nlen, double nugget, double *Lmat);
int dgpamd_vecchia_spsolve(dgpamd_ctx *ctx, int64_t n, int m, const double *Lmat, const int64_t *NNarray,
                           double inv_sqrt_scale, const double *b, double *x);
/* The same for nmat matrices (Lmat, NNarray: nmat x n x (m+1); inv_sqrt_scale: nmat, device) with nrhs right-hand
 * sides each (b, x: nmat x nrhs x n): one workgroup per chain -- fmvn_sp (vecchia.py:133-140) for all nodes of a
 * layer and all sweeps of imputer.sample (imputation.py:54-63) in one launch.                                   */
int dgpamd_vecchia_spsolve_batch(dgpamd_ctx *ctx, int64_t n, int m, int nmat, int nrhs, const double *Lmat,
                                 const int64_t *NNarray, const double *inv_sqrt_scale, const double *b, double *x);
/* The same substitution level-scheduled (rows whose dependencies are all solved run side by side: a few hundred steps
 * instead of n).  dgpamd_vecchia_levels builds the schedule of each of the nmat neighbour arrays (it depends on the array
 * only: once per ordering, kernel_class.py:245-277) into sched (device, dgpamd_vecchia_levels_bytes(n, nmat));
 * dgpamd_vecchia_spsolve_levels then solves like dgpamd_vecchia_spsolve_batch (same arguments, same results up to the
 * order of each row's sum: a fixed 32-lane butterfly instead of left to right). */
size_t dgpamd_vecchia_levels_bytes(int64_t n, int nmat);
int dgpamd_vecchia_levels(dgpamd_ctx *ctx, int64_t n, int m, int nmat, const int64_t *NNarray, void *sched);
int dgpamd_vecchia_spsolve_levels(dgpamd_ctx *ctx, int64_t n, int m, int nmat, int nrhs, const double *Lmat,
                                  const int64_t *NNarray, const double *inv_sqrt_scale, const double *b, double *x,
                                  const void *sched);
/* Hetero likelihood under Vecchia: rows of the sparse factor of the latent mean's conditional posterior
 * (vecchia.U_matrix :426-446 / U_matrix_sp :599-610 through kernel.ord_nn(pointer=True) kernel_class.py:268-275;
 * consumer Hetero.post_het_vecch likelihood_class.py:166-182).  X: (n x D) ORDERED inputs; impNN: (n x (m+1))
 * imp_NNarray; gamma, y: (n) ordered noise variances and observations (site-pooled with replicates).
 * Out, in dgpamd_vecchia_spsolve's layout: Lrows (n x (m+1)), NNl (n x (m+1)), and t = U_ol^T y (n):
 * f = -solve(t) + solve(z) is the draw in ordered coordinates.  info (device int32): 0, or 1 + the first row whose
 * block is not positive definite (the reference's np.linalg.cholesky raises LinAlgError there).             */
int dgpamd_vecchia_het_rows(dgpamd_ctx *ctx, int kind, int64_t n, int D, int m, const double *X,
                            const int64_t *impNN, const double *length_h, int nlen, double scale,
                            const double *gamma, const double *y, double *Lrows, int64_t *NNl, double *t, int32_t *info);

/* ---- a22/a23  Vecchia prediction -------------------------------------------------
 * gp_vecch vecchia.py:635-654 ; link_gp_vecch :758-796 (IJ_nb :838-907).
 * NN: (M x pm) conditioning sets, the valid entries first, -1 padded.  Up to 51 (gp; D <= 16) resp. 50 (link_gp, squared
 * exponential, Dw <= 8, Dz <= 8) neighbours run in the register-resident kernels (csrc/vecchia_pred.hip: one wave per test
 * point, no LDS); larger sets, wider inputs and the Matern link_gp in the one-wave-per-point LDS kernels (conditioning
 * sets up to what fits 160 KiB of LDS).  DGPAMD_VECCHIA_LDS=1 selects the LDS kernels throughout (the tests compare the two). */
int dgpamd_vecchia_gp(dgpamd_ctx *ctx, int kind, int64_t M, int64_t n, int D, int pm, const double *x,
                      const double *w, const int64_t *NN, const double *y, double scale,
                      const double *length_h, int nlen, double nugget, const double *nugget_diag,
                      double *mean, double *var);
int dgpamd_vecchia_linkgp(dgpamd_ctx *ctx, int kind, int64_t M, int64_t n, int Dw, int Dz, int pm,
                          const double *m, const double *v, const double *z, const double *w1, const double *wg,
                          const int64_t *NN, const double *y, double scale, const double *length_h, int nlen,
                          double nugget, const double *nugget_diag, double *mean, double *var);

#ifdef __cplusplus
}
#endif
#endif /* DGP_AMD_H */
