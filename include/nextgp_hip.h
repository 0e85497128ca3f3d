/*
 * nextgp_hip.h -- C ABI of libnextgp_hip.so, the MI355X (gfx950) replacement for the
 * marker-effect Gibbs hot path of NextGP.jl.
 *
 * Every entry point is what a Julia `ccall((:sym, "libnextgp_hip"), Cint, (...), ...)` binds;
 * INTEGRATION.md shows the reference-side shim.  All functions return 0 on success and a
 * negative code on failure; the message is retrievable with ngp_last_error().  No C++
 * exception, signal handler or exit() crosses this boundary.  Arrays are Julia-native:
 * column-major, contiguous, Float64 / Int64.  The library never retains a caller pointer
 * beyond the call; anything it keeps (panel, y, priors) is copied to device memory.
 *
 * Reference interfaces replaced (file:line under /root/reference):
 *   - coarse seam: samplers.runSampler!            src/samplers.jl:23-106 (called at src/MCMC.jl:39)
 *   - fine seam:   M[set].funct callback           src/samplers.jl:52, stored at src/mme.jl:326,333,355
 *                  = sampleBayesPR!/sampleBayesB!  src/functions.jl:118-137, 157-195
 *   - marker-matrix builders                       src/prepMatVec.jl:113-134, src/mme.jl:282-347,443-446,492-520
 */
#ifndef NEXTGP_HIP_H
#define NEXTGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_ABI_VERSION 4

#define NGP_OK 0
#define NGP_ERR_ARG (-1)     /* bad argument (null, size mismatch, non-finite value) */
#define NGP_ERR_STATE (-2)   /* call sequence error (panel / y / sets missing) */
#define NGP_ERR_HIP (-3)     /* HIP runtime error (message carries hipGetErrorString) */
#define NGP_ERR_NOMEM (-4)
#define NGP_ERR_NODEVICE (-5) /* no usable gfx950 device: the library has NO CPU fallback */
#define NGP_ERR_DEBUG (-6)    /* the call ran in a diagnostic timing mode (ngp_debug_set_mode): its results are invalid */

#define NGP_METHOD_BAYESPR 0 /* src/runTime.jl:30-45 */
#define NGP_METHOD_BAYESB 1  /* src/runTime.jl:48-61 */
#define NGP_METHOD_BAYESC 2  /* src/runTime.jl:64-77, sampler src/functions.jl:197-235 */
#define NGP_METHOD_BAYESR 3  /* src/runTime.jl:78-93, sampler src/functions.jl:238-289; added with ngp_add_marker_set_r */
#define NGP_METHOD_TUPLE 4   /* correlated sets, BayesPR's Tuple method: src/functions.jl:140-154; added with ngp_add_marker_set_tuple */

typedef struct ngp_handle ngp_handle;

int32_t ngp_abi_version(void);

/* One handle = one chain on one device (reference: one Julia task, src/samplers.jl:23).
 * seed/chain_id key every random stream (the reference never seeds its RNG). */
int32_t ngp_create(int32_t device, uint64_t seed, uint32_t chain_id, ngp_handle **out);
int32_t ngp_destroy(ngp_handle *h);
/* Message of the last failing call on h (h == NULL: last failing ngp_create). Valid until the next call. */
const char *ngp_last_error(ngp_handle *h);

/* Sweep engine, to be chosen BEFORE the panel is set (it fixes the tiling and the Gram window):
 * mode 1 (default) = one persistent kernel per iteration with look-ahead `lag` (1..8 blocks of 64 SNPs),
 * mode 0 = one streaming + one recursion launch per 64-SNP block (lag 1).  Both replace the same
 * reference loop (src/functions.jl:124-136) and draw the same chain; only summation order differs. */
int32_t ngp_configure(ngp_handle *h, int32_t mode, int32_t lag);
int32_t ngp_get_config(ngp_handle *h, int32_t *mode, int32_t *lag);
/* Persistent sweep only: look-ahead lags 1..near are corrected inside the sampler workgroup, lags near+1..lag-1 by the
 * reducer workgroups (another summation order, which the blocked oracle needs to know).  1..4, 0 = automatic (3 with the phase streamer,
 * 4 on its shards taller than 128 rows; 2 with the row-owning streamer from 64-row shards on); to be chosen before the panel is set.  ngp_get_near_lags reports the value in force. */
int32_t ngp_set_near_lags(ngp_handle *h, int32_t near);
int32_t ngp_get_near_lags(ngp_handle *h, int32_t *near);
/* Form of the block chain of BayesPR blocks (every lane BayesPR or unowned; replaces the per-SNP loop of src/functions.jl:124-136).
 * 0 (default): the 64 serial steps.  1: dlt = T e0 with T = inv(I + diag(c) strictLower(G)) of the block formed explicitly before every
 * sweep (k_tinv) -- the forward substitution the 64 steps carry out, as one 64 x 64 product.  The same Markov chain in real
 * arithmetic; the floating-point order differs, so the blocked oracle is told (ora_set_tform).  Measured:
 * pays where the sampler's serial chain is the bound (compact storage), costs 2-4 % where its CU's memory traffic is.  Any time
 * before ngp_run. */
int32_t ngp_set_chain_form(ngp_handle *h, int32_t form);
int32_t ngp_get_chain_form(ngp_handle *h, int32_t *form);
/* Diagnostic only: enable != 0 makes the persistent kernel write 100 MHz time stamps (sampler: 4 words per
 * block at [4u..4u+3]; streamer 0: 2 words per block from word 2^20); out/n copies the first n words back. */
int32_t ngp_debug_stamps(ngp_handle *h, int32_t enable, uint64_t *out, int64_t n);

/* Marker panel, N individuals x P SNPs, column-major with leading dimension ld >= N
 * (replaces M[set][:data] + the M[set][:Mp] copy, src/prepMatVec.jl:116-131, src/mme.jl:305-311).
 * centre != 0: subtract the column mean first (src/prepMatVec.jl:129).  Stored as fp32, re-tiled. */
int32_t ngp_set_panel_f64(ngp_handle *h, const double *M, int64_t N, int64_t P, int64_t ld, int32_t centre);
int32_t ngp_set_panel_f32(ngp_handle *h, const float *M, int64_t N, int64_t P, int64_t ld, int32_t centre);
/* The same panel in column ranges: the reference keeps ONE Float64 matrix per marker set (M[set].data, src/mme.jl:296-311) and the
 * sweep wants ONE panel with the sets side by side -- so the sets are handed over one after another, no concatenated copy on the
 * host.  ngp_begin_panel fixes N x P (layout, allocation; a new model like every ngp_set_panel_*), ngp_panel_columns_* writes
 * columns [col0, col0 + ncol) from a column-major matrix (any order, any boundaries; columns never written stay zero columns),
 * ngp_end_panel builds mpm and the Gram window.  Conversion, centring and tiling run on the device from 256 MiB staging chunks
 * (ngp_set_panel_f64 / _f32 are these three calls).  Nothing else may be called on the handle between begin and end. */
int32_t ngp_begin_panel(ngp_handle *h, int64_t N, int64_t P);
int32_t ngp_panel_columns_f64(ngp_handle *h, int64_t col0, const double *M, int64_t ncol, int64_t ld, int32_t centre);
int32_t ngp_panel_columns_f32(ngp_handle *h, int64_t col0, const float *M, int64_t ncol, int64_t ld, int32_t centre);
/* ... and from one byte per genotype (codes 0..255; the tiles of ngp_set_panel_u8 bit for bit): the only form the compact storage
 * (ngp_set_storage) takes, and a quarter of the host footprint and PCIe transfer for the fp32 tiles. */
int32_t ngp_panel_columns_u8(ngp_handle *h, int64_t col0, const uint8_t *G, int64_t ncol, int64_t ld, int32_t centre);
int32_t ngp_end_panel(ngp_handle *h);
/* Same panel from one byte per genotype (0/1/2 allele counts, or any value 0..255): a quarter of the fp32 host footprint and
 * of the PCIe transfer; centring and the fp32 conversion happen on the device and give bit for bit the tiles of
 * ngp_set_panel_f64 on the same values (integer column sum / N). */
int32_t ngp_set_panel_u8(ngp_handle *h, const uint8_t *G, int64_t N, int64_t P, int64_t ld, int32_t centre);
/* Binary panel file -- replaces the text genotype file the reference parses into a Float64 matrix (CSV.read + Matrix,
 * src/prepMatVec.jl:116-131: hours at 50k x 600k) by a header and the genotype codes, column after column:
 *   bytes 0-7 "NGPPNL01" | int64 N | int64 P | int32 bits (8 or 2) | int32 0 | P columns of N bytes (bits 8) or of
 *   ceil(N/4) bytes, individual i in bits 2(i mod 4)..2(i mod 4)+1 of byte i/4 (bits 2: allele counts 0, 1, 2; code 3 refused).
 * ngp_write_panel_file writes one (no handle, host only; 0 or NGP_ERR_ARG), ngp_read_panel_header reads N, P, bits,
 * ngp_load_panel_file streams the file through a pinned buffer in chunks of whole column blocks straight into the tiles of the
 * handle's storage (fp32 centred tiles or, with NGP_STORAGE_U8, the bytes themselves): same tiles, bit for bit, as
 * ngp_set_panel_u8 on the same codes. */
int32_t ngp_write_panel_file(const char *path, const uint8_t *G, int64_t N, int64_t P, int64_t ld, int32_t bits);
int32_t ngp_read_panel_header(const char *path, int64_t *N, int64_t *P, int32_t *bits);
int32_t ngp_load_panel_file(ngp_handle *h, const char *path, int32_t centre);
/* Synthetic panel generated on the device: g_ij ~ Binomial(2,p_j), p_j ~ U(maf_lo,maf_hi). */
int32_t ngp_generate_panel(ngp_handle *h, int64_t N, int64_t P, double maf_lo, double maf_hi, uint64_t panel_seed);
/* Device tiling chosen for the panel: rows per shard R, shards S, 64-SNP blocks NBLK (the blocked
 * oracle needs R and S to reproduce the reduction tree). */
int32_t ngp_get_layout(ngp_handle *h, int64_t *R, int64_t *S, int64_t *nblk);
/* Wall time of the last ngp_generate_panel, in its three parts (milliseconds): device allocation with its zeroing, tile generation,
 * Gram window -- the set-up of a 120 GB panel is dominated by the first, which varies from box to box. */
int32_t ngp_get_setup_timing(ngp_handle *h, double *alloc_ms, double *tiles_ms, double *gram_ms);
/* x'x per SNP (M[set][:mpm], src/mme.jl:305-307); out has P entries. */
int32_t ngp_get_mpm(ngp_handle *h, double *out, int64_t P);
/* One 64x64 Gram block X_t'X_t (row-major); parity probe. */
int32_t ngp_get_gram(ngp_handle *h, int64_t t, double *out);
/* out = X * beta (N entries), X the stored fp32 panel: used to simulate phenotypes and to check
 * the invariant ycorr = y - 1b - X beta. */
int32_t ngp_xbeta(ngp_handle *h, const double *beta, int64_t P, double *out, int64_t N);

/* A marker set = columns [col0, col0+ncol) with its prior (src/mme.jl:324-361, 492-520).
 * method NGP_METHOD_*; df, scale as computed at src/mme.jl:493,501; regions are 0-based
 * [reg_start[r], reg_stop[r]) relative to the set (M[set][:regionArray], src/mme.jl:335-358);
 * varBeta0 holds nreg initial variances (src/mme.jl:516); BayesB needs nreg == ncol (one region
 * per locus, src/mme.jl:356), pi0 = prior inclusion probability, estPi (src/mme.jl:359); BayesC has nreg == 1 (one
 * variance for the whole set, src/functions.jl:205,231) and, like the reference, ignores rhs0 (src/functions.jl:220);
 * lhs0/rhs0 (ncol each or NULL) are the summary-statistics terms (src/mme.jl:314-322). */
int32_t ngp_add_marker_set(ngp_handle *h, int64_t col0, int64_t ncol, int32_t method, double df, double scale,
                           const int64_t *reg_start, const int64_t *reg_stop, int64_t nreg, const double *varBeta0,
                           double pi0, int32_t estPi, const double *lhs0, const double *rhs0, int32_t *set_id);

/* BayesR marker set (set-up src/mme.jl:374-383): ONE variance for the set (varBeta0, nVarCov = 1), K = 2..16 variance classes with
 * multipliers vClass[v] of that variance (multiplier 0: the effect is exactly 0) and class probabilities pi[v] (sum 1);
 * estPi: pi ~ Dirichlet(nLoci + 1) after every sweep (src/functions.jl:284-288).  delta then holds the CLASS of a locus,
 * counted from 1 as the reference writes it (src/functions.jl:262).  The reference's class search compares cumulative
 * probabilities with a FRESH uniform per comparison (src/functions.jl:261); that is reproduced, one keyed uniform per (locus,
 * comparison).  ngp_get_state / ngp_get_posterior_sums report [0.5, 0.5]-style placeholders in piHat / sum_pi for such a set;
 * its K probabilities are read and restored with ngp_get_class_state / ngp_set_class_state, and travel in the packed
 * posterior (ngp_posterior_len) and in snapshots. */
int32_t ngp_add_marker_set_r(ngp_handle *h, int64_t col0, int64_t ncol, double df, double scale, double varBeta0, const double *vClass,
                             const double *pi, int32_t K, int32_t estPi, const double *lhs0, const double *rhs0, int32_t *set_id);
int32_t ngp_get_class_state(ngp_handle *h, int32_t set_id, double *piHat, double *sum_pi, int64_t *K);
int32_t ngp_set_class_state(ngp_handle *h, int32_t set_id, const double *piHat, const double *sum_pi, int64_t K);

/* Correlated marker sets -- sampleBayesPR!(::Tuple) (src/functions.jl:140-154), sampleVarCovBetaPR (:513-516), set-up
 * src/mme.jl:448-489: k = 1..4 sets (e.g. breeds) share nloc loci; every locus draws its k effects jointly, beta_l ~ MvNormal(inv(LHS)
 * RHS, inv(LHS)) with LHS = X_l'X_l / varE + inv(varBeta_r), and every region r draws its k x k variance matrix
 * varBeta_r ~ InverseWishart(df + n_r, scale + B_r'B_r).  df = 3 + k and scale = v (df - k - 1) (k x k, row-major) are the caller's
 * (src/mme.jl:493, 501); varBeta0 = v, ONE k x k matrix every region starts from (src/mme.jl:516); regions are ranges of LOCI.
 * Panel layout: the k columns of a locus are adjacent -- component m of locus l is panel column col0 + 64 (l / Lb) + k (l % Lb) + m,
 * Lb = floor(64 / k), col0 a multiple of 64 (a locus never straddles a 64-column block; for k = 3 column 63 of every block of the
 * set is unused and must hold zeros).  In ngp_get_state / ngp_get_posterior_sums the set contributes nreg k x k matrices
 * (row-major) to varBeta, and its effects sit at their panel columns.  k = 1 is the Symbol method of BayesPR, bit for bit (with
 * scale = the Symbol path's scale * df).  (In the reference snapshot this method is unreachable -- SURVEY.md fact 6 -- so parity is
 * against the specification in oracle/.)  MvNormal / InverseWishart: Cholesky and Bartlett constructions on the keyed draws. */
int32_t ngp_add_marker_set_tuple(ngp_handle *h, int64_t col0, int64_t nloc, int32_t k, double df, const double *scale,
                                 const int64_t *reg_start, const int64_t *reg_stop, int64_t nreg, const double *varBeta0, int32_t *set_id);

/* A fixed-effect set beyond the intercept (X[xSet] of src/prepMatVec.jl:150-165; set-up src/mme.jl:120-152): the columns of one
 * model term or of one `blockThese` group, N x ncol column-major (ncol <= 64), sampled after the intercept in the order the
 * sets are added (the `keys(X)` order of src/samplers.jl:39; to put the intercept elsewhere switch it off and add a column of
 * ones).  One column: sampleX! (src/functions.jl:41-47) with summary-statistics terms lhs0 / rhs0 (NULL = 0); several columns:
 * sampleb! (src/functions.jl:22-36), Gauss-Seidel over X'X + min|diag| / 10000 (src/mme.jl:149-152).  ngp_get_fixed returns the
 * current effects and their posterior sums (all columns of all sets, in order); ngp_set_fixed restores them (resume). */
int32_t ngp_add_fixed_set(ngp_handle *h, const double *X, int64_t N, int64_t ncol, int64_t ld, const double *lhs0, const double *rhs0,
                          int32_t *set_id);
int32_t ngp_get_fixed(ngp_handle *h, double *b, double *sum_b, int64_t *ncols_total);
int32_t ngp_set_fixed(ngp_handle *h, const double *b, const double *sum_b, int64_t ncols_total);

/* Phenotypes; resets the chain: ycorr = y (src/mme.jl:57), b = 0, beta = 0, delta = 1, iter = 0, every variance and pi back to
 * the values given to ngp_add_marker_set (src/mme.jl:351-360, 516), all posterior sums and nKept zero. */
int32_t ngp_set_y(ngp_handle *h, const double *y, int64_t N);
/* E.df, E.scale (src/mme.jl:87-94). */
int32_t ngp_set_residual_prior(ngp_handle *h, double df, double scale);
/* Whether the model has the intercept column (src/functions.jl:41-47); default on. */
int32_t ngp_set_intercept(ngp_handle *h, int32_t on);
/* chainLength, burnIn, outputFreq of runSampler! (src/samplers.jl:23-26): iterations
 * burnIn+thin : thin : chainLength are accumulated into the posterior sums. */
int32_t ngp_set_schedule(ngp_handle *h, int64_t chainLength, int64_t burnIn, int64_t thin);

/* Coarse seam: advance the chain by niter full iterations on the device
 * (varE -> intercept -> every marker set -> variance / pi draws; src/samplers.jl:29-55). */
int32_t ngp_run(ngp_handle *h, int64_t niter);

/* Chain state after the last iteration.  Any pointer may be NULL.  ycorr: N; beta, delta: P;
 * varBeta: sum of nreg over sets; piHat: 2 per set ([1-pi, pi], src/mme.jl:360). */
int32_t ngp_get_state(ngp_handle *h, double *ycorr, double *beta, int64_t *delta, double *varBeta, double *piHat, double *varE,
                      double *b, int64_t *iter);
/* Resume support: overwrite the chain state (same shapes as ngp_get_state). */
int32_t ngp_set_state(ngp_handle *h, const double *ycorr, const double *beta, const int64_t *delta, const double *varBeta,
                      const double *piHat, double varE, double b, int64_t iter);
/* varE and intercept of each iteration of the last ngp_run (n <= niter entries). */
int32_t ngp_get_trace(ngp_handle *h, double *varE, double *b, int64_t n);
/* Sums over kept iterations (posterior mean = sum / nKept; replaces summaryMCMC, src/misc.jl:241-244). */
int32_t ngp_get_posterior_sums(ngp_handle *h, double *sum_beta, double *sum_beta2, double *sum_delta, double *sum_varBeta,
                               double *sum_pi, double *sum_varE, double *sum_b, int64_t *nKept);
/* Same sums packed into a DEVICE buffer [sum_beta P | sum_beta2 P | sum_delta P | sum_varBeta nvb |
 * sum_pi 2*nsets | class-probability sums of the BayesR sets, K each | sums of the fixed effects beyond the intercept, all
 * columns of all sets in order (ngp_get_fixed) | sum_varE | sum_b | nKept] so the host can all-reduce them over RCCL without a
 * PCIe round trip.  len = 3P + nvb + 2 nsets + sum K + nfixcol + 3 doubles. */
int32_t ngp_posterior_len(ngp_handle *h, int64_t *len);
int32_t ngp_export_posterior_device(ngp_handle *h, void *device_ptr, int64_t len);

/* Fine seam: one call of M[set].funct(mSet, M, beta, delta, ycorr, varE, varBeta) (src/samplers.jl:52).
 * In/out arrays are the caller's: ycorr N, beta ncol, delta ncol (out), varBeta nreg, piHat 2 (BayesB). */
int32_t ngp_sweep_set(ngp_handle *h, int32_t set_id, double varE, double *ycorr, double *beta, int64_t *delta, double *varBeta,
                      double *piHat);
/* The same call with the caller's state in DEVICE memory of the handle's device (ycorr N, beta ncol, varBeta nreg doubles; delta
 * ncol int64, may be null; piHat 2 doubles, BayesB / BayesC): device-to-device copies on the handle's stream, no PCIe round trip per
 * set and iteration for a host that keeps its state on the GPU (ROCArrays).  Values are not validated on the host: a variance that
 * is not finite poisons the chain visibly, as in ngp_run.  Returns after the stream has drained. */
int32_t ngp_sweep_set_dev(ngp_handle *h, int32_t set_id, double varE, void *d_ycorr, void *d_beta, void *d_delta, void *d_varBeta,
                          void *d_piHat);

/* Device time of the iterations of ngp_run (HIP events on the handle's stream) and the number of sweep-kernel launches,
 * both accumulated since the last call. */
int32_t ngp_get_timing(ngp_handle *h, int64_t *sweep_launches, double *iter_ms, int64_t *iters);
/* Runs ONE extra iteration with a HIP event pair around every sweep-kernel launch and returns the
 * average launch duration of the dominant (panel-streaming) kernel, its launch count and the
 * algorithmic bytes one launch streams (bench.py roofline). */
int32_t ngp_profile_iteration(ngp_handle *h, double *avg_ms, int64_t *launches, double *bytes_per_launch);

/* Test probe of the device draw layer: n first draws of streams (seed,chain,iter,kind,index0+i);
 * what: 0 uniform, 1 normal, 2 chisq(p1), 3 beta(p1,p2), 4 gamma(p1). */
int32_t ngp_draws_indexed(ngp_handle *h, uint64_t iter, uint64_t kind, uint64_t index0, int32_t what, double p1, double p2,
                          int64_t n, double *out);
/* det_log / ppnd16 evaluated on the device (bit-parity probe), n inputs -> n outputs. */
int32_t ngp_eval_math(ngp_handle *h, int32_t which, const double *in, int64_t n, double *out);

/* ---- streamer variants of the persistent sweep (before the panel is set) ----
 * 0 = automatic, 1 = phase streamer (every shard height), 2 = row-owning waves + loader wave (shards of at most 224 rows,
 * lags 3..6; the default for shards of 64 to 224 rows).  Both replace the loop of src/functions.jl:124-136; they differ in
 * the summation order of the shard partial of X_t'ycorr only: ngp_get_streamer reports the variant in force and the number of
 * GEMV chains per partial (8 or 7), which the blocked oracle needs like R, S, lag and near lags.
 * 4 / 6 = variant 2 with TWO / THREE shards per streamer workgroup (lag 3 / lag 2): what fp32 panels of more than 63,232 rows (one
 * resident wave of 256-row shards) run in automatically -- two up to 107,520 rows, three up to 156,576 on 256 CUs; the layout (R, S)
 * and every result are those of variant 2 with that layout and lag, only the grid is S / 2 or S / 3 streamers (ngp_get_streamer
 * reports 2).  Taller fp32 panels fall back to the per-block engine (mode 0) -- or take the compact storage, whose shards reach
 * 896 rows. */
int32_t ngp_set_streamer(ngp_handle *h, int32_t variant);
int32_t ngp_get_streamer(ngp_handle *h, int32_t *variant, int32_t *gemv_chains);

/* ---- panel storage (before the panel is set) ----
 * NGP_STORAGE_F32 (default): the centred panel as fp32 tiles (4 bytes per genotype).
 * NGP_STORAGE_U8 ("compact"): the genotype codes stay one byte each and the centring of src/prepMatVec.jl:129 is applied
 * analytically with the Float64 column means m_j = (sum_i g_ij) / N: x_j'ycorr = sum_i g_ij ycorr_i - m_j sum_i ycorr_i,
 * ycorr -= x_j dlt = ycorr_i - (g_ij dlt - m_j dlt), x_k'x_j = (exact integer dot product) - N m_k m_j.  A quarter of the
 * memory and of the bytes streamed per iteration, and no fp32 rounding of the panel: the chain agrees with the reference's
 * Float64 arithmetic to rounding error of the sums (the fp32 tiles deviate by ~1e-7 relative).  Persistent sweep only; shards
 * are multiples of 16 rows (up to 896 rows, N up to ~196k on one MI355X); look-ahead lags 3, 4, 6, 8, 12 (4 or 8 for shards
 * taller than 224 rows, 4 above 448 rows; a request is rounded down to the next of these); ngp_get_streamer reports variant 3, 7 GEMV chains.  Input: ngp_set_panel_u8,
 * ngp_load_panel_file, ngp_generate_panel (ngp_set_panel_f64 / _f32 are refused: they carry centred values, not codes).
 * ngp_get_storage: the storage in force and, optionally, the P column means the library subtracted (src/prepMatVec.jl:129; zeros where
 * the caller passed centre = 0): with them a host can rebuild any centred row of the panel from the genotype codes. */
/* ngp_set_max_shards: the persistent sweep normally splits the rows over every CU but the sampler's and the reducers' (one
 * streamer workgroup per CU, all co-resident).  A smaller number makes the shards taller and leaves CUs free -- for a second
 * chain on the same device (each chain's whole grid must be resident at once), or to exercise tall-shard layouts on small
 * panels.  0 = automatic.  Before the panel is set.  Chains of ONE process that share a device (one handle and one host thread
 * each) are kept apart by the library: a call that launches sweeps leases ceil(grid / 8) CUs of every XCD for its duration, and
 * a call whose grid does not fit beside the running ones waits for them to return (the chains then take turns); sweeps of other
 * processes cannot be seen -- against those the bounded waits of the kernel remain (NGP_ERR_HIP, chain to be set again). */
int32_t ngp_set_max_shards(ngp_handle *h, int32_t max_shards);
/* The largest max_shards with which `chains` chains of this handle's device are co-resident (256 CUs: 247 for one chain, 123
 * for two, 76 for three, 61 for four). */
int32_t ngp_shards_for_chains(ngp_handle *h, int32_t chains, int32_t *max_shards);
/* ---- K chains per pass over the panel ----
 * Independent chains (one handle each) that share ONE panel run their sweeps in ONE kernel launch per iteration: every streamer
 * workgroup forms X_t'[y_1 .. y_K] from each tile it reads, so the panel is streamed once for K iterations' worth of sampling;
 * every chain keeps its own sampler workgroup, reducers, hand-off rings and draws, and stays bit for bit the chain it is alone with
 * the same layout.  Set-up: the first handle sets the panel (after ngp_set_max_shards(ngp_shards_for_pass(K))), the others call
 * ngp_share_panel(h, first) in place of a panel upload -- no copy of the panel is made; then each handle gets its own marker sets,
 * y and seeds, and ngp_run_many(handles, K, niter) runs them fused (engines served: persistent sweep over fp32 tiles with shards of at
 * most 64 rows, lag 6 or 8, 2..8 chains -- e.g. 10k x 100k; shards of 64..224 rows, lag 4..6, two chains -- e.g. 50k x 600k; compact storage,
 * two or three chains; every method, Tuple and BayesR sets included; other layouts run the handles side by side as before).  Independent chains are
 * the path's own parallelism (src/samplers.jl:23: one chain per Julia task). */
int32_t ngp_share_panel(ngp_handle *h, ngp_handle *owner);
int32_t ngp_shards_for_pass(ngp_handle *h, int32_t chains, int32_t *max_shards);
#define NGP_STORAGE_F32 0
#define NGP_STORAGE_U8 1
int32_t ngp_set_storage(ngp_handle *h, int32_t storage);
int32_t ngp_get_storage(ngp_handle *h, int32_t *storage, double *means, int64_t P);

/* Diagnostic timing modes of the persistent kernel (1..6: parts of the pipeline switched off, ngp_sweep.h).  They are an
 * explicit, per-handle setting -- never read from the environment -- and while one is active ngp_run / ngp_sweep_set do
 * their launches and then return NGP_ERR_DEBUG: the chain they leave behind is invalid.  0 = off. */
int32_t ngp_debug_set_mode(ngp_handle *h, int32_t mode);
/* Tuning knob of the row-owning streamer: pacing of its loader wave, 0..4 = s_sleep units (64 clocks) after every four tile
 * requests (default 0), + 16 = count every partial before the block's barrier, + 1024 (before the panel is set) = build the Gram window
 * on the matrix cores (v_mfma_f64_16x16x4_f64) instead of the fp64 VALU kernel: the same sums in the same order, bit for bit.  Changes timing
 * only, never results. */
int32_t ngp_debug_set_knob(ngp_handle *h, int32_t knob);

/* Resume support, second half: overwrite the posterior sums (same shapes as ngp_get_posterior_sums).  With ngp_set_state a
 * resumed run then reproduces the posterior means of the uninterrupted one -- the role of the reference's append-only *Out
 * files (src/outFiles.jl:17-21, rows written at src/samplers.jl:56-104). */
int32_t ngp_set_posterior_sums(ngp_handle *h, const double *sum_beta, const double *sum_beta2, const double *sum_delta,
                               const double *sum_varBeta, const double *sum_pi, double sum_varE, double sum_b, int64_t nKept);
/* Binary snapshot of chain state + posterior sums + draw-stream identity (seed, chain) in one file (written to path.tmp, then
 * renamed).  ngp_load_snapshot needs the same model (panel, sets, y) built first and verifies N, P, variance components and
 * sets; afterwards ngp_run continues the interrupted chain bit for bit. */
int32_t ngp_save_snapshot(ngp_handle *h, const char *path);
int32_t ngp_load_snapshot(ngp_handle *h, const char *path);

/* Per-iteration traces beyond varE / b (ngp_get_trace): effects of up to 4096 chosen loci (0-based panel columns), the first
 * n_varBeta variance components and pi of every set, recorded for every iteration of the following ngp_run calls (what
 * the reference writes as rows of beta<set>Out / var<set>Out / pi<set>Out, src/samplers.jl:80-84, for these columns).
 * ngp_get_trace_ext copies the first n iterations of the last ngp_run: beta_tr[n][nloci], varBeta_tr[n][n_varBeta], pi_tr[n][nsets]. */
int32_t ngp_set_trace_loci(ngp_handle *h, const int64_t *loci, int64_t n, int64_t n_varBeta);
int32_t ngp_get_trace_ext(ngp_handle *h, double *beta_tr, double *varBeta_tr, double *pi_tr, int64_t n);

/* Posterior sums pooled over n chains (one handle per chain, src/samplers.jl:23 runs one chain per Julia task): afterwards
 * every handle holds the sums over all chains.  Handles on different devices: ONE RCCL all-reduce (fp64 sum) over xGMI, RCCL
 * loaded on first use; handles sharing a device are added on the device.  Errors are reported on hs[0]. */
int32_t ngp_allreduce_posterior(ngp_handle **hs, int32_t n);
/* niter iterations of n chains at once, one host thread per handle inside the library (what n Julia tasks calling ngp_run
 * would do).  Chains on different devices run in parallel; chains sharing a device run side by side when their grids fit it
 * together (ngp_set_max_shards: e.g. three chains of 10k x 100k on one MI355X, 801 instead of 346 iterations/s in all) and in
 * turns otherwise.  Every chain is bit for bit what it is alone.  Returns the first non-zero status (message on that handle). */
int32_t ngp_run_many(ngp_handle **hs, int32_t n, int64_t niter);

/* Kept samples to a binary file without stopping the chain -- the role of the reference's per-iteration text rows (src/samplers.jl:56-104,
 * src/outFiles.jl:17-21).  From the next ngp_run on, every kept iteration leaves one record (packed on the device, copied on a second
 * stream, written by a thread of the library); ngp_run returns when its last record is in the file.  NULL closes the file.  Layout:
 * "NGPSMP01" | int64 P, nvb, nsets, nfix, nclass, record bytes | per set int64 {method, K, col0, ncol, variance entries, tuple k} |
 * records: int64 iteration | varE | b | b_fixed[nfix] | beta[P] | varBeta[nvb] | piHat[2 nsets] | class probabilities[nclass] |
 * delta[P] (bytes, padded to 8). */
int32_t ngp_set_sample_file(ngp_handle *h, const char *path);
/* Placement census of the last persistent-sweep launch: out[b] = (XCC id + 1) << 32 | HW_REG_HW_ID of workgroup b (0: never resident),
 * n >= *grid entries.  Every launch of the persistent kernel opens with a census of its own grid (all of its workgroups wait for
 * each other, so all must be resident at once); a launch whose grid is not complete within 20 ms ends before any role has touched
 * the chain, and the call runs it again with the whole device leased (the chains of one process then take turns) -- *retries counts
 * those, *exclusive says whether this handle now always leases the whole device.  Any pointer may be NULL. */
int32_t ngp_get_census(ngp_handle *h, uint64_t *out, int64_t n, int64_t *grid, int64_t *retries, int32_t *exclusive);
/* Test hook of that fallback: the sweep of iteration `iteration` (counted as ngp_get_state's iter) closes its own census as timed
 * out, once; the call must resume it and end bit for bit where an undisturbed chain ends.  0 = off. */
int32_t ngp_debug_fail_census(ngp_handle *h, int64_t iteration);
/* Test hook of ngp_allreduce_posterior: group this handle under virtual device vdev (-1: its real device).  Handles of ONE GPU with
 * different virtual devices then take the multi-device branch (leaders, packing, unpacking); the collective itself is a sum on that
 * GPU instead of ncclAllReduce. */
int32_t ngp_debug_set_virtual_device(ngp_handle *h, int32_t vdev);
/* Test hook of the exception barrier: throws a C++ exception inside an entry point (kind 0 std::bad_alloc, 1 std::length_error,
 * 2 a non-standard one); what comes back is a negative status and a message -- never an unwind into the caller.  h may be NULL. */
int32_t ngp_debug_throw(ngp_handle *h, int32_t kind);

#ifdef __cplusplus
}
#endif
#endif
