/* sfm_hip.h — C ABI of libsfm_hip.so: MI355X (gfx950) kernels for the RANSAC essential-matrix,
 * cheirality and triangulation hot path of Bazs/structure_from_motion.
 *
 * The reference has no FFI for this path (it is pure Python); each entry point below replaces the
 * Python inner loop cited next to it (paths relative to the reference repo).  A maintainer binds them
 * with ctypes as shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer marked "dev" is device memory (hipMalloc / a torch ROCm tensor's data_ptr());
 *     "host" pointers are ordinary process memory.  All arrays are dense, row-major, float64 unless
 *     another element type is given.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work;
 *     nothing synchronises with the host.
 *   - a leading `batch` dimension runs independent image pairs (blockIdx.y); batch == 1 for one pair.
 *   - size limits of ONE call: a kernel that gives every work item a thread or wave of its own is launched
 *     with at most 2^31-1 blocks and 2^32-1 threads along x, and `batch` (grid y) at most 65535; a call
 *     beyond that returns SFM_EINVAL before anything is launched (never a silently partial result).  In
 *     hypothesis counts per call: sfm_sample_philox* 2^32-256, sfm_fit_eight_point* / sfm_sample_fit_philox
 *     2^32-64, sfm_score_sed 2^28 (16 hypotheses per 256-thread block), sfm_cheirality / sfm_triangulate
 *     2^32-64 points.  Element-wise kernels (normalise, mask, SED values, select) walk their items with
 *     grid-stride loops and have no such limit.
 *   - return value: 0 on success, a negative SFM_E* code otherwise; sfm_last_error() returns a
 *     thread-local message for the last failure.
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_OK 0
#define SFM_EINVAL (-1) /* bad argument (null pointer, negative size, n < 8, ...) */
#define SFM_EHIP (-2)   /* a HIP runtime call failed (message has the hipError string) */

/* error aggregation of reference lib/ransac/ransac.py:12-16 */
#define SFM_AGG_SUM 0
#define SFM_AGG_SQUARE 1
#define SFM_AGG_MEAN 2
#define SFM_AGG_RMS 3

/* per-hypothesis fit flags written by sfm_fit_eight_point */
#define SFM_FIT_DEGENERATE 1 /* eight_point.py:415-421 predicate: second-smallest eigenvalue <= 1e-10 */

/* Result record of sfm_select_best (one per batch entry, device memory, 40 bytes). */
typedef struct sfm_select_result {
    uint64_t key;          /* IEEE bits of best_err (monotone for err >= 0); INT64_MAX (0x7FFF...F) if no model */
    int64_t best_h;        /* global hypothesis index (h_offset + local), -1 if no model */
    double best_err;       /* aggregated inlier error of the winner (ransac.py:80-82), +inf if none */
    int64_t first_flagged; /* lowest global index with a fit flag set, INT64_MAX if none */
    int32_t n_flagged;     /* number of flagged hypotheses */
    int32_t best_cnt;      /* extra-inlier count of the winner */
} sfm_select_result;

/* Version of this interface: libsfm_hip.so reports the one it was compiled from (sfm_abi_version), the Python binding
 * and the torch op library (sfm_torch_ops_abi_version) refuse a library of another version. */
#define SFM_ABI_VERSION 12

const char* sfm_last_error(void);
int sfm_abi_version(void);

/* K-normalise matched pixel coordinates (eight_point.py:127-133, hoisted out of the H x N loop):
 * corr[i] = [(xa-cx)/fx, (ya-cy)/fy, (xb-cx)/fx, (yb-cy)/fy].
 * pix_a, pix_b: dev [count,2]; corr: dev [count,4]. */
int sfm_normalize_correspondences(const double* pix_a, const double* pix_b, int64_t count, double fx,
                                  double fy, double cx, double cy, double* corr, void* stream);

/* Counter-based hypothesis sampler (replaces the sequential random.shuffle of ransac.py:62-63 for
 * large H): S[b,h,:] = 8 distinct indices in [0,n) from Philox(seed + b*seed_stride, h_begin + h).
 * S: dev int32 [batch,h_count,8]. */
int sfm_sample_philox(uint64_t seed, uint64_t seed_stride, int64_t h_begin, int64_t h_count, int64_t n,
                      int64_t batch, int32_t* S, void* stream);

/* Same sampler with the 64-bit seed read from device memory (seed_dev[0]) at kernel run time: a captured
 * hipGraph of sample -> fit -> score -> select can be replayed with a new seed by rewriting that word. */
int sfm_sample_philox_dev(const uint64_t* seed_dev, uint64_t seed_stride, int64_t h_begin, int64_t h_count,
                          int64_t n, int64_t batch, int32_t* S, void* stream);

/* Same sampler for ONE hypothesis per batch entry whose index is read from device memory:
 * S[b,:] = sample of hypothesis h_index[b] (0..7 if h_index[b] < 0).  Lets the multi-GPU winner be
 * re-derived on every rank without a host round trip.  h_index: dev int64 [batch]; S: dev int32 [batch,8]. */
int sfm_sample_philox_at(uint64_t seed, uint64_t seed_stride, const int64_t* h_index, int64_t n,
                         int64_t batch, int32_t* S, void* stream);

/* Normalised eight-point fit of every hypothesis (epipolar_ransac.py:28-42 -> eight_point.py:136-170).
 * corr: dev [batch,n,4]; S: dev int32 [batch,h_count,8]; E: dev [batch,h_count,9] (row-major 3x3,
 * E[8] == 1); flags: dev int32 [batch,h_count]; lambda2: optional dev [batch,h_count] receiving the
 * second-smallest eigenvalue of Y^T Y (NULL to skip). */
int sfm_fit_eight_point(const double* corr, int64_t n, const int32_t* S, int64_t h_count, int64_t batch,
                        double* E, int32_t* flags, double* lambda2, void* stream);

/* sfm_sample_philox (or, with seed_dev != NULL, sfm_sample_philox_dev) and sfm_fit_eight_point in ONE launch: the fit
 * kernel draws each hypothesis' sample itself and also stores it in S.  Same S, E, flags as the two calls. */
int sfm_sample_fit_philox(uint64_t seed, const uint64_t* seed_dev, uint64_t seed_stride, int64_t h_begin,
                          const double* corr, int64_t n, int64_t h_count, int64_t batch, int32_t* S, double* E,
                          int32_t* flags, void* stream);

/* The same fit with its intermediates written out, for parity checks of the fused stages against the
 * reference's private helpers (_normalize_coords :308-338, _get_yT_y :363-375, _compute_f_est :396-427,
 * _enforce_fundamental_mat_constraints :430-446).  trace: dev [batch,h_count,sfm_fit_trace_doubles()] doubles:
 * normalised coords a [8][2] | b [8][2] | T1 {scale,cx,cy} | T2 {scale,cx,cy} | Y^T Y [9][9] |
 * eigenvalues [9] | f_est [9] | rank-2 F [9]. */
int sfm_fit_trace_doubles(void);
int sfm_fit_eight_point_traced(const double* corr, int64_t n, const int32_t* S, int64_t h_count, int64_t batch,
                               double* E, int32_t* flags, double* trace, void* stream);

/* Single-problem stages of the fit (device pointers):
 *   stage 0  Y^T Y of 8 coordinate pairs as given   in [8][4] {xa,ya,xb,yb}  out [81]
 *   stage 1  _compute_f_est                          in [81]                  out f_est[9] | eigenvalues[9] | flag
 *   stage 2  _enforce_fundamental_mat_constraints    in [9]                   out [9] */
int sfm_fit_stage(int stage, const double* in, double* out, void* stream);

/* _normalize_coords (eight_point.py:308-338) for n points.  coords: dev [n,2]; out: dev [2n+3] =
 * normalised coords [n][2] then {scale, centroid x, centroid y} of the forward transform. */
int sfm_hartley_normalize(const double* coords, int64_t n, double* out, void* stream);

/* Symmetric-epipolar-distance scoring of all n correspondences under all hypotheses
 * (ransac.py:66-82 with epipolar_ransac.py:18-25 / sed.py:7-30 as the scorer).
 * cnt[b,h] = #non-sample points with sed <= thr; s1 / s2 = sum of sed / sed^2 over the 8 sample points
 * plus those survivors.  cnt: dev int32 [batch,h_count]; s1, s2: dev [batch,h_count].
 * workspace: dev scratch of at least sfm_score_workspace_bytes(n, h_count, batch) bytes, 16-byte aligned; it
 * enables the two-tier kernels (a conservative reject filter + exact fp64 evaluation of the survivors, hypotheses
 * processed longest-first; identical counts and inlier decisions — the exact tier decides with sed.py's own operation sequence
 * wherever a cheaper form of the same fp64 value is not clear of the threshold by 1e-13 — sums in a fixed order, each summand
 * within 1.2e-15 of the all-fp64 kernel's): the fp32 VALU filter, and for a
 * single pair of at least 4000 points, 2048 hypotheses and 3.5e8 evaluations (at most 4 194 304 points) the kernel with the
 * filter on the fp16 / bf16 matrix pipe and a lane-per-hypothesis exact tier (sfm_score_options.kernel forces it on / off).
 * Large single-pair launches are cut into ranges of the points whose partial results are added in range order.  NULL selects
 * the all-fp64 kernel.
 * Footprint (ABI 11: sized by what the call will launch; until then every region was reserved whether or not a launch could
 * pick it): always 16 bytes per point (fp32 points of the VALU filter; the matrix-pipe kernel keeps its partial maxima there),
 * 16 KiB of counters per pair and 4 bytes per hypothesis (scoring order); for one pair 4 more per hypothesis (state of a fused
 * pass's selection); where the launch is cut into k ranges of the points 20 k bytes per hypothesis (their partials); where the
 * matrix-pipe kernel runs its tables — 96 bytes per point, 96 + 20 bytes per hypothesis — and, for a single pair whose cost
 * pre-pass hands its filter results to the scoring launch, 512 bytes per hypothesis.  50 000 x 100 000 (matrix-pipe kernel, 8
 * ranges): 86 MB; 256 pairs x 10 000 x 2 000: 444 MB with the matrix-pipe kernel (9 ranges), 47 MB with the VALU filter
 * (521 MB either way until ABI 10).
 * sfm_score_workspace_bytes_ex sizes for a call made with `options` (NULL = the process-wide defaults, what the plain form
 * uses): options that pick another kernel or more ranges than the workspace was sized for make the call return SFM_EINVAL —
 * never a silent overrun.  -1: negative size or bad options. */
int64_t sfm_score_workspace_bytes(int64_t n, int64_t h_count, int64_t batch);   /* (_ex: below, behind sfm_score_options) */
int sfm_score_sed(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                  int64_t batch, double thr, int32_t* cnt, double* s1, double* s2, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* Launch options of the two-tier scoring kernels.  They change HOW a call is launched, never what it returns: counts and
 * decisions are identical under every setting, the sums differ in their last bits between kernels / numbers of ranges
 * (summation order; the two-tier kernels' summands within 1.2e-15 of the all-fp64 kernel's) and are bit-identical from run to
 * run under one setting.  The library does not read the process
 * environment: a call carries its options (sfm_score_sed_ex; NULL = the process-wide defaults) and the embedding application
 * sets the defaults once (sfm_score_set_default_options; the Python package translates SFM_SCORE_MATRIX / _HPW / _SPLIT /
 * _ORDER / _ONE_SIDED / _XCD / _SYNC / _PERSISTENT there when it is imported).  Safe to call from several threads: a default set is
 * replaced as a whole. */
#define SFM_SCORE_KERNEL_AUTO 0     /* the size rule described at sfm_score_sed */
#define SFM_SCORE_KERNEL_FILTERED 1 /* fp32 VALU filter */
#define SFM_SCORE_KERNEL_MATRIX 2   /* fp16 / bf16 matrix-pipe filter, where it applies (<= 4 194 304 points per pair), else FILTERED */
typedef struct sfm_score_options {
    int32_t kernel;        /* SFM_SCORE_KERNEL_AUTO / _FILTERED / _MATRIX */
    int32_t hyps_per_wave; /* VALU-filter kernel: 0 = by launch size, or 1 / 2 / 4 */
    int32_t split;         /* ranges of the points of a single pair: -1 = by launch size, 0 = none, k = k ranges (at most 16) */
    int32_t order;         /* heaviest-first processing order: -1 = by launch size, 0 = index order, 1 = on */
    int32_t one_sided;     /* VALU filter: -1 / 1 = one-sided test r^2 / dB (default), 0 = two-sided (ablation) */
    int32_t xcd_map;       /* batches: -1 / 1 = all blocks of a pair on one XCD (default), 0 = plain (block, pair) grid */
    int32_t block_sync;    /* sfm_ransac_pass_small: -1 = by size, k = block barrier every k loop iterations, 0 = never */
    int32_t persistent;    /* matrix-pipe kernel, single pair: 1 = persistent waves taking (hypothesis group, range) items from
                              per-XCD counters; -1 / 0 = one block per four items, placed by the hardware dispatcher (default:
                              measured equal at 5e9 evaluations and faster below — the 4096 first tickets cost ~45 us) */
    /* Measurement hook of THIS call (ABI 11; until then process-global state behind sfm_score_set_timing_events): hipEvent_t
     * handles (either may be NULL) recorded on the launch stream immediately before / after the scoring kernel itself — not the
     * workspace preparation or the ordering pre-pass — so that a benchmark times exactly the kernel a profiler reports.  Never
     * part of the process-wide defaults (sfm_score_set_default_options clears them). */
    void* timing_before;
    void* timing_after;
} sfm_score_options;
#define SFM_SCORE_OPTIONS_DEFAULT {SFM_SCORE_KERNEL_AUTO, 0, -1, -1, -1, -1, -1, -1, 0, 0}
int sfm_score_sed_ex(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                     int64_t batch, double thr, int32_t* cnt, double* s1, double* s2, void* workspace,
                     int64_t workspace_bytes, void* stream, const sfm_score_options* options);
int64_t sfm_score_workspace_bytes_ex(int64_t n, int64_t h_count, int64_t batch, const sfm_score_options* options);
/* Process-wide defaults for calls without options (sfm_score_sed, sfm_ransac_pass_small / _large and sfm_score_sed_ex with
 * options == NULL); NULL restores SFM_SCORE_OPTIONS_DEFAULT.  _get copies the current set out. */
int sfm_score_set_default_options(const sfm_score_options* options);
int sfm_score_get_default_options(sfm_score_options* out);
/* Which kernel a call would launch for these sizes with a workspace: SFM_SCORE_KERNEL_FILTERED or SFM_SCORE_KERNEL_MATRIX;
 * negative sizes or bad options: -1.  (For reports and tests: the results do not depend on it.)  The plain form uses the
 * process-wide defaults. */
int sfm_score_kernel_choice(int64_t n, int64_t h_count, int64_t batch);
int sfm_score_kernel_choice_ex(int64_t n, int64_t h_count, int64_t batch, const sfm_score_options* options);

/* One whole RANSAC pass of a SMALL problem (one image pair, 8 <= n <= 8192, h_count <= 32768) in THREE lean launches
 * instead of the five of sfm_sample_fit_philox / sfm_fit_eight_point -> sfm_score_sed -> sfm_select_best -> sfm_inlier_mask, and the same
 * outputs (S, E, flags, cnt, s1, s2, result, mask) bit for bit:
 *   launch 1  the eight-point fits — use_philox != 0: samples drawn in the kernel from (seed or *seed_dev, h_begin + h)
 *             and stored in S; use_philox == 0: the caller's table in S — and, in spare blocks of the same launch, the
 *             preparation of the scoring workspace (no launch of its own);
 *   launch 2  SED scoring;
 *   launch 3  the selection of ransac.py:75-86 spread over up to 32 blocks, folded by the block that arrives last
 *             (a single block walking all hypotheses is latency-bound: 14 us at 10 000 hypotheses); when a mask is
 *             wanted the same launch carries ceil(n / 256) more blocks that wait for the published record and write
 *             the winner's inlier mask.
 * A small pass is a chain of dependent, latency-bound launches: what shortens it is fewer and leaner ones.
 * h_offset as in sfm_select_best (the record carries global indices = local + h_offset); the mask is always the
 * winner's, whatever h_offset: the kernels index E and S with the local winner.
 * workspace: sfm_score_workspace_bytes_ex(n, h_count, 1, options) bytes, 16-byte aligned.  options: launch options of the
 * scoring launch (NULL = the process-wide defaults). */
int sfm_ransac_pass_small(uint64_t seed, const uint64_t* seed_dev, int use_philox, int64_t h_begin, const double* corr,
                          int64_t n, int64_t h_count, double thr, double min_extra, int aggregation, int64_t h_offset,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2,
                          sfm_select_result* result, uint8_t* mask, void* workspace, int64_t workspace_bytes,
                          void* stream, const sfm_score_options* options);

/* Diagnostic of the matrix-pipe reject filter (tests measure the margin of its error bound with it; not on the product
 * path): prepares the operand tables of one pair exactly as sfm_score_sed does and evaluates tier 1 — the three 16-bit
 * matrix instructions — for EVERY (hypothesis, point), writing the raw fp32 accumulators instead of deciding on them.
 * n <= 4 194 304.  workspace: sfm_score_workspace_bytes_ex(n, h_count, 1, {kernel = SFM_SCORE_KERNEL_MATRIX, split = 0}) bytes.
 * With n_pad = 32 * ceil(n / 32):
 *   r_out  dev float [h_count, n_pad]   r'' = the scaled bilinear form c b^T E a s_p s_h as the matrix unit accumulated it
 *   d_out  dev float [h_count, n_pad]   the accumulated upper bound of (dA + dB) / 4 s_p^2 s_h^2 + slack (a point is rejected
 *                                       iff fma(-r'', r'', d) is negative); columns >= n are padding rows of zeros + slack
 *   bound_out dev float [h_count, 8]    {delta'' (bound on |r''_mfma - r''|), slack'' = delta''^2 (1 + k) / k, s_h, armed (1 / 0),
 *                                       s_p, the bf16-rounding term of the denominator chain in units of s_h^2, the constant
 *                                       slot as stored, 0} */
int sfm_debug_matrix_filter(const double* corr, int64_t n, const double* E, int64_t h_count, double thr, void* workspace,
                            int64_t workspace_bytes, float* r_out, float* d_out, float* bound_out, void* stream);

/* The same for a LARGE problem (one image pair, any n and h_count the separate calls take): sfm_sample_fit_philox /
 * sfm_fit_eight_point -> sfm_score_sed -> sfm_select_best -> sfm_inlier_mask in SEVEN launches instead of eighteen where the
 * matrix-pipe scoring kernel applies, with the same outputs (S, E, flags, cnt, result, mask bit for bit; s1 / s2 bit for
 * bit too: the ranges of the scoring launch are added in the same order):
 *   1  partial maxima of the points + every zeroing the pass needs;  2  the eight-point fits — every lane also writes its
 *   hypothesis' rows of the scoring kernel's operand table and its sample correction (E and the sample are in registers there),
 *   blocks behind the fit blocks write the point operand table;  3  cost pre-pass;  4  class histogram;  5  scan + scatter (the
 *   heaviest-first order);  6  the scoring kernel;  7  fold of the point ranges + selection (ransac.py:75-86) over up to 256
 *   blocks + the blocks that write the winner's inlier mask.
 * Sizes for which sfm_score_sed picks another kernel run the fit, that call's launches and launch 7.  Arguments as for
 * sfm_ransac_pass_small. */
int sfm_ransac_pass_large(uint64_t seed, const uint64_t* seed_dev, int use_philox, int64_t h_begin, const double* corr,
                          int64_t n, int64_t h_count, double thr, double min_extra, int aggregation, int64_t h_offset,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2,
                          sfm_select_result* result, uint8_t* mask, void* workspace, int64_t workspace_bytes,
                          void* stream, const sfm_score_options* options);

/* The same for a BATCH of independent image pairs (BASELINE configs[4]: 256 pairs x 10 000 x 2 000; every array with a leading
 * pair dimension as in the separate calls, pair b sampling from Philox(seed + b * seed_stride, h_begin + h)): where the batch takes
 * the matrix-pipe scoring kernel (sfm_score_kernel_choice(n, h_count, batch)) —
 *   1  partial maxima of every pair's points + every zeroing;  2  the pairs' point operand tables;  3  the eight-point fits,
 *   whose lanes also write their hypothesis' operand rows and sample correction;  4  cost pre-pass;  5  class histogram;
 *   6  scan + scatter;  7  the scoring kernel;  8  per pair one block: fold of the point ranges + selection + inlier mask
 * — instead of the fifteen launches of sfm_sample_fit_philox -> sfm_score_sed -> sfm_select_best -> sfm_inlier_mask, with the same
 * outputs bit for bit.  Other sizes: the fit, that scoring call's own launches, and launch 8.  result: dev [batch]; mask: dev
 * uint8 [batch, n] or NULL. */
int sfm_ransac_pass_batch(uint64_t seed, const uint64_t* seed_dev, uint64_t seed_stride, int use_philox, int64_t h_begin,
                          const double* corr, int64_t n, int64_t h_count, int64_t batch, double thr, double min_extra, int aggregation,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2, sfm_select_result* result,
                          uint8_t* mask, void* workspace, int64_t workspace_bytes, void* stream, const sfm_score_options* options);

/* Model selection (ransac.py:75-86): lowest aggregated error among hypotheses with
 * cnt >= min_extra, strict <, earliest index wins, NaN/inf never win.  result: dev [batch]. */
int sfm_select_best(const int32_t* cnt, const double* s1, const double* s2, const int32_t* flags,
                    int64_t h_count, int64_t batch, double min_extra, int aggregation, int64_t h_offset,
                    sfm_select_result* result, void* stream);

/* Cross-shard selection for hypothesis-sharded RANSAC (one rank per GPU).  Each rank runs
 * sfm_select_best over its own block of hypotheses with h_offset = the block's first global index, the ranks
 * all-gather their [batch] records (40 bytes each) into gathered[world][batch], and this folds them: lowest error
 * key, then lowest global index (the sequential "strictly lower error replaces the incumbent" rule of
 * ransac.py:83-86 for any partition); first_flagged = minimum, n_flagged = sum (saturating), so every rank takes
 * the same decision about degenerate samples (eight_point.py:415-421 raised through ransac.py:65).
 * Outputs (each optional, dev): global [batch] = the folded records; best_h [batch] = the winners' global indices
 * (-1 if none; the form sfm_sample_philox_at reads); single [batch] = the folded records with best_h replaced by
 * 0 / -1, i.e. addressed at a one-hypothesis E / S array holding the locally re-derived winner (for sfm_inlier_mask).
 * The _host variant takes host pointers and runs on the calling thread (the fold is 40 bytes per rank: it exists so
 * that the N > 1 selection logic is testable over a CPU process group). */
int sfm_fold_select_records(const sfm_select_result* gathered, int64_t world, int64_t batch,
                            sfm_select_result* global, int64_t* best_h, sfm_select_result* single, void* stream);
int sfm_fold_select_records_host(const sfm_select_result* gathered, int64_t world, int64_t batch,
                                 sfm_select_result* global, int64_t* best_h, sfm_select_result* single);

/* Inlier mask of the selected model: mask[b,i] = 1 if point i is a non-sample survivor, 2 if it is one of
 * the 8 sample points, 0 otherwise.  `result` as written by sfm_select_best with h_offset 0 (or with
 * best_h rewritten to a local index); entries with best_h < 0 leave their mask zeroed.
 * mask: dev uint8 [batch,n]. */
int sfm_inlier_mask(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                    int64_t batch, const sfm_select_result* result, double thr, uint8_t* mask,
                    void* stream);

/* SED of n correspondences under one E (sed.py:7-30).  E: dev [9]; out: dev [n]. */
int sfm_sed_values(const double* corr, int64_t n, const double* E, double* out, void* stream);

/* Cheirality test of m normalised correspondences under `poses` candidate poses
 * (eight_point.py:449-488).  pose_rt: dev [poses,12] rows [R(9) | t(3)]; pass: dev uint8 [poses,m]. */
int sfm_cheirality(const double* corr, int64_t m, const double* pose_rt, int64_t poses,
                   double distance_threshold, uint8_t* pass, void* stream);

/* Linear (DLT) triangulation of m correspondences (triangulation.py:9-39).  P1, P2: dev [12] each
 * (rows 0..2 of the camera matrices, 4 columns); X: dev [m,3]. */
int sfm_triangulate(const double* corr, int64_t m, const double* P1, const double* P2, double* X,
                    void* stream);

/* Decompose an essential matrix into the four candidate poses in the order (R1,t),(R1,-t),(R2,t),(R2,-t)
 * (eight_point.py:245-280, 210-212).  E: dev [batch,9]; pose_rt: dev [batch,4,12];
 * status: dev int32 [batch], 0 ok, 1 = smallest singular value not ~0 (eight_point.py:268-271). */
int sfm_decompose_essential(const double* E, int64_t batch, double* pose_rt, int32_t* status,
                            void* stream);

/* ---- local optimisation of the winner (an extension, the reference has no such step) ---- */

typedef struct sfm_refine_info {
    double error;     /* aggregated error of the model left in E_out (over its `count` inliers) */
    int32_t count;    /* inliers of that model (for an unrefined model: non-zero entries of mask_in) */
    int32_t accepted; /* refits that were accepted (0 = E_out is E_in) */
} sfm_refine_info;

/* Per image pair, up to `iterations` times: refit E on all current inliers with the N-point form of the reference's
 * normalised eight-point pipeline (eight_point.py:308-446 and :163-166 applied to M >= 8 pairs), re-score all n
 * correspondences (sed.py:7-30, sed <= thr), and keep the refit iff it has more inliers, or as many and a lower
 * aggregated error (ransac.py:96-108 over the inliers); stop at the first refit that is not kept, is degenerate
 * (eight_point.py:415-421) or has fewer than 8 inliers to work from.
 * corr: dev [batch,n,4]; E_in: dev [batch,9]; mask_in: dev uint8 [batch,n] (non-zero = inlier, as written by
 * sfm_inlier_mask); err_in: dev [batch] aggregated error of E_in (sfm_select_result.best_err);
 * E_out: dev [batch,9]; mask_out: dev uint8 [batch,n] (1 = inlier; must not alias mask_in); info: dev [batch]. */
int sfm_refine_inliers(const double* corr, int64_t n, int64_t batch, const double* E_in, const uint8_t* mask_in,
                       const double* err_in, double thr, int aggregation, int iterations, double* E_out,
                       uint8_t* mask_out, sfm_refine_info* info, void* stream);

/* ---- batched, device-resident pose selection + triangulation (chains after sfm_inlier_mask) ---- */

/* Cheirality of every correspondence of every pair under its 4 candidate poses.  corr: dev [batch,n,4]
 * (K-normalised); pose_rt: dev [batch,4,12] as written by sfm_decompose_essential; mask: dev uint8
 * [batch,n] inlier mask (points with 0 are reported as failing) or NULL; pass: dev uint8 [batch,4,n]. */
int sfm_cheirality_batched(const double* corr, int64_t n, int64_t batch, const double* pose_rt,
                           const uint8_t* mask, double distance_threshold, uint8_t* pass, void* stream);

/* Pose vote of eight_point.py:213-237 per pair: votes[b,p] = #passing correspondences except the one
 * with index skip_index[b] (the reference does not count position 0 of its list: np.count_nonzero of the
 * index array); best[b] = first maximum, -1 if all votes are 0.  skip_index: dev int32 [batch] or NULL;
 * votes: dev int32 [batch,4]; best: dev int32 [batch]. */
int sfm_pose_vote(const uint8_t* pass, int64_t n, int64_t batch, const int32_t* skip_index, int32_t* votes,
                  int32_t* best, void* stream);

/* Triangulate (triangulation.py:42-62) the correspondences that pass under the chosen pose:
 * P1 = [K|0], P2 = [K|0][R t; 0 1], pixel coordinates.  pix_a, pix_b: dev [batch,n,2]; K: HOST [9]
 * row-major intrinsics; X: dev [batch,n,3] (zeros where valid == 0); valid: dev uint8 [batch,n]. */
int sfm_triangulate_selected(const double* pix_a, const double* pix_b, int64_t n, int64_t batch,
                             const double* K, const double* pose_rt, const int32_t* best,
                             const uint8_t* pass, double* X, uint8_t* valid, void* stream);

/* ---- brute-force window matching (reference lib/feature_matching: matching.py, ncc.py, ssd.py, util.py) ---- */
#define SFM_MATCH_NCC 0 /* ncc.py:7-54: 1 - normalised cross-correlation, in [0,2]; 2.0 if a window leaves the image */
#define SFM_MATCH_SSD 1 /* ssd.py:7-36: mean squared difference; +inf if a window leaves the image */
/* ssd.py:31-36 on INTEGER images: the reference subtracts and squares in the image dtype — both wrap modulo 2^bits, into the
 * signed range for signed types (uint8 images, what apps/sfm.py:222-224 produces: modulo 256) —, np.sum accumulates in
 * int64 / uint64 and the division by the window size is float64.  bits: 8, 16, 32 or 64 (the NumPy result type of
 * image_a - image_b).  The patches then hold the pixels as int64 bit patterns (SFM_PATCH_RAW64).  Bit-exact. */
#define SFM_MATCH_SSD_INT(bits, is_signed) (0x100 | ((is_signed) ? 0x80 : 0) | (bits))

/* patch modes of sfm_patch_extract (its `subtract_mean` argument) */
#define SFM_PATCH_PLAIN 0        /* float64 pixels as they are (SSD on float images) */
#define SFM_PATCH_MEAN_REMOVED 1 /* float64 pixels minus the window mean (ncc.py:33-37) */
#define SFM_PATCH_RAW64 2        /* image holds int64 pixels: 8-byte copies, no arithmetic (SFM_MATCH_SSD_INT) */

/* Windows of n features of one image (util.py:8-27).  image: dev f64 [height,width] (int64 with SFM_PATCH_RAW64);
 * feats: dev f64 [n,2] (x, y); patches: dev f64 [window_size^2][stride] k-major (stride >= n), by `subtract_mean` =
 * SFM_PATCH_PLAIN / SFM_PATCH_MEAN_REMOVED (ncc.py:33-37) / SFM_PATCH_RAW64 (int64 bit patterns in the 8-byte slots);
 * ssq: dev [n] sum of squares of the stored patch (0 with SFM_PATCH_RAW64); ok: dev uint8 [n] window inside the image. */
int sfm_patch_extract(const double* image, int64_t height, int64_t width, const double* feats, int64_t n,
                      int window_size, int subtract_mean, int64_t stride, double* patches, double* ssq,
                      uint8_t* ok, void* stream);

/* scores[a,b] for every pair of features (the score_function calls of matching.py:57-65).
 * metric SFM_MATCH_NCC needs mean-removed patches, SFM_MATCH_SSD_INT(bits, signed) raw int64 ones.  scores: dev f64 [n_a,n_b].
 * Fast path (LDS-DMA staging) when both patch arrays are 16-byte aligned with an even `stride` >= n rounded up to a
 * multiple of 128 (the padding may hold anything); any other layout works through a slower staging loop. */
int sfm_pair_scores(int metric, const double* patches_a, int64_t stride_a, const double* patches_b,
                    int64_t stride_b, const double* ssq_a, const double* ssq_b, const uint8_t* ok_a,
                    const uint8_t* ok_b, int64_t n_a, int64_t n_b, int window_elements, double* scores,
                    void* stream);

/* Per A-feature, what the reference's heapq holds after pushing its row of scores in order
 * (matching.py:60-65): best[a] / arg[a] = heap[0] score and b index (first minimum), second[a] = heap[1]
 * score (the value the ratio test of matching.py:84-97 divides by; NaN if n_b == 1).
 * best, second: dev f64 [n_a]; arg: dev int32 [n_a]. */
int sfm_match_row_summary(const double* scores, int64_t n_a, int64_t n_b, double* best, int32_t* arg,
                          double* second, void* stream);

/* sfm_pair_scores + sfm_match_row_summary in one pass that never materialises the |A| x |B| matrix: each tile of
 * scores is reduced on chip to four numbers per row, a second small kernel walks them in b order.  Same outputs,
 * bit for bit.  workspace: dev, 16-byte aligned, >= sfm_match_summary_workspace_bytes(n_a, n_b) bytes. */
int64_t sfm_match_summary_workspace_bytes(int64_t n_a, int64_t n_b);
int sfm_match_summary(int metric, const double* patches_a, int64_t stride_a, const double* patches_b,
                      int64_t stride_b, const double* ssq_a, const double* ssq_b, const uint8_t* ok_a,
                      const uint8_t* ok_b, int64_t n_a, int64_t n_b, int window_elements, void* workspace,
                      int64_t workspace_bytes, double* best, int32_t* arg, double* second, void* stream);

/* ---- Harris corner detector stencils (reference lib/harris/harris_detector.py, lib/common/correlate.py) ---- */

/* Zero-'same' cross-correlation with an odd square kernel (correlate.py:4-39).  image, out: dev f64
 * [height,width]; kernel: dev f64 [kernel_size,kernel_size]. */
int sfm_cross_correlate(const double* image, int64_t height, int64_t width, const double* kernel,
                        int kernel_size, double* out, void* stream);

/* Harris cornerness det(M) - k trace(M)^2 from block sums of the Sobel products (harris_detector.py:57-86).
 * out: dev f64 [out_height,out_width] (the reference uses height/width - round(block_size/2)); entries outside
 * range(height-block_size) x range(width-block_size) are 0; clamp_negative applies harris_detector.py:29. */
int sfm_harris_cornerness(const double* sobel_x, const double* sobel_y, int64_t height, int64_t width,
                          int block_size, double k, int clamp_negative, int64_t out_height, int64_t out_width,
                          double* out, void* stream);

/* 3x3 non-maximum suppression IN PLACE in raster order, with the reference's sequential semantics
 * (harris_detector.py:95-105): a neighbour visited earlier may already be zero.  image: dev f64 [height,width]. */
int sfm_nms_inplace(double* image, int64_t height, int64_t width, void* stream);

/* The same suppression as a parallel fixpoint (fast path): call sfm_nms_round repeatedly on a zero-initialised
 * state array (dev uint8 [height,width]; bits 0-1: 0 unknown, 1 survives, 2 suppressed; bits 4-7 of an unknown pixel:
 * which raster-earlier neighbours are larger; a zero byte: not looked at yet — the first round classifies every pixel
 * from the image, the following ones read state bytes only) until *unresolved (dev int32, zeroed by the caller
 * before each round) stays 0, then sfm_nms_finalize zeroes the suppressed pixels of `image` in place.
 * The number of rounds is the longest chain of strictly increasing raster-earlier neighbours (a handful on natural
 * images); the result is identical to sfm_nms_inplace. */
int sfm_nms_round(const double* image, uint8_t* state, int64_t height, int64_t width, int32_t* unresolved,
                  void* stream);
int sfm_nms_finalize(double* image, const uint8_t* state, int64_t height, int64_t width, void* stream);

/* (flat index, value) of every non-zero (or NaN) element of `image` [count], in no particular order: what the top-k
 * selection of harris_detector.py:32-42 needs from the suppressed cornerness image.  counter: dev int32 [1], receives
 * the number found (may exceed `capacity`; only the first `capacity` slots are written); index: dev int32 [capacity];
 * value: dev f64 [capacity]. */
int sfm_compact_nonzero(const double* image, int64_t count, int32_t capacity, int32_t* counter, int32_t* index,
                        double* value, void* stream);

/* (new, ABI 12) Of the candidates sfm_compact_nonzero left — value / index: dev [capacity], *found: dev int32, the counter it
 * wrote (read on the device: no host round trip in between; min(*found, capacity) candidates are looked at) — those that can be
 * among the m largest values: every candidate at or above the m-th largest (ties included; a NaN of either sign counts as the
 * largest value there is), plus at most the ones that share the m-th's sign, exponent and top twelve mantissa bits.  What
 * np.argsort of harris_detector.py:32-42 needs of a 1080p image is then num_corners + a few pairs instead of 220 000.
 * workspace: dev, 8192 int32 (zeroed by the call).  *counter_out receives the number kept (may exceed capacity_out; only that many slots are written);
 * fewer than m candidates: all are kept. */
int sfm_prune_top(const double* value, const int32_t* index, const int32_t* found, int32_t capacity, int32_t m, void* workspace,
                  int32_t capacity_out, int32_t* counter_out, int32_t* index_out, double* value_out, void* stream);

/* HOST: exact replay of CPython's random.shuffle as used by ransac.py:59-64.  `mt_state` is the 624-word
 * MT19937 state and `*mt_index` its position (random.getstate()[1]); both are advanced.  The cumulative
 * permutation of range(n) is shuffled `iterations` times; S_out[it,:] receives its first 8 entries.  If
 * perm_io != NULL it supplies the starting permutation (n entries) and receives the final one; if
 * snapshot_iteration >= 0 and snapshot != NULL, the permutation after that iteration is stored there. */
int sfm_pyshuffle_table(uint32_t* mt_state, int32_t* mt_index, int64_t n, int64_t iterations,
                        int32_t* S_out, int32_t* perm_io, int64_t snapshot_iteration, int32_t* snapshot);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H */
