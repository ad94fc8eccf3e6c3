/*
 * ctamd.h -- C ABI of the MI355X-native hot path for 3DeeCellTracker-style tracking.
 *
 * The reference (WenChentao/3DeeCellTracker) is pure Python and has NO native / FFI boundary:
 * its hot path calls tensorflow.keras `model.predict`, numpy/LAPACK and scikit-learn.  The
 * "binding a maintainer would add" is therefore a ctypes stub at the Python call
 * sites cited on every entry point below (paths relative to the reference tree).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / framework types;
 *   - pointers marked [dev] are device (HBM) addresses owned by the caller and kept alive by the
 *     caller until the stream has drained; pointers marked [host] are host addresses;
 *   - every launch-type function takes a hipStream_t (as void*), is asynchronous w.r.t. the host
 *     unless stated, returns 0 on success, a negative CT_E* code for argument/shape errors, or a
 *     positive hipError_t value; nothing throws;
 *   - no global mutable state: handles own their device-side weights only; all scratch memory is a
 *     caller-provided workspace whose size the *_workspace_bytes functions report;
 *   - thread-compatible: one stream per host thread, a handle may be shared for inference;
 *   - entry points that take a handle run on the handle's device (the caller's current device is restored on return);
 *     all others run on the calling thread's current device;
 *   - size limits of the kernels (CT_ESHAPE beyond them; the Python mirrors check them up front, _dev.check_match_sizes):
 *       ct_knn_features        n <= 4096 points, k <= 31 neighbours
 *       ct_greedy_match        min(m, n) <= 16384, m * n < 2^32
 *       ct_prgls_two_ref / ct_prgls_legacy / ct_solve_movements   no kernel limit (workspace ~ 5 n^2 doubles; the mirrors stop at 16384)
 *       ct_trim_mean           k <= 64 predictions
 *   - collectives: there are no ct_comm_* entry points.  The path's only inter-GPU exchange is a gather of results
 *     (centre crops, centroid sets, ensemble predictions), issued by the host layer through
 *     torch.distributed (backend "nccl" == RCCL over xGMI on ROCm) on the buffers these entry points fill
 *     (3deecelltracker_amd/parallel.py); a C wrapper around ncclAllGather would add nothing but a second RCCL
 *     communicator next to torch's.
 */
#ifndef CTAMD_H
#define CTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT_OK            0
#define CT_EINVAL       -1   /* bad argument (null pointer, non-positive size, unknown id)      */
#define CT_ESHAPE       -2   /* shape unsupported by the kernels (e.g. fewer than k+1 points)   */
#define CT_EWORKSPACE   -3   /* workspace too small                                             */
#define CT_ENOTCONV     -4   /* iterative routine hit its iteration bound (informational)       */

#define CT_ARCH_UNET3_A  0   /* reference CellTracker/unet3d.py:26-37  (160x160x16, pool 2,2,1) */
#define CT_ARCH_UNET3_B  1   /* reference CellTracker/unet3d.py:40-67  ( 96x 96x 8, pool 2,2,1) */
#define CT_ARCH_UNET3_C  2   /* reference CellTracker/unet3d.py:70-81  ( 64x 64x64, pool 2,2,2) */

typedef struct ct_unet ct_unet_t;
typedef struct ct_ffn  ct_ffn_t;
typedef void*          ct_stream_t;    /* hipStream_t */

int         ct_version(void);
const char* ct_error_string(int code);
/* Device properties the host side reports next to benchmark numbers. */
int         ct_device_info(int device, int* n_cu, size_t* hbm_bytes, char* name, size_t name_len);

/* HIP streams limited to the CU range [first_cu, first_cu + n_cu) (hipExtStreamCreateWithCUMask): lets the
 * latency-bound matching chain and the throughput-bound U-Net of different frames share a GPU without the
 * tiny kernels queueing behind resident conv workgroups.  Destroy with ct_stream_destroy.                  */
int ct_stream_create_cu_range(int device, int first_cu, int n_cu, ct_stream_t* out);
int ct_stream_destroy(ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3D U-Net  (replaces keras Model construction + `model.predict`, unet3d.py:26-98, :253)
 * ------------------------------------------------------------------------------------------
 * weights [host]: flat fp32 array in Keras layouts, in execution order:
 *   for every 3x3x3 conv block: kernel (3,3,3,Cin,Cout) | bias (Cout) | gamma | beta | moving_mean |
 *   moving_var (Cout each);  then the 1x1x1 head: kernel (C,1) | bias (1).
 * The handle keeps a device copy re-packed for the MFMA kernels with BatchNorm folded into a
 * per-channel (scale, shift) epilogue (BN follows the activation, so it cannot be folded into
 * the conv itself: unet3d.py:117-119).                                                        */
int    ct_unet_create(int arch_id, const float* weights, size_t n_floats, int device, ct_unet_t** out);
void   ct_unet_destroy(ct_unet_t* h);
size_t ct_unet_num_weights(int arch_id);
int    ct_unet_patch_shape(int arch_id, int shape_xyz[3]);
size_t ct_unet_workspace_bytes(const ct_unet_t* h, int n_patches);

/* model.predict on a batch of patches (unet3d.py:253).
 * patches_in [dev]  fp32 [n_patches][X][Y][Z]   (channel dim of size 1 elided)
 * prob_out   [dev]  fp32 [n_patches][X][Y][Z]
 * If layer_dump [dev] is non-null the outputs of every conv block of patch 0 are additionally
 * copied there in Keras NDHWC order, back to back (parity tests only; sizes via
 * ct_unet_layer_dump_floats).                                                                  */
int    ct_unet_predict_patches(ct_unet_t* h, const float* patches_in, int n_patches, float* prob_out,
                               void* workspace, size_t workspace_bytes, float* layer_dump, ct_stream_t stream);
size_t ct_unet_layer_dump_floats(int arch_id);

/* Introspection + optional per-launch timing used by bench.py's roofline measurement: when enabled,
 * every conv launch is bracketed by a hipEvent pair recorded on the launch stream; ct_unet_get_timing
 * synchronises, returns the summed milliseconds and launch counts per conv layer (layer 0 is the
 * Cin=1 VALU conv, layers >= 1 the MFMA conv with `nt` cout tiles per block) and clears the log.     */
int ct_unet_num_conv_layers(const ct_unet_t* h);
int ct_unet_layer_info(const ct_unet_t* h, int layer, int* cin, int* cout, int dims_xyz[3], int* nt);
/* Decoder convs over concat([UpSampling3D(low), skip]) (unet3d.py:92-97) fold the taps of the upsampled channels that land
 * on the same low-res voxel (12 instead of 27 per channel): returns how many input channels of `layer` are folded (0 if
 * none).  `nt` above is then 100 + nt (conv3_mfma_fold_kernel) or -9 (Cout = 8 variant; -8 = unfolded Cout = 8 kernel).
 * Convs running the split-bf16 kernels (conv3_bf16x6_kernel<nt, c8, fold>; default, CT_CONV_MATH=f32 selects the f32-input
 * MFMA kernels when the model is created) report the same codes offset by 1000 (Cout = 8: -1008 / -1009).  Levels with
 * Z <= 8 run that kernel's 8 x 8 x 8 tile instantiation (CT_CONV_Z8=0: the 4 x 8 x 16 one).                            */
int ct_unet_layer_fold_channels(const ct_unet_t* h, int layer);
/* Output extent {x0, x1, y0, y1} (voxels of the layer's level, z always whole) that `layer` computed in the last run.  The patch
 * entry point computes every layer in full; ct_unet_predict_volume keeps only the centre crop of every patch (unet3d.py:246-254),
 * so its decoder convs compute only the tiles some kept voxel depends on (CT_CONV_CROP=0: everything, as the reference does).  */
int ct_unet_layer_region(const ct_unet_t* h, int layer, int region[4]);
/* Tile {x, y, z} a workgroup of `layer` computed in the last run: 4 x 8 x 16 by default, 8 x 8 x 8 on levels with Z <= 8 (CT_CONV_Z8),
 * 4 x 10 x 16 where that covers the level with fewer columns than 4 x 8 does (20-wide levels; CT_CONV_Y10).  Diagnostics
 * (bench.py names the kernel instantiation of every layer with it).                                                             */
int ct_unet_layer_tile(const ct_unet_t* h, int layer, int tile_xyz[3]);
int ct_unet_set_timing(ct_unet_t* h, int enable);
int ct_unet_get_timing(ct_unet_t* h, float* ms_per_layer, int* launches_per_layer, int n_layers);

/* Sliding-window tiler (replaces np.pad(..., 'reflect') + the patch loop + centre-crop stitch of
 * unet3_prediction, unet3d.py:221-255).  Patch p in [p_begin, p_begin+n) enumerates
 * itertools.product(range(gx), range(gy), range(gz)).
 * vol [dev] fp32 [x][y][z];  patches [dev] fp32 [n][nx][ny][nz];  out_vol [dev] fp32 [x][y][z].    */
int ct_tile_plan(const int vol_xyz[3], const int net_xyz[3], const int shrink[3],
                 int centre[3], int grid[3]);                       /* _get_sizes_padded_im :259-279 */
int ct_tile_gather_reflect(const float* vol, const int vol_xyz[3], const int net_xyz[3], const int shrink[3],
                           int p_begin, int n, float* patches, ct_stream_t stream);
int ct_tile_scatter_center(const float* pred, const int vol_xyz[3], const int net_xyz[3], const int shrink[3],
                           int p_begin, int n, float* out_vol, ct_stream_t stream);

/* Centre crops of the patch range [p_begin, p_begin+n) as a dense slab [n][cx][cy][cz] (centre = net - 2 shrink), the unit
 * of the multi-GPU exchange (BASELINE config 3; replaces the sequential patch loop unet3d.py:246-254 across ranks): a rank
 * packs the crops it computed out of its volume, all-gathers the slabs (host layer, RCCL) and unpacks the other ranks'
 * crops into its volume.  Crop voxels beyond the volume's upper faces are packed as 0 and skipped when unpacking.          */
int ct_tile_pack_crops(const float* vol, const int vol_xyz[3], const int net_xyz[3], const int shrink[3],
                       int p_begin, int n, float* crops, ct_stream_t stream);
int ct_tile_unpack_crops(const float* crops, const int vol_xyz[3], const int net_xyz[3], const int shrink[3],
                         int p_begin, int n, float* out_vol, ct_stream_t stream);

/* Whole unet3_prediction(img, model, shrink) for the patch range [p_begin, p_begin+n) of one
 * volume (unet3d.py:203-256): gather -> network -> scatter, processed in batches that fit the
 * workspace.  Voxels of out_vol not covered by the given patch range are left untouched, so R
 * ranks can each fill their share and exchange centre crops afterwards.                        */
int ct_unet_predict_volume(ct_unet_t* h, const float* vol, const int vol_xyz[3], const int shrink[3],
                           int p_begin, int n, float* out_vol,
                           void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FFN initial matching  (ffn.py:225-327, track.py:117-178)
 * ------------------------------------------------------------------------------------------ */
/* normalize_points (ffn.py:330-374): centre, divide by 3 x std of the projection on the first principal axis
 * (= 3 sqrt(lambda_max(Xc^T Xc) / n)).  points/out_points [dev] fp64 [n][3]; para [dev] fp64 [4] = mean xyz, scale.
 * apply_para != NULL: normalise with the given parameters instead (trackerlite.py:91-93); out_points may be NULL.    */
int ct_normalize_points(const double* points, int n, const double* apply_para, double* out_points, double* para,
                        ct_stream_t stream);
int ct_denormalize_points(const double* points, int n, const double* para, double* out_points, ct_stream_t stream);

/* kNN shape-context features (ffn.py:288-304): points [dev] fp64 [n][3] -> feat [dev] fp32 [n][3k+1].
 * Needs n >= k+1 (sklearn raises otherwise) -> CT_ESHAPE.                                        */
int ct_knn_features(const double* points, int n, int k, float* feat, ct_stream_t stream);

/* weights [host] flat fp32: w1 (61,512) | bn1 gamma,beta,mean,var (512 each) | w2 (1024,512) |
 * bn2 gamma,beta,mean,var | w3 (512) | b3 (1).                                                   */
int    ct_ffn_create(const float* weights, size_t n_floats, int device, ct_ffn_t** out);
void   ct_ffn_destroy(ct_ffn_t* h);
size_t ct_ffn_num_weights(void);
size_t ct_ffn_workspace_bytes(int n_ref, int n_tgt);

/* All n_tgt x n_ref pairs through the FFN without materialising the (m*n) x 122 grid
 * (ffn.py:306-326): corr [dev] fp32 [m][n], corr[t][r] = FFN([feat_ref[r] | feat_tgt[t]]).          */
int ct_ffn_pairgrid(ct_ffn_t* h, const float* feat_ref, int n, const float* feat_tgt, int m, float* corr,
                    void* workspace, size_t workspace_bytes, ct_stream_t stream);
/* Generic `ffn_model.predict(x)` on explicit rows: x [dev] fp32 [rows][122] -> out [dev] fp32 [rows]. */
int ct_ffn_predict(ct_ffn_t* h, const float* x, int rows, float* out,
                   void* workspace, size_t workspace_bytes, ct_stream_t stream);
size_t ct_ffn_predict_workspace_bytes(int rows);

/* ------------------------------------------------------------------------------------------
 * Greedy one-to-one assignment  (trackerlite.py:242-259; legacy prior track.py:58-70)
 * ------------------------------------------------------------------------------------------
 * corr [dev] fp32 [m][n] (not modified).  Repeats: global max (first in row-major order on ties)
 * -> pair -> clear row and column; stops below `threshold` or after n pairs.
 * pairs [dev] int32 [n][2] = (ref, tgt);  n_pairs [dev] int32[1];
 * prior [dev] fp64 [m][n]: mode 0 (TrackerLite): float32(0.1/(n-1)) everywhere, 0.9 at pairs;
 *                          mode 1 (legacy):  1/n, matched rows 0.1/(n-1) with 0.9 at the pair.   */
size_t ct_greedy_workspace_bytes(int m, int n);
int ct_greedy_match(const float* corr, int m, int n, float threshold, int mode,
                    int32_t* pairs, int32_t* n_pairs, double* prior,
                    void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PR-GLS  (TrackerLite dialect trackerlite.py:262-417; legacy dialect track.py:11-114)
 * ------------------------------------------------------------------------------------------ */
size_t ct_prgls_workspace_bytes(int m, int n, int l);

/* prgls_with_two_ref (trackerlite.py:309-358); prgls_quick is the case tracked == ref.
 * prior [dev] fp64 [m][n]; tgt [dev] fp64 [m][3]; ref [dev] fp64 [n][3]; tracked [dev] fp64 [l][3]
 * out_tracked [dev] fp64 [l][3]; out_ref [dev] fp64 [n][3] (may be null); posterior [dev] fp64 [m][n]
 * (may be null); iters [host] receives the number of EM iterations run.
 * Synchronous: the convergence test (|movement|_2 < 1e-3) is read back by the host.               */
int ct_prgls_two_ref(const double* prior, const double* tgt, int m, const double* ref, int n,
                     const double* tracked, int l, double beta, double lambda, int max_iteration,
                     double* out_tracked, double* out_ref, double* posterior, int* iters,
                     void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* The part of ct_prgls_two_ref that depends on the reference set ALONE, computed ahead of the call: the Gram matrix of `ref` (beta), its
 * pivoted-Cholesky factor and the two ranks of the low-rank M-step.  In a frame loop `ref` (the previous volume's segmentation,
 * trackerlite.py:86-93) is known before the current volume's U-Net starts, so ct_prgls_prepare_ref can run on a second stream beside it and
 * the ~0.3 ms single-workgroup factorisation leaves the frame's dependent chain.  Asynchronous; `prepared` [dev], 256-byte aligned, at least
 * ct_prgls_prepared_bytes(n) bytes, is written by ct_prgls_prepare_ref and only read by ct_prgls_two_ref_prepared (any number of times; the
 * caller orders the two streams).  ct_prgls_two_ref_prepared = ct_prgls_two_ref with the same arguments and the same results bit for bit;
 * CT_EINVAL if `prepared` does not carry what ct_prgls_prepare_ref wrote for this n and beta (the caller vouches for `ref` itself).      */
size_t ct_prgls_prepared_bytes(int n);
int ct_prgls_prepare_ref(const double* ref, int n, double beta, void* prepared, size_t prepared_bytes, ct_stream_t stream);
int ct_prgls_two_ref_prepared(const double* prior, const double* tgt, int m, const double* ref, int n,
                              const double* tracked, int l, double beta, double lambda, int max_iteration,
                              double* out_tracked, double* out_ref, double* posterior, int* iters,
                              void* workspace, size_t workspace_bytes, const void* prepared, size_t prepared_bytes,
                              ct_stream_t stream);

/* B independent prgls_with_two_ref problems (ragged sizes) in ONE chain of launches: problem b = blockIdx.z of every EM kernel.
 * A match is a chain of ~10 tiny dependent kernels per EM iteration and the GPU retires such kernels at a few hundred
 * thousand per second however many streams issue them, so the matches of independent frames (or of one ensemble prediction,
 * trackerlite.py:115-122) are cheaper batched than concurrent.  Arguments as ct_prgls_two_ref, as HOST arrays of B device
 * pointers / sizes; the EM state (moved reference set, posterior, iteration counts) is bit-identical to B separate calls and out_tracked agrees to ~1e-11
 * (the tracked set is moved once with the summed coefficients instead of once per iteration) (same kernels, same per-problem rank steering; a problem
 * whose low-rank M-step is rejected is finished by the single-problem routine).  Synchronous like ct_prgls_two_ref.        */
size_t ct_prgls_batched_workspace_bytes(int B, const int* m, const int* n, const int* l);
int ct_prgls_two_ref_batched(int B, const double* const* prior, const double* const* tgt, const int* m,
                             const double* const* ref, const int* n, const double* const* tracked, const int* l,
                             double beta, double lambda, int max_iteration, double* const* out_tracked,
                             double* const* out_ref, double* const* posterior, int* iters,
                             void* workspace, size_t workspace_bytes, ct_stream_t stream);
/* Test hook: copies an intermediate of the LAST ct_watershed_segment call out of its workspace (same dims / cap) into dst [dev]:
 * which = 0 thresholded map (uint8), 1 map without the 2-D boundaries (uint8), 2 EDT of the 2-D stage (fp64), 3 smoothed EDT (fp64),
 * 4 watershed labels before relabelling (int32), 5 window maximum (fp64).  `method | 0x100` makes ct_watershed_segment return after
 * watershed_2d so that 2-5 hold the per-slice stage (the stages share their scratch arrays).  Lets the tests hold the EDT and the
 * Gaussian to scipy's own functions (distance_transform_edt, gaussian_filter: the reference's, runnable here) voxel by voxel. */
int    ct_watershed_read_stage(const void* workspace, const int dims_xyz[3], int cap, int which, void* dst, ct_stream_t stream);

/* pr_gls_quick (track.py:11-114): X [dev] fp64 [n][3], Y [dev] fp64 [m][3], corr [dev] fp32 [m][n].
 * Runs max_iteration-1 EM iterations.  P [dev] fp64 [m][n], TX [dev] fp64 [n][3], C [dev] fp64 [3][n]. */
int ct_prgls_legacy(const double* X, int n, const double* Y, int m, const float* corr,
                    double BETA, int max_iteration, double LAMBDA, double vol,
                    double* P, double* TX, double* C,
                    void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* Fine-grained numpy helpers of trackerlite.py kept callable one by one (fp64, [dev]):
 *   dist_squares     :361-365  out[m][n] = |ref_r - tgt_t|^2
 *   gaussian_kernel  :368-372  out[m][n] = exp(-|ref_r - tgt_t|^2 / (2 sigma_square))
 *   estimate_posterior :375-382
 *   solve_movements_ref :409-417  C [3][n]; G [n][n] is the Gram matrix (symmetric)            */
int ct_dist_squares(const double* ref, int n, const double* tgt, int m, double* out, ct_stream_t stream);
int ct_gaussian_kernel(const double* ref, int n, const double* tgt, int m, double sigma_square, double* out,
                       ct_stream_t stream);
int ct_estimate_posterior(const double* prior, double sigma_square, const double* pred, int n, const double* tgt, int m,
                          double ratio_outliers, double vol, double* P, ct_stream_t stream);
int ct_solve_movements(double sigma_square, double lambda, const double* P, const double* ref, int n,
                       const double* tgt, int m, const double* G, double* C,
                       void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* Tracker._predict_one_rep (tracker.py:1269-1289): pred [dev] fp64 [l][3] (in/out),
 * inter [dev] fp64 [n][3], C [dev] fp64 [3][n]:  pred += (C . exp(-|pred - inter|^2 / 2 beta^2))^T.  */
int ct_gram_apply(double* pred, int l, const double* inter, int n, const double* C, double beta,
                  ct_stream_t stream);

/* The front half of TrackerLite.predict_cell_positions (trackerlite.py:83-91: initial_matching_ffn -> simple_match) for B independent
 * problems as ONE chain of launches (problem b = blockIdx.z; ragged reference and target sets): ref[b] [dev] fp64 [n[b]][3], tgt[b] [dev]
 * fp64 [m[b]][3] (host arrays of device pointers) -> prior_out[b] [dev] fp64 [m[b]][n[b]] (mode 0: simple_match's prior, 1: the legacy one).
 * Bit-identical to ct_knn_features x 2 + ct_ffn_pairgrid + ct_greedy_match per problem; feeds ct_prgls_two_ref_batched.
 * Synchronises `stream` (greedy round counters).                                                                                 */
size_t ct_match_front_batched_workspace_bytes(int B, int nmax, int mmax, int k_ptrs);
int ct_match_front_batched(ct_ffn_t* ffn, int B, const double* const* ref, const int* n, const double* const* tgt, const int* m, int k_ptrs,
                           float threshold, int mode, double* const* prior_out, void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* Tracker._predict_pos_once (tracker.py:1193-1222) for one source volume in ONE call: `reps` x [kNN features -> FFN pair grid ->
 * pr_gls_quick with beta * 0.8^i, every repetition starting from the previous one's transformed points] (_fit_ffn_prgls :1224-1254),
 * then _predict_one_rep (:1269-1289) for every repetition on the tracked points.  seg_pre [dev] fp64 [n][3], seg_tgt [m][3],
 * tracked_pre [l][3] (l may be 0) -> pred_out [l][3]; optional C_out [reps][3][n], inter_out [reps][n][3] (the list the reference
 * returns).  Same arithmetic as calling ct_knn_features / ct_ffn_pairgrid / ct_prgls_legacy / ct_gram_apply one by one; exists so that
 * the host threads driving the independent source volumes of an ensemble prediction (:1499-1506) are not serialised by the caller's
 * interpreter.  Synchronises `stream` (the greedy prior inside ct_prgls_legacy reads its round counter).                           */
size_t ct_legacy_predict_workspace_bytes(int n, int m, int reps, int k_ptrs);
int ct_legacy_predict_pos(ct_ffn_t* ffn, const double* seg_pre, int n, const double* seg_tgt, int m, const double* tracked_pre, int l,
                          double beta, double lambda, int max_iteration, int reps, int k_ptrs, double* pred_out, double* C_out,
                          double* inter_out, void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* The same chain for the B independent source volumes of one ensemble prediction (tracker.py:1499-1506) as ONE chain of launches:
 * problem b = blockIdx.z of every kernel (ragged n[b]; target set, tracked-point count l and all parameters shared).  seg_pre[b]
 * [dev] fp64 [n[b]][3], tracked_pre[b] [dev] fp64 [l][3] (host arrays of device pointers) -> pred_out [dev] fp64 [B][l][3].
 * Bit-identical to B calls of ct_legacy_predict_pos.  max n[b] <= 132 (the dense M-step of such a problem is one workgroup);
 * CT_ESHAPE otherwise -- call ct_legacy_predict_pos per volume then.  Synchronises `stream`.                                    */
size_t ct_legacy_predict_batched_workspace_bytes(int B, int nmax, int m, int l, int reps, int k_ptrs);
int ct_legacy_predict_pos_batched(ct_ffn_t* ffn, int B, const double* const* seg_pre, const int* n, const double* seg_tgt, int m,
                                  const double* const* tracked_pre, int l, double beta, double lambda, int max_iteration, int reps,
                                  int k_ptrs, double* pred_out, void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* trim_mean(stack, 0.1, axis=0) of k predictions (trackerlite.py:123, tracker.py:1508):
 * stack [dev] fp64 [k][n3] -> out [dev] fp64 [n3].                                               */
int ct_trim_mean(const double* stack, int k, int n3, double cut, double* out, ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pre-processing: local contrast normalisation  (preprocess.py:85-188)
 * ------------------------------------------------------------------------------------------
 * dtype: 0 = uint16, 1 = float32.  img/data [dev], [x][y][z].
 * ct_median: np.median (mean of the two middle order statistics) by radix select; median_out [dev] fp64.
 *   workspace [dev]: at least 4096 bytes; up to 33.5 KB are used (16 histogram tables, which shortens the atomic chains).
 * ct_normalize_image: subtract_median = 1 -> _normalize_image (:170-188: x = max(img - median, 0), then LCN);
 *   mode 0 = zero padding (lcn_gpu :136-167, the Keras ones-kernel Conv3D), mode 1 = scipy 'reflect' (lcn_cpu :85-114);
 *   filter: odd window sizes (reference default 27, 27, 1); out [dev] fp32 [x][y][z].                       */
size_t ct_normalize_workspace_bytes(const int dims_xyz[3]);
int ct_median(const void* data, int dtype, size_t n, double* median_out, void* workspace, size_t workspace_bytes,
              ct_stream_t stream);
int ct_normalize_image(const void* img, int dtype, const int dims_xyz[3], double noise_level, const int filter_xyz[3],
                       int mode, int subtract_median, float* out, void* workspace, size_t workspace_bytes,
                       ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Accurate correction of cell centres  (coord_image_transformer.py:292-369, :406-489)
 * ------------------------------------------------------------------------------------------
 * prob [dev] fp32 [x][y][z] on the original grid; the per-cell sub-regions live on the z-interpolated grid
 * (z * factor): bbox [dev] int32 [n][6] = start xyz, size xyz; subimages [dev] uint8 masks packed back to back
 * (C order x, y, z), sub_offsets [dev] int64 [n]; missed [dev] uint8 [n] (1 = cell skipped, e.g. on the boundary);
 * coord_vol1_raw / coords_raw [dev] fp32 [n][3] raw voxel coordinates (coords_raw is updated in place).
 * Repeats _correction_once until np.max(delta.interp) < 0.5 or max_repetition; iterations [host] receives the count.
 * Returns CT_ESHAPE where the reference raises "Slices are out of range".  Synchronises the stream every iteration.  */
size_t ct_correction_workspace_bytes(const int dims_xyz[3], int n_cells);
int ct_accurate_correction(const float* prob, const int dims_xyz[3], int factor, int n_cells, const int32_t* bbox,
                           const uint8_t* subimages, const long long* sub_offsets, const uint8_t* missed,
                           const float* coord_vol1_raw, float* coords_raw, int max_repetition, int* iterations,
                           void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* Legacy dialect: Tracker._accurate_correction (tracker.py:1177-1191; _correction_once_interp :1310-1348,
 * _transform_cells_quick :1350-1389, _evaluate_correction :1402-1413).
 * prob [dev] fp32 [x][y][z] (image_cell_bg); raw [dev] the raw frame (raw_dtype 0 = uint16, 1 = float32; NULL: no image_gcn
 * term): the centre-of-mass weight is prob + raw / 65536 in fp64 (:635, :1332).  Sub-regions as above but on the grid of
 * seg_cells_interpolated_corrected (depth interp_depth = z * z_scaling); pad_xyz = max sub-region width per axis (:1107): a
 * cell whose moved box leaves the image padded by that much is skipped for the round, like the reference's shape test.
 * on_boundary [dev] uint8 [n]; tracked_t0 [dev] fp64 [n][3] = r_coordinates_tracked_t0.
 * r_disp [dev] fp64 [n][3]: in = real displacement from volume 1 before the correction (:1179-1180), out = corrected;
 * i_disp [dev] int32 [n][3]: out = corrected displacement on the interpolated grid (i_disp_from_vol1_updated).
 * At most max_repetition rounds (reference: 20); synchronises the stream once per round.                                */
size_t ct_correction_legacy_workspace_bytes(const int dims_xyz[3], int n_cells);
int ct_accurate_correction_legacy(const float* prob, const void* raw, int raw_dtype, const int dims_xyz[3], int z_scaling,
                                  int interp_depth, double z_xy_ratio, int n_cells, const int32_t* bbox,
                                  const uint8_t* subimages, const long long* sub_offsets, const int pad_xyz[3],
                                  const uint8_t* on_boundary, const double* tracked_t0, double* r_disp, int32_t* i_disp,
                                  int max_repetition, int* iterations, void* workspace, size_t workspace_bytes,
                                  ct_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Probability map -> labelled regions -> cell centres  (tracker.py:636-648)
 * ------------------------------------------------------------------------------------------
 * The cheap variant of _segment's region step: threshold + 3D connected components (touching cells are not split; the reference's
 * marker watershed itself is ct_watershed_segment below), then applies the reference's own scipy.ndimage.center_of_mass(regions > 0, regions, 1..n) (tracker.py:646).
 * prob [dev] fp32 [x][y][z]; foreground = prob > threshold (watershed.py:38,48: 0.5); connectivity 1/2/3 =
 * scipy.ndimage.generate_binary_structure(3, c); regions with fewer than min_size voxels are dropped
 * (skimage remove_small_objects semantics) and the rest numbered 1..n in raster order of their first voxel
 * (= scipy.ndimage.label followed by relabel_sequential).
 * labels [dev] int32 [x][y][z] or NULL; centres [dev] fp64 [cap][3] voxel coordinates (x, y, z), first min(n, cap) rows;
 * sizes [dev] int32 [cap] or NULL; n_labels [dev] int32 = n (may exceed cap: caller re-runs with a larger cap).
 * Asynchronous on `stream`.                                                                                          */
size_t ct_segment_workspace_bytes(const int dims_xyz[3], int cap);
int ct_segment_centroids(const float* prob, const int dims_xyz[3], float threshold, int connectivity, int min_size,
                         int cap, int32_t* labels, double* centres, int32_t* sizes, int32_t* n_labels,
                         void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* Marker-watershed region step of the legacy Tracker on the device.
 * replaces: CellTracker/tracker.py:671-684 (Tracker._watershed) = CellTracker/watershed.py:16-53 (watershed_2d, per z slice: EDT ->
 *           gaussian_filter(2) -> peak_local_max(min_distance_2d) -> label -> watershed(-dist, markers, mask) -> find_boundaries removed)
 *           + :55-108 (watershed_3d: EDT(sampling 1, 1, z_xy_ratio) -> gaussian_filter((2, 2, 0.3)) -> peak_local_max(min_distance_3d,
 *           exclude_border=0) -> label -> watershed -> min_size / cell_num -> remove_small_objects) + relabel_sequential (tracker.py:680)
 *           + scipy.ndimage.center_of_mass(regions > 0, regions, 1..n) (tracker.py:646-647).
 * prob [dev] float32 [x][y][z]; method 0 = "min_size" (cell_num is derived), 1 = "cell_num" (min_size is derived);
 * gauss_xy / gauss_z [host]: the 2 r + 1 correlation weights of scipy's gaussian_filter1d for sigma 2 / 0.3 (truncate 4), computed by the
 * caller exactly as scipy does (3deecelltracker_amd/segment.py); labels_out [dev] int32 [x][y][z] or NULL; centres [dev] fp64 [cap][3] raw voxel
 * coordinates; sizes [dev] int32 [cap] or NULL; n_out [dev] int32 [3] = {number of cells, min_size in force, cell_num in force}.
 * More cells than `cap`: only the first `cap` centres are written (the caller retries with a larger table).  Asynchronous: nothing is waited
 * for (component lists and flags stay on the device).  CT_ESHAPE: an axis >= 16384.  More than 2048 peak candidates in a slice /
 * 8192 in the volume are latched on the device and reported as n_out[0] = -2 (labels / centres are meaningless then; n_out[1], n_out[2] = the
 * slots the stages wanted: ct_watershed_segment_ex takes larger tables).
 * Ties, as upstream resolves them (pinned against the reference on scikit-image 0.18.3, tests/test_watershed_pin.py): peak candidates of exactly
 * equal height closer than min_distance (strictly) are thinned in the order np.argsort(-values) leaves them -- numpy's generic introsort,
 * replayed on the device; seeds of exactly equal height inside one connected region are popped in the order upstream's image-wide binary heap
 * leaves them in: a z slice / volume that holds such a pair is replayed sequentially with that heap (heap_general.pxi restated; ~2 us per
 * foreground voxel of the group, mirror-symmetric shapes only), everything else is flooded component by component in parallel.
 * n_out[0] = -1: method "cell_num" asked for more cells than np.bincount has bins (the reference's IndexError, watershed.py:92).
 * Streams: everything is ordered on `stream` as seen from outside (work enqueued on it after the call sees the results).  Inside, the sweeps
 * that depend on the mask alone run on a helper stream the library keeps per device and priority of `stream` (fork / join by events, no host
 * wait; CT_WS_FORK=0 keeps everything on `stream`).  The call may be issued from several host threads (the helper's events are taken under a
 * lock for the duration of the enqueue); two calls must not share a workspace unless they are ordered on one stream.                        */
size_t ct_watershed_workspace_bytes(const int dims_xyz[3], int cap);
int    ct_watershed_segment(const float* prob, const int dims_xyz[3], double z_xy_ratio, int method, int min_size, int cell_num,
                            int min_distance_2d, int min_distance_3d, const double* gauss_xy, int radius_xy, const double* gauss_z, int radius_z,
                            int cap, int32_t* labels_out, double* centres, int32_t* sizes, int32_t* n_out,
                            void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* The same call with the peak-candidate tables sized by the caller (the reference's watershed.py:16-108 takes any stack: skimage's peak_local_max
 * has no table).  peak_cap_2d = candidate slots per z slice of the 2-D stage, peak_cap_3d = slots of the 3-D stage (ct_watershed_segment = 2048 /
 * 8192, enough for every stack measured; 16 .. 2^22 / 2^24).  A stage that wants more latches the overflow and n_out comes back as
 * {-2, slots the fullest z slice wanted, slots the volume wanted} (0 for a stage that was not reached with an overflow pending).  Retry protocol:
 * n_out[1] is exact; n_out[2] is meaningful ONLY when n_out[1] <= peak_cap_2d (after a 2-D overflow the 3-D stage ran on a truncated candidate set,
 * so its figure is a hint).  The caller therefore LOOPS -- grow each table to max(own cap, reported want), call again -- until n_out[0] != -2 or the
 * caps above are exceeded; two rounds are the common case (2-D grows, then 3-D), 3deecelltracker_amd/segment.py allows four.  Groups of up to 2048 candidates are selected by the counting
 * kernel, up to 8192 by the LDS bitonic form, beyond that the same algorithm runs on a scratch slab of the workspace (slower, same results).
 * Any z extent is accepted (per-slice statistics tables are sized from dims_xyz[2]); an axis must stay < 16384.                           */
size_t ct_watershed_workspace_bytes_ex(const int dims_xyz[3], int cap, int peak_cap_2d, int peak_cap_3d);
int    ct_watershed_segment_ex(const float* prob, const int dims_xyz[3], double z_xy_ratio, int method, int min_size, int cell_num,
                               int min_distance_2d, int min_distance_3d, const double* gauss_xy, int radius_xy, const double* gauss_z, int radius_z,
                               int cap, int peak_cap_2d, int peak_cap_3d, int32_t* labels_out, double* centres, int32_t* sizes, int32_t* n_out,
                               void* workspace, size_t workspace_bytes, ct_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* CTAMD_H */
