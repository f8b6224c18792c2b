/*
 * ppsurf_amd -- C ABI of the MI355X-native occupancy-query path of PPSurf.
 *
 * The reference (cg-tuwien/ppsurf) is pure Python: its seam is Python functions/classes, not an FFI
 * (SURVEY.md 8b).  This header is the boundary a maintainer of the reference would bind (ctypes stub in
 * INTEGRATION.md); each entry point names the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch tensors), row-major, unless it is
 *     marked [host];
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous;
 *   - return value: 0 ok, 1 bad argument, 2 launch failure (hipGetLastError != success).  No exceptions,
 *     no allocation inside: scratch comes from the caller;
 *   - "point-major" means [n,3] / [n,C] with the coordinate / channel fastest.
 */
#ifndef PPSURF_AMD_H
#define PPSURF_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library / device info --------------------------------------------------------------------------- */
int pps_abi_version(void);                       /* bumps when a signature changes */
int pps_device_cu_count(void);                   /* multiProcessorCount of the current device, <0 on error */

/* ---- spatial queries ---------------------------------------------------------------------------- */

/* Exact brute-force k nearest neighbours, k <= 64.
 * replaces: source/poco_utils.py:257-273 `knn` -> source/base/proximity.py:40-89 (pykdtree on the CPU).
 * d2 = ((dx*dx + dy*dy) + dz*dz) in fp32 without FMA contraction; neighbours sorted by (d2, index).
 * pts [n,3], query [m,3] point-major; out_idx int64 [m,k]; out_d2 f32 [m,k] or NULL.  Requires 1 <= k <= min(n,64). */
int pps_knn_f32(const float* pts, int64_t n, const float* query, int64_t m, int k,
                int64_t* out_idx, float* out_d2, void* stream);

/* Same search, same results (bit-identical indices and d2), over a cloud pre-arranged in blocks of 64 points with
 * bounding boxes so that whole blocks are rejected against the running k-th distance (exact: the box bound is evaluated in
 * the same rounding order as d2).  Built once per cloud by the caller (ppsurf_amd/ops.py `KnnBlocks`):
 * pts_blocked [nb*64,3] (points in block order, tail padded), orig_idx int32 [nb*64] (original index, -1 for padding),
 * bbox [nb,6] (min xyz, max xyz of the valid points of the block), n = number of valid points ((nb-1)*64 < n <= nb*64);
 * win_bbox [n_win,6]: boxes of windows of ceil(k/64) consecutive FULL blocks (each holds >= k points: their farthest
 * corner bounds the k-th distance from above; n_win may be 0).  Here k <= min(n, 256). */
int pps_knn_blocked_f32(const float* pts_blocked, const int32_t* orig_idx, const float* bbox, int64_t nb, int64_t n,
                        const float* win_bbox, int64_t n_win, const float* query, int64_t m, int k, int64_t* out_idx,
                        float* out_d2, void* stream);
/* The same with a second level of boxes: group_bbox [ceil(nb/64),6] = boxes of 64 consecutive blocks (NULL: none, = pps_knn_blocked_f32).
 * A query then tests the 64 block boxes of a group only where its ball reaches the group's box, and looks for its first bound among the windows
 * of one group instead of all of them (a 100k-point cloud has 1563 block boxes: testing them all was a third of the search).  Same results,
 * bit for bit: every bound is conservative and evaluated in the rounding order of d2. */
int pps_knn_blocked_groups_f32(const float* pts_blocked, const int32_t* orig_idx, const float* bbox, int64_t nb, int64_t n,
                               const float* win_bbox, int64_t n_win, const float* group_bbox, const float* query, int64_t m, int k,
                               int64_t* out_idx, float* out_d2, void* stream);

/* The neighbourhood tables of one encoder pass (or of several shapes of a fit batch) in a single launch (k <= 64 each, ntasks <= 64); arrays are [host] arrays of
 * device pointers / sizes.   replaces: the 13 `knn` calls of source/poco_data_loader.py:155-168. */
int pps_knn_multi_f32(int ntasks, const float* const* pts, const int64_t* n, const float* const* query, const int64_t* m,
                      const int* k, int64_t* const* out_idx, void* stream);

/* The large neighbourhood tables (k <= 64) of a BATCH of equally sized clouds in one launch of the block-culling search: kind t is one
 * table for all `nclouds` clouds -- points of a level arranged per cloud like for pps_knn_blocked_f32 (pts_blocked[t] [nclouds, nb*64, 3],
 * orig_idx[t] [nclouds, nb*64] (-1 = padding), bbox[t] [nclouds, nb, 6], win_bbox[t] [nclouds, n_win, 6] = boxes of the full blocks),
 * queries query[t] with query_stride[t] floats between clouds (m[t] queries each); q_orig[t] (int32 [nclouds, query_stride/3] or NULL) maps
 * the query position to its output row (queries given in Morton order: q_orig = that level's orig_idx).  out_idx[t] int64 [nclouds, m, k]
 * holds per-cloud ORIGINAL point indices, bit-identical to pps_knn_f32.  Arrays of the argument list are [host] arrays of length nkinds <= 8.
 * replaces: the kd-tree builds + queries of source/poco_data_loader.py:155-168 for the tables over the two finest levels. */
int pps_knn_blocked_batch_f32(int nkinds, int64_t nclouds, const float* const* pts_blocked, const int32_t* const* orig_idx,
                              const float* const* bbox, const int64_t* nb, const float* const* win_bbox, const int64_t* n_win,
                              const float* const* query, const int32_t* const* q_orig, const int64_t* query_stride, const int64_t* m,
                              const int* k, int64_t* const* out_idx, void* stream);

/* Voxel-stratified sub-sampling of one cloud to exactly `target` unique points, all rounds in one workgroup.
 * replaces: source/poco_data_loader.py:59-134 `sampling_quantized` (per batch item) with the semantics of its CPU execution
 * (restated and pinned in oracle/driver_oracle.py): per round the three axis rotations rots[r][0..2] (row-major 3x3 each, x
 * then y then z, drawn by the caller) applied one after the other in float32, one representative (LARGEST index) per occupied
 * voxel of edge `vox` anchored at the rotated bbox minimum, accept all and halve `vox` while fewer than `target` were taken,
 * else a random subset: the representatives with the smallest `priority[i]` (uint32 [n], ties to the lower index), or ranked by a
 * hash of `seed` when priority is NULL.
 * vox <= 0 selects the reference's default edge, bbox diagonal / sqrt(target) (:85-88), computed in the kernel.
 * pts [n,3], 2 <= n <= pps_voxel_sample_max_points(), 1 <= target < n; rots [nrot,3,9]; out_ids int64 [target] ascending;
 * out_rounds int32 or NULL. */
int pps_voxel_sample_max_points(void);
int pps_voxel_sample_f32(const float* pts, int64_t n, int64_t target, float vox, const float* rots, int nrot, uint32_t seed,
                         const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* stream);
/* The same for clouds of ANY size (n > pps_voxel_sample_max_points(): fit with manifold_points above 10240, whole clouds): hash table, state bytes
 * and compaction counts in a caller workspace `ws` (8-byte aligned, >= pps_voxel_sample_large_ws_bytes(n) bytes) instead of LDS; one workgroup,
 * identical selections. */
size_t pps_voxel_sample_large_ws_bytes(int64_t n);
int pps_voxel_sample_large_f32(const float* pts, int64_t n, int64_t target, float vox, const float* rots, int nrot, uint32_t seed,
                               const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* ws, size_t ws_bytes, void* stream);
/* The same for a batch of b equally sized clouds pts [b,n,3] in ONE launch (one workgroup per cloud; default voxel edge;
 * rots [b,nrot,3,9]; priority [b,n] or NULL; cloud i uses seed + i * 0x9e3779b9): out_ids int64 [b,target], out_rounds int32 [b] or NULL. */
int pps_voxel_sample_batch_f32(const float* pts, int64_t b, int64_t n, int64_t target, const float* rots, int nrot, uint32_t seed,
                               const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* stream);

/* Gather P neighbours per query from the raw cloud, centre at the query, divide by the max neighbour distance.
 * replaces: source/poco_utils.py:67-72 `_get_pts_local_ps` (gather + normalise part) and
 *           source/ppsurf_data_loader.py:91-123 `normalize_patches/get_patch_radii/model_space_to_patch_space`.
 * raw [n,3]; query [q,3]; idx int64 [q, idx_stride] (first p columns used); out f32 [q,p,3]. */
int pps_patch_normalize_f32(const float* raw, const float* query, const int64_t* idx, int64_t idx_stride,
                            int64_t q, int p, float* out, void* stream);

/* ---- region-growing driver on byte masks (source/poco_utils.py:178-254) ---------------------------------- */
/* Binary dilation with the box [-r, r]^3, clipped at the volume border: dst = dilate(src).
 * replaces: source/poco_utils.py:181-196 `_dilate_binary` (a Python loop over points marking arr[p - r : p + r + 1] per axis).
 * src, dst, tmp: uint8 [nx, ny, nz] (0 / non-zero; torch.bool storage), three distinct buffers; tmp is scratch. */
int pps_dilate_box_u8(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int64_t nx, int64_t ny, int64_t nz, int r, void* stream);
/* Frontier of one growth round: out[i] = to_see[i] && ((neg[i] && vol[i] >= 0) || (pos[i] && vol[i] <= 0)); NaN (never evaluated) is neither.
 * replaces: source/poco_utils.py:245-246 `new_mask = (mask_neg & (volume >= 0) & mask_to_see) | (mask_pos & (volume <= 0) & mask_to_see)`. */
int pps_grow_frontier_f64(const double* vol, const uint8_t* neg, const uint8_t* pos, const uint8_t* to_see, uint8_t* out, int64_t total, void* stream);
/* Voxels of the dilated band that still need a value: out[i] = band[i] && isnan(vol[i]) (the reference evaluates the whole band again,
 * source/poco_utils.py:198-232; the decoder is deterministic, so skipping known voxels gives the same volume). */
int pps_grow_band_todo_f64(const double* vol, const uint8_t* band, uint8_t* out, int64_t total, void* stream);

/* ---- decoder weights: host-side packing into MFMA operand order --------------------------------- */
/* All `W` are dense [out,in] row-major fp32 [host], already BatchNorm-folded / composed by the caller
 * (ppsurf_amd/decoder.py).  Packed images are plain float arrays the caller uploads to the device. */

/* size in floats of the packed image of a dense layer [out,in] (both padded to multiples of 16, out to 32) */
size_t pps_packed_dense_floats(int out, int in);
/* packed[ob][kb][lane][s] = W[16*ob + (lane&15)][16*kb + 4*(lane>>4) + s]  (zero padded) */
int pps_pack_dense_f32(const float* W, int out, int in, float* packed /* [host] */);
/* xyz layer [out,3]: packed[ob][lane] = W[16*ob + (lane&15)][lane>>4] (0 for lane>>4 == 3) */
/* Split-precision ("f16x3") A operands for v_mfma_f32_16x16x32_f16: hi = f16(W), lo = f16(W - hi), 2 x 64 x 8 halfs per
 * (16-output block, 32-input block), channel map documented in csrc/pps_common.h.  out padded to 32, in padded to 32. */
size_t pps_packed_dense_f16x3_halfs(int out, int in);
int pps_pack_dense_f16x3(const float* W /* [host] [out,in] */, int out, int in, uint16_t* packed /* [host] */);
size_t pps_packed_xyz_floats(int out);
int pps_pack_xyz_f32(const float* W, int out, float* packed /* [host] */);

/* ---- decoder kernels (eval mode, BatchNorm folded) ---------------------------------------------- */

/* out[m, n_out] = in[m, 256] * W^T + b   (one dense layer over rows; used for G = fc1[:, :256] * latent + b1).
 * replaces: the latent part of fc1 in source/poco_model.py:400-405, hoisted from per (query,neighbour) to per point.
 * in: element (row r, channel c) at in[r*in_row_stride + c*in_ch_stride]  (point-major: (256,1); channel-first: (1,n)).
 * wpack: pps_pack_dense_f32 image of W [256,256]; bias f32 [256]; out point-major [m,256]. */
int pps_rows_dense256_f32(const float* in, int64_t in_row_stride, int64_t in_ch_stride, int64_t m,
                          const float* wpack, const float* bias, float* out, void* stream);

/* Interpolation-attention branch up to the pooled feature (C = 256, heads = 64, k <= 64).
 * replaces: source/poco_model.py:381-414 `InterpAttentionKHeadsNet.forward` (gather, fc1..fc3, fc_query,
 *           softmax over neighbours, mean over heads, weighted sum); fc_value and fc8 are applied after the
 *           pooling by pps_decode_tail_f32 (sum_j a_j = 1).
 * G [n,256] (see above); pts [n,3]; query [q,3]; idx int64 [q,k];
 * wpack = concat(xyz pack of fc1[:,256:259] [256,3], dense packs of fc2, fc3 [256,256], fc_query [64,256]);
 * bias = concat(b2[256], b3[256], bq[64]); pooled out f32 [q,256]. */
int pps_interp_pool_f32(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                        const float* wpack, const float* bias, float* pooled, void* stream);

/* The whole POCO projection head for small latent sizes (c in {32,64}; configs/poco.yaml: c = 32, nout = 2).
 * replaces: source/poco_model.py:381-419 `InterpAttentionKHeadsNet.forward` incl. fc8 (POCO's `from_latent`, :357-359).
 * G [n,c] = fc1[:, :c] latent + b1 (pps_rows_gemm_f32); wpack = concat(xyz pack fc1[:,c:c+3], dense packs fc2, fc3 [c,c],
 * fc_query [64,c]); bias = concat(b2, b3, bq[64]); wtail = concat(fc8.fc_value composed [nout,c] row-major, bias [nout]);
 * out f32 [q,nout], nout <= 8. */
int pps_interp_small_f32(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k, int c,
                         const float* wpack, const float* bias, const float* wtail, int nout, float* out, void* stream);
/* The same head with fc2, fc3 and fc_query in split precision on the f16 matrix pipe (three f16 MFMA products per fp32 product, fp32 accumulation;
 * the xyz part of fc1, softmax, pooling and the composed fc8.fc_value stay fp32).  w16 = concat(pps_pack_dense_f16x3 of fc2, fc3, fc_query);
 * wpack / bias / wtail as above (the fp32 packs are the fall-back's operands).  guard: 2 device ints owned by the caller; guard[0] is cleared,
 * raised by the split-precision kernel when an activation leaves the f16 range (|x| > 65504), and the fp32 kernel queued behind it on the same
 * stream then recomputes the call (guard[1] counts those calls): the result is always finite-range safe, without a host round trip.
 * replaces: the same lines as pps_interp_small_f32 (the reference runs them under fp16 autocast on the GPU, configs/poco.yaml:10). */
int pps_interp_small_f16x3(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k, int c,
                           const float* wpack, const void* w16, const float* bias, const float* wtail, int nout, float* out, int* guard,
                           void* stream);

/* PointNet branch, phase A: per patch point conv0a, conv0b, STN conv1..3, max over the patch.
 * replaces: source/base/nn.py:323-324 and :164-170.   patches [q,p,3]; out g [q,256].
 * wpack = concat(xyz pack conv0a [64,3], dense conv0b [64,64], stn.conv1 [64,64], stn.conv2 [128,64], stn.conv3 [256,128]);
 * bias = concat(64,64,64,128,256). */
int pps_pointnet_stn_rows_f32(const float* patches, int64_t q, int p, const float* wpack, const float* bias,
                              float* g, void* stream);

/* PointNet branch, phase B: STN fully connected part 256 -> 128 -> 64 -> 4096 (+identity, folded into the bias).
 * replaces: source/base/nn.py:183-189.   g [q,256]; trans2 out [q,64,64].
 * wpack = concat(dense fc1 [128,256], fc2 [64,128], fc3 [4096,64]); bias = concat(128,64,4096).
 * The host may compose conv1 of phase C into fc3 (rows (o,b) = sum_a W1[o,a] fc3[(a,b),:], bias W1 (mat(b3) + I)): the kernel then emits
 * M = W1 trans2, which is what pps_pointnet_feat_rows_f32 expects (ppsurf_amd/decoder.py does). */
int pps_pointnet_stn_fc_f32(const float* g, int64_t q, const float* wpack, const float* bias, float* trans2, void* stream);

/* PointNet branch, phase C: conv0a/0b (recomputed), feature transform + conv1 as ONE per-query matrix (trans2 = W1 T of phase B; conv1's bias and
 * ReLU applied here), conv2, attention weights, and the attention-POOLED conv2 output.   replaces: source/base/nn.py:323-336 and :84-96 up to
 * linear maps that commute with the pooling: conv3 + bn3 (no activation, nn.py:336) and att.fc_value act on the pooled vector inside the tail.
 * patches [q,p,3]; trans2 [q,64,64]; xbar out [q,128].
 * wpack = concat(xyz conv0a, dense conv0b [64,64], conv2 [128,64]);
 * bias = concat(conv0a 64, conv0b 64, conv1 64, conv2 128, 256 unused, u = conv3^T att.fc_query weight [128] padded to 256,
 *               att.fc_query.conv3 bias + att.fc_query bias [1] padded to 4). */
int pps_pointnet_feat_rows_f32(const float* patches, const float* trans2, int64_t q, int p, const float* wpack,
                               const float* bias, float* xbar, void* stream);

/* Tail: [pooled 256 | xbar 128] -> 256 (ReLU) -> 256 (ReLU) -> 2 logits (wpack = [Wa 256x256][Wb 256x128][L2 256x256][L3 2x256 padded]; Wb carries
 * conv3 of the PointNet branch).
 * replaces: fc_value+fc8 (poco_model.py:410-417), att.fc_value (nn.py:89-93), the branch sum
 *           (source/ppsurf_model.py:100) and source/base/nn.py:415-417 `MLP.forward`, composed on the host.
 * wpack = concat(dense Wa [256,256], Wb [256,128], L2 [256,256], L3 [2,256] (padded to 32)); bias = concat(256,256,32).  pooled [q,256], xbar [q,128].
 * logits out f32 [q,2]; occ out f32 [q] = softmax(logits)[0]-softmax(logits)[1] (poco_utils.py:78-81) or NULL. */
int pps_decode_tail_f32(const float* pooled, const float* xbar, int64_t q, const float* wpack, const float* bias,
                        float* logits, float* occ, void* stream);

/* The decoder of one query chunk in ONE call (= the five launches above, in order, on `stream`).
 * replaces: source/ppsurf_model.py:82-117 `PPSurfNetwork.from_latent` (+ the occupancy of source/poco_utils.py:78-81) given the
 * per-point table G (pps_rows_dense256_f32) and the neighbour table.
 * table [n,256]; pts [n,3]; query [q,3]; idx int64 [q,k]; patches [q,p,3];
 * weights [host] array of 10 device pointers: interp (wpack, bias), stn_rows (wpack, bias), stn_fc (wpack, bias),
 * feat_rows (wpack, bias), tail (wpack, bias) -- layouts as documented at the single entry points;
 * logits out [q,2]; occ out [q] or NULL; ws: pps_decode_ws_bytes(q) bytes of device scratch, ZEROED once by the caller before its first use
 * (its first 64 bytes hold the range-guard words of pps_decode_fwd_mixed_f32: int32 [0] = flag of the chunk in flight, [1] = number of chunks
 * recomputed in fp32 so far). */
size_t pps_decode_ws_bytes(int64_t q);
int pps_decode_fwd_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                       const float* patches, int p, const float* const* weights, float* logits, float* occ, void* ws, void* stream);
/* The same call with measurement marks: events is a [host] array of 6 hipEvent_t (entries may be NULL) recorded on `stream`
 * before the first launch and after each of the five kernels (interp_pool, stn_rows, stn_fc, feat_rows, tail), so that the
 * per-kernel durations of the PRODUCT call can be read with hipEventElapsedTime (bench.py's roofline leg). */
int pps_decode_fwd_events_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                              const float* patches, int p, const float* const* weights, float* logits, float* occ, void* ws,
                              void* const* events, void* stream);

/* Opt-in split-precision interpolation branch (decoder dtype "f16x3"): same contract as pps_interp_pool_f32 with fc2 / fc3 / fc_query
 * evaluated as three f16 MFMA products per fp32 product (error ~1e-5 on logits of magnitude 30, tests/test_gpu_decoder.py).
 * wxyz: the 1024 packed floats of fc1's xyz part (front of the fp32 image); w16: pps_pack_dense_f16x3 images of fc2, fc3, fc_query
 * back to back (147456 halfs x 2); bias [256 | 256 | 64] floats.
 * range_flag (all three split-precision entries): device int32 or NULL.  x = hi + lo with hi = f16(x) needs |x| <= 65504; a kernel that splits an
 * activation beyond that ORs 1 into *range_flag (its results are then wrong: the conversion saturates, no inf appears).  The caller zeroes the
 * word, and on a non-zero value repeats the chunk with the fp32 entries -- pps_decode_fwd_mixed_f32 does both on the device.  The reference's
 * own GPU path (fp16 autocast, configs/poco.yaml:10) has the same range and no guard. */
int pps_interp_pool_f16x3(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                          const float* wxyz, const void* w16, const float* bias, float* pooled, int32_t* range_flag, void* stream);
/* The PointNet branch in split precision: weights = the six fp32 images (stn_rows, stn_fc, feat_rows: wpack, bias each -- xyz layers and
 * biases are read from them), w16 [host] array of 3 pps_pack_dense_f16x3 image sets: (c0b, s1, s2, s3), (fc1, fc2, fc3 with conv1 composed), (c0b, c2);
 * g [q,256], trans2 (q x 16 KiB: pre-split fragments of the per-query feature transform), xbar [q,128]; events: 3 hipEvent_t or NULL. */
int pps_pointnet_f16x3(const float* patches, int64_t q, int p, const float* const* weights, const void* const* w16, float* g, float* trans2,
                       float* xbar, int32_t* range_flag, void* const* events, void* stream);
/* The tail (pps_decode_tail_f32: source/ppsurf_model.py:100, source/base/nn.py:376-417 composed with fc8 . fc_value | att.fc_value) in split
 * precision: w16 = pps_pack_dense_f16x3 images of Wa 256x256 and Wb 256x128 interleaved per pair of output blocks (32 KiB of Wa for blocks 2c, 2c+1,
 * then 16 KiB of Wb for the same blocks), then [L2 256x256][L3 2x256]; bias as for the fp32 entry. */
int pps_decode_tail_f16x3(const float* pooled, const float* xbar, int64_t q, const void* w16, const float* bias, float* logits, float* occ,
                          int32_t* range_flag, void* stream);
/* pps_decode_fwd_events_f32 with branches in split precision: w16 [host] array of 5 image sets (interp, stn_rows, stn_fc, feat_rows, tail);
 * w16[0] NULL keeps the fp32 interpolation branch, any of w16[1..3] NULL keeps the fp32 PointNet branch, w16[4] NULL the fp32 tail.
 * events may be NULL.
 * Range guard: behind the split-precision kernels the call queues the five fp32 kernels of the same chunk, each returning at once unless a
 * split-precision kernel raised the flag word in `ws` -- a chunk whose activations leave the f16 range (|x| > 65504) is recomputed in exact fp32
 * on the device (bit-identical to pps_decode_fwd_f32), every other chunk pays five empty launches (~12 us). */
int pps_decode_fwd_mixed_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                             const float* patches, int p, const float* const* weights, const void* const* w16, float* logits, float* occ,
                             void* ws, void* const* events, void* stream);

/* ---- Marching Cubes on the device ------------------------------------------------------------------
 * replaces: skimage.measure.marching_cubes(volume, level=mc_value) called at source/poco_utils.py:95-96.
 * vol float64 [nx,ny,nz] (z fastest), NaN = never evaluated: a cube is triangulated when its 8 corners are finite and not all on one side of
 * `level` (inside = value > level).  Tables (host-derived, ppsurf_amd/mcubes.py): tri int8 [rows, width, 3] cube-edge ids (12 = the extra
 * vertex inside the cube, -1 = none), ntri uint8 [rows], amb uint8 [256].  Rows 0 .. 256*64-1: (corner pattern * 64 + one asymptotic-decider bit
 * per ambiguous face), every loop closed by a disc.  Interior ambiguity (skimage's Lewiner variant resolves it, cases 4, 6, 7, 10, 12, 13 of
 * Lewiner et al. 2003): tun_index int32 [256*64, 2] = (first, count) into tun_cand int32 [n, 3] = (sign: 1 inside / 0 outside corner groups,
 * mask: bit 2*axis + diagonal = a plane sweep that decides the pair, alternative row >= 256*64 with a tube between the two loops); the first
 * candidate whose interior test (test_interior of the paper, general form) succeeds replaces the row.
 * Two calls with two small prefix sums by the caller in between:
 *   pps_mc_count_f64: edge_flags uint8 [3*nx*ny*nz] (zeroed and filled here: 1 = the grid edge (voxel, axis) carries a vertex), block_tris /
 *                     block_centres int32 [pps_mc_cube_blocks] = triangles / centre vertices per block of 256 cubes, block_verts int32
 *                     [pps_mc_edge_blocks] = vertices per block of 1024 edge flags;
 *   pps_mc_emit_f64:  tri_offset / centre_offset / vert_offset int64 = EXCLUSIVE prefix sums of those counts, n_edge_verts = their vertex total;
 *                     vidx int32 [3*nx*ny*nz] scratch (edge -> vertex index); verts out float64 [n_edge_verts + centres, 3] in index space (edge
 *                     vertices in ascending (voxel, axis) order, then centre vertices in cube order); faces out int64 [triangles, 3], normals
 *                     towards lower values, in cube order. */
int64_t pps_mc_cube_blocks(int64_t nx, int64_t ny, int64_t nz);
int64_t pps_mc_edge_blocks(int64_t nx, int64_t ny, int64_t nz);
int pps_mc_count_f64(const double* vol, int64_t nx, int64_t ny, int64_t nz, double level, const int8_t* tri, int width, const uint8_t* ntri,
                     const uint8_t* amb, const int32_t* tun_index, const int32_t* tun_cand, uint8_t* edge_flags, int32_t* block_tris,
                     int32_t* block_centres, int32_t* block_verts, void* stream);
int pps_mc_emit_f64(const double* vol, int64_t nx, int64_t ny, int64_t nz, double level, const int8_t* tri, int width, const uint8_t* ntri,
                    const uint8_t* amb, const int32_t* tun_index, const int32_t* tun_cand, const uint8_t* edge_flags, const int64_t* tri_offset,
                    const int64_t* centre_offset, const int64_t* vert_offset, int64_t n_edge_verts, int32_t* vidx, double* verts, int64_t* faces,
                    void* stream);

/* ---- mesh clean-up (csrc/pps_mesh.hip) -------------------------------------------------------------
 * replaces: source/base/mesh.py:7-38 (trimesh merge_vertices(digits_vertex = 8), remove_degenerate_faces, remove_duplicate_faces,
 * remove_small_connected_components(num_faces = 6)) as called from source/poco_utils.py:98-107, 169-174.
 *   pps_mesh_small_components   small uint8 [nf] = 1 for the faces of face-connected components (faces sharing an edge that belongs to exactly two
 *                               faces: trimesh's face_adjacency; an edge of three or more faces joins nothing) of at most k faces
 *                               (1 <= k <= 32); faces int64 [nf, 3], vertex ids below nv.
 *   pps_mesh_corner_weld        for a mesh welded by grid-edge key (pps_mc_emit_f64) in index space: vertices within 10^-digits of a grid corner
 *                               that share their position rounded to `digits` digits are merged into the smallest id of their class.  remap int64
 *                               [nv] (identity elsewhere), hot uint8 [nv] = 1 where something was merged into the vertex, counters int32 [2] =
 *                               {merged vertices, 1 if a coordinate left [0, 524287]}.
 *   pps_mesh_face_filter        faces AFTER remapping: keep uint8 [nf] = 0 for degenerate faces and, among the faces touching a hot vertex, for all
 *                               but the first face of every vertex triple.
 * ws: the matching *_ws_bytes() bytes each; results do not depend on scheduling. */
size_t pps_mesh_components_ws_bytes(int64_t nf);
int pps_mesh_small_components(const int64_t* faces, int64_t nf, int64_t nv, int k, uint8_t* small, void* ws, void* stream);
size_t pps_mesh_weld_ws_bytes(int64_t nv);
int pps_mesh_corner_weld(const double* verts, int64_t nv, int digits, int64_t* remap, uint8_t* hot, int* counters, void* ws, void* stream);
size_t pps_mesh_face_filter_ws_bytes(int64_t nf);
int pps_mesh_face_filter(const int64_t* faces, int64_t nf, const uint8_t* hot, uint8_t* keep, void* ws, void* stream);

/* ---- FKAConv encoder (eval mode), point-major activations, one batch item per call ----------------- */

/* number of floats of the packed small parameters of one FKAConv layer:
 * [norm_radius, alpha, beta, act(1 relu | 2 silu), fc1[16][3], fc2[16][32], fc3[16][32], bn1.w[16], bn1.b[16], bn2.w[16], bn2.b[16]] */
size_t pps_fkaconv_geo_floats(void);
/* bytes of workspace for a layer with m support points and cin input channels
 * (per-block partial InstanceNorm statistics in double + the aggregated features F [m, cin*16]) */
size_t pps_fkaconv_ws_bytes(int64_t m, int cin);

/* One FKAConv layer.   replaces: source/base/nn.py:592-652 `FKAConvLayer.forward` (eval: norm_radius fixed).
 * x [n,cin], pts [n,3], sup [m,3], idx int64 [m,k] (k <= 16; InstanceNorms skipped when k == 1, nn.py:627-638);
 * wpack: pps_pack_dense_f32 image of W [cout, cin*16] with W[o][c*16+t] = cv.weight[o,c,0,t] (a following BatchNorm may be
 * folded in); bias [cout] or NULL; act_out 0 none | 1 ReLU; out [m,cout]; ws: pps_fkaconv_ws_bytes(m, cin) bytes. */
int pps_fkaconv_fwd_f32(const float* x, const float* pts, const float* sup, const int64_t* idx, int64_t n, int64_t m, int k,
                        int cin, int cout, const float* geo, const float* wpack, const float* bias, int act_out, float* out,
                        void* ws, void* stream);

/* Same contract as pps_rows_linear_f32 on the fp32 matrix cores; needs c1 % 16 == 0 and c2 % 16 == 0.
 * wpack: pps_pack_dense_f32 image of W [cout, c1+c2]. */
int pps_rows_gemm_f32(const float* in1, const int64_t* idx1, int c1, const float* in2, const int64_t* idx2, int c2,
                      const float* wpack, const float* bias, const float* residual, int act, int64_t m, int cout, float* out,
                      void* stream);

/* out[m,o] = act(bias[o] + sum_c A[m,c] wt[c,o] + residual[m,o]),  A[m] = [in1[idx1[m]] (c1) | in2[idx2[m]] (c2)].
 * replaces: Conv1d(k=1) + BatchNorm1d (folded) + ReLU, torch.cat, nearest-neighbour `interpolate` (nn.py:684-697, K=1)
 *           and the residual add of nn.py:440-448, 532-548.   idx1/idx2 int64 [m] or NULL (identity); in2 NULL if c2 == 0. */
int pps_rows_linear_f32(const float* in1, const int64_t* idx1, int c1, const float* in2, const int64_t* idx2, int c2,
                        const float* wt, const float* bias, const float* residual, int act, int64_t m, int cout, float* out,
                        void* stream);

/* out[m,c] = max_j x[idx[m,j], c].   replaces: source/base/nn.py:677-680 `max_pool` and the global max of :531. */
int pps_gather_max_f32(const float* x, const int64_t* idx, int64_t m, int k, int c, float* out, void* stream);

/* ---- training step: neighbourhood gathers with hand-written backward ------------------------------------------------ */
/* Scatter-adds (the transposes of the gathers) use no atomics: the caller passes the CSR of the id table,
 *   order   int64 [entries]  entry numbers (m*k + j, or the row number r for a flat table) stably sorted by target row,
 *   offsets int64 [n+1]      entries of target row i are order[offsets[i] .. offsets[i+1]),
 * and every target row adds its contributions in that fixed order (bit-reproducible gradients). */

/* The CSR itself, built on the device by a stable counting sort (pps_csr.hip: count, scan, fill, rank; deterministic, no comparison sort):
 *   ids      int64 [entries]  the id table, row-major [B, M, K] flattened (per_item = M*K > 0: entry e belongs to batch item e / per_item and points
 *                             at row ids[e] + item * rows_per_item of the [B * rows_per_item, C] activations), or already flat rows (per_item = 0);
 *   clamp_negative            -1 entries ("no neighbour" of the nearest up-sampling tables, nn.py:686-687) count as row 0;
 *   flat     int64 [entries]  out, the flat row numbers (NULL to skip);   order / offsets as above (offsets [rows + 1]);
 *   ws                        pps_csr_ws_bytes(entries, rows) bytes.  Ids must lie in [0, rows) after flattening (entries outside are left out).
 * replaces: the index_add backward of source/base/nn.py:655-674 `batch_gather` under autograd (atomics in arbitrary order there). */
size_t pps_csr_ws_bytes(int64_t entries, int64_t rows);
int pps_csr_build(const int64_t* ids, int64_t entries, int64_t per_item, int64_t rows_per_item, int64_t rows, int clamp_negative,
                  int64_t* flat, int64_t* order, int64_t* offsets, void* ws, size_t ws_bytes, void* stream);

/* out[r,:] = x[idx[r],:].   replaces: source/base/nn.py:655-674 `batch_gather` for the latent gather of
 * source/poco_model.py:400 and the nearest-neighbour up-sampling of nn.py:684-697. */
int pps_gather_rows_f32(const float* x, const int64_t* idx, int64_t r, int c, float* out, void* stream);

/* out[i,:] = sum_{e in offsets[i]..offsets[i+1]} vals[order[e],:]  (vals [entries,c], out [n,c]): backward of
 * pps_gather_rows_f32 and second half of the backward of pps_neighbour_contract_fwd_f32. */
int pps_segment_sum_rows_f32(const float* vals, const int64_t* order, const int64_t* offsets, int64_t n, int c, float* out,
                             void* stream);

/* the same with 16-bit values (dtype 1 bfloat16, 2 IEEE half; c % 4 == 0), fp32 accumulation and output */
int pps_segment_sum_rows_16(const void* vals, const int64_t* order, const int64_t* offsets, int64_t n, int c, int dtype, float* out, void* stream);

/* FKAConv feature aggregation, out[m, ch*16+t] = sum_j x[idx[m,j], ch] * g[m,j,t]  (x [n,c], idx int64 [m,k], g [m,k,16],
 * out [m, c*16] in the (channel, kernel-column) order of cv.weight[Cout, Cin, 1, 16]).
 * replaces: source/base/nn.py:598,647-649 (batch_gather of the features + the two transposes around torch.matmul). */
int pps_neighbour_contract_fwd_f32(const float* x, const int64_t* idx, const float* g, int64_t m, int k, int c, float* out,
                                   void* stream);
/* its backward: dxg [m,k,c] = per-entry gradient of the gathered rows (NULL to skip; reduce it to x rows with
 * pps_segment_sum_rows_f32), dg [m,k,16] (NULL to skip). */
int pps_neighbour_contract_bwd_f32(const float* x, const int64_t* idx, const float* g, const float* dout, int64_t m, int k, int c,
                                   float* dxg, float* dg, void* stream);

/* The same two with x, out and dout in the storage type of an autocast step (dtype 0 float = the _f32 entries, 1 bfloat16, 2 IEEE half; g, dxg, dg
 * and the arithmetic stay fp32): no cast kernels around the op and half the traffic.  16-bit storage needs pps_neighbour_contract_16_supported(k, c)
 * (c % 16 == 0, k <= 16: the matrix-pipe kernels). */
int pps_neighbour_contract_16_supported(int k, int c);
int pps_neighbour_contract_fwd(const void* x, const int64_t* idx, const float* g, int64_t m, int k, int c, int dtype, void* out, void* stream);
int pps_neighbour_contract_bwd(const void* x, const int64_t* idx, const float* g, const void* dout, int64_t m, int k, int c, int dtype,
                               float* dxg, float* dg, void* stream);

/* Neighbourhood max-pool that also records the winning neighbour: out [m,c], arg int32 [m,c] (first j attaining the max).
 * replaces: source/base/nn.py:677-680 `max_pool` in training. */
int pps_gather_max_arg_f32(const float* x, const int64_t* idx, int64_t m, int k, int c, float* out, int32_t* arg, void* stream);
/* its backward: dx [n,c] from dout [m,c], arg and the CSR of idx. */
int pps_gather_max_bwd_f32(const float* dout, const int32_t* arg, const int64_t* order, const int64_t* offsets, int64_t n, int k,
                           int c, float* dx, void* stream);
/* the same pair on 16-bit storage (dtype 1 bfloat16, 2 IEEE half): x, out, dout, dx in that type -- the maximum is one of the stored values, the
 * gradient is summed in fp32 in CSR order and rounded once; what an autocast step would otherwise do with four cast kernels around each op. */
int pps_gather_max_arg_16(const void* x, const int64_t* idx, int64_t m, int k, int c, int dtype, void* out, int32_t* arg, void* stream);
int pps_gather_max_bwd_16(const void* dout, const int32_t* arg, const int64_t* order, const int64_t* offsets, int64_t n, int k, int c, int dtype,
                          void* dx, void* stream);

/* FKAConv geometry branch in train() mode, forward and backward (the part of FKAConvLayer.forward that turns neighbour
 * offsets into the [M,K,16] kernel-weighting matrix).   replaces: source/base/nn.py:601-643 under autograd.
 * A batch of b shapes with m support points each: pts [rows,3], sup [b*m,3], idx int64 [b*m,k] = row numbers into pts
 * (batch offsets included), k <= 16.  geo_w [pps_fkaconv_geo_floats()] = the layer's packed small parameters (layout of
 * pps_fkaconv_fwd_f32: norm_radius, alpha, beta, activation, fc1/fc2/fc3 weights, InstanceNorm affine parameters);
 * momentum > 0 first updates geo_w[0] (norm_radius) in place with the EMA of nn.py:608-613 and then uses the new value.
 * g_out [b*m, k, 16];  stat [2][b][32] receives (mean, rstd) of the two InstanceNorms per shape (input of the backward).
 * The InstanceNorms (statistics per shape over all (point, neighbour) pairs) are skipped when k == 1 (nn.py:627-638). */
size_t pps_fka_train_ws_bytes(int64_t b, int64_t m, int k);
int pps_fka_geometry_fwd_f32(const float* pts, const float* sup, const int64_t* idx, int64_t b, int64_t m, int k, float* geo_w,
                             float momentum, float* g_out, float* stat, void* ws, void* stream);
/* backward: dg [b*m,k,16] -> dgeo [pps_fkaconv_geo_floats()] (gradients of alpha, beta, fc1/fc2/fc3, InstanceNorm affine
 * parameters in the geo layout; entries of norm_radius / activation are 0).  Recomputes the branch (nothing but `stat` is
 * kept from the forward); deterministic (fixed-order reductions, no atomics). */
int pps_fka_geometry_bwd_f32(const float* pts, const float* sup, const int64_t* idx, int64_t b, int64_t m, int k, const float* geo_w,
                             const float* stat, const float* dg, float* dgeo, void* ws, void* stream);

/* BatchNorm1d in train() mode with the ReLU fused, on point-major activations x [rows, c] (c % 4 == 0, 256 % (c/4) == 0,
 * c <= 1024), storage dtype 0 = fp32, 1 = bf16, 2 = IEEE half (x, y, dy, dx share it), fp32 arithmetic.
 * replaces: activation(bn(conv(x))) of source/base/nn.py:438-450,508-554,162-190,323-336,376-417 under autograd.
 * forward: batch statistics (biased variance) -> save [2][c] (mean, rstd); running_mean/var (NULL or both) updated in place with
 * `momentum` and the unbiased variance like torch.nn.BatchNorm1d; y = relu? max(0, .) : . of the normalised, affine output.
 * backward: dx, dgamma [c], dbeta [c] from x, dy and `save` (the ReLU mask is recomputed from x).  Deterministic. */
size_t pps_bn_train_ws_bytes(int64_t rows, int c);
int pps_bn_train_fwd(const void* x, int64_t rows, int c, int dtype, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int relu, void* y, float* save, void* ws, void* stream);
int pps_bn_train_bwd(const void* x, const void* dy, int64_t rows, int c, int dtype, const float* gamma, const float* beta, const float* save,
                     int relu, void* dx, float* dgamma, float* dbeta, void* ws, void* stream);

/* relu(BN(x) + res) in train() mode: the tail of a residual block (source/base/nn.py:448-450, `self.activation(x + shortcut)` behind bn2) inside the
 * BatchNorm's own apply pass (the separate add and ReLU were two more passes over [rows, c] forward and one backward).  res [rows, c] in the
 * storage type of x; statistics, save, ws and the running statistics as pps_bn_train_fwd.  Backward: dx as pps_bn_train_bwd with the ReLU mask
 * taken behind the sum; dres [rows, c] = the masked upstream gradient (the shortcut's gradient). */
int pps_bn_add_relu_fwd(const void* x, const void* res, int64_t rows, int c, int dtype, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, void* y, float* save, void* ws, void* stream);
int pps_bn_add_relu_bwd(const void* x, const void* res, const void* dy, int64_t rows, int c, int dtype, const float* gamma, const float* beta,
                        const float* save, void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream);

/* out[c] = sum over the rows of x [rows, c]: the bias gradient of a row layer (replaces the `grad_output.sum(0)` autograd runs for the bias of
 * every Conv1d(…,1) / Linear of source/base/nn.py).  Shapes and dtype codes as pps_bn_train_*; ws: pps_bn_train_ws_bytes(rows, c) bytes.
 * Deterministic (fixed-order fp32 per thread, double across threads and blocks). */
int pps_col_sum(const void* x, int64_t rows, int c, int dtype, float* out, void* ws, void* stream);
/* The same over c columns of a wider tensor: row r starts at x + r * ld elements (ld >= c, ld % 4 == 0, x aligned to 4 elements) -- the column
 * blocks of a [rows, 4096] gradient are summed where they lie, without a contiguous copy of each block. */
int pps_col_sum_strided(const void* x, int64_t rows, int c, int64_t ld, int dtype, float* out, void* ws, void* stream);

/* Attention pooling of the interpolation head in train(): a[j] = mean_h softmax_j(qy[q,j,h]), pooled[q,c] = sum_j a[j] h[q,j,c]
 * (replaces source/poco_model.py:412-414 under autograd, in the pooled form where fc_value follows the pooling).  qy [q,k,heads], h [q,k,c],
 * pooled [q,c]; heads <= 64 (64: interpolation head; 1: PointNet's AttentionPoco, source/base/nn.py:84-96 with k = patch points), k <= 64, c <= 256; storage float (bf16 = 0), bfloat16 (bf16 = 1) or IEEE half (bf16 = 2), arithmetic fp32.
 * Backward: dqy [q,k,heads], dh [q,k,c] from dpooled [q,c]; the softmax is recomputed from qy.
 * relu_h != 0: h is stored BEFORE its ReLU (the raw output of fc3): the ReLU is applied on load and dh is the gradient wrt the stored values. */
int pps_attn_pool_fwd(const void* qy, const void* h, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* pooled, void* stream);
int pps_attn_pool_bwd(const void* qy, const void* h, const void* dpooled, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* dqy,
                      void* dh, void* stream);
/* pps_attn_pool_bwd without dh: dh[q,j,c] = relu'(h[q,j,c]) * a[q,j] * dpooled[q,c] is one multiply per element, so the [q k, c] tensor is neither
 * written here nor read back by its consumer -- `weights` [q, k] fp32 receives a (the softmax averaged over the heads) and
 * pps_rows_layer_bwd_attn rebuilds the product where it adds the two gradients of h. */
int pps_attn_pool_bwd_weights(const void* qy, const void* h, const void* dpooled, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* dqy,
                              float* weights, void* stream);

/* The 16-bit tensors of the training-step entries below are bfloat16 (dtype = 1, trainer.precision bf16-mixed) or IEEE half (dtype = 2,
 * trainer.precision 16-mixed, the reference's default); "bf16" in the descriptions stands for either.
 *
 * Dense layer of the training step over point-major rows with the previous layer's BatchNorm + ReLU applied on load and this layer's
 * batch statistics taken on store (replaces the conv -> bn -> relu -> conv chains of source/base/nn.py:162-190, 323-336, 376-417 and the
 * fc -> relu -> fc chain of source/poco_model.py:400-410 under autograd and bf16 autocast; pps_rows_train.hip).
 *   x [rows, cin], y [rows, cout] bfloat16 RAW layer outputs; cin, cout in {64, 128, 256} (pps_rows_layer_supported)
 *   act(x) = relu?(x * in_scale + in_shift)   (in_scale / in_shift [cin], both NULL = no affine part: identity, or a bare ReLU with in_relu)
 *   fwd:  y = act(x) w^T + bias  (w [cout, cin] fp32, rounded to bf16 for the product; bias NULL = none).  With gamma != NULL the batch
 *         statistics of y give out_affine [2][cout] = (gamma rstd, beta - mean gamma rstd), save [2][cout] = (mean, rstd) and the
 *         running statistics are updated like torch.nn.BatchNorm1d (both NULL = not tracked).
 *   bwd:  from gy [rows, cout] bf16 and d_affine [2][cout] (loss gradient wrt out_affine; with gamma): dx [rows, cin] bf16 (NULL = skip;
 *         dx_add [rows, cin] bf16, NULL or a gradient of x from another consumer that is added to the result -- may be dx itself),
 *         d_in_affine [2][cin] (NULL = skip), dw [cout, cin], dbias [cout] (NULL = skip), dgamma, dbeta [cout].
 * ws: pps_rows_layer_ws_bytes(cin, cout) bytes of device scratch.  Deterministic. */
/* Single-head attention pooling over the rows of a group with the logit computed inside (PointNet's AttentionPoco, source/base/nn.py:84-96, applied to
 * the raw conv3 output: the logit is linear in it and softmax ignores the constant): a = softmax_j(h[q,j,:] . v), pooled[q,:] = sum_j a_j h[q,j,:].
 * h [q, k, 256] bfloat16, k <= 64, v [256] fp32, pooled [q, 256] fp32.  Backward: dh [q, k, 256] bfloat16; dv_part [pps_patch_attn_partials(q)][256],
 * to be summed over its first axis. */
int pps_patch_attn_partials(int64_t q);
int pps_patch_attn_fwd(const void* h, const float* v, int64_t q, int k, int c, int dtype, float* pooled, void* stream);
int pps_patch_attn_bwd(const void* h, const float* v, const float* dpooled, int64_t q, int k, int c, int dtype, void* dh, float* dv_part, void* stream);
/* pps_patch_attn_bwd without dh: dh[q, j, :] = a[q, j] dpooled[q, :] + dl[q, j] v has rank two per group, so the [q k, 256] tensor is not written --
 * `weights` and `dlogits` [q, k] fp32 receive a and dl (dl = a (dpooled . h_j - sum_i a_i dpooled . h_i)) and pps_rows_layer_bwd_rank2 rebuilds the
 * rows where the producing layer's backward reads them. */
int pps_patch_attn_bwd_weights(const void* h, const float* v, const float* dpooled, int64_t q, int k, int c, int dtype, float* weights, float* dlogits,
                               float* dv_part, void* stream);

/* Input of the interpolation head in train() (source/poco_model.py:400-404 with fc1 split into its latent and its offset part):
 * h1[(q,j),:] = table[ids[q,j],:] + wx (query[q] - pts[ids[q,j]]).  table [n, c] bf16, ids [q*k] (rows of table and pts), pts [n, 3] and
 * query [q, 3] fp32, wx [c, 3] fp32, h1 [q*k, c] bf16; c a multiple of 8 with (c / 8) dividing 256.  pps_head_input_dwx: d wx [c, 3] from
 * dh1 [q*k, c] bf16 (d table is the segmented sum of dh1, pps_segment_sum_rows_16).  ws: pps_head_input_ws_bytes(c) bytes. */
size_t pps_head_input_ws_bytes(int c);
int pps_head_input_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, int dtype, const float* wx,
                       void* h1, void* stream);
int pps_head_input_dwx(const void* dh1, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, int dtype, float* dwx, void* ws,
                       void* stream);

/* The dense chain of the interpolation head in train(), forward, in one launch (source/poco_model.py:400-409 on all (query, neighbour) rows of the
 * batch):  h1 = table[ids] + wx (query - pts[ids]),  y2 = fc2(relu(h1)),  y3 = fc3(relu(y2)),  qy = fc_query(relu(y3)) -- a wave carries its rows
 * through the three layers in registers; h1, y2, y3 [q*k, 256] and qy [q*k, 64] (16-bit, RAW layer outputs: what the backward entries
 * pps_attn_pool_bwd / pps_rows_layer_bwd / pps_head_input_dwx read) are written once each.  table [n, 256] 16-bit, ids [q*k], pts [n, 3], query [q, 3]
 * fp32; wx [256, 3], w2 / w3 [256, 256], wq [64, 256] fp32 master weights, b2 / b3 [256], bq [64] fp32 or NULL.  The four outputs must have room
 * for q*k rounded UP to a multiple of 256 rows (whole row units are written; the rows past q*k hold copies of the last row's results).
 * ws: pps_head_chain_ws_bytes() bytes, 16-byte aligned (the weights as MFMA fragments, rebuilt by every call: they change every step). */
size_t pps_head_chain_ws_bytes(void);
int pps_head_chain_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int dtype, const float* wx,
                       const float* w2, const float* b2, const float* w3, const float* b3, const float* wq, const float* bq, void* h1, void* y2, void* y3,
                       void* qy, void* ws, void* stream);

/* conv0a of PointNet in train() (3 coordinates -> 64 channels, source/base/nn.py:323): y [rows, 64] bf16 = x [rows, 3] w^T + bias with the batch
 * statistics of y -> out_affine / save / running statistics as in pps_rows_layer_fwd; backward: dw [64, 3], dbias [64] (NULL = skip), dgamma,
 * dbeta (x gets no gradient: it is the input patch).  ws: pps_rows3_ws_bytes() bytes. */
size_t pps_rows3_ws_bytes(void);
int pps_rows3_fwd(const float* x, int64_t rows, const float* w, const float* bias, int dtype, void* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float momentum, float eps, float* out_affine, float* save, void* ws, void* stream);
int pps_rows3_bwd(const float* x, const void* y, const void* gy, int64_t rows, int dtype, const float* gamma, const float* save, const float* d_affine,
                  float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, void* stream);

/* Feature transform of PointNet in train() (source/base/nn.py:330-331, torch.bmm(trans2, x)): out[q,i,:] = act(x)[q,i,:] T[q]^T per group q of
 * p <= 64 rows.  x [q*p, 64] bf16 stored activation, act = relu?(x * in_scale + in_shift) (NULL = identity), T [q, 64, 64] bf16 (+ I if
 * add_identity), out [q*p, 64] bf16.  Backward from g = d out: dx [q*p, 64] bf16, dt [q, 64, 64] bf16, d_in_affine [2][64] (NULL = skip).
 * ws: pps_patch_transform_ws_bytes() bytes. */
size_t pps_patch_transform_ws_bytes(void);
int pps_patch_transform_fwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, int64_t q,
                            int p, int dtype, void* out, void* stream);
int pps_patch_transform_bwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, const void* g,
                            int64_t q, int p, int dtype, void* dx, void* dt, float* d_in_affine, void* ws, void* stream);

/* Extrema over the p rows of every group of x [groups, p, c] (bfloat16, c % 4 == 0): mx, mn [groups, c] fp32 and the row of each
 * (first occurrence).  The max-pool over the patch points (source/base/nn.py:181) of relu(bn(x)) follows from them without the activated tensor. */
int pps_rows_extrema_16(const void* x, int64_t groups, int p, int c, int dtype, float* mx, float* mn, int* amx, int* amn, void* stream);
int pps_rows_layer_supported(int cin, int cout);
size_t pps_rows_layer_ws_bytes(int cin, int cout);
int pps_rows_layer_fwd(const void* x, int64_t rows, int cin, int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                       const float* bias, int cout, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, float* out_affine, float* save, void* ws, void* stream);
int pps_rows_layer_bwd(const void* x, const void* y, const void* gy, int64_t rows, int cin, int cout, int dtype, const float* in_scale,
                       const float* in_shift, int in_relu, const float* w, const float* gamma, const float* save, const float* d_affine,
                       void* dx, const void* dx_add, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                       void* stream);

/* pps_rows_layer_bwd for fc_query of the interpolation head (source/poco_model.py:404-409: no statistics, identity input affine, ReLU on the input
 * x = the raw output of fc3), with the OTHER gradient of x -- the attention pooling's -- rebuilt in the epilogue instead of read as dx_add:
 *   dx[r, c] = relu'(x[r, c]) * ((gy W)[r, c] + att_weights[r] * att_dpooled[r / att_k, c])
 * att_weights [rows] fp32 (pps_attn_pool_bwd_weights), att_dpooled [rows / att_k, cin] in the storage type, rows % att_k == 0. */
int pps_rows_layer_bwd_attn(const void* x, const void* gy, int64_t rows, int cin, int cout, int dtype, const float* w, const float* att_weights,
                            const void* att_dpooled, int att_k, void* dx, float* dw, float* dbias, void* ws, void* stream);

/* pps_rows_layer_bwd for a layer whose raw output only feeds a max over the p rows of every group (the last layer of PointNet's STN before
 * `torch.max(x, 2)`, source/base/nn.py:181): the incoming gradient is ONE value per (group, channel) -- gval [rows / pool_p, cout] in the storage type --
 * placed in the winning row garg [rows / pool_p, cout] (uint8, 0 <= garg < pool_p) and zero elsewhere; the kernels rebuild the rows of that 98 %-zero
 * tensor on load instead of reading it from memory (it is never written either).  Same results as pps_rows_layer_bwd on the scattered tensor.
 * Shapes: pps_rows_layer_pooled_supported(cin, cout, pool_p) (128 -> 256 with BatchNorm, 2 <= pool_p <= 255, rows % pool_p == 0, rows * pool_p < 2^32). */
int pps_rows_layer_pooled_supported(int cin, int cout, int pool_p);
int pps_rows_layer_bwd_pooled(const void* x, const void* y, const void* gval, const uint8_t* garg, int pool_p, int64_t rows, int cin, int cout,
                              int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma,
                              const float* save, const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma,
                              float* dbeta, void* ws, void* stream);

/* pps_rows_layer_bwd for the layer whose raw output only feeds pps_patch_attn_fwd over the pool_p rows of every group (conv3 / bn3 of PointNet in
 * front of AttentionPoco, source/base/nn.py:84-96,333-336): the incoming gradient gy[r, c] = g_a[r] g_dp[r / pool_p, c] + g_dl[r] g_v[c]
 * (g_a, g_dl [rows] and g_dp [rows / pool_p, cout], g_v [cout] fp32: pps_patch_attn_bwd_weights) is rebuilt on load in the input-gradient and the
 * weight-gradient kernel, rounded to the storage type like the tensor it replaces.  Shapes as pps_rows_layer_bwd_pooled (cin 128, cout 256,
 * BatchNorm statistics on the output). */
int pps_rows_layer_bwd_rank2(const void* x, const void* y, const float* g_a, const float* g_dl, const float* g_dp, const float* g_v, int pool_p, int64_t rows,
                             int cin, int cout, int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma,
                             const float* save, const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma,
                             float* dbeta, void* ws, void* stream);

/* AdamW step over all parameter tensors of a group in one launch: replaces the optimizer step of the reference's trainer
 * (configs/poco.yaml:60-69 torch.optim.AdamW; arithmetic of torch's fused implementation, amsgrad and maximize off).
 * pieces: device array of n_pieces records of pps_adamw_piece_bytes() = 48 bytes {float* param, float* grad, float* exp_avg, float* exp_avg_sq,
 * const float* step, int32 n, int32 pad}, one workgroup per record (n <= 4096 elements of one tensor); steps: device array of the n_steps
 * distinct `step` scalars (float, one per parameter tensor), each advanced by one before the update.  lr_dev (device float) overrides lr when
 * not NULL.  grad_scale / found_inf (device floats, NULL = none) as torch.amp.GradScaler hands them to a fused optimizer: gradients are divided by
 * *grad_scale (and written back), and nothing at all happens when *found_inf != 0. */
int pps_adamw_piece_bytes(void);
/* 16-bit images (dtype 1 = bfloat16, 2 = IEEE half; round to nearest even) of many fp32 tensors in one launch: pieces = device array of n_pieces
 * records of pps_cast_piece_bytes() = 24 bytes {const float* src, uint16* dst, int32 n, int32 pad}, one workgroup each (n <= 4096 elements).
 * replaces: the per-weight `.to(dtype)` casts torch.autocast inserts in front of every conv / linear of a training forward pass
 * (source/base/nn.py layers under trainer.precision 16-mixed / bf16-mixed). */
int pps_cast_piece_bytes(void);
int pps_cast_pieces(const void* pieces, int n_pieces, int dtype, void* stream);
int pps_adamw_step(const void* pieces, int n_pieces, const void* steps, int n_steps, const float* lr_dev, float lr, float beta1, float beta2,
                   float eps, float weight_decay, const float* grad_scale, const float* found_inf, void* stream);

/* ---- dense layers of the training step, any layer shape (csrc/pps_gemm_train.hip) ---------------------------------------------------------------
 * replaces: the library GEMMs behind F.linear / torch.mm / torch.bmm of the layers the fused row kernels above do not take: every 1x1 Conv1d / Linear
 * and the (1,16) Conv2d of the FKAConv encoder (source/base/nn.py:438-450, 508-554, 571, 650), the per-point table, fc_value, fc8 of the
 * interpolation head (source/poco_model.py:405-417), the STN's fully connected layers (nn.py:183-188), att.fc_value and the MLP (nn.py:376-417).
 * 16-bit storage (dtype 1 bfloat16, 2 IEEE half), fp32 accumulation on the matrix pipe; row pitches in elements, multiples of 8.
 *   pps_gemm_nt_16: y [m, n] = x [m, k] w [n, k]^T (+ bias [n] fp32); y 16-bit, or fp32 if out_f32.  The input gradient of the layer is the same
 *                   call with (g, transposed image of w).
 *   pps_gemm_tn_16: dw [n, k] fp32 = g [m, n]^T x [m, k] (contraction over the rows; slabs summed in a fixed order); ws: pps_gemm_tn_ws_bytes.
 *   pps_transpose_cast_pieces: transposed 16-bit images [k, n] of many fp32 matrices [n, k] in one launch; table = device array of
 *                   pps_transpose_entry_bytes() = 32-byte records {const float* src; void* dst; int32 n, k; int64 tile0}, tile0 = number of 32 x 32
 *                   tiles of the records before this one, tiles = their total. */
int pps_gemm_nt_16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y, int64_t ldy, int64_t m, int n, int k, int dtype,
                   int out_f32, void* stream);
size_t pps_gemm_tn_ws_bytes(int64_t m, int n, int k);
int pps_gemm_tn_16(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t m, int n, int k, int dtype, float* dw, void* ws, void* stream);
int pps_transpose_entry_bytes(void);
int pps_transpose_cast_pieces(const void* table, int entries, int64_t tiles, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPSURF_AMD_H */
