/*
 * octa_hip.h -- C-ABI of liboctahip.so, the MI355X (gfx950) implementation of the
 * OCTA-autosegmentation data-parallel hot path.
 *
 * The reference (aiforvision/OCTA-autosegmentation) is pure Python and has no FFI;
 * its seams are Python call sites. Every entry point below names the reference
 * call site (file:line under the reference tree) whose arithmetic it replaces.
 * The Python host in octa_autosegmentation_amd/ keeps the reference's function
 * signatures (tree2img.rasterize_forest, ...) and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - `d_` pointers are DEVICE pointers (HBM), owned by the caller (torch tensors);
 *    `h_` pointers are HOST pointers. Nothing is allocated for the caller.
 *  - every launch is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *    the default stream) and returns without synchronising unless stated.
 *  - return value: 0 = ok, <0 = error; octa_last_error() gives the message of the
 *    last failure on the calling thread. Functions never throw.
 *  - an octa_ctx owns grow-only scratch in HBM for one device; use one ctx per
 *    process/GPU. A ctx is not safe for concurrent use from two threads.
 *  - all simulator / rasteriser arithmetic is IEEE double or integer; device code
 *    is compiled with -ffp-contract=off.
 */
#ifndef OCTA_HIP_H
#define OCTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct octa_ctx octa_ctx;

/* ---- library / context ------------------------------------------------- */

/* ABI version of this header (bumped on incompatible change). */
int octa_abi_version(void);

/* Message of the last error on this thread ("" if none). */
const char *octa_last_error(void);

/* Create / destroy the per-device context. device = HIP device ordinal. */
int octa_ctx_create(int device, octa_ctx **out);
void octa_ctx_destroy(octa_ctx *ctx);

/* Bytes of device scratch currently held by the context. */
size_t octa_ctx_scratch_bytes(const octa_ctx *ctx);

/* ---- N5: 2-D anti-aliased graph rasteriser ------------------------------
 * Replaces: vessel_graph_generation/tree2img.py:65-113 (rasterize_forest: radius
 * filter, x1.3 width, LineCollection + figure.canvas.draw() = matplotlib Agg,
 * RGBA -> L) for a batch of B graphs at once.
 * Callers in the reference: generate_vessel_graph.py:79-86,
 * visualize_vessel_graphs.py:95-101, data/data_transforms.py:376-386.
 *
 *  d_edges     [n_total][7] double : node1.xyz, node2.xyz, radius (CSV column order,
 *                                    generate_vessel_graph.py:43-47)
 *  h_edge_off  [B+1] int64 (host)  : graph b owns edges [h_edge_off[b], h_edge_off[b+1])
 *  d_keep      [n_total] uint8 or NULL : 0 = edge dropped by the caller's dropout /
 *                                    blackdict logic (tree2img.py:78-80, host-side RNG)
 *  no_pixels_x, no_pixels_y        : `image_resolution` of the reference; the image is
 *                                    no_pixels_y rows by no_pixels_x columns
 *  mip_axis                        : projection axis (0,1,2)
 *  min_radius, max_radius          : inclusive radius window (tree2img.py:67)
 *  d_out       [B][no_pixels_y][no_pixels_x] uint8 : grey image, fully overwritten
 * Edges are blended in list order (order is part of the result).
 */
int octa_rasterize_2d(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off,
                      const uint8_t *d_keep, int no_pixels_x, int no_pixels_y, int mip_axis,
                      double min_radius, double max_radius, uint8_t *d_out, void *stream);

/* octa_rasterize_2d in two calls, for callers that share the GPU with a long-running kernel (pipeline.py: the generator rasterises one
 * batch while the persistent simulator kernel of the next one holds the CUs). _plan takes octa_rasterize_2d's arguments without d_out, runs
 * the per-edge records and the side-offset scan and performs the rasteriser's ONE host synchronisation (the side array is sized from the
 * scan total) and every scratch allocation; _draw enqueues tessellation, row binning and the ordered fold on `stream` and never waits for
 * the device. A context holds one plan: every _draw consumes the plan of the _plan before it (-2 without one). Same stream for both.
 * octa_rasterize_2d(...) == _plan(...) followed by _draw(d_out). */
int octa_rasterize_2d_plan(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off,
                           const uint8_t *d_keep, int no_pixels_x, int no_pixels_y, int mip_axis,
                           double min_radius, double max_radius, void *stream);
int octa_rasterize_2d_draw(octa_ctx *ctx, uint8_t *d_out, void *stream);

/* Diagnostics of the last octa_rasterize_2d call: h_out4 = {error flag, ticks in edge binning, ticks in
 * stroke tessellation, ticks in the ordered fold}; ticks are 100 MHz wall-clock ticks summed over workgroups. */
int octa_raster_prof(octa_ctx *ctx, int64_t *h_out4);

/* ---- N7: Floyd-Steinberg binarisation -----------------------------------
 * Replaces: Pillow Image.convert("1") as called at visualize_vessel_graphs.py:99
 * (label PNGs). d_in/d_out: [B][H][W] uint8; output values are 0 or 255.
 */
int octa_fs_dither(octa_ctx *ctx, int B, const uint8_t *d_in, int W, int H, uint8_t *d_out, void *stream);

/* ---- N6: 3-D tube voxeliser ------------------------------------------------
 * Replaces: vessel_graph_generation/tree2img.py:176-280 (voxelize_forest, cuboid getCrossSlice :151-172)
 * for B graphs. dims3 = volume_dimensions; every axis is padded to at least
 * ceil(scale/76 + 0.03*scale) voxels (scale = max(dims3)) exactly as the reference does -- query the
 * padded shape with octa_voxel_padded_dims. d_out: uint16 [B][X'][Y'][Z'] (padded dims), overwritten.
 * d_keep as in octa_rasterize_2d. Callers: generate_vessel_graph.py:69-77, visualize_vessel_graphs.py:77-94.
 */
int octa_voxel_padded_dims(const int *dims3, int *padded3);
int octa_voxelize_3d(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off, const uint8_t *d_keep,
                     const int *dims3, double min_radius, double max_radius, int ignore_z, uint16_t *d_out, void *stream);

/* ---- graph CSV and PNG files (host; csrc/fileio.cpp) ----------------------------
 * The reference's on-disk formats, byte for byte where text is the contract:
 *   `node1,node2,radius` rows of generate_vessel_graph.py:59-66 / forest.py:196-207 -- positions as str(np.ndarray)
 *   (numpy default print options), radius as repr(float), "\r\n" row ends (csv.writer default dialect).
 * octa_csv_format_edges: h_edges [n][7] (node xyz, parent xyz, radius) -> text in `out` (capacity `cap` >=
 *   octa_csv_bytes_bound(n)); returns the byte count. octa_csv_write_file does the same into a file.
 * octa_csv_parse_edges: the read-back of visualize_vessel_graphs.py:71-75 + the "Legacy" string branch of
 *   tree2img.py:73-76 (split on blanks, float()) -> h_out [rows][7]; returns the row count (octa_csv_count_rows sizes it).
 * octa_png_write_gray8 / octa_png_write_bits: tree2img.py:282-292 (uint8 "L" image) and
 *   visualize_vessel_graphs.py:99 (mode "1" label: one byte per pixel in, non-zero = white); `level` = zlib level
 *   (-1: fastest sensible). Pixels decode identically to Pillow's files; compressed bytes are not part of the contract.
 * All return < 0 on error (octa_last_error()). */
int64_t octa_csv_bytes_bound(int64_t n_edges);
int64_t octa_csv_format_edges(const double *h_edges, int64_t n_edges, char *out, int64_t cap);
int octa_csv_write_file(const char *path, const double *h_edges, int64_t n_edges);
int64_t octa_csv_count_rows(const char *text, int64_t len);
int64_t octa_csv_parse_edges(const char *text, int64_t len, double *h_out, int64_t cap_rows);
/* random.random() called n_draws times, on the state random.getstate() reports (uint32 [625]: 624 words + position): the draws of
 * the reference's per-edge dropout test (tree2img.py:62,78) replayed without a Python loop. */
int octa_py_random_advance(uint32_t *state625, int64_t n_draws);
int octa_png_write_gray8(const char *path, const uint8_t *h_pixels, int width, int height, int level);
int octa_png_write_bits(const char *path, const uint8_t *h_pixels, int width, int height, int level);
/* The files of a whole batch in one call, written by `threads` native threads (round 6): per sample k the directory dirs[k] (created
 * with its parents) receives config.yml (config_text, when given), <names[k]>.csv (rows edge_off[k] .. edge_off[k+1] of h_edges, when
 * given), art_ven_img_gray.png (h_images [n][image_h][image_w], when given) and <names[k]>_label.png (h_labels [n][label_h][label_w],
 * non-zero = white -- or packed rows, see below --, when given): generate_vessel_graph.py:43-86 + visualize_vessel_graphs.py:95-101 for n samples, outside the
 * interpreter lock. Returns 0, or < 0 with the first failure in octa_last_error(). */
int octa_write_sample_files(int64_t n_samples, const char *const *dirs, const char *const *names, const double *h_edges,
                            const int64_t *edge_off, const uint8_t *h_images, int image_w, int image_h, const uint8_t *h_labels,
                            int label_w, int label_h, int labels_packed, const char *config_text, int64_t config_len, int png_level, int threads);
/* labels_packed != 0: h_labels holds mode "1" rows already, [n][label_h][(label_w + 7) / 8] bytes (bit 7 of a byte = its first pixel), as
 * octa_pack_bits writes them ON THE DEVICE: d_in uint8 [n_rows][width] (non-zero = white) -> d_out uint8 [n_rows][(width + 7) / 8]; an eighth of
 * the label bytes crosses PCIe. */
int octa_pack_bits(octa_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, int64_t n_rows, int width, void *stream);

/* ---- CSV round trip of node positions ------------------------------------
 * Replaces the text round trip str(np.ndarray) -> float() of generate_vessel_graph.py:59-66 +
 * tree2img.py:73-76 for a whole edge array on the device: d_out[n][7] receives the positions exactly as
 * they read back from the CSV (radius copied). *h_n_unhandled (optional, forces a stream sync) counts
 * 3-vectors outside the exact range of the kernel (non-finite, |x| < 1e-14 in scientific rows, |x| >= 4e7);
 * they are copied through unchanged and the caller must redo those rows on the host.
 */
int octa_edges_read_back(octa_ctx *ctx, const double *d_edges, double *d_out, int64_t n_edges, int *h_n_unhandled, void *stream);

/* ---- element-wise max of two uint8 images -------------------------------
 * Replaces: np.maximum(art_mat, ven_mat) at generate_vessel_graph.py:83.
 */
int octa_max_u8(octa_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, uint8_t *d_out, size_t n, void *stream);

/* ---- N8 (part): fused InstanceNorm2d(affine) + LeakyReLU ---------------------
 * Replaces the norm1+lrelu / norm2+lrelu pairs of every UnetBasicBlock of DynUNet (MONAI, imported at
 * models/networks.py:6; configs/config_ves_seg-S.yml:6-13) in the training step of
 * models/base_model_abc.py:152-167. NCHW contiguous activations, dtype 0 = float32, 1 = bfloat16;
 * d_w / d_b: per-channel affine (float32, may be NULL); statistics in float32 [B*C].
 * Forward: y = lrelu((x - mean) * rstd * w + b). Backward: dx, and dw/db accumulated over the batch.
 */
int octa_instnorm_lrelu_fwd(octa_ctx *ctx, const void *d_x, void *d_y, const float *d_w, const float *d_b, float *d_mean,
                            float *d_rstd, int B, int C, int64_t hw, int dtype, float slope, float eps, void *stream);
int octa_instnorm_lrelu_bwd(octa_ctx *ctx, const void *d_x, const void *d_dy, const float *d_w, const float *d_b,
                            const float *d_mean, const float *d_rstd, void *d_dx, float *d_dw, float *d_db, int B, int C,
                            int64_t hw, int dtype, float slope, void *stream);

/* Channels-last variants for the MFMA convolution path: activations [B][HW][C] bfloat16, C a multiple of 8
 * (<= 512); same arithmetic and the same float32 statistics / affine gradients as the two functions above. */
int octa_instnorm_lrelu_nhwc_fwd(octa_ctx *ctx, const void *d_x, void *d_y, const float *d_w, const float *d_b, float *d_mean,
                                 float *d_rstd, int B, int C, int64_t hw, float slope, float eps, void *stream);
int octa_instnorm_lrelu_nhwc_bwd(octa_ctx *ctx, const void *d_x, const void *d_dy, const float *d_w, const float *d_b,
                                 const float *d_mean, const float *d_rstd, void *d_dx, float *d_dw, float *d_db, int B, int C,
                                 int64_t hw, float slope, void *stream);

/* Statistics only (for the normalise-on-load convolutions below): mean / rstd and the per-(image, channel) scale =
 * w * rstd and shift = b - mean * scale (float32 [B*C] each). octa_scale_shift_lrelu_nhwc materialises
 * y = lrelu(x * scale + shift) where a consumer needs the tensor itself. */
int octa_instnorm_nhwc_stats(octa_ctx *ctx, const void *d_x, const float *d_w, const float *d_b, float *d_mean, float *d_rstd,
                             float *d_scale, float *d_shift, int B, int C, int64_t hw, float eps, void *stream);
int octa_scale_shift_lrelu_nhwc(octa_ctx *ctx, const void *d_x, void *d_y, const float *d_scale, const float *d_shift, int B, int C,
                                int64_t hw, float slope, void *stream);

/* ---- 3x3 convolution on the matrix cores, NHWC bf16 (fp32 accumulate) --------------------
 * Replaces the bias-free 3x3 convolutions of DynUNet's UnetBasicBlock / UnetUpBlock (MONAI, imported at
 * models/networks.py:6; configs/config_ves_seg-S.yml:6-13: filters [32,64,128,256,512], strides [1,2,2,2,1])
 * in models/base_model_abc.py:152-167 (octa_conv2d_{fwd,dgrad}).
 * d_x: [N][H][W][Cin] bf16, d_w: the weights [tap = 3*r + s][Cout][Cin] bf16 in SLICE-MAJOR storage order -- element (t, co, ci) at
 * ((ci / 16 * 9 + t) * Cout + co) * 16 + ci % 16, i.e. [Cin/16][9][Cout][16]: the 16 input channels of one MFMA K-step are contiguous
 * for all taps and output channels (csrc/conv.hip wt_off(); produced by octa_pack_conv_weights / mfma_conv.slice_major()). Every "d_w packed"
 * of the MFMA entry points below (3x3, 4x4: 16 taps, the stride-2 "up" form, the padded / mirrored form) uses this order. d_y: [N][Ho][Wo][Cout]
 * bf16; padding 1; stride 1 or 2. in_dilation 2 reads d_x through a virtual zero insertion (x[i/2] at even
 * virtual positions, 0 elsewhere; virtual size 2H x 2W): with flipped + transposed weights that is the
 * data gradient of a stride-2 layer, with in_dilation 1 that of a stride-1 layer.
 * Ho = (H*in_dilation - 1) / stride + 1 (likewise Wo). Cin and Cout must be multiples of 32.
 */
int octa_conv3x3_nhwc_fwd(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin,
                          int Cout, int stride, int in_dilation, void *stream);

/* Same, with a VIRTUAL channel concatenation of two inputs (channels [0, C1) from d_x [.. C1], the rest from d_x2
 * [.. Cin - C1]; d_x2 NULL = single input) and a split output (channels [0, CY1) to d_y, the rest to d_y2; d_y2
 * NULL = single output): the decoder's torch.cat((up, skip), 1) of MONAI's UnetUpBlock and its backward slicing
 * never materialise. C1 and CY1 multiples of 32.
 * tap_mask: bit 3*r+s set = tap (r, s) is evaluated; taps whose weights are structurally zero (the 2x2 transposed
 * convolution written as the adjoint of a stride-2 3x3 layer uses 4 of 9) are skipped. 0x1ff = all. */
int octa_conv3x3_nhwc_fwd2(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2, int CY1,
                           int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask, void *stream);

/* Same, with a scattered output: result pixel (y, x) is stored at (y * out_scale + out_off_y, x * out_scale +
 * out_off_x) of a d_y image out_scale (1 or 2) times larger. With out_scale 2, one call per parity class and the
 * taps / weights of that class, a stride-2 data gradient or a 2x2 transposed convolution runs without any
 * multiplication by inserted zeros. */
int octa_conv3x3_nhwc_fwd3(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2, int CY1,
                           int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask, int out_scale,
                           int out_off_y, int out_off_x, void *stream);

/* Same, with NORMALISE-ON-LOAD: if d_scale1 / d_shift1 (float32 [N][C1]) are given, input 1 is the raw output of an
 * earlier convolution and is read as lrelu(x * scale + shift) -- InstanceNorm(affine) + LeakyReLU(slope) per image
 * and channel, rounded to bf16 exactly as the materialised tensor would be; likewise d_scale2 / d_shift2
 * ([N][Cin - C1]) for input 2. Padding stays zero. The normalised activations of MONAI's UnetBasicBlock then never
 * go to HBM. */
int octa_conv3x3_nhwc_fwd4(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2, int CY1,
                           int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask, int out_scale,
                           int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1, const float *d_scale2,
                           const float *d_shift2, float slope, void *stream);
int octa_conv3x3_nhwc_wgrad3(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H, int W,
                             int Cin, int Cout, int tap_mask, const float *d_scale1, const float *d_shift1, const float *d_scale2,
                             const float *d_shift2, float slope, void *stream);

/* Same, and the statistics of the InstanceNorm that follows come for free: d_stat_partials (float32
 * [N][octa_conv_stat_tiles(Ho, Wo)][Cout][2], may be NULL) receives per output tile and channel the sum and the sum of
 * squares of the bf16-rounded results; octa_instnorm_lrelu_nhwc_fwd_p folds them instead of re-reading the tensor. */
int octa_conv_stat_tiles(int Ho, int Wo);
int octa_conv3x3_nhwc_fwd5(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2, int CY1,
                           int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask, int out_scale,
                           int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1, const float *d_scale2,
                           const float *d_shift2, float slope, float *d_stat_partials, void *stream);
/* _fwd5 plus a residual: d_residual (NULL = none) has the shape of d_y and is ADDED to the bf16-rounded result (fp32 add, rounded
 * again: what a separate bf16 tensor addition would store). Used by the data gradient of a layer whose input has a second
 * consumer -- the skip connections of the U-Net (MONAI DynUNet, models/networks.py:6): the decoder's gradient of the skip tensor is
 * ready first and rides along in the epilogue of the encoder's data-gradient launch instead of a separate addition pass. Plain single
 * output only (no scatter, split, statistics or normalise-on-load). */
int octa_conv3x3_nhwc_fwd6(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2, int CY1,
                           int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask, int out_scale,
                           int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1, const float *d_scale2,
                           const float *d_shift2, float slope, float *d_stat_partials, const void *d_residual, void *stream);
/* The two layers of the U-Net whose output is twice the size of their input, with the four output parities fused (round 5): the data
 * gradient of a stride-2 3x3 convolution (MONAI UnetBasicBlock with stride 2) and the 2x2 stride-2 transposed convolution (UnetUpBlock.transp_conv;
 * both imported at models/networks.py:6). d_x [N][H][W][Cin] bf16 is the SMALL image, d_w [9][Cout][Cin] bf16 the pack the zero-insertion form
 * reads (octa_conv3x3_nhwc_fwd2 with in_dilation = 2 computes the same sums, three quarters of them against inserted zeros), d_y
 * [N][2H][2W][Cout] bf16. tap_mask: 0x1ff (data gradient of a stride-2 layer) or 0b000011011 (the transposed convolution: one tap per parity).
 * d_residual (shape of d_y, NULL = none) is added as octa_conv3x3_nhwc_fwd6 adds it. Cin, Cout multiples of 32. */
int octa_conv3x3_s2t_nhwc(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int tap_mask,
                          const void *d_residual, void *stream);
/* InstanceNorm(affine) + LeakyReLU + 1x1 convolution to ONE channel with bias, fused: the last norm of DynUNet's decoder followed
 * by UnetOutBlock (MONAI, imported at models/networks.py:6; 32 -> 1 channels at 1216^2). d_x [B][hw][C] bf16 (the raw output of the last
 * 3x3 convolution), d_w / d_b the norm's affine parameters (NULL = none), d_head_w float32[C], d_head_b float32[1] or NULL ->
 * d_logits bf16 [B][hw]; mean / rstd [B*C] are kept for backward. The normalised tensor and its gradient never reach HBM: backward
 * rebuilds dL/dy[p][c] = dlogit[p] * head_w[c] from d_dlogits (bf16 [B][hw]) and writes d_dx (bf16), d_dw / d_db (float32[C], may be
 * NULL), d_dhead_w (float32[C]) and d_dhead_b (float32[1] or NULL), all overwritten. C in {8, 16, 32, 64, 128, 256}. */
int octa_instnorm_lrelu_head1_nhwc_fwd(octa_ctx *ctx, const void *d_x, const float *d_w, const float *d_b, const float *d_head_w,
                                       const float *d_head_b, float *d_mean, float *d_rstd, void *d_logits, int B, int C, int64_t hw,
                                       float slope, float eps, void *stream);
/* octa_instnorm_lrelu_head1_nhwc_fwd with the statistics of d_x supplied in slot form (double [nslot][B][C][2]) by the convolution that wrote it
 * (octa_conv3x3_nhwc_fwd7): the layer then makes no statistics pass of its own. */
int octa_instnorm_lrelu_head1_nhwc_fwd_s(octa_ctx *ctx, const void *d_x, const float *d_w, const float *d_b, const float *d_head_w,
                                         const float *d_head_b, float *d_mean, float *d_rstd, void *d_logits, int B, int C, int64_t hw,
                                         float slope, float eps, const double *d_slots, int nslot, void *stream);
int octa_instnorm_lrelu_head1_nhwc_bwd(octa_ctx *ctx, const void *d_x, const void *d_dlogits, const float *d_w, const float *d_b,
                                       const float *d_head_w, const float *d_mean, const float *d_rstd, void *d_dx, float *d_dw, float *d_db,
                                       float *d_dhead_w, float *d_dhead_b, int B, int C, int64_t hw, float slope, void *stream);
int octa_instnorm_lrelu_nhwc_fwd_p(octa_ctx *ctx, const void *d_x, void *d_y, const float *d_w, const float *d_b, float *d_mean,
                                   float *d_rstd, int B, int C, int64_t hw, float slope, float eps, const float *d_partials, int tiles,
                                   void *stream);
/* The same statistics in SLOT form (round 5): MONAI's UnetBasicBlock is conv -> InstanceNorm(affine) -> LeakyReLU (imported at
 * models/networks.py:6); octa_conv3x3_nhwc_fwd7 is octa_conv3x3_nhwc_fwd2 (one or two virtually concatenated inputs, stride 1 or 2,
 * plain single output d_y [N][Ho][Wo][Cout]) whose epilogue also adds, per output channel, the sum and the sum of squares of the
 * bf16-rounded results of every output tile to d_stat_slots = double[nslot][N][Cout][2] (ZERO on entry, 1 <= nslot <= 1024; tile t adds
 * to slot t % nslot, so the chain of same-address atomics stays short). octa_instnorm_lrelu_nhwc_fwd_s is
 * octa_instnorm_lrelu_nhwc_fwd with those slots instead of its own statistics pass over d_x: one launch, one read of d_x. */
int octa_conv3x3_nhwc_fwd7(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, int N, int H, int W,
                           int Cin, int Cout, int stride, double *d_stat_slots, int nslot, void *stream);
int octa_instnorm_lrelu_nhwc_fwd_s(octa_ctx *ctx, const void *d_x, void *d_y, const float *d_w, const float *d_b, float *d_mean,
                                   float *d_rstd, int B, int C, int64_t hw, float slope, float eps, const double *d_stat_slots, int nslot,
                                   void *stream);

/* 3 x 3 convolution, stride 1, with an explicit padding (round 3): pad = 0 (valid), 1 (same), 2 (full: the data gradient of a valid
 * convolution), zeros outside the image; reflect = 1 (pad = 1 only) fuses nn.ReflectionPad2d(1) into the halo fetch -- the ResNet
 * blocks of the reference's generator (models/networks.py:151-176: ReflectionPad2d(1) + Conv2d(3, padding 0)) without a padded or a
 * cropped copy. d_x [N][H][W][Cin], d_w packed [9][Cout][Cin], d_y [N][H + 2 pad - 2][W + 2 pad - 2][Cout], all bf16; Cin, Cout
 * multiples of 32. octa_conv3x3_nhwc_wgrad_pad is its weight gradient (d_dw fp32 [9][Cout][Cin], overwritten). */
int octa_conv3x3_nhwc_fwd_pad(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                              int reflect, void *stream);
int octa_conv3x3_nhwc_wgrad_pad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin, int Cout, int pad,
                                int reflect, void *stream);
/* octa_conv3x3_nhwc_fwd_pad with the InstanceNorm statistics of the result accumulated by the kernel's epilogue in slot form (d_stat_slots: double
 * [nslot][N][Cout][2], zeroed by the caller, consumed by octa_instnorm_lrelu_nhwc_fwd_s; as octa_conv3x3_nhwc_fwd7): the generator's residual
 * blocks are reflect-padded convolution -> InstanceNorm (models/networks.py:151-176), 18 per pass. NULL slots = octa_conv3x3_nhwc_fwd_pad. */
int octa_conv3x3_nhwc_fwd_pad_s(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                                int reflect, double *d_stat_slots, int nslot, void *stream);

/* Weight gradient with a stride: stride 2 = d_x is the [N][H][W][Cin] input of a stride-2 layer (H, W even), d_dy its
 * [N][H/2][W/2][Cout] output gradient; tap_mask as above (the 2x2 transposed convolution, written as the adjoint of a
 * stride-2 layer, asks for 4 of the 9 taps). */
int octa_conv3x3_nhwc_wgrad4(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H, int W,
                             int Cin, int Cout, int stride, int tap_mask, const float *d_scale1, const float *d_shift1,
                             const float *d_scale2, const float *d_shift2, float slope, void *stream);

/* The weight gradient ADDED to a gradient buffer in the parameter's own layout, float32 [Cout][Cin][3][3] (models/networks.py / MONAI
 * state-dict layout; base_model_abc.py:152-167's `loss.backward()` accumulates into exactly this tensor): the training step passes
 * `weight.grad`, a view of its flat gradient arena -- no [9][Cout][Cin] temporary, fill, layout copy or accumulation launch per layer.
 * Operands as octa_conv3x3_nhwc_wgrad4 without the normalise-on-load vectors (stride 1, or 2 with even H, W); taps outside tap_mask add
 * zero. octa_conv3x3_nhwc_wgrad_pad_acc: the same for octa_conv3x3_nhwc_wgrad_pad. */
int octa_conv3x3_nhwc_wgrad_acc(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_grad, int N, int H, int W,
                                int Cin, int Cout, int stride, int tap_mask, int accumulate, void *stream);
int octa_conv3x3_nhwc_wgrad_pad_acc(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_grad, int N, int H, int W, int Cin, int Cout,
                                    int pad, int reflect, int accumulate, void *stream);

/* Weight gradient of the stride-1 layer above: d_dw [9][Cout][Cin] float32 (overwritten) =
 * sum over pixels of d_dy[N][H][W][Cout] (bf16) x d_x[N][H][W][Cin] (bf16) shifted by the tap (SURVEY.md 8b:
 * octa_conv2d_wgrad). fp32 accumulation; partial sums of the persistent workgroups meet in fp32 atomics, so the
 * last bits depend on the arrival order. */
int octa_conv3x3_nhwc_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin,
                            int Cout, void *stream);

/* Single-precision convolution on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulation), NCHW: the
 * convolutions of the reference's paths that run WITHOUT mixed precision -- test.py:79 / validate.py call model.inference outside
 * torch.cuda.amp.autocast -- i.e. torch.nn.Conv2d / ConvTranspose2d of models/networks.py (DynUNet 3x3 stride 1 / 2, 1x1 head with
 * bias, 2x2 stride-2 and 1x1 transposed convolutions) evaluated in fp32.
 * d_x [N][Cin][H][W], d_wp the weights packed [Cin][K*K][cout_w] (output channel innermost, cout_w >= Cout), d_bias [Cout] or NULL,
 * d_y [N][Cout][Ho*osc][Wo*osc]: osc = 1 dense output; osc = 2 writes only the pixels (2 oy + ooy, 2 ox + oox) -- one of the four
 * 1x1 products of a 2x2 stride-2 transposed convolution. K / stride in {1/1, 3/1, 3/2}; zero padding `pad`. */
int octa_conv2d_f32_nchw(octa_ctx *ctx, const float *d_x, const float *d_wp, const float *d_bias, float *d_y, int N, int Cin, int H, int W,
                         int Cout, int cout_w, int K, int stride, int pad, int Ho, int Wo, int osc, int ooy, int oox, void *stream);

/* torch.nn.ConvTranspose2d(Cin, Cout, 2, 2, bias=False) in fp32, one launch (DynUNet's upsampling: models/networks.py:6 -> MONAI
 * UnetUpBlock.transp_conv, on the test.py:79 / validate.py fp32 path): d_x [N][Cin][H][W], d_wp the weights packed [Cin][4][Cout]
 * (tap 2 a + b), d_y [N][Cout][2H][2W] (8-byte aligned), y(co, 2y + a, 2x + b) = sum_ci x(ci, y, x) w(ci, co, a, b). Same numbers as
 * four octa_conv2d_f32_nchw calls with osc = 2. */
int octa_convtranspose2x2_f32_nchw(octa_ctx *ctx, const float *d_x, const float *d_wp, float *d_y, int N, int Cin, int H, int W, int Cout,
                                   void *stream);

/* 4x4 convolution, stride 1, zero padding `pad`, on the same DMA-staged MFMA kernel (KS = 4 instantiation): the inner
 * layers of the 70x70 PatchGAN (models/networks.py:445-500 NLayerDiscriminator: Conv2d(ndf*m, ndf*2m, 4, 1, 1)).
 * d_x [N][H][W][Cin] bf16, d_w [16][Cout][Cin] bf16 (tap 4r+s), d_y [N][H+2pad-3][W+2pad-3][Cout] bf16; Cin, Cout
 * multiples of 32. Data gradient = the same entry point on dy with flipped + transposed weights and pad' = 3 - pad. */
int octa_conv4x4_nhwc_fwd(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                          void *stream);
/* Weight gradient of that layer with padding 1: d_dw [16][Cout][Cin] float32 (overwritten) from d_x [N][H][W][Cin] and
 * d_dy [N][H-1][W-1][Cout] (bf16); the KS = 4 instantiation of the 3x3 weight-gradient kernel (16 accumulator tiles per wave). */
int octa_conv4x4_nhwc_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin, int Cout,
                            void *stream);

/* 1x1 convolution head with ONE output channel and bias (DynUNet's UnetOutBlock, 32 -> 1; networks.py:6 / MONAI):
 * y[p] = bias + sum_c x[p][c] w[c] over NHWC bf16 pixels (y bf16 [npix]); backward: dx[p][c] = dy[p] w[c] (bf16),
 * dw[c] = sum_p x[p][c] dy[p], db = sum_p dy[p] (float32, overwritten). HBM-bound streaming kernels. */
int octa_head1_nhwc_fwd(octa_ctx *ctx, const void *d_x, const float *d_w, float bias, int64_t npix, int C, void *d_y, void *stream);
int octa_head1_nhwc_bwd(octa_ctx *ctx, const void *d_x, const void *d_dy, const float *d_w, int64_t npix, int C, void *d_dx,
                        float *d_dw, float *d_db, void *stream);
/* The same forward with the bias read from device memory (d_bias float32[1] or NULL): the training step never reads the
 * parameter back to the host. */
int octa_head1_nhwc_fwd_b(octa_ctx *ctx, const void *d_x, const float *d_w, const float *d_bias, int64_t npix, int C, void *d_y, void *stream);

/* Every KxK convolution weight of a network into both layouts the MFMA kernels read, in ONE launch per optimiser step
 * (the master copies stay float32 torch parameters in the reference's state-dict layout, models/networks.py /
 * MONAI DynUNet: [Cout][Cin][K][K]). d_table: int64 [L][8] in device memory, one row per layer =
 * {src pointer, off_fwd, off_dg, A, B, BP, KK, kind}: kind 0 = Conv2d weight float32 [A][B][K][K] (KK = K*K);
 * kind 1 = ConvTranspose2d(kernel 2, stride 2) weight [A][B][2][2] read as the 3x3 kernel whose taps r = 0 / s = 0 are
 * zero (KK = 9). Written to d_dst (bf16, element offsets): [KK][A][BP] at off_fwd and, taps reversed, [KK][BP][A] at
 * off_dg, both in the slice-major storage order described at octa_conv3x3_nhwc_fwd (a layer whose A is no multiple of 16 keeps its
 * data-gradient pack tap-major: no MFMA kernel reads it); columns B..BP-1 (channel padding to the kernels' multiple of 32) are zero. */
int octa_pack_conv_weights(octa_ctx *ctx, const int64_t *d_table, int L, void *d_dst, void *stream);

int octa_conv3x3_nhwc_wgrad2(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H, int W,
                             int Cin, int Cout, int tap_mask, void *stream);   /* wgrad with the virtual input concatenation and
                                                                                  the tap mask of _fwd2 (unmasked taps of d_dw
                                                                                  are left zero) */

/* ---- GPU data-augmentation chain between rasteriser and network --------------------
 * Replaces the MONAI CPU transforms of configs/config_ves_seg-S.yml:42-102 (registry data/data_transforms.py:587-611)
 * for a batch resident in HBM.
 * octa_resize_bilinear: ScaleIntensityd + Resized(mode bilinear): in [B][h][w] (in_dtype 0 = uint8, 1 = float32) ->
 *   float32 [B][H][W] with torch's upsample_bilinear2d arithmetic (align_corners False); every source value is mapped
 *   v * d_mul[b] + d_add[b] first (NULL = identity) -- ScaleIntensity's per-image min/max map.
 * octa_flip_rot90_rotate: RandFlipd (both axes, d_flip[b] != 0) -> RandRotate90d (d_rot_k[b] quarter turns, torch.rot90) ->
 *   RandRotated (d_angle[b] radians, bilinear, zeros padding, affine_grid + grid_sample arithmetic with align_corners
 *   False) -> optional AsDiscreted (v >= threshold ? 1 : 0) on square float32 images [B][N][N]; out must not alias in.
 */
/* Noise model of the GAN configs, on batches in HBM (reference data/data_transforms.py):
 * octa_background_noise: AddRandomBackgroundNoised (:498-516): max(img, noise * u) over n float32 pixels, u = the float64 factors of
 *   numpy's uniform stream; the product is formed in float64 like the reference's promoted tensor; d_out_f64 / d_out_f32 (either may be
 *   NULL) receive the float64 result and its float32 cast.
 * octa_speckle_brightness: SpeckleBrightnesd (:25-42): per image a 9x9 control grid (d_grid9 [B][81], values in [0.5, 1)) is bilinearly
 *   upsampled (torch arithmetic, align_corners False), R = C - u (1 - C) with d_u [B][H][W], out = img * R, divided by its maximum, then
 *   shifted by its minimum. d_minmax: int32 [B][2] scratch. */
int octa_background_noise(octa_ctx *ctx, const float *d_img, const float *d_noise, const double *d_u, int64_t n, double *d_out_f64,
                          float *d_out_f32, void *stream);
int octa_speckle_brightness(octa_ctx *ctx, const float *d_img, const float *d_grid9, const float *d_u, int B, int H, int W, float *d_out,
                            int *d_minmax, void *stream);
int octa_resize_bilinear(octa_ctx *ctx, const void *d_in, int in_dtype, int B, int h, int w, float *d_out, int H, int W,
                         const float *d_mul, const float *d_add, void *stream);
/* Adjoint of octa_resize_bilinear (float32, identity intensity map): d_dy [B][H][W] -> d_dx [B][h][w], the gradient of the bilinear
 * up-sampling GanSegModel applies in front of its segmentor (models/gan_seg_model.py:61,101-106: F.interpolate(x, upshape, "bilinear")). */
int octa_resize_bilinear_bwd(octa_ctx *ctx, const float *d_dy, int B, int h, int w, int H, int W, float *d_dx, void *stream);
int octa_flip_rot90_rotate(octa_ctx *ctx, const float *d_in, float *d_out, int B, int N, const float *d_angle, const int *d_rot_k,
                           const int *d_flip, float threshold, int use_threshold, void *stream);

/* First layer of the U-Net (UnetBasicBlock.conv1 of the input block: ONE input channel -> Cout in {8, 16, 32, 64}, 3x3,
 * padding 1, stride 1): d_x [N][H][W] bf16, d_w float32 [Cout][9] (tap = 3r + s), d_y [N][H][W][Cout] bf16; the weight
 * gradient d_dw float32 [Cout][9] (overwritten). Streaming kernels: 9 multiply-adds per output are not matrix-core work. */
int octa_conv3x3_c1_fwd(octa_ctx *ctx, const void *d_x, const float *d_w, void *d_y, int N, int H, int W, int Cout, void *stream);
int octa_conv3x3_c1_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cout, void *stream);
/* octa_conv3x3_c1_fwd with the InstanceNorm statistics of its result accumulated in the kernel's epilogue: d_stat double [nslot][N][Cout][2]
 * (pre-zeroed; sum and sum of squares of the bf16-rounded values per image and channel, spread over the slots -- the contract of
 * octa_conv3x3_nhwc_fwd7, consumed by octa_instnorm_lrelu_nhwc_fwd_s); d_stat NULL = octa_conv3x3_c1_fwd. W <= 3840. */
int octa_conv3x3_c1_fwd2(octa_ctx *ctx, const void *d_x, const float *d_w, void *d_y, int N, int H, int W, int Cout, double *d_stat, int nslot,
                         void *stream);

/* ---- DiceBCELoss in one pass each way --------------------
 * utils/losses.py:111-121: (DiceLoss(sigmoid=True) + BCEWithLogitsLoss) / 2 over logits [B][n] (dtype 0 = float32,
 * 1 = bfloat16) and float32 labels [B][n]. Forward fills d_sums double[B][4] = (sum p*y, sum p, sum y, sum bce), p =
 * sigmoid(logit); the scalar is dice = mean_b(1 - (2 S_py + nr) / (S_p + S_y + dr)), bce = sum_b S_bce / (B n), loss =
 * (dice + bce) / 2. Backward writes dloss/dlogits (same dtype as the logits) scaled by d_grad_out[0]. */
int octa_dice_bce_fwd(octa_ctx *ctx, const void *d_logits, int dtype, const float *d_y, int B, int64_t n, double *d_sums, void *stream);
/* The scalar loss from those sums: (mean_b(1 - (2 s0 + nr) / (s1 + s2 + dr)) + sum_b s3 / (B n)) / 2 in double, stored as float32 at d_loss --
 * utils/losses.py:111-121's (dice + bce) / 2 without the ten scalar torch launches between a step's forward and backward pass. */
int octa_dice_bce_finish(octa_ctx *ctx, const double *d_sums, int B, int64_t n, double smooth_nr, double smooth_dr, float *d_loss, void *stream);
int octa_dice_bce_bwd(octa_ctx *ctx, const void *d_logits, int dtype, const float *d_y, int B, int64_t n, const double *d_sums,
                      const float *d_grad_out, float smooth_nr, float smooth_dr, void *d_dlogits, void *stream);

/* ---- convolutions with ONE channel on one side: the GAN networks' stems and heads ----
 * Replace nn.Conv2d(1, 64, 7) / nn.Conv2d(64, 1, 7) of ResnetGenerator (models/networks.py:360-368, behind ReflectionPad2d(3)) and
 * nn.Conv2d(1, 64, 4, 1, 1) / nn.Conv2d(512, 1, 4, 1, 1) of NLayerDiscriminator (:433-442) -- forward, data gradient and weight
 * gradient -- which the reference leaves to the vendor library (im2col + GEMM + col2im). Stride 1, zero padding `pad`, K = 4 or 7,
 * C (the wide side) a multiple of 64. s / squeeze-out: bf16 [N][H][W]; a / expand-out: bf16 [N][H][W][C]; w, g: float32 [C][K*K]
 * (Conv2d.weight of either shape, contiguous); bias float32 or NULL. With t = ky*K + kx, tf = flip ? K*K-1-t : t:
 *   expand : out[n][y][x][c] = lrelu_slope(bias[c] + sum_t s[n][y+ky-pad][x+kx-pad] * w[c][tf]),  out extent Hs + 2 pad - K + 1
 *   squeeze: out[n][y][x]    = bias[0] + sum_t sum_c a[n][y+ky-pad][x+kx-pad][c] * w[c][tf]
 *   wgrad  : g[c][tf]        = sum_{n,y,x} a[n][y][x][c] * s[n][y+ky-pad][x+kx-pad] (zero outside s);  asum[c] = sum a (or NULL)
 * so that for y = conv(x, w, pad):  1 -> C layer: forward = expand(x), dx = squeeze(dy, flip, K-1-pad), dw = wgrad(a = dy, s = x, pad);
 * C -> 1 layer: forward = squeeze(x), dx = expand(dy, flip, K-1-pad), dw = wgrad(a = x, s = dy, flip, K-1-pad).
 * wgrad needs octa_thinconv_wgrad_scratch_floats(N, Ha, C, K) floats of device scratch (per-block partial sums: deterministic). */
int octa_thinconv_expand(octa_ctx *ctx, const void *d_s, const void *d_w, const void *d_bias, void *d_out, int N, int Hs, int Ws, int C, int K,
                         int pad, int flip, float slope, void *stream);
int octa_thinconv_squeeze(octa_ctx *ctx, const void *d_a, const void *d_w, const void *d_bias, void *d_out, int N, int Ha, int Wa, int C, int K,
                          int pad, int flip, void *stream);
/* (octa_thinconv_squeeze also takes K = 3 with 8, 16, 32 or 64 channels: the data gradient of the one-channel 3 x 3 first layer whose forward and
 * weight gradient are octa_conv3x3_c1_fwd2 / octa_conv3x3_c1_wgrad -- dx = squeeze(dy, w, K = 3, pad = 1, flip = 1) -- for a segmentor whose
 * input image is another network's output, models/gan_seg_model.py:147-149.) */
long long octa_thinconv_wgrad_scratch_floats(int N, int Ha, int C, int K);
/* LeakyReLU' on a gradient by the sign of the layer's bf16 OUTPUT: out[i] = y[i] > 0 ? dy[i] : bf16(dy[i] * slope), n elements, 16-byte aligned
 * tensors. The expand layer fuses LeakyReLU into its forward (the PatchGAN stem, models/networks.py:433-436); this is the first step of its backward. */
int octa_lrelu_bwd_bf16(octa_ctx *ctx, const void *d_y, const void *d_dy, void *d_out, int64_t n, float slope, void *stream);
int octa_thinconv_wgrad(octa_ctx *ctx, const void *d_a, const void *d_s, void *d_scratch, void *d_g, void *d_asum, int N, int Ha, int Wa, int Hs,
                        int Ws, int C, int K, int pad, int flip, void *stream);

/* ---- anti-aliased resampling and reflection pad of the GAN networks ----
 * Replace models/networks.py:244-262 (Upsample: ReplicationPad2d(1) + depth-wise conv_transpose2d with the
 * [1 3 3 1]^2/64*4 filter, stride 2, crop), :264-289 (Downsample: ReflectionPad2d(1) + depth-wise conv2d with the
 * [1 2 1]^2/16 filter, stride 2) and the nn.ReflectionPad2d(pad) of ResnetBlock / ResnetGenerator (:366-368, :404-421),
 * each as one streaming kernel per direction. Tensors are planes [B][H][W][C], C innermost: an NCHW tensor is passed as
 * B*C planes with C = 1, an NHWC tensor as it is. dtype 0 = float32, 1 = bfloat16 (fp32 arithmetic, one rounding).
 * H, W, C always describe the UNPADDED / LARGER-INPUT side named below; _bwd takes the gradient of the forward's
 * output in d_in and writes the gradient of the forward's input to d_out (gather form, deterministic).
 *   reflect_pad : [B][H][W][C] -> [B][H+2pad][W+2pad][C], 1 <= pad < min(H, W)
 *   blur_down   : [B][H][W][C] -> [B][(H-1)/2+1][(W-1)/2+1][C]
 *   blur_up     : [B][H][W][C] -> [B][2H][2W][C] */
int octa_reflect_pad_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, int pad, void *stream);
int octa_reflect_pad_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, int pad, void *stream);
int octa_blur_down_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream);
int octa_blur_down_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream);
int octa_blur_up_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream);
int octa_blur_up_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream);

/* ---- inference post-processing --------------------
 * RemoveSmallObjects(min_size) of the configs' post_processing lists (configs/config_ves_seg-S.yml:103-113; MONAI ->
 * skimage.morphology.remove_small_objects) for a batch of masks in HBM: d_in uint8 [B][H][W] (non-zero = foreground);
 * every connected component (connectivity 1 = 4-neighbourhood, 2 = 8-neighbourhood) with fewer than min_size pixels is
 * removed; d_out uint8 [B][H][W] gets on_value on the surviving pixels, 0 elsewhere (may alias d_in). */
int octa_remove_small_objects(octa_ctx *ctx, const uint8_t *d_in, int B, int H, int W, int min_size, int connectivity,
                              uint8_t on_value, uint8_t *d_out, void *stream);

/* ---- N1-N4: space-colonisation vessel-graph simulator --------------------
 * Replaces, for B independent samples advanced in lock-step on the GPU:
 *   vessel_graph_generation/greenhouse.py:57-137 (Greenhouse.develop_forest) with
 *   :319-341 sample_oxygen_sinks, :343-366 assign_attraction_points_to_node,
 *   :157-307 grow_vessels, :99-123 O2/CO2 bookkeeping, :139-155 expansion;
 *   forest.py:68-181 (stumps), simulation_space.py:36-110, arterial_tree.py:174-184 (Murray),
 *   element_mesh.py:87-232 (KD_Tree query semantics incl. cKDTree result order),
 *   and the edge list order of generate_vessel_graph.py:43-56.
 * Sample k behaves like `random.seed(py_seeds[k]); np.random.seed(np_seeds[k])` followed by
 * the reference's main() (generate_vessel_graph.py:24-39).
 * The only arithmetic left to the caller is the leaf-bifurcation geometry that the reference
 * computes with numpy/LAPACK (np.cov + np.linalg.eig, greenhouse.py:221-233; the sign of the dgeev
 * eigenvector is part of the result): the library hands batches of requests to `bif`.
 */
typedef struct octa_sim octa_sim;

typedef struct {
    double param_scale, d, r;             /* Greenhouse: param_scale, d, r                 */
    double faz_radius_mean, faz_radius_std; /* FAZ_radius_bound                             */
    double rotation_radius;
    double faz_center[2];
    double size[3];                       /* SimulationSpace no_voxel_x/y/z                 */
    int n_trees;                          /* Forest.N_trees                                 */
    int walls[4];                         /* source_walls x0, x1, y0, y1 (see n_source_walls) */
    int n_modes;
    /* per mode: I, N, eps_n, eps_s, eps_k, delta_art, delta_ven, gamma_art, gamma_ven, phi, omega, kappa, delta_sigma */
    double modes[8][13];
    /* Forest.type: 0 = 'stumps' (roots on the lateral walls, forest.py:68-181), 1 = 'nerve' (all roots inside the optic-nerve disc,
     * forest.py:38-66); Greenhouse nerve_center / nerve_radius as in the YAML (divided by param_scale inside). The disc is also cut
     * out of the sink-sampling mask when it lies inside the field of view (simulation_space.py:48-50). */
    int forest_type;
    double nerve_center[2];
    double nerve_radius;
    /* SimulationSpace.oxygen_sample_geometry_path (simulation_space.py:29-34,70-76): the loaded .npy mask, one byte per voxel in C
     * order [g0][g1][g2] (non-zero = sinks may be sampled there; each dimension 1..65535, at most 2^26 voxels), or NULL for the
     * analytic FAZ mask. With a geometry the space's extent is the mask's shape / its largest dimension (size[] is ignored), a
     * candidate sink is (valid voxel + U[0,1)^3) / that dimension, kept when the voxel it maps back to is set, and the stumps' wall
     * positions come from a random valid voxel of the wall's face (face 0 for every wall: `shape[axis] - 1` of the normalised
     * shape, simulation_space.py:71). The bytes are copied by octa_sim_create. */
    const uint8_t *geometry;
    int geometry_shape[3];
    /* n_source_walls > 0: the enabled source walls in the order of the configuration's mapping -- the reference draws the wall of
     * every tree by position in that list (forest.py:81-91) -- as 0..5 = x0 x1 y0 y1 z0 z1; walls[] is then ignored. The z walls
     * (forest.py:153-181) need a geometry: without one the reference fails (simulation_space.py:82-87). 0: walls[] in x0..y1 order. */
    int n_source_walls;
    int source_walls[6];
} octa_sim_config;

#define OCTA_BIF_MAX_ATTS 256
typedef struct {
    int sample, n;                        /* sample index in the batch, number of attractors */
    double pos[3];                        /* position of the bifurcating leaf               */
    double r, kappa, d;                   /* child radius, bifurcation exponent, segment length */
    double atts[OCTA_BIF_MAX_ATTS * 3];   /* the attractors kept by the angle filter, in order */
} octa_bif_request;

/* Fill out6[i] = (p_new_1.xyz, p_new_2.xyz) for every request (greenhouse.py:205-233). */
typedef void (*octa_bif_fn)(int n_req, const octa_bif_request *reqs, double *out6, void *user);

/* Native bifurcation service: an octa_bif_fn that evaluates requests in C++ through the BLAS/LAPACK library
 * numpy itself is linked against (path of numpy.libs/libscipy_openblas64_*.so), falling back to `fallback`
 * (the numpy callback) per request where it cannot guarantee numpy's bits (complex eigenpairs, unknown kappa).
 * kappas/cs/sn: the bifurcation exponents of the config's modes with cos/sin of the Murray half-angle as numpy
 * computes them. The Python host validates the native path against the numpy formula before using it. */
int octa_bif_native_init(const char *blas_path, int n_kappa, const double *kappas, const double *cs, const double *sn,
                         octa_bif_fn fallback, void *fallback_user);
void octa_bif_native(int n_req, const octa_bif_request *reqs, double *out6, void *user);
int octa_bif_native_counts(int64_t *h_out2);

int octa_sim_create(octa_ctx *ctx, const octa_sim_config *cfg, int B, octa_sim **out);
/* The same with the kernel build named by the caller: 0 = chosen from the configuration (octa_sim_create; OCTA_SIM_BUILD overrides),
 * 1 = the default build (3 x 3 mm^2 capacities), 2 = the wide-field build (the reference's 12 x 12 mm^2 notebook run,
 * example_custom_vessel_simulation.ipynb:138-156). The Python host re-runs a batch that outgrew the default capacities with 2. */
int octa_sim_create_ex(octa_ctx *ctx, const octa_sim_config *cfg, int B, int build, octa_sim **out);
void octa_sim_destroy(octa_sim *sim);

/* Ordering a rasterisation behind the NEXT launch of the persistent kernel, on the device (round 6; csrc/order.hip, csrc/sim_api.cpp).
 * Every persistent-kernel launch of the process takes a ticket; octa_sim_launch_count() is the last ticket issued (after a caller's own
 * octa_sim_run returned under the generators' common lock: that run's ticket). octa_order_wait_launch enqueues a one-wave gate kernel on
 * `stream`: what is enqueued behind it runs once the launch with `ticket` has all its workgroups on the GPU (they sign in as they start)
 * plus settle_us, or after timeout_us. Why: nothing co-resides with the simulator's workgroups, and a rasterisation dispatched while a
 * launch is being placed takes CUs it then keeps -- the launch lasts two samples. d_out3 (optional, device int[3]):
 * 1 = resident / 2 = timed out, 100 MHz ticks waited, workgroups signed in. The reference has no counterpart: its samples run in separate
 * processes on CPU cores (generate_vessel_graph.py:112-129). */
long long octa_sim_launch_count(void);
int octa_order_wait_launch(octa_ctx *ctx, long long ticket, int timeout_us, int settle_us, int *d_out3, void *stream);

/* Run all iterations for B samples. Synchronous (the bifurcation service needs the host). Returns 0, -1 (runtime
 * failure), -2 (bad arguments) or -3 (a sample set error bits: capacity, or 0x800 = the host did not answer in time). */
int octa_sim_run(octa_sim *sim, const uint32_t *h_np_seeds, const uint64_t *h_py_seeds, octa_bif_fn bif, void *user,
                 void *stream);

/* The same run for samples whose random generators stand where a CALLER's do -- the object API of the reference (Greenhouse(...),
 * Forest(...) x 2, develop_forest(), generate_vessel_graph.py:24-39) draws the FAZ radius and the stump nodes from the global numpy /
 * CPython generators in its constructors; the adapters do the same in Python and hand over: h_faz_radius [B]; h_stumps
 * [B][2 forests][2 * N_trees][3] (root, stump child per tree; arterial forest first); the generators' MT19937 states afterwards,
 * [B][625] each (624 words + position, as np.random.get_state() / random.getstate() report them). octa_sim_np_state returns numpy's
 * state after the run (CPython's then stands octa_sim_stats' `random.uniform draws` column of random.random() draws further). */
int octa_sim_run_states(octa_sim *sim, const double *h_faz_radius, const double *h_stumps, const uint32_t *h_np_states,
                        const uint32_t *h_py_states, octa_bif_fn bif, void *user, void *stream);
int octa_sim_np_state(octa_sim *sim, int sample, uint32_t *h_state625);

/* After octa_sim_run: per-sample edge offsets (h_edge_off[B+1]) and arterial edge counts (h_n_art[B]). */
int octa_sim_edge_offsets(octa_sim *sim, int64_t *h_edge_off, int64_t *h_n_art);

/* Edge list [total][7] doubles (node xyz, parent xyz, radius), arterial trees then venous, BFS per
 * tree -- the CSV row order of generate_vessel_graph.py:59-66. h_edges: host buffer. */
int octa_sim_export_edges(octa_sim *sim, double *h_edges);

/* The same edge list written on the DEVICE (round 3): d_edges [edge_off[B]][7] float64 in HBM, filled by one workgroup per
 * (sample, forest) that walks the trees level by level (children in parent order, child 0 before child 1 -- anytree's level
 * order, generate_vessel_graph.py:43-66) on `stream`. The rows equal octa_sim_export_edges' bit for bit; the list can go straight
 * to octa_rasterize_2d without crossing PCIe. */
int octa_sim_export_edges_device(octa_sim *sim, double *d_edges, void *stream);

/* Per-iteration statistics of the last run, h_trace int32 [B][n_iter][4]: arterial nodes, O2 sinks, venous nodes, CO2 sources at the
 * end of every iteration -- Greenhouse.art_nodes_per_step / oxys_per_step / ven_nodes_per_step / co2_per_step without their initial
 * entry (greenhouse.py:129-134). Rows of samples that stopped on an error are undefined from the failing iteration on. */
int octa_sim_trace(octa_sim *sim, int32_t *h_trace);

/* Per-sample statistics, h_stats[B][32] int64: error bits, random.uniform draws, Murray steps,
 * bifurcations, re-speculated inter-nodes, arterial nodes, venous nodes, FAZ radius bits, then 16
 * per-phase device timers (100 MHz ticks: 0 sample, 1 assign-art, 2 speculate-art, 3 ordered-art,
 * 4 O2->CO2, 6 assign-ven, 7 speculate-ven, 8 ordered-ven, 9 CO2 removal), then 8 timers of the kd-order build. */
int octa_sim_stats(octa_sim *sim, int64_t *h_stats);

/* When each sample held a CU (persistent form): h_spans[B][2] = the GPU's 100 MHz wall clock when a workgroup first took the sample
 * and when it last left it. The clock is common to all launches on the device, so spans of concurrent launches can be laid over
 * each other: sum of spans / (CUs x window) = the share of CU time the simulator used (bench.py's cu_time_used). */
int octa_sim_spans(octa_sim *sim, int64_t *h_spans);

/* Timing of the last octa_sim_run, h_out8: [0] sum of launch-A kernel durations (ms, HIP events on
 * the launch stream), [1] launches A, [2] sum of launch-B durations (ms), [3] launches B, [4] wall ms
 * of the iteration loop, [5] host ms spent in the bifurcation callback, [6] requests served,
 * [7] bytes of HBM held by the simulator. */
int octa_sim_timing(octa_sim *sim, double *h_out8);

/* Host-side record of the mailbox service of the last octa_sim_run (persistent form), h_out5 (five doubles): [0] tickets
 * served while kernels ran, [1] longest single pass of the service loop (ms: how long the thread was away from the mailbox),
 * [2] extra kernel launches because workgroups had parked, [3] parked workgroups served at kernel boundaries, [4] longest
 * bifurcation callback (ms). Environment read by octa_sim_create:
 *   OCTA_SIM_PARK_MS (default 3): a workgroup that has waited this long for its answer records its resume point and leaves the
 *     kernel; the host serves it at the kernel boundary and launches again. 0 = never park (round-1 behaviour: wait up to
 *     OCTA_SIM_MAIL_TIMEOUT_MS, default 30000, then fail the sample with error bit 0x800);
 *   OCTA_SIM_TEST_HOST_STALL_MS (test hook): the service thread sleeps once while a ticket is pending;
 *   OCTA_SIM_SPIN_SCANS: idle scans before the service thread sleeps between scans (4096; 256 when WORLD_SIZE > 1). */
int octa_sim_service_stats(octa_sim *sim, double *h_out5);

/* Launch geometry of the simulator kernels (build-time constants), h_out4 (four ints): [0] threads per workgroup, [1] workgroups
 * (= samples) resident per CU, [2] bytes of LDS per workgroup, [3] workgroups per launch of the persistent kernel on a device
 * with `num_cus` compute units (pass 0 for the 256 of an MI355X). One workgroup advances one sample at a time
 * (greenhouse.py:57-137 runs one sample per worker process). */
int octa_sim_geometry(int num_cus, int *h_out4);

/* 1 when the simulator was bound to the wide-field build (32-bit indices, 64-bit kd elements, per-sample tables in HBM instead of
 * LDS: the reference's 12 x 12 mm^2 notebook configuration, example_custom_vessel_simulation.ipynb:138-156), 0 for the default build.
 * Chosen by octa_sim_create from the configuration; OCTA_SIM_BUILD=large / default overrides. */
int octa_sim_is_large(const octa_sim *sim);

/* Final O2 / CO2 fields of one sample (host buffers, capacity in points); returns counts. */
int octa_sim_fields(octa_sim *sim, int sample, double *h_oxy, int64_t cap_oxy, int64_t *n_oxy, double *h_co2,
                    int64_t cap_co2, int64_t *n_co2);

/* Known-answer hook for the device restatement of scipy.spatial.cKDTree's build order (the order
 * `query_ball_point` reports neighbours in, greenhouse.py:101-107): host points [n][3] (n <= 13312),
 * optional host flags need[n] (NULL = every rank is read) -> host tree.indices as int32[n]. With `need`
 * only the relative order of flagged points is defined. */
int octa_sim_kat_kd_order(octa_ctx *ctx, const double *h_pts, int64_t n, const uint8_t *h_need, int32_t *h_indices);

#ifdef __cplusplus
}
#endif
#endif /* OCTA_HIP_H */
