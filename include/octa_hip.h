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

/* ---- N7: Floyd-Steinberg binarisation -----------------------------------
 * Replaces: Pillow Image.convert("1") as called at visualize_vessel_graphs.py:99
 * (label PNGs). d_in/d_out: [B][H][W] uint8; output values are 0 or 255.
 */
int octa_fs_dither(octa_ctx *ctx, int B, const uint8_t *d_in, int W, int H, uint8_t *d_out, void *stream);

/* ---- element-wise max of two uint8 images -------------------------------
 * Replaces: np.maximum(art_mat, ven_mat) at generate_vessel_graph.py:83.
 */
int octa_max_u8(octa_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, uint8_t *d_out, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OCTA_HIP_H */
