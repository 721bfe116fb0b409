/*
 * libh3d -- C ABI of the MI355X (gfx950) generator hot path.
 *
 * This is the drop-in boundary: plain C, device pointers + explicit sizes, a
 * HIP stream handle, int return codes.  No torch types, no exceptions across
 * the boundary, no allocation: the caller owns every buffer (inputs, outputs,
 * workspaces).  All pointers are DEVICE pointers unless a comment says HOST.
 * Every entry point is re-entrant and launches on the stream it is given.
 *
 * Return value: 0 on success, negative H3D_E* on failure; h3d_last_error()
 * returns a thread-local human readable message for the last failure.
 *
 * Each function cites the reference interface it replaces (file:line under
 * the upstream repository root).
 */
#ifndef H3D_H
#define H3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H3D_VERSION 101            /* major*10000 + minor*100 + patch */

#define H3D_OK            0
#define H3D_EINVAL       -1        /* bad argument (shape, alignment, flag)        */
#define H3D_EUNSUPPORTED -2        /* valid request this build has no kernel for   */
#define H3D_ELAUNCH      -3        /* HIP launch / runtime error                   */

typedef void* h3d_stream_t;        /* hipStream_t */

int h3d_version(void);
const char* h3d_last_error(void);
/* Number of compute units / name of the current HIP device (HOST out buffers). */
int h3d_device_info(int* n_cu, char* name, int name_len);

/* ------------------------------------------------------------------------
 * A6  volume integration  == lib/generators/volume_rendering.py:12-56
 *     (ray_integration(input, z_vals, device, noise_std, last_back, white_back, clamp_mode))
 * field   [n_rays, S, C+1] fp32, density in the last channel
 * z_vals  [n_rays, S]
 * noise   [n_rays, S] already multiplied by noise_std, or NULL
 * feats   [n_rays, C]   depth [n_rays]   weights [n_rays, S]
 * clamp_mode: 0 = relu, 1 = softplus.  fill_mode is not supported (None in every caller).
 */
int h3d_ray_integrate(const float* field, const float* z_vals, const float* noise,
                      float* feats, float* depth, float* weights,
                      int64_t n_rays, int S, int C,
                      int clamp_mode, int last_back, int white_back, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Hierarchical (coarse + fine) sampling, hierarchical_sample=True in Map3DGenerator.render
 *     (lib/generators/map3d_generator.py:449-509; off in every shipped config).
 * h3d_sample_pdf == sample_pdf (lib/generators/volume_rendering.py:261-303) with the uniform draws passed in:
 *     bins [n_rays, n_bins], weights [n_rays, n_bins-1], u [n_rays, n_samples] -> samples [n_rays, n_samples]
 * h3d_ray_points: points[b, r*S+s] = origin[b] + dirs[b, r] * z_vals[b, r, s]   (map3d_generator.py:464-466)
 * h3d_merge_samples == cat([fine, coarse]) + sort by depth + gather (map3d_generator.py:504-509): fine [n_rays,Sf,C1],
 *     coarse [n_rays,Sc,C1], fine_z [n_rays,Sf], coarse_z [n_rays,Sc] -> out [n_rays,Sf+Sc,C1], out_z [n_rays,Sf+Sc]
 *     (stable: equal depths keep the fine-before-coarse order of the concatenation).
 */
int h3d_sample_pdf(const float* bins, const float* weights, const float* u, float* samples, int64_t n_rays, int n_bins,
                   int n_samples, float eps, h3d_stream_t stream);
int h3d_ray_points(const float* origin, const float* dirs, const float* z_vals, float* points, int B, int64_t R, int S,
                   h3d_stream_t stream);
int h3d_merge_samples(const float* fine, const float* coarse, const float* fine_z, const float* coarse_z, float* out,
                      float* out_z, int64_t n_rays, int Sf, int Sc, int C1, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A3  ray set-up == lib/generators/volume_rendering.py:86-110 (get_initial_rays_weak_perspective),
 *     :124-130 (perturb_points), :133-170 (transform_sampled_points)
 * focals, scales [B]; cam2world [B,4,4] row-major; jitter [B,R,S] U(0,1) or NULL
 * points [B,R*S,3] world space; z_vals [B,R,S]
 */
int h3d_ray_setup(const float* focals, const float* scales, const float* cam2world, const float* jitter,
                  float* points, float* z_vals,
                  int B, int render_h, int render_w, int S, float ray_start, float ray_end,
                  h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A4  SMPL geometry features == lib/components/smpl.py:210-249 (get_geo_features)
 * points [B,N,3]; joints [B,24,3]; vertices, tpose_vertices [B,V,3];
 * vertex_ik [B,V,16]  = einsum(lbs_weights, inverse(fk_matrices)) (smpl.py:217-218; a [B,V,24]x[B,24,16]
 *                       product the host does once per pose with a library GEMM)
 * geo [B,N,geo_stride] (31 values written per point, order per legacy_mode), nn_index [B,N] int32 or NULL
 */
int h3d_geo_features(const float* points, const float* joints, const float* vertices,
                     const float* tpose_vertices, const float* vertex_ik,
                     float* geo, int32_t* nn_index,
                     int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream);

/* The K = 1 nearest-vertex search of h3d_geo_features alone (same filter + exact refine, the same arg-min bit for bit; replaces
 * pytorch3d.ops.knn_points at lib/components/smpl.py:220): nn_index [B, N] int32 is the only output.  With it the fused field
 * kernels build the 31 features themselves (h3d_render_fused_x2_geo / _x3_geo below): the [B, N, 31] tensor is never written. */
int h3d_nearest_vertex(const float* points, const float* vertices, int32_t* nn_index, int B, int64_t N, int V,
                       h3d_stream_t stream);

/* Round 4: the same search with chunk pruning on a spatially sorted mesh (same indices and features, bit for bit; ~4/5 of the
 * point-vertex work skipped).  h3d_mesh_sort writes, per pose, the vertices in Morton order with their original indices
 * ([Vpad] float4: x, y, z, index bits; Vpad = V rounded up to 64) followed by the bounding spheres of the chunks of 64
 * ([Vpad / 64] float4: centre, radius rounded up) into `workspace` (h3d_mesh_sort_bytes(B, V) bytes, 16-byte aligned; one
 * workgroup per pose, bitonic sort in LDS, V <= 16384).  The _sorted entry points take that workspace in place of `vertices`;
 * ties between equidistant vertices still go to the smallest ORIGINAL index (pytorch3d.ops.knn_points contract,
 * lib/components/smpl.py:220).  The unsorted entry points above stay for meshes outside these limits. */
int64_t h3d_mesh_sort_bytes(int B, int V);
int h3d_mesh_sort(const float* vertices, void* workspace, int B, int V, h3d_stream_t stream);
int h3d_geo_features_sorted(const float* points, const float* joints, const void* sorted_mesh,
                            const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                            int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream);
int h3d_nearest_vertex_sorted(const float* points, const void* sorted_mesh, int32_t* nn_index, int B, int64_t N, int V,
                              h3d_stream_t stream);
/* The same search for points that are the samples of a render grid, points [B, Hr, Wr, S, 3] (what h3d_ray_setup writes;
 * reference: the `transformed_points` of lib/generators/map3d_generator.py:397-415 handed to get_geo_features): a wave takes an
 * 8 x 8 patch of rays x 4 consecutive samples instead of 256 consecutive points -- a compact box, so the bounding-sphere test
 * skips far more of the mesh.  Same indices bit for bit; shapes that do not divide (Hr, Wr % 8, S % 4) use the linear assignment. */
int h3d_nearest_vertex_sorted_rays(const float* points, const void* sorted_mesh, int32_t* nn_index, int B, int Hr, int Wr, int S,
                                   int V, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A5  pose-conditioned FiLM-SIREN == lib/implicit_funcitions/modulated.py:41-75 (COORDCONCATSIREN.forward)
 *
 * Weights are handed over once, pre-packed by h3d_field_pack() into MFMA B-fragment order.
 * h3d_field_pack_size() returns the number of BYTES of the packed blob for hidden width Hd and
 * feature width F.  h3d_field_pack() consumes HOST pointers to the fp32 parameters in the reference's
 * state_dict layout ([out,in] row-major weights) and writes the HOST blob, which the caller uploads.
 */
typedef struct {
    const float *w_coord, *b_coord;      /* first_layer_coord.layer  [Hd,3],[Hd]      */
    const float *w_geo,   *b_geo;        /* first_layer_mod.layer    [Hd,31],[Hd]     */
    const float *w_film[4], *b_film[4];  /* network.k.layer          [Hd,2Hd|Hd],[Hd] */
    const float *w_sigma, *b_sigma;      /* sigma_layer              [1,Hd],[1]       */
    const float *w_color, *b_color;      /* color_layer_sine.layer   [Hd,Hd+3],[Hd]   */
    const float *w_rgb,   *b_rgb;        /* color_layer_linear       [3,Hd],[3]       */
    const float *w_feat,  *b_feat;       /* feature_layer_linear     [F,Hd],[F]       */
} h3d_field_params;

int64_t h3d_field_pack_size(int Hd, int F);
int h3d_field_pack(const h3d_field_params* p, int Hd, int F, void* blob /* HOST */);

/* points [B,N,3], geo [B,N,geo_stride] (31 used), dirs [B,N,3] or NULL (=> (0,0,-1), lock_view_dependence),
 * freq, phase [B,4*Hd] (raw mapping-network outputs; the *15+30 is applied inside, modulated.py:43)
 * out [B,N,F+4] channel order [rgb(3), feat(F), sigma(1)] */
int h3d_neural_field(const void* packed, const float* points, const float* geo, const float* dirs,
                     const float* freq, const float* phase, float* out,
                     int B, int64_t N, int Hd, int F, int geo_stride, float input_scaler,
                     h3d_stream_t stream);

/* Fused A5+A6: the field output never reaches HBM.  N = R*S, samples of a ray contiguous.
 * feats [B*R, F+3] (rgb first), depth [B*R], weights [B*R, S]. */
int h3d_render_fused(const void* packed, const float* points, const float* geo, const float* dirs,
                     const float* freq, const float* phase, const float* z_vals, const float* noise,
                     float* feats, float* depth, float* weights,
                     int B, int R, int S, int Hd, int F, int geo_stride, float input_scaler,
                     int clamp_mode, int last_back, int white_back, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A7  bilinear resize, align_corners=False == F.interpolate at lib/generators/map3d_generator.py:244-245
 * in [B,C,h,w] -> out [B,C,H,W]  (NCHW fp32)
 */
int h3d_bilinear_resize(const float* in, float* out, int B, int C, int h, int w, int H, int W,
                        h3d_stream_t stream);
/* The same resize on channels-last maps (the layout of the differentiable path, lib/generators/differentiable.py):
 * in [B,h,w,C] -> out [B,H,W,C], C a multiple of 4; and its adjoint (the backward of F.interpolate at
 * map3d_generator.py:244-245): dout [B,H,W,C] -> din [B,h,w,C], tmp = B*h*W*C floats of scratch.  The adjoint reads dout
 * once (a running two-row accumulation per output column, no atomics). */
int h3d_bilinear_resize_cl(const float* in, float* out, int B, int h, int w, int H, int W, int C, h3d_stream_t stream);
int h3d_bilinear_resize_cl_bwd(const float* dout, float* tmp, float* din, int B, int h, int w, int H, int W, int C,
                               h3d_stream_t stream);
/* relu(resize(in)) in one pass and its gradient din = resize^T(dout * (out > 0)), `out` being the forward's result (round 6): the ReLU of
 * the SPADEs' shared 128-channel maps (/root/reference/lib/components/map3d_layers.py:170-174, mlp_shared = conv + ReLU, evaluated at ray
 * resolution and resized) and its mask without passes of their own over the [B, H, W, 128 n] tensor. */
int h3d_bilinear_resize_cl_relu(const float* in, float* out, int B, int h, int w, int H, int W, int C, h3d_stream_t stream);
int h3d_bilinear_resize_cl_relu_bwd(const float* dout, const float* out, float* tmp, float* din, int B, int h, int w, int H, int W, int C,
                                    h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A5 / A5+A6 on the f16 matrix cores with split ("x3") operands: every fp32 operand is carried as hi + lo f16
 * halves and a product is hi*hi + hi*lo + lo*hi with fp32 accumulation (fp32-class results, 16/3 of the fp32 MFMA
 * rate).  Same arguments and results as h3d_neural_field / h3d_render_fused; widths up to 256; the fused variant
 * takes S in {8, 16, 32} or a multiple of 32.  Weights are packed by h3d_field_pack_x3 (its own blob format).
 */
int64_t h3d_field_pack_x3_size(int Hd, int F);
int h3d_field_pack_x3(const h3d_field_params* p, int Hd, int F, void* blob /* HOST */);
/* The same blob from parameters that live on the DEVICE (round 6): `p` (a HOST struct) holds DEVICE pointers to contiguous fp32
 * tensors, `blob` is DEVICE memory of h3d_field_pack_x3_size bytes, 16-byte aligned; one memset + one launch on `stream`, no
 * synchronisation.  Bit-identical to h3d_field_pack_x3's blob.  For weights that change every optimiser step: the D step's
 * no-grad generator forward (/root/reference/lib/trainers/phase_trainer.py:355-362) runs the fused render on them. */
int h3d_field_pack_x3_device(const h3d_field_params* p, int Hd, int F, void* blob /* DEVICE */, h3d_stream_t stream);
/* Layout of the blob h3d_field_pack_x3 writes (HOST helper, for tools and tests).  out[0..19] = tiles, k-steps per
 * hidden GEMM, padded width, weight-stream stages per 32-sample step, then BYTE offsets of the nine stream matrices in
 * consumption order (coord [1 k-step], film0 coord half, geo [2], film0 geo half, film1, film2, film3, colour [k-steps
 * + 1: the view direction], feature head), of inv_scale float[9], bias float[7][HdP] (coord, geo, film0..3, colour),
 * b_feat float[HdP], head_w f16 [4][2][k-steps][2][8], head_inv float[4], head_b float[4], and the total size.
 * A stage is [tile][hi|lo][64 lanes][8 f16]; element (lane, e) = s * W[n = 32*tile + (lane&31)][k], s the matrix's
 * power-of-two scale (inv_scale = 1 / (s * input scale)), K in accumulator-register order
 * k = 32*(ks/2) + (e&3) + 8*(2*(ks&1) + (e>>2)) + 4*(lane>>5) for the matrices fed by accumulators and in natural order
 * 16*ks + 8*(lane>>5) + e for coord, geo and the view-direction stage. */
int h3d_field_x3_layout(int Hd, int F, int64_t* out, int n_out);
int h3d_neural_field_x3(const void* packed, const float* points, const float* geo, const float* dirs,
                        const float* freq, const float* phase, float* out,
                        int B, int64_t N, int Hd, int F, int geo_stride, float input_scaler,
                        h3d_stream_t stream);
int h3d_render_fused_x3(const void* packed, const float* points, const float* geo, const float* dirs,
                        const float* freq, const float* phase, const float* z_vals, const float* noise,
                        float* feats, float* depth, float* weights,
                        int B, int R, int S, int Hd, int F, int geo_stride, float input_scaler,
                        int clamp_mode, int last_back, int white_back, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * the discriminator's convolutions == nn.Conv2d(fin, fout, 3, 1, 1) / (fin, fout, 1) of ResBlock
 * (lib/discriminators/unet_discriminators.py:8-72), channels-last fp32, on the bf16 matrix cores with split (x3) operands.
 *   h3d_conv_x3        out[p, o] = bias[o] + sum_{tap, i} W[o, i, tap] x[p + tap, i]   (x [B,H,W,Cin] -> out [B,H,W,Cout]; stride 1,
 *                      zero padding k/2, k = 1 or 3).  Forward and backward-data are the same kernel: backward-data runs it on the
 *                      flipped, transposed weights.  `stream` = the weights packed by the caller in consumption order
 *                      [output block][tap][chunk][k-step][tile][hi|lo][64 lanes][8 bf16]: element (lane, e) of tile nt, k-step ks,
 *                      chunk c, tap t, block ob = W[o = 32*(NT*ob + nt) + (lane&31)][i = 16*(KSC*c + ks) + 8*(lane>>5) + e][t],
 *                      hi = bf16(w), lo = bf16(w - hi); NT, blocks, KSC, chunks from h3d_conv_x3_tiling (lib/components/ops/conv.py
 *                      packs it with a handful of tensor ops on the device).  Cin, Cout multiples of 64.
 *   h3d_conv_wgrad_x3  partial[tap][slice][Co][Ci] = sum_{p in slice} dY[p, Co] * X[p + tap, Ci]; the caller sums the slices.
 */
int h3d_conv_x3_tiling(int Cin, int Cout, int* out /* [4]: NT, blocks, KSC, chunks */);
/* Device-side packer of that stream (one launch): w OIHW fp32 [Cout, Cin, k, k] -> stream (2 * Cout * Cin * k * k bf16);
 * transposed = 1: w is [Cin, Cout, k, k] and the stream is the one of its backward-data convolution. */
int h3d_conv_x3_pack(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_handle);
/* ... for h3d_conv_x3_f16 (AMP tier): the same stream layout with f16 hi + f16 lo planes (|w| < 65504). */
int h3d_conv_x3_pack_f16(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_handle);
int h3d_conv_x3(const float* x, const void* stream, const float* bias /* may be NULL */, float* out, int B, int H, int W,
                int Cin, int Cout, int k, int ldx, int ldo /* row strides in floats: channel slices of wider tensors */,
                h3d_stream_t stream_handle);
int h3d_conv_wgrad_x3(const float* dY, const float* X, float* partial, int B, int H, int W, int Co, int Ci, int k,
                      int ldy, int ldx, int slices, h3d_stream_t stream_handle);

/* AMP tier (round 4; reference: torch.cuda.amp.autocast around both networks, lib/trainers/base_trainer.py:50-51): the same three
 * kernels on f16 activations / gradients.  x, out, dY, X are _Float16 (row strides in ELEMENTS, multiples of 8 for h3d_conv_x3_f16);
 * weights, bias and the weight gradients stay fp32 -- the weight stream is the bf16 hi/lo split of the fp32 weights, so autocast
 * never rounds a weight here -- accumulation is fp32 and an f16 operand's bf16 hi/lo split is exact: the arithmetic is that of the
 * fp32 entry points on exactly representable inputs, with one rounding to f16 at h3d_conv_x3_f16's store. */
int h3d_conv_x3_f16(const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                    int k, int ldx, int ldo, h3d_stream_t stream_handle);
/* The AMP tier with the weights in ONE f16 plane (round 5) -- the reference's own arithmetic under float16 autocast
 * (lib/trainers/base_trainer.py:50-51: autocast rounds weight and activation to f16, fp32 accumulation): one F16 matrix product
 * per weight, half the weight stream of h3d_conv_x3_f16.  h3d_conv_x3_pack_f16x1 writes Cout * Cin * k * k halves: the layout of
 * h3d_conv_x3_pack with the two planes of a stage holding the fragments of two consecutive k-steps. */
int h3d_conv_x3_pack_f16x1(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_handle);
int h3d_conv_x3_f16x1(const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin,
                      int Cout, int k, int ldx, int ldo, h3d_stream_t stream_handle);
/* h3d_conv_x3 / _f16 / _f16x1 (mode 0 / 1 / 2) with a residual connection in the epilogue (round 6):  out = conv(x) + bias + add;
 * `add` [B, H, W, Cout] of the output's element type with row stride lda, added in fp32 before the store -- the skip block's
 * `x = h + x_in` (/root/reference/lib/components/map3d_layers.py:236) without a pass of its own. */
int h3d_conv_x3_add(int mode, const void* x, const void* stream, const float* bias, const void* add, void* out, int B, int H, int W,
                    int Cin, int Cout, int k, int ldx, int ldo, int lda, h3d_stream_t stream_handle);
/* The general entry of the convolution (round 6): h3d_conv_x3_add with the addend optional, the blocking explicit and, optionally,
 * the batch moments of the output.
 *   NT       tiles per output block the stream was packed for (h3d_conv_x3_pack_nt), 0 = the default blocking.  h3d_conv_x3_nt_for
 *            (HOST helper) returns the blocking for P output pixels: the widest whose grid still has a workgroup for every CU -- the
 *            discriminator's low-resolution 512-channel layers (/root/reference/lib/discriminators/unet_discriminators.py:100-118)
 *            launch 8 .. 128 workgroups at the default.  No bit of the result depends on NT.
 *   moments  null, or [ceil(B*H*W / h3d_conv_x3_moment_rows()), 2, Cout] fp32 -- per workgroup of the launch, the column sums of the
 *            output AS STORED (after bias, addend and, in the f16 modes, the rounding) and of its square, taken from the
 *            accumulators.  Their float64 sum over the rows (h3d_rows_sum_f64) is h3d_channel_moments' result without its pass over
 *            the stored tensor: the batch statistic of the BatchNorm inside the SPADE that reads this layer
 *            (/root/reference/lib/components/map3d_layers.py:162, 176-190: first_norm in training mode).
 * 1x1 convolutions (the dense layers) run two workgroups per CU on a 4-stage ring; 3x3 ones keep one workgroup and 7 stages. */
int h3d_conv_x3_ex(int mode, const void* x, const void* stream, const float* bias, const void* add, void* out, float* moments,
                   float* workspace, int slices, int B, int H, int W, int Cin, int Cout, int k, int ldx, int ldo, int lda, int NT,
                   h3d_stream_t stream_handle);
/*   slices   K-slices (<= 1: none): the tap x channel-chunk loop is cut into that many pieces, each a workgroup of its own writing
 *            fp32 sums into workspace [slices, B*H*W, Cout]; a second launch adds them in slice order with bias and addend and rounds
 *            once to the output's type.  h3d_conv_x3_slices (HOST helper) says how many are worth it: > 1 only when the grid at the
 *            narrowest blocking still leaves CUs idle (the discriminator's 16 x 8 and 8 x 4 layers: 288 k-steps in sequence on 32 or 8
 *            workgroups otherwise).  moments and slices exclude each other. */
int h3d_conv_x3_slices(int Cin, int Cout, int k, int64_t P, int NT);
int h3d_conv_x3_moment_rows(void);      /* output pixels behind one row of the moments buffer (HOST helper) */
int h3d_conv_x3_nt_for(int Cin, int Cout, int64_t P);      /* HOST helper; 0 = unsupported channel counts */
/* planes: 0 = bf16 hi + lo (h3d_conv_x3_pack), 1 = f16 hi + lo (_pack_f16), 2 = one f16 plane (_pack_f16x1) */
int h3d_conv_x3_pack_nt(const float* w, void* stream, int Cout, int Cin, int k, int transposed, int planes, int NT,
                        h3d_stream_t stream_handle);
int h3d_conv_wgrad_x3_f16(const void* dY, const void* X, float* partial, int B, int H, int W, int Co, int Ci, int k, int ldy,
                          int ldx, int slices, h3d_stream_t stream);
/* ... with the convolution's bias gradient riding along (round 4): colsum [slices][Co] fp32 = column sums of dY over each slice's
 * pixels (the caller sums the slices, as for `partial`); half = 1: f16 operands.  Replaces the separate dY.sum((0, 2, 3)) pass of
 * nn.Conv2d's backward (reference: the biased convolutions of lib/discriminators/unet_discriminators.py:8-72). */
int h3d_conv_wgrad_x3_bias(const void* dY, const void* X, float* partial, float* colsum, int B, int H, int W, int Co, int Ci, int k,
                           int ldy, int ldx, int slices, int half, h3d_stream_t stream);
int h3d_wgrad_x3_bias_f16(const void* dY, const void* X, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                          int ldx, int slices, h3d_stream_t stream);
/* K-slices to ask for (HOST helper), and whether a 3x3 problem runs on the kernel that fuses the three taps of a filter row
 * (three passes over dY and X instead of nine; image rows must be multiples of 16 pixels). */
int h3d_conv_wgrad_x3_slices(int B, int H, int W, int Co, int Ci, int k);
int h3d_conv_wgrad_x3_fused(int B, int H, int W, int Co, int Ci);

/* ------------------------------------------------------------------------
 * A5 / A5+A6 in the "x2" arithmetic (csrc/x3_common.hpp): the hidden-layer contractions evaluate W.x as one f16 product
 * hi*hi plus ONE block-scaled fp6 (e2m3) matrix instruction for the two cross terms hi*lo + lo*hi, which are 2^-11 of the
 * main term -- half the matrix-pipe time and 55 % of the energy of three f16 products; inputs (K = 3 / 31 / view direction)
 * and the four 1-row heads keep three f16 products.  2e-5 at the operator boundary at width 256 (error model:
 * tests/test_x2_error_model_cpu.py).  Same arguments, results and limits as the _x3 entry points; own blob format:
 * h3d_field_x2_layout reports the same 20 values as h3d_field_x3_layout; the stages of the seven accumulator-fed
 * matrices are [tile][1 KiB f16 hi fragment][1 KiB half of the K-tile's fp6 records] (record = 32 B per lane: 32 6-bit
 * codes, slots 0-15 q6(hi * alpha), slots 16-31 q6(lo * 2^12 * alpha) of the lane's 16 features in accumulator order, then
 * the e8m0 byte of 1 / alpha four times, then 0; dwords 0-3 travel with the even k-step, 4-7 with the odd one) and
 * head_w has a third plane (hi * 2^-12).
 */
int64_t h3d_field_pack_x2_size(int Hd, int F);
int h3d_field_pack_x2(const h3d_field_params* p, int Hd, int F, void* blob /* HOST */);
/* h3d_field_pack_x2 from DEVICE parameters into DEVICE memory (see h3d_field_pack_x3_device); bit-identical blob. */
int h3d_field_pack_x2_device(const h3d_field_params* p, int Hd, int F, void* blob /* DEVICE */, h3d_stream_t stream);
int h3d_field_x2_layout(int Hd, int F, int64_t* out, int n_out);
int h3d_neural_field_x2(const void* packed, const float* points, const float* geo, const float* dirs,
                        const float* freq, const float* phase, float* out,
                        int B, int64_t N, int Hd, int F, int geo_stride, float input_scaler,
                        h3d_stream_t stream);
int h3d_render_fused_x2(const void* packed, const float* points, const float* geo, const float* dirs,
                        const float* freq, const float* phase, const float* z_vals, const float* noise,
                        float* feats, float* depth, float* weights,
                        int B, int R, int S, int Hd, int F, int geo_stride, float input_scaler,
                        int clamp_mode, int last_back, int white_back, h3d_stream_t stream);

/* h3d_render_fused_x2 / _x3 with A4 inside (round 4): instead of the feature tensor `geo` the kernel takes every sample's
 * nearest-vertex index (h3d_nearest_vertex) and the pose tables -- joints [B,24,3], vertices / tpose_vertices [B,V,3],
 * vertex_ik [B,V,16] (the blended inverse bone transforms, smpl.py:217-218; 16-byte aligned) -- and builds the sample's geometry
 * features (lib/components/smpl.py:210-249: canonical coordinates, 24 joint distances, nearest T-pose vertex, distance) in the
 * prologue of its 32-sample step, as the B fragments of the K = 31 input GEMM.  Everything else as h3d_render_fused_x2 / _x3. */
int h3d_render_fused_x2_geo(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                            const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                            int legacy_mode, const float* dirs, const float* freq, const float* phase,
                            const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                            int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                            int white_back, h3d_stream_t stream);
int h3d_render_fused_x3_geo(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                            const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                            int legacy_mode, const float* dirs, const float* freq, const float* phase,
                            const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                            int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                            int white_back, h3d_stream_t stream);

/* Refinement of ill-conditioned last samples (round 6).  lib/generators/volume_rendering.py:21 gives a ray's last sample
 * delta = 1e9: its alpha is 0 or 1 by the SIGN of its density (and with white_back the background term flips by the whole
 * remaining transmittance, :48-49), so a ray whose last density lies within the ARITHMETIC's error of zero comes out wrong by
 * O(1).  The x2 arithmetic's density error (~1e-4 of the densities' scale) makes that ~100 times likelier than fp32-class
 * arithmetic does.  h3d_render_fused_x2_geo_ref is h3d_render_fused_x2_geo that also lists, per batch item, the wave units (a
 * unit = one ray when S > 32, else the 32 / S rays of 32 consecutive samples; unit u covers samples [u, u + 1) * max(S, 32))
 * holding a ray with
 *     |sigma_last| <= ref_eps * max(max_s |sigma_s|, ref_scale)          (sigma: the density as integrated, noise added)
 * in ref_list [B, ref_cap] int32 (any order) and counts them in ref_count [B] int32 (zeroed by the call on `stream`; it keeps
 * counting beyond ref_cap: units past the capacity stay on the x2 arithmetic).  h3d_render_fused_x3_geo_units is
 * h3d_render_fused_x3_geo (`packed` in the x3 format) restricted to exactly the listed units of every item: it overwrites their
 * feats / depth / weights, so that those rays take the sign of an fp32-class density.  Launched back to back on one stream:
 * no host synchronisation.  1 <= ref_cap <= 65536. */
int h3d_render_fused_x2_geo_ref(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                int white_back, float ref_eps, float ref_scale, int32_t* ref_list, int32_t* ref_count,
                                int ref_cap, h3d_stream_t stream);
int h3d_render_fused_x3_geo_units(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                  const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                  int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                  const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                  int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                  int white_back, const int32_t* ref_list, const int32_t* ref_count, int ref_cap,
                                  h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A5 / A5+A6, split-operand arithmetic as the _x3 entry points, for hidden widths up to 448 ("x3t": the activations of
 * a 64-sample tile live in LDS as ready-made MFMA fragments and the output features are split over the four waves, so
 * the width is bounded by LDS instead of registers; csrc/x3t_common.hpp).  Covers MAP3DBN (384) and MAP3DBN512L (420).
 * Same arguments and results as h3d_neural_field / h3d_render_fused; the fused variant takes S in {8,16,32,64} or a
 * multiple of 64.  Weights are packed by h3d_field_pack_x3t (its own blob format, see h3d_field_x3t_layout).
 */
int64_t h3d_field_pack_x3t_size(int Hd, int F);
int h3d_field_pack_x3t(const h3d_field_params* p, int Hd, int F, void* blob /* HOST */);
/* The same blob layout for the x2 tier (`products` = 4 of the _tier entry points below): the matrices fed by accumulators and
 * the head tile carry f16 hi fragments + the K-tiles' fp6 records (dwords 0-3 in the even k-step's lo plane, 4-7 in the odd
 * one's; record layout as h3d_field_pack_x2) instead of hi + lo fragments. */
int h3d_field_pack_x2t(const h3d_field_params* p, int Hd, int F, void* blob /* HOST */);
/* HOST helper: out[0..17] = tiles NT (even, >= 4), k-steps KS = 2*NT, padded width 32*NT, then BYTE offsets of the eight
 * weight matrices (coord [1 k-step], geo [2], film0 [2*KS: coordinate half, geometry half], film1..3 [KS], colour
 * [KS + 1: the last k-step carries the view direction], feature head [KS]), of inv_scale float[8], bias float[7][HdP],
 * b_feat float[HdP], head_w (f16 [KS][hi|lo][64][8]: ONE tile whose rows 0..3 are the sigma, r, g, b heads, each with its own
 * power-of-two scale), head_inv float[4] (1 / scale), head_b float[4] and the total size.
 * A matrix is [tile][k-step][hi|lo][64 lanes][8 f16], element (lane, e) = s * W[n = 32*tile + (lane&31)][k], K in
 * accumulator-register order for matrices fed by accumulators, natural order 16*ks + 8*(lane>>5) + e for coord, geo and
 * the view-direction k-step. */
int h3d_field_x3t_layout(int Hd, int F, int64_t* out, int n_out);
int h3d_neural_field_x3t(const void* packed, const float* points, const float* geo, const float* dirs,
                         const float* freq, const float* phase, float* out,
                         int B, int64_t N, int Hd, int F, int geo_stride, float input_scaler,
                         h3d_stream_t stream);
int h3d_render_fused_x3t(const void* packed, const float* points, const float* geo, const float* dirs,
                         const float* freq, const float* phase, const float* z_vals, const float* noise,
                         float* feats, float* depth, float* weights,
                         int B, int R, int S, int Hd, int F, int geo_stride, float input_scaler,
                         int clamp_mode, int last_back, int white_back, h3d_stream_t stream);
/* Precision tiers of the same kernels (same blob): products = 3 is the entry point above (hi*hi + hi*lo + lo*hi, fp32
 * class); products = 1 evaluates the hidden GEMMs with ONE f16 product per operand pair -- plain f16 matrix-core
 * arithmetic, ~3x less matrix work, ~1e-2 relative on the render (the freq ~ 45 sines amplify the 2^-11 operand rounding):
 * outside the 1e-3 parity budget, provided as BASELINE config 5's "fp16 MFMA path".  The K=3 / K=31 input layers always
 * run with 3 products. */
int h3d_neural_field_x3t_tier(const void* packed, const float* points, const float* geo, const float* dirs,
                              const float* freq, const float* phase, float* out,
                              int B, int64_t N, int Hd, int F, int geo_stride, float input_scaler, int products,
                              h3d_stream_t stream);
int h3d_render_fused_x3t_tier(const void* packed, const float* points, const float* geo, const float* dirs,
                              const float* freq, const float* phase, const float* z_vals, const float* noise,
                              float* feats, float* depth, float* weights,
                              int B, int R, int S, int Hd, int F, int geo_stride, float input_scaler,
                              int clamp_mode, int last_back, int white_back, int products, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * A7+A8+A9  SPADE synthesis network, eval mode == SynthesisNetwork.forward
 *     (lib/generators/map3d_generator.py:58-97) over SPADEBlock (lib/components/map3d_layers.py:218-238),
 *     fed by SynthesisInput (:260-275) and the bilinear F.interpolate of map3d_generator.py:244-245.
 *
 * Weight blob: fp32, every matrix packed by h3d_pack_matrix (MFMA B-fragment order
 *   packed[nt][kb][lane][e] = W[k = 8*kb + 4*(lane>>5) + e][n = 32*nt + (lane&31)], zero padded), offsets in
 *   FLOATS given by the descriptor; HdP = C rounded up to 32; vectors are HdP long, zero padded.
 * Per SPADE:  pixel_style = 1  gamma/beta depend on the pixel: a = relu(bilinear(G[:, g_offset : g_offset+128])
 *                              + cst[b, cst_index]);  gamma+1 = a*Wg + vec[0:HdP]; beta = a*Wb + vec[HdP:2HdP];
 *                              y = lrelu((x*vec[2HdP:3HdP] + vec[3HdP:4HdP]) * (gamma+1) + beta)
 *             pixel_style = 0  y = lrelu(x * ab[b, ab_index, 0] + ab[b, ab_index, 1])
 *             then conv: y * Wconv + b_conv (+ block input when `skip` and it is the block's second SPADE).
 * to_rgb: rgb += x * Wrgb^T + brgb, Wrgb stored as [3][HdP] followed by the 3 biases (+1 pad) at w_rgb.
 * x0 = sin(w_in[0][n]*i + w_in[1][n]*j + b_in[n]) with (i, j) = linspace(-1,1) pixel coordinates.
 */
#define H3D_MAX_BLOCKS 16
typedef struct {
    int32_t pixel_style, g_offset, cst_index, ab_index;
    int64_t w_gamma, w_beta, vec, w_conv, b_conv;
} h3d_spade_desc;
typedef struct {
    h3d_spade_desc spade[2];
    int32_t skip, to_rgb;
    int64_t w_rgb;
} h3d_block_desc;
typedef struct {
    int32_t n_blocks, C;
    int64_t w_in, b_in;
    h3d_block_desc block[H3D_MAX_BLOCKS];
} h3d_synth_desc;

/* HOST helper: pack W given as [n_out, ld_in] row-major (reference layout), input features
 * [in_begin, in_begin+in_count) -> dst [NT][KB][64][4] floats. */
int h3d_pack_matrix(const float* w, int ld_in, int in_begin, int in_count, int n_out, int KB, int NT, float* dst);

/* G [B, Hr*Wr, g_channels] channels-last low-resolution shared-conv maps; cst [B, n_cst, 128];
 * ab [B, n_ab, 2, HdP]; rgb [B, 3, H, W] (written).  desc is a HOST pointer (copied into the launch). */
int h3d_synthesis(const void* blob, const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                  const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                  h3d_stream_t stream);

/* Same network on the bf16 matrix cores with split ("x3") operands (bf16 hi + lo, three partial products, fp32
 * accumulation; ~1e-5 on the image).  `stream`: all conv / gamma / beta matrices of one tile in consumption order
 * (per block, per SPADE: [gamma (8 k-steps), beta (8)] if pixel_style, then conv), each k-step stage
 * [tile][hi|lo][64 lanes][8 bf16] with element (lane, e) = W[n = 32*tile + (lane&31)][k(kstep, lane>>5, e)], K in
 * accumulator-register order k(ks, h, e) = 32*(ks/2) + (e&3) + 8*(2*(ks&1) + (e>>2)) + 4*h (a lane's accumulator
 * registers of the producing layer are then its B-fragment elements; every matrix of this engine is fed that way);
 * tiles = 4 (C <= 128) or 8 (C <= 256), HdP = 32*tiles.  `tables`: fp32, the descriptor's vec / b_conv / w_rgb /
 * w_in / b_in offsets index it (vectors HdP long); w_gamma / w_beta / w_conv of the descriptor are ignored.
 * `ab` for THIS entry point is [B, n_ab, HdP/2, 4] = 0.4 * (scale[n], scale[n+1], shift[n], shift[n+1]) -- one 16-byte read per
 * two channels -- not the [B, n_ab, 2, HdP] of h3d_synthesis; the factor 0.4 because the kernel evaluates
 * lrelu(t) = 0.6 t + 0.4 |t| as 1.5 u + |u| on u = 0.4 t (one instruction after the affine).
 * The conv biases are NOT read by this engine: the caller folds them into the consumers' tables (the activations it
 * carries are the true ones minus a per-channel carry c; SPADE shift sh' = sh + sc*c in `ab` / `vec`, ToRGB bias
 * br' = br + Wr*c -- see SynthesisPlan.build_x3 in lib/generators/synthesis_pack.py).
 * Returns H3D_EUNSUPPORTED (use h3d_synthesis) for C > 256, a per-pixel-style block after the first skip block, a
 * block without skip connection after the first one that has it, or (with per-pixel-style blocks) a geometry
 * h3d_synthesis_x3_geometry_ok rejects.
 *
 * Segmented execution: the weight stream of the whole network (6.3 MB at C=256) does not fit the 4 MB L2 of an XCD,
 * so a caller may run the blocks in several launches whose streams do (desc = the blocks of one segment).  Between
 * launches the per-pixel activations and ToRGB partial sums live in `state`
 * (B * ceil(H*W/128) * 4 wave tiles * (tiles*4 + 1) * 64 float4, private lane-linear layout): store_state=1 writes
 * it instead of the image, load_state=1 resumes from it instead of generating the coordinate input. */
/* 1 when the x3 engine's matrix-core bilinear resize covers this geometry (only needed with per-pixel-style blocks):
 * W a multiple of 32 and 32 consecutive output pixels touching at most 8 low-res columns (31*Wr < 6*W). */
int h3d_synthesis_x3_geometry_ok(int H, int W, int Hr, int Wr);
/* HOST helper: LDS bytes h3d_synthesis_x3 (x2 = 0) / h3d_synthesis_x2 (x2 = 1) needs for a network of width C with
 * `table_floats` floats of static tables, n_ab constant-style and n_cst per-pixel-style SPADEs (must be <= 160 KiB);
 * x2 = 3: an h3d_synthesis_x2 plan with ToRGB head tables (below), which keeps no zero ToRGB table in LDS (3 * 4C bytes less). */
int64_t h3d_synthesis_x3_lds_bytes(int table_floats, int n_ab, int n_cst, int C, int x2);

int h3d_synthesis_x3(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                     const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                     const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                     float* state, int load_state, int store_state, h3d_stream_t stream_handle);

/* h3d_synthesis_x3 in the "x2" arithmetic (csrc/x3_common.hpp, see the _x2 field entry points): every conv / gamma /
 * beta contraction is one f16 product hi*hi plus one block-scaled fp6 (e2m3) matrix instruction for the two cross terms;
 * activations are not bounded here, so every pixel's K-tile record carries its own power-of-two scale (largest of its 16
 * values).  f16 planes: inputs of these convolutions must stay below 2^15 in magnitude (hi = f16(x) and f16(lo * 2^12) both
 * finite) -- the reference trains them under fp16 autocast (lib/trainers/base_trainer.py:50-51); h3d_synthesis_x2_guarded
 * below detects a violation.  Same arguments, limits and return codes as
 * h3d_synthesis_x3; the stream holds, per stage, [tile][1 KiB f16 hi fragment][1 KiB half of the K-tile's fp6 records]
 * (record layout as for h3d_field_pack_x2; SynthesisPlan.pack_stream_x2), and the kernel needs
 * h3d_synthesis_x2_extra_lds(C) more bytes of LDS than h3d_synthesis_x3 (one more ring buffer).
 * ToRGB heads (round 5, single-launch x2 plans only): a skip block whose spade[1].b_conv is >= 0 (this engine folds conv biases on
 * the host, so the field is free) carries at that FLOAT offset of `tables` a 4 KB table [k-step][f16 hi fragment | half of the
 * fp6 record][4 rows x 2 lane halves][16 B] = the x2 operands of M_j = (sum of the ToRGB weights of this and every later block)
 * x W_conv1 of the block, 3 rows used: the block's second convolution multiplies it with its own input fragments as a ninth
 * output tile, accumulated over all skip blocks into the image.  Such blocks have to_rgb = 0, and the block in front of the
 * first skip block carries the summed ToRGB weights and biases (exact algebra on lib/generators/map3d_generator.py:82-86).
 * Three-product middle blocks (round 6, single-launch x2 plans only): the constant-style blocks between the per-pixel-style blocks
 * and the first skip block (block 3 of the shipped configurations: the base of the residual stream) may carry g_offset = 1 in both
 * of their SPADEs (the field is unused for pixel_style = 0): their convolutions' stages are then in the h3d_synthesis_x3 format
 * (bf16 hi | lo) and run on three bf16 products inside this kernel -- all such blocks or none.  SynthesisPlan.build_x3(x2=True)
 * does that by default (H3D_SYNTH_MID_X3=0: all-x2 stream). */
int h3d_synthesis_x2(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                     const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                     const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                     float* state, int load_state, int store_state, h3d_stream_t stream_handle);
int h3d_synthesis_x2_extra_lds(int C);

/* Range-guarded pair (round 4).  The x2 engine's operand planes are f16: hi = f16(x) and f16(lo * 2^12) with |lo| <= ulp(hi)/2
 * are finite exactly for |x| < 2^15.  h3d_synthesis_x2_guarded is h3d_synthesis_x2 (single launch: no state) that also ORs 1
 * into overflow[b] (int32[B] in device memory, zeroed by the caller on the same stream; round 6: ONE FLAG PER SAMPLE of the
 * batch, rounds 4-5 had one per launch) when any activation of sample b it fed to the matrix cores was >= 2^15 in magnitude or
 * not finite -- that sample's image is then not to be used.  h3d_synthesis_x3_if is h3d_synthesis_x3 (bf16 planes: fp32
 * exponent range; `stream` in the x3 format) that leaves the image of every sample with run_if[b] == 0 untouched (its
 * workgroups return at once) and recomputes the others, each on a grid that fills the chip by itself.  Launched back to back on
 * one stream with the same flags the pair is "x2, the samples out of range redone on x3" -- a sample's pixels never depend on
 * its batch mates -- without a host synchronisation (SynthesisPlan.run does exactly that; replaces nothing in the reference -- its fp32
 * convolutions, lib/components/map3d_layers.py:193-238, have no range limit to guard). */
int h3d_synthesis_x2_guarded(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                             const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                             const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                             int* overflow, h3d_stream_t stream_handle);
int h3d_synthesis_x3_if(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                        const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                        const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                        const int* run_if, h3d_stream_t stream_handle);

/* Sampled error monitor of the x2 engine (round 5; replaces nothing in the reference -- its fp32 convolutions,
 * lib/components/map3d_layers.py:176-238, have no reduced-precision tier to watch).  The x2 arithmetic sits inside the 1e-3
 * parity budget with little room (measured over 192 images: up to 9.6e-4 of the channel maximum, median 2.2e-4), so every forward checks a
 * sample of its own output against the fp32-class engine:
 *   h3d_synthesis_x3_tiles  h3d_synthesis_x3 (single launch, `stream` in the x3 format) restricted to the 128-pixel tiles
 *                           tile_first, tile_first + tile_step, .. of every sample; writes those pixels of `rgb` (a scratch
 *                           image of the full [B,3,H,W] shape) and nothing else;
 *   h3d_synthesis_check     per sample: err = max over channels of (max|rgb - rgb_ref| over exactly those tiles) / (max|rgb|
 *                           over the WHOLE image: the scale the 1e-3 budget is relative to), written to err_out[b] when
 *                           err_out != NULL; ORs 1 into flag[b] (int32[B], device memory) when err > tol or anything it
 *                           read of sample b is not finite.
 * Launched behind h3d_synthesis_x2_guarded and in front of h3d_synthesis_x3_if with the same flags this is "x2, a sample redone
 * on x3 when one of ITS sampled pixels leaves the tolerance", without a host synchronisation (SynthesisPlan.run; the default
 * tolerance there is the 1e-3 budget divided by the measured ratio of a full image's maximum to its sample's).  tile_step >= 1
 * (a step beyond the tile count samples tile_first alone). */
int h3d_synthesis_x3_tiles(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                           const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                           const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                           int tile_first, int tile_step, h3d_stream_t stream_handle);
int h3d_synthesis_check(const float* rgb, const float* rgb_ref, int B, int H, int W, int tile_first, int tile_step,
                        float tol, int* flag, float* err_out, float* work /* [6 B] scratch, zeroed by the call */,
                        h3d_stream_t stream_handle);

/* Same network, split-bf16 arithmetic as h3d_synthesis_x3, for widths up to 448 ("x3t": the activations of a 64-pixel
 * tile live in LDS as ready-made MFMA fragments, the channels are split over the four waves; csrc/x3t_common.hpp).
 * tiles = h3d_synthesis_x3t_tiles(C) (even, >= 4; -1 when C > 448), HdP = 32*tiles.
 * `wblob`: every matrix as bf16 hi/lo A fragments [tile][k-step][hi|lo][64 lanes][8], element (lane, e) =
 * W[n = 32*tile + (lane&31)][k]; conv matrices (2*tiles k-steps) with K in accumulator-register order
 * k = 32*(ks/2) + (e&3) + 8*(2*(ks&1) + (e>>2)) + 4*(lane>>5), gamma / beta (8 k-steps) in natural order
 * 16*ks + 8*(lane>>5) + e.  The descriptor's w_gamma / w_beta / w_conv are BYTE offsets into wblob; vec / b_conv / w_rgb
 * / w_in / b_in are FLOAT offsets into `tables` (vectors HdP long, zero padded; conv biases are read, nothing is folded).
 * G, cst as h3d_synthesis; ab [B, n_ab, 2, HdP].  Any resize geometry.  Returns H3D_EUNSUPPORTED (use h3d_synthesis)
 * for C > 448, a per-pixel-style SPADE inside a skip block, or a block without skip connection after the first one
 * that has it. */
int h3d_synthesis_x3t_tiles(int C);
int h3d_synthesis_x3t(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G, int g_channels,
                      int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H,
                      int W, h3d_stream_t stream);
/* Precision tiers: (dtype 0 = bf16, products 3) is the entry point above.  (dtype 1 = f16, products 2): wblob holds f16
 * hi/lo fragments, activations are rounded to ONE f16 value (products hi*hi + lo*hi); (1, 1): plain f16 products.  Both
 * are outside the 1e-3 parity budget (BASELINE config 5's "fp16 MFMA path"); f16 because one bf16 value per activation
 * (8 significant bits) is too coarse. */
int h3d_synthesis_x3t_tier(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G, int g_channels,
                           int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H,
                           int W, int dtype, int products, h3d_stream_t stream);
/* Range-guarded pair of this engine (round 4; as h3d_synthesis_x2_guarded / h3d_synthesis_x3_if): with products == 4 (the x2
 * tier) the launch ORs 1 into flag[b] (int32[B], device memory, zeroed by the caller: one flag per sample) when an activation
 * of sample b it converted to the f16 planes was >= 2^15 in magnitude or non-finite; with any other tier -- use (dtype 0,
 * products 3): bf16 planes, fp32 exponent range, `wblob` in that tier's format -- it recomputes exactly the samples with
 * flag[b] != 0 and leaves the others' images untouched. */
int h3d_synthesis_x3t_tier_guarded(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G,
                                   int g_channels, int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab,
                                   float* rgb, int B, int H, int W, int dtype, int products, int* flag, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * P3a per-pixel modulated 1x1 convolution == SpatialStyleModLayer.forward (lib/components/map3d_layers.py:60-80)
 *     m = style * Wa^T + ba + 1 ;  out = (x*m) W * rsqrt((m*m) (W*W) + eps) + bias        (demodulate != 0)
 * x [rows, Cin], style [rows, S], out [rows, Cout] row-major fp32; w_aff_packed = pack(Wa [Cin,S]),
 * w_packed = pack(W^T [Cout,Cin]), w2_packed = pack((W*W)^T) (h3d_pack_matrix, KB = K/8 rounded up to a multiple
 * of 4, NT = N/32 rounded up); b_aff_plus1 [CinP], bias [CoutP] zero padded.  Widths <= 256.
 */
int h3d_modconv1x1(const float* x, const float* style, const float* w_aff_packed, const float* b_aff_plus1,
                   const float* w_packed, const float* w2_packed, const float* bias, float* out, int64_t rows,
                   int Cin, int Cout, int S, int demodulate, float eps, h3d_stream_t stream);

/* P3b StyleGAN2 modulated k x k convolution == StyleModLayer.forward_group_conv (lib/components/cips_layers.py:235-278)
 *     out[b,o] = dmod[b,o] * sum_{i,ky,kx} W[o,i,ky,kx] * (smod[b,i] * x[b,i,.+ky-k/2,.+kx-k/2]) + bias[o]
 * x [B,Cin,H,W], out [B,Cout,H,W] NCHW fp32; smod [B,CinP] = affine(style)+1, dmod [B,CoutP] (ones without
 * demodulation), w_packed = k*k consecutive pack(W[:,:,ky,kx] [Cout,Cin]) blocks, bias [CoutP].  Odd k <= 7.
 */
int h3d_modconv2d(const float* x, const float* smod, const float* dmod, const float* w_packed, const float* bias,
                  float* out, int B, int Cin, int Cout, int H, int W, int k, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Training side: streaming kernels the differentiable generator path is assembled from; the GEMMs between
 * them are library GEMMs (hipBLASLt through torch).  Caller owns every buffer.
 *
 * h3d_film_sin       y = sin(freq[b,c] * x[b,n,c] + phase[b,c])   == the activation of SineLayer / FiLMLayer
 *                    (lib/components/pigan_layers.py:63-71, 74-87) applied to the Linear output x.
 *                    x, y [B,N,C] dtype 0 = f32 / 1 = f16 (fp32 arithmetic); freq, phase [B,C] fp32, or both NULL:
 *                    y = sin(w0 * x).
 * h3d_film_sin_bwd   dx = dy * cos(.) * freq;  partial [B, nblk, 2, C] fp32 with nblk = ceil(N / h3d_film_sin_rows()):
 *                    per-workgroup sums of dy*cos(.)*x (-> d freq) and dy*cos(.) (-> d phase); the caller sums over nblk
 *                    (deterministic).  partial may be NULL when freq / phase need no gradient.
 * h3d_ray_integrate_bwd   gradient of h3d_ray_integrate w.r.t. `field` given the gradients of its three outputs
 *                    (g_feats [n_rays,C]; g_depth [n_rays] or NULL; g_weights [n_rays,S] or NULL) -> d_field [n_rays,S,C+1].
 *                    Same flags as the forward call.  S <= 2048.
 */
int h3d_film_sin_rows(void);
int h3d_film_sin(const void* x, const float* freq, const float* phase, void* y, int B, int64_t N, int C, int dtype,
                 float w0, h3d_stream_t stream);
int h3d_film_sin_bwd(const void* x, const float* freq, const float* phase, const void* dy, void* dx, float* partial,
                     int B, int64_t N, int C, int dtype, float w0, h3d_stream_t stream);
int h3d_ray_integrate_bwd(const float* field, const float* z_vals, const float* noise, const float* g_feats,
                          const float* g_depth, const float* g_weights, float* d_field, int64_t n_rays, int S, int C,
                          int clamp_mode, int last_back, int white_back, h3d_stream_t stream);

/* Zero-padded channels, channels-last:  out[b][p][c] = in[b*sb + c*sc + p*sp] for c < Cin, 0 for Cin <= c < Cout  (out [B, HW, Cout]
 * contiguous, 16-byte aligned, Cout % 4 == 0; element strides sb, sc, sp: any input layout; dtype 0 = f32, 1 = f16).  The padding in
 * front of a native convolution with a channel count that is not a multiple of 64 (the discriminator's RGB stem,
 * /root/reference/lib/discriminators/unet_discriminators.py:117) and the gradient of a narrowed output (its heads, :145-146):
 * one pass instead of torch's zero fill + layout copy + concatenation. */
int h3d_pad_channels_cl(const void* in, void* out, int B, int Cin, int Cout, int64_t HW, int64_t sb, int64_t sc, int64_t sp, int dtype,
                        h3d_stream_t stream);

/* Weight gradient of a dense layer of the training path:  dW[Co,Ci] = dY[M,Co]^T X[M,Ci]  (the `grad_weight` torch's
 * AddmmBackward computes for F.linear; lib/generators/differentiable.py routes every layer with enough rows here).
 * Split-K over the M rows on the bf16 matrix cores with split operands (fp32-class: 16 mantissa bits per operand, fp32
 * accumulation).  dY, X fp32 row-major with leading dimensions ldy, ldx (multiples of 4, 16-byte aligned bases);
 * partial [slices, Co, Ci] fp32 with slices = h3d_wgrad_x3_slices(M, Co, Ci) (or any value >= 1): the caller sums over slices.
 */
int h3d_wgrad_x3_slices(int64_t M, int Co, int Ci);
int h3d_wgrad_x3(const float* dY, const float* X, float* partial, int64_t M, int Co, int Ci, int ldy, int ldx, int slices,
                 h3d_stream_t stream);
/* h3d_wgrad_x3 that also writes colsum[slice][Co] = column sums of dY over each slice's rows (the bias gradient; the caller
 * sums the slices).  colsum may be NULL. */
int h3d_wgrad_x3_bias(const float* dY, const float* X, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                      int ldx, int slices, h3d_stream_t stream_handle);
/* The slices' sum of h3d_wgrad_x3[_bias] (taps = 1) and h3d_conv_wgrad_x3[_bias] (taps = k * k) in one launch, written in the
 * parameter's layout:  dw[Co][Ci][taps] = sum_s partial[tap][s][Co][Ci];  db[Co] = sum_s colsum[s][Co] (colsum and db both NULL:
 * no bias gradient).  Replaces torch's `partial.sum(1)...permute(2, 3, 0, 1).contiguous()` + `colsum.sum(0)` -- the grad_weight /
 * grad_bias a library convolution_backward returns ready-made (reference: the autograd of
 * /root/reference/lib/discriminators/unet_discriminators.py:32-47).  Deterministic (fixed summation order).  Co, Ci multiples
 * of 4, taps <= 9, 16-byte aligned buffers. */
int h3d_wgrad_reduce(const float* partial, const float* colsum, float* dw, float* db, int taps, int slices, int Co, int Ci,
                     h3d_stream_t stream_handle);

/* Weight gradient with one narrow side: out[j][c] = sum_r narrow[r][j] * wide[r][c], j < nn <= 4 (ToRGB 3 x C, density /
 * colour heads, the coordinate layer transposed).  wide [M, C] fp32 with leading dimension ldw, narrow [M, nn] fp32 contiguous;
 * partial [ceil(M / h3d_wgrad_narrow_rows()), nn, C] fp32, the caller sums over the first dimension.
 */
int h3d_wgrad_narrow_rows(void);
int h3d_wgrad_narrow(const float* wide, const float* narrow, float* partial, int64_t M, int C, int ldw, int nn,
                     h3d_stream_t stream);
/* AMP tier: the same with both operands in f16 (fp32 products, sums and result); replaces the f16 library GEMM torch autocast
 * would run for these layers' weight gradients (reference: nn.Linear / 1x1 Conv2d backward inside torch.cuda.amp.autocast,
 * lib/trainers/base_trainer.py:50-51). */
int h3d_wgrad_narrow_f16(const void* wide, const void* narrow, float* partial, int64_t M, int C, int ldw, int nn,
                         h3d_stream_t stream);
/* The general form of the two above: operands fp32 (half = 0) or f16 (half = 1); ones = 1 appends the column sums of `wide` as
 * output row nn (partial is then [nblk, nn + 1, C], nn <= 3) -- the bias gradient when `wide` is dY (coordinate layers); colsum
 * (or NULL) receives [nblk, 4] per-block column sums of `narrow` -- the bias gradient when `narrow` is dY (ToRGB, heads).  Both
 * ride along the one pass over the operands (torch's column sum of a [0.5 M, 3] tensor alone takes 0.24 ms).
 * Reference: the bias gradients of nn.Linear / 1x1 Conv2d backward (lib/components/cips_layers.py:199-233 ToRGB, the heads of
 * lib/implicit_funcitions/modulated.py). */
int h3d_wgrad_narrow_sums(const void* wide, const void* narrow, float* partial, float* colsum, int64_t M, int C, int ldw, int nn,
                          int ones, int half, h3d_stream_t stream);

/* Training-side SPADE (backward of A9): BatchNorm + SPADE modulation + LeakyReLU of one SPADEBlock half
 *     y = lrelu_slope( ((x - mean) * rstd * g + b) * (1 + gamma) + beta )
 * == SPADE2d.forward + the block's activation (lib/components/map3d_layers.py:176-190, 228-233) over channels-last fp32
 * activations x [B,P,C].  gamma / beta: [B,P,C] when per_pixel, else [B,C].  Per-channel vectors [C] fp32.
 * "partial" buffers are [B, nblk, 2, C] fp32 with nblk = ceil(P / h3d_spade_rows()): per-workgroup sums the caller adds up
 * (deterministic two-stage reductions; with a process group the caller all-reduces the [2,C] totals -- SyncBatchNorm).
 *   h3d_channel_moments    partial[..][0] = sum x, [1] = sum x^2                      (batch statistics)
 *   h3d_spade_fwd          scale = rstd * g, shift = b - mean * scale
 *   h3d_spade_bwd_reduce   partial[..][0] = sum dh, [1] = sum dh * n;  n = (x - mean) rstd, dh = dL/d(n g + b)   (= d b, d g)
 *   h3d_spade_bwd_apply    dx = rstd g (dh - c1 - n c2)  (c1, c2 = the reduced sums / row count for batch statistics, zeros for
 *                          running statistics); per_pixel: dgamma, dbeta [B,P,C]; else partial[..][0] = sum_p dgamma,
 *                          [1] = sum_p dbeta per sample
 */
int h3d_spade_rows(void);
int h3d_channel_moments(const float* x, float* partial, int B, int64_t P, int C, h3d_stream_t stream);
int h3d_spade_fwd(const float* x, const float* scale, const float* shift, const float* gamma, const float* beta, float* y,
                  int B, int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream);
int h3d_spade_bwd_reduce(const float* x, const float* mean, const float* rstd, const float* g, const float* b,
                         const float* gamma, const float* beta, const float* dy, float* partial, int B, int64_t P, int C,
                         int per_pixel, float slope, h3d_stream_t stream);
int h3d_spade_bwd_apply(const float* x, const float* mean, const float* rstd, const float* g, const float* b,
                        const float* gamma, const float* beta, const float* dy, const float* c1, const float* c2, float* dx,
                        float* dgamma, float* dbeta, float* partial, int B, int64_t P, int C, int per_pixel, float slope,
                        h3d_stream_t stream);

/* AMP tier (round 4): the same four passes on f16 activations -- x, y, dy, dx and the PER-PIXEL gamma / beta / dgamma / dbeta are
 * _Float16; per-channel vectors, per-sample gamma / beta and every partial sum stay fp32; arithmetic in fp32 registers, one
 * rounding at each store (reference: the SPADE chain under torch.cuda.amp.autocast, lib/trainers/base_trainer.py:50-51). */
int h3d_channel_moments_f16(const void* x, float* partial, int B, int64_t P, int C, h3d_stream_t stream);
int h3d_spade_fwd_f16(const void* x, const float* scale, const float* shift, const void* gamma, const void* beta, void* y, int B,
                      int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream);
int h3d_spade_bwd_reduce_f16(const void* x, const float* mean, const float* rstd, const float* g, const float* b, const void* gamma,
                             const void* beta, const void* dy, float* partial, int B, int64_t P, int C, int per_pixel, float slope,
                             h3d_stream_t stream);
int h3d_spade_bwd_apply_f16(const void* x, const float* mean, const float* rstd, const float* g, const float* b, const void* gamma,
                            const void* beta, const void* dy, const float* c1, const float* c2, void* dx, void* dgamma, void* dbeta,
                            float* partial, int B, int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream);
/* h3d_spade_bwd_apply[_f16] (dtype 0 / 1) whose dx also receives up to two more gradients of x (add1, add2: x's type and shape, or
 * NULL):  dx = (the SPADE term) + add1 + add2 -- x of a skip block feeds the SPADE, the residual connection and the previous
 * block's ToRGB head (/root/reference/lib/components/map3d_layers.py:228-236, 268-272); autograd would add the three gradients in
 * two passes of its own. */
int h3d_spade_bwd_apply_acc(int dtype, const void* x, const float* mean, const float* rstd, const float* g, const float* b,
                            const void* gamma, const void* beta, const void* dy, const float* c1, const float* c2, const void* add1,
                            const void* add2, void* dx, void* dgamma, void* dbeta, float* partial, int B, int64_t P, int C,
                            int per_pixel, float slope, h3d_stream_t stream);
/* The bookkeeping between the SPADE passes (round 6): ~20 tensor operations on [C] vectors per call become three launches.
 *   h3d_rows_sum_f64   out[c] = sum_r partial[r][c] accumulated in fp64 (partial [n_rows, n_cols] fp32: the per-workgroup sums of
 *                      h3d_channel_moments / h3d_spade_bwd_reduce viewed as rows of 2 C columns)
 *   h3d_bn_finish      sums [2, C] fp64 (sum x, sum x^2; all-reduced by the caller when a process group is on), count [1] fp64 (rows
 *                      behind them, DEVICE) -> mean, rstd = 1 / sqrt(var + eps) [C] fp32, and nn.BatchNorm's train-mode side
 *                      effects: running_mean / running_var (fp32, may be NULL) moved by `momentum` towards mean / the UNBIASED
 *                      variance, num_batches_tracked (int64, may be NULL) += 1  (/root/reference/lib/components/map3d_layers.py:162)
 *   h3d_bn_bwd_finish  local_sums [2, C] fp64 -> d_bias, d_weight [C] fp32; global_sums / count -> c1, c2 [C] fp32, the
 *                      batch-statistics terms h3d_spade_bwd_apply takes (count NULL: running statistics, c1 = c2 = 0)
 */
int h3d_rows_sum_f64(const float* partial, double* out, int64_t n_rows, int n_cols, h3d_stream_t stream);
int h3d_bn_finish(const double* sums, const double* count, float* mean, float* rstd, float* running_mean, float* running_var,
                  int64_t* num_batches_tracked, int C, float eps, float momentum, h3d_stream_t stream);
int h3d_bn_bwd_finish(const double* local_sums, const double* global_sums, const double* count, float* d_bias, float* d_weight, float* c1,
                      float* c2, int C, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * P1  bias_act forward == _plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)
 *     lib/components/ops/bias_act.cpp:32 with grad=0; kernel spec lib/components/ops/bias_act.cu:23-147
 * x, y: n dense elements; dtype: 0 = f32, 1 = f16, 2 = f64; b: size_b elements or NULL;
 * element i uses b[(i / step_b) % size_b]; act = 1..9 (linear, relu, lrelu, tanh, sigmoid, elu, selu,
 * softplus, swish); clamp < 0 disables clamping.
 */
int h3d_bias_act(const void* x, const void* b, void* y, int64_t n, int dtype,
                 int64_t size_b, int64_t step_b, int act, float alpha, float gain, float clamp,
                 h3d_stream_t stream);

/* P1  bias_act gradients == the same plugin entry with grad = 1 / 2 (bias_act.cpp:32, bias_act.cu:23-147; call sites
 *     lib/components/ops/bias_act.py:179, 198).
 * order 1:  out = g * gain * act'(.)            g = dL/dy              (the plugin's x argument with grad=1)
 * order 2:  out = g * dy2 * gain * act''(.)     g = d(dL)/d(dx), dy2 = the dy of the first-order call
 * The derivative is evaluated from what the forward kept: yref (= y) for every activation but swish, xref (= x, the
 * bias b is added here) for swish; linear needs neither unless clamp >= 0 (then yref).  Elements whose forward output was
 * clamped get 0.
 */
int h3d_bias_act_grad(const void* g, const void* b, const void* xref, const void* yref, const void* dy2, void* out,
                      int64_t n, int dtype, int64_t size_b, int64_t step_b, int order, int act, float alpha,
                      float gain, float clamp, h3d_stream_t stream);

/* ------------------------------------------------------------------------
 * P2  upfirdn2d forward == _plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
 *     lib/components/ops/upfirdn2d.cpp:16; kernel spec lib/components/ops/upfirdn2d.cu:29-200
 * x [B,C,H,W] with element strides xs[4]; f [fh,fw] fp32 dense; y [B,C,outH,outW] with strides ys[4];
 * outW = (W*upx + padx0 + padx1 - fw + downx) / downx (likewise outH).  dtype as bias_act.
 */
int h3d_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                  int B, int C, int H, int W, const int64_t xs[4],
                  int fh, int fw, int outH, int outW, const int64_t ys[4],
                  int upx, int upy, int downx, int downy, int padx0, int pady0, int flip, float gain,
                  h3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------------------
 * The discriminator's resampling / activation glue, fused (round 4).  Reference: lib/discriminators/unet_discriminators.py:8-72 --
 * nn.Sequential(LeakyReLU(0.2), nn.Upsample(scale_factor=2), conv) in the up blocks (:24-27), nn.AvgPool2d(2) (:44) and the
 * residual sum of forward() (:48-56).  Channels-last activations [B, H, W, C] of fp32 (half = 0) or f16 (half = 1, AMP tier).
 *   h3d_up2_mask:   out[b, 2y+i, 2x+j, c] = scale * m(mask[b,y,x,c]) * x[b,y,x,c] (+ addend[b, 2y+i, 2x+j, c]);  H, W = INPUT size
 *   h3d_pool2_mask: out[b, y, x, c] = scale * m(mask[b,y,x,c]) * sum_{i,j} (x (+ x2))[b, 2y+i, 2x+j, c];        Ho, Wo = OUTPUT size
 * m(t) = 1 for t > 0, else `slope` (LeakyReLU's derivative); mask / addend / x2 may be NULL (m = 1, nothing added).  For a fixed
 * mask the two are adjoint, each the other's backward: up(lrelu(x)) = up2_mask(x, mask = x), avgpool(s + d) = pool2_mask(s, d,
 * scale = 1/4). */
int h3d_up2_mask(const void* x, const void* mask, const void* addend, void* out, int B, int H, int W, int C, float slope, float scale,
                 int half, h3d_stream_t stream);
int h3d_pool2_mask(const void* x, const void* x2, const void* mask, void* out, int B, int Ho, int Wo, int C, float slope, float scale,
                   int half, h3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------------------
 * Spectral normalisation of a weight with one power iteration (round 4).  Reference: torch.nn.utils.spectral_norm as applied to
 * every convolution of the discriminator (lib/discriminators/unet_discriminators.py:17); arithmetic of SpectralNorm.compute_weight:
 *     v' = normalize(W^T u, eps),  u' = normalize(W v', eps),  sigma = u' . (W v'),  W_sn = W / sigma         W [R, K] row-major fp32
 * u', v' are written twice: to *_out (this call's copies, kept for the backward) and to *_buf (the module's buffers, updated in
 * place as torch does).  scratch: h3d_spectral_norm_scratch(R, K) floats.  Three launches; deterministic two-stage reductions.
 * Backward (u', v' constants, as in torch): dW = (G - sum(G * W_sn) u' v'^T) / sigma; two launches. */
int64_t h3d_spectral_norm_scratch(int R, int K);
int h3d_spectral_norm(const float* W, const float* u, float* u_out, float* u_buf, float* v_out, float* v_buf, float* sigma,
                      float* W_sn, float* scratch, int R, int K, float eps, h3d_stream_t stream);
int h3d_spectral_norm_bwd(const float* G, const float* W_sn, const float* u, const float* v, const float* sigma, float* dW,
                          float* scratch, int R, int K, h3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* H3D_H */
