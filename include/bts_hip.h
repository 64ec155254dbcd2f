/*
 * bts_hip.h -- C ABI of libbts_hip.so: the MI355X (gfx950) native BTS decoder hot path.
 *
 * Drop-in boundary for the hot path of minghanz/bts (reference files cited per entry,
 * paths relative to the reference checkout).  Conventions mirror the only native
 * interface the reference has, LocalPlanarGuidanceKernel<Device>::operator()
 * (tensorflow/custom_layer/local_planar_guidance.h:22-49): borrowed `const float*`
 * inputs, caller-allocated outputs, dims as `int`, work enqueued on a caller stream.
 *
 *  - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *  - nothing here allocates, frees or synchronises: calls are hipGraph-capturable;
 *  - return value: 0 = ok, >0 = a hipError_t from the launch, <0 = BTS_ERR_* (bad
 *    arguments); errors are raised by the wrapper, never thrown across the ABI
 *    (cf. OP_REQUIRES / errors::InvalidArgument in local_planar_guidance.cc:36-44,123);
 *  - re-entrant per (device, stream): every entry point works on the calling thread's current device and the given
 *    stream.  The library keeps no mutable state between calls except one idempotent, lock-free cache: "the
 *    dynamic-LDS limit of kernel K has been raised on device D" (a bit per device ordinal; the attribute is per
 *    device, so a process driving several GPUs -- nn.DataParallel, bts_test.py:91 -- gets it set on each).  Tuning
 *    knobs (BTS_CONV_*, BTS_WGRAD_* environment variables) are read once into immutable values.
 */
#ifndef BTS_HIP_H_
#define BTS_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define BTS_HIP_ABI_VERSION 14

#define BTS_ERR_INVALID      (-1)   /* bad argument (null pointer, non-positive dim, misalignment) */
#define BTS_ERR_UNSUPPORTED  (-2)   /* valid in the reference but not built here (e.g. odd upratio)  */

typedef void* bts_stream_t;

int         bts_hip_abi_version(void);
const char* bts_hip_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Local planar guidance, forward.
 * Replaces local_planar_guidance.forward (pytorch/bts.py:149-173); native statement:
 * LocalPlanarGuidanceKernel<GPUDevice> (tensorflow/custom_layer/local_planar_guidance.cu:33-72).
 *
 *   plane_eq : [B,4,h,w] NCHW planar (n1,n2,n3,n4), as the PyTorch module receives it
 *   depth    : [B,h*k,w*k]
 *   abs_min  : optional 1-float device scalar <- min |n1*u+n2*v+n3| (bts.py:167), may be NULL
 * Bit-exact with the reference on the same inputs: separate mul/add roundings (no FMA),
 * the +-1e-3 clamp of bts.py:168-171 and an IEEE division.  upratio in {1,2,4,8}.
 */
int bts_lpg_fwd_f32(const float* plane_eq, int B, int h, int w, int upratio,
                    float* depth, float* abs_min, bts_stream_t stream);

/* Local planar guidance, backward (training callers).
 * Replaces autograd through local_planar_guidance.forward (pytorch/bts.py:149-173); native statement:
 * LocalPlanarGuidanceGradKernel<GPUDevice> (tensorflow/custom_layer/local_planar_guidance.cu:95-150, same
 * one-thread-per-input-cell mapping and argument order: depth_grad, input -> grad_input).
 *   grad_depth [B,h*k,w*k] -> grad_plane_eq [B,4,h,w].  True derivative of the PyTorch module: includes the n4
 *   factor the TF op drops (cu:143-145) and zeroes d/dn1..n3 where the +-1e-3 clamp is active (bts.py:168-171).
 */
int bts_lpg_bwd_f32(const float* plane_eq, const float* grad_depth, int B, int h, int w, int upratio,
                    float* grad_plane_eq, bts_stream_t stream);

/* Fused LPG + glue, as bts.forward uses it (pytorch/bts.py:250-256, 264-270, 278-283):
 *   plane4   : [B*h*w,4] cell-interleaved (n1,n2,n3,n4) -- what bts_reduc_fwd_f32 emits
 *   normalize: !=0 -> n[0:3] /= max(||n||_2, 1e-12)  (F.normalize, bts.py:251); 0 if already done
 *   depth_scaled : [B,1,h*k,w*k] = lpg(plane)/max_depth                      (bts.py:255)
 *   ds_out   : optional nearest-downsampled copy [::ds_factor, ::ds_factor] (bts.py:256,270),
 *              element (b,y,x) written at ds_out[((b*Hd+y)*Wd+x)*ds_pix_stride]; NULL to skip
 *   abs_min  : optional, as above
 */
int bts_lpg_fused_fwd_f32(const float* plane4, int B, int h, int w, int upratio, int normalize,
                          float max_depth, float* depth_scaled,
                          float* ds_out, int ds_factor, long ds_pix_stride,
                          float* abs_min, bts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * reduction_1x1 forward: the whole 1x1-conv(+ELU) chain and its epilogue in one kernel.
 * Replaces reduction_1x1.forward (pytorch/bts.py:97-136).
 *
 *   x        : NHWC activations, pixel p channel c at x[p*x_pix_stride + c], c in [0,c_in)
 *   npix     : B*h*w
 *   c_in, c_first_out : the module's (num_in_filters, num_out_filters); the chain halves the
 *              width until it is < 8 exactly as bts.py:105-122 does.  Built chains:
 *              (128,128) (128,64) (64,32) (32,16).
 *   w_frag   : the chain's weights in MFMA fragment order (see bts_amd/ops.py:pack_reduc_weights)
 *   is_final : 0 -> out[p*4 + {0..3}] = (sin t cos f, sin t sin f, cos t, sigmoid(c2)*max_depth)
 *                   with t = sigmoid(c0)*pi/3, f = sigmoid(c1)*2pi (bts.py:127-134); if
 *                   `normalize` the normal is L2-normalised here (bts.py:251) so LPG need not;
 *              1 -> out[p] = sigmoid(c0) (bts.py:108-110)
 */
int bts_reduc_fwd_f32(const float* x, long x_pix_stride, long npix, int c_in, int c_first_out,
                      const float* w_frag, long w_frag_floats, float max_depth, int is_final,
                      int normalize, float* out, bts_stream_t stream);

/* reduction_1x1 (non-final) -> F.normalize -> local_planar_guidance -> /max_depth -> nearest downsample in ONE launch:
 * one scale of pytorch/bts.py:249-256 (8x8), 263-270 (4x4), 277-283 (2x2).  The plane equations go from the chain's
 * registers straight into the k x k depth block of their cell; they reach HBM only if `plane4` is given.
 *   x            : NHWC activations [B*h*w, >= c_in] (pixel stride x_pix_stride)
 *   (c_in, c_first_out, upratio) : (128,128,8), (128,64,4) or (64,32,2) -- the three scales of bts_size 512
 *   plane4       : optional [B*h*w,4] (n1,n2,n3 normalised, n4) for callers that want the plane equation; NULL to skip
 *   depth_scaled : [B,1,h*k,w*k] = lpg(plane)/max_depth                                  (bts.py:255, 269, 283)
 *   ds_out       : optional dense plane [B,2h,2w] = depth_scaled[..., ::k/2, ::k/2]        (bts.py:256, 270: scale_factor
 *                  0.25 at k=8, 0.5 at k=4); must be NULL for k=2
 *   abs_min      : optional 1-float device scalar <- min |den| over the batch, NaN if any denominator is NaN (bts.py:167)
 */
int bts_reduc_lpg_fwd_f32(const float* x, long x_pix_stride, int B, int h, int w, int c_in, int c_first_out,
                          const float* w_frag, long w_frag_floats, float max_depth, int upratio,
                          float* plane4, float* depth_scaled, float* ds_out, float* abs_min, bts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NHWC implicit-GEMM convolution on fp32-input MFMA (v_mfma_f32_32x32x2_f32), with fused
 * per-channel affine/activation prologue and epilogue.  Covers atrous_conv's two convolutions
 * (pytorch/bts.py:65-80: [first_bn]->ReLU->conv1x1->BN->ReLU->conv3x3 dilated) and the other
 * decoder convolutions (upconv bts.py:83-94, conv5..conv1, daspp_conv bts.py:183-221).
 *
 *   y[p, n] = E( sum_{tap,c} P(x[q(p,tap), c]) * w[tap][n][c] )
 *   P(v) = pre_relu( v*pre_scale[c] + pre_shift[c] )         (zero padding applied AFTER P)
 *   E(a) = post2( act( a*e1_scale[n] + e1_shift[n] ) ),  post2(v) = v*e2_scale[n] + e2_shift[n]
 *   q(p,tap): input pixel of output pixel p for the tap, with `stride`, `dil`ation, `pad`ding and an
 *   optional nearest `up`x upsample of the input folded into the index.
 */
typedef struct bts_conv_desc {
    const float* x;        /* input, NHWC: pixel q channel c at x[q*x_pix_stride + c]              */
    long  x_pix_stride;    /* floats between pixels (>= c_in_ld, multiple of 4)                   */
    int   c_in_ld;         /* loadable input channels (multiple of 4; pad channels must be 0)     */
    int   k_pad;           /* padded flattened K of the packed weights: multiple of 32,
                              >= ksize*ksize*c_in_ld; k = tap*c_in_ld + c                          */
    int   B, h_in, w_in;   /* input spatial size before the optional upsample                     */
    int   up;              /* 1 or 2: nearest upsample folded into the gather (bts.py:91)         */
    int   ksize;           /* odd, 1..7                                                            */
    int   dil;             /* dilation (>= 1)                                                      */
    int   stride;          /* output stride (1 or 2; must be 1 when up == 2)                       */
    int   pad;             /* zero padding on each side (reference convs: dil*(ksize/2))           */
    const float* w;        /* packed weights [c_out_pad][k_pad], c_out_pad % 32 == 0               */
    int   c_out;           /* real output channels                                                 */
    int   c_out_pad;
    const float* pre_scale;  /* [c_in_ld] or NULL (identity)                                       */
    const float* pre_shift;  /* [c_in_ld] or NULL                                                  */
    int   pre_relu;
    const float* e1_scale;   /* [c_out_pad] or NULL                                                */
    const float* e1_shift;
    int   act;               /* 0 none, 1 ReLU, 2 ELU, 3 sigmoid                                   */
    const float* e2_scale;   /* [c_out_pad] or NULL                                                */
    const float* e2_shift;
    float* y;                /* output                                                             */
    long  y_pix_stride;      /* NHWC: floats between pixels; ignored for NCHW                      */
    int   y_nchw;            /* 0: y[p*y_pix_stride + n]; 1: y[(b*c_out + n)*H*W + yx] (boundary)  */
    int   subpixel;          /* 1: "sub-pixel upconv" -- nearest-2x upsample + 3x3 conv (bts.py:90-92) computed as four
                                2x2 convolutions on the SOURCE pixels, one per output parity (py,px), 2.25x fewer
                                FLOPs, same result: requires ksize 2, up 1, stride 1, dil 1, NHWC output;
                                w = [4][c_out_pad][k_pad] (class = 2*py+px, taps pre-summed, see
                                bts_amd/ops.py:pack_upconv_subpixel); output is [B, 2*h_in, 2*w_in, c_out]   */
    float* y2;               /* optional second NHWC destination of the same result (a skip tensor
                                that must live in two concat buffers), or NULL                      */
    long  y2_pix_stride;
    float* splitk_ws;        /* optional scratch (caller-owned, exclusive to this stream while the call runs) that
                                lets under-filled launches split K over several workgroups: partial sums go here
                                and a second kernel reduces them in a fixed order (deterministic); NULL = never split */
    long  splitk_ws_floats;  /* its size in floats (8 * M * round_up(c_out,4) is always enough)                     */
    const float* res;        /* optional residual, NHWC [B,H,W,>=c_out]: y = act(e1(conv) + res) -- the bottleneck's
                                `out += identity; relu` of the ResNet/ResNeXt encoders (torchvision Bottleneck.forward,
                                caller side of pytorch/bts.py:327-338); NULL = none; NHWC output only                 */
    long  res_pix_stride;
    int   n_bundles;         /* 0/1: ordinary convolution.  > 1: grouped convolution run as n_bundles independent
                                channel bundles in ONE launch (ResNeXt's 32-group 3x3, groups packed block-diagonally
                                into bundles of >= 32 channels by the host): bundle j reads input channels
                                [j*c_in_ld, (j+1)*c_in_ld) and writes output channels [j*c_out, (j+1)*c_out);
                                w = [n_bundles][c_out_pad][k_pad]; pre_* hold n_bundles*c_in_ld and e1_/e2_*
                                n_bundles*c_out_pad entries; x_pix_stride >= n_bundles*c_in_ld,
                                y_pix_stride >= n_bundles*c_out; no sub-pixel, NCHW output or split-K               */
    int   precision;         /* 0: fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation).
                                1: fp32 EMULATED on the bf16 matrix cores -- every operand split into three bf16
                                pieces on the way to LDS, six bf16 MFMAs per product block, fp32 accumulation; the
                                result differs from mode 0 only by fp32 rounding (measured 1.2e-6 vs 1.1e-6 of
                                max|result| against fp64), at up to 2.6x the MFMA rate.  Same inputs, outputs, tiles. */
    const float* tail_planes[4];  /* planar tail operand: n_tail (1..4) extra input channels that live in dense one-channel
                                planes [B,h_in,w_in] instead of the NHWC buffer -- the LPG depth maps / reduc1x1 that the
                                reference concatenates behind the features (pytorch/bts.py:260, 274, 287: cat[upconv3, skip,
                                depth_8x8_scaled_ds] ...).  They are channels [c_in_ld-4, c_in_ld-4+n_tail) of the packed K
                                axis (c_in_ld = buffer channels + 4; weights of the unused tail slots are zero), x then
                                only needs c_in_ld-4 channels.  3x3, pad 1, stride 1, dil 1, up 1 only (the decoder's
                                conv3 / conv2 / conv1); computed on the fp32-input MFMA whatever `precision` says.         */
    int   n_tail;              /* 0 = no tail                                                                              */
    int   fill_frames;         /* 0 = default (8, or $BTS_CONV_FILL_FRAMES).  How many frames the caller expects to share one
                                launch: the launch-filling choices (whether and how far an under-filled layer splits K,
                                whether the wide 1x1 tile has enough pixel tiles) are sized for fill_frames x H x W pixels.
                                They are NEVER derived from B itself -- a frame's bits must not depend on its batch -- so a
                                caller that runs single frames (the reference's test loop, pytorch/bts_test.py:127-147) says
                                so here: fill_frames = 1 or 2 splits K on many more layers (352x1216, batch 1: 8.2 -> 5.7 ms of
                                GPU time per frame).  A batched caller declares the frames that share the chip (bench.py: its
                                per-GPU batch, at most 16): from 12-14 frames on, the DenseNet block-3 layers (22x76 maps) move
                                from the split-K kernels to the wide 1x1 tile and the halo kernel -- choices that would cost a
                                single-frame caller 45 % if they were tied to the default.  Results of different fill_frames
                                differ in summation order (fp32 rounding)                                                  */
    const void* w_split;       /* optional, precision = 1 only: the packed weights `w` pre-split into three bf16 planes --
                                [classes][3][c_out_pad][k_pad] bf16 (classes = 4 for sub-pixel, else 1), plane 0 = the top 16
                                bits of w, plane 1 = the top 16 bits of (w - plane 0), plane 2 = the top 16 bits of the rest
                                (the kernels' own truncation split; bts_amd/ops.py:split_bf16x3).  With it the halo-tile
                                kernel of the emulated mode streams weight tiles global -> LDS by LDS-DMA; NULL = the
                                row-tiled kernel splits `w` on the fly.  16-byte aligned.                                  */
    const float* w_wino;       /* optional, precision = 0, stride-1 3x3 / padding 1 / dilation 1 only: the weights in Winograd
                                F(2x2,3x3) form U = G g G^T, in MFMA B-fragment order [16 xi][c_in_ld/32][c_out_pad/32][4][64][4]
                                (bts_amd/ops.py:pack_wino_weight).  With it (and $BTS_CONV_WINO) eligible layers run the fused
                                Winograd kernel (csrc/conv_wino.inc): 2.25x fewer MFMA products, results equal to the direct
                                kernels' up to fp32 rounding of the transforms.  NULL = direct kernels.  16-byte aligned.   */
} bts_conv_desc;

int bts_conv_fwd_f32(const bts_conv_desc* desc, bts_stream_t stream);

/* Tap-sum stage of an upconv (nearest-2x + 3x3 conv + activation + affine, pytorch/bts.py:90-94 and the bn that follows,
 * :227, 231) computed as a TAP GEMM: first ONE 1x1 convolution of the source map with the nine kernel taps side by side
 * (bts_conv_fwd_f32, c_out = 9*c, weights [t*c + n][c_in] = w[n][c_in][t], no activation) into `taps`, then this call:
 *   y[b][2Y+py][2X+px][n] = e2(act( sum_{ky,kx} taps[b][Y+dy(py,ky)][X+dx(px,kx)][(3*ky+kx)*c + n] )),
 *   dy(0,.) = (-1,0,0), dy(1,.) = (0,0,+1), same for dx; source pixels outside the map contribute 0.
 * 9 tap-products per source pixel instead of the sub-pixel form's 16 (or the reference's 36): what a small, wide map wants
 * (upconv5: 11x38, 2208 -> 512).  taps: [B*h*w] pixels of taps_pix_stride >= 9*c floats; y: NHWC [B,2h,2w] with
 * y_pix_stride >= c; c % 4 == 0; act 0 none / 1 ReLU / 2 ELU; e2_* [c] or both NULL.  Sum order fixed (tap 0..8). */
int bts_upconv_combine_f32(const float* taps, long taps_pix_stride, int B, int h, int w, int c,
                           const float* e2_scale, const float* e2_shift, int act, float* y, long y_pix_stride,
                           bts_stream_t stream);

/* Which kernel bts_conv_fwd_f32 will launch for this descriptor (host-side query, no GPU work: it walks the real
 * dispatch path): lets a profiler attribute a launch to its kernel instantiation.
 *   kind & 15: 0 = conv_fwd_kernel (row-tiled, BM x BN), 1 = conv_halo_kernel (spatial 128-pixel tile x BN),
 *              2 = conv_halo_kernel with the planar tail operand, 3 = conv1x1_kernel (bm = 128 or 64 pixels x BN),
 *              4 = conv_stem_kernel (7x7 / stride-2 encoder stem, 8x32-pixel tiles x BN);
 *   kind & 16: split-K (+ splitk_reduce_kernel);  kind & 32: the eight-wave variant of the 48-wide halo tile (under-filled launches). */
int bts_conv_plan_f32(const bts_conv_desc* desc, int* bm, int* bn, int* kind);

/* Tap-steps the launch really issues vs. the dense count (host-side query, no GPU work).  The row-tiled kernel skips,
 * per 128/64-pixel row tile, the taps that fall into the zero padding for ALL of the tile's pixels (a dilated ASPP
 * branch, reference pytorch/bts.py:65-80: dilation 24 on a 44-row map leaves 6 of 9 taps for most tiles); the
 * skipped products are exact zeros, results are unchanged.  issued / dense = the share of the algorithmic FLOPs
 * the matrix pipe executes; both are 0 for launches that never skip (halo-tile and wide 1x1 kernels). */
int bts_conv_plan_ksteps_f32(const bts_conv_desc* desc, long* issued, long* dense);

/* ------------------------------------------------------------------------------------------
 * Training step (reference: autograd of the nn.Conv2d modules of pytorch/bts.py:70-77, 87-93, 108-119,
 * 180-221, driven by bts_main.py:476-500).  The input gradient of a stride-1 convolution IS a convolution
 * (flipped, transposed weights; pad' = dil*(ksize-1) - pad) and goes through bts_conv_fwd_f32; the weight
 * gradient is this entry point:
 *
 *   dw[co][tap][ci] = sum_p dy[p][co] * x[q(p,tap)][ci]        (q as in bts_conv_desc, incl. `up`)
 *
 * dw is written in OHWI order ([c_out][ksize*ksize][c_in], the layout the forward kernel packs from); the caller
 * permutes to the state dict's OIHW.  c_in % 4 == 0 and c_out % 4 == 0 (pad odd channel counts with zeros).
 * ws: optional scratch for the deterministic split over pixels; without it one workgroup column walks all pixels.
 */
typedef struct bts_conv_wgrad_desc {
    const float* x;         /* forward input, NHWC: pixel q channel c at x[q*x_pix_stride + c]           */
    long  x_pix_stride;
    int   c_in;
    const float* dy;        /* gradient of the forward output, NHWC [B,H,W,c_out]                        */
    long  dy_pix_stride;
    int   c_out;
    int   B, h_in, w_in;    /* forward input size before the optional upsample                           */
    int   up, ksize, dil, stride, pad;
    float* dw;              /* [c_out][ksize*ksize][c_in]                                                */
    float* ws;              /* NULL or scratch, exclusive to this stream while the call runs             */
    long  ws_floats;
    int   n_bundles;        /* 0/1: ordinary.  > 1: grouped convolution as channel bundles (see bts_conv_desc):
                               c_in / c_out are PER BUNDLE, bundle j uses x channels [j*c_in, ..) and dy channels
                               [j*c_out, ..); dw = [n_bundles][c_out][ksize*ksize][c_in] dense blocks, of which the
                               caller keeps each group's diagonal block                                            */
    const float* pre_scale; /* optional [n_bundles*c_in] (16-byte aligned): the forward convolution consumed           */
    const float* pre_shift; /* relu?(x*pre_scale + pre_shift) (a norm layer folded into its prologue, see               */
    int   pre_relu;         /* bts_conv_desc.pre_*); the gather applies the same transform, padding stays zero        */
} bts_conv_wgrad_desc;

int bts_conv_wgrad_f32(const bts_conv_wgrad_desc* desc, bts_stream_t stream);

/* Batched weight re-packing for the training step (the optimiser rewrites the OIHW parameters every iteration,
 * bts_main.py:606): one launch lays every registered weight out as bts_conv_fwd_f32 wants it.
 * table: n_entries device records of 14 int64 each --
 *   { src (OIHW float*), dst (float* [rows_pad][k_pad]), rows, inner, ksize, c_in_ld, rows_pad, k_pad,
 *     s_row, s_c (element strides of packed row / inner channel in src), flip (1 = spatially flipped taps),
 *     first_block (prefix sum of bts_pack_weights_blocks over the table),
 *     cg, gmode (grouped weights packed block-diagonally per bundle: channels per group and 1 = forward /
 *     2 = input-gradient layout; 0, 0 for dense weights) }
 *   forward layout : rows = c_out, inner = c_in, s_row = c_in*k*k, s_c = k*k, flip 0
 *   input gradient : rows = c_in, inner = c_out, s_row = k*k, s_c = c_in*k*k, flip 1   (transposed, flipped kernel)
 * dst[row][tap*c_in_ld + c] = src[row*s_row + c*s_c + (flip ? k*k-1-tap : tap)], zero elsewhere.
 */
long bts_pack_weights_blocks(long rows_pad, long k_pad);
int bts_pack_weights_f32(const void* table, int n_entries, long total_blocks, bts_stream_t stream);

/* Winograd F(2x2,3x3) form of a packed 3x3 weight (bts_conv_desc.w_wino; the reference has no counterpart: its 3x3
 * convolutions, pytorch/bts.py:83-94 / 183-221 and the DenseNet growth convolutions, go to cuDNN).  `w_packed` is the
 * [c_out_pad][k_pad] matrix bts_conv_fwd_f32 reads (K tap-major, k = tap * c_in_ld + c).  U = G g G^T per (output, input)
 * channel, computed in fp64 and rounded once, written in the B-fragment order of the fused Winograd kernel:
 *   32-wide channel tiles (c_out16 == 0): float ((((xi*nchunks + chunk)*n_ct + ct)*4 + g)*64 + lh*32 + li)*4 + q
 *       = U[xi][n = 32 ct + li][k = 32 chunk + 8 g + 4 lh + q],  n_ct = c_out_pad / 32;
 *   16-wide tiles (c_out16 = real output channels, a multiple of 16: the 48-wide DenseNet tile):
 *       ((((xi*nchunks + chunk)*n_ct + ct)*2 + g)*64 + l)*4 + q = U[xi][n = 16 ct + (l & 15)][k = 32 chunk + 16 g + 4 (l >> 4) + q].
 * n_tail > 0: the last 4 of the c_in_ld channels are the planar tail operand and are left out (the kernel adds their
 * products directly).  bts_pack_wino_floats = floats `out` must hold (16-byte aligned), or -1 for an unsupported shape
 * (buffer channels and c_out_pad must be whole multiples of 32). */
long bts_pack_wino_floats(int c_out_pad, int c_in_ld, int n_tail, int c_out16);
int bts_pack_wino_f32(const float* w_packed, int c_out_pad, long k_pad, int c_in_ld, int n_tail, int c_out16,
                      float* out, bts_stream_t stream);

/* Batch-statistic BatchNorm over NHWC rows [npix][C] (nn.BatchNorm2d in train() mode: pytorch/bts.py:69-76,
 * 182-202 and the DenseNet norm layers), C % 4 == 0, row strides % 4 == 0 and >= C (channel slices work in place).
 *
 * bts_bn_train_stats_f32 : per-channel batch mean and biased variance (deterministic tree of Chan-merged partials)
 *     -> mean, invstd = 1/sqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale; running_mean/var (may be
 *     NULL) updated as PyTorch does: r = (1-momentum)*r + momentum*{mean | unbiased var}.  gamma/beta NULL = 1/0.
 *     ws: scratch of bts_bn_train_ws_floats(npix, C) floats.
 * bts_bn_apply_nhwc_f32  : y = [relu](x*scale + shift)   (relu != 0 fuses the ReLU that follows the norm layer)
 * bts_bn_train_bwd_f32   : with dy' = dy masked by the fused ReLU (recomputed from x): dbeta = sum dy',
 *     dgamma = sum dy'*xhat, dx = scale*(dy' - dbeta/n - xhat*dgamma/n)  (dx may be NULL: statistics only).
 */
long bts_bn_train_ws_floats(long npix, int C);
int bts_bn_train_stats_f32(const float* x, long x_pix_stride, long npix, int C, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* ws, long ws_floats, float* mean, float* invstd, float* scale, float* shift,
                           bts_stream_t stream);
int bts_bn_apply_nhwc_f32(const float* x, long x_pix_stride, long npix, int C, const float* scale,
                          const float* shift, int relu, float* y, long y_pix_stride, bts_stream_t stream);
int bts_bn_train_bwd_f32(const float* x, long x_pix_stride, const float* dy, long dy_pix_stride, long npix, int C,
                         const float* mean, const float* invstd, const float* scale, const float* shift, int relu,
                         float* ws, long ws_floats, float* dgamma, float* dbeta, float* dx, long dx_pix_stride,
                         bts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layout movers between the NCHW boundary (pytorch/bts.py:347-349 tensors) and the NHWC
 * interior.  dst/src NHWC element (p,c) lives at base[p*pix_stride + c].
 *   relu != 0 applies max(v,0) on the way (bts.py:225: dense_features = ReLU(features[5])).
 */
int bts_nchw_to_nhwc_f32(const float* src, int B, int C, long HW, float* dst, long dst_pix_stride,
                         int relu, bts_stream_t stream);
int bts_nhwc_to_nchw_f32(const float* src, long src_pix_stride, int B, int C, long HW, float* dst,
                         bts_stream_t stream);

/* Interleave up to four 1-channel planes ([npix] each, e.g. reduc1x1 and the three LPG depth maps)
 * into consecutive channels of an NHWC buffer: dst[p*dst_pix_stride + i] = plane_i[p].
 * Builds the tail of concat1 = cat[upconv1, reduc1x1, depth_2x2, depth_4x4, depth_8x8]
 * (pytorch/bts.py:287) without a torch.cat pass.  Unused planes = NULL (n_planes 1..4).
 */
int bts_pack_planes_f32(const float* p0, const float* p1, const float* p2, const float* p3, int n_planes,
                        long npix, float* dst, long dst_pix_stride, bts_stream_t stream);

/* Encoder-side NHWC pooling (torchvision DenseNet pool0 = MaxPool2d(3,2,1), transition pool =
 * AvgPool2d(2,2); the encoder is the caller side of the hot path, pytorch/bts.py:295-338).
 * maxpool: [B,h,w,C] -> [B,ceil(h/2),ceil(w/2),C], optionally to two destinations.
 * bn_relu_avgpool2: dst = mean_{2x2} relu(src*scale + shift)  (transition norm+relu+pool; its 1x1 conv
 * commutes with the mean and runs afterwards on the pooled map).  C % 4 == 0, strides % 4 == 0.
 */
int bts_maxpool3x3s2_nhwc_f32(const float* src, long src_pix_stride, int B, int h, int w, int C, float* dst,
                              long dst_pix_stride, float* dst2, long dst2_pix_stride, bts_stream_t stream);
int bts_bn_relu_avgpool2_nhwc_f32(const float* src, long src_pix_stride, int B, int h, int w, int C,
                                  const float* scale, const float* shift, float* dst, long dst_pix_stride,
                                  bts_stream_t stream);

/* get_depth + final scaling (pytorch/bts.py:220-221, 289-291):
 *   final_depth[b,0,y,x] = max_depth * sigmoid( conv3x3(iconv1, w)[b,0,y,x] ) [* focal[b] / 715.0873]
 *   iconv1 : [B,C,H,W] NCHW (the tensor bts.forward returns), w : [1,C,3,3] as stored in the
 *   state dict (get_depth.0.weight), focal : [B] or NULL (non-KITTI datasets skip the focal term).
 */
int bts_get_depth_f32(const float* iconv1, const float* w, int B, int C, int H, int W, float max_depth,
                      const float* focal, float* final_depth, bts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics as GPU reductions.  Replaces the per-sample body of online_eval +
 * compute_errors (pytorch/bts_main.py:87-108, 221-254; same code in bts_eval.py:81-102, 237-307), which copies both
 * maps to the host and runs ~25 NumPy passes per sample.
 *
 *   pred  : [B,Hp,Wp] network output (metres).  With do_kb_crop the prediction is pasted into a zero canvas of the
 *           ground truth's size at (top,left) = (Hg-352, (Wg-1216)/2) (bts_main.py:221-227); otherwise Hp==Hg,
 *           Wp==Wg, top=left=0.
 *   gt    : [B,Hg,Wg] ground-truth depth.
 *   clamp : pred<min -> min, pred>max -> max, +inf -> max, NaN -> min (bts_main.py:229-232)
 *   valid : min < gt < max (bts_main.py:234) AND y in [y0,y1), x in [x0,x1) -- the Garg / Eigen crop rectangle of
 *           bts_main.py:236-249 in ground-truth coordinates (0,Hg,0,Wg = no crop)
 *   per_frame : [B][10] doubles <- silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3 (bts_main.py:84 order),
 *           then the number of valid pixels.
 *   accum : optional [10] doubles, the running `eval_measures` of online_eval: += the nine measures and [9] += 1 for
 *           every frame that has valid pixels (bts_main.py:253-254); all-reduced by the caller (bts_main.py:258-260).
 *   ws    : scratch of bts_eval_ws_doubles(B,Hg,Wg) doubles.
 * Sums are fp64 in a fixed order (no atomics): bit-reproducible; the reference reduces in float32.
 */
long bts_eval_ws_doubles(int B, int Hg, int Wg);
int bts_eval_depth_metrics_f32(const float* pred, int B, int Hp, int Wp, const float* gt, int Hg, int Wg,
                               int top, int left, float min_depth_eval, float max_depth_eval,
                               int y0, int y1, int x0, int x1, double* ws, long ws_doubles,
                               double* per_frame, double* accum, bts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-call execution of a recorded forward ("plan").  A BtsModel forward is a fixed sequence of ~120 (B=1) to ~460
 * (B=16, four sub-batches) launches of the entry points above with arguments that depend only on the input shape:
 * workspaces, packed weights and dims are fixed per shape, only the caller's input / output tensors move.  The host
 * records that sequence once per shape as an array of `bts_op` records -- bts_amd/plan.py -- and replays it with ONE call per stream:
 * the reference's inference loop is batch 1, eager (pytorch/bts_test.py:127-147), where the per-launch host path is
 * what bounds the frame rate.
 *
 *   ops      : the recorded calls, in order; each holds the argument list of its entry point (stream excluded)
 *   patches  : pointer fields that refer to per-call tensors: field := slots[slot] + delta (bytes), applied IN PLACE
 *              to `ops` before the launches (idempotent; a plan must not be replayed by two threads at once)
 *   slots    : base device pointers of this call's tensors (image, focal, the six outputs, abs_min scalars ...)
 * Returns the first non-zero code of an entry point (the remaining ops are not enqueued), 0 otherwise.
 */
enum { BTS_OP_CONV = 1, BTS_OP_REDUC = 2, BTS_OP_REDUC_LPG = 3, BTS_OP_LPG_FUSED = 4, BTS_OP_NCHW_TO_NHWC = 5,
       BTS_OP_NHWC_TO_NCHW = 6, BTS_OP_MAXPOOL = 7, BTS_OP_BN_RELU_AVGPOOL = 8, BTS_OP_GET_DEPTH = 9, BTS_OP_LPG = 10,
       BTS_OP_UPCONV_COMBINE = 11 };

typedef struct bts_op {
    int kind;          /* BTS_OP_*                                                              */
    int failed_code;   /* out: the entry point's return code when it was the one that failed   */
    union {
        bts_conv_desc conv;
        struct { const float* x; long x_pix_stride; long npix; int c_in, c_first_out; const float* w_frag; long w_frag_floats;
                 float max_depth; int is_final, normalize; float* out; } reduc;
        struct { const float* x; long x_pix_stride; int B, h, w, c_in, c_first_out; const float* w_frag; long w_frag_floats;
                 float max_depth; int upratio; float* plane4; float* depth_scaled; float* ds_out; float* abs_min; } reduc_lpg;
        struct { const float* plane4; int B, h, w, upratio, normalize; float max_depth; float* depth_scaled; float* ds_out;
                 int ds_factor; long ds_pix_stride; float* abs_min; } lpg_fused;
        struct { const float* plane_eq; int B, h, w, upratio; float* depth; float* abs_min; } lpg;
        struct { const float* src; int B, C; long HW; float* dst; long dst_pix_stride; int relu; } nchw_to_nhwc;
        struct { const float* src; long src_pix_stride; int B, C; long HW; float* dst; } nhwc_to_nchw;
        struct { const float* src; long src_pix_stride; int B, h, w, C; float* dst; long dst_pix_stride; float* dst2;
                 long dst2_pix_stride; } maxpool;
        struct { const float* src; long src_pix_stride; int B, h, w, C; const float* scale; const float* shift; float* dst;
                 long dst_pix_stride; } avgpool;
        struct { const float* iconv1; const float* w; int B, C, H, W; float max_depth; const float* focal;
                 float* final_depth; } get_depth;
        struct { const float* taps; long taps_pix_stride; int B, h, w, c; const float* e2_scale; const float* e2_shift; int act;
                 float* y; long y_pix_stride; } upconv_combine;
    } u;
} bts_op;

typedef struct bts_plan_patch { int op; int field_offset; int slot; int reserved; long delta; } bts_plan_patch;

int bts_plan_run(bts_op* ops, int n_ops, const bts_plan_patch* patches, int n_patches, void* const* slots, int n_slots,
                 bts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BTS_HIP_H_ */
