/*
 * rtpose_mi355x.h — C ABI of librtpose_mi355x.so (gfx950 / MI355X only).
 *
 * This is the drop-in boundary for ONE hot path of
 * tensorboy/pytorch_Realtime_Multi-Person_Pose_Estimation:
 *
 *   rtpose_vgg forward  ->  heat-map peak NMS + sub-pixel refine  ->  PAF
 *   line-integral scoring  ->  greedy limb assignment  ->  person grouping
 *
 * Plain pointers and sizes only: no torch / numpy types cross this line.
 * Device pointers are HIP device pointers; `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  The library never allocates device
 * memory on the forward path: the host language (Python + torch here) owns
 * every buffer and hands in pointers + byte counts that the *_bytes() queries
 * report.
 *
 * Each entry point cites the reference interface (file:line under
 * /root/reference) it stands in for.
 *
 * Return codes: 0 = ok, negative = RTPOSE_E_* below.  The library never
 * falls back to a CPU path: without a HIP device every compute entry point
 * returns RTPOSE_E_NODEVICE (or the HIP error, negated and offset).
 */
#ifndef RTPOSE_MI355X_H
#define RTPOSE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTPOSE_OK 0
#define RTPOSE_E_INVAL (-1)     /* bad argument / unsupported shape          */
#define RTPOSE_E_NODEVICE (-2)  /* no HIP device / HIP runtime error         */
#define RTPOSE_E_CAPACITY (-3)  /* a fixed-capacity device table overflowed  */
#define RTPOSE_E_STATE (-4)     /* call order violated (e.g. not bound)      */
#define RTPOSE_E_HIP(code) (-1000 - (int)(code))

/* ------------------------------------------------------------------------
 * 0. Library
 * ---------------------------------------------------------------------- */
const char* rtpose_version(void);
/* Last error text of the calling thread ("" if none). */
const char* rtpose_last_error(void);

/* ------------------------------------------------------------------------
 * 1. Activation layout ("shared-gap padded NHWC")
 *
 * Every activation tensor lives in HBM as rows of pixels, `cstride` floats
 * per pixel.  Pixel (n, y, x) sits at pixel index
 *        q = lead + (n * hs + y) * ws + x
 * with ws >= W + pad and hs >= H + pad, lead >= pad * ws + pad.  The pixels
 * that are not (n, y<H, x<W) are never written and stay zero, so a k x k
 * stencil tap (dy, dx) is the constant pixel offset dy * ws + dx and needs no
 * bounds test: the right gap of one row is the left gap of the next, and the
 * bottom gap of one image is the top gap of the next.
 * A plain dense NHWC tensor is the case lead = 0, ws = W, hs = H.
 * ---------------------------------------------------------------------- */
typedef struct rtpose_layout {
  int32_t cstride; /* floats per pixel                                     */
  int32_t choff;   /* first channel of the slice this view addresses       */
  int32_t ws;      /* row stride, pixels                                   */
  int32_t hs;      /* image stride, rows                                   */
  int32_t lead;    /* pixel index of (0,0,0)                               */
} rtpose_layout;

/* Pixels (not bytes) a buffer with this layout needs for N images of H x W,
 * including the tail slack the 2-D tile mode may read. */
size_t rtpose_layout_pixels(const rtpose_layout* l, int N, int H, int W);

/* ------------------------------------------------------------------------
 * 2. Convolution (stride 1, "same" padding, k in {1,3,7}) + bias (+ReLU)
 *    stands in for torch.nn.Conv2d + nn.ReLU as instantiated by
 *    lib/network/rtpose_vgg.py:23-35, :49-55 (ATen conv2d on the reference).
 *
 * Weights are consumed in the packed order produced by
 * rtpose_pack_conv_weights (shape independent).  fp32 in, fp32 accumulate
 * (v_mfma_f32_32x32x2_f32), fp32 out.
 * ---------------------------------------------------------------------- */
/* floats needed for the packed form of a [cout][cin][k][k] filter */
size_t rtpose_packed_weight_floats(int cout, int cin, int k);
/* floats needed for the padded bias */
size_t rtpose_packed_bias_floats(int cout);

/* Pack device OIHW fp32 weights (+bias) for the conv kernel.
 * `cin_map` (device or NULL): packed input channel c reads source channel
 * cin_map[c] (-1 = zero); NULL = identity over cin_src channels.
 * cin_packed is the channel count of the activation slice the conv will read
 * (>= cin_src, multiple of 8). */
int rtpose_pack_conv_weights(const float* w_oihw, const float* bias, int cout,
                             int cin_src, int k, const int32_t* cin_map,
                             int cin_packed, float* w_packed,
                             float* bias_packed, void* stream);

typedef struct rtpose_conv_desc {
  const float* in;       /* activation buffer base (layout `lin`)           */
  const float* w_packed; /* from rtpose_pack_conv_weights                   */
  const float* bias_packed;
  float* out;            /* activation buffer base (layout `lout`)          */
  rtpose_layout lin;
  rtpose_layout lout;
  int32_t cin;  /* packed input channels (multiple of 8)                    */
  int32_t cout; /* real output channels                                     */
  int32_t k;    /* 1, 3 or 7                                                */
  int32_t relu; /* 0/1                                                      */
  int32_t pool; /* 0, or 1 = fuse MaxPool2d(2,2,0): `lout` is then the
                   half-resolution layout (rtpose_vgg.py:49-50)             */
  const int32_t* out_cmap; /* device int32[cout] or NULL: output channel n is
                   written at channel out_cmap[n] of the pixel (absolute, lout.choff
                   ignored) - folds channel_shuffle / concat of the ShuffleNetV2
                   blocks (rtpose_shufflenetV2.py:56-62) into the store        */
  int32_t wino_m; /* rtpose_conv2d_winograd* only (rtpose_conv2d ignores it): the form of THIS launch.
                   k = 3: 0 or 2 = F(2x2,3x3), 4 = F(4x4,3x3);  k = 7: 6 = F(6,7), 4 = F(4,7), 0 = the
                   library default (6; 4 with RTPOSE_WINOGRAD7_M=4 in the environment).  `w_packed` must
                   come from the packing of the same form.  Any other value is refused
                   (RTPOSE_E_INVAL): ZERO-INITIALISE descriptors (memset / `= {0}`) - the struct has
                   grown by trailing fields and may again.                                  */
  int32_t in_plane_pixels;  /* F(4x4,3x3) launches (k = 3, wino_m = 4) only; every other launch refuses a non-zero
                   value.  0: `in` is pixel-major as §1 describes.  Q > 0: the input slice is stored as CHANNEL PLANES -
                   channel c of the pixel with index q (§1) at float ((c / 8) * Q + q) * 8 + c % 8 of `in`, c counted
                   from channel 0 of the buffer (lin.choff, a multiple of 8, selects the first plane; lin.cstride is not
                   used); Q >= rtpose_layout_pixels(&lin, N, H, W) is the number of pixel slots per plane.  A chunk of
                   the kernel reads 8 channels of the 36 pixels of a patch: in planes those are whole 128-byte lines
                   instead of a quarter of each (csrc/conv_wino4.hip; 3x3 layers of the 32-image forward -8 %).       */
  int32_t out_plane_pixels; /* the same for the output slice (cout and lout.choff multiples of 8): a chain of F(4x4,3x3)
                   convs keeps its intermediates in planes, its first launch reads and its last one writes pixel-major.
                   rtpose_conv_first_planes writes planes for conv1_1.                                               */
} rtpose_conv_desc;

/* Launch one conv, or `ngroups` (<= 2) convs of identical geometry in one
 * grid (the L1/L2 branches of a CPM stage, rtpose_vgg.py:163-164). */
int rtpose_conv2d(const rtpose_conv_desc* d, int ngroups, int N, int H, int W,
                  void* stream);

/* ---- the first layer: nn.Conv2d(3, 64, 3, 1, 1) (+ nn.ReLU), conv1_1 of the VGG-19 front end
 * (lib/network/rtpose_vgg.py:23-35, `model0.0`; csrc/conv_first.hip).  Reads the image where it is -
 * dense NCHW fp32 (`x_nchw`), or, with x_nchw = NULL, a layout buffer with >= 3 channels per pixel
 * (`x_layout` / `lx`: the plan's NHWC8 input written by rtpose_preprocess_u8) - and writes 64 channels
 * into `out` / `lout`: no NCHW -> NHWC conversion pass, no padding of the 3 input channels to 8.
 * `w_packed` (rtpose_conv_first_packed_floats() floats) from rtpose_pack_conv_first(w [64,3,3,3], bias [64]). */
size_t rtpose_conv_first_packed_floats(void);
int rtpose_pack_conv_first(const float* w_oihw, const float* bias, float* w_packed, void* stream);
int rtpose_conv_first(const float* x_nchw, const float* x_layout, const rtpose_layout* lx,
                      const float* w_packed, float* out, const rtpose_layout* lout, int relu, int N,
                      int H, int W, void* stream);
/* The same conv writing its 64 channels as 8 channel planes of `out_plane_pixels` pixel slots each (see
 * rtpose_conv_desc.in_plane_pixels; out_plane_pixels >= rtpose_layout_pixels(lout, N, H, W), lout->choff a multiple of 8,
 * lout->cstride unused): computed as the transposed product - a lane holds one pixel and 16 channels - so that 32 lanes
 * store 1 KB runs of a plane.  Same values, bit for bit. */
int rtpose_conv_first_planes(const float* x_nchw, const float* x_layout, const rtpose_layout* lx,
                             const float* w_packed, float* out, const rtpose_layout* lout, int out_plane_pixels,
                             int relu, int N, int H, int W, void* stream);

/* The same layer in the bf16 plan's arithmetic (BASELINE configs[2]; round 6): image and filters rounded to bf16 (round to
 * nearest even - what the plan's input conversion and rtpose_pack_conv_weights_bf16 do), products exact, fp32 accumulation,
 * + fp32 bias (+ReLU), output rounded to bf16.  `out` holds bf16 elements, `lout` counts them (cstride and choff multiples
 * of 8: a lane stores the 8 channels of a 16-byte piece); `w_packed` from rtpose_pack_conv_first_bf16 (the same
 * rtpose_conv_first_packed_floats() floats, filters rounded).  The source is fp32 either way - the image or an fp32 layout
 * buffer: the plan's NCHW -> bf16 NHWC16 conversion launch and its generic 16-channel conv1_1 are both replaced by this one
 * (the product of two bf16 values is exact in fp32, so the fp32 matrix instruction on rounded operands is this arithmetic). */
int rtpose_pack_conv_first_bf16(const float* w_oihw, const float* bias, float* w_packed, void* stream);
int rtpose_conv_first_bf16(const float* x_nchw, const float* x_layout, const rtpose_layout* lx,
                           const float* w_packed, void* out_bf16, const rtpose_layout* lout, int relu, int N,
                           int H, int W, void* stream);

/* ---- two pointwise convs back to back: nn.Conv2d(128, 128, 1) + nn.ReLU -> nn.Conv2d(128, cout2 <= 64, 1), the
 * Mconv6 / Mconv7 pair that ends every stage-2..6 branch (lib/network/rtpose_vgg.py:120-127), as ONE launch
 * (csrc/conv_tail.hip; `ngroups` <= 2 branches per grid).  d1[g] / d2[g] are rtpose_conv_desc of the two convs
 * with the plain k = 1 packing (rtpose_pack_conv_weights); d1[g].out / lout are ignored - the 128-channel
 * intermediate never leaves the CU - and d2[g] reads it.  Same sums, in the same order, as two rtpose_conv2d
 * launches.  rtpose_conv1x1_pair_fits: 1 if the pair has this shape. */
int rtpose_conv1x1_pair_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups);
int rtpose_conv1x1_pair(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N,
                        int H, int W, void* stream);

/* ---- fp32 Winograd forms of the 3x3 and 7x7 convs (csrc/conv_wino.hip, csrc/conv_wino7.hip) ------
 * Same module boundary as rtpose_conv2d (nn.Conv2d + nn.ReLU (+ nn.MaxPool2d) of
 * lib/network/rtpose_vgg.py:23-35, :49-55, :108-127), fewer matrix-core multiplies:
 *   k = 3: F(2x2, 3x3), 16 instead of 36 multiplies per 2 x 2 outputs and input channel (2.25x);
 *   k = 7: F(6, 7) along x, direct along y: 84 instead of 294 per 6 outputs (3.5x), no fused pool;
 *          or F(4, 7), 70 instead of 196 per 4 outputs (2.8x), selected per launch with
 *          rtpose_conv_desc.wino_m = 4 and the matching packing (rtpose_pack_conv_weights_winograd7).
 * fp32 MFMA throughout; results differ from the direct sum by rounding only (k = 3: a few ulp,
 * k = 7: ~3e-5 at magnitude 4; whole network < 4e-5 on the stage outputs; contract 1e-3).
 * The descriptor is rtpose_conv_desc with `w_packed` from rtpose_pack_conv_weights_winograd
 * (transformed filters: 16/9 resp. 70/49 the size) and out_cmap = NULL.
 * rtpose_conv2d_winograd_fits: 1 when the conv described by `d` (k, cin, cout, pool, lin.hs) has a
 * Winograd instance at N x H x W (k = 3: cin a multiple of 16, or of 8 when cout_pad is 64 modulo
 * 128; k = 7: cin a multiple of 8, cout_pad a multiple of 128, and the transformed rows of a block
 * fit the LDS), else 0 - callers then use rtpose_conv2d with the plain packing. */
int rtpose_conv2d_winograd_fits(const rtpose_conv_desc* d, int N, int H, int W);
size_t rtpose_packed_weight_floats_winograd(int cout, int cin, int k);
int rtpose_pack_conv_weights_winograd(const float* w_oihw, const float* bias, int cout,
                                      int cin_src, int k, const int32_t* cin_map,
                                      int cin_packed, float* w_packed, float* bias_packed,
                                      void* stream);
int rtpose_conv2d_winograd(const rtpose_conv_desc* d, int ngroups, int N, int H, int W,
                           void* stream);
/* k = 3 with an explicit form m: 0 / 2 = F(2x2,3x3) (as above), 4 = F(4x4,3x3) (csrc/conv_wino4.hip): 36 instead of
 * 144 multiplies per 4 x 4 outputs and input channel (4x fewer than the direct sum), interpolation points
 * 0, +-3/4, +-3/2, inf; cin a multiple of 16, >= 32; transformed filters 36/9 the size.  Launch with
 * rtpose_conv_desc.wino_m = 4 and this packing.  Results differ from the direct sum by rounding only
 * (element-wise error bound ~3x F(2x2,3x3)'s, tests/test_wino_numerics_gpu.py). */
size_t rtpose_packed_weight_floats_winograd3(int cout, int cin, int m);
int rtpose_pack_conv_weights_winograd3(const float* w_oihw, const float* bias, int cout,
                                       int cin_src, int m, const int32_t* cin_map,
                                       int cin_packed, float* w_packed, float* bias_packed,
                                       void* stream);
/* k = 7 with an explicit form m (4 or 6; 0 = default), see rtpose_conv_desc.wino_m */
size_t rtpose_packed_weight_floats_winograd7(int cout, int cin, int m);
int rtpose_pack_conv_weights_winograd7(const float* w_oihw, const float* bias, int cout,
                                       int cin_src, int m, const int32_t* cin_map,
                                       int cin_packed, float* w_packed, float* bias_packed,
                                       void* stream);
/* The 7x7 kernel balances launches whose tiles do not come out as whole rounds over the CUs by
 * running persistent blocks that split tiles (results bit-identical either way).  A split tile is
 * handed from one block to the next through `scratch`: device memory OWNED BY THE CALLER
 * (256-byte aligned, rtpose_conv2d_winograd_scratch_bytes() of it, for the current device),
 * because the library allocates nothing on the forward path.  One scratch serves all launches that
 * are serialised on one stream.  With scratch = NULL (and through rtpose_conv2d_winograd) every
 * launch runs one block per tile.  The last int of the flag area is a device error word: bit 0 is
 * raised if a hand-over wait ever ran out (results of that launch are then invalid);
 * rtpose_conv2d_winograd_scratch_error copies it back (synchronises the stream). */
size_t rtpose_conv2d_winograd_scratch_bytes(void);
int rtpose_conv2d_winograd_ex(const rtpose_conv_desc* d, int ngroups, int N, int H, int W,
                              void* scratch, size_t scratch_bytes, void* stream);
int rtpose_conv2d_winograd_scratch_error(const void* scratch, int* error_word, void* stream);
/* Amplification estimate of a filter bank w[cout][cin][k][k] (device, OIHW fp32) in Winograd form:
 * k = 3 -> F(2x2,3x3); k = 7 -> F(m,7), m = 4 or 6 (0 = default).  Writes ONE float to the device
 * address `amp_device`: the worst ratio over the output channels of the element-wise rounding-error
 * BOUND of the form to the direct sum's for inputs of uniform magnitude,
 *   max_i sum_f |AT[i][f]| (sum_n |BT[f][n]|) sum_{c,ky} |U[ky][f][c][o]|  /  sum_{c,ky,kx} |w[o][c][ky][kx]|
 * (i.i.d. Gaussian filters: 3.3, 62, 115; k = 3, m = 4 - F(4x4,3x3): see DESIGN.md §3.0).  The rtpose_vgg executor uses it to choose the form of a
 * layer (rtpose_net_options.winograd7 = RTPOSE_WINO7_AUTO). */
int rtpose_winograd_amplification(const float* w_oihw, int cout, int cin, int k, int m,
                                  float* amp_device, void* stream);

/* ---- fused pointwise chain of the ShuffleNetV2 pose network (BASELINE configs[3]) ----
 * stands in for lib/network/rtpose_shufflenetV2.py BasicBlock (:22-63): conv_bn_relu 1x1, optionally
 * preceded by the conv_bn depthwise 3x3 (stride 1) that feeds it and followed by
 * torch.cat((x1, x2), 1) + channel_shuffle(2) - ONE launch: the depthwise result is produced in LDS
 * and never stored, the pass-through half x1 is copied to its shuffled slots by the same blocks.
 *   out[p][cmap(n)] = act( bias[n] + sum_c A[p][c] * W[n][c] ),
 *   A[p][c] = in[p][c]                                            (dw_w == NULL)
 *           = dw_b[c] + sum_{ky,kx} dw_w[ky*3+kx][c] * in[p + (ky-1, kx-1)][c]   (zero padding = the layout gap)
 *   out[p][pt_cmap[i]] = pt_src[p][i], i < pt_c                   (pt_src != NULL)
 * fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32) / fp32 out. */
typedef struct rtpose_pw_desc {
  const float* in;          /* activation buffer base (layout `lin`; gap >= 1 when dw_w != NULL)   */
  const float* dw_w;        /* NULL, or device [9][cin] depthwise taps (BN folded), tap-major          */
  const float* dw_b;        /* device [cin] depthwise bias                                            */
  const float* w_packed;    /* from rtpose_pack_pw_weights: [cin/4][coutp][4]                          */
  const float* bias_packed; /* [coutp]                                                                */
  float* out;               /* activation buffer base (layout `lout`)                                 */
  rtpose_layout lin;
  rtpose_layout lout;
  int32_t cin;              /* packed input channels (multiple of 8)                                  */
  int32_t cout;             /* columns that are stored                                                */
  int32_t coutp;            /* columns of the packed matrix: 64, 128 or a multiple of 256             */
  int32_t relu;
  const int32_t* out_cmap;  /* NULL (column n -> channel lout.choff + n) or device int32[coutp]: column
                               n -> absolute channel of the pixel, < 0 = not stored                   */
  const float* pt_src;      /* NULL, or the buffer holding the pass-through channels (layout `lpt`)   */
  rtpose_layout lpt;
  const int32_t* pt_cmap;   /* device int32[pt_c]: pass-through channel i -> absolute output channel  */
  int32_t pt_c;
  /* pass-through, interleave form (pt_pairs > 0; pt_cmap / pt_c unused): output channel j < 2 pt_pairs is
   * source channel pt_a + j/2 (j even) or pt_b + j/2 (j odd) of `pt_src`, stored at channel
   * (j < pt_split ? pt_d0 + j : pt_d1 + j - pt_split): the next unit's x1 when both buffers keep their
   * channels as four runs [even-low | even-high | odd-low | odd-high] (contiguous loads and stores).     */
  int32_t pt_pairs, pt_a, pt_b, pt_split, pt_d0, pt_d1;
  const int32_t* in_planes; /* NULL (the input is the contiguous slice lin.choff .. + cin) or device
                               int32[cin / 4]: channel offset, relative to lin.choff, of every 4-channel
                               group of the GEMM's K axis (a 16-byte granular gather: x2 of a unit is two runs) */
} rtpose_pw_desc;
size_t rtpose_packed_pw_floats(int cin_packed, int coutp);
/* w_oi: device [cout][cin_src] (a 1x1 conv's OIHW weights), written to columns
 * [col_off, col_off + cout) of the packed matrix - several layers may share one matrix (the PAF and
 * heat-map heads, rtpose_shufflenetV2.py:107-108, run as one 128-column GEMM). */
int rtpose_pack_pw_weights(const float* w_oi, const float* bias, int cout, int cin_src,
                           const int32_t* cin_map, int cin_packed, int coutp, int col_off,
                           float* w_packed, float* bias_packed, void* stream);
int rtpose_pw_fused(const rtpose_pw_desc* d, int N, int H, int W, void* stream);

/* Column-mapped packing: packed column col_off + i = output channel col_map[i] of w_oi (NULL: i; < 0 or >= cout: a
 * zero column, zero bias) for i < ncols; K row c reads input channel cin_map[c] (NULL: c; < 0: a zero row).  The
 * ShuffleNetV2 plans store a layer's columns [0, cout) as CONTIGUOUS channels (out_cmap = NULL), so a layer that
 * writes runs of the four-run layout has its columns packed in memory order, and K may be padded with zero rows. */
int rtpose_pack_pw_weights_cols(const float* w_oi, const float* bias, int cout, int cin_src,
                                const int32_t* cin_map, int cin_packed, int ncols, const int32_t* col_map,
                                int coutp, int col_off, float* w_packed, float* bias_packed, void* stream);

/* ---- conv5 + both heads of the ShuffleNetV2 pose network as ONE back-to-back GEMM launch (csrc/pw_head.hip):
 *   slim.conv_bn_relu('conv5', 464, 1024, 1) -> { self.paf = nn.Conv2d(1024, 38, 1), self.heatmap = nn.Conv2d(1024, 19, 1) }
 * (lib/network/rtpose_shufflenetV2.py:104, :107-108, :143-147).  d1 = the wide conv (+ReLU): `in` / `lin` a contiguous
 * slice of cin channels (a multiple of 16, <= 1024) or, with d1->in_planes, a gather of cin / 4 16-byte planes of
 * the pixel; w_packed [cin / 4][coutp][4] with cout = coutp a multiple of 256; d1->out is
 * ignored - the intermediate never leaves the registers.  d2 = the heads: ONE shared matrix [coutp1 / 4][64][4] whose
 * columns sit at their output channels (rtpose_pack_pw_weights with col_off; columns nobody owns must be zero), bias
 * [64]; 64 channels are stored at d2->lout.choff (16 bytes per lane, no column map).  Both GEMMs run transposed so
 * that the accumulators of the first are the B operand of the second (see the source).  fp32 in / accumulate / out. */
int rtpose_pw_head_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2);
int rtpose_pw_head(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, void* stream);

/* bf16 form (the bf16 plan of BASELINE configs[3]): bf16 activations and pointwise weights, fp32 accumulate
 * (v_mfma_f32_32x32x16_bf16), depthwise taps / biases fp32, outputs rounded to bf16 (round-to-nearest-even)
 * or written fp32 (out_f32 != 0: the two heads).  In the desc `in`, `out`, `pt_src` point at 2-byte elements
 * (out: 4-byte when out_f32), the layouts count ELEMENTS, slices are multiples of 8 elements, cin a
 * multiple of 16.  The bf16 epilogue stores the GEMM's columns [0, cout) as the CONTIGUOUS channels
 * lout.choff .. (16 bytes per lane), so a layer that writes runs of the four-run layout is packed with a
 * column map (`col_map[i]` = output channel of packed column col_off + i, < 0 = a zero column).  out_cmap, if given,
 * is read per GROUP OF 8 COLUMNS by the bf16 epilogue: the columns 8 g .. 8 g + 7 are stored as the 8 contiguous
 * channels from out_cmap[8 g] (absolute, a multiple of 8; < 0: the group is not stored) - the zero-copy channel
 * shuffle of the ShuffleNetV2 plans scatters a layer's output over the slot groups the unit's x2 vacated; the fp32
 * (out_f32) epilogue reads it per column.  The pass-through half exists in its interleave form only.
 * Contract: oracle/shufflenet_oracle.py:forward_bf16_emulated (the reference has no bf16 path). */
size_t rtpose_packed_pw_bytes_bf16(int cin_packed, int coutp);
int rtpose_pack_pw_weights_bf16(const float* w_oi, const float* bias, int cout, int cin_src,
                                const int32_t* cin_map, int cin_packed, int ncols,
                                const int32_t* col_map, int coutp, int col_off, void* w_packed,
                                float* bias_packed, void* stream);
int rtpose_pw_fused_bf16(const rtpose_pw_desc* d, int out_f32, int N, int H, int W, void* stream);

/* conv5 + both heads as ONE launch in the bf16 plan (csrc/pw_head_bf16.hip; same reference lines as rtpose_pw_head):
 * d1 = the wide conv (+ReLU): bf16 input slice (layouts count ELEMENTS, multiples of 8) or a gather of cin / 8 16-byte
 * planes (d1->in_planes), cin a multiple of 16 up to 480, w_packed [cin / 8][coutp][8 bf16] from
 * rtpose_pack_pw_weights_bf16 (plain column order), cout = coutp a multiple of 256 up to 1024; d2 = the heads: ONE shared matrix [coutp1 / 8][64][8 bf16] packed by rtpose_pack_pw_head2_bf16
 * (the k order in which the first GEMM's accumulators come out of the matrix pipe; columns at their output channels,
 * columns nobody owns must be zero), fp32 bias [64], 64 fp32 channels stored at d2->lout.choff.  The conv5 feature is
 * rounded to bf16 (RNE) after its ReLU and never leaves the registers; sums are fp32 in a fixed order. */
/* One stride-1 ShuffleNetV2 unit - conv.0 (1x1 + ReLU) -> conv.1 (depthwise 3x3) -> conv.2 (1x1 + ReLU),
 * lib/network/rtpose_shufflenetV2.py:31-39 - as ONE launch in the bf16 plan (csrc/unit_bf16.hip): the conv.0 output only
 * exists in LDS (8 x 8 output tiles, conv.0 recomputed on the 10 x 10 halo; halo pixels outside the image are zero, as
 * the depthwise conv's padding wants).  d0 = conv.0: `in` / `lin` the stage buffer (bf16, ELEMENT counts, layout gap >= 1)
 * as a contiguous slice or a gather of cin / 8 planes (in_planes), cin a multiple of 16 up to 256, w_packed
 * [cin / 8][coutp][8 bf16] with coutp = 128 or 256 and zero columns past the real ones; d0->dw_w / dw_b = conv.1's fp32
 * taps [9][Kt] and bias [Kt], Kt = d2->cin (a multiple of 16, <= d0->coutp).  d2 = conv.2: w_packed [Kt / 8][coutp][8 bf16],
 * coutp = 128 or 256, `cout` existing columns (a multiple of 8), out_cmap[column] = absolute channel of the output pixel
 * (groups of 8 columns contiguous and 8-aligned, < 0: not stored), `out` / `lout` = the output buffer (may be the input
 * buffer when the output channels are not read by this launch: other blocks read x2 as their halo). */
int rtpose_unit_bf16_fits(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int H, int W);
int rtpose_unit_bf16(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int N, int H, int W, void* stream);

int rtpose_pw_head_bf16_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2);
int rtpose_pw_head_bf16(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, void* stream);
int rtpose_pack_pw_head2_bf16(const float* w_oi, const float* bias, int cout, int cin, int col_off, void* w_packed,
                              float* bias_packed, void* stream);

/* ---- bf16 variant (BASELINE config 3: "bf16, multi-scale x4 + flip") ----------------
 * Same modules, bf16 activations and weights, fp32 accumulate
 * (v_mfma_f32_32x32x16_bf16), bias/ReLU/pool in fp32, output rounded to bf16
 * (round-to-nearest-even) or written as fp32 (out_f32 != 0).  The reference has no
 * reduced-precision path; the contract is pinned by oracle/net_oracle.py
 * (forward_bf16_emulated).  In a desc used here `in`, `w_packed` and `out` point at 2-byte
 * elements (out: 4-byte when out_f32) and the layouts count ELEMENTS per pixel; input slices
 * are multiples of 8 elements, cin a multiple of 16. */
size_t rtpose_packed_weight_bytes_bf16(int cout, int cin, int k);
int rtpose_pack_conv_weights_bf16(const float* w_oihw, const float* bias, int cout,
                                  int cin_src, int k, const int32_t* cin_map,
                                  int cin_packed, void* w_packed, float* bias_packed,
                                  void* stream);
int rtpose_conv2d_bf16(const rtpose_conv_desc* d, int ngroups, int N, int H, int W,
                       int out_f32, void* stream);
/* Two pointwise convs back to back in this arithmetic (round 6; csrc/conv_tail_bf16.hip - the bf16 sibling of
 * rtpose_conv1x1_pair): nn.Conv2d(128, 128 | 512, 1) + nn.ReLU -> nn.Conv2d(128 | 512, cout2 <= 64, 1), the pair that ends every
 * stage branch (lib/network/rtpose_vgg.py:101-105, :120-127), `ngroups` <= 2 branches per grid, as ONE launch: the
 * intermediate is rounded to bf16 where the two-launch form rounded it and never leaves the CU.  d1[g] / d2[g] are the
 * descs of the two convs with the k = 1 packing of rtpose_pack_conv_weights_bf16; d1[g].in / lin: 16-byte aligned bf16
 * slice; d1[g].out is ignored; d2[g].out / lout: bf16 elements, or fp32 when out_f32 (any channel offset).  Same contract
 * as two rtpose_conv2d_bf16 launches; the fp32 sums run in another order, so not bit for bit the same. */
int rtpose_conv1x1_pair_bf16_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups);
int rtpose_conv1x1_pair_bf16(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N,
                             int H, int W, int out_f32, void* stream);
/* 3x3 convs with 64 bf16 INPUT channels - conv1_2 (+ fused 2x2 max-pool) and conv2_1 of the VGG-19 front end
 * (lib/network/rtpose_vgg.py:23-35, :69-72) - in their own kernel (round 6; csrc/conv_c64_bf16.hip): the whole K = 576 of a
 * 16 x 32 pixel tile in one LDS halo, persistent blocks, 16-byte stores from the accumulators.  d[0]: k = 3, cin = 64,
 * cout a multiple of 64, the packing of rtpose_pack_conv_weights_bf16, bf16 input and output slices 16-byte aligned, no
 * out_cmap; same contract as rtpose_conv2d_bf16, which takes this path by itself whenever `_fits` says 1 (the fp32 sums run
 * in another order than in its generic kernel: the last bits).  `_fits` is host-only. */
int rtpose_conv3x3_c64_bf16_fits(const rtpose_conv_desc* d, int ngroups, int N, int H, int W);
int rtpose_conv3x3_c64_bf16(const rtpose_conv_desc* d, int N, int H, int W, void* stream);
/* dense NCHW fp32 -> bf16 layout slice (channels [C, cpad) zero; cpad % 8 == 0) */
int rtpose_nchw_to_layout_bf16(const float* src_nchw, void* dst, const rtpose_layout* ldst,
                               int C, int cpad, int N, int H, int W, void* stream);
/* fp32 layout slice -> bf16 layout slice, and back */
int rtpose_layout_f32_to_bf16(const float* src, const rtpose_layout* lsrc, void* dst,
                              const rtpose_layout* ldst, int C, int cpad, int N, int H,
                              int W, void* stream);
int rtpose_layout_bf16_to_f32(const void* src, const rtpose_layout* lsrc, float* dst,
                              const rtpose_layout* ldst, int C, int N, int H, int W,
                              void* stream);

/* ---- bf16x3 variant: fp32-grade results from the bf16 matrix pipe ---------------------
 * Every fp32 operand v travels as two bf16s, hi = bf16(v) and lo = bf16(v - hi) (16 significant
 * bits), and a product is accumulated as hi*lo + lo*hi + hi*hi in fp32 (the lo*lo term, 2^-18
 * relative, is dropped): 3 MFMAs of the 16x faster bf16 pipe instead of fp32 MFMAs.  Activations
 * are stored per 8 channels as [hi x 8 | lo x 8] (32 B: the fp32 footprint); layouts of such
 * buffers count ELEMENTS (2 per channel, slices on 8-channel boundaries), `cin`/`cout` still
 * count channels.  Contract: oracle/net_oracle.py:forward_bf16x3_emulated; measured against the
 * fp32 oracle it stays inside the 1e-3 bound of the fp32 path (tests/test_bf16x3_gpu.py). */
size_t rtpose_packed_weight_bytes_bf16x3(int cout, int cin, int k);
int rtpose_pack_conv_weights_bf16x3(const float* w_oihw, const float* bias, int cout,
                                    int cin_src, int k, const int32_t* cin_map,
                                    int cin_packed, void* w_packed, float* bias_packed,
                                    void* stream);
int rtpose_conv2d_bf16x3(const rtpose_conv_desc* d, int ngroups, int N, int H, int W,
                         int out_f32, void* stream);
int rtpose_nchw_to_layout_split(const float* src_nchw, void* dst, const rtpose_layout* ldst,
                                int C, int cpad, int N, int H, int W, void* stream);
int rtpose_layout_f32_to_split(const float* src, const rtpose_layout* lsrc, void* dst,
                               const rtpose_layout* ldst, int C, int cpad, int N, int H,
                               int W, void* stream);
int rtpose_layout_split_to_f32(const void* src, const rtpose_layout* lsrc, float* dst,
                               const rtpose_layout* ldst, int C, int N, int H, int W,
                               void* stream);

/* MaxPool2d(kernel 2, stride 2, pad 0) between two layouts
 * (rtpose_vgg.py:49-50; floor semantics of nn.MaxPool2d). */
int rtpose_maxpool2x2(const float* in, const rtpose_layout* lin, float* out,
                      const rtpose_layout* lout, int C, int N, int H, int W,
                      void* stream);

/* NCHW dense fp32 -> layout (channels [0,C) of the slice; extra channels of
 * the slice up to cpad are written as zero). */
int rtpose_nchw_to_layout(const float* src_nchw, float* dst,
                          const rtpose_layout* ldst, int C, int cpad, int N,
                          int H, int W, void* stream);
/* layout slice -> dense NCHW fp32 */
int rtpose_layout_to_nchw(const float* src, const rtpose_layout* lsrc,
                          float* dst_nchw, int C, int N, int H, int W,
                          void* stream);
/* layout slice -> layout slice (C channels) */
int rtpose_layout_copy(const float* src, const rtpose_layout* lsrc, float* dst,
                       const rtpose_layout* ldst, int C, int N, int H, int W,
                       void* stream);

/* ---- ShuffleNetV2 building blocks (lib/network/rtpose_shufflenetV2.py) -------
 * All HBM-bound, one pass.  BatchNorm (eval) is folded into weights/bias by the host. */
/* NCHW -> layout with a per-channel affine (the input BatchNorm2d(3), :96);
 * scale/shift: device float[C] or NULL. */
int rtpose_nchw_to_layout_affine(const float* src_nchw, float* dst, const rtpose_layout* ldst,
                                 int C, int cpad, int N, int H, int W, const float* scale,
                                 const float* shift, void* stream);
/* dense 3x3 stride-2 pad-1 conv for a tiny input channel count (stem, :97):
 * w packed [ky][kx][cin_pad][cout] fp32, cout % 4 == 0, bias[cout], fused ReLU. */
int rtpose_stem_conv3x3_s2(const float* in, const rtpose_layout* lin, const float* w,
                           const float* bias, float* out, const rtpose_layout* lout,
                           int cin_pad, int cout, int N, int H, int W, int relu, void* stream);
/* The same conv fused with what precedes it (:96-97): dense NCHW fp32 image -> per-channel
 * affine (the input BatchNorm2d(3); scale/shift device float[3] or NULL) -> conv 3x3 s2 p1
 * (zero padding applied after the affine, as in the reference graph) -> ReLU -> NHWC. */
int rtpose_stem_conv3x3_s2_nchw(const float* x_nchw, const float* scale, const float* shift,
                                const float* w, const float* bias, float* out,
                                const rtpose_layout* lout, int cout, int N, int H, int W,
                                int relu, void* stream);
/* The stem conv AND the max-pool that follows it (:96-99) in one launch: the 184 x 184 x 24 conv tensor is
 * never stored.  out: fp32 or (out_bf16) bf16 NHWC layout of the pooled map; scale/shift = the input
 * BatchNorm2d(3) or NULL; w packed [ky][kx][8][24] as for rtpose_stem_conv3x3_s2. */
int rtpose_stem_pool_nchw(const float* x_nchw, const float* scale, const float* shift, const float* w,
                          const float* bias, void* out, const rtpose_layout* lout, int cout, int N,
                          int H, int W, int out_bf16, void* stream);
int rtpose_stem_conv3x3_s2_nchw_ex(const float* x_nchw, const float* scale, const float* shift,
                                   const float* w, const float* bias, void* out,
                                   const rtpose_layout* lout, int cout, int N, int H, int W,
                                   int relu, int out_bf16, void* stream);
/* bf16 forms (bf16 plans: 2-byte activations, fp32 arithmetic, fp32 weights; C % 8 == 0) */
int rtpose_maxpool3x3s2_ceil_bf16(const void* in, const rtpose_layout* lin, void* out,
                                  const rtpose_layout* lout, int C, int N, int H, int W, void* stream);
int rtpose_dwconv3x3_bf16(const void* in, const rtpose_layout* lin, const float* w, const float* bias,
                          void* out, const rtpose_layout* lout, int C, int N, int H, int W, int stride,
                          void* stream);
int rtpose_layout_copy_cmap_bf16(const void* src, const rtpose_layout* lsrc, void* dst,
                                 const rtpose_layout* ldst, int C, const int32_t* cmap, int N, int H,
                                 int W, void* stream);
/* MaxPool2d(3, 2, 0, ceil_mode=True) (:98) */
int rtpose_maxpool3x3s2_ceil(const float* in, const rtpose_layout* lin, float* out,
                             const rtpose_layout* lout, int C, int N, int H, int W, void* stream);
/* depthwise 3x3, pad 1, stride 1 or 2 (+bias, no activation) (:35-38, :48-50);
 * w [9][C] fp32 (tap-major), C % 4 == 0, input layout gap >= 1. */
int rtpose_dwconv3x3(const float* in, const rtpose_layout* lin, const float* w, const float* bias,
                     float* out, const rtpose_layout* lout, int C, int N, int H, int W, int stride,
                     void* stream);
/* dst[.., cmap[c]] = src[.., c] for c < C (cmap: device int32[C], absolute dst channel) */
int rtpose_layout_copy_cmap(const float* src, const rtpose_layout* lsrc, float* dst,
                            const rtpose_layout* ldst, int C, const int32_t* cmap, int N, int H,
                            int W, void* stream);

/* dst(n,y,x,c) = alpha * dst(n,y,x,c) + beta * src[n][y][x][c] over a layout
 * slice (src dense NHWC).  Not on the reference's path: bench.py and the tests
 * use it to superimpose rasterised synthetic scenes on the outputs of the
 * randomly initialised network (no trained weights exist offline), so that the
 * decoder sees realistic peak counts while still consuming what the net wrote. */
int rtpose_layout_axpby(float* dst, const rtpose_layout* ldst, const float* src_nhwc,
                        int C, int N, int H, int W, float alpha, float beta,
                        void* stream);

/* ------------------------------------------------------------------------
 * 3. The rtpose_vgg network (lib/network/rtpose_vgg.py:60-225)
 *
 *   get_model('vgg19')       -> rtpose_net_create + rtpose_net_load_conv x92
 *   rtpose_model.forward     -> rtpose_net_forward   (rtpose_vgg.py:158-198)
 *
 * Conv index order == state_dict order of the reference module:
 *   model0.{0,2,5,7,10,12,14,16,19,21,23,25}, model1_1.{0,2,4,6,8},
 *   model2_1.{0,2,4,6,8,10,12} ... model6_1, model1_2 ..., model6_2
 *   (rtpose_vgg.py:141-155 registration order).
 * ---------------------------------------------------------------------- */
typedef struct rtpose_net rtpose_net;

int rtpose_net_create(int N, int H, int W, rtpose_net** out);
/* dtype: arithmetic of the plan.  RTPOSE_DTYPE_BF16 = bf16 activations/weights with fp32
 * accumulation (H, W multiples of 8); inputs, stage-output records and the final PAF /
 * heat-map stay fp32 at the API.  Weight arenas of the two dtypes are NOT interchangeable. */
#define RTPOSE_DTYPE_F32 0
#define RTPOSE_DTYPE_BF16 1
#define RTPOSE_DTYPE_BF16X3 2 /* split bf16 operands, 3 MFMAs per product: fp32-grade results */
int rtpose_net_create_ex(int N, int H, int W, int dtype, rtpose_net** out);
/* Per-plan choice of the arithmetic of the fp32 convs (the Winograd forms sum fewer, transformed
 * products: results differ from the direct sum by rounding, bounds in DESIGN.md §3.0).  The weight
 * arena of fp32 plans holds every packing a plan may choose (direct, F(2x2,3x3), F(4,7), F(6,7)), so
 * plans with different options share one arena and the choice costs nothing at run time.
 *   winograd3: RTPOSE_WINO_DEFAULT (= RTPOSE_WINO3_AUTO since round 4; the environment's RTPOSE_WINOGRAD=0|7 ->
 *              direct, RTPOSE_WINOGRAD3_M=2|4 -> F(2x2,3x3) | F(4x4,3x3) forced), 0 = direct 3x3 kernels,
 *              1 = F(2x2,3x3), 4 = F(4x4,3x3) forced (layers without an F(4x4,3x3) instance: F(2x2,3x3)),
 *              RTPOSE_WINO3_AUTO = per layer F(4x4,3x3) if its amplification estimate is <= amp_limit, else
 *              F(2x2,3x3); decided by rtpose_net_finalize_weights
 *   winograd7: RTPOSE_WINO_DEFAULT (= RTPOSE_WINO7_AUTO since round 4; the environment's RTPOSE_WINOGRAD=0|3 ->
 *              direct, RTPOSE_WINOGRAD7_M=4|6 -> F(4,7) | F(6,7) forced), 0 = direct, 4 = F(4,7), 6 = F(6,7)
 *              forced, RTPOSE_WINO7_AUTO = per layer the fastest form whose amplification estimate
 *              (rtpose_winograd_amplification of the loaded filters) is <= amp_limit: F(6,7), else F(4,7),
 *              else direct; decided by rtpose_net_finalize_weights
 *   The default is the guarded one because the forms' error bounds scale with the estimate and nobody can
 *   vouch for filters that have not been seen (pose_model.pth, README.md:19): i.i.d. Gaussian / He filters
 *   estimate 115-120 in F(6,7) and 42-43 in F(4x4,3x3) and keep the fast forms; +-1 edge filters (273) do not.
 *   amp_limit: the AUTO modes only; <= 0 = the library default (256: twice what i.i.d. Gaussian
 *              filters give in F(6,7))
 * Fields are ignored by bf16 / bf16x3 plans.  A form that has no kernel instance at the plan's
 * geometry falls back to the next one (F(6,7) -> F(4,7) -> direct) whatever the options say. */
#define RTPOSE_WINO_DEFAULT (-1)
#define RTPOSE_WINO7_AUTO 1
#define RTPOSE_WINO3_AUTO 3
typedef struct rtpose_net_options {
  uint32_t struct_bytes; /* sizeof(rtpose_net_options) of the caller */
  int32_t dtype;         /* RTPOSE_DTYPE_*                            */
  int32_t winograd3;
  int32_t winograd7;
  float amp_limit;
} rtpose_net_options;
int rtpose_net_create_opts(int N, int H, int W, const rtpose_net_options* opt, rtpose_net** out);
int rtpose_net_dtype(const rtpose_net* net);
void rtpose_net_destroy(rtpose_net* net);
size_t rtpose_net_workspace_bytes(const rtpose_net* net);
size_t rtpose_net_weight_bytes(const rtpose_net* net);
/* Hand in the two device arenas.  The workspace is zero-filled on `stream`
 * (the gaps of every activation layout must be zero); the weight arena may be
 * shared by nets of different N/H/W (its packing is shape independent). */
int rtpose_net_bind(rtpose_net* net, void* workspace, size_t workspace_bytes,
                    void* weights, size_t weight_bytes, int zero_workspace,
                    void* stream);
int rtpose_net_num_convs(const rtpose_net* net);
/* name: e.g. "model0.0" (state_dict prefix); returns 0 or RTPOSE_E_INVAL */
int rtpose_net_conv_info(const rtpose_net* net, int idx, char* name,
                         int name_cap, int* cout, int* cin, int* k);
/* Pack one conv's OIHW weight + bias (device pointers) into the weight arena */
int rtpose_net_load_conv(rtpose_net* net, int idx, const float* w_oihw,
                         const float* bias, void* stream);
/* After the last rtpose_net_load_conv: fixes the form of every conv of an RTPOSE_WINO7_AUTO / RTPOSE_WINO3_AUTO plan from
 * the amplification estimates of the filters just loaded (synchronises `stream` once to read them).
 * Optional for the other modes and called by the next forward if the host did not.  Plans that share one
 * weight arena may be loaded through any one of them: every fp32 rtpose_net_load_conv advances the arena's
 * generation, and each plan re-reads the estimates and re-decides its forms (here, in rtpose_net_forward* and
 * in rtpose_net_conv_numerics) when the generation it decided at is no longer the arena's. */
int rtpose_net_finalize_weights(rtpose_net* net, void* stream);
/* Arithmetic of conv idx in this plan: *form = 0 direct, 3 = F(2x2,3x3), 43 = F(4x4,3x3), 4 = F(4,7), 6 = F(6,7);
 * amp[4] = amplification estimates of the loaded filters in F(2x2,3x3) / F(4,7) / F(6,7) / F(4x4,3x3) (0 where
 * the form does not apply; synchronises `stream` if they have not been read back yet).  Either may be NULL. */
int rtpose_net_conv_numerics(rtpose_net* net, int idx, int* form, float* amp, void* stream);
/* Device-side error word of the plan (synchronises `stream`): bit 0 = a split-tile hand-over of a
 * persistent 7x7 launch timed out (see rtpose_conv2d_winograd_ex); 0 = none.  The word is cleared. */
int rtpose_net_device_status(rtpose_net* net, int* error_word, void* stream);
/* The persistent 7x7 launches (>= one tile per CU, not whole rounds) split tiles between neighbouring blocks and hand
 * the partial sums over through device flags - valid on a device that dispatches a 1-D grid in order with the blocks of
 * a round resident together, i.e. an exclusive, unmasked MI355X.  enable = 0 makes every 7x7 launch of the plan run one
 * block per tile instead: the same results, bit for bit (the sums run in the order of an unsplit tile either way), a
 * few per cent slower at batch 32 - for CU-masked or shared devices.  Default 1, or 0 when RTPOSE_W7_PERSIST=0 is in the
 * environment; rtpose_net_device_status switches it off by itself when it reads a timed-out hand-over (captured launch
 * lists are dropped with it); a caller of rtpose_net_device_status_async that later finds bit 0 in its host word calls
 * rtpose_net_set_persistent7(net, 0) itself (pipeline.py does). */
int rtpose_net_set_persistent7(rtpose_net* net, int enable);
/* A consumer that reads the stage-6 maps where the net wrote them (rtpose_net_output_view) on ANOTHER stream - the pose
 * decoder of batch k under the forward of batch k + 1 - hands in the HIP event it records behind its last read: every
 * later forward of the plan waits for that event (hipStreamWaitEvent on the forward's stream) in front of its first launch
 * that writes the maps' buffer, and for nothing else (fp32: conv4_4_CPM, which writes the out1 channels of the same concat
 * buffer - the reader runs beside the trunk; bf16 / bf16x3: the last launch).  RTPOSE_GUARD_WHOLE_FORWARD=1 in the
 * environment moves the wait in FRONT of the launch list (the reader never beside this plan's kernels; 1 % slower).
 * The guard stays installed until it is replaced or removed: NULL = no guard.  The event must have been recorded and must
 * outlive the guard.  (History, DESIGN.md 3.3: a decoder built with packed-fp32 VALU instructions returned wrong limb scores
 * beside the bf16 plan's kernels; the library's decoder is built without them.) */
int rtpose_net_set_output_guard(rtpose_net* net, void* hip_event);
/* Index (rtpose_net_launch_info numbering) of the launch a guarded forward waits in front of - decided, like the guard
 * itself, the first time either function is called on the plan (the environment is read then); host-only. */
int rtpose_net_output_guard_launch(rtpose_net* net);
int rtpose_net_persistent7(const rtpose_net* net);
/* The same without the wait: queues the copy of the error word into *host_word (pinned host memory, or the copy
 * is not asynchronous) and its clearing on `stream` and returns; the word is valid once the caller has waited for
 * anything it queued on `stream` afterwards (an event behind the D2H of a batch's records: the host that
 * pipelines batches never synchronises the whole stream for it). */
int rtpose_net_device_status_async(rtpose_net* net, int* host_word, void* stream);
/* 1 when forwards of this plan replay a captured hipGraph (RTPOSE_GRAPH=1 in the environment and the capture
 * succeeded), else 0. */
int rtpose_net_graph_active(const rtpose_net* net);
/* Enqueue the whole forward on `stream`: x is dense NCHW fp32 [N,3,H,W]. */
int rtpose_net_forward(rtpose_net* net, const float* x_nchw, void* stream);
/* The net's own NHWC8 input buffer, for producers that write it directly
 * (rtpose_preprocess_u8), and the forward that then skips the NCHW conversion.
 * The view holds what the PRODUCER wrote: rtpose_net_forward of an fp32 plan reads the NCHW
 * image directly (conv1_1, csrc/conv_first.hip) and does not refresh it. */
int rtpose_net_input_view(const rtpose_net* net, float** base, rtpose_layout* layout);
int rtpose_net_forward_prepared(rtpose_net* net, void* stream);
/* Copy stage output `which` (0..11 = saved_for_loss order: out1_1, out1_2,
 * ... out6_1, out6_2; rtpose_vgg.py:166-196) as dense NCHW.  Only the last
 * two survive a forward unless keep_intermediates was set. */
int rtpose_net_set_keep_intermediates(rtpose_net* net, int keep);
int rtpose_net_read_output(rtpose_net* net, int which, float* dst_nchw,
                           void* stream);
/* In-place view of the final PAF (which=0, 38 ch) / heat-map (which=1, 19 ch)
 * so the post-processing reads them where the last conv wrote them. */
int rtpose_net_output_view(const rtpose_net* net, int which, const float** base,
                           rtpose_layout* layout, int* C, int* H, int* W);
/* Per-layer HIP-event timing of the next forwards (bench.py roofline leg). */
int rtpose_net_set_profiling(rtpose_net* net, int enable);
int rtpose_net_num_launches(const rtpose_net* net);
/* After the stream has been synchronised: milliseconds of launch i of the
 * LAST forward, its conv kernel size (0 = not a conv), and algorithmic flops. */
int rtpose_net_launch_info(rtpose_net* net, int i, float* ms, int* k,
                           double* flops, char* name, int name_cap);
/* Matrix-core flops launch i actually ISSUES (padded channels / columns included; the Winograd
 * forms issue 16/36 resp. 70/196 of the direct sum's), and whether it runs in Winograd form:
 * algorithmic flops / time can exceed the MFMA peak, executed flops / time cannot. */
int rtpose_net_launch_executed_flops(const rtpose_net* net, int i, double* flops,
                                     int* winograd);

/* ------------------------------------------------------------------------
 * 3b. The ShuffleNetV2 x1.0 pose network (lib/network/rtpose_shufflenetV2.py:80-148,
 *     BASELINE config 4).  Same ownership rules as rtpose_net.  The host folds eval-mode
 *     BatchNorm into each conv and loads layers in rtpose_shufflenet_layer_info order
 *     (names are the reference state_dict prefixes, e.g. "network.3.0.conv0.1").
 * ---------------------------------------------------------------------- */
typedef struct rtpose_shufflenet rtpose_shufflenet;
int rtpose_shufflenet_create(int N, int H, int W, rtpose_shufflenet** out);
/* dtype: RTPOSE_DTYPE_F32 or RTPOSE_DTYPE_BF16 (bf16 activations + pointwise weights, fp32
 * accumulate; depthwise / stem weights and all biases stay fp32; outputs fp32) */
int rtpose_shufflenet_create_ex(int N, int H, int W, int dtype, rtpose_shufflenet** out);
void rtpose_shufflenet_destroy(rtpose_shufflenet* net);
size_t rtpose_shufflenet_workspace_bytes(const rtpose_shufflenet* net);
size_t rtpose_shufflenet_weight_bytes(const rtpose_shufflenet* net);
int rtpose_shufflenet_bind(rtpose_shufflenet* net, void* workspace, size_t workspace_bytes,
                           void* weights, size_t weight_bytes, int zero_workspace, void* stream);
int rtpose_shufflenet_num_layers(const rtpose_shufflenet* net);
/* kind: 0 = input affine (w = scale[3], b = shift[3]), 1 = stem conv [24,3,3,3],
 * 2 = depthwise [C,1,3,3], 3 = pointwise [cout,cin,1,1] */
int rtpose_shufflenet_layer_info(const rtpose_shufflenet* net, int idx, char* name, int name_cap,
                                 int* kind, int* cout, int* cin);
int rtpose_shufflenet_load(rtpose_shufflenet* net, int idx, const float* w, const float* b,
                           void* stream);
int rtpose_shufflenet_forward(rtpose_shufflenet* net, const float* x_nchw, void* stream);
/* which: 0 = PAF (38), 1 = heat-map (19) */
int rtpose_shufflenet_read_output(rtpose_shufflenet* net, int which, float* dst_nchw, void* stream);
int rtpose_shufflenet_output_view(const rtpose_shufflenet* net, int which, const float** base,
                                  rtpose_layout* layout, int* C, int* H, int* W);
int rtpose_shufflenet_set_profiling(rtpose_shufflenet* net, int enable);
int rtpose_shufflenet_num_launches(const rtpose_shufflenet* net);
int rtpose_shufflenet_launch_info(rtpose_shufflenet* net, int i, float* ms, double* flops, char* name,
                                  int name_cap);

/* ------------------------------------------------------------------------
 * 4. Batched pose decoding on the device
 *    stands in for lib/utils/paf_to_pose.py:372-406 (paf_to_pose_cpp):
 *      NMS            paf_to_pose.py:67-145   (find_peaks :25-38)
 *      process_paf    lib/pafprocess/pafprocess.cpp:22-194
 * ---------------------------------------------------------------------- */
/* Fused image prep on the device (caller side of get_outputs, evaluate/coco_eval.py:87-94):
 * uint8 BGR HWC image (device) -> cv2.resize(fx=fy=im_scale, INTER_LINEAR, 11-bit fixed point) ->
 * zero pad to Hn x Wn -> rtpose (mode 0) / vgg (mode 1) normalisation -> image n_index of an
 * NHWC8 layout buffer (e.g. rtpose_net_input_view).  hr x wr = cvRound(h0*s) x cvRound(w0*s). */
int rtpose_preprocess_u8(const unsigned char* img_bgr, int h0, int w0, double im_scale, int mode,
                         float* dst, const rtpose_layout* ldst, int n_index, int Hn, int Wn, int hr,
                         int wr, void* stream);

/* A whole bucket of images in ONE launch: `images` is a HOST array of `count` descriptors (they
 * travel as kernel arguments, 64 per launch; nothing is allocated or copied), every image has its
 * own source size, scale, valid size and destination slot; all share the padded size Hn x Wn of
 * the destination (the evaluation driver buckets images by it, evaluate/coco_eval.py:258-272). */
typedef struct rtpose_prep_image {
  const void* img_bgr; /* device, uint8 [h0][w0][3]                               */
  double im_scale;     /* crop_with_factor's im_scale (im_transform.py:124)       */
  int32_t h0, w0;      /* source size                                             */
  int32_t hr, wr;      /* cvRound(h0 * im_scale), cvRound(w0 * im_scale)          */
  int32_t flip;        /* != 0: x-mirrored inside the valid width (flip TTA pass) */
  int32_t n_index;     /* image slot of the destination layout                    */
} rtpose_prep_image;
int rtpose_preprocess_u8_batch(const rtpose_prep_image* images, int count, int mode, float* dst,
                               const rtpose_layout* ldst, int Hn, int Wn, void* stream);

/* Same, with flip != 0 writing the x-mirrored RESIZED image (columns [0, wr) mirrored, the
 * zero padding stays on the right): the second pass of flip test-time augmentation. */
int rtpose_preprocess_u8_flip(const unsigned char* img_bgr, int h0, int w0, double im_scale, int mode,
                              float* dst, const rtpose_layout* ldst, int n_index, int Hn, int Wn, int hr,
                              int wr, int flip, void* stream);

/* Multi-scale test-time augmentation (BASELINE config 3; the scale set is NOT pinned by
 * the reference tree - SURVEY.md §3.2): dst = beta*dst + alpha*bilinear_resize(src), dense
 * NHWC, half-pixel centres, edge clamp.  Only the top-left src_h_valid x src_w_valid
 * (fractional) region of src is mapped onto dst, so the zero padding of a scaled input
 * (im_transform.py:128-132) is cropped away on the fly. */
int rtpose_resize_bilinear_accum(const float* src, int hs, int ws, float* dst, int hd, int wd,
                                 int C, int N, float src_h_valid, float src_w_valid,
                                 float alpha, float beta, void* stream);

/* One scale of batched multi-scale (+flip) TTA, fused: heat/paf are the network's own output
 * views (rtpose_net_output_view) of a batch of 2B images - [0,B) normal, [B,2B) mirrored
 * (B images if flip == 0).  Forms the handle_paf_and_heat average (coco_eval.py:197-242;
 * mirror inside the first w_valid columns) and accumulates
 *   acc = beta * acc + alpha * bilinear_resize(average)   (dense [B,hd,wd,19] / [B,hd,wd,38])
 * exactly as rtpose_flip_merge followed by rtpose_resize_bilinear_accum would. */
int rtpose_tta_accumulate(const float* heat, const rtpose_layout* lheat, const float* paf,
                          const rtpose_layout* lpaf, int B, int hs, int w_valid, float* acc_heat,
                          float* acc_paf, int hd, int wd, float src_h_valid, float src_w_valid,
                          float alpha, float beta, int flip, void* stream);

#define RTPOSE_NUM_PART 18
#define RTPOSE_NUM_LIMB 19

typedef struct rtpose_decode_cfg {
  int32_t num_keypoints;  /* cfg.MODEL.NUM_KEYPOINTS (18)   default.py:40  */
  int32_t upsample;       /* cfg.MODEL.DOWNSAMPLE   (8)     default.py:41  */
  float thresh_heatmap;   /* cfg.TEST.THRESH_HEATMAP (0.1)  default.py:126 */
  int32_t max_peaks_per_part; /* device table capacity per (image, part)   */
  int32_t max_humans;         /* device table capacity per image           */
} rtpose_decode_cfg;

/* One decoded peak: paf_to_pose.py:141-142 row (x, y, score, id). */
typedef struct rtpose_peak {
  int32_t x, y; /* refined, at upsampled (input) resolution                */
  float score;
  int32_t id;   /* running counter over parts then peaks (pafprocess cid)  */
} rtpose_peak;

/* Bytes of the caller-provided device scratch + result block for N images. */
size_t rtpose_decode_workspace_bytes(const rtpose_decode_cfg* cfg, int N);
/* Bytes of the compact result record block (device or host copy of it). */
size_t rtpose_decode_result_bytes(const rtpose_decode_cfg* cfg, int N);

/* Enqueue NMS + refine + PAF scoring + assignment + grouping for N images.
 * heat: 19-channel (>= num_keypoints used) map, paf: 38-channel map, any
 * layout (dense HWC = lead 0, ws = w, hs = h, cstride = C).
 * `result` (device, rtpose_decode_result_bytes) receives, per image:
 *   int32 header[8]: n_peaks, n_humans, overflow_flags, max_peaks_per_part, max_humans (the
 *                    capacities the record is laid out for: a block describes itself), 0...
 *   int32 part_count[18]
 *   rtpose_peak peaks[18 * max_peaks_per_part]   (grouped by part, cid order)
 *   int32 human_parts[max_humans][18]            (cid, -1 = absent)
 *   float human_score[max_humans]                (get_score, cpp:204-206)
 */
int rtpose_decode_batch(const float* heat, const rtpose_layout* lheat,
                        const float* paf, const rtpose_layout* lpaf, int N,
                        int h, int w, const rtpose_decode_cfg* cfg,
                        void* workspace, size_t workspace_bytes, void* result,
                        void* stream);

/* Stage-wise entry points (same kernels, for tests and the legacy API).      */
/* NMS only (paf_to_pose.py:67-145): fills part_count + peaks of `result`.    */
int rtpose_nms_batch(const float* heat, const rtpose_layout* lheat, int N,
                     int h, int w, const rtpose_decode_cfg* cfg, void* result,
                     void* stream);

/* The two optional branches of NMS (lib/utils/paf_to_pose.py:67; neither is enabled by
 * the reference's own callers): RTPOSE_NMS_GAUSSIAN = bool_gaussian_filt=True — the x8
 * patch is smoothed with scipy.ndimage.gaussian_filter(sigma=3) before the arg-max
 * (:121-122); RTPOSE_NMS_NO_REFINE = bool_refine_center=False — no patch, the peak is
 * the cell centre (c + 0.5) * up - 0.5 with the low-resolution map value as score
 * (:135-139); rtpose_peak.x / .y then hold that coordinate truncated to int, which is
 * what process_paf makes of the joint_list column (pafprocess.cpp:28-29). */
#define RTPOSE_NMS_NO_REFINE 1
#define RTPOSE_NMS_GAUSSIAN 2
int rtpose_nms_batch_ex(const float* heat, const rtpose_layout* lheat, int N,
                        int h, int w, const rtpose_decode_cfg* cfg, int nms_flags,
                        void* result, void* stream);
int rtpose_decode_batch_ex(const float* heat, const rtpose_layout* lheat,
                           const float* paf, const rtpose_layout* lpaf, int N,
                           int h, int w, const rtpose_decode_cfg* cfg,
                           int nms_flags, void* workspace,
                           size_t workspace_bytes, void* result, void* stream);
/* The 25 normalised float64 weights RTPOSE_NMS_GAUSSIAN correlates with
 * (scipy _gaussian_kernel1d(sigma=3, radius=12)); returns 25, or < 0. */
int rtpose_gaussian_kernel1d(double* weights, int cap);

/* ------------------------------------------------------------------------
 * 5. Flip test-time-augmentation merge
 *    stands in for evaluate/coco_eval.py:197-242 (handle_paf_and_heat).
 *    All four inputs dense HWC fp32 on the device, outputs likewise.
 * ---------------------------------------------------------------------- */
int rtpose_flip_merge(const float* heat, const float* heat_flipped,
                      const float* paf, const float* paf_flipped, int N, int h,
                      int w, float* heat_avg, float* paf_avg, void* stream);

/* ------------------------------------------------------------------------
 * 6. Legacy single-image API — same seven names and argument meaning as the
 *    SWIG module (lib/pafprocess/pafprocess.h:53-59, pafprocess.i:14-15).
 *    Host pointers in, results kept in process-global state until the next
 *    process_paf (pafprocess.cpp:12-13), guarded by a mutex.  The scoring /
 *    assignment / grouping run on the GPU; getters are bounds-checked and
 *    return -1 / NaN instead of the reference's undefined behaviour.
 * ---------------------------------------------------------------------- */
int process_paf(int p1, int p2, int p3, float* peaks, int h1, int h2, int h3,
                float* heatmap, int f1, int f2, int f3, float* pafmap);
int get_num_humans(void);
int get_part_cid(int human_id, int part_id);
float get_score(int human_id);
int get_part_x(int cid);
int get_part_y(int cid);
float get_part_score(int cid);

#ifdef __cplusplus
}
#endif
#endif /* RTPOSE_MI355X_H */
