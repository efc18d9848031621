/*
 * faceoff_hip.h -- C ABI of libfaceoff_hip.so: the MI355X (gfx950) kernels behind the FaceOff
 * VQ-VAE-2 + Conv3d-latent training step.
 *
 * The reference (skymanaditya1/FaceOff) has no FFI: its hot path is torch.nn modules
 * (SURVEY.md section 8b).  Each entry point below therefore replaces the library kernel that a
 * reference call site dispatches to; the call site is cited as file:line relative to the
 * reference root.  The Python mirror of the reference module API (faceoff_amd/models/...) binds
 * these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - Plain C: pointers, ints, a stream handle.  No C++/torch types.
 *  - All pointers are DEVICE pointers (fp32 unless noted).  The caller owns every buffer,
 *    including workspaces; functions only enqueue work on `stream` (hipStream_t passed as
 *    void*), never allocate, never synchronise: they are graph-capturable.
 *  - Activations are channels-last: [N,H,W,C] (frames outermost; a clip batch [B,T,H,W,C] is the
 *    same memory with N=B*T).  Every activation argument carries a row stride `ld*` in floats
 *    (>= its channel count, multiple of 4) so a tensor may be a channel slice of a wider
 *    buffer: torch.cat along channels (vqvae_conv3d_latent.py:271,282) costs nothing.
 *  - Filters are passed PACKED (see fo_pack_*), converted from the checkpoint layouts
 *    (OIHW / IOHW / OIDHW) each step.
 *  - Return value: 0 on success, negative FO_E_* otherwise; fo_last_error() gives a
 *    thread-local description.
 */
#ifndef FACEOFF_HIP_H
#define FACEOFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_OK 0
#define FO_E_SHAPE (-1)  /* unsupported geometry / channel count */
#define FO_E_ALIGN (-2)  /* pointer or stride not 16-byte aligned */
#define FO_E_ARCH (-3)   /* not running on gfx950 */
#define FO_E_HIP (-4)    /* a HIP runtime call failed */
#define FO_E_WORKSPACE (-5)

/* epilogue / prologue flags of the conv kernels (value = ((acc + bias) masked) + add, relu'd) */
#define FO_IN_RELU 1   /* relu() applied to the input operand as it is staged (ResBlock's leading ReLU, :91) */
#define FO_BIAS 2      /* + bias[co] */
#define FO_MASK 4      /* zero where mask[pixel][co] <= 0  (ReLU backward fused into a dgrad) */
#define FO_ADD 8       /* + add[pixel][co]  (residual `out += input` :99, or gradient fan-in) */
#define FO_OUT_RELU 16 /* relu() on the result (nn.ReLU after a conv, :110,112,119,126,145,151,186) */
#define FO_DEPTH2SPACE 32 /* fused k4 s2 p1 ConvTranspose2d with few output channels (dec.blocks.6, 64 -> 6, :152): one
                             3x3 conv over the INPUT grid whose 32 GEMM columns are 4 sub-pixel phases x 8 channels
                             (fo_pack_convT_k4s2_fused); column ph*8+c of input pixel (y,x) lands at output pixel
                             (2y+ph/2, 2x+ph%2), channel c.  desc: Cout=32, Hout=2*Hm, ldOut>=8, ophW = real channels; ophH = 1: the cell form
                             (fo_pack_convT_k4s2_cells), Hout = 2*(Hm-1).  fo_conv_bf16 takes the cell form at any width: Cout = 4 Cpp GEMM
                             columns, Cpp (a multiple of 8) per phase, ophW <= Cpp real channels, FO_MASK / FO_ADD at the OUTPUT pixel
                             (fo_pack_convT_k4s2_cells_n) */
#define FO_OUT_F32 64  /* fo_conv_bf16 only: the result is stored as fp32 (ldOut in floats) instead of being rounded to bf16 -- the
                          quantisers' inputs (quantize_conv_t / _b, :208,213: VQ distances and arg-min stay fp32) and the decoder output */

/* ABI version: bumped whenever an exported signature or the meaning of an argument changes (101: the loss kernels take a caller-owned
   partial-sum workspace `ws` in front of `stream` and OVERWRITE their scalars -- round 4; 102: fo_comm_broadcast_async added; fo_instnorm_lrelu_{fwd,bwd}_batch take a workspace, fo_lpips_tap_fwd_bwd_unpool_bf16, fo_selftest_lane_moves, fo_pack_convT_k4s2_cells_n, fo_comm_set_stream added -- round 6).  The Python binding refuses a library whose
   fo_version() differs from the FO_ABI_VERSION it was written against (faceoff_amd/_lib.py), so an older .so handed in through
   FACEOFF_HIP_LIB is a clean error and not a stream pointer read as a workspace. */
#define FO_ABI_VERSION 102
int fo_version(void);
const char* fo_last_error(void);
/* Kernel notes (measurement plumbing, off by default).  fo_kernel_notes(1): from now on every entry point records the symbol of the MAIN kernel
 * it launches -- name and template arguments as rocprofv3 --kernel-trace prints them, e.g. "conv_bf16_pp16_kernel<256, 256, 2, 4>" -- in a
 * thread-local; returns the previous setting.  fo_last_kernel(): the symbol recorded by the calling thread's last entry point ("" if none),
 * cleared by the read.  bench.py keys its per-kernel timings by these names, so `roofline.kernel` can be looked up in profiles/. */
int fo_kernel_notes(int enable);
const char* fo_last_kernel(void);
/* device properties the host needs for the roofline report: [0]=CU count, [1]=clock kHz, [2]=is gfx950 */
int fo_device_info(int32_t* out3);
/* diagnostic: bad64[lane] = 0 where the library's vector-ALU lane moves (DPP / v_permlane swaps in place of ds_bpermute shuffles) return what
   __shfl_xor returns for every distance 1 .. 32 and for the 8- and 64-lane butterfly sums (tests/test_ops_gpu.py) */
int fo_selftest_lane_moves(int32_t* bad64, void* stream);

/* ---------------------------------------------------------------- layout transforms */
/* [N,C,H,W] -> [N,H,W,ldy] (channels >= C zero-filled up to Cpad).  Replaces the implicit NCHW
 * layout of utils.py:32 `torch.cat([source, background], axis=2)` feeding Conv2d. */
int fo_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, int ldy, void* stream);
/* process_data (utils.py:29-38): img = cat([source, background], channels) fused with the layout change:
 * y[N,H,W,8] = channels of a [N,Ca,H,W], then of b [N,Cb,H,W], then zeros (Ca + Cb <= 8). */
int fo_nchw2_to_nhwc8(const float* a, int Ca, const float* b, int Cb, float* y, int N, int H, int W, void* stream);
/* [N,H,W,ldx] (first C channels) -> [N,C,H,W]; optionally accumulates (+=) for gradient returns. */
int fo_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, int accumulate, void* stream);

/* ---------------------------------------------------------------- filter packing */
/* All packers write K-contiguous rows: wp[o][tap][i] so both GEMM operands stream along K.
 * Opad/Ipad: channel counts rounded up by the caller (fo_packed_elems tells the size); padding is zero. */
/* Conv forward.  src [O][I][taps] (nn.Conv2d / nn.Conv3d weight) -> wp[Opad][taps][Ipad]. */
int fo_pack_conv(const float* w, float* wp, int O, int I, int taps, int Opad, int Ipad, void* stream);
/* Stride-1 conv dgrad: src [O][I][taps] -> wp[Ipad][taps (reversed)][Opad]: the conv of the output
 * gradient with the flipped, transposed filter (autograd of F.conv2d/F.conv3d, reference :100). */
int fo_pack_conv_dgrad(const float* w, float* wp, int O, int I, int taps, int Opad, int Ipad, void* stream);
/* k4 s2 p1 transposed conv as 4 sub-pixel phases.  src [Ci][Co][4][4] (nn.ConvTranspose2d weight,
 * :150,152,160,215) -> wp[4 phases][Copad][2x2 taps][Cipad].  Also packs the dgrad of a k4 s2 p1
 * Conv2d when handed that conv's OIHW weight (Ci:=O, Co:=I). */
int fo_pack_convT_k4s2(const float* w, float* wp, int Ci, int Co, int Cipad, int Copad, void* stream);
/* Same weight, Co <= 8, as ONE 3x3 filter bank for FO_DEPTH2SPACE: wp[32 = 4 phases x 8][9 taps][Cipad], zero where
 * a tap of the 3x3 input neighbourhood does not contribute to a phase (5 of 9 per phase). */
int fo_pack_convT_k4s2_fused(const float* w, float* wp, int Ci, int Co, int Cipad, void* stream);
/* The cell form of the same layer (FO_DEPTH2SPACE with desc.ophH = 1): a k2 p1 conv over an (Hin+1) x (Win+1) grid of 2x2-pixel
 * cells, wp[32 = 4 phases x 8][4 taps][Cipad]; cell (i,j), phase (a,b) lands at output pixel (2i+a-1, 2j+b-1) (the border cells'
 * outside pixels are dropped).  K = 4 Ci instead of 9 Ci. */
int fo_pack_convT_k4s2_cells(const float* w, float* wp, int Ci, int Co, int Cipad, void* stream);
/* ... with Cpp >= Co columns per phase (a multiple of 8): wp[4 Cpp][4 taps][Cipad] -- the cell form of ANY k4 s2 p1 transposed convolution, and of
 * the data gradient of a k4 s2 p1 convolution (dec.blocks.4, dec_t.blocks.4, upsample_t :147-151,222; the gradients of enc_b.blocks.2 / enc_t.blocks.0
 * :104-126 under loss.backward()) as ONE fo_conv_bf16 launch with FO_DEPTH2SPACE (desc.Cout = 4 Cpp) instead of four sub-pixel phase launches that each
 * read the whole input (round 6). */
int fo_pack_convT_k4s2_cells_n(const float* w, float* wp, int Ci, int Co, int Cpp, int Cipad, void* stream);

/* ---------------------------------------------------------------- convolution (implicit GEMM, fp32 MFMA) */
typedef struct fo_conv_desc {
  int32_t N, T;            /* frames; frames per clip (temporal taps never cross a clip). 2-D: T=1 */
  int32_t Hin, Win;        /* input spatial size */
  int32_t Hm, Wm;          /* GEMM-M grid (output pixels of this launch, per frame) */
  int32_t Hout, Wout;      /* spatial size of the output tensor */
  int32_t Cin, Cout;       /* packed (padded) input channels; real output channels (stores are clipped) */
  int32_t KD, KH, KW;      /* tap grid */
  int32_t stride;          /* input coord = m*stride + k - pad */
  int32_t padD, padH, padW;
  int32_t ostride, ophH, ophW; /* output coord = m*ostride + oph (sub-pixel phase of a transposed conv) */
  int32_t ldIn, ldOut, ldMask, ldAdd;
  int32_t flags;
} fo_conv_desc;

/* out[opix(m)][co] = epilogue( sum_{tap,ci} in[ipix(m,tap)][ci] * wp[co][tap][ci] ).
 * Replaces cuDNN implicit-GEMM fwd/dgrad behind nn.Conv2d (:92,94,109,111,113,118,120,140,208,213),
 * nn.ConvTranspose2d (:150,152,160,215), nn.Conv3d (:181,185) and their autograd dgrads. */
int fo_conv_igemm(const fo_conv_desc* d, const float* in, const float* wp, const float* bias,
                  const float* mask, const float* add, float* out, void* stream);

/* ResBlock.forward in ONE launch (models/vqvae_conv3d_latent.py:86-101: ReLU -> Conv 3x3 C->32 -> ReLU -> Conv 1x1 32->C ->
 * `out += input`): hbuf = relu(conv3x3(relu(x)) + b1) [N,H,W,ldOut of d] (kept for the backward), out = conv1x1(hbuf) + b3 + x,
 * ReLU'd when out_relu != 0 (the encoder's / decoder's trailing nn.ReLU, :126,145).  The 32-channel tile goes from the 3x3
 * conv's accumulators through LDS into the 1x1 contraction without a round trip to memory.  d = the 3x3 conv (Cin = C = 128,
 * Cout = 32, same-size grid; ldIn = ldAdd = pixel stride of x); wp1 / wp3 = fo_pack_conv of the two filters. */
int fo_resblock_fwd(const fo_conv_desc* d, const float* x, const float* wp1, const float* b1, const float* wp3, const float* b3,
                    float* hbuf, float* out, int ldOut2, int out_relu, void* stream);

/* Backward through a ResBlock's second convolution (reference models/vqvae_conv3d_latent.py:94-95,99: ReLU -> Conv2d(32, 128, 1); replaces the
 * autograd nodes of that conv and of the ReLU in front of it) in one pass over the block's output gradient g [M][ldG] (128 channels) and the
 * hidden activation h [M][ldH] (32 channels, post-ReLU, as fo_resblock_fwd left it):
 *   gh [M][ldGh] = (g W3) * (h > 0),   dw3 [128][32] = g^T h,   db3 [128] (may be NULL) = column sums of g.
 * wp3 = fo_pack_conv of the 1x1 filter.  ws: fo_resblock_bwd_conv3_ws_bytes(M) bytes.  Deterministic (fixed summation order). */
int64_t fo_resblock_bwd_conv3_ws_bytes(int64_t M);
int fo_resblock_bwd_conv3(int64_t M, const float* g, int ldG, const float* h, int ldH, const float* wp3, float* gh, int ldGh, float* dw3,
                          float* db3, float* ws, int64_t ws_bytes, void* stream);

/* fo_conv_igemm with the filter chosen per frame: frames [b*bank_frames, (b+1)*bank_frames) use the b-th of the
 * consecutive packed filter banks behind `wp`.  No bias / mask / residual.  Needs bank_frames*Hm*Wm % 128 == 0. */
int fo_conv_igemm_banked(const fo_conv_desc* d, const float* in, const float* wp, float* out, int bank_frames, void* stream);

/* The same plane-stack GEMM as a persistent kernel of its own (csrc/wino_gemm.hip), used when a plane is whole 128-row
 * tiles, Cin >= 64 and Cout % 128 == 0 (every C2 shape):
 *   M[xi][r][co] = sum_{kd,ci} V[xi][r + (kd - KD/2)*P][ci] * U[xi][co][kd][ci]      r < N*P rows per plane, P rows per frame,
 * depth taps outside a clip of T frames skipped.  V [planes][N*P][Cin], U [planes][Cout][KD][Cin], M [planes][N*P][Cout]. */
int fo_wino_gemm(const float* V, const float* U, float* M, int planes, int N, int T, int P, int Cin, int Cout, int KD, void* stream);
/* The same GEMM with every fp32 product formed on the bf16 matrix pipe (csrc/wino_gemm_split.hip): each operand is split
 * exactly into three bf16 pieces (V in registers, U by a small kernel into the scratch U3) and the six partial products of
 * weight >= 2^-16 are accumulated in fp32 -- relative error 2^-23 per product, i.e. fp32 arithmetic to the last bit, at 6/16
 * of the fp32 MFMA's matrix time.  Same arguments, constraints and layouts (all fp32 in memory) plus the scratch:
 * fo_wino_gemm_split_ws_bytes(planes, Cin, Cout, KD) bytes, 16-byte aligned, private to the stream until the GEMM ends. */
int64_t fo_wino_gemm_split_ws_bytes(int planes, int Cin, int Cout, int KD);
int fo_wino_gemm_split(const float* V, const float* U, void* U3, float* M, int planes, int N, int T, int P, int Cin, int Cout, int KD,
                       void* stream);

/* fo_conv_wgrad for `banks` independent planes of N/banks frames (whole clips each) in one launch: dw receives `banks`
 * consecutive [Areal][Breal][taps] tensors.  Conv3d geometry (KD > 1) only; no bias sum. */
int64_t fo_wgrad_banked_ws_bytes(const fo_conv_desc* d, int banks);
int fo_conv_wgrad_banked(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal, float* ws,
                         int64_t ws_bytes, int banks, void* stream);

/* Which kernel instantiation fo_conv_igemm launches for `d` (no launch): 128 / 64 / 32 = conv_igemm_kernel<BN>,
 * 3 = conv_igemm3_kernel (Conv3d, one workgroup per CU).  For profilers that attribute time per kernel. */
int fo_conv_igemm_variant(const fo_conv_desc* d);

/* Filter gradient:  dW[a][b][tap] = sum_m P[m][a] * Q[qpix(m,tap)][b]   (checkpoint layout out).
 * Conv:  P = grad_out (a=Cout), Q = input (b=Cin)  -> OIHW / OIDHW.
 * ConvT: P = input (a=Cin), Q = grad_out (b=Cout)  -> [Ci][Co][kh][kw].
 * Geometry fields: Hm/Wm = P's spatial grid, Hin/Win = Q's, Cin := channels of Q (b), Cout := channels
 * of P (a), ldIn := ld of Q, ldOut := ld of P; stride/pad/K* as for the forward conv.  FO_IN_RELU
 * applies relu to Q, FO_MASK is unused.  Split-K partial slabs live in `ws` (fo_wgrad_ws_bytes);
 * `dw` is overwritten; if `dbias` != NULL it receives sum_m P[m][a] (conv bias grad).
 * Replaces cuDNN wgrad behind loss.backward() (train_faceoff_perceptual.py:100). */
int64_t fo_wgrad_ws_bytes(const fo_conv_desc* d);
int fo_conv_wgrad(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal,
                  float* dbias, float* ws, int64_t ws_bytes, void* stream);
/* dbias[c] = sum over M rows of g[m*ld + c] for c < Creal (C = Creal rounded up to 4); ws >= 4*C*1024 bytes.
 * Bias gradient of a transposed conv (its grad_out is the wgrad's Q operand, so it cannot ride along). */
int fo_bias_grad(const float* g, float* dbias, int64_t M, int C, int Creal, int ld, float* ws, void* stream);

/* ---------------------------------------------------------------- vector quantiser (Quantize.forward :47-80) */
/* x[Nvec][ldx] (dim 64) vs embed[64][512].  Writes ind (int64, :54) and q_ste = x + (embed[:,ind] - x)
 * (:57,78); *sq_sum = sum((q-x)^2) (:77; OVERWRITTEN).  Distances use the reference's
 * expanded form ||x||^2 - 2 x.e + ||e||^2 in fp32 with k-ordered fma chains and first-index arg-min
 * (torch.max tie-break): restated bit-for-bit by oracle/vq_oracle.c.  `enorm`[512] from fo_vq_prepare.
 * ws: fo_vq_assign_ws_bytes() of scratch -- the commitment sum meets there as one partial per workgroup, added in workgroup
 * order by a one-wave finish launch (no float atomics: the latent loss is the same bits run after run). */
int fo_vq_prepare(const float* embed, float* embedT, float* enorm, void* stream); /* [64][512] -> [512][64], ||e||^2 */
int64_t fo_vq_assign_ws_bytes(void);
int fo_vq_assign(const float* x, int ldx, int64_t nvec, const float* embedT, const float* enorm, int64_t* ind,
                 float* q_ste, int ldq, float* sq_sum, float* ws, void* stream);
/* EMA statistics of the assignment (:60-61, replaces F.one_hot + the second sgemm): counts[512] and
 * esum[512][64] (code-major) are OVERWRITTEN.  ws: fo_vq_stats_ws_bytes(nvec).  No atomics: a wave owns the codes
 * c % 16 == wave and adds their vectors in vector order, workgroup slabs are summed in a fixed order -- bit-reproducible. */
int64_t fo_vq_stats_ws_bytes(int64_t nvec);
int fo_vq_stats(const float* x, int ldx, int64_t nvec, const int64_t* ind, float* counts, float* esum, float* ws,
                void* stream);
/* EMA codebook update (:66-75) after the statistics all-reduce: in-place on the three buffers. */
int fo_vq_ema(float* embed, float* cluster_size, float* embed_avg, const float* counts, const float* esum,
              float decay, float alpha /* = 1 - decay as the reference rounds it */, float eps, void* stream);
/* gx = gq + gdiff * 2 (x - q) / numel   (straight-through + commitment) */
int fo_vq_bwd(const float* gq, int ldg, const float* x, int ldx, const float* q, int ldq, const float* gdiff,
              float scale, float* gx, int ldgx, int64_t nvec, void* stream);
/* F.embedding(code, embed^T) (:82-83, decode_code :287-295) */
int fo_vq_gather(const int64_t* ind, const float* embedT, float* q, int ldq, int64_t nvec, void* stream);

/* ---------------------------------------------------------------- losses */
/* *sum = sum over n,c<3,h,w of (dec[n][h][w][c] - gt[n][c][h][w])^2 (OVERWRITTEN)
 * (criterion = nn.MSELoss() on out[:, :3], train_faceoff_perceptual.py:21,37-39).
 * Loss scalars never meet in float atomics: a kernel leaves one partial per workgroup in `ws` (FO_LOSS_WS_BYTES of scratch) and a one-wave
 * finish launch adds them in a fixed order -- recon / latent / perceptual / GAN losses are the same bits run after run. */
#define FO_LOSS_WS_BYTES 16384
int fo_mse_slice_fwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3, float* sum, float* ws, void* stream);
/* gdec[n][h][w][c] = c<3 ? gscale * 2 (dec-gt)/numel : 0, for c < ldg  (gscale read from device) */
int fo_mse_slice_bwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3,
                     const float* gscale, float inv_numel, float* gdec, int ldg, void* stream);
/* both in one pass over dec and gt (the training step, :37-39 then :100) */
int fo_mse_slice_fwd_bwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3, const float* gscale,
                         float inv_numel, float* gdec, int ldg, float* sum, float* ws, void* stream);

/* ---------------------------------------------------------------- LPIPS / VGG-16 (models/lpips.py:80-161, loss.py:27-33)
 * The 13 VGG convolutions (+ReLU) are fo_conv_igemm launches (first layer: Cin 3 padded to 8, KW padded
 * 3->4 with zero taps so K is a multiple of 32); these are the kernels between them. */
/* ScalingLayer (:96-103) + layout: y[N,H,W,8] = (x - shift)/scale on channels 0..2, zeros on 3..7.
 * src: NCHW [N,3,H,W] (src_is_nhwc=0) or channels-last with pixel stride ld (decoder output).
 * shift3 / scale3 are HOST pointers to 3 floats. */
int fo_lpips_prep(const float* src, int src_is_nhwc, int ld, float* y, int N, int H, int W, const float* shift3,
                  const float* scale3, void* stream);
/* gdec[p][c] += weight * gscale[0] * g[p][c] / scale_c for c < 3 (backward of the scaling into the decoder-output grad) */
int fo_lpips_prep_bwd(const float* g, int ldg, float* gdec, int ldd, int64_t npix, const float* scale3, const float* gscale,
                      float weight, void* stream);
/* nn.MaxPool2d(2,2) of torchvision vgg16.features[4,9,16,23] (lpips.py:118-134), dense [N,H,W,C]. */
int fo_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* gx = relu'(x) * ([x is the first maximum of its 2x2 window] * gy + add); add may be NULL (a LPIPS tap gradient). */
int fo_maxpool2_bwd(const float* x, const float* gy, const float* add, float* gx, int N, int H, int W, int C, void* stream);
/* One LPIPS tap (:85-89,155-161): val[n] += mean_hw sum_c lin_c (f0/(|f0|+eps) - f1/(|f1|+eps))^2, C in {64,128,256,512}. */
int64_t fo_lpips_tap_ws_bytes(int N, int H, int W);   /* scratch of fo_lpips_tap_fwd: per-pixel values, summed per frame in pixel order (no atomics) */
int fo_lpips_tap_fwd(const float* f0, const float* f1, const float* lin, float* val, int N, int H, int W, int C, float* ws, void* stream);
/* Gradient of mean_n(sum of taps) wrt f1 (the reconstruction branch), times gscale[0], through f1's own ReLU. */
int fo_lpips_tap_bwd(const float* f0, const float* f1, const float* lin, const float* gscale, float* gf1, int N, int H, int W,
                     int C, void* stream);

/* ---------------------------------------------------------------- Winograd F(m x m, 3x3), m = 2 or 4, for the Conv3d k3 p1 layers (:181,185)
 * out = A^T [ sum_{kd,ci} (G g_kd G^T) . (B^T d B) ] A: the two spatial dimensions are transformed ((m+2)^2 multiplies
 * per m x m outputs instead of 9 m^2: 2.25x / 4x fewer), depth taps and channels stay a contraction = a (3,1,1) Conv3d
 * over the stack of (m+2)^2 transformed planes, run by fo_conv_igemm_banked (or one fo_conv_igemm per plane).
 * fp32 error: m = 2 as the direct convolution, m = 4 about 10x that (3e-6 of the tensor's scale). */
/* U[(m+2)^2][Opad][KD][Ipad] from the checkpoint filter w[O][I][KD][3][3]; dgrad=1: the data-gradient filter (flipped
 * taps, channel roles swapped: rows = I, K columns = O). */
int fo_wino_filter(const float* w, float* U, int O, int I, int KD, int Opad, int Ipad, int dgrad, int m, void* stream);
/* V[(m+2)^2][N][H/m][W/m][C] = B^T d B of the zero-padded (m+2)x(m+2) patches of x [N,H,W,ldx]. */
int fo_wino_input(const float* x, int ldx, float* V, int N, int H, int W, int C, int m, void* stream);
/* out [N,H,W,ldOut] = epilogue(A^T M A), M[(m+2)^2][N][H/m][W/m][C]; flags: FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU. */
int fo_wino_output(const float* M, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd, float* out,
                   int ldOut, int N, int H, int W, int C, int flags, int m, void* stream);

/* Filter gradient of the same convolution in the transformed domain: dM[(m+2)^2][N][H/m][W/m][C] = A dY A^T of the output
 * gradient g [N,H,W,ldg]; then dU[xi] = sum_pixels dM[xi] (x) V[xi] shifted by the depth tap -- (m+2)^2 wgrad GEMMs in one
 * fo_conv_wgrad_banked launch ((3,1,1) geometry, planes as banks); then dW[O][I][KD][3][3] = G^T dU G. */
int fo_wino_gradout(const float* g, int ldg, float* dM, int N, int H, int W, int C, int m, void* stream);
/* The same transform with the layer's bias gradient riding along: dbias[c] = sum over the pixels of g[.][c], c < C (the transform reads every
 * pixel of g exactly once; a separate column-sum pass would stream g again).  C / 4 must divide 256.  ws: fo_wino_gradout_bias_ws_bytes. */
int64_t fo_wino_gradout_bias_ws_bytes(int N, int H, int W, int C, int m);
int fo_wino_gradout_bias(const float* g, int ldg, float* dM, int N, int H, int W, int C, int m, float* dbias, float* ws, int64_t ws_bytes,
                         void* stream);
/* The dU GEMMs above with the fp32 products on the bf16 matrix pipe (csrc/wino_wgrad_split.hip; the arithmetic of
 * fo_wino_gemm_split): dU[xi][co][ci][kd] = sum_r dM[xi][r][co] * V[xi][r + (kd - KD/2) * P][ci] over the N * P rows of each of
 * `planes` planes, frames in clips of T.  dM [planes][N*P][Cout], V [planes][N*P][Cin] (dense), dU [planes][Cout][Cin][KD].
 * Cin % 128 == 0, Cout % 128 == 0, N*P % 32 == 0 (P % 32 == 0 when KD == 3).  ws: fo_wino_wgrad_split_ws_bytes(...) bytes. */
int64_t fo_wino_wgrad_split_ws_bytes(int planes, int N, int P, int Cin, int Cout, int KD);
int fo_wino_wgrad_split(const float* dM, const float* V, float* dU, float* ws, int64_t ws_bytes, int planes, int N, int T, int P, int Cin,
                        int Cout, int KD, void* stream);
int fo_wino_wgrad_out(const float* dU /* [(m+2)^2][O][I][KD] */, float* dW, int O, int I, int KD, int m, void* stream);

/* ---------------------------------------------------------------- Winograd F(4x4, 2x2) for the k4 s2 p1 stems (csrc/wino42.hip)
 * A Conv2d k4 s2 p1 (reference models/vqvae_conv3d_latent.py:108,110,117) is a k2 s1 convolution over 2x2 pixel cells of the
 * padded input; its adjoint -- its data gradient, and the forward of ConvTranspose2d k4 s2 p1 (:157,160,215) -- a full k2
 * correlation that produces cells.  Both run as 25 Winograd-domain GEMMs (fo_wino_gemm, KD = 1) between these transforms.
 * w: [O][I][4][4] (Conv2d) or, same memory, [I_T][O_T][4][4] (ConvTranspose2d with O := I_T, I := O_T).
 *   conv form        rows = tiles of 4x4 outputs, K = 4 I:  V = input_cells(x),  U = filter(transposed = 0) [25][O][4 I],
 *                    M [25][rows][O] -> fo_w42_output -> y [N][H/2][W/2][O]
 *   transposed form  rows = tiles of 4x4 cells over the (h+1) x (w+1) cell grid, K = O:  V = input_full(g),
 *                    U = filter(transposed = 1) [25][4 I][O],  M [25][rows][4 I] -> fo_w42_output_cells -> [N][2h][2w][I]
 *   filter gradient  dM = fo_w42_gradout(dy) [25][rows][O], V = input_cells(x): dU [25][O][4 I] by fo_conv_wgrad_banked,
 *                    dW = fo_w42_wgrad_out(dU)
 * planeRows: rows per plane of V / M / dM (>= the tile count; the GEMM wants a multiple of 128).  Epilogue flags as
 * fo_wino_output (FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU). */
int fo_w42_filter(const float* w, float* U, int O, int I, int transposed, void* stream);
int fo_w42_input_cells(const float* x, int ldx, float* V, int N, int H, int W, int C, long long planeRows, void* stream);
int fo_w42_input_full(const float* g, int ldg, float* V, int N, int h, int w, int C, long long planeRows, void* stream);
int fo_w42_output(const float* M, long long planeRows, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd,
                  float* out, int ldOut, int N, int h, int w, int C, int flags, void* stream);
int fo_w42_output_cells(const float* M, long long planeRows, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd,
                        float* out, int ldOut, int N, int h, int w, int C, int flags, void* stream);
int fo_w42_gradout(const float* g, int ldg, float* dM, int N, int h, int w, int C, long long planeRows, void* stream);
/* ... with the layer's bias gradient (dbias[c] = column sums of g) riding along, as fo_wino_gradout_bias. */
int64_t fo_w42_gradout_bias_ws_bytes(int N, int h, int w, int C);
int fo_w42_gradout_bias(const float* g, int ldg, float* dM, int N, int h, int w, int C, long long planeRows, float* dbias, float* ws,
                        int64_t ws_bytes, void* stream);
int fo_w42_wgrad_out(const float* dU /* [25][O][4 I] */, float* dW, int O, int I, void* stream);

/* ---------------------------------------------------------------- input pipeline / validation helpers on the device
 * dst[n][c] = warp of src[n][c] ([N][C][H][W] frames) by the affine map M (6 floats, HOST pointer, row-major 2x3) taking
 * DESTINATION pixel (x, y) to its SOURCE position -- what cv2.warpAffine applies after inverting its argument
 * (TemporalAlignment/perturbations.py:45-83) -- zero outside the image; mode 0 = bilinear, 1 = bicubic (A = -0.75, the
 * INTER_CUBIC kernel of cv2.resize, :88).  Exact arithmetic: OpenCV's 1/32-pixel coordinate quantisation is not reproduced. */
int fo_affine_warp(const float* src, float* dst, int N, int C, int H, int W, const float* M_dst_to_src, int mode, void* stream);
/* The same perturbations on 8-bit frames [N][H][W][C] (C <= 4, the layout cv2 hands the reference), with OpenCV 4.6.0's own
 * fixed-point arithmetic (opencv-python==4.6.0.66, environment.yml:70; csrc/warp_u8.hip restates imgwarp.cpp / resize.cpp):
 *   fo_warp_affine_u8    cv2.warpAffine(image, M, (w, h)) -- perturbations.py:51 :63 :80 :117.  M_fwd: HOST pointer to the 2x3
 *                        FORWARD map(s) as doubles ([6], or [N][6] with per_frame != 0), inverted here as cv::warpAffine does;
 *                        1/32-pixel source coordinates, 15-bit bilinear weights, BORDER_CONSTANT 0.
 *   fo_resize_center_u8  resize_image (:87-105): cv2.resize(fx = fy = m, INTER_CUBIC) followed by the centre crop (m >= 1) or the
 *                        centre paste onto zeros (m < 1), one launch.  magnification: HOST pointer, [1] or [N].
 *   fo_flip_u8           cv2.flip(image, flip_code) (:125): 0 rows reversed, > 0 columns reversed, < 0 both.
 *   fo_u8_to_norm_nchw   transforms.ToTensor() + transforms.Normalize((mean,)*C, (std,)*C) (TemporalAlignment/dataset.py:244-256):
 *                        dst[n][c][y][x] = (float(src[n][y][x][c']) / 255 - mean) / std, c' = C-1-c with reverse_channels (BGR -> RGB). */
int fo_warp_affine_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, const double* M_fwd, int per_frame, void* stream);
int fo_resize_center_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, const double* magnification, int per_frame, void* stream);
int fo_flip_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, int flip_code, void* stream);
int fo_u8_to_norm_nchw(const uint8_t* src, float* dst, int N, int H, int W, int C, int reverse_channels, float mean, float stdv, void* stream);
/* out[n][y][x][3] uint8 = (uint8)(255 * (clamp(v, -1, 1) + 1) / 2) of channels c0..c0+2: `denormalize` + the uint8 cast of
 * the validation videos (train_faceoff_perceptual.py:71-77, utils.py:9-17).  src: [N][C][H][W] (ld = 0) or channels-last with
 * pixel stride ld (C unused).  bgr != 0 swaps to BGR (cv2.cvtColor(..., COLOR_RGB2BGR)). */
int fo_denorm_u8(const float* src, int ld, int C, int c0, uint8_t* out, int N, int H, int W, int bgr, void* stream);

/* ---------------------------------------------------------------- MoCoGAN-HD discriminators (BASELINE config 5)
 * Replaces the cuDNN kernels behind ModelD_3d / ModelD_img (TemporalAlignment/models/mocoganhd_video_disc.py:8-176,
 * mocoganhd_content_disc.py:8-165): Conv3d / Conv2d k4 s2|s1 p2 forward, data gradient, filter gradient;
 * InstanceNorm (affine=False, track_running_stats=True) + LeakyReLU(0.2); AvgPool(3, count_include_pad=False);
 * the relativistic average LSGAN loss (mocoganhd_losses.py:108-126); the frame pairing of the trainer
 * (disc_trainers/train_vqvae_mocoganhd_disc.py:364-365,395-396).  Activations are channels-last [N][D][H][W][ld]
 * (a 2-D tensor has D = 1). */
#define FO_OUT_LRELU 64   /* fo_convnd: LeakyReLU(slope) on the result */
#define FO_MASK_LRELU 128 /* fo_convnd: result *= (mask[pixel][c] > 0 ? 1 : slope)  (LeakyReLU backward fused into a data gradient) */
#define FO_KSPLIT 256     /* fo_convnd: the launch MAY cut the contraction into slices (few tiles behind a long K: the 256 -> 512 layers),
                             added in slice order by a second launch through the workspace of fo_convnd_ws_bytes */
typedef struct fo_convnd_desc {
  int32_t N;
  int32_t Ds, Hs, Ws, Cs, ldS; /* SOURCE tensor of the launch (forward: the conv's input; transposed: the output gradient) */
  int32_t Dd, Hd, Wd, Cd, ldD; /* DESTINATION tensor (forward: the conv's output; transposed: the input gradient) */
  int32_t KD, KH, KW, sD, sH, sW, pD, pH, pW; /* the CONVOLUTION's kernel / stride / padding in either direction */
  int32_t ldMask;              /* pixel stride of `mask` (FO_MASK_LRELU), on the destination grid */
  int32_t flags;               /* FO_BIAS | FO_OUT_LRELU | FO_MASK_LRELU | FO_ADD (accumulate into dst) | FO_KSPLIT */
  float slope;                 /* LeakyReLU negative slope */
} fo_convnd_desc;
/* w[O][I][taps] (checkpoint OIDHW / OIHW) -> forward pack [O pad 64][taps][I pad 32] (transposed = 0) or the data-gradient
 * pack [I pad 64][taps][O pad 32] (transposed = 1). */
int fo_pack_convnd(const float* w, float* wp, int O, int I, int taps, int transposed, void* stream);
/* transposed = 0: dst = conv(src) (+ bias, LeakyReLU).  transposed = 1: dst = data gradient of that conv for the output
 * gradient src (Cs = the conv's Cout padded to 32 with zero channels, Cd = the conv's Cin), as a gather (no atomics).
 * Source channels must be a multiple of 32. */
/* With FO_KSPLIT in d->flags a launch of few tiles behind a long contraction slices K over workgroups: the slices leave
 * partial tiles in ws (>= fo_convnd_ws_bytes(d, transposed); 0 = this shape is not sliced) and a second launch adds them in slice
 * order and applies the epilogue -- no atomics, bit-reproducible.  Taps that only padding can reach from a tile are skipped. */
int64_t fo_convnd_ws_bytes(const fo_convnd_desc* d, int transposed);
int fo_convnd(const fo_convnd_desc* d, int transposed, const float* src, const float* wp, const float* bias, const float* mask,
              float* dst, float* ws, int64_t ws_bytes, void* stream);
/* dw[Cd][CsReal][taps] = sum over output positions of g (x) src for the FORWARD conv d (g on the destination grid).
 * Rows (and, for a depth tap, only the frames whose input is not padding) are the contraction index; when the layer needs
 * fo_wgradnd_splits(d) > 1 row slices to fill the chip, the slices leave partial sums in ws (>= fo_wgradnd_ws_bytes(d), 0
 * otherwise) and a second launch adds them in slice order: no atomics, bit-reproducible. */
int fo_wgradnd_splits(const fo_convnd_desc* d);
int64_t fo_wgradnd_ws_bytes(const fo_convnd_desc* d);
int fo_wgradnd(const fo_convnd_desc* d, const float* g, const float* src, float* dw, int CsReal, float* ws, int64_t ws_bytes, void* stream);
/* InstanceNorm(affine=False) over the `rows` positions of one sample, per channel, then LeakyReLU:
 *   y = lrelu((x - mean_c) * rstd_c), biased variance, eps inside the sqrt.  stats = [mean(C) | rstd(C)] (kept for backward);
 * running (may be NULL) = [running_mean(C) | running_var(C)], updated with `momentum` and the UNBIASED variance, as
 * nn.InstanceNorm*d(track_running_stats=True) does in training mode.  use_running != 0 (eval mode): normalise with them. */
int fo_instnorm_lrelu_fwd(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, float eps, float slope, float* stats,
                          float* running, float momentum, int use_running, void* stream);
/* gx = d loss / d x given gy = d loss / d y and the saved y, stats (training-mode statistics). */
int fo_instnorm_lrelu_bwd(const float* gy, int ldg, const float* y, int ldy, const float* stats, float* gx, int ldgx, int64_t rows,
                          int C, float slope, void* stream);
/* The same for N samples stored one after the other ([N][rows][ld]) in ONE call; stats [N][2C].  Training mode moves the
 * running statistics once per sample in the order given by `order` (device int32[N]: the reference calls the module sample by
 * sample).  With a workspace of fo_instnorm_ws_bytes(N, rows, C) > 0 bytes (C / 4 a power of two in 8 .. 256: the discriminators' 128, 256 and
 * 512 channels) the training-mode statistics are made in three chip-filling launches -- per-chunk (mean, M2), a fixed-order merge, an
 * elementwise pass: bit-reproducible, no atomics -- instead of one workgroup per 8 channels; ws == NULL (or eval mode) runs the latter. */
int64_t fo_instnorm_ws_bytes(int N, int64_t rows, int C);
int fo_instnorm_lrelu_fwd_batch(const float* x, int ldx, float* y, int ldy, int N, int64_t rows, int C, float eps, float slope,
                                float* stats, float* running, const int32_t* order, float momentum, int use_running, float* ws, int64_t ws_bytes,
                                void* stream);
int fo_instnorm_lrelu_bwd_batch(const float* gy, int ldg, const float* y, int ldy, const float* stats, float* gx, int ldgx, int N,
                                int64_t rows, int C, float slope, float* ws, int64_t ws_bytes, void* stream);
/* AvgPool(k = 3 in every pooled dimension, padding 1, count_include_pad = False) of [D][H][W][C] with strides (sD, sH, sW);
 * kD = 1 leaves the depth axis alone (2-D pooling).  _bwd ADDS the gradient into gx (zero it first). */
int fo_avgpool3_fwd(const float* x, float* y, int D, int H, int W, int C, int ld, int kD, int sD, int sH, int sW, void* stream);
int fo_avgpool3_bwd(const float* gy, float* gx, int D, int H, int W, int C, int ld, int kD, int sD, int sH, int sW, void* stream);
/* Discriminator inputs: out[j][h][w][0:3] = frame f0, [3:6] = frame (first + j*step) of a window of frames, zero up to ldOut
 * channels (train_vqvae_mocoganhd_disc.py:364-365: `cat((x[:,0] repeated, x[:,1:]), dim=2)`; step = -1 with first = F-1 is
 * flip_video :169-174; n = 1 is the image discriminator's pair :354-357).  src: nchw != 0 -> [F][3][H][W] (ground truth),
 * else channels-last [F][H][W][ldSrc] (decoder output).  _bwd adds the pair gradient into gsrc [F][H][W][ldG] channels 0..2
 * times `scale`. */
int fo_disc_pairs(const float* src, int nchw, int ldSrc, int H, int W, int f0, int first, int step, int n, float* out, int ldOut,
                  void* stream);
int fo_disc_pairs_bwd(const float* gout, int ldOut, int H, int W, int f0, int first, int step, int n, float* gsrc, int ldG, float scale,
                      void* stream);
/* The 6-channel first layer of a discriminator (Conv k4 s2 p2, :133-136) as a k2 s1 p1 conv over the space-to-depth image:
 * xs[n][d'][h'][w'][((bd*2+bh)*2+bw)*C + c] = x[n][2d'+bd][2h'+bh][2w'+bw][c] (zero past an odd axis' end / in the channel padding;
 * depth_too = 0: 2-D, phases (bh, bw) only).  inverse != 0 scatters xs back into x (the input gradient's way back).
 * fo_s2d_filter maps the checkpoint filter w[O][C][KD][4][4] to w2[O][phases*C][taps2] (tap k = 2a + b) and back (inverse: the
 * filter gradient's way back). */
int fo_space_to_depth2(float* x, int ldx, float* xs, int ldxs, int N, int D, int H, int W, int C, int depth_too, int inverse, void* stream);
int fo_s2d_filter(float* w, float* w2, int O, int C, int KD, int inverse, void* stream);
/* Relativistic average LSGAN (mocoganhd_losses.py:108-126) on one scale's patch logits a[na], b[nb] (pixel stride ld):
 *   loss = w * ( mean((a - mean(b) - ta)^2) + mean((b - mean(a) - tb)^2) )       added to *loss_acc (a plain read-modify-write by one
 *   thread: calls that share a loss_acc must be ordered on one stream);
 * ga / gb (either may be NULL) receive d loss / d a, d loss / d b times gscale[0] (stride ld, written not added). */
int fo_ralsgan(const float* a, int na, const float* b, int nb, int ld, float ta, float tb, float w, float* loss_acc,
               const float* gscale, float* ga, float* gb, void* stream);

/* ---------------------------------------------------------------- bf16 LPIPS branch (BASELINE config 3)
 * The same VGG-16 / LPIPS chain with bf16 storage and bf16 MFMA operands, fp32 accumulation and fp32 head
 * arithmetic; every stored tensor is rounded to bf16 once (round-to-nearest-even).  `void*` tensors below are
 * bf16, channels-last; bias / lin / val / gdec stay fp32.  Replaces the cuDNN convs torch.autocast(bfloat16)
 * would run under `VQLPIPS.forward` (loss.py:27-33, models/lpips.py:115-152). */
/* wp[o][t][i] = bf16(w[o][i][t]) for t < taps, zero for taps <= t < tapsPad and the channel padding. */
int fo_pack_conv_bf16(const float* w, void* wp, int O, int I, int taps, int Opad, int Ipad, int tapsPad, void* stream);
/* wp[i][t][o] = bf16(w[o][i][taps-1-t]): the stride-1 data-gradient filter. */
int fo_pack_conv_dgrad_bf16(const float* w, void* wp, int O, int I, int taps, int Opad, int Ipad, void* stream);
/* 2-D implicit-GEMM conv, bf16 in / bf16 out.  d->ld* are in ELEMENTS; Cin % 64 == 0, or Cin == 8 == ldIn (RGB padded
 * to a 16-byte pixel; filter packed with tapsPad = taps rounded up to 8).  Filters packed with Opad = Cout rounded up
 * to 128 (Cout > 64), 64 (Cout > 32) or 32.  flags: FO_BIAS | FO_MASK (mask = bf16 activation, > 0) | FO_OUT_RELU. */
int fo_conv_igemm_bf16(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, void* out,
                       void* stream);
/* fo_conv_igemm_bf16 that also writes the 2x2 max-pool of its result (VGG: conv1_2 -> MaxPool2d(2), reference models/lpips.py:118-123; the
 * full-resolution result stays a LPIPS tap): pooled [N][Hout/2][Wout/2][ldPooled], bit for bit fo_maxpool2_fwd_bf16 of `out`.  Only where the
 * 64-input-channel halo-tile kernel applies (3x3, frames of whole 4 x 32 tiles, >= 8 tiles per CU): FO_E_SHAPE otherwise -- pool separately. */
int fo_conv_igemm_bf16_pool(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, void* pooled, int ldPooled,
                            void* stream);
/* fo_conv_bf16 with optional extras.  ReLU masks as BIT PLANES: a data gradient needs the sign of the forward activation only, and reads 1/16 of its
 * bytes from a plane [pixel][Cout/8] bytes, bit c % 8 of byte c / 8 = (value[pixel][c] > 0) -- config 3's masked data gradients read 3.7 GB of
 * activations for their signs, 1.2 ms of its 36.7 ms step (round 4, measured with the mask reads switched off).
 *   pooled / ldPooled / pool_idx  as fo_conv_igemm_bf16_pool_idx (NULL: none);
 *   mask_bits    FO_MASK from a plane of the OUTPUT's geometry instead of the bf16 tensor `mask` (which may then be NULL);
 *   out_bits     receives the plane of this launch's bf16 result (not with FO_ADD / FO_OUT_F32 / FO_DEPTH2SPACE);
 *   pooled_bits  receives the plane of the pooled output [N][Hout/2][Wout/2][Cout/8].
 * Cout % 8 == 0.  Reference: the ReLUs of models/lpips.py:118-134 under loss.backward() (train_faceoff_perceptual.py:100). */
typedef struct fo_conv_extra {
  void* pooled;
  int32_t ldPooled;
  void* pool_idx;
  const void* mask_bits;
  void* out_bits;
  void* pooled_bits;
} fo_conv_extra;
int fo_conv_bf16_ex(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, const void* add, void* out,
                    const fo_conv_extra* ex, void* stream);
/* ... and the pool's arg-max codes (fo_maxpool2_fwd_idx_bf16's idx, [N][Hout/2][Wout/2][Cout/4] bytes; NULL: none). */
int fo_conv_igemm_bf16_pool_idx(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, void* pooled, int ldPooled,
                                void* idx, void* stream);
/* ---------------------------------------------------------------- bf16-operand VQ-VAE step (BASELINE config 3 as SURVEY 8(d) defines it)
 * The VQ-VAE's own convolutions with bf16 MFMA operands, fp32 accumulation, fp32 master weights and fp32 VQ: activations and
 * activation gradients are STORED as bf16 (rounded once, after bias / ReLU mask / residual or fan-in add / ReLU were applied in
 * fp32 to the accumulator), filters are rounded from the fp32 master copy every step, filter and bias gradients stay fp32.
 *
 * fo_conv_bf16 = fo_conv_igemm_bf16 with the whole fo_conv_desc honoured: depth taps (KD = 3, clips of d->T frames: reference
 * nn.Conv3d :181,185; the taps of a row tile that see nothing but clip padding are skipped), stride 2, sub-pixel phases
 * (ostride / oph*), Cin % 32 == 0 for the same-size stride-1 forms with Cout % 128 == 0 (else Cin % 64 == 0, or 8), and the flags
 * FO_IN_RELU (Cout <= 32 forms only: the operand is staged through registers), FO_BIAS, FO_MASK, FO_ADD (bf16 tensor, ldAdd),
 * FO_OUT_RELU, FO_DEPTH2SPACE (Cout = 4 x a multiple of 8 columns: see the flag) and FO_OUT_F32.  Replaces the cuDNN kernels torch.autocast(bfloat16) would pick for
 * models/vqvae_conv3d_latent.py:92-190 and their autograd data gradients. */
int fo_conv_bf16(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, const void* add, void* out,
                 void* stream);
/* Filter gradient of the same convolutions: dw[a][b][taps] (fp32, the checkpoint layout; a < Areal, b < Breal) =
 * sum_m P[m][a] * Q[qpix(m,tap)][b] with bf16 P (the tensor on the conv's output grid: d->Cout channels, row stride d->ldOut, grid
 * d->Hm x d->Wm) and bf16 Q (d->Cin channels, d->ldIn, d->Hin x d->Win), fp32 accumulation (csrc/wgrad_bf16.hip).  Same descriptor
 * convention as fo_conv_wgrad; FO_IN_RELU applies relu() to Q as it is staged.  Channel counts are multiples of 8; an 8-channel Q
 * (the image layers) takes a form of its own at the k4 s2 geometry.  dbias (may be NULL): [Areal] column sums of P, formed by the
 * same launch (one extra MFMA per K-step against a fragment of ones in the workgroups of the centre tap).  P and Q are read through
 * 32-bit buffer offsets: each must be smaller than 2 GiB (FO_E_SHAPE otherwise).  3x3 (x KD) pad-1 stride-1 layers whose rows are
 * multiples of 32 pixels run wgrad9_bf16_kernel (all nine taps of a depth plane per workgroup, LDS-DMA staging: both operands fetched
 * about once) when P has >= 128 and Q >= 64 channels, or P <= 32 and Q >= 128; everything else the row-run / gather forms.
 * ws: fo_wgrad_bf16_ws_bytes(d) bytes of scratch, private to the stream. */
int64_t fo_wgrad_bf16_ws_bytes(const fo_conv_desc* d);
int fo_conv_wgrad_bf16(const fo_conv_desc* d, const void* P, const void* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                       int64_t ws_bytes, void* stream);
/* db[c] = sum over the rows of the bf16 tensor g[rows][ld] (bias gradients), c < Creal <= C. */
int64_t fo_bias_grad_bf16_ws_bytes(int C);
int fo_bias_grad_bf16(const void* g, float* db, int64_t rows, int C, int Creal, int ld, float* ws, void* stream);
/* Row-strided conversions, C % 8 == 0 (strides in elements of the respective type). */
int fo_f32_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, void* stream);
int fo_bf16_to_f32(const void* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, void* stream);
/* fo_nchw2_to_nhwc8 with the result rounded to bf16 (the network input of the bf16-operand engine; utils.py:32). */
int fo_nchw2_to_nhwc8_bf16(const float* a, int Ca, const float* b, int Cb, void* y, int N, int H, int W, void* stream);
/* fo_vq_assign that also writes a bf16 copy of the straight-through output (q_bf16 [nvec][ldqb], may be NULL): the quantiser itself
 * -- distances, arg-min, gather, commitment sum -- runs in fp32 on the fp32 input exactly as fo_vq_assign.
 * forced_ind (may be NULL): teacher-forced codes -- the search is skipped, `ind` receives forced_ind, and the gather, straight-through
 * value and commitment sum use it (what Quantize.forward :57,77-78 would compute had :54 returned these indices; decode_code's
 * direction, :287-295, with the commitment term).  Used by the parity tests to compare a bf16-operand step with the oracle on the
 * oracle's own codes, where a near-tie cannot open an O(1) gap. */
int fo_vq_assign2(const float* x, int ldx, int64_t nvec, const float* embedT, const float* enorm, int64_t* ind, float* q_ste, int ldq,
                  float* sq_sum, void* q_bf16, int ldqb, const int64_t* forced_ind, float* ws, void* stream);
/* fo_vq_bwd with bf16 gradients: gx = bf16(gq + gdiff[0] * scale * (x - q)); gq, gx bf16, x and q fp32. */
int fo_vq_bwd_bf16(const void* gq, int ldg, const float* x, int ldx, const float* q, int ldq, const float* gdiff, float scale, void* gx,
                   int ldgx, int64_t nvec, void* stream);
int fo_lpips_prep_bf16(const float* src, int src_is_nhwc, int ld, void* y, int N, int H, int W, const float* shift3,
                       const float* scale3, void* stream);
int fo_lpips_prep_bwd_bf16(const void* g /* [npix][8] */, float* gdec, int ldd, int64_t npix, const float* scale3,
                           const float* gscale, float weight, void* stream);
int fo_maxpool2_fwd_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int fo_maxpool2_bwd_bf16(const void* x, const void* gy, const void* add, void* gx, int N, int H, int W, int C, void* stream);
/* The same pool with its arg-max recorded (reference nn.MaxPool2d keeps indices for its backward, models/lpips.py:118-134 via torchvision): idx holds
 * 2 bits per element of y -- which pixel of the 2x2 window, scan order, the FIRST maximum -- as bytes [N][H/2][W/2][C/4] (channel c in byte c / 4,
 * bits 2 (c % 4) and up).  fo_maxpool2_bwd_idx_bf16: gx[t] = [t is the recorded pixel] * gy + add[t] (add may be NULL), rounded once; NO ReLU mask is
 * applied here: add must be zero where x is (the tap heads' gradients are) and gy masked by y > 0 (fo_conv_igemm_bf16 with mask = y).  It reads 1/16 of
 * x's bytes where fo_maxpool2_bwd_bf16 reads x. */
int fo_maxpool2_fwd_idx_bf16(const void* x, void* y, void* idx, void* ybits /* NULL or the bit plane of y (fo_conv_bf16_ex) */, int N, int H, int W, int C,
                             void* stream);
int fo_maxpool2_bwd_idx_bf16(const void* idx, const void* gy, const void* add, void* gx, int N, int H, int W, int C, void* stream);
/* ws: fo_lpips_tap_ws_bytes_bf16(N, H, W, C) of scratch (both head entry points): every wave leaves its per-frame sums in its own slots and a
 * finish launch adds a frame's slots in wave order -- val[] is reproducible bit for bit (it met in float atomics before round 4). */
int64_t fo_lpips_tap_ws_bytes_bf16(int N, int H, int W, int C);
int fo_lpips_tap_fwd_bf16(const void* f0, const void* f1, const float* lin, float* val, int N, int H, int W, int C, float* ws, void* stream);
int fo_lpips_tap_bwd_bf16(const void* f0, const void* f1, const float* lin, const float* gscale, void* gf1, int N, int H, int W,
                          int C, void* stream);
/* VGG conv1_1 + ReLU + conv1_2 + ReLU (+ the 2x2 max-pool in front of conv2_1) in ONE launch (models/lpips.py:118-127, slice1 and the head of slice2): the
 * halo-tile kernel of the 64-channel layers computes its relu1_1 input patch from the scaled image instead of reading it back.  x8 [N][H][W][8] bf16
 * (fo_lpips_prep_bf16); wp1 = fo_pack_conv_bf16 of the [64][8][3][3] filter with tapsPad 16; wp2 = fo_pack_conv_bf16 of [64][64][3][3].
 * out1 = relu1_1 [N][H][W][64] (NULL: not stored -- the ground-truth branch), out2 = relu1_2 [N][H][W][64], pooled [N][H/2][W/2][64] or NULL.
 * FO_E_SHAPE unless H % 4 == 0, W % 32 == 0 and there are >= 4 tiles of 4 x 32 pixels per CU: the caller then runs the layers one by one. */
int fo_vgg_conv1_fused_bf16(const void* x8, const void* wp1, const float* b1, const void* wp2, const float* b2, void* out1, void* out2, void* pooled,
                            int N, int H, int W, void* stream);
/* fo_lpips_tap_fwd_bf16 and fo_lpips_tap_bwd_bf16 in ONE pass over the two feature maps (training: the tap's upstream gradient
 * gscale[0] / (N H W) is known before its value is): val[n] += the tap's value per frame, gf1 = its gradient wrt f1. */
int fo_lpips_tap_fwd_bwd_bf16(const void* f0, const void* f1, const float* lin, float* val, const float* gscale, void* gf1, int N, int H,
                              int W, int C, float* ws, void* stream);
/* The same for a tap that also feeds a 2x2 max-pool, launched in the BACKWARD: gf1 = head gradient + the pool's backward of gpool
 * ([N][H/2][W/2][C] bf16) by its recorded arg-max `codes` (fo_maxpool2_fwd_idx_bf16 / fo_conv_igemm_bf16_pool_idx: 2 bits per channel), summed in
 * fp32 and rounded once -- replaces fo_lpips_tap_fwd_bwd_bf16 + fo_maxpool2_bwd_idx_bf16 for that tap (LPIPS.forward lpips.py:80-93 under
 * loss.backward(), train_faceoff_perceptual.py:100). */
int fo_lpips_tap_fwd_bwd_unpool_bf16(const void* f0, const void* f1, const float* lin, float* val, const float* gscale, const void* gpool,
                                     const void* codes, void* gf1, int N, int H, int W, int C, float* ws, void* stream);

/* ---------------------------------------------------------------- optimiser + utilities */
/* torch.optim.Adam defaults (train_faceoff_perceptual.py:190) over one flat parameter arena. */
int fo_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, float bias_corr1 /* 1-beta1^t */, float bias_corr2 /* 1-beta2^t */, float grad_scale,
                 void* stream);
int fo_zero(float* p, int64_t n, void* stream);
/* y = relu(x) elementwise over [rows][C] views with strides (only for API paths that hand out activations) */
int fo_relu(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, void* stream);
/* y = a + b over strided [rows][C] views (gradient fan-in where no conv epilogue can take it) */
int fo_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t rows, int C, void* stream);

/* The discriminators' 1-channel patch head (Conv3d / Conv2d(512 | 256, 1, k 4, s 1, p 2); reference mocoganhd_video_disc.py:150-158) as dot
 * products instead of a 64-column GEMM tile: d describes the FORWARD convolution (Cs channels -> Cd == 1, stride 1, <= 64 taps);
 * wp = its fo_pack_convnd forward pack (row 0 is used).  fwd writes y[position][0] only (ldD floats per position); dgrad writes every channel
 * of gx; wgrad writes dw[0][c][tap] for c < CsReal.  Bit-reproducible (fixed summation orders). */
int fo_disc_head_fwd(const fo_convnd_desc* d, const float* x, const float* wp, const float* bias, float* y, void* stream);
int fo_disc_head_dgrad(const fo_convnd_desc* d, const float* g, const float* wp, float* gx, void* stream);
int64_t fo_disc_head_wgrad_ws_bytes(const fo_convnd_desc* d);
int fo_disc_head_wgrad(const fo_convnd_desc* d, const float* g, const float* x, float* dw, int CsReal, float* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- data-parallel gradient exchange (csrc/comm.cpp)
 * What nn.parallel.DistributedDataParallel's reducer does for the reference (train_faceoff_perceptual.py:164-169; process group:
 * distributed/launch.py:61-66) for a host that does not go through torch.distributed: one communicator per process (= per GPU) over RCCL
 * (opened with dlopen on first use).  Rank 0 calls fo_comm_unique_id and hands the 128 opaque bytes to every rank over the job's existing
 * rendezvous (the reference's TCP dist_url); every rank then calls fo_comm_init.  All-reduces are in-place fp32 SUMs (averaging is folded
 * into the optimiser's grad_scale) on the communicator's own stream, ordered by events: no call here blocks the host except destroy. */
typedef struct fo_comm fo_comm;
int fo_comm_unique_id(void* id128);
int fo_comm_init(fo_comm** out, int rank, int world, const void* id128, int device);
/* From now on the collectives run on `stream` (the caller's; it must outlive the communicator) instead of the stream fo_comm_init created.  Why: HIP gives a
   stream its hardware queue at first use (four queues by default), and on the compute stream's queue every all-reduce lines up behind the kernels it is meant
   to overlap; the Python side hands in a stream it has checked (faceoff_amd.distributed.comm.AbiComm.create). */
int fo_comm_set_stream(fo_comm* c, void* stream);
int fo_comm_rank(const fo_comm* c);
int fo_comm_world(const fo_comm* c);
int64_t fo_comm_issued(const fo_comm* c);   /* all-reduces enqueued so far */
/* buf[0..count) := sum over ranks, enqueued BEHIND everything enqueued so far on after_stream (the stream that produced buf) */
int fo_comm_allreduce_async(fo_comm* c, float* buf, int64_t count, void* after_stream);
/* buf[0..count) on every rank := rank root's buf, ordered like the all-reduce (DDP's broadcast_buffers for buffers that are not summed:
   the discriminators' InstanceNorm running statistics); not counted by fo_comm_issued */
int fo_comm_broadcast_async(fo_comm* c, float* buf, int64_t count, int root, void* after_stream);
/* `stream` waits on the device for every collective issued so far */
int fo_comm_wait(fo_comm* c, void* stream);
int fo_comm_destroy(fo_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* FACEOFF_HIP_H */
