// plan.h -- op descriptors shared by the host plan builder (plan.cpp), the device launcher
// (kernels.hip) and the test-only host interpreter (tests/interp.cpp).
//
// Data layout in HBM ("PBC"): every internal activation is [P][B][ld] fp32, P = H*W pixel planes,
// B samples per plane, channel-contiguous rows of `ld` floats (ld % 4 == 0, channels padded with
// zeros).  A convolution output pixel p is then a dense GEMM over samples:
//     out[p] (B x N) = sum over valid taps t:  in[q(p,t)] (B x C) * W[t]^T (C x N)
// so zero-padding taps are skipped outright (no boundary predicates, no wasted MFMA work) and
// every A/B operand tile is a plain row-major 2-D block.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define SBM_HD __host__ __device__
#else
#define SBM_HD
#endif

namespace sbm {

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_LRELU02 = 2, ACT_RELU = 3, ACT_SIGMOID = 4, ACT_SILU = 5, ACT_SOFTPLUS = 6, ACT_LRELU01 = 7 };

enum OpKind : int {
  OP_GEMM = 0,        // segmented implicit-GEMM convolution / linear (MFMA fp32)
  OP_DWCONV7 = 1,     // depthwise 7x7 + bias + per-sample time bias, emits GroupNorm partials
  OP_LINATTN = 2,     // LinearAttention core (softmax_d(q), softmax_n(k), context, out)
  OP_ATTN = 3,        // softmax attention core
  OP_GN_RES = 4,      // y = GroupNorm(1,C)(u) + x   (LinearAttention tail + Residual)
  OP_SINUS = 5,       // sinusoidal time embedding
  OP_NCHW_TO_PBC = 6, // API boundary: (B,C,H,W) -> [P][B][ld] with zero channel / spatial padding
  OP_BILINEAR = 7,    // bilinear upsample (align_corners=False) PBC -> PBC
  OP_GN_FINAL = 8,    // reduce GroupNorm partials to one partial (used when there are many)
  OP_GN32 = 9,        // GroupNorm(G, C) (+ SiLU) materialised: unet_openai normalization()
  OP_ATTN_QKV = 10,   // unet_openai QKVAttention core (softmax over T tokens, any head width)
  OP_AVGPOOL = 12,    // AvgPool2d(k) PBC -> PBC (encoders); 11 is the profiling bucket of the chain kernel
  OP_IM2COL = 15,     // pixel mode only: the row-packed input of a score net -> the [pixel][C KH KW] matrix init_conv multiplies as a 1x1 conv (13, 14: profiling buckets)
};

// A view of an activation tensor inside the workspace arena.
struct View {
  int64_t off = 0;   // float offset of element (p=0,b=0,c=0) in the workspace (includes channel offset)
  int ld = 0;        // row stride in floats
  int C = 0;         // channels visible through this view (padded count, multiple of 4)
  int H = 1, W = 1;  // pixel grid
};

struct GemmOp {
  int B = 0;
  int Hout = 1, Wout = 1;
  // source 1: convolution input
  View a;
  int a_ext = 0;          // 1: source 1 is the external [B][lda] input matrix of the plan (decoder z)
  int up = 1;             // nearest-neighbour upsample folded into the read (source pixel = q / up)
  int KH = 1, KW = 1, stride = 1, pad = 0, transposed = 0;
  int a_bcast = 0;        // 1: source is a single-pixel tensor read by every output pixel
  int a_rows = 0;         // 1: source "pixels" are image ROWS (CvtOp::rowpack): tap ky of output pixel (oy, ox) reads row oy - pad + ky,
                          //    the horizontal window lives in the per-output-pixel weights (w_pix_stride)
  int64_t w_off = 0;      // params: [tap][Npad][Kp]
  int64_t w_pix_stride = 0;  // params: extra offset per output pixel (z_lin), else 0
  // GroupNorm(1,C) applied on load to source 1
  int gn = 0;
  int64_t gn_part_off = 0;   // workspace: float2 [gn_nparts][B] (sum, sumsq)
  int gn_nparts = 0;
  float gn_inv_count = 0.f;  // 1 / (true channels * pixels)
  // GroupNorm is folded algebraically:  conv(GN(h))[m,n] = rstd_m * conv_{W*gamma}(h)[m,n] - mean_m*rstd_m * S[p][n] + B1[p][n]
  // with S[p][n] = sum over the VALID taps of pixel p and over k of gamma_k W[t][n][k], B1 likewise with beta_k.  The
  // packed weights at w_off already carry gamma; S and B1 are host tables [P_out][Npad].
  int64_t gn_s_off = -1, gn_b1_off = -1;
  // source 2: fused 1x1 convolution of a second tensor at the same output pixel (ConvNeXt res_conv,
  // RBlock size_conv)
  int has2 = 0;
  View a2;
  int up2 = 1;            // nearest upsample folded into the reads of source 2 and of the residual r
  int64_t w2_off = 0;     // params: [Npad][Kp2]
  // epilogue: out = act(acc + bias + r)
  int64_t bias_off = -1;  // params [Npad]
  int64_t bias_pix_stride = 0;  // params: extra bias offset per output pixel (z_lin), else 0
  int has_res = 0;
  View r;
  int act = ACT_NONE;
  View out;               // PBC destination (out.C = stored channels, multiple of 4)
  int out_nchw = 0;       // 1: destination is the external (B, N, Hn, Wn) NCHW output of the plan
  int N = 0;              // true output channels
  int Npad = 0;           // weight rows (multiple of 32)
  int crop_y = 0, crop_x = 0, Hn = 0, Wn = 0;  // out_nchw: crop of the padded grid
  int64_t stat_off = -1;  // workspace: float2 [P_out * ntiles][B] partial (sum, sumsq) of the stored rows
  int BM = 64, BN = 64;   // tile configuration (fixes ntiles = ceil(out.C / BN))
  int split = 1;          // split-K: each output tile's chunk list is dealt to `split` workgroups; the last to arrive sums the partial tiles
                          // (fixed order) and runs the epilogue.  Chosen in finalize_gemm_tables for grids that cannot fill the chip.
  // Host-built chunk tables (params arena, stored as int32 bit patterns): the kernel walks K in 32-wide chunks and
  // reads one 16-byte descriptor per chunk instead of doing tap geometry on the device.
  //   hdr [P_out] x {first descriptor, chunk count, source-1 chunk count, residual / source-2 pixel p2}
  //   desc[...]  x {A float offset, W float offset, gamma/beta K offset or -1, flags (bit 0: source 2)}
  int64_t hdr_off = -1, desc_off = -1;
  int64_t wt_off = -1, wt2_off = -1;   // row-kernel ops only: transposed copies of the weights, [tap][Kp][Npad] / [Kp2][Npad] (n contiguous)
  int Kp = 0, Kp2 = 0;    // K stride of the weight rows of source 1 / 2: a.C / a2.C rounded up to 32, zero
                          // filled, so that the kernel needs no K-tail predicate (gamma/beta are padded alike)
};

struct DwOp {
  int B = 0;
  View in, out;
  int64_t w_off = 0;      // params [49][C]
  int64_t bias_off = 0;   // params [C]
  int has_cond = 0;
  int64_t cond_off = 0;   // workspace [B][cond_ld] (+channel offset)
  int cond_ld = 0;
  int64_t stat_off = -1;  // workspace float2 [P][B], or [stat_slabs][B] when stat_slabs > 1
  int stat_slabs = 1;     // > 1: ONE partial per slab of 32 channels (all pixels), the last slab possibly narrower (16x16 level:
                          // dwconv7_t16_kernel's workgroups are (sample, slab)); stat_slabs = ceil(C / 32)
};

struct AttnOp {           // OP_LINATTN / OP_ATTN
  int B = 0;
  View qkv;               // C = 3*heads*dh : [q | k | v], head h dim d at h*dh + d
  View out;               // C = heads*dh
  int heads = 4, dh = 32;
  float scale = 0.f;
};

struct GnResOp {
  int B = 0;
  View u, x, out;
  int64_t part_off = 0; int nparts = 0; float inv_count = 0.f;
  int64_t gamma_off = 0, beta_off = 0;
  int out_ext = 0;        // 1: `out` is the plan's external output [B][out.ld] (LayerNorm tail of the Cel*Rep projection heads)
};

struct SinusOp { int B = 0; int dim = 0; int64_t out_off = 0; int ld = 0; int64_t freq_off = 0; /* params [dim/2] */ int cos_first = 0; };

struct Gn32Op { int B = 0; View in, out; int groups = 32; int silu = 0; int64_t gamma_off = 0, beta_off = 0; };

// qkv channels per head h: [q | k | v] at h*3*chh (unet_openai.py:333,351-352); out channels h*chh
struct AttnQkvOp { int B = 0; View qkv, out; int heads = 1, chh = 0; float scale = 0.f; };

struct CvtOp {
  int B = 0; int C = 0; int Hn = 0, Wn = 0; int pad_y = 0, pad_x = 0; View out;
  int rowpack = 0;        // 1: `out` is a view of H "pixels" (image rows) with W*Cpad channels each: element (y, x, c) at channel x*Cpad + c;
  int Wo = 0, Cpad = 0;   //    Wo columns of Cpad channels per row (init_conv's row-packed input)
};

struct BilinearOp { int B = 0; int scale = 2; View in, out; };

// in: the row-packed input (CvtOp::rowpack: H rows of W * Cpad channels); out: [H W pixels][k = c * KH * KW + ky * KW + kx], zero outside the image and for k >= C KH KW
struct Im2colOp { int B = 0; View in, out; int C = 0, Cpad = 0, H = 0, W = 0, KH = 0, KW = 0, pad = 0; };

struct GnFinalOp { int B = 0; int64_t part_off = 0; int nparts = 0; int64_t out_off = 0; };

struct Op {
  int kind = 0;
  std::string name;       // reference layer this op implements (for profiles and error messages)
  double flops = 0;       // 2*MAC this op issues (padding taps skipped, channel padding included)
  double flops_dense = 0; // GEMM ops: 2*MAC in the dense convention (zero-padding taps counted), the roofline numerator
  int fuse_next = 0;      // set at model creation.  1: this to_qkv GEMM and the LinearAttention core behind it run as one kernel; 2: so do the to_out.0
                          // GEMM and the GroupNorm + residual behind those (attnblock_kernel), which carry -1
  int time_path = 0;      // 1: depends on the time input only (sinusoid, time MLP, per-block condition GEMM): the sampler tabulates it
  int chain = 0;          // 1: runs inside the XCD-resident chain kernel (low-resolution ops; GEMM tile 32x32) -- chain.hip
  int row = 0;            // 1: runs inside the row kernel (the one-position level, four samples per workgroup, LDS-resident) -- rowrun.hip
  GemmOp g; DwOp dw; AttnOp at; GnResOp gr; SinusOp si; CvtOp cv; BilinearOp bl; GnFinalOp gf; Gn32Op gn; AttnQkvOp aq; Im2colOp ic;
};

struct ChunkDesc { int a_off, w_off, gkc, flags; };
struct PixHdr { int first, total, n1, p2; };

// One executable plan: a straight-line op list over two arenas.
struct Plan {
  int B = 0;
  std::vector<Op> ops;
  std::vector<float> params;     // packed weights (host copy; uploaded once)
  int64_t ws_floats = 0;         // workspace size
  int64_t zero_off = 0, zero_floats = 0;  // region that must be zero before every run (none today)
  // external I/O description
  int in_C = 0, in_H = 0, in_W = 0;    // NCHW input (score net) or [B][in_C] latent rows (decoders)
  int in_ld = 0;
  int out_C = 0, out_H = 0, out_W = 0;
  int needs_time = 0;
  int64_t time_off = -1;         // workspace [B] : the plan's time input is copied here
  int64_t cond_off = -1; int cond_ld = 0;   // workspace [B][cond_ld]: output of the time path (per-block time conditions), -1 if none
  int64_t aux_off = -1; int aux_C = 0;   // workspace [B][aux_C]: auxiliary conditioning input (UNetModel z), copied before a run
  double flops_dense = 0;        // 2*MAC, dense convention (zero-padding taps counted)
  double flops_exec = 0;         // 2*MAC actually issued (padding taps skipped, channel padding included)
  double flops_dense_gemm = 0;   // the part of flops_dense computed by GEMM ops
  double flops_dense_cond = 0;   // ... of which the per-block time-condition GEMM (a time-path op emitted directly)
  int64_t splitk_part_off = -1, splitk_cnt_off = -1;   // workspace: partial tiles [tile][split][BM*BN] / arrival counters [tile] shared by all split-K ops
  int64_t gn32_part_off = -1;    // workspace: fp64 (sum, sumsq) [B][16 slices][64 groups] shared by the two-pass GroupNorm(32) launches
  int64_t ident_off = -1;        // params: 32 ones followed by 32 zeros (identity gamma/beta for chunks without GroupNorm)
  std::string error;
};

// Ops whose pixel grid has at most this many pixels run inside the XCD-resident chain kernel (0 = none).  Read from the
// environment variable SBM_CHAIN_PIX by the Unet plan builder (default 16: the 4x4, 2x2 and 1x1 levels); batches smaller
// than SBM_CHAIN_MIN_B (default 192) do not use the chain kernel.
int chain_pix_from_env(int B);
// Ops of the one-position (1x1) level -- and the strided convolutions that enter / leave it -- run in the row kernel when the chain
// kernel is in use (SBM_ROW_PIX, default 1; 0 = keep them in the chain kernel).
int row_pix_from_env();
// How the next build_unet_plan of this thread marks the low-resolution ops: 0 legacy (chain kernel + row kernel, batch-size gated),
// 1 pixel mode (sample-owned kernel, pixrun.hip: every batch size, the one-position level included, time path per-layer).
void plan_set_pix_mode(int mode);
bool attn_block_on();   // SBM_ATTN_BLOCK=0 (read when a plan is built and when a model is created): the 8x8 attention block stays to_qkv + core as one kernel,
                        // to_out.0 and GroupNorm + residual in the pixel kernel's run; default: the whole block is one kernel (kernels.hip attnblock_kernel)
int plan_pix_mode();

struct UnetCfg {
  int dim = 64, channels = 3, init_dim = 0, out_dim = 0;
  int n_mults = 4;
  int dim_mults[8] = {1, 2, 4, 8, 0, 0, 0, 0};
  int with_time_emb = 1;
  int convnext_mult = 2;
  int H = 8, W = 8;
};

struct TensorSpec { std::string name; std::vector<int64_t> shape; int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; } };

// state_dict entries the builder consumes, in order (mirrors score_based_multimodal_autoencoder_amd/arch.py)
std::vector<TensorSpec> unet_tensor_specs(const UnetCfg& cfg);
// Build the forward plan of reference unet_model.Unet (unet_model.py:189-323) for batch size B.
// `tensors[i]` points at host fp32 data of spec i.
bool build_unet_plan(const UnetCfg& cfg, int B, const float* const* tensors, Plan* plan);

struct ResDecCfg {
  int variant = 0;        // bit 0: 0 ResVAE/ResAE (LeakyReLU, nearest), 1 ResVAEN/ResAEN (GELU, bilinear, sigmoid);
                          // bit 1 (encoder plans): CelImgRep / CelMaskRep = encoder mu -> ProjectionHead (h_vae_model.py:1040-1072)
  int size_in = 32, size_z = 64, img_ch = 3;
  int n_enc = 0; int enc[8][4];
  int n_dec = 0; int dec[8][4];
};
std::vector<TensorSpec> resdec_tensor_specs(const ResDecCfg& cfg);   // decoder-side entries only
bool build_resdec_plan(const ResDecCfg& cfg, int B, int z_ld, const float* const* tensors, Plan* plan);
// encoder side of the same modules (ResEncoder / ResEncoderN, h_vae_model_copy.py:41-71, 379-409): image -> [mu | logvar]
std::vector<TensorSpec> resenc_tensor_specs(const ResDecCfg& cfg);
bool build_resenc_plan(const ResDecCfg& cfg, int B, const float* const* tensors, Plan* plan);

struct UnetOACfg {
  int in_channels = 3, model_channels = 128, out_channels = 3, num_res_blocks = 2;
  int n_mult = 4; int channel_mult[8] = {1, 2, 4, 8, 0, 0, 0, 0};
  int n_attn = 0; int attn_res[8] = {0};
  int num_heads = 1, z_dim = 0, use_z = 0;
  int H = 16, W = 16;
};
std::vector<TensorSpec> unet_oa_tensor_specs(const UnetOACfg& cfg);
// reference unet_openai.UNetModel.forward (unet_openai.py:538-575); with_z selects the `z is not None` branch
bool build_unet_oa_plan(const UnetOACfg& cfg, int B, int with_z, const float* const* tensors, Plan* plan);

// lat_sm2_model.LSMPoly64_sm: 7 Linear layers with Softplus between (lat_sm2_model.py:348-371)
struct MlpCfg { int n_mod = 5, size_z = 64; };
std::vector<TensorSpec> mlp_tensor_specs(const MlpCfg& cfg);
bool build_mlp_plan(const MlpCfg& cfg, int B, const float* const* tensors, Plan* plan);

// CelebAAttrNewBN[AE].encoder (h_vae_model.py:718-736, 756-759; AE :845-863, 884-887): attributes (B, att_size) -> [mu | logvar]
struct AttrEncCfg { int size_z = 64, att_size = 18, ae = 0; };   // ae: 0 VAE (mu|logvar), 1 AE (mu), 2 CelAttrRep (keys under enc., mu -> ProjectionHead, h_vae_model.py:964-1002)
std::vector<TensorSpec> attrenc_tensor_specs(const AttrEncCfg& cfg);
bool build_attrenc_plan(const AttrEncCfg& cfg, int B, const float* const* tensors, Plan* plan);

struct AttrDecCfg { int size_z = 256, att_size = 18; };
std::vector<TensorSpec> attrdec_tensor_specs(const AttrDecCfg& cfg);
bool build_attrdec_plan(const AttrDecCfg& cfg, int B, int z_ld, const float* const* tensors, Plan* plan);

// builds hdr/desc for every GEMM op and the identity gamma/beta block; called at the end of every build_*_plan
void finalize_gemm_tables(Plan* plan);

SBM_HD inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Source pixel (in the stored, pre-upsample grid of source 1) that tap (ky,kx) of output pixel (oy,ox)
// reads, or -1 when the tap falls into the zero padding (then it is skipped).
SBM_HD inline int tap_source_pixel(int oy, int ox, int ky, int kx, int stride, int pad, int transposed, int up, int Hs, int Ws) {
  int iy, ix;
  const int He = Hs * up, We = Ws * up;
  if (!transposed) {
    iy = oy * stride - pad + ky; ix = ox * stride - pad + kx;
  } else {
    const int ny = oy + pad - ky, nx = ox + pad - kx;
    if (ny < 0 || nx < 0 || ny % stride || nx % stride) return -1;
    iy = ny / stride; ix = nx / stride;
  }
  if (iy < 0 || iy >= He || ix < 0 || ix >= We) return -1;
  return (iy / up) * Ws + (ix / up);
}

}  // namespace sbm
