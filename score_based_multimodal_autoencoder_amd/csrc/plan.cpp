// plan.cpp -- host-side plan builder and weight packer (pure C++, no HIP).
//
// Walks the reference architectures (unet_model.Unet, h_vae_model_copy.Res*VAE*.decoder,
// h_vae_model.CelebAAttrNewBN*.decoder) and emits a straight-line list of fused ops over the PBC
// layout described in plan.h, packing every weight tensor into the layout its kernel reads.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>

namespace sbm {

static int pad8(int c) { return round_up(c, 8); }

// ------------------------------------------------------------------------------------------------
// tensor specs (mirror of arch.py)
// ------------------------------------------------------------------------------------------------
namespace {

struct SpecList {
  std::vector<TensorSpec> v;
  void add(const std::string& n, std::vector<int64_t> s) { v.push_back({n, std::move(s)}); }
};

void spec_convnext(SpecList& L, const std::string& pre, int cin, int cout, int td, int mult) {
  if (td) { L.add(pre + "mlp.1.weight", {cin, td}); L.add(pre + "mlp.1.bias", {cin}); }
  L.add(pre + "ds_conv.weight", {cin, 1, 7, 7}); L.add(pre + "ds_conv.bias", {cin});
  L.add(pre + "net.0.weight", {cin}); L.add(pre + "net.0.bias", {cin});
  L.add(pre + "net.1.weight", {cout * mult, cin, 3, 3}); L.add(pre + "net.1.bias", {cout * mult});
  L.add(pre + "net.3.weight", {cout * mult}); L.add(pre + "net.3.bias", {cout * mult});
  L.add(pre + "net.4.weight", {cout, cout * mult, 3, 3}); L.add(pre + "net.4.bias", {cout});
  if (cin != cout) { L.add(pre + "res_conv.weight", {cout, cin, 1, 1}); L.add(pre + "res_conv.bias", {cout}); }
}

void spec_attn(SpecList& L, const std::string& pre, int c, bool linear) {
  const int hid = 128;
  L.add(pre + "fn.fn.to_qkv.weight", {hid * 3, c, 1, 1});
  if (linear) {
    L.add(pre + "fn.fn.to_out.0.weight", {c, hid, 1, 1}); L.add(pre + "fn.fn.to_out.0.bias", {c});
    L.add(pre + "fn.fn.to_out.1.weight", {c}); L.add(pre + "fn.fn.to_out.1.bias", {c});
  } else {
    L.add(pre + "fn.fn.to_out.weight", {c, hid, 1, 1}); L.add(pre + "fn.fn.to_out.bias", {c});
  }
  L.add(pre + "fn.norm.weight", {c}); L.add(pre + "fn.norm.bias", {c});
}

struct UnetDims {
  int init_dim, out_dim, td;
  std::vector<int> dims;
  std::vector<std::pair<int, int>> in_out;
};

UnetDims unet_dims(const UnetCfg& c) {
  UnetDims d;
  d.init_dim = c.init_dim > 0 ? c.init_dim : c.dim / 3 * 2;       // unet_model.py:206
  d.out_dim = c.out_dim > 0 ? c.out_dim : c.channels;              // unet_model.py:269
  d.td = c.with_time_emb ? c.dim * 4 : 0;
  d.dims.push_back(d.init_dim);
  for (int i = 0; i < c.n_mults; ++i) d.dims.push_back(c.dim * c.dim_mults[i]);
  for (size_t i = 0; i + 1 < d.dims.size(); ++i) d.in_out.push_back({d.dims[i], d.dims[i + 1]});
  return d;
}

void spec_rblock(SpecList& L, const std::string& pre, int cin, int cmid, int cout) {
  auto bn = [&](const std::string& p, int c) {
    L.add(p + "weight", {c}); L.add(p + "bias", {c}); L.add(p + "running_mean", {c}); L.add(p + "running_var", {c});
  };
  L.add(pre + "conv.0.weight", {cmid, cin, 3, 3}); bn(pre + "conv.1.", cmid);
  L.add(pre + "conv.3.weight", {cout, cmid, 3, 3}); bn(pre + "conv.4.", cout);
  L.add(pre + "size_conv.weight", {cout, cin, 1, 1});
}

}  // namespace

std::vector<TensorSpec> unet_tensor_specs(const UnetCfg& c) {
  UnetDims d = unet_dims(c);
  SpecList L;
  L.add("init_conv.weight", {d.init_dim, c.channels, 7, 7}); L.add("init_conv.bias", {d.init_dim});
  if (d.td) {
    L.add("time_mlp.1.weight", {d.td, c.dim}); L.add("time_mlp.1.bias", {d.td});
    L.add("time_mlp.3.weight", {d.td, d.td}); L.add("time_mlp.3.bias", {d.td});
  }
  const int n = (int)d.in_out.size();
  for (int i = 0; i < n; ++i) {
    auto [ci, co] = d.in_out[i];
    std::string p = "downs." + std::to_string(i) + ".";
    spec_convnext(L, p + "0.", ci, co, d.td, c.convnext_mult);
    spec_convnext(L, p + "1.", co, co, d.td, c.convnext_mult);
    spec_attn(L, p + "2.", co, true);
    if (i < n - 1) { L.add(p + "3.weight", {co, co, 4, 4}); L.add(p + "3.bias", {co}); }
  }
  const int mid = d.dims.back();
  spec_convnext(L, "mid_block1.", mid, mid, d.td, c.convnext_mult);
  spec_attn(L, "mid_attn.", mid, false);
  spec_convnext(L, "mid_block2.", mid, mid, d.td, c.convnext_mult);
  for (int j = 0; j < n - 1; ++j) {
    auto [ci, co] = d.in_out[n - 1 - j];
    std::string p = "ups." + std::to_string(j) + ".";
    spec_convnext(L, p + "0.", co * 2, ci, d.td, c.convnext_mult);
    spec_convnext(L, p + "1.", ci, ci, d.td, c.convnext_mult);
    spec_attn(L, p + "2.", ci, true);
    L.add(p + "3.weight", {ci, ci, 4, 4}); L.add(p + "3.bias", {ci});
  }
  spec_convnext(L, "final_conv.0.", c.dim, c.dim, 0, c.convnext_mult);
  L.add("final_conv.1.weight", {d.out_dim, c.dim, 1, 1}); L.add("final_conv.1.bias", {d.out_dim});
  return L.v;
}

std::vector<TensorSpec> resdec_tensor_specs(const ResDecCfg& c) {
  SpecList L;
  int init = c.size_in;
  for (int i = 0; i < c.n_enc; ++i) init /= c.enc[i][3];
  const int clast = c.enc[c.n_enc - 1][2];
  L.add("z_lin.weight", {(int64_t)init * init * clast, c.size_z}); L.add("z_lin.bias", {(int64_t)init * init * clast});
  for (int i = 0; i < c.n_dec; ++i)
    spec_rblock(L, "dec.r_blocks." + std::to_string(i) + ".", c.dec[i][0], c.dec[i][1], c.dec[i][2]);
  const int cc = c.dec[c.n_dec - 1][2];
  spec_rblock(L, "dec.ch_dec.0.", cc, cc, cc);
  L.add("dec.ch_dec.1.weight", {c.img_ch, cc, 5, 5}); L.add("dec.ch_dec.1.bias", {c.img_ch});
  return L.v;
}

static void spec_proj_head(SpecList& L, int d) {
  L.add("proj.projection.weight", {d, d}); L.add("proj.projection.bias", {d});
  L.add("proj.fc.weight", {d, d}); L.add("proj.fc.bias", {d});
  L.add("proj.layer_norm.weight", {d}); L.add("proj.layer_norm.bias", {d});
}

std::vector<TensorSpec> resenc_tensor_specs(const ResDecCfg& c) {
  SpecList L;
  const int c0 = c.enc[0][0];
  L.add("enc.ch_enc.0.weight", {c0, c.img_ch, 5, 5}); L.add("enc.ch_enc.0.bias", {c0});
  L.add("enc.ch_enc.1.weight", {c0}); L.add("enc.ch_enc.1.bias", {c0}); L.add("enc.ch_enc.1.running_mean", {c0}); L.add("enc.ch_enc.1.running_var", {c0});
  for (int i = 0; i < c.n_enc; ++i) spec_rblock(L, "enc.r_blocks." + std::to_string(i) + ".", c.enc[i][0], c.enc[i][1], c.enc[i][2]);
  int s = c.size_in / 2;
  for (int i = 0; i < c.n_enc; ++i) s /= c.enc[i][3];
  const int64_t lin = (int64_t)s * s * (c.enc[c.n_enc - 1][2] / 2);
  L.add("enc.mu_lin.weight", {c.size_z, lin}); L.add("enc.mu_lin.bias", {c.size_z});
  if (c.variant & 2) spec_proj_head(L, c.size_z);
  else { L.add("enc.logvar_lin.weight", {c.size_z, lin}); L.add("enc.logvar_lin.bias", {c.size_z}); }
  return L.v;
}

static const int kAttrW[5] = {512, 512, 512, 256, 128};

static const int kAttrEncW[5] = {128, 256, 512, 512, 512};

std::vector<TensorSpec> attrenc_tensor_specs(const AttrEncCfg& c) {
  SpecList L;
  int prev = c.att_size;
  const std::string pre = c.ae == 2 ? "enc." : "";
  for (int i = 0; i < 5; ++i) {
    std::string l = pre + "enc_net." + std::to_string(3 * i) + ".", b = pre + "enc_net." + std::to_string(3 * i + 1) + ".";
    L.add(l + "weight", {kAttrEncW[i], prev}); L.add(l + "bias", {kAttrEncW[i]});
    L.add(b + "weight", {kAttrEncW[i]}); L.add(b + "bias", {kAttrEncW[i]});
    L.add(b + "running_mean", {kAttrEncW[i]}); L.add(b + "running_var", {kAttrEncW[i]});
    prev = kAttrEncW[i];
  }
  L.add(pre + "mu_lin.weight", {c.size_z, prev}); L.add(pre + "mu_lin.bias", {c.size_z});
  if (c.ae == 0) { L.add("logvar_lin.weight", {c.size_z, prev}); L.add("logvar_lin.bias", {c.size_z}); }
  if (c.ae == 2) spec_proj_head(L, c.size_z);
  return L.v;
}

std::vector<TensorSpec> attrdec_tensor_specs(const AttrDecCfg& c) {
  SpecList L;
  int prev = c.size_z;
  for (int i = 0; i < 5; ++i) {
    std::string l = "dec_net." + std::to_string(3 * i) + ".", b = "dec_net." + std::to_string(3 * i + 1) + ".";
    L.add(l + "weight", {kAttrW[i], prev}); L.add(l + "bias", {kAttrW[i]});
    L.add(b + "weight", {kAttrW[i]}); L.add(b + "bias", {kAttrW[i]});
    L.add(b + "running_mean", {kAttrW[i]}); L.add(b + "running_var", {kAttrW[i]});
    prev = kAttrW[i];
  }
  L.add("dec_net.15.weight", {c.att_size, prev}); L.add("dec_net.15.bias", {c.att_size});
  return L.v;
}

// ------------------------------------------------------------------------------------------------
// builder
// ------------------------------------------------------------------------------------------------
namespace {

bool attn_block_on_impl() { const char* e = std::getenv("SBM_ATTN_BLOCK"); return !(e && std::atoi(e) == 0); }
// pixel mode: the first and the last conv of a score net inside the sample-owned kernel (read when a plan is built)
bool pix_io_on() { const char* e = std::getenv("SBM_PIX_IO"); return !(e && std::atoi(e) == 0); }

// reference channel index -> channel index inside a (possibly concatenated, padded) buffer view
struct ChanMap {
  std::vector<int> idx;
  static ChanMap identity(int c) { ChanMap m; m.idx.resize(c); for (int i = 0; i < c; ++i) m.idx[i] = i; return m; }
  static ChanMap concat(int c0, int c0pad, int c1) {
    ChanMap m; m.idx.resize(c0 + c1);
    for (int i = 0; i < c0; ++i) m.idx[i] = i;
    for (int i = 0; i < c1; ++i) m.idx[c0 + i] = c0pad + i;
    return m;
  }
  bool is_identity() const { for (size_t i = 0; i < idx.size(); ++i) if (idx[i] != (int)i) return false; return true; }
  int n() const { return (int)idx.size(); }
};

struct Block { int64_t off, n; };

class Builder {
 public:
  Builder(Plan* p, int B, const std::vector<TensorSpec>& specs, const float* const* tensors) : plan(p), B(B) {
    plan->B = B;
    for (size_t i = 0; i < specs.size(); ++i) { index[specs[i].name] = (int)i; this->specs = &specs; }
    this->tensors = tensors;
  }
  Plan* plan; int B;
  const std::vector<TensorSpec>* specs = nullptr;
  const float* const* tensors = nullptr;
  std::map<std::string, int> index;
  std::vector<Block> free_list;

  const float* T(const std::string& name, int64_t expect_numel = -1) const {
    auto it = index.find(name);
    if (it == index.end()) throw std::runtime_error("plan: missing tensor " + name);
    if (expect_numel >= 0 && (*specs)[it->second].numel() != expect_numel)
      throw std::runtime_error("plan: tensor " + name + " has unexpected size");
    const float* p = tensors[it->second];
    if (!p) throw std::runtime_error("plan: null pointer for tensor " + name);
    return p;
  }
  bool has(const std::string& name) const { return index.count(name) != 0; }

  // ---- params arena (16-byte aligned blocks) ----
  int64_t add_params(const std::vector<float>& v) {
    int64_t off = (int64_t)plan->params.size();
    plan->params.insert(plan->params.end(), v.begin(), v.end());
    while (plan->params.size() % 4) plan->params.push_back(0.f);
    return off;
  }

  // ---- workspace allocator: first-fit free list, blocks rounded to 64 floats (256 B) ----
  // `pin`: the block is taken from fresh memory and never returned to the free list.  Buffers touched by ops of the chain
  // kernel need this: its clusters (one per XCD, one sample tile each) run through the op list independently, so memory
  // may not be recycled between ops with different layouts -- a fast cluster would overwrite what a slow one still reads.
  int64_t ws_alloc(int64_t n, bool pin = false) {
    // +32 floats of slack: the GEMM kernel reads whole 32-float K chunks, so the last row of a tensor whose
    // channel count is not a multiple of 32 is read up to 28 floats past its end (times zero weights)
    n = (n + 32 + 63) / 64 * 64;
    if (pin) {
      int64_t off = plan->ws_floats;
      plan->ws_floats += n;
      sizes[off] = n; pinned.insert(off);
      return off;
    }
    for (size_t i = 0; i < free_list.size(); ++i) {
      if (free_list[i].n >= n) {
        int64_t off = free_list[i].off;
        free_list[i].off += n; free_list[i].n -= n;
        if (free_list[i].n == 0) free_list.erase(free_list.begin() + i);
        sizes[off] = n;
        return off;
      }
    }
    int64_t off = plan->ws_floats;
    plan->ws_floats += n;
    sizes[off] = n;
    return off;
  }
  void ws_free(int64_t off) {
    auto it = sizes.find(off);
    if (it == sizes.end()) throw std::runtime_error("plan: double free");
    if (pinned.count(off)) { sizes.erase(it); pinned.erase(off); return; }
    Block b{off, it->second};
    sizes.erase(it);
    // insert sorted + coalesce
    auto pos = std::lower_bound(free_list.begin(), free_list.end(), b, [](const Block& x, const Block& y) { return x.off < y.off; });
    pos = free_list.insert(pos, b);
    size_t i = pos - free_list.begin();
    if (i + 1 < free_list.size() && free_list[i].off + free_list[i].n == free_list[i + 1].off) {
      free_list[i].n += free_list[i + 1].n; free_list.erase(free_list.begin() + i + 1);
    }
    if (i > 0 && free_list[i - 1].off + free_list[i - 1].n == free_list[i].off) {
      free_list[i - 1].n += free_list[i].n; free_list.erase(free_list.begin() + i);
    }
  }
  std::map<int64_t, int64_t> sizes;
  std::set<int64_t> pinned;
  bool in_chain(int P) const { return chain_pix > 0 && P <= chain_pix; }

  View new_act(int H, int W, int C) {
    View v; v.H = H; v.W = W; v.C = pad8(C); v.ld = v.C;
    v.off = ws_alloc((int64_t)H * W * B * v.ld, in_chain(H * W));
    return v;
  }
  void free_act(const View& v) { ws_free(v.off); }
  // a score net's converted input: pinned, because the sampler's update kernels keep it current BETWEEN forwards (sampler.hip store_pbc) --
  // nothing else may recycle the block, and its zero padding channels must stay zero
  View new_input_act(int H, int W, int C) {
    View v; v.H = H; v.W = W; v.C = pad8(C); v.ld = v.C;
    v.off = ws_alloc((int64_t)H * W * B * v.ld, true);
    return v;
  }
  int64_t new_stats(int nparts, bool pin = false) { return ws_alloc((int64_t)nparts * B * 2, pin); }
  int chain_pix = 0;        // ops with P <= chain_pix are marked for the chain kernel
  bool pix_mode = false;    // the marked ops are meant for the sample-owned pixel kernel (pixrun.hip) instead of the chain kernel
  int row_pix = 0;          // ... and of those, ops with P <= row_pix for the row kernel (rowrun.hip)
  // [tap][Npad][Kp] (k contiguous) -> [tap][Kp][Npad] (n contiguous): the row kernel streams weights with the output column on the lane
  int64_t add_transposed(const std::vector<float>& w, int ntap, int Npad, int Kp) {
    std::vector<float> t(w.size());
    for (int tp = 0; tp < ntap; ++tp)
      for (int n = 0; n < Npad; ++n)
        for (int k = 0; k < Kp; ++k) t[((size_t)tp * Kp + k) * Npad + n] = w[((size_t)tp * Npad + n) * Kp + k];
    return add_params(t);
  }

  // ---- tile choice ----
  // kchunks: K depth in 32-wide chunks; a very short K loop makes the launch prologue/epilogue-bound, where more, smaller
  // workgroups overlap better than few large ones
  // split-K policy (finalize_gemm_tables applies it): grids under splitk_max_tiles() tiles are split so that about
  // splitk_target() workgroups run, each keeping at least splitk_min_chunks() K chunks
  static bool splitk_enabled() { static const bool on = !getenv("SBM_NO_SPLITK"); return on; }
  static long splitk_max_tiles() { static const long v = getenv("SBM_SPLITK_TILES") ? atol(getenv("SBM_SPLITK_TILES")) : 384; return v; }
  static long splitk_target() { static const long v = getenv("SBM_SPLITK_TARGET") ? atol(getenv("SBM_SPLITK_TARGET")) : 512; return v; }
  static int splitk_min_chunks() { static const int v = getenv("SBM_SPLITK_MINCH") ? atoi(getenv("SBM_SPLITK_MINCH")) : 6; return v; }
  // SBM_SPLITK_BALANCE=<min chunks> (experiment): grids of one to two workgroup waves (384 .. 767 tiles: every CU gets its two resident workgroups
  // and the launch lasts as long as the CU with the two longest tiles) are cut in K halves, so that the hardware dispatcher deals ~4 units per
  // CU in arrival order
  static int splitk_balance_min() { static const int v = getenv("SBM_SPLITK_BALANCE") ? atoi(getenv("SBM_SPLITK_BALANCE")) : 0; return v; }
  static int splitk_for(long tiles, int min_chunks) {
    if (splitk_enabled() && splitk_balance_min() > 0 && tiles >= splitk_max_tiles() && tiles < 768 && min_chunks >= splitk_balance_min()) return 2;
    if (!splitk_enabled() || tiles >= splitk_max_tiles()) return 1;
    const long S = std::min<long>(std::min<long>((splitk_target() + tiles - 1) / tiles, min_chunks / splitk_min_chunks()), 16);
    return S >= 2 ? (int)S : 1;
  }

  // kchunks: dense K chunk count; kmin: the SHORTEST chunk list of an output pixel (border pixels of a padded convolution see fewer taps)
  static void choose_tile(int B, int P, int nstore, int* BM, int* BN, int kchunks = 1 << 20, int kmin = -1) {
    if (kmin < 0) kmin = kchunks;
    static const bool wide64 = !getenv("SBM_NO_TILE_64x128");
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    for (auto& c : cand) {
      if (c[0] == 64 && c[1] == 128 && (!wide64 || B > 64)) continue;
      if (kchunks <= 4 && c[0] > 64) continue;
      if (c[1] > 64 && nstore <= 64) continue;
      if (c[0] > 64 && B <= 64) continue;
      long blocks = (long)P * ceil_div(B, c[0]) * ceil_div(nstore, c[1]);
      // batch <= 64: the wide tile also wins when split-K can bring the grid back up (same workgroup count, better reuse per workgroup)
      // (measured per layer on configs 4 and 5: wide + split pays only while every split keeps >= 20 chunks and S <= 8)
      if (c[0] == 64 && c[1] == 128) {
        const int S = splitk_for(blocks, kmin);
        if (S > 1 && (S > 8 || kmin / S < 20)) continue;
        blocks *= S;
      }
      if (blocks >= 384) { *BM = c[0]; *BN = c[1]; return; }
    }
    *BM = 64; *BN = 64;
  }

  // ---- weight packing ----
  // conv weight [Cout][Cin][KH][KW] (or transposed-conv [Cin][Cout][KH][KW]) -> [tap][Npad][Ca]
  std::vector<float> pack_conv(const float* w, int cout, int cin, int kh, int kw, const ChanMap& cm, int Ca, int Npad,
                               bool transposed, const std::vector<double>* out_scale = nullptr, const float* in_scale = nullptr) {
    std::vector<float> o((size_t)kh * kw * Npad * Ca, 0.f);
    for (int n = 0; n < cout; ++n)
      for (int c = 0; c < cin; ++c)
        for (int t = 0; t < kh * kw; ++t) {
          size_t src = transposed ? (((size_t)c * cout + n) * kh * kw + t) : (((size_t)n * cin + c) * kh * kw + t);
          double v = w[src];
          if (out_scale) v *= (*out_scale)[n];
          if (in_scale) v *= (double)in_scale[c];
          o[((size_t)t * Npad + n) * Ca + cm.idx[c]] = (float)v;
        }
    return o;
  }
  std::vector<float> pack_vec(const float* v, int n, const ChanMap* cm, int npad) {
    std::vector<float> o(npad, 0.f);
    for (int i = 0; i < n; ++i) o[cm ? cm->idx[i] : i] = v[i];
    return o;
  }

  // ---- op emission helpers ----
  Op& push(int kind, const std::string& name) {
    plan->ops.emplace_back();
    plan->ops.back().kind = kind; plan->ops.back().name = name;
    return plan->ops.back();
  }

  struct ConvSpec {
    std::string name;
    View a; ChanMap cm; int cin_true = 0;
    int up = 1, kh = 1, kw = 1, stride = 1, pad = 0; bool transposed = false;
    const float* w = nullptr; const float* bias = nullptr;
    int cout = 0;
    int Hout = 1, Wout = 1;
    // GN on load
    bool gn = false; int64_t gn_part = 0; int gn_nparts = 0; float gn_inv = 0; const float* gamma = nullptr; const float* beta = nullptr;
    // source 2
    bool has2 = false; View a2; ChanMap cm2; const float* w2 = nullptr; const float* bias2 = nullptr; int cin2 = 0; int up2 = 1;
    bool has_res = false; View r;
    int act = ACT_NONE;
    bool want_stats = false;
    bool out_nchw = false; int crop_y = 0, crop_x = 0, Hn = 0, Wn = 0;
    const std::vector<double>* out_scale = nullptr;     // BN fold: per-output-channel scale on w
    const std::vector<double>* out_shift = nullptr;     // BN fold: replaces / adds to bias
    View* dest = nullptr;    // write into this existing view instead of a fresh tensor
  };

  struct ConvResult { View out; int64_t stat_off = -1; int stat_nparts = 0; };

  ConvResult emit_conv(const ConvSpec& s) {
    Op& op = push(OP_GEMM, s.name);
    GemmOp& g = op.g;
    g.B = B; g.Hout = s.Hout; g.Wout = s.Wout;
    g.a = s.a; g.up = s.up; g.KH = s.kh; g.KW = s.kw; g.stride = s.stride; g.pad = s.pad; g.transposed = s.transposed;
    g.N = s.cout;
    const int nstore = s.out_nchw ? s.cout : pad8(s.cout);
    g.Npad = round_up(nstore, 32);
    g.Kp = round_up(s.a.C, 32);
    const int P_in_px = s.a.H * s.a.W, P_out_px = s.Hout * s.Wout;
    const bool row_op = row_pix > 0 && chain_pix > 0 && !s.out_nchw && s.up == 1 && s.up2 == 1 &&
                        ((P_out_px <= row_pix && P_in_px <= 4 * row_pix) || (P_in_px <= row_pix && P_out_px <= 4 * row_pix)) && P_in_px <= 4 && P_out_px <= 4;
    {
      std::vector<float> packed = pack_conv(s.w, s.cout, s.cin_true, s.kh, s.kw, s.cm, g.Kp, g.Npad, s.transposed, s.out_scale, s.gn ? s.gamma : nullptr);
      // row-kernel ops also keep a transposed copy [tap][Kp][Npad] (the pixel kernel packs its own fragment-order copies, pixrun.hip)
      if (row_op) g.wt_off = add_transposed(packed, s.kh * s.kw, g.Npad, g.Kp);
      g.w_off = add_params(packed);
    }
    std::vector<double> bias(g.Npad, 0.0);
    bool any_bias = false;
    if (s.bias) { for (int n = 0; n < s.cout; ++n) bias[n] += (s.out_scale ? (*s.out_scale)[n] : 1.0) * s.bias[n]; any_bias = true; }
    if (s.out_shift) { for (int n = 0; n < s.cout; ++n) bias[n] += (*s.out_shift)[n]; any_bias = true; }
    if (s.gn) {
      g.gn = 1; g.gn_part_off = s.gn_part; g.gn_nparts = s.gn_nparts; g.gn_inv_count = s.gn_inv;
      // per-tap column sums of W*gamma and W*beta, then per-pixel sums over the valid taps
      const int ntap = s.kh * s.kw, P = s.Hout * s.Wout;
      std::vector<double> c2((size_t)ntap * g.Npad, 0.0), c1((size_t)ntap * g.Npad, 0.0);
      for (int n = 0; n < s.cout; ++n)
        for (int c = 0; c < s.cin_true; ++c)
          for (int t = 0; t < ntap; ++t) {
            const size_t src = s.transposed ? (((size_t)c * s.cout + n) * ntap + t) : (((size_t)n * s.cin_true + c) * ntap + t);
            const double w = (double)s.w[src] * (s.out_scale ? (*s.out_scale)[n] : 1.0);
            c2[(size_t)t * g.Npad + n] += w * (double)s.gamma[c];
            c1[(size_t)t * g.Npad + n] += w * (double)s.beta[c];
          }
      std::vector<float> S((size_t)P * g.Npad, 0.f), B1((size_t)P * g.Npad, 0.f);
      for (int p = 0; p < P; ++p) {
        const int oy = p / s.Wout, ox = p % s.Wout;
        std::vector<double> ss(g.Npad, 0.0), bb(g.Npad, 0.0);
        for (int ky = 0; ky < s.kh; ++ky) for (int kx = 0; kx < s.kw; ++kx) {
          if (tap_source_pixel(oy, ox, ky, kx, s.stride, s.pad, s.transposed ? 1 : 0, s.up, s.a.H, s.a.W) < 0) continue;
          const int t = ky * s.kw + kx;
          for (int n = 0; n < g.Npad; ++n) { ss[n] += c2[(size_t)t * g.Npad + n]; bb[n] += c1[(size_t)t * g.Npad + n]; }
        }
        for (int n = 0; n < g.Npad; ++n) { S[(size_t)p * g.Npad + n] = (float)ss[n]; B1[(size_t)p * g.Npad + n] = (float)bb[n]; }
      }
      g.gn_s_off = add_params(S);
      g.gn_b1_off = add_params(B1);
    }
    if (s.has2) {
      g.has2 = 1; g.a2 = s.a2; g.up2 = s.up2;
      g.Kp2 = round_up(s.a2.C, 32);
      std::vector<float> packed2 = pack_conv(s.w2, s.cout, s.cin2, 1, 1, s.cm2, g.Kp2, g.Npad, false, nullptr);
      if (row_op) g.wt2_off = add_transposed(packed2, 1, g.Npad, g.Kp2);
      g.w2_off = add_params(packed2);
      if (s.bias2) { for (int n = 0; n < s.cout; ++n) bias[n] += s.bias2[n]; any_bias = true; }
    }
    if (any_bias) { std::vector<float> bf(bias.begin(), bias.end()); g.bias_off = add_params(bf); }
    if (s.has_res) { g.has_res = 1; g.r = s.r; g.up2 = s.up2; }
    g.act = s.act;
    const int P = s.Hout * s.Wout;
    int min_taps = s.kh * s.kw;
    if (!g.a_rows && !g.a_bcast)
      for (int p = 0; p < P; ++p) {
        int nv = 0;
        for (int ky = 0; ky < s.kh; ++ky) for (int kx = 0; kx < s.kw; ++kx)
          nv += tap_source_pixel(p / s.Wout, p % s.Wout, ky, kx, s.stride, s.pad, s.transposed ? 1 : 0, s.up, s.a.H, s.a.W) >= 0;
        min_taps = std::min(min_taps, nv);
      }
    choose_tile(B, P, nstore, &g.BM, &g.BN, s.kh * s.kw * (g.Kp / 32) + (s.has2 ? g.Kp2 / 32 : 0), min_taps * (g.Kp / 32) + (s.has2 ? g.Kp2 / 32 : 0));
    // pixel mode: the sample-owned kernel keeps 16 rows per LDS slab, so a convolution that READS a larger grid (the strided one from
    // 8x8 down to 4x4) stays a per-layer launch
    // (pixel mode also takes the conv that writes the plan's NCHW output, uncropped: the run then ends in the external tensor, SBM_PIX_IO=0 keeps it a launch of its own)
    const bool nchw_pix = s.out_nchw && pix_mode && pix_io_on() && s.crop_y == 0 && s.crop_x == 0 && s.Hn == s.Hout && s.Wn == s.Wout && !s.want_stats;
    if (chain_pix > 0 && P <= chain_pix && (!s.out_nchw || nchw_pix) && !(pix_mode && P_in_px > chain_pix)) { op.chain = 1; g.BM = 32; g.BN = 32; }
    if (row_op) { op.row = 1; op.chain = 0; g.BM = 32; g.BN = 32; }
    ConvResult res;
    if (s.out_nchw) {
      g.out_nchw = 1; g.crop_y = s.crop_y; g.crop_x = s.crop_x; g.Hn = s.Hn; g.Wn = s.Wn;
      g.out.C = s.cout; g.out.H = s.Hout; g.out.W = s.Wout;
    } else {
      g.out = s.dest ? *s.dest : new_act(s.Hout, s.Wout, s.cout);
      res.out = g.out;
    }
    if (s.want_stats) {
      const int ntiles = ceil_div(nstore, g.BN);
      res.stat_nparts = P * ntiles;
      res.stat_off = g.stat_off = new_stats(res.stat_nparts, op.chain != 0 || op.row != 0);
    }
    // flop accounting
    double taps_valid = 0;
    for (int oy = 0; oy < s.Hout; ++oy) for (int ox = 0; ox < s.Wout; ++ox)
      for (int ky = 0; ky < s.kh; ++ky) for (int kx = 0; kx < s.kw; ++kx) {
        int iy, ix; bool ok;
        const int Hs = s.a.H * s.up, Ws = s.a.W * s.up;
        if (!s.transposed) { iy = oy * s.stride - s.pad + ky; ix = ox * s.stride - s.pad + kx; ok = iy >= 0 && iy < Hs && ix >= 0 && ix < Ws; }
        else { int ny = oy + s.pad - ky, nx = ox + s.pad - kx; ok = ny >= 0 && nx >= 0 && ny % s.stride == 0 && nx % s.stride == 0 && ny / s.stride < Hs && nx / s.stride < Ws; }
        taps_valid += ok ? 1 : 0;
      }
    double dense_taps = s.transposed ? (double)s.a.H * s.a.W * s.kh * s.kw : (double)P * s.kh * s.kw;
    plan->flops_dense += 2.0 * B * s.cout * (dense_taps * s.cin_true + (s.has2 ? (double)P * s.cin2 : 0.0));
    plan->flops_dense_gemm += 2.0 * B * s.cout * (dense_taps * s.cin_true + (s.has2 ? (double)P * s.cin2 : 0.0));
    op.flops_dense = 2.0 * B * s.cout * (dense_taps * s.cin_true + (s.has2 ? (double)P * s.cin2 : 0.0));
    op.flops = 2.0 * B * g.Npad * (taps_valid * s.a.C + (s.has2 ? (double)P * s.a2.C : 0.0));
    plan->flops_exec += op.flops;
    return res;
  }
};

// ConvNeXt block (unet_model.py:92-124).  `x` is the block input view with channel map `cm`.
struct BlockIO { View out; int64_t stat_off = -1; int stat_nparts = 0; };

BlockIO emit_convnext(Builder& b, const std::string& pre, const View& x, const ChanMap& cm, int cin, int cout, int mult,
                      bool has_time, int64_t cond_off, int cond_ld, bool want_out_stats, View* dest) {
  const int H = x.H, W = x.W, P = H * W;
  // depthwise 7x7 + bias + time condition -> h0, GroupNorm partials
  View h0; h0.H = H; h0.W = W; h0.C = x.C; h0.ld = x.C; h0.off = b.ws_alloc((int64_t)P * b.B * h0.ld, b.in_chain(P));
  static const bool dw_slabs_on = !getenv("SBM_NO_DWT16");
  const int slabs = (dw_slabs_on && H == 16 && W == 16 && x.C > 32 && !b.in_chain(P)) ? (x.C + 31) / 32 : 1;      // the last slab may be narrower
  int64_t st0 = b.new_stats(slabs > 1 ? slabs : P, b.in_chain(P));       // per pixel, or (16x16 slab kernel) per 32-channel slab
  {
    Op& op = b.push(OP_DWCONV7, pre + "ds_conv");
    if (b.chain_pix > 0 && P <= b.chain_pix) op.chain = 1;
    if (b.chain_pix > 0 && b.row_pix > 0 && P <= b.row_pix) { op.row = 1; op.chain = 0; }
    DwOp& d = op.dw; d.B = b.B; d.in = x; d.out = h0;
    const float* w = b.T(pre + "ds_conv.weight", (int64_t)cin * 49);
    std::vector<float> wp((size_t)49 * x.C, 0.f);
    for (int c = 0; c < cin; ++c) for (int t = 0; t < 49; ++t) wp[(size_t)t * x.C + cm.idx[c]] = w[(size_t)c * 49 + t];
    d.w_off = b.add_params(wp);
    d.bias_off = b.add_params(b.pack_vec(b.T(pre + "ds_conv.bias", cin), cin, &cm, x.C));
    d.has_cond = has_time ? 1 : 0; d.cond_off = cond_off; d.cond_ld = cond_ld;
    d.stat_off = st0; d.stat_slabs = slabs;
    b.plan->flops_dense += 2.0 * b.B * cin * 49.0 * P;
    b.plan->flops_exec += 2.0 * b.B * x.C * 49.0 * P;
  }
  // GN -> conv3x3 -> GELU  (net.0, net.1, net.2)
  Builder::ConvSpec c1;
  c1.name = pre + "net.1"; c1.a = h0; c1.cm = cm; c1.cin_true = cin; c1.kh = c1.kw = 3; c1.pad = 1;
  c1.w = b.T(pre + "net.1.weight", (int64_t)cout * mult * cin * 9); c1.bias = b.T(pre + "net.1.bias", cout * mult);
  c1.cout = cout * mult; c1.Hout = H; c1.Wout = W;
  c1.gn = true; c1.gn_part = st0; c1.gn_nparts = slabs > 1 ? slabs : P; c1.gn_inv = 1.0f / ((float)cin * P);
  c1.gamma = b.T(pre + "net.0.weight", cin); c1.beta = b.T(pre + "net.0.bias", cin);
  c1.act = ACT_GELU; c1.want_stats = true;
  Builder::ConvResult r1 = b.emit_conv(c1);
  b.ws_free(st0); b.free_act(h0);
  // GN -> conv3x3 (+ res)  (net.3, net.4, res_conv)
  Builder::ConvSpec c2;
  c2.name = pre + "net.4"; c2.a = r1.out; c2.cm = ChanMap::identity(cout * mult); c2.cin_true = cout * mult; c2.kh = c2.kw = 3; c2.pad = 1;
  c2.w = b.T(pre + "net.4.weight", (int64_t)cout * cout * mult * 9); c2.bias = b.T(pre + "net.4.bias", cout);
  c2.cout = cout; c2.Hout = H; c2.Wout = W;
  c2.gn = true; c2.gn_part = r1.stat_off; c2.gn_nparts = r1.stat_nparts; c2.gn_inv = 1.0f / ((float)cout * mult * P);
  c2.gamma = b.T(pre + "net.3.weight", cout * mult); c2.beta = b.T(pre + "net.3.bias", cout * mult);
  if (cin != cout) {
    c2.has2 = true; c2.a2 = x; c2.cm2 = cm; c2.cin2 = cin;
    c2.w2 = b.T(pre + "res_conv.weight", (int64_t)cout * cin); c2.bias2 = b.T(pre + "res_conv.bias", cout);
  } else {
    if (!cm.is_identity()) throw std::runtime_error("plan: identity residual over a padded concat");
    c2.has_res = true; c2.r = x;
  }
  c2.want_stats = want_out_stats; c2.dest = dest;
  Builder::ConvResult r2 = b.emit_conv(c2);
  b.ws_free(r1.stat_off); b.free_act(r1.out);
  BlockIO io; io.out = r2.out; io.stat_off = r2.stat_off; io.stat_nparts = r2.stat_nparts;
  return io;
}

// Residual(PreNorm(c, LinearAttention(c))) / Residual(PreNorm(c, Attention(c)))  (unet_model.py:126-187)
// x must carry GroupNorm partials (stat_off).  Result written to *dest (or a fresh tensor).
View emit_attention(Builder& b, const std::string& pre, const View& x, int c, int64_t x_stat, int x_nparts, bool linear, View* dest) {
  const int H = x.H, W = x.W, P = H * W, hid = 128;
  if (P == 1 && !std::getenv("SBM_NO_ATTN_FOLD")) {
    // A single position: softmax over one element is 1, so both attention forms collapse to a linear map of v --
    //   LinearAttention (unet_model.py:169-175): ctx[d][e] = v[e], out[e] = scale * v[e] * sum_d softmax_d(q)[d] = scale * v[e]
    //   Attention       (unet_model.py:141-148): softmax over the single key = 1, out = v
    // and to_qkv -> core -> to_out become ONE GEMM with W' = s * W_out * W_v (formed in double).  The only difference from the
    // reference's arithmetic is that its sum of a softmax is 1 to within an ulp.
    const float* wqkv = b.T(pre + "fn.fn.to_qkv.weight", (int64_t)3 * hid * c);
    const std::string on = linear ? "fn.fn.to_out.0" : "fn.fn.to_out";
    const float* wout = b.T(pre + on + ".weight", (int64_t)c * hid);
    const double sc = linear ? 1.0 / std::sqrt(32.0) : 1.0;
    std::vector<float> wf((size_t)c * c);
    for (int o = 0; o < c; ++o)
      for (int i = 0; i < c; ++i) {
        double acc = 0.0;
        for (int k = 0; k < hid; ++k) acc += (double)wout[(size_t)o * hid + k] * (double)wqkv[(size_t)(2 * hid + k) * c + i];
        wf[(size_t)o * c + i] = (float)(sc * acc);
      }
    Builder::ConvSpec f;
    f.name = pre + "fn.fn (to_qkv.v x to_out folded: one position)"; f.a = x; f.cm = ChanMap::identity(c); f.cin_true = c;
    f.w = wf.data(); f.bias = b.T(pre + on + ".bias", c); f.cout = c; f.Hout = H; f.Wout = W;
    f.gn = true; f.gn_part = x_stat; f.gn_nparts = x_nparts; f.gn_inv = 1.0f / ((float)c * P);
    f.gamma = b.T(pre + "fn.norm.weight", c); f.beta = b.T(pre + "fn.norm.bias", c);
    // executed 2*c*(3*hid) + attention + 2*hid*c per sample in the dense convention; account the reference's figure
    const double dense_ref = 2.0 * b.B * ((double)c * 3 * hid + (double)hid * c) + 2.0 * b.B * 4 * (linear ? 2.0 * 32 * 32 : 2.0 * 32);
    if (linear) {
      f.want_stats = true;
      Builder::ConvResult ro = b.emit_conv(f);
      b.plan->ops.back().flops_dense = 2.0 * b.B * ((double)c * 3 * hid + (double)hid * c);
      View out = dest ? *dest : b.new_act(H, W, c);
      Op& op = b.push(OP_GN_RES, pre + "fn.fn.to_out.1+residual");
      if (b.chain_pix > 0 && P <= b.chain_pix) op.chain = 1;
      if (b.chain_pix > 0 && b.row_pix > 0 && P <= b.row_pix) { op.row = 1; op.chain = 0; }
      GnResOp& g = op.gr; g.B = b.B; g.u = ro.out; g.x = x; g.out = out;
      g.part_off = ro.stat_off; g.nparts = ro.stat_nparts; g.inv_count = 1.0f / ((float)c * P);
      g.gamma_off = b.add_params(b.pack_vec(b.T(pre + "fn.fn.to_out.1.weight", c), c, nullptr, x.C));
      g.beta_off = b.add_params(b.pack_vec(b.T(pre + "fn.fn.to_out.1.bias", c), c, nullptr, x.C));
      b.ws_free(ro.stat_off); b.free_act(ro.out);
      b.plan->flops_dense += dense_ref - 2.0 * b.B * (double)c * c; b.plan->flops_dense_gemm += 2.0 * b.B * ((double)c * 3 * hid + (double)hid * c) - 2.0 * b.B * (double)c * c;
      return out;
    }
    f.has_res = true; f.r = x; f.dest = dest;
    View out = b.emit_conv(f).out;
    b.plan->ops.back().flops_dense = 2.0 * b.B * ((double)c * 3 * hid + (double)hid * c);
    b.plan->flops_dense += dense_ref - 2.0 * b.B * (double)c * c; b.plan->flops_dense_gemm += 2.0 * b.B * ((double)c * 3 * hid + (double)hid * c) - 2.0 * b.B * (double)c * c;
    return out;
  }
  // pixel mode: the sample-owned kernel's attention core takes at most 16 positions; above that to_qkv + core stay the fused per-layer kernel
  const int attn_saved_chain_pix = b.chain_pix;
  if (b.pix_mode && P > 16) b.chain_pix = 0;
  Builder::ConvSpec q;
  q.name = pre + "fn.fn.to_qkv"; q.a = x; q.cm = ChanMap::identity(c); q.cin_true = c;
  q.w = b.T(pre + "fn.fn.to_qkv.weight", (int64_t)3 * hid * c); q.cout = 3 * hid; q.Hout = H; q.Wout = W;
  q.gn = true; q.gn_part = x_stat; q.gn_nparts = x_nparts; q.gn_inv = 1.0f / ((float)c * P);
  q.gamma = b.T(pre + "fn.norm.weight", c); q.beta = b.T(pre + "fn.norm.bias", c);
  Builder::ConvResult rq = b.emit_conv(q);
  View att = b.new_act(H, W, hid);
  {
    Op& op = b.push(linear ? OP_LINATTN : OP_ATTN, pre + "fn.fn.core");
    if (b.chain_pix > 0 && P <= b.chain_pix) op.chain = 1;
    AttnOp& a = op.at; a.B = b.B; a.qkv = rq.out; a.out = att; a.heads = 4; a.dh = 32; a.scale = 1.0f / std::sqrt(32.0f);
    b.plan->flops_dense += 2.0 * b.B * 4 * (linear ? 2.0 * 32 * 32 * P : 2.0 * 32 * (double)P * P);
    b.plan->flops_exec += 2.0 * b.B * 4 * (linear ? 2.0 * 32 * 32 * P : 2.0 * 32 * (double)P * P);
  }
  b.free_act(rq.out);
  // ... and an 8x8 block of 64 channels and 4 heads runs WHOLE in one kernel (kernels.hip attnblock_kernel; marked at model creation by
  // attn_block_fusable): its to_out.0 and GroupNorm + residual then stay out of the pixel kernel's run too
  const bool whole_block = b.pix_mode && linear && P == 64 && c == 64 && hid == 128 && attn_block_on_impl();
  if (!whole_block) b.chain_pix = attn_saved_chain_pix;
  View out;
  if (linear) {
    Builder::ConvSpec o;
    o.name = pre + "fn.fn.to_out.0"; o.a = att; o.cm = ChanMap::identity(hid); o.cin_true = hid;
    o.w = b.T(pre + "fn.fn.to_out.0.weight", (int64_t)c * hid); o.bias = b.T(pre + "fn.fn.to_out.0.bias", c);
    o.cout = c; o.Hout = H; o.Wout = W; o.want_stats = true;
    Builder::ConvResult ro = b.emit_conv(o);
    out = dest ? *dest : b.new_act(H, W, c);
    Op& op = b.push(OP_GN_RES, pre + "fn.fn.to_out.1+residual");
    if (b.chain_pix > 0 && P <= b.chain_pix) op.chain = 1;
    if (b.chain_pix > 0 && b.row_pix > 0 && P <= b.row_pix) { op.row = 1; op.chain = 0; }
    GnResOp& g = op.gr; g.B = b.B; g.u = ro.out; g.x = x; g.out = out;
    g.part_off = ro.stat_off; g.nparts = ro.stat_nparts; g.inv_count = 1.0f / ((float)c * P);
    g.gamma_off = b.add_params(b.pack_vec(b.T(pre + "fn.fn.to_out.1.weight", c), c, nullptr, x.C));
    g.beta_off = b.add_params(b.pack_vec(b.T(pre + "fn.fn.to_out.1.bias", c), c, nullptr, x.C));
    b.ws_free(ro.stat_off); b.free_act(ro.out);
    b.chain_pix = attn_saved_chain_pix;
  } else {
    Builder::ConvSpec o;
    o.name = pre + "fn.fn.to_out"; o.a = att; o.cm = ChanMap::identity(hid); o.cin_true = hid;
    o.w = b.T(pre + "fn.fn.to_out.weight", (int64_t)c * hid); o.bias = b.T(pre + "fn.fn.to_out.bias", c);
    o.cout = c; o.Hout = H; o.Wout = W; o.has_res = true; o.r = x; o.dest = dest;
    out = b.emit_conv(o).out;
  }
  b.free_act(att);
  return out;
}

int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace

static thread_local int g_plan_pix_mode = 0;
void plan_set_pix_mode(int mode) { g_plan_pix_mode = mode; }
bool attn_block_on() { return attn_block_on_impl(); }
int plan_pix_mode() { return g_plan_pix_mode; }

int chain_pix_from_env(int B) {
  // The chain kernel gives each XCD (32 CUs) one 32-sample tile at a time: with fewer than ~6 tiles most of the chip idles
  // and per-layer launches are faster.
  const char* mb = std::getenv("SBM_CHAIN_MIN_B");
  const int min_b = mb ? std::atoi(mb) : 192;
  if (B < min_b) return 0;
  const char* e = std::getenv("SBM_CHAIN_PIX");
  const int v = e ? std::atoi(e) : 16;
  return v < 0 ? 0 : (v > 64 ? 64 : v);
}

int row_pix_from_env() {
  const char* e = std::getenv("SBM_ROW_PIX");
  const int v = e ? std::atoi(e) : 1;
  return v <= 0 ? 0 : 1;              // the row kernel carries the one-position level only
}

bool build_unet_plan(const UnetCfg& cfg, int B, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = unet_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    // Levels of SBM_CHAIN_PIX pixels and fewer.  Pixel mode (default, api.hip tries it first; default 64 = the 8x8 level and below): they
    // are marked for the sample-owned kernel at EVERY batch size, the one-position level included; the time path, the 8x8 attention core
    // with its to_qkv (a fused kernel of their own) and init_conv's row-packed form stay per-layer.  Legacy mode (default 16): XCD-resident
    // chain kernel for B >= SBM_CHAIN_MIN_B plus the row kernel for the one-position level.
    b.pix_mode = plan_pix_mode() == 1;
    if (b.pix_mode) { const char* e = std::getenv("SBM_CHAIN_PIX"); const int v = e ? std::atoi(e) : 64; b.chain_pix = v < 0 ? 0 : (v > 64 ? 64 : v); b.row_pix = 0; }
    else { b.chain_pix = chain_pix_from_env(B); b.row_pix = b.chain_pix > 0 ? row_pix_from_env() : 0; }
    const int level_chain_pix = b.chain_pix;
    if (b.pix_mode) b.chain_pix = 0;                                 // while the time path is emitted
    UnetDims d = unet_dims(cfg);
    const int n = (int)d.in_out.size();
    const int Hp = next_pow2(cfg.H), Wp = next_pow2(cfg.W);       // unet_model.py:276-284
    const int pad_y = (Hp - cfg.H) / 2, pad_x = (Wp - cfg.W) / 2;
    if ((Hp >> (n - 1)) < 1 || (Wp >> (n - 1)) < 1) throw std::runtime_error("plan: input too small for the UNet depth");
    plan->in_C = cfg.channels; plan->in_H = cfg.H; plan->in_W = cfg.W;
    plan->out_C = d.out_dim; plan->out_H = cfg.H; plan->out_W = cfg.W;
    plan->needs_time = d.td ? 1 : 0;

    // external NCHW input -> PBC
    View xin = b.new_input_act(Hp, Wp, cfg.channels);
    { Op& op = b.push(OP_NCHW_TO_PBC, "input"); op.cv.B = B; op.cv.C = cfg.channels; op.cv.Hn = cfg.H; op.cv.Wn = cfg.W; op.cv.pad_y = pad_y; op.cv.pad_x = pad_x; op.cv.out = xin; }

    // time path: sinusoid -> Linear -> GELU -> Linear (-> GELU, shared first op of every block mlp) -> all block Linears
    struct TB { std::string pre; int width; };      // width = channel count of the block's input view
    std::vector<TB> tblocks;
    if (d.td) {
      for (int i = 0; i < n; ++i) { tblocks.push_back({"downs." + std::to_string(i) + ".0.", pad8(d.in_out[i].first)}); tblocks.push_back({"downs." + std::to_string(i) + ".1.", pad8(d.in_out[i].second)}); }
      tblocks.push_back({"mid_block1.", pad8(d.dims.back())}); tblocks.push_back({"mid_block2.", pad8(d.dims.back())});
      for (int j = 0; j < n - 1; ++j) { auto [ci, co] = d.in_out[n - 1 - j]; tblocks.push_back({"ups." + std::to_string(j) + ".0.", 2 * pad8(co)}); tblocks.push_back({"ups." + std::to_string(j) + ".1.", pad8(ci)}); }
    }
    std::map<std::string, int> cond_col;
    int cond_ld = 0;
    int64_t cond_off = 0;
    View cond_all;
    if (d.td) {
      plan->time_off = b.ws_alloc(B);
      View e0 = b.new_act(1, 1, cfg.dim);
      {
        // unet_model.py:42-44: freqs = exp(arange(half) * -(log(1e4)/(half-1))) in fp32
        Op& op = b.push(OP_SINUS, "time_mlp.0"); op.si.B = B; op.si.dim = cfg.dim; op.si.out_off = e0.off; op.si.ld = e0.ld;
        const int half = cfg.dim / 2;
        std::vector<float> fr(half);
        const float k = (float)(-(std::log(10000.0) / (half - 1)));
        for (int j = 0; j < half; ++j) { float arg = (float)j * k; fr[j] = (float)std::exp((double)arg); }
        op.si.freq_off = b.add_params(fr);
      }
      Builder::ConvSpec l1; l1.name = "time_mlp.1"; l1.a = e0; l1.cm = ChanMap::identity(cfg.dim); l1.cin_true = cfg.dim;
      l1.w = b.T("time_mlp.1.weight", (int64_t)d.td * cfg.dim); l1.bias = b.T("time_mlp.1.bias", d.td); l1.cout = d.td; l1.act = ACT_GELU;
      View e1 = b.emit_conv(l1).out;
      Builder::ConvSpec l2; l2.name = "time_mlp.3(+GELU of every block mlp)"; l2.a = e1; l2.cm = ChanMap::identity(d.td); l2.cin_true = d.td;
      l2.w = b.T("time_mlp.3.weight", (int64_t)d.td * d.td); l2.bias = b.T("time_mlp.3.bias", d.td); l2.cout = d.td; l2.act = ACT_GELU;
      View e2 = b.emit_conv(l2).out;
      b.free_act(e0); b.free_act(e1);
      // every ConvNextBlock.mlp Linear as one GEMM; column ranges follow the channel layout of each block's input
      int col = 0;
      for (auto& tb : tblocks) { cond_col[tb.pre] = col; col += tb.width; }
      cond_ld = round_up(col, 8);
      // weights are filled in during the walk (they need each block's channel map) and attached at the end
      cond_all = b.new_act(1, 1, cond_ld);
      cond_off = cond_all.off;
      // placeholder op; weights patched in after the walk (needs the concat channel maps)
      Op& op = b.push(OP_GEMM, "convnext mlp.1 (all blocks)");
      GemmOp& g = op.g; g.B = B; g.a = e2; g.N = cond_ld; g.Npad = round_up(cond_ld, 32); g.out = cond_all;
      Builder::choose_tile(B, 1, cond_ld, &g.BM, &g.BN);
      op.flops = 2.0 * B * g.Npad * d.td;
      plan->flops_exec += op.flops;
      b.free_act(e2);   // ops run in stream order: e2 is dead once the GEMM above has executed
      plan->cond_off = cond_off; plan->cond_ld = cond_ld;
      for (Op& o : plan->ops) if (o.kind != OP_NCHW_TO_PBC) { o.time_path = 1; if (o.row) { o.row = 0; o.chain = 1; } }      // everything emitted so far except the input conversion (the time MLP stays in the chain kernel)
    }
    b.chain_pix = level_chain_pix;
    const size_t cond_op_index = plan->ops.size() - 1;
    std::vector<float> cond_w, cond_b;
    const int td_p = round_up(d.td, 32);
    if (d.td) { cond_w.assign((size_t)round_up(cond_ld, 32) * td_p, 0.f); cond_b.assign(round_up(cond_ld, 32), 0.f); }
    auto add_cond = [&](const std::string& pre, int cin, const ChanMap& cm) {
      if (!d.td) return;
      const float* w = b.T(pre + "mlp.1.weight", (int64_t)cin * d.td);
      const float* bb = b.T(pre + "mlp.1.bias", cin);
      const int col = cond_col.at(pre);
      for (int c = 0; c < cin; ++c) {
        std::memcpy(&cond_w[(size_t)(col + cm.idx[c]) * td_p], &w[(size_t)c * d.td], sizeof(float) * d.td);
        cond_b[col + cm.idx[c]] = bb[c];
      }
      plan->flops_dense += 2.0 * B * cin * d.td; plan->flops_dense_gemm += 2.0 * B * cin * d.td; plan->flops_dense_cond += 2.0 * B * cin * d.td;
    };

    // init conv 7x7.  With a handful of input channels a tap is 8 floats of a 32-float K chunk (3/4 of the MFMA work would
    // multiply padding), so the input is stored row-packed -- one "pixel" per image ROW with Wp*8 channels -- and the
    // horizontal window moves into per-output-pixel weights: 7 row taps x Wp*8/32 chunks instead of 49 taps x 1 chunk.
    View x;
    const int rowK = round_up(Wp * 8, 32);
    const bool rowpack = cfg.channels <= 8 && !std::getenv("SBM_NO_ROWPACK") &&
                         (int64_t)Hp * Wp * 7 * round_up(pad8(d.init_dim), 32) * rowK * 4 <= (16LL << 20);
    // pixel mode, 8x8 inputs: init_conv INSIDE the first run of the sample-owned kernel -- the row-packed input is expanded in LDS to the
    // [64 pixels][C * 49] matrix and the 7x7 conv is a 1x1 conv over it (K = 245 -> 256 for 5 modalities: 8 K steps instead of a launch)
    const bool im2col = rowpack && b.pix_mode && pix_io_on() && Hp == 8 && Wp == 8 && b.chain_pix >= 64 && cfg.channels * 49 <= 256;
    if (rowpack) {
      Op& cv = plan->ops[0];                                   // the input conversion emitted above
      View rows; rows.H = Hp; rows.W = 1; rows.C = Wp * 8; rows.ld = rowK;
      b.free_act(xin);
      rows.off = b.ws_alloc((int64_t)Hp * B * rows.ld, true);     // pinned: the sampler's update kernels keep it current between forwards (sampler.hip store_pbc), so nothing else may recycle it
      cv.cv.out = rows; cv.cv.rowpack = 1; cv.cv.Wo = Wp; cv.cv.Cpad = 8;
      xin = rows;
    }
    if (im2col) {
      const int K = cfg.channels * 49;
      View col; col.H = Hp; col.W = Wp; col.C = pad8(K); col.ld = 256;      // the expansion's thread mapping (pixrun.hip OP_IM2COL): 64 lanes x 4 columns
      col.off = b.ws_alloc((int64_t)Hp * Wp * B * col.ld, true);
      Op& op = b.push(OP_IM2COL, "init_conv im2col");
      op.chain = 1;
      Im2colOp& ic = op.ic; ic.B = B; ic.in = xin; ic.out = col; ic.C = cfg.channels; ic.Cpad = 8; ic.H = Hp; ic.W = Wp; ic.KH = ic.KW = 7; ic.pad = 3;
      Builder::ConvSpec cs; cs.name = "init_conv (1x1 over im2col)"; cs.a = col; cs.cm = ChanMap::identity(K); cs.cin_true = K;
      cs.w = b.T("init_conv.weight", (int64_t)d.init_dim * K); cs.bias = b.T("init_conv.bias", d.init_dim);
      cs.cout = d.init_dim; cs.Hout = Hp; cs.Wout = Wp;
      x = b.emit_conv(cs).out;
      b.ws_free(col.off);
    } else if (rowpack) {
      const float* w = b.T("init_conv.weight", (int64_t)d.init_dim * cfg.channels * 49); const float* bi = b.T("init_conv.bias", d.init_dim);
      const int P = Hp * Wp, cout = d.init_dim;
      x = b.new_act(Hp, Wp, cout);
      Op& op = b.push(OP_GEMM, "init_conv (row-packed)");
      GemmOp& g = op.g; g.B = B; g.Hout = Hp; g.Wout = Wp; g.a = xin; g.a_rows = 1; g.KH = 7; g.KW = 1; g.stride = 1; g.pad = 3;
      g.N = cout; g.Npad = round_up(pad8(cout), 32); g.Kp = rowK; g.out = x;
      std::vector<float> wp((size_t)P * 7 * g.Npad * g.Kp, 0.f), bp(g.Npad, 0.f);
      for (int pp = 0; pp < P; ++pp) {
        const int ox = pp % Wp;
        for (int ky = 0; ky < 7; ++ky) for (int n = 0; n < cout; ++n) for (int xs = 0; xs < Wp; ++xs) {
          const int kx = xs - ox + 3;
          if (kx < 0 || kx >= 7) continue;
          for (int c = 0; c < cfg.channels; ++c)
            wp[(((size_t)pp * 7 + ky) * g.Npad + n) * g.Kp + xs * 8 + c] = w[(((size_t)n * cfg.channels + c) * 7 + ky) * 7 + kx];
        }
      }
      for (int n = 0; n < cout; ++n) bp[n] = bi[n];
      g.w_off = b.add_params(wp); g.w_pix_stride = (int64_t)7 * g.Npad * g.Kp; g.bias_off = b.add_params(bp);
      Builder::choose_tile(B, P, pad8(cout), &g.BM, &g.BN);
      const double dense = 2.0 * B * cout * ((double)P * 49 * cfg.channels);
      plan->flops_dense += dense; plan->flops_dense_gemm += dense; op.flops_dense = dense;
      double rows_valid = 0;
      for (int oy = 0; oy < Hp; ++oy) for (int ky = 0; ky < 7; ++ky) rows_valid += (oy - 3 + ky >= 0 && oy - 3 + ky < Hp) ? Wp : 0;
      op.flops = 2.0 * B * g.Npad * rows_valid * g.Kp;
      plan->flops_exec += op.flops;
    } else {
      Builder::ConvSpec ic; ic.name = "init_conv"; ic.a = xin; ic.cm = ChanMap::identity(cfg.channels); ic.cin_true = cfg.channels;
      ic.kh = ic.kw = 7; ic.pad = 3; ic.w = b.T("init_conv.weight", (int64_t)d.init_dim * cfg.channels * 49); ic.bias = b.T("init_conv.bias", d.init_dim);
      ic.cout = d.init_dim; ic.Hout = Hp; ic.Wout = Wp;
      x = b.emit_conv(ic).out;
    }
    b.free_act(xin);

    // down path.  Level i>=1 writes its skip tensor straight into the second half of the concat
    // buffer its up level will read (torch.cat((x, h.pop()), dim=1), unet_model.py:311).
    std::vector<View> cat(n);       // cat[i]: concat buffer for the up level that pops h_i  (i >= 1)
    std::vector<View> skip(n);
    int H = Hp, W = Wp;
    for (int i = 0; i < n; ++i) {
      auto [ci, co] = d.in_out[i];
      std::string p = "downs." + std::to_string(i) + ".";
      ChanMap cmi = ChanMap::identity(ci);
      add_cond(p + "0.", ci, cmi);
      BlockIO b1 = emit_convnext(b, p + "0.", x, cmi, ci, co, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col[p + "0."] : 0), cond_ld, false, nullptr);
      b.free_act(x);
      ChanMap cmo = ChanMap::identity(co);
      add_cond(p + "1.", co, cmo);
      BlockIO b2 = emit_convnext(b, p + "1.", b1.out, cmo, co, co, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col[p + "1."] : 0), cond_ld, true, nullptr);
      b.free_act(b1.out);
      View dest; View* dp = nullptr;
      if (i >= 1) {
        // concat buffer [x_up (co) | h_i (co)], both padded to 8
        View cb; cb.H = H; cb.W = W; cb.C = 2 * pad8(co); cb.ld = cb.C; cb.off = b.ws_alloc((int64_t)H * W * B * cb.ld, b.in_chain(H * W));
        cat[i] = cb;
        dest = cb; dest.off = cb.off + pad8(co); dest.C = pad8(co);
        dp = &dest;
      }
      View y = emit_attention(b, p + "2.", b2.out, co, b2.stat_off, b2.stat_nparts, true, dp);
      b.ws_free(b2.stat_off); b.free_act(b2.out);
      skip[i] = y;
      if (i < n - 1) {
        Builder::ConvSpec ds; ds.name = p + "3"; ds.a = y; ds.cm = ChanMap::identity(co); ds.cin_true = co; ds.kh = ds.kw = 4; ds.stride = 2; ds.pad = 1;
        ds.w = b.T(p + "3.weight", (int64_t)co * co * 16); ds.bias = b.T(p + "3.bias", co); ds.cout = co; ds.Hout = H / 2; ds.Wout = W / 2;
        x = b.emit_conv(ds).out;
        H /= 2; W /= 2;
        if (i == 0) b.free_act(y);      // h_0 is never popped (unet_model.py:244-262: only n-1 up levels)
      } else {
        x = y;                          // last level: Identity instead of Downsample; x aliases the concat half
      }
    }
    // bottleneck.  x is skip[n-1], which lives in cat[n-1]'s second half when n >= 2.
    const int mid = d.dims.back();
    ChanMap cmm = ChanMap::identity(mid);
    add_cond("mid_block1.", mid, cmm);
    BlockIO m1 = emit_convnext(b, "mid_block1.", x, cmm, mid, mid, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col["mid_block1."] : 0), cond_ld, true, nullptr);
    if (n == 1) b.free_act(x);
    View ma = emit_attention(b, "mid_attn.", m1.out, mid, m1.stat_off, m1.stat_nparts, false, nullptr);
    b.ws_free(m1.stat_off); b.free_act(m1.out);
    add_cond("mid_block2.", mid, cmm);
    View up_dest; View* udp = nullptr;
    if (n >= 2) { up_dest = cat[n - 1]; up_dest.C = pad8(mid); udp = &up_dest; }
    BlockIO m2 = emit_convnext(b, "mid_block2.", ma, cmm, mid, mid, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col["mid_block2."] : 0), cond_ld, false, udp);
    b.free_act(ma);
    x = m2.out;
    // up path
    for (int j = 0; j < n - 1; ++j) {
      auto [ci, co] = d.in_out[n - 1 - j];
      std::string p = "ups." + std::to_string(j) + ".";
      View cb = cat[n - 1 - j];
      ChanMap cmc = ChanMap::concat(co, pad8(co), co);
      add_cond(p + "0.", co * 2, cmc);
      BlockIO b1 = emit_convnext(b, p + "0.", cb, cmc, co * 2, ci, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col[p + "0."] : 0), cond_ld, false, nullptr);
      b.ws_free(cb.off);
      ChanMap cmi = ChanMap::identity(ci);
      add_cond(p + "1.", ci, cmi);
      BlockIO b2 = emit_convnext(b, p + "1.", b1.out, cmi, ci, ci, cfg.convnext_mult, d.td != 0, cond_off + (d.td ? cond_col[p + "1."] : 0), cond_ld, true, nullptr);
      b.free_act(b1.out);
      View y = emit_attention(b, p + "2.", b2.out, ci, b2.stat_off, b2.stat_nparts, true, nullptr);
      b.ws_free(b2.stat_off); b.free_act(b2.out);
      // Upsample = ConvTranspose2d(ci, ci, 4, 2, 1) -> first half of the next concat buffer (or a plain tensor)
      Builder::ConvSpec us; us.name = p + "3"; us.a = y; us.cm = ChanMap::identity(ci); us.cin_true = ci; us.kh = us.kw = 4; us.stride = 2; us.pad = 1; us.transposed = true;
      us.w = b.T(p + "3.weight", (int64_t)ci * ci * 16); us.bias = b.T(p + "3.bias", ci); us.cout = ci; us.Hout = H * 2; us.Wout = W * 2;
      View ud;
      if (j + 1 < n - 1) { ud = cat[n - 2 - j]; ud.C = pad8(ci); us.dest = &ud; }
      x = b.emit_conv(us).out;
      b.free_act(y);
      H *= 2; W *= 2;
    }
    // final_conv = ConvNextBlock(dim, dim) (no time) + 1x1 conv -> NCHW score (unet_model.py:270-272)
    if (x.C != pad8(cfg.dim)) throw std::runtime_error("plan: final_conv expects `dim` channels (dim_mults[0] must be 1)");
    BlockIO f = emit_convnext(b, "final_conv.0.", x, ChanMap::identity(cfg.dim), cfg.dim, cfg.dim, cfg.convnext_mult, false, 0, 0, false, nullptr);
    b.free_act(x);
    Builder::ConvSpec fc; fc.name = "final_conv.1"; fc.a = f.out; fc.cm = ChanMap::identity(cfg.dim); fc.cin_true = cfg.dim;
    fc.w = b.T("final_conv.1.weight", (int64_t)d.out_dim * cfg.dim); fc.bias = b.T("final_conv.1.bias", d.out_dim);
    fc.cout = d.out_dim; fc.Hout = H; fc.Wout = W; fc.out_nchw = true; fc.crop_y = pad_y; fc.crop_x = pad_x; fc.Hn = cfg.H; fc.Wn = cfg.W;
    b.emit_conv(fc);
    b.free_act(f.out);
    // patch the concatenated mlp weights
    if (d.td) {
      GemmOp& g = plan->ops[cond_op_index].g;
      g.w_off = b.add_params(cond_w); g.Kp = td_p;
      g.bias_off = b.add_params(cond_b);
    }
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

void finalize_gemm_tables(Plan* plan) {
  int64_t max_part = 0, max_tiles = 0;
  auto add_ints = [&](const std::vector<int>& v) {
    const int64_t off = (int64_t)plan->params.size();
    plan->params.resize(plan->params.size() + v.size());
    std::memcpy(plan->params.data() + off, v.data(), v.size() * sizeof(int));
    while (plan->params.size() % 4) plan->params.push_back(0.f);
    return off;
  };
  {
    const int64_t off = (int64_t)plan->params.size();
    for (int i = 0; i < 32; ++i) plan->params.push_back(1.f);
    for (int i = 0; i < 32; ++i) plan->params.push_back(0.f);
    plan->ident_off = off;
  }
  auto chk = [](int64_t v) { if (v < 0 || v > 0x7fffffffLL) throw std::runtime_error("plan: operand offset does not fit 32 bits"); return (int)v; };
  for (Op& op : plan->ops) {
    if (op.kind != OP_GEMM) continue;
    GemmOp& g = op.g;
    const int P = g.Hout * g.Wout;
    std::vector<int> hdr((size_t)4 * P), desc;
    for (int p = 0; p < P; ++p) {
      const int oy = p / g.Wout, ox = p % g.Wout;
      const int first = (int)(desc.size() / 4);
      int n1 = 0;
      for (int ky = 0; ky < g.KH; ++ky) for (int kx = 0; kx < g.KW; ++kx) {
        int q;
        if (g.a_rows) { const int iy = oy * g.stride - g.pad + ky; q = (iy >= 0 && iy < g.a.H) ? iy : -1; }
        else q = g.a_bcast ? 0 : tap_source_pixel(oy, ox, ky, kx, g.stride, g.pad, g.transposed, g.up, g.a.H, g.a.W);
        if (q < 0) continue;
        for (int kc = 0; kc < g.Kp; kc += 32) {
          desc.push_back(chk((int64_t)q * g.B * g.a.ld + kc));
          desc.push_back(chk((int64_t)p * g.w_pix_stride + (int64_t)(ky * g.KW + kx) * g.Npad * g.Kp + kc));
          desc.push_back(kc);
          desc.push_back(0);
          ++n1;
        }
      }
      const int w2 = g.has2 ? g.a2.W : (g.has_res ? g.r.W : 1);
      const int p2 = (oy / g.up2) * w2 + (ox / g.up2);
      if (g.has2)
        for (int kc = 0; kc < g.Kp2; kc += 32) {
          desc.push_back(chk((int64_t)p2 * g.B * g.a2.ld + kc)); desc.push_back(kc); desc.push_back(-1); desc.push_back(1);
        }
      hdr[4 * p] = first; hdr[4 * p + 1] = (int)(desc.size() / 4) - first; hdr[4 * p + 2] = n1; hdr[4 * p + 3] = p2;
    }
    if (desc.empty()) desc.assign(4, 0);
    g.hdr_off = add_ints(hdr);
    g.desc_off = add_ints(desc);
    // split-K for grids that leave most of the 256 CUs idle (few pixels x few tiles, long K): aim at two workgroups per CU,
    // keep at least SPLIT_MIN_CHUNKS chunks per split so the DMA ring still fills
    if (!op.chain && !op.row && !g.out_nchw) {
      int minc = 1 << 30;
      for (int p = 0; p < P; ++p) minc = std::min(minc, hdr[4 * p + 1]);
      const long tiles = (long)P * ceil_div(g.B, g.BM) * ceil_div(g.out.C, g.BN);
      const int S = Builder::splitk_for(tiles, minc);
      if (getenv("SBM_NAME_TILES")) op.name += " [" + std::to_string(g.BM) + "x" + std::to_string(g.BN) + " s" + std::to_string(S) + " k" + std::to_string(minc) + "]";
      if (S >= 2) {
        g.split = S;
        max_part = std::max<int64_t>(max_part, (int64_t)tiles * S * g.BM * g.BN);
        max_tiles = std::max<int64_t>(max_tiles, tiles);
      }
    }
  }
  if (getenv("SBM_DUMP_PLAN")) {                       // diagnostics: one line per op
    int i = 0;
    for (const Op& op : plan->ops) {
      int P = 0, C = 0, N = 0;
      if (op.kind == OP_GEMM) { P = op.g.Hout * op.g.Wout; C = op.g.a.C; N = op.g.out.C; }
      else if (op.kind == OP_DWCONV7) { P = op.dw.in.H * op.dw.in.W; C = op.dw.in.C; }
      else if (op.kind == OP_LINATTN || op.kind == OP_ATTN) { P = op.at.qkv.H * op.at.qkv.W; C = op.at.qkv.C; N = op.at.out.C; }
      else if (op.kind == OP_GN_RES) { P = op.gr.u.H * op.gr.u.W; C = op.gr.u.C; }
      fprintf(stderr, "op %3d kind %2d %s%s%s P %3d C %4d N %4d  %s\n", i++, op.kind, op.chain ? "chain " : "", op.row ? "row " : "", op.time_path ? "time " : "", P, C, N, op.name.c_str());
    }
  }
  if (max_tiles > 0) {
    plan->splitk_part_off = plan->ws_floats; plan->ws_floats += (max_part + 63) / 64 * 64;
    plan->splitk_cnt_off = plan->ws_floats; plan->ws_floats += (max_tiles + 63) / 64 * 64;
  }
}

// ------------------------------------------------------------------------------------------------
// unet_openai.UNetModel
// ------------------------------------------------------------------------------------------------
namespace {

struct OALayer { int kind; int cin; int cout; };     // 0 conv, 1 res, 2 attn, 3 down, 4 up
struct OAStruct { std::vector<std::vector<OALayer>> inputs, outputs; std::vector<OALayer> middle; std::vector<int> chans; };

OAStruct oa_structure(const UnetOACfg& c) {
  OAStruct S;
  auto in_attn = [&](int ds) { for (int i = 0; i < c.n_attn; ++i) if (c.attn_res[i] == ds) return true; return false; };
  const int mc = c.model_channels;
  S.inputs.push_back({{0, c.in_channels, mc}});
  std::vector<int> chans{mc};
  int ch = mc, ds = 1;
  for (int level = 0; level < c.n_mult; ++level) {
    const int mult = c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; ++r) {
      std::vector<OALayer> L{{1, ch, mult * mc}};
      ch = mult * mc;
      if (in_attn(ds)) L.push_back({2, ch, ch});
      S.inputs.push_back(L); chans.push_back(ch);
    }
    if (level != c.n_mult - 1) { S.inputs.push_back({{3, ch, ch}}); chans.push_back(ch); ds *= 2; }
  }
  S.chans = chans;
  S.middle = {{1, ch, ch}, {2, ch, ch}, {1, ch, ch}};
  for (int level = c.n_mult - 1; level >= 0; --level) {
    const int mult = c.channel_mult[level];
    for (int i = 0; i <= c.num_res_blocks; ++i) {
      const int skip = chans.back(); chans.pop_back();
      std::vector<OALayer> L{{1, ch + skip, mc * mult}};
      ch = mc * mult;
      if (in_attn(ds)) L.push_back({2, ch, ch});
      if (level && i == c.num_res_blocks) { L.push_back({4, ch, ch}); ds /= 2; }
      S.outputs.push_back(L);
    }
  }
  return S;
}

void oa_layer_specs(SpecList& L, const std::string& pre, const OALayer& l, int emb) {
  if (l.kind == 0) { L.add(pre + "weight", {l.cout, l.cin, 3, 3}); L.add(pre + "bias", {l.cout}); }
  else if (l.kind == 1) {
    L.add(pre + "in_layers.0.weight", {l.cin}); L.add(pre + "in_layers.0.bias", {l.cin});
    L.add(pre + "in_layers.2.weight", {l.cout, l.cin, 3, 3}); L.add(pre + "in_layers.2.bias", {l.cout});
    L.add(pre + "emb_layers.1.weight", {l.cout, emb}); L.add(pre + "emb_layers.1.bias", {l.cout});
    L.add(pre + "out_layers.0.weight", {l.cout}); L.add(pre + "out_layers.0.bias", {l.cout});
    L.add(pre + "out_layers.3.weight", {l.cout, l.cout, 3, 3}); L.add(pre + "out_layers.3.bias", {l.cout});
    if (l.cin != l.cout) { L.add(pre + "skip_connection.weight", {l.cout, l.cin, 1, 1}); L.add(pre + "skip_connection.bias", {l.cout}); }
  } else if (l.kind == 2) {
    L.add(pre + "norm.weight", {l.cin}); L.add(pre + "norm.bias", {l.cin});
    L.add(pre + "qkv.weight", {3 * l.cin, l.cin, 1}); L.add(pre + "qkv.bias", {3 * l.cin});
    L.add(pre + "proj_out.weight", {l.cin, l.cin, 1}); L.add(pre + "proj_out.bias", {l.cin});
  } else if (l.kind == 3) { L.add(pre + "op.weight", {l.cin, l.cin, 3, 3}); L.add(pre + "op.bias", {l.cin}); }
  else { L.add(pre + "conv.weight", {l.cin, l.cin, 3, 3}); L.add(pre + "conv.bias", {l.cin}); }
}

}  // namespace

std::vector<TensorSpec> unet_oa_tensor_specs(const UnetOACfg& c) {
  SpecList L;
  const int emb = c.model_channels * 4;
  L.add("time_embed.0.weight", {emb, c.model_channels}); L.add("time_embed.0.bias", {emb});
  L.add("time_embed.2.weight", {emb, emb}); L.add("time_embed.2.bias", {emb});
  if (c.use_z) {
    L.add("proj.0.weight", {emb, c.z_dim}); L.add("proj.0.bias", {emb});
    L.add("proj.2.weight", {emb, emb}); L.add("proj.2.bias", {emb});
  }
  OAStruct S = oa_structure(c);
  for (size_t b = 0; b < S.inputs.size(); ++b) for (size_t l = 0; l < S.inputs[b].size(); ++l)
    oa_layer_specs(L, "input_blocks." + std::to_string(b) + "." + std::to_string(l) + ".", S.inputs[b][l], emb);
  for (size_t l = 0; l < S.middle.size(); ++l) oa_layer_specs(L, "middle_block." + std::to_string(l) + ".", S.middle[l], emb);
  for (size_t b = 0; b < S.outputs.size(); ++b) for (size_t l = 0; l < S.outputs[b].size(); ++l)
    oa_layer_specs(L, "output_blocks." + std::to_string(b) + "." + std::to_string(l) + ".", S.outputs[b][l], emb);
  L.add("out.0.weight", {c.model_channels}); L.add("out.0.bias", {c.model_channels});
  L.add("out.2.weight", {c.out_channels, c.model_channels, 3, 3}); L.add("out.2.bias", {c.out_channels});
  return L.v;
}

bool build_unet_oa_plan(const UnetOACfg& cfg, int B, int with_z, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = unet_oa_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    if (with_z && !cfg.use_z) throw std::runtime_error("plan: z given but the model has no proj (use_z=False)");
    OAStruct S = oa_structure(cfg);
    const int mc = cfg.model_channels, emb = mc * 4;
    plan->in_C = cfg.in_channels; plan->in_H = cfg.H; plan->in_W = cfg.W;
    plan->out_C = cfg.out_channels; plan->out_H = cfg.H; plan->out_W = cfg.W;
    plan->needs_time = 1;
    for (int v : {mc}) if (v % 32) throw std::runtime_error("plan: UNetModel channel counts must be multiples of 32 (GroupNorm32)");

    View xin = b.new_input_act(cfg.H, cfg.W, cfg.in_channels);
    { Op& op = b.push(OP_NCHW_TO_PBC, "input"); op.cv.B = B; op.cv.C = cfg.in_channels; op.cv.Hn = cfg.H; op.cv.Wn = cfg.W; op.cv.out = xin; }

    // ---- emb = time_embed(timestep_embedding(t)) [+ proj(z)]; every ResBlock starts its emb path with SiLU ----
    plan->time_off = b.ws_alloc(B);
    View e0 = b.new_act(1, 1, mc);
    {
      Op& op = b.push(OP_SINUS, "timestep_embedding"); op.si.B = B; op.si.dim = mc; op.si.out_off = e0.off; op.si.ld = e0.ld; op.si.cos_first = 1;
      const int half = mc / 2;
      std::vector<float> fr(half);
      for (int j = 0; j < half; ++j) { const float arg = (float)(-std::log(10000.0)) * (float)j / (float)half; fr[j] = (float)std::exp((double)arg); }   // unet_openai.py:75-77
      op.si.freq_off = b.add_params(fr);
    }
    auto linear = [&](const std::string& name, const View& a, int cin, int cout, int act, const View* res) {
      Builder::ConvSpec l; l.name = name; l.a = a; l.cm = ChanMap::identity(cin); l.cin_true = cin;
      l.w = b.T(name + ".weight", (int64_t)cout * cin); l.bias = b.T(name + ".bias", cout); l.cout = cout; l.act = act;
      if (res) { l.has_res = true; l.r = *res; }
      return b.emit_conv(l).out;
    };
    View e1 = linear("time_embed.0", e0, mc, emb, ACT_SILU, nullptr);
    View embs;
    if (with_z) {
      View e2 = linear("time_embed.2", e1, emb, emb, ACT_NONE, nullptr);
      plan->aux_C = cfg.z_dim;
      View zin; zin.H = zin.W = 1; zin.C = pad8(cfg.z_dim); zin.ld = zin.C; zin.off = b.ws_alloc((int64_t)B * zin.ld);
      if (zin.C != cfg.z_dim) throw std::runtime_error("plan: z_dim must be a multiple of 8");
      plan->aux_off = zin.off;
      View z1 = linear("proj.0", zin, cfg.z_dim, emb, ACT_SILU, nullptr);
      embs = linear("proj.2", z1, emb, emb, ACT_SILU, &e2);          // SiLU(emb + z_proj): first op of every emb_layers
    } else {
      embs = linear("time_embed.2", e1, emb, emb, ACT_SILU, nullptr);
    }
    // all ResBlock.emb_layers.1 Linears as one GEMM
    struct RB { std::string pre; int cout; int col; };
    std::vector<RB> rbs;
    auto collect = [&](const std::string& pre, const OALayer& l) { if (l.kind == 1) rbs.push_back({pre, l.cout, 0}); };
    for (size_t bi = 0; bi < S.inputs.size(); ++bi) for (size_t l = 0; l < S.inputs[bi].size(); ++l) collect("input_blocks." + std::to_string(bi) + "." + std::to_string(l) + ".", S.inputs[bi][l]);
    for (size_t l = 0; l < S.middle.size(); ++l) collect("middle_block." + std::to_string(l) + ".", S.middle[l]);
    for (size_t bi = 0; bi < S.outputs.size(); ++bi) for (size_t l = 0; l < S.outputs[bi].size(); ++l) collect("output_blocks." + std::to_string(bi) + "." + std::to_string(l) + ".", S.outputs[bi][l]);
    int cond_ld = 0;
    for (auto& r : rbs) { r.col = cond_ld; cond_ld += pad8(r.cout); }
    std::map<std::string, int> cond_col;
    for (auto& r : rbs) cond_col[r.pre] = r.col;
    View cond_all = b.new_act(1, 1, cond_ld);
    {
      const int Kp = round_up(emb, 32), Npad = round_up(cond_ld, 32);
      std::vector<float> w((size_t)Npad * Kp, 0.f), bb(Npad, 0.f);
      for (auto& r : rbs) {
        const float* ww = b.T(r.pre + "emb_layers.1.weight", (int64_t)r.cout * emb); const float* wb = b.T(r.pre + "emb_layers.1.bias", r.cout);
        for (int c = 0; c < r.cout; ++c) { std::memcpy(&w[(size_t)(r.col + c) * Kp], &ww[(size_t)c * emb], sizeof(float) * emb); bb[r.col + c] = wb[c]; }
        plan->flops_dense += 2.0 * B * r.cout * emb; plan->flops_dense_gemm += 2.0 * B * r.cout * emb;
      }
      Op& op = b.push(OP_GEMM, "emb_layers.1 (all ResBlocks)");
      GemmOp& g = op.g; g.B = B; g.a = embs; g.N = cond_ld; g.Npad = Npad; g.Kp = Kp; g.out = cond_all;
      g.w_off = b.add_params(w); g.bias_off = b.add_params(bb);
      Builder::choose_tile(B, 1, cond_ld, &g.BM, &g.BN);
      op.flops = 2.0 * B * Npad * emb; plan->flops_exec += op.flops;
    }

    // ---- layer emitters ----
    auto gn32 = [&](const std::string& pre, const View& x, int C, bool silu) {
      View y = b.new_act(x.H, x.W, C);
      if (plan->gn32_part_off < 0) plan->gn32_part_off = b.ws_alloc((int64_t)B * 16 * 64 * 4, true);
      Op& op = b.push(OP_GN32, pre + (silu ? "+silu" : ""));
      op.gn.B = B; op.gn.in = x; op.gn.out = y; op.gn.groups = 32; op.gn.silu = silu ? 1 : 0;
      op.gn.gamma_off = b.add_params(b.pack_vec(b.T(pre + "weight", C), C, nullptr, C));
      op.gn.beta_off = b.add_params(b.pack_vec(b.T(pre + "bias", C), C, nullptr, C));
      return y;
    };
    auto conv3 = [&](const std::string& pre, const View& a, int cin, int cout, int stride, int up, View* dest) {
      Builder::ConvSpec c; c.name = pre; c.a = a; c.cm = ChanMap::identity(cin); c.cin_true = cin; c.kh = c.kw = 3; c.pad = 1; c.stride = stride; c.up = up;
      c.w = b.T(pre + ".weight", (int64_t)cout * cin * 9); c.bias = b.T(pre + ".bias", cout); c.cout = cout;
      c.Hout = a.H * up / stride; c.Wout = a.W * up / stride; c.dest = dest;
      return b.emit_conv(c).out;
    };
    auto resblock = [&](const std::string& pre, const View& x, int cin, int cout, View* dest) {
      View xs = gn32(pre + "in_layers.0.", x, cin, true);
      Builder::ConvSpec c1; c1.name = pre + "in_layers.2"; c1.a = xs; c1.cm = ChanMap::identity(cin); c1.cin_true = cin; c1.kh = c1.kw = 3; c1.pad = 1;
      c1.w = b.T(pre + "in_layers.2.weight", (int64_t)cout * cin * 9); c1.bias = b.T(pre + "in_layers.2.bias", cout); c1.cout = cout; c1.Hout = x.H; c1.Wout = x.W;
      // + emb_out[:, :, None, None]: a 1-pixel residual read by every output pixel (up2 >= grid size maps all pixels to p2 = 0)
      c1.has_res = true; c1.r = cond_all; c1.r.off = cond_all.off + cond_col.at(pre); c1.r.C = pad8(cout); c1.up2 = 1 << 20;
      View h1 = b.emit_conv(c1).out;
      b.free_act(xs);
      View hs = gn32(pre + "out_layers.0.", h1, cout, true);
      b.free_act(h1);
      Builder::ConvSpec c2; c2.name = pre + "out_layers.3"; c2.a = hs; c2.cm = ChanMap::identity(cout); c2.cin_true = cout; c2.kh = c2.kw = 3; c2.pad = 1;
      c2.w = b.T(pre + "out_layers.3.weight", (int64_t)cout * cout * 9); c2.bias = b.T(pre + "out_layers.3.bias", cout); c2.cout = cout; c2.Hout = x.H; c2.Wout = x.W;
      if (cin != cout) { c2.has2 = true; c2.a2 = x; c2.cm2 = ChanMap::identity(cin); c2.cin2 = cin; c2.w2 = b.T(pre + "skip_connection.weight", (int64_t)cout * cin); c2.bias2 = b.T(pre + "skip_connection.bias", cout); }
      else { c2.has_res = true; c2.r = x; }
      c2.dest = dest;
      View o = b.emit_conv(c2).out;
      b.free_act(hs);
      return o;
    };
    auto attnblock = [&](const std::string& pre, const View& x, int ch, View* dest) {
      View n = gn32(pre + "norm.", x, ch, false);
      Builder::ConvSpec q; q.name = pre + "qkv"; q.a = n; q.cm = ChanMap::identity(ch); q.cin_true = ch;
      q.w = b.T(pre + "qkv.weight", (int64_t)3 * ch * ch); q.bias = b.T(pre + "qkv.bias", 3 * ch); q.cout = 3 * ch; q.Hout = x.H; q.Wout = x.W;
      View qkv = b.emit_conv(q).out;
      b.free_act(n);
      View att = b.new_act(x.H, x.W, ch);
      {
        Op& op = b.push(OP_ATTN_QKV, pre + "attention");
        op.aq.B = B; op.aq.qkv = qkv; op.aq.out = att; op.aq.heads = cfg.num_heads; op.aq.chh = ch / cfg.num_heads;
        op.aq.scale = (float)(1.0 / std::sqrt(std::sqrt((double)(ch / cfg.num_heads))));
        if (x.H * x.W > 64) throw std::runtime_error("plan: AttentionBlock over more than 64 positions is not built");
      }
      b.free_act(qkv);
      Builder::ConvSpec o; o.name = pre + "proj_out"; o.a = att; o.cm = ChanMap::identity(ch); o.cin_true = ch;
      o.w = b.T(pre + "proj_out.weight", (int64_t)ch * ch); o.bias = b.T(pre + "proj_out.bias", ch); o.cout = ch; o.Hout = x.H; o.Wout = x.W;
      o.has_res = true; o.r = x; o.dest = dest;
      View y = b.emit_conv(o).out;
      b.free_act(att);
      return y;
    };

    // ---- concat buffers: output block j reads cat([h, hs.pop()]); input block n_in-1-j writes its half directly ----
    const int n_in = (int)S.inputs.size(), n_out = (int)S.outputs.size();
    if (n_in != n_out) throw std::runtime_error("plan: UNetModel skip stack mismatch");
    std::vector<int> in_H(n_in);      // spatial size of each input block's output
    { int H = cfg.H; for (int bi = 0; bi < n_in; ++bi) { for (auto& l : S.inputs[bi]) if (l.kind == 3) H /= 2; in_H[bi] = H; } }
    // channel count entering each output block from the previous stage
    std::vector<int> prev_ch(n_out);
    { int ch = S.middle.back().cout; for (int j = 0; j < n_out; ++j) { prev_ch[j] = ch; ch = S.outputs[j][0].cout; } }
    std::vector<View> cat(n_out);
    for (int j = 0; j < n_out; ++j) {
      const int bi = n_in - 1 - j, Hs = in_H[bi], Ws = Hs * cfg.W / cfg.H;
      View cb; cb.H = Hs; cb.W = Ws; cb.C = prev_ch[j] + S.chans[bi]; cb.ld = cb.C; cb.off = b.ws_alloc((int64_t)Hs * Ws * B * cb.ld);
      if (prev_ch[j] % 8 || S.chans[bi] % 8) throw std::runtime_error("plan: UNetModel channels must be multiples of 8");
      cat[j] = cb;
    }
    auto skip_dest = [&](int bi) { View d = cat[n_in - 1 - bi]; d.off += prev_ch[n_in - 1 - bi]; d.C = S.chans[bi]; return d; };
    auto main_dest = [&](int j) { View d = cat[j]; d.C = prev_ch[j]; return d; };

    // ---- input blocks ----
    View h = xin;
    for (int bi = 0; bi < n_in; ++bi) {
      const std::string pb = "input_blocks." + std::to_string(bi) + ".";
      for (size_t li = 0; li < S.inputs[bi].size(); ++li) {
        const OALayer& l = S.inputs[bi][li];
        const bool last = li + 1 == S.inputs[bi].size();
        View d = skip_dest(bi); View* dp = last ? &d : nullptr;
        const std::string pre = pb + std::to_string(li) + ".";
        View y;
        if (l.kind == 0) y = conv3(pre.substr(0, pre.size() - 1), h, l.cin, l.cout, 1, 1, dp);
        else if (l.kind == 1) y = resblock(pre, h, l.cin, l.cout, dp);
        else if (l.kind == 2) y = attnblock(pre, h, l.cin, dp);
        else y = conv3(pre + "op", h, l.cin, l.cin, 2, 1, dp);
        h = y;
      }
    }
    // ---- middle ----
    for (size_t li = 0; li < S.middle.size(); ++li) {
      const OALayer& l = S.middle[li];
      const std::string pre = "middle_block." + std::to_string(li) + ".";
      View d = main_dest(0); View* dp = li + 1 == S.middle.size() ? &d : nullptr;
      h = l.kind == 1 ? resblock(pre, h, l.cin, l.cout, dp) : attnblock(pre, h, l.cin, dp);
    }
    // ---- output blocks ----
    for (int j = 0; j < n_out; ++j) {
      const std::string pb = "output_blocks." + std::to_string(j) + ".";
      h = cat[j];
      for (size_t li = 0; li < S.outputs[j].size(); ++li) {
        const OALayer& l = S.outputs[j][li];
        const bool last = li + 1 == S.outputs[j].size();
        View d; View* dp = nullptr;
        if (last && j + 1 < n_out) { d = main_dest(j + 1); dp = &d; }
        const std::string pre = pb + std::to_string(li) + ".";
        if (l.kind == 1) h = resblock(pre, h, l.cin, l.cout, dp);
        else if (l.kind == 2) h = attnblock(pre, h, l.cin, dp);
        else h = conv3(pre + "conv", h, l.cin, l.cin, 1, 2, dp);
      }
    }
    // ---- out: GroupNorm32 -> SiLU -> conv3x3 ----
    View hs = gn32("out.0.", h, mc, true);
    Builder::ConvSpec fc; fc.name = "out.2"; fc.a = hs; fc.cm = ChanMap::identity(mc); fc.cin_true = mc; fc.kh = fc.kw = 3; fc.pad = 1;
    fc.w = b.T("out.2.weight", (int64_t)cfg.out_channels * mc * 9); fc.bias = b.T("out.2.bias", cfg.out_channels);
    fc.cout = cfg.out_channels; fc.Hout = cfg.H; fc.Wout = cfg.W; fc.out_nchw = true; fc.Hn = cfg.H; fc.Wn = cfg.W;
    b.emit_conv(fc);
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

// ------------------------------------------------------------------------------------------------
// decoders
// ------------------------------------------------------------------------------------------------
namespace {

// BatchNorm(eval) folded into the preceding bias-free conv / Linear: y = conv * s + (beta - mean * s)
void bn_fold(const Builder& b, const std::string& pre, int c, std::vector<double>* scale, std::vector<double>* shift) {
  const float* w = b.T(pre + "weight", c); const float* bi = b.T(pre + "bias", c);
  const float* rm = b.T(pre + "running_mean", c); const float* rv = b.T(pre + "running_var", c);
  scale->resize(c); shift->resize(c);
  for (int i = 0; i < c; ++i) {
    double s = (double)w[i] / std::sqrt((double)rv[i] + 1e-5);
    (*scale)[i] = s; (*shift)[i] = (double)bi[i] - (double)rm[i] * s;
  }
}

// RBlock / RBlockN (h_vae_model_copy.py:9-39, 347-377) with BN folded.  `x` may be a lower-resolution
// tensor whose nearest upsample (factor up_in) is folded into the reads.
View emit_rblock(Builder& b, const std::string& pre, const View& x, int up_in, int cin, int cmid, int cout, int act) {
  const int H = x.H * up_in, W = x.W * up_in;
  std::vector<double> s1, t1, s2, t2;
  bn_fold(b, pre + "conv.1.", cmid, &s1, &t1);
  bn_fold(b, pre + "conv.4.", cout, &s2, &t2);
  Builder::ConvSpec c1; c1.name = pre + "conv.0+bn+act"; c1.a = x; c1.up = up_in; c1.cm = ChanMap::identity(cin); c1.cin_true = cin; c1.kh = c1.kw = 3; c1.pad = 1;
  c1.w = b.T(pre + "conv.0.weight", (int64_t)cmid * cin * 9); c1.cout = cmid; c1.Hout = H; c1.Wout = W; c1.out_scale = &s1; c1.out_shift = &t1; c1.act = act;
  View m = b.emit_conv(c1).out;
  Builder::ConvSpec c2; c2.name = pre + "conv.3+bn+skip+act"; c2.a = m; c2.cm = ChanMap::identity(cmid); c2.cin_true = cmid; c2.kh = c2.kw = 3; c2.pad = 1;
  c2.w = b.T(pre + "conv.3.weight", (int64_t)cout * cmid * 9); c2.cout = cout; c2.Hout = H; c2.Wout = W; c2.out_scale = &s2; c2.out_shift = &t2; c2.act = act;
  c2.up2 = up_in;
  if (cin != cout) { c2.has2 = true; c2.a2 = x; c2.cm2 = ChanMap::identity(cin); c2.cin2 = cin; c2.w2 = b.T(pre + "size_conv.weight", (int64_t)cout * cin); }
  else { c2.has_res = true; c2.r = x; }
  View o = b.emit_conv(c2).out;
  b.free_act(m);
  return o;
}

}  // namespace

bool build_resdec_plan(const ResDecCfg& cfg, int B, int z_ld, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = resdec_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    int init = cfg.size_in;
    for (int i = 0; i < cfg.n_enc; ++i) init /= cfg.enc[i][3];
    const int clast = cfg.enc[cfg.n_enc - 1][2];
    const int P0 = init * init;
    plan->in_C = cfg.size_z; plan->in_ld = z_ld; plan->in_H = plan->in_W = 1;
    plan->out_C = cfg.img_ch; plan->out_H = plan->out_W = cfg.size_in;
    if (cfg.size_z % 4 || z_ld % 4) throw std::runtime_error("plan: latent width / pitch must be a multiple of 4 floats");
    // z_lin + ReLU, written directly in PBC: weights per output pixel p: W_p[c][k] = z_lin.weight[c*P0 + p][k]
    // (h_vae_model_copy.py:124-125: relu(z_lin(z)).view(B, C, s, s))
    View h = b.new_act(init, init, clast);
    {
      Op& op = b.push(OP_GEMM, "z_lin+relu");
      GemmOp& g = op.g; g.B = B; g.Hout = init; g.Wout = init;
      g.a_ext = 1; g.a_bcast = 1; g.a.C = cfg.size_z; g.a.ld = z_ld; g.a.H = g.a.W = 1; g.a.off = 0;
      g.N = clast; g.Npad = round_up(pad8(clast), 32);
      const float* w = b.T("z_lin.weight", (int64_t)P0 * clast * cfg.size_z); const float* bi = b.T("z_lin.bias", (int64_t)P0 * clast);
      g.Kp = round_up(cfg.size_z, 32);
      std::vector<float> wp((size_t)P0 * g.Npad * g.Kp, 0.f), bp((size_t)P0 * g.Npad, 0.f);
      for (int p = 0; p < P0; ++p) for (int c = 0; c < clast; ++c) {
        std::memcpy(&wp[((size_t)p * g.Npad + c) * g.Kp], &w[((size_t)c * P0 + p) * cfg.size_z], sizeof(float) * cfg.size_z);
        bp[(size_t)p * g.Npad + c] = bi[(size_t)c * P0 + p];
      }
      g.w_off = b.add_params(wp); g.w_pix_stride = (int64_t)g.Npad * g.Kp;
      g.bias_off = b.add_params(bp); g.bias_pix_stride = g.Npad;
      g.act = ACT_RELU; g.out = h;
      Builder::choose_tile(B, P0, pad8(clast), &g.BM, &g.BN);
      plan->flops_dense += 2.0 * B * P0 * clast * cfg.size_z; plan->flops_dense_gemm += 2.0 * B * P0 * clast * cfg.size_z;
      plan->flops_exec += 2.0 * B * P0 * g.Npad * cfg.size_z;
    }
    const int act = cfg.variant == 0 ? ACT_LRELU02 : ACT_GELU;
    int up_in = 1;
    for (int i = 0; i < cfg.n_dec; ++i) {
      View o = emit_rblock(b, "dec.r_blocks." + std::to_string(i) + ".", h, up_in, cfg.dec[i][0], cfg.dec[i][1], cfg.dec[i][2], act);
      b.free_act(h);
      const int up = cfg.dec[i][3];
      if (cfg.variant == 0) { h = o; up_in = up; }          // nearest: folded into the next reads
      else {
        View u = b.new_act(o.H * up, o.W * up, cfg.dec[i][2]);
        Op& op = b.push(OP_BILINEAR, "dec.r_blocks." + std::to_string(i) + ".up_pool");
        op.bl.B = B; op.bl.scale = up; op.bl.in = o; op.bl.out = u;
        b.free_act(o); h = u; up_in = 1;
      }
    }
    const int cc = cfg.dec[cfg.n_dec - 1][2];
    View o = emit_rblock(b, "dec.ch_dec.0.", h, up_in, cc, cc, cc, ACT_LRELU02);   // plain RBlock in both variants
    b.free_act(h);
    Builder::ConvSpec fc; fc.name = "dec.ch_dec.1"; fc.a = o; fc.cm = ChanMap::identity(cc); fc.cin_true = cc; fc.kh = fc.kw = 5; fc.pad = 2;
    fc.w = b.T("dec.ch_dec.1.weight", (int64_t)cfg.img_ch * cc * 25); fc.bias = b.T("dec.ch_dec.1.bias", cfg.img_ch);
    fc.cout = cfg.img_ch; fc.Hout = o.H; fc.Wout = o.W; fc.out_nchw = true; fc.Hn = o.H; fc.Wn = o.W;
    fc.act = cfg.variant == 0 ? ACT_NONE : ACT_SIGMOID;
    if (o.H != cfg.size_in) throw std::runtime_error("plan: decoder geometry does not reach size_in");
    b.emit_conv(fc);
    b.free_act(o);
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

// ProjectionHead (h_vae_model.py:964-985, eval mode): p = projection(x); y = LayerNorm(fc(GELU(p)) + p).  The first Linear is
// emitted twice (plain and through GELU: the residual needs p itself), the LayerNorm is GroupNorm(1, d) over the single
// position and writes the plan's external output.
static void emit_proj_head(Builder& b, const View& mu, int d) {
  const float* wp = b.T("proj.projection.weight", (int64_t)d * d); const float* bp = b.T("proj.projection.bias", d);
  Builder::ConvSpec a; a.name = "proj.projection"; a.a = mu; a.cm = ChanMap::identity(d); a.cin_true = d; a.w = wp; a.bias = bp; a.cout = d;
  View p = b.emit_conv(a).out;
  Builder::ConvSpec g = a; g.name = "proj.projection+gelu"; g.act = ACT_GELU;
  View pg = b.emit_conv(g).out;
  Builder::ConvSpec f; f.name = "proj.fc+residual"; f.a = pg; f.cm = ChanMap::identity(d); f.cin_true = d;
  f.w = b.T("proj.fc.weight", (int64_t)d * d); f.bias = b.T("proj.fc.bias", d); f.cout = d; f.has_res = true; f.r = p; f.want_stats = true;
  Builder::ConvResult rf = b.emit_conv(f);
  View zero; zero.H = zero.W = 1; zero.C = pad8(d); zero.ld = zero.C;
  zero.off = b.ws_alloc((int64_t)b.B * zero.ld, /*pin=*/true);   // fresh, never recycled, never written: the workspace is zeroed once at creation
  Op& op = b.push(OP_GN_RES, "proj.layer_norm");
  GnResOp& r = op.gr; r.B = b.B; r.u = rf.out; r.x = zero; r.out = rf.out; r.out.off = 0; r.out.ld = d; r.out_ext = 1;
  r.part_off = rf.stat_off; r.nparts = rf.stat_nparts; r.inv_count = 1.0f / (float)d;
  r.gamma_off = b.add_params(b.pack_vec(b.T("proj.layer_norm.weight", d), d, nullptr, pad8(d)));
  r.beta_off = b.add_params(b.pack_vec(b.T("proj.layer_norm.bias", d), d, nullptr, pad8(d)));
  b.plan->out_C = d;
}

// ResEncoder / ResEncoderN (h_vae_model_copy.py:41-71, 379-409): conv5x5+BN+LeakyReLU -> AvgPool2 -> RBlocks (each followed
// by AvgPool(down)) -> channel halves -> mu_lin / logvar_lin on the flattened (c, y, x) features.  The two Linears are one
// full-image convolution with block-structured weights; output rows are [mu | logvar].
bool build_resenc_plan(const ResDecCfg& cfg, int B, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = resenc_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    const bool nvar = (cfg.variant & 1) != 0, rep = (cfg.variant & 2) != 0;
    plan->in_C = cfg.img_ch; plan->in_H = plan->in_W = cfg.size_in; plan->in_ld = 0;
    plan->out_C = 2 * cfg.size_z; plan->out_H = plan->out_W = 1;
    View xin = b.new_act(cfg.size_in, cfg.size_in, cfg.img_ch);
    { Op& op = b.push(OP_NCHW_TO_PBC, "input"); op.cv.B = B; op.cv.C = cfg.img_ch; op.cv.Hn = cfg.size_in; op.cv.Wn = cfg.size_in; op.cv.out = xin; }
    const int c0 = cfg.enc[0][0];
    std::vector<double> s0, t0;
    bn_fold(b, "enc.ch_enc.1.", c0, &s0, &t0);
    Builder::ConvSpec ic; ic.name = "enc.ch_enc.0+bn+act"; ic.a = xin; ic.cm = ChanMap::identity(cfg.img_ch); ic.cin_true = cfg.img_ch; ic.kh = ic.kw = 5; ic.pad = 2;
    ic.w = b.T("enc.ch_enc.0.weight", (int64_t)c0 * cfg.img_ch * 25); ic.bias = b.T("enc.ch_enc.0.bias", c0); ic.cout = c0; ic.Hout = ic.Wout = cfg.size_in;
    ic.out_scale = &s0; ic.out_shift = &t0; ic.act = nvar ? ACT_LRELU01 : ACT_LRELU02;
    View h = b.emit_conv(ic).out;
    b.free_act(xin);
    auto pool = [&](const View& in, int k, const std::string& name) {
      if (in.H % k || in.W % k) throw std::runtime_error("plan: AvgPool2d does not divide the image");
      View o = b.new_act(in.H / k, in.W / k, in.C);
      Op& op = b.push(OP_AVGPOOL, name); op.bl.B = B; op.bl.scale = k; op.bl.in = in; op.bl.out = o;
      b.free_act(in);
      return o;
    };
    h = pool(h, 2, "enc.ch_enc.3 (AvgPool2d 2)");
    for (int i = 0; i < cfg.n_enc; ++i) {
      View o = emit_rblock(b, "enc.r_blocks." + std::to_string(i) + ".", h, 1, cfg.enc[i][0], cfg.enc[i][1], cfg.enc[i][2], nvar ? ACT_GELU : ACT_LRELU02);
      b.free_act(h);
      h = cfg.enc[i][3] > 1 ? pool(o, cfg.enc[i][3], "enc.r_blocks." + std::to_string(i) + ".down_pool") : o;
    }
    const int C = cfg.enc[cfg.n_enc - 1][2], half = C / 2, P = h.H * h.W, sz = cfg.size_z;
    if (rep) {                                           // Cel{Img,Mask}Rep: mu only, then the projection head
      if (sz % 8) throw std::runtime_error("plan: projection head width must be a multiple of 8");
      const float* wm = b.T("enc.mu_lin.weight", (int64_t)sz * half * P);
      std::vector<float> w((size_t)sz * C * P, 0.f);
      for (int n = 0; n < sz; ++n) for (int c = 0; c < half; ++c) for (int q = 0; q < P; ++q) w[((size_t)n * C + c) * P + q] = wm[((size_t)n * half + c) * P + q];
      Builder::ConvSpec fc; fc.name = "enc.mu_lin"; fc.a = h; fc.cm = ChanMap::identity(C); fc.cin_true = C; fc.kh = h.H; fc.kw = h.W; fc.pad = 0;
      fc.w = w.data(); fc.bias = b.T("enc.mu_lin.bias", sz); fc.cout = sz; fc.Hout = fc.Wout = 1;
      View mu = b.emit_conv(fc).out;
      b.free_act(h);
      emit_proj_head(b, mu, sz);
      finalize_gemm_tables(plan);
      return true;
    }
    const float* wm = b.T("enc.mu_lin.weight", (int64_t)sz * half * P); const float* wl = b.T("enc.logvar_lin.weight", (int64_t)sz * half * P);
    const float* bm = b.T("enc.mu_lin.bias", sz); const float* bl = b.T("enc.logvar_lin.bias", sz);
    std::vector<float> w((size_t)2 * sz * C * P, 0.f), bias(2 * sz);
    for (int n = 0; n < sz; ++n) {
      for (int c = 0; c < half; ++c) for (int q = 0; q < P; ++q) {
        w[((size_t)n * C + c) * P + q] = wm[((size_t)n * half + c) * P + q];
        w[((size_t)(sz + n) * C + half + c) * P + q] = wl[((size_t)n * half + c) * P + q];
      }
      bias[n] = bm[n]; bias[sz + n] = bl[n];
    }
    Builder::ConvSpec fc; fc.name = "enc.mu_lin|enc.logvar_lin"; fc.a = h; fc.cm = ChanMap::identity(C); fc.cin_true = C; fc.kh = h.H; fc.kw = h.W; fc.pad = 0;
    fc.w = w.data(); fc.bias = bias.data(); fc.cout = 2 * sz; fc.Hout = fc.Wout = 1; fc.out_nchw = true; fc.Hn = fc.Wn = 1;
    b.emit_conv(fc);
    b.free_act(h);
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

static const int kMlpMul[8] = {1, 2, 6, 6, 4, 2, 1, 1};

std::vector<TensorSpec> mlp_tensor_specs(const MlpCfg& c) {
  SpecList L;
  const int d = c.n_mod * c.size_z;
  for (int i = 0; i < 7; ++i) {
    std::string l = "layers." + std::to_string(2 * i) + ".";
    L.add(l + "weight", {d * kMlpMul[i + 1], d * kMlpMul[i]}); L.add(l + "bias", {d * kMlpMul[i + 1]});
  }
  return L.v;
}

bool build_mlp_plan(const MlpCfg& cfg, int B, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = mlp_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    const int d = cfg.n_mod * cfg.size_z;
    if (d % 32) throw std::runtime_error("plan: n_mod*size_z must be a multiple of 32");
    plan->in_C = d; plan->in_ld = d; plan->in_H = plan->in_W = 1; plan->out_C = d; plan->out_H = plan->out_W = 1;
    View h;
    for (int i = 0; i < 7; ++i) {
      const int cin = d * kMlpMul[i], cout = d * kMlpMul[i + 1];
      std::string l = "layers." + std::to_string(2 * i);
      Builder::ConvSpec c; c.name = l + (i < 6 ? "+softplus" : ""); c.cm = ChanMap::identity(cin); c.cin_true = cin;
      c.w = b.T(l + ".weight", (int64_t)cout * cin); c.bias = b.T(l + ".bias", cout); c.cout = cout; c.act = i < 6 ? ACT_SOFTPLUS : ACT_NONE;
      if (i == 0) { c.a.C = d; c.a.ld = d; c.a.H = c.a.W = 1; c.a.off = 0; } else c.a = h;
      if (i == 6) { c.out_nchw = true; c.Hn = c.Wn = 1; }
      Builder::ConvResult r = b.emit_conv(c);
      if (i == 0) plan->ops.back().g.a_ext = 1; else b.free_act(h);
      h = r.out;
    }
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

bool build_attrdec_plan(const AttrDecCfg& cfg, int B, int z_ld, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = attrdec_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    plan->in_C = cfg.size_z; plan->in_ld = z_ld; plan->in_H = plan->in_W = 1;
    plan->out_C = cfg.att_size; plan->out_H = plan->out_W = 1;
    if (cfg.size_z % 4 || z_ld % 4) throw std::runtime_error("plan: latent width / pitch must be a multiple of 4 floats");
    View h; int prev = cfg.size_z;
    for (int i = 0; i < 5; ++i) {
      std::string l = "dec_net." + std::to_string(3 * i) + ".", bnp = "dec_net." + std::to_string(3 * i + 1) + ".";
      std::vector<double> s, t;
      bn_fold(b, bnp, kAttrW[i], &s, &t);
      Builder::ConvSpec c; c.name = l + "+bn+relu"; c.cm = ChanMap::identity(prev); c.cin_true = prev;
      c.w = b.T(l + "weight", (int64_t)kAttrW[i] * prev); c.bias = b.T(l + "bias", kAttrW[i]); c.cout = kAttrW[i];
      c.out_scale = &s; c.out_shift = &t; c.act = ACT_RELU;
      if (i == 0) { c.a.C = cfg.size_z; c.a.ld = z_ld; c.a.H = c.a.W = 1; c.a.off = 0; }
      else c.a = h;
      Builder::ConvResult r = b.emit_conv(c);
      if (i == 0) plan->ops.back().g.a_ext = 1;
      else b.free_act(h);
      h = r.out; prev = kAttrW[i];
    }
    Builder::ConvSpec c; c.name = "dec_net.15"; c.a = h; c.cm = ChanMap::identity(prev); c.cin_true = prev;
    c.w = b.T("dec_net.15.weight", (int64_t)cfg.att_size * prev); c.bias = b.T("dec_net.15.bias", cfg.att_size); c.cout = cfg.att_size;
    c.out_nchw = true; c.Hn = c.Wn = 1;
    b.emit_conv(c);
    b.free_act(h);
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

bool build_attrenc_plan(const AttrEncCfg& cfg, int B, const float* const* tensors, Plan* plan) {
  try {
    std::vector<TensorSpec> specs = attrenc_tensor_specs(cfg);
    Builder b(plan, B, specs, tensors);
    const int nout = cfg.ae ? cfg.size_z : 2 * cfg.size_z;
    const std::string pre = cfg.ae == 2 ? "enc." : "";
    plan->in_C = cfg.att_size; plan->in_H = plan->in_W = 1; plan->in_ld = 0;
    plan->out_C = nout; plan->out_H = plan->out_W = 1;
    View h = b.new_act(1, 1, cfg.att_size);
    { Op& op = b.push(OP_NCHW_TO_PBC, "input"); op.cv.B = B; op.cv.C = cfg.att_size; op.cv.Hn = 1; op.cv.Wn = 1; op.cv.out = h; }
    int prev = cfg.att_size;
    for (int i = 0; i < 5; ++i) {
      std::string l = pre + "enc_net." + std::to_string(3 * i) + ".", bnp = pre + "enc_net." + std::to_string(3 * i + 1) + ".";
      std::vector<double> s, t;
      bn_fold(b, bnp, kAttrEncW[i], &s, &t);
      Builder::ConvSpec c; c.name = l + "+bn+relu"; c.a = h; c.cm = ChanMap::identity(prev); c.cin_true = prev;
      c.w = b.T(l + "weight", (int64_t)kAttrEncW[i] * prev); c.bias = b.T(l + "bias", kAttrEncW[i]); c.cout = kAttrEncW[i];
      c.out_scale = &s; c.out_shift = &t; c.act = ACT_RELU;
      View o = b.emit_conv(c).out;
      b.free_act(h);
      h = o; prev = kAttrEncW[i];
    }
    std::vector<float> w((size_t)nout * prev), bias(nout);
    std::memcpy(w.data(), b.T(pre + "mu_lin.weight", (int64_t)cfg.size_z * prev), sizeof(float) * cfg.size_z * prev);
    std::memcpy(bias.data(), b.T(pre + "mu_lin.bias", cfg.size_z), sizeof(float) * cfg.size_z);
    if (cfg.ae == 2) {                                   // CelAttrRep: mu -> projection head
      Builder::ConvSpec c; c.name = "enc.mu_lin"; c.a = h; c.cm = ChanMap::identity(prev); c.cin_true = prev; c.w = w.data(); c.bias = bias.data(); c.cout = cfg.size_z;
      View mu = b.emit_conv(c).out;
      b.free_act(h);
      emit_proj_head(b, mu, cfg.size_z);
      finalize_gemm_tables(plan);
      return true;
    }
    if (!cfg.ae) {
      std::memcpy(w.data() + (size_t)cfg.size_z * prev, b.T("logvar_lin.weight", (int64_t)cfg.size_z * prev), sizeof(float) * cfg.size_z * prev);
      std::memcpy(bias.data() + cfg.size_z, b.T("logvar_lin.bias", cfg.size_z), sizeof(float) * cfg.size_z);
    }
    Builder::ConvSpec c; c.name = cfg.ae ? "mu_lin" : "mu_lin|logvar_lin"; c.a = h; c.cm = ChanMap::identity(prev); c.cin_true = prev;
    c.w = w.data(); c.bias = bias.data(); c.cout = nout; c.out_nchw = true; c.Hn = c.Wn = 1;
    b.emit_conv(c);
    b.free_act(h);
    finalize_gemm_tables(plan);
    return true;
  } catch (const std::exception& e) {
    plan->error = e.what();
    return false;
  }
}

}  // namespace sbm
