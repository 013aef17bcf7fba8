// Model-level entry points of the C-ABI (last row): an opaque vs_model_t built from a card + the reference
// state_dict (HOST fp32 tensors by name), immutable after creation, and whole-path calls at the granularity of
// Wam.embed / Videoseal.embed (wam.py:134-204, videoseal.py:258-350) and detect (wam.py:206-234, videoseal.py:352-388).
// Everything here is host code: weight preparation (eval-BatchNorm folding, [N][tap][CinP] packing, the 3 x bf16 split and the
// LDS-image blocking of engine.py) and the launch sequences of engine.py::embedder_forward / extractor_forward, issued on the
// caller's stream into the caller's workspace through the operator-level entry points of this same library.
// No allocation after vs_model_create, no synchronisation, no torch.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "vs_common.h"

#pragma clang fp contract(off)      // host-side weight folding must round like the torch ops it restates (no fused multiply-add)

namespace {

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

struct HostT { const float* p; int64_t n; };

struct CW {                      // one conv / linear layer on the device
  float* wt = nullptr; float* bias = nullptr; void* split = nullptr; void* blk = nullptr;
  int N = 0, KH = 0, KW = 0, CinP = 0;
  float w_mul = 1.f;             // power of two the weights were multiplied with before the 2 x f16 split (arith 2)
};
struct RB { CW c0, c1, res; int cout = 0; };
struct Up { CW conv; CW gemm; bool lowres = false; float* lnw = nullptr; float* lnb = nullptr; RB rb; };   // gemm: nine taps at the low resolution (vs_upconv_gather_ln)
struct Down { float* lnw = nullptr; float* lnb = nullptr; CW conv, gemm; };      // gemm: the same weights as a 1x1 GEMM on the patch matrix (vs_layernorm_patch2x2)
struct Blk {
  float *wdw = nullptr, *bdw = nullptr, *lnw = nullptr, *lnb = nullptr, *gamma = nullptr, *beta = nullptr; CW pw1, pw2;
  void* fuse = nullptr; float fm1 = 1.f, fm2 = 1.f;      // weight image of the fused block kernel (vs_cnx_block) and its two power-of-two scales
};

struct Act { float* p; int B, H, W, C, ld; int64_t rows() const { return (int64_t)B * H * W; } };

}  // namespace

struct vs_model {
  vs_model_cfg_t c;
  std::vector<void*> dev;
  std::vector<int> zc;
  int bott = 0;
  // embedder
  RB inc;
  std::vector<CW> down_conv;
  std::vector<RB> down_rb, bottleneck;
  CW b0_lat, b0_msg;             // first bottleneck block with the message channels as a border-class table (engine.py::resblock_msg0)
  bool b0_table = false;
  std::vector<Up> ups;
  float *outc_w = nullptr, *outc_b = nullptr, *table = nullptr;
  // extractor
  CW stem, head, head_gemm;      // head_gemm: the same 3x3 weights seen as a 1x1 GEMM on the patch matrix (vs_im2col3x3)
  float *stem_lnw = nullptr, *stem_lnb = nullptr, *head_lnw = nullptr, *head_lnb = nullptr, *lin_w = nullptr, *lin_b = nullptr;
  Down down[3];
  std::vector<Blk> stages[4];
  float ymat[3];
  float taps43[43];
  int arith = 3;                 // arithmetic of the split planes (vs_conv_desc_t::arith)
  bool ok = true;
};

static constexpr float A_MUL = 16.f;      // engine.py::A_MUL
static constexpr float A_MUL_GRN = 1.f;   // engine.py::A_MUL_GRN (the GRN-scaled operand of pwconv2)

namespace {

int xld(int C) { return C >= 128 ? rup(C, 32) : rup(C, 4); }     // engine.py::_xld

// ---------------------------------------------------------------- weight preparation (host) + upload
struct Packer {
  vs_model* m;
  const std::map<std::string, HostT>& sd;
  bool fail = false;

  const HostT& get(const std::string& k, bool optional = false) {
    static HostT none{nullptr, 0};
    auto it = sd.find(k);
    if (it == sd.end()) { if (!optional) fail = true; return none; }
    return it->second;
  }
  template <typename T>
  T* upload(const std::vector<T>& v) {
    void* d = nullptr;
    if (v.empty()) return nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) { fail = true; return nullptr; }
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) fail = true;
    m->dev.push_back(d);
    return static_cast<T*>(d);
  }
  float* vec(const std::string& k, int padded = 0) {
    const HostT& t = get(k);
    if (!t.p) return nullptr;
    std::vector<float> v((size_t)std::max<int64_t>(t.n, padded), 0.f);
    std::copy(t.p, t.p + t.n, v.begin());
    return upload(v);
  }
  // fp32 -> three bf16 terms by truncation (engine.py::split_bf16x3); planes [3][N][K]
  static void split3(const std::vector<float>& w, std::vector<uint16_t>& planes) {
    const size_t n = w.size();
    planes.assign(3 * n, 0);
    for (size_t i = 0; i < n; ++i) {
      float r = w[i];
      for (int p = 0; p < 3; ++p) {
        uint32_t u;
        std::memcpy(&u, &r, 4);
        const uint32_t hi = u & 0xffff0000u;
        planes[p * n + i] = (uint16_t)(hi >> 16);
        float h;
        std::memcpy(&h, &hi, 4);
        r = r - h;
      }
    }
  }
  // engine.py::split_f16x2: w * w_mul = hi + lo in round-to-nearest f16, w_mul = the power of two that puts max|w| into [2^13, 2^14)
  static float split2(const std::vector<float>& w, std::vector<uint16_t>& planes) {
    const size_t n = w.size();
    planes.assign(2 * n, 0);
    float amax = 0.f;
    for (float v : w) amax = std::max(amax, std::fabs(v));
    int kw = 0;
    if (amax > 0.f && std::isfinite(amax)) { int e; (void)std::frexp(amax, &e); kw = std::max(-100, std::min(100, 14 - e)); }
    for (size_t i = 0; i < n; ++i) {
      const float ws = std::ldexp(w[i], kw);
      const _Float16 hi = (_Float16)ws;
      const _Float16 lo = (_Float16)(ws - (float)hi);
      std::memcpy(&planes[i], &hi, 2);
      std::memcpy(&planes[n + i], &lo, 2);
    }
    return std::ldexp(1.f, kw);
  }
  // engine.py::pack_blocked: [P][N][K] -> [ceil(N/32)][K/16][P][64 slots][8], chunk order (channel chunk, tap), bank swizzle
  static void blocked(const std::vector<uint16_t>& planes, int n, int k, int ntaps, std::vector<uint16_t>& out) {
    const int NPL = (int)(planes.size() / ((size_t)n * k));
    const int G = (n + 31) / 32, nch = k / 16, spt = nch / ntaps;
    out.assign((size_t)G * nch * NPL * 64 * 8, 0);
    for (int p = 0; p < NPL; ++p)
      for (int row = 0; row < n; ++row) {
        const int g = row / 32, r = row % 32;
        for (int tap = 0; tap < ntaps; ++tap)
          for (int cc = 0; cc < spt; ++cc) {
            const int blkidx = cc * ntaps + tap;                  // step order of the kernels
            for (int h = 0; h < 2; ++h) {
              const int slot = 2 * r + (h ^ ((r >> 3) & 1));
              const uint16_t* src = &planes[(size_t)p * n * k + (size_t)row * k + (size_t)(tap * spt + cc) * 16 + h * 8];
              uint16_t* dst = &out[((((size_t)g * nch + blkidx) * NPL + p) * 64 + slot) * 8];
              std::copy(src, src + 8, dst);
            }
          }
      }
  }
  // engine.py::pack_cnx_block: weight image of csrc/convnext_fused.hip for one ConvNeXt block.  w1 [4C][C], w2 [C][4C], b1 / beta [4C].
  // Per block of 32 h-channels: W1 rows as A fragments [k-step][plane][half][row 32][8]; W2 as B fragments [n-block][s][plane][half][n 32][8]
  // with the reduction index enumerated as k(s, half, v) = 8 (2 s + v / 4) + 4 half + v % 4; then 256 floats: b1[32], beta[32], zeros.
  void* cnx_image(const HostT& w1, const HostT& b1, const HostT& w2, const HostT& beta, int C, float& m1, float& m2) {
    const int H4 = 4 * C, nhb = H4 / 32, ks1 = C / 16, nb = C / 32;
    if (!w1.p || !b1.p || !w2.p || !beta.p || w1.n != (int64_t)H4 * C || w2.n != (int64_t)H4 * C || b1.n < H4 || beta.n < H4) return nullptr;
    std::vector<uint16_t> p1, p2;
    m1 = split2(std::vector<float>(w1.p, w1.p + w1.n), p1);          // [plane][4C][C]
    m2 = split2(std::vector<float>(w2.p, w2.p + w2.n), p2);          // [plane][C][4C]
    const size_t per_hb = (size_t)2048 * ks1 + (size_t)4096 * nb + 1024;      // W1 fragments, W2 fragments (two k-steps per n-block), 256 floats
    if ((int64_t)(per_hb * nhb) != vs_cnx_block_image_bytes(C)) { fail = true; return nullptr; }      // (layout drifted from convnext_fused.hip: loud)
    std::vector<uint8_t> img(per_hb * nhb, 0);
    const size_t n1 = (size_t)H4 * C;
    for (int hb = 0; hb < nhb; ++hb) {
      uint16_t* i1 = reinterpret_cast<uint16_t*>(img.data() + per_hb * hb);
      uint16_t* i2 = i1 + (size_t)1024 * ks1;
      float* aux = reinterpret_cast<float*>(img.data() + per_hb * hb + (size_t)2048 * ks1 + (size_t)4096 * nb);
      for (int ks = 0; ks < ks1; ++ks)
        for (int pl = 0; pl < 2; ++pl)
          for (int half = 0; half < 2; ++half)
            for (int mm = 0; mm < 32; ++mm)
              for (int v = 0; v < 8; ++v)
                i1[((((size_t)ks * 2 + pl) * 2 + half) * 32 + mm) * 8 + v] = p1[pl * n1 + (size_t)(hb * 32 + mm) * C + ks * 16 + half * 8 + v];
      for (int b = 0; b < nb; ++b)
        for (int s2 = 0; s2 < 2; ++s2)
          for (int pl = 0; pl < 2; ++pl)
            for (int half = 0; half < 2; ++half)
              for (int n = 0; n < 32; ++n)
                for (int vh = 0; vh < 2; ++vh)
                  for (int e = 0; e < 4; ++e)
                    i2[((((((size_t)b * 2 + s2) * 2 + pl) * 2 + half) * 32 + n) * 2 + vh) * 4 + e] =
                        p2[pl * n1 + (size_t)(b * 32 + n) * H4 + 32 * hb + 8 * (2 * s2 + vh) + 4 * half + e];
      for (int i = 0; i < 32; ++i) { aux[i] = b1.p[hb * 32 + i]; aux[32 + i] = beta.p[hb * 32 + i]; }
    }
    return upload(img);
  }
  void finish(CW& cw, const std::vector<float>& wt, const std::vector<float>& bias, bool has_bias, CW* as_gemm = nullptr) {
    cw.wt = upload(wt);
    if (has_bias) cw.bias = upload(bias);
    std::vector<uint16_t> planes, blk;
    if (m->arith == 2) cw.w_mul = split2(wt, planes);
    else split3(wt, planes);
    cw.split = upload(planes);
    blocked(planes, cw.N, cw.KH * cw.KW * cw.CinP, cw.KH * cw.KW, blk);
    cw.blk = upload(blk);
    if (as_gemm) {                 // same rows read as one K run of KH*KW*CinP (K order (tap, channel)): blocked image in plain K order
      *as_gemm = cw;
      as_gemm->CinP = cw.KH * cw.KW * cw.CinP;
      as_gemm->KH = as_gemm->KW = 1;
      blocked(planes, cw.N, as_gemm->CinP, 1, blk);
      as_gemm->blk = upload(blk);
    }
  }
  // engine.py::pack_conv: [N,Cin,KH,KW] -> [N][tap][CinP], optional per-row scale (folded BatchNorm)
  void conv(CW& cw, const std::string& wkey, int cin, int kh, int kw, int in_ld, const std::vector<float>* scale,
            const std::vector<float>* bias_v, const std::string& bias_key, CW* as_gemm = nullptr, std::vector<float>* keep_wt = nullptr) {
    const HostT& w = get(wkey);
    if (!w.p) return;
    const int n = (int)(w.n / ((int64_t)cin * kh * kw));
    const int cinp = rup(std::max(in_ld, cin), 16);
    cw.N = n; cw.KH = kh; cw.KW = kw; cw.CinP = cinp;
    std::vector<float> wt((size_t)n * kh * kw * cinp, 0.f);
    for (int o = 0; o < n; ++o)
      for (int c = 0; c < cin; ++c)
        for (int t = 0; t < kh * kw; ++t) {
          float v = w.p[((size_t)o * cin + c) * kh * kw + t];
          if (scale) v = v * (*scale)[o];
          wt[((size_t)o * kh * kw + t) * cinp + c] = v;
        }
    std::vector<float> b;
    bool hb = false;
    if (bias_v) { b = *bias_v; hb = true; }
    else if (!bias_key.empty()) { const HostT& t = get(bias_key); if (t.p) { b.assign(t.p, t.p + t.n); hb = true; } }
    finish(cw, wt, b, hb, as_gemm);
    if (keep_wt) *keep_wt = std::move(wt);
  }
  // engine.py::_pack_embedder (Upsample groups): [Co,Cin,3,3] -> rows (tap, channel) of a 1x1 GEMM on the low-resolution [x | skip] map
  void upconv9(CW& cw, const std::string& wkey, int cin, int cout) {
    const HostT& w = get(wkey);
    if (!w.p) return;
    const int cinp = rup(cin, 16);
    cw.N = 9 * cout; cw.KH = 1; cw.KW = 1; cw.CinP = cinp;
    std::vector<float> wt((size_t)9 * cout * cinp, 0.f);
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c)
        for (int t = 0; t < 9; ++t) wt[((size_t)t * cout + o) * cinp + c] = w.p[((size_t)o * cin + c) * 9 + t];
    finish(cw, wt, {}, false);
  }
  // engine.py::pack_patch_conv: kw pixels of a row are one run of kw*pix_ld floats -> KH x 1 conv with Cin' = kw*pix_ld
  void patch_conv(CW& cw, const std::string& wkey, int cin, int k, int pix_ld, const std::string& bias_key, CW* as_gemm = nullptr) {
    const HostT& w = get(wkey);
    if (!w.p) return;
    const int n = (int)(w.n / ((int64_t)cin * k * k));
    const int run = k * pix_ld, cinp = rup(run, 16);
    cw.N = n; cw.KH = k; cw.KW = 1; cw.CinP = cinp;
    std::vector<float> wt((size_t)n * k * cinp, 0.f);
    for (int o = 0; o < n; ++o)
      for (int c = 0; c < cin; ++c)
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) wt[((size_t)o * k + ky) * cinp + kx * pix_ld + c] = w.p[(((size_t)o * cin + c) * k + ky) * k + kx];
    const HostT& bt = get(bias_key);
    std::vector<float> b;
    if (bt.p) b.assign(bt.p, bt.p + bt.n);
    finish(cw, wt, b, bt.p != nullptr, as_gemm);
  }
  void bn_fold(const std::string& p, std::vector<float>& s, std::vector<float>& b) {          // engine.py::_bn_fold
    const HostT &w = get(p + ".weight"), &bi = get(p + ".bias"), &mu = get(p + ".running_mean"), &var = get(p + ".running_var");
    if (!w.p || !bi.p || !mu.p || !var.p) return;
    s.resize(w.n); b.resize(w.n);
    for (int64_t i = 0; i < w.n; ++i) {
      s[i] = w.p[i] / std::sqrt(var.p[i] + 1e-5f);
      b[i] = bi.p[i] - mu.p[i] * s[i];
    }
  }
  // engine.py::resblock_msg0: c0 of the first bottleneck block split into its latent columns (3x3, K = 9*nlat) and the per-tap
  // message columns (rows (tap, n), K = hidden) from the BN-folded packed weight [N][9][cinp]
  void msg_table_convs(CW& lat, CW& msg, const std::vector<float>& wt, const std::vector<float>& bias, int n, int cinp, int nlat, int hidden) {
    std::vector<float> wl((size_t)n * 9 * nlat), wm((size_t)9 * n * hidden);
    for (int o = 0; o < n; ++o)
      for (int t = 0; t < 9; ++t) {
        const float* src = &wt[((size_t)o * 9 + t) * cinp];
        std::copy(src, src + nlat, &wl[((size_t)o * 9 + t) * nlat]);
        std::copy(src + nlat, src + nlat + hidden, &wm[((size_t)t * n + o) * hidden]);
      }
    lat.N = n; lat.KH = 3; lat.KW = 3; lat.CinP = nlat;
    finish(lat, wl, bias, true);
    msg.N = 9 * n; msg.KH = 1; msg.KW = 1; msg.CinP = hidden;
    finish(msg, wm, {}, false);
  }
  void resblock(RB& rb, const std::string& p, int cin, std::vector<float>* keep_c0 = nullptr, std::vector<float>* keep_b0 = nullptr) {   // engine.py::_pack_resblock
    const HostT& w0 = get(p + ".double_conv.0.weight");
    if (!w0.p) return;
    const int cout = (int)(w0.n / ((int64_t)cin * 9));
    rb.cout = cout;
    std::vector<float> s0, b0, s1, b1;
    bn_fold(p + ".double_conv.1", s0, b0);
    bn_fold(p + ".double_conv.4", s1, b1);
    conv(rb.c0, p + ".double_conv.0.weight", cin, 3, 3, rup(cin, 4), &s0, &b0, "", nullptr, keep_c0);
    if (keep_b0) *keep_b0 = b0;
    conv(rb.c1, p + ".double_conv.3.weight", cout, 3, 3, rup(cout, 4), &s1, &b1, "");
    conv(rb.res, p + ".res_conv.weight", cin, 1, 1, rup(cin, 4), nullptr, nullptr, p + ".res_conv.bias");
  }
};

// ---------------------------------------------------------------- launch sequences
struct Runner {
  vs_model* m;
  char* ws;             // nullptr: counting pass (vs_model_workspace_bytes), no launches
  int64_t cap, used = 0;
  void* st;
  int rc = VS_OK;

  float* alloc(int64_t floats) {
    const int64_t bytes = (floats * 4 + 255) & ~(int64_t)255;
    float* p = ws ? reinterpret_cast<float*>(ws + used) : nullptr;
    used += bytes;
    if (ws && used > cap) rc = VS_ERR_BAD_ARG;
    return p;
  }
  Act act(int B, int H, int W, int C, int ld = 0) {
    if (!ld) ld = rup(C, 4);
    return Act{alloc((int64_t)B * H * W * ld), B, H, W, C, ld};
  }
  bool live() const { return ws && rc == VS_OK; }
  void chk(int code) { if (rc == VS_OK && code != VS_OK) rc = code; }

  // engine.py::conv with tile_hint = 0 (the library's static heuristics) and the shape-only K-split rules
  void conv(const Act& x, const CW& w, const Act& out, int stride = 1, int pad = 0, int pad_mode = VS_PAD_ZERO, int act_ = VS_ACT_NONE,
            int out_coff = 0, int n_store = -1, const Act* res = nullptr, const Act* in2 = nullptr, const CW* w2 = nullptr,
            const float* a_scale = nullptr, int64_t a_scale_ld = 0, const float* a_shift = nullptr, const int* geom = nullptr,
            float* sumsq = nullptr, int cin_first = 0, bool pre_table = false, float a_mul_override = 0.f,
            const float* grn_part = nullptr, const float* grn_gamma = nullptr, int grn_hw = 0, int grn_c4 = 0) {
    vs_conv_desc_t d;
    std::memset(&d, 0, sizeof(d));
    int sh = stride, sw = stride, ph = pad, pw = pad, H = x.H, W = x.W, cin = x.ld;
    int64_t sx = x.ld;
    if (geom) { W = geom[0]; sx = geom[1]; cin = geom[2]; sh = geom[3]; sw = geom[4]; ph = geom[5]; pw = geom[6]; }
    if (cin_first) cin = cin_first;             // only the first channels of every pixel (pixel stride stays x.ld)
    d.in = x.p;
    d.in_sb = (int64_t)x.H * x.W * x.ld; d.in_sy = (int64_t)x.W * x.ld; d.in_sx = sx;
    d.B = x.B; d.H = H; d.W = W; d.Cin = cin;
    d.KH = w.KH; d.KW = w.KW; d.SH = sh; d.SW = sw; d.PH = ph; d.PW = pw; d.pad_mode = pad_mode;
    d.Ho = out.H; d.Wo = out.W;
    d.wt = w.wt; d.CinP = w.CinP; d.N = w.N;
    d.a_scale = a_scale; d.a_scale_ld = a_scale_ld; d.a_shift = a_shift;
    d.bias = w.bias; d.act = act_;
    d.n_store = n_store >= 0 ? n_store : (out_coff == 0 ? out.ld - out_coff : w.N);
    if (res) { d.res = res->p; d.res_ld = res->ld; }
    if (in2) { d.in2 = in2->p; d.in2_ld = in2->ld; d.Cin2 = in2->ld; d.Cin2P = w2->CinP; d.wt2 = w2->wt; d.bias2 = w2->bias; }
    d.out = out.p; d.out_ld = out.ld; d.out_coff = out_coff; d.tile_hint = 0;
    d.wt_split = w.split; d.wt_blk = w.blk;
    const float am = a_mul_override > 0.f ? a_mul_override : ((a_scale && !pre_table) ? A_MUL_GRN : A_MUL);
    d.arith = m->arith; d.a_mul = am; d.acc_mul = 1.f / (am * w.w_mul);
    if (in2) { d.wt2_split = w2->split; d.wt2_blk = w2->blk; d.acc_mul2 = 1.f / (am * w2->w_mul); }
    int split_k = 1;
    const bool dense_rows = d.in_sy == (int64_t)d.W * d.in_sx && d.in_sb == (int64_t)d.H * d.in_sy;
    const bool gemm_pc = d.KH == 1 && d.KW == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && !in2 && d.Ho == d.H && d.Wo == d.W &&
                         d.Cin % 32 == 0 && d.CinP == d.Cin && dense_rows && (!a_scale || d.H * d.W >= 128 || d.H * d.W == 64);
    const bool patch_pc = d.KH == 3 && d.KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && d.Ho == d.H && d.Wo == d.W && !a_scale &&
                          d.W % 16 == 0 && d.H % 8 == 0;
    if (pre_table) {                            // border-class table added before bias + activation (engine.py::resblock_msg0)
      d.tile_hint = (d.N % 192 == 0 ? (VS_CONV_TILE_HI | 0) : 15) | VS_CONV_PRE;
    } else if (sumsq) {
      d.sumsq_part = sumsq;
    } else if (gemm_pc && a_scale && m->arith == 2 && d.N % 96 == 0 && d.H * d.W >= 128 && d.CinP <= 3072 &&      // (K <= 3072: the tile's GRN rows live in LDS)
               (((int64_t)d.B * d.H * d.W + 127) / 128) * ((d.N + 127) / 128) < 256 && (((int64_t)d.B * d.H * d.W + 127) / 128) * ((d.N + 95) / 96) >= 200) {
      d.tile_hint = VS_CONV_TILE_HI | 10;                     // engine.py: pwconv2 on 128 x 96 tiles with the whole K per workgroup (tile 26) instead of K slices
    } else if (gemm_pc) {                                       // engine.py::_split_k_rule
      const int64_t rows = (int64_t)d.B * d.H * d.W;
      const int64_t blocks = ((rows + 127) / 128) * ((d.N + 127) / 128);
      const int pairs = d.CinP / 32;
      int sk = 1;
      while (blocks * sk < 256 && pairs % (sk * 2) == 0 && pairs / (sk * 2) >= 4) sk *= 2;
      if (!a_scale || d.CinP / sk <= 3072) split_k = sk;      // engine.py::conv: the GRN rows of a K SLICE are staged in LDS; longer slices stay unsplit
    } else if (patch_pc && d.N >= 128 && d.CinP >= 128) {      // engine.py::_split_k_rule_patch
      const int64_t blocks = (int64_t)d.B * (d.H / 8) * (d.W / 16) * (d.N % 192 == 0 ? (d.N + 191) / 192 : (d.N + 127) / 128);
      const int spt = d.CinP / 16;
      const int cands[5] = {2, 3, 4, 6, 8};
      for (int cand : cands) {
        if (blocks * split_k >= 192) break;
        if (spt % cand == 0 && spt / cand >= 2) split_k = cand;
      }
    }
    if (split_k > 1) {
      const int ws_ld = rup(w.N, 4);
      const int slices = split_k + ((patch_pc && in2) ? 1 : 0);
      d.splitk_ws = alloc((int64_t)slices * out.rows() * ws_ld);
      d.splitk_ld = ws_ld;
      d.split_k = split_k;
      d.tile_hint = d.KH == 3 ? (d.N % 192 == 0 ? VS_CONV_TILE_HI | 0 : 15) : (VS_CONV_TILE_HI | (d.N % 192 == 0 ? 2 : 1));
    }
    if (grn_part) {
      // engine.py::conv(grn_fold=...): GRN's finish inside the wave-specialised GEMM (ABI v3: explicit tile 17 / 18 / 26, <= 16 partial rows per frame,
      // K == the partials' channel count) or as the separate launch in front of it; the same scale values either way
      const int t = (d.tile_hint & 0xf) + ((d.tile_hint & VS_CONV_TILE_HI) ? 16 : 0);
      if ((t == 17 || t == 18 || t == 26) && grn_hw % 32 == 0 && grn_hw / 32 <= 16 && d.CinP == grn_c4 && a_scale_ld == grn_c4) {
        d.grn_part = grn_part; d.grn_gamma = grn_gamma; d.grn_nchunk = grn_hw / 32;
      } else if (live()) {
        chk(vs_grn_scale_from_partials(grn_part, d.B, grn_hw, grn_c4, grn_gamma, const_cast<float*>(a_scale), a_scale_ld, st));
      }
    }
    if (live()) chk(vs_conv_gemm(&d, st));
  }
  void layernorm(const Act& x, const float* w, const float* b, const Act& out, int act_ = VS_ACT_NONE) {
    if (live()) chk(vs_layernorm_act(x.p, x.rows(), x.C, x.ld, w, b, 1e-6f, act_, out.p, out.ld, st));
  }
  Act resblock(const Act& x, const RB& p, const Act* out_in = nullptr) {                      // engine.py::resblock
    const int tk = p.cout == 16 ? 16 : 32;
    if ((p.cout == 16 || (p.cout == 32 && m->arith == 2)) && x.ld <= tk && p.c0.CinP == tk && p.c1.CinP == tk && p.res.CinP == tk &&
        (!out_in || out_in->ld == p.cout) && vs_resblock_thin_supported(x.ld, p.cout, p.cout)) {   // engine.py::resblock_thin: t stays on chip
      Act out = out_in ? *out_in : act(x.B, x.H, x.W, p.cout);
      vs_resblock_thin_desc_t d;
      std::memset(&d, 0, sizeof(d));
      d.x = x.p; d.x_ld = x.ld; d.B = x.B; d.H = x.H; d.W = x.W; d.Cin = x.ld;
      d.w0_split = p.c0.split; d.w1_split = p.c1.split; d.wr_split = p.res.split;
      d.b0 = p.c0.bias; d.b1 = p.c1.bias; d.br = p.res.bias;
      d.arith = m->arith; d.Cout = p.cout; d.a_mul = A_MUL;
      d.acc_mul0 = 1.f / (A_MUL * p.c0.w_mul); d.acc_mul1 = 1.f / (A_MUL * p.c1.w_mul); d.acc_mulr = 1.f / (A_MUL * p.res.w_mul);
      d.out = out.p; d.out_ld = out.ld;
      if (live()) chk(vs_resblock_thin(&d, st));
      return out;
    }
    Act t = act(x.B, x.H, x.W, p.cout);
    conv(x, p.c0, t, 1, 1, VS_PAD_ZERO, VS_ACT_RELU);
    Act out = out_in ? *out_in : act(x.B, x.H, x.W, p.cout);
    conv(t, p.c1, out, 1, 1, VS_PAD_ZERO, VS_ACT_RELU, 0, out.ld != rup(p.cout, 4) ? p.cout : -1, nullptr, &x, &p.res);
    return out;
  }
  // 1x1 GEMM on operand planes (tile code 24, gemm_pl.hip): out = act(in_pl x w + bias) (+ res), optional GRN partials / K split
  void gemm_pl(int B, int H, int W, const CW& w, const void* in_pl, const Act& out, int act_, const Act* res, float* sumsq, int split_k,
               float am = A_MUL, int sumsq_hw = 0) {
    vs_conv_desc_t d;
    std::memset(&d, 0, sizeof(d));
    d.B = B; d.H = H; d.W = W; d.Cin = w.CinP; d.KH = d.KW = 1; d.SH = d.SW = 1; d.Ho = H; d.Wo = W;
    d.wt = w.wt; d.CinP = w.CinP; d.N = w.N; d.bias = w.bias; d.act = act_;
    d.wt_split = w.split; d.wt_blk = w.blk; d.arith = 2; d.a_mul = am; d.acc_mul = 1.f / (am * w.w_mul);
    d.in_pl = in_pl;
    d.out = out.p; d.out_ld = out.ld; d.n_store = out.ld;
    if (res) { d.res = res->p; d.res_ld = res->ld; }
    d.sumsq_part = sumsq;
    d.sumsq_hw = sumsq_hw;
    if (split_k > 1) {
      const int ws_ld = rup(w.N, 4);
      d.splitk_ws = alloc((int64_t)split_k * out.rows() * ws_ld);
      d.splitk_ld = ws_ld;
      d.split_k = split_k;
    }
    d.tile_hint = VS_CONV_TILE_HI | 8;
    if (live()) chk(vs_conv_gemm(&d, st));
  }
  // one launch of the all-DMA 3x3 kernel on operand planes (tile code 22, conv3x3_pl.hip); out == nullptr: planes only
  void conv_pl(int B, int H, int W, const CW& w, const void* in_pl, void* out_pl, const Act* out, const CW* w2 = nullptr,
               const void* in2_pl = nullptr, const float* pre_table = nullptr, int64_t pre_ld = 0) {
    vs_conv_desc_t d;
    std::memset(&d, 0, sizeof(d));
    d.B = B; d.H = H; d.W = W; d.Cin = w.CinP; d.KH = d.KW = 3; d.SH = d.SW = 1; d.PH = d.PW = 1; d.pad_mode = VS_PAD_ZERO; d.Ho = H; d.Wo = W;
    d.wt = w.wt; d.CinP = w.CinP; d.N = w.N; d.bias = w.bias; d.act = VS_ACT_RELU;
    d.wt_split = w.split; d.wt_blk = w.blk; d.arith = 2; d.a_mul = A_MUL; d.acc_mul = 1.f / (A_MUL * w.w_mul);
    d.in_pl = in_pl; d.out_pl = out_pl;
    if (w2) {
      d.in2_pl = in2_pl; d.Cin2 = d.Cin2P = w2->CinP; d.wt2 = w2->wt; d.bias2 = w2->bias; d.wt2_split = w2->split; d.wt2_blk = w2->blk;
      d.acc_mul2 = 1.f / (A_MUL * w2->w_mul);
    }
    d.n_store = w.N;
    if (out) { d.out = out->p; d.out_ld = out->ld; d.n_store = out->ld != rup(w.N, 4) ? w.N : out->ld; }
    d.tile_hint = VS_CONV_TILE_HI | 6;
    if (pre_table) { d.a_scale = pre_table; d.a_scale_ld = pre_ld; d.tile_hint |= VS_CONV_PRE; }      // border-class table in the epilogue (first block)
    if (live()) chk(vs_conv_gemm(&d, st));
  }
  // engine.py::bottleneck_planes: ResnetBlocks j0.. from the fp32 activation x with every intermediate tensor as f16 operand planes
  Act bottleneck_planes(const Act& x, int j0, const Act* last_out, void* xpl_in = nullptr) {
    const int B = x.B, H = x.H, W = x.W, C = x.C;
    const int64_t pl_floats = x.rows() * C;               // 2 planes x rows x C f16
    void* xpl = xpl_in ? xpl_in : alloc(pl_floats);       // (the first block may hand its output over as planes already)
    void* tpl = alloc(pl_floats);
    void* ypl = alloc(pl_floats);
    if (!xpl_in && live()) chk(vs_to_planes(x.p, x.rows(), C, x.ld, A_MUL, xpl, st));
    Act out{};
    const int nb = (int)m->bottleneck.size();
    for (int j = j0; j < nb; ++j) {
      const RB& p = m->bottleneck[j];
      conv_pl(B, H, W, p.c0, xpl, tpl, nullptr);
      if (j == nb - 1) {
        out = last_out ? *last_out : act(B, H, W, C);
        conv_pl(B, H, W, p.c1, tpl, nullptr, &out, &p.res, xpl);
      } else {
        conv_pl(B, H, W, p.c1, tpl, ypl, nullptr, &p.res, xpl);
        std::swap(xpl, ypl);
      }
    }
    return out;
  }
  bool planes_ok(const Act& x, int j0) const {            // engine.py::_planes_ok
    if (m->arith != 2 || x.ld != x.C || x.C % 192 || x.H % 16 || x.W % 16 || (int64_t)x.B * (x.H / 16) * (x.W / 16) * (x.C / 192) < 200) return false;
    for (size_t j = j0; j < m->bottleneck.size(); ++j) {
      const RB& p = m->bottleneck[j];
      if (p.cout != x.C || p.c0.CinP != x.C || p.res.CinP != x.C) return false;
    }
    return j0 < (int)m->bottleneck.size();
  }
  // engine.py::embedder_forward: key frames (NHWC, ld 4, mapped to [-1,1]) -> delta [B][out_ch][S][S]
  float* embedder(const Act& x, const int32_t* msgs, int n_msgs, int& Sh, int& Sw) {
    const vs_model_cfg_t& c = m->c;
    const int B = x.B, nlev = (int)m->zc.size() - 1;
    std::vector<Act> hid;
    hid.push_back(resblock(x, m->inc));
    for (int i = 0; i < nlev; ++i) {
      const Act src = hid.back();
      const int Ho = (src.H - 1) / 2 + 1, Wo = (src.W - 1) / 2 + 1;
      Act dwn = act(B, Ho, Wo, m->zc[i + 1]);
      conv(src, m->down_conv[i], dwn, 2, 1);
      if (i == nlev - 1) {
        Act h3 = act(B, Ho, Wo, m->bott);
        resblock(dwn, m->down_rb[i], &h3);
        hid.push_back(h3);
      } else {
        hid.push_back(resblock(dwn, m->down_rb[i]));
      }
    }
    Act h3 = hid.back();
    float* lat = alloc((int64_t)n_msgs * c.hidden);
    if (live()) chk(vs_msg_latent(m->table, msgs, n_msgs, c.nbits, c.hidden, lat, st));
    if (live()) chk(vs_broadcast_channels(lat, n_msgs, c.hidden, h3.p, B, h3.H * h3.W, h3.ld, m->zc.back(), st));
    Act cur = h3;
    // Upsample group k on the low-resolution map: its [x | skip] concat buffer is allocated before x's producer runs, so that the
    // producer (last bottleneck block / previous up block) writes columns [0, C) itself (engine.py::embedder_forward)
    auto fused_ok = [&](int k, int c1, int c2) -> bool {
      return m->ups[k].lowres && vs_upconv_fused_preferred(c1, c2, m->ups[k].gemm.N / 9) && m->ups[k].gemm.CinP == c1 + c2;
    };
    auto lowres_cat = [&](int k, const Act& like, Act& view) -> bool {
      if (k >= nlev || !m->ups[k].lowres) return false;
      const Act& skip = hid[nlev - k];
      if (fused_ok(k, like.C, skip.C)) return false;       // the one-kernel form reads x and skip themselves
      Act lc = act(B, like.H, like.W, like.C + skip.C);
      view = Act{lc.p, B, like.H, like.W, like.C, lc.ld};
      return true;
    };
    void* cur_pl = nullptr;
    for (int j = 0; j < c.num_blocks; ++j) {
      Act view{};
      const bool direct = j == c.num_blocks - 1 && lowres_cat(0, cur, view);
      const RB& rb = m->bottleneck[j];
      if (j == 0 && !direct && m->b0_table && h3.ld == h3.C && rb.c0.CinP == h3.C && h3.W % 16 == 0 && h3.H % 8 == 0 &&
          (int64_t)B * (h3.H / 8) * (h3.W / 16) >= 128) {                        // engine.py::resblock_msg0
        const int N9 = m->b0_msg.N, N = rb.cout;
        Act P = act(n_msgs, 1, 1, N9);
        conv(Act{lat, n_msgs, 1, 1, c.hidden, c.hidden}, m->b0_msg, P);
        float* table = alloc((int64_t)n_msgs * 9 * N);
        if (live()) chk(vs_msg_pre(P.p, n_msgs, N, table, st));
        const Act nxt{nullptr, B, h3.H, h3.W, N, N};
        if (c.num_blocks > 1 && N == h3.C && m->zc.back() % 16 == 0 && planes_ok(nxt, 1)) {
          // engine.py::resblock_msg0(planes_out=True), round 5: the block on the all-DMA planes kernel, output handed to the chain as planes
          const int nlat = m->zc.back();
          void* latpl = alloc(h3.rows() * nlat);
          void* h3pl = alloc(h3.rows() * h3.C);
          void* tpl = alloc(h3.rows() * N);
          void* xpl = alloc(h3.rows() * N);
          if (live()) {
            chk(vs_to_planes(h3.p, h3.rows(), nlat, h3.ld, A_MUL, latpl, st));
            chk(vs_to_planes(h3.p, h3.rows(), h3.C, h3.ld, A_MUL, h3pl, st));
          }
          conv_pl(B, h3.H, h3.W, m->b0_lat, latpl, tpl, nullptr, nullptr, nullptr, table, n_msgs == 1 ? 0 : 9 * (int64_t)N);
          conv_pl(B, h3.H, h3.W, rb.c1, tpl, xpl, nullptr, &rb.res, h3pl);
          cur = nxt;
          cur_pl = xpl;
          continue;
        }
        Act t = act(B, h3.H, h3.W, N);
        conv(h3, m->b0_lat, t, 1, 1, VS_PAD_ZERO, VS_ACT_RELU, 0, -1, nullptr, nullptr, nullptr, table, n_msgs == 1 ? 0 : 9 * (int64_t)N, nullptr,
             nullptr, nullptr, m->zc.back(), true);
        Act out = act(B, h3.H, h3.W, N);
        conv(t, rb.c1, out, 1, 1, VS_PAD_ZERO, VS_ACT_RELU, 0, -1, nullptr, &h3, &rb.res);
        cur = out;
        continue;
      }
      if (j >= 1 && planes_ok(cur, j)) {                 // the rest of the chain on operand planes
        Act v2 = view;
        const bool d2 = j == c.num_blocks - 1 ? direct : lowres_cat(0, cur, v2);
        cur = bottleneck_planes(cur, j, d2 ? &v2 : nullptr, cur_pl);
        break;
      }
      cur = resblock(cur, rb, direct ? &view : nullptr);
    }
    for (int k = 0; k < nlev; ++k) {                   // skips are popped deepest first; the first one is the [latent | message] map itself
      const Act skip = hid.back();
      hid.pop_back();
      const Up& up = m->ups[k];
      Act ln{};
      if (up.lowres && cur.ld == cur.C && fused_ok(k, cur.C, skip.C)) {
        const int co = up.gemm.N / 9;
        ln = act(B, 2 * cur.H, 2 * cur.W, co);
        if (live()) chk(vs_upconv_fused(cur.p, cur.C, cur.ld, skip.p, skip.C, skip.ld, 0.70710678118654752440f, up.gemm.split, B, cur.H, cur.W, co,
                                        up.lnw, up.lnb, 1e-6f, VS_ACT_RELU, ln.p, ln.ld, m->arith, A_MUL, 1.f / (A_MUL * up.gemm.w_mul), st));
      } else if (up.lowres) {
        const int co = up.gemm.N / 9;
        const bool direct = cur.ld == cur.C + skip.C;
        Act lc = direct ? Act{cur.p, B, cur.H, cur.W, cur.C + skip.C, cur.ld} : act(B, cur.H, cur.W, cur.C + skip.C);
        if (live()) chk(vs_cat2_scale(direct ? nullptr : cur.p, cur.C, cur.ld, skip.p, skip.C, skip.ld, 0.70710678118654752440f, lc.rows(), lc.p, lc.ld, st));
        Act z = act(B, lc.H, lc.W, 9 * co);
        conv(lc, up.gemm, z);
        ln = act(B, 2 * lc.H, 2 * lc.W, co);
        if (live()) chk(vs_upconv_gather_ln(z.p, z.ld, B, lc.H, lc.W, co, up.lnw, up.lnb, 1e-6f, VS_ACT_RELU, ln.p, ln.ld, st));
      } else {
        Act cat = act(B, 2 * cur.H, 2 * cur.W, cur.C + skip.C);
        if (live()) chk(vs_upcat2x(cur.p, cur.C, cur.ld, skip.p, skip.C, skip.ld, 0.70710678118654752440f, B, cur.H, cur.W, cat.p, cat.ld, st));
        Act cv = act(B, cat.H, cat.W, up.conv.N);
        conv(cat, up.conv, cv, 1, 1, VS_PAD_REFLECT);
        ln = act(B, cat.H, cat.W, up.conv.N);
        layernorm(cv, up.lnw, up.lnb, ln, VS_ACT_RELU);
      }
      Act view{};
      const bool direct = lowres_cat(k + 1, Act{nullptr, B, ln.H, ln.W, up.rb.cout, 0}, view);
      cur = resblock(ln, up.rb, direct ? &view : nullptr);
    }
    float* delta = alloc((int64_t)B * c.out_ch * cur.H * cur.W);
    if (live()) chk(vs_outc_tanh(cur.p, (int64_t)cur.H * cur.W, B, cur.C, cur.ld, m->outc_w, m->outc_b, c.out_ch, c.last_tanh ? 1 : 0, delta, st));
    Sh = cur.H; Sw = cur.W;
    return delta;
  }
  // engine.py::extractor_forward: frames (NHWC ld 4, mapped to [-1,1]) -> logits [B][1+nbits]
  void extractor(const Act& x, float* logits) {
    const vs_model_cfg_t& c = m->c;
    const int B = x.B, s = c.stem_stride;
    int Ho = (x.H - 4) / s + 1, Wo = (x.W - 4) / s + 1;
    Act cur = act(B, Ho, Wo, c.dims[0], xld(c.dims[0]));
    const int co = c.dims[0];
    static const bool stem_fused = [] { const char* e = getenv("VIDEOSEAL_STEM_FUSED"); return e && e[0] == '1'; }();      // (measured neutral: opt-in, as engine.py)
    if (stem_fused && x.ld == 4 && cur.ld == co && (co == 64 || co == 96 || co == 128) && m->stem.bias) {
      // engine.py (round 6): patchify conv + LayerNorm in one kernel on the vector ALUs
      if (live()) chk(vs_stem_conv_ln(x.p, B, x.H, x.W, s, m->stem.wt, m->stem.bias, m->stem_lnw, m->stem_lnb, 1e-6f, co, cur.p, cur.ld, st));
    } else {
      Act t = act(B, Ho, Wo, c.dims[0]);
      { const int geom[7] = {Wo, s * 4, 16, s, 1, 0, 0}; conv(x, m->stem, t, 1, 0, VS_PAD_ZERO, VS_ACT_NONE, 0, -1, nullptr, nullptr, nullptr, nullptr, 0, nullptr, geom); }
      layernorm(t, m->stem_lnw, m->stem_lnb, cur);
    }
    for (int sti = 0; sti < 4; ++sti) {
      if (sti > 0) {
        const Down& dn = m->down[sti - 1];
        Ho = cur.H / 2; Wo = cur.W / 2;
        if (cur.ld == cur.C && cur.C % 8 == 0 && dn.gemm.CinP == 4 * cur.C && Ho > 0 && Wo > 0) {
          // engine.py (round 6): LayerNorm straight into the 2 x 2 patch matrix, the conv as a dense GEMM
          Act lnp = act(B, Ho, Wo, 4 * cur.C, 4 * cur.C);
          Act nxt = act(B, Ho, Wo, c.dims[sti], xld(c.dims[sti]));
          if (live()) chk(vs_layernorm_patch2x2(cur.p, B, cur.H, cur.W, cur.C, cur.ld, dn.lnw, dn.lnb, 1e-6f, lnp.p, st));
          conv(lnp, dn.gemm, nxt);
          cur = nxt;
        } else {
          Act ln = act(B, cur.H, cur.W, cur.C, cur.ld);
          layernorm(cur, dn.lnw, dn.lnb, ln);
          Act nxt = act(B, Ho, Wo, c.dims[sti], xld(c.dims[sti]));
          const int geom[7] = {Wo, 2 * ln.ld, 2 * ln.ld, 2, 1, 0, 0};
          conv(ln, dn.conv, nxt, 1, 0, VS_PAD_ZERO, VS_ACT_NONE, 0, -1, nullptr, nullptr, nullptr, nullptr, 0, nullptr, geom);
          cur = nxt;
        }
      }
      const int Cc = c.dims[sti], HW = cur.H * cur.W;
      Act tn = act(B, cur.H, cur.W, Cc, xld(Cc));
      Act hh = act(B, cur.H, cur.W, 4 * Cc, xld(4 * Cc));
      float* part = alloc((int64_t)((HW + 63) / 64) * B * 4 * Cc);
      float* part32 = alloc((int64_t)B * ((HW + 31) / 32) * 4 * Cc);
      float* scale = alloc((int64_t)B * hh.ld + 16);
      float* parts = (HW % 32 != 0 && HW >= 32) ? alloc((((int64_t)B * HW + 31) / 32) * 2 * 4 * Cc) : nullptr;
      // engine.py::extractor_forward: the GEMMs on operand planes where the 256-row tiles fill the chip (pwconv1), resp. for long K (pwconv2)
      const int64_t rows = cur.rows();
      const CW& pw1w = m->stages[sti].empty() ? m->stem : m->stages[sti][0].pw1;
      const bool pl1 = m->arith == 2 && ((rows + 255) / 256) * ((4 * Cc + 191) / 192) >= 200 && pw1w.CinP % 16 == 0 && !m->stages[sti].empty();
      bool pl2 = false;
      int sk2 = 1;
      if (m->arith == 2 && hh.ld >= 2048 && hh.ld % 16 == 0) {
        const int64_t tiles2 = ((rows + 255) / 256) * ((Cc + 191) / 192);
        const int steps2 = hh.ld / 16;
        while (tiles2 * sk2 < 200 && steps2 / (sk2 * 2) >= 24 && steps2 % (sk2 * 2) == 0) sk2 *= 2;
        pl2 = tiles2 * sk2 >= 128;
        if (pl2 && rows <= 2048 && HW % 64 == 0 && hh.ld % 32 == 0) {     // engine.py (round 5): small-M layers on the wave-specialised GEMM with the GRN A path
          int skp = 1;
          const int64_t blocks = ((rows + 127) / 128) * ((Cc + 127) / 128);
          const int pairs = (int)(hh.ld / 32);
          while (blocks * skp < 256 && pairs % (skp * 2) == 0 && pairs / (skp * 2) >= 4) skp *= 2;
          if (hh.ld / skp <= 3072) pl2 = false;
        }
      }
      else if (m->arith == 2 && hh.ld % 16 == 0 && HW % 64 != 0 && ((rows + 255) / 256) * ((Cc + 191) / 192) >= 200) {
        pl2 = true; sk2 = 1;      // engine.py (round 6): odd frames pay a pass over h anyway -- make it the planes conversion
      }
      // engine.py::_extractor_forward: pwconv1 -> GELU -> GRN -> pwconv2 with h on chip (statistics pass, scale, apply pass in place on cur)
      const bool fused = m->arith == 2 && !m->stages[sti].empty() && m->stages[sti][0].fuse && pw1w.CinP == Cc && vs_cnx_block_supported(Cc, rows, HW);
      void* tnpl = (pl1 || fused) ? alloc(rows * pw1w.CinP) : nullptr;
      void* hpl = pl2 ? alloc(rows * hh.ld) : nullptr;
      for (const Blk& blk : m->stages[sti]) {
        if (fused && blk.fuse) {
          if (live()) {
            chk(vs_dwconv7_ln_planes(cur.p, B, cur.H, cur.W, Cc, cur.ld, blk.wdw, blk.bdw, blk.lnw, blk.lnb, 1e-6f, A_MUL, pw1w.CinP, tnpl, st));
            const float am1 = 1.f / (A_MUL * blk.fm1), am2 = 1.f / (A_MUL_GRN * blk.fm2);
            chk(vs_cnx_block(tnpl, blk.fuse, Cc, rows, HW, 1, am1, am2, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, part32, st));
            chk(vs_grn_scale_from_partials(part32, B, HW, 4 * Cc, blk.gamma, scale, hh.ld, st));
            chk(vs_cnx_block(tnpl, blk.fuse, Cc, rows, HW, 0, am1, am2, scale, hh.ld, blk.pw2.bias, cur.p, cur.ld, cur.p, cur.ld, nullptr, st));
          }
          continue;
        }
        if (pl1) {
          if (live()) chk(vs_dwconv7_ln_planes(cur.p, B, cur.H, cur.W, Cc, cur.ld, blk.wdw, blk.bdw, blk.lnw, blk.lnb, 1e-6f, A_MUL, pw1w.CinP, tnpl, st));
        } else if (live()) {
          chk(vs_dwconv7_ln(cur.p, B, cur.H, cur.W, Cc, cur.ld, blk.wdw, blk.bdw, blk.lnw, blk.lnb, 1e-6f, tn.p, tn.ld, st));
        }
        if (HW % 32 == 0) {
          if (pl1) gemm_pl(B, cur.H, cur.W, blk.pw1, tnpl, hh, VS_ACT_GELU, nullptr, part32, 1);
          else conv(tn, blk.pw1, hh, 1, 0, VS_PAD_ZERO, VS_ACT_GELU, 0, -1, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, part32);
          // (the finish is deferred to pwconv2's conv() where that launch can fold it, engine.py)
          if (!(HW % 64 == 0 && !pl2) && live()) chk(vs_grn_scale_from_partials(part32, B, HW, 4 * Cc, blk.gamma, scale, hh.ld, st));
        } else if (pl1 && HW >= 32) {      // engine.py (round 6): straddling 32-row-group partials from the planes GEMM's epilogue
          gemm_pl(B, cur.H, cur.W, blk.pw1, tnpl, hh, VS_ACT_GELU, nullptr, parts, 1, A_MUL, HW);
          if (live()) chk(vs_grn_scale_from_straddle_partials(parts, B, HW, 4 * Cc, blk.gamma, scale, hh.ld, st));
        } else {
          if (pl1) gemm_pl(B, cur.H, cur.W, blk.pw1, tnpl, hh, VS_ACT_GELU, nullptr, nullptr, 1);
          else conv(tn, blk.pw1, hh, 1, 0, VS_PAD_ZERO, VS_ACT_GELU);
          if (live()) chk(vs_grn_scale(hh.p, B, HW, 4 * Cc, hh.ld, blk.gamma, part, scale, st));
        }
        if (pl2) {
          if (live()) chk(vs_to_planes_affine(hh.p, hh.rows(), hh.ld, hh.ld, A_MUL_GRN, scale, hh.ld, blk.beta, HW, hpl, st));
          gemm_pl(B, cur.H, cur.W, blk.pw2, hpl, cur, VS_ACT_NONE, &cur, nullptr, sk2, A_MUL_GRN);
        } else if (HW % 64 == 0) {
          const bool defer = HW % 32 == 0;
          conv(hh, blk.pw2, cur, 1, 0, VS_PAD_ZERO, VS_ACT_NONE, 0, -1, &cur, nullptr, nullptr, scale, hh.ld, blk.beta, nullptr, nullptr, 0, false, 0.f,
               defer ? part32 : nullptr, blk.gamma, HW, 4 * Cc);
        } else {
          if (live()) chk(vs_grn_apply(hh.p, B, HW, 4 * Cc, hh.ld, scale, hh.ld, blk.beta, st));
          conv(hh, blk.pw2, cur, 1, 0, VS_PAD_ZERO, VS_ACT_NONE, 0, -1, &cur, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, false, A_MUL_GRN);
        }
      }
    }
    Act hc = act(B, cur.H, cur.W, c.dims[3]);
    if (cur.rows() <= 4096 && cur.H > 1 && cur.W > 1 && m->head.CinP == cur.ld) {       // engine.py::extractor_forward
      Act cols{alloc(cur.rows() * 9 * cur.ld), B, cur.H, cur.W, 9 * cur.ld, 9 * cur.ld};
      if (live()) chk(vs_im2col3x3(cur.p, B, cur.H, cur.W, cur.ld, VS_PAD_REFLECT, cols.p, st));
      conv(cols, m->head_gemm, hc);
    } else {
      conv(cur, m->head, hc, 1, 1, VS_PAD_REFLECT);
    }
    Act hl = act(B, cur.H, cur.W, c.dims[3]);
    layernorm(hc, m->head_lnw, m->head_lnb, hl, VS_ACT_GELU);
    if (live()) chk(vs_pool_linear(hl.p, B, hl.H * hl.W, hl.C, hl.ld, m->lin_w, m->lin_b, c.nbits + 1, logits, st));
  }
  // model.py::_embed_frames_eager
  void resize(const void* imgs, bool u8, int F, int H, int W, int S, int antialias, float* rgb, float mul, float add, float* key, int step,
              const float* ymat) {
    if (!live()) return;
    if (u8) chk(vs_resize_pre_u8(static_cast<const unsigned char*>(imgs), F, H, W, S, S, antialias, rgb, mul, add, key, step, ymat, st));
    else chk(vs_resize_pre(static_cast<const float*>(imgs), F, 3, H, W, S, S, antialias, rgb, mul, add, key, step, ymat, st));
  }
  void embed(const void* imgs, bool u8, const int32_t* msgs, int n_msgs, int F, int H, int W, int step, int video_mode, int lowres, int antialias,
             void* out, float* preds_w) {
    const vs_model_cfg_t& c = m->c;
    const int S = c.img_size, nk = (F + step - 1) / step;
    const bool att = c.attenuate != 0;
    Act rgb{nullptr, F, S, S, 3, 4}, key = act(nk, S, S, c.in_ch, 4);
    if (att && lowres) rgb = act(F, S, S, 3, 4);
    resize(imgs, u8, F, H, W, S, antialias, (att && lowres) ? rgb.p : nullptr, 1.0f, 0.0f, key.p, step, c.yuv ? m->ymat : nullptr);
    int Sh = 0, Sw = 0;
    float* delta = embedder(key, msgs, n_msgs, Sh, Sw);
    float* hmap = nullptr;
    if (att && lowres) {
      hmap = alloc((int64_t)F * S * S);
      if (live()) chk(vs_jnd_heatmap(rgb.p, F, S, S, (int64_t)S * S * 4, 1, (int64_t)S * 4, 4, m->taps43, hmap, st));
    }
    vs_tail_desc_t d;
    std::memset(&d, 0, sizeof(d));
    d.imgs = imgs; d.out = out; d.preds_w = preds_w; d.delta = delta; d.hmap_lowres = hmap; d.taps43 = m->taps43;
    d.F = F; d.H = H; d.W = W; d.S_h = Sh; d.S_w = Sw; d.Cd = c.out_ch;
    d.step = step; d.video_mode = video_mode; d.total_key = nk;
    d.attenuate = att ? 1 : 0; d.clamp = c.clamp; d.antialias = antialias;
    d.scaling_i = c.scaling_i; d.scaling_w = c.scaling_w;
    d.io_u8 = u8 ? 1 : 0;
    if (live()) chk(vs_embed_tail(&d, st));
  }
  void detect(const void* imgs, bool u8, int F, int H, int W, int antialias, float* logits) {
    const int S = m->c.img_size;
    Act rgb = act(F, S, S, 3, 4);
    resize(imgs, u8, F, H, W, S, antialias, rgb.p, 2.0f, -1.0f, nullptr, 1, nullptr);
    extractor(rgb, logits);
  }
};

}  // namespace

extern "C" int vs_model_create(const vs_model_cfg_t* cfg, const vs_tensor_t* tensors, int ntensors, vs_model_t** out) {
  VS_REQUIRE(cfg && tensors && ntensors > 0 && out);
  VS_REQUIRE(cfg->nlev >= 1 && cfg->nlev <= 7 && cfg->num_blocks >= 0 && (cfg->out_ch == 1 || cfg->out_ch == 3) && cfg->nbits > 0 && cfg->nbits <= 1024);
  std::map<std::string, HostT> sd;
  for (int i = 0; i < ntensors; ++i) {
    VS_REQUIRE(tensors[i].name && tensors[i].data && tensors[i].numel > 0);
    sd[tensors[i].name] = HostT{tensors[i].data, tensors[i].numel};
  }
  vs_model* m = new vs_model();
  m->c = *cfg;
  m->arith = cfg->arith == 2 || cfg->arith == 3 ? cfg->arith : vs_default_arith();
  for (int i = 0; i <= cfg->nlev; ++i) {
    m->zc.push_back(cfg->zc[i]);
    if (cfg->zc[i] % 4) { delete m; return VS_ERR_UNSUPPORTED; }
  }
  m->bott = m->zc.back() + cfg->hidden;
  Packer P{m, sd};
  const std::string u = "embedder.unet";
  P.resblock(m->inc, u + ".inc", cfg->in_ch);
  const int nlev = cfg->nlev;
  m->down_conv.resize(nlev); m->down_rb.resize(nlev); m->ups.resize(nlev); m->bottleneck.resize(cfg->num_blocks);
  for (int i = 0; i < nlev; ++i) {
    const std::string p = u + ".downs." + std::to_string(i);
    P.conv(m->down_conv[i], p + ".down.weight", m->zc[i], 3, 3, rup(m->zc[i], 4), nullptr, nullptr, p + ".down.bias");
    P.resblock(m->down_rb[i], p + ".conv", m->zc[i + 1]);
  }
  for (int j = 0; j < cfg->num_blocks; ++j) {
    std::vector<float> w0, b0;
    const int nlat = m->zc.back(), hidden = cfg->hidden;
    const bool table = j == 0 && nlat % 16 == 0 && hidden % 32 == 0 && m->bott % 16 == 0;
    P.resblock(m->bottleneck[j], u + ".bottleneck.model." + std::to_string(j), m->bott, table ? &w0 : nullptr, table ? &b0 : nullptr);
    if (table && !w0.empty() && m->bottleneck[0].cout >= 128) {
      P.msg_table_convs(m->b0_lat, m->b0_msg, w0, b0, m->bottleneck[0].cout, m->bottleneck[0].c0.CinP, nlat, hidden);
      m->b0_table = true;
    }
  }
  std::vector<int> zz(m->zc.begin(), m->zc.end() - 1);
  zz.push_back(m->bott);
  for (int k = 0; k < nlev; ++k) {
    const int i = nlev - 1 - k;
    const int cin = 2 * zz[i + 1], cout = zz[i];
    const std::string p = u + ".ups." + std::to_string(k);
    m->ups[k].lowres = vs_upconv_supported(cout) != 0;
    if (m->ups[k].lowres) P.upconv9(m->ups[k].gemm, p + ".up.upsample_block.2.weight", cin, cout);
    else P.conv(m->ups[k].conv, p + ".up.upsample_block.2.weight", cin, 3, 3, rup(cin, 4), nullptr, nullptr, "");
    m->ups[k].lnw = P.vec(p + ".up.upsample_block.3.weight");
    m->ups[k].lnb = P.vec(p + ".up.upsample_block.3.bias");
    P.resblock(m->ups[k].rb, p + ".conv", cout);
  }
  m->outc_w = P.vec(u + ".outc.weight");
  m->outc_b = P.vec(u + ".outc.bias");
  m->table = P.vec(u + ".msg_processor.msg_embeddings.weight");
  // extractor
  const std::string cn = "detector.convnext";
  P.patch_conv(m->stem, cn + ".downsample_layers.0.0.weight", 3, 4, 4, cn + ".downsample_layers.0.0.bias");
  m->stem_lnw = P.vec(cn + ".downsample_layers.0.1.weight");
  m->stem_lnb = P.vec(cn + ".downsample_layers.0.1.bias");
  for (int i = 0; i < 3; ++i) {
    const std::string p = cn + ".downsample_layers." + std::to_string(i + 1);
    m->down[i].lnw = P.vec(p + ".0.weight");
    m->down[i].lnb = P.vec(p + ".0.bias");
    P.patch_conv(m->down[i].conv, p + ".1.weight", cfg->dims[i], 2, xld(cfg->dims[i]), p + ".1.bias", &m->down[i].gemm);
  }
  for (int s = 0; s < 4; ++s) {
    const int Cc = cfg->dims[s], ld = xld(Cc), ld4 = xld(4 * Cc);
    m->stages[s].resize(cfg->depths[s]);
    for (int j = 0; j < cfg->depths[s]; ++j) {
      const std::string p = cn + ".stages." + std::to_string(s) + "." + std::to_string(j);
      Blk& b = m->stages[s][j];
      const HostT& dw = P.get(p + ".dwconv.weight");
      if (dw.p) {
        std::vector<float> wdw((size_t)49 * ld, 0.f);
        for (int c2 = 0; c2 < Cc; ++c2)
          for (int t = 0; t < 49; ++t) wdw[(size_t)t * ld + c2] = dw.p[(size_t)c2 * 49 + t];
        b.wdw = P.upload(wdw);
      }
      b.bdw = P.vec(p + ".dwconv.bias", ld);
      b.lnw = P.vec(p + ".norm.weight", ld);
      b.lnb = P.vec(p + ".norm.bias", ld);
      b.gamma = P.vec(p + ".grn.gamma");
      b.beta = P.vec(p + ".grn.beta", rup(ld4, 16));
      P.conv(b.pw1, p + ".pwconv1.weight", Cc, 1, 1, ld, nullptr, nullptr, p + ".pwconv1.bias");
      P.conv(b.pw2, p + ".pwconv2.weight", 4 * Cc, 1, 1, ld4, nullptr, nullptr, p + ".pwconv2.bias");
      if (s < 2 && (Cc == 96 || Cc == 192) && m->arith == 2)      // engine.py::_pack_extractor: the fused block kernel of stages 0 / 1
        b.fuse = P.cnx_image(P.get(p + ".pwconv1.weight"), P.get(p + ".pwconv1.bias"), P.get(p + ".pwconv2.weight"), P.get(p + ".grn.beta"), Cc,
                             b.fm1, b.fm2);
    }
  }
  const std::string pd = "detector.pixel_decoder";
  P.conv(m->head, pd + ".output_upscaling.0.upsample_block.2.weight", cfg->dims[3], 3, 3, xld(cfg->dims[3]), nullptr, nullptr, "", &m->head_gemm);
  m->head_lnw = P.vec(pd + ".output_upscaling.0.upsample_block.3.weight");
  m->head_lnb = P.vec(pd + ".output_upscaling.0.upsample_block.3.bias");
  m->lin_w = P.vec(pd + ".linear.weight");
  m->lin_b = P.vec(pd + ".linear.bias");
  const HostT& M = P.get("rgb2yuv.M", !cfg->yuv);
  if (M.p) for (int i = 0; i < 3; ++i) m->ymat[i] = M.p[i];
  const bool noatt = !cfg->attenuate;
  const HostT &tl = P.get("attenuation.conv_lum.weight", noatt), &tx = P.get("attenuation.conv_x.weight", noatt), &ty = P.get("attenuation.conv_y.weight", noatt);
  if (tl.p && tx.p && ty.p) {
    for (int i = 0; i < 25; ++i) m->taps43[i] = tl.p[i];
    for (int i = 0; i < 9; ++i) { m->taps43[25 + i] = tx.p[i]; m->taps43[34 + i] = ty.p[i]; }
  }
  if (P.fail) {
    for (void* d : m->dev) (void)hipFree(d);
    delete m;
    return VS_ERR_BAD_ARG;       // a tensor of the state_dict is missing (or the device ran out of memory)
  }
  *out = m;
  return VS_OK;
}

extern "C" void vs_model_destroy(vs_model_t* m) {
  if (!m) return;
  for (void* d : m->dev) (void)hipFree(d);
  delete m;
}

extern "C" int64_t vs_model_workspace_bytes(const vs_model_t* m, int frames, int H, int W, int step) {
  if (!m || frames <= 0 || H <= 0 || W <= 0 || step < 1) return -1;
  Runner e{const_cast<vs_model_t*>(m), nullptr, 0, 0, nullptr};
  e.embed(nullptr, false, nullptr, 1, frames, H, W, step, 0, 1, 1, nullptr, nullptr);
  Runner dt{const_cast<vs_model_t*>(m), nullptr, 0, 0, nullptr};
  dt.detect(nullptr, false, frames, H, W, 1, nullptr);
  return std::max(e.used, dt.used) + 256;
}

extern "C" int vs_model_embed(vs_model_t* m, const void* imgs, const int32_t* msgs, int n_msgs, int frames, int H, int W, int step,
                              int video_mode, int lowres_attenuation, int antialias, int io_u8, void* imgs_w, float* preds_w, void* ws,
                              int64_t ws_bytes, void* stream) {
  VS_REQUIRE(m && imgs && msgs && imgs_w && ws && frames > 0 && H > 0 && W > 0 && step >= 1 && ((uintptr_t)ws & 255) == 0);
  const int nk = (frames + step - 1) / step;
  VS_REQUIRE(n_msgs == 1 || n_msgs == nk);
  VS_REQUIRE(video_mode >= 0 && video_mode <= 2);
  Runner r{m, static_cast<char*>(ws), ws_bytes, 0, stream};
  if (io_u8) VS_REQUIRE(m->c.clamp);
  r.embed(imgs, io_u8 != 0, msgs, n_msgs, frames, H, W, step, video_mode, lowres_attenuation, antialias, imgs_w, preds_w);
  return r.rc;
}

extern "C" int vs_model_detect(vs_model_t* m, const void* imgs, int frames, int H, int W, int antialias, int io_u8, float* logits, void* ws,
                               int64_t ws_bytes, void* stream) {
  VS_REQUIRE(m && imgs && logits && ws && frames > 0 && H > 0 && W > 0 && ((uintptr_t)ws & 255) == 0);
  Runner r{m, static_cast<char*>(ws), ws_bytes, 0, stream};
  r.detect(imgs, io_u8 != 0, frames, H, W, antialias, logits);
  return r.rc;
}
