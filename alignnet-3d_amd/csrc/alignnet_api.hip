// C ABI of libalignnet_hip.so (include/alignnet_hip.h): handle lifetime, parameter table,
// eval-mode forward orchestration.  gfx950 only; no CPU fallback by design -- every entry
// point fails with an error when HIP is unavailable.
#include "engine.h"
#include "kernels_infer.h"
#include "kernels_dgcnn.h"
#include "kernels_infer_split.h"
#include "kernels_dgcnn_split.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

using namespace alignnet;

static thread_local std::string g_create_err;

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static int fail(const alignnet_handle* h, const std::string& m)
{
  h->err = m;
  return 1;
}

// ---------------------------------------------------------------------------------
// graph description -> layer + parameter tables (models/tp8.py:101-158; names SURVEY 8.A2)
// ---------------------------------------------------------------------------------
static const char* kTower[2] = {"siamese", "siamese_1"};

static int add_param(alignnet_handle* h, const std::string& name, int rows, int cols, bool trainable)
{
  ParamInfo p{name, rows, cols, trainable, 0};
  h->params.push_back(p);
  h->by_name[name] = (int)h->params.size() - 1;
  return (int)h->params.size() - 1;
}

static void add_layer(alignnet_handle* h, const std::string& name, int cin, int cout, bool bn, bool siamese, bool conv,
                      bool first_conv, int fan_in, int fan_out)
{
  Layer L;
  L.name = name; L.cin = cin; L.cout = cout; L.bn = bn; L.siamese = siamese; L.conv = conv; L.first_conv = first_conv;
  L.fan_in = fan_in; L.fan_out = fan_out;
  const std::string base = siamese ? "siamese/" + name : name;
  L.p_w = add_param(h, base + "/weights", cin, cout, true);
  L.p_b = add_param(h, base + "/biases", 1, cout, true);
  for (int s = 0; s < 2; ++s) for (int k = 0; k < 4; ++k) L.p_bn[s][k] = -1;
  if (bn) {
    static const char* leaf[4] = {"beta", "gamma", "moving_mean", "moving_var"};
    const int nsets = siamese ? 2 : 1;
    for (int s = 0; s < nsets; ++s) {
      const std::string b = siamese ? std::string(kTower[s]) + "/" + name : name;
      for (int k = 0; k < 4; ++k) L.p_bn[s][k] = add_param(h, b + "/bn/" + leaf[k], 1, cout, k < 2);
    }
  }
  h->layers.push_back(L);
}

static Stack conv_stack(alignnet_handle* h, const std::string& prefix, const alignnet_widths& w)
{
  Stack st{(int)h->layers.size(), w.n};
  const bool dg = h->cfg.backbone == 1;
  int cin = dg ? 2 * h->cfg.num_channels : h->cfg.num_channels;
  for (int i = 0; i < w.n; ++i) {
    const bool pn_first = (i == 0) && !dg;
    // utils/tf_util.py:148-152: fan = kh*kw*channels; first PointNet conv has kernel [1, num_channels] on 1 channel;
    // DGCNN's first conv is a 1x1 conv on the 2*num_channels edge feature (tf_util_dgcnn.py:705)
    const int fi = pn_first ? h->cfg.num_channels : cin, fo = pn_first ? h->cfg.num_channels * w.w[i] : w.w[i];
    add_layer(h, prefix + "/conv" + std::to_string(i + 1), cin, w.w[i], true, true, true, i == 0, fi, fo);
    cin = w.w[i];
  }
  return st;
}

static Stack fc_stack(alignnet_handle* h, const std::string& prefix, int cin, const alignnet_widths& w, int out, bool siamese)
{
  Stack st{(int)h->layers.size(), w.n + 1};
  for (int j = 0; j <= w.n; ++j) {
    const int c = j < w.n ? w.w[j] : out;
    const std::string nm = (prefix.empty() ? "" : prefix + "/") + "fc" + std::to_string(j + 1);
    add_layer(h, nm, cin, c, j < w.n, siamese, false, false, cin, c);
    cin = c;
  }
  return st;
}

static bool check_widths(const alignnet_widths& w, int lo)
{
  if (w.n < lo || w.n > ALIGNNET_MAX_WIDTHS) return false;
  for (int i = 0; i < w.n; ++i) if (w.w[i] <= 0) return false;
  return true;
}

extern "C" int alignnet_abi_version(void) { return ALIGNNET_ABI_VERSION; }

extern "C" const char* alignnet_last_error(const alignnet_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int alignnet_create(const alignnet_config* cfg, alignnet_handle** out)
{
  if (!cfg || !out) { g_create_err = "alignnet_create: null argument"; return 1; }
  *out = nullptr;
  if (cfg->abi_version != ALIGNNET_ABI_VERSION) { g_create_err = "alignnet_create: abi_version mismatch"; return 1; }
  if (cfg->num_channels != 3) { g_create_err = "alignnet_create: num_channels must be 3 (provider.py:123 feeds xyz)"; return 1; }
  if (cfg->num_points < 1 || cfg->num_bins < 1) { g_create_err = "alignnet_create: bad num_points/num_bins"; return 1; }
  if (cfg->backbone != 0 && cfg->backbone != 1) { g_create_err = "alignnet_create: backbone must be 0 (pointnet) or 1 (dgcnn)"; return 1; }
  if (!check_widths(cfg->s1_conv, 2) || !check_widths(cfg->s2_conv, 2) || !check_widths(cfg->emb_conv, 2) ||
      !check_widths(cfg->s1_fc, 1) || !check_widths(cfg->s2_fc, 1) || !check_widths(cfg->rem_fc, 1)) {
    g_create_err = "alignnet_create: width lists need >=2 conv / >=1 fc entries, at most ALIGNNET_MAX_WIDTHS, all > 0";
    return 1;
  }
  if (cfg->s1_conv.n > kMaxConv || cfg->s2_conv.n > kMaxConv || cfg->emb_conv.n > kMaxConv) {
    g_create_err = "alignnet_create: at most 6 conv layers per backbone"; return 1;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("alignnet_create: no HIP device available (") + hipGetErrorString(e) +
                   "); this library has no CPU fallback";
    return 1;
  }
  if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "alignnet_create: device ordinal out of range"; return 1; }
  if ((e = hipSetDevice(cfg->device)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return 1; }

  alignnet_handle* h = new alignnet_handle();
#ifdef ALIGNNET_ABLATE
  if (const char* e = getenv("ALIGNNET_DBG")) h->ablate_dbg = atoi(e);   // (the ablation build is the only one that reads the environment)
#endif
  h->cfg = *cfg;
  const int nb2 = 2 * cfg->num_bins;
  h->s1_conv = conv_stack(h, "transformer1/embedding", cfg->s1_conv);
  h->s1_fc = fc_stack(h, "transformer1/mlp", cfg->s1_conv.w[cfg->s1_conv.n - 1], cfg->s1_fc, 3, true);
  h->s2_conv = conv_stack(h, "transformer2/embedding", cfg->s2_conv);
  h->s2_fc = fc_stack(h, "transformer2/mlp", cfg->s2_conv.w[cfg->s2_conv.n - 1], cfg->s2_fc, 3 + nb2, true);
  h->emb_conv = conv_stack(h, "embedding", cfg->emb_conv);   // scope 'final_embedding' ignored: tp8.py:62-66
  h->rem_fc = fc_stack(h, "", 2 * cfg->emb_conv.w[cfg->emb_conv.n - 1], cfg->rem_fc, 3 + nb2, false);

  // hidden FC widths must be multiples of 8 (k-group of the MFMA weight image)
  for (const Layer& L : h->layers)
    if (!L.first_conv && (L.cin % 8) != 0) {
      g_create_err = "alignnet_create: layer " + L.name + " has input width " + std::to_string(L.cin) + ", not a multiple of 8";
      delete h; return 1;
    }

  // flat parameter buffer: trainable first (one contiguous block = the all-reduce / Adam vector), then EMA
  size_t off = 0;
  for (auto& p : h->params) if (p.trainable) { p.offset = off; off += p.count(); }
  h->n_trainable = off;
  for (auto& p : h->params) if (!p.trainable) { p.offset = off; off += p.count(); }
  h->n_total = off;

  size_t wp = 0, ss = 0;
  for (Layer& L : h->layers) {
    L.off_ss = ss; ss += 2 * (size_t)L.cout;
    L.off_wp = wp;
    if (!L.first_conv) wp += (size_t)((L.cout + 31) / 32) * ((L.cin + 7) / 8) * 256;
  }
  h->n_wp = wp; h->n_ss = ss;

  auto bail = [&](const char* what, hipError_t er) {
    g_create_err = std::string(what) + ": " + hipGetErrorString(er);
    alignnet_destroy(h);
    return 1;
  };
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipMalloc(&h->d_params, h->n_total * sizeof(float))) != hipSuccess) return bail("hipMalloc params", e);
  if ((e = hipMalloc(&h->d_wp, std::max<size_t>(wp, 1) * sizeof(float))) != hipSuccess) return bail("hipMalloc wp", e);
  if ((e = hipMalloc(&h->d_scale, ss * sizeof(float))) != hipSuccess) return bail("hipMalloc scale", e);
  if ((e = hipMalloc(&h->d_shift, ss * sizeof(float))) != hipSuccess) return bail("hipMalloc shift", e);
  if ((e = hipMemset(h->d_params, 0, h->n_total * sizeof(float))) != hipSuccess) return bail("hipMemset", e);
  hipEventCreate(&h->ev[0]);
  hipEventCreate(&h->ev[1]);
  *out = h;
  if (alignnet_init_params(h, cfg->seed) != 0) { g_create_err = h->err; alignnet_destroy(h); *out = nullptr; return 1; }
  return 0;
}

static void free_ws(alignnet_handle* h)
{
  for (int t = 0; t < 2; ++t) if (h->ws.d_pcs[t]) { hipFree(h->ws.d_pcs[t]); h->ws.d_pcs[t] = nullptr; }
  if (h->ws.d_all) { hipFree(h->ws.d_all); h->ws.d_all = nullptr; }
  if (h->ws.d_nn) { hipFree(h->ws.d_nn); h->ws.d_nn = nullptr; }
  h->ws.cap = 0;
}

extern "C" void alignnet_train_ws_free(alignnet_handle* h);
extern "C" int alignnet_dataset_free(alignnet_handle* h);
extern "C" void alignnet_comm_free(alignnet_handle* h);

namespace { void pipe_free(alignnet_handle* h); hipStream_t pipe_stream(alignnet_handle* h, int which); }   // pipelined host path, defined with alignnet_forward_submit below

extern "C" void alignnet_destroy(alignnet_handle* h)
{
  if (!h) return;
  hipSetDevice(h->cfg.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  alignnet_comm_free(h);
  alignnet_train_ws_free(h);
  alignnet_dataset_free(h);
  pipe_free(h);
  free_ws(h);
  for (auto& pr : h->prof_pending) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); }
  for (auto& pr : h->prof_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  if (h->ev[0]) hipEventDestroy(h->ev[0]);
  if (h->ev[1]) hipEventDestroy(h->ev[1]);
  if (h->d_params) hipFree(h->d_params);
  if (h->d_wp) hipFree(h->d_wp);
  if (h->d_wps) hipFree(h->d_wps);
  if (h->d_scale) hipFree(h->d_scale);
  if (h->d_shift) hipFree(h->d_shift);
  if (h->sync_buf) hipFree(h->sync_buf);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

// ---------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------
extern "C" int alignnet_init_params(alignnet_handle* h, uint64_t seed)
{
  if (!h) return 1;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::vector<float> host(h->n_total, 0.f);
  std::mt19937_64 rng(seed);
  for (const Layer& L : h->layers) {
    // tf.contrib.layers.xavier_initializer(): uniform(-limit, limit), limit = sqrt(6/(fan_in+fan_out))
    const float limit = std::sqrt(6.0f / (float)(L.fan_in + L.fan_out));
    std::uniform_real_distribution<float> U(-limit, limit);
    const ParamInfo& w = h->params[L.p_w];
    for (size_t i = 0; i < w.count(); ++i) host[w.offset + i] = U(rng);
    // biases 0 (tf_util.py:159), beta 0, gamma 1 (tf_util.py:470-473), EMA shadows 0
    for (int s = 0; s < 2; ++s)
      if (L.p_bn[s][1] >= 0) {
        const ParamInfo& g = h->params[L.p_bn[s][1]];
        for (size_t i = 0; i < g.count(); ++i) host[g.offset + i] = 1.f;
      }
  }
  HIP_TRY(h, hipMemcpy(h->d_params, host.data(), h->n_total * sizeof(float), hipMemcpyHostToDevice));
  h->folded = false;
  return 0;
}

extern "C" int alignnet_num_params(const alignnet_handle* h) { return h ? (int)h->params.size() : -1; }

extern "C" int alignnet_param_info(const alignnet_handle* h, int index, const char** name, int32_t* rows, int32_t* cols,
                                   int32_t* trainable)
{
  if (!h) return 1;
  if (index < 0 || index >= (int)h->params.size()) return fail(h, "alignnet_param_info: index out of range");
  const ParamInfo& p = h->params[index];
  if (name) *name = p.name.c_str();
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  if (trainable) *trainable = p.trainable;
  return 0;
}

static const ParamInfo* find_param(alignnet_handle* h, const char* name, size_t count, const char* who)
{
  if (!name) { h->err = std::string(who) + ": null name"; return nullptr; }
  auto it = h->by_name.find(name);
  if (it == h->by_name.end()) { h->err = std::string(who) + ": unknown variable '" + name + "'"; return nullptr; }
  const ParamInfo& p = h->params[it->second];
  if (p.count() != count) {
    h->err = std::string(who) + ": '" + name + "' has " + std::to_string(p.count()) + " elements, caller passed " + std::to_string(count);
    return nullptr;
  }
  return &p;
}

extern "C" int alignnet_get_param(alignnet_handle* h, const char* name, float* dst, size_t count)
{
  if (!h) return 1;
  if (!dst) return fail(h, "alignnet_get_param: null dst");
  const ParamInfo* p = find_param(h, name, count, "alignnet_get_param");
  if (!p) return 1;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(dst, h->d_params + p->offset, count * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int alignnet_set_param(alignnet_handle* h, const char* name, const float* src, size_t count)
{
  if (!h) return 1;
  if (!src) return fail(h, "alignnet_set_param: null src");
  const ParamInfo* p = find_param(h, name, count, "alignnet_set_param");
  if (!p) return 1;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(h->d_params + p->offset, src, count * sizeof(float), hipMemcpyHostToDevice));
  h->folded = false;
  return 0;
}

// eval-mode preparation: BN fold + MFMA weight images, re-done whenever a variable changed
static int fold_for_eval(alignnet_handle* h)
{
  if (h->folded) return 0;
  for (const Layer& L : h->layers) {
    const float* bias = h->d_params + h->params[L.p_b].offset;
    for (int s = 0; s < 2; ++s) {
      const int src = (L.bn && L.p_bn[s][0] < 0) ? 0 : s;   // single-set layers replicate set 0
      const float *beta = nullptr, *gamma = nullptr, *mean = nullptr, *var = nullptr;
      if (L.bn) {
        beta = h->d_params + h->params[L.p_bn[src][0]].offset;
        gamma = h->d_params + h->params[L.p_bn[src][1]].offset;
        mean = h->d_params + h->params[L.p_bn[src][2]].offset;
        var = h->d_params + h->params[L.p_bn[src][3]].offset;
      }
      hipLaunchKernelGGL(fold_bn_kernel, dim3((L.cout + 255) / 256), dim3(256), 0, h->stream, bias, beta, gamma, mean, var,
                         L.cout, h->d_scale + L.off_ss + (size_t)s * L.cout, h->d_shift + L.off_ss + (size_t)s * L.cout);
    }
    if (!L.first_conv) {
      const size_t total = (size_t)((L.cout + 31) / 32) * ((L.cin + 7) / 8) * 256;
      hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                         h->stream, h->d_params + h->params[L.p_w].offset, L.cin, L.cout, h->d_wp + L.off_wp);
    }
  }
  if (h->infer_split) {
    if (!h->d_wps) {
      size_t n = 0;
      h->off_wps.assign(h->layers.size(), 0);
      for (size_t i = 0; i < h->layers.size(); ++i) {
        const Layer& L = h->layers[i];
        if (!L.conv || L.first_conv) continue;
        h->off_wps[i] = n;
        n += (size_t)((L.cout + 31) / 32) * ((L.cin + 15) / 16) * 1024;
      }
      HIP_TRY(h, hipMalloc(&h->d_wps, std::max<size_t>(n, 1) * sizeof(unsigned short)));
    }
    for (size_t i = 0; i < h->layers.size(); ++i) {
      const Layer& L = h->layers[i];
      if (!L.conv || L.first_conv) continue;
      const size_t total = (size_t)((L.cout + 31) / 32) * ((L.cin + 15) / 16) * 512;
      hipLaunchKernelGGL(pack_weights_split_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, h->stream,
                         h->d_params + h->params[L.p_w].offset, L.cin, L.cout, h->d_wps + h->off_wps[i]);
    }
  }
  HIP_TRY(h, hipGetLastError());
  h->folded = true;
  return 0;
}

// ---------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------
static int max_fc_hidden(const alignnet_handle* h)
{
  int m = 8;
  for (const Layer& L : h->layers) if (!L.conv) m = std::max(m, L.cout);
  return m;
}

static int ensure_ws(alignnet_handle* h, int B, bool need_inputs)
{
  Workspace& w = h->ws;
  const int N = h->cfg.num_points;
  if (B > w.cap) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_ws(h);
    const int nb2 = 2 * h->cfg.num_bins;
    const int C1 = h->layers[h->s1_conv.first + h->s1_conv.n - 1].cout;
    const int C2 = h->layers[h->s2_conv.first + h->s2_conv.n - 1].cout;
    const int CE = h->layers[h->emb_conv.first + h->emb_conv.n - 1].cout;
    const int H = max_fc_hidden(h);
    const size_t B2 = 2 * (size_t)B;
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    size_t tot = 0;
    const size_t o_xform = tot; tot += al(B2 * 12);
    const size_t o_cm = tot; tot += al(B2 * 3);
    const size_t o_s1c = tot; tot += al(B2 * 3);
    const size_t o_s2c = tot; tot += al(B2 * 3);
    const size_t o_theta = tot; tot += al(B2);
    const size_t o_cls = tot; tot += al(B2);
    const size_t o_p1 = tot; tot += al(B2 * C1);
    const size_t o_p2 = tot; tot += al(B2 * C2);
    const size_t o_emb = tot; tot += al(B2 * CE);
    const size_t o_hs = tot; tot += al(B2 * H);    // (directly behind the pooled buffers: cleared with them by the centroid kernel)
    const size_t o_ha = tot; tot += al(B2 * H);
    const size_t o_hb = tot; tot += al(B2 * H);
    const size_t o_o1 = tot; tot += al(B2 * 3);
    const size_t o_o2 = tot; tot += al(B2 * (3 + nb2));
    const size_t o_o3 = tot; tot += al((size_t)B * (3 + nb2));
    size_t o_out[8];
    const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
    for (int i = 0; i < 8; ++i) { o_out[i] = tot; tot += al((size_t)B * widths[i]); }
    // the split-K head layer adds onto hid_s, which only centroid_kernel clears -- as part of the range [pool1, hid_a) it zeroes per forward
    if (!(o_hs >= o_p1 && o_hs + al(B2 * H) <= o_ha)) return fail(h, "workspace carve: hid_s must lie inside the range the centroid kernel clears ([pool1, hid_a))");
    HIP_TRY(h, hipMalloc(&w.d_all, tot * sizeof(float)));
    float* base = w.d_all;
    w.xform = base + o_xform; w.center_mean = base + o_cm; w.s1c = base + o_s1c; w.s2c = base + o_s2c;
    w.theta = base + o_theta; w.cls = reinterpret_cast<int*>(base + o_cls);
    w.pool1 = base + o_p1; w.pool2 = base + o_p2; w.emb = base + o_emb;
    w.hid_a = base + o_ha; w.hid_b = base + o_hb; w.hid_s = base + o_hs;
    w.o1 = base + o_o1; w.o2 = base + o_o2; w.o3 = base + o_o3;
    for (int i = 0; i < 8; ++i) w.outs[i] = base + o_out[i];
    w.cap = B;
  }
  if (h->cfg.backbone == 1 && !w.d_nn) {
    HIP_TRY(h, hipMalloc(&w.d_nn, (size_t)2 * w.cap * N * 20 * sizeof(int)));
  }
  if (need_inputs && !w.d_pcs[0]) {
    for (int t = 0; t < 2; ++t) HIP_TRY(h, hipMalloc(&w.d_pcs[t], (size_t)w.cap * N * 3 * sizeof(float)));
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
// Points per workgroup tile of the fused PointNet backbone.  A tile's time is set by its last layer (43 us at C3 = 1024 on 128 points) whatever the batch, so a
// batch whose 128-point tiles cover at most half of the 256 CUs is served sooner on twice as many 64-point tiles: 0.189 -> 0.134 ms per step at B = 1,
// 0.198 -> 0.142 at B = 8 (N = 1024); from one 128-point tile per CU on (B = 16: 0.207 against 0.221 ms; B = 32: 0.338 against 0.356) the 128-point tiling
// is the faster one again (a weight fragment feeds four row tiles instead of two).
static int infer_tile_pts(const alignnet_handle* h, int B)
{
  if (h->ab & AB_INFER_TILE64) return 64;
  if (h->infer_tile_opt) return h->infer_tile_opt;
  const long tiles128 = (long)2 * B * ((h->cfg.num_points + 127) / 128);
  return tiles128 <= 128 ? 64 : 128;
}

static size_t backbone_lds_bytes(const alignnet_handle* h, const Stack& st, int ld[2], int B)
{
  const int kTilePts = infer_tile_pts(h, B);
  int w[2] = {8, 8};
  for (int i = 0; i < st.n - 1; ++i) w[i & 1] = std::max(w[i & 1], (h->layers[st.first + i].cout + 7) & ~7);
  ld[0] = w[0] + 4; ld[1] = w[1] + 4;
  return ((size_t)kTilePts * 4 + (size_t)kTilePts * (ld[0] + ld[1])) * sizeof(float);
}

static int run_backbone(alignnet_handle* h, const Stack& st, const float* p1, const float* p2, int B, float* pooled,
                        long tower_stride, long row_stride, size_t pooled_floats)
{
  BackboneArgs a;
  a.pcs[0] = p1; a.pcs[1] = p2; a.xform = h->ws.xform; a.pooled = pooled;
  a.tower_stride = tower_stride; a.row_stride = row_stride;
  a.B = B; a.N = h->cfg.num_points; a.nlayers = st.n;
  const size_t lds = backbone_lds_bytes(h, st, a.ld, B);
  if (lds > 160 * 1024) return fail(h, "backbone hidden widths need more than 160 KiB of LDS per 128-point tile");
  for (int i = 0; i < st.n; ++i) {
    const Layer& L = h->layers[st.first + i];
    a.L[i].w = L.first_conv ? h->d_params + h->params[L.p_w].offset : h->d_wp + L.off_wp;
    a.L[i].scale = h->d_scale + L.off_ss;
    a.L[i].shift = h->d_shift + L.off_ss;
    a.L[i].cin = L.cin; a.L[i].cout = L.cout;
  }
  (void)pooled_floats;   // all pooled buffers are zeroed by one memset at the start of the step (forward_device)
#ifdef ALIGNNET_KSTAMP
  static long long* d_stamps = nullptr;
  if (!d_stamps) { hipMalloc(&d_stamps, 4 * 64 * sizeof(long long)); }
  hipMemsetAsync(d_stamps, 0, 4 * 64 * sizeof(long long), h->stream);
  a.stamps = d_stamps;
#endif
  static PerDeviceOnce attr_set;
  if (attr_set.need(h->cfg.device)) {
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_fused<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_fused<128, 68, 132>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_fused<128, 68, 132, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_fused<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_fused<64, 68, 132, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(h->cfg.device);
  }
  const int TP = infer_tile_pts(h, B);
  // tiles per workgroup: as many as still leave every CU several workgroups (the pooled max is published once per workgroup)
  const int ntiles = (a.N + TP - 1) / TP;
  int per = ntiles;
  while (per > 1 && (long)2 * B * ((ntiles + per - 1) / per) < 512) per = (per + 1) / 2;   // >= two workgroups per CU
  if (h->ab_tiles_per_wg > 0) per = std::min(ntiles, h->ab_tiles_per_wg);
  a.tiles_per_wg = per;
  const dim3 grid((ntiles + per - 1) / per, 2 * B);
  {
  ProfScope prof_scope(h, PK_BACKBONE, true);
  const int sc1 = h->layers[st.first].cout, sc2 = st.n == 3 ? h->layers[st.first + 1].cout : 0;
  if (h->infer_split && st.n == 3 && sc1 <= 16 * kSplitKB1 && sc2 <= 16 * kSplitKB2) {
    // split-bf16 backbone (opt-in): same tiling, three bf16 MFMAs per fp32 product
    SplitArgs sa;
    sa.pcs[0] = p1; sa.pcs[1] = p2; sa.xform = a.xform; sa.pooled = pooled; sa.tower_stride = tower_stride; sa.row_stride = row_stride;
    sa.B = B; sa.N = a.N; sa.C1 = sc1; sa.C2 = sc2; sa.C3 = h->layers[st.first + 2].cout;
    sa.w1 = a.L[0].w; sa.w2s = h->d_wps + h->off_wps[st.first + 1]; sa.w3s = h->d_wps + h->off_wps[st.first + 2];
    sa.dbg = h->ablate_dbg & 0xff;
    if (h->ablate_dbg >> 8) sa.prio_mask = (h->ablate_dbg >> 8) & 0xff;   // ablation build: ALIGNNET_DBG bits 8 .. 15
    sa.sc1 = a.L[0].scale; sa.sh1 = a.L[0].shift; sa.sc2 = a.L[1].scale; sa.sh2 = a.L[1].shift; sa.sc3 = a.L[2].scale; sa.sh3 = a.L[2].shift;
    const int ld1s = ((sc1 + 15) & ~15) + 8, ld2s = ((sc2 + 15) & ~15) + 8;
    const size_t slds = (size_t)kSplitTP * 4 * sizeof(float) + (size_t)2 * kSplitTP * (ld1s + ld2s) * sizeof(unsigned short);
    static PerDeviceOnce sattr;
    if (sattr.need(h->cfg.device)) { HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_split<>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); sattr.mark(h->cfg.device); }
    if (sc1 == 64 && sc2 == 128 && sa.C3 % 64 == 0 && sa.C3 >= 256 && split_persist_lds(sa.C3).total_bytes <= 160 * 1024 && !(h->ab & (AB_NO_LD_CONST | AB_SPLIT_TILEWISE))) {
      // persistent workgroups, one per CU (108 KB of activation tiles + the parameter tables: one resident workgroup per CU either way)
      static PerDeviceOnce pattr;
      static int cus[64];
      if (pattr.need(h->cfg.device)) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_split_persist<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_split_persist<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int n = 0;
        HIP_TRY(h, hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->cfg.device));
        cus[h->cfg.device & 63] = n > 0 ? n : 256;
        pattr.mark(h->cfg.device);
      }
      const long tiles = (long)2 * B * ((a.N + kSplitTP - 1) / kSplitTP);
      const int wgs = (int)std::min<long>(tiles, cus[h->cfg.device & 63]);
      {
#ifdef ALIGNNET_ABLATE
      if ((sa.dbg & 64) && sa.C3 == 1024) { sa.stamps = reinterpret_cast<long long*>(h->ws.hid_a); hipMemsetAsync(sa.stamps, 0, 8 * 16 * sizeof(long long), h->stream); }
#endif
      if (sa.C3 >= 512) { TIMED_LAUNCH(pointnet_split_persist<4>, dim3(wgs), dim3(kWaves * 64), (size_t)split_persist_lds(sa.C3).total_bytes, sa); }
      else { TIMED_LAUNCH(pointnet_split_persist<2>, dim3(wgs), dim3(kWaves * 64), (size_t)split_persist_lds(sa.C3).total_bytes, sa); }
      h->last_kernel = ALIGNNET_KERNEL_POINTNET_SPLIT_PERSIST;
#ifdef ALIGNNET_ABLATE
      if (sa.stamps) {
        long long hs[8 * 16];
        hipStreamSynchronize(h->stream);
        hipMemcpy(hs, sa.stamps, sizeof(hs), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
          std::fprintf(stderr, "psp wave %d: start +%lld |", w, hs[w * 16] - hs[0]);
          for (int i = 1; i < 7; ++i) std::fprintf(stderr, " %lld", hs[w * 16 + i] - hs[w * 16 + i - 1]);
          std::fprintf(stderr, "   (xs+barrier, lift, barrier, hidden, barrier, last layer)\n");
        }
      }
#endif
      }
    } else if (sc1 == 64 && sc2 == 128 && !(h->ab & AB_NO_LD_CONST)) {
      static PerDeviceOnce sattr;
      if (sattr.need(h->cfg.device)) { HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_split<64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); sattr.mark(h->cfg.device); }
      TIMED_LAUNCH((pointnet_split<64, 128>), dim3((a.N + kSplitTP - 1) / kSplitTP, 2 * B), dim3(kWaves * 64), slds, sa);
      h->last_kernel = ALIGNNET_KERNEL_POINTNET_SPLIT_64_128;
    } else {
      TIMED_LAUNCH(pointnet_split<>, dim3((a.N + kSplitTP - 1) / kSplitTP, 2 * B), dim3(kWaves * 64), slds, sa);
      h->last_kernel = ALIGNNET_KERNEL_POINTNET_SPLIT;
    }
  } else if (TP == 64 && !(h->ab & AB_INFER_TILE64) && a.ld[0] == 68 && a.ld[1] == 132 && st.n == 3 && h->layers[st.first + 2].cin == 128 && !(h->ab & AB_NO_LD_CONST)) {
    TIMED_LAUNCH((pointnet_fused<64, 68, 132, 16>), grid, dim3(kWaves * 64), lds, a);   // the shipped shape on 64-point tiles (small batches)
    h->last_kernel = ALIGNNET_KERNEL_POINTNET_FUSED_64_128_K16_TP64;
  } else if (TP == 64) { TIMED_LAUNCH(pointnet_fused<64>, grid, dim3(kWaves * 64), lds, a); h->last_kernel = ALIGNNET_KERNEL_POINTNET_FUSED_TP64; }
  else if (a.ld[0] == 68 && a.ld[1] == 132 && st.n == 3 && h->layers[st.first + 2].cin == 128 && !(h->ab & AB_NO_LD_CONST)) {
    TIMED_LAUNCH((pointnet_fused<128, 68, 132, 16>), grid, dim3(kWaves * 64), lds, a);
    h->last_kernel = ALIGNNET_KERNEL_POINTNET_FUSED_64_128_K16;
  } else if (a.ld[0] == 68 && a.ld[1] == 132 && !(h->ab & AB_NO_LD_CONST)) {   // the shipped widths 64, 128
    TIMED_LAUNCH((pointnet_fused<128, 68, 132>), grid, dim3(kWaves * 64), lds, a);
    h->last_kernel = ALIGNNET_KERNEL_POINTNET_FUSED_64_128;
  } else { TIMED_LAUNCH(pointnet_fused<128>, grid, dim3(kWaves * 64), lds, a); h->last_kernel = ALIGNNET_KERNEL_POINTNET_FUSED; }
  }
  HIP_TRY(h, hipGetLastError());
#ifdef ALIGNNET_KSTAMP
  if (st.first == h->emb_conv.first && true) {
    long long hs[4 * 64];
    hipStreamSynchronize(h->stream);
    hipMemcpy(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost);
    for (int q = 0; q < 4; ++q) {
      std::fprintf(stderr, "wave %d ct %d:", (q >> 1) * 4, (q & 1) * 8 + (q >> 1) * 4);
      for (int i = 1; i < 64 && hs[q * 64 + i]; ++i) std::fprintf(stderr, " %lld", hs[q * 64 + i] - hs[q * 64 + i - 1]);
      std::fprintf(stderr, "  | first %lld\n", hs[q * 64] - hs[0]);
    }
  }
#endif
  return 0;
}

static int run_backbone_dgcnn(alignnet_handle* h, const Stack& st, const float* p1, const float* p2, int B, float* pooled,
                              long tower_stride, long row_stride, size_t pooled_floats)
{
  DgcnnArgs a;
  a.stamps = nullptr;
  a.pcs[0] = p1; a.pcs[1] = p2; a.xform = h->ws.xform; a.nn = h->ws.d_nn; a.pooled = pooled;
  a.tower_stride = tower_stride; a.row_stride = row_stride;
  a.B = B; a.N = h->cfg.num_points; a.k = 20; a.nlayers = st.n;   // k = 20 is hard-coded in the reference (tp8.py:33)
  int wmax[2] = {8, 8};
  for (int i = 0; i < st.n - 1; ++i) wmax[i & 1] = std::max(wmax[i & 1], (h->layers[st.first + i].cout + 7) & ~7);
  a.ld[0] = wmax[0] + 4; a.ld[1] = wmax[1] + 4;
  // es [64][8] (+ [64][8] spare) | buf0 | buf1 | buf0' | lift table [C1 <= 64][8] (the shipped-shape instantiation)
  // (the shipped-shape instantiation: 52.5 KiB, three workgroups per CU -- layout in kernels_dgcnn.h)
  const bool dg_std = a.ld[0] == 68 && a.ld[1] == 132 && st.n == 3 && !(h->ab & AB_NO_LD_CONST);
  const size_t lds = dg_std ? ((size_t)kDgTile * 9 + (size_t)kDgTile * 68 + 4 * 8 * 64 * 4) * sizeof(float)   // es (row stride 9) | W2 image | one lift buffer
                            : ((size_t)kDgTile * 16 + (size_t)kDgTile * (2 * a.ld[0] + a.ld[1])) * sizeof(float);
  if (lds > 160 * 1024) return fail(h, "dgcnn hidden widths need more than 160 KiB of LDS per 64-point tile");
  if (h->layers[st.first + st.n - 2].cout > 256) return fail(h, "dgcnn: last edge-conv width limited to 256 channels");
  for (int i = 0; i < st.n; ++i) {
    const Layer& L = h->layers[st.first + i];
    a.L[i].w = L.first_conv ? h->d_params + h->params[L.p_w].offset : h->d_wp + L.off_wp;
    a.L[i].scale = h->d_scale + L.off_ss;
    a.L[i].shift = h->d_shift + L.off_ss;
    a.L[i].cin = L.cin; a.L[i].cout = L.cout;
  }
  (void)pooled_floats;
  static PerDeviceOnce attr_set;
  if (attr_set.need(h->cfg.device)) {
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dgcnn_fused<>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(h->cfg.device);
  }
  const dim3 grid((a.N + kDgTile - 1) / kDgTile, 2 * B);
  {
  ProfScope prof_scope(h, PK_BACKBONE, true);
  {
    const int dca = h->layers[st.first].cout, dcb = st.n == 3 ? h->layers[st.first + 1].cout : 0;
    if (h->infer_split && st.n == 3 && dca <= 16 * kSplitKB1 && dcb <= 128) {
      // split-bf16 variant (opt-in): three bf16 MFMAs per fp32 product
      DgcnnSplitArgs sa;
      sa.pcs[0] = p1; sa.pcs[1] = p2; sa.xform = a.xform; sa.nn = a.nn; sa.pooled = pooled; sa.tower_stride = tower_stride; sa.row_stride = row_stride;
      sa.B = B; sa.N = a.N; sa.k = a.k; sa.Ca = dca; sa.Cb = dcb; sa.C3 = h->layers[st.first + 2].cout;
      sa.w1 = a.L[0].w; sa.w2s = h->d_wps + h->off_wps[st.first + 1]; sa.w3s = h->d_wps + h->off_wps[st.first + 2];
      sa.sc1 = a.L[0].scale; sa.sh1 = a.L[0].shift; sa.sc2 = a.L[1].scale; sa.sh2 = a.L[1].shift; sa.sc3 = a.L[2].scale; sa.sh3 = a.L[2].shift;
      sa.dbg = h->ablate_dbg;
      const int dlda = ((dca + 15) & ~15) + 8, dldb = ((dcb + 15) & ~15) + 8;
      if (a.k > 3 * kWaves) return fail(h, "dgcnn split kernel: k limited to 24 neighbours");
      const size_t dlds = (size_t)2 * kDgTile * 8 * sizeof(float) + ((size_t)4 * kDgTile * dlda + (size_t)2 * kDgTile * dldb) * sizeof(unsigned short);
      static PerDeviceOnce dsattr;
      if (dsattr.need(h->cfg.device)) { HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dgcnn_split<>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dsattr.mark(h->cfg.device); }
      if (dca == 64 && dcb == 128 && !(h->ab & AB_NO_LD_CONST)) {
        static PerDeviceOnce dsattr2;
        if (dsattr2.need(h->cfg.device)) { HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dgcnn_split<64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dsattr2.mark(h->cfg.device); }
        TIMED_LAUNCH((dgcnn_split<64, 128>), grid, dim3(kWaves * 64), dlds, sa);
        h->last_kernel = ALIGNNET_KERNEL_DGCNN_SPLIT_64_128;
      } else {
        TIMED_LAUNCH(dgcnn_split<>, grid, dim3(kWaves * 64), dlds, sa);
        h->last_kernel = ALIGNNET_KERNEL_DGCNN_SPLIT;
      }
    } else {
      const int dbg = h->ablate_dbg;
      a.stamps = (dbg & 64) ? reinterpret_cast<long long*>(h->ws.hid_a) : nullptr;   // scratch that is idle during the backbone
      if (a.ld[0] == 68 && a.ld[1] == 132 && st.n == 3 && !(h->ab & AB_NO_LD_CONST)) {   // the shipped widths 64, 128
        static PerDeviceOnce sattr;
        if (sattr.need(h->cfg.device)) { HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dgcnn_fused<68, 132>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); sattr.mark(h->cfg.device); }
        TIMED_LAUNCH((dgcnn_fused<68, 132>), grid, dim3(kWaves * 64), lds, a);
        h->last_kernel = ALIGNNET_KERNEL_DGCNN_FUSED_64_128;
      } else {
        TIMED_LAUNCH(dgcnn_fused<>, grid, dim3(kWaves * 64), lds, a);
        h->last_kernel = ALIGNNET_KERNEL_DGCNN_FUSED;
      }
      if (a.stamps) {
        long long sv[9];
        hipStreamSynchronize(h->stream);
        hipMemcpy(sv, a.stamps, sizeof(sv), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "dgcnn_fused slot-5 cycles: mfma %lld es %lld barrier %lld lift %lld barrier %lld | k-max store + barrier %lld point conv %lld\n",
                     sv[1] - sv[0], sv[2] - sv[1], sv[3] - sv[2], sv[4] - sv[3], sv[5] - sv[4], sv[7] - sv[6], sv[8] - sv[7]);
      }
    }
  }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

static int run_fc(alignnet_handle* h, const Layer& L, const float* in, long ldin, float* out, long ldout, int M,
                  int rows_per_set, bool relu, const FcArgs* finish = nullptr, int ksplit = 1, const Layer* in_bn = nullptr)
{
  FcArgs a;
  if (finish) a = *finish;   // the stage glue folded into this (last) layer's epilogue
  a.in = in; a.ldin = ldin; a.wp = h->d_wp + L.off_wp; a.scale = h->d_scale + L.off_ss; a.shift = h->d_shift + L.off_ss;
  a.out = out; a.ldout = ldout; a.M = M; a.K = L.cin; a.Nout = L.cout; a.relu = relu; a.rows_per_set = rows_per_set;
  a.ksplit = ksplit;
  if (in_bn) { a.in_scale = h->d_scale + in_bn->off_ss; a.in_shift = h->d_shift + in_bn->off_ss; }   // (BN set 0: the pair head is not siamese)
  const dim3 grid((L.cout + 31) / 32, (M + 31) / 32, ksplit);
  if (h->ab & AB_FC_DIRECT) {
    if (L.cin >= 1024) hipLaunchKernelGGL(fc_mfma<8>, grid, dim3(512), 0, h->stream, a);
    else hipLaunchKernelGGL(fc_mfma<4>, grid, dim3(256), 0, h->stream, a);
  } else if (L.cin >= 1024) hipLaunchKernelGGL((fc_mfma<8, true>), grid, dim3(512), 0, h->stream, a);
  else hipLaunchKernelGGL((fc_mfma<4, true>), grid, dim3(256), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// head MLP (models/tp8.py:75-82) in eval mode: dropout is the identity (tf_util.py:571-574)
static int run_head(alignnet_handle* h, const Stack& st, const float* in, long ldin, float* out, long ldout, int M, int rows_per_set,
                    const FcArgs* finish = nullptr)
{
  const float* cur = in; long ldc = ldin;
  float* pp[2] = {h->ws.hid_a, h->ws.hid_b};
  const Layer* split_bn = nullptr;
  for (int j = 0; j < st.n; ++j) {
    const Layer& L = h->layers[st.first + j];
    const bool last = j == st.n - 1;
    float* dst = last ? out : pp[j & 1];
    const long ldd = last ? ldout : L.cout;
    // a deep first layer on few tiles (the pair head: K = 2048, 128 tiles) as two K halves onto the zeroed hid_s; the next layer finishes it
    // (hid_s is zero exactly once per forward -- armed by forward_device behind the centroid kernel, consumed here: a second qualifying
    //  layer in the same forward takes the one-piece path instead of summing onto the first one's partials)
    const bool split = j == 0 && !last && L.cin >= 2048 && ((L.cin / 8) % 2) == 0 && (L.cin % 8) == 0 && M <= rows_per_set && h->ws.hid_s_zeroed &&
                       (long)((L.cout + 31) / 32) * ((M + 31) / 32) <= 128 && !(h->ab & (AB_FC_DIRECT | AB_FC_NO_SPLITK));
    if (split) {
      h->ws.hid_s_zeroed = false;
      if (run_fc(h, L, cur, ldc, h->ws.hid_s, L.cout, M, rows_per_set, false, nullptr, 2)) return 1;
      cur = h->ws.hid_s; ldc = L.cout; split_bn = &L;
      continue;
    }
    if (run_fc(h, L, cur, ldc, dst, ldd, M, rows_per_set, !last, last ? finish : nullptr, 1, split_bn)) return 1;
    split_bn = nullptr;
    cur = dst; ldc = ldd;
  }
  return 0;
}

static int forward_device(alignnet_handle* h, const float* p1, const float* p2, int B, float* const outs[8])
{
  if (fold_for_eval(h)) return 1;
  const bool dg = h->cfg.backbone == 1;
  if (dg && h->cfg.num_points > 64 * kKnnMaxPerLane) return fail(h, "dgcnn: num_points limited to 4096 (register-resident kNN)");
  if (dg && h->cfg.num_points < 20) return fail(h, "dgcnn: num_points must be >= k = 20");
  auto backbone = [&](const Stack& st, float* pooled, long ts, long rs, size_t n) {
    return dg ? run_backbone_dgcnn(h, st, p1, p2, B, pooled, ts, rs, n) : run_backbone(h, st, p1, p2, B, pooled, ts, rs, n);
  };
  Workspace& w = h->ws;
  const int N = h->cfg.num_points, nb = h->cfg.num_bins, nb2 = 2 * nb;
  const int C1 = h->layers[h->s1_conv.first + h->s1_conv.n - 1].cout;
  const int C2 = h->layers[h->s2_conv.first + h->s2_conv.n - 1].cout;
  const int CE = h->layers[h->emb_conv.first + h->emb_conv.n - 1].cout;
  const int B2 = 2 * B;
  if (h->prof) hipEventRecord(h->ev[0], h->stream);
  // pool1 | pool2 | emb are carved back to back: the centroid kernel arms all three atomicMax targets (a slice per workgroup)
  hipLaunchKernelGGL(centroid_kernel, dim3(B2), dim3(256), 0, h->stream, p1, p2, B, N, w.xform, w.center_mean, w.pool1,
                     (size_t)((char*)w.hid_a - (char*)w.pool1) / sizeof(float));
  w.hid_s_zeroed = true;   // (pool1 | pool2 | emb | hid_s: all cleared by that launch)
  if (dg) {   // static kNN graph (tp8.py:35-36), once per cloud in the mean-centred frame
    ProfScope prof_scope(h, PK_KNN, true);
    prof_scope.used = true;
    HIP_TRY(h, launch_knn(h->cfg.device, h->stream, p1, p2, w.center_mean, B, N, 20, w.d_nn, prof_scope.a, prof_scope.b));
  }
  // stage 1 (tp8.py:108-109)
  if (backbone(h->s1_conv, w.pool1, (long)B * C1, C1, (size_t)B2 * C1)) return 1;
  {   // (+ the stage-1 glue in the last layer's epilogue: s1 = head + center_mean, next frame, pred_s1 centres)
    FcArgs fin; fin.finish = 1; fin.B = B; fin.addend = w.center_mean; fin.s1c = w.s1c; fin.xform = w.xform; fin.out_a = outs[2]; fin.out_b = outs[3];
    if (run_head(h, h->s1_fc, w.pool1, C1, w.o1, 3, B2, B, &fin)) return 1;
  }
  // stage 2 (tp8.py:113-125)
  if (backbone(h->s2_conv, w.pool2, (long)B * C2, C2, (size_t)B2 * C2)) return 1;
  if (run_head(h, h->s2_fc, w.pool2, C2, w.o2, 3 + nb2, B2, B)) return 1;
  hipLaunchKernelGGL(stage2_finish_kernel, dim3((B2 + 3) / 4), dim3(256), 0, h->stream, w.o2, 3 + nb2, w.s1c, B, nb, w.s2c,
                     w.xform, w.theta, w.cls, outs[4], outs[5], outs[6], outs[7]);
  // stage 3: embedding of the normalised clouds, concat (tp8.py:130,144,153) = row b holds [emb1 | emb2]
  if (backbone(h->emb_conv, w.emb, CE, 2L * CE, (size_t)B2 * CE)) return 1;
  {   // (+ the final glue in the last layer's epilogue: pred_translations = head[:, :3] + (s2c2 - s2c1), remaining-angle logits)
    FcArgs fin; fin.finish = 3; fin.B = B; fin.nb = nb; fin.addend = w.s2c; fin.out_a = outs[0]; fin.out_b = outs[1];
    if (run_head(h, h->rem_fc, w.emb, 2L * CE, w.o3, 3 + nb2, B, B, &fin)) return 1;
  }
  if (h->prof) hipEventRecord(h->ev[1], h->stream);
  HIP_TRY(h, hipGetLastError());
  h->last_B = B;
  return 0;
}

// (also called from alignnet_train.hip: a long profiled training loop must not grow prof_pending / create events without bound)
int alignnet_drain_profile(alignnet_handle* h)
{
  if (!h->prof) return 0;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (auto& pr : h->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      h->prof_ms[pr.id] += ms; h->prof_launches[pr.id]++;
      if (pr.id == PK_BACKBONE) { h->prof_backbone_ms += ms; h->prof_backbone_launches++; }
    }
    h->prof_pool.push_back({pr.a, pr.b});
  }
  h->prof_pending.clear();
  return 0;
}

extern "C" int alignnet_forward_device(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2, int32_t B,
                                       const alignnet_outputs* d_out)
{
  if (!h) return 1;
  if (!d_pcs1 || !d_pcs2 || !d_out) return fail(h, "alignnet_forward_device: null argument");
  if (B < 1) return fail(h, "alignnet_forward_device: B must be >= 1");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (ensure_ws(h, B, false)) return 1;
  float* outs[8] = {d_out->pred_translations, d_out->pred_remaining_angle_logits, d_out->pred_s1_pc1centers,
                    d_out->pred_s1_pc2centers, d_out->pred_s2_pc1centers, d_out->pred_s2_pc2centers,
                    d_out->pred_pc1angle_logits, d_out->pred_pc2angle_logits};
  if (h->prof_pending.size() > 4096 && alignnet_drain_profile(h)) return 1;
  return forward_device(h, d_pcs1, d_pcs2, B, outs);
}

// ---------------------------------------------------------------------------------
// Pipelined host path (include/alignnet_hip.h: alignnet_forward_submit / alignnet_forward_wait).  The blocking alignnet_forward
// serialises copy-in -> forward -> copy-out per batch from pageable buffers (the reference's own timing, train.py:447-449, has the
// feed copy inside the timed region too).  Here a batch goes through a pinned staging slot: its H2D copy runs on a copy stream while
// the compute stream is still busy with the previous batch, the forward waits for it on an event, and the D2H of the outputs goes
// through a third stream into pinned memory; two slots, so two batches are in flight.
// ---------------------------------------------------------------------------------
namespace {
struct PipeSlot {
  int cap = 0; int B = 0; bool busy = false;
  float* h_in[2] = {nullptr, nullptr};      // pinned staging of pcs1 / pcs2
  float* h_out = nullptr;                   // pinned staging of the eight outputs
  float* d_in[2] = {nullptr, nullptr};
  float* d_out = nullptr; float* d_outs[8];
  size_t out_off[8];
  alignnet_outputs user{};
  hipEvent_t ev_in = nullptr, ev_fwd = nullptr, ev_out = nullptr;
};
struct Pipe { PipeSlot slot[2]; hipStream_t s_in = nullptr, s_out = nullptr; uint64_t submitted = 0, completed = 0; };

void pipe_free(alignnet_handle* h)
{
  if (!h->pipe) return;
  Pipe* p = static_cast<Pipe*>(h->pipe);
  if (p->s_in) hipStreamSynchronize(p->s_in);
  if (p->s_out) hipStreamSynchronize(p->s_out);
  for (PipeSlot& s : p->slot) {
    for (int t = 0; t < 2; ++t) { if (s.h_in[t]) hipHostFree(s.h_in[t]); if (s.d_in[t]) hipFree(s.d_in[t]); }
    if (s.h_out) hipHostFree(s.h_out);
    if (s.d_out) hipFree(s.d_out);
    if (s.ev_in) hipEventDestroy(s.ev_in);
    if (s.ev_fwd) hipEventDestroy(s.ev_fwd);
    if (s.ev_out) hipEventDestroy(s.ev_out);
  }
  if (p->s_in) hipStreamDestroy(p->s_in);
  if (p->s_out) hipStreamDestroy(p->s_out);
  delete p;
  h->pipe = nullptr;
}
hipStream_t pipe_stream(alignnet_handle* h, int which) { Pipe* p = static_cast<Pipe*>(h->pipe); return which ? p->s_out : p->s_in; }
}  // namespace

extern "C" int alignnet_forward_wait(alignnet_handle* h)
{
  if (!h) return 1;
  Pipe* p = static_cast<Pipe*>(h->pipe);
  if (!p || p->completed == p->submitted) return fail(h, "alignnet_forward_wait: no batch in flight");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  PipeSlot& s = p->slot[p->completed & 1];
  HIP_TRY(h, hipEventSynchronize(s.ev_out));
  float* host[8] = {s.user.pred_translations, s.user.pred_remaining_angle_logits, s.user.pred_s1_pc1centers, s.user.pred_s1_pc2centers,
                    s.user.pred_s2_pc1centers, s.user.pred_s2_pc2centers, s.user.pred_pc1angle_logits, s.user.pred_pc2angle_logits};
  const int nb2 = 2 * h->cfg.num_bins;
  const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
  for (int i = 0; i < 8; ++i)
    if (host[i]) std::memcpy(host[i], s.h_out + s.out_off[i], (size_t)s.B * widths[i] * sizeof(float));
  s.busy = false;
  p->completed++;
  return 0;
}

extern "C" int alignnet_forward_submit(alignnet_handle* h, const float* pcs1, const float* pcs2, int32_t B, const alignnet_outputs* out)
{
  if (!h) return 1;
  if (!pcs1 || !pcs2 || !out) return fail(h, "alignnet_forward_submit: null argument");
  if (B < 1) return fail(h, "alignnet_forward_submit: B must be >= 1");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (!h->pipe) {
    Pipe* np = new Pipe();
    h->pipe = np;
    HIP_TRY(h, hipStreamCreateWithFlags(&np->s_in, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&np->s_out, hipStreamNonBlocking));
    for (PipeSlot& s : np->slot) {
      HIP_TRY(h, hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&s.ev_fwd, hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
  }
  Pipe* p = static_cast<Pipe*>(h->pipe);
  if (p->submitted - p->completed >= 2) return fail(h, "alignnet_forward_submit: two batches are in flight already -- call alignnet_forward_wait first");
  PipeSlot& s = p->slot[p->submitted & 1];
  const int N = h->cfg.num_points, nb2 = 2 * h->cfg.num_bins;
  const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
  if (B > s.cap) {
    for (int t = 0; t < 2; ++t) { if (s.h_in[t]) hipHostFree(s.h_in[t]); if (s.d_in[t]) hipFree(s.d_in[t]); s.h_in[t] = s.d_in[t] = nullptr; }
    if (s.h_out) hipHostFree(s.h_out);
    if (s.d_out) hipFree(s.d_out);
    s.h_out = s.d_out = nullptr; s.cap = 0;
    const size_t nin = (size_t)B * N * 3 * sizeof(float);
    size_t tot = 0;
    for (int i = 0; i < 8; ++i) { s.out_off[i] = tot; tot += ((size_t)B * widths[i] + 63) & ~(size_t)63; }
    for (int t = 0; t < 2; ++t) { HIP_TRY(h, hipHostMalloc(&s.h_in[t], nin, hipHostMallocDefault)); HIP_TRY(h, hipMalloc(&s.d_in[t], nin)); }
    HIP_TRY(h, hipHostMalloc(&s.h_out, tot * sizeof(float), hipHostMallocDefault));
    HIP_TRY(h, hipMalloc(&s.d_out, tot * sizeof(float)));
    for (int i = 0; i < 8; ++i) s.d_outs[i] = s.d_out + s.out_off[i];
    s.cap = B;
  }
  if (ensure_ws(h, B, false)) return 1;
  const size_t nin = (size_t)B * N * 3 * sizeof(float);
  std::memcpy(s.h_in[0], pcs1, nin);   // (host work: overlaps the previous batch's forward on the GPU)
  std::memcpy(s.h_in[1], pcs2, nin);
  // From the first enqueue on, a failure must not leave work of this slot in flight: `submitted` is not advanced, so the next submit
  // reuses the same pinned / device buffers.  Everything queued so far is drained before the error is returned.
  auto enqueue = [&]() -> int {
    HIP_TRY(h, hipMemcpyAsync(s.d_in[0], s.h_in[0], nin, hipMemcpyHostToDevice, p->s_in));
    HIP_TRY(h, hipMemcpyAsync(s.d_in[1], s.h_in[1], nin, hipMemcpyHostToDevice, p->s_in));
    HIP_TRY(h, hipEventRecord(s.ev_in, p->s_in));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, s.ev_in, 0));
    if (h->prof_pending.size() > 4096 && alignnet_drain_profile(h)) return 1;
    if (forward_device(h, s.d_in[0], s.d_in[1], B, s.d_outs)) return 1;
    HIP_TRY(h, hipEventRecord(s.ev_fwd, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(p->s_out, s.ev_fwd, 0));
    for (int i = 0; i < 8; ++i)
      HIP_TRY(h, hipMemcpyAsync(s.h_out + s.out_off[i], s.d_outs[i], (size_t)B * widths[i] * sizeof(float), hipMemcpyDeviceToHost, p->s_out));
    HIP_TRY(h, hipEventRecord(s.ev_out, p->s_out));
    return 0;
  };
  if (enqueue()) {
    const std::string keep = h->err;
    hipStreamSynchronize(p->s_in); hipStreamSynchronize(h->stream); hipStreamSynchronize(p->s_out);
    h->err = keep;
    return 1;
  }
  s.user = *out; s.B = B; s.busy = true;
  p->submitted++;
  return 0;
}

extern "C" int alignnet_forward(alignnet_handle* h, const float* pcs1, const float* pcs2, int32_t B, const alignnet_outputs* out)
{
  if (!h) return 1;
  if (!pcs1 || !pcs2 || !out) return fail(h, "alignnet_forward: null argument");
  if (B < 1) return fail(h, "alignnet_forward: B must be >= 1");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (ensure_ws(h, B, true)) return 1;
  Workspace& w = h->ws;
  const size_t nin = (size_t)B * h->cfg.num_points * 3 * sizeof(float);
  HIP_TRY(h, hipMemcpyAsync(w.d_pcs[0], pcs1, nin, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(w.d_pcs[1], pcs2, nin, hipMemcpyHostToDevice, h->stream));
  if (forward_device(h, w.d_pcs[0], w.d_pcs[1], B, w.outs)) return 1;
  float* host[8] = {out->pred_translations, out->pred_remaining_angle_logits, out->pred_s1_pc1centers, out->pred_s1_pc2centers,
                    out->pred_s2_pc1centers, out->pred_s2_pc2centers, out->pred_pc1angle_logits, out->pred_pc2angle_logits};
  const int nb2 = 2 * h->cfg.num_bins;
  const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
  for (int i = 0; i < 8; ++i)
    if (host[i]) HIP_TRY(h, hipMemcpyAsync(host[i], w.outs[i], (size_t)B * widths[i] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return alignnet_drain_profile(h);
}

// test hook: the kNN graph [2B][N][20] (tower 1's B clouds, then tower 2's) built by the last eval-mode forward of a dgcnn engine
extern "C" int alignnet_debug_knn_graph(alignnet_handle* h, int32_t* dst, size_t count)
{
  if (!h) return 1;
  if (!dst) return fail(h, "alignnet_debug_knn_graph: null argument");
  if (h->cfg.backbone != 1 || !h->ws.d_nn || h->last_B < 1) return fail(h, "alignnet_debug_knn_graph: no dgcnn forward has run on this handle");
  const size_t need = (size_t)2 * h->last_B * h->cfg.num_points * 20;
  if (count != need) return fail(h, "alignnet_debug_knn_graph: count must be 2 * B * num_points * 20 of the last forward");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(dst, h->ws.d_nn, need * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int alignnet_synchronize(alignnet_handle* h)
{
  if (!h) return 1;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->pipe) {   // the pipelined host path's copy streams (alignnet_forward_submit)
    HIP_TRY(h, hipStreamSynchronize(pipe_stream(h, 0)));
    HIP_TRY(h, hipStreamSynchronize(pipe_stream(h, 1)));
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// run-time options (not part of the reference's config surface)
// ---------------------------------------------------------------------------------
extern "C" int alignnet_set_option(alignnet_handle* h, const char* key, int64_t value)
{
  if (!h || !key) return 1;
  const std::string k(key);
  if (k == "train_matmul_bf16") { h->train_bf16 = value != 0; return 0; }
  if (k == "allreduce_overlap") { h->comm_overlap = value != 0; return 0; }
  if (k == "train_dw_side_stream") { if (value < 0 || value > 2) return fail(h, "train_dw_side_stream: 0, 1 or 2"); h->dw_side = (int)value; return 0; }
  if (k == "train_phase3_tile64") { h->p3_tile64 = value != 0; return 0; }
  if (k == "dropout_stream") { h->dropout_stream = (uint64_t)value; return 0; }
  if (k == "sync_bn") { h->sync_bn = value != 0; return 0; }
  if (k == "global_loss") { h->global_loss = value != 0; return 0; }
  if (k == "sync_bn_emulate_world") { if (value < 1) return fail(h, "sync_bn_emulate_world must be >= 1"); h->sync_emulate_world = (int)value; return 0; }
  if (k == "train_fused_tail") {
    if (h->fused_tail != (value != 0)) h->train_ws_stale = true;   // the workspace is carved per setting
    h->fused_tail = value != 0;
    return 0;
  }
  if (k == "infer_matmul_bf16x3") {
    if (h->infer_split != (value != 0)) h->folded = false;   // the split weight images are packed by the next eval forward
    h->infer_split = value != 0;
    return 0;
  }
  if (k == "ab_tiles_per_wg") { if (value < 0) return fail(h, "ab_tiles_per_wg must be >= 0"); h->ab_tiles_per_wg = (int)value; return 0; }
  if (k == "infer_tile_points") { if (value != 0 && value != 64 && value != 128) return fail(h, "infer_tile_points must be 0 (automatic), 64 or 128"); h->infer_tile_opt = (int)value; return 0; }
  if (k == "dg_cloud_parts") { if (value < 0 || value > 8) return fail(h, "dg_cloud_parts must be 0 (automatic) .. 8"); h->dg_parts_opt = (int)value; return 0; }
  if (k == "pn_cloud_parts") { if (value < 0 || value > 8) return fail(h, "pn_cloud_parts must be 0 (automatic) .. 8"); h->pn_parts_opt = (int)value; return 0; }
  for (const auto& ak : kAbKeys)
    if (k == ak.key) {
      const unsigned before = h->ab;
      h->ab = value ? (h->ab | ak.bit) : (h->ab & ~ak.bit);
      if ((before ^ h->ab) & AB_INFER_TILE64) h->folded = false;
      return 0;
    }
#ifdef ALIGNNET_ABLATE
  if (k == "ablate_dbg") { h->ablate_dbg = (int)value; return 0; }
  if (k == "ablate_mutation") { h->ablate_mutation = (int)value; return 0; }   // 1 = gather with the towers swapped, 2 = every rank keeps rank 0's rows of the loss gradient,
                                                                               // 3 = per-rank count behind a synchronised BatchNorm, 4 = a global weight-gradient term without 1 / world   // ablation build only: result-changing timing switches of the kernels
#endif
  if (const int rc = alignnet_train_set_option(h, k, value); rc >= 0) return rc;
  return fail(h, "alignnet_set_option: unknown key '" + k + "'");
}

extern "C" int alignnet_get_option(alignnet_handle* h, const char* key, int64_t* value)
{
  if (!h || !key || !value) return 1;
  const std::string k(key);
  if (k == "train_matmul_bf16") { *value = h->train_bf16 ? 1 : 0; return 0; }
  if (k == "infer_matmul_bf16x3") { *value = h->infer_split ? 1 : 0; return 0; }
  if (k == "allreduce_overlap") { *value = h->comm_overlap ? 1 : 0; return 0; }
  if (k == "train_dw_side_stream") { *value = h->dw_side; return 0; }
  if (k == "train_phase3_tile64") { *value = h->p3_tile64 ? 1 : 0; return 0; }
  if (k == "dropout_stream") { *value = (int64_t)h->dropout_stream; return 0; }
  if (k == "sync_bn") { *value = h->sync_bn ? 1 : 0; return 0; }
  if (k == "global_loss") { *value = h->global_loss ? 1 : 0; return 0; }
  if (k == "sync_bn_emulate_world") { *value = h->sync_emulate_world; return 0; }
  if (k == "comm_world") { *value = h->comm ? h->comm_world : 0; return 0; }
  if (k == "comm_order") { *value = h->comm_order; return 0; }
  if (k == "sync_collectives") { *value = h->sync_collectives; return 0; }
  if (k == "ab_tiles_per_wg") { *value = h->ab_tiles_per_wg; return 0; }
  if (k == "dg_cloud_parts") { *value = h->dg_parts_opt; return 0; }
  if (k == "infer_tile_points") { *value = h->infer_tile_opt; return 0; }
  if (k == "pn_cloud_parts") { *value = h->pn_parts_opt; return 0; }
  if (k == "ab_mask") { *value = h->ab; return 0; }
  for (const auto& ak : kAbKeys) if (k == ak.key) { *value = (h->ab & ak.bit) ? 1 : 0; return 0; }
  if (k == "comm_buckets") { *value = h->comm_buckets; return 0; }
  if (k == "last_backbone_kernel") { *value = h->last_kernel; return 0; }
  if (k == "last_train_kernel") { *value = h->last_train_kernel; return 0; }
  if (k == "train_fused_tail") { *value = h->fused_tail ? 1 : 0; return 0; }
  if (const int rc = alignnet_train_get_option(h, k, value); rc >= 0) return rc;
  return fail(h, "alignnet_get_option: unknown key '" + k + "'");
}

// ---------------------------------------------------------------------------------
// profiling hook (bench.py roofline leg)
// ---------------------------------------------------------------------------------
extern "C" int alignnet_profile_enable(alignnet_handle* h, int32_t on)
{
  if (!h) return 1;
  if (alignnet_drain_profile(h)) return 1;
  h->prof = on != 0;
  return 0;
}

extern "C" int alignnet_profile_read(alignnet_handle* h, double* backbone_ms, int64_t* backbone_launches, double* total_ms,
                                     int32_t reset)
{
  if (!h) return 1;
  if (alignnet_drain_profile(h)) return 1;
  if (backbone_ms) *backbone_ms = h->prof_backbone_ms;
  if (backbone_launches) *backbone_launches = h->prof_backbone_launches;
  if (total_ms) *total_ms = h->prof_total_ms;
  if (reset) {
    h->prof_backbone_ms = 0; h->prof_total_ms = 0; h->prof_backbone_launches = 0;
    for (int i = 0; i < PK_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_launches[i] = 0; }
  }
  return 0;
}

extern "C" int alignnet_profile_read_kernel(alignnet_handle* h, const char* name, double* ms, int64_t* launches)
{
  if (!h || !name) return 1;
  if (alignnet_drain_profile(h)) return 1;
  for (int i = 0; i < PK_COUNT; ++i)
    if (std::strcmp(name, kProfKernelNames[i]) == 0) {
      if (ms) *ms = h->prof_ms[i];
      if (launches) *launches = h->prof_launches[i];
      return 0;
    }
  return fail(h, std::string("alignnet_profile_read_kernel: unknown kernel '") + name + "'");
}

// ---------------------------------------------------------------------------------
// schedules (train.py:133-174)
// ---------------------------------------------------------------------------------
static double staircase(double base, double gstep, double decay_steps, double rate)
{
  return base * std::pow(rate, std::floor(gstep / decay_steps));
}

extern "C" int alignnet_get_state(alignnet_handle* h, alignnet_state* st)
{
  if (!h || !st) return 1;
  const alignnet_config& c = h->cfg;
  const double bs = c.batch_size > 0 ? c.batch_size : 1;
  const double per_epoch = bs * (double)(c.ntrain / (c.batch_size > 0 ? c.batch_size : 1));
  const double lr_ds = (double)c.lr_step * (c.lr_per_epoch ? per_epoch : 1.0);
  const double bn_ds = (double)c.bn_step * (c.bn_per_epoch ? per_epoch : 1.0);
  const double g = (double)h->step * bs;
  st->step = h->step;
  st->learning_rate = (float)std::max(lr_ds > 0 ? staircase(c.learning_rate, g, lr_ds, c.lr_rate) : (double)c.learning_rate, 1e-5);
  st->bn_decay = (float)std::min((double)c.bn_clip, 1.0 - (bn_ds > 0 ? staircase(c.bn_init, g, bn_ds, c.bn_rate) : (double)c.bn_init));
  return 0;
}

extern "C" int alignnet_set_step(alignnet_handle* h, int64_t step)
{
  if (!h) return 1;
  if (step < 0) return fail(h, "alignnet_set_step: negative step");
  h->step = step;
  return 0;
}
