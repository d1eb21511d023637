// hd_conv2d entry point: argument checks, tile / K-depth selection, dispatch to the two kernel families
//   conv_igemm_bk32.hip : 32-deep K tiles (half-line DMA pieces; any Cin % 8 == 0; the 16/32-channel layers)
//   conv_igemm_bk64.hip : 64-deep K tiles (full 128-byte-line DMA pieces, one barrier per 16 MFMAs; Cin % 64 == 0)
#include <stdlib.h>

#include "conv_params.h"

namespace {

int pick_bn(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }
// small problems: halve the M tile so that more of the 256 CUs get a block
int pick_bm(int M, int Cout) {
  int bn = pick_bn(Cout);
  int64_t blocks128 = (int64_t)hd_cdiv(M, 128) * hd_cdiv(Cout, bn);
  return (bn > 32 && blocks128 < 512) ? 64 : 128;
}

int fill_params(const hd_conv_args* a, ConvP& p) {
  HD_CHECK_ARG(a && a->x && a->w && a->y, "hd_conv2d: null pointer");
  HD_CHECK_ARG(a->C1 > 0 && a->C1 % 8 == 0 && a->C2 >= 0 && a->C2 % 8 == 0, "hd_conv2d: channel counts must be multiples of 8 (C1=%d C2=%d)", a->C1, a->C2);
  HD_CHECK_ARG((a->C2 == 0) == (a->x2 == nullptr), "hd_conv2d: x2/C2 mismatch");
  HD_CHECK_ARG(a->N > 0 && a->Ho > 0 && a->Wo > 0 && a->Cout > 0 && a->KH > 0 && a->KW > 0 && a->stride > 0, "hd_conv2d: bad extent");
  HD_CHECK_ARG(!(a->in_dil > 1 && (a->up1 || a->C2)), "hd_conv2d: in_dil excludes up1/x2");
  HD_CHECK_ARG(a->in_dil <= 2, "hd_conv2d: in_dil must be 1 or 2 (strides on the hot path)");
  HD_CHECK_ARG(a->C2 == 0 || (a->C1 % 32 == 0 && a->C2 % 32 == 0), "hd_conv2d: dual-source gather needs C1, C2 multiples of 32 (a K tile holds channels of one source)");
  HD_CHECK_ARG(!a->up1 || (a->Hin == 2 * a->Hsrc && a->Win == 2 * a->Wsrc), "hd_conv2d: up1 needs Hin=2*Hsrc");
  HD_CHECK_ARG((int64_t)a->N * a->Ho * a->Wo < (1ll << 31), "hd_conv2d: too many pixels");
  p.x = (const f16*)a->x;
  p.x2 = (const f16*)a->x2;
  p.w = (const f16*)a->w;
  p.bias = a->bias;
  p.res = (const f16*)a->res;
  p.mask = (const f16*)a->mask;
  p.y = a->y;
  p.stats = a->stats;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.Hin = a->Hin; p.Win = a->Win;
  p.C1 = a->C1; p.C2 = a->C2; p.Cin = a->C1 + a->C2;
  p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad = a->pad; p.up1 = a->up1; p.in_dil = a->in_dil < 1 ? 1 : a->in_dil;
  p.act = a->act; p.out_mode = a->out_mode;
  HD_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "hd_conv2d: in_scale / in_shift must be given together");
  p.in_scale = a->in_scale; p.in_shift = a->in_shift; p.in_relu = a->in_relu;
  p.pool2 = a->out_pool2;
  p.y2 = a->y2;
  p.bs_y = (const f16*)a->bs_y; p.bs_z = (const f16*)a->bs_z;
  p.bs_mean = a->bs_mean; p.bs_invstd = a->bs_invstd; p.bs_gamma = a->bs_gamma; p.bs_beta = a->bs_beta; p.bs_relu = a->bs_relu;
  HD_CHECK_ARG(!p.bs_y || (p.stats && p.bs_mean && p.bs_invstd && !a->mask && a->act == HD_ACT_NONE && a->out_mode == HD_OUT_NHWC_F16 && !a->bias),
               "hd_conv2d: bs_* (BatchNorm backward sums) need stats, bs_mean, bs_invstd and exclude mask / act / bias / non-f16 output");
  HD_CHECK_ARG(p.out_mode >= HD_OUT_NHWC_F16 && p.out_mode <= HD_OUT_NHWC_F32, "hd_conv2d: out_mode %d", p.out_mode);
  p.M = a->N * a->Ho * a->Wo;
  p.cin8 = p.Cin / 8;
  p.nchunks = a->KH * a->KW * p.cin8;
  p.nk = 0;
  p.Ktot = a->KH * a->KW * p.Cin;
  p.prio = 0;
  p.par = p.ph = p.pw = p.Hc = p.Wc = p.t0h = p.t0w = 0;
  p.gm = p.gn = 0;
  p.tgroup = 1;
  p.inv_cin8 = 1.0f / (float)p.cin8;
  p.inv_kw = 1.0f / (float)a->KW;
  int64_t xb = (int64_t)a->N * a->Hsrc * a->Wsrc * a->C1 * 2;
  int64_t x2b = a->x2 ? (int64_t)a->N * a->Hin * a->Win * a->C2 * 2 : 0;
  int64_t wb = (int64_t)a->Cout * p.Ktot * 2;
  HD_CHECK_ARG(xb < 0xFFFFFFF0ll && x2b < 0xFFFFFFF0ll && wb < 0xFFFFFFF0ll, "hd_conv2d: tensor larger than 4 GiB (buffer addressing)");
  p.xbytes = (unsigned)xb; p.x2bytes = (unsigned)x2b; p.wbytes = (unsigned)wb;
  return HD_OK;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

static bool g_small_ok = true;   // cleared while a tuning override forces an igemm variant

// the 16/32-channel 3x3 layers go to the direct small-channel kernel (conv3x3_small.hip); HD_CONV_SMALL=0 keeps them in the
// igemm family (A/B)
bool use_small(const ConvP& p) {
  static const int on = env_int("HD_CONV_SMALL", 1);
  return on && g_small_ok && hd_conv_small_eligible(p);
}

// the 64 -> 64 channel 3x3 layers go to the register-resident-weights kernel (conv3x3_c64.hip); HD_CONV_C64=0 keeps them in the igemm
// family (A/B)
bool use_c64(const ConvP& p) {
  static const int on = env_int("HD_CONV_C64", 1);
  return on && g_small_ok && hd_conv_c64_eligible(p);
}

// the ResNet stem (7x7 / stride 2, 8 -> 64 channels) goes to its own register-resident kernel (conv7x7s2_stem.hip); HD_CONV_STEM=0
// keeps it in the igemm family (A/B)
bool use_stem(const ConvP& p) {
  static const int on = env_int("HD_CONV_STEM", 1);
  return on && g_small_ok && hd_conv_stem_eligible(p);
}

// the 32 -> 128 channel 3x3 data gradient goes to its register-resident kernel (conv3x3_c32to128.hip); HD_CONV_C32=0: igemm family (A/B)
bool use_c32(const ConvP& p) {
  static const int on = env_int("HD_CONV_C32", 1);
  return on && g_small_ok && !p.stats && hd_conv_c32to128_eligible(p);
}

// decoder block 3's first convolution (upsample + concat, 64 + 64 -> 32 channels) goes to its register-resident kernel
// (conv3x3_cat128to32.hip); HD_CONV_CAT=0: igemm family (A/B)
bool use_cat(const ConvP& p) {
  static const int on = env_int("HD_CONV_CAT", 1);
  return on && g_small_ok && hd_conv_cat128to32_eligible(p);
}

// Large-tile GEMM (gemm_w8.hip) for the plain-GEMM problems that are K-heavy and big enough to fill the chip with its tiles: the box
// head's fc6 / fc7 and their data gradients.  Bit-identical to the igemm family (same products, same K order), so the choice may
// look at the batch.  Tiles: 256 x 128 (8 waves) when that grid is one (nearly) full round of the 256 CUs, else 128 x 128 (4 waves,
// two blocks per CU) from 200 tiles up (tools/probe_gemm8.py, one box: fc6 forward on 8 192 rows 254 -> 210 us, on 4 096 rows 156 -> 140,
// its data gradient 197 -> 135, fc7 43.7 -> 26.1; a 256 x 256 instance was built too and never won: 144 us on the data gradient).
// Not for the output-bound 1x1 layers: with K <= 256 or a residual the 4-wave family's LDS-transposed epilogue (16-byte coalesced
// loads / stores) beats this kernel's 8-byte-per-row register epilogue (bottleneck conv3 at 24 x 19 x 19: 18.2 vs 26.8 us).
// hd_gemm_w8_mode (tests / tools): -1 = this rule, 0 = never, 128 / 1128 = that tile wherever eligible.  HD_GEMM8=0: off (A/B).
static int g_gemm8_mode = -1;
extern "C" int hd_gemm_w8_mode(int mode) {
  HD_CHECK_ARG(mode == -1 || mode == 0 || mode == 128 || mode == 1128, "hd_gemm_w8_mode: mode in {-1, 0, 128, 1128}");
  g_gemm8_mode = mode;
  return HD_OK;
}
bool hd_gemm_w8_eligible(const ConvP& p);
void hd_gemm_w8_launch(ConvP& p, int tile, hipStream_t s);
static int choose_gemm8(const ConvP& p) {
  static const int on = env_int("HD_GEMM8", 1);
  if (!on || g_gemm8_mode == 0 || !g_small_ok || !hd_gemm_w8_eligible(p)) return 0;
  if (g_gemm8_mode > 0) return g_gemm8_mode;
  if (p.res || p.M < 2048 || p.Cout < 512 || p.Ktot < 512) return 0;
  const int64_t t128 = (int64_t)hd_cdiv(p.M, 256) * hd_cdiv(p.Cout, 128), t1128 = (int64_t)hd_cdiv(p.M, 128) * hd_cdiv(p.Cout, 128);
  if (t128 >= 160 && t128 <= 256) return 128;
  if (t1128 >= 200) return 1128;
  return 0;
}

// tuning hook (tools/tune_conv.py): force the tile / K-depth / stage choice of the igemm family; -1 = heuristic
static int g_ov_bm = -1, g_ov_bn = -1, g_ov_bk = -1, g_ov_deep = -1;
extern "C" int hd_conv_tune_override(int bm, int bn, int bk, int deep) {
  HD_CHECK_ARG((bm == -1 || bm == 64 || bm == 128) && (bn == -1 || bn == 32 || bn == 64 || bn == 128) && (bk == -1 || bk == 32 || bk == 64) &&
               deep >= -1 && deep <= 1, "hd_conv_tune_override: bm in {64,128}, bn in {32,64,128}, bk in {32,64}, deep in {0,1} or -1");
  g_ov_bm = bm; g_ov_bn = bn; g_ov_bk = bk; g_ov_deep = deep;
  g_small_ok = (bm == -1 && bn == -1 && bk == -1 && deep == -1);
  return HD_OK;
}

#ifdef HD_CONV_TRACE
// profiling build only (not part of the shipped ABI): where the 64-deep kernels write their per-block stamps
static unsigned long long* g_trace = nullptr;
extern "C" int hd_conv_trace_buffer(void* buf) {
  g_trace = (unsigned long long*)buf;
  return HD_OK;
}
#endif

// 3x3 / stride-1 layers with >= 128 output channels (or a decoder concat) go to the 8-wave input-patch family when its 8-wide
// tiles cover the feature map well.  Tile choice by a cost model fitted to per-block timelines (tools/w8_trace.py, shader clocks):
// one block per CU, so a launch costs ceil(blocks / 256) rounds of (fixed + K steps x step), fixed = set-up + first stage +
// epilogue.  Measured against the per-shape sweep of one training step's 136 launch shapes (tools/tune_w8.py).
// test / tuning hook: n > 0 evaluates the tile cost model at batch n whatever the launch's (rounds 2-5: 8 -- image n of a batched launch is
// then bit-identical to the same image alone); 0 (default) = the launch's own batch
static int g_nominal_batch = 0;
extern "C" int hd_conv_nominal_batch(int n) {
  HD_CHECK_ARG(n >= 0 && n <= 4096, "hd_conv_nominal_batch: n in [0, 4096]");
  g_nominal_batch = n;
  return HD_OK;
}

static int choose_p8(const ConvP& p, bool allow_m160 = true) {
  static const int on = env_int("HD_CONV_P8", 1);
  if (!on || !hd_conv_p8_eligible(p)) return -1;
  if (p.Cout < 128 && !p.x2) return -1;                     // 64 -> 64 layers: the 4-wave family's 2-3 blocks per CU win
  static const int th[4] = {32, 16, 32, 16}, bn[4] = {128, 128, 64, 64};
  // HD_W8_TS: the step-split main loop (conv3x3_w8.hip, TS) -- 1 (default): for the 128 x 64 tile, whose sub-step split prefetched only
  // two 540-clock steps ahead (the 512-channel layers waited on their weights: 16x20x512 26.7 -> 21.2 us); 2: for every WK >= 2 tile
  // (measured equal or slower on the 128-wide tiles: twice the DMA issue per LOAD phase, profiles/r05_probe_w8_ts.txt); 0: off
  static const int ts = env_int("HD_W8_TS", 1);
  static const double step_ss[4] = {1270., 790., 800., 540.}, step_ts[4] = {1270., 790., 860., 450.}, fixed[4] = {11500., 8000., 9000., 7000.};
  double step[4];
  for (int c = 0; c < 4; ++c) step[c] = ((ts >= 2 && c >= 1) || (ts == 1 && c == 3)) ? step_ts[c] : step_ss[c];
  const int nk = p.nchunks / 8;
  int best = -1;
  double best_t = 1e30;
  // The tiles split K differently (WK), so they do not round identically.  Rounds 2-5 evaluated this model at the TRAINING batch (8) whatever
  // the launch's, so that image n of a batched launch was bit-identical to the same image alone -- the builder's convenience, not a property
  // of the reference (cuDNN picks algorithms per shape), and it kept the 24-image detector launches on tiles chosen for 8 images.  Round 6
  // (review item 1b): the model sees the launch's own batch; a problem of a given shape always gets the same tile (run-to-run identical),
  // images of differently batched launches agree to fp16 rounding (tests: stated tolerance).  HD_CONV_NOMINAL_BATCH=8 / hd_conv_nominal_batch(8) restore the old rule.
  static const int nominal = env_int("HD_CONV_NOMINAL_BATCH", 0);
  const int nominal_batch = g_nominal_batch > 0 ? g_nominal_batch : (nominal > 0 ? nominal : p.N);
  for (int c = 0; c < 4; ++c) {
    if (bn[c] == 128 && p.Cout <= 64) continue;
    const int64_t tm = (int64_t)nominal_batch * hd_cdiv(p.Ho, th[c]) * hd_cdiv(p.Wo, 8);
    const double eff = (double)((int64_t)nominal_batch * p.Ho * p.Wo) / (double)(tm * th[c] * 8);
    if (eff < 0.7) continue;
    const int64_t blocks = tm * hd_cdiv(p.Cout, bn[c]);
    const double t = (double)hd_cdiv(blocks, 256) * (fixed[c] + nk * step[c])   /* nothing here may depend on p.stats: hd_conv2d_stats_rows asks before the slab exists */;
    if (t < best_t) {
      best_t = t;
      best = c;
    }
  }
  best = ((ts >= 2 && best >= 1) || (ts == 1 && best == 3)) ? best + 4 : best;
  // Round 6: the 160- / 320-pixel x 64-channel tiles (conv3x3_m160.hip, cfg 8 / 9; producer / consumer wave roles).  Same model: rounds x
  // (fixed + K steps x step), constants from per-block stamps (tools/w8_trace.py, CFGS=18,19: K loop 450 / 780 clocks per step, set-up +
  // epilogue 9 - 11 k).  No coverage gate: these kernels win on maps they cover badly too (8 x 10 x 10 x 256: 17.5 -> 13.5 us at 21 %
  // coverage) because what they replace there is the 4-wave implicit-GEMM family at ~900 clocks per K step of a lone 64 x 64 block --
  // estimated below when no 8-wide tile qualifies (profiles/r06_probe_m160_v4.txt has every signature of a step).  HD_CONV_M160=0: off (A/B).
  static const int m160_on = env_int("HD_CONV_M160", 1);
  static const double m160_fixed[2] = {(double)env_int("HD_M160_FIXED", 9000), (double)env_int("HD_M320_FIXED", 10500)};
  static const double m160_step[2] = {(double)env_int("HD_M160_STEP", 460), (double)env_int("HD_M320_STEP", 790)};
  if (m160_on && allow_m160 && hd_conv_m160_eligible(p)) {
    if (best < 0) {
      const int64_t b64 = (int64_t)hd_cdiv((int64_t)nominal_batch * p.Ho * p.Wo, 64) * hd_cdiv(p.Cout, 64);
      best_t = (double)hd_cdiv(b64, 512) * (4000. + nk * 900.);              // 4-wave family, two 64 x 64 blocks per CU
    }
    // cfg 10: the 96-pixel tile (4 x 24: the detector's 19 x 19 / 10 x 10 / 5 x 5 maps, which the 40-wide tiles cover to 47 % and less); single-source
    // problems only.  TWO blocks per CU (80 KiB of LDS, 116 registers): up to 256 blocks run alone (per-block stamps, tools/w8_trace.py CFGS=20:
    // set-up + epilogue 6.6 k clocks, 350 per K step), more than that in co-resident pairs, 512 per round (11 k, 420 per step each: fitted to whole launches, tools/probe_m160.py -- a full chip of pairs clocks lower than the stamps of one launch say).
    static const double m96_fixed[2] = {(double)env_int("HD_M96_FIXED", 6600), (double)env_int("HD_M96_FIXED2", 11000)};
    static const double m96_step[2] = {(double)env_int("HD_M96_STEP", 350), (double)env_int("HD_M96_STEP2", 420)};
    static const int m96_on = env_int("HD_CONV_M96", 1);
    for (int v = 0; v < 3; ++v) {
      if (v == 2 && (!m96_on || p.x2)) continue;
      const int th_ = v == 1 ? 8 : 4, tw_ = v == 2 ? 24 : 40;
      const int64_t tm = (int64_t)nominal_batch * hd_cdiv(p.Ho, th_) * hd_cdiv(p.Wo, tw_);
      const int64_t blocks = tm * hd_cdiv(p.Cout, 64);
      double t;
      if (v == 2) {
        const int pair = blocks > 256;
        t = (double)(pair ? hd_cdiv(blocks, 512) : 1) * (m96_fixed[pair] + nk * m96_step[pair]);
      } else {
        t = (double)hd_cdiv(blocks, 256) * (m160_fixed[v] + nk * m160_step[v]);
      }
      if (t < best_t) {
        best_t = t;
        best = 8 + v;
      }
    }
  }
  return best;
}

struct TileChoice {
  int bm, bn;
  bool use64, deep;
  int p8cfg;             // p8cfg >= 0: the 8-wave input-patch family (conv3x3_w8.hip) with that tile id
};

// tuning hook of the 8-wave patch-staged family (tools/tune_w8.py): cfg -1 = cost model, -2 = never, -3 = cost model without the 160-pixel tile
// (round 5's choice: A/B of conv3x3_m160.hip inside one process, tools/probe_m160.py), 10..13 = force that tile, 15..17 = force
// the step-split main loop (TS) of tiles 11..13, 18 / 19 / 20 = force the 160- / 320- / 96-pixel x 64-channel tile (conv3x3_m160.hip)
// wherever the family is eligible
static int g_w8_cfg = -1;
extern "C" int hd_conv_tune_w8(int cfg, int nslices) {
  (void)nslices;
  HD_CHECK_ARG(cfg == -1 || cfg == -2 || cfg == -3 || (cfg >= 10 && cfg <= 13) || (cfg >= 15 && cfg <= 20), "hd_conv_tune_w8: cfg in {-1, -2, -3, 10..13, 15..20}");
  g_w8_cfg = cfg;
  return HD_OK;
}

// Tile / K-depth / stage choice of the igemm family.  Rules come from an exhaustive per-shape search over the 136 launch
// shapes of one training step (tools/tune_conv.py; 7.71 ms with the previous rules, 7.37 ms with these, 7.13 ms with a
// perfect per-shape table):
//   * 1x1 convolutions onto > 64 channels are output-write bound (small K, wide N): 64-wide N tiles, 64 rows unless the
//     grid is already large, 64-deep K tiles only from 256 input channels, 2 stages;
//   * any other conv onto > 64 channels whose 64x128 grid cannot give each CU more than one block: 64x64 tiles, deep ring;
//   * otherwise: the widest N tile that the channel count fills, 64 rows below 512 blocks, 64-deep K where the channel
//     count allows, deep ring only at <= 1 block per CU.
static TileChoice choose_tile(const ConvP& p) {
  TileChoice c;
  const bool can64_ch = (p.Cin % 64 == 0) && (p.C2 == 0 || (p.C1 % 64 == 0 && p.C2 % 64 == 0));
  static const int small_n = env_int("HD_CONV_SMALLN", 512);
  static const int force_bk = env_int("HD_CONV_BK", 0);
  static const int force_deep = env_int("HD_CONV_DEEP", -1);
  static const int old_rules = env_int("HD_CONV_OLD_RULES", 0);     // A/B knob for the rule set below
  static const int k1_big = env_int("HD_CONV_K1_BIG", 1024), k1_bk64 = env_int("HD_CONV_K1_BK64", 256);     // A/B knobs of the 1x1 rule
  if (!old_rules && p.KH * p.KW == 1 && p.Cout > 64) {
    c.bn = 64;
    c.bm = ((int64_t)hd_cdiv(p.M, 128) * hd_cdiv(p.Cout, 64) < k1_big) ? 64 : 128;
    c.use64 = can64_ch && p.Cin >= k1_bk64;
    c.deep = false;
  } else if (!old_rules && p.Cout > 64 && (int64_t)hd_cdiv(p.M, 64) * hd_cdiv(p.Cout, 128) <= 256) {
    c.bm = 64;
    c.bn = 64;
    c.use64 = can64_ch;
    c.deep = true;
  } else {
    c.bn = pick_bn(p.Cout);
    c.bm = pick_bm(p.M, p.Cout);
    int64_t blocks = (int64_t)hd_cdiv(p.M, c.bm) * hd_cdiv(p.Cout, c.bn);
    if (c.bn == 128 && c.bm == 64 && blocks < small_n) {
      c.bn = 64;
      blocks = (int64_t)hd_cdiv(p.M, c.bm) * hd_cdiv(p.Cout, c.bn);
    }
    c.use64 = can64_ch && c.bn > 32;
    c.deep = blocks <= 256;
  }
  // Round 6: the deep ring everywhere.  Rounds 3-5 chose the stage count per shape from WARM back-to-back timings (deep only at <= 1 block
  // per CU); in STEP ORDER every layer's weights are a first touch from HBM / the Infinity Cache and the extra K tile in flight is worth more
  // than the third co-resident block: same-box A/B of the whole step, two alternations, HD_CONV_DEEP=0 / rule / 1: 9.860 / 9.773 / 9.713 and
  // 9.897 / 9.753 / 9.708 ms, detector_conv 4.06 / 3.97 / 3.89 ms.  (Four stages in the 64-deep family: 9.93 -- two blocks per CU are too few.)
  c.deep = true;
  // experiment knobs: HD_CONV_BK in {0 auto, 32, 64}; HD_CONV_DEEP in {-1 auto, 0, 1}; hd_conv_tune_override
  if (g_ov_bn > 0) c.bn = g_ov_bn;
  if (g_ov_bm > 0) c.bm = g_ov_bm;
  if (c.bn == 32) c.bm = 128;                     // the 32-wide tile only exists with 128 rows
  if (g_ov_bn > 0 || g_ov_bm > 0) c.use64 = can64_ch && c.bn > 32;
  if (force_bk == 32 || g_ov_bk == 32 || c.bn == 32) c.use64 = false;
  if (g_ov_bk == 64) c.use64 = can64_ch;
  if (force_deep >= 0) c.deep = force_deep != 0;
  if (g_ov_deep >= 0) c.deep = g_ov_deep != 0;
  c.p8cfg = -1;
  if (g_w8_cfg >= 10 && g_w8_cfg < 18 && hd_conv_p8_eligible(p) && g_small_ok) c.p8cfg = g_w8_cfg - 10;
  if (g_w8_cfg >= 18 && hd_conv_m160_eligible(p) && g_small_ok && !(g_w8_cfg == 20 && p.x2)) c.p8cfg = g_w8_cfg - 10;
  if ((g_w8_cfg == -1 || g_w8_cfg == -3) && g_small_ok && g_ov_bm < 0 && g_ov_bn < 0 && g_ov_bk < 0) c.p8cfg = choose_p8(p, g_w8_cfg == -1);
  return c;
}

extern "C" int hd_conv2d_stats_rows(const hd_conv_args* a) {
  if (!a) return HD_E_ARG;
  ConvP p;
  int rc = fill_params(a, p);
  if (rc) return rc;
  if (use_small(p)) return hd_conv_small_tiles(p);
  if (use_c64(p)) return hd_conv_c64_rows(p);
  if (use_stem(p)) return hd_conv_stem_rows(p);
  if (use_cat(p)) return hd_conv_cat128to32_rows(p);
  const TileChoice c = choose_tile(p);
  if (c.p8cfg >= 8) return hd_conv_m160_tiles(p, c.p8cfg == 9 ? 8 : 4, c.p8cfg == 10 ? 24 : 40);
  if (c.p8cfg >= 0) return hd_conv_p8_tiles(p, c.p8cfg);
  return hd_cdiv(p.M, c.bm);
}

extern "C" int hd_wgrad(const hd_wgrad_args* a, void* stream);
extern "C" int hd_conv2d(const hd_conv_args* a, void* stream);

// does this problem run in a kernel whose epilogue implements hd_conv_args.bs_* ?  (the 8-wave patch-staged 3x3 family)
static bool bstat_kernel(const ConvP& p) {
  static const int on = env_int("HD_CONV_BSTAT", 1);
  if (!on || use_small(p) || p.in_scale || p.in_dil != 1 || p.out_mode != HD_OUT_NHWC_F16) return false;
  if (use_c64(p)) return true;                         // the register-resident 64 -> 64 kernel implements them too
  return choose_tile(p).p8cfg >= 0;
}

// is this problem (out_pool2 set) routed to a kernel that implements the pooled / split output?
static bool pool2_kernel(const ConvP& p) {
  if (!p.pool2) return false;
  if (use_small(p)) return hd_conv_small_pool2_ok(p);
  if (use_c64(p) || use_stem(p)) return false;
  if (use_c32(p)) return true;                         // (its eligibility includes the out_pool2 = 64 / y2 form)
  if (use_cat(p) || choose_gemm8(p)) return false;
  const int cfg = choose_tile(p).p8cfg;
  return cfg >= 8 ? hd_conv_m160_pool2_ok(p) : (cfg >= 0 && hd_conv_p8_pool2_ok(p));
}

// does hd_conv2d implement out_pool2 for this problem?
extern "C" int hd_conv2d_pool2_ok(const hd_conv_args* a) {
  ConvP p;
  if (!a || fill_params(a, p)) return 0;
  return pool2_kernel(p) ? 1 : 0;
}

extern "C" int hd_conv2d_bstat_ok(const hd_conv_args* a) {
  ConvP p;
  if (fill_params(a, p)) return 0;
  return bstat_kernel(p) ? 1 : 0;
}

// Data gradient + weight gradient of one layer (both read the same dY): ONE grid when the convolution runs in the 8-wave
// patch-staged family and the weight gradient in its 8-wave kernel (conv3x3_w8_wgrad_kernel), otherwise hd_conv2d then hd_wgrad.
// Bit-identical to the two calls either way.  HD_FUSE_DGRAD_WGRAD=0: always two launches (A/B).
extern "C" int hd_conv2d_wgrad(const hd_conv_args* a, const hd_wgrad_args* wa, void* stream) {
  static const int fuse_on = env_int("HD_FUSE_DGRAD_WGRAD", 1);
  HD_CHECK_ARG(a && wa, "hd_conv2d_wgrad: null pointer");
  ConvP p;
  int rc = fill_params(a, p);
  if (rc) return rc;
  HD_CHECK_ARG(!p.pool2 || pool2_kernel(p), "hd_conv2d_wgrad: out_pool2 is not implemented for this problem; ask hd_conv2d_pool2_ok first");
  if (fuse_on && !use_small(p) && !use_c64(p) && !use_stem(p) && !p.in_scale && !p.x2 && (!p.stats || p.bs_y) && p.in_dil == 1 && hd_wgrad_takes_w8(wa)) {
    const TileChoice c = choose_tile(p);
    if (c.p8cfg >= 0 && c.p8cfg < 8) {             // (the 160- / 320-pixel tiles have no fused weight-gradient grid: two launches below)
      static const int w8_prio = env_int("HD_W8_PRIO", 0);
      p.prio = w8_prio;
#ifdef HD_CONV_TRACE
      p.trace = nullptr;
      p.trace_tid = 0;
#endif
      hd_conv_launch_p8_wgrad(p, c.p8cfg, wa, (hipStream_t)stream);
      HD_CHECK_LAUNCH();
      return HD_OK;
    }
  }
  rc = hd_conv2d(a, stream);
  if (rc) return rc;
  return hd_wgrad(wa, stream);
}

// n independent convolutions in ONE grid when they all resolve to the same 4-wave igemm variant (single source, no parity classes,
// none of them claimed by the small-channel or the 8-wave kernels); otherwise n hd_conv2d calls.  The tile of the FIRST problem
// (callers put the largest first) serves all of them: in this family the tile shape does not change a single output bit.
// HD_CONV_MULTI=0: always separate launches (A/B).
extern "C" int hd_conv2d_multi(const hd_conv_args* args, int n, void* stream) {
  static const int multi_on = env_int("HD_CONV_MULTI", 1);
  HD_CHECK_ARG(args && n >= 1, "hd_conv2d_multi: bad args");
  bool ok = multi_on && n >= 2 && n <= HD_CONV_MULTI_MAX && g_small_ok && g_w8_cfg < 0;
  static ConvMulti mp;          // 3.3 KB: not on the stack of a ctypes call; the boundary is not re-entrant (SURVEY 8b "Threading")
  TileChoice c0 = {};
  for (int i = 0; ok && i < n; ++i) {
    ConvP& p = mp.p[i];
    int rc = fill_params(&args[i], p);
    if (rc) return rc;
    if (use_small(p) || use_c64(p) || use_stem(p) || use_c32(p) || p.in_scale || p.x2 || p.in_dil != 1 || p.bs_y || p.pool2 || choose_gemm8(p)) { ok = false; break; }
    const TileChoice c = choose_tile(p);
    if (c.p8cfg >= 0) { ok = false; break; }
    if (i == 0) c0 = c;
    else if (c.use64 != c0.use64) { ok = false; break; }
#ifdef HD_CONV_TRACE
    p.trace = nullptr;
    p.trace_tid = 0;
#endif
  }
  if (ok) {
    mp.n = n;
    const bool launched = c0.use64 ? hd_conv_launch_bk64_multi(mp, c0.bm, c0.bn, c0.deep, (hipStream_t)stream)
                                   : hd_conv_launch_bk32_multi(mp, c0.bm, c0.bn, c0.deep, (hipStream_t)stream);
    if (launched) {
      HD_CHECK_LAUNCH();
      return HD_OK;
    }
  }
  // Not one igemm grid: the members the tile model sends to the 96-pixel tile (conv3x3_m160.hip, cfg 10: the small pyramid levels) still share
  // ONE grid -- the same blocks their own launches would run, bit for bit -- the others are launched one by one.  HD_CONV_M96_MULTI=0: off (A/B).
  static const int m96_multi_on = env_int("HD_CONV_M96_MULTI", 1);
  bool taken[HD_CONV_MULTI_MAX] = {};
  if (multi_on && m96_multi_on && n >= 2 && n <= HD_CONV_MULTI_MAX && g_small_ok && g_w8_cfg < 0) {
    static ConvMulti m96;
    m96.n = 0;
    for (int i = 0; i < n; ++i) {
      ConvP& p = m96.p[m96.n];
      int rc = fill_params(&args[i], p);
      if (rc) return rc;
      if (use_small(p) || use_c64(p) || use_stem(p) || use_c32(p) || use_cat(p) || choose_gemm8(p) || p.pool2 || p.x2) continue;
      if (choose_tile(p).p8cfg != 10) continue;
      HD_CHECK_ARG(!p.bs_y || bstat_kernel(p), "hd_conv2d_multi: bs_* on a problem whose kernel does not implement them");
#ifdef HD_CONV_TRACE
      p.trace = nullptr;
      p.trace_tid = 0;
#endif
      taken[i] = true;
      ++m96.n;
    }
    if (m96.n >= 2) {
      hd_conv_launch_m96_multi(m96, (hipStream_t)stream);
      HD_CHECK_LAUNCH();
    } else {
      for (int i = 0; i < n; ++i) taken[i] = false;
    }
  }
  for (int i = 0; i < n; ++i) {
    if (taken[i]) continue;
    int rc = hd_conv2d(&args[i], stream);
    if (rc) return rc;
  }
  return HD_OK;
}

extern "C" int hd_conv2d(const hd_conv_args* a, void* stream) {
  ConvP p;
  int rc = fill_params(a, p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
#ifdef HD_CONV_TRACE
  p.trace = g_trace;
  p.trace_tid = env_int("HD_TRACE_TID", 0);
#endif
  HD_CHECK_ARG(!p.bs_y || bstat_kernel(p), "hd_conv2d: bs_* (BatchNorm backward sums) are implemented by the 8-wave and the 64-channel 3x3 kernels only; "
                                           "ask hd_conv2d_bstat_ok first");
  HD_CHECK_ARG(!p.pool2 || pool2_kernel(p),
               "hd_conv2d: out_pool2 is implemented by the small-channel 3x3 kernel (all channels pooled), the 32 -> 128 channel kernel "
               "(64 pooled channels + y2) and the 8-wave 3x3 family (pooled channels a multiple of 128); ask hd_conv2d_pool2_ok first");
  if (use_small(p)) {
    hd_conv_launch_small(p, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  if (use_c64(p)) {
    hd_conv_launch_c64(p, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  if (use_stem(p)) {
    hd_conv_launch_stem(p, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  if (use_c32(p)) {
    hd_conv_launch_c32to128(p, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  if (use_cat(p)) {
    hd_conv_launch_cat128to32(p, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  if (const int g8 = choose_gemm8(p)) {
    hd_gemm_w8_launch(p, g8, s);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  HD_CHECK_ARG(!p.in_scale, "hd_conv2d: consumer-side BatchNorm (in_scale / in_shift) is implemented by the small-channel 3x3 kernel only "
                            "(3x3 / stride 1 / pad 1, one source, C1 in {8,16,32}, Cout in {16,32} or a <= 16-channel fp32 head)");
  // stride-2 data gradients: four output-parity classes, each walking only the taps that meet non-zero input (conv_params.h)
  static const int par_on = env_int("HD_CONV_PARITY", 1);
  const bool par = par_on && p.in_dil == 2 && !p.stats && p.stride == 1 && (p.cin8 % 4) == 0 && p.out_mode == HD_OUT_NHWC_F16 && g_small_ok &&
                   g_w8_cfg < 0;
  const int M_full = p.M;
  if (par) p.M = p.N * ((p.Ho + 1) / 2) * ((p.Wo + 1) / 2);      // tile choice / grid for the largest class (ph = pw = 0)
  const TileChoice c = choose_tile(p);
  static const int w8_prio = env_int("HD_W8_PRIO", 0);
  p.prio = w8_prio;
  const int bm = c.bm, bn = c.bn;
  const bool use64 = c.use64, deep = c.deep;
  if (par && c.p8cfg < 0 && (c.use64 ? (p.cin8 % 8) == 0 : true)) {
    p.par = 1;                     // the launchers add gridDim.y = 4
  } else {
    p.M = M_full;
  }
  if (c.p8cfg >= 8) {
    hd_conv_launch_m160(p, c.p8cfg == 9 ? 8 : 4, c.p8cfg == 10 ? 24 : 40, s);
  } else if (c.p8cfg >= 0) {
    hd_conv_launch_p8(p, c.p8cfg, s);
  } else if (use64) hd_conv_launch_bk64(p, bm, bn, deep, s);
  else hd_conv_launch_bk32(p, bm, bn, deep, s);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
