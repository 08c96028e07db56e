// Whole adapted ViT forward / backward as stream-ordered sequences of the kernels in this
// library (gfx950).  What `model(x)` and `loss.backward()` of
// /root/reference/image_classification/vit_cp.py:46-49 execute, with the patched forwards of
// /root/reference/src/cara/cara.py:15-95 in factored form: no dW is ever
// materialised and no dense weight gradient exists; the frozen backbone only propagates dX.
// Host side is plain C++ in the library so a train step costs two FFI calls and the whole
// sequence can be captured in a hipGraph (nothing here allocates or synchronises).
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "gemm8.h"

namespace {

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  }
};

struct LayerWs {
  size_t x_in, x_mid, mean1, rstd1, mean2, rstd2;
  size_t xn1, qkv, lse, ao, xn2, u, h;
  size_t T[4], Tt[4];   // qkv, proj, fc1, fc2
};

struct Ws {
  size_t pack, patches, emb, head_wb, clsn, meanF, rstdF, x_last;
  LayerWs layer[64];
  // backward
  size_t dx, dXn, dAO, slabs, dclsn, gscratch, gemm_scratch;
  // dY of the four linears, G' = dY Vs and its transpose: one set, reused by every block (everything runs in
  // stream order on the caller's stream)
  struct Bwd { size_t dyb_fc2, dyb_proj, dH, dQKV, G[4], Gt[4]; } bwd;
  size_t dU[4], dVs[4], dc[4];
  size_t slabU[4], slabV[4], strideU[4], strideV[4];   // per linear: depth regions of tskinny slabs
  // exact weight-dropout mode: merged weights of every layer (and their transposes), transposed activations, dense dW
  size_t weff[4], wefft[4], dYt, Xt, dWd, xscratch;
  // order-2 QKV tensorisation (cara_geom::cp_length == 2): the dense deltas of every layer and their transposes, the dense
  // dD = x^T dY_k of every (layer, projection), split-K slabs of one such product, scratch of cara_dense_delta_grad
  size_t dd, ddt, dD, dd_slabs, dd_scratch;
  int nslab;   // split-K slabs of the dense dW product
  int ldk;   // row stride of the transposed activations: M rounded up to the GEMM's K granule
  size_t total;
  int M, ldt;
};

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// split-K slabs of the dense dW = dY^T X of one linear (cara_gemm_tn_f32): enough of them that (out / 128)(in / 128) tiles
// times slabs fill the chip (proj: 36 tiles -- at four slabs each workgroup walked 98 K steps on a quarter of the CUs:
// 73 us for 15 GF), at least 512 token rows per slab, at most 16
int dw_slabs(int out, int in, int M) {
  const int tiles = ((out + 127) / 128) * ((in + 127) / 128);
  int n = (640 + tiles - 1) / tiles;
  n = n > 16 ? 16 : n;
  while (n > 1 && M / n < 512) --n;
  return n < 1 ? 1 : n;
}

bool epi_riders_geom(const cara_geom* g, int Mr, bool exact);
int env_once(const char* name, int dflt);
bool save_gelu_grad(const cara_geom* g, const cara_vit_shape* s);
bool layout(const cara_geom* g, const cara_vit_shape* s, Ws* w) {
  if (!g || !s || g->depth <= 0 || g->depth > 64 || g->dim % g->heads || g->dim / g->heads != 64) return false;
  if (!(g->Rp == 32 || g->Rp == 64) || g->rank > g->Rp || s->B <= 0 || s->tokens <= 1 || s->tokens > 608) return false;
  if (s->img % s->patch || (s->img / s->patch) * (s->img / s->patch) + 1 != s->tokens) return false;
  if ((s->chans * s->patch * s->patch) % 64 || g->dim % 256) return false;
  const size_t D = g->dim, M = (size_t)s->B * s->tokens, Rp = g->Rp;
  const size_t ldt = (M + 31) / 32 * 32;
  w->M = (int)M;
  w->ldt = (int)ldt;
  Carver c;
  cara_pack_layout pl;
  if (cara_pack_offsets(g, &pl) != CARA_OK) return false;
  w->pack = c.take(pl.total);
  const size_t P = s->tokens - 1, kp = (size_t)s->chans * s->patch * s->patch;
  w->patches = c.take((size_t)s->B * P * kp * 2);
  w->emb = c.take((size_t)s->B * P * D * 4);
  w->head_wb = c.take((size_t)s->num_classes * D * 2);
  w->clsn = c.take((size_t)s->B * D * 2);
  w->meanF = c.take((size_t)s->B * 4);
  w->rstdF = c.take((size_t)s->B * 4);
  w->x_last = c.take(M * D * 4);
  for (int l = 0; l < g->depth; ++l) {
    LayerWs& L = w->layer[l];
    L.x_in = c.take(M * D * 4);
    L.x_mid = c.take(M * D * 4);
    L.mean1 = c.take(M * 4); L.rstd1 = c.take(M * 4); L.mean2 = c.take(M * 4); L.rstd2 = c.take(M * 4);
    L.xn1 = c.take(M * D * 2);
    L.qkv = c.take(M * 3 * D * 2);
    L.lse = c.take((size_t)s->B * g->heads * s->tokens * 4);
    L.ao = c.take(M * D * 2);
    L.xn2 = c.take(M * D * 2);
    L.u = c.take(M * 4 * D * 2);
    L.h = c.take(M * 4 * D * 2);
    for (int i = 0; i < 4; ++i) {
      L.T[i] = c.take(M * Rp * 2);
      L.Tt[i] = c.take(Rp * ldt * 2);
    }
  }
  w->dx = c.take(M * D * 4);
  w->dXn = c.take(M * D * 2);
  w->dAO = c.take(M * D * 2);
  {
    Ws::Bwd& R = w->bwd;
    R.dyb_fc2 = c.take(M * D * 2);
    R.dyb_proj = c.take(M * D * 2);
    R.dH = c.take(M * 4 * D * 2);
    R.dQKV = c.take(M * 3 * D * 2);
    for (int i = 0; i < 4; ++i) {
      R.G[i] = c.take(M * Rp * 2);
      R.Gt[i] = c.take(Rp * ldt * 2);
    }
  }
  w->slabs = 0;
  w->dclsn = c.take((size_t)s->B * D * 2);
  w->gscratch = c.take(cara_factor_grad_scratch_bytes(g));
  w->gemm_scratch = c.take(cara_gemm_scratch_bytes());   // stream-K partial tiles + flags (workspace is zeroed at allocation)
  const size_t ins[4] = {D, D, D, 4 * D}, outs[4] = {3 * D, D, 4 * D, D};
  for (int i = 0; i < 4; ++i) {
    w->dU[i] = c.take((size_t)g->depth * ins[i] * Rp * 4);
    w->dVs[i] = c.take((size_t)g->depth * outs[i] * Rp * 4);
    w->dc[i] = c.take((size_t)g->depth * outs[i] * 4);
    w->strideU[i] = (cara_tskinny_scratch_bytes((int)M, (int)ins[i], (int)Rp) + 255) & ~(size_t)255;
    w->strideV[i] = (cara_tskinny_scratch_bytes((int)M, (int)outs[i], (int)Rp) + 255) & ~(size_t)255;
    // fc2's dU and fc1's dVs may come out of the fc2 dX epilogue, one slab per 160-row tile (cara_gemm_args::er_*)
    const size_t er_bytes = (cara_gemm_epi_rider_scratch_bytes((int)((M + 127) / 128), (int)(4 * D)) + 255) & ~(size_t)255;   // (row tiles of 128 or 160)
    if (i == 3 && epi_riders_geom(g, (int)M, s->wd_exact != 0) && er_bytes > w->strideU[i]) w->strideU[i] = er_bytes;
    if (i == 2 && epi_riders_geom(g, (int)M, s->wd_exact != 0) && er_bytes > w->strideV[i]) w->strideV[i] = er_bytes;
    // (CARA_DV: fc1's / qkv's dVs out of their dX tiles, one slab per 160-row tile and K step: K1 = the linear's out features)
    const size_t dv_bytes = (cara_gemm_epi_rider_scratch_bytes((int)((M + 159) / 160), (int)outs[i]) + 255) & ~(size_t)255;
    if ((i == 0 || i == 2) && env_once("CARA_DV", 0) != 0 && g->Rp == 32 && dv_bytes > w->strideV[i]) w->strideV[i] = dv_bytes;
    w->slabU[i] = c.take(w->strideU[i] * g->depth);
    w->slabV[i] = c.take(w->strideV[i] * g->depth);
  }
  w->ldk = (int)((M + 63) / 64 * 64);
  w->nslab = 1;
  if (s->wd_exact) {
    // dW = dY^T X has only (out/128) x (in/128) = 36..144 output tiles: its K = M is cut into equal slabs (one
    // batched launch fills the chip), which cara_dropout_grad_contract sums
    w->nslab = M >= 8192 ? 4 : (M >= 3072 ? 2 : 1);
    w->ldk = (int)((M + 64 * w->nslab - 1) / (64 * w->nslab) * (64 * w->nslab));
    for (int i = 0; i < 4; ++i) {
      w->weff[i] = c.take((size_t)g->depth * outs[i] * ins[i] * 2);
      w->wefft[i] = c.take((size_t)g->depth * outs[i] * ins[i] * 2);
    }
    w->dYt = c.take((size_t)4 * D * w->ldk * 2);
    w->Xt = c.take((size_t)4 * D * w->ldk * 2);
    size_t dwd = (size_t)w->nslab * 4 * D * D * 4;   // the transposed-copies route: nslab slabs of the largest product
    for (int i = 0; i < 4; ++i) dwd = max_sz(dwd, (size_t)dw_slabs((int)outs[i], (int)ins[i], (int)M) * outs[i] * ins[i] * 4);
    w->dWd = c.take(dwd);
    w->xscratch = c.take(max_sz(cara_dropout_grad_scratch_bytes((int)(4 * D), (int)D, (int)Rp), cara_colsum_scratch_bytes((int)(4 * D))));
  }
  w->dd = w->ddt = w->dD = w->dd_slabs = w->dd_scratch = 0;
  if (g->cp_length == 2) {
    if (s->wd_exact) return false;   // (the dense-delta QKV form and the exact weight-dropout mode are not combined)
    if (g->dim % 128) return false;  // (its backward's dense x^T dY: 128 x 128 output tiles, cara_gemm_tn_f32)
    w->dd = c.take((size_t)g->depth * 3 * D * D * 2);
    w->ddt = c.take((size_t)g->depth * 3 * D * D * 2);
    w->dD = c.take((size_t)g->depth * 3 * D * D * 4);
    w->dd_slabs = c.take((size_t)dw_slabs((int)D, (int)D, (int)M) * D * D * 4);
    w->dd_scratch = c.take(cara_dense_delta_grad_scratch_bytes(g));
  }
  w->total = c.off;
  return true;
}

#define TRY(expr)                 \
  do {                            \
    const int _st = (expr);       \
    if (_st != CARA_OK) return _st; \
  } while (0)

// Optional HIP-event brackets around the kernels of chosen call sites (cara_profile_sites), so that bench.py can
// report average launch durations from INSIDE its timed region, on the stream the kernels run on.  Diagnostic
// state, process-global, off by default; every bracket idles the chip ~15 us (three event records).
struct Prof {
  unsigned long long mask = 0;
  int every = 1;   // bracket the layers l with l % every == 0
  bool made = false;
  long n[CARA_SITE_COUNT] = {};
  hipEvent_t ev[CARA_SITE_COUNT][CARA_SITE_RING][3];
};
Prof g_prof;

// A linear's two transposed skinny products (dU = X^T G', dVs = dY^T T, dc) as arguments of cara_tskinny_partial2 /
// cara_gemm_with_tskinny: the pair waits here until a GEMM launch carries it (its own dX GEMM, or -- CARA_DEFER_TS -- the
// NEXT dX GEMM of the pass, which lets a GEMM compute the G' its own products read) or it is flushed as a launch of its own.
struct TsPending {
  bool valid = false;
  const void *Xa, *Gta, *Xb, *Gtb;
  void *slabs_a, *slabs_b;
  int ldxa, K1a, ldxb, K1b, want_cs, ldg, M, Rp;
  int slot = 0, layer = 0;   // the linear the pair belongs to
};

// How a linear's transposed skinny products left their partial sums, per layer: 0 = one slab per block, 1 = one per wave
// (cara_gemm_rider_slab_format: the launches of the 160 x 256 x 64 tile stream their riding products with helper waves).
// The end-of-pass reduction groups the layers of a linear by it.
struct SlabFormats {
  unsigned short U[4][64] = {}, V[4][64] = {};   // (>= 2: that many slabs per column block, cara_ts_reduce::wave_slabs)
};

// The products the fc2 dX launch computes in its epilogue (cara_gemm_args::er_*): fc1's dVs = dH^T T (+ dc) and fc2's own dU = h^T G'
struct EpiRider {
  const void* Tt;    // fc1's T^T
  void* slabV;       // fc1's dVs slab region of this layer
  size_t bytes;      // size of that region and of fc2's dU region
  bool done = false; // set by the launch that computed them
};

// CARA_FC1_SIDE=1 (default 0): the two heavy riders of the backward -- fc1's dVs = dH^T T (+ dc) and fc2's dU = h^T G', 154 MB -- as a
// launch of their own on a SIDE stream under the fc1 dX GEMM, which then runs on the 160 x 256 x 64 tile (one workgroup per CU: the
// products' blocks, 40 KiB of LDS and one wave per SIMD each, fit beside it).  One event record on the caller's stream per block
// (the fork); the join is a wait on an event that completed long before.  The stream and the two events are created once per process.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
};
SideStream* side_stream() {
  // one stream + event pair per DEVICE, created under a lock on first use there (a process-wide pair made on
  // whichever device was current at first use belongs to the wrong device in a multi-device or multi-threaded process)
  static SideStream per_dev[16];
  static bool tried[16] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  SideStream& s = per_dev[dev];
  if (!tried[dev]) {
    tried[dev] = true;
    s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
  }
  return s.ok ? &s : nullptr;
}
struct SideRiders {   // per backward pass
  SideStream* ss = nullptr;
  bool pending_join = false;   // the side stream holds work the caller's stream has not waited for
};

// per-call context: what lin_fwd / lin_bwd need besides their operands (nothing here outlives the call)
struct Ctx {
  void* stream;
  char* scratch;   // split-K scratch of the workspace in use (few-row products)
  int layer;
  bool full;       // this block runs on all token rows (not the cls-row-only last block)
  TsPending* pend = nullptr;   // backward: the pair of products waiting for a carrier
  int rank = 0;                // the adapter's rank (0: unknown, the skinny passes compute all Rp columns)
  SlabFormats* fmt = nullptr;  // backward: where the launches note the slab format of the products they carry
};

// the rank the transposed skinny products are told (0 in the context = unknown: all Rp columns).  Deferred products
// (CARA_DEFER_TS) ride in adapter-inside GEMMs, whose kernel has no 16-column form: they keep all columns.
bool defer_ts();
int env_once(const char* name, int dflt);
inline int ts_rank(const Ctx& cx, int Rp) {
  static const int full = env_once("CARA_TS_ALL_COLUMNS", 0);   // 1: the products compute all Rp columns whatever the rank (A/B runs)
  return (cx.rank > 0 && !defer_ts() && !full) ? cx.rank : Rp;
}

bool defer_du();
bool g_inside_enabled(int slot);

int flush_pending(const Ctx& cx) {
  TsPending* q = cx.pend;
  if (!q || !q->valid) return CARA_OK;
  q->valid = false;
  if (defer_du())   // (only the dU half of a linear waits in this mode: the first product of the entry)
    return cara_tskinny_partial2_r(nullptr, 0, nullptr, nullptr, 0, q->Xa, q->ldxa, q->Gta, q->slabs_a, q->K1a, 0, q->ldg, q->M, q->Rp,
                                   ts_rank(cx, q->Rp), cx.stream);
  return cara_tskinny_partial2_r(q->Xa, q->ldxa, q->Gta, q->slabs_a, q->K1a, q->Xb, q->ldxb, q->Gtb, q->slabs_b, q->K1b, q->want_cs,
                                 q->ldg, q->M, q->Rp, ts_rank(cx, q->Rp), cx.stream);
}

struct SiteBracket {   // RAII: event 0 .. kernel(s) .. event 1, event 2 (an empty bracket: the markers' own cost)
  hipEvent_t* ev = nullptr;
  hipStream_t st;
  SiteBracket(int site, const Ctx& cx) : st(static_cast<hipStream_t>(cx.stream)) {
    // (block every / 2 of every run of `every` blocks: never block 0, whose qkv dX and LayerNorm-1 backward do not exist)
    if (!(g_prof.mask >> site & 1ull) || !cx.full || cx.layer % g_prof.every != g_prof.every / 2) return;
    ev = g_prof.ev[site][g_prof.n[site]++ % CARA_SITE_RING];
    (void)hipEventRecord(ev[0], st);
  }
  ~SiteBracket() {
    if (!ev) return;
    (void)hipEventRecord(ev[1], st);
    (void)hipEventRecord(ev[2], st);
  }
};

struct Lin {  // one adapted linear of one layer
  const bf16 *W, *Wt;
  const bf16 *Ut, *U, *Vs, *Vst;
  const float* bias;
  int in, out, slot;
  const bf16 *Wp = nullptr, *Wtp = nullptr;   // K-panel-major images of W / Wt (cara_gemm_args::Bp), or null
};

// A/B switches, read ONCE per process (tools/ab_env.sh); the defaults are what DESIGN.md reports.
int env_once(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// CARA_FUSE_XU=0 keeps the K = dim adapter contractions (T = LN(x) U of qkv / fc1, G' = dY Vs of proj / fc2) as
// separate cara_skinny_xu passes instead of fusing them into the LayerNorm kernels
bool fuse_xu(const cara_geom* g) {
  static const int v = env_once("CARA_FUSE_XU", 1);
  return v != 0 && (g->Rp == 32 || g->Rp == 64) && (g->dim == 768 || g->dim == 1024 || g->dim == 256);   // what cara_layernorm_*_xu take
}

// CARA_FUSE_GEMM_T: bit 0 (default on) computes T = X U of forward proj / fc2 inside the GEMM that consumes it
// (cara_gemm_args::Ut) instead of a separate cara_skinny_xu pass; bit 1 (default off: measured slower) the same
// for G' = dY Vs of the backward's qkv / fc1.
bool fuse_gemm_t(int Mr, int Rp, bool backward) {
  static const int v = env_once("CARA_FUSE_GEMM_T", 1);
  return (v & (backward ? 2 : 1)) != 0 && (Rp == 32 || Rp == 64) && Mr >= 1024;
}

// CARA_PANEL_ACTS = mask of the activation groups written K-panel-major ([K/32][M][32], cara_gemm_args::c_panels /
// a_panels) so that the GEMM that consumes them stages whole cache lines: 1 = h / dH (written by GEMM epilogues),
// 2 = xn1 / xn2 (LayerNorm forward), 4 = the dY of fc2 / proj (LayerNorm backward).  Default 5: group 2 costs
// 0.2 ms per step although the GEMMs reading xn1 / xn2 gain 5-8 us each when timed alone.
// Not in the exact-dropout mode, not on the cls-row-only last block.
bool panel_acts(int Mr, const cara_vit_shape* s, int what = 1) {
  static const int v = env_once("CARA_PANEL_ACTS", 5);
  return (v & what) != 0 && !s->wd_exact && Mr >= 1024;
}

// CARA_FUSE_TS=0: the transposed skinny products of a linear run as their own launch behind its dX GEMM instead of
// riding in that GEMM's launch (cara_gemm_with_tskinny).  Rp = 32, full-size products only.
bool fuse_ts(int Mr, int Rp) {
  static const int v = env_once("CARA_FUSE_TS", 1);
  return v != 0 && (Rp == 32 || Rp == 64) && Mr >= 1024;
}

// CARA_DEFER_TS=1 (default off): a linear's transposed skinny products ride in the NEXT dX GEMM launch of the backward pass
// instead of its own (everything they read -- dY, G'^T, the saved input, T^T -- stays untouched until well after that launch:
// the dY / G' buffers of a linear are next written one block later).  That frees a dX GEMM to compute the G' = dY Vs its own
// products read (CARA_FUSE_GEMM_T bit 1: no separate pass over dY), and lets any launch be the carrier of any pair.
// Measured (same box, r03): it LOSES -- 8.95 -> 9.20 ms per step alone, 9.11 with G' inside the dX GEMMs (rank 16); rank 64
// 10.53 -> 10.85 / 10.27.  A pair that rides in its OWN linear's launch reads the dY the GEMM is streaming as its A operand
// at that moment (cache hits); a deferred pair reads 96 MB nothing else touches, and the heaviest pair lands on the
// shortest GEMM (fc1's on proj dX).
bool defer_ts() {
  static const int v = env_once("CARA_DEFER_TS", 0);
  return v != 0;
}

void with_scratch(cara_gemm_args& a, const Ctx& cx) {
  a.scratch = cx.scratch;
  a.scratch_bytes = cx.scratch ? cara_gemm_scratch_bytes() : 0;
}

constexpr int SITE_FWD[4] = {CARA_SITE_QKV_FWD, CARA_SITE_PROJ_FWD, CARA_SITE_FC1_FWD, CARA_SITE_FC2_FWD};
constexpr int SITE_BWD[4] = {CARA_SITE_QKV_BWD, CARA_SITE_PROJ_BWD, CARA_SITE_FC1_BWD, CARA_SITE_FC2_BWD};

// forward of one adapted linear on Mr rows of X (row stride ldx; ldx < 0: K-panel-major, -ldx rows per panel): T = X U ;
// C = [X | T] [W | Vs]^T + bias -> epilogue (a.ldc == 0: dense output)
// (have_T: the LayerNorm that produced X already left T = X U and its transpose, cara_layernorm_fwd_xu)
int lin_fwd(const Lin& L, const bf16* X, int ldx, int Mr, int Rp, int ldt, char* ws, const LayerWs& lw, cara_gemm_args a, const Ctx& cx,
            bool have_T = false) {
  void* st = cx.stream;
  bf16* T = reinterpret_cast<bf16*>(ws + lw.T[L.slot]);
  bf16* Tt = reinterpret_cast<bf16*>(ws + lw.Tt[L.slot]);
  const bool inside = !have_T && fuse_gemm_t(Mr, Rp, false);   // T computed by the GEMM itself
  if (!have_T && !inside) {
    SiteBracket b(CARA_SITE_SKINNY_FWD, cx);
    TRY(cara_skinny_xu_r(X, ldx, L.Ut, T, Tt, ldt, Mr, L.in, Rp, cx.rank > 0 ? cx.rank : Rp, st));
  }
  a.A = X; a.lda = ldx; a.B = L.W; a.Bp = L.Wp; a.ldb = L.in; a.A2 = inside ? nullptr : T; a.B2 = L.Vs; a.Rp = Rp;
  if (ldx < 0) { a.a_panels = -ldx; a.lda = 0; }   // (ldx < 0: X is K-panel-major with -ldx rows per panel, as in cara_skinny_xu)
  if (inside) {
    static const int t_all = env_once("CARA_GEMM_T_ALL_COLUMNS", 0);   // 1: T inside the GEMM with all Rp columns whatever the rank (A/B runs)
    a.Ut = L.Ut; a.T_out = T; a.Tt_out = Tt; a.ldt = ldt; a.Ut_rank = t_all ? 0 : cx.rank;
  }
  a.M = Mr; a.N = L.out; a.K = L.in; a.bias = L.bias;
  if (a.ldc == 0) a.ldc = L.out;
  with_scratch(a, cx);
  SiteBracket b(SITE_FWD[L.slot], cx);
  return cara_gemm_bf16(&a, st);
}

// backward of one adapted linear given dY (bf16, Mr rows, row stride lddy) and its saved input X
// (row stride ldx; a negative stride = K-panel-major with that many rows per panel):
//   G' = dY Vs ; dX = [dY | G'] [W^T | U]^T (optional) ; dU = X^T G' ; dVs = dY^T T ; dc = colsum dY
// The two transposed skinny products ride in the launch of the dX GEMM when they can (cara_gemm_with_tskinny),
// otherwise they are one launch of their own behind it, on the same stream.
int lin_bwd(const Lin& L, const bf16* dY, int lddy, const bf16* X, int ldx, int Mr, int Rp, int ldt, char* ws, const Ws& W,
            const LayerWs& lw, bool want_dx, cara_gemm_args a, bool want_dc, const Ctx& cx, bool have_G = false, EpiRider* er = nullptr,
            bool dvs_done = false, SideRiders* side = nullptr, int dv = 0) {
  // dv: 1 = this linear's dVs (+ dc) out of its dX GEMM's own A tiles where the launch can (cara_gemm_dv_chunks); 2 = this linear carries
  // BOTH its products in its own launch and leaves the waiting dU for the next one (the linear before a dv = 1 one: CARA_DV)
  void* st = cx.stream;
  const Ws::Bwd& R = W.bwd;
  bf16* G = reinterpret_cast<bf16*>(ws + R.G[L.slot]);
  bf16* Gt = reinterpret_cast<bf16*>(ws + R.Gt[L.slot]);
  void* slabU = ws + W.slabU[L.slot] + (size_t)cx.layer * W.strideU[L.slot];
  void* slabV = ws + W.slabV[L.slot] + (size_t)cx.layer * W.strideV[L.slot];
  const void* Tt = ws + lw.Tt[L.slot];
  // (have_G: the LayerNorm backward that produced dY already left G' and its transpose, cara_layernorm_bwd_xu)
  // inside: the dX GEMM computes G' = dY Vs itself (cara_gemm_args::Ut) and leaves G / Gt behind -- its own products then
  // cannot ride in it (they read that G'): only with deferred products (a later launch carries them)
  TsPending* pend = cx.pend;
  const bool can_carry = fuse_ts(Mr, Rp);
  const bool defer = pend && defer_ts() && can_carry;
  const bool inside = !have_G && want_dx && fuse_gemm_t(Mr, Rp, true) && defer && a.epi == CARA_EPI_BF16;
  // with the dU products riding one launch later (CARA_DEFER_DU) nothing in a linear's own dX launch reads its G', so the GEMM
  // can compute it inside and the separate pass over dY goes (CARA_GEMM_G_INSIDE=0 keeps the pass)
  const bool g_inside = !inside && !have_G && want_dx && can_carry && defer_du() && pend && g_inside_enabled(L.slot) &&
                        (Rp == 32 || Rp == 64) && Mr >= 1024 && a.epi == CARA_EPI_BF16;   // (its own switch: CARA_GEMM_G_INSIDE)
  if (!have_G && !inside && !g_inside) {
    SiteBracket b(CARA_SITE_SKINNY_BWD, cx);
    TRY(cara_skinny_xu_r(dY, lddy, L.Vst, G, Gt, ldt, Mr, L.out, Rp, cx.rank > 0 ? cx.rank : Rp, st));
  }
  TsPending mine;
  mine.valid = true;
  mine.Xa = X; mine.ldxa = ldx; mine.Gta = Gt; mine.slabs_a = slabU; mine.K1a = L.in;
  mine.Xb = dY; mine.ldxb = lddy; mine.Gtb = Tt; mine.slabs_b = slabV; mine.K1b = L.out; mine.want_cs = want_dc ? 1 : 0;
  mine.ldg = ldt; mine.M = Mr; mine.Rp = Rp; mine.slot = L.slot; mine.layer = cx.layer;
  if (want_dx) {
    a.A = dY; a.lda = lddy; a.B = L.Wt; a.Bp = L.Wtp; a.ldb = L.out; a.A2 = inside ? nullptr : G; a.B2 = L.U; a.Rp = Rp;
    if (lddy < 0) { a.a_panels = -lddy; a.lda = 0; }
    if (inside) { a.Ut = L.Vst; a.T_out = G; a.Tt_out = Gt; a.ldt = ldt; }
    a.M = Mr; a.N = L.in; a.K = L.out; a.bias = nullptr;
    if (a.ldc == 0) a.ldc = L.in;
    with_scratch(a, cx);
    SiteBracket b(SITE_BWD[L.slot], cx);
    // (CARA_DEFER_DU, default on) this launch carries its linear's dVs = dY^T T (which reads the dY the GEMM is streaming) and the dU = X^T G'
    // of the PREVIOUS linear of the pass (a dU never shares an operand with its own dX GEMM, so it loses nothing by riding
    // elsewhere): fc2's dU, the 77-MB read of h, leaves the two-round fc2 dX launch for the single-round fc1 dX launch
    if (can_carry && defer_du() && pend && !inside) {
      bool take = pend->valid && pend->Rp == Rp && pend->M == Mr && pend->ldg == ldt;
      if (pend->valid && (!take || dvs_done)) {
        TRY(flush_pending(cx));
        take = false;
      }
      if (g_inside) {   // G' = dY Vs computed by this GEMM on the tiles it streams (its own dVs does not read G'; its dU rides later)
        a.A2 = nullptr; a.Ut = L.Vst; a.T_out = G; a.Tt_out = Gt; a.ldt = ldt; a.Ut_rank = ts_rank(cx, Rp) <= 16 ? ts_rank(cx, Rp) : 0;
      }
      if (dv == 2 && !dvs_done && !er) {
        if (cx.fmt) {
          const unsigned short f = (unsigned short)cara_gemm_rider_slab_format(&a, Rp, ts_rank(cx, Rp));
          cx.fmt->U[L.slot][cx.layer] = f;
          cx.fmt->V[L.slot][cx.layer] = f;
        }
        TRY(cara_gemm_with_tskinny_r(&a, mine.Xa, mine.ldxa, mine.Gta, mine.slabs_a, mine.K1a, mine.Xb, mine.ldxb, mine.Gtb, mine.slabs_b, mine.K1b,
                                     mine.want_cs, ldt, Mr, Rp, ts_rank(cx, Rp), st));
        return CARA_OK;                // (the waiting dU, if any, stays for the next launch)
      }
      if (dv == 1 && !dvs_done && !er) {
        cara_gemm_args d = a;
        d.er_Tt = Tt; d.er_ldg = ldt; d.er_slabs_v = slabV; d.er_colsum = want_dc ? 1 : 0;
        const int ch = cara_gemm_dv_chunks(&d, take ? 1 : 0);
        if (ch > 0 && ch <= 65535 && cara_gemm_epi_rider_scratch_bytes(ch, L.out) <= W.strideV[L.slot]) {
          if (take) {   // the waiting dU rides behind the tiles (the launch's only product: slot b, no column sums)
            if (cx.fmt) cx.fmt->U[pend->slot][pend->layer] = 0;
            TRY(cara_gemm_with_tskinny_r(&d, nullptr, 0, nullptr, nullptr, 0, pend->Xa, pend->ldxa, pend->Gta, pend->slabs_a, pend->K1a, 0, ldt, Mr, Rp,
                                         ts_rank(cx, Rp), st));
          } else {
            TRY(cara_gemm_bf16(&d, st));
          }
          if (cx.fmt) cx.fmt->V[L.slot][cx.layer] = (unsigned short)ch;
          *pend = mine;   // (its dU half waits for the next dX GEMM of the pass)
          return CARA_OK;
        }
      }
      if (side && side->ss && !dvs_done) {
        // this linear's dVs (+ dc) and the waiting dU as a launch of their own on the side stream, under this GEMM
        SideStream* ss = side->ss;
        if (hipEventRecord(ss->fork, static_cast<hipStream_t>(st)) != hipSuccess || hipStreamWaitEvent(ss->stream, ss->fork, 0) != hipSuccess) return CARA_E_LAUNCH;
        TRY(cara_tskinny_partial2_small(take ? pend->Xa : nullptr, take ? pend->ldxa : 0, take ? pend->Gta : nullptr, take ? pend->slabs_a : nullptr,
                                        take ? pend->K1a : 0, mine.Xb, mine.ldxb, mine.Gtb, mine.slabs_b, mine.K1b, mine.want_cs, ldt, Mr, Rp,
                                        ts_rank(cx, Rp), ss->stream));
        if (hipEventRecord(ss->join, ss->stream) != hipSuccess) return CARA_E_LAUNCH;
        side->pending_join = true;
        if (cx.fmt) {
          if (take) cx.fmt->U[pend->slot][pend->layer] = 0;
          cx.fmt->V[L.slot][cx.layer] = 0;
        }
        TRY(cara_gemm_bf16(&a, st));
        *pend = mine;   // (its dU half waits for the next dX GEMM of the pass)
        return CARA_OK;
      }
      if (dvs_done) {   // dVs and dc of this linear came out of the epilogue of the launch that produced its dY: a plain GEMM, its dU waits
        TRY(cara_gemm_bf16(&a, st));
        *pend = mine;
        return CARA_OK;
      }
      int er_chunks = 0;
      if (er) {   // this launch's epilogue computes the next linear's dVs (+ dc) and this linear's dU
        a.er_Tt = er->Tt; a.er_Gt = Gt; a.er_slabs_v = er->slabV; a.er_slabs_u = slabU; a.er_ldg = ldt; a.er_colsum = 1;
        a.er_h = X; a.er_h_panels = ldx < 0 ? -ldx : 0;   // (this linear's saved input: h)
        er_chunks = cara_gemm_epi_rider_chunks(&a);
        if (!er_chunks || er_chunks > 65535 || cara_gemm_epi_rider_scratch_bytes(er_chunks, L.in) > er->bytes) {
          er_chunks = 0;
          a.er_Tt = a.er_Gt = a.er_h = nullptr; a.er_slabs_v = a.er_slabs_u = nullptr;
        }
      }
      if (cx.fmt) {   // (depends on the arguments only: asked before the launch, valid for it)
        const unsigned short f = (unsigned short)cara_gemm_rider_slab_format(&a, Rp, ts_rank(cx, Rp));
        if (take) cx.fmt->U[pend->slot][pend->layer] = f;
        cx.fmt->V[L.slot][cx.layer] = f;
      }
#ifdef CARA_ABLATE_DVS   // timing experiment only (tools/build_variant.sh): fc1 / qkv dX WITHOUT their own dVs products (wrong gradients) --
      // the bound on what computing dVs from the dX tile's own A sub-buffers could return
      if ((L.slot == 0 || L.slot == 2) && g_inside) {
        if (take) TRY(cara_gemm_with_tskinny_r(&a, nullptr, 0, nullptr, nullptr, 0, pend->Xa, pend->ldxa, pend->Gta, pend->slabs_a, pend->K1a, 0, ldt, Mr, Rp,
                                               ts_rank(cx, Rp), st));
        else TRY(cara_gemm_bf16(&a, st));
        *pend = mine;
        return CARA_OK;
      }
#endif
      TRY(cara_gemm_with_tskinny_r(&a, take ? pend->Xa : nullptr, take ? pend->ldxa : 0, take ? pend->Gta : nullptr, take ? pend->slabs_a : nullptr,
                                   take ? pend->K1a : 0, mine.Xb, mine.ldxb, mine.Gtb, mine.slabs_b, mine.K1b, mine.want_cs, ldt, Mr, Rp,
                                   ts_rank(cx, Rp), st));
      if (er_chunks) {   // (nothing of this linear waits: its dU is in the slabs the epilogue wrote)
        pend->valid = false;
        er->done = true;
        if (cx.fmt) cx.fmt->U[L.slot][cx.layer] = (unsigned short)er_chunks;
        return CARA_OK;
      }
      *pend = mine;   // (its dU half waits for the next dX GEMM of the pass; flush_pending runs it alone otherwise)
      return CARA_OK;
    }
    // which pair this launch carries: the one that waits (deferred), else its own
    const TsPending* carry = nullptr;
    if (can_carry) {
      if (defer) carry = (pend->valid && pend->Rp == Rp) ? pend : nullptr;
      else if (!inside) carry = &mine;
    }
#ifdef CARA_ABLATE_TS   // timing experiment only (tools/build_variant.sh): the dX GEMMs WITHOUT their riding products (wrong gradients)
    if (carry) { TRY(cara_gemm_bf16(&a, st)); return CARA_OK; }
#endif
    if (carry && cx.fmt) {
      const unsigned short f = (unsigned short)cara_gemm_rider_slab_format(&a, carry->Rp, ts_rank(cx, carry->Rp));
      cx.fmt->U[carry->slot][carry->layer] = f;
      cx.fmt->V[carry->slot][carry->layer] = f;
    }
    if (carry) {
      TRY(cara_gemm_with_tskinny_r(&a, carry->Xa, carry->ldxa, carry->Gta, carry->slabs_a, carry->K1a, carry->Xb, carry->ldxb, carry->Gtb,
                                   carry->slabs_b, carry->K1b, carry->want_cs, carry->ldg, carry->M, carry->Rp, ts_rank(cx, carry->Rp), st));
      if (carry == pend) pend->valid = false;
      if (carry == &mine) return CARA_OK;
    } else {
      TRY(cara_gemm_bf16(&a, st));
    }
  }
  if (defer) {
    TRY(flush_pending(cx));   // (a pair that found no carrier -- none waits in the steady state)
    *pend = mine;
    return CARA_OK;
  }
  if (pend) TRY(flush_pending(cx));
  return cara_tskinny_partial2_r(X, ldx, Gt, slabU, L.in, dY, lddy, Tt, slabV, L.out, want_dc ? 1 : 0, ldt, Mr, Rp, ts_rank(cx, Rp), st);
}

// ---- order-2 QKV tensorisation: the QKV linear as y = x W^T + x Dm^T with the dense scaled delta Dm (cara_dense_delta_*) ----
int lin_fwd_dense(const Lin& L, const bf16* X, int ldx, int Mr, const bf16* Dm, cara_gemm_args a, const Ctx& cx) {
  a.A = X; a.lda = ldx; a.B = L.W; a.B3 = Dm; a.ldb = L.in; a.A2 = nullptr; a.B2 = nullptr; a.Rp = 0;
  a.M = Mr; a.N = L.out; a.K = L.in; a.bias = L.bias;
  if (a.ldc == 0) a.ldc = L.out;
  with_scratch(a, cx);
  SiteBracket b(SITE_FWD[L.slot], cx);
  return cara_gemm_bf16(&a, cx.stream);
}
// backward: dX = dY W + dY Dm (optional), and dD[k] = X^T dY_k for the three projections (split-K slabs, summed)
int lin_bwd_dense(const Lin& L, const bf16* dY, const bf16* X, int Mr, const bf16* Dmt, float* dD_l, float* slabs, bool want_dx,
                  cara_gemm_args a, const Ctx& cx) {
  void* st = cx.stream;
  const int D = L.in;
  if (want_dx) {
    a.A = dY; a.lda = L.out; a.B = L.Wt; a.B3 = Dmt; a.ldb = L.out; a.A2 = nullptr; a.B2 = nullptr; a.Rp = 0;
    a.M = Mr; a.N = L.in; a.K = L.out; a.bias = nullptr;
    if (a.ldc == 0) a.ldc = L.in;
    with_scratch(a, cx);
    SiteBracket b(SITE_BWD[L.slot], cx);
    TRY(cara_gemm_bf16(&a, st));
  }
  const int ns = dw_slabs(D, D, Mr);
  const size_t dd = (size_t)D * D;
  for (int k = 0; k < 3; ++k) {
    TRY(cara_gemm_tn_f32(X, D, dY + (size_t)k * D, L.out, slabs, D, D, D, Mr, ns, dd, st));
    TRY(cara_sum_slabs_f32(slabs, ns, dd, dd, dD_l + (size_t)k * dd, st));
  }
  return CARA_OK;
}

// ---- exact weight-dropout mode (cara_vit_shape::wd_exact): y = x W^T + x (keep/(1-p) dW)^T, two accumulated products ----
// forward of one linear: materialise the masked delta (and its transpose, for dX) of this layer, then C = X (W + Dm)^T + bias
int lin_fwd_exact(const Lin& L, const bf16* X, int ldx, int Mr, int Rp, char* ws, const Ws& W, const cara_vit_shape* s,
                  cara_gemm_args a, const Ctx& cx) {
  void* st = cx.stream;
  const int layer = cx.layer;
  const size_t wbytes = (size_t)L.out * L.in * 2;
  bf16* weff = reinterpret_cast<bf16*>(ws + W.weff[L.slot] + layer * wbytes);
  bf16* wefft = reinterpret_cast<bf16*>(ws + W.wefft[L.slot] + layer * wbytes);
  // the masked adapter delta on its own (W == NULL), kept apart from the frozen weight: y = x W^T + x Dm^T as two
  // products accumulated in fp32 (cara_gemm_args::B3) -- merged into one bf16 weight, a delta below half an ulp of W
  // would vanish (the reference's zero-initialised A2 / P2: the whole adapter early in training)
  TRY(cara_materialize_merge(nullptr, L.U, L.Vs, Rp, L.out, L.in, s->wd_p, s->wd_seed, (unsigned)(4 * layer + L.slot), weff, st));
  TRY(cara_transpose_bf16_ld(weff, L.in, wefft, L.out, L.out, L.in, st));
  a.A = X; a.lda = ldx; a.B = L.W; a.B3 = weff; a.ldb = L.in; a.A2 = nullptr; a.B2 = nullptr; a.Rp = 0;
  a.M = Mr; a.N = L.out; a.K = L.in; a.bias = L.bias;
  if (a.ldc == 0) a.ldc = L.out;
  with_scratch(a, cx);
  return cara_gemm_bf16(&a, st);
}

// backward of one linear: dc = colsum dY; dW = dY^T X (dense, fp32) -> dU, dVs through the regenerated mask;
// dX = dY W_eff (optional).  X and dY are dense [Mr, in] / [Mr, out] (the cls-row shortcut is off in this mode).
int lin_bwd_exact(const Lin& L, const bf16* dY, const bf16* X, int Mr, int Rp, char* ws, const Ws& W,
                  const cara_vit_shape* s, bool want_dx, cara_gemm_args a, bool want_dc, const Ctx& cx) {
  void* st = cx.stream;
  const int layer = cx.layer;
  hipStream_t hs = static_cast<hipStream_t>(st);
  const size_t wbytes = (size_t)L.out * L.in * 2;
  bf16* wefft = reinterpret_cast<bf16*>(ws + W.wefft[L.slot] + layer * wbytes);
  bf16* dYt = reinterpret_cast<bf16*>(ws + W.dYt);
  bf16* Xt = reinterpret_cast<bf16*>(ws + W.Xt);
  float* dWd = reinterpret_cast<float*>(ws + W.dWd);
  const int ldk = W.ldk;
  if (want_dc)
    TRY(cara_colsum_bf16(dY, L.out, Mr, L.out, reinterpret_cast<float*>(ws + W.dc[L.slot]) + (size_t)layer * L.out, ws + W.xscratch, st));
  // dW = dY^T X as split-K slabs.  Straight from the row-major activations through the transposing-read GEMM when the shapes
  // allow (CARA_EXACT_TN=0: the first form -- two activation-sized transposes with zeroed pad columns, then one batched
  // launch of the default GEMM over the transposed copies)
  const size_t slab_stride = (size_t)L.out * L.in;
  int used = dw_slabs(L.out, L.in, Mr);
  static const int use_tn = env_once("CARA_EXACT_TN", 1);
  if (!(use_tn && cara_gemm_tn_f32(dY, L.out, X, L.in, dWd, L.in, L.out, L.in, Mr, used, slab_stride, st) == CARA_OK)) {
    used = W.nslab;
    if (ldk > Mr) {   // K of the dW product is Mr rounded up to 64: the pad columns of both transposes must be zero
      if (hipMemset2DAsync(dYt + Mr, (size_t)ldk * 2, 0, (size_t)(ldk - Mr) * 2, L.out, hs) != hipSuccess) return CARA_E_LAUNCH;
      if (hipMemset2DAsync(Xt + Mr, (size_t)ldk * 2, 0, (size_t)(ldk - Mr) * 2, L.in, hs) != hipSuccess) return CARA_E_LAUNCH;
    }
    TRY(cara_transpose_bf16_ld(dY, L.out, dYt, ldk, Mr, L.out, st));
    TRY(cara_transpose_bf16_ld(X, L.in, Xt, ldk, Mr, L.in, st));
    // split-K in one batched launch: slab z covers K columns [z * ldk/nslab, ...) of both transposes
    cara_gemm_args d = {};
    d.A = dYt; d.lda = ldk; d.B = Xt; d.ldb = ldk; d.M = L.out; d.N = L.in; d.K = ldk / used;
    d.epi = CARA_EPI_F32; d.C = dWd; d.ldc = L.in;
    d.batch = used; d.strideA = d.K; d.strideB = d.K; d.strideC = (long long)slab_stride;
    TRY(cara_gemm_bf16(&d, st));
  }
  TRY(cara_dropout_grad_contract(dWd, used, slab_stride, L.U, L.Vs, Rp, L.out, L.in, s->wd_p, s->wd_seed, (unsigned)(4 * layer + L.slot),
                                 reinterpret_cast<float*>(ws + W.dU[L.slot]) + (size_t)layer * L.in * Rp,
                                 reinterpret_cast<float*>(ws + W.dVs[L.slot]) + (size_t)layer * L.out * Rp, ws + W.xscratch, st));
  if (want_dx) {
    a.A = dY; a.lda = L.out; a.B = L.Wt; a.B3 = wefft; a.ldb = L.out; a.A2 = nullptr; a.B2 = nullptr; a.Rp = 0;
    a.M = Mr; a.N = L.in; a.K = L.out; a.bias = nullptr;
    if (a.ldc == 0) a.ldc = L.in;
    with_scratch(a, cx);
    TRY(cara_gemm_bf16(&a, st));
  }
  return CARA_OK;
}

void make_lins(const cara_geom* g, const cara_vit_weights* w, const char* pack, const cara_pack_layout& pl, int l, Lin* out) {
  const size_t D = g->dim;
  const char* pk = pack + (size_t)l * pl.layer_stride;
  auto B = [](const void* p, size_t elems) { return static_cast<const bf16*>(p) + elems; };
  auto P = [&](size_t off) { return reinterpret_cast<const bf16*>(pk + off); };
  out[0] = Lin{B(w->qkv_w, l * 3 * D * D), B(w->qkv_wt, l * 3 * D * D), P(pl.Ut_qkv), P(pl.U_qkv), P(pl.Vs_qkv), P(pl.Vst_qkv),
               w->qkv_b + (size_t)l * 3 * D, (int)D, (int)(3 * D), 0};
  out[1] = Lin{B(w->proj_w, l * D * D), B(w->proj_wt, l * D * D), P(pl.Ut_proj), P(pl.U_proj), P(pl.Vs_proj), P(pl.Vst_proj),
               reinterpret_cast<const float*>(pk + pl.bias_proj), (int)D, (int)D, 1};
  out[2] = Lin{B(w->fc1_w, l * 4 * D * D), B(w->fc1_wt, l * 4 * D * D), P(pl.Ut_fc1), P(pl.U_fc1), P(pl.Vs_fc1), P(pl.Vst_fc1),
               reinterpret_cast<const float*>(pk + pl.bias_fc1), (int)D, (int)(4 * D), 2};
  out[3] = Lin{B(w->fc2_w, l * 4 * D * D), B(w->fc2_wt, l * 4 * D * D), P(pl.Ut_fc2), P(pl.U_fc2), P(pl.Vs_fc2), P(pl.Vst_fc2),
               reinterpret_cast<const float*>(pk + pl.bias_fc2), (int)(4 * D), (int)D, 3};
  // CARA_GEMM_PACKED=0: stage the weights from their row-major images (A/B measurements)
  static const bool use_packed = env_once("CARA_GEMM_PACKED", 1) != 0;
  if (use_packed) {
    const void* wp[4][2] = {{w->qkv_wp, w->qkv_wtp}, {w->proj_wp, w->proj_wtp}, {w->fc1_wp, w->fc1_wtp}, {w->fc2_wp, w->fc2_wtp}};
    for (int i = 0; i < 4; ++i) {
      const size_t elems = (size_t)l * out[i].in * out[i].out;
      if (wp[i][0]) out[i].Wp = B(wp[i][0], elems);
      if (wp[i][1]) out[i].Wtp = B(wp[i][1], elems);
    }
  }
}

// CARA_CLS_SHORTCUT=0 turns the exact last-block shortcut off (A/B measurements only)
bool cls_shortcut_enabled() {
  static const int v = env_once("CARA_CLS_SHORTCUT", 1);
  return v != 0;
}

// CARA_CLS_ATTN=0: the last block runs the full attention kernels although only its cls query matters (A/B measurements only)
bool cls_attention_enabled() {
  static const int v = env_once("CARA_CLS_ATTN", 1);
  return v != 0;
}

bool g_inside_enabled(int slot) {   // CARA_GEMM_G_INSIDE: bit 0 qkv, bit 1 fc1 (default 3: both)
  static const int v = env_once("CARA_GEMM_G_INSIDE", 3);
  return (v & (slot == 0 ? 1 : 2)) != 0;
}

// A linear's dU = X^T G' rides in the NEXT dX GEMM of the pass, its dVs in its own (lin_bwd); CARA_DEFER_DU=0: both in its own
bool defer_du() {
  static const int v = env_once("CARA_DEFER_DU", 1);
  return v != 0 && !defer_ts();
}

// CARA_SAVE_GELU_GRAD (default: on where CARA_EPI_RIDERS is): fc1 forward keeps gelu'(u) (IEEE half) instead of the pre-activation u, and
// the fc2 dX epilogue is one multiply per element instead of the erf arithmetic (CARA_EPI_GELU_DG / CARA_EPI_MULH, include/cara_hip.h).
// Where the LayerNorms compute T = X U (the forward's fc1 GEMM then never has the adapter inside, a form the new epilogue does not
// take), factored mode.  Measured alone (r04, same box): a tie -- fc1 forward 77.3 -> 80.0 us, fc2 dX 76.9 -> 74.3: neither epilogue is
// bound by its VALU count any more (docs/findings/r04.md section 6).
static int epi_riders_env() {
  static const int v = env_once("CARA_EPI_RIDERS", 0);
  return v;
}
bool save_gelu_grad(const cara_geom* g, const cara_vit_shape* s) {
  static const int v = env_once("CARA_SAVE_GELU_GRAD", -1);
  return (v < 0 ? epi_riders_env() != 0 : v != 0) && fuse_xu(g) && !s->wd_exact;
}

static bool env_once_dv_off() {   // (the epilogue riders and CARA_DV are two placements of the same product)
  static const int v = env_once("CARA_DV", 0);
  return v == 0;
}

// CARA_EPI_RIDERS (default 0): fc1's dVs = dH^T T (+ dc) and fc2's dU = h^T G' are computed by the epilogue of the fc2 dX GEMM, on the dH
// tile it has just produced and the h tile at the same coordinates (cara_gemm_args::er_*), instead of as workgroups riding in the fc1 dX
// launch that re-read dH and h (154 MB per block at the headline shape); the fc1 dX launch then carries nothing and runs on the
// 160 x 256 x 64 tile.  Rank <= 16, full-size blocks; what the shape allows is asked of cara_gemm_epi_rider_chunks() at the launch.
// Measured (r04, same box, profiles/r04_n_*): fc1 dX 86.4 -> 56.2 us, fc2 dX 74.3 -> 102.8 us, and the end-of-pass slab sums read
// 99 slabs per product instead of 6 (+80 us per step): 8.005 -> 8.096 ms per step.  The riders' epilogue is a longer DEPENDENT chain
// per wave (two operand streams, an LDS round trip more per row tile) in a launch that is bound by such chains, not by any unit
// (MFMA-busy 0.25, LDS 24 %, +105 MB of reads at 3.5 TB/s): off.
bool epi_riders_geom(const cara_geom* g, int Mr, bool exact) {
  const int v = epi_riders_env();
  cara_vit_shape sh = {};
  sh.wd_exact = exact ? 1 : 0;
  return v != 0 && !exact && save_gelu_grad(g, &sh) && g->Rp == 32 && g->rank > 0 && g->rank <= 16 && Mr > 1024 && (Mr & 3) == 0 && 4 * g->dim >= 3072 && ((4 * g->dim) & 127) == 0;
}

// tiny classifier-head backward (B x classes x D, fp32 VALU).  The three outputs are independent: blocks
// [0, nbw) take dW (and db), the rest dxn, so the two long loops run side by side instead of one after the other in
// every thread (58 -> ~25 us; it sits alone at the head of the backward pass)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl, const bf16* __restrict__ xn,
                                                       const float* __restrict__ W, float* __restrict__ dW,
                                                       float* __restrict__ db, bf16* __restrict__ dxn, int B, int Cn, int D, int nbw,
                                                       const float* __restrict__ loss_scale, float* __restrict__ found_inf) {
  if ((int)blockIdx.x < nbw) {
    // the two gradients this kernel FINISHES leave unscaled (cara_vit_shape::loss_scale); dxn stays in the scaled domain
    const float gs = loss_scale ? 1.f / *loss_scale : 1.f;
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (e < Cn * D) {
      const int c = e / D, d = e - c * D;
      float s = 0.f;
#pragma unroll 8
      for (int b = 0; b < B; ++b) s += dl[b * Cn + c] * (float)xn[b * D + d];
      s *= gs;
      bad |= !(fabsf(s) <= 3.0e38f);
      dW[e] = s;
    }
    if (e < Cn) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dl[b * Cn + e];
      s *= gs;
      bad |= !(fabsf(s) <= 3.0e38f);
      db[e] = s;
    }
    if (found_inf && bad) *found_inf = 1.f;
  } else {
    const int e = (blockIdx.x - nbw) * 256 + threadIdx.x;
    if (e < B * D) {
      const int b = e / D, d = e - b * D;
      float s = 0.f;
#pragma unroll 10
      for (int c = 0; c < Cn; ++c) s += dl[b * Cn + c] * W[c * D + d];
      dxn[e] = (bf16)s;
    }
  }
}

// Final LayerNorm of one cls row + the classifier head in fp32 (cara_head_forward): workgroup (class chunk of HEAD_CHUNK, sample).
// Every workgroup normalises its sample's row itself (D floats: cheaper than a launch boundary); chunk 0 also writes the
// 16-bit xn row and mean / rstd for the backward.  The dot products run one class per wave and step over the row in float4
// pieces (a wave instruction reads 1 KiB of a head_w row), two classes in flight per wave.  A chunk is 16 classes = two rounds of
// the four waves: with 128 (64 workgroups for 64 x 100 logits, 13 dependent rounds each) the launch took 29 us (profiles/r05_g_*).
constexpr int HEAD_CHUNK = 16;
__global__ __launch_bounds__(256) void head_fwd_f32_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ Wh,
                                                           const float* __restrict__ bh, bf16* __restrict__ xn16,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ logits, int Cn, int D, float eps) {
  __shared__ __attribute__((aligned(16))) float xs[1024];
  __shared__ float red[8];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)b * ldx;
  float s = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = xr[d]; xs[d] = v; s += v; }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mu = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
  float q = 0.f;
  for (int d = tid; d < D; d += 256) { const float a = xs[d] - mu; q += a * a; }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rs = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)D + eps);
  for (int d = tid; d < D; d += 256) {
    const float v = (xs[d] - mu) * rs * gamma[d] + beta[d];
    xs[d] = v;
    if (blockIdx.x == 0) xn16[(size_t)b * D + d] = (bf16)v;
  }
  if (blockIdx.x == 0 && tid == 0) { mean[b] = mu; rstd[b] = rs; }
  __syncthreads();
  const int c0 = blockIdx.x * HEAD_CHUNK, c1 = c0 + HEAD_CHUNK < Cn ? c0 + HEAD_CHUNK : Cn;
  const int nd4 = D >> 2;   // float4 pieces of a row (D % 4 == 0)
  for (int c = c0 + wave * 2; c < c1; c += 8) {
    const int ca = c, cb = c + 1 < c1 ? c + 1 : c;
    const float4* wa = reinterpret_cast<const float4*>(Wh + (size_t)ca * D);
    const float4* wb = reinterpret_cast<const float4*>(Wh + (size_t)cb * D);
    float da = 0.f, db = 0.f;
    for (int i = lane; i < nd4; i += 64) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * i);
      const float4 a = wa[i], bq = wb[i];
      da += (a.x * xv.x + a.y * xv.y) + (a.z * xv.z + a.w * xv.w);
      db += (bq.x * xv.x + bq.y * xv.y) + (bq.z * xv.z + bq.w * xv.w);
    }
    da = wave_sum(da);
    db = wave_sum(db);
    if (lane == 0) {
      logits[(size_t)b * Cn + ca] = da + bh[ca];
      if (cb != ca) logits[(size_t)b * Cn + cb] = db + bh[cb];
    }
  }
}

}  // namespace

extern "C" int cara_head_forward(const float* x, long ldx, const float* gamma, const float* beta, const float* head_w,
                                 const float* head_b, void* xn_16, float* mean, float* rstd, float* logits, int B, int classes,
                                 int D, float eps, void* stream) {
  if (!x || !gamma || !beta || !head_w || !head_b || !xn_16 || !mean || !rstd || !logits || B <= 0 || classes <= 0 || D <= 0 ||
      D > 1024 || (D & 3) || ldx < D)
    return CARA_E_ARG;
  hipLaunchKernelGGL(head_fwd_f32_kernel, dim3((classes + HEAD_CHUNK - 1) / HEAD_CHUNK, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ldx, gamma, beta, head_w, head_b, (bf16*)xn_16, mean, rstd, logits, classes, D, eps);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}

extern "C" int cara_profile_sites(unsigned long long mask, int every) {
  if (mask && !g_prof.made) {
    for (int i = 0; i < CARA_SITE_COUNT; ++i)
      for (int k = 0; k < CARA_SITE_RING; ++k)
        for (int j = 0; j < 3; ++j)
          if (hipEventCreate(&g_prof.ev[i][k][j]) != hipSuccess) return CARA_E_LAUNCH;
    g_prof.made = true;
  }
  g_prof.mask = mask & ((1ull << CARA_SITE_COUNT) - 1);
  g_prof.every = every > 0 ? every : 1;
  for (int i = 0; i < CARA_SITE_COUNT; ++i) g_prof.n[i] = 0;
  return CARA_OK;
}

// avg_ms = mean (event 0 -> event 1) around the site's kernel(s) MINUS marker_ms = mean (event 1 -> event 2) with
// nothing between: two event records in a row are ~6-8 us apart on this stack, and that gap is inside every bracket
extern "C" int cara_profile_site_read(int site, float* avg_ms, float* marker_ms, int* launches) {
  if (site < 0 || site >= CARA_SITE_COUNT || !avg_ms || !marker_ms || !launches || !g_prof.made) return CARA_E_ARG;
  const long n = g_prof.n[site];
  *launches = 0;
  if (n == 0) return CARA_OK;
  double tot = 0, gap = 0;
  const int cnt = n < CARA_SITE_RING ? (int)n : CARA_SITE_RING;
  for (int i = 0; i < cnt; ++i) {
    float ms = 0.f, g = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[site][i][0], g_prof.ev[site][i][1]) != hipSuccess) return CARA_E_LAUNCH;
    if (hipEventElapsedTime(&g, g_prof.ev[site][i][1], g_prof.ev[site][i][2]) != hipSuccess) return CARA_E_LAUNCH;
    tot += ms;
    gap += g;
  }
  *marker_ms = (float)(gap / cnt);
  *avg_ms = (float)((tot - gap) / cnt);
  *launches = (int)n;   // brackets recorded since cara_profile_sites(); the averages cover the last min(n, CARA_SITE_RING)
  return CARA_OK;
}

extern "C" size_t cara_vit_workspace_bytes(const cara_geom* g, const cara_vit_shape* s) {
  Ws w;
  return layout(g, s, &w) ? w.total : 0;
}

static int head_backward_scaled(const float* dlogits, const void* xn, const float* head_w, float* dhead_w, float* dhead_b, void* dxn,
                                int B, int classes, int D, const float* loss_scale, float* found_inf, void* stream) {
  if (!dlogits || !xn || !head_w || !dhead_w || !dhead_b || !dxn || B <= 0 || classes <= 0 || D <= 0) return CARA_E_ARG;
  const int nbw = (classes * D + 255) / 256, nbx = (B * D + 255) / 256;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(nbw + nbx), dim3(256), 0, static_cast<hipStream_t>(stream), dlogits,
                     (const bf16*)xn, head_w, dhead_w, dhead_b, (bf16*)dxn, B, classes, D, nbw, loss_scale, found_inf);
  CARA_CHECK_LAUNCH();
  return CARA_OK;
}
extern "C" int cara_head_backward(const float* dlogits, const void* xn, const float* head_w, float* dhead_w,
                                  float* dhead_b, void* dxn, int B, int classes, int D, void* stream) {
  return head_backward_scaled(dlogits, xn, head_w, dhead_w, dhead_b, dxn, B, classes, D, nullptr, nullptr, stream);
}

// Does the 160 x 256 x 64 tile really take this product of the model?  cara_gemm8_policy looks at the shape only; the tile's
// adapter-inside path and its riders are built for one r-tile (cara_gemm8_plan: Rp = 32, rank <= 16).  The activation LAYOUT
// (K-panel-major h / dH for the 128-tile family, row-major for the tile) must follow what will run: at rank 64 the shape-only
// answer left fc2 forward on the 128-tile kernel reading row-major A (9.696 -> 9.75 ms on the rank-64 leg).
static bool tile_policy(const cara_geom* g, int M, int N, int K, int riders) {
  return g->Rp == 32 && g->rank > 0 && g->rank <= 16 && cara_gemm8_policy(M, N, K, riders);
}

extern "C" int cara_vit_forward(const cara_geom* g, const cara_vit_shape* s, const cara_vit_weights* w,
                                const cara_cp* cp, const float* head_w, const float* head_b, const float* images,
                                const float* droppath, void* workspace, float* logits, void* stream) {
  Ws W;
  if (!layout(g, s, &W) || !w || !cp || !head_w || !head_b || !images || !workspace || !logits) return CARA_E_ARG;
  char* ws = static_cast<char*>(workspace);
  const int D = g->dim, M = W.M, Rp = g->Rp, B = s->B, N = s->tokens, P = N - 1;
  const float att_scale = 1.0f / sqrtf((float)(D / g->heads));
  cara_pack_layout pl;
  TRY(cara_pack_offsets(g, &pl));
  TRY(cara_factor_prep(g, cp, w->proj_b, w->fc1_b, w->fc2_b, ws + W.pack, stream));
  const bool dense_qkv = g->cp_length == 2;   // order-2 tensorisation: the QKV linear in the dense-delta form
  if (dense_qkv) TRY(cara_dense_delta_materialize(g, cp, ws + W.dd, ws + W.ddt, stream));
  // patch embedding: Conv2d(k = s = patch) as a GEMM over im2col rows, then cls + pos_embed
  const int kp = s->chans * s->patch * s->patch;
  TRY(cara_im2col_patches(images, ws + W.patches, B, s->chans, s->img, s->img, s->patch, stream));
  cara_gemm_args a = {};
  a.A = ws + W.patches; a.lda = kp; a.B = w->patch_w; a.ldb = kp; a.M = B * P; a.N = D; a.K = kp;
  a.bias = w->patch_b; a.epi = CARA_EPI_F32; a.C = ws + W.emb; a.ldc = D;
  Ctx cx{stream, ws + W.gemm_scratch, 0, false};
  cx.rank = g->rank;
  with_scratch(a, cx);
  TRY(cara_gemm_bf16(&a, stream));
  TRY(cara_assemble_tokens(reinterpret_cast<float*>(ws + W.emb), w->cls, w->pos,
                           reinterpret_cast<float*>(ws + W.layer[0].x_in), B, P, D, stream));
  for (int l = 0; l < g->depth; ++l) {
    const LayerWs& lw = W.layer[l];
    Lin lin[4];
    make_lins(g, w, ws + W.pack, pl, l, lin);
    float* x_in = reinterpret_cast<float*>(ws + lw.x_in);
    float* x_mid = reinterpret_cast<float*>(ws + lw.x_mid);
    float* x_out = reinterpret_cast<float*>(ws + (l + 1 < g->depth ? W.layer[l + 1].x_in : W.x_last));
    const float* dp1 = droppath ? droppath + (size_t)(2 * l) * B : nullptr;
    const float* dp2 = droppath ? droppath + (size_t)(2 * l + 1) * B : nullptr;
    // Only the cls token of the LAST block's output reaches the logits (timm takes x[:, 0] after
    // the final norm), and proj / LayerNorm / fc1 / fc2 act per token: in the last block they run
    // on the B cls rows only (row stride N*D).  Exact, not an approximation: every other row of
    // that block's proj/MLP output is dead.  qkv and attention still see all tokens (keys/values).
    const bool cls_only = (l == g->depth - 1) && cls_shortcut_enabled() && !s->wd_exact;
    const int Mr = cls_only ? B : M;
    const int ldr = cls_only ? N * D : D;          // row stride of the residual stream rows used
    const int rps = cls_only ? 1 : N;              // rows per sample for the DropPath multipliers
    // x = x + drop_path(attn(norm1(x)))
    const bool ex = s->wd_exact != 0;   // exact weight-dropout mode: plain GEMMs on the merged weights
    const bool fx = fuse_xu(g) && !ex;
    // no backward will follow (cara_vit_shape::inference): do not keep what only it reads
    const bool inference = s->inference != 0 && !ex;
    cx.layer = l;
    cx.full = !cls_only;
    Ctx cx_all = cx;   // qkv and attention always run on all token rows
    cx_all.full = true;
    // K-panel-major activations (panel_acts): pa_x for what all M token rows produce (xn1), pa for the Mr rows of
    // the proj / MLP half of the block (xn2, h)
    // (order 2: xn1 stays row-major -- the dense dD = xn1^T dY of the backward reads it with transposing LDS reads)
    // (h stays row-major where fc2 forward / fc1 dX run on the 160 x 256 x 64 tile, which stages whole 128-byte lines of row-major operands)
    const bool pa_x = panel_acts(M, s, 2) && !dense_qkv, pa = panel_acts(Mr, s, 1) && !tile_policy(g, Mr, D, 4 * D, 0), pa_n = panel_acts(Mr, s, 2);
    {
      SiteBracket sb(CARA_SITE_LN1_FWD, cx_all);
      TRY(cara_layernorm_fwd_ex(x_in, D, w->ln1_g + (size_t)l * D, w->ln1_b + (size_t)l * D, ws + lw.xn1,
                                reinterpret_cast<float*>(ws + lw.mean1), reinterpret_cast<float*>(ws + lw.rstd1), M, D, s->eps,
                                (fx && !dense_qkv) ? lin[0].Ut : nullptr, g->rank, Rp, ws + lw.T[0], ws + lw.Tt[0], W.ldt, pa_x ? M : 0, stream));
    }
    cara_gemm_args e = {};
    e.epi = CARA_EPI_BF16; e.C = ws + lw.qkv;
    if (ex) TRY(lin_fwd_exact(lin[0], reinterpret_cast<bf16*>(ws + lw.xn1), D, M, Rp, ws, W, s, e, cx_all));
    else if (dense_qkv) TRY(lin_fwd_dense(lin[0], reinterpret_cast<bf16*>(ws + lw.xn1), D, M, reinterpret_cast<bf16*>(ws + W.dd) + (size_t)l * 3 * D * D, e, cx_all));
    else TRY(lin_fwd(lin[0], reinterpret_cast<bf16*>(ws + lw.xn1), pa_x ? -M : D, M, Rp, W.ldt, ws, lw, e, cx_all, fx));
    {
      SiteBracket sb(CARA_SITE_ATTN_FWD, cx_all);
      // (the last block: only the cls row of the attention output is read -- by proj on the cls rows -- so only the cls query runs)
      if (cls_only && cls_attention_enabled())
        TRY(cara_attention_cls_fwd(ws + lw.qkv, ws + lw.ao, reinterpret_cast<float*>(ws + lw.lse), B, N, g->heads, att_scale, stream));
      else
        TRY(cara_attention_fwd(ws + lw.qkv, ws + lw.ao, reinterpret_cast<float*>(ws + lw.lse), B, N, g->heads, att_scale, stream));
    }
    e = {};
    e.epi = CARA_EPI_RESID; e.C = x_mid; e.aux = x_in; e.rowscale = dp1; e.rows_per_sample = rps; e.ldc = ldr;
#ifdef CARA_ABLATE_RESID   // timing experiment only: the residual products with a plain bf16 epilogue (wrong results)
    e = {}; e.epi = CARA_EPI_BF16; e.C = ws + lw.u;
#endif
    if (ex) TRY(lin_fwd_exact(lin[1], reinterpret_cast<bf16*>(ws + lw.ao), ldr, Mr, Rp, ws, W, s, e, cx));
    else TRY(lin_fwd(lin[1], reinterpret_cast<bf16*>(ws + lw.ao), ldr, Mr, Rp, W.ldt, ws, lw, e, cx));
    // x = x + drop_path(mlp(norm2(x)))
    {
      SiteBracket sb(CARA_SITE_LN2_FWD, cx);
      TRY(cara_layernorm_fwd_ex(x_mid, ldr, w->ln2_g + (size_t)l * D, w->ln2_b + (size_t)l * D, ws + lw.xn2,
                                reinterpret_cast<float*>(ws + lw.mean2), reinterpret_cast<float*>(ws + lw.rstd2), Mr, D, s->eps,
                                fx ? lin[2].Ut : nullptr, g->rank, Rp, ws + lw.T[2], ws + lw.Tt[2], W.ldt, pa_n ? Mr : 0, stream));
    }
    e = {};
    // what the backward's fc2 dX epilogue needs: gelu'(u) itself, kept as IEEE half (CARA_EPI_GELU_DG / _MULH), or the pre-activation u
    e.epi = save_gelu_grad(g, s) ? CARA_EPI_GELU_DG : CARA_EPI_GELU; e.C = ws + lw.h;
    e.C2 = inference ? nullptr : ws + lw.u;   // (only read by the backward)
    if (pa) { e.c_panels = Mr; e.ldc = 4 * D; }   // h (and dH in the backward) K-panel-major
    if (ex) TRY(lin_fwd_exact(lin[2], reinterpret_cast<bf16*>(ws + lw.xn2), D, Mr, Rp, ws, W, s, e, cx));
    else TRY(lin_fwd(lin[2], reinterpret_cast<bf16*>(ws + lw.xn2), pa_n ? -Mr : D, Mr, Rp, W.ldt, ws, lw, e, cx, fx));
    e = {};
    e.epi = CARA_EPI_RESID; e.C = x_out; e.aux = x_mid; e.rowscale = dp2; e.rows_per_sample = rps; e.ldc = ldr;
#ifdef CARA_ABLATE_RESID
    e = {}; e.epi = CARA_EPI_BF16; e.C = ws + lw.u;
#endif
    if (ex) TRY(lin_fwd_exact(lin[3], reinterpret_cast<bf16*>(ws + lw.h), 4 * D, Mr, Rp, ws, W, s, e, cx));
    else TRY(lin_fwd(lin[3], reinterpret_cast<bf16*>(ws + lw.h), pa ? -Mr : 4 * D, Mr, Rp, W.ldt, ws, lw, e, cx));
  }
  // norm -> cls token -> head  (LayerNorm is per token, so only the cls rows are normalised)
  // ... all in fp32, one launch (cara_head_forward): the 16-bit xn and the row statistics are kept for the backward
  return cara_head_forward(reinterpret_cast<float*>(ws + W.x_last), (long)N * D, w->norm_g, w->norm_b, head_w, head_b, ws + W.clsn,
                           reinterpret_cast<float*>(ws + W.meanF), reinterpret_cast<float*>(ws + W.rstdF), logits, B, s->num_classes,
                           D, s->eps, stream);
}

extern "C" int cara_vit_backward(const cara_geom* g, const cara_vit_shape* s, const cara_vit_weights* w,
                                 const cara_cp* cp, const float* head_w, const float* dlogits, const float* droppath,
                                 void* workspace, const cara_cp* grads, float* dhead_w, float* dhead_b, void* stream) {
  Ws W;
  if (!layout(g, s, &W) || !w || !cp || !head_w || !dlogits || !workspace || !grads || !dhead_w || !dhead_b) return CARA_E_ARG;
  char* ws = static_cast<char*>(workspace);
  TsPending pending;
  SlabFormats formats;
  Ctx cx{stream, ws + W.gemm_scratch, 0, false, &pending};
  cx.rank = g->rank;
  cx.fmt = &formats;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const int D = g->dim, M = W.M, Rp = g->Rp, B = s->B, N = s->tokens;
  const float att_scale = 1.0f / sqrtf((float)(D / g->heads));
  cara_pack_layout pl;
  TRY(cara_pack_offsets(g, &pl));
  float* dx = reinterpret_cast<float*>(ws + W.dx);
  const Ws::Bwd& R = W.bwd;
  bf16* dyb = reinterpret_cast<bf16*>(ws + R.dyb_fc2);   // dY of the last block's fc2
  TRY(head_backward_scaled(dlogits, ws + W.clsn, head_w, dhead_w, dhead_b, ws + W.dclsn, B, s->num_classes, D, s->loss_scale,
                           s->found_inf, stream));
  // Gradient enters the token stream only through the cls rows.  With the last block on its cls rows (the default) every kernel of that
  // block reads dx / dyb on those rows only, and its LayerNorm-1 backward -- the first launch that touches every row of dx -- is told
  // so (cara_layernorm_bwd_rows_in: dx_in on every N-th row): the two buffers need no zeroing (58 MB of memset + 39 MB of zeros read
  // back, ~18 us per step, r05).  Otherwise (CARA_CLS_SHORTCUT=0, exact weight dropout) the last block runs on all rows: zeros.
  static const int force_zero = env_once("CARA_BWD_MEMSETS", 0);   // (1: the memsets and the all-rows read back, for A/B runs)
  const bool last_on_cls = cls_shortcut_enabled() && !s->wd_exact && !force_zero;
  if (!last_on_cls) {
    if (hipMemsetAsync(dx, 0, (size_t)M * D * 4, hs) != hipSuccess) return CARA_E_LAUNCH;
    if (hipMemsetAsync(dyb, 0, (size_t)M * D * 2, hs) != hipSuccess) return CARA_E_LAUNCH;
  }
  const float* dp_last = droppath ? droppath + (size_t)(2 * (g->depth - 1) + 1) * B : nullptr;
  TRY(cara_layernorm_bwd(ws + W.dclsn, reinterpret_cast<float*>(ws + W.x_last), (long)N * D, w->norm_g,
                         reinterpret_cast<float*>(ws + W.meanF), reinterpret_cast<float*>(ws + W.rstdF), nullptr, dx, dyb,
                         dp_last, 1, B, D, stream));
  const bool ex = s->wd_exact != 0;
  const bool fx = fuse_xu(g) && !ex;
  const bool dense_qkv = g->cp_length == 2;
  bool have_G_fc2 = false;   // G' of this block's fc2 was left by the LayerNorm backward of the block above
  SideRiders side;
  for (int l = g->depth - 1; l >= 0; --l) {
    const LayerWs& lw = W.layer[l];
    Lin lin[4];
    make_lins(g, w, ws + W.pack, pl, l, lin);
    const float* dp1 = droppath ? droppath + (size_t)(2 * l) * B : nullptr;
    const float* dp_prev = (droppath && l > 0) ? droppath + (size_t)(2 * (l - 1) + 1) * B : nullptr;
    // last block: only the cls rows carry gradient into proj / MLP (see cara_vit_forward)
    const bool cls_only = (l == g->depth - 1) && cls_shortcut_enabled() && !s->wd_exact;
    const int Mr = cls_only ? B : M;
    const int ldr = cls_only ? N * D : D;
    const int rps = cls_only ? 1 : N;
    cx.layer = l;
    cx.full = !cls_only;
    Ctx cx_all = cx;   // attention and qkv always run on all token rows
    cx_all.full = true;
    // (dyb and G'[3] of this block's fc2 were written by the block above / the final norm's backward)
    bf16* dyp = reinterpret_cast<bf16*>(ws + R.dyb_proj);
    bf16* dH = reinterpret_cast<bf16*>(ws + R.dH);
    bf16* dQKV = reinterpret_cast<bf16*>(ws + R.dQKV);
    // ---- mlp branch: dY = drop_path scale * dx (already in dyb) ----
    cara_gemm_args e = {};
    e.epi = save_gelu_grad(g, s) ? CARA_EPI_MULH : CARA_EPI_DGELU; e.C = dH; e.aux = ws + lw.u;
    // K-panel-major activations, as the forward wrote them (xn1: pa_x; xn2, h: pa) and as the kernels here write
    // theirs: dH and dyp (pa), dyb of the block below (pa_x).  This block's own dyb came from the block above --
    // panels -- except in the last block, where the final norm's backward left it row-major on the cls rows.
    // (h as the forward wrote it: row-major where fc2 forward runs on the 160 x 256 x 64 tile; dH: row-major where fc1 dX does)
    const bool pa_x = panel_acts(M, s, 2) && !dense_qkv, pa = panel_acts(Mr, s, 1) && !tile_policy(g, Mr, D, 4 * D, 0), pa_n = panel_acts(Mr, s, 2);
    // fc1's dVs / dc and fc2's dU out of the fc2 dX epilogue: the fc1 dX launch then carries nothing and runs on the 160 x 256 x 64 tile
    const bool er_on = epi_riders_geom(g, Mr, ex) && ts_rank(cx, Rp) <= 16 && fuse_ts(Mr, Rp) && defer_du() && env_once_dv_off();
    // CARA_DV (default 0; bit 0: fc1, bit 1: qkv): dVs (+ dc) of fc1 / qkv out of their dX GEMM's own A tiles on the 160 x 256 x 64 tile;
    // CARA_DV_DU_HOME (default 1): fc2 then carries BOTH its products in its own launch, and the dU waiting from the block above rides
    // behind fc1's tiles (19 MB instead of fc2's 77)
    static const int dv_env = env_once("CARA_DV", 0), dv_home = env_once("CARA_DV_DU_HOME", 1);
    const bool dv_ok = dv_env != 0 && !ex && Rp == 32 && ts_rank(cx, Rp) <= 16 && fuse_ts(Mr, Rp) && defer_du() && !cls_only;
    const bool dv_fc1 = dv_ok && (dv_env & 1) && g_inside_enabled(2) && tile_policy(g, Mr, D, 4 * D, 0);
    const bool dv_qkv = dv_ok && (dv_env & 2) && g_inside_enabled(0) && tile_policy(g, M, D, 3 * D, 0);
    static const int fc1_side = env_once("CARA_FC1_SIDE", 0);
    const bool side_on = fc1_side != 0 && !er_on && !ex && Rp == 32 && ts_rank(cx, Rp) <= 16 && fuse_ts(Mr, Rp) && defer_du() && g_inside_enabled(2) &&
                         tile_policy(g, Mr, D, 4 * D, 0) && !dv_fc1 && side_stream() != nullptr;
    side.ss = side_on ? side_stream() : nullptr;
    EpiRider er;
    er.Tt = ws + lw.Tt[2];
    er.slabV = ws + W.slabV[2] + (size_t)l * W.strideV[2];
    er.bytes = W.strideV[2] < W.strideU[3] ? W.strideV[2] : W.strideU[3];
    const bool pa_dh = panel_acts(Mr, s, 1) && !tile_policy(g, Mr, D, 4 * D, (er_on || side_on || dv_fc1) ? 0 : 1);
    const bool pa_dp = panel_acts(Mr, s, 4), pa_dx = panel_acts(M, s, 4);   // dyp here; dyb of the block below
    const bool pa_dyb = pa_dx && l < g->depth - 1;
    if (pa_dh) { e.c_panels = Mr; e.ldc = 4 * D; }
    if (ex) TRY(lin_bwd_exact(lin[3], dyb, reinterpret_cast<bf16*>(ws + lw.h), Mr, Rp, ws, W, s, true, e, true, cx));
    else TRY(lin_bwd(lin[3], dyb, pa_dyb ? -M : ldr, reinterpret_cast<bf16*>(ws + lw.h), pa ? -Mr : 4 * D, Mr, Rp, W.ldt, ws, W, lw, true, e, true, cx,
                     have_G_fc2, er_on ? &er : nullptr, false, nullptr, (dv_fc1 && dv_home) ? 2 : 0));
    if (er.done) formats.V[2][l] = formats.U[3][l];   // (fc1's dVs slabs: the same launch, the same count of row tiles)
    have_G_fc2 = false;
    e = {};
    e.epi = CARA_EPI_BF16; e.C = ws + W.dXn;
    if (ex) TRY(lin_bwd_exact(lin[2], dH, reinterpret_cast<bf16*>(ws + lw.xn2), Mr, Rp, ws, W, s, true, e, true, cx));
    else TRY(lin_bwd(lin[2], dH, pa_dh ? -Mr : 4 * D, reinterpret_cast<bf16*>(ws + lw.xn2), pa_n ? -Mr : D, Mr, Rp, W.ldt, ws, W, lw, true, e, true, cx, false,
                     nullptr, er.done, side_on ? &side : nullptr, dv_fc1 ? 1 : 0));
    // dyp = dY of this block's proj: its G' = dY Vs comes out of the same kernel (CARA_LN2B_XU=0: out of proj's dX GEMM instead)
    static const int ln2b_xu = env_once("CARA_LN2B_XU", 1);
    const bool fxp = fx && (ln2b_xu != 0 || cls_only);
    {
      SiteBracket sb(CARA_SITE_LN2_BWD, cx);
      TRY(cara_layernorm_bwd_ex(ws + W.dXn, reinterpret_cast<float*>(ws + lw.x_mid), ldr, w->ln2_g + (size_t)l * D,
                                reinterpret_cast<float*>(ws + lw.mean2), reinterpret_cast<float*>(ws + lw.rstd2), dx, dx, dyp, dp1,
                                rps, Mr, D, fxp ? lin[1].Vst : nullptr, g->rank, Rp, ws + R.G[1], ws + R.Gt[1], W.ldt, pa_dp ? Mr : 0, stream));
    }
    // ---- attention branch ----
    e = {};
    e.epi = CARA_EPI_BF16; e.C = ws + W.dAO; e.ldc = ldr;
    const bool cls_attn = cls_only && cls_attention_enabled();   // (then only the cls rows of dAO are ever read)
    if (cls_only && !cls_attn && hipMemsetAsync(ws + W.dAO, 0, (size_t)M * D * 2, hs) != hipSuccess) return CARA_E_LAUNCH;
    if (ex) TRY(lin_bwd_exact(lin[1], dyp, reinterpret_cast<bf16*>(ws + lw.ao), Mr, Rp, ws, W, s, true, e, true, cx));
    else TRY(lin_bwd(lin[1], dyp, pa_dp ? -Mr : ldr, reinterpret_cast<bf16*>(ws + lw.ao), ldr, Mr, Rp, W.ldt, ws, W, lw, true, e, true, cx, fxp));
    {
      SiteBracket sb(CARA_SITE_ATTN_BWD, cx_all);
      if (cls_attn)
        TRY(cara_attention_cls_bwd(ws + lw.qkv, ws + lw.ao, ws + W.dAO, reinterpret_cast<float*>(ws + lw.lse), dQKV, B, N,
                                   g->heads, att_scale, stream));
      else
        TRY(cara_attention_bwd(ws + lw.qkv, ws + lw.ao, ws + W.dAO, reinterpret_cast<float*>(ws + lw.lse), dQKV, B, N,
                               g->heads, att_scale, stream));
    }
    e = {};
    e.epi = CARA_EPI_BF16; e.C = ws + W.dXn;
    // block 0 has nothing trainable upstream of it: its dX GEMM and LayerNorm backward are skipped
    if (ex) TRY(lin_bwd_exact(lin[0], dQKV, reinterpret_cast<bf16*>(ws + lw.xn1), M, Rp, ws, W, s, l > 0, e, false, cx_all));
    else if (dense_qkv)
      TRY(lin_bwd_dense(lin[0], dQKV, reinterpret_cast<bf16*>(ws + lw.xn1), M, reinterpret_cast<bf16*>(ws + W.ddt) + (size_t)l * 3 * D * D,
                        reinterpret_cast<float*>(ws + W.dD) + (size_t)l * 3 * D * D, reinterpret_cast<float*>(ws + W.dd_slabs), l > 0, e, cx_all));
    else TRY(lin_bwd(lin[0], dQKV, 3 * D, reinterpret_cast<bf16*>(ws + lw.xn1), pa_x ? -M : D, M, Rp, W.ldt, ws, W, lw, l > 0, e, false, cx_all, false, nullptr,
                     false, nullptr, dv_qkv ? 1 : 0));
    // the side stream's products read dH and G' of fc2, which the kernels below overwrite: the caller's stream waits for them (they
    // were launched some 300 us ago)
    if (side.pending_join) {
      if (hipStreamWaitEvent(hs, side.ss ? side.ss->join : side_stream()->join, 0) != hipSuccess) return CARA_E_LAUNCH;
      side.pending_join = false;
    }
    if (l > 0) {
      // dyb = dY of fc2 of the block BELOW (all M rows there: only the last block runs on cls rows); this block's
      // fc2 is through with the buffer (stream order)
      Lin below[4];
      make_lins(g, w, ws + W.pack, pl, l - 1, below);
      SiteBracket sb(CARA_SITE_LN1_BWD, cx_all);
      TRY(cara_layernorm_bwd_rows_in(ws + W.dXn, reinterpret_cast<float*>(ws + lw.x_in), D, w->ln1_g + (size_t)l * D,
                                     reinterpret_cast<float*>(ws + lw.mean1), reinterpret_cast<float*>(ws + lw.rstd1), dx, dx, dyb,
                                     dp_prev, N, M, D, fx ? below[3].Vst : nullptr, g->rank, Rp, ws + R.G[3], ws + R.Gt[3], W.ldt,
                                     pa_dx ? M : 0, (cls_only && last_on_cls) ? N : 0, stream));   // (last block: dx holds the cls rows only so far)
      have_G_fc2 = fx;
    }
  }
  TRY(flush_pending(cx));   // block 0's qkv products have no dX GEMM behind them
  if (!ex) {   // (the exact mode wrote dU / dVs / dc of every layer directly)
    const int ins[4] = {D, D, D, 4 * D}, outs[4] = {3 * D, D, 4 * D, D};
    const int L = g->depth;
    cara_ts_reduce red[CARA_TS_REDUCE_MAX];   // all slab sums of the pass in ONE launch (8 + 6 of them, more where a linear's layers differ in slab format)
    int nred = 0;
    const int Rc = (Rp == 32 && ts_rank(cx, Rp) <= 16) ? 16 : 0;   // (every product of the pass was told the same rank)
    auto push = [&](const cara_ts_reduce& e) -> int {
      if (nred == CARA_TS_REDUCE_MAX) {
        TRY(cara_tskinny_reduce_many(red, nred, stream));
        nred = 0;
      }
      red[nred++] = e;
      return CARA_OK;
    };
    for (int i = 0; i < 4; ++i) {
      if (i == 0 && dense_qkv) continue;   // order 2: the QKV linear wrote no skinny slabs (its gradient is dense, below)
      float* dU = reinterpret_cast<float*>(ws + W.dU[i]);
      float* dVs = reinterpret_cast<float*>(ws + W.dVs[i]);
      float* dc = i == 0 ? nullptr : reinterpret_cast<float*>(ws + W.dc[i]);
      // qkv (i == 0) sees all tokens in every block; proj / fc1 / fc2 of the last block ran on B rows,
      // so that block's slabs have their own chunking
      const int full = (i == 0 || !cls_shortcut_enabled()) ? L : L - 1;   // (not reached in the exact mode)
      // runs of layers whose products left the same slab format (one per block / one per wave: SlabFormats)
      for (int l0 = 0; l0 < full;) {
        int l1 = l0 + 1;
        while (l1 < full && formats.U[i][l1] == formats.U[i][l0]) ++l1;
        TRY(push(cara_ts_reduce{ws + W.slabU[i] + (size_t)l0 * W.strideU[i], W.strideU[i], dU + (size_t)l0 * ins[i] * Rp, nullptr, l1 - l0, M, ins[i], Rp, Rc,
                                formats.U[i][l0]}));
        l0 = l1;
      }
      for (int l0 = 0; l0 < full;) {
        int l1 = l0 + 1;
        while (l1 < full && formats.V[i][l1] == formats.V[i][l0]) ++l1;
        TRY(push(cara_ts_reduce{ws + W.slabV[i] + (size_t)l0 * W.strideV[i], W.strideV[i], dVs + (size_t)l0 * outs[i] * Rp, dc ? dc + (size_t)l0 * outs[i] : nullptr,
                                l1 - l0, M, outs[i], Rp, Rc, formats.V[i][l0]}));
        l0 = l1;
      }
      if (i != 0 && cls_shortcut_enabled()) {
        const size_t l = L - 1;
        TRY(push(cara_ts_reduce{ws + W.slabU[i] + l * W.strideU[i], 0, dU + l * ins[i] * Rp, nullptr, 1, B, ins[i], Rp, Rc, 0}));
        TRY(push(cara_ts_reduce{ws + W.slabV[i] + l * W.strideV[i], 0, dVs + l * outs[i] * Rp, dc + l * outs[i], 1, B, outs[i], Rp, Rc, 0}));
      }
    }
    TRY(cara_tskinny_reduce_many(red, nred, stream));
  }
  cara_layer_grads lg;
  lg.dU_qkv = reinterpret_cast<float*>(ws + W.dU[0]); lg.dVs_qkv = reinterpret_cast<float*>(ws + W.dVs[0]);
  lg.dU_proj = reinterpret_cast<float*>(ws + W.dU[1]); lg.dVs_proj = reinterpret_cast<float*>(ws + W.dVs[1]);
  lg.dU_fc1 = reinterpret_cast<float*>(ws + W.dU[2]); lg.dVs_fc1 = reinterpret_cast<float*>(ws + W.dVs[2]);
  lg.dU_fc2 = reinterpret_cast<float*>(ws + W.dU[3]); lg.dVs_fc2 = reinterpret_cast<float*>(ws + W.dVs[3]);
  lg.dc_proj = reinterpret_cast<float*>(ws + W.dc[1]); lg.dc_fc1 = reinterpret_cast<float*>(ws + W.dc[2]);
  lg.dc_fc2 = reinterpret_cast<float*>(ws + W.dc[3]);
  TRY(cara_factor_grad_reduce_ex(g, cp, &lg, grads, ws + W.gscratch, s->loss_scale, s->found_inf, stream));
  if (dense_qkv)   // CP_A1 / CP_A2 / CP_R1 of the order-2 tensorisation, from the dense dD of every (layer, projection)
    TRY(cara_dense_delta_grad(g, cp, reinterpret_cast<const float*>(ws + W.dD), grads, ws + W.dd_scratch, stream));
  return CARA_OK;
}
