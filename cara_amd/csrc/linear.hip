// cara_linear_fwd / cara_linear_bwd: one adapted linear of /root/reference/src/cara/cara.py (:25-42 qkv, :50-58 proj, :75-82 fc1,
// :87-93 fc2) per call, in factored form, as plain compositions of this library's entry points -- what an
// integrator who patches one nn.Linear at a time binds (SURVEY 8b's minimum export set).  Host code only: every kernel is behind
// the calls below, all on the caller's stream, nothing allocated.
#include "common.h"

#define TRY(expr)                   \
  do {                              \
    const int _st = (expr);         \
    if (_st != CARA_OK) return _st; \
  } while (0)

static bool linear_ok(const cara_linear* L) {
  return L && L->W && L->Ut && L->Vs && L->in > 0 && L->out > 0 && (L->Rp == 32 || L->Rp == 64) && L->rank >= 1 && L->rank <= L->Rp;
}

extern "C" int cara_linear_fwd(const cara_linear* L, const void* X, int ldx, int M, void* T, void* Tt, int ldt, int epi, void* Y, int ldy,
                               void* Y2, const void* aux, const float* rowscale, int rows_per_sample, void* stream) {
  if (!linear_ok(L) || !X || !T || !Y || M <= 0 || ldx < L->in) return CARA_E_ARG;
  TRY(cara_skinny_xu_r(X, ldx, L->Ut, T, Tt, ldt, M, L->in, L->Rp, L->rank, stream));   // T = X U
  cara_gemm_args a = {};
  a.A = X; a.lda = ldx; a.B = L->W; a.ldb = L->in; a.A2 = T; a.B2 = L->Vs; a.Rp = L->Rp;
  a.M = M; a.N = L->out; a.K = L->in; a.bias = L->bias; a.epi = epi;
  a.C = Y; a.ldc = ldy ? ldy : L->out; a.C2 = Y2; a.aux = aux; a.rowscale = rowscale; a.rows_per_sample = rows_per_sample;
  return cara_gemm_bf16(&a, stream);                                                      // [X | T] [W | Vs]^T + bias -> epilogue
}

extern "C" int cara_linear_bwd(const cara_linear* L, const void* dY, int lddy, const void* X, int ldx, const void* Tt, int M, void* G,
                               void* Gt, int ldt, void* dX, int lddx, void* slabs_u, void* slabs_v, float* dU, float* dVs, float* dc,
                               void* stream) {
  if (!linear_ok(L) || !L->Vst || !dY || !X || !Tt || !G || !Gt || !slabs_u || !slabs_v || !dU || !dVs || M <= 0) return CARA_E_ARG;
  if (dX && (!L->Wt || !L->U)) return CARA_E_ARG;
  TRY(cara_skinny_xu_r(dY, lddy, L->Vst, G, Gt, ldt, M, L->out, L->Rp, L->rank, stream));   // G' = dY Vs
  if (dX) {                                                                                  // dX = [dY | G'] [W^T | U]^T
    cara_gemm_args a = {};
    a.A = dY; a.lda = lddy; a.B = L->Wt; a.ldb = L->out; a.A2 = G; a.B2 = L->U; a.Rp = L->Rp;
    a.M = M; a.N = L->in; a.K = L->out; a.epi = CARA_EPI_BF16; a.C = dX; a.ldc = lddx ? lddx : L->in;
    TRY(cara_gemm_bf16(&a, stream));
  }
  // dU = X^T G', dVs = dY^T T, dc = colsum dY: partial sums into the slabs, then one reduction launch
  TRY(cara_tskinny_partial2_r(X, ldx, Gt, slabs_u, L->in, dY, lddy, Tt, slabs_v, L->out, dc ? 1 : 0, ldt, M, L->Rp, L->rank, stream));
  const int Rc = (L->Rp == 32 && L->rank <= 16) ? 16 : 0;
  cara_ts_reduce red[2] = {{slabs_u, 0, dU, nullptr, 1, M, L->in, L->Rp, Rc, 0}, {slabs_v, 0, dVs, dc, 1, M, L->out, L->Rp, Rc, 0}};
  return cara_tskinny_reduce_many(red, 2, stream);
}
