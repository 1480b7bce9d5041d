// inr_api.hip -- the C-ABI of libinr_mi355x.so (see include/inr_abi.h).  Plain pointers and
// sizes only; validates arguments up front; never allocates device memory; never syncs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/inr_abi.h"
#include <string>

#include "inr_aux.h"
#include "inr_dw_gemm.h"
#include "inr_dw_gemm_bf16.h"
#include "inr_w2.h"

#ifdef INR_STAMPS
namespace inr {
long long* g_stamp_buf = nullptr;  // diagnostic build only (make dbg): phase stamps of the fused kernels, entry / exit
long long g_stamp_cap = 0;         // stamps of the GEMMs; entries behind it: a stamp whose index is not below this is dropped
}  // namespace inr
using inr::g_stamp_buf;
using inr::g_stamp_cap;
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(INR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// smallest built block count that holds `need` 32-row blocks (narrower nets run zero-padded), or -1
inline int pick_nb(int need, std::initializer_list<int> built) {
  for (int nb : built)
    if (nb >= need) return nb;
  return -1;
}

constexpr int kMaxBlocks = 256;  // one persistent workgroup per CU (MI355X: 256 CUs)


}  // namespace

struct inr_plan {
  inr_net_desc desc;
  NetDesc nd;
  int64_t packed_floats;
  // split steps (step_schedule below): a low-priority stream for the part of the weight-gradient GEMM that runs beside
  // the fused kernel's last, partial round.  Created on first use, destroyed with the plan; the only state a plan has.
  mutable std::mutex side_mu;
  mutable hipStream_t side = nullptr;
  mutable int side_dev = -1;
  mutable hipEvent_t fork = nullptr, join = nullptr;  // the split step's two events, created with the side stream
  // bf16 plans: the gradient-scale state of the 8-bit stash (inr_w2.h), W2_STATE_FLOATS floats on the device, allocated
  // with the plan; what the host remembers about it: whether a kind of step (0 fused, 1 split) has been calibrated, and
  // for which batch size / loss.  One stream at a time may step a bf16 plan.
  mutable float* dz_state = nullptr;
  mutable int dz_dev = -1;
  mutable bool dz_ready[2] = {false, false};
  mutable int64_t dz_rows[2] = {0, 0};
  mutable int dz_loss[2] = {-1, -1};
  // bf16 GEMM chunking knobs (tuning aids), read from the environment ONCE, when the plan is created: workspace sizes must
  // not depend on what the environment holds at the time of a later call
  bool gemm_one_class = false;
  double gemm_enc_cost = 0.0;  // 0: kEncCost
};

// gradient-scale state of a bf16 plan on the current device (inr_w2.h): S = mult = S_used = 1, amax = 0 for both kinds of
// step.  Allocated with the plan; moved if the plan is later driven on another device.
static float* dz_state_alloc(const inr_plan* p) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (p->dz_state != nullptr && p->dz_dev == dev) return p->dz_state;
  if (p->dz_state != nullptr) (void)hipFree(p->dz_state);
  p->dz_state = nullptr;
  p->dz_ready[0] = p->dz_ready[1] = false;
  const float init[W2_STATE_FLOATS] = {1.f, 0.f, 1.f, 1.f, 1.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (hipMalloc(reinterpret_cast<void**>(&p->dz_state), sizeof(init)) != hipSuccess) {
    p->dz_state = nullptr;
    return nullptr;
  }
  if (hipMemcpy(p->dz_state, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p->dz_state);
    p->dz_state = nullptr;
    return nullptr;
  }
  p->dz_dev = dev;
  return p->dz_state;
}

// Does this call have to find its gradient scale first (a pass of the kernel whose stash nobody reads, then the roll)?
// Yes for a plan's first step of a kind, and when what the remembered scale was derived from no longer applies: another
// loss (fused steps normalise the batch size away, not the loss), or -- split steps, whose d(loss)/d(out) carries the
// 1 / count -- a batch more than twice or less than half as large.  From then on the scale follows the gradient from step
// to step (dz_state_roll) with 2^10.8 of headroom above the window it aims for (inr_w2.h W2_DZ_TARGET_EXP) -- a calibration
// pass is not counted as a clipped / flushed step (its roll gets no counters).
static bool dz_needs_calibration(const inr_plan* p, int kind, int64_t rows, int loss_kind) {
  return !p->dz_ready[kind] || (kind == 0 ? p->dz_loss[0] != loss_kind
                                          : (rows > 2 * p->dz_rows[1] || 2 * rows < p->dz_rows[1]));
}
// ... remembered once the kernels that use (or found) the scale have been launched
static void dz_mark(const inr_plan* p, int kind, int64_t rows, int loss_kind) {
  p->dz_ready[kind] = true;
  p->dz_rows[kind] = rows;
  p->dz_loss[kind] = loss_kind;
}

// Multiplicative filter networks (models/mfn.py).  L[] = filters 0..n | linears 0..n-1 | heads; flat
// parameters keep the state_dict order  linear.* , output_linear(.k).* , filters.*  (SURVEY Appendix B).
static int create_mfn_plan(const inr_net_desc* d, inr_plan** out) {
  const bool multi = d->kind == INR_KIND_MSFOURIER || d->kind == INR_KIND_MSBOUNDED;
  const bool gabor = d->kind == INR_KIND_GABOR || d->kind == INR_KIND_KGABOR;
  const int n = d->depth, W = d->width;
  if (n < 1 || 2 * n + 1 + (multi ? n + 1 : 1) + (gabor ? n + 1 : 0) > INR_MAX_LAYERS)
    return fail(INR_ERR_INVALID, "inr_plan_create: MFN depth %d", n);
  const int NB = W < 1 ? -1 : pick_nb((W + 31) / 32, {1, 4, 8, 16});
  if (NB < 0)
    return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: MFN width %d (kernels are built for widths 1..512)", W);
  // The filters read their input features from a [2E'][TL] image in the stash, k-step s -> rows s (lane half 0)
  // and E' + s (half 1).  INR_INPUT_GAUSS: the fused encoder writes it, E' = enc_size.  INR_INPUT_X: the kernel
  // transposes the tile's rows of x [B,in_features] into it, E' = half of in_features rounded up to 16 (rows past
  // in_features are zero and carry zero weights).
  int Ehalf;
  if (d->input == INR_INPUT_GAUSS) {
    if (d->enc_size < 8 || (d->enc_size % 8) != 0 || d->in_features != 2 * d->enc_size)
      return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: fused gauss encoder needs in_features == 2*enc_size, "
                  "enc_size %% 8 == 0");
    Ehalf = d->enc_size;
  } else if (d->input == INR_INPUT_X) {
    if (d->in_features < 1 || d->in_features > 4096)
      return fail(INR_ERR_INVALID, "inr_plan_create: in_features %d", d->in_features);
    Ehalf = round_up(d->in_features, 16) / 2;
  } else {
    return fail(INR_ERR_INVALID, "inr_plan_create: input mode %d", d->input);
  }
  if (d->out_features < 1 || d->out_features > 4)
    return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: out_features %d outside [1,4]", d->out_features);
  inr_plan* p = new (std::nothrow) inr_plan();
  if (p == nullptr) return fail(INR_ERR_INVALID, "inr_plan_create: out of host memory");
  p->desc = *d;
  NetDesc& nd = p->nd;
  memset(&nd, 0, sizeof(nd));
  nd.NB = NB;
  nd.NW = NB == 16 ? 2 : 4;
  nd.hact = ACT_SIN;
  nd.last_act = ACT_ID;
  nd.input = d->input == INR_INPUT_GAUSS ? IN_GAUSS : IN_X;
  nd.E = Ehalf;
  nd.out_f = d->out_features;
  nd.mfn_n = n;
  if (d->kind == INR_KIND_MSBOUNDED) {
    nd.bounded = 1;  // bounds default to "everything" until inr_plan_set_bounds is called
    for (int i = 0; i < INR_MAX_LAYERS / 2; ++i) {
      nd.bound_lo[i] = -1e30f;
      nd.bound_hi[i] = 1e30f;
    }
  }
  // heads: FourierNet -> output_linear after the last stage (mfn.py:85-94); multiscale -> output_linear[i]
  // for i in output_layers = [1,3,5,7] (mfn.py:223,262-263), those beyond depth do not exist
  if (multi) {
    const int stages[4] = {1, 3, 5, 7};
    for (int k = 0; k < 4; ++k)
      if (stages[k] <= n) nd.head_stage[nd.n_heads++] = stages[k];
    if (nd.n_heads == 0) {
      delete p;
      return fail(INR_ERR_INVALID, "inr_plan_create: multiscale MFN of depth %d has no output layer", n);
    }
  } else {
    nd.n_heads = 1;
    nd.head_stage[0] = n;
  }
  nd.mfn_stages = nd.head_stage[nd.n_heads - 1] + 1;
  const int n_head_layers = multi ? n + 1 : 1;
  nd.gabor = gabor ? 1 : 0;
  nd.mu0 = (n + 1) + n + n_head_layers;
  nd.D = nd.mu0 + (gabor ? n + 1 : 0);
  nd.ND = nd.D;
  const int TL = 32 * nd.NW;
  auto fill = [&](LayerDesc& L, int K, int M, bool filter, bool head) {
    L.K = K;
    L.M = M;
    L.Kpad8 = filter ? 2 * Ehalf : NB * 32;  // == round_up(K, 8) for the gauss encoder
    L.Kblk = filter ? (K + 31) / 32 : NB;  // hidden images always span all NB blocks (zero padding)
    L.Mblk = head ? (M + 31) / 32 : NB;
    L.Mpad8 = head ? round_up(M, 8) : NB * 32;
    L.ltype = LT_REAL;
    L.wn = M * K;
    L.bn = M;
    L.korder = filter ? 1 : 0;
  };
  for (int i = 0; i <= n; ++i) {
    fill(nd.L[i], d->in_features, W, true, false);
    nd.L[i].live = i < nd.mfn_stages;
  }
  for (int i = 0; i < n; ++i) {
    fill(nd.L[n + 1 + i], W, W, false, false);
    nd.L[n + 1 + i].live = i < nd.mfn_stages - 1;
  }
  for (int i = 0; i < n_head_layers; ++i) {
    fill(nd.L[2 * n + 1 + i], W, d->out_features, false, true);
    nd.L[2 * n + 1 + i].live = 0;
  }
  for (int k = 0; k < nd.n_heads; ++k) {
    nd.head_layer[k] = 2 * n + 1 + (multi ? nd.head_stage[k] : 0);
    nd.L[nd.head_layer[k]].live = 1;
  }
  if (gabor)  // (mu_i, gamma_i) of GaborLayer i as a (weight, bias) pair: same shapes as the filter's Linear
    for (int i = 0; i <= n; ++i) {
      fill(nd.L[nd.mu0 + i], d->in_features, W, true, false);
      nd.L[nd.mu0 + i].ltype = LT_GABOR_MU;
      nd.L[nd.mu0 + i].live = 1;
    }
  // flat offsets in state_dict order: linears, heads, filters
  int poff = 0;
  auto place = [&](LayerDesc& L) {
    L.w_off = poff;
    poff += L.wn;
    L.b_off = poff;
    poff += L.bn;
  };
  for (int i = 0; i < n; ++i) place(nd.L[n + 1 + i]);
  for (int i = 0; i < n_head_layers; ++i) place(nd.L[2 * n + 1 + i]);
  for (int i = 0; i <= n; ++i) {  // filters.i.mu, filters.i.gamma, filters.i.linear.weight, filters.i.linear.bias
    if (gabor) place(nd.L[nd.mu0 + i]);
    place(nd.L[i]);
  }
  nd.P = poff;
  int goff = 0;
  int64_t pk = 0;
  for (int l = 0; l < nd.D; ++l) {
    LayerDesc& L = nd.L[l];
    const bool filter = l <= n || l >= nd.mu0;
    const bool mu = l >= nd.mu0;
    const bool head = l >= 2 * n + 1 && l < nd.mu0;
    // hidden-width layers keep whole 32-row blocks in the slab (the plain dW pass stores them without a bounds test)
    L.gw_off = goff;
    goff += (head ? L.M : NB * 32) * L.K;
    L.gb_off = goff;
    goff += mu ? 2 * NB * 32 : (head ? L.M : NB * 32);  // LT_GABOR_MU: [s0 | T], NB*32 apart
    L.pf_off = (int)pk;
    pk += (int64_t)L.Kpad8 * L.Mblk * 32;
    if (!filter) {
      L.pb_off = (int)pk;
      pk += (int64_t)L.Mpad8 * L.Kblk * 32;
    } else {
      L.pb_off = -1;
    }
    L.pbias_off = (int)pk;
    pk += (mu ? 2 : 1) * L.Mblk * 32;  // LT_GABOR_MU: [gamma | |mu_j|^2]
    L.rf_off = L.rb_off = -1;
  }
  nd.slab_loss_off = goff;
  nd.slab_floats = round_up(goff + 4, 64);
  // stash: [f | l cos u | h] per stage, encoder features, |x|^2 [TL]
  nd.w2_off = nd.w2_bias_off = -1;
  nd.save_floats_per_tile = 3 * nd.mfn_stages * NB * 32 * TL + nd.L[0].Kblk * 32 * TL + TL;  // Kblk*32 >= 2 E'
  p->packed_floats = pk;
  *out = p;
  return INR_OK;
}

extern "C" {

int inr_abi_version(void) { return INR_ABI_VERSION; }

int inr_last_error(char* buf, size_t cap) {
  const size_t n = strlen(g_err);
  if (buf != nullptr && cap > 0) {
    const size_t c = n < cap - 1 ? n : cap - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)n;
}

int inr_plan_create(const inr_net_desc* d, inr_plan** out) {
  if (d == nullptr || out == nullptr) return fail(INR_ERR_INVALID, "inr_plan_create: null argument");
  *out = nullptr;
  if (d->kind == INR_KIND_FOURIER || d->kind == INR_KIND_MSFOURIER || d->kind == INR_KIND_MSBOUNDED ||
      d->kind == INR_KIND_GABOR || d->kind == INR_KIND_KGABOR)
    return create_mfn_plan(d, out);
  if (d->kind != INR_KIND_SIREN && d->kind != INR_KIND_FFN && d->kind != INR_KIND_WIRE && d->kind != INR_KIND_WIRE2D)
    return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: kind %d has no kernel yet", d->kind);
  const bool wire2d = d->kind == INR_KIND_WIRE2D;
  const bool wire = d->kind == INR_KIND_WIRE || wire2d;
  // number of Linear layers: SIREN/FFN network_depth counts all of them (networks.py:114-117);
  // WIRE's counts the hidden complex layers only, total = depth + 2 (networks.py:234-250)
  const int D = wire ? d->depth + 2 : d->depth;
  if (D < 2 || D > INR_MAX_LAYERS)
    return fail(INR_ERR_INVALID, "inr_plan_create: depth %d gives %d layers, outside [2,%d]", d->depth, D,
                INR_MAX_LAYERS);
  if (d->out_features < 1 || d->out_features > 4)
    return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: out_features %d outside [1,4]", d->out_features);
  if (d->in_features < 1) return fail(INR_ERR_INVALID, "inr_plan_create: in_features %d", d->in_features);
  if (d->width < 1) return fail(INR_ERR_INVALID, "inr_plan_create: width %d", d->width);
  // rows of the hidden activations as the kernel sees them: complex features are (Re, Im) row pairs
  const int hid = wire ? 2 * d->width : d->width;
  const int need = (hid + 31) / 32;
  const int NB = wire2d ? pick_nb(need, {2, 4, 8, 16})
                        : (wire ? pick_nb(need, {2, 4, 8, 12}) : pick_nb(need, {1, 2, 4, 8, 16}));
  int NW;
  if (wire) {
    if (NB < 0)
      return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: WIRE with %d complex hidden features (kernels are built for "
                  "up to 192 = 384 interleaved rows, network_width 256 gives 181; WIRE2D: up to 256)", d->width);
    NW = (NB == 12 || NB == 16) ? 2 : 4;  // 12 / 16 blocks: 64-coordinate tiles, two waves per coordinate group
    if (wire2d && 2 * D - 1 > INR_MAX_LAYERS)
      return fail(INR_ERR_INVALID, "inr_plan_create: WIRE2D depth %d", d->depth);
    if (d->input != INR_INPUT_X)
      return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: WIRE takes raw coordinates (input must be INR_INPUT_X)");
    if (d->last_act == INR_ACT_CTANH) {
      if (!wire2d || d->out_features > 2)
        return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: INR_ACT_CTANH is WIRE2D's last_tanh, out_features <= 2");
    } else if (d->last_act != INR_ACT_ID) {
      return fail(INR_ERR_INVALID, "inr_plan_create: WIRE's output is linear (or INR_ACT_CTANH for WIRE2D)");
    }
  } else {
    if (NB < 0)
      return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: width %d (kernels are built for widths 1..512)", d->width);
    NW = NB == 16 ? 2 : 4;
  }
  if (d->input == INR_INPUT_GAUSS) {
    if (d->enc_size < 8 || (d->enc_size % 8) != 0)
      return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: enc_size %d must be a positive multiple of 8", d->enc_size);
    if (d->in_features != 2 * d->enc_size)
      return fail(INR_ERR_INVALID, "inr_plan_create: in_features %d != 2*enc_size %d", d->in_features,
                  2 * d->enc_size);
  } else if (d->input != INR_INPUT_X) {
    return fail(INR_ERR_INVALID, "inr_plan_create: input mode %d", d->input);
  }
  if ((d->last_act < INR_ACT_ID || d->last_act > INR_ACT_SIGMOID) && !(wire2d && d->last_act == INR_ACT_CTANH))
    return fail(INR_ERR_INVALID, "inr_plan_create: last_act %d", d->last_act);
  if (d->precision != INR_PRECISION_F32 && d->precision != INR_PRECISION_BF16)
    return fail(INR_ERR_INVALID, "inr_plan_create: precision %d", d->precision);
  if (d->precision == INR_PRECISION_BF16 &&
      (d->kind != INR_KIND_SIREN || d->input != INR_INPUT_GAUSS || NB != 8 || (d->enc_size % 32) != 0 || D < 3 || D > 8 ||
       d->enc_size > 1024))
    return fail(INR_ERR_UNSUPPORTED, "inr_plan_create: the bf16 path is built for SIREN with the fused gauss encoder, "
                "hidden width 129..256, 3 to 8 layers and an encoder size that is a multiple of 32 up to 1024 (got kind %d, input "
                "%d, width %d, depth %d, enc_size %d)", d->kind, d->input, d->width, d->depth, d->enc_size);

  inr_plan* p = new (std::nothrow) inr_plan();
  if (p == nullptr) return fail(INR_ERR_INVALID, "inr_plan_create: out of host memory");
  p->desc = *d;
  NetDesc& nd = p->nd;
  memset(&nd, 0, sizeof(nd));
  nd.D = D;
  nd.NB = NB;
  nd.NW = NW;
  nd.bf16 = d->precision == INR_PRECISION_BF16 ? 1 : 0;
  nd.hact = wire2d ? ACT_GABOR2D : (wire ? ACT_GABOR : (d->kind == INR_KIND_SIREN ? ACT_SIN : ACT_RELU));
  nd.ND = wire2d ? 2 * D - 1 : D;
  nd.orth0 = D;
  nd.last_act = d->last_act;
  nd.input = d->input;
  nd.E = d->enc_size;
  nd.out_f = d->out_features;
  nd.w0 = d->w0;
  const int TL = 32 * NW;
  int poff = 0, goff = 0;
  int64_t pk = 0;
  // descriptors in flat-parameter order: WIRE2D interleaves linear / scale_orth of each layer (wire2d.py:40-47)
  for (int t = 0; t < nd.ND; ++t) {
    const int l = wire2d ? (t == nd.ND - 1 ? D - 1 : t / 2) : t;
    const bool orth = wire2d && t != nd.ND - 1 && (t & 1);
    LayerDesc& L = nd.L[orth ? nd.orth0 + l : l];
    const bool first = l == 0, last = l == D - 1;
    const bool ctanh_last = last && d->last_act == INR_ACT_CTANH;  // complex output kept: (Re, Im) row pairs
    L.K = first ? d->in_features : hid;
    L.M = last ? (ctanh_last ? 2 * d->out_features : d->out_features) : hid;
    // hidden-to-hidden products run over all NB*32 image rows (padding rows carry zero weights)
    L.Kpad8 = first ? round_up(L.K, 8) : NB * 32;
    L.Kblk = first ? (L.K + 31) / 32 : NB;  // hidden images always span all NB blocks (zero padding)
    L.Mblk = last ? (L.M + 31) / 32 : NB;
    L.Mpad8 = last ? round_up(L.M, 8) : NB * 32;
    if (!wire) {
      L.ltype = LT_REAL;
      L.wn = L.M * L.K;
      L.bn = L.M;
      L.omega = d->w0;
      L.s0 = 0.f;
    } else if (first) {
      L.ltype = LT_WIRE_FIRST;  // real weights on real coordinates (networks.py:185-188)
      L.wn = d->width * L.K;
      L.bn = d->width;
      L.omega = d->first_omega_0;
      L.s0 = d->scale_0;
    } else if (!last) {
      L.ltype = LT_WIRE_HIDDEN;
      L.wn = d->width * d->width * 2;
      L.bn = d->width * 2;
      L.omega = d->hidden_omega_0;
      L.s0 = d->scale_0;
    } else {
      // complex Linear, output.real (networks.py:247-258): only the real rows exist -- unless a complex Tanh sits
      // before .real (WIRE2D last_tanh), which needs the imaginary rows too: the hidden-layer mapping
      L.ltype = ctanh_last ? LT_WIRE_HIDDEN : LT_WIRE_LAST;
      L.wn = d->out_features * d->width * 2;
      L.bn = d->out_features * 2;
    }
    L.w_off = poff;
    poff += L.wn;
    L.b_off = poff;
    poff += L.bn;
    // gradient slab: hidden layers keep all NB*32 rows (the dW pass stores whole row blocks without a
    // bounds test; padding rows receive exact zeros and are never read back)
    L.gw_off = goff;
    goff += (last ? L.M : NB * 32) * L.K;
    L.gb_off = goff;
    goff += last ? L.M : NB * 32;
    if (nd.bf16) {  // bf16 plans keep one image set only: the panel stream behind the layers (below)
      L.pf_off = L.pb_off = L.pbias_off = -1;
    } else {
      L.pf_off = (int)pk;
      pk += (int64_t)L.Kpad8 * L.Mblk * 32;  // (Kpad8/8 groups) x Mblk x 64 lanes x 4
      if (l >= 1) {
        L.pb_off = (int)pk;
        pk += (int64_t)L.Mpad8 * L.Kblk * 32;
      } else {
        L.pb_off = -1;
      }
      L.pbias_off = (int)pk;
      pk += L.Mblk * 32;
    }
    L.live = 1;
    L.korder = (first && d->input == INR_INPUT_GAUSS) ? 1 : 0;
  }
  nd.P = poff;
  nd.slab_loss_off = goff;
  nd.slab_floats = round_up(goff + 4, 64);
  const int ns = wire2d ? 7 : (wire ? 3 : 2);
  nd.save_floats_per_tile = ns * (D - 1) * NB * 32 * TL + 4 * TL +
                            (d->input == INR_INPUT_GAUSS ? nd.L[0].Kblk * 32 * TL : 0) +
                            (wire2d ? NB * 32 * TL : 0);  // WIRE2D: copy of a layer's output gradient
  nd.w2_off = nd.w2_bias_off = -1;
  // Row-split fused step (inr_mlp_rs_impl.h): SIREN / FFN behind the fused gauss encoder, hidden width 129..256, encoder
  // size a multiple of 32 that leaves room in LDS.  Such plans carry a second set of fragment images (16x16x4 MFMA
  // operands); forward / backward calls keep inr_mlp_kernel and its images.
  for (int t = 0; t < INR_MAX_LAYERS; ++t) nd.L[t].rf_off = nd.L[t].rb_off = -1;
  if (!nd.bf16 && !wire && NB == 8 && d->input == INR_INPUT_GAUSS && (d->enc_size % 32) == 0 && d->enc_size <= 512) {
    nd.rs = 1;
    for (int l = 0; l <= D - 2; ++l) {
      nd.L[l].rf_off = (int)pk;
      pk += (int64_t)256 * (l == 0 ? 2 * d->enc_size : 256);
      if (l >= 1) {
        nd.L[l].rb_off = (int)pk;
        pk += (int64_t)256 * 256;
      }
    }
  }
  if (nd.bf16) {
    // the images of the bf16 plans: the "weight panels in LDS" stream (inr_w2.h) + fp32 biases; 8-bit stash
    nd.w2_off = (int)pk;  // (0: 16-byte aligned, the panels are read by 16-byte LDS-DMA pieces)
    pk += (int64_t)w2_np(D, d->enc_size) * W2_PANEL_FLOATS;
    nd.w2_bias_off = (int)pk;
    pk += (int64_t)D * 256;
    nd.save_floats_per_tile = w2_stash_dwords(D);
    // (the gradient-scale state is allocated by the first call that needs it, on that call's device: creating and sizing
    // a plan touches no GPU -- tests/test_host.py sizes bf16 workspaces on the CPU)
  }
  p->packed_floats = pk;
  p->gemm_one_class = getenv("INR_GEMM_ONE_CLASS") != nullptr;
  if (const char* e = getenv("INR_GEMM_ENC_COST")) p->gemm_enc_cost = std::max(1.0, atof(e));
  *out = p;
  return INR_OK;
}

int inr_plan_destroy(inr_plan* plan) {
  if (plan != nullptr && plan->side != nullptr) (void)hipStreamDestroy(plan->side);
  if (plan != nullptr && plan->fork != nullptr) (void)hipEventDestroy(plan->fork);
  if (plan != nullptr && plan->join != nullptr) (void)hipEventDestroy(plan->join);
  if (plan != nullptr && plan->dz_state != nullptr) (void)hipFree(plan->dz_state);
  delete plan;
  return INR_OK;
}

// items, chunking and the flat-gradient range [lo, hi) of the layers the batch-level dW GEMM covers; false: the plan
// keeps its in-kernel dW passes
static bool dw_gemm_setup(const inr_plan* plan, int64_t nt, inr::DwGemmArgs* g, inr::SlabSplit* split) {
  const NetDesc& nd = plan->nd;
  memset(g, 0, sizeof(*g));
  split->lo = split->hi = split->n2 = 0;
  split->mask = 0;
  const bool g2d = nd.hact == ACT_GABOR2D;
  if (nd.bf16) return false;
  if (nd.mfn_n != 0) {
    // 512-wide filter networks (inr_mfn_wide_impl.h):  F_t: g_u_t (stash slot 3t+1) x encoder features, t < S;
    // L_{i-1}: g_l_i (slot 3i) x h_{i-1} (slot 3(i-1)+2), 1 <= i < S.  Their flat ranges interleave with heads and
    // Gabor centres, so the reduction learns the covered layers as a bit mask.
    if (nd.NB != 16) return false;
    const int S = nd.mfn_stages, n = nd.mfn_n, HSZ = 16 * 32 * 64;
    if (2 * S - 1 > INR_DWG_MAX_ITEMS) return false;
    g->TL = 64, g->WB = 4;
    g->save_floats_per_tile = nd.save_floats_per_tile, g->slab_floats = nd.slab_floats, g->n_tiles = (int)nt;
    int k = 0;
    unsigned mask = 0;
    for (int t = 0; t < 2 * S - 1; ++t) {
      const int i = t - S + 1, l = t < S ? t : n + 1 + (i - 1);
      const LayerDesc& L = nd.L[l];
      inr::DwGemmItem& it = g->it[k++];
      it.g_off = t < S ? (3 * t + 1) * HSZ : (3 * i) * HSZ;
      it.h_off = t < S ? 3 * S * HSZ : (3 * (i - 1) + 2) * HSZ;
      it.gw_off = L.gw_off, it.gb_off = L.gb_off, it.Mblk = 16, it.Kblk = L.Kblk, it.K = L.K;
      mask |= 1u << l;
    }
    g->n_items = k;
    const int bpc = inr::dw_gemm_units(*g);
    const int target = std::max(1, 256 / std::max(1, bpc));
    g->tiles_per_chunk = (int)((nt + target - 1) / target);
    g->n_chunks = (int)((nt + g->tiles_per_chunk - 1) / g->tiles_per_chunk);
    split->mask = mask;
    split->n2 = g->n_chunks;
    return true;
  }
  // the plain MLP kernels, fp32: 256-row tensors (one wave per coordinate group) and the two-waves-per-group shapes
  if (!(nd.NB == 8 && !g2d) && nd.NB != 12 && nd.NB != 16) return false;
  const int TL = 32 * nd.NW, HSZ = nd.NB * 32 * TL, D = nd.D;
  const int NS = g2d ? 7 : (nd.hact == ACT_GABOR ? 3 : 2);
  g->TL = TL;
  g->WB = nd.NB == 12 ? 3 : 4;
  g->save_floats_per_tile = nd.save_floats_per_tile;
  g->slab_floats = nd.slab_floats;
  g->n_tiles = (int)nt;
  int k = 0, covered = 0, lo = nd.P, hi = 0;
  auto add = [&](const LayerDesc& L, int g_off, int h_off) {
    inr::DwGemmItem& it = g->it[k++];
    it.g_off = g_off, it.h_off = h_off;
    it.gw_off = L.gw_off, it.gb_off = L.gb_off;
    it.Mblk = nd.NB, it.Kblk = L.Kblk, it.K = L.K;
    covered += L.wn + L.bn;
    lo = std::min(lo, std::min(L.w_off, L.b_off));
    hi = std::max(hi, std::max(L.w_off + L.wn, L.b_off + L.bn));
  };
  if (2 * D > INR_DWG_MAX_ITEMS) return false;
  if (nd.input == IN_GAUSS) add(nd.L[0], 1 * HSZ, NS * (D - 1) * HSZ + 4 * TL);  // dZ_0 x encoder features
  for (int l = 1; l <= D - 2; ++l) {
    add(nd.L[l], (NS * l + 1) * HSZ, NS * (l - 1) * HSZ);                  // dZ_l x h_{l-1}
    if (g2d) add(nd.L[nd.orth0 + l], (NS * l + 3) * HSZ, NS * (l - 1) * HSZ);  // WIRE2D: dZ_orth,l x h_{l-1}
  }
  if (k == 0 || covered != hi - lo) return false;  // the covered layers must be one contiguous flat range
  g->n_items = k;
  // about one workgroup per CU: the accumulators then stay in registers over as many tiles as possible
  auto chunking = [&]() {
    const int bpc = inr::dw_gemm_units(*g);
    const int target = std::max(1, 256 / std::max(1, bpc));
    g->tiles_per_chunk = (int)((nt + target - 1) / target);
    g->n_chunks = (int)((nt + g->tiles_per_chunk - 1) / g->tiles_per_chunk);
  };
  chunking();
  // short chunks (the graded 25 000 rows: K = 512 coordinates per 256 x 256 tile): half-height tiles over twice the K --
  // half as many slabs to store at the end of the launch and to reduce (inr_dw_gemm.hip)
  if (TL == 128 && g->WB == 4 && nt > 1 && (int64_t)g->tiles_per_chunk * TL < 1024) {
    g->WBM = 2;
    chunking();
  }
  split->lo = lo;
  split->hi = hi;
  split->n2 = g->n_chunks;
  return true;
}

// bf16 plans with the "weights in LDS" fused kernel: every weight gradient comes from inr_dw_gemm_bf16.hip
static bool w2_plan(const inr_plan* plan) { return plan->nd.bf16 != 0; }

constexpr double kEncCost = 1.5;
static void dw_gemm_bf16_setup(const inr_plan* plan, int64_t nt, inr::DwGemmBf16Args* g) {
  const NetDesc& nd = plan->nd;
  memset(g, 0, sizeof(*g));
  const int D = nd.D;
  int k = 0;
  for (int n0 = 0; n0 < nd.E; n0 += 128) {  // first layer: B = encoder features, 128 frequencies (sine + cosine) a unit
    inr::DwGemmBf16Unit& u = g->unit[k++];
    u.dz_off = w2_stash_G(0, D), u.z_off = -1;
    u.gw_off = nd.L[0].gw_off, u.gb_off = nd.L[0].gb_off, u.M = 256, u.K = nd.L[0].K, u.n0 = n0;
  }
  for (int l = 1; l <= D - 1; ++l) {
    inr::DwGemmBf16Unit& u = g->unit[k++];
    const bool last = l == D - 1;
    u.dz_off = last ? w2_stash_dzl(D) : w2_stash_G(l, D);
    u.z_off = w2_stash_P(l - 1);
    u.gw_off = nd.L[l].gw_off, u.gb_off = nd.L[l].gb_off;
    u.M = last ? nd.L[l].M : 256, u.K = nd.L[l].K, u.n0 = 0;
  }
  g->n_units = k;
  g->TL = W2_TL, g->E = nd.E;
  g->save_floats_per_tile = nd.save_floats_per_tile, g->slab_floats = nd.slab_floats, g->n_tiles = (int)nt;
  // About one workgroup per CU, in two classes (inr_dw_gemm_bf16.h): a first-layer unit costs kEncCost x a hidden unit per
  // tile (65 536 rows, depth 5: 68 us against 54 when each kind runs alone; factors 1.0 / 1.15 / 1.33 / 1.5 / 1.7 measured 72.6 / 70.6 / 69.3 / 66.8 / 67.9 us), so it gets that many times the
  // chunks.  Needs the first layer's weight and bias gradients to be one aligned run of the flat layout (the reduction
  // sums that run over another number of slabs); otherwise one class.
  const int n_enc = (nd.E + 127) / 128, others = k - n_enc;
  const LayerDesc& L0 = nd.L[0];
  const bool run0 = L0.gb_off == L0.gw_off + L0.M * L0.K && (L0.gw_off & 3) == 0 && ((L0.gb_off + L0.M) & 3) == 0;
  auto chunks = [&](int target, int* tpc, int* n) {
    target = std::max(1, target);
    *tpc = (int)((nt + target - 1) / target);
    *n = (int)((nt + *tpc - 1) / *tpc);
  };
  if (run0 && !plan->gemm_one_class) {
    const double cost = plan->gemm_enc_cost > 0.0 ? plan->gemm_enc_cost : kEncCost;
    const double per = 256.0 / (cost * n_enc + others);  // chunks of a non-first-layer unit
    g->n_enc_units = n_enc;
    chunks((int)(per * cost), &g->tiles_per_chunk_enc, &g->n_chunks_enc);
    chunks((256 - n_enc * g->n_chunks_enc) / others, &g->tiles_per_chunk, &g->n_chunks);
  } else {
    g->n_enc_units = n_enc;
    chunks(256 / k, &g->tiles_per_chunk, &g->n_chunks);
    g->tiles_per_chunk_enc = g->tiles_per_chunk, g->n_chunks_enc = g->n_chunks;
  }
}

// chunk slabs of the bf16 GEMM, and how the reduction reads them
static int dw_gemm_bf16_slabs(const inr::DwGemmBf16Args& g) { return std::max(g.n_chunks, g.n_chunks_enc); }
static inr::SlabSplit dw_gemm_bf16_split(const inr_plan* plan, const inr::DwGemmBf16Args& g) {
  inr::SlabSplit split{0, (plan->nd.P + 3) & ~3, g.n_chunks, 0};  // (a multiple of 4: the fast reduction works on float4)
  if (g.n_chunks_enc != g.n_chunks) {
    const LayerDesc& L0 = plan->nd.L[0];
    split.lo3 = L0.gw_off, split.hi3 = L0.gb_off + L0.M, split.n3 = g.n_chunks_enc;
  }
  return split;
}

static bool dw_gemm_plan(const inr_plan* plan) {
  inr::DwGemmArgs g;
  inr::SlabSplit split;
  return dw_gemm_setup(plan, 1, &g, &split);
}

// fused steps of these plans stash per TILE (n_tiles slots): a batch-level GEMM reads the whole batch's stash
static bool step_save_by_tile(const inr_plan* plan) { return dw_gemm_plan(plan) || w2_plan(plan); }

// ---------------------------------------------------------------------------------------------
// How a fused step of a batch-GEMM plan is launched.  With more tiles than workgroups the persistent grid runs whole
// rounds and then a partial one, during which `idle` = n_blocks - (n_tiles mod n_blocks) CUs have nothing to do (WIRE at
// 25 000 rows: 391 tiles of 64 coordinates = 256 + 135; the multiscale config: 1563 = 6 x 256 + 27) -- while the
// weight-gradient GEMM of the tiles already finished only needs their stash.  Split step:
//   main stream:  fused kernel on tiles [0, full)  ->  fused kernel on tiles [full, nt), `rem` workgroups, accumulating
//                 into the slabs of workgroups 0..rem-1  ->  (join)  ->  GEMM part B: tiles [tA, nt)  ->  reduction
//   side stream:  (after the first kernel)  GEMM part A: tiles [0, tA), at most `idle` workgroups
// Part A is sized to end with the partial round: a tile costs the GEMM about kGemmTileShare of the fused kernel's time
// for it on one CU (dW is half of forward + dX, at a slightly better MFMA rate).  Chunk slabs of A, then of B, follow
// the fused kernel's; every sum keeps a fixed order (deterministic), though not the order of the unsplit launch.
// INR_OVERLAP=0 in the environment turns the split off.
// ---------------------------------------------------------------------------------------------
constexpr double kGemmTileShare = 0.4;

struct StepSchedule {
  bool split;
  int64_t full, rem, tA;
  inr::DwGemmArgs gA, gB;  // (gB alone when !split)
  inr::SlabSplit red;      // for the reduction: n2 = all chunk slabs
};

static bool overlap_enabled() {  // (read per call: a test compares the two schedules in one process)
  const char* e = getenv("INR_OVERLAP");
  return !(e != nullptr && e[0] == '0');
}

// chunks of tiles [tile0, tile1) for about `max_wgs` workgroups (same rules as dw_gemm_setup: half-height tiles for short chunks)
static void rechunk(inr::DwGemmArgs& g, int64_t tile0, int64_t tile1, int max_wgs) {
  const int64_t n = tile1 - tile0;
  g.tile0 = (int)tile0, g.n_tiles = (int)tile1;
  auto chunking = [&]() {
    const int units = std::max(1, inr::dw_gemm_units(g));
    const int target = std::max(1, max_wgs / units);
    g.tiles_per_chunk = (int)((n + target - 1) / target);
    g.n_chunks = (int)((n + g.tiles_per_chunk - 1) / g.tiles_per_chunk);
  };
  g.WBM = 0;
  chunking();
  if (g.TL == 128 && g.WB == 4 && n > 1 && (int64_t)g.tiles_per_chunk * g.TL < 1024) {
    g.WBM = 2;
    chunking();
  }
}

static bool step_schedule(const inr_plan* plan, int64_t nt, int64_t nb, StepSchedule* sc) {
  sc->split = false;
  sc->full = nt, sc->rem = 0, sc->tA = 0;
  if (!dw_gemm_setup(plan, nt, &sc->gB, &sc->red)) return false;
  sc->gA = sc->gB;
  sc->gA.n_chunks = 0;
  if (!overlap_enabled() || nt <= nb || nt % nb == 0) return true;
  const int64_t rem = nt % nb, full = nt - rem, idle = nb - rem;
  {
    inr::DwGemmArgs probe = sc->gB;
    probe.WBM = 0;
    if (idle < inr::dw_gemm_units(probe)) return true;  // not even one chunk's workgroups fit beside the partial round
  }
  int64_t tA = (int64_t)(0.9 * (double)idle / kGemmTileShare);
  if (tA > full) tA = full;
  if (tA < nt / 16 || tA < 1) return true;  // nothing worth a second launch
  rechunk(sc->gA, 0, tA, (int)idle);
  rechunk(sc->gB, tA, nt, 256);
  sc->split = true;
  sc->full = full, sc->rem = rem, sc->tA = tA;
  sc->red.n2 = sc->gA.n_chunks + sc->gB.n_chunks;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Row-split fused step: how the 16-coordinate column blocks of a batch are dealt to workgroups.  The stash is read by the
// batch GEMM in whole 128-coordinate slots, so all 8 blocks of every slot are computed (rows past B masked).  `grid`
// workgroups run `rounds` tiles each; tile t has `hi` blocks if t < x, else `lo`; a tile's block count is the kernel's
// NCB, or even (the kernel pairs column blocks: inr_mlp_rs_impl.h rs_active).  Rounds are chosen by cost: a round costs
// its widest tile plus about one block of fixed work (epilogues, barriers, the weight stream's start).
// ---------------------------------------------------------------------------------------------
struct RsSchedule {
  int grid, rounds, ncb, hi, lo, x;
};
// Tiles are at most 7 column blocks wide: the kernel keeps 16 NCB accumulators and 16 NCB act' values per lane in AGPRs,
// and at NCB = 8 that is all 256 of them -- the compiler's own AGPR copies then push act' into scratch (measured: the
// forward GEMMs at 88-98 k cycles instead of 71 k).
constexpr int kRsMaxNcb = 7;
static RsSchedule rs_schedule(int64_t nt) {
  const int64_t nblk = 8 * nt;
  RsSchedule s;
  s.grid = (int)std::min<int64_t>(kMaxBlocks, nblk);
  double best = 1e30;
  s.rounds = 1, s.ncb = kRsMaxNcb;
  const int64_t rmax = nblk / s.grid + 1;
  for (int64_t R = 1; R <= rmax; ++R) {
    const int64_t T = s.grid * R, a = nblk / T, rem = nblk % T, ncb = rem ? a + 1 : a;
    if (ncb > kRsMaxNcb || ncb < 1) continue;
    // the busiest workgroup's blocks (workgroup 0: the `hi` tiles come first) + a block's worth of fixed work per round
    const int64_t lo = rem == 0 ? a : (a % 2 == 0 ? a : a - 1), x = rem == 0 ? T : (a % 2 == 0 ? rem : (nblk - (a - 1) * T) / 2);
    const int64_t nhi = std::min<int64_t>(R, (x + s.grid - 1) / s.grid);
    const double cost = (double)(nhi * ncb + (R - nhi) * lo) + 0.9 * (double)R;
    if (cost < best - 1e-9) best = cost, s.rounds = (int)R, s.ncb = (int)ncb;
  }
  const int64_t T = (int64_t)s.grid * s.rounds, a = nblk / T, rem = nblk % T;
  if (rem == 0) {
    s.hi = s.lo = (int)a, s.x = (int)T;
  } else if (a % 2 == 0) {  // NCB = a + 1 odd: full tiles and even ones
    s.hi = (int)a + 1, s.lo = (int)a, s.x = (int)rem;
  } else {                  // NCB = a + 1 even: the others give up a pair
    s.hi = (int)a + 1, s.lo = (int)a - 1, s.x = (int)((nblk - (a - 1) * T) / 2);
  }
  return s;
}

static bool rs_plan(const inr_plan* plan) { return plan->nd.rs != 0; }
// Which fused kernel runs a batch of nt 128-coordinate slots?  The row-split kernel, unless inr_mlp_kernel's rounds of 256
// tiles are (all but) full: then both do the same MFMA work and the row-split kernel only adds a round (65 536 rows:
// 6 + 6 + 4 column blocks per workgroup, 649 us against 629 us; 25 000 rows: 266 us against 316 us).
// INR_RS=0 / 1 in the environment forces one or the other (read per call: tests compare the two in one process).
static bool rs_enabled(int64_t nt) {
  const char* e = getenv("INR_RS");
  if (e != nullptr && e[0] == '0') return false;
  if (e != nullptr && e[0] == '1') return true;
  const int64_t rounds = (nt + kMaxBlocks - 1) / kMaxBlocks;
  return (double)nt < 0.97 * (double)(rounds * kMaxBlocks);
}

static int launch_rs(const inr_plan* plan, const LossDesc& ld, inr::MlpArgs a, int64_t nt, const RsSchedule& sc,
                     hipStream_t st) {
  a.n_tiles = (int)nt;
  a.rs_hi = sc.hi, a.rs_lo = sc.lo, a.rs_x = sc.x, a.rs_rounds = sc.rounds;
  a.tile0 = 0, a.accumulate = 0;
  hipError_t e;
  const NetDesc& nd = plan->nd;
  switch (sc.ncb) {
    case 1: e = inr::launch_mlp_rs_n1(nd, ld, a, sc.grid, st); break;
    case 2: e = inr::launch_mlp_rs_n2(nd, ld, a, sc.grid, st); break;
    case 3: e = inr::launch_mlp_rs_n3(nd, ld, a, sc.grid, st); break;
    case 4: e = inr::launch_mlp_rs_n4(nd, ld, a, sc.grid, st); break;
    case 5: e = inr::launch_mlp_rs_n5(nd, ld, a, sc.grid, st); break;
    case 6: e = inr::launch_mlp_rs_n6(nd, ld, a, sc.grid, st); break;
    default: e = inr::launch_mlp_rs_n7(nd, ld, a, sc.grid, st); break;
  }
  if (e != hipSuccess) return hip_fail(e, "inr row-split kernel launch");
  return INR_OK;
}

static hipStream_t side_stream(const inr_plan* plan) {
  std::lock_guard<std::mutex> lock(plan->side_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (plan->side != nullptr && plan->side_dev != dev) {
    (void)hipStreamDestroy(plan->side);
    if (plan->fork != nullptr) (void)hipEventDestroy(plan->fork);
    if (plan->join != nullptr) (void)hipEventDestroy(plan->join);
    plan->side = nullptr;
    plan->fork = plan->join = nullptr;
  }
  if (plan->side == nullptr) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&plan->side, hipStreamNonBlocking, least) != hipSuccess) plan->side = nullptr;
    if (plan->side != nullptr && (hipEventCreateWithFlags(&plan->fork, hipEventDisableTiming) != hipSuccess ||
                                  hipEventCreateWithFlags(&plan->join, hipEventDisableTiming) != hipSuccess)) {
      if (plan->fork != nullptr) (void)hipEventDestroy(plan->fork);
      (void)hipStreamDestroy(plan->side);
      plan->side = nullptr;
      plan->fork = plan->join = nullptr;
    }
    plan->side_dev = dev;
  }
  return plan->side;
}

// a call's scratch against what the plan needs: `save_slots` stash slots (0: none), `n_slabs` slabs (0: none)
static int check_ws(const inr_plan* plan, const inr_workspace* ws, int64_t save_slots, int64_t n_slabs,
                    const char* who) {
  if (ws == nullptr) return fail(INR_ERR_INVALID, "%s: null workspace", who);
  const int64_t need_save = save_slots * (int64_t)plan->nd.save_floats_per_tile;
  const int64_t need_slabs = n_slabs * (int64_t)plan->nd.slab_floats;
  if (save_slots > 0 && (ws->save == nullptr || ws->save_floats < need_save))
    return fail(INR_ERR_INVALID, "%s: stash of %lld floats, the call needs %lld (%lld slots of %d; see "
                "inr_plan_workspace)", who, (long long)(ws->save == nullptr ? 0 : ws->save_floats),
                (long long)need_save, (long long)save_slots, plan->nd.save_floats_per_tile);
  if (n_slabs > 0 && (ws->slabs == nullptr || ws->slab_floats < need_slabs))
    return fail(INR_ERR_INVALID, "%s: %lld slab floats, the call needs %lld (%lld slabs of %d; see "
                "inr_plan_workspace)", who, (long long)(ws->slabs == nullptr ? 0 : ws->slab_floats),
                (long long)need_slabs, (long long)n_slabs, plan->nd.slab_floats);
  return INR_OK;
}

int inr_plan_sizes(const inr_plan* plan, inr_sizes* out) {
  if (plan == nullptr || out == nullptr) return fail(INR_ERR_INVALID, "inr_plan_sizes: null argument");
  out->n_params = plan->nd.P;
  out->packed_floats = plan->packed_floats;
  out->tile_rows = 32 * plan->nd.NW;
  out->save_bytes_per_tile = (int64_t)plan->nd.save_floats_per_tile * 4;
  out->max_blocks = kMaxBlocks;
  out->slab_floats = plan->nd.slab_floats;
  out->step_save_by_tile = step_save_by_tile(plan) ? 1 : 0;
  return INR_OK;
}

int inr_plan_workspace(const inr_plan* plan, int64_t B, int64_t* step_save_slots, int64_t* n_slabs) {
  if (plan == nullptr || step_save_slots == nullptr || n_slabs == nullptr)
    return fail(INR_ERR_INVALID, "inr_plan_workspace: null argument");
  int64_t nt, nb;
  int rc = inr_plan_launch_dims(plan, B, &nt, &nb);
  if (rc != INR_OK) return rc;
  *step_save_slots = step_save_by_tile(plan) ? nt : nb;
  *n_slabs = nb;
  if (w2_plan(plan)) {  // (the unfused backward of these plans needs nb slabs only: covered)
    inr::DwGemmBf16Args g;
    dw_gemm_bf16_setup(plan, nt, &g);
    *n_slabs = nb + dw_gemm_bf16_slabs(g);
  }
  if (dw_gemm_plan(plan)) {
    inr::DwGemmArgs g;
    inr::SlabSplit split;
    dw_gemm_setup(plan, nt, &g, &split);
    StepSchedule sc;
    step_schedule(plan, nt, nb, &sc);  // (a split step has its own chunking; the unfused backward keeps the plain one)
    // (row-split fused steps run rs_schedule's grid -- more workgroups than tiles while the batch is under 256 slots)
    const int64_t nb_rs = rs_plan(plan) ? std::max<int64_t>(nb, rs_schedule(nt).grid) : nb;
    *n_slabs = nb_rs + std::max(g.n_chunks, sc.red.n2);
  }
  return INR_OK;
}

int inr_plan_step_info(const inr_plan* plan, int64_t B, inr_step_info* out) {
  if (plan == nullptr || out == nullptr) return fail(INR_ERR_INVALID, "inr_plan_step_info: null argument");
  int64_t nt, nb;
  const int rc = inr_plan_launch_dims(plan, B, &nt, &nb);
  if (rc != INR_OK) return rc;
  memset(out, 0, sizeof(*out));
  if (rs_plan(plan) && dw_gemm_plan(plan) && rs_enabled(nt)) {
    const RsSchedule s = rs_schedule(nt);
    out->row_split = 1, out->ncb = s.ncb, out->grid = s.grid, out->rounds = s.rounds;
    out->hi = s.hi, out->lo = s.lo, out->n_hi = s.x;
  } else {
    out->grid = (int32_t)nb, out->rounds = (int32_t)((nt + nb - 1) / nb);
  }
  return INR_OK;
}

int inr_plan_grad_scale_state(const inr_plan* plan, float* host_out, void* stream) {
  if (plan == nullptr || host_out == nullptr) return fail(INR_ERR_INVALID, "inr_plan_grad_scale_state: null argument");
  if (!plan->nd.bf16) return fail(INR_ERR_INVALID, "inr_plan_grad_scale_state: not an INR_PRECISION_BF16 plan");
  if (dz_state_alloc(plan) == nullptr)  // (before the first step: the initial state)
    return fail(INR_ERR_HIP, "inr_plan_grad_scale_state: no gradient-scale state on this device");
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipMemcpy(host_out, plan->dz_state, W2_STATE_FLOATS * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "inr_plan_grad_scale_state");
  return INR_OK;
}

int inr_plan_launch_dims(const inr_plan* plan, int64_t B, int64_t* n_tiles, int64_t* n_blocks) {
  if (plan == nullptr || n_tiles == nullptr || n_blocks == nullptr)
    return fail(INR_ERR_INVALID, "inr_plan_launch_dims: null argument");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_plan_launch_dims: B = %lld", (long long)B);
  const int tl = 32 * plan->nd.NW;
  *n_tiles = (B + tl - 1) / tl;
  *n_blocks = *n_tiles < kMaxBlocks ? *n_tiles : kMaxBlocks;
  if (plan->nd.bf16) {  // the bf16 kernel's workgroups take two 128-coordinate tiles each
    const int64_t wt = *n_tiles > kMaxBlocks ? (*n_tiles + 1) / 2 : *n_tiles;  // (one each while that fills fewer CUs)
    *n_blocks = wt < kMaxBlocks ? wt : kMaxBlocks;
  }
  return INR_OK;
}

static int launch(const inr_plan* plan, const LossDesc& ld, const inr::MlpArgs& a, int mode, int grid,
                  hipStream_t st) {
  hipError_t e;
  const NetDesc& nd = plan->nd;
  if (nd.mfn_n > 0)
    switch (nd.NB) {
      case 1: e = inr::launch_mfn_nb1(nd, ld, a, mode, grid, st); break;
      case 4: e = inr::launch_mfn_nb4(nd, ld, a, mode, grid, st); break;
      case 8: e = inr::launch_mfn_nb8(nd, ld, a, mode, grid, st); break;
      default: e = inr::launch_mfn_nb16(nd, ld, a, mode, grid, st); break;
    }
  else if (nd.hact == ACT_GABOR2D)
    switch (nd.NB) {
      case 2: e = inr::launch_wire2d_nb2(nd, ld, a, mode, grid, st); break;
      case 4: e = inr::launch_wire2d_nb4(nd, ld, a, mode, grid, st); break;
      case 8: e = inr::launch_wire2d_nb8(nd, ld, a, mode, grid, st); break;
      default: e = inr::launch_wire2d_nb16(nd, ld, a, mode, grid, st); break;
    }
  else if (nd.hact == ACT_GABOR)
    switch (nd.NB) {
      case 2: e = inr::launch_wire_nb2(nd, ld, a, mode, grid, st); break;
      case 4: e = inr::launch_wire_nb4(nd, ld, a, mode, grid, st); break;
      case 8: e = inr::launch_wire_nb8(nd, ld, a, mode, grid, st); break;
      default: e = inr::launch_wire_nb12(nd, ld, a, mode, grid, st); break;
    }
  else if (nd.bf16)
    e = inr::launch_siren_bf16(nd, ld, a, mode, grid, st);  // weight panels in LDS, dW by inr_dw_gemm_bf16.hip
  else
    switch (nd.NB) {
      case 1: e = inr::launch_mlp_nb1(nd, ld, a, mode, grid, st); break;
      case 2: e = inr::launch_mlp_nb2(nd, ld, a, mode, grid, st); break;
      case 4: e = inr::launch_mlp_nb4(nd, ld, a, mode, grid, st); break;
      case 8: e = inr::launch_mlp_nb8(nd, ld, a, mode, grid, st); break;
      default: e = inr::launch_mlp_nb16(nd, ld, a, mode, grid, st); break;
    }
  if (e != hipSuccess) return hip_fail(e, "inr mlp kernel launch");
  return INR_OK;
}

int inr_pack_params(const inr_plan* plan, const float* params, float* packed, void* stream) {
  if (plan == nullptr || params == nullptr || packed == nullptr)
    return fail(INR_ERR_INVALID, "inr_pack_params: null argument");
  inr::AdamArgs aa;
  memset(&aa, 0, sizeof(aa));
  aa.do_update = 0;
  hipError_t e = inr::launch_adam_pack(plan->nd, const_cast<float*>(params), nullptr, nullptr, nullptr, packed, aa,
                                       (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_pack_params");
  return INR_OK;
}

int inr_encode_logf(const float* coords, const float* bands, int64_t B, int32_t n_bands, float* out, void* stream) {
  if (coords == nullptr || bands == nullptr || out == nullptr) return fail(INR_ERR_INVALID, "inr_encode_logf: null argument");
  if (B <= 0 || n_bands <= 0) return fail(INR_ERR_INVALID, "inr_encode_logf: B %lld, n_bands %d", (long long)B, n_bands);
  hipError_t e = inr::launch_encode_logf(coords, bands, B, n_bands, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_encode_logf");
  return INR_OK;
}

int inr_encode_gauss(const float* coords, const float* enc_B, int64_t B, int32_t E, float* out, void* stream) {
  if (coords == nullptr || enc_B == nullptr || out == nullptr)
    return fail(INR_ERR_INVALID, "inr_encode_gauss: null argument");
  if (B <= 0 || E <= 0) return fail(INR_ERR_INVALID, "inr_encode_gauss: B = %lld, E = %d", (long long)B, E);
  hipError_t e = inr::launch_encode_gauss(coords, enc_B, B, E, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_encode_gauss");
  return INR_OK;
}

int inr_forward(const inr_plan* plan, const float* params, const float* packed, const float* x,
                const float* enc_B, int64_t B, float* out, const inr_workspace* ws, void* stream) {
  if (plan == nullptr || params == nullptr || packed == nullptr || x == nullptr || out == nullptr)
    return fail(INR_ERR_INVALID, "inr_forward: null argument");
  if (plan->nd.mfn_n > 0) return fail(INR_ERR_INVALID, "inr_forward: multiplicative-filter plans use inr_forward_multi");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_forward: enc_B is null");
  float* save = ws != nullptr ? ws->save : nullptr;
  if (plan->nd.hact == ACT_GABOR2D && save == nullptr)
    return fail(INR_ERR_INVALID, "inr_forward: WIRE2D plans need a save buffer (n_tiles * save_floats_per_tile floats)");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_forward: B = %lld", (long long)B);
  int64_t nt, nb;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  if (save != nullptr) {
    const int rc = check_ws(plan, ws, nt, 0, "inr_forward");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = x;
  a.encB = enc_B;
  a.out = out;
  a.save = save;
  a.B = B;
  a.n_tiles = (int)nt;
  a.save_by_block = 0;
  LossDesc ld;
  memset(&ld, 0, sizeof(ld));
  return launch(plan, ld, a, 0, (int)nb, (hipStream_t)stream);
}

// the Adam update folded into the slab reduction's launch (inr_train_adam_step)
struct AdamFuse {
  float *params, *m1, *m2, *packed;
  inr::AdamArgs aa;
};

static hipError_t reduce_stage(const inr_plan* plan, const float* slabs, int nb, float* grads, float* loss_out,
                               const float* params, const float* packed, hipStream_t st, const inr::SlabSplit& split,
                               const AdamFuse* af) {
  if (af != nullptr)
    return inr::launch_reduce_slabs_adam(plan->nd, slabs, nb, grads, loss_out, af->params, af->m1, af->m2, af->packed,
                                         af->aa, st, split);
  return inr::launch_reduce_slabs(plan->nd, slabs, nb, grads, loss_out, params, packed, st, split);
}

// dW GEMM (plans that use it) + deterministic slab reduction into flat gradients
static int finish_gradients(const inr_plan* plan, const inr::MlpArgs& a, int64_t nt, int64_t nb, float* grads,
                            float* loss_out, const float* params, const float* packed, hipStream_t st,
                            const char* who, const AdamFuse* af = nullptr) {
  inr::SlabSplit split{0, 0, 0, 0};
  if (a.dw_gemm == 2) {  // bf16 fused step: all of dW / db from the bf16 batch GEMM, summed over its chunk slabs
    inr::DwGemmBf16Args g;
    dw_gemm_bf16_setup(plan, nt, &g);
    g.save = a.save;
    g.slabs = a.slabs + (size_t)nb * plan->nd.slab_floats;
    g.coords = a.x;
    g.encB = a.encB;
    g.B = a.B;
    g.dz_state = a.dz_state + (a.dout != nullptr ? 4 : 0);  // split steps keep their own scale
    g.dz_count = a.dz_state + 8 + (a.dout != nullptr ? 2 : 0);
    hipError_t e = inr::launch_dw_gemm_bf16(g, st);
    if (e != hipSuccess) return hip_fail(e, (std::string(who) + ": bf16 weight-gradient GEMM").c_str());
    split = dw_gemm_bf16_split(plan, g);
  } else if (a.dw_gemm) {
    inr::DwGemmArgs g;
    dw_gemm_setup(plan, nt, &g, &split);
    g.save = a.save;
    g.slabs = a.slabs + (size_t)nb * plan->nd.slab_floats;
    hipError_t e = inr::launch_dw_gemm(g, st);
    if (e != hipSuccess) return hip_fail(e, (std::string(who) + ": weight-gradient GEMM").c_str());
  }
  hipError_t e = reduce_stage(plan, a.slabs, (int)nb, grads, loss_out, params, packed, st, split, af);
  if (e != hipSuccess) return hip_fail(e, (std::string(who) + ": slab reduction").c_str());
  return INR_OK;
}

int inr_backward(const inr_plan* plan, const float* params, const float* packed, const float* x,
                 const float* enc_B, int64_t B, const float* dout, const inr_workspace* ws,
                 float* grads, void* stream) {
  if (plan == nullptr || params == nullptr || packed == nullptr || x == nullptr || dout == nullptr ||
      ws == nullptr || grads == nullptr)
    return fail(INR_ERR_INVALID, "inr_backward: null argument");
  if (plan->nd.mfn_n > 0) return fail(INR_ERR_INVALID, "inr_backward: multiplicative-filter plans use inr_backward_multi");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_backward: enc_B is null");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_backward: B = %lld", (long long)B);
  int64_t nt, nb, slots, n_slabs;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  inr_plan_workspace(plan, B, &slots, &n_slabs);
  {
    const int rc = check_ws(plan, ws, nt, n_slabs, "inr_backward");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = x;
  a.encB = enc_B;
  a.dout = dout;
  a.save = ws->save;
  a.slabs = ws->slabs;
  a.B = B;
  a.n_tiles = (int)nt;
  a.save_by_block = 0;
  a.dw_gemm = w2_plan(plan) ? 2 : (dw_gemm_plan(plan) ? 1 : 0);
  LossDesc ld;
  memset(&ld, 0, sizeof(ld));
  if (w2_plan(plan)) {
    a.dz_state = dz_state_alloc(plan);
    if (a.dz_state == nullptr) return fail(INR_ERR_HIP, "inr_backward: no gradient-scale state on this device");
    if (dz_needs_calibration(plan, 1, B, 0)) {  // a pass for the scale (the forward half's stash is not touched)
      int rc = launch(plan, ld, a, 1, (int)nb, (hipStream_t)stream);
      if (rc != INR_OK) return rc;
      hipError_t e = inr::launch_dz_roll(a.dz_state + 4, nullptr, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(e, "inr_backward: gradient-scale calibration");
    }
  }
  int rc = launch(plan, ld, a, 1, (int)nb, (hipStream_t)stream);
  if (rc != INR_OK) return rc;
  if (w2_plan(plan)) dz_mark(plan, 1, B, 0);
  return finish_gradients(plan, a, nt, nb, grads, nullptr, params, packed, (hipStream_t)stream, "inr_backward");
}

// fused step (mode 2) + weight gradients + reduction, split over two streams where step_schedule says so
static int run_fused_step(const inr_plan* plan, const LossDesc& ld, const inr::MlpArgs& a, int64_t nt, int64_t nb,
                          float* grads, float* loss_out, const float* params, const float* packed, hipStream_t st,
                          const char* who, const AdamFuse* af = nullptr) {
  if (a.dw_gemm == 1 && rs_plan(plan) && rs_enabled(nt)) {
    // row-split kernel: one launch of whole rounds (no partial round to overlap), then the batch GEMM and the reduction
    // over its grid's slabs
    const RsSchedule rs = rs_schedule(nt);
    int rc = launch_rs(plan, ld, a, nt, rs, st);
    if (rc != INR_OK) return rc;
    if (grads == nullptr) return INR_OK;
    return finish_gradients(plan, a, nt, rs.grid, grads, loss_out, params, packed, st, who, af);
  }
  StepSchedule sc;
  hipStream_t side = nullptr;
  if (grads != nullptr && a.dw_gemm == 1 && step_schedule(plan, nt, nb, &sc) && sc.split) side = side_stream(plan);
  if (a.dw_gemm == 2 && dz_needs_calibration(plan, 0, a.B, ld.kind)) {  // bf16: a pass of the kernel for the scale
    int rc = launch(plan, ld, a, 2, (int)nb, st);
    if (rc != INR_OK) return rc;
    hipError_t e = inr::launch_dz_roll(a.dz_state, nullptr, st);
    if (e != hipSuccess) return hip_fail(e, (std::string(who) + ": gradient-scale calibration").c_str());
  }
  if (side == nullptr) {
    int rc = launch(plan, ld, a, 2, (int)nb, st);
    if (rc != INR_OK) return rc;
    if (a.dw_gemm == 2) dz_mark(plan, 0, a.B, ld.kind);
    if (grads == nullptr) return INR_OK;  // profiling: leave the per-block slabs unreduced
    return finish_gradients(plan, a, nt, nb, grads, loss_out, params, packed, st, who, af);
  }
  const hipEvent_t fork = plan->fork, join = plan->join;  // (created once, with the side stream)
  float* chunk_slabs = a.slabs + (size_t)nb * plan->nd.slab_floats;
  inr::MlpArgs a1 = a, a2 = a;
  a1.n_tiles = (int)sc.full;
  a2.tile0 = (int)sc.full, a2.accumulate = 1;
  sc.gA.save = sc.gB.save = a.save;
  sc.gA.slabs = chunk_slabs;
  sc.gB.slabs = chunk_slabs + (size_t)sc.gA.n_chunks * plan->nd.slab_floats;
  int rc = launch(plan, ld, a1, 2, (int)nb, st);
  hipError_t e = hipSuccess;
  if (rc == INR_OK) e = hipEventRecord(fork, st);
  if (rc == INR_OK && e == hipSuccess) rc = launch(plan, ld, a2, 2, (int)sc.rem, st);  // (queued before the GEMM: the critical path)
  if (rc == INR_OK && e == hipSuccess) e = hipStreamWaitEvent(side, fork, 0);
  if (rc == INR_OK && e == hipSuccess) e = inr::launch_dw_gemm(sc.gA, side);
  if (rc == INR_OK && e == hipSuccess) e = hipEventRecord(join, side);
  if (rc == INR_OK && e == hipSuccess) e = hipStreamWaitEvent(st, join, 0);
  if (rc == INR_OK && e == hipSuccess) e = inr::launch_dw_gemm(sc.gB, st);
  if (rc == INR_OK && e == hipSuccess)
    e = reduce_stage(plan, a.slabs, (int)nb, grads, loss_out, params, packed, st, sc.red, af);
  if (rc != INR_OK) return rc;
  if (e != hipSuccess) return hip_fail(e, (std::string(who) + ": split step").c_str());
  return INR_OK;
}

static void to_loss_desc(const inr_loss_desc* l, LossDesc* o) {
  memset(o, 0, sizeof(*o));
  o->scale = l->scale == 0.f ? 1.f : l->scale;
  o->cons_w = l->cons_w;
  o->cons_chan = l->cons_chan == 1 ? 1 : 2;
  for (int i = 0; i < INR_MAX_HEADS; ++i) {
    o->cons_lo[i] = l->cons_lo[i];
    o->cons_hi[i] = l->cons_hi[i];
    o->cons_inv[i] = l->cons_inv[i];
  }
  o->kind = l->kind;
  o->eps = l->eps;
  o->sigma = l->sigma;
  o->factor = l->factor;
  o->inv_count = l->inv_count;
  o->hdr_A = l->hdr_A;
}

int inr_loss_grad(const inr_loss_desc* loss, const float* out, const float* gt, const float* kcoords,
                  const uint8_t* mask, int64_t B, float* loss_out, float* dout, void* stream) {
  if (loss == nullptr || out == nullptr || gt == nullptr || loss_out == nullptr || dout == nullptr)
    return fail(INR_ERR_INVALID, "inr_loss_grad: null argument");
  if (loss->kind < INR_LOSS_L2_HALF || loss->kind > INR_LOSS_CENTER)
    return fail(INR_ERR_INVALID, "inr_loss_grad: loss kind %d", loss->kind);
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_loss_grad: B = %lld", (long long)B);
  LossDesc ld;
  to_loss_desc(loss, &ld);
  hipError_t e = inr::launch_loss_grad(ld, out, gt, kcoords, mask, B, loss_out, dout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_loss_grad");
  return INR_OK;
}

int inr_loss_grad_multi(const inr_loss_desc* loss, const float* outs, const float* gt, const float* dist,
                        const uint8_t* mask, int32_t n_heads, int64_t B, float* loss_out, float* douts, void* stream) {
  if (loss == nullptr || outs == nullptr || gt == nullptr || loss_out == nullptr || douts == nullptr)
    return fail(INR_ERR_INVALID, "inr_loss_grad_multi: null argument");
  if (loss->kind < INR_LOSS_L2_HALF || loss->kind > INR_LOSS_CENTER)
    return fail(INR_ERR_INVALID, "inr_loss_grad_multi: loss kind %d", loss->kind);
  if (n_heads < 1 || n_heads > INR_MAX_HEADS || B <= 0)
    return fail(INR_ERR_INVALID, "inr_loss_grad_multi: n_heads %d, B %lld", n_heads, (long long)B);
  if (loss->cons_w != 0.f && dist == nullptr)
    return fail(INR_ERR_INVALID, "inr_loss_grad_multi: the consistency term needs dist");
  LossDesc ld;
  to_loss_desc(loss, &ld);
  hipError_t e = inr::launch_loss_grad_multi(ld, outs, gt, dist, mask, n_heads, B, loss_out, douts, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_loss_grad_multi");
  return INR_OK;
}

int inr_tv_grad(const float* out, int64_t R, int64_t R_own, int64_t W, int64_t H, float weight,
                float* loss_out, float* dout, void* stream) {
  if (out == nullptr || loss_out == nullptr || dout == nullptr) return fail(INR_ERR_INVALID, "inr_tv_grad: null argument");
  if (R <= 0 || R_own <= 0 || R_own > R || R > R_own + 1 || W < 2 || H < 2 || R > H)
    return fail(INR_ERR_INVALID, "inr_tv_grad: R %lld R_own %lld W %lld H %lld", (long long)R, (long long)R_own,
                (long long)W, (long long)H);
  const float cw = (float)((double)weight / ((double)H * (double)(W - 1) * 2.0));
  const float ch = (float)((double)weight / ((double)(H - 1) * (double)W * 2.0));
  hipError_t e = inr::launch_tv_grad(out, R, R_own, W, cw, ch, loss_out, dout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_tv_grad");
  return INR_OK;
}

int inr_loss_tv_grad(const inr_loss_desc* loss, const float* out, const float* gt, const uint8_t* mask, int64_t R,
                     int64_t R_own, int64_t W, int64_t H, float tv_weight, float* loss_out, float* dout, void* stream) {
  if (loss == nullptr || out == nullptr || gt == nullptr || loss_out == nullptr || dout == nullptr)
    return fail(INR_ERR_INVALID, "inr_loss_tv_grad: null argument");
  if (loss->kind < INR_LOSS_L2_HALF || loss->kind > INR_LOSS_CENTER)
    return fail(INR_ERR_INVALID, "inr_loss_tv_grad: loss kind %d", loss->kind);
  if (R <= 0 || R_own <= 0 || R_own > R || R > R_own + 1 || W < 2 || H < 2 || R > H)
    return fail(INR_ERR_INVALID, "inr_loss_tv_grad: R %lld R_own %lld W %lld H %lld", (long long)R, (long long)R_own,
                (long long)W, (long long)H);
  LossDesc ld;
  to_loss_desc(loss, &ld);
  const float cw = (float)((double)tv_weight / ((double)H * (double)(W - 1) * 2.0));
  const float ch = (float)((double)tv_weight / ((double)(H - 1) * (double)W * 2.0));
  hipError_t e = inr::launch_loss_tv_grad(ld, out, gt, mask, R, R_own, W, cw, ch, loss_out, dout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_loss_tv_grad");
  return INR_OK;
}

int inr_center_pairs_grad(const float* out, const float* gt, const int64_t* idx_a, const int64_t* idx_b, int64_t n,
                          int64_t B, float weight, float* loss_out, float* dout, void* stream) {
  if (out == nullptr || gt == nullptr || idx_a == nullptr || idx_b == nullptr || loss_out == nullptr || dout == nullptr)
    return fail(INR_ERR_INVALID, "inr_center_pairs_grad: null argument");
  if (n <= 0 || B <= 0) return fail(INR_ERR_INVALID, "inr_center_pairs_grad: n %lld B %lld", (long long)n, (long long)B);
  hipError_t e = inr::launch_center_pairs(out, gt, (const long long*)idx_a, (const long long*)idx_b, n, B,
                                          (float)((double)weight / (double)n), loss_out, dout, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_center_pairs_grad");
  return INR_OK;
}

static int train_step_impl(const inr_plan* plan, const inr_loss_desc* loss, const float* params, const float* packed,
                           const float* x, const float* enc_B, const float* gt, const uint8_t* mask, int64_t B,
                           const inr_workspace* ws, float* grads, float* loss_out, void* stream, const AdamFuse* af) {
  if (plan == nullptr || loss == nullptr || params == nullptr || packed == nullptr || x == nullptr ||
      gt == nullptr || ws == nullptr || loss_out == nullptr)
    return fail(INR_ERR_INVALID, "inr_train_step: null argument");
  if (plan->nd.mfn_n > 0)
    return fail(INR_ERR_INVALID, "inr_train_step: multiplicative-filter plans use inr_train_step_multi");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_train_step: enc_B is null");
  if (loss->kind < INR_LOSS_L2_HALF || loss->kind > INR_LOSS_CENTER)
    return fail(INR_ERR_INVALID, "inr_train_step: loss kind %d", loss->kind);
  if (loss->kind >= INR_LOSS_LOGSPACE && plan->nd.out_f != 2)
    return fail(INR_ERR_INVALID, "inr_train_step: complex-row losses need out_features == 2");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_train_step: B = %lld", (long long)B);
  int64_t nt, nb, slots, n_slabs;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  inr_plan_workspace(plan, B, &slots, &n_slabs);
  {
    const int rc = check_ws(plan, ws, slots, n_slabs, "inr_train_step");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = x;
  a.encB = enc_B;
  a.gt = gt;
  a.mask = mask;
  a.save = ws->save;
  a.slabs = ws->slabs;
  a.B = B;
  a.n_tiles = (int)nt;
  a.dw_gemm = w2_plan(plan) ? 2 : (dw_gemm_plan(plan) ? 1 : 0);
  a.save_by_block = a.dw_gemm ? 0 : 1;
  if (a.dw_gemm == 2) {
    a.dz_state = dz_state_alloc(plan);
    if (a.dz_state == nullptr) return fail(INR_ERR_HIP, "inr_train_step: no gradient-scale state on this device");
  }
#ifdef INR_STAMPS
  a.dbg = g_stamp_buf;
  a.dbg_cap = g_stamp_cap;
#endif
  LossDesc ld;
  to_loss_desc(loss, &ld);
  return run_fused_step(plan, ld, a, nt, nb, grads, loss_out, params, packed, (hipStream_t)stream, "inr_train_step", af);
}

int inr_train_step(const inr_plan* plan, const inr_loss_desc* loss, const float* params, const float* packed,
                   const float* x, const float* enc_B, const float* gt, const uint8_t* mask, int64_t B,
                   const inr_workspace* ws, float* grads, float* loss_out, void* stream) {
  return train_step_impl(plan, loss, params, packed, x, enc_B, gt, mask, B, ws, grads, loss_out, stream, nullptr);
}

int inr_plan_set_bounds(inr_plan* plan, const float* lo, const float* hi, int32_t n) {
  if (plan == nullptr || lo == nullptr || hi == nullptr) return fail(INR_ERR_INVALID, "inr_plan_set_bounds: null argument");
  if (!plan->nd.bounded) return fail(INR_ERR_INVALID, "inr_plan_set_bounds: not a MultiscaleBoundedFourier plan");
  if (n != plan->nd.mfn_n) return fail(INR_ERR_INVALID, "inr_plan_set_bounds: %d bounds for %d linears", n, plan->nd.mfn_n);
  for (int i = 0; i < n; ++i) {
    plan->nd.bound_lo[i] = lo[i];
    plan->nd.bound_hi[i] = hi[i];
  }
  return INR_OK;
}

int inr_plan_heads(const inr_plan* plan, int32_t* n_heads) {
  if (plan == nullptr || n_heads == nullptr) return fail(INR_ERR_INVALID, "inr_plan_heads: null argument");
  *n_heads = plan->nd.mfn_n > 0 ? plan->nd.n_heads : 1;
  return INR_OK;
}

int inr_forward_multi(const inr_plan* plan, const float* params, const float* packed, const float* coords,
                      const float* enc_B, const float* dist, int64_t B, float* out, const inr_workspace* ws,
                      int32_t by_block, void* stream) {
  if (plan == nullptr || params == nullptr || packed == nullptr || coords == nullptr || out == nullptr ||
      ws == nullptr)
    return fail(INR_ERR_INVALID, "inr_forward_multi: null argument");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_forward_multi: enc_B is null");
  if (plan->nd.bounded && dist == nullptr) return fail(INR_ERR_INVALID, "inr_forward_multi: bounded model needs dist");
  if (plan->nd.mfn_n == 0) return fail(INR_ERR_INVALID, "inr_forward_multi: not a multiplicative-filter plan");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_forward_multi: B = %lld", (long long)B);
  int64_t nt, nb;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  {
    const int rc = check_ws(plan, ws, by_block ? nb : nt, 0, "inr_forward_multi");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = coords;
  a.encB = enc_B;
  a.out = out;
  a.dist = dist;
  a.save = ws->save;
  a.B = B;
  a.n_tiles = (int)nt;
  a.save_by_block = by_block ? 1 : 0;
  LossDesc ld;
  memset(&ld, 0, sizeof(ld));
  return launch(plan, ld, a, 0, (int)nb, (hipStream_t)stream);
}

int inr_backward_multi(const inr_plan* plan, const float* params, const float* packed, const float* coords,
                       const float* enc_B, const float* dist, int64_t B, const float* dout,
                       const inr_workspace* ws, float* grads, void* stream) {
  if (plan == nullptr || params == nullptr || packed == nullptr || coords == nullptr || dout == nullptr ||
      ws == nullptr || grads == nullptr)
    return fail(INR_ERR_INVALID, "inr_backward_multi: null argument");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_backward_multi: enc_B is null");
  if (plan->nd.bounded && dist == nullptr) return fail(INR_ERR_INVALID, "inr_backward_multi: bounded model needs dist");
  if (plan->nd.mfn_n == 0) return fail(INR_ERR_INVALID, "inr_backward_multi: not a multiplicative-filter plan");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_backward_multi: B = %lld", (long long)B);
  int64_t nt, nb, slots, n_slabs;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  inr_plan_workspace(plan, B, &slots, &n_slabs);
  {
    const int rc = check_ws(plan, ws, nt, n_slabs, "inr_backward_multi");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = coords;
  a.encB = enc_B;
  a.dout = dout;
  a.dist = dist;
  a.save = ws->save;
  a.slabs = ws->slabs;
  a.B = B;
  a.n_tiles = (int)nt;
  a.save_by_block = 0;
  LossDesc ld;
  memset(&ld, 0, sizeof(ld));
  a.dw_gemm = dw_gemm_plan(plan) ? 1 : 0;
  int rc = launch(plan, ld, a, 1, (int)nb, (hipStream_t)stream);
  if (rc != INR_OK) return rc;
  return finish_gradients(plan, a, nt, nb, grads, nullptr, params, packed, (hipStream_t)stream, "inr_backward_multi");
}

int inr_train_step_multi(const inr_plan* plan, const inr_loss_desc* loss, const float* params, const float* packed,
                         const float* coords, const float* enc_B, const float* gt, const float* dist,
                         const uint8_t* mask, int64_t B, const inr_workspace* ws, float* grads, float* loss_out,
                         void* stream) {
  if (plan == nullptr || loss == nullptr || params == nullptr || packed == nullptr || coords == nullptr ||
      gt == nullptr || ws == nullptr || loss_out == nullptr)
    return fail(INR_ERR_INVALID, "inr_train_step_multi: null argument");
  if (plan->nd.input == IN_GAUSS && enc_B == nullptr) return fail(INR_ERR_INVALID, "inr_train_step_multi: enc_B is null");
  if (plan->nd.mfn_n == 0) return fail(INR_ERR_INVALID, "inr_train_step_multi: not a multiplicative-filter plan");
  if (loss->kind < INR_LOSS_L2_HALF || loss->kind > INR_LOSS_CENTER)
    return fail(INR_ERR_INVALID, "inr_train_step_multi: loss kind %d", loss->kind);
  if ((loss->cons_w != 0.f || plan->nd.bounded) && dist == nullptr)
    return fail(INR_ERR_INVALID, "inr_train_step_multi: the consistency term / bounded linears need dist");
  if (B <= 0) return fail(INR_ERR_INVALID, "inr_train_step_multi: B = %lld", (long long)B);
  int64_t nt, nb, slots, n_slabs;
  inr_plan_launch_dims(plan, B, &nt, &nb);
  inr_plan_workspace(plan, B, &slots, &n_slabs);
  {
    const int rc = check_ws(plan, ws, slots, n_slabs, "inr_train_step_multi");
    if (rc != INR_OK) return rc;
  }
  inr::MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.params = params;
  a.packed = packed;
  a.x = coords;
  a.encB = enc_B;
  a.gt = gt;
  a.dist = dist;
  a.mask = mask;
  a.save = ws->save;
  a.slabs = ws->slabs;
  a.B = B;
  a.n_tiles = (int)nt;
  a.dw_gemm = dw_gemm_plan(plan) ? 1 : 0;
  a.save_by_block = a.dw_gemm ? 0 : 1;
#ifdef INR_STAMPS
  a.dbg = g_stamp_buf;
  a.dbg_cap = g_stamp_cap;
#endif
  LossDesc ld;
  to_loss_desc(loss, &ld);
  return run_fused_step(plan, ld, a, nt, nb, grads, loss_out, params, packed, (hipStream_t)stream,
                        "inr_train_step_multi");
}

static bool has_complex_tensors(const NetDesc& nd) {
  for (int l = 0; l < nd.ND; ++l)
    if (nd.L[l].ltype == LT_WIRE_HIDDEN || nd.L[l].ltype == LT_WIRE_LAST) return true;
  return false;
}

// the Adam kernels form the penalty gradients per REAL entry: right for every tensor of a real model, wrong for complex64
static int check_real_penalty(const inr_plan* plan, double l1, double l2, const char* who) {
  if ((l1 != 0.0 || l2 != 0.0) && has_complex_tensors(plan->nd))
    return fail(INR_ERR_INVALID, "%s: l1 / l2 on a plan with complex64 tensors -- add the penalty gradient with "
                "inr_reg_grad and pass l1 = l2 = 0", who);
  return INR_OK;
}

// torch computes these in Python doubles and passes them to fp32 kernels as scalars
static void adam_bias_terms(double lr, double beta1, double beta2, int32_t step, float* step_size, float* bc2_sqrt) {
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  *step_size = (float)(lr / bc1);
  *bc2_sqrt = (float)std::sqrt(bc2);
}

int inr_adam_schedule(double lr, double beta1, double beta2, int32_t n, float* host_out) {
  if (host_out == nullptr || n < 1) return fail(INR_ERR_INVALID, "inr_adam_schedule: null table or n < 1");
  for (int32_t t = 0; t < n; ++t) adam_bias_terms(lr, beta1, beta2, t + 1, host_out + 2 * t, host_out + 2 * t + 1);
  return INR_OK;
}

int inr_adam_step_dev(const inr_plan* plan, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      float* packed, const float* sched, int32_t n_sched, int32_t* step_dev, double beta1,
                      double beta2, double eps, double weight_decay, double l1, double l2, void* stream) {
  if (plan == nullptr || params == nullptr || grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr ||
      packed == nullptr || sched == nullptr || step_dev == nullptr)
    return fail(INR_ERR_INVALID, "inr_adam_step_dev: null argument");
  if (n_sched < 1) return fail(INR_ERR_INVALID, "inr_adam_step_dev: empty schedule");
  if (int rc = check_real_penalty(plan, l1, l2, "inr_adam_step_dev")) return rc;
  inr::AdamArgs aa;
  aa.do_update = 1;
  aa.step_size = 0.f;
  aa.bc2_sqrt = 1.f;
  aa.sched = sched;
  aa.step_dev = step_dev;
  aa.n_sched = n_sched;
  aa.omb1 = (float)(1.0 - beta1);
  aa.beta2 = (float)beta2;
  aa.omb2 = (float)(1.0 - beta2);
  aa.eps = (float)eps;
  aa.weight_decay = (float)weight_decay;
  aa.l1 = (float)l1;
  aa.l2 = (float)l2;
  hipError_t e = inr::launch_adam_pack(plan->nd, params, grads, exp_avg, exp_avg_sq, packed, aa, (hipStream_t)stream);
  if (e == hipSuccess) e = inr::launch_step_advance(step_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_adam_step_dev");
  return INR_OK;
}

int inr_adam_step(const inr_plan* plan, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  float* packed, double lr, double beta1, double beta2, double eps, double weight_decay,
                  double l1, double l2, int32_t step, void* stream) {
  if (plan == nullptr || params == nullptr || grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr ||
      packed == nullptr)
    return fail(INR_ERR_INVALID, "inr_adam_step: null argument");
  if (step < 1) return fail(INR_ERR_INVALID, "inr_adam_step: step %d (counts from 1)", step);
  if (int rc = check_real_penalty(plan, l1, l2, "inr_adam_step")) return rc;
  inr::AdamArgs aa;
  aa.do_update = 1;
  aa.sched = nullptr;
  aa.step_dev = nullptr;
  aa.n_sched = 0;
  adam_bias_terms(lr, beta1, beta2, step, &aa.step_size, &aa.bc2_sqrt);
  aa.omb1 = (float)(1.0 - beta1);
  aa.beta2 = (float)beta2;
  aa.omb2 = (float)(1.0 - beta2);
  aa.eps = (float)eps;
  aa.weight_decay = (float)weight_decay;
  aa.l1 = (float)l1;
  aa.l2 = (float)l2;
  hipError_t e = inr::launch_adam_pack(plan->nd, params, grads, exp_avg, exp_avg_sq, packed, aa, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_adam_step");
  return INR_OK;
}

int inr_reg_grad(const inr_plan* plan, const float* params, float* grads, int64_t lo, int64_t hi, double l1, double l2,
                 const float* l2_dir, void* stream) {
  if (plan == nullptr || params == nullptr || grads == nullptr) return fail(INR_ERR_INVALID, "inr_reg_grad: null argument");
  if (lo < 0 || hi < lo || hi > plan->nd.P)
    return fail(INR_ERR_INVALID, "inr_reg_grad: entries [%lld, %lld) of %d", (long long)lo, (long long)hi, plan->nd.P);
  if (l2 != 0.0 && l2_dir == nullptr && has_complex_tensors(plan->nd))
    return fail(INR_ERR_INVALID, "inr_reg_grad: l2 on a plan with complex64 tensors needs l2_dir (conj(S) / |S|)");
  hipError_t e = inr::launch_reg_grad(plan->nd, params, grads, (int)lo, (int)hi, (float)l1, (float)l2, l2_dir,
                                      (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_reg_grad");
  return INR_OK;
}

int inr_adam_step_shard(const inr_plan* plan, float* params, const float* grads_shard, float* exp_avg,
                        float* exp_avg_sq, int64_t lo, int64_t hi, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double l1, double l2, int32_t step, void* stream) {
  if (plan == nullptr || params == nullptr || grads_shard == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr)
    return fail(INR_ERR_INVALID, "inr_adam_step_shard: null argument");
  if (step < 1) return fail(INR_ERR_INVALID, "inr_adam_step_shard: step %d (counts from 1)", step);
  if (int rc = check_real_penalty(plan, l1, l2, "inr_adam_step_shard")) return rc;
  if (lo < 0 || hi < lo || hi > plan->nd.P)
    return fail(INR_ERR_INVALID, "inr_adam_step_shard: entries [%lld, %lld) of %d", (long long)lo, (long long)hi,
                plan->nd.P);
  inr::AdamArgs aa;
  aa.do_update = 1;
  aa.sched = nullptr;
  aa.step_dev = nullptr;
  aa.n_sched = 0;
  adam_bias_terms(lr, beta1, beta2, step, &aa.step_size, &aa.bc2_sqrt);
  aa.omb1 = (float)(1.0 - beta1);
  aa.beta2 = (float)beta2;
  aa.omb2 = (float)(1.0 - beta2);
  aa.eps = (float)eps;
  aa.weight_decay = (float)weight_decay;
  aa.l1 = (float)l1;
  aa.l2 = (float)l2;
  hipError_t e = inr::launch_adam_shard(plan->nd, params, grads_shard, exp_avg, exp_avg_sq, (int)lo, (int)hi, aa,
                                        (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "inr_adam_step_shard");
  return INR_OK;
}

int inr_train_adam_step(const inr_plan* plan, const inr_loss_desc* loss, float* params, float* packed, const float* x,
                        const float* enc_B, const float* gt, const uint8_t* mask, int64_t B, const inr_workspace* ws,
                        float* grads, float* loss_out, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                        double beta2, double eps, double weight_decay, double l1, double l2, int32_t step,
                        void* stream) {
  if (grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr)
    return fail(INR_ERR_INVALID, "inr_train_adam_step: null argument");
  if (step < 1) return fail(INR_ERR_INVALID, "inr_train_adam_step: step %d (counts from 1)", step);
  if (plan != nullptr)
    if (int rc = check_real_penalty(plan, l1, l2, "inr_train_adam_step")) return rc;
  AdamFuse af;
  af.params = params, af.m1 = exp_avg, af.m2 = exp_avg_sq, af.packed = packed;
  memset(&af.aa, 0, sizeof(af.aa));
  af.aa.do_update = 1;
  adam_bias_terms(lr, beta1, beta2, step, &af.aa.step_size, &af.aa.bc2_sqrt);
  af.aa.omb1 = (float)(1.0 - beta1);
  af.aa.beta2 = (float)beta2;
  af.aa.omb2 = (float)(1.0 - beta2);
  af.aa.eps = (float)eps;
  af.aa.weight_decay = (float)weight_decay;
  af.aa.l1 = (float)l1;
  af.aa.l2 = (float)l2;
  return train_step_impl(plan, loss, params, packed, x, enc_B, gt, mask, B, ws, grads, loss_out, stream, &af);
}

#ifdef INR_STAMPS
// entries = 64 per WAVE of the grid (kernels index (blockIdx.x * waves_per_workgroup + wave) * 64 + stamp)
int inr_debug_set_stamp_buffer(long long* buf, long long entries) {
  g_stamp_buf = buf;
  g_stamp_cap = buf != nullptr ? entries : 0;
  return INR_OK;
}
#endif

}  // extern "C"
