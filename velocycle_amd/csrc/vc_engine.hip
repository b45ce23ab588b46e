// Host side of the engine behind the C ABI of include/velocycle_hip.h: configuration, HBM layout,
// count histograms, workspaces, kernel sequencing.  No exception leaves this file (VC_GUARD_* around every entry
// point that allocates host memory).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "vc_common.h"
#include "vc_host_logic.h"
#include "vc_tail_spec.h"

namespace {

thread_local std::string g_create_error;

}  // namespace

// RCCL, bound at run time (vc_comm_init_rccl): the four entry points the sharded step needs.  Declared here instead of
// including rccl.h so that the library neither links nor requires RCCL (single-GPU use never loads it).
struct VcNcclId { char internal[128]; };
typedef void* VcNcclComm;
struct VcRccl {
  void* dl = nullptr;
  int (*GetUniqueId)(VcNcclId*) = nullptr;
  int (*CommInitRank)(VcNcclComm*, int, VcNcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, VcNcclComm, hipStream_t) = nullptr;
  int (*CommDestroy)(VcNcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load(const char* path, std::string* err) {
    if (dl) return true;
    dl = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) { *err = std::string("dlopen(rccl): ") + dlerror(); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(dl, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(dl, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(dl, "ncclAllReduce");
    CommDestroy = (decltype(CommDestroy))dlsym(dl, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(dl, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { *err = "librccl.so lacks the nccl* entry points"; return false; }
    return true;
  }
};
static VcRccl g_rccl;

struct vc_engine {
  vc_config cfg{};
  vc_tuning tun{};                    // vc_set_tuning: all-zero = defaults (the library reads no environment variable)
  int opt_kind = VC_OPT_CLIPPED_ADAM; // vc_set_optimizer: which optimiser the step entry points apply
  double opt_wd = 0.0;                // ... and its weight decay
  const unsigned char* opt_frozen = nullptr;   // ... and the parameter tensors it must not touch (caller-owned device bytes [total], or null)
  VcAdamHyper hyper(double lr, double lrd, double b1, double b2, double eps, double clip) const {
    VcAdamHyper h;
    h.lr0 = lr; h.lrd = opt_kind == VC_OPT_ADAM ? 1.0 : lrd; h.b1 = b1; h.b2 = b2;
    h.eps = (float)eps; h.clip = opt_kind == VC_OPT_ADAM ? __builtin_inff() : (float)clip; h.wd = (float)opt_wd; h.kind = opt_kind;
    h.frozen = opt_frozen; h.frozen_off = 0;
    return h;
  }
  void fill(VcAdamArgs& a, float* m, float* v, double lr, double lrd, double b1, double b2, double eps, double clip) const {
    const VcAdamHyper h = hyper(lr, lrd, b1, b2, eps, clip);
    a.m = m; a.v = v;
    a.lr0 = h.lr0; a.lrd_l = log(h.lrd); a.b1l = log(h.b1); a.b2l = log(h.b2);
    a.b1 = (float)h.b1; a.b2 = (float)h.b2; a.eps = h.eps; a.clip = h.clip;
    a.header = (int)layout.header; a.wd = h.wd; a.kind = h.kind; a.frozen = h.frozen;
  }
  VcNcclComm comm = nullptr;          // the engine's own communicator (vc_comm_init_rccl), or null
  double* particle_lsum = nullptr;    // vc_svi_run_particles: scratch slots of the particles' K_fin launches
  // vc_svi_run_particles: particle k >= 1 has its own per-step workspaces, gradient buffer and stream (particle 0: e->b, the
  // caller's gradient buffer and stream), so that the small launches of one particle run beside the likelihood kernel of another
  struct Particle { VcBufs b; float* grad = nullptr; hipStream_t st = nullptr; hipEvent_t main_done = nullptr, done = nullptr; };
  std::vector<Particle> particles;
  hipEvent_t ev_params = nullptr, ev_main0 = nullptr;
  VcBufs* particle_bufs_dev = nullptr;   // [VC_MAX_PARTICLES] the particles' VcBufs on the device (entry 0 = b): the one-launch K_pre / K_post
  int particle_bufs_n = 0;               // entries of it that are filled
  std::vector<size_t> alloc_bytes;    // parallel to `allocs`
  float* sis = nullptr;               // phase A's snapshot of shape_inv {parameter, exp_avg, exp_avg_sq} [3][Ng_pad]
  int xb_pw_off = 0, xb_pw_cap = 0, xb_loss_off = 0;
  long long xb_total = 0;
  // one-shot peer-to-peer exchange (vc_p2p_alloc / vc_p2p_connect, vc_p2p_exchange.hip)
  VcP2p p2p{};
  int p2p_nblk = 0;                  // flags per rank of the one-launch exchange (vc_tail_x_kernel)
  long long p2p_bflag_off = 0;       // words from a region's base to those flags
  void** p2p_tab = nullptr;          // device: [3][VC_P2P_MAX_RANKS] pointers -- the regions, the parity-0 slots, the parity-1 slots
  void* p2p_own = nullptr;            // this rank's region (hipMalloc, IPC-exported)
  bool p2p_connected = false;
  int p2p_mem_kind = -1;              // 0 fine-grained, 1 uncached, 2 plain hipMalloc (vc_p2p_alloc)
  long long p2p_step = 0;             // steps exchanged so far: slot parity and flag value, identical on every rank
  double p2p_timeout_s = 2.0;
  unsigned long long* p2p_verdict = nullptr;   // {step + 1, dead}: the one verdict of an exchange launch (vc_p2p_exchange.hip)
  VcDims d{};
  VcBufs b{};
  vc_layout layout{};
  std::string err;
  std::vector<void*> allocs;
  bool finalized = false;
  vc_main_launch_fn main_fn_rows = nullptr;   // pw_lane engines: the U-only kernel WITH per-cell rows (main_fn stores none)
  bool D_all_ones = false;            // velocity: one condition and D == 1 for every cell (vc_set_cell_data): W_c = zeta_omega(phi_c)
  bool finalize_started = false;      // vc_finalize got past its call-order checks
  bool finalize_failed = false;       // vc_finalize returned an error after it had started to consume its inputs
  bool generic_needed = false;        // the configuration lies outside the compiled fast set for a reason other than its batches (vc_create)
  bool generic_nb = false;            // ... because it has more than VC_MAXNB batches: generic only if their design matrix is not one-hot
  std::vector<float> hDb;             // host copy of the batch design matrix (vc_finalize: is it one-hot?)
  std::vector<float> hD;              // host copy of the condition design matrix (vc_finalize: one condition per workgroup of the U-only kernel?)
  std::vector<int> bat_cond;          // pw_lane with several conditions: the condition every cell of batch q belongs to
  bool have_counts = false, have_cells = false;
  bool prior_set[VC_PRIOR_COUNT] = {};
  // host copies needed at finalize
  std::vector<float> hS, hU;            // tuning.host_hist only (the host histogram pass kept as the checker of the device one)
  bool host_hist = false;
  std::vector<float> h_prior[VC_PRIOR_COUNT];
  std::vector<float> h_cond[VC_SITE_COUNT];
  // the count matrices as handed over, until vc_finalize re-lays them out: dense strided (device pointer: the caller's
  // when on_device, else an owned compact upload) or CSR (cells x genes)
  struct CountSrc {
    int kind = 0;                       // 0 none, 1 dense, 2 CSR
    const float* dense = nullptr;
    const long long* indptr = nullptr;
    const int* indices = nullptr;
    const float* data = nullptr;
    long long nnz = 0;
    void* owned[3] = {nullptr, nullptr, nullptr};
    void release() {
      for (void*& p : owned) { if (p) (void)hipFree(p); p = nullptr; }
      kind = 0; dense = nullptr; indptr = nullptr; indices = nullptr; data = nullptr; nnz = 0;
    }
  } src[2];
  // what the last vc_finalize measured about its own set-up
  std::vector<int> h_ptr_host;          // the histogram CSR as uploaded (vc_get_histogram)
  std::vector<float> h_val_host, h_cnt_host;
  size_t setup_transient_bytes = 0;     // peak device memory held only during vc_finalize
  int hist_on_device = 0;
  vc_main_launch_fn main_fn = nullptr;
  // vc_set_loss_every: the gradient-only instantiation of the U-only kernel (null: none for this configuration), the period and
  // the count of likelihood launches of the fused single-rank runs since it was set (launch n evaluates the loss iff n % k == 0)
  vc_main_launch_fn main_fn_nl = nullptr;
  const char* main_name_nl = "";
  int loss_every = 1;
  long long loss_ctr = 0;
  vc_main_launch_fn phase_fn = nullptr;  // S-only kernel used once to hoist the S term (VU kind)
  const char* main_name = "";
  long long gs = 0, cs = 0;          // strides of the host copies hS / hU
  long long dgs = 0, dcs = 0;        // strides of the dense device sources
  bool hist_each_step = false;
  // tutorial flow: ordinary (three-launch) fused steps since the tables were last primed; from the third on the merged tail
  // launch may be used (both halves of the loss terms K_tail's cell blocks write are then in place)
  int plain_steps = 0;
  // optional hipEvent timing of the likelihood kernel (bench.py roofline)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double main_ms = 0.0;
  long long main_launches = 0;
  int drain_events() {
    for (size_t i = 0; i < ev_used; ++i) {
      float ms = 0.f;
      if (hipEventSynchronize(ev_pool[i].second) != hipSuccess ||
          hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second) != hipSuccess)
        return fail(VC_ERR_HIP, "hipEventElapsedTime failed");
      main_ms += ms;
      main_launches++;
    }
    ev_used = 0;
    return VC_OK;
  }

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  void dfree(const void* p) {           // release one tracked allocation before vc_destroy
    for (size_t i = 0; i < allocs.size(); ++i)
      if (allocs[i] == p) { (void)hipFree(allocs[i]); allocs.erase(allocs.begin() + i); alloc_bytes.erase(alloc_bytes.begin() + i); return; }
  }
  size_t bytes_of(const void* p) const {
    for (size_t i = 0; i < allocs.size(); ++i) if (allocs[i] == p) return alloc_bytes[i];
    return 0;
  }
  // a second buffer of the same size and content (particle workspaces); *pp may be null (a buffer this model does not have)
  template <class T>
  int dclone(T** pp) {
    if (!*pp) return VC_OK;
    const size_t nbytes = bytes_of(*pp);
    if (!nbytes) return fail(VC_ERR_STATE, "dclone: not an engine allocation");
    char* q = nullptr;
    int rc = dalloc(&q, nbytes);
    if (rc != VC_OK) return rc;
    if (hipMemcpy(q, *pp, nbytes, hipMemcpyDeviceToDevice) != hipSuccess) return fail(VC_ERR_HIP, "dclone: copy failed");
    *pp = (T*)q;
    return VC_OK;
  }
  template <class T>
  int dalloc(T** out, size_t n) {
    void* p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(&p, n * sizeof(T));
    if (e != hipSuccess) return fail(VC_ERR_HIP, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    allocs.push_back(p);
    alloc_bytes.push_back(n * sizeof(T));
    *out = (T*)p;
    return VC_OK;
  }
};

#define HIPCHK(e_, call)                                                                        \
  do {                                                                                          \
    hipError_t _st = (call);                                                                    \
    if (_st != hipSuccess) return (e_)->fail(VC_ERR_HIP, "%s: %s", #call, hipGetErrorString(_st)); \
  } while (0)
#define TRY(x)                 \
  do {                         \
    int _rc = (x);             \
    if (_rc != VC_OK) return _rc; \
  } while (0)

// Every entry point that allocates host memory runs its body through this guard: no C++ exception crosses the C ABI.
#define VC_GUARD_BEGIN try {
#define VC_GUARD_END(e_)                                                                         \
  } catch (const std::bad_alloc&) {                                                              \
    return (e_) ? (e_)->fail(VC_ERR_ARG, "out of host memory") : VC_ERR_ARG;                     \
  } catch (const std::exception& ex) {                                                           \
    return (e_) ? (e_)->fail(VC_ERR_STATE, "internal error: %s", ex.what()) : VC_ERR_STATE;      \
  } catch (...) {                                                                                \
    return (e_) ? (e_)->fail(VC_ERR_STATE, "internal error") : VC_ERR_STATE;                     \
  }

static bool cond(const vc_engine* e, int site) { return (e->d.cond >> site) & 1u; }
static int fused_tail_kind(const vc_engine* e);

static long long site_size(const vc_engine* e, int site) {
  const VcDims& d = e->d;
  switch (site) {
    case VC_SITE_PHIXY: return 2LL * d.Nc;
    case VC_SITE_NU: return (long long)d.Ng * d.Nh;
    case VC_SITE_DNU: return (long long)d.Nb * d.Ng;
    case VC_SITE_NUOMEGA: return d.NW;
    default: return d.Ng;
  }
}

static bool site_exists(const vc_engine* e, int site) {
  const VcDims& d = e->d;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  switch (site) {
    case VC_SITE_PHIXY: case VC_SITE_NU: return true;
    case VC_SITE_DNU: return d.with_dnu != 0;
    case VC_SITE_SHAPE_INV: return d.noise == VC_NOISE_NB;
    case VC_SITE_LOGGAMMA: case VC_SITE_LOGBETA: case VC_SITE_NUOMEGA: return vel;
    case VC_SITE_RHO_REAL: return vel && d.guide == VC_GUIDE_LRMN;
  }
  return false;
}

static void build_layout(vc_engine* e) {
  VcDims& d = e->d;
  vc_layout& L = e->layout;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  for (int i = 0; i < VC_P_COUNT; ++i) { L.offset[i] = -1; L.size[i] = 0; }
  for (int i = 0; i < VC_E_COUNT; ++i) { L.eps_offset[i] = -1; L.eps_size[i] = 0; }
  L.header = 4;
  long long off = L.header;
  auto add = [&](int id, long long n) { L.offset[id] = off; L.size[id] = n; off += n; };
  add(VC_P_NU_LOCS, (long long)d.Ng * d.Nh);
  add(VC_P_NU_USCALES, (long long)d.Ng * d.Nh);
  if (d.with_dnu) add(VC_P_DNU_LOCS, (long long)d.Nb * d.Ng);
  if (vel) {
    add(VC_P_LOGBETA_LOCS, d.Ng);
    add(VC_P_LOGBETA_USCALES, d.Ng);
    if (!lrmn) {
      add(VC_P_LOGGAMMA_LOCS, d.Ng);
      add(VC_P_LOGGAMMA_USCALES, d.Ng);
      add(VC_P_NUOMEGA_LOCS, d.NW);
      add(VC_P_NUOMEGA_USCALES, d.NW);
    } else {
      add(VC_P_LRMN_LOC, d.M);
      add(VC_P_LRMN_UCOV_FACTOR, (long long)d.M * d.R);
      add(VC_P_LRMN_UCOV_DIAG, d.M);
      add(VC_P_RHO_REAL_LOC, d.Ng);
    }
  }
  if (d.noise == VC_NOISE_NB) add(VC_P_SHAPE_INV_ULOCS, d.Ng);
  L.n_global = off - L.header;
  add(VC_P_PHIXY_LOCS, 2LL * d.Nc);
  L.n_local = 2LL * d.Nc;
  L.total = off;
  long long eo = 0;
  auto adde = [&](int id, long long n) { L.eps_offset[id] = eo; L.eps_size[id] = n; eo += n; };
  if (vel && !lrmn) { adde(VC_E_LOGGAMMA, d.Ng); adde(VC_E_LOGBETA, d.Ng); }
  if (lrmn) { adde(VC_E_LRMN_W, d.R); adde(VC_E_LRMN_D, d.M); }
  adde(VC_E_NU, (long long)d.Ng * d.Nh);
  if (lrmn) adde(VC_E_LOGBETA, d.Ng);
  if (vel && !lrmn) adde(VC_E_NUOMEGA, d.NW);
  if (eo & 1) eo++;       // phi_xy pairs start at an even index: (x, y) of a cell are the two normals of ONE Philox block
  L.eps_n_global = eo;
  adde(VC_E_PHIXY, 2LL * d.Nc);
  L.eps_total = eo;
  for (int i = 0; i < VC_P_COUNT; ++i) d.poff[i] = L.offset[i];
  for (int i = 0; i < VC_E_COUNT; ++i) d.eoff[i] = L.eps_offset[i];
  d.eps_n_global = L.eps_n_global;
  d.eps_total = L.eps_total;
}

extern "C" int vc_abi_version(void) { return VC_ABI_VERSION; }

extern "C" const char* vc_last_error(const vc_engine* e) {
  return e ? e->err.c_str() : g_create_error.c_str();
}

extern "C" int vc_create(const vc_config* c, vc_engine** out) {
  auto bad = [&](const char* m) { g_create_error = m; return VC_ERR_ARG; };
  if (!c || !out) return bad("vc_create: null argument");
  if (c->abi_version != VC_ABI_VERSION) return bad("vc_create: abi_version mismatch");
  if (c->model != VC_MODEL_PHASE && c->model != VC_MODEL_VELOCITY) return bad("vc_create: bad model");
  if (c->noise < 0 || c->noise > 2) return bad("vc_create: bad noise model");
  if (c->Ng <= 0 || c->Nc_local <= 0) return bad("vc_create: Ng and Nc_local must be positive");
  if (c->Ng > (1 << 24) || c->Nc_local > (1LL << 30)) return bad("vc_create: problem too large");
  const bool vel = c->model == VC_MODEL_VELOCITY;
  // Outside the compiled fast set (H <= 3, <= 4 batches, omega harmonics <= 3, <= 64 angular-speed coefficients, LRMN rank <= 8)
  // the run-time-sized kernel set takes over (vc_generic_kernels.hip).  What is left as a bound is memory: the likelihood
  // kernel keeps 2 K rows of per-gene state in the LDS of one wave (K = 2 H + 1 + Nb <= 150: 150 KB of the CU's 160).
  if (c->n_harmonics < 1) return bad("vc_create: n_harmonics must be >= 1");
  if (c->with_delta_nu && c->Nb < 1) return bad("vc_create: with_delta_nu needs Nb >= 1");
  bool generic = c->n_harmonics > VC_MAXH;
  const bool generic_nb = c->with_delta_nu && c->Nb > VC_MAXNB;      // (not with a one-hot design matrix: vc_finalize decides)
  if (vel) {
    if (c->n_harmonics_w < 0) return bad("vc_create: negative omega harmonics");
    if (c->Nx < 1) return bad("vc_create: bad Nx");
    if (c->guide == VC_GUIDE_LRMN && c->lrmn_rank < 1) return bad("vc_create: lrmn_rank must be >= 1");
    generic = generic || c->n_harmonics_w > VC_MAXH || c->Nx * (2 * c->n_harmonics_w + 1) > VC_MAX_NW ||
              (c->guide == VC_GUIDE_LRMN && c->lrmn_rank > VC_MAX_RANK);
    if ((long long)c->Nx * (2 * c->n_harmonics_w + 1) > 8192 || c->lrmn_rank > 4096) {
      g_create_error = "more than 8192 angular-speed coefficients / LRMN rank > 4096";
      return VC_ERR_UNSUPPORTED;
    }
  }
  if (2 * c->n_harmonics + 1 + (c->with_delta_nu ? c->Nb : 0) > 150) {
    g_create_error = "2 n_harmonics + 1 + Nb > 150: the per-gene state of one wave no longer fits the LDS";
    return VC_ERR_UNSUPPORTED;
  }
  if (c->world_size < 1 || c->rank < 0 || c->rank >= c->world_size) return bad("vc_create: bad rank/world_size");
  vc_engine* e = new (std::nothrow) vc_engine();
  if (!e) return bad("vc_create: out of host memory");
  e->cfg = *c;
  VcDims& d = e->d;
  d.generic = (generic || generic_nb) ? 1 : 0;      // provisional: vc_finalize
  e->generic_needed = generic;
  e->generic_nb = generic_nb;
  d.Ng = (int)c->Ng;
  d.gpl = 4; d.gbw = 256;
  d.nGB = (d.Ng + d.gbw - 1) / d.gbw;
  d.Ng_pad = d.nGB * d.gbw;       // provisional; fixed by vc_finalize
  d.Nc = (int)c->Nc_local;
  d.cell_offset = c->cell_offset;
  d.H = c->n_harmonics; d.Nh = 2 * d.H + 1;
  d.Hw = vel ? c->n_harmonics_w : 0; d.Nhw = 2 * d.Hw + 1;
  d.with_dnu = c->with_delta_nu ? 1 : 0;
  d.Nb = d.with_dnu ? c->Nb : 0;
  d.Nx = vel ? c->Nx : 0;
  d.NW = d.Nx * d.Nhw;
  d.model = c->model;
  d.guide = vel ? c->guide : VC_GUIDE_MEANFIELD;
  d.noise = c->noise;
  d.R = (vel && d.guide == VC_GUIDE_LRMN) ? c->lrmn_rank : 0;
  d.M = d.Ng + d.NW;
  d.K = d.Nh + d.Nb;
  d.Kq = d.K; d.nbk = d.Nb; d.onehot = 0;      // provisional: vc_finalize
  d.ctw = 2 * ((vc_rec_pairs(d.H, d.Nb, false) + VC_REC_PAD - 1) / VC_REC_PAD * VC_REC_PAD);   // provisional (vc_finalize: the S+U kernel's record is longer)
  d.pw_inline = 0;
  d.cond = 0;
  d.root_w = c->rank == 0 ? 1.f : 0.f;
  d.gamma_alpha = c->gamma_alpha; d.gamma_beta = c->gamma_beta;
  d.sigma_ln_s = c->sigma_ln_s; d.sigma_ln_u = c->sigma_ln_u;
  d.rho_mean = c->rho_mean; d.rho_std = c->rho_std; d.rho_scale = c->rho_scale;
  build_layout(e);
  *out = e;
  return VC_OK;
}

extern "C" void vc_destroy(vc_engine* e) {
  if (!e) return;
  if (e->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->comm);
  if (e->p2p_connected)
    for (int q = 0; q < e->p2p.world; ++q)
      if (q != e->p2p.rank && e->p2p.region[q]) (void)hipIpcCloseMemHandle(e->p2p.region[q]);
  if (e->p2p_own) (void)hipFree(e->p2p_own);
  for (void* p : e->allocs) (void)hipFree(p);
  for (auto& c : e->src) c.release();
  for (auto& pr : e->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pt : e->particles) {
    if (pt.st) (void)hipStreamDestroy(pt.st);
    if (pt.main_done) (void)hipEventDestroy(pt.main_done);
    if (pt.done) (void)hipEventDestroy(pt.done);
  }
  if (e->ev_params) (void)hipEventDestroy(e->ev_params);
  if (e->ev_main0) (void)hipEventDestroy(e->ev_main0);
  delete e;
}

extern "C" int vc_get_layout(const vc_engine* e, vc_layout* out) {
  if (!e || !out) return VC_ERR_ARG;
  *out = e->layout;
  return VC_OK;
}

extern "C" int vc_set_tuning(vc_engine* e, const vc_tuning* t) {
  if (!e) return VC_ERR_ARG;
  if (e->finalized || e->have_counts) return e->fail(VC_ERR_STATE, "vc_set_tuning after the counts were handed over");
  vc_tuning z{};
  if (t) z = *t;
  auto in = [](int v, std::initializer_list<int> ok) { for (int o : ok) if (v == o) return true; return false; };
  if (!in(z.genes_per_lane, {0, 4, 8})) return e->fail(VC_ERR_ARG, "vc_set_tuning: genes_per_lane must be 0, 4 or 8");
  if (z.blocks_per_cu < 0 || z.cells_per_wave < 0 || z.pass_min_cw < 0) return e->fail(VC_ERR_ARG, "vc_set_tuning: negative value");
  if (z.n_pass_shares < 0 || z.n_pass_shares > 4) return e->fail(VC_ERR_ARG, "vc_set_tuning: n_pass_shares must be 0..4");
  for (int p = 0; p < z.n_pass_shares && z.n_pass_shares >= 2; ++p)
    if (!(z.pass_shares[p] > 0.f)) return e->fail(VC_ERR_ARG, "vc_set_tuning: pass_shares must be positive");
  if (!in(z.tail_cells, {0, 256, 512, 1024})) return e->fail(VC_ERR_ARG, "vc_set_tuning: tail_cells must be 0, 256, 512 or 1024");
  if (!in(z.count_storage, {0, 1}) || !in(z.host_hist, {0, 1}) || !in(z.hist_dense, {0, 1, 2}) || !in(z.pw_inline, {0, 1, 2}) ||
      !in(z.no_tail2, {0, 1}) || !in(z.no_tail_merged, {0, 1}) || !in(z.force_generic, {0, 1}) || !in(z.particles_layout, {0, 1, 2}) ||
      !in(z.dense_batches, {0, 1}) || !in(z.no_tail_spec, {0, 1}) || !in(z.no_pw_lane, {0, 1}) || !in(z.p2p_separate, {0, 1}) || !in(z.p2p_one_launch, {0, 1}))
    return e->fail(VC_ERR_ARG, "vc_set_tuning: a switch is outside its documented values");
  if (z.p2p_timeout_s < 0.f) return e->fail(VC_ERR_ARG, "vc_set_tuning: negative p2p_timeout_s");
  e->tun = z;
  e->d.generic = (z.force_generic || e->generic_needed || e->generic_nb) ? 1 : 0;      // provisional: vc_finalize
  if (z.p2p_timeout_s > 0.f) e->p2p_timeout_s = z.p2p_timeout_s;
  return VC_OK;
}

extern "C" int vc_get_tuning(const vc_engine* e, vc_tuning* out) {
  if (!e || !out) return VC_ERR_ARG;
  *out = e->tun;
  return VC_OK;
}

extern "C" int vc_dbg_dump_times(vc_engine* e, const char* path) {
  if (!e || !path) return VC_ERR_ARG;
#ifdef VC_DBG_TIMES
  if (!e->b.dbg) return e->fail(VC_ERR_STATE, "vc_dbg_dump_times before vc_finalize");
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(VC_DBG_WORDS(e->d.n_main_wg));
  HIPCHK(e, hipMemcpy(h.data(), e->b.dbg, h.size() * 8, hipMemcpyDeviceToHost));
  FILE* f = fopen(path, "wb");
  if (!f) return e->fail(VC_ERR_ARG, "vc_dbg_dump_times: cannot open %s", path);
  fwrite(h.data(), 8, h.size(), f);
  fclose(f);
  return VC_OK;
#else
  return e->fail(VC_ERR_UNSUPPORTED, "vc_dbg_dump_times: this build of the library carries no time stamps (-DVC_DBG_TIMES)");
#endif
}

extern "C" int vc_dbg_signature(const vc_engine* e, int32_t* out, int n) {
  if (!e || !out || n < VC_SIG_INTS) return VC_ERR_ARG;
  if (!e->finalized) return VC_ERR_STATE;
  const VcSig s = vc_sig_of(e->d);
  int i = 0;
#define VC_SIG_PUT(f) out[i++] = s.f;
  VC_SIG_FIELDS(VC_SIG_PUT)
#undef VC_SIG_PUT
  out[i++] = (int32_t)s.cond;
  return VC_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int vc_set_counts(vc_engine* e, const float* S, const float* U, int64_t gs, int64_t cs, int on_device) {
  if (!e) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_set_counts after vc_finalize");
  VcDims& d = e->d;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  if (!S || (vel && !U)) return e->fail(VC_ERR_ARG, "vc_set_counts: missing matrix");
  if (gs <= 0 || cs <= 0) return e->fail(VC_ERR_ARG, "vc_set_counts: strides must be positive");
  const size_t span = (size_t)(d.Ng - 1) * gs + (size_t)(d.Nc - 1) * cs + 1;
  const float* src[2] = {S, vel ? U : nullptr};
  std::vector<float>* hcopy[2] = {&e->hS, &e->hU};
  e->host_hist = e->tun.host_hist != 0;
  long long ngs = gs, ncs = cs;
  for (int m = 0; m < 2; ++m) {
    e->src[m].release();
    if (!src[m]) continue;
    e->src[m].kind = 1;
    if (e->host_hist && d.noise != VC_NOISE_LOGNORMAL) {     // checker path: host copy of the raw counts
      std::vector<float> tmp(span);
      if (on_device) HIPCHK(e, hipMemcpy(tmp.data(), src[m], span * sizeof(float), hipMemcpyDeviceToHost));
      else memcpy(tmp.data(), src[m], span * sizeof(float));
      hcopy[m]->swap(tmp);
    }
    if (on_device) {
      // no copy: the caller's buffer is read by vc_finalize (it must stay valid until vc_finalize has returned)
      e->src[m].dense = src[m];
      continue;
    }
    // host memory: upload THIS shard only, row by row when one of the strides is 1 (a column slice of a gene-major
    // matrix spans almost the whole matrix, but only Nc_local values per row are this rank's)
    float* dev = nullptr;
    if (gs == 1 && cs >= d.Ng) {
      HIPCHK(e, hipMalloc((void**)&dev, (size_t)d.Ng * d.Nc * sizeof(float)));
      e->src[m].owned[0] = dev;
      HIPCHK(e, hipMemcpy2D(dev, (size_t)d.Ng * 4, src[m], (size_t)cs * 4, (size_t)d.Ng * 4, d.Nc, hipMemcpyHostToDevice));
      ngs = 1; ncs = d.Ng;
    } else if (cs == 1 && gs >= d.Nc) {
      HIPCHK(e, hipMalloc((void**)&dev, (size_t)d.Ng * d.Nc * sizeof(float)));
      e->src[m].owned[0] = dev;
      HIPCHK(e, hipMemcpy2D(dev, (size_t)d.Nc * 4, src[m], (size_t)gs * 4, (size_t)d.Nc * 4, d.Ng, hipMemcpyHostToDevice));
      ngs = d.Nc; ncs = 1;
    } else {
      HIPCHK(e, hipMalloc((void**)&dev, span * sizeof(float)));
      e->src[m].owned[0] = dev;
      HIPCHK(e, hipMemcpy(dev, src[m], span * sizeof(float), hipMemcpyHostToDevice));
    }
    e->src[m].dense = dev;
  }
  e->gs = gs; e->cs = cs;            // strides of the host copies (checker path)
  e->dgs = ngs; e->dcs = ncs;        // strides of the device sources
  e->have_counts = true;
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_set_counts_csr(vc_engine* e, int which, const int64_t* indptr, const int32_t* indices, const float* data,
                                 int64_t nnz, int on_device) {
  if (!e) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_set_counts_csr after vc_finalize");
  VcDims& d = e->d;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  if (which < 0 || which > 1 || (which == 1 && !vel)) return e->fail(VC_ERR_ARG, "vc_set_counts_csr: bad matrix id %d", which);
  if (!indptr || nnz < 0 || (nnz > 0 && (!indices || !data))) return e->fail(VC_ERR_ARG, "vc_set_counts_csr: null array");
  auto& c = e->src[which];
  c.release();
  if (on_device) {
    c.indptr = (const long long*)indptr; c.indices = indices; c.data = data;
  } else {
    if (indptr[0] != 0 || indptr[d.Nc] != nnz) return e->fail(VC_ERR_ARG, "vc_set_counts_csr: indptr[0] / indptr[Nc] do not match nnz");
    for (int64_t r = 0; r < d.Nc; ++r)
      if (indptr[r + 1] < indptr[r]) return e->fail(VC_ERR_ARG, "vc_set_counts_csr: indptr is not non-decreasing");
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    HIPCHK(e, hipMalloc(&p0, (size_t)(d.Nc + 1) * sizeof(long long)));
    c.owned[0] = p0;
    HIPCHK(e, hipMalloc(&p1, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
    c.owned[1] = p1;
    HIPCHK(e, hipMalloc(&p2, (size_t)std::max<int64_t>(nnz, 1) * sizeof(float)));
    c.owned[2] = p2;
    HIPCHK(e, hipMemcpy(p0, indptr, (size_t)(d.Nc + 1) * sizeof(long long), hipMemcpyHostToDevice));
    if (nnz > 0) {
      HIPCHK(e, hipMemcpy(p1, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(e, hipMemcpy(p2, data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
    }
    c.indptr = (const long long*)p0; c.indices = (const int*)p1; c.data = (const float*)p2;
  }
  c.kind = 2;
  c.nnz = nnz;
  e->host_hist = false;
  e->have_counts = e->src[0].kind != 0 && (!vel || e->src[1].kind != 0);
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_set_cell_data(vc_engine* e, const float* count_factor, const float* D, const float* Db,
                                const float* phixy_prior) {
  if (!e) return VC_ERR_ARG;
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_set_cell_data after vc_finalize");
  VcDims& d = e->d;
  if (!count_factor || !phixy_prior) return e->fail(VC_ERR_ARG, "vc_set_cell_data: null count_factor / phixy_prior");
  if (d.Nx > 0 && !D) return e->fail(VC_ERR_ARG, "vc_set_cell_data: D required for the velocity model");
  if (d.Nb > 0 && !Db) return e->fail(VC_ERR_ARG, "vc_set_cell_data: Db required when with_delta_nu");
  float *cf, *dm = nullptr, *dbm = nullptr, *pxy;
  TRY(e->dalloc(&cf, d.Nc));
  TRY(e->dalloc(&pxy, 2 * (size_t)d.Nc));
  HIPCHK(e, hipMemcpy(cf, count_factor, sizeof(float) * d.Nc, hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(pxy, phixy_prior, sizeof(float) * 2 * d.Nc, hipMemcpyHostToDevice));
  if (d.Nx > 0) {
    TRY(e->dalloc(&dm, (size_t)d.Nx * d.Nc));
    HIPCHK(e, hipMemcpy(dm, D, sizeof(float) * d.Nx * d.Nc, hipMemcpyHostToDevice));
    e->D_all_ones = d.Nx == 1;
    for (long long i = 0; i < (long long)d.Nc && e->D_all_ones; ++i) e->D_all_ones = D[i] == 1.f;
    try { e->hD.assign(D, D + (size_t)d.Nx * d.Nc); } catch (...) { return e->fail(VC_ERR_ARG, "out of host memory"); }
  }
  if (d.Nb > 0) {
    TRY(e->dalloc(&dbm, (size_t)d.Nb * d.Nc));
    HIPCHK(e, hipMemcpy(dbm, Db, sizeof(float) * d.Nb * d.Nc, hipMemcpyHostToDevice));
  }
  e->b.cf = cf; e->b.Dm = dm; e->b.Dbm = dbm; e->b.pxy = pxy;
  if (d.Nb > 0) {
    try { e->hDb.assign(Db, Db + (size_t)d.Nb * d.Nc); } catch (...) { return e->fail(VC_ERR_ARG, "out of host memory"); }
  }
  e->have_cells = true;
  return VC_OK;
}

static long long prior_size(const vc_engine* e, int which) {
  const VcDims& d = e->d;
  switch (which) {
    case VC_PRIOR_MU_NU: case VC_PRIOR_SD_NU: return (long long)d.Ng * d.Nh;
    case VC_PRIOR_MU_NUOMEGA: case VC_PRIOR_SD_NUOMEGA: return d.NW;
    case VC_PRIOR_SD_DNU: return (long long)d.Nb * d.Ng;
    default: return d.Ng;
  }
}

extern "C" int vc_set_prior(vc_engine* e, int which, const float* data, int64_t n) {
  if (!e) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_set_prior after vc_finalize");
  if (which < 0 || which >= VC_PRIOR_COUNT || !data) return e->fail(VC_ERR_ARG, "vc_set_prior: bad id / null data");
  if (n != prior_size(e, which))
    return e->fail(VC_ERR_ARG, "vc_set_prior(%d): expected %lld values, got %lld", which, prior_size(e, which), (long long)n);
  for (int64_t i = 0; i < n; ++i) {
    const bool is_sd = which == VC_PRIOR_SD_NU || which == VC_PRIOR_SD_GAMMA || which == VC_PRIOR_SD_BETA ||
                       which == VC_PRIOR_SD_NUOMEGA || which == VC_PRIOR_SD_DNU;
    if (!std::isfinite(data[i]) || (is_sd && !(data[i] > 0.f)))
      return e->fail(VC_ERR_ARG, "vc_set_prior(%d): value %lld is not finite / not positive", which, (long long)i);
  }
  e->h_prior[which].assign(data, data + n);
  e->prior_set[which] = true;
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_set_conditioned(vc_engine* e, int site, const float* values, int64_t n) {
  if (!e) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_set_conditioned after vc_finalize");
  if (site < 0 || site >= VC_SITE_COUNT || !values) return e->fail(VC_ERR_ARG, "vc_set_conditioned: bad site / null data");
  if (!site_exists(e, site)) return e->fail(VC_ERR_ARG, "vc_set_conditioned: site %d does not exist in this model", site);
  if (n != site_size(e, site))
    return e->fail(VC_ERR_ARG, "vc_set_conditioned(%d): expected %lld values, got %lld", site, site_size(e, site), (long long)n);
  e->h_cond[site].assign(values, values + n);
  e->d.cond |= (1u << site);
  return VC_OK;
  VC_GUARD_END(e)
}

template <class T>
static int upload(vc_engine* e, const std::vector<T>& h, const T** dev) {
  T* p = nullptr;
  TRY(e->dalloc(&p, h.size()));
  if (!h.empty()) HIPCHK(e, hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = p;
  return VC_OK;
}

static int finalize_impl(vc_engine* e, void* hip_stream);
extern "C" int vc_finalize(vc_engine* e, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  // A finalize that failed half way has consumed part of its inputs (host copies of the counts are released as they are
  // ingested) and left workspaces behind: it is not restartable on the same engine -- say so instead of running on freed inputs.
  if (e->finalize_failed)
    return e->fail(VC_ERR_STATE, "vc_finalize failed earlier on this engine: vc_destroy it and create a new one");
  auto guarded = [&]() -> int {
    VC_GUARD_BEGIN
    return finalize_impl(e, hip_stream);
    VC_GUARD_END(e)
  };
  const int rc = guarded();
  if (rc != VC_OK && e->finalize_started) e->finalize_failed = true;   // before that mark nothing was touched (inputs missing)
  return rc;
}

static int finalize_impl(vc_engine* e, void* hip_stream) {
  if (e->finalized) return e->fail(VC_ERR_STATE, "vc_finalize called twice");
  if (!e->have_counts || !e->have_cells) return e->fail(VC_ERR_STATE, "vc_finalize: counts / cell data not set");
  hipStream_t st = (hipStream_t)hip_stream;
  VcDims& d = e->d;
  VcBufs& b = e->b;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  const bool nb = d.noise == VC_NOISE_NB;
  // required priors
  const int need_all[] = {VC_PRIOR_MU_NU, VC_PRIOR_SD_NU};
  for (int w : need_all)
    if (!e->prior_set[w]) return e->fail(VC_ERR_STATE, "vc_finalize: prior %d not set", w);
  if (vel)
    for (int w = VC_PRIOR_MU_GAMMA; w <= VC_PRIOR_SD_NUOMEGA; ++w)
      if (!e->prior_set[w]) return e->fail(VC_ERR_STATE, "vc_finalize: prior %d not set", w);
  if (!vel && d.with_dnu && !e->prior_set[VC_PRIOR_SD_DNU])
    return e->fail(VC_ERR_STATE, "vc_finalize: sd_dnu prior not set");
  e->finalize_started = true;

  // Batch offsets.  The reference's design matrix is one-hot by construction (make_design_matrix, preprocessing.py:65-93): then the
  // offset of a cell's batch is folded into the constant harmonic per workgroup of the likelihood kernel (cells ordered by batch,
  // workgroups aligned to the batch boundaries), the kernel without batch terms runs for ANY number of batches and the gradient
  // of an offset is a sum over its batch's workgroups.  Any other matrix keeps the dense contraction (<= VC_MAXNB batches on the
  // fast kernel set, the run-time-sized set beyond).
  std::vector<int> bat_id, bat_pos, bat_ord, bat_len;
  bool bat_sorted = true;
  d.onehot = 0;
  if (d.with_dnu && d.Nb >= 1 && !e->tun.dense_batches && !e->tun.force_generic && !e->generic_needed &&
      vc_onehot_batches(e->hDb.data(), d.Nb, d.Nc, bat_id)) {
    d.onehot = 1;
    bat_sorted = vc_order_by_batch(bat_id, d.Nb, bat_pos, bat_ord, bat_len);
    // the condition design D (Nx, Nc) is one-hot and CONSTANT within every batch (the tutorials pass the same matrix for both:
    // Tutorial_Aissa_PC9_TwoSample cell 40)?  Then every workgroup of the likelihood kernel lies inside one condition too.
    e->bat_cond.assign((size_t)d.Nb, -1);
    bool okc = vel && d.Nx >= 1 && e->hD.size() == (size_t)d.Nx * d.Nc;
    for (long long c = 0; c < (long long)d.Nc && okc; ++c) {
      int xc = -1;
      for (int x = 0; x < d.Nx; ++x) {
        const float v = e->hD[(size_t)x * d.Nc + c];
        if (v == 1.f && xc < 0) xc = x;
        else if (v != 0.f) okc = false;
      }
      if (xc < 0) okc = false;
      int& bc = e->bat_cond[(size_t)bat_id[(size_t)c]];
      if (bc < 0) bc = xc;
      else if (bc != xc) okc = false;
    }
    for (int& v : e->bat_cond) if (v < 0) v = 0;       // (a batch without cells on this rank)
    if (!okc) e->bat_cond.clear();
  }
  d.generic = (e->tun.force_generic || e->generic_needed || (e->generic_nb && !d.onehot)) ? 1 : 0;
  d.Kq = d.onehot ? d.Nh : d.K;
  d.nbk = d.onehot ? 0 : d.Nb;
  const int knb = d.nbk;            // the NB template argument of the likelihood kernel
  // kernel kind
  if (!vel) d.kind = VC_KIND_PHASE;
  else {
    const bool vu = cond(e, VC_SITE_PHIXY) && cond(e, VC_SITE_NU) && (!d.with_dnu || cond(e, VC_SITE_DNU)) &&
                    (!nb || cond(e, VC_SITE_SHAPE_INV));
    d.kind = vu ? VC_KIND_VU : VC_KIND_VFULL;
  }
  auto nq_of = [&](int kind) { return kind == VC_KIND_PHASE ? d.Kq + 1 : (kind == VC_KIND_VFULL ? d.Kq + 3 : 2); };
  auto nco_of = [&](int kind) { return kind == VC_KIND_VFULL ? 3 : 1; };
  d.nq = nq_of(d.kind);
  d.nco = nco_of(d.kind);
  // genes per lane: 8 amortises the per-cell work over twice the genes and is chosen whenever that instantiation's
  // per-gene state (latents + accumulators) fits 2 waves per SIMD without scratch, which the code object itself
  // tells (private segment size 0); else 4.
  d.gpl = d.generic ? 2 : 4;
  const size_t max_scratch = 0;      // asm-issued count loads: a spilled destination tuple would be stored before its data has landed
  if (!d.generic) {
    const void* k8 = nullptr;
    hipFuncAttributes fa;
    if (vc_find_main_kernel(d.H, knb, d.kind, d.noise, 8, 0, nullptr, &k8) && k8 &&
        hipFuncGetAttributes(&fa, k8) == hipSuccess && fa.localSizeBytes <= max_scratch)
      d.gpl = 8;
    if (d.gpl == 8) {
      // Small shards (round 3, profiles/r03_small_shard.md): with few cells per wave the per-wave prologue / epilogue of the
      // 8-genes-per-lane kernel (twice the gene-table loads, twice the rows to combine) outweighs what it saves per cell --
      // 6 250 cells x 2 000 genes: 22.5 vs 24.7 us (S+U), 16.8 vs 17.8 (U only); 12 500: 35.6 vs 36.8 (S+U) but 26.1 vs 23.7
      // (U only); 25 000: 61.7 vs 62.9; 50 000: 8 genes per lane wins.  Decided from the cells a wave would get under the
      // 8-genes-per-lane tiling -- a pure function of (cells, genes, kernel kind, occupancy, CUs), like the tiling itself.
      int bpc8 = 0, n_cu = 256, dev = 0;
      hipDeviceProp_t prop;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc8, k8, 256, 0) == hipSuccess && bpc8 >= 1 &&
          hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        if (prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        const int ngb8 = (d.Ng + 511) / 512;
        // decided from a RANK-INVARIANT cell count (the largest balanced shard): balanced shards differ by one cell, and two
        // ranks on opposite sides of the threshold would disagree on Ng_pad -- and with it on the exchange buffer's layout
        const long long world = e->cfg.world_size, ncg = e->cfg.Nc_global > 0 ? e->cfg.Nc_global : d.Nc;
        const int shard_cells = world > 1 ? (int)((ncg + world - 1) / world) : d.Nc;
        const VcTiling t8 = vc_tile_cells(shard_cells, ngb8, n_cu, bpc8, VC_WAVES, 0, nullptr, 12);
        const int small_cw = d.kind == VC_KIND_VFULL ? 52 : 10;
        // Round 4, S+U kernel on ONE rank: the 8-genes-per-lane kernel can emit the nu_omega partials itself (room in the LDS),
        // which makes the step two launches instead of three; the 4-genes-per-lane kernel can do that only where a wave has
        // <= 12 cells (its reduction tiles fill the LDS: profiles/r04_small_shard.md).  Measured at 2 000 genes, 6 250 ... 20 000
        // cells: K_main + 4-5 us, the step - 1.3 ... - 3.4 us (- 3 ... - 7 %); equal at 25 000.  So: 4 genes per lane only where
        // that kernel keeps the step at two launches as well.
        bool two_launch8 = false;
        if (d.kind == VC_KIND_VFULL && world == 1 && VC_PW_INLINE && d.NW >= 1 && d.NW <= VC_PWQ) {
          const void* k4 = nullptr;
          int bpc4 = 0;
          if (e->tun.pw_inline != 1 && !e->tun.no_tail2 &&
              vc_find_main_kernel(d.H, knb, d.kind, d.noise, 4, 0, nullptr, &k4) && k4 &&
              hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc4, k4, 256, 0) == hipSuccess && bpc4 >= 1) {
            const VcTiling t4 = vc_tile_cells(shard_cells, (d.Ng + 255) / 256, n_cu, bpc4, VC_WAVES, 0, nullptr, 12);
            two_launch8 = t4.cw > 12;
          }
        }
        if (t8.cw <= small_cw && !two_launch8) d.gpl = 4;
      }
    }
  }
  if (!d.generic && e->tun.genes_per_lane) {
    if (e->tun.genes_per_lane == 4) d.gpl = 4;
    if (e->tun.genes_per_lane == 8) {       // honoured only where the 8-genes-per-lane kernel may run at all (no scratch: see above)
      const void* k8 = nullptr;
      hipFuncAttributes fa;
      if (vc_find_main_kernel(d.H, knb, d.kind, d.noise, 8, 0, nullptr, &k8) && k8 &&
          hipFuncGetAttributes(&fa, k8) == hipSuccess && fa.localSizeBytes <= max_scratch)
        d.gpl = 8;
    }
  }
  d.gbw = 64 * d.gpl;
  d.nGB = (d.Ng + d.gbw - 1) / d.gbw;
  d.Ng_pad = d.nGB * d.gbw;
  const void* main_kernel = nullptr;
  d.c16 = 0;
  if (d.generic) {
    // the run-time-sized set: one wave per workgroup, 2 genes per lane (gene blocks of 128), float32 counts
    static const char* gen_names[3][3] = {{"generic_phase_nb", "generic_phase_poisson", "generic_phase_lognormal"},
                                          {"generic_vfull_nb", "generic_vfull_poisson", "generic_vfull_lognormal"},
                                          {"generic_vu_nb", "generic_vu_poisson", "generic_vu_lognormal"}};
    e->main_fn = vc_find_generic_main_kernel(d.kind, d.noise, &main_kernel);
    e->main_name = gen_names[d.kind][d.noise];
    if (d.kind == VC_KIND_VU) e->phase_fn = vc_find_generic_main_kernel(VC_KIND_PHASE, d.noise, nullptr);
  } else {
    e->main_fn = vc_find_main_kernel(d.H, knb, d.kind, d.noise, d.gpl, 0, &e->main_name, &main_kernel);
  }
  if (!e->main_fn) return e->fail(VC_ERR_UNSUPPORTED, "no likelihood kernel for H=%d Nb=%d kind=%d noise=%d", d.H, knb, d.kind, d.noise);
  if (d.kind == VC_KIND_VU && !d.generic) {
    e->phase_fn = vc_find_main_kernel(d.H, knb, VC_KIND_PHASE, d.noise, d.gpl, 0, nullptr, nullptr);
    if (!e->phase_fn) return e->fail(VC_ERR_UNSUPPORTED, "no S-only kernel for the hoisted term");
  }
  // HBM layout of the counts: [gene block][cell][gbw], zero padded in genes; the per-gene count histograms are built on
  // the device in the same pass (dense bins + overflow list), so that no dense matrix crosses back to the host
  const bool want_hist = d.noise != VC_NOISE_LOGNORMAL;
  const bool dev_hist = want_hist && !e->host_hist;
  const unsigned OVF_CAP = 1u << 22;
  unsigned* tab[2] = {nullptr, nullptr};
  float* ovf_val[2] = {nullptr, nullptr};
  int* ovf_gene[2] = {nullptr, nullptr};
  unsigned* ovf_n = nullptr;
  int* bad = nullptr;
  size_t transient = 0;
  auto tmalloc = [&](void** p, size_t bytes) -> int {
    hipError_t er = hipMalloc(p, bytes);
    if (er != hipSuccess) return e->fail(VC_ERR_HIP, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(er));
    transient += bytes;
    return VC_OK;
  };
  auto free_transients = [&]() {
    for (int m = 0; m < 2; ++m) {
      if (tab[m]) (void)hipFree(tab[m]);
      if (ovf_val[m]) (void)hipFree(ovf_val[m]);
      if (ovf_gene[m]) (void)hipFree(ovf_gene[m]);
      tab[m] = nullptr; ovf_val[m] = nullptr; ovf_gene[m] = nullptr;
    }
    if (ovf_n) (void)hipFree(ovf_n);
    if (bad) (void)hipFree(bad);
    ovf_n = nullptr; bad = nullptr;
  };
  // cells ordered by batch (one-hot batches that are not contiguous): the blocked counts, the cell table and the per-cell partial
  // rows live in that order; every per-cell kernel maps its cell through cell_pos (vc_common.h: vc_pos)
  const int* dev_ord = nullptr;
  b.cell_pos = nullptr;
  if (d.onehot && !bat_sorted) {
    TRY(upload(e, bat_pos, &b.cell_pos));
    TRY(upload(e, bat_ord, &dev_ord));
  }
  TRY(tmalloc((void**)&ovf_n, 2 * sizeof(unsigned)));
  TRY(tmalloc((void**)&bad, 4 * sizeof(int)));
  HIPCHK(e, hipMemsetAsync(ovf_n, 0, 2 * sizeof(unsigned), st));
  HIPCHK(e, hipMemsetAsync(bad, 0, 4 * sizeof(int), st));
  {
    const size_t blocked = (size_t)d.nGB * d.Nc * d.gbw;
    const float** dstp[2] = {&b.S, &b.U};
    for (int m = 0; m < 2; ++m) {
      *dstp[m] = nullptr;
      auto& c = e->src[m];
      if (c.kind == 0) continue;
      for (void* p : c.owned) if (p) transient += 0;   // owned uploads are released below
      float* packed = nullptr;
      TRY(e->dalloc(&packed, blocked));
      if (dev_hist) {
        int rc = tmalloc((void**)&tab[m], (size_t)d.Ng * VC_HIST_CAP * sizeof(unsigned));
        if (rc == VC_OK) rc = tmalloc((void**)&ovf_val[m], (size_t)OVF_CAP * sizeof(float));
        if (rc == VC_OK) rc = tmalloc((void**)&ovf_gene[m], (size_t)OVF_CAP * sizeof(int));
        if (rc != VC_OK) { free_transients(); return rc; }
        HIPCHK(e, hipMemsetAsync(tab[m], 0, (size_t)d.Ng * VC_HIST_CAP * sizeof(unsigned), st));
      }
      const int ln = d.noise == VC_NOISE_LOGNORMAL;
      if (c.kind == 1) {
        vc_launch_pack_counts(c.dense, packed, e->dgs, e->dcs, d.Ng, d.Nc, d.nGB, d.gbw, ln, tab[m], ovf_val[m], ovf_gene[m],
                              ovf_n + m, OVF_CAP, bad, dev_ord, st);
      } else {
        HIPCHK(e, hipMemsetAsync(packed, 0, blocked * sizeof(float), st));
        vc_launch_scatter_csr(c.indptr, c.indices, c.data, packed, d.Ng, d.Nc, d.gbw, ln, tab[m], ovf_val[m], ovf_gene[m],
                              ovf_n + m, OVF_CAP, bad, b.cell_pos, st);
      }
      HIPCHK(e, hipStreamSynchronize(st));
      HIPCHK(e, hipGetLastError());
      if (c.owned[0]) transient += (c.kind == 1 ? (size_t)d.Ng * d.Nc * 4 : (size_t)(d.Nc + 1) * 8 + (size_t)c.nnz * 8);
      *dstp[m] = packed;
    }
  }
  {
    int hbad[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemcpy(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost));
    if (hbad[1]) { free_transients(); return e->fail(VC_ERR_ARG, "vc_set_counts_csr: gene index outside [0, Ng)"); }
    if (hbad[0]) { free_transients(); return e->fail(VC_ERR_ARG, "count matrices must be finite and >= 0 (NaN / Inf / negative value found)"); }
    // Count storage: when every count of this rank's matrices is an integer <= 65535 (checked by the pass above) the
    // blocked layout is narrowed to uint16 -- half the bytes K_main streams per step; exact.  tuning.count_storage = 1 keeps
    // the reference's float32 (A/B measurements, tests); Lognormal noise stores log(k + 1) and stays float32.
    const bool want16 = e->tun.count_storage != 1 && d.noise != VC_NOISE_LOGNORMAL && !hbad[2] && !d.generic;
    if (want16) {
      const void* k16 = nullptr;
      const char* nm = nullptr;
      hipFuncAttributes fa;
      vc_main_launch_fn f16 = vc_find_main_kernel(d.H, knb, d.kind, d.noise, d.gpl, 1, &nm, &k16);
      vc_main_launch_fn p16 = d.kind == VC_KIND_VU ? vc_find_main_kernel(d.H, knb, VC_KIND_PHASE, d.noise, d.gpl, 1, nullptr, nullptr) : nullptr;
      if (f16 && k16 && (d.kind != VC_KIND_VU || p16) && hipFuncGetAttributes(&fa, k16) == hipSuccess && fa.localSizeBytes <= max_scratch) {
        const size_t blocked = (size_t)d.nGB * d.Nc * d.gbw;
        const float** dstp[2] = {&b.S, &b.U};
        for (int m = 0; m < 2; ++m) {
          if (!*dstp[m]) continue;
          unsigned short* p16buf = nullptr;
          int rc = e->dalloc(&p16buf, blocked);
          if (rc != VC_OK) { free_transients(); return rc; }
          vc_launch_counts_to_u16(*dstp[m], p16buf, (long long)blocked, st);
          HIPCHK(e, hipStreamSynchronize(st));
          transient += blocked * sizeof(float);      // the float32 layout lives until its uint16 copy exists
          e->dfree(*dstp[m]);
          *dstp[m] = reinterpret_cast<const float*>(p16buf);
        }
        d.c16 = 1;
        e->main_fn = f16;
        main_kernel = k16;
        if (p16) e->phase_fn = p16;
      }
    }
  }
  std::vector<int> tile_host, bat_chunk_host;      // the workgroup table / the batches' chunk ranges (one-hot batches: built with the tiling)
  // tiling: one balanced round.  The grid is sized to the workgroups the chip holds at once for
  // this kernel (occupancy x 256 CUs); each wave gets an equal share of the cells of its gene block,
  // so no partially filled last round is left over (a 2.04-round grid costs 3 rounds).
  {
    int n_cu = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(e, hipGetDevice(&dev));
    HIPCHK(e, hipGetDeviceProperties(&prop, dev));
    if (prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    const int wg_waves = d.generic ? 1 : VC_WAVES;
    const unsigned gen_dyn = d.generic ? (unsigned)(2 * d.K * 64 * 2 * sizeof(float)) : 0u;      // nu~ and gradient rows of one wave
    auto occupancy = [&](unsigned dyn_bytes, int* out) -> int {
      int bpc = 0;
      dyn_bytes += gen_dyn;
      if (dyn_bytes > 0) {
        // the per-gene state of the run-time-sized kernels / the staged W rows must fit the LDS a workgroup of THIS device may ask
        // for (gfx950: 160 KB; a part with 64 KB refuses here with the dimension named instead of failing at launch: ADVICE r4)
        const size_t lds_cap = std::max<size_t>(prop.sharedMemPerBlockOptin, prop.maxSharedMemoryPerMultiProcessor);
        hipError_t er = lds_cap > 0 && dyn_bytes > lds_cap ? hipErrorInvalidValue
                                                            : hipFuncSetAttribute(main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes);
        if (er != hipSuccess) {
          (void)hipGetLastError();
          return e->fail(VC_ERR_UNSUPPORTED, "the likelihood kernel needs %u bytes of LDS per workgroup (2 n_harmonics + 1 + Nb = %d coefficient rows per gene"
                                             "%s), this device grants %zu", dyn_bytes, d.K, d.generic ? " on the run-time-sized kernel set" : "", lds_cap);
        }
      }
      HIPCHK(e, hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, main_kernel, 64 * wg_waves, dyn_bytes));
      if (d.generic && bpc > 8) bpc = 8;      // (one wave each: two per SIMD hide the LDS latency, more only shorten the runs of cells)
      if (e->tun.blocks_per_cu > 0) bpc = e->tun.blocks_per_cu;
      *out = bpc;
      return VC_OK;
    };
    // Unequal shares per dispatch pass.  The workgroups of pass p (the p-th one on every CU) are older than those of pass
    // p + 1 and the SIMD arbiter issues the oldest ready wave first: with equal shares the first pass ends its cells at
    // ~60 % of the kernel and the last pass then runs alone, one wave per SIMD (profiles/tools/wave_timeline.py).  Measured
    // (profiles/r02_pass_shares.md): shares falling by 1/2 per pass (2 passes 0.67 : 0.33, 3 passes 0.57 : 0.29 : 0.14) cut the
    // kernel by 5-9 %; that is the default for a full multi-pass grid.  tuning.pass_shares overrides (n_pass_shares = 1: the
    // balanced tiling).  The tiling (vc_host_logic.h, also what the kernel evaluates per wave) stays a pure function of
    // (Nc, Ng, occupancy, CUs): results are reproducible.
    auto tile = [&](int blocks_per_cu) {
      if (blocks_per_cu < 1) blocks_per_cu = 1;
      tile_host.clear(); bat_chunk_host.clear();
      double share[4] = {1.0, 0.5, 0.25, 0.125};
      if (d.kind == VC_KIND_VFULL && blocks_per_cu == 2) {
        // round 3: with two cells of the S+U kernel's counts in flight (hand-placed waits) the older wave of a SIMD stalls less
        // and leaves the younger one fewer issue slots: 0.67 : 0.33 ends the passes at 104 / 120 us, 0.75 : 0.25 together
        // (profiles/r03_kmain.md: 119.4 vs 121.4 us)
        share[1] = 1.0 / 3.0; share[2] = 1.0 / 9.0; share[3] = 1.0 / 27.0;
      } else if (d.kind != VC_KIND_VFULL && blocks_per_cu == 3) {
        // the three-pass one-matrix kernels: 0.62 : 0.26 : 0.12 instead of 0.57 : 0.29 : 0.14 (the third pass ended 12 us after
        // the first): phase 58.3 -> 56.2-56.9 us, U only 70.6 -> 70.0-70.1 us (profiles/r03_kmain.md section 3)
        share[1] = 0.42; share[2] = 0.19; share[3] = 0.09;
      }
      bool want = true;
      if (e->tun.n_pass_shares == 1) want = false;                  // equal shares: the balanced tiling
      else if (e->tun.n_pass_shares >= 2) {
        const int n = e->tun.n_pass_shares;
        for (int p = 0; p < n; ++p) share[p] = (double)e->tun.pass_shares[p];
        for (int p = n; p < 4; ++p) share[p] = share[p - 1] * (share[n - 1] / share[n - 2]);
      }
      // below 12 cells per wave the passes' fixed prologue / epilogue dominate (measured at the 6 250-cell shard)
      const VcTiling t = vc_tile_cells(d.Nc, d.nGB, n_cu, blocks_per_cu, wg_waves, e->tun.cells_per_wave,
                                       (want && !d.generic) ? share : nullptr, e->tun.pass_min_cw > 0 ? e->tun.pass_min_cw : 12);
      d.cw = t.cw;
      d.n_chunks = t.n_chunks;
      d.pass_wgs = n_cu;
      for (int p = 0; p < 4; ++p) d.pass_cw[p] = t.pass_cw[p];
      if (d.onehot) {
        // every workgroup inside one batch: the chunks of a gene block are dealt out to the batches in proportion to their cells
        // (vc_host_logic.h: vc_tile_batches); a non-empty batch needs at least one chunk
        int nonempty = 0;
        for (int v : bat_len) nonempty += v > 0;
        if (d.n_chunks < nonempty) d.n_chunks = nonempty;
        const int cwm = vc_tile_batches(d.Nc, d.nGB, d.n_chunks, d.pass_wgs, d.pass_cw, wg_waves, bat_len, tile_host, bat_chunk_host);
        if (cwm > d.cw) d.cw = cwm;
      }
      d.n_main_wg = d.nGB * d.n_chunks;
    };
    int bpc0 = 0;
    TRY(occupancy(0, &bpc0));
    tile(bpc0);
    // K_main's own partials of d loglik / d nu_omega (vc_common.h: VC_PW_INLINE; one rank, <= VC_PWQ coefficients): the W rows of a
    // wave's cells are staged in DYNAMIC shared memory, which may cost the kernel a resident workgroup per CU -- the tiling
    // follows the occupancy the launch will really have (at most one adjustment; if even that does not fit: off)
    d.pw_inline = 0;
    d.pw_slots = 0;
    const bool pw_kind = VC_PW_INLINE && (d.kind == VC_KIND_VU || d.kind == VC_KIND_VFULL) && d.NW >= 1 && d.NW <= VC_PWQ &&
                         e->cfg.world_size == 1 && !d.generic;
    d.pw_lane = 0;
    const bool lane_conds = e->D_all_ones || (d.onehot && !e->bat_cond.empty());     // one condition per workgroup of K_main
    if (pw_kind && e->tun.pw_inline != 1 && d.kind == VC_KIND_VU && lane_conds && d.Hw <= d.H && !e->tun.no_pw_lane) {
      // Round 6: ONE condition with D == 1 (every one-sample fit): the W row of a cell is (1, sin k phi_c, cos k phi_c) -- what its
      // record already holds as wave-uniform SGPR pairs.  The U-only kernel's PWL instantiation (count storage | 4) accumulates
      // A3 x W per LANE (1 + 2 Hw plain VALU per cell) instead of reducing A3 over the wave first (a 64-lane DPP tree + staging
      // per cell): no W rows in the LDS, no per-cell rows stored (nothing reads them once K_main delivers the partials itself).
      const void* kl = nullptr;
      const char* nm = nullptr;
      hipFuncAttributes fa;
      vc_main_launch_fn fl = vc_find_main_kernel(d.H, knb, d.kind, d.noise, d.gpl, d.c16 | 4, &nm, &kl);
      if (fl && kl && hipFuncGetAttributes(&fa, kl) == hipSuccess && fa.localSizeBytes == 0) {
        e->main_fn_rows = e->main_fn;       // the instantiation that stores per-cell rows: what the sharded phases launch (vc_svi_run_sharded)
        e->main_fn = fl;
        main_kernel = kl;
        d.pw_inline = d.NW <= 4 ? 4 : 8;
        d.pw_slots = 0;
        d.pw_lane = 1;
        // (without the staging registers of the wave-level reduction this instantiation needs 120 VGPRs instead of 160: four
        // workgroups per CU instead of three -- the tiling follows the occupancy of the kernel that will run)
        int bl = 0;
        TRY(occupancy(0, &bl));
        if (bl != bpc0) tile(bl);
      }
    }
    if (d.pw_lane) {
    } else
    if (pw_kind && e->tun.pw_inline != 1) {
      const int row = d.NW <= 4 ? 4 : 8;
      int bpc = bpc0;
      for (int attempt = 0; attempt < 2; ++attempt) {
        const int cwmax = std::max(d.cw, std::max(std::max(d.pass_cw[0], d.pass_cw[1]), std::max(d.pass_cw[2], d.pass_cw[3])));
        VcDims probe = d;
        probe.pw_inline = row;
        probe.pw_slots = (cwmax * (row / 4) + 15) / 16 * 16;
        const unsigned dyn = vc_main_dyn_lds(probe);
        int got = 0;
        if (dyn > 96u * 1024u || occupancy(dyn, &got) != VC_OK || got < 1) break;
        if (got == bpc) { d.pw_inline = row; d.pw_slots = probe.pw_slots; break; }
        // The W rows would cost a resident workgroup per CU (the 4-genes-per-lane S+U kernel: its reduction tiles already take
        // 52 KB of LDS, three workgroups just fit).  Measured at 2 000 genes (profiles/r04_small_shard.md): K_main 58.9 -> 70.8 us
        // at 25 000 cells, 33.6 -> 40.1 at 12 500, 23.0 -> 24.6 at 6 250 -- more than the launch it saves, except where a wave
        // has so few cells that the kernel is prologue and epilogue anyway: accepted only there.
        if (d.cw > 12 && e->tun.pw_inline != 2) break;      // (tuning.pw_inline = 2: accept the loss anyway -- tests, A/B)
        bpc = got;                      // tile for the occupancy that is left and look again
        tile(bpc);
      }
      if (!d.pw_inline) tile(bpc0);     // off: the tiling of the plain kernel
    }
  }
  if (!d.generic && d.noise == VC_NOISE_NB) {
    // the gradient-only twin of the selected kernel (same tiling, same dynamic LDS); used only after vc_set_loss_every(k > 1)
    const void* knl = nullptr;
    e->main_fn_nl = vc_find_main_kernel(d.H, knb, d.kind, d.noise, d.gpl, d.c16 | 2 | (d.pw_lane ? 4 : 0), &e->main_name_nl, &knl);
    hipFuncAttributes fa;
    if (e->main_fn_nl && knl && (hipFuncGetAttributes(&fa, knl) != hipSuccess || fa.localSizeBytes > 0)) e->main_fn_nl = nullptr;
    const unsigned dyn = vc_main_dyn_lds(d);
    if (e->main_fn_nl && dyn > 0 && hipFuncSetAttribute(knl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
      (void)hipGetLastError();
      e->main_fn_nl = nullptr;          // (no gradient-only twin then: vc_set_loss_every says so)
    }
  }
  {
    // the tiling as a table: {first cell of wave 0, cells per wave} of every workgroup of the likelihood kernel, evaluated
    // HERE by the function the kernel used to run itself (vc_host_logic.h; 64-bit divisions = ~700 scalar instructions in
    // front of every wave's first load)
    // {first cell, cells per wave, batch, end of the workgroup's cells}; one-hot batches: the batch-aligned table built with the tiling
    if (!d.onehot) {
      tile_host.assign(4 * (size_t)d.n_main_wg, 0);
      for (int w = 0; w < d.n_main_wg; ++w) {
        int cw = 0;
        const long long first = vc_wave_first_cell(w / d.nGB, w % d.nGB, 0, d.nGB, d.pass_wgs, d.pass_cw, d.generic ? 1 : VC_WAVES, &cw);
        tile_host[4 * (size_t)w] = (int)std::min<long long>(first, (long long)d.Nc);
        tile_host[4 * (size_t)w + 1] = cw;
        tile_host[4 * (size_t)w + 3] = d.Nc;
      }
      b.bat_chunk = nullptr;
    } else {
      if (tile_host.size() != 4 * (size_t)d.n_main_wg) return e->fail(VC_ERR_STATE, "internal: batch-aligned tile table of the wrong size");
      TRY(upload(e, bat_chunk_host, &b.bat_chunk));
      if (d.pw_lane && !e->bat_cond.empty())       // {.., batch | condition << 16, ..}: the PWL kernel places its partials in its condition's columns
        for (int w = 0; w < d.n_main_wg; ++w) {
          const int q = tile_host[4 * (size_t)w + 2];
          tile_host[4 * (size_t)w + 2] = q | (e->bat_cond[(size_t)q] << 16);
        }
    }
    TRY(upload(e, tile_host, &b.wg_tile));
  }
  if (dev_ord) { e->dfree(dev_ord); dev_ord = nullptr; }
  d.nb_pre_gene = d.Ng_pad / 64;       // 64 genes per block, the sites of a gene spread over its 4 waves
  d.nb_pre_cell = (d.Nc + 255) / 256;
  d.nb_post_gene = d.Ng_pad / 64;
  d.nb_post_cell = (d.Nc + 1023) / 1024;
  if (d.generic) {                     // thread = gene / thread = cell, 256 per block (vc_generic_kernels.hip)
    d.nb_pre_gene = d.nb_post_gene = (d.Ng_pad + 255) / 256;
    d.nb_post_cell = (d.Nc + 255) / 256;
  }
  d.hist_has_S = nb && d.kind != VC_KIND_VU;
  d.hist_has_U = nb && vel;
  d.nmat_r = nb ? (d.kind == VC_KIND_VFULL ? 2 : 1) : 0;
  e->hist_each_step = nb && !cond(e, VC_SITE_SHAPE_INV);
  d.hist_par = e->hist_each_step ? 1 : 0;

  // uploads
  TRY(upload(e, e->h_prior[VC_PRIOR_MU_NU], &b.mu_nu));
  TRY(upload(e, e->h_prior[VC_PRIOR_SD_NU], &b.sd_nu));
  if (vel) {
    TRY(upload(e, e->h_prior[VC_PRIOR_MU_GAMMA], &b.mu_g));
    TRY(upload(e, e->h_prior[VC_PRIOR_SD_GAMMA], &b.sd_g));
    TRY(upload(e, e->h_prior[VC_PRIOR_MU_BETA], &b.mu_b));
    TRY(upload(e, e->h_prior[VC_PRIOR_SD_BETA], &b.sd_b));
    TRY(upload(e, e->h_prior[VC_PRIOR_MU_NUOMEGA], &b.mu_w));
    TRY(upload(e, e->h_prior[VC_PRIOR_SD_NUOMEGA], &b.sd_w));
  }
  if (!vel && d.with_dnu) TRY(upload(e, e->h_prior[VC_PRIOR_SD_DNU], &b.sd_dnu));
  for (int s = 0; s < VC_SITE_COUNT; ++s) {
    b.cnd[s] = nullptr;
    if (cond(e, s)) TRY(upload(e, e->h_cond[s], &b.cnd[s]));
    b.lat[s] = nullptr;
    if (site_exists(e, s)) TRY(e->dalloc(&b.lat[s], (size_t)site_size(e, s)));
  }
  TRY(e->dalloc(&b.eps_used, (size_t)e->layout.eps_total));
  HIPCHK(e, hipMemset(b.eps_used, 0, sizeof(float) * e->layout.eps_total));
  TRY(e->dalloc(&b.GT, (size_t)(d.K + 3) * d.Ng_pad));
  d.ctw = 2 * ((vc_rec_pairs(d.H, d.nbk, d.kind == VC_KIND_VFULL) + VC_REC_PAD - 1) / VC_REC_PAD * VC_REC_PAD);   // values duplicated {x,x}, record padded to VC_REC_PAD pairs
  TRY(e->dalloc(&b.CT, (size_t)d.Nc * d.ctw));
  HIPCHK(e, hipMemset(b.CT, 0, sizeof(float) * (size_t)d.Nc * d.ctw));
  TRY(e->dalloc(&b.lat_delta, (size_t)d.M));
  TRY(e->dalloc(&b.lat_sgam, (size_t)d.Ng));
  TRY(e->dalloc(&b.lat_phi, (size_t)d.Nc));
  TRY(e->dalloc(&b.lat_omega, (size_t)d.Nc));
  TRY(e->dalloc(&b.lat_domega, (size_t)d.Nc));
  const int nq_max = std::max(d.nq, d.kind == VC_KIND_VU ? nq_of(VC_KIND_PHASE) : 0);
  TRY(e->dalloc(&b.GO, (size_t)d.n_chunks * nq_max * d.Ng_pad));
  TRY(e->dalloc(&b.CO, (size_t)d.nGB * 3 * d.Nc));
  HIPCHK(e, hipMemset(b.CO, 0, sizeof(float) * (size_t)d.nGB * 3 * d.Nc));   // (pw_lane: K_main stores no per-cell rows; their unfused readers see zeros)
  TRY(e->dalloc(&b.LO, (size_t)d.n_main_wg));
#ifdef VC_DBG_TIMES
  TRY(e->dalloc(&b.dbg, VC_DBG_WORDS(d.n_main_wg)));   // + per-block stamps of K_pre / K_post / K_fin, per-wave of K_tail / K_omega
  HIPCHK(e, hipMemset(b.dbg, 0, VC_DBG_WORDS(d.n_main_wg) * 8));
#endif
  TRY(e->dalloc(&b.LP, (size_t)(d.nb_pre_gene + d.nb_pre_cell + d.nb_post_gene)));
  // (pw_inline -- the angular-speed gradient partials out of K_main -- was decided with the tiling above)
  TRY(e->dalloc(&b.PWM, (size_t)d.n_main_wg * VC_PWQ));
  HIPCHK(e, hipMemset(b.PWM, 0, sizeof(float) * (size_t)d.n_main_wg * VC_PWQ));
  TRY(e->dalloc(&b.WT, (size_t)VC_PWQ * d.Nc));
  HIPCHK(e, hipMemset(b.WT, 0, sizeof(float) * (size_t)VC_PWQ * d.Nc));
  TRY(e->dalloc(&b.PW, (size_t)((d.Nc + 255) / 256) * std::max(1, d.NW)));      // K_post: 1024-cell blocks; K_tail: 256
  // K_tail's cell blocks: 256 cells on 4 of the block's 16 waves keep the per-cell chain shortest, but a block still takes a
  // 1024-thread slot when it is placed (2 per CU): beyond ~2 rounds of them the placement is what costs (400 000 cells:
  // K_tail 68 us), and 1024 cells per block -- every wave busy, a quarter of the blocks -- is faster (measured threshold)
  d.tail_tc = d.Nc > 160000 ? 1024 : 256;
  // the one-launch tail (vc_launch_tail2) puts the loss / histogram / eps blocks into the same launch, and every 1024-thread block
  // of it holds a CU on its own: with 256-cell blocks the cell blocks alone take most of the chip's 256 slots and the others
  // wait a round (50k x 2k: phase 17.5 -> 13.0 us outside K_main with 1024-cell blocks, V-joint 25 -> 22.5)
  // -- and with the dense histogram tables (S+U kernel) 512: its cell blocks carry the longest chain of that launch (nu_omega inside)
  if (fused_tail_kind(e) == 2 && d.Nc > 16384) d.tail_tc = d.kind == VC_KIND_VFULL ? 512 : 1024;
  if (e->tun.tail_cells) d.tail_tc = e->tun.tail_cells;
  if (e->tun.p2p_one_launch && e->cfg.world_size >= 1 && d.Nc <= 160000) d.tail_tc = 256;      // vc_tail_x_kernel: cell block = K_omega's block
  d.nb_tail_cell = (d.Nc + d.tail_tc - 1) / d.tail_tc;
  d.nlpf = d.nb_post_gene + d.nb_tail_cell + 1;
  d.lgamma_alpha = lgammaf(d.gamma_alpha);
  d.spec = VC_SPEC_NONE;      // (chosen at the end: the histogram form below is part of the signature)
  {
    // exchange buffer of the sharded fused step (include/velocycle_hip.h): gradient region, PW rows, loss pairs
    const long long world = e->cfg.world_size, ncg = e->cfg.Nc_global > 0 ? e->cfg.Nc_global : d.Nc;
    const long long max_shard = std::max<long long>((ncg + world - 1) / world, d.Nc);
    e->xb_pw_cap = (int)((max_shard + 255) / 256);
    long long off = e->layout.header + e->layout.n_global;
    off = (off + 3) / 4 * 4;
    e->xb_pw_off = (int)off;
    off += (long long)e->xb_pw_cap * d.NW;
    off = (off + 3) / 4 * 4;
    e->xb_loss_off = (int)off;
    off += 4LL * (1 + d.nb_post_gene);      // VC_LOSS_PIECES floats per loss term (vc_fused_kernels.hip)
    e->xb_total = (off + 3) / 4 * 4;
    TRY(e->dalloc(&e->sis, 3 * (size_t)d.Ng_pad));
    HIPCHK(e, hipMemset(e->sis, 0, 3 * sizeof(float) * d.Ng_pad));
  }
  TRY(e->dalloc(&b.LPF, 2 * (size_t)d.nlpf));
  HIPCHK(e, hipMemset(b.LPF, 0, 2 * sizeof(double) * d.nlpf));
  TRY(e->dalloc(&b.LPP, (size_t)d.nb_post_gene));
  HIPCHK(e, hipMemset(b.LPP, 0, sizeof(double) * d.nb_post_gene));
  TRY(e->dalloc(&b.LPR, 2 * (size_t)d.nb_post_gene));
  HIPCHK(e, hipMemset(b.LPR, 0, 2 * sizeof(double) * d.nb_post_gene));
  TRY(e->dalloc(&b.SIS, 2 * 4 * (size_t)d.Ng_pad));
  HIPCHK(e, hipMemset(b.SIS, 0, 2 * 4 * sizeof(float) * d.Ng_pad));
  TRY(e->dalloc(&b.NWS, 2 * 4 * (size_t)VC_MAX_NW * (VC_MAX_RANK + 2)));      // two copies, by the parity of the step
  HIPCHK(e, hipMemset(b.NWS, 0, 2 * 4 * sizeof(float) * VC_MAX_NW * (VC_MAX_RANK + 2)));
  TRY(e->dalloc(&b.EPS, 3 * (size_t)e->layout.eps_total));
  HIPCHK(e, hipMemset(b.EPS, 0, 3 * sizeof(float) * e->layout.eps_total));
  b.step_ctr = nullptr;
  TRY(e->dalloc(&b.step_size, 2));
  HIPCHK(e, hipMemset(b.step_size, 0, 2 * sizeof(float)));
  TRY(e->dalloc(&b.status, 4));       // [0] steps with a non-finite loss, [1] 1 + the first of them, [2] 1 + step of an exchange time-out
  HIPCHK(e, hipMemset(b.status, 0, 4 * sizeof(long long)));

  // histograms + step-invariant constants -----------------------------------------------------
  double lg_S = 0.0, lg_U = 0.0;   // sum lgamma(k+1)
  {
    std::vector<int> ptr;
    std::vector<float> val, cnt;
    bool dense_ok = false;
    std::vector<float> hc;
    std::vector<int> hc_off, hc_rows;
    if (want_hist) {
      bool bad_host = false, fallback = false;
      unsigned novf[2] = {0, 0};
      if (dev_hist) {
        HIPCHK(e, hipMemcpy(novf, ovf_n, sizeof novf, hipMemcpyDeviceToHost));
        fallback = novf[0] > OVF_CAP || novf[1] > OVF_CAP;     // mostly non-integer data: the overflow list does not hold it
      }
      if (dev_hist && !fallback) {
        std::vector<unsigned> htab((size_t)d.Ng * VC_HIST_CAP);
        // dense tail-count tables (vc_host_logic.h: vc_build_dense_hist) when every non-zero count of the matrices is an integer
        // below VC_HIST_CAP (no overflow entries): the histogram sums then cost a logarithm and a reciprocal per (gene, count
        // level) and are evaluated per gene block.
        // Default: the S+U kernel's models only -- 4 000+ task waves there (two matrices) against 32 blocks; measured at 50k x 2k
        // (profiles/r04_two_launch.md) V-joint 21-24 -> 17.5-19 us outside K_main, phase (one matrix, nothing to hide the blocks
        // of its few outlier genes behind) 13.4 -> 15.5: the phase model keeps the lists.  tuning.hist_dense = 2 / 1 forces either.
        // Only where the one-launch tail runs (single rank with K_main's own nu_omega partials): through K_omega's / K_pre's
        // 4-wave blocks the dense evaluation is slower than the lists (6 250-cell shard: sharded step 41.6 -> 44.2 us).
        dense_ok = nb && novf[0] == 0 && novf[1] == 0 && d.Nc <= (1 << 24) &&
                   (e->tun.hist_dense ? e->tun.hist_dense == 2 : (d.kind == VC_KIND_VFULL && fused_tail_kind(e) == 2));
        for (int m = 0; m < (vel ? 2 : 1); ++m) {
          HIPCHK(e, hipMemcpy(htab.data(), tab[m], htab.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
          if (dense_ok) vc_build_dense_hist(htab.data(), d.Ng, d.Ng_pad, hc, hc_off, hc_rows);
          std::vector<float> ov(novf[m]);
          std::vector<int> og(novf[m]);
          if (novf[m]) {
            HIPCHK(e, hipMemcpy(ov.data(), ovf_val[m], novf[m] * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemcpy(og.data(), ovf_gene[m], novf[m] * sizeof(int), hipMemcpyDeviceToHost));
          }
          std::vector<std::pair<int, float>> ovf(novf[m]);
          for (unsigned i = 0; i < novf[m]; ++i) ovf[i] = {og[i], ov[i]};
          const double lg = vc_compact_hist(htab.data(), d.Ng, ovf, ptr, val, cnt);
          (m == 0 ? lg_S : lg_U) = lg;
        }
        if (!vel) for (int g = 0; g < d.Ng; ++g) ptr.push_back((int)val.size());
        e->hist_on_device = 1;
      } else {
        // checker path (tuning.host_hist) or fallback: histograms from a host copy of the raw counts
        if (e->hS.empty()) {
          if (e->src[0].kind != 1) { free_transients(); return e->fail(VC_ERR_UNSUPPORTED, "CSR input with more than %u non-integer / large counts per matrix", OVF_CAP); }
          const size_t span = (size_t)(d.Ng - 1) * e->dgs + (size_t)(d.Nc - 1) * e->dcs + 1;
          e->hS.resize(span);
          HIPCHK(e, hipMemcpy(e->hS.data(), e->src[0].dense, span * sizeof(float), hipMemcpyDeviceToHost));
          if (vel) { e->hU.resize(span); HIPCHK(e, hipMemcpy(e->hU.data(), e->src[1].dense, span * sizeof(float), hipMemcpyDeviceToHost)); }
          e->gs = e->dgs; e->cs = e->dcs;
        }
        lg_S = vc_build_hist_host(e->hS.data(), e->gs, e->cs, d.Ng, d.Nc, ptr, val, cnt, &bad_host);
        if (vel) lg_U = vc_build_hist_host(e->hU.data(), e->gs, e->cs, d.Ng, d.Nc, ptr, val, cnt, &bad_host);
        else for (int g = 0; g < d.Ng; ++g) ptr.push_back((int)val.size());
        e->hist_on_device = 0;
        if (bad_host) { free_transients(); return e->fail(VC_ERR_ARG, "count matrices must be finite and >= 0 (NaN / Inf / negative value found)"); }
      }
    } else {
      ptr.assign(2 * (size_t)d.Ng, 0);
    }
    ptr.push_back((int)val.size());
    free_transients();
    for (auto& c : e->src) c.release();
    e->setup_transient_bytes = transient;
    // tasks: runs of <= 64 histogram entries of one gene and matrix, sorted by gene
    std::vector<int> task, tptr;
    d.hist_dense = dense_ok ? 1 : 0;
    if (dense_ok) {
      // one task per gene and matrix, in that order: what role 12 / the loss block / K_post add up per gene is unchanged
      const int nm = vel ? 2 : 1, nblk = d.Ng_pad / 64;
      for (int g = 0; g < d.Ng; ++g) {
        tptr.push_back(nm * g);
        for (int m = 0; m < nm; ++m) task.insert(task.end(), {g, m, 0, 0});
      }
      tptr.push_back(nm * d.Ng);
      if (!vel) {                               // the U half of the tables: empty
        hc_off.resize(2 * (size_t)nblk, (int)(hc.size() / 64));
        hc_rows.resize(2 * (size_t)nblk, 0);
      }
      hc.resize(hc.size() + 64, 0.f);          // (a row of padding: the prefetch of an empty block reads row 0)
      // gene blocks that hold a highly expressed gene: the one-launch tail gives each of them four quarter blocks (16 genes, one wave per
      // SIMD of a CU of their own: vc_common.h, vc_hist_dense16_finish<true>), every quarter to ITS genes' largest count
      std::vector<int> hc_rows_q(hc_rows.size() * 4, 0), hc_split;
      for (size_t i = 0; i < hc_rows.size(); ++i) {
        for (int qd = 0; qd < 4; ++qd) {
          int rq = 0;
          for (int j = hc_rows[i] - 1; j >= 0 && !rq; --j)
            for (int l = 16 * qd; l < 16 * qd + 16; ++l)
              if (hc[((size_t)hc_off[i] + j) * 64 + l] != 0.f) { rq = j + 1; break; }
          hc_rows_q[i * 4 + qd] = rq;
        }
        if (hc_rows[i] > VC_HIST_SPLIT_ROWS) hc_split.push_back((int)i);
      }
      b.n_hc_split = (int)hc_split.size();
      if (hc_split.empty()) hc_split.push_back(0);
      TRY(upload(e, hc, &b.HC));
      TRY(upload(e, hc_off, &b.hc_off));
      TRY(upload(e, hc_rows, &b.hc_rows));
      TRY(upload(e, hc_rows_q, &b.hc_rows_q));
      TRY(upload(e, hc_split, &b.hc_split));
    } else {
      vc_build_hist_tasks(ptr, d.Ng, task, tptr);
      b.HC = nullptr; b.hc_off = nullptr; b.hc_rows = nullptr; b.hc_rows_q = nullptr; b.hc_split = nullptr; b.n_hc_split = 0;
    }
    b.n_tasks = (int)task.size() / 4;
    TRY(upload(e, task, &b.h_task));
    TRY(upload(e, tptr, &b.h_tptr));
    TRY(e->dalloc(&b.HL, 2 * (size_t)std::max(1, b.n_tasks)));       // two halves: by the parity of the sample's step (d.hist_par)
    TRY(e->dalloc(&b.HD, 2 * (size_t)std::max(1, b.n_tasks)));
    HIPCHK(e, hipMemset(b.HL, 0, 2 * sizeof(double) * std::max(1, b.n_tasks)));
    HIPCHK(e, hipMemset(b.HD, 0, 2 * sizeof(double) * std::max(1, b.n_tasks)));
    TRY(upload(e, ptr, &b.h_ptr));
    TRY(upload(e, val, &b.h_val));
    TRY(upload(e, cnt, &b.h_cnt));
    e->h_ptr_host = ptr; e->h_val_host = val; e->h_cnt_host = cnt;
    {
      // sum_c k_U per gene (this rank's cells): the S+U kernel adds -log beta * sum_c k_U once per gene instead of subtracting
      // log beta from eta_U once per (gene, cell) (VC_HOIST_LB)
      std::vector<float> gsu((size_t)d.Ng_pad, 0.f);
      if (vel && want_hist)
        for (int g = 0; g < d.Ng; ++g) {
          double sum = 0.0;
          for (int j = ptr[(size_t)d.Ng + g]; j < ptr[(size_t)d.Ng + g + 1]; ++j) sum += (double)val[j] * (double)cnt[j];
          gsu[g] = (float)sum;
        }
      TRY(upload(e, gsu, &b.gene_sum_u));
    }
    std::vector<float>().swap(e->hS);
    std::vector<float>().swap(e->hU);
  }
  const double ln_const_s = (double)d.Ng * d.Nc * (std::log((double)d.sigma_ln_s) + 0.5 * std::log(2.0 * M_PI));
  const double ln_const_u = (double)d.Ng * d.Nc * (std::log((double)d.sigma_ln_u) + 0.5 * std::log(2.0 * M_PI));
  auto obs_const = [&](bool is_u) {   // constant part of -loglik of one matrix
    if (d.noise == VC_NOISE_LOGNORMAL) return is_u ? ln_const_u : ln_const_s;
    return is_u ? lg_U : lg_S;
  };
  double cl = 0.0;
  if (d.kind != VC_KIND_VU) cl += obs_const(false);
  if (vel) cl += obs_const(true);

  if (d.kind == VC_KIND_VU) {
    // hoist the S likelihood: with phi, nu, dnu, shape_inv fixed it does not change between steps
    VcDims d2 = d;
    d2.kind = VC_KIND_PHASE; d2.nq = nq_of(VC_KIND_PHASE); d2.nco = 1;
    d2.pw_inline = 0; d2.pw_slots = 0;
    d2.hist_has_S = nb; d2.hist_has_U = 0;
    vc_launch_pre(d2, b, nullptr, nullptr, 0, 0, nullptr, 1, nb ? 1 : 0, st);
    e->phase_fn(d2, b, st);
    HIPCHK(e, hipStreamSynchronize(st));
    HIPCHK(e, hipGetLastError());
    std::vector<float> lo(d.n_main_wg), rrow(d.Ng);
    std::vector<double> hl(std::max(1, b.n_tasks), 0.0);
    HIPCHK(e, hipMemcpy(lo.data(), b.LO, sizeof(float) * d.n_main_wg, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(rrow.data(), b.GT + (size_t)(d.K + 2) * d.Ng_pad, sizeof(float) * d.Ng, hipMemcpyDeviceToHost));
    if (nb) HIPCHK(e, hipMemcpy(hl.data(), b.HL, sizeof(double) * b.n_tasks, hipMemcpyDeviceToHost));
    double sconst = obs_const(false);
    for (float v : lo) sconst -= (double)v;
    if (nb) {
      for (int g = 0; g < d.Ng; ++g) sconst -= (double)d.Nc * rrow[g] * std::log((double)rrow[g]);
      for (int t = 0; t < b.n_tasks; ++t) sconst -= hl[t];          // only S tasks are non-zero in this pass
    }
    cl += sconst;
  }
  if (nb && !e->hist_each_step) {
    // shape_inv conditioned: the lgamma / digamma sums never change -> evaluate them once
    vc_launch_hist(d, b, nullptr, 1, st);
    HIPCHK(e, hipStreamSynchronize(st));
  }
  b.const_loss = cl;
  HIPCHK(e, hipStreamSynchronize(st));
  HIPCHK(e, hipGetLastError());
  // the instantiation of the small kernels compiled for this configuration, if there is one (vc_tail_spec.h)
  d.spec = e->tun.no_tail_spec ? VC_SPEC_NONE : vc_spec_match(d);
  e->finalized = true;
  std::vector<float>().swap(e->hD);
  std::vector<float>().swap(e->hDb);   // kept until here: every failure above leaves the batch design matrix in place
  return VC_OK;
}

static int launch_front(vc_engine* e, const float* params, const float* eps, uint64_t seed, int64_t step,
                        int64_t* step_dev, float* grad, hipStream_t st) {
  vc_launch_pre(e->d, e->b, params, eps, seed, (long long)step, (const long long*)step_dev, 0,
                e->hist_each_step ? 1 : 0, st);
  if (e->timing) {
    if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
    auto& pr = e->ev_pool[e->ev_used++];
    HIPCHK(e, hipEventRecord(pr.first, st));
    e->main_fn(e->d, e->b, st);
    HIPCHK(e, hipEventRecord(pr.second, st));
  } else {
    e->main_fn(e->d, e->b, st);
  }
  vc_launch_post(e->d, e->b, params, grad, (long long*)step_dev, st);
  return VC_OK;
}

extern "C" int vc_elbo_grad(vc_engine* e, const float* params, const float* eps, uint64_t seed, int64_t step,
                            int64_t* step_dev, float* grad, double* loss_dev, int64_t loss_slots,
                            void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_elbo_grad before vc_finalize");
  if (!params || !grad) return e->fail(VC_ERR_ARG, "vc_elbo_grad: null params / grad");
  hipStream_t st = (hipStream_t)hip_stream;
  TRY(launch_front(e, params, eps, seed, step, step_dev, grad, st));
  vc_launch_fin(e->d, e->b, params, grad, loss_dev, (long long)loss_slots, (long long)step, (const long long*)step_dev, st);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

#define VC_NO_GENERIC(e_, what)                                                                                              \
  do {                                                                                                                      \
    if ((e_)->d.generic)                                                                                                    \
      return (e_)->fail(VC_ERR_UNSUPPORTED, what ": the run-time-sized kernel set (a configuration outside the compiled "   \
                                                  "fast set) has the unfused step only: vc_elbo_grad + vc_clipped_adam");  \
  } while (0)

extern "C" int vc_svi_step(vc_engine* e, float* params, const float* eps, uint64_t seed, int64_t step,
                           int64_t* step_dev, float* grad, double* loss_dev, int64_t loss_slots, float* exp_avg,
                           float* exp_avg_sq, double lr, double lrd, double beta1, double beta2, double adam_eps,
                           double clip_norm, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_svi_step before vc_finalize");
  VC_NO_GENERIC(e, "vc_svi_step");
  if (e->cfg.world_size != 1)
    return e->fail(VC_ERR_STATE, "vc_svi_step merges the optimiser with the last gradient kernel: single rank only "
                                 "(use vc_elbo_grad + all-reduce + vc_clipped_adam when cells are sharded)");
  if (!params || !grad || !exp_avg || !exp_avg_sq) return e->fail(VC_ERR_ARG, "vc_svi_step: null buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  TRY(launch_front(e, params, eps, seed, step, step_dev, grad, st));
  vc_launch_fin_adam(e->d, e->b, params, grad, loss_dev, (long long)loss_slots, (long long)step, (long long*)step_dev,
                     exp_avg, exp_avg_sq, e->hyper(lr, lrd, beta1, beta2, adam_eps, clip_norm),
                     (int)e->layout.header, (long long)e->layout.total, st);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

// Which launch follows K_main in the steady state of vc_svi_run_fused: 1 = the tutorial flow's merged tail (vc_launch_tail_merged),
// 2 = the one-launch tail of every other single-rank step that has what it needs (vc_launch_tail2), 0 = K_tail + K_omega
static int fused_tail_kind(const vc_engine* e) {
  if (e->d.generic) return 0;
  const bool with_hist = e->hist_each_step;
  bool merged = e->d.pw_inline && e->d.kind == VC_KIND_VU && !with_hist && (e->d.cond >> VC_SITE_PHIXY & 1u);
  if (e->tun.no_tail_merged) merged = false;
  if (merged) return 1;
  // Round 4: every other single-rank step in TWO launches as well -- the phase model (no nu_omega chain), and the velocity
  // models whenever K_main supplies the partials of d loglik / d nu_omega itself (pw_inline), so that the chain runs inside the
  // cell blocks.  tuning.no_tail2 keeps the three-launch step (A/B, tests: the same bits).
  bool tail2 = e->cfg.world_size == 1 && (e->d.model == VC_MODEL_PHASE || e->d.pw_inline != 0);
  if (e->tun.no_tail2) tail2 = false;
  return tail2 ? 2 : 0;
}
static int fused_launches_per_step(const vc_engine* e) { return fused_tail_kind(e) ? 2 : 3; }

extern "C" int vc_svi_run_fused(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad,
                                double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr,
                                double lrd, double beta1, double beta2, double adam_eps, double clip_norm, int prime,
                                int64_t n_steps, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_svi_run_fused before vc_finalize");
  VC_NO_GENERIC(e, "vc_svi_run_fused");
  if (e->cfg.world_size != 1)
    return e->fail(VC_ERR_STATE, "vc_svi_run_fused applies the optimiser inside the gradient kernels: single rank only "
                                 "(use vc_elbo_grad + all-reduce + vc_clipped_adam when cells are sharded)");
  if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev)
    return e->fail(VC_ERR_ARG, "vc_svi_run_fused: null buffer (the device step counter is required)");
  if (n_steps < 0) return e->fail(VC_ERR_ARG, "vc_svi_run_fused: negative n_steps");
  if (!(lr > 0.0) || !(lrd > 0.0) || !(beta1 > 0.0 && beta1 < 1.0) || !(beta2 > 0.0 && beta2 < 1.0))
    return e->fail(VC_ERR_ARG, "vc_svi_run_fused: lr, lrd must be positive and the betas inside (0, 1)");
  hipStream_t st = (hipStream_t)hip_stream;
  VcAdamArgs a;
  e->fill(a, exp_avg, exp_avg_sq, lr, lrd, beta1, beta2, adam_eps, clip_norm);
  const long long* sd = (const long long*)step_dev;
  const int with_hist = e->hist_each_step ? 1 : 0;
  if (prime && n_steps > 0) {   // sample the step *step_dev from the parameters as they are: tables, site values, prior terms
    vc_launch_tail(e->d, e->b, params, grad, sd, seed, a, 1, 0, VcXb{}, st);
    vc_launch_omega(e->d, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, 1, with_hist, st);
    e->plain_steps = 0;
  }
  VcBufs b2 = e->b;
  b2.step_ctr = (long long*)step_dev;
  b2.adam_lr0 = a.lr0; b2.adam_lrd_l = a.lrd_l; b2.adam_b1l = a.b1l; b2.adam_b2l = a.b2l; b2.adam_kind = a.kind;
  // Tutorial flow (U-only kernel with its own nu_omega partials, no histogram terms per step): from the third step after the
  // tables were primed, K_tail's gene blocks and K_omega's blocks go out as ONE launch (vc_launch_tail_merged: why that is
  // safe); the first two steps fill both halves of the loss terms the absent cell blocks would have written.
  const int tail_kind = fused_tail_kind(e);
  const bool merged = tail_kind == 1, tail2 = tail_kind == 2;
  for (int64_t i = 0; i < n_steps; ++i) {
    // vc_set_loss_every(k > 1): the gradient-only kernel except at every k-th likelihood launch
    const vc_main_launch_fn fn = (e->loss_every > 1 && e->main_fn_nl && (e->loss_ctr % e->loss_every) != 0) ? e->main_fn_nl : e->main_fn;
    ++e->loss_ctr;
    if (e->timing) {
      if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
      auto& pr = e->ev_pool[e->ev_used++];
      HIPCHK(e, hipEventRecord(pr.first, st));
      fn(e->d, b2, st);
      HIPCHK(e, hipEventRecord(pr.second, st));
    } else {
      fn(e->d, b2, st);
    }
    if (merged && e->plain_steps >= 2) {
      vc_launch_tail_merged(e->d, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, st);
    } else if (tail2) {
      vc_launch_tail2(e->d, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, with_hist, st);
    } else {
      vc_launch_tail(e->d, e->b, params, grad, sd, seed, a, 0, 0, VcXb{}, st);
      vc_launch_omega(e->d, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, 0, with_hist, st);
      if (e->plain_steps < 2) ++e->plain_steps;
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

extern "C" int vc_svi_run_particles(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, int64_t step0, float* grad,
                                    float* grad_acc, double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq,
                                    double lr, double lrd, double beta1, double beta2, double adam_eps, double clip_norm,
                                    int num_particles, int64_t n_steps, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_svi_run_particles before vc_finalize");
  if (e->cfg.world_size != 1)
    return e->fail(VC_ERR_STATE, "vc_svi_run_particles applies the optimiser right after the particles' average: single rank only "
                                 "(cells sharded: vc_elbo_grad per particle, average, all-reduce, vc_clipped_adam)");
  if (!params || !grad || !grad_acc || !exp_avg || !exp_avg_sq || !step_dev) return e->fail(VC_ERR_ARG, "vc_svi_run_particles: null buffer");
  if (num_particles < 1 || n_steps < 0) return e->fail(VC_ERR_ARG, "vc_svi_run_particles: num_particles >= 1, n_steps >= 0");
  hipStream_t st = (hipStream_t)hip_stream;
  const int K = num_particles;
  if (K > VC_MAX_PARTICLES) return e->fail(VC_ERR_UNSUPPORTED, "vc_svi_run_particles: at most %d particles", VC_MAX_PARTICLES);
  const long long total = e->layout.total, header = e->layout.header;
  if (!e->particle_lsum) TRY(e->dalloc(&e->particle_lsum, VC_MAX_PARTICLES));
  // workspaces, gradient buffer, stream and events of the particles beyond the first (created once, at the first call that
  // needs them; the copies start from the first particle's content: what vc_finalize left there for good is in them too)
  while ((int)e->particles.size() < K - 1) {
    vc_engine::Particle pt;
    pt.b = e->b;
    VcBufs& b = pt.b;
    for (int sidx = 0; sidx < VC_SITE_COUNT; ++sidx) TRY(e->dclone(&b.lat[sidx]));
    TRY(e->dclone(&b.eps_used)); TRY(e->dclone(&b.GT)); TRY(e->dclone(&b.CT));
    TRY(e->dclone(&b.lat_delta)); TRY(e->dclone(&b.lat_sgam)); TRY(e->dclone(&b.lat_phi)); TRY(e->dclone(&b.lat_omega));
    TRY(e->dclone(&b.lat_domega)); TRY(e->dclone(&b.GO)); TRY(e->dclone(&b.CO)); TRY(e->dclone(&b.LO)); TRY(e->dclone(&b.LP));
    TRY(e->dclone(&b.PW)); TRY(e->dclone(&b.PWM)); TRY(e->dclone(&b.WT)); TRY(e->dclone(&b.HL)); TRY(e->dclone(&b.HD));
    TRY(e->dalloc(&pt.grad, (size_t)total));
    HIPCHK(e, hipMemset(pt.grad, 0, sizeof(float) * (size_t)total));
    HIPCHK(e, hipStreamCreateWithFlags(&pt.st, hipStreamNonBlocking));
    HIPCHK(e, hipEventCreateWithFlags(&pt.main_done, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&pt.done, hipEventDisableTiming));
    e->particles.push_back(pt);
  }
  if (!e->ev_params) {
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_params, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_main0, hipEventDisableTiming));
  }
  if (e->particle_bufs_n < K) {
    if (!e->particle_bufs_dev) TRY(e->dalloc(&e->particle_bufs_dev, VC_MAX_PARTICLES));
    std::vector<VcBufs> hb((size_t)K);
    for (int k = 0; k < K; ++k) hb[(size_t)k] = k == 0 ? e->b : e->particles[(size_t)k - 1].b;
    HIPCHK(e, hipMemcpy(e->particle_bufs_dev, hb.data(), sizeof(VcBufs) * (size_t)K, hipMemcpyHostToDevice));
    e->particle_bufs_n = K;
    HIPCHK(e, hipDeviceSynchronize());        // (the clones above were copied on the null stream; once per engine and K)
  }
  VcParticleGrads pg;
  pg.K = K;
  for (int k = 0; k < VC_MAX_PARTICLES; ++k) pg.g[k] = k == 0 ? grad : (k < K ? e->particles[k - 1].grad : nullptr);
  // How the particles of a step are laid out in launches (tuning.particles_layout, measured at 50k x 2k, K = 3, profiles/r04_particles.md):
  //   "batched" (default; fast kernel set): K_pre of all particles as ONE launch, the K likelihood kernels, K_post of all particles
  //             as one launch, K_fin of each + average + ClippedAdam as one launch = K + 3 launches per step
  //   "serial":  K_pre, K_main, K_post, K_fin per particle on the caller's stream, then average, ClippedAdam = 4 K + 2 launches
  //   "streams": as "serial" with particle k >= 1 on a stream of its own (the small launches of one particle beside the
  //             likelihood kernel of another): the event records and cross-stream waits cost what the overlap gains
  const bool streams = e->tun.particles_layout == 2;
  const bool batched = !e->d.generic && e->tun.particles_layout == 0;
  const bool serial = !streams;
  for (int64_t i = 0; i < n_steps && batched; ++i) {
    vc_launch_pre_particles(e->d, e->b, e->particle_bufs_dev, params, seed, (const long long*)step_dev, e->hist_each_step ? 1 : 0, K, st);
    for (int k = 0; k < K; ++k) {
      const VcBufs& b = k == 0 ? e->b : e->particles[(size_t)k - 1].b;
      if (e->timing) {
        if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
        auto& pr = e->ev_pool[e->ev_used++];
        HIPCHK(e, hipEventRecord(pr.first, st));
        e->main_fn(e->d, b, st);
        HIPCHK(e, hipEventRecord(pr.second, st));
      } else {
        e->main_fn(e->d, b, st);
      }
    }
    vc_launch_post_particles(e->d, e->b, e->particle_bufs_dev, pg, params, st);
    vc_launch_particle_fin_adam(e->d, e->particle_bufs_dev, pg, params, loss_dev, (long long)loss_slots, (long long)(step0 + i),
                                (long long*)step_dev, e->particle_lsum, exp_avg, exp_avg_sq, e->hyper(lr, lrd, beta1, beta2, adam_eps, clip_norm),
                                (int)header, total, st);
  }
  for (int64_t i = 0; i < n_steps && !batched; ++i) {
    // Particle k runs the unfused sequence K_pre -> K_main -> K_post -> K_fin on the Philox stream (seed, t K + k), t read from
    // the device counter, on its own HIP stream and workspaces.  The parameters are the same for all of them (the optimiser
    // runs once, behind the average), so K_pre of every particle starts at once; the likelihood kernels are chained one behind
    // the other (each fills the chip; side by side they would only share it), and K_post / K_fin of particle k run beside
    // K_main of particle k + 1.
    if (K > 1 && !serial) {
      HIPCHK(e, hipEventRecord(e->ev_params, st));
      for (int k = 1; k < K; ++k) HIPCHK(e, hipStreamWaitEvent(e->particles[k - 1].st, e->ev_params, 0));
    }
    for (int k = 0; k < K; ++k) {
      const VcBufs& b = k == 0 ? e->b : e->particles[k - 1].b;
      hipStream_t sk = (k == 0 || serial) ? st : e->particles[k - 1].st;
      vc_launch_pre(e->d, b, params, nullptr, seed, 0, (const long long*)step_dev, 0, e->hist_each_step ? 1 : 0, sk, K, k);
      if (k > 0 && !serial) HIPCHK(e, hipStreamWaitEvent(sk, k == 1 ? e->ev_main0 : e->particles[k - 2].main_done, 0));
      if (e->timing) {
        if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
        auto& pr = e->ev_pool[e->ev_used++];
        HIPCHK(e, hipEventRecord(pr.first, sk));
        e->main_fn(e->d, b, sk);
        HIPCHK(e, hipEventRecord(pr.second, sk));
      } else {
        e->main_fn(e->d, b, sk);
      }
      if (k + 1 < K && !serial) HIPCHK(e, hipEventRecord(k == 0 ? e->ev_main0 : e->particles[k - 1].main_done, sk));
      // (the step counter advances once per step, behind the average: no K_post touches it)
      vc_launch_post(e->d, b, params, pg.g[k], nullptr, sk);
      vc_launch_fin(e->d, b, params, pg.g[k], e->particle_lsum + k, 1, (long long)(step0 + i), nullptr, sk);
      if (k > 0 && !serial) HIPCHK(e, hipEventRecord(e->particles[k - 1].done, sk));
    }
    if (!serial) for (int k = 1; k < K; ++k) HIPCHK(e, hipStreamWaitEvent(st, e->particles[k - 1].done, 0));
    // average of the K gradients and losses in particle order (what a host loop's g_0 + g_1 + ... and its division by K give),
    // left in the first particle's buffer; advances the step counter
    vc_launch_particle_avg(pg, total, loss_dev, (long long)loss_slots, (long long)(step0 + i), (long long*)step_dev, st);
    VcAdamHyper hy = e->hyper(lr, lrd, beta1, beta2, adam_eps, clip_norm);
    hy.frozen_off = header;             // (the flat update starts behind the header)
    vc_launch_adam(params + header, grad + header, exp_avg, exp_avg_sq, total - header, hy, 0, (const long long*)step_dev, nullptr, nullptr, 0, st);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

extern "C" int vc_set_loss_every(vc_engine* e, int32_t k) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_set_loss_every before vc_finalize");
  if (k < 1) return e->fail(VC_ERR_ARG, "vc_set_loss_every: k >= 1");
  if (k > 1 && !e->main_fn_nl)
    return e->fail(VC_ERR_UNSUPPORTED, "vc_set_loss_every: no gradient-only likelihood kernel for this configuration (it exists for "
                                       "negative-binomial noise on the compiled fast kernel set)");
  e->loss_every = k;
  e->loss_ctr = 0;
  return VC_OK;
}

extern "C" int vc_exchange_size(const vc_engine* e, int64_t* n_floats) {
  if (!e || !n_floats) return VC_ERR_ARG;
  if (!e->finalized) return VC_ERR_STATE;
  *n_floats = e->xb_total;
  return VC_OK;
}

extern "C" int vc_comm_rccl_unique_id(const char* rccl_path, void* id_out) {
  if (!id_out) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  std::string err;
  if (!g_rccl.load(rccl_path, &err)) { g_create_error = err; return VC_ERR_STATE; }
  VcNcclId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) { g_create_error = std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed"); return VC_ERR_STATE; }
  memcpy(id_out, id.internal, sizeof id.internal);
  return VC_OK;
  VC_GUARD_END((vc_engine*)nullptr)
}

extern "C" int vc_comm_init_rccl(vc_engine* e, const char* rccl_path, const void* id_bytes) {
  if (!e || !id_bytes) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (e->comm) return e->fail(VC_ERR_STATE, "vc_comm_init_rccl: the engine already has a communicator");
  std::string err;
  if (!g_rccl.load(rccl_path, &err)) return e->fail(VC_ERR_STATE, "%s", err.c_str());
  VcNcclId id;
  memcpy(id.internal, id_bytes, sizeof id.internal);
  const int rc = g_rccl.CommInitRank(&e->comm, e->cfg.world_size, id, e->cfg.rank);
  if (rc != 0) {
    e->comm = nullptr;
    return e->fail(VC_ERR_STATE, "ncclCommInitRank(rank %d of %d): %s", e->cfg.rank, e->cfg.world_size,
                   g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed");
  }
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_comm_allreduce(vc_engine* e, float* buf, int64_t n, void* hip_stream) {
  if (!e || !buf || n < 0) return VC_ERR_ARG;
  if (!e->comm) return e->fail(VC_ERR_STATE, "vc_comm_allreduce before vc_comm_init_rccl");
  const int rc = g_rccl.AllReduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, e->comm, (hipStream_t)hip_stream);
  if (rc != 0) return e->fail(VC_ERR_STATE, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed");
  return VC_OK;
}

extern "C" int vc_p2p_alloc(vc_engine* e, void* ipc_handle_out) {
  if (!e || !ipc_handle_out) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_p2p_alloc before vc_finalize");
  if (e->p2p_own) return e->fail(VC_ERR_STATE, "vc_p2p_alloc: the region exists already");
  if (e->cfg.world_size > VC_P2P_MAX_RANKS) return e->fail(VC_ERR_UNSUPPORTED, "peer-to-peer exchange: at most %d ranks", VC_P2P_MAX_RANKS);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI documents a 64-byte handle");
  VcP2p& p = e->p2p;
  p.world = e->cfg.world_size; p.rank = e->cfg.rank;
  p.flag_words = p.world * 16;                                   // one 64-byte line per flag
  p.slot_floats = (e->xb_total + 15) / 16 * 16;
  // (+ the per-block flags of the one-launch exchange: [world][gene blocks + cell blocks of the widest shard + the loss block] words)
  e->p2p_nblk = e->d.nb_post_gene + e->xb_pw_cap + 1;
  e->p2p_bflag_off = (long long)p.flag_words + 2 * (long long)p.slot_floats;
  const size_t bytes = ((size_t)e->p2p_bflag_off + ((size_t)p.world * (size_t)e->p2p_nblk + 15) / 16 * 16) * sizeof(float);
  // Fine-grained device memory: peers raise the flags with remote stores and the owner polls them in LOCAL memory; on
  // coarse-grained memory (plain hipMalloc) the owner's L2 may keep serving the old line until a kernel boundary.
  // Tried in this order: fine-grained, uncached, plain (the last only so that a driver without IPC for the first two still
  // runs the opt-in path; p2p_mem_kind says which one the region got)
  const unsigned kinds[3] = {hipDeviceMallocFinegrained, hipDeviceMallocUncached, hipDeviceMallocDefault};
  hipIpcMemHandle_t h;
  hipError_t last = hipSuccess;
  for (int k = 0; k < 3 && !e->p2p_own; ++k) {
    void* ptr = nullptr;
    last = hipExtMallocWithFlags(&ptr, bytes, kinds[k]);
    if (last != hipSuccess) { (void)hipGetLastError(); continue; }
    last = hipMemset(ptr, 0, bytes);
    if (last == hipSuccess) last = hipDeviceSynchronize();
    if (last == hipSuccess) last = hipIpcGetMemHandle(&h, ptr);
    if (last != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ptr); continue; }
    e->p2p_own = ptr;
    e->p2p_mem_kind = k;
  }
  if (!e->p2p_own) return e->fail(VC_ERR_HIP, "vc_p2p_alloc: no IPC-exportable region: %s", hipGetErrorString(last));
  memcpy(ipc_handle_out, &h, sizeof h);
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_p2p_connect(vc_engine* e, const void* all_handles) {
  if (!e || !all_handles) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (!e->p2p_own) return e->fail(VC_ERR_STATE, "vc_p2p_connect before vc_p2p_alloc");
  if (e->p2p_connected) return e->fail(VC_ERR_STATE, "vc_p2p_connect called twice");
  VcP2p& p = e->p2p;
  for (int q = 0; q < p.world; ++q) {
    if (q == p.rank) { p.region[q] = e->p2p_own; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)all_handles + (size_t)q * sizeof h, sizeof h);
    void* ptr = nullptr;
    hipError_t er = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (er != hipSuccess) {
      for (int k = 0; k < q; ++k)
        if (k != p.rank && p.region[k]) { (void)hipIpcCloseMemHandle(p.region[k]); p.region[k] = nullptr; }
      return e->fail(VC_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(er));
    }
    p.region[q] = ptr;
  }
  if (!e->p2p_verdict) {
    TRY(e->dalloc(&e->p2p_verdict, 1));
    HIPCHK(e, hipMemset(e->p2p_verdict, 0, sizeof(unsigned long long)));
  }
  {
    // device tables the kernels index by rank: [0] the regions, [1] / [2] the slot of parity 0 / 1 inside every region
    void* tab[3 * VC_P2P_MAX_RANKS] = {};
    for (int q = 0; q < p.world; ++q) {
      tab[q] = p.region[q];
      for (int par = 0; par < 2; ++par)
        tab[VC_P2P_MAX_RANKS * (1 + par) + q] = reinterpret_cast<float*>(p.region[q]) + (size_t)p.flag_words + (size_t)par * (size_t)p.slot_floats;
    }
    if (!e->p2p_tab) TRY(e->dalloc(&e->p2p_tab, 3 * VC_P2P_MAX_RANKS));
    HIPCHK(e, hipMemcpy(e->p2p_tab, tab, sizeof tab, hipMemcpyHostToDevice));
  }
  e->p2p_connected = true;
  e->p2p_step = 0;
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_svi_run_sharded(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad, float* xbuf,
                                  double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr,
                                  double lrd, double beta1, double beta2, double adam_eps, double clip_norm, int prime,
                                  int phase, int64_t n_steps, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_svi_run_sharded before vc_finalize");
  VC_NO_GENERIC(e, "vc_svi_run_sharded");
  if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev || !xbuf)
    return e->fail(VC_ERR_ARG, "vc_svi_run_sharded: null buffer (step counter and exchange buffer are required)");
  if (phase != VC_PHASE_A && phase != VC_PHASE_B && phase != VC_PHASE_AB) return e->fail(VC_ERR_ARG, "vc_svi_run_sharded: bad phase %d", phase);
  if (n_steps < 0 || (phase != VC_PHASE_AB && n_steps != 1)) return e->fail(VC_ERR_ARG, "vc_svi_run_sharded: phase A / B take exactly one step");
  if (phase == VC_PHASE_AB && e->cfg.world_size > 1 && !e->comm && !e->p2p_connected)
    return e->fail(VC_ERR_STATE, "vc_svi_run_sharded(VC_PHASE_AB) on %d ranks needs vc_comm_init_rccl or vc_p2p_connect first", e->cfg.world_size);
  const bool use_p2p = phase == VC_PHASE_AB && e->p2p_connected;
  if (!(lr > 0.0) || !(lrd > 0.0) || !(beta1 > 0.0 && beta1 < 1.0) || !(beta2 > 0.0 && beta2 < 1.0))
    return e->fail(VC_ERR_ARG, "vc_svi_run_sharded: lr, lrd must be positive and the betas inside (0, 1)");
  hipStream_t st = (hipStream_t)hip_stream;
  VcAdamArgs a;
  e->fill(a, exp_avg, exp_avg_sq, lr, lrd, beta1, beta2, adam_eps, clip_norm);
  const long long* sd = (const long long*)step_dev;
  const int with_hist = e->hist_each_step ? 1 : 0;
  // A single-rank engine whose U-only kernel keeps the nu_omega partials per lane (pw_lane) stores no per-cell rows -- but phase A
  // builds the exchange buffer's per-cell-block rows from exactly those.  The sharded phases therefore run such an engine as a rank
  // of an N > 1 run would run it: K_main without its own partials (the tiling is the same: pw_lane holds nothing in the LDS).
  VcDims ds = e->d;
  vc_main_launch_fn main_fn = e->main_fn;
  if (ds.pw_lane) { ds.pw_lane = 0; ds.pw_inline = 0; ds.pw_slots = 0; main_fn = e->main_fn_rows; }
  VcXb xb;
  xb.x = xbuf; xb.pw_off = e->xb_pw_off; xb.pw_cap = e->xb_pw_cap; xb.loss_off = e->xb_loss_off; xb.sis = e->sis;
  if (prime && n_steps > 0 && phase != VC_PHASE_B) {      // sampling is rank-local: the single-rank priming launches
    vc_launch_tail(ds, e->b, params, grad, sd, seed, a, 1, 0, VcXb{}, st);
    vc_launch_omega(ds, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, 1, with_hist, st);
  }
  VcBufs b2 = e->b;
  b2.step_ctr = (long long*)step_dev;
  b2.adam_lr0 = a.lr0; b2.adam_lrd_l = a.lrd_l; b2.adam_b1l = a.b1l; b2.adam_b2l = a.b2l; b2.adam_kind = a.kind;
  // Round 6, opt-in (vc_tuning.p2p_one_launch with the peer-to-peer exchange): phases A and B in ONE launch, the exchange at block
  // granularity inside it (vc_tail_x_kernel).  Admitted where every block that may spin on another is resident at one 1024-thread block
  // per CU -- gene blocks + the cell blocks of the widest shard + the loss block <= 240 -- and the cell blocks are K_omega's (256 cells).
  const bool one_launch = use_p2p && e->tun.p2p_one_launch && !e->tun.p2p_separate && ds.tail_tc == 256 &&
                          ds.nb_post_gene + e->xb_pw_cap + 1 <= 240 && ds.nb_tail_cell <= e->xb_pw_cap;
  for (int64_t i = 0; i < n_steps; ++i) {
    if (one_launch) {
      if (e->timing) {
        if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
        auto& pr = e->ev_pool[e->ev_used++];
        HIPCHK(e, hipEventRecord(pr.first, st));
        main_fn(ds, b2, st);
        HIPCHK(e, hipEventRecord(pr.second, st));
      } else {
        main_fn(ds, b2, st);
      }
      const size_t slot_off = (size_t)e->p2p.flag_words + (size_t)(e->p2p_step & 1) * (size_t)e->p2p.slot_floats;
      VcXb xw = xb, xr = xb;
      xw.x = reinterpret_cast<float*>(e->p2p_own) + slot_off;
      xw.xmode = 1;
      xr.nslots = e->p2p.world; xr.xmode = 1;
      xr.slots = reinterpret_cast<const float* const*>(e->p2p_tab) + VC_P2P_MAX_RANKS * (1 + (int)(e->p2p_step & 1));
      VcGateX gx;
      gx.regions = reinterpret_cast<void* const*>(e->p2p_tab); gx.world = e->p2p.world; gx.rank = e->p2p.rank;
      gx.step = e->p2p_step; gx.status = e->b.status; gx.timeout_ticks = (unsigned long long)(e->p2p_timeout_s * 1e8);
      gx.bflag_off = e->p2p_bflag_off; gx.nblk = e->p2p_nblk;
      vc_launch_tail_x(ds, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, with_hist, xw, xr, gx, e->xb_pw_cap, st);
      e->p2p_step++;
      continue;
    }
    if (phase != VC_PHASE_B) {
      if (e->timing) {
        if (e->ev_used == e->ev_pool.size()) TRY(e->drain_events());
        auto& pr = e->ev_pool[e->ev_used++];
        HIPCHK(e, hipEventRecord(pr.first, st));
        main_fn(ds, b2, st);
        HIPCHK(e, hipEventRecord(pr.second, st));
      } else {
        main_fn(ds, b2, st);
      }
      VcXb xa = xb;
      if (use_p2p)        // phase A writes this rank's slot of the step's parity; K_xchg sums every rank's slot into xbuf
        xa.x = reinterpret_cast<float*>(e->p2p_own) + e->p2p.flag_words + (size_t)(e->p2p_step & 1) * (size_t)e->p2p.slot_floats;
      vc_launch_tail(ds, e->b, params, grad, sd, seed, a, 0, 1, xa, st);
    }
    VcXb xbb = xb;
    VcGate gate;
    if (use_p2p && !e->tun.p2p_separate) {
      // Round 6: the exchange FOLDED into phase B -- no launch for the sum: phase B's blocks pass the exchange's gate themselves
      // (block 0 publishes and waits; the kernel boundary behind phase A released this rank's slot) and add the ranks' slots of this
      // step's parity where they read them, in rank order (vc_xget: the bits of the separate kernel).  K_main -> phase A -> phase B.
      const size_t slot_off = (size_t)e->p2p.flag_words + (size_t)(e->p2p_step & 1) * (size_t)e->p2p.slot_floats;
      (void)slot_off;
      xbb.nslots = e->p2p.world;
      xbb.slots = reinterpret_cast<const float* const*>(e->p2p_tab) + VC_P2P_MAX_RANKS * (1 + (int)(e->p2p_step & 1));
      gate.regions = reinterpret_cast<void* const*>(e->p2p_tab); gate.world = e->p2p.world; gate.rank = e->p2p.rank;
      gate.step = e->p2p_step; gate.status = e->b.status;
      gate.timeout_ticks = (unsigned long long)(e->p2p_timeout_s * 1e8); gate.verdict = e->p2p_verdict;
      e->p2p_step++;
    } else if (use_p2p) {
      vc_launch_p2p_xchg(e->p2p, reinterpret_cast<void* const*>(e->p2p_tab), e->p2p_step, xbuf, (long long)e->xb_total, e->b.status,
                         e->p2p_timeout_s, e->p2p_verdict, st);
      e->p2p_step++;
    } else if (phase == VC_PHASE_AB && e->comm) {      // (a 1-rank communicator is summed too: the single-GPU measurement of this path)
      const int rc = g_rccl.AllReduce(xbuf, xbuf, (size_t)e->xb_total, /*ncclFloat32*/ 7, /*ncclSum*/ 0, e->comm, st);
      if (rc != 0) return e->fail(VC_ERR_STATE, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed");
    }
    if (phase != VC_PHASE_A)
      vc_launch_phase_b(ds, e->b, params, grad, sd, seed, a, loss_dev, (long long)loss_slots, with_hist, xbb, st,
                        xbb.nslots > 0 ? &gate : nullptr);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

extern "C" int vc_svi_step_fused(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad,
                                 double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr,
                                 double lrd, double beta1, double beta2, double adam_eps, double clip_norm, int prime,
                                 void* hip_stream) {
  return vc_svi_run_fused(e, params, seed, step_dev, grad, loss_dev, loss_slots, exp_avg, exp_avg_sq, lr, lrd, beta1, beta2,
                          adam_eps, clip_norm, prime, 1, hip_stream);
}

extern "C" int vc_sample_guide(vc_engine* e, const float* params, const float* eps, uint64_t seed, int64_t step,
                               void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_sample_guide before vc_finalize");
  if (!params) return e->fail(VC_ERR_ARG, "vc_sample_guide: null params");
  vc_launch_pre(e->d, e->b, params, eps, seed, (long long)step, nullptr, 0, 0, (hipStream_t)hip_stream);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return VC_OK;
}

// device pointer and length of a readable site (sampled or deterministic), nullptr if the id is unknown / absent
static const float* site_source(vc_engine* e, int site, long long* sz) {
  if (site >= 0 && site < VC_SITE_COUNT) {
    if (!site_exists(e, site)) return nullptr;
    *sz = site_size(e, site);
    return e->b.lat[site];
  }
  if (site == VC_DET_PHI) { *sz = e->d.Nc; return e->b.lat_phi; }
  if (site == VC_DET_OMEGA) { *sz = e->d.Nc; return e->b.lat_omega; }
  if (site == VC_DET_EPS) { *sz = e->layout.eps_total; return e->b.eps_used; }
  return nullptr;
}

extern "C" int vc_sample_posterior(vc_engine* e, const float* params, uint64_t seed, int64_t step0, int64_t n_draws,
                                   int n_sites, const int* sites, float* const* out_dev, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  VC_GUARD_BEGIN
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_sample_posterior before vc_finalize");
  if (!params || n_draws < 0 || n_sites < 0 || (n_sites > 0 && (!sites || !out_dev)))
    return e->fail(VC_ERR_ARG, "vc_sample_posterior: bad arguments");
  std::vector<const float*> src(n_sites);
  std::vector<long long> len(n_sites);
  for (int k = 0; k < n_sites; ++k) {
    src[k] = (sites[k] == VC_DET_EPS) ? nullptr : site_source(e, sites[k], &len[k]);
    if (!src[k]) return e->fail(VC_ERR_ARG, "vc_sample_posterior: site %d not in this model", sites[k]);
    if (n_draws > 0 && !out_dev[k]) return e->fail(VC_ERR_ARG, "vc_sample_posterior: null output for site %d", sites[k]);
  }
  hipStream_t st = (hipStream_t)hip_stream;
  for (int64_t i = 0; i < n_draws; ++i) {
    vc_launch_pre(e->d, e->b, params, nullptr, seed, (long long)(step0 + i), nullptr, 0, 0, st);
    for (int k = 0; k < n_sites; ++k)
      HIPCHK(e, hipMemcpyAsync(out_dev[k] + (size_t)i * len[k], src[k], sizeof(float) * len[k], hipMemcpyDeviceToDevice, st));
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "vc_sample_posterior: %s", hipGetErrorString(err));
  return VC_OK;
  VC_GUARD_END(e)
}

extern "C" int vc_expected_logs(vc_engine* e, const float* nu, const float* dnu, const float* phi, const float* omega,
                                const float* logbeta, const float* gamma, float cf_avg, float* out_S, float* out_S2,
                                float* out_U, float* out_U2, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_expected_logs before vc_finalize");
  const VcDims& d = e->d;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  if (!nu || !phi || !out_S || !out_S2) return e->fail(VC_ERR_ARG, "vc_expected_logs: null nu / phi / output");
  if (d.Nb > 0 && !dnu) return e->fail(VC_ERR_ARG, "vc_expected_logs: the model has batches, dnu is required");
  if (vel != (out_U != nullptr) || vel != (out_U2 != nullptr))
    return e->fail(VC_ERR_ARG, "vc_expected_logs: out_U / out_U2 are required for the velocity model and only there");
  if (vel && (!omega || !logbeta || !gamma)) return e->fail(VC_ERR_ARG, "vc_expected_logs: null omega / logbeta / gamma");
  if (d.Nc == 0 || d.Ng == 0) return VC_OK;
  vc_launch_expected_logs(d, e->b, nu, dnu, phi, omega, logbeta, gamma, cf_avg, out_S, out_S2, out_U, out_U2,
                          (hipStream_t)hip_stream);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return e->fail(VC_ERR_HIP, "vc_expected_logs: %s", hipGetErrorString(err));
  return VC_OK;
}

extern "C" int vc_read_site(vc_engine* e, int site, float* host_out, int64_t n, void* hip_stream) {
  if (!e || !host_out) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_read_site before vc_finalize");
  long long sz = 0;
  const float* src = site_source(e, site, &sz);
  if (!src) return e->fail(VC_ERR_ARG, "vc_read_site: site %d unknown or not in this model", site);
  if (n != sz) return e->fail(VC_ERR_ARG, "vc_read_site(%d): expected %lld values, got %lld", site, sz, (long long)n);
  HIPCHK(e, hipStreamSynchronize((hipStream_t)hip_stream));
  HIPCHK(e, hipMemcpy(host_out, src, sizeof(float) * sz, hipMemcpyDeviceToHost));
  return VC_OK;
}

extern "C" int vc_get_stats(const vc_engine* e, vc_stats* out) {
  if (!e || !out) return VC_ERR_ARG;
  memset(out, 0, sizeof *out);
  const VcDims& d = e->d;
  const int nmat = d.kind == VC_KIND_VFULL ? 2 : 1;
  out->algorithmic_bytes = (int64_t)nmat * 4 * d.Ng * (int64_t)d.Nc;
  out->streamed_bytes = (int64_t)nmat * (d.c16 ? 2 : 4) * d.Ng_pad * (int64_t)d.Nc;
  out->count_storage_bytes = d.c16 ? 2 : 4;
  out->main_grid = d.n_main_wg;
  out->main_block = 256;
  out->main_kind = d.kind;
  out->hist_on_device = e->hist_on_device;
  out->setup_transient_bytes = (int64_t)e->setup_transient_bytes;
  for (int p = 0; p < 4; ++p) out->pass_cells[p] = d.pass_cw[p];
  out->launches_per_step = fused_launches_per_step(e);
  out->pw_inline = d.pw_inline;
  out->generic = d.generic;
  snprintf(out->main_kernel_name, sizeof out->main_kernel_name, "vc_main_kernel<%d,%d,%s,gpl%d%s%s>", d.H, d.nbk, e->main_name, d.gpl,
           d.c16 ? ",u16" : "", d.pw_lane ? ",pwl" : "");
  out->onehot_batches = d.onehot ? d.Nb : 0;
  // The row the signature matched, and the row the steps of vc_svi_run_fused / vc_svi_run_sharded actually LAUNCH: a matched row
  // serves only the launch structures its kind bits name (a single-rank engine kept at three launches by tuning.no_tail2 /
  // pw_inline matches a "*_rank" row and still runs the run-time-flag kernels) -- report what runs, and the match beside it.
  int speck = 0;
  if (e->cfg.world_size > 1) speck = VC_SPECK_SHARDED;
  else { const int tk = fused_tail_kind(e); speck = tk == 1 ? VC_SPECK_MERGED : (tk == 2 ? VC_SPECK_TAIL2 : 0); }
  const int used = (d.spec > 0 && (VC_SPECS[d.spec].kind & speck)) ? d.spec : VC_SPEC_NONE;
  out->pw_lane = d.pw_lane;
  out->hist_split = d.hist_dense ? e->b.n_hc_split : 0;
  out->tail_spec = used;
  out->tail_spec_matched = d.spec;
  memset(out->tail_spec_name, 0, sizeof(out->tail_spec_name));
  strncpy(out->tail_spec_name, VC_SPECS[used].name, sizeof(out->tail_spec_name) - 1);
  return VC_OK;
}

extern "C" int vc_get_histogram(const vc_engine* e, int64_t* n_entries, int32_t* ptr_out, float* val_out, float* cnt_out) {
  if (!e || !n_entries) return VC_ERR_ARG;
  if (!e->finalized) return VC_ERR_STATE;
  *n_entries = (int64_t)e->h_val_host.size();
  if (ptr_out) memcpy(ptr_out, e->h_ptr_host.data(), e->h_ptr_host.size() * sizeof(int));
  if (val_out) memcpy(val_out, e->h_val_host.data(), e->h_val_host.size() * sizeof(float));
  if (cnt_out) memcpy(cnt_out, e->h_cnt_host.data(), e->h_cnt_host.size() * sizeof(float));
  return VC_OK;
}

extern "C" int vc_get_status(vc_engine* e, int64_t* first_bad_step, int64_t* n_bad, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_get_status before vc_finalize");
  long long h[4] = {0, 0, 0, 0};
  HIPCHK(e, hipStreamSynchronize((hipStream_t)hip_stream));
  HIPCHK(e, hipMemcpy(h, e->b.status, sizeof h, hipMemcpyDeviceToHost));
  if (h[2] > 0)
    return e->fail(VC_ERR_STATE, "peer-to-peer exchange: a rank did not publish its buffer of step %lld within %.1f s", h[2] - 1,
                   e->p2p_timeout_s);
  if (first_bad_step) *first_bad_step = h[1] - 1;
  if (n_bad) *n_bad = h[0];
  if (h[0] > 0)
    return e->fail(VC_ERR_NONFINITE, "non-finite loss in %lld step(s), first at step %lld", h[0], h[1] - 1);
  return VC_OK;
}

extern "C" int vc_clear_status(vc_engine* e, void* hip_stream) {
  if (!e) return VC_ERR_ARG;
  if (!e->finalized) return e->fail(VC_ERR_STATE, "vc_clear_status before vc_finalize");
  HIPCHK(e, hipMemsetAsync(e->b.status, 0, 4 * sizeof(long long), (hipStream_t)hip_stream));
  return VC_OK;
}

extern "C" int vc_device_clock_mhz(double window_us, double* mhz_out, void* hip_stream) {
  if (!mhz_out || !(window_us > 0.0) || window_us > 1e6) return VC_ERR_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  unsigned long long* dev = nullptr;
  if (hipMalloc((void**)&dev, 2 * sizeof(unsigned long long)) != hipSuccess) return VC_ERR_HIP;
  unsigned long long h[2] = {0, 0};
  vc_launch_clock_probe((unsigned long long)(window_us * 100.0), dev, st);        // wall clock: 100 MHz
  hipError_t err = hipStreamSynchronize(st);
  if (err == hipSuccess) err = hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (err != hipSuccess || h[1] == 0) return VC_ERR_HIP;
  *mhz_out = (double)h[0] / ((double)h[1] / 100.0);
  return VC_OK;
}

extern "C" int vc_set_timing(vc_engine* e, int enable) {
  if (!e) return VC_ERR_ARG;
  if (enable && e->ev_pool.empty()) {
    for (int i = 0; i < 512; ++i) {
      hipEvent_t a, b2;
      HIPCHK(e, hipEventCreate(&a));
      HIPCHK(e, hipEventCreate(&b2));
      e->ev_pool.emplace_back(a, b2);
    }
  }
  if (enable) { e->ev_used = 0; e->main_ms = 0.0; e->main_launches = 0; }
  e->timing = enable != 0;
  return VC_OK;
}

extern "C" int vc_get_timing(vc_engine* e, double* main_ms_total, int64_t* n_launches) {
  if (!e || !main_ms_total || !n_launches) return VC_ERR_ARG;
  TRY(e->drain_events());
  *main_ms_total = e->main_ms;
  *n_launches = e->main_launches;
  return VC_OK;
}

extern "C" int vc_clipped_adam(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               double lr, double lrd, double beta1, double beta2, double eps, double clip_norm,
                               int64_t t, const int64_t* t_dev, const float* loss_hdr, double* loss_ring,
                               int64_t loss_slots, void* hip_stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || n < 0) return VC_ERR_ARG;
  if (n == 0) return VC_OK;
  VcAdamHyper h;
  h.lr0 = lr; h.lrd = lrd; h.b1 = beta1; h.b2 = beta2; h.eps = (float)eps; h.clip = (float)clip_norm; h.wd = 0.f; h.kind = VC_OPT_CLIPPED_ADAM;
  h.frozen = nullptr; h.frozen_off = 0;
  vc_launch_adam(params, grad, exp_avg, exp_avg_sq, (long long)n, h, (long long)t, (const long long*)t_dev, loss_hdr, loss_ring,
                 (long long)loss_slots, (hipStream_t)hip_stream);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_ERR_HIP;
}

extern "C" int vc_adam_update(int kind, float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                              double lrd, double beta1, double beta2, double eps, double clip_norm, double weight_decay,
                              const uint8_t* frozen, int64_t t, const int64_t* t_dev, const float* loss_hdr, double* loss_ring,
                              int64_t loss_slots, void* hip_stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || n < 0) return VC_ERR_ARG;
  if (kind != VC_OPT_CLIPPED_ADAM && kind != VC_OPT_ADAM) return VC_ERR_ARG;
  if (n == 0) return VC_OK;
  VcAdamHyper h;
  h.lr0 = lr; h.lrd = kind == VC_OPT_ADAM ? 1.0 : lrd; h.b1 = beta1; h.b2 = beta2; h.eps = (float)eps;
  h.clip = kind == VC_OPT_ADAM ? __builtin_inff() : (float)clip_norm; h.wd = (float)weight_decay; h.kind = kind;
  h.frozen = frozen; h.frozen_off = 0;
  vc_launch_adam(params, grad, exp_avg, exp_avg_sq, (long long)n, h, (long long)t, (const long long*)t_dev, loss_hdr, loss_ring,
                 (long long)loss_slots, (hipStream_t)hip_stream);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_ERR_HIP;
}

extern "C" int vc_set_optimizer(vc_engine* e, int kind, double weight_decay, const uint8_t* frozen) {
  if (!e) return VC_ERR_ARG;
  if (kind != VC_OPT_CLIPPED_ADAM && kind != VC_OPT_ADAM) return e->fail(VC_ERR_ARG, "vc_set_optimizer: kind must be VC_OPT_CLIPPED_ADAM or VC_OPT_ADAM");
  if (!(weight_decay >= 0.0)) return e->fail(VC_ERR_ARG, "vc_set_optimizer: negative weight_decay");
  e->opt_kind = kind;
  e->opt_wd = weight_decay;
  e->opt_frozen = frozen;
  return VC_OK;
}
