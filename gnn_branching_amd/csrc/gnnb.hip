// gnnb.hip -- MI355X (gfx950 / CDNA4) GNN branching-score forward pass: HIP kernels + C-ABI.
//
// What runs here is the reference's graphnet/graph_conv.py (EmbedLayerUpdate.forward :77-388,
// ComputeFinalScore.forward :442-470), the argmax of graphnet/graph_score.py :41-47 and the BaBSR heuristic of
// plnn/kw_score_conv.py :41-113, for a batch of B subproblems, re-designed for CDNA4 (DESIGN.md sections 3-5):
//
//   * embeddings mu[k] live in HBM as (B, N_k, 64) fp32, one 256-B row per node;
//   * every node MLP is a chain of exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) run TRANSPOSED: weights are the A operand
//     (staged once per workgroup in LDS, pre-permuted on the host, gnnb_pack.h), the 32 nodes of a tile sit on the lanes,
//     and the accumulators of one layer are the B operands of the next -- no LDS round trip between layers;
//   * node-feature-only sub-chains do not depend on the embeddings: evaluated ONCE per forward (k_pre) and folded
//     into a cached 64-vector per ambiguous node; linear layers that meet are pre-multiplied on the host; every producer's
//     last Linear is deferred into its consumers ("deferred projection", gnnb_pack.h);
//   * node classes (live / ambiguous / scored) are compacted once per forward (k_classify) and the node MLPs run over
//     the lists only;
//   * conv / conv-transpose message passing is a dense local block per tile on the MFMA (k_gather, k_gather16,
//     k_gather_input_update), Linear edges and everything above the last conv layer run per sample out of LDS (k_top);
//   * provably dead work of the reference is not executed: the `ratio` chain (:214-216,228,243,356) and the last
//     round's input-layer update (:360-385), whose result nothing reads.
//
// One translation unit: gnnb_dev.h (fragments, GEMM blocks, tile maps), gnnb_k_mlp.h (setup + node-MLP kernels),
// gnnb_k_gather.h (conv-edge message passing + score head), gnnb_k_fusedq.h (gather + node update in one kernel), gnnb_k_edges.h (other edges, k_top), gnnb_k_misc.h (k_livesum,
// k_babsr, k_gather_scored), gnnb_train.h (online learning) are included below; this file holds the host side and the C-ABI.
//
// gfx950 only.  No HIP call at load time.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <unistd.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>

#include "../../include/gnnb.h"
#include "gnnb_pack.h"
#include "gnnb_train.h"

using namespace gnnb;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "gnnb_dev.h"
#include "gnnb_k_mlp.h"
#include "gnnb_k_gather.h"
#include "gnnb_k_fusedq.h"
#include "gnnb_k_edges.h"
#include "gnnb_k_misc.h"

#define N_PACKS 14   // == PK_COUNT
enum { PK_EMBED, PK_PRE_FWD, PK_PRE_BWD, PK_PRE_INP, PK_PROP, PK_UPD_FWD_E, PK_UPD_FWD_I, PK_UPD_FWD_F, PK_UPD_BWD, PK_UPD_BWD_B,
       PK_UPD_INP, PK_POST_INP, PK_SCORE_B, PK_SCORE_F, PK_COUNT };
static_assert(PK_COUNT == N_PACKS, "pack table");

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  fprintf(stderr, "[gnnb] error: %s\n", buf);   // the BaB harness swallows exceptions (bab_mip.py:73-76): log first
  return code;
}
#define HIPCHK(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) return fail(GNNB_E_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

enum ProfClass {
  PC_EMBED, PC_PRE, PC_PRE_INP, PC_CONV_FWD, PC_CONVT_BWD, PC_DENSE_AGG, PC_PROP_FWD,
  PC_NODE_UPDATE, PC_INPUT_UPDATE, PC_SCORE, PC_ARGMAX, PC_GATHER, PC_GATHER_INPUT, PC_CLASSIFY, PC_LIVESUM, PC_TOP, PC_GATHER_UPDATE, PC_COUNT
};
static const char* kProfNames[PC_COUNT] = {
    "k_embed", "k_pre", "k_pre_inp", "k_conv_fwd", "k_convT_bwd", "k_dense_agg", "k_prop",
    "k_node_update", "k_input_update", "k_score", "k_argmax", "k_gather", "k_gather_input_update", "k_classify", "k_livesum", "k_top", "k_gather_update"};

struct DevEdge {
  float *w_fwd = nullptr, *w_bwd = nullptr, *bias = nullptr;   // conv: tap-major copies; linear: W^T / W, zero-padded
  int ld_fwd = 0, mt_fwd = 0, ksq_fwd = 0, ld_bwd = 0, mt_bwd = 0, ksq_bwd = 0, kpad_fwd = 0, kpad_bwd = 0;
};

struct DevGather {          // one conv edge in one direction, as MFMA gather tables on the device
  bool ok = false;
  GatherGeom g;
  float* cmat = nullptr;
  uint32_t* taps3 = nullptr;   // 32-node tiles: the taps in three bf16 pieces (GatherHost::taps3)
  int* koff = nullptr;
  int* ttab = nullptr;
};

// Host helper threads of a handle, created on the first call that wants them (gnnb_pack_amb_records) and joined by gnnb_destroy: creating
// and joining threads per call cost 0.25 ms of a 0.65-ms pack.  Idle workers sleep on a condition variable.  (A handle is created in the
// process that uses it -- the BaB harness forks first, bab_mip.py:244-249 -- so no thread ever has to survive a fork.)
struct WorkPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, done_cv;
  const std::function<void()>* job = nullptr;
  long gen = 0;
  int busy = 0;
  bool stop = false;
  pid_t owner = getpid();             // a forked child inherits the object but not the threads: it makes a pool of its own (gnnb_pack_amb_records)
  explicit WorkPool(int n) {
    for (int i = 0; i < n; ++i)
      th.emplace_back([this] {
        long seen = 0;
        for (;;) {
          const std::function<void()>* f;
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            f = job;
          }
          (*f)();
          {
            std::lock_guard<std::mutex> lk(m);
            if (--busy == 0) done_cv.notify_one();
          }
        }
      });
  }
  std::mutex callers;                 // one run() at a time: `job` points into the caller's frame (two pipelines on one engine, ctypes drops the GIL)
  void run(const std::function<void()>& f) {      // f on every worker and on the caller; returns when all are done
    std::lock_guard<std::mutex> one(callers);
    {
      std::lock_guard<std::mutex> lk(m);
      job = &f; ++gen; busy = (int)th.size();
    }
    cv.notify_all();
    f();
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return busy == 0; });
  }
  ~WorkPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

struct gnnb_handle {
  WorkPool* work_pool = nullptr;      // see WorkPool
  int T = 2, p = 64, device = 0, n_cu = 256;
  bool use_gather = true;       // MFMA gather for conv edges (false: VALU gather kernels)
  // (k_node_update: 12 waves per workgroup = 3 per SIMD with the bf16x3 blocks (142-152 VGPRs, no scratch); the fp32-MFMA-only
  // form (GNNB_BF3=0) needs 167-181 VGPRs and runs 8 waves per workgroup)
                                // than 8 (16 waves: 27 % slower); k_gather_input_update prefers 8, k_gather 8 x 2 workgroups
  int gather_occ = 2;           // workgroups per CU for k_gather (its LDS footprint is only the tap matrix)
  bool dense_lds = true;        // Linear edges: one workgroup per sample with the source rows in LDS (false: per-tile kernel)
  bool restrict_last = true;    // last backward step of layer 1 only for the scored nodes (nothing else reads it)
  bool s_in_gather = true;      // the sparse gathers compute the bias sums of their edge themselves (fewer k_livesum jobs); GNNB_S_IN_GATHER=0
  bool zero_dead = false;       // GNNB_ZERO_DEAD=1: always write the zero rows of dead nodes (default: only where something reads them)
  int giu_occ = 2;              // workgroups per CU of k_gather_input_update (<= 128 VGPRs: two 8-wave workgroups fit)
  bool bf3 = true;              // node update: 64x64 blocks on the bf16 matrix rate with three-piece operands (fp32 accuracy)
  int gather_sparse = 7;        // gathers behind a ReLU layer walk only the live rows of their window: bit 0 = 16-node forward
                                // gathers, bit 1 = 32-node gathers, bit 2 = the input-layer gather
  bool gather16 = true;         // forward conv edges: 16-node tiles on the 16x16x4 MFMA when their window is smaller
  bool embed_fuse = true;       // round 0: the first forward gather computes the input embedding itself (no k_embed, no mu[0] rows)
  int fuse = 1;                 // conv half-passes as ONE kernel (k_gather_update_q: the aggregate never reaches HBM) wherever that kernel
                                // exists (measured faster at every batch size and on all three networks: base B = 256 0.975 vs 1.014 ms,
                                // deep B = 1024 6.59 vs 7.31 ms, B = 1 0.344 vs 0.359 ms); GNNB_FUSE=0: always two kernels.  Both forms
                                // compute the same arithmetic per node -- bit-identical results -- so this is a pure scheduling choice.
  bool scored_gather = true;    // the restricted last step's aggregate one wave per scored node (k_gather_scored); GNNB_DEV: GNNB_NO_SCORED_GATHER=1
  bool use_top = true;          // fuse the top of the network (last Linear edge, last ReLU layer, property node) into k_top
  bool top_ok = false;          // ... which the bound network allows (set by gnnb_bind_network)
  int clspre_max_b = 1;         // GNNB_CLSPRE_MAX_B: batches up to it classify and run the hoisted feature chains in one launch (k_classify_pre);
                                // measured (base, us): B = 1 27.5 vs 7.6 + 22.1, B = 2 34.0 vs 30.0, B = 8 42.5 vs 31.8 -- a block's share of
                                // the ambiguous nodes is uneven, so beyond one subproblem the two kernels' even dealing wins
  bool gather_bf3 = false;      // GNNB_DEV builds only (GNNB_GATHER_BF3=1): the input update's aggregate on the bf16 matrix rate (rows of layer 1 as three bf16
                                // pieces, gather_tile_sparse_bf3).  Measured SLOWER (base B=256: 115 -> 140 us, docs/DESIGN_HISTORY.md): not instantiated in the shipped library
  int tail_max_b = 1 << 30;     // GNNB_TAIL_MAX_B: batches up to it end in k_scored_tail (scored gather + restricted update + score head in one launch); 0: three kernels
  bool top_fuse_upd = true;     // GNNB_TOP_FUSE_UPD=0: the backward node update of layer L-1 as its own launch behind k_top (it runs inside k_top otherwise)
  int top_split_max = 4;        // GNNB_TOP_SPLIT: 4 (default) = four workgroups per sample while B <= n_cu / 4, two while B <= n_cu / 2; 2 = two at most; 1 = never
  int per_sample_min_b = 0;     // GNNB_PER_SAMPLE_MIN_B: batches below it take the per-tile dense kernel + separate launches
                                // instead of the one-workgroup-per-sample kernels (k_top, k_dense_*_lds), which need a batch
                                // that fills the CUs (B=2: 0.40 vs 0.49 ms, B=64: 0.64 vs 0.66, B=128: 0.96 vs 0.88 ms; 96 is
                                // the break-even).  Off by default: the two paths round differently, and with one path for
                                // every batch size a sample's scores do not depend on what it is batched or sharded with.
  Packs packs;
  std::vector<float> blob;      // the GNN parameters as handed to gnnb_create / gnnb_set_weights / left by gnnb_online_step
  gnnb_train::Trainer* trainer = nullptr;     // online learning (gnnb_online_create)
  float* d_pack[N_PACKS] = {nullptr};
  float* d_zero = nullptr;      // 64 zero floats: where masked gather loads point
  // List counters of a forward (64 ints) live HERE, not in the caller's workspace: a control block per workspace address (CTL_SLOTS
  // of them, least recently used replaced), zero whenever no forward is running on it -- the last workgroup of k_score, the last
  // kernel of a forward and the last reader of the counters, puts them back to zero.  So no launch has to zero them first
  // (k_reset is gone), and what the caller's workspace holds between calls does not matter.
  float* pack_stage = nullptr; size_t pack_stage_floats = 0;     // pinned staging of the weight packs (load_weights)
  int* d_ctl = nullptr;
  const void* ctl_ws[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long ctl_age[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long ctl_clock = 0;
  // gnnb_forward_host: pinned staging of the host inputs, their device image, workspace and outputs (grown on demand)
  float* hs_pinned = nullptr; float* hs_dev = nullptr; size_t hs_floats = 0;
  void* hs_ws = nullptr; size_t hs_ws_bytes = 0;
  float* hs_scores = nullptr; int32_t* hs_dec = nullptr; size_t hs_out_B = 0;
  float* hs_out_pinned = nullptr;
  float* d_s1 = nullptr;        // (N_1) bias sums of edge 1 forward over the (all live) input layer: sum of the weights that reach each node
  int last_proj[MAXL + 2];      // per graph layer: which Linear (LayerId) the rows of mu[k] written by the LAST forward still have to
                                // go through (-1: final) -- the "deferred projection" of gnnb_pack.h; inspection only (gnnb_mu_projection),
                                // gnnb_forward itself keeps this state on its stack
  std::vector<DevGather> gf, gb;   // gf[k]: edge k forward (dst = layer k); gb[k]: edge k transposed (dst = layer k-1)
  bool bound = false;
  std::vector<Edge> edges;       // edges[k], k = 1..L (edges[0] unused)
  std::vector<DevEdge> dev;
  std::vector<int> N;            // graph layer sizes, N[0..L+1]
  std::vector<int> relu_q;       // fixed-layer index of the ReLU of graph layer k
  std::vector<int> hw;           // nodes per bias entry of layer k
  int n_fixed = 0, R = 0;
  int halfpass_limit = 0;
  bool prof = false;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> pending;
  struct Tr { int cls; float ms; };
  std::vector<Tr> trace;          // per-launch record of what gnnb_profile_read resolved (gnnb_profile_trace)
  std::vector<hipEvent_t> pool;
  double prof_ms[PC_COUNT] = {0};
  int64_t prof_n[PC_COUNT] = {0};
  hipStream_t prof_stream = nullptr;
};


static int upload(float** d, const float* h, size_t n) {
  HIPCHK(hipMalloc((void**)d, n * sizeof(float)));
  HIPCHK(hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// (re)build the operand packs of the scorer from a parameter blob and put them on the device
static int load_weights(gnnb_t* h, const float* w_blob, hipStream_t st) {
#ifdef GNNB_DEV
  static const bool timing = std::getenv("GNNB_PACK_TIMING") != nullptr;       // dev aid: where the time of this call goes, to stderr
#else
  constexpr bool timing = false;
#endif
  const auto t_start = std::chrono::steady_clock::now();
  h->blob.assign(w_blob, w_blob + blob_floats());
  build_packs(h->blob.data(), h->packs);
  const auto t_built = std::chrono::steady_clock::now();
  const std::vector<float>* pv[N_PACKS] = {&h->packs.embed, &h->packs.pre_fwd, &h->packs.pre_bwd, &h->packs.pre_inp, &h->packs.prop,
                                           &h->packs.upd_fwd_e, &h->packs.upd_fwd_i, &h->packs.upd_fwd_f, &h->packs.upd_bwd,
                                           &h->packs.upd_bwd_b, &h->packs.upd_inp, &h->packs.post_inp, &h->packs.score_b,
                                           &h->packs.score_f};
  // the packs go through ONE pinned staging buffer: 14 copies out of pageable memory were each staged synchronously by the
  // runtime (0.3 ms of the 1.4 ms this call took behind every online-learning step)
  size_t total = 0;
  for (int i = 0; i < N_PACKS; ++i) total += (pv[i]->size() + 63) & ~(size_t)63;
  if (h->pack_stage_floats < total) {
    if (h->pack_stage) (void)hipHostFree(h->pack_stage);
    h->pack_stage = nullptr; h->pack_stage_floats = 0;
    HIPCHK(hipHostMalloc((void**)&h->pack_stage, total * sizeof(float), hipHostMallocDefault));
    h->pack_stage_floats = total;
  }
  if (!h->d_pack[0]) {                    // one device block, pack i at the offset it has in the staging buffer: one copy per call
    float* base = nullptr;
    HIPCHK(hipMalloc((void**)&base, total * sizeof(float)));
    size_t o = 0;
    for (int i = 0; i < N_PACKS; ++i) { h->d_pack[i] = base + o; o += (pv[i]->size() + 63) & ~(size_t)63; }
  }
  size_t off = 0;
  for (int i = 0; i < N_PACKS; ++i) {
    std::memcpy(h->pack_stage + off, pv[i]->data(), pv[i]->size() * sizeof(float));
    off += (pv[i]->size() + 63) & ~(size_t)63;
  }
  HIPCHK(hipMemcpyAsync(h->d_pack[0], h->pack_stage, total * sizeof(float), hipMemcpyHostToDevice, st));
  const auto t_issued = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(st));       // the staging buffer is reused by the next call
  if (timing) {
    const auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "load_weights: build %.0f us, stage+issue %.0f us (%zu KiB), wait %.0f us\n", us(t_start, t_built), us(t_built, t_issued),
            total * sizeof(float) / 1024, us(t_issued, std::chrono::steady_clock::now()));
  }
  return 0;
}

extern "C" int gnnb_abi_version(void) { return GNNB_ABI_VERSION; }
#ifndef GNNB_SRC_HASH
#define GNNB_SRC_HASH "unhashed-build------------------"
#endif
// the hash of the sources this binary was compiled from (gnn_branching_amd/_lib.py source_hash): the loader refuses a library
// whose id differs from the tree's, the build skips one whose id matches (mtimes are not trusted: *.so ships outside git)
static const char g_build_id[] = "GNNB_BUILD_ID:" GNNB_SRC_HASH;
extern "C" const char* gnnb_build_id(void) { return g_build_id + 14; }
extern "C" const char* gnnb_last_error(void) { return g_err.c_str(); }


// ---- handle options (include/gnnb.h gnnb_set_option) ---------------------------------------------------------------------------------
// The shipped library reads NO environment variable: every switch a caller, a test or bench.py can flip goes through this table.  Each
// option selects between implementations that compute the same scores (bit-identical unless the comment says otherwise), so the parity
// tests use them as independent implementations of one another (INTEGRATION.md lists the test of each).
struct OptDesc { const char* name; int lo, hi; bool before_bind; };
static const OptDesc kOptions[] = {
    {"bf3", 0, 1, false},             // 1: 64x64 blocks on the bf16 matrix rate with three-piece operands; 0: every block exact fp32 MFMA (and no k_top)
    {"fuse", 0, 1, false},            // 1: a conv half-pass is ONE kernel (k_gather_update_q); 0: k_gather + k_node_update
    {"top", 0, 1, false},             // 1: the top of the network in k_top; 0: separate kernels (fp32 MFMA edges)
    {"gather", 0, 1, true},           // 1: MFMA gathers for conv edges; 0: the VALU conv kernels + flat node update
    {"embed_fuse", 0, 1, false},      // 1: round 0's input embedding computed inside the first gather; 0: k_embed writes the rows
    {"dense_lds", 0, 1, true},        // 1: Linear edges one workgroup per sample out of LDS; 0: the per-tile dense kernel (the fallback of wide layers)
    {"tail_max_b", 0, 1 << 30, false},    // batches up to it end in k_scored_tail; 0: k_gather_scored + k_node_update + k_score
    {"top_split", 1, 4, false},       // most workgroups k_top spreads one sample over (1, 2 or 4); 1 = never wait for a partner workgroup
    {"top_fuse_upd", 0, 1, false},    // 1: the backward update of layer L-1 inside k_top; 0: its own launch behind it
    {"clspre_max_b", 0, 1 << 30, false},  // batches up to it classify and run the hoisted feature chains in one launch (k_classify_pre)
};
static int* opt_field(gnnb_t* h, int i, bool** bf) {
  *bf = nullptr;
  switch (i) {
    case 0: *bf = &h->bf3; return nullptr;
    case 1: return &h->fuse;
    case 2: *bf = &h->use_top; return nullptr;
    case 3: *bf = &h->use_gather; return nullptr;
    case 4: *bf = &h->embed_fuse; return nullptr;
    case 5: *bf = &h->dense_lds; return nullptr;
    case 6: return &h->tail_max_b;
    case 7: return &h->top_split_max;
    case 8: *bf = &h->top_fuse_upd; return nullptr;
    default: return &h->clspre_max_b;
  }
}
static int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < (int)(sizeof kOptions / sizeof kOptions[0]); ++i)
    if (!strcmp(name, kOptions[i].name)) return i;
  return -1;
}
extern "C" int gnnb_option_count(void) { return (int)(sizeof kOptions / sizeof kOptions[0]); }
extern "C" const char* gnnb_option_name(int i) { return i >= 0 && i < gnnb_option_count() ? kOptions[i].name : ""; }
extern "C" int gnnb_set_option(gnnb_t* h, const char* name, int value) {
  if (!h) return fail(GNNB_E_INVALID, "gnnb_set_option: null handle");
  const int i = opt_index(name);
  if (i < 0) return fail(GNNB_E_INVALID, "gnnb_set_option: unknown option '%s'", name ? name : "(null)");
  const OptDesc& d = kOptions[i];
  if (value < d.lo || value > d.hi) return fail(GNNB_E_INVALID, "gnnb_set_option: %s = %d outside [%d, %d]", d.name, value, d.lo, d.hi);
  if (d.before_bind && h->bound) return fail(GNNB_E_STATE, "gnnb_set_option: %s must be set before gnnb_bind_network (it shapes the tables built there)", d.name);
  bool* bf = nullptr;
  int* f = opt_field(h, i, &bf);
  if (bf) *bf = value != 0;
  else *f = (i == 7) ? (value >= 4 ? 4 : (value >= 2 ? 2 : 1)) : value;
  return GNNB_OK;
}
extern "C" int gnnb_get_option(const gnnb_t* h, const char* name, int* value) {
  if (!h || !value) return fail(GNNB_E_INVALID, "gnnb_get_option: null argument");
  const int i = opt_index(name);
  if (i < 0) return fail(GNNB_E_INVALID, "gnnb_get_option: unknown option '%s'", name ? name : "(null)");
  bool* bf = nullptr;
  int* f = opt_field(const_cast<gnnb_t*>(h), i, &bf);
  *value = bf ? (*bf ? 1 : 0) : *f;
  return GNNB_OK;
}
#ifdef GNNB_DEV
// development builds (-DGNNB_DEV: tools/ablate) keep environment overrides for A/B runs of paths the shipped library does not expose
static void dev_env_overrides(gnnb_t* h) {
  if (const char* e = getenv("GNNB_ZERO_DEAD")) h->zero_dead = e[0] == '1';
  if (const char* e = getenv("GNNB_S_IN_GATHER")) h->s_in_gather = e[0] != '0';
  if (const char* e = getenv("GNNB_GIU_OCC")) h->giu_occ = atoi(e) < 1 ? 1 : atoi(e);
  if (const char* e = getenv("GNNB_GATHER_OCC")) h->gather_occ = atoi(e) < 1 ? 1 : atoi(e);
  if (const char* e = getenv("GNNB_NO_GATHER16")) h->gather16 = !(e[0] == '1');
  if (const char* e = getenv("GNNB_SPARSE")) h->gather_sparse = atoi(e);
  if (const char* e = getenv("GNNB_NO_RESTRICT")) h->restrict_last = !(e[0] == '1');
  if (const char* e = getenv("GNNB_NO_SCORED_GATHER")) h->scored_gather = !(e[0] == '1');
  if (const char* e = getenv("GNNB_PER_SAMPLE_MIN_B")) h->per_sample_min_b = atoi(e);
  if (const char* e = getenv("GNNB_GATHER_BF3")) h->gather_bf3 = e[0] == '1';
  for (int i = 0; i < gnnb_option_count(); ++i) {      // GNNB_OPT_<NAME>=<int> for every option of the table
    std::string k = std::string("GNNB_OPT_") + kOptions[i].name;
    for (auto& c : k) c = (char)toupper((unsigned char)c);
    if (const char* e = getenv(k.c_str())) (void)gnnb_set_option(h, kOptions[i].name, atoi(e));
  }
}
#endif

extern "C" int gnnb_create(gnnb_t** out, const float* w_blob, size_t n_floats, int T, int p) {
  if (!out || !w_blob) return fail(GNNB_E_INVALID, "gnnb_create: null argument");
  if (p != P) return fail(GNNB_E_INVALID, "gnnb_create: embedding size %d unsupported (kernels are built for p=64)", p);
  if (T < 1 || T > 16) return fail(GNNB_E_INVALID, "gnnb_create: T=%d out of range", T);
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_create: weight blob has %zu floats, expected %zu", n_floats, blob_floats());
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(GNNB_E_HIP, "gnnb_create: no HIP device (%s)", hipGetErrorString(e));
  gnnb_t* h = new gnnb_handle();
  h->T = T;
  h->p = p;
  HIPCHK(hipGetDevice(&h->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->device));
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (int rc = load_weights(h, w_blob, nullptr)) return rc;
  HIPCHK(hipMalloc((void**)&h->d_zero, 256 * sizeof(float)));
  HIPCHK(hipMemset(h->d_zero, 0, 256 * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->d_ctl, 8 * 64 * sizeof(int)));
  HIPCHK(hipMemset(h->d_ctl, 0, 8 * 64 * sizeof(int)));
  // > 64 KiB of dynamic LDS needs the attribute
  HIPCHK(hipFuncSetAttribute((const void*)k_pre<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackPreFwd::FLOATS + PackPreBwd::FLOATS) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_pre<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PackPreBwdL3::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_pre_inp, hipFuncAttributeMaxDynamicSharedMemorySize, PackPreInp::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_input_update, hipFuncAttributeMaxDynamicSharedMemorySize, PackUpdInp::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_score, hipFuncAttributeMaxDynamicSharedMemorySize, PackScore::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_livesum, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef GNNB_DEV
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));

#define FUSEDQ_ATTR(L, S, P) HIPCHK(hipFuncSetAttribute((const void*)k_gather_update_q<L, S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  FUSEDQ_ATTR(16, 0, false); FUSEDQ_ATTR(16, 1, false); FUSEDQ_ATTR(16, 2, false); FUSEDQ_ATTR(32, 1, false); FUSEDQ_ATTR(32, 1, true);
#undef FUSEDQ_ATTR
  HIPCHK(hipFuncSetAttribute((const void*)k_classify_pre, hipFuncAttributeMaxDynamicSharedMemorySize, CLSPRE_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void*)k_scored_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));      // (the kernel also has a few static words)
  HIPCHK(hipFuncSetAttribute((const void*)k_top<4>, hipFuncAttributeMaxDynamicSharedMemorySize, TOP_LDS_FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_top<2>, hipFuncAttributeMaxDynamicSharedMemorySize, TOP_LDS_FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_top<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TOP_LDS_FLOATS * 4));
#ifdef GNNB_DEV
  dev_env_overrides(h);
#endif
  *out = h;
  return GNNB_OK;
}

static void free_trainer(gnnb_t* h) {
  gnnb_train::Trainer* t = h->trainer;
  if (!t) return;
  for (float* p : {t->d_w, t->d_g, t->d_m, t->d_v, t->d_scores, t->d_ds, t->d_loss, t->d_imp})
    if (p) (void)hipFree(p);
  if (t->d_kw) (void)hipFree(t->d_kw);
  if (t->d_sel) (void)hipFree(t->d_sel);
  for (float* p : t->edge_w)
    if (p) (void)hipFree(p);
  t->arena.release();
  if (t->desc) (void)hipHostFree(t->desc);
  delete t;
  h->trainer = nullptr;
}

static void free_network(gnnb_t* h) {
  if (h->trainer) {                       // the edge weights of the trainer belong to the network that goes away
    for (float* p : h->trainer->edge_w)
      if (p) (void)hipFree(p);
    h->trainer->edge_w.clear();
  }
  for (auto& d : h->dev) {
    if (d.w_fwd) (void)hipFree(d.w_fwd);
    if (d.w_bwd) (void)hipFree(d.w_bwd);
    if (d.bias) (void)hipFree(d.bias);
  }
  for (auto* v : {&h->gf, &h->gb})
    for (auto& d : *v) {
      if (d.cmat) (void)hipFree(d.cmat);
      if (d.taps3) (void)hipFree(d.taps3);
      if (d.koff) (void)hipFree(d.koff);
      if (d.ttab) (void)hipFree(d.ttab);
    }
  h->gf.clear();
  h->gb.clear();
  h->dev.clear();
  h->edges.clear();
  h->N.clear();
  h->relu_q.clear();
  h->hw.clear();
  h->bound = false;
}

extern "C" int gnnb_destroy(gnnb_t* h) {
  if (!h) return GNNB_OK;
  free_network(h);
  if (h->d_pack[0]) (void)hipFree(h->d_pack[0]);      // one block (load_weights)
  if (h->d_zero) (void)hipFree(h->d_zero);
  if (h->d_ctl) (void)hipFree(h->d_ctl);
  if (h->pack_stage) (void)hipHostFree(h->pack_stage);
  if (h->d_s1) (void)hipFree(h->d_s1);
  if (h->hs_pinned) (void)hipHostFree(h->hs_pinned);
  if (h->hs_out_pinned) (void)hipHostFree(h->hs_out_pinned);
  if (h->hs_dev) (void)hipFree(h->hs_dev);
  if (h->hs_ws) (void)hipFree(h->hs_ws);
  if (h->hs_scores) (void)hipFree(h->hs_scores);
  if (h->hs_dec) (void)hipFree(h->hs_dec);
  free_trainer(h);
  if (h->work_pool && h->work_pool->owner == getpid()) delete h->work_pool;
  for (auto& ev : h->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto& ev : h->pool) (void)hipEventDestroy(ev);
  delete h;
  return GNNB_OK;
}

static bool conv_channels_ok(int c) { return c == 3 || c == 8 || c == 16 || c == 32; }

extern "C" int gnnb_bind_network(gnnb_t* h, const gnnb_layer_desc* L, int n, int c0, int h0, int w0) {
  if (!h || !L || n < 2) return fail(GNNB_E_INVALID, "gnnb_bind_network: bad arguments");
  free_network(h);
  int C = c0, H = h0, W = w0;
  bool flat = false;
  int nflat = c0 * h0 * w0;
  h->N.push_back(nflat);
  h->edges.emplace_back();
  h->relu_q.push_back(-1);
  h->hw.push_back(1);
  Edge pend;
  bool have = false;
  int pend_hw = 1;
  for (int q = 0; q < n; ++q) {
    const gnnb_layer_desc& d = L[q];
    if (d.kind == GNNB_CONV) {
      if (flat) return fail(GNNB_E_INVALID, "layer %d: conv after flatten", q);
      if (have) return fail(GNNB_E_INVALID, "layer %d: two linear maps without a ReLU between them", q);
      if (d.c_in != C) return fail(GNNB_E_INVALID, "layer %d: conv expects %d input channels, graph has %d", q, d.c_in, C);
      if (!d.weight || !d.bias) return fail(GNNB_E_INVALID, "layer %d: null weight/bias", q);
      Edge e;
      e.kind = 0;
      e.c_in = C; e.h_in = H; e.w_in = W; e.c_out = d.c_out; e.kh = d.kh; e.kw = d.kw; e.stride = d.stride; e.pad = d.pad;
      if (d.stride < 1 || d.kh < 1 || d.kw < 1) return fail(GNNB_E_INVALID, "layer %d: bad conv geometry", q);
      if ((H + 2 * d.pad - d.kh) % d.stride || (W + 2 * d.pad - d.kw) % d.stride)
        return fail(GNNB_E_INVALID, "layer %d: conv geometry leaves a remainder (conv_transpose2d of the reference would need output_padding)", q);
      e.h_out = (H + 2 * d.pad - d.kh) / d.stride + 1;
      e.w_out = (W + 2 * d.pad - d.kw) / d.stride + 1;
      e.n_in = C * H * W;
      e.n_out = e.c_out * e.h_out * e.w_out;
      e.w.assign(d.weight, d.weight + (size_t)e.c_out * e.c_in * e.kh * e.kw);
      e.b.assign(d.bias, d.bias + e.c_out);
      C = e.c_out; H = e.h_out; W = e.w_out;
      nflat = e.n_out;
      pend_hw = H * W;
      pend = e;
      have = true;
    } else if (d.kind == GNNB_LINEAR) {
      if (have) return fail(GNNB_E_INVALID, "layer %d: two linear maps without a ReLU between them", q);
      if (d.n_in != nflat) return fail(GNNB_E_INVALID, "layer %d: linear expects %d inputs, graph has %d", q, d.n_in, nflat);
      if (!d.weight || !d.bias) return fail(GNNB_E_INVALID, "layer %d: null weight/bias", q);
      Edge e;
      e.kind = 1;
      e.n_in = d.n_in; e.n_out = d.n_out;
      e.c_in = e.h_in = e.w_in = e.c_out = e.h_out = e.w_out = e.kh = e.kw = e.stride = e.pad = 0;
      e.w.assign(d.weight, d.weight + (size_t)d.n_out * d.n_in);
      e.b.assign(d.bias, d.bias + d.n_out);
      nflat = d.n_out;
      flat = true;
      pend_hw = 1;
      pend = e;
      have = true;
    } else if (d.kind == GNNB_RELU) {
      if (!have) return fail(GNNB_E_INVALID, "layer %d: ReLU without a preceding conv/linear", q);
      h->N.push_back(nflat);
      h->edges.push_back(pend);
      h->relu_q.push_back(q);
      h->hw.push_back(pend_hw);
      have = false;
    } else if (d.kind == GNNB_FLATTEN) {
      flat = true;
    } else {
      return fail(GNNB_E_INVALID, "layer %d: unknown kind %d", q, d.kind);
    }
  }
  if (have) return fail(GNNB_E_INVALID, "fixed layers must end after a ReLU (the property layer is passed per batch)");
  const int Lr = (int)h->N.size() - 1;
  if (Lr < 1 || Lr > MAXL) return fail(GNNB_E_INVALID, "unsupported number of ReLU layers %d (max %d)", Lr, MAXL);
  h->N.push_back(1);   // property node
  h->n_fixed = n;
  h->R = 0;
  for (int k = 1; k <= Lr; ++k) h->R += h->N[k];
  h->dev.resize(Lr + 1);
  for (int k = 0; k < MAXL + 2; ++k) h->last_proj[k] = -1;
  for (int k = 0; k <= Lr; ++k)
    if (h->N[k] > LIVESUM_MAXSRC) return fail(GNNB_E_INVALID, "graph layer %d has %d nodes, more than the %d k_livesum holds in LDS", k, h->N[k], LIVESUM_MAXSRC);
  for (int k = 1; k <= Lr; ++k) {
    const Edge& e = h->edges[k];
    DevEdge& d = h->dev[k];
    if (int rc = upload(&d.bias, e.b.data(), e.b.size())) return rc;
    if (k == 1) {                       // row sums of edge 1 (for a conv: of the taps inside the image) -- k_livesum's job for this edge
      std::vector<float> s1(h->N[1], 0.0f);
      if (e.kind == 0) {
        for (int co = 0; co < e.c_out; ++co)
          for (int oy = 0; oy < e.h_out; ++oy)
            for (int ox = 0; ox < e.w_out; ++ox) {
              float acc = 0.0f;
              for (int ci = 0; ci < e.c_in; ++ci)
                for (int ky = 0; ky < e.kh; ++ky) {
                  const int iy = oy * e.stride - e.pad + ky;
                  if (iy < 0 || iy >= e.h_in) continue;
                  for (int kx = 0; kx < e.kw; ++kx) {
                    const int ix = ox * e.stride - e.pad + kx;
                    if (ix < 0 || ix >= e.w_in) continue;
                    acc += e.w[(((size_t)co * e.c_in + ci) * e.kh + ky) * e.kw + kx];
                  }
                }
              s1[((size_t)co * e.h_out + oy) * e.w_out + ox] = acc;
            }
      } else {
        for (int i = 0; i < e.n_out; ++i) {
          float acc = 0.0f;
          for (int q = 0; q < e.n_in; ++q) acc += e.w[(size_t)i * e.n_in + q];
          s1[i] = acc;
        }
      }
      if (h->d_s1) { (void)hipFree(h->d_s1); h->d_s1 = nullptr; }
      if (int rc = upload(&h->d_s1, s1.data(), s1.size())) return rc;
    }
    if (e.kind == 0) {
      std::vector<float> t(e.w.size());
      pack_conv_fwd(t.data(), e);
      if (int rc = upload(&d.w_fwd, t.data(), t.size())) return rc;
      pack_conv_bwd(t.data(), e);
      if (int rc = upload(&d.w_bwd, t.data(), t.size())) return rc;
    } else {
      // k_dense_agg operands At[k][i] = A[i][k], zero-padded to 32*MT columns and 8*ksq rows (ksq = k-steps per wave)
      auto ksq_of = [](int K) { return (((K + 1) / 2 + 3) / 4 + DENSE_CH - 1) / DENSE_CH * DENSE_CH; };
      d.mt_fwd = (e.n_out + 31) / 32;
      d.ld_fwd = d.mt_fwd * 32;
      d.ksq_fwd = ksq_of(e.n_in);
      d.kpad_fwd = (e.n_in + 63) / 64 * 64;
      d.kpad_bwd = (e.n_out + 15) / 16 * 16;
      const size_t rows_f = std::max<size_t>(8 * d.ksq_fwd + 2 * DENSE_CH, d.kpad_fwd + 32);
      std::vector<float> t(rows_f * d.ld_fwd, 0.f);           // forward: A = W, k = input node
      for (int o = 0; o < e.n_out; ++o)
        for (int i = 0; i < e.n_in; ++i) t[(size_t)i * d.ld_fwd + o] = e.w[(size_t)o * e.n_in + i];
      if (int rc = upload(&d.w_fwd, t.data(), t.size())) return rc;
      d.mt_bwd = (e.n_in + 31) / 32;
      d.ld_bwd = d.mt_bwd * 32;
      d.ksq_bwd = ksq_of(e.n_out);
      const size_t rows_b = std::max<size_t>(8 * d.ksq_bwd + 2 * DENSE_CH, d.kpad_bwd + 64);   // k_dense_bwd_lds reads up to 3 chunks past kpad
      t.assign(rows_b * d.ld_bwd, 0.f);                        // transposed: A = W^T, k = output node
      for (int o = 0; o < e.n_out; ++o)
        for (int i = 0; i < e.n_in; ++i) t[(size_t)o * d.ld_bwd + i] = e.w[(size_t)o * e.n_in + i];
      if (int rc = upload(&d.w_bwd, t.data(), t.size())) return rc;
    }
  }
  h->top_ok = Lr >= 2 && h->edges[Lr].kind == 1 && h->N[Lr] <= 128 && h->dense_lds && h->dev[Lr].mt_fwd <= 4 && h->dev[Lr].kpad_bwd <= 128;
  // MFMA gather tables for every conv edge, both directions (the input layer's transposed edge is not normalised)
  h->gf.assign(Lr + 1, DevGather());
  h->gb.assign(Lr + 1, DevGather());
  if (h->use_gather)
    for (int k = 1; k <= Lr; ++k) {
      if (h->edges[k].kind != 0) continue;
      for (int dir = 0; dir < 2; ++dir) {
        GatherHost gh;
        // the input layer's transposed gather is fused with its feature chain and update (132 MFMAs per tile)
        if (!build_gather(h->edges[k], dir, dir == 1 && k > 1, gh, (dir == 1 && k == 1) ? 132 : 0, h->gather16)) continue;
        DevGather& d = dir == 0 ? h->gf[k] : h->gb[k];
        d.g = gh.g;
        if (int rc = upload(&d.cmat, gh.cmat.data(), gh.cmat.size())) return rc;
        if (!gh.taps3.empty()) {
          HIPCHK(hipMalloc((void**)&d.taps3, gh.taps3.size() * sizeof(uint32_t)));
          HIPCHK(hipMemcpy(d.taps3, gh.taps3.data(), gh.taps3.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        HIPCHK(hipMalloc((void**)&d.koff, gh.koff.size() * sizeof(int)));
        HIPCHK(hipMemcpy(d.koff, gh.koff.data(), gh.koff.size() * sizeof(int), hipMemcpyHostToDevice));
        {
          const TileMap& tm = gh.g.tm;
          if (tm.NCG > 255 || tm.NBY > 4095 || tm.NBX > 4095) return fail(GNNB_E_INVALID, "layer %d: tile table overflow", k);
          std::vector<int> tt(tm.TPS);
          for (int t = 0; t < tm.TPS; ++t) {
            const int cg = t / (tm.NBY * tm.NBX), rem = t % (tm.NBY * tm.NBX);
            tt[t] = cg | ((rem / tm.NBX) << 8) | ((rem % tm.NBX) << 20);
          }
          HIPCHK(hipMalloc((void**)&d.ttab, tt.size() * sizeof(int)));
          HIPCHK(hipMemcpy(d.ttab, tt.data(), tt.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        d.ok = true;
      }
    }
  // an edge without MFMA gather tables falls back to the VALU gathers, which are compiled for a few channel counts only
  for (int k = 1; k <= Lr; ++k) {
    const Edge& e = h->edges[k];
    if (e.kind != 0) continue;
    if ((!h->gf[k].ok && !conv_channels_ok(e.c_out)) || (!h->gb[k].ok && !conv_channels_ok(e.c_in)))
      return fail(GNNB_E_INVALID, "conv edge %d (%d -> %d channels): no MFMA gather tables and the fallback kernels only cover channel counts "
                  "{3, 8, 16, 32}", k, e.c_in, e.c_out);
  }
  h->bound = true;
  return GNNB_OK;
}

// tile map of the input layer's update: the tiles of the transposed gather of edge 1 when it exists, else flat
static TileMap flat_map(int N) { TileMap t; t.mode = 0; t.N = N; return t; }
static TileMap bwd_map(const gnnb_t* h, int k) {
  const int L = (int)h->N.size() - 2;
  return (k + 1 <= L && h->gb[k + 1].ok) ? h->gb[k + 1].g.tm : flat_map(h->N[k]);
}
static long map_tiles(const TileMap& t, int B) { return t.mode ? (long)B * t.TPS : ((long)B * t.N + 31) / 32; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static DTileMap to_dtm(const TileMap& t) {
  return DTileMap{t.mode, t.N, t.C, t.H, t.W, t.CT, t.PY, t.PX, t.ay, t.ax, t.NBY, t.NBX, t.NCG, t.TPS, ilog2(t.PY), ilog2(t.PX),
                  t.TPS > 1 ? (unsigned)((1ull << 32) / (unsigned)t.TPS) + 1u : 0u};      // tile_sample (gnnb_dev.h); TPS <= 1 is handled there
}
static DGather to_dg(const DevGather& d, const float* zero) {
  const GatherGeom& g = d.g;
  return DGather{d.cmat, reinterpret_cast<const int2*>(d.koff), d.ttab, zero, g.K2, g.tm.NCG * g.K2, g.Hs, g.Ws, g.Ns, g.ystep, g.ybase,
                 g.xstep, g.xbase, g.WY, g.WX, g.normalise, g.kh, g.kw, g.stride, g.pad, g.lanes, reinterpret_cast<const uint2*>(d.taps3)};
}
static size_t gather_lds_bytes(const DevGather& d, size_t pack_floats) {
  return (pack_floats + (size_t)d.g.tm.NCG * d.g.K2 * 64) * 4 + (size_t)gather_slots(d.g.K2, d.g.lanes) * 12 + (size_t)((d.g.tm.TPS + 3) & ~3) * 4;
}

// per-wave live-slot tables of the sparse gathers (behind the shared tables, 8-byte aligned)
static size_t sparse_tab_bytes(const DevGather& d) {
  return 8 + (size_t)WAVES_MLP * ((d.g.lanes == 16 ? 4 : 2) * d.g.K2 + 32) * 8;
}

#if defined(FUSED_TIMING) && FUSED_TIMING == 6
// dev: start / end (100 MHz chip-wide clock) of every wave of the instrumented k_gather_update_q launch: 2 x 16 x gridDim words
extern "C" int gnnb_debug_wall(unsigned long long* out, int nwords) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_wall), (size_t)nwords * 8) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FUSED_TIMING
// dev: cycle sums of k_gather_update's phases over every wave since the last reset (index 15: number of waves)
extern "C" int gnnb_debug_read(unsigned long long* out, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_t), sizeof z) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fused_t), z, sizeof z) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" int gnnb_graph_info(const gnnb_t* h, int* n_graph, int* sizes, int* n_relu_total) {
  if (!h || !h->bound) return fail(GNNB_E_STATE, "gnnb_graph_info: no network bound");
  if (n_graph) *n_graph = (int)h->N.size();
  if (sizes)
    for (size_t k = 0; k < h->N.size(); ++k) sizes[k] = h->N[k];
  if (n_relu_total) *n_relu_total = h->R;
  return GNNB_OK;
}

// k_gather_update_q over conv edge `d`: dynamic LDS bytes (weights, row queue, gather tables, per-gather-wave slot tables), and
// whether the edge can take it at all (a sparse walk behind a ReLU layer, 16-node tiles without POST, everything in 160 KB)
static size_t fusedq_lds_bytes(const DevGather& d, bool sparse, bool post, int qtiles) {
  const size_t tables = (size_t)d.g.tm.NCG * d.g.K2 * 64 * 4 + (size_t)gather_slots(d.g.K2, d.g.lanes) * 8 + (size_t)((d.g.tm.TPS + 3) & ~3) * 4 +
                        (size_t)((gather_slots(d.g.K2, d.g.lanes) + 3) & ~3) * 4;
  return (size_t)(PackUpdL3::FLOATS + (post ? 6144 : 0) + fusedq_queue_floats(qtiles)) * 4 + tables +
         (sparse ? (size_t)QG_WAVES * ((d.g.lanes == 16 ? 4 : 2) * d.g.K2 + 32) * 8 : 0);
}
// ring slots that fit (0: the kernel does not fit at all)
static int fusedq_qtiles(const DevGather& d, bool sparse, bool post) {
  for (int q = QTILES; q >= 2; --q)
    if (fusedq_lds_bytes(d, sparse, post, q) <= 160 * 1024) return q;
  return 0;
}
static bool fusedq_ok(const gnnb_t* h, const DevGather& d, int src_layer, bool embed_src, bool post) {
  if (h->fuse == 0 || !h->bf3 || !d.ok) return false;
  const bool sparse = (h->gather_sparse & (d.g.lanes == 16 ? 1 : 2)) && !embed_src && src_layer >= 1;
  if (src_layer >= 1 && !sparse) return false;
  if (d.g.lanes == 32 && !sparse) return false;
  if (d.g.lanes == 16 && post) return false;
  return fusedq_qtiles(d, sparse, post) > 0;
}

// JSON description of the launch plan of one forward (per B=1): which kernel updates which layer, tile shapes and
// MFMA counts.  bench.py derives the algorithmic flops per kernel class from it; DESIGN.md quotes it.
extern "C" int gnnb_describe(const gnnb_t* h, char* buf, size_t cap) {
  if (!h || !h->bound || !buf || cap < 64) return fail(GNNB_E_INVALID, "gnnb_describe: bad arguments");
  const int L = (int)h->N.size() - 2;
  std::string o = "{\"T\": " + std::to_string(h->T) + ", \"bf3\": " + std::to_string(h->bf3 ? 1 : 0) + ", \"embed_fused\": " + std::to_string(h->embed_fuse && h->gf.size() > 1 && h->gf[1].ok ? 1 : 0) + ", \"sizes\": [";
  for (size_t k = 0; k < h->N.size(); ++k) o += (k ? ", " : "") + std::to_string(h->N[k]);
  o += "], \"updates\": [";
  auto nnz = [&](int e) -> long {     // edges of the layer graph between layer e-1 and e (no-padding upper bound)
    if (e > L) return h->N[L];
    const Edge& ed = h->edges[e];
    return ed.kind == 0 ? (long)ed.c_out * ed.h_out * ed.w_out * ed.c_in * ed.kh * ed.kw : (long)ed.n_in * ed.n_out;
  };
  auto item = [&](const char* what, int k, const DevGather* d, const char* fallback, int n_src) {
    char t[640];
    const long ez = nnz(what[0] == 'f' ? k : k + 1);
    if (d && d->ok) {
      const GatherGeom& g = d->g;
      snprintf(t, sizeof t,
               "{\"update\": \"%s\", \"layer\": %d, \"kernel\": \"%s\", \"nodes\": %d, \"tiles_per_sample\": %d, \"tile_nodes\": %d, "
               "\"tile\": [%d, %d, %d], \"align\": [%d, %d], \"window\": [%d, %d], \"gather_ksteps\": %d, \"n_src\": %d, \"edge_nnz\": %ld}",
               what, k, k == 0 ? "k_gather_input_update" : (fusedq_ok(h, *d, what[0] == 'f' ? k - 1 : k + 1, false, false) ? "k_gather_update" : "k_gather+k_node_update"),
               h->N[k], g.tm.TPS, g.lanes, g.tm.CT, g.tm.PY, g.tm.PX,
               g.tm.ay, g.tm.ax, g.WY, g.WX, g.K2, n_src, ez);
    } else {
      snprintf(t, sizeof t, "{\"update\": \"%s\", \"layer\": %d, \"kernel\": \"%s\", \"nodes\": %d, \"n_src\": %d, \"edge_nnz\": %ld}",
               what, k, fallback, h->N[k], n_src, ez);
    }
    o += t;
  };
  const bool top = h->use_top && h->bf3 && h->top_ok;     // k_top covers the edge into layer L, both updates of layer L and the edge back
  bool first = true;
  for (int k = 1; k <= L; ++k) {
    if (!first) o += ", ";
    first = false;
    if (top && k == L) item("fwd", k, nullptr, "k_top+k_top", h->N[k - 1]);
    else item("fwd", k, &h->gf[k], h->edges[k].kind == 0 ? "k_conv_fwd+k_node_update" : "k_dense_agg+k_node_update", h->N[k - 1]);
  }
  for (int k = L; k >= 1; --k) {
    o += ", ";
    if (k == L) item("bwd", k, nullptr, top ? "k_top+k_top" : "k_prop+k_node_update", 1);
    else if (top && k == L - 1)      // (the update of layer L-1 rides k_top's transposed edge when its live-row list is kept: gnnb_forward `top_upd`)
      item("bwd", k, nullptr, h->top_fuse_upd && L >= 3 && TOP_LIST_KEEP_OK(h->edges[L].n_in) ? "k_top+k_top" : "k_top+k_node_update", h->N[k + 1]);
    else item("bwd", k, &h->gb[k + 1], h->edges[k + 1].kind == 0 ? "k_convT_bwd+k_node_update" : "k_dense_agg+k_node_update", h->N[k + 1]);
  }
  o += ", ";
  item("input", 0, &h->gb[1], h->edges[1].kind == 0 ? "k_convT_bwd+k_input_update" : "k_dense_agg+k_input_update", h->N[1]);
  o += "]}";
  if (o.size() + 1 > cap) return fail(GNNB_E_NOMEM, "gnnb_describe: buffer too small (%zu needed)", o.size() + 1);
  memcpy(buf, o.c_str(), o.size() + 1);
  return GNNB_OK;
}

// ---- workspace layout (float offsets, every region 256-B aligned) ----
struct WsLayout {                // plain arrays: gnnb_forward computes it on its stack (no allocation in the call)
  size_t mu[MAXL + 2], Pf[MAXL + 2], Pb[MAXL + 2], live[MAXL + 2], amb[MAXL + 2], score[MAXL + 2];
  size_t lf[MAXL + 2];          // live flags (B, N_k) as floats
  size_t sf[MAXL + 2], sb[MAXL + 2];   // k_livesum outputs: sf[k] (B, N_k) over edge k, sb[k] (B, N_k) over edge k+1 transposed
  size_t F1 = 0;                // rows of layer 1 after the producer-side map of the input update (PackPostInp)
  size_t F3 = 0;                // the same as three bf16 pieces (rows3): what the input update's bf16 x 3 aggregate reads
  size_t cnt = 0, best = 0, nb = 0, Q = 0, total = 0;     // best: B 64-bit decision keys + the finished-workgroup counter of k_score
  size_t topflag = 0, topx = 0;                           // k_top's workgroup split: arrival counters, (B, 8, 64) exchange buffer
};
#define TOP_SPLIT_MAXB 128      // k_top only splits a sample over workgroups while B x S workgroups fit the chip: B <= n_cu / 2
static size_t align64(size_t nfloats) { return (nfloats + 63) & ~(size_t)63; }
static WsLayout ws_layout(const gnnb_t* h, int B) {
  WsLayout w;
  for (int k = 0; k < MAXL + 2; ++k) w.mu[k] = w.Pf[k] = w.Pb[k] = w.live[k] = w.amb[k] = w.score[k] = w.lf[k] = w.sf[k] = w.sb[k] = 0;
  const int K = (int)h->N.size() - 1;
  size_t off = 0;
  w.cnt = off; off += 64;                      // (unused: the list counters live in the handle's control blocks)
  w.best = off; off += align64((size_t)2 * B + 2);
  w.topflag = off; off += align64(TOP_SPLIT_MAXB);
  w.topx = off; off += align64((size_t)std::min(B, TOP_SPLIT_MAXB) * 512);
  for (int k = 0; k <= K; ++k) { w.mu[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  size_t maxn = 0;
  for (int k = 0; k < K; ++k) maxn = std::max(maxn, (size_t)h->N[k]);
  w.nb = off; off += align64((size_t)B * maxn * 64);
  for (int k = 1; k < K; ++k) { w.Pf[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  for (int k = 1; k < K; ++k) { w.Pb[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  for (int k = 1; k < K; ++k) {
    w.live[k] = off; off += align64((size_t)B * h->N[k]);
    w.amb[k] = off; off += align64((size_t)B * h->N[k]);
    w.score[k] = off; off += align64((size_t)B * h->N[k]);
  }
  for (int k = 1; k < K; ++k) { w.lf[k] = off; off += align64((size_t)B * h->N[k]); }
  for (int k = 1; k < K; ++k) { w.sf[k] = off; off += align64((size_t)B * h->N[k]); }
  for (int k = 0; k < K - 1; ++k) { w.sb[k] = off; off += align64((size_t)B * h->N[k]); }
  w.F1 = off; off += align64((size_t)B * h->N[1] * 64);
  w.F3 = off; off += h->gather_bf3 ? align64((size_t)B * h->N[1] * ROW3_FLOATS) : 0;      // rows3 of layer 1: only the opt-in GNNB_GATHER_BF3=1 path writes / reads them (base B=256: 400 MB)
  w.Q = off; off += (size_t)map_tiles(bwd_map(h, 0), B) * 2048;
  w.total = off;
  return w;
}

extern "C" size_t gnnb_workspace_bytes(const gnnb_t* h, int B) {
  if (!h || !h->bound || B < 1) return 0;
  return ws_layout(h, B).total * sizeof(float);
}

extern "C" int gnnb_mu_location(const gnnb_t* h, int B, int k, size_t* offset_bytes, size_t* n_floats) {
  if (!h || !h->bound) return fail(GNNB_E_STATE, "gnnb_mu_location: no network bound");
  if (k < 0 || k >= (int)h->N.size() || B < 1) return fail(GNNB_E_INVALID, "gnnb_mu_location: bad layer/batch");
  WsLayout w = ws_layout(h, B);
  if (offset_bytes) *offset_bytes = w.mu[k] * sizeof(float);
  if (n_floats) *n_floats = (size_t)B * h->N[k] * 64;
  return GNNB_OK;
}

// Inspection: the rows of mu[k] written by the last forward are E with mu = W.E + b for the Linear `*linear_id`
// (index into the checkpoint's 26 Linear layers in state-dict order), or final embeddings when *linear_id = -1.
extern "C" int gnnb_mu_projection(const gnnb_t* h, int k, int* linear_id) {
  if (!h || !linear_id) return fail(GNNB_E_INVALID, "gnnb_mu_projection: null argument");
  if (k < 0 || k >= (int)h->N.size()) return fail(GNNB_E_INVALID, "gnnb_mu_projection: bad layer");
  *linear_id = k < MAXL + 2 ? h->last_proj[k] : -1;
  return GNNB_OK;
}

extern "C" int gnnb_set_halfpass_limit(gnnb_t* h, int n) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  h->halfpass_limit = n;
  return GNNB_OK;
}

// Inspection: occupy `n_workgroups` CUs (one workgroup each when lds_bytes > 80 KiB) for `ms` milliseconds (<= 500) on `stream` with a kernel that
// only spins on the clock -- the stand-in for "something else holds CUs" (an RCCL kernel, a second batch) in tests/test_gpu_dist_safety.py.
extern "C" int gnnb_debug_occupy(int n_workgroups, int threads, size_t lds_bytes, double ms, void* stream) {
  if (n_workgroups < 1 || n_workgroups > 4096 || threads < 64 || threads > 1024 || lds_bytes > 160 * 1024 || !(ms > 0.0) || ms > 500.0)
    return fail(GNNB_E_INVALID, "gnnb_debug_occupy: bad arguments");
  static bool attr = false;
  if (!attr) { HIPCHK(hipFuncSetAttribute((const void*)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
  hipLaunchKernelGGL(k_occupy, dim3((unsigned)n_workgroups), dim3((unsigned)threads), lds_bytes, (hipStream_t)stream, (unsigned long long)(ms * 1e5));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "launch of k_occupy failed: %s", hipGetErrorString(e));
  return GNNB_OK;
}

// ---- profiling ----
extern "C" int gnnb_profile_enable(gnnb_t* h, int on) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  h->prof = on != 0;
  return GNNB_OK;
}
extern "C" int gnnb_profile_classes(void) { return PC_COUNT; }
extern "C" const char* gnnb_profile_class_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
extern "C" int gnnb_profile_read(gnnb_t* h, double* total_ms, int64_t* launches, int n, int reset) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  if (!h->pending.empty()) {
    for (auto& ev : h->pending) {
      HIPCHK(hipEventSynchronize(ev.b));        // launches may sit on several streams (batch pipelining)
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
      h->prof_ms[ev.cls] += ms;
      h->prof_n[ev.cls] += 1;
      if (h->trace.size() < 65536) h->trace.push_back({ev.cls, ms});
      h->pool.push_back(ev.a);
      h->pool.push_back(ev.b);
    }
    h->pending.clear();
  }
  for (int i = 0; i < n && i < PC_COUNT; ++i) {
    if (total_ms) total_ms[i] = h->prof_ms[i];
    if (launches) launches[i] = h->prof_n[i];
  }
  if (reset)
    for (int i = 0; i < PC_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  return GNNB_OK;
}
// The launches gnnb_profile_read has resolved since the last call of this function, in launch order: class and duration of each
// (bench.py prices single launches of a class with it).  Returns their number (at most cap are copied); the list is cleared.
extern "C" int gnnb_profile_trace(gnnb_t* h, int* cls, double* ms, int cap) {
  if (!h) { fail(GNNB_E_INVALID, "null handle"); return -1; }
  const int n = (int)h->trace.size();
  for (int i = 0; i < n && i < cap; ++i) {
    if (cls) cls[i] = h->trace[i].cls;
    if (ms) ms[i] = h->trace[i].ms;
  }
  h->trace.clear();
  return n;
}

struct Launcher {
  gnnb_t* h;
  hipStream_t st;
  int rc = 0;
  hipEvent_t get_event() {
    if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) rc = fail(GNNB_E_HIP, "hipEventCreate failed");
    return e;
  }
  template <class F>
  void run(int cls, F&& f) {
    if (rc) return;
    if (h->prof) {
      gnnb_handle::Ev ev{cls, get_event(), get_event()};
      if (rc) return;
      (void)hipEventRecord(ev.a, st);
      f();
      (void)hipEventRecord(ev.b, st);
      h->pending.push_back(ev);
      h->prof_stream = st;
    } else {
      f();
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(GNNB_E_HIP, "launch of %s failed: %s", kProfNames[cls], hipGetErrorString(e));
  }
};

static int mlp_grid(const gnnb_t* h, long ntiles) {
  long g = (ntiles + WAVES_MLP - 1) / WAVES_MLP;
  if (g > h->n_cu) g = h->n_cu;
  return (int)(g < 1 ? 1 : g);
}

template <int C>
static void launch_conv_fwd(const ConvArgs& a, hipStream_t st) {
  const long waves = (long)a.B * a.H_out * a.W_out;
  hipLaunchKernelGGL(k_conv_fwd<C>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}
template <int C>
static void launch_convT(const ConvArgs& a, hipStream_t st) {
  const long waves = (long)a.B * a.H_in * a.W_in;
  hipLaunchKernelGGL(k_convT_bwd<C>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}

extern "C" int gnnb_forward(gnnb_t* h, const gnnb_batch* in, int B, float* scores, int32_t* decisions, int32_t* status,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !in || !scores || !decisions || !status || !workspace) return fail(GNNB_E_INVALID, "gnnb_forward: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_forward: call gnnb_bind_network first");
  if (B < 1) return fail(GNNB_E_INVALID, "gnnb_forward: B=%d", B);
  const int K = (int)h->N.size() - 1, L = K - 1;
  if (in->n_graph != K + 1 || in->n_relu != L || in->n_primal != h->n_fixed + 1)
    return fail(GNNB_E_INVALID, "gnnb_forward: batch has %d graph layers / %d dual / %d primal tensors, network needs %d / %d / %d",
                in->n_graph, in->n_relu, in->n_primal, K + 1, L, h->n_fixed + 1);
  for (int k = 0; k <= K; ++k)
    if (!in->lb[k] || !in->ub[k]) return fail(GNNB_E_INVALID, "gnnb_forward: null bounds pointer for graph layer %d", k);
  for (int k = 0; k < L; ++k)
    if (!in->dual[k]) return fail(GNNB_E_INVALID, "gnnb_forward: null dual pointer %d", k);
  for (int m = 0; m < in->n_primal; ++m)
    if (!in->primal[m]) return fail(GNNB_E_INVALID, "gnnb_forward: null primal pointer %d", m);
  if (!in->x_lp || !in->prop_w || !in->prop_b || !in->mask) return fail(GNNB_E_INVALID, "gnnb_forward: null input pointer");
  if ((long)B * h->N[0] * 64 >= (1L << 40)) return fail(GNNB_E_INVALID, "gnnb_forward: batch too large");
  const WsLayout w = ws_layout(h, B);
  if (workspace_bytes < w.total * sizeof(float))
    return fail(GNNB_E_NOMEM, "gnnb_forward: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  Launcher lz{h, st};
  auto mu = [&](int k) { return ws + w.mu[k]; };
  float* nb = ws + w.nb;

  unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + w.best);
  int* done_ctr = reinterpret_cast<int*>(ws + w.best + 2 * (size_t)B);
  // the counters' control block of this workspace (see gnnb_handle::d_ctl); a workspace seen for the first time (or after a failed
  // call) gets a freshly zeroed one -- an async memset on the stream, the only time anything but kernels is enqueued
  int slot = -1;
  for (int i = 0; i < 8; ++i)
    if (h->ctl_ws[i] == workspace) slot = i;
  if (slot < 0) {
    slot = 0;
    for (int i = 1; i < 8; ++i)
      if (h->ctl_age[i] < h->ctl_age[slot]) slot = i;
    HIPCHK(hipMemsetAsync(h->d_ctl + 64 * slot, 0, 64 * sizeof(int), st));
    h->ctl_ws[slot] = workspace;
  }
  h->ctl_age[slot] = ++h->ctl_clock;
  int* cnt = h->d_ctl + 64 * slot;
  auto ilist = [&](size_t off) { return reinterpret_cast<int*>(ws + off); };
  int roff[MAXL + 2] = {0};                 // offset of layer k inside the flat ReLU index
  for (int k = 2; k <= L + 1; ++k) roff[k] = roff[k - 1] + h->N[k - 1];
  int proj[MAXL + 2];                       // deferred projection of the rows of mu[k] after the kernels enqueued so far (call-local)
  for (int k = 0; k < MAXL + 2; ++k) proj[k] = -1;

  const int total_halfpasses = 2 * h->T;
  const int limit = h->halfpass_limit > 0 ? std::min(h->halfpass_limit, total_halfpasses) : total_halfpasses;
  const bool debug_full = h->halfpass_limit > 0;   // with a limit set nothing is restricted or skipped as dead
  const bool per_sample = B >= h->per_sample_min_b;   // batch large enough for the one-workgroup-per-sample kernels
  // The rows of layer 1 the input-layer update aggregates went through its 64x64 map on the producer side (PackPostInp).
  // Outside inspection runs nothing else reads the plain rows of that half-pass, so the mapped rows simply take their place
  // in mu[1] (whose dead rows k_classify already zeroed); inspection runs keep both, the mapped ones in F1.
  float* const rows1_for_input = debug_full ? ws + w.F1 : mu(1);

  // the input update's aggregate on the bf16 matrix rate: needs the sparse 32-node walk, the bf16 x 3 blocks and a producer that writes F as
  // three pieces (the POST block of layer 1's backward update); inspection runs keep fp32 rows
  const bool input_rows3 = h->gather_bf3 && h->bf3 && !debug_full && h->gb[1].ok && h->gb[1].taps3 && h->gb[1].g.lanes == 32 && (h->gather_sparse & 4);
  const bool embed_in_gather = h->embed_fuse && !debug_full && h->gf[1].ok;
  const bool top_fused = h->use_top && h->bf3 && h->top_ok && !debug_full && per_sample;      // (k_top only exists on the bf16 x 3 rate)
  // The rows of dead nodes are zero by definition (mu = (.) * live).  Every default consumer of a layer's rows walks only the
  // live ones (sparse gathers, the compacted Linear edges of k_top, the score head), so nothing needs them in memory; they
  // are written (k_classify) only for a layer with a consumer that reads every row: VALU / non-sparse gathers, the
  // per-sample / per-tile dense kernels, k_prop, inspection runs.
  auto reads_live_rows_only = [&](int e, bool transposed) {      // edge e between layers e-1 and e; transposed: reads layer e
    if (e == L && top_fused) return transposed || TOP_LIST_OK(h->edges[L].n_in);
    const DevGather& d = transposed ? h->gb[e] : h->gf[e];
    if (!d.ok) return false;
    if (transposed && e == 1) return (h->gather_sparse & 4) != 0;
    return (h->gather_sparse & (d.g.lanes == 16 ? 1 : 2)) != 0;
  };
  // k_top's Linear edges walk live rows only (when their lists fit, top_sample `compact` / `keep`), so they produce the bias sums
  // of edge L in both directions themselves and k_livesum skips those jobs
  const bool s1_table = h->s_in_gather && L >= 2 && h->d_s1 != nullptr;      // bias sums of edge 1 forward: bind-time table
  const int topK = L >= 1 && h->edges[L].kind == 1 ? h->edges[L].n_in : 0;
  const bool top_s_fwd = top_fused && h->s_in_gather && TOP_LIST_OK(topK);
  const bool top_s_bwd = top_fused && h->s_in_gather && TOP_LIST_KEEP_OK(topK) && limit >= 2;
  // k_top also runs the backward node update of layer L-1 on its transposed edge's row tiles (the aggregate never reaches memory):
  // needs the kept live-row list (B2 then walks live rows only) and a layer L-1 that is not layer 1 (whose update has the
  // restricted / input-mapping forms)
  const bool top_upd = top_fused && h->top_fuse_upd && L >= 3 && TOP_LIST_KEEP_OK(topK);
  auto zero_dead_rows = [&](int k) {
    if (debug_full || h->zero_dead) return true;
    if (k == L) return !top_fused;                                // k_top writes every row of layer L itself
    return !(reads_live_rows_only(k + 1, false) && reads_live_rows_only(k, true));
  };
  // ---- once per forward: classification lists, input embedding, embedding-independent feature chains ----
  PreAllArgs pre{};
  {
    pre.pack_f = h->d_pack[PK_PRE_FWD]; pre.pack_b = h->d_pack[PK_PRE_BWD];
    pre.L = L; pre.do_bwd = limit >= 2 ? 1 : 0; pre.cnt = cnt + 4;
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1, q = h->relu_q[k];
      pre.lb[i] = in->lb[k]; pre.ub[i] = in->ub[k]; pre.dual[i] = in->dual[k - 1];
      pre.z_pre[i] = in->primal[q - 1]; pre.z_post[i] = in->primal[q]; pre.bias[i] = h->dev[k].bias;
      pre.Pf[i] = ws + w.Pf[k]; pre.Pb[i] = ws + w.Pb[k]; pre.list[i] = ilist(w.amb[k]);
      pre.N[i] = h->N[k]; pre.hw[i] = h->hw[k];
    }
  }
  // a single subproblem: k_classify and k_pre in one launch (k_classify_pre; GNNB_CLSPRE_MAX_B, default 1)
  const bool cls_pre = h->bf3 && B <= h->clspre_max_b;
  {
    ClassifyArgs a{};
    a.L = L; a.mask = in->mask; a.scores = scores; a.cnt = cnt + 4; a.R = h->R;
    a.status = status; a.best = best; a.done = done_ctr; a.B = B;
    a.topflag = reinterpret_cast<int*>(ws + w.topflag); a.nflag = std::min(B, TOP_SPLIT_MAXB);
    a.mu2 = debug_full ? ws + w.F1 : nullptr;      // inspection runs keep the plain rows in mu[1] and the mapped ones in F1
    int blk = 0;
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1;
      a.lb[i] = in->lb[k]; a.ub[i] = in->ub[k]; a.mu[i] = mu(k); a.zero[i] = zero_dead_rows(k) ? 1 : 0;
      a.live[i] = ilist(w.live[k]); a.amb[i] = ilist(w.amb[k]); a.score[i] = ilist(w.score[k]);
      a.livef[i] = ws + w.lf[k];
      a.G[i] = (long)B * h->N[k]; a.N[i] = h->N[k]; a.off[i] = roff[k];
      a.blk0[i] = blk;
      blk += (int)((a.G[i] + CLS_BLOCK - 1) / CLS_BLOCK);
    }
    a.blk0[L] = blk;
    if (cls_pre) lz.run(PC_CLASSIFY, [&] { hipLaunchKernelGGL(k_classify_pre, dim3((unsigned)blk), dim3(CLS_THREADS), CLSPRE_LDS_BYTES, st, a, pre); });
    else lz.run(PC_CLASSIFY, [&] { hipLaunchKernelGGL(k_classify, dim3((unsigned)blk), dim3(CLS_THREADS), 0, st, a); });
  }

  {   // bias-sum scalars of every edge and direction (the rows carry deferred projections)
    LiveSumArgs a{};
    a.B = B;
    int q = 0, maxw = 0;
    auto push = [&](int kind, const Edge& e, const float* wt, int ld, const float* lf, float* out, int Ndst, int Nsrc, int normalise) {
      LiveSumJob& j = a.job[q++];
      j.kind = kind; j.w = wt; j.lf = lf; j.out = out; j.Ndst = Ndst; j.Nsrc = Nsrc; j.ld = ld; j.normalise = normalise;
      j.c_in = e.c_in; j.h_in = e.h_in; j.w_in = e.w_in; j.c_out = e.c_out; j.h_out = e.h_out; j.w_out = e.w_out;
      j.kh = e.kh; j.kw = e.kw; j.stride = e.stride; j.pad = e.pad;
      const long nw = (long)e.c_in * e.c_out * e.kh * e.kw;
      j.wlds = (e.kind == 0 && nw <= LIVESUM_MAXW) ? (int)nw : 0;
      maxw = std::max(maxw, j.wlds);
    };
    // edges whose aggregate comes from a sparse gather get their bias sums from that gather (GArgs.sout / GIArgs.s_from_gather)
    auto gather_has_s = [&](const DevGather& d, int bit) { return h->s_in_gather && d.ok && (h->gather_sparse & bit) != 0; };
    for (int k = 1; k <= L; ++k) {            // forward edge k: source layer k-1 (the input layer is all live)
      const Edge& e = h->edges[k];
      if (k == 1 && s1_table) continue;
      if (k >= 2 && gather_has_s(h->gf[k], h->gf[k].g.lanes == 16 ? 1 : 2)) continue;
      if (k == L && top_s_fwd) continue;
      push(e.kind == 0 ? 0 : 1, e, e.kind == 0 ? h->dev[k].w_fwd : h->dev[k].w_bwd, h->dev[k].ld_bwd, k > 1 ? ws + w.lf[k - 1] : nullptr,
           ws + w.sf[k], h->N[k], h->N[k - 1], 0);
    }
    if (limit >= 2)
      for (int k = 0; k < L; ++k) {           // edge k+1 transposed: source layer k+1
        const Edge& e = h->edges[k + 1];
        if (k == 0 ? gather_has_s(h->gb[1], 4) : gather_has_s(h->gb[k + 1], h->gb[k + 1].g.lanes == 16 ? 1 : 2)) continue;
        if (k == L - 1 && top_s_bwd) continue;
        push(e.kind == 0 ? 2 : 3, e, h->dev[k + 1].w_bwd, h->dev[k + 1].ld_bwd, ws + w.lf[k + 1], ws + w.sb[k], h->N[k], h->N[k + 1],
             k >= 1 ? 1 : 0);
      }
    a.njobs = q;
    if (q > 0) {
    int maxn = 0;
    for (int k = 0; k <= L; ++k) maxn = std::max(maxn, h->N[k]);
    a.lv_floats = (maxn + 3) & ~3;
    if ((size_t)(a.lv_floats + maxw) * 4 > 160 * 1024) {      // very wide layers: leave the weights in global memory
      for (int i = 0; i < q; ++i) a.job[i].wlds = 0;
      maxw = 0;
    }
    // (running this and k_pre on a side stream under k_embed / the first aggregation was measured: 1.72 ms vs 1.59 ms in-line)
    lz.run(PC_LIVESUM, [&] { hipLaunchKernelGGL(k_livesum, dim3((unsigned)B, (unsigned)q), dim3(256), (size_t)(a.lv_floats + maxw) * sizeof(float), st, a); });
    }                                   // (q == 0: every edge's bias sums come from its gather, k_top or the bind-time table)
  }
  {
    const long G = (long)B * h->N[0];
    EmbedArgs a{h->d_pack[PK_EMBED] + PackEmbed::W, h->d_pack[PK_EMBED] + PackEmbed::B, in->lb[0], in->x_lp, in->ub[0], mu(0), G};
    long grid = (G + 16 * EMBED_UNROLL - 1) / (16 * EMBED_UNROLL);
    if (grid > (long)h->n_cu * 16) grid = (long)h->n_cu * 16;
    // with the MFMA gather on the first edge, round 0 computes the embedding inside that gather (k_gather<true>): nothing
    // else reads mu[0] before the input-layer update overwrites it.  Inspection runs keep the rows.
    if (!embed_in_gather) lz.run(PC_EMBED, [&] { hipLaunchKernelGGL(k_embed, dim3((unsigned)grid), dim3(256), 0, st, a); });
    proj[0] = L_INP_F_1;
  }
  if (!cls_pre) {
    long nt = 0;                                      // upper bound: the kernel reads the real counts on the device
    for (int k = 1; k <= L; ++k) nt += (((long)B * h->N[k] + 31) / 32) * 2;
    const PreAllArgs& a = pre;
    const size_t lds = (h->bf3 ? (size_t)PackPreBwdL3::FLOATS : (size_t)PackPreFwd::FLOATS + PackPreBwd::FLOATS) * 4;
    lz.run(PC_PRE, [&] {
      if (h->bf3) hipLaunchKernelGGL(k_pre<true>, dim3(mlp_grid(h, nt / 8)), dim3(PRE_WAVES * 64), lds, st, a);
      else hipLaunchKernelGGL(k_pre<false>, dim3(mlp_grid(h, nt / 8)), dim3(PRE_WAVES * 64), lds, st, a);
    });
  }
  const bool need_inp = (limit >= 2) && (h->T > 1 || debug_full) && !h->gb[1].ok;    // the fused input kernel computes Q itself
  if (need_inp) {
    const long G = (long)B * h->N[0];
    const TileMap tm = bwd_map(h, 0);
    const long nt = map_tiles(tm, B);
    PreArgs a{h->d_pack[PK_PRE_INP], in->lb[0], in->ub[0], nullptr, nullptr, nullptr, nullptr, ws + w.Q, G, nt, h->N[0], 1,
              to_dtm(tm), nullptr, nullptr};
    lz.run(PC_PRE_INP, [&] { hipLaunchKernelGGL(k_pre_inp, dim3(mlp_grid(h, nt)), dim3(WG_MLP), PackPreInp::FLOATS * 4, st, a); });
  }

  auto conv_args = [&](const Edge& e, const float* src, float* dst, const float* wt, int normalise) {
    return ConvArgs{src, dst, wt, B, e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, normalise};
  };
  auto gather = [&](const DevGather& d, int k, const float* src, bool scored, bool embed_src, int src_layer, float* sout) {      // phase A over a conv edge, MFMA
    const long nt = map_tiles(d.g.tm, B);
    // sparse: a 16-node forward gather behind a ReLU layer skips the (zero) rows of that layer's dead nodes
    const bool sparse = (h->gather_sparse & (d.g.lanes == 16 ? 1 : 2)) && !embed_src && src_layer >= 1;
    GArgs a{in->lb[k], in->ub[k], in->mask, src, nb, nt, scored ? 1 : 0, h->R, roff[k], to_dtm(d.g.tm), to_dg(d, h->d_zero),
            EmbedSrc{in->lb[0], in->x_lp, in->ub[0], h->d_pack[PK_EMBED]}, sparse ? in->lb[src_layer] : nullptr, sparse ? in->ub[src_layer] : nullptr,
            sparse && h->s_in_gather ? sout : nullptr};
    const size_t lds = gather_lds_bytes(d, 0) + (sparse ? sparse_tab_bytes(d) : 0) + (d.g.lanes == 16 ? 16 + (size_t)WAVES_MLP * STAGE16_FLOATS * 4 : 0);
    long grid = (nt + WAVES_MLP - 1) / WAVES_MLP;
    if (grid > (long)h->n_cu * h->gather_occ) grid = (long)h->n_cu * h->gather_occ;
    lz.run(PC_GATHER, [&] {
      if (d.g.lanes == 16) {
        if (embed_src) hipLaunchKernelGGL((k_gather16<true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
        else if (sparse) hipLaunchKernelGGL((k_gather16<false, true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
        else hipLaunchKernelGGL((k_gather16<false>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      } else if (embed_src) hipLaunchKernelGGL(k_gather<true>, dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      else if (sparse) hipLaunchKernelGGL((k_gather<false, true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      else hipLaunchKernelGGL(k_gather<false>, dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
    });
  };
  // phase A: nb <- A_k mu[k-1]
  auto agg_fwd = [&](int k) {
    const Edge& e = h->edges[k];
    if (h->gf[k].ok) { gather(h->gf[k], k, mu(k - 1), false, k == 1 && embed_in_gather && proj[0] == L_INP_F_1, k - 1, ws + w.sf[k]); return; }
    if (e.kind == 0) {
      ConvArgs a = conv_args(e, mu(k - 1), nb, h->dev[k].w_fwd, 0);
      lz.run(PC_CONV_FWD, [&] {
        switch (e.c_out) {
          case 3: launch_conv_fwd<3>(a, st); break;
          case 8: launch_conv_fwd<8>(a, st); break;
          case 16: launch_conv_fwd<16>(a, st); break;
          default: launch_conv_fwd<32>(a, st); break;
        }
      });
    } else {
      const DevEdge& de = h->dev[k];
      if (h->dense_lds && per_sample && de.mt_fwd <= 4) {        // one workgroup per sample, source rows staged in LDS
        DenseLArgs a{de.w_fwd, mu(k - 1), nb, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.kpad_fwd};
        lz.run(PC_DENSE_AGG, [&] { hipLaunchKernelGGL(k_dense_fwd_lds, dim3(B), dim3(512), 0, st, a); });
        return;
      }
      DenseArgs a{de.w_fwd, mu(k - 1), nb, h->d_zero, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.ksq_fwd};
      const long tiles = (long)B * a.MT;
      lz.run(PC_DENSE_AGG, [&] {
        if (a.K >= 512) hipLaunchKernelGGL(k_dense_agg<true>, dim3((unsigned)tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_dense_agg<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
      });
    }
  };
  // phase A: nb <- A_{k+1}^T mu[k+1]  (k+1 <= L), conv case divided by the tap count when `normalise`
  auto agg_bwd = [&](int k, int normalise, bool scored) {
    const Edge& e = h->edges[k + 1];
    // the input layer (k = 0) aggregates the rows of layer 1 that already went through its 64x64 map (PackPostInp)
    const float* srcb = k == 0 ? rows1_for_input : mu(k + 1);
    if (k >= 1 && scored && h->gb[k + 1].ok && h->scored_gather && (h->gather_sparse & 2) &&
        e.c_out * ((e.kh + e.stride - 1) / e.stride) * ((e.kw + e.stride - 1) / e.stride) <= GS_SLOT_LIMIT) {
      // the restricted last step as three kernels (GNNB_TAIL_MAX_B=0; the default is k_scored_tail): one wave per scored node instead of
      // every tile that holds one (k_gather_scored).  Windows up to GS_SLOT_LIMIT source nodes (base, 64 slots: 31 vs 38 us for the tile
      // gather; deep 18 vs 38; wide, 128 slots: 116 vs 99 -- kept on the list-driven form all the same, so that this path and
      // k_scored_tail evaluate a scored node's aggregate with the same arithmetic)
      GSArgs a{ilist(w.score[k]), cnt + 4 * k + 2, mu(k + 1), h->dev[k + 1].w_bwd, in->lb[k + 1], in->ub[k + 1], nb, h->s_in_gather ? ws + w.sb[k] : nullptr,
               h->N[k], e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, normalise};
      lz.run(PC_GATHER, [&] { hipLaunchKernelGGL(k_gather_scored, dim3((unsigned)h->n_cu * 4), dim3(GS_WAVES * 64), 0, st, a); });
      return;
    }
    if (k >= 1 && h->gb[k + 1].ok) { gather(h->gb[k + 1], k, mu(k + 1), scored, false, k + 1, ws + w.sb[k]); return; }
    if (e.kind == 0) {
      ConvArgs a = conv_args(e, srcb, nb, h->dev[k + 1].w_bwd, normalise);
      lz.run(PC_CONVT_BWD, [&] {
        switch (e.c_in) {
          case 3: launch_convT<3>(a, st); break;
          case 8: launch_convT<8>(a, st); break;
          case 16: launch_convT<16>(a, st); break;
          default: launch_convT<32>(a, st); break;
        }
      });
    } else {
      const DevEdge& de = h->dev[k + 1];
      if (h->dense_lds && per_sample && de.kpad_bwd <= 128) {    // one workgroup per sample, the whole source layer in LDS
        DenseLArgs a{de.w_bwd, srcb, nb, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.kpad_bwd};
        lz.run(PC_DENSE_AGG, [&] { hipLaunchKernelGGL(k_dense_bwd_lds, dim3(B), dim3(512), 0, st, a); });
        return;
      }
      DenseArgs a{de.w_bwd, srcb, nb, h->d_zero, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.ksq_bwd};
      const long tiles = (long)B * a.MT;
      lz.run(PC_DENSE_AGG, [&] {
        if (a.K >= 512) hipLaunchKernelGGL(k_dense_agg<true>, dim3((unsigned)tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_dense_agg<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
      });
    }
  };
  // phase B: node MLP over a compacted list of nodes
  // post_input: this is the backward update of layer 1 and an input-layer update follows -- the kernel also applies the input
  // update's 64x64 map to its rows (PackPostInp) and writes them to F1; only inspection runs still need the plain rows
  // the arguments of the node update of layer k (shared by k_node_update and the fused k_gather_update)
  auto upd_args = [&](int k, bool fwd, bool scored, bool post_input) {
    // the aggregate in `nb` was built from rows whose last Linear is deferred (gnnb_pack.h), except the one k_prop writes
    const int src_proj = fwd ? proj[k - 1] : (k < L ? proj[k + 1] : -1);
    int pack = PK_UPD_BWD;
    const float* sarr = nullptr;
    int smod = 0;
    if (fwd) {
      pack = src_proj == L_INP_F_1 ? PK_UPD_FWD_E : (src_proj == L_INP_B2_2 ? PK_UPD_FWD_I : PK_UPD_FWD_F);
      sarr = ws + w.sf[k];
      if (k == 1 && s1_table) { sarr = h->d_s1; smod = h->N[1]; }      // the input layer is all live: one table for every sample
    } else if (k < L) {
      pack = PK_UPD_BWD_B;
      sarr = ws + w.sb[k];
    }
    // normal: list0 = live non-ambiguous nodes (short chain), list1 = ambiguous nodes; restricted: the scored nodes, general chain
    UpdArgs a{h->d_pack[pack], in->lb[k], in->ub[k], nb, ws + (fwd ? w.Pf[k] : w.Pb[k]), (post_input && !debug_full) ? nullptr : mu(k), status,
              ilist(w.live[k]), cnt + 4 * k + (scored ? 3 : 0), ilist(scored ? w.score[k] : w.amb[k]), cnt + 4 * k + (scored ? 2 : 1), sarr, smod,
              post_input ? rows1_for_input : nullptr, nullptr, nullptr};
    if (post_input && input_rows3) a.post3 = ws + w.F3;
    a.wp = h->d_pack[PK_POST_INP] + (h->bf3 ? (h->gb[1].ok ? PackPostInp::WPG3 : PackPostInp::WPN3) : (h->gb[1].ok ? PackPostInp::WPG : PackPostInp::WPN));
    return a;
  };
  // phase B: node MLP over a compacted list of nodes
  // post_input: this is the backward update of layer 1 and an input-layer update follows -- the kernel also applies the input
  // update's 64x64 map to its rows (PackPostInp) and writes them to F1; only inspection runs still need the plain rows
  auto node_update = [&](int k, bool fwd, bool scored, bool post_input = false) {
    const long nt = ((long)B * h->N[k] + 31) / 32;
    const UpdArgs a = upd_args(k, fwd, scored, post_input);
    const bool deferred = a.sarr != nullptr;
    const bool bf3 = h->bf3;
    const int wv = bf3 ? 12 : 8;  // waves per workgroup (one workgroup per CU shares the LDS weights)
    const size_t ldsb = bf3 ? (size_t)(PackUpdL3::FLOATS + (post_input ? 6144 : 0)) * 4 : (size_t)(PackUpd::FLOATS + (post_input ? 4096 : 0)) * 4;
    long grid = (nt + wv - 1) / wv;
    if (grid > h->n_cu) grid = h->n_cu;
    lz.run(PC_NODE_UPDATE, [&] {
      const dim3 g((unsigned)grid), b12(768), b8(512);
      if (bf3) {
        if (post_input && deferred) hipLaunchKernelGGL((k_node_update<12, true, true, true>), g, b12, ldsb, st, a);
        else if (post_input) hipLaunchKernelGGL((k_node_update<12, false, true, true>), g, b12, ldsb, st, a);
        else if (deferred) hipLaunchKernelGGL((k_node_update<12, true, false, true>), g, b12, ldsb, st, a);
        else hipLaunchKernelGGL((k_node_update<12, false, false, true>), g, b12, ldsb, st, a);
      } else if (post_input) {
        if (deferred) hipLaunchKernelGGL((k_node_update<8, true, true>), g, b8, ldsb, st, a);
        else hipLaunchKernelGGL((k_node_update<8, false, true>), g, b8, ldsb, st, a);
      } else if (deferred) hipLaunchKernelGGL((k_node_update<8, true>), g, b8, ldsb, st, a);
      else hipLaunchKernelGGL((k_node_update<8, false>), g, b8, ldsb, st, a);
    });
    proj[k] = fwd ? L_FC4_2 : L_BC4_1;
  };
  // One half-pass over a conv edge as ONE kernel (k_gather_update): the gather of edge k (forward) / k + 1 (transposed) and the
  // node update of layer k, the aggregate staying in registers.  Returns false where the two-kernel path has to run: inspection
  // runs, the restricted last step (every node it updates takes the general chain), gathers without the sparse walk behind a
  // ReLU layer, tile forms the fused kernel is not built for, tables that do not fit beside the weights in LDS.
  auto fused_halfpass = [&](int k, bool fwd, bool post_input) -> bool {
    if (debug_full) return false;
    const DevGather& d = fwd ? h->gf[k] : h->gb[k + 1];
    if (!d.ok || (!fwd && k >= L)) return false;
    const int src_layer = fwd ? k - 1 : k + 1;
    const bool embed_src = fwd && k == 1 && embed_in_gather && proj[0] == L_INP_F_1;
    if (!fusedq_ok(h, d, src_layer, embed_src, post_input)) return false;
    const bool sparse = (h->gather_sparse & (d.g.lanes == 16 ? 1 : 2)) && !embed_src && src_layer >= 1;
    const long nt = map_tiles(d.g.tm, B);
    {
      const int nq = fusedq_qtiles(d, sparse, post_input);
      const size_t ldsq = fusedq_lds_bytes(d, sparse, post_input, nq);
      FArgs a{};
      float* sout = fwd ? ws + w.sf[k] : ws + w.sb[k];
      a.sw_from_gather = sparse && h->s_in_gather ? 1 : 0;
      a.g = GArgs{in->lb[k], in->ub[k], in->mask, fwd ? mu(k - 1) : mu(k + 1), nb, nt, 0, h->R, roff[k], to_dtm(d.g.tm), to_dg(d, h->d_zero),
                  EmbedSrc{in->lb[0], in->x_lp, in->ub[0], h->d_pack[PK_EMBED]}, sparse ? in->lb[src_layer] : nullptr, sparse ? in->ub[src_layer] : nullptr,
                  a.sw_from_gather ? sout : nullptr};
      a.u = upd_args(k, fwd, false, post_input);
      a.qtiles = nq;
      const long nrounds = (nt + QG_WAVES - 1) / QG_WAVES;
      const dim3 g((unsigned)std::max<long>(1, std::min<long>(nrounds, h->n_cu))), b((QG_WAVES + QC_WAVES) * 64);
      lz.run(PC_GATHER_UPDATE, [&] {
        if (d.g.lanes == 16) {
          if (embed_src) hipLaunchKernelGGL((k_gather_update_q<16, 2, false>), g, b, ldsq, st, a);
          else if (sparse) hipLaunchKernelGGL((k_gather_update_q<16, 1, false>), g, b, ldsq, st, a);
          else hipLaunchKernelGGL((k_gather_update_q<16, 0, false>), g, b, ldsq, st, a);
        } else if (post_input) hipLaunchKernelGGL((k_gather_update_q<32, 1, true>), g, b, ldsq, st, a);
        else hipLaunchKernelGGL((k_gather_update_q<32, 1, false>), g, b, ldsq, st, a);
      });
      proj[k] = fwd ? L_FC4_2 : L_BC4_1;
      return true;
    }
  };
  auto update_input = [&]() {
    proj[0] = L_INP_B2_2;
    if (h->gb[1].ok) {
      const DevGather& d = h->gb[1];
      const long nt = map_tiles(d.g.tm, B);
      const bool sparse = (h->gather_sparse & 4) != 0;
      GIArgs a{h->d_pack[PK_PRE_INP], h->d_pack[PK_UPD_INP], in->lb[0], in->ub[0], rows1_for_input, ws + w.sb[0], mu(0), nt, to_dtm(d.g.tm), to_dg(d, h->d_zero),
               in->lb[1], in->ub[1], sparse && h->s_in_gather ? 1 : 0, (sparse && input_rows3) ? (const void*)(ws + w.F3) : nullptr};
      const bool r3 = a.mu_src3 != nullptr;
      const int nw = r3 ? GIU_R3_WAVES : WAVES_MLP;
      const size_t lds = gather_lds_bytes(d, PackUpdInp::FLOATS + PackPreInp::FLOATS) +
                         (sparse ? 8 + (size_t)nw * (2 * d.g.K2 + 32) * 8 : 0) + (r3 ? (size_t)d.g.tm.NCG * d.g.K2 * 128 * 4 : 0);
      long giu_grid = (nt + nw - 1) / nw;
      if (giu_grid > (long)h->n_cu * (r3 ? 1 : h->giu_occ)) giu_grid = (long)h->n_cu * (r3 ? 1 : h->giu_occ);
      lz.run(PC_GATHER_INPUT, [&] {
#ifdef GNNB_DEV
        if (r3) { hipLaunchKernelGGL((k_gather_input_update<true, true, true>), dim3(giu_grid), dim3(GIU_R3_WAVES * 64), lds, st, a); return; }
#endif
        if (sparse && h->bf3) hipLaunchKernelGGL((k_gather_input_update<true, true>), dim3(giu_grid), dim3(WG_MLP), lds, st, a);
        else if (sparse) hipLaunchKernelGGL((k_gather_input_update<true, false>), dim3(giu_grid), dim3(WG_MLP), lds, st, a);
        else if (h->bf3) hipLaunchKernelGGL((k_gather_input_update<false, true>), dim3(giu_grid), dim3(WG_MLP), lds, st, a);
        else hipLaunchKernelGGL((k_gather_input_update<false, false>), dim3(giu_grid), dim3(WG_MLP), lds, st, a);
      });
      return;
    }
    agg_bwd(0, 0, false);
    const long G = (long)B * h->N[0], nt = (G + 31) / 32;
    UpdInpArgs a{h->d_pack[PK_UPD_INP], nb, ws + w.Q, ws + w.sb[0], mu(0), G, nt};
    lz.run(PC_INPUT_UPDATE, [&] { hipLaunchKernelGGL(k_input_update, dim3(mlp_grid(h, nt)), dim3(WG_MLP), PackUpdInp::FLOATS * 4, st, a); });
  };

  // the top of the network as one launch per round (k_top); with a half-pass limit (inspection) the separate kernels run
  int top_launches = 0;
  auto top = [&]() {
    const Edge& e = h->edges[L];
    const DevEdge& de = h->dev[L];
    TopArgs a{};
    a.df = DenseLArgs{de.w_fwd, mu(L - 1), nullptr, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.kpad_fwd};
    a.db = DenseLArgs{de.w_bwd, nullptr, nb, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.kpad_bwd};
    a.pack_f = h->d_pack[PK_UPD_FWD_F]; a.pack_b = h->d_pack[PK_UPD_BWD]; a.pack_p = h->d_pack[PK_PROP];
    a.Pf = ws + w.Pf[L]; a.Pb = ws + w.Pb[L];
    a.sf = top_s_fwd ? nullptr : ws + w.sf[L];              // null: F1 walks exactly the live rows of layer L-1 and sums its weights itself
    a.sb_out = top_s_bwd ? ws + w.sb[L - 1] : nullptr;
    a.lb = in->lb[L]; a.ub = in->ub[L];
    a.lbm = in->lb[L - 1]; a.ubm = in->ub[L - 1];
    a.prop_w = in->prop_w; a.prop_b = in->prop_b; a.lbK = in->lb[K]; a.ubK = in->ub[K]; a.z_out = in->primal[in->n_primal - 1];
    a.mu_prop = mu(K); a.mu = mu(L); a.status = status; a.N = h->N[L];
    // A sample's top spread over S = 2 / 4 workgroups (by output tile of both Linear edges) while all B x S of them are resident
    // at one per CU (they wait for each other); GNNB_TOP_SPLIT = 1 / 2 / 4 caps S.  The results do not depend on S.
    // S = 4 while B <= n_cu / 4 (base B = 1: 52 -> 38 us per launch), S = 2 while B <= n_cu / 2.  (Before k_top also ran the update of
    // layer L-1, S = 2 was a draw -- the two hand-offs cost what the shorter edges saved; with the update's tiles split over both
    // workgroups too it wins: deep B = 128 67 -> 59 us, wide B = 128 99 -> 79 us per launch.)
    int S = 1;
    if (h->top_split_max >= 4 && (long)B * 4 <= h->n_cu && B <= TOP_SPLIT_MAXB) S = 4;      // (topflag / xbuf are sized for TOP_SPLIT_MAXB samples)
    else if (h->top_split_max >= 2 && (long)B * 2 <= h->n_cu && B <= TOP_SPLIT_MAXB) S = 2;
    a.xbuf = ws + w.topx; a.xflag = reinterpret_cast<int*>(ws + w.topflag); a.xbase = top_launches * 2 * S;
    a.fuse_um = top_upd ? 1 : 0;
    if (top_upd) a.um = upd_args(L - 1, false, false, false);
    ++top_launches;
    lz.run(PC_TOP, [&] {
      if (S == 4) hipLaunchKernelGGL(k_top<1>, dim3(B * 4), dim3(512), TOP_LDS_FLOATS * 4, st, a);
      else if (S == 2) hipLaunchKernelGGL(k_top<2>, dim3(B * 2), dim3(512), TOP_LDS_FLOATS * 4, st, a);
      else hipLaunchKernelGGL(k_top<4>, dim3(B), dim3(512), TOP_LDS_FLOATS * 4, st, a);
    });
    proj[L] = L_BC4_1;
  };

  // The restricted last step (scored gather + node update of layer 1) and the score head are ONE launch, k_scored_tail
  // (GNNB_TAIL_MAX_B=0: the three kernels k_gather_scored, k_node_update, k_score)
  bool tail_fused = false;
  TailArgs tail{};
  auto try_tail = [&](int k) -> bool {
    const Edge& e = h->edges[k + 1];
    if (!(B <= h->tail_max_b && h->bf3 && k == 1 && L >= 2 && h->restrict_last && !debug_full && e.kind == 0 && h->gb[k + 1].ok && h->scored_gather &&
          (h->gather_sparse & 2) && e.c_out * ((e.kh + e.stride - 1) / e.stride) * ((e.kw + e.stride - 1) / e.stride) <= GS_SLOT_LIMIT &&
          h->N[k + 1] < 65536))
      return false;
    tail.g = GSArgs{ilist(w.score[k]), cnt + 4 * k + 2, mu(k + 1), h->dev[k + 1].w_bwd, in->lb[k + 1], in->ub[k + 1], nullptr, nullptr,
                    h->N[k], e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, 1};
    tail.f = FArgs{};
    tail.f.u = upd_args(k, false, true, false);
    tail.f.sw_from_gather = 1;
    tail_fused = true;
    proj[k] = L_BC4_1;
    return true;
  };
  int done = 0;
  for (int t = 0; t < h->T && done < limit; ++t) {
    if (top_fused) {
      for (int k = 1; k < L; ++k) {
        if (fused_halfpass(k, true, false)) continue;
        agg_fwd(k);
        node_update(k, true, false);
      }
      top();                                     // F1 .. B2: both half-passes of layer L, aggregate of layer L-1 in `nb`
      for (int k = L - 1; k >= 1; --k) {
        if (k == L - 1 && top_upd) { proj[k] = L_BC4_1; continue; }      // done inside k_top
        const bool scored = h->restrict_last && t == h->T - 1 && k == 1;
        if (scored && k < L - 1 && try_tail(k)) continue;
        if (k < L - 1 && !scored && fused_halfpass(k, false, k == 1 && t < h->T - 1)) continue;
        if (k < L - 1) agg_bwd(k, 1, scored);
        node_update(k, false, scored, k == 1 && t < h->T - 1);
      }
      if (t < h->T - 1) update_input();
      done += 2;
      continue;
    }
    // forward sweep (graph_conv.py:107-192) + property node (:194-210)
    for (int k = 1; k <= L; ++k) {
      if (fused_halfpass(k, true, false)) continue;
      agg_fwd(k);
      node_update(k, true, false);
    }
    {
      // the backward sweep starts with the edge from the property node: its aggregate is written by the same kernel
      const bool bwd_follows = done + 1 < limit;
      PropArgs a{h->d_pack[PK_PROP], mu(L), in->prop_w, in->prop_b, in->lb[K], in->ub[K], in->primal[in->n_primal - 1], mu(K),
                 bwd_follows ? nb : nullptr, B, h->N[L], in->lb[L], in->ub[L]};
      lz.run(PC_PROP_FWD, [&] { hipLaunchKernelGGL(k_prop, dim3(B), dim3(256), 0, st, a); });
    }
    if (++done >= limit) break;
    // backward sweep (:222-350), Gauss-Seidel order: layer k reads the already-updated mu[k+1]
    for (int k = L; k >= 1; --k) {
      // after the last backward step mu[1] is only read by the score head, i.e. at the scored nodes
      const bool scored = h->restrict_last && !debug_full && t == h->T - 1 && k == 1;
      if (k < L && !scored && fused_halfpass(k, false, k == 1 && t < h->T - 1)) continue;
      if (k < L) agg_bwd(k, 1, scored);          // (k == L: k_prop already wrote the aggregate from the property node)
      node_update(k, false, scored, k == 1 && (t < h->T - 1 || debug_full));
    }
    // input layer (:360-385): its last-round result is never read, so it only runs when another round follows
    if (t < h->T - 1 || debug_full) update_input();
    ++done;
  }

  // scores (graph_conv.py:442-450) and decision (graph_score.py:41-47)
  {
    ScoreArgs a{};
    a.best = best; a.done = done_ctr; a.dec = decisions; a.B = B; a.n_relu = L;
    a.pack = h->d_pack[proj[1] == L_FC4_2 ? PK_SCORE_F : PK_SCORE_B]; a.scores = scores; a.L = L; a.R = h->R; a.cnt = cnt + 4; a.cnt_all = cnt;
    long nt = 0;
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1;
      a.mu[i] = mu(k); a.list[i] = ilist(w.score[k]); a.N[i] = h->N[k]; a.off[i] = roff[k];
      a.lb[i] = in->lb[k]; a.ub[i] = in->ub[k];
      nt += ((long)B * h->N[k] + 31) / 32;
      a.cum[k - 1] = roff[k] + h->N[k];
    }
    if (tail_fused) {
      tail.s = a;
      int grid = (int)std::min<long>(h->n_cu, std::max<long>(1, ((long)B * h->N[1] + 15) / 16));
      const int nslots = tail.g.Co * ((tail.g.kh + tail.g.stride - 1) / tail.g.stride) * ((tail.g.kw + tail.g.stride - 1) / tail.g.stride);
      tail.sp = tail_slots_pad(nslots);
      lz.run(PC_SCORE, [&] { hipLaunchKernelGGL(k_scored_tail, dim3(grid), dim3(TAIL_WAVES * 64), tail_lds_bytes(nslots), st, tail); });
    } else
    lz.run(PC_SCORE, [&] { hipLaunchKernelGGL(k_score, dim3(mlp_grid(h, nt / 4)), dim3(WG_MLP), PackScore::FLOATS * 4, st, a); });
  }
  for (int k = 0; k < MAXL + 2; ++k) h->last_proj[k] = proj[k];      // inspection (gnnb_mu_projection); one handle per thread
  if (lz.rc) h->ctl_ws[slot] = nullptr;          // a launch failed: the counters may be left non-zero, the block is re-zeroed on its next use
  return lz.rc;
}

// gnnb_forward for HOST inputs -- the reference's own call pattern: one or two subproblems per decision, every tensor a CPU
// tensor (graph_score.py:26-30 moves them with ~14 .cuda() calls).  All inputs the forward reads are packed into one pinned
// buffer (256-B aligned slots) and cross PCIe as ONE copy; the forward runs on `stream`; decisions, status and (optionally) the
// padded scores come back in one pinned block; the call returns after synchronising the stream.  Buffers live in the handle.
extern "C" int gnnb_forward_host(gnnb_t* h, const gnnb_batch* in, int B, float* scores, int32_t* decisions, int32_t* status, void* stream) {
  if (!h || !in || !decisions || !status) return fail(GNNB_E_INVALID, "gnnb_forward_host: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_forward_host: call gnnb_bind_network first");
  const int K = (int)h->N.size() - 1, L = K - 1, R = h->R;
  if (B < 1 || in->n_graph != K + 1 || in->n_relu != L || in->n_primal != h->n_fixed + 1)
    return fail(GNNB_E_INVALID, "gnnb_forward_host: batch does not match the bound network");
  hipStream_t st = (hipStream_t)stream;
  // ---- slots: (host pointer, floats); primals the forward never reads are not transferred
  struct Slot { const float* src; size_t n, off; };
  std::vector<Slot> slots;
  size_t total = 0;
  auto add = [&](const float* p, size_t n) { slots.push_back(Slot{p, n, total}); total += (n + 63) & ~(size_t)63; return slots.size() - 1; };
  std::vector<size_t> i_lb(K + 1), i_ub(K + 1), i_dual(L), i_prim(in->n_primal, (size_t)-1);
  for (int k = 0; k <= K; ++k) { i_lb[k] = add(in->lb[k], (size_t)B * h->N[k]); i_ub[k] = add(in->ub[k], (size_t)B * h->N[k]); }
  for (int k = 0; k < L; ++k) i_dual[k] = add(in->dual[k], (size_t)B * h->N[k + 1] * 3);
  for (int k = 1; k <= L; ++k) {
    const int q = h->relu_q[k];
    for (int m : {q - 1, q})
      if (i_prim[m] == (size_t)-1) i_prim[m] = add(in->primal[m], (size_t)B * h->N[k]);
  }
  if (i_prim[in->n_primal - 1] == (size_t)-1) i_prim[in->n_primal - 1] = add(in->primal[in->n_primal - 1], (size_t)B);
  const size_t i_x = add(in->x_lp, (size_t)B * h->N[0]), i_pw = add(in->prop_w, (size_t)B * h->N[L]), i_pb = add(in->prop_b, (size_t)B);
  const size_t i_mask = add(in->mask, (size_t)B * R);
  for (const Slot& sl : slots) {
    if (!sl.src) return fail(GNNB_E_INVALID, "gnnb_forward_host: null input pointer");
    // these are read by memcpy on the host: device memory here is a caller bug (e.g. data_ptr() of a `.cuda()` tensor), refused
    // rather than dereferenced.  ~0.2 us per pointer: the runtime's allocation map, no driver call
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, sl.src) != hipSuccess) { (void)hipGetLastError(); continue; }      // plain host memory
    if (at.type == hipMemoryTypeDevice) return fail(GNNB_E_INVALID, "gnnb_forward_host: an input pointer is device memory; this entry point takes host pointers (gnnb_forward takes device pointers)");
  }
  // ---- buffers
  if (h->hs_floats < total) {
    if (h->hs_pinned) (void)hipHostFree(h->hs_pinned);
    if (h->hs_dev) (void)hipFree(h->hs_dev);
    h->hs_pinned = nullptr; h->hs_dev = nullptr; h->hs_floats = 0;
    HIPCHK(hipHostMalloc((void**)&h->hs_pinned, total * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&h->hs_dev, total * sizeof(float)));
    h->hs_floats = total;
  }
  const size_t wsb = gnnb_workspace_bytes(h, B);
  if (h->hs_ws_bytes < wsb) {
    if (h->hs_ws) (void)hipFree(h->hs_ws);
    h->hs_ws = nullptr; h->hs_ws_bytes = 0;
    HIPCHK(hipMalloc(&h->hs_ws, wsb));
    h->hs_ws_bytes = wsb;
  }
  if (h->hs_out_B < (size_t)B) {
    if (h->hs_scores) (void)hipFree(h->hs_scores);
    if (h->hs_dec) (void)hipFree(h->hs_dec);
    if (h->hs_out_pinned) (void)hipHostFree(h->hs_out_pinned);
    h->hs_scores = nullptr; h->hs_dec = nullptr; h->hs_out_pinned = nullptr; h->hs_out_B = 0;
    HIPCHK(hipMalloc((void**)&h->hs_scores, (size_t)B * R * sizeof(float)));
    HIPCHK(hipMalloc((void**)&h->hs_dec, ((size_t)B * 2 + 1) * sizeof(int32_t)));          // decisions, then the status word
    HIPCHK(hipHostMalloc((void**)&h->hs_out_pinned, ((size_t)B * R + (size_t)B * 2 + 1) * sizeof(float), hipHostMallocDefault));
    h->hs_out_B = B;
  }
  // ---- stage, one transfer, forward
  // staging: one memcpy per input tensor; big batches (36.5 MB at base B = 256: 2.3 ms on one thread) are split over a few helper threads
  if (total * sizeof(float) < (size_t)4 << 20) {
    for (const Slot& sl : slots) memcpy(h->hs_pinned + sl.off, sl.src, sl.n * sizeof(float));
  } else {
    const int nthr = 8;
    const size_t chunk = (size_t)1 << 18;                 // floats (1 MB) per work item
    std::vector<std::pair<size_t, size_t>> items;         // (slot, first float)
    for (size_t i = 0; i < slots.size(); ++i)
      for (size_t o = 0; o < slots[i].n; o += chunk) items.emplace_back(i, o);
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t it = next.fetch_add(1); it < items.size(); it = next.fetch_add(1)) {
        const Slot& sl = slots[items[it].first];
        const size_t o = items[it].second, n = std::min(chunk, sl.n - o);
        memcpy(h->hs_pinned + sl.off + o, sl.src + o, n * sizeof(float));
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  }
  HIPCHK(hipMemcpyAsync(h->hs_dev, h->hs_pinned, total * sizeof(float), hipMemcpyHostToDevice, st));
  std::vector<const float*> lb(K + 1), ub(K + 1), dual(L), prim(in->n_primal);
  for (int k = 0; k <= K; ++k) { lb[k] = h->hs_dev + slots[i_lb[k]].off; ub[k] = h->hs_dev + slots[i_ub[k]].off; }
  for (int k = 0; k < L; ++k) dual[k] = h->hs_dev + slots[i_dual[k]].off;
  for (int m = 0; m < in->n_primal; ++m) prim[m] = i_prim[m] == (size_t)-1 ? h->hs_dev : h->hs_dev + slots[i_prim[m]].off;   // (unread ones: any valid pointer)
  gnnb_batch dv{lb.data(), ub.data(), dual.data(), prim.data(), h->hs_dev + slots[i_x].off, h->hs_dev + slots[i_pw].off,
                h->hs_dev + slots[i_pb].off, h->hs_dev + slots[i_mask].off, in->n_graph, in->n_relu, in->n_primal};
  int32_t* d_status = h->hs_dec + (size_t)B * 2;
  if (int rc = gnnb_forward(h, &dv, B, h->hs_scores, h->hs_dec, d_status, h->hs_ws, h->hs_ws_bytes, stream)) return rc;
  int32_t* out_i = reinterpret_cast<int32_t*>(h->hs_out_pinned);
  HIPCHK(hipMemcpyAsync(out_i, h->hs_dec, ((size_t)B * 2 + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  float* out_s = h->hs_out_pinned + (size_t)B * 2 + 1;
  if (scores) HIPCHK(hipMemcpyAsync(out_s, h->hs_scores, (size_t)B * R * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  memcpy(decisions, out_i, (size_t)B * 2 * sizeof(int32_t));
  *status = out_i[(size_t)B * 2];
  if (scores) memcpy(scores, out_s, (size_t)B * R * sizeof(float));
  return GNNB_OK;
}

// ---- host-fed batches: compact records of the ambiguous nodes instead of whole dual / primal tensors (include/gnnb.h) --------------------
extern "C" size_t gnnb_amb_records_bytes(const gnnb_t* h, int B) {
  if (!h || !h->bound || B < 1) return 0;
  const int L = (int)h->N.size() - 2;
  size_t words = 16 + (size_t)((B + 3) & ~3);
  for (int k = 1; k <= L; ++k) words += (size_t)AMBREC_WORDS * B * h->N[k];
  return words * 4;
}

extern "C" int gnnb_pack_amb_records(const gnnb_t* hc, const gnnb_batch* in, int B, void* dst, size_t cap, size_t* used) {
  gnnb_t* h = const_cast<gnnb_t*>(hc);                  // (the helper threads live in the handle)
  if (!h || !in || !dst || !used) return fail(GNNB_E_INVALID, "gnnb_pack_amb_records: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_pack_amb_records: call gnnb_bind_network first");
  const int K = (int)h->N.size() - 1, L = K - 1;
  if (B < 1 || L > MAXL || in->n_graph != K + 1 || in->n_relu != L || in->n_primal != h->n_fixed + 1)
    return fail(GNNB_E_INVALID, "gnnb_pack_amb_records: batch does not match the bound network");
  for (int k = 1; k <= L; ++k) {
    const int q = h->relu_q[k];
    if (!in->lb[k] || !in->ub[k] || !in->dual[k - 1] || !in->primal[q - 1] || !in->primal[q])
      return fail(GNNB_E_INVALID, "gnnb_pack_amb_records: null input pointer (layer %d)", k);
    if ((long)B * h->N[k] > 0x7fffffffL) return fail(GNNB_E_INVALID, "gnnb_pack_amb_records: batch too large");
  }
  if (!in->primal[in->n_primal - 1]) return fail(GNNB_E_INVALID, "gnnb_pack_amb_records: null primals[-1]");
  const size_t zwords = (size_t)((B + 3) & ~3);
  if (cap < (16 + zwords) * 4) return fail(GNNB_E_NOMEM, "gnnb_pack_amb_records: buffer of %zu bytes is too small", cap);
  const size_t max_rec = (cap / 4 - 16 - zwords) / AMBREC_WORDS;
  // ONE pass: work items = contiguous node ranges of a layer; a thread collects the records of its range in a local block and copies
  // the block behind an atomic cursor.  The order of the records in the image is whatever the threads make it: the scatter does not care.
  struct Item { int k; long lo, hi; };
  std::vector<Item> items;
  const long chunk = 1L << 15;
  for (int k = 1; k <= L; ++k) {
    const long G = (long)B * h->N[k];
    for (long lo = 0; lo < G; lo += chunk) items.push_back(Item{k, lo, std::min(G, lo + chunk)});
  }
  int32_t* img = reinterpret_cast<int32_t*>(dst);
  int32_t* recs = img + 16;
  std::atomic<size_t> next{0}, cursor{0};
  std::atomic<int> overflow{0};
  const std::function<void()> work = [&]() {
    constexpr int BLK = 1024;
    int32_t local[BLK * AMBREC_WORDS];
    for (size_t it = next.fetch_add(1); it < items.size(); it = next.fetch_add(1)) {
      const Item& w = items[it];
      const int k = w.k, q = h->relu_q[k];
      const float *lb = in->lb[k], *ub = in->ub[k], *du = in->dual[k - 1], *zp = in->primal[q - 1], *zq = in->primal[q];
      int n = 0;
      auto flush = [&]() {
        if (!n) return;
        const size_t at = cursor.fetch_add((size_t)n);
        if (at + n > max_rec) overflow.store(1);
        else memcpy(recs + at * AMBREC_WORDS, local, (size_t)n * AMBREC_WORDS * 4);
        n = 0;
      };
      for (long g = w.lo; g < w.hi; ++g) {
        if (!(lb[g] < 0.0f && ub[g] > 0.0f)) continue;
        int32_t* r = local + n * AMBREC_WORDS;
        r[0] = k - 1; r[1] = (int32_t)g;
        memcpy(r + 2, du + g * 3 + 1, 8);
        memcpy(r + 4, zp + g, 4);
        memcpy(r + 5, zq + g, 4);
        if (++n == BLK) flush();
      }
      flush();
    }
  };
  if (items.size() >= 8) {
    if (h->work_pool && h->work_pool->owner != getpid()) h->work_pool = nullptr;      // (after a fork: the parent's threads are not here; its object is left alone)
    if (!h->work_pool) {                                 // helpers: what the machine (or the cgroup's CPU set) offers, a dozen threads with the caller at most
      const unsigned hc = std::thread::hardware_concurrency();
      cpu_set_t set;
      int avail = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)hc;
      if (avail < 1) avail = hc ? (int)hc : 1;
      h->work_pool = new WorkPool(std::max(0, std::min(avail, 12) - 1));
    }
    h->work_pool->run(work);
  } else {
    work();
  }
  const size_t total = cursor.load();
  if (overflow.load() || total > max_rec) return fail(GNNB_E_NOMEM, "gnnb_pack_amb_records: %zu records do not fit a buffer of %zu bytes", total, cap);
  memset(img, 0, 64);
  img[0] = AMBREC_MAGIC; img[1] = L; img[2] = (int32_t)total; img[3] = B;
  memcpy(recs + total * AMBREC_WORDS, in->primal[in->n_primal - 1], (size_t)B * sizeof(float));
  *used = (16 + total * AMBREC_WORDS + zwords) * 4;
  return GNNB_OK;
}

extern "C" int gnnb_scatter_amb_records(gnnb_t* h, const void* dev_image, int B, float* const* dual, int n_relu, float* const* primal, int n_primal,
                                        int32_t* status, void* stream) {
  if (!h || !dev_image || !dual || !primal) return fail(GNNB_E_INVALID, "gnnb_scatter_amb_records: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_scatter_amb_records: call gnnb_bind_network first");
  const int L = (int)h->N.size() - 2;
  if (B < 1 || L > MAXL || n_relu != L || n_primal != h->n_fixed + 1) return fail(GNNB_E_INVALID, "gnnb_scatter_amb_records: arrays do not match the bound network");
  ScatterArgs a{};
  a.image = reinterpret_cast<const int*>(dev_image); a.L = L; a.B = B;
  long total = B;
  for (int k = 1; k <= L; ++k) {
    const int q = h->relu_q[k];
    if (!dual[k - 1] || !primal[q - 1] || !primal[q]) return fail(GNNB_E_INVALID, "gnnb_scatter_amb_records: null array (layer %d)", k);
    a.dual[k - 1] = dual[k - 1]; a.z_pre[k - 1] = primal[q - 1]; a.z_post[k - 1] = primal[q];
    a.G[k - 1] = (long)B * h->N[k];
    total += (long)B * h->N[k];                          // (upper bound: the kernel reads the real counts from the image)
  }
  a.max_rec = total - B;
  a.status = status;
  if (!primal[n_primal - 1]) return fail(GNNB_E_INVALID, "gnnb_scatter_amb_records: null primals[-1]");
  a.z_out = primal[n_primal - 1];
  // (grid-stride: sized for an eighth of the nodes being ambiguous, correct for any share)
  hipLaunchKernelGGL(k_scatter_amb, dim3((unsigned)std::min<long>(1024, (total / 8 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "launch of k_scatter_amb failed: %s", hipGetErrorString(e));
  return GNNB_OK;
}

// BaBSR scores of a batch (reference plnn/kw_score_conv.py choose_node_conv :41-113; the decision rule :115-156 stays
// on the host).  lb/ub: HOST tables of n_graph DEVICE pointers exactly as in gnnb_batch; prop_w (B, N_L), mask (B, R),
// scores/intercepts (B, R) device.  Stream-ordered, no allocation.
extern "C" int gnnb_babsr(gnnb_t* h, const float* const* lb, const float* const* ub, int n_graph, const float* prop_w,
                          const float* mask, int B, float* scores, float* intercepts, void* stream) {
  if (!h || !lb || !ub || !prop_w || !mask || !scores || !intercepts) return fail(GNNB_E_INVALID, "gnnb_babsr: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_babsr: call gnnb_bind_network first");
  const int K = (int)h->N.size() - 1, L = K - 1;
  if (n_graph != K + 1 || B < 1) return fail(GNNB_E_INVALID, "gnnb_babsr: %d graph layers given, network has %d", n_graph, K + 1);
  BabsrArgs a{};
  a.L = L; a.R = h->R; a.prop_w = prop_w; a.mask = mask; a.scores = scores; a.icp = intercepts;
  int off = 0, maxN = 0;
  for (int k = 1; k <= L; ++k) {
    const int i = k - 1;
    if (!lb[k] || !ub[k]) return fail(GNNB_E_INVALID, "gnnb_babsr: null bounds pointer for graph layer %d", k);
    a.lb[i] = lb[k]; a.ub[i] = ub[k]; a.bias[i] = h->dev[k].bias; a.N[i] = h->N[k]; a.hw[i] = h->hw[k]; a.off[i] = off;
    off += h->N[k];
    maxN = std::max(maxN, h->N[k]);
    if (k < L) {                              // edge k+1 (between graph layers k and k+1)
      const Edge& e = h->edges[k + 1];
      a.ekind[i] = e.kind; a.ew[i] = h->dev[k + 1].w_bwd;
      a.c_in[i] = e.c_in; a.h_in[i] = e.h_in; a.w_in[i] = e.w_in; a.c_out[i] = e.c_out; a.h_out[i] = e.h_out; a.w_out[i] = e.w_out;
      a.kh[i] = e.kh; a.kw[i] = e.kw; a.stride[i] = e.stride; a.pad[i] = e.pad; a.ld[i] = h->dev[k + 1].ld_bwd;
    }
  }
  a.maxN = maxN;
  hipLaunchKernelGGL(k_babsr, dim3(B), dim3(256), (size_t)2 * maxN * sizeof(float), (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "launch of k_babsr failed: %s", hipGetErrorString(e));
  return GNNB_OK;
}


// ================================================================================================================
// Online learning (reference graphnet/graph_score_online.py:9-23, :62-77)
// ================================================================================================================
extern "C" int gnnb_get_weights(const gnnb_t* h, float* w_blob, size_t n_floats) {
  if (!h || !w_blob) return fail(GNNB_E_INVALID, "gnnb_get_weights: null argument");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_get_weights: %zu floats, expected %zu", n_floats, blob_floats());
  memcpy(w_blob, h->blob.data(), n_floats * sizeof(float));
  return GNNB_OK;
}

extern "C" int gnnb_set_weights(gnnb_t* h, const float* w_blob, size_t n_floats) {
  if (!h || !w_blob) return fail(GNNB_E_INVALID, "gnnb_set_weights: null argument");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_set_weights: %zu floats, expected %zu", n_floats, blob_floats());
  HIPCHK(hipDeviceSynchronize());          // no forward may still be reading the packs
  if (int rc = load_weights(h, w_blob, nullptr)) return rc;
  if (h->trainer) HIPCHK(hipMemcpy(h->trainer->d_w, w_blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
  return GNNB_OK;
}

// torch.optim.Adam(model.parameters(), lr, weight_decay) of graph_score_online.py:15; the moments start at zero
extern "C" int gnnb_online_create(gnnb_t* h, float lr, float weight_decay) {
  if (!h) return fail(GNNB_E_INVALID, "gnnb_online_create: null handle");
  free_trainer(h);
  gnnb_train::Trainer* t = new gnnb_train::Trainer();
  h->trainer = t;
  t->lr = lr; t->wd = weight_decay;
  const size_t n = blob_floats();
  for (float** p : {&t->d_w, &t->d_g, &t->d_m, &t->d_v}) {
    HIPCHK(hipMalloc((void**)p, n * sizeof(float)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(float)));
  }
  HIPCHK(hipMemcpy(t->d_w, h->blob.data(), n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipHostMalloc((void**)&t->desc, sizeof(gnnb_train::TChain) * gnnb_train::Trainer::kDescCap, hipHostMallocDefault));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd_multi<TL_ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd_multi<TL_ROWS_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd_multi<TL_ROWS_TINY>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd<TL_ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd<TL_ROWS_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tchain_fwd<TL_ROWS_TINY>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  return GNNB_OK;
}

extern "C" int gnnb_online_grad(const gnnb_t* h, float* grad, size_t n_floats) {
  if (!h || !grad || !h->trainer) return fail(GNNB_E_STATE, "gnnb_online_grad: no trainer (gnnb_online_create)");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_online_grad: %zu floats, expected %zu", n_floats, blob_floats());
  HIPCHK(hipMemcpy(grad, h->trainer->d_g, n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return GNNB_OK;
}

// One GraphChoice.online_learning step (graph_score_online.py:62-77) for B subproblems (the reference: B = 1):
//   loss = sum_b ( max_j scores_b[j] - scores_b[kw_b] + improvement_b );  backward;  Adam step;  scorer packs rebuilt.
// in: the batch exactly as for gnnb_forward.  kw_index (HOST, B): the KW decision as a flat index into the R ReLU nodes
// (trans_len[lay-1] + idx, :63-67), which must be an undecided node of the mask.  improvement (HOST, B).  loss (HOST, B,
// may be NULL).  scores_padded (DEVICE (B, R), may be NULL): the scores of the training-form forward BEFORE the update.
// apply = 0: gradient only (gnnb_online_grad), the parameters and the Adam state stay as they are.
extern "C" int gnnb_online_step(gnnb_t* h, const gnnb_batch* in, int B, const int32_t* kw_index, const float* improvement,
                                float* loss, float* scores_padded, int apply, void* stream) {
  using namespace gnnb_train;
  if (!h || !in || !kw_index || !improvement) return fail(GNNB_E_INVALID, "gnnb_online_step: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_online_step: call gnnb_bind_network first");
  if (!h->trainer) return fail(GNNB_E_STATE, "gnnb_online_step: call gnnb_online_create first");
  const int K = (int)h->N.size() - 1, L = K - 1, R = h->R, T = h->T;
  if (B < 1 || in->n_graph != K + 1 || in->n_relu != L || in->n_primal < h->n_fixed)
    return fail(GNNB_E_INVALID, "gnnb_online_step: batch does not match the bound network");
  for (int b = 0; b < B; ++b)
    if (kw_index[b] < 0 || kw_index[b] >= R) return fail(GNNB_E_INVALID, "gnnb_online_step: kw_index[%d] = %d outside [0, %d)", b, kw_index[b], R);
  Trainer& t = *h->trainer;
  hipStream_t st = (hipStream_t)stream;
  t.st = st;
  t.n_cu = h->n_cu;
  t.tape.clear();
  t.ndesc = 0;
  if (t.arena.reset(st)) return fail(GNNB_E_HIP, "gnnb_online_step: arena reset failed");
  if (t.edge_w.empty()) {                                  // torch-layout copies of the verified network's weights
    t.edge_w.assign(h->edges.size(), nullptr);
    for (int k = 1; k <= L; ++k)
      if (int rc = upload(&t.edge_w[k], h->edges[k].w.data(), h->edges[k].w.size())) return rc;
  }
  if (t.cap_B < B) {
    for (float** p : {&t.d_scores, &t.d_ds, &t.d_loss, &t.d_imp}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (t.d_kw) (void)hipFree(t.d_kw);
    if (t.d_sel) (void)hipFree(t.d_sel);
    HIPCHK(hipMalloc((void**)&t.d_sel, (size_t)B * 8));
    HIPCHK(hipMalloc((void**)&t.d_scores, (size_t)B * R * 4));
    HIPCHK(hipMalloc((void**)&t.d_ds, (size_t)B * R * 4));
    HIPCHK(hipMalloc((void**)&t.d_loss, (size_t)B * 4));
    HIPCHK(hipMalloc((void**)&t.d_imp, (size_t)B * 4));
    HIPCHK(hipMalloc((void**)&t.d_kw, (size_t)B * 4));
    t.cap_B = B;
  }
  HIPCHK(hipMemcpyAsync(t.d_kw, kw_index, (size_t)B * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.d_imp, improvement, (size_t)B * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(t.d_ds, 0, (size_t)B * R * 4, st));
  HIPCHK(hipMemsetAsync(t.d_g, 0, blob_floats() * 4, st));

  // ---- per-node constants ----
  struct LC { float *r0, *r1, *amb, *live, *nd2, *d1, *ff, *fb; Trainer::List ambl, livel; };
  std::vector<LC> lc(L + 1);
  if (L > T_MAXL) return fail(GNNB_E_INVALID, "gnnb_online_step: more than %d ReLU layers", T_MAXL);
  {
    TPrepMulti pm{};
    TCompactMulti cm{};
    long nmax = 0;
    for (int k = 1; k <= L; ++k) {
      const long n = (long)B * h->N[k];
      nmax = n > nmax ? n : nmax;
      LC& c = lc[k];
      for (float** p : {&c.r0, &c.r1, &c.amb, &c.live, &c.nd2, &c.d1}) *p = t.arena.alloc(n);
      c.ff = t.arena.alloc(7 * n);
      c.fb = t.arena.alloc(7 * n);
      // node lists: the relaxation chains run over the ambiguous nodes, the update chains over the live ones
      int* buf = reinterpret_cast<int*>(t.arena.alloc(2 * n + 2));
      if (t.arena.err || !buf) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
      const int q = h->relu_q[k];
      pm.a[k - 1] = TPrepArgs{in->lb[k], in->ub[k], in->dual[k - 1], in->primal[q - 1], in->primal[q], h->dev[k].bias, h->N[k], h->hw[k], n,
                              c.r0, c.r1, c.amb, c.live, c.nd2, c.d1, c.ff, c.fb};
      c.ambl = Trainer::List{buf, buf + 2 * n, n};
      c.livel = Trainer::List{buf + n, buf + 2 * n + 1, n};
      cm.a[2 * (k - 1)] = TCompact{c.amb, buf, buf + 2 * n, n};
      cm.a[2 * (k - 1) + 1] = TCompact{c.live, buf + n, buf + 2 * n + 1, n};
    }
    hipLaunchKernelGGL(k_tprep, dim3((unsigned)((nmax + 255) / 256), (unsigned)L), dim3(256), 0, st, pm);      // every layer, one launch
    hipLaunchKernelGGL(k_tcompact, dim3((unsigned)(2 * L)), dim3(256), 0, st, cm);                              // every list, one launch
  }
  const long n0 = (long)B * h->N[0];
  const float *inp3, *inp2, *featp;
  {
    TColsMulti m{};
    auto job = [&](int j, std::initializer_list<const float*> cs, long n) {
      TColsArgs& a = m.a[j];
      int w = 0;
      for (const float* c : cs) a.c[w++] = c;
      a.w = w; a.n = n; a.dst = t.arena.alloc((size_t)n * w);
      return (const float*)a.dst;
    };
    inp3 = job(0, {in->lb[0], in->x_lp, in->ub[0]}, n0);                                                    // graph_conv.py:90-93
    inp2 = job(1, {in->lb[0], in->ub[0]}, n0);                                                              // :380-381
    featp = job(2, {in->lb[K], in->ub[K], in->primal[in->n_primal - 1], in->prop_b}, B);                   // :202-205
    hipLaunchKernelGGL(k_tcols, dim3((unsigned)(((n0 > B ? n0 : B) + 255) / 256), 3), dim3(256), 0, st, m);
  }

  auto edge = [&](int k, int dir, int norm, const TT& src) {       // nb = A_k src (dir 0) or A_k^T src (dir 1, / tap count if norm)
    const Edge& e = h->edges[k];
    TT y = t.rows((long)B * (dir == 0 ? h->N[k] : h->N[k - 1]));
    if (e.kind == 0) {
      TConv a{src.v, y.v, t.edge_w[k], B, e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, dir, norm, 0};
      hipLaunchKernelGGL(k_tconv, dim3((unsigned)((y.n + 3) / 4)), dim3(256), 0, st, a);
      TConv b = a;
      b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
      const long nsrc = src.n;
      t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tconv, dim3((unsigned)((nsrc + 3) / 4)), dim3(256), 0, st, b); });
    } else {
      TDense a{t.edge_w[k], 0, src.v, y.v, B, e.n_out, e.n_in, dir, 0};
      hipLaunchKernelGGL(k_tdense, dim3((unsigned)y.n), dim3(TD_WAVES * 64), 0, st, a);
      TDense b = a;
      b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
      const long nsrc = src.n;
      t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tdense, dim3((unsigned)nsrc), dim3(TD_WAVES * 64), 0, st, b); });
    }
    return y;
  };
  auto prop_edge = [&](int dir, const TT& src) {                   // the property layer: one (1, N_L) matrix per sample
    TT y = t.rows(dir == 0 ? (long)B : (long)B * h->N[L]);
    TDense a{in->prop_w, (long)h->N[L], src.v, y.v, B, 1, h->N[L], dir, 0};
    hipLaunchKernelGGL(k_tdense, dim3((unsigned)y.n), dim3(TD_WAVES * 64), 0, st, a);
    TDense b = a;
    b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
    const long nsrc = src.n;
    t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tdense, dim3((unsigned)nsrc), dim3(TD_WAVES * 64), 0, st, b); });
    return y;
  };
  auto S = [](const TT& x, const float* s = nullptr) { return Trainer::seg(x, s); };
  auto SF = [](const TT& x, const float* s = nullptr) { return Trainer::seg(x, s, true); };     // a segment addressed by node
  auto P = [](const float* s = nullptr) { return Trainer::prev(s); };                            // the previous op of the chain
  using Spec = Trainer::Spec;

  // ---- relaxation terms (graph_conv.py:153-161, :273-293): functions of the node features only, so the same in every round
  // -- computed once (the reference recomputes them per round; their gradient contributions from all rounds add up in
  // relax.g before the chain is walked back once) and only for the ambiguous nodes (`* amb` zeroes every other row).
  // Each MLP chain is one launch (Trainer::chain).
  std::vector<TT> relax_f(L + 1), relax_b(L + 1);
  {
    std::vector<Trainer::Job> jf, jb;
    for (int k = 1; k <= L; ++k) {
      const LC& c = lc[k];
      const long n = (long)B * h->N[k];
      jf.push_back({{Spec{L_FC1, {}, c.ff, true, nullptr, false},
                     Spec{L_FC1_1, {P()}, nullptr, false, c.amb, true}}, n, &c.ambl});                            // :160-161
      jb.push_back({{Spec{L_BC1, {}, c.fb, true, nullptr, false},
                     Spec{L_BC1_1, {P()}, nullptr, true, nullptr, false},
                     Spec{L_BC1_2, {P()}, nullptr, false, nullptr, false},                                        // :285
                     Spec{L_BC2, {P(), P(c.nd2), P(c.d1)}, nullptr, true, nullptr, false},                        // :287-291
                     Spec{L_BC2_1, {P()}, nullptr, false, c.amb, true}}, n, &c.ambl});                            // :293
    }
    auto of = t.chain_multi(jf);         // every layer's chain in one launch
    auto ob = t.chain_multi(jb);
    for (int k = 1; k <= L; ++k) { relax_f[k] = of[k - 1][1]; relax_b[k] = ob[k - 1][4]; }
  }

  // ---- the forward of graph_conv.py:77-388, every Linear on the tape ----
  std::vector<TT> mu(K + 1);
  for (int r = 0; r < T; ++r) {
    if (r == 0)
      mu[0] = t.chain({Spec{L_INP_F, {}, inp3, true, nullptr, false}, Spec{L_INP_F_1, {P()}, nullptr, false, nullptr, false}}, n0)[1];   // :94
    for (int k = 1; k <= L; ++k) {                                                       // :107-192
      const LC& c = lc[k];
      const long n = (long)B * h->N[k];
      TT nb = edge(k, 0, 0, mu[k - 1]);
      // the update chain over the live nodes only (`* live` zeroes the rows of the others, :178)
      mu[k] = t.chain({Spec{L_FC3, {SF(nb, c.r0), SF(nb, c.r1)}, nullptr, true, nullptr, false},                  // :169-170
                       Spec{L_FC3_2, {P()}, nullptr, false, nullptr, false},
                       Spec{L_FC4, {SF(relax_f[k]), P()}, nullptr, true, nullptr, false},                         // :176-177
                       Spec{L_FC4_2, {P()}, nullptr, false, c.live, true}}, n, &c.livel)[3];                      // :178
    }
    {                                                                                    // :194-210
      TT nb = prop_edge(0, mu[L]);
      mu[K] = t.chain({Spec{L_OUT1, {}, featp, true, nullptr, false},
                       Spec{L_OUT2, {P(), S(nb)}, nullptr, true, nullptr, false},
                       Spec{L_OUT3, {P()}, nullptr, false, nullptr, false}}, B)[2];
    }
    for (int k = L; k >= 1; --k) {                                                       // :222-350
      const LC& c = lc[k];
      const long n = (long)B * h->N[k];
      TT nb = k == L ? prop_edge(1, mu[K]) : edge(k + 1, 1, h->edges[k + 1].kind == 0 ? 1 : 0, mu[k + 1]);   // :299-326
      mu[k] = t.chain({Spec{L_BC3, {SF(nb, c.r0), SF(nb, c.r1)}, nullptr, true, nullptr, false},                  // :331-336
                       Spec{L_BC3_1, {P()}, nullptr, false, nullptr, false},
                       Spec{L_BC4, {SF(relax_b[k]), P()}, nullptr, true, nullptr, false},                         // :344-345
                       Spec{L_BC4_1, {P()}, nullptr, false, c.live, true}}, n, &c.livel)[3];                      // :347
    }
    if (r + 1 < T) {                                                                     // :360-385 (the last round's input rows feed nothing)
      TT nb = edge(1, 1, 0, mu[1]);
      mu[0] = t.chain({Spec{L_INP_B, {}, inp2, true, nullptr, false},
                       Spec{L_INP_B_1, {P()}, nullptr, false, nullptr, false},
                       Spec{L_INP_B2, {P(), S(nb)}, nullptr, true, nullptr, false},
                       Spec{L_INP_B2_2, {P()}, nullptr, false, nullptr, false}}, n0)[3];
    }
  }
  // ---- scores (:442-450) and the loss ----
  {
    std::vector<Trainer::Job> jn;
    for (int k = 1; k <= L; ++k) jn.push_back({{Spec{L_FNODE, {S(mu[k])}, nullptr, true, nullptr, false}}, (long)B * h->N[k], nullptr});
    auto hk = t.chain_multi(jn);
    TScoreMulti sm{};
    int off = 0;
    long nmax = 0;
    for (int k = 1; k <= L; ++k) {
      const long n = (long)B * h->N[k];
      nmax = n > nmax ? n : nmax;
      sm.a[k - 1] = TScore{hk[k - 1][0].v, hk[k - 1][0].g, t.d_w + weight_offset(L_FSCORE), t.d_w + bias_offset(L_FSCORE), in->mask, t.d_scores,
                           t.d_ds, h->N[k], R, off, n, t.d_g + weight_offset(L_FSCORE), t.d_g + bias_offset(L_FSCORE), t.d_sel, B};
      off += h->N[k];
    }
    const dim3 grid((unsigned)((nmax + 3) / 4), (unsigned)L);
    hipLaunchKernelGGL(k_tscore_fwd, grid, dim3(256), 0, st, sm);
    t.tape.push_back([sm, grid, L, st]() {
      hipLaunchKernelGGL(k_tscore_bwd, grid, dim3(256), 0, st, sm);
      hipLaunchKernelGGL(k_tscore_bwd_w, dim3(1), dim3(64), 0, st, sm, L);
    });
  }
  if (t.arena.err) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
  if (scores_padded) HIPCHK(hipMemcpyAsync(scores_padded, t.d_scores, (size_t)B * R * 4, hipMemcpyDeviceToDevice, st));
  TLoss la{t.d_scores, t.d_ds, t.d_kw, t.d_imp, t.d_loss, R, t.d_sel};
  hipLaunchKernelGGL(k_tloss, dim3(B), dim3(256), 0, st, la);
  // ---- backward: the tape in reverse ----
  t.wops.clear();
  for (auto it = t.tape.rbegin(); it != t.tape.rend(); ++it) (*it)();
  t.tape.clear();
  if (t.weight_grads()) return fail(GNNB_E_HIP, "gnnb_online_step: the weight-gradient launches failed");
  if (t.arena.err) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "gnnb_online_step: a launch failed: %s", hipGetErrorString(e));
  if (loss) {
    t.h_loss.resize(B);
    HIPCHK(hipMemcpyAsync(t.h_loss.data(), t.d_loss, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  }
  if (apply) {
    t.step += 1;
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - std::pow(b1, t.step), bc2 = 1.0 - std::pow(b2, t.step);
    TAdam a{t.d_w, t.d_g, t.d_m, t.d_v, (int)blob_floats(), (float)(t.lr / bc1), t.wd, (float)b1, (float)b2, 1e-8f, (float)std::sqrt(bc2)};
    hipLaunchKernelGGL(k_tadam, dim3((unsigned)((blob_floats() + 255) / 256)), dim3(256), 0, st, a);
    std::vector<float> nw(blob_floats());
    HIPCHK(hipMemcpyAsync(nw.data(), t.d_w, nw.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (int rc = load_weights(h, nw.data(), st)) return rc;      // the scorer's folded packs follow the new parameters
  } else {
    HIPCHK(hipStreamSynchronize(st));
  }
  if (loss) memcpy(loss, t.h_loss.data(), (size_t)B * 4);
  return GNNB_OK;
}
