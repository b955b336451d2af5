/*
 * gnnb.h -- C-ABI of libgnnb.so: the MI355X (gfx950) GNN branching-score forward pass.
 *
 * The reference (oval-group/GNN_branching) is pure Python/PyTorch and has NO native
 * interface for this path; the entry points below are what a binding for the hot path
 * would call.  Each one cites the reference code it replaces.  Plain pointers and sizes
 * only -- no torch types.  The Python host side (gnn_branching_amd/graphnet) binds them
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative GNNB_E_* code and never
 * throws; gnnb_last_error() returns a thread-local message for the last failure.
 * gnnb_forward is stream-ordered and asynchronous, allocates nothing and never
 * synchronises; one handle per host thread (no entry point is re-entrant on one handle).  No HIP
 * call happens at library load time (the reference creates its GPU context inside a forked child:
 * experiments/bab_mip.py:244-249).  The library reads NO environment variable: every switch is a
 * handle option (gnnb_set_option).
 */
#ifndef GNNB_H
#define GNNB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNB_ABI_VERSION 2

enum {
  GNNB_OK = 0,
  GNNB_E_INVALID = -1,   /* bad argument / unsupported shape */
  GNNB_E_HIP = -2,       /* a HIP runtime call failed */
  GNNB_E_STATE = -3,     /* call order (e.g. forward before bind_network) */
  GNNB_E_NOMEM = -4      /* workspace too small */
};

/* layer kinds of the verified network's layer list (reference graph_conv.py:110,130,139,188
 * dispatches on type(layer) is nn.Conv2d / nn.Linear / nn.ReLU / Flatten) */
enum { GNNB_CONV = 0, GNNB_LINEAR = 1, GNNB_RELU = 2, GNNB_FLATTEN = 3 };

typedef struct gnnb_handle gnnb_t;

/* One entry of layers['fixed_layers'] (reference plnn/relu_conv_gnnkwthreshold.py:111-112).
 * weight/bias are HOST pointers in torch's row-major layout (conv: [c_out][c_in][kh][kw],
 * linear: [n_out][n_in]); they are copied, the caller keeps ownership. */
typedef struct {
  int32_t kind;
  int32_t c_in, c_out, kh, kw, stride, pad;   /* GNNB_CONV  (square stride/pad, dilation 1, groups 1) */
  int32_t n_in, n_out;                         /* GNNB_LINEAR */
  const float* weight;
  const float* bias;
} gnnb_layer_desc;

/* One batch of B subproblems = the tensor arguments of GraphNet.forward
 * (reference graph_conv.py:479) as DEVICE pointers to contiguous fp32 arrays.
 * The pointer tables themselves (lb, ub, dual, primal) are HOST arrays. */
typedef struct {
  const float* const* lb;      /* n_graph ptrs: lower_bounds_all[k], (B, N_k)                */
  const float* const* ub;      /* n_graph ptrs: upper_bounds_all[k]                          */
  const float* const* dual;    /* n_relu  ptrs: dual_vars[j], (B*N_{j+1}, 3)                 */
  const float* const* primal;  /* n_primal ptrs: primals[m], (B*n_m), one per net.layers[m]  */
  const float* x_lp;           /* primal_inputs (B, N_0)                                     */
  const float* prop_w;         /* layers['prop_layers'][b].weight, (B, N_L)                  */
  const float* prop_b;         /* layers['prop_layers'][b].bias,   (B)                       */
  const float* mask;           /* masks (B, R): 1.0 where the BaB mask is -1                 */
  int32_t n_graph, n_relu, n_primal;
} gnnb_batch;

/* GraphNet(T, p) + load_state_dict (reference graph_score.py:9-13, graph_conv.py:22-74,
 * :421-432): w_blob = the 52 tensors of the checkpoint concatenated in state-dict order
 * (weight (out,in) row-major, then bias), 117 825 floats for T=2, p=64.  HOST pointer. */
int gnnb_create(gnnb_t** out, const float* w_blob, size_t n_floats, int T, int p);

/* Handle options.  The reference has one implementation of the path and therefore no switches (graph_conv.py:77-388); here an option
 * selects between kernels that compute the same scores, which is what lets the parity tests check them against each other and lets a
 * caller that shares the GPU (two batches in flight, a collective on another stream) turn off what needs co-resident workgroups.
 * Set options after gnnb_create; the ones marked (*) shape the tables of gnnb_bind_network and return GNNB_E_STATE on a bound handle.
 *   "bf3"          1 (default): 64x64 node-MLP blocks on the bf16 matrix rate, both operands in three bf16 pieces, fp32 accumulate
 *                  (fp32-grade); 0: every block on the exact-fp32 MFMA (and the separate kernels instead of k_top)
 *   "fuse"         1: a conv half-pass (graph_conv.py:110-181, :299-349) is ONE kernel; 0: aggregate kernel + node-update kernel
 *   "top"          1: last Linear edge, last ReLU layer, property node (graph_conv.py:130-137, :194-210, :320-326) in k_top; 0: separate kernels
 *   "gather" (*)   1: conv edges as MFMA tap blocks; 0: the scalar conv kernels
 *   "embed_fuse"   1: round 0's input embedding (graph_conv.py:90-95) computed inside the first aggregate; 0: written by k_embed
 *   "dense_lds" (*) 1: Linear edges one workgroup per sample out of LDS; 0: the per-tile kernel (what very wide layers fall back to)
 *   "tail_max_b"   batches up to it run the last restricted update + score head (graph_conv.py:442-470) as ONE launch; 0: three kernels
 *   "top_split"    1 / 2 / 4: most workgroups k_top spreads one sample over.  2 and 4 make workgroups WAIT for partner workgroups and
 *                  need all of them resident: set 1 when anything else may occupy CUs while a forward runs (status bit 1 otherwise)
 *   "top_fuse_upd" 1: the backward update of layer L-1 (graph_conv.py:253-350) inside k_top; 0: its own launch
 *   "clspre_max_b" batches up to it classify nodes and run the hoisted feature chains in one launch
 * gnnb_option_count / gnnb_option_name enumerate the table; gnnb_get_option reads a value back. */
int gnnb_set_option(gnnb_t* h, const char* name, int value);
int gnnb_get_option(const gnnb_t* h, const char* name, int* value);
int gnnb_option_count(void);
const char* gnnb_option_name(int i);

/* The verified network's fixed layers, i.e. the static part of the `layers` argument
 * (reference relu_conv_gnnkwthreshold.py:110-113; graph structure walked at
 * graph_conv.py:107-192 and :222-385).  (c0,h0,w0) = input tensor shape. */
int gnnb_bind_network(gnnb_t* h, const gnnb_layer_desc* layers, int n_layers, int c0, int h0, int w0);

/* Sizes of the bound layer graph: n_graph = L+2 graph layers, sizes[k] = N_k, *n_relu_total = R. */
int gnnb_graph_info(const gnnb_t* h, int* n_graph, int* sizes /* >= n_graph ints or NULL */, int* n_relu_total);

/* Bytes of device scratch gnnb_forward needs for a batch of B (embeddings `mu`, init_mu
 * graph_conv.py:487-496, plus aggregation and cached feature terms). */
size_t gnnb_workspace_bytes(const gnnb_t* h, int B);

/* GraphNet.forward + the argmax of GraphChoice.decision (reference graph_conv.py:479-483,
 * graph_score.py:32-47).  scores_padded: device (B, R) fp32, score of every ambiguous ReLU in
 * flat ReLU order, -inf elsewhere (the reference returns the ragged list of graph_conv.py:470).
 * decisions: device (B, 2) int32 [dec_lay, dec_idx], first maximal score; [-1,-1] if a sample
 * has no ambiguous ReLU.  status: device int32[1], bit 0 set if an embedding was NaN (the reference
 * enters pdb, graph_conv.py:184-186, :339-341), bit 1 if a wait inside a kernel (k_gather_update_q's LDS ring; k_top's workgroup
 * split waiting for its partner workgroups -- option "top_split" = 1 turns that split off) hit its iteration cap (never in a correct run on a
 * GPU the caller does not share with long-running kernels; the results are then invalid).  stream: hipStream_t (NULL = default). */
int gnnb_forward(gnnb_t* h, const gnnb_batch* in, int B, float* scores_padded, int32_t* decisions,
                 int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* gnnb_forward for HOST inputs -- the reference's own call pattern (relu_conv_gnnkwthreshold.py:117, :230, :239: one
 * subproblem per graph.decision call, every argument a CPU tensor that graph_score.py:26-30 moves with ~14 .cuda() calls).
 * `in` holds HOST pointers laid out exactly as for gnnb_forward.  The inputs cross PCIe as ONE pinned copy, the forward runs
 * on `stream`, and decisions (B, 2), status (1) and -- if scores is not NULL -- the padded scores (B, R) are written to HOST
 * memory; the call returns after synchronising `stream`.  Staging buffers and the workspace belong to the handle. */
int gnnb_forward_host(gnnb_t* h, const gnnb_batch* in, int B, float* scores, int32_t* decisions, int32_t* status, void* stream);

/* Host-fed batches, fewer bytes over the link (reference graph_score.py:26-30 moves EVERY tensor whole).  Of `dual_vars` and `primals` the
 * forward reads only the entries of AMBIGUOUS ReLU nodes (the relaxation terms of graph_conv.py:153-161 and :273-293 are multiplied by
 * amb = 0 everywhere else) and the network output primals[-1]: 4 floats for ~7 % of the nodes instead of 5 for all of them.
 *   gnnb_amb_records_bytes   upper bound of the record image of a batch of B (every node ambiguous)
 *   gnnb_pack_amb_records    HOST: `in` holds HOST pointers laid out as for gnnb_forward (only lb / ub of the ReLU layers, dual, primal
 *                            are read).  Writes to `dst` (host memory, e.g. pinned; >= cap bytes) for every ReLU layer the nodes with
 *                            lb < 0 < ub -- a superset of the nodes the device classifies as ambiguous -- as records {layer, flat index
 *                            b N_k + n, dual[:, 1], dual[:, 2], primal_pre, primal_post} in no particular order, then primals[-1];
 *                            *used = bytes written.  Runs on up to a dozen helper threads that belong to the handle (created on first
 *                            use, joined by gnnb_destroy); calls on one handle are serialised, every tensor's element count is the
 *                            caller's responsibility (B N_k per bound / primal tensor, 3 B N_k per dual tensor of the BOUND network).
 *   gnnb_scatter_amb_records DEVICE: one launch on `stream` that writes the records of an image copied to device memory into full-size
 *                            device arrays dual[k] (B N_k, 3) / primal[m] laid out as gnnb_forward expects them; entries of other nodes
 *                            are left as they are (the forward never reads them).  Then call gnnb_forward on those arrays: scores are
 *                            the bits of a forward on the whole tensors.  status: device int32[1] the caller zeroed, or NULL; bit 2
 *                            (value 4) is set -- and nothing, or not that record, is written -- when the image's header does not match
 *                            this binding and B, or a record points outside its arrays (a stale or foreign image). */
size_t gnnb_amb_records_bytes(const gnnb_t* h, int B);
int gnnb_pack_amb_records(const gnnb_t* h, const gnnb_batch* in, int B, void* dst, size_t cap, size_t* used);
int gnnb_scatter_amb_records(gnnb_t* h, const void* dev_image, int B, float* const* dual, int n_relu, float* const* primal, int n_primal,
                             int32_t* status, void* stream);

/* BaBSR ("KW") branching heuristic for a batch -- the fallback scorer of the BaB loop (reference
 * plnn/kw_score_conv.py choose_node_conv :41-113, called at plnn/relu_conv_gnnkwthreshold.py:157).  lb/ub: HOST tables of
 * n_graph DEVICE pointers laid out like struct gnnb_batch.lb, .ub -- only the ReLU layers 1..L are read; prop_w (B, N_L); mask (B, R) 1.0 where the
 * BaB mask is -1.  Outputs, device (B, R): scores = `score` (:103), intercepts = `intercept_tb` (:86), both already
 * multiplied by the mask.  The decision rule (:115-156, with its counters and random fall-back) stays on the host. */
int gnnb_babsr(gnnb_t* h, const float* const* lb, const float* const* ub, int n_graph, const float* prop_w,
               const float* mask, int B, float* scores, float* intercepts, void* stream);

int gnnb_destroy(gnnb_t* h);

/* ---- online learning (reference graphnet/graph_score_online.py) ----
 * gnnb_get_weights / gnnb_set_weights: the 117 825 GNN parameters in gnnb_create's blob order (HOST) -- the
 * state_dict()/load_state_dict() of the model GraphChoice holds (graph_score_online.py:11-14).  set_weights rebuilds the
 * scorer's operand packs (device-synchronising). */
int gnnb_get_weights(const gnnb_t* h, float* w_blob, size_t n_floats);
int gnnb_set_weights(gnnb_t* h, const float* w_blob, size_t n_floats);

/* torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd) of graph_score_online.py:15 (betas 0.9/0.999, eps 1e-8,
 * moments start at zero).  Calling it again resets the optimizer state. */
int gnnb_online_create(gnnb_t* h, float lr, float weight_decay);

/* GraphChoice.online_learning (graph_score_online.py:62-77) for B subproblems (the reference: B = 1; B > 1 sums the B
 * losses):  loss_b = max_j scores_b[j] - scores_b[kw_b] + improvement_b;  backward through GraphNet.forward;  one Adam step;
 * the scorer (gnnb_forward) uses the new parameters from the next call on.
 * in: the batch as for gnnb_forward (device pointers).  kw_index: HOST (B), the KW decision as a flat index into the R ReLU
 * nodes (trans_len[lay-1] + idx, :63-67) -- must be an undecided node of the mask.  improvement: HOST (B).  loss: HOST (B)
 * or NULL.  scores_padded: DEVICE (B, R) or NULL, the scores of the training-form forward before the update.  apply = 0:
 * compute the gradient only (read it with gnnb_online_grad), parameters and optimizer state untouched.
 * Synchronises `stream` before returning. */
int gnnb_online_step(gnnb_t* h, const gnnb_batch* in, int B, const int32_t* kw_index, const float* improvement,
                     float* loss, float* scores_padded, int apply, void* stream);

/* d loss / d parameters of the last gnnb_online_step (before weight decay), HOST, blob order. */
int gnnb_online_grad(const gnnb_t* h, float* grad, size_t n_floats);

const char* gnnb_last_error(void);
int gnnb_abi_version(void);
/* 32 hex digits: hash of the sources (and compiler flags) the library was built from; the Python loader compares it with the
 * tree's and refuses a stale binary. */
const char* gnnb_build_id(void);

/* ---- inspection hooks used by the parity tests and bench.py (not needed by a caller) ---- */

/* JSON text describing the launch plan of one forward for the bound network: per half-pass update the
 * kernel used, the tile shape of the MFMA gather (channels x pixel block, window, k-steps) and node counts. */
int gnnb_describe(const gnnb_t* h, char* buf, size_t cap);

/* Location of embedding mu[k] inside the workspace: row-major (B, N_k, p) fp32. */
int gnnb_mu_location(const gnnb_t* h, int B, int k, size_t* offset_bytes, size_t* n_floats);

/* Inspection: some producers store their embedding rows before their last Linear layer (the projection is folded into
 * the consumer): the rows of mu[k] left by the last forward are E with mu = W.E + b for Linear
 * `*linear_id` (index into the checkpoint's 26 Linear layers, state-dict order), or final when *linear_id = -1. */
int gnnb_mu_projection(const gnnb_t* h, int k, int* linear_id);

/* Stop after `n` half-passes (1 = round-0 forward sweep, 2 = + round-0 backward sweep, ...;
 * <= 0 = run everything).  With a limit set the scores are computed from the embeddings so far. */
int gnnb_set_halfpass_limit(gnnb_t* h, int n);

/* Occupy n_workgroups CUs (one workgroup per CU when lds_bytes > 80 KiB) for `ms` (<= 500) milliseconds on `stream` with a kernel that only
 * watches the clock: the test stand-in for "something else holds CUs while a forward runs" (a collective, a second batch). */
int gnnb_debug_occupy(int n_workgroups, int threads, size_t lds_bytes, double ms, void* stream);

/* Per-kernel-class timing with HIP events recorded on the launch stream.  When enabled every
 * launch in gnnb_forward is bracketed by a pair of events; gnnb_profile_read synchronises the
 * stream, accumulates and returns per class: total ms and launch count since the last reset. */
int gnnb_profile_enable(gnnb_t* h, int on);
int gnnb_profile_classes(void);                          /* number of classes */
const char* gnnb_profile_class_name(int cls);            /* kernel (class) name */
int gnnb_profile_read(gnnb_t* h, double* total_ms, int64_t* launches, int n, int reset);
/* The launches resolved by gnnb_profile_read since the previous call of this function, in launch order: class index and duration
 * (ms) of each.  Returns their number (>= 0; at most cap entries are written), or -1 for a null handle; clears the list. */
int gnnb_profile_trace(gnnb_t* h, int* cls, double* ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GNNB_H */
