/*
 * pastml_hip.h -- C-ABI of libpastml_hip.so: the MI355X (gfx950) implementation of PastML's maximum-likelihood ACR
 * hot path.  Plain pointers and sizes only; the caller owns every host buffer, the library owns the device buffers
 * behind the opaque pml_ctx.  Every function returns a status (0 = PML_OK); pml_last_error() gives the message of the
 * last failure on the calling thread.  A ctx may be used from one thread at a time; different ctxs are independent
 * (each has its own HIP stream), which is how pastml.acr.acr()'s per-character thread pool (pastml/acr.py:226-231)
 * maps onto the device.
 *
 * The reference (evolbioinfo/pastml) is pure Python and has no FFI: each entry point below names the reference
 * function(s) whose arithmetic it replaces (paths relative to the reference repository).
 *
 * Layout conventions
 *   nodes     forest-wide level (breadth-first) order: roots are ids 0..n_roots-1, the children of a node are
 *             contiguous (first_child[n] .. first_child[n]+n_children[n]-1), every depth is a contiguous id range
 *   columns   the batch axis: one column = one character (with its own masks and model parameters); all columns of
 *             a ctx share the tree, the number of states k and the model kind
 *   masks     allowed-state bit sets, W = (k+63)/64 uint64 words per (column, node), bit s of word s/64 = state s
 *   vectors   double[k] per (column, node), column-major: index ((col * n_nodes + node) * k + state)
 *   scales    every likelihood vector v carries a base-2 exponent E (true value = v * 2^E); the *_sf outputs are
 *             converted to the reference's convention (base-10, true = v / 10^sf, pastml/ml.py:121,148)
 */
#ifndef PASTML_HIP_H
#define PASTML_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pml_ctx pml_ctx;

enum {
    PML_OK = 0,
    PML_ERR_INVALID = 1,      /* bad argument / call order */
    PML_ERR_HIP = 2,          /* HIP runtime failure (message has the HIP error string) */
    PML_ERR_UNSUPPORTED = 3,  /* e.g. k > 512 (F81 family), k > 256 (HKY / eigen models) */
    PML_ZERO_LIKELIHOOD = 4   /* pastml/ml.py:139-145: a parent/child pair has non-intersecting states */
};

enum { PML_MODEL_F81 = 0, PML_MODEL_HKY = 1, PML_MODEL_EIGEN = 2 };

/* what pml_download() can fetch (debug / tests / host-side state selection) */
enum {
    PML_BUF_BU = 0,          /* double[n_nodes][k]  bottom-up vectors of one column (tips: their masks as 0/1)      */
    PML_BUF_BU_SF = 1,       /* double[n_nodes]     their base-10 scale (reference BU_LH_SF convention)              */
    PML_BUF_TD = 2,          /* double[n_nodes][k]  top-down vectors of all nodes (pastml/ml.py:273-290)             */
    PML_BUF_TD_SF = 3,       /* double[n_nodes]                                                                        */
    PML_BUF_POSTERIOR = 4,   /* double[n_nodes][k]  marginal posteriors                                                */
    PML_BUF_LH_SUM = 5,      /* double[n_nodes]     sum of the (scaled) marginal likelihoods, in [1, 2)                */
    PML_BUF_LH_SF = 6,       /* double[n_nodes]     base-10 scale of the marginal likelihoods (reference LH_SF)        */
    PML_BUF_JOINT_TABLE = 7, /* int32[n_nodes][k]   argmax tables of the joint sweep (roots: undefined)               */
    PML_BUF_JOINT_STATE = 8, /* int32[n_nodes]                                                                          */
    PML_BUF_BRANCH_EXP = 9   /* double[n_nodes]     F81 family: e = exp(-mu t') per branch                             */
};

const char* pml_last_error(void);
int pml_version(void);
/* first 16 hex digits of the sha256 over the sources (everything under pastml_amd/csrc and this header) the library was compiled from --
 * pastml_amd/build.py hands it to the compiler and rebuilds when the sources in the tree give another one;
 * "unknown" for a build made by hand without -DPML_BUILD_DIGEST */
const char* pml_build_digest(void);
int pml_device_count(int* count);

/* ---- context ------------------------------------------------------------------------------------------------- */
int pml_ctx_create(int device, pml_ctx** out);
int pml_ctx_destroy(pml_ctx* ctx);
int pml_ctx_sync(pml_ctx* ctx);
/*
 * Options (set before pml_tree_upload).  PML_OPT_CHERRY_FUSION (default 1): in the F81-family marginal sweeps,
 * internal nodes whose children are all tips are recomputed in registers instead of being stored in HBM (their
 * vectors, like the top-down vectors of tips, are computed when pml_download asks for them).  With it, on large
 * forests and 17 <= k <= 64, nodes with two stored children that each carry two cherries of two tips run as two-level
 * units, and above them nodes take pairs of plain children over as stacked units: those children's
 * bottom-up vectors are not stored either, same rule for downloads.
 * PML_OPT_KEEP_TD (default 0, may be changed at any time): the F81-family top-down sweep works from the stored
 * posteriors of the level above (TD o BU = posterior * sum / pi) and does not write the top-down vectors themselves;
 * with this option it stores them too.  pml_download(PML_BUF_TD / PML_BUF_TD_SF) repeats the sweep with the stores on
 * when they are missing and fills in the vectors no sweep stores (tips, fused cherries).  The matrix models (HKY,
 * eigen) always store the vectors of internal nodes.
 * PML_OPT_EIGEN_FUSED (default 1, may be changed at any time): eigen-decomposed models with 16 <= k <= 32 run the fused
 * matrix-core sweeps (P(t) built and consumed in registers); 0 selects the sweeps that read materialised P(t) from HBM.
 * PML_OPT_EIGEN_JOINT_VALU (default 1, may be changed at any time): the JOINT sweep of eigen-decomposed models with
 * k <= 32 runs on the FP64 vector units (the default: faster than the matrix cores for this sweep);
 * 0 sends it to the kernels PML_OPT_EIGEN_FUSED selects -- kept as the cross-check of the default path.
 * PML_OPT_IMPLICIT_TIP_POSTERIORS (default 0, may be changed at any time; F81 family, k <= 64): the top-down sweep does
 * not write the posterior rows of observed tips -- unit vectors, a third of the sweep's traffic on a binary tree; they
 * are written when the table is read (pml_download*, posterior_out, pml_select_states, pml_marginal_counts) or before
 * the masks that define them change.  The sums / exponents of every node are written as always.
 */
enum { PML_OPT_CHERRY_FUSION = 1, PML_OPT_KEEP_TD = 2, PML_OPT_EIGEN_FUSED = 3, PML_OPT_EIGEN_JOINT_VALU = 4,
       PML_OPT_IMPLICIT_TIP_POSTERIORS = 5 };
int pml_ctx_set_option(pml_ctx* ctx, int option, int value);
/*
 * The switches of the schedules (tuning and test variants; DESIGN.md section 6a lists them).  Every ctx has its own set:
 * the defaults are the environment variables PASTML_HIP_<NAME> as they stand when the ctx is created, and this call
 * overrides one of them for this ctx -- name with or without the PASTML_HIP_ prefix; is_set = 0 returns the switch to
 * "not given" (the built-in default), otherwise value is what the environment variable would have held (for the NO_* /
 * on-off switches: value != 0 = on).  Switches that shape what pml_tree_upload / pml_chars_alloc build (NO_SUPER,
 * SUPER_MIN, STACK_MIN, NO_STACK, BLOCK_NODES, SMALL_MAX_NODES, NO_SHAPE_SORT, F81_R, ...) must be set before the tree
 * is uploaded (PML_ERR_INVALID afterwards); the others take effect with the next sweep (captured launch sequences are
 * dropped).  Unknown names are PML_ERR_INVALID.  There is no process-wide state: two contexts may run different schedules.
 */
int pml_ctx_set_tunable(pml_ctx* ctx, const char* name, int64_t value, int is_set);
/* bytes of device memory currently held by the ctx / free on its device */
int pml_ctx_memory(pml_ctx* ctx, uint64_t* held, uint64_t* device_free);
/*
 * What the F81-family sweeps of this context (tree uploaded, columns allocated) will run: *level_schedule = 1 if they
 * take the level schedule with two-level / stacked units (0: single launch, subtree blocks, plain levels, other models),
 * and the numbers of two-level and of stacked nodes in it.  For tools that model the sweeps' traffic (bench.py's byte
 * model is checked against this); the reference has no counterpart.
 */
int pml_schedule_info(pml_ctx* ctx, int32_t* level_schedule, int32_t* n_two_level, int32_t* n_stacked);
/*
 * Which of the schedules the marginal sweeps of the F81 family will take on this context (tree uploaded, columns
 * allocated; enqueue_bottom_up's own decision): the whole sweep in one launch (small forests), subtree blocks + the top
 * above them (mid-size forests; *n_blocks = their number), the level schedule with two-level / stacked units
 * (*n_absorbed: always 0 -- round 4's general two-level units for ragged trees lost their A/B twice and were removed in
 * round 6; the argument is kept for binary compatibility), plain level launches; PML_SCHEDULE_OTHER_MODEL for the matrix / eigen models.  For tests that compare schedules bit for bit: they
 * assert that the schedule under test is the one that runs.
 */
enum { PML_SCHEDULE_SINGLE_LAUNCH = 0, PML_SCHEDULE_BLOCKS = 1, PML_SCHEDULE_TWO_LEVEL = 2, PML_SCHEDULE_LEVELS = 3,
       PML_SCHEDULE_OTHER_MODEL = 4 };
int pml_sweep_schedule(pml_ctx* ctx, int32_t* kind, int32_t* n_blocks, int32_t* n_absorbed);
/*
 * The library's own node numbering.  The layout rules above leave the order of the sibling groups inside a depth to the
 * caller; pml_tree_upload renumbers the nodes of a ragged forest so that the sibling groups a level's units gather lie
 * next to each other in memory ("height order", pml_api.hip: the children of stored nodes by the nodes' fused height, the tips
 * of cherries by the height of the cherry's parent).  Every per-node array of this interface -- masks, tip ids, posteriors,
 * downloads, joint states, selections, the ids of a zero-likelihood report -- stays in the CALLER's numbering: the library
 * permutes rows on the way in and out (host side; whole-table outputs of a renumbered forest cost that pass).  This call
 * reports the numbering in use: new_of_old[caller's id] = the library's id (int32[n_nodes]; the identity for forests that are
 * in height order as given -- balanced trees are).  For tools that model the schedules (bench.py); PASTML_HIP_NO_HEIGHT_ORDER
 * keeps the caller's numbering.
 */
int pml_tree_order(pml_ctx* ctx, int32_t* new_of_old);

/* ---- tree (replaces the ete3 traversals of pastml/ml.py:109,269,449) ------------------------------------------ */
/*
 * bu_order        internal nodes sorted by height (1..n_bu_levels); bu_offsets[n_bu_levels+1] delimits the levels
 * td_parents      internal nodes sorted by depth; td_parent_offsets[n_td_levels+1] delimits them per depth
 * td_offsets      [n_td_levels+1] id range of every depth level
 * post_rank       rank of each node in the reference's processing order (trees in turn, post-order each); only used
 *                 to report the same (parent, child) pair as pastml/ml.py:139-145 when the likelihood is zero
 */
int pml_tree_upload(pml_ctx* ctx, int32_t n_nodes, int32_t n_roots,
                    const int32_t* parent, const int32_t* first_child, const int32_t* n_children, const double* dist,
                    int32_t n_bu_levels, const int32_t* bu_offsets, const int32_t* bu_order,
                    int32_t n_td_levels, const int32_t* td_offsets,
                    const int32_t* td_parent_offsets, const int32_t* td_parents,
                    const int32_t* post_rank);

/* ---- characters (replaces initialize_allowed_states, pastml/ml.py:293-318) ------------------------------------- */
int pml_chars_alloc(pml_ctx* ctx, int32_t n_cols, int32_t k);
/* masks[col_end-col_begin][n_nodes][W] */
int pml_masks_upload(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const uint64_t* masks);
/* fast path: internal nodes all ones, tip j (node id tip_ids[j]) one-hot at states[col][j], or all ones if < 0 */
int pml_masks_from_tip_states(pml_ctx* ctx, int32_t col_begin, int32_t col_end,
                              int32_t n_tips, const int32_t* tip_ids, const int32_t* states);
/*
 * Masks before zero-branch alteration (pastml/ml.py:352-387), needed only by the joint sweep to rewrite the argmax
 * tables of altered nodes (unalter_zero_node_joint_states, pastml/ml.py:408-428).  NULL clears them.
 */
int pml_masks_initial_upload(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const uint64_t* masks);

/* ---- model parameters (replaces Model.transform_t + get_Pij_t state, pastml/models/__init__.py:269) -------------- */
/* per column: pi[k]; scalars sf, tau, tau_factor (t' = (t + tau) * tau_factor * sf) */
int pml_model_set_f81(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi,
                      const double* sf, const double* tau, const double* tau_factor);
int pml_model_set_hky(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi, const double* kappa,
                      const double* sf, const double* tau, const double* tau_factor);
/* d[k], A[k][k], Ainv[k][k] (row-major) per column: pastml/models/generator.py:16-30.  k <= 256 (PML_ERR_UNSUPPORTED beyond).
 * 65 <= k <= 128: the sum sweeps keep one matrix in LDS, which a reversible model allows (generator.py:33-51 builds no other):
 * the call checks that Ainv is A transposed and rescaled (Ainv[m][j] = c_m A[j][m] pi[j], to 1e-8) and re-orthonormalises the
 * eigenvectors on the device (they come from a general solver, orthonormal to 1e-12); where the check fails -- an eigenvalue that
 * repeats, handed over with a non-orthogonal basis -- the sweeps read P(t) of every branch from HBM, as they do beyond 128 states. */
int pml_model_set_eigen(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi,
                        const double* d, const double* A, const double* Ainv,
                        const double* sf, const double* tau, const double* tau_factor);

/* ---- P(t) ---------------------------------------------------------------------------------------------------------- */
/*
 * P(t) of column col for n_t explicit branch lengths -> P_out[n_t][k][k] row-major.
 * Replaces Model.get_Pij_t: pastml/models/F81Model.py:28-46, HKYModel.py:44-82, CustomRatesModel.py:70-79 +
 * generator.py:54-65.
 */
int pml_pij(pml_ctx* ctx, int32_t col, int32_t n_t, const double* t, double* P_out);
/*
 * Materialises the per-branch transition data of every column on the device (F81 family: e = exp(-mu t') per
 * branch; HKY / eigen models: the full k x k matrix per branch).  The sweeps call it implicitly when the model or
 * the tree changed.  If P_out != NULL, additionally copies out P for every branch: P_out[n_cols][n_nodes][k][k].
 */
int pml_pij_batch(pml_ctx* ctx, double* P_out);

/* ---- sweeps --------------------------------------------------------------------------------------------------------- */
/*
 * Bottom-up (Felsenstein / Pupko) sweep over all columns: pastml/ml.py:82-148 (get_bottom_up_loglikelihood,
 * calc_node_bu_likelihood, rescale_log).  is_marginal != 0: sum over child states; == 0: max + argmax tables.
 * loglik_out[n_cols] = sum over the trees of the forest of ln L.
 * On PML_ZERO_LIKELIHOOD err_parent[col] / err_child[col] (node ids, -1 if the column is fine) name the pair the
 * reference would have raised PastMLLikelihoodError for.
 */
int pml_bottom_up(pml_ctx* ctx, int is_marginal, double* loglik_out, int32_t* err_parent, int32_t* err_child);
/*
 * pml_bottom_up in two halves: submit puts the sweep on the context's stream and returns at once, collect waits for it
 * and hands out what pml_bottom_up would have.  A host loop that serves several contexts (groups of characters with
 * different numbers of states, each with its own optimisers: the reference runs them one after the other,
 * pastml/acr.py:213-231) submits all their sweeps before it waits for any: the sweeps overlap on the device and with
 * the host work between them.  Between the two calls nothing else may be called on the context.
 * How the wait is done is the library's business: for short sweeps of at most 64 columns the last kernel raises a word in
 * pinned memory and the host spins on it (the runtime reports the end of a replayed launch sequence 10 - 14 us late);
 * everything else synchronises the stream (always with the switch NO_SPIN_WAIT).  The results are the same.
 */
int pml_bottom_up_submit(pml_ctx* ctx, int is_marginal);
/*
 * pml_bottom_up_submit for some of the columns: active[n_cols], 0 = the column sits this sweep out (NULL = all).  What
 * pml_bottom_up_collect returns for such a column is what the last sweep that computed it returned (ln L) and "no error";
 * its parameters and masks must be those of that sweep.  For the optimisers of a group of characters, most of which
 * are done long before the last one (the reference optimises one character at a time, pastml/acr.py:213-231).  The
 * next sweep or pass submitted in any other way computes every column again.  F81 family, marginal sweep; otherwise the
 * flags are ignored.
 */
int pml_bottom_up_submit_columns(pml_ctx* ctx, int is_marginal, const uint8_t* active);
int pml_bottom_up_collect(pml_ctx* ctx, int is_marginal, double* loglik_out, int32_t* err_parent, int32_t* err_child);
/*
 * Top-down sweep + marginal likelihoods + posteriors, fused: pastml/ml.py:240-290 (calculate_top_down_likelihood),
 * :431-465 (calculate_marginal_likelihoods), :486-502 (convert_likelihoods_to_probabilities).  Needs a preceding
 * marginal pml_bottom_up with the same masks.  Outputs are optional (NULL = keep on the device only):
 * posterior_out[n_cols][n_nodes][k], lh_sum_out / lh_sf_out[n_cols][n_nodes] such that
 * log10(lh_sum) - lh_sf is the node's total log10-likelihood (the invariant of pastml/ml.py:468-483).
 */
int pml_top_down_marginals(pml_ctx* ctx, double* posterior_out, double* lh_sum_out, double* lh_sf_out);
/*
 * pml_bottom_up(marginal) + pml_top_down_marginals in one call: both sweeps are submitted before the host waits, so a
 * marginal pass costs one host round trip instead of two (what pastml/ml.py:700-707 does per tree in ml_acr).  Same
 * outputs and statuses as the two calls; on PML_ZERO_LIKELIHOOD the top-down results are invalid.
 */
int pml_marginal_pass(pml_ctx* ctx, double* loglik_out, int32_t* err_parent, int32_t* err_child,
                      double* posterior_out, double* lh_sum_out, double* lh_sf_out);
/*
 * Joint reconstruction after a joint pml_bottom_up: pastml/ml.py:598-622 (choose_ancestral_states_joint).
 * joint_state_out[n_cols][n_nodes].
 */
int pml_joint_backtrace(pml_ctx* ctx, int32_t* joint_state_out);
/* pml_bottom_up(joint) + pml_joint_backtrace in one call (one host round trip); outputs and statuses of the two */
int pml_joint_pass(pml_ctx* ctx, double* loglik_out, int32_t* err_parent, int32_t* err_child, int32_t* joint_state_out);

/*
 * State selection on the device after pml_top_down_marginals: method 0 = MAP (pastml/ml.py:577-595), 1 = MPPA
 * (pastml/ml.py:505-574).  lh_mask (optional, [n_cols][n_nodes][W]): masks multiplied into the marginal likelihoods
 * first, i.e. the '.initial' masks of the nodes altered by zero-branch handling and all ones elsewhere
 * (ml.py:541-542, 593-594).  force_joint (MPPA): the joint states of the last pml_joint_backtrace are always kept.
 * The selected masks REPLACE the columns' allowed-state masks on the device (ready for the restricted sweep) and are
 * copied to masks_out[n_cols][n_nodes][W] / n_states_out[n_cols][n_nodes] if not NULL.
 */
int pml_select_states(pml_ctx* ctx, int method, int force_joint, const uint64_t* lh_mask, uint64_t* masks_out,
                      int32_t* n_states_out);

/*
 * Expected numbers of state changes i -> j per scenario, estimated from n_repetitions ancestral scenarios drawn from
 * the marginal posterior of column col: the sampling scheme of pastml/ml.py:753-862 (marginal_counts) for forests
 * without nodes altered by the zero-branch handling (ml.py:352-387; with them: pml_marginal_counts_altered), with a counter-based
 * generator (Philox-4x32-10 keyed by seed): statistical, not bitwise, parity with the reference's numpy draws.
 * Needs pml_bottom_up (marginal) and pml_top_down_marginals first.  counts_out[k][k].
 */
int pml_marginal_counts(pml_ctx* ctx, int32_t col, int32_t n_repetitions, uint64_t seed, double* counts_out);
/*
 * The same for a forest with nodes altered by the zero-branch handling (altered[n_nodes], 1 = altered; pastml/ml.py:352-387).  The
 * scenarios are drawn as above.  A (parent, child) pair with an altered end does not enter the sums, and a parent of such a pair
 * keeps its diagonal correction: the reference gives those pairs fractional counts from the state counts of their two nodes
 * (ml.py:806-812, 840-853, 857-858), which the caller forms from what comes back -- sums_out[k][k]: the sums of the draws of all
 * other pairs, diagonal corrected for all other parents, NOT divided by n_repetitions; state_counts_out[n_nodes][k]: how often
 * each node was in each state; same_out[n_nodes][k]: for a parent of such a pair, its same-state draws over its other children
 * (rows of other nodes are zero).  pastml_amd.ml.marginal_counts is that caller.
 */
int pml_marginal_counts_altered(pml_ctx* ctx, int32_t col, int32_t n_repetitions, uint64_t seed, const uint8_t* altered,
                                double* sums_out, int32_t* state_counts_out, int32_t* same_out);

/* ---- multi-GPU (one process per GPU) ----------------------------------------------------------------------------------- */
/*
 * The characters of a run are independent (pastml/acr.py:213-231 hands them to a pool one by one, each with its own
 * model instance), so they shard over the GPUs of a node with the tree replicated and no exchange on the data path.
 * The one collective is the sum of the per-rank log-likelihoods (what a caller optimising shared parameters over all
 * characters, or reporting the total, needs): an all-reduce of 8 bytes over RCCL / xGMI on the ctx's own stream.
 * librccl is opened with dlopen on first use.  Rank 0 makes the id with pml_comm_unique_id and the host hands it to the
 * other ranks (file, socket, environment: 128 opaque bytes); every rank then calls pml_comm_init on its ctx (collective).
 * world == 1 needs neither an id nor librccl.  The communicator survives pml_tree_upload and dies with the ctx.
 */
#define PML_COMM_ID_BYTES 128
enum { PML_COMM_SUM = 0, PML_COMM_MAX = 1 };
int pml_comm_unique_id(unsigned char* id_out /* [PML_COMM_ID_BYTES] */);
int pml_comm_init(pml_ctx* ctx, int rank, int world, const unsigned char* id /* [PML_COMM_ID_BYTES] */);
int pml_comm_destroy(pml_ctx* ctx);
/*
 * What the attached communicator is, for a run's report (bench.py prints it in every N > 1 line so that the first run on a
 * real 8-GPU node can be read without a debugger): *rank, *world as given to pml_comm_init; *backend = 0 for a world of one
 * without librccl, 1 for RCCL; *rccl_ranks = what ncclCommCount says of the communicator (0 when backend == 0, -1 when the
 * loaded librccl has no ncclCommCount).  pml_device_uuid: the 16 bytes of hipDeviceGetUuid as 32 hex digits + NUL --
 * two ranks that report the same UUID share a GPU.
 */
int pml_comm_info(pml_ctx* ctx, int32_t* rank, int32_t* world, int32_t* backend, int32_t* rccl_ranks);
int pml_device_uuid(int device, char* uuid_out /* [33] */);
/* out[i] = sum / max over the ranks of in[i]; in and out are host arrays of count doubles (may alias); collective */
int pml_comm_allreduce(pml_ctx* ctx, const double* in, double* out, int32_t count, int op);
/*
 * total_out[0] = sum over all ranks of (loglik[0] + ... + loglik[n_cols-1]), loglik being this rank's pml_bottom_up
 * output: the forest-wide, all-character log-likelihood (the sum pastml/ml.py:112-121 forms per character, added up
 * over the characters of pastml/acr.py:226-231).  Collective.
 */
int pml_allreduce_loglik(pml_ctx* ctx, const double* loglik, int32_t n_cols, double* total_out);
/*
 * The same total without a host round trip of its own: when a communicator is attached, pml_marginal_pass forms the sum
 * of its columns' log-likelihoods on the device (column order), all-reduces it on the sweep's stream behind the sweeps
 * and copies it back before its one wait; pml_loglik_total hands that value out (once per pml_marginal_pass; every rank
 * must use the same route).
 */
int pml_loglik_total(pml_ctx* ctx, double* total_out);
/* hipDeviceSynchronize on the given device (benchmarks bracket their timed region with it) */
int pml_device_sync(int device);

/* ---- host-side helper of the optimiser loop (no GPU involved) -------------------------------------------------------- */
/*
 * The points of ONE forward-difference gradient of an F81-family character, decoded for the device, in one call: what
 * scipy's approx_derivative(method='2-point', abs_step=1e-8, bounds=...) evaluates around x -- x itself, then x + h_i e_i
 * with h_i = 1e-8 (the default L-BFGS-B differencing of pastml/ml.py:231) -- pushed through the model's parameter layout
 * (pastml/models/__init__.py:159-181, :328-330: x = [sf?] [tau?] [pi_0/pi_{k-1} .. pi_{k-2}/pi_{k-1}]?,
 * pi = (ratios, 1) / their sum with numpy's pairwise summation; tau factor of models/__init__.py:39-42).  Between two
 * sweeps of an optimiser round this arithmetic was a third of the host's time when done in numpy, a few small arrays at a
 * time.  n = len(x) = opt_sf + opt_tau + (free_pi ? k - 1 : 0); outputs: rows 0 .. n of pi_out[(n + 1) * k], sf_out,
 * tau_out, tf_out [n + 1] (row 0 = x itself) and steps_out[n] = (x_i + h_i) - x_i, the divisors of the differences.
 * Returns PML_OK, or PML_ERR_UNSUPPORTED when some x_i + h_i leaves [lower_i, upper_i] or equals x_i (scipy then mirrors
 * or rescales the step: the caller takes its general path) -- nothing is written in that case.
 * step: the absolute step h (1e-8 = scipy's; the polish run of a many-parameter search takes 1e-6).
 */
int pml_host_f81_fd_points(int32_t n, int32_t k, const double* x, const double* lower, const double* upper, int32_t opt_sf,
                           int32_t opt_tau, int32_t free_pi, double sf_fixed, double tau_fixed, const double* pi_fixed,
                           double forest_length, double num_nodes, double* pi_out, double* sf_out, double* tau_out,
                           double* tf_out, double* steps_out, double step);

/* ---- inspection ------------------------------------------------------------------------------------------------------ */
int pml_download(pml_ctx* ctx, int what, int32_t col, void* out);
/*
 * Rows first, first + stride, ... (count of them) of one column's buffer: what a caller that reports a sample of the
 * nodes needs (the marginal-probability table of named nodes, pastml/ml.py:486-502; validation of full-size runs)
 * without moving n_nodes * k doubles over PCIe.  what: PML_BUF_POSTERIOR (double[count][k]), PML_BUF_LH_SUM,
 * PML_BUF_LH_SF (double[count]) or PML_BUF_JOINT_STATE (int32[count]).
 */
int pml_download_strided(pml_ctx* ctx, int what, int32_t col, int32_t first, int32_t stride, int32_t count, void* out);
/* HIP-event timer on the ctx's stream */
int pml_timer_start(pml_ctx* ctx);
int pml_timer_stop(pml_ctx* ctx, float* milliseconds);
/*
 * Kernel-time accounting with HIP events on the ctx's stream (the stream the kernels are launched on).
 * While enabled, every pml_bottom_up / pml_top_down_marginals / pml_pij_batch call brackets its level-kernel
 * launches with an event pair and adds the elapsed time to an accumulator.
 * which: 0 = bottom-up level kernels, 1 = top-down level kernels, 2 = P(t) / prep kernels, 3 = the top-down launch of
 * the two-level units, 4 = their bottom-up launch (F81 family, level schedule of large forests; zero elsewhere).
 * launches counts kernel launches.  reset != 0 clears the accumulator after reading.
 */
int pml_profile_enable(pml_ctx* ctx, int on);
int pml_profile_read(pml_ctx* ctx, int which, double* total_ms, int64_t* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
