/* gs_amd.h -- C ABI of the MI355X-native Groth-Sahai prove/verify engine.
 *
 * Drop-in boundary for the hot path of jdwhite48/groth-sahai-rs.  The reference
 * has no FFI layer; these entry points are what a Rust `extern "C"` shim
 * implementing its public traits would bind (INTEGRATION.md shows the shim):
 *
 *   gs_commit_*            <- src/prover/commit.rs:59-256  (commit_G1, batch_commit_G1, ... scalar_to_B2)
 *   gs_prove_batch         <- src/prover/prove.rs:92-171,195-274,298-379,409-488  (Provable::prove), and with
 *                             xcoms/ycoms != NULL src/prover/prove.rs:72-90,175-193,278-296,383-408 (commit_and_prove)
 *   gs_verify_batch        <- src/verifier.rs:23-157       (Verifiable::verify, exact semantics, bool per equation)
 *   gs_verify_batch_rlc    <- (new) batched pairing-product check, one final exponentiation per batch
 *   gs_mat_left_mul_com1/2 <- src/data_structures.rs:696-742 (Mat::left_mul on Matrix<Com1/Com2>)
 *   gs_pairing_sum         <- src/data_structures.rs:494-502 (ComT::pairing_sum)
 *   gs_set_crs             <- consumes src/generator.rs:35-42 (struct CRS)
 *
 * DATA LAYOUT (arkworks' in-memory limbs; all little-endian):
 *   Fq  = NQ u64 limbs of a*2^(64 NQ) mod p   (BLS12-381: NQ=6, 48 B; BN254: NQ=4, 32 B)
 *   Fr  = 4 u64 limbs of a*2^256 mod r        (32 B, Montgomery)
 *   G1  = x || y                (identity = all-zero bytes)
 *   G2  = x.c0 || x.c1 || y.c0 || y.c1        (identity = all-zero bytes)
 *   Com1 = G1 || G1,  Com2 = G2 || G2
 *   GT  = 12 Fq, order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1
 *   CRS = u[0],u[1] (Com1) || v[0],v[1] (Com2) || g1 || g2 || gt
 * Batches are arrays-of-structs, equation-major, row-major inside an equation
 * (Gamma[N][m][n], R[N][m][kx], S[N][n][ky], T[N][ky][kx]).
 *
 * Equation types and shapes (kx = columns of R = #pi, ky = columns of S = #theta):
 *   GS_PPE    : X in G1^m, Y in G2^n, A in G1^n, B in G2^m, target GT,  kx=2, ky=2
 *   GS_MSMEG1 : X in G1^m, y in Fr^n, A in G1^n, b in Fr^m, target G1,  kx=2, ky=1
 *   GS_MSMEG2 : x in Fr^m, Y in G2^n, a in Fr^n, B in G2^m, target G2,  kx=1, ky=2
 *   GS_QUAD   : x in Fr^m, y in Fr^n, a in Fr^n, b in Fr^m, target Fr,  kx=1, ky=1
 *
 * Randomness (R, S, T) is ALWAYS an input: the shim draws it from the caller's
 * Rng in the reference's order (R row-major, then S, then T).
 *
 * Pointers: the *_dev entry points take DEVICE pointers (hipMalloc'ed, 16-byte
 * aligned) and enqueue on the context's stream without synchronising; the
 * un-suffixed ones take HOST pointers, stage through device memory and return
 * after the results are back.  A gs_ctx is bound to one GPU and may be used from
 * one host thread at a time.  Points must lie in the prime-order subgroups (as
 * arkworks' deserialisation guarantees).  Nothing is checked on the compute entry
 * points, and -- UNLIKE the reference, whose plain double-and-add works on any curve
 * point -- results for on-curve points OUTSIDE the subgroups are UNDEFINED here
 * (scalar multiplications use the GLV / psi-GLS endomorphisms, which act as a
 * scalar only on the r-torsion).  Three ways to be safe: decode untrusted bytes with
 * gs_wire_decode_* (validate = 1); run in-memory limbs through gs_validate_points[_dev]
 * (the same canonical / on-curve / r-torsion tests); or gs_set_option("endo", 0), which
 * runs every variable-base scalar multiplication as plain double-and-add on any curve
 * point, exactly the reference's tolerance.
 *
 * Errors: the reference panics on shape mismatch (assert_eq!, e.g.
 * src/prover/prove.rs:106-113); here every call returns a status and a shim
 * turns GS_ERR_SHAPE back into a panic.  There is NO CPU fallback: without a
 * usable GPU every compute call returns GS_ERR_DEVICE.
 */
#ifndef GS_AMD_H
#define GS_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gs_ctx gs_ctx;

enum { GS_CURVE_BLS12_381 = 0, GS_CURVE_BN254 = 1 };
enum { GS_PPE = 0, GS_MSMEG1 = 1, GS_MSMEG2 = 2, GS_QUAD = 3 }; /* = EquType, src/statement.rs:42-49 */
enum { GS_OK = 0, GS_ERR_SHAPE = 1, GS_ERR_DEVICE = 2, GS_ERR_ARG = 3, GS_ERR_NOCRS = 4, GS_ERR_ALLOC = 5 };

/* ---- context ---------------------------------------------------------- */
int gs_ctx_create(int curve_id, int device_ordinal, gs_ctx** out);
void gs_ctx_destroy(gs_ctx* ctx);
/* Several GPUs: one context per device ordinal, sharded by the caller, or gs_ctx_create_multi below. */
int gs_set_stream(gs_ctx* ctx, void* hip_stream); /* hipStream_t; NULL = default stream */
/* Planner overrides (results never change, only which kernel shapes run; tests force every shape through them):
 *   "miller_twin"  -1 planned | 0 one accumulator per Miller lane | 1 two (lines of Q shared by both G1 partners)
 *                   | 2 the same triples on a PAIR of lanes, one accumulator each, lines exchanged through LDS | 3 the
 *                   same with a DPP exchange (planned: the pair form, LDS on BLS12-381, DPP on BN254)
 *   "miller_ch"     0 planned | 1..12 pairs (triples) per Miller lane
 *   "var_tm"        0 planned | 1..8 variable-base terms per Straus lane
 *   "var_mo"        0 planned | 1, 2, 4 outputs over the same bases served by one lane's table build
 *   "var_w"         0 planned | 4, 5 window width of the Straus lanes (8 or 16 table entries per base)
 *   "var_ws_lanes"  0 = 2^19 | multiple of 64: Straus lanes per launch; bounds their table workspace (3.5 .. 28 KB
 *                   of device memory per lane), larger batches run as several launches over it
 *   "red_k"         0 planned | 1, 2, 4, 8 outputs per reduction lane (one inversion per lane)
 *   "coop_fe"       0 one lane per final exponentiation | 1 planned | 2 always the 3-lane cooperative form
 *   "line_tables"   1 CRS G2 arguments read precomputed Miller lines | 0 they are stepped like any other point
 *   "overlap"       1 independent kernels of a small batch on internal side streams | 0 one stream
 *   "var_tab"      -1 planned | 0 the verifier's Gamma^T c on Straus lanes with their own tables | 1 on window tables of
 *                   the commitment components shared by all outputs (8-bit windows; what large arities use)
 *   "var_w2"       -1 planned | 0 | 1 the G1 Straus lanes in the kernels built for TWO waves per SIMD (256 registers: the
 *                   register-only G1 point operations leave room); planned when a launch puts two waves on every SIMD
 *   "endo"          1 (default) GLV / psi-GLS scalar multiplications: r-torsion points only | 0 every variable-base scalar
 *                   multiplication is a plain signed-window double-and-add lane ("k_var.plain", "k_smul_batch.plain"):
 *                   defined on ANY curve point, like the reference's Com::scalar_mul (data_structures.rs:336-342), at
 *                   ~2.5x the variable-base work.  For callers that cannot vouch for subgroup membership and do not want
 *                   to pay gs_validate_points first.  (This one changes WHICH inputs are supported, not the results on
 *                   supported ones.)
 *   "mixed_merge"  -1 planned (merged up to 2^14 equations per call on a 256-CU device) | 0 the parts of a mixed call run
 *                   one after the other | 1 their launches are merged
 * The same knobs are read ONCE at gs_ctx_create from the environment for experiments without recompiling the caller:
 * GS_MILLER_TWIN, GS_MILLER_CH, GS_VAR_TM, GS_VAR_MO, GS_VAR_W, GS_VAR_W2, GS_RED_K, GS_COOP_FE, GS_LINE_TABLES, GS_OVERLAP,
 * GS_ENDO (same values).
 * Unset = planned.  GS_COPY_THREADS (default 4): memcpy workers of the host-pointer entry points; GS_ROCTX=1: load the
 * roctx library for phase markers ("gs.prove", "gs.prove.g1", "gs.verify.miller" ...) even when no profiler mapped it.
 * Diagnostics on stderr: GS_PLAN_TRACE=1 (the verifier's Miller plan per call: mode, budget, tasks per equation, planned
 * cost of each task's lane, waves), GS_PIPE_TRACE=1 (time line of a host-pointer call's staging, uploads and arrivals);
 * GS_PIPE_NO_DIRECT=1 stages registered arrays like pageable ones. */
int gs_set_option(gs_ctx* ctx, const char* key, int value);
int gs_sync(gs_ctx* ctx);                         /* hipStreamSynchronize on the context's stream */
/* Page-locked caller memory.  The host-pointer entry points below (gs_prove_batch, gs_verify_batch, the mixed and the
 * Statement calls, gs_multi_*) stage pageable arrays through a pinned buffer of their own (one memcpy per array and
 * direction on worker threads).  An array that lies inside a range registered HERE is moved by DMA straight from / to
 * the caller's memory: no staging copy, uploads start at once (profiles/r3/host_path_rate.txt).  A Rust caller
 * registers the Vecs it reuses across calls once (registration costs about as much as copying the buffer a few times).
 * [prove.rs:29-52 / verifier.rs:18-21 take slices; this is the cheap way to hand them over]
 * Only ranges registered through this call count: memory page-locked by other means (hipHostMalloc, another library's
 * hipHostRegister) is staged like pageable memory -- the runtime cannot tell such memory from ranges it has pinned
 * itself for an earlier pageable copy, and those may be mapped read-only or belong to a buffer freed since.
 * Registrations are per process, not per context: they outlive gs_ctx_destroy and end with gs_host_unregister (any live
 * context may be passed).  gs_host_unregister first drains EVERY live context of the process (compute stream, side
 * streams and the copy queues -- the shards of a gs_multi context included): no DMA touches the range afterwards.  GS_ERR_ARG: null / empty / already registered (here or
 * elsewhere) / unknown. */
int gs_host_register(gs_ctx* ctx, void* ptr, size_t bytes);
int gs_host_unregister(gs_ctx* ctx, void* ptr);
/* The easy way to the same: gs_host_alloc returns `bytes` of host memory on a mapping of its own (page-aligned, never
 * part of the allocator's heap), every page already touched (no first-touch faults inside a call: result arrays made
 * per call cost 27-35 ms of them at 2^16), page-locked and registered -- a shim that keeps its Vec-like buffers in such
 * memory gets DMA-direct transfers with one call per buffer.  gs_host_free drains every context, unregisters and unmaps.
 * (What a Rust shim would wrap in a Drop type: INTEGRATION.md.) */
int gs_host_alloc(gs_ctx* ctx, size_t bytes, void** out);
int gs_host_free(gs_ctx* ctx, void* ptr);
const char* gs_last_error(gs_ctx* ctx);
const char* gs_version(void);
/* sizes in bytes of the boundary PODs for a curve: out[0..5] = Fq, Fr, G1, G2, GT, CRS */
int gs_sizes(int curve_id, size_t out[6]);

/* CRS (host pointer): uploads, derives W1 = u[1]+(O,g1), W2 = v[1]+(O,g2) and
 * builds the fixed-base window tables (16-bit windows: 1.8 GB of device memory) and the Miller
 * line tables of its G2 elements on the device.  The tables are immutable and SHARED: every context of the process
 * that installs the same CRS bytes on the same device uses one copy (reference-counted; freed with the last context),
 * so a pool of worker contexts costs 1.8 GB per device, not per worker.
 * What IS per context (grow-only, sized by the largest batch the context has seen; freed by gs_ctx_destroy):
 *   - engine scratch: scalar pool, Jacobian partial slots, Miller partials: ~1 GB at 2^16 PPEs of 4 + 4 variables;
 *   - the Straus lanes' workspaces (affine tables + the Jacobian staging of their build): 9 .. 70 KB per lane, at most
 *     2 x (SIMDs of the device) x 64 lanes per side ("var_ws_lanes" lowers that): <= 9.2 GB for the largest lane
 *     shape (G2, 8 terms, 5-bit windows), ~6 GB in total for the 2^16 PPE shapes;
 *   - host-pointer entry points only: a pinned host buffer and device staging of the call's input + output bytes each
 *     (a 2^16 PPE 4x4: 0.45 GB for prove, 0.37 GB for verify; grow-only, so a context that does both holds 0.45 GB);
 *   - large arities only: the shared per-base window tables, N x 2 m bases x 128 entries (affine + Jacobian staging:
 *     36 KB per base in G1; planned only while that stays below 8 GB);
 *   - mixed calls: the above once per PART (the parts' scratch is live at the same time).
 * A pool of W worker contexts on one GPU therefore costs 1.8 GB + W x (the above for the workers' batch size). */
int gs_set_crs(gs_ctx* ctx, const void* crs_host);

/* CRS of the reference's shape (src/generator.rs:81-118, binding key :48-60) from caller-drawn values:
 * generators p1 (G1), p2 (G2) and scalars[4] = a1, a2, t1, t2 (Fr):
 *   u = [(p1, a1 p1), (t1 p1, t1 a1 p1)],  v = [(p2, a2 p2), (t2 p2, t2 a2 p2)],  g1 = p1, g2 = p2, gt = e(p1, p2).
 * Host pointers; the result is written to crs_out (CRS layout) and NOT installed (call gs_set_crs). */
int gs_crs_generate(gs_ctx* ctx, const void* p1_g1, const void* p2_g2, const void* scalars_fr4, void* crs_out);
/* The simulation ("hiding") key of src/generator.rs:65-77 (dead code there): as above with
 *   u[1] = (t1 p1, t1 a1 p1 - p1),  v[1] = (t2 p2, t2 a2 p2 - p2). */
int gs_crs_generate_hiding(gs_ctx* ctx, const void* p1_g1, const void* p2_g2, const void* scalars_fr4, void* crs_out);

/* ---- commit (src/prover/commit.rs) -------------------------------------- */
/* c_i = iota1(X_i) + r_i0 u0 + r_i1 u1   (commit.rs:78-100; N=1 is commit_G1 :59-75) */
int gs_commit_g1_dev(gs_ctx*, size_t count, const void* x_g1, const void* rand_fr2, void* out_com1);
int gs_commit_g2_dev(gs_ctx*, size_t count, const void* y_g2, const void* rand_fr2, void* out_com2);
/* c_i = x_i W1 + r_i u0                  (commit.rs:125-156, 225-256) */
int gs_commit_fr_b1_dev(gs_ctx*, size_t count, const void* x_fr, const void* rand_fr, void* out_com1);
int gs_commit_fr_b2_dev(gs_ctx*, size_t count, const void* y_fr, const void* rand_fr, void* out_com2);
int gs_commit_g1(gs_ctx*, size_t count, const void* x_g1, const void* rand_fr2, void* out_com1);
int gs_commit_g2(gs_ctx*, size_t count, const void* y_g2, const void* rand_fr2, void* out_com2);
int gs_commit_fr_b1(gs_ctx*, size_t count, const void* x_fr, const void* rand_fr, void* out_com1);
int gs_commit_fr_b2(gs_ctx*, size_t count, const void* y_fr, const void* rand_fr, void* out_com2);

/* ---- prove (src/prover/prove.rs) ---------------------------------------- */
/* Batch of N independent equations of one type and shape over the shared CRS.
 * xcoms/ycoms may be NULL (prove only) or receive the commitments
 * (commit_and_prove).  pi: N*kx Com2, theta: N*ky Com1.
 * Host-pointer forms (no _dev suffix; what a Rust caller of prove.rs:29-52 / verifier.rs:18-21 holds): the arrays are
 * staged through a grow-only PINNED buffer by a few memcpy workers, uploaded on a copy stream array by array, and the
 * kernels wait only for the arrays they read (scalars before the preparation kernel, G1 arguments before the G1 side,
 * G2 arguments before the G2 side / the Miller loop), so most of the transfer runs under kernels; outputs come back
 * the same way.  Per context: pinned staging = the largest call's input + output bytes (2^16 PPE 4x4:
 * 0.45 GB for prove, 0.37 GB for verify; 0.83 GB is what ONE prove + verify step moves across PCIe, not what is held),
 * device staging the same, plus the engine's scratch (section "memory" of DESIGN.md). */
int gs_prove_batch_dev(gs_ctx*, int equ_type, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                       const void* B, const void* Gamma, const void* R, const void* S, const void* T, void* xcoms,
                       void* ycoms, void* pi, void* theta);
int gs_prove_batch(gs_ctx*, int equ_type, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                   const void* B, const void* Gamma, const void* R, const void* S, const void* T, void* xcoms,
                   void* ycoms, void* pi, void* theta);

/* A Statement (src/statement.rs:24-28,109: "a list of equations ... defined with respect to the list of variables that
 * span across ALL equations"): E equations of ONE type over the SAME m + n variables.  The variables are committed ONCE
 * (X[m], Y[n] with randomness R[m][kx], S[n][ky]: exactly gs_commit_*), every equation e gets its own proof from its
 * A[e], B[e], Gamma[e] and T[e][ky][kx] -- what calling Provable::prove E times with the same commitments does
 * (prove.rs:92-171 ...).  xcoms[m] / ycoms[n] may be NULL (already committed).  pi: E*kx Com2, theta: E*ky Com1. */
int gs_prove_statement_dev(gs_ctx*, int equ_type, size_t E, int m, int n, const void* X, const void* Y, const void* A,
                           const void* B, const void* Gamma, const void* R, const void* S, const void* T, void* xcoms,
                           void* ycoms, void* pi, void* theta);
int gs_prove_statement(gs_ctx*, int equ_type, size_t E, int m, int n, const void* X, const void* Y, const void* A,
                       const void* B, const void* Gamma, const void* R, const void* S, const void* T, void* xcoms,
                       void* ycoms, void* pi, void* theta);

/* ---- verify (src/verifier.rs) ------------------------------------------- */
/* ok[i] = 1 iff equation i verifies; exact reference semantics (four GT cell
 * equalities per equation). */
int gs_verify_batch_dev(gs_ctx*, int equ_type, size_t N, int m, int n, const void* A, const void* B,
                        const void* Gamma, const void* target, const void* xcoms, const void* ycoms, const void* pi,
                        const void* theta, uint8_t* ok);
int gs_verify_batch(gs_ctx*, int equ_type, size_t N, int m, int n, const void* A, const void* B, const void* Gamma,
                    const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                    uint8_t* ok);
/* ok[e] = 1 iff equation e of a Statement verifies against the SHARED commitments xcoms[m], ycoms[n]. */
int gs_verify_statement_dev(gs_ctx*, int equ_type, size_t E, int m, int n, const void* A, const void* B,
                            const void* Gamma, const void* target, const void* xcoms, const void* ycoms,
                            const void* pi, const void* theta, uint8_t* ok);
int gs_verify_statement(gs_ctx*, int equ_type, size_t E, int m, int n, const void* A, const void* B, const void* Gamma,
                        const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                        uint8_t* ok);
/* ---- mixed batches and mixed-type Statements: several sub-batches in ONE call ------------------------------------
 * configs[2] of the baseline mixes PPE, MSMEG1 and MSMEG2 equations; the reference's Statement is a list of equations
 * of ANY type over one list of variables (src/statement.rs:24-28,109).  A part is a homogeneous sub-batch (type, N, m,
 * n and the arrays of gs_prove_batch / gs_verify_batch for it); parts may differ in type AND shape.  Outputs are
 * byte-identical to one gs_prove_batch / gs_verify_batch call per part.  How the parts run ("mixed_merge"):
 *   - up to 2^14 equations per call (16 x the device's SIMDs): RECORD AND MERGE on the context's own stream.  Every part's
 *     launches are recorded (nothing is enqueued; the part's scratch buffers carry a per-part tag so that they are live
 *     side by side), then replayed in step: launches of different parts that run the same kernel body become ONE
 *     segmented launch (k_seg, up to 4 segments), so a mixed batch of a few thousand equations fills the chip like a
 *     homogeneous batch of its total size (2^12: 0.92-0.94 of the PPE-only rate);
 *   - larger calls, a single part, or while the kernel profile is on: the parts run one after the other on the
 *     context's stream, each with its own lane shapes (every part fills the chip by itself from ~2^15 on).
 * There are no child contexts or extra streams (measured and rejected: profiles/r3/mixed_streams.txt).
 * shared_vars != 0 makes the part a Statement's: X, Y, R, S hold ONE copy (m / n entries) that all N equations of the
 * part use, xcoms / ycoms are the statement's commitments (m / n entries; prove writes them when non-NULL -- pass them
 * with ONE of the parts that use a variable group and NULL with the others -- verify reads them), i.e. exactly
 * gs_prove_statement / gs_verify_statement per part.  A mixed-type Statement over G1 variables Xg, G2 variables Yg
 * and scalar variables xs, ys is then one call with the parts PPE (Xg, Yg), MSMEG1 (Xg, ys), MSMEG2 (xs, Yg) and
 * QuadEqu (xs, ys) pointing at the shared groups.  At most GS_MIXED_MAX parts per call. */
#define GS_MIXED_MAX 8
typedef struct {
  int equ_type;
  size_t N;
  int m, n;
  const void *X, *Y, *A, *B, *Gamma, *R, *S, *T;
  void *xcoms, *ycoms, *pi, *theta;
  int shared_vars;
} gs_prove_part;
typedef struct {
  int equ_type;
  size_t N;
  int m, n;
  const void *A, *B, *Gamma, *target, *xcoms, *ycoms, *pi, *theta;
  uint8_t* ok;
  int shared_vars;
} gs_verify_part;
int gs_prove_mixed_dev(gs_ctx*, int nparts, const gs_prove_part* parts);
int gs_prove_mixed(gs_ctx*, int nparts, const gs_prove_part* parts);
int gs_verify_mixed_dev(gs_ctx*, int nparts, const gs_verify_part* parts);
int gs_verify_mixed(gs_ctx*, int nparts, const gs_verify_part* parts);

/* Batched pairing-product check: random linear combination of all 4N cell
 * equations with caller-supplied 64-bit exponents rho[N][4] (device/host u64).
 * RHO CONTRACT (soundness rests on it): every rho is drawn from a CSPRNG, fresh for every call, NON-ZERO, secret
 * until the verdict is out, and drawn AFTER the commitments and proofs of the batch are fixed.  A predictable or
 * reused rho lets a prover craft a batch that passes the combined check while single equations fail; the soundness
 * error is 2^-64 per batch for honest-size exponents.  The host entry rejects rho == 0 (GS_ERR_ARG); the _dev entry
 * cannot look.  PPE targets are raised to rho WITHOUT a subgroup check: a target outside the order-r subgroup of GT
 * (impossible for a decoded, validated PairingOutput, gs_wire_decode_gt validate = 1) is the caller's to exclude.
 * The exact entry points above have no such contract.
 * one final exponentiation for the whole batch.  *ok_all = 1 iff the combined
 * check passes; acc (may be NULL for the host entry) receives the accumulator PAIR
 * (2 GT): acc[0] = prod Miller(e,cell)^rho (un-exponentiated), acc[1] = prod t_e^rho
 * (1 for the non-PPE types).  Ranks exchange their pairs (all-gather) and
 * gs_gt_finalize(count, pairs) multiplies them in the given order and checks
 * FE(prod acc[0]) == prod acc[1]: the verdict for the union of the batches. */
int gs_verify_batch_rlc_dev(gs_ctx*, int equ_type, size_t N, int m, int n, const void* A, const void* B,
                            const void* Gamma, const void* target, const void* xcoms, const void* ycoms,
                            const void* pi, const void* theta, const uint64_t* rho, void* acc_gt);
int gs_verify_batch_rlc(gs_ctx*, int equ_type, size_t N, int m, int n, const void* A, const void* B,
                        const void* Gamma, const void* target, const void* xcoms, const void* ycoms, const void* pi,
                        const void* theta, const uint64_t* rho, void* acc_gt, uint8_t* ok_all);
/* product of `count` GT accumulator pairs (host), final exponentiation, FE(prod acc[0]) == prod acc[1] ? */
int gs_gt_finalize(gs_ctx*, size_t count, const void* accs_gt_host, uint8_t* ok_all);
/* the same with the pairs in this context's device memory (e.g. the receive buffer of an all-gather); ok_all on the host */
int gs_gt_finalize_dev(gs_ctx*, size_t count, const void* accs_gt_dev, uint8_t* ok_all);

/* ---- several GPUs of one node ----------------------------------------------------------
 * gs_ctx_create_multi owns one SHARD per listed device ordinal: a context (stream, scratch; CRS tables shared per
 * device) and a persistent host thread bound to that device.  A batch is cut into contiguous blocks of equation
 * indices, block i on shard i (sizes differ by at most one: gs_multi_shard); every block runs the single-device entry
 * point of the same name on its shard's thread.  Identical bytes to a single-device run.  Prove and exact verify use
 * no collective.  The batched verifier's per-shard accumulator pairs (2 GT each) are written into per-shard exchange
 * buffers on their devices, all-gathered (RCCL over xGMI when the shards sit on distinct devices; device-to-device /
 * peer copies for one shard, shards sharing a device or a missing librccl: gs_multi_exchange_note says which),
 * cross-checked, multiplied in shard order on device 0 and finished with ONE final exponentiation: the verdict for the
 * union of the blocks (same rho contract as gs_verify_batch_rlc; rho is indexed by GLOBAL equation number).
 * acc_pairs (host, may be NULL) receives the ndev gathered pairs.  RCCL is bound at run time (dlopen of librccl.so,
 * preferring a copy already in the process).
 *   host-pointer family: whole batch in, whole batch out; every shard stages its block through its context's pinned
 *     pipeline (see gs_prove_batch).
 *   _dev family: arrays of ndev DEVICE pointers, entry i = shard i's block already resident on devices[i] (what a
 *     caller that produced the data on the GPUs holds); nothing crosses PCIe; calls return once every shard has
 *     enqueued its kernels, gs_multi_sync joins.  (gs_multi_verify_batch_rlc_dev returns the verdict, so it joins.)
 * GS_MULTI_SHARED_DEVICES lets several shards name the SAME ordinal: the split, offsets, empty blocks and the pair
 * exchange then run on one GPU exactly as they would on eight (how tests exercise ndev > 1 on a one-GPU box). */
typedef struct gs_multi gs_multi;
enum { GS_MULTI_SHARED_DEVICES = 1 };
int gs_ctx_create_multi(int curve_id, const int* device_ordinals, int ndev, gs_multi** out);
int gs_ctx_create_multi_ex(int curve_id, const int* device_ordinals, int ndev, int flags, gs_multi** out);
void gs_multi_destroy(gs_multi*);
int gs_multi_ndev(gs_multi*);
gs_ctx* gs_multi_ctx(gs_multi*, int i);                      /* shard i's context, e.g. for gs_set_option */
const char* gs_multi_last_error(gs_multi*);
int gs_multi_uses_rccl(gs_multi*);                           /* 1 once the RCCL communicators exist */
const char* gs_multi_exchange_note(gs_multi*);               /* how the accumulator pairs travel (and why not RCCL) */
/* The pair exchange walks three legs -- 0 RCCL all-gather, 1 device / peer copies (hipMemcpyPeerAsync; peer access is
 * probed and enabled at gs_ctx_create_multi), 2 host-staged copies -- and a leg that FAILS AT RUN TIME is marked failed
 * and the same call retries on the next one (it is not an error while a slower way exists; gs_multi_exchange_note names
 * the leg used and why earlier ones failed).  Options: "exchange_leg" 0..2 start the chain there; "exchange_fail" bit
 * mask: leg k fails when it runs (test hook: how the chain is exercised on a one-GPU box; also GS_MULTI_EXCHANGE_FAIL /
 * GS_MULTI_EXCHANGE_LEG at create); "exchange_reset": forget earlier failures. */
int gs_multi_set_option(gs_multi*, const char* key, int value);
int gs_multi_shard(gs_multi*, size_t N, int i, size_t* lo, size_t* hi); /* block [lo, hi) of shard i */
int gs_multi_set_crs(gs_multi*, const void* crs_host);
int gs_multi_sync(gs_multi*);                                /* drain every shard's stream */
int gs_multi_prove_batch(gs_multi*, int equ_type, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                         const void* B, const void* Gamma, const void* R, const void* S, const void* T, void* xcoms,
                         void* ycoms, void* pi, void* theta);
int gs_multi_verify_batch(gs_multi*, int equ_type, size_t N, int m, int n, const void* A, const void* B,
                          const void* Gamma, const void* target, const void* xcoms, const void* ycoms, const void* pi,
                          const void* theta, uint8_t* ok);
int gs_multi_verify_batch_rlc(gs_multi*, int equ_type, size_t N, int m, int n, const void* A, const void* B,
                              const void* Gamma, const void* target, const void* xcoms, const void* ycoms,
                              const void* pi, const void* theta, const uint64_t* rho, void* acc_pairs_gt,
                              uint8_t* ok_all);
int gs_multi_prove_batch_dev(gs_multi*, int equ_type, size_t N, int m, int n, const void* const* X,
                             const void* const* Y, const void* const* A, const void* const* B,
                             const void* const* Gamma, const void* const* R, const void* const* S,
                             const void* const* T, void* const* xcoms, void* const* ycoms, void* const* pi,
                             void* const* theta);
int gs_multi_verify_batch_dev(gs_multi*, int equ_type, size_t N, int m, int n, const void* const* A,
                              const void* const* B, const void* const* Gamma, const void* const* target,
                              const void* const* xcoms, const void* const* ycoms, const void* const* pi,
                              const void* const* theta, uint8_t* const* ok);
/* rho[i]: shard i's exponents on its device, 4 per equation of ITS block */
int gs_multi_verify_batch_rlc_dev(gs_multi*, int equ_type, size_t N, int m, int n, const void* const* A,
                                  const void* const* B, const void* const* Gamma, const void* const* target,
                                  const void* const* xcoms, const void* const* ycoms, const void* const* pi,
                                  const void* const* theta, const uint64_t* const* rho, void* acc_pairs_gt_host,
                                  uint8_t* ok_all);

/* ---- L2 parity hooks (host pointers) ------------------------------------- */
/* out[i] = sum_k lhs[i][k] * col[k]       (data_structures.rs:696-742) */
int gs_mat_left_mul_com1(gs_ctx*, int rows, int k, const void* lhs_fr, const void* col_com1, void* out_com1);
int gs_mat_left_mul_com2(gs_ctx*, int rows, int k, const void* lhs_fr, const void* col_com2, void* out_com2);
/* ComT::pairing_sum(x[0..k), y[0..k)) -> 4 GT cells (00,01,10,11)   (data_structures.rs:494-502) */
int gs_pairing_sum(gs_ctx*, int k, const void* x_com1, const void* y_com2, void* out_comt);
/* Matrix<Fr> product out = lhs (rows x inner) * rhs (inner x cols), row-major Montgomery Fr, host pointers, computed by
 * the prover's scalar-preparation kernels (Psi = R^T Gamma, prove.rs:133): the reference's Mat::right_mul / left_mul on
 * Matrix<Fr> (data_structures.rs:824-912) and the hook its matrix KATs (:1726-1947) run through. */
int gs_fr_matmul(gs_ctx*, int rows, int inner, int cols, const void* lhs_fr, const void* rhs_fr, void* out_fr);
/* batch helpers: out[i] = k[i] * P[i] (P broadcast if p_stride == 0), E::multi_pairing per row */
int gs_g1_mul_batch(gs_ctx*, size_t count, const void* p_g1, int p_broadcast, const void* k_fr, void* out_g1);
int gs_g2_mul_batch(gs_ctx*, size_t count, const void* p_g2, int p_broadcast, const void* k_fr, void* out_g2);
int gs_g1_mul_batch_dev(gs_ctx*, size_t count, const void* p_g1, int p_broadcast, const void* k_fr, void* out_g1);
int gs_g2_mul_batch_dev(gs_ctx*, size_t count, const void* p_g2, int p_broadcast, const void* k_fr, void* out_g2);
int gs_multi_pairing_batch(gs_ctx*, size_t count, int k, const void* p_g1, const void* q_g2, void* out_gt);
int gs_multi_pairing_batch_dev(gs_ctx*, size_t count, int k, const void* p_g1, const void* q_g2, void* out_gt);
/* out[i] = base^(k[i]) in GT (k canonical-Montgomery Fr); used to synthesise satisfied PPE targets */
int gs_gt_pow_batch_dev(gs_ctx*, size_t count, const void* base_gt_dev, const void* k_fr, void* out_gt);

/* ---- canonical wire format (ark-serialize) -----------------------------------
 * Arrays of elements between the boundary form above and the byte strings ark-serialize produces for
 * the reference's derives (src/data_structures.rs:128,132 Com1/Com2; src/prover/commit.rs:18,24;
 * src/prover/prove.rs:55; src/statement.rs:117-179; src/generator.rs:35): Fr / Fq little-endian canonical
 * integers; GT = 12 Fq; BLS12-381 points in ark-bls12-381's zcash-compatible big-endian flagged form,
 * BN254 points in ark-ec's default little-endian flagged form (details: csrc/gs_wire.cuh).  Struct
 * framing (Vec<T> = u64-LE length + items, fields in declaration order, EquType = 1 byte) is host-side:
 * include/gs_amd.hpp, groth_sahai_rs_amd/wire.py.  Host pointers.  decode: ok[i] = 1 iff element i is a
 * well-formed encoding (canonical coordinates, consistent flags, on the curve) and, with validate != 0,
 * passes the r-torsion check of ark-serialize's Validate::Yes; rejected elements decode to the identity
 * (points) / zero.  The identity has exactly ONE accepted encoding (infinity flag, all-zero payload, no sort flag):
 * ark-bls12-381 ignores the payload once it sees the infinity flag, this decoder does not (stricter; arkworks'
 * own serialiser never produces the other byte strings).  sizes: out[0..5] = G1 compressed, G1 uncompressed, G2 compressed, G2 uncompressed, Fr, GT. */
int gs_wire_sizes(int curve_id, size_t out[6]);
int gs_wire_encode_g1(gs_ctx*, size_t n, int compressed, const void* pts_g1, uint8_t* out);
int gs_wire_encode_g2(gs_ctx*, size_t n, int compressed, const void* pts_g2, uint8_t* out);
int gs_wire_decode_g1(gs_ctx*, size_t n, int compressed, int validate, const uint8_t* in, void* pts_g1, uint8_t* ok);
int gs_wire_decode_g2(gs_ctx*, size_t n, int compressed, int validate, const uint8_t* in, void* pts_g2, uint8_t* ok);
int gs_wire_encode_fr(gs_ctx*, size_t n, const void* fr, uint8_t* out);
int gs_wire_decode_fr(gs_ctx*, size_t n, const uint8_t* in, void* fr, uint8_t* ok);
int gs_wire_encode_gt(gs_ctx*, size_t n, const void* gt, uint8_t* out);
int gs_wire_decode_gt(gs_ctx*, size_t n, int validate, const uint8_t* in, void* gt, uint8_t* ok);

/* ---- subgroup safety for in-memory points (the wire decoder's tests without the wire format) ----------------------
 * ok[i] = 1 iff point i of pts (G1: group = 1, G2: group = 2; boundary limbs as everywhere) has canonical coordinates
 * (every word string < p), is the identity (0, 0) or lies on the curve, AND is in the prime-order subgroup (BLS12-381:
 * the endomorphism tests arkworks uses; BN254 G1: cofactor 1, G2: [r]Q = O).  The reference needs no such call: its
 * double-and-add takes any curve point (src/data_structures.rs:336-342) and arkworks' deserialisation has validated
 * everything it holds. */
int gs_validate_points_dev(gs_ctx*, int group, size_t n, const void* pts_dev, uint8_t* ok_dev);
int gs_validate_points(gs_ctx*, int group, size_t n, const void* pts, uint8_t* ok);

/* ---- measurement hook ----------------------------------------------------
 * Name and average duration (ms, HIP events on the context's stream) of the
 * kernels launched since gs_prof_reset; used by bench.py for the roofline. */
int gs_prof_enable(gs_ctx*, int on);
int gs_prof_reset(gs_ctx*);
int gs_prof_get(gs_ctx*, int idx, char* name, size_t name_cap, double* total_ms, uint64_t* launches);
/* lane-tasks launched under that name and its kernel-specific work items (fixed-base scalars for k_fix, terms for
 * k_var_multi, partial sums for k_red, pairs for k_miller, lanes otherwise): inputs of bench.py's ALU roofline */
int gs_prof_get_work(gs_ctx*, int idx, uint64_t* lanes, uint64_t* work);
/* the clock (GHz) the launches under entry idx ran at, measured INSIDE them: every wave of a segmented launch stamps
 * s_memtime (shader cycles) and s_memrealtime (100 MHz) around its body while the profile is on; 0 where unavailable.
 * bench.py prices the multiply-add issue peak at this clock (one measurement: no separate clock correction). */
int gs_prof_get_clock(gs_ctx*, int idx, double* ghz);

#ifdef __cplusplus
}
#endif
#endif /* GS_AMD_H */
