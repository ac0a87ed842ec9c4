/*
 * nbody3d_hip.h -- C ABI of the MI355X direct N-body engine (libnbody3d_hip.so)
 *
 * This is the drop-in boundary for the ONE hot path of huj31415/nbody3d-webgpu:
 * the tiled O(N^2) force accumulation + leapfrog update that the reference runs
 * as a single WGSL compute pass.  The reference has no FFI/plugin interface for
 * it: the path sits behind the WebGPU object protocol inside nbody3d.js.  Each
 * entry point below names the piece of that protocol it replaces (file:line
 * relative to /root/reference).  INTEGRATION.md shows the N-API / ctypes binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every call returns an nb_status (0 = ok).  Nothing throws or aborts across
 *     the boundary; nb_last_error() gives the message for the last failure.
 *   - Host arrays use the reference's packed layout (nbody3d.js:49,132):
 *       bodies[4*i..] = x, y, z, mass     vel[4*i..] = vx, vy, vz, 0
 *       accel [4*i..] = ax, ay, az, 0     (acceleration of the previous step)
 *     element type float for NB_F32 sims, double for NB_F64 sims.
 *   - The engine never keeps a host pointer past the call (the reference's
 *     writeBuffer copies, nbody3d.js:186,193); it owns all device memory unless
 *     nb_config.ext_bodies is given.  nb_download fills caller-owned arrays
 *     (util.js:163-178 returns a fresh copy).
 *   - One host thread per handle at a time; distinct handles are independent.
 *   - nb_step enqueues on the engine's HIP stream and returns (the reference's
 *     queue.submit is asynchronous, nbody3d.js:490); nb_download / nb_sync block.
 */
#ifndef NBODY3D_HIP_H
#define NBODY3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built -fvisibility=hidden */
#endif

#define NB_ABI_VERSION 2u /* major: a client and a library must agree on it (nb_abi_version) */
#define NB_ABI_MINOR 3u   /* additions within major 2; a client needs nb_abi_minor() >= the minor it was written against:
                             2.1 (round 3)  nb_force_pass, nb_frame_request / nb_frame_acquire, nb_shape_info, NB_FLAG_SYM_SHARD
                             2.2 (round 4)  nb_step_times2, nb_plan_query, NB_FLAG_WHOLE_SWEEPS, nb_config.layer_budget_mib
                             2.3 (round 5)  nb_abi_minor, NB_MULTI_PEER_OVERLAP; nb_plan_info / nb_plan_query moved to nbody3d_hip_plan.h; force_variant 7 II LL 3 takes LL up to 64;
                                            nb_plan_query's table holds four words per wave instead of the W + 1 starts */

typedef struct nb_sim nb_sim; /* opaque */

typedef enum nb_status {
    NB_OK = 0,
    NB_ERR_INVALID = 1,   /* bad argument / bad config                       */
    NB_ERR_NO_DEVICE = 2, /* no usable HIP device (reference: nbody3d.js:151) */
    NB_ERR_HIP = 3,       /* a HIP runtime call failed                       */
    NB_ERR_STATE = 4,     /* call out of order (e.g. step before upload)     */
    NB_ERR_NOMEM = 5,
    NB_ERR_COMM = 6,      /* the exchange hook / collective failed           */
    NB_NOT_READY = 7      /* nb_frame_acquire(wait = 0): no finished frame yet */
} nb_status;

typedef enum nb_precision { NB_F32 = 0, NB_F64 = 1 } nb_precision;

/* nb_config.flags */
#define NB_FLAG_EXT_STREAM 1u /* ext_stream is meaningful even when NULL (the
                                 HIP null stream, e.g. torch's default stream) */
/* 2u was NB_FLAG_XCD_REMAP in ABI 1 (an XCD-aware workgroup mapping, measured useless for this
 * VALU-bound kernel -- profiles/r01/xcd_remap_ab.txt -- and removed); the bit is ignored. */
#define NB_FLAG_LDS_ONLY 4u  /* tuning/A-B: never pick the SGPR-broadcast force kernel */
#define NB_FLAG_NO_FUSE 8u   /* tuning/A-B: never pick the fused one-launch step (force kernel +
                                integrate kernel instead; bit-identical results) */
#define NB_FLAG_POISON 16u   /* validation: the j-packed step (K = 6) overwrites every partial sum with NaN
                                once its i-block is reduced, so a partial that is ever read stale (a missing
                                release / acquire between workgroups) shows as NaN instead of a small error */

#define NB_FLAG_JPK_FENCED 32u /* the j-packed step (K = 6) hands its partial sums between workgroups with a plain store +
                                  agent-scope RELEASE on the ticket (the form the compiler's memory model guarantees on any
                                  part / partition mode) instead of write-through (sc1) stores + a relaxed ticket: the
                                  conservative fallback, 2-8 us per step slower; bit-identical results */

#define NB_FLAG_NO_SYM 64u     /* tuning/A-B: never pick the symmetric force pass (K = 7: each unordered pair once, both
                                  accelerations), i.e. keep the ordered-pair kernels of ABI 2 at every size */

#define NB_FLAG_SYM_SHARD 128u  /* a SHARD handle (shard_count != 0) may take the rank form of the symmetric force pass: the rank
                                  sweeps the pair lists of its own rows only -- every unordered pair of the system is evaluated
                                  by exactly one rank -- and the ranks reduce-scatter their partial accelerations before the
                                  integrate kernel.  The reduce-scatter is the engine's own (nb_rccl_attach: in-place
                                  ncclReduceScatter; nb_multi sets this flag itself): nb_step fails with NB_ERR_STATE on such a
                                  handle until a communicator is attached.  Needs shard rows that are whole super-blocks
                                  (shard_begin, shard_count and n multiples of 1,024, or of 512); ignored otherwise */

#define NB_FLAG_WHOLE_SWEEPS 256u /* tuning/A-B: the symmetric pass cuts its wave ranges at whole chunk-sweeps (64 rotation steps), as in
                                   ABI 2.0; by default systems with few sweeps per wave cut them in quarter sweeps (variant suffix
                                   "_u4"), which evens out the SIMDs' work (N = 16,384: the longest SIMD runs 4.25 sweeps instead of 5) */

/* nb_array: selector for nb_device_ptr */
typedef enum nb_array { NB_BODIES = 0, NB_VEL = 1, NB_ACCEL = 2 } nb_array;

/*
 * Engine configuration.  Replaces: buffer creation (nbody3d.js:179-204), the
 * constants TILE_SIZE (:4) and the hard-coded softening 1e-4 (:234).
 * Zero-initialise, set struct_size = sizeof(nb_config), then fill what you need;
 * zero fields take the defaults noted.
 */
typedef struct nb_config {
    uint32_t struct_size;  /* sizeof(nb_config), for ABI evolution             */
    uint32_t n;            /* total bodies N (any 1 <= N <= 2^30; reference is only
                              defined for N % 256 == 0)        */
    uint32_t precision;    /* nb_precision; default NB_F32                     */
    uint32_t tile;         /* j-tile staged in LDS; 0 -> 256 (nbody3d.js:4)    */
    double eps2;           /* Plummer softening; 0 -> 1e-4 (nbody3d.js:234).
                              Must be > 0 (the self term relies on it)         */
    int32_t device;        /* HIP device ordinal; -1 -> current device         */
    /* i-shard owned by this handle.  The handle integrates
     * bodies [shard_begin, shard_begin+shard_count) against ALL n bodies and
     * keeps vel/accel only for its shard.  shard_count == 0 -> whole system.   */
    uint32_t shard_begin;
    uint32_t shard_count;
    /* Optional: run on a caller-owned hipStream_t (e.g. torch's current
     * stream) instead of a private one.  Used when non-NULL or when
     * NB_FLAG_EXT_STREAM is set.                                               */
    void *ext_stream;
    /* Optional: caller-owned DEVICE buffer of 4*n elements used as the
     * replicated bodies array (so a host framework can run its collective
     * directly on it).  NULL -> engine allocates.                              */
    void *ext_bodies;
    /* Force-kernel launch shape overrides for tuning; 0 -> engine heuristics.  */
    uint32_t force_variant; /* 6 decimal digits K II LL X: K = 1 scalar loop, 2 packed f32 with
                               the j-tile in LDS, 3 packed f32 with j broadcast from SGPRs,
                               4 fused one-launch step (packed, LDS tile), 5 the same with the
                               j-bodies in registers (N <= 1,024 * X; II = 02, LL = 64); II = bodies per
                               lane (01..08); LL = lanes sharing a body (01..64); X = tile
                               units (256 bodies) per LDS stage (K = 2, 4: 1, 4 or 8) or waves splitting j
                               (K = 3: 1 or 4; 5 = 4 waves with 64-bit pair loads).  E.g. 402644.
                               K = 6: fused step with two j-bodies per packed instruction streamed
                               from a pair-transposed copy of the positions (II = 01, LL = 01;
                               X = waves per workgroup splitting j: 4, 8, or 6 for 16); jsplit > 1
                               splits j over workgroups too (reduced in the same launch).
                               K = 7: the symmetric force pass (whole-system f32 handles): every unordered pair is
                               evaluated once and both accelerations accumulated; II = resident bodies per lane (08
                               or 16), LL = 01 (02 / 04 / 08: wave ranges cut in half / quarter / eighth sweeps whatever the size),
                               X = 3 / 1: wave-granular form with 1 / 2 traveling bodies per lane
                               (jsplit = waves per SIMD), X = 4: workgroup form (II = 08; jsplit = segments per
                               super-block).  E.g. 716013.
                               See nb_variant_name().                              */
    uint32_t jsplit;        /* number of j-partitions (grid.y)                  */
    uint32_t flags;         /* NB_FLAG_*                                        */
    uint32_t layer_budget_mib; /* most device memory (MiB) the symmetric pass may take for its partial-sum layers
                               (~ 6 N^2 / S bytes, S = 512 or 1,024 rows: 6.4 GB at N = 1,048,576); a system whose
                               layers would not fit runs the ordered-pair kernels instead.  0 -> a third of the
                               device's memory, at most 96 GiB (N up to ~4 M on an MI355X).  nb_create also checks
                               the memory that is FREE: a whole-system handle whose layers would not fit it falls back
                               the same way (a rank-form shard fails instead: its peers expect the reduce-scatter) */
    uint32_t reserved[4];
} nb_config;

/* Library / ABI version; callable with no device. */
uint32_t nb_abi_version(void);
uint32_t nb_abi_minor(void);   /* NB_ABI_MINOR of the library */

/* Number of visible HIP devices (0 when there is none; never fails). */
int nb_device_count(void);

/* create: nbody3d.js:179-204 (three 16*N-byte buffers + uniform block) and
 * :296-311 (pipeline + bind group).  *out is NULL on failure; the message is
 * then available from nb_last_error(NULL). */
int nb_create(const nb_config *cfg, nb_sim **out);

/* destroy: the reference never frees (util.js:72-73 commented out). */
void nb_destroy(nb_sim *s);

/* upload: queue.writeBuffer of bodyData / velData (nbody3d.js:186,193) and the
 * checkpoint restore (util.js:230-244).  bodies and vel hold 4*n elements
 * (whole system, every rank passes the same arrays); accel may be NULL -> zeros
 * (WebGPU zero-initialises accelBuffer, nbody3d.js:195-199). */
int nb_upload(nb_sim *s, const void *bodies, const void *vel, const void *accel);

/* set_params: uni.dtValue / uni.GValue + per-frame queue.writeBuffer of the
 * uniform block (nbody3d.js:470,516-517; util.js:45,53). */
int nb_set_params(nb_sim *s, double dt, double G);

/* step: `if (dt > 0)` compute pass + submit (nbody3d.js:474-480,489-490),
 * nsteps times back to back.  dt <= 0 -> no-op, state untouched (:474). */
int nb_step(nb_sim *s, uint32_t nsteps);

/* download: exportSimulation's three readBuffer() copies (util.js:163-178).
 * Blocks until all enqueued steps are done.  Any pointer may be NULL (skipped).
 * bodies receives all 4*n elements; vel/accel receive the 4*n-element arrays
 * with only this handle's shard rows filled (others untouched). */
int nb_download(nb_sim *s, void *bodies, void *vel, void *accel);

/* sync: await device idle (the mapAsync await of util.js:174). */
int nb_sync(nb_sim *s);

/* Message for the last failed call on s (s == NULL: last failed nb_create on
 * this thread).  Never NULL; valid until the next call on the same handle. */
const char *nb_last_error(nb_sim *s);

/* ---- multi-GPU support (no reference analogue) ----------- */

/* Device address of a state array (bodies: 4*n elements, replicated;
 * vel/accel: 4*shard_count elements).  Valid until the next nb_step: a fused handle
 * ping-pongs between two position buffers, a jsplit = 1 handle swaps accel buffers. */
int nb_device_ptr(nb_sim *s, int which /* nb_array */, void **out);

/* Exchange hook: called on the calling thread once per step, after the
 * integrate kernel has been ENQUEUED for this handle's shard, with the stream
 * the work was enqueued on.  The hook must make bodies[shard rows of every
 * other rank] current on that stream (an all-gather of position rows) before
 * returning control, and return 0 on success.  NULL -> no exchange (single
 * shard).  The reference has no collective; this is where RCCL plugs in. */
typedef int (*nb_exchange_fn)(void *user, void *bodies_dev, size_t elem_size,
                              uint32_t n, uint32_t shard_begin,
                              uint32_t shard_count, void *hip_stream);
int nb_set_exchange(nb_sim *s, nb_exchange_fn fn, void *user);

/* Overlapped (two-phase) exchange.  begin() is called where the one-phase hook
 * would be (after the integrate kernel is enqueued) and must START the
 * all-gather without making the engine stream wait for it; wait() is called
 * before the engine enqueues work that reads other ranks' rows and must make
 * `hip_stream` wait for the gather begin() started.  Between the two the engine
 * enqueues the next step's force work on the j-range of its OWN rows (which the
 * gather does not touch), hiding the collective behind 1/world of the force
 * pass.  Falls back to calling wait() right after begin() when the j-splits do
 * not line up with the shard boundaries.  Replaces any one-phase hook. */
typedef int (*nb_exchange_wait_fn)(void *user, void *hip_stream);
int nb_set_exchange_overlapped(nb_sim *s, nb_exchange_fn begin, nb_exchange_wait_fn wait, void *user);

/* ---- native RCCL collective, one process per GPU (SURVEY.md §8(e) "Collective") -----------
 * Instead of an exchange hook the engine itself issues the per-step all-gather of position
 * rows: ncclAllGather, in place (send pointer = bodies + rank * rows), on the engine's stream
 * right after the integrate kernel -- no host code between the kernels of a step.  The host
 * only transports the 128-byte ncclUniqueId from rank 0 to the other ranks (any channel).
 * Every rank owns the same number of rows: n == nranks * shard_count and
 * shard_begin == rank * shard_count (pad with zero-mass rows).  librccl is loaded on first
 * use (dlopen; a copy already loaded into the process -- e.g. PyTorch's -- is reused). */
#define NB_RCCL_ID_BYTES 128
#define NB_RCCL_OVERLAP 1u /* all-gather on its own stream, hidden behind the next step's force
                              work on the rank's OWN j-range (see nb_set_exchange_overlapped) */
int nb_rccl_unique_id(void *id_out /* NB_RCCL_ID_BYTES */);
int nb_rccl_attach(nb_sim *s, const void *id, int nranks, int rank, uint32_t flags);
int nb_rccl_detach(nb_sim *s);
/* Communicator facts for reports: ncclCommCount / ncclCommUserRank and RCCL's version code
 * (0 when no communicator is attached). */
int nb_rccl_info(nb_sim *s, int *nranks, int *rank, int *rccl_version);

/* ---- single-process multi-device (for hosts that cannot run one process per
 *      GPU, e.g. Node; no reference analogue) --------------------------------
 * nb_multi_* wraps n_shards shard handles: shard k owns a contiguous block of
 * rows (256-aligned; the system is padded with zero-mass rows at the origin,
 * which exert and feel exactly nothing) on device devices[k].  After every
 * step each shard pulls the other shards' new rows straight into its
 * replicated bodies array (one pull kernel per shard over peer-mapped memory,
 * ordered with HIP events; g*(g-1) device-to-device copies when peer access is
 * missing): on the fully connected xGMI fabric of an 8-GPU node every transfer
 * has its own link, which is the all-gather the topology wants.  f32 / f64
 * systems with >= 2,048 rows per shard divide the PAIRS instead of the rows (rank
 * form of the symmetric force pass, see NB_FLAG_SYM_SHARD) and reduce-scatter
 * their partial accelerations the same way before the integrate kernel.  devices == NULL
 * -> round-robin over the visible devices; several shards may share a device
 * ("virtual shards": the way the partition logic is tested on one GPU).
 * Host arrays hold the UNPADDED n rows, exactly as for a single handle. */
typedef struct nb_multi nb_multi; /* opaque */
int nb_multi_create(const nb_config *cfg, uint32_t n_shards, const int32_t *devices, nb_multi **out);
void nb_multi_destroy(nb_multi *m);
int nb_multi_upload(nb_multi *m, const void *bodies, const void *vel, const void *accel);
int nb_multi_set_params(nb_multi *m, double dt, double G);
int nb_multi_step(nb_multi *m, uint32_t nsteps);
int nb_multi_download(nb_multi *m, void *bodies, void *vel, void *accel);
int nb_multi_sync(nb_multi *m);
const char *nb_multi_last_error(nb_multi *m);
/* Energy / momentum of the whole system: sum of the shards' nb_diagnostics (same out[5]). */
int nb_multi_diagnostics(nb_multi *m, double out[5]);
/* Name of shard 0's force-kernel variant (all shards resolve to the same shape). */
const char *nb_multi_variant_name(nb_multi *m);
/* How the shards exchange their new rows after a step:
 *   NB_MULTI_PEER  g*(g-1) hipMemcpyAsync device-to-device copies ordered by HIP events (default);
 *   NB_MULTI_RCCL  ncclCommInitAll over the shards' devices once, then per step
 *                  ncclGroupStart / in-place ncclAllGather on every shard's stream / ncclGroupEnd
 *                  (SURVEY.md §8(e)).  Needs every shard on its own device.
 *   NB_MULTI_PEER_OVERLAP  (ABI 2.3; rank form of the symmetric pass with pull kernels) the all-gather pull of step n runs on a
 *                  second stream per shard while step n + 1 sweeps the pairs whose travelers are the shard's own rows; the shard's
 *                  stream waits for its pull only in front of the sweeps that read the other shards' rows -- what NB_RCCL_OVERLAP
 *                  does across processes, here with real rows in flight between g shards (also g virtual shards on ONE device:
 *                  the multi-rank check of the overlapped protocol a one-GPU box can run).  Falls back to NB_MULTI_PEER's order
 *                  when the handle is not in the rank form or G != 1.
 * Results are bit-identical; the choice is a measured A/B. */
typedef enum nb_multi_collective { NB_MULTI_PEER = 0, NB_MULTI_RCCL = 1, NB_MULTI_PEER_OVERLAP = 2 } nb_multi_collective;
int nb_multi_set_collective(nb_multi *m, int mode /* nb_multi_collective */);
/* mode in use, communicator size (0 for NB_MULTI_PEER), RCCL version code. */
int nb_multi_collective_info(nb_multi *m, int *mode, int *nranks, int *rccl_version);

/* ---- measurement (role of TimingHelper, util.js:297-423) ------------------- */

/* When enabled, each nb_step records HIP events around every force-kernel and
 * integrate-kernel launch on the engine stream.  nb_kernel_times then blocks
 * for them and returns the averages since the last call (milliseconds) and the
 * number of launches averaged; it resets the accumulators. */
int nb_enable_timing(nb_sim *s, int on);
int nb_kernel_times(nb_sim *s, double *force_ms, double *integrate_ms,
                    uint32_t *launches);
/* Same, plus the average time of the native RCCL collectives of a step (the position all-gather and, for the
 * rank form of the symmetric pass, the reduce-scatter of the partial accelerations before it; 0 when none ran).
 * A fused one-launch step reports its whole kernel as force_ms and 0 for integrate_ms. */
int nb_step_times(nb_sim *s, double *force_ms, double *integrate_ms, double *exchange_ms,
                  uint32_t *launches);
/* The full breakdown of a step, part by part as the engine stream runs them (averages over the recorded steps,
 * milliseconds): force pass [rank form: nb_sym_reduce, ncclReduceScatter] integrate [ncclAllGather].  span_ms is
 * the time from the first event of a step to its last one, so force + sym_reduce + reduce_scatter + integrate +
 * allgather = span - (idle gaps between the kernels).  An overlapped all-gather (NB_RCCL_OVERLAP) runs on its own
 * stream and is not timed: allgathers = 0.  Set struct_size = sizeof(nb_step_timing) before the call.
 * (No reference analogue: TimingHelper times one pass, util.js:297-423; the reference is single-device.) */
typedef struct nb_step_timing {
    uint32_t struct_size;
    uint32_t launches;              /* steps averaged */
    double force_ms;                /* the force kernel (both launches of an overlapped step) */
    double sym_reduce_ms;           /* rank form: this rank's sums for every row (nb_sym_reduce) */
    double reduce_scatter_ms;       /* rank form: in-place ncclReduceScatter of those sums */
    double integrate_ms;
    double allgather_ms;            /* in-place ncclAllGather of the new positions on the engine stream */
    double span_ms;
    uint32_t reduce_scatters;       /* steps whose reduce-scatter was timed */
    uint32_t allgathers;            /* steps whose all-gather was timed */
} nb_step_timing;
int nb_step_times2(nb_sim *s, nb_step_timing *out);
/* Runs ONLY the integrate kernel `reps` times back to back on the state as it stands (the
 * force sums are whatever the last force pass left; first zeroed if none ran) and returns
 * the average launch time: the memory-bound kernel measured on its own at sizes where a
 * full O(N^2) step would take minutes (N >= 4M: state no longer cache-resident).  The
 * particle state is garbage afterwards -- measurement only; not available on a fused handle.
 * A rank-form handle (NB_FLAG_SYM_SHARD shard, or a whole system whose ring distances go in passes) runs what its step runs:
 * the plain integrate kernel on its rows of the reduced sums. */
int nb_integrate_pass(nb_sim *s, uint32_t reps, double *avg_ms);
/* Runs ONLY the force pass `reps` times back to back on the positions as they stand and returns the average time per pass
 * (a rank-form handle: both phases of the force kernel + nb_sym_reduce).  The state is left untouched (the pass writes partial
 * sums only).  Measurement: what ONE rank of an N-rank partition spends in its force pass can be timed on a single GPU
 * without a communicator -- create the shard handle (shard_begin / shard_count, NB_FLAG_SYM_SHARD), upload, call this. */
int nb_force_pass(nb_sim *s, uint32_t reps, double *avg_ms);

/* Name of the force-kernel variant a handle resolved to (for reports), e.g.
 * "f32pk_fused_lds1024_ipl2_ls64" or "f32pk_sgpr_ipl8_ws4_js8".  Valid until nb_destroy. */
const char *nb_variant_name(nb_sim *s);

/* Launch-shape facts of a handle (for reports and tests): the number of j-partitions the force pass runs
 * (grid.y), the bodies per partition, and which partitions lie ENTIRELY inside the handle's own rows -- the
 * part of the next force pass that the overlapped exchange (NB_RCCL_OVERLAP / nb_set_exchange_overlapped)
 * issues before it waits for the other ranks' rows.  own_splits == 0 means the overlapped forms degenerate to
 * begin-then-wait on this handle.  Any out pointer may be NULL. */
int nb_shape_info(nb_sim *s, uint32_t *jsplit, uint32_t *j_per_split, uint32_t *own_split0,
                  uint32_t *own_splits);

/* Planner introspection (the launch plan nb_create WOULD build for a configuration, with the symmetric pass's kernel-internal
 * plan words and tables: for reports, sizing runs and the host-side planner tests) lives in nbody3d_hip_plan.h -- nothing a host
 * that replaces nbody3d.js:179-204,470-490 needs. */

/* ---- viewer frame feed ------------------------------------------------
 * The reference's render pass reads bodyBuffer and velBuffer in place every frame
 * (nbody3d.js:408-415,482-487: billboard position + radius from (x,y,z,mass), colour from
 * length(vel.xyz), :380).  A host-side viewer gets the same two arrays without stalling the
 * step stream: nb_frame_request enqueues a small pack kernel behind the steps issued so far
 * (f32 bodies[4n] + speed[n] into a staging buffer) and the copy to pinned host memory runs
 * on a second stream beside the following steps.  nb_frame_acquire returns the newest frame
 * that has landed: pointers into engine-owned pinned memory, valid until the fourth
 * nb_frame_request after the one that produced it (a ring of four slots: the host may run
 * that far ahead of the copies before a request blocks).  On a shard handle speed[]
 * is filled for the handle's own rows only. */
int nb_frame_request(nb_sim *s);
int nb_frame_acquire(nb_sim *s, int wait, const float **bodies, const float **speed,
                     uint64_t *step_index);

/* ---- on-device diagnostics (no reference analogue) ------- */

/* out[0] = kinetic energy of this handle's shard (sum 1/2 m v^2),
 * out[1] = potential energy share of this handle's shard
 *          (-G/2 * sum_i sum_{j != i} m_i m_j / sqrt(r^2 + eps2), i in shard),
 * out[2..4] = momentum of the shard.  Accumulated in fp64 on the device. */
int nb_diagnostics(nb_sim *s, double out[5]);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NBODY3D_HIP_H */
