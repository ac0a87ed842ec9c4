// nb_internal.h -- state shared by the translation units of libnbody3d_hip.so
// (nb_engine.hip: single handle; nb_comm.hip: RCCL loader, per-process collective,
// nb_multi).  Not part of the ABI.
#pragma once
#include "../../include/nbody3d_hip.h"
#include "nb_plan.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

// Kernel-exact timing: the events are handed to hipExtLaunchKernel, which stamps them at the
// kernel's own begin and end (no marker packets between the kernels of a step).
// e[0]..e[1]: force launch issued before a pending gather is waited for (or the only force
// launch, or the whole fused step); e[3]..e[4]: force launch issued after it; e[6]..e[2]: integrate;
// e[2]..e[5]: position exchange (RCCL all-gather on the engine stream; e[5] is a plain record).
// Rank form of the symmetric pass (plain records on the engine stream): e[0] force pass e[7] nb_sym_reduce e[1]
// ncclReduceScatter e[6] integrate e[2] ncclAllGather e[5]; with the overlapped gather the force pass is two launches,
// e[0]..e[7] (own-row sweeps, issued before the wait) and e[3]..e[4] (the rest), and `rs` says the e[7]/e[1]/e[6] chain is valid.
struct nb_events { hipEvent_t e[8]; bool two, xchg, rs; };

struct nb_rccl;   // nb_comm.hip

struct nb_frame_slot {
    float* h_bodies = nullptr;     // pinned host, 4*n floats
    float* h_speed = nullptr;      // pinned host, n floats
    float* d_bodies = nullptr;     // device staging
    float* d_speed = nullptr;
    hipEvent_t packed = nullptr;   // staging written (engine stream)
    hipEvent_t landed = nullptr;   // host copy complete (frame stream)
    uint64_t step = 0;
    bool in_flight = false, valid = false;
};

struct nb_sim {
    uint32_t n = 0, sb = 0, sc = 0;
    bool f64 = false;
    size_t esz = 4;
    int device = 0;
    bool no_device = false;        // nb_plan_query on a host without a GPU: the planner skips its occupancy queries
    double eps2 = 1e-4;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // bodies[cur] is the live position array.  The fused step (force + integrate in ONE launch)
    // reads bodies[cur] and writes bodies[cur ^ 1] (ping-pong: removes the in-place race of the
    // reference, nbody3d.js:283 vs :257); two-kernel steps stay in bodies[cur].
    void* bodies[2] = {nullptr, nullptr};
    int cur = 0;
    bool own_bodies = false;
    // The j-stream of the packed f32 K1 forms (LDS-tile, SGPR, fused, registers-only): rows (x, y, z, G*m), so that a pair
    // multiplies (G*m_j)*inv as the reference does (nbody3d.js:236) at no per-pair cost.  With G == 1 it IS bodies[k]
    // (no copy); otherwise gm[k] goes with bodies[k]: rebuilt by nb_gm_pack when the positions were written from outside the
    // step (upload, exchange, raw pointer) or G changed, and kept current by K2 / the fused epilogues for the rows they write.
    void* gm[2] = {nullptr, nullptr};
    bool gm_ok = false;            // gm[cur] matches bodies[cur] and gm_G
    double gm_G = 0.0;
    void* vel = nullptr;
    void* acc = nullptr;
    void* partial = nullptr;
    size_t partial_bytes = 0;      // what `partial` was allocated with (nb_create checks it against the handle's form)
    double* diag = nullptr;
    void* zero_row = nullptr;      // 64 zero bytes (LDS-DMA source for j past the range)
    uint32_t diag_blocks = 0, diag_chunk = 256;   // nb_diag: workgroups (row blocks x j-chunks), bodies per j-chunk
    double dt = 0.0, G = 0.0;
    bool params_set = false, uploaded = false;
    uint64_t steps_done = 0;
    // launch shape
    int ipl = 1, ls = 1;
    bool packed = false;   // nb_force_pk (f32 only)
    bool sgpr = false;     // nb_force_pk_sgpr: j broadcast from SGPRs instead of the LDS tile
    int ws = 1;            // SGPR kernel: waves of a workgroup that split the j-range (1 or 4)
    int tl = 1;            // LDS kernels: 256-body tiles staged at once (1 or 4)
    bool fused = false;    // nb_step_fused / nb_step_direct: K2 folded into K1's epilogue (jsplit == 1, whole system)
    bool direct = false;   // nb_step_direct: the fused step with each lane's j-bodies in registers (N <= 2,048)
    // nb_step_jpk: the fused step with the j-bodies streamed as SGPR pairs from a pair-transposed copy
    // of the positions (pairs[k] goes with bodies[k]); js > 1 splits j over workgroups that meet at a ticket
    bool jpk = false;
    void* pairs[2] = {nullptr, nullptr};
    bool pairs_ok = false;         // pairs[cur] matches bodies[cur] and pairs_G
    double pairs_G = 0.0;          // G folded into the mass lanes of pairs[cur]
    void* jpartial = nullptr;      // jsplit x 64-row blocks of partial sums (jsplit > 1)
    uint32_t* tickets = nullptr;   // one arrival counter per i-block, zero between launches
    uint32_t junits = 0;           // 4-pair units per wave
    bool poison = false;           // NB_FLAG_POISON
    bool jpk_fenced = false;       // NB_FLAG_JPK_FENCED: release-fenced ticket instead of write-through partial stores
    int acc_parity = 0;    // swap_acc: how often acc/partial have swapped roles (mod 2)
    bool swap_acc = false; // two-kernel step with jsplit == 1: K2 reads the partial as a_new and the
                           // acc/partial buffers swap roles (96 B per body)
    uint32_t jsplit = 1, j_per_split = 0;
    // nb_force_sym: the symmetric force pass (each unordered pair once, both accelerations); whole-system f32 handles.
    // bodies / gm are allocated with sym_np >= n rows (zero-mass padding); `partial` holds sym_layers x sym_np rows.
    bool sym = false;
    uint32_t sym_np = 0, sym_layers = 0;
    uint32_t sym_plan[16] = {0};   // nb::SymPlan / nb::SymWPlan, kept as plain words here (nb_comm.hip does not see the kernels' types)
    bool symw = false;             // wave-granular form (nb_force_symw): sym_plan holds a SymWPlan, sym_tab the per-super-block table
    uint32_t* sym_tab = nullptr;   // device: {first wave, resident layers} per super-block
    std::vector<uint32_t> sym_tab_host;
    void* sym_spill = nullptr;     // ups > 1: one spill row set per wave (traveler sums of the sweep a wave's range starts inside)
    uint32_t sym_spill_rows = 0;
    uint32_t sym_pieces = 0;       // sweeps in the shared queue of nb_force_symw (nb_plan.cpp::lay_out_symw)
    uint32_t* sym_queue = nullptr; // device: the queue's draw counter, zeroed in front of every force launch
    // rank form (NB_FLAG_SYM_SHARD: a shard handle whose cross-rank reduction the engine's native exchange provides): the
    // handle's own rows are the resident super-blocks [sym_g0, sym_g1); sym_A[np] = this rank's sums for EVERY row, reduce-
    // scattered across the ranks before the integrate kernel reads the rank's own rows of it
    bool sym_rank = false;
    uint32_t sym_g0 = 0, sym_g1 = 0;
    uint32_t sym_rank_plan[16] = {0};   // nb::SymRankPlan: the two phases (own-row travelers first), their wave counts and layer bases
    std::vector<nbp::LaunchPlan::SymPass> sym_passes;   // the passes over the ring distances (one, or several when the layers of one would not fit)
    uint32_t sym_pass = 0;              // the pass launch_force runs
    bool sym_local = false;             // the rank-form pipeline of a WHOLE system on this device: no communicator, no collectives
    void* sym_A = nullptr;
    std::string variant, err;
    nb_exchange_fn xfn = nullptr;
    nb_exchange_wait_fn xwait = nullptr;   // non-null: two-phase (overlapped) exchange
    void* xuser = nullptr;
    bool gather_pending = false;           // begin() called, wait() not yet
    uint32_t own_split0 = 0, own_splits = 0;   // j-splits lying entirely inside this shard's rows
    nb_rccl* rccl = nullptr;               // per-process RCCL communicator (nb_rccl_attach)
    bool timing = false;
    // HIP-graph replay of multi-step calls (launch-bound small N): kGraphChunk steps captured
    // once per (dt, G, buffer parity) and replayed
    struct graph_slot {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        double dt = 0.0, G = 0.0;
        int parity = 0;
    } graphs[2];                     // [0]: kGraphChunk steps, [1]: kGraphBig steps (a replay costs the host
                                     // ~10-16 us: amortised over 128 steps it stops showing at N ~ 1,024)
    bool graphs_ok = true;           // cleared if capture ever fails: fall back to plain launches
    std::vector<nb_events> pool;     // recycled events
    std::vector<nb_events> pending;
    size_t pool_next = 0;
    // viewer frame feed
    hipStream_t frame_stream = nullptr;
    static constexpr int kFrameSlots = 4;   // the host may run this many snapshots ahead of the copies
    nb_frame_slot frame[kFrameSlots];
    int frame_next = 0;              // slot the next request writes
    int frame_latest = -1;           // most recently requested slot
};

namespace nbi {

int fail(nb_sim* s, int code, const std::string& msg);
void set_create_error(const std::string& msg);
const std::string& create_error();

// nb_comm.hip: called by nb_step after the integrate kernel when a communicator is attached.
// begin: enqueue the in-place all-gather of this rank's rows (on the engine stream, or on the
// communicator's own stream when overlapped); wait: make the engine stream wait for it.
int rccl_exchange_begin(nb_sim* s);
// rank form of the symmetric pass: in-place ncclReduceScatter of sym_A on the engine stream (this rank's rows receive the sum)
int rccl_reduce_scatter_A(nb_sim* s);
// the two halves of a rank-form step, for nb_multi (which runs its own reduce-scatter between them)
// force pass [+ a record of the hipEvent_t `after_force`] + nb_sym_reduce.  split_at_gather: the sweeps whose travelers are the rank's
// own rows are launched first, then the engine stream waits for the pending all-gather, then the rest (bit-identical to one launch)
// `stamps` (timed steps, split_at_gather only): the two force launches are stamped at their own begin / end -- e[0]..e[7] and e[3]..e[4],
// stamps->two set -- so the wait for the gather between them is not counted as kernel time
int sym_rank_phase_a(nb_sim* s, void* after_force = nullptr, bool split_at_gather = false, nb_events* stamps = nullptr);
int sym_rank_phase_b(nb_sim* s);     // integrate kernel on the handle's rows of sym_A
int rccl_exchange_wait(nb_sim* s);
bool rccl_overlapped(const nb_sim* s);
void rccl_release(nb_sim* s);

}  // namespace nbi
