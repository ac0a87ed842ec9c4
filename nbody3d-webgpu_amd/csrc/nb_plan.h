// nb_plan.h -- the launch planner's interface: plain host C++ (no HIP), shared by the engine (nb_engine.hip, nb_comm.hip) and
// the kernels (nb_kernels.hip.h takes the plan structs below as kernel arguments).
//
// The reference dispatches ceil(N / 256) workgroups of one fixed kernel (nbody3d.js:296-311,478).  Here the force pass has
// several forms (ordered pairs through an LDS tile / SGPR broadcast / fused with the integrator, the symmetric pass), and
// nb_plan.cpp picks one per handle with a cost model and lays out its partitions.  Everything in here is deterministic
// host arithmetic on (n, shard, precision, flags, CU count, clock): nb_plan_query exposes it without a device and
// tests/test_planner_cpu.py checks the plans it produces.
#pragma once
#include "../../include/nbody3d_hip.h"

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace nb {
constexpr int kBlock = 256;  // threads per workgroup = reference TILE_SIZE (nbody3d.js:4,240)
constexpr int kTile = 256;   // j-bodies per LDS tile unit (nbody3d.js:229); TL units are staged at once

struct SymPlan {
    uint32_t np, nsb;          // padded rows, super-blocks
    uint32_t q;                // segments per super-block's chunk list (workgroups per super-block)
    uint32_t total_hi, total_lo;   // chunks in the list of a super-block g < n_hi (it has the antipodal partner) / of the others
    uint32_t n_hi, H;          // n_hi = nsb/2 when nsb is even, else 0; H = (nsb-1)/2
    uint32_t r_layer0, t_layer0;
};

struct SymWPlan {
    uint32_t np, nsb, W;
    uint32_t total_hi, total_lo, n_hi, H;
    uint32_t r_layer0, t_layer0;
    uint32_t L;                 // chunk-sweeps of this handle: n_hi * total_hi + (nsb - n_hi) * total_lo + zc
    uint32_t zc;                // real chunks of the short block Z behind the nsb whole super-blocks of the ring (0: n is a multiple of the
                                // super-block): every whole super-block sweeps them, Z only its own (nb_plan.cpp::lay_out_symw)
    uint32_t ups;               // work units per chunk-sweep (1, 2, 4 or 8): the W wave ranges are floor/ceil-equal in UNITS of
                                // 64 / ups rotation steps, so a sweep may be shared by consecutive waves.  The wave that runs a
                                // sweep's steps from 0 stores its traveler sums in the sweep's traveler layer; a wave that starts
                                // mid-sweep stores them in its own SPILL row (one per wave), which K2 adds through the per-chunk
                                // spill lists that follow the {first wave, count} table: {offset, count} per traveler chunk, then
                                // the wave numbers
};
// The rank form of the pass (NB_FLAG_SYM_SHARD: the handle keeps the super-blocks [g0, g1) of its own rows resident and sweeps THEIR
// chunk lists), in TWO phases so that the part that needs nothing from the other ranks can run while their rows are still on the
// wire (NB_RCCL_OVERLAP):
//   phase A  the sweeps whose travelers are the rank's OWN rows: ring targets g + 1 + d < g1, and the resident-only sweeps of
//            each super-block's own chunks -- about 1 / ranks of the work;
//   phase B  the rest: travelers from the other ranks' rows.
// Each phase lays its sweeps end to end (prefixA / prefixB: first position of every own super-block's part) and cuts them into its
// own WA / WB floor/ceil-equal wave ranges; waves [0, WA) run phase A, [WA, WA + WB) phase B -- in ONE launch, or in two with the
// wait for the all-gather between them: the same waves do the same sweeps either way, so the results are bit-identical.
// A wave's resident sums of super-block g go to layer r_layer0 + (wave - first A wave of g) or rb_layer0 + (wave - first B wave of g);
// table: {first A wave, A waves, first B wave (counted from WA), B waves} per super-block, then prefixA[g1 - g0 + 1], prefixB[g1 - g0 + 1].
// ups: work units per sweep, as in SymWPlan -- each phase's wave ranges are floor/ceil-equal in units of 64 / ups rotation steps; a
// wave that starts inside a sweep keeps that sweep's traveler sums in its spill row (rows of wave w at w * 64; B waves numbered from
// WA), and the per-chunk spill lists ({offset, count} per 64-row chunk, then the wave numbers) follow the prefix tables.
struct SymRankPlan {
    uint32_t np, nsb;
    uint32_t total_hi, total_lo, n_hi, H;
    uint32_t r_layer0, rb_layer0, t_layer0;
    uint32_t g0, g1;
    uint32_t LA, LB, WA, WB;
    uint32_t ups;
};
}  // namespace nb

namespace nbp {

enum Kind { kScalar = 1, kPkLds = 2, kPkSgpr = 3, kFused = 4, kDirect = 5, kJpk = 6, kSym = 7 };   // kDirect: fused, registers only (x = MAXJ/16)
struct Shape { int kind, ipl, ls, x; };   // the force_variant digits K II LL X; x: tile units (LDS kinds), j-splitting waves (SGPR kind), form (kSym)

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline int sgpr_ws(int x) { return x == 5 ? 4 : x; }   // SGPR kind: x = 5 is WS = 4 with 64-bit pair loads
inline int jpk_ws(int x) { return x == 6 ? 16 : x; }    // j-packed kind: x = 6 is 16 waves per workgroup

// Is there a kernel instantiation for this shape?  (Mirrors kernel_of in nb_engine.hip, which holds the function pointers;
// nb_create and nb_plan_query fail loudly if the two ever disagree.)
bool shape_exists(bool f64, const Shape& sh);
// i-bodies per workgroup (symmetric pass: rows per super-block)
uint32_t ipb_of(const Shape& sh);

struct PlanInput {
    uint32_t n = 0, sb = 0, sc = 0;     // bodies; this handle's rows [sb, sb + sc)
    bool f64 = false;
    nb_config cfg{};                     // as normalised by the entry point (force_variant, jsplit, flags, shard_count, ext_bodies)
    int n_cu = 256;
    double clock_hz = 2.4e9;
    double device_mem = 288.0e9;         // bytes of device memory (what cfg.layer_budget_mib == 0 takes a third of); MI355X when unknown
    // resident workgroups per CU of a shape's kernel at the given block size, 0 = unknown (no device): the model's defaults apply
    std::function<int(const Shape&, int)> occupancy;
};

// What nb_create allocates and launches by.  Field names follow nb_sim (nb_internal.h), which copies them.
struct LaunchPlan {
    Shape sh{kPkLds, 2, 1, 1};
    int ipl = 1, ls = 1, ws = 1, tl = 1;
    bool packed = false, sgpr = false, fused = false, direct = false, jpk = false, swap_acc = false;
    uint32_t jsplit = 1, j_per_split = 0, junits = 0, own_split0 = 0, own_splits = 0;
    bool sym = false, symw = false, sym_rank = false;
    uint32_t sym_np = 0, sym_layers = 0, sym_g0 = 0, sym_g1 = 0;
    uint32_t sym_plan[16] = {0};         // nb::SymWPlan (symw: 13 words) or nb::SymPlan, as plain words
    uint32_t sym_spill_rows = 0;         // wave-granular form with ups > 1: rows of the spill buffer (W x travelers per chunk)
    uint32_t sym_pieces = 0;             // whole-system form, two waves per SIMD, whole sweeps: sweeps left to the shared queue ({unit, layer} pairs behind the wave records)
    uint32_t sym_rank_plan[16] = {0};    // rank form: nb::SymRankPlan as plain words (sym_plan then holds the SymWPlan summary: W = WA + WB, L = LA + LB)
    // The rank-form pipeline in PASSES over the ring distances (layers reused from pass to pass, nb_sym_reduce accumulating): one
    // pass for an ordinary rank; several for a whole system whose traveler layers would not fit the layer budget (sym_local: no
    // communicator -- force passes, reduce, integrate on one device; N = 4 M bodies: 100 GB of layers in one pass).
    struct SymPass {
        uint32_t plan[16];               // nb::SymRankPlan of the pass
        uint32_t tab_off;                // where the pass's tables start in sym_tab_host
        uint32_t k_lo, k_hi, d0;         // the pass's window of every super-block's ring sweeps [k_lo, k_hi) and its first ring distance
    };
    std::vector<SymPass> sym_passes;
    bool sym_local = false;
    std::vector<uint32_t> sym_tab_host;  // wave-granular form: {first wave, resident layers} per super-block [2 nsb words]; with ups > 1
                                         // followed by {offset, count} per traveler chunk [2 np / CH words] and the spill lists' wave numbers
    std::string variant;
};

LaunchPlan plan_launch(const PlanInput& in);

}  // namespace nbp
