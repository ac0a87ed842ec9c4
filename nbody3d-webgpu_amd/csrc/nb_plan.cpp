// nb_plan.cpp -- the launch planner (see nb_plan.h): which force-pass form a handle runs, with how many j-partitions, and the
// symmetric pass's super-block ring, wave ranges and layer table.  Plain host C++: no HIP call in this file.
#include "nb_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace nbp {

bool pow2(int v) { return v >= 1 && (v & (v - 1)) == 0; }

bool shape_exists(bool f64, const Shape& sh)
{
    // packed LDS kernels: 2 / 4 / 8 bodies per lane, LS in {1..64}; 1,024-body stages (x = 4) for LS >= 16, 2,048-body (x = 8) for LS >= 32
    auto pk_table = [](int ipl, int ls, int tl) {
        if (ipl != 2 && ipl != 4 && ipl != 8) return false;
        if (!pow2(ls) || ls > 64) return false;
        return tl == 1 || (tl == 4 && ls >= 16) || (tl == 8 && ls >= 32);
    };
    switch (sh.kind) {
        case kScalar: return sh.ls == 1 ? (sh.ipl == 1 || sh.ipl == 2 || sh.ipl == 4) : (sh.ipl == 1 && (sh.ls == 4 || sh.ls == 16 || sh.ls == 64));
        case kPkLds:
        case kFused: return !f64 && pk_table(sh.ipl, sh.ls, sh.x);
        case kPkSgpr: return !f64 && sh.ls == 1 && (sh.x == 1 || sh.x == 4 || sh.x == 5) && (sh.ipl == 4 || sh.ipl == 8);
        case kDirect: return !f64 && sh.ipl == 2 && sh.ls == 64 && (sh.x == 1 || sh.x == 2);
        case kSym:       // ipl = residents per lane; x = 1 / 3: wave-granular form with 2 / 1 travelers per lane, x = 4: workgroup form
            if (sh.ls != 1) return false;
            if (f64) return sh.ipl == 8 && sh.x == 3;
            if (sh.x == 4) return sh.ipl == 8;
            if (sh.ipl == 4) return sh.x == 3;
            return (sh.ipl == 8 || sh.ipl == 16) && (sh.x == 1 || sh.x == 3);
        case kJpk: return !f64 && sh.ipl == 1 && sh.ls == 1 && (sh.x == 4 || sh.x == 8 || sh.x == 6);
        default: return false;
    }
}

uint32_t ipb_of(const Shape& sh)
{
    if (sh.kind == kJpk) return 64;
    if (sh.kind == kSym) return 64u * (uint32_t)sh.ipl * (sh.x == 4 ? 4u : 1u);      // rows per super-block
    if (sh.kind == kPkSgpr) return (uint32_t)(nb::kBlock / sgpr_ws(sh.x)) * sh.ipl;
    return (uint32_t)(nb::kBlock / sh.ls) * sh.ipl;
}

// force_variant codes: the 6-digit K II LL X form of nbody3d_hip.h, plus the short codes of ABI 1.
bool decode_variant(uint32_t v, Shape* out)
{
    switch (v) {
        case 1: *out = {kScalar, 1, 1, 1}; return true;
        case 2: *out = {kScalar, 2, 1, 1}; return true;
        case 4: *out = {kScalar, 4, 1, 1}; return true;
        case 14: *out = {kScalar, 1, 4, 1}; return true;
        case 116: *out = {kScalar, 1, 16, 1}; return true;
        case 164: *out = {kScalar, 1, 64, 1}; return true;
        case 22: *out = {kPkLds, 2, 1, 1}; return true;
        case 24: *out = {kPkLds, 4, 1, 1}; return true;
        case 28: *out = {kPkLds, 8, 1, 1}; return true;
        case 34: *out = {kPkSgpr, 4, 1, 1}; return true;
        case 38: *out = {kPkSgpr, 8, 1, 1}; return true;
        default: break;
    }
    if (v < 100000) return false;
    Shape sh{(int)(v / 100000), (int)(v / 1000 % 100), (int)(v / 10 % 100), (int)(v % 10)};
    if (sh.kind < kScalar || sh.kind > kSym || !pow2(sh.ls)) return false;
    if (sh.kind == kScalar) sh.x = 1;
    *out = sh;
    return true;
}

static void name_variant(LaunchPlan* s, bool f64, const Shape& sh)
{
    s->sh = sh;          // every path of plan_launch ends here: the plan carries the shape it was laid out for
    char buf[112];
    if (sh.kind == kPkSgpr)
        snprintf(buf, sizeof buf, "f32pk_sgpr_ipl%d%s_js%u", sh.ipl, sh.x == 4 ? "_ws4" : sh.x == 5 ? "_ws4p" : "", s->jsplit);
    else if (sh.kind == kFused)
        snprintf(buf, sizeof buf, "f32pk_fused_lds%d_ipl%d_ls%d", nb::kTile * sh.x, sh.ipl, sh.ls);
    else if (sh.kind == kDirect)
        snprintf(buf, sizeof buf, "f32pk_fused_regs%d_ipl%d_ls%d", 64 * 16 * sh.x, sh.ipl, sh.ls);
    else if (sh.kind == kJpk)
        snprintf(buf, sizeof buf, "f32pk_fused_jpairs_ws%d_js%u", jpk_ws(sh.x), s->jsplit);
    else if (sh.kind == kSym)
    {
        // "_u4": wave ranges cut in quarter sweeps (SymWPlan::ups, word 11); whole sweeps carry no suffix
        const int len = snprintf(buf, sizeof buf, f64 ? (s->sym_rank ? "f64_symwrank_ipl%d_j%d_w%u_r%ut%u" : "f64_symw_ipl%d_j%d_w%u_r%ut%u") : s->sym_rank ? "f32pk_symwrank_ipl%d_j%d_w%u_r%ut%u" : s->symw ? "f32pk_symw_ipl%d_j%d_w%u_r%ut%u" : "f32pk_sym_ipl%d_ws%d_q%u_r%ut%u", sh.ipl, s->symw ? (sh.x == 3 ? 1 : 2) : sh.x,
                 s->sym_plan[2], s->sym_plan[8] - s->sym_plan[7], s->sym_layers - (s->sym_plan[8] - s->sym_plan[7]));     // words 7, 8: r_layer0, t_layer0 in both plans
        if (s->symw && s->sym_plan[11] > 1 && len > 0 && (size_t)len < sizeof buf) snprintf(buf + len, sizeof buf - (size_t)len, "_u%u", s->sym_plan[11]);
    }
    else
        snprintf(buf, sizeof buf, "%s%s_lds%d_ipl%d_ls%d_js%u", f64 ? "f64" : "f32", sh.kind == kPkLds ? "pk" : "",
                 nb::kTile * (sh.kind == kPkLds ? sh.x : 1), sh.ipl, sh.ls, s->jsplit);
    s->variant = buf;
}

// Launch-shape model (inputs measured on MI355X: profiles/r01/sweep_*.txt, profiles/r02/).
//   grid = (i-blocks, jsplit) workgroups of 4 waves, all with the same amount of work, so a
//   launch runs in rounds of `slots` resident workgroups.  For every kernel shape and split count
//     t = sum over rounds [ max(compute, latency) + prologue ] / balance + what follows K1
//       compute  = loop iterations per wave * SIMD cycles per iteration * resident workgroups
//                  per CU (the waves of a SIMD share its issue port) / fill
//       latency  = tile stages per wave * ~3000 cycles (global load + LDS store + barrier)
//       balance  = 1 - 0.03 / rounds, 0.045 for the SGPR kernel (more rounds even out DVFS/tail)
//       follows  = two-kernel step: the K1 -> K2 boundary (~1.5 us) + K2 reading every split's
//                  partial back; fused step: nothing (its epilogue is the integrator)
//   and keeps the minimum.  A split is any multiple of 8 bodies >= 128 -- not a multiple of
//   the 256-body tile: the kernels run an exact trip count on the last, partial tile -- and
//   there are at most 128 splits.
struct Cand { Shape sh; double cyc_iter; };   // SIMD cycles of one wave per loop iteration (= LS j-bodies)

// Constants of the launch-shape model.  A release build compiles them in; the calibration build (`make tuning`:
// -DNB_TUNING, libnbody3d_hip_tuning.so, used by tools/fit_model.py) reads the NB_MODEL_* environment variables instead.
struct ModelKnobs {
    double tile_latency = 3000.0, prologue = 3000.0, hand_over = 350.0, lanes_scale = 1.0;
    double boundary = 4e-6;    // K1 -> K2 boundary + the K2 launch (refit on shape_scan_final_2k_16k.txt:
                               // worst regret 8.5 -> 4.8 %, mean 1.6 -> 1.0 % over 14 sizes; fused shapes now to N = 12,000)
    double sustained = 0.958;  // share of hipDeviceProp_t::clockRate the chip holds under this kernel's load
                               // (2.24-2.29 of 2.4 GHz measured, profiles/r02/rocprof_f32_default: GRBM_GUI_ACTIVE)
    double jpk_lo = 7000, jpk_hi = 12500;   // sizes at which the j-packed step is scored at all (see plan_launch)
    double tail = -1.0;        // two waves per SIMD, whole sweeps: part of an older wave's range left to the shared queue (< 0: kTailShare; 0: none) -- lay_out_symw
    double old_share = 0.0;    // two waves per SIMD: share of a pair of ranges given to the OLDER wave of a SIMD (0: kOldShareF32 / F64; 0.5: equal ranges in list order)
};
#ifdef NB_TUNING
// calibration build: read on every nb_create, so that one process can walk a grid of constants (tools/fit_model.py)
ModelKnobs model_knobs()
{
    ModelKnobs m;
    auto knob = [](const char* name, double dflt) { const char* e = getenv(name); return e && *e ? atof(e) : dflt; };
    m.tile_latency = knob("NB_MODEL_TILE_LATENCY", m.tile_latency); m.prologue = knob("NB_MODEL_PROLOGUE", m.prologue);
    m.hand_over = knob("NB_MODEL_HANDOVER", m.hand_over); m.lanes_scale = knob("NB_MODEL_LANES_SCALE", m.lanes_scale);
    m.boundary = knob("NB_MODEL_BOUNDARY", m.boundary); m.sustained = knob("NB_MODEL_SUSTAINED", m.sustained);
    m.jpk_lo = knob("NB_MODEL_JPK_LO", m.jpk_lo); m.jpk_hi = knob("NB_MODEL_JPK_HI", m.jpk_hi);
    m.old_share = knob("NB_MODEL_OLD_SHARE", m.old_share); m.tail = knob("NB_MODEL_TAIL", m.tail);
    return m;
}
#else
inline ModelKnobs model_knobs() { return ModelKnobs(); }       // release build: the compiled-in constants, no environment access
#endif


// The symmetric pass, wave-granular form (nb_force_symw<NG, 1>): predicted step time for n bodies with 2*NG residents per lane
// and k waves per SIMD.  A chunk-sweep is 64 rotation steps of NG * (16 packed + 2 transcendental) + 10 DPP issue slots; the
// loop runs at 93.5 % of that.  The L chunk-sweeps are cut into W = k * SIMDs equal ranges:
//   k = 1: ceil(L / SIMDs) sweeps per SIMD, 1.9 % slower per sweep (nothing hides a chunk's traveler loads);
//   k = 2: a wave gets floor or ceil(L / 2 SIMDs) sweeps; with a share p of ceil-waves a SIMD's two waves both round up
//          about min(1, 2p) of the time somewhere on the chip: 2 floor + 2 min(1, 2p) sweeps, 1 % over the bare rate
//          (N = 40,002: 13.64 predicted, 13.66 measured; 32,768: 8.5 / 8.9; 65,536: 33 / 32.7; 14,000: 4 / 4.1);
//   + 3.5 us of kernel fixed cost, 1.5 us per super-block a range touches, the K1 -> K2 boundary and K2's layer traffic
//   (12 B per layer and body at ~5 TB/s: the layers are Infinity-Cache resident).  16 residents per lane run ~1.5 % closer to
//   their issue count than 8 (half the rotations per pair).
// Fitted on profiles/r03/sym_variants_scan_wave_granular*.txt (N = 12,000 .. 262,144, both resident counts: within 2 %);
// 4 residents per lane (half the chunk-sweep of 8: finer rounding) win from N ~ 14,000 to 18,000 (sym_4_residents_per_lane.txt);
// k = 3 measured behind k = 2 (N = 131,072: 2,682 vs 2,615 us).
struct SymChoice { int ipl; uint32_t k; uint32_t ups; double t; };

// Bytes of partial-sum layers a symmetric handle allocates: one traveler layer per ring distance, i.e. ~ 3 * esz * N^2 / (2 S)
// (N = 1,048,576 with 1,024-row super-blocks: 6.4 GB, N = 4 M: 103 GB; it grows with N^2, so very large systems fall back to the
// ordered-pair kernels).
double sym_layer_bytes(uint32_t n, uint32_t S, size_t esz)
{
    const double nsb = std::ceil((double)n / S);
    return (nsb / 2.0 + 8.0) * nsb * S * 3.0 * (double)esz;
}
// The budget: nb_config::layer_budget_mib, or a third of the device's memory and at most 96 GiB (K2 reads every layer once per
// step: 100 GB at N = 4 M is 20 ms against 2.4 s of pair work).
double sym_layer_budget(const nb_config& cfg, double device_mem)
{
    if (cfg.layer_budget_mib) return 1048576.0 * cfg.layer_budget_mib;
    return std::min(device_mem / 3.0, 96.0 * 1073741824.0);
}

// Work units per chunk-sweep for a handle of L sweeps cut into W wave ranges.  With whole sweeps a wave gets floor or ceil(L / W)
// of them and the launch lasts as long as its longest SIMD: at N = 16,384 (4.1 sweeps per SIMD) some SIMDs run 5 -- 56 us against
// 44 us of pair work (profiles/r04/step_parts_base.txt).  Quarter sweeps (16 rotation steps) bring that to 4.25.  A system with
// dozens of sweeps per wave does not need them.
uint32_t sym_units(uint64_t L, uint32_t W, bool whole_only, bool whole_system = false)
{
    if (whole_only || W == 0) return 1;
    // The whole-system form cuts a range to 2 rotation steps when a wave has fewer than five sweeps to its name: with eighths a wave's
    // share rounds to +-1 of 8 .. 40 units.  Round 5 first took it for one-wave plans below two sweeps per wave (N = 9,000, 8 residents per
    // lane: 24.5 -> 23.75 us); with the four-word wave records it pays at every resident count and at two waves per SIMD up to N ~ 32,768
    // (14,000: 42.1 -> 41.0 us, 18,000: 61.9 -> 60.4, 22,000: 86.0 -> 84.2, 28,000: 130.7 -> 129.7, 32,768: 170.9 -> 170.2; level from
    // 40,002: profiles/r05/sym_units_scan_u32_mid_sizes.txt).  The rank form keeps eighths (its phases were fitted with them).
    if (whole_system && L < (uint64_t)5 * W) return 32u;
    // eighths below a dozen sweeps per wave (N = 16,384: 57.2 vs 58.3 us), quarters below 48
    return L < (uint64_t)12 * W ? 8u : L < (uint64_t)48 * W ? 4u : 1u;
}

// Work of one chunk-sweep on a scale where a sweep that keeps traveler sums (both sides) counts kSweepCost: what the wave ranges are
// made equal in (lay_out_symw).  A sweep over an own chunk (resident-only, no traveler sums) counts 7/8 (measured 0.88 at one and two
// waves per SIMD, profiles/r04/README.md).
constexpr uint32_t kSweepCost = 8, kOwnSweepCost = 7;
// Two waves per SIMD: the share of a pair of consecutive ranges that goes to the OLDER wave of the SIMD (lay_out_symw).  Measured with the
// arms timed in turns (profiles/r05/old_share_paired_interleaved.txt): f32 +1.0 .. +1.7 % per step from N = 40,002 to 1,048,576 at 0.90-0.93
// (0.95 falls off below 65,536), f64 +3.0 .. +4.2 % at 0.85 (0.9: +1.3 %).
constexpr double kOldShareF32 = 0.92, kOldShareF64 = 0.85;
// ... and, with whole sweeps (large systems), the last part of every older wave's range is not the wave's own: its sweeps go to a QUEUE
// that every wave draws from once its own range is done (one atomic per sweep).  The XCDs of a chip do not hold the same clock
// (1.4-2.4 % apart under this kernel, which ones differs from box to box: profiles/r05/README.md §6) and start one after the other; the
// launch used to last as long as its slowest XCD.  Every queued sweep stores its resident sums in a layer of its own, so the sums -- and
// their order in K2 -- do not depend on which wave ran it: the results stay bit-reproducible.
constexpr double kTailShare = 0.035;      // (a piece's layer and length share a table word: fewer than 65,536 of either)

SymChoice sym_estimate(uint32_t n, int n_cu, double clock, double boundary, bool f64, double layer_budget, bool whole_only)
{
    SymChoice best{0, 0, 1, 1e300};
    for (int ipl : {4, 8, 16}) {
        if (f64 && ipl != 8) continue;              // nb_force_symw64<8>: 8 residents per lane, 19 DP instructions + v_rsq_f64 per pair
        if (ipl == 4 && !whole_only) continue;      // 4 residents per lane only won by their finer rounding of WHOLE sweeps (N ~ 14,000 .. 18,000); with
                                                    // eighth sweeps they are 10-20 % behind at every size (profiles/r04/sym_units_scan_workgroup_reduce.txt)
        const uint32_t NG = (uint32_t)ipl / 2, S = 64u * (uint32_t)ipl, cps = S / 64u;
        if (ceil_div(n, S) < 4) continue;
        if (sym_layer_bytes(n, S, f64 ? 8 : 4) > layer_budget) continue;
        // the ring of whole super-blocks; a ragged N leaves a short block of zc real chunks that every super-block sweeps (lay_out_symw)
        const uint32_t nsb = n / S, zc = ceil_div(n % S, 64u);
        const uint32_t H = (nsb - 1) / 2, n_hi = (nsb & 1u) ? 0u : nsb / 2;
        const uint64_t total_hi = (uint64_t)(H + 1 + (n_hi ? 1u : 0u)) * cps + zc, total_lo = (uint64_t)(H + 1) * cps + zc;
        const uint64_t L = n_hi * total_hi + (nsb - n_hi) * total_lo + zc - ((uint64_t)nsb * cps + zc) / 8u;      // (a sweep over an own chunk counts 7/8)
        // f64: 92 issue cycles per resident and step + 14 DPP; the loop runs at 96 % of that (N = 262,144: 23.7 ms, profiles/r03/sym_f64_first.txt)
        const double t_chunk = f64 ? 64.0 * (92.0 * ipl + 56.0) / 0.96 / clock : 64.0 * (80.0 * NG + 40.0) / (NG == 8 ? 0.97 : 0.935) / clock;      // (16 residents: 0.95 until round 5's scan of N = 14,000 .. 28,000, profiles/r05/sym_units_scan_u32_mid_sizes.txt)
        const double simds = 4.0 * n_cu, per_simd = (double)L / simds;
        if (per_simd < 0.6) continue;                // (with ranges cut to 2 rotation steps a wave needs no whole sweep to its name: N = 7,000, 0.8 sweeps per SIMD)
        for (uint32_t k = 1; k <= 2; ++k) {
            if (k == 2 && per_simd < 2.0) continue;      // every wave needs a whole sweep or so of work
            // units per sweep: whole sweeps when a wave's share happens to round well (N = 11,000: 1.93 sweeps per wave, 31.3 us against
            // 32.7 with eighths), else eighths below a dozen sweeps per wave, quarters below 48 (sym_units)
            const uint32_t ups_fine = sym_units(L, (uint32_t)simds * k, whole_only, true);
            for (uint32_t ups : {1u, ups_fine}) {
                if (ups == 1 && ups_fine > 1 && (k == 2 || per_simd >= 4.0)) continue;       // one wave per SIMD and a few sweeps only: with two waves the
                                                                                             // rounding below is an average, good for fine units only
                // in units of 1 / ups sweep: a wave gets floor or ceil of its share; two waves of a SIMD both round up about min(1, 2p) of the time
                const double pu = per_simd * ups, pw = pu / 2.0, fl = std::floor(pw);
                const double units = k == 1 ? std::ceil(pu) * 1.042 : 2.0 * fl + 2.0 * std::min(1.0, 2.0 * (pw - fl));
                const double sweeps = units / ups;
                const double segs = per_simd / k / (double)total_lo + 1.0;             // super-blocks a wave's range touches
                const double spill = ups > 1 ? simds * k * 64.0 / n : 0.0;             // spill rows K2 adds per body (one 64-row spill per wave)
                const double layers = (double)(H + 1) + (double)total_hi * k / per_simd / 4.0 + 1.5 + spill;     // traveler + resident layers (one per workgroup of four waves) K2 reads per body
                // refitted on profiles/r04/sym_units_scan_workgroup_reduce.txt (N = 9,000 .. 40,002, 8 and 16 residents per lane, one and two
                // waves per SIMD: rms 1.6 %): the layers cost next to nothing since a workgroup's waves add their resident sums up in LDS;
                // two waves of 16 residents per SIMD pay ~2 us for their second set of resident loads; a range cut inside sweeps costs
                // ~1.5 us (the travelers of the shared sweeps are loaded twice, their sums stored twice)
                // (round 5, profiles/r05/sym_small_n_scan.txt: with 8 residents per lane and 32 units per sweep the estimate sat 1.2-1.6 us over
                // the measured step from N = 8,192 to 10,000: 0.2 us for the cut there instead of 1.5; 16 residents and 32 units: 1.0 us, with the
                // loop at 0.97 of its issue count instead of 0.95 -- within 1 us from N = 9,000 to 32,768 except 20,000 / 24,000 (2.5 under),
                // profiles/r05/sym_units_scan_u32_mid_sizes.txt)
                const double t = sweeps * t_chunk + 2.3e-6 + segs * 1.66e-6 + (ups > 1 ? (ups >= 32 ? (ipl == 8 ? 0.2e-6 : 1.0e-6) : 1.5e-6) : 0.0) + (k == 2 ? (ipl == 16 ? 2.1e-6 : 0.6e-6) : 0.0) + boundary
                                 + layers * n * (f64 ? 24.0 : 12.0) / 20.0e12;
#ifdef NB_TUNING
                if (getenv("NB_MODEL_TRACE")) fprintf(stderr, "  sym_estimate n=%u: %d residents, %u waves per SIMD, %u units per sweep: %.2f us\n", n, ipl, k, ups, 1e6 * t);
#endif
                if (t < best.t) best = {ipl, k, ups, t};
                if (ups_fine == 1) break;
            }
        }
    }
    return best;
}

// Wave-granular form of the symmetric pass (nb_force_symw / nb_force_symw64), whole system: the ring of super-blocks, the chunk lists
// laid end to end, their cut into W wave ranges and the {first wave, resident layers} table of every super-block.
//   sym_k: waves per SIMD the cost model asked for (0: cfg.jsplit, else 1);  sym_ups: its units per sweep (0: sym_units)
//
// A ragged N (n % S != 0) leaves a SHORT block Z of zc real chunks behind the nsb whole super-blocks.  Z stays out of the ring: as a
// resident it would run a whole list with mostly padding in its lanes (N = 40,002: 320 sweeps for 66 real rows, 2.3 % of the step),
// and the sweeps of others over its padded chunks would be dead entries in their lists.  Instead EVERY whole super-block sweeps Z's
// zc real chunks (both sides; the traveler sums go to a row of their own, "z-row" g * zc + c of the spill buffer, which K2 adds for
// the rows of Z), and Z sweeps only its own chunks (zc sweeps at the end of the list).  List of a whole super-block g:
//   [ring sweeps: the chunks of the H (+1) super-blocks after it] [zc sweeps over Z's chunks] [cps sweeps over its own chunks]
static void lay_out_symw(LaunchPlan* s, const Shape& sh, bool f64, uint32_t n, const nb_config& cfg, int n_cu, uint32_t sym_k, uint32_t sym_ups)
{
    const uint32_t S = ipb_of(sh), J = sh.x == 3 ? 1u : 2u, CH = 64u * J, cps = S / CH;
    const uint32_t blocks = ceil_div(n, S), rem = n % S;                       // blocks of S rows the arrays are padded to; rows of the short one
    const uint32_t zc = rem ? ceil_div(rem, CH) : 0u;
    const uint32_t nsb = rem ? blocks - 1u : blocks;                            // the ring (plan_launch asks for n > S: at least one whole super-block)
    const uint32_t H = (nsb - 1) / 2, n_hi = (nsb & 1u) ? 0u : nsb / 2;
    nb::SymWPlan pl;
    pl.np = blocks * S; pl.nsb = nsb; pl.zc = zc;
    pl.total_hi = (H + 1 + (n_hi ? 1u : 0u)) * cps + zc; pl.total_lo = (H + 1) * cps + zc;
    pl.n_hi = n_hi; pl.H = H;
    const uint32_t first_lo = n_hi * pl.total_hi, first_z = first_lo + (nsb - n_hi) * pl.total_lo;
    pl.L = first_z + zc;
    auto offset_of = [&](uint32_t g) { return g <= n_hi ? g * pl.total_hi : g < nsb ? first_lo + (g - n_hi) * pl.total_lo : first_z; };
    auto total_of = [&](uint32_t g) { return g < n_hi ? pl.total_hi : g < nsb ? pl.total_lo : zc; };
    const uint32_t kw = sym_k ? sym_k : (cfg.jsplit ? cfg.jsplit : 1u);
    uint32_t W = 4u * (uint32_t)n_cu * kw;
    const uint32_t ups = (cfg.flags & NB_FLAG_WHOLE_SWEEPS) ? 1u : (sym_ups ? sym_ups : sym_units(pl.L, W, false, true));
    pl.ups = ups;
    const uint64_t Lu = (uint64_t)pl.L * ups;                  // the list in units
    if (W > Lu) W = (uint32_t)Lu;                               // never more waves than units: every wave has work, so every resident layer the table
                                                               // counts is written (the kernel's `w >= W` guard idles the rest of the last workgroup)
    pl.W = W;
    // position in the list -> super-block, place in its list, length of that list
    auto sweep_at = [&](uint32_t p, uint32_t& g, uint32_t& k, uint32_t& total) {
        if (p < first_lo) { g = p / pl.total_hi; k = p - g * pl.total_hi; total = pl.total_hi; }
        else if (p < first_z) { const uint32_t r = p - first_lo; g = n_hi + r / pl.total_lo; k = r - (g - n_hi) * pl.total_lo; total = pl.total_lo; }
        else { g = nsb; k = p - first_z; total = zc; }
    };
    // the traveler chunk of a sweep (first row) and whether it keeps traveler sums
    auto traveler_of = [&](uint32_t g, uint32_t k, uint32_t total, bool& both) {
        if (g == nsb) { both = false; return g * S + k * CH; }                              // Z over its own chunks
        const uint32_t ring = total - cps - zc;
        if (k < ring) { uint32_t tb = g + 1 + k / cps; if (tb >= nsb) tb -= nsb; both = true; return tb * S + (k % cps) * CH; }
        if (k < ring + zc) { both = true; return nsb * S + (k - ring) * CH; }               // a chunk of Z
        both = false;
        return g * S + (k - ring - zc) * CH;                                                // an own chunk
    };
    // The wave ranges: equal in WORK.  A sweep over an own chunk runs the loop without traveler sums and takes 7/8 of another (measured 0.88
    // at one and two waves per SIMD: profiles/r04/README.md); a block's list is [both-sides sweeps at 8][own chunks at 7] when the ranges
    // are cut inside sweeps.
    struct Run { uint64_t at; uint32_t len, cost; };
    std::vector<Run> runs;
    for (uint32_t g = 0; g < blocks; ++g) {
        const uint32_t total = total_of(g), own = g < nsb ? cps : zc;
        if (total > own) runs.push_back({offset_of(g), total - own, kSweepCost});
        runs.push_back({(uint64_t)offset_of(g) + total - own, own, ups > 1 ? kOwnSweepCost : kSweepCost});      // whole sweeps cannot be cut finer than the difference: even cut
    }
    uint64_t Cu = 0;
    for (const Run& r : runs) Cu += (uint64_t)r.len * ups * r.cost;
    if ((uint64_t)W * kSweepCost > Cu) { W = (uint32_t)(Cu / kSweepCost); pl.W = W; }       // a wave's share is at least the dearest unit
    // POSITIONS and WAVES.  The list is cut into W consecutive ranges ("positions", in list order); position p is run by the physical
    // wave phys[p] (= 4 * workgroup + wave in it).  With one wave per SIMD the two are the same.  With TWO waves per SIMD the first half of
    // the workgroups (one per CU: dispatched first) puts the OLDER wave on every SIMD, the second half the younger one, and the older
    // wave wins the issue arbitration whenever both are ready: with equal ranges the first half of the waves finishes at ~53 % of the
    // launch and the younger half then runs alone, at the one-wave-per-SIMD rate (N = 40,002: 130 / 244 us, profiles/r05/
    // stamps_symw.txt; s_setprio does not change who wins).  So the older wave of a SIMD gets the share of the pair work at which the
    // two END together (kOldShareF32 / kOldShareF64 -- the younger wave only fills the issue slots the older one leaves), and
    // the ranges are PAIRED: positions 2i and 2i + 1 belong to waves i and W / 2 + i, so that a super-block's list is still covered
    // by about the same number of workgroups as with equal ranges (its resident layers: one per workgroup that ends a range in it).
    const double old_share = kw == 2 && W == 8u * (uint32_t)n_cu && Cu >= (uint64_t)W * 8u * kSweepCost ? (model_knobs().old_share > 0 ? model_knobs().old_share : f64 ? kOldShareF64 : kOldShareF32) : 0.5;
    const bool paired = old_share != 0.5;
    std::vector<uint32_t> starts((size_t)W + 1), phys(W);
    {
        size_t q = 0;                                          // the run the position's first unit lies in
        uint64_t before = 0;                                   // cost of the runs before it
        for (uint32_t p = 0; p < W; ++p) {
            phys[p] = paired ? (p & 1u ? W / 2 + p / 2 : p / 2) : p;
            uint64_t target = (uint64_t)p * Cu / W;            // cost before the position
            if (paired) {
                const uint64_t lo = (uint64_t)(p / 2) * Cu / (W / 2), hi = (uint64_t)(p / 2 + 1) * Cu / (W / 2);
                target = p & 1u ? lo + (uint64_t)(old_share * (double)(hi - lo)) : lo;
            }
            while (q + 1 < runs.size() && before + (uint64_t)runs[q].len * ups * runs[q].cost <= target) { before += (uint64_t)runs[q].len * ups * runs[q].cost; ++q; }
            starts[p] = (uint32_t)(runs[q].at * ups + (target - before) / runs[q].cost);
        }
        starts[0] = 0;
        starts[W] = (uint32_t)Lu;
        for (uint32_t p = 1; p <= W; ++p)
            if (starts[p] <= starts[p - 1]) starts[p] = starts[p - 1] + 1;      // (never an empty range: W <= Cu / kSweepCost leaves room)
        starts[W] = (uint32_t)Lu;
    }
    auto start_of = [&](uint32_t p) { return (uint64_t)starts[p]; };
    auto pos_of = [&](uint64_t u) { return (uint32_t)(std::upper_bound(starts.begin(), starts.begin() + W, (uint32_t)u) - starts.begin()) - 1u; };
    // the queued tails (paired ranges of whole sweeps only): own_end[p] = where position p's OWN part ends
    std::vector<uint32_t> own_end(starts.begin() + 1, starts.end());
    struct Piece { uint32_t at, len; };
    std::vector<Piece> pieces;                                  // the queued pieces (whole sweeps), in queue order
    {
        const double tail = model_knobs().tail < 0 ? kTailShare : model_knobs().tail;
        if (paired && ups == 1 && tail > 0) {
            // a tail is cut into pieces of a third of what is left of it (17 sweeps: 6, 4, 3, 2, 1, 1): the queue hands out every wave's
            // first piece, then every wave's second ... -- long pieces while there is much to do, single sweeps at the end, when the
            // length of a piece is what the waves' ends differ by.  A piece stays inside one super-block's list (its resident sums
            // have one layer): one that would cross a list's end is two.
            std::vector<std::vector<Piece>> by_rank;
            for (uint32_t p = 0; p < W; p += 2) {
                const uint32_t len = starts[p + 1] - starts[p];
                uint32_t t = std::min(len - 1u, (uint32_t)(tail * len + 0.5));
                own_end[p] = starts[p + 1] - t;
                uint32_t at = own_end[p];
                for (size_t j = 0; t > 0; ++j) {
                    uint32_t sz = (t + 2u) / 3u;
                    uint32_t g, k, total;
                    sweep_at(at, g, k, total);
                    sz = std::min(sz, total - k);                // to the end of g's list at most
                    if (by_rank.size() <= j) by_rank.resize(j + 1);
                    by_rank[j].push_back({at, sz});
                    at += sz; t -= sz;
                }
            }
            for (const auto& r : by_rank) pieces.insert(pieces.end(), r.begin(), r.end());
        }
    }
    const uint32_t nch = pl.np / CH, zrows = nsb * zc;
    // the table: {first position's wave, resident layers} per block of S rows (Z last); FOUR words per physical wave -- {first unit, end of
    // its own part, resident layer of the super-block that part ends in, spill row}: one 16-byte scalar load; then (ups > 1) the spill
    // lists, or (queued tails) {first unit, resident layer | sweeps << 16} per queued piece
    const size_t waves0 = 2 * (size_t)blocks, spill0 = waves0 + 4 * (size_t)W;
    s->sym_tab_host.assign(spill0 + (ups > 1 ? 2 * (size_t)nch : 2 * pieces.size()), 0);
    for (uint32_t p = 0; p < W; ++p) {
        uint32_t* rec = &s->sym_tab_host[waves0 + 4 * (size_t)phys[p]];
        rec[0] = starts[p]; rec[1] = own_end[p];
    }
    std::vector<uint32_t> by_place(pieces.size()), piece_layer(pieces.size(), 0);      // the pieces in list order; piece -> its resident layer
    for (size_t e = 0; e < pieces.size(); ++e) by_place[e] = (uint32_t)e;
    std::sort(by_place.begin(), by_place.end(), [&](uint32_t x, uint32_t y) { return pieces[x].at < pieces[y].at; });
    size_t next_piece = 0;                                      // (the super-blocks are walked in list order too)
    uint32_t max_r = 1;
    for (uint32_t g = 0; g < blocks; ++g) {
        const uint64_t off = (uint64_t)offset_of(g) * ups, end = off + (uint64_t)total_of(g) * ups;
        const uint32_t first = pos_of(off), last = pos_of(end - 1);
        // resident layers of g: the waves whose own part ENDS in g's list add their sums up per workgroup of four (in LDS) -- one layer
        // per workgroup, numbered in list order -- then one layer per queued sweep of g's list, and the last position, if its own part
        // goes on into g + 1, stores its part on its own (g's LAST layer)
        const uint32_t goes_on = own_end[last] > end ? 1u : 0u;
        uint32_t layers = 0;
        std::vector<uint32_t> wg_seen;                          // (a handful of workgroups per super-block)
        for (uint32_t p = first; p + goes_on <= last; ++p) {
            if (std::min<uint64_t>(own_end[p], end) <= std::max<uint64_t>(starts[p], off)) continue;      // only queued sweeps of this position lie in g
            const uint32_t wg = phys[p] >> 2;
            size_t at = 0;
            while (at < wg_seen.size() && wg_seen[at] != wg) ++at;
            if (at == wg_seen.size()) wg_seen.push_back(wg);
            s->sym_tab_host[waves0 + 4 * (size_t)phys[p] + 2] = (uint32_t)at;
        }
        layers = (uint32_t)wg_seen.size();
        for (; next_piece < by_place.size() && pieces[by_place[next_piece]].at < end; ++next_piece) piece_layer[by_place[next_piece]] = layers++;
        layers += goes_on;
        s->sym_tab_host[2 * g] = phys[first]; s->sym_tab_host[2 * g + 1] = layers;
        if (layers > max_r) max_r = layers;
    }
    for (size_t e = 0; e < pieces.size(); ++e) {                // {first unit, resident layer | sweeps << 16}
        s->sym_tab_host[spill0 + 2 * e] = pieces[e].at;
        s->sym_tab_host[spill0 + 2 * e + 1] = piece_layer[e] | pieces[e].len << 16;
    }
    s->sym_pieces = (uint32_t)pieces.size();
    // The spill buffer: the z-rows (whole super-block g's sums for chunk c of Z: row g * zc + c), then the spill rows of the waves.
    s->sym_spill_rows = zrows * CH;
    if (ups > 1) {
        // spill rows: a wave whose range starts inside a sweep keeps that sweep's traveler sums in a spill row of its own; K2 adds
        // them to the rows of the sweep's traveler chunk.  The rows are numbered chunk by chunk (list order inside a chunk), so K2
        // reads rows [first, first + count) of its chunk: {first, count} per chunk, then the wave numbers in row order (for the tests);
        // a wave's own row is word 3 of its record.
        struct Spill { uint32_t chunk, wave; };
        std::vector<Spill> sp;
        for (uint32_t p = 0; p < pl.W; ++p) {
            const uint64_t u = start_of(p);
            if (u % ups == 0) continue;                                         // starts a sweep
            uint32_t g, k, total;
            sweep_at((uint32_t)(u / ups), g, k, total);
            bool both;
            const uint32_t tstart = traveler_of(g, k, total, both);
            if (!both) continue;                                                // a sweep over an own chunk: no traveler sums
            sp.push_back({tstart / CH, phys[p]});
        }
        std::stable_sort(sp.begin(), sp.end(), [](const Spill& a, const Spill& b) { return a.chunk < b.chunk; });      // list order stays inside a chunk
        const size_t ids0 = spill0 + 2 * (size_t)nch;
        s->sym_tab_host.resize(ids0 + sp.size(), 0);
        for (size_t e = 0; e < sp.size(); ++e) {
            uint32_t* ent = &s->sym_tab_host[spill0 + 2 * (size_t)sp[e].chunk];
            if (ent[1] == 0) ent[0] = zrows + (uint32_t)e;
            ++ent[1];
            s->sym_tab_host[ids0 + e] = sp[e].wave;
            s->sym_tab_host[waves0 + 4 * (size_t)sp[e].wave + 3] = zrows + (uint32_t)e;
        }
        s->sym_spill_rows = (zrows + std::max<uint32_t>(1u, (uint32_t)sp.size())) * CH;
    }
    s->sym_rank = false; s->sym_g0 = 0; s->sym_g1 = blocks;
    pl.r_layer0 = 0; pl.t_layer0 = max_r;
    static_assert(sizeof(pl) <= sizeof(s->sym_plan), "LaunchPlan::sym_plan holds a SymWPlan");
    memcpy(s->sym_plan, &pl, sizeof pl);
    s->sym = true; s->symw = true; s->sym_np = pl.np; s->sym_layers = max_r + H + (n_hi ? 1u : 0u);
    s->ipl = sh.ipl; s->ls = 1; s->packed = !f64; s->sgpr = false; s->fused = false; s->direct = false; s->jpk = false;
    s->ws = sh.x; s->tl = 1;
    s->jsplit = max_r; s->j_per_split = (uint32_t)ceil_div((uint32_t)((Lu + pl.W - 1) / pl.W), ups) * 64u * J; s->swap_acc = false; s->own_split0 = 0; s->own_splits = 0;
    name_variant(s, f64, sh);
}

// Rank form (NB_FLAG_SYM_SHARD): the handle's own super-blocks [sb / S, (sb + sc) / S), their lists split into the sweeps whose
// travelers are own rows (phase A) and the rest (phase B); see nb::SymRankPlan.
// One pass of the rank form: the sweeps of the own super-blocks [g0, g1) whose position in their list lies in the window
// [k_lo, k_hi) (ring sweeps) -- plus, when `with_own_chunks`, each super-block's resident-only sweeps -- split into phase A (travelers
// inside the own rows) and phase B, each cut into its wave ranges.  Appends the pass's tables to s->sym_tab_host.
static LaunchPlan::SymPass build_rank_pass(LaunchPlan* s, nb::SymRankPlan rp, uint32_t n, uint32_t S, uint32_t Wfull, bool whole_sweeps,
                                           uint32_t k_lo, uint32_t k_hi, bool with_own_chunks, uint32_t* layers_out, uint32_t* spill_rows_out)
{
    const uint32_t cps = S / 64u, nsb = rp.nsb, n_hi = rp.n_hi;
    const uint32_t ng = rp.g1 - rp.g0;
    std::vector<uint32_t> preA(ng + 1, 0), preB(ng + 1, 0), a_lo(ng, 0), a_len(ng, 0), b_lo(ng, 0);
    for (uint32_t gi = 0; gi < ng; ++gi) {
        const uint32_t g = rp.g0 + gi;
        const uint32_t total = g < n_hi ? rp.total_hi : rp.total_lo, ring = total - cps;
        const uint32_t a = std::min(ring, (rp.g1 - 1 - g) * cps);       // ring distances d with g + 1 + d < g1: travelers inside the own rows
        a_lo[gi] = std::min(a, k_lo);
        a_len[gi] = std::min(a, k_hi) - a_lo[gi];
        b_lo[gi] = std::min(ring, std::max(a, k_lo));
        const uint32_t b_hi = std::min(ring, std::max(a, k_hi));
        preA[gi + 1] = preA[gi] + a_len[gi] + (with_own_chunks ? cps : 0u);      // ... and the super-block's own chunks (resident-only)
        preB[gi + 1] = preB[gi] + (b_hi - b_lo[gi]);
    }
    rp.LA = preA[ng]; rp.LB = preB[ng];
    // units per sweep: a rank's share is small by construction (1 / ranks of the system in each phase), so quarter sweeps unless there
    // are dozens of sweeps per wave in BOTH phases
    const uint32_t ups = whole_sweeps ? 1u : std::max(sym_units(rp.LA ? rp.LA : rp.LB, Wfull, false), sym_units(rp.LB ? rp.LB : rp.LA, Wfull, false));
    rp.ups = ups;
    rp.WA = std::min(Wfull, rp.LA);           // at least one sweep's worth of units per wave: every wave that touches a super-block adds a
    rp.WB = std::min(Wfull, rp.LB);           // resident layer to it, which nb_sym_reduce reads back (N = 65,536 over 8 ranks: 744 layers otherwise)
    const uint32_t nch = rp.np / 64u;
    const size_t tab0 = s->sym_tab_host.size(), pre0 = tab0 + 4 * (size_t)nsb, sp0 = pre0 + 2 * ((size_t)ng + 1);
    s->sym_tab_host.resize(sp0 + (ups > 1 ? 2 * (size_t)nch : 0), 0);
    uint32_t max_ra = rp.LA ? 1u : 0u, max_rb = 0;
    auto wave_of = [](uint64_t u, uint64_t Lu, uint32_t Wp) {
        uint32_t w = (uint32_t)(u * Wp / Lu);
        while (w + 1 < Wp && (uint64_t)(w + 1) * Lu / Wp <= u) ++w;
        while (w > 0 && (uint64_t)w * Lu / Wp > u) --w;
        return w;
    };
    for (uint32_t gi = 0; gi < ng; ++gi) {
        uint32_t* t = &s->sym_tab_host[tab0 + 4 * (size_t)(rp.g0 + gi)];
        const uint64_t LuA = (uint64_t)rp.LA * ups, LuB = (uint64_t)rp.LB * ups;
        if (preA[gi + 1] > preA[gi]) {
            const uint32_t fa = wave_of((uint64_t)preA[gi] * ups, LuA, rp.WA), la = wave_of((uint64_t)preA[gi + 1] * ups - 1, LuA, rp.WA);
            t[0] = fa; t[1] = la - fa + 1;
            max_ra = std::max(max_ra, t[1]);
        }
        if (preB[gi + 1] > preB[gi]) {
            const uint32_t fb = wave_of((uint64_t)preB[gi] * ups, LuB, rp.WB), lb = wave_of((uint64_t)preB[gi + 1] * ups - 1, LuB, rp.WB);
            t[2] = fb; t[3] = lb - fb + 1;
            max_rb = std::max(max_rb, t[3]);
        }
    }
    std::copy(preA.begin(), preA.end(), s->sym_tab_host.begin() + pre0);
    std::copy(preB.begin(), preB.end(), s->sym_tab_host.begin() + pre0 + ng + 1);
    *spill_rows_out = 0;
    if (ups > 1) {
        // a wave whose range starts inside a sweep spills that sweep's traveler sums: per 64-row chunk the waves to add (B waves numbered from WA)
        struct Spill { uint32_t chunk, wave; };
        std::vector<Spill> sp;
        for (int phase = 0; phase < 2; ++phase) {
            const std::vector<uint32_t>& pre = phase ? preB : preA;
            const uint64_t Lu = (uint64_t)(phase ? rp.LB : rp.LA) * ups;
            const uint32_t Wp = phase ? rp.WB : rp.WA;
            for (uint32_t w = 0; w < Wp; ++w) {
                const uint64_t u = (uint64_t)w * Lu / Wp;
                if (u % ups == 0 || (uint64_t)(w + 1) * Lu / Wp == u) continue;
                const uint32_t p = (uint32_t)(u / ups);
                const uint32_t gi = (uint32_t)(std::upper_bound(pre.begin(), pre.end(), p) - pre.begin()) - 1;
                const uint32_t g = rp.g0 + gi, jj = p - pre[gi];
                const uint32_t total = g < n_hi ? rp.total_hi : rp.total_lo, ring = total - cps;
                const uint32_t k = phase ? b_lo[gi] + jj : (jj < a_len[gi] ? a_lo[gi] + jj : ring + (jj - a_len[gi]));
                if (k >= ring) continue;                                    // resident-only sweep
                uint32_t tb = g + 1 + k / cps;
                if (tb >= nsb) tb -= nsb;
                const uint32_t tstart = tb * S + (k % cps) * 64u;
                if (tstart >= n) continue;
                sp.push_back({tstart / 64u, (phase ? rp.WA : 0u) + w});
            }
        }
        std::stable_sort(sp.begin(), sp.end(), [](const Spill& x, const Spill& y) { return x.chunk < y.chunk; });
        const size_t ids0 = sp0 + 2 * (size_t)nch;
        s->sym_tab_host.resize(ids0 + sp.size(), 0);
        for (size_t e = 0; e < sp.size(); ++e) {
            uint32_t* ent = &s->sym_tab_host[sp0 + 2 * (size_t)sp[e].chunk];
            if (ent[1] == 0) ent[0] = (uint32_t)e;
            ++ent[1];
            s->sym_tab_host[ids0 + e] = sp[e].wave;
        }
        *spill_rows_out = (rp.WA + rp.WB) * 64u;
    }
    rp.r_layer0 = 0; rp.rb_layer0 = max_ra; rp.t_layer0 = max_ra + max_rb;
    const uint32_t d0 = k_lo / cps;
    const uint32_t dmax = rp.H + (n_hi ? 1u : 0u);                         // ring distances of the system
    const uint32_t d1 = std::min(dmax, k_hi == 0xffffffffu ? dmax : k_hi / cps);
    *layers_out = rp.t_layer0 + (d1 > d0 ? d1 - d0 : 0u);
    LaunchPlan::SymPass pass;
    static_assert(sizeof(rp) == sizeof(pass.plan), "SymPass::plan holds a SymRankPlan");
    memcpy(pass.plan, &rp, sizeof rp);
    pass.tab_off = (uint32_t)tab0; pass.k_lo = k_lo; pass.k_hi = k_hi; pass.d0 = d0;
    return pass;
}

// Rank form (NB_FLAG_SYM_SHARD): the handle's own super-blocks [sb / S, (sb + sc) / S), their lists split into the sweeps whose
// travelers are own rows (phase A) and the rest (phase B); see nb::SymRankPlan.  local_passes > 0: the same pipeline for a WHOLE
// system on one device (no communicator), its ring distances processed in that many passes so that the traveler layers fit the budget.
static void lay_out_symw_rank(LaunchPlan* s, const Shape& sh, bool f64, uint32_t n, uint32_t sb, uint32_t sc, const nb_config& cfg, int n_cu,
                              uint32_t local_passes)
{
    const uint32_t S = ipb_of(sh), cps = S / 64u;                    // one traveler per lane
    const uint32_t nsb = ceil_div(n, S), H = (nsb - 1) / 2, n_hi = (nsb & 1u) ? 0u : nsb / 2;
    nb::SymRankPlan rp;
    memset(&rp, 0, sizeof rp);
    rp.np = nsb * S; rp.nsb = nsb;
    rp.total_hi = (H + 1 + (n_hi ? 1u : 0u)) * cps; rp.total_lo = (H + 1) * cps;
    rp.n_hi = n_hi; rp.H = H;
    rp.g0 = local_passes ? 0u : sb / S; rp.g1 = local_passes ? nsb : (sb + sc) / S;
    const uint64_t Lown = (uint64_t)(rp.g1 - rp.g0) * rp.total_lo;
    const uint32_t kw = cfg.jsplit ? cfg.jsplit : (Lown >= 16u * (uint64_t)n_cu ? 2u : 1u);
    const uint32_t Wfull = 4u * (uint32_t)n_cu * kw;
    const bool whole_sweeps = (cfg.flags & NB_FLAG_WHOLE_SWEEPS) != 0 || local_passes > 1;       // spill rows belong to one pass: several passes keep whole sweeps
    s->sym_tab_host.clear();
    s->sym_passes.clear();
    uint32_t layers = 0, spill_rows = 0;
    const uint32_t dmax = H + (n_hi ? 1u : 0u), passes = local_passes ? local_passes : 1u;
    const uint32_t per = ceil_div(dmax, passes);                      // ring distances per pass
    for (uint32_t q = 0; q < passes; ++q) {
        const uint32_t k_lo = q * per * cps, k_hi = q + 1 == passes ? 0xffffffffu : (q + 1) * per * cps;
        uint32_t lay = 0, sp = 0;
        s->sym_passes.push_back(build_rank_pass(s, rp, n, S, Wfull, whole_sweeps, k_lo, k_hi, q + 1 == passes, &lay, &sp));
        layers = std::max(layers, lay); spill_rows = std::max(spill_rows, sp);
    }
    memcpy(&rp, s->sym_passes[0].plan, sizeof rp);
    memcpy(s->sym_rank_plan, s->sym_passes[0].plan, sizeof s->sym_rank_plan);
    s->sym_spill_rows = spill_rows;
    s->sym_local = local_passes != 0;
    // the SymWPlan summary the reports read (first pass)
    nb::SymWPlan pl;
    uint32_t Lsum = 0;
    for (const auto& ps : s->sym_passes) Lsum += ps.plan[11] + ps.plan[12];
    pl.np = rp.np; pl.nsb = nsb; pl.W = rp.WA + rp.WB; pl.total_hi = rp.total_hi; pl.total_lo = rp.total_lo; pl.n_hi = n_hi; pl.H = H;
    pl.r_layer0 = 0; pl.t_layer0 = rp.t_layer0; pl.L = Lsum; pl.ups = rp.ups;
    pl.zc = 0;
    memcpy(s->sym_plan, &pl, sizeof pl);
    s->sym_rank = true; s->sym_g0 = rp.g0; s->sym_g1 = rp.g1;
    s->sym = true; s->symw = true; s->sym_np = rp.np; s->sym_layers = layers;
    s->ipl = sh.ipl; s->ls = 1; s->packed = !f64; s->sgpr = false; s->fused = false; s->direct = false; s->jpk = false;
    s->ws = sh.x; s->tl = 1;
    s->jsplit = rp.WA + rp.WB; s->j_per_split = ceil_div(Lsum, std::max(1u, rp.WA + rp.WB)) * 64u; s->swap_acc = false;
    s->own_split0 = 0; s->own_splits = local_passes ? 0u : rp.WA;          // the waves whose sweeps read the rank's own rows only: issued before the wait for the gather
    name_variant(s, f64, sh);
    if (local_passes) { char buf[24]; snprintf(buf, sizeof buf, "_p%u", local_passes); s->variant += buf; }
}

// Workgroup form (nb_force_sym<4,4,2>, the A/B arm): Q segments per super-block's chunk list.
static void lay_out_sym_wg(LaunchPlan* s, const Shape& sh, bool f64, uint32_t n, const nb_config& cfg, int n_cu)
{
    // workgroup form: super-blocks of S rows on a ring; workgroup (g, q) sweeps segment q of Q of g's chunk list (the H or
    // H+1 super-blocks ahead on the ring, then g itself in resident-only mode).  cfg.jsplit, if given, is Q.
    // One wave per SIMD already issues this loop at ~90 % of its rate (profiles/r03/symsweep_*.txt), so what matters is
    // that every CU gets the same number of equal workgroups: time ~ ceil(nsb * Q / CUs) * ceil(chunks / Q) sweeps.
    const uint32_t S = ipb_of(sh), cps = S / 128u;
    const uint32_t nsb = ceil_div(n, S), H = (nsb - 1) / 2, n_hi = (nsb & 1u) ? 0u : nsb / 2;
    const uint32_t total_hi = (H + 1 + (n_hi ? 1u : 0u)) * cps, total_lo = (H + 1) * cps;     // + the resident-only chunks of g itself
    uint32_t q = cfg.jsplit;
    if (q == 0) {
        double best = 1e300;
        for (uint32_t c = 1; c <= total_hi && c <= 512; ++c) {
            const double t = (double)ceil_div(nsb * c, (uint32_t)n_cu) * (ceil_div(total_hi, c) + 0.15);   // + ~15 % of a sweep per workgroup
            if (t < best * 0.999) { best = t; q = c; }
        }
    }
    if (q > total_hi) q = total_hi;
    if (q < 1) q = 1;
    nb::SymPlan pl;
    pl.np = nsb * S; pl.nsb = nsb; pl.q = q; pl.total_hi = total_hi; pl.total_lo = total_lo;
    pl.n_hi = n_hi; pl.H = H; pl.r_layer0 = 0; pl.t_layer0 = q;
    static_assert(sizeof(pl) <= sizeof(s->sym_plan), "LaunchPlan::sym_plan holds a SymPlan");
    memcpy(s->sym_plan, &pl, sizeof pl);
    s->sym = true; s->sym_np = pl.np; s->sym_layers = q + H + (n_hi ? 1u : 0u);
    s->ipl = sh.ipl; s->ls = 1; s->packed = true; s->sgpr = false; s->fused = false; s->direct = false; s->jpk = false;
    s->ws = sh.x; s->tl = 1;
    s->jsplit = q; s->j_per_split = ceil_div(total_hi, q) * 128u; s->swap_acc = false; s->own_split0 = 0; s->own_splits = 0;
    name_variant(s, f64, sh);
}

LaunchPlan plan_launch(const PlanInput& in)
{
    LaunchPlan plan;
    LaunchPlan* const s = &plan;
    const nb_config& cfg = in.cfg;
    const int n_cu = in.n_cu;
    const double clock_hz = in.clock_hz;
    const bool f64 = in.f64;
    const size_t esz = f64 ? 8 : 4;
    const uint32_t sc = in.sc, n = in.n, sb = in.sb;
    const double layer_budget = sym_layer_budget(cfg, in.device_mem);
    const uint32_t kMaxSplit = 128, kMinSplitLen = 128;
    const ModelKnobs mk = model_knobs();
    const double kTileLatency = mk.tile_latency, kPrologue = mk.prologue, kClock = clock_hz * mk.sustained;
    const double kHandOver = mk.hand_over, kLanesScale = mk.lanes_scale, kBoundary = mk.boundary;
    auto split_len = [&](uint32_t js) { return ceil_div(ceil_div(n, js), 8u) * 8u; };
    // an explicit shard (even one that covers every row: a 1-rank distributed run) keeps the
    // two-kernel step, whose position array stays put for the exchange
    const bool whole = sc == n && sb == 0 && cfg.shard_count == 0;
    const bool may_fuse = !f64 && whole && !cfg.ext_bodies && !(cfg.flags & NB_FLAG_NO_FUSE);

    std::vector<Cand> cands;
    if (f64) {
        // 15 DP instructions (4 cycles) + v_rsq_f64 (16) per pair, +7 % for LDS reads / loop (ubench3)
        cands = {{{kScalar, 4, 1, 1}, 326}, {{kScalar, 2, 1, 1}, 172}, {{kScalar, 1, 1, 1}, 92},
                 {{kScalar, 1, 4, 1}, 98}, {{kScalar, 1, 16, 1}, 100}, {{kScalar, 1, 64, 1}, 104}};
    } else {
        // packed: 64 issue cycles per (j, 2 i-bodies) + ~6 % LDS/loop overhead
        for (int ipl : {8, 4, 2})
            for (int ls : {1, 2, 4, 8, 16, 32, 64})
                for (int tl : {1, 4, 8}) {
                    if (tl == 4 && ls < 16) continue;
                    if (tl == 8 && ls < 32) continue;
                    // 2 per lane: lane-sharing shapes carry no mass v_mov any more (0.97, refit on shape_scan_after_hi_broadcast.txt);
                    // one lane per body keeps the round-2 figure (1.05): 0.97 there pulled N = 12,000 .. 32,768 off the SGPR kernel (-1..-6 %)
                    const double cyc = 34.0 * ipl * (ipl == 2 ? (ls == 1 ? 1.05 : 0.97) : 1.0);
                    cands.push_back({{kPkLds, ipl, ls, tl}, cyc});
                    if (may_fuse) cands.push_back({{kFused, ipl, ls, tl}, cyc});
                }
        // SGPR loop, bodies fetched as quads (s_load_dwordx4).  Since z is taken from the (z, m) SGPR pair by an explicit
        // low-half broadcast and the loop body is one basic block, it carries NO VALU instruction beside the pair
        // arithmetic (224 / 448 per 8 bodies at 4 / 8 bodies per lane; the mass moves by s_mov_b32).  The 64-bit
        // pair-load form (X = 5: twice the scalar requests, measured ahead only while the quad form still copied z and
        // m through VGPRs) is now behind it at every size -- N=262,144: 14.58 vs 14.70 ms, N=40,002: 351 vs 357 us
        // (profiles/r02/ab_sgpr_zbcast.txt) -- and stays selectable as an A/B arm only.
        for (int ipl : {8, 4})
            for (int ws : {1, 4}) cands.push_back({{kPkSgpr, ipl, 1, ws}, (ipl == 8 ? 32.0 : 32.2) * ipl});
        // registers-only fused step: no tile hand-over at all (64 issue cycles per j, nothing to wait for)
        if (may_fuse && n <= 1024) cands.push_back({{kDirect, 2, 64, 1}, 64.0});
        if (may_fuse && n <= 1536) cands.push_back({{kDirect, 2, 64, 2}, 64.0});   // at 2,048 the 2,048-body LDS stage is 5 % ahead
    }

    Shape sh{f64 ? kScalar : kPkLds, 2, 1, 1};
    uint32_t js = cfg.jsplit, sym_k = 0, sym_ups = 0, sym_passes = 0, ordered_js = 0;
    Shape ordered_sh{kScalar, 1, 1, 1};
    // NB_FLAG_SYM_SHARD: a rank's shard whose cross-rank reduction the engine's native exchange provides takes the RANK form of the
    // symmetric pass when its rows are whole super-blocks (1,024 rows, or 512); otherwise the flag is ignored
    int rank_ipl = 0;
    if ((cfg.flags & NB_FLAG_SYM_SHARD) && !(cfg.flags & NB_FLAG_NO_SYM) && !cfg.ext_bodies && cfg.shard_count != 0)
        for (uint32_t S : {1024u, 512u})
            if (!rank_ipl && !(f64 && S != 512u) && sb % S == 0 && sc % S == 0 && n % S == 0 && n / S >= 2 &&        // f64: 8 residents per lane only
                sym_layer_bytes(n, S, esz) * ((double)sc / (double)n) <= layer_budget) rank_ipl = (int)(S / 64u);       // a rank's layers hold its own super-blocks' rows only
    const uint32_t variant = rank_ipl ? 0u : cfg.force_variant;
    bool pinned = false;
    if (rank_ipl) { sh = {kSym, rank_ipl, 1, 3}; pinned = true; if (js == 0) js = 0xffffffffu; }     // js: placeholder, set with the plan below
    if (variant != 0) {
        Shape want;
        if (decode_variant(variant, &want)) {
            if (f64 && want.kind != kScalar && want.kind != kSym) want = {kScalar, want.ipl > 4 ? 4 : want.ipl, 1, 1};
            if ((want.kind == kFused || want.kind == kDirect) && !may_fuse && !f64) want = {kPkLds, want.ipl, want.ls, 1};   // same loop, two kernels
            if (want.kind == kDirect && n > 1024u * (uint32_t)want.x) want = {kFused, 2, 64, 4};
            if (want.kind == kJpk && !may_fuse) want = {kPkSgpr, 4, 1, 4};      // whole-system f32 handles only
            if (want.kind == kSym && want.ls > 1) { sym_ups = (uint32_t)want.ls <= 64u ? (uint32_t)want.ls : 0u; want.ls = 1; }      // K = 7: LL = 02 .. 64 pins the units per sweep
            if (want.kind == kSym && (!whole || cfg.ext_bodies || !shape_exists(f64, want) || n <= ipb_of(want) ||       // likewise; >= 2 super-blocks,
                                      sym_layer_bytes(n, ipb_of(want), esz) > std::max(layer_budget, 0.6 * in.device_mem)))    // and layers that fit (pinned: up to 60 % of the memory)
                want = f64 ? Shape{kScalar, 4, 1, 1} : Shape{kPkSgpr, 8, 1, 4};
            if (shape_exists(f64, want)) { sh = want; pinned = true; }
        }
    }
    if (!rank_ipl && (!pinned || js == 0)) {
        struct Scored { Shape sh; uint32_t q; double t; };
        std::vector<Scored> scored;
        double best_t = 1e300;
        for (const Cand& c : cands) {
            if (pinned && !(c.sh.kind == sh.kind && c.sh.ipl == sh.ipl && c.sh.ls == sh.ls && c.sh.x == sh.x)) continue;
            if (!pinned && (cfg.flags & NB_FLAG_LDS_ONLY) && c.sh.kind == kPkSgpr) continue;
            if (!shape_exists(f64, c.sh)) continue;
            int occ = in.occupancy ? in.occupancy(c.sh, nb::kBlock) : 0;
            if (occ < 1) occ = 4;
            if (occ > 8) occ = 8;
            const uint64_t slots = (uint64_t)occ * n_cu;
            const uint32_t iblocks = ceil_div(sc, ipb_of(c.sh));
            uint32_t js_hi = n / kMinSplitLen;
            if (js_hi < 1) js_hi = 1;
            if (js_hi > kMaxSplit) js_hi = kMaxSplit;
            if (c.sh.kind == kFused || c.sh.kind == kDirect) js_hi = 1;
            uint32_t js_lo = cfg.jsplit ? cfg.jsplit : 1, js_top = cfg.jsplit ? cfg.jsplit : js_hi;
            if (c.sh.kind == kFused || c.sh.kind == kDirect) { if (cfg.jsplit > 1) continue; js_lo = js_top = 1; }
            for (uint32_t q = js_lo; q <= js_top; ++q) {
                const uint32_t len = split_len(q), used = ceil_div(n, len);
                if (c.sh.kind == kPkSgpr && len / sgpr_ws(c.sh.x) < 256) continue;   // SGPR loop wants >= 256 bodies per wave
                if (c.sh.kind == kPkSgpr && c.sh.x == 5 && len / 4 < (c.sh.ipl == 8 ? 6144u : 320u)) continue;   // pair loads: long loops only
                const uint64_t blocks = (uint64_t)iblocks * used;
                if (c.sh.kind == kPkSgpr && c.sh.x == 5 && c.sh.ipl == 4 && 2 * blocks < 3 * slots) continue;   // >= 1.5 rounds
                const uint64_t full = blocks / slots, rem = blocks % slots;
                const double tile = c.sh.kind == kPkSgpr ? 256.0 : 256.0 * c.sh.x;
                const double wave_len = c.sh.kind == kPkSgpr ? (double)len / sgpr_ws(c.sh.x) : (double)len;
                const double stages = c.sh.kind == kDirect ? 0.5 : std::ceil(wave_len / tile);
                const double iters = std::ceil(wave_len / c.sh.ls);
                // a SIMD with fewer than 4 resident waves cannot keep its issue port full
                // (measured with the pure-ALU loop, profiles/r01/ubench_run1.txt, profiles/r02/ubench3_*.txt)
                // Share of the issue rate a SIMD reaches with 4 / 3 / 2 / 1 resident waves, and the hand-over
                // cost per tile stage: least-regret fit over 208 (size, shape) timings, N = 1,024 .. 65,536
                // (profiles/r02/shape_scan_run5_calibrated.txt, shape_scan_run6_tl8.txt; worst mis-pick 1.6 %).
                // SGPR loop: no barriers, no tile hand-over.  LDS tiles shared by LS > 1 lanes: 45 % of a tile
                // period was hand-over at low occupancy with register staging (ubench4_tile_phases.txt); with
                // LDS-DMA staging the per-stage charge went from 500 to 350 cycles (tools/fit_model.py over
                // profiles/r02/shape_scan_dma_{a,b}.txt: worst regret 2.9 %, mean 0.5 % at 16 sizes).  LS = 1: the round-1 figures.
                const bool sg = c.sh.kind == kPkSgpr, lanes = !sg && c.sh.ls > 1;
                auto round_cycles = [&](double per_cu) {
                    const int k = per_cu >= 4 ? 0 : per_cu >= 3 ? 1 : per_cu >= 2 ? 2 : 3;
                    static const double f_sgpr[4] = {1.0, 0.92, 0.94, 0.62}, f_lanes[4] = {0.72, 0.75, 0.85, 0.75},
                                        f_tile[4] = {1.0, 0.92, 0.82, 0.62};
                    const double fill = sg ? f_sgpr[k] : lanes ? f_lanes[k] * kLanesScale : f_tile[k];
                    return std::max(iters * per_cu * c.cyc_iter / fill, stages * kTileLatency) + kPrologue + (lanes ? stages * kHandOver : 0.0);
                };
                double cyc = full * round_cycles(occ);
                if (rem) cyc += round_cycles((double)ceil_div((uint32_t)rem, (uint32_t)n_cu));
                const double rounds = (double)full + (rem ? 1 : 0);
                // every i-block streams all n rows through L2 once per step: what rules out many
                // lanes per body at large N (f64, 64 lanes per body, N=262,144: 550 GB per step)
                const double stream_s = (double)iblocks * n * 4 * esz / 8.0e12;
                // K2 reads every split's partial back (and K1 writes it): priced at 2 TB/s so that, when the
                // balance gain is a wash (N = 262,144: 8 vs 16 splits), the smaller HBM footprint wins
                const double after = (c.sh.kind == kFused || c.sh.kind == kDirect) ? 0.0 : (double)used * sc * 4 * esz / 2.0e12 + kBoundary;
                // balance: the last round runs partly empty and the first at a lower clock.  SGPR kernel at
                // N = 262,144: 2 / 3 / 4 / 6 / 8 rounds lose 2.0 / 1.1 / 0.7 / 0.5 / 0 % (sweep_sgpr_pair_loads.txt)
                const double t = std::max(cyc / kClock, stream_s) / (1.0 - (sg ? 0.045 : 0.03) / rounds) + after;
                scored.push_back({c.sh, q, t});
                if (t < best_t) best_t = t;
            }
        }
        // The j-packed fused step (force_variant K = 6; whole-system f32 handles): 64 i-bodies per workgroup of ws
        // waves, the j-pairs split over ws waves x q workgroups.  One scalar request (4 pairs, 256 issue cycles)
        // is in flight per wave and returns after ~1,100 cycles, so a SIMD needs > 4 resident waves to stay
        // busy.  Since its partial rows go out write-through (no release fence per workgroup) its split forms are the
        // fastest ORDERED-PAIR step from N ~ 8,000 to ~ 18,000 (through the engine: 8,192 19.9 vs 20.4 us, 10,000 28.7 vs 31.7,
        // 12,000 39.1 vs 41.2, 14,000 50.8 vs 51.4, 16,384 64.5 vs 65.3; level at 20,000 and behind above and below:
        // profiles/r02/shape_scan_jpk_sc1.txt).  The automatic choice offers it from 7,000 to 12,500, where it wins by 5-10 %
        // whatever split count this model lands on; from 13,000 to 20,000 the margin is 1-2 % with the best split and the
        // model's split choice is off by more than that (size_scan_jpk_auto.txt), so the SGPR step stays there.
        // The two bounds are model constants (ModelKnobs::jpk_lo / jpk_hi).  Scored at every size instead (profiles/r03/
        // size_scan_2k_15k_jpk_no_window.txt) this cost model picks it at 5,000 / 6,000 (1-4 % behind the LDS-tile step) and at
        // 13,000 (5 % behind the symmetric pass), and wins 8 % at 14,000: its estimate is good to ~5 %, the window is where it
        // wins by more than that.
        if (may_fuse && (pinned ? sh.kind == kJpk : (!(cfg.flags & NB_FLAG_LDS_ONLY) && n >= (uint32_t)mk.jpk_lo && n <= (uint32_t)mk.jpk_hi))) {
            const uint32_t units = ((ceil_div(ceil_div(n, 2u), 4u) + 1u) & ~1u);
            for (int x : {4, 8, 6}) {
                const Shape jsh{kJpk, 1, 1, x};
                if (pinned && sh.x != x) continue;
                const int ws = jpk_ws(x);
                int occ = in.occupancy ? in.occupancy(jsh, 64 * ws) : 0;
                if (occ < 1) occ = 28 / ws;
                const uint32_t iblocks = ceil_div(n, 64u);
                const uint32_t q_lo = cfg.jsplit ? cfg.jsplit : 1, q_hi = cfg.jsplit ? cfg.jsplit : 16;
                for (uint32_t q = q_lo; q <= q_hi; ++q) {
                    const uint32_t upw = 2 * ceil_div(units / 2, (uint32_t)ws * q);
                    if (upw < 8 && !cfg.jsplit && q > 1) continue;              // at least 64 bodies per wave
                    const uint64_t blocks = (uint64_t)iblocks * q, slots = (uint64_t)occ * n_cu;
                    const uint64_t full = blocks / slots, rem = blocks % slots;
                    auto round_cycles = [&](double per_cu) {
                        const double waves_per_simd = per_cu * ws / 4.0;
                        return upw * std::max(260.0 * waves_per_simd, 1100.0) + 4000.0;
                    };
                    double cyc = full * round_cycles(occ);
                    if (rem) cyc += round_cycles((double)ceil_div((uint32_t)rem, (uint32_t)n_cu));
                    const double rounds = (double)full + (rem ? 1 : 0);
                    const double t = cyc / kClock / (1.0 - 0.03 / rounds) + (q > 1 ? 1.0e-6 : 0.0);
                    scored.push_back({jsh, q, t});
                    if (t < best_t) best_t = t;
                }
            }
        }
        // Among the shapes within 0.4 % of the best estimate take the one with the FEWEST j-splits:
        // measured at N = 262,144, 8 / 16 / 32 splits differ by 0.1-0.3 % in step time
        // (profiles/r02/sweep_jsplit_n262144.txt) while every split is another partial array
        // written by K1 and read back by K2.
        const Scored* pick = nullptr;
        for (const Scored& c : scored) {
            if (c.t > best_t * 1.004) continue;
            if (!pick || c.q < pick->q || (c.q == pick->q && c.t < pick->t)) pick = &c;
        }
        if (pick) { if (!pinned) sh = pick->sh; js = pick->q; }
        // the symmetric pass (whole-system f32 handles; every unordered pair once): from N ~ 14,000 up it beats every
        // ordered-pair shape above (N = 16,384: 61.7 vs 65.7 us, 40,002: 270 vs 357 us, 262,144: 10.5 vs 14.6 ms)
        if (!pinned && whole && !cfg.ext_bodies && !(cfg.flags & (NB_FLAG_NO_SYM | NB_FLAG_LDS_ONLY)) && n >= 6144) {
            const SymChoice sc2 = sym_estimate(n, n_cu, kClock, kBoundary, f64, layer_budget, (cfg.flags & NB_FLAG_WHOLE_SWEEPS) != 0);
#ifdef NB_TUNING
            if (getenv("NB_MODEL_TRACE"))
                fprintf(stderr, "plan n=%u: ordered-pair estimate %.2f us, symmetric %.2f us (%d residents, %u waves per SIMD)\n", n,
                        1e6 * (pick ? pick->t : best_t), 1e6 * sc2.t, sc2.ipl, sc2.k);
#endif
            // (the j-packed step runs 2.1-2.9 us behind its estimate from N = 8,192 to 12,000: profiles/r04/sym_units_scan_workgroup_reduce.txt)
            // (... and 3.1-3.5 us at 7,000 / 7,500: profiles/r05/sym_small_n_scan.txt)
            const double ordered_t = (pick ? pick->t : best_t) + (pick && pick->sh.kind == kJpk ? 3.4e-6 : 0.0);
            if (sc2.ipl && sc2.t < 0.98 * ordered_t) { sh = {kSym, sc2.ipl, 1, 3}; sym_k = sc2.k; sym_ups = sc2.ups; }     // a clear win only: both estimates are good to ~3 %
            else if (!sc2.ipl) {
                // no resident count whose layers fit the budget (they grow with N^2: 103 GB at 4 M bodies): the rank-form pipeline on
                // this one device, its ring distances in PASSES that reuse the layers -- still every unordered pair once (~80 % of the
                // roofline) instead of the ordered-pair kernels' 60 %
                const uint32_t S = f64 ? 512u : 1024u, nsb = ceil_div(n, S);
                const double full = sym_layer_bytes(n, S, esz);
                if (nsb >= 8 && nsb <= 32768 && full > layer_budget) {          // up to 32 M bodies: the tables are O(super-blocks) per pass
                    const uint32_t dmax = (nsb - 1) / 2 + ((nsb & 1u) ? 0u : 1u);
                    uint32_t passes = (uint32_t)std::ceil(full / (0.75 * layer_budget));
                    passes = std::min(std::min(std::max(passes, 2u), dmax), 64u);
                    ordered_sh = sh; ordered_js = js;                           // what to fall back to if even one distance per pass does not fit
                    sh = {kSym, (int)(S / 64u), 1, 3};
                    sym_passes = ceil_div(dmax, ceil_div(dmax, passes));        // no empty pass
                }
            }
        }
    }
    if (js < 1) {   // pinned shape outside the model's candidate list: fill ~4096 workgroups
        js = ceil_div((uint32_t)n_cu * 16, ceil_div(sc, ipb_of(sh)));
        const uint32_t hi = n / kMinSplitLen < 1 ? 1 : (n / kMinSplitLen > kMaxSplit ? kMaxSplit : n / kMinSplitLen);
        if (js > hi) js = hi;
        if (js < 1) js = 1;
    }
    if (sh.kind == kFused || sh.kind == kDirect) js = 1;
    if (sh.kind == kSym && sym_passes) {
        // the layers actually needed (traveler layers of a pass + the resident layers of its waves) against the budget: more passes
        // until they fit; if one distance per pass still does not, the ordered-pair kernels
        const uint32_t S = ipb_of(sh), nsb = ceil_div(n, S), dmax = (nsb - 1) / 2 + ((nsb & 1u) ? 0u : 1u);
        for (int tries = 0; tries < 6; ++tries) {
            plan = LaunchPlan();
            lay_out_symw_rank(s, sh, f64, n, sb, sc, cfg, n_cu, sym_passes);
            if (3.0 * (double)esz * s->sym_np * s->sym_layers <= layer_budget) return plan;
            if (sym_passes >= dmax || sym_passes >= 256u) break;
            sym_passes = std::min(std::min(dmax, 256u), sym_passes * 2);
            sym_passes = ceil_div(dmax, ceil_div(dmax, sym_passes));
        }
        plan = LaunchPlan();
        sh = ordered_sh; js = ordered_js;
    }
    if (sh.kind == kSym && rank_ipl) { lay_out_symw_rank(s, sh, f64, n, sb, sc, cfg, n_cu, 0); return plan; }
    if (sh.kind == kSym && sh.x != 4) { lay_out_symw(s, sh, f64, n, cfg, n_cu, sym_k, sym_ups); return plan; }
    if (sh.kind == kSym) { lay_out_sym_wg(s, sh, f64, n, cfg, n_cu); return plan; }
    s->ipl = sh.ipl; s->ls = sh.ls;
    s->packed = sh.kind != kScalar; s->sgpr = sh.kind == kPkSgpr;
    s->fused = sh.kind == kFused || sh.kind == kDirect || sh.kind == kJpk;
    s->direct = sh.kind == kDirect;
    s->jpk = sh.kind == kJpk;
    s->ws = (sh.kind == kPkSgpr || sh.kind == kJpk) ? sh.x : 1;
    s->tl = (sh.kind == kPkLds || sh.kind == kFused || sh.kind == kDirect) ? sh.x : 1;
    if (s->jpk) {
        // j in whole 4-pair units, an even number per wave; splits that would be empty are dropped
        const uint32_t units = ((ceil_div(ceil_div(n, 2u), 4u) + 1u) & ~1u);
        if (js > 64) js = 64;
        s->junits = 2 * ceil_div(units / 2, (uint32_t)jpk_ws(sh.x) * js);
        s->jsplit = ceil_div(units, s->junits * (uint32_t)jpk_ws(sh.x));
        s->j_per_split = s->junits * (uint32_t)jpk_ws(sh.x) * 8;
        s->swap_acc = false; s->own_split0 = 0; s->own_splits = 0;
        name_variant(s, f64, sh);
        return plan;
    }
    s->j_per_split = split_len(js);
    s->jsplit = ceil_div(n, s->j_per_split);   // a split may end up empty after rounding
    s->swap_acc = !s->fused && s->jsplit == 1;
    // j-splits that lie ENTIRELY inside this shard's own rows: what the overlapped exchange issues before it waits for
    // the other ranks' rows (an in-place all-gather never writes the rank's own rows).  The shard need not be a whole
    // number of splits: a split that straddles a shard boundary simply belongs to the second launch.
    s->own_split0 = 0; s->own_splits = 0;
    if (sc < n) {
        const uint32_t end = sb + sc;
        const uint32_t first = ceil_div(sb, s->j_per_split);
        const uint32_t last = end == n ? s->jsplit : end / s->j_per_split;     // one past the last whole split inside
        if (last > first) { s->own_split0 = first; s->own_splits = last - first; }
    }
    name_variant(s, f64, sh);
    return plan;
}


}  // namespace nbp
