/*
 * nbody3d_hip_plan.h -- planner introspection of libnbody3d_hip.so (ABI 2.3; part of nbody3d_hip.h up to 2.2).
 *
 * NOT part of the drop-in surface: the reference dispatches ceil(N / 256) workgroups of one kernel (nbody3d.js:478) and has
 * nothing to introspect.  The engine picks a force-pass form per handle with a cost model and lays out its partitions
 * (csrc/nb_plan.cpp); this header exposes that plan -- including the kernel-internal plan words and tables of the symmetric
 * pass -- to reports, sizing runs and tests/test_planner_cpu.py, which walks the plan the way the kernels do.  The words
 * follow the kernels' structs (csrc/nb_plan.h) and may change with any minor version.
 */
#ifndef NBODY3D_HIP_PLAN_H
#define NBODY3D_HIP_PLAN_H

#include "nbody3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* The launch plan nb_create WOULD build for `cfg` on a device with n_cu compute units at clock_hz -- the engine's planner
 * (kernel form, j-partitions, the symmetric pass's super-block ring, its wave ranges and layer table) run on the host
 * alone.  No device is needed or touched: with n_cu > 0 and clock_hz > 0 the call works on a machine without a GPU (the
 * planner's occupancy queries then use their built-in defaults); n_cu <= 0 or clock_hz <= 0 means "as on cfg->device".
 * For reports, for sizing runs ahead of time, and for the host-side tests of the planner (tests/test_planner_cpu.py walks
 * the plan the way the kernels do and checks that every pair is covered exactly once).  No reference analogue: the
 * reference's dispatch is the one line ceil(N / 256) of nbody3d.js:478.
 *   kind/ipl/ls/x      the force_variant digits K, II, LL, X of the chosen form
 *   jsplit..own_splits as nb_shape_info
 *   sym*               symmetric pass only: padded rows, partial-sum layers, the rank form's super-block range, and the
 *                      plan words the kernels receive (nb::SymWPlan: np, nsb, W, total_hi, total_lo, n_hi, H, r_layer0,
 *                      t_layer0, L, zc -- nsb: the whole super-blocks of the ring, zc: the real chunks of the short block a ragged
 *                      N leaves behind them; workgroup form nb::SymPlan: np, nsb, q, total_hi, total_lo, n_hi, H, r_layer0, t_layer0)
 *   tab                (caller's array of tab_cap words, may be NULL) first wave and resident layer count of every super-block's
 *                      list (a layer per workgroup of four waves ending there, one more if the last wave goes on), 2 words per
 *                      block of rows (sym_np / rows per super-block: the short block last); then FOUR words per wave (4 * workgroup +
 *                      wave in it) -- {first unit, end, resident layer of the super-block the range ends in, spill row}: the ranges
 *                      partition the list of units (with two waves per SIMD consecutive ranges alternate between wave i and wave
 *                      W / 2 + i, and the first of the two is the longer one; with whole sweeps its record ends a few sweeps short of
 *                      the next range: those are the queued pieces, {first unit, resident layer | sweeps << 16} each, behind the
 *                      records, which every wave draws from when its own part is done); with sym_ups > 1 followed by the spill lists -- {first
 *                      spill row, count} per traveler chunk (2 * sym_np / 64 words), then the wave numbers in spill-row order;
 *                      tab_len reports how many words there are
 *   sym_ups            work units per chunk-sweep: the wave ranges are floor/ceil-equal in units of 64 / sym_ups rotation steps */
typedef struct nb_plan_info {
  uint32_t struct_size; /* sizeof(nb_plan_info), set by the caller */
  uint32_t kind, ipl, ls, x;
  uint32_t jsplit, j_per_split, own_split0, own_splits;
  uint32_t sym, symw, sym_rank, sym_np, sym_layers, sym_g0, sym_g1;
  uint32_t sym_plan[11];
  uint32_t tab_len;
  char variant[112];
  uint32_t sym_ups;        /* wave-granular symmetric pass: work units per chunk-sweep (1: whole sweeps; 4: quarter sweeps) */
  uint32_t sym_spill_rows; /* rows of the spill buffer: the z-rows (a super-block's sums for a chunk of the short block), then one row set per wave that starts inside a sweep */
  uint32_t sym_rank_plan[16]; /* rank form (sym_rank): np, nsb, total_hi, total_lo, n_hi, H, r_layer0, rb_layer0, t_layer0, g0, g1, LA, LB,
                                 WA, WB, ups -- phase A = the sweeps whose travelers are the rank's own rows (LA sweeps, waves [0, WA):
                                 what an overlapped step issues before it waits for the all-gather), phase B the rest.  `tab` then
                                 holds {first A wave, A waves, first B wave, B waves} per super-block (4 * nsb words), the two
                                 phases' prefix tables (g1 - g0 + 1 words each) and, with ups > 1, the spill lists ({offset, count}
                                 per 64-row chunk, then the wave numbers) */
  uint32_t sym_pass;       /* IN/OUT: which pass of the rank-form pipeline sym_rank_plan and `tab` describe (set before the call; 0 when
                              in doubt).  A whole system whose traveler layers would not fit the layer budget runs its ring distances
                              in sym_passes passes that reuse the layers (variant suffix "_pN") */
  uint32_t sym_passes;     /* passes of the rank-form pipeline (1 for an ordinary rank; 0 when the handle is not in the rank form) */
  uint32_t sym_pass_k_lo, sym_pass_k_hi, sym_pass_d0; /* the pass's window of every super-block's ring sweeps [k_lo, k_hi) and its first ring distance */
  uint32_t sym_local;      /* the rank-form pipeline of a WHOLE system on one device: no communicator, nothing exchanged */
} nb_plan_info;
int nb_plan_query(const nb_config *cfg, int n_cu, double clock_hz, nb_plan_info *out, uint32_t *tab,
                  uint32_t tab_cap);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NBODY3D_HIP_PLAN_H */
