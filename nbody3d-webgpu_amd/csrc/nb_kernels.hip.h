// nb_kernels.hip.h -- device code of the MI355X (gfx950) direct N-body engine.
//
// The reference runs ONE WGSL compute pass per frame (/root/reference nbody3d.js:219-292):
// tiled O(N^2) softened-gravity accumulation (:232-237 pair force, :255-272 tile loop)
// followed by the "velocity verlet with frame shift" update (:274-290), writing positions in
// place (:283) while other workgroups still stage them (:257) -- a cross-workgroup race.
// Here a step is well defined in both of its forms:
//
//   two kernels   K1 nb_force*   (reads positions only)  ->  K2 nb_integrate* (after all of K1)
//   one kernel    nb_step_fused  (reads bodies_in, writes bodies_out: ping-pong buffers)
//
//   K1 forms:   nb_force_symw<NG,J>      f32, SYMMETRIC pass (default from N ~ 13,000): every unordered pair once, both
//               nb_force_symw64<8>       accelerations; residents in registers, travelers rotate through the wave by DPP
//               nb_force_sym<WS,NG,J>    (f64 form; workgroup form with LDS-combined traveler sums: A/B arm)
//               nb_force_pk_sgpr<NG,WS>  f32, packed math, ordered pairs, j broadcast from SGPRs (shards without the native exchange)
//               nb_force_pk<NG,LS,TL>    f32, packed math, j-tile staged in LDS
//               nb_force<T,IPL,LS>       scalar template: f64, and the unpacked f32 shapes
//   fused form: nb_step_fused<NG,LS,TL>  nb_force_pk's loop over ALL j + the integrator in the
//                                        epilogue (no j-split across workgroups: LS lanes of a
//                                        wave share an i-body instead) -- one launch per step,
//                                        no partial sums through memory
//               nb_step_direct<MAXJ>     the same for N <= 2,048 with each lane's j-bodies loaded
//                                        straight into registers (no LDS tile, no barrier)
//               nb_step_jpk<WS>          A/B arm: packed across TWO j-BODIES streamed as SGPR pairs from a
//                                        pair-transposed position copy; j split over the waves of a workgroup
//                                        and over workgroups that meet at a ticket inside the launch
//
// CDNA4 mapping of the force loop (wave = 64 lanes, 4 SIMDs/CU, 160 KiB LDS/CU):
//   * a 256-thread workgroup (4 waves, one per SIMD) stages a j-tile of 256*TL bodies
//     (x, y, z, m) in LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write),
//     double buffered, ONE s_barrier per tile; the next tile lands while the current one computes
//     (the scalar template nb_force<T,...> stages (x, y, z, G*m) through registers);
//   * the inner loop reads the tile with ds_read_b128 at a wave-uniform address
//     (LDS broadcast: one read feeds 64*IPL pair evaluations) -- LS == 1 -- or
//     at LS consecutive addresses when LS lanes share one i-body;
//   * each lane keeps IPL i-bodies in VGPRs (register blocking), loaded with coalesced 16-B accesses;
//   * f32: the arithmetic is packed across TWO i-bodies of the lane (v_pk_add/fma/mul_f32):
//     per two pairs 3 v_pk_add, 3 v_pk_fma (r^2 + eps2), 2 v_pk_mul (cube), 2 v_rsq_f32,
//     1 v_pk_mul (m_j), 3 v_pk_fma (accumulate) = 12 packed (4 cycles each) + 2 transcendental
//     (8 cycles each) = 64 issue cycles per 128 pairs, issued stage-major over 4 independent
//     chains so no hazard s_nop is needed;
//   * no branch in the loop: with eps2 > 0 the self term is exactly 0*finite = 0 and bodies
//     past the range are staged as zero-mass;
//   * LS > 1: the LS partial sums of a body are reduced inside the wave (DPP row operations
//     and row broadcasts for f32, wavefront shuffles for f64) before ONE lane stores/integrates;
//   * grid = (i-blocks, jsplit): j may also be partitioned over blockIdx.y (any multiple of 8
//     bodies per split); K2 sums the jsplit partials in ascending order (deterministic, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nb_plan.h"

// The device code lives in kernels/*.hip.h, in dependency order:
#include "kernels/common.hip.h"      // row loads, pair(), in-wave sums, leapfrog()
#include "kernels/ordered.hip.h"     // nb_force, nb_force_pk, nb_step_fused, nb_step_direct, nb_force_pk_sgpr
#include "kernels/symmetric.hip.h"   // nb_force_sym*, nb_sym_reduce, nb_peer_*, nb_integrate_sym*
#include "kernels/jpk.hip.h"         // nb_step_jpk, nb_pairs_pack, nb_gm_pack
#include "kernels/integrate.hip.h"   // nb_integrate, nb_integrate_swap, nb_frame_pack, nb_diag
