// gfx950 kernels for the INVERSE-compositional visual update (vio/inverse_composition_en).
//   k_visual_ref_precompute   : reference src/vio.cpp:1327-1396 (precomputeReferencePatches, once per pyramid level)
//   k_visual_inverse_residual : reference src/vio.cpp:1418-1477 (residuals + Jacobian rows of updateStateInverse)
//   accept / revert / solve   : k_visual_solve (visual_kernels.hpp) — the 6x6 block of the reference is the 7x7 block with a zero
//                               exposure row/column, which yields the identical K_1[:,0:6], G and solution.
// Factorisation: the reference stores H_sub_inv rows [J_dR, J_dt] = g_ref * M_ref with the per-pixel REFERENCE-image gradient
// g_ref = [du,dv]/scale and a patch-constant 2x6 M_ref = [Jpi R_ref_w [p]x | -Jpi R_ref_w]; per iteration every row is multiplied by the
// state-dependent 6x6 T = [[Rwi, 0],[ [Pwi]x Rwi, Rwi ]] (vio.cpp:1470-1474).  So the precompute keeps g_ref (2 doubles / pixel), M_ref
// and sum(g g^T) per patch, and an iteration only needs the residuals and sum(g z): H^T H = N^T (sum g g^T) N, H^T z = N^T sum(g z),
// N = M_ref T — no 64M x 6 matrix is ever read back.
#pragma once
#include "visual_kernels.hpp"

// one patch per wave (grid = visual_grid_inverse(M))
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_ref_precompute(VisualKernelArgs a, VisualRefArgs r) {
  __shared__ float Wf[VIS_WAVES][11 * 11 + 3];
  __shared__ float Bf[VIS_WAVES][10 * 10 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch = blockIdx.x * VIS_WAVES + wave;
  if (patch >= a.M) return;                                   // wave-uniform; no block barrier below
  ref_precompute_patch(a, r, a.level, patch, lane, Wf[wave], Bf[wave]);
}

// one evaluation of updateStateInverse: the shared wave body (visual_kernels.hpp, INV), VIS_PPB patches per block like k_visual_residual (grid = visual_grid(M))
template <bool DEBUG_ROWS>
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_inverse_residual(VisualKernelArgs a, VisualRefArgs r, const DevCtl *__restrict__ ctl,
                                                                       double *__restrict__ partials, int check_stop) {
  if (check_stop && ctl->hdr.stop) return;
  __shared__ VisWaveLds lds[VIS_WAVES];
  __shared__ double red[VIS_WAVES][VIS_PSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch0 = (blockIdx.x * VIS_WAVES + wave) * VIS_PPW;   // wave-uniform
  double out_val = 0.0;
  if (patch0 < a.M) out_val = visual_wave_body<DEBUG_ROWS, false, 0, true>(a, a.level, ctl->cur.rot, ctl->cur.pos, &ctl->cur.inv_expo, a.errors, lds[wave], patch0, lane, &r);
  vis_block_store(red, out_val, partials + (size_t)blockIdx.x * VIS_PSTRIDE);
}
