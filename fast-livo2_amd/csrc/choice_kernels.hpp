// Visual sub-map retrieval, the step between selection and tail (reference src/vio.cpp:644-696 and
// src/visual_point.cpp:57-95): for every grid cell whose selected visual point passed the depth-continuity test, pick the observation
// (Feature) whose patch is warped into the current frame, then line the chosen (point, observation) pairs up, in grid-cell order, as the
// candidate arrays k_warp_candidates reads — selection, choice and tail run back to back on the device, no host round trip.
//   k_choose_ref        one wave per grid cell, one lane per observation of the cell's point:
//                         normal_en:  the observation whose stored patch has the smallest mean squared difference to the patches of the
//                                     point's OTHER observations (other = different id_, i.e. made in a different frame); computed once
//                                     per point, remembered in ref_patch (pt->ref_patch / has_ref_patch_);
//                         !normal_en: VisualPoint::getCloseViewObs — the observation with the closest viewing direction, 60 degree gate.
//                       Every lane runs its observation's sums serially in the reference's order (float accumulators), the wave then takes
//                       the minimum / maximum with the serial loop's tie rule (the FIRST observation of the best score).
//   k_gather_candidates chosen (cell -> observation) -> dense candidate arrays in cell order (slots from k_warp_scan)
//   k_leader_insert / k_leader_lookup   warp_map of the !normal_en branch (vio.cpp:716-734): the cache key ref_ftr->id_ is the id of the
//                       FRAME a feature was made in, so all candidates whose ref_ftr share a frame reuse the warp of the first of them.
// The visual map's observations live on the device as global arrays (append-only) + one fixed-stride index list per point in obs_ list order: a full upload
// (livo2_visual_obs_upload, a CSR table) and the per-frame deltas of the map maintenance (livo2_visual_map_apply) maintain the same structure.
#pragma once
#include "livo2_device.hpp"

#define CHOICE_WAVES 4
#define LEADER_EMPTY 0x7FFFFFFF

struct ChoiceArgs {
  int32_t normal_en, length;
  double cam_pos[3];                         // new_frame_->pos()
  const int32_t *cell_point;                 // [length] from k_sel_cells
  const uint8_t *cell_discont;               // [length]
  const double *pos;                         // [n_pts][3]
  const int32_t *obs_list;                   // [n_pts][obs_stride]: VisualPoint::obs_ in list order, as GLOBAL observation indices (livo2_visual_obs_upload fills it from
                                             // the CSR table; livo2_visual_map_apply rewrites the lists of the points a frame touched: addFrameRef pushes to the FRONT)
  const int32_t *obs_count;                  // [n_pts] obs_.size()
  int32_t obs_stride, pad_;
  const int32_t *obs_id;
  const double *obs_R, *obs_t;
  const float *obs_patch;                    // [n_obs][64]
  const uint8_t *normal_init;                // [n_pts]
  int32_t *ref_patch;                        // [n_pts] global observation index or -1; updated
  int32_t *cell_obs;                         // [length] out: global observation index of ref_ftr or -1
  int32_t *cell_flag;                        // [length] out: 1 = the cell yields a candidate
};

__global__ void __launch_bounds__(CHOICE_WAVES *LIVO2_WAVE) k_choose_ref(ChoiceArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * CHOICE_WAVES + (threadIdx.x >> 6));
  if (c >= a.length) return;
  int chosen = -1;                                                   // wave-uniform
  const int p = a.cell_point[c];
  if (p >= 0 && !a.cell_discont[c] && a.normal_init[p]) {
    const int32_t *ol = a.obs_list + (size_t)p * a.obs_stride;          // the point's observations, list order
    const int n = a.obs_count[p];
    if (a.normal_en) {
      const int preset = a.ref_patch[p];
      if (n == 1) {
        chosen = ol[0];
        if (lane == 0) a.ref_patch[p] = chosen;
      } else if (n > 1 && preset < 0) {
        float best = 3.402823466e+38f;                               // FLT_MAX (phtometric_errors_min): a score must be below it to win
        int best_a = -1;
        for (int it = lane; it < n; it += 64) {
          const int oa = ol[it];
          const float *pa = a.obs_patch + (size_t)oa * 64;
          const int id_a = a.obs_id[oa];
          float err = 0.0f;
          int count = 0;
          for (int itm = 0; itm < n; itm++) {
            const int ob = ol[itm];
            if (a.obs_id[ob] == id_a) continue;
            const float *pb = a.obs_patch + (size_t)ob * 64;
            for (int ind = 0; ind < 64; ind++) { const float d = pa[ind] - pb[ind]; err = err + d * d; }
            count++;
          }
          err = err / (float)count;                                  // 0/0 = NaN when every observation carries one id: never wins
          if (err < best) { best = err; best_a = it; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {                    // minimum score; equal scores: the earlier observation
          const float e2 = __shfl_xor(best, off, 64);
          const int a2 = __shfl_xor(best_a, off, 64);
          if (e2 < best || (e2 == best && (unsigned)a2 < (unsigned)best_a)) { best = e2; best_a = a2; }
        }
        if (best_a >= 0) {
          chosen = ol[best_a];
          if (lane == 0) a.ref_patch[p] = chosen;
        }
      } else if (n > 1) chosen = preset;
    } else if (n > 0) {
      const double px = a.pos[(size_t)p * 3], py = a.pos[(size_t)p * 3 + 1], pz = a.pos[(size_t)p * 3 + 2];
      double od[3] = {a.cam_pos[0] - px, a.cam_pos[1] - py, a.cam_pos[2] - pz};
      const double on = sqrt((od[0] * od[0] + od[1] * od[1]) + od[2] * od[2]);
      od[0] = od[0] / on; od[1] = od[1] / on; od[2] = od[2] / on;
      double best = 0.0;                                             // min_cos_angle starts at 0: only a positive cosine replaces the first observation
      int best_a = -1;
      for (int it = lane; it < n; it += 64) {
        const double *R = a.obs_R + (size_t)ol[it] * 9, *t = a.obs_t + (size_t)ol[it] * 3;
        double d[3];
#pragma unroll
        for (int r = 0; r < 3; r++) d[r] = ((R[r] * t[0] + R[3 + r] * t[1]) + R[6 + r] * t[2]) * (-1.0);     // Feature::pos()
        d[0] = d[0] - px; d[1] = d[1] - py; d[2] = d[2] - pz;
        const double dn = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        d[0] = d[0] / dn; d[1] = d[1] / dn; d[2] = d[2] / dn;
        const double cs = (od[0] * d[0] + od[1] * d[1]) + od[2] * d[2];
        if (cs > best) { best = cs; best_a = it; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double e2 = __shfl_xor(best, off, 64);
        const int a2 = __shfl_xor(best_a, off, 64);
        if (e2 > best || (e2 == best && (unsigned)a2 < (unsigned)best_a)) { best = e2; best_a = a2; }
      }
      if (best_a >= 0 && !(best < 0.5)) chosen = ol[best_a];          // min_cos_angle < 0.5: more than 60 degrees off
    }
  }
  if (lane == 0) { a.cell_obs[c] = chosen; a.cell_flag[c] = chosen >= 0 ? 1 : 0; }
}

struct GatherCandArgs {
  int32_t length, pad;
  const int32_t *slot;                       // [length] from k_warp_scan over cell_flag
  const int32_t *cell_point, *cell_obs;
  const double *pos, *normal;                // per visual point
  const int32_t *obs_id, *obs_img_idx, *obs_level;
  const double *obs_px, *obs_f, *obs_R, *obs_t, *obs_inv_expo;
  double *c_pos, *c_normal, *c_px, *c_f, *c_R, *c_t, *c_ie;
  int32_t *c_idx, *c_lvl, *c_id, *cand_cell, *cand_point, *cand_obs;
};

// 32 lanes per cell: lane k copies word k of the candidate record
__global__ void __launch_bounds__(256) k_gather_candidates(GatherCandArgs a) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
  if (c >= a.length) return;
  const int s = a.slot[c];
  if (s < 0) return;
  const int p = a.cell_point[c], o = a.cell_obs[c];
  if (k < 3) a.c_pos[(size_t)s * 3 + k] = a.pos[(size_t)p * 3 + k];
  else if (k < 6) a.c_normal[(size_t)s * 3 + (k - 3)] = a.normal[(size_t)p * 3 + (k - 3)];
  else if (k < 8) a.c_px[(size_t)s * 2 + (k - 6)] = a.obs_px[(size_t)o * 2 + (k - 6)];
  else if (k < 11) a.c_f[(size_t)s * 3 + (k - 8)] = a.obs_f[(size_t)o * 3 + (k - 8)];
  else if (k < 20) a.c_R[(size_t)s * 9 + (k - 11)] = a.obs_R[(size_t)o * 9 + (k - 11)];
  else if (k < 23) a.c_t[(size_t)s * 3 + (k - 20)] = a.obs_t[(size_t)o * 3 + (k - 20)];
  else if (k == 23) a.c_ie[s] = a.obs_inv_expo[o];
  else if (k == 24) a.c_idx[s] = a.obs_img_idx[o];
  else if (k == 25) a.c_lvl[s] = a.obs_level[o];
  else if (k == 26) a.c_id[s] = a.obs_id[o];
  else if (k == 27) a.cand_cell[s] = c;
  else if (k == 28) a.cand_point[s] = p;
  else if (k == 29) a.cand_obs[s] = o;
}

__device__ __forceinline__ uint32_t leader_hash(int32_t id) { uint32_t h = (uint32_t)id * 0x9E3779B1u; return h ^ (h >> 15); }

// warp_map: table of {id -> lowest candidate index carrying it}.  keys / vals: capacity mask + 1, initialised to LEADER_EMPTY.
// (a call has a handful of frame ids for thousands of candidates: the lanes of a wave that carry the same id send ONE pair of atomics, from the lowest
// lane = the lowest candidate index of the group; with one atomic pair per candidate the kernel took 27 us on six hot addresses)
__global__ void __launch_bounds__(256) k_leader_insert(const int32_t *__restrict__ id, const int32_t *__restrict__ n_dev, int n_host, int32_t *keys, int32_t *vals,
                                                       uint32_t mask) {
  const int n = n_dev ? n_dev[0] : n_host;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const bool act = i < n;
  const int32_t key = act ? id[i] : 0;
  unsigned long long todo = __ballot(act);
  while (todo) {                                                      // wave-uniform: one turn per distinct id in the wave
    const int first = __ffsll((long long)todo) - 1;
    const int32_t k0 = __shfl(key, first, 64);
    const unsigned long long same = __ballot(act && key == k0) & todo;
    if (lane == first) {
      uint32_t h = leader_hash(k0) & mask;
      for (;;) {
        const int32_t old = atomicCAS(&keys[h], LEADER_EMPTY, k0);
        if (old == LEADER_EMPTY || old == k0) break;
        h = (h + 1) & mask;
      }
      atomicMin(&vals[h], i);
    }
    todo &= ~same;
  }
}
__global__ void __launch_bounds__(256) k_leader_lookup(const int32_t *__restrict__ id, const int32_t *__restrict__ n_dev, int n_host, const int32_t *__restrict__ keys,
                                                       const int32_t *__restrict__ vals, uint32_t mask, int32_t *__restrict__ leader) {
  const int n = n_dev ? n_dev[0] : n_host;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t key = id[i];
  uint32_t h = leader_hash(key) & mask;
  while (keys[h] != key) h = (h + 1) & mask;
  leader[i] = vals[h];
}

// ---- the per-point observation lists ---------------------------------------------------------------------------------------------------------------
// CSR table of a full upload -> fixed-stride lists (identity ranges)
__global__ void __launch_bounds__(256) k_ob_lists_from_csr(const int32_t *__restrict__ off, int n, int stride, int32_t *__restrict__ list, int32_t *__restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, p = t / stride, k = t % stride;
  if (p >= n) return;
  const int b = off[p], m = off[p + 1] - b;
  if (k == 0) count[p] = m;
  list[(size_t)p * stride + k] = k < m ? b + k : -1;
}

// One frame's changes of the visual map (livo2_visual_map_apply), from ONE staging blob that travelled in one copy:
//   new points  [n_new_points] : pos, packed voxel key, active                                      -> rows n_points .. of the point arrays
//   new observations [n_new_obs]: id, image slot, level, px, f, R, t, inv_expo, patch               -> rows n_obs .. of the observation arrays
//   touched points [n_touched] : whole obs_ list (global indices, list order), normal_, is_normal_initialized_, active, ref_patch
struct VmDeltaArgs {
  int32_t n_points, n_obs, n_new_points, n_new_obs, n_touched, stride, pad0, pad1;
  // staged sections (device addresses inside the blob)
  const double *s_pos; const unsigned long long *s_pkey; const uint8_t *s_active;
  const int32_t *s_oid, *s_oimg, *s_olvl; const double *s_opx, *s_of, *s_oR, *s_ot, *s_oie; const float *s_opatch;
  const int32_t *s_tpoint, *s_toff, *s_tobs, *s_trefp; const double *s_tnormal; const uint8_t *s_tninit, *s_tactive;
  // the resident arrays
  double *pos; unsigned long long *pkey; uint8_t *active, *fov;
  int32_t *oid, *oimg, *olvl; double *opx, *of, *oR, *ot, *oie; float *opatch;
  int32_t *list, *count, *refp; double *normal; uint8_t *ninit;
};
__global__ void __launch_bounds__(256) k_vm_apply(VmDeltaArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // section 1: new points, one thread each
  if (t < a.n_new_points) {
    const size_t p = (size_t)a.n_points + t;
    a.pos[p * 3] = a.s_pos[(size_t)t * 3]; a.pos[p * 3 + 1] = a.s_pos[(size_t)t * 3 + 1]; a.pos[p * 3 + 2] = a.s_pos[(size_t)t * 3 + 2];
    a.pkey[p] = a.s_pkey[t]; a.active[p] = a.s_active[t]; a.fov[p] = 0;
    a.count[p] = 0; a.refp[p] = -1; a.ninit[p] = 0; a.normal[p * 3] = 0.0; a.normal[p * 3 + 1] = 0.0; a.normal[p * 3 + 2] = 0.0;      // until its `touched` row lands (k_vm_apply_touched)
    for (int k = 0; k < a.stride; k++) a.list[p * a.stride + k] = -1;              // as k_ob_lists_from_csr leaves the unused entries of a full upload (a new point without a touched row: advisor, round 5)
  }
  // section 2: new observations, 32 threads each (the patch is 64 floats = 32 x 8 bytes)
  {
    const int o = t >> 5, k = t & 31;
    if (o < a.n_new_obs) {
      const size_t g = (size_t)a.n_obs + o;
      reinterpret_cast<double *>(a.opatch + g * 64)[k] = reinterpret_cast<const double *>(a.s_opatch + (size_t)o * 64)[k];
      if (k < 9) a.oR[g * 9 + k] = a.s_oR[(size_t)o * 9 + k];
      else if (k < 12) a.ot[g * 3 + (k - 9)] = a.s_ot[(size_t)o * 3 + (k - 9)];
      else if (k < 15) a.of[g * 3 + (k - 12)] = a.s_of[(size_t)o * 3 + (k - 12)];
      else if (k < 17) a.opx[g * 2 + (k - 15)] = a.s_opx[(size_t)o * 2 + (k - 15)];
      else if (k == 17) a.oie[g] = a.s_oie[o];
      else if (k == 18) a.oid[g] = a.s_oid[o];
      else if (k == 19) a.oimg[g] = a.s_oimg[o];
      else if (k == 20) a.olvl[g] = a.s_olvl[o];
    }
  }
}
// touched points in their own launch (a new point's defaults of k_vm_apply must be in place first): `stride` threads per touched point
__global__ void __launch_bounds__(256) k_vm_apply_touched(VmDeltaArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, q = t / a.stride, k = t % a.stride;
  if (q >= a.n_touched) return;
  const size_t p = (size_t)a.s_tpoint[q];
  const int b = a.s_toff[q], m = a.s_toff[q + 1] - b;
  a.list[p * a.stride + k] = k < m ? a.s_tobs[b + k] : -1;
  if (k == 0) { a.count[p] = m; a.refp[p] = a.s_trefp[q]; a.ninit[p] = a.s_tninit[q]; a.active[p] = a.s_tactive[q]; }
  if (k < 3) a.normal[p * 3 + k] = a.s_tnormal[(size_t)q * 3 + k];
}
