"""A host-side model of the visual map as the reference's (out-of-scope) map maintenance leaves it frame after frame — the INPUT side of the incremental device
mirror (livo2_visual_map_apply): points with obs_ lists, append-only observation arrays, one new reference image per frame.
  generateVisualMapPoints (reference src/vio.cpp:804-895)   new VisualPoints with ONE Feature made in the current frame, filed by insertPointIntoVoxelMap (227-246)
  updateVisualMapPoints   (vio.cpp:908-967)                 addFrameRef = obs_.push_front(new Feature) for sub-map points (visual_point.cpp:35-38); deleteFeatureRef of
                                                            one observation where a list is full (visual_point.cpp:40-55: ref_patch cleared if it was the deleted one)
  updateReferencePatch    (vio.cpp:969-1100)                normal_ flipped / replaced, ref_patch := some observation of the list, has_ref_patch_ = true
What changes is SCRIPTED here (seeded), not decided by the reference's criteria: the maintenance logic is out of scope, its effect on the containers is what the
mirror has to follow.  `step` returns one frame's delta in the vocabulary of livo2_visual_map_delta; `flat` returns the equivalent whole map (CSR, observations
compacted in point / list order) for a full upload or for the oracle, with the translation global observation index -> position in that table.
Pure numpy: neither the oracle nor the product is imported here."""
import copy

import numpy as np

from scenarios import synth


class GrowingMap:
    def __init__(self, cs):
        """cs: scenarios.synth.RetrieveChainScenario — the map when the first frame arrives (global observation index == its CSR position)"""
        self.cs = cs
        self.pos, self.keys, self.active = [np.array(cs.sel.pos, np.float64)], [np.array(cs.sel.keys, np.int64)], [np.array(cs.sel.active, np.uint8)]
        n = len(cs.sel.pos)
        self.normal, self.ninit, self.ref_patch = np.array(cs.normal, np.float64), np.array(cs.normal_initialized, np.uint8), np.array(cs.ref_patch, np.int32)
        self.lists = [list(range(int(cs.obs_offset[i]), int(cs.obs_offset[i + 1]))) for i in range(n)]
        self.obs = {k: [np.asarray(getattr(cs, "obs_" + k))] for k in ("id", "img_idx", "px", "f", "R", "t", "level", "inv_expo", "patch")}
        self.ref_imgs = [im for im in np.asarray(cs.ref_imgs)]
        self.n_obs = len(cs.obs_id)

    @property
    def n_points(self):
        return sum(len(p) for p in self.pos)

    def _cat(self):
        self.pos, self.keys, self.active = [np.concatenate(self.pos)], [np.concatenate(self.keys)], [np.concatenate(self.active)]
        for k in self.obs:
            self.obs[k] = [np.concatenate(self.obs[k])]

    def set_ref_patch(self, ref_patch):
        """what retrieveFromVisualSparseMap remembered (pt->ref_patch, vio.cpp:660-661, 689-690): global observation indices per point"""
        self.ref_patch = np.array(ref_patch, np.int32)

    def step(self, rng, R_fw, t_fw, img, frame_id, n_new=100, n_touch=100, max_list=30, inv_expo=1.0):
        """One frame of scripted maintenance at camera pose (R_fw, t_fw) with current image `img`.  Returns the delta (dict) and applies it to the model."""
        self._cat()
        cam = self.cs.sel.cam
        W, H = cam["width"], cam["height"]
        pos_all = self.pos[0]
        n0, m0 = len(pos_all), self.n_obs
        slot = len(self.ref_imgs)
        self.ref_imgs.append(np.array(img, np.uint8))

        def feature(p_w):
            pr = R_fw @ p_w + t_fw
            z = pr[2] if abs(pr[2]) > 1e-3 else 1e-3
            px = np.array([cam["fx"] * pr[0] / z + cam["cx"], cam["fy"] * pr[1] / z + cam["cy"]]) + rng.normal(0, 0.3, 2)
            px = np.clip(px, [6.0, 6.0], [W - 7.0, H - 7.0])
            f = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
            xi, yi = int(px[0]), int(px[1])
            patch = img[yi - 4:yi + 4, xi - 4:xi + 4].astype(np.float32).ravel() + rng.normal(0, 2.0, 64).astype(np.float32)
            return dict(id=frame_id, img_idx=slot, px=px, f=f / np.linalg.norm(f), R=R_fw.ravel(), t=t_fw, level=int(rng.integers(0, 3)), inv_expo=inv_expo, patch=patch)
        new_obs = []
        # generateVisualMapPoints: new points near existing ones (same surfaces), one Feature each
        src = rng.integers(0, n0, n_new)
        new_pos = pos_all[src] + rng.normal(0, 0.03, (n_new, 3))
        new_keys = synth.feat_map_key_np(new_pos)
        touched, t_lists, t_normal, t_ninit, t_ref, t_active = [], [], [], [], [], []
        cam_c = -R_fw.T @ t_fw
        normal = np.concatenate([self.normal, np.zeros((n_new, 3))]); ninit = np.concatenate([self.ninit, np.zeros(n_new, np.uint8)])
        refp = np.concatenate([self.ref_patch, np.full(n_new, -1, np.int32)])
        for k in range(n_new):
            g = m0 + len(new_obs); new_obs.append(feature(new_pos[k]))
            nv = cam_c - new_pos[k]; nv = nv / np.linalg.norm(nv) + rng.normal(0, 0.1, 3); nv /= np.linalg.norm(nv)
            self.lists.append([g])
            normal[n0 + k], ninit[n0 + k] = nv, 1                          # pt_new->is_normal_initialized_ = true (vio.cpp:886)
            touched.append(n0 + k)
        # updateVisualMapPoints / updateReferencePatch on a share of the resident points
        for p in rng.permutation(n0)[:n_touch]:
            p = int(p)
            lst = self.lists[p]
            if not self.active[0][p]:
                continue
            if len(lst) >= max_list or (len(lst) >= 3 and rng.uniform() < 0.15):       # deleteFeatureRef (the scripted victim: the last of the list)
                victim = lst.pop()
                if refp[p] == victim:
                    refp[p] = -1
            g = m0 + len(new_obs); new_obs.append(feature(pos_all[p]))
            lst.insert(0, g)                                                             # addFrameRef: push_front
            u = rng.uniform()
            if u < 0.25:
                refp[p] = lst[int(rng.integers(0, len(lst)))]                            # updateReferencePatch picked one
            elif u < 0.35:
                refp[p] = -1
            if rng.uniform() < 0.1:
                normal[p] = -normal[p]
            if rng.uniform() < 0.03:
                ninit[p] = 1 - ninit[p]
            touched.append(p)
        # a few points leave the map (touched with active = 0 and an empty list)
        gone = [int(p) for p in rng.permutation(n0)[:3] if int(p) not in set(touched)]
        active = np.concatenate([self.active[0], np.ones(n_new, np.uint8)])
        for p in gone:
            self.lists[p] = []; active[p] = 0; refp[p] = -1; touched.append(p)
        for p in touched:
            t_lists.append(list(self.lists[p])); t_normal.append(normal[p]); t_ninit.append(ninit[p]); t_ref.append(refp[p]); t_active.append(active[p])
        ob = {k: np.array([o[k] for o in new_obs]) for k in ("id", "img_idx", "px", "f", "R", "t", "level", "inv_expo", "patch")}
        # apply to the model
        self.pos.append(new_pos); self.keys.append(new_keys); self.active = [active]
        self.normal, self.ninit, self.ref_patch = normal, ninit, refp
        for k in self.obs:
            self.obs[k].append(ob[k])
        self.n_obs += len(new_obs)
        self._cat()
        return dict(new_pos=new_pos, new_keys=new_keys, new_active=np.ones(n_new, np.uint8), obs=ob,
                    touched=dict(point=np.array(touched, np.int32), lists=t_lists, normal=np.array(t_normal), normal_initialized=np.array(t_ninit, np.uint8),
                                 ref_patch=np.array(t_ref, np.int32), active=np.array(t_active, np.uint8)), img=self.ref_imgs[slot], img_slot=slot)

    def flat(self, template=None):
        """the whole map as a RetrieveChainScenario (CSR, observations compacted in point / list order) + g2c: global observation index -> CSR position (-1: unreferenced)"""
        self._cat()
        cs = copy.copy(template if template is not None else self.cs)
        cs.sel = copy.copy(cs.sel)
        n = self.n_points
        order = [g for lst in self.lists for g in lst]
        off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(l) for l in self.lists])
        g2c = np.full(self.n_obs, -1, np.int64); g2c[order] = np.arange(len(order))
        idx = np.array(order, np.int64)
        cs.sel.pos, cs.sel.keys, cs.sel.active = self.pos[0], self.keys[0], self.active[0]
        cs.normal, cs.normal_initialized = self.normal, self.ninit
        cs.ref_patch = np.where(self.ref_patch >= 0, g2c[np.maximum(self.ref_patch, 0)], -1).astype(np.int32)
        cs.obs_offset = off
        for k in self.obs:
            setattr(cs, "obs_" + k, self.obs[k][0][idx] if len(idx) else self.obs[k][0][:0])
        cs.obs_id, cs.obs_img_idx, cs.obs_level = cs.obs_id.astype(np.int32), cs.obs_img_idx.astype(np.int32), cs.obs_level.astype(np.int32)
        cs.ref_imgs = np.stack(self.ref_imgs)
        return cs, g2c
