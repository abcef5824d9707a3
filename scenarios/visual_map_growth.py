"""A host-side model of the visual map as the reference's (out-of-scope) map maintenance leaves it frame after frame — the INPUT side of the incremental device
mirror (livo2_visual_map_apply): points with obs_ lists, append-only observation arrays, one new reference image per frame.
  generateVisualMapPoints (reference src/vio.cpp:804-895)   new VisualPoints with ONE Feature made in the current frame, filed by insertPointIntoVoxelMap (227-246)
  updateVisualMapPoints   (vio.cpp:908-967)                 addFrameRef = obs_.push_front(new Feature) for sub-map points (visual_point.cpp:35-38); deleteFeatureRef of
                                                            one observation where a list is full (visual_point.cpp:40-55: ref_patch cleared if it was the deleted one)
  updateReferencePatch    (vio.cpp:969-1100)                normal_ flipped / replaced, ref_patch := some observation of the list, has_ref_patch_ = true
What changes is SCRIPTED here (seeded), not decided by the reference's criteria: the maintenance logic is out of scope, its effect on the containers is what the
mirror has to follow.  A script (`generate`) is a list of container operations that does not depend on what the retrieval remembered in pt->ref_patch, so it can be
written to disk ahead of a run (scenarios/live_inputs.py, host/live_chain.cpp replays it on the shim's objects); `apply` replays it on this model and returns that
frame's delta in the vocabulary of livo2_visual_map_delta; `flat` returns the equivalent whole map (CSR, observations compacted in point / list order) for a full
upload or for the oracle, with the translation global observation index -> position in that table.
Pure numpy: neither the oracle nor the product is imported here."""
import copy

import numpy as np

from scenarios import synth

REF_KEEP = -2                  # ref action of a touched point: leave pt->ref_patch as it is (unless the observation it points at was just deleted)
OBS_KEYS = ("id", "img_idx", "px", "f", "R", "t", "level", "inv_expo", "patch")


class GrowingMap:
    def __init__(self, cs):
        """cs: scenarios.synth.RetrieveChainScenario — the map when the first frame arrives (global observation index == its CSR position)"""
        self.cs = cs
        self.pos, self.keys, self.active = np.array(cs.sel.pos, np.float64), np.array(cs.sel.keys, np.int64), np.array(cs.sel.active, np.uint8)
        n = len(cs.sel.pos)
        self.normal, self.ninit, self.ref_patch = np.array(cs.normal, np.float64), np.array(cs.normal_initialized, np.uint8), np.array(cs.ref_patch, np.int32)
        self.lists = [list(range(int(cs.obs_offset[i]), int(cs.obs_offset[i + 1]))) for i in range(n)]
        self.obs = {k: np.asarray(getattr(cs, "obs_" + k)) for k in OBS_KEYS}
        self.ref_imgs = [im for im in np.asarray(cs.ref_imgs)]

    @property
    def n_points(self):
        return len(self.pos)

    @property
    def n_obs(self):
        return len(self.obs["id"])

    def set_ref_patch(self, ref_patch):
        """what retrieveFromVisualSparseMap remembered (pt->ref_patch, vio.cpp:660-661, 689-690): global observation indices per point"""
        self.ref_patch = np.array(ref_patch, np.int32)

    # ---- one frame of scripted maintenance ----------------------------------------------------------------------------------------------------------
    def generate(self, rng, R_fw, t_fw, img, frame_id, n_new=100, n_touch=100, max_list=30, inv_expo=1.0):
        """The operations of one frame at camera pose (R_fw, t_fw) with current image `img` (read-only on the model).  Global indices of the new observations:
        n_obs + k in the order of `obs`; indices of the new points: n_points + k."""
        cam = self.cs.sel.cam
        W, H = cam["width"], cam["height"]
        n0, m0 = self.n_points, self.n_obs
        slot = len(self.ref_imgs)
        img = np.array(img, np.uint8)

        def feature(p_w):
            pr = R_fw @ p_w + t_fw
            z = pr[2] if abs(pr[2]) > 1e-3 else 1e-3
            px = np.array([cam["fx"] * pr[0] / z + cam["cx"], cam["fy"] * pr[1] / z + cam["cy"]]) + rng.normal(0, 0.3, 2)
            px = np.clip(px, [6.0, 6.0], [W - 7.0, H - 7.0])
            f = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
            xi, yi = int(px[0]), int(px[1])
            patch = img[yi - 4:yi + 4, xi - 4:xi + 4].astype(np.float32).ravel() + rng.normal(0, 2.0, 64).astype(np.float32)
            return dict(id=frame_id, img_idx=slot, px=px, f=f / np.linalg.norm(f), R=np.asarray(R_fw, np.float64).ravel(), t=np.asarray(t_fw, np.float64),
                        level=int(rng.integers(0, 3)), inv_expo=inv_expo, patch=patch)
        new_obs = []
        src = rng.integers(0, n0, n_new)
        new_pos = self.pos[src] + rng.normal(0, 0.03, (n_new, 3))
        cam_c = -np.asarray(R_fw).T @ np.asarray(t_fw)
        new_normal = np.zeros((n_new, 3))
        for k in range(n_new):
            new_obs.append(feature(new_pos[k]))                                              # global index m0 + k: the new point's only observation
            nv = cam_c - new_pos[k]; nv = nv / np.linalg.norm(nv) + rng.normal(0, 0.1, 3)
            new_normal[k] = nv / np.linalg.norm(nv)
        t_point, t_pop, t_push, t_ref, t_flip, t_toggle, t_remove = [], [], [], [], [], [], []
        picked = set()
        for p in rng.permutation(n0)[:n_touch]:
            p = int(p)
            if not self.active[p]:
                continue
            picked.add(p)
            ln = len(self.lists[p])
            pop = 1 if (ln >= max_list or (ln >= 3 and rng.uniform() < 0.15)) else 0          # deleteFeatureRef (the scripted victim: the last of the list)
            g = m0 + len(new_obs); new_obs.append(feature(self.pos[p]))
            u = rng.uniform()
            after = [g] + self.lists[p][: ln - pop]
            ref = after[int(rng.integers(0, len(after)))] if u < 0.25 else (-1 if u < 0.35 else REF_KEEP)      # updateReferencePatch picked one / lost it / did not run
            t_point.append(p); t_pop.append(pop); t_push.append(g); t_ref.append(ref); t_flip.append(int(rng.uniform() < 0.1)); t_toggle.append(int(rng.uniform() < 0.03)); t_remove.append(0)
        for p in rng.permutation(n0)[:3]:                                                   # a few points leave the map
            if int(p) not in picked and self.active[int(p)]:
                t_point.append(int(p)); t_pop.append(0); t_push.append(-1); t_ref.append(-1); t_flip.append(0); t_toggle.append(0); t_remove.append(1)
        ob = {k: np.array([o[k] for o in new_obs]) for k in OBS_KEYS}
        i32 = lambda a: np.array(a, np.int32)
        return dict(new_pos=new_pos, new_keys=synth.feat_map_key_np(new_pos), new_normal=new_normal, obs=ob, img=img, img_slot=slot,
                    t_point=i32(t_point), t_pop=i32(t_pop), t_push=i32(t_push), t_ref=i32(t_ref), t_flip=i32(t_flip), t_toggle=i32(t_toggle), t_remove=i32(t_remove))

    def apply(self, s):
        """Replays a script on the model (ref_patch as it stands: set_ref_patch first) and returns the delta for livo2_visual_map_apply: new points, new
        observations, and the WHOLE new state of every touched point (the new points included)."""
        n0, m0 = self.n_points, self.n_obs
        n_new = len(s["new_pos"])
        assert s["img_slot"] == len(self.ref_imgs)
        self.ref_imgs.append(np.array(s["img"], np.uint8))
        self.pos = np.concatenate([self.pos, s["new_pos"]]); self.keys = np.concatenate([self.keys, s["new_keys"]]); self.active = np.concatenate([self.active, np.ones(n_new, np.uint8)])
        self.normal = np.concatenate([self.normal, s["new_normal"]]); self.ninit = np.concatenate([self.ninit, np.ones(n_new, np.uint8)])      # is_normal_initialized_ = true (vio.cpp:886)
        self.ref_patch = np.concatenate([self.ref_patch, np.full(n_new, -1, np.int32)])
        for k in OBS_KEYS:
            self.obs[k] = np.concatenate([self.obs[k], np.asarray(s["obs"][k]).astype(self.obs[k].dtype)])
        touched = []
        for k in range(n_new):
            self.lists.append([m0 + k]); touched.append(n0 + k)
        for q, p in enumerate(s["t_point"]):
            p = int(p)
            lst = self.lists[p]
            if s["t_remove"][q]:
                self.lists[p] = []; self.active[p] = 0; self.ref_patch[p] = -1
            else:
                if s["t_pop"][q]:
                    victim = lst.pop()
                    if self.ref_patch[p] == victim:
                        self.ref_patch[p] = -1                                               # deleteFeatureRef (visual_point.cpp:42-46)
                if s["t_push"][q] >= 0:
                    lst.insert(0, int(s["t_push"][q]))                                       # addFrameRef: push_front
                if s["t_ref"][q] != REF_KEEP:
                    self.ref_patch[p] = s["t_ref"][q]
                if s["t_flip"][q]:
                    self.normal[p] = -self.normal[p]
                if s["t_toggle"][q]:
                    self.ninit[p] = 1 - self.ninit[p]
            touched.append(p)
        tp = np.array(touched, np.int32)
        return dict(new_pos=s["new_pos"], new_keys=s["new_keys"], new_active=np.ones(n_new, np.uint8), obs=s["obs"],
                    touched=dict(point=tp, lists=[list(self.lists[p]) for p in touched], normal=self.normal[tp], normal_initialized=self.ninit[tp], ref_patch=self.ref_patch[tp],
                                 active=self.active[tp]), img=s["img"], img_slot=s["img_slot"])

    def step(self, rng, R_fw, t_fw, img, frame_id, **kw):
        return self.apply(self.generate(rng, R_fw, t_fw, img, frame_id, **kw))

    def flat(self, template=None):
        """the whole map as a RetrieveChainScenario (CSR, observations compacted in point / list order) + g2c: global observation index -> CSR position (-1: unreferenced)"""
        cs = copy.copy(template if template is not None else self.cs)
        cs.sel = copy.copy(cs.sel)
        n = self.n_points
        order = [g for lst in self.lists for g in lst]
        off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(l) for l in self.lists])
        g2c = np.full(self.n_obs, -1, np.int64); g2c[order] = np.arange(len(order))
        idx = np.array(order, np.int64)
        cs.sel.pos, cs.sel.keys, cs.sel.active = self.pos, self.keys, self.active
        cs.normal, cs.normal_initialized = self.normal, self.ninit
        cs.ref_patch = np.where(self.ref_patch >= 0, g2c[np.maximum(self.ref_patch, 0)], -1).astype(np.int32)
        cs.obs_offset = off
        for k in OBS_KEYS:
            setattr(cs, "obs_" + k, self.obs[k][idx] if len(idx) else self.obs[k][:0])
        cs.obs_id, cs.obs_img_idx, cs.obs_level = cs.obs_id.astype(np.int32), cs.obs_img_idx.astype(np.int32), cs.obs_level.astype(np.int32)
        cs.ref_imgs = np.stack(self.ref_imgs)
        return cs, g2c

    def c2g(self, g2c):
        """inverse of the translation flat() returned: CSR position -> global observation index"""
        keep = np.nonzero(g2c >= 0)[0]
        out = np.full(int(g2c.max()) + 1 if len(keep) else 0, -1, np.int64)
        out[g2c[keep]] = keep
        return out
