"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference) on CPU.

Run in the build container only (`python tests/golden/make_golden.py`); the GPU box has no
/root/reference.  Each .npz holds inputs (or the seed that regenerates them through
mvtracker_amd.synth) and the reference's outputs -- data only, no reference source text.
The oracle (oracle/mvt_oracle.py) is pinned against these files by tests/test_oracle_golden.py.

Two kNN variants are recorded wherever kNN is involved:
  *_cdist : the reference exactly as imported here (CPU fallback `torch.cdist + topk`).
  *_exact : the reference with its module-level `knn` swapped for cdist in
            `donot_use_mm_for_euclid_dist` mode, i.e. exact distances like its GPU backend
            (pointops.knn_query, not installable here).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402
from mvtracker_amd import synth  # noqa: E402

R = import_reference()
torch.manual_seed(0)
torch.set_num_threads(8)


def _knn_exact_ref(k, xyz_ref, xyz_query):
    d = torch.cdist(xyz_query, xyz_ref, p=2, compute_mode="donot_use_mm_for_euclid_dist")
    return torch.topk(d, k, dim=-1, largest=False, sorted=True)


def set_knn(mode):
    R.mvt.knn = _knn_exact_ref if mode == "exact" else R.mvt._knn_torch


def ref_model(seed=0):
    m = R.mvt.MVTracker(hidden_size=256).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed).items()}
    m.load_state_dict(sd, strict=True)
    return m, shapes


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def t(a):
    return torch.from_numpy(np.asarray(a))


# ---------------------------------------------------------------- state_dict contract
model, shapes = ref_model()
save("state_dict_shapes", keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
     n_params=sum(int(np.prod(s)) for s in shapes.values()), token_dim=model.updateformer_input_dim)

# ---------------------------------------------------------------- encoder (a3)
rng = np.random.default_rng(11)
img = rng.uniform(-1, 1, size=(2, 3, 64, 96)).astype(np.float32)
with torch.no_grad():
    save("encoder_64x96", seed=0, img=img, out=model.fnet(t(img)))

# ---------------------------------------------------------------- pyramid / unprojection (a5)
clip = synth.make_clip(5, V=2, T=3, H=64, W=64, N=4, invalid_frac=0.02)
fm = rng.standard_normal((1, 2, 3, 8, 16, 16)).astype(np.float32)
dp = torch.nn.functional.interpolate(t(clip["depths"]).reshape(-1, 1, 64, 64), scale_factor=0.25, mode="nearest")
dp = dp.reshape(1, 2, 3, 1, 16, 16)
pc = {}
for lvl in range(3):
    xyz, fvec, valid = R.mu.init_pointcloud_from_rgbd(t(fm), dp, t(clip["intrs"]), t(clip["extrs"]), stride=4, level=lvl,
                                                      return_validity_mask=True)
    pc[f"xyz{lvl}"], pc[f"fvec{lvl}"], pc[f"valid{lvl}"] = xyz, fvec, valid
save("pyramid_small", clip_seed=5, fmaps=fm, depths_strided=dp, intrs=clip["intrs"], extrs=clip["extrs"], **pc)

# ---------------------------------------------------------------- kNN + corr_sample (a6, a7)
B, P, C, M, K = 2, 400, 128, 24, 16
xyz = rng.uniform(-2, 2, size=(B, P, 3)).astype(np.float32)
fvec = rng.standard_normal((B, P, C)).astype(np.float32)
tgt = rng.standard_normal((B, M, C)).astype(np.float32)
crd = rng.uniform(-2, 2, size=(B, M, 3)).astype(np.float32)
outs = {}
for mode in ("cdist", "exact"):
    set_knn(mode)
    blk = R.mvt.PointcloudCorrBlock(k=K, groups=1, xyz=t(xyz), fvec=t(fvec), corr_add_neighbor_offset=True,
                                    corr_add_neighbor_xyz=False)
    outs["out_" + mode] = blk.corr_sample(t(tgt), t(crd))
    outs["idx_" + mode] = R.mvt.knn(K, t(xyz), t(crd))[1]
    outs["dist_" + mode] = R.mvt.knn(K, t(xyz), t(crd))[0]
blk2 = R.mvt.PointcloudCorrBlock(k=8, groups=4, xyz=t(xyz), fvec=t(fvec), corr_add_neighbor_offset=True,
                                 corr_add_neighbor_xyz=True)
outs["out_exact_k8_g4_xyz"] = blk2.corr_sample(t(tgt), t(crd))
save("corr_sample_small", xyz=xyz, fvec=fvec, targets=tgt, coords=crd, **outs)

# ---------------------------------------------------------------- bilinear-window CorrBlock (a7')
fm2 = rng.standard_normal((1, 2, 32, 24, 40)).astype(np.float32)
tg2 = rng.standard_normal((1, 2, 10, 32)).astype(np.float32)
cd2 = np.stack([rng.uniform(-3, 43, size=(1, 2, 10)), rng.uniform(-3, 27, size=(1, 2, 10))], -1).astype(np.float32)
wc = {}
for r in (3, 4):
    cb = R.spa.CorrBlock(t(fm2), num_levels=3, radius=r)
    wc[f"out_r{r}"] = cb.corr_sample(t(tg2), t(cd2))
save("window_corr_small", fmaps=fm2, targets=tg2, coords=cd2, **wc)

# ---------------------------------------------------------------- embeddings (a8, a9)
c0 = rng.uniform(-3, 3, size=(1, 1, 7, 3)).astype(np.float32)
pe = R.emb.get_3d_sincos_pos_embed_from_grid(582, t(c0))
te = R.emb.get_1d_sincos_pos_embed_from_grid(582, (torch.linspace(0, 11, 12).reshape(1, 12, 1) / 12)[0])
fl = rng.uniform(-0.5, 0.5, size=(5, 12, 3)).astype(np.float32)
save("embeddings", coords0=c0, pos_embed=pe, times_embed=te, flows=fl, flow_embed=R.emb.get_3d_embedding(t(fl), 64, True))

# ---------------------------------------------------------------- updater transformer (a11)
xin = rng.standard_normal((1, 16, 12, 581)).astype(np.float32)
with torch.no_grad():
    save("updateformer_16x12", seed=0, x=xin, out=model.updateformer(t(xin)))

# ---------------------------------------------------------------- refinement loop, one window (a6-a13)
clip = synth.make_clip(21, V=2, T=12, H=128, W=128, N=12)
with torch.no_grad():
    fmaps = model.fnet(2 * (t(clip["rgbs"]).reshape(-1, 3, 128, 128) / 255.0) - 1).reshape(1, 2, 12, 128, 32, 32)
    dstr = torch.nn.functional.interpolate(t(clip["depths"]).reshape(-1, 1, 128, 128), scale_factor=0.25, mode="nearest")
    dstr = dstr.reshape(1, 2, 12, 1, 32, 32)
    q = t(clip["query_points"])
    N = q.shape[1]
    coords_init = q[:, None, :, 1:].repeat(1, 12, 1, 1)
    vis_init = torch.full((1, 12, N, 1), 10.0)
    tmask = torch.ones(1, 12, N, 1, dtype=torch.bool)
    feat = t(rng.standard_normal((1, 1, N, 128)).astype(np.float32)).repeat(1, 12, 1, 1)
    rw = {}
    for mode in ("cdist", "exact"):
        set_knn(mode)
        cp, vis, _ = model.forward_iteration(fmaps, dstr, t(clip["intrs"]), t(clip["extrs"]), coords_init, vis_init, tmask,
                                             iters=3, feat_init=feat)
        rw["coords_" + mode] = torch.stack(cp)
        rw["vis_" + mode] = vis
save("refine_window_small", clip_seed=21, feat_init=feat, **rw)

# ---------------------------------------------------------------- end to end (a2), tiny + two-window
for name, kw in (("e2e_tiny", dict(seed=31, V=2, T=12, H=128, W=128, N=16)),
                 ("e2e_two_windows", dict(seed=32, V=2, T=18, H=128, W=128, N=12, late_queries=True,
                                          query_frames=(3, 7))),
                 ("e2e_short_clip", dict(seed=33, V=1, T=8, H=128, W=128, N=6))):
    clip = synth.make_clip(**kw)
    res = {}
    for mode in ("cdist", "exact"):
        set_knn(mode)
        logits = []
        orig = model.forward_iteration

        def spy(*a, **k):
            o = orig(*a, **k)
            logits.append(o[1].clone())
            return o

        model.forward_iteration = spy
        with torch.no_grad():
            r = model(t(clip["rgbs"]), t(clip["depths"]), t(clip["query_points"]), t(clip["intrs"]), t(clip["extrs"]),
                      iters=4)
        model.forward_iteration = orig
        res["traj_" + mode] = r["traj_e"]
        res["vis_" + mode] = r["vis_e"]
        res["feat_init_" + mode] = r["feat_init"]
        res["last_window_logits_" + mode] = logits[-1]
        res["n_windows"] = len(logits)
    save(name, **{k: np.asarray(v) for k, v in kw.items()}, **res)

# ---------------------------------------------------------------- predictor (a1) G4
clip = synth.make_clip(41, V=2, T=12, H=160, W=192, N=5)
set_knn("exact")
pred = R.pred.EvaluationPredictor(model, interp_shape=(128, 160), grid_size=3, n_grids_per_view=2, n_iters=2)
captured = {}
orig_fwd = model.forward


def spy_fwd(rgbs, depths=None, query_points=None, intrs=None, extrs=None, **kw):
    captured["query_points"] = query_points.clone()
    captured["intrs"] = intrs.clone()
    captured["depths_sample"] = depths[0, :, :, 0, ::8, ::8].clone()
    captured["rgbs_sample"] = rgbs[0, :, :, :, ::8, ::8].clone()
    return orig_fwd(rgbs, depths=depths, query_points=query_points, intrs=intrs, extrs=extrs, **kw)


model.forward = spy_fwd
with torch.no_grad():
    pr = pred(t(clip["rgbs"]), t(clip["depths"]), t(clip["query_points"]), t(clip["intrs"]), t(clip["extrs"]))
model.forward = orig_fwd
save("predictor_small", clip_seed=41, traj_e=pr["traj_e"], vis_e=pr["vis_e"], vis_e_as_prob=pr["vis_e_as_prob"],
     model_query_points=captured["query_points"], model_intrs=captured["intrs"],
     depths_sample=captured["depths_sample"], rgbs_sample=captured["rgbs_sample"])

# helpers (a15, a16)
w = rng.uniform(-1, 1, size=(4, 9, 3)).astype(np.float32)
pix, z = R.mu.world_space_to_pixel_xy_and_camera_z(t(w), t(clip["intrs"][0, 0, :4]), t(clip["extrs"][0, 0, :4]))
im = rng.standard_normal((2, 3, 9, 11)).astype(np.float32)
xs = rng.uniform(-2, 13, size=(2, 17)).astype(np.float32)
ys = rng.uniform(-2, 11, size=(2, 17)).astype(np.float32)
save("helpers", world=w, intrs=clip["intrs"][0, 0, :4], extrs=clip["extrs"][0, 0, :4], pix=pix, z=z, im=im, xs=xs, ys=ys,
     bil=R.mu.bilinear_sample2d(t(im), t(xs), t(ys)), grid5=R.mu.get_points_on_a_grid(5, (48, 64)),
     grid3c=R.mu.get_points_on_a_grid(3, (50, 50), center=(20.5, 31.25)))
print("done")
