"""Generates tests/golden/*.npz (+ anomalyclip_amd/data/prompts.json) by EXECUTING THE REFERENCE.

Run in the development container only (needs /root/reference):
    python tests/golden/make_golden.py

What is captured is DATA: seeded inputs and the outputs the reference's own modules produce for
them.  No reference source is stored.  Weights are not stored either: the
fixtures record the seed of anomalyclip_amd.init_weights (our own deterministic recipe), the
reference modules get those weights through load_state_dict, and the tests regenerate them.

Pinned by reference code: a1 ViT, a2 text path + tokeniser, a3/a4 selector, a5 assembly,
a6 tilings, a8 classifier, a9 loss, start-index logic, LR schedule.
NOT pinned by reference code: a7 axial transformer (un-vendored dependency; the fixture comes
from oracle/axial_attention_restated.py running under the reference's TemporalModel).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_harness as H  # noqa: E402
import recipes as R  # noqa: E402
from anomalyclip_amd import init_weights as IW  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)
ns = H.ref_modules()
OUT = HERE


def save(name, **arrs):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------- 1. prompts / token ids
LABEL_SETS = {"ucf": ("data/ucf_labels.csv", 7), "sht": ("data/sht_labels.csv", 8), "xd": ("data/xd_labels.csv", 4)}


def gen_prompts():
    import pandas as pd
    table = {}
    for key, (csv, normal_id) in LABEL_SETS.items():
        df = pd.read_csv(os.path.join(H.REF_ROOT, csv))
        classnames = sorted(c for i, c in df.values.tolist())            # anomaly_clip.py:70
        names = [n.replace("_", " ") for n in classnames]                # coop.py:51
        prompts = [" ".join(["X"] * 8) + " " + n + "." for n in names]   # coop.py:44,53
        toks = torch.cat([ns.clip.tokenize(p) for p in prompts])         # coop.py:56
        table[key] = {
            "classnames": classnames, "normal_id": normal_id, "n_ctx": 8,
            "prompts": prompts, "tokenized_prompts": toks.tolist(),
            "eot_index": toks.argmax(dim=-1).tolist(),
        }
    os.makedirs(os.path.join(REPO, "anomalyclip_amd", "data"), exist_ok=True)
    with open(os.path.join(REPO, "anomalyclip_amd", "data", "prompts.json"), "w") as f:
        json.dump(table, f)
    save("tokens", **{k: np.asarray(v["tokenized_prompts"], dtype=np.int32) for k, v in table.items()})
    return table


def toks_of(table, key):
    return torch.tensor(table[key]["tokenized_prompts"], dtype=torch.int32)


# ---------------------------------------------------------------- 2/3. ViT
def gen_vit(tag, geom, seed, nframes, store_tokens):
    sd = IW.init_vit_state_dict(geom, seed, prefix="")
    vit = ns.clip_model.VisionTransformer(geom.image_resolution, geom.vision_patch_size, geom.vision_width,
                                          geom.vision_layers, geom.vision_heads, geom.embed_dim)
    vit.load_state_dict(sd, strict=True)
    vit.eval()
    frames = R.vit_frames(seed, nframes, geom.image_resolution)
    out = vit(frames)
    arrs = dict(seed=seed, frames_seed=seed + 100, out=out)
    if store_tokens:
        # activations after ln_pre / after each block, captured with forward hooks
        acts = []
        hooks = [blk.register_forward_hook(lambda m, i, o: acts.append(o.permute(1, 0, 2).clone()))
                 for blk in vit.transformer.resblocks]
        vit(frames)
        for h in hooks:
            h.remove()
        arrs["frames"] = frames
        for i, a in enumerate(acts):
            arrs[f"block{i}"] = a
    else:
        arrs["frames_checksum"] = frames.double().sum()
    save(tag, **arrs)


# ---------------------------------------------------------------- 4. text path
def gen_text(tag, geom, seed, table, key):
    toks = toks_of(table, key)
    hc = IW.HeadConfig(num_classes=toks.shape[0], normal_id=table[key]["normal_id"])
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, seed, with_image_encoder=False)
    torch.manual_seed(0)
    clip_model = ns.clip_model.CLIP(**geom.as_kwargs()).float()
    te = ns.text_encoder.TextEncoder(clip_model)
    te_sd = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
    te.load_state_dict(te_sd, strict=True)
    prompts = torch.cat([sd["prompt_learner.token_prefix"], sd["prompt_learner.ctx"],
                         sd["prompt_learner.token_suffix"]], dim=1)
    out = te(prompts, toks)
    save(tag, seed=seed, out=out, eot=toks.argmax(-1))


# ---------------------------------------------------------------- 5. selector
def gen_selector(seed=11):
    inp = R.selector_inputs(seed)
    x, tf, nc, labels, rm0, rv0, m1, m2 = (inp[k] for k in ("x", "tf", "nc", "labels", "rm0", "rv0", "topk_mask", "bottomk_mask"))
    N, L, C, normal_id = 32, 16, 14, 7
    sel = ns.selector_model.SelectorModel([str(i) for i in range(C)], normal_id, torch.nn.Parameter(torch.ones([])),
                                          N, L, 0.7, 0.7, 3, 3)
    sel.bn_layer.running_mean.copy_(rm0)
    sel.bn_layer.running_var.copy_(rv0)
    sel.eval()
    ev = sel(x, tf, labels, nc, True)
    masks = [m1, m2]

    def fake_mask(logits):
        return (masks[0].unsqueeze(2).expand(-1, -1, logits.shape[-1]),
                masks[1].unsqueeze(2).expand(-1, -1, logits.shape[-1]))

    sel.generate_mask = fake_mask
    sel.train()
    lg, lt, lb, ia, in_, ba = sel(x, tf, labels, nc, False)
    save("selector", seed=seed, eval_logits=ev, logits=lg, logits_topk=lt, logits_bottomk=lb,
         idx_topk_abn=ia, idx_topk_nor=in_, idx_bottomk_abn=ba,
         rm1=sel.bn_layer.running_mean, rv1=sel.bn_layer.running_var)


# ---------------------------------------------------------------- 6. temporal (a6 pinned, a7 unpinned)
def gen_temporal(seed=21):
    hc = IW.HeadConfig(emb_size=64, heads=2, depth=2, num_segments=32, seg_length=16)
    in_size = 16
    sd = IW.init_temporal_state_dict(in_size, hc, seed, prefix="")
    tm = ns.temporal_model.TemporalModel(in_size, hc.emb_size, 1, hc.heads, None, hc.depth, 32, 16)
    tm.load_state_dict(sd, strict=True)
    tm.eval()
    g = torch.Generator().manual_seed(seed + 1)
    arrs = dict(seed=seed, in_size=in_size)
    for S in (1, 2, 3):
        nb = 2 if S == 1 else 1
        f = torch.randn(nb * 512 * S, in_size, generator=g) * 0.5     # b=2 (e.g. 2 crops) for S=1
        arrs[f"feats_S{S}"] = f
        arrs[f"scores_test_S{S}"] = tm(f, S, True)
    f = torch.randn(2 * 512, in_size, generator=g) * 0.5
    arrs["feats_train"] = f
    arrs["scores_train"] = tm(f, 1, False)
    # tiling index tables (pure einops of the reference; pinned)
    from einops import rearrange
    for S in (1, 2, 3):
        idx = torch.arange((2 if S == 1 else 1) * 512 * S).view(-1, 1)
        t = rearrange(idx, "(b n s l) d -> b n s l d", n=32, s=S, l=16)
        t = rearrange(t, "b n s l d -> (b s) n l d")
        arrs[f"tile_index_S{S}"] = t.reshape(-1)
    arrs["axial_source"] = H.AXIAL_SOURCE
    save("temporal", **arrs)


# ---------------------------------------------------------------- 7. loss
def gen_loss(seed=31):
    g = torch.Generator().manual_seed(seed)
    B, N, L, C1, K, normal_id = 8, 32, 16, 13, 3, 7
    sim = torch.randn(B * N * L, C1, generator=g)
    sim_topk = torch.randn(B * K * L, C1, generator=g)
    scores = torch.rand(B * N * L, generator=g) * 0.9 + 0.05
    labels = torch.tensor([0, 3, 9, 13, 7, 7, 7, 7])
    ia = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(B // 2)])
    in_ = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(B // 2)])
    ba = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(B // 2)])
    crit = ns.loss.ComputeLoss(normal_id, K, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, L, N)
    with torch.enable_grad():
        s1 = sim.clone().requires_grad_(True)
        s2 = sim_topk.clone().requires_grad_(True)
        s3 = scores.clone().requires_grad_(True)
        outs = crit(s1, s2, labels.clone(), s3, ia, in_, ba)
        outs[0].backward()
    save("loss", sim=sim, sim_topk=sim_topk, scores=scores, labels=labels, idx_topk_abn=ia, idx_topk_nor=in_,
         idx_bottomk_abn=ba, losses=torch.stack([o.detach() for o in outs]),
         g_sim=s1.grad, g_sim_topk=s2.grad, g_scores=s3.grad)


# ---------------------------------------------------------------- 8. end-to-end (tiny geometry)
def gen_e2e(table, seed=41):
    geom = IW.TINY
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    toks = toks_of(table, "ucf")
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, seed)
    H.patch_clip_load(ns, geom.as_kwargs(), seed)
    cfgs = dict(arch="ViT-B/16", labels_file=os.path.join(H.REF_ROOT, "data/ucf_labels.csv"), emb_size=hc.emb_size,
                depth=hc.depth, heads=hc.heads, dim_heads=None, num_segments=32, seg_length=16,
                concat_features=False, normal_id=7, stride=1, load_from_features=True,
                select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=1, num_topk=3,
                num_bottomk=3, n_ctx=8, shared_context=False, ctx_init="")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ns.anomaly_clip.AnomalyCLIP(**cfgs)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    assert torch.equal(net.tokenized_prompts, toks)
    D = geom.embed_dim
    inp = R.e2e_inputs(seed, D)
    nc = inp["nc"]
    arrs = dict(seed=seed)
    # test mode, features, S=2
    net.eval()
    sim, sc = net(inp["test_feats"], torch.zeros(1000), nc, 2, True)
    arrs.update(test_sim=sim, test_scores=sc)
    # test mode from FRAMES (load_from_features False): 1 video, S=1 -> 512 tiny frames
    net.load_from_features = False
    sim2, sc2 = net(inp["frames"], torch.zeros(500), nc, 1, True)
    arrs.update(test_frames_sim=sim2, test_frames_scores=sc2)
    net.load_from_features = True
    # train mode, B=4
    net.train()
    labels = inp["labels"]
    feats_tr = inp["train_feats"]
    m1 = inp["mask"]
    net.selector_model.generate_mask = lambda logits: (
        m1.unsqueeze(2).expand(-1, -1, logits.shape[-1]), m1.unsqueeze(2).expand(-1, -1, logits.shape[-1]))
    crit = ns.loss.ComputeLoss(7, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
    # gradients of the trainable parameters (SURVEY 8c fixture 8)
    net.zero_grad()
    for p in net.parameters():
        p.requires_grad_(False)
    train_params = {"temporal_model.projection.weight": net.temporal_model.projection.weight,
                    "temporal_model.classifier.linear.weight": net.temporal_model.classifier.linear.weight,
                    "prompt_learner.ctx": net.prompt_learner.ctx,
                    "text_encoder.text_projection": net.text_encoder.text_projection}
    for p in net.temporal_model.parameters():
        p.requires_grad_(True)
    for p in train_params.values():
        p.requires_grad_(True)
    # reset BN running stats so the second train call starts from the same state
    net.selector_model.bn_layer.running_mean.copy_(sd["selector_model.bn_layer.running_mean"])
    net.selector_model.bn_layer.running_var.copy_(sd["selector_model.bn_layer.running_var"])
    with torch.enable_grad():
        lg, lt, sc, ia, in_, ba = net(feats_tr, labels, nc)
        outs = crit(lg, lt, labels.clone(), sc, ia, in_, ba)
        outs[0].backward()
    arrs.update(train_logits=lg, train_logits_topk=lt,
                train_scores=sc, idx_topk_abn=ia, idx_topk_nor=in_, idx_bottomk_abn=ba,
                losses=torch.stack([o.detach() for o in outs]),
                rm1=net.selector_model.bn_layer.running_mean, rv1=net.selector_model.bn_layer.running_var)
    for k, p in train_params.items():
        arrs["grad:" + k] = p.grad
    gsum = {k: float(p.grad.double().abs().sum()) for k, p in net.temporal_model.named_parameters() if p.grad is not None}
    arrs["temporal_grad_abs_sums"] = np.asarray([gsum[k] for k in sorted(gsum)])
    arrs["temporal_grad_names"] = np.asarray(sorted(gsum))
    arrs["axial_source"] = H.AXIAL_SOURCE
    save("e2e_tiny", **arrs)


# ---------------------------------------------------------------- 9. data-side index tables, LR schedule
def gen_tables():
    arrs = {}
    fd = None
    try:
        import importlib
        fd = importlib.import_module("src.data.components.feature_dataset")
    except Exception as e:  # pragma: no cover
        print("feature_dataset import failed:", e)
    if fd is not None:
        class _Rec:
            def __init__(self, n):
                self.num_frames = n
        for stride in (1, 2):
            for T in (1, 511, 512, 513, 1000, 1025, 5000):
                ds = fd.FeatureDataset.__new__(fd.FeatureDataset) if hasattr(fd, "FeatureDataset") else None
                if ds is None:
                    # class name differs; find the class that has _get_start_indices
                    for v in vars(fd).values():
                        if isinstance(v, type) and hasattr(v, "_get_start_indices"):
                            ds = v.__new__(v)
                            break
                ds.test_mode, ds.num_segments, ds.frames_per_segment, ds.stride = True, 32, 16, stride
                si = ds._get_start_indices(_Rec(T))
                arrs[f"start_T{T}_s{stride}"] = np.asarray(si)
                # frame index table of _get (feature_dataset.py:359-367), T frames available
                fi = [(int(s) + i * stride) % T for s in si for i in range(16)]
                arrs[f"frames_T{T}_s{stride}"] = np.asarray(fi, dtype=np.int64)
    # LR table: WarmupCosineAnnealingLR(warmup 5, total 50) stepped per epoch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-5}], weight_decay=0.2)
    succ = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 50.0)
    sch = ns.scheduler.WarmupCosineAnnealingLR(optimizer=opt, successor=succ, warmup_epochs=5, total_epoch=50)
    lrs = []
    for ep in range(55):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    arrs["lr_table"] = np.asarray(lrs, dtype=np.float64)
    save("tables", **arrs)


# ---------------------------------------------------------------- 10. BASELINE configs[0], end to end through the REFERENCE
def gen_config0(table, seed=51):
    """ShanghaiTech-shaped evaluation from pre-extracted features (SURVEY.md 8d Config 1): eight synthetic .npy files ->
    the reference's FeatureDataset (test mode) + DataLoader(batch_size=1) -> the reference's AnomalyCLIP (ShanghaiTech
    head: 18 classes, depth 2, concat on; full ViT-B/16 text tower, random init) -> the two post-processing lines of
    test_step (anomaly_clip_module.py:474-483, restated here because the module itself needs Lightning).  Stored:
    per video the scores of every real frame, every 8th row of class_probs, the f64 sum of all of class_probs, the
    per-frame labels and segment_size the reference's dataset produced."""
    import importlib
    import tempfile
    fd = importlib.import_module("src.data.components.feature_dataset")
    geom, hc = IW.VIT_B16, IW.SHT_HEAD
    toks = toks_of(table, "sht")
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, seed, with_image_encoder=False)
    H.patch_clip_load(ns, geom.as_kwargs(), seed)
    cfgs = dict(arch="ViT-B/16", labels_file=os.path.join(H.REF_ROOT, "data/sht_labels.csv"), emb_size=hc.emb_size,
                depth=hc.depth, heads=hc.heads, dim_heads=None, num_segments=32, seg_length=16,
                concat_features=True, normal_id=hc.normal_id, stride=1, load_from_features=True,
                select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=1, num_topk=3,
                num_bottomk=3, n_ctx=8, shared_context=False, ctx_init="")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ns.anomaly_clip.AnomalyCLIP(**cfgs)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("image_encoder.") for k in missing), missing      # features path: the ViT is never run
    net.eval()
    arrs = dict(seed=seed, lengths=np.asarray(R.CONFIG0_LENGTHS), axial_source=H.AXIAL_SOURCE)
    with tempfile.TemporaryDirectory() as d:
        paths, arrays, labels, nc = R.config0_feature_files(d)
        ann, tmp = R.config0_annotation_files(d, paths, labels)
        ds = fd.VideoFrameDataset(root_path=d, annotationfile_path=ann, normal_id=hc.normal_id, num_segments=32,
                               frames_per_segment=16, test_mode=True, ncrops=1, temporal_annotation_file=tmp,
                               labels_file=os.path.join(H.REF_ROOT, "data/sht_labels.csv"), stride=1)
        loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
        for i, (image_features, lab, label, segment_size, path) in enumerate(loader):
            lab = lab.squeeze(0)
            similarity, scores = net(image_features, lab, nc, segment_size, test_mode=True)
            class_probs = torch.softmax(similarity, dim=1) * scores.unsqueeze(1)       # :474-477
            n = lab.shape[0]                                                           # :480-483
            class_probs, scores = class_probs[:n], scores[:n]
            assert n == R.CONFIG0_LENGTHS[i] and np.array_equal(lab.numpy(), labels[i])
            arrs[f"scores{i}"] = scores
            arrs[f"probs8_{i}"] = class_probs[::8]
            arrs[f"probsum{i}"] = class_probs.double().sum()
            arrs[f"labels{i}"] = lab.numpy().astype(np.int8)
            arrs[f"S{i}"] = int(segment_size)
    save("config0", **arrs)


if __name__ == "__main__":
    if sys.argv[1:] == ["config0"]:
        gen_config0(json.load(open(os.path.join(REPO, "anomalyclip_amd", "data", "prompts.json"))))
        sys.exit(0)
    table = gen_prompts()
    gen_vit("vit_tiny", IW.TINY, seed=1, nframes=3, store_tokens=True)
    gen_vit("vit_b16", IW.VIT_B16, seed=2, nframes=2, store_tokens=False)
    gen_text("text_tiny", IW.TINY, 3, table, "ucf")
    gen_text("text_b16_xd", IW.VIT_B16, 4, table, "xd")
    gen_selector()
    gen_temporal()
    gen_loss()
    gen_e2e(table)
    gen_tables()
    gen_config0(table)
