"""Pins oracle/anomalyclip_oracle.py against the golden vectors produced by the REFERENCE
(tests/golden/make_golden.py).  CPU only.  Tolerances are fp32 round-off level: both sides are
fp32 CPU arithmetic of the same algorithm in a different operation order."""
import numpy as np
import pytest
import torch

from anomalyclip_amd import init_weights as IW
from oracle import anomalyclip_oracle as O
import recipes as R

torch.set_grad_enabled(False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=2e-4, atol=2e-5):
    a, b = T(a).double(), T(b).double()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err}"


def test_tokens_match_table(golden, prompts_table):
    g = golden("tokens")
    for key in ("ucf", "sht", "xd"):
        assert np.array_equal(g[key], np.asarray(prompts_table[key]["tokenized_prompts"], dtype=np.int32))
        # "X X X X X X X X Abuse." -> [49406, 343*8, 7678, 269, 49407, 0...]
    assert g["ucf"][0, :12].tolist() == [49406] + [343] * 8 + [7678, 269, 49407]


def test_vit_tiny(golden):
    g = golden("vit_tiny")
    sd = IW.init_vit_state_dict(IW.TINY, int(g["seed"]))
    frames = T(g["frames"])
    assert torch.equal(frames, R.vit_frames(int(g["seed"]), 3, 32))
    close(O.vit_forward(sd, frames), g["out"])
    toks = O.vit_forward(sd, frames, return_tokens=True)
    close(toks, g["block1"])


def test_vit_b16(golden):
    g = golden("vit_b16")
    sd = IW.init_vit_state_dict(IW.VIT_B16, int(g["seed"]))
    frames = R.vit_frames(int(g["seed"]), 2, 224)
    assert abs(frames.double().sum().item() - float(g["frames_checksum"])) < 1e-6
    close(O.vit_forward(sd, frames), g["out"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("tag,geom,key", [("text_tiny", IW.TINY, "ucf"), ("text_b16_xd", IW.VIT_B16, "xd")])
def test_text(golden, prompts_table, tag, geom, key):
    g = golden(tag)
    toks = torch.tensor(prompts_table[key]["tokenized_prompts"], dtype=torch.int32)
    hc = IW.HeadConfig(num_classes=toks.shape[0], normal_id=prompts_table[key]["normal_id"])
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, int(g["seed"]), with_image_encoder=False)
    eot = toks.argmax(-1)
    assert np.array_equal(eot.numpy(), g["eot"])
    close(O.text_features(sd, eot, geom.transformer_heads), g["out"], rtol=1e-3, atol=1e-4)


def test_selector(golden):
    g = golden("selector")
    inp = R.selector_inputs(int(g["seed"]))
    x, tf, nc = inp["x"], inp["tf"], inp["nc"]
    ev, _, _ = O.selector_logits(x, tf, nc, 7, inp["rm0"], inp["rv0"], training=False)
    close(ev, g["eval_logits"])
    out = O.selector_train(x, tf, inp["labels"], nc, 7, inp["rm0"], inp["rv0"], inp["topk_mask"],
                           inp["bottomk_mask"], 32, 16, 3, 3)
    logits, lt, lb, ia, in_, ba, rm, rv = out
    close(logits, g["logits"])
    assert torch.equal(ia, T(g["idx_topk_abn"]))          # bit-exact segment indices
    assert torch.equal(in_, T(g["idx_topk_nor"]))
    assert torch.equal(ba, T(g["idx_bottomk_abn"]))
    close(lt, g["logits_topk"])
    close(lb, g["logits_bottomk"])
    close(rm, g["rm1"])
    close(rv, g["rv1"])


def test_temporal(golden):
    """a6 (tilings) is pinned by the reference's einops; a7 is PARITY UNPINNED (restated dep)."""
    g = golden("temporal")
    hc = IW.HeadConfig(emb_size=64, heads=2, depth=2)
    sd = IW.init_temporal_state_dict(int(g["in_size"]), hc, int(g["seed"]))
    for S in (1, 2, 3):
        rows = g[f"feats_S{S}"].shape[0]
        assert np.array_equal(O.test_tile_index(rows, 32, 16, S).numpy(), g[f"tile_index_S{S}"])
        close(O.temporal_forward(T(g[f"feats_S{S}"]), sd, hc, S, True), g[f"scores_test_S{S}"])
    close(O.temporal_forward(T(g["feats_train"]), sd, hc, 1, False), g["scores_train"])


def test_loss(golden):
    g = golden("loss")
    with torch.enable_grad():
        s1 = T(g["sim"]).requires_grad_(True)
        s2 = T(g["sim_topk"]).requires_grad_(True)
        s3 = T(g["scores"]).requires_grad_(True)
        outs = O.compute_loss(s1, s2, T(g["labels"]), s3, T(g["idx_topk_abn"]), T(g["idx_topk_nor"]),
                              T(g["idx_bottomk_abn"]), normal_id=7, num_topk=3, num_segments=32,
                              frames_per_segment=16)
        outs[0].backward()
    close(torch.stack([o.detach() for o in outs]), g["losses"], rtol=1e-5, atol=1e-6)
    close(s1.grad, g["g_sim"], rtol=1e-4, atol=1e-8)
    close(s2.grad, g["g_sim_topk"], rtol=1e-4, atol=1e-8)
    close(s3.grad, g["g_scores"], rtol=1e-4, atol=1e-8)


def test_e2e_tiny(golden, prompts_table):
    g = golden("e2e_tiny")
    seed = int(g["seed"])
    geom = IW.TINY
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    toks = torch.tensor(prompts_table["ucf"]["tokenized_prompts"], dtype=torch.int32)
    eot = toks.argmax(-1)
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, seed)
    inp = R.e2e_inputs(seed, geom.embed_dim)
    sim, sc = O.anomaly_clip_forward_test(sd, hc, inp["test_feats"], inp["nc"], eot, geom.transformer_heads, 2)
    close(sim, g["test_sim"], rtol=1e-3, atol=1e-4)
    close(sc, g["test_scores"], rtol=1e-3, atol=1e-5)
    sim, sc = O.anomaly_clip_forward_test(sd, hc, None, inp["nc"], eot, geom.transformer_heads, 1,
                                          frames=inp["frames"])
    close(sim, g["test_frames_sim"], rtol=1e-3, atol=1e-4)
    close(sc, g["test_frames_scores"], rtol=1e-3, atol=1e-5)
    # train branch + loss + gradients wrt the trainable parameters
    names = ["temporal_model.projection.weight", "temporal_model.classifier.linear.weight",
             "prompt_learner.ctx", "text_encoder.text_projection"]
    with torch.enable_grad():
        for n in names:
            sd[n] = sd[n].clone().requires_grad_(True)
        lg, lt, scr, ia, in_, ba, rm, rv = O.anomaly_clip_forward_train(
            sd, hc, inp["train_feats"], inp["labels"], inp["nc"], eot, geom.transformer_heads,
            inp["mask"], inp["mask"])
        outs = O.compute_loss(lg, lt, inp["labels"], scr, ia, in_, ba, normal_id=7, num_topk=3,
                              num_segments=32, frames_per_segment=16)
        outs[0].backward()
    assert torch.equal(ia, T(g["idx_topk_abn"])) and torch.equal(in_, T(g["idx_topk_nor"]))
    assert torch.equal(ba, T(g["idx_bottomk_abn"]))
    close(lg, g["train_logits"], rtol=1e-3, atol=1e-4)
    close(lt, g["train_logits_topk"], rtol=1e-3, atol=1e-4)
    close(scr, g["train_scores"], rtol=1e-3, atol=1e-5)
    close(torch.stack([o.detach() for o in outs]), g["losses"], rtol=1e-4, atol=1e-6)
    close(rm, g["rm1"])
    close(rv, g["rv1"])
    for n in names:
        ref = T(g["grad:" + n])
        scale = ref.abs().max().item()
        close(sd[n].grad, ref, rtol=2e-3, atol=2e-4 * scale)


def test_tables(golden):
    g = golden("tables")
    for stride in (1, 2):
        for T_ in (1, 511, 512, 513, 1000, 1025, 5000):
            si, S = O.start_indices(T_, 32, 16, stride)
            assert np.array_equal(si, g[f"start_T{T_}_s{stride}"].astype(np.int64)) and S == len(si) // 32
    lrs = [O.warmup_cosine_lr(1e-5, ep, 5, 50) for ep in range(55)]
    assert np.allclose(lrs, g["lr_table"], rtol=1e-12, atol=0)


def test_config0_shanghaitech_eval_from_feature_files(golden, prompts_table, tmp_path):
    """BASELINE.json configs[0], end to end on the CPU path: eight synthetic `.npy` feature files (T = 300 ... 5000) ->
    this package's loader index logic (feature_index, one vectorised gather) -> ShanghaiTech head (18 classes, depth 2,
    concat on) -> scores / class probabilities -> padded frames stripped, against what the REFERENCE produced for the same
    files through its own FeatureDataset + DataLoader + AnomalyCLIP (fixture config0.npz, tests/golden/make_golden.py)."""
    from anomalyclip_amd import feature_index as FI
    g = golden("config0")
    assert tuple(g["lengths"]) == R.CONFIG0_LENGTHS
    paths, arrays, labels, nc = R.config0_feature_files(tmp_path)
    hc = IW.SHT_HEAD
    toks = torch.tensor(prompts_table["sht"]["tokenized_prompts"], dtype=torch.int32)
    sd = IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, int(g["seed"]), with_image_encoder=False)
    eot = toks.argmax(-1)
    tf = O.text_features(sd, eot, IW.VIT_B16.transformer_heads)          # independent of the video: evaluated once
    for i, (path, T_) in enumerate(zip(paths, R.CONFIG0_LENGTHS)):
        feats, S = FI.gather_test_features(np.load(path), 32, 16, 1, 1)    # [1, 512*S, 512]
        assert S == int(g[f"S{i}"]) == -(-T_ // 512) and feats.shape == (1, 512 * S, 512)
        sim, sc = O.anomaly_clip_forward_test(sd, hc, torch.from_numpy(feats).unsqueeze(0), nc, eot,
                                              IW.VIT_B16.transformer_heads, S, text_feats=tf)
        assert sim.shape == (512 * S, 17) and sc.shape == (512 * S,)
        probs, sc = O.eval_postprocess(sim, sc, T_)                        # truncation to the real frames
        assert probs.shape == (T_, 17) and sc.shape == (T_,)
        assert np.array_equal(g[f"labels{i}"].astype(np.int64), labels[i])
        close(sc, g[f"scores{i}"])
        close(probs[::8], g[f"probs8_{i}"])
        assert abs(float(probs.double().sum()) - float(g[f"probsum{i}"])) < 1e-4 * abs(float(g[f"probsum{i}"]))


def test_bf16x6_product_is_f32_accurate():
    """The arithmetic of the library's f32x6 GEMM mode, on the CPU: the three-plane split is exact (hi + mid + lo == x bit for
    bit), and the six leading cross products reproduce the exact product to <= 4 * 2^-24 of sum |a||w| -- the three dropped
    terms are <= 2^-24 of the leading one each, plus one final f32 rounding.  (An f32 FMA chain over K = 768 is allowed
    K * 2^-24 by the same measure; the GPU tests compare both against fp64.)"""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(96, 768, generator=g) * torch.exp2(torch.randint(-8, 8, (96, 1), generator=g).float())
    a[:, 11] *= 80.0
    w = torch.randn(64, 768, generator=g) * 0.05
    hi, mid, lo = O.split_bf16x3(a)
    assert torch.equal(hi + mid + lo, a)
    assert torch.equal(hi, hi.to(torch.bfloat16).float()) and torch.equal(lo, lo.to(torch.bfloat16).float())
    y = O.matmul_bf16x6(a, w).double()
    ref = a.double() @ w.double().t()
    bound = 4 * 2.0 ** -24 * (a.double().abs() @ w.double().abs().t())
    assert bool(((y - ref).abs() <= bound + 1e-300).all()), float(((y - ref).abs() / bound).max())
    plain = (a.to(torch.bfloat16).float().double() @ w.to(torch.bfloat16).float().double().t())
    assert float((plain - ref).abs().max()) > 1e3 * float((y - ref).abs().max())          # what a single bf16 product loses

