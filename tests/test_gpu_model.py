"""-m gpu: the assembled HIP path (through the C ABI) against (a) the golden vectors produced by the
REFERENCE and (b) the oracle on the same seeded inputs.  Tolerance from BASELINE.json north_star:
logits within 1e-3 relative; the exact-f32 MFMA path is held to a 10x tighter bound here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from anomalyclip_amd import init_weights as IW
from anomalyclip_amd import ops
from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
from anomalyclip_amd.components.clip_vit import VisionTransformer
from anomalyclip_amd.components.temporal_model import TemporalModel
from oracle import anomalyclip_oracle as O
import recipes as R

DEV = "cuda"
TOL = 1e-4      # 10x tighter than the 1e-3 north-star tolerance


def relerr(a, b):
    """max |a - b| / max |b| (a NORM-wise relative error, not element-wise: near-zero elements are judged against the
    tensor's scale); the tolerances quoted in the tests are for this quantity."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def elem_ok(a, b):
    """north_star's tolerance, element-wise: |a - b| <= 1e-3 |b| + 1e-5 max|b| for EVERY element (recipes.elem_excess)."""
    return R.elem_excess(a, b) <= 1.0


def make_vit(geom, seed, precision="f32"):
    vit = VisionTransformer(geom.image_resolution, geom.vision_patch_size, geom.vision_width, geom.vision_layers,
                            geom.vision_heads, geom.embed_dim, precision=precision)
    sd = IW.init_vit_state_dict(geom, seed, prefix="")
    vit.load_state_dict(sd, strict=True)
    return vit.to(DEV), sd


def test_vit_tiny_golden(golden):
    g = golden("vit_tiny")
    vit, _ = make_vit(IW.TINY, int(g["seed"]))
    out = vit(torch.from_numpy(g["frames"]).to(DEV))
    assert relerr(out, g["out"]) < TOL and elem_ok(out, g["out"])


@pytest.mark.parametrize("which", ["vit_tiny", "vit_b16"])
def test_vit_f32x6_small_launches_vs_reference(golden, which):
    """precision 'f32x6' with the persistent kernel's minimum size lowered to one tile, so that even the 2-frame golden launches
    take the bf16 x 6 route: every large-GEMM site of the layer loop (LayerNorm -> planes, the attention writing planes (L = 197)
    or the split pass (L = 5), c_fc's epilogue writing planes, the K | V product of the CLS-only last layer), ragged row tiles
    (10 / 394 rows), against the REFERENCE's output under the f32 path's bounds."""
    from anomalyclip_amd import _lib as L
    g = golden(which)
    geom = IW.TINY if which == "vit_tiny" else IW.VIT_B16
    vit, _ = make_vit(geom, int(g["seed"]), precision="f32x6")
    frames = torch.from_numpy(g["frames"]) if which == "vit_tiny" else R.vit_frames(int(g["seed"]), 2, 224)
    h = L.ctx(torch.cuda.current_device())
    ops.set_x6_min_tiles(torch.cuda.current_device(), 1)
    try:
        out = vit(frames.to(DEV))
    finally:
        ops.set_x6_min_tiles(torch.cuda.current_device(), ops.X6_MIN_TILES_DEFAULT)
    assert relerr(out, g["out"]) < TOL and elem_ok(out, g["out"])
    vit32, _ = make_vit(geom, int(g["seed"]))
    assert relerr(out, vit32(frames.to(DEV))) < 5e-6                     # and round-off away from the f32 MFMA path


def test_vit_f32x6_ragged_launch_equals_f32_path(golden):
    """230 frames in one launch (45 310 rows: not a multiple of the 256-row tile, just above the persistent kernel's minimum for
    the N = 768 products): the f32x6 mode agrees with the f32 MFMA path to round-off on every frame."""
    g = golden("vit_b16")
    frames = torch.randn(230, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    frames[:2] = R.vit_frames(int(g["seed"]), 2, 224)
    vit6, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="f32x6")
    vit32, _ = make_vit(IW.VIT_B16, int(g["seed"]))
    vit6.chunk = vit32.chunk = 230
    o6, o32 = vit6(frames.to(DEV)), vit32(frames.to(DEV))
    assert torch.isfinite(o6).all() and relerr(o6, o32) < 5e-6 and elem_ok(o6, o32)
    assert relerr(o6[:2], g["out"]) < TOL


def test_vit_b16_golden(golden):
    g = golden("vit_b16")
    vit, sd = make_vit(IW.VIT_B16, int(g["seed"]))
    frames = R.vit_frames(int(g["seed"]), 2, 224)
    out = vit(frames.to(DEV))
    assert relerr(out, g["out"]) < TOL and elem_ok(out, g["out"])      # every one of the 2 x 512 features, element-wise
    # more frames than one chunk / ragged chunking gives the same rows.  Rows inside ONE launch are bit-identical
    # wherever they sit; across chunk sizes the library may pick a different K split for the GEMMs (64x64 tiles,
    # split-K for skinny problems and for the tail round of tiles), i.e. a different summation order: round-off only.
    vit.chunk = 3
    f5 = torch.cat([frames, frames.flip(0), frames[:1]], 0)
    out5 = vit(f5.to(DEV))
    assert torch.equal(out5[1], out5[2])            # same frame twice inside the 3-frame launch
    assert torch.equal(out5[4], out[0])             # 2-frame launches: slot 1 of the second chunk == slot 0 of `out`
    assert relerr(out5[:2], out) < 2e-6             # 3-frame vs 2-frame launch: summation order only


@pytest.mark.parametrize("precision", ["f32", "bf16", "auto"])
def test_vit_b16_full_clip_properties(golden, precision):
    """BASELINE.json's full size (512 frames in ONE launch: 8-wave kernels, the 64x64-tile tail launch, CLS-only last
    layer, 256x256 ring kernels in bf16 mode) through size-independent properties: identical frames give bit-identical
    rows wherever they sit in the launch, and the rows agree with the golden-pinned 2-frame launch to round-off."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision=precision)
    vit.chunk = 512
    base = R.vit_frames(int(g["seed"]), 2, 224)
    extra = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    eight = torch.cat([base, extra], 0)
    idx = torch.arange(512) % 8
    idx[500:] = torch.tensor([7, 3, 0, 1, 5, 5, 2, 6, 4, 0, 1, 7])          # break the period near the tail rows
    out = vit(eight[idx].to(DEV))
    assert out.shape == (512, 512) and torch.isfinite(out).all()
    for k in range(8):
        rows = out[idx == k]
        if precision in ("f32", "auto"):
            assert torch.equal(rows, rows[:1].expand_as(rows)), k              # bit-identical across ALL slots
        else:
            # bf16 mode: the tail round of tiles runs through another kernel (other f32 summation order), and a
            # last-bit difference before a bf16 rounding is a bf16 ulp after it -- bit-identical inside the main
            # launch, bf16 round-off for the last frames
            head = out[:480][idx[:480] == k]
            assert torch.equal(head, head[:1].expand_as(head)), k
            assert relerr(rows, rows[:1].expand_as(rows)) < 3e-2, k
    small = vit(base.to(DEV))                                                  # 2-frame launch (other kernels)
    assert relerr(out[:2], small) < (3e-2 if precision == "bf16" else 2e-6)
    if precision != "bf16":
        # "auto" (the default: the large GEMMs as f32-accurate bf16 x 6 products; the 2-frame launch above ran the f32 kernels: too few rows
        # for the persistent kernel) meets the SAME bounds against the reference's output as the f32 MFMA path
        assert relerr(out[:2], g["out"]) < TOL and elem_ok(out[:2], g["out"])


@pytest.mark.parametrize("chunk", [32, 64, 160, 256])
def test_vit_b16_launch_sizes_auto(golden, chunk):
    """The default precision at the launch sizes the reference's callers produce (anomaly_clip.py:119-123,158-161: whatever
    b * ncrops * N * S * L the video gives): 32 / 64 / 160 (a 5-crop XD window) / 256 (BASELINE configs[2]'s literal batch) frames per
    launch all take the bf16 x 6 route (round 5: only >= 222 frames did) -- few-tile launches split K for EVERY tile alike, a
    partly filled last round of tiles goes out as column strips with unchanged K order -- so identical frames give bit-identical
    rows wherever they sit in the launch, and the rows meet the golden bounds against the REFERENCE's output."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="auto")
    vit.chunk = chunk
    base = R.vit_frames(int(g["seed"]), 2, 224)
    extra = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    eight = torch.cat([base, extra], 0)
    idx = torch.arange(chunk) % 8
    idx[chunk - 12:] = torch.tensor([7, 3, 0, 1, 5, 5, 2, 6, 4, 0, 1, 7])    # break the period near the tail rows
    out = vit(eight[idx].to(DEV))
    assert out.shape == (chunk, 512) and torch.isfinite(out).all()
    for k in range(8):
        rows = out[idx == k]
        assert torch.equal(rows, rows[:1].expand_as(rows)), k                  # bit-identical across ALL slots
    first = [int((idx == k).nonzero()[0]) for k in (0, 1)]
    assert relerr(out[first], g["out"]) < TOL and elem_ok(out[first], g["out"])
    vit32, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="f32")
    vit32.chunk = chunk
    o32 = vit32(eight[idx].to(DEV))
    assert relerr(out, o32) < 5e-6 and elem_ok(out, o32)                       # and round-off away from the f32 MFMA path, every frame


@pytest.mark.parametrize("chunk", [64, 256])
def test_vit_b16_three_product_mode(golden, chunk):
    """precision "bf16x3" (opt-in; ACX_PREC_F32X3: the plane kernels with the three leading cross products, sixteen significant bits
    per operand -- NOT the f32-accurate default): documents its distance from the REFERENCE's output (within 5e-5 of the largest
    feature: two orders of magnitude inside BASELINE.json's 1e-3, two orders below the bf16 mode) and from the f32 MFMA path, and
    keeps the default's position independence (identical frames -> bit-identical rows wherever they sit in the launch)."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="bf16x3")
    vit.chunk = chunk
    base = R.vit_frames(int(g["seed"]), 2, 224)
    extra = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    eight = torch.cat([base, extra], 0)
    idx = torch.arange(chunk) % 8
    idx[chunk - 12:] = torch.tensor([7, 3, 0, 1, 5, 5, 2, 6, 4, 0, 1, 7])
    out = vit(eight[idx].to(DEV))
    assert out.shape == (chunk, 512) and torch.isfinite(out).all()
    for k in range(8):
        rows = out[idx == k]
        assert torch.equal(rows, rows[:1].expand_as(rows)), k
    first = [int((idx == k).nonzero()[0]) for k in (0, 1)]
    e = relerr(out[first], g["out"])
    print("bf16x3 ViT-B/16 rel err vs reference:", e)
    assert e < 5e-5
    vit32, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="f32")
    vit32.chunk = chunk
    assert relerr(out, vit32(eight[idx].to(DEV))) < 5e-5


@pytest.mark.parametrize("chunk", [64, 256])
def test_vit_b16_f16x3_mode(golden, chunk):
    """precision "f16x3" (opt-in; ACX_PREC_F16X3: two fp16 planes per operand, three exact products) holds the DEFAULT's bounds: the
    golden bounds against the REFERENCE's output (relerr < TOL, element-wise elem_ok), round-off distance to the f32 MFMA path on every
    frame, identical frames -> bit-identical rows wherever they sit in the launch."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="f16x3")
    vit.chunk = chunk
    base = R.vit_frames(int(g["seed"]), 2, 224)
    extra = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    eight = torch.cat([base, extra], 0)
    idx = torch.arange(chunk) % 8
    idx[chunk - 12:] = torch.tensor([7, 3, 0, 1, 5, 5, 2, 6, 4, 0, 1, 7])
    out = vit(eight[idx].to(DEV))
    assert out.shape == (chunk, 512) and torch.isfinite(out).all()
    for k in range(8):
        rows = out[idx == k]
        assert torch.equal(rows, rows[:1].expand_as(rows)), k
    first = [int((idx == k).nonzero()[0]) for k in (0, 1)]
    print("f16x3 ViT-B/16 rel err vs reference:", relerr(out[first], g["out"]))
    assert relerr(out[first], g["out"]) < TOL and elem_ok(out[first], g["out"])
    vit32, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="f32")
    vit32.chunk = chunk
    o32 = vit32(eight[idx].to(DEV))
    assert relerr(out, o32) < 5e-6 and elem_ok(out, o32)


def test_model_precision_bf16x3_end_to_end(prompts_table):
    """AnomalyCLIP(precision = "bf16x3") from FRAMES at the ViT-B/16 geometry (512 frames = one test tile, UCF head): the ViT runs the
    three-product plane kernels, the head the default's arithmetic; similarity logits and scores stay within 1e-4 (relative to the
    largest value) of the default precision's on the same weights and frames -- an order of magnitude inside BASELINE.json's 1e-3."""
    hc = IW.UCF_HEAD
    frames = torch.randn(1, 512, 3, 224, 224, generator=torch.Generator().manual_seed(21)) * 0.7
    nc = torch.randn(512, generator=torch.Generator().manual_seed(22)) * 0.05
    out = {}
    for precision in ("auto", "bf16x3", "f16x3"):
        net, sd, eot = build_net("ViT-B/16", hc, "ucf", 31, prompts_table, precision=precision)
        net.load_from_features = False
        net.eval()
        assert net.image_encoder.precision == precision and net.temporal_model.precision == "auto"
        with torch.no_grad():
            out[precision] = tuple(t.clone() for t in net(frames.to(DEV), None, nc.to(DEV), 1, True))
        del net
    (s6, c6), (s3, c3), (sf, cf) = out["auto"], out["bf16x3"], out["f16x3"]
    assert torch.isfinite(s3).all() and torch.isfinite(c3).all() and torch.isfinite(sf).all() and torch.isfinite(cf).all()
    print("bf16x3 vs auto: similarity", relerr(s3, s6), "scores", relerr(c3, c6), "| f16x3 vs auto:", relerr(sf, s6), relerr(cf, c6))
    assert relerr(s3, s6) < 1e-4 and relerr(c3, c6) < 1e-4
    assert relerr(sf, s6) < 5e-6 and relerr(cf, c6) < 5e-6              # f16x3: round-off away from the default
    assert not torch.equal(s3, s6) and not torch.equal(sf, s6)            # (the modes are really in use)


def test_vit_two_streams_equals_sequential_half_launches(golden):
    """VisionTransformer(streams = 2) (opt-in): the chunk as two half chunks on two side streams with their own workspaces -- the
    features are bit for bit those of the same half chunks launched one after the other, for an even and a ragged frame count, and
    a second call (workspaces and streams reused) reproduces them."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="auto")
    frames = torch.randn(150, 3, 224, 224, generator=torch.Generator().manual_seed(11)).to(DEV)
    vit.chunk = 64
    ref = vit(frames)                               # launches of 64, 64, 22 frames
    vit.chunk = 128
    vit.streams = 2                                 # halves of 64: the same launches, pairwise concurrent
    out = vit(frames)
    assert torch.equal(out, ref)
    assert torch.equal(vit(frames), ref)
    vit.streams = 1


def test_vit_b16_bf16_mode(golden):
    """bf16 MFMA mode: NOT the parity path; documents its distance from the f32 reference."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="bf16")
    out = vit(R.vit_frames(int(g["seed"]), 2, 224).to(DEV))
    e = relerr(out, g["out"])
    print("bf16 ViT-B/16 rel err vs reference:", e)
    assert e < 5e-2


def build_net(geom_name, hc, key, seed, prompts_table, **kw):
    toks = torch.tensor(prompts_table[key]["tokenized_prompts"], dtype=torch.int32)
    geom = IW.TINY if geom_name == "tiny" else IW.VIT_B16
    net = AnomalyCLIP(arch=geom_name if geom_name == "tiny" else "ViT-B/16", labels_key=key, emb_size=hc.emb_size,
                      depth=hc.depth, heads=hc.heads, dim_heads=hc.dim_heads, num_segments=32, seg_length=16,
                      concat_features=hc.concat_features, normal_id=hc.normal_id, stride=1, load_from_features=True,
                      select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=hc.ncrops, num_topk=3,
                      num_bottomk=3, n_ctx=8, shared_context=False, ctx_init="", **kw)
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, seed)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return net.to(DEV), sd, toks.argmax(-1)


@pytest.mark.parametrize("tag,geom_name,key", [("text_tiny", "tiny", "ucf"), ("text_b16_xd", "ViT-B/16", "xd")])
def test_text_golden(golden, prompts_table, tag, geom_name, key):
    g = golden(tag)
    C = len(prompts_table[key]["classnames"])
    hc = IW.HeadConfig(num_classes=C, normal_id=prompts_table[key]["normal_id"])
    geom = IW.TINY if geom_name == "tiny" else IW.VIT_B16
    toks = torch.tensor(prompts_table[key]["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch=geom_name, labels_key=key, emb_size=256, depth=1, heads=8, dim_heads=None, num_segments=32,
                      seg_length=16, concat_features=False, normal_id=hc.normal_id, select_idx_dropout_topk=0.7,
                      select_idx_dropout_bottomk=0.7, num_topk=3, num_bottomk=3)
    sd = IW.init_anomalyclip_state_dict(geom, hc, toks, int(g["seed"]), with_image_encoder=False)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("image_encoder.", "temporal_model.")) for k in missing)
    net = net.to(DEV)
    with torch.no_grad():
        tf = net.get_text_features()
    assert relerr(tf, g["out"]) < TOL and elem_ok(tf, g["out"])
    # reference-shaped API: PromptLearner() then TextEncoder(prompts, tokenized_prompts)
    with torch.no_grad():
        tf2 = net.text_encoder(net.prompt_learner(), net.tokenized_prompts)
    assert relerr(tf2, g["out"]) < TOL and elem_ok(tf2, g["out"])


def test_temporal_golden(golden):
    """a6 tilings pinned by the reference; a7 PARITY UNPINNED (restated dependency)."""
    g = golden("temporal")
    hc = IW.HeadConfig(emb_size=64, heads=2, depth=2)
    in_size = int(g["in_size"])
    tm = TemporalModel(in_size, 64, 1, 2, None, 2, 32, 16)
    tm.load_state_dict(IW.init_temporal_state_dict(in_size, hc, int(g["seed"]), prefix=""), strict=True)
    tm = tm.to(DEV)
    Kp = tm.prepared()["Kp"]
    with torch.no_grad():
        for S in (1, 2, 3):
            f = torch.from_numpy(g[f"feats_S{S}"])
            fp = torch.cat([f, f.new_zeros(f.shape[0], Kp - in_size)], 1).to(DEV)
            out = tm(fp, S, True)
            assert relerr(out, g[f"scores_test_S{S}"]) < TOL and elem_ok(out, g[f"scores_test_S{S}"])
        f = torch.from_numpy(g["feats_train"])
        fp = torch.cat([f, f.new_zeros(f.shape[0], Kp - in_size)], 1).to(DEV)
        out = tm(fp, 1, False)
        assert relerr(out, g["scores_train"]) < TOL and elem_ok(out, g["scores_train"])


def test_e2e_tiny_golden_test_mode(golden, prompts_table):
    g = golden("e2e_tiny")
    seed = int(g["seed"])
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    net, sd, eot = build_net("tiny", hc, "ucf", seed, prompts_table)
    inp = R.e2e_inputs(seed, IW.TINY.embed_dim)
    with torch.no_grad():
        sim, sc = net(inp["test_feats"].to(DEV), torch.zeros(1000), inp["nc"], 2, True)
    assert relerr(sim, g["test_sim"]) < TOL and relerr(sc, g["test_scores"]) < TOL
    assert elem_ok(sim, g["test_sim"]) and elem_ok(sc, g["test_scores"])
    net.load_from_features = False
    with torch.no_grad():
        sim, sc = net(inp["frames"].to(DEV), torch.zeros(500), inp["nc"], 1, True)
    assert relerr(sim, g["test_frames_sim"]) < TOL and relerr(sc, g["test_frames_scores"]) < TOL
    assert elem_ok(sim, g["test_frames_sim"]) and elem_ok(sc, g["test_frames_scores"])


@pytest.mark.parametrize("cfg", ["ucf", "sht", "xd"])
def test_head_vs_oracle_full_configs(prompts_table, cfg):
    """Full-size head configurations (UCF / ShanghaiTech concat+depth2 / XD E=128, 5 crops)
    against the oracle on seeded features, test mode with S=2."""
    hc = {"ucf": IW.UCF_HEAD, "sht": IW.SHT_HEAD, "xd": IW.XD_HEAD}[cfg]
    net, sd, eot = build_net("ViT-B/16", hc, cfg, 7, prompts_table)
    g = torch.Generator().manual_seed(1)
    S = 2
    feats = torch.randn(1, hc.ncrops, 512 * S, 512, generator=g) * 0.3
    nc = torch.randn(512, generator=g) * 0.05
    with torch.no_grad():
        sim, sc = net(feats.to(DEV), None, nc, S, True)
        rs, rc = O.anomaly_clip_forward_test(sd, hc, feats, nc, eot, 8, S)
    assert relerr(sim, rs) < TOL and relerr(sc, rc) < TOL
    assert elem_ok(sim, rs) and elem_ok(sc, rc)


@pytest.mark.parametrize("hw", [(240, 320), (480, 360), (224, 224), (120, 160), (720, 1280)])
def test_frame_preprocessing_matches_pil(hw):
    """row f2: uint8 frames -> CLIP input; the 8-bit resample stages are bit-exact with Pillow, the float tail
    (/255, normalise) within one ulp-class tolerance."""
    from anomalyclip_amd.preprocess import preprocess_frames
    g = torch.Generator().manual_seed(hw[0])
    frames = torch.randint(0, 256, (3, hw[0], hw[1], 3), generator=g, dtype=torch.uint8)
    frames[0, : hw[0] // 2] = 255           # saturated / flat regions exercise the clip8 path
    frames[1, :, : hw[1] // 3] = 0
    ref = O.preprocess_frames_ref(frames.numpy())
    out = preprocess_frames(frames.to(DEV))
    assert out.shape == (3, 3, 224, 224)
    assert (out.cpu() - ref).abs().max().item() < 2e-6
    # recover the uint8 image exactly: v = round((out*std + mean) * 255)
    from anomalyclip_amd.preprocess import CLIP_MEAN, CLIP_STD
    m, s = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    assert torch.equal(((out.cpu() * s + m) * 255).round(), ((ref * s + m) * 255).round())


def test_shared_context_and_stride_and_long_segments(prompts_table):
    """edge configurations the reference supports: shared (2-D) CoOp context (coop.py:37-39,76-77), stride 2
    (repeat_interleave, anomaly_clip.py:149-150) and a long video (S = 16 tiles, XD head)."""
    import dataclasses
    hc = dataclasses.replace(IW.XD_HEAD, shared_context=True, stride=2, ncrops=1)
    toks = torch.tensor(prompts_table["xd"]["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="xd", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads, dim_heads=None,
                      num_segments=32, seg_length=16, concat_features=False, normal_id=hc.normal_id, stride=2,
                      load_from_features=True, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=1,
                      num_topk=3, num_bottomk=3, n_ctx=8, shared_context=True)
    sd = IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, 3)
    assert sd["prompt_learner.ctx"].dim() == 2
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    g = torch.Generator().manual_seed(2)
    S = 16
    feats = torch.randn(1, 1, 512 * S, 512, generator=g) * 0.3
    nc = torch.randn(512, generator=g) * 0.05
    with torch.no_grad():
        sim, sc = net(feats.to(DEV), None, nc, S, True)
        rs, rc = O.anomaly_clip_forward_test(sd, hc, feats, nc, toks.argmax(-1), 8, S)
    assert sim.shape == (512 * S * 2, 6) and sc.shape == (512 * S * 2,)
    assert relerr(sim, rs) < TOL and relerr(sc, rc) < TOL
    assert elem_ok(sim, rs) and elem_ok(sc, rc)


def test_frames_path_with_crops_tiny(prompts_table):
    """test mode from FRAMES with ncrops = 2 and S = 2: the "(b ncrops n s l) d" view of anomaly_clip.py:124-131."""
    import dataclasses
    hc = dataclasses.replace(IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1), ncrops=2)
    net, sd, eot = build_net("tiny", hc, "ucf", 9, prompts_table)
    net.load_from_features = False
    g = torch.Generator().manual_seed(3)
    frames = torch.randn(1, 2 * 512 * 2, 3, 32, 32, generator=g)        # b=1, ncrops*n*s*l frames
    nc = torch.randn(128, generator=g) * 0.1
    with torch.no_grad():
        sim, sc = net(frames.to(DEV), None, nc, 2, True)
        rs, rc = O.anomaly_clip_forward_test(sd, hc, None, nc, eot, 2, 2, frames=frames)
    assert relerr(sim, rs) < TOL and relerr(sc, rc) < TOL
    assert elem_ok(sim, rs) and elem_ok(sc, rc)


def test_ncentroid_and_module_test_step(prompts_table):
    """a10 (mean of normal features) and a11 (class probabilities, padded frames stripped) through the module mirror."""
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    net, sd, eot = build_net("tiny", hc, "ucf", 4, prompts_table)
    mod = AnomalyCLIPModule(net, None, None, None, num_classes=14, solver={"lr": 1e-5}).to(DEV)
    g = torch.Generator().manual_seed(6)
    vids = [torch.randn(1, 1, 512 * s, 128, generator=g) * 0.3 + 0.05 for s in (1, 2, 1)]
    lens = [300, 1000, 512]
    loader = [(v, torch.zeros(1, n), 7, s) for v, n, s in zip(vids, lens, (1, 2, 1))]
    nc = mod.compute_ncentroid(loader)
    ref = O.ncentroid_from_features([v.reshape(-1, 128)[:n] for v, n in zip(vids, lens)])
    assert relerr(nc, ref) < 1e-5
    out = mod.test_step((vids[1], torch.zeros(1, 1000), 7, 2, "x"))
    rs, rc = O.anomaly_clip_forward_test(sd, hc, vids[1], ref, eot, 2, 2)
    cp, sc = O.eval_postprocess(rs, rc, 1000)
    assert out["class_probs"].shape == (1000, 13) and out["abnormal_scores"].shape == (1000,)
    assert relerr(out["class_probs"], cp) < TOL and relerr(out["abnormal_scores"], sc) < TOL
    assert elem_ok(out["class_probs"], cp) and elem_ok(out["abnormal_scores"], sc)


def test_feature_stream_matches_reference_loop(tmp_path, prompts_table):
    """row f1: .npy files -> HBM tiles (pinned + async copy) equal the reference's per-frame loop, and scoring a
    stream of videos equals scoring them one by one."""
    from anomalyclip_amd.feature_stream import FeatureStream
    rng = np.random.default_rng(0)
    paths, raws = [], []
    for i, T_ in enumerate((300, 513, 1000, 100)):            # 100 frames: the 512-row tile wraps around five times
        a = (rng.standard_normal((T_, 128)) * 0.3).astype(np.float32)
        p = str(tmp_path / f"v{i}.npy")
        np.save(p, a)
        paths.append(p)
        raws.append(a)
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    net, sd, eot = build_net("tiny", hc, "ucf", 2, prompts_table)
    nc = torch.zeros(128)
    for (feats, T_, S, path), raw in zip(FeatureStream(paths, device=torch.device(DEV)), raws):
        # the reference's loop (feature_dataset.py:359-367), literally
        starts = np.arange(np.ceil(T_ / 512) * 512 / 16) * 16
        ref = np.stack([raw[(int(s) + i) % T_] for s in starts for i in range(16)])
        assert S == len(starts) // 32 and feats.shape == (1, 1, 512 * S, 128)
        assert np.array_equal(feats[0, 0].cpu().numpy(), ref)
        with torch.no_grad():
            sim, sc = net(feats, None, nc, S, True)
            rs, rc = O.anomaly_clip_forward_test(sd, hc, torch.from_numpy(ref).view(1, 1, -1, 128), nc, eot, 2, S)
        assert relerr(sc[:T_], rc[:T_]) < TOL and elem_ok(sc[:T_], rc[:T_])


@pytest.mark.parametrize("ncrops,stride,T_", [(5, 1, 700), (5, 1, 90), (1, 2, 700), (5, 2, 1500)])
def test_feature_stream_crops_and_stride(tmp_path, ncrops, stride, T_):
    """row f1, the other loader branches: multi-crop files (XD-Violence: five crops interleaved per frame) through the
    per-crop strided copy, strided sampling through the index gather -- both equal feature_index.gather_test_features,
    the function pinned to the reference's tables (feature_dataset.py:347-376)."""
    from anomalyclip_amd.feature_stream import FeatureStream
    from anomalyclip_amd import feature_index as FI
    rng = np.random.default_rng(T_ + ncrops)
    paths, raws = [], []
    for i in range(3):
        a = rng.standard_normal(((T_ + 37 * i) * ncrops, 64)).astype(np.float32)
        p = str(tmp_path / f"c{i}.npy")
        np.save(p, a)
        paths.append(p)
        raws.append(a)
    fs = FeatureStream(paths, stride=stride, ncrops=ncrops, device=torch.device(DEV))
    for (feats, T, S, path), raw in zip(fs, raws):
        ref, S_ref = FI.gather_test_features(raw, 32, 16, stride, ncrops)
        assert T == raw.shape[0] // ncrops and S == S_ref and feats.shape == (1,) + ref.shape
        assert np.array_equal(feats[0].cpu().numpy(), ref)


def test_config0_shanghaitech_eval_from_feature_files(golden, prompts_table, tmp_path):
    """BASELINE.json configs[0] on the HIP path, end to end: eight `.npy` feature files (T = 300 ... 5000, up to S = 10
    tiles) -> FeatureStream (one gather into pinned memory, async copy) -> AnomalyCLIPModule.test_step (ShanghaiTech head:
    18 classes, depth 2, concat on; text tower, selector, tiled axial transformer, classifier, class_probs, truncation to
    the real frames) -> test_epoch_end, against the outputs the REFERENCE produced for the same files (config0.npz) --
    element-wise within north_star's 1e-3 -- and the metrics epilogue against the metrics oracle on the reference's
    numbers."""
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.feature_stream import FeatureStream
    g = golden("config0")
    paths, arrays, labels, nc = R.config0_feature_files(tmp_path)
    hc = IW.SHT_HEAD
    toks = torch.tensor(prompts_table["sht"]["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="sht", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads,
                      dim_heads=None, num_segments=32, seg_length=16, concat_features=True, normal_id=hc.normal_id,
                      stride=1, load_from_features=True, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7,
                      ncrops=1, num_topk=3, num_bottomk=3, n_ctx=8, shared_context=False, ctx_init="")
    sd = IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, int(g["seed"]), with_image_encoder=False)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("image_encoder.") for k in missing)
    net = net.to(DEV).eval()
    mod = AnomalyCLIPModule(net, None, None, None, num_classes=18, solver={"lr": 1e-5}, logs_root=str(tmp_path / "logs"))
    mod.ncentroid = nc.to(DEV)
    outs = []
    for i, (feats, T_, S, path) in enumerate(FeatureStream(paths, device=torch.device(DEV))):
        assert T_ == R.CONFIG0_LENGTHS[i] and S == int(g[f"S{i}"]) and feats.shape == (1, 1, 512 * S, 512)
        out = mod.test_step((feats, torch.from_numpy(labels[i]).unsqueeze(0), int(labels[i].min()), S, path), i)
        assert out["abnormal_scores"].shape == (T_,) and out["class_probs"].shape == (T_, 17)      # padded frames stripped
        assert relerr(out["abnormal_scores"], g[f"scores{i}"]) < TOL and elem_ok(out["abnormal_scores"], g[f"scores{i}"])
        assert elem_ok(out["class_probs"][::8], g[f"probs8_{i}"])
        ps = float(out["class_probs"].double().sum())
        assert abs(ps - float(g[f"probsum{i}"])) < 1e-5 * abs(float(g[f"probsum{i}"]))
        outs.append(out)
    # ---- the same eight videos BATCHED: FeatureStream.batched (groups of consecutive videos, one H2D copy each) ->
    # test_step_many -> AnomalyCLIP.forward_test_many (ONE selector / temporal launch sequence over all tiles of a group, a
    # per-tile gather table instead of one segment size, text features once).  Against the reference-produced fixture with
    # the same bounds as above; against the one-video path to summation-order round-off (NOT bit-identical: the library
    # picks its GEMM kernels -- few-row, 64x64 tiles, split-K -- by the launch's row count, hence a different f32 summation
    # order); and position-independent: a video gives bit-identical rows wherever it sits inside a group.
    fs = FeatureStream(paths, device=torch.device(DEV))
    bouts = []
    for feats, meta in fs.batched(videos=3, max_tiles=16):
        i0 = len(bouts)
        batches, r0 = [], 0
        for k, (T_, S, rows, path) in enumerate(meta):
            i = i0 + k
            assert T_ == R.CONFIG0_LENGTHS[i] and S == int(g[f"S{i}"]) and rows == 512 * S
            batches.append((feats[r0:r0 + rows].view(1, 1, rows, 512), torch.from_numpy(labels[i]).unsqueeze(0),
                            int(labels[i].min()), S, path))
            r0 += rows
        assert r0 == feats.shape[0]
        bouts.extend(mod.test_step_many(batches, i0))
    assert len(bouts) == 8
    for i, (a_, b_) in enumerate(zip(bouts, outs)):
        assert a_["abnormal_scores"].shape == b_["abnormal_scores"].shape and a_["class_probs"].shape == b_["class_probs"].shape
        assert relerr(a_["abnormal_scores"], g[f"scores{i}"]) < TOL and elem_ok(a_["abnormal_scores"], g[f"scores{i}"])
        assert elem_ok(a_["class_probs"][::8], g[f"probs8_{i}"])
        assert relerr(a_["abnormal_scores"], b_["abnormal_scores"]) < 5e-6 and relerr(a_["class_probs"], b_["class_probs"]) < 5e-6
        assert torch.equal(a_["labels"], b_["labels"])
    mk = lambda i: (torch.from_numpy(np.ascontiguousarray(np.resize(arrays[i], (512 * int(g[f"S{i}"]), 512)))).view(1, 1, -1, 512).to(DEV),
                    torch.from_numpy(labels[i]).unsqueeze(0), int(labels[i].min()), int(g[f"S{i}"]), paths[i])
    twice = mod.test_step_many([mk(2), mk(0), mk(2)], 0)            # video 2 (S = 2) at two positions of one group
    assert torch.equal(twice[0]["abnormal_scores"], twice[2]["abnormal_scores"])
    assert torch.equal(twice[0]["class_probs"], twice[2]["class_probs"])
    m = mod.test_epoch_end(outs)
    mb = mod.test_epoch_end(bouts)
    assert abs(mb["auc_roc"] - m["auc_roc"]) < 1e-4
    from oracle import metrics_oracle as MO
    ref_scores = np.concatenate([g[f"scores{i}"] for i in range(8)])
    ref_bin = (np.concatenate(labels) != hc.normal_id).astype(np.int64)
    # scores agree to ~1e-6, so a few near-tied pairs may swap rank between the two sets: AUROC within 1e-3
    assert abs(m["auc_roc"] - MO.binary_auroc(ref_scores, ref_bin)) < 1e-3
    assert (tmp_path / "logs" / "eval" / "runs" / "ckpt" / "metrics.json").is_file()


def test_lightning_checkpoint_load_then_score(tmp_path, prompts_table):
    """f4 on the device: a Lightning-shaped `last.ckpt` (`state_dict` keyed `net.*`, torchmetrics state, CLIP towers
    exported in half precision) + the `ncentroid.pt` side-car -> checkpoint.load_into a fresh module -> test-mode
    scores equal the oracle evaluated on the SAME (half-rounded) weights."""
    from anomalyclip_amd import checkpoint
    hc = IW.HeadConfig(num_classes=14, normal_id=7, emb_size=64, heads=2, depth=1)
    src, sd, eot = build_net("tiny", hc, "ucf", 31, prompts_table)
    lsd = checkpoint.to_lightning_state_dict(src)
    for k in list(lsd):
        if k.startswith(("net.image_encoder.", "net.text_encoder.transformer.")):
            lsd[k] = lsd[k].half()
    lsd["train_loss.mean_value"] = torch.zeros(1)
    lsd["auroc.confmat"] = torch.zeros(2, 2)
    path = tmp_path / "logs" / "last.ckpt"
    path.parent.mkdir(parents=True)
    torch.save({"state_dict": lsd, "epoch": 7, "global_step": 123, "hyper_parameters": {"num_classes": 14}}, str(path))
    nc = torch.randn(IW.TINY.embed_dim, generator=torch.Generator().manual_seed(2)) * 0.1
    torch.save(nc, str(path.parent / "ncentroid.pt"))

    dst, _, _ = build_net("tiny", hc, "ucf", 99, prompts_table)          # other weights: everything must come from the file
    missing, unexpected = checkpoint.load_into(dst, str(path))
    assert not missing and not unexpected
    ncl = torch.load(str(path.parent / "ncentroid.pt"))
    feats = torch.randn(1, 1, 2 * 512, IW.TINY.embed_dim, generator=torch.Generator().manual_seed(3)) * 0.3
    dst.eval()
    with torch.no_grad():
        sim, sc = dst(feats.to(DEV), None, ncl, 2, True)
    sd_half = {k[4:]: (v.float() if v.is_floating_point() else v) for k, v in lsd.items() if k.startswith("net.")}
    rsim, rsc = O.anomaly_clip_forward_test(sd_half, hc, feats, ncl, eot, IW.TINY.transformer_heads, 2)
    assert relerr(sim, rsim) < TOL and relerr(sc, rsc) < TOL
    assert elem_ok(sim, rsim) and elem_ok(sc, rsc)


@pytest.mark.parametrize("S", [4, 8, 16])
def test_xd_long_segments_bf16_head(prompts_table, S):
    """BASELINE.json configs[4]: XD-Violence head (C = 7, E = 128 -> head dim 16, 5 crops) on long test videos
    (S = 4 / 8 / 16 tiles per crop -> 10 240 .. 40 960 rows), head GEMMs and implicit-GEMM convolutions on the bf16 MFMA
    (bf16 weights / LayerNorm outputs / conv hidden activations, f32 accumulation, f32 residual streams).
    NOT the parity path: compared with the f32 ORACLE on the same inputs at a stated bf16 tolerance -- anomaly scores
    (sigmoid outputs in [0,1]) within 2e-2 absolute, similarity logits (bf16 text tower -> f32 selector) within 2e-2
    relative -- while the f32 mode of the same configuration holds the parity tolerance."""
    hc = IW.XD_HEAD
    g = torch.Generator().manual_seed(S)
    feats = torch.randn(1, hc.ncrops, 512 * S, 512, generator=g) * 0.3 + 0.02
    nc = torch.randn(512, generator=g) * 0.05
    out = {}
    for precision in ("bf16", "f32"):
        net, sd, eot = build_net("ViT-B/16", hc, "xd", 17, prompts_table, precision=precision)
        net.eval()
        with torch.no_grad():
            out[precision] = net(feats.to(DEV), None, nc, S, True)
        del net
    rsim, rsc = O.anomaly_clip_forward_test(sd, hc, feats, nc, eot, 8, S)
    sim, sc = out["bf16"]
    assert sim.shape == (hc.ncrops * 512 * S, hc.num_classes - 1) and sc.shape == (hc.ncrops * 512 * S,)
    assert relerr(out["f32"][0], rsim) < TOL and relerr(out["f32"][1], rsc) < TOL     # f32 mode: parity
    err = (sc.double().cpu() - rsc.double()).abs().max().item()
    print(f"bf16 head, S={S}: max |score - oracle| = {err:.3e}, similarity rel = {relerr(sim, rsim):.3e}")
    assert err < 2e-2 and relerr(sim, rsim) < 2e-2


def test_forward_test_many_crops_and_text_cache(prompts_table):
    """AnomalyCLIP.forward_test_many with several crops (XD-Violence: 5) and different segment sizes in one batch against the
    per-video forward (same rows to summation-order round-off, element-wise within north_star's bound), and the evaluation
    text-feature cache: on by default, returns the SAME tensor for consecutive videos, bit-identical to an uncached
    evaluation, recomputed after an optimizer step / an in-place edit of the context."""
    hc = IW.XD_HEAD
    net, sd, eot = build_net("ViT-B/16", hc, "xd", 29, prompts_table)
    net.eval()
    g = torch.Generator().manual_seed(12)
    nc = (torch.randn(512, generator=g) * 0.05).to(DEV)
    vids = [(torch.randn(1, hc.ncrops, 512 * S, 512, generator=g) * 0.3 + 0.02).to(DEV) for S in (2, 1, 3)]
    segs = [2, 1, 3]
    with torch.no_grad():
        singles = [net(v, None, nc, S, True) for v, S in zip(vids, segs)]
        x = torch.cat([v.reshape(-1, 512) for v in vids], 0)
        sim, sc = net.forward_test_many(x, [512 * S for S in segs], segs, nc)
    r0 = 0
    for (rs, rc), S in zip(singles, segs):
        r1 = r0 + hc.ncrops * 512 * S
        assert relerr(sim[r0:r1], rs) < 5e-6 and relerr(sc[r0:r1], rc) < 5e-6
        assert elem_ok(sim[r0:r1], rs) and elem_ok(sc[r0:r1], rc)
        r0 = r1
    assert r0 == sim.shape[0]
    # ---- text-feature cache
    assert net.cache_text_features
    with torch.no_grad():
        t1 = net.get_text_features()
        t2 = net.get_text_features()
        assert t2 is t1
        net.cache_text_features = False
        t3 = net.get_text_features()                 # uncached evaluation: the tower replayed from a HIP graph
        assert net.text_eval_graph and net.__dict__.get("_text_graph") is not None, net.__dict__.get("_text_graph_error")
        graph0 = net.__dict__["_text_graph"][1]
        t3b = net.get_text_features()
        assert net.__dict__["_text_graph"][1] is graph0 and t3b is not t3 and torch.equal(t3b, t3)
        assert torch.equal(t3, net._text_features_eager())
        net.cache_text_features = True
        assert t3 is not t1 and torch.equal(t3, t1)
        from anomalyclip_amd import ops as OPS
        OPS.WEIGHT_EPOCH[0] += 1                     # what an optimizer step through raw pointers does
        t4 = net.get_text_features()
        assert t4 is not t1 and torch.equal(t4, t1)
        net.prompt_learner.ctx.mul_(1.5)             # an in-place edit bumps the tensor version
        t5 = net.get_text_features()
        assert t5 is not t4 and not torch.equal(t5, t4)
        net.cache_text_features = False
        t6 = net.get_text_features()                 # the edit re-captured the graph
        net.cache_text_features = True
        assert net.__dict__["_text_graph"][1] is not graph0 and torch.equal(t6, t5)


def test_vit_bf16_160_frame_window(golden):
    """configs[4], frames variant: a 5-crop x 32-frame window = 160 frames in ONE bf16 ViT launch; rows agree with the
    2-frame launch (bf16 round-off) and identical frames give identical rows."""
    g = golden("vit_b16")
    vit, _ = make_vit(IW.VIT_B16, int(g["seed"]), precision="bf16")
    vit.chunk = 160
    base = R.vit_frames(int(g["seed"]), 2, 224)
    idx = torch.arange(160) % 2
    out = vit(base[idx].to(DEV))
    small = vit(base.to(DEV))
    assert out.shape == (160, 512) and torch.isfinite(out).all()
    assert relerr(out[:2], small) < 3e-2 and relerr(out[158:], small) < 3e-2
    assert relerr(out[:2], g["out"]) < 5e-2                                            # vs the REFERENCE's f32 output
