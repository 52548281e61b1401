"""CPU-only (-m "not gpu") checks: the C ABI library loads and exports every symbol include/acx.h declares,
host-side logic (index tables, LR schedule, mask RNG, sharding, SyncBN statistics combination) and the
no-fallback rule.  No kernel is launched here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(REPO, "include", "acx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(acx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from anomalyclip_amd import _build, _lib
    assert os.path.exists(_build.LIB), "libacx.so missing: run python -m anomalyclip_amd._build"
    lib = ctypes.CDLL(_build.LIB)
    declared = header_functions()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/acx.h but not exported"
    # and the ctypes binding covers the same set
    assert sorted(_lib.declared_symbols()) == declared
    assert _lib.lib().acx_version() == 100


def test_no_kernel_spills_or_uses_scratch():
    """Kernel hygiene of the library AS BUILT (hipcc's kernel-resource-usage remarks, written next to libacx.so by the build): no
    kernel of libacx.so may use scratch memory or spill a register -- a kernel that silently spills runs its hot loop through
    memory (round 1: a demoted staging array cost the f32 GEMM 7 %; round 5: 20 bytes per lane in the bf16 256 x 256 kernel)."""
    from anomalyclip_amd import _build, _lib
    _lib.lib()                                              # builds when stale
    assert os.path.exists(_build.RESOURCES), "resource table missing: python -m anomalyclip_amd._build --force"
    assert os.path.getmtime(_build.RESOURCES) >= os.path.getmtime(_build.LIB) - 5
    rows = _build.kernel_resources()
    assert len(rows) >= 300, len(rows)
    assert any("gemm_x6_p4_kernel" in k for k in rows) and any("gemm_bf16_p8_kernel" in k for k in rows)
    bad = {k: (v.get("ScratchSize [bytes/lane]"), v.get("VGPRs Spill")) for k, v in rows.items()
           if v.get("ScratchSize [bytes/lane]", 0) != 0 or v.get("VGPRs Spill", 0) != 0}
    assert not bad, bad


def test_product_path_has_no_cpu_fallback():
    from anomalyclip_amd import ops, _lib
    with pytest.raises(_lib.AcxError):
        ops.gemm(torch.randn(8, 8), torch.randn(8, 8))
    with pytest.raises(_lib.AcxError):
        ops.layernorm(torch.randn(4, 64), torch.ones(64), torch.zeros(64))
    # nothing under anomalyclip_amd/ may import the oracle
    for root, _, files in os.walk(os.path.join(REPO, "anomalyclip_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    from anomalyclip_amd import _lib
    monkeypatch.setenv("ACX_LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.AcxError):
        _lib.lib()
    monkeypatch.delenv("ACX_LIB_PATH")
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.lib()


def test_index_tables_match_reference(golden):
    from anomalyclip_amd import feature_index as FI
    g = golden("tables")
    for stride in (1, 2):
        for T in (1, 511, 512, 513, 1000, 1025, 5000):
            starts, S = FI.test_start_indices(T, 32, 16, stride)
            assert np.array_equal(starts, g[f"start_T{T}_s{stride}"].astype(np.int64))
            assert np.array_equal(FI.frame_index_table(starts, 16, stride, T), g[f"frames_T{T}_s{stride}"])
    feats = np.arange(1000 * 2 * 4, dtype=np.float32).reshape(1000 * 2, 4)
    out, S = FI.gather_test_features(feats, 32, 16, 1, ncrops=2)
    assert out.shape == (2, 512 * S, 4) and S == 2
    assert np.array_equal(out[1, 5], feats[5 * 2 + 1]) and np.array_equal(out[0, 1000], feats[0])   # wrap-around padding


def test_lr_schedule_matches_reference(golden):
    from anomalyclip_amd.components.scheduler import WarmupCosineAnnealingLR
    g = golden("tables")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-5}], weight_decay=0.2)
    sch = WarmupCosineAnnealingLR(optimizer=opt, successor=None, warmup_epochs=5, total_epoch=50)
    lrs = []
    for _ in range(55):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(lrs, g["lr_table"], rtol=1e-12, atol=0)


def test_mask_rng_matches_reference_draw_order():
    """generate_mask draws from torch's CPU generator exactly like selector_model.py:101-117."""
    from anomalyclip_amd.components.selector_model import SelectorModel
    sel = SelectorModel([str(i) for i in range(14)], 7, 1.0, 32, 16, 0.7, 0.7, 3, 3)
    torch.manual_seed(123)
    a, b = sel.generate_mask(6)
    torch.manual_seed(123)
    ones = torch.ones((6, 32))
    t = torch.bernoulli(ones * (1 - 0.7))
    bt = torch.bernoulli(ones * (1 - 0.7))
    assert torch.equal(b, bt) and torch.equal(a, bt) and a.shape == (6, 32)    # equal dropouts -> shared mask (:114-115)
    sel2 = SelectorModel([str(i) for i in range(14)], 7, 1.0, 32, 16, 0.7, 0.5, 3, 3)
    torch.manual_seed(123)
    a2, b2 = sel2.generate_mask(6)
    assert torch.equal(a2, t) and not torch.equal(a2, b2)


def test_shard_videos_balanced():
    from anomalyclip_amd import parallel
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            idx = parallel.shard_videos(64, world, r)
            half = len(idx) // 2
            assert all(i < 32 for i in idx[:half]) and all(i >= 32 for i in idx[half:])
            seen += idx
        assert sorted(seen) == list(range(64))
    with pytest.raises(ValueError):
        parallel.shard_videos(64, 3, 0)


def test_combine_bn_stats_equals_global_stats():
    from anomalyclip_amd import parallel
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, 13, generator=g, dtype=torch.float64) * 3 + 1
    parts = [x[:100], x[100:640], x[640:]]
    means = torch.stack([p.mean(0) for p in parts])
    m2s = torch.stack([p.var(0, unbiased=False) * p.shape[0] for p in parts])
    counts = torch.tensor([float(p.shape[0]) for p in parts], dtype=torch.float64)
    m, vb, vu, n = parallel.combine_bn_stats(means, m2s, counts)
    assert torch.allclose(m, x.mean(0)) and torch.allclose(vb, x.var(0, unbiased=False)) and torch.allclose(vu, x.var(0, unbiased=True))
    assert int(n) == 1000


def test_state_dict_keys_match_reference_layout(prompts_table):
    """module tree == reference AnomalyCLIP.state_dict() key layout (SURVEY section 5 checkpoint row)."""
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
    for key, hc in (("ucf", IW.UCF_HEAD), ("sht", IW.SHT_HEAD), ("xd", IW.XD_HEAD)):
        toks = torch.tensor(prompts_table[key]["tokenized_prompts"], dtype=torch.int32)
        net = AnomalyCLIP(arch="tiny", labels_key=key, emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads, dim_heads=None,
                          num_segments=32, seg_length=16, concat_features=hc.concat_features, normal_id=hc.normal_id,
                          select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, num_topk=3, num_bottomk=3, ncrops=hc.ncrops)
        sd = IW.init_anomalyclip_state_dict(IW.TINY, hc, toks, 1)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not missing and not unexpected
        assert set(net.state_dict().keys()) == set(sd.keys())
    # trainable parameter count of the UCF temporal model at the real geometry (10,110,977)
    tm = IW.init_temporal_state_dict(512, IW.UCF_HEAD, 0)
    assert sum(v.numel() for v in tm.values()) == 10110977


def test_abi_error_behaviour_without_gpu():
    """argument validation happens before any launch: error codes + acx_last_error text, never an abort."""
    import ctypes as C
    from anomalyclip_amd import _lib as L
    lib = L.lib()
    h = C.c_void_p()
    if not torch.cuda.is_available():
        rc = lib.acx_create(C.byref(h), 0)
        assert rc in (L_E := (-3, -2))            # ACX_E_HIP (no device) / ACX_E_UNSUPPORTED (not gfx950)
        assert b"acx_create" in lib.acx_last_error(None)
    assert lib.acx_gemm(None, None, None) == -1 and b"null" in lib.acx_last_error(None)
    d = L.GemmDesc()
    buf = (C.c_float * 64)()
    d.A = d.W = d.C = C.addressof(buf)
    d.M, d.N, d.K = 4, 4, 6                       # K % 4 != 0
    d.lda = d.ldw = 6
    d.ldc = 4
    assert lib.acx_gemm(None, C.byref(d), None) == -1
    d.K = d.lda = d.ldw = 8
    d.prec = 7
    assert lib.acx_gemm(None, C.byref(d), None) == -2       # ACX_E_UNSUPPORTED
    d.prec = 0
    d.amap = 1                                    # conv3x3 without geometry
    assert lib.acx_gemm(None, C.byref(d), None) == -1
    assert lib.acx_layernorm(None, None, 0, None, None, None, 0, 0, 4, 64, 1e-5, 0, None) == -1
    assert lib.acx_attention(None, C.addressof(buf), 192, C.addressof(buf), 64, 1, 500, 1, 0, None) == -2   # L > 224
    assert lib.acx_select_idx(None, None, None, None, None, None, None, 4, 32, 16, 13, 7, 3, 3, None) == -1
    assert lib.acx_adamw(None, C.addressof(buf), C.addressof(buf), C.addressof(buf), C.addressof(buf), 8, 1e-3, 0.9, 0.999,
                         1e-8, 0.0, 0, None) == -1                                                           # step starts at 1
    # empty work is a no-op, not an error
    assert lib.acx_layernorm(None, C.addressof(buf), 64, C.addressof(buf), C.addressof(buf), C.addressof(buf), 64, 0, 0, 64,
                             1e-5, 0, None) == 0
    assert lib.acx_gather_rows(None, C.addressof(buf), C.addressof(buf), C.addressof(buf), 0, 64, None) == 0


def test_lightning_checkpoint_round_trip(tmp_path, prompts_table):
    from anomalyclip_amd import checkpoint, init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
    kw = dict(arch="tiny", labels_key="ucf", emb_size=64, depth=1, heads=2, dim_heads=None, num_segments=32, seg_length=16,
              concat_features=False, normal_id=7, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, num_topk=3,
              num_bottomk=3)
    a, b = AnomalyCLIP(**kw), AnomalyCLIP(**kw)
    toks = torch.tensor(prompts_table["ucf"]["tokenized_prompts"], dtype=torch.int32)
    a.load_state_dict(IW.init_anomalyclip_state_dict(IW.TINY, IW.HeadConfig(emb_size=64, heads=2), toks, 5), strict=True)
    ck = {"state_dict": {**checkpoint.to_lightning_state_dict(a), "train_loss.mean_value": torch.zeros(1)},
          "epoch": 3, "hyper_parameters": {}}
    # CLIP towers are stored in half precision by some exports: must be up-cast on load
    ck["state_dict"]["net.image_encoder.proj"] = ck["state_dict"]["net.image_encoder.proj"].half()
    path = str(tmp_path / "last.ckpt")
    torch.save(ck, path)
    checkpoint.load_into(b, path)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb
        if ka == "image_encoder.proj":
            assert torch.equal(vb, va.half().float())
        else:
            assert torch.equal(va, vb), ka


# ====================================================================================================== operator surface
def _config_keys():
    import json
    return json.load(open(os.path.join(REPO, "tests", "golden", "config_keys.json")))


def test_module_hook_signatures_match_reference():
    """drop-in boundary, Python half: every hook pytorch_lightning.Trainer calls on the reference's AnomalyCLIPModule
    exists here with the SAME positional argument names (fixture: names parsed from the reference with `ast`), the
    same **kwargs constructor and the same @rank_zero_only hooks (anomaly_clip_module.py:46-53,134,203,301,339,406,
    458,501,693)."""
    import inspect
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
    from anomalyclip_amd.components.loss import ComputeLoss
    hooks = _config_keys()["module_hooks"]
    for name, spec in hooks.items():
        owner = {"AnomalyCLIP.forward": AnomalyCLIP.forward, "ComputeLoss.__call__": ComputeLoss.__call__}.get(name)
        fn = owner if owner is not None else getattr(AnomalyCLIPModule, name)
        sig = inspect.signature(inspect.unwrap(fn))
        params = list(sig.parameters.values())
        pos = [p.name for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        assert pos[:len(spec["args"])] == spec["args"], (name, pos, spec["args"])
        # anything beyond the reference's arguments must be optional (the Trainer never passes it)
        for p in params[len(spec["args"]):]:
            assert p.default is not p.empty or p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL), (name, p.name)
        if spec["kwarg"]:
            assert any(p.kind == p.VAR_KEYWORD for p in params), name
        if "rank_zero_only" in spec["decorators"]:
            assert hasattr(fn, "__wrapped__"), f"{name} must be @rank_zero_only"


@pytest.mark.parametrize("key", ["ucf", "sht", "xd"])
def test_module_builds_from_reference_config_blocks(key, tmp_path):
    """`net:` / `loss:` / `solver:` / `optimizer:` / `scheduler:` blocks of the reference's three model configs (fixture
    copies of keys + scalar values, `${data.*}` resolved) instantiate the mirrors unchanged -- including the keys the
    reference's code never reads (`dropout_prob`, `temporal_module`, ...) and `labels_file` paths that do not exist here."""
    from functools import partial
    from types import SimpleNamespace
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
    from anomalyclip_amd.components.loss import ComputeLoss
    from anomalyclip_amd.components.scheduler import WarmupCosineAnnealingLR
    from anomalyclip_amd.optim import AcxAdamW
    cfg = _config_keys()["configs"][key]
    net_kw = {k: v for k, v in cfg["net"].items() if k != "_target_"}
    assert cfg["net"]["_target_"] == "src.models.components.anomaly_clip.AnomalyCLIP"
    net = AnomalyCLIP(**net_kw, clip_geometry=IW.TINY)          # tiny CLIP geometry: construction only, no 150 M-param init
    assert len(net.classnames) == cfg["num_classes"] and net.normal_id == cfg["data"]["normal_id"]
    hc = {"ucf": IW.UCF_HEAD, "sht": IW.SHT_HEAD, "xd": IW.XD_HEAD}[key]
    assert (net.emb_size, net.depth, net.heads, net.concat_features) == (hc.emb_size, hc.depth, hc.heads, hc.concat_features)
    assert net.temporal_model.input_size == net.embedding_dim + (cfg["num_classes"] - 1) * int(cfg["net"]["concat_features"])
    loss = ComputeLoss(**{k: v for k, v in cfg["loss"].items() if k != "_target_"})
    opt_kw = {k: v for k, v in cfg["optimizer"].items() if not k.startswith("_")}
    sch_kw = {k: v for k, v in cfg["scheduler"].items() if not k.startswith("_")}
    mod = AnomalyCLIPModule(net=net, optimizer=partial(AcxAdamW, **opt_kw), scheduler=partial(WarmupCosineAnnealingLR, **sch_kw),
                            loss=loss, num_classes=cfg["num_classes"], solver=cfg["solver"], save_dir=str(tmp_path))
    assert mod.hparams.num_classes == cfg["num_classes"] and mod.hparams.solver.lr == cfg["solver"]["lr"]
    # frozen backbone, trainable set (anomaly_clip_module.py:68-74)
    assert not any(p.requires_grad for p in net.image_encoder.parameters())
    assert net.text_encoder.text_projection.requires_grad and net.prompt_learner.ctx.requires_grad
    assert not any(p.requires_grad for n, p in net.text_encoder.named_parameters() if n != "text_projection")
    object.__setattr__(mod, "trainer", SimpleNamespace(max_epochs=50, current_epoch=0, datamodule=None, ckpt_path=None))
    out = mod.configure_optimizers()                              # no arguments, like the Trainer calls it
    groups = out["optimizer"].param_groups
    assert [g["name"] for g in groups] == ["selector_model", "temporal_model", "prompt_learner", "text_projection"]
    assert all(abs(g["initial_lr"] - cfg["solver"]["lr"]) < 1e-12 and g["weight_decay"] == 0.2 for g in groups)
    assert out["lr_scheduler"]["interval"] == "epoch" and out["lr_scheduler"]["monitor"] == "train/loss"
    assert isinstance(out["lr_scheduler"]["scheduler"], WarmupCosineAnnealingLR)


def test_builtin_trainer_calls_hooks_in_lightning_order(tmp_path):
    """anomalyclip_amd.trainer.Trainer drives a module exactly through the hooks above (stub module: no kernels)."""
    from types import SimpleNamespace
    from anomalyclip_amd.trainer import Trainer
    calls = []

    class Stub:
        device = torch.device("cpu")
        hparams = {}
        net = SimpleNamespace(train=lambda: calls.append("net.train"), eval=lambda: calls.append("net.eval"),
                              state_dict=lambda: {})

        def configure_optimizers(self):
            calls.append("configure_optimizers")
            return {"optimizer": "OPT"}

        def on_train_start(self):
            calls.append("on_train_start")

        def train_batch(self, batch, opt, i):
            assert opt == "OPT" and len(batch) == 2            # (nbatch, abatch) zipped from the two loaders
            calls.append(f"train_batch{i}")

        def on_train_epoch_end(self):
            calls.append("on_train_epoch_end")

        def validation_step(self, batch, i):
            calls.append(f"validation_step{i}")

        def on_validation_epoch_end(self):
            calls.append("on_validation_epoch_end")
            return {"auc_roc": 0.5}

        def on_test_start(self):
            calls.append("on_test_start")

        def test_step(self, batch, i):
            calls.append(f"test_step{i}")
            return {"i": i}

        def test_epoch_end(self, outputs):
            calls.append(f"test_epoch_end{len(outputs)}")
            return {"ok": 1}

    dm = SimpleNamespace(setup=lambda stage: calls.append("setup:" + stage),
                         train_dataloader=lambda: [[("n0", 0), ("n1", 1)], [("a0", 0), ("a1", 1)]],
                         val_dataloader=lambda: [1, 2], test_dataloader=lambda: [1, 2, 3])
    tr = Trainer(max_epochs=1, accelerator="gpu", devices=4, strategy="ddp", sync_batchnorm=True, num_sanity_val_steps=0,
                 default_root_dir=str(tmp_path))
    m = Stub()
    tr.fit(m, dm)
    assert calls == ["setup:fit", "configure_optimizers", "on_train_start", "net.train", "train_batch0", "train_batch1",
                     "on_train_epoch_end", "net.eval", "validation_step0", "validation_step1", "on_validation_epoch_end"]
    assert m.trainer is tr and tr.datamodule is dm and os.path.isfile(tr.ckpt_path)
    ck = torch.load(tr.ckpt_path, weights_only=False)
    assert set(ck) >= {"state_dict", "epoch", "global_step", "hyper_parameters"} and ck["global_step"] == 2
    calls.clear()
    assert tr.test(m, dm) == [{"ok": 1}]
    assert calls == ["setup:test", "net.eval", "on_test_start", "test_step0", "test_step1", "test_step2", "test_epoch_end3"]


def test_compat_install_resolves_every_reference_target():
    """compat.install() makes every `_target_` of the reference's model configs (fixture: config_keys.json, read from the
    reference's YAML) importable and bound to the mirror classes -- the YAML runs unedited."""
    import importlib
    import inspect
    from anomalyclip_amd import compat
    import json
    import sys
    keys = json.load(open(os.path.join(REPO, "tests", "golden", "config_keys.json")))
    targets = set()

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "_target_":
                    targets.add(v)
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(keys["configs"])
    targets.add("src.models.anomaly_clip_module.AnomalyCLIPModule")               # configs/model/*.yaml:1
    assert {t for t in targets if t.startswith("src.")} == {
        "src.models.components.anomaly_clip.AnomalyCLIP", "src.models.components.loss.ComputeLoss",
        "src.models.components.scheduler.WarmupCosineAnnealingLR", "src.models.anomaly_clip_module.AnomalyCLIPModule"}
    before = {k for k in sys.modules if k == "src" or k.startswith("src.")}
    try:
        compat.install()
        compat.install()                                                            # idempotent
        for t in sorted(targets):
            obj = compat.resolve(t)
            assert inspect.isclass(obj), t
            if t.startswith("src."):
                assert obj.__module__.startswith("anomalyclip_amd."), (t, obj.__module__)
        from src.models.components.loss import ComputeLoss                           # plain import statements work too
        from anomalyclip_amd.components.loss import ComputeLoss as Mirror
        assert ComputeLoss is Mirror
        assert importlib.import_module("src.models.anomaly_clip_module").AnomalyCLIPModule.__name__ == "AnomalyCLIPModule"
    finally:
        compat.uninstall()
    assert {k for k in sys.modules if k == "src" or k.startswith("src.")} == before


def test_bench_gpus_n_self_launches_under_torchrun(monkeypatch):
    """`python bench.py --gpus N` typed as such (no WORLD_SIZE in the environment) becomes the launcher: the same argv under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`."""
    import importlib
    import subprocess
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_trainer_cycles_the_shorter_train_loader():
    """Lightning 1.8 iterates a list of train loaders in 'max_size_cycle' mode: max(len) steps per epoch, the shorter
    loader restarted (the reference's [normal, abnormal] loaders have unequal lengths on every dataset)."""
    from anomalyclip_amd.trainer import Trainer
    got = list(Trainer()._train_batches([["n0", "n1", "n2", "n3", "n4"], ["a0", "a1"]], 0))
    assert got == [("n0", "a0"), ("n1", "a1"), ("n2", "a0"), ("n3", "a1"), ("n4", "a0")]
    assert list(Trainer()._train_batches(iter(["x", "y"]), 0)) == ["x", "y"]          # a single loader passes through


def test_ctx_init_from_token_embeddings(prompts_table):
    """coop.py:19-34: a non-empty `ctx_init` initialises the context from the token embeddings of its words (ids = positions
    1..n_ctx of the tokenised prompts), class-specific (repeated) or shared; n_ctx follows the word count."""
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP
    toks = torch.tensor(prompts_table["ucf"]["tokenized_prompts"], dtype=torch.int32).clone()
    toks[:, 1:4] = torch.tensor([320, 1125, 539])                  # pretend "a video of" replaces the first three X tokens
    kw = dict(arch="tiny", labels_key="ucf", emb_size=64, depth=1, heads=2, dim_heads=None, num_segments=32, seg_length=16,
              concat_features=False, normal_id=7, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, num_topk=3,
              num_bottomk=3, clip_geometry=IW.TINY, tokenized_prompts=toks)
    for shared in (False, True):
        net = AnomalyCLIP(**kw, ctx_init="a_video of", shared_context=shared)
        pl = net.prompt_learner
        want = net.token_embedding.weight.detach()[toks[0, 1:4].long()]
        assert pl.n_ctx == 3 and pl.token_suffix.shape[1] == 77 - 1 - 3
        if shared:
            assert pl.ctx.shape == (3, IW.TINY.transformer_width) and torch.equal(pl.ctx.detach(), want)
        else:
            assert pl.ctx.shape == (14, 3, IW.TINY.transformer_width)
            assert all(torch.equal(pl.ctx.detach()[c], want) for c in range(14))
    with pytest.raises(ValueError):
        AnomalyCLIP(**{k: v for k, v in kw.items() if k != "tokenized_prompts"}, ctx_init="a video of")


# ---------------------------------------------------------------------------------------------------------------------
# Executable statements of two scheduling arguments the HIP kernels rely on (pure Python models of the index / phase
# arithmetic in csrc/acx_gemm_p256.h and csrc/acx_gemm_p8.h; the kernels themselves are checked numerically on the GPU).
@pytest.mark.parametrize("M,N,nb", [(100864, 2304, 256), (100864, 768, 256), (33001, 2300, 256), (20000, 3072, 304),
                                     (65, 256, 256), (4097, 1000, 13)])
def test_strip_stream_partition_model(M, N, nb):
    """gemm_f32_p256_kernel: work units = 64-row x 256-column strips in column-major order; CU c (XCD-contiguous index)
    owns [U c / n, U (c + 1) / n) and walks it in tiles of <= 4 units that never cross a column tile.  Every unit must be
    covered exactly once, and the block -> range map must be a bijection for any grid size."""
    TN, RU = (N + 255) // 256, (M + 63) // 64
    U = TN * RU
    seen = np.zeros(U, dtype=np.int32)
    cidxs = set()
    for bid in range(nb):
        xcd, qb, rb = bid & 7, nb >> 3, nb & 7
        cidx = (xcd * (qb + 1) if xcd < rb else rb * (qb + 1) + (xcd - rb) * qb) + (bid >> 3)
        cidxs.add(cidx)
        pos, end = U * cidx // nb, U * (cidx + 1) // nb
        while pos < end:
            col = pos // RU
            ru = pos - col * RU
            n = min(4, end - pos, RU - ru)
            assert 1 <= n <= 4 and ru + n <= RU
            seen[pos:pos + n] += 1
            pos += n
    assert cidxs == set(range(nb))
    assert (seen == 1).all()


def test_phase_interleaved_dma_schedule_model():
    """gemm_bf16_p8_kernel: phase p of K-tile s issues one half-tile -- BH1(s+1), AH1(s+1), AH0(s+2), BH0(s+2) -- into the
    slot of that K-tile and ends with vmcnt(8) (everything but the four newest half-tiles has landed); phase 1 reads AH0 and
    BH0 of K-tile s, phase 2 BH1, phase 3 AH1.  Rules (guide): a half-tile is READ at least one phase after the wait that
    retired it, and a buffer is RESTAGED at least two phases after its last read.  Checked over a long stream, including
    the tile-end variant (the next tile's second K-tile completed before the epilogue, no waits for eight phases)."""
    issues = {}            # (half, ktile) -> global phase of issue; prologue: K-tile 0 entirely, AH0 / BH0 of K-tile 1
    order = []             # issue order (for the counted wait)
    for h in ("AH0", "BH0", "BH1", "AH1"):
        issues[(h, 0)] = 0; order.append((h, 0))
    for h in ("AH0", "BH0"):
        issues[(h, 1)] = 0; order.append((h, 1))
    retired_at = {k: 0 for k in list(issues)[:2]}      # prologue vmcnt(8): all but the newest four
    nkt = 40
    plan = {1: ("BH1", 1), 2: ("AH1", 1), 3: ("AH0", 2), 4: ("BH0", 2)}
    reads = {1: ("AH0", "BH0"), 2: ("BH1",), 3: ("AH1",), 4: ()}
    last_read = {}
    for s in range(nkt):
        for ph in (1, 2, 3, 4):
            g = 4 * s + ph
            for h in reads[ph]:
                key = (h, s)
                assert key in retired_at and retired_at[key] <= g - 1, (key, g)      # read >= 1 phase after the retiring wait
                last_read[(h, s & 1)] = g
            h, ds = plan[ph]
            key = (h, s + ds)
            slot = (h, (s + ds) & 1)
            assert slot not in last_read or last_read[slot] <= g - 2, (key, g)       # restage >= 2 phases after the last read
            issues[key] = g; order.append(key)
            for k in order[:-4]:                                                     # vmcnt(8): all but the newest four
                retired_at.setdefault(k, g)
    # tile end after K-tile s (slot 1): BH1 / AH1 of K-tile s + 2 go out early, vmcnt(0), then two K-tiles without waits
    s = 11
    g_end = 4 * s + 4
    read_phase = {"AH0": 1, "BH0": 1, "BH1": 2, "AH1": 3}
    for h in ("BH1", "AH1"):                             # their slot (that of K-tile s) was last read in phases 2 / 3 of s,
        assert 4 * s + read_phase[h] <= g_end - 1        # i.e. before the tile end's re-align barrier
    landed = {k for k, g in issues.items() if g <= g_end} | {("BH1", s + 2), ("AH1", s + 2)}
    for kt in (s + 1, s + 2):
        for h in ("AH0", "BH0", "BH1", "AH1"):
            assert (h, kt) in landed, (h, kt)


def test_f16x3_range_report_on_vit_b16():
    """VisionTransformer.f16x3_range_report: the operand magnitudes the opt-in precision "f16x3" must keep inside fp16's range, proven
    from the weights -- a randomly initialised ViT-B/16 (the benchmark's weights) has a margin of more than an order of magnitude; a
    weight of 100 (2^10 x 100 > 65504) is reported as unsafe."""
    import torch
    from anomalyclip_amd.components.clip_vit import VisionTransformer
    torch.manual_seed(0)
    vit = VisionTransformer(224, 16, 768, 2, 12, 512, precision="f16x3")
    rep = vit.f16x3_range_report()
    assert rep["safe"] and rep["margin"] > 10.0, rep
    assert set(rep["bounds"]) == {"weight_planes", "layernorm_out", "attention_out", "mlp_hidden"}
    with torch.no_grad():
        vit.transformer.resblocks[1].mlp.c_proj.weight[3, 5] = 100.0
    rep2 = vit.f16x3_range_report()
    assert not rep2["safe"] and rep2["bounds"]["weight_planes"] > 65504.0


@pytest.mark.parametrize("E,N,L,want", [(256, 32, 16, True), (512, 32, 16, True), (128, 32, 16, True), (384, 32, 16, False),
                                         (64, 32, 16, False), (256, 8, 16, False), (256, 24, 16, False)])
def test_x6_convs_only_for_shapes_the_kernels_take(E, N, L, want):
    """TemporalModel.x6_convs(): the bf16 x 6 convolutions only where acx_gemm / acx_gemm_tn_x6 have tile geometries -- E % 256 == 0 or
    E == 128 (the shipped configs: UCF-Crime / ShanghaiTech 256, XD-Violence 128), a power-of-two token grid of whole 256-row tiles;
    every other head keeps the f32 MFMA convolutions instead of failing with ACX_E_UNSUPPORTED (no fallback inside _ff / TemporalFn)."""
    from anomalyclip_amd.components.temporal_model import TemporalModel
    m = TemporalModel(512, E, 1, 8, None, 1, N, L)
    assert m.x6_convs() is False              # stand-alone default "f32"; AnomalyCLIP(precision="auto") sets it on its head
    m.precision = "auto"
    assert m.x6_convs() is want
