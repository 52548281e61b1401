"""AnomalyCLIPModule -- host-side mirror of the reference's LightningModule
(src/models/anomaly_clip_module.py:31-750) for the hot path.  Same constructor (`net, optimizer, scheduler, loss,
**kwargs` with kwargs = num_classes / solver / save_dir, configs/model/*.yaml:2,63-68) and the SAME Trainer-called
hook signatures:

    on_train_start(self)                    :134   ncentroid.pt under hparams.save_dir, else computed from
                                                   self.trainer.datamodule.train_dataloader_test_mode()
    training_step(self, batch, batch_idx)   :203   8 loss terms -> running means + self.log("train/*")
    validation_step(self, batch, batch_idx) :301   4- or 5-tuple batches, results ACCUMULATED on self
    on_validation_epoch_end(self)           :339   metrics_{epoch}.json under hparams.save_dir, lists cleared
    on_test_start(self)                     :406   ncentroid next to the checkpoint's run directory
    test_step(self, batch, batch_idx)       :458   @rank_zero_only, returns the per-video dict
    test_epoch_end(self, outputs)           :501   @rank_zero_only, metrics.json under <logs>/eval/runs/<run>
    configure_optimizers(self)              :693   4 param groups, CosineAnnealingLR(T_max = trainer.max_epochs) successor

`self.trainer` is whatever drives the module: pytorch_lightning's Trainer when that package is importable (the
class then derives from LightningModule), otherwise anomalyclip_amd.trainer.Trainer -- a thin loop that calls
exactly these hooks in Lightning's order (one process per GPU; gradients through parallel.GradBuckets).  Both are
read through the same attributes: `trainer.datamodule` (`.train_dataloader_test_mode()`, `.num_classes`,
`.hparams.{load_from_features, normal_id, labels_file, visualize}`), `trainer.current_epoch`, `trainer.max_epochs`,
`trainer.ckpt_path`.

The metric numbers (AUROC, AP, mAUC, mAP, top-1/5, optimal threshold; :339-404, :501-626) come from libacx's sort /
scan kernels (metrics.py) instead of torchmetrics; plots and the Visualizer are out of scope.
The reference hard-codes /usr/src/app/logs/{train,eval}/runs/<run> (:409,:596); the same layout is used under
`hparams.logs_root` (default "/usr/src/app/logs")."""
from __future__ import annotations

import contextlib
import json
import os
from pathlib import Path
from typing import Any, List, Optional

import torch

from . import ops, parallel
from .optim import AcxAdamW

X6_RESERVE_ROWS = 16384    # rank shares of at most this many feature rows ...
X6_RESERVE_CUS = 32        # ... keep this many CUs out of the persistent bf16 x 6 kernels' grids (train_batch)
X6_SPLIT_ROWS = 8192       # up to here every convolution of the share is K-split anyway (fewer tiles than CUs); above, the capped
                           # grid would turn 256 tiles into two rounds: the partly filled last round is K-split too (ACX_OPT_X6_TAIL_SPLIT)

try:  # pragma: no cover - not installed in the build image
    from pytorch_lightning import LightningModule as _Base
    _HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module
    _HAVE_LIGHTNING = False


class AttrDict(dict):
    """`self.hparams` without Lightning: attribute + item access, nested dicts wrapped on the way out."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v


def _get(obj, key, default=None):
    """hparams / solver blocks arrive as dict, AttrDict, DictConfig or Namespace."""
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(key, default)
    try:
        return getattr(obj, key)
    except Exception:  # noqa: BLE001
        try:
            return obj[key]
        except Exception:  # noqa: BLE001
            return default


class MeanMetric:
    """torchmetrics.MeanMetric as the module uses it (:77-84): running mean of scalars, kept on the device (no host
    sync per step); under data parallelism `compute()` averages over ranks like torchmetrics' epoch-end sync."""

    def __init__(self, device_fn=None):
        self.total: Optional[torch.Tensor] = None
        self.count = 0
        self._device_fn = device_fn             # where an update-less rank builds its (0, 0) contribution

    def __call__(self, value):
        self.update(value)

    def update(self, value):
        v = torch.as_tensor(value).detach().float().reshape(())
        self.total = v.clone() if self.total is None else self.total + v
        self.count += 1

    def compute(self) -> torch.Tensor:
        # a rank that saw no update still JOINS the all-reduce (with (0, 0)): returning early would leave the others blocked
        if self.total is None:
            dev = self._device_fn() if self._device_fn is not None else torch.device("cpu")
            t = torch.zeros(2, dtype=torch.float32, device=dev)
        else:
            t = torch.stack([self.total, self.total.new_tensor(float(self.count))])
        if parallel.is_distributed():
            parallel.all_reduce_sum_(t)
        return t[0] / t[1]                      # nan when nobody updated, like torchmetrics' empty MeanMetric

    def reset(self):
        self.total, self.count = None, 0


class _MeterBank:
    """The eight loss meters of a training step as ONE device vector: the criterion returns its eight terms as views of one
    tensor, so a step costs one add (not eight) and an epoch end one all-reduce (not eight).  `meter(i)` hands out objects with
    MeanMetric's interface (the reference's per-loss attributes, anomaly_clip_module.py:77-84)."""

    def __init__(self, n: int, device_fn):
        self.n, self._device_fn = n, device_fn
        self.total: Optional[torch.Tensor] = None
        self.count = 0
        self._attached = False

    def update(self, values):
        base = getattr(values[0], "_base", None)
        same = base is not None and base.shape == (self.n,) and all(getattr(v, "_base", None) is base for v in values)
        vec = (base if same else torch.stack([torch.as_tensor(v).reshape(()) for v in values])).detach().float()
        if self._attached:
            self.total += vec
        else:
            self.total = vec.clone() if self.total is None else self.total + vec
        self.count += 1

    def compute(self) -> torch.Tensor:
        """[n] means; every rank takes part in the exchange, also one that saw no update (it contributes zeros)."""
        if self.total is None:
            t = torch.zeros(self.n + 1, dtype=torch.float32, device=self._device_fn())
        else:
            t = torch.cat([self.total, self.total.new_full((1,), float(self.count))])
        if parallel.is_distributed():
            parallel.all_reduce_sum_(t)
        return t[: self.n] / t[self.n]           # nan when nobody updated, like torchmetrics' empty MeanMetric

    def reset(self):
        if self._attached:                       # the step graph's accumulator: zeroed in place, its address is captured
            self.total.zero_()
            self.count = 0
        else:
            self.total, self.count = None, 0

    def attach(self, total: torch.Tensor):
        """use `total` ([n] device tensor, accumulated inside the whole-step graph) as the running sums"""
        if self.total is not None and self.total is not total:
            total.copy_(self.total)
        else:
            total.zero_()
            self.count = 0
        self.total, self._attached = total, True

    def meter(self, i: int):
        bank = self

        class _View:
            count = property(lambda _s: bank.count)

            def compute(_s):
                return bank.compute()[i]

            def reset(_s):
                bank.reset()
        return _View()


_LOSS_NAMES = ("train_loss", "dir_abn_loss", "dir_nor_loss", "topk_abn_loss", "bottomk_abn_loss", "topk_nor_loss",
               "smooth_loss", "sparse_loss")


class AnomalyCLIPModule(_Base):
    def __init__(self, net: torch.nn.Module, optimizer=None, scheduler=None, loss=None, **kwargs):
        super().__init__()
        if _HAVE_LIGHTNING:  # pragma: no cover
            self.save_hyperparameters(logger=False, ignore=["net"])
        else:
            self.hparams = AttrDict(kwargs)
        self.net = net
        self.criterion = loss
        self.optimizer = optimizer
        self.scheduler = scheduler
        # freezing backbone (anomaly_clip_module.py:68-74)
        for p in self.net.image_encoder.parameters():
            p.requires_grad = False
        for p in self.net.text_encoder.parameters():
            p.requires_grad = False
        self.net.text_encoder.text_projection.requires_grad = True
        for p in self.net.token_embedding.parameters():
            p.requires_grad = False
        # for averaging loss across batches (:77-84)
        self._meters = _MeterBank(len(_LOSS_NAMES), lambda: self.device)
        for i, n in enumerate(_LOSS_NAMES):
            object.__setattr__(self, n, self._meters.meter(i))
        self.ncentroid: Optional[torch.Tensor] = None
        self.labels: List[torch.Tensor] = []
        self.abnormal_scores: List[torch.Tensor] = []
        self.class_probs: List[torch.Tensor] = []
        self.visualizer = None
        self.logged: dict = {}                    # last value per self.log() name (the fallback logger)
        self._buckets: Optional[parallel.GradBuckets] = None
        if not _HAVE_LIGHTNING:
            object.__setattr__(self, "trainer", None)

    # ------------------------------------------------------------------ small Lightning stand-ins
    if not _HAVE_LIGHTNING:
        @property
        def device(self) -> torch.device:
            return next(self.net.temporal_model.parameters()).device

        def log(self, name, value, **kw):
            self.logged[name] = value

    def _datamodule(self):
        tr = getattr(self, "trainer", None)
        return getattr(tr, "datamodule", None) if tr is not None else None

    def _logs_root(self) -> str:
        return str(_get(self.hparams, "logs_root", "/usr/src/app/logs"))

    # ------------------------------------------------------------------ forward (anomaly_clip_module.py:118-132)
    def forward(self, image_features, labels, ncentroid, segment_size: int = 1, test_mode: bool = False):
        return self.net(image_features, labels, ncentroid, segment_size, test_mode)

    # ------------------------------------------------------------------ ncentroid (:145-171, :419-445)
    @torch.no_grad()
    def compute_ncentroid(self, loader, load_from_features: bool = True) -> torch.Tensor:
        """mean feature over every frame of every normal training video.  Batches are the test-mode datasets' 4- or
        5-tuples (features, labels, label, segment_size[, path]) -- both arities occur in the reference (:153,:427).
        Under data parallelism a rank may be handed a shard of the loader: (sum, count) is all-reduced; handing every
        rank the full loader (what the reference does, SURVEY section 5 "DDP quirks") gives the same mean."""
        dev = self.device
        D = self.net.embedding_dim
        acc = torch.zeros(D, dtype=torch.float32, device=dev)
        count = 0
        for batch in loader:
            feats, nlabels = batch[0], batch[1]
            n = int(torch.as_tensor(nlabels).reshape(-1).shape[0])                   # len(nlabels.squeeze())
            if load_from_features:
                f = feats.reshape(-1, feats.shape[-1])[:n].to(dev, torch.float32).contiguous()
            else:
                b, t, c, h, w = feats.shape
                f = self.net.image_encoder(feats.view(-1, c, h, w)[:n].to(dev))
            ops.colsum_(acc, f)
            count += f.shape[0]
        cnt = torch.tensor([float(count)], device=dev)
        if parallel.is_distributed():
            parallel.all_reduce_sum_(acc)
            parallel.all_reduce_sum_(cnt)
        self._set_ncentroid(acc / cnt)
        return self.ncentroid

    def _set_ncentroid(self, value: torch.Tensor):
        """ncentroid lives in ONE persistent device buffer: reloads copy into it, so its address -- part of the temporal
        graphs' key, and the operand the captured kernels read -- never changes and no step pays a host-to-device copy."""
        value = value.detach().to(torch.float32).reshape(-1)
        cur = self.ncentroid
        if torch.is_tensor(cur) and cur.device == self.device and cur.shape == value.shape and cur.dtype == torch.float32:
            cur.copy_(value)
        else:
            self.ncentroid = value.to(self.device).contiguous()

    def _load_or_compute_ncentroid(self, save_dir: Path):
        save_dir.mkdir(parents=True, exist_ok=True)
        f = save_dir / "ncentroid.pt"
        if f.is_file():
            self._set_ncentroid(torch.load(f, map_location="cpu"))
            return
        dm = self._datamodule()
        if dm is None:
            raise RuntimeError(f"{f} does not exist and there is no trainer.datamodule to compute it from")
        loader = dm.train_dataloader_test_mode()
        self.compute_ncentroid(loader, bool(_get(_get(dm, "hparams"), "load_from_features", True)))
        if parallel.rank() == 0:                 # the reference lets every rank write the same file
            torch.save(self.ncentroid.cpu(), f)

    def on_train_start(self):
        self._load_or_compute_ncentroid(Path(_get(self.hparams, "save_dir")))

    # ------------------------------------------------------------------ training (:173-293)
    def model_step(self, batch: Any):
        nbatch, abatch = batch
        nimage_features, nlabel = nbatch
        aimage_features, alabel = abatch
        image_features = torch.cat((aimage_features, nimage_features), 0)
        labels = torch.cat((alabel, nlabel), 0)
        out = self.forward(image_features, labels, ncentroid=self.ncentroid)
        logits, logits_topk, scores, ia, in_, ba = out
        return logits, logits_topk, labels, scores, ia, in_, ba

    def training_step(self, batch: Any, batch_idx: int = 0):
        sim, sim_topk, labels, scores, ia, in_, ba = self.model_step(batch)
        losses = self.criterion(sim, sim_topk, labels, scores, ia, in_, ba)
        self.last_losses = losses
        if _HAVE_LIGHTNING:  # pragma: no cover - Lightning is not installed in the build image
            # the reference logs torchmetrics MeanMetric objects, which Lightning computes (synchronised over ranks) and
            # resets at epoch end.  self.log accepts numbers / tensors / torchmetrics.Metric only, so the step value is logged
            # with on_epoch=True: Lightning's own epoch mean, rank-synchronised (sync_dist) like the reference's metric, one
            # sample per step (batch_size=1: the (normal, abnormal) tuple has no unambiguous batch dimension)
            for name, value in zip(_LOSS_NAMES, losses):
                self.log("train/" + ("loss" if name == "train_loss" else name), value.detach(), on_step=False, on_epoch=True,
                         prog_bar=True, sync_dist=True, batch_size=1)
        else:
            self._meters.update(losses)                                             # :244-293, all eight meters in one add
            for name in _LOSS_NAMES:
                # the built-in loop keeps the meter; on_train_epoch_end turns it into the epoch mean
                self.log("train/" + ("loss" if name == "train_loss" else name), getattr(self, name), on_step=False,
                         on_epoch=True, prog_bar=True)
        return {"loss": losses[0]}

    def on_train_epoch_end(self):
        """Per-epoch means (the reference's torchmetrics are reset by Lightning after every epoch): compute -- every rank
        takes part in the exchange -- publish, reset."""
        if _HAVE_LIGHTNING:  # pragma: no cover - Lightning reduces and resets what training_step logged
            return
        means = self._meters.compute()
        for i, name in enumerate(_LOSS_NAMES):
            self.logged["train/" + ("loss" if name == "train_loss" else name)] = means[i]
        self._meters.reset()

    # ------------------------------------------------------------------ evaluation (:301-337, :458-498)
    def _score_video(self, batch):
        image_features, labels, segment_size = batch[0], batch[1], batch[3]        # 4-tuple (:302) or 5-tuple (:460)
        dev = self.device
        image_features = image_features.to(dev)
        labels = torch.as_tensor(labels).squeeze(0).to(dev)
        with torch.no_grad():
            similarity, abnormal_scores = self.forward(image_features, labels, self.ncentroid, int(segment_size), test_mode=True)
            class_probs = ops.class_probs(similarity.contiguous(), abnormal_scores.contiguous())   # softmax * score
        n = labels.shape[0]                                                          # remove padded frames
        return abnormal_scores[:n], labels, class_probs[:n]

    def score_videos(self, batches: List[Any]):
        """`_score_video` for SEVERAL test batches in one forward (AnomalyCLIP.forward_test_many): one selector / temporal /
        post-processing launch sequence over all tiles, one text-tower evaluation, then split per video and truncated to the
        real frames like anomaly_clip_module.py:474-483.  -> list of (abnormal_scores, labels, class_probs)."""
        net = self.net
        if len(batches) == 1 or not getattr(net, "load_from_features", True):
            return [self._score_video(b) for b in batches]
        dev = self.device
        feats, rows_per_crop, segs, labs = [], [], [], []
        for b in batches:
            f, labels, S = b[0], b[1], int(b[3])
            f = f.to(dev)
            if f.dim() != 4 or f.shape[0] != 1 or f.shape[1] != net.ncrops:
                return [self._score_video(bb) for bb in batches]
            feats.append(f.reshape(-1, f.shape[-1]))
            rows_per_crop.append(f.shape[2])
            segs.append(S)
            labs.append(torch.as_tensor(labels).squeeze(0).to(dev))
        x = feats[0] if len(feats) == 1 else torch.cat(feats, 0)
        with torch.no_grad():
            sim, sc = net.forward_test_many(x, rows_per_crop, segs, self.ncentroid)
            probs = ops.class_probs(sim.contiguous(), sc.contiguous())
        out, r0 = [], 0
        for rows, lab in zip(rows_per_crop, labs):
            r1 = r0 + rows * net.ncrops
            s_v, p_v = sc[r0:r1], probs[r0:r1]
            if net.stride != 1:                                                      # anomaly_clip.py:149-150
                s_v, p_v = s_v.repeat_interleave(net.stride, dim=0), p_v.repeat_interleave(net.stride, dim=0)
            n = lab.shape[0]
            out.append((s_v[:n], lab, p_v[:n]))
            r0 = r1
        return out

    @parallel.rank_zero_only
    def test_step_many(self, batches: List[Any], batch_idx: int = 0):
        """test_step for a list of test batches (one video each): the per-video dicts test_epoch_end consumes"""
        return [{"abnormal_scores": s, "labels": l, "class_probs": p} for s, l, p in self.score_videos(batches)]

    def validation_step(self, batch: Any, batch_idx: int = 0):
        save_dir = _get(self.hparams, "save_dir")
        f = Path(save_dir) / "ncentroid.pt" if save_dir else None
        if f is not None and f.is_file():
            # re-read only when the file changed (the reference reloads it for every validation video, :305-311): a fresh
            # CPU tensor per video would cost a blocking pageable H2D copy per forward and re-key the temporal graphs
            st = f.stat()
            stamp = (str(f), st.st_mtime_ns, st.st_size)
            if self.ncentroid is None or getattr(self, "_ncentroid_stamp", None) != stamp:
                self._set_ncentroid(torch.load(f, map_location="cpu"))
                self._ncentroid_stamp = stamp
        elif self.ncentroid is None:
            raise FileNotFoundError(f"ncentroid file {f} not found")
        scores, labels, probs = self._score_video(batch)
        # the reference extends its lists frame by frame and stacks at epoch end; one tensor per video is the same data
        self.labels.append(labels)
        self.class_probs.append(probs)
        self.abnormal_scores.append(scores)

    def _evaluate(self, scores, labels, probs, per_frame: bool):
        from . import metrics as M
        dm = self._datamodule()
        C = int(_get(dm, "num_classes", None) or _get(self.hparams, "num_classes", probs.shape[1] + 1))
        normal_idx = int(_get(_get(dm, "hparams"), "normal_id", self.net.normal_id))
        r = M.evaluate(scores, labels, probs, normal_idx, C, per_frame=per_frame)
        self.last_metrics = r
        return r

    def on_validation_epoch_end(self):
        if not self.labels:
            return None
        r = self._evaluate(torch.cat(self.abnormal_scores), torch.cat(self.labels), torch.cat(self.class_probs), False)
        self.log("test/AUC", r["auc_roc"], on_step=False, on_epoch=True, prog_bar=True)
        self.log("test/AP", r["auc_pr"], on_step=False, on_epoch=True, prog_bar=True)
        self.log("test/mAUC", r["mean_mc_auroc"], on_step=False, on_epoch=True, prog_bar=True)
        self.log("test/mAP", r["mean_mc_aupr"], on_step=False, on_epoch=True, prog_bar=True)
        epoch = int(getattr(getattr(self, "trainer", None), "current_epoch", 0) or 0)
        keys = ("auc_roc", "auc_pr", "mean_mc_auroc", "mean_mc_aupr", "mc_auroc", "mc_aupr", "optimal_threshold")
        metrics = {"epoch": epoch, **{k: r[k] for k in keys}}
        save_dir = _get(self.hparams, "save_dir")
        if save_dir:
            Path(save_dir).mkdir(parents=True, exist_ok=True)
            with open(Path(save_dir) / f"metrics_{epoch}.json", "w") as fp:
                json.dump(metrics, fp, indent=4, sort_keys=True)
        self.labels.clear()
        self.class_probs.clear()
        self.abnormal_scores.clear()
        return metrics

    def _run_dir(self, kind: str) -> Path:
        """<logs_root>/<kind>/runs/<name of the checkpoint's parent directory> (:407-409, :594-596)."""
        ckpt_path = Path(str(getattr(getattr(self, "trainer", None), "ckpt_path", None) or "ckpt/last.ckpt"))
        run = os.path.normpath(ckpt_path.parent).split(os.path.sep)[-1]
        d = Path(os.path.join(self._logs_root(), kind, "runs", str(run)))
        d.mkdir(parents=True, exist_ok=True)
        return d

    def on_test_start(self):
        self._load_or_compute_ncentroid(self._run_dir("train"))
        dm = self._datamodule()
        if bool(_get(_get(dm, "hparams"), "visualize", False)):
            raise NotImplementedError("data.visualize=True: the qualitative Visualizer is out of scope of this path")
        self.visualizer = None

    @parallel.rank_zero_only
    def test_step(self, batch: Any, batch_idx: int = 0):
        scores, labels, probs = self._score_video(batch)
        return {"abnormal_scores": scores, "labels": labels, "class_probs": probs}

    @parallel.rank_zero_only
    def test_epoch_end(self, outputs: List[Any]):
        outputs = [o for o in outputs if o is not None]
        r = self._evaluate(torch.cat([o["abnormal_scores"] for o in outputs]), torch.cat([o["labels"] for o in outputs]),
                           torch.cat([o["class_probs"] for o in outputs]), True)
        keys = ("auc_roc", "auc_pr", "mean_mc_auroc", "mean_mc_aupr", "mc_auroc", "mc_aupr", "top1_accuracy",
                "top5_accuracy", "optimal_threshold")
        epoch = int(getattr(getattr(self, "trainer", None), "current_epoch", 0) or 0)
        metrics = {"epoch": epoch, **{k: r[k] for k in keys}}
        with open(self._run_dir("eval") / "metrics.json", "w") as fp:
            json.dump(metrics, fp, indent=4, sort_keys=True)
        return metrics

    # ------------------------------------------------------------------ optimizer (:693-746)
    def configure_optimizers(self):
        s = _get(self.hparams, "solver", {})
        lr = _get(s, "lr", 1e-5)
        groups = [
            {"params": list(self.net.selector_model.parameters()), "lr": lr * _get(s, "selector_model_ratio", 1), "name": "selector_model"},
            {"params": list(self.net.temporal_model.parameters()), "lr": lr * _get(s, "temporal_model_ratio", 1), "name": "temporal_model"},
            {"params": list(self.net.prompt_learner.parameters()), "lr": lr * _get(s, "prompt_learner_ratio", 1), "name": "prompt_learner"},
            {"params": [self.net.text_encoder.text_projection], "lr": lr * _get(s, "text_projection_ratio", 1), "name": "text_projection"},
        ]
        opt = self.optimizer(params=groups) if self.optimizer is not None else AcxAdamW(groups, weight_decay=0.2)
        if self.scheduler is None:
            return {"optimizer": opt}
        max_epochs = getattr(getattr(self, "trainer", None), "max_epochs", None) or 50
        successor = torch.optim.lr_scheduler.CosineAnnealingLR(opt, float(max_epochs))
        sch = self.scheduler(optimizer=opt, successor=successor)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sch, "monitor": "train/loss", "interval": "epoch", "frequency": 1}}

    # ------------------------------------------------------------------ one optimisation step of the built-in loop
    def trainable_parameters(self):
        seen, out = set(), []
        for mod in (self.net.selector_model, self.net.temporal_model, self.net.prompt_learner):
            for p in mod.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        out.append(self.net.text_encoder.text_projection)
        return out

    def _make_buckets(self):
        if self._buckets is None:
            # forward order: prompt/text first ... temporal last; buckets fill in reverse
            order = [self.net.prompt_learner.ctx, self.net.text_encoder.text_projection, self.net.selector_model.logit_scale]
            order += list(self.net.temporal_model.parameters())
            # [text_projection, ctx] and the never-used logit_scale get buckets of their own: the step graph exchanges the
            # text gradients on the text stream, ahead of the temporal model's buckets
            self._buckets = parallel.GradBuckets(order, cuts=[order[2], order[1]])
        return self._buckets

    def _step_graph_for(self, batch, optimizer):
        """The whole-step graph (components/step_graph.TrainStepGraph) for this batch geometry, captured on first use; None
        when the step cannot run that way (training from frames, bf16, foreign optimizer state, capture failure ...) -- the
        caller then takes the autograd path.  `net.step_graph = False` switches it off."""
        net = self.net
        if not getattr(net, "step_graph", True) or not getattr(net, "load_from_features", True) or self.criterion is None:
            return None
        (nf, nl), (af, al) = batch
        if not (torch.is_tensor(af) and af.is_cuda and nf.is_cuda and af.dtype == torch.float32 and nf.dtype == torch.float32
                and af.dim() == 4 and af.shape[1] == 1 and nf.shape[1:] == af.shape[1:] and torch.is_tensor(self.ncentroid)):
            return None
        from .components import step_graph as SG
        key = SG.TrainStepGraph.make_key(self, optimizer, af.shape[0], nf.shape[0], af.shape[2])
        cache = self.__dict__.setdefault("_step_graphs", {})
        if key in cache:
            return cache[key]
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        sg = None
        dist_on = parallel.is_distributed()

        def unavailable(e):
            import warnings
            warnings.warn(f"whole-step training graph unavailable ({type(e).__name__}: {e}); using the autograd path")
            self.step_graph_error = f"{type(e).__name__}: {e}"

        try:
            sg = SG.TrainStepGraph(self, optimizer, af.shape[0], nf.shape[0], af.shape[2])
            # dry-run inputs: this batch and an all-ones mask -- NOT generate_mask(): the host RNG stream must advance exactly
            # once per real step, like the reference's (selector_model.py:101-117)
            ones = torch.ones(sg.B, self.net.selector_model.num_segments)
            sg.load_inputs((af, al), (nf, nl), (ones, ones))
        except Exception as e:  # noqa: BLE001
            unavailable(e)
            sg = None
        if dist_on:
            # all ranks take the same path (the graph path and the autograd path issue different collectives), and they agree
            # BEFORE capture(): its warm-up and capture runs issue real collectives, so a rank that skipped them would leave
            # the others inside an all_gather it never joins
            flag = torch.tensor([1.0 if sg is not None else 0.0], device=self.device)
            parallel.dist.all_reduce(flag, op=parallel.dist.ReduceOp.MIN)
            if float(flag.item()) < 0.5:
                sg = None
        if sg is not None:
            try:
                sg.capture()
            except Exception as e:  # noqa: BLE001
                if dist_on:
                    # the ranks' collective sequences no longer match: there is no safe fallback from here
                    raise RuntimeError(f"whole-step training graph: capture failed on rank {parallel.rank()} inside a "
                                       f"distributed run ({type(e).__name__}: {e}); set net.step_graph = False") from e
                unavailable(e)
                sg = None
        cache[key] = sg
        return sg

    def train_batch(self, batch, optimizer, batch_idx: int = 0) -> torch.Tensor:
        """forward, loss, backward (+ bucketed gradient all-reduce overlapped with it), AdamW: what Lightning's
        automatic optimisation + DDP do around `training_step` (configs/trainer/ddp.yaml).  Default: the whole step replayed
        from HIP graphs (step_graph.TrainStepGraph); automatic fallback: the autograd path below, itself with the text tower /
        temporal model as replayed graphs where they capture, eager otherwise."""
        buckets = self._make_buckets()
        with self._x6_cu_reservation(batch):
            return self._train_batch(batch, optimizer, batch_idx, buckets)

    # The Lightning entry point (automatic optimisation: training_step -> backward -> optimizer.step, all between these two hooks)
    # runs under the SAME per-step options as train_batch: the two entry points give the same bits for the same batch, and the
    # cached graphs (their keys carry the option state) are shared between them.
    def on_train_batch_start(self, batch: Any, batch_idx: int = 0, *unused):
        cm = self._x6_cu_reservation(batch)
        cm.__enter__()
        self.__dict__.setdefault("_x6_batch_cms", []).append(cm)

    def on_train_batch_end(self, outputs: Any = None, batch: Any = None, batch_idx: int = 0, *unused):
        cms = self.__dict__.get("_x6_batch_cms") or []
        if cms:
            cms.pop().__exit__(None, None, None)

    @contextlib.contextmanager
    def _x6_cu_reservation(self, batch):
        """A small rank share (<= X6_RESERVE_ROWS feature rows: 4 or more ranks at the UCF batch) leaves X6_RESERVE_CUS CUs to the
        text stream for the duration of the step (eager launches and graph capture alike -- the grid and the K split are part
        of a captured launch): the head's bf16 x 6 convolutions are persistent one-workgroup-per-CU kernels that hold a CU's
        whole register file for ~90 us at a time, and the text tower's ~160 few-row launches (5-10 us each) otherwise queue
        behind every one of them (emulated world 8: 2.74 -> 2.33 ms per step, world 4: 3.89 -> 3.74).  At 16 384 rows (world 2)
        the convolutions are 256 tiles, two rounds on a capped grid (6.04 -> 6.47 ms): there the partly filled last round is
        K-split as well (ACX_OPT_X6_TAIL_SPLIT for the duration of the step: 6.14 -> 5.88 ms); at 32 768 rows the two cancel
        (10.85 -> 10.83) and nothing is reserved.  The choice depends on the batch geometry only: every rank makes the same
        one, the step graph and the autograd path stay bit-identical."""
        (nf, _), (af, _) = batch
        dev = af.device
        rows = (nf.numel() + af.numel()) // max(1, int(af.shape[-1]))            # feature rows of this rank's share
        reserve = dev.type == "cuda" and getattr(self.net, "precision", "auto") == "auto" and rows <= X6_RESERVE_ROWS
        tail = reserve and rows > X6_SPLIT_ROWS
        depth = self.__dict__.get("_x6_res_depth", 0)             # re-entrant: train_batch inside the Lightning hooks
        self.__dict__["_x6_res_depth"] = depth + 1
        reserve = reserve and depth == 0
        if reserve:
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            ops.set_x6_cus(dev.index or 0, max(1, ncu - X6_RESERVE_CUS))
            if tail:
                ops.set_x6_tail_split(dev.index or 0, True)
        try:
            yield
        finally:
            self.__dict__["_x6_res_depth"] = depth
            if reserve:
                ops.set_x6_cus(dev.index or 0, 0)
                if tail:
                    ops.set_x6_tail_split(dev.index or 0, False)

    def _train_batch(self, batch, optimizer, batch_idx, buckets) -> torch.Tensor:
        sg = self._step_graph_for(batch, optimizer)
        if sg is not None:
            (nf, nl), (af, al) = batch
            sg.load_inputs((af, al), (nf, nl), self.net.selector_model.generate_mask(sg.B))
            if self._meters.total is not sg.meter_sum:
                self._meters.attach(sg.meter_sum)        # before the step: the graph adds this step's terms to it
            losses = sg.step()
            self._after_step_graph(sg, losses)
            return losses[0].detach()
        tm = self.net.temporal_model
        tm.__dict__["_grad_sink"] = buckets        # graph-replayed temporal backward: one accumulate launch, this step only
        try:
            buckets.zero()
            with torch.enable_grad():
                loss = self.training_step(batch, batch_idx)["loss"]
                loss.backward()
            buckets.finish()
        finally:
            tm.__dict__.pop("_grad_sink", None)
            buckets._armed = False
        optimizer.step()
        return loss.detach()

    def _after_step_graph(self, sg, losses):
        vals = tuple(losses.unbind(0))
        self.last_losses = vals
        self._meters.count += 1                  # the sums are accumulated inside the graph
        for name in _LOSS_NAMES:
            self.log("train/" + ("loss" if name == "train_loss" else name), getattr(self, name), on_step=False, on_epoch=True,
                     prog_bar=True)
