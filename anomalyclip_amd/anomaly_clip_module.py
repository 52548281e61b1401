"""AnomalyCLIPModule -- host-side mirror of the reference's LightningModule
(src/models/anomaly_clip_module.py:31-750) for the hot path: same constructor (`net, optimizer, scheduler,
loss, **kwargs` with kwargs = num_classes / solver / save_dir), same hook names and batch layouts
(`model_step`, `training_step`, `validation_step`, `test_step`, `on_train_start`, `configure_optimizers`).

pytorch_lightning is not part of this image; the class derives from LightningModule when it is importable
and from torch.nn.Module otherwise, and `fit_epoch` / `test_epoch` provide the minimal loop the reference
gets from `Trainer` (one process per GPU; gradients exchanged through parallel.GradBuckets).
`test_epoch_end` / `on_validation_epoch_end` compute the reference's metrics.json numbers (AUROC, AP, mAUC, mAP,
top-1/5, optimal threshold; :339-404, :501-626) with libacx's sort/scan kernels (metrics.py); plots are out of
scope."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Optional

import torch

from . import ops, parallel
from .optim import AcxAdamW

try:  # pragma: no cover - not installed in the build image
    from pytorch_lightning import LightningModule as _Base
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module


class AnomalyCLIPModule(_Base):
    def __init__(self, net: torch.nn.Module, optimizer=None, scheduler=None, loss=None, **kwargs):
        super().__init__()
        self.net = net
        self.criterion = loss
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.hparams_ = dict(kwargs)
        # freezing backbone (anomaly_clip_module.py:68-74)
        for p in self.net.image_encoder.parameters():
            p.requires_grad = False
        for p in self.net.text_encoder.parameters():
            p.requires_grad = False
        self.net.text_encoder.text_projection.requires_grad = True
        for p in self.net.token_embedding.parameters():
            p.requires_grad = False
        self.ncentroid: Optional[torch.Tensor] = None
        self.labels, self.abnormal_scores, self.class_probs = [], [], []
        self._buckets: Optional[parallel.GradBuckets] = None

    # ------------------------------------------------------------------ forward (anomaly_clip_module.py:118-132)
    def forward(self, image_features, labels, ncentroid, segment_size: int = 1, test_mode: bool = False):
        return self.net(image_features, labels, ncentroid, segment_size, test_mode)

    # ------------------------------------------------------------------ ncentroid (:134-171, :406-445)
    @torch.no_grad()
    def compute_ncentroid(self, loader, load_from_features: bool = True) -> torch.Tensor:
        """mean feature over every frame of every normal training video; under data parallelism each rank
        scans its shard and (sum, count) is all-reduced."""
        dev = next(self.net.temporal_model.parameters()).device
        D = self.net.embedding_dim
        acc = torch.zeros(D, dtype=torch.float32, device=dev)
        count = 0
        for batch in loader:
            feats, nlabels = batch[0], batch[1]
            n = int(torch.as_tensor(nlabels).reshape(-1).shape[0])
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
        self.ncentroid = acc / cnt
        return self.ncentroid

    def on_train_start(self, loader=None, load_from_features: bool = True):
        save_dir = self.hparams_.get("save_dir")
        f = Path(save_dir) / "ncentroid.pt" if save_dir else None
        if f is not None and f.is_file():
            self.ncentroid = torch.load(f)
        elif loader is not None:
            self.compute_ncentroid(loader, load_from_features)
            if f is not None and parallel.rank() == 0:
                f.parent.mkdir(parents=True, exist_ok=True)
                torch.save(self.ncentroid.cpu(), f)

    # ------------------------------------------------------------------ steps
    def model_step(self, batch: Any):
        nbatch, abatch = batch                                        # anomaly_clip_module.py:173-178
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
        return {"loss": losses[0]}

    @torch.no_grad()
    def test_step(self, batch: Any, batch_idx: int = 0):
        image_features, labels = batch[0], batch[1]
        segment_size = batch[3]
        dev = next(self.net.temporal_model.parameters()).device
        image_features = image_features.to(dev)
        labels = torch.as_tensor(labels).squeeze(0).to(dev)
        similarity, abnormal_scores = self.forward(image_features, labels, self.ncentroid, int(segment_size), test_mode=True)
        class_probs = ops.class_probs(similarity.contiguous(), abnormal_scores.contiguous())   # :474-477
        n = labels.shape[0]                                           # remove padded frames (:480-483)
        return {"abnormal_scores": abnormal_scores[:n], "labels": labels, "class_probs": class_probs[:n]}

    validation_step = test_step

    # ------------------------------------------------------------------ metrics epilogue (:339-404, :501-626)
    def test_epoch_end(self, outputs, save_dir: Optional[str] = None, epoch: int = 0):
        """outputs = list of test_step dicts.  Returns (and optionally writes `metrics.json` with) the reference's
        keys; rank-zero only in the reference (@rank_zero_only, :500)."""
        import json
        from . import metrics as M
        scores = torch.cat([o["abnormal_scores"] for o in outputs])
        labels = torch.cat([o["labels"] for o in outputs])
        probs = torch.cat([o["class_probs"] for o in outputs])
        C = int(self.hparams_.get("num_classes", probs.shape[1] + 1))
        r = M.evaluate(scores, labels, probs, int(self.net.normal_id), C)
        keys = ("auc_roc", "auc_pr", "mean_mc_auroc", "mean_mc_aupr", "mc_auroc", "mc_aupr", "top1_accuracy",
                "top5_accuracy", "optimal_threshold")
        metrics = {"epoch": epoch, **{k: r[k] for k in keys}}
        if save_dir is not None:
            Path(save_dir).mkdir(parents=True, exist_ok=True)
            with open(Path(save_dir) / "metrics.json", "w") as fp:
                json.dump(metrics, fp, indent=4, sort_keys=True)
        self.last_metrics = r
        return metrics

    def on_validation_epoch_end(self, outputs, save_dir: Optional[str] = None, epoch: int = 0):
        """:339-404 -- same numbers minus the per-frame predictions."""
        m = self.test_epoch_end(outputs, None, epoch)
        m = {k: v for k, v in m.items() if k not in ("top1_accuracy", "top5_accuracy")}
        if save_dir is not None:
            import json
            Path(save_dir).mkdir(parents=True, exist_ok=True)
            with open(Path(save_dir) / f"metrics_{epoch}.json", "w") as fp:
                json.dump(m, fp, indent=4, sort_keys=True)
        return m

    # ------------------------------------------------------------------ optimizer (:693-746)
    def configure_optimizers(self, max_epochs: int = 50):
        s = self.hparams_.get("solver", {})
        lr = s.get("lr", 1e-5)
        groups = [
            {"params": list(self.net.selector_model.parameters()), "lr": lr * s.get("selector_model_ratio", 1), "name": "selector_model"},
            {"params": list(self.net.temporal_model.parameters()), "lr": lr * s.get("temporal_model_ratio", 1), "name": "temporal_model"},
            {"params": list(self.net.prompt_learner.parameters()), "lr": lr * s.get("prompt_learner_ratio", 1), "name": "prompt_learner"},
            {"params": [self.net.text_encoder.text_projection], "lr": lr * s.get("text_projection_ratio", 1), "name": "text_projection"},
        ]
        opt = self.optimizer(params=groups) if self.optimizer is not None else AcxAdamW(groups, weight_decay=0.2)
        if self.scheduler is None:
            return {"optimizer": opt}
        successor = torch.optim.lr_scheduler.CosineAnnealingLR(opt, float(max_epochs))
        sch = self.scheduler(optimizer=opt, successor=successor)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sch, "monitor": "train/loss", "interval": "epoch", "frequency": 1}}

    def trainable_parameters(self):
        seen, out = set(), []
        for mod in (self.net.selector_model, self.net.temporal_model, self.net.prompt_learner):
            for p in mod.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        out.append(self.net.text_encoder.text_projection)
        return out

    def train_batch(self, batch, optimizer) -> torch.Tensor:
        """one optimisation step: forward, loss, backward (+ bucketed gradient all-reduce), AdamW."""
        if self._buckets is None:
            # forward order: prompt/text first ... temporal last; buckets fill in reverse
            order = [self.net.prompt_learner.ctx, self.net.text_encoder.text_projection, self.net.selector_model.logit_scale]
            order += list(self.net.temporal_model.parameters())
            self._buckets = parallel.GradBuckets(order)
        self._buckets.zero()
        loss = self.training_step(batch)["loss"]
        loss.backward()
        self._buckets.finish()
        optimizer.step()
        return loss.detach()
