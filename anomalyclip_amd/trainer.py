"""Minimal stand-in for `pytorch_lightning.Trainer` (not installed in this image): drives an AnomalyCLIPModule through
the SAME hooks, in Lightning 1.8's order, that `src/train.py:71-105` / `src/eval.py:73` reach through `trainer.fit` /
`trainer.test`.  One process per GPU (launch with torch.distributed.run for data parallelism: gradients are exchanged
by AnomalyCLIPModule.train_batch through parallel.GradBuckets, SyncBN statistics inside the selector).  Nothing here is
arithmetic -- it is the thin loop SURVEY.md section 7 step 3 asks for so that the module's hooks can keep the
reference's Trainer-called signatures.

Attributes the module reads: `datamodule`, `current_epoch`, `max_epochs`, `ckpt_path`."""
from __future__ import annotations

import os
from typing import Any, Optional

import torch

from . import checkpoint, parallel


def _to_device(x: Any, dev):
    if torch.is_tensor(x):
        return x.to(dev, non_blocking=True)
    if isinstance(x, (list, tuple)):
        return type(x)(_to_device(v, dev) for v in x)
    return x


class Trainer:
    def __init__(self, max_epochs: int = 50, min_epochs: int = 1, check_val_every_n_epoch: int = 1,
                 default_root_dir: Optional[str] = None, limit_train_batches: Optional[int] = None, **ignored):
        # accepted and ignored like any Trainer kwarg that has no meaning here: accelerator, devices, strategy,
        # sync_batchnorm (always on under DP), num_sanity_val_steps, deterministic, callbacks, logger ...
        self.max_epochs, self.min_epochs = int(max_epochs), int(min_epochs)
        self.check_val_every_n_epoch = int(check_val_every_n_epoch)
        self.default_root_dir = default_root_dir
        self.limit_train_batches = limit_train_batches
        self.datamodule = None
        self.current_epoch = 0
        self.global_step = 0
        self.ckpt_path: Optional[str] = None
        self.callback_metrics: dict = {}

    # ------------------------------------------------------------------------------------------------------
    def _attach(self, model, datamodule, stage):
        self.datamodule = datamodule
        object.__setattr__(model, "trainer", self)
        if hasattr(datamodule, "setup"):
            datamodule.setup(stage)

    def _validate(self, model, dev):
        model.net.eval()
        loader = self.datamodule.val_dataloader()
        for i, batch in enumerate(loader):
            model.validation_step(_to_device(batch, dev), i)
        m = model.on_validation_epoch_end()
        if m:
            self.callback_metrics.update({k: v for k, v in m.items() if isinstance(v, (int, float))})

    def save_checkpoint(self, model, path: str):
        """Lightning-shaped `.ckpt` (configs/callbacks/default.yaml:8-14: save_last) on rank 0."""
        if parallel.rank() != 0:
            return
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"state_dict": checkpoint.to_lightning_state_dict(model.net), "epoch": self.current_epoch,
                    "global_step": self.global_step, "hyper_parameters": dict(getattr(model, "hparams", {}) or {}),
                    "pytorch-lightning_version": "1.8.3"}, path)

    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _shard_loader(loader, epoch: int):
        """What Lightning's DDP strategy does to every train DataLoader (replace_sampler_ddp): a DistributedSampler over
        the same dataset -- shuffling iff the original sampler shuffled -- re-seeded per epoch with set_epoch.  Without it
        every rank would iterate the full set: identical batches (redundant compute) or world-times the steps per epoch,
        and the per-epoch LR schedule would no longer line up with the reference's."""
        from torch.utils.data import BatchSampler, DataLoader, DistributedSampler, RandomSampler, SequentialSampler
        if not parallel.is_distributed():
            return loader
        if not isinstance(loader, DataLoader):
            if getattr(loader, "already_sharded", False):       # opt-in: an iterable that yields THIS rank's shard
                return loader
            raise TypeError(f"data-parallel training needs torch DataLoaders to shard (got {type(loader).__name__}); "
                            f"an iterable that already yields this rank's shard may set `already_sharded = True`")
        if isinstance(loader.sampler, DistributedSampler):
            loader.sampler.set_epoch(epoch)
            return loader
        if loader.batch_size is None or not isinstance(loader.batch_sampler, BatchSampler) or not isinstance(
                loader.sampler, (RandomSampler, SequentialSampler)):
            # what Lightning's replace_sampler refuses as well: a custom batch_sampler / sampler cannot be re-created around a
            # DistributedSampler without knowing its constructor
            raise TypeError("cannot shard a DataLoader built with a custom sampler / batch_sampler: construct it with a "
                            "DistributedSampler (it is then only re-seeded per epoch)")
        seed = os.environ.get("PL_GLOBAL_SEED")
        if seed is None:
            # no global seed: every rank must still agree on the permutation -- rank 0's torch seed, shared once per loader
            seed = getattr(loader, "_acx_shard_seed", None)
            if seed is None:
                import torch.distributed as dist
                t = torch.tensor([torch.initial_seed() % (1 << 31)], dtype=torch.int64)
                if dist.get_backend() == "nccl":
                    t = t.cuda()
                dist.broadcast(t, 0)
                seed = int(t.item())
                try:
                    loader._acx_shard_seed = seed
                except Exception:  # noqa: BLE001
                    pass
        sampler = DistributedSampler(loader.dataset, num_replicas=parallel.world_size(), rank=parallel.rank(),
                                     shuffle=isinstance(loader.sampler, RandomSampler), seed=int(seed))
        sampler.set_epoch(epoch)
        kw = dict(batch_size=loader.batch_size, sampler=sampler, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                  pin_memory=loader.pin_memory, drop_last=loader.drop_last, timeout=loader.timeout,
                  worker_init_fn=loader.worker_init_fn, generator=loader.generator,
                  multiprocessing_context=loader.multiprocessing_context)
        if loader.num_workers > 0:
            kw.update(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor)
        return DataLoader(loader.dataset, **kw)

    def _train_batches(self, loaders, epoch: int):
        """Lightning 1.8's default for a LIST of train loaders, multiple_trainloader_mode='max_size_cycle': the epoch has
        max(len) steps and a shorter loader starts over when exhausted (ShanghaiTech: 175 normal vs 63 abnormal videos --
        `zip` would stop at the shortest and run a third of the reference's optimizer steps per epoch)."""
        if not isinstance(loaders, (list, tuple)):
            yield from self._shard_loader(loaders, epoch)
            return
        loaders = [self._shard_loader(l, epoch) for l in loaders]
        its = [iter(l) for l in loaders]
        for _ in range(max(len(l) for l in loaders)):
            batch = []
            for k in range(len(loaders)):
                try:
                    b = next(its[k])
                except StopIteration:
                    its[k] = iter(loaders[k])
                    b = next(its[k])
                batch.append(b)
            yield tuple(batch)

    def fit(self, model, datamodule=None, ckpt_path: Optional[str] = None):
        self._attach(model, datamodule, "fit")
        dev = model.device
        if ckpt_path:
            checkpoint.load_into(model.net, ckpt_path)
            self.ckpt_path = ckpt_path
        cfg = model.configure_optimizers()
        opt = cfg["optimizer"]
        sched = cfg.get("lr_scheduler", {}).get("scheduler") if isinstance(cfg.get("lr_scheduler"), dict) else None
        model.on_train_start()
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            model.net.train()
            loaders = self.datamodule.train_dataloader()        # [normal loader, abnormal loader] (datamodule:144-163)
            for i, batch in enumerate(self._train_batches(loaders, epoch)):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                model.train_batch(_to_device(batch, dev), opt, i)
                self.global_step += 1
            model.on_train_epoch_end()
            if sched is not None:
                sched.step()
            if (epoch + 1) % self.check_val_every_n_epoch == 0 and hasattr(self.datamodule, "val_dataloader"):
                self._validate(model, dev)
            if self.default_root_dir:
                self.ckpt_path = os.path.join(self.default_root_dir, "checkpoints", "last.ckpt")
                self.save_checkpoint(model, self.ckpt_path)
        return self.callback_metrics

    def test(self, model, datamodule=None, ckpt_path: Optional[str] = None):
        self._attach(model, datamodule, "test")
        dev = model.device
        if ckpt_path:
            checkpoint.load_into(model.net, ckpt_path)
            self.ckpt_path = ckpt_path
        model.net.eval()
        model.on_test_start()
        # the reference scores one video per step (batch_size_test: 1); tiles of different videos are independent in
        # evaluation, so `eval_batch_videos` consecutive test batches go through ONE forward (test_step_many) -- bounded by
        # `eval_batch_tiles` 512-frame tiles so that a run of long videos does not blow up the activation memory
        outputs, group, tiles = [], [], 0
        kmax, tmax = int(getattr(self, "eval_batch_videos", 8)), int(getattr(self, "eval_batch_tiles", 96))
        many = getattr(model, "test_step_many", None) if kmax > 1 else None

        def flush():
            if group:
                res = many([_to_device(b, dev) for b in group], len(outputs))
                outputs.extend(res if res is not None else [None] * len(group))
                group.clear()
        for i, batch in enumerate(self.datamodule.test_dataloader()):
            if many is None or not isinstance(batch, (tuple, list)) or len(batch) < 4:
                flush()
                tiles = 0
                outputs.append(model.test_step(_to_device(batch, dev), i))
                continue
            t = int(batch[3])
            if group and (len(group) >= kmax or tiles + t > tmax):
                flush()
                tiles = 0
            group.append(batch)
            tiles += t
        flush()
        m = model.test_epoch_end(outputs)
        return [m] if m is not None else []
