"""A minimal fit loop standing in for pytorch_lightning.Trainer (absent here) with the behaviour the
reference's scripts rely on (scripts/train_model_with_multimodal.py:186-230): automatic optimisation
(training_step -> backward -> [clip] -> optimizer.step), two sanity validation batches, a validation
pass per epoch, top-1 `val_loss` checkpointing (ModelCheckpoint(monitor="val_loss", mode="min")),
EarlyStopping(patience), and `.ckpt` files holding {"state_dict", "optimizer_states", "epoch"}."""
from __future__ import annotations

import json
import os
import time

import torch


class Trainer:
    def __init__(self, max_epochs=1, gradient_clip_val=None, default_root_dir="checkpoints", patience=30,
                 monitor="val_loss", logger_path=None, num_sanity_val_steps=2, device=None, enable_checkpointing=True,
                 sync_every_step=False, deterministic=False):
        """sync_every_step: True reproduces the reference's per-step `loss.item()` host sync (hippie/model.py:114) and
        checks labels on the host every step; False (default) keeps the step asynchronous — hipGraph replays, per-step
        losses kept on the device and averaged at the epoch end, label range errors raised at the epoch end."""
        self.sync_every_step = sync_every_step
        self.deterministic = deterministic      # Lightning's flag: bit-reproducible runs (ordered weight-gradient sums, no fp32 atomics)
        self.max_epochs, self.gradient_clip_val = max_epochs, gradient_clip_val
        self.root, self.patience, self.monitor = default_root_dir, patience, monitor
        self.logger_path = logger_path
        self.num_sanity_val_steps = num_sanity_val_steps
        self.device = torch.device(device) if device is not None else None
        self.enable_checkpointing = enable_checkpointing
        self.best_model_path, self.best_score = "", float("inf")
        self.current_epoch, self.global_step = 0, 0
        self.history = []

    def _to_device(self, batch, device):
        return tuple(t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def _log(self, rec):
        self.history.append(rec)
        if self.logger_path:
            with open(self.logger_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    def validate(self, module, loader, limit=None):
        module.eval()
        losses = []
        for i, batch in enumerate(loader):
            if limit is not None and i >= limit:
                break
            loss = module.validation_step(self._to_device(batch, self._dev(module)), i)
            losses.append(loss.value)
        module.on_validation_epoch_end()
        module.model.check_deferred_errors()
        module.train()
        return float(torch.stack(losses).double().mean()) if losses else 0.0

    def _dev(self, module):
        if self.device is not None:
            return self.device
        return module.model._any_engine().device

    def save_checkpoint(self, module, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save({"state_dict": {k: v.cpu() for k, v in module.state_dict().items()},
                    "optimizer_states": [module.optimizer.state_dict()],
                    "epoch": self.current_epoch, "global_step": self.global_step}, path)

    def fit(self, module, train_dataloaders, val_dataloaders=None):
        module.trainer = self
        if self.deterministic and not module.model.deterministic:
            module.model.deterministic = True
            module._apply_cfg()
        module.set_gradient_clip(self.gradient_clip_val)
        # The asynchronous mode (device-side label flag, per-step losses kept on the device) is a property of THIS loop,
        # which raises the deferred errors at every epoch end.  Outside it — `model(...)`, `get_embeddings` after fit() —
        # nobody would, so both settings are restored on the way out: labels are range-checked on the host again
        # (IndexError at once, like nn.Embedding, hippie/model.py:65-66).
        saved = (module.sync_every_step, module.model.label_check)
        module.sync_every_step = self.sync_every_step
        module.model.label_check = "sync" if self.sync_every_step else "deferred"
        try:
            return self._fit(module, train_dataloaders, val_dataloaders)
        finally:
            module.sync_every_step, module.model.label_check = saved
            module.model.check_deferred_errors()

    def _fit(self, module, train_dataloaders, val_dataloaders):
        dev = self._dev(module)
        if val_dataloaders is not None and self.num_sanity_val_steps:
            self.validate(module, val_dataloaders, self.num_sanity_val_steps)
        bad_epochs = 0
        for epoch in range(self.max_epochs):
            self.current_epoch = module.current_epoch = epoch
            module.train()
            t0 = time.perf_counter()
            n = 0
            for i, batch in enumerate(train_dataloaders):
                batch = self._to_device(batch, dev)
                module.optimizer.zero_grad()
                loss = module.training_step(batch, i)
                loss.backward()
                module.optimizer.step()
                self.global_step += 1
                n += batch[0].shape[0]
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            module.model.check_deferred_errors()
            module.on_train_epoch_end()
            rec = {"epoch": epoch, "train_samples_per_s": n / dt}
            rec.update({k: float(v) for k, v in module.logged.items() if k.startswith("train")})
            if val_dataloaders is not None:
                val = self.validate(module, val_dataloaders)
                rec[self.monitor] = val
                if val < self.best_score:
                    self.best_score, bad_epochs = val, 0
                    if self.enable_checkpointing:
                        self.best_model_path = os.path.join(self.root, f"epoch={epoch}-step={self.global_step}.ckpt")
                        self.save_checkpoint(module, self.best_model_path)
                else:
                    bad_epochs += 1
            self._log(rec)
            if val_dataloaders is not None and bad_epochs >= self.patience:
                break
        return self
