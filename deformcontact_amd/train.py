"""Training step / loop with the reference's step maths (``/root/reference/train.py:46-58,71-73``).

    pred = model(rest, rigid);  pred.pos -= rest.pos;  target.pos = def.pos - rest.pos
    loss = L1(pred.pos, target.pos) + lambda_gradient * GradientConsistency(pred, target)
    Adam(lr = 4e-4)

No wandb (hard dependency of the reference's loop): scalars go to a JSONL file, and are only
pulled off the device when they are logged (the reference does 7 ``.item()`` syncs per step,
``train.py:60-70``).  With ``torch.distributed`` initialised the gradients are averaged through
one flat bucket (``deformcontact_amd.dp``).
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dp
from .graphnet import EVERYDAY_NETWORK, gradient_consistency_loss, load_model
from .loaders import PrefetchLoader, SyntheticEverydayDataset, iterate_batches, to_batches


def losses(model, rest, deff, rig, lambda_gradient: float = 1.0) -> Dict[str, torch.Tensor]:
    pred = model(rest, rig)
    pred.pos = pred.pos - rest.pos
    tgt = deff.clone()
    tgt.pos = deff.pos - rest.pos
    l1 = F.l1_loss(pred.pos, tgt.pos)
    gcl = gradient_consistency_loss(pred, tgt)
    return {"loss": l1 + lambda_gradient * gcl, "l1": l1, "consistency": gcl}


def train_step(model, optimizer, rest, deff, rig, lambda_gradient: float = 1.0,
               bucket: Optional[dp.GradBucket] = None) -> Dict[str, torch.Tensor]:
    out = losses(model, rest, deff, rig, lambda_gradient)
    if bucket is not None:
        bucket.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    out["loss"].backward()
    if bucket is not None:
        bucket.all_reduce_mean()
    optimizer.step()
    return {k: v.detach() for k, v in out.items()}


def train(network_cfg=None, *, device="cuda", epochs: int = 2, num_train: int = 64, num_val: int = 16,
          batch_size: int = 4, lr: float = 4e-4, lambda_gradient: float = 1.0, out_dir: str = "runs/dc",
          soft_vertices: int = 1024, sphere_resolution: int = 20, log_every: int = 1, seed: int = 0,
          conv_module=None):
    """Synthetic-data counterpart of ``train.py:15-135`` (best-val checkpoint + config dump)."""
    cfg = dict(EVERYDAY_NETWORK if network_cfg is None else network_cfg)
    torch.manual_seed(seed)
    model = load_model(cfg, conv_module=conv_module).to(device)
    dp.broadcast_parameters(model)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    bucket = dp.GradBucket(model.parameters()) if dp.world_size() > 1 else None
    rank = torch.distributed.get_rank() if dp.world_size() > 1 else 0
    train_ds = SyntheticEverydayDataset(num_train, rank * num_train, soft_vertices, sphere_resolution)
    val_ds = SyntheticEverydayDataset(num_val, 10_000_000, soft_vertices, sphere_resolution)
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, f"log_rank{rank}.jsonl"), "a")
    best, step = float("inf"), 0
    for epoch in range(epochs):
        model.train()
        # worker thread assembles + pins the next batches, uploads overlap the current step
        for collated, (rest, deff, rig) in PrefetchLoader(train_ds, batch_size, device, shuffle=True,
                                                          seed=seed + epoch):
            out = train_step(model, opt, rest, deff, rig, lambda_gradient, bucket)
            if step % log_every == 0:
                log.write(json.dumps({"step": step, "epoch": epoch, "t": time.time(),
                                      **{k: float(v) for k, v in out.items()}}) + "\n")
            step += 1
        model.eval()
        tot, nb = 0.0, 0
        with torch.no_grad():
            for collated in iterate_batches(val_ds, batch_size):
                rest, deff, rig = to_batches(collated, device)
                pred = model(rest, rig)                     # validation on absolute positions
                val = F.l1_loss(pred.pos, deff.pos) + lambda_gradient * gradient_consistency_loss(pred, deff)
                tot, nb = tot + float(val), nb + 1
        val_loss = tot / max(nb, 1)
        log.write(json.dumps({"epoch": epoch, "validation_loss": val_loss}) + "\n")
        log.flush()
        if rank == 0 and val_loss < best:
            best = val_loss
            torch.save(model.state_dict(), os.path.join(out_dir, "model_weights.pth"))
            with open(os.path.join(out_dir, "config.json"), "w") as f:
                json.dump({"network": cfg, "training": {"learning_rate": lr,
                                                        "lambda_gradient": lambda_gradient}}, f, indent=4)
    log.close()
    return model, best


if __name__ == "__main__":
    train()
