"""Checkpoint / resume for the diffusion trainer.

The reference only pickles the module on the best validation epoch (`torch.save(model, .../model.pth)`,
main.py:373-375) and cannot resume: the AdamW moments and the importance-sampling history
(`Lt_history`, `Lt_count`) are lost.  `save_checkpoint` stores everything a bit-exact resume needs:
model `state_dict` (reference key names, so it also loads into the reference's DNN), optimizer state,
diffusion history, and the Philox stream position of the denoiser engine."""
import torch


def diffusion_state(diffusion):
    return {"Lt_history": diffusion.Lt_history.detach().cpu().clone(),
            "Lt_count": diffusion.Lt_count.detach().cpu().clone(),
            "ts_calls": int(getattr(diffusion, "_ts_calls", 0)), "q_calls": int(getattr(diffusion, "_q_calls", 0))}


def load_diffusion_state(diffusion, state):
    diffusion.Lt_history.copy_(state["Lt_history"].to(diffusion.Lt_history.device))
    diffusion.Lt_count.copy_(state["Lt_count"].to(diffusion.Lt_count.device))
    diffusion._ts_calls = int(state.get("ts_calls", 0))
    diffusion._q_calls = int(state.get("q_calls", 0))


def save_checkpoint(path, model, diffusion=None, optimizer=None, epoch=0, extra=None):
    eng = model._engine
    ckpt = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "epoch": int(epoch),
            "engine": {"seed": eng.seed, "offset": eng.offset} if eng is not None else None,
            "diffusion": diffusion_state(diffusion) if diffusion is not None else None,
            "optimizer": optimizer.state_dict() if optimizer is not None else None, "extra": extra}
    torch.save(ckpt, path)


def load_checkpoint(path, model, diffusion=None, optimizer=None, map_location=None):
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["model"])
    if ckpt.get("engine") is not None:
        model.engine.seed, model.engine.offset = ckpt["engine"]["seed"], ckpt["engine"]["offset"]
    if diffusion is not None and ckpt.get("diffusion") is not None:
        load_diffusion_state(diffusion, ckpt["diffusion"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return ckpt["epoch"], ckpt.get("extra")
