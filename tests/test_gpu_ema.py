"""muse.EMAModel on the HIP path (muse_ema_multi): the weight average that training/train_muse.py:779-780 advances right behind the
optimizer step.  Bit-exact bar: the update is three f32 roundings per element, the reference does the same on its tensors."""
import json
import os

import numpy as np
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_ema_step_bit_identical_to_reference_golden(golden_dir):
    """14 calls of step() on changing parameters under two schedules (update_after_step; warmup + min_decay + update_every 2), a frozen
    tensor among the tracked ones: every shadow tensor after every call equals what the REAL reference class produced, bit for bit
    (tests/golden/ema_tiny.npz), as do the decays"""
    import muse
    g = np.load(os.path.join(golden_dir, "ema_tiny.npz"))
    seed, steps = int(g["seed"]), int(g["steps"])
    for si, kw in enumerate(W.EMA_SCHEDULES):
        params = [torch.nn.Parameter(t.to(DEV)) for t in W.ema_params(seed, 0)]
        params[4].requires_grad_(False)
        ema = muse.EMAModel(params, **kw)
        assert all(s.is_cuda for s in ema.shadow_params)
        for step in range(1, steps + 1):
            with torch.no_grad():
                for p, t in zip(params, W.ema_params(seed, step)):
                    p.copy_(t)
            ema.step(params)
            want = float(g[f"s{si}.decay{step}"])
            if want >= 0:
                assert ema.cur_decay_value == want
            for i, sh in enumerate(ema.shadow_params):
                assert np.array_equal(sh.cpu().numpy(), g[f"s{si}.shadow{step}.{i}"]), (si, step, i)
        assert ema.optimization_step == steps


def test_ema_step_chunks_views_and_frozen_tensors_vs_oracle():
    """the one-launch table form on what a real model hands it: tensors spanning several 4096-element chunks, ragged tails, a 0-dim
    tensor, parameters that are unaligned views into one flat buffer (the flat-buffer MaskGitTransformer), a frozen tensor, an empty one -
    against the pinned numpy oracle, bit for bit, over several steps; the device table is reused while the tensors stay where they are"""
    import muse
    from oracle import ema_oracle as E
    sizes = [4096, 4097, 8200, 1, 3, 12289, 0, 65536 + 5]
    g = torch.Generator().manual_seed(11)
    flat = torch.randn(sum(sizes) + 16, generator=g).to(DEV)
    params, off = [], 1                                    # offset 1: every view starts 4 bytes off a 16-byte boundary
    for n in sizes:
        params.append(torch.nn.Parameter(flat[off:off + n]))
        off += n
    params.append(torch.nn.Parameter(torch.randn((), generator=g).to(DEV)))
    params[2].requires_grad_(False)
    rg = [p.requires_grad for p in params]
    ema = muse.EMAModel(params, decay=0.999, update_after_step=1)
    shadow = [s.cpu().numpy().copy() for s in ema.shadow_params]
    sched = E.Schedule(decay=0.999, update_after_step=1)
    table = None
    for step in range(6):
        with torch.no_grad():
            flat.add_(torch.randn(flat.shape, generator=g).to(DEV) * 0.1)
            params[-1].add_(0.25)
        ema.step(params)
        shadow = E.ema_update(shadow, [p.detach().cpu().numpy() for p in params], rg, sched.next())
        for i, (a, b) in enumerate(zip(ema.shadow_params, shadow)):
            assert np.array_equal(a.cpu().numpy(), b), (step, i)
        if table is None:
            table = ema._pairs["table"]
        assert ema._pairs["table"] is table
    assert ema.cur_decay_value == E.get_decay(6, decay=0.999, update_after_step=1) and ema.cur_decay_value > 0.2
    ema.to(DEV)                                             # moving the shadow drops the table
    assert ema._pairs is None


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_ema_around_a_training_step_and_validation_swap(golden_dir, cd):
    """the sequence of train_muse.py: optimizer step -> ema.step (:779-780), then store / copy_to -> evaluate -> restore (:856-871) on
    muse.MaskGiTUViT with muse.FusedAdamW.  The averaged model must be what a fresh model loaded with the shadow weights computes - in the
    bf16 mode that only holds if the cached bf16 weight copies were refreshed after copy_to / restore (they are keyed on the parameters'
    version counters, which Tensor.copy_ bumps)"""
    import muse
    from oracle import ema_oracle as E
    g = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train().set_compute_dtype(cd)
    ema = muse.EMAModel(model.parameters(), decay=0.9, update_after_step=0, model_cls=muse.MaskGiTUViT, model_config=model.config)
    ema.to(DEV)
    opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=5e-3)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    shadow = [s.cpu().numpy().copy() for s in ema.shadow_params]
    sched = E.Schedule(decay=0.9)
    for _ in range(4):
        _, loss = model(*args, labels=labels)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        ema.step(model.parameters())
        shadow = E.ema_update(shadow, [p.detach().cpu().numpy() for p in model.parameters()], [True] * len(shadow), sched.next())
    assert ema.cur_decay_value == 4 / 13 and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(ema.shadow_params, shadow))
    assert max(float((s - p).abs().max()) for s, p in zip(ema.shadow_params, model.parameters())) > 1e-3      # the average really lags
    live = [p.detach().clone() for p in model.parameters()]
    model.eval()
    with torch.no_grad():
        logits_live = model(*args)
        ema.store(model.parameters())
        ema.copy_to(model.parameters())
        logits_avg = model(*args)
        fresh = muse.MaskGiTUViT(**cfg)
        fresh.load_state_dict({k: s.cpu() for (k, _), s in zip(model.named_parameters(), ema.shadow_params)}, strict=True)
        fresh.to(DEV).eval().set_compute_dtype(cd)
        assert torch.equal(logits_avg, fresh(*args)) and not torch.equal(logits_avg, logits_live)
        ema.restore(model.parameters())
        assert all(torch.equal(a, b) for a, b in zip(live, model.parameters()))
        assert torch.equal(model(*args), logits_live)


def test_offline_ema_script_flow(golden_dir, tmp_path):
    """scripts/compute_offline_ema.py of the reference, statement for statement: the class from the checkpoint's config.json, the first
    checkpoint loaded with from_pretrained(Path).to(device), EMAModel(parameters=, decay=, update_every=interval), ema.to(device), one
    step() per training step with a NEW model object loaded at every checkpoint (other tensors behind the same EMA), copy_to,
    save_pretrained - against the oracle on the same parameter sequence"""
    from pathlib import Path
    import muse
    from muse import EMAModel, MaskGiTUViT, MaskGitTransformer
    from oracle import ema_oracle as E
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    g = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    interval, end_step, decay = 3, 9, 0.95
    root = Path(tmp_path)
    states = {}
    for step in (0, 3, 6, 9):                                    # checkpoint-<n>/unwrapped_model, as the training script writes them
        m = MaskGiTUViT(**cfg)
        m.load_state_dict({k: v * (1.0 + 0.01 * step) for k, v in sd.items()}, strict=True)
        m.save_pretrained(root / f"checkpoint-{step}" / "unwrapped_model")
        states[step] = [p.detach().numpy().copy() for p in m.parameters()]
    dirs = sorted(root.glob("checkpoint-*"), key=lambda p: int(p.name.split("-")[-1]))
    transformer_config = MaskGitTransformer.load_config(dirs[0] / "unwrapped_model")
    model_cls = MaskGitTransformer if transformer_config["_class_name"] == "MaskGitTransformer" else MaskGiTUViT
    assert transformer_config["_class_name"] in ("MaskGiTUViT", "MaskGiTUViT_v2") and model_cls is MaskGiTUViT
    device = "cuda"
    model = model_cls.from_pretrained(dirs[0] / "unwrapped_model").to(device)
    ema_model = EMAModel(parameters=model.parameters(), decay=decay, update_every=interval)
    ema_model.to(device)
    shadow, sched, current = [a.copy() for a in states[0]], E.Schedule(decay=decay, update_every=interval), states[0]
    for step in range(0, end_step):
        if (step + 1) % interval == 0:
            model = model_cls.from_pretrained(root / f"checkpoint-{step + 1}" / "unwrapped_model")
            model.to(device)
            current = states[step + 1]
        ema_model.step(model.parameters())
        d = sched.next()
        if d is not None:
            shadow = E.ema_update(shadow, current, [True] * len(shadow), d)
    assert ema_model.optimization_step == end_step
    for a, b in zip(ema_model.shadow_params, shadow):
        assert np.array_equal(a.cpu().numpy(), b)
    ema_model.copy_to(model.parameters())
    model.save_pretrained(root / "ema")
    back = model_cls.from_pretrained(root / "ema")
    assert all(np.array_equal(p.detach().numpy(), b) for p, b in zip(back.parameters(), shadow))

