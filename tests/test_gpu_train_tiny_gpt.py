"""Drop-in on a real nn.Module: a tiny GPT (tied embedding, 2 pre-norm blocks, causal self-attention) trained with
psgd_torch_amd.KWNS4 exactly the way misc/gpt2.py drives the reference optimizer (zero_grad / backward / step, bf16
preconditioner, fp32 parameters) -- tensors of every kind the wrapper meets there: 2-D weights with one and with two dense
factors, a (vocab, d) embedding whose long side is diagonal, 1-D gains and biases.  Checks: the loss on a learnable synthetic task
drops, every parameter and preconditioner factor stays finite, the state layout the reference exposes is there, and a run with the
preconditioner in fp32 follows the bf16 run closely (the whitening, not rounding noise, drives the trajectory)."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
V, D, H, T, B = 96, 64, 4, 32, 16


class Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.ln1, self.ln2 = nn.LayerNorm(D), nn.LayerNorm(D)
        self.qkv, self.proj = nn.Linear(D, 3 * D), nn.Linear(D, D)
        self.fc, self.out = nn.Linear(D, 4 * D), nn.Linear(4 * D, D)

    def forward(self, x):
        b, t, _ = x.shape
        q, k, v = self.qkv(self.ln1(x)).view(b, t, 3, H, D // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(b, t, D)
        x = x + self.proj(a)
        return x + self.out(F.gelu(self.fc(self.ln2(x))))


class TinyGPT(nn.Module):
    def __init__(self):
        super().__init__()
        self.wte, self.wpe = nn.Embedding(V, D), nn.Embedding(T, D)
        self.blocks = nn.ModuleList([Block(), Block()])
        self.lnf = nn.LayerNorm(D)

    def forward(self, idx):
        x = self.wte(idx) + self.wpe(torch.arange(idx.shape[1], device=idx.device))
        for blk in self.blocks:
            x = blk(x)
        return self.lnf(x) @ self.wte.weight.t()          # tied head (misc/gpt2.py:252-254)


def _batch(gen):
    start = torch.randint(0, V, (B, 1), generator=gen)
    step = torch.randint(1, 4, (B, 1), generator=gen)
    seq = (start + step * torch.arange(T + 1)) % V          # arithmetic progressions mod V: learnable from two tokens of context
    return seq[:, :-1].to(DEV), seq[:, 1:].to(DEV)


def _train(pd, steps=120):
    import psgd_torch_amd
    torch.manual_seed(0)
    model = TinyGPT().to(DEV)
    opt = psgd_torch_amd.KWNS4(model.parameters(), lr_params=3e-3, preconditioner_dtype=pd, weight_decay=0.0, seed=1)
    gen = torch.Generator().manual_seed(1)
    losses = []
    for _ in range(steps):
        x, y = _batch(gen)
        loss = F.cross_entropy(model(x).reshape(-1, V), y.reshape(-1))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return model, opt, losses


def test_tiny_gpt_trains_with_kwns4():
    model, opt, losses = _train(torch.bfloat16)
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    # (observed on MI355X: 39.7 at the start -- unit-variance embeddings through a tied head -- 8.2 / 2.0 / 0.76 / 0.26 / 0.20 / 0.11
    #  after 20 / 40 / ... / 120 steps, the same to two digits with an fp32 preconditioner)
    assert first > math.log(V) and last < 0.05 * first, (first, last)        # learns the progressions
    for p in model.parameters():
        assert bool(torch.isfinite(p).all())
        st = opt.state[p]
        assert st["step"] == len(losses) and len(st["QL"]) == 2             # the reference's state layout (..._ddp.py:131-143)
        for q in st["QL"][0]:
            assert bool(torch.isfinite(q.float()).all())
    kinds = {tuple(p.shape): [q.dim() for q in opt.state[p]["QL"][0]] for p in model.parameters()}
    assert kinds[(V, D)] == [1, 2] and kinds[(D, D)] == [2, 2] and kinds[(4 * D, D)] == [1, 2] and kinds[(D,)] == [1], kinds
    info = next(iter(opt._buckets.values())).engine.info()
    assert info["nlb_fallbacks"] == 0, info
    # the same run with an fp32 preconditioner ends at a similar loss: the trajectory is driven by the whitening, not by bf16 noise
    _, _, losses32 = _train(torch.float32)
    last32 = sum(losses32[-5:]) / 5
    assert last32 < 0.05 * first and abs(last32 - last) < 0.02 * first, (last, last32)
