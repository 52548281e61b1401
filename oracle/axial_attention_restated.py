"""Restatement of the third-party `axial_attention` package (lucidrains/axial-attention).

TEST INFRASTRUCTURE ONLY.  Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import anything under oracle/.

Why this file exists: the reference imports `AxialImageTransformer` from the PyPI package
`axial_attention` (reference src/models/components/temporal_model.py:1, ctor call :32-39,
forward call :64-66).  The dependency is UNPINNED in the reference (requirements.txt:30 lists
the bare name), its source is not under /root/reference, it is not installed in this image and
cannot be downloaded.  The reference has no test that pins results at this boundary.

    ==> PARITY UNPINNED for everything computed here.

This module restates the published algorithm of upstream v0.6.x from its documented structure
(module/attribute names are kept so a reference checkpoint's state_dict keys would line up:
`pos_emb.param_{0,1}`, `layers.blocks.{i}.{f,g}.net...`).  It is injected as `axial_attention`
into sys.modules by tests/golden/ref_harness.py so the reference's own TemporalModel /
AnomalyCLIP can be executed unmodified in the development container to produce golden vectors.

Algorithm (per upstream):
  * AxialPositionalEmbedding: one randn parameter per axial dim, broadcast-added in order.
  * calculate_permutations(2, emb_dim=1): [0,3,2,1] (attend along dim 2) then [0,2,3,1]
    (attend along dim 3).
  * PermuteToFrom(perm, PreNorm(dim, SelfAttention(dim, heads, dim_heads)))
  * SelfAttention: to_q (no bias), to_kv (no bias, chunk 2), softmax(q k^T e^-0.5) v, to_out (bias)
  * feed-forward: ChanLayerNorm -> Conv2d(d,4d,3,pad 1) -> LeakyReLU -> Conv2d(4d,d,3,pad 1)
  * ChanLayerNorm: (x-mean)/(sqrt(var_biased)+eps)*g+b  (eps added to the std)
  * ReversibleSequence: x1=x2=x; per block y1=x1+f(x2), y2=x2+g(y1); result mean(x1,x2).
    (upstream recomputes activations in backward; a stored-activation backward is equivalent
    up to fp round-off, so plain autograd is used here.)
"""
import torch
from torch import nn


def calculate_permutations(num_dimensions, emb_dim):
    total = num_dimensions + 2
    emb_dim = emb_dim if emb_dim > 0 else emb_dim + total
    axial_dims = [i for i in range(1, total) if i != emb_dim]
    perms = []
    for axial_dim in axial_dims:
        last_two = [axial_dim, emb_dim]
        rest = sorted(set(range(total)) - set(last_two))
        perms.append([*rest, *last_two])
    return perms


class ChanLayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))

    def forward(self, x):
        std = torch.var(x, dim=1, unbiased=False, keepdim=True).sqrt()
        mean = torch.mean(x, dim=1, keepdim=True)
        return (x - mean) / (std + self.eps) * self.g + self.b


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.fn(self.norm(x))


class PermuteToFrom(nn.Module):
    def __init__(self, permutation, fn):
        super().__init__()
        self.fn = fn
        inv = [0] * len(permutation)
        for i, p in enumerate(permutation):
            inv[p] = i
        self.permutation = permutation
        self.inv_permutation = inv

    def forward(self, x, **kwargs):
        axial = x.permute(*self.permutation).contiguous()
        shape = axial.shape
        *_, t, d = shape
        axial = axial.reshape(-1, t, d)
        axial = self.fn(axial, **kwargs)
        axial = axial.reshape(*shape)
        return axial.permute(*self.inv_permutation).contiguous()


class SelfAttention(nn.Module):
    def __init__(self, dim, heads, dim_heads=None):
        super().__init__()
        self.dim_heads = (dim // heads) if dim_heads is None else dim_heads
        hidden = self.dim_heads * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, hidden, bias=False)
        self.to_kv = nn.Linear(dim, 2 * hidden, bias=False)
        self.to_out = nn.Linear(hidden, dim)

    def forward(self, x, kv=None):
        kv = x if kv is None else kv
        q, k, v = (self.to_q(x), *self.to_kv(kv).chunk(2, dim=-1))
        b, t, d, h, e = *q.shape, self.heads, self.dim_heads

        def merge(z):
            return z.reshape(b, -1, h, e).transpose(1, 2).reshape(b * h, -1, e)

        q, k, v = map(merge, (q, k, v))
        dots = torch.einsum("bie,bje->bij", q, k) * (e ** -0.5)
        dots = dots.softmax(dim=-1)
        out = torch.einsum("bij,bje->bie", dots, v)
        out = out.reshape(b, h, -1, e).transpose(1, 2).reshape(b, -1, d)
        return self.to_out(out)


class AxialPositionalEmbedding(nn.Module):
    def __init__(self, dim, shape, emb_dim_index=1):
        super().__init__()
        total = len(shape) + 2
        ax_idx = [i for i in range(1, total) if i != emb_dim_index]
        self.num_axials = len(shape)
        for i, (axial_dim, axial_dim_index) in enumerate(zip(shape, ax_idx)):
            shp = [1] * total
            shp[emb_dim_index] = dim
            shp[axial_dim_index] = axial_dim
            setattr(self, f"param_{i}", nn.Parameter(torch.randn(*shp)))

    def forward(self, x):
        for i in range(self.num_axials):
            x = x + getattr(self, f"param_{i}")
        return x


class _Wrap(nn.Module):
    """upstream wraps f/g in a `Deterministic` module whose child is `net`."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, *a, **k):
        return self.net(*a, **k)


class ReversibleBlock(nn.Module):
    def __init__(self, f, g):
        super().__init__()
        self.f = _Wrap(f)
        self.g = _Wrap(g)

    def forward(self, x):
        x1, x2 = torch.chunk(x, 2, dim=1)
        y1 = x1 + self.f(x2)
        y2 = x2 + self.g(y1)
        return torch.cat([y1, y2], dim=1)


class ReversibleSequence(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList([ReversibleBlock(f, g) for (f, g) in blocks])

    def forward(self, x):
        x = torch.cat((x, x), dim=1)
        for blk in self.blocks:
            x = blk(x)
        return torch.stack(x.chunk(2, dim=1)).mean(dim=0)


class Sequential(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = blocks

    def forward(self, x):
        for f, g in self.blocks:
            x = x + f(x)
            x = x + g(x)
        return x


class AxialImageTransformer(nn.Module):
    def __init__(self, dim, depth, heads=8, dim_heads=None, dim_index=1, reversible=True,
                 axial_pos_emb_shape=None):
        super().__init__()
        permutations = calculate_permutations(2, dim_index)

        def get_ff():
            return nn.Sequential(
                ChanLayerNorm(dim),
                nn.Conv2d(dim, dim * 4, 3, padding=1),
                nn.LeakyReLU(inplace=True),
                nn.Conv2d(dim * 4, dim, 3, padding=1),
            )

        self.pos_emb = (
            AxialPositionalEmbedding(dim, axial_pos_emb_shape, dim_index)
            if axial_pos_emb_shape is not None else nn.Identity()
        )
        layers = nn.ModuleList([])
        for _ in range(depth):
            attn = nn.ModuleList([
                PermuteToFrom(p, PreNorm(dim, SelfAttention(dim, heads, dim_heads)))
                for p in permutations
            ])
            conv = nn.ModuleList([get_ff(), get_ff()])
            layers.append(attn)
            layers.append(conv)
        self.layers = (ReversibleSequence if reversible else Sequential)(layers)

    def forward(self, x):
        x = self.pos_emb(x)
        return self.layers(x)
