"""Oracle restatement of the ViTPose backbone forward.  TEST INFRASTRUCTURE ONLY.

Follows models/ViTPose/mmpose/models/backbones/vit.py (reference):
PatchEmbed :143-165 (conv k16 s16, padding 4 + 2*(ratio//2 - 1) = 2 for ratio=1),
pos-embed add :320, Attention :78-115, Mlp :61-76, Block :117-140 (pre-LN, eps 1e-6 :212),
last_norm + NCHW permute :328-330.  Functional over a state dict with the
reference's key names (prefix e.g. 'feature_extractor.backbone.').
"""
import torch
import torch.nn.functional as F


def drop_path_schedule(drop_path_rate, depth):
    """vit.py:233: dpr = linspace(0, drop_path_rate, depth) -- block i drops BOTH of its residual branches with probability dpr[i]"""
    return [v.item() for v in torch.linspace(0, drop_path_rate, depth)]


def vit_tokens(sd, x, prefix='', num_heads=12, depth=None, eps=1e-6, taps=None, drop_masks=None, drop_path_rate=0.0):
    """x [B,3,H,W] fp32 -> (tokens [B,N,C] after last_norm, (Hp, Wp)).

    ``taps`` (optional dict) receives intermediates for per-kernel parity tests.
    ``drop_masks`` (training mode, vit.py:132-139 + timm's drop_path): [2*depth, B] 0/1 keep masks, row 2i for the attention branch of
    block i and row 2i+1 for its MLP branch; a kept sample's branch is divided by keep_prob = 1 - dpr[i] (timm.models.layers.drop_path
    [3P timm 0.4.9, restated]: mask = floor(keep_prob + rand(B,1,1)); out = x / keep_prob * mask).  Blocks with dpr[i] == 0 are identities
    (vit.py:132: nn.Identity) and their mask rows are ignored.
    """
    p = prefix
    w = sd[p + 'patch_embed.proj.weight']
    ps = w.shape[-1]
    t = F.conv2d(x, w, sd[p + 'patch_embed.proj.bias'], stride=ps, padding=2)
    B, C, Hp, Wp = t.shape
    t = t.flatten(2).transpose(1, 2)
    pos = sd[p + 'pos_embed']
    t = t + pos[:, 1:] + pos[:, :1]
    if taps is not None:
        taps['embed'] = t
    if depth is None:
        depth = 0
        while (p + 'blocks.%d.norm1.weight' % depth) in sd:
            depth += 1
    hd = C // num_heads
    scale = hd ** -0.5
    N = t.shape[1]
    dpr = drop_path_schedule(drop_path_rate, depth) if drop_masks is not None else None

    def dp(branch, i, which):
        if dpr is None or dpr[i] == 0.0:
            return branch
        keep = 1.0 - dpr[i]
        return branch / keep * drop_masks[2 * i + which].to(branch.dtype).view(B, 1, 1)
    for i in range(depth):
        b = p + 'blocks.%d.' % i
        h = F.layer_norm(t, (C,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'], eps)
        qkv = F.linear(h, sd[b + 'attn.qkv.weight'], sd.get(b + 'attn.qkv.bias'))
        qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * scale, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        h = (a @ v).transpose(1, 2).reshape(B, N, C)
        t = t + dp(F.linear(h, sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias']), i, 0)
        h = F.layer_norm(t, (C,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'], eps)
        h = F.gelu(F.linear(h, sd[b + 'mlp.fc1.weight'], sd[b + 'mlp.fc1.bias']))
        t = t + dp(F.linear(h, sd[b + 'mlp.fc2.weight'], sd[b + 'mlp.fc2.bias']), i, 1)
        if taps is not None and i == 0:
            taps['block0'] = t
    t = F.layer_norm(t, (C,), sd[p + 'last_norm.weight'], sd[p + 'last_norm.bias'], eps)
    return t, (Hp, Wp)


def vit_forward(sd, x, prefix='', num_heads=12, depth=None, drop_masks=None, drop_path_rate=0.0):
    """-> s_feat [B,C,Hp,Wp] (vit.py:330)."""
    t, (Hp, Wp) = vit_tokens(sd, x, prefix, num_heads, depth, drop_masks=drop_masks, drop_path_rate=drop_path_rate)
    B, N, C = t.shape
    return t.permute(0, 2, 1).reshape(B, C, Hp, Wp).contiguous()
